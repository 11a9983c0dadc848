"""GPU parity: the HIP path (through the C ABI / VecMapfEnv) against the reference's recorded
outputs (tests/golden) and against the pinned CPU oracles.  Bit-exact: integers, flags and the
float64 reward/prob bit patterns.  All tests here need a real MI355X (-m gpu)."""
import numpy as np
import pytest

import c_oracle
import mapf_oracle as mo
import philox
from conftest import set_tune, load_json
from gym_mapf_amd.envs.grid import MapfGrid
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

pytestmark = pytest.mark.gpu
KERNELS = ('thread_per_env', 'lane_group')


def _families(n_agents):
    """Kernel families that exist for this agent count (thread-per-env is specialised for A <= 16)."""
    return [k for k in KERNELS if k == 'lane_group' or n_agents <= 16]
CRIT = {'Makespan': OptimizationCriteria.Makespan, 'SoC': OptimizationCriteria.SoC}
OCRIT = {'Makespan': mo.MAKESPAN, 'SoC': mo.SOC}


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def _vec(meta, g, sel, env_id_offset=0, **kw):
    return VecMapfEnv(MapfGrid(meta['lines']), meta['n_agents'], g['start_loc'][sel], g['goal_loc'][sel],
                      meta['fail_prob'], meta['r_clash'], meta['r_goal'], meta['r_living'], CRIT[meta['criteria']],
                      seed=meta['seed'], env_id_offset=env_id_offset, **kw)


def _check_step(local, reward, done, info, g, t, sel, tag):
    assert np.array_equal(local, g['next_local'][t][sel]), tag
    assert np.array_equal(_bits(reward), _bits(g['reward'][t][sel])), tag
    assert np.array_equal(_bits(info['prob']), _bits(g['prob'][t][sel])), tag
    assert np.array_equal(done, g['done'][t][sel]), tag
    assert np.array_equal(info['collision'], g['collision'][t][sel]), tag
    assert np.array_equal(info['was_terminal'], g['was_terminal'][t][sel]), tag


def _id_runs(env_ids):
    """Split golden env ids into runs of consecutive ids (one handle per run)."""
    runs, start = [], 0
    ids = [int(x) for x in env_ids]
    for k in range(1, len(ids) + 1):
        if k == len(ids) or ids[k] != ids[k - 1] + 1:
            runs.append((start, k))
            start = k
    return runs


def test_step_with_injected_uniforms_matches_reference(trajectory_set):
    """mapf_step(uniforms=...) fed the exact rand() values the reference consumed."""
    meta, g = trajectory_set
    A, T, E = meta['n_agents'], meta['T'], len(g['env_ids'])
    sel = np.arange(E)
    us = [np.stack([philox.slip_uniforms_np(meta['seed'], [e], t, A)[0] for e in g['env_ids']]) for t in range(T)]
    for kernel in _families(A):
        env = _vec(meta, g, sel, kernel=kernel)
        assert np.array_equal(env.start_local, g['start_local']) and np.array_equal(env.goal_local, g['goal_local'])
        for t in range(T):
            local, reward, done, info = env.step(g['actions'][t], uniforms=us[t], auto_reset=meta['auto_reset'])
            _check_step(local, reward, done, info, g, t, sel, '%s %s t=%d' % (meta['name'], kernel, t))
        env.close()


def test_step_with_device_philox_matches_reference(trajectory_set):
    """mapf_step(uniforms=NULL): the kernel's own Philox4x32-10 draws, keyed by global env id."""
    meta, g = trajectory_set
    for kernel in _families(meta['n_agents']):
        for lo, hi in _id_runs(g['env_ids']):
            sel = np.arange(lo, hi)
            env = _vec(meta, g, sel, env_id_offset=int(g['env_ids'][lo]), kernel=kernel)
            for t in range(meta['T']):
                local, reward, done, info = env.step(g['actions'][t][sel], auto_reset=meta['auto_reset'])
                _check_step(local, reward, done, info, g, t, sel, '%s %s ids[%d:%d] t=%d' % (meta['name'], kernel, lo, hi, t))
            env.close()


def test_fused_rollout_matches_reference(trajectory_set):
    """mapf_rollout: T steps in one launch, recorded trajectory == reference step by step, with
    streamed actions and with the in-kernel policy stream (the goldens' actions are that stream)."""
    meta, g = trajectory_set
    T = meta['T']
    for kernel, (lo, hi) in [(k, r) for k in _families(meta['n_agents']) for r in _id_runs(g['env_ids'])]:
        sel = np.arange(lo, hi)
        # (sets with scripted actions -- the goal-seeking ones -- have no in-kernel counterpart of their action source)
        for actions in (np.ascontiguousarray(g['actions'][:, sel]),) + (() if 'actions_from' in meta else (None,)):
            env = _vec(meta, g, sel, env_id_offset=int(g['env_ids'][lo]), kernel=kernel)
            res = env.rollout(T, actions=actions, auto_reset=meta['auto_reset'], record=True)
            assert np.array_equal(res['local'], g['next_local'][:, sel])
            assert np.array_equal(_bits(res['reward']), _bits(g['reward'][:, sel]))
            assert np.array_equal(_bits(res['prob']), _bits(g['prob'][:, sel]))
            assert np.array_equal(res['done'], g['done'][:, sel])
            assert np.array_equal(res['collision'], g['collision'][:, sel])
            ret = np.zeros(hi - lo)
            for t in range(T):
                ret = ret + g['reward'][t, sel]          # same left-to-right float64 sum
            assert np.array_equal(_bits(res['returns']), _bits(ret))
            assert np.array_equal(res['episodes'], g['done'][:, sel].sum(0))
            assert np.array_equal(res['collisions'], g['collision'][:, sel].sum(0))
            # state after the rollout == state a step-by-step run leaves behind
            state, t_now = env.get_state()
            assert t_now == T
            env.close()


def test_rollout_split_equals_single_and_accumulates(trajectory_set):
    meta, g = trajectory_set
    if not meta['auto_reset']:
        pytest.skip('accumulation check uses the auto-reset sets')
    lo, hi = _id_runs(g['env_ids'])[0]
    sel = np.arange(lo, hi)
    T = meta['T']
    one = _vec(meta, g, sel, kernel='lane_group')
    full = one.rollout(T, auto_reset=True)
    st_full, _ = one.get_state()
    two = _vec(meta, g, sel, kernel=_families(meta['n_agents'])[0])
    part = two.rollout(T // 3, auto_reset=True)
    part = two.rollout(T - T // 3, auto_reset=True, accumulate_into=part)
    st_two, t_two = two.get_state()
    assert t_two == T and np.array_equal(st_full, st_two)
    assert np.array_equal(_bits(full['returns']), _bits(part['returns']))
    assert np.array_equal(full['episodes'], part['episodes']) and np.array_equal(full['collisions'], part['collisions'])
    one.close(), two.close()


def test_rollout_out_reuses_an_earlier_calls_arrays(trajectory_set):
    """rollout(out=previous result): the same arrays are written again (no allocation per call -- what a training loop with
    short rollouts wants), totals OVERWRITTEN; the values are those of a fresh call."""
    meta, g = trajectory_set
    lo, hi = _id_runs(g['env_ids'])[0]
    sel = np.arange(lo, hi)
    T = min(meta['T'], 12)
    auto = bool(meta['auto_reset'])
    a, b = _vec(meta, g, sel), _vec(meta, g, sel)
    first = a.rollout(T, auto_reset=auto, record=True)
    ids = {k: id(v) for k, v in first.items()}
    keep = {k: v.copy() for k, v in first.items()}
    again = a.rollout(T, auto_reset=auto, record=True, out=first)             # steps T .. 2T-1, into the same arrays
    assert again is first and {k: id(v) for k, v in again.items()} == ids
    b.rollout(T, auto_reset=auto, record=True)
    fresh = b.rollout(T, auto_reset=auto, record=True)
    for k in fresh:
        assert np.array_equal(_bits(again[k]) if again[k].dtype == np.float64 else again[k],
                              _bits(fresh[k]) if fresh[k].dtype == np.float64 else fresh[k]), k
    assert any(not np.array_equal(keep[k], again[k]) for k in ('local', 'returns'))
    with pytest.raises(ValueError):
        a.rollout(T, auto_reset=auto, out=first, accumulate_into=first)
    a.close(), b.close()


def test_rollout_out_in_device_mode_reuses_its_argument_block_and_follows_the_arrays():
    """Device-array mode: rollout(out=...) keeps the argument block of the last such call and uses it again while every array
    of the dict is the same object and the actions buffer, T and flags are the same -- each such call against a fresh call
    of a second handle; an array swapped inside the dict, another T, another actions buffer and the in-kernel policy all
    take effect (the block is dropped, never followed into an array the caller no longer passes)."""
    import torch
    rs = np.random.RandomState(77)
    grid = MapfGrid([''.join('@' if rs.rand() < 0.12 else '.' for _ in range(16)) for _ in range(16)])
    V, E, A, T = len(grid.tables()[0]), 4096, 8, 8
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    mk = lambda: VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=9,  # noqa: E731
                            start_local=start, goal_local=goal, device_arrays=True)
    a, b = mk(), mk()
    acts = torch.zeros((T, E, A), dtype=torch.uint8, device='cuda')
    def same(x, y):                                               # bit for bit; the envs run on streams of their own
        a.sync(), b.sync()
        return all(torch.equal(x[k].view(torch.uint8), y[k].view(torch.uint8)) for k in y)

    def fill(t):                                                  # (torch's stream: finished before an env reads the buffer)
        t.copy_(torch.from_numpy(rs.randint(0, 5, size=tuple(t.shape)).astype(np.uint8)))
        torch.cuda.synchronize()
    out = None
    for i in range(4):                                            # calls 1 .. 3 run from the kept block; the buffer's CONTENTS change
        fill(acts)
        out = a.rollout(T, actions=acts, record=True, out=out)
        assert i == 0 or a._rollout_io[0] is out                   # (the first call had no out= to keep a block for)
        assert same(out, b.rollout(T, actions=acts, record=True)), i
    # an array swapped inside the dict: the new one is written, the old one is left alone
    old_reward = out['reward']
    a.sync()
    before = old_reward.clone()
    torch.cuda.synchronize()
    out['reward'] = torch.empty_like(old_reward)
    fill(acts)
    res = a.rollout(T, actions=acts, record=True, out=out)
    assert same(res, b.rollout(T, actions=acts, record=True))
    assert res['reward'] is out['reward'] and torch.equal(old_reward, before)
    # another actions buffer, another T, the in-kernel policy, totals only
    acts2 = torch.from_numpy(rs.randint(0, 5, size=(T, E, A)).astype(np.uint8)).cuda()
    torch.cuda.synchronize()
    assert same(a.rollout(T, actions=acts2, record=True, out=out), b.rollout(T, actions=acts2, record=True))
    half = a.rollout(T // 2, actions=acts2[:T // 2], record=True, out=out)
    assert tuple(half['local'].shape) == (T // 2, E, A) and same(half, b.rollout(T // 2, actions=acts2[:T // 2], record=True))
    for _ in range(2):
        assert same(a.rollout(T // 2, record=True, out=half), b.rollout(T // 2, record=True))
    tot = a.rollout(T, actions=acts2, out={})
    for _ in range(2):
        tot = a.rollout(T, actions=acts2, out=tot)
    for _ in range(3):
        ref = b.rollout(T, actions=acts2)
    assert same(tot, ref) and a.get_state()[1] == b.get_state()[1]
    sa, sb = a.get_state()[0], b.get_state()[0]
    a.sync(), b.sync()
    assert torch.equal(sa.view(torch.uint8), sb.view(torch.uint8))
    a.close(), b.close()
    assert a._rollout_io is None


def test_scripted_edge_cases_match_reference():
    """Hand-picked uniforms: all-False argmax, merges, swap-not-sticky, terminal no-ops, SoC rules,
    exotic fail_prob.  Driven through reset/set_state as the reference run did."""
    for case in load_json('scripted_cases.json'):
        A = len(case['starts'])
        env = VecMapfEnv(MapfGrid(case['lines']), A, case['starts'], case['goals'], case['fail_prob'],
                         case['r_clash'], case['r_goal'], case['r_living'], CRIT[case['criteria']], n_envs=1)
        for k, st in enumerate(case['steps']):
            if st.get('reset'):
                env.reset()
                continue
            local, reward, done, info = env.step(np.asarray([st['actions']], np.uint8),
                                                 uniforms=np.asarray([st['uniforms']]))
            tag = '%s step %d' % (case['name'], k)
            assert local[0].tolist() == st['next_local'], tag
            assert _bits(reward[0]) == _bits(st['reward']) and _bits(info['prob'][0]) == _bits(st['prob']), tag
            assert bool(done[0]) == st['done'], tag
            was_term = bool(info['was_terminal'][0])
            assert (None if was_term else bool(info['collision'][0])) == st['collision'], tag
        env.close()


def test_masked_reset_set_get_state_and_query_terminal():
    lines = ['....', '.@..', '....']
    grid = MapfGrid(lines)
    E, A = 37, 3
    rs = np.random.RandomState(5)
    V = len(grid.tables()[0])
    start = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    goal = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    env = VecMapfEnv(grid, A, None, None, 0.2, -10.0, 5.0, -1.0, OptimizationCriteria.SoC,
                     start_local=start, goal_local=goal)
    st, t = env.get_state()
    assert t == 0 and np.array_equal(st, start)
    new = rs.randint(0, V, size=(E, A)).astype(np.uint16)
    env.set_state(new, t=11)
    st, t = env.get_state()
    assert t == 11 and np.array_equal(st, new)
    term = env.query_terminal()
    expect = np.array([len(set(r.tolist())) < A or np.array_equal(r, gl) for r, gl in zip(new, goal)], np.uint8)
    assert np.array_equal(term, expect)
    mask = (rs.rand(E) < 0.5).astype(np.uint8)
    env.reset(mask)
    st, _ = env.get_state()
    assert np.array_equal(st, np.where(mask[:, None] != 0, start, new))
    env.reset()
    assert np.array_equal(env.get_state()[0], start)
    with pytest.raises(Exception):
        env.set_state(np.full((E, A), V, np.uint16))      # out-of-range cell
    env.close()


@pytest.mark.parametrize('kernel', KERNELS)
@pytest.mark.parametrize('n_agents', list(range(1, 18)) + [24, 29, 31, 32, 33, 47, 64, 65, 100, 128])
def test_every_agent_count_against_c_oracle(n_agents, kernel):
    """Every thread-per-env specialisation (A = 1..16) and every lane-group width (A up to 128, odd
    counts included): 257 envs x 40 steps on a 20% random map against the C oracle (Philox on both
    sides), stepwise and fused."""
    if kernel not in _families(n_agents):
        pytest.skip('thread-per-env kernels exist for A <= 16')
    rs = np.random.RandomState(100 + n_agents)
    H = W = 24
    lines = [''.join('@' if rs.rand() < 0.2 else '.' for _ in range(W)) for _ in range(H)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    V, E, T, A = len(valid), 257, 40, n_agents
    start = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    goal = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    for crit, ocrit, auto in ((OptimizationCriteria.SoC, mo.SOC, True), (OptimizationCriteria.Makespan, mo.MAKESPAN, False)):
        env = VecMapfEnv(grid, A, None, None, 0.3, -1000.0, 100.0, -1.5, crit, seed=7, env_id_offset=12345678901,
                         start_local=start, goal_local=goal, kernel=kernel)
        co = c_oracle.COracle(nbr, A, start, goal, 0.3, -1000.0, 100.0, -1.5, ocrit, seed=7, env_id_offset=12345678901)
        for t in range(T):
            acts = rs.randint(0, 5, size=(E, A)).astype(np.uint8)
            local, reward, done, info = env.step(acts, auto_reset=auto)
            ref = co.step(acts, auto_reset=auto)
            assert np.array_equal(local, ref['local']), (A, t)
            assert np.array_equal(_bits(reward), _bits(ref['reward'])) and np.array_equal(_bits(info['prob']), _bits(ref['prob']))
            assert np.array_equal(done, ref['done']) and np.array_equal(info['collision'], ref['collision'])
            assert np.array_equal(info['was_terminal'], ref['was_terminal'])
        res = env.rollout(25, auto_reset=auto)
        ref = co.rollout(25, auto_reset=auto)
        assert np.array_equal(_bits(res['returns']), _bits(ref['returns']))
        assert np.array_equal(res['episodes'], ref['episodes']) and np.array_equal(res['collisions'], ref['collisions'])
        assert np.array_equal(env.get_state()[0], co.state)
        env.close()


def test_fill_random_actions_matches_policy_stream():
    grid = MapfGrid(['.....'] * 5)
    for A in (1, 3, 4, 8, 13):
        env = VecMapfEnv(grid, A, None, None, 0.0, -1.0, 1.0, -1.0, OptimizationCriteria.Makespan, seed=99,
                         env_id_offset=(1 << 33) + 7, start_local=np.zeros((70, A), np.uint16) + np.arange(A, dtype=np.uint16),
                         goal_local=np.zeros((70, A), np.uint16) + np.arange(A, dtype=np.uint16)[::-1])
        got = env.fill_random_actions(5, 6)
        ids = (1 << 33) + 7 + np.arange(70)
        exp = np.stack([philox.random_actions_np(99, ids, 5 + s, A) for s in range(6)])
        assert np.array_equal(got, exp)
        env.close()


def test_results_do_not_depend_on_how_envs_are_sharded():
    """One handle with 600 envs == three handles owning [0,200), [200,400), [400,600) with global env ids
    (what bench.py's ranks do): same trajectories, same returns, stepwise and fused."""
    rs = np.random.RandomState(3)
    lines = [''.join('@' if rs.rand() < 0.15 else '.' for _ in range(20)) for _ in range(20)]
    grid = MapfGrid(lines)
    V, E, A, T = len(grid.tables()[0]), 600, 6, 50
    start = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    goal = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    mk = lambda lo, hi: VecMapfEnv(grid, A, None, None, 0.25, -50.0, 10.0, -1.0, OptimizationCriteria.SoC, seed=5,  # noqa: E731
                                   env_id_offset=1000 + lo, start_local=start[lo:hi], goal_local=goal[lo:hi])
    whole, parts = mk(0, E), [mk(lo, lo + 200) for lo in (0, 200, 400)]
    acts = rs.randint(0, 5, size=(T, E, A)).astype(np.uint8)
    for t in range(T):
        l, r, d, info = whole.step(acts[t], auto_reset=True)
        for k, p in enumerate(parts):
            sl = slice(200 * k, 200 * k + 200)
            lp, rp, dp, ip = p.step(acts[t][sl], auto_reset=True)
            assert np.array_equal(lp, l[sl]) and np.array_equal(_bits(rp), _bits(r[sl])) and np.array_equal(dp, d[sl])
            assert np.array_equal(_bits(ip['prob']), _bits(info['prob'][sl]))
    full = whole.rollout(40)
    for k, p in enumerate(parts):
        sl = slice(200 * k, 200 * k + 200)
        assert np.array_equal(_bits(p.rollout(40)['returns']), _bits(full['returns'][sl]))
    whole.close()
    [p.close() for p in parts]


@pytest.mark.parametrize('n_agents,n_envs', [
    (2, 16384), (2, 2048),                       # pair layout L = 1: LDS table / global table
    (4, 16384), (4, 16512), (4, 1024),           # quad layout Q = 1; pair layout L = 2 (LDS: 16512 = 129 pair blocks); global
    (8, 8192), (8, 16448), (8, 1024),            # quad Q = 2; pair L = 4 (LDS: 257 pair blocks); global
    (16, 4096), (16, 4128), (16, 512),           # quad Q = 4; pair L = 8 (LDS); global
    (32, 2048), (32, 1024),                      # quad Q = 8; pair L = 16 (LDS)
    (64, 1024), (64, 512),                       # quad Q = 16; pair L = 32 (LDS)
    (128, 256)])                                 # pair L = 64
def test_recorded_rollout_of_full_groups_against_c_oracle(n_agents, n_envs):
    """Full groups whose env count fills every block run the predicate-free rollout kernels -- the packed
    layout (four agents per lane) where A = 4Q and the batch fills its blocks, else the pair layout (LDS move
    table for the larger batches, global table for the smaller): every recorded step -- cells, reward, prob,
    done, collision -- against the C oracle stepped with the same actions, streamed and policy-generated,
    both criteria, with envs that stay terminal (no auto-reset) in the second pass."""
    rs = np.random.RandomState(500 + n_agents)
    H = W = 20
    lines = [''.join('@' if rs.rand() < 0.15 else '.' for _ in range(W)) for _ in range(H)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    V, E, A, T = len(valid), n_envs, n_agents, 14
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal[::7] = start[::7]                                   # some envs start in a terminal state
    ids = 77 + np.arange(E)
    # third and fourth pass: no slip at all (no uniform is consumed) and fail_prob = 1 (the intended move has
    # probability 0 and is dropped from the list), with one start / goal row broadcast to every env
    for crit, ocrit, auto, streamed, fail_prob, bcast in (
            (OptimizationCriteria.Makespan, mo.MAKESPAN, True, True, 0.2, False),
            (OptimizationCriteria.SoC, mo.SOC, False, False, 0.2, False),
            (OptimizationCriteria.Makespan, mo.MAKESPAN, True, False, 0.0, True),
            (OptimizationCriteria.SoC, mo.SOC, True, True, 1.0, True)):
        st, gl = (start[0], goal[1]) if bcast else (start, goal)
        env = VecMapfEnv(grid, A, None, None, fail_prob, -1000.0, 100.0, -1.0, crit, seed=11, env_id_offset=77,
                         start_local=st, goal_local=gl, n_envs=E)
        co = c_oracle.COracle(nbr, A, np.broadcast_to(st, (E, A)).copy(), np.broadcast_to(gl, (E, A)).copy(), fail_prob,
                              -1000.0, 100.0, -1.0, ocrit, seed=11, env_id_offset=77)
        acts = np.stack([philox.random_actions_np(11, ids, t, A) for t in range(T)])
        res = env.rollout(T, actions=acts if streamed else None, auto_reset=auto, record=True)
        ret = np.zeros(E)
        for t in range(T):
            ref = co.step(acts[t], auto_reset=auto)
            assert np.array_equal(res['local'][t], ref['local']), (A, t)
            assert np.array_equal(_bits(res['reward'][t]), _bits(ref['reward'])), (A, t)
            assert np.array_equal(_bits(res['prob'][t]), _bits(ref['prob'])), (A, t)
            assert np.array_equal(res['done'][t], ref['done']) and np.array_equal(res['collision'][t], ref['collision'])
            ret = ret + ref['reward']
        assert np.array_equal(_bits(res['returns']), _bits(ret))
        assert np.array_equal(env.get_state()[0], co.state)
        env.close()


@pytest.mark.parametrize('n_agents,n_envs', [(8, 8192), (16, 4096), (32, 2048)])
def test_recorded_rollout_with_eight_agents_per_lane(n_agents, n_envs, monkeypatch):
    """The same passes with the packed layout pinned to eight agents per lane (its default range starts at two waves
    per SIMD, i.e. 131072 envs of 8 agents), and a split rollout starting at a step index that is not a multiple of 4."""
    set_tune(monkeypatch, k='8')
    test_recorded_rollout_of_full_groups_against_c_oracle(n_agents, n_envs)
    if n_agents <= 16:
        test_dense_rollout_split_launches_accumulate_and_single_steps(n_agents, n_envs)
    grid = MapfGrid(['....', '....', '....', '....', '....', '....', '....', '....'])
    env = VecMapfEnv(grid, n_agents, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=1, n_envs=n_envs,
                     start_local=np.arange(n_agents, dtype=np.uint16), goal_local=np.arange(n_agents, dtype=np.uint16)[::-1].copy())
    env.rollout(5, auto_reset=True)
    assert 'K=8' in env.last_kernel('rollout'), env.last_kernel('rollout')
    env.close()


@pytest.mark.parametrize('n_agents,n_envs', [(4, 16384), (8, 8192), (8, 16448), (16, 4096)])
def test_dense_rollout_split_launches_accumulate_and_single_steps(n_agents, n_envs):
    """Quad-lane and pair layouts: a rollout split into launches of 1, 4 and 7 steps that accumulate into the same
    totals equals one 12-step launch (returns bit for bit, episode and collision counts, final state), starting at a
    step index that is not a multiple of 4 (the slip stream's call granularity)."""
    rs = np.random.RandomState(900 + n_agents)
    grid = MapfGrid([''.join('@' if rs.rand() < 0.15 else '.' for _ in range(18)) for _ in range(18)])
    valid, _, nbr = grid.tables()
    V, E, A = len(valid), n_envs, n_agents
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    mk = lambda: VecMapfEnv(grid, A, None, None, 0.25, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=3,  # noqa: E731
                            start_local=start, goal_local=goal)
    one, parts = mk(), mk()
    for env in (one, parts):
        env.set_state(None, t=3)
    full = one.rollout(12, auto_reset=True)
    acc = parts.rollout(1, auto_reset=True)
    acc = parts.rollout(4, auto_reset=True, accumulate_into=acc)
    acc = parts.rollout(7, auto_reset=True, accumulate_into=acc)
    assert np.array_equal(_bits(full['returns']), _bits(acc['returns']))
    assert np.array_equal(full['episodes'], acc['episodes']) and np.array_equal(full['collisions'], acc['collisions'])
    (s1, t1), (s2, t2) = one.get_state(), parts.get_state()
    assert t1 == t2 == 15 and np.array_equal(s1, s2)
    co = c_oracle.COracle(nbr, A, start, goal, 0.25, -1000.0, 100.0, -1.0, mo.MAKESPAN, seed=3)
    co.t = 3
    ref = co.rollout(12, auto_reset=True)
    assert np.array_equal(_bits(full['returns']), _bits(ref['returns'])) and np.array_equal(s1, co.state)
    one.close(), parts.close()


@pytest.mark.parametrize('n_agents,n_envs,k,streamed', [(8, 8192, None, True), (16, 4096, None, True), (32, 2048, None, True), (8, 8192, '8', True),
                                                        (16, 4096, '8', True), (4, 16384, '2', True),
                                                        (8, 8192, None, False), (32, 2048, None, False), (16, 4096, '8', False)])
def test_recorded_launches_of_every_length_and_phase(n_agents, n_envs, k, streamed, monkeypatch):
    """The packed kernels' step loop has different code for a launch's head (the steps up to the slip stream's next call
    boundary, with action registers of their own), its groups of four steps and its last one to three steps: a chain of
    recorded launches whose lengths and first step indices cover every (t mod 4, length) combination up to length 9 -- and a
    few longer ones -- every recorded step against the C oracle stepped with the same actions; streamed actions and the
    in-kernel policy (whose loop differs again: single steps to the boundary, groups of four to the end)."""
    if k is not None:
        set_tune(monkeypatch, k=k)
    rs = np.random.RandomState(4100 + n_agents)
    grid = MapfGrid([''.join('@' if rs.rand() < 0.15 else '.' for _ in range(18)) for _ in range(18)])
    valid, _, nbr = grid.tables()
    V, E, A = len(valid), n_envs, n_agents
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=21, start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, -1000.0, 100.0, -1.0, mo.MAKESPAN, seed=21)
    # lengths 1..9 from each phase: after a launch of n steps from phase f the next one starts at phase (f + n) & 3
    lengths, seen, t = [], set(), 0
    while len(seen) < 36:
        n = next((n for n in range(1, 10) if (t & 3, n) not in seen), None)
        if n is None:                                             # every length seen from this phase: move on by one step
            n = 1
        seen.add((t & 3, n)); lengths.append(n); t += n
    lengths += [13, 22, 17]
    ids = np.arange(E)
    t = 0
    for n in lengths:
        acts = np.stack([philox.random_actions_np(77 if streamed else 21, ids, t + j, A) for j in range(n)])
        res = env.rollout(n, actions=acts if streamed else None, auto_reset=True, record=True)
        if k is not None:
            assert 'K=%s' % k in env.last_kernel('rollout'), env.last_kernel('rollout')
        assert 'lq_rollout_kernel' in env.last_kernel('rollout') and ('STREAM' if streamed else 'POLICY') in env.last_kernel('rollout')
        for j in range(n):
            ref = co.step(acts[j], auto_reset=True)
            assert np.array_equal(res['local'][j], ref['local']), (n, t, j)
            assert np.array_equal(_bits(res['reward'][j]), _bits(ref['reward'])) and np.array_equal(_bits(res['prob'][j]), _bits(ref['prob'])), (n, t, j)
            assert np.array_equal(res['done'][j], ref['done']) and np.array_equal(res['collision'][j], ref['collision']), (n, t, j)
        t += n
    assert np.array_equal(env.get_state()[0], co.state) and env.get_state()[1] == t
    env.close()


@pytest.mark.parametrize('n_agents,n_envs,kernel', [(2, 300, 'thread_per_env'), (5, 300, 'thread_per_env'), (3, 257, 'lane_group'),
                                                    (8, 8192, 'auto'), (8, 16448, 'auto'), (16, 4096, 'auto'), (7, 1000, 'auto')])
def test_greedy_policy_rollout_against_c_oracle(n_agents, n_envs, kernel):
    """mapf_set_policy(MAPF_POLICY_GREEDY): the fused rollout with the on-device greedy policy (thread-per-env,
    lane-group and packed-layout kernels) records exactly the trajectory the C oracle produces when it is stepped with the
    greedy actions of its own restatement; switching back to the random policy restores the policy stream."""
    rs = np.random.RandomState(700 + n_agents)
    lines = [''.join('@' if rs.rand() < 0.15 else '.' for _ in range(20)) for _ in range(20)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    rc = np.asarray([r | (c << 16) for r, c in valid], np.uint32)
    V, E, A, T = len(valid), n_envs, n_agents, 20
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=21,
                     start_local=start, goal_local=goal, kernel=kernel)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, -1000.0, 100.0, -1.0, mo.MAKESPAN, seed=21)
    env.set_policy('greedy')
    res = env.rollout(T, auto_reset=True, record=True)
    goals_reached = 0
    for t in range(T):
        ref = co.step(co.greedy_actions(rc), auto_reset=True)
        assert np.array_equal(res['local'][t], ref['local']), (A, t)
        assert np.array_equal(_bits(res['reward'][t]), _bits(ref['reward'])) and np.array_equal(_bits(res['prob'][t]), _bits(ref['prob']))
        assert np.array_equal(res['done'][t], ref['done']) and np.array_equal(res['collision'][t], ref['collision'])
        goals_reached += int((ref['done'] & ~ref['collision']).sum())
    assert np.array_equal(env.get_state()[0], co.state)
    assert A > 3 or goals_reached > 0                        # small teams do arrive within 20 greedy steps
    env.set_policy('random')
    ref = co.rollout(6, auto_reset=True)
    out = env.rollout(6, auto_reset=True)
    assert np.array_equal(_bits(out['returns']), _bits(ref['returns'])) and np.array_equal(env.get_state()[0], co.state)
    env.close()


# ----------------------------------------------------------------------- episodes that end on goals
def _goal_scenario_tables(n_agents, n_envs, seed):
    import goal_scenarios
    lines, start_loc, goal_loc = goal_scenarios.goal_scenario(n_agents, n_envs, seed)
    grid = MapfGrid(lines)
    valid, l2i, nbr = grid.tables()
    ids = np.zeros((len(lines), len(lines[0])), np.uint16)
    for loc, k in l2i.items():
        ids[loc] = k
    start = np.ascontiguousarray(ids[start_loc[..., 0], start_loc[..., 1]])
    goal = np.ascontiguousarray(ids[goal_loc[..., 0], goal_loc[..., 1]])
    rc = np.asarray([r | (c << 16) for r, c in valid], np.uint32)
    return grid, nbr, rc, start, goal


@pytest.mark.parametrize('n_agents,n_envs,layout,env_vars', [
    (4, 16384, 'lq_rollout_kernel<Q=1,K=4', {'k': '4'}), (4, 16512, 'lq_rollout_kernel<Q=2,K=2', {'k': '2'}),
    (8, 8192, 'lq_rollout_kernel<Q=2,K=4', {'k': '4'}), (8, 16448, 'lq_rollout_kernel<Q=4,K=2', {'k': '2'}),
    (8, 16448, 'lg_rollout_kernel<L=4,FULL,MV_LDS', {'quad_lanes': '0'}),
    (16, 4096, 'lq_rollout_kernel<Q=4,K=4', {'k': '4'}), (16, 4128, 'lq_rollout_kernel<Q=8,K=2', {'k': '2'}),
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,NO_TERMINAL> block=', {'k': '4', 'bitmap_pairs': '0'}),
    # (32 agents, full table rows in LDS: occupancy bitmaps behind them -- what MAPF_TUNE k=4 or a batch of one wave per SIMD gets)
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,NO_TERMINAL,BITMAP> block=', {'k': '4'}),
    (32, 1024, 'lq_rollout_kernel<Q=16,K=2', {'k': '2'}),
    (32, 1024, 'lg_rollout_kernel<L=16,FULL,MV_GLOBAL', {'mv_lds_max_bytes': '0'}),
    # eight agents per lane (the default only for batches of two waves per SIMD and more)
    (8, 8192, 'lq_rollout_kernel<Q=1,K=8', {'k': '8'}), (16, 4096, 'lq_rollout_kernel<Q=2,K=8', {'k': '8'}),
    (32, 2048, 'lq_rollout_kernel<Q=4,K=8', {'k': '8'}),
    (32, 2048, 'lq_rollout_kernel<Q=4,K=8,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL> block=512', {'mv_lds_max_bytes': '2048', 'k': '8'}),
    # (a full table "too large" for the LDS budget: the 8-byte-row form of the packed kernel, 512- and 1024-thread blocks)
    (16, 4096, 'lq_rollout_kernel<Q=4,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL> block=512', {'mv_lds_max_bytes': '2048'}),
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL> block=512', {'mv_lds_max_bytes': '2048', 'bitmap_pairs': '0'}),
    # (32 agents, 8-byte rows: collisions through per-env LDS occupancy bitmaps instead of the 496 agent pairs -- the default)
    # (five-column table where it leaves room for the bitmaps -- here it does -- else four columns and a made-up STAY row)
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAP5> block=512', {'mv_lds_max_bytes': '2048', 'bitmap_delta': '0'}),
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAP> block=512', {'mv_lds_max_bytes': '2048', 'bitmap_staycol': '0', 'bitmap_delta': '0'}),
    # (... in 1024-thread blocks, 128 bitmaps behind the table: what a batch that fills every CU with such a block gets)
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAP> block=1024', {'mv_lds_max_bytes': '2048', 'bitmap_block': '1024', 'bitmap_staycol': '0', 'bitmap_delta': '0'}),
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAP5> block=1024', {'mv_lds_max_bytes': '2048', 'bitmap_block': '1024', 'bitmap_delta': '0'}),
    # (... behind 4-byte delta rows -- the default wherever a map's neighbour ids lie within +-127 of their cell's)
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAPD> block=512', {'mv_lds_max_bytes': '2048'}),
    (32, 2048, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAPD> block=1024', {'mv_lds_max_bytes': '2048', 'bitmap_block': '1024'}),
    (64, 16384, 'lq_rollout_kernel<Q=16,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL> block=1024', {'mv_lds_max_bytes': '2048'}),
    (8, 300, 'lg_rollout_kernel<L=4,FULL,MV_GLOBAL', {}), (5, 600, 'lg_rollout_kernel<L=4,RAGGED', {}),
    (8, 16417, 'lg_rollout_kernel<L=4,FULL,MV_LDS', {}), (6, 512, 'rollout_kernel<A=6>', {})])
def test_goal_reaching_episodes_against_c_oracle(n_agents, n_envs, layout, env_vars, monkeypatch):
    """The goal-reached branch (reward_of_goal + living, done, no collision, then terminal / auto-reset) and "vertex
    collision while every agent sits on its goal" (collision wins) at 4..32 agents, in every rollout kernel -- the
    packed layout with four and with two agents per lane, the lane-group layout (LDS and global table, dense and
    guarded), the thread-per-env kernel -- and in the single-step kernel: agents start one move from their goals on an
    open map and are driven towards them (oracle/goal_scenarios.py -- the family the reference itself stepped for
    tests/golden/goals_*).  Every recorded step against the C oracle; each pass must actually contain both outcomes,
    and the library must report the kernel this case is meant to reach."""
    set_tune(monkeypatch, **env_vars)
    A, E = n_agents, n_envs
    grid, nbr, rc, start, goal = _goal_scenario_tables(A, E, 8100 + A)
    for fail_prob, crit, ocrit, auto, mode, T in (
            (0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, True, 'streamed', 10),
            (0.0, OptimizationCriteria.SoC, mo.SOC, False, 'policy', 4),
            (0.2, OptimizationCriteria.SoC, mo.SOC, True, 'single', 8),
            (0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, False, 'policy', 10),
            (0.0, OptimizationCriteria.Makespan, mo.MAKESPAN, True, 'streamed', 3)):
        env = VecMapfEnv(grid, A, None, None, fail_prob, -1000.0, 100.0, -1.0, crit, seed=31, env_id_offset=5,
                         start_local=start, goal_local=goal,
                         kernel='thread_per_env' if layout.startswith('rollout_kernel') else 'auto')
        co = c_oracle.COracle(nbr, A, start, goal, fail_prob, -1000.0, 100.0, -1.0, ocrit, seed=31, env_id_offset=5)
        acts, refs = [], []
        for t in range(T):
            acts.append(co.greedy_actions(rc))
            refs.append(co.step(acts[-1], auto_reset=auto))
        if mode == 'single':
            got = []
            for t in range(T):
                local, reward, done, info = env.step(acts[t], auto_reset=auto)
                got.append((local, reward, info['prob'], done, info['collision']))
            assert 'step_kernel' in env.last_kernel('step')
        else:
            if mode == 'policy':
                env.set_policy('greedy')
            res = env.rollout(T, actions=np.stack(acts) if mode == 'streamed' else None, auto_reset=auto, record=True)
            got = [(res['local'][t], res['reward'][t], res['prob'][t], res['done'][t], res['collision'][t]) for t in range(T)]
            seen = env.last_kernel('rollout')
            wanted = layout if (mode == 'streamed' and crit == OptimizationCriteria.Makespan) else layout.split(',RECORD')[0]
            assert wanted in seen, seen
            assert ('BITMAP' in seen) == ('BITMAP' in layout), seen
        goals = clash_on_goal = 0
        for t in range(T):
            ref, (local, reward, prob, done, coll) = refs[t], got[t]
            tag = (A, E, mode, fail_prob, t)
            assert np.array_equal(local, ref['local']), tag
            assert np.array_equal(_bits(reward), _bits(ref['reward'])) and np.array_equal(_bits(prob), _bits(ref['prob'])), tag
            assert np.array_equal(done, ref['done']) and np.array_equal(coll, ref['collision']), tag
            fresh = ref['was_terminal'] == 0
            on_goal = (ref['local'] == goal).all(axis=1)
            goals += int((fresh & (ref['done'] == 1) & (ref['collision'] == 0)).sum())
            clash_on_goal += int((fresh & (ref['collision'] == 1) & on_goal).sum())
        assert np.array_equal(env.get_state()[0], co.state)
        assert goals > 0 and clash_on_goal > 0, (A, E, mode, fail_prob, goals, clash_on_goal)
        env.close()


@pytest.mark.parametrize('env_vars,want', [
    ({'mv_lds_max_bytes': '2048'}, 'COMPACT,NO_TERMINAL,BITMAPD> block=512'),
    ({'mv_lds_max_bytes': '2048', 'bitmap_delta': '0'}, 'COMPACT,NO_TERMINAL,BITMAP5> block=512'),
    ({'mv_lds_max_bytes': '2048', 'bitmap_block': '1024', 'bitmap_staycol': '0', 'bitmap_delta': '0'}, 'COMPACT,NO_TERMINAL,BITMAP> block=1024'),
    ({'k': '4'}, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,NO_TERMINAL,BITMAP> block='),
    ({'k': '4', 'bitmap_pairs': '0'}, 'lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,NO_TERMINAL> block=')])
def test_systolic_probability_chain_over_short_and_split_launches(env_vars, want, monkeypatch):
    """32 agents in eight lanes: the ordered probability product is a systolic chain -- one hand-over per step, the last lane
    completing a step's product seven steps later and every launch ending with seven draining rounds.  Launches SHORTER than
    the chain (1, 2, 7 steps), of its length, and longer ones, issued back to back: every recorded probability (and
    everything else) against the C oracle, slip 0.2 and 0 (all factors 1.0), episodes ending and restarting in between."""
    set_tune(monkeypatch, **env_vars)
    A, E = 32, 2048
    grid, nbr, rc, start, goal = _goal_scenario_tables(A, E, 8100 + A)
    for fail_prob in (0.2, 0.0):
        env = VecMapfEnv(grid, A, None, None, fail_prob, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=77, env_id_offset=9,
                         start_local=start, goal_local=goal)
        co = c_oracle.COracle(nbr, A, start, goal, fail_prob, -1000.0, 100.0, -1.0, mo.MAKESPAN, seed=77, env_id_offset=9)
        rs = np.random.RandomState(5)
        for T in (1, 2, 7, 8, 9, 1, 13, 3, 16):
            acts = rs.randint(0, 5, size=(T, E, A)).astype(np.uint8)
            res = env.rollout(T, actions=acts, auto_reset=True, record=True)
            assert want in env.last_kernel('rollout'), env.last_kernel('rollout')
            for t in range(T):
                ref = co.step(acts[t], auto_reset=True)
                assert np.array_equal(_bits(res['prob'][t]), _bits(ref['prob'])), (fail_prob, T, t)
                assert np.array_equal(res['local'][t], ref['local']) and np.array_equal(_bits(res['reward'][t]), _bits(ref['reward']))
                assert np.array_equal(res['done'][t], ref['done']) and np.array_equal(res['collision'][t], ref['collision'])
        assert np.array_equal(env.get_state()[0], co.state)
        env.close()


# ----------------------------------------------------------------------- BASELINE.json full sizes
def _full_size_check(grid, nbr, A, start, goal, fail_prob, crit, ocrit, n_step, n_roll, kernel='auto', env_id_offset=0,
                     want_step=None, want_rollout=None, n_streamed=0):
    """Every env, every step against the C oracle: n_step single steps, an n_roll-step recorded rollout driven by the in-kernel
    policy stream and (n_streamed > 0) a recorded rollout with streamed actions.  want_step / want_rollout: substrings the
    dispatched kernel names must contain (the test is ABOUT that kernel instance)."""
    E = start.shape[0]
    # the oracle steps on ITS OWN neighbour table, built from the map's text (mapf_env.py:43-94 restated in oracle/mapf_oracle.py):
    # on maps without a golden an error of the product's table builder would otherwise be shared by both sides
    lines = [''.join('@' if f else '.' for f in row) for row in grid.obstacles.tolist()]
    own_nbr = np.asarray(mo.neighbour_table(lines), np.uint16)
    assert np.array_equal(nbr, own_nbr)
    env = VecMapfEnv(grid, A, None, None, fail_prob, -1000.0, 100.0, -1.0, crit, seed=42, start_local=start,
                     goal_local=goal, kernel=kernel, env_id_offset=env_id_offset)
    co = c_oracle.COracle(own_nbr, A, start, goal, fail_prob, -1000.0, 100.0, -1.0, ocrit, seed=42, env_id_offset=env_id_offset)
    ids = env_id_offset + np.arange(E)
    for t in range(n_step):
        acts = philox.random_actions_np(42, ids, t, A)
        local, reward, done, info = env.step(acts, auto_reset=True)
        ref = co.step(acts, auto_reset=True)
        assert np.array_equal(local, ref['local']), t
        assert np.array_equal(_bits(reward), _bits(ref['reward'])) and np.array_equal(_bits(info['prob']), _bits(ref['prob']))
        assert np.array_equal(done, ref['done']) and np.array_equal(info['collision'], ref['collision'])
    if want_step is not None:
        assert want_step in env.last_kernel('step'), env.last_kernel('step')
    if n_streamed:
        t0 = n_step
        acts = np.stack([philox.random_actions_np(43, ids, t0 + k, A) for k in range(n_streamed)])   # (not the policy stream's key)
        res = env.rollout(n_streamed, actions=acts, auto_reset=True, record=True)
        if want_rollout is not None:
            assert want_rollout in env.last_kernel('rollout'), env.last_kernel('rollout')
        for k in range(n_streamed):
            ref = co.step(acts[k], auto_reset=True)
            assert np.array_equal(res['local'][k], ref['local']), k
            assert np.array_equal(_bits(res['reward'][k]), _bits(ref['reward'])) and np.array_equal(_bits(res['prob'][k]), _bits(ref['prob'])), k
            assert np.array_equal(res['done'][k], ref['done']) and np.array_equal(res['collision'][k], ref['collision']), k
        assert np.array_equal(env.get_state()[0], co.state)
    t_pol = env.t
    res = env.rollout(n_roll, auto_reset=True, record=True)          # in-kernel policy stream, recorded ...
    returns, episodes, collisions = np.zeros(E), np.zeros(E, np.uint32), np.zeros(E, np.uint32)
    for k in range(n_roll):                                           # ... and compared step by step, every env
        acts = philox.random_actions_np(42, ids, t_pol + k, A)        # (what the kernel must have drawn: key seed + 1)
        ref = co.step(acts, auto_reset=True)
        assert np.array_equal(res['local'][k], ref['local']), k
        assert np.array_equal(_bits(res['reward'][k]), _bits(ref['reward'])) and np.array_equal(_bits(res['prob'][k]), _bits(ref['prob'])), k
        assert np.array_equal(res['done'][k], ref['done']) and np.array_equal(res['collision'][k], ref['collision']), k
        returns = returns + ref['reward']
        episodes += ref['done']
        collisions += ref['collision']
    assert np.array_equal(_bits(res['returns']), _bits(returns))
    assert np.array_equal(res['episodes'], episodes) and np.array_equal(res['collisions'], collisions)
    assert np.array_equal(env.get_state()[0], co.state)
    # size-independent properties of the recorded trajectory
    assert np.array_equal(res['done'].sum(0), res['episodes']) and np.array_equal(res['collision'].sum(0), res['collisions'])
    ret = np.zeros(E)
    for t in range(n_roll):
        ret = ret + res['reward'][t]
    assert np.array_equal(_bits(ret), _bits(res['returns']))          # returns == ordered sum of recorded rewards
    assert np.all((res['prob'] > 0) & (res['prob'] <= 1.0)) and np.all(res['local'] < nbr.shape[0])
    assert np.all(res['collision'] <= res['done'])                    # a collision always ends the episode
    env.close()
    return int(res['episodes'].sum())


def _scen_tables(map_name, scen_ids, A, E):
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    grid = MapfGrid(parse_map_file(map_name_to_files(map_name, scen_ids[0])[0]))
    _, l2i, nbr = grid.tables()
    per = [parse_scen_file(map_name_to_files(map_name, sid)[1], A) for sid in scen_ids]
    which = np.arange(E) % len(scen_ids)
    start = np.asarray([[l2i[l] for l in p[0]] for p in per], np.uint16)[which]
    goal = np.asarray([[l2i[l] for l in p[1]] for p in per], np.uint16)[which]
    return grid, nbr, np.ascontiguousarray(start), np.ascontiguousarray(goal)


@pytest.mark.parametrize('criteria', ['Makespan', 'SoC'])
def test_large_map_maze128_32agents_16384_envs(monkeypatch, criteria):
    """The reference's LARGE maps at batch (its own grid test opens them: mapf_grid_tests.py:22-32; mapf_env.py:142-143
    builds valid_locations for any size).  maze-128-128-10, scen 18 (the only scenario that constructs with 32 agents),
    14818 free cells: a 1.2 MB move table, i.e. the kernels that gather rows from GLOBAL memory -- the lane-group rollout
    <L=16,FULL,MV_GLOBAL,...,DENSE> and the packed single step <Q=8,K=4,SCEN> -- under default dispatch, every env of
    every step against the C oracle (round 3 only TIMED these instances)."""
    set_tune(monkeypatch, quad_min_lanes=None)
    grid, nbr, start, goal = _scen_tables('maze-128-128-10', [18], 32, 16384)
    assert nbr.shape[0] == 14818
    crit, ocrit = (OptimizationCriteria.SoC, mo.SOC) if criteria == 'SoC' else (OptimizationCriteria.Makespan, mo.MAKESPAN)
    n = _full_size_check(grid, nbr, 32, start, goal, 0.2, crit, ocrit, 6, 24, want_step='lq_step_kernel<Q=8,K=4,SCEN',
                         want_rollout='lg_rollout_kernel<L=16,FULL,MV_GLOBAL,RECORD,STREAM,DENSE>', n_streamed=16)
    assert n > 100                                                # (32 agents leaving one scenario's start cells do collide)


@pytest.mark.parametrize('criteria,delta_rows', [('Makespan', True), ('SoC', True), ('Makespan', False), ('SoC', False),
                                                 ('Makespan', 'step'), ('SoC', 'step'), ('Makespan', 'step_pairs')])
def test_reference_room64_maps_32agents_under_default_dispatch(monkeypatch, criteria, delta_rows):
    """32 agents on the reference's own 64x64 room maps (room-64-64-16: 3646 free cells, room-64-64-8: 3232; scenarios that
    construct with 32 agents), 8192 envs over several scenarios each: the occupancy-bitmap / systolic-chain form of the packed
    rollout under default dispatch -- the four-column table where five columns would not leave room for the bitmaps, the
    five-column one where they do (MAPF_TUNE bitmap_delta=0), or 4-byte delta rows (the default on these maps) -- and the packed
    single step, every env of every step against the C oracle."""
    set_tune(monkeypatch, quad_min_lanes=None)
    if not delta_rows:
        set_tune(monkeypatch, bitmap_delta='0')
    # 'step': the single step's LDS table of delta rows too, with the occupancy bitmaps behind it (by default only from a batch
    # that fills the device on; these 8192 envs take the scenario-table instances of it, which the per-env rows of configs[4]
    # never reach); 'step_pairs': the same table with all agent pairs (MAPF_TUNE bitmap_pairs=0)
    want_step = 'lq_step_kernel<Q=8,K=4'
    if delta_rows in ('step', 'step_pairs'):
        set_tune(monkeypatch, step_delta='2')
        want_step = 'lq_step_kernel<Q=8,K=4,SCEN,NO_TERMINAL,DELTA,BITMAP> block=512'
        if delta_rows == 'step_pairs':
            set_tune(monkeypatch, bitmap_pairs='0')
            want_step = 'lq_step_kernel<Q=8,K=4,SCEN,NO_TERMINAL,DELTA> block=512'
    crit, ocrit = (OptimizationCriteria.SoC, mo.SOC) if criteria == 'SoC' else (OptimizationCriteria.Makespan, mo.MAKESPAN)
    for map_name, want in (('room-64-64-16', ',BITMAP> block=512'), ('room-64-64-8', ',BITMAP5> block=512')):
        want = ',BITMAPD> block=512' if delta_rows else want
        scen_ids = []
        for sid in range(1, 26):                                   # (a scenario whose first 32 agents do not construct is skipped)
            try:
                _scen_tables(map_name, [sid], 32, 1)
                scen_ids.append(sid)
            except Exception:
                pass
            if len(scen_ids) == 4:
                break
        assert len(scen_ids) >= 2, (map_name, scen_ids)
        grid, nbr, start, goal = _scen_tables(map_name, scen_ids, 32, 8192)
        n = _full_size_check(grid, nbr, 32, start, goal, 0.2, crit, ocrit, 3, 16, want_step=want_step,
                             want_rollout='lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,%s,COMPACT' % ('SOC' if criteria == 'SoC' else 'MAKESPAN'),
                             n_streamed=16)
        assert n > 50, (map_name, n)
        env = VecMapfEnv(grid, 32, None, None, 0.2, -1000.0, 100.0, -1.0, crit, seed=42, start_local=start, goal_local=goal)
        if delta_rows not in ('step', 'step_pairs'):
            env.rollout(4, auto_reset=True, record=True)
            assert want in env.last_kernel('rollout'), (map_name, env.last_kernel('rollout'))
        else:   # ... and the instance WITH is_terminal(prev): steps without auto-reset (finished episodes stay terminal)
            co = c_oracle.COracle(nbr, 32, start, goal, 0.2, -1000.0, 100.0, -1.0, ocrit, seed=42)
            for t in range(6):
                acts = philox.random_actions_np(42, np.arange(8192), t, 32)
                local, reward, done, info = env.step(acts, auto_reset=False)
                ref = co.step(acts, auto_reset=False)
                assert np.array_equal(local, ref['local']) and np.array_equal(done, ref['done']), t
                assert np.array_equal(_bits(reward), _bits(ref['reward'])) and np.array_equal(_bits(info['prob']), _bits(ref['prob'])), t
                assert np.array_equal(info['collision'], ref['collision']) and np.array_equal(info['was_terminal'], ref['was_terminal']), t
            seen = env.last_kernel('step')
            assert seen.startswith(want_step.replace(',NO_TERMINAL', '').split('>')[0]) and 'NO_TERMINAL' not in seen, seen
            assert int(ref['was_terminal'].sum()) > 0                # terminal envs were stepped
        env.close()


@pytest.mark.parametrize('criteria', ['Makespan', 'SoC'])
def test_large_map_berlin256_4agents_65536_envs(monkeypatch, criteria):
    """Berlin_1_256 (47540 free cells, a 3.8 MB move table; the map mapf_grid_tests.py:22-32 opens), scen 11, 4 agents,
    65536 envs: packed single step <Q=1,K=4,SCEN> at V = 47540 and the lane-group rollout <L=2,FULL,MV_GLOBAL,...,DENSE>."""
    set_tune(monkeypatch, quad_min_lanes=None)
    grid, nbr, start, goal = _scen_tables('Berlin_1_256', [11], 4, 65536)
    assert nbr.shape[0] == 47540
    crit, ocrit = (OptimizationCriteria.SoC, mo.SOC) if criteria == 'SoC' else (OptimizationCriteria.Makespan, mo.MAKESPAN)
    _full_size_check(grid, nbr, 4, start, goal, 0.2, crit, ocrit, 8, 32, want_step='lq_step_kernel<Q=1,K=4,SCEN',
                     want_rollout='lg_rollout_kernel<L=2,FULL,MV_GLOBAL,RECORD,STREAM,DENSE>', n_streamed=24)


def test_large_map_berlin256_2agents_eight_scenarios_65536_envs():
    """Berlin_1_256 with 2 agents over eight scenario ids (the thread-per-env family is the default at A <= 2): step and
    rollout at V = 47540, slip 0.1."""
    grid, nbr, start, goal = _scen_tables('Berlin_1_256', [2, 4, 8, 11, 14, 18, 22, 24], 2, 65536)
    _full_size_check(grid, nbr, 2, start, goal, 0.1, OptimizationCriteria.Makespan, mo.MAKESPAN, 6, 24, want_step='step_kernel<A=2',
                     want_rollout='rollout_kernel<A=2>', n_streamed=12)


def test_config2_empty16_4agents_4096_envs():
    """BASELINE configs[1]: empty-16-16, 4 agents, slip 0.1, 4096 envs, scen id 1 + e mod 25."""
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    grid = MapfGrid(parse_map_file(map_name_to_files('empty-16-16', 1)[0]))
    _, l2i, nbr = grid.tables()
    per = [parse_scen_file(map_name_to_files('empty-16-16', sid)[1], 4) for sid in range(1, 26)]
    E = 4096
    start = np.asarray([[l2i[l] for l in per[e % 25][0]] for e in range(E)], np.uint16)
    goal = np.asarray([[l2i[l] for l in per[e % 25][1]] for e in range(E)], np.uint16)
    for kernel in KERNELS:
        assert _full_size_check(grid, nbr, 4, start, goal, 0.1, OptimizationCriteria.Makespan, mo.MAKESPAN, 60, 120, kernel) > 0


def test_config3_room32_8agents_65536_envs():
    """BASELINE configs[2] (the bench workload): room-32-32-4, 8 agents, slip 0.2, 65536 envs; every env of
    every step against the C oracle, both criteria."""
    import bench
    grid, _, nbr, start, goal = bench.workload_tables(bench.CONFIGS['c3'], 65536, 0)
    # the run below is large enough to take the kernel's tie path (top 16 bits of a uniform equal to those of a
    # threshold -> 53-bit refinement) many times: thresholds 0.8 and 0.9 have top-16-bit values 52428 and 58982
    ties = 0
    for t in range(12):
        hi16 = np.floor(philox.slip_uniforms_np(42, np.arange(65536), t, 8) * 65536.0).astype(np.int64)
        ties += int(np.isin(hi16, (52428, 58982)).sum())
    assert ties > 50
    assert _full_size_check(grid, nbr, 8, start, goal, 0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, 12, 48) > 0
    assert _full_size_check(grid, nbr, 8, start, goal, 0.2, OptimizationCriteria.SoC, mo.SOC, 4, 24, 'thread_per_env') > 0


def test_config4_share_32768_envs_under_default_dispatch(monkeypatch):
    """BASELINE configs[3]: 262144 room-32-32-4 envs over 8 GPUs -- rank 3's share (global env ids 98304 .. 131071)
    with the library's DEFAULT layout choice (as every test since round 3: layouts are forced only where a test names one)."""
    import bench
    from gym_mapf_amd import sharding
    set_tune(monkeypatch, quad_min_lanes=None)
    offset, count = sharding.split_evenly(bench.CONFIGS['c4']['envs'], 3, 8)
    assert (offset, count) == (98304, 32768)
    grid, _, nbr, start, goal = bench.workload_tables(bench.CONFIGS['c4'], count, offset)
    assert _full_size_check(grid, nbr, 8, start, goal, 0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, 6, 40,
                            env_id_offset=offset) > 0


def test_double_bench_batch_runs_eight_agents_per_lane_by_default(monkeypatch):
    """131072 room-32-32-4 envs of 8 agents -- twice the bench batch, half of BASELINE configs[3] on one GPU: from two
    waves per SIMD on, the library's default choice is the eight-agents-per-lane form of the packed rollout."""
    import bench
    set_tune(monkeypatch, quad_min_lanes=None)
    grid, _, nbr, start, goal = bench.workload_tables(bench.CONFIGS['c4'], 131072, 0)
    assert _full_size_check(grid, nbr, 8, start, goal, 0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, 2, 24) > 0
    env = VecMapfEnv(grid, 8, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42, start_local=start,
                     goal_local=goal)
    env.rollout(4, auto_reset=True, record=True)
    assert 'lq_rollout_kernel<Q=1,K=8,RECORD' in env.last_kernel('rollout'), env.last_kernel('rollout')
    env.close()


def test_config5_bench_tables_16384_envs_under_default_dispatch(monkeypatch):
    """BASELINE configs[4] exactly as `bench.py --config c5 --gpus 8` builds rank 5's share (synthetic map, per-env
    seeded distinct start / goal cells that depend on the global env id only), default layout choice."""
    import bench
    from gym_mapf_amd import sharding
    set_tune(monkeypatch, quad_min_lanes=None)
    cfg = bench.CONFIGS['c5']
    offset, count = sharding.split_evenly(cfg['envs'], 5, 8)
    grid, _, nbr, start, goal = bench.workload_tables(cfg, count, offset)
    whole = bench.workload_tables(cfg, 4096 + 64, offset - 64)          # a slice that straddles the shard boundary
    assert np.array_equal(whole[3][64:], start[:4096]) and np.array_equal(whole[4][64:], goal[:4096])
    assert all(len(set(r.tolist())) == cfg['agents'] for r in start[:512])
    assert _full_size_check(grid, nbr, cfg['agents'], start, goal, cfg['fail_prob'], OptimizationCriteria.Makespan,
                            mo.MAKESPAN, 4, 16, env_id_offset=offset,
                            want_step='lq_step_kernel<Q=8,K=4,NO_TERMINAL> block=') > 500   # (below one residency: the plain step, table rows gathered)


def test_config5_whole_131072_envs_under_default_dispatch(monkeypatch):
    """BASELINE configs[4] whole on ONE GPU (131072 envs x 32 agents, as `bench.py --config c5` builds it): every CU gets a
    1024-thread block, so the default dispatch is the occupancy-bitmap form with 128 bitmaps behind the (delta-row) table -- every env of
    every step against the C oracle (single steps, a streamed recorded rollout, a policy-stream rollout)."""
    import bench
    set_tune(monkeypatch, quad_min_lanes=None)
    cfg = bench.CONFIGS['c5']
    grid, _, nbr, start, goal = bench.workload_tables(cfg, cfg['envs'], 0)
    assert _full_size_check(grid, nbr, cfg['agents'], start, goal, cfg['fail_prob'], OptimizationCriteria.Makespan, mo.MAKESPAN, 2, 8,
                            want_rollout='lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAPD> block=1024',
                            want_step='lq_step_kernel<Q=8,K=4,NO_TERMINAL,DELTA,BITMAP> block=1024', n_streamed=8) > 4000


def test_tall_map_keeps_eight_byte_rows_behind_the_bitmaps(monkeypatch):
    """A map whose columns are taller than 127 cells: a horizontal neighbour's id is more than 127 away from its cell's, so the
    4-byte delta rows do not apply (mapf_create looks at every neighbour) and the bitmap form stages 8-byte rows -- 32 agents
    on an open 160 x 12 map with a few walls, every env of every step against the C oracle."""
    set_tune(monkeypatch, mv_lds_max_bytes='2048')
    rs = np.random.RandomState(7)
    obst = rs.rand(160, 12) < 0.08
    grid = MapfGrid([''.join('@' if obst[r, c] else '.' for c in range(12)) for r in range(160)])
    valid, _, nbr = grid.tables()
    V, E, A = len(valid), 2048, 32
    assert int(np.abs(nbr.astype(np.int64) - np.arange(V)[:, None]).max()) > 127
    r2 = np.random.RandomState(11)
    start = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    assert _full_size_check(grid, nbr, A, start, goal, 0.2, OptimizationCriteria.Makespan, mo.MAKESPAN, 3, 12,
                            want_rollout='lq_rollout_kernel<Q=8,K=4,RECORD,STREAM,MAKESPAN,COMPACT,NO_TERMINAL,BITMAP5> block=512',
                            n_streamed=12) > 50


def test_config5_random64_32agents_16384_envs():
    """BASELINE configs[4], one GPU's share (131072 / 8): synthetic 64x64 map with 20 % obstacles
    (RandomState(20); the reference does not ship random-64-64-20), 32 agents, slip 0.2, seeded random
    distinct starts / goals -- the conflict-heavy stress case."""
    rs = np.random.RandomState(20)
    obst = rs.rand(64, 64) < 0.20
    grid = MapfGrid([''.join('@' if obst[r, c] else '.' for c in range(64)) for r in range(64)])
    valid, _, nbr = grid.tables()
    V, E, A = len(valid), 16384, 32
    r2 = np.random.RandomState(42)
    start = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)      # distinct cells per env
    goal = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    assert _full_size_check(grid, nbr, A, start, goal, 0.2, OptimizationCriteria.SoC, mo.SOC, 6, 24) > 1000


# ----------------------------------------------------------------------- env.P enumeration (mapf_transitions)
def test_transition_tables_match_reference_on_device():
    """env.P[s][a] from the mapf_transitions kernel == the lists the reference enumerated (order, float64
    bits of prob and reward, flags, next cells)."""
    for tab in load_json('transition_tables.json'):
        A = len(tab['starts'])
        env = VecMapfEnv(MapfGrid(tab['lines']), A, tab['starts'], tab['goals'], tab['fail_prob'], tab['r_clash'],
                         tab['r_goal'], tab['r_living'], CRIT[tab['criteria']], n_envs=1)
        local = np.asarray([row['local'] for row in tab['rows']], np.uint16)
        acts = np.asarray([row['actions'] for row in tab['rows']], np.uint8)
        res = env.transitions(local, acts)
        for q, row in enumerate(tab['rows']):
            exp = row['transitions']
            assert int(res['count'][q]) == len(exp), tab['name']
            for b, e in enumerate(exp):
                assert res['next'][q, b].tolist() == e['next_local']
                assert _bits(res['prob'][q, b]) == _bits(e['prob']) and _bits(res['reward'][q, b]) == _bits(e['reward'])
                assert bool(res['done'][q, b]) == e['done'] and bool(res['collision'][q, b]) == e['collision']
        env.close()


@pytest.mark.parametrize('n_agents,criteria', [(1, 'SoC'), (2, 'Makespan'), (3, 'SoC'), (4, 'Makespan'), (5, 'SoC'), (6, 'SoC')])
def test_transitions_random_queries_against_oracle(n_agents, criteria):
    """Random (state, action) queries, including terminal and colliding states and per-query goals, against the
    pinned Python oracle's enumeration."""
    rs = np.random.RandomState(40 + n_agents)
    lines = [''.join('@' if rs.rand() < 0.2 else '.' for _ in range(7)) for _ in range(6)]
    grid = MapfGrid(lines)
    V, A, E, N = len(grid.tables()[0]), n_agents, 5, 60
    start = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    goal = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    fp = 0.3 if n_agents % 2 else 1.0
    env = VecMapfEnv(grid, A, None, None, fp, -20.0, 7.5, -0.5, CRIT[criteria], start_local=start, goal_local=goal)
    valid = grid.tables()[0]
    oracles = [mo.OracleEnv(lines, A, [valid[c] for c in start[e]], [valid[c] for c in goal[e]], fp, -20.0, 7.5, -0.5,
                            OCRIT[criteria]) for e in range(E)]
    env_index = rs.randint(0, E, size=N).astype(np.uint32)
    local = rs.randint(0, V, size=(N, A)).astype(np.uint16)           # duplicates (terminal states) do occur
    local[:5] = goal[env_index[:5]]                                    # all-on-goal terminal states
    acts = rs.randint(0, 5, size=(N, A)).astype(np.uint8)
    res = env.transitions(local, acts, env_index=env_index)
    for q in range(N):
        exp = oracles[env_index[q]].transitions(tuple(int(c) for c in local[q]), acts[q].tolist())
        assert int(res['count'][q]) == len(exp), q
        for b, ((p, c), nxt, r, d) in enumerate(exp):
            assert res['next'][q, b].tolist() == list(nxt), (q, b)
            assert _bits(res['prob'][q, b]) == _bits(p) and _bits(res['reward'][q, b]) == _bits(r), (q, b)
            assert bool(res['done'][q, b]) == d and bool(res['collision'][q, b]) == c, (q, b)
    small = env.transitions(local, acts, max_branches=2, env_index=env_index)       # truncated rows, true counts
    assert np.array_equal(small['count'], res['count'])
    assert np.array_equal(small['next'][:, 0], res['next'][:, 0]) and np.array_equal(_bits(small['prob'][:, 0]), _bits(res['prob'][:, 0]))
    _check_compact_against_reserved(env, local, acts, env_index, res)
    env.close()


def _check_compact_against_reserved(env, local, acts, env_index, res, first=0, window=None):
    """mapf_transitions_compact: the same rows as the reserved form (already checked against the oracle), back to back
    behind the exclusive scan of the windows' lengths; a capacity that is too small drops rows instead of overrunning."""
    N = local.shape[0]
    kw = dict(env_index=env_index, first_branch=first)
    if window is not None:
        kw['max_branches'] = window
    cmp = env.transitions_compact(local, acts, **kw)
    M = res['prob'].shape[1]
    rows = np.minimum(np.maximum(res['count'].astype(np.int64) - first, 0), M)
    assert np.array_equal(cmp['count'], res['count'])
    assert np.array_equal(cmp['offset'].astype(np.int64), np.concatenate([[0], np.cumsum(rows)]))
    for q in range(N):
        o, n = int(cmp['offset'][q]), int(rows[q])
        assert np.array_equal(cmp['next'][o:o + n], res['next'][q, :n]), q
        assert np.array_equal(_bits(cmp['prob'][o:o + n]), _bits(res['prob'][q, :n])), q
        assert np.array_equal(_bits(cmp['reward'][o:o + n]), _bits(res['reward'][q, :n])), q
        assert np.array_equal(cmp['done'][o:o + n], res['done'][q, :n]) and np.array_equal(cmp['collision'][o:o + n], res['collision'][q, :n]), q
    total = int(cmp['offset'][N])
    if total > 3:                                                 # too small a capacity: the needed size is reported, nothing overruns
        cap = total // 2
        part = env.transitions_compact(local, acts, capacity=cap, **kw)
        assert int(part['offset'][N]) == total and part['prob'].shape[0] == cap
        assert np.array_equal(_bits(part['prob']), _bits(cmp['prob'][:cap])) and np.array_equal(part['next'], cmp['next'][:cap])


@pytest.mark.parametrize('n_agents', [2, 4, 8])
def test_compacted_transitions_across_the_scan_block_boundaries(n_agents):
    """The compacted rows' offsets come from a two-level scan: 256 queries per scan block, the blocks' totals added up by the
    emission's waves themselves for calls of up to 256 blocks (65536 queries) and by a second pass beyond.  Query counts on
    both sides of every boundary -- one query, a block +- 1, many blocks, 256 blocks +- 1 (small teams) -- each against the
    reserved-row form of the same queries: offsets = the exclusive scan of the windows' lengths (the grand total in
    offset[N]), every row in its place."""
    rs = np.random.RandomState(31 + n_agents)
    grid = MapfGrid([''.join('@' if rs.rand() < 0.1 else '.' for _ in range(12)) for _ in range(12)])
    V, A = len(grid.tables()[0]), n_agents
    goal = np.argsort(rs.rand(1, V), axis=1)[:, :A].astype(np.uint16)
    env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.SoC, seed=1, start_local=goal[:, ::-1].copy(), goal_local=goal)
    sizes = [1, 255, 256, 257, 2049] + ([65535, 65536, 65537, 70001] if A <= 4 else [5000])
    M = 3 ** A if A <= 4 else 300                                 # (8 agents: a window of 300 rows keeps the reserved form small)
    for N in sizes:
        local = np.argsort(rs.rand(N, V), axis=1)[:, :A].astype(np.uint16)
        local[::97] = goal[0]                                     # terminal states in between: windows of one row
        acts = rs.randint(0, 5, size=(N, A)).astype(np.uint8)
        res = env.transitions(local, acts, max_branches=M)
        cmp = env.transitions_compact(local, acts, max_branches=M)
        rows = np.minimum(res['count'].astype(np.int64), M)
        offset = np.concatenate([[0], np.cumsum(rows)])
        assert np.array_equal(cmp['count'], res['count']) and np.array_equal(cmp['offset'].astype(np.int64), offset), (A, N)
        keep = np.arange(M)[None, :] < rows[:, None]              # the reserved form's live rows, query by query = the compacted rows
        total = int(offset[N])
        assert np.array_equal(cmp['next'][:total], res['next'][keep]), (A, N)
        assert np.array_equal(_bits(cmp['prob'][:total]), _bits(res['prob'][keep])) and np.array_equal(_bits(cmp['reward'][:total]), _bits(res['reward'][keep])), (A, N)
        assert np.array_equal(cmp['done'][:total], res['done'][keep]) and np.array_equal(cmp['collision'][:total], res['collision'][keep]), (A, N)
    env.close()


def test_transitions_of_large_teams_come_in_windows():
    """env.P for more than 8 agents (reference mapf_env.py:448-478 has no limit): the 3^A branches of a query are
    fetched window by window (mapf_transitions_window); the concatenation equals the pinned Python oracle's
    enumeration, whatever the window size, and MapfEnv.P pages through them by itself."""
    from gym_mapf_amd.envs.mapf_env import MapfEnv
    rs = np.random.RandomState(91)
    lines = ['.....', '..@..', '.....', '.....']
    grid = MapfGrid(lines)
    valid = grid.tables()[0]
    V, A = len(valid), 9
    start = rs.choice(V, A, replace=False).astype(np.uint16)
    goal = rs.choice(V, A, replace=False).astype(np.uint16)
    env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.SoC, start_local=start, goal_local=goal)
    orc = mo.OracleEnv(lines, A, [valid[c] for c in start], [valid[c] for c in goal], 0.2, -1000.0, 100.0, -1.0, mo.SOC)
    acts = rs.randint(0, 5, size=(1, A)).astype(np.uint8)
    exp = orc.transitions(tuple(int(c) for c in start), acts[0].tolist())
    assert len(exp) > 2000                                   # several windows below
    for window in (4096, 1000):
        got, first = [], 0
        while first < len(exp):
            res = env.transitions(start.reshape(1, A), acts, max_branches=window, first_branch=first)
            assert int(res['count'][0]) == len(exp)
            k = min(window, len(exp) - first)
            got += [(res['next'][0, b].tolist(), res['prob'][0, b], res['reward'][0, b], bool(res['done'][0, b]),
                     bool(res['collision'][0, b])) for b in range(k)]
            first += window
        for (nxt, p, r, d, c), ((ep, ec), enxt, er, ed) in zip(got, exp):
            assert nxt == list(enxt) and _bits(p) == _bits(ep) and _bits(r) == _bits(er) and (d, c) == (ed, ec)
    env.close()
    # 12 agents through the scalar API: MapfEnv.P pages by itself (window 65536 < 3^12 possible branches)
    lines = ['......', '......', '......']
    starts = tuple((r, c) for r in range(2) for c in range(6))
    goals = tuple((2 - r, 5 - c) for r in range(2) for c in range(6))
    menv = MapfEnv(MapfGrid(lines), 12, starts, goals, 0.1, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan)
    orc = mo.OracleEnv(lines, 12, list(starts), list(goals), 0.1, -1000.0, 100.0, -1.0, mo.MAKESPAN)
    digits = [2] * 11 + [0]
    joint = sum(d * 5 ** i for i, d in enumerate(digits))
    got = menv.P[menv.s][joint]
    exp = orc.transitions(tuple(orc.start), digits)
    assert len(got) == len(exp) > 65536
    for k in (0, 1, 65535, 65536, 65537, len(exp) - 1):
        ((p, c), s_next, r, d), ((ep, ec), enxt, er, ed) = got[k], exp[k]
        assert _bits(p) == _bits(ep) and _bits(r) == _bits(er) and (d, c) == (ed, ec)
        assert s_next == mo.encode_mixed_radix(enxt, orc.V)
    assert abs(sum(t[0][0] for t in got) - 1.0) < 1e-9
    menv.close()


@pytest.mark.parametrize('n_agents', list(range(7, 17)))
def test_every_transitions_kernel_instance_against_oracle(n_agents):
    """mapf_transitions dispatches transitions_rows_kernel<2|4|6|8> up to 8 agents (the 1..6-agent instances: the test
    above) and an exact-size transitions_kernel instance for each of 9..16 (mapf_transitions.hip launch_transitions), each
    with reserved and with compacted rows: every one of them against the pinned Python oracle's enumeration
    (reference mapf_env.py:448-478), both criteria, per-query goals, several window sizes and offsets.  At most seven
    agents of a query move (the others STAY: one-entry lists), which keeps the oracle's list at <= 3^7 branches while
    every agent still takes part in the pair tests, the goal test and the SoC living reward."""
    A = n_agents
    rs = np.random.RandomState(700 + A)
    lines = ['......', '.@....', '....@.', '......', '..@...', '......']
    grid = MapfGrid(lines)
    valid = grid.tables()[0]
    V, E, N = len(valid), 3, 6
    start = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    goal = np.stack([rs.choice(V, A, replace=False) for _ in range(E)]).astype(np.uint16)
    criteria = 'SoC' if A % 2 else 'Makespan'
    env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, CRIT[criteria], start_local=start, goal_local=goal)
    oracles = [mo.OracleEnv(lines, A, [valid[c] for c in start[e]], [valid[c] for c in goal[e]], 0.2, -1000.0, 100.0, -1.0,
                            OCRIT[criteria]) for e in range(E)]
    env_index = (np.arange(N) % E).astype(np.uint32)
    local = np.stack([rs.choice(V, A, replace=False) for _ in range(N)]).astype(np.uint16)
    local[1] = goal[env_index[1]]                                     # terminal: every agent on its goal
    local[2, 1] = local[2, 0]                                         # terminal: two agents share a cell
    local[3] = goal[env_index[3]]
    local[3, A - 1] = grid.tables()[2][goal[env_index[3], A - 1]][1]  # one move short of the goal branch (if not blocked)
    acts = np.zeros((N, A), np.uint8)
    for q in range(N):
        movers = rs.choice(A, min(A, 7), replace=False)
        acts[q, movers] = rs.randint(1, 5, size=len(movers))
    acts[3] = 0
    acts[3, A - 1] = 3                                                # DOWN undoes UP
    exp = [oracles[env_index[q]].transitions(tuple(int(c) for c in local[q]), acts[q].tolist()) for q in range(N)]
    assert max(len(x) for x in exp) > 300 and len(exp[1]) == 1 and len(exp[2]) == 1
    longest = max(len(x) for x in exp)
    for window in (longest, 1000, 257):
        first = 0
        while first < longest:
            res = env.transitions(local, acts, max_branches=window, env_index=env_index, first_branch=first)
            for q in range(N):
                assert int(res['count'][q]) == len(exp[q]), (A, q)
                for b in range(min(window, max(0, len(exp[q]) - first))):
                    (ep, ec), enxt, er, ed = exp[q][first + b]
                    assert res['next'][q, b].tolist() == list(enxt), (A, window, q, first + b)
                    assert _bits(res['prob'][q, b]) == _bits(ep) and _bits(res['reward'][q, b]) == _bits(er), (A, window, q, first + b)
                    assert (bool(res['done'][q, b]), bool(res['collision'][q, b])) == (ed, ec), (A, window, q, first + b)
            _check_compact_against_reserved(env, local, acts, env_index, res, first=first, window=window)   # ... and the compacted rows
            first += window
    assert sum(1 for x in exp for t in x if t[0][1]) > 0              # collision branches were among them
    name = env.last_kernel('transitions')
    assert name.startswith('transitions_rows_kernel<8>' if A <= 8 else 'transitions_kernel<%d,EXACT>' % A) and 'compacted' in name, name
    env.transitions(local, acts, max_branches=257, env_index=env_index)
    assert 'reserved' in env.last_kernel('transitions')
    env.close()


# ----------------------------------------------------------------------- edge cases of the boundary
def test_empty_batch_and_single_cell_map():
    """E = 0 handles are legal no-ops; a 1-cell map with one agent is terminal from the start (start == goal)."""
    grid = MapfGrid(['..', '..'])
    empty = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC,
                       start_local=np.zeros((0, 2), np.uint16), goal_local=np.zeros((0, 2), np.uint16), n_envs=0)
    local, reward, done, info = empty.step(np.zeros((0, 2), np.uint8), auto_reset=True)
    assert local.shape == (0, 2) and reward.shape == (0,) and empty.t == 1
    assert empty.rollout(5)['returns'].shape == (0,) and empty.t == 6
    empty.reset(), empty.close()
    one = VecMapfEnv(MapfGrid(['.']), 1, ((0, 0),), ((0, 0),), 0.3, -1.0, 1.0, -1.0, OptimizationCriteria.Makespan, n_envs=3)
    local, reward, done, info = one.step(np.full((3, 1), 2, np.uint8))
    assert np.all(info['was_terminal'] == 1) and np.all(reward == 0) and np.all(info['prob'] == 0) and np.all(done == 1)
    one.close()


def test_mixed_map_batch_keeps_every_envs_own_stream():
    """Every reference env owns its grid (mapf_env.py:127).  MultiMapVecEnv steps a batch whose envs live on three
    different maps (one handle per run of consecutive envs on the same map) -- each env against its own pure-Python
    oracle with the uniforms of its GLOBAL id, single steps and a fused rollout."""
    from gym_mapf_amd.envs.multi_map import MultiMapVecEnv
    rs = np.random.RandomState(17)
    maps = [['....', '.@..', '....'], ['.....', '..@..', '.....', '.....'], ['...', '...', '...']]
    A, E, off = 3, 14, 1000
    pick = [0, 0, 0, 1, 1, 2, 2, 2, 2, 0, 1, 1, 0, 2]
    grids = [MapfGrid(m) for m in maps]
    starts, goals, oracles = [], [], []
    for e in range(E):
        valid = grids[pick[e]].tables()[0]
        s = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        g = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        starts.append(s), goals.append(g)
        oracles.append(mo.OracleEnv(maps[pick[e]], A, s, g, 0.3, -10.0, 5.0, -1.0, mo.SOC))
    env = MultiMapVecEnv([grids[k] for k in pick], A, starts, goals, 0.3, -10.0, 5.0, -1.0, OptimizationCriteria.SoC,
                         seed=8, env_id_offset=off)
    assert env.n_handles == 7 and len(env.grids) == 3
    ids = off + np.arange(E)
    for t in range(40):
        acts = philox.random_actions_np(8, ids, t, A)
        u = philox.slip_uniforms_np(8, ids, t, A)
        local, reward, done, info = env.step(acts, auto_reset=True)
        for e, o in enumerate(oracles):
            nxt, r, d, c, p, wt = o.step(acts[e].tolist(), u[e].tolist())
            assert list(nxt) == local[e].tolist() and _bits(r) == _bits(reward[e]) and _bits(p) == _bits(info['prob'][e]), (t, e)
            assert (d, c, wt) == (bool(done[e]), bool(info['collision'][e]), bool(info['was_terminal'][e])), (t, e)
            if d:
                o.reset()
    res = env.rollout(25, auto_reset=True)                         # in-kernel policy stream = philox.random_actions_np
    for e, o in enumerate(oracles):
        ret, epi = 0.0, 0
        for t in range(40, 65):
            a = philox.random_actions_np(8, [off + e], t, A)[0].tolist()
            nxt, r, d, c, p, wt = o.step(a, philox.slip_uniforms_np(8, [off + e], t, A)[0].tolist())
            ret, epi = ret + r, epi + int(d)
            if d:
                o.reset()
        assert _bits(ret) == _bits(res['returns'][e]) and epi == res['episodes'][e], e
    state, t_now = env.get_state()
    assert t_now == 65 and all(state[e].tolist() == list(o.local) for e, o in enumerate(oracles))
    env.close()


def test_mixed_map_batch_in_one_launch_through_the_union_table():
    """UnionMapVecEnv: the batch's distinct maps side by side in ONE move table (block-diagonal neighbour table) behind one
    handle -- a step / rollout of the mixed batch is one launch.  Every env against its own pure-Python oracle (cells in its
    OWN map's numbering, the uniforms of its global id), then a 4096-env batch on three 16x16 maps in an interleaved order
    against MultiMapVecEnv (one handle per run), host and device mode."""
    from gym_mapf_amd.envs.multi_map import MultiMapVecEnv, UnionMapVecEnv
    rs = np.random.RandomState(17)
    maps = [['....', '.@..', '....'], ['.....', '..@..', '.....', '.....'], ['...', '...', '...']]
    A, E, off = 3, 14, 1000
    pick = [0, 0, 0, 1, 1, 2, 2, 2, 2, 0, 1, 1, 0, 2]
    grids = [MapfGrid(m) for m in maps]
    starts, goals, oracles = [], [], []
    for e in range(E):
        valid = grids[pick[e]].tables()[0]
        s = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        g = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        starts.append(s), goals.append(g)
        oracles.append(mo.OracleEnv(maps[pick[e]], A, s, g, 0.3, -10.0, 5.0, -1.0, mo.SOC))
    env = UnionMapVecEnv([grids[k] for k in pick], A, starts, goals, 0.3, -10.0, 5.0, -1.0, OptimizationCriteria.SoC, seed=8, env_id_offset=off)
    assert env.n_handles == 1 and len(env.grids) == 3
    ids = off + np.arange(E)
    for t in range(40):
        acts = philox.random_actions_np(8, ids, t, A)
        u = philox.slip_uniforms_np(8, ids, t, A)
        local, reward, done, info = env.step(acts, auto_reset=True)
        for e, o in enumerate(oracles):
            nxt, r, d, c, p, wt = o.step(acts[e].tolist(), u[e].tolist())
            assert list(nxt) == local[e].tolist() and _bits(r) == _bits(reward[e]) and _bits(p) == _bits(info['prob'][e]), (t, e)
            assert (d, c, wt) == (bool(done[e]), bool(info['collision'][e]), bool(info['was_terminal'][e])), (t, e)
            if d:
                o.reset()
    res = env.rollout(25, auto_reset=True, record=True)            # in-kernel policy stream; recorded cells in each map's own ids
    for e, o in enumerate(oracles):
        for k, t in enumerate(range(40, 65)):
            a = philox.random_actions_np(8, [off + e], t, A)[0].tolist()
            nxt, r, d, c, p, wt = o.step(a, philox.slip_uniforms_np(8, [off + e], t, A)[0].tolist())
            assert list(nxt) == res['local'][k, e].tolist() and _bits(r) == _bits(res['reward'][k, e]), (e, t)
            if d:
                o.reset()
    state, t_now = env.get_state()
    assert t_now == 65 and all(state[e].tolist() == list(o.local) for e, o in enumerate(oracles))
    env.set_state(state, t=65)                                     # (round trip through the union numbering)
    assert np.array_equal(env.get_state()[0], state)
    env.close()
    # a batch large enough for the packed kernels, maps interleaved in runs of 32 envs: one launch against one per run
    rs = np.random.RandomState(23)
    big = [MapfGrid([''.join('@' if rs.rand() < 0.15 else '.' for _ in range(16)) for _ in range(16)]) for _ in range(3)]
    E, A = 4096, 4
    order = [big[(e // 32) % 3] for e in range(E)]
    starts, goals = [], []
    for e in range(E):
        valid = order[e].tables()[0]
        cells = rs.choice(len(valid), 2 * A, replace=False)
        starts.append([valid[i] for i in cells[:A]]), goals.append([valid[i] for i in cells[A:]])
    args = (order, A, starts, goals, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan)
    one, many = UnionMapVecEnv(*args, seed=5), MultiMapVecEnv(*args, seed=5)
    assert many.n_handles == 128
    acts = rs.randint(0, 5, size=(6, E, A)).astype(np.uint8)
    for t in range(6):
        a, b = one.step(acts[t], auto_reset=True), many.step(acts[t], auto_reset=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1])) and np.array_equal(a[2], b[2]), t
        assert np.array_equal(_bits(a[3]['prob']), _bits(b[3]['prob'])) and np.array_equal(a[3]['collision'], b[3]['collision']), t
    ra, rb = one.rollout(20, auto_reset=True), many.rollout(20, auto_reset=True)
    assert np.array_equal(_bits(ra['returns']), _bits(rb['returns'])) and np.array_equal(ra['episodes'], rb['episodes'])
    assert np.array_equal(one.get_state()[0], many.get_state()[0]) and int(ra['episodes'].sum()) > 0
    assert one.last_kernel('rollout').startswith('lq_rollout_kernel'), one.last_kernel('rollout')
    one.close(), many.close()
    import torch
    dev = UnionMapVecEnv(*args, seed=5, device_arrays=True)
    ref = MultiMapVecEnv(*args, seed=5)
    with torch.cuda.stream(torch.cuda.ExternalStream(dev.stream)):
        acts_t = torch.from_numpy(acts).cuda()
    for t in range(3):
        a = dev.step(acts_t[t], auto_reset=True)
        dev.sync()
        b = ref.step(acts[t], auto_reset=True)
        assert np.array_equal(a[0].cpu().numpy(), b[0]) and np.array_equal(_bits(a[1].cpu().numpy()), _bits(b[1])), t
    dev.close(), ref.close()


def test_rollout_beyond_one_launch_is_issued_in_slices(monkeypatch):
    """The C ABI rejects a launch whose largest array exceeds 4 GiB or that has more than 65535 steps; VecMapfEnv.rollout
    then issues the steps as several launches over consecutive slices of the same arrays.  Forced here with a limit of
    five steps per launch: 13 recorded steps in three launches == the same 13 steps in one, and == the C oracle."""
    rs = np.random.RandomState(5)
    lines = [''.join('@' if rs.rand() < 0.1 else '.' for _ in range(10)) for _ in range(10)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    V, E, A, T = len(valid), 256, 8, 13
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    mk = lambda: VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.SoC, seed=4, start_local=start, goal_local=goal)
    one, sliced = mk(), mk()
    acts = np.stack([philox.random_actions_np(4, np.arange(E), t, A) for t in range(T)])
    whole = one.rollout(T, actions=acts, auto_reset=True, record=True)
    monkeypatch.setattr(VecMapfEnv, '_MAX_LAUNCH_STEPS', 5)
    parts = sliced.rollout(T, actions=acts, auto_reset=True, record=True)
    for k in ('local', 'done', 'collision', 'episodes', 'collisions'):
        assert np.array_equal(whole[k], parts[k]), k
    for k in ('reward', 'prob', 'returns'):
        assert np.array_equal(_bits(whole[k]), _bits(parts[k])), k
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, -1000.0, 100.0, -1.0, mo.SOC, seed=4)
    ref = co.rollout(T, actions=acts, auto_reset=True)
    assert np.array_equal(_bits(parts['returns']), _bits(ref['returns'])) and np.array_equal(parts['episodes'], ref['episodes'])
    assert sliced.t == one.t == T and np.array_equal(sliced.get_state()[0], co.state)
    more = sliced.rollout(7, auto_reset=True, accumulate_into={k: parts[k] for k in ('returns', 'episodes', 'collisions')})   # policy, no arrays: one launch
    ref2 = co.rollout(7, auto_reset=True)
    assert np.array_equal(more['episodes'], ref['episodes'] + ref2['episodes'])
    one.close(), sliced.close()


def test_mapf_tune_is_the_one_override_and_rejects_what_it_does_not_know(monkeypatch):
    """MAPF_TUNE="key=value,...": read at mapf_create; a typo must fail the create call, not silently measure the default."""
    from gym_mapf_amd import _native as nat
    grid = MapfGrid(['....', '....', '....', '....'])
    make = lambda: VecMapfEnv(grid, 4, ((0, 0), (1, 1), (2, 2), (3, 3)), ((3, 0), (2, 1), (1, 2), (0, 3)), 0.2, -1000.0, 100.0, -1.0,
                              OptimizationCriteria.Makespan, n_envs=4096)
    for bad, what in (('k=4,no_such_key=1', 'unknown key'), ('k', 'malformed'), ('k=four', 'malformed')):
        monkeypatch.setenv('MAPF_TUNE', bad)
        with pytest.raises(nat.MapfNativeError) as err:
            make()
        assert what in str(err.value) and err.value.code == nat.MAPF_EINVAL, str(err.value)
    monkeypatch.setenv('MAPF_TUNE', 'k=2,quad_min_lanes=0')
    env = make()
    env.rollout(4, auto_reset=True, record=True)
    assert 'lq_rollout_kernel<Q=2,K=2' in env.last_kernel('rollout'), env.last_kernel('rollout')
    env.close()
    monkeypatch.setenv('MAPF_TUNE', 'quad_lanes=0')
    env = make()
    env.rollout(4, auto_reset=True, record=True)
    assert env.last_kernel('rollout').startswith('lg_rollout_kernel'), env.last_kernel('rollout')
    env.close()


def test_out_of_range_actions_are_stay_and_device_pointers_must_be_aligned():
    import torch
    grid = MapfGrid(['....', '....'])
    kw = dict(start_local=np.array([[0, 5]], np.uint16).repeat(64, 0), goal_local=np.array([[7, 2]], np.uint16).repeat(64, 0))
    a = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, seed=3, **kw)
    b = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, seed=3, **kw)
    la, ra, da, ia = a.step(np.full((64, 2), 200, np.uint8))              # garbage action values ...
    lb, rb, db, ib = b.step(np.zeros((64, 2), np.uint8))                  # ... behave like STAY
    assert np.array_equal(la, lb) and np.array_equal(_bits(ra), _bits(rb)) and np.array_equal(_bits(ia['prob']), _bits(ib['prob']))
    a.close(), b.close()
    dev = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, device_arrays=True, **kw)
    buf = torch.zeros(64 * 2 + 3, dtype=torch.uint8, device='cuda')
    with pytest.raises(Exception) as err:
        dev.step(buf[3:].view(64, 2))                                     # data_ptr() not 16-byte aligned
    assert 'aligned' in str(err.value)
    local, reward, done, info = dev.step(buf[:128].view(64, 2))
    dev.sync()
    assert local.shape == (64, 2) and bool((info['was_terminal'] == 0).all())
    dev.close()


def test_rollout_with_large_lds_move_table():
    """A 40x40 map (V ~ 1440): the move table is ~57 KB, i.e. LDS-resident but beyond the default dynamic-LDS cap
    (needs the explicit opt-in) -- fused rollout against the C oracle, 4 and 8 agents."""
    rs = np.random.RandomState(77)
    lines = [''.join('@' if rs.rand() < 0.1 else '.' for _ in range(40)) for _ in range(40)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    V, E = len(valid), 8192
    assert 32 * 1024 < V * 40 < 80 * 1024
    for A in (4, 8):
        start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
        goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
        env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.SoC, seed=9,
                         start_local=start, goal_local=goal)
        co = c_oracle.COracle(nbr, A, start, goal, 0.2, -1000.0, 100.0, -1.0, mo.SOC, seed=9)
        res, ref = env.rollout(40, auto_reset=True, record=True), co.rollout(40, auto_reset=True)
        assert np.array_equal(_bits(res['returns']), _bits(ref['returns'])) and np.array_equal(res['episodes'], ref['episodes'])
        assert np.array_equal(env.get_state()[0], co.state)
        env.close()


def test_mixed_map_batch_in_device_mode_on_one_stream():
    """MultiMapVecEnv with device_arrays=True (mapf_env.py:127: every env owns its grid): torch tensors in and out, every
    run's launch on ONE stream, nothing waits.  48 envs on three maps in runs of 16 / 16 / 5 / 11: the runs that start at a
    multiple of 16 envs work on the caller's tensors in place, the last one (env 37) through staging copies -- each env
    against its own pure-Python oracle, prepared step replayed 30 times, then a fused rollout, then the gathered state."""
    import torch
    from gym_mapf_amd.envs.multi_map import MultiMapVecEnv
    rs = np.random.RandomState(23)
    maps = [['....', '.@..', '....'], ['.....', '..@..', '.....', '.....'], ['...', '...', '...']]
    A, E, off = 4, 48, 5000
    pick = [0] * 16 + [1] * 16 + [2] * 5 + [0] * 11
    grids = [MapfGrid(m) for m in maps]
    starts, goals, oracles = [], [], []
    for e in range(E):
        valid = grids[pick[e]].tables()[0]
        s = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        g = [valid[i] for i in rs.choice(len(valid), A, replace=False)]
        starts.append(s), goals.append(g)
        oracles.append(mo.OracleEnv(maps[pick[e]], A, s, g, 0.3, -10.0, 5.0, -1.0, mo.MAKESPAN))
    env = MultiMapVecEnv([grids[k] for k in pick], A, starts, goals, 0.3, -10.0, 5.0, -1.0, OptimizationCriteria.Makespan,
                         seed=9, env_id_offset=off, device_arrays=True)
    assert env.n_handles == 4 and env.stream
    ids = off + np.arange(E)
    actions = torch.zeros((E, A), dtype=torch.uint8, device='cuda')
    call, out = env.prepare_step(actions, auto_reset=True)
    assert call.in_place_runs == 3                                   # runs at envs 0, 16, 32; the run at env 37 is staged
    stream = torch.cuda.ExternalStream(env.stream)
    for t in range(30):
        acts = philox.random_actions_np(9, ids, t, A)
        with torch.cuda.stream(stream):
            actions.copy_(torch.as_tensor(acts, device='cuda'))       # the caller refills its tensor in place, on that stream
        call()
        env.sync()
        u = philox.slip_uniforms_np(9, ids, t, A)
        local, reward, done = out['local'].cpu().numpy(), out['reward'].cpu().numpy(), out['done'].cpu().numpy()
        prob, coll, wt_ = out['prob'].cpu().numpy(), out['collision'].cpu().numpy(), out['was_terminal'].cpu().numpy()
        for e, o in enumerate(oracles):
            nxt, r, d, c, p, wt = o.step(acts[e].tolist(), u[e].tolist())
            assert list(nxt) == local[e].tolist() and _bits(r) == _bits(reward[e]) and _bits(p) == _bits(prob[e]), (t, e)
            assert (d, c, wt) == (bool(done[e]), bool(coll[e]), bool(wt_[e])), (t, e)
            if d:
                o.reset()
    acts_ro = np.stack([philox.random_actions_np(10, ids, 30 + k, A) for k in range(12)])
    res = env.rollout(12, actions=torch.as_tensor(acts_ro, device='cuda'), auto_reset=True)
    env.sync()
    returns, episodes = res['returns'].cpu().numpy(), res['episodes'].cpu().numpy()
    for e, o in enumerate(oracles):
        ret, epi = 0.0, 0
        for k in range(12):
            nxt, r, d, c, p, wt = o.step(acts_ro[k, e].tolist(), philox.slip_uniforms_np(9, [off + e], 30 + k, A)[0].tolist())
            ret, epi = ret + r, epi + int(d)
            if d:
                o.reset()
        assert _bits(ret) == _bits(returns[e]) and epi == episodes[e], e
    state, t_now = env.get_state()
    env.sync()
    state = state.cpu().numpy()
    assert t_now == 42 and all(state[e].tolist() == list(o.local) for e, o in enumerate(oracles))
    env.close()
