"""GPU parity of the single-step fast path added in round 3: mapf_step launches recorded into a hipGraph and replayed
(device-side step index: every replay must draw fresh random numbers), the scenario table (start / goal rows looked up
through one byte per env) and the handle's state view (out_local = NULL: the cells are written once).  Everything is
compared with the C oracle bit for bit: cells, flags and the float64 bit patterns of reward and prob."""
import numpy as np
import pytest

import c_oracle
from conftest import set_tune
import mapf_oracle as mo
from gym_mapf_amd import _native as nat
from gym_mapf_amd.envs.grid import MapfGrid
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

pytestmark = pytest.mark.gpu
R = (-1000.0, 100.0, -1.0)


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def _c3_tables(n_envs, offset=0):
    import bench
    return bench.workload_tables(bench.CONFIGS['c3'], n_envs, offset)


def _check(out, ref, tag, local=True):
    if local:
        assert np.array_equal(out['local'].cpu().numpy(), ref['local']), tag
    assert np.array_equal(_bits(out['reward'].cpu().numpy()), _bits(ref['reward'])), tag
    assert np.array_equal(_bits(out['prob'].cpu().numpy()), _bits(ref['prob'])), tag
    assert np.array_equal(out['done'].cpu().numpy(), ref['done']), tag
    assert np.array_equal(out['collision'].cpu().numpy(), ref['collision']), tag
    assert np.array_equal(out['was_terminal'].cpu().numpy(), ref['was_terminal']), tag


@pytest.mark.parametrize('criteria', ['Makespan', 'SoC'])
def test_graph_replay_of_recorded_steps_matches_the_oracle(criteria):
    """16 mapf_step calls recorded once, replayed 6 times = 96 env-steps of 2048 room-32-32-4 envs (the bench workload's
    tables, six scenarios -> scenario table): every step of every replay against the C oracle.  The odd nodes leave
    out_local out and are checked through the state view the next node would read."""
    E, A, N = 2048, 8, 16
    grid, _, nbr, start, goal = _c3_tables(E, 4096)
    crit, ocrit = (OptimizationCriteria.SoC, mo.SOC) if criteria == 'SoC' else (OptimizationCriteria.Makespan, mo.MAKESPAN)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, crit, seed=7, env_id_offset=4096, device_arrays=True,
                     start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, ocrit, seed=7, env_id_offset=4096)
    actions = env.fill_random_actions(0, N)
    env.sync()
    acts_host = actions.cpu().numpy()
    # three plain steps first: the recording must pick the step index up from where the host side left it
    for t in range(3):
        local, reward, done, info = env.step(actions[t], auto_reset=True)
        env.sync()
        _check(dict(local=local, reward=reward, done=done, **info), co.step(acts_host[t], auto_reset=True), 'plain %d' % t)
    assert 'SCEN' in env.last_kernel('step'), env.last_kernel('step')
    env.graph_begin()
    outs = []
    for k in range(N):
        call, out = env.prepare_step(actions[k], auto_reset=True, write_local=(k % 2 == 0))
        call()
        outs.append(out)
    with pytest.raises(nat.MapfNativeError):
        env.sync()                                   # waiting for the stream is refused while recording
    assert 'NO_TERMINAL' in env.last_kernel('step')  # (a later recorded step follows a recorded auto-reset step)
    graph = env.graph_end()
    assert graph.steps == N and env.t == 3           # recording executed nothing
    view = env.state_view()
    for rep in range(6):
        graph.launch(1)
        env.sync()
        for k in range(N):
            ref = co.step(acts_host[k], auto_reset=True)
            _check(outs[k], ref, 'replay %d node %d' % (rep, k), local=(k % 2 == 0))
        assert np.array_equal(view.cpu().numpy(), co.state), rep      # the state the next step starts from
        assert env.t == 3 + (rep + 1) * N
    # several replays in one call, then plain steps again: same stream of random numbers throughout
    graph.launch(4)
    env.sync()
    for rep in range(4):
        refs = [co.step(acts_host[k], auto_reset=True) for k in range(N)]
    for k in range(N):                               # the output arrays hold the last replay's results
        _check(outs[k], refs[k], 'batched replays, node %d' % k, local=(k % 2 == 0))
    assert np.array_equal(env.get_state()[0].cpu().numpy(), co.state) and env.t == co.t == 3 + 10 * N
    local, reward, done, info = env.step(actions[5], auto_reset=True)
    env.sync()
    _check(dict(local=local, reward=reward, done=done, **info), co.step(acts_host[5], auto_reset=True), 'plain after graph')
    graph.close()
    env.close()


def test_graph_of_rollout_and_steps_and_host_side_index_moves():
    """A recording that mixes a fused rollout (8 steps, in-kernel policy) with two single steps; between replays the host
    side moves the step index (set_state), which the next replay must honour."""
    E, A = 1024, 8
    grid, _, nbr, start, goal = _c3_tables(E)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.Makespan, seed=11, device_arrays=True,
                     start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.MAKESPAN, seed=11)
    actions = env.fill_random_actions(100, 2)
    env.sync()
    acts_host = actions.cpu().numpy()
    env.graph_begin()
    res = env.rollout(8, auto_reset=True)
    outs = []
    for k in range(2):
        call, out = env.prepare_step(actions[k], auto_reset=True)
        call()
        outs.append(out)
    graph = env.graph_end()
    assert graph.steps == 10
    for rep, t0 in enumerate((0, 10, 1000, 1010)):
        if t0 == 1000:
            env.set_state(None, t=1000)
            co.t = 1000
        graph.launch(1)
        env.sync()
        ref = co.rollout(8, auto_reset=True)
        assert np.array_equal(_bits(res['returns'].cpu().numpy()), _bits(ref['returns'])), rep
        assert np.array_equal(res['episodes'].cpu().numpy(), ref['episodes']), rep
        for k in range(2):
            _check(outs[k], co.step(acts_host[k], auto_reset=True), 'replay %d step %d' % (rep, k))
        assert env.t == co.t == t0 + 10
    graph.close()
    env.close()


@pytest.mark.parametrize('n_agents,n_envs', [(4, 512), (8, 256), (16, 256), (32, 128)])
def test_scenario_table_step_equals_plain_rows(monkeypatch, n_agents, n_envs):
    """The packed single step with and without the scenario table (MAPF_TUNE scen_table=0), 5 scenarios, goals reachable so
    that episodes end and auto-reset takes the start rows from the table: both against the C oracle, and the library
    must report the intended kernel instance."""
    rs = np.random.RandomState(100 + n_agents)
    lines = [''.join('@' if rs.rand() < 0.08 else '.' for _ in range(12)) for _ in range(12)]
    grid = MapfGrid(lines)
    valid, _, nbr = grid.tables()
    V = len(valid)
    scen_start = np.argsort(rs.rand(5, V), axis=1)[:, :n_agents].astype(np.uint16)
    scen_goal = scen_start.copy()
    for s in range(5):                               # goals one move away from the starts: episodes end quickly
        for i in range(n_agents):
            scen_goal[s, i] = nbr[scen_start[s, i]][1 + (s + i) % 4]
    which = rs.randint(0, 5, n_envs)
    start, goal = np.ascontiguousarray(scen_start[which]), np.ascontiguousarray(scen_goal[which])
    for use_table in (True, False):
        if use_table:
            set_tune(monkeypatch, scen_table=None)
        else:
            set_tune(monkeypatch, scen_table='0')
        env = VecMapfEnv(grid, n_agents, None, None, 0.1, *R, OptimizationCriteria.SoC, seed=5, start_local=start, goal_local=goal)
        co = c_oracle.COracle(nbr, n_agents, start, goal, 0.1, *R, mo.SOC, seed=5)
        n_done = 0
        for t in range(24):
            acts = ((np.arange(n_envs)[:, None] + np.arange(n_agents)[None, :] + t + which[:, None]) % 5).astype(np.uint8)
            acts[t % 3::3] = ((1 + (which[t % 3::3, None] + np.arange(n_agents)[None, :]) % 4)).astype(np.uint8)   # head for the goals
            local, reward, done, info = env.step(acts, auto_reset=(t % 8 != 7))
            ref = co.step(acts, auto_reset=(t % 8 != 7))
            assert np.array_equal(local, ref['local']) and np.array_equal(_bits(reward), _bits(ref['reward'])), (use_table, t)
            assert np.array_equal(_bits(info['prob']), _bits(ref['prob'])) and np.array_equal(done, ref['done']), (use_table, t)
            assert np.array_equal(info['collision'], ref['collision']) and np.array_equal(info['was_terminal'], ref['was_terminal']), (use_table, t)
            assert np.array_equal(env.get_state()[0], co.state), (use_table, t)
            n_done += int(done.sum())
        name = env.last_kernel('step')
        assert name.startswith('lq_step_kernel') and (',SCEN' in name) == use_table, name
        assert n_done > 0
        env.close()


@pytest.mark.parametrize('big', ['2', '2k4', '0'])
def test_single_step_at_a_batch_larger_than_the_device_holds(monkeypatch, big):
    """393216 envs x 8 agents: more than an MI355X holds at once.  MAPF_TUNE step_big=2 forces the form the library uses from
    1 M envs on -- a resident grid of 512 blocks of 1024 threads with the move table in LDS, walking 768 chunks (half
    of the blocks take two) -- step_big=0 the one-block-per-256-lanes form.  Six steps against the C oracle, the
    first one with the is_terminal test (a state set by the caller), then without."""
    set_tune(monkeypatch, step_big=big[0])
    if big == '2k4':
        set_tune(monkeypatch, k='4')                      # (the BIG form prefers eight agents per lane: pin four)
    expect = {'2': 'lq_step_kernel<Q=1,K=8,SCEN', '2k4': 'lq_step_kernel<Q=2,K=4,SCEN', '0': 'lq_step_kernel<Q=2,K=4,SCEN'}[big]
    E, A = 393216, 8
    grid, _, nbr, start, goal = _c3_tables(E)
    import philox
    ids = np.arange(E)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.SoC, seed=3, start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.SOC, seed=3)
    env.set_state(np.ascontiguousarray(start))            # (same cells; the library no longer knows they are not terminal)
    for t in range(6):
        acts = philox.random_actions_np(3, ids, t, A)
        local, reward, done, info = env.step(acts, auto_reset=True)
        ref = co.step(acts, auto_reset=True)
        assert np.array_equal(local, ref['local']) and np.array_equal(_bits(reward), _bits(ref['reward'])), t
        assert np.array_equal(_bits(info['prob']), _bits(ref['prob'])) and np.array_equal(done, ref['done']), t
        assert np.array_equal(info['collision'], ref['collision']) and np.array_equal(info['was_terminal'], ref['was_terminal'])
        name = env.last_kernel('step')
        assert name.startswith(expect) and (',BIG>' in name) == (big != '0'), name
        assert ('NO_TERMINAL' in name) == (t > 0), name
    assert np.array_equal(env.get_state()[0], co.state)
    env.close()


@pytest.mark.parametrize('n_agents,n_envs', [(4, 4096), (16, 1024), (32, 512)])
def test_big_form_of_the_single_step_at_other_team_sizes(monkeypatch, n_agents, n_envs):
    """The resident-grid / LDS-table form at 4, 16 and 32 agents (one, four and eight lanes per env), forced on small
    batches, goal-seeking actions so that episodes end; against the C oracle."""
    import goal_scenarios
    set_tune(monkeypatch, step_big='2')
    A, E = n_agents, n_envs
    lines, start_loc, goal_loc = goal_scenarios.goal_scenario(A, E, 5150 + A)
    grid = MapfGrid(lines)
    valid, l2i, nbr = grid.tables()
    ids = np.zeros((len(lines), len(lines[0])), np.uint16)
    for loc, k in l2i.items():
        ids[loc] = k
    start = np.ascontiguousarray(ids[start_loc[..., 0], start_loc[..., 1]])
    goal = np.ascontiguousarray(ids[goal_loc[..., 0], goal_loc[..., 1]])
    rc = np.asarray([r | (c << 16) for r, c in valid], np.uint32)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.Makespan, seed=21, start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.MAKESPAN, seed=21)
    n_done = 0
    for t in range(10):
        acts = co.greedy_actions(rc)
        auto = t % 4 != 3
        local, reward, done, info = env.step(acts, auto_reset=auto)
        ref = co.step(acts, auto_reset=auto)
        assert np.array_equal(local, ref['local']) and np.array_equal(_bits(reward), _bits(ref['reward'])), t
        assert np.array_equal(_bits(info['prob']), _bits(ref['prob'])) and np.array_equal(done, ref['done']), t
        assert np.array_equal(info['collision'], ref['collision']) and np.array_equal(info['was_terminal'], ref['was_terminal']), t
        assert np.array_equal(env.get_state()[0], co.state), t
        n_done += int(done.sum())
    assert ',BIG>' in env.last_kernel('step') and n_done > 0, env.last_kernel('step')
    env.close()


def test_step_drops_is_terminal_only_when_no_env_can_be_terminal():
    """The packed single step has an instance without is_terminal(prev) (NO_TERMINAL).  The library may use it only when no
    env can be terminal: after a step that auto-reset every finished episode -- never for the first recorded step of a
    graph, after a step without auto-reset, or after set_state.  Goal-seeking actions on the goal-scenario map make
    episodes end all the time; every step against the C oracle."""
    import goal_scenarios
    A, E = 8, 512
    lines, start_loc, goal_loc = goal_scenarios.goal_scenario(A, E, 4242)
    grid = MapfGrid(lines)
    valid, l2i, nbr = grid.tables()
    ids = np.zeros((len(lines), len(lines[0])), np.uint16)
    for loc, k in l2i.items():
        ids[loc] = k
    start = np.ascontiguousarray(ids[start_loc[..., 0], start_loc[..., 1]])
    goal = np.ascontiguousarray(ids[goal_loc[..., 0], goal_loc[..., 1]])
    rc = np.asarray([r | (c << 16) for r, c in valid], np.uint32)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.SoC, seed=9, start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.SOC, seed=9)
    seen = []
    plan = [True, True, True, False, False, True, True, 'set_state', True, True, False, True]
    n_terminal = 0
    for what in plan:
        if what == 'set_state':
            terminal_state = np.ascontiguousarray(goal)              # every agent on its goal: terminal
            env.set_state(terminal_state)
            co.state[:] = terminal_state
            continue
        acts = co.greedy_actions(rc)
        local, reward, done, info = env.step(acts, auto_reset=what)
        ref = co.step(acts, auto_reset=what)
        assert np.array_equal(local, ref['local']) and np.array_equal(_bits(reward), _bits(ref['reward'])), len(seen)
        assert np.array_equal(_bits(info['prob']), _bits(ref['prob'])) and np.array_equal(done, ref['done']), len(seen)
        assert np.array_equal(info['collision'], ref['collision']) and np.array_equal(info['was_terminal'], ref['was_terminal']), len(seen)
        assert np.array_equal(env.get_state()[0], co.state)
        n_terminal += int(ref['was_terminal'].sum())
        seen.append('NO_TERMINAL' in env.last_kernel('step'))
    # step 0 follows create (starts are not terminal here): no test needed; 1, 2 follow auto-reset steps; 3 follows one too;
    # 4 follows a step without auto-reset; 5 likewise; 6 follows an auto-reset step; 7 follows set_state; ...
    assert seen == [True, True, True, True, False, False, True, False, True, True, False], seen
    assert n_terminal > 0
    env.close()


def test_recording_with_the_callers_own_kernels_on_a_shared_stream():
    """INTEGRATION.md's loop: the env enqueues on a stream the caller owns (a torch stream), the caller's policy -- torch
    kernels reading the zero-copy state view and writing the action tensor in place -- and ONE mapf_step are recorded
    together and replayed 40 times; every replay must see the state the previous one left and draw fresh random
    numbers.  The same policy in numpy drives the C oracle."""
    import torch
    E, A = 4096, 8
    grid, _, nbr, start, goal = _c3_tables(E)
    stream = torch.cuda.Stream()
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.Makespan, seed=13, device_arrays=True,
                     start_local=start, goal_local=goal, stream=stream.cuda_stream)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.MAKESPAN, seed=13)
    with torch.cuda.stream(stream):
        obs = env.state_view()
        actions = torch.zeros((E, A), dtype=torch.uint8, device='cuda')
        tmp = torch.zeros((E, A), dtype=torch.int32, device='cuda')
        lane = (torch.arange(E * A, dtype=torch.int32, device='cuda') % 7).view(E, A)
        total = torch.zeros(E, dtype=torch.float64, device='cuda')
        stream.synchronize()
        env.graph_begin()
        tmp.copy_(obs)                               # the "policy": a = (cell + lane pattern) mod 5, all in place
        tmp.add_(lane)
        tmp.remainder_(5)
        actions.copy_(tmp)
        call, out = env.prepare_step(actions, auto_reset=True, write_local=False)
        call()
        total.add_(out['reward'])                    # a consumer of the step's outputs, recorded as well
        graph = env.graph_end()
        lane_host = lane.cpu().numpy()
        ref_total = np.zeros(E)
        for rep in range(40):
            graph.launch(1)
            stream.synchronize()
            acts = ((co.state.astype(np.int32) + lane_host) % 5).astype(np.uint8)
            assert np.array_equal(actions.cpu().numpy(), acts), rep
            ref = co.step(acts, auto_reset=True)
            _check(out, ref, 'replay %d' % rep, local=False)
            assert np.array_equal(obs.cpu().numpy(), co.state), rep
            ref_total = ref_total + ref['reward']
        assert np.array_equal(_bits(total.cpu().numpy()), _bits(ref_total)) and env.t == 40
        graph.close()
    env.close()


def test_recording_is_refused_where_it_cannot_work():
    grid = MapfGrid(['....', '....'])
    kw = dict(start_local=np.array([[0, 5]], np.uint16).repeat(64, 0), goal_local=np.array([[7, 2]], np.uint16).repeat(64, 0))
    host = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, **kw)
    with pytest.raises(nat.MapfNativeError) as err:
        host.graph_begin()                           # host-pointer calls wait for the stream: nothing to record
    assert 'MAPF_FLAG_DEVICE_PTRS' in str(err.value)
    host.close()
    dev = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, device_arrays=True, **kw)
    with pytest.raises(nat.MapfNativeError):
        dev.graph_end()                              # not recording
    dev.graph_begin()
    with pytest.raises(nat.MapfNativeError):
        dev.graph_begin()
    with pytest.raises(nat.MapfNativeError):
        dev.set_state(None, t=5)
    g = dev.graph_end()                              # an empty recording is a valid (empty) graph
    assert g.steps == 0
    g.launch(3)
    dev.sync()
    assert dev.t == 0
    g.close()
    dev.close()
    # a stream the CALLER captures (torch.cuda.graph): a step enqueued there would bake its step index into the graph and
    # replay the same random numbers for ever -- refused, loudly
    import torch
    stream = torch.cuda.Stream()
    ext = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, device_arrays=True,
                     stream=stream.cuda_stream, **kw)
    acts = torch.zeros((64, 2), dtype=torch.uint8, device='cuda')
    call, out = ext.prepare_step(acts)
    call()                                           # fine outside a capture
    ext.sync()
    foreign = torch.cuda.CUDAGraph()
    with pytest.raises(nat.MapfNativeError) as err:
        with torch.cuda.graph(foreign, stream=stream):
            call()
    assert 'mapf_graph_begin' in str(err.value)
    ext.graph_begin()                                # the library's own recording on that stream still works
    call()
    own = ext.graph_end()
    own.launch(2)
    ext.sync()
    assert ext.t == 3
    ext.close()                                      # (destroys the recording with the handle)


def _partial_record_io(env, n_steps, rec_reward):
    """mapf_rollout_io that asks for the reward trajectory only: the library substitutes its own stand-ins for the other four."""
    import ctypes
    return nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=n_steps, step_flags=nat.MAPF_STEP_AUTO_RESET,
                             accumulate=0, rec_reward=rec_reward.data_ptr())


def test_stand_in_trajectory_buffers_cannot_move_under_a_recorded_rollout():
    """A recorded rollout that leaves some rec_* arrays out names handle-owned stand-ins.  Growing a stand-in frees and
    reallocates it, and the next replay of the recorded node would write into freed memory: a later rollout that needs
    larger stand-ins is refused (inside the recording and after it) while the graph lives; the replay still matches."""
    import ctypes
    import torch
    E, A = 1024, 8
    grid, _, nbr, start, goal = _c3_tables(E)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.Makespan, seed=5, device_arrays=True,
                     start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.MAKESPAN, seed=5)
    small = torch.zeros((8, E), dtype=torch.float64, device='cuda')
    large = torch.zeros((16, E), dtype=torch.float64, device='cuda')
    io8, io16 = _partial_record_io(env, 8, small), _partial_record_io(env, 16, large)
    env.graph_begin()
    nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io8)))
    with pytest.raises(nat.MapfNativeError) as err:          # same recording: the first node already names the stand-ins
        nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io16)))
    assert 'stand-in' in str(err.value)
    graph = env.graph_end()
    assert graph.steps == 8
    with pytest.raises(nat.MapfNativeError):                 # ... and after it, while the graph is alive
        nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io16)))
    nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io8)))   # the same size is fine (plain launch: steps 0..7)
    env.sync()
    first = small.cpu().numpy().copy()
    graph.launch(1)                                          # steps 8..15 through the recorded node
    env.sync()
    import philox
    ref = np.stack([co.step(philox.random_actions_np(5, np.arange(E), t, A), auto_reset=True)['reward'] for t in range(16)])
    assert np.array_equal(_bits(first), _bits(ref[:8]))
    assert np.array_equal(_bits(small.cpu().numpy()), _bits(ref[8:]))
    assert env.t == 16
    graph.close()
    nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io16)))   # no graph left: the stand-ins may grow again
    env.sync()
    env.close()


def test_own_stream_capture_is_refused_once_the_stream_was_handed_out():
    """A handle that created its own stream cannot be captured by anybody else -- until mapf_get_stream hands the stream
    out (torch ExternalStream interop).  From then on a step enqueued under a foreign capture is refused like on a
    caller-owned stream: it would bake its step index into the caller's graph."""
    import torch
    grid = MapfGrid(['....', '....'])
    kw = dict(start_local=np.array([[0, 5]], np.uint16).repeat(64, 0), goal_local=np.array([[7, 2]], np.uint16).repeat(64, 0))
    env = VecMapfEnv(grid, 2, None, None, 0.2, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, device_arrays=True, **kw)
    acts = torch.zeros((64, 2), dtype=torch.uint8, device='cuda')
    call, out = env.prepare_step(acts)
    call()
    env.sync()
    ext = torch.cuda.ExternalStream(env.stream)
    foreign = torch.cuda.CUDAGraph()
    with pytest.raises(nat.MapfNativeError) as err:
        with torch.cuda.graph(foreign, stream=ext):
            call()
    assert 'mapf_graph_begin' in str(err.value)
    call()                                                   # outside the capture the handle works as before
    env.sync()
    assert env.t == 2
    env.close()


def test_invalidate_state_restores_the_terminal_test_after_a_write_through_the_view():
    """After an auto-reset step the library knows no env can be terminal and runs the step instance without
    is_terminal(prev) (mapf_env.py:238-240).  A caller that writes the state buffer through the view must call
    invalidate_state(): the next step then reports the terminal envs exactly like the reference."""
    import torch
    E, A = 1024, 8
    grid, _, nbr, start, goal = _c3_tables(E)
    env = VecMapfEnv(grid, A, None, None, 0.2, *R, OptimizationCriteria.Makespan, seed=3, device_arrays=True,
                     start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, 0.2, *R, mo.MAKESPAN, seed=3)
    actions = env.fill_random_actions(0, 3)
    env.sync()
    acts = actions.cpu().numpy()
    local, reward, done, info = env.step(actions[0], auto_reset=True)
    env.sync()
    _check(dict(local=local, reward=reward, done=done, **info), co.step(acts[0], auto_reset=True), 'step 0')
    local, reward, done, info = env.step(actions[1], auto_reset=True)
    env.sync()
    _check(dict(local=local, reward=reward, done=done, **info), co.step(acts[1], auto_reset=True), 'step 1')
    assert 'NO_TERMINAL' in env.last_kernel('step'), env.last_kernel('step')
    # every fourth env is put on its goal cells (terminal), behind the library's back
    view = env.state_view()
    goal_t = torch.as_tensor(goal.astype(np.int32), device='cuda').to(torch.uint16) if goal.ndim == 2 else None
    assert goal_t is not None
    view[::4] = goal_t[::4]
    torch.cuda.synchronize()
    co.state[::4] = goal[::4]
    env.invalidate_state()
    local, reward, done, info = env.step(actions[2], auto_reset=True)
    env.sync()
    assert 'NO_TERMINAL' not in env.last_kernel('step'), env.last_kernel('step')
    ref = co.step(acts[2], auto_reset=True)
    assert ref['was_terminal'][::4].all() and not ref['was_terminal'][1::4].any()
    _check(dict(local=local, reward=reward, done=done, **info), ref, 'step 2')
    env.close()
