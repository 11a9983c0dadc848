#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED
reference (read-only at /root/reference) in the build container.

Run from the repo root:   python tests/golden/make_golden.py

The reference needs two third-party modules that are not installed here
(``gym==0.13.0`` -- requirements.txt:7 -- and ``colorama``).  Before importing
it this script registers in-memory stand-ins for exactly the names the
reference touches (gym_mapf/envs/mapf_env.py:7-11): ``gym.Env``,
``gym.spaces.Discrete``, ``gym.envs.toy_text.discrete.categorical_sample``
(published algorithm: ``(cumsum(p) > rand()).argmax()``),
``gym.utils.seeding.np_random`` and ``colorama.Fore``.  Nothing of the
reference (source, bytecode) is written anywhere; only inputs and the outputs
it computed are stored.

Randomness: after construction ``env.np_random`` is swapped for an object
whose ``rand()`` returns the build's Philox uniform for (seed, env_id, step t,
agent = call index within the step) -- see oracle/philox.py.  ``step()`` itself
is the reference's code, untouched.
"""
import hashlib
import io
import json
import os
import struct
import sys
import types
import unittest

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REFERENCE = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import philox  # noqa: E402
import goal_scenarios  # noqa: E402


# ----------------------------------------------------------------- stand-ins
def _install_standins():
    gym = types.ModuleType('gym')

    class Env(object):
        pass

    gym.Env = Env
    spaces = types.ModuleType('gym.spaces')

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    spaces.Discrete = Discrete
    envs = types.ModuleType('gym.envs')
    toy = types.ModuleType('gym.envs.toy_text')
    discrete = types.ModuleType('gym.envs.toy_text.discrete')

    def categorical_sample(prob_n, np_random):
        prob_n = np.asarray(prob_n)
        csprob_n = np.cumsum(prob_n)
        return (csprob_n > np_random.rand()).argmax()

    discrete.categorical_sample = categorical_sample
    utils = types.ModuleType('gym.utils')
    seeding = types.ModuleType('gym.utils.seeding')

    def np_random(seed=None):
        # gym 0.13 seeding: sha512(str(seed))[:8] (+4 zero bytes) -> uint32 list
        digest = hashlib.sha512(str(seed).encode('utf8')).digest()[:8] + b'\0' * 4
        words = struct.unpack('3I', digest)
        big = sum(w << (32 * i) for i, w in enumerate(words))
        ints = []
        while big > 0:
            big, mod = divmod(big, 2 ** 32)
            ints.append(mod)
        rng = np.random.RandomState()
        rng.seed(ints)
        return rng, seed

    seeding.np_random = np_random
    gym.spaces, gym.envs, gym.utils = spaces, envs, utils
    envs.toy_text, toy.discrete, utils.seeding = toy, discrete, seeding
    colorama = types.ModuleType('colorama')
    colorama.Fore = types.SimpleNamespace(RED='', GREEN='', YELLOW='', BLUE='', RESET='')
    for name, mod in (('gym', gym), ('gym.spaces', spaces), ('gym.envs', envs),
                      ('gym.envs.toy_text', toy), ('gym.envs.toy_text.discrete', discrete),
                      ('gym.utils', utils), ('gym.utils.seeding', seeding), ('colorama', colorama)):
        sys.modules[name] = mod


_install_standins()
sys.path.insert(0, REFERENCE)
from gym_mapf.envs import ACTIONS  # noqa: E402
from gym_mapf.envs.grid import MapfGrid  # noqa: E402
from gym_mapf.envs.mapf_env import MapfEnv, OptimizationCriteria, vector_action_to_integer  # noqa: E402
from gym_mapf.envs.utils import create_mapf_env, parse_map_file, parse_scen_file  # noqa: E402
from gym_mapf.envs import map_name_to_files  # noqa: E402

CRITERIA = {'Makespan': OptimizationCriteria.Makespan, 'SoC': OptimizationCriteria.SoC}
R_CLASH, R_GOAL, R_LIVING = -1000.0, 100.0, -1.0
SEED = 42


class InjectedUniforms(object):
    """Replacement for env.np_random: rand() -> Philox uniform (or a scripted list)."""

    def __init__(self, seed, env_id):
        self.seed, self.env_id, self.t, self.calls = seed, env_id, 0, 0
        self.script = None

    def begin(self, t, script=None):
        self.t, self.calls, self.script = t, 0, script

    def rand(self):
        k = self.calls
        self.calls += 1
        if self.script is not None:
            return self.script[k]
        return philox.slip_uniform(self.seed, self.env_id, self.t, k)


def synth_random_map(size=64, p_obst=0.20, seed=20):
    """SURVEY.md 8(d) C5 stand-in for the unshipped random-64-64-20 map."""
    rs = np.random.RandomState(seed)
    obst = rs.rand(size, size) < p_obst
    return [''.join('@' if obst[r, c] else '.' for c in range(size)) for r in range(size)]


def synth_starts_goals(lines, n_agents, env_id, seed=SEED):
    free = [(r, c) for c in range(len(lines[0])) for r in range(len(lines)) if lines[r][c] == '.']
    rs = np.random.RandomState([seed, env_id & 0xFFFFFFFF, env_id >> 32])
    s = rs.choice(len(free), size=n_agents, replace=False)
    g = rs.choice(len(free), size=n_agents, replace=False)
    return tuple(free[i] for i in s), tuple(free[i] for i in g)


def movement_tables(env):
    V = len(env.valid_locations)
    n = np.zeros((V, 5), np.uint8)
    nxt = np.zeros((V, 5, 3), np.uint16)
    prob = np.zeros((V, 5, 3), np.float64)
    for v in range(V):
        for a in range(5):
            mv = env.single_agent_movements(v, a)
            n[v, a] = len(mv)
            for k, (_, ns, p) in enumerate(mv):
                nxt[v, a, k], prob[v, a, k] = ns, p
    return n, nxt, prob


def run_trajectories(name, lines, per_env_locs, n_agents, fail_prob, criteria, env_ids, T,
                     auto_reset, map_name=None, n_uniform_steps=32, action_fn=None):
    """Lock-step run of one reference env per env id; returns dict of arrays.  ``action_fn(env, env_id, t)`` ->
    per-agent action indices, or None for the uniform-random policy stream."""
    E, A = len(env_ids), n_agents
    grid = MapfGrid(lines)
    out = dict(
        actions=np.zeros((T, E, A), np.uint8), next_local=np.zeros((T, E, A), np.uint16),
        reward=np.zeros((T, E), np.float64), prob=np.zeros((T, E), np.float64),
        done=np.zeros((T, E), np.uint8), collision=np.zeros((T, E), np.uint8),
        was_terminal=np.zeros((T, E), np.uint8),
        uniforms=np.zeros((min(T, n_uniform_steps), E, A), np.float64),
        env_ids=np.asarray(env_ids, np.uint64),
        start_loc=np.zeros((E, A, 2), np.int32), goal_loc=np.zeros((E, A, 2), np.int32),
        start_local=np.zeros((E, A), np.uint16), goal_local=np.zeros((E, A), np.uint16))
    s_dec = []
    first = None
    for j, env_id in enumerate(env_ids):
        starts, goals = per_env_locs(j, env_id)
        env = MapfEnv(grid, A, starts, goals, fail_prob, R_CLASH, R_GOAL, R_LIVING, CRITERIA[criteria])
        rng = InjectedUniforms(SEED, int(env_id))
        env.np_random = rng
        if first is None:
            first = env
        out['start_loc'][j], out['goal_loc'][j] = starts, goals
        out['start_local'][j] = [env.loc_to_int[l] for l in starts]
        out['goal_local'][j] = [env.loc_to_int[l] for l in goals]
        acts = np.stack([philox.random_actions_np(SEED, [env_id], t, A)[0] for t in range(T)])
        env_s = []
        for t in range(T):
            rng.begin(t)
            if action_fn is not None:
                picked = action_fn(env, int(env_id), t)
                if picked is not None:
                    acts[t] = picked
            a_vec = tuple(ACTIONS[k] for k in acts[t])
            s, r, done, info = env.step(vector_action_to_integer(a_vec))
            out['actions'][t, j] = acts[t]
            out['next_local'][t, j] = [env.loc_to_int[l] for l in env.state_to_locations(s)]
            out['reward'][t, j], out['prob'][t, j] = r, info['prob']
            out['done'][t, j] = done
            out['collision'][t, j] = bool(info.get('collision', False))
            out['was_terminal'][t, j] = 'collision' not in info
            if t < out['uniforms'].shape[0]:
                out['uniforms'][t, j] = [philox.slip_uniform(SEED, int(env_id), t, k) for k in range(A)]
                assert out['was_terminal'][t, j] or rng.calls == A
            if t < 4:
                env_s.append(str(s))
            if done and auto_reset:
                env.reset()
        s_dec.append(env_s)
    mv_n, mv_next, mv_prob = movement_tables(first)
    out.update(valid_locations=np.asarray(first.valid_locations, np.int32),
               mv_n=mv_n, mv_next=mv_next, mv_prob=mv_prob)
    meta = dict(name=name, map_name=map_name, lines=[l.strip() for l in lines], n_agents=A,
                fail_prob=fail_prob, criteria=criteria, r_clash=R_CLASH, r_goal=R_GOAL,
                r_living=R_LIVING, seed=SEED, T=T, auto_reset=auto_reset, V=len(first.valid_locations),
                nS=str(first.nS), nA=str(first.nA), joint_state_first_steps=s_dec)
    if action_fn is not None:   # (key absent = the uniform-random policy stream, as in the round-1 sets)
        meta['actions_from'] = 'towards_goal, every fifth step (t % 5 == 4) the policy stream'  # see main()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    with open(os.path.join(HERE, name + '.json'), 'w') as f:
        json.dump(meta, f, indent=0)
    n_done = int(out['done'].sum())
    print('%-28s E=%3d T=%4d done=%5d collisions=%5d goals=%4d terminal-steps=%5d' % (
        name, E, T, n_done, int(out['collision'].sum()),
        int(((out['done'] == 1) & (out['collision'] == 0) & (out['was_terminal'] == 0)).sum()),
        int(out['was_terminal'].sum())))


def ref_map_lines(map_name):
    return parse_map_file(map_name_to_files(map_name, 1)[0])


def scen_locs(map_name, scen_ids, n_agents):
    def f(j, env_id):
        scen = scen_ids[int(env_id) % len(scen_ids)]
        return parse_scen_file(map_name_to_files(map_name, scen)[1], n_agents)
    return f


# ---------------------------------------------------------------- scripted cases
def scripted_cases():
    """Edge cases with hand-picked uniforms (SURVEY.md 8(c) list)."""
    cases = []
    one_m = 1.0 - 2.0 ** -53

    def run(name, lines, starts, goals, fail_prob, criteria, steps, rewards=(R_CLASH, R_GOAL, R_LIVING)):
        env = MapfEnv(MapfGrid(lines), len(starts), tuple(starts), tuple(goals), fail_prob,
                      rewards[0], rewards[1], rewards[2], CRITERIA[criteria])
        rng = InjectedUniforms(0, 0)
        env.np_random = rng
        rec = []
        for t, (acts, us) in enumerate(steps):
            if acts == 'reset':
                env.reset()
                rec.append(dict(reset=True))
                continue
            rng.begin(t, script=list(us))
            s, r, done, info = env.step(vector_action_to_integer(tuple(acts)))
            rec.append(dict(actions=[ACTIONS.index(a) for a in acts], uniforms=list(us),
                            next_local=[env.loc_to_int[l] for l in env.state_to_locations(s)],
                            next_loc=[list(l) for l in env.state_to_locations(s)],
                            s=str(s), reward=float(r), done=bool(done),
                            collision=(bool(info['collision']) if 'collision' in info else None),
                            prob=float(info['prob']), draws=rng.calls))
        cases.append(dict(name=name, lines=lines, starts=[list(x) for x in starts],
                          goals=[list(x) for x in goals], fail_prob=fail_prob, criteria=criteria,
                          r_clash=rewards[0], r_goal=rewards[1], r_living=rewards[2], steps=rec))

    U, R, D, L, S = 'UP', 'RIGHT', 'DOWN', 'LEFT', 'STAY'
    corridor = ['@.@', '@.@', '@.@', '...']
    # right==left!=main row: cumsum [0.8999999999999999, 0.9999999999999999]; u = 1-2^-53 -> argmax of all-False = 0
    run('corridor_allfalse_argmax0', corridor, [(2, 1)], [(0, 1)], 0.1, 'Makespan',
        [((U,), (one_m,)), ((U,), (0.95,)), ((U,), (0.8999999999999999,)), ((D,), (0.0,))])
    run('corridor_slip02', corridor, [(2, 1)], [(0, 1)], 0.2, 'Makespan',
        [((U,), (0.85,)), ((U,), (0.79,)), ((U,), (0.8,)), ((U,), (one_m,))])
    # border clamp + obstacle bump + all five merge patterns on a small map, slip 0.1 and 0.2
    small = ['..@.', '....', '.@..']
    for fp in (0.1, 0.2):
        run('small_patterns_fp%s' % fp, small, [(0, 0), (2, 3)], [(2, 2), (0, 3)], fp, 'Makespan',
            [((U, D), (0.1, 0.1)), ((L, R), (0.91, 0.96)), ((R, U), (0.85, 0.94)),
             ((D, L), (0.96, 0.5)), ((S, S), (0.99, 0.2)), ((R, U), (0.2, 0.951)),
             ((R, L), (0.9, 0.9)), ((D, D), (0.94999, 0.95))])
    # vertex clash -> terminal -> further steps are no-ops (reward 0, prob 0, no collision key)
    run('vertex_then_terminal', ['...'], [(0, 0), (0, 2)], [(0, 2), (0, 0)], 0.0, 'Makespan',
        [((R, L), (0.3, 0.4)), ((L, R), (0.1, 0.2)), ((S, S), (0.5, 0.5)), ('reset', ()), ((S, S), (0.5, 0.5))])
    # swap: done=True but the state is not terminal -> next step moves again; collision wins over goal
    run('swap_not_sticky_collision_beats_goal', ['..'], [(0, 0), (0, 1)], [(0, 1), (0, 0)], 0.0, 'Makespan',
        [((R, L), (0.3, 0.4)), ((S, S), (0.1, 0.2)), ((L, R), (0.1, 0.2))])
    run('swap_then_move', ['....'], [(0, 1), (0, 2)], [(0, 3), (0, 0)], 0.0, 'SoC',
        [((R, L), (0.3, 0.4)), ((R, L), (0.1, 0.2)), ((R, L), (0.1, 0.2)), ((S, S), (0.0, 0.0))])
    # SoC living reward: on goal + STAY is free, STAY off goal pays, moving off goal pays
    four = ['....', '....', '....', '....']
    run('soc_stay_rules', four, [(0, 0), (3, 3), (1, 1)], [(0, 1), (1, 3), (1, 2)], 0.0, 'SoC',
        [((R, S, S), (0.5, 0.5, 0.5)), ((S, U, R), (0.5, 0.5, 0.5)), ((S, U, S), (0.5, 0.5, 0.5)),
         ((S, S, S), (0.5, 0.5, 0.5))], rewards=(-1000.0, 100.0, -1))
    run('soc_float_living', four, [(0, 0), (3, 3), (1, 1)], [(0, 1), (1, 3), (1, 2)], 0.3, 'SoC',
        [((R, S, S), (0.5, 0.86, 0.99)), ((S, U, R), (0.5, 0.5, 0.5)), ((S, U, S), (0.84, 0.85, 0.86)),
         ((L, D, S), (0.1, 0.9, 0.99))], rewards=(-33.25, 7.125, -0.1))
    # makespan goal reached; terminal afterwards
    run('goal_then_terminal', ['..', '..'], [(0, 0)], [(1, 1)], 0.0, 'Makespan',
        [((R,), (0.5,)), ((D,), (0.5,)), ((U,), (0.5,)), ((D,), (0.5,))], rewards=(-1000.0, 100.0, -1))
    # exotic fail probabilities: 1.0 drops the intended move (p0 == 0), 0.5, >1, <0
    for fp in (1.0, 0.5, 1.5, -0.5, 0.7):
        run('failprob_%s' % fp, small, [(1, 1), (0, 3)], [(2, 2), (2, 0)], fp, 'SoC',
            [((U, D), (0.1, 0.6)), ((L, R), (0.49, 0.51)), ((R, U), (0.75, 0.25)),
             ((D, L), (0.999, 0.001)), ((S, S), (0.5, 0.5)), ((U, U), (0.3, 0.7))])
    with open(os.path.join(HERE, 'scripted_cases.json'), 'w') as f:
        json.dump(cases, f, indent=0)
    print('scripted cases: %d' % len(cases))


def transition_tables():
    """env.P[s][a] enumerations (mapf_env.py:448-478) as known-answer tables."""
    tabs = []

    def dump(name, env, states_locs, actions_list):
        rows = []
        for locs in states_locs:
            s = env.locations_to_state(tuple(locs))
            for acts in actions_list:
                a = vector_action_to_integer(tuple(acts))
                tr = [dict(prob=float(p), collision=bool(c),
                           next_local=[env.loc_to_int[l] for l in env.state_to_locations(ns)],
                           s=str(ns), reward=float(r), done=bool(d))
                      for ((p, c), ns, r, d) in env.P[s][a]]
                rows.append(dict(local=[env.loc_to_int[tuple(l)] for l in locs],
                                 actions=[ACTIONS.index(x) for x in acts], transitions=tr))
        tabs.append(dict(name=name, lines=[''.join('.' if env.grid[r][c].__name__ == 'EmptyCell' else '@'
                                                   for c in range(len(env.grid[0]))) for r in range(len(env.grid))],
                         starts=[list(l) for l in env.agents_starts], goals=[list(l) for l in env.agents_goals],
                         fail_prob=env.fail_prob, criteria=env.optimization_criteria.value,
                         r_clash=env.reward_of_clash, r_goal=env.reward_of_goal,
                         r_living=env.reward_of_living, rows=rows))

    U, R, D, L, S = 'UP', 'RIGHT', 'DOWN', 'LEFT', 'STAY'
    e88 = MapfGrid(ref_map_lines('empty-8-8'))
    env = MapfEnv(e88, 2, ((0, 0), (7, 7)), ((0, 2), (5, 7)), 0.2, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
    dump('empty88_right_up', env, [((0, 0), (7, 7)), ((0, 1), (6, 7))], [(R, U), (S, S), (L, D)])
    env = MapfEnv(e88, 2, ((0, 0), (0, 2)), ((7, 7), (5, 5)), 0.2, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
    dump('empty88_clash', env, [((0, 0), (0, 2)), ((0, 1), (0, 1))], [(R, L), (U, U)])
    env = MapfEnv(MapfGrid(['..', '..']), 1, ((0, 0),), ((1, 1),), 0.1, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
    dump('two_by_two_merge', env, [((0, 0),), ((1, 0),), ((1, 1),)], [(S,), (U,), (R,), (D,), (L,)])
    env = MapfEnv(MapfGrid(['..@.', '....', '.@..']), 3, ((0, 0), (2, 3), (1, 1)), ((2, 2), (0, 3), (1, 2)),
                  0.2, R_CLASH, R_GOAL, -1.0, CRITERIA['SoC'])
    dump('small_soc_three', env, [((0, 0), (2, 3), (1, 1)), ((1, 0), (1, 2), (1, 1)), ((2, 2), (0, 3), (1, 1))],
         [(R, U, S), (D, L, R), (S, S, S)])
    with open(os.path.join(HERE, 'transition_tables.json'), 'w') as f:
        json.dump(tabs, f, indent=0)
    print('transition tables: %d envs' % len(tabs))


def host_api_cases():
    """Planner-side helpers of the reference: sanity maps, local views, predecessors,
    render / render_with_policy output -- recorded as plain data."""
    import contextlib
    from gym_mapf.envs.utils import create_sanity_mapf_env, get_local_view, manhattan_distance
    out = {}
    san = []
    for (rooms, size, agents) in ((2, 8, 4), (3, 8, 7), (1, 16, 3), (4, 8, 4), (2, 32, 5)):
        env = create_mapf_env('sanity-%d-%d' % (rooms, size), None, agents, 0.2, R_CLASH, R_GOAL, R_LIVING, CRITERIA['SoC'])
        san.append(dict(name='sanity-%d-%d' % (rooms, size), n_agents=agents,
                        lines=[''.join('.' if c.__name__ == 'EmptyCell' else '@' for c in env.grid[r]) for r in range(len(env.grid))],
                        starts=[list(l) for l in env.agents_starts], goals=[list(l) for l in env.agents_goals],
                        s=str(env.s), nS=str(env.nS)))
    out['sanity'] = san
    bad = []
    for (rooms, size, agents) in ((3, 8, 2), (5, 8, 4)):
        try:
            create_sanity_mapf_env(rooms, size, agents, 0.1, R_CLASH, R_GOAL, R_LIVING, CRITERIA['SoC'])
            bad.append(None)
        except ValueError as e:
            bad.append(str(e))
    out['sanity_errors'] = bad
    env = create_mapf_env('room-32-32-4', 12, 6, 0.2, R_CLASH, R_GOAL, R_LIVING, CRITERIA['Makespan'])
    views = []
    for idx in ([0, 2], [5], [1, 3, 4], [4, 1]):
        v = get_local_view(env, idx)
        v2 = get_local_view(env, idx, fail_prob=0.35)
        views.append(dict(agents=idx, starts=[list(l) for l in v.agents_starts], goals=[list(l) for l in v.agents_goals],
                          n_agents=v.n_agents, fail_prob=v.fail_prob, fail_prob_override=v2.fail_prob, s=str(v.s),
                          same_grid=v.grid is env.grid))
    out['local_view'] = dict(map='room-32-32-4', scen=12, n_agents=6, views=views,
                             manhattan=[manhattan_distance(env, env.s, a, b) for a, b in ((0, 1), (2, 5), (3, 3))])
    preds = []
    for lines, starts, goals in ((['....', '....', '....'], ((1, 2), (2, 1)), ((0, 0), (2, 3))),
                                 (['..@.', '....', '.@..'], ((0, 0), (1, 2)), ((2, 2), (0, 3))),
                                 (['.@.', '...', '.@.'], ((1, 1),), ((0, 0),)),
                                 (['...', '.@.', '...'], ((0, 0), (2, 2), (0, 2)), ((2, 2), (0, 0), (2, 0)))):
        e = MapfEnv(MapfGrid(lines), len(starts), starts, goals, 0, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
        states = [e.s, e.locations_to_state(goals)]
        preds.append(dict(lines=lines, starts=[list(l) for l in starts], goals=[list(l) for l in goals],
                          queries=[dict(s=str(q), predecessors=sorted(str(x) for x in e.predecessors(q))) for q in states]))
    out['predecessors'] = preds
    rend = []
    e = MapfEnv(MapfGrid(['....', '....', '....']), 2, ((0, 0), (1, 2)), ((2, 2), (0, 0)), 0, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
    for joint in (None, ('RIGHT', 'LEFT'), ('STAY', 'LEFT'), ('LEFT', 'LEFT')):
        if joint is not None:
            e.step(vector_action_to_integer(joint))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            e.render()
        pol = io.StringIO()
        with contextlib.redirect_stdout(pol):
            e.render_with_policy(0, lambda st: (st * 7 + 3) % e.nA)
        rend.append(dict(after=list(joint) if joint else None, s=str(e.s), render=buf.getvalue(), render_with_policy=pol.getvalue()))
    out['render'] = dict(lines=['....', '....', '....'], starts=[[0, 0], [1, 2]], goals=[[2, 2], [0, 0]], frames=rend)
    # on a map with obstacles the reference's render_with_policy raises KeyError (it encodes obstacle cells too)
    eo = MapfEnv(MapfGrid(['..@.', '....', '.@..']), 2, ((0, 0), (1, 2)), ((2, 2), (0, 0)), 0, R_CLASH, R_GOAL, -1, CRITERIA['Makespan'])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        eo.render()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            eo.render_with_policy(0, lambda st: 0)
        err = None
    except KeyError as ex:
        err = repr(ex.args[0])
    out['render_obstacles'] = dict(lines=['..@.', '....', '.@..'], starts=[[0, 0], [1, 2]], goals=[[2, 2], [0, 0]],
                                   render=buf.getvalue(), render_with_policy_keyerror=err)
    gpa = []
    for fp in (0.2, 0.1, 0.0):
        e = MapfEnv(MapfGrid(['....', '....']), 3, ((0, 0), (0, 1), (1, 3)), ((1, 0), (1, 1), (0, 3)), fp, R_CLASH, R_GOAL, -1,
                    CRITERIA['SoC'])
        for joint in (('UP',), ('STAY', 'LEFT'), ('RIGHT', 'DOWN', 'STAY')):
            gpa.append(dict(fail_prob=fp, action=list(joint),
                            result=[[float(p), list(a)] for p, a in e.get_possible_actions(joint)]))
    out['get_possible_actions'] = gpa
    term = []
    e = MapfEnv(MapfGrid(['..@.', '....', '.@..']), 2, ((0, 0), (1, 2)), ((2, 2), (0, 0)), 0.1, R_CLASH, R_GOAL, -1, CRITERIA['SoC'])
    for locs in (((0, 0), (1, 2)), ((2, 2), (0, 0)), ((1, 1), (1, 1)), ((2, 2), (1, 0)), ((0, 3), (0, 3))):
        term.append(dict(locs=[list(l) for l in locs], terminal=bool(e.is_terminal(locs))))
    out['is_terminal'] = dict(lines=['..@.', '....', '.@..'], starts=[[0, 0], [1, 2]], goals=[[2, 2], [0, 0]], cases=term)
    # hot-path helper methods planners call directly (mapf_env.py:225-235, :378-389, :436-446), on local cell ids
    helper = []
    lines3 = ['..@.', '....', '.@..']
    for crit, rewards in (('SoC', (-1000.0, 100.0, -1.0)), ('Makespan', (-33.25, 7.125, -0.1)), ('SoC', (-5, 3, -2))):
        e = MapfEnv(MapfGrid(lines3), 3, ((0, 0), (1, 2), (2, 3)), ((2, 2), (0, 0), (1, 3)), 0.2,
                    rewards[0], rewards[1], rewards[2], CRITERIA[crit])
        l2i = e.loc_to_int
        trips = []
        for prev, acts, nxt in ((((0, 0), (1, 2), (2, 3)), ('DOWN', 'LEFT', 'UP'), ((1, 0), (1, 1), (1, 3))),
                                (((2, 2), (0, 0), (1, 3)), ('STAY', 'STAY', 'STAY'), ((2, 2), (0, 0), (1, 3))),
                                (((2, 2), (0, 1), (1, 3)), ('STAY', 'LEFT', 'STAY'), ((2, 2), (0, 0), (1, 3))),
                                (((1, 1), (1, 2), (2, 3)), ('RIGHT', 'LEFT', 'STAY'), ((1, 2), (1, 1), (2, 3))),
                                (((1, 0), (1, 2), (2, 3)), ('RIGHT', 'LEFT', 'UP'), ((1, 1), (1, 1), (1, 3))),
                                (((2, 2), (0, 0), (1, 3)), ('STAY', 'RIGHT', 'STAY'), ((2, 2), (0, 1), (1, 3))),
                                (((2, 3), (0, 0), (2, 2)), ('LEFT', 'STAY', 'RIGHT'), ((2, 2), (0, 0), (2, 3)))):
            pl, nl = tuple(l2i[x] for x in prev), tuple(l2i[x] for x in nxt)
            a = vector_action_to_integer(acts)
            r, d, c = e.calc_transition_reward_from_local_states(pl, a, nl)
            trips.append(dict(prev_local=list(pl), action=a, next_local=list(nl), reward=float(r), done=bool(d), collision=bool(c),
                              living=float(e._living_reward(pl, a)),
                              is_collision=bool(e._is_collision_transition_from_local_states(pl, nl))))
        helper.append(dict(lines=lines3, starts=[[0, 0], [1, 2], [2, 3]], goals=[[2, 2], [0, 0], [1, 3]], criteria=crit,
                           rewards=list(rewards), fail_prob=0.2, cases=trips))
    out['transition_reward_helpers'] = helper
    # the single-location movers (mapf_env.py:43-75)
    import gym_mapf.envs.mapf_env as ref_env
    g3 = MapfGrid(lines3)
    movers = []
    for name in ('execute_up', 'execute_down', 'execute_right', 'execute_left', 'execute_stay'):
        fn = getattr(ref_env, name)
        movers.append(dict(name=name, results=[[list(loc), list(fn(loc, g3))]
                                               for loc in ((0, 0), (0, 1), (1, 1), (1, 2), (2, 0), (0, 3), (2, 3), (2, 2), (1, 3))]))
    out['single_location_movers'] = dict(lines=lines3, movers=movers)
    with open(os.path.join(HERE, 'host_api_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('host api cases: %d sanity, %d views, %d predecessor envs, %d render frames' % (len(san), len(views), len(preds), len(rend)))


def reference_selftest():
    """Run the reference's own 25 unit tests under the stand-ins."""
    suite = unittest.defaultTestLoader.discover(os.path.join(REFERENCE, 'gym_mapf', 'tests'),
                                                pattern='*_tests.py', top_level_dir=REFERENCE)
    res = unittest.TextTestRunner(stream=io.StringIO(), verbosity=0).run(suite)
    print('reference unit tests: ran=%d failures=%d errors=%d' % (res.testsRun, len(res.failures), len(res.errors)))
    return dict(ran=res.testsRun, failures=len(res.failures), errors=len(res.errors))


def main():
    info = dict(reference_tests=reference_selftest())
    rng42, _ = sys.modules['gym.utils.seeding'].np_random(42)
    info['mt19937_seed42_first_draws_UNPINNED'] = [rng42.rand() for _ in range(4)]

    scripted_cases()
    transition_tables()
    host_api_cases()

    # C1: empty-8-8, scen 1, 2 agents, slip 0, one env (reference CPU path config)
    run_trajectories('c1_empty8_a2_slip0', ref_map_lines('empty-8-8'),
                     scen_locs('empty-8-8', [1], 2), 2, 0.0, 'Makespan', [0], 1500, True, 'empty-8-8')
    # C2: empty-16-16, 4 agents, slip 0.1, scen id 1 + e mod 25
    run_trajectories('c2_empty16_a4_slip01', ref_map_lines('empty-16-16'),
                     scen_locs('empty-16-16', list(range(1, 26)), 4), 4, 0.1, 'Makespan',
                     list(range(25)) + [4095], 400, True, 'empty-16-16')
    # C3: room-32-32-4, 8 agents, slip 0.2, valid scen ids, both criteria; ids beyond 32 bits too
    c3_ids = list(range(18)) + [65535, 262143, (1 << 32) + 5, (1 << 40) + 123456789]
    for crit in ('Makespan', 'SoC'):
        run_trajectories('c3_room32_a8_slip02_%s' % crit.lower(), ref_map_lines('room-32-32-4'),
                         scen_locs('room-32-32-4', [6, 12, 13, 23, 24, 25], 8), 8, 0.2, crit,
                         c3_ids, 600, True, 'room-32-32-4')
    run_trajectories('c3_room32_a8_slip02_noreset', ref_map_lines('room-32-32-4'),
                     scen_locs('room-32-32-4', [6, 12, 13, 23, 24, 25], 8), 8, 0.2, 'SoC',
                     list(range(12)), 96, False, 'room-32-32-4')
    # 64x64 / 32 agents on reference data
    run_trajectories('c64_room64_a32_slip02', ref_map_lines('room-64-64-16'),
                     scen_locs('room-64-64-16', [1, 2, 5, 7], 32), 32, 0.2, 'Makespan',
                     list(range(8)), 160, True, 'room-64-64-16')
    # C5: synthetic random-64-64-20 stand-in, seeded random distinct starts/goals
    c5 = synth_random_map()
    run_trajectories('c5_random64_a32_slip02', c5,
                     lambda j, env_id: synth_starts_goals(c5, 32, int(env_id)), 32, 0.2, 'SoC',
                     list(range(8)), 160, True, None)
    # tiny maps where random walks do reach goals and swap; odd agent counts
    tiny = ['....', '.@..', '....']
    run_trajectories('tiny_a2_slip02_goals', tiny,
                     lambda j, env_id: synth_starts_goals(tiny, 2, int(env_id)), 2, 0.2, 'SoC',
                     list(range(16)), 800, True, None)
    run_trajectories('tiny_a3_slip03_noreset', tiny,
                     lambda j, env_id: synth_starts_goals(tiny, 3, int(env_id)), 3, 0.3, 'Makespan',
                     list(range(16)), 64, False, None)
    run_trajectories('tiny_a1_slip1', tiny,
                     lambda j, env_id: synth_starts_goals(tiny, 1, int(env_id)), 1, 1.0, 'Makespan',
                     list(range(8)), 400, True, None)
    maze = ref_map_lines('maze-32-32-4')
    run_trajectories('maze32_a5_slip05', maze, scen_locs('maze-32-32-4', [10, 12, 16, 24], 5), 5, 0.5, 'SoC',
                     list(range(8)), 300, True, 'maze-32-32-4')
    # episodes that END ON GOALS at 4..32 agents (oracle/goal_scenarios.py): starts one move from the goals on an
    # open map, every agent driven towards its goal (every fifth step: the random policy stream instead); every
    # third env lets two agents share a goal cell -> vertex collision with every agent on its goal
    for A, fp, crit, auto, T in ((4, 0.2, 'Makespan', True, 60), (8, 0.2, 'SoC', True, 60), (8, 0.0, 'Makespan', False, 6),
                                 (16, 0.2, 'Makespan', True, 60), (32, 0.2, 'SoC', True, 80), (32, 0.0, 'Makespan', True, 12)):
        n_env = 12
        lines, st, gl = goal_scenarios.goal_scenario(A, n_env, 7000 + A)

        def towards(env, env_id, t, gl=gl, lines=lines):
            if t % 5 == 4:
                return None
            locs = env.state_to_locations(env.s)
            return [goal_scenarios.towards_goal_action(locs[i], tuple(gl[env_id, i]), len(lines), len(lines[0]))
                    for i in range(len(locs))]
        run_trajectories('goals_a%d_slip%s_%s%s' % (A, str(fp).replace('.', ''), crit.lower(), '' if auto else '_noreset'),
                         lines, lambda j, env_id, st=st, gl=gl: (tuple(map(tuple, st[env_id].tolist())), tuple(map(tuple, gl[env_id].tolist()))),
                         A, fp, crit, list(range(n_env)), T, auto, None, action_fn=towards)

    # the reference's LARGE maps (the only scenario that constructs at 32 agents on maze-128-128-10 is
    # scen 18; Berlin_1_256 -- the map the reference's own grid test opens, mapf_grid_tests.py:22-32 -- constructs at
    # 4 agents for scen 11 only and at 2 agents for eight scen ids): 14818 and 47540 free cells, i.e. move tables far
    # beyond the LDS budget -- the global-table kernels
    run_trajectories('maze128_a32_slip02', ref_map_lines('maze-128-128-10'), scen_locs('maze-128-128-10', [18], 32), 32, 0.2,
                     'Makespan', list(range(6)), 96, True, 'maze-128-128-10')
    run_trajectories('berlin256_a4_slip02', ref_map_lines('Berlin_1_256'), scen_locs('Berlin_1_256', [11], 4), 4, 0.2,
                     'SoC', list(range(6)), 96, True, 'Berlin_1_256')
    run_trajectories('berlin256_a2_slip01', ref_map_lines('Berlin_1_256'),
                     scen_locs('Berlin_1_256', [2, 4, 8, 11, 14, 18, 22, 24], 2), 2, 0.1, 'Makespan', list(range(8)), 64, True,
                     'Berlin_1_256')

    with open(os.path.join(HERE, 'generation_info.json'), 'w') as f:
        json.dump(info, f, indent=1)


if __name__ == '__main__':
    main()
