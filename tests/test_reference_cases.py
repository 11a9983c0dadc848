"""The reference's own unit tests (gym_mapf/tests/*.py, 25 cases) re-expressed against this package.

Host-only cases run anywhere; every case that calls step() is marked gpu because the transition
runs in the HIP kernel.  Expected values are the ones the reference's tests assert (cited per test).
``env.P`` (the planner transition enumeration, reference mapf_env.py:448-483) runs in the
``mapf_transitions`` kernel; its cases below are therefore GPU tests too.
"""
import os
from copy import copy

import pytest

from gym_mapf_amd.envs import (ACTIONS, DOWN, LEFT, MAPS_PATH, RIGHT, STAY, UP, integer_to_vector, vector_to_integer,
                               integer_to_vector_multiple_numbers, vector_to_integer_multiple_numbers)
from gym_mapf_amd.envs.grid import EmptyCell, MapfGrid, ObstacleCell
from gym_mapf_amd.envs.mapf_env import (MapfEnv, OptimizationCriteria, execute_action, integer_action_to_vector,
                                        vector_action_to_integer)
from gym_mapf_amd.envs.utils import create_mapf_env, parse_map_file, parse_scen_file

REWARD_OF_CLASH, REWARD_OF_LIVING, REWARD_OF_GOAL = -1000.0, -1, 100.0


def _map(name):
    return os.path.join(MAPS_PATH, name, name + '.map')


# ------------------------------------------------------------------ mapf_grid_tests.py
def test_empty_8_8_grid():                                  # mapf_grid_tests.py:9-20
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    for loc in ((0, 0), (1, 1), (0, 1), (2, 1), (7, 7)):
        assert grid[loc] is EmptyCell
    with pytest.raises(IndexError):
        grid[8, 1]


def test_berlin_1_256_grid_crlf():                          # mapf_grid_tests.py:22-32
    grid = MapfGrid(parse_map_file(_map('Berlin_1_256')))
    assert grid[0, 0] is EmptyCell and grid[0, 104] is EmptyCell and grid[0, 109] is EmptyCell
    for c in (105, 106, 107, 108):
        assert grid[0, c] is ObstacleCell


def test_grid_iteration_is_column_major_and_rows_index():   # grid.py:27-46 behaviour used by mapf_env.py:142
    grid = MapfGrid(['..@.', '....', '.@..'])
    assert list(grid)[:5] == [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1)]
    assert len(grid) == 3 and len(grid[0]) == 4 and grid[0][2] is ObstacleCell
    assert grid == MapfGrid(['..@.\n', '....\r\n', '.@..'])
    with pytest.raises(KeyError):
        MapfGrid(['..T.'])
    valid, loc_to_int, nbr = grid.tables()
    assert valid == [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (1, 2), (2, 2), (0, 3), (1, 3), (2, 3)]


# ------------------------------------------------------------------ parsers_tests.py
def test_scen_parser_empty_8_8():                           # parsers_tests.py:10-15
    starts, goals = parse_scen_file(os.path.join(MAPS_PATH, 'empty-8-8', 'empty-8-8-even-1.scen'), 4)
    assert starts == ((0, 0), (5, 3), (1, 7), (0, 5))
    assert goals == ((1, 0), (5, 6), (6, 4), (7, 4))


# ------------------------------------------------------------------ action_execution_tests.py
def test_moving_on_empty_grid():                            # action_execution_tests.py:13-23
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    s = ((0, 0), (7, 7))
    assert execute_action(grid, s, (RIGHT, UP)) == ((0, 1), (6, 7))
    assert execute_action(grid, s, (DOWN, LEFT)) == ((1, 0), (7, 6))


def test_against_the_wall():                                # action_execution_tests.py:25-32
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    assert execute_action(grid, ((0, 0), (7, 7)), (LEFT, RIGHT)) == ((0, 0), (7, 7))


def test_against_obstacle_stays_in_place():                 # action_execution_tests.py:34-45
    grid = MapfGrid(['..@..', '..@..', '.....', '..@..', '..@..'])
    assert execute_action(grid, ((0, 1),), (RIGHT,)) == ((0, 1),)


def test_stay_action():                                     # action_execution_tests.py:47-54
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    assert execute_action(grid, ((0, 0), (7, 7)), (STAY, STAY)) == ((0, 0), (7, 7))


# ------------------------------------------------------------------ utils_tests.py
def test_create_mapf_env_start_states():                    # utils_tests.py:14-35
    e1 = create_mapf_env('empty-8-8', 1, 2, 0.2, -1000.0, 100.0, 0.0, OptimizationCriteria.Makespan)
    assert e1.s == e1.locations_to_state(((0, 0), (5, 3)))
    e2 = create_mapf_env('empty-48-48', 16, 2, 0.2, -1000.0, 100.0, 0.0, OptimizationCriteria.Makespan)
    assert e2.s == e2.locations_to_state(((40, 42), (17, 2)))


def test_integer_to_vector():                               # utils_tests.py:37-54
    assert integer_to_vector(10, [4] * 2, 2, lambda n: n) == (2, 2)
    assert integer_to_vector(28, [len(ACTIONS)] * 3, 3, lambda n: ACTIONS[n]) == (DOWN, STAY, UP)
    f = lambda n: (int(n / 3), n % 3)  # noqa: E731
    assert integer_to_vector(10, [12] * 2, 2, f) == ((3, 1), (0, 0))
    assert integer_to_vector(13, [12] * 2, 2, f) == ((0, 1), (0, 1))
    assert integer_to_vector(14, [12] * 2, 2, f) == ((0, 2), (0, 1))
    assert integer_to_vector(23, [12] * 2, 2, f) == ((3, 2), (0, 1))
    assert integer_to_vector(143, [12] * 2, 2, f) == ((3, 2), (3, 2))


def test_vector_to_integer():                               # utils_tests.py:56-72
    assert vector_to_integer((2, 1), [4, 4], lambda n: n) == 6
    assert vector_to_integer((DOWN, STAY, UP), [len(ACTIONS)] * 3, lambda a: ACTIONS.index(a)) == 28
    f = lambda v: 3 * v[0] + v[1]  # noqa: E731
    for vec, val in ((((3, 1), (0, 0)), 10), (((0, 1), (0, 1)), 13), (((0, 2), (0, 1)), 14),
                     (((3, 2), (0, 1)), 23), (((3, 2), (3, 2)), 143)):
        assert vector_to_integer(vec, [12] * 2, f) == val


def test_multiple_option_counts():                          # utils_tests.py:74-78
    assert vector_to_integer_multiple_numbers((0, 2), [2, 3], lambda x: x) == 4
    assert integer_to_vector_multiple_numbers(4, [2, 3], 2, lambda x: x) == (0, 2)


def test_vector_action_to_integer_roundtrip():              # utils_tests.py:80-82
    assert integer_action_to_vector(vector_action_to_integer((DOWN, UP)), 2) == (DOWN, UP)


def test_constructor_errors_match_reference():              # mapf_env.py:143, :366-369
    grid = MapfGrid(['..@..', '.....'])
    with pytest.raises(KeyError):
        MapfEnv(grid, 1, ((0, 2),), ((1, 1),), 0, -1.0, 1.0, -1.0, OptimizationCriteria.Makespan)   # start on '@'
    with pytest.raises(KeyError):
        MapfEnv(grid, 1, ((0, 0),), ((0, 2),), 0, -1.0, 1.0, -1.0, OptimizationCriteria.Makespan)   # goal on '@'
    env = MapfEnv(grid, 2, ((0, 0), (1, 1)), ((1, 0), (0, 1)), 0, -1.0, 1.0, -1.0, OptimizationCriteria.Makespan)
    with pytest.raises(AssertionError):
        env.locations_to_state(((0, 0),))
    assert env.nS == 9 ** 2 and env.nA == 25 and env.seed == 42
    assert env.state_to_locations(env.s) == ((0, 0), (1, 1))


# ------------------------------------------------------------------ mapf_env_tests.py (step() cases: GPU)
@pytest.mark.gpu
def test_copy_mapf_env():                                   # mapf_env_tests.py:92-105
    env = MapfEnv(MapfGrid(['....'] * 5), 1, ((0, 0),), ((4, 0),), 0, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    s1, _, _, _ = env.step(vector_action_to_integer((RIGHT,)))
    twin = copy(env)
    s2, _, _, _ = twin.step(vector_action_to_integer((RIGHT,)))
    assert env.s == s1 and twin.s == s2 and s2 != s1          # the copy owns its state
    assert twin.np_random is env.np_random                    # ... and shares the RNG, as in the reference


@pytest.mark.gpu
def test_action_from_terminal_state_has_no_effect():        # mapf_env_tests.py:107-128
    env = MapfEnv(MapfGrid(['..', '..']), 1, ((0, 0),), ((1, 1),), 0, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    state, reward, done, _ = env.step(vector_action_to_integer((RIGHT,)))
    assert reward == REWARD_OF_LIVING and done is False
    state, reward, done, _ = env.step(vector_action_to_integer((DOWN,)))
    assert reward == REWARD_OF_LIVING + REWARD_OF_GOAL and done is True
    for a in (UP, DOWN):
        s2, r2, d2, info = env.step(vector_action_to_integer((a,)))
        assert s2 == state and d2 is True and r2 == 0 and info == {"prob": 0}


@pytest.mark.gpu
def test_switch_spots_is_a_collision():                     # mapf_env_tests.py:130-143
    env = MapfEnv(MapfGrid(['..']), 2, ((0, 0), (0, 1)), ((0, 1), (0, 0)), 0, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    s, r, done, info = env.step(vector_action_to_integer((RIGHT, LEFT)))
    assert done is True and r == REWARD_OF_LIVING + REWARD_OF_CLASH and info['collision'] is True


def _three_agents(criteria):
    grid = MapfGrid(['....'] * 4)
    return MapfEnv(grid, 3, ((0, 0), (3, 3), (1, 1)), ((0, 1), (1, 3), (1, 2)), 0, REWARD_OF_CLASH, REWARD_OF_GOAL,
                   REWARD_OF_LIVING, criteria), ((0, 1), (1, 3), (1, 2))


@pytest.mark.gpu
def test_reward_multiagent_soc():                           # mapf_env_tests.py:247-279
    env, goals = _three_agents(OptimizationCriteria.SoC)
    s, r, done, _ = env.step(vector_action_to_integer((RIGHT, UP, RIGHT)))
    assert r == -3 and not done
    total = r
    s, r, done, _ = env.step(vector_action_to_integer((STAY, UP, STAY)))
    total += r
    assert s == env.locations_to_state(goals) and done
    assert total == 4 * REWARD_OF_LIVING + REWARD_OF_GOAL


@pytest.mark.gpu
def test_reward_multiagent_soc_stay_actions():              # mapf_env_tests.py:281-303
    env, _ = _three_agents(OptimizationCriteria.SoC)
    _, r, _, _ = env.step(vector_action_to_integer((RIGHT, STAY, STAY)))
    assert r == -3


@pytest.mark.gpu
def test_reward_multiagent_makespan():                      # mapf_env_tests.py:305-330
    env, goals = _three_agents(OptimizationCriteria.Makespan)
    s, r1, done, _ = env.step(vector_action_to_integer((RIGHT, UP, RIGHT)))
    assert not done
    s, r2, done, _ = env.step(vector_action_to_integer((STAY, UP, STAY)))
    assert s == env.locations_to_state(goals) and done
    assert r1 + r2 == 2 * REWARD_OF_LIVING + REWARD_OF_GOAL


@pytest.mark.gpu
@pytest.mark.parametrize('criteria', [OptimizationCriteria.SoC, OptimizationCriteria.Makespan])
def test_reward_single_agent(criteria):                     # mapf_env_tests.py:332-387
    env = MapfEnv(MapfGrid(['....'] * 5), 1, ((0, 0),), ((4, 0),), 0, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, criteria)
    total = 0
    for _ in range(4):
        s, r, done, _ = env.step(vector_action_to_integer((DOWN,)))
        total += r
    assert s == env.locations_to_state(((4, 0),)) and r == REWARD_OF_LIVING + REWARD_OF_GOAL
    assert total == REWARD_OF_GOAL + 4 * REWARD_OF_LIVING


@pytest.mark.gpu
def test_scalar_env_follows_reference_trajectories_with_injected_rng():
    """Scalar MapfEnv.step() over the scripted reference cases: joint-int states, rewards, done and
    info dicts, with env.np_random replaced by the same scripted uniforms the reference consumed."""
    from conftest import load_json

    class Scripted:
        def __init__(self):
            self.values = []

        def rand(self):
            return self.values.pop(0)

    crit = {'Makespan': OptimizationCriteria.Makespan, 'SoC': OptimizationCriteria.SoC}
    for case in load_json('scripted_cases.json'):
        env = MapfEnv(MapfGrid(case['lines']), len(case['starts']), tuple(map(tuple, case['starts'])),
                      tuple(map(tuple, case['goals'])), case['fail_prob'], case['r_clash'], case['r_goal'],
                      case['r_living'], crit[case['criteria']])
        rng = Scripted()
        env.np_random = rng
        for k, st in enumerate(case['steps']):
            if st.get('reset'):
                env.reset()
                continue
            rng.values = list(st['uniforms'])
            joint = vector_action_to_integer(tuple(ACTIONS[a] for a in st['actions']))
            s, r, done, info = env.step(joint)
            tag = '%s step %d' % (case['name'], k)
            assert str(s) == st['s'] and r == st['reward'] and done == st['done'], tag
            assert info['prob'] == st['prob'] and info.get('collision') == st['collision'], tag
            assert len(rng.values) == len(st['uniforms']) - st['draws'], tag     # draws only on non-terminal steps
        env.close()


@pytest.mark.gpu
def test_render_marks_agents_goals_and_clashes(capsys):     # mapf_env.py:295-322
    env = MapfEnv(MapfGrid(['...', '.@.']), 2, ((0, 0), (0, 2)), ((1, 0), (0, 0)), 0, -1.0, 1.0, -1.0,
                  OptimizationCriteria.Makespan)
    env.render()
    out = capsys.readouterr().out
    assert out.replace('\x1b', '') .count('\n') == 2 and '@' in out and '0' in out and '1' in out
    env.step(vector_action_to_integer((RIGHT, LEFT)))          # both move into (0, 1): vertex clash
    env.render()
    assert '*' in capsys.readouterr().out


# ------------------------------------------------------------------ mapf_env_tests.py (env.P cases: GPU)
FAIL_PROB = 0.2


@pytest.mark.gpu
def test_transition_function_empty_grid():                  # mapf_env_tests.py:20-71
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    env = MapfEnv(grid, 2, ((0, 0), (7, 7)), ((0, 2), (5, 7)), FAIL_PROB, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    lts = env.locations_to_state
    first = {((round(p, 2), c), ns, r, d) for ((p, c), ns, r, d) in env.P[env.s][vector_action_to_integer((RIGHT, UP))]}
    assert first == {
        ((0.64, False), lts(((0, 1), (6, 7))), REWARD_OF_LIVING, False), ((0.08, False), lts(((1, 0), (6, 7))), REWARD_OF_LIVING, False),
        ((0.08, False), lts(((0, 0), (6, 7))), REWARD_OF_LIVING, False), ((0.08, False), lts(((0, 1), (7, 7))), REWARD_OF_LIVING, False),
        ((0.08, False), lts(((0, 1), (7, 6))), REWARD_OF_LIVING, False), ((0.01, False), lts(((1, 0), (7, 7))), REWARD_OF_LIVING, False),
        ((0.01, False), lts(((1, 0), (7, 6))), REWARD_OF_LIVING, False), ((0.01, False), lts(((0, 0), (7, 7))), REWARD_OF_LIVING, False),
        ((0.01, False), lts(((0, 0), (7, 6))), REWARD_OF_LIVING, False)}
    wish = lts(((0, 1), (6, 7)))
    second = {((round(p, 2), c), ns, r, d) for ((p, c), ns, r, d) in env.P[wish][vector_action_to_integer((RIGHT, UP))]}
    assert second == {
        ((0.64, False), lts(((0, 2), (5, 7))), REWARD_OF_LIVING + REWARD_OF_GOAL, True),
        ((0.08, False), lts(((1, 1), (5, 7))), REWARD_OF_LIVING, False), ((0.08, False), lts(((0, 1), (5, 7))), REWARD_OF_LIVING, False),
        ((0.08, False), lts(((0, 2), (6, 7))), REWARD_OF_LIVING, False), ((0.08, False), lts(((0, 2), (6, 6))), REWARD_OF_LIVING, False),
        ((0.01, False), lts(((1, 1), (6, 7))), REWARD_OF_LIVING, False), ((0.01, False), lts(((1, 1), (6, 6))), REWARD_OF_LIVING, False),
        ((0.01, False), lts(((0, 1), (6, 7))), REWARD_OF_LIVING, False), ((0.01, False), lts(((0, 1), (6, 6))), REWARD_OF_LIVING, False)}


@pytest.mark.gpu
def test_colliding_agents_state_is_terminal_and_negative_reward():   # mapf_env_tests.py:73-90
    grid = MapfGrid(parse_map_file(_map('empty-8-8')))
    env = MapfEnv(grid, 2, ((0, 0), (0, 2)), ((7, 7), (5, 5)), FAIL_PROB, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    tr = {((round(p, 2), c), ns, r, d) for ((p, c), ns, r, d) in env.P[env.s][vector_action_to_integer((RIGHT, LEFT))]}
    assert ((0.64, True), env.locations_to_state(((0, 1), (0, 1))), REWARD_OF_LIVING + REWARD_OF_CLASH, True) in tr


@pytest.mark.gpu
def test_similar_transitions_probability_summed():          # mapf_env_tests.py:229-236
    env = MapfEnv(MapfGrid(['..', '..']), 1, ((0, 0),), ((1, 1),), 0.1, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    a = vector_action_to_integer((STAY, STAY))
    assert env.P[env.s][a] == [((1, False), env.s, REWARD_OF_LIVING, False)]


@pytest.mark.gpu
def test_transition_lists_are_memoised_like_the_reference():    # mapf_env.py:448 lru_cache, :481-483
    env = MapfEnv(MapfGrid(['..', '..']), 2, ((0, 0), (1, 1)), ((1, 1), (0, 0)), 0.1, REWARD_OF_CLASH, REWARD_OF_GOAL,
                  REWARD_OF_LIVING, OptimizationCriteria.Makespan)
    a = vector_action_to_integer((RIGHT, LEFT))
    first = env.P[env.s][a]
    assert env.P[env.s][a] is first                          # the same list object, not a second enumeration
    assert env._partial_get_transitions(env.s)[a] is first
    assert env._get_transitions(env.s, a) is first
    assert abs(sum(p for ((p, _), _, _, _) in first) - 1.0) < 1e-12


@pytest.mark.gpu
def test_is_terminal_and_single_agent_movements_match_reference():      # mapf_env.py:210-223, :163-184
    import numpy as np
    from conftest import load_json, load_trajectory_set
    t = load_json('host_api_cases.json')['is_terminal']
    env = MapfEnv(MapfGrid(t['lines']), 2, tuple(map(tuple, t['starts'])), tuple(map(tuple, t['goals'])), 0.1,
                  REWARD_OF_CLASH, REWARD_OF_GOAL, REWARD_OF_LIVING, OptimizationCriteria.SoC)
    for case in t['cases']:
        assert env.is_terminal(tuple(map(tuple, case['locs']))) == case['terminal']
    env.close()
    # the reference's own single_agent_movements tables (recorded per map in the trajectory goldens)
    for name in ('tiny_a3_slip03_noreset', 'maze32_a5_slip05', 'tiny_a1_slip1'):
        meta, g = load_trajectory_set(name)
        env = MapfEnv(MapfGrid(meta['lines']), meta['n_agents'], tuple(map(tuple, g['start_loc'][0].tolist())),
                      tuple(map(tuple, g['goal_loc'][0].tolist())), meta['fail_prob'], meta['r_clash'], meta['r_goal'],
                      meta['r_living'], OptimizationCriteria.SoC)
        V = len(env.valid_locations)
        for v in list(range(0, V, max(1, V // 40))) + [V - 1]:
            for a in range(5):
                got = env.single_agent_movements(v, a)
                n = int(g['mv_n'][v, a])
                assert [m[0] for m in got] == [v] * n and [m[1] for m in got] == g['mv_next'][v, a, :n].tolist()
                assert np.array_equal(np.asarray([m[2] for m in got]).view(np.uint64), g['mv_prob'][v, a, :n].view(np.uint64))
        env.close()
