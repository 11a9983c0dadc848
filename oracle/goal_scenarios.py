"""Scenario family in which episodes END ON GOALS (test infrastructure, like everything under oracle/).

Random walks on the benchmark maps end almost every episode by collision, so the goal-reached branch of
calc_transition_reward_from_local_states (reference mapf_env.py:225-235: ``reward_of_goal + living``, done, no
collision) and "vertex collision while every agent is on its goal" (collision wins, :228-230) would never be
exercised at 4+ agents.  Here every agent starts ONE move away from its goal on an open map and is driven towards
it; with slip 0 the episode ends at its first step, with slip > 0 after a few.

Used by tests/golden/make_golden.py (the reference steps these scenarios; the outputs are committed fixtures) and
by the GPU parity tests (the same family at batch sizes that reach every kernel layout, against the C oracle).
"""
import numpy as np

_DELTA = ((-1, 0), (0, 1), (1, 0), (0, -1))   # UP, RIGHT, DOWN, LEFT as (d row, d col): actions 1..4


def lattice_side(n_agents):
    m = 1
    while m * m < n_agents:
        m += 1
    return m


def open_map(n_agents):
    """Obstacle-free square map with room for a 3-spaced lattice of >= n_agents goal cells, one cell off the border."""
    size = 3 * lattice_side(n_agents) + 1
    return ['.' * size for _ in range(size)]


def goal_scenario(n_agents, n_envs, seed, first_env=0):
    """(lines, start_loc int32[E, A, 2], goal_loc int32[E, A, 2]) for envs first_env .. first_env + E - 1.

    Env j (global index): goals = a seeded random assignment of the agents to lattice cells; every agent starts on
    one of its goal's four neighbours (seeded direction).  Every third env (j % 3 == 2) gives agent b the SAME goal
    cell as agent a, starting on the opposite side: driven towards their goals both arrive together -- a vertex
    collision in a state where every agent sits on its goal."""
    A = n_agents
    lines = open_map(A)
    m = lattice_side(A)
    start = np.zeros((n_envs, A, 2), np.int32)
    goal = np.zeros((n_envs, A, 2), np.int32)
    for k in range(n_envs):
        j = first_env + k
        rs = np.random.RandomState([seed, j & 0xFFFFFFFF, j >> 32])
        cells = rs.permutation(m * m)[:A]
        dirs = rs.randint(0, 4, size=A)
        for i in range(A):
            g = (1 + 3 * (cells[i] // m), 1 + 3 * (cells[i] % m))
            d = _DELTA[dirs[i]]
            goal[k, i] = g
            start[k, i] = (g[0] + d[0], g[1] + d[1])
        if A >= 2 and j % 3 == 2:
            a = (j // 3) % A
            b = (a + 1 + (j // (3 * A)) % (A - 1)) % A
            goal[k, b] = goal[k, a]
            start[k, b] = 2 * goal[k, a] - start[k, a]        # the opposite neighbour
    return lines, start, goal


def towards_goal_action(loc, goal, n_rows, n_cols):
    """First action in ACTIONS order (STAY, UP, RIGHT, DOWN, LEFT) whose intended target is closest (Manhattan) to
    the goal on an OPEN map -- the same rule as the library's greedy policy (include/mapf_hip.h MAPF_POLICY_GREEDY),
    written for (row, col) locations so that the golden generator can apply it to the reference's own state."""
    best, best_d = 0, abs(loc[0] - goal[0]) + abs(loc[1] - goal[1])
    for a, (dr, dc) in enumerate(_DELTA, start=1):
        r = min(max(loc[0] + dr, 0), n_rows - 1)
        c = min(max(loc[1] + dc, 0), n_cols - 1)
        d = abs(r - goal[0]) + abs(c - goal[1])
        if d < best_d:
            best, best_d = a, d
    return best
