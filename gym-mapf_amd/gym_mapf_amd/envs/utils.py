"""MovingAI map / scenario loaders and the env factory (reference gym_mapf/envs/utils.py)."""
from gym_mapf_amd.envs import map_name_to_files
from gym_mapf_amd.envs.grid import MapfGrid
from gym_mapf_amd.envs.mapf_env import MapfEnv


def parse_map_file(map_file):
    """The map body: every line after the 4 header lines (type/height/width/map); reference :33-37."""
    with open(map_file, 'r') as f:
        return f.readlines()[4:]


def parse_scen_file(scen_file, n_agents):
    """First ``n_agents`` (start, goal) pairs of a MovingAI ``.scen`` file.

    Columns 4..7 (``x_start y_start x_goal y_goal``) are returned as ``(int(x), int(y))`` and are
    used by the env as (row, col) -- the reference's convention (:8-30, pinned by its
    parsers_tests.py:14-15).  A file with fewer rows silently yields fewer agents.
    """
    starts, goals = [], []
    with open(scen_file, 'r') as f:
        next(f)                                     # 'version 1'
        for line in f:
            if len(starts) >= n_agents:
                break
            fields = line.split('\t')
            if len(fields) != 9:
                raise ValueError('malformed scenario row: %r' % line)
            starts.append((int(fields[4]), int(fields[5])))
            goals.append((int(fields[6]), int(fields[7])))
    return tuple(starts), tuple(goals)


def create_sanity_mapf_env(n_rooms, room_size, n_agents, fail_prob, reward_of_clash, reward_of_goal,
                           reward_of_living, optimization_criteria):
    """A row of ``n_rooms`` empty ``room_size`` x ``room_size`` rooms separated by two-column walls with a
    door on the bottom row; room i takes its agents from ``empty-S-S`` scenario ``i mod 25 + 1`` shifted by
    ``i * (S + 2)`` columns (reference utils.py:40-98).  Every room but the last gets ``n_agents // n_rooms``
    agents, the last one the remainder."""
    per_room = int(n_agents / n_rooms)
    last_room = n_agents - per_room * (n_rooms - 1)
    if last_room == 0 or per_room == 0:
        raise ValueError(
            f"asked for a sanity env with {n_rooms} rooms  and {n_agents} agents, There are redundant rooms")
    open_row, wall_gap, door_gap = '.' * room_size, '@@', '..'
    lines = []
    for r in range(room_size):
        gap = door_gap if r == room_size - 1 else wall_gap
        lines.append(gap.join([open_row] * n_rooms))
    starts, goals = (), ()
    for room in range(n_rooms):
        _, scen_file = map_name_to_files(f'empty-{room_size}-{room_size}', room % 25 + 1)
        count = per_room if room != n_rooms - 1 else last_room
        room_starts, room_goals = parse_scen_file(scen_file, count)
        shift = room * (room_size + 2)
        starts += tuple((r, c + shift) for r, c in room_starts)
        goals += tuple((r, c + shift) for r, c in room_goals)
    return MapfEnv(MapfGrid(lines), n_agents, starts, goals, fail_prob, reward_of_clash, reward_of_goal,
                   reward_of_living, optimization_criteria)


def create_mapf_env(map_name, scen_id, n_agents, fail_prob, reward_of_clash, reward_of_goal,
                    reward_of_living, optimization_criteria):
    """Reference :101-135: ``sanity-<rooms>-<size>`` names build a synthetic map, anything else is a MovingAI
    map + scenario under ``MAPS_PATH``; a scenario with fewer rows silently yields fewer agents."""
    if map_name.startswith('sanity'):
        n_rooms, room_size = (int(n) for n in map_name.split('-')[1:])
        return create_sanity_mapf_env(n_rooms, room_size, n_agents, fail_prob, reward_of_clash, reward_of_goal,
                                      reward_of_living, optimization_criteria)
    map_file, scen_file = map_name_to_files(map_name, scen_id)
    grid = MapfGrid(parse_map_file(map_file))
    agents_starts, agents_goals = parse_scen_file(scen_file, n_agents)
    return MapfEnv(grid, len(agents_goals), agents_starts, agents_goals, fail_prob,
                   reward_of_clash, reward_of_goal, reward_of_living, optimization_criteria)


def get_local_view(env: MapfEnv, agent_indexes: list, **kwargs):
    """A new env over a subset of ``env``'s agents (kept in their original order), sharing its grid; the only
    override is ``fail_prob`` (reference utils.py:138-157).  Used by planners that decompose a problem."""
    keep = [i for i in range(env.n_agents) if i in agent_indexes]
    return MapfEnv(env.grid, len(agent_indexes), tuple(env.agents_starts[i] for i in keep),
                   tuple(env.agents_goals[i] for i in keep), kwargs.get('fail_prob', env.fail_prob),
                   env.reward_of_clash, env.reward_of_goal, env.reward_of_living, env.optimization_criteria)


def mapf_env_load_from_json(json_str: str) -> MapfEnv:
    """Unimplemented in the reference as well (utils.py:160-161)."""
    raise NotImplementedError()


def manhattan_distance(env: MapfEnv, s, a1, a2):
    """|d row| + |d col| between agents ``a1`` and ``a2`` in joint state ``s`` (reference utils.py:164-167)."""
    locations = env.state_to_locations(s)
    return abs(locations[a1][0] - locations[a2][0]) + abs(locations[a1][1] - locations[a2][1])
