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


def create_mapf_env(map_name, scen_id, n_agents, fail_prob, reward_of_clash, reward_of_goal,
                    reward_of_living, optimization_criteria):
    """Reference :101-135 (``sanity-R-S`` pseudo maps are a later row: SURVEY.md 8(f)-3)."""
    if map_name.startswith('sanity'):
        raise NotImplementedError('sanity-<rooms>-<size> maps are not built yet (SURVEY.md 8(f)-3)')
    map_file, scen_file = map_name_to_files(map_name, scen_id)
    grid = MapfGrid(parse_map_file(map_file))
    agents_starts, agents_goals = parse_scen_file(scen_file, n_agents)
    return MapfEnv(grid, len(agents_goals), agents_starts, agents_goals, fail_prob,
                   reward_of_clash, reward_of_goal, reward_of_living, optimization_criteria)
