"""MultiMapVecEnv -- a batch whose envs do NOT share one map.

Every reference env owns its grid (gym_mapf/envs/mapf_env.py:127); a ``VecMapfEnv`` handle steps a batch on ONE map (the
move table is per handle).  This wrapper keeps one ``VecMapfEnv`` per RUN of consecutive envs that share a map and
scatters / gathers the env-major arrays, so that callers see a single batch in their own env order.  A handle numbers its
envs consecutively, which is why runs and not whole groups: env e keeps the global id ``env_id_offset + e`` -- it draws
exactly what it would draw alone, whatever its neighbours' maps.  A batch sorted by map needs one handle per map; an
interleaved one needs more, which costs launches, not correctness.

Host mode only (numpy in, numpy out).  There is no CPU implementation behind it either.
"""
import numpy as np

from gym_mapf_amd.envs.vec_env import VecMapfEnv


class MultiMapVecEnv:
    def __init__(self, grids, n_agents, start_locations, goal_locations, fail_prob, reward_of_collision, reward_of_goal,
                 reward_of_living, optimization_criteria, *, seed=42, env_id_offset=0, device=0):
        """``grids``: one MapfGrid per env (the same object -- or an equal grid -- may repeat); ``start_locations`` /
        ``goal_locations``: per env, A (row, col) pairs.  The other arguments as ``VecMapfEnv`` / the reference."""
        self.n_envs, self.n_agents = len(grids), int(n_agents)
        if len(start_locations) != self.n_envs or len(goal_locations) != self.n_envs:
            raise ValueError('one start / goal row per env')
        distinct = []                                   # [grid]; MapfGrid.__eq__ compares cell contents (reference grid.py:42-46)
        which = np.empty(self.n_envs, np.int64)
        for e, g in enumerate(grids):
            for k, d in enumerate(distinct):
                if g is d or g == d:
                    which[e] = k
                    break
            else:
                which[e] = len(distinct)
                distinct.append(g)
        self.grids = distinct
        self._parts = []                                # (env indices of the run, VecMapfEnv)
        e = 0
        while e < self.n_envs:
            run_end = e + 1
            while run_end < self.n_envs and which[run_end] == which[e]:
                run_end += 1
            idx = np.arange(e, run_end)
            starts = np.asarray([start_locations[i] for i in idx]).reshape(len(idx), self.n_agents, 2)
            goals = np.asarray([goal_locations[i] for i in idx]).reshape(len(idx), self.n_agents, 2)
            env = VecMapfEnv(distinct[which[e]], self.n_agents, starts, goals, fail_prob, reward_of_collision, reward_of_goal,
                             reward_of_living, optimization_criteria, seed=seed, env_id_offset=int(env_id_offset) + e,
                             device=device)
            self._parts.append((idx, env))
            e = run_end

    @property
    def n_handles(self):
        return len(self._parts)

    def reset(self, mask=None):
        for idx, env in self._parts:
            env.reset(None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8)[idx]))

    def step(self, actions, uniforms=None, auto_reset=False):
        """One ``MapfEnv.step()`` per env; arrays as ``VecMapfEnv.step`` (cells are local ids OF EACH ENV'S OWN MAP)."""
        E, A = self.n_envs, self.n_agents
        actions = np.ascontiguousarray(actions, np.uint8).reshape(E, A)
        local, reward = np.empty((E, A), np.uint16), np.empty(E, np.float64)
        done = np.empty(E, np.uint8)
        info = {'prob': np.empty(E, np.float64), 'collision': np.empty(E, np.uint8), 'was_terminal': np.empty(E, np.uint8)}
        for idx, env in self._parts:
            u = None if uniforms is None else np.ascontiguousarray(np.asarray(uniforms, np.float64)[idx])
            l, r, d, i = env.step(np.ascontiguousarray(actions[idx]), uniforms=u, auto_reset=auto_reset)
            local[idx], reward[idx], done[idx] = l, r, d
            for k in info:
                info[k][idx] = i[k]
        return local, reward, done, info

    def rollout(self, n_steps, actions=None, auto_reset=True):
        """``n_steps`` fused steps per env (one launch per handle): ``returns`` / ``episodes`` / ``collisions`` [E]."""
        E = self.n_envs
        res = {'returns': np.empty(E, np.float64), 'episodes': np.empty(E, np.uint32), 'collisions': np.empty(E, np.uint32)}
        for idx, env in self._parts:
            a = None if actions is None else np.ascontiguousarray(np.asarray(actions, np.uint8)[:, idx])
            part = env.rollout(n_steps, actions=a, auto_reset=auto_reset)
            for k in res:
                res[k][idx] = part[k]
        return res

    def get_state(self):
        local = np.empty((self.n_envs, self.n_agents), np.uint16)
        t = 0
        for idx, env in self._parts:
            local[idx], t = env.get_state()
        return local, t

    def close(self):
        for _, env in self._parts:
            env.close()
        self._parts = []
