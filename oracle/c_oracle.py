"""ctypes front-end of oracle/mapf_oracle.c (TEST INFRASTRUCTURE ONLY -- see that file)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_build', 'libmapf_oracle.so')
_lib = None

_P = ctypes.c_void_p


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('%s missing: run `make -C oracle`' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.oracle_step.restype = ctypes.c_int
        lib.oracle_step.argtypes = [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, _P, ctypes.c_int, _P,
                                    ctypes.c_int, _P, _P, _P, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_uint32, ctypes.c_int, _P, _P, _P, _P, _P, _P]
        lib.oracle_greedy_actions.restype = None
        lib.oracle_greedy_actions.argtypes = [_P, ctypes.c_uint32, ctypes.c_uint64, _P, _P, _P, ctypes.c_int, _P]
        lib.oracle_rollout.restype = ctypes.c_uint64
        lib.oracle_rollout.argtypes = [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, _P, ctypes.c_int, _P,
                                       ctypes.c_int, _P, _P, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_uint32, ctypes.c_int, _P, _P, _P]
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


class COracle:
    """E envs on one map; same argument meaning as VecMapfEnv but local ids in, CPU out."""

    def __init__(self, nbr, n_agents, start_local, goal_local, fail_prob, r_clash, r_goal, r_living,
                 criteria, seed=42, env_id_offset=0):
        self.nbr = np.ascontiguousarray(nbr, dtype=np.uint16)
        self.V, self.A = self.nbr.shape[0], int(n_agents)
        self.start = np.ascontiguousarray(start_local, dtype=np.uint16).reshape(-1, self.A)
        self.goal = np.ascontiguousarray(goal_local, dtype=np.uint16).reshape(-1, self.A)
        self.E = max(self.start.shape[0], self.goal.shape[0])
        self.sb, self.gb = int(self.start.shape[0] == 1 and self.E > 1), int(self.goal.shape[0] == 1 and self.E > 1)
        self.args = (float(fail_prob), float(r_clash), float(r_goal), float(r_living), int(criteria))
        self.seed, self.off, self.t = int(seed), int(env_id_offset), 0
        self.state = np.ascontiguousarray(np.broadcast_to(self.start, (self.E, self.A))).copy()

    def reset(self, mask=None):
        src = np.broadcast_to(self.start, (self.E, self.A))
        if mask is None:
            self.state[:] = src
        else:
            m = np.asarray(mask).astype(bool)
            self.state[m] = src[m]

    def step(self, actions, uniforms=None, auto_reset=False):
        E, A = self.E, self.A
        actions = np.ascontiguousarray(actions, dtype=np.uint8).reshape(E, A)
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(E, A)
        out = dict(local=np.empty((E, A), np.uint16), reward=np.empty(E), done=np.empty(E, np.uint8),
                   collision=np.empty(E, np.uint8), prob=np.empty(E), was_terminal=np.empty(E, np.uint8))
        rc = load().oracle_step(_p(self.nbr), self.V, A, E, _p(self.start), self.sb, _p(self.goal), self.gb,
                                _p(self.state), _p(actions), _p(uniforms), self.seed, self.off, self.t,
                                *self.args, int(auto_reset), _p(out['local']), _p(out['reward']), _p(out['done']),
                                _p(out['collision']), _p(out['prob']), _p(out['was_terminal']))
        assert rc == 0
        self.t += 1
        return out

    def greedy_actions(self, cell_rc):
        """MAPF_POLICY_GREEDY actions u8[E, A] for the current states; cell_rc u32[V] = row | col << 16."""
        cell_rc = np.ascontiguousarray(cell_rc, dtype=np.uint32)
        out = np.empty((self.E, self.A), np.uint8)
        load().oracle_greedy_actions(_p(self.nbr), self.A, self.E, _p(cell_rc), _p(self.state), _p(self.goal), self.gb, _p(out))
        return out

    def rollout(self, n_steps, actions=None, auto_reset=True):
        E, A = self.E, self.A
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.uint8).reshape(n_steps, E, A)
        out = dict(returns=np.empty(E), episodes=np.empty(E, np.uint32), collisions=np.empty(E, np.uint32))
        n = load().oracle_rollout(_p(self.nbr), self.V, A, E, _p(self.start), self.sb, _p(self.goal), self.gb,
                                  _p(self.state), _p(actions), int(n_steps), self.seed, self.off, self.t,
                                  *self.args, int(auto_reset), _p(out['returns']), _p(out['episodes']),
                                  _p(out['collisions']))
        assert n == n_steps * E * A
        self.t += n_steps
        out['agent_steps'] = n
        return out
