"""VecMapfEnv -- E concurrent MAPF environments stepped by one HIP kernel launch.

This is the batched form of the reference's ``MapfEnv`` (gym_mapf/envs/mapf_env.py:115-266):
same constructor arguments, same transition semantics per env, but state, actions and
results are env-major arrays (``x[e, i]`` = agent i of env e) and the work happens in
``libmapf_hip.so``.  Two memory modes:

* host mode (default): numpy arrays in, numpy arrays out, each call synchronises;
* device mode (``device_arrays=True``): torch CUDA tensors in/out, calls only enqueue on the
  env's HIP stream -- call ``sync()`` before reading results on another stream.

There is no CPU implementation behind this class: without the library or a GPU the
constructor raises ``MapfNativeError``.
"""
import ctypes
import enum

import numpy as np

from gym_mapf_amd import _native as nat


class OptimizationCriteria(enum.Enum):
    """Reference mapf_env.py:31-33."""
    SoC = 'SoC'
    Makespan = 'Makespan'


_CRITERIA_CODE = {OptimizationCriteria.Makespan: nat.MAPF_MAKESPAN, OptimizationCriteria.SoC: nat.MAPF_SOC}


def _locs_to_local(loc_to_int, locs):
    """(row, col) sequence -> local ids; KeyError for obstacles / out-of-map cells (reference :143, :369)."""
    return [loc_to_int[(int(l[0]), int(l[1]))] for l in locs]


class _DeviceView:
    """``__cuda_array_interface__`` carrier: lets torch wrap a device pointer the library owns without copying."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}
        self._owner = owner   # the view must not outlive the env


class StepGraph:
    """A recorded sequence of ``step`` / ``rollout`` calls (``VecMapfEnv.graph_begin`` .. ``graph_end``): ``launch(n)``
    replays it n times.  The step index lives in device memory for recorded launches, so every replay draws fresh
    random numbers -- n replays of an N-step recording equal n * N ``step`` calls with the same arrays."""

    def __init__(self, env, handle, steps):
        self._env, self._g, self.steps = env, handle, steps

    def launch(self, n_replays=1):
        rc = self._env._lib.mapf_graph_launch(self._env._h, self._g, int(n_replays))
        if rc:
            nat.check(rc)

    def close(self):
        g, self._g = self._g, None
        if g and self._env._h:
            nat.check(self._env._lib.mapf_graph_destroy(self._env._h, g))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VecMapfEnv:
    def __init__(self, grid, n_agents, start_locations, goal_locations, fail_prob,
                 reward_of_collision, reward_of_goal, reward_of_living, optimization_criteria,
                 *, n_envs=None, seed=42, env_id_offset=0, device=0, device_arrays=False, stream=None,
                 start_local=None, goal_local=None, kernel='auto'):
        self.grid = grid
        self.n_agents = int(n_agents)
        self.fail_prob = fail_prob
        self.reward_of_clash = reward_of_collision
        self.reward_of_goal = reward_of_goal
        self.reward_of_living = reward_of_living
        self.optimization_criteria = optimization_criteria
        self.seed = int(seed)
        self.env_id_offset = int(env_id_offset)
        self.device = int(device)
        self.device_arrays = bool(device_arrays)
        self.valid_locations, self.loc_to_int, self._nbr = grid.tables()
        self.n_cells = len(self.valid_locations)

        start, start_bcast = self._as_local(start_locations, start_local, 'start')
        goal, goal_bcast = self._as_local(goal_locations, goal_local, 'goal')
        if n_envs is None:
            n_envs = 1
            for arr, bc in ((start, start_bcast), (goal, goal_bcast)):
                if not bc:
                    n_envs = arr.shape[0]
        self.n_envs = int(n_envs)
        for arr, bc, what in ((start, start_bcast, 'start'), (goal, goal_bcast, 'goal')):
            if not bc and arr.shape[0] != self.n_envs:
                raise ValueError('%s_locations has %d envs, expected %d' % (what, arr.shape[0], self.n_envs))
        self.start_local, self.goal_local = start, goal
        self._start_bcast, self._goal_bcast = start_bcast, goal_bcast

        flags = 0
        flags |= nat.MAPF_FLAG_START_BROADCAST if start_bcast else 0
        flags |= nat.MAPF_FLAG_GOAL_BROADCAST if goal_bcast else 0
        flags |= nat.MAPF_FLAG_DEVICE_PTRS if self.device_arrays else 0
        # kernel family: both give identical results; 'auto' lets the library choose
        flags |= {'auto': 0, 'thread_per_env': nat.MAPF_FLAG_THREAD_PER_ENV,
                  'lane_group': nat.MAPF_FLAG_LANE_GROUP}[kernel]
        self.kernel = kernel
        nbr = np.ascontiguousarray(self._nbr, dtype=np.uint16)
        desc = nat.MapfDesc(
            struct_size=ctypes.sizeof(nat.MapfDesc), n_cells=self.n_cells, n_agents=self.n_agents,
            criteria=_CRITERIA_CODE[optimization_criteria], n_envs=self.n_envs,
            env_id_offset=self.env_id_offset, seed=self.seed & 0xFFFFFFFFFFFFFFFF,
            nbr=nbr.ctypes.data, start=start.ctypes.data, goal=goal.ctypes.data,
            fail_prob=float(fail_prob), r_clash=float(reward_of_collision), r_goal=float(reward_of_goal),
            r_living=float(reward_of_living), device=self.device, flags=flags,
            stream=(int(stream) if stream else None))
        self._torch = None
        if self.device_arrays:
            # torch wheels bundle their own HIP runtime: it must be initialised BEFORE libmapf_hip pulls in the
            # system one, otherwise torch later reports "No HIP GPUs are available".
            import torch
            torch.cuda.init()
            self._torch = torch
            self._tdev = torch.device('cuda', self.device)
        self._lib = nat.load()
        handle = ctypes.c_void_p()
        nat.check(self._lib.mapf_create(ctypes.byref(desc), ctypes.byref(handle)))
        self._h = handle
        self._rollout_io = None        # rollout(out=...): the argument block of the last such call, with the arrays it points into

    # ------------------------------------------------------------------ construction
    def _as_local(self, locations, local_ids, what):
        """``locations``: A (row, col) pairs shared by every env, or an [E, A, 2] array of
        per-env locations.  ``local_ids`` (keyword form): integer [A] or [E, A] local ids."""
        A = self.n_agents
        if local_ids is not None:
            ids = np.asarray(local_ids)
            if ids.ndim == 1:
                ids = ids.reshape(1, -1)
            if ids.ndim != 2 or ids.shape[1] != A or not np.issubdtype(ids.dtype, np.integer):
                raise ValueError('%s_local must be an integer array of shape [A] or [E, A]' % what)
            if ids.size and (ids.min() < 0 or ids.max() >= self.n_cells):
                raise KeyError('%s_local: local id out of range' % what)
            return np.ascontiguousarray(ids, dtype=np.uint16), local_ids is not None and np.asarray(local_ids).ndim == 1
        if len(locations) != A and not (np.asarray(locations).ndim == 3):
            raise AssertionError('%r locations number is different than the number of agents %d' % (locations, A))
        arr = np.asarray(locations)
        if arr.ndim == 2 and arr.shape == (A, 2):
            return np.asarray(_locs_to_local(self.loc_to_int, arr), dtype=np.uint16).reshape(1, A), True
        if arr.ndim == 3 and arr.shape[1:] == (A, 2):
            flat = _locs_to_local(self.loc_to_int, arr.reshape(-1, 2))
            return np.ascontiguousarray(np.asarray(flat, dtype=np.uint16).reshape(arr.shape[0], A)), False
        raise ValueError('cannot interpret %s_locations of shape %r' % (what, arr.shape))

    # --------------------------------------------------------------------- plumbing
    def _ptr(self, arr, dtype, shape, name):
        """Raw pointer of a caller-supplied array after checking dtype/shape/contiguity."""
        if arr is None:
            return None
        if self.device_arrays:
            t = self._torch
            want = {np.uint8: t.uint8, np.uint16: t.uint16, np.float64: t.float64, np.uint32: t.uint32, np.uint64: t.uint64}[dtype]
            if not (isinstance(arr, t.Tensor) and arr.is_cuda and arr.dtype == want and arr.is_contiguous()
                    and tuple(arr.shape) == tuple(shape)):
                raise ValueError('%s must be a contiguous CUDA %s tensor of shape %r' % (name, want, tuple(shape)))
            return arr.data_ptr()
        if not (isinstance(arr, np.ndarray) and arr.dtype == dtype and arr.flags.c_contiguous
                and tuple(arr.shape) == tuple(shape)):
            raise ValueError('%s must be a C-contiguous %s array of shape %r' % (name, np.dtype(dtype), tuple(shape)))
        return arr.ctypes.data

    def _empty(self, shape, dtype):
        if self.device_arrays:
            t = self._torch
            td = {np.uint8: t.uint8, np.uint16: t.uint16, np.float64: t.float64, np.uint32: t.uint32, np.uint64: t.uint64}[dtype]
            return t.empty(shape, dtype=td, device=self._tdev)
        return np.empty(shape, dtype=dtype)

    def _coerce(self, arr, dtype, shape, name):
        """Inputs: accept anything array-like in host mode, strict tensors in device mode."""
        if arr is None or self.device_arrays:
            return arr
        out = np.ascontiguousarray(arr, dtype=dtype)
        if tuple(out.shape) != tuple(shape):
            raise ValueError('%s must have shape %r, got %r' % (name, tuple(shape), tuple(out.shape)))
        return out

    # -------------------------------------------------------------------------- API
    def reset(self, mask=None):
        """``MapfEnv.reset()`` for all envs, or those with a non-zero mask byte.  No reseed."""
        mask = self._coerce(mask, np.uint8, (self.n_envs,), 'mask')
        nat.check(self._lib.mapf_reset(self._h, self._ptr(mask, np.uint8, (self.n_envs,), 'mask')))

    def step(self, actions, uniforms=None, auto_reset=False, out=None):
        """One ``MapfEnv.step()`` per env.

        actions: uint8 [E, A] (0 STAY, 1 UP, 2 RIGHT, 3 DOWN, 4 LEFT).  uniforms: float64 [E, A]
        values the reference would draw from ``np_random.rand()`` in agent order, or None for the
        device Philox stream.  Returns ``(local, reward, done, info)`` with ``local`` uint16
        [E, A], ``reward`` float64 [E], ``done`` uint8 [E] and ``info`` holding ``prob``,
        ``collision`` and ``was_terminal`` arrays.  ``out`` may carry preallocated arrays under
        those names (plus ``local``, ``reward``, ``done``).
        """
        E, A = self.n_envs, self.n_agents
        actions = self._coerce(actions, np.uint8, (E, A), 'actions')
        uniforms = self._coerce(uniforms, np.float64, (E, A), 'uniforms')
        out = dict(out) if out else {}
        spec = (('local', np.uint16, (E, A)), ('reward', np.float64, (E,)), ('done', np.uint8, (E,)),
                ('collision', np.uint8, (E,)), ('prob', np.float64, (E,)), ('was_terminal', np.uint8, (E,)))
        for name, dt, shp in spec:
            if name not in out:
                out[name] = self._empty(shp, dt)
        p = {name: self._ptr(out[name], dt, shp, name) for name, dt, shp in spec}
        nat.check(self._lib.mapf_step(
            self._h, self._ptr(actions, np.uint8, (E, A), 'actions'),
            self._ptr(uniforms, np.float64, (E, A), 'uniforms'),
            p['local'], p['reward'], p['done'], p['collision'], p['prob'], p['was_terminal'],
            nat.MAPF_STEP_AUTO_RESET if auto_reset else 0))
        info = {'prob': out['prob'], 'collision': out['collision'], 'was_terminal': out['was_terminal']}
        return out['local'], out['reward'], out['done'], info

    def prepare_step(self, actions, uniforms=None, auto_reset=False, out=None, write_local=True):
        """Validate once, call many times: returns ``(call, out)`` where ``call()`` performs
        ``mapf_step`` on exactly these arrays (the per-call Python cost is one ctypes call).  Meant
        for device mode, where a training loop refills ``actions`` in place every iteration.
        ``write_local=False`` leaves ``out_local`` out of the call: the next observation is then read from
        ``state_view()`` (the handle's own state buffer, after auto-reset) and the step writes the cells once."""
        E, A = self.n_envs, self.n_agents
        actions = self._coerce(actions, np.uint8, (E, A), 'actions')
        uniforms = self._coerce(uniforms, np.float64, (E, A), 'uniforms')
        out = dict(out) if out else {}
        spec = (('local', np.uint16, (E, A)), ('reward', np.float64, (E,)), ('done', np.uint8, (E,)),
                ('collision', np.uint8, (E,)), ('prob', np.float64, (E,)), ('was_terminal', np.uint8, (E,)))
        for name, dt, shp in spec:
            if name not in out and (write_local or name != 'local'):
                out[name] = self._empty(shp, dt)
        args = (self._h, self._ptr(actions, np.uint8, (E, A), 'actions'),
                self._ptr(uniforms, np.float64, (E, A), 'uniforms'),
                self._ptr(out['local'], np.uint16, (E, A), 'local') if write_local else None,
                self._ptr(out['reward'], np.float64, (E,), 'reward'),
                self._ptr(out['done'], np.uint8, (E,), 'done'), self._ptr(out['collision'], np.uint8, (E,), 'collision'),
                self._ptr(out['prob'], np.float64, (E,), 'prob'),
                self._ptr(out['was_terminal'], np.uint8, (E,), 'was_terminal'),
                nat.MAPF_STEP_AUTO_RESET if auto_reset else 0)
        fn, check, keep = self._lib.mapf_step, nat.check, (actions, uniforms, out)

        def call():
            rc = fn(*args)
            if rc:
                check(rc)
            return keep

        return call, out

    def state_view(self):
        """The handle's own state buffer as a uint16 [E, A] CUDA tensor (device mode only; no copy): ``env.s`` of every
        env as per-agent cells -- after an auto-reset step the state the NEXT step starts from (a finished episode
        shows its start cells).  Its contents change with every step / rollout / reset enqueued on the env's stream.
        READ-ONLY by contract (the C API returns a const pointer; torch cannot import a read-only CUDA array, so the
        tensor itself is writable): change the state through ``set_state``, or call ``invalidate_state()`` after writing
        the tensor -- the library otherwise assumes that no env is terminal after an auto-reset step and skips
        ``is_terminal(prev)`` (reference mapf_env.py:238-240) in the next one."""
        if not self.device_arrays:
            raise ValueError('state_view() needs device_arrays=True (use get_state() in host mode)')
        ptr = ctypes.c_void_p()
        nat.check(self._lib.mapf_state_view(self._h, ctypes.byref(ptr)))
        return self._torch.as_tensor(_DeviceView(ptr.value, (self.n_envs, self.n_agents), '<u2', self), device=self._tdev)

    def invalidate_state(self):
        """Tell the library that the state buffer was written behind its back (see ``state_view``)."""
        nat.check(self._lib.mapf_invalidate_state(self._h))

    def graph_begin(self):
        """Start recording ``step`` / ``prepare_step`` calls / ``rollout`` / ``reset`` into a hipGraph (device mode only)."""
        nat.check(self._lib.mapf_graph_begin(self._h))

    def graph_end(self):
        """Finish the recording: returns a ``StepGraph``."""
        g = ctypes.c_void_p()
        nat.check(self._lib.mapf_graph_end(self._h, ctypes.byref(g)))
        steps = ctypes.c_uint64(0)
        nat.check(self._lib.mapf_graph_steps(g, ctypes.byref(steps)))
        return StepGraph(self, g, steps.value)

    # one launch addresses every array with 32-bit byte offsets and counts per-launch events in 16 bits (include/mapf_hip.h)
    _MAX_ARRAY_BYTES = (1 << 32) - 1
    _MAX_LAUNCH_STEPS = 65535

    def rollout(self, n_steps, actions=None, auto_reset=True, record=False, accumulate_into=None, out=None):
        """``n_steps`` fused steps.  ``actions`` uint8 [T, E, A] or None for the on-device policy.  Returns a dict with
        ``returns`` f64 [E], ``episodes`` u32 [E], ``collisions`` u32 [E] and, when ``record``, the per-step
        ``local``/``reward``/``done``/``collision``/``prob`` trajectories (step-major).  One launch when every array
        of the call stays below 4 GiB and T <= 65535 (the C ABI's limits); otherwise the steps are issued as a few
        launches over consecutive slices of the same arrays (totals accumulate, the trajectory is identical).
        ``out``: the dict an earlier call of the same shape returned -- its arrays are written again instead of allocating
        eight new ones per call (totals overwritten, unlike ``accumulate_into``); a training loop that calls
        ``rollout(T=16..64)`` thousands of times wants this (profiles/r05_rollout_T_sweep.txt)."""
        E, A, T = self.n_envs, self.n_agents, int(n_steps)
        if out is not None and accumulate_into is None:
            # the repeat call of a training loop: the same dict of arrays, the same shape.  The argument block of the earlier call
            # is reused once every array of the dict is still the OBJECT it was made from (an array swapped in the dict, another T
            # or another actions buffer take the full path below) -- slicing, checking and packing nine arrays is ~6 us of host
            # time per call, which is what a T <= 16 launch is bound by (profiles/r05_rollout_T_sweep.txt, column b)
            cached = self._rollout_io
            if cached is not None and cached[0] is out and cached[1] == (T, bool(auto_reset), bool(record)):
                arrays, act_ref, io = cached[2], cached[3], cached[4]
                same = all(out.get(k) is v for k, v in arrays)
                if actions is None:
                    same = same and act_ref is None
                elif act_ref is not None and self.device_arrays:
                    same = same and isinstance(actions, self._torch.Tensor) and actions.data_ptr() == act_ref[0] and \
                        tuple(actions.shape) == act_ref[1] and actions.dtype == self._torch.uint8 and actions.is_contiguous() and actions.is_cuda
                else:
                    same = False
                if same:
                    nat.check(self._lib.mapf_rollout(self._h, ctypes.byref(io)))
                    return out
        actions = self._coerce(actions, np.uint8, (T, E, A), 'actions')
        if out is not None and accumulate_into is not None:
            raise ValueError('pass either out= (overwrite) or accumulate_into= (add), not both')
        res = accumulate_into if accumulate_into is not None else (out if out is not None else {})
        for name, dt in (('returns', np.float64), ('episodes', np.uint32), ('collisions', np.uint32)):
            if name not in res:
                res[name] = self._empty((E,), dt)
        if record:
            for name, dt, shp in (('local', np.uint16, (T, E, A)), ('reward', np.float64, (T, E)),
                                  ('done', np.uint8, (T, E)), ('collision', np.uint8, (T, E)),
                                  ('prob', np.float64, (T, E))):
                if out is None or name not in res or tuple(res[name].shape) != shp:
                    res[name] = self._empty(shp, dt)
        per_step = max(E * A * 2, E * 8) if (record or actions is not None) else 0    # bytes of the widest per-step row
        t_max = min(self._MAX_LAUNCH_STEPS, self._MAX_ARRAY_BYTES // per_step if per_step else self._MAX_LAUNCH_STEPS)
        if T > 0 and t_max < 1:
            raise ValueError('one env-step of this batch exceeds 4 GiB: use fewer envs per handle')
        first, accumulate = 0, accumulate_into is not None
        while True:
            n = min(T - first, t_max) if T else 0
            sl = slice(first, first + n)
            part = lambda name: res[name][sl] if record else None
            io = nat.MapfRolloutIO(
                struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=n,
                step_flags=nat.MAPF_STEP_AUTO_RESET if auto_reset else 0, accumulate=1 if accumulate else 0,
                actions=self._ptr(actions[sl] if actions is not None else None, np.uint8, (n, E, A), 'actions'),
                out_returns=self._ptr(res['returns'], np.float64, (E,), 'returns'),
                out_episodes=self._ptr(res['episodes'], np.uint32, (E,), 'episodes'),
                out_collisions=self._ptr(res['collisions'], np.uint32, (E,), 'collisions'),
                rec_local=self._ptr(part('local'), np.uint16, (n, E, A), 'local'),
                rec_reward=self._ptr(part('reward'), np.float64, (n, E), 'reward'),
                rec_done=self._ptr(part('done'), np.uint8, (n, E), 'done'),
                rec_collision=self._ptr(part('collision'), np.uint8, (n, E), 'collision'),
                rec_prob=self._ptr(part('prob'), np.float64, (n, E), 'prob'))
            nat.check(self._lib.mapf_rollout(self._h, ctypes.byref(io)))
            first += n
            if first >= T:
                if out is not None and accumulate_into is None and n == T and self.device_arrays:
                    # (one launch covered the call: its argument block serves the next call with the same arrays; the arrays are
                    # referenced here, so their memory cannot be handed to anyone else while the block is kept)
                    keys = ('returns', 'episodes', 'collisions') + (('local', 'reward', 'done', 'collision', 'prob') if record else ())
                    self._rollout_io = (out, (T, bool(auto_reset), bool(record)), tuple((k, res[k]) for k in keys),
                                        None if actions is None else (actions.data_ptr(), tuple(actions.shape), actions), io)
                return res
            accumulate = True

    def transitions(self, local, actions, max_branches=None, env_index=None, first_branch=0, out=None):
        """``env.P[s][a]`` for N (state, joint action) queries (reference mapf_env.py:448-478): every branch of the
        joint slip distribution in the reference's order.  ``local`` uint16 [N, A], ``actions`` uint8 [N, A],
        ``env_index`` uint32 [N] picks whose goals apply (default env 0).  Returns a dict: ``count`` uint32 [N] (always
        the full number of branches) and, for the window of ``max_branches`` branches (default 3**A) that starts at
        ``first_branch``, ``next`` uint16 [N, M, A], ``prob`` / ``reward`` float64 [N, M], ``done`` / ``collision``
        uint8 [N, M].  Rows whose branch index is >= count[q] are unspecified.  Up to 16 agents."""
        A = self.n_agents
        local = np.asarray(local) if not self.device_arrays else local
        N = int(local.shape[0])
        M = int(max_branches) if max_branches is not None else 3 ** A
        local = self._coerce(local, np.uint16, (N, A), 'local')
        actions = self._coerce(actions, np.uint8, (N, A), 'actions')
        env_index = self._coerce(env_index, np.uint32, (N,), 'env_index')
        res = out if out is not None else {
            'count': self._empty((N,), np.uint32), 'next': self._empty((N, M, A), np.uint16),
            'prob': self._empty((N, M), np.float64), 'reward': self._empty((N, M), np.float64),
            'done': self._empty((N, M), np.uint8), 'collision': self._empty((N, M), np.uint8)}
        nat.check(self._lib.mapf_transitions_window(
            self._h, N, self._ptr(local, np.uint16, (N, A), 'local'), self._ptr(actions, np.uint8, (N, A), 'actions'),
            self._ptr(env_index, np.uint32, (N,), 'env_index'), int(first_branch), M, self._ptr(res['count'], np.uint32, (N,), 'count'),
            self._ptr(res['next'], np.uint16, (N, M, A), 'next'), self._ptr(res['prob'], np.float64, (N, M), 'prob'),
            self._ptr(res['reward'], np.float64, (N, M), 'reward'), self._ptr(res['done'], np.uint8, (N, M), 'done'),
            self._ptr(res['collision'], np.uint8, (N, M), 'collision')))
        return res

    def transitions_compact(self, local, actions, env_index=None, first_branch=0, max_branches=None, capacity=None, out=None):
        """``env.P[s][a]`` for N queries with COMPACTED rows (``mapf_transitions_compact``): the branches of query q are
        rows ``offset[q] .. offset[q + 1] - 1`` of ``next`` uint16 [R, A], ``prob`` / ``reward`` float64 [R], ``done`` /
        ``collision`` uint8 [R], in the reference's order (mapf_env.py:448-478); ``offset`` uint64 [N + 1], ``count``
        uint32 [N] (full branch counts).  ``capacity`` = R, the rows the arrays hold (default: N * min(3**A, max_branches),
        which always suffices); rows beyond it are not written -- check ``offset[N] <= R`` (after ``sync()`` in device
        mode).  ``out`` reuses the arrays of an earlier call."""
        A = self.n_agents
        local = np.asarray(local) if not self.device_arrays else local
        N = int(local.shape[0])
        M = min(int(max_branches), 3 ** A) if max_branches is not None else 3 ** A
        local = self._coerce(local, np.uint16, (N, A), 'local')
        actions = self._coerce(actions, np.uint8, (N, A), 'actions')
        env_index = self._coerce(env_index, np.uint32, (N,), 'env_index')
        if out is not None:
            res, R = out, int(out['prob'].shape[0])
        else:
            R = int(capacity) if capacity is not None else N * M
            res = {'offset': self._empty((N + 1,), np.uint64), 'count': self._empty((N,), np.uint32),
                   'next': self._empty((R, A), np.uint16), 'prob': self._empty((R,), np.float64),
                   'reward': self._empty((R,), np.float64), 'done': self._empty((R,), np.uint8),
                   'collision': self._empty((R,), np.uint8)}
        nat.check(self._lib.mapf_transitions_compact(
            self._h, N, self._ptr(local, np.uint16, (N, A), 'local'), self._ptr(actions, np.uint8, (N, A), 'actions'),
            self._ptr(env_index, np.uint32, (N,), 'env_index'), int(first_branch), max(1, M), R,
            self._ptr(res['offset'], np.uint64, (N + 1,), 'offset'), self._ptr(res['count'], np.uint32, (N,), 'count'),
            self._ptr(res['next'], np.uint16, (R, A), 'next'), self._ptr(res['prob'], np.float64, (R,), 'prob'),
            self._ptr(res['reward'], np.float64, (R,), 'reward'), self._ptr(res['done'], np.uint8, (R,), 'done'),
            self._ptr(res['collision'], np.uint8, (R,), 'collision')))
        return res

    def transition_rewards(self, prev_local, actions, next_local, env_index=None):
        """``calc_transition_reward_from_local_states`` (reference mapf_env.py:225-235) for N given transitions:
        ``prev_local`` / ``next_local`` uint16 [N, A], ``actions`` uint8 [N, A].  Returns ``(reward f64 [N],
        done u8 [N], collision u8 [N])``."""
        A = self.n_agents
        prev_local = np.asarray(prev_local) if not self.device_arrays else prev_local
        N = int(prev_local.shape[0])
        prev_local = self._coerce(prev_local, np.uint16, (N, A), 'prev_local')
        next_local = self._coerce(next_local, np.uint16, (N, A), 'next_local')
        actions = self._coerce(actions, np.uint8, (N, A), 'actions')
        env_index = self._coerce(env_index, np.uint32, (N,), 'env_index')
        reward, done, coll = self._empty((N,), np.float64), self._empty((N,), np.uint8), self._empty((N,), np.uint8)
        nat.check(self._lib.mapf_transition_rewards(
            self._h, N, self._ptr(prev_local, np.uint16, (N, A), 'prev_local'),
            self._ptr(actions, np.uint8, (N, A), 'actions'), self._ptr(next_local, np.uint16, (N, A), 'next_local'),
            self._ptr(env_index, np.uint32, (N,), 'env_index'), self._ptr(reward, np.float64, (N,), 'reward'),
            self._ptr(done, np.uint8, (N,), 'done'), self._ptr(coll, np.uint8, (N,), 'collision')))
        return reward, done, coll

    def fill_random_actions(self, t0, n_steps, out=None):
        """Synthetic policy stream: uint8 [n_steps, E, A] uniform over the 5 actions."""
        shape = (int(n_steps), self.n_envs, self.n_agents)
        if out is None:
            out = self._empty(shape, np.uint8)
        nat.check(self._lib.mapf_fill_random_actions(self._h, self._ptr(out, np.uint8, shape, 'out'),
                                                     int(t0), int(n_steps)))
        return out

    def set_policy(self, policy='random'):
        """On-device policy of ``rollout(actions=None)``: ``'random'`` (default; the uniform-random action stream)
        or ``'greedy'`` -- every agent takes the first action in ACTIONS order whose intended target is closest
        (Manhattan distance) to its goal, i.e. the first unblocked move one step closer, else STAY.  The reference
        has no policy; this stands in for the caller-side ``a = policy(s)`` of the loop around ``step``."""
        if policy == 'random':
            nat.check(self._lib.mapf_set_policy(self._h, nat.MAPF_POLICY_RANDOM, None))
        elif policy == 'greedy':
            valid, _, _ = self.grid.tables()
            rc = np.ascontiguousarray([r | (c << 16) for r, c in valid], dtype=np.uint32)
            nat.check(self._lib.mapf_set_policy(self._h, nat.MAPF_POLICY_GREEDY, rc.ctypes.data))
        else:
            raise ValueError("policy must be 'random' or 'greedy'")
        self.policy = policy

    def query_terminal(self, out=None):
        """``MapfEnv.is_terminal`` of every env's current state: uint8 [E]."""
        if out is None:
            out = self._empty((self.n_envs,), np.uint8)
        nat.check(self._lib.mapf_query_terminal(self._h, self._ptr(out, np.uint8, (self.n_envs,), 'out')))
        return out

    def get_state(self, out=None):
        """(local uint16 [E, A], step index t)."""
        shape = (self.n_envs, self.n_agents)
        if out is None:
            out = self._empty(shape, np.uint16)
        t = ctypes.c_uint64(0)
        nat.check(self._lib.mapf_get_state(self._h, self._ptr(out, np.uint16, shape, 'out'), ctypes.byref(t)))
        return out, t.value

    def set_state(self, local=None, t=None):
        shape = (self.n_envs, self.n_agents)
        local = self._coerce(local, np.uint16, shape, 'local')
        if t is None:
            t = self.t
        nat.check(self._lib.mapf_set_state(self._h, self._ptr(local, np.uint16, shape, 'local'), int(t)))

    @property
    def t(self):
        t = ctypes.c_uint64(0)
        nat.check(self._lib.mapf_get_state(self._h, None, ctypes.byref(t)))
        return t.value

    def sync(self):
        nat.check(self._lib.mapf_sync(self._h))

    def timer_begin(self):
        nat.check(self._lib.mapf_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_double(0.0)
        nat.check(self._lib.mapf_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    def last_kernel(self, which='rollout'):
        """Name of the kernel instance that took this env's last ``step`` / ``rollout`` launch (the library picks
        the lane layout from A, E and the table size); '' before the first launch."""
        code = {'step': nat.MAPF_KERNEL_STEP, 'rollout': nat.MAPF_KERNEL_ROLLOUT, 'transitions': nat.MAPF_KERNEL_TRANSITIONS}[which]
        return self._lib.mapf_last_kernel(self._h, code).decode()

    @property
    def stream(self):
        s = ctypes.c_void_p()
        nat.check(self._lib.mapf_get_stream(self._h, ctypes.byref(s)))
        return s.value

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        self._rollout_io = None
        if h:
            self._lib.mapf_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
