"""MultiMapVecEnv -- a batch whose envs do NOT share one map.

Every reference env owns its grid (gym_mapf/envs/mapf_env.py:127); a ``VecMapfEnv`` handle steps a batch on ONE map (the
move table is per handle).  This wrapper keeps one ``VecMapfEnv`` per RUN of consecutive envs that share a map and
scatters / gathers the env-major arrays, so that callers see a single batch in their own env order.  A handle numbers its
envs consecutively, which is why runs and not whole groups: env e keeps the global id ``env_id_offset + e`` -- it draws
exactly what it would draw alone, whatever its neighbours' maps.  A batch sorted by map needs one handle per map; an
interleaved one needs more, which costs launches, not correctness.

Two modes, as ``VecMapfEnv``:
  * host mode (default): numpy in, numpy out, one blocking call per run;
  * ``device_arrays=True``: torch CUDA tensors in and out, every run's launch enqueued on ONE stream (the caller's, or the
    one the first handle creates), nothing waits.  A run is a contiguous range of envs, so its slice of an env-major
    batch tensor is itself contiguous: where the slice is 16-byte aligned (the C ABI's rule for device pointers -- a run
    that starts at a multiple of 16 envs) the run's handle reads and writes the caller's tensors IN PLACE; other runs go
    through per-run staging tensors and device-to-device copies on the same stream.

``UnionMapVecEnv`` (below) is the ONE-LAUNCH form of the same thing: the batch's distinct maps become one move table -- their
disjoint union, a block-diagonal neighbour table: no move leads from one map into another -- behind a single handle, so a
step / rollout of the whole mixed batch is one kernel launch whatever the order of the maps (at most 65536 free cells in total).

There is no CPU implementation behind either.
"""
import numpy as np

from gym_mapf_amd.envs.vec_env import VecMapfEnv

_STEP_SPEC = (('local', np.uint16, True), ('reward', np.float64, False), ('done', np.uint8, False),
              ('collision', np.uint8, False), ('prob', np.float64, False), ('was_terminal', np.uint8, False))


class MultiMapVecEnv:
    def __init__(self, grids, n_agents, start_locations, goal_locations, fail_prob, reward_of_collision, reward_of_goal,
                 reward_of_living, optimization_criteria, *, seed=42, env_id_offset=0, device=0, device_arrays=False,
                 stream=None):
        """``grids``: one MapfGrid per env (the same object -- or an equal grid -- may repeat); ``start_locations`` /
        ``goal_locations``: per env, A (row, col) pairs.  The other arguments as ``VecMapfEnv`` / the reference."""
        self.n_envs, self.n_agents = len(grids), int(n_agents)
        self.device_arrays = bool(device_arrays)
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
        self._stream = stream
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
                             device=device, device_arrays=self.device_arrays, stream=self._stream)
            if self.device_arrays and self._stream is None:
                self._stream = env.stream               # every later handle enqueues on the first one's stream
            self._parts.append((idx, env))
            e = run_end
        self._torch = self._parts[0][1]._torch if self.device_arrays and self._parts else None

    @property
    def n_handles(self):
        return len(self._parts)

    @property
    def stream(self):
        """The HIP stream every run enqueues on (device mode), as an integer handle."""
        return self._stream

    # ------------------------------------------------------------------ device-mode plumbing
    def _torch_stream(self):
        return self._torch.cuda.stream(self._torch.cuda.ExternalStream(self._stream))

    @staticmethod
    def _aligned(tensor):
        return tensor.data_ptr() % 16 == 0

    def _check(self, tensor, dtype, shape, name):
        t = self._torch
        want = {np.uint8: t.uint8, np.uint16: t.uint16, np.float64: t.float64, np.uint32: t.uint32}[dtype]
        if not (isinstance(tensor, t.Tensor) and tensor.is_cuda and tensor.dtype == want and tensor.is_contiguous()
                and tuple(tensor.shape) == tuple(shape)):
            raise ValueError('%s must be a contiguous CUDA %s tensor of shape %r' % (name, want, tuple(shape)))
        return tensor

    def prepare_step(self, actions, uniforms=None, auto_reset=False, out=None, write_local=True):
        """Device mode: validate once, call many times (``VecMapfEnv.prepare_step`` for the whole batch).  Returns
        ``(call, out)``: ``call()`` enqueues one ``mapf_step`` per run -- on the caller's tensors in place where a run's
        slice is 16-byte aligned, through staging copies where it is not -- all on ``self.stream``; nothing is waited for."""
        if not self.device_arrays:
            raise ValueError('prepare_step() needs device_arrays=True')
        E, A, t = self.n_envs, self.n_agents, self._torch
        self._check(actions, np.uint8, (E, A), 'actions')
        if uniforms is not None:
            self._check(uniforms, np.float64, (E, A), 'uniforms')
        out = dict(out) if out else {}
        any_env = self._parts[0][1]
        ctx = self._torch_stream
        # Everything this method allocates -- missing outputs, staging copies -- is only ever used on self.stream, so it is
        # allocated UNDER that stream: the caching allocator then orders a later reuse of the blocks behind the work queued
        # there (allocated on the caller's current stream they could be handed out again while self.stream still writes them).
        # The caller's own tensors (actions, uniforms, a supplied `out`) must be produced and consumed on self.stream, or be
        # synchronised with it (`torch.cuda.stream(env.torch_stream)` / `sync()`): the calls below only enqueue.
        with ctx():
            for name, dt, per_agent in _STEP_SPEC:
                if name == 'local' and not write_local:
                    continue
                shape = (E, A) if per_agent else (E,)
                if name not in out:
                    out[name] = any_env._empty(shape, dt)
                self._check(out[name], dt, shape, name)
        calls, copies_in, copies_out, staged_runs = [], [], [], 0
        for idx, env in self._parts:
            lo, hi = int(idx[0]), int(idx[-1]) + 1
            n_before = len(copies_in) + len(copies_out)

            def view(batch, inputs):
                part = batch[lo:hi]
                if self._aligned(part):
                    return part                                      # in place
                with ctx():
                    stage = t.empty_like(part)
                (copies_in if inputs else copies_out).append((stage, part))
                return stage
            a = view(actions, True)
            u = view(uniforms, True) if uniforms is not None else None
            o = {name: view(out[name], False) for name, _, _ in _STEP_SPEC if name in out}
            call, _ = env.prepare_step(a, uniforms=u, auto_reset=auto_reset, out=o, write_local=write_local)
            calls.append(call)
            staged_runs += (len(copies_in) + len(copies_out)) > n_before

        def call_all():
            if copies_in or copies_out:
                with ctx():
                    for stage, part in copies_in:
                        stage.copy_(part, non_blocking=True)
                    for c in calls:
                        c()
                    for stage, part in copies_out:
                        part.copy_(stage, non_blocking=True)
            else:
                for c in calls:
                    c()
            return out

        call_all.in_place_runs = len(self._parts) - staged_runs       # runs that read and write the caller's tensors directly
        return call_all, out

    # -------------------------------------------------------------------------- API
    def reset(self, mask=None):
        for idx, env in self._parts:
            if mask is None:
                env.reset(None)
            elif self.device_arrays:
                with self._torch_stream():
                    env.reset(mask[int(idx[0]):int(idx[-1]) + 1].clone())   # (an aligned, owned copy of the run's mask bytes)
            else:
                env.reset(np.ascontiguousarray(np.asarray(mask, np.uint8)[idx]))

    def step(self, actions, uniforms=None, auto_reset=False, out=None):
        """One ``MapfEnv.step()`` per env; arrays as ``VecMapfEnv.step`` (cells are local ids OF EACH ENV'S OWN MAP)."""
        E, A = self.n_envs, self.n_agents
        if self.device_arrays:
            call, out = self.prepare_step(actions, uniforms=uniforms, auto_reset=auto_reset, out=out)
            call()
            return out['local'], out['reward'], out['done'], {k: out[k] for k in ('prob', 'collision', 'was_terminal')}
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
        """``n_steps`` fused steps per env (one launch per handle): ``returns`` / ``episodes`` / ``collisions`` [E].
        Device mode: ``actions`` is a CUDA uint8 [T, E, A] tensor (a run's [T, run, A] part is strided, so it is copied
        once per call) and the results are CUDA tensors; everything is enqueued on ``self.stream``."""
        E = self.n_envs
        if self.device_arrays:
            t = self._torch
            parts = []
            with self._torch_stream():
                for idx, env in self._parts:
                    lo, hi = int(idx[0]), int(idx[-1]) + 1
                    a = None if actions is None else actions[:, lo:hi].contiguous()
                    parts.append(env.rollout(n_steps, actions=a, auto_reset=auto_reset))
                return {k: t.cat([p[k] for p in parts]) for k in ('returns', 'episodes', 'collisions')}
        res = {'returns': np.empty(E, np.float64), 'episodes': np.empty(E, np.uint32), 'collisions': np.empty(E, np.uint32)}
        for idx, env in self._parts:
            a = None if actions is None else np.ascontiguousarray(np.asarray(actions, np.uint8)[:, idx])
            part = env.rollout(n_steps, actions=a, auto_reset=auto_reset)
            for k in res:
                res[k][idx] = part[k]
        return res

    def get_state(self):
        """(cells [E, A] of each env's own map, step index).  Device mode: a CUDA tensor gathered from the runs' state views
        on ``self.stream`` (the runs' buffers are separate allocations: this is a copy)."""
        if self.device_arrays:
            with self._torch_stream():
                local = self._torch.cat([env.state_view() for _, env in self._parts])
            return local, self._parts[-1][1].t
        local = np.empty((self.n_envs, self.n_agents), np.uint16)
        t = 0
        for idx, env in self._parts:
            local[idx], t = env.get_state()
        return local, t

    def sync(self):
        for _, env in self._parts:
            env.sync()

    def close(self):
        for _, env in reversed(self._parts):            # (the first handle owns the shared stream: it goes last)
            env.close()
        self._parts = []


class UnionGrid:
    """The disjoint union of several ``MapfGrid`` s as ONE table: the free cells of grid k get the ids ``base[k] .. base[k] +
    V_k - 1`` (each grid's own column-major order, reference mapf_env.py:142), ``nbr`` is block-diagonal.  Only what
    ``VecMapfEnv`` asks of a grid: ``tables()``."""

    def __init__(self, grids):
        self.grids = list(grids)
        self.base, valid, rows = [], [], []
        for g in self.grids:
            v, _, nbr = g.tables()
            self.base.append(len(valid))
            rows.append(nbr.astype(np.int64) + len(valid))
            valid.extend(v)
        if len(valid) > 65536:
            raise ValueError('the maps of one batch have %d free cells together: local ids are uint16 (at most 65536)' % len(valid))
        self._tables = (valid, None, np.concatenate(rows).astype(np.uint16))

    def tables(self):
        return self._tables


class UnionMapVecEnv:
    """A batch whose envs live on different maps, stepped by ONE launch (``MultiMapVecEnv``: one per run of same-map envs).

    Every reference env owns its grid (gym_mapf/envs/mapf_env.py:127).  Here the distinct grids of the batch are laid side by
    side in one move table (``UnionGrid``) behind a single ``VecMapfEnv`` handle; an env's cells are its own map's local ids
    plus the map's base, which this wrapper adds on the way in (``set_state``) and takes off on the way out -- callers see each
    env's cells in ITS map's numbering, exactly as from ``MultiMapVecEnv`` or from one ``VecMapfEnv`` per env.  Env e keeps
    the global id ``env_id_offset + e``, so it draws what it would draw alone.  Same constructor arguments as
    ``MultiMapVecEnv``; host mode (numpy) or ``device_arrays=True`` (torch CUDA tensors, everything on the handle's stream)."""

    def __init__(self, grids, n_agents, start_locations, goal_locations, fail_prob, reward_of_collision, reward_of_goal,
                 reward_of_living, optimization_criteria, *, seed=42, env_id_offset=0, device=0, device_arrays=False,
                 stream=None):
        self.n_envs, self.n_agents = len(grids), int(n_agents)
        self.device_arrays = bool(device_arrays)
        if len(start_locations) != self.n_envs or len(goal_locations) != self.n_envs:
            raise ValueError('one start / goal row per env')
        distinct, which = [], np.empty(self.n_envs, np.int64)
        for e, g in enumerate(grids):
            for k, d in enumerate(distinct):
                if g is d or g == d:
                    which[e] = k
                    break
            else:
                which[e] = len(distinct)
                distinct.append(g)
        self.grids = distinct
        self.union = UnionGrid(distinct)
        base = np.asarray(self.union.base, np.int64)[which]
        l2i = [g.tables()[1] for g in distinct]
        A = self.n_agents

        def to_union(locations, what):
            out = np.empty((self.n_envs, A), np.int64)
            for e in range(self.n_envs):
                locs = locations[e]
                if len(locs) != A:
                    raise AssertionError('%r locations number is different than the number of agents %d' % (locs, A))
                out[e] = [l2i[which[e]][(int(l[0]), int(l[1]))] for l in locs]      # KeyError: obstacle / out-of-map cell
            return (out + base[:, None]).astype(np.uint16)
        self._base_host = base.astype(np.uint16).reshape(-1, 1)
        self._env = VecMapfEnv(self.union, A, None, None, fail_prob, reward_of_collision, reward_of_goal, reward_of_living,
                               optimization_criteria, seed=seed, env_id_offset=env_id_offset, device=device,
                               device_arrays=self.device_arrays, stream=stream,
                               start_local=to_union(start_locations, 'start'), goal_local=to_union(goal_locations, 'goal'))
        self._torch = self._env._torch
        self._base = self._base_host
        if self.device_arrays:
            t = self._torch
            self._base = t.from_numpy(self._base_host.view(np.int16)).to(self._env._tdev)     # (int16 bits: torch has no uint16 arithmetic)

    n_handles = 1

    @property
    def stream(self):
        return self._env.stream

    def _on_stream(self):
        t = self._torch
        return t.cuda.stream(t.cuda.ExternalStream(self._env.stream))

    def _own(self, local):
        """union ids -> each env's own map's ids (any leading step axis)"""
        if self.device_arrays:
            t = self._torch
            with self._on_stream():
                return (local.view(t.int16) - self._base).view(t.uint16)        # (mod 2^16: the same bits as unsigned subtraction)
        return (local - self._base_host).astype(np.uint16)

    def _union(self, local):
        if self.device_arrays:
            t = self._torch
            with self._on_stream():
                return (local.view(t.int16) + self._base).view(t.uint16)
        return (np.asarray(local, np.uint16) + self._base_host).astype(np.uint16)

    # -------------------------------------------------------------------------- API (as VecMapfEnv / MultiMapVecEnv)
    def reset(self, mask=None):
        self._env.reset(mask)

    def step(self, actions, uniforms=None, auto_reset=False):
        """One ``MapfEnv.step()`` per env in ONE launch; cells are local ids OF EACH ENV'S OWN MAP."""
        local, reward, done, info = self._env.step(actions, uniforms=uniforms, auto_reset=auto_reset)
        return self._own(local), reward, done, info

    def rollout(self, n_steps, actions=None, auto_reset=True, record=False):
        res = self._env.rollout(n_steps, actions=actions, auto_reset=auto_reset, record=record)
        if record:
            res['local'] = self._own(res['local'])
        return res

    def set_policy(self, policy='random'):
        self._env.set_policy(policy)

    def get_state(self):
        local, t = self._env.get_state()
        return self._own(local), t

    def set_state(self, local=None, t=None):
        self._env.set_state(None if local is None else self._union(local), t)

    def query_terminal(self):
        return self._env.query_terminal()

    def last_kernel(self, which='rollout'):
        return self._env.last_kernel(which)

    def sync(self):
        self._env.sync()

    def close(self):
        self._env.close()
