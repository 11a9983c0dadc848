// Thread-per-env kernel family (A <= 16, opt-in with MAPF_FLAG_THREAD_PER_ENV; the default family is the
// lane-group one in mapf_lg_kernels.hip / mapf_lg_rollout.hip).  Kept as an independent implementation of the same
// semantics that the parity tests cross-check against the lane-group kernels.
//
//   step_kernel<A>      one transition per env per launch; every output written to HBM
//                       (MapfEnv.step, mapf_env.py:237-266) + optional fused auto-reset
//   rollout_kernel<A>   T transitions per env in one launch, state in registers,
//                       optional trajectory recording (the caller-side loop around step)
// (reset / query_terminal / fill_actions are run-time-A kernels in mapf_lg_kernels.hip)
//
// Mapping: one thread = one env, all A agents in registers; rows are env-major so a lane
// reads/writes its whole row with the widest aligned access (dwordx4 at A = 8).
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"

namespace mapf {

template <int A>
__device__ __forceinline__ void unpack_cells(const Row<uint16_t, A> &r, uint32_t (&dst)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) dst[i] = r.v[i];
}

template <int A>
__device__ __forceinline__ void load_cells(const uint16_t *base, uint64_t e, bool broadcast,
                                           uint32_t (&dst)[A]) {
    const Row<uint16_t, A> r = load_row<uint16_t, A>(base, broadcast ? 0 : e);
    unpack_cells<A>(r, dst);
}

template <int A, bool EXT_UNIFORMS>
__global__ void __launch_bounds__(256) step_kernel(const StepArgs p) {
    __shared__ SlipRow slip[8];
    stage_slip_table(p.slip, slip);
    const uint64_t e = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= p.n_envs) return;

    uint32_t prev[A], goal[A], act[A];
    load_cells<A>(p.state, e, false, prev);
    load_cells<A>(p.goal, e, p.goal_broadcast, goal);
    {
        const Row<uint8_t, A> ar = load_row<uint8_t, A>(p.actions, e);
#pragma unroll
        for (int i = 0; i < A; ++i) act[i] = ar.v[i];
    }
    double uext[A];
    if (EXT_UNIFORMS) {
        const Row<double, A> ur = load_row<double, A>(p.uniforms, e);
#pragma unroll
        for (int i = 0; i < A; ++i) uext[i] = ur.v[i];
    }

    StepResult<A> res;
    env_transition<A, EXT_UNIFORMS>(p.c, p.mv, slip, prev, goal, act, uext, p.env_id_offset + e, first_step_index(p), res);

    Row<uint16_t, A> nx;
#pragma unroll
    for (int i = 0; i < A; ++i) nx.v[i] = uint16_t(res.next[i]);
    if (p.out_local) store_row<uint16_t, A>(p.out_local, e, nx);
    if (p.out_reward) p.out_reward[e] = res.reward;
    if (p.out_prob) p.out_prob[e] = res.prob;
    if (p.out_done) p.out_done[e] = res.done ? 1 : 0;
    if (p.out_collision) p.out_collision[e] = res.collision ? 1 : 0;
    if (p.out_was_terminal) p.out_was_terminal[e] = res.was_terminal ? 1 : 0;

    if (p.auto_reset && res.done) {
        store_row<uint16_t, A>(p.state, e, load_row<uint16_t, A>(p.start, p.start_broadcast ? 0 : e));
    } else if (!res.was_terminal) {
        store_row<uint16_t, A>(p.state, e, nx);
    }
    signal_step_done(p.done_flag, p.done_seq);
}

template <int A>
__global__ void __launch_bounds__(256) rollout_kernel(const RolloutArgs p) {
    __shared__ SlipRow slip[8];
    stage_slip_table(p.slip, slip);
    const uint64_t e = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= p.n_envs) return;

    uint32_t cur[A], goal[A];
    load_cells<A>(p.state, e, false, cur);
    load_cells<A>(p.goal, e, p.goal_broadcast, goal);

    double ret = (p.accumulate && p.out_returns) ? p.out_returns[e] : 0.0;
    uint32_t episodes = (p.accumulate && p.out_episodes) ? p.out_episodes[e] : 0u;
    uint32_t collisions = (p.accumulate && p.out_collisions) ? p.out_collisions[e] : 0u;
    const uint64_t env_id = p.env_id_offset + e, t0 = first_step_index(p);

    for (uint32_t s = 0; s < p.n_steps; ++s) {
        const uint64_t t = t0 + s;
        const uint64_t row = uint64_t(s) * p.n_envs + e;
        uint32_t act[A];
        if (p.actions) {
            const Row<uint8_t, A> ar = load_row<uint8_t, A>(p.actions, row);
#pragma unroll
            for (int i = 0; i < A; ++i) act[i] = ar.v[i];
        } else if (p.policy_cells) {
#pragma unroll
            for (int i = 0; i < A; ++i) act[i] = greedy_action(p.policy_cells, p.c.n_cells, cur[i], p.policy_cells[goal[i]].x);
        } else {
            policy_actions<A>(p.c, env_id, t, act);
        }
        StepResult<A> res;
        env_transition<A, false>(p.c, p.mv, slip, cur, goal, act, nullptr, env_id, t, res);

        ret = __dadd_rn(ret, res.reward);
        episodes += res.done ? 1u : 0u;
        collisions += res.collision ? 1u : 0u;
        if (p.rec_local) {
            Row<uint16_t, A> nx;
#pragma unroll
            for (int i = 0; i < A; ++i) nx.v[i] = uint16_t(res.next[i]);
            store_row<uint16_t, A>(p.rec_local, row, nx);
        }
        if (p.rec_reward) p.rec_reward[row] = res.reward;
        if (p.rec_prob) p.rec_prob[row] = res.prob;
        if (p.rec_done) p.rec_done[row] = res.done ? 1 : 0;
        if (p.rec_collision) p.rec_collision[row] = res.collision ? 1 : 0;

        if (p.auto_reset && res.done) {
            load_cells<A>(p.start, e, p.start_broadcast, cur);   // rare: re-read the start row instead of pinning A registers
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) cur[i] = res.next[i];
        }
    }

    Row<uint16_t, A> fin;
#pragma unroll
    for (int i = 0; i < A; ++i) fin.v[i] = uint16_t(cur[i]);
    store_row<uint16_t, A>(p.state, e, fin);
    if (p.out_returns) p.out_returns[e] = ret;
    if (p.out_episodes) p.out_episodes[e] = episodes;
    if (p.out_collisions) p.out_collisions[e] = collisions;
}

// ------------------------------------------------------------------- launchers
static inline unsigned pick_block(uint64_t n) { return n <= (1u << 18) ? 64u : 256u; }
static inline unsigned grid_for(uint64_t n, unsigned block) { return unsigned((n + block - 1) / block); }

// This file is compiled once per agent-count group g (-DMAPF_GROUP=g, A in 4g+1 .. 4g+4) so the
// 32 specialisations build in parallel; mapf_dispatch.hip routes a launch to its group.
#ifndef MAPF_GROUP
#error "compile with -DMAPF_GROUP=0..3"
#endif
#define MAPF_A0 (4 * MAPF_GROUP + 1)
#define MAPF_FOR_EACH_A(X) X((MAPF_A0)) X((MAPF_A0 + 1)) X((MAPF_A0 + 2)) X((MAPF_A0 + 3))
#define MAPF_CAT2(a, b) a##b
#define MAPF_CAT(a, b) MAPF_CAT2(a, b)
#define MAPF_G(name) MAPF_CAT(name, MAPF_GROUP)

hipError_t MAPF_G(launch_step_g)(int n_agents, const StepArgs &args, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    const unsigned block = pick_block(args.n_envs), grid = grid_for(args.n_envs, block);
    note_kernel("step_kernel<A=%d,%s> block=%u (thread per env)", n_agents, args.uniforms ? "EXT_UNIFORMS" : "PHILOX", block);
    switch (n_agents) {
#define X(N)                                                                                       \
    case N:                                                                                        \
        if (args.uniforms) hipLaunchKernelGGL((step_kernel<N, true>), dim3(grid), dim3(block), 0, stream, args);  \
        else hipLaunchKernelGGL((step_kernel<N, false>), dim3(grid), dim3(block), 0, stream, args); \
        break;
        MAPF_FOR_EACH_A(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Only the spill-free specialisations exist in the shipped objects (kTpeRolloutMaxAgents; inside a template the
// discarded branch is not instantiated): the larger ones are built by `make tpe16` for tools/exp/tpe_spill_repro.sh only.
template <int N>
static hipError_t launch_tpe_rollout(const RolloutArgs &args, unsigned grid, unsigned block, hipStream_t stream) {
    if constexpr (N <= kTpeRolloutMaxAgents) {
        hipLaunchKernelGGL((rollout_kernel<N>), dim3(grid), dim3(block), 0, stream, args);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

hipError_t MAPF_G(launch_rollout_g)(int n_agents, const RolloutArgs &args, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    const unsigned block = pick_block(args.n_envs), grid = grid_for(args.n_envs, block);
    note_kernel("rollout_kernel<A=%d> block=%u (thread per env)", n_agents, block);
    switch (n_agents) {
#define X(N) case N: return launch_tpe_rollout<N>(args, grid, block, stream);
        MAPF_FOR_EACH_A(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mapf
