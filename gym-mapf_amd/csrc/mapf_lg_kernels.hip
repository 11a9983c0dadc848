// Lane-group family: single-step kernel, run-time-A helper kernels and their launchers (device code: mapf_lg.hpp).
#include "mapf_lg.hpp"

namespace mapf {

template <int L, bool FULL, bool EXT_UNIFORMS>
__global__ void __launch_bounds__(256) lg_step_kernel(const StepArgs p, const uint32_t n_agents) {
    bool live;
    const LaneCtx<L> x = lane_ctx<L>(n_agents, p.n_envs, live);
    const uint32_t e = x.e;

    uint32_t cur0, cur1, goal0, goal1, act0, act1;
    load_pair<uint16_t>(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    load_pair<uint16_t>(p.goal, p.goal_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, goal0, goal1);
    load_pair<uint8_t>(p.actions, e, n_agents, x.g, x.v0, x.v1, act0, act1);
    double u0 = 0.0, u1 = 0.0;
    if (EXT_UNIFORMS) {
        const double *up = at(p.uniforms, e * n_agents + 2u * x.g);
        if (x.v0) u0 = up[0];
        if (x.v1) u1 = up[1];
    }
    // A single step is launch-latency bound: the sampled probability is rebuilt from its members (no third dependent
    // memory round trip); the 8 slip rows are only read on the exact-tie path and for caller-supplied uniforms,
    // straight from global memory (they stay in L1/L2) rather than staged into LDS behind a barrier.
    const SlipRow *rows = p.slip;

    uint32_t next0, next1;
    EnvOut o;
#ifdef MAPF_STAMPS
    StampCtx st{};
#endif
    uint32_t word = 0u;   // this step's slip word of my pair: the call of my quad (g >> 1), word 2 * (t & 1) + (g & 1)
    const uint64_t t = first_step_index(p);
    if (!EXT_UNIFORMS && p.c.need_rng) word = quad_step_word(slip_words(p.c, p.env_id_offset + e, t >> 1, x.g >> 1, 0u, 0u), t, x.g & 1u);
    lg_transition<L, FULL, EXT_UNIFORMS, false, false, false, !EXT_UNIFORMS>(p.c, p.mv, rows, nullptr, x, n_agents, cur0, cur1, goal0, goal1, act0, act1,
                                                u0, u1, p.env_id_offset + e, t, word, false, next0, next1, o STAMP_ARG);
    if (!live) return;

    if (p.out_local) store_cells(p.out_local, e, n_agents, x.g, x.v0, x.v1, next0, next1);
    if (x.g == uint32_t(L - 1) && p.out_prob) *at(p.out_prob, e) = o.prob;   // the product chain ends in the last lane
    if (x.g == 0u) {
        if (p.out_reward) *at(p.out_reward, e) = o.reward;
        if (p.out_done) *at(p.out_done, e) = o.done() ? 1 : 0;
        if (p.out_collision) *at(p.out_collision, e) = o.collision() ? 1 : 0;
        if (p.out_was_terminal) *at(p.out_was_terminal, e) = o.was_terminal ? 1 : 0;
    }
    if (p.auto_reset && o.done()) {
        uint32_t s0, s1;
        load_pair<uint16_t>(p.start, p.start_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, s0, s1);
        store_cells(p.state, e, n_agents, x.g, x.v0, x.v1, s0, s1);
    } else if (!o.was_terminal) {
        store_cells(p.state, e, n_agents, x.g, x.v0, x.v1, next0, next1);
    }
    signal_step_done(p.done_flag, p.done_seq);
}

// ------------------------------------------------------- run-time-A helper kernels
// masked MapfEnv.reset (mapf_env.py:290-293): one thread per cell
__global__ void __launch_bounds__(256) reset_kernel(uint16_t *state, const uint16_t *start, bool start_broadcast,
                                                    const uint8_t *mask, uint64_t n_envs, uint32_t n_agents) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_envs * n_agents) return;
    const uint64_t e = i / n_agents;
    if (mask && mask[e] == 0) return;
    state[i] = start[start_broadcast ? i - e * n_agents : i];
}

// MapfEnv.is_terminal (mapf_env.py:210-223) of the stored state: one thread per env
__global__ void __launch_bounds__(256) query_terminal_kernel(const uint16_t *state, const uint16_t *goal,
                                                             bool goal_broadcast, uint8_t *out, uint64_t n_envs,
                                                             uint32_t n_agents) {
    const uint64_t e = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    const uint16_t *cur = state + e * n_agents, *g = goal + (goal_broadcast ? 0 : e * n_agents);
    bool dup = false, all_goal = true;
    for (uint32_t i = 0; i < n_agents; ++i) {
        all_goal &= (cur[i] == g[i]);
        for (uint32_t j = i + 1; j < n_agents; ++j) dup |= (cur[i] == cur[j]);
    }
    out[e] = (dup || all_goal) ? 1 : 0;
}

// policy stream (oracle/philox.py random_actions_np): one thread per (four-step block, env, agent quad) -- one Philox
// call, whose word t & 3 holds the quad's four action bytes of step t
__global__ void __launch_bounds__(256) fill_actions_kernel(uint8_t *actions, EnvConsts c, uint64_t env_id_offset,
                                                           uint64_t n_envs, uint64_t t0, uint64_t n_steps,
                                                           uint32_t n_agents) {
    const uint32_t quads = (n_agents + 3u) / 4u;
    const uint64_t m0 = t0 >> 2, n_blocks = ((t0 + n_steps - 1u) >> 2) - m0 + 1u;
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_blocks * n_envs * quads) return;
    const uint64_t row = i / quads;                              // (block, env)
    const uint32_t q = uint32_t(i - row * quads);
    const uint64_t blk = row / n_envs, e = row - blk * n_envs;
    const uint64_t m = m0 + blk;
    const Words4 w = policy_words(c, env_id_offset + e, m, q);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t t = 4u * m + j;
        if (t < t0 || t >= t0 + n_steps) continue;
        const uint32_t word = pick_word(w, j);
        uint8_t *dst = actions + ((t - t0) * n_envs + e) * n_agents + 4u * q;
        if ((n_agents & 3u) == 0u) {                            // a quad's four bytes as one store
            *reinterpret_cast<uint32_t *>(dst) = policy_action_rt(word, 0u) | (policy_action_rt(word, 1u) << 8) |
                                                 (policy_action_rt(word, 2u) << 16) | (policy_action_rt(word, 3u) << 24);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (4u * q + k < n_agents) dst[k] = uint8_t(policy_action_rt(word, k));
        }
    }
}

// the device-side step index of a recorded graph (StepArgs::t_dev): advanced by the graph's last node
__global__ void __launch_bounds__(64) advance_step_index_kernel(uint64_t *t_dev, uint64_t n, bool set) {
    if (blockIdx.x == 0u && threadIdx.x == 0u) *t_dev = set ? n : *t_dev + n;
}

hipError_t launch_advance_step_index(uint64_t *t_dev, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(advance_step_index_kernel, dim3(1), dim3(64), 0, stream, t_dev, n, false);
    return hipGetLastError();
}

hipError_t launch_set_step_index(uint64_t *t_dev, uint64_t value, hipStream_t stream) {
    hipLaunchKernelGGL(advance_step_index_kernel, dim3(1), dim3(64), 0, stream, t_dev, value, true);
    return hipGetLastError();
}

static inline hipError_t grid_1d(uint64_t n, unsigned block, unsigned &grid) {
    const uint64_t g = (n + block - 1) / block;
    if (g > 0x7FFFFFFFull) return hipErrorInvalidValue;
    grid = unsigned(g);
    return hipSuccess;
}

hipError_t launch_reset(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream) {
    if (n_envs == 0) return hipSuccess;
    unsigned grid;
    if (hipError_t e = grid_1d(n_envs * uint64_t(n_agents), 256, grid)) return e;
    hipLaunchKernelGGL(reset_kernel, dim3(grid), dim3(256), 0, stream, state, start, start_broadcast, mask, n_envs,
                       uint32_t(n_agents));
    return hipGetLastError();
}

hipError_t launch_query_terminal(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream) {
    if (n_envs == 0) return hipSuccess;
    unsigned grid;
    if (hipError_t e = grid_1d(n_envs, 256, grid)) return e;
    hipLaunchKernelGGL(query_terminal_kernel, dim3(grid), dim3(256), 0, stream, state, goal, goal_broadcast, out,
                       n_envs, uint32_t(n_agents));
    return hipGetLastError();
}

hipError_t launch_fill_actions(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream) {
    if (n_envs == 0 || n_steps == 0) return hipSuccess;
    const uint64_t n_blocks = ((t0 + n_steps - 1u) >> 2) - (t0 >> 2) + 1u;
    unsigned grid;
    if (hipError_t e = grid_1d(n_blocks * n_envs * uint64_t((n_agents + 3) / 4), 256, grid)) return e;
    hipLaunchKernelGGL(fill_actions_kernel, dim3(grid), dim3(256), 0, stream, actions, c, env_id_offset, n_envs, t0,
                       n_steps, uint32_t(n_agents));
    return hipGetLastError();
}

// ------------------------------------------------------------------- launchers
int lg_group_size(int n_agents) {
    int pairs = (n_agents + 1) / 2, L = 1;
    while (L < pairs) L <<= 1;
    return L;
}

hipError_t launch_step_lg(int n_agents, const StepArgs &args, const RolloutTuning &tune, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    hipError_t packed_status;
    if (try_launch_step_lq(n_agents, args, tune, stream, &packed_status)) return packed_status;
    const int L = lg_group_size(n_agents);
    const bool full = n_agents == 2 * L;
    unsigned grid, block;
    lg_geometry(L, args.n_envs, grid, block);
    const uint32_t A = uint32_t(n_agents);
    note_kernel("lg_step_kernel<L=%d,%s,%s> block=%u (pair layout: 2 agents per lane)", L, full ? "FULL" : "RAGGED",
                args.uniforms ? "EXT_UNIFORMS" : "PHILOX", block);
    switch (L) {
#define X(N)                                                                                                         \
    case N:                                                                                                          \
        if (args.uniforms) {                                                                                         \
            if (full) hipLaunchKernelGGL((lg_step_kernel<N, true, true>), dim3(grid), dim3(block), 0, stream, args, A);   \
            else hipLaunchKernelGGL((lg_step_kernel<N, false, true>), dim3(grid), dim3(block), 0, stream, args, A);       \
        } else {                                                                                                     \
            if (full) hipLaunchKernelGGL((lg_step_kernel<N, true, false>), dim3(grid), dim3(block), 0, stream, args, A);  \
            else hipLaunchKernelGGL((lg_step_kernel<N, false, false>), dim3(grid), dim3(block), 0, stream, args, A);      \
        }                                                                                                            \
        break;
        MAPF_FOR_EACH_L(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mapf
