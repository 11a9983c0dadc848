// Packed-layout single step: one MapfEnv.step() of every env per launch (mapf_step with device-drawn uniforms), K = 2 or
// 4 agents per lane, Q = A/K lanes per env -- the lane layout of mapf_lq_rollout.hip applied to the one-launch-per-step
// path, for full groups and batches that fill their blocks; everything else (caller-supplied uniforms, odd agent
// counts, ragged batches) stays with lg_step_kernel (mapf_lg_kernels.hip), whose semantics this kernel reproduces
// line by line.
//
// A single step is launch- and latency-bound (three dependent round trips: state/actions -> table row -> stores), so
// nothing is staged into LDS: the 16-byte table rows are gathered from global memory (L2-resident), the sampled
// probability is rebuilt from the slot's members (no read of the slip rows on the common path), the reward is
// computed in registers.  What the packed layout buys is fewer waves and fewer replicated per-env instructions -- it
// shows at large batches (profiles/r02_single_step_scaling.txt).
#include "mapf_lq.hpp"

namespace mapf {

namespace {

// The slip-stream call(s) of a lane as a state that can be advanced a few rounds at a time, so that the rounds fill
// the two memory waits of the step (first loads, then table gathers) instead of running in one piece before the gathers
// are even issued.  NS = 1 or 2 calls in lockstep (same key, counters differ in the last word).
// Diagnostic build only (make step_stamps; never shipped): every wave records when it passed the stages of the step --
// s_memrealtime (100 MHz, one clock for the whole chip) at entry and exit, s_memtime (shader cycles) deltas in between,
// each stage stamp behind a full wait for the memory operations issued so far -- into the buffer passed as `uniforms`.
#ifdef MAPF_STEP_STAMPS
#define STEP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        stamp_[i] = t_ - cyc0_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STEP_STAMP(i)
#endif

template <int NS>
struct PhiloxRounds {
    uint32_t c[NS][4], k0, k1;
    template <int R>
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint64_t p0[NS], p1[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { p0[s] = uint64_t(0xD2511F53u) * c[s][0]; p1[s] = uint64_t(0xCD9E8D57u) * c[s][2]; }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                c[s][0] = __builtin_amdgcn_bitop3_b32(uint32_t(p1[s] >> 32), c[s][1], k0, 0x96);
                c[s][1] = uint32_t(p1[s]);
                c[s][2] = __builtin_amdgcn_bitop3_b32(uint32_t(p0[s] >> 32), c[s][3], k1, 0x96);
                c[s][3] = uint32_t(p0[s]);
            }
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
            asm volatile("" : "+s"(k0), "+s"(k1));   // (see philox4x32_10: keeps the key schedule out of 20 SGPRs)
        }
    }
};

// SCEN: the env's start / goal rows come from the handle's scenario table (StepArgs::scen) -- one byte per env and two
// small gathers that hit in L1 / L2, issued beside the move-table gathers, instead of two A-cell rows from HBM; the
// start cells are then at hand when the step ends the episode, so the state store needs no branch either.
//
// A single step is a chain of memory round trips with ~370 vector instructions between them; what this kernel is
// written around is the LENGTH of that chain (profiles/r03_single_step_*.txt):
//   * the pointers of the first loads, the agent count and the block size are LEADING SCALAR ARGUMENTS: this file is
//     compiled with -amdgpu-kernarg-preload-count, so the command processor delivers them in SGPRs with the wave and the
//     first loads are issued before any s_load of the argument block has come back (reading blockDim.x would be one
//     more scalar load in front of the address arithmetic: the block size travels as an argument);
//   * state, actions and the scenario byte are requested together; the scenario's rows and the move-table rows are the
//     second trip; the Philox rounds are split over the two waits.
template <int Q, int K, bool SCEN>
__global__ void __launch_bounds__(256) lq_step_kernel(uint16_t *const state, const uint8_t *const actions, const uint8_t *const scen,
                                                      const uint16_t *const rows, const MoveEntry *const mv, const uint32_t n_agents,
                                                      const uint32_t block_threads, const StepArgs p) {
    constexpr int P = K / 2;
#ifdef MAPF_STEP_STAMPS
    unsigned long long stamp_[8] = {}, real0_, cyc0_;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0_), "=s"(cyc0_) :: "memory");
#endif
    LaneCtx<Q> x;
    x.lane = threadIdx.x & 63u;
    x.g = x.lane & uint32_t(Q - 1);
    x.base = x.lane & ~uint32_t(Q - 1);
    x.e = ((blockIdx.x * block_threads + threadIdx.x) >> 6) * uint32_t(64 / Q) + x.lane / uint32_t(Q);
    x.v0 = x.v1 = true;
    const uint32_t e = x.e;
    const uint32_t lane_cell = e * n_agents + uint32_t(K) * x.g;    // my first agent's element index
    const uint32_t fixed_cell = uint32_t(K) * x.g;                  // ... in a broadcast row

    // ---- first trip: state, actions, scenario byte (or my cells of the goal row)
    uint32_t c[P], g[P], sc[P];
    const Packed<P> cells = Packed<P>::load(at(state, lane_cell));
    const uint32_t raw = K == 4 ? *reinterpret_cast<const uint32_t *>(at(actions, lane_cell))
                                : uint32_t(*reinterpret_cast<const uint16_t *>(at(actions, lane_cell)));
    uint32_t scen_id = 0u;
    Packed<P> gl{}, sl{};
    if (SCEN) scen_id = *at(scen, e);
    else gl = Packed<P>::load(at(rows, p.goal_broadcast ? fixed_cell : lane_cell));      // rows = the goal array
    __builtin_amdgcn_sched_barrier(0);

    // ---- the slip call of my pair(s) for steps 4h .. 4h+3 (one call per pair per step: a single step cannot amortise
    // it), first six rounds while the loads are in flight
    const uint64_t env_id = p.env_id_offset + e, t = first_step_index(p);
    PhiloxRounds<P> rng_state;
    {
        const uint32_t hi16 = uint32_t((t >> 2) >> 32) & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            rng_state.c[i][0] = uint32_t(env_id); rng_state.c[i][1] = uint32_t(env_id >> 32); rng_state.c[i][2] = uint32_t(t >> 2);
            rng_state.c[i][3] = hi16 | ((uint32_t(P) * x.g + uint32_t(i)) << 16);   // pair index; rslot = refine = 0 (slip_words)
        }
        rng_state.k0 = p.c.seed_lo; rng_state.k1 = p.c.seed_hi;
    }
#ifdef MAPF_STEP_STAMPS
    { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[0] = t_ - cyc0_; __builtin_amdgcn_sched_barrier(0); }   // argument block arrived
#endif
    if (p.c.need_rng) rng_state.template run<6>();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MAPF_STEP_STAMPS
    { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[1] = t_ - cyc0_; __builtin_amdgcn_sched_barrier(0); }   // six rounds done
    STEP_STAMP(2);   // first loads arrived
#endif

    // ---- second trip: the scenario's rows, then the move-table rows of my agents
    if (SCEN) {
        const uint32_t scen_row = scen_id * 2u * n_agents + fixed_cell;   // my cells of the env's start row; goal row: + A
        gl = Packed<P>::load(at(rows, scen_row + n_agents));
        sl = Packed<P>::load(at(rows, scen_row));
    }
#pragma unroll
    for (int i = 0; i < P; ++i) c[i] = cells.v[i];
    uint32_t cur[K], act[K], hi[K];
    MoveEntry entry[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        cur[k] = (k & 1) ? c[k / 2] >> 16 : c[k / 2] & 0xFFFFu;
        const uint32_t byte = (raw >> (8 * k)) & 0xFFu;
        act[k] = byte > 4u ? 0u : byte;
        entry[k] = move_entry<true>(mv, p.c.n_cells, cur[k], act[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (p.c.need_rng) rng_state.template run<4>();
#ifdef MAPF_STEP_STAMPS
    { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[3] = t_ - cyc0_; __builtin_amdgcn_sched_barrier(0); }   // gathers issued, four rounds done
    STEP_STAMP(4);   // gathers arrived
#endif
    Words4 rng[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
        rng[i] = p.c.need_rng ? Words4{rng_state.c[i][0], rng_state.c[i][1], rng_state.c[i][2], rng_state.c[i][3]} : Words4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < P; ++i) { g[i] = gl.v[i]; sc[i] = sl.v[i]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const uint32_t word = step_word(rng[i], t);
        hi[2 * i] = word & 0xFFFFu;
        hi[2 * i + 1] = word >> 16;
    }
    double q[K];
    uint32_t nx[K], tie_all = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t tie;
        slip_move_hi_members(p.c, entry[k], hi[k], nx[k], q[k], tie);
        tie_all = min(tie_all, tie);
    }
    if (__builtin_expect(__any(tie_all == 0u && p.c.need_rng), 0)) {
        // a top-16-bit tie somewhere in the wave: redo with all 53 bits (the slip rows are read from global memory here)
#pragma unroll
        for (int k = 0; k < K; ++k)
            slip_move<false>(p.slip, entry[k], refine_mantissa(p.c, env_id, t, uint32_t(K) * x.g + uint32_t(k), hi[k]), 0.0, nx[k], q[k]);
    }
    uint32_t n[P];
#pragma unroll
    for (int i = 0; i < P; ++i) n[i] = nx[2 * i] | (nx[2 * i + 1] << 16);

    // is_terminal(prev) and the collision tests in one pass over the agent pairs; per-env facts OR-reduced over the group
    const PairAcc<true> acc = packed_pair_tests<Q, P, true, true>(x, c, n);
    uint32_t away_next = 0u, away_prev = 0u;
#pragma unroll
    for (int i = 0; i < P; ++i) { away_next |= n[i] ^ g[i]; away_prev |= c[i] ^ g[i]; }
    uint32_t flags = (PairAcc<true>::hit(acc.vertex) ? 1u : 0u) | (PairAcc<true>::hit(acc.swap) ? 2u : 0u) | (away_next ? 4u : 0u) |
                     (PairAcc<true>::hit(acc.dup) ? 8u : 0u) | (away_prev ? 16u : 0u);
    flags = group_reduce<Q, false>(flags, x);
    const bool was_terminal = (flags & 8u) != 0u || (flags & 16u) == 0u;     // mapf_env.py:210-223
    const uint32_t f = flags & 7u;

    // total_prob: left-to-right product over agents 0..A-1 (mapf_env.py:257); the total ends in the group's last lane
    const double prob = packed_prob_product<Q, K>(q);

    // _living_reward (mapf_env.py:436-446), calc_transition_reward_from_local_states (:225-235: collision before goal)
    double living = p.c.r_living;
    if (p.c.criteria == 1u) {
        uint32_t mine = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t goal_k = (k & 1) ? g[k / 2] >> 16 : g[k / 2] & 0xFFFFu;
            mine += (cur[k] == goal_k && act[k] == 0u) ? 1u : 0u;
        }
        const int stayed = int(group_reduce<Q, true>(mine, x));
        living = __dmul_rn(double(int(n_agents) - stayed), p.c.r_living);
    }
    const bool coll = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
    const double r = coll ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
    const double reward = was_terminal ? 0.0 : r;                  // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0})
    const bool done = coll || goal_next || was_terminal;
    if (was_terminal) {
#pragma unroll
        for (int i = 0; i < P; ++i) n[i] = c[i];
    }

    Packed<P> out;
#pragma unroll
    for (int i = 0; i < P; ++i) out.v[i] = n[i];
#ifdef MAPF_STEP_STAMPS
    asm volatile("" :: "v"(reward), "v"(prob), "v"(out.v[0]));
    STEP_STAMP(5);   // everything computed
#endif
    if (p.out_local) out.store(at(p.out_local, lane_cell));
    if (x.g == uint32_t(Q - 1) && p.out_prob) *at(p.out_prob, e) = was_terminal ? 0.0 : prob;
    if (x.g == 0u) {
        if (p.out_reward) *at(p.out_reward, e) = reward;
        if (p.out_done) *at(p.out_done, e) = done ? 1 : 0;
        if (p.out_collision) *at(p.out_collision, e) = (coll && !was_terminal) ? 1 : 0;
        if (p.out_was_terminal) *at(p.out_was_terminal, e) = was_terminal ? 1 : 0;
    }
    if (SCEN) {                                                    // MapfEnv.reset(): start cells, no reseed
        const bool back = p.auto_reset && done;                    // (a terminal state that stays is rewritten as it is)
        Packed<P> keep;
#pragma unroll
        for (int i = 0; i < P; ++i) keep.v[i] = back ? sc[i] : n[i];
        keep.store(at(state, lane_cell));
    } else if (p.auto_reset && done) {
        Packed<P>::load(at(p.start, p.start_broadcast ? fixed_cell : lane_cell)).store(at(state, lane_cell));
    } else if (!was_terminal) {
        out.store(at(state, lane_cell));
    }
#ifdef MAPF_STEP_STAMPS
    {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t6_, t7_, real1_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t6_) :: "memory");                          // stores issued
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t7_), "=s"(real1_) :: "memory");   // stores acknowledged
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        uint32_t xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        if (x.lane == 0u && p.uniforms) {
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(const_cast<double *>(p.uniforms)) +
                                      uint64_t((blockIdx.x * block_threads + threadIdx.x) >> 6) * 12u;
            dst[0] = real0_; dst[1] = real1_;
            for (int i = 0; i < 6; ++i) dst[2 + i] = stamp_[i];
            dst[8] = t6_ - cyc0_; dst[9] = t7_ - cyc0_; dst[10] = hw_id; dst[11] = xcc_id;
        }
    }
#endif
    signal_step_done(p.done_flag, p.done_seq);
}

}  // namespace

// true when the packed layout took the launch (*err = its status); false = not applicable, use lg_step_kernel
bool try_launch_step_lq(int n_agents, const StepArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err) {
#ifndef MAPF_STEP_STAMPS   // (the diagnostic build receives its stamp buffer through `uniforms`)
    if (args.uniforms != nullptr) return false;
#endif
    if (!tune.quad_lanes) return false;
    int K = 0;
    if (tune.force_k != 2 && n_agents % 4 == 0) K = 4;
    else if (tune.force_k != 4 && n_agents % 2 == 0 && n_agents >= 4) K = 2;
    else return false;
    const int Q = n_agents / K;
    if (Q > 16 || (Q & (Q - 1)) != 0) return false;
    const uint64_t lanes = args.n_envs * uint64_t(Q);
    unsigned block = 256u;
    while (block > 64u && lanes < 256u * uint64_t(block)) block /= 2u;   // small batches: spread over the CUs
    const uint64_t per_block = block / unsigned(Q);
    if (args.n_envs == 0 || args.n_envs % per_block != 0) return false;
    const unsigned grid = unsigned(args.n_envs / per_block);
    const uint32_t A = uint32_t(n_agents);
    const bool scen = args.scen != nullptr;
    note_kernel("lq_step_kernel<Q=%d,K=%d%s> block=%u (packed layout: %d agents per lane%s)", Q, K, scen ? ",SCEN" : "", block, K,
                scen ? ", start / goal rows from the scenario table" : "");
#define MAPF_LQ_STEP(QQ, KK)                                                                                   \
    if (Q == QQ && K == KK) {                                                                                  \
        if (scen) hipLaunchKernelGGL((lq_step_kernel<QQ, KK, true>), dim3(grid), dim3(block), 0, stream, args.state, args.actions,  \
                                     args.scen, args.scen_rows, args.mv, A, block, args);                                          \
        else hipLaunchKernelGGL((lq_step_kernel<QQ, KK, false>), dim3(grid), dim3(block), 0, stream, args.state, args.actions,    \
                                static_cast<const uint8_t *>(nullptr), args.goal, args.mv, A, block, args);                        \
        *err = hipGetLastError();                                                                              \
        return true;                                                                                           \
    }
    MAPF_LQ_STEP(1, 4) MAPF_LQ_STEP(2, 4) MAPF_LQ_STEP(4, 4) MAPF_LQ_STEP(8, 4) MAPF_LQ_STEP(16, 4)
    MAPF_LQ_STEP(2, 2) MAPF_LQ_STEP(4, 2) MAPF_LQ_STEP(8, 2) MAPF_LQ_STEP(16, 2)
#undef MAPF_LQ_STEP
    return false;
}

}  // namespace mapf
