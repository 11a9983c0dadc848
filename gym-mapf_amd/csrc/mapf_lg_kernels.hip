// Lane-group kernels: one env is stepped by L adjacent lanes of a wavefront, two agents per lane.
//
// Why: with one thread per env the pair tests and the slip logic of all A agents sit in one lane's
// registers, so 65536 envs are only 1024 waves -- one per SIMD, no latency hiding, and at A = 32 the
// register file overflows.  Here a 64-lane wave carries 64/L envs (L = pow2 >= ceil(A/2)); each lane
// owns agents 2g and 2g+1 of its env (g = lane % L): one Philox4x32 call yields exactly its two
// uniforms, its two cells travel as one packed dword, and the O(A^2) pair tests become L/2 rotations of
// that dword inside the group (DPP quad_perm / row_ror where the group fits, ds_bpermute otherwise).
// Per-env facts are combined with wave ballots; the float64 probability product is evaluated in agent
// order so it rounds exactly like the reference's left-to-right `total_prob *= p` (mapf_env.py:257).
// A is a run-time value: slots >= A are ghosts that never match anything, sit "on goal" and contribute a
// factor 1.0; FULL specialisations (A == 2L) drop all ghost bookkeeping.
//
// Same semantics, arguments and outputs as step_kernel / rollout_kernel in mapf_kernels.hip.
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"

namespace mapf {

template <int L>
struct LaneCtx {
    uint32_t lane, g, base;      // lane in wave, position in group, first lane of the group
    uint64_t e;                  // env index (local to the handle)
    bool v0, v1;                 // my two agent slots exist (2g < A, 2g+1 < A)
};

// ------------------------------------------------------------------ cross-lane moves inside a group
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xF, 0xF, false));
}

// value held by lane (g + S) mod L of my group
template <int L, int S>
__device__ __forceinline__ uint32_t group_rot(uint32_t v, const LaneCtx<L> &x) {
    static_assert(S >= 1 && S < (L > 1 ? L : 2), "rotation out of range");
    if constexpr (L == 2) {
        return dpp_mov<0xB1>(v);                                   // quad_perm [1,0,3,2]
    } else if constexpr (L == 4) {
        constexpr int ctrl = ((0 + S) & 3) | (((1 + S) & 3) << 2) | (((2 + S) & 3) << 4) | (((3 + S) & 3) << 6);
        return dpp_mov<ctrl>(v);                                   // quad_perm rotation
    } else if constexpr (L == 8) {
        const uint32_t fwd = dpp_mov<0x120 + (16 - S)>(v);         // row_ror: lane i <- lane (i + S) mod 16
        const uint32_t wrap = dpp_mov<0x120 + (8 - S)>(v);         //          lane i <- lane (i + S - 8) mod 16
        uint32_t g = x.g;
        asm volatile("" : "+v"(g));                                // recompute the predicate here: hoisting it out
        return (g + uint32_t(S) < 8u) ? fwd : wrap;                // of the step loop costs an SGPR pair per round
    } else if constexpr (L == 16) {
        return dpp_mov<0x120 + (16 - S)>(v);
    } else {
        return uint32_t(__shfl(int(v), int(x.base + ((x.g + uint32_t(S)) & uint32_t(L - 1))), 64));
    }
}

// value held by lane K of my group
template <int L, int K>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, const LaneCtx<L> &x) {
    if constexpr (L == 1) {
        return v;
    } else if constexpr (L == 2) {
        return dpp_mov<(K == 0 ? 0xA0 : 0xF5)>(v);                 // quad_perm [K,K,K+2,K+2]
    } else if constexpr (L == 4) {
        return dpp_mov<K * 0x55>(v);                               // quad_perm [K,K,K,K]
    } else {
        return uint32_t(__shfl(int(v), int(x.base) + K, 64));
    }
}

template <int L, int K>
__device__ __forceinline__ double group_bcast_f64(double v, const LaneCtx<L> &x) {
    const uint32_t lo = group_bcast<L, K>(uint32_t(__double2loint(v)), x);
    const uint32_t hi = group_bcast<L, K>(uint32_t(__double2hiint(v)), x);
    return __hiloint2double(int(hi), int(lo));
}

// bits of a wave ballot that belong to my group, right-aligned
template <int L>
__device__ __forceinline__ uint64_t group_bits(uint64_t ballot, uint32_t base) {
    if (L == 64) return ballot;
    return (ballot >> base) & ((uint64_t(1) << L) - 1u);
}

// ------------------------------------------------------------------ pair tests
// min over agent pairs of xor (0 <=> equal).  dup: prev_i == prev_j (is_terminal, mapf_env.py:210-223);
// vertex: next_i == next_j; swap: prev_i == next_j and prev_j == next_i (mapf_env.py:378-389).
struct PairAcc {
    uint32_t dup = 0xFFFFFFFFu, vertex = 0xFFFFFFFFu, swap = 0xFFFFFFFFu;
};

// one rotation step: my two agents against the two agents of group position `og`, whose packed cells
// arrive in o_prev / o_next
template <int L, bool FULL, bool DUP, bool MOVES>
__device__ __forceinline__ void pair_apply(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                           uint32_t next0, uint32_t next1, uint32_t o_prev, uint32_t o_next,
                                           uint32_t og, PairAcc &acc) {
    const uint32_t op0 = o_prev & 0xFFFFu, op1 = o_prev >> 16;
    uint32_t g0 = 0u, g1 = 0u, g2 = 0u, g3 = 0u;    // ghost masks: 1 forces "different"
    if (!FULL) {
        const bool o0 = 2u * og < n_agents, o1 = 2u * og + 1u < n_agents;
        g0 = (x.v0 && o0) ? 0u : 1u; g1 = (x.v0 && o1) ? 0u : 1u;
        g2 = (x.v1 && o0) ? 0u : 1u; g3 = (x.v1 && o1) ? 0u : 1u;
    }
    if (DUP) {
        acc.dup = min(acc.dup, min((cur0 ^ op0) | g0, (cur0 ^ op1) | g1));
        acc.dup = min(acc.dup, min((cur1 ^ op0) | g2, (cur1 ^ op1) | g3));
    }
    if (MOVES) {
        const uint32_t on0 = o_next & 0xFFFFu, on1 = o_next >> 16;
        const uint32_t fwd0 = cur0 | (next0 << 16), fwd1 = cur1 | (next1 << 16);
        const uint32_t rev0 = on0 | (op0 << 16), rev1 = on1 | (op1 << 16);
        acc.vertex = min(acc.vertex, min((next0 ^ on0) | g0, (next0 ^ on1) | g1));
        acc.vertex = min(acc.vertex, min((next1 ^ on0) | g2, (next1 ^ on1) | g3));
        acc.swap = min(acc.swap, min((fwd0 ^ rev0) | g0, (fwd0 ^ rev1) | g1));
        acc.swap = min(acc.swap, min((fwd1 ^ rev0) | g2, (fwd1 ^ rev1) | g3));
    }
}

// rotations 1..L/2: unrolled with DPP moves for groups up to 16 lanes, a rolled ds_bpermute loop beyond
// (32 unrolled rounds would cost hundreds of registers for no gain)
template <int L, int S, bool FULL, bool DUP, bool MOVES>
struct PairRounds {
    static __device__ __forceinline__ void run(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                               uint32_t next0, uint32_t next1, uint32_t pk_prev, uint32_t pk_next,
                                               PairAcc &acc) {
        if constexpr (L >= 32) {
#pragma unroll 2
            for (uint32_t s = 1; s <= uint32_t(L / 2); ++s) {
                const uint32_t og = (x.g + s) & uint32_t(L - 1);
                const int src = int(x.base + og);
                const uint32_t o_prev = uint32_t(__shfl(int(pk_prev), src, 64));
                const uint32_t o_next = MOVES ? uint32_t(__shfl(int(pk_next), src, 64)) : 0u;
                pair_apply<L, FULL, DUP, MOVES>(x, n_agents, cur0, cur1, next0, next1, o_prev, o_next, og, acc);
            }
        } else if constexpr (S <= L / 2 && L > 1) {
            const uint32_t o_prev = group_rot<L, S>(pk_prev, x);
            const uint32_t o_next = MOVES ? group_rot<L, S>(pk_next, x) : 0u;
            pair_apply<L, FULL, DUP, MOVES>(x, n_agents, cur0, cur1, next0, next1, o_prev, o_next,
                                            (x.g + uint32_t(S)) & uint32_t(L - 1), acc);
            PairRounds<L, S + 1, FULL, DUP, MOVES>::run(x, n_agents, cur0, cur1, next0, next1, pk_prev, pk_next, acc);
        }
    }
};

// all pairs of the env: my own two agents, then rotations 1..L/2 (every unordered lane pair is met)
template <int L, bool FULL, bool DUP, bool MOVES>
__device__ __forceinline__ PairAcc pair_tests(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                              uint32_t next0, uint32_t next1) {
    PairAcc acc;
    const uint32_t ghost = (FULL || x.v1) ? 0u : 1u;
    if (DUP) acc.dup = (cur0 ^ cur1) | ghost;
    if (MOVES) {
        acc.vertex = (next0 ^ next1) | ghost;
        acc.swap = ((cur0 | (next0 << 16)) ^ (next1 | (cur1 << 16))) | ghost;
    }
    PairRounds<L, 1, FULL, DUP, MOVES>::run(x, n_agents, cur0, cur1, next0, next1, cur0 | (cur1 << 16),
                                             next0 | (next1 << 16), acc);
    return acc;
}

// MapfEnv.is_terminal (mapf_env.py:210-223) of the group's env
template <int L, bool FULL>
__device__ __forceinline__ bool lg_is_terminal(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                               uint32_t goal0, uint32_t goal1) {
    const PairAcc acc = pair_tests<L, FULL, true, false>(x, n_agents, cur0, cur1, 0u, 0u);
    const bool off_goal = ((FULL || x.v0) && cur0 != goal0) || ((FULL || x.v1) && cur1 != goal1);
    const uint64_t b_dup = group_bits<L>(__ballot(acc.dup == 0u), x.base);
    const uint64_t b_off = group_bits<L>(__ballot(off_goal), x.base);
    return (b_dup != 0) || (b_off == 0);
}

// ordered product over agents 0..A-1 of the sampled probabilities (ghosts hold 1.0)
template <int L, int K>
struct ProbChain {
    static __device__ __forceinline__ double run(const LaneCtx<L> &x, double q0, double q1, double p) {
        if constexpr (L >= 32) {
#pragma unroll 4
            for (int k = 0; k < L; ++k) {
                const double a = __shfl(q0, int(x.base) + k, 64), b = __shfl(q1, int(x.base) + k, 64);
                p = __dmul_rn(__dmul_rn(p, a), b);
            }
            return p;
        } else if constexpr (K < L) {
            const double a = group_bcast_f64<L, K>(q0, x), b = group_bcast_f64<L, K>(q1, x);
            return ProbChain<L, K + 1>::run(x, q0, q1, __dmul_rn(__dmul_rn(p, a), b));
        } else {
            return p;
        }
    }
};

struct EnvOut {
    double reward, prob;
    bool done, collision, was_terminal;
    bool next_terminal;          // is_terminal of the state step() returned (used by the rollout loop)
};

// One transition for the group's env.  cur0/cur1: my agents' cells (ghost slots hold 0).  Every lane of the
// group returns the same per-env results; next0/next1 are this lane's.  KNOWN_TERM: the caller already knows
// is_terminal(prev) (rollout carries it from step to step); otherwise it is derived here.
template <int L, bool FULL, bool EXT_UNIFORMS, bool KNOWN_TERM>
__device__ __forceinline__ void lg_transition(const EnvConsts &c, const uint64_t *__restrict__ mv,
                                              const SlipRow *lds_slip, const LaneCtx<L> &x, uint32_t n_agents,
                                              uint32_t cur0, uint32_t cur1, uint32_t goal0, uint32_t goal1,
                                              uint32_t act0_in, uint32_t act1_in, double u0, double u1,
                                              uint64_t env_id, uint64_t t, bool prev_terminal,
                                              uint32_t &next0, uint32_t &next1, EnvOut &out) {
    const uint32_t act0 = act0_in > 4u ? 0u : act0_in, act1 = act1_in > 4u ? 0u : act1_in;
    const bool v0 = FULL || x.v0, v1 = FULL || x.v1;

    // --- my two agents' moves (computed even if the env turns out terminal; discarded then)
    const uint64_t entry0 = move_entry(mv, c.n_cells, cur0, act0), entry1 = move_entry(mv, c.n_cells, cur1, act1);
    uint64_t mant0 = 0, mant1 = 0;
    if (!EXT_UNIFORMS && c.need_rng) {
        uint32_t w[4];
        const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | (x.g << 24);   // pair index = g
        philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3, c.seed_lo, c.seed_hi, w);
        mant0 = mantissa53(w[0], w[1]);
        mant1 = mantissa53(w[2], w[3]);
    }
    double q0, q1;
    slip_move<EXT_UNIFORMS>(lds_slip, entry0, mant0, u0, next0, q0);
    slip_move<EXT_UNIFORMS>(lds_slip, entry1, mant1, u1, next1, q1);
    if (!v0) { next0 = cur0; q0 = 1.0; }
    if (!v1) { next1 = cur1; q1 = 1.0; }

    // --- pair tests and per-env facts from wave ballots
    const PairAcc acc = pair_tests<L, FULL, !KNOWN_TERM, true>(x, n_agents, cur0, cur1, next0, next1);
    const bool off_goal_next = (v0 && next0 != goal0) || (v1 && next1 != goal1);
    const uint64_t b_vertex = group_bits<L>(__ballot(acc.vertex == 0u), x.base);
    const uint64_t b_swap = group_bits<L>(__ballot(acc.swap == 0u), x.base);
    const uint64_t b_off_next = group_bits<L>(__ballot(off_goal_next), x.base);
    bool was_terminal = prev_terminal;
    if (!KNOWN_TERM) {
        const bool off_goal = (v0 && cur0 != goal0) || (v1 && cur1 != goal1);
        const uint64_t b_dup = group_bits<L>(__ballot(acc.dup == 0u), x.base);
        const uint64_t b_off = group_bits<L>(__ballot(off_goal), x.base);
        was_terminal = (b_dup != 0) || (b_off == 0);
    }

    // --- total_prob: left-to-right product over agents 0..A-1 (ghosts contribute 1.0)
    const double p = ProbChain<L, 0>::run(x, q0, q1, 1.0);

    // _living_reward: mapf_env.py:436-446
    double living = c.r_living;
    if (c.criteria == 1u) {
        const bool st0 = v0 && cur0 == goal0 && act0 == 0u, st1 = v1 && cur1 == goal1 && act1 == 0u;
        const int stayed = __popcll(group_bits<L>(__ballot(st0), x.base)) + __popcll(group_bits<L>(__ballot(st1), x.base));
        living = __dmul_rn(double(int(n_agents) - stayed), c.r_living);
    }
    // calc_transition_reward_from_local_states: mapf_env.py:225-235 (collision before goal)
    const bool vertex = b_vertex != 0, coll = vertex || (b_swap != 0), goal_next = b_off_next == 0;
    out.was_terminal = was_terminal;
    if (was_terminal) {   // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0}), nothing drawn
        next0 = cur0; next1 = cur1;
        out.reward = 0.0; out.prob = 0.0; out.done = true; out.collision = false;
        out.next_terminal = true;
    } else {
        out.prob = p;
        out.collision = coll;
        out.done = coll || goal_next;
        out.reward = coll ? __dadd_rn(c.r_clash, living) : (goal_next ? __dadd_rn(c.r_goal, living) : living);
        out.next_terminal = vertex || goal_next;   // a swap leaves a non-terminal state (mapf_env.py:210-223)
    }
}

// ---- row access for a lane's two slots.  A even: one dword (cells) / one short (actions) per lane, fully
// coalesced: 4 B x 64 lanes.  `guard` = this lane really owns slot 0 (FULL kernels pass true for live lanes).
template <typename T>
__device__ __forceinline__ void load_pair(const T *base, uint64_t row, uint32_t n_agents, uint32_t g, bool v0, bool v1,
                                          uint32_t &a, uint32_t &b) {
    const T *p = base + row * n_agents + 2u * g;
    a = 0u; b = 0u;
    if ((n_agents & 1u) == 0u) {
        if (v0) {
            if (sizeof(T) == 2) { const uint32_t w = *reinterpret_cast<const uint32_t *>(p); a = w & 0xFFFFu; b = w >> 16; }
            else { const uint32_t w = *reinterpret_cast<const uint16_t *>(p); a = w & 0xFFu; b = w >> 8; }
        }
    } else {
        if (v0) a = p[0];
        if (v1) b = p[1];
    }
}

__device__ __forceinline__ void store_cells(uint16_t *base, uint64_t row, uint32_t n_agents, uint32_t g, bool v0, bool v1,
                                            uint32_t a, uint32_t b) {
    uint16_t *p = base + row * n_agents + 2u * g;
    if ((n_agents & 1u) == 0u) {
        if (v0) *reinterpret_cast<uint32_t *>(p) = a | (b << 16);
    } else {
        if (v0) p[0] = uint16_t(a);
        if (v1) p[1] = uint16_t(b);
    }
}

template <int L>
__device__ __forceinline__ LaneCtx<L> lane_ctx(uint32_t n_agents, uint64_t n_envs, bool &live) {
    LaneCtx<L> x;
    x.lane = threadIdx.x & 63u;
    x.g = x.lane & uint32_t(L - 1);
    x.base = x.lane & ~uint32_t(L - 1);
    const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    x.e = wave * uint64_t(64 / L) + (x.lane / uint32_t(L));
    // envs past the end keep their lanes alive (ballots / cross-lane moves are wave-wide) but own no agents
    live = x.e < n_envs;
    x.v0 = live && 2u * x.g < n_agents;
    x.v1 = live && 2u * x.g + 1u < n_agents;
    if (!live) x.e = 0;
    return x;
}

template <int L, bool FULL, bool EXT_UNIFORMS>
__global__ void __launch_bounds__(256) lg_step_kernel(const StepArgs p, const uint32_t n_agents) {
    __shared__ SlipRow slip[8];
    bool live;
    const LaneCtx<L> x = lane_ctx<L>(n_agents, p.n_envs, live);
    const uint64_t e = x.e;

    // rows first, LDS staging second: both sets of loads are in flight together
    uint32_t cur0, cur1, goal0, goal1, act0, act1;
    load_pair<uint16_t>(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    load_pair<uint16_t>(p.goal, p.goal_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, goal0, goal1);
    load_pair<uint8_t>(p.actions, e, n_agents, x.g, x.v0, x.v1, act0, act1);
    double u0 = 0.0, u1 = 0.0;
    if (EXT_UNIFORMS) {
        const double *up = p.uniforms + e * n_agents + 2u * x.g;
        if (x.v0) u0 = up[0];
        if (x.v1) u1 = up[1];
    }
    stage_slip_table(p.slip, slip);

    uint32_t next0, next1;
    EnvOut o;
    lg_transition<L, FULL, EXT_UNIFORMS, false>(p.c, p.mv, slip, x, n_agents, cur0, cur1, goal0, goal1, act0, act1,
                                                u0, u1, p.env_id_offset + e, p.t, false, next0, next1, o);
    if (!live) return;

    if (p.out_local) store_cells(p.out_local, e, n_agents, x.g, x.v0, x.v1, next0, next1);
    if (x.g == 0u) {
        if (p.out_reward) p.out_reward[e] = o.reward;
        if (p.out_prob) p.out_prob[e] = o.prob;
        if (p.out_done) p.out_done[e] = o.done ? 1 : 0;
        if (p.out_collision) p.out_collision[e] = o.collision ? 1 : 0;
        if (p.out_was_terminal) p.out_was_terminal[e] = o.was_terminal ? 1 : 0;
    }
    if (p.auto_reset && o.done) {
        uint32_t s0, s1;
        load_pair<uint16_t>(p.start, p.start_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, s0, s1);
        store_cells(p.state, e, n_agents, x.g, x.v0, x.v1, s0, s1);
    } else if (!o.was_terminal) {
        store_cells(p.state, e, n_agents, x.g, x.v0, x.v1, next0, next1);
    }
}

template <int L, bool FULL>
__global__ void __launch_bounds__(256) lg_rollout_kernel(const RolloutArgs p, const uint32_t n_agents) {
    __shared__ SlipRow slip[8];
    bool live;
    const LaneCtx<L> x = lane_ctx<L>(n_agents, p.n_envs, live);
    const uint64_t e = x.e;
    const bool leader = live && x.g == 0u;

    uint32_t cur0, cur1, goal0, goal1;
    load_pair<uint16_t>(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    load_pair<uint16_t>(p.goal, p.goal_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, goal0, goal1);
    stage_slip_table(p.slip, slip);

    // is_terminal is carried from step to step instead of re-deriving it from the cells every step
    bool terminal = lg_is_terminal<L, FULL>(x, n_agents, cur0, cur1, goal0, goal1);
    bool start_terminal = false;
    if (p.auto_reset) {
        uint32_t s0, s1;
        load_pair<uint16_t>(p.start, p.start_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, s0, s1);
        start_terminal = lg_is_terminal<L, FULL>(x, n_agents, s0, s1, goal0, goal1);
    }

    double ret = (p.accumulate && p.out_returns && leader) ? p.out_returns[e] : 0.0;
    uint32_t episodes = (p.accumulate && p.out_episodes && leader) ? p.out_episodes[e] : 0u;
    uint32_t collisions = (p.accumulate && p.out_collisions && leader) ? p.out_collisions[e] : 0u;
    const uint64_t env_id = p.env_id_offset + e;

    for (uint32_t s = 0; s < p.n_steps; ++s) {
        const uint64_t t = p.t + s;
        const uint64_t row = uint64_t(s) * p.n_envs + e;
        uint32_t act0, act1;
        if (p.actions) {
            load_pair<uint8_t>(p.actions, row, n_agents, x.g, x.v0, x.v1, act0, act1);
        } else {   // policy stream: one Philox call covers agents 4q..4q+3; this lane needs words 2(g&1), 2(g&1)+1
            uint32_t w[4];
            const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | ((x.g >> 1) << 24);
            philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3, p.c.pol_lo, p.c.pol_hi, w);
            const bool hi = (x.g & 1u) != 0u;
            act0 = __umulhi(hi ? w[2] : w[0], 5u);
            act1 = __umulhi(hi ? w[3] : w[1], 5u);
        }
        uint32_t next0, next1;
        EnvOut o;
        lg_transition<L, FULL, false, true>(p.c, p.mv, slip, x, n_agents, cur0, cur1, goal0, goal1, act0, act1, 0.0, 0.0,
                                            env_id, t, terminal, next0, next1, o);
        ret = __dadd_rn(ret, o.reward);
        episodes += o.done ? 1u : 0u;
        collisions += o.collision ? 1u : 0u;
        if (live) {
            if (p.rec_local) store_cells(p.rec_local, row, n_agents, x.g, x.v0, x.v1, next0, next1);
            if (leader) {
                if (p.rec_reward) p.rec_reward[row] = o.reward;
                if (p.rec_prob) p.rec_prob[row] = o.prob;
                if (p.rec_done) p.rec_done[row] = o.done ? 1 : 0;
                if (p.rec_collision) p.rec_collision[row] = o.collision ? 1 : 0;
            }
        }
        if (p.auto_reset && o.done) {
            load_pair<uint16_t>(p.start, p.start_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
            terminal = start_terminal;
        } else {
            cur0 = next0; cur1 = next1;
            terminal = o.next_terminal;
        }
    }
    if (!live) return;
    store_cells(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    if (leader) {
        if (p.out_returns) p.out_returns[e] = ret;
        if (p.out_episodes) p.out_episodes[e] = episodes;
        if (p.out_collisions) p.out_collisions[e] = collisions;
    }
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

// policy stream (oracle/philox.py random_actions_np): one thread per (row, group of 4 agents)
__global__ void __launch_bounds__(256) fill_actions_kernel(uint8_t *actions, EnvConsts c, uint64_t env_id_offset,
                                                           uint64_t n_envs, uint64_t t0, uint64_t n_rows,
                                                           uint32_t n_agents) {
    const uint32_t quads = (n_agents + 3u) / 4u;
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_rows * quads) return;
    const uint64_t row = i / quads;
    const uint32_t q = uint32_t(i - row * quads);
    const uint64_t s = row / n_envs, e = row - s * n_envs;
    const uint64_t env_id = env_id_offset + e, t = t0 + s;
    uint32_t w[4];
    const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | (q << 24);
    philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3, c.pol_lo, c.pol_hi, w);
    uint8_t *dst = actions + row * n_agents + 4u * q;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (4u * q + k < n_agents) dst[k] = uint8_t(__umulhi(w[k], 5u));
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
    const uint64_t n_rows = n_envs * n_steps;
    if (n_rows == 0) return hipSuccess;
    unsigned grid;
    if (hipError_t e = grid_1d(n_rows * uint64_t((n_agents + 3) / 4), 256, grid)) return e;
    hipLaunchKernelGGL(fill_actions_kernel, dim3(grid), dim3(256), 0, stream, actions, c, env_id_offset, n_envs, t0,
                       n_rows, uint32_t(n_agents));
    return hipGetLastError();
}

// ------------------------------------------------------------------- launchers
int lg_group_size(int n_agents) {
    int pairs = (n_agents + 1) / 2, L = 1;
    while (L < pairs) L <<= 1;
    return L;
}

static inline void lg_geometry(int L, uint64_t n_envs, unsigned &grid, unsigned &block) {
    const uint64_t threads = n_envs * uint64_t(L);
    block = threads <= (uint64_t(1) << 19) ? 64u : 256u;       // keep >= ~2 blocks per CU at small sizes
    const uint64_t per_block = block / unsigned(L);
    grid = unsigned((n_envs + per_block - 1) / per_block);
}

#define MAPF_FOR_EACH_L(X) X(1) X(2) X(4) X(8) X(16) X(32) X(64)

hipError_t launch_step_lg(int n_agents, const StepArgs &args, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    const int L = lg_group_size(n_agents);
    const bool full = n_agents == 2 * L;
    unsigned grid, block;
    lg_geometry(L, args.n_envs, grid, block);
    const uint32_t A = uint32_t(n_agents);
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

hipError_t launch_rollout_lg(int n_agents, const RolloutArgs &args, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    const int L = lg_group_size(n_agents);
    const bool full = n_agents == 2 * L;
    unsigned grid, block;
    lg_geometry(L, args.n_envs, grid, block);
    const uint32_t A = uint32_t(n_agents);
    switch (L) {
#define X(N)                                                                                                         \
    case N:                                                                                                          \
        if (full) hipLaunchKernelGGL((lg_rollout_kernel<N, true>), dim3(grid), dim3(block), 0, stream, args, A);     \
        else hipLaunchKernelGGL((lg_rollout_kernel<N, false>), dim3(grid), dim3(block), 0, stream, args, A);         \
        break;
        MAPF_FOR_EACH_L(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mapf
