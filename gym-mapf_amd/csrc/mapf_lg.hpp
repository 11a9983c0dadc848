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
// This header holds the device code shared by mapf_lg_kernels.hip (step + helpers) and mapf_lg_rollout.hip.
#pragma once
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"
#include <mutex>
#include <set>
#include <utility>

#ifdef MAPF_STAMPS   // diagnostic build only: per-segment cycle sums of the rollout loop (never shipped)
struct StampCtx { unsigned long long seg[8]; unsigned long long last; };
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); st.seg[i] += _t - st.last; st.last = _t; } while (0)
#define STAMP_PARAM , StampCtx &st
#define STAMP_ARG , st
#else
#define STAMP(i)
#define STAMP_PARAM
#define STAMP_ARG
#endif

namespace mapf {

template <int L>
struct LaneCtx {
    uint32_t lane, g, base;      // lane in wave, position in group, first lane of the group
    uint32_t e;                  // env index (local to the handle)
    bool v0, v1;                 // my two agent slots exist (2g < A, 2g+1 < A)
};

// ------------------------------------------------------------------ cross-lane moves inside a group
// Every control word used here (quad_perm, row_ror, row_mirror, row_half_mirror) reads a lane that exists
// and all lanes are active at the call sites, so no `old` value is needed: mov_dpp is a single v_mov_b32_dpp
// (update_dpp with old = 0 costs an extra v_mov per use).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return uint32_t(__builtin_amdgcn_mov_dpp(int(v), CTRL, 0xF, 0xF, true));
}

// The slip words of a lane that owns ONE agent pair (pair `g` of its env: the lane-group layout, the packed layout with two
// agents per lane) for the four-step block that contains step t, in step order (step_word).  The block's two calls belong
// to the QUAD, i.e. to this lane and its neighbour g ^ 1: the even lane computes the first call, the odd lane the second,
// and they trade the halves the other one needs (two DPP moves) -- one call per lane per four steps.  PAIRED = false (a
// group of one lane has no neighbour of the same env): both calls, in lockstep.  All lanes must be active.
template <bool PAIRED>
__device__ __forceinline__ Words4 pair_block_words(const EnvConsts &c, uint64_t env_id, uint64_t t, uint32_t g) {
    const uint64_t h0 = block_first_call(t);
    if constexpr (!PAIRED) {
        Words4 a, b;
        slip_words_x2(c, env_id, h0, g >> 1, h0 | 1u, g >> 1, a, b);
        return block_words(a, b, g & 1u);
    } else {
        const bool odd = (g & 1u) != 0u;
        const Words4 w = slip_words(c, env_id, h0 | uint64_t(g & 1u), g >> 1, 0u, 0u);
        const uint32_t r0 = dpp_mov<0xB1>(odd ? w.w0 : w.w1), r1 = dpp_mov<0xB1>(odd ? w.w2 : w.w3);   // quad_perm [1,0,3,2]
        return odd ? Words4{r0, r1, w.w1, w.w3} : Words4{w.w0, w.w2, r0, r1};
    }
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

// OR / sum of a per-lane word over my group, result in every lane.  Butterfly of DPP steps inside a 16-lane
// row (pairs, quads, half-row mirror, row mirror); ds_bpermute xor-partners beyond a row.
template <int L, bool ADD>
__device__ __forceinline__ uint32_t group_reduce(uint32_t v, const LaneCtx<L> &x) {
    auto comb = [](uint32_t a, uint32_t b) { return ADD ? a + b : (a | b); };
    if constexpr (L >= 2) v = comb(v, dpp_mov<0xB1>(v));           // quad_perm [1,0,3,2]
    if constexpr (L >= 4) v = comb(v, dpp_mov<0x4E>(v));           // quad_perm [2,3,0,1]
    if constexpr (L >= 8) v = comb(v, dpp_mov<0x141>(v));          // row_half_mirror
    if constexpr (L >= 16) v = comb(v, dpp_mov<0x140>(v));         // row_mirror
    if constexpr (L >= 32) v = comb(v, uint32_t(__shfl_xor(int(v), 16, 64)));
    if constexpr (L >= 64) v = comb(v, uint32_t(__shfl_xor(int(v), 32, 64)));
    return v;
}

// ------------------------------------------------------------------ pair tests
// min over agent pairs of xor (0 <=> equal).  dup: prev_i == prev_j (is_terminal, mapf_env.py:210-223);
// vertex: next_i == next_j; swap: prev_i == next_j and prev_j == next_i (mapf_env.py:378-389).
// PK = false: one 32-bit minimum per fact.  PK = true (every slot of the group is a real agent): each
// accumulator is TWO 16-bit minima, one per half-word, fed by packed xors of the lane's cell pair against the
// other lane's pair (straight and half-swapped) -- four agent pairs per v_pk_min_u16.
template <bool PK>
struct PairAcc {
    uint32_t dup = 0xFFFFFFFFu, vertex = 0xFFFFFFFFu, swap = 0xFFFFFFFFu;
    static __device__ __forceinline__ bool hit(uint32_t acc) {
        return PK ? ((acc & 0xFFFFu) == 0u || (acc >> 16) == 0u) : acc == 0u;
    }
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t swap_halves(uint32_t v) { return __builtin_amdgcn_alignbit(v, v, 16); }

// one rotation step, packed form: my cell pair (pk_prev, pk_next) against the pair of another group position
template <bool DUP, bool MOVES>
__device__ __forceinline__ void pair_apply_packed(uint32_t pk_prev, uint32_t pk_next, uint32_t o_prev, uint32_t o_next,
                                                  PairAcc<true> &acc) {
    const uint32_t o_prev_sw = swap_halves(o_prev);
    if (DUP) acc.dup = pk_min_u16(acc.dup, pk_min_u16(pk_prev ^ o_prev, pk_prev ^ o_prev_sw));
    if (MOVES) {
        const uint32_t o_next_sw = swap_halves(o_next);
        acc.vertex = pk_min_u16(acc.vertex, pk_min_u16(pk_next ^ o_next, pk_next ^ o_next_sw));
        // half h of (next ^ o_prev) | (prev ^ o_next) is zero <=> my agent h and the other lane's agent h swap
        const uint32_t same = (pk_next ^ o_prev) | (pk_prev ^ o_next);
        const uint32_t cross = (pk_next ^ o_prev_sw) | (pk_prev ^ o_next_sw);
        acc.swap = pk_min_u16(acc.swap, pk_min_u16(same, cross));
    }
}

// one rotation step: my two agents against the two agents of group position `og`, whose packed cells
// arrive in o_prev / o_next
template <int L, bool FULL, bool DUP, bool MOVES>
__device__ __forceinline__ void pair_apply(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                           uint32_t next0, uint32_t next1, uint32_t o_prev, uint32_t o_next,
                                           uint32_t og, PairAcc<FULL> &acc) {
    if constexpr (FULL) {
        pair_apply_packed<DUP, MOVES>(cur0 | (cur1 << 16), next0 | (next1 << 16), o_prev, o_next, acc);
    } else {
        const uint32_t op0 = o_prev & 0xFFFFu, op1 = o_prev >> 16;
        // ghost masks: 1 forces "different"
        const bool o0 = 2u * og < n_agents, o1 = 2u * og + 1u < n_agents;
        const uint32_t g0 = (x.v0 && o0) ? 0u : 1u, g1 = (x.v0 && o1) ? 0u : 1u;
        const uint32_t g2 = (x.v1 && o0) ? 0u : 1u, g3 = (x.v1 && o1) ? 0u : 1u;
        if (DUP) {
            acc.dup = min(acc.dup, min((cur0 ^ op0) | g0, (cur0 ^ op1) | g1));
            acc.dup = min(acc.dup, min((cur1 ^ op0) | g2, (cur1 ^ op1) | g3));
        }
        if (MOVES) {
            const uint32_t on0 = o_next & 0xFFFFu, on1 = o_next >> 16;
            const uint32_t fwd0 = cur0 | (next0 << 16), fwd1 = cur1 | (next1 << 16);
            // rev_k = next_k | prev_k << 16 of the other lane: one v_perm_b32 each from the two packed words
            const uint32_t rev0 = __builtin_amdgcn_perm(o_prev, o_next, 0x05040100u);
            const uint32_t rev1 = __builtin_amdgcn_perm(o_prev, o_next, 0x07060302u);
            acc.vertex = min(acc.vertex, min((next0 ^ on0) | g0, (next0 ^ on1) | g1));
            acc.vertex = min(acc.vertex, min((next1 ^ on0) | g2, (next1 ^ on1) | g3));
            acc.swap = min(acc.swap, min((fwd0 ^ rev0) | g0, (fwd0 ^ rev1) | g1));
            acc.swap = min(acc.swap, min((fwd1 ^ rev0) | g2, (fwd1 ^ rev1) | g3));
        }
    }
}

// rotations 1..L/2: unrolled with DPP moves for groups up to 16 lanes, a rolled ds_bpermute loop beyond
// (32 unrolled rounds would cost hundreds of registers for no gain)
template <int L, int S, bool FULL, bool DUP, bool MOVES, int LAST = L / 2>
struct PairRounds {
    static __device__ __forceinline__ void run(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                               uint32_t next0, uint32_t next1, uint32_t pk_prev, uint32_t pk_next,
                                               PairAcc<FULL> &acc) {
        if constexpr (L >= 32) {
#pragma unroll 2
            for (uint32_t s = 1; s <= uint32_t(L / 2); ++s) {
                const uint32_t og = (x.g + s) & uint32_t(L - 1);
                const int src = int(x.base + og);
                const uint32_t o_prev = uint32_t(__shfl(int(pk_prev), src, 64));
                const uint32_t o_next = MOVES ? uint32_t(__shfl(int(pk_next), src, 64)) : 0u;
                pair_apply<L, FULL, DUP, MOVES>(x, n_agents, cur0, cur1, next0, next1, o_prev, o_next, og, acc);
            }
        } else if constexpr (S <= LAST && L > 1) {
            const uint32_t o_prev = group_rot<L, S>(pk_prev, x);
            const uint32_t o_next = MOVES ? group_rot<L, S>(pk_next, x) : 0u;
            pair_apply<L, FULL, DUP, MOVES>(x, n_agents, cur0, cur1, next0, next1, o_prev, o_next,
                                            (x.g + uint32_t(S)) & uint32_t(L - 1), acc);
            PairRounds<L, S + 1, FULL, DUP, MOVES, LAST>::run(x, n_agents, cur0, cur1, next0, next1, pk_prev, pk_next, acc);
        }
    }
};

// all pairs of the env: my own two agents, then rotations 1..L/2 (every unordered lane pair is met)
template <int L, bool FULL, bool DUP, bool MOVES>
__device__ __forceinline__ PairAcc<FULL> pair_tests(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                                    uint32_t next0, uint32_t next1) {
    PairAcc<FULL> acc;
    const uint32_t pk_prev = cur0 | (cur1 << 16), pk_next = next0 | (next1 << 16);
    if constexpr (FULL) {   // my own pair: both halves carry the same test
        const uint32_t prev_sw = swap_halves(pk_prev), next_sw = swap_halves(pk_next);
        if (DUP) acc.dup = pk_prev ^ prev_sw;
        if (MOVES) {
            acc.vertex = pk_next ^ next_sw;
            acc.swap = (pk_next ^ prev_sw) | (pk_prev ^ next_sw);
        }
    } else {
        const uint32_t ghost = x.v1 ? 0u : 1u;
        if (DUP) acc.dup = (cur0 ^ cur1) | ghost;
        if (MOVES) {
            acc.vertex = (next0 ^ next1) | ghost;
            acc.swap = ((cur0 | (next0 << 16)) ^ (next1 | (cur1 << 16))) | ghost;
        }
    }
    if constexpr (FULL && L >= 2 && L <= 16) {
        // rotations 1 .. L/2-1 as usual; the half rotation pairs lane g with lane g + L/2 in BOTH directions, so the
        // two lanes split its four agent pairs: every lane offers its pair swapped if it sits in the lower half, the
        // receiver therefore sees a straight pair (lower half: tests 0-0', 1-1') or a swapped one (upper half: 0-1',
        // 1-0') and runs only the "same half-word" tests.
        PairRounds<L, 1, FULL, DUP, MOVES, L / 2 - 1>::run(x, n_agents, cur0, cur1, next0, next1, pk_prev, pk_next, acc);
        const bool lower = x.g < uint32_t(L / 2);
        const uint32_t o_prev = group_rot<L, L / 2>(lower ? swap_halves(pk_prev) : pk_prev, x);
        if (DUP) acc.dup = pk_min_u16(acc.dup, pk_prev ^ o_prev);
        if (MOVES) {
            const uint32_t o_next = group_rot<L, L / 2>(lower ? swap_halves(pk_next) : pk_next, x);
            acc.vertex = pk_min_u16(acc.vertex, pk_next ^ o_next);
            acc.swap = pk_min_u16(acc.swap, (pk_next ^ o_prev) | (pk_prev ^ o_next));
        }
    } else {
        PairRounds<L, 1, FULL, DUP, MOVES>::run(x, n_agents, cur0, cur1, next0, next1, pk_prev, pk_next, acc);
    }
    return acc;
}

// MapfEnv.is_terminal (mapf_env.py:210-223) of the group's env
template <int L, bool FULL>
__device__ __forceinline__ bool lg_is_terminal(const LaneCtx<L> &x, uint32_t n_agents, uint32_t cur0, uint32_t cur1,
                                               uint32_t goal0, uint32_t goal1) {
    const PairAcc<FULL> acc = pair_tests<L, FULL, true, false>(x, n_agents, cur0, cur1, 0u, 0u);
    const bool off_goal = ((FULL || x.v0) && cur0 != goal0) || ((FULL || x.v1) && cur1 != goal1);
    const uint32_t flags = group_reduce<L, false>((PairAcc<FULL>::hit(acc.dup) ? 1u : 0u) | (off_goal ? 2u : 0u), x);
    return (flags & 1u) != 0u || (flags & 2u) == 0u;
}

// ordered product over agents 0..A-1 of the sampled probabilities (ghosts hold 1.0).  Groups inside a quad hand
// the running product from lane k-1 to lane k (k rotations of one double); larger groups broadcast every lane's two
// factors and multiply redundantly.  Either way the multiplications happen in agent order, so the result rounds
// exactly like the reference's `total_prob *= p` (mapf_env.py:257).  The product ends up in every lane.
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

// value held by the previous lane of my group (meaningless in the group's first lane)
template <int L>
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v) {
    if constexpr (L == 2) return dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]
    else if constexpr (L == 4) return dpp_mov<0x93>(v);            // quad_perm [3,0,1,2]
    else return dpp_mov<0x121>(v);                                 // row_ror:1 -- lane i <- lane i-1 of the 16-lane row
}

template <int L>
__device__ __forceinline__ double prob_product(const LaneCtx<L> &x, double q0, double q1) {
    if constexpr (L == 1) {
        return __dmul_rn(__dmul_rn(1.0, q0), q1);
    } else if constexpr (L <= 16) {
        // stage k: every lane continues the product it receives from the lane before it.  Lane k's value is the
        // true prefix after stage k (by induction from lane 0's stage-0 value; what other lanes hold at that point
        // is never read by a lane that matters), so after stage L-1 lane L-1 holds the total -- no selects.
        double run = __dmul_rn(__dmul_rn(1.0, q0), q1);              // correct in lane 0
#pragma unroll
        for (int k = 1; k < L; ++k) {
            const uint32_t lo = from_prev_lane<L>(uint32_t(__double2loint(run)));
            const uint32_t hi = from_prev_lane<L>(uint32_t(__double2hiint(run)));
            run = __dmul_rn(__dmul_rn(__hiloint2double(int(hi), int(lo)), q0), q1);
        }
        return run;   // the total sits in lane L-1 only (the lane that stores it)
    } else {
        return ProbChain<L, 0>::run(x, q0, q1, 1.0);
    }
}

struct EnvOut {
    double reward;
    double prob;                 // valid in the group's LAST lane (g == L-1), the one the product chain ends in
    uint32_t status;             // one 0/1 fact per BYTE (sub-dword operand selects read them for free):
                                 // byte 0 done, byte 1 collision, byte 2 is_terminal of the returned state
    bool was_terminal;
    __device__ __forceinline__ bool done() const { return (status & 0xFFu) != 0u; }
    __device__ __forceinline__ bool collision() const { return (status & 0xFF00u) != 0u; }
};

__device__ __forceinline__ void stage_outcome_rows(const EnvConsts &c, OutcomeRow *lds, const uint32_t i) {   // lanes i = 0..15 write
    if (i < 16u) {
        const uint32_t st = outcome_status(i & 7u);
        const double r = (st & 0x100u) ? __dadd_rn(c.r_clash, c.r_living) : ((st & 1u) ? __dadd_rn(c.r_goal, c.r_living) : c.r_living);
        lds[i].reward = i < 8u ? r : 0.0;
        lds[i].status = i < 8u ? st : kTerminalStatus;
        lds[i].pad = (lds[i].status & 1u) | ((lds[i].status & 0x100u) << 8);   // done | collision << 16: summed as two 16-bit counts
    }
}
__device__ __forceinline__ void stage_outcome_table(const EnvConsts &c, OutcomeRow *lds) {   // before a __syncthreads()
    stage_outcome_rows(c, lds, threadIdx.x);
}

// One transition for the group's env.  cur0/cur1: my agents' cells (ghost slots hold 0).  Every lane of the
// group returns the same per-env results; next0/next1 are this lane's.  KNOWN_TERM: the caller already knows
// is_terminal(prev) (rollout carries it from step to step); otherwise it is derived here.
template <int L, bool FULL, bool EXT_UNIFORMS, bool KNOWN_TERM, bool MV_IN_LDS = false, bool OUTCOME_LDS = false,
          bool Q_FROM_MEMBERS = false>
__device__ __forceinline__ void lg_transition(const EnvConsts &c, const MoveEntry *__restrict__ mv,
                                              const SlipRow *lds_slip, const OutcomeRow *lds_outcome,
                                              const LaneCtx<L> &x, uint32_t n_agents,
                                              uint32_t cur0, uint32_t cur1, uint32_t goal0, uint32_t goal1,
                                              uint32_t act0_in, uint32_t act1_in, double u0, double u1,
                                              uint64_t env_id, uint64_t t, const uint32_t word, bool prev_terminal,
                                              uint32_t &next0, uint32_t &next1, EnvOut &out STAMP_PARAM) {
    const uint32_t act0 = act0_in > 4u ? 0u : act0_in, act1 = act1_in > 4u ? 0u : act1_in;
    const bool v0 = FULL || x.v0, v1 = FULL || x.v1;

    // --- my two agents' moves (computed even if the env turns out terminal; discarded then)
    const MoveEntry entry0 = move_entry<!MV_IN_LDS>(mv, c.n_cells, cur0, act0);
    const MoveEntry entry1 = move_entry<!MV_IN_LDS>(mv, c.n_cells, cur1, act1);
    double q0, q1;
    if (EXT_UNIFORMS) {
        slip_move<true>(lds_slip, entry0, 0, u0, next0, q0);
        slip_move<true>(lds_slip, entry1, 0, u1, next1, q1);
    } else {
        // `word` = this step's slip word of my pair: low half for agent 2g, high half for agent 2g+1
        const uint32_t hi0 = word & 0xFFFFu, hi1 = word >> 16;
        STAMP(1);   // philox + gather issue
        uint32_t tie0, tie1;
        if (Q_FROM_MEMBERS) {
            slip_move_hi_members(c, entry0, hi0, next0, q0, tie0);
            slip_move_hi_members(c, entry1, hi1, next1, q1, tie1);
        } else {
            slip_move_hi(lds_slip, entry0, hi0, next0, q0, tie0);
            slip_move_hi(lds_slip, entry1, hi1, next1, q1, tie1);
        }
        if (__builtin_expect(__any(min(tie0, tie1) == 0u && c.need_rng), 0)) {
            // a top-16-bit tie somewhere in the wave (~2^-8 of wave-steps): redo with all 53 bits
            slip_move<false>(lds_slip, entry0, refine_mantissa(c, env_id, t, 2u * x.g, hi0), 0.0, next0, q0);
            slip_move<false>(lds_slip, entry1, refine_mantissa(c, env_id, t, 2u * x.g + 1u, hi1), 0.0, next1, q1);
        }
    }
    if (!v0) { next0 = cur0; q0 = 1.0; }
    if (!v1) { next1 = cur1; q1 = 1.0; }

    STAMP(2);   // slip_move (gather wait, LDS rows, sampling)
    // --- pair tests, then per-env facts: one flag word per lane, OR-reduced over the group
    const PairAcc<FULL> acc = pair_tests<L, FULL, !KNOWN_TERM, true>(x, n_agents, cur0, cur1, next0, next1);
    STAMP(3);   // pair tests
    // (full groups: both cells in one compare of the packed pairs)
    const bool off_goal_next = FULL ? (next0 | (next1 << 16)) != (goal0 | (goal1 << 16))
                                    : (v0 && next0 != goal0) || (v1 && next1 != goal1);
    uint32_t flags = (PairAcc<FULL>::hit(acc.vertex) ? 1u : 0u) | (PairAcc<FULL>::hit(acc.swap) ? 2u : 0u) | (off_goal_next ? 4u : 0u);
    if (!KNOWN_TERM) {
        const bool off_goal = (v0 && cur0 != goal0) || (v1 && cur1 != goal1);
        flags |= (PairAcc<FULL>::hit(acc.dup) ? 8u : 0u) | (off_goal ? 16u : 0u);
    }
    flags = group_reduce<L, false>(flags, x);
    bool was_terminal = prev_terminal;
    if (!KNOWN_TERM) was_terminal = (flags & 8u) != 0u || (flags & 16u) == 0u;

    STAMP(4);   // flags + group reduce
    // --- total_prob: left-to-right product over agents 0..A-1 (ghosts contribute 1.0)
    const double p = prob_product<L>(x, q0, q1);

    STAMP(5);   // prob chain
    const uint32_t f = flags & 7u;
    out.was_terminal = was_terminal;
    if (OUTCOME_LDS && c.criteria == 0u) {
        const OutcomeRow row = lds_outcome[f | (was_terminal ? 8u : 0u)];
        out.reward = row.reward;
        out.status = row.status;
    } else {
        // _living_reward: mapf_env.py:436-446
        double living = c.r_living;
        if (c.criteria == 1u) {
            const uint32_t mine = ((v0 && cur0 == goal0 && act0 == 0u) ? 1u : 0u) + ((v1 && cur1 == goal1 && act1 == 0u) ? 1u : 0u);
            const int stayed = int(group_reduce<L, true>(mine, x));
            living = __dmul_rn(double(int(n_agents) - stayed), c.r_living);
        }
        // calc_transition_reward_from_local_states: mapf_env.py:225-235 (collision before goal)
        const bool vertex = (f & 1u) != 0u, coll = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
        const uint32_t st = ((coll || goal_next) ? 1u : 0u) | (coll ? 0x100u : 0u) | ((vertex || goal_next) ? 0x10000u : 0u);
        const double r = coll ? __dadd_rn(c.r_clash, living) : (goal_next ? __dadd_rn(c.r_goal, living) : living);
        out.reward = was_terminal ? 0.0 : r;           // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0}), nothing drawn
        out.status = was_terminal ? kTerminalStatus : st;
    }
    out.prob = was_terminal ? 0.0 : p;
    if (was_terminal) { next0 = cur0; next1 = cur1; }
}

// ---- memory access.  All element indices are 32-bit and turned into 32-bit BYTE offsets from a uniform base
// pointer, so every access uses the SGPR-base + VGPR-offset addressing form (no 64-bit address arithmetic per
// lane).  The C ABI rejects calls whose largest array would exceed 4 GiB (mapf_capi.hip: check_extent).
template <typename T>
__device__ __forceinline__ const T *at(const T *base, uint32_t index) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + index * uint32_t(sizeof(T)));
}
template <typename T>
__device__ __forceinline__ T *at(T *base, uint32_t index) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + index * uint32_t(sizeof(T)));
}

// A lane's two slots of row `row`.  A even: one dword (cells) / one short (actions) per lane, fully coalesced
// (4 B x 64 lanes).
template <typename T>
__device__ __forceinline__ void load_pair(const T *base, uint32_t row, uint32_t n_agents, uint32_t g, bool v0, bool v1,
                                          uint32_t &a, uint32_t &b) {
    const T *p = at(base, row * n_agents + 2u * g);
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

template <bool EVEN = false>
__device__ __forceinline__ void store_cells(uint16_t *base, uint32_t row, uint32_t n_agents, uint32_t g, bool v0, bool v1,
                                            uint32_t a, uint32_t b) {
    uint16_t *p = at(base, row * n_agents + 2u * g);
    if (EVEN || (n_agents & 1u) == 0u) {
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
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    x.e = wave * uint32_t(64 / L) + (x.lane / uint32_t(L));
    // envs past the end keep their lanes alive (cross-lane moves are wave-wide) but own no agents
    live = x.e < uint32_t(n_envs);
    x.v0 = live && 2u * x.g < n_agents;
    x.v1 = live && 2u * x.g + 1u < n_agents;
    if (!live) x.e = 0;
    return x;
}

// Dynamic LDS beyond the 32 KB default needs an explicit opt-in per (device, kernel).  The driver call is made once per
// pair and remembered: it would otherwise sit in the enqueue path of every launch (and of every launch a bench times).
static inline hipError_t allow_large_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) return e;
    done.insert({dev, kernel});
    return hipSuccess;
}

static inline void lg_geometry(int L, uint64_t n_envs, unsigned &grid, unsigned &block) {
    const uint64_t threads = n_envs * uint64_t(L);
    // one-wave blocks keep >= ~2 blocks per CU at small sizes; from two waves per SIMD upward four-wave blocks launch
    // faster (measured at 65536 envs x 8 agents: 5.29 us per step instead of 5.67)
    block = threads < (uint64_t(1) << 17) ? 64u : 256u;
    const uint64_t per_block = block / unsigned(L);
    grid = unsigned((n_envs + per_block - 1) / per_block);
}

#define MAPF_FOR_EACH_L(X) X(1) X(2) X(4) X(8) X(16) X(32) X(64)


}  // namespace mapf
