// Device helpers shared by the packed-layout kernels (mapf_lq_rollout.hip: fused rollout, mapf_lq_step.hip: single
// step): K = 2, 4 or 8 agents per lane as P = K/2 packed 2 x u16 registers, Q = A/K lanes per env.
#pragma once
#include "mapf_lg.hpp"

#pragma clang diagnostic ignored "-Wunused-function"   // (each including file uses its own subset of these helpers)

namespace mapf {
namespace {

using gf64 = __attribute__((address_space(1))) double *;
using gu32 = __attribute__((address_space(1))) uint32_t *;
using gu16 = __attribute__((address_space(1))) uint16_t *;
using gu8 = __attribute__((address_space(1))) uint8_t *;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// A pointer the kernel KNOWS to name device memory, typed so: pointers that reach a kernel through a re-read argument block
// (lq_step_kernel<BIG>) are generic to the compiler, which then emits FLAT loads / stores -- and with a flat operation
// pending every wait becomes s_waitcnt vmcnt(0), because flat operations may complete out of order.
#define MAPF_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ MAPF_GLOBAL T *as_global(T *p) { return (MAPF_GLOBAL T *)p; }
template <typename T> __device__ __forceinline__ MAPF_GLOBAL T *gat(T *base, uint32_t index) {   // at() of a pointer that is global
    return (MAPF_GLOBAL T *)((MAPF_GLOBAL char *)as_global(base) + index * uint32_t(sizeof(T)));
}

// P packed dwords (2P cells) of a lane, moved as one global access
template <int P> struct Packed;
template <> struct Packed<1> {
    uint32_t v[1];
    static __device__ __forceinline__ Packed load(const void *p) { return Packed{{*reinterpret_cast<const uint32_t *>(p)}}; }
    static __device__ __forceinline__ Packed load(const MAPF_GLOBAL uint16_t *p) { return Packed{{*(const MAPF_GLOBAL uint32_t *)p}}; }
    __device__ __forceinline__ void store(void *p) const { *reinterpret_cast<uint32_t *>(p) = v[0]; }
    __device__ __forceinline__ void store_global(gu16 p) const { *(gu32)p = v[0]; }
};
template <> struct Packed<4> {
    uint32_t v[4];
    static __device__ __forceinline__ Packed load(const void *p) {
        const uint4 w = *reinterpret_cast<const uint4 *>(p);
        return Packed{{w.x, w.y, w.z, w.w}};
    }
    static __device__ __forceinline__ Packed load(const MAPF_GLOBAL uint16_t *p) {
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ w = *(const MAPF_GLOBAL u32x4_ *)p;
        return Packed{{w.x, w.y, w.z, w.w}};
    }
    __device__ __forceinline__ void store(void *p) const { *reinterpret_cast<uint4 *>(p) = make_uint4(v[0], v[1], v[2], v[3]); }
    __device__ __forceinline__ void store_global(gu16 p) const {
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        *(__attribute__((address_space(1))) u32x4_ *)p = u32x4_{v[0], v[1], v[2], v[3]};
    }
};
template <> struct Packed<2> {
    uint32_t v[2];
    static __device__ __forceinline__ Packed load(const void *p) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(p);
        return Packed{{w.x, w.y}};
    }
    static __device__ __forceinline__ Packed load(const MAPF_GLOBAL uint16_t *p) {
        const u32x2 w = *(const MAPF_GLOBAL u32x2 *)p;
        return Packed{{w.x, w.y}};
    }
    __device__ __forceinline__ void store(void *p) const { *reinterpret_cast<u32x2 *>(p) = u32x2{v[0], v[1]}; }
    __device__ __forceinline__ void store_global(gu16 p) const {
        *(__attribute__((address_space(1))) u32x2 *)p = u32x2{v[0], v[1]};
    }
};

// An LDS location is named by its BYTE OFFSET inside the kernel's LDS image: plain integer arithmetic ending in one
// v_lshl_add_u32, the constant part folded into the ds instruction's offset field.  Two kinds of image:
//   * LdsObject -- a static __shared__ object (the plain single step's 1 KB table image): every access goes THROUGH the
//     object, whose address the compiler knows and folds;
//   * LdsAbsolute -- the kernel's dynamic LDS segment used as a raw scratchpad that starts at LDS address 0 (the kernel
//     has no static LDS object: checked against the compiled code objects by tests/test_cabi_and_host.py): EVERY access,
//     staging stores included, is an integer address turned into an LDS pointer.  (Going through the `extern __shared__`
//     object costs one vector add per access -- its address is resolved after instruction selection -- and mixing the two
//     forms is what rounds 2-3 did: stores through the object, loads from integer addresses, i.e. dead stores in the
//     optimiser's eyes, which a build without -fPIC indeed deleted.)
using lds_ptr = __attribute__((address_space(3))) unsigned char *;
struct LdsAbsolute {};
struct LdsObject { lds_ptr base; };
__device__ __forceinline__ lds_ptr lds_addr(LdsAbsolute, uint32_t byte_offset) { return (lds_ptr)(uintptr_t(byte_offset)); }
__device__ __forceinline__ lds_ptr lds_addr(LdsObject image, uint32_t byte_offset) { return image.base + byte_offset; }
template <typename T, typename Image>
__device__ __forceinline__ T lds_at(Image image, uint32_t byte_offset) {
    return *(const __attribute__((address_space(3))) T *)lds_addr(image, byte_offset);
}
// a generic pointer to the image's bytes from `byte_offset` on, for staging code written against ordinary pointers
template <typename T, typename Image>
__device__ __forceinline__ T *lds_generic(Image image, uint32_t byte_offset) {
    return (T *)lds_addr(image, byte_offset);
}
// half-word H of x times a wave-uniform factor: ONE v_mul_u32_u24 with a word select (the compiler prefers to extract
// the half-word first and fold the multiply into a v_mad: one instruction more per agent)
template <int H>
__device__ __forceinline__ uint32_t half_times(uint32_t x, uint32_t factor) {
    uint32_t r;
    if (H == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(factor));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(factor));
    return r;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename Image>
__device__ __forceinline__ MoveEntry lds_entry_at(Image image, uint32_t byte_offset) {   // one ds_read_b128
    const u32x4 v = lds_at<u32x4>(image, byte_offset);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- packed slot sampling of the fused rollout: both threshold compares of an agent in one packed subtract.
// The 16-bit uniform and the entry's two thresholds are biased by 0x8000 (unsigned order -> signed order), so that a
// SATURATING signed 16-bit subtract keeps the sign of the true difference and is zero exactly on a tie:
//   d = sat(h - t) per half-word;  d < 0 <=> hi < th_k (slot not passed);  d == 0 <=> tie (exact 53-bit redo)
// sign = d >> 15 is 0 / -1 per half; slot = 2 + sign0 + sign1 turns into the probability's LDS address and into the
// v_perm selector of the slot's cell by one dot product each (v_dot2_i32_i16 with (8, 8) resp. (0x0202, 0x0202)).
typedef short s16x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kHalfBias = 0x80008000u;
__device__ __forceinline__ uint32_t pk_sub_sat_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sign_i16(uint32_t a) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) >> 15); }
// a.lo * b.lo + a.hi * b.hi + c (signed halves), all three operands in vector registers: the two-operand form the
// compiler prefers accumulates into its destination, which would cost a v_mov of `c` per use
__device__ __forceinline__ uint32_t dot2_i16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// one agent: `h` = its biased uniform in both half-words, `entry.z` = its biased thresholds.  Returns the packed
// difference (tie <=> a zero half); the slot's probability address (relative to the slip rows) and cell selector go out
__device__ __forceinline__ uint32_t sample_slot_packed(const MoveEntry &entry, uint32_t h, uint32_t eights, uint32_t steps,
                                                       uint32_t sel_base, uint32_t &q_at, uint32_t &cell) {
    const uint32_t d = pk_sub_sat_i16(h, entry.z);
    const uint32_t sign = pk_sign_i16(d);
    q_at = dot2_i16(sign, eights, entry_row_offset(entry));       // row + 8 * (sign0 + sign1); slot 2's address is 16 further
    cell = __builtin_amdgcn_perm(entry.y, entry.x, dot2_i16(sign, steps, sel_base));
    return d;
}

// The same for a 4-byte DELTA row (bytes 0..2: the three candidates' cell ids minus the row's own cell, signed; byte 3: the slip
// row's biased offset / 16): the slot selects a BYTE, and the sampled cell is my current cell -- half-word H of `packed_cells` --
// plus the sign-extended delta (one add with a word select on one operand and a sign-extending byte select on the other).
template <int H>
__device__ __forceinline__ uint32_t sample_slot_delta(uint32_t row, uint32_t th, uint32_t row_off, uint32_t h, uint32_t eights, uint32_t ones,
                                                      uint32_t byte_base, uint32_t packed_cells, uint32_t &q_at, uint32_t &cell) {
    const uint32_t d = pk_sub_sat_i16(h, th);
    const uint32_t sign = pk_sign_i16(d);
    q_at = dot2_i16(sign, eights, row_off);
    const uint32_t delta = __builtin_amdgcn_perm(row, row, dot2_i16(sign, ones, byte_base));   // {byte `slot` of the row, 0, 0, 0}
    if (H == 0) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "=v"(cell) : "v"(packed_cells), "v"(delta));
    else asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0" : "=v"(cell) : "v"(packed_cells), "v"(delta));
    return d;
}

// "same half-word" tests only: my pair against a pair that arrives straight or half-swapped (half rotation)
template <bool DUP, bool MOVES>
__device__ __forceinline__ void pair_apply_same(uint32_t pk_prev, uint32_t pk_next, uint32_t o_prev, uint32_t o_next,
                                                PairAcc<true> &acc) {
    if (DUP) acc.dup = pk_min_u16(acc.dup, pk_prev ^ o_prev);
    if (MOVES) {
        acc.vertex = pk_min_u16(acc.vertex, pk_next ^ o_next);
        acc.swap = pk_min_u16(acc.swap, (pk_next ^ o_prev) | (pk_prev ^ o_next));
    }
}

// value held by lane (g + S) mod Q of my group.  Groups of 2, 4 and 16 lanes rotate with one DPP move; a group of 8
// would need two DPP moves and a select per value (a 16-lane row holds two groups), so it goes through the LDS
// crossbar instead: one ds_bpermute_b32 per value, all of a round's values in flight together
template <int Q, int S>
__device__ __forceinline__ uint32_t packed_rot(uint32_t v, const LaneCtx<Q> &x) {
    if constexpr (Q == 8) return uint32_t(__builtin_amdgcn_ds_bpermute(int((x.base + ((x.g + uint32_t(S)) & 7u)) << 2), int(v)));
    else return group_rot<Q, S>(v, x);
}

// full rotations 1 .. Q/2-1: my pairs against every pair of group position g + S
template <int Q, int P, int S, bool DUP, bool MOVES>
struct PackedRounds {
    static __device__ __forceinline__ void run(const LaneCtx<Q> &x, const uint32_t (&c)[P], const uint32_t (&n)[P],
                                               PairAcc<true> &acc) {
        if constexpr (S <= Q / 2 - 1) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const uint32_t oc = packed_rot<Q, S>(c[j], x);
                const uint32_t on = MOVES ? packed_rot<Q, S>(n[j], x) : 0u;
#pragma unroll
                for (int i = 0; i < P; ++i) pair_apply_packed<DUP, MOVES>(c[i], n[i], oc, on, acc);
            }
            PackedRounds<Q, P, S + 1, DUP, MOVES>::run(x, c, n, acc);
        }
    }
};

// all agent pairs of the env.  c: my packed current cells (agents K*g .. K*g+K-1, two per dword), n: next cells.
template <int Q, int P, bool DUP, bool MOVES>
__device__ __forceinline__ PairAcc<true> packed_pair_tests(const LaneCtx<Q> &x, const uint32_t (&c)[P], const uint32_t (&n)[P]) {
    PairAcc<true> acc;
    uint32_t csw[P], nsw[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        csw[i] = swap_halves(c[i]);
        nsw[i] = MOVES ? swap_halves(n[i]) : 0u;
    }
    // inside each pair (both half-words carry the same test) ...
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const uint32_t d = c[i] ^ csw[i], v = n[i] ^ nsw[i], w = (n[i] ^ csw[i]) | (c[i] ^ nsw[i]);
        if (DUP) acc.dup = i == 0 ? d : pk_min_u16(acc.dup, d);
        if (MOVES) {
            acc.vertex = i == 0 ? v : pk_min_u16(acc.vertex, v);
            acc.swap = i == 0 ? w : pk_min_u16(acc.swap, w);
        }
    }
    // ... then pair against pair inside the lane
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = i + 1; j < P; ++j) pair_apply_packed<DUP, MOVES>(c[i], n[i], c[j], n[j], acc);
    if constexpr (Q >= 2) {
        PackedRounds<Q, P, 1, DUP, MOVES>::run(x, c, n, acc);
        // half rotation: lane g meets lane g + Q/2 from both sides, so the two lanes split the agent pairs --
        // lower-half lanes offer their pairs half-swapped; whoever receives runs only the same-half-word tests
        const bool lower = x.g < uint32_t(Q / 2);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const uint32_t oc = packed_rot<Q, Q / 2>(lower ? csw[j] : c[j], x);
            const uint32_t on = MOVES ? packed_rot<Q, Q / 2>(lower ? nsw[j] : n[j], x) : 0u;
#pragma unroll
            for (int i = 0; i < P; ++i) pair_apply_same<DUP, MOVES>(c[i], n[i], oc, on, acc);
        }
    }
    return acc;
}

// ---- vertex / swap facts of an env through a per-env LDS occupancy bitmap instead of all agent pairs (32 agents: 496
// pairs are ~3/4 of a step's vector instructions; this is O(A)).  ONE bit per cell -- "an agent ends here" -- and three
// LDS operations per agent (rounds 3-4 kept a second bit, "an agent stood here when the step began": four operations and a
// bitmap twice the size).  The env's lanes sit in ONE wave and a wave's LDS operations execute in program order, so no
// barrier is needed between the phases:
//   1. every agent sets the bit of its NEXT cell with a returning atomic OR and looks at the old word:
//      bit already set  <=>  two agents end in one cell: the vertex collision (mapf_env.py:386-387);
//   2. every agent READS the word of its CURRENT cell: bit set and next != current  <=>  another agent ends on the cell it
//      leaves (current cells of a non-terminal state are distinct) -- both halves of a swap (:382-384) see that, so they
//      are the "candidates";
//   3. the touched words (next cells) are cleared;
//   4. a swap needs TWO candidates in one env (each partner moves onto the other's cell), which is rare (an agent in a
//      hundred is a candidate on the 32-agent maps): only while some env of the wave has two does each group elect its
//      lowest candidate, broadcast its (current, next) pair, and every lane compare its own agents' (next, current)
//      against it -- equal <=> the two swap.
// min over my group (result in every lane); Q <= 16 lanes
template <int Q>
__device__ __forceinline__ uint32_t group_reduce_min(uint32_t v) {
    if constexpr (Q >= 2) v = min(v, dpp_mov<0xB1>(v));            // quad_perm [1,0,3,2]
    if constexpr (Q >= 4) v = min(v, dpp_mov<0x4E>(v));            // quad_perm [2,3,0,1]
    if constexpr (Q >= 8) v = min(v, dpp_mov<0x141>(v));           // row_half_mirror
    if constexpr (Q >= 16) v = min(v, dpp_mov<0x140>(v));          // row_mirror
    return v;
}
using lds_u32 = __attribute__((address_space(3))) uint32_t *;
template <int Q, int K, typename Image>
__device__ __forceinline__ PairAcc<true> bitmap_pair_tests(const LaneCtx<Q> &x, Image image, uint32_t bitmap_at, const uint32_t (&c)[K / 2],
                                                           const uint32_t (&n)[K / 2]) {
    static_assert(Q <= 16 && K <= 8, "a group is at most one 16-lane row");
    // cell -> word (cell >> 5) * 4 bytes, bit cell & 31: shifts and bit-field extracts take the low five bits of their
    // count register by themselves, so an even agent's count is its packed register as it is
    uint32_t cnt_c[K], cnt_n[K], wc[K], wn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        cnt_c[k] = (k & 1) ? c[k / 2] >> 16 : c[k / 2];             // (even agents: the odd agent's cell rides in the high half)
        cnt_n[k] = (k & 1) ? n[k / 2] >> 16 : n[k / 2];
        wc[k] = bitmap_at + ((c[k / 2] >> ((k & 1) ? 19 : 3)) & 0x1FFCu);
        wn[k] = bitmap_at + ((n[k / 2] >> ((k & 1) ? 19 : 3)) & 0x1FFCu);
    }
    uint32_t old[K], seen[K];
#pragma unroll
    for (int k = 0; k < K; ++k) old[k] = __hip_atomic_fetch_or((lds_u32)lds_addr(image, wn[k]), 1u << (cnt_n[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int k = 0; k < K; ++k) seen[k] = __hip_atomic_load((lds_u32)lds_addr(image, wc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int k = 0; k < K; ++k) __hip_atomic_store((lds_u32)lds_addr(image, wn[k]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // (v_bfe_u32 reads bits 4:0 of its offset operand, so the count registers go in unmasked)
    // candidate k: the bit of my current cell is set and I moved (half-word k of c ^ n is not zero: a v_min with a word
    // select, written out -- the optimiser turns min(x, 1) into compare + select through a scalar mask)
    uint32_t is_cand[K], n_mine = 0u, vertex_hit = 0u, one = 1u;
    asm volatile("" : "+v"(one));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t diff = c[k / 2] ^ n[k / 2];
        uint32_t moved;
        if (k & 1) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(moved) : "v"(diff), "v"(one));
        else asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(moved) : "v"(diff), "v"(one));
        is_cand[k] = __builtin_amdgcn_ubfe(seen[k], cnt_c[k], 1u) & moved;
        n_mine += is_cand[k];
        vertex_hit |= __builtin_amdgcn_ubfe(old[k], cnt_n[k], 1u);
    }
    uint32_t swap_hit = 0u;
    const uint32_t n_cand = group_reduce<Q, true>(n_mine, x);
    if (__builtin_expect(__any(n_cand >= 2u), 0)) {
        uint32_t cand = 0u, cur[K], nxt[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            cand |= is_cand[k] << k;
            cur[k] = cnt_c[k] & 0xFFFFu;
            nxt[k] = cnt_n[k] & 0xFFFFu;
        }
        while (__any(cand != 0u)) {
            const uint32_t k_low = uint32_t(__builtin_ctz(cand | 0x100u));              // my lowest candidate slot (8 = none)
            const uint32_t key = cand ? (x.g << 3) | k_low : 0xFFFFu;
            const uint32_t best = group_reduce_min<Q>(key);                              // the group's elected (lane, slot)
            const bool owner = best == key && cand != 0u;
            uint32_t mine = 0u;                                                           // the elected agent's current | next << 16
#pragma unroll
            for (int k = 0; k < K; ++k) mine = (owner && k_low == uint32_t(k)) ? (cur[k] | (nxt[k] << 16)) : mine;
            const uint32_t fwd = group_reduce<Q, false>(mine, x);                        // (every other lane contributes 0)
            const uint32_t rev = swap_halves(fwd);                                        // a partner's current | next << 16
            uint32_t diff = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < K; ++k) diff = min(diff, (cur[k] | (nxt[k] << 16)) ^ rev);
            swap_hit |= (diff == 0u && best != 0xFFFFu) ? 1u : 0u;
            cand = owner ? cand & (cand - 1u) : cand;                                     // done with my lowest candidate
        }
    }
    PairAcc<true> acc;
    acc.dup = 0xFFFFFFFFu;
    acc.vertex = vertex_hit - 1u;            // 0 <=> hit (both half-words zero), else all ones
    acc.swap = swap_hit - 1u;
    return acc;
}

// non-zero <=> one of the two half-words of a is zero (the classic "has a zero byte" test on 16-bit fields): the only
// bits that can be set are 15 and 31
__device__ __forceinline__ uint32_t zero_half(uint32_t a) {
    return (a - 0x00010001u) & ~a & 0x80008000u;
}

// MapfEnv.is_terminal (mapf_env.py:210-223) of the group's env, in every lane
template <int Q, int P>
__device__ __forceinline__ bool packed_is_terminal(const LaneCtx<Q> &x, const uint32_t (&c)[P], const uint32_t (&g)[P]) {
    const uint32_t none[P] = {};
    const PairAcc<true> acc = packed_pair_tests<Q, P, true, false>(x, c, none);
    bool off_goal = false;
#pragma unroll
    for (int i = 0; i < P; ++i) off_goal |= c[i] != g[i];
    const uint32_t flags = group_reduce<Q, false>((PairAcc<true>::hit(acc.dup) ? 1u : 0u) | (off_goal ? 2u : 0u), x);
    return (flags & 1u) != 0u || (flags & 2u) == 0u;
}

// ordered product over agents 0..A-1: every lane continues the product handed over by the lane before it (see
// prob_product in mapf_lg.hpp); the total ends in lane Q-1
template <int Q, int K>
__device__ __forceinline__ double packed_prob_product(const double (&q)[K]) {
    double run = q[0];
#pragma unroll
    for (int i = 1; i < K; ++i) run = __dmul_rn(run, q[i]);
#pragma unroll
    for (int k = 1; k < Q; ++k) {
        const uint32_t lo = from_prev_lane<Q>(uint32_t(__double2loint(run)));
        const uint32_t hi = from_prev_lane<Q>(uint32_t(__double2hiint(run)));
        run = __hiloint2double(int(hi), int(lo));
#pragma unroll
        for (int i = 0; i < K; ++i) run = __dmul_rn(run, q[i]);
    }
    return run;
}

}  // namespace
}  // namespace mapf
