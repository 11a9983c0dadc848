// Device-side building blocks shared by both kernel families of the batched MapfEnv.step() path (gfx950 only):
// Philox4x32-10 and the slip-stream layout, env-major row I/O, the per-agent slip sampling against the host-built
// tables (move table + slip rows), and the thread-per-env transition (one lane owns all A agents of an env; the
// lane-group transition lives in mapf_lg.hpp).
// Float64 arithmetic follows the reference operation by operation (mapf_env.py:163-184, :225-266, :436-446); every
// add/mul that decides a bit is an explicit round-to-nearest intrinsic so no FMA contraction can change it.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mapf_kernels.hpp"

namespace mapf {


// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0;   // one v_mad_u64_u32 gives hi and lo
        const uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        c0 = __builtin_amdgcn_bitop3_b32(uint32_t(p1 >> 32), c1, k0, 0x96);   // three-input xor in one VALU op
        c1 = uint32_t(p1);
        c2 = __builtin_amdgcn_bitop3_b32(uint32_t(p0 >> 32), c3, k1, 0x96);
        c3 = uint32_t(p0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        // Keep the key schedule as two live scalars: without this the compiler hoists all 20 round keys
        // of each stream into SGPRs for the whole kernel, which forces SGPR spills at larger A.
        asm volatile("" : "+s"(k0), "+s"(k1));
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Two calls that share the key and the env words of the counter (they differ in the last two words), advanced in
// lockstep: the round's four multiplies are independent of each other (one call alone is a chain of dependent
// multiply -> xor pairs, and the scalar launder below is a scheduling fence, so two separate calls would run back to back).
__device__ __forceinline__ void philox4x32_10_x2(uint32_t c0, uint32_t c1, uint32_t c2a, uint32_t c3a, uint32_t c2b, uint32_t c3b,
                                                 uint32_t k0, uint32_t k1, uint32_t (&outa)[4], uint32_t (&outb)[4]) {
    uint32_t a0 = c0, a1 = c1, a2 = c2a, a3 = c3a, b0 = c0, b1 = c1, b2 = c2b, b3 = c3b;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t pa0 = uint64_t(0xD2511F53u) * a0, pa1 = uint64_t(0xCD9E8D57u) * a2;
        const uint64_t pb0 = uint64_t(0xD2511F53u) * b0, pb1 = uint64_t(0xCD9E8D57u) * b2;
        a0 = __builtin_amdgcn_bitop3_b32(uint32_t(pa1 >> 32), a1, k0, 0x96);
        b0 = __builtin_amdgcn_bitop3_b32(uint32_t(pb1 >> 32), b1, k0, 0x96);
        a1 = uint32_t(pa1);
        b1 = uint32_t(pb1);
        a2 = __builtin_amdgcn_bitop3_b32(uint32_t(pa0 >> 32), a3, k1, 0x96);
        b2 = __builtin_amdgcn_bitop3_b32(uint32_t(pb0 >> 32), b3, k1, 0x96);
        a3 = uint32_t(pa0);
        b3 = uint32_t(pb0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        asm volatile("" : "+s"(k0), "+s"(k1));
    }
    outa[0] = a0; outa[1] = a1; outa[2] = a2; outa[3] = a3;
    outb[0] = b0; outb[1] = b1; outb[2] = b2; outb[3] = b3;
}

// 53-bit integer of a uniform in [0,1): u = mant * 2^-53 -- same construction as RandomState.rand().
__device__ __forceinline__ uint64_t mantissa53(uint32_t a, uint32_t b) {
    return (uint64_t(a >> 5) << 26) | uint64_t(b >> 6);
}

// ------------------------------------------------------------ env-major row I/O
constexpr int lowbit(int x) { return x & -x; }
constexpr int row_align(int bytes) { return lowbit(bytes) < 16 ? lowbit(bytes) : 16; }

template <typename T, int N>
struct alignas(row_align(int(sizeof(T)) * N)) Row { T v[N]; };

template <typename T, int N>
__device__ __forceinline__ Row<T, N> load_row(const T *base, uint64_t row) {
    return *reinterpret_cast<const Row<T, N> *>(base + row * N);
}
template <typename T, int N>
__device__ __forceinline__ void store_row(T *base, uint64_t row, const Row<T, N> &r) {
    *reinterpret_cast<Row<T, N> *>(base + row * N) = r;
}

// step index of a launch's first step (StepArgs::t_dev): one scalar load when the launch was recorded into a graph
template <typename ArgsT>
__device__ __forceinline__ uint64_t first_step_index(const ArgsT &p) {
    // (constant address space: a scalar load -- nothing writes *t_dev while a kernel that reads it runs)
    return p.t_dev ? p.t + *(const __attribute__((address_space(4))) uint64_t *)(uintptr_t)p.t_dev : p.t;
}

// end-of-step signal of a ONE-WAVE launch (StepArgs::done_flag): every store of the wave is older than this release
__device__ __forceinline__ void signal_step_done(uint32_t *flag, uint32_t seq) {
    if (flag) {
        __threadfence_system();
        // (a flagged launch has at most 64 live threads, all in the first wave of block 0: no other wave may publish)
        if (blockIdx.x == 0u && threadIdx.x == 0u) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------ one env step
template <int A>
struct StepResult {
    uint32_t next[A];
    double reward, prob;
    bool done, collision, was_terminal;
};

// The slip table lives in LDS: 8 rows (one per equality code of the three candidate cells), built on
// the host by replaying single_agent_movements (mapf_env.py:163-184) -- see build_slip_table().
__device__ __forceinline__ void stage_slip_table(const SlipRow *__restrict__ src, SlipRow *lds) {
    const uint64_t *s = reinterpret_cast<const uint64_t *>(src);
    uint64_t *d = reinterpret_cast<uint64_t *>(lds);
    constexpr int n_words = int(sizeof(SlipRow) * 8 / sizeof(uint64_t));
    for (int w = threadIdx.x; w < n_words; w += blockDim.x) d[w] = s[w];
    __syncthreads();
}

// One agent.  `entry` = move table row of (cell, action): the merged movement list's cells in list order
// (c0 | c1 << 16 | c2 << 32) and the equality code of its three candidates (bits 48..50) -- the code picks
// the LDS row holding that list's probabilities and cumulative thresholds.  `mant` is the 53-bit integer
// of the uniform (u = mant * 2^-53), used when !EXT_UNIFORMS.
__device__ __forceinline__ uint32_t entry_cell(const MoveEntry &entry, uint32_t idx) {   // list slot idx = 0, 1, 2
    // half-word idx of {y, x}, zero-extended: one v_perm_b32 with selector bytes (0x0c = constant 0, 2*idx+1, 2*idx)
    return __builtin_amdgcn_perm(entry.y, entry.x, 0x0C0C0100u + idx * 0x0202u);
}
__device__ __forceinline__ uint32_t entry_code(const MoveEntry &entry) { return (entry.y >> 16) & 7u; }
// byte offset of that code's SlipRow (code * sizeof(SlipRow), packed beside the code by the host)
__device__ __forceinline__ uint32_t entry_row_offset(const MoveEntry &entry) { return entry.w; }
static_assert(offsetof(SlipRow, q) == 0, "entry_row_offset() addresses q[] directly");

template <bool EXT_UNIFORMS>
__device__ __forceinline__ void slip_move(const SlipRow *lds_slip, const MoveEntry &entry, uint64_t mant, double u,
                                          uint32_t &next, double &q) {
    const SlipRow &row = lds_slip[entry_code(entry)];
    // categorical_sample (call site mapf_env.py:255): (cumsum(p) > u).argmax(), all-False -> 0.
    // cum[k] > u  <=>  mant < ceil(cum[k] * 2^53) = thr[k]; rows shorter than 3 carry thr = 0 / cum = -inf.
    bool b0, b1, b2;
    if (EXT_UNIFORMS) {
        b0 = row.cum[0] > u; b1 = row.cum[1] > u; b2 = row.cum[2] > u;
    } else {
        b0 = mant < row.thr[0]; b1 = mant < row.thr[1]; b2 = mant < row.thr[2];
    }
    // the index is turned into an integer at once (wave-mask booleans that stay live cost an SGPR pair each)
    const uint32_t idx = b0 ? 0u : (b1 ? 1u : (b2 ? 2u : 0u));
    next = entry_cell(entry, idx);
    q = row.q[idx];
}

// The list slot alone (exact, 53-bit mantissa), for callers that fetch the slot's cell and probability themselves
__device__ __forceinline__ uint32_t slip_slot_exact(const SlipRow *lds_slip, const MoveEntry &entry, uint64_t mant) {
    const SlipRow &row = lds_slip[entry_code(entry)];
    const bool b0 = mant < row.thr[0], b1 = mant < row.thr[1], b2 = mant < row.thr[2];
    return b0 ? 0u : (b1 ? 1u : (b2 ? 2u : 0u));
}

// Fast path of the same sampling with only the top 16 bits of the uniform (hi = mant >> 37): hi < th[k] decides
// mant < thr[k] unless hi == th[k]; `tie_dist` is 0 in that (rare) case and the caller repeats the move
// with the full 53-bit mantissa.
__device__ __forceinline__ void slip_move_hi(const SlipRow *lds_slip, const MoveEntry &entry, uint32_t hi, uint32_t &next,
                                             double &q, uint32_t &tie_dist) {
    // the thresholds travel with the move-table row: no LDS access until the sampled probability is fetched
    // One subtraction per threshold serves both questions: negative <=> hi < t, zero <=> a tie.  A list's last
    // cumulative sum is 1 up to rounding (or more for an out-of-range fail_prob), so the top 16 bits of its threshold
    // are always 65535 -- as are the entries past the list end -- and `hi < 65535` only fails in a tie: away from
    // ties the slot is how many of the first two thresholds hi has passed.
    const uint32_t t0 = entry.z & 0xFFFFu, t1 = entry.z >> 16;
    const uint32_t d0 = hi - t0, d1 = hi - t1, d2 = hi - 0xFFFFu;
    tie_dist = min(d0, min(d1, d2));   // 0 <=> hi ties with a threshold (integer, no wave-mask booleans)
    const uint32_t idx = 2u - (d0 >> 31) - (d1 >> 31);
    next = entry_cell(entry, idx);
    q = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lds_slip) + entry_row_offset(entry) + idx * 8u);
}

// The same, with the probability rebuilt from the slot's members (bits 19.. of entry.y) instead of read from the
// row: ((m ? p_m : 0) + (r ? p_r : 0)) + (l ? p_l : 0) in the order the host merged them (old + new, first-seen
// order; adding 0.0 is exact).  Used where the row would be one more dependent memory round trip (single steps).
__device__ __forceinline__ void slip_move_hi_members(const EnvConsts &c, const MoveEntry &entry, uint32_t hi, uint32_t &next,
                                                     double &q, uint32_t &tie_dist) {
    const uint32_t t0 = entry.z & 0xFFFFu, t1 = entry.z >> 16;
    const uint32_t d0 = hi - t0, d1 = hi - t1, d2 = hi - 0xFFFFu;   // see slip_move_hi
    tie_dist = min(d0, min(d1, d2));
    const uint32_t idx = 2u - (d0 >> 31) - (d1 >> 31);
    next = entry_cell(entry, idx);
    const uint32_t mem = (entry.y >> (19u + 3u * idx)) & 7u;
    q = __dadd_rn(__dadd_rn((mem & 1u) ? c.p_cand[0] : 0.0, (mem & 2u) ? c.p_cand[1] : 0.0), (mem & 4u) ? c.p_cand[2] : 0.0);
}

// Exact sampling with the full 53-bit mantissa and NO memory access (the single step's answer to a 16-bit tie: the wave
// that takes this path is the launch's straggler, so it must not add a dependent load): the list's probabilities are
// rebuilt from the slots' members as in slip_move_hi_members, the cumulative sums are formed in list order like the
// host's (build_slip_table: run = q0, run + q1, ...), and u = mant * 2^-53 is exact -- so `cum[k] > u` is literally
// categorical_sample's comparison (and equals the rows' `mant < ceil(cum[k] * 2^53)`).  An absent slot adds 0.0 and
// repeats the comparison of the slot before it, which cannot turn a False into the first True.
__device__ __forceinline__ void slip_move_exact_members(const EnvConsts &c, const MoveEntry &entry, uint64_t mant, uint32_t &next, double &q) {
    const double u = __dmul_rn(__ull2double_rn(mant), 0x1p-53);
    double qs[3], cum[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t mem = (entry.y >> (19u + 3u * uint32_t(k))) & 7u;
        qs[k] = __dadd_rn(__dadd_rn((mem & 1u) ? c.p_cand[0] : 0.0, (mem & 2u) ? c.p_cand[1] : 0.0), (mem & 4u) ? c.p_cand[2] : 0.0);
        cum[k] = k == 0 ? qs[0] : __dadd_rn(cum[k - 1], qs[k]);
    }
    const uint32_t idx = cum[0] > u ? 0u : (cum[1] > u ? 1u : (cum[2] > u ? 2u : 0u));
    next = entry_cell(entry, idx);
    q = idx == 0u ? qs[0] : (idx == 1u ? qs[1] : qs[2]);
}

// Slip stream (oracle/philox.py): one call with rslot = refine = 0 yields the four words that serve the agent QUAD
// (4*quad .. 4*quad+3) at steps 2h and 2h+1 -- word 2 * (t & 1) + j serves the quad's pair j (agents 4*quad + 2j, + 2j+1):
// low half = top 16 bits of the even agent's uniform, high half = the odd agent's.  The low 37 bits of a slot's uniform
// come from a separate call (refine = 1, rslot = 4 * (t & 1) + (agent & 3)), needed only when the top 16 bits tie with a
// threshold.
struct Words4 { uint32_t w0, w1, w2, w3; };

__device__ __forceinline__ Words4 slip_words(const EnvConsts &c, uint64_t env_id, uint64_t h, uint32_t quad, uint32_t rslot,
                                             uint32_t refine) {
    const uint32_t c3 = (uint32_t(h >> 32) & 0xFFFFu) | (quad << 16) | (rslot << 23) | (refine << 31);
    uint32_t w[4];
    philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(h), c3, c.seed_lo, c.seed_hi, w);
    return Words4{w[0], w[1], w[2], w[3]};
}

// two calls of the same env in lockstep: (h_a, quad_a) and (h_b, quad_b)
__device__ __forceinline__ void slip_words_x2(const EnvConsts &c, uint64_t env_id, uint64_t h_a, uint32_t quad_a, uint64_t h_b,
                                              uint32_t quad_b, Words4 &wa, Words4 &wb) {
    uint32_t a[4], b[4];
    philox4x32_10_x2(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(h_a), (uint32_t(h_a >> 32) & 0xFFFFu) | (quad_a << 16),
                     uint32_t(h_b), (uint32_t(h_b >> 32) & 0xFFFFu) | (quad_b << 16), c.seed_lo, c.seed_hi, a, b);
    wa = Words4{a[0], a[1], a[2], a[3]};
    wb = Words4{b[0], b[1], b[2], b[3]};
}

// word k of a call (k = 0..3, run-time)
__device__ __forceinline__ uint32_t pick_word(const Words4 &w, uint32_t k) {
    return k == 0u ? w.w0 : (k == 1u ? w.w1 : (k == 2u ? w.w2 : w.w3));
}
// A fused rollout keeps, per agent pair, the words of the FOUR-step block 4m .. 4m+3 (two calls: h = 2m, 2m+1) in step
// order: word (t & 3) of that
__device__ __forceinline__ uint32_t step_word(const Words4 &w, uint64_t t) { return pick_word(w, uint32_t(t) & 3u); }
// ... built from the block's two calls for pair j (0 / 1) of the quad
__device__ __forceinline__ Words4 block_words(const Words4 &a, const Words4 &b, uint32_t j) {
    return j == 0u ? Words4{a.w0, a.w2, b.w0, b.w2} : Words4{a.w1, a.w3, b.w1, b.w3};
}
// this step's word of pair j of a quad, from the call of (t >> 1)
__device__ __forceinline__ uint32_t quad_step_word(const Words4 &w, uint64_t t, uint32_t j) {
    return pick_word(w, 2u * (uint32_t(t) & 1u) + j);
}
// first call index of the four-step block that contains step t
__device__ __forceinline__ uint64_t block_first_call(uint64_t t) { return (t >> 1) & ~uint64_t(1); }

// full 53-bit mantissa of (t, agent) given the top 16 bits: one refinement call
__device__ __forceinline__ uint64_t refine_mantissa(const EnvConsts &c, uint64_t env_id, uint64_t t, uint32_t agent, uint32_t hi16) {
    const Words4 r = slip_words(c, env_id, t >> 1, agent >> 2, 4u * (uint32_t(t) & 1u) + (agent & 3u), 1u);
    return (uint64_t(hi16) << 37) | (uint64_t(r.w0 & 0x1Fu) << 32) | uint64_t(r.w1);
}

// move-table row of (cell, action); cells beyond V (only reachable through a corrupted state) are clamped
template <bool CLAMP = true>
__device__ __forceinline__ MoveEntry move_entry(const MoveEntry *__restrict__ mv, uint32_t n_cells, uint32_t cell,
                                                uint32_t action) {
    // CLAMP = false only for an LDS-resident table: an out-of-range LDS read returns zeros instead of faulting
    const uint32_t c = (!CLAMP || cell < n_cells) ? cell : n_cells - 1u;
    uint32_t row = __umul24(c, kMvCols) + action;   // cells are 16-bit: one v_mad_u32_u24
    asm volatile("" : "+v"(row));             // keep row * 16 + base as one shift-add (not c * 80 + action * 16 + base)
    return mv[row];
}

template <int A, bool EXT_UNIFORMS>
__device__ __forceinline__ void env_transition(const EnvConsts &c, const MoveEntry *__restrict__ mv,
                                               const SlipRow *lds_slip,
                                               const uint32_t (&prev)[A], const uint32_t (&goal)[A],
                                               const uint32_t (&act_in)[A], const double *ext_u,
                                               uint64_t env_id, uint64_t t, StepResult<A> &out) {
    // is_terminal(prev): mapf_env.py:210-223.  Pair tests accumulate min(xor) in a VGPR (zero <=> some pair
    // equal) instead of OR-ing wave masks: 1.5 VALU per pair and no SGPR pressure at A = 32.
    uint32_t dup_acc = 0xFFFFFFFFu, goal_acc = 0u;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        goal_acc |= prev[i] ^ goal[i];
#pragma unroll
        for (int j = i + 1; j < A; ++j) dup_acc = min(dup_acc, prev[i] ^ prev[j]);
    }
    const bool dup = (A > 1) && dup_acc == 0u, all_goal = goal_acc == 0u;
    if (dup || all_goal) {  // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0}), nothing drawn
#pragma unroll
        for (int i = 0; i < A; ++i) out.next[i] = prev[i];
        out.reward = 0.0; out.prob = 0.0;
        out.done = true; out.collision = false; out.was_terminal = true;
        return;
    }
    out.was_terminal = false;

    uint32_t act[A];
#pragma unroll
    for (int i = 0; i < A; ++i) act[i] = act_in[i] > 4u ? 0u : act_in[i];

    // the table rows of all agents are independent gathers: issue them together
    MoveEntry entry[A];
#pragma unroll
    for (int i = 0; i < A; ++i) entry[i] = move_entry(mv, c.n_cells, prev[i], act[i]);

    double prob = 1.0;
    if (EXT_UNIFORMS) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            double pr;
            slip_move<true>(lds_slip, entry[i], 0, ext_u[i], out.next[i], pr);
            prob = (i == 0) ? pr : __dmul_rn(prob, pr);  // total_prob *= p, agent order (:257)
        }
    } else {
        // fast path on the top 16 bits of every agent's uniform; the (rare) wave with a tie redoes all its agents
        // with the full 53-bit mantissas
        uint32_t hi[A];
        uint32_t tie = 0xFFFFFFFFu;
        Words4 w{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if ((i & 3) == 0 && c.need_rng) w = slip_words(c, env_id, t >> 1, uint32_t(i >> 2), 0u, 0u);
            const uint32_t word = quad_step_word(w, t, uint32_t(i >> 1) & 1u);
            hi[i] = (i & 1) ? (word >> 16) : (word & 0xFFFFu);
            double pr;
            uint32_t dist;
            slip_move_hi(lds_slip, entry[i], hi[i], out.next[i], pr, dist);
            tie = min(tie, dist);
            prob = (i == 0) ? pr : __dmul_rn(prob, pr);
        }
        if (__builtin_expect(__any(tie == 0u && c.need_rng), 0)) {
            prob = 1.0;
#pragma unroll
            for (int i = 0; i < A; ++i) {
                double pr;
                slip_move<false>(lds_slip, entry[i], refine_mantissa(c, env_id, t, uint32_t(i), hi[i]), 0.0, out.next[i], pr);
                prob = (i == 0) ? pr : __dmul_rn(prob, pr);
            }
        }
    }
    out.prob = prob;

    // _living_reward: mapf_env.py:436-446
    double living = c.r_living;
    if (c.criteria == 1u) {
        int stayed = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) stayed += (prev[i] == goal[i] && act[i] == 0u) ? 1 : 0;
        living = __dmul_rn(double(A - stayed), c.r_living);
    }
    // _is_collision_transition_from_local_states: mapf_env.py:378-389
    uint32_t coll_acc = 0xFFFFFFFFu, goal_next_acc = 0u;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        goal_next_acc |= out.next[i] ^ goal[i];
        const uint32_t fwd = prev[i] | (out.next[i] << 16);
#pragma unroll
        for (int j = i + 1; j < A; ++j) {
            const uint32_t rev = out.next[j] | (prev[j] << 16);
            coll_acc = min(coll_acc, min(out.next[i] ^ out.next[j], fwd ^ rev));   // vertex, swap
        }
    }
    const bool coll = (A > 1) && coll_acc == 0u, goal_next = goal_next_acc == 0u;
    // calc_transition_reward_from_local_states: mapf_env.py:225-235 (collision before goal)
    out.collision = coll;
    out.done = coll || goal_next;
    out.reward = coll ? __dadd_rn(c.r_clash, living) : (goal_next ? __dadd_rn(c.r_goal, living) : living);
}

// Greedy policy (include/mapf_hip.h MAPF_POLICY_GREEDY; no reference counterpart): cells[c] = {row | col << 16,
// nine 3-bit actions indexed by 3 * (sgn(goal_row - row) + 1) + (sgn(goal_col - col) + 1)} -- the first action in
// ACTIONS order that is not blocked and moves one step closer to a goal lying in that direction, else STAY.
__device__ __forceinline__ uint32_t greedy_action(const uint2 *cells, uint32_t n_cells, uint32_t cell, uint32_t goal_rc) {
    const uint2 pc = cells[min(cell, n_cells - 1u)];   // (a corrupted state must not turn into a wild read)
    const int r = int(pc.x & 0xFFFFu), c = int(pc.x >> 16), gr = int(goal_rc & 0xFFFFu), gc = int(goal_rc >> 16);
    const int k = 3 * ((gr > r) - (gr < r) + 1) + ((gc > c) - (gc < c) + 1);
    return (pc.y >> (3 * k)) & 7u;
}

// Policy stream (oracle/philox.py random_actions_np; key = seed + 1): ONE call serves an agent quad for FOUR consecutive
// steps -- counter (env, m = t >> 2, quad), word t & 3 belongs to step t, its byte (agent & 3) to the agent, and
// action = (byte * 5) >> 8.  A fused rollout pays one call per lane per four steps and a multiply + shift per action.
__device__ __forceinline__ Words4 policy_words(uint32_t key_lo, uint32_t key_hi, uint64_t env_id, uint64_t m, uint32_t quad) {
    uint32_t w[4];
    philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(m), (uint32_t(m >> 32) & 0xFFFFu) | (quad << 16), key_lo, key_hi, w);
    return Words4{w[0], w[1], w[2], w[3]};
}
__device__ __forceinline__ Words4 policy_words(const EnvConsts &c, uint64_t env_id, uint64_t m, uint32_t quad) {
    return policy_words(c.pol_lo, c.pol_hi, env_id, m, quad);
}
// two quads of one env in lockstep (eight agents per lane)
__device__ __forceinline__ void policy_words_x2(uint32_t key_lo, uint32_t key_hi, uint64_t env_id, uint64_t m, uint32_t quad_a, uint32_t quad_b, Words4 &wa, Words4 &wb) {
    uint32_t a[4], b[4];
    const uint32_t hi = uint32_t(m >> 32) & 0xFFFFu;
    philox4x32_10_x2(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(m), hi | (quad_a << 16), uint32_t(m), hi | (quad_b << 16), key_lo, key_hi, a, b);
    wa = Words4{a[0], a[1], a[2], a[3]};
    wb = Words4{b[0], b[1], b[2], b[3]};
}
// action of byte B (0..3) of a step's policy word: one multiply with a byte select, one (fast-issue) right shift
template <int B>
__device__ __forceinline__ uint32_t policy_action(uint32_t word, uint32_t five) {   // five: the constant 5 in a VECTOR register
    uint32_t scaled;
    if (B == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(scaled) : "v"(word), "v"(five));
    else if (B == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(scaled) : "v"(word), "v"(five));
    else if (B == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(scaled) : "v"(word), "v"(five));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(scaled) : "v"(word), "v"(five));
    return scaled >> 8;
}
__device__ __forceinline__ uint32_t policy_action_rt(uint32_t word, uint32_t byte) { return (((word >> (8u * byte)) & 0xFFu) * 5u) >> 8; }

// thread-per-env form: every quad's call of the step's block, recomputed each step (that family is the small-batch /
// cross-check path; the packed and lane-group rollouts keep a block's words for its four steps)
template <int A>
__device__ __forceinline__ void policy_actions(const EnvConsts &c, uint64_t env_id, uint64_t t,
                                               uint32_t (&act)[A]) {
    Words4 w{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if ((i & 3) == 0) w = policy_words(c, env_id, t >> 2, uint32_t(i >> 2));
        act[i] = policy_action_rt(step_word(w, t), uint32_t(i & 3));
    }
}

}  // namespace mapf
