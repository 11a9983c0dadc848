// env.P[s][a] on the device: every branch of the joint slip distribution of a (state, joint action) query
// (reference mapf_env.py:448-478 `_get_transitions`, the enumeration planners iterate over).
//
// Branch b of a query selects entry k_i of agent i's merged movement list (single_agent_movements, :163-184) in
// itertools.product order -- agent 0 varies slowest -- so k_i = digit i of b in the mixed radix (n_0, ..., n_{A-1}).
// prob is the left-to-right float64 product of the selected probabilities (functools.reduce at :467); reward / done /
// collision come from the same rules as step() (:225-235).  A terminal state has the single branch
// ((1.0, False), s, 0, True) (:455-456).  A call returns the WINDOW [first_branch, first_branch + max_branches) of every
// query's enumeration, so a caller pages through the 3^A branches of a large team in pieces.
//
// Two output layouts: RESERVED rows (row j of query q's window at q * max_branches + j: mapf_transitions /
// mapf_transitions_window) and COMPACTED rows (the windows of all queries back to back behind an exclusive scan of their
// lengths: mapf_transitions_compact -- a room map uses a fifth of the 3^A rows a query reserves).
//
// Two kernels:
//   * transitions_rows_kernel<MAXA> (A <= 8, i.e. everything planners enumerate in practice).  The kernel is bound by its
//     vector instructions, not by the bytes it writes (profiles/r05_transitions_*), so it is built around the per-ROW
//     instruction count.  A wave owns QW consecutive queries.  SET-UP is one LANE per query: cells, goals, actions, table
//     rows, list lengths, is_terminal(prev), the SoC living reward -- and, per (agent, list entry), a 16-byte LDS record
//     {cell | off-goal bit, CONFLICT MASK, probability}: bit 3j + k_j of the mask of (i, k_i) says that entry k_i of agent
//     i collides (vertex or swap, :378-389) with entry k_j of a LATER agent j.  EMISSION is one lane per output ROW of the
//     wave's queries taken together (a binary search over the wave's prefix sums names the row's query): per agent one
//     digit, one ds_read_b128 of its record, `hit |= mask & chosen; chosen |= 1 << (3i + k)` -- the O(A^2) pair tests of
//     a row become O(A) and the off-goal test one OR per agent.  Consecutive lanes write consecutive rows.
//     (The form this replaces -- 64 lanes per query each repeating the query's set-up, 28 pair tests x 6 instructions per
//     row -- measured 330 vector instructions per 64 rows at 8 agents and one wave per ~35 rows at 4 agents.)
//   * transitions_kernel<A, EXACT> for 9..16 agents (48-bit choice sets do not fit the records): a group of 64 lanes owns a
//     chunk of one query's window, query set-up per lane, all agent pairs per row.
#include <algorithm>
#include <type_traits>
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"

namespace mapf {

namespace {

// rows of a query's window: branches [first, first + max) of `count`
__device__ __forceinline__ uint32_t window_rows(uint32_t count, uint64_t first, uint32_t max_branches, uint32_t &lo) {
    lo = first < count ? uint32_t(first) : count;
    const uint64_t end = first + max_branches;
    const uint32_t hi = end < count ? uint32_t(end) : count;
    return hi - lo;
}

// number of branches of query q (product of the agents' list lengths; 1 for a terminal state), run-time A
__device__ uint32_t query_branch_count(const TransitionsArgs &p, const SlipRow *slip, uint64_t q) {
    const uint32_t A = p.n_agents;
    const uint64_t env = p.env_index ? p.env_index[q] : 0;
    const uint16_t *goal_row = p.goal + (p.goal_broadcast ? 0 : env * A);
    const uint16_t *state_row = p.local + q * A;
    const uint8_t *act_row = p.actions + q * A;
    uint32_t count = 1u, off_goal = 0u;
    bool dup = false;
    for (uint32_t i = 0; i < A; ++i) {
        const uint32_t cell = state_row[i], a = act_row[i];
        const MoveEntry entry = move_entry(p.mv, p.c.n_cells, cell, a > 4u ? 0u : a);
        count *= slip[entry_code(entry)].n;
        off_goal |= cell ^ goal_row[i];
        for (uint32_t j = i + 1; j < A; ++j) dup |= cell == state_row[j];
    }
    return (dup || off_goal == 0u) ? 1u : count;                 // is_terminal: mapf_env.py:210-223
}

// inclusive scan over the wave's lanes (every lane active)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = uint32_t(__shfl_up(int(v), int(d), 64));
        v += lane >= d ? up : 0u;
    }
    return v;
}

}  // namespace

// ---- compacted output, pass 1: every query's window length; exclusive scan inside blocks of 256 queries (rel[q]) and the
// blocks' totals.  Pass 2 (one block) scans the totals into block_base[] and writes the grand total.
constexpr uint32_t kScanBlock = 256;
__global__ void __launch_bounds__(kScanBlock) transitions_count_kernel(const TransitionsArgs p, uint32_t *rel, uint64_t *block_total) {
    __shared__ SlipRow slip[8];
    __shared__ uint32_t wave_total[kScanBlock / 64];
    stage_slip_table(p.slip, slip);
    const uint64_t q = uint64_t(blockIdx.x) * kScanBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t rows = 0u;
    if (q < p.n_queries) {
        const uint32_t count = query_branch_count(p, slip, q);
        uint32_t lo;
        rows = window_rows(count, p.first_branch, p.max_branches, lo);
        if (p.out_count) p.out_count[q] = count;
    }
    const uint32_t incl = wave_inclusive_scan(rows, lane);
    if (lane == 63u) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (uint32_t w = 0; w < kScanBlock / 64; ++w) {
        before += w < wave ? wave_total[w] : 0u;
        total += wave_total[w];
    }
    if (q < p.n_queries) rel[q] = before + incl - rows;
    if (threadIdx.x == 0u) block_total[blockIdx.x] = total;
}

// pass 2: block_total[0 .. n) -> exclusive prefix in place (64-bit), grand total to *out_total and out_offset_last
__global__ void __launch_bounds__(1024) scan_block_totals_kernel(uint64_t *block_total, uint32_t n, uint64_t *out_total) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0u) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n ? block_total[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t lo = uint32_t(__shfl_up(int(uint32_t(incl)), int(d), 64)), hi = uint32_t(__shfl_up(int(uint32_t(incl >> 32)), int(d), 64));
            incl += lane >= d ? (uint64_t(hi) << 32 | lo) : 0;
        }
        if (lane == 63u) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t before = carry_s, all = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            before += w < wave ? wave_sum[w] : 0;
            all += wave_sum[w];
        }
        if (i < n) block_total[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0u) carry_s += all;
        __syncthreads();
    }
    if (threadIdx.x == 0u) {
        block_total[n] = carry_s;                                  // (the array has n + 1 entries)
        if (out_total) *out_total = carry_s;
    }
}

// ---------------------------------------------------------------------------------------------------- A <= 8
namespace {

constexpr uint32_t kQueryHeader = 48, kRecord = 16;
struct QueryHeader {
    uint32_t radix;        // 2 bits per agent: its list length (1 for absent agents and for every agent of a terminal state)
    uint32_t first;        // first branch of the window (<= count)
    uint32_t terminal;     // 1: the single branch ((1.0, False), s, 0, True) of a terminal state
    uint32_t pad;
    double living;         // _living_reward of the query (mapf_env.py:436-446)
    uint8_t n[8];          // the list lengths again, a byte each, and
    uint16_t magic[8];     // ceil(2^15 / n): x / n == (x * magic) >> 15 for x < 2^15 (a branch index is below 3^8) -- the emission's
                           // mixed-radix digits by two 24-bit multiplies with a byte / word operand select each instead of a 32-bit
                           // mul_hi + mul_lo, two compares and two selects: 8 agents x 20000 queries 106-110 -> 131-133 G branches/s
    __device__ void set_divisors(int max_agents) {
        for (int i = 0; i < 8; ++i) {
            const uint32_t len = i < max_agents ? (radix >> (2 * i)) & 3u : 1u;
            n[i] = uint8_t(len);
            magic[i] = uint16_t(len == 3u ? 10923u : (32768u >> (len - 1u)));
        }
    }
};
static_assert(sizeof(QueryHeader) == kQueryHeader, "three ds_read_b128");

// a * (16-bit word W of packed) and a * (byte B of packed), 24-bit multiplies with an operand select
template <int W> __device__ __forceinline__ uint32_t mul24_word(uint32_t a, uint32_t packed) {
    uint32_t r;
    if constexpr (W == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(packed));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(packed));
    return r;
}
template <int B> __device__ __forceinline__ uint32_t mul24_byte(uint32_t a, uint32_t packed) {
    uint32_t r;
    if constexpr (B == 0) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(packed));
    else if constexpr (B == 1) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(packed));
    else if constexpr (B == 2) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(packed));
    else asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(packed));
    return r;
}

// ALL_OUT: every output array is given (what the Python wrapper and the bench pass): no per-array branches around the stores.
// A batch (the QW queries of a wave) whose windows hold more than `rows_per_wave` rows is cut into PIECES, one wave each
// (pieces_max > 1: the scan has run, a wave whose piece lies beyond its batch's rows leaves at once) -- a room map's 8-agent
// queries have 1 to 6561 branches, and with whole batches per wave the longest waves decided the launch's length.
template <int MAXA, bool ALL_OUT>
__global__ void __launch_bounds__(256) transitions_rows_kernel(const TransitionsArgs p, const uint32_t qw_log2, const uint32_t pieces_max,
                                                               const uint32_t rows_per_wave, const uint32_t unscanned_blocks) {
    static_assert(MAXA <= 10, "a choice set is 3 bits per agent in one 32-bit word");
    __shared__ SlipRow slip[8];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    stage_slip_table(p.slip, slip);                                // ends with __syncthreads(): the only block-wide one
    constexpr uint32_t kQueryBytes = kQueryHeader + uint32_t(MAXA) * 3u * kRecord;
    const uint32_t QW = 1u << qw_log2;                             // queries per wave (>= 4: the prefix array stays 16-byte aligned)
    const uint32_t lane = threadIdx.x & 63u;
    // COOP (the 6- and 8-agent instances): a query's set-up is spread over the G = 64 / QW >= 8 lanes of its group, one lane per
    // AGENT -- done by one lane per query, four lanes of a wave working, it was 27 % of the kernel's vector instructions
    // (profiles/r05_transitions_a8_q20000_compact.txt: ~2600 per piece against ~180 per 64 rows)
    constexpr bool COOP = MAXA >= 6;
    constexpr uint32_t kScratch = COOP ? uint32_t(MAXA) * 16u : 0u;    // per query: {prev, the three list cells} of every agent
    unsigned char *const wave_lds = lds_dyn + (threadIdx.x >> 6) * (QW * (kQueryBytes + 4u + kScratch));
    uint32_t *const prefix = reinterpret_cast<uint32_t *>(wave_lds);                     // [QW] exclusive prefix of the windows' lengths
    unsigned char *const queries = wave_lds + QW * 4u;                                     // [QW] header + records
    unsigned char *const scratch = queries + QW * kQueryBytes;                             // [QW][MAXA] uint4 (COOP)
    const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    uint64_t batch = wave;
    uint32_t piece = 0u;
    if (pieces_max > 1u) { batch = wave / pieces_max; piece = uint32_t(wave - batch * pieces_max); }
    const uint64_t q0 = batch << qw_log2;
    if (q0 >= p.n_queries) return;                                 // (wave-uniform)
    const uint32_t A = p.n_agents;
    const uint32_t row_begin = piece * rows_per_wave;              // my piece of the batch's rows
    uint64_t base = 0;                                             // first row of the batch in the compacted arrays
    if (p.compact || pieces_max > 1u) {
        const uint64_t q1 = q0 + QW, blocks = (p.n_queries + kScanBlock - 1) / kScanBlock;
        uint64_t end;
        if (unscanned_blocks == 0u) {                              // block_base[] holds the scanned totals (pass 2 has run)
            base = p.block_base[q0 / kScanBlock] + p.rel[q0];
            end = q1 < p.n_queries ? p.block_base[q1 / kScanBlock] + p.rel[q1] : p.block_base[blocks];
        } else {
            // a call of at most kFusedScanBlocks scan blocks (65536 queries) skips pass 2 -- a one-block launch behind pass 1, ~6 us
            // of a call that takes 50 (8 agents x 2000 queries) to 190 us (x 20000): every wave adds up the totals below its block
            // itself (four loads a lane at most; 32-bit sums: 256 blocks x 256 queries x 3^8 rows < 2^32)
            const uint32_t b0 = uint32_t(q0 / kScanBlock);
            uint32_t below = 0u;
            for (uint32_t i = lane; i < b0; i += 64u) below += uint32_t(p.block_base[i]);
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) below += uint32_t(__shfl_xor(int(below), int(d), 64));
            base = uint64_t(below) + p.rel[q0];
            if (q1 < p.n_queries) {
                const uint32_t b1 = uint32_t(q1 / kScanBlock);     // b0 or b0 + 1 (QW <= 64)
                end = uint64_t(below) + (b1 > b0 ? uint32_t(p.block_base[b0]) : 0u) + p.rel[q1];
            } else {                                               // the call's last batch: its end is the grand total
                end = below;
                for (uint32_t i = b0; i < unscanned_blocks; ++i) end += uint32_t(p.block_base[i]);   // (b0 is the last block or the one before)
                if (p.compact && p.out_offset && piece == 0u && lane == 0u) p.out_offset[p.n_queries] = end;
            }
        }
        if (pieces_max > 1u && piece > 0u && uint64_t(row_begin) >= end - base) return;   // (piece 0 stays: it reports the counts)
    }

    uint32_t rows = 0u;                                            // the window's length, in the lane that reports it to the scan
    if constexpr (COOP) {
        // ---- set-up, one lane per (query, agent): group gq = lane / G owns query q0 + gq, lane r = lane % G < MAXA its agent r
        const uint32_t g_log2 = 6u - qw_log2, G = 1u << g_log2, gq = lane >> g_log2, r = lane & (G - 1u);
        const uint64_t q = q0 + gq;
        const bool live = q < p.n_queries, slot = r < uint32_t(MAXA), on = live && r < A;
        unsigned char *const mine = queries + gq * kQueryBytes;
        uint4 *const mine_scratch = reinterpret_cast<uint4 *>(scratch + gq * kScratch);
        // group reductions over the G lanes of a query (G = 8 or 16: xor shuffles stay inside the group)
        auto group_or = [&](uint32_t v) __attribute__((always_inline)) {
            for (uint32_t d = 1; d < G; d <<= 1) v |= uint32_t(__shfl_xor(int(v), int(d), 64));
            return v;
        };
        auto group_sum = [&](uint32_t v) __attribute__((always_inline)) {
            for (uint32_t d = 1; d < G; d <<= 1) v += uint32_t(__shfl_xor(int(v), int(d), 64));
            return v;
        };
        auto group_product = [&](uint32_t v) __attribute__((always_inline)) {
            for (uint32_t d = 1; d < G; d <<= 1) v *= uint32_t(__shfl_xor(int(v), int(d), 64));
            return v;
        };
        uint32_t prev = 0x10000u + r, goal = 0x10000u + r, act = 0u, n = 1u, cell[3];
        double pr[3] = {1.0, 1.0, 1.0};                             // (x * 1.0 is exact: absent agents leave the product alone)
        MoveEntry entry{0u, 0u, 0u, 7u * uint32_t(sizeof(SlipRow))};
        if (on) {
            const uint64_t env = p.env_index ? p.env_index[q] : 0;
            prev = p.local[q * A + r];
            goal = p.goal[(p.goal_broadcast ? 0 : env * A) + r];
            const uint32_t a = p.actions[q * A + r];
            act = a > 4u ? 0u : a;
            entry = move_entry(p.mv, p.c.n_cells, prev, act);
            const SlipRow &row = slip[entry_code(entry)];
            n = row.n;
#pragma unroll
            for (int k = 0; k < 3; ++k) pr[k] = row.q[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t c = on ? entry_cell(entry, uint32_t(k)) : prev;
            if (slot) *reinterpret_cast<uint4 *>(mine + kQueryHeader + (3u * r + uint32_t(k)) * kRecord) =
                make_uint4((c & 0xFFFFu) | (c != goal ? 0x10000u : 0u), 0u, uint32_t(__double2loint(pr[k])), uint32_t(__double2hiint(pr[k])));
            cell[k] = uint32_t(k) < n ? c : 0x20000u + 3u * r + uint32_t(k);   // an entry past the list's end: a cell nothing can equal
        }
        if (slot) mine_scratch[r] = make_uint4(prev, cell[0], cell[1], cell[2]);
        const uint32_t radix_all = group_or(slot ? n << (2u * r) : 0u);
        uint32_t count = group_product(n);
        const uint32_t goal_acc = group_or(prev ^ goal);
        const int stayed = int(group_sum((on && prev == goal && act == 0u) ? 1u : 0u));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // my agent's entries against every entry of every LATER agent: vertex (same next cell) or swap (each moves onto the
        // other's current cell) -- _is_collision_transition_from_local_states, mapf_env.py:378-389; integer tests (t == 0 <=> hit)
        uint32_t mask[3] = {0u, 0u, 0u}, dup = 0u;
#pragma unroll
        for (int j = 1; j < MAXA; ++j) {
            const uint4 other = mine_scratch[j];                   // {prev_j, c_j[0..2]}
            const uint32_t later = uint32_t(j) > r ? 1u : 0u;
            dup |= (((prev ^ other.x) - 1u) >> 31) & later;         // is_terminal: two agents share a cell (mapf_env.py:210-223)
            const uint32_t cj[3] = {other.y, other.z, other.w};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int kj = 0; kj < 3; ++kj) {
                    const uint32_t t = min(cell[k] ^ cj[kj], (prev ^ cj[kj]) | (other.x ^ cell[k]));
                    mask[k] |= (((t - 1u) >> 31) & later) << (3 * j + kj);
                }
        }
        const bool terminal = group_or(dup) != 0u || goal_acc == 0u;
        if (terminal) count = 1u;
        if (slot) {
            if (terminal) {   // the single branch ((1.0, False), s, 0, True): every agent's one entry is its own cell, probability 1.0
                *reinterpret_cast<uint4 *>(mine + kQueryHeader + (3u * r) * kRecord) =
                    make_uint4((prev & 0xFFFFu) | (prev != goal ? 0x10000u : 0u), 0u, 0u, 0x3FF00000u);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<uint32_t *>(mine + kQueryHeader + (3u * r + uint32_t(k)) * kRecord + 4) = mask[k];
            }
        }
        if (r == 0u) {                                              // the group's first lane: header, counts, the window's length
            QueryHeader hdr{};
            if (live) {
                uint32_t lo;
                rows = window_rows(count, p.first_branch, p.max_branches, lo);
                if (p.out_count && piece == 0u) p.out_count[q] = count;
                hdr.first = lo;
                hdr.terminal = terminal ? 1u : 0u;
                hdr.living = p.c.criteria == 1u ? __dmul_rn(double(int(A) - stayed), p.c.r_living) : p.c.r_living;
                hdr.radix = terminal ? (0x55555555u >> (32 - 2 * MAXA)) : radix_all;
            }
            hdr.set_divisors(MAXA);
            *reinterpret_cast<QueryHeader *>(mine) = hdr;
        }
    } else
    // ---- set-up: lane l < QW owns query q0 + l
    if (lane < QW) {
        const uint64_t q = q0 + lane;
        unsigned char *const mine = queries + lane * kQueryBytes;
        QueryHeader hdr{};
        if (q < p.n_queries) {
            const uint64_t env = p.env_index ? p.env_index[q] : 0;
            const uint16_t *goal_row = p.goal + (p.goal_broadcast ? 0 : env * A);
            const uint16_t *state_row = p.local + q * A;
            const uint8_t *act_row = p.actions + q * A;
            // (each agent's records are written as soon as its list is known -- cells, off-goal bits, probabilities -- and only
            // the cells stay in registers for the conflict masks, which are filled in afterwards: held all at once, the 24
            // probabilities alone were 48 registers and the kernel lost a wave per SIMD to its set-up)
            uint32_t prev[MAXA], n[MAXA], cell[MAXA][3];
            uint32_t count = 1u, dup_acc = 0xFFFFFFFFu, goal_acc = 0u, off_goal_prev = 0u;
            int stayed = 0;
#pragma unroll
            for (int i = 0; i < MAXA; ++i) {
                const bool on = uint32_t(i) < A;
                prev[i] = on ? state_row[i] : 0x10000u + uint32_t(i);   // absent agents: unique cells, never equal to a real one
                const uint32_t goal = on ? goal_row[i] : prev[i];
                const uint32_t a = on ? act_row[i] : 0u, act = a > 4u ? 0u : a;
                const MoveEntry entry = on ? move_entry(p.mv, p.c.n_cells, prev[i], act) : MoveEntry{0u, 0u, 0u, 7u * uint32_t(sizeof(SlipRow))};
                const SlipRow &row = slip[on ? entry_code(entry) : 7u];
                n[i] = on ? row.n : 1u;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t c = on ? entry_cell(entry, uint32_t(k)) : prev[i];
                    const double pr = on ? row.q[k] : 1.0;          // (x * 1.0 is exact: absent agents leave the product alone)
                    *reinterpret_cast<uint4 *>(mine + kQueryHeader + (3 * i + k) * kRecord) =
                        make_uint4((c & 0xFFFFu) | (c != goal ? 0x10000u : 0u), 0u, uint32_t(__double2loint(pr)), uint32_t(__double2hiint(pr)));
                    // for the conflict masks: an entry past the list's end (never chosen) gets a cell nothing can equal
                    cell[i][k] = uint32_t(k) < n[i] ? c : 0x20000u + uint32_t(3 * i + k);
                }
                count *= n[i];
                goal_acc |= prev[i] ^ goal;
                off_goal_prev |= (prev[i] != goal ? 1u : 0u) << i;
                stayed += (on && prev[i] == goal && act == 0u) ? 1 : 0;
            }
#pragma unroll
            for (int i = 0; i < MAXA; ++i)
#pragma unroll
                for (int j = i + 1; j < MAXA; ++j) dup_acc = min(dup_acc, prev[i] ^ prev[j]);
            const bool terminal = dup_acc == 0u || goal_acc == 0u;     // is_terminal: mapf_env.py:210-223
            if (terminal) count = 1u;
            if (p.out_count && piece == 0u) p.out_count[q] = count;
            uint32_t lo;
            rows = window_rows(count, p.first_branch, p.max_branches, lo);
            hdr.first = lo;
            hdr.terminal = terminal ? 1u : 0u;
            hdr.living = p.c.criteria == 1u ? __dmul_rn(double(int(A) - stayed), p.c.r_living) : p.c.r_living;
#pragma unroll
            for (int i = 0; i < MAXA; ++i) hdr.radix |= (terminal ? 1u : n[i]) << (2 * i);
            if (terminal) {   // the single branch ((1.0, False), s, 0, True): every agent's one entry is its own cell, probability 1.0
#pragma unroll
                for (int i = 0; i < MAXA; ++i)
                    *reinterpret_cast<uint4 *>(mine + kQueryHeader + (3 * i) * kRecord) =
                        make_uint4((prev[i] & 0xFFFFu) | (((off_goal_prev >> i) & 1u) << 16), 0u, 0u, 0x3FF00000u);
            } else {
#pragma unroll
                for (int i = 0; i < MAXA; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        // entry k of agent i against every entry of every later agent: vertex (same next cell) or swap (each moves
                        // onto the other's current cell) -- _is_collision_transition_from_local_states, mapf_env.py:378-389
                        uint32_t mask = 0u;
#pragma unroll
                        for (int j = i + 1; j < MAXA; ++j)
#pragma unroll
                            for (int kj = 0; kj < 3; ++kj) {
                                // (integer arithmetic: as booleans the 252 tests of a query live in scalar mask pairs and spill them;
                                // entries past a list's end were given cells no agent can have, so no length test is needed)
                                const uint32_t t = min(cell[i][k] ^ cell[j][kj], (prev[i] ^ cell[j][kj]) | (prev[j] ^ cell[i][k]));
                                mask |= ((t - 1u) >> 31) << (3 * j + kj);   // t == 0 <=> they collide (t < 2^31)
                            }
                        *reinterpret_cast<uint32_t *>(mine + kQueryHeader + (3 * i + k) * kRecord + 4) = mask;
                    }
            }
        }
        hdr.set_divisors(MAXA);
        *reinterpret_cast<QueryHeader *>(mine) = hdr;
    }
    const uint32_t incl = wave_inclusive_scan(rows, lane);
    // (COOP: a query's length sits in its group's first lane, the other lanes contribute zero)
    const uint32_t g_shift = COOP ? 6u - qw_log2 : 0u;
    const bool reports = COOP ? (lane & ((1u << g_shift) - 1u)) == 0u : lane < QW;
    const uint32_t my_query = lane >> g_shift;
    if (reports) prefix[my_query] = incl - rows;
    const uint32_t total = uint32_t(__shfl(int(incl), 63, 64));
    // compacted rows: the wave's queries are consecutive, so their rows are one contiguous range from the first one's offset
    const bool compact = p.compact;
    if (compact && piece == 0u && p.out_offset && reports && q0 + my_query < p.n_queries) p.out_offset[q0 + my_query] = base + (incl - rows);
    // (the wave's lanes wrote the records / prefix sums the others read below: same wave, LDS operations execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- emission: one lane per row of the wave's windows
    const double r_clash = p.c.r_clash, r_goal = p.c.r_goal;
    const uint32_t row_end = (pieces_max > 1u && total - row_begin > rows_per_wave && row_begin < total) ? row_begin + rows_per_wave : total;
    // The wave's output rows as 32-bit offsets from wave-uniform bases held in scalar registers (compacted: the batch's first row;
    // reserved: its first query's slot): row index, capacity test and the five addresses were 64-bit vector arithmetic per row --
    // three v_mad_u64_u32 among it -- in a loop that is bound by its vector instructions (the launch without any store takes
    // 0.124 of its 0.19 ms: tools/exp/transitions_outputs.py).  (A reserved layout whose slots are wider than 2^20 rows keeps
    // the 64-bit form: lane offsets of pos * max_branches rows would not fit.)
    const bool narrow = compact || p.max_branches <= (1u << 20);
    const uint64_t rows0_v = compact ? base : q0 * uint64_t(p.max_branches);
    const uint64_t rows0 = (uint64_t(__builtin_amdgcn_readfirstlane(int(uint32_t(rows0_v >> 32)))) << 32) |
                           uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(rows0_v))));
    const uint64_t room = p.capacity > rows0 ? p.capacity - rows0 : 0u;
    const uint32_t cap_rel = room > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(room);
    unsigned char *const next0 = reinterpret_cast<unsigned char *>(p.out_next) + rows0 * A * 2u;
    unsigned char *const prob0 = reinterpret_cast<unsigned char *>(p.out_prob) + rows0 * 8u;
    unsigned char *const reward0 = reinterpret_cast<unsigned char *>(p.out_reward) + rows0 * 8u;
    unsigned char *const done0 = reinterpret_cast<unsigned char *>(p.out_done) + rows0;
    unsigned char *const coll0 = reinterpret_cast<unsigned char *>(p.out_collision) + rows0;
    for (uint32_t r0 = row_begin; r0 < row_end; r0 += 64u) {
        const uint32_t r = r0 + lane;
        const bool valid = r < row_end;
        uint32_t pos = 0u;                                         // the row's query: the last one whose prefix is <= r
        for (uint32_t step = QW >> 1; step != 0u; step >>= 1) {
            const uint32_t cand = pos + step;
            pos = prefix[cand] <= r ? cand : pos;
        }
        const unsigned char *const qbase = queries + pos * kQueryBytes;
        const uint4 h = *reinterpret_cast<const uint4 *>(qbase);   // {radix, first, terminal, -}
        const uint4 h2 = *reinterpret_cast<const uint4 *>(qbase + 16);   // {living, the list lengths as bytes}
        const uint4 mg = *reinterpret_cast<const uint4 *>(qbase + 32);   // the divisors' magics, a 16-bit word each
        const double living = __hiloint2double(int(h2.y), int(h2.x));
        const uint32_t nb[2] = {h2.z, h2.w}, mw[4] = {mg.x, mg.y, mg.z, mg.w};
        const uint32_t j = r - prefix[pos];                        // row of the query's window
        uint32_t rest = j + h.y;                                   // branch index: digits in the mixed radix, last agent fastest
        uint32_t chosen = 0u, hit = 0u, flags = 0u, cells[MAXA];
        double qv[MAXA];
        static_assert(MAXA <= 8, "3^MAXA < 2^15: the magic division is exact");
        auto take = [&](auto tag) __attribute__((always_inline)) {
            constexpr int I = decltype(tag)::value;
            const uint32_t quot = mul24_word<I & 1>(rest, mw[I / 2]) >> 15;          // rest / n_I
            const uint32_t d = rest - mul24_byte<I & 3>(quot, nb[I / 4]);            // rest % n_I: agent I's entry
            rest = quot;
            const uint4 rec = *reinterpret_cast<const uint4 *>(qbase + kQueryHeader + uint32_t(3 * I) * kRecord + d * kRecord);
            hit |= rec.y & chosen;                                 // my entry against the entries the later agents chose
            chosen |= (1u << (3 * I)) << d;
            flags |= rec.x;                                        // bit 16: off its goal
            cells[I] = rec.x & 0xFFFFu;
            qv[I] = __hiloint2double(int(rec.w), int(rec.z));
        };
        if constexpr (MAXA >= 8) { take(std::integral_constant<int, 7>{}); take(std::integral_constant<int, 6>{}); }
        if constexpr (MAXA >= 6) { take(std::integral_constant<int, 5>{}); take(std::integral_constant<int, 4>{}); }
        if constexpr (MAXA >= 4) { take(std::integral_constant<int, 3>{}); take(std::integral_constant<int, 2>{}); }
        take(std::integral_constant<int, 1>{});
        take(std::integral_constant<int, 0>{});
        double prob = qv[0];                                       // left to right: functools.reduce at mapf_env.py:467
#pragma unroll
        for (int i = 1; i < MAXA; ++i) prob = __dmul_rn(prob, qv[i]);
        const bool terminal = h.z != 0u, coll = hit != 0u, goal_next = (flags & 0x10000u) == 0u;
        const double reward = terminal ? 0.0 : (coll ? __dadd_rn(r_clash, living) : (goal_next ? __dadd_rn(r_goal, living) : living));
        auto store_row = [&](uint16_t *dst, double *prob_at, double *reward_at, uint8_t *done_at, uint8_t *coll_at) __attribute__((always_inline)) {
            if (ALL_OUT || p.out_next) {
                if (A == uint32_t(MAXA)) {                          // a full team: one store per row (rows are 2 MAXA bytes apart)
                    if constexpr (MAXA == 8) *reinterpret_cast<uint4 *>(dst) = make_uint4(cells[0] | cells[1] << 16, cells[2] | cells[3] << 16, cells[4] | cells[5] << 16, cells[6] | cells[7] << 16);
                    else if constexpr (MAXA == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(cells[0] | cells[1] << 16, cells[2] | cells[3] << 16);
                    else {
#pragma unroll
                        for (int i = 0; i + 1 < MAXA; i += 2) reinterpret_cast<uint32_t *>(dst)[i / 2] = cells[i] | cells[i + 1] << 16;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < MAXA; ++i)
                        if (uint32_t(i) < A) dst[i] = uint16_t(cells[i]);
                }
            }
            if (ALL_OUT || p.out_prob) *prob_at = prob;
            if (ALL_OUT || p.out_reward) *reward_at = reward;
            if (ALL_OUT || p.out_done) *done_at = (terminal || coll || goal_next) ? 1 : 0;
            if (ALL_OUT || p.out_collision) *coll_at = (coll && !terminal) ? 1 : 0;
        };
        if (narrow) {
            const uint32_t rel = compact ? r : pos * p.max_branches + j;   // rows from the wave's base
            if (valid && rel < cap_rel)
                store_row(reinterpret_cast<uint16_t *>(next0 + uint64_t(rel * (A * 2u))), reinterpret_cast<double *>(prob0 + uint64_t(rel * 8u)),
                          reinterpret_cast<double *>(reward0 + uint64_t(rel * 8u)), done0 + uint64_t(rel), coll0 + uint64_t(rel));
        } else {
            const uint64_t o = (q0 + pos) * uint64_t(p.max_branches) + j;
            if (valid && o < p.capacity) store_row(p.out_next + o * A, p.out_prob + o, p.out_reward + o, p.out_done + o, p.out_collision + o);
        }
    }
}

// one branch's next cells, out[row * A .. row * A + A): the widest stores the row's alignment allows when the team
// fills the instance, two bytes at a time otherwise
template <int MAXA>
__device__ __forceinline__ void store_branch_cells(uint16_t *out, uint64_t row, uint32_t A, const uint32_t (&cells)[MAXA]) {
    uint16_t *dst = out + row * A;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out);
    if (MAXA % 4 == 0 && A == uint32_t(MAXA) && (base & 7u) == 0u) {
#pragma unroll
        for (int i = 0; i < MAXA; i += 4)
            *reinterpret_cast<uint2 *>(dst + i) = make_uint2(cells[i] | (cells[i + 1] << 16), cells[i + 2] | (cells[i + 3] << 16));
    } else if (MAXA % 2 == 0 && A == uint32_t(MAXA) && (base & 3u) == 0u) {
#pragma unroll
        for (int i = 0; i < MAXA; i += 2) *reinterpret_cast<uint32_t *>(dst + i) = cells[i] | (cells[i + 1] << 16);
    } else {
#pragma unroll
        for (int i = 0; i < MAXA; ++i)
            if (uint32_t(i) < A) dst[i] = uint16_t(cells[i]);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- 9..16 agents
// EXACT: the team has exactly MAXA agents (no absent slots, no per-slot predicates: the 9..16-agent instances would
// otherwise keep sixteen wave masks alive and spill scalar registers).
//
// Work split: a GROUP of `lanes` (a power of two <= 64) consecutive lanes owns one chunk of `chunk` window rows of one
// query and walks it `lanes` rows at a time.  Everything that depends on the query only is set up ONCE per lane and reused
// for every branch the lane emits; lanes whose rows lie beyond the query's branch count leave after that set-up.
template <int MAXA, bool EXACT = false>
__global__ void __launch_bounds__(256) transitions_kernel(const TransitionsArgs p, const uint32_t lanes_log2, const uint32_t chunk,
                                                          const uint32_t chunks_per_query) {
    __shared__ SlipRow slip[8];
    stage_slip_table(p.slip, slip);
    const uint64_t gid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t group = gid >> lanes_log2;
    const uint32_t lane = uint32_t(gid) & ((1u << lanes_log2) - 1u);
    uint64_t q;
    uint32_t piece;                                              // which chunk of the query's window
    if (chunks_per_query == 1u) { q = group; piece = 0u; }
    else if (group <= 0xFFFFFFFFull) { const uint32_t g32 = uint32_t(group), quot = g32 / chunks_per_query; q = quot; piece = g32 - quot * chunks_per_query; }
    else { q = group / chunks_per_query; piece = uint32_t(group - q * chunks_per_query); }
    if (q >= p.n_queries) return;
    const uint32_t A = EXACT ? uint32_t(MAXA) : p.n_agents;
    const uint64_t env = p.env_index ? p.env_index[q] : 0;
    const uint16_t *goal_row = p.goal + (p.goal_broadcast ? 0 : env * A);
    const uint16_t *state_row = p.local + q * A;
    const uint8_t *act_row = p.actions + q * A;

    uint32_t prev[MAXA], goal[MAXA], n[MAXA];
    MoveEntry entry[MAXA];
    uint32_t dup_acc = 0xFFFFFFFFu, goal_acc = 0u;
    uint32_t count = 1;                                          // <= 3^16
    int stayed = 0;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        const bool on = EXACT || uint32_t(i) < A;
        prev[i] = on ? state_row[i] : 0x10000u + uint32_t(i);       // ghosts: unique, never equal to a real cell
        goal[i] = on ? goal_row[i] : prev[i];
        const uint32_t a = on ? act_row[i] : 0u;
        const uint32_t act = a > 4u ? 0u : a;
        entry[i] = on ? move_entry(p.mv, p.c.n_cells, prev[i], act) : MoveEntry{0u, 0u, 0u, 0u};
        n[i] = on ? slip[entry_code(entry[i])].n : 1u;
        count *= n[i];
        goal_acc |= prev[i] ^ goal[i];
        stayed += (on && prev[i] == goal[i] && act == 0u) ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i < MAXA; ++i)
#pragma unroll
        for (int j = i + 1; j < MAXA; ++j) dup_acc = min(dup_acc, prev[i] ^ prev[j]);
    const bool terminal = dup_acc == 0u || goal_acc == 0u;           // is_terminal: mapf_env.py:210-223
    if (terminal) count = 1;
    if (piece == 0u && lane == 0u && p.out_count) p.out_count[q] = count;

    const uint32_t piece_begin = piece * chunk;                  // (piece < chunks_per_query, so this stays below max_branches)
    const uint32_t piece_end = chunk < p.max_branches - piece_begin ? piece_begin + chunk : p.max_branches;
    // the query's first output row: reserved rows, or its offset in the compacted arrays
    uint64_t row0 = q * p.max_branches;
    if (p.compact) {
        row0 = p.block_base[q / kScanBlock] + p.rel[q];
        if (piece == 0u && lane == 0u && p.out_offset) p.out_offset[q] = row0;
    }
    if (terminal) {                                              // the single branch ((1.0, False), s, 0, True)
        if (piece == 0u && lane == 0u && p.first_branch == 0u && row0 < p.capacity) {
            if (p.out_next) store_branch_cells<MAXA>(p.out_next, row0, A, prev);
            if (p.out_prob) p.out_prob[row0] = 1.0;
            if (p.out_reward) p.out_reward[row0] = 0.0;
            if (p.out_done) p.out_done[row0] = 1;
            if (p.out_collision) p.out_collision[row0] = 0;
        }
        return;
    }
    double living = p.c.r_living;                                // _living_reward: mapf_env.py:436-446
    if (p.c.criteria == 1u) living = __dmul_rn(double(int(A) - stayed), p.c.r_living);
    const double r_coll = __dadd_rn(p.c.r_clash, living), r_goal = __dadd_rn(p.c.r_goal, living);

    for (uint32_t slot = piece_begin + lane; slot < piece_end; slot += 1u << lanes_log2) {
        const uint64_t b = p.first_branch + slot;                // branch index in the query's full enumeration
        if (b >= count) break;
        // digits of b, last agent fastest.  b < count <= 3^16 fits 32 bits and every radix is 1, 2 or 3: the quotient is a
        // select between rest, rest >> 1 and a multiply-high by the reciprocal of 3
        uint32_t next[MAXA];
        uint32_t rest = uint32_t(b);
        double qv[MAXA];
        uint32_t goal_next_acc = 0u;
#pragma unroll
        for (int i = MAXA - 1; i >= 0; --i) {
            const bool on = EXACT || uint32_t(i) < A;
            const uint32_t third = __umulhi(rest, 0xAAAAAAABu) >> 1;
            const uint32_t quot = n[i] == 3u ? third : (n[i] == 2u ? rest >> 1 : rest);
            const uint32_t k = rest - quot * n[i];
            rest = quot;
            next[i] = on ? entry_cell(entry[i], k) : prev[i];
            qv[i] = on ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(slip) + entry_row_offset(entry[i]) + k * 8u) : 1.0;
            goal_next_acc |= next[i] ^ goal[i];
        }
        double prob = qv[0];                                     // left to right: functools.reduce at mapf_env.py:467
#pragma unroll
        for (int i = 1; i < MAXA; ++i) prob = (EXACT || uint32_t(i) < A) ? __dmul_rn(prob, qv[i]) : prob;
        uint32_t coll_acc = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
#pragma unroll
            for (int j = i + 1; j < MAXA; ++j) {
                // vertex: next_i == next_j; swap: prev_i == next_j and prev_j == next_i.  Integer min-of-xor accumulators
                // (no wave-mask booleans); ghost cells are >= 0x10000 and unique, so they never produce a zero.
                const uint32_t swap = (prev[i] ^ next[j]) | (prev[j] ^ next[i]);
                coll_acc = min(coll_acc, min(next[i] ^ next[j], swap));
            }
        }
        const bool coll = coll_acc == 0u, goal_next = goal_next_acc == 0u;
        // compacted rows hold the window's rows only: row `slot` of the window is its slot-th row there too
        const uint64_t o = row0 + slot;
        if (o >= p.capacity) break;
        if (p.out_next) store_branch_cells<MAXA>(p.out_next, o, A, next);
        if (p.out_prob) p.out_prob[o] = prob;
        if (p.out_reward) p.out_reward[o] = coll ? r_coll : (goal_next ? r_goal : living);
        if (p.out_done) p.out_done[o] = (coll || goal_next) ? 1 : 0;
        if (p.out_collision) p.out_collision[o] = coll ? 1 : 0;
    }
}

// MapfEnv.calc_transition_reward_from_local_states (mapf_env.py:225-235) for N given (prev, joint action, next)
// triples: _living_reward (:436-446), then collision (:378-389, vertex or swap over every agent pair) before goal.
// Unlike step() it does not look at is_terminal(prev) -- neither does the reference method.  One thread per query,
// run-time A (the reference's own double loop; queries are independent, so lanes never diverge on the trip count).
__global__ void __launch_bounds__(256) transition_reward_kernel(const TransitionsArgs p, const uint16_t *next) {
    const uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= p.n_queries) return;
    const uint32_t A = p.n_agents;
    const uint64_t env = p.env_index ? p.env_index[q] : 0;
    const uint16_t *goal = p.goal + (p.goal_broadcast ? 0 : env * A);
    const uint16_t *prev = p.local + q * A, *nxt = next + q * A;
    const uint8_t *act = p.actions + q * A;
    uint32_t coll_acc = 0xFFFFFFFFu, goal_next_acc = 0u;
    int stayed = 0;
    for (uint32_t i = 0; i < A; ++i) {
        const uint32_t pi = prev[i], ni = nxt[i], a = act[i] > 4u ? 0u : act[i];
        stayed += (pi == goal[i] && a == 0u) ? 1 : 0;
        goal_next_acc |= ni ^ goal[i];
        for (uint32_t j = i + 1; j < A; ++j) {
            const uint32_t pj = prev[j], nj = nxt[j];
            coll_acc = min(coll_acc, min(ni ^ nj, (pi ^ nj) | (pj ^ ni)));
        }
    }
    const bool coll = coll_acc == 0u, goal_next = goal_next_acc == 0u;
    double living = p.c.r_living;
    if (p.c.criteria == 1u) living = __dmul_rn(double(int(A) - stayed), p.c.r_living);
    if (p.out_reward) p.out_reward[q] = coll ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
    if (p.out_done) p.out_done[q] = (coll || goal_next) ? 1 : 0;
    if (p.out_collision) p.out_collision[q] = coll ? 1 : 0;
}

hipError_t launch_transition_rewards(const TransitionsArgs &args, const uint16_t *next, hipStream_t stream) {
    if (args.n_queries == 0) return hipSuccess;
    const uint64_t grid64 = (args.n_queries + 255) / 256;
    if (grid64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transition_reward_kernel, dim3(unsigned(grid64)), dim3(256), 0, stream, args, next);
    return hipGetLastError();
}

uint64_t transitions_scan_blocks(uint64_t n_queries) { return (n_queries + kScanBlock - 1) / kScanBlock; }

// passes 1 and 2: args.rel / args.block_base (see TransitionsArgs), the total to args.out_offset[N] when compacting; also fills
// args.out_count
constexpr uint32_t kFusedScanBlocks = 256;   // calls of up to this many scan blocks leave pass 2 to the rows kernel's waves
static hipError_t launch_transitions_offsets(const TransitionsArgs &args, hipStream_t stream, bool scan = true) {
    const uint64_t blocks = transitions_scan_blocks(args.n_queries);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return blocks == 0 ? hipSuccess : hipErrorInvalidValue;
    hipLaunchKernelGGL(transitions_count_kernel, dim3(unsigned(blocks)), dim3(kScanBlock), 0, stream, args, args.rel, args.block_base);
    if (scan)
        hipLaunchKernelGGL(scan_block_totals_kernel, dim3(1), dim3(1024), 0, stream, args.block_base, uint32_t(blocks),
                           (args.compact && args.out_offset) ? args.out_offset + args.n_queries : nullptr);
    return hipGetLastError();
}

template <int MAXA>
static hipError_t launch_rows(const TransitionsArgs &args, hipStream_t stream) {
    constexpr uint32_t kQueryBytes = kQueryHeader + uint32_t(MAXA) * 3u * kRecord;
    // queries per wave: as many as keep a block's LDS near 32 KB (five or more blocks per CU), fewer while that leaves the
    // device short of waves (16 per SIMD queued) -- set-up uses one lane per query, so small teams want many per wave
    uint32_t qw_log2 = 6;
    while (qw_log2 > 2 && 4u * (kQueryBytes + 4u) * (1u << qw_log2) > 36u * 1024u) --qw_log2;
    while (qw_log2 > (MAXA >= 8 ? 2u : 3u) && (args.n_queries >> qw_log2) < 16384u) --qw_log2;
    if (MAXA >= 6 && qw_log2 > 3u) qw_log2 = 3u;                     // (cooperative set-up: a query's group has a lane per agent)
    // rows per wave: a batch's windows can hold QW x min(3^A, max_branches) rows; beyond two pieces' worth they are cut into
    // pieces -- which needs the scan, as compacted rows do.  Pieces of 1024 rows (16 sweeps) while the call is small, so that it
    // still makes ~24 K waves' worth of work (a room map fills about a fifth of the rows a query can have), up to 4096 for large
    // calls, where fewer set-ups per row count (profiles/r05_transitions_launch_shapes.txt: 20000 queries of 8 agents 0.43 of the
    // roofline with 1024-row pieces, 0.37 with 4096; 200000 queries 0.54 against 0.60-0.68)
    uint64_t most = 1;
    for (uint32_t i = 0; i < args.n_agents && most < args.max_branches; ++i) most *= 3u;
    if (most > args.max_branches) most = args.max_branches;
    const uint64_t est_rows = args.n_queries * most / 5u;
    uint32_t rows_per_wave = 1024u;
    while (rows_per_wave < 4096u && est_rows / (2u * rows_per_wave) >= 24576u) rows_per_wave *= 2u;
    most <<= qw_log2;
    // (the small teams' batches -- at most 64 x 81 rows, set up by one lane per query -- stay whole: their waves all do the same work)
    const uint32_t pieces_max = (MAXA >= 6 && most > 2u * rows_per_wave) ? uint32_t((most + rows_per_wave - 1) / rows_per_wave) : 1u;
    const uint64_t scan_blocks = transitions_scan_blocks(args.n_queries);
    const uint32_t unscanned_blocks = scan_blocks <= kFusedScanBlocks ? uint32_t(scan_blocks) : 0u;
    if (args.compact || pieces_max > 1u) {
        if (hipError_t e = launch_transitions_offsets(args, stream, unscanned_blocks == 0u)) return e;
    }
    const uint64_t waves = ((args.n_queries + (1u << qw_log2) - 1) >> qw_log2) * pieces_max;
    // waves per block: ONE for the large teams -- their waves live for very different times (a piece that does not exist leaves
    // at once, a full one sweeps 64 times) and a block keeps its place until its last wave is done: four waves per block
    // measured 0.407 ms against 0.277 ms at 8 agents x 20000 queries (profiles/r05_transitions_launch_shapes.txt); the small
    // teams' waves all do the same work and share a block's slip table
    const unsigned wpb = MAXA >= 6 ? 1u : 4u;
    const uint64_t grid64 = (waves + wpb - 1) / wpb;
    if (grid64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = wpb * size_t(kQueryBytes + 4u + (MAXA >= 6 ? MAXA * 16u : 0u)) * (size_t(1) << qw_log2);
    const bool all_out = args.out_next && args.out_prob && args.out_reward && args.out_done && args.out_collision;
    note_kernel("transitions_rows_kernel<%d> %u agents, %u queries per wave%s, %s rows", MAXA, args.n_agents, 1u << qw_log2,
                pieces_max > 1u ? (rows_per_wave == 4096u ? " in pieces of 4096 rows" : (rows_per_wave == 2048u ? " in pieces of 2048 rows" : " in pieces of 1024 rows")) : "",
                args.compact ? "compacted" : "reserved");
    if (all_out) hipLaunchKernelGGL((transitions_rows_kernel<MAXA, true>), dim3(unsigned(grid64)), dim3(64 * wpb), lds, stream, args, qw_log2, pieces_max, rows_per_wave, unscanned_blocks);
    else hipLaunchKernelGGL((transitions_rows_kernel<MAXA, false>), dim3(unsigned(grid64)), dim3(64 * wpb), lds, stream, args, qw_log2, pieces_max, rows_per_wave, unscanned_blocks);
    return hipGetLastError();
}

hipError_t launch_transitions(const TransitionsArgs &args, hipStream_t stream) {
    if (args.n_queries == 0 || args.max_branches == 0) return hipSuccess;
    if (args.n_agents <= 2) return launch_rows<2>(args, stream);
    if (args.n_agents <= 4) return launch_rows<4>(args, stream);
    if (args.n_agents <= 6) return launch_rows<6>(args, stream);
    if (args.n_agents <= 8) return launch_rows<8>(args, stream);
    if (args.compact) {
        if (hipError_t e = launch_transitions_offsets(args, stream)) return e;
    }
    // lanes per group: a wave; a group walks up to 16 x lanes rows, a long window is cut into that many-row chunks (so
    // that a single query of a large team still fills the device)
    uint32_t lanes_log2 = 0;
    while (lanes_log2 < 6 && (1u << lanes_log2) < args.max_branches) ++lanes_log2;
    const uint32_t lanes = 1u << lanes_log2;
    const uint32_t walks = std::min<uint32_t>(16u, (args.max_branches + lanes - 1) / lanes);
    const uint32_t chunk = lanes * walks;
    const uint32_t chunks_per_query = (args.max_branches + chunk - 1) / chunk;
    const uint64_t threads = args.n_queries * uint64_t(chunks_per_query) * lanes;
    const uint64_t grid64 = (threads + 255) / 256;
    if (grid64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid{unsigned(grid64)}, block{256};
    note_kernel("transitions_kernel<%d,EXACT> %u agents, %u lanes x %u rows per group, %u groups per query, %s rows", int(args.n_agents),
                args.n_agents, lanes, walks, chunks_per_query, args.compact ? "compacted" : "reserved");
    switch (args.n_agents) {
#define X(N) case N: hipLaunchKernelGGL((transitions_kernel<N, true>), grid, block, 0, stream, args, lanes_log2, chunk, chunks_per_query); break;
        X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mapf
