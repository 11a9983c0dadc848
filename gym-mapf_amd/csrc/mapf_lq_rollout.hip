// Packed-lane rollout kernel: K = 2, 4 or 8 agents per lane (one, two or four packed cell pairs), Q = A/K lanes per env.
//
// Why a second lane layout beside mapf_lg.hpp's: there everything that is per ENV -- flag reduction, outcome lookup,
// totals, reset handling, the hand-over steps of the probability product -- is replicated over the L = A/2 lanes of a
// group, and that part is about 40 % of a step's vector instructions at A = 8.  Four agents per lane halve the lanes
// per env, so the replicated part halves, the in-lane half of the pair tests needs no cross-lane move at all, and
// each lane carries two independent Philox calls / four independent table gathers.
//
// How the step loop is written (measured: tools/microbench/single_wave_latency.hip, profiles/r02_single_wave_costs.txt).
// At the sizes that matter only one or two waves share a SIMD, and a wave alone issues ONE instruction per ~4.6 cycles
// whatever its kind or dependences; a scalar branch costs ~12 cycles when it falls through and ~25 when taken, and a
// vector compare whose mask goes through a scalar AND/OR back into a vector select ~15 on top.  So the loop
//   * is unrolled by FOUR steps aligned to the slip stream's call blocks (two calls of two steps each, refreshed
//     together in lockstep): which word a step uses, and whether it refreshes the calls, are compile-time facts there
//     (generic steps run before / after the aligned part of a launch);
//   * derives the per-env facts with integer arithmetic in vector registers (zero-half-word tests, min / shifts)
//     and feeds selects from VCC written by the instruction before them -- no scalar mask algebra in the loop;
//   * keeps the auto-reset / terminal bookkeeping as one integer code = vertex | swap << 1 | off_goal << 2 |
//     was_terminal << 3 per env that indexes the LDS outcome table and, compared against wave-uniform constants,
//     drives every select;
//   * samples an agent's list slot with packed 16-bit arithmetic (sample_slot_packed in mapf_lq.hpp): both threshold
//     compares are one saturating v_pk_sub_i16 of bias-shifted operands, and the slot's probability address and cell
//     selector each come out of one v_dot2_i32_i16 of the sign halves -- the kernel is bound by vector-instruction issue
//     (SQ_ACTIVE_INST_VALU ~ 98 % of the SIMD's cycles at two waves per SIMD), so instructions are what is saved;
//   * owns the whole LDS image (slip rows at 0, outcome rows at 768, move table at 1024), so every LDS address is a
//     register plus an immediate offset, and stages the move table with SIX columns per cell (column 5 = STAY again):
//     an action byte is extracted and clamped by one v_min_u32 with a byte select, a table address is
//     cell * 96 + action * 16.
//
// COMPACT form for maps whose full table does not fit (64x64 maps: ~3300 free cells): the LDS table keeps only the first
// 8 bytes of each 16-byte row (the three cells and the equality code; five columns), one block per CU owns up to 158 KB
// of it, and the code's thresholds come from a second, dependent LDS read of the code's slip row.
//
// 32 agents on such maps (BASELINE configs[4]): four agents per lane, eight lanes per env, and two things that are O(A)
// instead of what the other instances do -- the vertex / swap facts through a per-env ONE-BIT occupancy bitmap in LDS behind
// the table (BITMAP, bitmap_pair_tests in mapf_lq.hpp: three LDS operations per agent instead of 496 agent pairs per env;
// behind 4-byte delta rows where the map's ids allow them, else 128 bitmaps behind a four-column table of 8-byte rows in
// 1024-thread blocks, 64 behind the five-column one in 512-thread blocks), and the
// ordered probability product as a systolic chain over the steps (SYS below: one hand-over per step and lane instead of
// seven).  DESIGN.md section 4.1 has the measurements of each step.
//
// Scope: the fused rollout of FULL groups only (A = K * Q, Q a power of two <= 16), every block full, move table in
// LDS -- the bench configurations and their neighbours.  Everything else (odd agent counts, ragged batches, tables
// beyond the LDS budget, single steps) stays with mapf_lg_rollout.hip; launch_rollout_lg() picks.  Same stream,
// same arithmetic, same outputs: the parity tests run all layouts against the oracle.
#include "mapf_lq.hpp"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace mapf {

namespace {

constexpr size_t kLdsBytes = 160 * 1024, kLdsReserve = 1024;
static_assert(sizeof(SlipRow) * 8 + sizeof(OutcomeRow) * 16 <= kLdsReserve, "static LDS of the rollout kernel");

// RECORD: all five trajectory arrays are written every step; STREAM: actions come from memory, else from the
// in-kernel policy.  Memory pipeline and store scheme as lg_rollout_kernel<DENSE>.
constexpr uint32_t kSlipAt = 0, kOutcomeAt = sizeof(SlipRow) * 8, kMoveAt = kLdsReserve, kMoveCols = 6;
constexpr uint32_t kCompactCols = 5, kCompactEntry = 8;   // COMPACT: cells + code only, no sixth column
constexpr uint32_t kBitmapCols = 4;                       // COMPACT + BITMAP == 1: no STAY column either
constexpr uint32_t kDeltaEntry = 4;                       // COMPACT + BITMAP == 3: 4-byte delta rows, six columns (kDeltaCols, mapf_kernels.hpp: STAY twice, as the full table)
static_assert(kOutcomeAt + sizeof(OutcomeRow) * 16 <= kMoveAt, "LDS image: slip rows, outcome rows, then the move table");
// The LDS copy of a table row carries its slip row's byte offset PLUS kRowBias, so that sample_slot_packed's probability
// address -- that operand minus 8 per threshold not passed -- is never negative and packs into an unsigned field (the
// systolic probability chain below files four of them per word); the immediates of the LDS reads absorb the bias.
constexpr uint32_t kRowBias = kDeltaRowBias;              // (the host-built delta rows carry it too)
// index (in doubles from kSlipAt) of a +0.0: the all-equal code's list has ONE entry, so thr[1] of its row is the integer 0
constexpr uint32_t kZeroFactor = (7u * uint32_t(sizeof(SlipRow)) + uint32_t(offsetof(SlipRow, thr)) + 8u) / 8u;
static_assert(offsetof(SlipRow, thr) % 8 == 0 && kZeroFactor < 128u, "a zero factor the packed probability indices can name");

// TERM = an env may be terminal when a step begins.  With auto-reset on and no env whose START state is itself
// terminal (the handle knows: mapf_create looks) that cannot happen after the launch's first step -- a done env is back
// on its start cells -- and the !TERM instance runs every later step without the was-terminal selects.
// BITMAP = the vertex / swap facts come from a per-env LDS occupancy bitmap (bitmap_pair_tests in mapf_lq.hpp) instead of
// all agent pairs: O(A) instead of O(A^2) -- the 32-agent configurations, where 496 pairs were three quarters of a step.
// The bitmaps (one per env of the block, ceil(V / 32) words each) follow the move table in the LDS image at `bitmap_base`.
// BITMAP == 1: the table has FOUR columns (the moves; a STAY row is made up in registers) -- the form that leaves room for 128
// bitmaps, i.e. 1024-thread blocks; BITMAP == 2: five columns (STAY included: no selects per agent), 64 bitmaps, 512 threads.
// BITMAP == 3: 4-BYTE rows -- the three candidates as signed byte DELTAS against the row's own cell (a neighbour's id differs
// from a cell's by less than a column's height, which mapf_create checks: RolloutArgs::mv_delta8) plus the slip row's offset
// in the fourth byte -- so that SIX columns (STAY twice: an action byte is extracted and clamped by one v_min_u32, and no STAY
// row is made up) take half the room of the five 8-byte ones: 128 bitmaps fit behind them on the 64x64 maps.
// (the four-column form with the in-kernel policy holds its actions across the table reads -- the made-up STAY row asks for
// them -- and does not fit the 128 registers of a 1024-thread block: launched with 512 threads, see try_launch_rollout_lq)
template <int Q, int K, bool RECORD, bool STREAM, bool SOC, bool COMPACT, bool TERM, int BITMAP = 0>
__global__ void __launch_bounds__((K == 8 || (COMPACT && BITMAP == 1 && !STREAM)) ? 512 : 1024) lq_rollout_kernel(const RolloutArgs p, const uint32_t n_agents, const uint32_t bitmap_base) {
    constexpr int P = K / 2;   // packed dwords per lane
    static_assert(K == 2 || K == 4 || K == 8, "two, four or eight agents per lane");
    // the kernel's LDS image is its dynamic segment, used as a raw scratchpad from LDS address 0 (LdsAbsolute, mapf_lq.hpp: no
    // static LDS object exists in this kernel): every offset below is an instruction immediate
    const LdsAbsolute lds;
    SlipRow *slip = lds_generic<SlipRow>(lds, kSlipAt);
    OutcomeRow *outcome = lds_generic<OutcomeRow>(lds, kOutcomeAt);
    MoveEntry *lds_mv = lds_generic<MoveEntry>(lds, kMoveAt);
    LaneCtx<Q> x;
    x.lane = threadIdx.x & 63u;
    x.g = x.lane & uint32_t(Q - 1);
    x.base = x.lane & ~uint32_t(Q - 1);
    x.e = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * uint32_t(64 / Q) + x.lane / uint32_t(Q);
    x.v0 = x.v1 = true;
    const uint32_t e = x.e;
    const uint32_t lane_cell = e * n_agents + uint32_t(K) * x.g;    // my first agent's element index
    const uint32_t fixed_cell = uint32_t(K) * x.g;                  // ... in a broadcast row

    uint32_t c[P], g[P], start_c[P];
    {
        const Packed<P> cells = Packed<P>::load(at(p.state, lane_cell));
        const Packed<P> gl = Packed<P>::load(at(p.goal, p.goal_broadcast ? fixed_cell : lane_cell));
        Packed<P> sc{};
        if (p.auto_reset) sc = Packed<P>::load(at(p.start, p.start_broadcast ? fixed_cell : lane_cell));
#pragma unroll
        for (int i = 0; i < P; ++i) { c[i] = cells.v[i]; g[i] = gl.v[i]; start_c[i] = sc.v[i]; }
    }
    // (requested HERE, with the state / goal / start rows and ahead of the table copy: a launch's fixed cost -- 8-12 us, a third
    // of a T = 32 launch, profiles/r05_rollout_T_sweep.txt -- is mostly memory round trips in a row, so they travel together)
    const uint8_t *act_lane = STREAM ? p.actions + lane_cell : nullptr;

    // Action words are fetched kAhead steps ahead of their use (four with four agents per lane, eight with two, whose
    // steps are shorter): the loaded-HBM round trip, with the trajectory stores of the same wave queued in front of it
    // (vmcnt counts loads and stores in order), is longer than two steps -- fetched two ahead, the two-agents-per-lane
    // form ran 23 % slower than with L2-resident actions, the wait for the action word being the largest stall left.
    // A step consumes the register that holds its row and reloads THAT register with row s+kAhead: a register is never
    // moved while its load is in flight (a move is a use, i.e. a wait for the round trip just requested).
    // kAhead == 4 (WORD_SLOTS): raw[k] holds the row of the next step whose index t has t & 3 == k -- the slip stream's word
    // index, which every step body knows statically; the launch's head steps (head_steps: h of them) have registers of
    // their own.  Invariant at the top of step s >= h: raw[(t_first+s+j) & 3] holds row min(s+j, last) (j < 4), act_lane
    // points at row min(s+3, last).
    // Otherwise (two agents per lane: kAhead == 8): raw[j] holds row s+j at the top of step s, act_lane points at row min(s+kAhead-1, last); a single step
    // outside the unrolled loop uses raw[0] and shifts the others down afterwards.
    constexpr uint32_t kAhead = K == 2 ? 8 : 4;
    // (Streamed actions only: an in-kernel policy has no action registers to keep still, and its instances -- the policy words of
    // four steps live across the loop -- spill under this loop's extra step bodies, whatever form the head takes: 9 to 45 of the
    // 96 recording instances with four or eight agents per lane did, up to 157 registers.  They keep the older loop.)
    constexpr bool WORD_SLOTS = STREAM && kAhead == 4;
    using RawWord = std::conditional_t<K == 8, uint64_t, uint32_t>;   // one action byte per agent of the lane
    const uint32_t last_row = p.n_steps ? p.n_steps - 1u : 0u;
    auto load_raw_at = [&](const uint8_t *at_row) __attribute__((always_inline)) {
        if constexpr (K == 8) return *reinterpret_cast<const uint64_t *>(at_row);
        else return K == 4 ? *reinterpret_cast<const uint32_t *>(at_row) : uint32_t(*reinterpret_cast<const uint16_t *>(at_row));
    };
    auto load_raw = [&]() __attribute__((always_inline)) { return load_raw_at(act_lane); };
    RawWord raw[kAhead] = {};
    RawWord raw_first = 0, raw_head[2] = {};
    // head steps: the first one, and those up to the first word boundary of a launch that does not start at one (or is
    // shorter than four steps) -- they have registers of their own, loaded with everything else at the kernel's start
    auto head_steps = [](const uint32_t t0, const uint32_t n) __attribute__((always_inline)) {
        return (t0 & 3u) == 0u ? (n >= kAhead ? 1u : n) : min(n, 4u - (t0 & 3u));
    };
    if (STREAM && p.n_steps > 0) {
        const uint64_t row_stride = uint64_t(uint32_t(p.n_envs)) * n_agents;
        if constexpr (WORD_SLOTS) {
            raw_first = load_raw();
            raw_head[0] = load_raw_at(act_lane + min(1u, last_row) * row_stride);
            raw_head[1] = load_raw_at(act_lane + min(2u, last_row) * row_stride);
            const uint32_t h = head_steps(uint32_t(first_step_index(p)), p.n_steps);
#pragma unroll
            for (uint32_t k = 0; k < kAhead; ++k)                     // clamped, not guarded: late rows are re-read
                raw[k] = load_raw_at(act_lane + min(h + ((k - uint32_t(first_step_index(p)) - h) & 3u), last_row) * row_stride);
            act_lane += min(h + kAhead - 1u, last_row) * row_stride;
        } else {
            raw[0] = load_raw();
#pragma unroll
            for (uint32_t j = 1; j < kAhead; ++j) {
                act_lane += last_row >= j ? row_stride : 0u;           // clamped, not guarded: late rows are re-read
                raw[j] = load_raw();
            }
        }
    }
    if constexpr (COMPACT && BITMAP == 3) {
        // the host-built delta rows (RolloutArgs::mv4) as they are: 16 bytes per thread and load, ten loads in flight (the
        // 16-byte rows this form was first staged from are 13 times the bytes: 263 KB per block against 79 KB on a 64x64 map)
        const uint32_t n_vec = uint32_t(delta_table_words(p.c.n_cells) / 4u);
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.mv4);
        constexpr uint32_t kInFlight = 10;                         // (79 KB of a 64x64 map: one round trip for a 512-thread block)
        for (uint32_t w0 = threadIdx.x; w0 < n_vec; w0 += kInFlight * blockDim.x) {
            u32x4 part[kInFlight];
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k) part[k] = src[min(w0 + k * blockDim.x, n_vec - 1u)];
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k)
                if (w0 + k * blockDim.x < n_vec) *(__attribute__((address_space(3))) u32x4 *)lds_addr(lds, kMoveAt + 16u * (w0 + k * blockDim.x)) = part[k];
        }
    } else
    {   // move table -> LDS with six columns per cell (0..4 = the actions, 5 = STAY again: where out-of-range action
        // bytes are clamped to), batches of eight independent loads per thread
        // (COMPACT: five columns, the first 8 bytes of every row)
        // (COMPACT + BITMAP == 1: FOUR columns -- the moves; a STAY row is (cell, cell, cell) with the all-equal code and is made
        // up in registers -- which leaves room for the occupancy bitmaps behind the table)
        constexpr uint32_t kCols = !COMPACT ? kMoveCols : (BITMAP == 1 ? kBitmapCols : (BITMAP == 3 ? kDeltaCols : kCompactCols));
        const uint32_t n_words = p.c.n_cells * kCols;
        constexpr uint32_t kInFlight = 8;                          // (room-32-32-4's 65 KB: one round trip for a 512-thread block)
        for (uint32_t w0 = threadIdx.x; w0 < n_words; w0 += kInFlight * blockDim.x) {
            MoveEntry part[kInFlight];
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k) {
                const uint32_t w = min(w0 + k * blockDim.x, n_words - 1u);
                const uint32_t cell = w / kCols, col = w - cell * kCols;
                part[k] = p.mv[(!COMPACT || BITMAP == 3) ? cell * kMvCols + (col < kMvCols ? col : 0u) : cell * kMvCols + col + (BITMAP == 1 ? 1u : 0u)];
            }
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k) {
                const uint32_t w = w0 + k * blockDim.x;
                if (w < n_words) {
                    // COMPACT rows: {c0 | c1 << 16, c2 | byte offset of the code's slip row << 16}
                    if (COMPACT && BITMAP == 3) {
                        // {c0 - cell, c1 - cell, c2 - cell (low bytes: a slot past the list's end is never sampled), (row offset + bias) / 8}
                        const uint32_t cell = w / kCols;
                        reinterpret_cast<uint32_t *>(lds_mv)[w] = ((part[k].x - cell) & 0xFFu) | ((((part[k].x >> 16) - cell) & 0xFFu) << 8) |
                                                                  (((part[k].y - cell) & 0xFFu) << 16) | (((part[k].w + kRowBias) >> 3) << 24);
                    } else if (COMPACT) reinterpret_cast<u32x2 *>(lds_mv)[w] = u32x2{part[k].x, (part[k].y & 0xFFFFu) | ((part[k].w + kRowBias) << 16)};
                    else lds_mv[w] = make_uint4(part[k].x, part[k].y, part[k].z ^ kHalfBias, part[k].w + kRowBias);   // thresholds: see sample_slot_packed
                }
            }
        }
    }
    uint32_t bitmap_at = 0u;
    if (BITMAP) {
        const uint32_t stride = (((p.c.n_cells + 31u) >> 5) * 4u + 15u) & ~15u;    // bytes per env: one bit per cell
        bitmap_at = bitmap_base + (threadIdx.x / uint32_t(Q)) * stride;
        const uint32_t n_words = (blockDim.x / uint32_t(Q)) * (stride >> 2);
        for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) *(lds_u32)lds_addr(lds, bitmap_base + 4u * w) = 0u;
    }
    stage_outcome_table(p.c, outcome);
    stage_slip_table(p.slip, slip);   // ends with __syncthreads()

    uint32_t terminal = packed_is_terminal<Q, P>(x, c, g) ? 1u : 0u;
    const uint32_t start_terminal = (p.auto_reset && packed_is_terminal<Q, P>(x, start_c, g)) ? 1u : 0u;
    // Every select of the reset logic compares the env's integer code against a wave-uniform constant:
    // (code ^ 4) > 0 <=> the step ended the episode (or the env was terminal already); with auto-reset off the
    // threshold is unreachable, so "reset" never fires and the state simply stays where the step left it.
    const uint32_t reset_above = p.auto_reset ? 0u : 0xFFFFFFFFu;

    const bool leader = x.g == 0u, tail = x.g == uint32_t(Q - 1);
    // the totals' addresses: held in vector registers across the step loop (formed again at the end they keep their argument
    // fields alive in scalar registers, which the eight-agents-per-lane instances do not have), except with four agents per
    // lane, whose 1024-thread forms have 128 vector registers per lane and none to spare: those form them again where they
    // store.  The running counts are added to what the arrays hold at the end.
    constexpr bool HOLD_TOTALS = K != 4;
    auto totals_at = [&](gf64 &ret_p, gu32 &epi_p, gu32 &col_p) __attribute__((always_inline)) {
        ret_p = (gf64)(p.out_returns ? at(p.out_returns, e) : nullptr);
        epi_p = (gu32)(p.out_episodes ? at(p.out_episodes, e) : nullptr);
        col_p = (gu32)(p.out_collisions ? at(p.out_collisions, e) : nullptr);
    };
    gf64 ret_p = nullptr;
    gu32 epi_p = nullptr, col_p = nullptr;
    if (HOLD_TOTALS) {
        totals_at(ret_p, epi_p, col_p);
        asm volatile("" : "+v"(ret_p), "+v"(epi_p), "+v"(col_p));
    }
    double ret = 0.0;
    if (p.accumulate && p.out_returns && leader) ret = *(gf64)at(p.out_returns, e);
    // (formed where it is used -- the slip refresh, one step in four, and the tie path: not held across the loop)
#define env_id (p.env_id_offset + x.e)
    const uint64_t t_first = first_step_index(p);
    const uint32_t n_envs = uint32_t(p.n_envs);

    // per-lane pointers into the step rows; they advance by wave-uniform strides
    const uint64_t step_rows = n_envs, step_cells = uint64_t(n_envs) * n_agents;
    const bool odd = (x.g & 1u) != 0u;
    const uint32_t flag_shift = (x.g & 1u) * 16u;
    gf64 wide_lane = nullptr, prob_lane = nullptr;
    gu8 narrow_lane = nullptr, coll_lane = nullptr;
    gu16 rec_lane = nullptr;
    if (RECORD) {
        gf64 reward_lane = (gf64)p.rec_reward + e;
        prob_lane = (gf64)p.rec_prob + e;
        gu8 done_lane = (gu8)p.rec_done + e;
        coll_lane = (gu8)p.rec_collision + e;
        // Q >= 2: the last lane writes prob, the others reward; even lanes write done, odd lanes collision
        wide_lane = (Q > 1 && tail) ? prob_lane : reward_lane;
        narrow_lane = (Q > 1 && odd) ? coll_lane : done_lane;
        rec_lane = (gu16)p.rec_local + lane_cell;
    }
    asm volatile("" : "+v"(wide_lane), "+v"(prob_lane), "+v"(narrow_lane), "+v"(coll_lane), "+v"(rec_lane));
#pragma unroll
    for (uint32_t j = 0; j < kAhead; ++j) asm volatile("" : "+v"(raw[j]));   // consumed here: the loop's waits are counted ones
    asm volatile("" : "+v"(raw_first), "+v"(raw_head[0]), "+v"(raw_head[1]));
    Words4 rng[P];
#pragma unroll
    for (int i = 0; i < P; ++i) rng[i] = Words4{0u, 0u, 0u, 0u};
    // "Pending" = what is left of step s-1 when step s begins: its probability chain, its totals and its trajectory
    // stores.  They are finished at the top of step s, right after step s's table reads have been issued, so the
    // chain of dependent float64 multiplies runs while those reads are in flight, and the outcome row / probability
    // reads of step s-1 (requested in step s-1, consumed only here) never stall anything.  (The launch's first step has
    // nothing pending and skips this.)
    double pq[K], p_reward = -0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) pq[i] = 0.0;
    // SYS: the ordered probability product as a systolic chain.  With Q lanes per env the product of a step is Q - 1
    // hand-overs of K multiplies each, and the lock-step form (packed_prob_product) has EVERY lane execute all of them --
    // 45 of the ~200 vector instructions of a 32-agent lane-step.  Here every lane does ONE round per step: lane g continues,
    // with its factors of step u - g, the product lane g - 1 handed over a step ago; the last lane completes step u - (Q-1)
    // and stores it Q - 1 rows behind the other trajectory arrays (a launch ends with Q - 1 draining rounds).  A lane keeps
    // its factors of the last Q steps as packed LDS indices (four 7-bit fields per word: the probabilities live in the slip
    // rows) and reads the delayed ones when their turn comes.  Same multiplications in the same order: bit-identical.
    // (Q = 16: a ring of 16 words does not fit the 128 registers of a 1024-thread block; nor do 8 beside the greedy policy's goal
    // coordinates and the SoC bookkeeping -- tests/test_cabi_and_host.py keeps every instance free of spills)
    constexpr bool SYS = RECORD && K == 4 && Q == 8 && !(SOC && !STREAM);
    constexpr int kRing = SYS ? Q : 1;
    uint32_t qring[kRing], p_qword = 0u;   // qring[j]: my factors' indices of the step j before the pending one
#pragma unroll
    for (int j = 0; j < kRing; ++j) qring[j] = 0u;
    double chain_run = 1.0;                // what I handed on in the last round
    uint32_t eight = 8u;
    asm volatile("" : "+v"(eight));
    uint32_t p_cells[P], p_status = 0u, counts = 0u;   // p_status: done | collision << 16 of the pending step; counts: their sums
#pragma unroll
    for (int i = 0; i < P; ++i) p_cells[i] = 0u;

#ifdef MAPF_STAMPS
    StampCtx st_{};
    StampCtx &st = st_;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st.last = _t; }
#endif
    // SYS, once per step (behind the step's table reads): file the pending step's indices, pick the word whose turn it is in
    // this lane (the one filed g steps ago: a select tree over the bits of g, masks hoisted) and request its four probabilities
    auto chain_fetch = [&](double (&qv)[K]) __attribute__((always_inline)) {
        if constexpr (SYS) {
            asm volatile("" : "+v"(p_qword));
#pragma unroll
            for (int j = kRing - 1; j > 0; --j) qring[j] = qring[j - 1];
            qring[0] = p_qword;
            uint32_t level[kRing];
#pragma unroll
            for (int j = 0; j < kRing; ++j) level[j] = qring[j];
#pragma unroll
            for (int width = kRing, bit = 1; width > 1; width /= 2, bit *= 2) {
                const bool upper = (x.g & uint32_t(bit)) != 0u;
#pragma unroll
                for (int j = 0; j < width / 2; ++j) level[j] = upper ? level[2 * j + 1] : level[2 * j];
            }
            const uint32_t mine = level[0];
            uint32_t at[K];
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(at[0]) : "v"(mine), "v"(eight));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(at[1]) : "v"(mine), "v"(eight));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(at[2]) : "v"(mine), "v"(eight));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(at[3]) : "v"(mine), "v"(eight));
#pragma unroll
            for (int k = 0; k < K; ++k) qv[k] = lds_at<double>(lds, kSlipAt + at[k]);
        }
    };
    // SYS: one round -- continue what the lane before me handed over (the group's first lane starts a product: 1.0 * q is q)
    auto chain_round = [&](const double (&qv)[K]) __attribute__((always_inline)) {
        const uint32_t lo = from_prev_lane<Q>(uint32_t(__double2loint(chain_run)));
        const uint32_t hi = from_prev_lane<Q>(uint32_t(__double2hiint(chain_run)));
        double run = __hiloint2double(int(x.g == 0u ? 0x3FF00000u : hi), int(x.g == 0u ? 0u : lo));
#pragma unroll
        for (int k = 0; k < K; ++k) run = __dmul_rn(run, qv[k]);
        chain_run = run;
        return run;                                            // in the last lane: the product of the step Q - 1 before the pending one
    };
    auto finish_pending = [&](const double (&qv)[K]) __attribute__((always_inline)) {
        // opaque from here on: otherwise the optimiser moves these consumers back to where the values are produced
        // (the end of the previous step), which is exactly the stall this pipeline removes
        asm volatile("" : "+v"(p_reward), "+v"(p_status));
        if (RECORD && !SYS) {
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("" : "+v"(pq[i]));
        }
        ret = __dadd_rn(ret, p_reward);
        counts += p_status;                                    // two 16-bit counts (a launch has at most 65535 steps)
        if (RECORD) {
            double prob;
            if constexpr (SYS) prob = chain_round(qv);
            else prob = packed_prob_product<Q, K>(pq);           // total in the last lane
            Packed<P> out;
#pragma unroll
            for (int i = 0; i < P; ++i) out.v[i] = p_cells[i];
            out.store_global(rec_lane);
            *wide_lane = (Q > 1 && tail) ? prob : p_reward;
            *narrow_lane = uint8_t(Q > 1 ? p_status >> flag_shift : p_status);
            if (Q == 1) {
                *prob_lane = prob;
                *coll_lane = uint8_t(p_status >> 16);
            }
        }
    };

    uint32_t goal_rc[K];   // greedy policy: my agents' goal coordinates
#pragma unroll
    for (int k = 0; k < K; ++k) goal_rc[k] = 0u;
    if (!STREAM && p.policy_cells) {
#pragma unroll
        for (int k = 0; k < K; ++k) goal_rc[k] = p.policy_cells[(k & 1) ? g[k / 2] >> 16 : g[k / 2] & 0xFFFFu].x;
    }

    // In-kernel policy stream (!STREAM, no greedy table): the words of the current four-step block, one call per agent quad
    constexpr int kPolicyCalls = K == 8 ? 2 : 1;
    constexpr uint32_t kColShift = COMPACT ? (BITMAP == 3 ? 2u : 3u) : 4u;     // log2 of a table row's bytes
    // the policy word's bytes go straight into the table address (no action integer is formed) where nothing else asks for
    // the action: not in the SoC instances (_living_reward counts STAY) nor behind the four-column table (STAY has no row there)
    constexpr bool FAST_POLICY = !STREAM && !SOC && !(COMPACT && BITMAP == 1);
    Words4 pol[kPolicyCalls];
#pragma unroll
    for (int j = 0; j < kPolicyCalls; ++j) pol[j] = Words4{0u, 0u, 0u, 0u};
    // (eight agents per lane: the key waits in two VECTOR registers -- those instances have them to spare, while their SoC
    // form is two scalar registers short of keeping it beside the slip stream's)
    uint32_t pol_key_lo = p.c.pol_lo, pol_key_hi = p.c.pol_hi;
    if (K == 8) asm volatile("" : "+v"(pol_key_lo), "+v"(pol_key_hi));
    // PRECOL: the four steps' column offsets are formed when the call is made (behind that step's table reads, off the path
    // from the step's top to its own reads) and kept instead of the words: pol[j] = the even bytes' pairs of the four steps,
    // pol_odd[j] = the odd bytes' -- four registers more, which the 32-agent (bitmap) instances do not have
    constexpr bool PRECOL = FAST_POLICY && !(COMPACT && BITMAP != 0);
    Words4 pol_odd[kPolicyCalls];
#pragma unroll
    for (int j = 0; j < kPolicyCalls; ++j) pol_odd[j] = Words4{0u, 0u, 0u, 0u};
    // the column offsets (action << kColShift) of two agents at a time, never leaving their half-words: byte * (5 << kColShift)
    // has the action in bits 8 + kColShift .. of its half -- i.e. byte 1 of the half IS the column offset once the fraction
    // below it is masked off; the table address adds it with a byte select
    auto column_pairs = [&](uint32_t pw, uint32_t &even_pair, uint32_t &odd_pair) __attribute__((always_inline)) {
        if constexpr (K == 2) pw >>= 16u * (x.g & 1u);             // the quad is shared with the neighbour lane: bytes 2 (g & 1), + 1
        const uint32_t even = __builtin_amdgcn_perm(pw, pw, 0x0C020C00u), odd = __builtin_amdgcn_perm(pw, pw, 0x0C030C01u);   // {b0, 0, b2, 0}, {b1, 0, b3, 0}
        even_pair = __umul24(even, 5u << kColShift) & (0x00070007u << (8 + kColShift));
        odd_pair = __umul24(odd, 5u << kColShift) & (0x00070007u << (8 + kColShift));
    };
    auto refresh_policy = [&](const uint64_t m) __attribute__((always_inline)) {
        if constexpr (K == 8) policy_words_x2(__builtin_amdgcn_readfirstlane(pol_key_lo), __builtin_amdgcn_readfirstlane(pol_key_hi), env_id, m,
                                              2u * x.g, 2u * x.g + 1u, pol[0], pol[1]);
        else pol[0] = policy_words(p.c, env_id, m, K == 4 ? x.g : x.g >> 1);
        if constexpr (PRECOL) {
#pragma unroll
            for (int j = 0; j < kPolicyCalls; ++j) {
                const Words4 w = pol[j];
                column_pairs(w.w0, pol[j].w0, pol_odd[j].w0);
                column_pairs(w.w1, pol[j].w1, pol_odd[j].w1);
                column_pairs(w.w2, pol[j].w2, pol_odd[j].w2);
                column_pairs(w.w3, pol[j].w3, pol_odd[j].w3);
            }
        }
    };

    // One step.  W = which word of the slip calls this step uses (t & 3) when that is a compile-time fact, -1 = generic
    // (word picked at run time, call refreshed when t is a multiple of four).  FIRST = the launch's first step: nothing
    // is pending yet and the slip call is refreshed whatever t is.  TAIL = 1: the action rows may run out within kAhead
    // steps, so the prefetch address is clamped; 2: no prefetch (the first step of a WORD_SLOTS launch, whose register is its
    // own).  `raw` is the register that holds this step's action word.
    // (delta rows: the slot selects a byte -- steps of one, the row's byte 2 down to 0, zeros above it)
    uint32_t pk_eights = 0x00080008u, pk_steps = BITMAP == 3 ? 0x00010001u : 0x02020202u, sel_base = BITMAP == 3 ? 0x0C0C0C02u : 0x0C0C0504u;   // sample_slot_packed's constants,
    asm volatile("" : "+v"(pk_eights), "+v"(pk_steps), "+v"(sel_base));                   // one vector register each
    uint32_t row_bytes = COMPACT ? (BITMAP == 3 ? kDeltaCols * kDeltaEntry : (BITMAP == 1 ? kBitmapCols : kCompactCols) * kCompactEntry) : kMoveCols * uint32_t(sizeof(MoveEntry));
    asm volatile("" : "+v"(row_bytes));   // (one register for the whole loop; as an SGPR operand the assembler rejects the SDWA form)
    auto one_step = [&](const uint32_t s, RawWord &raw, auto w_tag, auto first_tag, auto tail_tag) __attribute__((always_inline)) {
        constexpr int W = decltype(w_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int TAIL = int(decltype(tail_tag)::value);
        constexpr bool MAYBE_TERMINAL = TERM || FIRST;   // (the launch's first step finds whatever state the last launch left)
        const uint64_t t = t_first + s;
        double qv[K];                                              // SYS: the factors of this step's chain round
        uint32_t cur[K], act[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cur[k] = (k & 1) ? c[k / 2] >> 16 : c[k / 2] & 0xFFFFu;
        // --- my agents' table rows: cell * (row bytes) + (action << kColShift), the table's LDS offset is an immediate
        uint32_t cell_at[K], col_at[K];
        auto rows_of_cells = [&]() __attribute__((always_inline)) {   // (all the word-select multiplies first: back to back with their users each one costs an s_nop)
#pragma unroll
            for (int k = 0; k < K; ++k) cell_at[k] = (k & 1) ? half_times<1>(c[k / 2], row_bytes) : half_times<0>(c[k / 2], row_bytes);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(cell_at[k]));
        };
        auto columns_from_actions = [&]() __attribute__((always_inline)) {   // (behind the actions: the 1024-thread forms have no register to hold both for long)
            rows_of_cells();
#pragma unroll
            for (int k = 0; k < K; ++k) col_at[k] = (act[k] << kColShift) + cell_at[k];
        };
        if (STREAM) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t byte = uint32_t(raw >> (8 * k)) & 0xFFu;
                act[k] = (COMPACT && BITMAP != 3) ? (byte > 4u ? 0u : byte) : min(byte, 5u);   // six columns: extract + clamp is one v_min_u32 (byte select)
            }
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(act[k]));   // the wait for `raw` sits here
            if constexpr (TAIL != 2) {
                act_lane += (!TAIL || s + kAhead <= last_row) ? step_cells : 0u;   // row min(s + kAhead, last)
                raw = load_raw();
            }
        } else if (p.policy_cells) {   // greedy policy
#pragma unroll
            for (int k = 0; k < K; ++k) act[k] = greedy_action(p.policy_cells, p.c.n_cells, cur[k], goal_rc[k]);
            if constexpr (FAST_POLICY) columns_from_actions();
        } else {   // policy stream: the step's word of my quad's call (one call per quad per four steps), a byte per agent
            if (FIRST) refresh_policy(t >> 2);                     // (later blocks: requested in the step before their first one, below)
            if constexpr (FAST_POLICY) rows_of_cells();
#pragma unroll
            for (int j = 0; j < kPolicyCalls; ++j) {
                uint32_t pw = W == 0 ? pol[j].w0 : W == 1 ? pol[j].w1 : W == 2 ? pol[j].w2 : W == 3 ? pol[j].w3 : step_word(pol[j], t);
                if constexpr (FAST_POLICY) {
                    uint32_t pair[2];
                    if constexpr (PRECOL) {
                        pair[0] = pw;
                        pair[1] = W == 0 ? pol_odd[j].w0 : W == 1 ? pol_odd[j].w1 : W == 2 ? pol_odd[j].w2 : W == 3 ? pol_odd[j].w3 : step_word(pol_odd[j], t);
                    } else column_pairs(pw, pair[0], pair[1]);
#pragma unroll
                    for (int b = 0; b < (K == 2 ? 2 : 4); ++b) {       // byte b of the word: pair b & 1, low / high half
                        const int k = K == 2 ? b : 4 * j + b;
                        if (K == 2 || b < 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(col_at[k]) : "v"(cell_at[k]), "v"(pair[b & 1]));
                        else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(col_at[k]) : "v"(cell_at[k]), "v"(pair[b & 1]));
                    }
                } else {
                    // (the instances that need the action itself -- SoC's STAY count, the four-column table's made-up STAY row)
                    if constexpr (K == 2) pw >>= 16u * (x.g & 1u);  // the quad is shared with the neighbour lane: bytes 2 (g & 1), + 1
                    act[K == 2 ? 0 : 4 * j] = policy_action_rt(pw, 0u);
                    act[K == 2 ? 1 : 4 * j + 1] = policy_action_rt(pw, 1u);
                    if constexpr (K != 2) {
                        act[4 * j + 2] = policy_action_rt(pw, 2u);
                        act[4 * j + 3] = policy_action_rt(pw, 3u);
                    }
                }
            }
            if constexpr (!FAST_POLICY) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("" : "+v"(act[k]));   // (an integer 0..4 from here on: the shift is not folded into the address)
            }
        }
        if constexpr (!FAST_POLICY) columns_from_actions();       // (FAST_POLICY: each branch above has formed its columns)

        // --- the rows are requested first ...
        MoveEntry entry[K];
        u32x2 cells_code[K];
        uint32_t delta_row[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (COMPACT && BITMAP == 3) delta_row[k] = lds_at<uint32_t>(lds, kMoveAt + col_at[k]);
            else if (COMPACT && BITMAP == 1) cells_code[k] = lds_at<u32x2>(lds, kMoveAt - kCompactEntry + col_at[k]);   // column act - 1 (STAY: see below)
            else if (COMPACT) cells_code[k] = lds_at<u32x2>(lds, kMoveAt + col_at[k]);
            else entry[k] = lds_entry_at(lds, kMoveAt + col_at[k]);
        }
        STAMP(0);   // loop top: action fetch / policy, table read issue
        // --- ... then the previous step is finished while they are in flight
        if (!FIRST) {
            if (SYS) chain_fetch(qv);                              // behind the table reads: its factors arrive with the rows
            finish_pending(qv);
            if (RECORD) {
                rec_lane += step_cells;
                // SYS: the last lane's probability rows trail by Q - 1 steps -- its pointer rests on row 0 (which the early,
                // incomplete products overwrite until the right one arrives) while s < Q; the unrolled loop only runs beyond that
                if (SYS && TAIL != 0) wide_lane += (tail && s < uint32_t(Q)) ? 0u : step_rows;
                else wide_lane += step_rows;
                narrow_lane += step_rows;
                if (Q == 1) { prob_lane += step_rows; coll_lane += step_rows; }
            }
        }
        STAMP(1);   // previous step: probability chain, totals, trajectory stores
        if (COMPACT) {   // the code's thresholds: a second LDS read that depends on the first; the row completes to a MoveEntry
            uint32_t row_off[K], th[K];
            if (BITMAP == 1) {   // a STAY row: the cell itself, the all-equal code (one entry: candidates m = r = l)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool stay = act[k] == 0u;
                    cells_code[k].x = stay ? cur[k] : cells_code[k].x;
                    cells_code[k].y = stay ? (7u * uint32_t(sizeof(SlipRow)) + kRowBias) << 16 : cells_code[k].y;
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (BITMAP == 3) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(row_off[k]) : "v"(delta_row[k]), "v"(eight));
                else row_off[k] = cells_code[k].y >> 16;
                th[k] = lds_at<uint32_t>(lds, kSlipAt + uint32_t(offsetof(SlipRow, th_biased)) - kRowBias + row_off[k]);   // (th[0] | th[1] << 16) ^ bias
            }
#pragma unroll
            for (int k = 0; k < K; ++k) entry[k] = BITMAP == 3 ? make_uint4(delta_row[k], 0u, th[k], row_off[k]) : make_uint4(cells_code[k].x, cells_code[k].y, th[k], row_off[k]);
        }
        // A slip-stream call serves an agent quad for two steps: every four steps a lane refreshes the two calls of the
        // block (h0, h0 + 1) for each of its quads, in lockstep, and files their words per pair in step order (rng[i] =
        // pair i's words of steps 4m .. 4m+3: register renaming, no instructions).  Two agents per lane: the quad is shared
        // with the neighbour lane -- one call each, halves traded (pair_block_words).
        const bool refresh = FIRST || W == 0 || (W < 0 && (t & 3u) == 0u);
        if (refresh && p.c.need_rng) {
            const uint64_t h0 = block_first_call(t);
            if constexpr (P == 1) {
                rng[0] = pair_block_words<true>(p.c, env_id, t, x.g);
            } else {
#pragma unroll
                for (int j = 0; j < P / 2; ++j) {
                    const uint32_t quad = uint32_t(P / 2) * x.g + uint32_t(j);
                    Words4 a, b;
                    slip_words_x2(p.c, env_id, h0, quad, h0 | 1u, quad, a, b);
                    rng[2 * j] = block_words(a, b, 0u);
                    rng[2 * j + 1] = block_words(a, b, 1u);
                }
            }
        }
        // ... and the policy stream's call of the NEXT block is made in the block's last step (its words are free by then:
        // this step's actions were taken from them at the top), so a block's first step finds its actions ready
        if (!STREAM && !p.policy_cells && (W == 3 || (W < 0 && (t & 3u) == 3u))) refresh_policy((t >> 2) + 1u);
        STAMP(2);   // slip Philox (1 step in 4)
        double q[K];
        uint32_t n[P], word[P], d[K], q_at[K], tie_all = 0u;   // q_at: byte offset of the sampled slot's probability from kSlipAt
#pragma unroll
        for (int i = 0; i < P; ++i) {
            word[i] = W == 0 ? rng[i].w0 : W == 1 ? rng[i].w1 : W == 2 ? rng[i].w2 : W == 3 ? rng[i].w3 : step_word(rng[i], t);
            const uint32_t biased = word[i] ^ kHalfBias;             // low half: agent 2i's uniform, high half: agent 2i+1's
            uint32_t cell[2];
            if constexpr (BITMAP == 3) {
                d[2 * i] = sample_slot_delta<0>(entry[2 * i].x, entry[2 * i].z, entry[2 * i].w, __builtin_amdgcn_perm(biased, biased, 0x01000100u),
                                                pk_eights, pk_steps, sel_base, c[i], q_at[2 * i], cell[0]);
                d[2 * i + 1] = sample_slot_delta<1>(entry[2 * i + 1].x, entry[2 * i + 1].z, entry[2 * i + 1].w,
                                                    __builtin_amdgcn_perm(biased, biased, 0x03020302u), pk_eights, pk_steps, sel_base, c[i],
                                                    q_at[2 * i + 1], cell[1]);
            } else {
                d[2 * i] = sample_slot_packed(entry[2 * i], __builtin_amdgcn_perm(biased, biased, 0x01000100u), pk_eights, pk_steps,
                                              sel_base, q_at[2 * i], cell[0]);
                d[2 * i + 1] = sample_slot_packed(entry[2 * i + 1], __builtin_amdgcn_perm(biased, biased, 0x03020302u), pk_eights, pk_steps,
                                                  sel_base, q_at[2 * i + 1], cell[1]);
            }
            if (!SYS) {   // (SYS reads the probabilities when their chain round comes)
                q[2 * i] = lds_at<double>(lds, kSlipAt + 16u - kRowBias + q_at[2 * i]);
                q[2 * i + 1] = lds_at<double>(lds, kSlipAt + 16u - kRowBias + q_at[2 * i + 1]);
            }
            n[i] = cell[0] | (cell[1] << 16);
            tie_all = i == 0 ? pk_min_u16(d[0], d[1]) : pk_min_u16(tie_all, pk_min_u16(d[2 * i], d[2 * i + 1]));
        }
        // (without slip the words stay zero and every threshold is 65535: no tie can fire, so need_rng is not asked here)
        if (__builtin_expect(__any(zero_half(tie_all) != 0u), 0)) {
            // a top-16-bit tie somewhere in the wave: the agents that tie (in any lane: the test is wave-uniform) are redone
            // with all 53 bits -- for the lanes that did not tie the exact path repeats what the fast path found
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (__any(zero_half(d[k]) != 0u)) {
                    MoveEntry full = entry[k];
                    if (COMPACT && BITMAP == 3) {   // the candidates' cells back from their deltas
                        const uint32_t mine = (k & 1) ? c[k / 2] >> 16 : c[k / 2] & 0xFFFFu, row = full.x;
                        const uint32_t c0 = (mine + uint32_t(int32_t(int8_t(row)))) & 0xFFFFu, c1 = (mine + uint32_t(int32_t(int8_t(row >> 8)))) & 0xFFFFu,
                                       c2 = (mine + uint32_t(int32_t(int8_t(row >> 16)))) & 0xFFFFu;
                        full.x = c0 | (c1 << 16);
                        full.y = c2;
                    }
                    if (COMPACT) full.y = (full.y & 0xFFFFu) | (((full.w - kRowBias) / uint32_t(sizeof(SlipRow))) << 16);   // the code, where entry_code() looks
                    const uint32_t hi = (k & 1) ? word[k / 2] >> 16 : word[k / 2] & 0xFFFFu;
                    uint32_t nx;
                    const uint64_t mant = refine_mantissa(p.c, env_id, t, uint32_t(K) * x.g + uint32_t(k), hi);
                    if (SYS) {
                        const uint32_t slot = slip_slot_exact(slip, full, mant);
                        nx = entry_cell(full, slot);
                        q_at[k] = entry_row_offset(entry[k]) - 16u + 8u * slot;   // what sample_slot_packed makes of that slot
                    } else {
                        slip_move<false>(slip, full, mant, 0.0, nx, q[k]);
                    }
                    n[k / 2] = (k & 1) ? (n[k / 2] & 0xFFFFu) | (nx << 16) : (n[k / 2] & 0xFFFF0000u) | nx;
                }
            }
        }
        STAMP(3);   // sampling (table wait, thresholds, probability read issue)

        // --- pair tests, then the per-env facts as ONE integer: f = vertex | swap << 1 | off_goal << 2
        PairAcc<true> acc;
        if constexpr (BITMAP != 0) acc = bitmap_pair_tests<Q, K>(x, lds, bitmap_at, c, n);
        else acc = packed_pair_tests<Q, P, false, true>(x, c, n);
        STAMP(4);   // pair tests
        uint32_t away = n[0] ^ g[0];
#pragma unroll
        for (int i = 1; i < P; ++i) away |= n[i] ^ g[i];
        asm volatile("" : "+v"(away));   // stays an integer: as a compare it would travel through scalar masks
        // code16 = code * 16 (the byte offset of the code's outcome row), code = vertex | swap << 1 | off_goal << 2 | was_terminal << 3
        uint32_t code16;
        if constexpr (!MAYBE_TERMINAL) {
            // every finished episode is reset, so a vertex collision and a swap need not be told apart (same reward,
            // same status, and is_terminal of the outcome is never asked): one zero test over both minima -> bit 0.
            // Both facts are clamped by a v_min (written out: the optimiser turns min(x, 1) into compare + select).
            const uint32_t hit = zero_half(pk_min_u16(acc.vertex, acc.swap));   // 0, or bits 15 / 31
            uint32_t off_goal, coll16;
            asm("v_min_u32 %0, 1, %1" : "=v"(off_goal) : "v"(away));
            asm("v_min_u32 %0, 16, %1" : "=v"(coll16) : "v"(hit));
            code16 = group_reduce<Q, false>((off_goal << 6) | coll16, x);
        } else {
            // zero_half() leaves bits 15 / 31: vertex -> bits 0 / 16, swap -> bits 1 / 17; both halves folded onto bits 0, 1
            uint32_t bits = (zero_half(acc.vertex) >> 15) | (zero_half(acc.swap) >> 14);
            bits |= bits >> 16;
            const uint32_t flags = group_reduce<Q, false>((min(away, 1u) << 2) | bits, x);
            code16 = ((flags & 7u) | (terminal << 3)) << 4;   // terminal is 0 / 1
        }
        STAMP(5);   // flags + group reduce

        // --- outcome: the row (status for both criteria, reward for Makespan) is only REQUESTED here; everything the
        // next step's table address depends on is derived from `code` without waiting for it
        static_assert(sizeof(OutcomeRow) == 16, "code16 addresses the outcome rows");
        const u32x4 row = lds_at<u32x4>(lds, kOutcomeAt + code16);   // {reward lo, hi, status, pad}
        const uint32_t row_status = row.w;                     // done | collision << 16
        double reward = __hiloint2double(int(row.y), int(row.x));
        const bool was_terminal = MAYBE_TERMINAL && code16 > 7u * 16u;
        if (SOC) {
            // _living_reward: mapf_env.py:436-446
            uint32_t mine = 0u;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t goal_k = (k & 1) ? g[k / 2] >> 16 : g[k / 2] & 0xFFFFu;
                mine += (cur[k] == goal_k && (act[k] == 0u || ((!COMPACT || BITMAP == 3) && act[k] == 5u))) ? 1u : 0u;
            }
            const int stayed = int(group_reduce<Q, true>(mine, x));
            const double living = __dmul_rn(double(int(n_agents) - stayed), p.c.r_living);
            const uint32_t f = (code16 >> 4) & 7u;
            const bool coll = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
            const double r = coll ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
            reward = was_terminal ? 0.0 : r;
        }
        // a step from a terminal state changes nothing: mapf_env.py:239-240 -- (s, 0, True, {"prob": 0})
#pragma unroll
        for (int i = 0; i < P; ++i) n[i] = was_terminal ? c[i] : n[i];
        p_reward = reward;
        p_status = row_status;
#pragma unroll
        for (int i = 0; i < P; ++i) p_cells[i] = n[i];
        if (RECORD && SYS) {                                   // the factors' indices: (offset from kSlipAt) / 8, seven bits each
            static_assert(!SYS || K == 4, "four fields per word");
            uint32_t w = (q_at[0] >> 3) | (q_at[1] << 5) | (q_at[2 % K] << 13) | (q_at[3 % K] << 21);
            if (MAYBE_TERMINAL) w = was_terminal ? (w & ~0x7Fu) | kZeroFactor : w;
            p_qword = w;
        } else if (RECORD) {                                   // a zero factor makes the whole product +0.0
            pq[0] = was_terminal ? 0.0 : q[0];
#pragma unroll
            for (int k = 1; k < K; ++k) pq[k] = q[k];
        }
        STAMP(6);   // outcome request, SoC living reward
        // MapfEnv.reset(): start cells, no reseed.  Every code except "off goal, no collision, not terminal" (= 4) ends
        // the episode; the returned state is terminal after a vertex collision or on goal (mapf_env.py:210-223), a swap
        // alone is not: bits 0 (vertex), 2 (flipped: on goal) and 3 (was terminal) of code ^ 4
        const uint32_t ended = code16 ^ (4u * 16u);
        // (the instance without terminal handling only runs with auto-reset on: one compare against the code itself)
        const bool back = MAYBE_TERMINAL ? ended > reset_above : code16 != 4u * 16u;   // never with auto-reset off
#pragma unroll
        for (int i = 0; i < P; ++i) c[i] = back ? start_c[i] : n[i];
        if (MAYBE_TERMINAL) terminal = back ? start_terminal : min(ended & (13u * 16u), 1u);
        STAMP(7);   // reset handling
    };
    using Generic = std::integral_constant<int, -1>;
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;
    using Yes = std::true_type;
    using No = std::false_type;
    using Skip = std::integral_constant<int, 2>;
    auto shift_raw = [&]() __attribute__((always_inline)) {   // after a single step: raw[0] was reloaded with row s + kAhead
        const RawWord newest = raw[0];
#pragma unroll
        for (uint32_t j = 0; j + 1 < kAhead; ++j) raw[j] = raw[j + 1];
        raw[kAhead - 1] = newest;
    };
    uint32_t s = 0;
    if constexpr (WORD_SLOTS) {
        // The step loop.  Every step outside the first one is one of four bodies (one per slip word) with the register of its
        // word and a clamped prefetch address, so a launch that starts at a word boundary -- t a multiple of four: every launch
        // of a caller whose launches are multiples of four steps -- is straight-line code: the first four steps, then four
        // steps per iteration while they last, then up to three more.  No step moves a register whose load is in flight
        // and no join sits between two steps but the loop's own, so the waits for the action words are the counted ones
        // (vmcnt(15): four steps of one load and three stores each).
        // (Before, the steps before the first boundary and the last four to seven ran as single steps that shifted the
        // registers down -- s_waitcnt vmcnt(0) / vmcnt(3) on the load just issued, +370 cycles a step: 8 of a T = 32 launch's
        // steps, profiles/r05_rollout_short_launch_stamps.txt.)
        // A launch that starts elsewhere takes up to three steps with the word picked at run time to the boundary first.
        const uint32_t n = p.n_steps;
        if (n > 0) {
            const uint32_t h = head_steps(uint32_t(t_first), n);
            one_step(0u, raw_first, Generic{}, Yes{}, Skip{});
            if ((uint32_t(t_first) & 3u) == 0u && n >= kAhead) {     // started at a boundary: h = 1
                one_step(1u, raw[1], W1{}, No{}, Yes{});
                one_step(2u, raw[2], W2{}, No{}, Yes{});
                one_step(3u, raw[3], W3{}, No{}, Yes{});
                s = kAhead;
            } else {
                if (h > 1u) one_step(1u, raw_head[0], Generic{}, No{}, Skip{});
                if (h > 2u) one_step(2u, raw_head[1], Generic{}, No{}, Skip{});
                s = h;
            }
        }
        for (; s + kAhead <= n; s += kAhead) {
            one_step(s, raw[0], W0{}, No{}, Yes{});
            one_step(s + 1u, raw[1], W1{}, No{}, Yes{});
            one_step(s + 2u, raw[2], W2{}, No{}, Yes{});
            one_step(s + 3u, raw[3], W3{}, No{}, Yes{});
        }
        if (s < n) {                                               // (s is at a word boundary here)
            one_step(s, raw[0], W0{}, No{}, Yes{});
            if (s + 1u < n) {
                one_step(s + 1u, raw[1], W1{}, No{}, Yes{});
                if (s + 2u < n) one_step(s + 2u, raw[2], W2{}, No{}, Yes{});
            }
        }
    } else {
        // a single step outside the unrolled loop: its slip word is still picked statically (one four-way branch instead of
        // the word selects inside the step), its prefetch address is clamped
        auto single_step = [&](const uint32_t s) __attribute__((always_inline)) {
            switch (uint32_t(t_first + s) & 3u) {
                case 0: one_step(s, raw[0], W0{}, No{}, Yes{}); break;
                case 1: one_step(s, raw[0], W1{}, No{}, Yes{}); break;
                case 2: one_step(s, raw[0], W2{}, No{}, Yes{}); break;
                default: one_step(s, raw[0], W3{}, No{}, Yes{}); break;
            }
            shift_raw();
        };
        // the first step; single steps up to the slip stream's call boundary; kAhead steps per iteration with static word
        // and register selection and unclamped prefetch while the action rows last; single steps for the rest
        if (p.n_steps > 0) {
            one_step(0u, raw[0], Generic{}, Yes{}, Yes{});
            shift_raw();
            s = 1;
        }
        for (; s < p.n_steps && (((t_first + s) & 3u) != 0u || (SYS && s < uint32_t(Q))); ++s) single_step(s);
        // (streamed actions: the group's last step prefetches row s + 2 kAhead - 1, so the last rows are single steps; an in-kernel
        // policy prefetches nothing and stays in the loop while whole groups are left -- a single step sits in a basic block of its
        // own and cannot start its table reads under the step before it: +500 cycles, profiles/r05_rollout_short_launch_stamps.txt)
        for (; s + (STREAM ? 2u : 1u) * kAhead <= p.n_steps; s += kAhead) {
            one_step(s, raw[0], W0{}, No{}, No{});
            one_step(s + 1u, raw[1], W1{}, No{}, No{});
            one_step(s + 2u, raw[2], W2{}, No{}, No{});
            one_step(s + 3u, raw[3], W3{}, No{}, No{});
            if constexpr (kAhead == 8) {
                one_step(s + 4u, raw[4], W0{}, No{}, No{});
                one_step(s + 5u, raw[5], W1{}, No{}, No{});
                one_step(s + 6u, raw[6], W2{}, No{}, No{});
                one_step(s + 7u, raw[7], W3{}, No{}, No{});
            }
        }
        for (; s < p.n_steps; ++s) single_step(s);
    }
    if (p.n_steps > 0) {                                       // the last step's chain, totals and stores
        double qv[K];
        if (SYS) chain_fetch(qv);
        finish_pending(qv);
        if constexpr (SYS) {
            // ... and the Q - 1 rounds that complete the products still on their way through the group (the other lanes
            // re-store the last reward in place)
            for (uint32_t u = p.n_steps; u < p.n_steps + uint32_t(Q - 1); ++u) {
                wide_lane += (tail && u >= uint32_t(Q)) ? step_rows : 0u;
                p_qword = 0u;
                chain_fetch(qv);
                const double prob = chain_round(qv);
                *wide_lane = tail ? prob : p_reward;
            }
        }
    }
#ifdef MAPF_STAMPS
    if (x.lane == 0u && p.out_episodes) {   // diagnostic build: segment sums replace the episode counts
        for (int k = 0; k < 8; ++k) at(p.out_episodes, e)[k] = uint32_t(st.seg[k]);
        return;
    }
#endif
    {
        Packed<P> fin;
#pragma unroll
        for (int i = 0; i < P; ++i) fin.v[i] = c[i];
        // (the address is formed again from the env index -- laundered, so that it is not the kernel's first address kept in
        // two registers across the whole step loop: the 1024-thread instances have none to spare and would spill it)
        uint32_t e_end = x.e;
        asm volatile("" : "+v"(e_end));
        fin.store(at(p.state, e_end * n_agents + uint32_t(K) * x.g));
    }
    if (leader) {
        if (!HOLD_TOTALS) totals_at(ret_p, epi_p, col_p);
        if (ret_p) *ret_p = ret;
        if (epi_p) *epi_p = (p.accumulate ? *epi_p : 0u) + (counts & 0xFFFFu);
        if (col_p) *col_p = (p.accumulate ? *col_p : 0u) + (counts >> 16);
    }
}

#undef env_id

// bytes of one env's occupancy bitmap (BITMAP instances)
static size_t bitmap_stride(uint32_t n_cells) { return (size_t((n_cells + 31u) / 32u) * 4u + 15u) & ~size_t(15); }   // one bit per cell

template <int Q, int K, bool RECORD, bool STREAM, bool COMPACT = false, int BITMAP = 0>
hipError_t launch_impl(const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream) {
    // (criteria, may-be-terminal): the instance without terminal handling exists for Makespan only
    const bool term = !(args.auto_reset && !args.start_terminal_any);
    auto kern = args.c.criteria != 0u ? lq_rollout_kernel<Q, K, RECORD, STREAM, true, COMPACT, true, BITMAP>
                : term            ? lq_rollout_kernel<Q, K, RECORD, STREAM, false, COMPACT, true, BITMAP>
                                  : lq_rollout_kernel<Q, K, RECORD, STREAM, false, COMPACT, false, BITMAP>;
    const uint32_t bitmap_base = uint32_t(lds_bytes);           // the bitmaps follow the table
    if (BITMAP) lds_bytes += size_t(block / unsigned(Q)) * bitmap_stride(args.c.n_cells);
    if (lds_bytes > 32 * 1024) {
        // (this kernel has no static LDS object: its dynamic segment may be the CU's whole 160 KB -- the limit every form's
        // "does it fit" test in try_launch_rollout_lq compares against)
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(kLdsBytes))) return e;
    }
    const unsigned grid = unsigned(args.n_envs / (block / unsigned(Q)));
    note_kernel("lq_rollout_kernel<Q=%d,K=%d,%s,%s,%s%s%s%s> block=%u (packed layout: %d agents per lane%s%s)", Q, K, RECORD ? "RECORD" : "TOTALS",
                STREAM ? "STREAM" : "POLICY", args.c.criteria != 0u ? "SOC" : "MAKESPAN", COMPACT ? ",COMPACT" : "",
                (args.c.criteria == 0u && !term) ? ",NO_TERMINAL" : "", (BITMAP == 2 && COMPACT) ? ",BITMAP5" : (BITMAP == 3 ? ",BITMAPD" : (BITMAP ? ",BITMAP" : "")), block, K,
                COMPACT ? (BITMAP == 3 ? ", 4-byte delta rows" : (BITMAP == 1 ? ", 8-byte table rows without the STAY column" : ", 8-byte table rows")) : "",
                BITMAP ? ", collisions through per-env occupancy bitmaps" : "");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds_bytes, stream, args, A, bitmap_base);
    return hipGetLastError();
}

}  // namespace

// This file is compiled once per (agents per lane, recording) pair -- -DMAPF_LQ_K=8|4|2 -DMAPF_LQ_RECORD=1|0 -- so that
// its kernel instances build in parallel; each object exports one launcher, the K=4 / RECORD=1 object also the router.
#if !defined(MAPF_LQ_K) || !defined(MAPF_LQ_RECORD)
#error "compile with -DMAPF_LQ_K=8|4|2 -DMAPF_LQ_RECORD=1|0"
#endif
#define MAPF_LQ_CAT3(a, b, c) a##b##_r##c
#define MAPF_LQ_NAME(k, r) MAPF_LQ_CAT3(launch_rollout_lq_k, k, r)

hipError_t MAPF_LQ_NAME(MAPF_LQ_K, MAPF_LQ_RECORD)(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream) {
    constexpr int K = MAPF_LQ_K;
    constexpr bool R = MAPF_LQ_RECORD != 0;
    const bool stream_actions = args.actions != nullptr;
    // form: 0 full table rows, 1 8-byte rows, 2 / 3 8-byte rows + occupancy bitmaps (four / five columns), 4 full rows + bitmaps,
    // 5 4-byte delta rows + bitmaps
    const bool compact = (form >= 1 && form <= 3) || form == 5, bitmap = form >= 2;
    (void)bitmap;
#if MAPF_LQ_K == 8
    // eight agents per lane: 8, 16 and 32 agents (Q = 1, 2, 4); 8-byte table rows for the 32-agent maps only
    if (compact) {
        if (Q != 4 || bitmap) return hipErrorInvalidValue;
        return stream_actions ? launch_impl<4, K, R, true, true>(args, A, block, lds_bytes, stream)
                              : launch_impl<4, K, R, false, true>(args, A, block, lds_bytes, stream);
    }
    switch (Q) {
#define X(QQ)                                                                                                        \
    case QQ: return stream_actions ? launch_impl<QQ, K, R, true>(args, A, block, lds_bytes, stream)                        \
                                   : launch_impl<QQ, K, R, false>(args, A, block, lds_bytes, stream);
        X(1) X(2) X(4)
#undef X
        default: return hipErrorInvalidValue;
    }
}
#else
#if MAPF_LQ_K == 4
    if (bitmap) {    // 32 agents only (that is where the 496 pairs dominate)
        if (Q != 8) return hipErrorInvalidValue;
        if (form == 5) return stream_actions ? launch_impl<8, K, R, true, true, 3>(args, A, block, lds_bytes, stream)
                                             : launch_impl<8, K, R, false, true, 3>(args, A, block, lds_bytes, stream);
        if (form == 4) return stream_actions ? launch_impl<8, K, R, true, false, 2>(args, A, block, lds_bytes, stream)
                                             : launch_impl<8, K, R, false, false, 2>(args, A, block, lds_bytes, stream);
        if (form == 3) return stream_actions ? launch_impl<8, K, R, true, true, 2>(args, A, block, lds_bytes, stream)
                                             : launch_impl<8, K, R, false, true, 2>(args, A, block, lds_bytes, stream);
        return stream_actions ? launch_impl<8, K, R, true, true, 1>(args, A, block, lds_bytes, stream)
                              : launch_impl<8, K, R, false, true, 1>(args, A, block, lds_bytes, stream);
    }
    if (compact) {   // instantiated for the group sizes whose maps need it: 16, 32 and 64 agents
        switch (Q) {
#define X(QQ)                                                                                                        \
    case QQ: return stream_actions ? launch_impl<QQ, K, R, true, true>(args, A, block, lds_bytes, stream)                  \
                                   : launch_impl<QQ, K, R, false, true>(args, A, block, lds_bytes, stream);
            X(4) X(8) X(16)
#undef X
            default: return hipErrorInvalidValue;
        }
    }
#else
    if (compact) return hipErrorInvalidValue;
#endif
    switch (Q) {
#define X(QQ)                                                                                                        \
    case QQ: return stream_actions ? launch_impl<QQ, K, R, true>(args, A, block, lds_bytes, stream)                        \
                                   : launch_impl<QQ, K, R, false>(args, A, block, lds_bytes, stream);
#if MAPF_LQ_K == 4
        X(1)
#endif
        X(2) X(4) X(8) X(16)
#undef X
        default: return hipErrorInvalidValue;
    }
}
#endif

#if MAPF_LQ_K == 4 && MAPF_LQ_RECORD == 1
hipError_t launch_rollout_lq_k8_r1(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream);
hipError_t launch_rollout_lq_k8_r0(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream);
hipError_t launch_rollout_lq_k4_r0(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream);
hipError_t launch_rollout_lq_k2_r1(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream);
hipError_t launch_rollout_lq_k2_r0(int Q, int form, const RolloutArgs &args, uint32_t A, unsigned block, size_t lds_bytes, hipStream_t stream);

// does the K-agents-per-lane form apply to this launch?  (full groups, power-of-two group size, full blocks)
static bool layout_fits(int n_agents, int K, const RolloutArgs &args, size_t lds_bytes, unsigned *block_out, int *q_out) {
    if (n_agents < K || n_agents % K != 0) return false;
    const int Q = n_agents / K;
    if (Q > 16 || (Q & (Q - 1)) != 0 || (K == 2 && Q < 2) || (K == 8 && Q > 4)) return false;
    const size_t copies = kLdsBytes / lds_bytes;   // blocks per CU by LDS
    unsigned block = copies >= 4 ? 256u : 512u;
    // a small batch is spread over the CUs in smaller blocks (down to one wave): every block stages its own table copy,
    // which is cheap next to a rollout's steps, and an idle CU is not
    const uint64_t lanes = args.n_envs * uint64_t(Q);
    while (block > 64u && lanes < 256u * uint64_t(block)) block /= 2u;
    const uint64_t per_block = block / unsigned(Q);
    if (args.n_envs % per_block != 0 || lanes < 64 * 16) return false;
    *block_out = block;
    *q_out = Q;
    return true;
}

// the dispatch decision (see LqPlan): which packed form, block size and LDS image a launch of this shape takes
bool plan_rollout_lq(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, const int n_cu, LqPlan *plan) {
    // top_tie: a three-entry list whose last cumulative sum rounds below 1.0 needs a third compare per agent (hi = 65535);
    // the packed sampling does two, so such a table (none arises from fail_prob / 2 splits) stays with the lane-group kernel
    if (!tune.quad_lanes || args.c.top_tie || args.n_steps > 65535u) return false;   // (per-launch counts are 16-bit)
    unsigned block = 0;
    int Q = 0, K = 0;
    bool compact = false, bitmap = false, stay_column = false, full_rows_bitmap = false, delta_rows = false;
    size_t lds_bytes = kMoveAt + size_t(args.c.n_cells) * kMoveCols * sizeof(MoveEntry);   // the kernel's whole LDS image
    if (lds_bytes <= tune.mv_lds_max_bytes && lds_bytes <= kLdsBytes - kLdsReserve) {
        // Four agents per lane halve the waves: that form needs tune.quad_min_lanes lanes (default: enough to put one
        // wave on every SIMD); below that the two-agents-per-lane form of the same kernel runs.
        // Eight agents per lane halve them again (at 8 agents nothing crosses lanes any more): worth it from two waves
        // per SIMD of THAT form on, i.e. 131072 envs at 8 agents.
        // 32 agents: four per lane with the occupancy bitmaps behind the full table (O(A) collision tests, see below) wherever that
        // form applies -- 496 agent pairs per env are most of either all-pairs form's step
        if (tune.bitmap_pairs && n_agents == 32 && (tune.force_k == 0 || tune.force_k == 4) && layout_fits(n_agents, 4, args, lds_bytes, &block, &Q) &&
            lds_bytes + (block / 8u) * bitmap_stride(args.c.n_cells) <= kLdsBytes &&
            (tune.force_k == 4 || args.n_envs * uint64_t(Q) >= tune.quad_min_lanes)) {
            K = 4;
            bitmap = true;
            full_rows_bitmap = true;
        } else
        if ((tune.force_k == 0 || tune.force_k == 8) && layout_fits(n_agents, 8, args, lds_bytes, &block, &Q) &&
            (tune.force_k == 8 || args.n_envs * uint64_t(Q) >= tune.oct_min_lanes) && block <= 512u) K = 8;
        else if (tune.force_k != 2 && tune.force_k != 8 && layout_fits(n_agents, 4, args, lds_bytes, &block, &Q) &&
                 (tune.force_k == 4 || args.n_envs * uint64_t(Q) >= tune.quad_min_lanes)) K = 4;
        else if (tune.force_k != 4 && tune.force_k != 8 && layout_fits(n_agents, 2, args, lds_bytes, &block, &Q)) K = 2;
        else return false;
    } else {
        // the full table is too large: 8-byte rows, one block per CU (512 threads = two waves per SIMD; 1024 when the
        // batch gives every CU a block of that size), four agents per lane, group sizes 4 / 8 / 16 only
        lds_bytes = kMoveAt + size_t(args.c.n_cells) * kCompactCols * kCompactEntry;
        if (tune.mv_lds_max_bytes == 0 || lds_bytes > kLdsBytes - kLdsReserve) return false;
        const size_t bitmap_lds = kMoveAt + size_t(args.c.n_cells) * kBitmapCols * kCompactEntry;   // (no STAY column in that form)
        // 32 agents: four per lane, collisions through per-env occupancy bitmaps behind the table (one bit per cell) -- O(A)
        // instead of 496 pair tests per env.  64 envs per 512-thread block; 128 per 1024-thread block (four waves per SIMD)
        // once the batch gives every CU a block of that size and 128 bitmaps fit (C5's share of one GPU: 481 G against 377 G
        // for the all-pairs form; C5 whole: profiles/r04_c5_one_bit_bitmap_ab.txt).  MAPF_TUNE k=8 / bitmap_pairs=0 keep the
        // all-pairs forms reachable (eight agents per lane, Q = 4, one 512-thread block per CU; four per lane below).
        const size_t per_env = bitmap_stride(args.c.n_cells);
        unsigned bitmap_block = 512u;
        if (tune.bitmap_block == 1024u || (tune.bitmap_block == 0u && args.n_envs * 8u >= uint64_t(n_cu) * 1024u)) bitmap_block = 1024u;
        if (bitmap_block == 1024u && (args.n_envs % (1024u / 8u) != 0 || bitmap_lds + (1024u / 8u) * per_env > kLdsBytes)) bitmap_block = 512u;
        if (args.actions == nullptr) bitmap_block = 512u;           // (in-kernel policy behind 8-byte rows: that instance is built for 512 threads)
        // ... behind 4-byte delta rows where the map's ids allow them (six columns in 79 KB on the 64x64 maps: 128 bitmaps fit, no
        // STAY row to make up, one-instruction action clamp)
        const size_t delta_lds = kMoveAt + delta_table_words(args.c.n_cells) * kDeltaEntry;   // (the host-built image, zero-padded to 16 bytes)
        unsigned delta_block = (tune.bitmap_block == 1024u || (tune.bitmap_block == 0u && args.n_envs * 8u >= uint64_t(n_cu) * 1024u)) ? 1024u : 512u;
        if (delta_block == 1024u && (args.n_envs % (1024u / 8u) != 0 || delta_lds + (1024u / 8u) * per_env > kLdsBytes)) delta_block = 512u;
        if (tune.bitmap_pairs && tune.bitmap_delta_rows && args.mv_delta8 && args.mv4 && n_agents == 32 && tune.force_k != 8 && tune.force_k != 2 &&
            layout_fits(n_agents, 4, args, delta_lds, &block, &Q) && args.n_envs % (delta_block / 8u) == 0 &&
            delta_lds + (delta_block / 8u) * per_env <= kLdsBytes) {
            block = delta_block;
            K = 4;
            bitmap = true;
            delta_rows = true;
            lds_bytes = delta_lds;
        } else
        if (tune.bitmap_pairs && n_agents == 32 && tune.force_k != 8 && tune.force_k != 2 && layout_fits(n_agents, 4, args, bitmap_lds, &block, &Q) &&
            args.n_envs % (bitmap_block / 8u) == 0 && bitmap_lds + (bitmap_block / 8u) * per_env <= kLdsBytes) {
            block = bitmap_block;
            K = 4;
            bitmap = true;
            // where the five-column table (STAY included: four selects per agent-step less) still leaves room for the block's
            // bitmaps -- 64 of them on the 64x64 maps, not 128 -- it is the one staged (C5's share: profiles/r04_c5_stay_column_ab.txt)
            stay_column = tune.bitmap_stay_column && lds_bytes + (bitmap_block / 8u) * per_env <= kLdsBytes;
            if (!stay_column) lds_bytes = bitmap_lds;
        } else if ((tune.force_k == 0 || tune.force_k == 8) && n_agents == 32 && layout_fits(n_agents, 8, args, lds_bytes, &block, &Q) &&
                   args.n_envs % (512u / 4u) == 0 && (tune.force_k == 8 || args.n_envs * 4u >= tune.oct_min_lanes)) {
            block = 512u;
            K = 8;
        } else {
            if (tune.force_k == 8 || !layout_fits(n_agents, 4, args, lds_bytes, &block, &Q) || Q < 4) return false;
            block = 512u;
            if (args.n_envs % (1024u / unsigned(Q)) == 0 && args.n_envs * uint64_t(Q) >= uint64_t(n_cu) * 1024u) block = 1024u;
            if (args.n_envs % (block / unsigned(Q)) != 0) return false;
            K = 4;
        }
        compact = true;
    }
    plan->K = K;
    plan->Q = Q;
    plan->form = delta_rows ? 5 : (full_rows_bitmap ? 4 : (bitmap ? (stay_column ? 3 : 2) : (compact ? 1 : 0)));
    plan->block = block;
    plan->lds_bytes = lds_bytes;
    plan->lds_total = lds_bytes + (bitmap ? size_t(block / unsigned(Q)) * bitmap_stride(args.c.n_cells) : 0u);   // (as launch_impl adds them)
    return true;
}

// the device's CU count, asked once per device
static int device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// true when a packed layout took the launch (*err = its status); false = not applicable, use the lane-group kernel
bool try_launch_rollout_lq(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err) {
    LqPlan plan;
    if (!plan_rollout_lq(n_agents, args, tune, device_cu_count(), &plan)) return false;
    const bool record = args.rec_local != nullptr;
    const uint32_t A = uint32_t(n_agents);
    const int K = plan.K, Q = plan.Q, form = plan.form;
    const unsigned block = plan.block;
    const size_t lds_bytes = plan.lds_bytes;
    if (record && !(args.rec_reward && args.rec_prob && args.rec_done && args.rec_collision)) {
        *err = hipErrorInvalidValue;
        return true;
    }
    if (K == 8) *err = record ? launch_rollout_lq_k8_r1(Q, form, args, A, block, lds_bytes, stream) : launch_rollout_lq_k8_r0(Q, form, args, A, block, lds_bytes, stream);
    else if (K == 4) *err = record ? launch_rollout_lq_k4_r1(Q, form, args, A, block, lds_bytes, stream) : launch_rollout_lq_k4_r0(Q, form, args, A, block, lds_bytes, stream);
    else *err = record ? launch_rollout_lq_k2_r1(Q, form, args, A, block, lds_bytes, stream) : launch_rollout_lq_k2_r0(Q, form, args, A, block, lds_bytes, stream);
    return true;
}
#endif

}  // namespace mapf
