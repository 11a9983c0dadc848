// Packed-layout single step: one MapfEnv.step() of every env per launch (mapf_step with device-drawn uniforms), K = 2 or
// 4 agents per lane, Q = A/K lanes per env -- the lane layout of mapf_lq_rollout.hip applied to the one-launch-per-step
// path, for full groups and batches that fill their blocks; everything else (caller-supplied uniforms, odd agent
// counts, ragged batches) stays with lg_step_kernel (mapf_lg_kernels.hip), whose semantics this kernel reproduces
// line by line.
//
// A single step is launch- and latency-bound (argument block -> state / actions -> table rows -> stores, with ~225 vector
// instructions in between and ~1 us of dispatch overhead around it): what the kernel is built around is described at its
// template below; the measurements behind every choice are profiles/r03_step_stamps_*.txt, r03_step_index_costs.txt,
// r03_kernel_end_costs.txt (round 3) and r04_step_stamps_65536.txt, r04_step_table_forms.txt, r04_single_step_scaling.txt
// (round 4: one Philox call per lane, scalar t_dev load, no block barrier, 8-byte table rows) -- DESIGN.md 4.2.
#include "mapf_lq.hpp"

#include <type_traits>

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
// A single step is a chain of memory round trips with vector work between them; the kernel is written around the
// LENGTH of that chain and the instruction count of its last link (profiles/r03_step_stamps_*.txt: of a 4.1 us launch
// at 65536 envs ~1 us is dispatch / completion overhead, the waves live 2.2 us, and the wave that finishes last is the
// launch's duration):
//   * what the first instructions need travels as LEADING SCALAR ARGUMENTS (14 dwords): this file is compiled with
//     -amdgpu-kernarg-preload-count, so the command processor delivers them in SGPRs with the wave: the pointers of the
//     first loads, the agent count and the block size (reading blockDim.x would be a scalar load in front of the address
//     arithmetic) let those loads go out before any s_load of the argument block has come back; the step index's
//     device-side base (StepArgs::t_dev) is requested at once as well;
//   * state, actions, the scenario byte and the 1 KB table image (slip rows + outcome rows: one 16-byte load per lane, every
//     wave for itself) are requested together; the scenario's rows and the move-table rows are the second trip; the Philox
//     rounds are split over the two waits;
//   * the image lives in LDS (written while the table gathers are in flight; no barrier: see the staging code), so sampling
//     and the per-env outcome are the fused rollout's code: packed 16-bit threshold compares (sample_slot_packed), one
//     integer outcome code per env, reward / flags read from the code's LDS row;
//   * a 16-bit tie is resolved by register arithmetic alone (slip_move_exact_members), only for the slots that tie.
constexpr uint32_t kStepSlipAt = offsetof(TableImage, slip), kStepOutcomeAt = offsetof(TableImage, outcome), kStepLds = sizeof(TableImage);

//   * TERM = an env may be terminal when the step begins.  The host knows when none can (StepArgs::state_not_terminal:
//     the previous call auto-reset every finished episode and no START state is itself terminal) -- the usual training
//     loop -- and the !TERM instance drops is_terminal(prev): the duplicate-cell half of the pair tests, the on-goal test
//     of the current cells and every was-terminal select.
//   * BIG = the form for batches several times larger than the device holds at once (profiles/r04_single_step_scaling.txt):
//     there the step is bound by the texture path's rate for DIVERGENT gathers -- four 16-byte table rows per lane, 64
//     different cache lines per wave-instruction, ~1 line per cycle -- not by arithmetic or HBM.  A resident grid of
//     1024-thread blocks (two per CU) stages the whole move table into LDS once and walks the batch in chunks of 1024
//     lanes: the table rows become ds_read_b128 (a quarter of the cost), and the argument block, the LDS image and the
//     barrier are paid once per block instead of once per chunk.
#ifndef MAPF_BIG_COLS
#define MAPF_BIG_COLS 6
#endif
constexpr uint32_t kStepMoveAt = 1024, kBigCols = MAPF_BIG_COLS;
// offset of the StepArgs block in the kernel's argument segment: five pointers and four 32-bit scalars precede it
// (compared with the .args metadata of every compiled instance by tests/test_cabi_and_host.py: a wrong value fails the CPU suite)
constexpr uint32_t kStepArgsOffset = 5 * 8 + 4 * 4;
static_assert(kStepArgsOffset % alignof(StepArgs) == 0, "the block follows the leading scalars without padding");
static_assert(kStepLds <= kStepMoveAt, "LDS image of the BIG form: slip rows, outcome rows, then the move table");

// BIG == 2: the same resident grid with the table as 4-BYTE DELTA ROWS (StepArgs::mv4, six columns, 24 bytes per cell): what
//     lets a 64x64 map's table (79 KB; 316 KB as 16-byte rows) sit in LDS at all.  That is the 32-agent configurations' form:
//     their plain step is bound by the texture path's line rate too -- 2048 lanes x 4 divergent gathers per CU are ~3.4 us of a
//     4.9 us launch at 16384 envs (profiles/r05_step_delta_*) -- at every batch size, not just the large ones.  The row gives the
//     three candidates as signed byte offsets from the lane's own cell and the slip row's offset; the thresholds come from that
//     row in LDS (a second, dependent LDS read), the sampled cell from one add with a sign-extending byte select
//     (sample_slot_delta, as in the 32-agent rollout).
template <int Q, int K, bool SCEN, bool TERM, int BIG = 0>
__global__ void __launch_bounds__(BIG ? 1024 : 512) lq_step_kernel(uint16_t *const state, const uint8_t *const actions, const uint8_t *const scen,
                                                      const SlipRow *const slip_rows, const uint64_t *const t_dev,
                                                      const uint32_t agents_block, const uint32_t t_lo, const uint32_t seed_lo,
                                                      const uint32_t seed_hi, const StepArgs p_block, const uint32_t n_chunks) {
    constexpr int P = K / 2;
#ifdef MAPF_STEP_STAMPS
    unsigned long long stamp_[8] = {}, real0_, cyc0_;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0_), "=s"(cyc0_) :: "memory");
#endif
    // the kernel's LDS image (mapf_lq.hpp): 1 KB static (the table image: slip rows, outcome rows) reached through the
    // object; the BIG form's image is the dynamic segment -- the move table behind those 1 KB -- used as a raw scratchpad
    using Image = std::conditional_t<BIG != 0, LdsAbsolute, LdsObject>;
    Image lds;
    if constexpr (!BIG) {
        __shared__ __attribute__((aligned(16))) unsigned char lds_static[kStepLds];
        lds.base = (lds_ptr)lds_static;
    }
    const uint32_t n_agents = agents_block & 0xFFu, block_threads = agents_block >> 8;
    // BIG == 3: ... and per-env ONE-BIT occupancy bitmaps behind the table (bitmap_pair_tests, mapf_lq.hpp: the 32-agent rollout's
    // O(A) collision tests) -- with the gathers gone the 32-agent step is bound by its 496 agent pairs per env
    const uint32_t bitmap_stride = (((p_block.c.n_cells + 31u) >> 5) * 4u + 15u) & ~15u;    // bytes per env: one bit per cell
    const uint32_t bitmap_base = kStepMoveAt + uint32_t(delta_table_words(p_block.c.n_cells)) * 4u;
    const uint32_t bitmap_at = bitmap_base + (threadIdx.x / uint32_t(Q)) * bitmap_stride;
    // BIG == 2: the host-built delta rows as they are (StepArgs::mv4), 16 bytes per thread and load, TEN loads in flight -- a
    // 64x64 map's 79 KB are one round trip for a 512-thread block (in rounds of four loads the copy alone was ~3 us of the
    // launch) -- requested BEHIND the block's first trip (see the chunk loop), so that state and actions travel with the table
    auto stage_delta_table = [&]() __attribute__((always_inline)) {
        constexpr uint32_t kInFlight = 10;
        const uint32_t n_vec = uint32_t(delta_table_words(p_block.c.n_cells) / 4u);
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p_block.mv4);
        for (uint32_t w0 = threadIdx.x; w0 < n_vec; w0 += kInFlight * block_threads) {
            u32x4 part[kInFlight];
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k) part[k] = src[min(w0 + k * block_threads, n_vec - 1u)];
#pragma unroll
            for (uint32_t k = 0; k < kInFlight; ++k)
                if (w0 + k * block_threads < n_vec) *(__attribute__((address_space(3))) u32x4 *)lds_addr(lds, kStepMoveAt + 16u * (w0 + k * block_threads)) = part[k];
        }
        if constexpr (BIG == 3) {   // the block's occupancy bitmaps (one per env of a chunk; every use clears what it set)
            const uint32_t n_words = (block_threads / uint32_t(Q)) * (bitmap_stride >> 2);
            for (uint32_t w = threadIdx.x; w < n_words; w += block_threads) *(lds_u32)lds_addr(lds, bitmap_base + 4u * w) = 0u;
        }
        stage_outcome_table(p_block.c, lds_generic<OutcomeRow>(lds, kStepOutcomeAt));
        stage_slip_table(slip_rows, lds_generic<SlipRow>(lds, kStepSlipAt));   // ends with __syncthreads()
    };
    if constexpr (BIG >= 2) {
    } else
    if (BIG) {   // move table -> LDS: 16-byte rows, SIX columns per cell (kBigCols: column 5 = STAY again, so that an action byte is
        // extracted and clamped by one v_min_u32 -- in LDS the sixth column costs room, not gather traffic); four independent
        // loads per thread and round; the thresholds bias-shifted as the packed sampling compares them
        const uint32_t n_rows = p_block.c.n_cells * kBigCols;
        MoveEntry *const dst = lds_generic<MoveEntry>(lds, kStepMoveAt);
        auto source = [](uint32_t w) __attribute__((always_inline)) {
            const uint32_t cell = w / kBigCols, col = w - cell * kBigCols;
            return cell * kMvCols + (col < kMvCols ? col : 0u);
        };
        auto biased = [](MoveEntry r) __attribute__((always_inline)) { r.z ^= kHalfBias; return r; };
        for (uint32_t w0 = threadIdx.x; w0 < n_rows; w0 += 4u * block_threads) {
            const uint32_t w1 = w0 + block_threads, w2 = w1 + block_threads, w3 = w2 + block_threads, last = n_rows - 1u;
            const MoveEntry r0 = p_block.mv[source(w0)], r1 = p_block.mv[source(min(w1, last))], r2 = p_block.mv[source(min(w2, last))],
                            r3 = p_block.mv[source(min(w3, last))];
            dst[w0] = biased(r0);
            if (w1 < n_rows) dst[w1] = biased(r1);
            if (w2 < n_rows) dst[w2] = biased(r2);
            if (w3 < n_rows) dst[w3] = biased(r3);
        }
        // ... and the slip / outcome rows, behind ONE barrier: every chunk of the block then finds the whole image in place
        stage_outcome_table(p_block.c, lds_generic<OutcomeRow>(lds, kStepOutcomeAt));
        stage_slip_table(slip_rows, lds_generic<SlipRow>(lds, kStepSlipAt));   // ends with __syncthreads()
    }
    // ---- first trip of a chunk: state, actions, scenario byte -- issued by themselves, so that the BIG form can request the
    // NEXT chunk's before it computes the current one (a third of a wave's cycles there went to waiting for this trip)
    struct FirstTrip { Packed<K / 2> cells; uint64_t raw; uint32_t scen_id; };
    auto first_trip = [&](const uint32_t chunk) __attribute__((always_inline)) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t e = ((chunk * block_threads + threadIdx.x) >> 6) * uint32_t(64 / Q) + lane / uint32_t(Q);
        const uint32_t lane_cell = e * n_agents + uint32_t(K) * (lane & uint32_t(Q - 1));
        FirstTrip f;
        f.cells = Packed<K / 2>::load(at(state, lane_cell));
        if constexpr (K == 8) f.raw = *reinterpret_cast<const uint64_t *>(at(actions, lane_cell));   // one action byte per agent of the lane
        else f.raw = K == 4 ? uint64_t(*reinterpret_cast<const uint32_t *>(at(actions, lane_cell)))
                            : uint64_t(*reinterpret_cast<const uint16_t *>(at(actions, lane_cell)));
        f.scen_id = 0u;
        if (SCEN) f.scen_id = *at(scen, e);
        return f;
    };
    // (!BIG: one block per 256 lanes, one pass.  A resident grid WITHOUT the LDS table was measured 6-8 % slower than that.)
    auto one_chunk = [&](const uint32_t chunk, const FirstTrip &in, auto first_tag) __attribute__((always_inline)) {
    // BIG: the loop would keep every field of the argument block live in SGPRs across iterations (106 of them, one block per
    // CU); the fields are re-read from the kernarg segment through a pointer the optimiser cannot see through, so that they
    // are fetched where an iteration uses them, as in the straight-line form (scalar cache hits).
    union { StepArgs args; uint32_t words[sizeof(StepArgs) / 4]; } reread;
    if constexpr (BIG) {
        using KernWord = const __attribute__((address_space(4))) uint32_t;
        KernWord *ka = (KernWord *)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + kStepArgsOffset);
        asm volatile("" : "+s"(ka));
#pragma unroll
        for (uint32_t i = 0; i < sizeof(StepArgs) / 4; ++i) reread.words[i] = ka[i];
    }
    const StepArgs &p = BIG ? reread.args : p_block;
    constexpr bool first_pass = decltype(first_tag)::value;   // !BIG: the LDS image is written behind the chunk's first loads
    LaneCtx<Q> x;
    x.lane = threadIdx.x & 63u;
    x.g = x.lane & uint32_t(Q - 1);
    x.base = x.lane & ~uint32_t(Q - 1);
    x.e = ((chunk * block_threads + threadIdx.x) >> 6) * uint32_t(64 / Q) + x.lane / uint32_t(Q);
    x.v0 = x.v1 = true;
    const uint32_t e = x.e;
    const uint32_t lane_cell = e * n_agents + uint32_t(K) * x.g;    // my first agent's element index
    const uint32_t fixed_cell = uint32_t(K) * x.g;                  // ... in a broadcast row

    // ---- first trip (requested by first_trip()): state, actions, scenario byte; with it the table image / my cells of the goal row
    uint32_t c[P], g[P], sc[P];
    const Packed<P> cells = in.cells;
    const uint64_t raw = in.raw;
    const uint32_t scen_id = in.scen_id;
    Packed<P> gl{}, sl{};
    // (!BIG) EVERY wave fetches the 1 KB table image (slip rows + outcome rows, built on the host: TableImage) with one
    // 16-byte load per lane and writes it to LDS itself: the waves of a block write identical bytes to identical addresses
    // and a wave's LDS operations execute in order, so each wave may read the image right after its own write -- no barrier
    // couples the block's waves (the barrier made all four wait for the slowest one's loads)
    u32x4 image_word = {0u, 0u, 0u, 0u};
    const bool stager = first_pass;
    if (stager) image_word = reinterpret_cast<const u32x4 *>(slip_rows)[x.lane];
    // the recording's first step index: a SCALAR load (constant address space: nothing writes it while this kernel runs).
    // As a vector load + readfirstlane it put an s_waitcnt vmcnt(0) -- i.e. the whole first trip -- in front of the Philox rounds
    const uint64_t t_base = t_dev ? *(const __attribute__((address_space(4))) uint64_t *)(uintptr_t)t_dev : 0ull;
    __builtin_amdgcn_sched_barrier(0);
    if (!SCEN) gl = Packed<P>::load(gat(p.goal, p.goal_broadcast ? fixed_cell : lane_cell));

    // ---- the slip call of my quad(s) for steps 2h, 2h+1 (ONE call per four agents: the stream's call unit is what a lane
    // of this kernel owns, so a single step pays one call per lane and uses half of its words; with two agents per lane the
    // neighbour lane repeats the call), first six rounds while the loads are in flight
    // (p.t's low word is preloaded; the high word -- zero for the first 2^32 steps of a handle -- comes with the block)
    const uint64_t env_id = p.env_id_offset + e;
    const uint64_t t = ((p.t & 0xFFFFFFFF00000000ull) | t_lo) + t_base;
    constexpr int NQ = K >= 4 ? K / 4 : 1;   // calls per lane
    PhiloxRounds<NQ> rng_state;
    {
        const uint32_t hi16 = uint32_t((t >> 1) >> 32) & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            rng_state.c[i][0] = uint32_t(env_id); rng_state.c[i][1] = uint32_t(env_id >> 32); rng_state.c[i][2] = uint32_t(t >> 1);
            rng_state.c[i][3] = hi16 | ((K >= 4 ? uint32_t(NQ) * x.g + uint32_t(i) : x.g >> 1) << 16);   // quad index; rslot = refine = 0 (slip_words)
        }
        rng_state.k0 = seed_lo; rng_state.k1 = seed_hi;
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
        gl = Packed<P>::load(gat(p.scen_rows, scen_row + n_agents));   // (as_global: see mapf_lq.hpp -- the BIG form's
        sl = Packed<P>::load(gat(p.scen_rows, scen_row));              //  argument block is re-read, its pointers generic)
    }
#pragma unroll
    for (int i = 0; i < P; ++i) c[i] = cells.v[i];
    uint32_t cur[K], act[K];
    MoveEntry entry[K];
    CompactEntry compact[K];
    uint32_t delta_row[K];
    const uint32_t last_cell = p.c.n_cells - 1u;
    constexpr uint32_t kCols = BIG ? kBigCols : kMvCols;
    uint32_t row_bytes = kDeltaCols * 4u, eight = 8u;
    asm volatile("" : "+v"(row_bytes), "+v"(eight));   // (SDWA operands must be vector registers)
    // the LDS image first (it arrived with the first trip): the plain form reads its thresholds from it right behind the gathers
    if (stager) *(__attribute__((address_space(3))) u32x4 *)lds_addr(lds, 16u * x.lane) = image_word;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // cell: half-word extract + clamp (a corrupted state must not turn into a wild read) in one v_min_u32 with a word
        // select; action: an out-of-range byte is STAY (with a sixth table column -- STAY again -- one v_min_u32, byte select)
        cur[k] = min((k & 1) ? c[k / 2] >> 16 : c[k / 2] & 0xFFFFu, last_cell);
        if constexpr (kCols == 6) act[k] = min(uint32_t(raw >> (8 * k)) & 0xFFu, 5u);
        else { const uint32_t byte = uint32_t(raw >> (8 * k)) & 0xFFu; act[k] = byte > 4u ? 0u : byte; }
        uint32_t row = __umul24(cur[k], kCols) + act[k];
        asm volatile("" : "+v"(row));             // (keep row * 8 + base as one shift-add)
        // BIG: 16-byte rows from the LDS copy.  Plain: 8-BYTE rows from global memory (CompactEntry) -- a launch's gathers are
        // bound by the texture path's line rate and every launch re-fetches the table into eight L2s, so half the bytes is
        // what counts; the code's thresholds then come from the slip row in LDS (profiles/r04_step_table_forms.txt)
        // BIG == 2: 4-byte delta rows from the LDS copy, cell * 24 + action * 4 (an out-of-range LDS address reads zeros: no clamp)
        if (BIG >= 2) delta_row[k] = lds_at<uint32_t>(lds, kStepMoveAt + (act[k] << 2) + ((k & 1) ? half_times<1>(c[k / 2], row_bytes) : half_times<0>(c[k / 2], row_bytes)));
        else if (BIG) entry[k] = lds_entry_at(lds, kStepMoveAt + row * 16u);
        else compact[k] = p.mv8[row];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (p.c.need_rng) rng_state.template run<4>();
#ifdef MAPF_STEP_STAMPS
    { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_[3] = t_ - cyc0_; __builtin_amdgcn_sched_barrier(0); }   // gathers issued, four rounds done
    STEP_STAMP(4);   // gathers arrived
#endif
#pragma unroll
    for (int i = 0; i < P; ++i) { g[i] = gl.v[i]; sc[i] = sl.v[i]; }
    // (the start cells are consumed HERE, with the goal cells: left pending until the state store at the end, their wait would
    // sit behind the output stores -- whose number the compiler cannot count across the null-pointer branches -- as a full
    // s_waitcnt vmcnt(0), i.e. a wait for the stores' acknowledgements and for the next chunk's prefetch)
    if (SCEN) {
#pragma unroll
        for (int i = 0; i < P; ++i) asm volatile("" : "+v"(sc[i]));
    }
    if constexpr (BIG >= 2) {   // the code's thresholds: a second LDS read that depends on the first; the row completes to a MoveEntry
        uint32_t row_off[K], th[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(row_off[k]) : "v"(delta_row[k]), "v"(eight));
            th[k] = lds_at<uint32_t>(lds, kStepSlipAt + uint32_t(offsetof(SlipRow, th_biased)) - kDeltaRowBias + row_off[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) entry[k] = make_uint4(delta_row[k], 0u, th[k], row_off[k]);
    }
    if constexpr (!BIG) {   // complete the rows: thresholds (bias-shifted, as the packed sampling compares them) by the code's row offset
        uint32_t th[K];
#pragma unroll
        for (int k = 0; k < K; ++k) th[k] = lds_at<uint32_t>(lds, kStepSlipAt + uint32_t(offsetof(SlipRow, th_biased)) + (compact[k].y >> 16));
#pragma unroll
        for (int k = 0; k < K; ++k) entry[k] = make_uint4(compact[k].x, compact[k].y, th[k], compact[k].y >> 16);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- sampling (the fused rollout's packed form): both threshold compares of an agent in one saturating packed
    // subtract, the slot's probability address and cell selector from one dot product each
    // (delta rows: the slot selects a byte -- steps of one, the row's byte 2 down to 0, zeros above it)
    uint32_t pk_eights = 0x00080008u, pk_steps = BIG >= 2 ? 0x00010001u : 0x02020202u, sel_base = BIG >= 2 ? 0x0C0C0C02u : 0x0C0C0504u;
    asm volatile("" : "+v"(pk_eights), "+v"(pk_steps), "+v"(sel_base));
    double q[K];
    uint32_t n[P], word[P], d[K], tie_all = 0u;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        {   // pair i of the lane: word 2 * (t & 1) + (pair's place in its quad) of the quad's call
            const uint32_t (&cw)[4] = rng_state.c[K >= 4 ? i / 2 : 0];
            const uint32_t place = K >= 4 ? uint32_t(i & 1) : (x.g & 1u);
            word[i] = p.c.need_rng ? quad_step_word(Words4{cw[0], cw[1], cw[2], cw[3]}, t, place) : 0u;
        }
        const uint32_t biased = word[i] ^ kHalfBias;                 // low half: agent 2i's uniform, high half: agent 2i+1's
        uint32_t q_at[2], cell[2];
        MoveEntry e0 = entry[2 * i], e1 = entry[2 * i + 1];
        // (the thresholds are bias-shifted already: th_biased / the BIG form's LDS copy of the table)
        if constexpr (BIG >= 2) {
            d[2 * i] = sample_slot_delta<0>(e0.x, e0.z, e0.w, __builtin_amdgcn_perm(biased, biased, 0x01000100u), pk_eights, pk_steps, sel_base, c[i], q_at[0], cell[0]);
            d[2 * i + 1] = sample_slot_delta<1>(e1.x, e1.z, e1.w, __builtin_amdgcn_perm(biased, biased, 0x03020302u), pk_eights, pk_steps, sel_base, c[i], q_at[1], cell[1]);
            q[2 * i] = lds_at<double>(lds, kStepSlipAt + 16u - kDeltaRowBias + q_at[0]);
            q[2 * i + 1] = lds_at<double>(lds, kStepSlipAt + 16u - kDeltaRowBias + q_at[1]);
        } else {
        d[2 * i] = sample_slot_packed(e0, __builtin_amdgcn_perm(biased, biased, 0x01000100u), pk_eights, pk_steps, sel_base, q_at[0], cell[0]);
        d[2 * i + 1] = sample_slot_packed(e1, __builtin_amdgcn_perm(biased, biased, 0x03020302u), pk_eights, pk_steps, sel_base, q_at[1], cell[1]);
        q[2 * i] = lds_at<double>(lds, kStepSlipAt + 16u + q_at[0]);
        q[2 * i + 1] = lds_at<double>(lds, kStepSlipAt + 16u + q_at[1]);
        }
        n[i] = cell[0] | (cell[1] << 16);
        tie_all = i == 0 ? pk_min_u16(d[0], d[1]) : pk_min_u16(tie_all, pk_min_u16(d[2 * i], d[2 * i + 1]));
    }
    // (without slip the words are zero and every threshold is 65535: no tie can fire)
    if (__builtin_expect(__any(zero_half(tie_all) != 0u), 0)) {
        // A top-16-bit tie somewhere in the wave (about one wave in 80 at 8 agents): the agent slots that tie are redone
        // with all 53 bits.  This wave finishes last, i.e. it IS the launch's duration, so the redo is one Philox call per
        // tying slot and register arithmetic only (the four calls + three dependent loads of the slip rows this path
        // once cost kept the slowest wave alive 1.1 us after 90 % of the others had left).
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (__any(zero_half(d[k]) != 0u)) {
                const uint32_t hi = (k & 1) ? word[k / 2] >> 16 : word[k / 2] & 0xFFFFu;
                uint32_t nx;
                MoveEntry full = entry[k];
                // (8-byte rows: the list's members, where slip_move_exact_members looks for them, from the code's slip row)
                if (!BIG) full.y = (full.y & 0xFFFFu) | (lds_at<uint32_t>(lds, kStepSlipAt + uint32_t(offsetof(SlipRow, members)) + full.w) << 19);
                if (BIG >= 2) {   // the candidates' cells back from their deltas, the members from the slip row
                    const uint32_t mine = (k & 1) ? c[k / 2] >> 16 : c[k / 2] & 0xFFFFu, row = full.x;
                    const uint32_t c0 = (mine + uint32_t(int32_t(int8_t(row)))) & 0xFFFFu, c1 = (mine + uint32_t(int32_t(int8_t(row >> 8)))) & 0xFFFFu,
                                   c2 = (mine + uint32_t(int32_t(int8_t(row >> 16)))) & 0xFFFFu;
                    full.x = c0 | (c1 << 16);
                    full.y = c2 | (lds_at<uint32_t>(lds, kStepSlipAt + uint32_t(offsetof(SlipRow, members)) - kDeltaRowBias + full.w) << 19);
                }
                slip_move_exact_members(p.c, full, refine_mantissa(p.c, env_id, t, uint32_t(K) * x.g + uint32_t(k), hi), nx, q[k]);
                n[k / 2] = (k & 1) ? (n[k / 2] & 0xFFFFu) | (nx << 16) : (n[k / 2] & 0xFFFF0000u) | nx;
            }
        }
    }

    // ---- is_terminal(prev) and the collision tests in one pass over the agent pairs; the per-env facts as ONE integer
    // code = vertex | swap << 1 | off_goal << 2 | was_terminal << 3 (mapf_env.py:210-223, :225-235, :378-389)
    PairAcc<true> acc;
    if constexpr (BIG == 3) {
        acc = bitmap_pair_tests<Q, K>(x, lds, bitmap_at, c, n);   // (a terminal env's bits are set and cleared like any other's: its outcome row ignores them)
        if (TERM) {                                                // is_terminal(prev)'s duplicate test stays with the agent pairs
            const uint32_t none[P] = {};
            acc.dup = packed_pair_tests<Q, P, true, false>(x, c, none).dup;
        }
    } else {
        acc = packed_pair_tests<Q, P, TERM, true>(x, c, n);
    }
    uint32_t away_next = n[0] ^ g[0], away_prev = TERM ? c[0] ^ g[0] : 1u;
#pragma unroll
    for (int i = 1; i < P; ++i) { away_next |= n[i] ^ g[i]; if (TERM) away_prev |= c[i] ^ g[i]; }
    asm volatile("" : "+v"(away_next), "+v"(away_prev));   // stay integers: as compares they would travel through scalar masks
    // zero_half() leaves bits 15 / 31: vertex -> bits 0 / 16, swap -> bits 1 / 17, dup -> bits 3 / 19; halves folded together
    uint32_t bits = (zero_half(acc.vertex) >> 15) | (zero_half(acc.swap) >> 14) | (TERM ? zero_half(acc.dup) >> 12 : 0u);
    bits |= bits >> 16;
    const uint32_t flags = group_reduce<Q, false>((min(away_next, 1u) << 2) | (TERM ? min(away_prev, 1u) << 4 : 0u) | (bits & 0xBu), x);
    // was_terminal: two agents share a cell (bit 3), or every agent is on its goal (bit 4 clear)
    const uint32_t term = TERM ? ((flags >> 3) | (~flags >> 4)) & 1u : 0u;
    const uint32_t code16 = ((flags & 7u) | (term << 3)) << 4;
    const bool was_terminal = TERM && code16 > 7u * 16u;
    const u32x4 row = lds_at<u32x4>(lds, kStepOutcomeAt + code16);      // {reward lo, hi, status, done | collision << 16}
    double reward = __hiloint2double(int(row.y), int(row.x));
    if (p.c.criteria == 1u) {
        // _living_reward: mapf_env.py:436-446
        uint32_t mine = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t goal_k = (k & 1) ? g[k / 2] >> 16 : g[k / 2] & 0xFFFFu;
            mine += (cur[k] == goal_k && (act[k] == 0u || act[k] == 5u)) ? 1u : 0u;   // (5 = an out-of-range byte: STAY)
        }
        const int stayed = int(group_reduce<Q, true>(mine, x));
        const double living = __dmul_rn(double(int(n_agents) - stayed), p.c.r_living);
        const uint32_t f = (code16 >> 4) & 7u;
        const bool coll_ = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
        const double r = coll_ ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
        reward = was_terminal ? 0.0 : r;                           // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0})
    }
    const uint32_t done_coll = row.w;                              // done | collision << 16
    const bool done = (done_coll & 1u) != 0u;
    // a step from a terminal state changes nothing
#pragma unroll
    for (int i = 0; i < P; ++i) n[i] = was_terminal ? c[i] : n[i];

    // total_prob: left-to-right product over agents 0..A-1 (mapf_env.py:257); the total ends in the group's last lane
    q[0] = was_terminal ? 0.0 : q[0];                              // (a zero factor makes the whole product +0.0)
    const double prob = packed_prob_product<Q, K>(q);

    Packed<P> out;
#pragma unroll
    for (int i = 0; i < P; ++i) out.v[i] = n[i];
#ifdef MAPF_STEP_STAMPS
    asm volatile("" :: "v"(reward), "v"(prob), "v"(out.v[0]));
    STEP_STAMP(5);   // everything computed
#endif
    if (p.out_local) out.store_global(gat(p.out_local, lane_cell));
    if (x.g == uint32_t(Q - 1) && p.out_prob) *gat(p.out_prob, e) = prob;
    if (x.g == 0u) {
        if (p.out_reward) *gat(p.out_reward, e) = reward;
        if (p.out_done) *gat(p.out_done, e) = uint8_t(done_coll);
        if (p.out_collision) *gat(p.out_collision, e) = uint8_t(done_coll >> 16);
        if (p.out_was_terminal) *gat(p.out_was_terminal, e) = was_terminal ? 1 : 0;
    }
    if (SCEN) {                                                    // MapfEnv.reset(): start cells, no reseed
        const bool back = p.auto_reset && done;                    // (a terminal state that stays is rewritten as it is)
        Packed<P> keep;
#pragma unroll
        for (int i = 0; i < P; ++i) keep.v[i] = back ? sc[i] : n[i];
        keep.store(at(state, lane_cell));
    } else if (p.auto_reset && done) {
        Packed<P>::load(gat(p.start, p.start_broadcast ? fixed_cell : lane_cell)).store(at(state, lane_cell));
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
                                      uint64_t((chunk * block_threads + threadIdx.x) >> 6) * 12u;
            dst[0] = real0_; dst[1] = real1_;
            for (int i = 0; i < 6; ++i) dst[2 + i] = stamp_[i];
            dst[8] = t6_ - cyc0_; dst[9] = t7_ - cyc0_; dst[10] = hw_id; dst[11] = xcc_id;
        }
    }
#endif
    };   // one_chunk
    if constexpr (BIG) {
        // Two chunks per iteration, their first trips in two sets of registers that take turns: "cur = next" would be a
        // register move of values still in flight, i.e. a wait for the very loads the prefetch is there to hide.
        const uint32_t stride = gridDim.x;
        FirstTrip a = first_trip(blockIdx.x);
        if constexpr (BIG >= 2) {
            __builtin_amdgcn_sched_barrier(0);                           // (the first trip's requests go out ahead of the table's)
            stage_delta_table();
        }
#pragma nounroll
        for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += 2u * stride) {
            const uint32_t second = chunk + stride, third = second + stride;
            const FirstTrip b = first_trip(second < n_chunks ? second : chunk);   // (clamped: past the end it re-reads its own)
            __builtin_amdgcn_sched_barrier(0);                       // the requests go out HERE, ahead of this chunk's work
            one_chunk(chunk, a, std::false_type{});
            if (second >= n_chunks) break;
            a = first_trip(third < n_chunks ? third : second);
            __builtin_amdgcn_sched_barrier(0);
            one_chunk(second, b, std::false_type{});
        }
    } else {
        const FirstTrip in = first_trip(blockIdx.x);
        one_chunk(blockIdx.x, in, std::true_type{});
    }
    signal_step_done(p_block.done_flag, p_block.done_seq);
}

}  // namespace

// true when the packed layout took the launch (*err = its status); false = not applicable, use lg_step_kernel
bool try_launch_step_lq(int n_agents, const StepArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err) {
#ifndef MAPF_STEP_STAMPS   // (the diagnostic build receives its stamp buffer through `uniforms`)
    if (args.uniforms != nullptr) return false;
#endif
    // top_tie: a three-entry list whose last cumulative sum rounds below 1.0 needs a third compare per agent; the packed
    // sampling does two (as in the packed rollout), so such a table stays with the lane-group kernel
    if (!tune.quad_lanes || args.c.top_tie) return false;
    int K = 0;
    if (tune.force_k != 2 && n_agents % 4 == 0) K = 4;
    else if (tune.force_k != 4 && n_agents % 2 == 0 && n_agents >= 4) K = 2;
    else return false;
    const int Q = n_agents / K;
    if (Q > 16 || (Q & (Q - 1)) != 0) return false;
    const uint64_t lanes = args.n_envs * uint64_t(Q);
    const uint32_t A = uint32_t(n_agents);
    const bool scen = args.scen != nullptr, term = !args.state_not_terminal;
    const uint8_t *const no_scen = nullptr;
    // The BIG form (resident grid, move table in LDS): batches several times what the device holds at once
    // (profiles/r04_single_step_scaling.txt), a table that leaves room for two 1024-thread blocks per CU.
    // MAPF_TUNE step_big=0 never, =2 whenever it fits.
    const size_t big_lds = kStepMoveAt + size_t(args.c.n_cells) * kBigCols * sizeof(MoveEntry);
    int n_cu = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const uint64_t resident_lanes = uint64_t(n_cu) * 2048u;
    const bool big_fits = K == 4 && Q <= 8 && tune.step_big != 0 && args.n_envs > 0 && args.n_envs % (1024u / unsigned(Q)) == 0 &&
                          2u * big_lds <= 160u * 1024u;
    const bool big = big_fits && (tune.step_big == 2 || lanes >= 4u * resident_lanes);
    // ... with EIGHT agents per lane where the team allows it (8, 16, 32 agents): the large-batch step is bound by its vector
    // instructions once the gathers are gone, and what a lane does once per env (lane context, flags, outcome row, stores,
    // the hand-over of the probability product) is then paid for 64 envs per wave instead of 32
    // (measured, 8 agents: 0.289 against 0.262 at 0.5 M envs, 0.35 / 0.41 / 0.42 at 1 / 2 / 4 M -- this form from TWICE the
    // device's resident lanes on, the four-agents-per-lane form below from four times)
    if (big_fits && (tune.step_big == 2 || lanes >= 2u * resident_lanes) && Q >= 2 && args.n_envs % (1024u / unsigned(Q / 2)) == 0 &&
        tune.force_k != 4) {
        const int Q8 = Q / 2;
        const unsigned block = 1024u, n_chunks = unsigned(args.n_envs * uint64_t(Q8) / block), grid = n_chunks < unsigned(n_cu) ? n_chunks : unsigned(n_cu);
        note_kernel("lq_step_kernel<Q=%d,K=8%s%s,BIG> block=1024 resident grid (packed layout: 8 agents per lane, move table in LDS%s)", Q8,
                    scen ? ",SCEN" : "", term ? "" : ",NO_TERMINAL", scen ? ", start / goal rows from the scenario table" : "");
#define MAPF_LQ_BIG8(QQ, SS, TT, SCEN_PTR)                                                                                          \
        {                                                                                                                           \
            auto kern = lq_step_kernel<QQ, 8, SS, TT, 1>;                                                                        \
            if (big_lds > 32u * 1024u) { if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(160u * 1024u - 1024u))) { *err = e; return true; } } \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), big_lds, stream, args.state, args.actions, SCEN_PTR, args.slip, args.t_dev, \
                               A | (block << 8), uint32_t(args.t), args.c.seed_lo, args.c.seed_hi, args, n_chunks);                 \
        }
#define MAPF_LQ_BIG8_Q(QQ)                                                                       \
        if (Q8 == QQ) {                                                                           \
            if (scen && term) MAPF_LQ_BIG8(QQ, true, true, args.scen)                             \
            else if (scen) MAPF_LQ_BIG8(QQ, true, false, args.scen)                               \
            else if (term) MAPF_LQ_BIG8(QQ, false, true, no_scen)                                 \
            else MAPF_LQ_BIG8(QQ, false, false, no_scen)                                          \
            *err = hipGetLastError();                                                             \
            return true;                                                                          \
        }
        MAPF_LQ_BIG8_Q(1) MAPF_LQ_BIG8_Q(2) MAPF_LQ_BIG8_Q(4)
#undef MAPF_LQ_BIG8_Q
#undef MAPF_LQ_BIG8
    }
    // The delta-row forms (BIG == 2, 3): where the 16-byte rows do not fit (64x64 maps) but the 4-byte ones do, from a batch of one
    // full residency on (65536 envs of 32 agents) -- below that the table copy per block (79 KB through the XCD's L2 for each of
    // its 32 CUs: 1.8 us in front of the first instruction that needs a row, profiles/r05_step_stamps_c5_share.txt) costs more
    // than the gathers it replaces: configs[4]'s share of one GPU (16384 envs) runs 4.95 us plain against 5.7-5.9 us.
    const size_t delta_lds = kStepMoveAt + delta_table_words(args.c.n_cells) * sizeof(uint32_t);
    if (K == 4 && Q <= 8 && args.mv4 && tune.step_delta != 0 && delta_lds <= 160u * 1024u && args.n_envs > 0 &&
        (tune.step_delta == 2 || (!big_fits && lanes >= resident_lanes))) {
        // 32 agents: the occupancy bitmaps of a chunk's envs behind the table -- one block per CU then, so 1024 threads as soon as
        // every CU gets such a block (measured on configs[4]'s map, profiles/r05_step32_forms.txt: 131072 envs 15.6 us with
        // bitmaps in 1024-thread blocks, 18.0 without, 21.5 for the plain step; at 65536 envs 512-thread blocks with bitmaps
        // 11.7, without 10.8, plain 11.9)
        const size_t per_env = (size_t((args.c.n_cells + 31u) / 32u) * 4u + 15u) & ~size_t(15);
        const bool bitmaps_1024 = Q == 8 && tune.bitmap_pairs && delta_lds + (1024u / 8u) * per_env <= 160u * 1024u && args.n_envs % (1024u / 8u) == 0 &&
                                  lanes >= uint64_t(n_cu) * 1024u;
        unsigned block = (bitmaps_1024 || lanes >= 2u * resident_lanes) ? 1024u : 512u;
        if (args.n_envs % (block / unsigned(Q)) != 0) block = 512u;
        bool bitmaps = Q == 8 && tune.bitmap_pairs && delta_lds + (block / 8u) * per_env <= 160u * 1024u;
        if (!bitmaps && Q == 8 && tune.bitmap_pairs && block == 1024u && delta_lds + (512u / 8u) * per_env <= 160u * 1024u) { block = 512u; bitmaps = true; }
        const size_t form_lds = delta_lds + (bitmaps ? (block / 8u) * per_env : 0u);
        if (args.n_envs % (block / unsigned(Q)) == 0) {
            unsigned per_cu = unsigned((160u * 1024u) / form_lds);
            if (per_cu > 2048u / block) per_cu = 2048u / block;
            const unsigned n_chunks = unsigned(lanes / block), grid = n_chunks < per_cu * unsigned(n_cu) ? n_chunks : per_cu * unsigned(n_cu);
            note_kernel("lq_step_kernel<Q=%d,K=%d%s%s,DELTA%s> block=%u resident grid (packed layout: 4 agents per lane, 4-byte delta rows of the move table in LDS%s%s)", Q, K,
                        scen ? ",SCEN" : "", term ? "" : ",NO_TERMINAL", bitmaps ? ",BITMAP" : "", block, bitmaps ? ", collisions through per-env occupancy bitmaps" : "",
                        scen ? ", start / goal rows from the scenario table" : "");
            if (bitmaps) {   // (Q == 8)
#define MAPF_LQ_DELTA_BITMAP(SS, TT, SCEN_PTR)                                                                                      \
                {                                                                                                                   \
                    auto kern = lq_step_kernel<8, 4, SS, TT, 3>;                                                                    \
                    if (form_lds > 32u * 1024u) { if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(160u * 1024u))) { *err = e; return true; } } \
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), form_lds, stream, args.state, args.actions, SCEN_PTR, args.slip, args.t_dev, \
                                       A | (block << 8), uint32_t(args.t), args.c.seed_lo, args.c.seed_hi, args, n_chunks);         \
                }
                if (scen && term) MAPF_LQ_DELTA_BITMAP(true, true, args.scen)
                else if (scen) MAPF_LQ_DELTA_BITMAP(true, false, args.scen)
                else if (term) MAPF_LQ_DELTA_BITMAP(false, true, no_scen)
                else MAPF_LQ_DELTA_BITMAP(false, false, no_scen)
#undef MAPF_LQ_DELTA_BITMAP
                *err = hipGetLastError();
                return true;
            }
#define MAPF_LQ_DELTA(QQ, SS, TT, SCEN_PTR)                                                                                         \
            {                                                                                                                       \
                auto kern = lq_step_kernel<QQ, 4, SS, TT, 2>;                                                                       \
                if (delta_lds > 32u * 1024u) { if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(160u * 1024u))) { *err = e; return true; } } \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(block), delta_lds, stream, args.state, args.actions, SCEN_PTR, args.slip, args.t_dev, \
                                   A | (block << 8), uint32_t(args.t), args.c.seed_lo, args.c.seed_hi, args, n_chunks);             \
            }
#define MAPF_LQ_DELTA_Q(QQ)                                                                      \
            if (Q == QQ) {                                                                        \
                if (scen && term) MAPF_LQ_DELTA(QQ, true, true, args.scen)                        \
                else if (scen) MAPF_LQ_DELTA(QQ, true, false, args.scen)                          \
                else if (term) MAPF_LQ_DELTA(QQ, false, true, no_scen)                            \
                else MAPF_LQ_DELTA(QQ, false, false, no_scen)                                     \
                *err = hipGetLastError();                                                         \
                return true;                                                                      \
            }
            MAPF_LQ_DELTA_Q(1) MAPF_LQ_DELTA_Q(2) MAPF_LQ_DELTA_Q(4) MAPF_LQ_DELTA_Q(8)
#undef MAPF_LQ_DELTA_Q
#undef MAPF_LQ_DELTA
        }
    }
    if (big) {
        const unsigned block = 1024u, n_chunks = unsigned(lanes / block), grid = n_chunks < 2u * unsigned(n_cu) ? n_chunks : 2u * unsigned(n_cu);
        note_kernel("lq_step_kernel<Q=%d,K=%d%s%s,BIG> block=1024 resident grid (packed layout: 4 agents per lane, move table in LDS%s)", Q, K,
                    scen ? ",SCEN" : "", term ? "" : ",NO_TERMINAL", scen ? ", start / goal rows from the scenario table" : "");
#define MAPF_LQ_BIG(QQ, SS, TT, SCEN_PTR)                                                                                           \
        {                                                                                                                           \
            auto kern = lq_step_kernel<QQ, 4, SS, TT, 1>;                                                                        \
            if (big_lds > 32u * 1024u) { if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(160u * 1024u - 1024u))) { *err = e; return true; } } \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), big_lds, stream, args.state, args.actions, SCEN_PTR, args.slip, args.t_dev, \
                               A | (block << 8), uint32_t(args.t), args.c.seed_lo, args.c.seed_hi, args, n_chunks);                 \
        }
#define MAPF_LQ_BIG_Q(QQ)                                                                        \
        if (Q == QQ) {                                                                            \
            if (scen && term) MAPF_LQ_BIG(QQ, true, true, args.scen)                              \
            else if (scen) MAPF_LQ_BIG(QQ, true, false, args.scen)                                \
            else if (term) MAPF_LQ_BIG(QQ, false, true, no_scen)                                  \
            else MAPF_LQ_BIG(QQ, false, false, no_scen)                                           \
            *err = hipGetLastError();                                                             \
            return true;                                                                          \
        }
        MAPF_LQ_BIG_Q(1) MAPF_LQ_BIG_Q(2) MAPF_LQ_BIG_Q(4) MAPF_LQ_BIG_Q(8)
#undef MAPF_LQ_BIG_Q
#undef MAPF_LQ_BIG
    }
    unsigned block = 256u;
    while (block > 64u && lanes < 256u * uint64_t(block)) block /= 2u;   // small batches: spread over the CUs
    // from two 256-thread blocks per CU on, four 128-thread blocks measure 3 % faster (65536 and 131072 envs of 8 agents: 3.22
    // against 3.33 us, 4.45 against 4.60; equal at 262144; at ONE block per CU -- 32768 envs -- 256 threads are 1 % ahead)
    if (lanes >= 512u * 256u) block = 128u;
    if (tune.step_block == 64u || tune.step_block == 128u || tune.step_block == 256u || tune.step_block == 512u) block = tune.step_block;
    const uint64_t per_block = block / unsigned(Q);
    if (args.n_envs == 0 || args.n_envs % per_block != 0) return false;
    const unsigned grid = unsigned(args.n_envs / per_block);
    note_kernel("lq_step_kernel<Q=%d,K=%d%s%s> block=%u (packed layout: %d agents per lane%s)", Q, K, scen ? ",SCEN" : "",
                term ? "" : ",NO_TERMINAL", block, K, scen ? ", start / goal rows from the scenario table" : "");
#define MAPF_LQ_LAUNCH(QQ, KK, SS, TT, SCEN_PTR)                                                                                   \
    hipLaunchKernelGGL((lq_step_kernel<QQ, KK, SS, TT>), dim3(grid), dim3(block), 0, stream, args.state, args.actions, SCEN_PTR,       \
                       args.slip, args.t_dev, A | (block << 8), uint32_t(args.t), args.c.seed_lo, args.c.seed_hi, args, grid)
#define MAPF_LQ_STEP(QQ, KK)                                                                                   \
    if (Q == QQ && K == KK) {                                                                                  \
        if (scen && term) MAPF_LQ_LAUNCH(QQ, KK, true, true, args.scen);                                       \
        else if (scen) MAPF_LQ_LAUNCH(QQ, KK, true, false, args.scen);                                         \
        else if (term) MAPF_LQ_LAUNCH(QQ, KK, false, true, no_scen);                                           \
        else MAPF_LQ_LAUNCH(QQ, KK, false, false, no_scen);                                                    \
        *err = hipGetLastError();                                                                              \
        return true;                                                                                           \
    }
    MAPF_LQ_STEP(1, 4) MAPF_LQ_STEP(2, 4) MAPF_LQ_STEP(4, 4) MAPF_LQ_STEP(8, 4) MAPF_LQ_STEP(16, 4)
    MAPF_LQ_STEP(2, 2) MAPF_LQ_STEP(4, 2) MAPF_LQ_STEP(8, 2) MAPF_LQ_STEP(16, 2)
#undef MAPF_LQ_STEP
#undef MAPF_LQ_LAUNCH
    return false;
}

}  // namespace mapf
