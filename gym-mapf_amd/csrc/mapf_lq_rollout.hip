// Quad-lane rollout kernel: FOUR agents per lane (two packed cell pairs), Q = A/4 lanes per env.
//
// Why a second lane layout: in the pair layout (mapf_lg.hpp, two agents per lane) everything that is per ENV --
// flag reduction, outcome lookup, totals, reset handling, the hand-over steps of the probability product -- is
// replicated over the L = A/2 lanes of a group, and that part is about 40 % of a step's vector instructions at
// A = 8.  Four agents per lane halve the lanes per env, so the replicated part halves, the in-lane half of the
// pair tests needs no cross-lane move at all, and each lane carries two independent Philox calls / four
// independent table gathers (more instruction-level parallelism per wave, which matters because the same env
// count now fills only half as many waves).
//
// Scope: the fused rollout of FULL groups only (A = 4Q, Q in {1, 2, 4, 8, 16}), every block full, move table in
// LDS -- the bench configuration and its neighbours.  Everything else (odd agent counts, ragged batches, tables
// beyond the LDS budget, single steps) stays with the pair layout; launch_rollout_lg() picks.  Same stream,
// same arithmetic, same outputs: the parity tests run both layouts against the oracle.
#include "mapf_lg.hpp"

#include <cstdlib>

namespace mapf {

namespace {

constexpr size_t kLdsBytes = 160 * 1024, kLdsReserve = 1024;
static_assert(sizeof(SlipRow) * 8 + sizeof(OutcomeRow) * 16 <= kLdsReserve, "static LDS of the rollout kernel");

using gf64 = __attribute__((address_space(1))) double *;
using gu32 = __attribute__((address_space(1))) uint32_t *;
using gu8 = __attribute__((address_space(1))) uint8_t *;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
using gu32x2 = __attribute__((address_space(1))) u32x2 *;

// sampled list slot of one agent from the top 16 bits of its uniform (see slip_move_hi): idx, its probability, and
// the tie distance (0 <=> hi equals a threshold -> exact path)
__device__ __forceinline__ uint32_t sample_slot(const SlipRow *lds_slip, const MoveEntry &entry, uint32_t hi, double &q,
                                                uint32_t &tie_dist) {
    // (slot = how many of the first two thresholds hi has passed: see slip_move_hi in mapf_device.hpp)
    const uint32_t t0 = entry.z & 0xFFFFu, t1 = entry.z >> 16;
    const uint32_t d0 = hi - t0, d1 = hi - t1, d2 = hi - 0xFFFFu;
    tie_dist = min(d0, min(d1, d2));
    const uint32_t idx = 2u - (d0 >> 31) - (d1 >> 31);
    q = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lds_slip) + entry_row_offset(entry) + idx * 8u);
    return idx;
}
// list slot idx of an entry, zero-extended: one v_perm_b32
__device__ __forceinline__ uint32_t cell_lo(const MoveEntry &entry, uint32_t idx) {
    return __builtin_amdgcn_perm(entry.y, entry.x, 0x0C0C0100u + idx * 0x0202u);
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

// full rotations 1 .. Q/2-1: my two pairs against both pairs of group position g + S
template <int Q, int S, bool DUP, bool MOVES>
struct QuadRounds {
    static __device__ __forceinline__ void run(const LaneCtx<Q> &x, uint32_t ca, uint32_t cb, uint32_t na, uint32_t nb,
                                               PairAcc<true> &acc) {
        if constexpr (S <= Q / 2 - 1) {
            const uint32_t oa_c = group_rot<Q, S>(ca, x), ob_c = group_rot<Q, S>(cb, x);
            const uint32_t oa_n = MOVES ? group_rot<Q, S>(na, x) : 0u, ob_n = MOVES ? group_rot<Q, S>(nb, x) : 0u;
            pair_apply_packed<DUP, MOVES>(ca, na, oa_c, oa_n, acc);
            pair_apply_packed<DUP, MOVES>(ca, na, ob_c, ob_n, acc);
            pair_apply_packed<DUP, MOVES>(cb, nb, oa_c, oa_n, acc);
            pair_apply_packed<DUP, MOVES>(cb, nb, ob_c, ob_n, acc);
            QuadRounds<Q, S + 1, DUP, MOVES>::run(x, ca, cb, na, nb, acc);
        }
    }
};

// all agent pairs of the env.  ca/cb: my packed current cells (agents 4g,4g+1 / 4g+2,4g+3), na/nb: next cells.
template <int Q, bool DUP, bool MOVES>
__device__ __forceinline__ PairAcc<true> quad_pair_tests(const LaneCtx<Q> &x, uint32_t ca, uint32_t cb, uint32_t na,
                                                         uint32_t nb) {
    PairAcc<true> acc;
    const uint32_t ca_sw = swap_halves(ca), cb_sw = swap_halves(cb);
    const uint32_t na_sw = MOVES ? swap_halves(na) : 0u, nb_sw = MOVES ? swap_halves(nb) : 0u;
    // inside each pair (both half-words carry the same test), then pair A against pair B
    if (DUP) acc.dup = pk_min_u16(ca ^ ca_sw, cb ^ cb_sw);
    if (MOVES) {
        acc.vertex = pk_min_u16(na ^ na_sw, nb ^ nb_sw);
        acc.swap = pk_min_u16((na ^ ca_sw) | (ca ^ na_sw), (nb ^ cb_sw) | (cb ^ nb_sw));
    }
    pair_apply_packed<DUP, MOVES>(ca, na, cb, nb, acc);
    if constexpr (Q >= 2) {
        QuadRounds<Q, 1, DUP, MOVES>::run(x, ca, cb, na, nb, acc);
        // half rotation: lane g meets lane g + Q/2 from both sides, so the two lanes split the 16 agent pairs --
        // lower-half lanes offer their pairs half-swapped; whoever receives runs only the same-half-word tests
        const bool lower = x.g < uint32_t(Q / 2);
        const uint32_t oa_c = group_rot<Q, (Q + 1) / 2>(lower ? ca_sw : ca, x), ob_c = group_rot<Q, (Q + 1) / 2>(lower ? cb_sw : cb, x);
        const uint32_t oa_n = MOVES ? group_rot<Q, (Q + 1) / 2>(lower ? na_sw : na, x) : 0u;
        const uint32_t ob_n = MOVES ? group_rot<Q, (Q + 1) / 2>(lower ? nb_sw : nb, x) : 0u;
        pair_apply_same<DUP, MOVES>(ca, na, oa_c, oa_n, acc);
        pair_apply_same<DUP, MOVES>(ca, na, ob_c, ob_n, acc);
        pair_apply_same<DUP, MOVES>(cb, nb, oa_c, oa_n, acc);
        pair_apply_same<DUP, MOVES>(cb, nb, ob_c, ob_n, acc);
    }
    return acc;
}

// MapfEnv.is_terminal (mapf_env.py:210-223) of the group's env, in every lane
template <int Q>
__device__ __forceinline__ bool quad_is_terminal(const LaneCtx<Q> &x, uint32_t ca, uint32_t cb, uint32_t ga, uint32_t gb) {
    const PairAcc<true> acc = quad_pair_tests<Q, true, false>(x, ca, cb, 0u, 0u);
    const bool off_goal = ca != ga || cb != gb;
    const uint32_t flags = group_reduce<Q, false>((PairAcc<true>::hit(acc.dup) ? 1u : 0u) | (off_goal ? 2u : 0u), x);
    return (flags & 1u) != 0u || (flags & 2u) == 0u;
}

// ordered product over agents 0..A-1: every lane continues the product handed over by the lane before it (see
// prob_product in mapf_lg.hpp); the total ends in lane Q-1
template <int Q>
__device__ __forceinline__ double quad_prob_product(double q0, double q1, double q2, double q3) {
    double run = __dmul_rn(__dmul_rn(__dmul_rn(q0, q1), q2), q3);
#pragma unroll
    for (int k = 1; k < Q; ++k) {
        const uint32_t lo = from_prev_lane<Q>(uint32_t(__double2loint(run)));
        const uint32_t hi = from_prev_lane<Q>(uint32_t(__double2hiint(run)));
        run = __dmul_rn(__dmul_rn(__dmul_rn(__dmul_rn(__hiloint2double(int(hi), int(lo)), q0), q1), q2), q3);
    }
    return run;
}

// RECORD: all five trajectory arrays are written every step; STREAM: actions come from memory, else from the
// in-kernel policy stream.  Loop structure, software pipeline and store scheme as lg_rollout_kernel<DENSE>.
template <int Q, bool RECORD, bool STREAM>
__global__ void __launch_bounds__(512) lq_rollout_kernel(const RolloutArgs p, const uint32_t n_agents) {
    __shared__ SlipRow slip[8];
    __shared__ OutcomeRow outcome[16];
    extern __shared__ __attribute__((aligned(16))) MoveEntry lds_mv[];
    LaneCtx<Q> x;
    x.lane = threadIdx.x & 63u;
    x.g = x.lane & uint32_t(Q - 1);
    x.base = x.lane & ~uint32_t(Q - 1);
    x.e = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * uint32_t(64 / Q) + x.lane / uint32_t(Q);
    x.v0 = x.v1 = true;
    const uint32_t e = x.e;
    const uint32_t lane_cell = e * n_agents + 4u * x.g;             // my first agent's element index
    const uint32_t fixed_cell = 4u * x.g;                           // ... in a broadcast row

    const u32x2 cells = *reinterpret_cast<const u32x2 *>(at(p.state, lane_cell));
    const u32x2 gl = *reinterpret_cast<const u32x2 *>(at(p.goal, p.goal_broadcast ? fixed_cell : lane_cell));
    u32x2 sc = {0u, 0u};
    if (p.auto_reset) sc = *reinterpret_cast<const u32x2 *>(at(p.start, p.start_broadcast ? fixed_cell : lane_cell));
    uint32_t ca = cells.x, cb = cells.y;                                   // packed current cells
    const uint32_t ga = gl.x, gb = gl.y, sa = sc.x, sb = sc.y;
    {   // move table -> LDS, batches of four independent loads per thread
        const uint32_t n_words = p.c.n_cells * 5u;
        for (uint32_t w0 = threadIdx.x; w0 < n_words; w0 += 4u * blockDim.x) {
            MoveEntry part[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t w = w0 + k * blockDim.x;
                part[k] = p.mv[w < n_words ? w : n_words - 1u];
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t w = w0 + k * blockDim.x;
                if (w < n_words) lds_mv[w] = part[k];
            }
        }
    }
    stage_outcome_table(p.c, outcome);
    stage_slip_table(p.slip, slip);   // ends with __syncthreads()

    uint32_t terminal = quad_is_terminal<Q>(x, ca, cb, ga, gb) ? 1u : 0u;
    const uint32_t start_terminal = (p.auto_reset && quad_is_terminal<Q>(x, sa, sb, ga, gb)) ? 1u : 0u;

    const bool leader = x.g == 0u, tail = x.g == uint32_t(Q - 1);
    gf64 ret_p = (gf64)(p.out_returns ? at(p.out_returns, e) : nullptr);
    gu32 epi_p = (gu32)(p.out_episodes ? at(p.out_episodes, e) : nullptr);
    gu32 col_p = (gu32)(p.out_collisions ? at(p.out_collisions, e) : nullptr);
    asm volatile("" : "+v"(ret_p), "+v"(epi_p), "+v"(col_p));
    double ret = (p.accumulate && ret_p && leader) ? *ret_p : 0.0;
    uint32_t episodes = (p.accumulate && epi_p && leader) ? *epi_p : 0u;
    uint32_t collisions = (p.accumulate && col_p && leader) ? *col_p : 0u;
    const uint64_t env_id = p.env_id_offset + e;
    const uint32_t n_envs = uint32_t(p.n_envs);

    // per-lane pointers into the step rows; they advance by wave-uniform strides
    const uint64_t step_rows = n_envs, step_cells = uint64_t(n_envs) * n_agents;
    const bool odd = (x.g & 1u) != 0u;
    const uint32_t flag_shift = (x.g & 1u) * 8u;
    gf64 wide_lane = nullptr, prob_lane = nullptr;
    gu8 narrow_lane = nullptr, coll_lane = nullptr;
    gu32x2 rec_lane = nullptr;
    if (RECORD) {
        gf64 reward_lane = (gf64)p.rec_reward + e;
        prob_lane = (gf64)p.rec_prob + e;
        gu8 done_lane = (gu8)p.rec_done + e;
        coll_lane = (gu8)p.rec_collision + e;
        // Q >= 2: the last lane writes prob, the others reward; even lanes write done, odd lanes collision
        wide_lane = (Q > 1 && tail) ? prob_lane : reward_lane;
        narrow_lane = (Q > 1 && odd) ? coll_lane : done_lane;
        rec_lane = (gu32x2)((__attribute__((address_space(1))) uint16_t *)p.rec_local + lane_cell);
    }
    asm volatile("" : "+v"(wide_lane), "+v"(prob_lane), "+v"(narrow_lane), "+v"(coll_lane), "+v"(rec_lane));
    const uint8_t *act_lane = STREAM ? p.actions + lane_cell : nullptr;

    // Action words are fetched TWO steps ahead (one step is shorter than a loaded HBM round trip): two registers take
    // turns -- even steps use raw_even, odd steps raw_odd, each reloading its own register for two steps later --
    // so the loop is unrolled by two (rotating one register through a move would be a use, i.e. a wait).
    const uint32_t last_row = p.n_steps ? p.n_steps - 1u : 0u;
    uint32_t raw_even = 0u, raw_odd = 0u;
    if (STREAM && p.n_steps > 0) {
        raw_even = *reinterpret_cast<const uint32_t *>(act_lane);
        act_lane += last_row >= 1u ? step_cells : 0u;             // clamped, not guarded: late rows are re-read
        raw_odd = *reinterpret_cast<const uint32_t *>(act_lane);
    }
    asm volatile("" : "+v"(raw_even), "+v"(raw_odd));   // consumed here: the loop's waits are the back edge's counted ones
    Words4 rng_a{0u, 0u, 0u, 0u}, rng_b{0u, 0u, 0u, 0u};
    // "Pending" = what is left of step s-1 when step s begins: its probability chain, its totals and its trajectory
    // stores.  They are finished at the top of step s, right after step s's table reads have been issued, so the
    // chain of dependent float64 multiplies runs while those reads are in flight, and the outcome row / probability
    // reads of step s-1 (requested in step s-1, consumed only here) never stall anything.  Step 0 finishes a dummy:
    // reward -0.0 leaves the running return unchanged bit for bit, the stores hit row 0 and are overwritten by step 1.
    double pq0 = 0.0, pq1 = 0.0, pq2 = 0.0, pq3 = 0.0, p_reward = -0.0;
    uint32_t p_a = 0u, p_b = 0u, p_status = 0u;

#ifdef MAPF_STAMPS
    StampCtx st{};
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st.last = _t; }
#endif
    auto finish_pending = [&]() __attribute__((always_inline)) {
        // opaque from here on: otherwise the optimiser moves these consumers back to where the values are produced
        // (the end of the previous step), which is exactly the stall this pipeline removes
        asm volatile("" : "+v"(p_reward), "+v"(p_status));
        if (RECORD) asm volatile("" : "+v"(pq0), "+v"(pq1), "+v"(pq2), "+v"(pq3));
        ret = __dadd_rn(ret, p_reward);
        episodes += p_status & 0xFFu;                          // byte 0 done, byte 1 collision
        collisions += (p_status >> 8) & 0xFFu;
        if (RECORD) {
            const double prob = quad_prob_product<Q>(pq0, pq1, pq2, pq3);   // total in the last lane
            *rec_lane = u32x2{p_a, p_b};
            *wide_lane = (Q > 1 && tail) ? prob : p_reward;
            *narrow_lane = uint8_t(Q > 1 ? p_status >> flag_shift : p_status);
            if (Q == 1) {
                *prob_lane = prob;
                *coll_lane = uint8_t(p_status >> 8);
            }
        }
    };

    uint32_t goal_rc[4] = {0u, 0u, 0u, 0u};   // greedy policy: my agents' goal coordinates
    if (!STREAM && p.policy_cells) {
        const uint32_t goal_cell[4] = {ga & 0xFFFFu, ga >> 16, gb & 0xFFFFu, gb >> 16};
#pragma unroll
        for (int k = 0; k < 4; ++k) goal_rc[k] = p.policy_cells[goal_cell[k]].x;
    }

    auto one_step = [&](const uint32_t s, uint32_t &raw) __attribute__((always_inline)) {
        const uint64_t t = p.t + s;
        uint32_t act[4];
        if (STREAM) {
            act[0] = raw & 0xFFu; act[1] = (raw >> 8) & 0xFFu; act[2] = (raw >> 16) & 0xFFu; act[3] = raw >> 24;
            asm volatile("" : "+v"(act[0]), "+v"(act[1]), "+v"(act[2]), "+v"(act[3]));   // the wait for `raw` sits here
            act_lane += (s + 2u <= last_row) ? step_cells : 0u;   // row min(s + 2, last)
            raw = *reinterpret_cast<const uint32_t *>(act_lane);
        } else if (p.policy_cells) {   // greedy policy
            const uint32_t at_cell[4] = {ca & 0xFFFFu, ca >> 16, cb & 0xFFFFu, cb >> 16};
#pragma unroll
            for (int k = 0; k < 4; ++k) act[k] = greedy_action(p.policy_cells, p.c.n_cells, at_cell[k], goal_rc[k]);
        } else {   // policy stream: one Philox call covers exactly my agents 4g .. 4g+3
            uint32_t w[4];
            const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | (x.g << 24);
            philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3, p.c.pol_lo, p.c.pol_hi, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) act[k] = __umulhi(w[k], 5u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) act[k] = act[k] > 4u ? 0u : act[k];

        // --- my four agents' table rows: requested first ...
        const uint32_t cur[4] = {ca & 0xFFFFu, ca >> 16, cb & 0xFFFFu, cb >> 16};
        MoveEntry entry[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) entry[k] = move_entry<false>(lds_mv, p.c.n_cells, cur[k], act[k]);
        STAMP(0);   // loop top: action fetch / policy, table read issue
        // --- ... then the previous step is finished while they are in flight
        finish_pending();
        if (RECORD && s > 0) {
            rec_lane = (gu32x2)((__attribute__((address_space(1))) uint16_t *)rec_lane + step_cells);
            wide_lane += step_rows;
            narrow_lane += step_rows;
            if (Q == 1) { prob_lane += step_rows; coll_lane += step_rows; }
        }
        STAMP(1);   // previous step: probability chain, totals, trajectory stores
        // one slip-stream call per pair serves four steps: refresh when t is a multiple of 4 (and at the first step)
        if (p.c.need_rng && ((t & 3u) == 0u || s == 0u)) {
            slip_words_x2(p.c, env_id, t >> 2, 2u * x.g, 2u * x.g + 1u, rng_a, rng_b);
        }
        STAMP(2);   // slip Philox (1 step in 4)
        const uint32_t word_a = step_word(rng_a, t), word_b = step_word(rng_b, t);
        const uint32_t hi[4] = {word_a & 0xFFFFu, word_a >> 16, word_b & 0xFFFFu, word_b >> 16};
        double q[4];
        uint32_t idx[4], tie[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) idx[k] = sample_slot(slip, entry[k], hi[k], q[k], tie[k]);
        uint32_t na = cell_lo(entry[0], idx[0]) | (cell_lo(entry[1], idx[1]) << 16);   // one v_lshl_or_b32 per pair
        uint32_t nb = cell_lo(entry[2], idx[2]) | (cell_lo(entry[3], idx[3]) << 16);
        if (__builtin_expect(__any(min(min(tie[0], tie[1]), min(tie[2], tie[3])) == 0u && p.c.need_rng), 0)) {
            // a top-16-bit tie somewhere in the wave: redo with all 53 bits
            uint32_t nx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                slip_move<false>(slip, entry[k], refine_mantissa(p.c, env_id, t, 4u * x.g + uint32_t(k), hi[k]), 0.0, nx[k], q[k]);
            na = nx[0] | (nx[1] << 16);
            nb = nx[2] | (nx[3] << 16);
        }
        STAMP(3);   // sampling (table wait, thresholds, probability read issue)

        // --- pair tests, per-env facts
        const PairAcc<true> acc = quad_pair_tests<Q, false, true>(x, ca, cb, na, nb);
        STAMP(4);   // pair tests
        const bool off_goal_next = na != ga || nb != gb;
        uint32_t flags = (PairAcc<true>::hit(acc.vertex) ? 1u : 0u) | (PairAcc<true>::hit(acc.swap) ? 2u : 0u) | (off_goal_next ? 4u : 0u);
        flags = group_reduce<Q, false>(flags, x);
        const uint32_t f = flags & 7u;
        STAMP(5);   // flags + group reduce

        // --- outcome: the row (status for both criteria, reward for Makespan) is only REQUESTED here; everything the
        // next step's table address depends on is re-derived from f without waiting for it
        const bool was_terminal = terminal != 0u;
        const OutcomeRow *row = &outcome[f | (terminal << 3)];   // terminal is 0 / 1
        const uint32_t row_status = row->status;
        const double row_reward = row->reward;                 // two plain reads, both unconditional
        double soc_reward = 0.0;
        if (p.c.criteria != 0u) {
            // _living_reward: mapf_env.py:436-446
            const uint32_t goal[4] = {ga & 0xFFFFu, ga >> 16, gb & 0xFFFFu, gb >> 16};
            uint32_t mine = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) mine += (cur[k] == goal[k] && act[k] == 0u) ? 1u : 0u;
            const int stayed = int(group_reduce<Q, true>(mine, x));
            const double living = __dmul_rn(double(int(n_agents) - stayed), p.c.r_living);
            const bool coll = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
            const double r = coll ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
            soc_reward = was_terminal ? 0.0 : r;
        }
        const double reward = p.c.criteria != 0u ? soc_reward : row_reward;
        if (was_terminal) { na = ca; nb = cb; }                // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0})
        p_reward = reward;
        p_status = row_status;
        p_a = na; p_b = nb;
        if (RECORD) {                                          // a zero factor makes the whole product +0.0
            pq0 = was_terminal ? 0.0 : q[0];
            pq1 = q[1]; pq2 = q[2]; pq3 = q[3];
        }
        STAMP(6);   // outcome request, SoC living reward
        // MapfEnv.reset(): start cells, no reseed.  Every f except "off goal, no collision" ends the episode; the
        // returned state is terminal after a vertex collision or on goal (mapf_env.py:210-223), a swap alone is not
        const bool done = f != 4u || was_terminal;
        const bool next_terminal = ((f ^ 4u) & 5u) != 0u || was_terminal;
        const bool back = p.auto_reset && done;
        ca = back ? sa : na;
        cb = back ? sb : nb;
        terminal = back ? start_terminal : (next_terminal ? 1u : 0u);
        STAMP(7);   // reset handling
    };
    uint32_t s = 0;
    for (; s + 1u < p.n_steps; s += 2u) {
        one_step(s, raw_even);
        one_step(s + 1u, raw_odd);
    }
    if (s < p.n_steps) one_step(s, raw_even);
    if (p.n_steps > 0) finish_pending();                       // the last step's chain, totals and stores
#ifdef MAPF_STAMPS
    if (x.lane == 0u && epi_p) {   // diagnostic build: segment sums replace the episode counts
        for (int k = 0; k < 8; ++k) epi_p[k] = uint32_t(st.seg[k]);
        return;
    }
#endif
    *reinterpret_cast<u32x2 *>(at(p.state, lane_cell)) = u32x2{ca, cb};
    if (leader) {
        if (ret_p) *ret_p = ret;
        if (epi_p) *epi_p = episodes;
        if (col_p) *col_p = collisions;
    }
}

template <int Q, bool RECORD, bool STREAM>
hipError_t launch_impl(const RolloutArgs &args, uint32_t A, unsigned block, size_t mv_bytes, hipStream_t stream) {
    auto kern = lq_rollout_kernel<Q, RECORD, STREAM>;
    if (mv_bytes > 32 * 1024) {   // dynamic LDS beyond the default cap needs an explicit opt-in (per device: not cached)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           int(kLdsBytes - kLdsReserve));
        if (e != hipSuccess) return e;
    }
    const unsigned grid = unsigned(args.n_envs / (block / unsigned(Q)));
    note_kernel("lq_rollout_kernel<Q=%d,%s,%s> block=%u (quad layout: 4 agents per lane)", Q, RECORD ? "RECORD" : "TOTALS",
                STREAM ? "STREAM" : "POLICY", block);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), mv_bytes, stream, args, A);
    return hipGetLastError();
}

}  // namespace

// true when the quad layout took the launch (*err = its status); false = not applicable, use the pair layout
bool try_launch_rollout_lq(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err) {
    if (!tune.quad_lanes || n_agents < 4 || n_agents % 4 != 0) return false;
    // Half as many lanes per env means half as many waves: the quad layout only pays while it still keeps two waves on
    // every SIMD (measured: 65536 envs x 8 agents = 2 waves/SIMD -> 460 G vs 413 G agent-steps/s for the pair layout;
    // 32768 envs = 1 wave/SIMD -> 233 G vs 300 G): tune.quad_min_lanes.
    const uint64_t min_lanes = tune.quad_min_lanes;
    const size_t mv_lds_limit = tune.mv_lds_max_bytes;
    const int Q = n_agents / 4;
    if (Q > 16 || (Q & (Q - 1)) != 0) return false;
    const size_t mv_bytes = size_t(args.c.n_cells) * 5 * sizeof(MoveEntry);
    if (mv_bytes + kLdsReserve > mv_lds_limit) return false;
    const size_t copies = (kLdsBytes - kLdsReserve) / (mv_bytes + kLdsReserve);   // blocks per CU by LDS
    const unsigned block = copies >= 4 ? 256u : 512u;
    const uint64_t per_block = block / unsigned(Q);
    const uint64_t lanes = args.n_envs * uint64_t(Q);
    if (args.n_envs % per_block != 0 || lanes < 64 * 256 || lanes < min_lanes) return false;
    const bool record = args.rec_local != nullptr, stream_actions = args.actions != nullptr;
    if (record && !(args.rec_reward && args.rec_prob && args.rec_done && args.rec_collision)) {
        *err = hipErrorInvalidValue;
        return true;
    }
    const uint32_t A = uint32_t(n_agents);
    switch (Q) {
#define X(N)                                                                                                         \
    case N:                                                                                                          \
        *err = record ? (stream_actions ? launch_impl<N, true, true>(args, A, block, mv_bytes, stream)                    \
                                        : launch_impl<N, true, false>(args, A, block, mv_bytes, stream))                  \
                      : (stream_actions ? launch_impl<N, false, true>(args, A, block, mv_bytes, stream)                   \
                                        : launch_impl<N, false, false>(args, A, block, mv_bytes, stream));                \
        return true;
        X(1) X(2) X(4) X(8) X(16)
#undef X
        default: return false;
    }
}

}  // namespace mapf
