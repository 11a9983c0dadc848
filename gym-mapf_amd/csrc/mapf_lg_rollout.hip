// Lane-group family: fused T-step rollout kernel and its launcher (device code: mapf_lg.hpp).
#include "mapf_lg.hpp"

#include <cstdlib>
#include <string>

namespace mapf {

// Largest block a rollout kernel may be launched with: groups of 16 lanes unroll 8 rotation rounds and need more
// than the 128 registers a 1024-thread block leaves per lane.
template <int L> constexpr unsigned rollout_max_block() { return L == 16 ? 512u : 1024u; }

// raw (still packed) action bytes of a lane's two slots: byte 0 = agent 2g, byte 1 = agent 2g+1.  Kept packed so
// that a prefetch issued one step ahead is not forced to complete by an unpack.
template <bool EVEN>
__device__ __forceinline__ uint32_t load_actions_raw(const uint8_t *base, uint32_t row, uint32_t n_agents, uint32_t g,
                                                     bool v0, bool v1) {
    const uint8_t *p = at(base, row * n_agents + 2u * g);
    uint32_t raw = 0u;
    if (EVEN || (n_agents & 1u) == 0u) {
        if (v0) raw = *reinterpret_cast<const uint16_t *>(p);
    } else {
        uint32_t lo = 0u, hi = 0u;
        if (v0) lo = p[0];
        if (v1) hi = p[1];
        raw = lo | (hi << 8);
    }
    return raw;
}

// MV_LDS: the whole move table (V*6 entries of 16 B) is staged into LDS once per block and the two gathers of
// every step become ds_read_b128 (a random 64-lane gather through the vector-memory pipe touches up to 64 cache
// lines).  RECORD: all five trajectory arrays are written every step (the C ABI substitutes scratch for absent
// ones), STREAM: actions come from memory, else from the in-kernel policy stream.  Both are compile-time so the
// loop body has no uniform branches around its memory operations.  DENSE (A == 2L and the env count fills every
// block) additionally removes the per-lane predicates: every lane owns two real agents, the action prefetch is
// clamped instead of guarded, and the four per-env scalars are stored by ALL lanes with per-lane addresses (even
// lanes of a group write done, odd lanes collision, the last lane prob, the others reward -- duplicates carry
// identical data), so the
// loop contains no exec-masked memory operation and the compiler can wait for the prefetched action word with a
// counted vmcnt(N) instead of draining every store.  Start cells stay in two registers per lane, so an
// auto-reset touches no memory.
template <int L, bool FULL, bool MV_LDS, bool RECORD, bool STREAM, bool DENSE>
__global__ void __launch_bounds__(rollout_max_block<L>()) lg_rollout_kernel(const RolloutArgs p, const uint32_t n_agents) {
    __shared__ SlipRow slip[8];
    __shared__ OutcomeRow outcome[16];
    extern __shared__ __attribute__((aligned(16))) MoveEntry lds_mv[];
    bool live_rt;
    LaneCtx<L> x = lane_ctx<L>(n_agents, p.n_envs, live_rt);
    const bool live = DENSE || live_rt;
    if (DENSE) { x.v0 = true; x.v1 = true; }
    const uint32_t e = x.e;
    const bool leader = live && x.g == 0u;
    const bool tail = live && x.g == uint32_t(L - 1);   // holds the step's probability product

    uint32_t cur0, cur1, goal0, goal1, start0 = 0u, start1 = 0u;
    load_pair<uint16_t>(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    load_pair<uint16_t>(p.goal, p.goal_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, goal0, goal1);
    if (p.auto_reset) load_pair<uint16_t>(p.start, p.start_broadcast ? 0 : e, n_agents, x.g, x.v0, x.v1, start0, start1);
    if (MV_LDS) {   // batches of four independent loads per thread, then the four LDS writes (not load-wait-write)
        const uint32_t n_words = p.c.n_cells * kMvCols;
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
    const MoveEntry *mv = MV_LDS ? lds_mv : p.mv;

    // is_terminal is carried from step to step instead of re-deriving it from the cells every step
    uint32_t terminal = lg_is_terminal<L, FULL>(x, n_agents, cur0, cur1, goal0, goal1) ? 1u : 0u;   // an integer: no wave-mask phi
    const uint32_t start_terminal = (p.auto_reset && lg_is_terminal<L, FULL>(x, n_agents, start0, start1, goal0, goal1)) ? 1u : 0u;

    // per-env totals and the scalar trajectory arrays: their addresses are parked in VGPRs so that seven base
    // pointers do not occupy SGPRs across the step loop (it already keeps ~100 scalars live).  They stay typed as
    // GLOBAL pointers: a generic pointer would turn the stores into flat_store, which also counts on lgkmcnt and
    // would chain every LDS wait of the loop to the stores' completion.
    using gf64 = __attribute__((address_space(1))) double *;
    using gu32 = __attribute__((address_space(1))) uint32_t *;
    using gu8 = __attribute__((address_space(1))) uint8_t *;
    using gu16 = __attribute__((address_space(1))) uint16_t *;
    gf64 ret_p = (gf64)(p.out_returns ? at(p.out_returns, e) : nullptr);
    gu32 epi_p = (gu32)(p.out_episodes ? at(p.out_episodes, e) : nullptr);
    gu32 col_p = (gu32)(p.out_collisions ? at(p.out_collisions, e) : nullptr);
    gu8 done_base = (gu8)(RECORD ? p.rec_done : nullptr), coll_base = (gu8)(RECORD ? p.rec_collision : nullptr);
    gf64 reward_base = (gf64)(RECORD ? p.rec_reward : nullptr), prob_base = (gf64)(RECORD ? p.rec_prob : nullptr);
    asm volatile("" : "+v"(ret_p), "+v"(epi_p), "+v"(col_p), "+v"(done_base), "+v"(coll_base), "+v"(reward_base),
                 "+v"(prob_base));
    double ret = (p.accumulate && ret_p && leader) ? *ret_p : 0.0;
    uint32_t episodes = (p.accumulate && epi_p && leader) ? *epi_p : 0u;
    uint32_t collisions = (p.accumulate && col_p && leader) ? *col_p : 0u;
    const uint64_t env_id = p.env_id_offset + e;
    const uint32_t n_envs = uint32_t(p.n_envs);

#ifdef MAPF_STAMPS
    StampCtx st{};
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st.last = _t; }
#endif
    // Software pipeline of the loop's memory operations.  The compiler waits for the prefetched action word with
    // vmcnt(0), i.e. for EVERYTHING outstanding, so each iteration is ordered: (1) use the word loaded one
    // iteration ago, (2) only then issue the next load and the PREVIOUS step's trajectory stores, (3) compute.
    // Whatever the wait at (1) sees was issued a whole transition earlier and has long completed.
    uint32_t raw = 0u;
    if (STREAM && p.n_steps > 0) raw = load_actions_raw<FULL>(p.actions, e, n_agents, x.g, x.v0, x.v1);
    // consume the first word here, so that the wait at the loop head is the back edge's counted one
    if (DENSE) asm volatile("" : "+v"(raw));
    Words4 rng{0u, 0u, 0u, 0u}, pol{0u, 0u, 0u, 0u};   // the slip / policy words of the current four-step block
    // step s-1's results, stored during step s (DENSE: the very first store writes zeros into step 0's row, which
    // step 1 then overwrites with the real values)
    uint32_t d_next0 = 0u, d_next1 = 0u, d_flags = 0u;
    double d_reward = 0.0, d_prob = 0.0;
    // Addresses advance by one step's worth of elements per iteration (wave-uniform strides added to per-lane
    // pointers) -- no per-step row * width multiplications.
    const uint64_t step_rows = n_envs, step_cells = uint64_t(n_envs) * n_agents;
    const bool odd = (x.g & 1u) != 0u;
    const uint32_t flag_shift = (x.g & 1u) * 8u;
    const uint32_t lane_cell = e * n_agents + 2u * x.g;
    gf64 reward_lane = reward_base + e, prob_lane = prob_base + e;      // the delayed step's row
    gu8 done_lane = done_base + e, coll_lane = coll_base + e;
    gu16 rec_lane = (gu16)(RECORD ? p.rec_local : nullptr) + lane_cell;
    const bool wide_is_prob = x.g == uint32_t(L - 1);   // the probability product ends in the group's last lane
    if (DENSE && L > 1) {   // last lane writes prob, the others reward; even lanes write done, odd lanes collision
        reward_lane = wide_is_prob ? prob_lane : reward_lane;
        done_lane = odd ? coll_lane : done_lane;
    }
    asm volatile("" : "+v"(reward_lane), "+v"(prob_lane), "+v"(done_lane), "+v"(coll_lane), "+v"(rec_lane));

    auto store_record = [&]() __attribute__((always_inline)) {
        const uint32_t cells = d_next0 | (d_next1 << 16);
        if (DENSE) {
            *(gu32)rec_lane = cells;
            *reward_lane = (L > 1 && wide_is_prob) ? d_prob : d_reward;
            *done_lane = uint8_t(L > 1 ? d_flags >> flag_shift : d_flags);   // flag_shift: 8 in odd lanes
            if (L == 1) {
                *prob_lane = d_prob;
                *coll_lane = uint8_t(d_flags >> 8);
            }
        } else {
            if (FULL || (n_agents & 1u) == 0u) {
                if (x.v0) *(gu32)rec_lane = cells;
            } else {
                if (x.v0) rec_lane[0] = uint16_t(d_next0);
                if (x.v1) rec_lane[1] = uint16_t(d_next1);
            }
            if (tail) *prob_lane = d_prob;
            if (leader) {
                *reward_lane = d_reward;
                *done_lane = uint8_t(d_flags & 1u);
                *coll_lane = uint8_t(d_flags >> 8);
            }
        }
    };
    auto advance_record = [&]() __attribute__((always_inline)) {
        rec_lane += step_cells;
        reward_lane += step_rows;
        done_lane += step_rows;
        if (!(DENSE && L > 1)) { prob_lane += step_rows; coll_lane += step_rows; }
    };
    const uint8_t *act_lane = STREAM ? p.actions + lane_cell : nullptr;   // the row being prefetched

    uint32_t goal_rc0 = 0u, goal_rc1 = 0u;   // greedy policy: my agents' goal coordinates
    if (!STREAM && p.policy_cells) { goal_rc0 = p.policy_cells[goal0].x; goal_rc1 = p.policy_cells[goal1].x; }

    const uint64_t t_first = first_step_index(p);
    for (uint32_t s = 0; s < p.n_steps; ++s) {
        const uint64_t t = t_first + s;
        uint32_t act0, act1;
        if (STREAM) {
            act0 = raw & 0xFFu; act1 = (raw >> 8) & 0xFFu;   // only the low half-word of `raw` is defined
            asm volatile("" : "+v"(act0), "+v"(act1));       // (1) pins the wait for `raw` here, ahead of (2)
            if (DENSE) {                                     // clamped, not guarded: the last step re-reads its own row
                act_lane += (s + 1u < p.n_steps) ? step_cells : 0u;
                raw = *reinterpret_cast<const uint16_t *>(act_lane);
            } else if (s + 1 < p.n_steps) {
                act_lane += step_cells;
                raw = load_actions_raw<FULL>(act_lane, 0u, 0u, 0u, x.v0, x.v1);
            }
        } else if (p.policy_cells) {   // greedy policy (ghost slots read cell 0: their actions are never used)
            act0 = greedy_action(p.policy_cells, p.c.n_cells, cur0, goal_rc0);
            act1 = greedy_action(p.policy_cells, p.c.n_cells, cur1, goal_rc1);
        } else {   // policy stream: one Philox call covers agents 4q..4q+3 for the four steps of a block; this lane's
            // agents are bytes 2(g&1), 2(g&1)+1 of the step's word
            if ((t & 3u) == 0u || s == 0u) pol = policy_words(p.c, env_id, t >> 2, x.g >> 1);
            const uint32_t mine = step_word(pol, t) >> (16u * (x.g & 1u));
            act0 = policy_action_rt(mine, 0u);
            act1 = policy_action_rt(mine, 1u);
        }
        if (RECORD && (DENSE || s > 0)) {                    // (2) the previous step's outputs
            store_record();
            if (!DENSE || s > 0) advance_record();
        }
        uint32_t next0, next1;
        EnvOut o;
        STAMP(0);   // loop top: action fetch / policy / delayed stores
        // my pair's words of a four-step block (one slip-stream call per lane, traded with the neighbour lane): refresh when
        // t is a multiple of 4 (and at the first step)
        if (p.c.need_rng && ((t & 3u) == 0u || s == 0u)) rng = pair_block_words<(L > 1)>(p.c, env_id, t, x.g);
        lg_transition<L, FULL, false, true, MV_LDS, true>(p.c, mv, slip, outcome, x, n_agents, cur0, cur1, goal0, goal1, act0, act1, 0.0, 0.0,
                                            env_id, t, step_word(rng, t), terminal != 0u, next0, next1, o STAMP_ARG);
        STAMP(6);   // reward / selects
        ret = __dadd_rn(ret, o.reward);
        episodes += o.status & 0xFFu;
        collisions += (o.status >> 8) & 0xFFu;
        if (RECORD) {
            d_next0 = next0; d_next1 = next1; d_reward = o.reward; d_prob = o.prob;
            d_flags = o.status;                            // byte 0 done, byte 1 collision
        }
        const bool back = p.auto_reset && (o.status & 0xFFu) != 0u;   // MapfEnv.reset(): start cells, no reseed
        cur0 = back ? start0 : next0;
        cur1 = back ? start1 : next1;
        terminal = back ? start_terminal : (o.status >> 16);
        STAMP(7);   // reset handling
    }
    if (RECORD && p.n_steps > 0) store_record();             // flush the last step's outputs
#ifdef MAPF_STAMPS
    if (live && x.lane == 0u && epi_p) {   // diagnostic build: segment sums replace the episode counts
        for (int k = 0; k < 8; ++k) epi_p[k] = uint32_t(st.seg[k]);
        return;
    }
#endif
    if (!live) return;
    store_cells<FULL>(p.state, e, n_agents, x.g, x.v0, x.v1, cur0, cur1);
    if (leader) {
        if (ret_p) *ret_p = ret;
        if (epi_p) *epi_p = episodes;
        if (col_p) *col_p = collisions;
    }
}

// LDS budget for the move table: the CU has 160 KiB; keep room for the slip rows and the outcome table
static constexpr size_t kLdsBytes = 160 * 1024, kLdsReserve = 1024;
static_assert(sizeof(SlipRow) * 8 + sizeof(OutcomeRow) * 16 <= kLdsReserve, "static LDS of the rollout kernel");

// Defaults of the layout choices: a move table is staged into LDS while two blocks per CU still fit (a table that
// allows only one block per CU starves the SIMDs of waves); four agents per lane need one wave on every SIMD.
// ONE override: the environment variable MAPF_TUNE, "key=value,key=value,...", read here -- at handle creation, so a process
// can hold handles with different settings (the tests and the A/B tools do).  Keys (include/mapf_hip.h documents them):
//   quad_lanes, k, quad_min_lanes, oct_min_lanes, mv_lds_max_bytes, scen_table, bitmap_pairs, bitmap_block, bitmap_staycol,
//   bitmap_delta, step_big, step_block, step_delta.
// An unknown key or a malformed item is an error (*err names it): a typo must not silently measure the default.
RolloutTuning default_rollout_tuning(int device, std::string *err) {
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) n_cu = 256;
    return rollout_tuning_for(n_cu, getenv("MAPF_TUNE"), err);
}

RolloutTuning rollout_tuning_for(int n_cu, const char *text, std::string *err) {
    RolloutTuning t;
    // measured on 8 agents x 32768 envs (one wave per SIMD with four agents per lane, two with two): 517 G vs 467 G
    // agent-steps/s -- fewer, fatter waves win as long as no SIMD stays empty
    t.quad_min_lanes = uint64_t(n_cu) * 4u * 64u;        // CUs x SIMDs x lanes
    t.oct_min_lanes = uint64_t(n_cu) * 4u * 64u * 2u;    // (eight agents per lane: see try_launch_rollout_lq)
    t.mv_lds_max_bytes = (kLdsBytes - kLdsReserve) / 2;
    if (!text) return t;
    std::string items(text);
    size_t pos = 0;
    while (pos <= items.size()) {
        size_t end = items.find(',', pos);
        if (end == std::string::npos) end = items.size();
        const std::string item = items.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        char *rest = nullptr;
        const std::string key = item.substr(0, eq), val = eq == std::string::npos ? "" : item.substr(eq + 1);
        const unsigned long long v = val.empty() ? 0 : strtoull(val.c_str(), &rest, 10);
        if (eq == std::string::npos || val.empty() || (rest && *rest)) { if (err) *err = "MAPF_TUNE: malformed item '" + item + "' (want key=integer)"; return t; }
        if (key == "quad_lanes") t.quad_lanes = v != 0;
        else if (key == "k") t.force_k = int(v);
        else if (key == "quad_min_lanes") t.quad_min_lanes = v;
        else if (key == "oct_min_lanes") t.oct_min_lanes = v;
        else if (key == "mv_lds_max_bytes") t.mv_lds_max_bytes = size_t(v);
        else if (key == "scen_table") t.scen_table = v != 0;
        else if (key == "bitmap_pairs") t.bitmap_pairs = v != 0;
        else if (key == "bitmap_block") t.bitmap_block = unsigned(v);
        else if (key == "bitmap_staycol") t.bitmap_stay_column = v != 0;
        else if (key == "bitmap_delta") t.bitmap_delta_rows = v != 0;
        else if (key == "step_big") t.step_big = int(v);
        else if (key == "step_block") t.step_block = unsigned(v);
        else if (key == "step_delta") t.step_delta = int(v);
        else { if (err) *err = "MAPF_TUNE: unknown key '" + key + "'"; return t; }
    }
    return t;
}

template <int L, bool FULL, bool RECORD, bool STREAM>
static hipError_t launch_rollout_lg_impl(const RolloutArgs &args, uint32_t A, const RolloutTuning &tune, hipStream_t stream) {
    const size_t mv_bytes = size_t(args.c.n_cells) * kMvCols * sizeof(MoveEntry);
    const uint64_t threads = args.n_envs * uint64_t(L);
    if (mv_bytes + kLdsReserve <= tune.mv_lds_max_bytes && mv_bytes + kLdsReserve <= kLdsBytes && threads >= 64 * 256) {
        // block size: as many waves as can share one table copy while >= 16 waves stay resident per CU
        const size_t copies = (kLdsBytes - kLdsReserve) / (mv_bytes + sizeof(SlipRow) * 8);   // blocks per CU by LDS
        unsigned block = copies >= 4 ? 256u : (copies >= 2 ? 512u : 1024u);
        if (block > rollout_max_block<L>()) block = rollout_max_block<L>();
        const uint64_t per_block = block / unsigned(L);
        const unsigned grid = unsigned((args.n_envs + per_block - 1) / per_block);
        const bool dense = FULL && args.n_envs % per_block == 0;
        auto kern = dense ? lg_rollout_kernel<L, FULL, true, RECORD, STREAM, FULL> : lg_rollout_kernel<L, FULL, true, RECORD, STREAM, false>;
        if (mv_bytes > 32 * 1024) {
            if (hipError_t e = allow_large_lds(reinterpret_cast<const void *>(kern), int(kLdsBytes - kLdsReserve))) return e;
        }
        note_kernel("lg_rollout_kernel<L=%d,%s,MV_LDS,%s,%s,%s> block=%u (pair layout: 2 agents per lane)", L,
                    FULL ? "FULL" : "RAGGED", RECORD ? "RECORD" : "TOTALS", STREAM ? "STREAM" : "POLICY", dense ? "DENSE" : "GUARDED", block);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), mv_bytes, stream, args, A);
    } else {
        unsigned grid, block;
        lg_geometry(L, args.n_envs, grid, block);
        note_kernel("lg_rollout_kernel<L=%d,%s,MV_GLOBAL,%s,%s,%s> block=%u (pair layout: 2 agents per lane)", L,
                    FULL ? "FULL" : "RAGGED", RECORD ? "RECORD" : "TOTALS", STREAM ? "STREAM" : "POLICY",
                    (FULL && args.n_envs % (block / unsigned(L)) == 0) ? "DENSE" : "GUARDED", block);
        if (FULL && args.n_envs % (block / unsigned(L)) == 0)
            hipLaunchKernelGGL((lg_rollout_kernel<L, FULL, false, RECORD, STREAM, FULL>), dim3(grid), dim3(block), 0, stream, args, A);
        else
            hipLaunchKernelGGL((lg_rollout_kernel<L, FULL, false, RECORD, STREAM, false>), dim3(grid), dim3(block), 0, stream, args, A);
    }
    return hipGetLastError();
}

hipError_t launch_rollout_lg(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, hipStream_t stream) {
    if (args.n_envs == 0) return hipSuccess;
    const int L = lg_group_size(n_agents);
    const bool full = n_agents == 2 * L;
    const uint32_t A = uint32_t(n_agents);
    // the record variant writes all five trajectory arrays: the C ABI passes either all of them or none
    const bool record = args.rec_local != nullptr, stream_actions = args.actions != nullptr;
    if (record && !(args.rec_reward && args.rec_prob && args.rec_done && args.rec_collision)) return hipErrorInvalidValue;
    hipError_t quad_status;
    if (try_launch_rollout_lq(n_agents, args, tune, stream, &quad_status)) return quad_status;
    switch (L) {
#define X(N)                                                                                                         \
    case N:                                                                                                          \
        if (full) return record ? (stream_actions ? launch_rollout_lg_impl<N, true, true, true>(args, A, tune, stream)          \
                                                  : launch_rollout_lg_impl<N, true, true, false>(args, A, tune, stream))        \
                                : (stream_actions ? launch_rollout_lg_impl<N, true, false, true>(args, A, tune, stream)         \
                                                  : launch_rollout_lg_impl<N, true, false, false>(args, A, tune, stream));      \
        return record ? (stream_actions ? launch_rollout_lg_impl<N, false, true, true>(args, A, tune, stream)                   \
                                        : launch_rollout_lg_impl<N, false, true, false>(args, A, tune, stream))                 \
                      : (stream_actions ? launch_rollout_lg_impl<N, false, false, true>(args, A, tune, stream)                  \
                                        : launch_rollout_lg_impl<N, false, false, false>(args, A, tune, stream));
        MAPF_FOR_EACH_L(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mapf
