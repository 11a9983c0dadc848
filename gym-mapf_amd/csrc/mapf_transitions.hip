// env.P[s][a] on the device: every branch of the joint slip distribution of a (state, joint action) query
// (reference mapf_env.py:448-478 `_get_transitions`, the enumeration planners iterate over).
//
// One thread per (query, branch).  Branch b of a query selects entry k_i of agent i's merged movement list
// (single_agent_movements, :163-184) in itertools.product order -- agent 0 varies slowest -- so
// k_i = digit i of b in the mixed radix (n_0, ..., n_{A-1}).  prob is the left-to-right float64 product of the
// selected probabilities (functools.reduce at :467); reward / done / collision come from the same rules as
// step() (:225-235).  A terminal state has the single branch ((1.0, False), s, 0, True) (:455-456).
// A call returns the WINDOW [first_branch, first_branch + max_branches) of every query's enumeration, so a caller
// pages through the 3^A branches of a large team in pieces (A <= 16: at most 43 M branches per query).
#include <algorithm>
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"

namespace mapf {

// one branch's next cells, out[row * A .. row * A + A): the widest stores the row's alignment allows when the team
// fills the instance (rows of a full even team are 4-byte aligned, of a multiple of four 8-byte aligned, given an
// equally aligned array), two bytes at a time otherwise
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

// EXACT: the team has exactly MAXA agents (no ghost slots, no per-slot predicates: the 9..16-agent instances would
// otherwise keep sixteen wave masks alive and spill scalar registers).
//
// Work split: a GROUP of `lanes` (a power of two <= 64) consecutive lanes owns one chunk of `chunk` window rows of one
// query and walks it `lanes` rows at a time.  Everything that depends on the query only -- cells, goals, actions, the
// agents' table rows, list lengths, is_terminal(prev), the SoC living reward -- is set up ONCE per lane and reused for
// every branch the lane emits; lanes whose rows lie beyond the query's branch count leave after that set-up.  (One
// thread per reserved row, the first form of this kernel, spent most of its time setting up threads whose row did not
// exist: a query reserves 3^A rows and uses a third of them on a room map.)  A group's lanes write consecutive rows.
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
    const uint64_t row0 = q * p.max_branches;
    if (terminal) {                                              // the single branch ((1.0, False), s, 0, True)
        if (piece == 0u && lane == 0u && p.first_branch == 0u) {
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
        // select between rest, rest >> 1 and a multiply-high by the reciprocal of 3 (a 64-bit divide per agent costs
        // more than the rest of the branch together)
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
                // (no wave-mask booleans: those cost an SGPR pair per pair test); ghost cells are >= 0x10000 and unique,
                // so they never produce a zero.
                const uint32_t swap = (prev[i] ^ next[j]) | (prev[j] ^ next[i]);
                coll_acc = min(coll_acc, min(next[i] ^ next[j], swap));
            }
        }
        const bool coll = coll_acc == 0u, goal_next = goal_next_acc == 0u;
        const uint64_t o = row0 + slot;
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

hipError_t launch_transitions(const TransitionsArgs &args, hipStream_t stream) {
    if (args.n_queries == 0 || args.max_branches == 0) return hipSuccess;
    // lanes per group: the window if it is shorter than a wave, else a wave; a group walks up to 16 x lanes rows, a long
    // window is cut into that many-row chunks (so that a single query of a large team still fills the device)
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
    const int inst = args.n_agents <= 4 ? 4 : (args.n_agents <= 8 ? 8 : int(args.n_agents));
    note_kernel("transitions_kernel<%d%s> %u agents, %u lanes x %u rows per group, %u groups per query", inst, inst > 8 ? ",EXACT" : "",
                args.n_agents, lanes, walks, chunks_per_query);
    if (args.n_agents <= 4) hipLaunchKernelGGL(transitions_kernel<4>, grid, block, 0, stream, args, lanes_log2, chunk, chunks_per_query);
    else if (args.n_agents <= 8) hipLaunchKernelGGL(transitions_kernel<8>, grid, block, 0, stream, args, lanes_log2, chunk, chunks_per_query);
    else switch (args.n_agents) {
#define X(N) case N: hipLaunchKernelGGL((transitions_kernel<N, true>), grid, block, 0, stream, args, lanes_log2, chunk, chunks_per_query); break;
        X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mapf
