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
#include "mapf_kernels.hpp"
#include "mapf_device.hpp"

namespace mapf {

// EXACT: the team has exactly MAXA agents (no ghost slots, no per-slot predicates: the 9..16-agent instances would
// otherwise keep sixteen wave masks alive and spill scalar registers).
template <int MAXA, bool EXACT = false>
__global__ void __launch_bounds__(256) transitions_kernel(const TransitionsArgs p) {
    __shared__ SlipRow slip[8];
    stage_slip_table(p.slip, slip);
    const uint64_t gid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t q = gid / p.max_branches;
    const uint32_t slot = uint32_t(gid - q * p.max_branches);   // output row of the query's window
    const uint64_t b = p.first_branch + slot;                    // branch index in the query's full enumeration
    if (q >= p.n_queries) return;
    const uint32_t A = EXACT ? uint32_t(MAXA) : p.n_agents;
    const uint64_t env = p.env_index ? p.env_index[q] : 0;
    const uint16_t *goal_row = p.goal + (p.goal_broadcast ? 0 : env * A);
    const uint16_t *state_row = p.local + q * A;
    const uint8_t *act_row = p.actions + q * A;

    uint32_t prev[MAXA], goal[MAXA], act[MAXA], n[MAXA];
    MoveEntry entry[MAXA];
    uint32_t dup_acc = 0xFFFFFFFFu, goal_acc = 0u;
    uint64_t count = 1;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        const bool on = EXACT || uint32_t(i) < A;
        prev[i] = on ? state_row[i] : 0x10000u + uint32_t(i);       // ghosts: unique, never equal to a real cell
        goal[i] = on ? goal_row[i] : prev[i];
        const uint32_t a = on ? act_row[i] : 0u;
        act[i] = a > 4u ? 0u : a;
        entry[i] = on ? move_entry(p.mv, p.c.n_cells, prev[i], act[i]) : MoveEntry{0u, 0u, 0u, 0u};
        n[i] = on ? slip[entry_code(entry[i])].n : 1u;
        count *= n[i];
        goal_acc |= prev[i] ^ goal[i];
    }
#pragma unroll
    for (int i = 0; i < MAXA; ++i)
#pragma unroll
        for (int j = i + 1; j < MAXA; ++j) dup_acc = min(dup_acc, prev[i] ^ prev[j]);
    const bool terminal = dup_acc == 0u || goal_acc == 0u;           // is_terminal: mapf_env.py:210-223
    if (terminal) count = 1;
    if (slot == 0 && p.out_count) p.out_count[q] = uint32_t(count > 0xFFFFFFFFull ? 0xFFFFFFFFull : count);
    if (b >= count) return;

    const uint64_t o = q * p.max_branches + slot;
    if (terminal) {
        if (p.out_next) for (uint32_t i = 0; i < A; ++i) p.out_next[o * A + i] = uint16_t(prev[i]);
        if (p.out_prob) p.out_prob[o] = 1.0;
        if (p.out_reward) p.out_reward[o] = 0.0;
        if (p.out_done) p.out_done[o] = 1;
        if (p.out_collision) p.out_collision[o] = 0;
        return;
    }
    // digits of b, last agent fastest
    uint32_t k[MAXA];
    uint64_t rest = b;
#pragma unroll
    for (int i = MAXA - 1; i >= 0; --i) { k[i] = uint32_t(rest % n[i]); rest /= n[i]; }

    uint32_t next[MAXA];
    double prob = 1.0;
    int stayed = 0;
    uint32_t coll_acc = 0xFFFFFFFFu, goal_next_acc = 0u;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        const bool on = EXACT || uint32_t(i) < A;
        const SlipRow &row = slip[entry_code(entry[i])];
        const uint32_t cell = entry_cell(entry[i], k[i]);
        next[i] = on ? cell : prev[i];
        const double qv = on ? row.q[k[i]] : 1.0;
        prob = (i == 0) ? qv : __dmul_rn(prob, qv);
        stayed += (on && prev[i] == goal[i] && act[i] == 0u) ? 1 : 0;
        goal_next_acc |= next[i] ^ goal[i];
    }
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
    double living = p.c.r_living;
    if (p.c.criteria == 1u) living = __dmul_rn(double(int(A) - stayed), p.c.r_living);
    const double reward = coll ? __dadd_rn(p.c.r_clash, living) : (goal_next ? __dadd_rn(p.c.r_goal, living) : living);
    if (p.out_next) for (uint32_t i = 0; i < A; ++i) p.out_next[o * A + i] = uint16_t(next[i]);
    if (p.out_prob) p.out_prob[o] = prob;
    if (p.out_reward) p.out_reward[o] = reward;
    if (p.out_done) p.out_done[o] = (coll || goal_next) ? 1 : 0;
    if (p.out_collision) p.out_collision[o] = coll ? 1 : 0;
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
    const uint64_t threads = args.n_queries * uint64_t(args.max_branches);
    if (threads == 0) return hipSuccess;
    const uint64_t grid64 = (threads + 255) / 256;
    if (grid64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid{unsigned(grid64)}, block{256};
    if (args.n_agents <= 4) hipLaunchKernelGGL(transitions_kernel<4>, grid, block, 0, stream, args);
    else if (args.n_agents <= 8) hipLaunchKernelGGL(transitions_kernel<8>, grid, block, 0, stream, args);
    else switch (args.n_agents) {
#define X(N) case N: hipLaunchKernelGGL((transitions_kernel<N, true>), grid, block, 0, stream, args); break;
        X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mapf
