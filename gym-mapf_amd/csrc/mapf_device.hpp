// Device-side building blocks of the batched MapfEnv.step() path (gfx950 only).
//
// One thread owns one env for a whole transition: its A cells, actions and the
// A(A-1)/2 pair tests live in registers, so the only memory traffic is the
// env-major rows themselves (16 B per lane at A = 8 -- one global_load_dwordx4).
// Float64 arithmetic follows the reference operation by operation
// (mapf_env.py:163-184, :225-266, :436-446); every add/mul that decides a bit is
// an explicit round-to-nearest intrinsic so no FMA contraction can change it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mapf_kernels.hpp"

namespace mapf {


// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53-bit uniform in [0,1) from two words -- same construction as RandomState.rand().
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b) {
    const uint64_t mant = (uint64_t(a >> 5) << 26) | uint64_t(b >> 6);
    return __dmul_rn(double(mant), 1.1102230246251565e-16 /* 2^-53 */);
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

// ------------------------------------------------------------------ one env step
template <int A>
struct StepResult {
    uint32_t next[A];
    double reward, prob;
    bool done, collision, was_terminal;
};

// nbr4[v] = {up | right << 16, down | left << 16}; STAY is the cell itself.
__device__ __forceinline__ uint32_t pick_move(uint32_t cell, uint64_t n64, uint32_t a) {
    const uint32_t moved = uint32_t(n64 >> (((a - 1u) & 3u) * 16u)) & 0xFFFFu;
    return a == 0u ? cell : moved;
}

template <int A, bool EXT_UNIFORMS>
__device__ __forceinline__ void env_transition(const EnvConsts &c, const uint2 *__restrict__ nbr4,
                                               const uint32_t (&prev)[A], const uint32_t (&goal)[A],
                                               const uint32_t (&act_in)[A], const double *ext_u,
                                               uint64_t env_id, uint64_t t, StepResult<A> &out) {
    // is_terminal(prev): mapf_env.py:210-223
    bool dup = false, all_goal = true;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        all_goal &= (prev[i] == goal[i]);
#pragma unroll
        for (int j = i + 1; j < A; ++j) dup |= (prev[i] == prev[j]);
    }
    if (dup || all_goal) {  // mapf_env.py:239-240 -- (s, 0, True, {"prob": 0}), nothing drawn
#pragma unroll
        for (int i = 0; i < A; ++i) out.next[i] = prev[i];
        out.reward = 0.0; out.prob = 0.0;
        out.done = true; out.collision = false; out.was_terminal = true;
        return;
    }
    out.was_terminal = false;

    const bool k0 = c.keep & 1u, k1 = c.keep & 2u, k2 = c.keep & 4u;
    const bool need_u = (c.keep & (c.keep - 1u)) != 0u;  // more than one candidate survives
    uint32_t act[A];
#pragma unroll
    for (int i = 0; i < A; ++i) act[i] = act_in[i] > 4u ? 0u : act_in[i];

    // the table rows of all agents are independent gathers: issue them together
    uint64_t n64[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const uint32_t cell = prev[i] < c.n_cells ? prev[i] : c.n_cells - 1u;
        const uint2 nb = nbr4[cell];
        n64[i] = uint64_t(nb.x) | (uint64_t(nb.y) << 32);
    }

    double prob = 1.0;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < A; ++i) {
        double u = 0.0;
        if (need_u) {  // uniform branch
            if (EXT_UNIFORMS) {
                u = ext_u[i];
            } else {
                if ((i & 1) == 0) {
                    const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | (uint32_t(i >> 1) << 24);
                    philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3,
                                  c.seed_lo, c.seed_hi, w);
                    u = uniform53(w[0], w[1]);
                } else {
                    u = uniform53(w[2], w[3]);
                }
            }
        }
        const uint32_t a = act[i];
        // POSSIBILITIES (gym_mapf/envs/__init__.py:19-25): right/left slips of UP,RIGHT,DOWN,LEFT
        const uint32_t ar = a == 0u ? 0u : (a & 3u) + 1u;
        const uint32_t al = a == 0u ? 0u : ((a + 2u) & 3u) + 1u;
        const uint32_t m = pick_move(prev[i], n64[i], a);
        const uint32_t r = pick_move(prev[i], n64[i], ar);
        const uint32_t l = pick_move(prev[i], n64[i], al);

        // single_agent_movements (mapf_env.py:163-184): drop p <= 0, merge equal targets
        // in first-seen order with old + new.
        uint32_t c0 = 0u, c1 = 0u, c2 = 0u;
        double q0 = 0.0, q1 = 0.0, q2 = 0.0;
        int n = 0;
        if (k0) { c0 = m; q0 = c.p0; n = 1; }
        if (k1) {
            const bool hit0 = (n >= 1) && (c0 == r);
            const bool app0 = (n == 0);
            q0 = hit0 ? __dadd_rn(q0, c.rf) : (app0 ? c.rf : q0);
            c0 = app0 ? r : c0;
            const bool app1 = !hit0 && !app0;
            c1 = app1 ? r : c1;
            q1 = app1 ? c.rf : q1;
            n = hit0 ? n : n + 1;
        }
        if (k2) {
            const bool hit0 = (n >= 1) && (c0 == l);
            const bool hit1 = !hit0 && (n >= 2) && (c1 == l);
            const bool app = !hit0 && !hit1;
            const bool app0 = app && n == 0, app1 = app && n == 1, app2 = app && n == 2;
            q0 = hit0 ? __dadd_rn(q0, c.lf) : (app0 ? c.lf : q0);
            q1 = hit1 ? __dadd_rn(q1, c.lf) : (app1 ? c.lf : q1);
            q2 = app2 ? c.lf : q2;
            c0 = app0 ? l : c0;
            c1 = app1 ? l : c1;
            c2 = app2 ? l : c2;
            n = app ? n + 1 : n;
        }
        // categorical_sample (call site mapf_env.py:255): (cumsum(p) > u).argmax(), all-False -> 0
        const double s0 = q0;
        const double s1 = __dadd_rn(s0, q1);
        const double s2 = __dadd_rn(s1, q2);
        const bool pick1 = !(s0 > u) && (n > 1) && (s1 > u);
        const bool pick2 = !(s0 > u) && !pick1 && (n > 2) && (s2 > u);
        out.next[i] = pick2 ? c2 : (pick1 ? c1 : c0);
        const double pr = pick2 ? q2 : (pick1 ? q1 : q0);
        prob = (i == 0) ? pr : __dmul_rn(prob, pr);  // total_prob *= p, agent order (:257)
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
    bool coll = false, goal_next = true;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        goal_next &= (out.next[i] == goal[i]);
        const uint32_t fwd = prev[i] | (out.next[i] << 16);
#pragma unroll
        for (int j = i + 1; j < A; ++j) {
            const uint32_t rev = out.next[j] | (prev[j] << 16);
            coll |= (out.next[i] == out.next[j]) | (fwd == rev);
        }
    }
    // calc_transition_reward_from_local_states: mapf_env.py:225-235 (collision before goal)
    out.collision = coll;
    out.done = coll || goal_next;
    out.reward = coll ? __dadd_rn(c.r_clash, living) : (goal_next ? __dadd_rn(c.r_goal, living) : living);
}

// Policy stream (oracle/philox.py random_actions_np): one Philox call per 4 agents.
template <int A>
__device__ __forceinline__ void policy_actions(const EnvConsts &c, uint64_t env_id, uint64_t t,
                                               uint32_t (&act)[A]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if ((i & 3) == 0) {
            const uint32_t c3 = (uint32_t(t >> 32) & 0x00FFFFFFu) | (uint32_t(i >> 2) << 24);
            philox4x32_10(uint32_t(env_id), uint32_t(env_id >> 32), uint32_t(t), c3, c.pol_lo, c.pol_hi, w);
        }
        act[i] = __umulhi(w[i & 3], 5u);
    }
}

}  // namespace mapf
