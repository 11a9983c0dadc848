/*
 * mapf_oracle.c -- plain-C restatement of gym-mapf's MapfEnv.step() for many envs.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (full-size parity), __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The product never links or calls it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this file (through
 * oracle/c_oracle.py) against the golden vectors the unmodified reference produced
 * (tests/golden/, generator tests/golden/make_golden.py).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/gym_mapf/envs/).  Scalar loops, one env after another, no SIMD
 * intrinsics, IEEE double arithmetic evaluated in the reference's order (build with
 * -ffp-contract=off; x86-64 SSE2 doubles round like CPython's floats).
 */
#include <stdint.h>
#include <stddef.h>

#define MAXA 128

/* ---- Philox4x32-10 (Salmon et al. SC'11); stream layout: oracle/philox.py ---- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static double slip_uniform(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t agent) {
    const uint64_t h = t >> 1;                                     /* one call: four agents, two steps */
    const uint32_t slot = 4u * (uint32_t)(t & 1) + (agent & 3u);
    const uint32_t c3 = ((uint32_t)(h >> 32) & 0xFFFFu) | (((agent >> 2) & 0x1Fu) << 16);
    uint32_t w[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)h, c3};
    uint32_t r[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)h, c3 | (slot << 23) | 0x80000000u};
    philox4x32_10(w, (uint32_t)seed, (uint32_t)(seed >> 32));
    philox4x32_10(r, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t hi16 = (w[slot >> 1] >> (16u * (slot & 1u))) & 0xFFFFu;
    const uint64_t lo37 = ((uint64_t)(r[0] & 0x1Fu) << 32) | (uint64_t)r[1];
    return (double)((hi16 << 37) | lo37) / 9007199254740992.0;
}

/* policy stream (oracle/philox.py random_actions_np): key = seed + 1, one call per agent quad per FOUR steps --
 * counter (env, t >> 2, quad), word t & 3, byte agent & 3; action = (byte * 5) >> 8 */
static uint8_t policy_action(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t agent) {
    uint64_t key = seed + 1, m = t >> 2;
    uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)m,
                     ((uint32_t)(m >> 32) & 0xFFFFu) | ((agent >> 2) << 16)};
    philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
    return (uint8_t)((((c[t & 3] >> (8u * (agent & 3u))) & 0xFFu) * 5u) >> 8);
}

/* __init__.py:19-25 POSSIBILITIES, indexed by ACTIONS order STAY,UP,RIGHT,DOWN,LEFT (:26) */
static const uint8_t SLIP_RIGHT[5] = {0, 2, 3, 4, 1};
static const uint8_t SLIP_LEFT[5]  = {0, 4, 1, 2, 3};

typedef struct {
    const uint16_t *nbr;  /* [V*5] noise-free moves (mapf_env.py:43-94 folded to a table) */
    uint32_t V, A;
    double fail_prob, r_clash, r_goal, r_living;
    uint32_t criteria;    /* 0 Makespan, 1 SoC */
} oracle_cfg;

/* mapf_env.py:210-223 */
static int is_terminal(const uint16_t *loc, const uint16_t *goal, uint32_t A) {
    for (uint32_t i = 0; i < A; ++i)
        for (uint32_t j = i + 1; j < A; ++j)
            if (loc[i] == loc[j]) return 1;
    for (uint32_t i = 0; i < A; ++i)
        if (loc[i] != goal[i]) return 0;
    return 1;
}

/* mapf_env.py:163-184 single_agent_movements: returns n, fills cells/probs */
static int movements(const oracle_cfg *g, uint16_t cell, uint8_t a, uint16_t cells[3], double probs[3]) {
    const double rf = g->fail_prob / 2, lf = g->fail_prob / 2;
    const double cand_p[3] = {1 - rf - lf, rf, lf};
    const uint8_t cand_a[3] = {a, SLIP_RIGHT[a], SLIP_LEFT[a]};
    int n = 0;
    for (int k = 0; k < 3; ++k) {
        if (!(cand_p[k] > 0)) continue;                       /* :172 */
        uint16_t nxt = g->nbr[(size_t)cell * 5 + cand_a[k]];
        int hit = -1;
        for (int m = 0; m < n; ++m) if (cells[m] == nxt) { hit = m; break; }
        if (hit >= 0) probs[hit] = probs[hit] + cand_p[k];   /* :177-179 */
        else { cells[n] = nxt; probs[n] = cand_p[k]; ++n; }  /* :181-182 */
    }
    return n;
}

/* One env, one step: mapf_env.py:237-266.  `u` NULL -> Philox.  Returns was_terminal. */
static int env_step(const oracle_cfg *g, const uint16_t *prev, const uint16_t *goal, const uint8_t *act_in,
                    const double *u, uint64_t seed, uint64_t env_id, uint64_t t,
                    uint16_t *next, double *reward, double *prob, int *done, int *collision) {
    const uint32_t A = g->A;
    if (is_terminal(prev, goal, A)) {                         /* :239-240 */
        for (uint32_t i = 0; i < A; ++i) next[i] = prev[i];
        *reward = 0.0; *prob = 0.0; *done = 1; *collision = 0;
        return 1;
    }
    uint8_t act[MAXA];
    double total = 1.0;
    for (uint32_t i = 0; i < A; ++i) {                        /* :253-257 */
        act[i] = act_in[i] > 4 ? 0 : act_in[i];
        uint16_t cells[3]; double probs[3];
        int n = movements(g, prev[i], act[i], cells, probs);
        double ui = u ? u[i] : slip_uniform(seed, env_id, t, i);
        int idx = 0; double run = 0.0;                        /* (cumsum(p) > u).argmax() */
        for (int k = 0; k < n; ++k) {
            run = (k == 0) ? probs[0] : run + probs[k];
            if (run > ui) { idx = k; break; }
        }
        next[i] = cells[idx];
        total = (i == 0) ? probs[idx] : total * probs[idx];
    }
    *prob = total;
    double living = g->r_living;                              /* :436-446 */
    if (g->criteria == 1) {
        int stayed = 0;
        for (uint32_t i = 0; i < A; ++i) if (prev[i] == goal[i] && act[i] == 0) ++stayed;
        living = (double)((int)A - stayed) * g->r_living;
    }
    int coll = 0;                                             /* :378-389 */
    for (uint32_t i = 0; i < A && !coll; ++i)
        for (uint32_t j = i + 1; j < A; ++j) {
            if (prev[i] == next[j] && prev[j] == next[i]) { coll = 1; break; }
            if (next[i] == next[j]) { coll = 1; break; }
        }
    int all_goal = 1;
    for (uint32_t i = 0; i < A; ++i) if (next[i] != goal[i]) { all_goal = 0; break; }
    if (coll) { *reward = g->r_clash + living; *done = 1; *collision = 1; }          /* :228-229 */
    else if (all_goal) { *reward = g->r_goal + living; *done = 1; *collision = 0; }   /* :231-233 */
    else { *reward = living; *done = 0; *collision = 0; }
    return 0;
}

/* Batched single step over E envs; mirrors mapf_step() of include/mapf_hip.h. */
int oracle_step(const uint16_t *nbr, uint32_t V, uint32_t A, uint64_t E,
                const uint16_t *start, int start_bcast, const uint16_t *goal, int goal_bcast,
                uint16_t *state, const uint8_t *actions, const double *uniforms,
                uint64_t seed, uint64_t env_id_offset, uint64_t t,
                double fail_prob, double r_clash, double r_goal, double r_living, uint32_t criteria,
                int auto_reset,
                uint16_t *out_local, double *out_reward, uint8_t *out_done, uint8_t *out_collision,
                double *out_prob, uint8_t *out_was_terminal) {
    if (A == 0 || A > MAXA) return -1;
    oracle_cfg g = {nbr, V, A, fail_prob, r_clash, r_goal, r_living, criteria};
    for (uint64_t e = 0; e < E; ++e) {
        uint16_t *st = state + e * A;
        const uint16_t *gl = goal + (goal_bcast ? 0 : e * A);
        uint16_t next[MAXA]; double reward, prob; int done, coll;
        int wt = env_step(&g, st, gl, actions + e * A, uniforms ? uniforms + e * A : NULL,
                          seed, env_id_offset + e, t, next, &reward, &prob, &done, &coll);
        if (out_local) for (uint32_t i = 0; i < A; ++i) out_local[e * A + i] = next[i];
        if (out_reward) out_reward[e] = reward;
        if (out_prob) out_prob[e] = prob;
        if (out_done) out_done[e] = (uint8_t)done;
        if (out_collision) out_collision[e] = (uint8_t)coll;
        if (out_was_terminal) out_was_terminal[e] = (uint8_t)wt;
        const uint16_t *src = (auto_reset && done) ? start + (start_bcast ? 0 : e * A) : next;
        for (uint32_t i = 0; i < A; ++i) st[i] = src[i];
    }
    return 0;
}

/* T steps with the synthetic policy stream (or given actions [T*E*A]) and per-env return sums:
 * the caller-side loop `s, r, done, _ = env.step(a); if done: env.reset()`.  Returns agent-steps. */
uint64_t oracle_rollout(const uint16_t *nbr, uint32_t V, uint32_t A, uint64_t E,
                        const uint16_t *start, int start_bcast, const uint16_t *goal, int goal_bcast,
                        uint16_t *state, const uint8_t *actions, uint32_t T,
                        uint64_t seed, uint64_t env_id_offset, uint64_t t0,
                        double fail_prob, double r_clash, double r_goal, double r_living, uint32_t criteria,
                        int auto_reset, double *out_returns, uint32_t *out_episodes, uint32_t *out_collisions) {
    if (A == 0 || A > MAXA) return 0;
    oracle_cfg g = {nbr, V, A, fail_prob, r_clash, r_goal, r_living, criteria};
    for (uint64_t e = 0; e < E; ++e) {
        uint16_t *st = state + e * A;
        const uint16_t *gl = goal + (goal_bcast ? 0 : e * A);
        const uint16_t *sl = start + (start_bcast ? 0 : e * A);
        double ret = 0.0; uint32_t epi = 0, ncoll = 0;
        for (uint32_t s = 0; s < T; ++s) {
            uint8_t act[MAXA];
            for (uint32_t i = 0; i < A; ++i)
                act[i] = actions ? actions[((size_t)s * E + e) * A + i] : policy_action(seed, env_id_offset + e, t0 + s, i);
            uint16_t next[MAXA]; double reward, prob; int done, coll;
            env_step(&g, st, gl, act, NULL, seed, env_id_offset + e, t0 + s, next, &reward, &prob, &done, &coll);
            ret = ret + reward; epi += (uint32_t)done; ncoll += (uint32_t)coll;
            const uint16_t *src = (auto_reset && done) ? sl : next;
            for (uint32_t i = 0; i < A; ++i) st[i] = src[i];
        }
        if (out_returns) out_returns[e] = ret;
        if (out_episodes) out_episodes[e] = epi;
        if (out_collisions) out_collisions[e] = ncoll;
    }
    return (uint64_t)T * E * A;
}

/* MAPF_POLICY_GREEDY of include/mapf_hip.h, straight from its definition (no reference counterpart): for every agent
 * the action a = 0..4 (STAY, UP, RIGHT, DOWN, LEFT) whose intended target nbr[cell][a] minimises the Manhattan distance
 * to the agent's goal over cell_rc (row | col << 16), first minimum wins.  out_actions u8[E*A]. */
void oracle_greedy_actions(const uint16_t *nbr, uint32_t A, uint64_t E, const uint32_t *cell_rc,
                           const uint16_t *state, const uint16_t *goal, int goal_bcast, uint8_t *out_actions) {
    for (uint64_t e = 0; e < E; ++e)
        for (uint32_t i = 0; i < A; ++i) {
            const uint32_t cur = state[e * A + i], g = goal[(goal_bcast ? 0 : e * A) + i];
            const int gr = (int)(cell_rc[g] & 0xFFFFu), gc = (int)(cell_rc[g] >> 16);
            int best = 0, best_d = 0;
            for (int a = 0; a < 5; ++a) {
                const uint32_t t = nbr[(size_t)cur * 5 + a];
                const int dr = (int)(cell_rc[t] & 0xFFFFu) - gr, dc = (int)(cell_rc[t] >> 16) - gc;
                const int d = (dr < 0 ? -dr : dr) + (dc < 0 ? -dc : dc);
                if (a == 0 || d < best_d) { best = a; best_d = d; }
            }
            out_actions[e * A + i] = (uint8_t)best;
        }
}
