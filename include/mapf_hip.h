/*
 * mapf_hip.h -- C ABI of libmapf_hip.so: the MI355X (gfx950) implementation of
 * gym-mapf's batched MapfEnv.step() hot path.
 *
 * The reference has no FFI; its hot path sits behind a Python class
 * (gym_mapf/envs/mapf_env.py:115 `class MapfEnv`).  Each entry point below
 * names the reference method/lines it replaces.  The reference-side binding
 * (ctypes) a maintainer would add is shown in INTEGRATION.md and implemented in
 * gym-mapf_amd/gym_mapf_amd/_native.py.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative
 *     MAPF_E* code, and mapf_last_error() (thread-local) holds the message;
 *   - the handle owns its device buffers and (unless one is supplied) its HIP
 *     stream; the caller owns every array it passes in;
 *   - array arguments are HOST pointers unless the handle was created with
 *     MAPF_FLAG_DEVICE_PTRS, in which case they are DEVICE pointers (e.g.
 *     torch tensors' data_ptr()) and calls only enqueue work -- use
 *     mapf_sync() before reading results;
 *   - one handle is driven by one host thread at a time;
 *   - layouts are env-major: x[e*A + i] is agent i of env e.  Cells are
 *     "local ids": the index of a free cell in the reference's column-major
 *     enumeration (mapf_env.py:142 valid_locations, grid.py:37-40), uint16
 *     (every reference map has <= 65536 free cells);
 *   - actions: 0 STAY, 1 UP, 2 RIGHT, 3 DOWN, 4 LEFT
 *     (gym_mapf/envs/__init__.py:26 ACTIONS); values > 4 are treated as STAY.
 */
#ifndef MAPF_HIP_H
#define MAPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPF_ABI_VERSION 5

/* status codes */
#define MAPF_OK            0
#define MAPF_EINVAL       -1  /* bad argument / descriptor                   */
#define MAPF_EHIP         -2  /* HIP runtime error (message has the detail)  */
#define MAPF_ENODEVICE    -3  /* no usable HIP device                        */
#define MAPF_EUNSUPPORTED -4  /* e.g. n_agents beyond MAPF_MAX_AGENTS        */

#define MAPF_MAX_AGENTS 128

/* optimisation criteria: mapf_env.py:31-33 OptimizationCriteria */
#define MAPF_MAKESPAN 0
#define MAPF_SOC      1

/* mapf_desc.flags */
#define MAPF_FLAG_DEVICE_PTRS     0x1u  /* array args of step/rollout/... are device pointers */
#define MAPF_FLAG_START_BROADCAST 0x2u  /* desc.start is [A], shared by all envs              */
#define MAPF_FLAG_GOAL_BROADCAST  0x4u  /* desc.goal  is [A], shared by all envs              */
/* Kernel family (default: chosen from A and E).  Both compute identical results.
 *   THREAD_PER_ENV: one lane owns all A agents of an env (A <= 16 only);
 *   LANE_GROUP:     an env is spread over pow2(ceil(A/2)) adjacent lanes, two agents per lane. */
#define MAPF_FLAG_THREAD_PER_ENV  0x10u
#define MAPF_FLAG_LANE_GROUP      0x20u

/* step_flags / rollout flags */
#define MAPF_STEP_AUTO_RESET      0x1u  /* after a step returns done, the env's stored state
                                           goes back to its start cells (mapf_env.py:290-293
                                           reset(), no reseed) -- the outputs still report the
                                           state step() returned                              */

typedef struct mapf_handle_s *mapf_handle_t;

/*
 * Everything MapfEnv.__init__ (mapf_env.py:116-161) derives from its
 * arguments, precomputed by the host wrapper and uploaded once.
 * desc.nbr / start / goal are always HOST pointers.
 */
typedef struct mapf_desc {
    uint32_t struct_size;    /* = sizeof(mapf_desc)                                        */
    uint32_t n_cells;        /* V = len(valid_locations), 1..65536                         */
    uint32_t n_agents;       /* A, 1..MAPF_MAX_AGENTS                                      */
    uint32_t criteria;       /* MAPF_MAKESPAN | MAPF_SOC                                   */
    uint64_t n_envs;         /* E: envs owned by this handle (this rank's shard)           */
    uint64_t env_id_offset;  /* global id of local env 0 (RNG counters use global ids)     */
    uint64_t seed;           /* Philox key (slip stream); policy stream uses seed + 1      */
    const uint16_t *nbr;     /* [V*5]: nbr[v*5+a] = cell reached from v by noise-free a    */
                             /*        (mapf_env.py:43-94 execute_action, folded to a table) */
    const uint16_t *start;   /* [E*A] or [A]: start cells (mapf_env.py:128 agents_starts)   */
    const uint16_t *goal;    /* [E*A] or [A]: goal cells                                    */
    double fail_prob;        /* mapf_env.py:130; right_fail = left_fail = fail_prob / 2     */
    double r_clash;          /* reward_of_collision                                         */
    double r_goal;           /* reward_of_goal                                              */
    double r_living;         /* reward_of_living                                            */
    int32_t device;          /* HIP device ordinal                                          */
    uint32_t flags;          /* MAPF_FLAG_*                                                 */
    void *stream;            /* hipStream_t to enqueue on, or NULL: the handle creates one  */
} mapf_desc;

/*
 * Kernel-dispatch overrides (tests, A/B measurements; none is needed for correct or fast operation): ONE environment
 * variable, MAPF_TUNE="key=value,key=value,...", read by mapf_create -- so a process can hold handles created under
 * different settings; an unknown key or a malformed item fails mapf_create with MAPF_EINVAL.  Keys (integers):
 *   quad_lanes=0         never the packed rollout / step layouts: the lane-group kernels take every launch
 *   k=2|4|8              pin the packed layout's agents per lane (default: by batch -- 8 from two waves per SIMD of that form
 *                        on, 4 from one, else 2; 32 agents on 64x64 maps: 4 with occupancy bitmaps)
 *   quad_min_lanes=n, oct_min_lanes=n   lane counts from which four / eight agents per lane are used
 *   mv_lds_max_bytes=n   largest full move table the rollout kernels stage into LDS (default: half of the 160 KB); above it
 *                        8-byte / 4-byte rows, 0 = always gather from global memory
 *   scen_table=0         never build the scenario table (<= 256 distinct (start row, goal row) pairs: one byte per env)
 *   bitmap_pairs=0       32 agents: all agent pairs instead of the per-env LDS occupancy bitmaps
 *   bitmap_block=512|1024, bitmap_staycol=0, bitmap_delta=0   forms of the 32-agent bitmap rollout (block size; four-column
 *                        table with made-up STAY rows; never 4-byte delta rows)
 *   step_big=0|1|2       the single step's resident grid with the move table in LDS: never / by batch (default) / whenever it fits
 *   step_delta=0|1|2     ... with 4-byte delta rows (64x64 maps): never / from one full residency on (default) / whenever it fits
 *   step_block=64|128|256|512   block size of the plain packed single step
 */

/* Replaces MapfEnv.__init__'s state setup; state = start cells, step index t = 0. */
int mapf_create(const mapf_desc *desc, mapf_handle_t *out_handle);
int mapf_destroy(mapf_handle_t h);

/*
 * MapfEnv.reset() (mapf_env.py:290-293) for every env, or for the envs whose
 * mask byte is non-zero (mask: u8[E] or NULL).  Does not touch the RNG counters.
 */
int mapf_reset(mapf_handle_t h, const uint8_t *mask);

/*
 * MapfEnv.step() (mapf_env.py:237-266) for all E envs in one kernel launch.
 *   actions   u8 [E*A]   per-agent actions (the decoded joint action, :242-243)
 *   uniforms  f64[E*A]   the rand() values the reference would draw in agent order
 *                        (:255), or NULL: drawn on device from Philox4x32-10 keyed by
 *                        (seed; global env id, step index t, agent) -- the exact counter layout is
 *                        oracle/philox.py's (since ABI 4: one call per agent QUAD per two steps; ABI 3
 *                        used one per agent pair per four steps, so the same seed draws other
 *                        numbers than it did there)
 *   out_local u16[E*A]   the state step() returns, as per-agent cells
 *   out_reward f64[E], out_done u8[E], out_collision u8[E], out_prob f64[E]
 *   out_was_terminal u8[E]: 1 where the env was already terminal, i.e. the reference
 *                        returned (s, 0, True, {"prob": 0}) and drew nothing (:239-240)
 * Any out_* may be NULL.  Increments the handle's step index t.
 */
int mapf_step(mapf_handle_t h, const uint8_t *actions, const double *uniforms,
              uint16_t *out_local, double *out_reward, uint8_t *out_done,
              uint8_t *out_collision, double *out_prob, uint8_t *out_was_terminal,
              uint32_t step_flags);

/*
 * T fused steps in ONE launch (the caller-side loop around step()):
 * state stays in registers, actions are streamed from `actions` u8[T*E*A] (step-major) or,
 * when NULL, drawn from the policy stream (uniform over the 5 actions, key seed + 1).
 * Every reward is added, in step order, to out_returns f64[E]; out_episodes u32[E] counts
 * steps that returned done; out_collisions u32[E] counts collision steps.  Optional
 * trajectory recording (any pointer may be NULL), step-major like `actions`:
 *   rec_local u16[T*E*A], rec_reward f64[T*E], rec_done u8[T*E], rec_collision u8[T*E],
 *   rec_prob f64[T*E].
 * Advances t by T.  Equivalent to T mapf_step calls with the same flags.
 */
typedef struct mapf_rollout_io {
    uint32_t struct_size;
    uint32_t n_steps;           /* T */
    uint32_t step_flags;        /* MAPF_STEP_* */
    uint32_t accumulate;        /* 0: out_* are overwritten; 1: added to existing values */
    const uint8_t *actions;     /* [T*E*A] or NULL */
    double   *out_returns;      /* [E] or NULL */
    uint32_t *out_episodes;     /* [E] or NULL */
    uint32_t *out_collisions;   /* [E] or NULL */
    uint16_t *rec_local;
    double   *rec_reward;
    uint8_t  *rec_done;
    uint8_t  *rec_collision;
    double   *rec_prob;
} mapf_rollout_io;
int mapf_rollout(mapf_handle_t h, const mapf_rollout_io *io);

/* Synthetic policy used by bench/rollout: fills actions u8[n_steps*E*A] for step indices
 * t0 .. t0+n_steps-1 from the policy stream (oracle/philox.py random_actions_np; ABI 5: key seed + 1, ONE Philox
 * call per agent quad per FOUR steps -- counter (env, t >> 2, quad), word t & 3, byte agent & 3, action =
 * (byte * 5) >> 8; ABI <= 4 drew one call per quad per step and a 32-bit word per action). */
int mapf_fill_random_actions(mapf_handle_t h, uint8_t *actions, uint64_t t0, uint32_t n_steps);

/*
 * On-device policy of mapf_rollout calls that pass actions == NULL (the reference has no policy; this stands in
 * for the caller-side `a = policy(s)` of the loop around MapfEnv.step).
 *   MAPF_POLICY_RANDOM (default): the uniform-random policy stream, as mapf_fill_random_actions.
 *   MAPF_POLICY_GREEDY: every agent takes the first action in ACTIONS order (STAY, UP, RIGHT, DOWN, LEFT) whose
 *     intended target cell is closest (Manhattan distance over grid rows / columns) to its goal -- i.e. the first
 *     unblocked move that brings it one step closer, else STAY.  Slip still applies.  cell_rc u32[V] (HOST
 *     pointer, copied): row | col << 16 of every free cell (rows and columns below 65536).
 */
#define MAPF_POLICY_RANDOM 0
#define MAPF_POLICY_GREEDY 1
int mapf_set_policy(mapf_handle_t h, int policy, const uint32_t *cell_rc);

/*
 * MapfEnv.P[s][a] (mapf_env.py:448-478 _get_transitions): for each of n_queries (state, joint action) pairs,
 * every branch of the joint slip distribution, in the reference's order (itertools.product over the agents'
 * merged movement lists, agent 0 slowest).  Does not touch the handle's env state or step index.
 *   local u16[N*A], actions u8[N*A]; env_index u32[N] selects whose goals apply (NULL: env 0)
 *   max_branches M: rows reserved per query (3^A always suffices); out_count u32[N] reports the true number
 *   out_next u16[N*M*A], out_prob f64[N*M], out_reward f64[N*M], out_done u8[N*M], out_collision u8[N*M]
 * Rows b >= out_count[q] are left untouched.  A terminal state yields one branch (prob 1.0, reward 0, done).
 * Supported for n_agents <= 16 (3^A branches per query: up to 43 M -- page through them with
 * mapf_transitions_window; planners decompose larger problems with get_local_view).  Any out_* may be NULL.
 */
int mapf_transitions(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                     const uint32_t *env_index, uint32_t max_branches, uint32_t *out_count, uint16_t *out_next,
                     double *out_prob, double *out_reward, uint8_t *out_done, uint8_t *out_collision);
/* The same enumeration, one WINDOW at a time: row j of query q's outputs holds branch first_branch + j (rows whose
 * branch index is >= out_count[q] are left untouched); out_count always reports the full branch count.  The reference
 * builds the whole list at once (mapf_env.py:465-476); for more than ~8 agents that list does not fit anywhere. */
int mapf_transitions_window(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                            const uint32_t *env_index, uint64_t first_branch, uint32_t max_branches, uint32_t *out_count,
                            uint16_t *out_next, double *out_prob, double *out_reward, uint8_t *out_done,
                            uint8_t *out_collision);

/*
 * The same enumeration with COMPACTED rows (ABI 5): the window [first_branch, first_branch + max_branches) of every query,
 * back to back -- query q's rows are rows out_offset[q] .. out_offset[q + 1] - 1 of every output array, in the reference's
 * branch order (mapf_env.py:465-476 builds exactly such a list per (s, a); a query reserves 3^A rows in the calls above and
 * a room map fills a fifth of them, so a planner that sweeps many states wants them dense).
 *   out_offset u64[N + 1] (required): exclusive scan of the windows' lengths; out_offset[N] = the number of rows needed;
 *   capacity_rows: rows the out_* arrays hold -- rows at or beyond it are NOT written (no overrun): a caller checks
 *     out_offset[N] <= capacity_rows (after mapf_sync in device-pointer mode) and calls again with larger arrays if not;
 *   out_count u32[N] (optional): the FULL branch count of every query, as above;
 *   out_next u16[R*A], out_prob f64[R], out_reward f64[R], out_done u8[R], out_collision u8[R] with R = capacity_rows.
 * Three launches: window lengths + block-local scan, scan of the block totals, rows.  n_agents <= 16.
 */
int mapf_transitions_compact(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                             const uint32_t *env_index, uint64_t first_branch, uint32_t max_branches, uint64_t capacity_rows,
                             uint64_t *out_offset, uint32_t *out_count, uint16_t *out_next, double *out_prob, double *out_reward,
                             uint8_t *out_done, uint8_t *out_collision);

/*
 * MapfEnv.calc_transition_reward_from_local_states (mapf_env.py:225-235, with _living_reward :436-446 and
 * _is_collision_transition_from_local_states :378-389) for n_queries given transitions:
 *   prev_local u16[N*A], actions u8[N*A] (the decoded joint action), next_local u16[N*A];
 *   env_index u32[N] selects whose goals apply (NULL: env 0);
 *   out_reward f64[N], out_done u8[N], out_collision u8[N]  (the method's (reward, done, collision) triple).
 * Like the reference method it does not test is_terminal(prev).  Any n_agents; any out_* may be NULL.
 */
int mapf_transition_rewards(mapf_handle_t h, uint64_t n_queries, const uint16_t *prev_local, const uint8_t *actions,
                            const uint16_t *next_local, const uint32_t *env_index, double *out_reward, uint8_t *out_done,
                            uint8_t *out_collision);

/* MapfEnv.is_terminal (mapf_env.py:210-223) of every env's CURRENT state: out_terminal u8[E] is 1
 * where two agents share a cell or every agent is on its goal (a step there is a no-op). */
int mapf_query_terminal(mapf_handle_t h, uint8_t *out_terminal);

/* env.s as per-agent cells + the step index (the whole mutable state: the RNG is
 * counter-based).  set_state validates cells < V. */
int mapf_get_state(mapf_handle_t h, uint16_t *local, uint64_t *t);
int mapf_set_state(mapf_handle_t h, const uint16_t *local, uint64_t t);

/*
 * The handle's state buffer itself: out_state receives a DEVICE pointer to u16[E*A], env.s of every env as per-agent cells
 * -- after a step with MAPF_STEP_AUTO_RESET that is the state the next step starts from (a finished episode shows its start
 * cells, the usual vector-env convention; mapf_step's out_local reports the state step() itself returned).  A caller that
 * reads the next observation here passes out_local = NULL to mapf_step and the step writes the cells ONCE.  The pointer
 * stays valid for the handle's lifetime; its contents change with every step / rollout / reset / set_state enqueued on
 * the handle's stream (read it on that stream or after mapf_sync).  No reference counterpart: MapfEnv.s (mapf_env.py:265).
 */
int mapf_state_view(mapf_handle_t h, const uint16_t **out_state);

/*
 * The view is READ-ONLY (const): the library tracks on the host whether some env can be terminal (after a step that
 * auto-reset every finished episode none can) and then runs the step instance without is_terminal(prev)
 * (mapf_env.py:238-240).  A caller that writes the state buffer itself (custom reset, curriculum) -- through a
 * cast of this pointer, or a framework tensor that wraps it -- must say so BEFORE the next step / rollout:
 * mapf_invalidate_state makes the next launches test is_terminal(prev) again, as mapf_set_state does.  (ABI 4.)
 */
int mapf_invalidate_state(mapf_handle_t h);

/*
 * Recording the caller-side loop around MapfEnv.step (mapf_env.py:237-266) into a hipGraph.  Between mapf_graph_begin and
 * mapf_graph_end every mapf_step / mapf_rollout / mapf_reset / mapf_fill_random_actions call on a MAPF_FLAG_DEVICE_PTRS
 * handle is recorded on the handle's stream instead of executed (work the caller enqueues on that stream in between --
 * e.g. its policy network -- is recorded with it).  mapf_graph_launch replays the recording n_replays times; the step
 * index t is kept in device memory for recorded launches and advanced by the recording's last node, so every replay draws
 * fresh random numbers: K replays of an N-step recording produce exactly the results of K*N mapf_step calls with the same
 * arguments.  The arrays named in recorded calls must stay allocated while the graph lives.  Calls that wait for the
 * stream or move the step index from the host (mapf_sync, mapf_timer_*, mapf_get_state, mapf_set_state, mapf_set_policy)
 * fail with MAPF_EINVAL while recording, as does recording on a host-pointer handle.  A mapf_step / mapf_rollout enqueued while
 * the CALLER captures the stream it gave the handle (hipStreamBeginCapture, torch.cuda.graph) is refused too: it would
 * bake its step index -- and random numbers -- into the caller's graph; record it here instead.  Recordings that are
 * still alive are destroyed with their handle.
 */
typedef struct mapf_graph_s *mapf_graph_t;
int mapf_graph_begin(mapf_handle_t h);
int mapf_graph_end(mapf_handle_t h, mapf_graph_t *out_graph);
int mapf_graph_launch(mapf_handle_t h, mapf_graph_t graph, uint32_t n_replays);
int mapf_graph_steps(mapf_graph_t graph, uint64_t *out_steps);   /* env-steps one replay advances the handle by */
int mapf_graph_destroy(mapf_handle_t h, mapf_graph_t graph);

/* Block until everything enqueued on the handle's stream has finished. */
int mapf_sync(mapf_handle_t h);

/* HIP-event timing on the handle's stream: begin records an event, end records another,
 * waits for it and returns the elapsed milliseconds between the two. */
int mapf_timer_begin(mapf_handle_t h);
int mapf_timer_end(mapf_handle_t h, double *out_ms);

/* The hipStream_t the handle enqueues on (for interop with other libraries). */
int mapf_get_stream(mapf_handle_t h, void **out_stream);

/* Which kernel instance took the handle's most recent mapf_step (MAPF_KERNEL_STEP), mapf_rollout
 * (MAPF_KERNEL_ROLLOUT) or mapf_transitions (MAPF_KERNEL_TRANSITIONS) launch, e.g. "lq_rollout_kernel<Q=2,RECORD,STREAM> block=512": the library chooses the
 * lane layout from A, E and the table size, so measurements label themselves with what actually ran (no reference
 * counterpart: the reference has one code path, mapf_env.py:237-266).  "" before the first launch.  The string is
 * owned by the handle and valid until its next launch of that kind. */
#define MAPF_KERNEL_STEP    0
#define MAPF_KERNEL_ROLLOUT 1
#define MAPF_KERNEL_TRANSITIONS 2   /* mapf_transitions / mapf_transitions_window */
const char *mapf_last_kernel(mapf_handle_t h, int which);

int mapf_device_count(int *out_count);
const char *mapf_last_error(void);
const char *mapf_version(void);
/* MAPF_ABI_VERSION of the loaded library: a binding written against another number must refuse to drive it (the Philox
 * counter layouts are part of the ABI: the same seed draws other numbers under another version). */
int mapf_abi_version(void);

/* Diagnostic, no device needed and none touched: which packed rollout form (agents per lane, lanes per env, table rows,
 * block size, LDS bytes) a mapf_rollout launch of this shape would take on a device with n_cu compute units -- the
 * arithmetic the launcher runs before every launch (no reference counterpart: the reference has one code path).
 * n_cells = cells of the map incl. the blocked id; streamed = actions given (1) or drawn in the kernel (0); delta_rows = the
 * map admits 4-byte delta rows (every neighbour id within +-127 of its cell); tune = a MAPF_TUNE string or NULL for the
 * defaults (the MAPF_TUNE environment variable is NOT read).
 * out[0..5] = {agents per lane K, lanes per env Q, form (0 full 16-byte rows, 1 8-byte rows, 2 / 3 bitmaps behind four /
 * five 8-byte columns, 4 bitmaps behind full rows, 5 bitmaps behind delta rows), threads per block, LDS bytes of the table
 * image, LDS bytes of the launch}.  Returns 1 when a packed form applies, 0 when the lane-group kernel takes the launch,
 * MAPF_EINVAL (< 0) for a malformed tune string or null out. */
int mapf_debug_rollout_plan(uint32_t n_cells, int n_agents, uint64_t n_envs, uint32_t n_steps, int streamed, int delta_rows,
                            int n_cu, const char *tune, uint64_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_HIP_H */
