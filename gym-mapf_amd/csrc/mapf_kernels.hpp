// Argument blocks and launcher prototypes shared by mapf_kernels.hip and mapf_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace mapf {

// Merged slip distribution of one agent for one equality pattern of its three candidate cells
// (m = intended move, r = right slip, l = left slip; code = (m==r) | (m==l) << 1 | (r==l) << 2).
// Built on the host by replaying single_agent_movements (mapf_env.py:163-184): drop p <= 0, merge equal
// cells in first-seen order with old + new, then cumsum left to right.
// Move table row of one (cell, action): everything the fast sampling path needs in ONE 16-byte read.
//   x = c0 | c1 << 16, y = c2 | code << 16 | members << 19 : the merged movement list's cells in list order, the
//       equality code of the three candidates (selects the SlipRow with the list's probabilities / full-width
//       thresholds) and SlipRow::members of that code (which candidates merged into each slot);
//   z = t0 | t1 << 16 : top 16 bits of the first two cumulative thresholds, saturated to 65535 (65535 past the end of
//       the list too).  The last threshold of a list is always 65535 (its cumulative sum is 1 up to rounding), so
//       away from ties the sampled slot is the number of passed thresholds among these two; hi16 equal to t0, t1 or
//       65535 is a tie and is resolved by the exact 53-bit path;
//   w = code * sizeof(SlipRow) : byte offset of the code's row (the sampled probability is q[slot] at its start).
using MoveEntry = uint4;
// Columns per cell: the five actions (STAY, UP, RIGHT, DOWN, LEFT).  MAPF_MV_COLS=6 (experiment builds) appends STAY again,
// so that an action byte is extracted and clamped by ONE v_min_u32 with a byte select -- measured on the packed single step
// (profiles/r04_step_table_forms.txt): the 20 % larger table costs more (+0.10 us per launch at 65536 envs: every launch
// re-fetches the table into eight L2s and its gathers are bound by the texture path's line rate) than the two vector
// instructions per agent save, so the shipped table has five columns.
#ifndef MAPF_MV_COLS
#define MAPF_MV_COLS 5
#endif
constexpr uint32_t kMvCols = MAPF_MV_COLS;
static_assert(kMvCols == 5 || kMvCols == 6, "five action columns, optionally STAY again");

struct SlipRow {
    double q[3];                   // merged probabilities, list order
    uint32_t th[3];                // top 16 bits of the thresholds, saturated to 65535 (65535 past the list end too): the
                                   // fast path of slip_move_hi; MoveEntry::z carries th[0], th[1] of the entry's code;
                                   // th[2] = th[0] | th[1] << 16 (that same word, for the COMPACT rollout form)
    uint32_t n;                    // list length, 1..3
    uint64_t thr[3];               // ceil(cum[k] * 2^53): cum[k] > u  <=>  mant(u) < thr[k]; 0 past the list end
    double cum[3];                 // running float64 sums (for caller-supplied uniforms); -inf past the list end
    uint32_t th_biased;            // (th[0] | th[1] << 16) ^ 0x80008000: MoveEntry::z as sample_slot_packed wants it, for
                                   // kernels that gather 8-byte rows and fetch the thresholds from the LDS copy of this row
    uint32_t members;              // bits 3k..3k+2: which candidates (m, r, l) merged into list slot k
};

// 8-byte form of a move-table row, for kernels that are bound by the rate of their table gathers (the packed single step:
// half the table bytes to re-fetch per launch, half the bytes per gathered lane):
//   x = c0 | c1 << 16 (as MoveEntry::x), y = c2 | (code * sizeof(SlipRow)) << 16.
// The thresholds (MoveEntry::z) are SlipRow::th_biased of the code's row, its members (exact path) SlipRow::members.
using CompactEntry = uint2;

// 4-byte DELTA form of a move-table row, for maps on which every candidate cell lies within +-127 ids of its own cell (ids run
// down the columns, so every map of at most 127 rows; mapf_create checks): bytes 0..2 = the three list cells minus the row's
// own cell (signed), byte 3 = (byte offset of the code's slip row + kDeltaRowBias) / 8.  SIX columns per cell (column 5 = STAY
// again: an action byte is extracted and clamped by one v_min_u32), padded with zero words to a multiple of 16 bytes -- the
// image the 32-agent rollout and the LDS-table single step keep in LDS (24 bytes per cell: 79 KB on a 64x64 map, where the
// 16-byte rows take 316 KB), built once on the host.
constexpr uint32_t kDeltaCols = 6, kDeltaRowBias = 16;
__host__ __device__ constexpr size_t delta_table_words(uint32_t n_cells) { return (size_t(n_cells) * kDeltaCols + 3u) & ~size_t(3); }

struct EnvConsts {
    double r_clash, r_goal, r_living;
    double p_cand[3];              // probabilities of the three candidates: intended move, right slip, left slip
    uint32_t need_rng;             // 0 when every slip list has one entry (e.g. fail_prob == 0)
    uint32_t top_tie;              // 1 when some list's last cumulative sum rounds below 1.0 (a uniform can lie past it)
    uint32_t criteria;             // 0 Makespan, 1 SoC
    uint32_t n_cells;              // V
    uint32_t seed_lo, seed_hi;     // slip-stream Philox key
    uint32_t pol_lo, pol_hi;       // policy-stream Philox key (seed + 1)
};

// Outcome of a transition as a function of the group's reduced facts f = vertex | swap << 1 | off_goal_next << 2
// (calc_transition_reward_from_local_states, mapf_env.py:225-235: collision before goal; is_terminal,
// :210-223: a swap alone leaves a non-terminal state): done | collision << 8 | next_terminal << 16 (EnvOut::status).
__host__ __device__ constexpr uint32_t outcome_status(uint32_t f) {
    const bool vertex = (f & 1u) != 0u, coll = (f & 3u) != 0u, goal_next = (f & 4u) == 0u;
    return ((coll || goal_next) ? 1u : 0u) | (coll ? 0x100u : 0u) | ((vertex || goal_next) ? 0x10000u : 0u);
}
constexpr uint32_t kTerminalStatus = 0x10001u;   // a step from a terminal state: done, no collision, still terminal

// LDS outcome table of the rollout kernel (Makespan: the living reward is a constant, so the whole reward is a
// function of f): rows 0..7 = f, rows 8..15 = "the state was terminal" (mapf_env.py:239-240: reward 0, done).
struct OutcomeRow {
    double reward;
    uint32_t status, pad;   // pad: done | collision << 16 (the packed rollout sums it and stores its bytes)
};
static_assert(sizeof(OutcomeRow) == 16, "read as one 16-byte LDS word");
// The 1 KB table image every step / rollout kernel keeps in LDS: the eight slip rows, then the sixteen outcome rows.  The
// handle's device copy (StepArgs::slip, RolloutArgs::slip) holds exactly this image -- the outcome rows built on the host
// with the same float64 additions (build_outcome_rows in mapf_capi.hip) -- so a wave can stage it with one 16-byte load
// and one 16-byte LDS write per lane.
struct TableImage {
    SlipRow slip[8];
    OutcomeRow outcome[16];
};
static_assert(sizeof(TableImage) == 1024, "64 lanes x 16 bytes");

struct StepArgs {
    EnvConsts c;
    const MoveEntry *mv;           // [V*5] move table (see MoveEntry, kMvCols)
    const uint2 *mv8;              // [V*5] the same table with 8-byte rows (CompactEntry: cells + the code's slip-row offset)
    const uint32_t *mv4;           // [delta_table_words(V)] the same table as 4-byte delta rows, six columns; null when the map's ids do not allow them
    const SlipRow *slip;           // [8] device copy of the slip table (the first part of a TableImage)
    uint16_t *state;               // [E*A] persistent env state
    const uint16_t *start, *goal;  // [E*A] or [A]
    const uint8_t *actions;        // [E*A]
    const double *uniforms;        // [E*A] or null
    uint16_t *out_local;
    double *out_reward, *out_prob;
    uint8_t *out_done, *out_collision, *out_was_terminal;
    uint64_t n_envs, env_id_offset, t;
    // Step index of a launch = t + (t_dev ? *t_dev : 0).  Plain launches bake the handle's index into t (t_dev null); a launch
    // recorded into a hipGraph (mapf_graph_begin .. mapf_graph_end) carries its offset inside the recording in t and reads
    // the recording's first index from device memory, which the graph's last node advances -- so a replay draws fresh numbers.
    const uint64_t *t_dev;
    // Scenario table (mapf_create builds it when the batch has at most 256 distinct (start row, goal row) pairs): scen[e]
    // names env e's pair, scen_rows[(2 * scen + 0 | 1) * A ..] are its start / goal cells -- the packed single step reads
    // one byte per env instead of two A-cell rows.  Null when the rows are broadcast or too varied.
    const uint8_t *scen;
    const uint16_t *scen_rows;
    uint32_t *done_flag;           // one-wave launches only (else null): host-visible word that receives done_seq
    uint32_t done_seq;             //   after every output of the step has been written (system-scope release)
    bool start_broadcast, goal_broadcast, auto_reset;
    bool state_not_terminal;       // host promise: no env is terminal when this step begins (see mapf_handle_s::may_be_terminal)
};

struct RolloutArgs {
    EnvConsts c;
    const MoveEntry *mv;
    const uint32_t *mv4;           // see StepArgs (non-null <=> mv_delta8)
    const SlipRow *slip;
    uint16_t *state;
    const uint16_t *start, *goal;
    const uint8_t *actions;        // [T*E*A] or null (on-device policy)
    const uint2 *policy_cells;     // greedy policy: [V] {row | col << 16, nine 3-bit actions}; null = random policy stream
    double *out_returns;
    uint32_t *out_episodes, *out_collisions;
    uint16_t *rec_local;
    double *rec_reward, *rec_prob;
    uint8_t *rec_done, *rec_collision;
    uint64_t n_envs, env_id_offset, t;
    const uint64_t *t_dev;         // see StepArgs: first step index = t + (t_dev ? *t_dev : 0)
    uint32_t n_steps;
    bool start_broadcast, goal_broadcast, auto_reset, accumulate;
    bool start_terminal_any;       // some env's START state is terminal (two starts coincide / every start is its goal)
    bool mv_delta8;                // every candidate cell of the move table lies within +-127 ids of its own cell (mapf_create looks)
};

// Every launcher names the kernel instance (and block size) that took the launch; the C ABI keeps the name of a
// handle's last step / rollout launch (mapf_last_kernel) so a benchmark labels its numbers with what actually ran.
void note_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// routed by agent count (mapf_dispatch.hip)
struct TransitionsArgs {
    EnvConsts c;
    const MoveEntry *mv;
    const SlipRow *slip;
    const uint16_t *goal;          // [E*A] or [A]
    const uint16_t *local;         // [N*A] query states
    const uint8_t *actions;        // [N*A] query joint actions
    const uint32_t *env_index;     // [N] env whose goals apply, or null (env 0)
    uint32_t *out_count;           // [N] number of branches (product of list lengths; 1 for a terminal state)
    uint16_t *out_next;            // [N*M*A]
    double *out_prob, *out_reward; // [N*M]
    uint8_t *out_done, *out_collision;
    uint64_t n_queries;
    uint64_t first_branch;         // first branch of the returned window (0 = from the start)
    uint32_t max_branches, n_agents;
    bool goal_broadcast;
    // The scan of the windows' lengths (launch_transitions makes it when the output is compacted or a query's window is cut
    // into several waves' pieces): query q's window starts at row block_base[q / 256] + rel[q] of the COMPACTED arrays,
    // block_base[number of blocks] = the rows in total.  compact: the output arrays ARE compacted (mapf_transitions_compact;
    // out_offset[q] receives q's first row, out_offset[N] the total) -- else reserved rows (row j of query q's window at
    // q * max_branches + j).  Rows at or beyond `capacity` are not written.
    uint32_t *rel;                 // scratch u32[N]
    uint64_t *block_base;          // scratch u64[transitions_scan_blocks(N) + 1]
    uint64_t *out_offset;
    uint64_t capacity;
    bool compact;
};
constexpr int kTransitionsMaxAgents = 16;   // 3^16 = 43 M branches per query, returned in windows
hipError_t launch_transitions(const TransitionsArgs &args, hipStream_t stream);
uint64_t transitions_scan_blocks(uint64_t n_queries);
// calc_transition_reward_from_local_states for N (prev = args.local, args.actions, next) triples; fills
// args.out_reward / out_done / out_collision [N] (max_branches, out_count, out_next, out_prob unused)
hipError_t launch_transition_rewards(const TransitionsArgs &args, const uint16_t *next, hipStream_t stream);

hipError_t launch_step(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

// *t_dev += n from one thread: the last node of a recorded graph (mapf_graph_end)
hipError_t launch_advance_step_index(uint64_t *t_dev, uint64_t n, hipStream_t stream);
// *t_dev = value (mapf_graph_launch, when the host side moved the step index since the last replay)
hipError_t launch_set_step_index(uint64_t *t_dev, uint64_t value, hipStream_t stream);

hipError_t launch_query_terminal(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);

// lane-group family (mapf_lg_kernels.hip): any A up to 128, run-time A
constexpr int kTpeMaxAgents = 16;         // thread-per-env step kernels are specialised for A = 1..16
#ifndef MAPF_TPE_ROLLOUT_MAX              // (tools/exp/tpe_spill_repro.sh builds a library that dispatches the spilling ones)
#define MAPF_TPE_ROLLOUT_MAX 6
#endif
constexpr int kTpeRolloutMaxAgents = MAPF_TPE_ROLLOUT_MAX;   // ... their rollout form is dispatched only where it is spill-free
// Layout choices of the kernels, fixed per handle at mapf_create (the MAPF_TUNE override -- "key=value,..." -- is read
// there, so a process can hold handles with different settings: the tests do).  The key of each field is named beside it.
struct RolloutTuning {
    bool quad_lanes = true;          // quad_lanes=0 forces the pair layout
    uint64_t quad_min_lanes = 0;     // four agents per lane need at least this many lanes (quad_min_lanes; default: one
                                     // wave on every SIMD of the device); below that two agents per lane
    uint64_t oct_min_lanes = 0;      // eight agents per lane need at least this many lanes (oct_min_lanes; default:
                                     //   two waves on every SIMD)
    int force_k = 0;                 // k=2|4|8 pins the agents per lane of the packed layout (tests)
    size_t mv_lds_max_bytes = 0;     // largest move table staged into LDS (mv_lds_max_bytes; default: two blocks per CU)
    bool bitmap_pairs = true;        // bitmap_pairs=0: the 32-agent rollout keeps the all-pairs collision tests (tests compare both)
    bool bitmap_delta_rows = true;   // bitmap_delta=0: the bitmap form never uses the 4-byte delta rows (tests compare the tables)
    bool bitmap_stay_column = true;  // bitmap_staycol=0: the bitmap form always stages the four-column table (tests)
    unsigned bitmap_block = 0;       // bitmap_block=512|1024: block size of the bitmap form (experiments / tests; 0 = by batch)
    int step_big = 1;                // step_big: the packed single step's resident-grid / LDS-table form -- 0 never, 1 for batches
                                     //   of at least four times what the device holds at once (default), 2 whenever it fits (tests)
    unsigned step_block = 0;         // step_block=64|128|256|512: block size of the plain packed single step (experiments; 0 = by batch)
    int step_delta = 1;              // step_delta: the single step's LDS table of 4-byte delta rows -- 0 never, 1 where the 16-byte rows
                                     //   do not fit and the batch gives every CU a block (default), 2 whenever it fits (tests, experiments)
    bool scen_table = true;          // scen_table=0: never build the scenario table (StepArgs::scen) -- tests compare both forms
};
RolloutTuning default_rollout_tuning(int device, std::string *err);   // (reads MAPF_TUNE: mapf_lg_rollout.hip)
// ... its arithmetic: the defaults of a device with n_cu compute units, overridden by `text` ("key=value,...", may be null)
RolloutTuning rollout_tuning_for(int n_cu, const char *text, std::string *err);
hipError_t launch_step_lg(int n_agents, const StepArgs &args, const RolloutTuning &tune, hipStream_t stream);
// packed layout of the single step (mapf_lq_step.hip): true when it took the launch (*err = its status)
bool try_launch_step_lq(int n_agents, const StepArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err);
hipError_t launch_rollout_lg(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, hipStream_t stream);
// packed layout of the fused rollout (2 or 4 agents per lane, mapf_lq_rollout.hip): true when it took the launch (*err = its status)
bool try_launch_rollout_lq(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, hipStream_t stream, hipError_t *err);
// What try_launch_rollout_lq decides before it launches -- pure arithmetic over the launch's shape (args.c.n_cells, n_envs, n_steps,
// c.top_tie, actions / mv4 / mv_delta8 present or not), the tuning and the device's CU count, so it can be swept without a
// device (mapf_debug_rollout_plan, tests/test_cabi_and_host.py): false = no packed form applies.
struct LqPlan {
    int K = 0, Q = 0;                // agents per lane, lanes per env
    int form = 0;                    // 0 full 16-byte rows, 1 8-byte rows, 2 / 3 bitmaps behind four / five 8-byte columns, 4 bitmaps behind
                                     // full rows, 5 bitmaps behind 4-byte delta rows
    unsigned block = 0;              // threads per block
    size_t lds_bytes = 0;            // the kernel's LDS image without the bitmaps (what the launcher is handed)
    size_t lds_total = 0;            // ... with them: the dynamic LDS segment of the launch, <= 160 KB
};
bool plan_rollout_lq(int n_agents, const RolloutArgs &args, const RolloutTuning &tune, int n_cu, LqPlan *plan);
int lg_group_size(int n_agents);

// per-group entry points: group g holds the kernels specialised for A in 4g+1 .. 4g+4
hipError_t launch_step_g0(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g0(int n_agents, const RolloutArgs &args, hipStream_t stream);

hipError_t launch_step_g1(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g1(int n_agents, const RolloutArgs &args, hipStream_t stream);

hipError_t launch_step_g2(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g2(int n_agents, const RolloutArgs &args, hipStream_t stream);

hipError_t launch_step_g3(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g3(int n_agents, const RolloutArgs &args, hipStream_t stream);





}  // namespace mapf
