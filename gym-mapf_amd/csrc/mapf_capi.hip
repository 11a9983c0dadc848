// C ABI of libmapf_hip.so (declared in include/mapf_hip.h): handle management, host<->device
// staging for the host-pointer mode, argument validation, and the launches.
#include "mapf_hip.h"
#include "mapf_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_last_error;
thread_local char g_noted_kernel[160] = "";

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(MAPF_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(_e, #expr);    \
    } while (0)

// Scratch device buffer that grows on demand (host-pointer mode staging).
struct DeviceBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// Pinned host block mapped into the device's address space: for tiny host-mode calls (the scalar MapfEnv.step()
// regime: one env, a handful of agents) the kernel reads its inputs from and writes its outputs to this block
// directly, so a call is one launch + one stream sync instead of up to nine hipMemcpyAsync round trips.
struct PinnedBlock {
    char *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), bytes, hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) { host = nullptr; return e; }
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0);
        if (e != hipSuccess) { (void)hipHostFree(host); host = dev = nullptr; return e; }
        cap = bytes;
        return hipSuccess;
    }
    void release() { if (host) (void)hipHostFree(host); host = dev = nullptr; cap = 0; }
};
constexpr size_t kZeroCopyMaxBytes = 16 * 1024;   // beyond this the DMA copies win over PCIe-direct accesses

}  // namespace

namespace mapf {
void note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_noted_kernel, sizeof(g_noted_kernel), fmt, ap);
    va_end(ap);
}
}  // namespace mapf

struct mapf_handle_s {
    int device = 0;
    std::string last_step_kernel, last_rollout_kernel, last_transitions_kernel;
    uint32_t V = 0, A = 0, flags = 0;
    uint64_t E = 0, env_id_offset = 0, t = 0;
    mapf::EnvConsts c{};
    bool start_broadcast = false, goal_broadcast = false, device_ptrs = false, own_stream = false;
    bool stream_exposed = false;   // mapf_get_stream was called: somebody else may capture the stream (check_foreign_capture)
    bool lane_group = false;   // kernel family
    // The thread-per-env rollout specialisations for A >= 8 need SGPR spills (the pointer-heavy argument block
    // plus A-wide unrolling); only spill-free kernels are dispatched, so those sizes use the lane-group rollout.
    bool lane_group_rollout = false;
    bool start_terminal_any = true;   // is_terminal(start) for some env (looked up once at create)
    bool mv_delta8 = false;           // every neighbour id lies within +-127 of its cell's id (the 4-byte delta rows of the bitmap rollout)
    // Can some env be terminal right now?  Not after a call that auto-reset every finished episode (unless a START state
    // is itself terminal) or after a full reset; yes after steps without auto-reset and after set_state.  The packed
    // single step runs its instance without is_terminal(prev) when the answer is no.  While recording a graph the
    // question is asked about the recording's own history only (cap_may_be_terminal: the first recorded step cannot know
    // what precedes a replay).
    bool may_be_terminal = true, cap_may_be_terminal = true;
    mapf::RolloutTuning tune;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    mapf::MoveEntry *mv = nullptr;
    mapf::CompactEntry *mv8 = nullptr;   // the 8-byte-row form of the move table (the packed single step gathers from it)
    uint32_t *mv4 = nullptr;          // the 4-byte delta-row form (kernels that keep the table of a 64x64 map in LDS); null unless mv_delta8
    mapf::SlipRow *slip = nullptr;
    std::vector<uint16_t> nbr;        // host copy of the neighbour table (policy tables are derived from it)
    uint2 *policy_cells = nullptr;    // greedy policy table (mapf_set_policy); null = random policy stream
    uint16_t *state = nullptr, *start = nullptr, *goal = nullptr;
    // host-pointer mode staging
    DeviceBuf s_actions, s_uniforms, s_local, s_reward, s_prob, s_done, s_coll, s_term, s_mask, s_ret, s_epi, s_ncoll;
    DeviceBuf x_local, x_reward, x_prob, x_done, x_coll;   // stand-ins for trajectory arrays the caller left out
    DeviceBuf q_local, q_actions, q_env, q_count, q_next, q_prob, q_reward, q_done, q_coll, q_next_in, q_offset;   // mapf_transitions staging
    DeviceBuf q_rel, q_blocks;        // mapf_transitions_compact: the scan's scratch (in-block offsets, block bases)
    PinnedBlock pinned;               // zero-copy staging of tiny host-mode steps
    // scenario table (StepArgs::scen): built at create when the batch has <= 256 distinct (start row, goal row) pairs
    uint8_t *scen = nullptr;
    uint16_t *scen_rows = nullptr;
    uint32_t n_scen = 0;
    // recording into a hipGraph (mapf_graph_begin .. mapf_graph_end): recorded launches take their step index from
    // *t_dev + their offset inside the recording; t_dev_value = what *t_dev holds once the stream has drained
    uint64_t *t_dev = nullptr;
    uint64_t t_dev_value = 0, cap_steps = 0;
    bool capturing = false;
    std::vector<struct mapf_graph_s *> graphs;   // recordings that are still alive (destroyed with the handle at the latest)
};

struct mapf_graph_s {
    mapf_handle_t owner = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t steps = 0;               // env-steps one replay advances the handle by
    bool ends_may_be_terminal = true; // mapf_handle_s::may_be_terminal after a replay (conservative: true unless the
                                      // recording's last state-changing call auto-resets every finished episode)
};

namespace {

int check_handle(mapf_handle_t h) {
    if (!h) return fail(MAPF_EINVAL, "null handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    return MAPF_OK;
}

// entry points that wait for the stream, move the step index from the host side or free what a recorded node uses
// cannot run between mapf_graph_begin and mapf_graph_end
int check_not_recording(mapf_handle_t h, const char *what) {
    if (h->capturing) return fail(MAPF_EINVAL, std::string(what) + ": not allowed while a graph is being recorded (call mapf_graph_end first)");
    return MAPF_OK;
}

// A step or rollout enqueued while somebody ELSE is capturing the stream (a caller-owned stream under torch.cuda.graph /
// hipStreamBeginCapture) would bake the handle's current step index into the captured launch: every replay would reuse
// the same random numbers, silently.  Only mapf_graph_begin knows how to record a launch (device-side step index).
int check_foreign_capture(mapf_handle_t h, const char *what) {
    // (nobody else can capture a stream the handle created -- until mapf_get_stream has handed it out)
    if (h->capturing || (h->own_stream && !h->stream_exposed)) return MAPF_OK;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &status) != hipSuccess) { (void)hipGetLastError(); return MAPF_OK; }
    if (status != hipStreamCaptureStatusNone)
        return fail(MAPF_EINVAL, std::string(what) + ": the stream is being captured outside mapf_graph_begin -- the launch would bake its "
                                 "step index (and random numbers) into the graph; record it between mapf_graph_begin and mapf_graph_end");
    return MAPF_OK;
}

// The lane-group kernels address every array with 32-bit byte offsets (one SGPR base + one VGPR offset per
// access): the largest array of a call must stay below 4 GiB.  rows = E (step) or T*E (rollout).
int check_extent(mapf_handle_t h, uint64_t rows, bool has_uniforms) {
    const uint64_t limit = uint64_t(1) << 32;
    const uint64_t per_agent = has_uniforms ? sizeof(double) : sizeof(uint16_t);
    if (rows * h->A * per_agent >= limit || rows * sizeof(double) >= limit)
        return fail(MAPF_EINVAL, "call too large: every array of one call must stay below 4 GiB (use fewer steps per rollout)");
    return MAPF_OK;
}

bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// stage a host input on the device (host mode) or pass the device pointer through
template <typename T>
int stage_in(mapf_handle_t h, DeviceBuf &buf, const T *src, size_t count, const T **out, const char *name) {
    if (!src) { *out = nullptr; return MAPF_OK; }
    if (h->device_ptrs) {
        if (misaligned(src)) return fail(MAPF_EINVAL, std::string(name) + ": device pointer must be 16-byte aligned");
        *out = src;
        return MAPF_OK;
    }
    HIP_TRY(buf.reserve(count * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(buf.ptr, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *out = static_cast<const T *>(buf.ptr);
    return MAPF_OK;
}

template <typename T>
int stage_out(mapf_handle_t h, DeviceBuf &buf, T *dst, size_t count, T **out, const char *name) {
    if (!dst) { *out = nullptr; return MAPF_OK; }
    if (h->device_ptrs) {
        if (misaligned(dst)) return fail(MAPF_EINVAL, std::string(name) + ": device pointer must be 16-byte aligned");
        *out = dst;
        return MAPF_OK;
    }
    HIP_TRY(buf.reserve(count * sizeof(T)));
    *out = static_cast<T *>(buf.ptr);
    return MAPF_OK;
}

template <typename T>
int fetch_out(mapf_handle_t h, const T *dev, T *dst, size_t count) {
    if (!dst || h->device_ptrs) return MAPF_OK;
    HIP_TRY(hipMemcpyAsync(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    return MAPF_OK;
}

// Replay single_agent_movements (mapf_env.py:163-184) for each equality pattern of the candidate cells
// (m intended, r right slip, l left slip), in IEEE double and the reference's evaluation order -- the
// same operations CPython performs: rf = lf = fail_prob / 2 (:131-132), p0 = 1 - rf - lf (:167), drop
// p <= 0 (:172), merge equal cells with old + new in first-seen order (:177-182); cum = np.cumsum.
// Returns true when some list has more than one entry, i.e. a uniform is actually consumed.
bool build_slip_table(double fail_prob, mapf::SlipRow (&rows)[8], double (&cand_p)[3]) {
    const double rf = fail_prob / 2, lf = fail_prob / 2;
    cand_p[0] = (1 - rf) - lf; cand_p[1] = rf; cand_p[2] = lf;
    bool any_multi = false;
    for (unsigned code = 0; code < 8; ++code) {
        // representative cells realising the pattern (inconsistent codes cannot occur at run time)
        const int m = 0, r = (code & 1u) ? 0 : 1, l = (code & 2u) ? 0 : ((code & 4u) ? r : 2);
        const int cand_cell[3] = {m, r, l};
        int cells[3] = {-1, -1, -1}, members[3] = {0, 0, 0}, n = 0;
        double q[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            if (!(cand_p[k] > 0)) continue;
            int hit = -1;
            for (int j = 0; j < n; ++j) if (cells[j] == cand_cell[k]) { hit = j; break; }
            if (hit >= 0) { q[hit] = q[hit] + cand_p[k]; members[hit] |= 1 << k; }
            else { cells[n] = cand_cell[k]; q[n] = cand_p[k]; members[n] = 1 << k; ++n; }
        }
        mapf::SlipRow &row = rows[code];
        std::memset(&row, 0, sizeof(row));
        row.n = uint32_t(n);
        double run = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (k < n) {
                run = (k == 0) ? q[0] : run + q[k];
                row.cum[k] = run;
                row.q[k] = q[k];
                const double scaled = std::ceil(std::ldexp(run, 53));          // exact: power-of-two scaling
                row.thr[k] = scaled >= 9007199254740992.0 ? (uint64_t(1) << 53) : (scaled <= 0 ? 0 : uint64_t(scaled));
                row.th[k] = uint32_t(row.thr[k] >> 37) > 65535u ? 65535u : uint32_t(row.thr[k] >> 37);   // saturated (see SlipRow)
                row.members |= uint32_t(members[k]) << (3 * k);
            } else {
                row.cum[k] = -HUGE_VAL;
                row.q[k] = 0.0;
                row.thr[k] = 0;
                row.th[k] = 65535u;
            }
        }
        // th[2] is never compared against (a list's last threshold is 65535 by construction): it carries th[0] | th[1] << 16,
        // the word MoveEntry::z holds, for kernels that keep only the cells of a row in LDS (mapf_lq_rollout.hip COMPACT)
        row.th[2] = row.th[0] | (row.th[1] << 16);
        row.th_biased = row.th[2] ^ 0x80008000u;   // (sample_slot_packed compares bias-shifted half-words)
        any_multi |= n > 1;
    }
    return any_multi;
}

// The sixteen outcome rows of the table image (mapf_kernels.hpp TableImage; device twin: stage_outcome_rows in mapf_lg.hpp):
// rows 0..7 = f = vertex | swap << 1 | off_goal << 2, rows 8..15 = the state was terminal (mapf_env.py:239-240).  Makespan's
// reward is a function of f: r_clash + living / r_goal + living / living (calc_transition_reward_from_local_states,
// mapf_env.py:225-235; one float64 addition each, as the reference's `reward + living_reward`).
void build_outcome_rows(const mapf::EnvConsts &c, mapf::OutcomeRow (&rows)[16]) {
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t st = mapf::outcome_status(i & 7u);
        const double r = (st & 0x100u) ? c.r_clash + c.r_living : ((st & 1u) ? c.r_goal + c.r_living : c.r_living);
        rows[i].reward = i < 8u ? r : 0.0;
        rows[i].status = i < 8u ? st : mapf::kTerminalStatus;
        rows[i].pad = (rows[i].status & 1u) | ((rows[i].status & 0x100u) << 8);
    }
}

void destroy_impl(mapf_handle_t h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (mapf_graph_s *g : h->graphs) {   // recordings name the handle's buffers: they go first
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
    }
    h->graphs.clear();
    for (DeviceBuf *b : {&h->s_actions, &h->s_uniforms, &h->s_local, &h->s_reward, &h->s_prob, &h->s_done,
                         &h->s_coll, &h->s_term, &h->s_mask, &h->s_ret, &h->s_epi, &h->s_ncoll, &h->x_local, &h->x_reward,
                         &h->x_prob, &h->x_done, &h->x_coll, &h->q_local, &h->q_actions, &h->q_env, &h->q_count, &h->q_next,
                         &h->q_prob, &h->q_reward, &h->q_done, &h->q_coll, &h->q_next_in, &h->q_offset, &h->q_rel, &h->q_blocks})
        b->release();
    h->pinned.release();
    if (h->scen) (void)hipFree(h->scen);
    if (h->scen_rows) (void)hipFree(h->scen_rows);
    if (h->t_dev) (void)hipFree(h->t_dev);
    if (h->mv) (void)hipFree(h->mv);
    if (h->mv8) (void)hipFree(h->mv8);
    if (h->mv4) (void)hipFree(h->mv4);
    if (h->policy_cells) (void)hipFree(h->policy_cells);
    if (h->slip) (void)hipFree(h->slip);
    if (h->state) (void)hipFree(h->state);
    if (h->start) (void)hipFree(h->start);
    if (h->goal) (void)hipFree(h->goal);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

extern "C" {

const char *mapf_last_error(void) { return g_last_error.c_str(); }

const char *mapf_version(void) { return "mapf_hip 0.5.0 (abi 5, gfx950)"; }

int mapf_abi_version(void) { return MAPF_ABI_VERSION; }

int mapf_debug_rollout_plan(uint32_t n_cells, int n_agents, uint64_t n_envs, uint32_t n_steps, int streamed, int delta_rows,
                            int n_cu, const char *tune, uint64_t out[6]) {
    if (!out) return fail(MAPF_EINVAL, "out is null");
    if (n_agents < 1 || n_cells < 2 || n_cu < 1) return fail(MAPF_EINVAL, "mapf_debug_rollout_plan: n_agents >= 1, n_cells >= 2, n_cu >= 1");
    std::string tune_error;
    const mapf::RolloutTuning t = mapf::rollout_tuning_for(n_cu, tune, &tune_error);
    if (!tune_error.empty()) return fail(MAPF_EINVAL, tune_error);
    // only the fields the plan reads: the shape, and which optional arrays are present (never dereferenced)
    static const uint8_t present = 0;
    mapf::RolloutArgs args{};
    args.c.n_cells = n_cells;
    args.n_envs = n_envs;
    args.n_steps = n_steps;
    args.actions = streamed ? &present : nullptr;
    args.mv_delta8 = delta_rows != 0;
    args.mv4 = delta_rows ? reinterpret_cast<const uint32_t *>(&present) : nullptr;
    mapf::LqPlan plan;
    const bool packed = mapf::plan_rollout_lq(n_agents, args, t, n_cu, &plan);
    out[0] = uint64_t(plan.K); out[1] = uint64_t(plan.Q); out[2] = uint64_t(plan.form);
    out[3] = plan.block; out[4] = plan.lds_bytes; out[5] = plan.lds_total;
    return packed ? 1 : 0;
}

int mapf_device_count(int *out_count) {
    if (!out_count) return fail(MAPF_EINVAL, "out_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; (void)hipGetLastError(); return fail(MAPF_ENODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *out_count = n;
    return MAPF_OK;
}

int mapf_create(const mapf_desc *d, mapf_handle_t *out_handle) {
    if (!d || !out_handle) return fail(MAPF_EINVAL, "null descriptor or output");
    *out_handle = nullptr;
    if (d->struct_size != sizeof(mapf_desc)) return fail(MAPF_EINVAL, "mapf_desc.struct_size does not match this library");
    if (d->n_cells == 0 || d->n_cells > 65536u) return fail(MAPF_EINVAL, "n_cells must be in 1..65536 (uint16 local ids)");
    if (d->n_agents == 0) return fail(MAPF_EINVAL, "n_agents must be >= 1");
    if (d->n_agents > MAPF_MAX_AGENTS) return fail(MAPF_EUNSUPPORTED, "n_agents beyond MAPF_MAX_AGENTS (128)");
    if ((d->flags & MAPF_FLAG_THREAD_PER_ENV) && (d->flags & MAPF_FLAG_LANE_GROUP))
        return fail(MAPF_EINVAL, "MAPF_FLAG_THREAD_PER_ENV and MAPF_FLAG_LANE_GROUP are exclusive");
    if ((d->flags & MAPF_FLAG_THREAD_PER_ENV) && d->n_agents > uint32_t(mapf::kTpeMaxAgents))
        return fail(MAPF_EUNSUPPORTED, "thread-per-env kernels exist for n_agents <= 16 only");
    if (d->criteria > MAPF_SOC) return fail(MAPF_EINVAL, "criteria must be MAPF_MAKESPAN or MAPF_SOC");
    if (!d->nbr || !d->start || !d->goal) return fail(MAPF_EINVAL, "nbr/start/goal must be non-null host pointers");
    if (!std::isfinite(d->fail_prob) || !std::isfinite(d->r_clash) || !std::isfinite(d->r_goal) || !std::isfinite(d->r_living))
        return fail(MAPF_EINVAL, "fail_prob and rewards must be finite");
    if (d->n_envs > (uint64_t(1) << 31)) return fail(MAPF_EINVAL, "n_envs too large for one handle");

    const uint32_t V = d->n_cells, A = d->n_agents;
    const uint64_t E = d->n_envs;
    const bool sb = d->flags & MAPF_FLAG_START_BROADCAST, gb = d->flags & MAPF_FLAG_GOAL_BROADCAST;
    for (uint64_t i = 0; i < uint64_t(V) * 5; ++i)
        if (d->nbr[i] >= V) return fail(MAPF_EINVAL, "nbr entry out of range");
    for (uint32_t v = 0; v < V; ++v)
        if (d->nbr[uint64_t(v) * 5] != v) return fail(MAPF_EINVAL, "nbr[v][STAY] must be v");
    const uint64_t n_start = (sb ? 1 : E) * A, n_goal = (gb ? 1 : E) * A;
    for (uint64_t i = 0; i < n_start; ++i)
        if (d->start[i] >= V) return fail(MAPF_EINVAL, "start cell out of range");
    for (uint64_t i = 0; i < n_goal; ++i)
        if (d->goal[i] >= V) return fail(MAPF_EINVAL, "goal cell out of range");

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return fail(MAPF_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                        "); this library has no CPU fallback");
    }
    if (d->device < 0 || d->device >= n_dev) return fail(MAPF_EINVAL, "device ordinal out of range");
    HIP_TRY(hipSetDevice(d->device));

    mapf_handle_t h = new (std::nothrow) mapf_handle_s();
    if (!h) return fail(MAPF_EHIP, "out of host memory");
    h->device = d->device; h->V = V; h->A = A; h->E = E; h->flags = d->flags;
    h->env_id_offset = d->env_id_offset; h->t = 0;
    h->start_broadcast = sb; h->goal_broadcast = gb;
    h->device_ptrs = d->flags & MAPF_FLAG_DEVICE_PTRS;
    // kernel family: forced by flag, else lane groups whenever an env has more than one agent pair
    if (d->flags & MAPF_FLAG_THREAD_PER_ENV) h->lane_group = false;
    else if (d->flags & MAPF_FLAG_LANE_GROUP) h->lane_group = true;
    else h->lane_group = A > 2;
    h->lane_group_rollout = h->lane_group || A > uint32_t(mapf::kTpeRolloutMaxAgents);
    {
        std::string tune_error;
        h->tune = mapf::default_rollout_tuning(d->device, &tune_error);
        if (!tune_error.empty()) { destroy_impl(h); return fail(MAPF_EINVAL, tune_error); }
    }
    h->mv_delta8 = true;
    for (uint32_t v = 0; v < V && h->mv_delta8; ++v)
        for (uint32_t a = 0; a < 5; ++a) {
            const int64_t delta = int64_t(d->nbr[uint64_t(v) * 5 + a]) - int64_t(v);
            if (delta < -127 || delta > 127) h->mv_delta8 = false;
        }

    mapf::SlipRow slip_host[8];
    h->c.need_rng = build_slip_table(d->fail_prob, slip_host, h->c.p_cand) ? 1u : 0u;
    h->c.top_tie = 0u;
    for (unsigned code = 0; code < 8; ++code)
        if (slip_host[code].n == 3 && slip_host[code].thr[2] < (uint64_t(1) << 53)) h->c.top_tie = 1u;   // (shorter lists compare against their last threshold, 65535)
    // the single-step kernels rebuild a merged probability from its members instead of reading the row: the ordered
    // sum ((m ? p_m : 0) + (r ? p_r : 0)) + (l ? p_l : 0) must reproduce the table bit for bit
    for (unsigned code = 0; code < 8; ++code)
        for (unsigned k = 0; k < slip_host[code].n; ++k) {
            const unsigned mem = (slip_host[code].members >> (3 * k)) & 7u;
            const double q = (((mem & 1u) ? h->c.p_cand[0] : 0.0) + ((mem & 2u) ? h->c.p_cand[1] : 0.0)) + ((mem & 4u) ? h->c.p_cand[2] : 0.0);
            if (std::memcmp(&q, &slip_host[code].q[k], sizeof(q)) != 0) {
                destroy_impl(h);
                return fail(MAPF_EINVAL, "create: merged slip probabilities are not reproducible from their members");
            }
        }
    h->c.r_clash = d->r_clash; h->c.r_goal = d->r_goal; h->c.r_living = d->r_living;
    h->c.criteria = d->criteria; h->c.n_cells = V;
    h->c.seed_lo = uint32_t(d->seed); h->c.seed_hi = uint32_t(d->seed >> 32);
    const uint64_t pol = d->seed + 1;
    h->c.pol_lo = uint32_t(pol); h->c.pol_hi = uint32_t(pol >> 32);

#define CREATE_TRY(expr)                                                   \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) { destroy_impl(h); return hip_fail(_e, #expr); } \
    } while (0)

    if (d->stream) { h->stream = static_cast<hipStream_t>(d->stream); h->own_stream = false; }
    else { CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    CREATE_TRY(hipEventCreate(&h->ev_begin));
    CREATE_TRY(hipEventCreate(&h->ev_end));

    // Move table: for every (cell, action) the merged movement list of single_agent_movements
    // (mapf_env.py:163-184) -- its cells in list order and the equality code of the three candidates.
    const double rf_ = d->fail_prob / 2, lf_ = d->fail_prob / 2;
    const bool keep[3] = {((1 - rf_) - lf_) > 0, rf_ > 0, lf_ > 0};
    static const uint8_t kSlipRight[5] = {0, 2, 3, 4, 1}, kSlipLeft[5] = {0, 4, 1, 2, 3};   // __init__.py:19-25
    std::vector<mapf::MoveEntry> packed(size_t(V) * mapf::kMvCols);   // column 5 = STAY again (kMvCols)
    for (uint32_t v = 0; v < V; ++v) {
        const uint16_t *r = d->nbr + uint64_t(v) * 5;
        for (uint32_t col = 0; col < mapf::kMvCols; ++col) {
            const uint32_t a = col < 5 ? col : 0;
            const uint16_t cand[3] = {r[a], r[kSlipRight[a]], r[kSlipLeft[a]]};
            const uint64_t code = (cand[0] == cand[1] ? 1u : 0u) | (cand[0] == cand[2] ? 2u : 0u) | (cand[1] == cand[2] ? 4u : 0u);
            uint16_t cells[3] = {0, 0, 0};
            int n = 0;
            for (int k = 0; k < 3; ++k) {
                if (!keep[k]) continue;
                bool seen = false;
                for (int j = 0; j < n; ++j) seen |= (cells[j] == cand[k]);
                if (!seen) cells[n++] = cand[k];
            }
            // top 16 bits of the list's cumulative thresholds, saturated (see MoveEntry)
            uint32_t t16[3];
            // (past the list end: 65535 as well -- `hi < 65535` only fails in a tie, and an earlier slot has matched by then)
            for (int k = 0; k < 3; ++k)
                t16[k] = slip_host[code].th[k];
            packed[size_t(v) * mapf::kMvCols + col] = make_uint4(uint32_t(cells[0]) | (uint32_t(cells[1]) << 16),
                                                    uint32_t(cells[2]) | (uint32_t(code) << 16) | (slip_host[code].members << 19),
                                                    t16[0] | (t16[1] << 16), uint32_t(code * sizeof(mapf::SlipRow)));
        }
    }
    const size_t row = size_t(A) * sizeof(uint16_t);
    h->nbr.assign(d->nbr, d->nbr + size_t(V) * 5);
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->mv), packed.size() * sizeof(mapf::MoveEntry)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->state), (E ? E : 1) * row));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->start), (sb ? 1 : (E ? E : 1)) * row));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->goal), (gb ? 1 : (E ? E : 1)) * row));
    {   // the 1 KB table image: slip rows, then the outcome rows
        mapf::TableImage image;
        std::memcpy(image.slip, slip_host, sizeof(slip_host));
        build_outcome_rows(h->c, image.outcome);
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->slip), sizeof(image)));
        CREATE_TRY(hipMemcpy(h->slip, &image, sizeof(image), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMemcpy(h->mv, packed.data(), packed.size() * sizeof(mapf::MoveEntry), hipMemcpyHostToDevice));
    {
        std::vector<mapf::CompactEntry> compact(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) compact[i] = make_uint2(packed[i].x, (packed[i].y & 0xFFFFu) | (packed[i].w << 16));
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->mv8), compact.size() * sizeof(mapf::CompactEntry)));
        CREATE_TRY(hipMemcpy(h->mv8, compact.data(), compact.size() * sizeof(mapf::CompactEntry), hipMemcpyHostToDevice));
    }
    if (h->mv_delta8) {   // 4-byte delta rows, six columns (mapf_kernels.hpp kDeltaCols); the padding words stay zero
        std::vector<uint32_t> delta(mapf::delta_table_words(V), 0u);
        for (uint32_t v = 0; v < V; ++v)
            for (uint32_t col = 0; col < mapf::kDeltaCols; ++col) {
                const mapf::MoveEntry &e = packed[size_t(v) * mapf::kMvCols + (col < mapf::kMvCols ? col : 0u)];
                delta[size_t(v) * mapf::kDeltaCols + col] = ((e.x - v) & 0xFFu) | ((((e.x >> 16) - v) & 0xFFu) << 8) | (((e.y - v) & 0xFFu) << 16) |
                                                            (((e.w + mapf::kDeltaRowBias) >> 3) << 24);
            }
        CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->mv4), delta.size() * sizeof(uint32_t)));
        CREATE_TRY(hipMemcpy(h->mv4, delta.data(), delta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (n_start) CREATE_TRY(hipMemcpy(h->start, d->start, n_start * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (n_goal) CREATE_TRY(hipMemcpy(h->goal, d->goal, n_goal * sizeof(uint16_t), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->t_dev), sizeof(uint64_t)));
    CREATE_TRY(hipMemset(h->t_dev, 0, sizeof(uint64_t)));
    if (!(sb && gb) && E > 0 && h->tune.scen_table) {
        // Scenario table: the distinct (start row, goal row) pairs of the batch, when there are few (the BASELINE
        // configurations draw every env's rows from 6 or 25 scenario files), and one byte per env naming its pair.
        std::unordered_map<std::string, uint32_t> ids;
        std::vector<uint8_t> scen(E);
        std::vector<uint16_t> rows;
        bool few = true;
        std::string key(2 * row, '\0');
        for (uint64_t e = 0; e < E && few; ++e) {
            std::memcpy(&key[0], d->start + (sb ? 0 : e * A), row);
            std::memcpy(&key[row], d->goal + (gb ? 0 : e * A), row);
            auto it = ids.find(key);
            if (it == ids.end()) {
                if (ids.size() == 256) { few = false; break; }
                it = ids.emplace(key, uint32_t(ids.size())).first;
                rows.insert(rows.end(), reinterpret_cast<const uint16_t *>(key.data()), reinterpret_cast<const uint16_t *>(key.data()) + 2 * A);
            }
            scen[e] = uint8_t(it->second);
        }
        if (few) {
            h->n_scen = uint32_t(ids.size());
            CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->scen), E));
            CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&h->scen_rows), rows.size() * sizeof(uint16_t)));
            CREATE_TRY(hipMemcpy(h->scen, scen.data(), E, hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(h->scen_rows, rows.data(), rows.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    }
    CREATE_TRY(mapf::launch_reset(int(A), h->state, h->start, sb, nullptr, E, h->stream));
    {   // is any env's start state terminal?  (the rollout kernels specialise on "no": state == start right now)
        std::vector<uint8_t> flags(E ? E : 1, 0);
        CREATE_TRY(h->s_term.reserve(E ? E : 1));
        CREATE_TRY(mapf::launch_query_terminal(int(A), h->state, h->goal, gb, static_cast<uint8_t *>(h->s_term.ptr), E, h->stream));
        if (E) CREATE_TRY(hipMemcpyAsync(flags.data(), h->s_term.ptr, E, hipMemcpyDeviceToHost, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream));
        h->start_terminal_any = false;
        for (uint64_t e = 0; e < E; ++e) h->start_terminal_any |= flags[e] != 0;
        h->may_be_terminal = h->start_terminal_any;
    }
#undef CREATE_TRY
    *out_handle = h;
    return MAPF_OK;
}

int mapf_destroy(mapf_handle_t h) {
    if (!h) return fail(MAPF_EINVAL, "null handle");
    if (h->capturing) {   // an open recording dies with the handle
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(h->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        h->capturing = false;
    }
    destroy_impl(h);
    return MAPF_OK;
}

int mapf_sync(mapf_handle_t h) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_sync")) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

const char *mapf_last_kernel(mapf_handle_t h, int which) {
    if (!h) return "";
    if (which == MAPF_KERNEL_TRANSITIONS) return h->last_transitions_kernel.c_str();
    return which == MAPF_KERNEL_ROLLOUT ? h->last_rollout_kernel.c_str() : h->last_step_kernel.c_str();
}

int mapf_get_stream(mapf_handle_t h, void **out_stream) {
    if (!h || !out_stream) return fail(MAPF_EINVAL, "null handle or output");
    *out_stream = static_cast<void *>(h->stream);
    h->stream_exposed = true;
    return MAPF_OK;
}

int mapf_timer_begin(mapf_handle_t h) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_timer_begin")) return rc;
    HIP_TRY(hipEventRecord(h->ev_begin, h->stream));
    return MAPF_OK;
}

int mapf_timer_end(mapf_handle_t h, double *out_ms) {
    if (int rc = check_handle(h)) return rc;
    if (!out_ms) return fail(MAPF_EINVAL, "out_ms is null");
    if (int rc = check_not_recording(h, "mapf_timer_end")) return rc;
    HIP_TRY(hipEventRecord(h->ev_end, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev_end));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    *out_ms = double(ms);
    return MAPF_OK;
}

int mapf_reset(mapf_handle_t h, const uint8_t *mask) {
    if (int rc = check_handle(h)) return rc;
    const uint8_t *d_mask = nullptr;
    if (int rc = stage_in(h, h->s_mask, mask, size_t(h->E), &d_mask, "mask")) return rc;
    HIP_TRY(mapf::launch_reset(int(h->A), h->state, h->start, h->start_broadcast, d_mask, h->E, h->stream));
    if (!mask) (h->capturing ? h->cap_may_be_terminal : h->may_be_terminal) = h->start_terminal_any;
    if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

int mapf_step(mapf_handle_t h, const uint8_t *actions, const double *uniforms, uint16_t *out_local,
              double *out_reward, uint8_t *out_done, uint8_t *out_collision, double *out_prob,
              uint8_t *out_was_terminal, uint32_t step_flags) {
    if (int rc = check_handle(h)) return rc;
    if (!actions) return fail(MAPF_EINVAL, "actions is null");
    if (step_flags & ~MAPF_STEP_AUTO_RESET) return fail(MAPF_EINVAL, "unknown step flag");
    if (int rc = check_foreign_capture(h, "mapf_step")) return rc;
    if (int rc = check_extent(h, h->E, uniforms != nullptr)) return rc;
    const size_t E = size_t(h->E), EA = E * h->A;
    mapf::StepArgs a{};
    a.c = h->c; a.mv = h->mv; a.mv8 = h->mv8; a.mv4 = h->mv4; a.slip = h->slip; a.state = h->state; a.start = h->start; a.goal = h->goal;
    a.n_envs = h->E; a.env_id_offset = h->env_id_offset;
    // a recorded launch: offset inside the recording + the device-side index (see StepArgs::t_dev)
    a.t = h->capturing ? h->cap_steps : h->t;
    a.t_dev = h->capturing ? h->t_dev : nullptr;
    a.scen = h->scen; a.scen_rows = h->scen_rows;
    a.start_broadcast = h->start_broadcast; a.goal_broadcast = h->goal_broadcast;
    a.auto_reset = step_flags & MAPF_STEP_AUTO_RESET;
    bool &may_be_terminal = h->capturing ? h->cap_may_be_terminal : h->may_be_terminal;
    a.state_not_terminal = !may_be_terminal;
    // after this step: every finished episode is back on its start cells (auto-reset), or anything goes
    const bool may_be_terminal_after = a.auto_reset ? h->start_terminal_any : true;
    if (!h->device_ptrs) {
        // tiny host-mode call: inputs and outputs live in one pinned, device-mapped block (16-byte aligned slots)
        auto slot = [](size_t &off, size_t bytes) { const size_t at = off; off += (bytes + 15u) & ~size_t(15); return at; };
        size_t total = 0;
        const size_t o_act = slot(total, EA), o_uni = slot(total, uniforms ? EA * sizeof(double) : 0),
                     o_loc = slot(total, out_local ? EA * sizeof(uint16_t) : 0), o_rew = slot(total, out_reward ? E * sizeof(double) : 0),
                     o_prob = slot(total, out_prob ? E * sizeof(double) : 0), o_done = slot(total, out_done ? E : 0),
                     o_coll = slot(total, out_collision ? E : 0), o_term = slot(total, out_was_terminal ? E : 0),
                     o_flag = slot(total, sizeof(uint32_t));
        if (total <= kZeroCopyMaxBytes && E > 0) {
            HIP_TRY(h->pinned.reserve(kZeroCopyMaxBytes));
            char *hp = h->pinned.host, *dp = h->pinned.dev;
            std::memcpy(hp + o_act, actions, EA);
            a.actions = reinterpret_cast<const uint8_t *>(dp + o_act);
            if (uniforms) { std::memcpy(hp + o_uni, uniforms, EA * sizeof(double)); a.uniforms = reinterpret_cast<const double *>(dp + o_uni); }
            if (out_local) a.out_local = reinterpret_cast<uint16_t *>(dp + o_loc);
            if (out_reward) a.out_reward = reinterpret_cast<double *>(dp + o_rew);
            if (out_prob) a.out_prob = reinterpret_cast<double *>(dp + o_prob);
            if (out_done) a.out_done = reinterpret_cast<uint8_t *>(dp + o_done);
            if (out_collision) a.out_collision = reinterpret_cast<uint8_t *>(dp + o_coll);
            if (out_was_terminal) a.out_was_terminal = reinterpret_cast<uint8_t *>(dp + o_term);
            const uint64_t launch_threads = h->lane_group ? E * uint64_t(mapf::lg_group_size(int(h->A))) : E;
            const bool flagged = launch_threads <= 64;             // one wave: see below
            const uint32_t seq = uint32_t(h->t) + 1u;
            if (flagged) {
                *reinterpret_cast<volatile uint32_t *>(hp + o_flag) = seq - 1u;
                a.done_flag = reinterpret_cast<uint32_t *>(dp + o_flag);
                a.done_seq = seq;
            }
            HIP_TRY(h->lane_group ? mapf::launch_step_lg(int(h->A), a, h->tune, h->stream) : mapf::launch_step(int(h->A), a, h->stream));
            if (h->last_step_kernel != g_noted_kernel) h->last_step_kernel = g_noted_kernel;
            h->t += 1;
            may_be_terminal = may_be_terminal_after;
            // A one-wave launch signals its end itself: its last instruction stores the call's sequence number into the
            // pinned block (system-scope release after all outputs), and the host spins on that word instead of paying the
            // sleeping stream wait (~6 us of a ~16 us call).  Anything larger, or a slow launch, uses hipStreamSynchronize.
            bool signalled = false;
            if (flagged) {
                volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(hp + o_flag);
                for (int spin = 0; spin < 200000 && !signalled; ++spin) signalled = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
            }
            if (!signalled) HIP_TRY(hipStreamSynchronize(h->stream));
            if (out_local) std::memcpy(out_local, hp + o_loc, EA * sizeof(uint16_t));
            if (out_reward) std::memcpy(out_reward, hp + o_rew, E * sizeof(double));
            if (out_prob) std::memcpy(out_prob, hp + o_prob, E * sizeof(double));
            if (out_done) std::memcpy(out_done, hp + o_done, E);
            if (out_collision) std::memcpy(out_collision, hp + o_coll, E);
            if (out_was_terminal) std::memcpy(out_was_terminal, hp + o_term, E);
            return MAPF_OK;
        }
    }
    if (int rc = stage_in(h, h->s_actions, actions, EA, &a.actions, "actions")) return rc;
    if (int rc = stage_in(h, h->s_uniforms, uniforms, EA, &a.uniforms, "uniforms")) return rc;
    if (int rc = stage_out(h, h->s_local, out_local, EA, &a.out_local, "out_local")) return rc;
    if (int rc = stage_out(h, h->s_reward, out_reward, E, &a.out_reward, "out_reward")) return rc;
    if (int rc = stage_out(h, h->s_prob, out_prob, E, &a.out_prob, "out_prob")) return rc;
    if (int rc = stage_out(h, h->s_done, out_done, E, &a.out_done, "out_done")) return rc;
    if (int rc = stage_out(h, h->s_coll, out_collision, E, &a.out_collision, "out_collision")) return rc;
    if (int rc = stage_out(h, h->s_term, out_was_terminal, E, &a.out_was_terminal, "out_was_terminal")) return rc;
    HIP_TRY(h->lane_group ? mapf::launch_step_lg(int(h->A), a, h->tune, h->stream) : mapf::launch_step(int(h->A), a, h->stream));
    if (h->last_step_kernel != g_noted_kernel) h->last_step_kernel = g_noted_kernel;
    if (h->capturing) h->cap_steps += 1; else h->t += 1;
    may_be_terminal = may_be_terminal_after;
    if (!h->device_ptrs) {
        if (int rc = fetch_out(h, a.out_local, out_local, EA)) return rc;
        if (int rc = fetch_out(h, a.out_reward, out_reward, E)) return rc;
        if (int rc = fetch_out(h, a.out_prob, out_prob, E)) return rc;
        if (int rc = fetch_out(h, a.out_done, out_done, E)) return rc;
        if (int rc = fetch_out(h, a.out_collision, out_collision, E)) return rc;
        if (int rc = fetch_out(h, a.out_was_terminal, out_was_terminal, E)) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return MAPF_OK;
}

namespace {
// The recording rollout kernels write all five trajectory arrays (no per-array branches in the step loop): when
// the caller asked for only some of them, the others go to handle-owned scratch.
// A stand-in buffer that a RECORDED rollout node names must not move: DeviceBuf::reserve() frees and reallocates on growth,
// and the next replay of that node would write into freed memory.  So while a graph is being recorded or recorded graphs
// of the handle are alive, a stand-in may be allocated (nothing names it yet) but not grown.
int reserve_stand_in(mapf_handle_t h, DeviceBuf &buf, size_t bytes, const char *name) {
    if (buf.ptr && bytes > buf.cap && (h->capturing || !h->graphs.empty()))
        return fail(MAPF_EINVAL, std::string("rollout: the handle's stand-in for ") + name + " would have to grow while a recorded graph names it -- pass all "
                                 "five rec_* arrays, record the longest rollout first, or destroy the handle's graphs");
    HIP_TRY(buf.reserve(bytes));
    return MAPF_OK;
}
int complete_recording(mapf_handle_t h, mapf::RolloutArgs &a, size_t TE, size_t TEA) {
    if (!(a.rec_local || a.rec_reward || a.rec_prob || a.rec_done || a.rec_collision)) return MAPF_OK;
    if (!a.rec_local) { if (int rc = reserve_stand_in(h, h->x_local, TEA * sizeof(uint16_t), "rec_local")) return rc; a.rec_local = static_cast<uint16_t *>(h->x_local.ptr); }
    if (!a.rec_reward) { if (int rc = reserve_stand_in(h, h->x_reward, TE * sizeof(double), "rec_reward")) return rc; a.rec_reward = static_cast<double *>(h->x_reward.ptr); }
    if (!a.rec_prob) { if (int rc = reserve_stand_in(h, h->x_prob, TE * sizeof(double), "rec_prob")) return rc; a.rec_prob = static_cast<double *>(h->x_prob.ptr); }
    if (!a.rec_done) { if (int rc = reserve_stand_in(h, h->x_done, TE, "rec_done")) return rc; a.rec_done = static_cast<uint8_t *>(h->x_done.ptr); }
    if (!a.rec_collision) { if (int rc = reserve_stand_in(h, h->x_coll, TE, "rec_collision")) return rc; a.rec_collision = static_cast<uint8_t *>(h->x_coll.ptr); }
    return MAPF_OK;
}
}  // namespace

int mapf_rollout(mapf_handle_t h, const mapf_rollout_io *io) {
    if (int rc = check_handle(h)) return rc;
    if (!io || io->struct_size != sizeof(mapf_rollout_io)) return fail(MAPF_EINVAL, "bad mapf_rollout_io");
    if (io->step_flags & ~MAPF_STEP_AUTO_RESET) return fail(MAPF_EINVAL, "unknown step flag");
    if (int rc = check_foreign_capture(h, "mapf_rollout")) return rc;
    if (int rc = check_extent(h, uint64_t(h->E) * io->n_steps, false)) return rc;
    const size_t E = size_t(h->E), T = io->n_steps, TE = T * E, TEA = TE * h->A;
    mapf::RolloutArgs a{};
    a.c = h->c; a.mv = h->mv; a.mv4 = h->mv4; a.slip = h->slip; a.state = h->state; a.start = h->start; a.goal = h->goal;
    a.n_envs = h->E; a.env_id_offset = h->env_id_offset; a.n_steps = io->n_steps;
    a.t = h->capturing ? h->cap_steps : h->t;
    a.t_dev = h->capturing ? h->t_dev : nullptr;
    a.policy_cells = h->policy_cells;
    a.start_broadcast = h->start_broadcast; a.goal_broadcast = h->goal_broadcast;
    a.auto_reset = io->step_flags & MAPF_STEP_AUTO_RESET;
    a.accumulate = io->accumulate != 0;
    a.start_terminal_any = h->start_terminal_any;
    a.mv_delta8 = h->mv_delta8;
    if (h->device_ptrs) {
        for (const void *p : {(const void *)io->actions, (const void *)io->out_returns, (const void *)io->out_episodes,
                              (const void *)io->out_collisions, (const void *)io->rec_local, (const void *)io->rec_reward,
                              (const void *)io->rec_done, (const void *)io->rec_collision, (const void *)io->rec_prob})
            if (p && misaligned(p)) return fail(MAPF_EINVAL, "rollout: device pointers must be 16-byte aligned");
        a.actions = io->actions; a.out_returns = io->out_returns; a.out_episodes = io->out_episodes;
        a.out_collisions = io->out_collisions; a.rec_local = io->rec_local; a.rec_reward = io->rec_reward;
        a.rec_done = io->rec_done; a.rec_collision = io->rec_collision; a.rec_prob = io->rec_prob;
        if (int rc = complete_recording(h, a, TE, TEA)) return rc;
        HIP_TRY(h->lane_group_rollout ? mapf::launch_rollout_lg(int(h->A), a, h->tune, h->stream) : mapf::launch_rollout(int(h->A), a, h->stream));
        if (h->last_rollout_kernel != g_noted_kernel) h->last_rollout_kernel = g_noted_kernel;
        if (h->capturing) h->cap_steps += io->n_steps; else h->t += io->n_steps;
        if (io->n_steps) (h->capturing ? h->cap_may_be_terminal : h->may_be_terminal) = a.auto_reset ? h->start_terminal_any : true;
        return MAPF_OK;
    }
    // host-pointer mode: stage everything through device scratch
    if (int rc = stage_in(h, h->s_actions, io->actions, TEA, &a.actions, "actions")) return rc;
    if (io->out_returns) {
        HIP_TRY(h->s_ret.reserve(E * sizeof(double)));
        a.out_returns = static_cast<double *>(h->s_ret.ptr);
        if (a.accumulate) HIP_TRY(hipMemcpyAsync(a.out_returns, io->out_returns, E * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (io->out_episodes) {
        HIP_TRY(h->s_epi.reserve(E * sizeof(uint32_t)));
        a.out_episodes = static_cast<uint32_t *>(h->s_epi.ptr);
        if (a.accumulate) HIP_TRY(hipMemcpyAsync(a.out_episodes, io->out_episodes, E * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    }
    if (io->out_collisions) {
        HIP_TRY(h->s_ncoll.reserve(E * sizeof(uint32_t)));
        a.out_collisions = static_cast<uint32_t *>(h->s_ncoll.ptr);
        if (a.accumulate) HIP_TRY(hipMemcpyAsync(a.out_collisions, io->out_collisions, E * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    }
    if (int rc = stage_out(h, h->s_local, io->rec_local, TEA, &a.rec_local, "rec_local")) return rc;
    if (int rc = stage_out(h, h->s_reward, io->rec_reward, TE, &a.rec_reward, "rec_reward")) return rc;
    if (int rc = stage_out(h, h->s_prob, io->rec_prob, TE, &a.rec_prob, "rec_prob")) return rc;
    if (int rc = stage_out(h, h->s_done, io->rec_done, TE, &a.rec_done, "rec_done")) return rc;
    if (int rc = stage_out(h, h->s_coll, io->rec_collision, TE, &a.rec_collision, "rec_collision")) return rc;
    if (int rc = complete_recording(h, a, TE, TEA)) return rc;
    HIP_TRY(h->lane_group_rollout ? mapf::launch_rollout_lg(int(h->A), a, h->tune, h->stream) : mapf::launch_rollout(int(h->A), a, h->stream));
    if (h->last_rollout_kernel != g_noted_kernel) h->last_rollout_kernel = g_noted_kernel;
    h->t += io->n_steps;
    if (io->n_steps) h->may_be_terminal = a.auto_reset ? h->start_terminal_any : true;
    if (int rc = fetch_out(h, a.out_returns, io->out_returns, E)) return rc;
    if (int rc = fetch_out(h, a.out_episodes, io->out_episodes, E)) return rc;
    if (int rc = fetch_out(h, a.out_collisions, io->out_collisions, E)) return rc;
    if (int rc = fetch_out(h, a.rec_local, io->rec_local, TEA)) return rc;
    if (int rc = fetch_out(h, a.rec_reward, io->rec_reward, TE)) return rc;
    if (int rc = fetch_out(h, a.rec_prob, io->rec_prob, TE)) return rc;
    if (int rc = fetch_out(h, a.rec_done, io->rec_done, TE)) return rc;
    if (int rc = fetch_out(h, a.rec_collision, io->rec_collision, TE)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

int mapf_set_policy(mapf_handle_t h, int policy, const uint32_t *cell_rc) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_set_policy")) return rc;
    if (!h->graphs.empty()) return fail(MAPF_EINVAL, "set_policy: recorded graphs hold the current policy table (destroy them first)");
    if (policy != MAPF_POLICY_RANDOM && policy != MAPF_POLICY_GREEDY) return fail(MAPF_EINVAL, "set_policy: unknown policy");
    HIP_TRY(hipStreamSynchronize(h->stream));   // no launch may still be reading the old table
    if (policy == MAPF_POLICY_RANDOM) {
        if (h->policy_cells) { (void)hipFree(h->policy_cells); h->policy_cells = nullptr; }
        return MAPF_OK;
    }
    if (!cell_rc) return fail(MAPF_EINVAL, "set_policy: the greedy policy needs cell_rc");
    // For every cell and every direction (sign of goal row - row, sign of goal col - col) the first action in
    // ACTIONS order that is not blocked and lands one step closer; whether a move helps is read off the
    // coordinates of its target, so no axis convention is assumed.
    std::vector<uint2> cells(h->V);
    for (uint32_t v = 0; v < h->V; ++v) {
        const int r = int(cell_rc[v] & 0xFFFFu), c = int(cell_rc[v] >> 16);
        uint32_t best = 0;
        for (int sr = -1; sr <= 1; ++sr)
            for (int sc = -1; sc <= 1; ++sc) {
                uint32_t pick = 0;   // STAY
                for (uint32_t a = 1; a < 5 && pick == 0; ++a) {
                    const uint32_t tgt = h->nbr[size_t(v) * 5 + a];
                    if (tgt == v) continue;   // blocked
                    const int dr = int(cell_rc[tgt] & 0xFFFFu) - r, dc = int(cell_rc[tgt] >> 16) - c;
                    if ((std::abs(dr) + std::abs(dc)) != 1)
                        return fail(MAPF_EINVAL, "set_policy: cell_rc does not match the neighbour table (a move must change one coordinate by one)");
                    if ((dr != 0 && dr == sr) || (dc != 0 && dc == sc)) pick = a;
                }
                best |= pick << (3 * (3 * (sr + 1) + (sc + 1)));
            }
        cells[v] = make_uint2(cell_rc[v], best);
    }
    if (!h->policy_cells) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->policy_cells), cells.size() * sizeof(uint2)));
    HIP_TRY(hipMemcpy(h->policy_cells, cells.data(), cells.size() * sizeof(uint2), hipMemcpyHostToDevice));
    return MAPF_OK;
}

int mapf_fill_random_actions(mapf_handle_t h, uint8_t *actions, uint64_t t0, uint32_t n_steps) {
    if (int rc = check_handle(h)) return rc;
    if (!actions) return fail(MAPF_EINVAL, "actions is null");
    const size_t n = size_t(n_steps) * size_t(h->E) * h->A;
    uint8_t *d_actions = nullptr;
    if (int rc = stage_out(h, h->s_actions, actions, n, &d_actions, "actions")) return rc;
    HIP_TRY(mapf::launch_fill_actions(int(h->A), d_actions, h->c, h->env_id_offset, h->E, t0, n_steps, h->stream));
    if (int rc = fetch_out(h, d_actions, actions, n)) return rc;
    if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

namespace {
// the scan's scratch (handle-owned; a growing buffer is reallocated behind hipFree's implicit device synchronisation)
int transitions_scratch(mapf_handle_t h, mapf::TransitionsArgs &a) {
    HIP_TRY(h->q_rel.reserve((a.n_queries ? a.n_queries : 1) * sizeof(uint32_t)));
    HIP_TRY(h->q_blocks.reserve((mapf::transitions_scan_blocks(a.n_queries) + 1) * sizeof(uint64_t)));
    a.rel = static_cast<uint32_t *>(h->q_rel.ptr);
    a.block_base = static_cast<uint64_t *>(h->q_blocks.ptr);
    return MAPF_OK;
}
}  // namespace

int mapf_transitions_window(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                            const uint32_t *env_index, uint64_t first_branch, uint32_t max_branches, uint32_t *out_count, uint16_t *out_next,
                     double *out_prob, double *out_reward, uint8_t *out_done, uint8_t *out_collision) {
    if (int rc = check_handle(h)) return rc;
    if (!local || !actions) return fail(MAPF_EINVAL, "local / actions are null");
    if (max_branches == 0) return fail(MAPF_EINVAL, "max_branches must be >= 1");
    if (h->A > uint32_t(mapf::kTransitionsMaxAgents)) return fail(MAPF_EUNSUPPORTED, "mapf_transitions supports n_agents <= 16 (3^A branches per query)");
    const size_t N = size_t(n_queries), NA = N * h->A, NM = N * max_branches;
    if (!h->device_ptrs) {
        for (size_t i = 0; i < NA; ++i) if (local[i] >= h->V) return fail(MAPF_EINVAL, "transitions: cell out of range");
        if (env_index) for (size_t i = 0; i < N; ++i) if (env_index[i] >= h->E) return fail(MAPF_EINVAL, "transitions: env_index out of range");
    }
    mapf::TransitionsArgs a{};
    a.c = h->c; a.mv = h->mv; a.slip = h->slip; a.goal = h->goal; a.goal_broadcast = h->goal_broadcast;
    a.n_queries = n_queries; a.max_branches = max_branches; a.n_agents = h->A; a.first_branch = first_branch;
    a.capacity = ~uint64_t(0);
    if (int rc = stage_in(h, h->q_local, local, NA, &a.local, "local")) return rc;
    if (int rc = stage_in(h, h->q_actions, actions, NA, &a.actions, "actions")) return rc;
    if (int rc = stage_in(h, h->q_env, env_index, N, &a.env_index, "env_index")) return rc;
    if (int rc = stage_out(h, h->q_count, out_count, N, &a.out_count, "out_count")) return rc;
    if (int rc = stage_out(h, h->q_next, out_next, NM * h->A, &a.out_next, "out_next")) return rc;
    if (int rc = stage_out(h, h->q_prob, out_prob, NM, &a.out_prob, "out_prob")) return rc;
    if (int rc = stage_out(h, h->q_reward, out_reward, NM, &a.out_reward, "out_reward")) return rc;
    if (int rc = stage_out(h, h->q_done, out_done, NM, &a.out_done, "out_done")) return rc;
    if (int rc = stage_out(h, h->q_coll, out_collision, NM, &a.out_collision, "out_collision")) return rc;
    if (int rc = transitions_scratch(h, a)) return rc;
    HIP_TRY(mapf::launch_transitions(a, h->stream));
    if (h->last_transitions_kernel != g_noted_kernel) h->last_transitions_kernel = g_noted_kernel;
    if (int rc = fetch_out(h, a.out_count, out_count, N)) return rc;
    if (int rc = fetch_out(h, a.out_next, out_next, NM * h->A)) return rc;
    if (int rc = fetch_out(h, a.out_prob, out_prob, NM)) return rc;
    if (int rc = fetch_out(h, a.out_reward, out_reward, NM)) return rc;
    if (int rc = fetch_out(h, a.out_done, out_done, NM)) return rc;
    if (int rc = fetch_out(h, a.out_collision, out_collision, NM)) return rc;
    if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

int mapf_transitions(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                     const uint32_t *env_index, uint32_t max_branches, uint32_t *out_count, uint16_t *out_next,
                     double *out_prob, double *out_reward, uint8_t *out_done, uint8_t *out_collision) {
    return mapf_transitions_window(h, n_queries, local, actions, env_index, 0, max_branches, out_count, out_next, out_prob,
                                   out_reward, out_done, out_collision);
}

int mapf_transitions_compact(mapf_handle_t h, uint64_t n_queries, const uint16_t *local, const uint8_t *actions,
                             const uint32_t *env_index, uint64_t first_branch, uint32_t max_branches, uint64_t capacity_rows,
                             uint64_t *out_offset, uint32_t *out_count, uint16_t *out_next, double *out_prob, double *out_reward,
                             uint8_t *out_done, uint8_t *out_collision) {
    if (int rc = check_handle(h)) return rc;
    if (!local || !actions) return fail(MAPF_EINVAL, "local / actions are null");
    if (!out_offset) return fail(MAPF_EINVAL, "transitions_compact: out_offset (u64[n_queries + 1]) is required");
    if (max_branches == 0) return fail(MAPF_EINVAL, "max_branches must be >= 1");
    if (h->A > uint32_t(mapf::kTransitionsMaxAgents)) return fail(MAPF_EUNSUPPORTED, "mapf_transitions supports n_agents <= 16 (3^A branches per query)");
    const size_t N = size_t(n_queries), NA = N * h->A, R = size_t(capacity_rows);
    if (!h->device_ptrs) {
        for (size_t i = 0; i < NA; ++i) if (local[i] >= h->V) return fail(MAPF_EINVAL, "transitions: cell out of range");
        if (env_index) for (size_t i = 0; i < N; ++i) if (env_index[i] >= h->E) return fail(MAPF_EINVAL, "transitions: env_index out of range");
    }
    mapf::TransitionsArgs a{};
    a.c = h->c; a.mv = h->mv; a.slip = h->slip; a.goal = h->goal; a.goal_broadcast = h->goal_broadcast;
    a.n_queries = n_queries; a.max_branches = max_branches; a.n_agents = h->A; a.first_branch = first_branch;
    a.capacity = capacity_rows;
    if (int rc = stage_in(h, h->q_local, local, NA, &a.local, "local")) return rc;
    if (int rc = stage_in(h, h->q_actions, actions, NA, &a.actions, "actions")) return rc;
    if (int rc = stage_in(h, h->q_env, env_index, N, &a.env_index, "env_index")) return rc;
    if (int rc = stage_out(h, h->q_offset, out_offset, N + 1, &a.out_offset, "out_offset")) return rc;
    if (int rc = stage_out(h, h->q_count, out_count, N, &a.out_count, "out_count")) return rc;
    if (int rc = stage_out(h, h->q_next, out_next, R * h->A, &a.out_next, "out_next")) return rc;
    if (int rc = stage_out(h, h->q_prob, out_prob, R, &a.out_prob, "out_prob")) return rc;
    if (int rc = stage_out(h, h->q_reward, out_reward, R, &a.out_reward, "out_reward")) return rc;
    if (int rc = stage_out(h, h->q_done, out_done, R, &a.out_done, "out_done")) return rc;
    if (int rc = stage_out(h, h->q_coll, out_collision, R, &a.out_collision, "out_collision")) return rc;
    if (int rc = transitions_scratch(h, a)) return rc;
    a.compact = true;
    if (N == 0) HIP_TRY(hipMemsetAsync(a.out_offset, 0, sizeof(uint64_t), h->stream));
    HIP_TRY(mapf::launch_transitions(a, h->stream));
    if (h->last_transitions_kernel != g_noted_kernel) h->last_transitions_kernel = g_noted_kernel;
    if (!h->device_ptrs) {
        // host arrays: the offsets first -- only the rows that exist (and fit) are copied back
        HIP_TRY(hipMemcpyAsync(out_offset, a.out_offset, (N + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const size_t rows = size_t(std::min<uint64_t>(out_offset[N], capacity_rows));
        if (int rc = fetch_out(h, a.out_count, out_count, N)) return rc;
        if (int rc = fetch_out(h, a.out_next, out_next, rows * h->A)) return rc;
        if (int rc = fetch_out(h, a.out_prob, out_prob, rows)) return rc;
        if (int rc = fetch_out(h, a.out_reward, out_reward, rows)) return rc;
        if (int rc = fetch_out(h, a.out_done, out_done, rows)) return rc;
        if (int rc = fetch_out(h, a.out_collision, out_collision, rows)) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return MAPF_OK;
}

int mapf_transition_rewards(mapf_handle_t h, uint64_t n_queries, const uint16_t *prev_local, const uint8_t *actions,
                            const uint16_t *next_local, const uint32_t *env_index, double *out_reward, uint8_t *out_done,
                            uint8_t *out_collision) {
    if (int rc = check_handle(h)) return rc;
    if (!prev_local || !actions || !next_local) return fail(MAPF_EINVAL, "prev_local / actions / next_local are null");
    const size_t N = size_t(n_queries), NA = N * h->A;
    if (!h->device_ptrs) {
        for (size_t i = 0; i < NA; ++i)
            if (prev_local[i] >= h->V || next_local[i] >= h->V) return fail(MAPF_EINVAL, "transition_rewards: cell out of range");
        if (env_index) for (size_t i = 0; i < N; ++i) if (env_index[i] >= h->E) return fail(MAPF_EINVAL, "transition_rewards: env_index out of range");
    }
    mapf::TransitionsArgs a{};
    a.c = h->c; a.mv = h->mv; a.slip = h->slip; a.goal = h->goal; a.goal_broadcast = h->goal_broadcast;
    a.n_queries = n_queries; a.max_branches = 1; a.n_agents = h->A; a.capacity = ~uint64_t(0);
    const uint16_t *d_next = nullptr;
    if (int rc = stage_in(h, h->q_local, prev_local, NA, &a.local, "prev_local")) return rc;
    if (int rc = stage_in(h, h->q_actions, actions, NA, &a.actions, "actions")) return rc;
    if (int rc = stage_in(h, h->q_next_in, next_local, NA, &d_next, "next_local")) return rc;
    if (int rc = stage_in(h, h->q_env, env_index, N, &a.env_index, "env_index")) return rc;
    if (int rc = stage_out(h, h->q_reward, out_reward, N, &a.out_reward, "out_reward")) return rc;
    if (int rc = stage_out(h, h->q_done, out_done, N, &a.out_done, "out_done")) return rc;
    if (int rc = stage_out(h, h->q_coll, out_collision, N, &a.out_collision, "out_collision")) return rc;
    HIP_TRY(mapf::launch_transition_rewards(a, d_next, h->stream));
    if (int rc = fetch_out(h, a.out_reward, out_reward, N)) return rc;
    if (int rc = fetch_out(h, a.out_done, out_done, N)) return rc;
    if (int rc = fetch_out(h, a.out_collision, out_collision, N)) return rc;
    if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

int mapf_query_terminal(mapf_handle_t h, uint8_t *out_terminal) {
    if (int rc = check_handle(h)) return rc;
    if (!out_terminal) return fail(MAPF_EINVAL, "out_terminal is null");
    uint8_t *d_out = nullptr;
    if (int rc = stage_out(h, h->s_term, out_terminal, size_t(h->E), &d_out, "out_terminal")) return rc;
    HIP_TRY(mapf::launch_query_terminal(int(h->A), h->state, h->goal, h->goal_broadcast, d_out, h->E, h->stream));
    if (int rc = fetch_out(h, d_out, out_terminal, size_t(h->E))) return rc;
    if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    return MAPF_OK;
}

int mapf_get_state(mapf_handle_t h, uint16_t *local, uint64_t *t) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_get_state")) return rc;
    if (t) *t = h->t;
    if (local) {
        const size_t bytes = size_t(h->E) * h->A * sizeof(uint16_t);
        HIP_TRY(hipMemcpyAsync(local, h->state, bytes, h->device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return MAPF_OK;
}

int mapf_set_state(mapf_handle_t h, const uint16_t *local, uint64_t t) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_set_state")) return rc;
    if (local) {
        const size_t n = size_t(h->E) * h->A;
        if (!h->device_ptrs) {
            for (size_t i = 0; i < n; ++i)
                if (local[i] >= h->V) return fail(MAPF_EINVAL, "set_state: cell out of range");
        }
        HIP_TRY(hipMemcpyAsync(h->state, local, n * sizeof(uint16_t),
                               h->device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        h->may_be_terminal = true;   // (an arbitrary state)
        if (!h->device_ptrs) HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->t = t;
    return MAPF_OK;
}

int mapf_state_view(mapf_handle_t h, const uint16_t **out_state) {
    if (int rc = check_handle(h)) return rc;
    if (!out_state) return fail(MAPF_EINVAL, "out_state is null");
    *out_state = h->state;
    return MAPF_OK;
}

int mapf_invalidate_state(mapf_handle_t h) {
    if (int rc = check_handle(h)) return rc;
    h->may_be_terminal = true;
    h->cap_may_be_terminal = true;
    for (mapf_graph_s *g : h->graphs) g->ends_may_be_terminal = true;   // (conservative: a replay may run over edited state too)
    return MAPF_OK;
}

int mapf_graph_begin(mapf_handle_t h) {
    if (int rc = check_handle(h)) return rc;
    if (!h->device_ptrs) return fail(MAPF_EINVAL, "graph_begin: only handles created with MAPF_FLAG_DEVICE_PTRS can be recorded (host-pointer calls wait for the stream)");
    if (h->capturing) return fail(MAPF_EINVAL, "graph_begin: already recording");
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
    h->capturing = true;
    h->cap_steps = 0;
    h->cap_may_be_terminal = true;   // whatever precedes a replay: the first recorded step tests is_terminal itself
    return MAPF_OK;
}

int mapf_graph_end(mapf_handle_t h, mapf_graph_t *out_graph) {
    if (int rc = check_handle(h)) return rc;
    if (!h->capturing) return fail(MAPF_EINVAL, "graph_end: not recording");
    hipError_t adv = hipSuccess;
    if (out_graph && h->cap_steps) adv = mapf::launch_advance_step_index(h->t_dev, h->cap_steps, h->stream);   // the recording's last node
    hipGraph_t graph = nullptr;
    const hipError_t end = hipStreamEndCapture(h->stream, &graph);
    h->capturing = false;
    if (out_graph) *out_graph = nullptr;
    if (adv != hipSuccess || end != hipSuccess || !graph || !out_graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (!out_graph) return fail(MAPF_EINVAL, "graph_end: out_graph is null (the recording was dropped)");
        return hip_fail(adv != hipSuccess ? adv : (end != hipSuccess ? end : hipErrorUnknown), "graph_end: the recording failed");
    }
    mapf_graph_t g = new (std::nothrow) mapf_graph_s();
    if (!g) { (void)hipGraphDestroy(graph); return fail(MAPF_EHIP, "out of host memory"); }
    g->owner = h; g->graph = graph; g->steps = h->cap_steps;
    g->ends_may_be_terminal = h->cap_may_be_terminal;
    const hipError_t inst = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (inst != hipSuccess) { (void)hipGraphDestroy(graph); delete g; return hip_fail(inst, "hipGraphInstantiate"); }
    h->graphs.push_back(g);
    *out_graph = g;
    return MAPF_OK;
}

int mapf_graph_launch(mapf_handle_t h, mapf_graph_t g, uint32_t n_replays) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_not_recording(h, "mapf_graph_launch")) return rc;
    if (!g || std::find(h->graphs.begin(), h->graphs.end(), g) == h->graphs.end()) return fail(MAPF_EINVAL, "graph_launch: not a live graph of this handle");
    // the device-side index must hold the handle's step index when the first recorded launch reads it
    if (h->t_dev_value != h->t) {
        HIP_TRY(mapf::launch_set_step_index(h->t_dev, h->t, h->stream));
        h->t_dev_value = h->t;
    }
    for (uint32_t r = 0; r < n_replays; ++r) HIP_TRY(hipGraphLaunch(g->exec, h->stream));
    h->t += uint64_t(n_replays) * g->steps;
    h->t_dev_value = h->t;
    if (n_replays) h->may_be_terminal = g->ends_may_be_terminal;
    return MAPF_OK;
}

int mapf_graph_steps(mapf_graph_t g, uint64_t *out_steps) {
    if (!g || !out_steps) return fail(MAPF_EINVAL, "null graph or output");
    *out_steps = g->steps;
    return MAPF_OK;
}

int mapf_graph_destroy(mapf_handle_t h, mapf_graph_t g) {
    if (int rc = check_handle(h)) return rc;
    const auto it = g ? std::find(h->graphs.begin(), h->graphs.end(), g) : h->graphs.end();
    if (it == h->graphs.end()) return fail(MAPF_EINVAL, "graph_destroy: not a live graph of this handle");
    HIP_TRY(hipStreamSynchronize(h->stream));   // no replay may still be running
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    h->graphs.erase(it);
    delete g;
    return MAPF_OK;
}

}  // extern "C"
