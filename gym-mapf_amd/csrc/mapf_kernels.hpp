// Argument blocks and launcher prototypes shared by mapf_kernels.hip and mapf_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mapf {

struct EnvConsts {
    double p0, rf, lf;             // 1 - rf - lf, right_fail, left_fail (host-computed; mapf_env.py:131-132, :167-169)
    double r_clash, r_goal, r_living;
    uint32_t keep;                 // bit k set <=> slip candidate k has p > 0 (mapf_env.py:172)
    uint32_t criteria;             // 0 Makespan, 1 SoC
    uint32_t n_cells;              // V
    uint32_t seed_lo, seed_hi;     // slip-stream Philox key
    uint32_t pol_lo, pol_hi;       // policy-stream Philox key (seed + 1)
};

struct StepArgs {
    EnvConsts c;
    const uint2 *nbr4;             // [V] {up | right << 16, down | left << 16}
    uint16_t *state;               // [E*A] persistent env state
    const uint16_t *start, *goal;  // [E*A] or [A]
    const uint8_t *actions;        // [E*A]
    const double *uniforms;        // [E*A] or null
    uint16_t *out_local;
    double *out_reward, *out_prob;
    uint8_t *out_done, *out_collision, *out_was_terminal;
    uint64_t n_envs, env_id_offset, t;
    bool start_broadcast, goal_broadcast, auto_reset;
};

struct RolloutArgs {
    EnvConsts c;
    const uint2 *nbr4;
    uint16_t *state;
    const uint16_t *start, *goal;
    const uint8_t *actions;        // [T*E*A] or null (policy stream)
    double *out_returns;
    uint32_t *out_episodes, *out_collisions;
    uint16_t *rec_local;
    double *rec_reward, *rec_prob;
    uint8_t *rec_done, *rec_collision;
    uint64_t n_envs, env_id_offset, t;
    uint32_t n_steps;
    bool start_broadcast, goal_broadcast, auto_reset, accumulate;
};

// routed by agent count (mapf_dispatch.hip)
hipError_t launch_step(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_query_terminal(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);

// per-group entry points: group g holds the kernels specialised for A in 4g+1 .. 4g+4
hipError_t launch_step_g0(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g0(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g0(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g0(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g1(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g1(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g1(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g1(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g2(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g2(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g2(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g2(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g3(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g3(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g3(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g3(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g4(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g4(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g4(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g4(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g5(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g5(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g5(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g5(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g6(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g6(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g6(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g6(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_step_g7(int n_agents, const StepArgs &args, hipStream_t stream);
hipError_t launch_rollout_g7(int n_agents, const RolloutArgs &args, hipStream_t stream);
hipError_t launch_reset_g7(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream);
hipError_t launch_fill_actions_g7(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream);

hipError_t launch_query_terminal_g0(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g1(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g2(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g3(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g4(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g5(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g6(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
hipError_t launch_query_terminal_g7(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream);
}  // namespace mapf
