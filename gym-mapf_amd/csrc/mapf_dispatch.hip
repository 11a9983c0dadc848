// Routes a launch to the translation unit that holds the kernels specialised for its agent count.
#include "mapf_kernels.hpp"

namespace mapf {

#define MAPF_ROUTE(fn, A, ...)                          \
    switch (((A) - 1) / 4) {                            \
        case 0: return fn##_g0(A, __VA_ARGS__);         \
        case 1: return fn##_g1(A, __VA_ARGS__);         \
        case 2: return fn##_g2(A, __VA_ARGS__);         \
        case 3: return fn##_g3(A, __VA_ARGS__);         \
        case 4: return fn##_g4(A, __VA_ARGS__);         \
        case 5: return fn##_g5(A, __VA_ARGS__);         \
        case 6: return fn##_g6(A, __VA_ARGS__);         \
        case 7: return fn##_g7(A, __VA_ARGS__);         \
        default: return hipErrorInvalidValue;           \
    }

hipError_t launch_step(int n_agents, const StepArgs &args, hipStream_t stream) {
    if (n_agents < 1) return hipErrorInvalidValue;
    MAPF_ROUTE(launch_step, n_agents, args, stream)
}

hipError_t launch_rollout(int n_agents, const RolloutArgs &args, hipStream_t stream) {
    if (n_agents < 1) return hipErrorInvalidValue;
    MAPF_ROUTE(launch_rollout, n_agents, args, stream)
}

hipError_t launch_reset(int n_agents, uint16_t *state, const uint16_t *start, bool start_broadcast,
                        const uint8_t *mask, uint64_t n_envs, hipStream_t stream) {
    if (n_agents < 1) return hipErrorInvalidValue;
    MAPF_ROUTE(launch_reset, n_agents, state, start, start_broadcast, mask, n_envs, stream)
}

hipError_t launch_query_terminal(int n_agents, const uint16_t *state, const uint16_t *goal, bool goal_broadcast,
                                 uint8_t *out, uint64_t n_envs, hipStream_t stream) {
    if (n_agents < 1) return hipErrorInvalidValue;
    MAPF_ROUTE(launch_query_terminal, n_agents, state, goal, goal_broadcast, out, n_envs, stream)
}

hipError_t launch_fill_actions(int n_agents, uint8_t *actions, const EnvConsts &c, uint64_t env_id_offset,
                               uint64_t n_envs, uint64_t t0, uint64_t n_steps, hipStream_t stream) {
    if (n_agents < 1) return hipErrorInvalidValue;
    MAPF_ROUTE(launch_fill_actions, n_agents, actions, c, env_id_offset, n_envs, t0, n_steps, stream)
}

}  // namespace mapf
