// Routes a launch to the translation unit that holds the kernels specialised for its agent count.
#include "mapf_kernels.hpp"

namespace mapf {

#define MAPF_ROUTE(fn, A, ...)                          \
    switch (((A) - 1) / 4) {                            \
        case 0: return fn##_g0(A, __VA_ARGS__);         \
        case 1: return fn##_g1(A, __VA_ARGS__);         \
        case 2: return fn##_g2(A, __VA_ARGS__);         \
        case 3: return fn##_g3(A, __VA_ARGS__);         \
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

}  // namespace mapf
