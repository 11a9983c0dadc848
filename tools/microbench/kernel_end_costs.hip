// What the END of a small kernel costs in a back-to-back chain (hipGraph of 64 nodes), as a function of how its results
// are stored: the step kernel's waves are gone after ~2.6 us, yet a launch takes ~4.6 us (profiles/r03_step_stamps_*).
// Each thread reads W x 8 B and writes W x 8 B (W = 1, 4); store forms:
//   plain        global_store (dirty lines stay in the XCD's L2 until the end-of-kernel release writes them back)
//   nontemporal  __builtin_nontemporal_store (nt)
//   system       system-scope relaxed atomic store (sc0 sc1: write-through)
//   none         no store at all (the floor: dispatch + loads)
// and the loads either plain or nontemporal.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/kernel_end_costs.hip -o /tmp/kec && /tmp/kec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int W, int STORE, bool NT_LOAD>
__global__ void __launch_bounds__(256) step(const uint64_t *in, uint64_t *out, uint64_t t) {
    const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * W;
    uint64_t v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = NT_LOAD ? __builtin_nontemporal_load(in + i + k) : in[i + k];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint64_t r = v[k] * 0x9E3779B97F4A7C15ull + t;
        if (STORE == 0) out[i + k] = r;
        else if (STORE == 1) __builtin_nontemporal_store(r, out + i + k);
        else if (STORE == 2) __hip_atomic_store(out + i + k, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (r == 0x1234567ull) out[i + k] = r;   // (never true in practice: keeps the loads alive)
    }
}

template <int W, int STORE, bool NT_LOAD>
float run(int G, hipStream_t s, const uint64_t *a, uint64_t *b, hipEvent_t e0, hipEvent_t e1) {
    const int NODES = 64, REPLAYS = 40;
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < NODES; ++k) {   // ping-pong: every node reads what the node before it wrote
        if (k & 1) hipLaunchKernelGGL((step<W, STORE, NT_LOAD>), dim3(G), dim3(256), 0, s, b, const_cast<uint64_t *>(a), uint64_t(k));
        else hipLaunchKernelGGL((step<W, STORE, NT_LOAD>), dim3(G), dim3(256), 0, s, a, b, uint64_t(k));
    }
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int w = 0; w < 4; ++w) CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < REPLAYS; ++r) CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    return ms * 1e3f / (REPLAYS * NODES);
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-8s %-4s %12s %12s %12s %12s %14s %14s   (us per node)\n", "blocks", "W", "plain", "nontemporal", "system", "none", "nt-load+plain", "nt-load+nt");
    for (int G : {512, 2048, 8192}) {
        uint64_t *a, *b;
        const size_t n = size_t(G) * 256 * 4;
        CHECK(hipMalloc(&a, n * 8));
        CHECK(hipMalloc(&b, n * 8));
        CHECK(hipMemset(a, 1, n * 8));
        CHECK(hipMemset(b, 2, n * 8));
        printf("%-8d %-4d %12.3f %12.3f %12.3f %12.3f %14.3f %14.3f\n", G, 1, run<1, 0, false>(G, s, a, b, e0, e1), run<1, 1, false>(G, s, a, b, e0, e1),
               run<1, 2, false>(G, s, a, b, e0, e1), run<1, 3, false>(G, s, a, b, e0, e1), run<1, 0, true>(G, s, a, b, e0, e1), run<1, 1, true>(G, s, a, b, e0, e1));
        printf("%-8d %-4d %12.3f %12.3f %12.3f %12.3f %14.3f %14.3f\n", G, 4, run<4, 0, false>(G, s, a, b, e0, e1), run<4, 1, false>(G, s, a, b, e0, e1),
               run<4, 2, false>(G, s, a, b, e0, e1), run<4, 3, false>(G, s, a, b, e0, e1), run<4, 0, true>(G, s, a, b, e0, e1), run<4, 1, true>(G, s, a, b, e0, e1));
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    return 0;
}
