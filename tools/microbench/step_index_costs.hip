// What a capture-safe, device-side step index costs a launch-bound single step (gfx950), and what a hipGraph replay
// saves: one small kernel (each thread reads 8 B and writes 8 B; G blocks of 256) is run back to back
//   (a) as plain stream launches, step index a kernel argument                      [plain]
//   (b) the same launches captured once into a hipGraph of NODES kernel nodes        [graph]
//   (c) graph, step index read from device memory, advanced by a one-thread kernel node after every step [graph+advance]
//   (d) graph, advanced by the kernel itself: the last block to finish (ticket = returning atomic per block) [graph+ticket]
//   (e) graph, every block draws its index with a returning atomic at its START (block 0 adds the complement to 2^20,
//       so old >> 20 is the step index whatever the order)                           [graph+startatomic]
//   (f) graph, one counter per block, read at the start and written back + 1 at the end [graph+perblock]
// Prints microseconds per step (HIP events over REPLAYS x NODES steps) for G = 64, 512, 2048, 8192 blocks.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/step_index_costs.hip -o /tmp/sic && /tmp/sic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
    const uint64_t *in;
    uint64_t *out;
    unsigned long long *t_dev;      // device step index (forms c, d, e)
    unsigned *ticket;               // form d
    unsigned long long *per_block;  // form f
    uint64_t t;                     // form a, b
    int mode;
};

__global__ void __launch_bounds__(256) step(Args a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ unsigned long long t_lds;
    unsigned long long t = a.t;
    if (a.mode == 2 || a.mode == 3) t = *(volatile unsigned long long *)a.t_dev;
    if (a.mode == 4) {
        if (threadIdx.x == 0) {
            const unsigned long long add = blockIdx.x == 0 ? (1ull << 20) - (gridDim.x - 1) : 1ull;
            t_lds = atomicAdd(a.t_dev, add) >> 20;
        }
        __syncthreads();
        t = t_lds;
    }
    if (a.mode == 5) t = a.per_block[blockIdx.x];
    const uint64_t v = a.in[i];
    a.out[i] = v * 0x9E3779B97F4A7C15ull + t;
    if (a.mode == 3) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned old = atomicAdd(a.ticket, 1u);
            if (old == gridDim.x - 1) {
                *a.ticket = 0u;
                *(volatile unsigned long long *)a.t_dev = t + 1;
            }
        }
    }
    if (a.mode == 5 && threadIdx.x == 0) a.per_block[blockIdx.x] = t + 1;
}

__global__ void advance(unsigned long long *t_dev) { *t_dev += 1; }

int main() {
    const int NODES = 64, REPLAYS = 40;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {64, 512, 2048, 8192};
    const char *names[] = {"plain", "graph", "graph+advance", "graph+ticket", "graph+startatomic", "graph+perblock"};
    printf("%-8s", "blocks");
    for (const char *n : names) printf(" %18s", n);
    printf("   (us per step)\n");
    for (int G : grids) {
        const size_t n = size_t(G) * 256;
        uint64_t *in, *out;
        unsigned long long *t_dev, *per_block;
        unsigned *ticket;
        CHECK(hipMalloc(&in, n * 8));
        CHECK(hipMalloc(&out, n * 8));
        CHECK(hipMalloc(&t_dev, 8));
        CHECK(hipMalloc(&ticket, 4));
        CHECK(hipMalloc(&per_block, size_t(G) * 8));
        CHECK(hipMemset(in, 1, n * 8));
        CHECK(hipMemset(t_dev, 0, 8));
        CHECK(hipMemset(ticket, 0, 4));
        CHECK(hipMemset(per_block, 0, size_t(G) * 8));
        printf("%-8d", G);
        for (int mode = 0; mode < 6; ++mode) {
            Args a{in, out, t_dev, ticket, per_block, 0, mode};
            float ms = 0.f;
            if (mode == 0) {
                for (int w = 0; w < 200; ++w) { a.t = w; hipLaunchKernelGGL(step, dim3(G), dim3(256), 0, s, a); }
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventRecord(e0, s));
                for (int r = 0; r < REPLAYS * NODES; ++r) { a.t = r; hipLaunchKernelGGL(step, dim3(G), dim3(256), 0, s, a); }
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            } else {
                hipGraph_t graph;
                hipGraphExec_t exec;
                CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < NODES; ++k) {
                    a.t = k;
                    hipLaunchKernelGGL(step, dim3(G), dim3(256), 0, s, a);
                    if (mode == 2) hipLaunchKernelGGL(advance, dim3(1), dim3(1), 0, s, t_dev);
                }
                CHECK(hipStreamEndCapture(s, &graph));
                CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                for (int w = 0; w < 4; ++w) CHECK(hipGraphLaunch(exec, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventRecord(e0, s));
                for (int r = 0; r < REPLAYS; ++r) CHECK(hipGraphLaunch(exec, s));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipGraphExecDestroy(exec));
                CHECK(hipGraphDestroy(graph));
                if (mode == 2 || mode == 3 || mode == 4) {   // the counter must have advanced once per step
                    unsigned long long t_host = 0;
                    CHECK(hipMemcpy(&t_host, t_dev, 8, hipMemcpyDeviceToHost));
                    const unsigned long long steps = mode == 4 ? t_host >> 20 : t_host;
                    if (steps != (unsigned long long)(REPLAYS + 4) * NODES) printf("[mode %d: counter %llu != %d] ", mode, steps, (REPLAYS + 4) * NODES);
                    CHECK(hipMemset(t_dev, 0, 8));
                }
            }
            printf(" %18.3f", ms * 1e3 / (REPLAYS * NODES));
            fflush(stdout);
        }
        printf("\n");
        CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(t_dev)); CHECK(hipFree(ticket)); CHECK(hipFree(per_block));
    }
    return 0;
}
