// Cost of LDS atomics against plain LDS accesses (gfx950): cycles per wave-instruction when every lane of a wave hits a
// random word of a small per-wave region (8 regions of 104 words -- the per-env occupancy bitmaps of the 32-agent
// rollout), for 1, 2 and 4 waves per SIMD on one CU.  OP: 0 ds_read_b32, 1 ds_write_b32, 2 ds_or_b32 (no return),
// 3 ds_or_rtn_b32 (returning).  Each timed loop issues 8 independent operations and waits for them.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_atomic_cost.hip -o /tmp/lac && /tmp/lac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using lds_u32 = __attribute__((address_space(3))) unsigned *;

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, unsigned seed) {
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned x = (threadIdx.x + 1) * 2654435761u ^ seed;
    unsigned acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < 256; ++it) {
        unsigned addr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            addr[j] = (wave * 832u + (lane >> 3) * 104u + ((x >> 10) % 104u)) * 4u;   // my "env"'s region (8 lanes share one)
        }
        unsigned r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lds_u32 p = (lds_u32)(uintptr_t(addr[j]));
            if (OP == 0) r[j] = *p;
            else if (OP == 1) { *p = x; r[j] = 0; }
            else if (OP == 2) { __hip_atomic_fetch_or(p, 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); r[j] = 0; }
            else r[j] = __hip_atomic_fetch_or(p, 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= r[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) out[wave] = (t1 - t0) | (acc == 0x12345u ? 1ull << 63 : 0);
}

int main() {
    unsigned long long *out;
    CHECK(hipMalloc(&out, 16 * 8));
    const char *names[] = {"ds_read_b32", "ds_write_b32", "ds_or_b32", "ds_or_rtn_b32"};
    printf("%-16s %18s %18s %18s   (cycles per wave-instruction incl. the wait; 8 independent ops per wait)\n", "op", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (int op = 0; op < 4; ++op) {
        printf("%-16s", names[op]);
        for (int waves : {4, 8, 16}) {
            unsigned long long h[16];
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(waves * 64), 64 * 1024, 0, out, 7u);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(waves * 64), 64 * 1024, 0, out, 7u);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(waves * 64), 64 * 1024, 0, out, 7u);
                if (op == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(waves * 64), 64 * 1024, 0, out, 7u);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h, out, waves * 8, hipMemcpyDeviceToHost));
            double s = 0;
            for (int w = 0; w < waves; ++w) s += double(h[w] & ~(1ull << 63));
            printf(" %18.1f", s / waves / (256.0 * 8.0));
        }
        printf("\n");
    }
    return 0;
}
