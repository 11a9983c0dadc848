// instruction throughput microbenchmark: 8 waves/SIMD worth of blocks, 4 independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b * 3u;
    uint64_t A = a | (uint64_t(b) << 32), B = c | (uint64_t(d) << 32), C = A ^ 0x1234567ull, D = B + 99;
    double fa = double(a) * 1e-9 + 1.0, fb = 1.0000001, fc = 0.9999999, fd = 1.0000002;
    for (int i = 0; i < N_ITER; ++i) {
        if (OP == 0) { a = a * 0xD2511F53u + 1u; b = b * 0xD2511F53u + 1u; c = c * 0xD2511F53u + 1u; d = d * 0xD2511F53u + 1u; }          // v_mul_lo (mad_u32?)
        if (OP == 1) { a = __umulhi(a, 0xD2511F53u) ^ 5u; b = __umulhi(b, 0xD2511F53u) ^ 5u; c = __umulhi(c, 0xD2511F53u) ^ 5u; d = __umulhi(d, 0xD2511F53u) ^ 5u; }
        if (OP == 2) { A = uint64_t(uint32_t(A)) * 0xD2511F53u + (A >> 32); B = uint64_t(uint32_t(B)) * 0xD2511F53u + (B >> 32); C = uint64_t(uint32_t(C)) * 0xD2511F53u + (C >> 32); D = uint64_t(uint32_t(D)) * 0xD2511F53u + (D >> 32); }   // mad_u64_u32
        if (OP == 3) { a = (a ^ b) + 7u; b = (b ^ c) + 7u; c = (c ^ d) + 7u; d = (d ^ a) + 7u; }    // 8 simple VALU
        if (OP == 4) { fa = fa * fb; fb = fb * fc; fc = fc * fd; fd = fd * fa; }                      // v_mul_f64
        if (OP == 5) { fa = fa + fb; fb = fb + fc; fc = fc + fd; fd = fd + fa; }                      // v_add_f64
        if (OP == 6) { a += (A < B) ? 1u : 2u; A += 3; b += (B < C) ? 1u : 2u; B += 5; c += (C < D) ? 1u : 2u; C += 7; d += (D < A) ? 1u : 2u; D += 9; }  // cmp_lt_u64 + cndmask + add64
        if (OP == 7) { a = __builtin_amdgcn_update_dpp(0, a, 0x39, 0xF, 0xF, false) + 1u; b = __builtin_amdgcn_update_dpp(0, b, 0x39, 0xF, 0xF, false) + 1u; c = __builtin_amdgcn_update_dpp(0, c, 0x39, 0xF, 0xF, false) + 1u; d = __builtin_amdgcn_update_dpp(0, d, 0x39, 0xF, 0xF, false) + 1u; }
        if (OP == 8) { a = __shfl(a, (threadIdx.x + 1) & 63) + 1u; b = __shfl(b, (threadIdx.x + 1) & 63) + 1u; c = __shfl(c, (threadIdx.x + 1) & 63) + 1u; d = __shfl(d, (threadIdx.x + 1) & 63) + 1u; }
        if (OP == 9) { A = (A >> (a & 31)) + 1; B = (B >> (b & 31)) + 1; C = (C >> (c & 31)) + 1; D = (D >> (d & 31)) + 1; }   // lshr_b64
        if (OP == 10) { a = min(a ^ b, min(c, d)) + 1u; b = min(b ^ c, min(d, a)) + 1u; c = min(c ^ d, min(a, b)) + 1u; d = min(d ^ a, min(b, c)) + 1u; }  // xor + min3 + add
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ uint32_t(A ^ B ^ C ^ D) ^ uint32_t(fa + fb + fc + fd);
}
template <int OP> void run(const char *name, int ops_per_iter, uint32_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 blocks of 256 per CU = 8 waves/SIMD
    k<OP><<<blocks, 256>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * N_ITER * ops
    double winst = double(blocks) * 4 / 1024.0 * N_ITER * ops_per_iter;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<3>("8x simple valu (xor,add)", 8, out);
    run<0>("4x mul_lo(+add)", 4, out);
    run<1>("4x mul_hi(+xor)", 4, out);
    run<2>("4x mad_u64_u32", 4, out);
    run<4>("4x mul_f64", 4, out);
    run<5>("4x add_f64", 4, out);
    run<6>("4x (cmp_lt_u64+cndmask+add+add64)", 4, out);
    run<7>("4x (mov_dpp+add)", 4, out);
    run<8>("4x (bpermute+add)", 4, out);
    run<9>("4x (lshr_b64+and+add64)", 4, out);
    run<10>("4x (xor+min3/min+add)", 4, out);
    return 0;
}
