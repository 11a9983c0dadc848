// Issue cost of the 64-bit / packed vector instructions the step kernels use, relative to v_xor_b32: 8 waves per SIMD, 8
// independent register (pairs) per lane, each op applied to all 8 per iteration (throughput, not latency).  The weights
// tools/derive_valu.py applies to SQ_INSTS_VALU_INT64 / _MUL_F64 / _ADD_F64 come from here (profiles/r04_valu_issue_cost.txt).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vic64 tools/microbench/valu_issue_cost64.hip && /tmp/vic64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048
#define OP8_64(TEMPLATE) \
  asm volatile(TEMPLATE(0) TEMPLATE(1) TEMPLATE(2) TEMPLATE(3) TEMPLATE(4) TEMPLATE(5) TEMPLATE(6) TEMPLATE(7) \
   : "+v"(r[0]),"+v"(r[1]),"+v"(r[2]),"+v"(r[3]),"+v"(r[4]),"+v"(r[5]),"+v"(r[6]),"+v"(r[7]) : "v"(k0), "v"(k1), "v"(kd) : "vcc");
#define T_XOR(i)    "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_MAD64(i)  "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define T_MULF64(i) "v_mul_f64 %" #i ", %" #i ", %10\n"
#define T_ADDF64(i) "v_add_f64 %" #i ", %" #i ", %10\n"
#define T_FMAF64(i) "v_fma_f64 %" #i ", %" #i ", %10, %10\n"
#define T_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %10\n"
#define T_PKSUB(i)  "v_pk_sub_i16 %" #i ", %" #i ", %8 clamp\n"
#define T_PKMIN(i)  "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define T_DOT2(i)   "v_dot2_i32_i16 %" #i ", %" #i ", %8, %9\n"
#define T_MINSDWA(i) "v_min_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
template <int OP, typename R>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    R r[8]; for (int i = 0; i < 8; ++i) r[i] = R(threadIdx.x * 2654435761u + seed + i);
    uint32_t k0 = seed * 7 + 3, k1 = seed ^ 0x55aa;
    double kd = 1.0000001 + seed * 1e-9;
    for (int it = 0; it < N_ITER; ++it) {
        if (OP == 0) OP8_64(T_XOR) if (OP == 1) OP8_64(T_MAD64) if (OP == 2) OP8_64(T_MULF64) if (OP == 3) OP8_64(T_ADDF64)
        if (OP == 4) OP8_64(T_FMAF64) if (OP == 5) OP8_64(T_LSHLADD64) if (OP == 6) OP8_64(T_PKSUB) if (OP == 7) OP8_64(T_PKMIN)
        if (OP == 8) OP8_64(T_DOT2) if (OP == 9) OP8_64(T_MINSDWA)
    }
    uint32_t a = 0; for (int i = 0; i < 8; ++i) a ^= uint32_t(uint64_t(r[i]));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int OP, typename R> double run(const char *name, uint32_t *out, double base_ns) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    k<OP, R><<<blocks, 256>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP, R><<<blocks, 256>>>(out, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = double(blocks) * 4 / 1024.0 * N_ITER * 8;      // wave-instructions per SIMD
    const double ns = ms * 1e6 / winst;
    printf("%-22s %7.3f ms  %.3f ns/wave-instr/SIMD  = %.2f x v_xor_b32 = %.1f cycles at 4 per v_xor_b32\n", name, ms, ns, base_ns > 0 ? ns / base_ns : 1.0,
           base_ns > 0 ? 4.0 * ns / base_ns : 4.0);
    return ns;
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0, uint32_t>("warm", out, 0);
    const double base = run<0, uint32_t>("v_xor_b32", out, 0);
    run<0, uint32_t>("v_xor_b32", out, base);
    run<1, uint64_t>("v_mad_u64_u32", out, base);
    run<2, double>("v_mul_f64", out, base);
    run<3, double>("v_add_f64", out, base);
    run<4, double>("v_fma_f64", out, base);
    run<5, uint64_t>("v_lshl_add_u64", out, base);
    run<6, uint32_t>("v_pk_sub_i16 clamp", out, base);
    run<7, uint32_t>("v_pk_min_u16", out, base);
    run<8, uint32_t>("v_dot2_i32_i16", out, base);
    run<9, uint32_t>("v_min_u32_sdwa", out, base);
    return 0;
}
