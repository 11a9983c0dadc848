// per-instruction issue cost: 8 waves/SIMD, 8 independent registers per lane, each op applied to all 8 per iteration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048
#define OP8(TEMPLATE) \
  asm volatile(TEMPLATE(0) TEMPLATE(1) TEMPLATE(2) TEMPLATE(3) TEMPLATE(4) TEMPLATE(5) TEMPLATE(6) TEMPLATE(7) \
   : "+v"(r[0]),"+v"(r[1]),"+v"(r[2]),"+v"(r[3]),"+v"(r[4]),"+v"(r[5]),"+v"(r[6]),"+v"(r[7]) : "v"(k0), "v"(k1), "s"(sk) : "vcc");
#define T_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_MIN3(i)  "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define T_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 3, 16\n"
#define T_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define T_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define T_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T_DPP(i)   "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define T_XORDPP(i) "v_xor_b32_dpp %" #i ", %8, %" #i " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define T_SDWA(i)  "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define T_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_XORS(i)  "v_xor_b32 %" #i ", %10, %" #i "\n"
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, uint32_t sk) {
    uint32_t r[8]; for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + seed + i;
    uint32_t k0 = seed * 7 + 3, k1 = seed ^ 0x55aa;
    for (int it = 0; it < N_ITER; ++it) {
        if (OP == 0) OP8(T_XOR) if (OP == 1) OP8(T_ADD) if (OP == 2) OP8(T_MIN3) if (OP == 3) OP8(T_BITOP) if (OP == 4) OP8(T_PERM)
        if (OP == 5) OP8(T_BFE) if (OP == 6) OP8(T_ANDOR) if (OP == 7) OP8(T_LSHLOR) if (OP == 8) OP8(T_CMPCND) if (OP == 9) OP8(T_CND)
        if (OP == 10) OP8(T_DPP) if (OP == 11) OP8(T_XORDPP) if (OP == 12) OP8(T_SDWA) if (OP == 13) OP8(T_MULLO) if (OP == 14) OP8(T_MUL24)
        if (OP == 15) OP8(T_MAD24) if (OP == 16) OP8(T_ADD3) if (OP == 17) OP8(T_XORS)
    }
    uint32_t a = 0; for (int i = 0; i < 8; ++i) a ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int OP> void run(const char *name, int instr_per_op, uint32_t *out, double base_ns) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    k<OP><<<blocks, 256>>>(out, 1, 5); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 2, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = double(blocks) * 4 / 1024.0 * N_ITER * 8 * instr_per_op;
    double ns = ms * 1e6 / winst;
    printf("%-22s %7.3f ms  %.3f ns/wave-instr/SIMD  = %.2f x v_xor\n", name, ms, ns, base_ns > 0 ? ns / base_ns : 1.0);
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("warm", 1, out, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<0><<<2048, 256>>>(out, 1, 5); hipDeviceSynchronize();
    hipEventRecord(e0); k<0><<<2048, 256>>>(out, 2, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double base = ms * 1e6 / (2048.0 * 4 / 1024.0 * N_ITER * 8);
    run<0>("v_xor_b32", 1, out, base); run<1>("v_add_u32", 1, out, base); run<2>("v_min3_u32", 1, out, base); run<3>("v_bitop3_b32", 1, out, base);
    run<4>("v_perm_b32", 1, out, base); run<5>("v_bfe_u32", 1, out, base); run<6>("v_and_or_b32", 1, out, base); run<7>("v_lshl_or_b32", 1, out, base);
    run<8>("v_cmp+v_cndmask", 2, out, base); run<9>("v_cndmask(vcc)", 1, out, base); run<10>("v_mov_b32_dpp", 1, out, base); run<11>("v_xor_b32_dpp", 1, out, base);
    run<12>("v_xor_b32_sdwa", 1, out, base); run<13>("v_mul_lo_u32", 1, out, base); run<14>("v_mul_u32_u24", 1, out, base); run<15>("v_mad_u32_u24", 1, out, base);
    run<16>("v_add3_u32", 1, out, base); run<17>("v_xor_b32 (sgpr)", 1, out, base);
    return 0;
}
