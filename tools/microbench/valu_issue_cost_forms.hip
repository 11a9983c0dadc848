// Which ENCODINGS / OPERAND KINDS of the common 32-bit vector instructions issue at the fast rate (v_xor_b32 on vector
// registers) and which at the ordinary one (~1.6 x)?  Same harness as valu_issue_cost.hip: 8 waves per SIMD, 8 independent
// registers per lane, each op applied to all 8 per iteration.  profiles/r04_valu_issue_cost_forms.txt
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/vicf tools/microbench/valu_issue_cost_forms.hip && /tmp/vicf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048
#define OP8(TEMPLATE) \
  asm volatile(TEMPLATE(0) TEMPLATE(1) TEMPLATE(2) TEMPLATE(3) TEMPLATE(4) TEMPLATE(5) TEMPLATE(6) TEMPLATE(7) \
   : "+v"(r[0]),"+v"(r[1]),"+v"(r[2]),"+v"(r[3]),"+v"(r[4]),"+v"(r[5]),"+v"(r[6]),"+v"(r[7]) : "v"(k0), "v"(k1), "s"(sk) : "vcc");
#define T0(i)  "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T1(i)  "v_and_b32 %" #i ", %" #i ", %8\n"
#define T2(i)  "v_or_b32 %" #i ", %" #i ", %8\n"
#define T3(i)  "v_lshlrev_b32 %" #i ", %8, %" #i "\n"
#define T4(i)  "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define T5(i)  "v_lshrrev_b32 %" #i ", 16, %" #i "\n"
#define T6(i)  "v_and_b32 %" #i ", 15, %" #i "\n"
#define T7(i)  "v_and_b32 %" #i ", 0xffff, %" #i "\n"
#define T8(i)  "v_mov_b32 %" #i ", %8\n"
#define T9(i)  "v_mov_b32 %" #i ", 7\n"
#define T10(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define T11(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define T12(i) "v_add_u32 %" #i ", 5, %" #i "\n"
#define T13(i) "v_xor_b32 %" #i ", 0x80008000, %" #i "\n"
#define T14(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 16\n"
#define T15(i) "v_lshl_add_u32 %" #i ", %" #i ", 4, %8\n"
#define T16(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define T17(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
#define T18(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %10 bitop3:0x96\n"
#define T19(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xc8\n"
#define T20(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
#define T21(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T22(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T23(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define T24(i) "v_xor_b32_e64 %" #i ", %" #i ", %8\n"
#define T25(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define T26(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T27(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, uint32_t sk) {
    uint32_t r[8]; for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + seed + i;
    uint32_t k0 = seed * 7 + 3, k1 = seed ^ 0x55aa;
    for (int it = 0; it < N_ITER; ++it) {
#define CASE(n) if (OP == n) OP8(T##n)
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
        CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27)
    }
    uint32_t a = 0; for (int i = 0; i < 8; ++i) a ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
static double g_base = 0;
template <int OP> void run(const char *name, int instr_per_op, uint32_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    k<OP><<<blocks, 256>>>(out, 1, 5); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 2, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (double(blocks) * 4 / 1024.0 * N_ITER * 8 * instr_per_op);
    if (g_base == 0) g_base = ns;
    printf("%-44s %7.3f ms  %.3f ns/wave-instr/SIMD  = %.2f x v_xor_b32\n", name, ms, ns, ns / g_base);
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("warm", 1, out); g_base = 0;
    run<0>("v_xor_b32 v, v, v", 1, out);
    run<1>("v_and_b32 v, v, v", 1, out); run<2>("v_or_b32 v, v, v", 1, out); run<3>("v_lshlrev_b32 v, v, v", 1, out);
    run<4>("v_lshlrev_b32 v, 3, v", 1, out); run<5>("v_lshrrev_b32 v, 16, v", 1, out); run<6>("v_and_b32 v, 15, v", 1, out);
    run<7>("v_and_b32 v, 0xffff, v   (literal)", 1, out); run<8>("v_mov_b32 v, v", 1, out); run<9>("v_mov_b32 v, 7", 1, out);
    run<10>("v_sub_u32 v, v, v", 1, out); run<11>("v_min_u32 v, v, v", 1, out); run<12>("v_add_u32 v, 5, v", 1, out);
    run<13>("v_xor_b32 v, 0x80008000, v   (literal)", 1, out); run<14>("v_alignbit_b32 v, v, v, 16", 1, out);
    run<15>("v_lshl_add_u32 v, v, 4, v", 1, out); run<16>("v_or3_b32 v, v, v, v", 1, out); run<17>("v_add_lshl_u32 v, v, v, 1", 1, out);
    run<18>("v_bitop3_b32 v, v, v, s", 1, out); run<19>("v_bitop3_b32 v, v, v, v  (and-or)", 1, out);
    run<20>("v_cmp_eq_u32 vcc, v, v", 1, out); run<21>("v_cndmask_b32 v, v, v, vcc  (vcc stale)", 1, out);
    run<22>("v_cmp_lt_u32 + v_cndmask_b32", 2, out); run<23>("v_pk_min_u16 v, v, v", 1, out);
    run<24>("v_xor_b32_e64 v, v, v  (VOP3 encoding)", 1, out); run<25>("v_add_u32_e64 v, v, v  (VOP3 encoding)", 1, out);
    run<26>("v_mul_u32_u24 v, v, v", 1, out); run<27>("v_perm_b32 v, v, v, v", 1, out);
    return 0;
}
