// Issue / dependency cost of instruction sequences for ONE wave alone on its SIMD (gfx950): what a latency-bound loop
// (the fused rollout at <= 2 waves per SIMD) pays per instruction.  Each case is an inline-asm sequence repeated
// REPS times inside a timed loop; cycles = s_memtime delta / (iterations * REPS * instructions in the sequence).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/single_wave_latency.hip -o /tmp/swl && /tmp/swl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 256;

#define BODY8(S) S S S S S S S S

template <int CASE>
__global__ void __launch_bounds__(64) k(unsigned long long *out, unsigned *sink, const unsigned *lds_src) {
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = lds_src ? lds_src[i] : (i * 16) & 4095;
    __syncthreads();
    unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 7, c = threadIdx.x * 5 + 11, d = threadIdx.x * 7 + 13, e = 0x12345, f = 0x777;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0000001, z = 0.9999999, w = 1.00000003;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITER; ++it) {
        if (CASE == 0) {        // 8 dependent VOP2 (e32) ops
            asm volatile(BODY8("v_xor_b32_e32 %0, %1, %0\n\t") : "+v"(a) : "v"(b));
        } else if (CASE == 1) { // 8 independent VOP2 ops (4 chains, 2 each)
            asm volatile("v_xor_b32_e32 %0, %4, %0\n\tv_xor_b32_e32 %1, %4, %1\n\tv_xor_b32_e32 %2, %4, %2\n\tv_xor_b32_e32 %3, %4, %3\n\t"
                         "v_xor_b32_e32 %0, %4, %0\n\tv_xor_b32_e32 %1, %4, %1\n\tv_xor_b32_e32 %2, %4, %2\n\tv_xor_b32_e32 %3, %4, %3\n\t"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
        } else if (CASE == 2) { // 8 dependent VOP3 ops
            asm volatile(BODY8("v_lshl_add_u32 %0, %0, 1, %1\n\t") : "+v"(a) : "v"(b));
        } else if (CASE == 3) { // 8 independent VOP3 ops
            asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n\tv_lshl_add_u32 %1, %1, 1, %4\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_lshl_add_u32 %3, %3, 1, %4\n\t"
                         "v_lshl_add_u32 %0, %0, 1, %4\n\tv_lshl_add_u32 %1, %1, 1, %4\n\tv_lshl_add_u32 %2, %2, 1, %4\n\tv_lshl_add_u32 %3, %3, 1, %4\n\t"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
        } else if (CASE == 4) { // 8 dependent SDWA ops
            asm volatile(BODY8("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t") : "+v"(a) : "v"(b));
        } else if (CASE == 5) { // 8 independent SDWA ops
            asm volatile("v_sub_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         "v_sub_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
        } else if (CASE == 6) { // VALU then DPP of its result (needs wait states): 4 x (xor; nop; mov_dpp)
            asm volatile("v_xor_b32_e32 %0, %1, %0\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_xor_b32_e32 %0, %1, %0\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_xor_b32_e32 %0, %1, %0\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_xor_b32_e32 %0, %1, %0\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         : "+v"(a) : "v"(b));
        } else if (CASE == 7) { // v_cmp -> vcc -> v_cndmask (e32), 4 pairs dependent
            asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                         "v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                         "v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                         "v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                         : "+v"(a) : "v"(b), "v"(c) : "vcc");
        } else if (CASE == 8) { // v_cmp -> sgpr pair -> s_and -> v_cndmask (e64): 2 groups of 4 instr
            asm volatile("v_cmp_gt_u32_e64 s[20:21], %1, %0\n\ts_and_b64 s[20:21], s[20:21], exec\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %2, %0, s[20:21]\n\t"
                         "v_cmp_gt_u32_e64 s[20:21], %1, %0\n\ts_and_b64 s[20:21], s[20:21], exec\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %2, %0, s[20:21]\n\t"
                         : "+v"(a) : "v"(b), "v"(c) : "s20", "s21", "scc");
        } else if (CASE == 9) { // 8 dependent v_mad_u64_u32
            unsigned long long p = a;
            asm volatile(BODY8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t") : "+v"(p) : "v"(b), "v"(c) : "vcc");
            a = unsigned(p) ^ unsigned(p >> 32);
        } else if (CASE == 10) { // 8 independent-ish v_mad_u64_u32 (4 accumulators)
            unsigned long long p0 = a, p1 = b, p2 = c, p3 = d;
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
                         "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(e), "v"(f) : "vcc");
            a = unsigned(p0); b = unsigned(p1); c = unsigned(p2); d = unsigned(p3);
        } else if (CASE == 11) { // 8 dependent v_mul_f64
            asm volatile(BODY8("v_mul_f64 %0, %0, %1\n\t") : "+v"(x) : "v"(y));
        } else if (CASE == 12) { // 8 independent v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                         "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(1.0000000001));
        } else if (CASE == 13) { // 8 SALU ops dependent
            unsigned sv = it;
            asm volatile(BODY8("s_add_u32 %0, %0, 3\n\t") : "+s"(sv) :: "scc");
            a ^= sv;
        } else if (CASE == 14) { // ds_read_b32 dependent chain (pointer chase), 4 per iteration
            a &= 0xffcu;                                       // 4-byte aligned byte address inside lds[]
            asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "+v"(a));
        } else if (CASE == 15) { // ds_read_b128 random rows, 4 issued then waited (like the table gather)
            unsigned r0, r1, r2, r3;
            uint4 q0, q1, q2, q3;
            r0 = (a * 16u) & 0x3ff0u; r1 = (b * 16u) & 0x3ff0u; r2 = (c * 16u) & 0x3ff0u; r3 = (d * 16u) & 0x3ff0u;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)\n\t"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            a += q0.x + q0.w; b += q1.y; c += q2.z; d += q3.w;
        } else if (CASE == 16) { // taken scalar branches: 4 per iteration
            asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_mov_b32 %0, 0\n1:\n\t"
                         "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 2f\n\tv_mov_b32 %0, 0\n2:\n\t"
                         "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 3f\n\tv_mov_b32 %0, 0\n3:\n\t"
                         "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 4f\n\tv_mov_b32 %0, 0\n4:\n\t" : "+v"(a) :: "scc");
        } else if (CASE == 17) { // not-taken scalar branches: 4 per iteration
            asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\tv_xor_b32 %0, %1, %0\n1:\n\t"
                         "s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 2f\n\tv_xor_b32 %0, %1, %0\n2:\n\t"
                         "s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 3f\n\tv_xor_b32 %0, %1, %0\n3:\n\t"
                         "s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 4f\n\tv_xor_b32 %0, %1, %0\n4:\n\t" : "+v"(a) : "v"(b) : "scc");
        } else if (CASE == 18) { // v_perm_b32 dependent x8
            asm volatile(BODY8("v_perm_b32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(0x07020500u));
        } else if (CASE == 19) { // v_pk_min_u16 dependent x8
            asm volatile(BODY8("v_pk_min_u16 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        } else if (CASE == 20) { // alternating VALU / SALU independent (8 each)
            unsigned sv = it;
            asm volatile(BODY8("v_xor_b32_e32 %0, %2, %0\n\ts_add_u32 %1, %1, 3\n\t") : "+v"(a), "+s"(sv) : "v"(b) : "scc");
            a ^= sv;
        } else if (CASE == 21) { // v_readfirstlane -> s use -> v use chain x4
            asm volatile("v_readfirstlane_b32 s20, %0\n\ts_add_u32 s20, s20, 1\n\tv_xor_b32 %0, s20, %0\n\t"
                         "v_readfirstlane_b32 s20, %0\n\ts_add_u32 s20, s20, 1\n\tv_xor_b32 %0, s20, %0\n\t"
                         "v_readfirstlane_b32 s20, %0\n\ts_add_u32 s20, s20, 1\n\tv_xor_b32 %0, s20, %0\n\t"
                         "v_readfirstlane_b32 s20, %0\n\ts_add_u32 s20, s20, 1\n\tv_xor_b32 %0, s20, %0\n\t" : "+v"(a) :: "s20", "scc");
        } else if (CASE == 22) { // v_cmp e64 -> s_or with another pair -> v_cndmask: the flags idiom, 2 x 5 instrs
            asm volatile("v_cmp_eq_u32_e64 s[20:21], %1, %0\n\tv_cmp_gt_u32_e32 vcc, %2, %0\n\ts_or_b64 s[20:21], vcc, s[20:21]\n\tv_cndmask_b32_e64 %0, 0, 1, s[20:21]\n\tv_add_u32 %0, %0, %1\n\t"
                         "v_cmp_eq_u32_e64 s[20:21], %1, %0\n\tv_cmp_gt_u32_e32 vcc, %2, %0\n\ts_or_b64 s[20:21], vcc, s[20:21]\n\tv_cndmask_b32_e64 %0, 0, 1, s[20:21]\n\tv_add_u32 %0, %0, %1\n\t"
                         : "+v"(a) : "v"(b), "v"(c) : "s20", "s21", "vcc", "scc");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d ^ unsigned(__double2loint(x + y + z + w));
}

struct Case { const char *name; int instrs; };
static const Case kCases[] = {
    {"8 dependent v_xor (VOP2 e32)", 8}, {"8 independent v_xor (4 chains)", 8}, {"8 dependent v_lshl_add (VOP3)", 8},
    {"8 independent v_lshl_add (VOP3)", 8}, {"8 dependent v_sub_sdwa", 8}, {"8 independent v_sub_sdwa", 8},
    {"4 x (v_xor; s_nop 1; v_mov_dpp of it)", 12}, {"4 x (v_cmp e32 -> vcc -> v_cndmask e32) dependent", 8},
    {"2 x (v_cmp e64 -> s_and -> s_nop -> v_cndmask e64)", 8}, {"8 dependent v_mad_u64_u32", 8}, {"8 v_mad_u64_u32, 4 accumulators", 8},
    {"8 dependent v_mul_f64", 8}, {"8 independent v_mul_f64", 8}, {"8 dependent s_add_u32", 8},
    {"4 dependent ds_read_b32 (pointer chase, incl. wait)", 4}, {"4 ds_read_b128 random rows + one wait (per group of 4)", 1},
    {"4 x (s_cmp; taken s_cbranch)", 8}, {"4 x (s_cmp; not-taken s_cbranch; v_xor)", 12}, {"8 dependent v_perm_b32", 8},
    {"8 dependent v_pk_min_u16", 8}, {"8 x (v_xor; s_add) alternating", 16}, {"4 x (v_readfirstlane; s_add; v_xor)", 12},
    {"2 x (v_cmp e64; v_cmp e32; s_or; v_cndmask e64; v_add)", 10},
};

template <int CASE>
int run(unsigned long long *d_out, unsigned *d_sink, int blocks) {
    hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(64), 0, 0, d_out, d_sink, (const unsigned *)nullptr);
    hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(64), 0, 0, d_out, d_sink, (const unsigned *)nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), d_out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += double(v);
    const double cyc = sum / blocks / ITER;
    printf("%-58s %8.1f cycles per group  = %6.2f per instruction\n", kCases[CASE].name, cyc, cyc / kCases[CASE].instrs);
    fflush(stdout);
    return 0;
}

int main() {
    const int blocks = 256;   // one single-wave block per CU at most per SIMD: waves do not share SIMDs much
    unsigned long long *d_out; unsigned *d_sink;
    CHECK(hipMalloc(&d_out, blocks * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, blocks * 64 * sizeof(unsigned)));
    printf("# one wave per block, %d blocks (<= 1 wave per CU): s_memtime cycles\n", blocks);
    fflush(stdout);
    run<0>(d_out, d_sink, blocks); run<1>(d_out, d_sink, blocks); run<2>(d_out, d_sink, blocks); run<3>(d_out, d_sink, blocks);
    run<4>(d_out, d_sink, blocks); run<5>(d_out, d_sink, blocks); run<6>(d_out, d_sink, blocks); run<7>(d_out, d_sink, blocks);
    run<8>(d_out, d_sink, blocks); run<9>(d_out, d_sink, blocks); run<10>(d_out, d_sink, blocks); run<11>(d_out, d_sink, blocks);
    run<12>(d_out, d_sink, blocks); run<13>(d_out, d_sink, blocks); run<14>(d_out, d_sink, blocks); run<15>(d_out, d_sink, blocks);
    run<16>(d_out, d_sink, blocks); run<17>(d_out, d_sink, blocks); run<18>(d_out, d_sink, blocks); run<19>(d_out, d_sink, blocks);
    run<20>(d_out, d_sink, blocks); run<21>(d_out, d_sink, blocks); run<22>(d_out, d_sink, blocks);
    return 0;
}
