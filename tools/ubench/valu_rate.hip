// Issue cost of vector instructions on MI355X (gfx950): one block per CU of 4 waves per SIMD, every wave a long stream of ONE
// instruction on four independent register chains; whole-kernel time -> SIMD cycles per wave-instruction.
// (Timed as a whole: the oldest wave of a SIMD wins the issue arbitration, so one wave's own clock says nothing about the rest.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate ; results: DESIGN.md section 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP16(X) X X X X X X X X X X X X X X X X
#define OPS(_) \
    _(0, "v_add_u32", "v_add_u32 %0, %0, %8", A) _(1, "v_sub_u32", "v_sub_u32 %0, %0, %8", A) _(2, "v_and_b32", "v_and_b32 %0, %0, %8", A) \
    _(3, "v_or_b32", "v_or_b32 %0, %0, %8", A) _(4, "v_xor_b32", "v_xor_b32 %0, %0, %8", A) _(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 2, %0", A) \
    _(6, "v_lshrrev_b32", "v_lshrrev_b32 %0, 1, %0", A) _(7, "v_min_u32", "v_min_u32 %0, %0, %8", A) _(8, "v_max_u32", "v_max_u32 %0, %0, %8", A) \
    _(9, "v_min3_u32", "v_min3_u32 %0, %0, %8, %9", A) _(10, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %8", A) _(11, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %8, %9", A) \
    _(12, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %8", A) _(13, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %8", A) _(14, "v_add3_u32", "v_add3_u32 %0, %0, %8, %9", A) \
    _(15, "v_and_or_b32", "v_and_or_b32 %0, %0, %8, %9", A) _(16, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8", A) _(17, "v_perm_b32", "v_perm_b32 %0, %0, %8, %9", A) \
    _(18, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %8, vcc", A) _(19, "v_mov_b32", "v_mov_b32 %0, %8", A) _(20, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %8", A) \
    _(26, "v_cndmask_b32_e64", "v_cndmask_b32_e64 %0, %0, %8, s[20:21]", A) _(27, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %8", A) _(28, "v_lshlrev_b32 by v", "v_lshlrev_b32 %0, %9, %0", A) _(29, "v_min_i32", "v_min_i32 %0, %0, %8", A) _(21, "v_cmp_lt_u32", "v_cmp_lt_u32 vcc, %0, %8", A) _(22, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %8, 7", A) _(23, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %4, %0", A) \
    _(24, "v_cvt_f32_u32", "v_cvt_f32_u32 %4, %0", A) _(25, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %4", A) \
    _(30, "v_add_f32", "v_add_f32 %4, %4, %10", A) _(31, "v_sub_f32 clamp", "v_sub_f32 %4, %4, %10 clamp", A) _(32, "v_mul_f32", "v_mul_f32 %4, %4, %10", A) \
    _(33, "v_fma_f32", "v_fma_f32 %4, %4, %10, %10", A) _(34, "v_fmac_f32", "v_fmac_f32 %4, %10, %10", A) _(35, "v_min_f32", "v_min_f32 %4, %4, %10", A) \
    _(36, "v_max_f32", "v_max_f32 %4, %4, %10", A) _(37, "v_med3_f32", "v_med3_f32 %4, %4, %10, %10", A) _(38, "v_rcp_f32", "v_rcp_f32 %4, %4", A) \
    _(39, "v_exp_f32", "v_exp_f32 %4, %4", A) _(40, "v_ldexp_f32", "v_ldexp_f32 %4, %4, %8", A) _(41, "v_rndne_f32", "v_rndne_f32 %4, %4", A) \
    _(42, "v_floor_f32", "v_floor_f32 %4, %4", A) _(43, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %4, %10", A) \
    _(50, "v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 0, %8", L) _(51, "v_add_f64", "v_add_f64 %4, %4, %10", L) _(52, "v_mul_f64", "v_mul_f64 %4, %4, %10", L) \
    _(53, "v_fma_f64", "v_fma_f64 %4, %4, %10, %10", L) _(54, "v_cvt_f64_f32", "v_cvt_f64_f32 %4, %11", L) _(55, "v_cvt_f32_f64", "v_cvt_f32_f64 %11, %4", L) \
    _(56, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %12, %12, %0", L) _(57, "v_pk_mul_f32", "v_pk_mul_f32 %4, %4, %10", L) _(58, "v_pk_add_f32", "v_pk_add_f32 %4, %4, %10", L) \
    _(59, "v_pk_fma_f32", "v_pk_fma_f32 %4, %4, %10, %10", L) _(60, "v_rcp_f64", "v_rcp_f64 %4, %4", L) _(61, "v_cvt_f64_u32", "v_cvt_f64_u32 %4, %12", L) \
    _(62, "v_cvt_i32_f64", "v_cvt_i32_f64 %12, %4", L) _(63, "v_ldexp_f64", "v_ldexp_f64 %4, %4, %12", L) _(64, "v_floor_f64", "v_floor_f64 %4, %4", L)

// A: four chains of 32-bit registers: ints %0-%3, floats %4-%7; %8, %9 int operands, %10 float operand
// L: four chains of 64-bit registers: u64 %0-%3, doubles %4-%7; %8 u64 operand, %10 double operand, %11 a float, %12 an int
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 7, c = 0x01020304, e = 3;
    float s0 = a0, s1 = a1, s2 = a2, s3 = a3, t = 1.0001f;
    unsigned long long x0 = a0, x1 = a1, x2 = a2, x3 = a3, d = 0x12345678abcdULL;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3, g = 1.000001;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#define CASE_A(ID, NAME, INSN)                                                                                                          \
    if (OP == ID) {                                                                                                                     \
        REP16(asm volatile(INSN : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b), "v"(c), "v"(t) : "vcc"); \
              asm volatile(INSN : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3), "+v"(s1), "+v"(s0), "+v"(s2), "+v"(s3) : "v"(b), "v"(c), "v"(t) : "vcc"); \
              asm volatile(INSN : "+v"(a2), "+v"(a1), "+v"(a0), "+v"(a3), "+v"(s2), "+v"(s1), "+v"(s0), "+v"(s3) : "v"(b), "v"(c), "v"(t) : "vcc"); \
              asm volatile(INSN : "+v"(a3), "+v"(a1), "+v"(a2), "+v"(a0), "+v"(s3), "+v"(s1), "+v"(s2), "+v"(s0) : "v"(b), "v"(c), "v"(t) : "vcc");) \
    }
#define CASE_L(ID, NAME, INSN)                                                                                                          \
    if (OP == ID) {                                                                                                                     \
        REP16(asm volatile(INSN : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(d), "v"(c), "v"(g), "v"(s0), "v"(e) : "vcc"); \
              asm volatile(INSN : "+v"(x1), "+v"(x0), "+v"(x2), "+v"(x3), "+v"(f1), "+v"(f0), "+v"(f2), "+v"(f3) : "v"(d), "v"(c), "v"(g), "v"(s1), "v"(e) : "vcc"); \
              asm volatile(INSN : "+v"(x2), "+v"(x1), "+v"(x0), "+v"(x3), "+v"(f2), "+v"(f1), "+v"(f0), "+v"(f3) : "v"(d), "v"(c), "v"(g), "v"(s2), "v"(e) : "vcc"); \
              asm volatile(INSN : "+v"(x3), "+v"(x1), "+v"(x2), "+v"(x0), "+v"(f3), "+v"(f1), "+v"(f2), "+v"(f0) : "v"(d), "v"(c), "v"(g), "v"(s3), "v"(e) : "vcc");) \
    }
#define CASE(ID, NAME, INSN, KIND) CASE_##KIND(ID, NAME, INSN)
        OPS(CASE)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)x0 ^ (unsigned)x1 ^ (unsigned)x2 ^ (unsigned)x3) == 0x7fffffff && f0 + f1 + f2 + f3 + s0 + s1 + s2 + s3 == 1.25) out[2] = 1;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 64 * waves_per_simd;  // per SIMD
    const double ghz = (double)h[0] / (h[1] * 10.0);           // wave 0's cycle counter against the 100 MHz clock
    printf("%-18s %d waves/SIMD: %5.2f SIMD cycles per wave-instruction (%.0f us at %.2f GHz)\n", name, waves_per_simd, ms * 1e-3 * ghz * 1e9 / instr, ms * 1e3, ghz);
    (void)hipFree(d);
}
int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 4;
#define RUN(ID, NAME, INSN, KIND) run<ID>(NAME, w);
    OPS(RUN)
    return 0;
}
