// Latency of DEPENDENT vector instructions on MI355X (gfx950): one wave per CU runs a chain in which every instruction needs the one before it;
// cycles per instruction = what an ordered sum pays per term when nothing else is there to issue (valu_rate.hip measures the issue cost with independent chains).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/dep_latency.hip -o tools/ubench/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters, int waves) {
    __shared__ unsigned lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = ((i + 1) & 255) * 4;  // a ring: each word holds the byte address of the next
    __syncthreads();
    double f = threadIdx.x, g = 1.000001;
    float s = threadIdx.x, t = 1.0001f;
    unsigned a = threadIdx.x * 4, b = 3;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(f) : "v"(g));) }
        if (OP == 1) { REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f) : "v"(g));) }
        if (OP == 2) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(t));) }
        if (OP == 3) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f) : "v"(g));) }
        if (OP == 4) { REP32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 5) { REP32(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a));) }
        if (OP == 6) { REP32(asm volatile("v_cvt_f64_f32 %0, %1\n\tv_cvt_f32_f64 %1, %0" : "+v"(f), "+v"(s));) }
        if (OP == 7) { REP32(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s) : "v"(t));) }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
    if (f + s + a == 1.2345) out[1] = 1;
}
template <int OP>
void run(const char* name, int per) {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    const int iters = 2000;
    for (int waves : {1, 4, 8, 16}) {  // blocks (= waves) per CU
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(64), 0, 0, d, iters, waves);
        (void)hipDeviceSynchronize();
        unsigned long long h = 0;
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-28s %2d waves/CU: %7.2f cycles per dependent instruction (wave 0's clock)\n", name, waves, (double)h / ((double)iters * 32 * per));
    }
    (void)hipFree(d);
}
int main() {
    run<0>("v_add_f64", 1);
    run<1>("v_mul_f64", 1);
    run<3>("v_fma_f64", 1);
    run<2>("v_add_f32", 1);
    run<7>("v_mul_f32", 1);
    run<4>("v_add_u32", 1);
    run<6>("v_cvt_f64_f32 + v_cvt_f32_f64", 2);
    run<5>("ds_read_b32 (pointer chase)", 1);
    return 0;
}
