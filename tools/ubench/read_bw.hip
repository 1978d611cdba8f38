// Read-only HBM bandwidth of MI355X as a streaming kernel sees it (what bounds K1f, which reads every pixel once and writes 1/128 of that):
// every lane reads 16 B per load, UNROLL loads in flight, blocks walk the buffer either grid-strided or in one contiguous slab each.
// usage: tools/ubench/read_bw [GiB]   -> one line per variant, TB/s.  `make ubench` builds it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool SLAB>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, size_t n16, uint32_t* out) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t step = (size_t)256 * UNROLL;
    size_t i, end, stride;
    if (SLAB) {
        const size_t per = (n16 / gridDim.x) / step * step;
        i = (size_t)blockIdx.x * per; end = i + per; stride = step;
    } else {
        i = (size_t)blockIdx.x * step; end = n16 / step * step; stride = (size_t)gridDim.x * step;
    }
    for (; i < end; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const u32x4* p = src + i + (size_t)u * 256 + threadIdx.x;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345679u) out[blockIdx.x] = r;  // never true for the fill below; keeps the loads alive
}

__global__ void k_fill(u32x4* dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)i * 2654435761u;
        dst[i] = u32x4{x, x ^ 1u, x ^ 2u, x ^ 3u};
    }
}

template <int UNROLL, bool NT, bool SLAB>
static void run(const char* name, const u32x4* src, size_t n16, uint32_t* out, int grid) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) k_read<UNROLL, NT, SLAB><<<grid, 256>>>(src, n16, out);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; r++) k_read<UNROLL, NT, SLAB><<<grid, 256>>>(src, n16, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s grid %6d: %.3f ms per pass, %.2f TB/s\n", name, grid, ms / reps, (double)n16 * 16 * reps / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t n16 = (size_t)(gib * (1 << 30)) / 16;
    u32x4* src; uint32_t* out;
    CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&out, 1 << 20));
    k_fill<<<4096, 256>>>(src, n16); CK(hipDeviceSynchronize());
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1, false, false>("1 x 16 B in flight, grid-strided", src, n16, out, grid);
        run<4, false, false>("4 x 16 B in flight, grid-strided", src, n16, out, grid);
        run<8, false, false>("8 x 16 B in flight, grid-strided", src, n16, out, grid);
        run<4, true, false>("4 x 16 B in flight, grid-strided, nontemporal", src, n16, out, grid);
        run<4, false, true>("4 x 16 B in flight, one slab per block", src, n16, out, grid);
        run<4, true, true>("4 x 16 B in flight, one slab per block, nt", src, n16, out, grid);
    }
    // the device's own copy, for scale (bytes read + written)
    u32x4* dst; CK(hipMalloc(&dst, n16 * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemcpyAsync(dst, src, n16 * 16, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) CK(hipMemcpyAsync(dst, src, n16 * 16, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpy device to device: %.3f ms per copy, %.2f TB/s read + written\n", ms / 3, (double)n16 * 32 * 3 / (ms * 1e-3) / 1e12);
    return 0;
}
