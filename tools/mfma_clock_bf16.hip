// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 (random operands in registers, one wave per SIMD) and the
// shader clock it is held at -- what "100 % of the matrix pipe" is worth on this chip under load.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_bf16.hip -o /tmp/mfma_clock_bf16 && /tmp/mfma_clock_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    u32x4 a, b;
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (0x3f00u + ((s >> 20) & 0x7fu)) | ((0x3f00u + ((s >> 8) & 0x7fu)) << 16); };
    a = u32x4{rnd(), rnd(), rnd(), rnd()}; b = u32x4{rnd(), rnd(), rnd(), rnd()};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int wgs, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, wgs * 256 * 4); hipMalloc(&clk, wgs * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, out, clk, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double nm = (double)wgs * 4 * iters * 8 * NACC;
    double tf = nm * 32768 / (ms * 1e-3) / 1e12;
    printf("NACC=%d wgs=%d: %.3f ms  %.0f TFLOP/s  cycles/MFMA=%.1f  clock=%.2f GHz\n", NACC, wgs, ms, tf,
           (double)h[0] / (iters * 8.0 * NACC), (double)h[0] / ((double)h[1] / 100e6) / 1e9);
}
int main() {
    run<1>(256, 20000); run<2>(256, 10000); run<4>(256, 5000); run<8>(256, 2500); run<8>(512, 2500);
    return 0;
}
