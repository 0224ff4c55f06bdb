// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 and the shader clock it is held at.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int wgs, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, wgs * 256 * 4); hipMalloc(&clk, wgs * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(256), 0, 0, out, clk, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double nm = (double)wgs * 4 * iters * 16 * NACC;        // MFMAs
    double tf = nm * 4096 / (ms * 1e-3) / 1e12;
    double shader_clk = (double)h[0] / ((double)h[1] / 100e6) / 1e9;   // s_memrealtime ticks at 100 MHz
    printf("NACC=%d wgs=%d: %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD=%.1f  clock=%.2f GHz\n", NACC, wgs, ms, tf,
           (double)h[0] / (iters * 16.0 * NACC) * 1.0, shader_clk);
}
int main() {
    run<1>(256, 4000); run<2>(256, 2000); run<4>(256, 1000); run<1>(512, 4000); run<1>(1024, 2000);
    return 0;
}
