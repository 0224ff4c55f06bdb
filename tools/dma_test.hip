// LDS-DMA semantics check (gfx950): global_load_lds_dwordx4 -- lane l's 16 bytes land at M0 base + 16 l ?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* o) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = -1.f;
    __syncthreads();
    // lane l of wave w reads g[1024 w' + 4 l'] with a permuted l' to see where each lane's data lands
    const int lp = (lane * 7) & 63;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + wave * 256 + 4 * lp),
                                     (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) o[i] = lds[i];
}
int main() {
    float *g, *o, h[1024], r[1024];
    for (int i = 0; i < 1024; i++) h[i] = (float)i;
    hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, g, o);
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int w = 0; w < 4; w++) for (int l = 0; l < 64; l++) for (int c = 0; c < 4; c++) {
        const int lp = (l * 7) & 63;
        if (r[w * 256 + 4 * l + c] != (float)(w * 256 + 4 * lp + c)) ok = 0;
    }
    printf("lane l lands at base + 16 l: %s; first: %g %g %g %g | %g %g\n", ok ? "YES" : "NO", r[0], r[1], r[2], r[3], r[4], r[8]);
    return 0;
}
