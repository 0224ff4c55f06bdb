// Probe: how fast is a barrier + data hand-off between workgroups that sit on ONE XCD (shared L2) compared with the
// agent-scope (sc1) form that is correct across XCDs?  Also prints the blockIdx -> XCC_ID map of a 256-workgroup launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/xcd_sync_probe tools/probe/xcd_sync_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

// mode 0: agent scope (sc1 / agent atomics); mode 1: L2-local (sc0 loads, plain stores, L2 atomics)
template <int MODE>
__device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (MODE == 0) __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned ld(unsigned* p) {
    if (MODE == 0) return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void add1(unsigned* p) {
    if (MODE == 0) __hip_atomic_fetch_add((gu32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else { unsigned one = 1; asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(one) : "memory"); }
}

// participants: workgroups with (blockIdx % stride) == 0 and blockIdx / stride < n
template <int MODE>
__global__ void __launch_bounds__(64) probe(unsigned* sync, unsigned* slots, unsigned* xcc, unsigned* bad, long long* cyc, int stride, int n,
                                            int rounds) {
    const int tid = threadIdx.x;
    if (tid == 0) xcc[blockIdx.x] = xcc_id();
    if (blockIdx.x % stride != 0 || (int)blockIdx.x / stride >= n) return;
    const int me = blockIdx.x / stride;
    unsigned nbad = 0;
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (tid == 0) st<MODE>(slots + me * 32, (unsigned)r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            add1<MODE>(sync);
            unsigned spins = 0;
            while (ld<MODE>(sync) < (unsigned)(r * n)) { if (++spins > (1u << 20)) { nbad |= 0x80000000u; break; } }
        }
        __syncthreads();
        if (tid == 0) { const unsigned v = ld<MODE>(slots + ((me + 1) % n) * 32); if (v != (unsigned)r) ++nbad; }
    }
    if (tid == 0) { bad[me] = nbad; cyc[me] = clock64() - t0; }
}

int main() {
    unsigned *sync, *slots, *xcc, *bad;
    long long* cyc;
    hipMalloc(&sync, 256); hipMalloc(&slots, 256 * 32 * 4); hipMalloc(&xcc, 2048 * 4); hipMalloc(&bad, 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int rounds = 200;
    auto run = [&](int mode, int grid, int stride, int n, const char* what) {
        hipMemset(sync, 0, 256); hipMemset(slots, 0, 256 * 32 * 4); hipMemset(bad, 0, 256 * 4); hipMemset(xcc, 0xff, 2048 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 0, 0, sync, slots, xcc, bad, cyc, stride, n, rounds);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, sync, slots, xcc, bad, cyc, stride, n, rounds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> hb(256), hx(2048);
        hipMemcpy(hb.data(), bad, 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, 2048 * 4, hipMemcpyDeviceToHost);
        unsigned nb = 0; for (int i = 0; i < n; ++i) nb += hb[i] != 0;
        int same = 1; for (int i = 0; i < n; ++i) if (hx[i * stride] != hx[0]) same = 0;
        printf("%-52s %7.2f us/round  bad workgroups %u  participants on one XCC: %s (xcc of wg0 = %u)\n", what, 1e3 * ms / rounds, nb,
               same ? "yes" : "no", hx[0]);
        return hx;
    };
    auto hx = run(0, 256, 1, 32, "32 wgs on consecutive blockIdx, agent scope");
    printf("blockIdx -> XCC_ID (first 32): "); for (int i = 0; i < 32; ++i) printf("%u ", hx[i]); printf("\n");
    run(0, 256, 8, 32, "32 wgs blockIdx % 8 == 0, agent scope");
    run(1, 256, 8, 32, "32 wgs blockIdx % 8 == 0, L2-local ops");
    run(1, 256, 1, 32, "32 wgs consecutive blockIdx, L2-local ops (expect BAD)");
    run(0, 128, 1, 128, "128 wgs, agent scope (the chain's form)");
    run(0, 256, 8, 16, "16 wgs blockIdx % 8 == 0, agent scope");
    run(1, 256, 8, 16, "16 wgs blockIdx % 8 == 0, L2-local ops");
    return 0;
}
