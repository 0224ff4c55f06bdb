// LDS staging helpers shared by the image-conv kernels.
//
// Every conv kernel keeps bands of NCHW rows in LDS as [channel][row][padded col].  Two rules
// make the copy fast on CDNA4: (1) the widest aligned global load the row width allows
// (16 B for W % 4 == 0, 8 B for even W, uchar4 for u8 images) and (2) many loads in flight per
// lane -- the loop is unrolled UF-fold with all loads issued before the first LDS store, so a
// lane has UF x 16 B outstanding instead of one dependent load per iteration.
#pragma once
#include "var_common.h"

template <int NT>
__device__ __forceinline__ void lds_zero(float* dst, int nfloats, int tid) {
    // nfloats is a multiple of 4 by construction of the callers (LDS carve rounded up)
    float4 z = {0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < nfloats / 4; e += NT) ((float4*)dst)[e] = z;
}

// Zero columns [c0, c0+nc) of `nrows` LDS rows of `stride` floats: the padding cells of a band layout whose
// data cells are all (re)written by the staging pass, so no full clear and no barrier before staging is needed.
template <int NT>
__device__ __forceinline__ void lds_zero_cols(float* dst, int nrows, int stride, int c0, int nc, int tid) {
    if (nc == 1) {                                       // the common case: no division per element
        for (int r = tid; r < nrows; r += NT) dst[r * stride + c0] = 0.f;
        return;
    }
    for (int e = tid; e < nrows * nc; e += NT) {
        const int r = e / nc, c = e - r * nc;
        dst[r * stride + c0 + c] = 0.f;
    }
}

// Copy rows [iy0, iy0+IR) x cols [0,W) of CIN planes of one NCHW image into dst[c*PLANE + r*PW + 1 + x].
// Rows outside [0,H) are written as zeros.  Column 0 and columns > W are NOT touched (pre-zeroed).
template <int CIN, int H, int W, int IR, int PW, int PLANE, bool U8, int NT, int UFMAX = 64>
__device__ __forceinline__ void stage_x_band(float* __restrict__ dst, const void* __restrict__ img,
                                             int iy0, bool uvalid, int tid) {
    constexpr int V = U8 ? 4 : (W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1));
    constexpr int WV = W / V;
    constexpr int TOT = CIN * IR * WV;
    // the whole band in ONE batch of loads when it fits the register budget (one memory latency, not several)
    constexpr int NEED = (TOT + NT - 1) / NT;
    constexpr int UF = NEED < UFMAX ? NEED : UFMAX;
    static_assert(W % V == 0, "row width must be a multiple of the vector width");
#pragma unroll 1
    for (int e0 = tid; e0 < TOT; e0 += NT * UF) {
        float v[UF][V];
        int off[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int e = e0 + u * NT;
            const bool ok = e < TOT;
            const int ee = ok ? e : 0;
            const int c = ee / (IR * WV);
            const int rem = ee - c * (IR * WV);
            const int r = rem / WV;
            const int xv = rem - r * WV;
            const int iy = iy0 + r;
            const bool valid = ok && uvalid && iy >= 0 && iy < H;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            const int so = (c * H + iyc) * W + xv * V;
            if constexpr (U8) {
                const uchar4 q = *(const uchar4*)((const uint8_t*)img + so);
                v[u][0] = (float)q.x / 255.f; v[u][1] = (float)q.y / 255.f;
                v[u][2] = (float)q.z / 255.f; v[u][3] = (float)q.w / 255.f;
            } else if constexpr (V == 4) {
                const float4 q = *(const float4*)((const float*)img + so);
                v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
            } else if constexpr (V == 2) {
                const float2 q = *(const float2*)((const float*)img + so);
                v[u][0] = q.x; v[u][1] = q.y;
            } else {
                v[u][0] = ((const float*)img)[so];
            }
            if (!valid) {
#pragma unroll
                for (int j = 0; j < V; ++j) v[u][j] = 0.f;
            }
            off[u] = ok ? c * PLANE + r * PW + 1 + xv * V : -1;
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            if (off[u] >= 0) {
#pragma unroll
                for (int j = 0; j < V; ++j) dst[off[u] + j] = v[u][j];
            }
        }
    }
}

// Copy rows [oy0, oy0+NR) x cols [0,WO) of COUT planes of one (COUT,HO,WO) gradient image into
// dst[n*PLANE + r*POW + x]; rows >= HO are written as zeros; columns >= WO are NOT touched.
template <int COUT, int HO, int WO, int NR, int POW, int PLANE, int NT, int UFMAX = 64>
__device__ __forceinline__ void stage_y_band(float* __restrict__ dst, const float* __restrict__ img,
                                             int oy0, bool uvalid, int tid) {
    constexpr int V = (WO % 4 == 0) ? 4 : (WO % 2 == 0 ? 2 : 1);
    constexpr int WV = WO / V;
    constexpr int TOT = COUT * NR * WV;
    constexpr int NEED = (TOT + NT - 1) / NT;
    constexpr int UF = NEED < UFMAX ? NEED : UFMAX;
#pragma unroll 1
    for (int e0 = tid; e0 < TOT; e0 += NT * UF) {
        float v[UF][V];
        int off[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int e = e0 + u * NT;
            const bool ok = e < TOT;
            const int ee = ok ? e : 0;
            const int n = ee / (NR * WV);
            const int rem = ee - n * (NR * WV);
            const int r = rem / WV;
            const int xv = rem - r * WV;
            const int oy = oy0 + r;
            const bool valid = ok && uvalid && oy < HO;
            const int oyc = oy >= HO ? HO - 1 : oy;
            const int so = (n * HO + oyc) * WO + xv * V;
            if constexpr (V == 4) {
                const float4 q = *(const float4*)(img + so);
                v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
            } else if constexpr (V == 2) {
                const float2 q = *(const float2*)(img + so);
                v[u][0] = q.x; v[u][1] = q.y;
            } else {
                v[u][0] = img[so];
            }
            if (!valid) {
#pragma unroll
                for (int j = 0; j < V; ++j) v[u][j] = 0.f;
            }
            off[u] = ok ? n * PLANE + r * POW + xv * V : -1;
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            if (off[u] >= 0) {
#pragma unroll
                for (int j = 0; j < V; ++j) dst[off[u] + j] = v[u][j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// ColStager: column-mapped, register-staged copy of a band of NCHW rows into LDS, in two phases
// (issue = global loads into registers, store = registers -> LDS) so that the loads of the NEXT
// tile fly while the CURRENT tile is on the matrix cores.
//   band  = ROWS rows x W cols of NCH planes of NUNITS images  ->  lds[u*UNIT + ch*PLANE + row*PW + COL0 + col]
//   lane  = PP (channel, V-wide column) pairs; it walks the ROWS rows of each pair, so global
//           addresses are base + row*W and LDS offsets base + row*PW (compile-time immediates),
//           and row validity (rows above/below the image) is wave-uniform.
// No VALU instruction touches a loaded value in issue(): the compiler therefore waits for the
// loads only in store().
// ------------------------------------------------------------------------------------------
template <int NCH, int H, int W, int ROWS, int PW, int PLANE, int COL0, bool U8, int NT, int NUNITS>
struct ColStager {
    static constexpr int V = U8 ? 4 : (W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1));
    static constexpr int WV = W / V;
    static constexpr int NP = NCH * WV;                       // pairs per unit
    static constexpr int PP = (NUNITS * NP + NT - 1) / NT;    // pairs per lane
    static constexpr int XREG = U8 ? 1 : V;
    static constexpr int UNIT = NCH * PLANE;
    int gcol[PP];      // ch*H*W + xv*V (elements from the unit's plane-0 row-0), -1 = no pair
    int lcol[PP];      // u*UNIT + ch*PLANE + COL0 + xv*V
    int ucol[PP];      // unit of the pair
    float data[PP][ROWS][XREG];

    __device__ __forceinline__ void init(int tid) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int e = tid + p * NT;
            if (e < NUNITS * NP) {
                const int u = e / NP, rem = e - u * NP;
                const int ch = rem / WV, xv = rem - ch * WV;
                gcol[p] = ch * H * W + xv * V;
                lcol[p] = u * UNIT + ch * PLANE + COL0 + xv * V;
                ucol[p] = u;
            } else {
                gcol[p] = -1; lcol[p] = 0; ucol[p] = 0;
            }
        }
    }
    // base: plane 0 / row 0 of the FIRST unit's image; further units (NUNITS > 1) are the following
    // images, `ustride` elements apart, and share row0.  nvalid = units that exist (>= 1 when called);
    // pairs of missing units load from unit 0 and are zeroed in store().
    template <class PT>
    __device__ __forceinline__ void issue(const PT* base, long ustride, int row0, int nvalid) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int gc = gcol[p] < 0 ? 0 : gcol[p];
            const PT* b = base + gc;
            if constexpr (NUNITS > 1) b += (size_t)(ucol[p] < nvalid ? ucol[p] : 0) * ustride;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = row0 + r;
                row = row < 0 ? 0 : (row >= H ? H - 1 : row);
                const PT* src = b + row * W;
                if constexpr (U8) {
                    data[p][r][0] = __uint_as_float(*(const uint32_t*)src);
                } else if constexpr (V == 4) {
                    const float4 q = *(const float4*)src;
                    data[p][r][0] = q.x; data[p][r][1] = q.y; data[p][r][2] = q.z; data[p][r][3] = q.w;
                } else if constexpr (V == 2) {
                    const float2 q = *(const float2*)src;
                    data[p][r][0] = q.x; data[p][r][1] = q.y;
                } else {
                    data[p][r][0] = *src;
                }
            }
        }
    }
    // rows outside [0,H) and units >= nvalid are written as zeros
    __device__ __forceinline__ void store(float* __restrict__ lds, int row0, int nvalid) const {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            if (gcol[p] < 0) continue;
            const bool ok = ucol[p] < nvalid;
            float* d = lds + lcol[p];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int row = row0 + r;
                const bool rok = ok && row >= 0 && row < H;
                if constexpr (U8) {
                    const uint32_t q = rok ? __float_as_uint(data[p][r][0]) : 0u;
                    d[r * PW + 0] = (float)(q & 0xff) / 255.f;
                    d[r * PW + 1] = (float)((q >> 8) & 0xff) / 255.f;
                    d[r * PW + 2] = (float)((q >> 16) & 0xff) / 255.f;
                    d[r * PW + 3] = (float)(q >> 24) / 255.f;
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j) d[r * PW + j] = rok ? data[p][r][j] : 0.f;
                }
            }
        }
    }
    // One-shot form for loader waves (no state kept between calls): all loads of the band, then the LDS stores.
    template <class PT>
    static __device__ __forceinline__ void copy(float* __restrict__ lds, const PT* base, int row0, bool live, int tid) {
        ColStager st;
        st.init(tid);
        st.issue(base, 0, row0, 1);
        st.store(lds, row0, live ? 1 : 0);
    }
};

// ------------------------------------------------------------------------------------------
// FlatStager: ColStager's job for the case "the band is the WHOLE map of NCH consecutive planes of one image" with an odd
// row width (21 x 21, 11 x 11 maps at 84 x 84): there ColStager can only move 4 bytes per lane and load (rows are not
// 8- or 16-byte aligned) -- 23 + 11 load instructions per lane and unit in the conv-3 weight gradient, and a CU accepts a
// vector-memory instruction only every ~35 cycles: the kernel was bound by load ISSUE (408 wave-loads per unit against
// 3.3 K cycles of matrix work).  The NCH planes are one contiguous, 16-byte aligned run in HBM: this stager reads it as
// float4 (84 wave-loads per unit for the same bytes) and scatters the four floats to their padded LDS cells, whose offsets
// are computed once per kernel.
//   lds[ch*PLANE + (row + ROW0)*PW + COL0 + col] = plane[ch][row][col]
// ------------------------------------------------------------------------------------------
template <int NCH, int H, int W, int PW, int PLANE, int ROW0, int COL0, int NT, int NUNITS = 1>
struct FlatStager {
    static constexpr int NF = NCH * H * W, NV = NF / 4;      // floats / float4 of one unit's run
    static constexpr int PP = (NUNITS * NV + NT - 1) / NT;
    static constexpr int UNIT = NCH * PLANE;
    static_assert(NF % 4 == 0, "the run must be whole float4");
    int off[PP][4];        // LDS offsets of the lane's floats
    float4 data[PP];
    int voff[PP];          // float4 index inside its unit's run (clamped for lanes past the end)
    int ucol[PP];          // unit (-1: no vector)

    __device__ __forceinline__ void init(int tid) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int v = tid + p * NT;
            const bool live = v < NUNITS * NV;
            const int u = live ? v / NV : 0;
            ucol[p] = live ? u : -1;
            voff[p] = live ? v - u * NV : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * voff[p] + j;
                const int ch = e / (H * W), rem = e - ch * (H * W);
                const int r = rem / W, x = rem - r * W;
                off[p][j] = u * UNIT + ch * PLANE + (r + ROW0) * PW + COL0 + x;
            }
        }
    }
    // base: the first unit's run; further units are the following images, `ustride` elements apart; units >= nvalid load
    // unit 0's bytes and are written as zeros
    template <class PT>
    __device__ __forceinline__ void issue(const PT* base, long ustride, int, int nvalid) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int u = ucol[p] > 0 && ucol[p] < nvalid ? ucol[p] : 0;
            data[p] = ((const float4*)(base + (size_t)u * ustride))[voff[p]];
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds, int, int nvalid) const {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            if (ucol[p] < 0) continue;
            const bool ok = ucol[p] < nvalid;
            lds[off[p][0]] = ok ? data[p].x : 0.f; lds[off[p][1]] = ok ? data[p].y : 0.f;
            lds[off[p][2]] = ok ? data[p].z : 0.f; lds[off[p][3]] = ok ? data[p].w : 0.f;
        }
    }
};

// ------------------------------------------------------------------------------------------
// Two-phase copy of NU bands (ROWS rows x W cols of NCH planes each) from global memory to LDS: issue() puts
// every load of the bands in flight into registers, store() writes them to LDS later -- the matrix work of the
// current tile runs in between.  NT / (NU*NCH) consecutive lanes walk one plane's band (contiguous in memory),
// so the index math is one div/mod per phase plus add-and-wrap per element; it is redone in both phases from
// the caller's opaque copy of tid so that nothing of it lives in registers across the tile loop.
// u8 sources become f32 through a 256-entry table in LDS (exact x / 255.f without a division per element).
template <int NCH, int H, int W, int ROWS, bool U8, int NT, int NU>
struct BandCopy {
    static constexpr int V = U8 ? 4 : (W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1));
    static constexpr int WV = W / V, VP = ROWS * WV;          // vectors per plane band
    static constexpr int PLANES = NU * NCH, TP = NT / PLANES;  // lanes per plane
    static constexpr int PP = (VP + TP - 1) / TP;
    static constexpr int XREG = U8 ? 1 : V;
    static_assert(NT % PLANES == 0, "lanes must split evenly over the planes");
    float data[PP][XREG];

    struct Lane { int u, ch, r0, x0; };
    static __device__ __forceinline__ Lane lane_of(int tid) {
        Lane l;
        const int plane = tid / TP, tp = tid - plane * TP;
        l.u = plane / NCH; l.ch = plane - l.u * NCH;
        l.r0 = tp / WV; l.x0 = tp - l.r0 * WV;
        return l;
    }
    // (row, vector column) of this lane's p-th vector; false when the band has no such vector
    static __device__ __forceinline__ bool elem(const Lane& l, int p, int& r, int& xv) {
        constexpr int DMAX = ((PP - 1) * TP) / WV;
        (void)DMAX;
        const int dr = (p * TP) / WV, dx = (p * TP) % WV;     // compile-time after unrolling
        xv = l.x0 + dx; r = l.r0 + dr;
        if (xv >= WV) { xv -= WV; ++r; }
        return r < ROWS;
    }
    template <class PT>
    __device__ __forceinline__ void issue(const PT* base0, const PT* base1, int row00, int row01, int tid) {
        const Lane l = lane_of(tid);
        const PT* base = (l.u ? base1 : base0) + l.ch * H * W;
        const int row0 = l.u ? row01 : row00;
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            int r, xv;
            elem(l, p, r, xv);
            int row = row0 + r;
            row = row < 0 ? 0 : (row >= H ? H - 1 : row);
            const PT* src = base + row * W + xv * V;
            if constexpr (U8) {
                data[p][0] = __uint_as_float(*(const uint32_t*)src);
            } else if constexpr (V == 4) {
                const float4 q = *(const float4*)src;
                data[p][0] = q.x; data[p][1] = q.y; data[p][2] = q.z; data[p][3] = q.w;
            } else if constexpr (V == 2) {
                const float2 q = *(const float2*)src;
                data[p][0] = q.x; data[p][1] = q.y;
            } else {
                data[p][0] = *src;
            }
        }
    }
    // dst[u*UNIT + ch*PLANE + r*PW + COL0 + col]; rows outside [0,H) and bands that do not exist become zeros
    template <int UNIT, int PLANE, int PW, int COL0>
    __device__ __forceinline__ void store(float* __restrict__ dst, const float* __restrict__ lut, int row00, int row01,
                                          bool ok0, bool ok1, int tid) const {
        const Lane l = lane_of(tid);
        float* d0 = dst + l.u * UNIT + l.ch * PLANE + COL0;
        const int row0 = l.u ? row01 : row00;
        const bool uok = l.u ? ok1 : ok0;
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            int r, xv;
            if (!elem(l, p, r, xv)) continue;
            const int row = row0 + r;
            const bool rok = uok && row >= 0 && row < H;
            float* d = d0 + r * PW + xv * V;
            if constexpr (U8) {
                const uint32_t q = rok ? __float_as_uint(data[p][0]) : 0u;      // lut[0] = 0
                d[0] = lut[q & 0xff];
                d[1] = lut[(q >> 8) & 0xff];
                d[2] = lut[(q >> 16) & 0xff];
                d[3] = lut[q >> 24];
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) d[j] = rok ? data[p][j] : 0.f;
            }
        }
    }
};
