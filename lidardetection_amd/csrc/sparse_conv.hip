// Sparse 3D convolution (spconv v1.x semantics) on gfx950: rulebook build + output-stationary implicit GEMM.
// Replaces the external `spconv` package the reference imports (call sites: pcdet/models/backbones_3d/
// spconv_backbone.py:3-26,76-116; spconv_unet.py; roi_heads/partA2_head.py).
//
// Rulebook representation: a NEIGHBOUR TABLE nbr (N_out, K) i32 — nbr[j][k] = input row that reaches output
// row j through kernel offset k, or -1 — instead of spconv's per-offset (in, out) pair lists.  With it
//     out[j] = sum_k in[nbr[j][k]] @ W[k]
// is an output-stationary implicit GEMM: no scatter-add, no atomics, a fixed summation order (k ascending),
// one launch per layer instead of up to 81 (gather + SGEMM + scatter per offset).  The pair-list view is
// recovered by enumerating the valid table entries (tests check set equality with the brute-force oracle).
//
//   SubM   : outputs = inputs; nbr[j][k] = row of site_j + (k - centre), looked up in a coordinate hash table.
//   Regular: o = (q + p - k) / s for every (input q, offset k) that divides evenly and lands in bounds.  Output
//            rows are numbered in first-touch order of the (input row, offset) scan — deterministic (this is
//            the order upstream's CPU rulebook builder produces): atomicMin picks each output site's first
//            candidate, an exclusive scan over the candidate flags ranks them.
//   Inverse: the transposed table of the forward conv it is paired with (same offset index, no flip).
// GEMM: fp32 MFMA (v_mfma_f32_32x32x2_f32 == exact fp32 FMA chain), wave tile 32 rows x 32 cols, workgroup =
// 4 waves = 128 output rows; per offset the gathered rows (32 x Cin per wave) and W[k] (Cin x Cout, shared by
// the workgroup) are staged in LDS; offsets with no neighbour in a wave's tile are skipped.
#include "common.h"
#ifndef SC_PROBE                         // timing probes of sc_implicit_gemm_rega_kernel (WRONG results): bit 0 no row gathers, bit 1 no weight
#define SC_PROBE 0                       // fetch / LDS store, bit 2 no stage barrier
#endif
#include <stdlib.h>

#define SC_EMPTY 0xFFFFFFFFFFFFFFFFull

struct ScGeom {
    int B, D, H, W;          // input spatial shape
    int oD, oH, oW;          // output spatial shape
    int kD, kH, kW, K;
    int sD, sH, sW, pD, pH, pW;
};

__device__ __forceinline__ unsigned long long sc_key(int b, int z, int y, int x, int D, int H, int W) {
    return (((unsigned long long)b * D + z) * H + y) * W + x;
}

__device__ __forceinline__ unsigned sc_hash(unsigned long long key, unsigned mask) {
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdull;
    key ^= key >> 33;
    return (unsigned)key & mask;
}

// insert keys of `indices` (N,4) [b,z,y,x] -> table (keys, vals=row).  Duplicate coordinates keep the lowest row.
__global__ void sc_hash_build_kernel(const int *__restrict__ indices, int N, int D, int H, int W,
                                     unsigned long long *__restrict__ keys, int *__restrict__ vals, unsigned mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    const unsigned long long key = sc_key(c.x, c.y, c.z, c.w, D, H, W);
    unsigned h = sc_hash(key, mask);
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long old = atomicCAS(&keys[h], SC_EMPTY, key);
        if (old == SC_EMPTY || old == key) {
            atomicMin(&vals[h], i);
            return;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int sc_lookup(const unsigned long long *__restrict__ keys, const int *__restrict__ vals,
                                         unsigned mask, unsigned long long key) {
    unsigned h = sc_hash(key, mask);
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long k = keys[h];
        if (k == key) return vals[h];
        if (k == SC_EMPTY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

__global__ void sc_fill_kernel(unsigned long long *__restrict__ keys, int *__restrict__ vals, long long n, int valfill) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        keys[i] = SC_EMPTY;
        vals[i] = valfill;
    }
}

__global__ void sc_fill_i32_kernel(int *__restrict__ p, long long n, int v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------ SubM table
__global__ void sc_subm_table_kernel(const int *__restrict__ indices, int N, ScGeom g, const unsigned long long *__restrict__ keys,
                                     const int *__restrict__ vals, unsigned mask, int *__restrict__ nbr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)N * g.K) return;
    const int j = (int)(e / g.K), k = (int)(e - (long long)j * g.K);
    const int kz = k / (g.kH * g.kW), ky = (k / g.kW) % g.kH, kx = k % g.kW;
    const int4 c = reinterpret_cast<const int4 *>(indices)[j];
    const int z = c.y + kz - g.kD / 2, y = c.z + ky - g.kH / 2, x = c.w + kx - g.kW / 2;
    int r = -1;
    if (z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W) r = sc_lookup(keys, vals, mask, sc_key(c.x, z, y, x, g.D, g.H, g.W));
    nbr[e] = r;
}

// ------------------------------------------------------------------ regular conv: candidates -> unique outputs
// candidate c = in_row * K + k.  cand_slot[c] = slot of its output site in the output hash (or -1);
// owner[slot] = smallest candidate that produced the site.
__global__ void sc_candidates_kernel(const int *__restrict__ indices, int N, ScGeom g, unsigned long long *__restrict__ okeys,
                                     int *__restrict__ owner, unsigned omask, int *__restrict__ cand_slot) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)N * g.K) return;
    const int i = (int)(e / g.K), k = (int)(e - (long long)i * g.K);
    const int kz = k / (g.kH * g.kW), ky = (k / g.kW) % g.kH, kx = k % g.kW;
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    const int tz = c.y + g.pD - kz, ty = c.z + g.pH - ky, tx = c.w + g.pW - kx;
    int slot = -1;
    if (tz >= 0 && ty >= 0 && tx >= 0 && tz % g.sD == 0 && ty % g.sH == 0 && tx % g.sW == 0) {
        const int oz = tz / g.sD, oy = ty / g.sH, ox = tx / g.sW;
        if (oz < g.oD && oy < g.oH && ox < g.oW) {
            const unsigned long long key = sc_key(c.x, oz, oy, ox, g.oD, g.oH, g.oW);
            unsigned h = sc_hash(key, omask);
            for (unsigned probe = 0; probe <= omask; ++probe) {
                const unsigned long long old = atomicCAS(&okeys[h], SC_EMPTY, key);
                if (old == SC_EMPTY || old == key) {
                    slot = (int)h;
                    break;
                }
                h = (h + 1) & omask;
            }
            if (slot >= 0) atomicMin(&owner[slot], (int)e);
        }
    }
    cand_slot[e] = slot;
}

// three-kernel exclusive scan over candidate flags (flag = candidate owns its output site)
#define SCAN_TPB 1024
__global__ __launch_bounds__(SCAN_TPB) void sc_scan_blocks_kernel(const int *__restrict__ cand_slot, const int *__restrict__ owner,
                                                                  long long n, int *__restrict__ block_sums) {
    __shared__ int s_w[16];
    const long long e = (long long)blockIdx.x * SCAN_TPB + threadIdx.x;
    int f = 0;
    if (e < n) {
        const int s = cand_slot[e];
        f = (s >= 0 && owner[s] == (int)e) ? 1 : 0;
    }
    const int c = __popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0;
        for (int k = 0; k < 16; ++k) a += s_w[k];
        block_sums[blockIdx.x] = a;
    }
}

__global__ __launch_bounds__(1024) void sc_scan_sums_kernel(int *__restrict__ block_sums, int nblocks, int *__restrict__ total) {
    // single block: sequential chunks of 1024 with a carried prefix
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_sums[i] : 0;
        const int inc = wave_incl_scan(v);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) woff += s_w[k];
        const int carry = s_carry;
        if (i < nblocks) block_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// rank the owners, emit output coordinates and out_row[slot]
__global__ __launch_bounds__(SCAN_TPB) void sc_assign_outputs_kernel(const int *__restrict__ cand_slot, const int *__restrict__ owner,
                                                                     long long n, const int *__restrict__ block_sums,
                                                                     const unsigned long long *__restrict__ okeys, ScGeom g,
                                                                     int *__restrict__ out_row, int *__restrict__ out_indices) {
    __shared__ int s_w[16];
    const long long e = (long long)blockIdx.x * SCAN_TPB + threadIdx.x;
    int f = 0, s = -1;
    if (e < n) {
        s = cand_slot[e];
        f = (s >= 0 && owner[s] == (int)e) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (f) {
        int r = block_sums[blockIdx.x] + __popcll(bal & lanemask_lt());
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) r += s_w[k];
        out_row[s] = r;
        unsigned long long key = okeys[s];
        const int x = (int)(key % g.oW); key /= g.oW;
        const int y = (int)(key % g.oH); key /= g.oH;
        const int z = (int)(key % g.oD); key /= g.oD;
        reinterpret_cast<int4 *>(out_indices)[r] = make_int4((int)key, z, y, x);
    }
}

// forward table nbr (N_out, K) and (optionally) the transposed table nbr_t (N_in, K): nbr_t[i][k] = output row j
__global__ void sc_fill_tables_kernel(const int *__restrict__ cand_slot, const int *__restrict__ out_row, long long n, int K,
                                      int *__restrict__ nbr, int *__restrict__ nbr_t) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int s = cand_slot[e];
    int j = -1;
    if (s >= 0) {
        j = out_row[s];
        const int i = (int)(e / K), k = (int)(e - (long long)i * K);
        atomicMin(reinterpret_cast<unsigned *>(nbr) + (size_t)j * K + k, (unsigned)i);   // duplicate input coordinates: the lowest row
    }                                                                                      // (the table arrives filled with -1)
    if (nbr_t) nbr_t[e] = j;
}

// ------------------------------------------------------------------ C ABI: rulebooks
LIDAR_EXPORT size_t lidar_spconv_hash_capacity(int n) {
    size_t c = 1024;
    while (c < 2 * (size_t)(n > 0 ? n : 1)) c <<= 1;
    return c;
}

// table memory = capacity * 12 bytes: [keys u64 x cap][vals i32 x cap]
LIDAR_EXPORT int lidar_spconv_build_hash(const int *indices, int n, int D, int H, int W, void *table, size_t capacity,
                                         void *stream) {
    if (!table || n < 0 || (capacity & (capacity - 1)) || capacity < 2 * (size_t)(n > 0 ? n : 1)) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *keys = (unsigned long long *)table;
    int *vals = (int *)(keys + capacity);
    hipLaunchKernelGGL(sc_fill_kernel, dim3(divup(capacity, 1024) > 1024 ? 1024 : divup(capacity, 1024)), dim3(1024), 0, s, keys, vals,
                       (long long)capacity, 0x7FFFFFFF);
    if (n > 0)
        hipLaunchKernelGGL(sc_hash_build_kernel, dim3(divup(n, 256)), dim3(256), 0, s, indices, n, D, H, W, keys, vals,
                           (unsigned)(capacity - 1));
    return lidar_check_launch("lidar_spconv_build_hash");
}

// SubMConv3d rulebook: nbr (n, K) with K = kD*kH*kW (odd kernel sizes), spatial shape (D,H,W)
LIDAR_EXPORT int lidar_spconv_subm_table(const int *indices, int n, int D, int H, int W, int kD, int kH, int kW,
                                         const void *table, size_t capacity, int *nbr, void *stream) {
    if (n < 0 || kD <= 0 || kH <= 0 || kW <= 0) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!indices || !table || !nbr) return LIDAR_ERR_ARG;
    ScGeom g = {};
    g.D = D; g.H = H; g.W = W; g.kD = kD; g.kH = kH; g.kW = kW; g.K = kD * kH * kW;
    const unsigned long long *keys = (const unsigned long long *)table;
    const int *vals = (const int *)(keys + capacity);
    hipLaunchKernelGGL(sc_subm_table_kernel, dim3(divup((long long)n * g.K, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, g,
                       keys, vals, (unsigned)(capacity - 1), nbr);
    return lidar_check_launch("lidar_spconv_subm_table");
}

// distinct outputs <= n * prod ceil(k/s): the output hash is sized for that, not for the n*K candidates
static long long sc_out_bound(int n, int kD, int kH, int kW, int sD, int sH, int sW) {
    const long long b = (long long)(n > 0 ? n : 1) * divup(kD, sD) * divup(kH, sH) * divup(kW, sW);
    const long long nc = (long long)(n > 0 ? n : 1) * kD * kH * kW;
    return b < nc ? b : nc;
}

static size_t sc_conv_ws_layout(int n, int K, long long bound, size_t *cap_out) {
    const size_t nc = (size_t)(n > 0 ? n : 1) * K;
    const size_t cap = lidar_spconv_hash_capacity((int)(bound > 0x3FFFFFFF ? 0x3FFFFFFF : bound));
    if (cap_out) *cap_out = cap;
    return align_up(cap * 8, 256) + align_up(cap * 4, 256) * 2 + align_up(nc * 4, 256) + align_up((nc / SCAN_TPB + 2) * 4, 256) + 256;
}

LIDAR_EXPORT size_t lidar_spconv_conv_table_workspace_bytes(int n, int kD, int kH, int kW, int sD, int sH, int sW) {
    if (kD <= 0 || kH <= 0 || kW <= 0 || sD <= 0 || sH <= 0 || sW <= 0) return 0;
    return sc_conv_ws_layout(n, kD * kH * kW, sc_out_bound(n, kD, kH, kW, sD, sH, sW), nullptr);
}

// SparseConv3d rulebook, phase 1: unique output sites.  out_indices (out_cap, 4) receives the coordinates of the
// output rows (first-touch order), *num_out (device int) their number.  The caller reads num_out (one host sync),
// allocates nbr (num_out, K) and calls phase 2 with the same workspace.
LIDAR_EXPORT int lidar_spconv_conv_outputs(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW,
                                           int sD, int sH, int sW, int pD, int pH, int pW, int *out_indices, int out_cap,
                                           int *num_out, void *ws, size_t ws_bytes, void *stream) {
    if (n < 0 || batch <= 0 || kD <= 0 || kH <= 0 || kW <= 0 || sD <= 0 || sH <= 0 || sW <= 0) return LIDAR_ERR_ARG;
    if (!num_out) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(num_out, 0, 4, s) == hipSuccess ? LIDAR_OK : LIDAR_ERR_LAUNCH;
    if (!indices || !out_indices || !ws) return LIDAR_ERR_ARG;
    ScGeom g = {};
    g.B = batch; g.D = D; g.H = H; g.W = W; g.kD = kD; g.kH = kH; g.kW = kW; g.K = kD * kH * kW;
    g.sD = sD; g.sH = sH; g.sW = sW; g.pD = pD; g.pH = pH; g.pW = pW;
    g.oD = (D + 2 * pD - kD) / sD + 1; g.oH = (H + 2 * pH - kH) / sH + 1; g.oW = (W + 2 * pW - kW) / sW + 1;
    if (g.oD <= 0 || g.oH <= 0 || g.oW <= 0) return LIDAR_ERR_ARG;
    const long long nc = (long long)n * g.K;
    if (nc > 0x3FFFFFFF) return LIDAR_ERR_ARG;
    // an input site reaches at most prod ceil(k/s) outputs
    const long long bound = sc_out_bound(n, kD, kH, kW, sD, sH, sW);
    if (out_cap < bound) return LIDAR_ERR_ARG;
    size_t cap;
    if (ws_bytes < sc_conv_ws_layout(n, g.K, bound, &cap)) return LIDAR_ERR_WORKSPACE;
    char *p = (char *)ws;
    unsigned long long *okeys = (unsigned long long *)p; p += align_up(cap * 8, 256);
    int *owner = (int *)p; p += align_up(cap * 4, 256);
    int *out_row = (int *)p; p += align_up(cap * 4, 256);
    int *cand_slot = (int *)p; p += align_up((size_t)nc * 4, 256);
    int *block_sums = (int *)p;
    const int nblocks = divup(nc, SCAN_TPB);
    hipLaunchKernelGGL(sc_fill_kernel, dim3(divup(cap, 1024) > 1024 ? 1024 : divup(cap, 1024)), dim3(1024), 0, s, okeys, owner,
                       (long long)cap, 0x7FFFFFFF);
    hipLaunchKernelGGL(sc_candidates_kernel, dim3(divup(nc, 256)), dim3(256), 0, s, indices, n, g, okeys, owner, (unsigned)(cap - 1),
                       cand_slot);
    hipLaunchKernelGGL(sc_scan_blocks_kernel, dim3(nblocks), dim3(SCAN_TPB), 0, s, cand_slot, owner, nc, block_sums);
    hipLaunchKernelGGL(sc_scan_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nblocks, num_out);
    hipLaunchKernelGGL(sc_assign_outputs_kernel, dim3(nblocks), dim3(SCAN_TPB), 0, s, cand_slot, owner, nc, block_sums, okeys, g,
                       out_row, out_indices);
    return lidar_check_launch("lidar_spconv_conv_outputs");
}

// phase 2: nbr (num_out, K) forward table (filled with -1 here first) and nbr_t (n, K) transposed table
// (nbr_t[i][k] = output row reached from input i through offset k, or -1) — used by the inverse conv and dgrad.
LIDAR_EXPORT int lidar_spconv_conv_tables(int n, int kD, int kH, int kW, int sD, int sH, int sW, int num_out, int *nbr,
                                          int *nbr_t, void *ws, size_t ws_bytes, void *stream) {
    if (n < 0 || kD <= 0 || kH <= 0 || kW <= 0 || sD <= 0 || sH <= 0 || sW <= 0 || num_out < 0) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!nbr || !ws) return LIDAR_ERR_ARG;
    const int K = kD * kH * kW;
    size_t cap;
    if (ws_bytes < sc_conv_ws_layout(n, K, sc_out_bound(n, kD, kH, kW, sD, sH, sW), &cap)) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long nc = (long long)n * K;
    char *p = (char *)ws;
    p += align_up(cap * 8, 256);
    p += align_up(cap * 4, 256);
    int *out_row = (int *)p; p += align_up(cap * 4, 256);
    int *cand_slot = (int *)p;
    const long long nt = (long long)num_out * K;
    if (nt > 0)
        hipLaunchKernelGGL(sc_fill_i32_kernel, dim3(divup(nt, 1024) > 2048 ? 2048 : divup(nt, 1024)), dim3(1024), 0, s, nbr, nt, -1);
    hipLaunchKernelGGL(sc_fill_tables_kernel, dim3(divup(nc, 256)), dim3(256), 0, s, cand_slot, out_row, nc, K, nbr, nbr_t);
    return lidar_check_launch("lidar_spconv_conv_tables");
}

// ------------------------------------------------------------------ implicit GEMM (fp32 MFMA 32x32x2)
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define IG_ROWS 128          // rows per workgroup (4 waves x 32)
#define IG_MAX_C 128

// out (n_out, Cout) = sum_k in[nbr[., k]] @ W[k] (+ bias).  W is (K, Cin, Cout) row-major (spconv's
// (kD,kH,kW,Cin,Cout) parameter viewed flat).  NT = number of 32-column tiles (Cout <= 32*NT).
template <int NT>
__global__ __launch_bounds__(256) void sc_implicit_gemm_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out, int K,
                                                               int Cin, int Cout, const float *__restrict__ Wt,
                                                               const float *__restrict__ bias, const float *__restrict__ residual,
                                                               int relu, float *__restrict__ out) {
    extern __shared__ float s_mem[];
    const int Cp = Cin + 1;                       // padded row stride of the gathered tiles
    float *s_w = s_mem;                           // [Cin][NT*32]
    float *s_a = s_mem + (size_t)Cin * NT * 32;   // [4 waves][32][Cp]
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    float *A = s_a + (size_t)wv * 32 * Cp;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int ar = l & 31, ak = l >> 5;           // MFMA operand coordinates of this lane
    const int CW = NT * 32;
    for (int k = 0; k < K; ++k) {
        __syncthreads();                          // previous offset's LDS reads are done
        // W[k] -> LDS (zero padded to NT*32 columns); 16-B loads when the row length allows
        if ((Cout & 3) == 0) {
            const int CW4 = CW >> 2, Co4 = Cout >> 2;
            for (int e = t; e < Cin * CW4; e += 256) {
                const int ci = e / CW4, q4 = e - ci * CW4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q4 < Co4) v = reinterpret_cast<const float4 *>(Wt + ((size_t)k * Cin + ci) * Cout)[q4];
                reinterpret_cast<float4 *>(s_w + (size_t)ci * CW)[q4] = v;
            }
        } else {
            for (int e = t; e < Cin * CW; e += 256) {
                const int ci = e / CW, co = e - ci * CW;
                s_w[e] = co < Cout ? Wt[((size_t)k * Cin + ci) * Cout + co] : 0.f;
            }
        }
        // this wave's 32 gathered rows -> LDS
        const int myrow = row0 + (l & 31);
        const int src = (myrow < n_out) ? nbr[(size_t)myrow * K + k] : -1;
        const bool any = __ballot(src >= 0) != 0ull;
        if (any) {
            if ((Cin & 3) == 0) {
                // 16-B gathers: all of a lane's loads are issued before the first LDS store (independent, in flight together)
                const int C4 = Cin >> 2;
                const int nq = 32 * C4;                 // float4 pieces of the 32 x Cin tile
                for (int it0 = 0; it0 * 64 < nq; it0 += 4) {   // 4 independent 16-B loads per lane per pass
                    float4 v[4];
                    int rr[4], cc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (it0 + u) * 64 + l;
                        rr[u] = e / C4;
                        cc[u] = e - rr[u] * C4;
                        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        const int sr = __shfl(src, rr[u] & 31, 64);
                        if (e < nq && sr >= 0) v[u] = reinterpret_cast<const float4 *>(in + (size_t)sr * Cin)[cc[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (it0 + u) * 64 + l;
                        if (e < nq) {
                            float *dst = A + rr[u] * Cp + cc[u] * 4;
                            dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                        }
                    }
                }
            } else {
                for (int e = l; e < 32 * Cin; e += 64) {
                    const int r = e / Cin, c = e - r * Cin;
                    const int sr = __shfl(src, r, 64);
                    A[r * Cp + c] = sr >= 0 ? in[(size_t)sr * Cin + c] : 0.f;
                }
            }
        }
        __syncthreads();
        if (any) {
            for (int c0 = 0; c0 < Cin; c0 += 2) {
                const float a = (c0 + ak < Cin) ? A[ar * Cp + c0 + ak] : 0.f;
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const float b = (c0 + ak < Cin) ? s_w[(c0 + ak) * CW + q * 32 + ar] : 0.f;
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int col = q * 32 + (l & 31);
        if (col < Cout) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out) {              // fused epilogue: (+ bias) (+ residual) (ReLU)
                    float v = acc[q][r] + bv;
                    if (residual) v += residual[(size_t)row * Cout + col];
                    out[(size_t)row * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// Software-pipelined variant for Cin = 4*C4 in {16, 32, 64, 128} and Cout % 4 == 0: the gathered rows and W[k+1] are
// fetched into registers while the MFMAs of offset k run out of LDS (global latency hidden behind the matrix pipe).
// SL > 1: the input row is consumed in SL slices of 4*C4 channels, one LDS stage each (Cin total = SL * 4 * C4): for 128 input
// channels the one-slice stage needs 130 KB of LDS and 276 registers (ONE workgroup per CU, one wave per SIMD); two 64-channel
// slices fit twice.  Channels are still accumulated in ascending order per offset, so the sums do not change by a bit.
template <int NT, int C4, int SL = 1>
__global__ __launch_bounds__(256) void sc_implicit_gemm_pipe_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                    int K, int Cout, const float *__restrict__ Wt,
                                                                    const float *__restrict__ bias, const float *__restrict__ residual,
                                                                    int relu, float *__restrict__ out,
                                                                    const int *__restrict__ row_mask, const int *__restrict__ out_row) {
    constexpr int Cin = C4 * 4, Cp = Cin + 1, CW = NT * 32, CW4 = CW / 4, CinT = Cin * SL;
    constexpr int NG = (32 * C4) / 64;                    // float4 gathers per lane per stage
    constexpr int NW = (Cin * CW4 + 255) / 256;           // float4 weight pieces per thread per offset
    extern __shared__ float s_mem[];
    float *s_w = s_mem;                                   // [Cin][CW]
    float *s_a = s_mem + (size_t)Cin * CW;                // [4 waves][32][Cp]
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    float *A = s_a + (size_t)wv * 32 * Cp;
    const int myrow = row0 + (l & 31);
    // table row this lane-row reads: the permutation entry for mask-sorted execution (out_row), else the row itself
    const int trow = (out_row && myrow < n_out) ? out_row[myrow] : myrow;
    const int Co4 = Cout >> 2;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int ar = l & 31, ak = l >> 5;
    float4 g[NG], wr[NW];
    bool any_next;
    // Mask-sorted tables (row_mask != NULL, K <= 32): rows arrive grouped by their neighbour-offset bit mask, so the 32 rows
    // of a wave tile (and the 128 of the workgroup) mostly share one mask: offsets no row of the WORKGROUP uses are skipped
    // outright (no W staging, no barriers) and a wave's MFMAs run only for offsets its own rows use.
    __shared__ unsigned s_wmask[4];
    unsigned wave_mask = 0xffffffffu, wg_mask = 0xffffffffu;
    if (row_mask) {
        unsigned m = (myrow < n_out) ? (unsigned)row_mask[trow] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
        wave_mask = m;
        if (l == 0) s_wmask[wv] = m;
        __syncthreads();
        wg_mask = s_wmask[0] | s_wmask[1] | s_wmask[2] | s_wmask[3];
    }
    auto next_k = [&](int k) {                            // next offset > k some row of the workgroup uses (K when none)
        if (!row_mask) return k + 1;
        const unsigned rest = (k + 1 < 32) ? (wg_mask >> (k + 1)) : 0u;
        return rest ? k + 1 + __builtin_ctz(rest) : K;
    };
    // table entries are requested one offset ahead of the gathers that use them (two ahead of the MFMAs), so no wave waits
    // for a table load with nothing else in flight
    auto load_src = [&](int k) { return (myrow < n_out && k < K) ? nbr[(size_t)trow * K + k] : -1; };
    auto fetch = [&](int k, int h, const int src) {       // stage (offset k, channel slice h)
        any_next = row_mask ? ((wave_mask >> k) & 1u) != 0u : __ballot(src >= 0) != 0ull;
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int e = u * 64 + l, r = e / C4, c = e - r * C4;
            const int sr = __shfl(src, r, 64);
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any_next && sr >= 0) g[u] = reinterpret_cast<const float4 *>(in + (size_t)sr * CinT + h * Cin)[c];
        }
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 256 + t, ci = e / CW4, q4 = e - ci * CW4;
            wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Cin * CW4 && q4 < Co4) wr[q] = reinterpret_cast<const float4 *>(Wt + ((size_t)k * CinT + h * Cin + ci) * Cout)[q4];
        }
    };
    int k = next_k(-1), h = 0;
    int src_cur = load_src(k), src_ahead = -1;            // table entries of the offset being fetched / of the one after it
    if (k < K) {
        fetch(k, 0, src_cur);
        src_ahead = load_src(next_k(k));
    }
    for (; k < K;) {
        int kn = k, hn = h + 1;                           // the stage after this one
        if (hn == SL) { kn = next_k(k); hn = 0; }
        const bool any = any_next;
        __syncthreads();                                  // LDS of the previous offset is no longer read
        // a workgroup-uniform "somebody needs W[k]" is not known per wave: every wave stores its W pieces whenever IT has work;
        // waves without work skip both the stores and the MFMAs, so W[k] must be written by the waves that do compute.
        if (any) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int e = u * 64 + l, r = e / C4, c = e - r * C4;
                float *dst = A + r * Cp + c * 4;
                dst[0] = g[u].x; dst[1] = g[u].y; dst[2] = g[u].z; dst[3] = g[u].w;
            }
        }
        {   // W[k]: pieces are distributed over all 256 threads, so every thread must hold them -> fetched unconditionally below
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const int e = q * 256 + t;
                if (e < Cin * CW4) reinterpret_cast<float4 *>(s_w)[e] = wr[q];
            }
        }
        __syncthreads();
        if (kn < K) {                                     // in flight while the MFMAs below run
            if (kn != k) {
                src_cur = src_ahead;
                src_ahead = load_src(next_k(kn));
            }
            fetch(kn, hn, src_cur);
        }
        if (any) {
#pragma unroll 4
            for (int c0 = 0; c0 < Cin; c0 += 2) {
                const float a = A[ar * Cp + c0 + ak];
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const float b = s_w[(c0 + ak) * CW + q * 32 + ar];
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                }
            }
        }
        k = kn;
        h = hn;
    }
    const int last = n_out - 1;                           // n_out >= 1 (checked by the launcher)
    // fused epilogue: (+ bias) (+ residual) (ReLU); sorted tables scatter rows.  Output row indices and residual values are
    // requested eight rows at a time before any is used (clamped, unconditional loads: no wait per element).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int orow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = h * 8 + j;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last);
            orow[j] = out_row ? out_row[row] : row;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = q * 32 + (l & 31);
            const int cc = min(col, Cout - 1);
            const float bv = bias ? bias[cc] : 0.f;
            float res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] = 0.f;
            if (residual) {
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = residual[(size_t)orow[j] * Cout + cc];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = h * 8 + j;
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out && col < Cout) {
                    const float v = acc[q][r] + bv + res[j];
                    out[(size_t)orow[j] * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// Register-A variant (r03; Cin = 4*C4*SL in {16, 32, 64, 128}, Cout % 4 == 0) — the kernel the launcher uses.
// The MFMA's A operand wants lane (row r = lane & 31, half a = lane >> 5) to supply channel (step, a) of row r.  The
// pipe kernel above gathers rows with coalescing lanes, transposes them through LDS (four ds_write_b32 per float4 at an odd
// pitch) and reads them back one float per MFMA step, two barriers per offset.  Any fixed assignment of channels to
// (step, half) gives the same products summed in a fixed order, so here half a of row r simply OWNS the 16-byte pieces a, a + 2,
// a + 4, ... of the row (r04; r03: the contiguous half [a * Cin/2, (a + 1) * Cin/2) — then one gather instruction touched every
// 64-byte sector of a row with a single 16-byte piece and each sector was requested by four different instructions; interleaved, the
// two halves of a row read neighbouring pieces in the same instruction and a sector is requested twice): lane (r, a) loads those
// Cin/2 floats of its gathered row straight into registers and feeds them to the MFMAs — no LDS for A, no transpose, no cross-lane shuffles of the table
// entries.  Only W[k] goes through LDS, double-buffered: the pieces of W[k+1] and the gathered rows of offset k+1 are in flight
// while the MFMAs of offset k run, and ONE barrier per offset separates a buffer's last reader from its next writer.
// Per output row the channels are accumulated in a different (but fixed) order than in the pipe kernel: results differ from
// it in the last bits, and are bit-identical between table order and mask order as before (same kernel, same order).
template <int NT, int C4, int SL = 1>
__global__ __launch_bounds__(256) void sc_implicit_gemm_rega_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                    int K, int Cout, const float *__restrict__ Wt,
                                                                    const float *__restrict__ bias, const float *__restrict__ residual,
                                                                    int relu, float *__restrict__ out,
                                                                    const int *__restrict__ row_mask, const int *__restrict__ out_row) {
    constexpr int Cin = C4 * 4, HALF = Cin / 2, CW = NT * 32, CW4 = CW / 4, CinT = Cin * SL;
    constexpr int NG = HALF / 4;                          // float4 gathers per lane per stage
    constexpr int NW = (Cin * CW4 + 255) / 256;           // float4 weight pieces per thread per stage
    extern __shared__ float s_mem[];                      // W stage buffers: [2][Cin][CW]
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    const int ar = l & 31, ak = l >> 5;
    const int myrow = row0 + ar;
    const int trow = (out_row && myrow < n_out) ? out_row[myrow] : myrow;
    const int Co4 = Cout >> 2;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __shared__ unsigned s_wmask[4];
    unsigned wave_mask = 0xffffffffu, wg_mask = 0xffffffffu;
    if (row_mask) {                                       // mask-sorted tables: see the pipe kernel
        unsigned m = (myrow < n_out) ? (unsigned)row_mask[trow] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
        wave_mask = m;
        if (l == 0) s_wmask[wv] = m;
        __syncthreads();
        wg_mask = s_wmask[0] | s_wmask[1] | s_wmask[2] | s_wmask[3];
    }
    auto next_k = [&](int k) {                            // next offset > k some row of the workgroup uses (K when none)
        if (!row_mask) return k + 1;
        const unsigned rest = (k + 1 < 32) ? (wg_mask >> (k + 1)) : 0u;
        return rest ? k + 1 + __builtin_ctz(rest) : K;
    };
    auto load_src = [&](int k) { return (myrow < n_out && k < K) ? nbr[(size_t)trow * K + k] : -1; };
    auto wave_uses = [&](int k, int src) { return row_mask ? ((wave_mask >> k) & 1u) != 0u : __ballot(src >= 0) != 0ull; };
    float4 wr[NW], ga[NG], gn[NG];
    auto fetch_w = [&](int k, int h) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 256 + t, ci = e / CW4, q4 = e - ci * CW4;
            wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Cin * CW4 && q4 < Co4 && !(SC_PROBE & 2)) wr[q] = reinterpret_cast<const float4 *>(Wt + ((size_t)k * CinT + h * Cin + ci) * Cout)[q4];
        }
    };
    auto store_w = [&](int buf) {
        float4 *dst = reinterpret_cast<float4 *>(s_mem + (size_t)buf * Cin * CW);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 256 + t;
            if (e < Cin * CW4 && !(SC_PROBE & 2)) dst[e] = wr[q];
        }
    };
    auto fetch_a = [&](float4 (&g)[NG], int h, int src, bool any) {
        const float4 *rowp = reinterpret_cast<const float4 *>(in + (size_t)max(src, 0) * CinT + h * Cin) + ak;    // pieces ak, ak + 2, ...: see the ownership note
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any && src >= 0 && !(SC_PROBE & 1)) g[u] = rowp[2 * u];
        }
    };
    int k = next_k(-1), h = 0, buf = 0;
    if (k >= K) k = K;
    int src_cur = load_src(k), src_ahead = -1;
    bool any = false;
    if (k < K) {
        any = wave_uses(k, src_cur);
        fetch_w(k, 0);
        fetch_a(ga, 0, src_cur, any);
        src_ahead = load_src(next_k(k));
        store_w(0);
    }
    for (; k < K;) {
        int kn = k, hn = h + 1;                           // the stage after this one
        if (hn == SL) { kn = next_k(k); hn = 0; }
        if (!(SC_PROBE & 4)) __syncthreads();             // W of this stage is visible; the other buffer's readers are done
        int src_next = src_cur;
        bool any_next = false;
        if (kn < K) {                                     // in flight while the MFMAs below run
            if (kn != k) {
                src_next = src_ahead;
                src_ahead = load_src(next_k(kn));
            }
            any_next = wave_uses(kn, src_next);
            fetch_w(kn, hn);
            fetch_a(gn, hn, src_next, any_next);
        }
        if (any) {
            const float *Wb = s_mem + (size_t)buf * Cin * CW + (size_t)ak * 4 * CW + ar;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float av[4] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], Wb[(u * 8 + j) * CW + q * 32], acc[q], 0, 0, 0);
                }
            }
        }
        if (kn < K) {
            store_w(buf ^ 1);
#pragma unroll
            for (int u = 0; u < NG; ++u) ga[u] = gn[u];
        }
        src_cur = src_next;
        any = any_next;
        k = kn;
        h = hn;
        buf ^= 1;
    }
    const int last = n_out - 1;                           // n_out >= 1 (checked by the launcher)
    // fused epilogue: (+ bias) (+ residual) (ReLU); sorted tables scatter rows (as in the pipe kernel)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        int orow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = hh * 8 + j;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last);
            orow[j] = out_row ? out_row[row] : row;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = q * 32 + (l & 31);
            const int cc = min(col, Cout - 1);
            const float bv = bias ? bias[cc] : 0.f;
            float res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] = 0.f;
            if (residual) {
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = residual[(size_t)orow[j] * Cout + cc];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = hh * 8 + j;
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out && col < Cout) {
                    const float v = acc[q][r] + bv + res[j];
                    out[(size_t)orow[j] * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// Two row tiles per wave (r04, Cin <= 64): the register-A kernel with every wave owning 64 rows — row r and row r + 32 per lane pair.
// A B operand read from LDS feeds two MFMAs, a staged W[k] serves 256 rows and the stage barrier comes once per 2 x NT x Cin / 2
// MFMAs per wave: the 64 -> 64 layers ran at 0.37 of the peak against 0.61 for the 128 -> 128 layers, whose stages carry twice the MFMAs
// for the same staging and barrier (profiles/r04/spconv_gemm_ablation.log: barrier 8 %, staging 4 %).  One workgroup per CU at these
// register counts (the second set of gathered rows), so nothing hides behind another workgroup — and that is what it costs: 45 TF where
// the one-tile kernel reaches 68 on the 64 -> 64 layers (2.89 vs 1.99 ms on the SECOND stack).  Correct (bit-identical, under test via
// LIDAR_SPCONV_RT2=1), off by default; kept as the measured answer to "more MFMAs per barrier".
// Row tile rt of a wave is skipped for an offset none of its 32 rows uses (wave-uniform), as before.  Same channel ownership and
// summation order per row as the register-A kernel: bit-identical results.
template <int NT, int C4>
__global__ __launch_bounds__(256) void sc_implicit_gemm_rega2_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                     int K, int Cout, const float *__restrict__ Wt,
                                                                     const float *__restrict__ bias, const float *__restrict__ residual,
                                                                     int relu, float *__restrict__ out,
                                                                     const int *__restrict__ row_mask, const int *__restrict__ out_row) {
    constexpr int Cin = C4 * 4, HALF = Cin / 2, CW = NT * 32, CW4 = CW / 4;
    constexpr int NG = HALF / 4;                          // float4 gathers per lane, row and stage
    constexpr int NW = (Cin * CW4 + 255) / 256;           // float4 weight pieces per thread per stage
    extern __shared__ float s_mem[];                      // W stage buffers: [2][Cin][CW]
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int row0 = blockIdx.x * (2 * IG_ROWS) + wv * 64;
    const int ar = l & 31, ak = l >> 5;
    int myrow[2], trow[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        myrow[rt] = row0 + 32 * rt + ar;
        trow[rt] = (out_row && myrow[rt] < n_out) ? out_row[myrow[rt]] : myrow[rt];
    }
    const int Co4 = Cout >> 2;
    f32x16 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][q][r] = 0.f;
    __shared__ unsigned s_wmask[4];
    unsigned tile_mask[2] = {0xffffffffu, 0xffffffffu}, wg_mask = 0xffffffffu;
    if (row_mask) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            unsigned m = (myrow[rt] < n_out) ? (unsigned)row_mask[trow[rt]] : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
            tile_mask[rt] = m;
        }
        if (l == 0) s_wmask[wv] = tile_mask[0] | tile_mask[1];
        __syncthreads();
        wg_mask = s_wmask[0] | s_wmask[1] | s_wmask[2] | s_wmask[3];
    }
    auto next_k = [&](int k) {
        if (!row_mask) return k + 1;
        const unsigned rest = (k + 1 < 32) ? (wg_mask >> (k + 1)) : 0u;
        return rest ? k + 1 + __builtin_ctz(rest) : K;
    };
    auto load_src = [&](int rt, int k) { return (myrow[rt] < n_out && k < K) ? nbr[(size_t)trow[rt] * K + k] : -1; };
    auto tile_uses = [&](int rt, int k, int src) { return row_mask ? ((tile_mask[rt] >> k) & 1u) != 0u : __ballot(src >= 0) != 0ull; };
    float4 wr[NW], ga[2][NG], gn[2][NG];
    auto fetch_w = [&](int k) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 256 + t, ci = e / CW4, q4 = e - ci * CW4;
            wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Cin * CW4 && q4 < Co4) wr[q] = reinterpret_cast<const float4 *>(Wt + ((size_t)k * Cin + ci) * Cout)[q4];
        }
    };
    auto store_w = [&](int buf) {
        float4 *dst = reinterpret_cast<float4 *>(s_mem + (size_t)buf * Cin * CW);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 256 + t;
            if (e < Cin * CW4) dst[e] = wr[q];
        }
    };
    auto fetch_a = [&](float4 (&g)[NG], int src, bool any) {
        const float4 *rowp = reinterpret_cast<const float4 *>(in + (size_t)max(src, 0) * Cin) + ak;      // pieces ak, ak + 2, ...
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any && src >= 0) g[u] = rowp[2 * u];
        }
    };
    int k = next_k(-1), buf = 0;
    if (k >= K) k = K;
    int src_cur[2] = {load_src(0, k), load_src(1, k)}, src_ahead[2] = {-1, -1};
    bool any[2] = {false, false};
    if (k < K) {
        fetch_w(k);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            any[rt] = tile_uses(rt, k, src_cur[rt]);
            fetch_a(ga[rt], src_cur[rt], any[rt]);
            src_ahead[rt] = load_src(rt, next_k(k));
        }
        store_w(0);
    }
    for (; k < K;) {
        const int kn = next_k(k);
        __syncthreads();                                  // W of this stage is visible; the other buffer's readers are done
        int src_next[2] = {src_cur[0], src_cur[1]};
        bool any_next[2] = {false, false};
        if (kn < K) {                                     // in flight while the MFMAs below run
            const int kk = next_k(kn);
            fetch_w(kn);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                src_next[rt] = src_ahead[rt];
                src_ahead[rt] = load_src(rt, kk);
                any_next[rt] = tile_uses(rt, kn, src_next[rt]);
                fetch_a(gn[rt], src_next[rt], any_next[rt]);
            }
        }
        const float *Wb = s_mem + (size_t)buf * Cin * CW + (size_t)ak * 4 * CW + ar;
        if (any[0] && any[1]) {                           // the usual case: one B read, two MFMAs
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float a0[4] = {ga[0][u].x, ga[0][u].y, ga[0][u].z, ga[0][u].w};
                const float a1[4] = {ga[1][u].x, ga[1][u].y, ga[1][u].z, ga[1][u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const float b = Wb[(u * 8 + j) * CW + q * 32];
                        acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b, acc[0][q], 0, 0, 0);
                        acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b, acc[1][q], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (!any[rt]) continue;                   // wave-uniform
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    const float av[4] = {ga[rt][u].x, ga[rt][u].y, ga[rt][u].z, ga[rt][u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int q = 0; q < NT; ++q)
                            acc[rt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], Wb[(u * 8 + j) * CW + q * 32], acc[rt][q], 0, 0, 0);
                    }
                }
            }
        }
        if (kn < K) {
            store_w(buf ^ 1);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int u = 0; u < NG; ++u) ga[rt][u] = gn[rt][u];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            src_cur[rt] = src_next[rt];
            any[rt] = any_next[rt];
        }
        k = kn;
        buf ^= 1;
    }
    const int last = n_out - 1;                           // n_out >= 1 (checked by the launcher)
    // fused epilogue, per row tile: (+ bias) (+ residual) (ReLU); sorted tables scatter rows (as in the register-A kernel)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int rbase = row0 + 32 * rt;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            int orow[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = hh * 8 + j;
                const int row = min(rbase + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last);
                orow[j] = out_row ? out_row[row] : row;
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int col = q * 32 + (l & 31);
                const int cc = min(col, Cout - 1);
                const float bv = bias ? bias[cc] : 0.f;
                float res[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = 0.f;
                if (residual) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) res[j] = residual[(size_t)orow[j] * Cout + cc];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = hh * 8 + j;
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    if (row < n_out && col < Cout) {
                        const float v = acc[rt][q][r] + bv + res[j];
                        out[(size_t)orow[j] * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
        }
    }
}

// Packed-weight variant of the register-A kernel (r04; same shapes, same channel ownership, same summation order: results are
// bit-identical to the register-A kernel's).  What changes is how W[k] reaches the matrix cores.  The register-A kernel fetched
// W[k+1] into registers (wr[NW]), stored it to LDS behind the MFMAs (store_w) and read it back ONE FLOAT PER MFMA (64 ds_read_b32
// per 64->64 stage, each with its own wait).  Here the folded weights are packed ONCE PER WEIGHT UPDATE (lidar_spconv_pack_weights)
// in exactly the order the lanes consume them —
//     P[k][h][u][q][lane = ak * 32 + ar][j] = W[k][h * Cin + 4 (2 u + ak) + j][32 q + ar]      (float4 over j; 0 past Cout)
// — so that (a) a stage is a straight copy and goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no store
// phase; NG * NT wave-instructions of 1 KB per stage), and (b) a lane reads the B operands of FOUR consecutive MFMA steps with
// one conflict-free ds_read_b128 (lane-linear image).  One barrier per stage as before (it also drains the DMA: vmcnt(0)).
template <int NT, int C4, int SL = 1>
__global__ __launch_bounds__(256) void sc_implicit_gemm_pk_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                  int K, int Cout, const float *__restrict__ Wp,
                                                                  const float *__restrict__ bias, const float *__restrict__ residual,
                                                                  int relu, float *__restrict__ out,
                                                                  const int *__restrict__ row_mask, const int *__restrict__ out_row) {
    constexpr int Cin = C4 * 4, HALF = Cin / 2, CinT = Cin * SL;
    constexpr int NG = HALF / 4;                          // float4 gathers per lane per stage = groups of four MFMA steps
    constexpr int NI = NG * NT;                           // 1 KB pieces (DMA wave-instructions) per stage
    constexpr int NIW = (NI + 3) / 4;                     // ... per wave
    extern __shared__ float4 s_w4[];                      // W stage buffers: [2][NI][64 lanes] float4
    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    const int ar = l & 31, ak = l >> 5;
    const int myrow = row0 + ar;
    const int trow = (out_row && myrow < n_out) ? out_row[myrow] : myrow;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __shared__ unsigned s_wmask[4];
    unsigned wave_mask = 0xffffffffu, wg_mask = 0xffffffffu;
    if (row_mask) {
        unsigned m = (myrow < n_out) ? (unsigned)row_mask[trow] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
        wave_mask = m;
        if (l == 0) s_wmask[wv] = m;
        __syncthreads();
        wg_mask = s_wmask[0] | s_wmask[1] | s_wmask[2] | s_wmask[3];
    }
    auto next_k = [&](int k) {
        if (!row_mask) return k + 1;
        const unsigned rest = (k + 1 < 32) ? (wg_mask >> (k + 1)) : 0u;
        return rest ? k + 1 + __builtin_ctz(rest) : K;
    };
    auto load_src = [&](int k) { return (myrow < n_out && k < K) ? nbr[(size_t)trow * K + k] : -1; };
    auto wave_uses = [&](int k, int src) { return row_mask ? ((wave_mask >> k) & 1u) != 0u : __ballot(src >= 0) != 0ull; };
    float4 ga[NG], gn[NG];
    auto dma_w = [&](int k, int h, int buf) {             // stage (k, h) -> buffer buf: piece i = wv + 4 e by wave wv
        const float4 *srcp = reinterpret_cast<const float4 *>(Wp) + ((size_t)(k * SL + h) * NI) * 64 + l;
#pragma unroll
        for (int e = 0; e < NIW; ++e) {
            const int i = wv + 4 * e;
            if (i < NI)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcp + (size_t)i * 64),
                                                 (__attribute__((address_space(3))) void *)(s_w4 + ((size_t)buf * NI + i) * 64), 16, 0, 0);
        }
    };
    auto fetch_a = [&](float4 (&g)[NG], int h, int src, bool any) {
        const float4 *rowp = reinterpret_cast<const float4 *>(in + (size_t)max(src, 0) * CinT + h * Cin) + ak;    // pieces ak, ak + 2, ...: see the ownership note
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any && src >= 0) g[u] = rowp[2 * u];
        }
    };
    int k = next_k(-1), h = 0, buf = 0;
    if (k >= K) k = K;
    int src_cur = load_src(k), src_ahead = -1;
    bool any = false;
    if (k < K) {
        any = wave_uses(k, src_cur);
        dma_w(k, 0, 0);
        fetch_a(ga, 0, src_cur, any);
        src_ahead = load_src(next_k(k));
    }
    for (; k < K;) {
        int kn = k, hn = h + 1;                           // the stage after this one
        if (hn == SL) { kn = next_k(k); hn = 0; }
        __syncthreads();                                  // this stage's W has landed for every wave; the other buffer's readers are done
        int src_next = src_cur;
        bool any_next = false;
        if (kn < K) {                                     // in flight while the MFMAs below run
            if (kn != k) {
                src_next = src_ahead;
                src_ahead = load_src(next_k(kn));
            }
            any_next = wave_uses(kn, src_next);
            dma_w(kn, hn, buf ^ 1);
            fetch_a(gn, hn, src_next, any_next);
        }
        if (any) {
            const float4 *Wb = s_w4 + (size_t)buf * NI * 64 + l;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float av[4] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w};
                float4 bq[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) bq[q] = Wb[(u * NT + q) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const float bv4[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv4[j], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        if (kn < K) {
#pragma unroll
            for (int u = 0; u < NG; ++u) ga[u] = gn[u];
        }
        src_cur = src_next;
        any = any_next;
        k = kn;
        h = hn;
        buf ^= 1;
    }
    const int last = n_out - 1;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        int orow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = hh * 8 + j;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last);
            orow[j] = out_row ? out_row[row] : row;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = q * 32 + (l & 31);
            const int cc = min(col, Cout - 1);
            const float bv = bias ? bias[cc] : 0.f;
            float res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] = 0.f;
            if (residual) {
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = residual[(size_t)orow[j] * Cout + cc];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = hh * 8 + j;
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out && col < Cout) {
                    const float v = acc[q][r] + bv + res[j];
                    out[(size_t)orow[j] * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// W (K, CinT, Cout) -> the packed order above; one thread per float4
__global__ __launch_bounds__(256) void sc_pack_weights_kernel(const float *__restrict__ W, int K, int CinT, int Cout, int Cin, int NT,
                                                              float4 *__restrict__ P, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int HALF = Cin / 2, NG = HALF / 4, SL = CinT / Cin;
    long long r = e;
    const int lane = (int)(r & 63); r >>= 6;
    const int q = (int)(r % NT); r /= NT;
    const int u = (int)(r % NG); r /= NG;
    const int h = (int)(r % SL);
    const int k = (int)(r / SL);
    const int ar = lane & 31, ak = lane >> 5, col = q * 32 + ar;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < Cout) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = W[((size_t)k * CinT + h * Cin + (2 * u + ak) * 4 + j) * Cout + col];
    }
    P[e] = make_float4(v[0], v[1], v[2], v[3]);
}

// Wave-private variant of the register-A kernel (r03 experiment, LIDAR_SPCONV_WAVE_KERNEL=1; needs 4 waves x Cin x CW floats <= 64 KB
// of LDS, i.e. two workgroups per CU).  MEASURED SLOWER than the register-A kernel (64->64: 418-424 vs 382-384 us, SECOND stack
// 2.18 vs 2.01 ms): the per-offset barrier is NOT what holds the matrix pipe at 49 %.  The register-A kernel shares each W[k] stage among the four waves of a workgroup: one barrier per
// offset, and a wave whose 32 rows skip an offset still waits for the waves that use it — the matrix pipe was busy 49 % of the
// time.  Here every wave stages ITS OWN copy of W[k] (single-buffered in LDS, the next stage prefetched into registers while the
// MFMAs of the current one run), walks only the offsets its own rows use, and never meets a barrier: a wave's LDS operations
// execute in order, so the stores of stage s + 1 follow the last read of stage s by themselves.  The price is 4 x the W
// traffic (L2 hits: a layer's 27 slices are 0.4 MB) and 64 more registers at Cin = Cout = 64 (2 waves per SIMD instead of 3).
// Same channel ownership, same summation order: results are bit-identical to the register-A kernel's.
template <int NT, int C4, int SL = 1>
__global__ __launch_bounds__(256, 2) void sc_implicit_gemm_wave_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                       int K, int Cout, const float *__restrict__ Wt,
                                                                       const float *__restrict__ bias, const float *__restrict__ residual,
                                                                       int relu, float *__restrict__ out,
                                                                       const int *__restrict__ row_mask, const int *__restrict__ out_row) {
    constexpr int Cin = C4 * 4, HALF = Cin / 2, CW = NT * 32, CW4 = CW / 4, CinT = Cin * SL;
    constexpr int NG = HALF / 4;                          // float4 gathers per lane per stage
    constexpr int NW = (Cin * CW4 + 63) / 64;             // float4 weight pieces per lane per stage
    extern __shared__ float s_mem[];                      // [4 waves][Cin][CW]: each wave's own W stage
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    float *Wl = s_mem + (size_t)wv * Cin * CW;
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    const int ar = l & 31, ak = l >> 5;
    const int myrow = row0 + ar;
    const int trow = (out_row && myrow < n_out) ? out_row[myrow] : myrow;
    const int Co4 = Cout >> 2;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    unsigned wave_mask = 0xffffffffu;
    if (row_mask) {                                       // mask-sorted tables: the offsets some row of THIS wave uses
        unsigned m = (myrow < n_out) ? (unsigned)row_mask[trow] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
        wave_mask = m;
    }
    auto next_k = [&](int k) {                            // next offset > k some row of the wave uses (K when none)
        if (!row_mask) return k + 1;
        const unsigned rest = (k + 1 < 32) ? (wave_mask >> (k + 1)) : 0u;
        return rest ? k + 1 + __builtin_ctz(rest) : K;
    };
    auto load_src = [&](int k) { return (myrow < n_out && k < K) ? nbr[(size_t)trow * K + k] : -1; };
    float4 wr[NW], ga[NG], gn[NG];
    auto fetch_w = [&](int k, int h) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 64 + l, ci = e / CW4, q4 = e - ci * CW4;
            wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Cin * CW4 && q4 < Co4) wr[q] = reinterpret_cast<const float4 *>(Wt + ((size_t)k * CinT + h * Cin + ci) * Cout)[q4];
        }
    };
    auto store_w = [&]() {
        float4 *dst = reinterpret_cast<float4 *>(Wl);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int e = q * 64 + l;
            if (e < Cin * CW4) dst[e] = wr[q];
        }
    };
    auto fetch_a = [&](float4 (&g)[NG], int h, int src, bool any) {
        const float4 *rowp = reinterpret_cast<const float4 *>(in + (size_t)max(src, 0) * CinT + h * Cin) + ak;    // pieces ak, ak + 2, ...: see the ownership note
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any && src >= 0) g[u] = rowp[2 * u];
        }
    };
    int k = next_k(-1), h = 0;
    if (k >= K) k = K;
    int src_cur = load_src(k), src_ahead = -1;
    bool any = false;
    if (k < K) {
        any = __ballot(src_cur >= 0) != 0ull;
        fetch_w(k, 0);
        fetch_a(ga, 0, src_cur, any);
        src_ahead = load_src(next_k(k));
        store_w();
    }
    for (; k < K;) {
        int kn = k, hn = h + 1;                           // the stage after this one
        if (hn == SL) { kn = next_k(k); hn = 0; }
        int src_next = src_cur;
        bool any_next = false;
        if (kn < K) {                                     // in flight while the MFMAs below run
            if (kn != k) {
                src_next = src_ahead;
                src_ahead = load_src(next_k(kn));
            }
            any_next = __ballot(src_next >= 0) != 0ull;
            fetch_w(kn, hn);
            fetch_a(gn, hn, src_next, any_next);
        }
        if (any) {
            const float *Wb = Wl + (size_t)ak * 4 * CW + ar;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float av[4] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], Wb[(u * 8 + j) * CW + q * 32], acc[q], 0, 0, 0);
                }
            }
        }
        if (kn < K) {
            store_w();                                    // (after this stage's last LDS read, in the wave's own order)
#pragma unroll
            for (int u = 0; u < NG; ++u) ga[u] = gn[u];
        }
        src_cur = src_next;
        any = any_next;
        k = kn;
        h = hn;
    }
    const int last = n_out - 1;                           // n_out >= 1 (checked by the launcher)
    // fused epilogue: (+ bias) (+ residual) (ReLU); sorted tables scatter rows (as in the pipe kernel)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        int orow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = hh * 8 + j;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last);
            orow[j] = out_row ? out_row[row] : row;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = q * 32 + (l & 31);
            const int cc = min(col, Cout - 1);
            const float bv = bias ? bias[cc] : 0.f;
            float res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] = 0.f;
            if (residual) {
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = residual[(size_t)orow[j] * Cout + cc];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = hh * 8 + j;
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out && col < Cout) {
                    const float v = acc[q][r] + bv + res[j];
                    out[(size_t)orow[j] * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// Input layer (Cin == 4: x, y, z, intensity means; K * 4 <= 128): all K offsets in ONE stage.  The per-offset kernels above
// spend two barriers per offset on 2 MFMA steps of work here; this one gathers the wave's 32 x (K * 4) tile and the whole
// (K * 4, Cout) weight once, then runs the K * 2 MFMA steps back to back.  Same operand pairs in the same order as the
// per-offset kernels (skipped offsets contribute exact zeros), so the sums are the same.
template <int NT>
__global__ __launch_bounds__(256) void sc_input_layer_gemm_kernel(const float *__restrict__ in, const int *__restrict__ nbr, int n_out,
                                                                  int K, int Cout, const float *__restrict__ Wt,
                                                                  const float *__restrict__ bias, const float *__restrict__ residual,
                                                                  int relu, float *__restrict__ out) {
    constexpr int CW = NT * 32, MAXG = 14;                // MAXG * 64 >= 32 rows * 27 offsets
    const int KC = K * 4, KCp = KC + 1;
    extern __shared__ float s_mem[];
    float *s_w = s_mem;                                   // [KC][CW]
    float *s_a = s_mem + (size_t)KC * CW;                 // [4 waves][32][KCp]
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int row0 = blockIdx.x * IG_ROWS + wv * 32;
    const int last = n_out - 1;
    float *A = s_a + (size_t)wv * 32 * KCp;
    // the wave's 32 x K table entries are contiguous: coalesced, all requested before the first gather
    const int npairs = 32 * K;
    int src[MAXG];
    const long long tbase = (long long)row0 * K, tend = (long long)n_out * K;
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
        const long long e = tbase + u * 64 + l;
        src[u] = nbr[e < tend ? e : tend - 1];
    }
    float4 g[MAXG];
#pragma unroll
    for (int u = 0; u < MAXG; ++u) g[u] = reinterpret_cast<const float4 *>(in)[max(src[u], 0)];
    for (int e = t; e < KC * CW; e += 256) {
        const int ci = e / CW, co = e - ci * CW;
        s_w[e] = co < Cout ? Wt[(size_t)ci * Cout + co] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
        const int p = u * 64 + l;
        if (p < npairs) {
            const int r = p / K, k = p - r * K;
            const bool ok = src[u] >= 0 && tbase + p < tend;
            float *dst = A + r * KCp + k * 4;
            dst[0] = ok ? g[u].x : 0.f; dst[1] = ok ? g[u].y : 0.f; dst[2] = ok ? g[u].z : 0.f; dst[3] = ok ? g[u].w : 0.f;
        }
    }
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int ar = l & 31, ak = l >> 5;
#pragma unroll 4
    for (int c0 = 0; c0 < KC; c0 += 2) {
        const float a = A[ar * KCp + c0 + ak];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const float b = s_w[(c0 + ak) * CW + q * 32 + ar];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int col = q * 32 + (l & 31);
            const int cc = min(col, Cout - 1);
            const float bv = bias ? bias[cc] : 0.f;
            float res[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) res[j] = 0.f;
            if (residual) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = h * 8 + j;
                    res[j] = residual[(size_t)min(row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), last) * Cout + cc];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = h * 8 + j;
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row < n_out && col < Cout) {
                    const float v = acc[q][r] + bv + res[j];
                    out[(size_t)row * Cout + col] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// indice_conv forward / dgrad (with the transposed table and W^T) — spconv.functional indice_conv (Appendix A.3) —
// with the inference epilogue of the reference's conv/BatchNorm1d/ReLU triplets (spconv_backbone.py:20-26) and of
// SparseBasicBlock (spconv_backbone.py:49-63): out = act(gemm + bias + residual).
static int sc_pk_stage_cin(int Cin) { return Cin == 128 ? 64 : Cin; }
static bool sc_pk_supported(int K, int Cin, int Cout) {
    return K > 0 && K <= 32 && (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128) && Cout > 0 && Cout <= IG_MAX_C && (Cout & 3) == 0;
}

static int sc_gemm_launch(const float *in_features, const int *nbr, int n_out, int K, int Cin, int Cout, const float *weight,
                          const float *bias, const float *residual, int relu, float *out_features, const int *row_mask,
                          const int *out_row, void *stream, const float *packed = nullptr) {
    if (n_out < 0 || K <= 0 || Cin <= 0 || Cout <= 0 || Cin > IG_MAX_C || Cout > IG_MAX_C) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!in_features || !nbr || !weight || !out_features) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = divup(Cout, 32);
    const int cin_stage = (Cin == 128 && (Cout & 3) == 0) ? 64 : Cin;     // 128 input channels: two 64-channel stages per offset
    const size_t lds = ((size_t)cin_stage * nt * 32 + (size_t)4 * 32 * (cin_stage + 1)) * sizeof(float);
    const dim3 grid(divup(n_out, IG_ROWS));
    static const bool use_pipe = getenv("LIDAR_SPCONV_PIPE_KERNEL") != nullptr;       // A/B switch: the r02 LDS-transposing kernel
    const size_t lds_rega = (size_t)2 * cin_stage * nt * 32 * sizeof(float);          // two W stage buffers
    static const bool want_wave = getenv("LIDAR_SPCONV_WAVE_KERNEL") != nullptr;      // A/B switch: the barrier-free wave-private kernel
    const size_t lds_wave = (size_t)4 * cin_stage * nt * 32 * sizeof(float);          // one W stage per wave
    const bool use_wave = want_wave && lds_wave <= 65536;                             // (two workgroups per CU)
#define IGP(NT, C4) do { if (use_pipe) hipLaunchKernelGGL((sc_implicit_gemm_pipe_kernel<NT, C4>), grid, dim3(256), lds, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row); \
                         else if (use_wave && NT <= 2) hipLaunchKernelGGL((sc_implicit_gemm_wave_kernel<(NT <= 2 ? NT : 1), C4>), grid, dim3(256), lds_wave, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row); \
                         else hipLaunchKernelGGL((sc_implicit_gemm_rega_kernel<NT, C4>), grid, dim3(256), lds_rega, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row); } while (0)
#define IGP2(NT, C4) do { if (use_pipe) hipLaunchKernelGGL((sc_implicit_gemm_pipe_kernel<NT, C4, 2>), grid, dim3(256), lds, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row); \
                          else hipLaunchKernelGGL((sc_implicit_gemm_rega_kernel<NT, C4, 2>), grid, dim3(256), lds_rega, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row); } while (0)
    const bool pipe = (Cout & 3) == 0 && (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128);
    if ((row_mask || out_row) && (!pipe || K > 32 || !row_mask || !out_row)) return LIDAR_ERR_ARG;
    if (packed && !sc_pk_supported(K, Cin, Cout)) return LIDAR_ERR_ARG;
    if (packed) {                                         // packed weights: LDS-DMA staging + 128-bit operand reads
        const size_t lds_pk = (size_t)2 * cin_stage * nt * 32 * sizeof(float);
#define IGK(NT, C4, SLV) hipLaunchKernelGGL((sc_implicit_gemm_pk_kernel<NT, C4, SLV>), grid, dim3(256), lds_pk, s, in_features, nbr, n_out, K, Cout, packed, bias, residual, relu, out_features, row_mask, out_row)
#define IGK_NT(C4, SLV) switch (nt) { case 1: IGK(1, C4, SLV); break; case 2: IGK(2, C4, SLV); break; case 3: IGK(3, C4, SLV); break; default: IGK(4, C4, SLV); break; }
        if (Cin == 16) { IGK_NT(4, 1) } else if (Cin == 32) { IGK_NT(8, 1) } else if (Cin == 64) { IGK_NT(16, 1) } else { IGK_NT(16, 2) }
#undef IGK_NT
#undef IGK
        return lidar_check_launch("lidar_spconv_implicit_gemm(packed)");
    }
    // A/B switch, default OFF: LIDAR_SPCONV_RT2=1 selects the two-row-tile kernel for Cin 32 / 64.  Measured (profiles/r04/
    // spconv_gemm_rt2.log, SECOND stack): 2.89 vs 1.99 ms — at 272 registers one workgroup per CU is left, and this kernel family hides its
    // stage barrier and gather latency behind the OTHER resident workgroup, not behind a software pipeline of its own.
    static const bool rt2_on = getenv("LIDAR_SPCONV_RT2") && atoi(getenv("LIDAR_SPCONV_RT2")) != 0;
    const bool rt2_off = !rt2_on;
    if (pipe && !rt2_off && !use_pipe && !use_wave && (Cin == 64 || Cin == 32) && nt <= 2 && n_out >= 8 * 2 * IG_ROWS) {
        const dim3 grid2(divup(n_out, 2 * IG_ROWS));
#define IGR2(NT, C4) hipLaunchKernelGGL((sc_implicit_gemm_rega2_kernel<NT, C4>), grid2, dim3(256), lds_rega, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features, row_mask, out_row)
        if (Cin == 64) { if (nt == 1) IGR2(1, 16); else IGR2(2, 16); }
        else { if (nt == 1) IGR2(1, 8); else IGR2(2, 8); }
#undef IGR2
        return lidar_check_launch("lidar_spconv_implicit_gemm(two row tiles)");
    }
    if (pipe) {
        const int c4 = Cin / 4;
#define IGP_NT(C4) switch (nt) { case 1: IGP(1, C4); break; case 2: IGP(2, C4); break; case 3: IGP(3, C4); break; default: IGP(4, C4); break; }
        if (c4 == 4) { IGP_NT(4) } else if (c4 == 8) { IGP_NT(8) } else if (c4 == 16) { IGP_NT(16) } else {
            switch (nt) { case 1: IGP2(1, 16); break; case 2: IGP2(2, 16); break; case 3: IGP2(3, 16); break; default: IGP2(4, 16); break; }
        }
#undef IGP_NT
#undef IGP2
        return lidar_check_launch("lidar_spconv_implicit_gemm(pipe)");
    }
#undef IGP
    if (Cin == 4 && K * 32 <= 14 * 64 && nt <= 2) {       // the network's input layer: one stage for all offsets
        const size_t lds_in = ((size_t)K * 4 * nt * 32 + (size_t)4 * 32 * (K * 4 + 1)) * sizeof(float);
        if (nt == 1) hipLaunchKernelGGL(sc_input_layer_gemm_kernel<1>, grid, dim3(256), lds_in, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features);
        else hipLaunchKernelGGL(sc_input_layer_gemm_kernel<2>, grid, dim3(256), lds_in, s, in_features, nbr, n_out, K, Cout, weight, bias, residual, relu, out_features);
        return lidar_check_launch("lidar_spconv_implicit_gemm(input layer)");
    }
#define IG_CASE(NT) hipLaunchKernelGGL(sc_implicit_gemm_kernel<NT>, grid, dim3(256), lds, s, in_features, nbr, n_out, K, Cin, Cout, weight, bias, residual, relu, out_features)
    switch (nt) {
        case 1: IG_CASE(1); break;
        case 2: IG_CASE(2); break;
        case 3: IG_CASE(3); break;
        default: IG_CASE(4); break;
    }
#undef IG_CASE
    return lidar_check_launch("lidar_spconv_implicit_gemm");
}

LIDAR_EXPORT int lidar_spconv_implicit_gemm_fused(const float *in_features, const int *nbr, int n_out, int K, int Cin, int Cout,
                                                  const float *weight, const float *bias, const float *residual, int relu,
                                                  float *out_features, void *stream) {
    return sc_gemm_launch(in_features, nbr, n_out, K, Cin, Cout, weight, bias, residual, relu, out_features, nullptr, nullptr,
                          stream);
}

// Per-row neighbour-offset bit masks of an (n_out, K <= 32) table: bit k set when nbr[row][k] >= 0.
__global__ __launch_bounds__(256) void sc_row_masks_kernel(const int *__restrict__ nbr, int n_out, int K, int *__restrict__ masks) {
    // a block owns 256 consecutive rows = one contiguous run of 256*K table entries, read coalesced; bits meet in LDS
    __shared__ unsigned s_mask[256];
    const int t = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 256;
    const int rows = (int)min((long long)256, (long long)n_out - row0);
    s_mask[t] = 0u;
    __syncthreads();
    const int *base = nbr + row0 * K;
    const int total = rows * K;
    for (int i = t; i < total; i += 256) {
        const int r = i / K, k = i - r * K;
        if (base[i] >= 0) atomicOr(&s_mask[r], 1u << k);
    }
    __syncthreads();
    if (t < rows) masks[row0 + t] = (int)s_mask[t];
}

LIDAR_EXPORT int lidar_spconv_row_masks(const int *nbr, int n_out, int K, int *masks, void *stream) {
    if (n_out < 0 || K <= 0 || K > 32) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!nbr || !masks) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(sc_row_masks_kernel, dim3(divup(n_out, 256)), dim3(256), 0, (hipStream_t)stream, nbr, n_out, K, masks);
    return lidar_check_launch("lidar_spconv_row_masks");
}

// ------------------------------------------------------------------ mask order without a sort
// The mask-ordered GEMM needs the rows of a table grouped by their K-bit neighbour-offset mask, similar masks near each other.
// Measured on the SECOND-KITTI tables (tools/mask_stats.py, tools/group_order_probe.py): 150-430 k rows hold only 8-25 k DISTINCT
// masks; equal masks contiguous but groups in random order costs +26 % GEMM time against a full sort, while groups ordered by
// their top 12 mask bits only (any order inside such a bin) is as fast as the full sort (2.09 vs 2.08 ms for the stack).
// So no sort (rocPRIM onesweep took 6 launches and 130-150 us per table):
//   1. every row finds / claims the slot of its mask in a plain open-addressing hash table and takes a rank inside that group
//      (one wave-aggregated atomic per distinct mask of a wave); the group's 12-bit bin is counted on the side;
//   2. every group adds its size to its bin's cursor (one atomic per GROUP) -> first position = scanned bin offset + that;
//   3. position of a row = first position of its group + its rank: scatter.
// 3 launches, no host sync.  Any grouping gives bit-identical GEMM results (each output row is summed by its own lane in fixed
// offset order); the order inside a bin / a group depends on atomic timing and does not matter.
#define MG_BINS 4096
#define MG_BIN_BITS 12
#define MG_EMPTY 0xFFFFFFFFu

struct MgWs {
    unsigned *keys;   // [capmax] distinct masks (MG_EMPTY = free); persistent: empty between calls
    int *cnt;         // [capmax] rows per group; zero between calls
    int *off;         // [capmax] first position of the group
    int *bincnt;      // [MG_BINS] rows per 12-bit bin; zero between calls
    int *bincur;      // [MG_BINS] running cursor inside the bin; zero between calls
    int *rslot;       // [capmax / 2] slot of the row
    int *rrank;       // [capmax / 2] rank of the row inside its group
    size_t capmax;
};

static size_t mg_bytes(size_t capmax) { return 256 + 12 * capmax + 2 * MG_BINS * 4 + 8 * (capmax / 2) + 2048; }

static size_t mg_cap_of(int n) {
    size_t c = 1 << 16;
    while (c < 2 * (size_t)(n > 0 ? n : 1)) c <<= 1;
    return c;
}

// the layout is a function of the buffer size alone (the same for init and for every call on that buffer)
static bool mg_carve(void *base, size_t ws_bytes, MgWs *w) {
    size_t cap = 1 << 16;
    if (mg_bytes(cap) > ws_bytes) return false;
    while (mg_bytes(cap * 2) <= ws_bytes) cap <<= 1;
    char *p = (char *)base + 256;
    w->capmax = cap;
    w->keys = (unsigned *)p; p += 4 * cap;
    w->cnt = (int *)p; p += 4 * cap;
    w->off = (int *)p; p += 4 * cap;
    w->bincnt = (int *)p; p += 4 * MG_BINS;
    w->bincur = (int *)p; p += 4 * MG_BINS;
    p = (char *)base + align_up((size_t)(p - (char *)base), 256);
    w->rslot = (int *)p; p += 4 * (cap / 2);
    w->rrank = (int *)p;
    return true;
}

__global__ void mg_init_kernel(MgWs w) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < w.capmax; i += (size_t)gridDim.x * blockDim.x) {
        w.keys[i] = MG_EMPTY;
        w.cnt[i] = 0;
        if (i < MG_BINS) w.bincnt[i] = w.bincur[i] = 0;
    }
}

// group order along the GEMM's launch order: HEAVY masks first (a workgroup's time is proportional to the offsets its rows use, and
// the hardware deals workgroups in launch order: with the heavy ones last, the launch ends on a long tail of half-empty CUs — measured
// on the SECOND stack: offsets per workgroup 8 -> 19 along the r02 order (ascending top 12 mask bits), 19 -> 5 along this one),
// similar masks next to each other inside a weight class: bin = (K - popcount) : 5 bits | top 7 mask bits.  Measured (SECOND stack,
// register-A kernel): r02 order 2.141 ms, r02 order dealt back to front 2.041, this 2.025, 3 weight bits + 9 mask bits 2.084.
__device__ __forceinline__ int mg_bin(unsigned m, int K) {
    const unsigned top = K > 7 ? (m >> (K - 7)) : m;
    return (int)((((unsigned)K - (unsigned)__popc(m)) & 31u) << 7 | (top & 127u));
}

// masks of 1024 consecutive rows (as sc_row_masks_kernel), then every row finds / claims the group of its mask and a rank in it.
// Global atomics on a group / bin counter serialise at the L2 (~10 ns each on one address), and 40 % of a table's rows can carry
// ONE mask: the rows of a workgroup therefore meet in an LDS table first (LDS atomics), and the workgroup touches each global
// counter once per distinct mask it holds.
#define MG_ROWS 1024
#define MG_LSLOTS 2048
__global__ __launch_bounds__(MG_ROWS) void mg_insert_kernel(const int *__restrict__ nbr, int n_out, int K, int *__restrict__ masks, MgWs w,
                                                           int cap_shift) {
    __shared__ unsigned s_mask[MG_ROWS];
    __shared__ unsigned s_key[MG_LSLOTS];
    __shared__ int s_cnt[MG_LSLOTS];
    __shared__ int s_first[MG_LSLOTS];
    __shared__ int s_gslot[MG_LSLOTS];
    __shared__ int s_bin[MG_BINS];                                  // rows of this workgroup per 12-bit bin
    const int t = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * MG_ROWS;
    const int rows = (int)min((long long)MG_ROWS, (long long)n_out - row0);
    s_mask[t] = 0u;
    s_key[t] = s_key[t + MG_ROWS] = MG_EMPTY;
    s_cnt[t] = s_cnt[t + MG_ROWS] = 0;
    for (int q = t; q < MG_BINS; q += MG_ROWS) s_bin[q] = 0;
    __syncthreads();
    const int *base = nbr + row0 * K;
    const int total = rows * K;
    for (int i = t; i < total; i += MG_ROWS) {
        const int r = i / K, k = i - r * K;
        if (base[i] >= 0) atomicOr(&s_mask[r], 1u << k);
    }
    __syncthreads();
    const bool valid = t < rows;
    const unsigned m = s_mask[t];
    int ls = 0, lrank = 0;
    if (valid) {
        masks[row0 + t] = (int)m;
        ls = (int)((m * 2654435761u) >> (32 - 11));
        for (;;) {                                                  // 1024 rows into 2048 slots: always terminates
            const unsigned old = atomicCAS(&s_key[ls], MG_EMPTY, m);
            if (old == MG_EMPTY || old == m) break;
            ls = (ls + 1) & (MG_LSLOTS - 1);
        }
        lrank = atomicAdd(&s_cnt[ls], 1);
    }
    __syncthreads();
    const unsigned capm = (1u << (32 - cap_shift)) - 1u;
    for (int q = t; q < MG_LSLOTS; q += MG_ROWS) {                  // one global visit per distinct mask of the workgroup
        const unsigned key = s_key[q];
        if (key == MG_EMPTY) continue;
        unsigned slot = (key * 2654435761u) >> cap_shift;
        for (unsigned probe = 0; probe <= capm; ++probe) {
            unsigned old = __hip_atomic_load(&w.keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == MG_EMPTY) old = atomicCAS(&w.keys[slot], MG_EMPTY, key);
            if (old == MG_EMPTY || old == key) break;
            slot = (slot + 1u) & capm;
        }
        const int c = s_cnt[q];
        s_gslot[q] = (int)slot;
        s_first[q] = atomicAdd(&w.cnt[slot], c);
        atomicAdd(&s_bin[mg_bin(key, K)], c);
    }
    __syncthreads();
    for (int q = t; q < MG_BINS; q += MG_ROWS)
        if (s_bin[q]) atomicAdd(&w.bincnt[q], s_bin[q]);            // once per bin this workgroup touches
    if (valid) {
        w.rslot[row0 + t] = s_gslot[ls];
        w.rrank[row0 + t] = s_first[ls] + lrank;
    }
}

// one thread per table slot: a group's first position = offset of its 12-bit bin (every workgroup scans the 4096 bin counts for
// itself) + the bin's running cursor; the slot is left empty for the next call
__global__ __launch_bounds__(1024) void mg_assign_kernel(MgWs w, int K, size_t cap) {
    __shared__ int s_bin[MG_BINS];      // scanned bin offsets
    __shared__ int s_cur[MG_BINS];      // rows of this workgroup's groups per bin -> the workgroup's base inside the bin
    __shared__ int s_w[16];
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int4 c = reinterpret_cast<const int4 *>(w.bincnt)[t];
    const int mine = c.x + c.y + c.z + c.w;
    const int inc = wave_incl_scan(mine);
    if (l == 63) s_w[wv] = inc;
    for (int q = t; q < MG_BINS; q += 1024) s_cur[q] = 0;
    __syncthreads();
    int o = inc - mine;
    for (int k = 0; k < wv; ++k) o += s_w[k];
    s_bin[4 * t] = o; s_bin[4 * t + 1] = o + c.x; s_bin[4 * t + 2] = o + c.x + c.y; s_bin[4 * t + 3] = o + c.x + c.y + c.z;
    const size_t slot = (size_t)blockIdx.x * 1024 + t;
    const int n = slot < cap ? w.cnt[slot] : 0;
    int b = 0, loc = 0;
    if (n > 0) {
        b = mg_bin(w.keys[slot], K);
        loc = atomicAdd(&s_cur[b], n);                              // LDS: place among this workgroup's groups of the bin
    }
    __syncthreads();
    for (int q = t; q < MG_BINS; q += 1024) {                       // one global atomic per bin this workgroup touches
        const int v = s_cur[q];
        if (v) s_cur[q] = atomicAdd(&w.bincur[q], v);
    }
    __syncthreads();
    if (n > 0) {
        w.off[slot] = s_bin[b] + s_cur[b] + loc;
        w.keys[slot] = MG_EMPTY;
        w.cnt[slot] = 0;
    }
}

__global__ void mg_scatter_kernel(int n_out, MgWs w, int *__restrict__ order) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < MG_BINS) w.bincnt[r] = w.bincur[r] = 0;                 // (nobody reads them in this launch)
    if (r < n_out) order[w.off[w.rslot[r]] + w.rrank[r]] = r;
}

LIDAR_EXPORT size_t lidar_spconv_mask_group_workspace_bytes(int n_out) { return mg_bytes(mg_cap_of(n_out)); }

// once per workspace (and after a failed call): the table must be empty when lidar_spconv_mask_group starts; it leaves it empty
LIDAR_EXPORT int lidar_spconv_mask_group_init(void *ws, size_t ws_bytes, void *stream) {
    MgWs w;
    if (!ws || !mg_carve(ws, ws_bytes, &w)) return LIDAR_ERR_WORKSPACE;
    hipLaunchKernelGGL(mg_init_kernel, dim3(512), dim3(1024), 0, (hipStream_t)stream, w);
    return lidar_check_launch("lidar_spconv_mask_group_init");
}

// nbr (n_out, K <= 31) -> masks (n_out): bit k = nbr[row][k] >= 0; order (n_out): order[i] = table row visited i-th, rows with
// equal masks contiguous, groups ordered by their top 12 mask bits
LIDAR_EXPORT int lidar_spconv_mask_group(const int *nbr, int n_out, int K, int *masks, int *order, void *ws, size_t ws_bytes,
                                         void *stream) {
    if (n_out < 0 || K <= 0 || K > 31) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!nbr || !masks || !order || !ws) return LIDAR_ERR_ARG;
    MgWs w;
    const size_t cap = mg_cap_of(n_out);
    if (!mg_carve(ws, ws_bytes, &w) || cap > w.capmax) return LIDAR_ERR_WORKSPACE;
    int lg = 0;
    while (((size_t)1 << lg) < cap) ++lg;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mg_insert_kernel, dim3(divup(n_out, MG_ROWS)), dim3(MG_ROWS), 0, s, nbr, n_out, K, masks, w, 32 - lg);
    hipLaunchKernelGGL(mg_assign_kernel, dim3(divup((long long)cap, 1024)), dim3(1024), 0, s, w, K, cap);
    hipLaunchKernelGGL(mg_scatter_kernel, dim3(divup(n_out > MG_BINS ? n_out : MG_BINS, 256)), dim3(256), 0, s, n_out, w, order);
    return lidar_check_launch("lidar_spconv_mask_group");
}

// 1 when lidar_spconv_implicit_gemm_sorted can run this shape
LIDAR_EXPORT int lidar_spconv_sorted_gemm_supported(int K, int Cin, int Cout) {
    return (K > 0 && K <= 32 && (Cout & 3) == 0 && Cout > 0 && Cout <= IG_MAX_C &&
            (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128)) ? 1 : 0;
}

// The fused GEMM with its rows visited in mask order: workgroup row i computes table / output row order[i] (row_mask is in
// table order).  Same sums in the same order as the table-order call: bit-identical results.
LIDAR_EXPORT int lidar_spconv_implicit_gemm_sorted(const float *in_features, const int *nbr, const int *row_mask,
                                                   const int *out_row, int n_out, int K, int Cin, int Cout,
                                                   const float *weight, const float *bias, const float *residual, int relu,
                                                   float *out_features, void *stream) {
    if (!lidar_spconv_sorted_gemm_supported(K, Cin, Cout) || (n_out > 0 && (!row_mask || !out_row))) return LIDAR_ERR_ARG;
    return sc_gemm_launch(in_features, nbr, n_out, K, Cin, Cout, weight, bias, residual, relu, out_features, row_mask,
                          out_row, stream);
}

// Packed weights for the mask-ordered GEMM (sc_implicit_gemm_pk_kernel): floats needed for (K, Cin, Cout), 0 = shape not supported
LIDAR_EXPORT size_t lidar_spconv_packed_floats(int K, int Cin, int Cout) {
    if (!sc_pk_supported(K, Cin, Cout)) return 0;
    return (size_t)K * Cin * divup(Cout, 32) * 32;
}

// weight (K, Cin, Cout) row-major (BatchNorm scale folded in) -> packed; once per weight update
LIDAR_EXPORT int lidar_spconv_pack_weights(const float *weight, int K, int Cin, int Cout, float *packed, void *stream) {
    if (!weight || !packed || !sc_pk_supported(K, Cin, Cout) || (reinterpret_cast<uintptr_t>(packed) & 15)) return LIDAR_ERR_ARG;
    const long long total = (long long)lidar_spconv_packed_floats(K, Cin, Cout) / 4;
    hipLaunchKernelGGL(sc_pack_weights_kernel, dim3(divup(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, K, Cin, Cout,
                       sc_pk_stage_cin(Cin), divup(Cout, 32), reinterpret_cast<float4 *>(packed), total);
    return lidar_check_launch("lidar_spconv_pack_weights");
}

// lidar_spconv_implicit_gemm_sorted with the weights in packed form (same results, bit for bit)
LIDAR_EXPORT int lidar_spconv_implicit_gemm_sorted_packed(const float *in_features, const int *nbr, const int *row_mask,
                                                          const int *out_row, int n_out, int K, int Cin, int Cout,
                                                          const float *packed, const float *bias, const float *residual, int relu,
                                                          float *out_features, void *stream) {
    if (!lidar_spconv_sorted_gemm_supported(K, Cin, Cout) || !sc_pk_supported(K, Cin, Cout) || (n_out > 0 && (!row_mask || !out_row)) ||
        !packed || (reinterpret_cast<uintptr_t>(packed) & 15))
        return LIDAR_ERR_ARG;
    return sc_gemm_launch(in_features, nbr, n_out, K, Cin, Cout, packed, bias, residual, relu, out_features, row_mask, out_row, stream,
                          packed);
}

LIDAR_EXPORT int lidar_spconv_implicit_gemm(const float *in_features, const int *nbr, int n_out, int K, int Cin, int Cout,
                                            const float *weight, const float *bias, float *out_features, void *stream) {
    return lidar_spconv_implicit_gemm_fused(in_features, nbr, n_out, K, Cin, Cout, weight, bias, nullptr, 0, out_features,
                                            stream);
}

// ------------------------------------------------------------------ weight gradient
// dW[k] (Cin, Cout) = sum_j in[nbr[j][k]]^T (x) dout[j].  grid = (row chunks, K); each workgroup reduces its
// chunk in registers (thread = one (ci, 4-wide co strip) cell set) and adds it to dW with float atomics.
#define WG_CHUNK 512
template <int PER>
__global__ __launch_bounds__(256) void sc_wgrad_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                       const int *__restrict__ nbr, int n_out, int K, int Cin, int Cout,
                                                       float *__restrict__ dW) {
    extern __shared__ float s_mem[];
    float *s_in = s_mem;                    // [16][Cin]
    float *s_do = s_mem + 16 * Cin;         // [16][Cout]
    __shared__ int s_src[16];
    const int k = blockIdx.y, t = threadIdx.x;
    const int row_begin = blockIdx.x * WG_CHUNK, row_end = min(row_begin + WG_CHUNK, n_out);
    const int cells = Cin * Cout;
    float acc[PER];                         // PER * 256 >= Cin * Cout
#pragma unroll
    for (int q = 0; q < PER; ++q) acc[q] = 0.f;
    for (int r0 = row_begin; r0 < row_end; r0 += 16) {
        __syncthreads();
        if (t < 16) s_src[t] = (r0 + t < row_end) ? nbr[(size_t)(r0 + t) * K + k] : -1;
        __syncthreads();
        for (int e = t; e < 16 * Cin; e += 256) {
            const int r = e / Cin, c = e - r * Cin;
            s_in[e] = s_src[r] >= 0 ? in[(size_t)s_src[r] * Cin + c] : 0.f;
        }
        for (int e = t; e < 16 * Cout; e += 256) {
            const int r = e / Cout, c = e - r * Cout;
            s_do[e] = (s_src[r] >= 0) ? dout[(size_t)(r0 + r) * Cout + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int cell = q * 256 + t;
            if (cell < cells) {
                const int ci = cell / Cout, co = cell - ci * Cout;
                float a = acc[q];
#pragma unroll
                for (int r = 0; r < 16; ++r) a = fmaf(s_in[r * Cin + ci], s_do[r * Cout + co], a);
                acc[q] = a;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int cell = q * 256 + t;
        if (cell < cells && acc[q] != 0.f) atomicAdd(&dW[(size_t)k * cells + cell], acc[q]);
    }
}

// dW (K, Cin, Cout) must be zero-filled by the caller
LIDAR_EXPORT int lidar_spconv_wgrad(const float *in_features, const float *grad_out, const int *nbr, int n_out, int K, int Cin,
                                    int Cout, float *grad_weight, void *stream) {
    if (n_out < 0 || K <= 0 || Cin <= 0 || Cout <= 0 || Cin > IG_MAX_C || Cout > IG_MAX_C) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!in_features || !grad_out || !nbr || !grad_weight) return LIDAR_ERR_ARG;
    const size_t lds = (size_t)16 * (Cin + Cout) * sizeof(float);
    const dim3 grid(divup(n_out, WG_CHUNK), K);
    const int per = divup(Cin * Cout, 256);
#define WG_CASE(P) hipLaunchKernelGGL(sc_wgrad_kernel<P>, grid, dim3(256), lds, (hipStream_t)stream, in_features, grad_out, nbr, n_out, K, Cin, Cout, grad_weight)
    if (per <= 1) WG_CASE(1);
    else if (per <= 4) WG_CASE(4);
    else if (per <= 16) WG_CASE(16);
    else if (per <= 32) WG_CASE(32);
    else WG_CASE(64);
#undef WG_CASE
    return lidar_check_launch("lidar_spconv_wgrad");
}

// ------------------------------------------------------------------ weight gradient on the matrix cores
// dW[k] = A_k^T @ G_k with A_k = gathered input rows (zero where the neighbour is missing) and G_k = grad_out rows: the same
// FLOPs as the forward pass, reduced over rows.  grid = (K, NCH): a workgroup owns offset k and one chunk of rows, walks
// its 128-row tiles (rows in mask order when `order` is given, so tiles without a row using offset k are skipped after one
// 4-byte load per row), stages A (128 x Cin) and G (128 x Cout) in LDS and accumulates the Cin x Cout block in MFMA
// registers (32x32 tiles spread over the 4 waves).  Partials go to a (K, NCH, Cin, Cout) workspace and a second kernel sums
// them in chunk order: deterministic, no float atomics.
#define WGM_ROWS 128
// offsets handled per workgroup (they share the grad_out tile and the row scan).  Measured on the SECOND stack: 1 -> 19.2 ms
// per training step, 3 (where the accumulators fit) -> 24.6 ms: fewer, fatter workgroups lose more than the shared tile saves.
#define WGM_KG(NTI, NTO) 1
template <int NTI, int NTO>      // 32-wide tiles along Cin / Cout (channels padded up to the tile with zeros)
__global__ __launch_bounds__(256) void sc_wgrad_mfma_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                            const int *__restrict__ nbr, const int *__restrict__ order,
                                                            int n_out, int K, int Cin, int Cout, int rows_per_chunk,
                                                            float *__restrict__ part) {
    constexpr int SA = NTI * 32, SG = NTO * 32, NT = NTI * NTO, TPW = (NT + 3) / 4;   // tiles per wave
    constexpr int KG = WGM_KG(NTI, NTO);
    extern __shared__ float s_mem[];
    float *s_a = s_mem;                       // [WGM_ROWS][SA]
    float *s_g = s_mem + WGM_ROWS * SA;       // [WGM_ROWS][SG]
    __shared__ int s_src[KG][WGM_ROWS], s_row[WGM_ROWS];
    const int k0 = blockIdx.x * KG, chunk = blockIdx.y, t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int Ci4 = Cin >> 2, Co4 = Cout >> 2;
    for (int i = t; i < WGM_ROWS * (SA + SG); i += 256) s_mem[i] = 0.f;      // padding columns stay zero for good
    f32x16 acc[KG][TPW];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][q][r] = 0.f;
    const int i_begin = chunk * rows_per_chunk, i_end = min(i_begin + rows_per_chunk, n_out);
    const int ar = l & 31, ak = l >> 5;
    for (int i0 = i_begin; i0 < i_end; i0 += WGM_ROWS) {
        int src[KG], row = -1, bits = 0;
#pragma unroll
        for (int g = 0; g < KG; ++g) src[g] = -1;
        if (t < WGM_ROWS && i0 + t < i_end) {
            row = order ? order[i0 + t] : i0 + t;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                if (k0 + g < K) src[g] = nbr[(size_t)row * K + k0 + g];
                bits |= (src[g] >= 0) ? (1 << g) : 0;
            }
        }
        int work = 0;                                        // offsets of this group used by some row of the tile
#pragma unroll
        for (int g = 0; g < KG; ++g) work |= __syncthreads_or((bits >> g) & 1) ? (1 << g) : 0;   // (__syncthreads_or returns a flag,
        if (work == 0) continue;                             //  not the OR; the barriers also fence the LDS reuse)
        if (t < WGM_ROWS) {
#pragma unroll
            for (int g = 0; g < KG; ++g) s_src[g][t] = src[g];
            s_row[t] = row;
        }
        __syncthreads();
        for (int e = t; e < WGM_ROWS * Co4; e += 256) {      // grad_out rows of the tile: shared by the KG offsets
            const int r = e / Co4, c = e - r * Co4;
            const int rr = s_row[r];
            const float4 v = rr >= 0 ? reinterpret_cast<const float4 *>(dout)[(size_t)rr * Co4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(s_g + r * SG + c * 4) = v;
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (!((work >> g) & 1)) continue;                // block-uniform
            for (int e = t; e < WGM_ROWS * Ci4; e += 256) {
                const int r = e / Ci4, c = e - r * Ci4;
                const int sr = s_src[g][r];
                const float4 v = sr >= 0 ? reinterpret_cast<const float4 *>(in)[(size_t)sr * Ci4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(s_a + r * SA + c * 4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int tile = wv + 4 * q;
                if (tile < NT) {
                    const int ci0 = (tile / NTO) * 32, co0 = (tile % NTO) * 32;
#pragma unroll 8
                    for (int j = 0; j < WGM_ROWS; j += 2) {
                        const float a = s_a[(j + ak) * SA + ci0 + ar];
                        const float b = s_g[(j + ak) * SG + co0 + ar];
                        acc[g][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g][q], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                 // A tile is overwritten by the next offset / tile
        }
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); row = ci, col = co
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (k0 + g >= K) continue;
        float *P = part + ((size_t)(k0 + g) * gridDim.y + chunk) * Cin * Cout;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int tile = wv + 4 * q;
            if (tile < NT) {
                const int ci0 = (tile / NTO) * 32, co = (tile % NTO) * 32 + (l & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    if (ci < Cin && co < Cout) P[(size_t)ci * Cout + co] = acc[g][q][r];
                }
            }
        }
    }
}

__global__ void sc_wgrad_reduce_kernel(const float *__restrict__ part, int K, int nch, int cells, float *__restrict__ dW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)K * cells) return;
    const int k = (int)(i / cells), c = (int)(i - (long long)k * cells);
    float a = 0.f;
    for (int ch = 0; ch < nch; ++ch) a += part[((size_t)k * nch + ch) * cells + c];
    dW[i] = a;
}

static int sc_wgrad_chunks(int n_out) {
    // >= 8 row tiles per workgroup, at most 128 chunks: K x 128 workgroups of very unequal length (the centre offset is used by
    // every row, the corners by a third) balance over the 512 resident slots; with 32 chunks of >= 16 tiles (864 workgroups, 1.7
    // rounds) the slowest ones set the time: SECOND training step 18.8 -> 15.7 ms (64 x 8: 16.7, 256 x 4: 16.0)
    int nch = divup(n_out, WGM_ROWS * 8);
    return nch < 1 ? 1 : (nch > 128 ? 128 : nch);
}

LIDAR_EXPORT int lidar_spconv_wgrad_mfma_supported(int K, int Cin, int Cout) {
    return (K > 0 && (Cin == 16 || Cin == 32 || Cin == 64 || Cin == 128) && (Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128)) ? 1 : 0;
}

LIDAR_EXPORT size_t lidar_spconv_wgrad_workspace_bytes(int n_out, int K, int Cin, int Cout) {
    return (size_t)K * sc_wgrad_chunks(n_out) * Cin * Cout * sizeof(float);
}

// grad_weight (K, Cin, Cout) is overwritten (no zero fill needed).  order: mask order of nbr's rows or NULL.
LIDAR_EXPORT int lidar_spconv_wgrad_mfma(const float *in_features, const float *grad_out, const int *nbr, const int *order, int n_out,
                                         int K, int Cin, int Cout, float *grad_weight, void *ws, size_t ws_bytes, void *stream) {
    if (n_out < 0 || !lidar_spconv_wgrad_mfma_supported(K, Cin, Cout) || !grad_weight) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_out == 0) return hipMemsetAsync(grad_weight, 0, (size_t)K * Cin * Cout * sizeof(float), s) == hipSuccess ? LIDAR_OK : LIDAR_ERR_LAUNCH;
    if (!in_features || !grad_out || !nbr || !ws) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_spconv_wgrad_workspace_bytes(n_out, K, Cin, Cout)) return LIDAR_ERR_WORKSPACE;
    const int nch = sc_wgrad_chunks(n_out);
    const int rows_per_chunk = divup(divup(n_out, nch), WGM_ROWS) * WGM_ROWS;
    const int nti = divup(Cin, 32), nto = divup(Cout, 32);
    const size_t lds = (size_t)WGM_ROWS * (nti + nto) * 32 * sizeof(float);
    const int kg = WGM_KG(nti, nto);
    const dim3 grid(divup(K, kg), nch);
    float *part = (float *)ws;
#define WGM(I, O) hipLaunchKernelGGL((sc_wgrad_mfma_kernel<I, O>), grid, dim3(256), lds, s, in_features, grad_out, nbr, order, n_out, K, Cin, Cout, rows_per_chunk, part)
    switch (nti * 8 + nto) {
        case 1 * 8 + 1: WGM(1, 1); break; case 1 * 8 + 2: WGM(1, 2); break; case 1 * 8 + 4: WGM(1, 4); break;
        case 2 * 8 + 1: WGM(2, 1); break; case 2 * 8 + 2: WGM(2, 2); break; case 2 * 8 + 4: WGM(2, 4); break;
        case 4 * 8 + 1: WGM(4, 1); break; case 4 * 8 + 2: WGM(4, 2); break; case 4 * 8 + 4: WGM(4, 4); break;
        default: return LIDAR_ERR_ARG;
    }
#undef WGM
    const long long cells = (long long)K * Cin * Cout;
    hipLaunchKernelGGL(sc_wgrad_reduce_kernel, dim3(divup(cells, 256)), dim3(256), 0, s, part, K, nch, Cin * Cout, grad_weight);
    return lidar_check_launch("lidar_spconv_wgrad_mfma");
}
