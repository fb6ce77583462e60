// Rulebooks through DENSE index grids (the 288 GB way).  Same tables as the hash-table builder in sparse_conv.hip, bit for bit
// — same first-touch numbering of a strided convolution's output sites — but a coordinate lookup is ONE load from a
// persistent (B, D, H, W) int32 grid instead of a probe sequence through a (u64 key, i32 value) hash table, the three
// x-neighbours of a site share a cache line, and nothing has to be built or cleared per table: a level's grid is written once
// (one store per active site), serves the SubM table of the level, the tables of the strided convolution that leaves it AND the
// output-site search of the one that enters it, and is wiped by revisiting the same sites.
// The SECOND-KITTI input level is 16 x 41 x 1600 x 1408 cells = 5.9 GB of int32: nothing on a 288 GB part.  Grids are owned by
// the caller (include/lidar_hip.h: "empty" = RG_EMPTY everywhere between uses).
// Reference boundary: spconv.ops.get_indice_pairs (external; call sites pcdet/models/backbones_3d/spconv_backbone.py:11-16,77,
// 89-113); semantics SURVEY.md Appendix A.2.
#include "common.h"

#define RG_EMPTY 0x7FFFFFFF

struct RgGeom {
    int B;                   // frames the grids were allocated for: a row whose batch index is >= B is treated like a padding row
    int D, H, W;             // input spatial shape
    int oD, oH, oW;          // output spatial shape
    int kD, kH, kW, K;
    int sD, sH, sW, pD, pH, pW;
};

__device__ __forceinline__ size_t rg_cell(int b, int z, int y, int x, int D, int H, int W) {
    return (((size_t)b * D + z) * H + y) * W + x;
}

__global__ void rg_fill_kernel(int *__restrict__ grid, size_t cells) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < cells; i += stride) {
        if (i + 4 <= cells) *reinterpret_cast<int4 *>(grid + i) = make_int4(RG_EMPTY, RG_EMPTY, RG_EMPTY, RG_EMPTY);
        else for (size_t j = i; j < cells; ++j) grid[j] = RG_EMPTY;
    }
}

// mode 0: grid[cell of row i] = min(., i) (rows of an input tensor: duplicate coordinates keep the lowest row, as the hash builder
// does); mode 1: the same cells back to empty; mode 2: grid[cell] = i (unique output sites over the candidate ids left behind by
// lidar_spconv_grid_outputs)
__global__ void rg_scatter_rows_kernel(const int *__restrict__ indices, int n, const int *__restrict__ n_dev, int B, int D, int H, int W,
                                       int *__restrict__ grid, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (n_dev ? min(*n_dev, n) : n)) return;
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    if (c.x < 0 || c.x >= B || c.y < 0 || c.y >= D || c.z < 0 || c.z >= H || c.w < 0 || c.w >= W) return;      // never outside the grid
    int *cell = grid + rg_cell(c.x, c.y, c.z, c.w, D, H, W);
    if (mode == 1) *cell = RG_EMPTY;
    else if (mode == 2) *cell = i;
    else atomicMin(cell, i);
}

// forward table of a SubM or regular convolution, output-stationary: nbr[j][k] = row at input site o * s - p + k, or -1
// A row whose batch index is negative is a PADDING row (capacity-sized tensors, see lidar_spconv_grid_pad_rows): it has no
// neighbours, reaches no output and is never scattered.  limit: values >= it are not rows of the looked-up tensor (-> -1).
__global__ void rg_table_kernel(const int *__restrict__ out_indices, int n_out, RgGeom g, const int *__restrict__ grid_in,
                                int limit, int *__restrict__ nbr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n_out * g.K) return;
    const int j = (int)(e / g.K), k = (int)(e - (long long)j * g.K);
    const int kz = k / (g.kH * g.kW), ky = (k / g.kW) % g.kH, kx = k % g.kW;
    const int4 c = reinterpret_cast<const int4 *>(out_indices)[j];
    const int z = c.y * g.sD - g.pD + kz, y = c.z * g.sH - g.pH + ky, x = c.w * g.sW - g.pW + kx;
    int r = -1;
    if (c.x >= 0 && c.x < g.B && z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W) {
        const int v = grid_in[rg_cell(c.x, z, y, x, g.D, g.H, g.W)];
        r = (v >= limit) ? -1 : v;
    }
    nbr[e] = r;
}

// output site reached from input site (z, y, x) through offset k: false when it does not divide / is out of bounds
__device__ __forceinline__ bool rg_out_of(const RgGeom &g, int z, int y, int x, int k, int &oz, int &oy, int &ox) {
    const int kz = k / (g.kH * g.kW), ky = (k / g.kW) % g.kH, kx = k % g.kW;
    const int tz = z + g.pD - kz, ty = y + g.pH - ky, tx = x + g.pW - kx;
    if (tz < 0 || ty < 0 || tx < 0 || tz % g.sD || ty % g.sH || tx % g.sW) return false;
    oz = tz / g.sD; oy = ty / g.sH; ox = tx / g.sW;
    return oz < g.oD && oy < g.oH && ox < g.oW;
}

// transposed table: nbr_t[i][k] = output row reached from input i through offset k, or -1
__global__ void rg_table_t_kernel(const int *__restrict__ indices, int n, RgGeom g, const int *__restrict__ grid_out,
                                  int limit, int *__restrict__ nbr_t) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * g.K) return;
    const int i = (int)(e / g.K), k = (int)(e - (long long)i * g.K);
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    int oz, oy, ox, r = -1;
    if (c.x >= 0 && c.x < g.B && rg_out_of(g, c.y, c.z, c.w, k, oz, oy, ox)) {
        const int v = grid_out[rg_cell(c.x, oz, oy, ox, g.oD, g.oH, g.oW)];
        r = (v >= limit) ? -1 : v;
    }
    nbr_t[e] = r;
}

// the same table from the forward table alone (no grid): nbr[j][k] == i  <=>  nbr_t[i][k] == j.  nbr_t arrives filled with -1.
__global__ void rg_transpose_table_kernel(const int *__restrict__ nbr, long long n_entries, int K, int n_in, int *__restrict__ nbr_t) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const int i = nbr[e];
    if (i < 0 || i >= n_in) return;
    const int j = (int)(e / K), k = (int)(e - (long long)j * K);
    nbr_t[(long long)i * K + k] = j;
}

// rows [min(*num, cap), cap) of a capacity-sized coordinate tensor -> padding rows (all -1)
__global__ void rg_pad_rows_kernel(int *__restrict__ out_indices, const int *__restrict__ num, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && i >= *num) reinterpret_cast<int4 *>(out_indices)[i] = make_int4(-1, -1, -1, -1);
}

// ---- unique output sites of a regular convolution, numbered in first-touch order of the (input row, offset) scan:
// candidate e = i * K + k; the output grid cell keeps the smallest candidate that reaches it (phase 1), the candidates that
// own their cell are ranked by an ordered count (phases 2-4), and the ranks are scattered into the grid afterwards by
// rg_scatter_rows on out_indices (a cell must not change while other candidates still compare themselves with its owner).
__global__ void rg_candidates_kernel(const int *__restrict__ indices, int n, RgGeom g, int *__restrict__ grid_out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * g.K) return;
    const int i = (int)(e / g.K), k = (int)(e - (long long)i * g.K);
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    int oz, oy, ox;
    if (c.x >= 0 && c.x < g.B && rg_out_of(g, c.y, c.z, c.w, k, oz, oy, ox)) atomicMin(grid_out + rg_cell(c.x, oz, oy, ox, g.oD, g.oH, g.oW), (int)e);
}

#define RG_TPB 1024
// does candidate e own its output cell?  (oc = that cell's coordinates when it is in bounds at all)
__device__ __forceinline__ bool rg_candidate_cell(const int *__restrict__ indices, long long e, long long nc, const RgGeom &g, int4 &oc) {
    if (e >= nc) return false;
    const int i = (int)(e / g.K), k = (int)(e - (long long)i * g.K);
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    int oz, oy, ox;
    if (c.x < 0 || !rg_out_of(g, c.y, c.z, c.w, k, oz, oy, ox)) return false;
    oc = make_int4(c.x, oz, oy, ox);
    return true;
}

// pass 2: owner flags (one ballot word per wave, kept for pass 4: no second random walk through the grid) + per-block counts
__global__ __launch_bounds__(RG_TPB) void rg_count_kernel(const int *__restrict__ indices, int n, RgGeom g,
                                                          const int *__restrict__ grid_out, int *__restrict__ block_sums,
                                                          unsigned long long *__restrict__ owner_bits) {
    __shared__ int s_w[16];
    const long long e = (long long)blockIdx.x * RG_TPB + threadIdx.x;
    int4 oc;
    bool f = rg_candidate_cell(indices, e, (long long)n * g.K, g, oc);
    if (f) f = grid_out[rg_cell(oc.x, oc.y, oc.z, oc.w, g.oD, g.oH, g.oW)] == (int)e;
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = __popcll(bal);
        owner_bits[e >> 6] = bal;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0;
        for (int k = 0; k < 16; ++k) a += s_w[k];
        block_sums[blockIdx.x] = a;
    }
}

__global__ __launch_bounds__(1024) void rg_scan_sums_kernel(int *__restrict__ block_sums, int nblocks, int *__restrict__ total) {
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

__global__ __launch_bounds__(RG_TPB) void rg_assign_kernel(const int *__restrict__ indices, int n, RgGeom g,
                                                           const unsigned long long *__restrict__ owner_bits,
                                                           const int *__restrict__ block_sums, int *__restrict__ out_indices) {
    __shared__ int s_w[16];
    const long long e = (long long)blockIdx.x * RG_TPB + threadIdx.x;
    const unsigned long long bal = owner_bits[e >> 6];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if ((bal >> (threadIdx.x & 63)) & 1ull) {
        int4 oc = make_int4(0, 0, 0, 0);
        rg_candidate_cell(indices, e, (long long)n * g.K, g, oc);
        int r = block_sums[blockIdx.x] + __popcll(bal & lanemask_lt());
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) r += s_w[k];
        reinterpret_cast<int4 *>(out_indices)[r] = oc;
    }
}

// ------------------------------------------------------------------ C ABI
static bool rg_geom(RgGeom &g, int B, int D, int H, int W, int kD, int kH, int kW, int sD, int sH, int sW, int pD, int pH, int pW) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || kD <= 0 || kH <= 0 || kW <= 0 || sD <= 0 || sH <= 0 || sW <= 0) return false;
    g.B = B; g.D = D; g.H = H; g.W = W; g.kD = kD; g.kH = kH; g.kW = kW; g.K = kD * kH * kW;
    g.sD = sD; g.sH = sH; g.sW = sW; g.pD = pD; g.pH = pH; g.pW = pW;
    g.oD = (D + 2 * pD - kD) / sD + 1; g.oH = (H + 2 * pH - kH) / sH + 1; g.oW = (W + 2 * pW - kW) / sW + 1;
    return g.oD > 0 && g.oH > 0 && g.oW > 0;
}

// every cell of a fresh grid -> empty (once per grid)
LIDAR_EXPORT int lidar_spconv_grid_init(int *grid, size_t cells, void *stream) {
    if (!grid || cells == 0) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rg_fill_kernel, dim3(4096), dim3(1024), 0, (hipStream_t)stream, grid, cells);
    return lidar_check_launch("lidar_spconv_grid_init");
}

// mode 0: rows -> grid (lowest row wins on duplicate coordinates); 1: the same cells back to empty; 2: rows -> grid by plain store
// (after lidar_spconv_grid_outputs).  n_dev: optional device count (rows >= it are skipped).  batch: frames the grid was allocated
// for — a row whose batch index is outside [0, batch) is skipped like a padding row, never an out-of-bounds store.
LIDAR_EXPORT int lidar_spconv_grid_rows(const int *indices, int n, const int *n_dev, int batch, int D, int H, int W, int *grid, int mode,
                                        void *stream) {
    if (n < 0 || batch <= 0 || D <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!indices || !grid) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rg_scatter_rows_kernel, dim3(divup(n, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, n_dev, batch, D, H, W,
                       grid, mode);
    return lidar_check_launch("lidar_spconv_grid_rows");
}

// forward table: SubM (stride 1, padding k / 2, out_indices = the input rows) or regular convolution; grid_in = input level
// limit: number of rows of the tensor held by grid_in (grid values >= it read as "no row"; pass INT_MAX-like 0x7FFFFFFF for none)
LIDAR_EXPORT int lidar_spconv_grid_table(const int *out_indices, int n_out, int batch, int D, int H, int W, int kD, int kH, int kW, int sD,
                                         int sH, int sW, int pD, int pH, int pW, const int *grid_in, int limit, int *nbr, void *stream) {
    RgGeom g;
    if (n_out < 0 || !rg_geom(g, batch, D, H, W, kD, kH, kW, sD, sH, sW, pD, pH, pW)) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!out_indices || !grid_in || !nbr) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rg_table_kernel, dim3(divup((long long)n_out * g.K, 256)), dim3(256), 0, (hipStream_t)stream, out_indices, n_out,
                       g, grid_in, limit > 0 ? limit : RG_EMPTY, nbr);
    return lidar_check_launch("lidar_spconv_grid_table");
}

// transposed table nbr_t (n, K) of a regular convolution; grid_out = output level (rows scattered)
LIDAR_EXPORT int lidar_spconv_grid_table_t(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW, int sD, int sH,
                                           int sW, int pD, int pH, int pW, const int *grid_out, int limit, int *nbr_t, void *stream) {
    RgGeom g;
    if (n < 0 || !rg_geom(g, batch, D, H, W, kD, kH, kW, sD, sH, sW, pD, pH, pW)) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!indices || !grid_out || !nbr_t) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rg_table_t_kernel, dim3(divup((long long)n * g.K, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, g,
                       grid_out, limit > 0 ? limit : RG_EMPTY, nbr_t);
    return lidar_check_launch("lidar_spconv_grid_table_t");
}

LIDAR_EXPORT size_t lidar_spconv_grid_outputs_workspace_bytes(int n, int K) {
    const size_t nc = (size_t)(n > 0 ? n : 1) * (K > 0 ? K : 1);
    return align_up((nc / RG_TPB + 2) * 4, 256) + align_up((nc / 64 + 16) * 8, 256) + 256;
}

// unique output sites of SparseConv3d: out_indices (>= n * prod ceil(k / s) rows, 4) in first-touch order, *num_out (device).
// grid_out: the OUTPUT level's grid, all empty on entry; on return it holds candidate ids — the caller reads *num_out and then
// calls lidar_spconv_grid_rows(out_indices, ...) on it, which overwrites exactly those cells with the output rows.
LIDAR_EXPORT int lidar_spconv_grid_outputs(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW, int sD, int sH,
                                           int sW, int pD, int pH, int pW, int *grid_out, int *out_indices, int *num_out, void *ws,
                                           size_t ws_bytes, void *stream) {
    RgGeom g;
    if (n < 0 || !num_out || !rg_geom(g, batch, D, H, W, kD, kH, kW, sD, sH, sW, pD, pH, pW)) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(num_out, 0, 4, s) == hipSuccess ? LIDAR_OK : LIDAR_ERR_LAUNCH;
    if (!indices || !grid_out || !out_indices || !ws) return LIDAR_ERR_ARG;
    const long long nc = (long long)n * g.K;
    if (nc > 0x3FFFFFFF) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_spconv_grid_outputs_workspace_bytes(n, g.K)) return LIDAR_ERR_WORKSPACE;
    int *block_sums = (int *)ws;
    const int nblocks = divup(nc, RG_TPB);
    unsigned long long *owner_bits = (unsigned long long *)((char *)ws + align_up(((size_t)nc / RG_TPB + 2) * 4, 256));
    hipLaunchKernelGGL(rg_candidates_kernel, dim3(divup(nc, 256)), dim3(256), 0, s, indices, n, g, grid_out);
    hipLaunchKernelGGL(rg_count_kernel, dim3(nblocks), dim3(RG_TPB), 0, s, indices, n, g, grid_out, block_sums, owner_bits);
    hipLaunchKernelGGL(rg_scan_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nblocks, num_out);
    hipLaunchKernelGGL(rg_assign_kernel, dim3(nblocks), dim3(RG_TPB), 0, s, indices, n, g, owner_bits, block_sums, out_indices);
    return lidar_check_launch("lidar_spconv_grid_outputs");
}

// rows [min(*num_dev, cap), cap) of out_indices (cap, 4) -> padding rows (batch index -1): a capacity-sized coordinate tensor
// whose true row count lives on the device.  Padding rows have no neighbours (tables hold -1), reach no output and are never
// scattered into a grid, so every kernel of this file can run on the capacity without reading the count.
LIDAR_EXPORT int lidar_spconv_grid_pad_rows(int *out_indices, const int *num_dev, int cap, void *stream) {
    if (cap < 0) return LIDAR_ERR_ARG;
    if (cap == 0) return LIDAR_OK;
    if (!out_indices || !num_dev) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rg_pad_rows_kernel, dim3(divup(cap, 256)), dim3(256), 0, (hipStream_t)stream, out_indices, num_dev, cap);
    return lidar_check_launch("lidar_spconv_grid_pad_rows");
}

// transposed table from the forward table: nbr (n_out, K) -> nbr_t (n_in, K), which the CALLER has filled with -1.
// Equals lidar_spconv_grid_table_t / lidar_spconv_conv_tables' nbr_t when the input coordinates are unique.
LIDAR_EXPORT int lidar_spconv_transpose_table(const int *nbr, int n_out, int K, int n_in, int *nbr_t, void *stream) {
    if (n_out < 0 || n_in < 0 || K <= 0) return LIDAR_ERR_ARG;
    if (n_out == 0 || n_in == 0) return LIDAR_OK;
    if (!nbr || !nbr_t) return LIDAR_ERR_ARG;
    const long long ne = (long long)n_out * K;
    hipLaunchKernelGGL(rg_transpose_table_kernel, dim3(divup(ne, 256)), dim3(256), 0, (hipStream_t)stream, nbr, ne, K, n_in, nbr_t);
    return lidar_check_launch("lidar_spconv_transpose_table");
}
