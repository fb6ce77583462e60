// points -> voxels on gfx950, bit-identical to the sequential spconv VoxelGeneratorV2 scan
// (reference call site: pcdet/datasets/processor/data_processor.py:48-80; collate layout:
// pcdet/datasets/dataset.py:153-185).  One batched launch sequence for B frames.
//
// Parallel formulation of the sequential loop:
//   key(i)   = linear (z,y,x) cell of point i, or "outside"
//   first(k) = min{ i : key(i) = k }                       -> hash table + atomicMin
//   vid(k)   = rank of first(k) among all firsts           -> ordered prefix sum over points
//   voxels with vid >= max_voxels are dropped with all their points ("continue" semantics, v1.2)
//   slot(i)  = #{ i' < i : key(i') = key(i) }, kept iff slot < max_points
//            -> the max_points smallest point indices of each voxel, ascending: built with an
//               order-independent atomicMin insertion chain (each chain cell keeps the minimum of
//               everything that passes through it and forwards the rest).
// Kernels (HBM-bound; algorithmic bytes = 16*N in + V*(P*C*4 + 16 + 4) out per frame):
//   vx_hash   : 1 thread/point, coalesced float4 read of the raw N x 4 buffer, hash insert
//   vx_tile_sums / vx_assign : ordered scan over points (tile sums, then in-tile wave ballot scan)
//   vx_insert : per point, atomicMin chain into the compact per-voxel index list
//   vx_rows   : writes every padded voxel row exactly once with 16-B/lane stores (zeros included),
//               coords + counts, and restores the workspace (hash table, lists) to its clean state
#include "common.h"

#define VX_EMPTY 0xFFFFFFFFu
#define VX_INF 0x7FFFFFFF
#define VX_TILE 1024
#define VX_ROWS_PER_BLOCK 64

struct VxParams {
    float lo[3];
    float vs[3];
    int grid[3];  // nx, ny, nz
    int C, P, max_voxels, batch, n_max, compact;
    int H, hshift, ntiles;
};

struct VxWs {
    uint32_t *keys;  // [B][H]
    int *first;      // [B][H]
    int *cnt;        // [B][H]
    int *vid;        // [B][H]
    int *pslot;      // [B][n_max]
    int *list;       // [B][n_max]
    int *voff;       // [B][max_voxels]
    int *vcnt;       // [B][max_voxels]
    uint32_t *vcell; // [B][max_voxels]
    int *tile_sums;  // [B][ntiles][2]
    int *nvox;       // [B]
};

static int vx_hash_capacity(int n_max) {
    int h = 1024;
    while (h < 2 * n_max) h <<= 1;
    return h;
}

static size_t vx_carve(void *base, int B, int n_max, int max_voxels, VxWs *w) {
    const size_t H = (size_t)vx_hash_capacity(n_max);
    const int ntiles = divup(n_max > 0 ? n_max : 1, VX_TILE);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return (char *)base + o;
    };
    char *p;
    p = take(B * H * 4); if (w) w->keys = (uint32_t *)p;
    p = take(B * H * 4); if (w) w->first = (int *)p;
    p = take(B * H * 4); if (w) w->cnt = (int *)p;
    p = take(B * H * 4); if (w) w->vid = (int *)p;
    p = take((size_t)B * n_max * 4); if (w) w->pslot = (int *)p;
    p = take((size_t)B * n_max * 4); if (w) w->list = (int *)p;
    p = take((size_t)B * max_voxels * 4); if (w) w->voff = (int *)p;
    p = take((size_t)B * max_voxels * 4); if (w) w->vcnt = (int *)p;
    p = take((size_t)B * max_voxels * 4); if (w) w->vcell = (uint32_t *)p;
    p = take((size_t)B * ntiles * 2 * 4); if (w) w->tile_sums = (int *)p;
    p = take((size_t)B * 4); if (w) w->nvox = (int *)p;
    return off;
}

// ------------------------------------------------------------------ workspace init
__global__ void vx_ws_init_kernel(VxWs w, long long nh, long long nl) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = i; k < nh; k += stride) {
        w.keys[k] = VX_EMPTY;
        w.first[k] = VX_INF;
        w.cnt[k] = 0;
        w.vid[k] = -1;
    }
    for (long long k = i; k < nl; k += stride) {
        w.list[k] = VX_INF;
        w.pslot[k] = -1;
    }
}

// ------------------------------------------------------------------ K1: hash insert
__device__ __forceinline__ uint32_t vx_hash(uint32_t key, int hshift) {
    return (key * 2654435761u) >> hshift;
}

template <bool C4>
__global__ __launch_bounds__(256) void vx_hash_kernel(const float *__restrict__ points,
                                                      const int *__restrict__ offsets, VxParams p, VxWs w) {
    const int f = blockIdx.y;
    const int start = offsets[f];
    const int n = min(offsets[f + 1] - start, p.n_max);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x, y, z;
    if (C4) {
        const float4 v = reinterpret_cast<const float4 *>(points)[(size_t)start + i];
        x = v.x; y = v.y; z = v.z;
    } else {
        const float *q = points + ((size_t)start + i) * p.C;
        x = q[0]; y = q[1]; z = q[2];
    }
    // c = floor((p - lo) / vs) in fp32, IEEE division (same expression as the sequential scan)
    const float fx = floorf((x - p.lo[0]) / p.vs[0]);
    const float fy = floorf((y - p.lo[1]) / p.vs[1]);
    const float fz = floorf((z - p.lo[2]) / p.vs[2]);
    int *pslot = w.pslot + (size_t)f * p.n_max;
    const bool inside = (fx >= 0.f) & (fx < (float)p.grid[0]) & (fy >= 0.f) & (fy < (float)p.grid[1]) &
                        (fz >= 0.f) & (fz < (float)p.grid[2]);
    if (!inside) {
        pslot[i] = -1;
        return;
    }
    const uint32_t key = ((uint32_t)fz * (uint32_t)p.grid[1] + (uint32_t)fy) * (uint32_t)p.grid[0] + (uint32_t)fx;
    uint32_t *keys = w.keys + (size_t)f * p.H;
    const uint32_t mask = (uint32_t)p.H - 1u;
    uint32_t h = vx_hash(key, p.hshift);
    int slot = -1;
    for (int probe = 0; probe < p.H; ++probe) {
        const uint32_t old = atomicCAS(&keys[h], VX_EMPTY, key);
        if (old == VX_EMPTY || old == key) {
            slot = (int)h;
            break;
        }
        h = (h + 1u) & mask;
    }
    pslot[i] = slot;
    if (slot >= 0) {
        atomicMin(&w.first[(size_t)f * p.H + slot], i);
        atomicAdd(&w.cnt[(size_t)f * p.H + slot], 1);
    }
}

// ------------------------------------------------------------------ K2: ordered scan over points
// flags of point i: isfirst (opens a voxel) and w = min(count, P) list cells it reserves
__device__ __forceinline__ void vx_point_flags(const VxParams &p, const VxWs &w, int f, int i, int n, int &h,
                                               int &isf, int &wt) {
    h = -1;
    isf = 0;
    wt = 0;
    if (i < n) {
        h = w.pslot[(size_t)f * p.n_max + i];
        if (h >= 0 && w.first[(size_t)f * p.H + h] == i) {
            isf = 1;
            wt = min(w.cnt[(size_t)f * p.H + h], p.P);
        }
    }
}

__global__ __launch_bounds__(VX_TILE) void vx_tile_sums_kernel(const int *__restrict__ offsets, VxParams p, VxWs w) {
    __shared__ int s_f[16], s_w[16];
    const int f = blockIdx.y, tile = blockIdx.x;
    const int n = min(offsets[f + 1] - offsets[f], p.n_max);
    int h, isf, wt;
    vx_point_flags(p, w, f, tile * VX_TILE + (int)threadIdx.x, n, h, isf, wt);
    const int cf = __popcll(__ballot(isf));
    const int cw = wave_sum(wt);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_f[wv] = cf;
        s_w[wv] = cw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) {
            a += s_f[k];
            b += s_w[k];
        }
        w.tile_sums[((size_t)f * p.ntiles + tile) * 2 + 0] = a;
        w.tile_sums[((size_t)f * p.ntiles + tile) * 2 + 1] = b;
    }
}

__global__ __launch_bounds__(VX_TILE) void vx_assign_kernel(const int *__restrict__ offsets, VxParams p, VxWs w) {
    __shared__ int s_f[17], s_w[17], s_base[2];
    const int f = blockIdx.y, tile = blockIdx.x;
    const int n = min(offsets[f + 1] - offsets[f], p.n_max);
    const int wv = threadIdx.x >> 6, l = lane_id();
    // base = sum of the sums of all earlier tiles of this frame (wave 0)
    if (wv == 0) {
        int a = 0, b = 0;
        for (int t = l; t < tile; t += 64) {
            a += w.tile_sums[((size_t)f * p.ntiles + t) * 2 + 0];
            b += w.tile_sums[((size_t)f * p.ntiles + t) * 2 + 1];
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (l == 0) {
            s_base[0] = a;
            s_base[1] = b;
        }
    }
    int h, isf, wt;
    vx_point_flags(p, w, f, tile * VX_TILE + (int)threadIdx.x, n, h, isf, wt);
    const unsigned long long bal = __ballot(isf);
    const int ex_f = __popcll(bal & lanemask_lt());
    const int in_w = wave_incl_scan(wt);
    if (l == 63) {
        s_f[wv] = __popcll(bal);
        s_w[wv] = in_w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) {
            int ta = s_f[k], tb = s_w[k];
            s_f[k] = a;
            s_w[k] = b;
            a += ta;
            b += tb;
        }
        s_f[16] = a;
        s_w[16] = b;
    }
    __syncthreads();
    const int r = s_base[0] + s_f[wv] + ex_f;
    const int o = s_base[1] + s_w[wv] + (in_w - wt);
    if (isf) {
        if (r < p.max_voxels) {
            w.vid[(size_t)f * p.H + h] = r;
            w.voff[(size_t)f * p.max_voxels + r] = o;
            w.vcnt[(size_t)f * p.max_voxels + r] = wt;
            w.vcell[(size_t)f * p.max_voxels + r] = w.keys[(size_t)f * p.H + h];
        } else {
            w.vid[(size_t)f * p.H + h] = -1;
        }
    }
    if (tile == (int)gridDim.x - 1 && threadIdx.x == 0) {
        w.nvox[f] = min(s_base[0] + s_f[16], p.max_voxels);
    }
}

// ------------------------------------------------------------------ K3: ordered per-voxel lists
__global__ __launch_bounds__(256) void vx_insert_kernel(const int *__restrict__ offsets, VxParams p, VxWs w) {
    const int f = blockIdx.y;
    const int n = min(offsets[f + 1] - offsets[f], p.n_max);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int h = w.pslot[(size_t)f * p.n_max + i];
    if (h < 0) return;
    const int r = w.vid[(size_t)f * p.H + h];
    if (r < 0) return;
    const int m = w.vcnt[(size_t)f * p.max_voxels + r];
    int *L = w.list + (size_t)f * p.n_max + w.voff[(size_t)f * p.max_voxels + r];
    if (w.cnt[(size_t)f * p.H + h] == 1) {  // the common case: one point in the voxel
        L[0] = i;
        return;
    }
    int x = i;
    for (int s = 0; s < m; ++s) {
        const int old = atomicMin(&L[s], x);
        if (old == VX_INF) break;  // cell was free: x is stored, nothing to forward
        x = max(old, x);           // cell keeps min(old, x); the larger one moves on
    }
}

// ------------------------------------------------------------------ K4: write rows + restore workspace
__device__ __forceinline__ int vx_frame_base(const VxParams &p, const VxWs &w, int f) {
    if (!p.compact) return f * p.max_voxels;
    int b = 0;
    for (int k = 0; k < f; ++k) b += w.nvox[k];
    return b;
}

template <bool C4>
__global__ __launch_bounds__(256) void vx_rows_kernel(const float *__restrict__ points,
                                                      const int *__restrict__ offsets, VxParams p, VxWs w,
                                                      float *__restrict__ voxels, int *__restrict__ coords,
                                                      int *__restrict__ num_points, int *__restrict__ voxel_offsets,
                                                      int row_blocks) {
    const int f = blockIdx.y;
    const int start = offsets[f];
    if ((int)blockIdx.x >= row_blocks) {
        // ---- cleanup role: restore the hash table cells touched by this frame's points
        const int n = min(offsets[f + 1] - start, p.n_max);
        const int i = ((int)blockIdx.x - row_blocks) * 256 + threadIdx.x;
        if (i < n) {
            const int h = w.pslot[(size_t)f * p.n_max + i];
            if (h >= 0) {
                w.keys[(size_t)f * p.H + h] = VX_EMPTY;
                w.first[(size_t)f * p.H + h] = VX_INF;
                w.cnt[(size_t)f * p.H + h] = 0;
            }
        }
        if (blockIdx.x == (unsigned)row_blocks && f == 0 && threadIdx.x == 0) {
            int b = 0;
            for (int k = 0; k < p.batch; ++k) {
                voxel_offsets[k] = p.compact ? b : k * p.max_voxels;
                b += w.nvox[k];
            }
            voxel_offsets[p.batch] = p.compact ? b : p.batch * p.max_voxels;
        }
        return;
    }
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = vx_frame_base(p, w, f);
    __syncthreads();
    const int base = s_base;
    const int nv = w.nvox[f];
    const int row0 = blockIdx.x * VX_ROWS_PER_BLOCK;
    if (row0 >= nv) return;
    const int rows = min(VX_ROWS_PER_BLOCK, nv - row0);
    const int *voff = w.voff + (size_t)f * p.max_voxels;
    const int *vcnt = w.vcnt + (size_t)f * p.max_voxels;
    int *list = w.list + (size_t)f * p.n_max;
    if (C4) {
        // one float4 (= one point slot) per item; a wave stores 1 KiB contiguous
        const int items = rows * p.P;
        float4 *out4 = reinterpret_cast<float4 *>(voxels) + (size_t)(base + row0) * p.P;
        const float4 *pts4 = reinterpret_cast<const float4 *>(points) + start;
        for (int it = threadIdx.x; it < items; it += 256) {
            const int rr = it / p.P, slot = it - rr * p.P;
            const int r = row0 + rr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < vcnt[r]) {
                const int li = voff[r] + slot;
                const int pi = list[li];
                list[li] = VX_INF;  // restore
                v = pts4[pi];
            }
            out4[it] = v;
        }
    } else {
        const int rowlen = p.P * p.C;
        const int items = rows * rowlen;
        float *out = voxels + (size_t)(base + row0) * rowlen;
        const float *pts = points + (size_t)start * p.C;
        for (int it = threadIdx.x; it < items; it += 256) {
            const int rr = it / rowlen, e = it - rr * rowlen;
            const int slot = e / p.C, ch = e - slot * p.C;
            const int r = row0 + rr;
            float v = 0.f;
            if (slot < vcnt[r]) v = pts[(size_t)list[voff[r] + slot] * p.C + ch];
            out[it] = v;
        }
        __syncthreads();  // all reads of list done before restoring it
        for (int it = threadIdx.x; it < rows * p.P; it += 256) {
            const int rr = it / p.P, slot = it - rr * p.P;
            const int r = row0 + rr;
            if (slot < vcnt[r]) list[voff[r] + slot] = VX_INF;
        }
    }
    // coords (b, z, y, x) and per-voxel counts
    for (int rr = threadIdx.x; rr < rows; rr += 256) {
        const int r = row0 + rr;
        const uint32_t cell = w.vcell[(size_t)f * p.max_voxels + r];
        const uint32_t nx = p.grid[0], ny = p.grid[1];
        const int cx = (int)(cell % nx), cy = (int)((cell / nx) % ny), cz = (int)(cell / (nx * ny));
        reinterpret_cast<int4 *>(coords)[base + r] = make_int4(f, cz, cy, cx);
        num_points[base + r] = vcnt[r];
    }
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_voxelize_workspace_bytes(int batch, int n_max, int max_voxels) {
    if (batch <= 0 || n_max < 0 || max_voxels <= 0) return 0;
    return vx_carve(nullptr, batch, n_max > 0 ? n_max : 1, max_voxels, nullptr);
}

LIDAR_EXPORT int lidar_voxelize_workspace_init(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels,
                                               void *stream) {
    if (!ws || batch <= 0 || max_voxels <= 0) return LIDAR_ERR_ARG;
    if (n_max <= 0) n_max = 1;
    VxWs w;
    if (vx_carve(ws, batch, n_max, max_voxels, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    const long long nh = (long long)batch * vx_hash_capacity(n_max), nl = (long long)batch * n_max;
    hipLaunchKernelGGL(vx_ws_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, nh, nl);
    return lidar_check_launch("vx_ws_init");
}

LIDAR_EXPORT int lidar_voxelize(const float *points, const int *point_offsets, int batch, int n_max,
                                int num_features, const float *range6, const float *voxel_size3,
                                const int *grid3, int max_points, int max_voxels, int compact, float *voxels,
                                int *coords, int *num_points, int *voxel_offsets, void *ws, size_t ws_bytes,
                                void *stream) {
    if (!points || !point_offsets || !voxels || !coords || !num_points || !voxel_offsets || !ws) return LIDAR_ERR_ARG;
    if (batch <= 0 || num_features < 3 || max_points <= 0 || max_voxels <= 0 || n_max < 0) return LIDAR_ERR_ARG;
    if ((double)grid3[0] * grid3[1] * grid3[2] >= 4294967295.0) return LIDAR_ERR_ARG;
    if (n_max == 0) n_max = 1;
    VxParams p;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = range6[j];
        p.vs[j] = voxel_size3[j];
        p.grid[j] = grid3[j];
    }
    p.C = num_features;
    p.P = max_points;
    p.max_voxels = max_voxels;
    p.batch = batch;
    p.n_max = n_max;
    p.compact = compact;
    p.H = vx_hash_capacity(n_max);
    int hb = 0;
    while ((1 << hb) < p.H) ++hb;
    p.hshift = 32 - hb;
    p.ntiles = divup(n_max, VX_TILE);
    VxWs w;
    if (vx_carve(ws, batch, n_max, max_voxels, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool c4 = (num_features == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(voxels) & 15) == 0);
    const dim3 gpt(divup(n_max, 256), batch), gtile(p.ntiles, batch);
    if (c4)
        hipLaunchKernelGGL(vx_hash_kernel<true>, gpt, dim3(256), 0, s, points, point_offsets, p, w);
    else
        hipLaunchKernelGGL(vx_hash_kernel<false>, gpt, dim3(256), 0, s, points, point_offsets, p, w);
    hipLaunchKernelGGL(vx_tile_sums_kernel, gtile, dim3(VX_TILE), 0, s, point_offsets, p, w);
    hipLaunchKernelGGL(vx_assign_kernel, gtile, dim3(VX_TILE), 0, s, point_offsets, p, w);
    hipLaunchKernelGGL(vx_insert_kernel, gpt, dim3(256), 0, s, point_offsets, p, w);
    const int row_blocks = divup(max_voxels, VX_ROWS_PER_BLOCK);
    const dim3 grow(row_blocks + divup(n_max, 256), batch);
    if (c4)
        hipLaunchKernelGGL(vx_rows_kernel<true>, grow, dim3(256), 0, s, points, point_offsets, p, w, voxels, coords,
                           num_points, voxel_offsets, row_blocks);
    else
        hipLaunchKernelGGL(vx_rows_kernel<false>, grow, dim3(256), 0, s, points, point_offsets, p, w, voxels, coords,
                           num_points, voxel_offsets, row_blocks);
    return lidar_check_launch("lidar_voxelize");
}
