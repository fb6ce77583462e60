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
#include <stdlib.h>

#define VX_EMPTY 0xFFFFFFFFu
#define VX_INF 0x7FFFFFFF
#define VX_TILE 1024
#define VX_ROWS_PER_BLOCK 64
#ifndef VXL_PTS_PER_BIN
#define VXL_PTS_PER_BIN 2560   // LDS-binned path: expected points per hash bin (bins per frame = n_max / this)
#endif

typedef float vx_f4 __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-B store: the padded voxel rows are written once and not re-read here
__device__ __forceinline__ void vx_store_nt(float4 *dst, float4 v) {
    vx_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vx_f4 *>(dst));
}

struct VxParams {
    float lo[3];
    float vs[3];
    float rvs[3];  // fl(1 / vs): only used to skip the IEEE division when the quotient is far from an integer
    float eabs[3]; // fused launch: absolute form of the same margin, (grid + 2) * 4.8e-7 (valid for every in-grid quotient)
    int grid[3];  // nx, ny, nz
    int C, P, max_voxels, batch, n_max, compact;
    int H, hshift, ntiles;
};

struct VxWs {
    // ---- LDS-binned path (algo 1)
    int *pfirst;     // [B][n_max] index of the first point of the point's voxel (-1 = outside)
    int *flagw;      // [B][n_max] first points: min(count,P) | VX_SINGLE ; others 0
    int2 *vinfo;     // [B][n_max] at first points: (voxel rank or -1, list offset)
    int *err;        // [1] sticky error flag (LDS table / entry list overflow)
    int2 *queue;     // [B][tile][G][1024] (point, key): each 1024-point tile partitioned by bin
    int *qcnt;       // [B][tile][G] entries per (tile, bin) segment
    int *tcnt;       // [B][G][32] first points of bin g per 1024-point tile (the emit stage turns them into voxel ids)
    // ---- global-hash path (algo 2)
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
    int *fillst;     // [2] LDS path, compact mode: [0] rows the NEXT call should pre-clear (last total + 25 % + 1024),
                     //     [1] the value this call's fill role used (copied by the bin launch; read by the emit launch)
    int *bvox;       // [B][VXL_GMAX] fused launch: first points (= voxels) found by bin g of frame f (plain stores, no zeroing)
    long long *resident;  // [4] resident-output mode (algo 4): {voxels ptr, num_points ptr, rows the previous call produced, P * C};
                          //     [0] == 0: no valid history (the next resident call clears the whole buffer)
    int **mirror;    // [1] optional device-visible HOST address (pinned, mapped) that also receives the error bits, so the
                     //     host can poll the flag without a copy or a sync (lidar_voxelize_set_error_mirror); null = none
};

// error bits: 1 = LDS table full, 2 = bin entry / list capacity exceeded.  Only ever reached on degenerate input, so the
// extra pointer load and the system-scope atomic cost nothing on the normal path.
__device__ __noinline__ void vx_raise_at(int *err, int **mirror, int bits) {
    atomicOr(err, bits);
    int *m = *mirror;
    if (m) __hip_atomic_fetch_or(m, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void vx_raise(const VxWs &w, int bits) { vx_raise_at(w.err, w.mirror, bits); }


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
    p = take((size_t)B * (divup(n_max, VXL_PTS_PER_BIN) * 6144) * 4); if (w) w->pfirst = (int *)p;  // also the LDS path's staging lists
    p = take((size_t)B * n_max * 4); if (w) w->flagw = (int *)p;
    p = take((size_t)B * n_max * 8); if (w) w->vinfo = (int2 *)p;
    p = take(65536); if (w) w->err = (int *)p;  // [0] sticky error flag
    p = take((size_t)B * divup(n_max, 1024) * divup(n_max, VXL_PTS_PER_BIN) * 1024 * 8 + 65536); if (w) w->queue = (int2 *)p;  // [B][tile][G][1024]
    p = take((size_t)B * divup(n_max, 1024) * divup(n_max, VXL_PTS_PER_BIN) * 4 + 256); if (w) w->qcnt = (int *)p;             // [B][tile][G]
    p = take((size_t)B * divup(n_max, VXL_PTS_PER_BIN) * 32 * 4 + 256); if (w) w->tcnt = (int *)p;                               // [B][G][32]
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
    p = take(256); if (w) w->fillst = (int *)p;
    p = take((size_t)B * 16 * 4 + 256); if (w) w->bvox = (int *)p;
    p = take(256); if (w) w->resident = (long long *)p;
    p = take(256); if (w) w->mirror = (int **)p;
    return off;
}

// ------------------------------------------------------------------ workspace init
__global__ void vx_ws_init_kernel(VxWs w, long long nh, long long nl, long long nq) {
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
    if (i == 0) {
        *w.err = 0;
        *w.mirror = nullptr;
        w.resident[0] = w.resident[1] = w.resident[2] = w.resident[3] = 0;
        w.fillst[0] = w.fillst[1] = 0x7fffffff;     // no history yet: clear the whole buffer
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
            w.resident[0] = 0;                       // this path leaves no resident-output history
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


// ================================================================== LDS-binned path (algo 1)
// No global atomics at all.  A frame's points are hash-partitioned by voxel key into G bins by the key kernel (whose
// launch also zero-fills the whole padded output, see vxl_key_kernel); workgroup (g, f) of the bin kernel takes its bin's
// (point, key) pairs and resolves everything that is local to a voxel inside LDS with LDS atomics:
//   first point, point count, and the ascending list of its first P point indices (the same
//   order-independent atomicMin insertion chain as the global path, but on LDS).
// It leaves one 32-bit word per point (0, or for a voxel's first point: count | list position) and
// the bin's packed index lists.  A one-block-per-frame ballot scan then ranks the first points
// (= voxel ids in first-appearance order), and the scatter stage writes the occupied slots, coords and counts.
#define VXL_S 8192          // LDS table slots per bin
#define VXL_CAP 6144        // LDS entry / list capacity (points per bin)
#define VXL_MBITS 14        // pinfo word: m = min(count, P) in the low 14 bits, list position above
#define VXL_MMASK ((1 << VXL_MBITS) - 1)
#define VXL_MAX_ITEMS 32    // rank kernel: n_max <= 32 * 1024
#define VXL_U 8             // points per thread per prefetch step

// floor(fl(d / vs)) — the reference's expression — without paying for the IEEE division when it
// cannot matter: q' = fl(d * fl(1/vs)) is within 2^-22 (relative) of the true quotient, and so is
// fl(d / vs); if no integer lies that close to q', both floors are the floor of the true quotient.
// Otherwise (≈0.1 % of coordinates, NaN/huge values included) the exact division is evaluated.
__device__ __forceinline__ float vx_floor_div(float d, float vs, float rvs) {
    float q = d * rvs;
    if (!(fabsf(q - rintf(q)) > fabsf(q) * 4.8e-7f)) {
        asm volatile("" ::: "memory");  // keep this a real (rarely taken) branch: no if-conversion of the division
        q = d / vs;
    }
    return floorf(q);
}

__device__ __forceinline__ bool vx_cell(const VxParams &p, float x, float y, float z, uint32_t &key) {
    const float fx = vx_floor_div(x - p.lo[0], p.vs[0], p.rvs[0]);
    const float fy = vx_floor_div(y - p.lo[1], p.vs[1], p.rvs[1]);
    const float fz = vx_floor_div(z - p.lo[2], p.vs[2], p.rvs[2]);
    const bool inside = (fx >= 0.f) & (fx < (float)p.grid[0]) & (fy >= 0.f) & (fy < (float)p.grid[1]) &
                        (fz >= 0.f) & (fz < (float)p.grid[2]);
    key = inside ? ((uint32_t)fz * (uint32_t)p.grid[1] + (uint32_t)fy) * (uint32_t)p.grid[0] + (uint32_t)fx : 0u;
    return inside;
}

// x / y cell only (z is not looked at): pillar index cy * nx + cx, false when outside in x or y
__device__ __forceinline__ bool vx_pillar(const VxParams &p, float x, float y, uint32_t &pillar) {
    const float fx = vx_floor_div(x - p.lo[0], p.vs[0], p.rvs[0]);
    const float fy = vx_floor_div(y - p.lo[1], p.vs[1], p.rvs[1]);
    const bool inside = (fx >= 0.f) & (fx < (float)p.grid[0]) & (fy >= 0.f) & (fy < (float)p.grid[1]);
    pillar = inside ? (uint32_t)fy * (uint32_t)p.grid[0] + (uint32_t)fx : 0u;
    return inside;
}

// Branch-free first look at one coordinate (phase A of the fused launch evaluates x and y of EVERY point of a frame in
// each of its bin workgroups, so instructions count): c = floor(d * fl(1/vs)), ok = 0 <= c < n, and risky = "an integer
// lies within the rounding margin of the quotient (or it is NaN / inf): the exact IEEE division must decide" — the same
// criterion as vx_floor_div with the margin in absolute form (eabs >= |q| * 4.8e-7 for every quotient inside the grid; a
// quotient far outside the grid is outside whatever its last bit).  Risky coordinates (~0.1 %) are re-evaluated with
// vx_pillar afterwards.
__device__ __forceinline__ void vx_cell_fast(float d, float rvs, float eabs, int n, int &c, bool &ok, bool &risky) {
    const float q = d * rvs;
    const float fl = floorf(q);
    const float fr = q - fl;                                     // in [0, 1], exact
    risky = !(fabsf(fr - 0.5f) <= 0.5f - eabs);                   // true for NaN as well
    c = (int)__builtin_amdgcn_fmed3f(fl, -1.0f, (float)n);        // clamped: the conversion is always defined
    ok = (unsigned)c < (unsigned)n;
}

// full cell: key as vx_cell, plus the pillar index the fused launch bins by
__device__ __forceinline__ bool vx_cell_pillar(const VxParams &p, float x, float y, float z, uint32_t &key, uint32_t &pillar) {
    const bool in_xy = vx_pillar(p, x, y, pillar);
    const float fz = vx_floor_div(z - p.lo[2], p.vs[2], p.rvs[2]);
    const bool inside = in_xy & (fz >= 0.f) & (fz < (float)p.grid[2]);
    key = inside ? (uint32_t)fz * (uint32_t)p.grid[1] * (uint32_t)p.grid[0] + pillar : 0u;
    return inside;
}

// K0: one workgroup per 1024-point tile: voxel key per point, bin = hash(key) % G, and an in-block
// partition of the tile's (point, key) pairs by bin (wave ballots + a 16 x G LDS count table; no
// atomics).  Outside-the-grid points get their per-point word (0) here and enter no bin.
#define VXL_GMAX 16
#define ITEMS_TILES(p) (((p).n_max + 1023) >> 10)
// Workgroups x < ntiles: one 1024-point tile each (cells, partition by bin).  Workgroups x >= ntiles: zero-fill role — the
// padded voxel rows are 96 % zeros and none of them depends on the index build, so the whole output buffer is cleared
// HERE, inside the first launch of the sequence (a plain 136 MB fill runs at 7.6 TB/s on this part, i.e. 19 us, and hides
// the key stage completely); the row stage then only scatters the occupied slots.  Running the fill beside the later,
// latency-critical bin / rank launches instead was measured and loses.
#define VXL_FILL_F4_PER_WG (1024 * 4)      // 64 KiB of zeros per fill workgroup
template <bool C4>
__global__ __launch_bounds__(1024) void vxl_key_kernel(const float *__restrict__ points,
                                                       const int *__restrict__ offsets, VxParams p, VxWs w, int G,
                                                       int ntiles, float *__restrict__ voxels, long long fill_f4_per_frame,
                                                       long long fill_tail_floats) {
    __shared__ int s_wc[16][VXL_GMAX];   // per wave, per bin counts -> exclusive offsets
    if ((int)blockIdx.x >= ntiles) {     // ---- zero-fill role (block-uniform)
        const long long c = (long long)blockIdx.x - ntiles;
        float4 *dst = reinterpret_cast<float4 *>(voxels) + (long long)blockIdx.y * fill_f4_per_frame;
        // compact layout: only the rows a call can plausibly produce are cleared here — the previous call's total + 25 % (kept
        // in the workspace, no host involvement); rows beyond that are written whole by their emit thread (rare)
        const long long lim_rows = p.compact ? (long long)w.fillst[0] : 0x7fffffffll;
        const long long lim_f4 = (lim_rows >= 0x7fffffffll) ? (1ll << 62) : (lim_rows * p.P * p.C + 3) / 4;
        const long long g0 = (long long)blockIdx.y * fill_f4_per_frame + c * VXL_FILL_F4_PER_WG;
        if (g0 >= lim_f4) return;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = c * VXL_FILL_F4_PER_WG + k * 1024 + threadIdx.x;
            if (i < fill_f4_per_frame && g0 + k * 1024 + threadIdx.x < lim_f4)
                dst[i] = z;                              // plain stores: non-temporal ones measured slower (49.6 vs 43.9 us)
        }
        if (blockIdx.y == gridDim.y - 1 && c == 0 && (long long)threadIdx.x < fill_tail_floats)   // bytes past the last float4
            voxels[(long long)gridDim.y * fill_f4_per_frame * 4 + threadIdx.x] = 0.f;
        return;
    }
    const int tile = blockIdx.x, f = blockIdx.y, t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int start = offsets[f];
    const int n = min(offsets[f + 1] - start, p.n_max);
    if (tile == 0 && t == 0) w.nvox[f] = 0;        // LDS path: first points of the frame, summed up by the bin workgroups
    const int j = tile * 1024 + t;
    const int jc = min(j, max(n - 1, 0));
    float x = 0.f, y = 0.f, z = 0.f;
    if (n > 0) {  // block-uniform: an empty frame has no row to read
        if (C4) {
            const float4 v = reinterpret_cast<const float4 *>(points)[(size_t)start + jc];
            x = v.x; y = v.y; z = v.z;
        } else {
            const float *q = points + ((size_t)start + jc) * p.C;
            x = q[0]; y = q[1]; z = q[2];
        }
    }
    uint32_t key;
    const bool inside = vx_cell(p, x, y, z, key) && (j < n);
    // every point's word starts at 0: outside points keep it, and so does a point dropped by a bin overflow (its word
    // must never be stale memory — the rank / row kernels index with it)
    if (j < n) w.flagw[(size_t)f * p.n_max + j] = 0;
    const uint32_t h2 = (key * 0x85EBCA6Bu) >> 16;
    const int bin = inside ? (int)((h2 * (uint32_t)G) >> 16) : -1;
    int myrank = 0;
    for (int b = 0; b < G; ++b) {
        const unsigned long long mm = __ballot(bin == b);
        if (bin == b) myrank = __popcll(mm & lanemask_lt());
        if (l == 0) s_wc[wv][b] = __popcll(mm);
    }
    __syncthreads();
    if (t < G) {  // exclusive scan over the 16 waves for bin t; total -> segment count
        int acc = 0;
        for (int k = 0; k < 16; ++k) {
            const int c = s_wc[k][t];
            s_wc[k][t] = acc;
            acc += c;
        }
        w.qcnt[((size_t)f * ntiles + tile) * G + t] = acc;
    }
    __syncthreads();
    if (bin >= 0)
        w.queue[(((size_t)f * ntiles + tile) * G + bin) * 1024 + s_wc[wv][bin] + myrank] = make_int2(j, (int)key);
}

// K1: workgroup (g, f) = bin g of frame f.
template <int ITEMS>
__global__ __launch_bounds__(1024) void vxl_bin_kernel(const int *__restrict__ offsets, VxParams p, VxWs w, int G) {
    __shared__ uint32_t s_key[VXL_S];   // phase B: keys; phase C..E: list offset of the slot
    __shared__ int s_first[VXL_S];
    __shared__ int s_cnt[VXL_S];
    __shared__ int2 s_q[VXL_CAP];       // phase B: (point, key); afterwards x = point | slot << 15, y = list cell
    __shared__ int s_wtot[16];
    __shared__ int s_tc[32];            // first points of this bin per 1024-point tile
    __shared__ int s_nent, s_total;
    const int g = blockIdx.x, f = blockIdx.y, t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int n = min(offsets[f + 1] - offsets[f], p.n_max);
    if (t < 32) s_tc[t] = 0;
    if (g == 0 && f == 0 && t == 0) w.fillst[1] = w.fillst[0];   // what the fill role of THIS call used; the emit launch reads it
    // ---- phase B1 (loads): my bin's (point, key) pairs from the ITEMS tile segments written by K0.
    // Counts and the first 256 entries of every segment are requested together (one memory round trip,
    // overlapped with the LDS initialisation below); longer segments are topped up afterwards.
    const int nt = (n + 1023) >> 10;
    const int *qc = w.qcnt + (size_t)f * ITEMS_TILES(p) * G;
    int cnt_u[ITEMS];
    int2 ent[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) cnt_u[u] = qc[min(u, max(nt - 1, 0)) * G + g];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int2 *seg = w.queue + (((size_t)f * ITEMS_TILES(p) + min(u, max(nt - 1, 0))) * G + g) * 1024;
        ent[u] = seg[min(t, 255)];
    }
    for (int k = t; k < VXL_S; k += 1024) {
        s_key[k] = VX_EMPTY;
        s_first[k] = VX_INF;
        s_cnt[k] = 0;
    }
    __syncthreads();
    {
        int base = 0;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int c = (u < nt) ? cnt_u[u] : 0;
            if (t < min(c, 256)) {
                if (base + t < VXL_CAP) s_q[base + t] = ent[u];
                else vx_raise(w, 2);
            }
            if (c > 256) {  // block-uniform, rare: a tile that sends more than a quarter of its points to one bin
                const int2 *seg = w.queue + (((size_t)f * ITEMS_TILES(p) + u) * G + g) * 1024;
                if (t >= 256 && t < c) {
                    if (base + t < VXL_CAP) s_q[base + t] = seg[t];
                    else vx_raise(w, 2);
                }
            }
            base += c;
        }
        if (t == 0) s_nent = base;
    }
    __syncthreads();
    const int ne = min(s_nent, VXL_CAP);
    // ---- phase B2: dense insertion into the LDS hash table (first point, count per voxel)
    for (int e = t; e < ne; e += 1024) {
        const int2 q = s_q[e];
        const int j = q.x;
        const uint32_t key = (uint32_t)q.y;
        uint32_t h = (key * 2654435761u) >> (32 - 13);  // log2(VXL_S) == 13
        int slot = -1;
        for (int probe = 0; probe < VXL_S; ++probe) {
            const uint32_t old = atomicCAS(&s_key[h], VX_EMPTY, key);
            if (old == VX_EMPTY || old == key) {
                slot = (int)h;
                break;
            }
            h = (h + 1u) & (VXL_S - 1);
        }
        if (slot >= 0) {
            atomicMin(&s_first[slot], j);
            atomicAdd(&s_cnt[slot], 1);
            s_q[e].x = j | (slot << 15);
        } else {
            vx_raise(w, 1);
            s_q[e].x = j | (int)0x80000000;
        }
    }
    __syncthreads();
    // ---- phase C: list offsets = exclusive scan of m = min(count, P) over the slots (8 per thread)
    int mloc[8];
    int run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mloc[k] = min(s_cnt[t * 8 + k], p.P);
        run += mloc[k];
    }
    const int inc = wave_incl_scan(run);
    if (l == 63) s_wtot[wv] = inc;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int k = 0; k < 16; ++k) {
            const int v = s_wtot[k];
            s_wtot[k] = acc;
            acc += v;
        }
        s_total = acc;
    }
    __syncthreads();
    int off = s_wtot[wv] + inc - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_key[t * 8 + k] = (uint32_t)off;
        off += mloc[k];
    }
    const int L = min(s_total, VXL_CAP);
    for (int k = t; k < L; k += 1024) s_q[k].y = VX_INF;
    __syncthreads();
    // ---- phase D: ordered lists (P smallest point indices per voxel, ascending) in LDS
    for (int e = t; e < ne; e += 1024) {
        const int en = s_q[e].x;
        if (en < 0) continue;
        const int j = en & 0x7FFF, slot = (en >> 15) & 0x1FFF;
        const int c = s_cnt[slot];
        int2 *Lp = s_q + s_key[slot];
        if (c == 1) {
            Lp[0].y = j;
        } else {
            const int m = min(c, p.P);
            int x = j;
            for (int s = 0; s < m; ++s) {
                const int old = atomicMin(&Lp[s].y, x);
                if (old == VX_INF) break;
                x = max(old, x);
            }
        }
    }
    __syncthreads();
    // ---- phase E: per-point word + this bin's packed lists
    int *pinfo = w.flagw + (size_t)f * p.n_max;
    for (int e = t; e < ne; e += 1024) {
        const int en = s_q[e].x;
        const int j = en & 0x7FFF;
        int word = 0;
        if (en >= 0) {
            const int slot = (en >> 15) & 0x1FFF;
            if (s_first[slot] == j) word = min(s_cnt[slot], p.P) | ((g * VXL_CAP + (int)s_key[slot]) << VXL_MBITS);
        }
        pinfo[j] = word;
        // per-tile first-point counts, aggregated per wave: the entries arrive tile by tile, so a wave sees 1-2 tile ids
        const int tau = j >> 10;
        unsigned long long rem = __ballot(word != 0);
        while (rem) {                                  // wave-uniform
            const int lead = __builtin_ctzll(rem);
            const int t0 = __shfl(tau, lead, 64);
            const unsigned long long m = __ballot(word != 0 && tau == t0);
            if (l == lead) atomicAdd(&s_tc[t0], __popcll(m));
            rem &= ~m;
        }
    }
    int *stg = w.pfirst + ((size_t)f * G + g) * VXL_CAP;   // staging lists live in the pfirst/vinfo region
    for (int k = t; k < L; k += 1024) stg[k] = s_q[k].y;
    __syncthreads();
    if (t < 32) w.tcnt[((size_t)f * G + g) * 32 + t] = s_tc[t];
    if (t == 0) {
        int tot = 0;
        for (int k = 0; k < 32; ++k) tot += s_tc[k];
        if (tot) atomicAdd(&w.nvox[f], tot);               // visible to the next launch; no ordering needed inside this one
    }
}

// Emit stage: workgroup (tile, f) owns the 1024 points of its tile.  A first point's voxel id = (first points of the frame in
// earlier tiles, from the bin kernel's per-tile counts) + (its ballot rank inside the tile) = first-appearance order; the
// thread then writes that voxel's row itself: occupied slots only (the buffer is already zero), coords from the first
// point's cell, count.  No separate ranking launch.
// rows (= voxels kept) of frame k: the 3-launch path sums them with atomics (nvox), the fused launch leaves one count per bin
__device__ __forceinline__ int vxl_frame_rows(const VxParams &p, const VxWs &w, int G, int k, int fused) {
    int c;
    if (fused) {
        c = 0;
        for (int g = 0; g < G; ++g) c += w.bvox[k * VXL_GMAX + g];
    } else {
        c = w.nvox[k];
    }
    return min(c, p.max_voxels);
}

template <bool C4>
__global__ __launch_bounds__(1024) void vxl_emit_kernel(const float *__restrict__ points, const int *__restrict__ offsets,
                                                        VxParams p, VxWs w, int G, float *__restrict__ voxels,
                                                        int *__restrict__ coords, int *__restrict__ num_points,
                                                        int *__restrict__ voxel_offsets, int fused, int resident) {
    __shared__ int s_part[16], s_wcnt[16];
    __shared__ int s_base;
    const int tile = blockIdx.x, f = blockIdx.y, t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int start = offsets[f];
    const int n = min(offsets[f + 1] - start, p.n_max);
    const int i = tile * 1024 + t;
    const int wd = w.flagw[(size_t)f * p.n_max + min(i, max(n - 1, 0))];      // unconditional load, masked below
    const int word = (i < n) ? wd : 0;
    // this thread's own point, requested together with its word: for a first point it IS slot 0 of the voxel's row, so the
    // 80 % of voxels that hold a single point need no dependent load at all
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (C4 && n > 0) me = reinterpret_cast<const float4 *>(points)[(size_t)start + min(i, n - 1)];   // n: block-uniform
    // first points of the frame in earlier tiles: sum of tcnt[f][g][tau < tile] over the G bins
    int v = 0;
    if (t < G * 32) {
        const int tau = t & 31;
        const int c = w.tcnt[(size_t)f * G * 32 + t];
        v = (tau < tile) ? c : 0;
    }
    v = wave_sum(v);
    if (l == 0) s_part[wv] = v;
    // rows of the earlier frames (every frame's count is complete: the bin launch has finished)
    if (wv == 15) {
        int part = 0;
        if (p.compact)
            for (int k0 = 0; k0 < f; k0 += 64) {
                const int k = k0 + l;
                part += (k < f) ? vxl_frame_rows(p, w, G, k, fused) : 0;
            }
        part = wave_sum(part);
        if (l == 0) s_base = p.compact ? part : f * p.max_voxels;
    }
    if (tile == 0 && f == 0 && wv == 14) {            // the batch's offsets table (off the critical path)
        int carry = 0;
        for (int k0 = 0; k0 < p.batch; k0 += 64) {
            const int k = k0 + l;
            const int c = (k < p.batch) ? vxl_frame_rows(p, w, G, k, fused) : 0;
            const int inc = wave_incl_scan(c);
            if (k < p.batch) voxel_offsets[k] = p.compact ? carry + inc - c : k * p.max_voxels;
            carry += __shfl(inc, 63, 64);
        }
        if (l == 0) {
            voxel_offsets[p.batch] = p.compact ? carry : p.batch * p.max_voxels;
            // rows the next call's fill role should clear up front (nobody reads fillst[0] during this launch)
            const long long next = (long long)carry + carry / 4 + 1024;
            w.fillst[0] = p.compact ? (int)min(next, (long long)p.batch * p.max_voxels) : 0x7fffffff;
            // resident-output history for the next call (a non-resident call leaves rows beyond its fill extent unspecified,
            // so it invalidates the history)
            w.resident[0] = resident ? (long long)reinterpret_cast<uintptr_t>(voxels) : 0ll;
            w.resident[1] = (long long)reinterpret_cast<uintptr_t>(num_points);
            w.resident[2] = carry;
            w.resident[3] = (long long)p.P * p.C;
        }
    }
    const long long cleared_rows = p.compact ? (long long)w.fillst[1] : 0x7fffffffll;
    const unsigned long long bal = __ballot(word != 0);
    if (l == 0) s_wcnt[wv] = __popcll(bal);
    __syncthreads();
    int r = 0;
    for (int k = 0; k < (G * 32 + 63) / 64 && k < 16; ++k) r += s_part[k];
    for (int k = 0; k < wv; ++k) r += s_wcnt[k];
    r += __popcll(bal & lanemask_lt());
    if (word == 0 || r >= p.max_voxels) return;       // not a first point / voxel beyond the cap (dropped with its points)
    const int cnt = word & VXL_MMASK;
    const int *lst = w.pfirst + (size_t)f * G * VXL_CAP + (word >> VXL_MBITS);
    const uint32_t nx = p.grid[0], ny = p.grid[1];
    const size_t row = (size_t)s_base + r;
    float x0 = 0.f, y0 = 0.f, z0 = 0.f;
    if (C4) {
        const float4 *pts4 = reinterpret_cast<const float4 *>(points) + start;
        float4 *out4 = reinterpret_cast<float4 *>(voxels) + row * p.P;
        out4[0] = me;                                  // slot 0 is this very point (the list is ascending, it is the first)
        if ((long long)row >= cleared_rows)            // beyond what the fill role cleared: this thread owns the whole row
            for (int sl = cnt; sl < p.P; ++sl) out4[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
        x0 = me.x; y0 = me.y; z0 = me.z;
        for (int s0 = 1; s0 < cnt; s0 += 4) {          // up to 4 independent gathers in flight
            int pi[4];
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pi[k] = lst[min(s0 + k, cnt - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = pts4[pi[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (s0 + k < cnt) out4[s0 + k] = q[k];
        }
    } else {
        const float *pts = points + (size_t)start * p.C;
        float *out = voxels + row * p.P * p.C;
        for (int sl = 0; sl < cnt; ++sl) {
            const float *q = pts + (size_t)lst[sl] * p.C;
            for (int c = 0; c < p.C; ++c) out[sl * p.C + c] = q[c];
            if (sl == 0) { x0 = q[0]; y0 = q[1]; z0 = q[2]; }
        }
        if ((long long)row >= cleared_rows)
            for (int e = cnt * p.C; e < p.P * p.C; ++e) out[e] = 0.f;
    }
    uint32_t key;
    vx_cell(p, x0, y0, z0, key);
    reinterpret_cast<int4 *>(coords)[row] = make_int4(f, (int)(key / (nx * ny)), (int)((key / nx) % ny), (int)(key % nx));
    num_points[row] = cnt;
}


// ================================================================== fused key + bin launch (algo 3)
// One launch instead of two, and no (point, key) queue in HBM (8 MB written + read per 16 frames): workgroup (g, f) reads
// ALL points of frame f itself (320 KB, L2 / Infinity-Cache hits for 7 of the 8 bins), evaluates their cells and keeps
// the ones whose key hashes to bin g, appended to the LDS entry list with one wave-aggregated LDS atomic per 64 points.
// Phases B2..E are those of vxl_bin_kernel.  Every workgroup of the launch reserves the bin role's 144 KB of LDS, so the
// launch is sized to ONE resident workgroup per CU: ids [0, nbinwg) are bin roles, the rest fill roles that clear the
// padded output with grid-stride stores while the bin roles run their LDS phases (bin roles join in when they are done).
// Degenerate input (thousands of points in one voxel, e.g. zero-padded clouds, where (0,0,0) lies inside the KITTI range):
// when a bin receives more than VXL_CAP entries the workgroup switches to a streaming variant without an entry list —
// pass 1 builds the table (first / count per voxel) straight from the points, pass 2 re-reads them for the insertion
// chains and the per-point words — which is exact for any multiplicity; only more than ~VXL_S distinct voxels or more
// than VXL_CAP list cells in ONE bin still raise the error flag.
#define VXL_RISK_CAP 512     // parked points per bin workgroup (expected ~20 per frame); more -> the streaming variant
#define VXL_A_U 20          // points per thread requested together in phase A (one round trip for a 20 k-point frame)
struct VxlShared {
    uint32_t *key;
    int *first, *cnt;
    int2 *q;
    int *wtot, *tc, *nent, *total;
};

__device__ __forceinline__ void vxl_fill_chunks(float4 *__restrict__ dst, long long c0, long long cstep, long long cend,
                                                long long lim_f4, int t) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long c = c0; c < cend; c += cstep) {
        const long long b = c * VXL_FILL_F4_PER_WG + t;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (b + k * 1024 < lim_f4) dst[b + k * 1024] = z;
    }
}

template <bool C4>
__device__ __forceinline__ void vxl_load_xyz(const float *__restrict__ points, size_t idx, int C, float &x, float &y, float &z) {
    if (C4) {
        const float4 v = reinterpret_cast<const float4 *>(points)[idx];
        x = v.x; y = v.y; z = v.z;
    } else {
        const float *q = points + idx * C;
        x = q[0]; y = q[1]; z = q[2];
    }
}

__device__ __forceinline__ int vxl_bin_of(uint32_t key, int G) {
    const uint32_t h2 = (key * 0x85EBCA6Bu) >> 16;
    return (int)((h2 * (uint32_t)G) >> 16);
}

// the fused launch's bin of a pillar: full-rate 24-bit multiplies only (v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ int vxl_bin_of24(uint32_t pillar, int G) {
    const uint32_t h = __umul24(pillar, 0x5BCA6Bu) ^ (pillar >> 9);
    return (int)(__umul24((h >> 8) & 0xFFFFu, (uint32_t)G) >> 16);
}

// LDS table insert: slot of `key` (claimed if new), or -1 when the table is full
__device__ __forceinline__ int vxl_table_insert(uint32_t *s_key, uint32_t key) {
    uint32_t h = (key * 2654435761u) >> (32 - 13);  // log2(VXL_S) == 13
    for (int probe = 0; probe < VXL_S; ++probe) {
        const uint32_t old = atomicCAS(&s_key[h], VX_EMPTY, key);
        if (old == VX_EMPTY || old == key) return (int)h;
        h = (h + 1u) & (VXL_S - 1);
    }
    return -1;
}

// exclusive scan of m = min(count, P) over the table slots -> list offsets (into sh.key, whose keys are no longer needed by
// the caller when `keep_keys` is false; the streaming variant passes a separate array); returns the total list length
__device__ __forceinline__ int vxl_list_offsets(const VxlShared &sh, int P, int t, uint32_t *off_out) {
    const int l = t & 63, wv = t >> 6;
    int mloc[8];
    int run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mloc[k] = min(sh.cnt[t * 8 + k], P);
        run += mloc[k];
    }
    const int inc = wave_incl_scan(run);
    if (l == 63) sh.wtot[wv] = inc;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int k = 0; k < 16; ++k) {
            const int v = sh.wtot[k];
            sh.wtot[k] = acc;
            acc += v;
        }
        *sh.total = acc;
    }
    __syncthreads();
    int off = sh.wtot[wv] + inc - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        off_out[t * 8 + k] = (uint32_t)off;
        off += mloc[k];
    }
    return *sh.total;
}

// order-independent insertion of point j into the ascending list of its voxel's m smallest point indices
__device__ __forceinline__ void vxl_chain_insert(int *cell0, int stride_ints, int m, int j) {
    int x = j;
    for (int s = 0; s < m; ++s) {
        const int old = atomicMin(cell0 + (size_t)s * stride_ints, x);
        if (old == VX_INF) break;
        x = max(old, x);
    }
}

template <bool C4>
__device__ __forceinline__ void vxl_bin_streaming(const float *__restrict__ points, const VxParams &p, const VxWs &w, int G,
                                               int g, int f, int start, int n, const VxlShared &sh) {
    const int t = threadIdx.x, l = t & 63;
    uint32_t *s_off = reinterpret_cast<uint32_t *>(sh.q);            // [VXL_S] list offset per slot
    int *s_list = reinterpret_cast<int *>(sh.q) + VXL_S;             // [VXL_CAP - VXL_S / 2 ... ] list cells
    const int LCAP = 2 * VXL_CAP - VXL_S;                            // ints left in the entry region (4096)
    const int nt = (n + 1023) >> 10;
    __syncthreads();                                                 // phase A's entry list is dropped: its LDS is re-used
    // ---- pass 1: first point and count per voxel, straight from the points
    for (int u = 0; u < nt; ++u) {
        const int j = u * 1024 + t;
        float x, y, z;
        vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x, y, z);
        uint32_t key, pil;
        const bool mine = vx_cell_pillar(p, x, y, z, key, pil) && j < n && vxl_bin_of24(pil, G) == g;
        if (mine) {
            const int slot = vxl_table_insert(sh.key, key);
            if (slot >= 0) {
                atomicMin(&sh.first[slot], j);
                atomicAdd(&sh.cnt[slot], 1);
            } else {
                vx_raise(w, 1);
            }
        }
    }
    __syncthreads();
    const int total = vxl_list_offsets(sh, p.P, t, s_off);
    if (total > min(LCAP, VXL_CAP) && t == 0) vx_raise(w, 2);
    const int L = min(total, min(LCAP, VXL_CAP));
    for (int k = t; k < L; k += 1024) s_list[k] = VX_INF;
    __syncthreads();
    // ---- pass 2: insertion chains, per-point words, per-tile first-point counts
    int *pinfo = w.flagw + (size_t)f * p.n_max;
    for (int u = 0; u < nt; ++u) {
        const int j = u * 1024 + t;
        float x, y, z;
        vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x, y, z);
        uint32_t key, pil;
        const bool in_xy = vx_pillar(p, x, y, pil) && j < n && vxl_bin_of24(pil, G) == g;     // z-outside points of my pillars
        const bool mine = vx_cell_pillar(p, x, y, z, key, pil) && in_xy;                       // get their word (0) from me too
        int word = 0;
        if (mine) {
            uint32_t h = (key * 2654435761u) >> (32 - 13);
            int slot = -1;
            for (int probe = 0; probe < VXL_S; ++probe) {
                const uint32_t k2 = sh.key[h];
                if (k2 == key) { slot = (int)h; break; }
                if (k2 == VX_EMPTY) break;
                h = (h + 1u) & (VXL_S - 1);
            }
            if (slot >= 0) {
                const int m = min(sh.cnt[slot], p.P), off = (int)s_off[slot];
                if (off + m <= L) {
                    if (sh.cnt[slot] == 1) s_list[off] = j;
                    else vxl_chain_insert(s_list + off, 1, m, j);
                    if (sh.first[slot] == j) word = m | ((g * VXL_CAP + off) << VXL_MBITS);
                }
            }
        }
        if (in_xy) pinfo[j] = word;
        const unsigned long long bal = __ballot(word != 0);
        if (l == 0 && bal) atomicAdd(&sh.tc[u], __popcll(bal));
    }
    __syncthreads();
    int *stg = w.pfirst + ((size_t)f * G + g) * VXL_CAP;
    for (int k = t; k < L; k += 1024) stg[k] = s_list[k];
}

template <bool C4>
__global__ __launch_bounds__(1024) void vxl_keybin_kernel(const float *__restrict__ points, const int *__restrict__ offsets,
                                                          VxParams p, VxWs w, int G, int nbinwg, int nfillwg,
                                                          float *__restrict__ voxels, long long total_f4,
                                                          long long tail_floats, int help16, int resident,
                                                          const int *__restrict__ prev_counts) {
    __shared__ uint32_t s_key[VXL_S];   // keys; after phase C: list offset of the slot
    __shared__ int s_first[VXL_S];
    __shared__ int s_cnt[VXL_S];
    __shared__ int2 s_q[VXL_CAP];       // (point, key); afterwards x = point | slot << 15, y = list cell
    __shared__ int s_wtot[16];
    __shared__ int s_tc[32];
    __shared__ int s_nent, s_total, s_nrisk;
    __shared__ float4 s_risk[VXL_RISK_CAP];   // points whose cell the exact division must decide: (x, y, z, index)
    const int id = blockIdx.x, t = threadIdx.x, l = t & 63;
    // ---- resident output (algo 4): the buffer still holds the previous call's result on an otherwise all-zero background, so
    // only the slots that call filled are re-zeroed (rows x count x 16 B instead of the whole padded buffer).  Valid only if the
    // history in the workspace names this very buffer / count array / row width; otherwise: clear everything, as a first call.
    const bool res_hist = resident && p.compact && w.resident[0] == (long long)reinterpret_cast<uintptr_t>(voxels) &&
                          w.resident[1] == (long long)reinterpret_cast<uintptr_t>(prev_counts) &&
                          w.resident[3] == (long long)p.P * p.C;
    // ---- fill geometry (all roles): chunks of 64 KiB; the last help16/16 of them belong to the bin roles
    const long long lim_rows = (p.compact && !resident) ? (long long)w.fillst[0] : 0x7fffffffll;
    const long long lim_f4 = (lim_rows >= 0x7fffffffll) ? total_f4 : min(total_f4, (lim_rows * p.P * p.C + 3) / 4);
    const long long nchunks = (lim_f4 + VXL_FILL_F4_PER_WG - 1) / VXL_FILL_F4_PER_WG;
    const long long nhelp = nchunks * help16 / 16, nmain = nchunks - nhelp;
    float4 *dst = reinterpret_cast<float4 *>(voxels);
    if (id >= nbinwg) {                  // ---- fill role (block-uniform)
        if (res_hist) {                  // re-zero what the previous call wrote: one thread per row, `count` slots each
            const long long prev_rows = w.resident[2];
            const int rowlen = p.P * p.C;
            for (long long r = (long long)(id - nbinwg) * 1024 + t; r < prev_rows; r += (long long)nfillwg * 1024) {
                const int c = min(prev_counts[r], p.P);
                if (C4) {
                    float4 *o = dst + r * p.P;
                    for (int sl = 0; sl < c; ++sl) o[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    float *o = voxels + r * rowlen;
                    for (int e = 0; e < c * p.C; ++e) o[e] = 0.f;
                }
            }
            return;
        }
        vxl_fill_chunks(dst, id - nbinwg, nfillwg, nmain, lim_f4, t);
        if (id == nbinwg && (long long)t < tail_floats) voxels[total_f4 * 4 + t] = 0.f;     // bytes past the last float4
        return;
    }
    // ---- bin role: id -> (g, f) with the G bins of a frame on ONE XCD (workgroup i runs on XCD i % 8): they share the
    // frame's points through that XCD's L2
    const int q8 = id >> 3;
    const int f = (id & 7) + 8 * (q8 / G), g = q8 % G;
    VxlShared sh;
    sh.key = s_key; sh.first = s_first; sh.cnt = s_cnt; sh.q = s_q;
    sh.wtot = s_wtot; sh.tc = s_tc; sh.nent = &s_nent; sh.total = &s_total;
#ifdef VXL_STAMPS   // -DVXL_STAMPS: shader-clock stamps of bin role 0's phases into the error page (tools/vx_phase_probe.py)
#define VXL_STAMP(k) do { if (id == 0 && t == 0) w.err[16 + (k)] = (int)clock64(); } while (0)
#else
#define VXL_STAMP(k) do { } while (0)
#endif
    VXL_STAMP(0);
    if (f < p.batch) {
        const int start = offsets[f];
        const int n = min(offsets[f + 1] - start, p.n_max);
        const int nt = (n + 1023) >> 10;
        // rows the fill roles of THIS call leave zero (the emit launch reads it): everything in resident mode
        if (id == 0 && t == 0) w.fillst[1] = resident ? 0x7fffffff : w.fillst[0];
        for (int k = t; k < VXL_S / 4; k += 1024) {
            reinterpret_cast<uint4 *>(s_key)[k] = make_uint4(VX_EMPTY, VX_EMPTY, VX_EMPTY, VX_EMPTY);
            reinterpret_cast<int4 *>(s_first)[k] = make_int4(VX_INF, VX_INF, VX_INF, VX_INF);
            reinterpret_cast<int4 *>(s_cnt)[k] = make_int4(0, 0, 0, 0);
        }
        if (t < 32) s_tc[t] = 0;
        if (t == 0) s_nent = s_nrisk = 0;
        __syncthreads();
        VXL_STAMP(1);
        // ---- phase A: cells of ALL points of the frame — the 8-fold redundant part of the fused design and VALU-bound, so
        // kept lean: branch-free fast cells (vx_cell_fast, 24-bit multiplies), ONE rarely taken wave-level branch per point
        // for the exact re-evaluation of risky coordinates (the operands are still in registers: no reload) and one for
        // points outside the grid.  Bins are by pillar, so a voxel's points meet in one bin.  A wave owns a contiguous run
        // of points (its entries come out ascending, which keeps the later per-entry stores of a wave close together); one
        // LDS atomic per wave for the whole frame.
        int *pinfo = w.flagw + (size_t)f * p.n_max;
        const int wv = t >> 6;
        const int wbase = wv * nt * 64;     // first point of this wave's run; point (u, l) = wbase + u * 64 + l
        const uint32_t nxy = (uint32_t)p.grid[0] * (uint32_t)p.grid[1];
        uint32_t bits = 0;                  // bit (u - u0): that point goes to my bin
        for (int u0 = 0; u0 < nt; u0 += VXL_A_U) {
            float x[VXL_A_U], y[VXL_A_U], z[VXL_A_U];
            uint32_t key[VXL_A_U];
#pragma unroll
            for (int u = 0; u < VXL_A_U; ++u) {
                const int j = wbase + (u0 + u) * 64 + l;
                vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x[u], y[u], z[u]);
            }
            bits = 0;
            int wtotal = 0;                 // wave-uniform: entries of this wave in this round
#pragma unroll
            for (int u = 0; u < VXL_A_U; ++u) {
                const int j = wbase + (u0 + u) * 64 + l;
                int cx, cy, cz;
                bool okx, oky, okz, rx, ry, rz;
                vx_cell_fast(x[u] - p.lo[0], p.rvs[0], p.eabs[0], p.grid[0], cx, okx, rx);
                vx_cell_fast(y[u] - p.lo[1], p.rvs[1], p.eabs[1], p.grid[1], cy, oky, ry);
                vx_cell_fast(z[u] - p.lo[2], p.rvs[2], p.eabs[2], p.grid[2], cz, okz, rz);
                const bool valid = (j < n) & (u0 + u < nt);
                const bool risky = (rx | ry | rz) & valid;
                const bool inside = okx & oky & okz;
                const uint32_t pil = __umul24((uint32_t)cy, (uint32_t)p.grid[0]) + (uint32_t)cx;   // grids up to 2^24 cells per layer
                key[u] = (uint32_t)cz * nxy + pil;
                if (__builtin_expect(__ballot(risky) != 0ull, 0)) {       // wave-uniform, ~1 in 20 wave-iterations
                    if (risky) {                                           // parked with its coordinates: settled after the barrier
                        const int k = atomicAdd(&s_nrisk, 1);
                        if (k < VXL_RISK_CAP) s_risk[k] = make_float4(x[u], y[u], z[u], __int_as_float(j));
                    }
                }
                if (__builtin_expect(__ballot(!inside & valid & !risky) != 0ull, 0)) {            // wave-uniform, rare
                    if (g == 0 && !inside && valid && !risky) pinfo[j] = 0;                       // a word of 0, written once
                }
                const bool mine = inside && valid && !risky && vxl_bin_of24(pil, G) == g;
                bits |= (uint32_t)mine << u;
                wtotal += __popcll(__ballot(mine));
            }
            int base = 0;
            if (l == 0 && wtotal) base = atomicAdd(&s_nent, wtotal);
            base = __shfl(base, 0, 64);
#pragma unroll
            for (int u = 0; u < VXL_A_U; ++u) {
                const bool mine = (bits >> u) & 1u;
                const unsigned long long bal = __ballot(mine);
                const int pos = base + __popcll(bal & lanemask_lt());
                if (mine && pos < VXL_CAP) s_q[pos] = make_int2(wbase + (u0 + u) * 64 + l, (int)key[u]);
                base += __popcll(bal);
            }
        }
        __syncthreads();
        // the parked points (~0.1 %): the reference's expression with the IEEE division, once per workgroup
        if (t < min(s_nrisk, VXL_RISK_CAP)) {
            const float4 v = s_risk[t];
            const int j = __float_as_int(v.w);
            uint32_t key2, pil2;
            if (vx_cell_pillar(p, v.x, v.y, v.z, key2, pil2)) {
                if (vxl_bin_of24(pil2, G) == g) {
                    const int pos = atomicAdd(&s_nent, 1);
                    if (pos < VXL_CAP) s_q[pos] = make_int2(j, (int)key2);
                }
            } else if (g == 0) {
                pinfo[j] = 0;
            }
        }
        __syncthreads();
        VXL_STAMP(2);
        if (__builtin_expect(s_nent > VXL_CAP || s_nrisk > VXL_RISK_CAP, 0)) {   // block-uniform, degenerate input only
            vxl_bin_streaming<C4>(points, p, w, G, g, f, start, n, sh);
        } else {
            const int ne = s_nent;
            // ---- phase B2: dense insertion into the LDS hash table (first point, count per voxel)
            for (int e = t; e < ne; e += 1024) {
                const int2 q = s_q[e];
                const int slot = vxl_table_insert(s_key, (uint32_t)q.y);
                int en = q.x | (int)0x80000000;                       // table full: dead entry, word 0
                if (slot >= 0) {
                    atomicMin(&s_first[slot], q.x);
                    atomicAdd(&s_cnt[slot], 1);
                    en = q.x | (slot << 15);
                } else {
                    vx_raise(w, 1);
                }
                s_q[e].x = en;
            }
            __syncthreads();
            VXL_STAMP(3);
            // ---- phase C: list offsets (the keys are not needed any more: offsets overwrite them)
            const int total = vxl_list_offsets(sh, p.P, t, s_key);
            const int L = min(total, VXL_CAP);       // total <= ne <= VXL_CAP
            for (int k = t; k < L; k += 1024) s_q[k].y = VX_INF;
            __syncthreads();
            VXL_STAMP(4);
            // ---- phase D: ordered lists in LDS
            for (int e = t; e < ne; e += 1024) {
                const int en = s_q[e].x;
                if (en < 0) continue;
                const int j = en & 0x7FFF, slot = (en >> 15) & 0x1FFF;
                const int c = s_cnt[slot];
                int2 *Lp = s_q + s_key[slot];
                if (c == 1) Lp[0].y = j;
                else vxl_chain_insert(&Lp[0].y, 2, min(c, p.P), j);
            }
            __syncthreads();
            VXL_STAMP(5);
            // ---- phase E: per-point word + this bin's packed lists
            for (int e = t; e < ne; e += 1024) {
                const int en = s_q[e].x;
                const int j = en & 0x7FFF;
                int word = 0;
                if (en >= 0) {
                    const int slot = (en >> 15) & 0x1FFF;
                    if (s_first[slot] == j) word = min(s_cnt[slot], p.P) | ((g * VXL_CAP + (int)s_key[slot]) << VXL_MBITS);
                }
                pinfo[j] = word;
                const int tau = j >> 10;
                unsigned long long rem = __ballot(word != 0);
                while (rem) {                                  // wave-uniform; a wave sees few distinct tile ids
                    const int lead = __builtin_ctzll(rem);
                    const int t0 = __shfl(tau, lead, 64);
                    const unsigned long long m = __ballot(word != 0 && tau == t0);
                    if (l == lead) atomicAdd(&s_tc[t0], __popcll(m));
                    rem &= ~m;
                }
            }
            int *stg = w.pfirst + ((size_t)f * G + g) * VXL_CAP;
            for (int k = t; k < L; k += 1024) stg[k] = s_q[k].y;
        }
        __syncthreads();
        VXL_STAMP(6);
        if (t < 32) w.tcnt[((size_t)f * G + g) * 32 + t] = s_tc[t];
        if (t == 0) {
            int tot = 0;
            for (int k = 0; k < 32; ++k) tot += s_tc[k];
            w.bvox[f * VXL_GMAX + g] = tot;                    // read by the emit launch
        }
    }
    VXL_STAMP(7);
    // ---- done with the index build: help with the tail of the fill
    if (!res_hist) vxl_fill_chunks(dst, nmain + id, nbinwg, nchunks, lim_f4, t);
}

static int vxl_bins(int n_max) { return divup(n_max, VXL_PTS_PER_BIN); }

static int vxl_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// fused key + bin + fill launch, then emit: 2 launches
static void vxl_run_fused(const float *points, const int *point_offsets, const VxParams &p, const VxWs &w, bool c4,
                          float *voxels, int *coords, int *num_points, int *voxel_offsets, hipStream_t s, int resident) {
    // share (in 1/16) of the fill left to the bin roles once they are done: 0 measured best (33.9 / 34.7 / 35.4 / 36.4 us
    // for 1 / 2 / 3 / 4 sixteenths) — the bin roles are the longer pole of the launch
    const int help16 = 0;
    const int G = vxl_bins(p.n_max);
    const int ntiles = divup(p.n_max, 1024);
    const long long total_floats = (long long)p.max_voxels * p.P * p.C * p.batch;
    const bool fill_f4 = (reinterpret_cast<uintptr_t>(voxels) & 15) == 0;
    const long long total_f4 = fill_f4 ? total_floats / 4 : 0;
    const long long tail_floats = total_floats - total_f4 * 4;
    const int nbinwg = 8 * G * divup(p.batch, 8);
    const int nfillwg = nbinwg < 192 ? 256 - nbinwg : 64;              // one resident workgroup per CU (LDS-bound)
    if (tail_floats > 1024) (void)hipMemsetAsync(voxels, 0, (size_t)total_floats * sizeof(float), s);   // unaligned buffer
    const long long tf = tail_floats <= 1024 ? tail_floats : 0;
    if (c4) hipLaunchKernelGGL(vxl_keybin_kernel<true>, dim3(nbinwg + nfillwg), dim3(1024), 0, s, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, num_points);
    else hipLaunchKernelGGL(vxl_keybin_kernel<false>, dim3(nbinwg + nfillwg), dim3(1024), 0, s, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, num_points);
    const dim3 ge(ntiles, p.batch);
    if (c4) hipLaunchKernelGGL(vxl_emit_kernel<true>, ge, dim3(1024), 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, 1, resident);
    else hipLaunchKernelGGL(vxl_emit_kernel<false>, ge, dim3(1024), 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, 1, resident);
}

static void vxl_run(const float *points, const int *point_offsets, const VxParams &p, const VxWs &w, bool c4,
                    float *voxels, int *coords, int *num_points, int *voxel_offsets, hipStream_t s) {
    const int G = vxl_bins(p.n_max);
    const int ntiles = divup(p.n_max, 1024);
    // zero-fill share of every frame index (positional split of the B * max_voxels rows, whatever the compaction)
    const long long frame_floats = (long long)p.max_voxels * p.P * p.C;
    const long long total_floats = frame_floats * p.batch;
    const bool fill_f4 = (reinterpret_cast<uintptr_t>(voxels) & 15) == 0;
    // float4 units per frame slice; a slice boundary need not be a row boundary (pure zeros), leftovers go to the tail
    const long long f4_per_frame = (fill_f4 && total_floats / 4 >= p.batch) ? (total_floats / 4) / p.batch : 0;
    const long long tail_floats = total_floats - f4_per_frame * 4 * p.batch;      // < 4 * batch + 4 (or everything if unaligned)
    const int nfill = (int)divup(f4_per_frame, VXL_FILL_F4_PER_WG);
    const dim3 gk(ntiles + nfill, p.batch);
    if (c4) hipLaunchKernelGGL(vxl_key_kernel<true>, gk, dim3(1024), 0, s, points, point_offsets, p, w, G, ntiles, voxels, f4_per_frame, tail_floats <= 1024 ? tail_floats : 0);
    else hipLaunchKernelGGL(vxl_key_kernel<false>, gk, dim3(1024), 0, s, points, point_offsets, p, w, G, ntiles, voxels, f4_per_frame, tail_floats <= 1024 ? tail_floats : 0);
    if (tail_floats > 1024)   // unaligned output buffer (never the case for torch allocations): plain memset of everything
        (void)hipMemsetAsync(voxels, 0, (size_t)total_floats * sizeof(float), s);
    const int items = divup(p.n_max, 1024);
    const dim3 gb(G, p.batch);
    if (items <= 4) hipLaunchKernelGGL(vxl_bin_kernel<4>, gb, dim3(1024), 0, s, point_offsets, p, w, G);
    else if (items <= 8) hipLaunchKernelGGL(vxl_bin_kernel<8>, gb, dim3(1024), 0, s, point_offsets, p, w, G);
    else if (items <= 16) hipLaunchKernelGGL(vxl_bin_kernel<16>, gb, dim3(1024), 0, s, point_offsets, p, w, G);
    else if (items <= 24) hipLaunchKernelGGL(vxl_bin_kernel<24>, gb, dim3(1024), 0, s, point_offsets, p, w, G);
    else hipLaunchKernelGGL(vxl_bin_kernel<32>, gb, dim3(1024), 0, s, point_offsets, p, w, G);
    const dim3 ge(ntiles, p.batch);
    if (c4) hipLaunchKernelGGL(vxl_emit_kernel<true>, ge, dim3(1024), 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, 0, 0);
    else hipLaunchKernelGGL(vxl_emit_kernel<false>, ge, dim3(1024), 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, 0, 0);
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
    hipLaunchKernelGGL(vx_ws_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, nh, nl, 0LL);
    return lidar_check_launch("vx_ws_init");
}

// Registers a device-visible host address (pinned + mapped, e.g. a pinned torch tensor's data_ptr) that receives the
// error bits as well: the host can then poll the flag at no cost (no copy, no synchronisation).  nullptr unregisters.
__global__ void vx_set_mirror_kernel(VxWs w, int *host_flag) { *w.mirror = host_flag; }

LIDAR_EXPORT int lidar_voxelize_set_error_mirror(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels,
                                                 int *host_flag, void *stream) {
    VxWs w;
    if (n_max <= 0) n_max = 1;
    if (!ws || vx_carve(ws, batch, n_max, max_voxels, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    hipLaunchKernelGGL(vx_set_mirror_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, host_flag);
    return lidar_check_launch("vx_set_mirror");
}

// sticky overflow flag of the LDS-binned path (0 = fine).  Host-synchronous: call outside captures.
LIDAR_EXPORT int lidar_voxelize_error_flag(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels) {
    VxWs w;
    if (n_max <= 0) n_max = 1;
    if (!ws || vx_carve(ws, batch, n_max, max_voxels, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    int v = 0;
    if (hipMemcpy(&v, w.err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return LIDAR_ERR_LAUNCH;
    return v;
}

LIDAR_EXPORT int lidar_voxelize(const float *points, const int *point_offsets, int batch, int n_max,
                                int num_features, const float *range6, const float *voxel_size3,
                                const int *grid3, int max_points, int max_voxels, int compact, int algo, float *voxels,
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
        p.rvs[j] = 1.0f / voxel_size3[j];
        p.eabs[j] = fminf(((float)grid3[j] + 2.0f) * 4.9e-7f, 0.5f);
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
    // algo 0 = auto, 1 = LDS-binned 3 launches, 3 = LDS-binned fused 2 launches, 4 = 3 with a resident output buffer
    // (1 / 3 / 4: n_max <= 32768, max_points < 16384), 2 = global hash table (any size)
    const bool lds_ok = (n_max <= VXL_MAX_ITEMS * 1024) && (max_points <= VXL_MMASK);
    if ((algo == 1 || algo == 3 || algo == 4) && !lds_ok) return LIDAR_ERR_ARG;
    static const int auto_algo = vxl_env_int("LIDAR_VXL_ALGO", 3);
    if (algo == 0 && lds_ok) algo = (auto_algo == 1) ? 1 : 3;
    if (algo == 3 || algo == 4) {
        const bool unaligned = ((reinterpret_cast<uintptr_t>(voxels) & 15) != 0);      // the resident clear wants 16-B rows
        vxl_run_fused(points, point_offsets, p, w, c4, voxels, coords, num_points, voxel_offsets, s,
                      (algo == 4 && compact && !unaligned) ? 1 : 0);
        return lidar_check_launch("lidar_voxelize(fused)");
    }
    if (algo == 1) {
        vxl_run(points, point_offsets, p, w, c4, voxels, coords, num_points, voxel_offsets, s);
        return lidar_check_launch("lidar_voxelize(lds)");
    }
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
