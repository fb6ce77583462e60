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
#ifndef VXL_FILL_SHARE16                 // sixteenths of the predicted rows the first launch's fill roles clear; the emit launch clears the rest
#define VXL_FILL_SHARE16 16
#endif
#ifndef VXL_FILL_NT                      // A/B: 1 = the zero fill as streaming (nt) stores
#define VXL_FILL_NT 0
#endif
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>

#define VX_EMPTY 0xFFFFFFFFu
#define VX_INF 0x7FFFFFFF
#define VX_TILE 1024
#define VX_ROWS_PER_BLOCK 64
// LDS-binned path (algo 3 / 4): a frame's points are hash-partitioned by PILLAR into G ~ n_max / VXL_PTS_PER_BIN bins (a power
// of two between 8 and 32), one
// workgroup per (bin, frame); sizes below are per bin workgroup (74 KB of LDS: two workgroups per CU)
#ifndef VXL_PTS_PER_BIN
#define VXL_PTS_PER_BIN 1280   // expected points per bin
#endif
#define VXL_SBITS 12
#define VXL_S (1 << VXL_SBITS)  // LDS table slots per bin
#define VXL_CAP 3072            // LDS entry / list-cell capacity per bin
#define VXL_GMAX 32             // bins per frame (n_max <= 32768)

typedef float vx_f4 __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-B store: the padded voxel rows are written once and not re-read here
__device__ __forceinline__ void vx_store_nt(float4 *dst, float4 v) {
    vx_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vx_f4 *>(dst));
}

#define VX_HOFF_MAX 64    // frames whose offsets travel as kernel arguments (lidar_voxelize_hostoff)

struct VxParams {
    float lo[3];
    float vs[3];
    float rvs[3];  // fl(1 / vs): only used to skip the IEEE division when the quotient is far from an integer
    float eabs[3]; // fused launch: absolute form of the same margin, (grid + 2) * 4.8e-7 (valid for every in-grid quotient)
    int grid[3];  // nx, ny, nz
    int C, P, max_voxels, batch, n_max, compact;
    int H, hshift, ntiles;
    int hoff_n;           // > 0: hoff[0 .. batch] holds the frame offsets (the caller knew them on the host): the LDS-binned launches
    int hoff[VX_HOFF_MAX + 1];   // take them from the kernel arguments instead of a dependent (cold, ~1.5 us) load in front of the first point read
};

struct VxWs {
    // ---- LDS-binned path (algo 3 / 4)
    int *flagw;      // [B][n_max] one word per point: 0, or for a voxel's first point min(count, P) | list position << 14
    int *stgi;       // [B][G][VXL_CAP] every bin's packed point-index lists (read by the emit launch when C != 4)
    float4 *stg4;    // [B][G][VXL_CAP] C == 4: the POINTS of the list cells beyond each list's head, same positions
    int *err;        // [1] sticky error flag (LDS table / entry list overflow)
    int *tcnt;       // [B][G][32] first points of bin g per 1024-point tile (the emit stage turns them into voxel ids)
    uint32_t *pkey;  // [B][align1024(n_max)] the frame's shared keys (vx_key2): VX_NOKEY between calls (reset by the emit launch)
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

__device__ __forceinline__ int vx_offset(const VxParams &p, const int *__restrict__ offsets, int f) {
    return p.hoff_n ? p.hoff[f] : offsets[f];
}

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

static int vxl_bins(int n_max) {              // bins per frame: the power of two >= n_max / VXL_PTS_PER_BIN, 8 .. VXL_GMAX
    int g = 8;
    while (g < VXL_GMAX && g * VXL_PTS_PER_BIN < n_max) g <<= 1;
    return g;
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
    const size_t G = (size_t)vxl_bins(n_max);
    p = take((size_t)B * n_max * 4); if (w) w->flagw = (int *)p;
    p = take((size_t)B * G * VXL_CAP * 4); if (w) w->stgi = (int *)p;
    p = take((size_t)B * G * VXL_CAP * 16); if (w) w->stg4 = (float4 *)p;
    p = take(65536); if (w) w->err = (int *)p;  // [0] sticky error flag (+ debug stamps)
    p = take((size_t)B * G * 32 * 4 + 256); if (w) w->tcnt = (int *)p;                               // [B][G][32]
    p = take((size_t)B * ((n_max + 1023) & ~1023) * 4 + 1280 * 4 + 256); if (w) w->pkey = (uint32_t *)p;   // (+ one phase-A1 round of slack)
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
    p = take((size_t)B * VXL_GMAX * 4 + 256); if (w) w->bvox = (int *)p;
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
    for (long long k = i; k < nq; k += stride) w.pkey[k] = 0xFFFFFFFFu;       // VX_NOKEY
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




// ================================================================== LDS-binned path (algo 3 / 4): 2 launches
// No global atomics on the data path.  A frame's points are hash-partitioned by PILLAR (x / y cell) into G bins; workgroup
// (g, f) of the first launch resolves everything that is local to a voxel inside LDS with LDS atomics: point count and the
// ascending list of its first P point indices (order-independent atomicMin insertion chain, whose head is the voxel's first
// point).  It leaves one 32-bit word per point (0, or for a voxel's first point: count | list position) and the bin's packed
// lists.  The second launch ranks the first points (= voxel ids in first-appearance order) and writes the rows.
#define VXL_MBITS 14        // per-point word: m = min(count, P) in the low 14 bits, list position above
#define VXL_MMASK ((1 << VXL_MBITS) - 1)
#define VXL_MAX_ITEMS 32    // n_max <= 32 * 1024 (a point index fits 15 bits)
#define VXL_A_KEYS 1280     // keys per wave and round in phase A1: 20 per lane, five 16-byte sc1 loads in flight on one address register
#define VXL_FILL_F4_PER_WG (1024 * 4)      // 64 KiB of zeros per fill chunk
#ifndef VXL_WAIT_TICKS                     // (-DVXL_WAIT_TICKS=0: the test build whose every bin workgroup takes the deadline exit)
#define VXL_WAIT_TICKS 20000               // 200 us of the 100 MHz wall clock: how long a wave re-reads a key that has not arrived
#endif
#define VX_NOKEY 0xFFFFFFFFu               // a shared key that has not been published (never a valid key: vx_key2)

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

// The 32-bit key the LDS path works with: cz << 24 | pillar (pillar = cy * nx + cx < 2^24, cz < 255) for a point inside the
// grid — its low 24 bits choose the bin, so a voxel's points meet in one bin — and 0xFF000000 | (j & 0xFFFFFF) for point j
// outside the grid (NaN included): such points spread over the bins by their index and are dead entries there (word 0).
__device__ __forceinline__ uint32_t vx_key2(const VxParams &p, float x, float y, float z, int j) {
    const float fx = vx_floor_div(x - p.lo[0], p.vs[0], p.rvs[0]);
    const float fy = vx_floor_div(y - p.lo[1], p.vs[1], p.rvs[1]);
    const float fz = vx_floor_div(z - p.lo[2], p.vs[2], p.rvs[2]);
    const bool inside = (fx >= 0.f) & (fx < (float)p.grid[0]) & (fy >= 0.f) & (fy < (float)p.grid[1]) &
                        (fz >= 0.f) & (fz < (float)p.grid[2]);
    return inside ? ((uint32_t)fz << 24) | ((uint32_t)fy * (uint32_t)p.grid[0] + (uint32_t)fx) : 0xFF000000u | ((uint32_t)j & 0xFFFFFFu);
}

// rows (= voxels kept) of frame k: one count per bin, left by the bin workgroups with plain stores
__device__ __forceinline__ int vxl_frame_rows(const VxParams &p, const VxWs &w, int G, int k) {
    // G is 8, 16 or 32: all of a frame's counts are requested together, as 16-byte loads (a scalar loop over G entries is G
    // dependent round trips — the emit workgroups' first barrier waited 3 k cycles for the wave that sums the earlier frames)
    const int4 *q = reinterpret_cast<const int4 *>(w.bvox + k * VXL_GMAX);
    int4 v[VXL_GMAX / 4];
#pragma unroll
    for (int i = 0; i < VXL_GMAX / 4; ++i) v[i] = (i * 4 < G) ? q[i] : make_int4(0, 0, 0, 0);
    int c = 0;
#pragma unroll
    for (int i = 0; i < VXL_GMAX / 4; ++i) c += v[i].x + v[i].y + v[i].z + v[i].w;
    return min(c, p.max_voxels);
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

__device__ __forceinline__ uint32_t vx_mul_u24(uint32_t a, uint32_t b) {   // low 32 bits of (a & 0xFFFFFF) * (b & 0xFFFFFF)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b), "v"(a));               // (the compiler picks the quarter-rate v_mul_lo_u32 here)
    return r;
}

// bin of a key: the top log2(G) bits of a 24-bit multiplicative hash of its low 24 bits (full-rate v_mul_u32_u24, which
// ignores the top byte by itself); gsh = 32 - log2(G)
__device__ __forceinline__ int vxl_bin_of24(uint32_t key, int gsh) {
    return (int)(vx_mul_u24(key, 0x5BCA6Bu) >> gsh);
}

// LDS table insert: slot of `key` (claimed if new), or -1 when the table is full
__device__ __forceinline__ int vxl_table_insert(uint32_t *s_key, uint32_t key) {
    uint32_t h = (key * 2654435761u) >> (32 - VXL_SBITS);
    for (int probe = 0; probe < VXL_S; ++probe) {
        const uint32_t old = atomicCAS(&s_key[h], VX_EMPTY, key);
        if (old == VX_EMPTY || old == key) return (int)h;
        h = (h + 1u) & (VXL_S - 1);
    }
    return -1;
}

__device__ __forceinline__ int vxl_table_find(const uint32_t *s_key, uint32_t key) {
    uint32_t h = (key * 2654435761u) >> (32 - VXL_SBITS);
    for (int probe = 0; probe < VXL_S; ++probe) {
        const uint32_t k2 = s_key[h];
        if (k2 == key) return (int)h;
        if (k2 == VX_EMPTY) return -1;
        h = (h + 1u) & (VXL_S - 1);
    }
    return -1;
}

// exclusive scan of m = min(count, P) over the VXL_S table slots (4 per thread) -> list offset per slot in off[4];
// returns the total list length.  Two barriers inside.
__device__ __forceinline__ int vxl_list_offsets(const int *s_cnt, int *s_wtot, int *s_total, int P, int t, int (&m)[VXL_S / 1024],
                                                int (&off)[VXL_S / 1024]) {
    const int l = t & 63, wv = t >> 6;
    int run = 0;
#pragma unroll
    for (int k = 0; k < VXL_S / 1024; ++k) {
        m[k] = min(s_cnt[t * (VXL_S / 1024) + k], P);
        run += m[k];
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
        *s_total = acc;
    }
    __syncthreads();
    int o = s_wtot[wv] + inc - run;
#pragma unroll
    for (int k = 0; k < VXL_S / 1024; ++k) {
        off[k] = o;
        o += m[k];
    }
    return *s_total;
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

// rows of a compact buffer the first launch clears: a share of the prediction (the rest is cleared by the emit launch, whose first-point
// threads know which rows exist); "everything" stays everything
__device__ __forceinline__ long long vxl_fill_share(long long rows) {
    return (rows >= 0x7fffffffll || VXL_FILL_SHARE16 >= 16) ? rows : rows * VXL_FILL_SHARE16 / 16;
}

__device__ __forceinline__ void vxl_fill_chunks(float4 *__restrict__ dst, long long c0, long long cstep, long long cend,
                                                long long lim_f4, int t) {
    typedef float vxf4 __attribute__((ext_vector_type(4)));
    const vxf4 z = {0.f, 0.f, 0.f, 0.f};
    vxf4 *d = reinterpret_cast<vxf4 *>(dst);
    for (long long c = c0; c < cend; c += cstep) {
        const long long b = c * VXL_FILL_F4_PER_WG + t;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (b + k * 1024 < lim_f4) {
#if VXL_FILL_NT
                __builtin_nontemporal_store(z, d + b + k * 1024);
#else
                d[b + k * 1024] = z;
#endif
            }
    }
}

// per-tile first-point counts of one wave's entries (the entries arrive in ascending runs, so a wave sees 1-2 tile ids)
__device__ __forceinline__ void vxl_count_firsts(int *s_tc, bool isfirst, int j, int l) {
    const int tau = j >> 10;
    unsigned long long rem = __ballot(isfirst);
    while (rem) {                                  // wave-uniform
        const int lead = __builtin_ctzll(rem);
        const int t0 = __shfl(tau, lead, 64);
        const unsigned long long m = __ballot(isfirst && tau == t0);
        if (l == lead) atomicAdd(&s_tc[t0], __popcll(m));
        rem &= ~m;
    }
}

// 20 keys of one lane — uint4 number r * 64 + lane of the wave's run, r = 0..4 — with five L1-bypassing (`sc1`) 16-byte loads issued
// together and waited for once.  Hand-written: __hip_atomic_load has no 16-byte form, the polling loop must re-issue the loads every
// time, and the saddr + immediate-offset form needs ONE address VGPR for all five (ten dword loads with their own 64-bit addresses
// were the register limit of this kernel, and two dependent rounds of them cost 2.5 us each).
typedef unsigned vx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vxl_load_keys20(const uint32_t *base, uint32_t voff, vx_u4 (&k)[5]) {
    const uint32_t vmid = voff + 2048u;                   // immediate offsets are 13-bit signed
    asm volatile(
        "global_load_dwordx4 %0, %5, %6 offset:-2048 sc1\n"
        "global_load_dwordx4 %1, %5, %6 offset:-1024 sc1\n"
        "global_load_dwordx4 %2, %5, %6 sc1\n"
        "global_load_dwordx4 %3, %5, %6 offset:1024 sc1\n"
        "global_load_dwordx4 %4, %5, %6 offset:2048 sc1\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]), "=&v"(k[4])
        : "v"(vmid), "s"(base)
        : "memory");
}

struct VxlShared {
    uint32_t *key;
    int *cnt, *aux;
    int2 *q;
    int *wtot, *tc, *total;
};

// Degenerate input (thousands of points in one voxel, e.g. zero-padded clouds, where (0,0,0) lies inside the KITTI range):
// a bin that receives more than VXL_CAP entries runs WITHOUT an entry list — pass 1 builds the table (first point / count
// per voxel) straight from the points, pass 2 re-reads them for the insertion chains and the per-point words — which is
// exact for any multiplicity; only more than ~VXL_S distinct voxels or more than VXL_CAP list cells in ONE bin still raise
// the error flag.  Works from the points alone (no shared keys).
template <bool C4>
__device__ __forceinline__ void vxl_bin_streaming(const float *__restrict__ points, const VxParams &p, const VxWs &w, int G,
                                               int g, int f, int start, int n, const VxlShared &sh) {
    const int t = threadIdx.x, l = t & 63;
    const int gsh = 32 - (__ffs(G) - 1);
    int *s_first = sh.aux;                                           // [VXL_S] first point per slot
    int *s_list = reinterpret_cast<int *>(sh.q);                     // list cells (the entry list is dropped)
    const int LCAP = VXL_CAP;                                        // list positions index the [G][VXL_CAP] staging arrays
    const int nt = (n + 1023) >> 10;
    for (int k = t; k < VXL_S; k += 1024) s_first[k] = VX_INF;
    __syncthreads();
    // ---- pass 1: first point and count per voxel, straight from the points
    for (int u = 0; u < nt; ++u) {
        const int j = u * 1024 + t;
        float x, y, z;
        vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x, y, z);
        const uint32_t key = vx_key2(p, x, y, z, j);
        if (j < n && (key >> 24) != 0xFFu && vxl_bin_of24(key, gsh) == g) {
            const int slot = vxl_table_insert(sh.key, key);
            if (slot >= 0) {
                atomicMin(&s_first[slot], j);
                atomicAdd(&sh.cnt[slot], 1);
            } else {
                vx_raise(w, 1);
            }
        }
    }
    __syncthreads();
    int m[VXL_S / 1024], off[VXL_S / 1024];
    const int total = vxl_list_offsets(sh.cnt, sh.wtot, sh.total, p.P, t, m, off);
    if (total > LCAP && t == 0) vx_raise(w, 2);
    const int L = min(total, LCAP);
#pragma unroll
    for (int k = 0; k < VXL_S / 1024; ++k) sh.cnt[t * (VXL_S / 1024) + k] = (off[k] & 8191) | (m[k] << 13);   // m < 16384
    for (int k = t; k < L; k += 1024) s_list[k] = VX_INF;
    __syncthreads();
    // ---- pass 2: insertion chains, per-point words, per-tile first-point counts
    int *pinfo = w.flagw + (size_t)f * p.n_max;
    for (int u = 0; u < nt; ++u) {
        const int j = u * 1024 + t;
        float x, y, z;
        vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x, y, z);
        const uint32_t key = vx_key2(p, x, y, z, j);
        const bool my_bin = j < n && vxl_bin_of24(key, gsh) == g;     // points outside the grid get their word (0) from their bin too
        int word = 0;
        if (my_bin && (key >> 24) != 0xFFu) {
            const int slot = vxl_table_find(sh.key, key);
            if (slot >= 0) {
                const int pk = sh.cnt[slot];
                const int mm = pk >> 13, o = pk & 8191;
                if (o + mm <= L) {
                    vxl_chain_insert(s_list + o, 1, mm, j);
                    if (s_first[slot] == j) word = mm | ((g * VXL_CAP + o) << VXL_MBITS);
                }
            }
        }
        if (my_bin) pinfo[j] = word;
        const unsigned long long bal = __ballot(word != 0);
        if (l == 0 && bal) atomicAdd(&sh.tc[u], __popcll(bal));
    }
    __syncthreads();
    int *stgi = w.stgi + ((size_t)f * G + g) * VXL_CAP;
    float4 *stg4 = w.stg4 + ((size_t)f * G + g) * VXL_CAP;
    for (int k = t; k < L; k += 1024) {
        const int j = s_list[k];
        stgi[k] = j;
        if (C4 && j != VX_INF) stg4[k] = reinterpret_cast<const float4 *>(points)[(size_t)start + j];   // heads included: harmless
    }
}

// First launch.  Workgroups [0, nbinwg): bin roles; the rest: fill roles, which clear the padded output (96 % of its bytes
// are zeros and none depends on the index build) while the bin roles run — 72 KB of LDS and <= 64 VGPRs per workgroup, so a
// fill workgroup is resident beside every bin workgroup on all 256 CUs.
//   bin role (g, f):
//   A0  SHARED KEYS.  Every bin workgroup of a frame needs the cell of EVERY point of the frame (to pick its own); each
//       reading and evaluating all the points was measured L2-bound (16 workgroups x 320 KB per frame through one XCD's L2:
//       9 us).  Instead workgroup g evaluates the exact cell of 1 / G of the frame's points and publishes 32-bit keys
//       (vx_key2) in global memory with plain stores; the others read them with L1-bypassing (sc1) loads.  There is NO flag
//       and no ordering requirement: a key word is VX_NOKEY from the end of the previous call (the emit launch resets it,
//       kernel boundaries make that visible everywhere) until its producer's aligned 4-byte store lands, so a reader that
//       sees VX_NOKEY just reads again — the data is its own flag.  The G bin workgroups of a frame are dealt to ONE XCD
//       (ids f, f + 8, ...; observed placement, not a contract), where the store is visible in the shared L2 within ~1 us.
//       Nothing depends on that placement for correctness: a reader on another XCD, or one whose producers are not resident
//       beside it (another process on the GPU), keeps reading VX_NOKEY until its deadline (VXL_WAIT_TICKS) and then
//       builds its bin from the points themselves (the streaming variant) — same result, no deadlock, every wave exits.
//   A1  the frame's keys (4 B per point instead of a 16-B point): bin = hash(low 24 bits); the (index, key) pairs of MY
//       points are appended to the LDS entry list (one LDS atomic per wave and round).
//   B   LDS hash table insert, count per voxel (keys of points outside the grid are dead entries).
//   C   list space per voxel (wave-aggregated allocation), list heads marked in a bitmap
//   D   ordered lists (atomicMin insertion chains, whose head is the voxel's first point)
//   E   per-point words, per-tile first-point counts, staging of the lists (C == 4: of the points themselves for the cells
//       beyond each head, so the emit launch needs no dependent gather)
template <bool C4>
__global__ __launch_bounds__(1024, 8) void vxl_keybin_kernel(const float *__restrict__ points, const int *__restrict__ offsets,
                                                             VxParams p, VxWs w, int G, int nbinwg, int nfillwg,
                                                             float *__restrict__ voxels, long long total_f4,
                                                             long long tail_floats, int help16, int resident,
                                                             const int *__restrict__ prev_counts) {
    __shared__ uint32_t s_key[VXL_S];   // keys; after phase C: list offset of the slot
    __shared__ int s_cnt[VXL_S];
    __shared__ int s_aux[VXL_S];        // streaming variant only: first point per slot
    __shared__ int2 s_q[VXL_CAP];       // x: entry (point, then point | slot << 15), y: key, then list cell
    __shared__ uint32_t s_head[VXL_CAP / 32];
    __shared__ int s_wtot[16];
    __shared__ int s_tc[32];
    __shared__ int s_nent, s_total, s_slow;
    const int id = blockIdx.x, t = threadIdx.x, l = t & 63;
#ifdef VXL_STAMPS   // -DVXL_STAMPS (tools/vx_phase_probe.py): shader-clock stamps of bin role 0's phases and the 100 MHz wall
                    // clock at the start / end of EVERY workgroup of both launches, into the error page
#define VXL_STAMP(k) do { if (id == 0 && t == 0) w.err[16 + (k)] = (int)clock64(); } while (0)
#define VXL_WALL(slot) do { if (t == 0) w.err[64 + (slot)] = (int)wall_clock64(); } while (0)
#define VXL_ESTAMP(k) do { if (f == p.batch - 1 && tile == 5 && t == 0) w.err[32 + (k)] = (int)clock64(); } while (0)   // one emit workgroup's phases
#else
#define VXL_ESTAMP(k) do { } while (0)
#define VXL_STAMP(k) do { } while (0)
#define VXL_WALL(slot) do { } while (0)
#endif
    VXL_WALL(2 * id);
    // ---- resident output (algo 4): the buffer still holds the previous call's result on an otherwise all-zero background, so
    // only the slots that call filled are re-zeroed (rows x count x 16 B instead of the whole padded buffer).  Valid only if the
    // history in the workspace names this very buffer / count array / row width; otherwise: clear everything, as a first call.
    const bool res_hist = resident && p.compact && w.resident[0] == (long long)reinterpret_cast<uintptr_t>(voxels) &&
                          w.resident[1] == (long long)reinterpret_cast<uintptr_t>(prev_counts) &&
                          w.resident[3] == (long long)p.P * p.C;
    // ---- fill geometry (all roles): chunks of 64 KiB; the last help16/16 of them belong to the bin roles
    const long long lim_rows = (p.compact && !resident) ? vxl_fill_share((long long)w.fillst[0]) : 0x7fffffffll;
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
            VXL_WALL(2 * id + 1);
            return;
        }
        vxl_fill_chunks(dst, id - nbinwg, nfillwg, nmain, lim_f4, t);
        if (id == nbinwg && (long long)t < tail_floats) voxels[total_f4 * 4 + t] = 0.f;     // bytes past the last float4
        VXL_WALL(2 * id + 1);
        return;
    }
    // ---- bin role: id -> (g, f) with the G bins of a frame on ONE XCD (workgroup i runs on XCD i % 8)
    const int q8 = id >> 3;
    const int lg = __ffs(G) - 1, gsh = 32 - lg;            // G is a power of two
    const int f = (id & 7) + 8 * (q8 >> lg), g = q8 & (G - 1);
    VXL_STAMP(0);
    if (f < p.batch) {
        const int start = vx_offset(p, offsets, f);
        const int n = min(vx_offset(p, offsets, f + 1) - start, p.n_max);
        const int nt = (n + 1023) >> 10;
        int *pinfo = w.flagw + (size_t)f * p.n_max;
        const int kstride = (p.n_max + 1023) & ~1023;
        uint32_t *pkey = w.pkey + (size_t)f * kstride;
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
        // ---- phase A0: my share of the frame's keys — 64-point groups [g * per, (g + 1) * per), exact cells
        const int ngroups = (n + 63) >> 6, per = (ngroups + G - 1) >> lg;
        const int q0 = g * per + wv, qend = min((g + 1) * per, ngroups);       // (per <= 32: at most two groups per wave)
        float ax[2], ay[2], az[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)                                              // requested first: the LDS tables are set up in
            vxl_load_xyz<C4>(points, (size_t)start + min(max((q0 + 16 * k) * 64 + l, 0), max(n - 1, 0)), p.C, ax[k], ay[k], az[k]);   // their shadow
        for (int k = t; k < VXL_S / 4; k += 1024) {
            reinterpret_cast<uint4 *>(s_key)[k] = make_uint4(VX_EMPTY, VX_EMPTY, VX_EMPTY, VX_EMPTY);
            reinterpret_cast<int4 *>(s_cnt)[k] = make_int4(0, 0, 0, 0);
        }
        if (t < VXL_CAP / 32) s_head[t] = 0u;
        if (t < 32) s_tc[t] = 0;
        if (t == 0) s_nent = s_total = s_slow = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = q0 + 16 * k, j = q * 64 + l;
            if (q < qend) pkey[j] = j < n ? vx_key2(p, ax[k], ay[k], az[k], j) : 0xFF000000u | (uint32_t)j;   // (padding of the last group)
        }
        for (int q = q0 + 32; q < qend; q += 16) {                               // (frames beyond 32 768 points never get here)
            const int j = q * 64 + l;
            float x, y, z;
            vxl_load_xyz<C4>(points, (size_t)start + min(j, n - 1), p.C, x, y, z);
            pkey[j] = j < n ? vx_key2(p, x, y, z, j) : 0xFF000000u | (uint32_t)j;
        }
        __syncthreads();
        VXL_STAMP(1);
        // ---- phase A1.  A wave owns a contiguous run of points (its entries come out ascending, which keeps the later
        // per-entry stores of a wave close together); point (u, lane) of wave wv = wbase + u * 64 + lane.  The frame's keys are
        // read with L1-bypassing loads and a key that still reads VX_NOKEY (its producer has not stored it yet) is simply read
        // again, until the wave's deadline; past it the workgroup falls back to the streaming variant, which needs no shared keys.
        // ---- phase A1.  A wave owns a contiguous run of points (its entries come out ascending, which keeps the later
        // per-entry stores of a wave close together); point (u, lane) of wave wv = wbase + u * 64 + lane
        const int wbase = wv * nt * 64;
        const long long deadline = wall_clock64() + VXL_WAIT_TICKS;
        for (int u0 = 0; u0 < nt * 64; u0 += VXL_A_KEYS) {         // rounds of 1 280 keys per wave: one for frames up to 20 480 points
            vx_u4 kq[5];
            const int kb = wbase + u0;                              // key (r, lane, c) of the round = kb + (r * 64 + lane) * 4 + c
            uint32_t live = 0;                  // bit (r * 4 + c): key (r, lane, c) is a point of the frame in this wave's run
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = kb + (r * 64 + l) * 4 + c;
                    live |= (uint32_t)((j < n) & (j - wbase < nt * 64)) << (r * 4 + c);
                }
            for (;;) {
                vxl_load_keys20(pkey, (uint32_t)(kb + l * 4) * 4u, kq);     // (the row has VXL_A_KEYS entries of slack behind it)
                uint32_t miss = 0;
#pragma unroll
                for (int r = 0; r < 5; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) miss |= (uint32_t)(kq[r][c] == VX_NOKEY) << (r * 4 + c);
                constexpr bool expired_at_once = (VXL_WAIT_TICKS) <= 0;   // test build (build.py: liblidar_hip_vxl_nowait.so)
                if (!expired_at_once && __ballot((miss & live) != 0) == 0ull) break;       // wave-uniform
                if (expired_at_once || wall_clock64() > deadline) {    // the frame's other workgroups are not (all) running beside
                    s_slow = 1;                                        // this one: the workgroup redoes its bin from the points
                    live = 0;                                          // themselves (streaming variant) after the barrier
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            uint32_t bits = 0;                  // bit (r * 4 + c): that point goes to my bin
            int wtotal = 0;                     // wave-uniform: entries of this wave in this round
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool mine = (vxl_bin_of24(kq[r][c], gsh) == g) & ((live >> (r * 4 + c)) & 1u);
                    bits |= (uint32_t)mine << (r * 4 + c);
                    wtotal += __popcll(__ballot(mine));
                }
            int base = 0;
            if (l == 0 && wtotal) base = atomicAdd(&s_nent, wtotal);
            base = __shfl(base, 0, 64);
            int j0 = kb + l * 4;                // (opaque to the optimiser: the twenty point numbers are re-derived from this one
            asm volatile("" : "+v"(j0));        //  register at the store, not kept in twenty registers since the `live` loop above)
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool mine = (bits >> (r * 4 + c)) & 1u;
                    const unsigned long long bal = __ballot(mine);
                    const int pos = base + __popcll(bal & lanemask_lt());
                    if (mine && pos < VXL_CAP) s_q[pos] = make_int2(j0 + r * 256 + c, (int)kq[r][c]);
                    base += __popcll(bal);
                }
        }
        VXL_STAMP(2);
        __syncthreads();
        VXL_STAMP(3);
        VxlShared sh;
        sh.key = s_key; sh.cnt = s_cnt; sh.aux = s_aux; sh.q = s_q; sh.wtot = s_wtot; sh.tc = s_tc; sh.total = &s_total;
        if (__builtin_expect(s_nent > VXL_CAP || s_slow != 0, 0)) {  // block-uniform: degenerate input, or keys that never arrived
            vxl_bin_streaming<C4>(points, p, w, G, g, f, start, n, sh);
        } else {
            const int ne = s_nent;
            // ---- phase B: LDS hash table insert, count per voxel; every entry's list cell starts out free (a bin has at
            // most as many list cells as entries)
            for (int e = t; e < ne; e += 1024) {
                const int2 q = s_q[e];
                int en = q.x | (int)0x80000000;                       // outside the grid / table full: dead entry, word 0
                if (((uint32_t)q.y >> 24) != 0xFFu) {
                    const int slot = vxl_table_insert(s_key, (uint32_t)q.y);
                    if (slot >= 0) {
                        atomicAdd(&s_cnt[slot], 1);
                        en = q.x | (slot << 15);
                    } else {
                        vx_raise(w, 1);
                    }
                }
                s_q[e] = make_int2(en, VX_INF);
            }
            __syncthreads();
            // ---- phase C: list space per voxel (the keys are not needed any more: offsets overwrite them), list heads marked.
            // Allocated wave by wave with ONE LDS atomic per wave (no scan across the workgroup, no barrier inside): where a
            // voxel's list lies is internal — the per-point words carry the positions.
            {
                int m[VXL_S / 1024];
                int run = 0;
#pragma unroll
                for (int k = 0; k < VXL_S / 1024; ++k) {
                    m[k] = min(s_cnt[t * (VXL_S / 1024) + k], p.P);
                    run += m[k];
                }
                const int inc = wave_incl_scan(run);
                const int wtot = __shfl(inc, 63, 64);
                int o = 0;
                if (l == 0 && wtot) o = atomicAdd(&s_total, wtot);
                o = __shfl(o, 0, 64) + inc - run;
#pragma unroll
                for (int k = 0; k < VXL_S / 1024; ++k) {
                    s_key[t * (VXL_S / 1024) + k] = (uint32_t)o;
                    if (m[k] > 0) atomicOr(&s_head[o >> 5], 1u << (o & 31));
                    o += m[k];
                }
            }
            __syncthreads();
            VXL_STAMP(4);
            const int L = min(s_total, VXL_CAP);     // total <= ne <= VXL_CAP
            // ---- phase D: ordered lists in LDS
            for (int e = t; e < ne; e += 1024) {
                const int en = s_q[e].x;
                if (en < 0) continue;
                const int j = en & 0x7FFF, slot = (en >> 15) & (VXL_S - 1);
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
                    const int slot = (en >> 15) & (VXL_S - 1);
                    const int o = (int)s_key[slot];
                    if (s_q[o].y == j) word = min(s_cnt[slot], p.P) | ((g * VXL_CAP + o) << VXL_MBITS);
                }
                pinfo[j] = word;
                vxl_count_firsts(s_tc, word != 0, j, l);
            }
            if (C4) {                    // the points of the cells beyond each list's head, where the emit launch will look
                float4 *stg4 = w.stg4 + ((size_t)f * G + g) * VXL_CAP;
                const float4 *pts4 = reinterpret_cast<const float4 *>(points) + start;
                for (int k = t; k < L; k += 1024)
                    if (!((s_head[k >> 5] >> (k & 31)) & 1u)) stg4[k] = pts4[s_q[k].y];
            } else {
                int *stgi = w.stgi + ((size_t)f * G + g) * VXL_CAP;
                for (int k = t; k < L; k += 1024) stgi[k] = s_q[k].y;
            }
        }
        __syncthreads();
        VXL_STAMP(6);
        if (t < 32) w.tcnt[((size_t)f * G + g) * 32 + t] = s_tc[t];
        if (t == 0) {
            int tot = 0;
            for (int k = 0; k < 32; ++k) tot += s_tc[k];
            w.bvox[f * VXL_GMAX + g] = tot;                    // read by the emit launch
            // rows the fill roles of THIS call leave zero (the emit launch reads it): everything in resident mode
            if (id == 0) w.fillst[1] = resident ? 0x7fffffff : (int)vxl_fill_share((long long)w.fillst[0]);
        }
    }
    VXL_STAMP(7);
    // ---- done with the index build: the bin roles' share of the fill (help16 sixteenths of it, normally none)
    if (!res_hist && nhelp > 0) vxl_fill_chunks(dst, nmain + id, nbinwg, nchunks, lim_f4, t);
    VXL_WALL(2 * id + 1);
}

// Second launch: workgroup (tile, f) owns the 1024 points of its tile.  A first point's voxel id = (first points of the frame
// in earlier tiles, from the bin roles' per-tile counts) + (its ballot rank inside the tile) = first-appearance order; the
// thread writes slot 0 of that voxel's row (its own point: the list is ascending), coords from its own cell, and the count.
// The further slots of multi-point voxels are flattened over the WHOLE workgroup — (voxel, slot) pairs through an LDS
// descriptor list, one pair per thread and pass — so a 32-point pillar costs what a 2-point pillar costs: one round trip.
template <bool C4>
__global__ __launch_bounds__(1024, 8) void vxl_emit_kernel(const float *__restrict__ points, const int *__restrict__ offsets,
                                                        VxParams p, VxWs w, int G, float *__restrict__ voxels,
                                                        int *__restrict__ coords, int *__restrict__ num_points,
                                                        int *__restrict__ voxel_offsets, int resident) {
    __shared__ int s_part[16], s_wcnt[16], s_w2[16], s_e2[16];
    __shared__ int s_base;
    __shared__ int4 s_desc[1024];      // multi-point voxels of this tile: (row, staging position, first flat item, items)
    const int tile = blockIdx.x, f = blockIdx.y, t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int id = 1024 + f * (int)gridDim.x + tile;        // (stamp slot, -DVXL_STAMPS builds only)
    (void)id;
    VXL_WALL(2 * id);
    VXL_ESTAMP(0);
    const int i = tile * 1024 + t;
    // requested before anything else, at an address that does not depend on the frame's offsets: inside a detector step both are
    // cold misses (~1.5 us each) and the word heads the longest dependency chain of the launch (start -> word: 6.7 k -> 4.4 k cycles)
    const int wd = w.flagw[(size_t)f * p.n_max + min(i, p.n_max - 1)];        // unconditional load, masked below
    const int start = vx_offset(p, offsets, f);
    const int n = min(vx_offset(p, offsets, f + 1) - start, p.n_max);
    const int word = (i < n) ? wd : 0;
    // this thread's own point, requested together with its word: for a first point it IS slot 0 of the voxel's row
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (C4 && n > 0) me = reinterpret_cast<const float4 *>(points)[(size_t)start + min(i, n - 1)];   // n: block-uniform
    // first points of the frame in earlier tiles: sum of tcnt[f][g][tau < tile] over the G bins
    int v = 0;
    if (t < G * 32) {
        const int tau = t & 31;
        const int c = w.tcnt[(size_t)f * G * 32 + t];
        v = (tau < tile) ? c : 0;
    }
    // rows of the earlier frames (every frame's count is complete: the first launch has finished) — requested by wave 15 BEFORE
    // anybody waits for the per-tile counts above, so that the two are one round trip, not two
    int part = 0;
    if (wv == 15 && p.compact)
        for (int k0 = 0; k0 < f; k0 += 64) {
            const int k = k0 + l;
            part += (k < f) ? vxl_frame_rows(p, w, G, k) : 0;
        }
    v = wave_sum(v);
    if (l == 0) s_part[wv] = v;
    if (wv == 15) {
        part = wave_sum(part);
        if (l == 0) s_base = p.compact ? part : f * p.max_voxels;
    }
    if (i < ((n + 63) & ~63)) w.pkey[(size_t)f * ((p.n_max + 1023) & ~1023) + i] = VX_NOKEY;      // the frame's shared keys: unpublished again for the next call
    if (tile == 0 && f == 0 && wv == 14) {            // the batch's offsets table (off the critical path)
        int carry = 0;
        for (int k0 = 0; k0 < p.batch; k0 += 64) {
            const int k = k0 + l;
            const int c = (k < p.batch) ? vxl_frame_rows(p, w, G, k) : 0;
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
    // slots 1 and 2 of a multi-point voxel: their staged points are requested NOW, in the shadow of the ranking below (95 % of the
    // multi-point pillars of a uniform cloud hold two points, nearly all the rest three): most workgroups then skip the
    // flattened copy further down altogether
    const size_t fbase = (size_t)f * G * VXL_CAP;
    float4 pre1 = make_float4(0.f, 0.f, 0.f, 0.f), pre2 = pre1;
    if (C4 && word != 0) {
        const int c0 = word & VXL_MMASK;
        const float4 *sp = w.stg4 + fbase + (word >> VXL_MBITS);
        if (c0 >= 2) pre1 = sp[1];
        if (c0 >= 3) pre2 = sp[2];
    }
    VXL_ESTAMP(1);
    const unsigned long long bal = __ballot(word != 0);
    if (l == 0) s_wcnt[wv] = __popcll(bal);
    __syncthreads();
    VXL_ESTAMP(2);
    int r = 0;
    for (int k = 0; k < (G * 32 + 63) / 64 && k < 16; ++k) r += s_part[k];
    for (int k = 0; k < wv; ++k) r += s_wcnt[k];
    r += __popcll(bal & lanemask_lt());
    const bool first = (word != 0) && (r < p.max_voxels);     // else: not a first point / voxel beyond the cap (dropped with its points)
    const int cnt = word & VXL_MMASK;
    const int lpos = word >> VXL_MBITS;                       // list position inside the frame's staging arrays
    const size_t row = (size_t)s_base + r;
    const int rowlen = p.P * p.C;
    if (first) {
        float x0, y0, z0;
        if (C4) {
            float4 *out4 = reinterpret_cast<float4 *>(voxels) + row * p.P;
            out4[0] = me;
            if (cnt >= 2) out4[1] = pre1;
            if (cnt >= 3) out4[2] = pre2;
            x0 = me.x; y0 = me.y; z0 = me.z;
        } else {
            const float *q = points + ((size_t)start + i) * p.C;
            float *out = voxels + row * rowlen;
            for (int c = 0; c < p.C; ++c) out[c] = q[c];
            if ((long long)row >= cleared_rows)
                for (int e = cnt * p.C; e < rowlen; ++e) out[e] = 0.f;
            x0 = q[0]; y0 = q[1]; z0 = q[2];
        }
        const uint32_t nx = p.grid[0], ny = p.grid[1];
        uint32_t key;
        vx_cell(p, x0, y0, z0, key);
        reinterpret_cast<int4 *>(coords)[row] = make_int4(f, (int)(key / (nx * ny)), (int)((key / nx) % ny), (int)(key % nx));
        num_points[row] = cnt;
    }
    if (C4) {
        // rows beyond what the fill roles cleared: their zeros (slots cnt .. P - 1) are written here, the WAVE working through its
        // lanes' rows two at a time — 32 lanes x 16 B per row and pass, whole 512-byte runs (one thread per row wrote 16 B per
        // instruction at a 512 B stride)
        unsigned long long need = __ballot(first && (long long)row >= cleared_rows);      // wave-uniform
        while (need) {
            const int la = __builtin_ctzll(need);
            need &= need - 1ull;
            int lb = la;
            if (need) { lb = __builtin_ctzll(need); need &= need - 1ull; }
            const int src = (l < 32) ? la : lb;
            const size_t rr = (size_t)__shfl((long long)row, src, 64);
            const int cc = __shfl(cnt, src, 64);
            float4 *o = reinterpret_cast<float4 *>(voxels) + rr * p.P;
            if (l < 32 || lb != la)
                for (int sl = cc + (l & 31); sl < p.P; sl += 32) o[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    VXL_ESTAMP(3);
    // ---- the further slots of the multi-point voxels (from slot 3 when C == 4: 1 and 2 were prefetched; else from slot 1),
    // flattened over the workgroup
    constexpr int S0 = C4 ? 3 : 1;
    const int extra = first ? max(cnt - S0, 0) : 0;
    const unsigned long long bal2 = __ballot(extra > 0);
    const int inc2 = wave_incl_scan(extra);
    if (l == 63) s_e2[wv] = inc2;
    if (l == 0) s_w2[wv] = __popcll(bal2);
    __syncthreads();
    int dbase = 0, ibase = 0, nd = 0, ne = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int a = s_w2[k], b = s_e2[k];
        if (k < wv) { dbase += a; ibase += b; }
        nd += a; ne += b;
    }
    if (ne == 0) {                                             // block-uniform
        VXL_WALL(2 * id + 1);
        return;
    }
    if (extra > 0) s_desc[dbase + __popcll(bal2 & lanemask_lt())] = make_int4((int)row, lpos, ibase + inc2 - extra, extra);
    __syncthreads();
    for (int it = t; it < ne; it += 1024) {
        int lo = 0, hi = nd - 1;                               // the last descriptor whose first item is <= it
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_desc[mid].z <= it) lo = mid; else hi = mid - 1;
        }
        const int4 d = s_desc[lo];
        const int s = it - d.z + S0;
        if (C4) {
            reinterpret_cast<float4 *>(voxels)[(size_t)d.x * p.P + s] = w.stg4[fbase + d.y + s];
        } else {
            const float *q = points + ((size_t)start + w.stgi[fbase + d.y + s]) * p.C;
            float *out = voxels + (size_t)d.x * rowlen + (size_t)s * p.C;
            for (int c = 0; c < p.C; ++c) out[c] = q[c];
        }
    }
    VXL_WALL(2 * id + 1);
}

static int vxl_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// a = start of the first launch, a2 = its end, b0 = start of the last launch, b = its end; `recorded` only once a call has
// really carried the events (a call that takes another path leaves the timer untouched and disarms it)
struct VxTimer { hipEvent_t a, a2, b0, b; int recorded; };
static thread_local VxTimer *g_next_timer = nullptr;

// bin + fill launch, then emit: 2 launches
static void vxl_run_fused(const float *points, const int *point_offsets, const VxParams &p, const VxWs &w, bool c4,
                          float *voxels, int *coords, int *num_points, int *voxel_offsets, hipStream_t s, int resident) {
    static const int help16 = vxl_env_int("LIDAR_VXL_HELP16", 0);     // share (in 1/16) of the fill left to the bin roles
    static const int fillwg = vxl_env_int("LIDAR_VXL_FILLWG", 256);   // fill workgroups beside the bin workgroups
    const int G = vxl_bins(p.n_max);
    const int ntiles = divup(p.n_max, 1024);
    const long long total_floats = (long long)p.max_voxels * p.P * p.C * p.batch;
    const bool fill_f4 = (reinterpret_cast<uintptr_t>(voxels) & 15) == 0;
    const long long total_f4 = fill_f4 ? total_floats / 4 : 0;
    const long long tail_floats = total_floats - total_f4 * 4;
    const int nbinwg = 8 * G * divup(p.batch, 8);
    const int nfillwg = nbinwg < 512 - fillwg ? 512 - nbinwg : fillwg;       // two resident workgroups per CU
    if (tail_floats > 1024) (void)hipMemsetAsync(voxels, 0, (size_t)total_floats * sizeof(float), s);   // unaligned buffer
    const long long tf = tail_floats <= 1024 ? tail_floats : 0;
#ifdef VXL_STAMPS
    const bool emit_only = vxl_env_int("LIDAR_VXL_DEBUG_EMIT_ONLY", 0) != 0;    // probe: the second launch alone (workspace of the last call)
#else
    const bool emit_only = false;
#endif
    // a timer armed for this call (lidar_voxelize_time_next): the first launch carries its start event, the second its stop event —
    // timestamps of the dispatches themselves (what a kernel trace reports), free of the event-marker and queue overhead that a
    // hipEventRecord bracket around the call adds (4.7-5.5 us, more when other streams are alive in the process)
    VxTimer *tm = g_next_timer;
    g_next_timer = nullptr;
    hipEvent_t ev_a = tm ? tm->a : nullptr, ev_a2 = tm ? tm->a2 : nullptr, ev_b0 = tm ? tm->b0 : nullptr, ev_b = tm ? tm->b : nullptr;
    if (tm) tm->recorded = 1;
    const dim3 g1(nbinwg + nfillwg), blk(1024);
    const dim3 ge(ntiles, p.batch);
    if (!tm) {                           // the ordinary path: plain launches (the event-carrying form costs ~5 us of queue time per launch)
        if (emit_only) {}
        else if (c4) hipLaunchKernelGGL(vxl_keybin_kernel<true>, g1, blk, 0, s, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, (const int *)num_points);
        else hipLaunchKernelGGL(vxl_keybin_kernel<false>, g1, blk, 0, s, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, (const int *)num_points);
        if (c4) hipLaunchKernelGGL(vxl_emit_kernel<true>, ge, blk, 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, resident);
        else hipLaunchKernelGGL(vxl_emit_kernel<false>, ge, blk, 0, s, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, resident);
        return;
    }
    if (emit_only) {}
    else if (c4) hipExtLaunchKernelGGL(vxl_keybin_kernel<true>, g1, blk, 0, s, ev_a, ev_a2, 0, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, (const int *)num_points);
    else hipExtLaunchKernelGGL(vxl_keybin_kernel<false>, g1, blk, 0, s, ev_a, ev_a2, 0, points, point_offsets, p, w, G, nbinwg, nfillwg, voxels, total_f4, tf, help16, resident, (const int *)num_points);
    if (c4) hipExtLaunchKernelGGL(vxl_emit_kernel<true>, ge, blk, 0, s, ev_b0, ev_b, 0, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, resident);
    else hipExtLaunchKernelGGL(vxl_emit_kernel<false>, ge, blk, 0, s, ev_b0, ev_b, 0, points, point_offsets, p, w, G, voxels, coords, num_points, voxel_offsets, resident);
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
    hipLaunchKernelGGL(vx_ws_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, nh, nl, (long long)batch * ((n_max + 1023) & ~1023));
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

LIDAR_EXPORT int lidar_voxelize_hostoff(const float *points, const int *point_offsets, const int *host_offsets, int batch, int n_max,
                                        int num_features, const float *range6, const float *voxel_size3,
                                        const int *grid3, int max_points, int max_voxels, int compact, int algo, float *voxels,
                                        int *coords, int *num_points, int *voxel_offsets, void *ws, size_t ws_bytes,
                                        void *stream);

LIDAR_EXPORT int lidar_voxelize(const float *points, const int *point_offsets, int batch, int n_max,
                                int num_features, const float *range6, const float *voxel_size3,
                                const int *grid3, int max_points, int max_voxels, int compact, int algo, float *voxels,
                                int *coords, int *num_points, int *voxel_offsets, void *ws, size_t ws_bytes,
                                void *stream) {
    return lidar_voxelize_hostoff(points, point_offsets, nullptr, batch, n_max, num_features, range6, voxel_size3, grid3, max_points,
                                  max_voxels, compact, algo, voxels, coords, num_points, voxel_offsets, ws, ws_bytes, stream);
}

// The same call for a caller that ALSO knows the frame offsets on the host (a collate function always does): host_offsets =
// batch + 1 ints in host memory, equal to the device array's contents (null: none).  Up to VX_HOFF_MAX frames they travel as kernel
// arguments to the LDS-binned launches, which then start their first point read without waiting for a load of the offsets.
LIDAR_EXPORT int lidar_voxelize_hostoff(const float *points, const int *point_offsets, const int *host_offsets, int batch, int n_max,
                                        int num_features, const float *range6, const float *voxel_size3,
                                        const int *grid3, int max_points, int max_voxels, int compact, int algo, float *voxels,
                                        int *coords, int *num_points, int *voxel_offsets, void *ws, size_t ws_bytes,
                                        void *stream) {
    // a timer armed with lidar_voxelize_time_next belongs to THIS call whatever path it takes: only the LDS-binned launches
    // carry it, every other exit (argument error, global-hash path) disarms it so that no later call records into a stale handle
    VxTimer *armed = g_next_timer;
    g_next_timer = nullptr;
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
    p.hoff_n = 0;
    if (host_offsets && batch <= VX_HOFF_MAX) {
        p.hoff_n = batch + 1;
        for (int k = 0; k <= batch; ++k) p.hoff[k] = host_offsets[k];
    }
    VxWs w;
    if (vx_carve(ws, batch, n_max, max_voxels, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool c4 = (num_features == 4) && ((reinterpret_cast<uintptr_t>(points) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(voxels) & 15) == 0);
    // algo 0 = auto, 3 (and 1, its former three-launch variant) = LDS-binned, 2 launches, 4 = 3 with a resident output buffer
    // (n_max <= 32768, max_points < 16384, x / y grid below 2^24 cells), 2 = global hash table (any size)
    const bool lds_ok = (n_max <= VXL_MAX_ITEMS * 1024) && (max_points <= VXL_MMASK) &&
                        ((double)grid3[0] * (double)grid3[1] < 16777216.0) && grid3[2] < 255;
    if (algo == 1) algo = 3;
    if ((algo == 3 || algo == 4) && !lds_ok) return LIDAR_ERR_ARG;
    if (algo == 0) algo = lds_ok ? 3 : 2;
    if (algo == 3 || algo == 4) {
        const bool unaligned = ((reinterpret_cast<uintptr_t>(voxels) & 15) != 0);      // the resident clear wants 16-B rows
        g_next_timer = armed;
        vxl_run_fused(points, point_offsets, p, w, c4, voxels, coords, num_points, voxel_offsets, s,
                      (algo == 4 && compact && !unaligned) ? 1 : 0);
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

// ------------------------------------------------------------------ launch timer (measurement only)
LIDAR_EXPORT void lidar_timer_destroy(void *timer);
LIDAR_EXPORT void *lidar_timer_create(void) {
    VxTimer *t = new VxTimer{nullptr, nullptr, nullptr, nullptr, 0};
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->a2) != hipSuccess || hipEventCreate(&t->b0) != hipSuccess ||
        hipEventCreate(&t->b) != hipSuccess) {
        lidar_timer_destroy(t);
        return nullptr;
    }
    return t;
}
LIDAR_EXPORT void lidar_timer_destroy(void *timer) {
    VxTimer *t = (VxTimer *)timer;
    if (!t) return;
    if (g_next_timer == t) g_next_timer = nullptr;
    if (t->a) (void)hipEventDestroy(t->a);
    if (t->a2) (void)hipEventDestroy(t->a2);
    if (t->b0) (void)hipEventDestroy(t->b0);
    if (t->b) (void)hipEventDestroy(t->b);
    delete t;
}
// the calling thread's NEXT lidar_voxelize / lidar_voxelize_hostoff call records, if it takes the LDS-binned path, its first
// launch's start / end and its last launch's start / end into `timer`; on any other path the call disarms the timer unrecorded
LIDAR_EXPORT void lidar_voxelize_time_next(void *timer) { g_next_timer = (VxTimer *)timer; }
// waits for the stop event; milliseconds from the start of the first launch to the end of the last one (< 0: nothing recorded)
LIDAR_EXPORT float lidar_timer_elapsed_ms(void *timer) {
    VxTimer *t = (VxTimer *)timer;
    float ms = -1.f;
    if (!t || !t->recorded || hipEventSynchronize(t->b) != hipSuccess || hipEventElapsedTime(&ms, t->a, t->b) != hipSuccess) return -1.f;
    return ms;
}
// the same span split: out3 = {first launch, gap between the launches, last launch} in milliseconds (what a kernel trace of the
// call shows as the two durations and the idle time between them); 0 on success, LIDAR_ERR_ARG when nothing was recorded
LIDAR_EXPORT int lidar_timer_parts_ms(void *timer, float *out3) {
    VxTimer *t = (VxTimer *)timer;
    if (!t || !out3 || !t->recorded || hipEventSynchronize(t->b) != hipSuccess) return LIDAR_ERR_ARG;
    if (hipEventElapsedTime(&out3[0], t->a, t->a2) != hipSuccess || hipEventElapsedTime(&out3[1], t->a2, t->b0) != hipSuccess ||
        hipEventElapsedTime(&out3[2], t->b0, t->b) != hipSuccess) return LIDAR_ERR_LAUNCH;
    return LIDAR_OK;
}
