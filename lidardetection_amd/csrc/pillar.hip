// PointPillar / SECOND front-end on gfx950:
//   pfn_kernel      PillarVFE + single PFNLayer, eval mode (BatchNorm folded to scale/shift)
//                   reference: pcdet/models/backbones_3d/vfe/pillar_vfe.py:29-49 (PFNLayer), :94-123
//   mean_vfe_kernel MeanVFE: pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31
//   scatter_*       PointPillarScatter: pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37
//
// All HBM-bound.  PFN reads only the occupied point slots of each padded voxel row (padded slots are
// zero by contract and contribute the constant relu(shift) to the max), so traffic is ~one 128-B
// line per voxel in + 256 B out instead of the reference's ~10 passes over the padded tensor.
// Scatter writes each canvas element exactly once (zero-fill fused with the scatter) through an
// inverse cell->pillar map, instead of zeros + index-assign + stack (3 passes over 55 MB/frame).
#include "common.h"

// ------------------------------------------------------------------ PFN (10 -> Cout<=64, max over points)
struct PfnParams {
    float vx, vy, vz, xo, yo, zo;
    int P, C, cout, nfeat;     // nfeat = C + 6 (use_absolute_xyz) (+1 with_distance)
    int with_distance, coords_are_float, num_are_float;
};

// lane = output channel.  One wave walks voxels v = wave_global, wave_global + nwaves, ...
// C and DIST are template parameters so that the weight registers wt[] are statically indexed.
// The walk is software-pipelined: a voxel costs two DEPENDENT memory round trips (its point count, then the occupied slots that
// count selects) and ~100 instructions, so a wave that waits for each voxel in turn is idle 95 % of the time (77 us for 256 k
// pillars).  Here the count / coordinates of voxel i + 2 and the occupied slots of voxel i + 1 are in flight while voxel i is
// computed; sums and broadcasts use DPP / readlane (no LDS round trips).
// sum of lanes 0..31 (lanes >= 32 hold zero), wave-uniform result
__device__ __forceinline__ float pfn_sum32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // lane ^ 1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // lane ^ 2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // mirror within 8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // mirror within 16
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
}

template <int C, bool DIST>
__global__ __launch_bounds__(256) void pfn_kernel(const float *__restrict__ voxels, const void *__restrict__ num_points,
                                                  const void *__restrict__ coords, const float *__restrict__ weight /*(cout,nfeat)*/,
                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                  const int *__restrict__ nvox_dev, int nvox_host, PfnParams p,
                                                  float *__restrict__ out /*(V,cout)*/) {
    constexpr int NF = C + 6 + (DIST ? 1 : 0);
    const int l = lane_id();
    const int nv = nvox_dev ? min(*nvox_dev, nvox_host) : nvox_host;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const bool chan = l < p.cout;
    float wt[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) wt[k] = chan ? weight[l * NF + k] : 0.f;
    const float sc = chan ? scale[l] : 0.f, sh = chan ? shift[l] : 0.f;
    const float pad_val = fmaxf(sh, 0.f);  // a zeroed (padded) row: relu(0*scale + shift)
    struct Head { int n; float cz, cy, cx; };
    auto load_head = [&](int v) {          // wave-uniform addresses: scalar-ish broadcast loads
        Head h = {0, 0.f, 0.f, 0.f};
        if (v < nv) {
            h.n = p.num_are_float ? (int)((const float *)num_points)[v] : ((const int *)num_points)[v];
            if (p.coords_are_float) {
                const float4 c = ((const float4 *)coords)[v];
                h.cz = c.y; h.cy = c.z; h.cx = c.w;
            } else {
                const int4 c = ((const int4 *)coords)[v];
                h.cz = (float)c.y; h.cy = (float)c.z; h.cx = (float)c.w;
            }
            h.n = min(max(h.n, 0), p.P);
        }
        return h;
    };
    struct Pts { float v[C]; };
    auto load_pts = [&](int v, int n) {    // lanes 0..n-1 hold one point each (P <= 64); the rest zero
        Pts t;
#pragma unroll
        for (int k = 0; k < C; ++k) t.v[k] = 0.f;
        if (v < nv && l < n) {
            const float *q = voxels + ((size_t)v * p.P + l) * C;
            if (C == 4) {
                const float4 u = *reinterpret_cast<const float4 *>(q);
                t.v[0] = u.x; t.v[1] = u.y; t.v[2] = u.z; t.v[3 % C] = u.w;
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k) t.v[k] = q[k];
            }
        }
        return t;
    };
    Head ha = load_head(wave), hb = load_head(wave + nwaves);
    Pts pa = load_pts(wave, ha.n);
    for (int v = wave; v < nv; v += nwaves) {
        const Head hc = load_head(v + 2 * nwaves);          // two voxels ahead: its count is here when its slots are requested
        const Pts pb = load_pts(v + nwaves, hb.n);           // one voxel ahead
        const int n = ha.n;
        // mean over the real points == sum over the padded row / n (padded rows are zero)
        float sx, sy, sz;
        if (p.P <= 32) {
            sx = pfn_sum32(pa.v[0]); sy = pfn_sum32(pa.v[1]); sz = pfn_sum32(pa.v[2]);
        } else {
            sx = pa.v[0]; sy = pa.v[1]; sz = pa.v[2];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sx += __shfl_xor(sx, d, 64);
                sy += __shfl_xor(sy, d, 64);
                sz += __shfl_xor(sz, d, 64);
            }
        }
        const float fn = (float)n;
        const float mx = sx / fn, my = sy / fn, mz = sz / fn;
        const float ox = ha.cx * p.vx + p.xo, oy = ha.cy * p.vy + p.yo, oz = ha.cz * p.vz + p.zo;
        float best = (n < p.P) ? pad_val : -INFINITY;
        for (int q = 0; q < n; ++q) {                        // q is wave-uniform: v_readlane broadcasts
            float f[C];
#pragma unroll
            for (int k = 0; k < C; ++k) f[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pa.v[k]), q));
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < C; ++k) acc = fmaf(f[k], wt[k], acc);
            acc = fmaf(f[0] - mx, wt[C + 0], acc);
            acc = fmaf(f[1] - my, wt[C + 1], acc);
            acc = fmaf(f[2] - mz, wt[C + 2], acc);
            acc = fmaf(f[0] - ox, wt[C + 3], acc);
            acc = fmaf(f[1] - oy, wt[C + 4], acc);
            acc = fmaf(f[2] - oz, wt[C + 5], acc);
            if (DIST) acc = fmaf(sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]), wt[NF - 1], acc);
            best = fmaxf(best, fmaxf(fmaf(acc, sc, sh), 0.f));
        }
        if (chan) out[(size_t)v * p.cout + l] = best;
        ha = hb; hb = hc; pa = pb;
    }
}

// P <= 32: TWO voxels per wave iteration — lanes 0..31 hold the points of voxel A, lanes 32..63 those of voxel B, so the count /
// coordinate loads, the slot loads, the DPP row sums and the mean / pillar-centre arithmetic are issued once for both; only the
// (point, channel) FMA loop runs per voxel.  The kernel is bound by instruction issue (one wave per voxel, ~110 instructions for
// 1-2 points), not by its 77 MB of traffic: sharing the per-voxel overhead is what pays.  Same expressions per voxel as pfn_kernel.
template <int C, bool DIST>
__global__ __launch_bounds__(256) void pfn_pair_kernel(const float *__restrict__ voxels, const void *__restrict__ num_points,
                                                       const void *__restrict__ coords, const float *__restrict__ weight,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const int *__restrict__ nvox_dev, int nvox_host, PfnParams p,
                                                       float *__restrict__ out) {
    constexpr int NF = C + 6 + (DIST ? 1 : 0);
    const int l = lane_id(), half = l >> 5, pl = l & 31;
    const int nv = nvox_dev ? min(*nvox_dev, nvox_host) : nvox_host;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const bool chan = l < p.cout;
    float wt[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) wt[k] = chan ? weight[l * NF + k] : 0.f;
    const float sc = chan ? scale[l] : 0.f, sh = chan ? shift[l] : 0.f;
    const float pad_val = fmaxf(sh, 0.f);
    struct Head { int n; float cz, cy, cx; };                  // per lane: the voxel of this lane's half
    auto load_head = [&](int pair) {
        const int v = 2 * pair + half;
        Head h = {0, 0.f, 0.f, 0.f};
        if (v < nv) {
            h.n = p.num_are_float ? (int)((const float *)num_points)[v] : ((const int *)num_points)[v];
            if (p.coords_are_float) {
                const float4 c = ((const float4 *)coords)[v];
                h.cz = c.y; h.cy = c.z; h.cx = c.w;
            } else {
                const int4 c = ((const int4 *)coords)[v];
                h.cz = (float)c.y; h.cy = (float)c.z; h.cx = (float)c.w;
            }
            h.n = min(max(h.n, 0), p.P);
        }
        return h;
    };
    struct Pts { float v[C]; };
    auto load_pts = [&](int pair, int n) {
        const int v = 2 * pair + half;
        Pts t;
#pragma unroll
        for (int k = 0; k < C; ++k) t.v[k] = 0.f;
        if (v < nv && pl < n) {
            const float *q = voxels + ((size_t)v * p.P + pl) * C;
            if (C == 4) {
                const float4 u = *reinterpret_cast<const float4 *>(q);
                t.v[0] = u.x; t.v[1] = u.y; t.v[2] = u.z; t.v[3 % C] = u.w;
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k) t.v[k] = q[k];
            }
        }
        return t;
    };
    auto row_sum16 = [](float v) {                              // sum of each 16-lane row, in all of its lanes
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
        return v;
    };
    auto rl = [](float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); };
    const int npairs = (nv + 1) >> 1;
    Head ha = load_head(wave), hb = load_head(wave + nwaves);
    Pts pa = load_pts(wave, ha.n);
    for (int pr = wave; pr < npairs; pr += nwaves) {
        const Head hc = load_head(pr + 2 * nwaves);             // two pairs ahead: the counts are here when their slots are requested
        const Pts pb = load_pts(pr + nwaves, hb.n);             // one pair ahead
        const float vox = ha.cx * p.vx + p.xo, voy = ha.cy * p.vy + p.yo, voz = ha.cz * p.vz + p.zo;
        if (__builtin_amdgcn_readlane(ha.n, 0) <= 1 && __builtin_amdgcn_readlane(ha.n, 32) <= 1) {
            // both voxels hold at most ONE point (3 pairs in 5 on a 20k-point KITTI frame): the mean IS the point, so the three
            // (point - mean) features are exactly zero and their products add exactly nothing — no row sums, no divisions
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int v = 2 * pr + h;
                if (v >= nv) break;                              // wave-uniform
                const int b0 = 32 * h;
                float best = pad_val;                            // n <= 1 < P: padded slots exist
                if (__builtin_amdgcn_readlane(ha.n, b0) == 1) {
                    float f[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) f[k] = rl(pa.v[k], b0);
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < C; ++k) acc = fmaf(f[k], wt[k], acc);
                    acc = fmaf(f[0] - rl(vox, b0), wt[C + 3], acc);
                    acc = fmaf(f[1] - rl(voy, b0), wt[C + 4], acc);
                    acc = fmaf(f[2] - rl(voz, b0), wt[C + 5], acc);
                    if (DIST) acc = fmaf(sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]), wt[NF - 1], acc);
                    best = fmaxf(best, fmaxf(fmaf(acc, sc, sh), 0.f));
                }
                if (chan) out[(size_t)v * p.cout + l] = best;
            }
            ha = hb; hb = hc; pa = pb;
            continue;
        }
        // both voxels at once, on the lanes of their halves: sums -> means, pillar centre
        const float rx = row_sum16(pa.v[0]), ry = row_sum16(pa.v[1]), rz = row_sum16(pa.v[2]);
        const float hx = half ? rl(rx, 32) + rl(rx, 48) : rl(rx, 0) + rl(rx, 16);
        const float hy = half ? rl(ry, 32) + rl(ry, 48) : rl(ry, 0) + rl(ry, 16);
        const float hz = half ? rl(rz, 32) + rl(rz, 48) : rl(rz, 0) + rl(rz, 16);
        const float fn = (float)ha.n;
        const float vmx = hx / fn, vmy = hy / fn, vmz = hz / fn;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = 2 * pr + h;
            if (v >= nv) break;                                  // wave-uniform
            const int b0 = 32 * h;
            const int n = __builtin_amdgcn_readlane(ha.n, b0);
            const float mx = rl(vmx, b0), my = rl(vmy, b0), mz = rl(vmz, b0);
            const float ox = rl(vox, b0), oy = rl(voy, b0), oz = rl(voz, b0);
            float best = (n < p.P) ? pad_val : -INFINITY;
            for (int q = 0; q < n; ++q) {                        // wave-uniform: v_readlane broadcasts
                float f[C];
#pragma unroll
                for (int k = 0; k < C; ++k) f[k] = rl(pa.v[k], b0 + q);
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < C; ++k) acc = fmaf(f[k], wt[k], acc);
                acc = fmaf(f[0] - mx, wt[C + 0], acc);
                acc = fmaf(f[1] - my, wt[C + 1], acc);
                acc = fmaf(f[2] - mz, wt[C + 2], acc);
                acc = fmaf(f[0] - ox, wt[C + 3], acc);
                acc = fmaf(f[1] - oy, wt[C + 4], acc);
                acc = fmaf(f[2] - oz, wt[C + 5], acc);
                if (DIST) acc = fmaf(sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]), wt[NF - 1], acc);
                best = fmaxf(best, fmaxf(fmaf(acc, sc, sh), 0.f));
            }
            if (chan) out[(size_t)v * p.cout + l] = best;
        }
        ha = hb; hb = hc; pa = pb;
    }
}

template <int C>
static void pfn_launch(int blocks, hipStream_t s, bool dist, const float *voxels, const void *num_points,
                       const void *coords, const float *weight, const float *scale, const float *shift,
                       const int *nvd, int nv, PfnParams p, float *out) {
    if (p.P <= 32) {       // two voxels per wave iteration
        if (dist)
            hipLaunchKernelGGL((pfn_pair_kernel<C, true>), dim3(blocks), dim3(256), 0, s, voxels, num_points, coords, weight, scale,
                               shift, nvd, nv, p, out);
        else
            hipLaunchKernelGGL((pfn_pair_kernel<C, false>), dim3(blocks), dim3(256), 0, s, voxels, num_points, coords, weight, scale,
                               shift, nvd, nv, p, out);
        return;
    }
    if (dist)
        hipLaunchKernelGGL((pfn_kernel<C, true>), dim3(blocks), dim3(256), 0, s, voxels, num_points, coords, weight, scale,
                           shift, nvd, nv, p, out);
    else
        hipLaunchKernelGGL((pfn_kernel<C, false>), dim3(blocks), dim3(256), 0, s, voxels, num_points, coords, weight, scale,
                           shift, nvd, nv, p, out);
}

LIDAR_EXPORT int lidar_pillar_vfe(const float *voxels, const void *num_points, const void *coords, int num_voxels,
                                  const int *num_voxels_dev, int max_points, int num_features,
                                  const float *weight, const float *scale, const float *shift, int cout,
                                  const float *voxel_size3, const float *range6, int with_distance,
                                  int coords_are_float, int num_are_float, float *out, void *stream) {
    if (!voxels || !num_points || !coords || !weight || !scale || !shift || !out) return LIDAR_ERR_ARG;
    if (num_voxels < 0 || max_points <= 0 || max_points > 64 || num_features < 3 || num_features > 8) return LIDAR_ERR_ARG;
    if (cout <= 0 || cout > 64) return LIDAR_ERR_ARG;
    if (num_voxels == 0) return LIDAR_OK;
    PfnParams p;
    p.vx = voxel_size3[0]; p.vy = voxel_size3[1]; p.vz = voxel_size3[2];
    p.xo = p.vx / 2 + range6[0]; p.yo = p.vy / 2 + range6[1]; p.zo = p.vz / 2 + range6[2];
    p.P = max_points; p.C = num_features; p.cout = cout;
    p.with_distance = with_distance ? 1 : 0;
    p.nfeat = num_features + 6 + p.with_distance;
    p.coords_are_float = coords_are_float; p.num_are_float = num_are_float;
    const int waves_needed = num_voxels;
    int blocks = divup(waves_needed, 4 * 8);  // ~8 voxels per wave
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    const bool d = with_distance != 0;
#define PFN_CASE(CC) case CC: pfn_launch<CC>(blocks, s, d, voxels, num_points, coords, weight, scale, shift, num_voxels_dev, num_voxels, p, out); break;
    switch (num_features) {
        PFN_CASE(3) PFN_CASE(4) PFN_CASE(5) PFN_CASE(6) PFN_CASE(7) PFN_CASE(8)
        default: return LIDAR_ERR_ARG;
    }
#undef PFN_CASE
    return lidar_check_launch("lidar_pillar_vfe");
}

// ------------------------------------------------------------------ MeanVFE
__global__ __launch_bounds__(256) void mean_vfe_kernel(const float *__restrict__ voxels, const void *__restrict__ num_points,
                                                       int nv, int P, int C, int num_are_float, float *__restrict__ out) {
    const long long total = (long long)nv * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int v = (int)(t / C), c = (int)(t - (long long)v * C);
        float n = num_are_float ? ((const float *)num_points)[v] : (float)((const int *)num_points)[v];
        const float *row = voxels + (size_t)v * P * C + c;
        float s = 0.f;
        for (int q = 0; q < P; ++q) s += row[(size_t)q * C];  // sequential sum == torch.sum over dim 1 for small P
        out[t] = s / fmaxf(n, 1.0f);
    }
}

LIDAR_EXPORT int lidar_mean_vfe(const float *voxels, const void *num_points, int num_voxels, int max_points,
                                int num_features, int num_are_float, float *out, void *stream) {
    if (!voxels || !num_points || !out || num_voxels < 0 || max_points <= 0 || num_features <= 0) return LIDAR_ERR_ARG;
    if (num_voxels == 0) return LIDAR_OK;
    int blocks = divup((long long)num_voxels * num_features, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mean_vfe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, voxels, num_points, num_voxels,
                       max_points, num_features, num_are_float, out);
    return lidar_check_launch("lidar_mean_vfe");
}

// ------------------------------------------------------------------ scatter to BEV canvas
// pass 1: cell -> pillar-row map (B*ny*nx ints, -1 = empty).  pass 2: canvas written once.
__global__ void scatter_fill_kernel(int *__restrict__ map, long long n) {
    const long long n4 = n >> 2;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n4; i += stride) reinterpret_cast<int4 *>(map)[i] = make_int4(-1, -1, -1, -1);
    if (tid < (n & 3)) map[(n4 << 2) + tid] = -1;
}

__global__ void scatter_index_kernel(const void *__restrict__ coords, int coords_are_float, int nvox_host,
                                     const int *__restrict__ nvox_dev, int B, int nx, int ny, int hrows, int *__restrict__ map) {
    const int nv = nvox_dev ? min(*nvox_dev, nvox_host) : nvox_host;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x) {
        int b, z, y, x;
        if (coords_are_float) {
            const float4 c = ((const float4 *)coords)[v];
            b = (int)c.x; z = (int)c.y; y = (int)c.z; x = (int)c.w;
        } else {
            const int4 c = ((const int4 *)coords)[v];
            b = c.x; z = c.y; y = c.z; x = c.w;
        }
        // PointPillarScatter: z + y*nx + x with nz == 1 (pointpillar_scatter.py:27) == (z*hrows + y)*nx + x with z == 0;
        // SparseConvTensor.dense(): the (D*H, W) view of a (D, H, W) volume, hrows = H
        const long long cell = ((long long)z * hrows + y) * nx + x;
        if (b >= 0 && b < B && cell >= 0 && cell < (long long)nx * ny) map[(size_t)b * nx * ny + cell] = v;
    }
}

#define SC_XT 128  // cells per tile along x
// block = (x-tile, y, b); 256 threads.  LDS holds the features of the occupied cells of the tile.
template <int CH>
__global__ __launch_bounds__(256) void scatter_canvas_kernel(const float *__restrict__ feat /*(V,CH)*/,
                                                             const int *__restrict__ map, int nx, int ny,
                                                             float *__restrict__ canvas /*(B,CH,ny,nx)*/) {
    __shared__ float s_feat[SC_XT][CH + 1];
    __shared__ int s_idx[SC_XT];
    const int x0 = blockIdx.x * SC_XT, y = blockIdx.y, b = blockIdx.z;
    const int w = min(SC_XT, nx - x0);
    const int *mrow = map + ((size_t)b * ny + y) * nx + x0;
    if (threadIdx.x < SC_XT) s_idx[threadIdx.x] = ((int)threadIdx.x < w) ? mrow[threadIdx.x] : -1;
    __syncthreads();
    // gather occupied rows (coalesced 4*CH-byte rows) into LDS
    for (int t = threadIdx.x; t < SC_XT * CH; t += 256) {
        const int cx = t / CH, c = t - cx * CH;
        const int v = s_idx[cx];
        if (v >= 0) s_feat[cx][c] = feat[(size_t)v * CH + c];
    }
    __syncthreads();
    float *cbase = canvas + (((size_t)b * CH) * ny + y) * nx + x0;
    const size_t cstride = (size_t)ny * nx;
    if (((nx & 3) == 0) && ((w & 3) == 0)) {
        const int w4 = w >> 2;
        for (int t = threadIdx.x; t < CH * w4; t += 256) {
            const int c = t / w4, q = t - c * w4;
            float4 o;
            const int i0 = q * 4;
            o.x = s_idx[i0 + 0] >= 0 ? s_feat[i0 + 0][c] : 0.f;
            o.y = s_idx[i0 + 1] >= 0 ? s_feat[i0 + 1][c] : 0.f;
            o.z = s_idx[i0 + 2] >= 0 ? s_feat[i0 + 2][c] : 0.f;
            o.w = s_idx[i0 + 3] >= 0 ? s_feat[i0 + 3][c] : 0.f;
            reinterpret_cast<float4 *>(cbase + c * cstride)[q] = o;
        }
    } else {
        for (int t = threadIdx.x; t < CH * w; t += 256) {
            const int c = t / w, i = t - c * w;
            cbase[c * cstride + i] = s_idx[i] >= 0 ? s_feat[i][c] : 0.f;
        }
    }
}

// channels-last canvas (B, ny, nx, CH): a cell's CH channels are contiguous, so the scatter is a row copy or a zero row;
// one float4 per thread, both sides fully coalesced.  (Logical shape (B, CH, ny, nx) with torch.channels_last strides.)
template <int CH>
__global__ __launch_bounds__(256) void scatter_canvas_nhwc_kernel(const float *__restrict__ feat, const int *__restrict__ map,
                                                                  long long cells, float *__restrict__ canvas) {
    constexpr int Q = CH / 4;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= cells * Q) return;
    const long long cell = e / Q;
    const int q = (int)(e - cell * Q);
    const int v = map[cell];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v >= 0) o = reinterpret_cast<const float4 *>(feat)[(size_t)v * Q + q];
    reinterpret_cast<float4 *>(canvas)[e] = o;
}

LIDAR_EXPORT size_t lidar_pillar_scatter_workspace_bytes(int batch, int nx, int ny) {
    return align_up((size_t)batch * nx * ny * 4, 256);
}

LIDAR_EXPORT int lidar_pillar_scatter(const float *pillar_features, const void *coords, int coords_are_float,
                                      int num_voxels, const int *num_voxels_dev, int channels, int batch, int nx,
                                      int ny, int channels_last, float *canvas, void *ws, size_t ws_bytes, void *stream) {
    if (!pillar_features || !coords || !canvas || !ws) return LIDAR_ERR_ARG;
    if (batch <= 0 || nx <= 0 || ny <= 0 || num_voxels < 0) return LIDAR_ERR_ARG;
    if (channels != 64 && channels != 32 && channels != 128) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_pillar_scatter_workspace_bytes(batch, nx, ny)) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int *map = (int *)ws;
    const long long cells = (long long)batch * nx * ny;
    int fb = divup(cells, 256 * 4);
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(fb), dim3(256), 0, s, map, cells);
    if (num_voxels > 0) {
        int ib = divup(num_voxels, 256);
        hipLaunchKernelGGL(scatter_index_kernel, dim3(ib), dim3(256), 0, s, coords, coords_are_float, num_voxels,
                           num_voxels_dev, batch, nx, ny, ny, map);
    }
    if (channels_last) {
        const int nb = divup(cells * (channels / 4), 256);
        if (channels == 64) hipLaunchKernelGGL(scatter_canvas_nhwc_kernel<64>, dim3(nb), dim3(256), 0, s, pillar_features, map, cells, canvas);
        else if (channels == 32) hipLaunchKernelGGL(scatter_canvas_nhwc_kernel<32>, dim3(nb), dim3(256), 0, s, pillar_features, map, cells, canvas);
        else hipLaunchKernelGGL(scatter_canvas_nhwc_kernel<128>, dim3(nb), dim3(256), 0, s, pillar_features, map, cells, canvas);
        return lidar_check_launch("lidar_pillar_scatter(nhwc)");
    }
    const dim3 grid(divup(nx, SC_XT), ny, batch);
    if (channels == 64)
        hipLaunchKernelGGL(scatter_canvas_kernel<64>, grid, dim3(256), 0, s, pillar_features, map, nx, ny, canvas);
    else if (channels == 32)
        hipLaunchKernelGGL(scatter_canvas_kernel<32>, grid, dim3(256), 0, s, pillar_features, map, nx, ny, canvas);
    else
        hipLaunchKernelGGL(scatter_canvas_kernel<128>, grid, dim3(256), 0, s, pillar_features, map, nx, ny, canvas);
    return lidar_check_launch("lidar_pillar_scatter");
}

// ------------------------------------------------------------------ the backbone's FIRST convolution straight from the pillars
// PointPillarScatter writes a canvas that is > 90 % zeros and the backbone's first layer (base_bev_backbone.py:34-39: ZeroPad2d(1) +
// Conv2d(3x3, stride s) + BatchNorm + ReLU) then multiplies all of them: 63 GFLOP for PointPillar-KITTI at bs 16, of which the
// 16 000 pillars per frame reach 2.25 output pixels each: 4.7 GFLOP.  This builds, from the pillar coordinates alone, the
// neighbour table of that convolution over ALL output pixels in map order — nbr[(b * OH + oy) * OW + ox][ky * k + kx] = the
// pillar row at input cell (oy * stride - pad + ky, ox * stride - pad + kx) or -1 — so that the sparse implicit GEMM
// (csrc/sparse_conv.hip, rows grouped by their tap mask) can write the layer's DENSE NHWC output directly: rows with taps get
// the convolution, rows without get act(bias), every output element written exactly once.  The canvas is never built.
// ws: (batch * ny * nx) ints (inverse cell -> pillar map, rebuilt every call).
__global__ __launch_bounds__(256) void pillar_conv_table_kernel(const int *__restrict__ map, int B, int ny, int nx, int OH, int OW, int k,
                                                                int stride, int pad, int *__restrict__ nbr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one thread per (output pixel, tap): coalesced table writes
    const int K = k * k;
    const long long total = (long long)B * OH * OW * K;
    if (i >= total) return;
    const int tap = (int)(i % K);
    long long o = i / K;
    const int ox = (int)(o % OW);
    o /= OW;
    const int oy = (int)(o % OH);
    const int b = (int)(o / OH);
    const int iy = oy * stride - pad + tap / k, ix = ox * stride - pad + tap % k;
    nbr[i] = (iy >= 0 && iy < ny && ix >= 0 && ix < nx) ? map[((size_t)b * ny + iy) * nx + ix] : -1;
}

LIDAR_EXPORT size_t lidar_pillar_conv_table_workspace_bytes(int batch, int nx, int ny) {
    return align_up((size_t)batch * nx * ny * 4, 256);
}

LIDAR_EXPORT int lidar_pillar_conv_table(const void *coords, int coords_are_float, int num_voxels, const int *num_voxels_dev, int batch,
                                         int nx, int ny, int k, int stride, int pad, int *nbr, void *ws, size_t ws_bytes, void *stream) {
    if (!coords || !nbr || !ws || batch <= 0 || nx <= 0 || ny <= 0 || num_voxels < 0 || k <= 0 || k > 5 || stride <= 0 || pad < 0)
        return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_pillar_conv_table_workspace_bytes(batch, nx, ny)) return LIDAR_ERR_WORKSPACE;
    const int OH = (ny + 2 * pad - k) / stride + 1, OW = (nx + 2 * pad - k) / stride + 1;
    if (OH <= 0 || OW <= 0) return LIDAR_ERR_ARG;
    const long long cells = (long long)batch * nx * ny, total = (long long)batch * OH * OW * k * k;
    if (cells > 0x7fffffffll || total > 0x7fffffffll * 256ll) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int *map = (int *)ws;
    int fb = divup(cells, 256 * 4);
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(fb), dim3(256), 0, s, map, cells);
    if (num_voxels > 0)
        hipLaunchKernelGGL(scatter_index_kernel, dim3(divup(num_voxels, 256)), dim3(256), 0, s, coords, coords_are_float, num_voxels,
                           num_voxels_dev, batch, nx, ny, ny, map);
    hipLaunchKernelGGL(pillar_conv_table_kernel, dim3((unsigned)divup(total, 256)), dim3(256), 0, s, map, batch, ny, nx, OH, OW, k, stride, pad, nbr);
    return lidar_check_launch("lidar_pillar_conv_table");
}

// Resident canvas: a BEV canvas is > 90 % zeros, and a step's pillars touch a few per cent of its cells.  Instead of rewriting
// the whole canvas every call (877 MB for PointPillar-KITTI at bs 16), the caller keeps ONE channels-last canvas in HBM; a
// call clears the cells the previous call wrote (their ids are remembered) and writes the new pillars: ~2 x V x CH x 4 bytes.
// The result is identical to a freshly zeroed + scattered canvas (PointPillarScatter, pointpillar_scatter.py:14-37).
template <int CH>
__global__ __launch_bounds__(256) void canvas_clear_cells_kernel(const int *__restrict__ prev_cells, const int *__restrict__ prev_count,
                                                                 int cap, float *__restrict__ canvas) {
    constexpr int Q = CH / 4;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n = min(*prev_count, cap);
    const long long v = e / Q;
    if (v >= n) return;
    const int cell = prev_cells[v];
    if (cell >= 0) reinterpret_cast<float4 *>(canvas)[(long long)cell * Q + (e - v * Q)] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int CH>
__global__ __launch_bounds__(256) void canvas_write_cells_kernel(const float *__restrict__ feat, const void *__restrict__ coords,
                                                                 int coords_are_float, int nvox_host, const int *__restrict__ nvox_dev,
                                                                 int B, int nx, int ny, float *__restrict__ canvas,
                                                                 int *__restrict__ prev_cells, int *__restrict__ prev_count) {
    constexpr int Q = CH / 4;
    const int nv = nvox_dev ? min(*nvox_dev, nvox_host) : nvox_host;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) *prev_count = nv;             // the clear kernel of THIS call has already consumed the old value
    const long long v = e / Q;
    if (v >= nv) return;
    const int q = (int)(e - v * Q);
    int b, z, y, x;
    if (coords_are_float) {
        const float4 c = ((const float4 *)coords)[v];
        b = (int)c.x; z = (int)c.y; y = (int)c.z; x = (int)c.w;
    } else {
        const int4 c = ((const int4 *)coords)[v];
        b = c.x; z = c.y; y = c.z; x = c.w;
    }
    const long long cell_in = (long long)z * ny * nx + (long long)y * nx + x;      // z + y*nx + x with nz == 1 (pointpillar_scatter.py:27)
    const bool ok = b >= 0 && b < B && z == 0 && y >= 0 && y < ny && x >= 0 && x < nx;
    const long long cell = (long long)b * nx * ny + cell_in;
    if (q == 0) prev_cells[v] = ok ? (int)cell : -1;
    if (ok) reinterpret_cast<float4 *>(canvas)[cell * Q + q] = reinterpret_cast<const float4 *>(feat)[v * Q + q];
}

// canvas (B, ny, nx, CH) channels-last, persistent across calls, initially all zeros with *prev_count == 0;
// prev_cells: num_voxels ints of caller-owned state.
LIDAR_EXPORT int lidar_pillar_scatter_update(const float *pillar_features, const void *coords, int coords_are_float, int num_voxels,
                                             const int *num_voxels_dev, int channels, int batch, int nx, int ny, float *canvas,
                                             int *prev_cells, int *prev_count, void *stream) {
    if (!pillar_features || !coords || !canvas || !prev_cells || !prev_count) return LIDAR_ERR_ARG;
    if (batch <= 0 || nx <= 0 || ny <= 0 || num_voxels < 0 || (long long)batch * nx * ny > 0x7fffffffll) return LIDAR_ERR_ARG;
    if (channels != 64 && channels != 32 && channels != 128) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nb = divup((long long)(num_voxels > 0 ? num_voxels : 1) * (channels / 4), 256);
#define CU_CASE(CH)                                                                                                               \
    hipLaunchKernelGGL(canvas_clear_cells_kernel<CH>, dim3(nb), dim3(256), 0, s, prev_cells, prev_count, num_voxels, canvas);         \
    hipLaunchKernelGGL(canvas_write_cells_kernel<CH>, dim3(nb), dim3(256), 0, s, pillar_features, coords, coords_are_float, num_voxels, \
                       num_voxels_dev, batch, nx, ny, canvas, prev_cells, prev_count)
    if (channels == 64) { CU_CASE(64); } else if (channels == 32) { CU_CASE(32); } else { CU_CASE(128); }
#undef CU_CASE
    return lidar_check_launch("lidar_pillar_scatter_update");
}

// SparseConvTensor.dense() (spconv; consumer pcdet/models/backbones_2d/map_to_bev/height_compression.py:21-23):
// features (N, C) at indices (N, 4) [b,z,y,x] -> zeros-filled (B, C, D, H, W), written once. C in {32,64,128}.
LIDAR_EXPORT size_t lidar_sparse_to_dense_workspace_bytes(int batch, int D, int H, int W) {
    return align_up((size_t)batch * D * H * W * 4, 256);
}

LIDAR_EXPORT int lidar_sparse_to_dense(const float *features, const int *indices, int n, int channels, int batch, int D, int H,
                                       int W, float *out, void *ws, size_t ws_bytes, void *stream) {
    if (!features || !indices || !out || !ws || batch <= 0 || D <= 0 || H <= 0 || W <= 0 || n < 0) return LIDAR_ERR_ARG;
    if (channels != 64 && channels != 32 && channels != 128) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_sparse_to_dense_workspace_bytes(batch, D, H, W)) return LIDAR_ERR_WORKSPACE;
    if ((long long)D * H > 65535) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int *map = (int *)ws;
    const long long cells = (long long)batch * D * H * W;
    int fb = divup(cells, 256 * 4);
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(fb), dim3(256), 0, s, map, cells);
    if (n > 0)
        hipLaunchKernelGGL(scatter_index_kernel, dim3(divup(n, 256)), dim3(256), 0, s, (const void *)indices, 0, n,
                           (const int *)nullptr, batch, W, D * H, H, map);
    const dim3 grid(divup(W, SC_XT), D * H, batch);
    if (channels == 64) hipLaunchKernelGGL(scatter_canvas_kernel<64>, grid, dim3(256), 0, s, features, map, W, D * H, out);
    else if (channels == 32) hipLaunchKernelGGL(scatter_canvas_kernel<32>, grid, dim3(256), 0, s, features, map, W, D * H, out);
    else hipLaunchKernelGGL(scatter_canvas_kernel<128>, grid, dim3(256), 0, s, features, map, W, D * H, out);
    return lidar_check_launch("lidar_sparse_to_dense");
}

// HeightCompression output straight in NHWC (height_compression.py:21-24: dense (N, C, D, H, W).view(N, C*D, H, W), which
// the dense BEV backbone then wants channels-last): out[b][h][w][c*D + d] = features[row at (b, d, h, w)][c] or 0, every
// element written once; saves the NCDHW volume + the layout conversion pass.  map: (B, D*H, W) inverse index from
// scatter_index_kernel.  One thread = one pixel's 4 consecutive source channels x all D levels = 4*D consecutive floats.
template <int D>
__global__ __launch_bounds__(256) void bev_nhwc_kernel(const float *__restrict__ feat, const int *__restrict__ map, long long n_pix,
                                                       int C4, int H, int W, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix * C4) return;
    const long long pix = i / C4;
    const int c4 = (int)(i - pix * C4);
    const int w = (int)(pix % W);
    const long long bh = pix / W;
    const int h = (int)(bh % H);
    const long long b = bh / H;
    float4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int row = map[((b * D + d) * H + h) * (long long)W + w];
        v[d] = row >= 0 ? reinterpret_cast<const float4 *>(feat)[(size_t)row * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float *o = out + (pix * C4 + c4) * (4 * D);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        o[0 * D + d] = v[d].x;
        o[1 * D + d] = v[d].y;
        o[2 * D + d] = v[d].z;
        o[3 * D + d] = v[d].w;
    }
}

LIDAR_EXPORT int lidar_sparse_to_bev_nhwc(const float *features, const int *indices, int n, int channels, int batch, int D, int H,
                                          int W, float *out, void *ws, size_t ws_bytes, void *stream) {
    if (!features || !indices || !out || !ws || batch <= 0 || D <= 0 || D > 4 || H <= 0 || W <= 0 || n < 0) return LIDAR_ERR_ARG;
    if (channels <= 0 || (channels & 3)) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_sparse_to_dense_workspace_bytes(batch, D, H, W)) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int *map = (int *)ws;
    const long long cells = (long long)batch * D * H * W;
    int fb = divup(cells, 256 * 4);
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(fb), dim3(256), 0, s, map, cells);
    if (n > 0)
        hipLaunchKernelGGL(scatter_index_kernel, dim3(divup(n, 256)), dim3(256), 0, s, (const void *)indices, 0, n,
                           (const int *)nullptr, batch, W, D * H, H, map);
    const long long n_pix = (long long)batch * H * W;
    const int C4 = channels / 4;
    const long long blocks = (n_pix * C4 + 255) / 256;
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
#define BEV_CASE(DD) hipLaunchKernelGGL(bev_nhwc_kernel<DD>, dim3((unsigned)blocks), dim3(256), 0, s, features, map, n_pix, C4, H, W, out)
    if (D == 1) BEV_CASE(1); else if (D == 2) BEV_CASE(2); else if (D == 3) BEV_CASE(3); else BEV_CASE(4);
#undef BEV_CASE
    return lidar_check_launch("lidar_sparse_to_bev_nhwc");
}
