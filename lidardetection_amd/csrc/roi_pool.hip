// RoI-aware / RoI-point pooling and point-in-box assignment on gfx950.
// Reference: pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu (in-box test :14-36, mask :39-75,
//            collect :78-108, max/avg pool :111-190, backward :236-286, points_in_boxes :313-336) and
//            pcdet/ops/roipoint_pool3d/src/roipoint_pool3d_kernel.cu (:38-134).
//
// Order semantics that must survive parallelisation: per voxel the FIRST max_pts-1 in-box points in point
// order (roiaware), per box the FIRST S in-box points (roipoint), per point the LOWEST box id.  The reference
// gets them from one serial thread per box over a (boxes x points) scratch matrix; here one wave owns a box,
// walks the points 64 at a time in order and compacts with ballot / popcount, so there is no scratch matrix,
// no cudaMalloc, and the serial loops become wave-wide.  The in-box test is evaluated exactly as written in
// the reference (including its float->double promoted comparisons).
#include "common.h"

struct BoxCS {
    float cx, cy, cz, dx, dy, dz, cosa, sina;
};

// cos(-rz), sin(-rz): correctly rounded fp32 via double (see iou3d.hip heading_cs)
__device__ __forceinline__ BoxCS make_boxcs(const float *b) {
    BoxCS r;
    r.cx = b[0]; r.cy = b[1]; r.cz = b[2]; r.dx = b[3]; r.dy = b[4]; r.dz = b[5];
    const float a = -b[6];
    r.cosa = (float)cos((double)a);
    r.sina = (float)sin((double)a);
    return r;
}

// check_pt_in_box3d (roiaware_pool3d_kernel.cu:23-36 == roipoint_pool3d_kernel.cu:22-35), MARGIN = 1e-5f
__device__ __forceinline__ bool pt_in_box(const BoxCS &b, float x, float y, float z, float &lx, float &ly) {
    const float MARGIN = 1e-5f;
    if ((double)fabsf(z - b.cz) > (double)b.dz / 2.0) return false;
    const float sx = x - b.cx, sy = y - b.cy;
    lx = sx * b.cosa + sy * (-b.sina);
    ly = sx * b.sina + sy * b.cosa;
    return ((double)fabsf(lx) < (double)b.dx / 2.0 + (double)MARGIN) && ((double)fabsf(ly) < (double)b.dy / 2.0 + (double)MARGIN);
}

// ------------------------------------------------------------------ points_in_boxes_gpu
// boxes staged through LDS in tiles of PIB_TILE (cos / sin once per box and workgroup): any number of boxes, as the reference
// (roiaware_pool3d_kernel.cu:313-336: lowest box id containing the point, first hit wins)
#define PIB_TILE 1024
__global__ __launch_bounds__(256) void points_in_boxes_kernel(int B, int T, int P, const float *__restrict__ boxes,
                                                              const float *__restrict__ pts, int *__restrict__ out) {
    __shared__ BoxCS s_box[PIB_TILE];
    const int bb = blockIdx.y, t = threadIdx.x;
    const int p = blockIdx.x * 256 + t;
    float x = 0.f, y = 0.f, z = 0.f;
    if (p < P) {
        const float *q = pts + ((size_t)bb * P + p) * 3;
        x = q[0]; y = q[1]; z = q[2];
    }
    int found = p < P ? -1 : 0;                       // threads past the end have nothing to look for
    for (int k0 = 0; k0 < T; k0 += PIB_TILE) {
        const int nk = min(PIB_TILE, T - k0);
        __syncthreads();                              // the previous tile has been read by everyone
        for (int k = t; k < nk; k += 256) s_box[k] = make_boxcs(boxes + ((size_t)bb * T + k0 + k) * 7);
        __syncthreads();
        if (found < 0) {
            float lx, ly;
            for (int k = 0; k < nk; ++k)
                if (pt_in_box(s_box[k], x, y, z, lx, ly)) {
                    found = k0 + k;                   // lowest box id
                    break;
                }
        }
        if (__syncthreads_and(found >= 0)) break;     // every point of the workgroup is settled
    }
    if (p < P && found >= 0) out[(size_t)bb * P + p] = found;   // the caller pre-fills -1
}

LIDAR_EXPORT int lidar_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts,
                                       int *box_idx_of_points, void *stream) {
    if (batch <= 0 || boxes_num < 0 || pts_num < 0) return LIDAR_ERR_ARG;
    if (pts_num == 0 || boxes_num == 0) return LIDAR_OK;
    if (!boxes || !pts || !box_idx_of_points) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(points_in_boxes_kernel, dim3(divup(pts_num, 256), batch), dim3(256), 0, (hipStream_t)stream, batch, boxes_num,
                       pts_num, boxes, pts, box_idx_of_points);
    return lidar_check_launch("lidar_points_in_boxes");
}

// ------------------------------------------------------------------ roiaware_pool3d forward
// phase 1 (one wave per box): in-box test + voxel index + ordered per-voxel lists (pts_idx_of_voxels).
#define RA_LDS_VOX 8192
__global__ __launch_bounds__(64) void roiaware_collect_kernel(int R, int P, int ox, int oy, int oz, int maxpts,
                                                              const float *__restrict__ rois, const float *__restrict__ pts,
                                                              int *__restrict__ pidx) {
    __shared__ int s_cnt[RA_LDS_VOX];   // per-voxel counters of this box (LDS when the RoI grid fits, else global slot 0)
    const int r = blockIdx.x, l = threadIdx.x;
    const int nv = ox * oy * oz;
    const bool lds_cnt = nv <= RA_LDS_VOX;
    const BoxCS b = make_boxcs(rois + (size_t)r * 7);
    int *pv = pidx + (size_t)r * nv * maxpts;
    if (lds_cnt)
        for (int v = l; v < nv; v += 64) s_cnt[v] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const float xr = b.dx / ox, yr = b.dy / oy, zr = b.dz / oz;
    const int cap = maxpts - 1;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + l;
        int vox = -1;
        if (p < P) {
            const float x = pts[(size_t)p * 3], y = pts[(size_t)p * 3 + 1], z = pts[(size_t)p * 3 + 2];
            float lx, ly;
            if (pt_in_box(b, x, y, z, lx, ly)) {
                const float lz = z - b.cz;
                unsigned xi = (unsigned)(int)((lx + b.dx / 2) / xr);
                unsigned yi = (unsigned)(int)((ly + b.dy / 2) / yr);
                unsigned zi = (unsigned)(int)((lz + b.dz / 2) / zr);
                xi = min(xi, (unsigned)(ox - 1));   // min(max(u,0),out-1) on unsigned (kernel.cu:68-70)
                yi = min(yi, (unsigned)(oy - 1));
                zi = min(zi, (unsigned)(oz - 1));
                vox = (int)((xi * (unsigned)oy + yi) * (unsigned)oz + zi);
            }
        }
        // ordered append: points of this 64-chunk that share a voxel are ranked by lane order
        unsigned long long pending = __ballot(vox >= 0);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int v = __shfl(vox, leader, 64);
            const unsigned long long same = __ballot(vox == v);
            int *lst = pv + (size_t)v * maxpts;
            const int base = lds_cnt ? s_cnt[v] : __hip_atomic_load(&lst[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (vox == v) {
                const int pos = base + __popcll(same & lanemask_lt());
                if (pos < cap) lst[pos + 1] = p;
            }
            __builtin_amdgcn_wave_barrier();
            if (l == leader) {
                const int nc = min(base + __popcll(same), cap);
                if (lds_cnt) s_cnt[v] = nc;
                else __hip_atomic_store(&lst[0], nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pending &= ~same;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lds_cnt)
        for (int v = l; v < nv; v += 64) pv[(size_t)v * maxpts] = s_cnt[v];
}

// phase 2: pooling, channel on the lane (coalesced feature rows); one thread per (voxel, channel)
__global__ __launch_bounds__(256) void roiaware_pool_kernel(long long nvox, int C, int maxpts, int pool_method,
                                                            const float *__restrict__ feat, const int *__restrict__ pidx,
                                                            float *__restrict__ pooled, int *__restrict__ argmax) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nvox * C) return;
    const long long v = e / C;
    const int c = (int)(e - v * C);
    const int *lst = pidx + v * maxpts;
    const int total = lst[0];
    if (pool_method == 0) {
        int am = -1;
        float mv = -INFINITY;
        for (int k = 1; k <= total; ++k) {
            const int pi = lst[k];
            const float f = feat[(size_t)pi * C + c];
            if (f > mv) {
                mv = f;
                am = pi;
            }
        }
        if (am != -1) pooled[e] = mv;
        argmax[e] = am;
    } else {
        float s = 0.f;
        for (int k = 1; k <= total; ++k) s += feat[(size_t)lst[k] * C + c];
        if (total > 0) pooled[e] = s / total;
    }
}

// roiaware_pool3d_gpu (roiaware_pool3d.cpp:29-66): argmax / pts_idx_of_voxels / pooled are zero-filled by the caller
LIDAR_EXPORT int lidar_roiaware_pool3d_forward(int boxes_num, int pts_num, int channels, int max_pts_each_voxel, int out_x,
                                               int out_y, int out_z, const float *rois, const float *pts,
                                               const float *pts_feature, int *argmax, int *pts_idx_of_voxels,
                                               float *pooled_features, int pool_method, void *stream) {
    if (boxes_num < 0 || pts_num < 0 || channels <= 0 || max_pts_each_voxel < 2) return LIDAR_ERR_ARG;
    if (out_x <= 0 || out_y <= 0 || out_z <= 0 || out_x >= 256 || out_y >= 256 || out_z >= 256) return LIDAR_ERR_ARG;
    if (pool_method != 0 && pool_method != 1) return LIDAR_ERR_ARG;
    if (boxes_num == 0) return LIDAR_OK;
    if (!rois || !pts || !pts_feature || !pts_idx_of_voxels || !pooled_features || (pool_method == 0 && !argmax)) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (pts_num > 0)
        hipLaunchKernelGGL(roiaware_collect_kernel, dim3(boxes_num), dim3(64), 0, s, boxes_num, pts_num, out_x, out_y, out_z,
                           max_pts_each_voxel, rois, pts, pts_idx_of_voxels);
    const long long nvox = (long long)boxes_num * out_x * out_y * out_z;
    hipLaunchKernelGGL(roiaware_pool_kernel, dim3(divup(nvox * channels, 256)), dim3(256), 0, s, nvox, channels,
                       max_pts_each_voxel, pool_method, pts_feature, pts_idx_of_voxels, pooled_features, argmax);
    return lidar_check_launch("lidar_roiaware_pool3d_forward");
}

// backward (kernel.cu:236-286): grad_in (P, C) zero-filled by the caller; channel on the lane
__global__ __launch_bounds__(256) void roiaware_backward_kernel(long long nvox, int C, int maxpts, int pool_method,
                                                                const int *__restrict__ pidx, const int *__restrict__ argmax,
                                                                const float *__restrict__ grad_out, float *__restrict__ grad_in) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nvox * C) return;
    const long long v = e / C;
    const int c = (int)(e - v * C);
    if (pool_method == 0) {
        const int a = argmax[e];
        if (a == -1) return;
        atomicAdd(&grad_in[(size_t)a * C + c], grad_out[e] * 1);
    } else {
        const int *lst = pidx + v * maxpts;
        const int total = lst[0];
        const float g = 1 / fmaxf((float)total, 1.0f);
        for (int k = 1; k <= total; ++k) atomicAdd(&grad_in[(size_t)lst[k] * C + c], grad_out[e] * g);
    }
}

LIDAR_EXPORT int lidar_roiaware_pool3d_backward(int boxes_num, int out_x, int out_y, int out_z, int channels,
                                                int max_pts_each_voxel, const int *pts_idx_of_voxels, const int *argmax,
                                                const float *grad_out, float *grad_in, int pool_method, void *stream) {
    if (boxes_num < 0 || channels <= 0 || out_x <= 0 || out_y <= 0 || out_z <= 0) return LIDAR_ERR_ARG;
    if (pool_method != 0 && pool_method != 1) return LIDAR_ERR_ARG;
    if (boxes_num == 0) return LIDAR_OK;
    if (!grad_out || !grad_in || (pool_method == 0 && !argmax) || (pool_method == 1 && !pts_idx_of_voxels)) return LIDAR_ERR_ARG;
    const long long nvox = (long long)boxes_num * out_x * out_y * out_z;
    hipLaunchKernelGGL(roiaware_backward_kernel, dim3(divup(nvox * channels, 256)), dim3(256), 0, (hipStream_t)stream, nvox,
                       channels, max_pts_each_voxel, pool_method, pts_idx_of_voxels, argmax, grad_out, grad_in);
    return lidar_check_launch("lidar_roiaware_pool3d_backward");
}

// ------------------------------------------------------------------ roipoint_pool3d forward
// one workgroup per (box, batch): wave 0 selects the first S in-box points in order (ballot compaction,
// early exit), pads cyclically, then all waves gather xyz + features rows (channel on the lane).
#define RP_MAX_S 1024
__global__ __launch_bounds__(256) void roipoint_pool_kernel(int B, int N, int M, int C, int S, const float *__restrict__ xyz,
                                                            const float *__restrict__ boxes, const float *__restrict__ feat,
                                                            float *__restrict__ pooled, int *__restrict__ empty_flag) {
    __shared__ int s_sel[RP_MAX_S];
    __shared__ int s_cnt;
    const int m = blockIdx.x, bb = blockIdx.y, t = threadIdx.x;
    if (t < 64) {
        const BoxCS b = make_boxcs(boxes + ((size_t)bb * M + m) * 7);
        int cnt = 0;
        for (int p0 = 0; p0 < N && cnt < S; p0 += 64) {
            const int p = p0 + t;
            bool in = false;
            if (p < N) {
                const float *q = xyz + ((size_t)bb * N + p) * 3;
                float lx, ly;
                in = pt_in_box(b, q[0], q[1], q[2], lx, ly);
            }
            const unsigned long long bal = __ballot(in);
            const int pos = cnt + __popcll(bal & lanemask_lt());
            if (in && pos < S) s_sel[pos] = p;
            cnt = min(cnt + __popcll(bal), S);
        }
        if (t == 0) s_cnt = cnt;
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt == 0) {
        if (t == 0) empty_flag[(size_t)bb * M + m] = 1;
        return;
    }
    const int W = 3 + C;
    float *dst = pooled + ((size_t)bb * M + m) * S * W;
    for (long long e = t; e < (long long)S * W; e += 256) {
        const int s = (int)(e / W), j = (int)(e - (long long)s * W);
        const int src = s_sel[s < cnt ? s : (s % cnt)];   // cyclic duplication (kernel.cu:91-98)
        dst[e] = j < 3 ? xyz[((size_t)bb * N + src) * 3 + j] : feat[((size_t)bb * N + src) * C + (j - 3)];
    }
}

// roipool3d_gpu (roipoint_pool3d.cpp:23-54): pooled and empty_flag zero-filled by the caller
LIDAR_EXPORT int lidar_roipoint_pool3d_forward(int batch, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                               const float *xyz, const float *boxes3d, const float *pts_feature,
                                               float *pooled_features, int *pooled_empty_flag, void *stream) {
    if (batch <= 0 || pts_num < 0 || boxes_num < 0 || feature_len < 0 || sampled_pts_num <= 0 || sampled_pts_num > RP_MAX_S)
        return LIDAR_ERR_ARG;
    if (boxes_num == 0) return LIDAR_OK;
    if (!xyz || !boxes3d || !pooled_features || !pooled_empty_flag || (feature_len > 0 && !pts_feature)) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(roipoint_pool_kernel, dim3(boxes_num, batch), dim3(256), 0, (hipStream_t)stream, batch, pts_num, boxes_num,
                       feature_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag);
    return lidar_check_launch("lidar_roipoint_pool3d_forward");
}
