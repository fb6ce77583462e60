// Host-side (CPU) entry points of the reference's extension modules.  The reference calls these from DataLoader
// worker processes (GT-sampling augmentation, database creation), where no GPU context may be touched:
//   boxes_iou_bev_cpu    pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252 (geometry :38-229)
//   points_in_boxes_cpu  pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168 (MARGIN 1e-2)
// Plain C++ compiled for the host only (no kernels); fp32 arithmetic in the reference's order, libm float trig.
#include "common.h"
#include <cmath>

namespace {

struct V2 {
    float x, y;
};

inline float cr3(const V2 &p1, const V2 &p2, const V2 &p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

inline bool crossing(const V2 &p1, const V2 &p0, const V2 &q1, const V2 &q0, V2 &ans) {
    const float pxmin = p0.x > p1.x ? p1.x : p0.x, pxmax = p0.x > p1.x ? p0.x : p1.x;
    const float pymin = p0.y > p1.y ? p1.y : p0.y, pymax = p0.y > p1.y ? p0.y : p1.y;
    const float qxmin = q0.x > q1.x ? q1.x : q0.x, qxmax = q0.x > q1.x ? q0.x : q1.x;
    const float qymin = q0.y > q1.y ? q1.y : q0.y, qymax = q0.y > q1.y ? q0.y : q1.y;
    if (!(pxmin <= qxmax && qxmin <= pxmax && pymin <= qymax && qymin <= pymax)) return false;
    const float s1 = cr3(q0, p1, p0), s2 = cr3(p1, q1, p0), s3 = cr3(p0, q1, q0), s4 = cr3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cr3(q1, p1, p0);
    if (std::fabs(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

inline bool corner_inside(const float *box, const V2 &p) {
    const float MARGIN = 1e-2f;
    const float ac = std::cos(-box[6]), as = std::sin(-box[6]);   // float overloads, as in the reference
    const float rx = (p.x - box[0]) * ac + (p.y - box[1]) * (-as);
    const float ry = (p.x - box[0]) * as + (p.y - box[1]) * ac;
    return std::fabs(rx) < box[3] / 2 + MARGIN && std::fabs(ry) < box[4] / 2 + MARGIN;
}

inline void corners_of(const float *b, V2 *c) {
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    const float ca = std::cos(b[6]), sa = std::sin(b[6]);
    const V2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; ++k) {
        c[k].x = (raw[k].x - b[0]) * ca + (raw[k].y - b[1]) * (-sa) + b[0];
        c[k].y = (raw[k].x - b[0]) * sa + (raw[k].y - b[1]) * ca + b[1];
    }
    c[4] = c[0];
}

float bev_overlap(const float *a, const float *b) {
    V2 A[5], B[5], v[24];
    corners_of(a, A);
    corners_of(b, B);
    int cnt = 0;
    V2 ctr = {0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (crossing(A[i + 1], A[i], B[j + 1], B[j], v[cnt])) {
                ctr.x = ctr.x + v[cnt].x;
                ctr.y = ctr.y + v[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (corner_inside(a, B[k])) { ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y; v[cnt++] = B[k]; }
        if (corner_inside(b, A[k])) { ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y; v[cnt++] = A[k]; }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (std::atan2(v[i].y - ctr.y, v[i].x - ctr.x) > std::atan2(v[i + 1].y - ctr.y, v[i + 1].x - ctr.x)) {
                const V2 t = v[i];
                v[i] = v[i + 1];
                v[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; ++k)
        area += (v[k].x - v[0].x) * (v[k + 1].y - v[0].y) - (v[k].y - v[0].y) * (v[k + 1].x - v[0].x);
    return (float)(std::fabs(area) / 2.0);
}

}  // namespace

// boxes_iou_bev_cpu: HOST pointers; out (n_a, n_b)
LIDAR_EXPORT int lidar_boxes_iou_bev_cpu(const float *boxes_a, int n_a, const float *boxes_b, int n_b, float *out) {
    if (n_a < 0 || n_b < 0 || (n_a && n_b && (!boxes_a || !boxes_b || !out))) return LIDAR_ERR_ARG;
    for (int i = 0; i < n_a; ++i)
        for (int j = 0; j < n_b; ++j) {
            const float *a = boxes_a + 7 * i, *b = boxes_b + 7 * j;
            const float sa = a[3] * a[4], sb = b[3] * b[4];
            const float s = bev_overlap(a, b);
            out[(size_t)i * n_b + j] = s / fmaxf(sa + sb - s, 1e-8f);
        }
    return LIDAR_OK;
}

// points_in_boxes_cpu: HOST pointers; out (n_boxes, n_pts) 0/1
LIDAR_EXPORT int lidar_points_in_boxes_cpu(const float *boxes, int n_boxes, const float *pts, int n_pts, int *out) {
    if (n_boxes < 0 || n_pts < 0 || (n_boxes && n_pts && (!boxes || !pts || !out))) return LIDAR_ERR_ARG;
    const float MARGIN = 1e-2f;
    for (int i = 0; i < n_boxes; ++i) {
        const float *b = boxes + 7 * i;
        const float cosa = std::cos(-b[6]), sina = std::sin(-b[6]);
        for (int j = 0; j < n_pts; ++j) {
            const float *p = pts + 3 * j;
            int in = 0;
            if (!(fabsf(p[2] - b[2]) > b[5] / 2.0)) {
                const float sx = p[0] - b[0], sy = p[1] - b[1];
                const float lx = sx * cosa + sy * (-sina), ly = sx * sina + sy * cosa;
                const float flag = (std::fabs(lx) < b[3] / 2.0 + MARGIN) & (std::fabs(ly) < b[4] / 2.0 + MARGIN);
                in = (int)flag;
            }
            out[(size_t)i * n_pts + j] = in;
        }
    }
    return LIDAR_OK;
}

// ------------------------------------------------------------------ host voxel generator
// spconv's VoxelGeneratorV2.generate as the reference calls it from DataLoader WORKER processes
// (pcdet/datasets/processor/data_processor.py:48-80; workers are forked, pcdet/datasets/__init__.py:73, and must not touch
// the GPU).  Same result as lidar_voxelize / the sequential scan (SURVEY.md Appendix A.1, v1.2 `continue` semantics), but
// written for a host core: an open-addressing hash map over the occupied cells (a few hundred KB in L2) instead of spconv's
// dense coor_to_voxelidx grid (360 MB for the SECOND grid, touched at random), and only the rows that are produced get
// their zero padding written (the reference zero-fills max_voxels x P x C up front).
LIDAR_EXPORT size_t lidar_voxelize_cpu_scratch_bytes(int n) {
    size_t cap = 1024;
    while (cap < 2 * (size_t)(n > 0 ? n : 1)) cap <<= 1;
    return cap * 8;
}

LIDAR_EXPORT int lidar_voxelize_cpu(const float *points, int n, int num_features, const float *range6, const float *voxel_size3,
                                    const int *grid3, int max_points, int max_voxels, float *voxels, int *coords_zyx,
                                    int *num_points, void *scratch, size_t scratch_bytes) {
    if (!points || !range6 || !voxel_size3 || !grid3 || !voxels || !coords_zyx || !num_points || !scratch) return LIDAR_ERR_ARG;
    if (n < 0 || num_features < 3 || max_points <= 0 || max_voxels <= 0) return LIDAR_ERR_ARG;
    if ((double)grid3[0] * grid3[1] * grid3[2] >= 4294967295.0) return LIDAR_ERR_ARG;
    size_t cap = 1024;
    while (cap < 2 * (size_t)(n > 0 ? n : 1)) cap <<= 1;
    if (scratch_bytes < cap * 8) return LIDAR_ERR_WORKSPACE;
    uint32_t *keys = static_cast<uint32_t *>(scratch);
    int *vals = reinterpret_cast<int *>(keys + cap);
    for (size_t k = 0; k < cap; ++k) keys[k] = 0xFFFFFFFFu;
    const uint32_t mask = (uint32_t)cap - 1u;
    const int C = num_features, P = max_points;
    const uint32_t nx = (uint32_t)grid3[0], ny = (uint32_t)grid3[1];
    const size_t row = (size_t)P * C;
    int nvox = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * C;
        const float fx = std::floor((p[0] - range6[0]) / voxel_size3[0]);
        const float fy = std::floor((p[1] - range6[1]) / voxel_size3[1]);
        const float fz = std::floor((p[2] - range6[2]) / voxel_size3[2]);
        if (!(fx >= 0.f && fx < (float)grid3[0] && fy >= 0.f && fy < (float)grid3[1] && fz >= 0.f && fz < (float)grid3[2])) continue;
        const uint32_t cx = (uint32_t)fx, cy = (uint32_t)fy, cz = (uint32_t)fz;
        const uint32_t key = (cz * ny + cy) * nx + cx;
        uint32_t h = (key * 2654435761u) & mask;
        int vid = -1;
        for (;;) {
            const uint32_t k = keys[h];
            if (k == key) { vid = vals[h]; break; }
            if (k == 0xFFFFFFFFu) break;
            h = (h + 1u) & mask;
        }
        if (vid < 0) {
            if (nvox >= max_voxels) continue;          // v1.2: skip the point, keep scanning
            vid = nvox++;
            keys[h] = key;
            vals[h] = vid;
            coords_zyx[3 * vid + 0] = (int)cz;
            coords_zyx[3 * vid + 1] = (int)cy;
            coords_zyx[3 * vid + 2] = (int)cx;
            num_points[vid] = 0;
            float *r = voxels + (size_t)vid * row;
            for (size_t e = 0; e < row; ++e) r[e] = 0.f;
        }
        const int slot = num_points[vid];
        if (slot < P) {
            float *dst = voxels + (size_t)vid * row + (size_t)slot * C;
            for (int c = 0; c < C; ++c) dst[c] = p[c];
            num_points[vid] = slot + 1;
        }
    }
    return nvox;
}
