// KITTI-eval rotated BEV IoU: rotate_iou_gpu_eval / rotate_iou_kernel_eval of the reference
// (pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py:256-330, numba.cuda — not available on ROCm).
// Box format (x, y, w, l, angle); criterion -1: IoU, 0: inter / area(query), 1: inter / area(box), 2: inter.
// One thread per (box, query) pair; the arithmetic follows numba's typing of the reference source (float32 arrays, mixed
// int / float-literal expressions in float64, intersection area accumulated in float64), without FMA contraction.
#include "common.h"

namespace {

struct Quad { float c[8]; };

__device__ __forceinline__ Quad ev_corners(const float *rb) {   // rbbox_to_corners (:200-223)
    const float a_cos = cosf(rb[4]), a_sin = sinf(rb[4]);
    const float hx = (float)((double)rb[2] / 2), hy = (float)((double)rb[3] / 2);
    const float cx[4] = {-hx, -hx, hx, hx}, cy[4] = {-hy, hy, hy, -hy};
    Quad q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q.c[2 * i] = a_cos * cx[i] + a_sin * cy[i] + rb[0];
        q.c[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + rb[1];
    }
    return q;
}

__device__ __forceinline__ bool ev_inside(float px, float py, const float *c) {   // point_in_quadrilateral (:157-173)
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

__device__ __forceinline__ bool ev_cross(const float *p1, const float *p2, int i, int j, float &ox, float &oy) {   // (:72-116)
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd == bcd) return false;
    const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
    if (abc == abd) return false;
    const float DC0 = D0 - C0, DC1 = D1 - C1;
    const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
    const float DH = BA1 * DC0 - BA0 * DC1;
    ox = (ABBA * DC0 - BA0 * CDDC) / DH;
    oy = (ABBA * DC1 - BA1 * CDDC) / DH;
    return true;
}

__device__ double ev_inter(const float *rb1, const float *rb2) {   // inter (:226-239)
    const Quad q1 = ev_corners(rb1), q2 = ev_corners(rb2);
    float pts[32];   // up to 16 points (the reference's unchecked 8-point buffer can overflow on coincident boxes)
    int n = 0;
    for (int i = 0; i < 4; ++i) {                                  // quadrilateral_intersection (:176-197)
        if (ev_inside(q1.c[2 * i], q1.c[2 * i + 1], q2.c)) { pts[2 * n] = q1.c[2 * i]; pts[2 * n + 1] = q1.c[2 * i + 1]; ++n; }
        if (ev_inside(q2.c[2 * i], q2.c[2 * i + 1], q1.c)) { pts[2 * n] = q2.c[2 * i]; pts[2 * n + 1] = q2.c[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float x, y;
            if (n < 16 && ev_cross(q1.c, q2.c, i, j, x, y)) { pts[2 * n] = x; pts[2 * n + 1] = y; ++n; }
        }
    if (n <= 0) return 0.0;
    // sort_vertex_in_convex_polygon (:33-69)
    float c0 = 0.f, c1 = 0.f;
    for (int i = 0; i < n; ++i) { c0 += pts[2 * i]; c1 += pts[2 * i + 1]; }
    c0 = (float)((double)c0 / (double)n);
    c1 = (float)((double)c1 / (double)n);
    float vs[16];
    for (int i = 0; i < n; ++i) {
        float v0 = pts[2 * i] - c0, v1 = pts[2 * i + 1] - c1;
        const float d = sqrtf(v0 * v0 + v1 * v1);
        v0 = v0 / d;
        v1 = v1 / d;
        if (v1 < 0.f) v0 = (float)(-2.0 - (double)v0);
        vs[i] = v0;
    }
    for (int i = 1; i < n; ++i) {
        if (vs[i - 1] > vs[i]) {
            const float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
            int j = i;
            while (j > 0 && vs[j - 1] > temp) {
                vs[j] = vs[j - 1];
                pts[2 * j] = pts[2 * j - 2];
                pts[2 * j + 1] = pts[2 * j - 1];
                --j;
            }
            vs[j] = temp;
            pts[2 * j] = tx;
            pts[2 * j + 1] = ty;
        }
    }
    double area = 0.0;                                              // area / trangle_area (:17-30)
    for (int i = 0; i < n - 2; ++i) {
        const float *a = pts, *b = pts + 2 * i + 2, *c = pts + 2 * i + 4;
        area += fabs((double)((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0);
    }
    return area;
}

__global__ __launch_bounds__(256) void rotate_iou_eval_kernel(const float *__restrict__ boxes, int N,
                                                              const float *__restrict__ qboxes, int K, int criterion,
                                                              float *__restrict__ iou) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (long long)n * K);
    float rb1[5], rb2[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { rb1[i] = qboxes[(size_t)k * 5 + i]; rb2[i] = boxes[(size_t)n * 5 + i]; }
    const float area1 = rb1[2] * rb1[3], area2 = rb2[2] * rb2[3];
    const double ai = ev_inter(rb1, rb2);
    double r;
    if (criterion == -1) r = ai / ((double)area1 + (double)area2 - ai);
    else if (criterion == 0) r = ai / (double)area1;
    else if (criterion == 1) r = ai / (double)area2;
    else r = ai;
    iou[e] = (float)r;
}

}  // namespace

LIDAR_EXPORT int lidar_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion, float *iou,
                                       void *stream) {
    if (n < 0 || k < 0 || criterion < -1 || criterion > 2) return LIDAR_ERR_ARG;
    if (n == 0 || k == 0) return LIDAR_OK;
    if (!boxes || !query_boxes || !iou) return LIDAR_ERR_ARG;
    const long long pairs = (long long)n * k;
    const long long blocks = (pairs + 255) / 256;
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, boxes, n, query_boxes, k,
                       criterion, iou);
    return lidar_check_launch("lidar_rotate_iou_eval");
}
