/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into or called from the product path.
 *
 * Sequential CPU restatement of the reference's rotated-BEV IoU / NMS algorithm.
 * Follows (arithmetic order preserved, fp32 everywhere, no FMA contraction: build with
 * -ffp-contract=off):
 *   box geometry      /root/reference/pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:14-234
 *                     (== iou3d_cpu.cpp:38-229, the compile-able twin used to pin this file)
 *   axis-aligned IoU  iou3d_nms_kernel.cu:314-325
 *   suppression mask  iou3d_nms_kernel.cu:267-311 (rotated), :328-372 (normal)
 *   greedy keep       iou3d_nms.cpp:116-132
 * Pinned by tests/test_oracle_pins.py against oracle/_ref (the reference's own
 * boxes_iou_bev_cpu compiled unmodified) and against tests/golden/iou3d_*.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS 1e-8f

typedef struct { float x, y; } pt2;

static inline float fmin2(float a, float b) { return a > b ? b : a; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }

/* cross(p1,p2,p0): iou3d_nms_kernel.cu:39-41 */
static inline float cross3(pt2 p1, pt2 p2, pt2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
/* cross(a,b): :35-37 */
static inline float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }

/* check_rect_cross :43-49 */
static inline int rect_cross(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
    return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
           fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* check_in_box2d :51-61 (MARGIN 1e-2, rotation by -heading, strict <) */
static inline int in_box2d(const float *box, pt2 p) {
    const float margin = 1e-2f;
    float cx = box[0], cy = box[1];
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin);
}

/* intersection :63-92; argument order (p1,p0,q1,q0) as at the call site :162 */
static inline int seg_intersect(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *ans) {
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORC_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* rotate_around_center :94-98 */
static inline pt2 rot_center(pt2 c, float ac, float as, pt2 p) {
    pt2 r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

/* box_overlap :104-225.  The reference's vertex buffer is Point[16]; two convex quads give at most
 * 8 crossings + 8 contained corners, so 16 is never exceeded for valid boxes; we size 24 to be safe. */
float orc_box_overlap(const float *a, const float *b) {
    float a_ang = a[6], b_ang = b[6];
    float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
    float ax1 = a[0] - adx, ay1 = a[1] - ady, ax2 = a[0] + adx, ay2 = a[1] + ady;
    float bx1 = b[0] - bdx, by1 = b[1] - bdy, bx2 = b[0] + bdx, by2 = b[1] + bdy;
    pt2 ca = {a[0], a[1]}, cb = {b[0], b[1]};
    pt2 A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
    pt2 B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
    float acs = cosf(a_ang), asn = sinf(a_ang), bcs = cosf(b_ang), bsn = sinf(b_ang);
    for (int k = 0; k < 4; k++) {
        A[k] = rot_center(ca, acs, asn, A[k]);
        B[k] = rot_center(cb, bcs, bsn, B[k]);
    }
    A[4] = A[0];
    B[4] = B[0];

    pt2 v[24];
    pt2 ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (seg_intersect(A[i + 1], A[i], B[j + 1], B[j], &v[cnt])) {
                ctr.x = ctr.x + v[cnt].x;
                ctr.y = ctr.y + v[cnt].y;
                cnt++;
            }
    for (int k = 0; k < 4; k++) {
        if (in_box2d(a, B[k])) {
            ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y;
            v[cnt++] = B[k];
        }
        if (in_box2d(b, A[k])) {
            ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y;
            v[cnt++] = A[k];
        }
    }
    ctr.x /= cnt;   /* cnt==0 -> NaN centre, harmless: loops below are empty (:196-197) */
    ctr.y /= cnt;

    /* bubble sort by atan2 around the centre, strict > (:199-209, point_cmp :100-102) */
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++) {
            float t0 = atan2f(v[i].y - ctr.y, v[i].x - ctr.x);
            float t1 = atan2f(v[i + 1].y - ctr.y, v[i + 1].x - ctr.x);
            if (t0 > t1) { pt2 t = v[i]; v[i] = v[i + 1]; v[i + 1] = t; }
        }

    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        pt2 d0 = {v[k].x - v[0].x, v[k].y - v[0].y};
        pt2 d1 = {v[k + 1].x - v[0].x, v[k + 1].y - v[0].y};
        area += cross2(d0, d1);
    }
    return (float)(fabsf(area) / 2.0);
}

/* iou_bev :227-234 */
float orc_iou_bev(const float *a, const float *b) {
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float s = orc_box_overlap(a, b);
    return s / fmaxf(sa + sb - s, ORC_EPS);
}

/* iou_normal :314-325 */
float orc_iou_normal(const float *a, const float *b) {
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return inter / fmaxf(Sa + Sb - inter, ORC_EPS);
}

/* mode: 0 = overlap area (boxes_overlap_kernel :236-249), 1 = rotated IoU (:251-265),
 *       2 = axis-aligned IoU (iou_normal).  out is (na, nb) row-major. */
void orc_pairwise(const float *A, int na, const float *B, int nb, int mode, float *out) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) {
            const float *a = A + 7 * i, *b = B + 7 * j;
            out[(size_t)i * nb + j] = mode == 0 ? orc_box_overlap(a, b) : mode == 1 ? orc_iou_bev(a, b) : orc_iou_normal(a, b);
        }
}

/* Upper-triangular suppression mask (nms_kernel :267-311).  Words of column blocks left of the
 * row's own block are written as 0 here (the reference computes them, the greedy never reads them). */
void orc_nms_mask(const float *boxes, int n, float thresh, int normal, uint64_t *mask) {
    int cb = (n + 63) / 64;
    memset(mask, 0, (size_t)n * cb * sizeof(uint64_t));
    for (int i = 0; i < n; i++) {
        int rb = i / 64;
        for (int c = rb; c < cb; c++) {
            int cs = n - c * 64 < 64 ? n - c * 64 : 64;
            int start = (c == rb) ? (i % 64) + 1 : 0;
            uint64_t t = 0;
            for (int j = start; j < cs; j++) {
                const float *bj = boxes + 7 * (c * 64 + j);
                float v = normal ? orc_iou_normal(boxes + 7 * i, bj) : orc_iou_bev(boxes + 7 * i, bj);
                if (v > thresh) t |= 1ULL << j;
            }
            mask[(size_t)i * cb + c] = t;
        }
    }
}

/* greedy reduce (iou3d_nms.cpp:116-132): keep gets positions into the (sorted) box array */
int orc_nms_greedy(const uint64_t *mask, int n, int64_t *keep) {
    int cb = (n + 63) / 64;
    uint64_t *remv = (uint64_t *)calloc(cb > 0 ? cb : 1, sizeof(uint64_t));
    int nk = 0;
    for (int i = 0; i < n; i++) {
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[nk++] = i;
            const uint64_t *p = mask + (size_t)i * cb;
            for (int j = nb; j < cb; j++) remv[j] |= p[j];
        }
    }
    free(remv);
    return nk;
}

/* nms_gpu / nms_normal_gpu host entry (iou3d_nms.cpp:90-136, :139-186) on already-sorted boxes */
int orc_nms(const float *boxes, int n, float thresh, int normal, int64_t *keep) {
    int cb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)malloc((size_t)(n > 0 ? n : 1) * (cb > 0 ? cb : 1) * sizeof(uint64_t));
    orc_nms_mask(boxes, n, thresh, normal, mask);
    int nk = orc_nms_greedy(mask, n, keep);
    free(mask);
    return nk;
}
