// Rotated BEV overlap / IoU matrices and rotated / axis-aligned NMS on gfx950.
// Reference: pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu (geometry :14-234, kernels :236-372),
//            pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:49-186 (host entry points; NMS greedy :116-132).
//
// The arithmetic of one box pair follows the reference expression by expression in fp32 (the
// library is built with -ffp-contract=off) so that `iou > thresh` decisions are reproducible.
// What is organised differently for CDNA4:
//   * everything that depends on ONE box (cos/sin of the heading, the 4 rotated corners, area,
//     bounding radius) is computed once per box in a prologue (the reference redoes it per pair);
//   * an exact-zero early-out: boxes whose bounding circles (inflated by the reference's own 1e-2
//     in-box margin) do not touch have no crossings and no contained corners, so the reference
//     returns exactly 0 for them — skipped without evaluating the polygon code;
//   * wave64: one wave owns a 64x64 tile, lane = row, the 64-bit ballot-sized word is the mask
//     word; candidate pairs are compacted per lane before the divergent polygon code runs;
//   * the greedy keep runs on the device (no D2H of the N x N/64 mask, no host loop).
#include "common.h"
#include <math.h>

#define IOU_EPS 1e-8f

struct __attribute__((aligned(16))) BoxPre {
    float cx, cy, hx, hy;     // centre, dx/2, dy/2  (hx,hy as computed by the reference: box[3]/2)
    float c, s;               // cosf(heading), sinf(heading)
    float area, rad;          // dx*dy ; conservative bounding radius incl. margins
    float px[4], py[4];       // rotated corners, reference order
};

struct pt2 { float x, y; };

__device__ __forceinline__ float cross3(pt2 p1, pt2 p2, pt2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// float trig of the heading.  Evaluated in double and rounded once: this is the correctly rounded
// fp32 result except in ~1e-8 of cases, which is what a good host libm returns as well.
__device__ __forceinline__ void heading_cs(float a, float &c, float &s) {
    c = (float)cos((double)a);
    s = (float)sin((double)a);
}

__device__ __forceinline__ BoxPre make_pre(const float *b) {
    BoxPre r;
    r.cx = b[0]; r.cy = b[1];
    r.hx = b[3] / 2; r.hy = b[4] / 2;
    heading_cs(b[6], r.c, r.s);
    r.area = b[3] * b[4];
    const float x1 = r.cx - r.hx, y1 = r.cy - r.hy, x2 = r.cx + r.hx, y2 = r.cy + r.hy;
    const float xs[4] = {x1, x2, x2, x1}, ys[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // rotate_around_center, kernel.cu:94-98
        r.px[k] = (xs[k] - r.cx) * r.c + (ys[k] - r.cy) * (-r.s) + r.cx;
        r.py[k] = (xs[k] - r.cx) * r.s + (ys[k] - r.cy) * r.c + r.cy;
    }
    r.rad = sqrtf(r.hx * r.hx + r.hy * r.hy) * 1.0001f + 0.02f;
    return r;
}

// check_in_box2d (kernel.cu:51-61): cos(-h) == cos(h), sin(-h) == -sin(h) bit for bit
__device__ __forceinline__ bool in_box2d(const BoxPre &B, float px, float py) {
    const float MARGIN = 1e-2f;
    const float ac = B.c, as = -B.s;
    const float rx = (px - B.cx) * ac + (py - B.cy) * (-as);
    const float ry = (px - B.cx) * as + (py - B.cy) * ac;
    return (fabsf(rx) < B.hx + MARGIN) && (fabsf(ry) < B.hy + MARGIN);
}

// intersection (kernel.cu:63-92)
__device__ __forceinline__ bool seg_intersect(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 &ans) {
    const bool rc = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                    fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!rc) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
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

__device__ __forceinline__ bool circles_apart(const BoxPre &A, const BoxPre &B) {
    const float dx = A.cx - B.cx, dy = A.cy - B.cy;
    const float rr = A.rad + B.rad;
    return dx * dx + dy * dy > rr * rr;
}

// Separating-axis test on the two rectangles, each inflated by the reference's in-box margin (and a
// little more for rounding): if an axis of either box separates them, the reference finds no edge
// crossing and no contained corner, i.e. it returns exactly 0 for the pair.
__device__ __forceinline__ bool sat_separated_one(const BoxPre &A, const BoxPre &B) {
    const float m = 0.011f;
    float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = B.px[k] - A.cx, dy = B.py[k] - A.cy;
        const float u = dx * A.c + dy * A.s, v = dy * A.c - dx * A.s;
        umin = fminf(umin, u); umax = fmaxf(umax, u);
        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    const float ex = A.hx * 1.0001f + m, ey = A.hy * 1.0001f + m;
    return (umin > ex) || (umax < -ex) || (vmin > ey) || (vmax < -ey);
}
__device__ __forceinline__ bool sat_separated(const BoxPre &A, const BoxPre &B) {
    return sat_separated_one(A, B) || sat_separated_one(B, A);
}

// Per-thread vertex scratch lives in LDS, laid out [slot][thread] (conflict-free): the polygon vertices
// are APPENDED there (data-dependent count, write-only, no waiting); sorting and the area sum then run
// on registers with statically indexed, predicated steps.  TPB = threads per block of the caller.
template <int TPB>
struct VertScratch {
    float x[16][TPB], y[16][TPB];
};

// the reference's bubble sort by atan2 around the centre (strict >, kernel.cu:199-209) and shoelace
// fan (:218-224), as the same sequence of compare/swap decisions on M register slots
template <int M, int TPB>
__device__ __forceinline__ float polygon_area_sorted(const VertScratch<TPB> &S, int t, int cnt, float ctrx, float ctry) {
    float vx[M], vy[M], va[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        vx[k] = S.x[k][t];
        vy[k] = S.y[k][t];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) va[k] = atan2f(vy[k] - ctry, vx[k] - ctrx);
#pragma unroll
    for (int j = 0; j < M - 1; ++j) {
#pragma unroll
        for (int i = 0; i < M - 1 - j; ++i) {
            const bool sw = (i < cnt - j - 1) && (va[i] > va[i + 1]);
            const float ax = vx[i], ay = vy[i], aa = va[i];
            vx[i] = sw ? vx[i + 1] : ax; vy[i] = sw ? vy[i + 1] : ay; va[i] = sw ? va[i + 1] : aa;
            vx[i + 1] = sw ? ax : vx[i + 1]; vy[i + 1] = sw ? ay : vy[i + 1]; va[i + 1] = sw ? aa : va[i + 1];
        }
    }
    float area = 0.f;
#pragma unroll
    for (int k = 0; k < M - 1; ++k) {
        const float ax = vx[k] - vx[0], ay = vy[k] - vy[0];
        const float bx = vx[k + 1] - vx[0], by = vy[k + 1] - vy[0];
        const float term = ax * by - ay * bx;
        if (k < cnt - 1) area += term;
    }
    return fabsf(area) * 0.5f;
}

// box_overlap (kernel.cu:104-225) on pre-computed boxes.  Polygon of at most 16 vertices.
template <int TPB>
__device__ float box_overlap_pre(const BoxPre &A, const BoxPre &B, VertScratch<TPB> &S, int t) {
    int cnt = 0;
    float sumx = 0.f, sumy = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const pt2 a0 = {A.px[i], A.py[i]}, a1 = {A.px[(i + 1) & 3], A.py[(i + 1) & 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const pt2 b0 = {B.px[j], B.py[j]}, b1 = {B.px[(j + 1) & 3], B.py[(j + 1) & 3]};
            pt2 ans;
            if (seg_intersect(a1, a0, b1, b0, ans)) {
                sumx = sumx + ans.x;
                sumy = sumy + ans.y;
                if (cnt < 16) { S.x[cnt][t] = ans.x; S.y[cnt][t] = ans.y; }
                cnt++;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(A, B.px[k], B.py[k])) {
            sumx = sumx + B.px[k]; sumy = sumy + B.py[k];
            if (cnt < 16) { S.x[cnt][t] = B.px[k]; S.y[cnt][t] = B.py[k]; }
            cnt++;
        }
        if (in_box2d(B, A.px[k], A.py[k])) {
            sumx = sumx + A.px[k]; sumy = sumy + A.py[k];
            if (cnt < 16) { S.x[cnt][t] = A.px[k]; S.y[cnt][t] = A.py[k]; }
            cnt++;
        }
    }
    if (cnt == 0) return 0.f;
    if (cnt > 16) cnt = 16;  // the reference's buffer is Point[16]; unreachable for convex quads
    const float ctrx = sumx / cnt, ctry = sumy / cnt;
    if (cnt <= 8) return polygon_area_sorted<8, TPB>(S, t, cnt, ctrx, ctry);
    return polygon_area_sorted<16, TPB>(S, t, cnt, ctrx, ctry);
}

__device__ __forceinline__ float iou_from_overlap(float sa, float sb, float s) {
    return s / fmaxf(sa + sb - s, IOU_EPS);
}

// iou_normal (kernel.cu:314-325)
__device__ __forceinline__ float iou_normal_dev(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = width * height;
    const float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return interS / fmaxf(Sa + Sb - interS, IOU_EPS);
}

// ------------------------------------------------------------------ prologue
__global__ void iou_prep_kernel(const float *__restrict__ boxes, int n, BoxPre *__restrict__ pre) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pre[i] = make_pre(boxes + (size_t)i * 7);
}

// ------------------------------------------------------------------ pairwise matrices
// mode 0: overlap area (boxes_overlap_kernel :236-249); 1: IoU (boxes_iou_bev_kernel :251-265)
// block = 256 threads = 4 rows x 64 cols of the (N, M) matrix; cols on lanes -> coalesced stores.
__global__ __launch_bounds__(256) void iou_pairwise_kernel(const BoxPre *__restrict__ pa, int na,
                                                           const BoxPre *__restrict__ pb, int nb, int mode,
                                                           float *__restrict__ out) {
    __shared__ VertScratch<256> S;
    const int t = threadIdx.x;
    const int col = blockIdx.x * 64 + (t & 63);
    const int row = blockIdx.y * 4 + (t >> 6);
    if (row >= na || col >= nb) return;
    const BoxPre A = pa[row];
    const BoxPre B = pb[col];
    float v = 0.f;
    if (!circles_apart(A, B)) {
        const float s = box_overlap_pre<256>(A, B, S, t);
        v = mode == 0 ? s : iou_from_overlap(A.area, B.area, s);
    }
    out[(size_t)row * nb + col] = v;
}

// ------------------------------------------------------------------ NMS suppression mask
// grid = (col_block, ceil(row_blocks/4), frame), 256 threads: the 4 waves of a block own 4 row blocks
// against one 64-box column block (shared in LDS); wave = one 64x64 tile, lane = row.  Only tiles with
// col_block >= row_block are computed (the greedy never reads the others).  mask is (n, cb) u64.
// Rotated IoU per tile:
//   1. bounding-circle test for all 64x64 pairs -> per-lane candidate bits -> compacted pair list (LDS)
//   2. separating-axis test on the dense pair list; survivors compacted in place
//   3. polygon intersection, one surviving pair per lane (32 lanes at a time, vertex scratch in LDS)
#define NMS_PAIR_CAP 1024
struct NmsWaveLds {
    unsigned short pairs[NMS_PAIR_CAP];
    unsigned long long word[64];
    VertScratch<32> S;
};

template <bool NORMAL>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float *__restrict__ boxes_all, const BoxPre *__restrict__ pre_all,
                                                       const int *__restrict__ counts, int n_max, float thresh,
                                                       unsigned long long *__restrict__ mask_all, int n_lim, const int *__restrict__ gate) {
    __shared__ BoxPre s_pre[64];       // column boxes (rotated)   | NORMAL: raw boxes in the same bytes
    __shared__ NmsWaveLds s_w[4];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int rb = blockIdx.y * 4 + wv, cbk = blockIdx.x, f = blockIdx.z;
    if (gate && gate[f] == 0) return;                          // second stage of a limited call: only frames that still need boxes
    const int n = min(counts ? min(counts[f], n_max) : n_max, n_lim);
    const int cb_total = (n_max + 63) / 64;
    const float *boxes = boxes_all + (size_t)f * n_max * 7;
    const BoxPre *pre = pre_all + (size_t)f * n_max;
    unsigned long long *mask = mask_all + (size_t)f * n_max * cb_total;
    const int colbase = cbk * 64;
    const int col_size = min(n - colbase, 64);
    const bool block_live = (cbk >= (int)blockIdx.y * 4) && (colbase < n) && ((int)blockIdx.y * 256 < n);
    if (!block_live) return;                                   // block-uniform
    float *s_raw = reinterpret_cast<float *>(s_pre);
    if (NORMAL) {
        for (int k = threadIdx.x; k < col_size * 7; k += 256) s_raw[k] = boxes[(size_t)colbase * 7 + k];
    } else {
        if (threadIdx.x < col_size) s_pre[threadIdx.x] = pre[colbase + threadIdx.x];
    }
    __syncthreads();
    if (cbk < rb || rb * 64 >= n) return;                      // wave-uniform; no block barrier below
    const int row = rb * 64 + l;
    const bool rvalid = row < n;
    const int start = (rb == cbk) ? l + 1 : 0;
    unsigned long long word = 0ull;
    if (NORMAL) {
        if (rvalid) {
            float a[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) a[k] = boxes[(size_t)row * 7 + k];
            for (int j = start; j < col_size; ++j)
                if (iou_normal_dev(a, s_raw + j * 7) > thresh) word |= 1ull << j;
        }
    } else if (thresh < 0.f) {
        // every pair is a hit, disjoint ones included (IoU == 0 > thresh)
        const unsigned long long all = col_size >= 64 ? ~0ull : ((1ull << col_size) - 1ull);
        word = (start >= 64) ? 0ull : (all & ~((1ull << start) - 1ull));
    } else {
        NmsWaveLds &W = s_w[wv];
        const BoxPre A = pre[rvalid ? row : rb * 64];
        // 1. circle test, lane = row
        unsigned long long cand = 0ull;
        if (rvalid) {
#pragma unroll 16
            for (int j = 0; j < 64; ++j) {   // fixed trip count: the LDS broadcasts pipeline
                const bool hit = !circles_apart(A, s_pre[j < col_size ? j : 0]);
                if (hit && j >= start && j < col_size) cand |= 1ull << j;
            }
        }
        const int total_all = wave_sum(__popcll(cand));
        W.word[l] = 0ull;
        // The pair list holds NMS_PAIR_CAP entries.  A tile with more candidates (most of its 4096 pairs nearly coincide:
        // a row of parked cars, a few objects with thousands of proposals each) goes through the SAME three steps in four
        // rounds of 16 rows (16 x 64 pairs always fit) — no per-lane special path, no lanes sharing a scratch column.
        const int ngrp = (total_all > NMS_PAIR_CAP) ? 4 : 1;           // wave-uniform
        for (int grp = 0; grp < ngrp; ++grp) {
            unsigned long long cnd = (ngrp == 1 || (l >> 4) == grp) ? cand : 0ull;
            const int myc = __popcll(cnd);
            const int incl = wave_incl_scan(myc);
            const int total = __shfl(incl, 63, 64);
            int pos = incl - myc;
            while (cnd) {
                const int j = __ffsll((long long)cnd) - 1;
                cnd &= cnd - 1ull;
                W.pairs[pos++] = (unsigned short)((l << 6) | j);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 2. separating-axis test on the dense list, survivors compacted in place
            int total2 = 0;
            for (int e0 = 0; e0 < total; e0 += 64) {
                const int e = e0 + l;
                unsigned short pr = 0;
                bool keepit = false;
                if (e < total) {
                    pr = W.pairs[e];
                    keepit = !sat_separated(pre[rb * 64 + (pr >> 6)], s_pre[pr & 63]);
                }
                const unsigned long long bal = __ballot(keepit);
                __builtin_amdgcn_wave_barrier();
                if (keepit) W.pairs[total2 + __popcll(bal & lanemask_lt())] = pr;
                total2 += __popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 3. polygon intersection: 32 surviving pairs per pass, one per lane of the lower half-wave
            for (int e0 = 0; e0 < total2; e0 += 32) {
                const int e = e0 + l;
                if (l < 32 && e < total2) {
                    const unsigned short pr = W.pairs[e];
                    const int i = pr >> 6, j = pr & 63;
                    const BoxPre Ar = pre[rb * 64 + i], Bc = s_pre[j];
                    const float sov = box_overlap_pre<32>(Ar, Bc, W.S, l);
                    if (iou_from_overlap(Ar.area, Bc.area, sov) > thresh) atomicOr(&W.word[i], 1ull << j);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();       // also: the next round re-uses the pair list
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        word = W.word[l];
    }
    if (rvalid) mask[(size_t)row * cb_total + cbk] = word;
}

// ------------------------------------------------------------------ device-side greedy keep
// one block per frame (iou3d_nms.cpp:116-132).  keep: int64 positions; num_keep: int per frame.
#define GREEDY_TPB 1024
__global__ __launch_bounds__(GREEDY_TPB) void nms_greedy_kernel(const unsigned long long *__restrict__ mask_all,
                                                                const int *__restrict__ counts, int n_max, int max_keep,
                                                                long long *__restrict__ keep_all,
                                                                int *__restrict__ num_keep) {
    __shared__ unsigned long long s_remv[1024];  // cb words (n_max <= 65536)
    __shared__ unsigned long long s_keepmask;
    __shared__ int s_nkeep;
    const int f = blockIdx.x;
    const int n = counts ? min(counts[f], n_max) : n_max;
    const int cb_total = (n_max + 63) / 64;
    const int cb = (n + 63) / 64;
    const unsigned long long *mask = mask_all + (size_t)f * n_max * cb_total;
    long long *keep = keep_all + (size_t)f * n_max;
    const int t = threadIdx.x, l = t & 63;
    for (int c = t; c < cb; c += GREEDY_TPB) s_remv[c] = 0ull;
    if (t == 0) s_nkeep = 0;
    __syncthreads();
    for (int rb = 0; rb < cb; ++rb) {
        const int row0 = rb * 64;
        if (t < 64) {
            const int row = row0 + l;
            const bool valid = row < n;
            const unsigned long long diag = valid ? mask[(size_t)row * cb_total + rb] : 0ull;
            unsigned long long r = s_remv[rb];
            unsigned long long km = 0ull;
            const int lim = min(64, n - row0);
            for (int i = 0; i < lim; ++i) {
                const unsigned long long d = __shfl(diag, i, 64);
                if (!((r >> i) & 1ull)) {
                    km |= 1ull << i;
                    r |= d;
                }
            }
            const int base = s_nkeep;
            if ((km >> l) & 1ull) keep[base + __popcll(km & lanemask_lt())] = row;
            if (l == 0) {
                s_keepmask = km;
                s_nkeep = base + __popcll(km);
            }
        }
        __syncthreads();
        const unsigned long long km = s_keepmask;
        for (int c = rb + 1 + t; c < cb; c += GREEDY_TPB) {
            unsigned long long acc = 0ull;
            unsigned long long m = km;
            while (m) {
                const int i = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                acc |= mask[(size_t)(row0 + i) * cb_total + c];
            }
            s_remv[c] |= acc;
        }
        __syncthreads();
    }
    if (t == 0) num_keep[f] = min(s_nkeep, max_keep);
}

// ------------------------------------------------------------------ greedy keep, N <= 4096 fast path
// One block per frame.  All 256 threads stream the mask one 64-row block at a time into LDS (two LDS buffers fed from a
// three-deep register ring, coalesced, independent of the decisions), wave 0 walks the decision chain on LDS data only;
// the walk stops once max_keep boxes are kept (the caller's NMS_POST_MAXSIZE):
//   lane c keeps remv word c in a register; the 64x64 diagonal tile is resolved by jumping from kept
//   box to kept box (ctz over the not-yet-suppressed bits, v_readlane of the row's diagonal word);
//   the kept rows are then OR-ed into remv from LDS (one conflict-free 512-B row read each).
#define GF_TPB 256
__global__ __launch_bounds__(GF_TPB) void nms_greedy_fast_kernel(const unsigned long long *__restrict__ mask_all,
                                                                 const int *__restrict__ counts, int n_max, int max_keep,
                                                                 long long *__restrict__ keep_all,
                                                                 int *__restrict__ num_keep, int n_lim, const int *__restrict__ gate,
                                                                 int *__restrict__ need_more) {
    __shared__ unsigned long long s_tile[2][64][64];
    __shared__ int s_stop;
    const int f = blockIdx.x;
    if (gate && gate[f] == 0) return;
    const int n_all = counts ? min(counts[f], n_max) : n_max;
    const int n = min(n_all, n_lim);
    const int cb_total = (n_max + 63) / 64;   // <= 64
    const int cb = (n + 63) / 64;
    const unsigned long long *mask = mask_all + (size_t)f * n_max * cb_total;
    long long *keep = keep_all + (size_t)f * n_max;
    const int t = threadIdx.x, l = t & 63;
    // register ring: the 64-row blocks rb+1 and rb+2 are in flight while block rb (already in LDS) is decided, so a block's
    // global-load latency is spread over two decision steps instead of sitting in front of every step
    unsigned long long pre[3][16];
    auto fetch = [&](int rb, unsigned long long (&dst)[16]) {  // rows rb*64 .. +63, columns 0..cb_total-1 -> registers
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = t + GF_TPB * k, row = idx >> 6, col = idx & 63;
            const int grow = rb * 64 + row;
            // unconditional load from a clamped address + select: a guarded load compiles to a branch with a full
            // s_waitcnt per load, which serialises the 16 loads of a block
            const unsigned long long v = mask[(size_t)min(grow, max(n - 1, 0)) * cb_total + min(col, cb_total - 1)];
            dst[k] = (rb < cb && grow < n && col < cb_total && col >= rb) ? v : 0ull;
        }
    };
    auto stash = [&](int buf, const unsigned long long (&src)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = t + GF_TPB * k;
            s_tile[buf][idx >> 6][idx & 63] = src[k];
        }
    };
    fetch(0, pre[0]);
    fetch(1, pre[1]);
    fetch(2, pre[2]);
    stash(0, pre[0]);
    __syncthreads();
    unsigned long long remv = 0ull;  // wave 0: lane c owns column block c
    int nkeep = 0;
    for (int rb0 = 0; rb0 < cb; rb0 += 3) {
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rb = rb0 + ph;
            if (rb >= cb) break;                   // block-uniform
            const int buf = rb & 1;
            fetch(rb + 3, pre[ph]);                // slot ph held block rb (stashed one step ago)
            if (t < 64) {
                const int row0 = rb * 64;
                const int lim = min(64, n - row0);
                const unsigned long long valid = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
                const unsigned long long diag = s_tile[buf][l][rb];   // lane i: diagonal word of row i
                const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
                unsigned long long r = __shfl(remv, rb, 64);
                r = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(r >> 32)) << 32) |
                    (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)r);
                unsigned long long km = 0ull;
                unsigned long long avail = ~r & valid;
                while (avail) {   // wave-uniform: one iteration per KEPT box of this block
                    const int i = __ffsll((long long)avail) - 1;
                    km |= 1ull << i;
                    const unsigned long long d =
                        ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                        (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, i);
                    r |= d | (1ull << i);
                    avail = ~r & valid & ~((2ull << i) - 1ull);
                }
                if ((km >> l) & 1ull) keep[nkeep + __popcll(km & lanemask_lt())] = row0 + l;
                nkeep += __popcll(km);
                unsigned long long acc = 0ull, m = km;
                while (m) {
                    const int i = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    acc |= s_tile[buf][i][l];
                }
                if (l > rb) remv |= acc;
                if (l == 0) s_stop = (nkeep >= max_keep) ? 1 : 0;   // the caller only wants the first max_keep survivors
            }
            __syncthreads();          // everyone is done reading the buffer block rb+1 goes into
            if (s_stop) { rb0 = cb; break; }   // block-uniform
            stash(buf ^ 1, pre[(ph + 1) % 3]);
            __syncthreads();
        }
    }
    if (t == 0) {
        num_keep[f] = min(nkeep, max_keep);
        // first stage of a limited call: the greedy order is a prefix property, so max_keep survivors among the first n_lim boxes ARE
        // the answer; a frame that ran out of boxes first asks for the full pass
        if (need_more) need_more[f] = (nkeep < max_keep && n_all > n_lim) ? 1 : 0;
    }
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_iou_workspace_bytes(int n_a, int n_b) {
    return align_up((size_t)(n_a > 0 ? n_a : 1) * sizeof(BoxPre), 256) + align_up((size_t)(n_b > 0 ? n_b : 1) * sizeof(BoxPre), 256);
}

// mode 0: boxes_overlap_bev_gpu (iou3d_nms.cpp:49-68); mode 1: boxes_iou_bev_gpu (:70-88)
LIDAR_EXPORT int lidar_boxes_pairwise_bev(const float *boxes_a, int n_a, const float *boxes_b, int n_b, int mode,
                                          float *out, void *ws, size_t ws_bytes, void *stream) {
    if (n_a < 0 || n_b < 0 || (mode != 0 && mode != 1)) return LIDAR_ERR_ARG;
    if (n_a == 0 || n_b == 0) return LIDAR_OK;
    if (!boxes_a || !boxes_b || !out || !ws) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_iou_workspace_bytes(n_a, n_b)) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    BoxPre *pa = (BoxPre *)ws;
    BoxPre *pb = (BoxPre *)((char *)ws + align_up((size_t)n_a * sizeof(BoxPre), 256));
    hipLaunchKernelGGL(iou_prep_kernel, dim3(divup(n_a, 256)), dim3(256), 0, s, boxes_a, n_a, pa);
    hipLaunchKernelGGL(iou_prep_kernel, dim3(divup(n_b, 256)), dim3(256), 0, s, boxes_b, n_b, pb);
    hipLaunchKernelGGL(iou_pairwise_kernel, dim3(divup(n_b, 64), divup(n_a, 4)), dim3(256), 0, s, pa, n_a, pb, n_b,
                       mode, out);
    return lidar_check_launch("lidar_boxes_pairwise_bev");
}

LIDAR_EXPORT size_t lidar_nms_workspace_bytes(int batch, int n_max) {
    if (batch <= 0 || n_max <= 0) return 256;
    const size_t cb = (size_t)(n_max + 63) / 64;
    return align_up((size_t)batch * n_max * sizeof(BoxPre), 256) + align_up((size_t)batch * n_max * cb * 8, 256) +
           align_up((size_t)batch * sizeof(int), 256);          // + the limited call's per-frame "needs the full pass" flags
}

// Batched NMS on already score-sorted boxes.  boxes (batch, n_max, 7); counts (batch) device ints or
// NULL (= n_max for every frame); keep (batch, n_max) int64 device; num_keep (batch) int device.
// normal = 0: nms_gpu (iou3d_nms.cpp:90-136), 1: nms_normal_gpu (:139-186).
LIDAR_EXPORT int lidar_nms_batch_limited(const float *boxes, const int *counts, int batch, int n_max, float thresh, int normal,
                                         int max_keep, long long *keep, int *num_keep, void *ws, size_t ws_bytes, void *stream) {
    if (batch <= 0 || n_max < 0 || !num_keep || max_keep <= 0) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_max == 0) return hipMemsetAsync(num_keep, 0, sizeof(int) * batch, s) == hipSuccess ? LIDAR_OK : LIDAR_ERR_LAUNCH;
    if (!boxes || !keep || !ws) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_nms_workspace_bytes(batch, n_max)) return LIDAR_ERR_WORKSPACE;
    const int cb = (n_max + 63) / 64;
    if (cb > 1024) return LIDAR_ERR_ARG;  // greedy keeps remv[] in LDS
    BoxPre *pre = (BoxPre *)ws;
    unsigned long long *mask = (unsigned long long *)((char *)ws + align_up((size_t)batch * n_max * sizeof(BoxPre), 256));
    if (!normal)
        hipLaunchKernelGGL(iou_prep_kernel, dim3(divup((long long)batch * n_max, 256)), dim3(256), 0, s, boxes,
                           batch * n_max, pre);
    // Speculative two-stage form when the caller wants only the first max_keep survivors (NMS_POST_MAXSIZE, model_nms_utils.py:19-21):
    // stage 1 builds the mask of, and runs the greedy pass over, the first n1 candidates only — max_keep survivors among them are the
    // exact answer (greedy NMS is a prefix property); a frame that ran out of candidates first raises its flag and stage 2 — the full
    // mask + greedy, whose workgroups exit at once for every other frame — redoes exactly that frame.  n1 = 2 max_keep rounded up to
    // 256 (bench frames, tools/nms_keep_pos.py: the 500th survivor is candidate 589-633 of 4 096, so the mask shrinks 16 x; a
    // trained model's frames hold fewer candidates than n1 after the score threshold and never reach the second stage either).
    const int n1 = (cb <= 64 && max_keep < n_max) ? ((2 * max_keep + 255) / 256) * 256 : n_max;
    const bool two_stage = n1 < n_max;
    int *need_more = two_stage ? (int *)((char *)mask + align_up((size_t)batch * n_max * cb * sizeof(unsigned long long), 256)) : nullptr;
    const int cb1 = two_stage ? (n1 + 63) / 64 : cb;
    const dim3 grid1(cb1, divup(cb1, 4), batch), grid(cb, divup(cb, 4), batch);
    if (normal)
        hipLaunchKernelGGL(nms_mask_kernel<true>, grid1, dim3(256), 0, s, boxes, pre, counts, n_max, thresh, mask, two_stage ? n1 : n_max, (const int *)nullptr);
    else
        hipLaunchKernelGGL(nms_mask_kernel<false>, grid1, dim3(256), 0, s, boxes, pre, counts, n_max, thresh, mask, two_stage ? n1 : n_max, (const int *)nullptr);
    if (cb <= 64)
        hipLaunchKernelGGL(nms_greedy_fast_kernel, dim3(batch), dim3(GF_TPB), 0, s, mask, counts, n_max, max_keep, keep, num_keep,
                           two_stage ? n1 : n_max, (const int *)nullptr, need_more);
    else
        hipLaunchKernelGGL(nms_greedy_kernel, dim3(batch), dim3(GREEDY_TPB), 0, s, mask, counts, n_max, max_keep, keep, num_keep);
    if (two_stage) {
        if (normal)
            hipLaunchKernelGGL(nms_mask_kernel<true>, grid, dim3(256), 0, s, boxes, pre, counts, n_max, thresh, mask, n_max, (const int *)need_more);
        else
            hipLaunchKernelGGL(nms_mask_kernel<false>, grid, dim3(256), 0, s, boxes, pre, counts, n_max, thresh, mask, n_max, (const int *)need_more);
        hipLaunchKernelGGL(nms_greedy_fast_kernel, dim3(batch), dim3(GF_TPB), 0, s, mask, counts, n_max, max_keep, keep, num_keep, n_max,
                           (const int *)need_more, (int *)nullptr);
    }
    return lidar_check_launch("lidar_nms_batch");
}

LIDAR_EXPORT int lidar_nms_batch(const float *boxes, const int *counts, int batch, int n_max, float thresh, int normal,
                                 long long *keep, int *num_keep, void *ws, size_t ws_bytes, void *stream) {
    return lidar_nms_batch_limited(boxes, counts, batch, n_max, thresh, normal, n_max > 0 ? n_max : 1, keep, num_keep, ws, ws_bytes,
                                   stream);
}

// test hook: copy of the suppression mask words (row-major (n_max, cb)) of frame 0 left in ws by
// the last lidar_nms_batch call with the same (batch, n_max).
LIDAR_EXPORT const void *lidar_nms_mask_ptr(void *ws, int batch, int n_max) {
    return (char *)ws + align_up((size_t)batch * n_max * sizeof(BoxPre), 256);
}
