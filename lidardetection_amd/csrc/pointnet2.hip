// PointNet++ set-abstraction primitives on gfx950 (stacked-batch and dense-batch layouts).
// Reference: pcdet/ops/pointnet2/pointnet2_stack/src/*.cu and pcdet/ops/pointnet2/pointnet2_batch/src/*.cu
// (per-kernel citations at each entry point below).  Integer outputs (ball-query / FPS / 3-NN indices)
// follow the reference's sequential scan order exactly; the distance expression is evaluated as
// written, (dx*dx + dy*dy) + dz*dz in fp32 without contraction.
//
// CDNA4 organisation:
//   * brute-force searches (ball query, 3-NN) stage the candidate points through LDS tiles that the whole
//     workgroup streams coalesced once, instead of every thread re-reading them from global memory; every
//     lane then reads the same LDS address (broadcast, conflict-free) while walking the tile in index order;
//   * FPS keeps each sample's points and running min-distances in registers (<= 20480 points per sample),
//     one barrier per round: per-wave argmax by DPP/shuffle, 16 partials in double-buffered LDS, every wave
//     reduces them redundantly; the winner's coordinates travel with the partial (no global re-read);
//   * grouping transposes (sample-major rows -> channel-major output) through an LDS tile so both the
//     gathered feature rows and the (M, C, nsample) output are accessed with full 256-B wave transactions;
//   * scatter-add backward passes issue float atomics as contiguous per-wave row segments.
#include "common.h"
#include <algorithm>
#include <type_traits>

#define PN_TPB 256
#define PN_TILE 1024

__device__ __forceinline__ float pn_dist2(float ax, float ay, float az, float x, float y, float z) {
    return (ax - x) * (ax - x) + (ay - y) * (ay - y) + (az - z) * (az - z);
}

// batch of stacked element `i` given per-batch counts; also the exclusive start of that batch in
// a second stacked array (reference: the cumulative-count walk at the top of every *_stack kernel)
__device__ __forceinline__ void pn_batch_of(const int *__restrict__ cnt, int B, int i, const int *__restrict__ other,
                                            int &bs, int &start_other, int &n_other) {
    int b = 0, acc = cnt[0];
    for (int k = 1; k < B; ++k) {
        if (i < acc) break;
        acc += cnt[k];
        b = k;
    }
    int s = 0;
    for (int k = 0; k < b; ++k) s += other[k];
    bs = b;
    start_other = s;
    n_other = other[b];
}

// ------------------------------------------------------------------ ball query
// STACK: ball_query_kernel_stack (pointnet2_stack/src/ball_query_gpu.cu:16-66), -1 sentinel for empty balls.
// !STACK: ball_query_kernel_fast (pointnet2_batch/src/ball_query_gpu.cu:15-51), dense (b, m, .) layout.
// The reference walks the candidates of one query serially in one thread.  Here a WAVE owns a query: its 64 lanes test 64
// consecutive candidates at a time (LDS tile, structure-of-arrays), a ballot + prefix count appends the hits in candidate
// order, so the result is the reference's "first nsample in index order" exactly, with 64x the parallelism per query
// (PV-RCNN has only 2 048 keypoints per frame: one thread per query leaves the GPU 1 % occupied).
#define BQ_QPW 4                    // queries per wave, processed one after the other against each staged tile (2/4/8 measured: 4 best)
#define BQ_QPB (4 * BQ_QPW)         // queries per 256-thread workgroup
// NR radii in one pass (NR = 1 or 2): a multi-scale set-abstraction layer queries the same (centre, candidate) pairs once per
// radius; here each squared distance is computed once and tested against every radius, each with its own hit list.
template <bool STACK, int NR>
__global__ __launch_bounds__(PN_TPB) void ball_query_kernel(int B, int M, int N, float radius_a, int nsample_a, float radius_b,
                                                            int nsample_b, const float *__restrict__ new_xyz,
                                                            const int *__restrict__ new_cnt, const float *__restrict__ xyz,
                                                            const int *__restrict__ xyz_cnt, int *__restrict__ idx_a,
                                                            int *__restrict__ idx_b) {
    __shared__ float s_x[PN_TILE], s_y[PN_TILE], s_z[PN_TILE];
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int q0 = blockIdx.x * BQ_QPB;                       // first query (within the batch element for !STACK) of this block
    const size_t qbase = STACK ? 0 : (size_t)blockIdx.y * M;  // row offset of the batch element in new_xyz / idx
    const float r2[2] = {radius_a * radius_a, radius_b * radius_b};
    const int ns[2] = {nsample_a, nsample_b};
    int *const out[2] = {idx_a, idx_b};
    // per-query state of this wave (wave-uniform)
    int qb[BQ_QPW], cnt[BQ_QPW][NR], first[BQ_QPW][NR];
    float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
    int b_lo = 0x7fffffff, b_hi = -1;
#pragma unroll
    for (int i = 0; i < BQ_QPW; ++i) {
        const int q = q0 + wv * BQ_QPW + i;
#pragma unroll
        for (int r = 0; r < NR; ++r) { cnt[i][r] = 0; first[i][r] = 0; }
        qb[i] = -1;
        qx[i] = qy[i] = qz[i] = 0.f;
        if (q < M) {
            if (STACK) {
                int s0, n0;
                pn_batch_of(new_cnt, B, q, xyz_cnt, qb[i], s0, n0);
            } else {
                qb[i] = blockIdx.y;
            }
            qx[i] = new_xyz[(qbase + q) * 3 + 0];
            qy[i] = new_xyz[(qbase + q) * 3 + 1];
            qz[i] = new_xyz[(qbase + q) * 3 + 2];
        }
    }
    {   // batch elements covered by the block's queries (block-uniform; usually one)
        const int qa = q0, qe = min(q0 + BQ_QPB, M) - 1;
        if (STACK) {
            int s0, n0;
            pn_batch_of(new_cnt, B, qa, xyz_cnt, b_lo, s0, n0);
            pn_batch_of(new_cnt, B, qe, xyz_cnt, b_hi, s0, n0);
        } else {
            b_lo = b_hi = blockIdx.y;
        }
    }
    auto open_lists = [&](int i) {                            // does query i still collect hits for some radius?
        bool o = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) o = o || cnt[i][r] < ns[r];
        return o;
    };
    for (int bb = b_lo; bb <= b_hi; ++bb) {
        int bstart, bn;
        if (STACK) {
            bstart = 0;
            for (int k = 0; k < bb; ++k) bstart += xyz_cnt[k];
            bn = xyz_cnt[bb];
        } else {
            bstart = bb * N;
            bn = N;
        }
        for (int t0 = 0; t0 < bn; t0 += PN_TILE) {
            const int tn = min(PN_TILE, bn - t0);
            bool wave_busy = false;
#pragma unroll
            for (int i = 0; i < BQ_QPW; ++i) wave_busy = wave_busy || (qb[i] == bb && open_lists(i));
            if (__syncthreads_or(wave_busy) == 0) break;      // every query of the block that uses this batch element is full
            for (int k = t; k < tn * 3; k += PN_TPB) {
                const float v = xyz[((size_t)bstart + t0) * 3 + k];
                const int p = k / 3, c = k - p * 3;
                (c == 0 ? s_x : (c == 1 ? s_y : s_z))[p] = v;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < BQ_QPW; ++i) {
                if (qb[i] != bb || !open_lists(i)) continue;                  // wave-uniform
                const size_t qrow = qbase + q0 + wv * BQ_QPW + i;
                for (int k0 = 0; k0 < tn; k0 += 64) {
                    const int k = k0 + l;
                    const int kc = min(k, tn - 1);
                    const float d2 = pn_dist2(qx[i], qy[i], qz[i], s_x[kc], s_y[kc], s_z[kc]);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if (cnt[i][r] >= ns[r]) continue;                      // wave-uniform
                        const bool hit = (k < tn) && (d2 < r2[r]);
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            if (cnt[i][r] == 0) first[i][r] = t0 + k0 + __builtin_ctzll(bal);
                            const int pos = cnt[i][r] + __popcll(bal & lanemask_lt());
                            if (hit && pos < ns[r]) out[r][qrow * (size_t)ns[r] + pos] = t0 + k;
                            cnt[i][r] += __popcll(bal);
                        }
                    }
                    if (!open_lists(i)) break;
                }
            }
        }
    }
    // the reference pre-fills all nsample slots with the first hit: slots past the last hit keep it
#pragma unroll
    for (int i = 0; i < BQ_QPW; ++i) {
        const int q = q0 + wv * BQ_QPW + i;
        if (q >= M) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int *o = out[r] + (qbase + q) * (size_t)ns[r];
            if (cnt[i][r] == 0) {
                if (STACK && l == 0) o[0] = -1;
            } else {
                for (int p = cnt[i][r] + l; p < ns[r]; p += 64) o[p] = first[i][r];
            }
        }
    }
}

LIDAR_EXPORT int lidar_ball_query_stack(int B, int M, float radius, int nsample, const float *new_xyz,
                                        const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt,
                                        int *idx, void *stream) {
    if (B <= 0 || M < 0 || nsample <= 0) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!new_xyz || !new_xyz_batch_cnt || !xyz || !xyz_batch_cnt || !idx) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL((ball_query_kernel<true, 1>), dim3(divup(M, BQ_QPB)), dim3(PN_TPB), 0, (hipStream_t)stream, B, M, 0,
                       radius, nsample, 0.f, 0, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx, (int *)nullptr);
    return lidar_check_launch("lidar_ball_query_stack");
}

// two radii over the same centres and candidates in one pass (StackSAModuleMSG's scales): idx_a (M, nsample_a), idx_b (M, nsample_b),
// each exactly what lidar_ball_query_stack returns for its radius
LIDAR_EXPORT int lidar_ball_query_stack2(int B, int M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                         const float *new_xyz, const int *new_xyz_batch_cnt, const float *xyz,
                                         const int *xyz_batch_cnt, int *idx_a, int *idx_b, void *stream) {
    if (B <= 0 || M < 0 || nsample_a <= 0 || nsample_b <= 0) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!new_xyz || !new_xyz_batch_cnt || !xyz || !xyz_batch_cnt || !idx_a || !idx_b) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL((ball_query_kernel<true, 2>), dim3(divup(M, BQ_QPB)), dim3(PN_TPB), 0, (hipStream_t)stream, B, M, 0,
                       radius_a, nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b);
    return lidar_check_launch("lidar_ball_query_stack2");
}

// ------------------------------------------------------------------ ball query through a cell grid (stack layout)
// The kernels above test every (centre, candidate) pair of a batch element: 20 000 tests per centre where a 0.4 - 1.6 m ball
// holds a few dozen points.  Here the candidates of each batch element are first binned into an x / y grid whose cells are at
// least one (largest) radius wide — one workgroup per element: bounding box, LDS histogram, scan, scatter of (x, y, z, index) —
// and a centre then tests only the three cell rows around it (each row's three cells are contiguous in the binned order).
// Hits arrive in no particular index order, so "the first nsample in index order" (ball_query_gpu.cu:16-66) is restored at
// the end: every hit's rank among the hits is counted and the nsample lowest indices are written in ascending order — the same
// lists, the same -1 marker, the same first-hit padding as the exhaustive kernel (tests compare them element for element).
// A point within r of a centre is at most one cell away in x and in y: the cell coordinate is a monotone function of the
// coordinate and the cell width exceeds r by 0.1 % (rounding moves a quotient by ~1e-5 of a cell).
#define BQG_DIM 128                                 // grid cells per axis at most (LDS histogram: 128 * 128 ints)
#define BQG_QPW 1                                   // centres per wave (a centre in a dense spot takes many times longer than one in a sparse spot: one each balances best)
#define BQG_CAP 128                                 // slots of a hit list (per centre and radius); cut down to nsample when nearly full
struct BqgGrid { float xlo, ylo, inv, pad; int gw, gh, start, n; };   // per batch element (start / n: its rows in xyz)

__device__ __forceinline__ int bqg_cell1(float v, float lo, float inv, int g) {
    const float q = (v - lo) * inv;
    return q >= 0.f ? min((int)q, g - 1) : 0;       // NaN -> 0
}

__global__ __launch_bounds__(1024) void bqg_build_kernel(int B, const float *__restrict__ xyz, const int *__restrict__ xyz_cnt,
                                                         float cell_min, BqgGrid *__restrict__ grids, int *__restrict__ cell_start,
                                                         float4 *__restrict__ binned) {
    __shared__ int s_hist[BQG_DIM * BQG_DIM];
    __shared__ float s_red[4][16];
    __shared__ int s_wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, l = t & 63, wv = t >> 6;
    int start = 0;
    for (int k = 0; k < b; ++k) start += xyz_cnt[k];
    const int n = xyz_cnt[b];
    const float *P = xyz + (size_t)start * 3;
    float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
    for (int k = t; k < n; k += 1024) {
        const float x = P[(size_t)k * 3], y = P[(size_t)k * 3 + 1];
        if (fabsf(x) < 1.0e30f) { xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); }     // (NaN / inf stay out of the box)
        if (fabsf(y) < 1.0e30f) { ylo = fminf(ylo, y); yhi = fmaxf(yhi, y); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        xlo = fminf(xlo, __shfl_xor(xlo, d, 64)); xhi = fmaxf(xhi, __shfl_xor(xhi, d, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, d, 64)); yhi = fmaxf(yhi, __shfl_xor(yhi, d, 64));
    }
    if (l == 0) { s_red[0][wv] = xlo; s_red[1][wv] = xhi; s_red[2][wv] = ylo; s_red[3][wv] = yhi; }
    for (int k = t; k < BQG_DIM * BQG_DIM; k += 1024) s_hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        xlo = fminf(xlo, s_red[0][q]); xhi = fmaxf(xhi, s_red[1][q]); ylo = fminf(ylo, s_red[2][q]); yhi = fmaxf(yhi, s_red[3][q]);
    }
    if (!(xhi >= xlo)) { xlo = 0.f; xhi = 0.f; }                                   // no finite point
    if (!(yhi >= ylo)) { ylo = 0.f; yhi = 0.f; }
    const float cell = fmaxf(cell_min, fmaxf(xhi - xlo, yhi - ylo) * (1.0f / (BQG_DIM - 1)));
    const float inv = 1.0f / cell;
    const int gw = min(BQG_DIM, (int)((xhi - xlo) * inv) + 1), gh = min(BQG_DIM, (int)((yhi - ylo) * inv) + 1);
    auto cell_of = [&](int k) {
        return bqg_cell1(P[(size_t)k * 3 + 1], ylo, inv, gh) * gw + bqg_cell1(P[(size_t)k * 3], xlo, inv, gw);
    };
    for (int k = t; k < n; k += 1024) atomicAdd(&s_hist[cell_of(k)], 1);
    __syncthreads();
    {   // exclusive scan of the BQG_DIM^2 counts: 16 per thread
        int loc[16], sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { loc[q] = s_hist[t * 16 + q]; sum += loc[q]; }
        const int inc = wave_incl_scan(sum);
        if (l == 63) s_wsum[wv] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int q = 0; q < wv; ++q) base += s_wsum[q];
        int *cs = cell_start + (size_t)b * (BQG_DIM * BQG_DIM + 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            cs[t * 16 + q] = base;
            s_hist[t * 16 + q] = base;                                             // becomes the scatter cursor
            base += loc[q];
        }
        if (t == 1023) cs[BQG_DIM * BQG_DIM] = base;
    }
    __syncthreads();
    for (int k = t; k < n; k += 1024) {
        const int pos = atomicAdd(&s_hist[cell_of(k)], 1);
        binned[(size_t)start + pos] = make_float4(P[(size_t)k * 3], P[(size_t)k * 3 + 1], P[(size_t)k * 3 + 2], __int_as_float(k));
    }
    if (t == 0) grids[b] = BqgGrid{xlo, ylo, inv, 0.f, gw, gh, start, n};
}

// NR hit lists per wave in LDS.  Only the nsample LOWEST indices matter: whenever a list is about to outgrow its BQG_CAP slots
// it is cut down to the nsample lowest, and from then on a hit must also lie below the highest index kept (`lim`) — in a dense
// neighbourhood (hundreds of points inside the ball) almost every later hit is dropped by that one comparison.
template <int NR>
__global__ __launch_bounds__(256) void bqg_query_kernel(int B, int M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                                        const float *__restrict__ new_xyz, const int *__restrict__ new_cnt,
                                                        const BqgGrid *__restrict__ grids, const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ binned, int *__restrict__ idx_a,
                                                        int *__restrict__ idx_b) {
    __shared__ int s_hits[4][NR][BQG_CAP];
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const float r2[2] = {radius_a * radius_a, radius_b * radius_b};
    const int ns[2] = {nsample_a, nsample_b};
    int *const out[2] = {idx_a, idx_b};
    // rank of every entry among the H entries of list[] (indices are distinct); entries with rank < nsel go to dst[rank]
    auto select_lowest = [&](const int *list, int H, int nsel, int *dst) {
#pragma unroll
        for (int u = 0; u < BQG_CAP / 64; ++u) {
            const int i = u * 64 + l;
            if (u * 64 >= H) break;                                                // wave-uniform
            const int v = i < H ? list[i] : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < H; ++j) rank += (list[j] < v) ? 1 : 0;
            if (i < H && rank < nsel) dst[rank] = v;
        }
    };
    for (int qi = 0; qi < BQG_QPW; ++qi) {
        const int q = (blockIdx.x * 4 + wv) * BQG_QPW + qi;                        // wave-uniform
        if (q >= M) break;
        int bs = 0, acc = new_cnt[0];
        for (int k = 1; k < B; ++k) {
            if (q < acc) break;
            acc += new_cnt[k];
            bs = k;
        }
        const BqgGrid g = grids[bs];
        const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
        const int cx = bqg_cell1(qx, g.xlo, g.inv, g.gw), cy = bqg_cell1(qy, g.ylo, g.inv, g.gh);
        const int *cs = cell_start + (size_t)bs * (BQG_DIM * BQG_DIM + 1);
        // the (up to) three cell rows around the centre: ranges of the binned order, requested together
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gw - 1) + 1;
        int p0[3], p1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int y = cy - 1 + d;
            const bool ok = y >= 0 && y < g.gh;
            const int yc = min(max(y, 0), g.gh - 1);
            p0[d] = cs[yc * g.gw + x0];
            p1[d] = ok ? cs[yc * g.gw + x1] : p0[d];
        }
        int cnt[NR], lim[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { cnt[r] = 0; lim[r] = 0x7fffffff; }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            for (int p = p0[d]; p < p1[d]; p += 64) {
                const int k = p + l;
                const float4 c = binned[(size_t)g.start + min(k, p1[d] - 1)];
                const float d2 = pn_dist2(qx, qy, qz, c.x, c.y, c.z);
                const int ci = __float_as_int(c.w);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const bool hit = (k < p1[d]) && (d2 < r2[r]) && (ci < lim[r]);
                    const unsigned long long bal = __ballot(hit);
                    if (!bal) continue;
                    int *list = s_hits[wv][r];
                    if (hit) list[cnt[r] + __popcll(bal & lanemask_lt())] = ci;     // cnt <= BQG_CAP - 64 here: room for one batch
                    cnt[r] += __popcll(bal);
                    if (cnt[r] > BQG_CAP - 64) {                                   // cut down to the nsample lowest, tighten `lim`
                        const int H = cnt[r], keep = min(H, ns[r]);
                        int rk[BQG_CAP / 64], vv[BQG_CAP / 64];
#pragma unroll
                        for (int u = 0; u < BQG_CAP / 64; ++u) {
                            const int i = u * 64 + l;
                            rk[u] = -1;
                            vv[u] = 0;
                            if (i < H) {
                                const int v = list[i];
                                int rank = 0;
                                for (int j = 0; j < H; ++j) rank += (list[j] < v) ? 1 : 0;
                                vv[u] = v;
                                rk[u] = rank < keep ? rank : -1;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < BQG_CAP / 64; ++u)                       // (every rank was computed from the old list above)
                            if (rk[u] >= 0) list[rk[u]] = vv[u];
                        cnt[r] = keep;
                        if (keep == ns[r]) lim[r] = list[keep - 1];                // ascending now: the highest index still of interest
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            int *o = out[r] + (size_t)q * ns[r];
            const int H = cnt[r];
            if (H == 0) {
                if (l == 0) o[0] = -1;
                continue;
            }
            const int *list = s_hits[wv][r];
            select_lowest(list, H, ns[r], o);
            if (H < ns[r]) {                                                       // slots past the last hit repeat the first one
                int lowest = 0x7fffffff;
                for (int i = l; i < H; i += 64) lowest = min(lowest, list[i]);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) lowest = min(lowest, __shfl_xor(lowest, d, 64));
                for (int p = H + l; p < ns[r]; p += 64) o[p] = lowest;
            }
        }
    }
}

LIDAR_EXPORT size_t lidar_ball_query_grid_workspace_bytes(int B, int N) {
    return align_up((size_t)(B > 0 ? B : 1) * sizeof(BqgGrid), 256) +
           align_up((size_t)(B > 0 ? B : 1) * (BQG_DIM * BQG_DIM + 1) * sizeof(int), 256) + align_up((size_t)(N > 0 ? N : 1) * 16, 256) + 256;
}

// lidar_ball_query_stack / _stack2 through a cell grid over the candidates (radius_b <= 0 or idx_b == NULL: one radius).
// N = rows of xyz (sum of xyz_batch_cnt).  Same outputs as the exhaustive kernels; pays off from a few thousand candidates per
// batch element.  ws: lidar_ball_query_grid_workspace_bytes(B, N), no initialisation needed.
LIDAR_EXPORT int lidar_ball_query_stack_grid(int B, int M, int N, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                             const float *new_xyz, const int *new_xyz_batch_cnt, const float *xyz,
                                             const int *xyz_batch_cnt, int *idx_a, int *idx_b, void *ws, size_t ws_bytes,
                                             void *stream) {
    const bool two = idx_b != nullptr && radius_b > 0.f;
    if (B <= 0 || M < 0 || N < 0 || nsample_a <= 0 || (two && nsample_b <= 0) || !(radius_a > 0.f)) return LIDAR_ERR_ARG;
    if (nsample_a > BQG_CAP - 64 || (two && nsample_b > BQG_CAP - 64)) return LIDAR_ERR_ARG;       // a list must hold nsample + one 64-lane batch
    if (M == 0) return LIDAR_OK;
    if (!new_xyz || !new_xyz_batch_cnt || !xyz || !xyz_batch_cnt || !idx_a || !ws) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_ball_query_grid_workspace_bytes(B, N)) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)ws;
    BqgGrid *grids = (BqgGrid *)p;
    p += align_up((size_t)B * sizeof(BqgGrid), 256);
    int *cell_start = (int *)p;
    p += align_up((size_t)B * (BQG_DIM * BQG_DIM + 1) * sizeof(int), 256);
    float4 *binned = (float4 *)p;
    const float rmax = two ? fmaxf(radius_a, radius_b) : radius_a;
    hipLaunchKernelGGL(bqg_build_kernel, dim3(B), dim3(1024), 0, s, B, xyz, xyz_batch_cnt, rmax * 1.001f, grids, cell_start, binned);
    const dim3 grid(divup(M, 4 * BQG_QPW));
    if (two)
        hipLaunchKernelGGL((bqg_query_kernel<2>), grid, dim3(256), 0, s, B, M, radius_a, nsample_a, radius_b, nsample_b, new_xyz,
                           new_xyz_batch_cnt, grids, cell_start, binned, idx_a, idx_b);
    else
        hipLaunchKernelGGL((bqg_query_kernel<1>), grid, dim3(256), 0, s, B, M, radius_a, nsample_a, 0.f, 0, new_xyz,
                           new_xyz_batch_cnt, grids, cell_start, binned, idx_a, (int *)nullptr);
    return lidar_check_launch("lidar_ball_query_stack_grid");
}

LIDAR_EXPORT int lidar_ball_query_batch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                        const float *xyz, int *idx, void *stream) {
    if (b <= 0 || n < 0 || m < 0 || nsample <= 0) return LIDAR_ERR_ARG;
    if (m == 0) return LIDAR_OK;
    if (!new_xyz || !xyz || !idx) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL((ball_query_kernel<false, 1>), dim3(divup(m, BQ_QPB), b), dim3(PN_TPB), 0, (hipStream_t)stream, b, m, n,
                       radius, nsample, 0.f, 0, new_xyz, (const int *)nullptr, xyz, (const int *)nullptr, idx, (int *)nullptr);
    return lidar_check_launch("lidar_ball_query_batch");
}

// ------------------------------------------------------------------ grouping (stack)
// group_points_kernel_stack (pointnet2_stack/src/group_points_gpu.cu:71-102): out (M, C, ns).
// One workgroup per query point: feature rows are read channel-contiguous, the (C, ns) tile is transposed in LDS.
#define GP_MAX_TILE 8192
__global__ __launch_bounds__(PN_TPB) void group_points_stack_kernel(int B, int M, int C, int ns, const float *__restrict__ feat,
                                                                    const int *__restrict__ feat_cnt, const int *__restrict__ idx,
                                                                    const int *__restrict__ idx_cnt, float *__restrict__ out) {
    extern __shared__ float s_tile[];  // [C][ns + 1]
    __shared__ int s_start;
    const int m = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        int bs, st, nn;
        pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
        s_start = st;
    }
    __syncthreads();
    const int start = s_start;
    const int *row = idx + (size_t)m * ns;
    for (int e = t; e < C * ns; e += PN_TPB) {
        const int s = e / C, c = e - s * C;
        s_tile[c * (ns + 1) + s] = feat[((size_t)start + row[s]) * C + c];
    }
    __syncthreads();
    float *o = out + (size_t)m * C * ns;
    for (int e = t; e < C * ns; e += PN_TPB) {
        const int c = e / ns, s = e - c * ns;
        o[e] = s_tile[c * (ns + 1) + s];
    }
}

// fallback for very wide tiles: one thread per output element (the reference's mapping)
__global__ void group_points_stack_naive_kernel(int B, int M, int C, int ns, const float *__restrict__ feat,
                                                const int *__restrict__ feat_cnt, const int *__restrict__ idx,
                                                const int *__restrict__ idx_cnt, float *__restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)M * C * ns) return;
    const int s = (int)(e % ns), c = (int)((e / ns) % C), m = (int)(e / ns / C);
    int bs, st, nn;
    pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
    out[e] = feat[((size_t)st + idx[(size_t)m * ns + s]) * C + c];
}

LIDAR_EXPORT int lidar_group_points_stack(int B, int M, int C, int nsample, const float *features,
                                          const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                                          float *out, void *stream) {
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!features || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (C * (nsample + 1) <= GP_MAX_TILE)
        hipLaunchKernelGGL(group_points_stack_kernel, dim3(M), dim3(PN_TPB), (size_t)C * (nsample + 1) * 4, s, B, M, C, nsample,
                           features, features_batch_cnt, idx, idx_batch_cnt, out);
    else
        hipLaunchKernelGGL(group_points_stack_naive_kernel, dim3(divup((long long)M * C * nsample, 256)), dim3(256), 0, s, B, M, C,
                           nsample, features, features_batch_cnt, idx, idx_batch_cnt, out);
    return lidar_check_launch("lidar_group_points_stack");
}

// Row-major grouping for the inference path of a set-abstraction scale: out (M, ns, stride) with row (m, s) =
// [xyz[idx] - new_xyz[m] (when use_xyz) | features[idx] | zero padding up to stride], an all-zero row set for an empty ball
// (ball query's marker: idx[m][0] < 0).  What QueryAndGroup builds (pointnet2_stack/pointnet2_utils.py:119-155) transposed:
// the shared MLP then runs as plain row-major GEMMs over (M * ns) rows instead of a 1x1 convolution over a strided
// (C, M, ns) view, every row is read and written as one contiguous run, and the (M, C, ns) tensor never exists.
__global__ __launch_bounds__(256) void group_rows_stack_kernel(int B, int M, int C, int ns, int use_xyz, int stride,
                                                               const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const float *__restrict__ feat, const int *__restrict__ feat_cnt,
                                                               const int *__restrict__ idx, const int *__restrict__ idx_cnt,
                                                               float *__restrict__ out) {
    __shared__ int s_start;
    const int m = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        int bs, st, nn;
        pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
        s_start = st;
    }
    __syncthreads();
    const int start = s_start, X = use_xyz ? 3 : 0, Ct = X + C;
    const int *row = idx + (size_t)m * ns;
    const bool empty = row[0] < 0;
    float *o = out + (size_t)m * ns * stride;
    for (int e = t; e < ns * stride; e += 256) {
        const int s = e / stride, c = e - s * stride;
        float v = 0.f;
        if (!empty && c < Ct) {
            const size_t r = (size_t)start + row[s];
            v = c < X ? xyz[r * 3 + c] - new_xyz[(size_t)m * 3 + c] : feat[r * C + (c - X)];
        }
        o[e] = v;
    }
}

// The first layer of the shared MLP commutes with the gather: [xyz[idx] - new_xyz[m] | features[idx]] @ W + b
//   == ([xyz | features] @ W + b)[idx]  -  (new_xyz @ W_xyz)[m],
// i.e. ONE small GEMM per source point (N rows) instead of one per (query, sample) pair (M * nsample rows, 170 x more for
// PV-RCNN's RoI-grid pooling), after which the layer is a gather of H-wide rows, a per-query subtraction and the ReLU:
//   out (M, ns, H) = relu(table[idx] - query_term[m]),   an empty ball (idx[m][0] < 0) -> empty_row (= relu(b): zero inputs).
// H % 4 == 0; query_term may be null (use_xyz == False).
__global__ __launch_bounds__(256) void group_rows_affine_stack_kernel(int B, int M, int H4, int ns, const float4 *__restrict__ table,
                                                                      const float4 *__restrict__ query_term,
                                                                      const float4 *__restrict__ empty_row,
                                                                      const int *__restrict__ feat_cnt, const int *__restrict__ idx,
                                                                      const int *__restrict__ idx_cnt, float4 *__restrict__ out) {
    __shared__ int s_start;
    const int m = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        int bs, st, nn;
        pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
        s_start = st;
    }
    __syncthreads();
    const int start = s_start;
    const int *row = idx + (size_t)m * ns;
    const bool empty = row[0] < 0;
    float4 *o = out + (size_t)m * ns * H4;
    for (int e = t; e < ns * H4; e += 256) {
        const int s = e / H4, c = e - s * H4;
        float4 v;
        if (empty) {
            v = empty_row[c];
        } else {
            v = table[((size_t)start + row[s]) * H4 + c];
            if (query_term) {
                const float4 q = query_term[(size_t)m * H4 + c];
                v.x -= q.x; v.y -= q.y; v.z -= q.z; v.w -= q.w;
            }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        o[e] = v;
    }
}

LIDAR_EXPORT int lidar_group_rows_affine_stack(int B, int M, int H, int nsample, const float *table, const float *query_term,
                                               const float *empty_row, const int *features_batch_cnt, const int *idx,
                                               const int *idx_batch_cnt, float *out, void *stream) {
    if (B <= 0 || M < 0 || H <= 0 || (H & 3) || nsample <= 0) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!table || !empty_row || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(group_rows_affine_stack_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, B, M, H / 4, nsample,
                       (const float4 *)table, (const float4 *)query_term, (const float4 *)empty_row, features_batch_cnt, idx,
                       idx_batch_cnt, (float4 *)out);
    return lidar_check_launch("lidar_group_rows_affine_stack");
}

// Two-layer scale in ONE kernel: the gather of layer-1 rows above, the second layer on the matrix cores and the max over the
// samples, without the (M * ns, H1) and (M * ns, H2) intermediates ever reaching HBM:
//   out[m] = max_s relu(relu(table[idx[m][s]] - query_term[m]) @ W2 + b2)            (empty ball: relu(relu(b1) @ W2 + b2))
// A wave owns 32 (query, sample) rows = 32 / NS queries: it gathers them into its own LDS tile (no workgroup barrier in the loop:
// W2 is staged once per workgroup and never changes), runs H1 / 2 steps of v_mfma_f32_32x32x2_f32 per 32 output columns, and
// reduces the accumulator rows of each query (register subsets of the MFMA layout, then the two lane halves).
// H1 = 4 * C4 in {16, 32, 64}, H2 <= NT * 32, NS in {8, 16, 32}.
typedef float sa_f32x16 __attribute__((ext_vector_type(16)));
template <int NT, int C4, int NS>
__global__ __launch_bounds__(256) void sa_layer2_max_kernel(int B, int M, int H2, const float4 *__restrict__ table,
                                                            const float4 *__restrict__ query_term,
                                                            const float4 *__restrict__ empty_row, const float *__restrict__ W2,
                                                            const float *__restrict__ b2, const int *__restrict__ feat_cnt,
                                                            const int *__restrict__ idx, const int *__restrict__ idx_cnt,
                                                            float *__restrict__ out) {
    constexpr int H1 = C4 * 4, Cp = H1 + 1, CW = NT * 32, QW = 32 / NS, NG = (32 * C4) / 64, RPQ = NS / 2;
    extern __shared__ float s_sa[];
    float *s_w = s_sa;                                    // [H1][CW], zero beyond H2
    float *A = s_sa + H1 * CW + (threadIdx.x >> 6) * 32 * Cp;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int e = t; e < H1 * CW; e += 256) {
        const int ci = e / CW, co = e - ci * CW;
        s_w[e] = co < H2 ? W2[(size_t)ci * H2 + co] : 0.f;
    }
    float bias[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) bias[q] = (q * 32 + (l & 31)) < H2 ? b2[q * 32 + (l & 31)] : 0.f;
    __syncthreads();
    const int ar = l & 31, ak = l >> 5;
    const int ngroups = (M + QW - 1) / QW;
    for (int g = blockIdx.x * 4 + wv; g < ngroups; g += gridDim.x * 4) {              // wave-uniform
        // row r = l & 31 of the tile: query m0 + r / NS, sample r % NS
        const int mq = g * QW + (l & 31) / NS;
        int src = -1;                                                              // source row in `table`, -1: empty ball / no query
        if (mq < M) {
            const int *row = idx + (size_t)mq * NS;
            if (row[0] >= 0) {
                int b = 0, acc = idx_cnt[0], start = 0;
                for (int k = 1; k < B; ++k) {
                    if (mq < acc) break;
                    acc += idx_cnt[k];
                    start += feat_cnt[k - 1];
                    b = k;
                }
                (void)b;
                src = start + row[(l & 31) % NS];
            }
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int e = u * 64 + l, r = e / C4, c = e - r * C4;
            const int sr = __shfl(src, r, 64);
            const int mr = g * QW + r / NS;
            float4 v = empty_row[c];
            if (sr >= 0) {
                v = table[(size_t)sr * C4 + c];
                if (query_term) {
                    const float4 qv = query_term[(size_t)mr * C4 + c];
                    v.x -= qv.x; v.y -= qv.y; v.z -= qv.z; v.w -= qv.w;
                }
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            float *dst = A + r * Cp + c * 4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        sa_f32x16 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll 8
        for (int c0 = 0; c0 < H1; c0 += 2) {
            const float a = A[ar * Cp + c0 + ak];
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float b = s_w[(c0 + ak) * CW + q * 32 + ar];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            }
        }
        // accumulator register r of lane l holds row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column q * 32 + (l & 31):
        // query j of the tile = rows [j * NS, (j + 1) * NS) = registers [j * RPQ, (j + 1) * RPQ) of both lane halves
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int j = 0; j < QW; ++j) {
                float mx = -3.0e38f;
#pragma unroll
                for (int r = 0; r < RPQ; ++r) mx = fmaxf(mx, acc[q][j * RPQ + r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const int m = g * QW + j, col = q * 32 + (l & 31);
                if (l < 32 && m < M && col < H2) out[(size_t)m * H2 + col] = fmaxf(mx + bias[q], 0.f);   // relu and max commute
            }
        }
    }
}

LIDAR_EXPORT int lidar_sa_layer2_max_supported(int H1, int H2, int nsample) {
    return ((H1 == 16 || H1 == 32 || H1 == 64) && H2 >= 1 && H2 <= 128 && (nsample == 8 || nsample == 16 || nsample == 32)) ? 1 : 0;
}

// table (N, H1), query_term (M, H1) or null, empty_row (H1), W2 (H1, H2) row-major, b2 (H2), idx = RAW ball-query result
// (M, nsample); out (M, H2).  See lidar_group_rows_affine_stack for the first layer's algebra.
LIDAR_EXPORT int lidar_sa_layer2_max_stack(int B, int M, int H1, int H2, int nsample, const float *table, const float *query_term,
                                           const float *empty_row, const float *W2, const float *b2,
                                           const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt, float *out,
                                           void *stream) {
    if (B <= 0 || M < 0 || !lidar_sa_layer2_max_supported(H1, H2, nsample)) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!table || !empty_row || !W2 || !b2 || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return LIDAR_ERR_ARG;
    const int nt = divup(H2, 32), c4 = H1 / 4;
    const size_t lds = ((size_t)H1 * nt * 32 + (size_t)4 * 32 * (H1 + 1)) * sizeof(float);
    const int qw = 32 / nsample, ngroups = divup(M, qw);
    const int blocks = (int)std::min<long long>(divup(ngroups, 4), 256 * 3);
    hipStream_t s = (hipStream_t)stream;
#define SAL(NT, C4, NS) hipLaunchKernelGGL((sa_layer2_max_kernel<NT, C4, NS>), dim3(blocks), dim3(256), lds, s, B, M, H2, (const float4 *)table, (const float4 *)query_term, (const float4 *)empty_row, W2, b2, features_batch_cnt, idx, idx_batch_cnt, out)
#define SAL_NS(NT, C4) do { if (nsample == 8) SAL(NT, C4, 8); else if (nsample == 16) SAL(NT, C4, 16); else SAL(NT, C4, 32); } while (0)
#define SAL_C4(NT) do { if (c4 == 4) SAL_NS(NT, 4); else if (c4 == 8) SAL_NS(NT, 8); else SAL_NS(NT, 16); } while (0)
    switch (nt) { case 1: SAL_C4(1); break; case 2: SAL_C4(2); break; case 3: SAL_C4(3); break; default: SAL_C4(4); break; }
#undef SAL_C4
#undef SAL_NS
#undef SAL
    return lidar_check_launch("lidar_sa_layer2_max_stack");
}

// idx: the RAW ball-query result (-1 in column 0 marks an empty ball).  features may be null (C = 0, use_xyz required).
LIDAR_EXPORT int lidar_group_rows_stack(int B, int M, int C, int nsample, int use_xyz, int stride, const float *xyz,
                                        const float *new_xyz, const float *features, const int *features_batch_cnt, const int *idx,
                                        const int *idx_batch_cnt, float *out, void *stream) {
    if (B <= 0 || M < 0 || C < 0 || nsample <= 0 || stride < C + (use_xyz ? 3 : 0) || (C == 0 && !use_xyz)) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if ((C > 0 && !features) || (use_xyz && (!xyz || !new_xyz)) || !features_batch_cnt || !idx || !idx_batch_cnt || !out)
        return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(group_rows_stack_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, B, M, C, nsample, use_xyz, stride, xyz,
                       new_xyz, features, features_batch_cnt, idx, idx_batch_cnt, out);
    return lidar_check_launch("lidar_group_rows_stack");
}

// group_points_grad_kernel_stack (:15-45): grad_features (N, C) += grad_out (M, C, ns); atomics issued with
// the channel on the lane (contiguous row segments of grad_features per wave instruction)
__global__ __launch_bounds__(PN_TPB) void group_points_grad_stack_kernel(int B, int M, int C, int ns, const float *__restrict__ grad_out,
                                                                         const int *__restrict__ idx, const int *__restrict__ idx_cnt,
                                                                         const int *__restrict__ feat_cnt, float *__restrict__ grad_feat) {
    extern __shared__ float s_tile[];  // [C][ns + 1]
    __shared__ int s_start;
    const int m = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        int bs, st, nn;
        pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
        s_start = st;
    }
    const float *g = grad_out + (size_t)m * C * ns;
    for (int e = t; e < C * ns; e += PN_TPB) {
        const int c = e / ns, s = e - c * ns;
        s_tile[c * (ns + 1) + s] = g[e];
    }
    __syncthreads();
    const int start = s_start;
    const int *row = idx + (size_t)m * ns;
    for (int e = t; e < C * ns; e += PN_TPB) {
        const int s = e / C, c = e - s * C;
        atomicAdd(&grad_feat[((size_t)start + row[s]) * C + c], s_tile[c * (ns + 1) + s]);
    }
}

__global__ void group_points_grad_stack_naive_kernel(int B, int M, int C, int ns, const float *__restrict__ grad_out,
                                                     const int *__restrict__ idx, const int *__restrict__ idx_cnt,
                                                     const int *__restrict__ feat_cnt, float *__restrict__ grad_feat) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)M * C * ns) return;
    const int s = (int)(e % ns), c = (int)((e / ns) % C), m = (int)(e / ns / C);
    int bs, st, nn;
    pn_batch_of(idx_cnt, B, m, feat_cnt, bs, st, nn);
    atomicAdd(&grad_feat[((size_t)st + idx[(size_t)m * ns + s]) * C + c], grad_out[e]);
}

LIDAR_EXPORT int lidar_group_points_grad_stack(int B, int M, int C, int N, int nsample, const float *grad_out, const int *idx,
                                               const int *idx_batch_cnt, const int *features_batch_cnt,
                                               float *grad_features, void *stream) {
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0 || N < 0) return LIDAR_ERR_ARG;
    if (M == 0) return LIDAR_OK;
    if (!grad_out || !idx || !idx_batch_cnt || !features_batch_cnt || !grad_features) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (C * (nsample + 1) <= GP_MAX_TILE)
        hipLaunchKernelGGL(group_points_grad_stack_kernel, dim3(M), dim3(PN_TPB), (size_t)C * (nsample + 1) * 4, s, B, M, C, nsample,
                           grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
    else
        hipLaunchKernelGGL(group_points_grad_stack_naive_kernel, dim3(divup((long long)M * C * nsample, 256)), dim3(256), 0, s, B, M,
                           C, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
    return lidar_check_launch("lidar_group_points_grad_stack");
}

// ------------------------------------------------------------------ furthest point sampling
// furthest_point_sampling_kernel (pointnet2_stack/src/sampling_gpu.cu:24-140; identical in pointnet2_batch).
// Reference tie rule reproduced: the winner is the maximum running distance; among equal maxima the smallest
// (k mod block_ref, k) wins, where block_ref = largest power of two <= min(n, 1024) is the reference's block size.
struct FpsBest {
    float v;
    int key;  // (k mod block_ref) << 15 | k   (k < 32768); smaller key wins ties
};

__device__ __forceinline__ bool fps_better(float v2, int k2, float v1, int k1) {
    return (v2 > v1) || (v2 == v1 && k2 < k1);
}

// Cross-lane helpers on the VALU's DPP path (no LDS traffic): lane <-> lane^1, lane^2 (quad permutes), mirror within 8 and
// within 16 lanes.  Applied in this order with a commutative "better of two" they leave the best of each 16-lane row in
// all of its lanes.
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within each 8
#define DPP_MIRROR 0x140       // lane i <-> 15 - i within each 16
template <int CTRL>
__device__ __forceinline__ void fps_dpp_step(float &v, int &key) {
    const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    const int ok = __builtin_amdgcn_update_dpp(0, key, CTRL, 0xf, 0xf, true);
    const bool take = fps_better(ov, ok, v, key);
    v = take ? ov : v;
    key = take ? ok : key;
}
__device__ __forceinline__ void fps_row_reduce(float &v, int &key) {
    fps_dpp_step<DPP_XOR1>(v, key);
    fps_dpp_step<DPP_XOR2>(v, key);
    fps_dpp_step<DPP_HALF_MIRROR>(v, key);
    fps_dpp_step<DPP_MIRROR>(v, key);
}

typedef float fps_f2 __attribute__((ext_vector_type(2)));

// One workgroup per sample; point k = j * 1024 + sl * TPB + t belongs to thread t (j < JN, sl < 1024 / TPB).  Coordinates
// stay in registers, running min-distances in LDS (80 KB for 20 480 points: with them in registers too the 1024-thread
// kernel spills a third of its state to scratch).  A round is VALU-bound on its single CU (n distance updates), so the loop
// is kept to packed fp32 math (two points per instruction), branch-free selects of (distance, item) only, DPP reductions,
// and the winner's coordinates are picked out of the registers by its owner wave alone.  A thread's points of one sl share
// the reference's tree slot (k mod 1024): among them the first strictly greater distance wins; different sl are merged
// with the full tie rule.  Measured, 8 x 19 968 points -> 2 048 samples: first version 9.4 ms, this one 6.5 ms with
// TPB = 1024 (the launcher's choice), 8.8 ms with TPB = 512.
template <int JN, int TPB>
__global__ __launch_bounds__(TPB) void fps_kernel(int n, int m, int block_ref_mask, const float *__restrict__ data,
                                                  float *__restrict__ temp, int *__restrict__ idxs) {
    static_assert(JN % 2 == 0 && (TPB == 512 || TPB == 1024), "two points per packed instruction");
    constexpr int SL = 1024 / TPB, JH = JN / 2;
    extern __shared__ float s_dyn[];                   // running distances: [SL][JH][TPB] pairs (points 2h, 2h+1)
    fps_f2 *s_pt = reinterpret_cast<fps_f2 *>(s_dyn);
    __shared__ float s_v[2][16];
    __shared__ int s_k[2][16];
    __shared__ float s_c[2][3];
    const int bidx = blockIdx.x, t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *D = data + (size_t)bidx * n * 3;
    float *T = temp + (size_t)bidx * n;
    int *O = idxs + (size_t)bidx * m;
    fps_f2 px[SL][JH], py[SL][JH], pz[SL][JH];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int k = j * 1024 + sl * TPB + t;
            const int kc = min(k, n - 1);
            px[sl][j / 2][j & 1] = D[(size_t)kc * 3 + 0];
            py[sl][j / 2][j & 1] = D[(size_t)kc * 3 + 1];
            pz[sl][j / 2][j & 1] = D[(size_t)kc * 3 + 2];
            s_dyn[(((sl * JH) + j / 2) * TPB + t) * 2 + (j & 1)] = (k < n) ? T[kc] : -1.0f;   // slots past n never win
        }
    float x1 = D[0], y1 = D[1], z1 = D[2];
    if (t == 0) O[0] = 0;
    if (t < 16) { s_v[0][t] = s_v[1][t] = -2.f; s_k[0][t] = s_k[1][t] = 0x7fffffff; }   // unused partial slots never win
    __syncthreads();
    for (int r = 1; r < m; ++r) {
        const fps_f2 qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
        float v = -3.f;
        int key = 0x7fffffff;
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
            float best = -1.f;
            int bj = 0;
#pragma unroll
            for (int h = 0; h < JH; ++h) {
                const fps_f2 dx = px[sl][h] - qx, dy = py[sl][h] - qy, dz = pz[sl][h] - qz;
                const fps_f2 d = dx * dx + dy * dy + dz * dz;
                const fps_f2 old = s_pt[(sl * JH + h) * TPB + t];
                const float d0 = fminf(d[0], old[0]), d1 = fminf(d[1], old[1]);
                s_pt[(sl * JH + h) * TPB + t] = fps_f2{d0, d1};
                const bool g0 = d0 > best;
                best = g0 ? d0 : best;
                bj = g0 ? 2 * h : bj;
                const bool g1 = d1 > best;
                best = g1 ? d1 : best;
                bj = g1 ? 2 * h + 1 : bj;
            }
            const int bk = bj * 1024 + sl * TPB + t;
            // reference slot of the candidate = k mod block_ref; its LDS tree keeps, among equal values, the slot whose index
            // is smaller when read from the least-significant bit up (bit-reversed order)
            int kk = (int)(__brev((unsigned)(bk & block_ref_mask)) >> 17);
            kk = (kk << 15) | bk;
            const bool take = fps_better(best, kk, v, key);
            v = take ? best : v;
            key = take ? kk : key;
        }
        fps_row_reduce(v, key);
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const float ov = __shfl_xor(v, d, 64);
            const int ok = __shfl_xor(key, d, 64);
            const bool take = fps_better(ov, ok, v, key);
            v = take ? ov : v;
            key = take ? ok : key;
        }
        const int buf = r & 1;
        if (l == 0) { s_v[buf][wv] = v; s_k[buf][wv] = key; }
        __syncthreads();
        // every wave reduces the (<= 16) partials redundantly (each 16-lane row holds all of them)
        float fv = s_v[buf][l & 15];
        int fk = s_k[buf][l & 15];
        fps_row_reduce(fv, fk);
        const int kw = __builtin_amdgcn_readfirstlane(fk) & 0x7FFF;      // winning point of the round
        const int slot = kw & 1023, sw = slot / TPB, tw = slot - sw * TPB;
        if (wv == (tw >> 6)) {                                            // its owner wave digs the coordinates out
            const int jw = kw >> 10;
            float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    const bool is = (j == jw) && (sl == sw);
                    cx = is ? px[sl][j / 2][j & 1] : cx;
                    cy = is ? py[sl][j / 2][j & 1] : cy;
                    cz = is ? pz[sl][j / 2][j & 1] : cz;
                }
            if (l == (tw & 63)) { s_c[buf][0] = cx; s_c[buf][1] = cy; s_c[buf][2] = cz; }
        }
        if (t == 0) O[r] = kw;
        __syncthreads();
        x1 = s_c[buf][0]; y1 = s_c[buf][1]; z1 = s_c[buf][2];
    }
#pragma unroll
    for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int k = j * 1024 + sl * TPB + t;
            if (k < n) T[k] = s_dyn[(((sl * JH) + j / 2) * TPB + t) * 2 + (j & 1)];   // the reference leaves the distances in temp
        }
}

// ------------------------------------------------------------------ bucketed FPS (4 096 <= n <= 20 480)
// The kernel above spends its round on n distance updates although a new sample only lowers the running distances of the
// points around it.  Here the points of a sample are sorted once into spatial buckets of 64 (Morton order of a 32 x 32 x/y
// grid over the sample's extent; a bucket = one register slot of one wave, a point = one lane), every bucket keeps its bounding
// box and the largest running distance of its points (with that point's tie key and coordinates), and a round
//   1. tests every bucket: lb = squared distance from the new sample to the bucket's box, evaluated with the SAME fp32
//      expression shape as a point distance, so lb <= d(p) for every point p of the bucket by monotonicity of each rounding;
//      lb >= the bucket's largest running distance  =>  min(running, d) changes nothing there: the bucket is skipped;
//   2. updates the few buckets that remain (a wave-uniform branch per register slot), re-deriving their maximum;
//   3. takes the argmax over the buckets' maxima: wave -> one LDS partial per wave -> ONE barrier -> every wave finishes.
// The running-distance array is the reference's, element for element, and the argmax uses the same total order (value, then
// the reference's tie key), so the sample sequence is identical — only work that cannot change it is left out.
// 8 waves x 40 slots x 64 lanes = 20 480 points; coordinates live in registers (2 waves per SIMD), running distances in LDS.
#define FB_W 8
#define FB_S 40
#define FB_TPB (FB_W * 64)
#define FB_CAP (FB_W * FB_S * 64)

__device__ __forceinline__ float fb_dpp_max_row(float v) {        // max of each 16-lane row, in all of its lanes
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_XOR1, 0xf, 0xf, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_XOR2, 0xf, 0xf, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_HALF_MIRROR, 0xf, 0xf, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_MIRROR, 0xf, 0xf, true)));
    return v;
}
__device__ __forceinline__ float fb_rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float fb_wave_max(float v) {            // wave-uniform max of 64 lanes
    v = fb_dpp_max_row(v);
    return fmaxf(fmaxf(fb_rl(v, 15), fb_rl(v, 31)), fmaxf(fb_rl(v, 47), fb_rl(v, 63)));
}
__device__ __forceinline__ float fb_wave_min(float v) { return -fb_wave_max(-v); }
// best (value, key) of 64 lanes, wave-uniform
__device__ __forceinline__ void fb_wave_best(float &v, int &key) {
    fps_row_reduce(v, key);
    float bv = fb_rl(v, 15);
    int bk = __builtin_amdgcn_readlane(key, 15);
#pragma unroll
    for (int q = 31; q < 64; q += 16) {
        const float ov = fb_rl(v, q);
        const int ok = __builtin_amdgcn_readlane(key, q);
        const bool take = fps_better(ov, ok, bv, bk);
        bv = take ? ov : bv;
        bk = take ? ok : bk;
    }
    v = bv;
    key = bk;
}
__device__ __forceinline__ int fb_tie_key(int k, int block_ref_mask) {
    return (int)((__brev((unsigned)(k & block_ref_mask)) >> 17) << 15) | k;
}
__device__ __forceinline__ int fb_morton5(int cx, int cy) {
    int m = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) m |= (((cx >> b) & 1) << (2 * b)) | (((cy >> b) & 1) << (2 * b + 1));
    return m;
}

typedef float fb_v16 __attribute__((ext_vector_type(16)));
// register slot (hi * 16 + lo) of a 48-slot bank, hi / lo wave-uniform
__device__ __forceinline__ float fb_pick(fb_v16 b0, fb_v16 b1, fb_v16 b2, int hi, int lo) {
    const float a = b0[lo], b = b1[lo], c = b2[lo];
    return hi == 0 ? a : (hi == 1 ? b : c);
}
#define FB_PUT(B, j, v) do { if ((j) < 16) B##0[(j) & 15] = (v); else if ((j) < 32) B##1[(j) & 15] = (v); else B##2[(j) & 15] = (v); } while (0)

__global__ __launch_bounds__(FB_TPB) void fps_bucket_kernel(int n, int m, int block_ref_mask, const float *__restrict__ data,
                                                            float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ unsigned short s_perm[FB_CAP];        // sorted rank -> original point index
    __shared__ float s_md[FB_CAP];                   // running distances by sorted rank (only active buckets touch them)
    __shared__ int s_hist[1024];
    __shared__ int s_wsum[FB_W];
    __shared__ float s_ext[4][FB_W];
    __shared__ float4 s_rec[2][FB_W][2];             // per round parity and wave: (value, tie key, x, y), (z, -, -, -)
    const int bidx = blockIdx.x, t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *D = data + (size_t)bidx * n * 3;
    float *T = temp + (size_t)bidx * n;
    int *O = idxs + (size_t)bidx * m;
    // ---- x / y extent of the sample
    float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
    for (int k = t; k < n; k += FB_TPB) {
        const float x = D[(size_t)k * 3], y = D[(size_t)k * 3 + 1];
        xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); ylo = fminf(ylo, y); yhi = fmaxf(yhi, y);
    }
    xlo = fb_wave_min(xlo); xhi = fb_wave_max(xhi); ylo = fb_wave_min(ylo); yhi = fb_wave_max(yhi);
    if (l == 0) { s_ext[0][wv] = xlo; s_ext[1][wv] = xhi; s_ext[2][wv] = ylo; s_ext[3][wv] = yhi; }
    for (int k = t; k < 1024; k += FB_TPB) s_hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FB_W; ++q) {
        xlo = fminf(xlo, s_ext[0][q]); xhi = fmaxf(xhi, s_ext[1][q]); ylo = fminf(ylo, s_ext[2][q]); yhi = fmaxf(yhi, s_ext[3][q]);
    }
    const float sx = 32.0f / fmaxf(xhi - xlo, 1e-6f), sy = 32.0f / fmaxf(yhi - ylo, 1e-6f);
    auto cell_of = [&](int k) {
        const float x = D[(size_t)k * 3], y = D[(size_t)k * 3 + 1];
        const int cx = min(max((int)((x - xlo) * sx), 0), 31), cy = min(max((int)((y - ylo) * sy), 0), 31);   // NaN -> 0
        return fb_morton5(cx, cy);
    };
    // ---- counting sort by cell (any assignment of points to buckets gives the same samples; compact ones skip more)
    for (int k = t; k < n; k += FB_TPB) atomicAdd(&s_hist[cell_of(k)], 1);
    __syncthreads();
    {
        const int a = s_hist[2 * t], b = s_hist[2 * t + 1];
        const int inc = wave_incl_scan(a + b);
        if (l == 63) s_wsum[wv] = inc;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wv; ++q) base += s_wsum[q];
        s_hist[2 * t] = base + inc - a - b;
        s_hist[2 * t + 1] = base + inc - b;
    }
    __syncthreads();
    for (int k = t; k < n; k += FB_TPB) s_perm[atomicAdd(&s_hist[cell_of(k)], 1)] = (unsigned short)k;
    __syncthreads();
    // ---- registers: slot j of wave wv = bucket j * FB_W + wv, lane l = its l-th point
    // coordinates: 3 x 16 register slots per axis; a wave-uniform slot number selects one with M0-relative register addressing
    // (v_movrels), so the code below is the same for every bucket (40 specialised copies of it thrash the instruction cache)
    fb_v16 PX0 = {}, PX1 = {}, PX2 = {}, PY0 = {}, PY1 = {}, PY2 = {}, PZ0 = {}, PZ1 = {}, PZ2 = {};
    // bucket state, meaningful in lane j (< FB_S) for bucket j of this wave
    float blx = 0.f, bly = 0.f, blz = 0.f, bhx = 0.f, bhy = 0.f, bhz = 0.f;    // bounding box
    float bm = -1.0f, bcx = 0.f, bcy = 0.f, bcz = 0.f;                          // largest running distance, its point
    int bk = 0x7fffffff;                                                        // ... and that point's tie key
#pragma unroll
    for (int j = 0; j < FB_S; ++j) {
        const int rank = (j * FB_W + wv) * 64 + l;
        const bool valid = rank < n;
        const int k = valid ? (int)s_perm[rank] : 0;
        const float qx = D[(size_t)k * 3], qy = D[(size_t)k * 3 + 1], qz = D[(size_t)k * 3 + 2];
        FB_PUT(PX, j, qx); FB_PUT(PY, j, qy); FB_PUT(PZ, j, qz);
        s_md[(j * FB_W + wv) * 64 + l] = valid ? T[k] : -1.0f;                   // lanes without a point never win
        const float inf = 3.0e38f;
        const float lx = fb_wave_min(valid ? qx : inf), ly = fb_wave_min(valid ? qy : inf), lz = fb_wave_min(valid ? qz : inf);
        const float hx = fb_wave_max(valid ? qx : -inf), hy = fb_wave_max(valid ? qy : -inf), hz = fb_wave_max(valid ? qz : -inf);
        const bool any = (j * FB_W + wv) * 64 < n;
        if (l == j) {
            blx = lx; bly = ly; blz = lz; bhx = hx; bhy = hy; bhz = hz;
            bm = any ? 3.0e38f : -1.0f;                                          // "unknown, large": the first round visits it
        }
    }
    float x1 = D[0], y1 = D[1], z1 = D[2];
    if (t == 0) O[0] = 0;
#ifdef FB_STATS
    long long fb_nact = 0, fb_t[6] = {0, 0, 0, 0, 0, 0}, fb_c0 = clock64();
#define FB_T(k) do { const long long c_ = clock64(); fb_t[k] += c_ - fb_c0; fb_c0 = c_; } while (0)
#else
#define FB_T(k) do { } while (0)
#endif
    for (int r = 1; r < m; ++r) {
        // 1. which of my buckets can change?  (same expression shape as the point distance below)
        const float gx = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f), gy = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f),
                    gz = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
        const float lb = gx * gx + gy * gy + gz * gz;
        unsigned long long act = __ballot(l < FB_S && lb < bm);
#ifdef FB_STATS
        fb_nact += __popcll(act);
#endif
        FB_T(0);
        // 2. update them (about one bucket per wave and round): new running distances, and the bucket's new maximum with the
        // point that holds it.  The register slot is reached through a balanced tree of wave-uniform branches (fb_dispatch).
        while (act) {                                                            // wave-uniform
            const int j = __builtin_ctzll(act);
            act &= act - 1ull;
            const int base = (j * FB_W + wv) * 64;
            const float old = s_md[base + l];
            const int kpt = (int)s_perm[base + l];                               // (requested together with the distances)
            const int hi = j >> 4, lo = j & 15;                                   // wave-uniform
            const float qx = fb_pick(PX0, PX1, PX2, hi, lo), qy = fb_pick(PY0, PY1, PY2, hi, lo), qz = fb_pick(PZ0, PZ1, PZ2, hi, lo);
            const float dx = qx - x1, dy = qy - y1, dz = qz - z1;
            const float nd = fminf(dx * dx + dy * dy + dz * dz, old);
            s_md[base + l] = nd;
            const float vmax = fb_wave_max(nd);
            const unsigned long long tie = __ballot(nd == vmax);
            int wl = __builtin_ctzll(tie);
            if (__builtin_expect(__popcll(tie) > 1, 0)) {                        // equal maxima: the reference's tie rule decides
                float tv = (nd == vmax) ? 1.f : 0.f;
                int tk = fb_tie_key(kpt, block_ref_mask);
                const int mine = tk;
                fb_wave_best(tv, tk);
                wl = __builtin_ctzll(__ballot(nd == vmax && mine == tk));
            }
            const float cx = fb_rl(qx, wl), cy = fb_rl(qy, wl), cz = fb_rl(qz, wl);
            const int key = fb_tie_key(__builtin_amdgcn_readlane(kpt, wl), block_ref_mask);
            if (l == j) { bm = vmax; bk = key; bcx = cx; bcy = cy; bcz = cz; }
        }
        FB_T(1);
        // 3. best bucket of this wave -> one LDS record per wave -> barrier -> every wave finishes from the FB_W records
        // (reductions on the VALUE only; the tie key is looked at when two maxima coincide, which is rare)
        const float v = fb_wave_max((l < FB_S) ? bm : -2.0f);
        unsigned long long cand = __ballot(l < FB_S && bm == v);
        int jl = __builtin_ctzll(cand);
        int key = __builtin_amdgcn_readlane(bk, jl);
        cand &= cand - 1ull;
        while (__builtin_expect(cand != 0ull, 0)) {
            const int j2 = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            const int k2 = __builtin_amdgcn_readlane(bk, j2);
            if (k2 < key) { key = k2; jl = j2; }
        }
        const int buf = r & 1;
        if (l == 0) {
            s_rec[buf][wv][0] = make_float4(v, __int_as_float(key), fb_rl(bcx, jl), fb_rl(bcy, jl));
            s_rec[buf][wv][1].x = fb_rl(bcz, jl);
        }
        FB_T(2);
        __syncthreads();
        FB_T(3);
        const float4 ra = s_rec[buf][l & (FB_W - 1)][0];
        const float rz = s_rec[buf][l & (FB_W - 1)][1].x;
        const float fv = fb_rl(fb_dpp_max_row(ra.x), 0);                          // every 16-lane row holds all FB_W records
        unsigned long long wc = __ballot(l < FB_W && ra.x == fv);
        int wq = __builtin_ctzll(wc);
        int fk = __builtin_amdgcn_readlane(__float_as_int(ra.y), wq);
        wc &= wc - 1ull;
        while (__builtin_expect(wc != 0ull, 0)) {
            const int w2 = __builtin_ctzll(wc);
            wc &= wc - 1ull;
            const int k2 = __builtin_amdgcn_readlane(__float_as_int(ra.y), w2);
            if (k2 < fk) { fk = k2; wq = w2; }
        }
        x1 = fb_rl(ra.z, wq); y1 = fb_rl(ra.w, wq); z1 = fb_rl(rz, wq);
        if (t == 0) O[r] = fk & 0x7FFF;
        FB_T(4);
    }
#ifdef FB_STATS   // debug build: bucket-rounds visited by this wave -> temp[wave]; cycles per phase of wave 0 -> temp[16..20]
    __syncthreads();
    if (l == 0) T[wv] = (float)fb_nact;
    if (t == 0) for (int q = 0; q < 5; ++q) T[16 + q] = (float)fb_t[q];
    return;
#endif
#pragma unroll
    for (int j = 0; j < FB_S; ++j) {
        const int rank = (j * FB_W + wv) * 64 + l;
        if (rank < n) T[s_perm[rank]] = s_md[rank];                               // the reference leaves the distances in temp
    }
}

// generic fallback (any n): running distances stay in global memory, same tie rule
__global__ __launch_bounds__(1024) void fps_generic_kernel(int n, int m, int block_ref, const float *__restrict__ data,
                                                           float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ float s_v[2][16];
    __shared__ long long s_k[2][16];
    const int bidx = blockIdx.x, t = threadIdx.x, l = t & 63, wv = t >> 6;
    const float *D = data + (size_t)bidx * n * 3;
    float *T = temp + (size_t)bidx * n;
    int *O = idxs + (size_t)bidx * m;
    int old = 0;
    if (t == 0) O[0] = 0;
    for (int r = 1; r < m; ++r) {
        const float x1 = D[(size_t)old * 3], y1 = D[(size_t)old * 3 + 1], z1 = D[(size_t)old * 3 + 2];
        float best = -1.f;
        long long key = 0;
        for (int k = t; k < n; k += 1024) {
            const float x2 = D[(size_t)k * 3], y2 = D[(size_t)k * 3 + 1], z2 = D[(size_t)k * 3 + 2];
            const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
            const float d2 = fminf(d, T[k]);
            T[k] = d2;
            const long long kk = ((long long)(__brev((unsigned)(k % block_ref)) >> 1) << 32) | (unsigned)k;
            if (d2 > best || (d2 == best && kk < key)) {
                best = d2; key = kk;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(best, d, 64);
            const long long ok = __shfl_xor(key, d, 64);
            if (ov > best || (ov == best && ok < key)) { best = ov; key = ok; }
        }
        const int buf = r & 1;
        if (l == 0) { s_v[buf][wv] = best; s_k[buf][wv] = key; }
        __syncthreads();
        float v = s_v[buf][l & 15];
        long long kk = s_k[buf][l & 15];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(v, d, 64);
            const long long ok = __shfl_xor(kk, d, 64);
            if (ov > v || (ov == v && ok < kk)) { v = ov; kk = ok; }
        }
        old = (int)(kk & 0xFFFFFFFFll);
        if (t == 0) O[r] = old;
        __syncthreads();   // T[] of this round is complete before the next round reads D[old] / T
    }
}

LIDAR_EXPORT int lidar_furthest_point_sampling(int b, int n, int m, const float *points, float *temp, int *idx, void *stream) {
    if (b <= 0 || n <= 0 || m < 0) return LIDAR_ERR_ARG;
    if (m == 0) return LIDAR_OK;
    if (!points || !temp || !idx) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int block_ref = 1;
    while (block_ref * 2 <= n && block_ref * 2 <= 1024) block_ref *= 2;
    const int items = divup(n, 1024);
    if (n >= 4096 && n <= FB_CAP && m >= 64) {       // large samples: spatial buckets, most of every round is skipped
        hipLaunchKernelGGL(fps_bucket_kernel, dim3(b), dim3(FB_TPB), 0, s, n, m, block_ref - 1, points, temp, idx);
        return lidar_check_launch("lidar_furthest_point_sampling(bucket)");
    }
#define FPS_CASE(J, TPB) hipLaunchKernelGGL((fps_kernel<J, TPB>), dim3(b), dim3(TPB), (size_t)J * 1024 * sizeof(float), s, n, m, block_ref - 1, points, temp, idx)
    if (items <= 2) FPS_CASE(2, 1024);
    else if (items <= 4) FPS_CASE(4, 1024);
    else if (items <= 8) FPS_CASE(8, 1024);
    else if (items <= 16) FPS_CASE(16, 1024);
    else if (items <= 20) FPS_CASE(20, 1024);
    else hipLaunchKernelGGL(fps_generic_kernel, dim3(b), dim3(1024), 0, s, n, m, block_ref, points, temp, idx);
#undef FPS_CASE
    return lidar_check_launch("lidar_furthest_point_sampling");
}

// ------------------------------------------------------------------ 3-NN
// three_nn_kernel_stack (pointnet2_stack/src/interpolate_gpu.cu:16-75) / three_nn_kernel_fast (batch :16-59).
// The reference keeps the running bests in double but only ever stores fp32 distances (or the 1e40 start,
// which every finite fp32 beats and +inf does not): identical to fp32 bests initialised to +inf.
template <bool STACK>
__global__ __launch_bounds__(PN_TPB) void three_nn_kernel(int B, int N, int Mb, const float *__restrict__ unknown,
                                                          const int *__restrict__ unk_cnt, const float *__restrict__ known,
                                                          const int *__restrict__ known_cnt, float *__restrict__ dist2,
                                                          int *__restrict__ idx) {
    __shared__ float s_pts[PN_TILE * 3];
    __shared__ int s_range[2];
    const int t = threadIdx.x;
    int q, bs = 0, start = 0, m = Mb;
    bool valid;
    if (STACK) {
        q = blockIdx.x * PN_TPB + t;
        valid = q < N;
        if (valid) pn_batch_of(unk_cnt, B, q, known_cnt, bs, start, m);
        if (t == 0) {
            int b0, s0, n0, b1, s1, n1;
            const int qa = blockIdx.x * PN_TPB, qb = min(qa + PN_TPB, N) - 1;
            pn_batch_of(unk_cnt, B, qa, known_cnt, b0, s0, n0);
            pn_batch_of(unk_cnt, B, qb, known_cnt, b1, s1, n1);
            s_range[0] = b0;
            s_range[1] = b1;
        }
        __syncthreads();
    } else {
        bs = blockIdx.y;
        q = blockIdx.x * PN_TPB + t;
        valid = q < N;
        start = bs * Mb;
        q += bs * N;
    }
    const int b_lo = STACK ? s_range[0] : bs, b_hi = STACK ? s_range[1] : bs;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (valid) {
        ux = unknown[(size_t)q * 3];
        uy = unknown[(size_t)q * 3 + 1];
        uz = unknown[(size_t)q * 3 + 2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int bb = b_lo; bb <= b_hi; ++bb) {
        int bstart = start, bn = m;
        if (STACK) {
            bstart = 0;
            for (int k = 0; k < bb; ++k) bstart += known_cnt[k];
            bn = known_cnt[bb];
        }
        const bool mine = valid && (bs == bb);
        for (int t0 = 0; t0 < bn; t0 += PN_TILE) {
            const int tn = min(PN_TILE, bn - t0);
            __syncthreads();
            for (int k = t; k < tn * 3; k += PN_TPB) s_pts[k] = known[((size_t)bstart + t0) * 3 + k];
            __syncthreads();
            if (mine) {
                for (int k = 0; k < tn; ++k) {
                    const float d = pn_dist2(ux, uy, uz, s_pts[3 * k], s_pts[3 * k + 1], s_pts[3 * k + 2]);
                    const int gi = t0 + k;
                    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = gi; }
                    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = gi; }
                    else if (d < b3) { b3 = d; i3 = gi; }
                }
            }
        }
    }
    if (valid) {
        const int off = STACK ? start : 0;
        dist2[(size_t)q * 3] = b1; dist2[(size_t)q * 3 + 1] = b2; dist2[(size_t)q * 3 + 2] = b3;
        idx[(size_t)q * 3] = i1 + off; idx[(size_t)q * 3 + 1] = i2 + off; idx[(size_t)q * 3 + 2] = i3 + off;
    }
}

LIDAR_EXPORT int lidar_three_nn_stack(int B, int N, const float *unknown, const int *unknown_batch_cnt, const float *known,
                                      const int *known_batch_cnt, float *dist2, int *idx, void *stream) {
    if (B <= 0 || N < 0) return LIDAR_ERR_ARG;
    if (N == 0) return LIDAR_OK;
    if (!unknown || !unknown_batch_cnt || !known || !known_batch_cnt || !dist2 || !idx) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_nn_kernel<true>, dim3(divup(N, PN_TPB)), dim3(PN_TPB), 0, (hipStream_t)stream, B, N, 0, unknown,
                       unknown_batch_cnt, known, known_batch_cnt, dist2, idx);
    return lidar_check_launch("lidar_three_nn_stack");
}

LIDAR_EXPORT int lidar_three_nn_batch(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                      void *stream) {
    if (b <= 0 || n < 0 || m < 0) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!unknown || !known || !dist2 || !idx) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_nn_kernel<false>, dim3(divup(n, PN_TPB), b), dim3(PN_TPB), 0, (hipStream_t)stream, b, n, m, unknown,
                       (const int *)nullptr, known, (const int *)nullptr, dist2, idx);
    return lidar_check_launch("lidar_three_nn_batch");
}

// ------------------------------------------------------------------ 3-point interpolation
// stack: three_interpolate_kernel_stack (:107-126) / grad (:151-172); features (M, C), out (N, C): channel on the lane
__global__ void three_interp_stack_kernel(int N, int C, const float *__restrict__ feat, const int *__restrict__ idx,
                                          const float *__restrict__ w, float *__restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)N * C) return;
    const int p = (int)(e / C), c = (int)(e - (long long)p * C);
    const int *ii = idx + (size_t)p * 3;
    const float *ww = w + (size_t)p * 3;
    out[e] = ww[0] * feat[(size_t)ii[0] * C + c] + ww[1] * feat[(size_t)ii[1] * C + c] + ww[2] * feat[(size_t)ii[2] * C + c];
}

__global__ void three_interp_grad_stack_kernel(int N, int C, const float *__restrict__ grad_out, const int *__restrict__ idx,
                                               const float *__restrict__ w, float *__restrict__ grad_feat) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)N * C) return;
    const int p = (int)(e / C), c = (int)(e - (long long)p * C);
    const float g = grad_out[e];
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(&grad_feat[(size_t)idx[(size_t)p * 3 + k] * C + c], g * w[(size_t)p * 3 + k]);
}

LIDAR_EXPORT int lidar_three_interpolate_stack(int N, int C, const float *features, const int *idx, const float *weight,
                                               float *out, void *stream) {
    if (N < 0 || C <= 0) return LIDAR_ERR_ARG;
    if (N == 0) return LIDAR_OK;
    if (!features || !idx || !weight || !out) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_interp_stack_kernel, dim3(divup((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, N, C,
                       features, idx, weight, out);
    return lidar_check_launch("lidar_three_interpolate_stack");
}

LIDAR_EXPORT int lidar_three_interpolate_grad_stack(int N, int C, const float *grad_out, const int *idx, const float *weight,
                                                    float *grad_features, void *stream) {
    if (N < 0 || C <= 0) return LIDAR_ERR_ARG;
    if (N == 0) return LIDAR_OK;
    if (!grad_out || !idx || !weight || !grad_features) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_interp_grad_stack_kernel, dim3(divup((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, N, C,
                       grad_out, idx, weight, grad_features);
    return lidar_check_launch("lidar_three_interpolate_grad_stack");
}

// batch (channel-major): three_interpolate_kernel_fast (:84-104) / grad (:127-149); points (b,c,m), out (b,c,n)
__global__ void three_interp_batch_kernel(int b, int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
                                          const float *__restrict__ w, float *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, cc = blockIdx.y, bb = blockIdx.z;
    if (p >= n) return;
    const float *P = points + ((size_t)bb * c + cc) * m;
    const size_t o = ((size_t)bb * n + p) * 3;
    out[((size_t)bb * c + cc) * n + p] = w[o] * P[idx[o]] + w[o + 1] * P[idx[o + 1]] + w[o + 2] * P[idx[o + 2]];
}

__global__ void three_interp_grad_batch_kernel(int b, int c, int n, int m, const float *__restrict__ grad_out,
                                               const int *__restrict__ idx, const float *__restrict__ w,
                                               float *__restrict__ grad_points) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, cc = blockIdx.y, bb = blockIdx.z;
    if (p >= n) return;
    float *G = grad_points + ((size_t)bb * c + cc) * m;
    const size_t o = ((size_t)bb * n + p) * 3;
    const float g = grad_out[((size_t)bb * c + cc) * n + p];
    atomicAdd(&G[idx[o]], g * w[o]);
    atomicAdd(&G[idx[o + 1]], g * w[o + 1]);
    atomicAdd(&G[idx[o + 2]], g * w[o + 2]);
}

LIDAR_EXPORT int lidar_three_interpolate_batch(int b, int c, int m, int n, const float *points, const int *idx,
                                               const float *weight, float *out, void *stream) {
    if (b <= 0 || c <= 0 || m < 0 || n < 0) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!points || !idx || !weight || !out) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_interp_batch_kernel, dim3(divup(n, 256), c, b), dim3(256), 0, (hipStream_t)stream, b, c, m, n, points,
                       idx, weight, out);
    return lidar_check_launch("lidar_three_interpolate_batch");
}

LIDAR_EXPORT int lidar_three_interpolate_grad_batch(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                                    const float *weight, float *grad_points, void *stream) {
    if (b <= 0 || c <= 0 || m < 0 || n < 0) return LIDAR_ERR_ARG;
    if (n == 0) return LIDAR_OK;
    if (!grad_out || !idx || !weight || !grad_points) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(three_interp_grad_batch_kernel, dim3(divup(n, 256), c, b), dim3(256), 0, (hipStream_t)stream, b, c, n, m,
                       grad_out, idx, weight, grad_points);
    return lidar_check_launch("lidar_three_interpolate_grad_batch");
}

// ------------------------------------------------------------------ grouping / gathering (batch, channel-major)
// group_points_kernel_fast (pointnet2_batch/src/group_points_gpu.cu:53-72) / grad (:14-31)
__global__ void group_points_batch_kernel(int b, int c, int n, int np, int ns, const float *__restrict__ points,
                                          const int *__restrict__ idx, float *__restrict__ out, int grad, float *__restrict__ gpoints) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, cc = blockIdx.y, bb = blockIdx.z;
    if (e >= np * ns) return;
    const int src = idx[(size_t)bb * np * ns + e];
    const size_t oo = ((size_t)bb * c + cc) * np * ns + e;
    if (!grad) out[oo] = points[((size_t)bb * c + cc) * n + src];
    else atomicAdd(&gpoints[((size_t)bb * c + cc) * n + src], points[oo]);   // points == grad_out here
}

LIDAR_EXPORT int lidar_group_points_batch(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                                          float *out, void *stream) {
    if (b <= 0 || c <= 0 || n < 0 || npoints < 0 || nsample <= 0) return LIDAR_ERR_ARG;
    if (npoints == 0) return LIDAR_OK;
    if (!points || !idx || !out) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(group_points_batch_kernel, dim3(divup(npoints * nsample, 256), c, b), dim3(256), 0, (hipStream_t)stream, b, c,
                       n, npoints, nsample, points, idx, out, 0, (float *)nullptr);
    return lidar_check_launch("lidar_group_points_batch");
}

LIDAR_EXPORT int lidar_group_points_grad_batch(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                               const int *idx, float *grad_points, void *stream) {
    if (b <= 0 || c <= 0 || n < 0 || npoints < 0 || nsample <= 0) return LIDAR_ERR_ARG;
    if (npoints == 0) return LIDAR_OK;
    if (!grad_out || !idx || !grad_points) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(group_points_batch_kernel, dim3(divup(npoints * nsample, 256), c, b), dim3(256), 0, (hipStream_t)stream, b, c,
                       n, npoints, nsample, grad_out, idx, (float *)nullptr, 1, grad_points);
    return lidar_check_launch("lidar_group_points_grad_batch");
}

// gather_points_kernel_fast (pointnet2_batch/src/sampling_gpu.cu:15-31) / grad (:53-70): == grouping with nsample 1
LIDAR_EXPORT int lidar_gather_points_batch(int b, int c, int n, int npoints, const float *points, const int *idx, float *out,
                                           void *stream) {
    return lidar_group_points_batch(b, c, n, npoints, 1, points, idx, out, stream);
}

LIDAR_EXPORT int lidar_gather_points_grad_batch(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                                float *grad_points, void *stream) {
    return lidar_group_points_grad_batch(b, c, n, npoints, 1, grad_out, idx, grad_points, stream);
}
