// Exact, deterministic top-k of the masked class scores that feeds NMS (post_processing's
// `torch.topk(box_scores, k=min(NMS_PRE_MAXSIZE, n))`, pcdet/models/model_utils/model_nms_utils.py:6-25 via
// pcdet/models/detectors/detector3d_template.py:205-230) — for every frame the k best of N = 321 408 anchor scores, sorted
// descending, ties broken by ascending anchor index (torch.topk leaves the order among equal scores unspecified; real BEV maps
// hold tens of thousands of bit-equal scores — empty regions — so the rule matters and is tested).  Three launches instead of
// torch's 17 (mbtopk radix passes + radixSortKVInPlace), no host synchronisation:
//   1. lidar_anchor_scores_hist   the score kernel of anchor_post.hip, which also counts every valid score (>= valid_min) into
//                                 a 2 048-bin histogram of its frame (LDS-privatised, one global add per bin and workgroup);
//                                 bins are the top bits of key = float_bits(score) - float_bits(valid_min) + 1.
//   2. tk_collect_kernel          TK_W workgroups per frame, each over a contiguous index range: the histogram gives the bin b1
//                                 that holds the k-th score; keys above b1 are selected outright (list A, unordered, wave-
//                                 aggregated appends); keys IN b1 are written, in INDEX ORDER, to the workgroup's own segment.
//   3. tk_finalize_kernel         one workgroup per frame: picks the remaining need = k - |A| best of bin b1 — a bitonic sort
//                                 in LDS when the bin is small (the usual case, ~N / 350 elements), the first `need` in index
//                                 order when the bin is one tie mass (all keys equal: no data pass at all), a 32-bit radix
//                                 select over the segments otherwise — then sorts the k winners (key desc, index asc) and writes
//                                 scores, int64 indices and the per-frame count of valid entries.
// Workspace state that must be zero between calls (histograms, |A| counters) is re-zeroed by launch 3.
#include "common.h"
#include <string.h>

#define TK_BINS 2048
#define TK_W 32            // collect workgroups per frame
#define TK_KMAX 4096       // k <= 4096 (NMS_PRE_MAXSIZE of every reference config)
#define TK_LB 8192         // bin-b1 elements the finalize launch sorts in LDS
#define TK_LA 2048         // selected keys a collect workgroup gathers in LDS before its one global append
#define TK_AS_ITER 16      // x 256 anchors per workgroup of the score + histogram launch
#define TK_NA_STRIDE 64    // ints between two frames' list-A counters (one 256-B line each)

typedef unsigned long long tk_u64;

struct TkWs {
    int *hist;             // [B][TK_BINS]   zero between calls
    int *nA;               // [B][TK_NA_STRIDE] ([0] used) zero between calls
    tk_u64 *A;             // [B][TK_KMAX]   keys above bin b1 (unordered)
    int *cB;               // [B][TK_W]      bin-b1 elements of each collect workgroup
    unsigned *mm;          // [B][TK_W][2]   their smallest / largest key
    tk_u64 *seg;           // [B][TK_W][per] bin-b1 elements of each collect workgroup, in index order
    long long per;         // elements per collect workgroup (multiple of 4)
};

static long long tk_per(long long n) { return ((n + TK_W - 1) / TK_W + 3) / 4 * 4; }

static size_t tk_carve(void *base, int B, long long n, TkWs *w) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return (char *)base + o;
    };
    const long long per = tk_per(n);
    char *p;
    p = take((size_t)B * TK_BINS * 4); if (w) w->hist = (int *)p;
    p = take((size_t)B * TK_NA_STRIDE * 4); if (w) w->nA = (int *)p;
    p = take((size_t)B * TK_KMAX * 8); if (w) w->A = (tk_u64 *)p;
    p = take((size_t)B * TK_W * 4); if (w) w->cB = (int *)p;
    p = take((size_t)B * TK_W * 8); if (w) w->mm = (unsigned *)p;
    p = take((size_t)B * TK_W * per * 8); if (w) w->seg = (tk_u64 *)p;
    if (w) w->per = per;
    return off;
}

// Order-preserving map of a float's bits onto unsigned integers (any sign; -0.0 sorts just below +0.0): positive floats keep their
// order with the sign bit set, negative ones are complemented.  For positive scores and thresholds the differences below equal the
// plain bit-pattern differences the r03 kernels used, so lidar_anchor_scores_hist's histogram (sigmoid scores) stays compatible.
__device__ __host__ __forceinline__ unsigned tk_ord(unsigned bits) { return bits ^ ((bits & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ unsigned tk_unord(unsigned o) { return o ^ ((o & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu); }
// key of a score: 0 = not a candidate (below valid_min: the masked -1; NaN), else ord(score) - ord(valid_min) + 1; base_bits = ord(valid_min)
// (positive floats order like their bit patterns)
__device__ __forceinline__ unsigned tk_key(float s, float valid_min, unsigned base_bits) {
    return (s >= valid_min) ? tk_ord(__float_as_uint(s + 0.0f)) - base_bits + 1u : 0u;      // (-0.0 + 0.0 = +0.0: the two zeros tie, as they compare)
}
__device__ __forceinline__ int tk_bin(unsigned key, int shift) { return (int)min(key >> shift, (unsigned)(TK_BINS - 1)); }
// list entry: sorts descending by (key, then ascending index)
__device__ __forceinline__ tk_u64 tk_entry(unsigned key, unsigned idx) { return ((tk_u64)key << 32) | (tk_u64)(0xFFFFFFFFu - idx); }

// ------------------------------------------------------------------ launch 1: scores + labels + histogram
__device__ __forceinline__ float tk_sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void anchor_scores_hist_kernel(const float *__restrict__ head, long long n_per_frame, int row_stride,
                                                                 int cls_off, int A, int ncls, float thresh, int shift,
                                                                 float *__restrict__ scores, unsigned char *__restrict__ labels,
                                                                 int *__restrict__ hist) {
    __shared__ int s_h[TK_BINS];
    const int t = threadIdx.x, b = blockIdx.y;
    for (int q = t; q < TK_BINS; q += 256) s_h[q] = 0;
    __syncthreads();
    // TK_AS_ITER x 256 consecutive anchors per workgroup: a workgroup flushes one global add per histogram bin it touched, and with
    // 256 anchors per workgroup (20 k workgroups x ~350 bins = 7 M global atomics) the flush tripled the kernel's time
    for (int it = 0; it < TK_AS_ITER; ++it) {
        const long long i = ((long long)blockIdx.x * TK_AS_ITER + it) * 256 + t;
        if (i >= n_per_frame) break;
        // same arithmetic, in the same order, as anchor_scores_kernel (anchor_post.hip)
        const long long g = (long long)b * n_per_frame + i;
        const long long loc = g / A;
        const int a = (int)(g - loc * A);
        const float *p = head + loc * row_stride + cls_off + a * ncls;
        float best = tk_sigmoid_ref(p[0]);
        int bl = 0;
        for (int c = 1; c < ncls; ++c) {
            const float s = tk_sigmoid_ref(p[c]);
            if (s > best) { best = s; bl = c; }
        }
        scores[g] = (best >= thresh) ? best : -1.0f;
        labels[g] = (unsigned char)bl;
        if (best >= thresh) atomicAdd(&s_h[tk_bin(__float_as_uint(best) - __float_as_uint(thresh) + 1u, shift)], 1);
    }
    __syncthreads();
    for (int q = t; q < TK_BINS; q += 256)
        if (s_h[q]) atomicAdd(&hist[(size_t)b * TK_BINS + q], s_h[q]);
}

// histogram only (scores that do not come from lidar_anchor_scores_hist): one pass over (B, N)
__global__ __launch_bounds__(256) void tk_hist_kernel(const float *__restrict__ scores, long long n, float valid_min, unsigned base_bits,
                                                      int shift, int *__restrict__ hist) {
    __shared__ int s_h[TK_BINS];
    const int t = threadIdx.x, b = blockIdx.y;
    for (int q = t; q < TK_BINS; q += 256) s_h[q] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += stride) {
        const unsigned key = tk_key(scores[(size_t)b * n + i], valid_min, base_bits);
        if (key) atomicAdd(&s_h[tk_bin(key, shift)], 1);
    }
    __syncthreads();
    for (int q = t; q < TK_BINS; q += 256)
        if (s_h[q]) atomicAdd(&hist[(size_t)b * TK_BINS + q], s_h[q]);
}

// From the frame's histogram (in LDS): kk = min(k, valid scores), the bin b1 holding the kk-th best key and the number of
// keys in higher bins.  All 1024 threads call it; two barriers inside.  -> sel[0] = b1 (-1 when kk == 0), sel[1] = above, sel[2] = kk
__device__ __forceinline__ void tk_select_bin(const int *s_h, int *s_w, int *sel, int k, int t) {
    const int l = t & 63, wv = t >> 6;
    const int h0 = s_h[TK_BINS - 1 - 2 * t], h1 = s_h[TK_BINS - 2 - 2 * t];       // descending bins
    const int sum = h0 + h1;
    const int inc = wave_incl_scan(sum);
    if (l == 63) s_w[wv] = inc;
    if (t == 0) sel[0] = -1;
    __syncthreads();
    int excl = inc - sum, total = 0;
    for (int q = 0; q < 16; ++q) {
        const int v = s_w[q];
        if (q < wv) excl += v;
        total += v;
    }
    const int kk = min(k, total);
    if (kk > 0) {
        if (excl < kk && kk <= excl + h0) { sel[0] = TK_BINS - 1 - 2 * t; sel[1] = excl; }
        else if (excl + h0 < kk && kk <= excl + sum) { sel[0] = TK_BINS - 2 - 2 * t; sel[1] = excl + h0; }
    }
    if (t == 0) sel[2] = kk;
    __syncthreads();
}

// ------------------------------------------------------------------ launch 2: collect
__global__ __launch_bounds__(1024, 8) void tk_collect_kernel(const float *__restrict__ scores, long long n, int k, float valid_min,
                                                          unsigned base_bits, int shift, TkWs w) {
    __shared__ int s_h[TK_BINS];
    __shared__ int s_w[16], s_wb[16], s_sel[3];
    __shared__ unsigned s_min, s_max;
    __shared__ tk_u64 s_a[TK_LA];
    __shared__ int s_na;
    const int t = threadIdx.x, l = t & 63, wv = t >> 6, wg = blockIdx.x, f = blockIdx.y;
    for (int q = t; q < TK_BINS; q += 1024) s_h[q] = w.hist[(size_t)f * TK_BINS + q];
    if (t == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; s_na = 0; }
    __syncthreads();
    tk_select_bin(s_h, s_w, s_sel, k, t);
    const int b1 = s_sel[0];
    const long long e0 = (long long)wg * w.per, e1 = min(n, e0 + w.per);
    tk_u64 *seg = w.seg + ((size_t)f * TK_W + wg) * w.per;
    tk_u64 *listA = w.A + (size_t)f * TK_KMAX;
    const float *sc = scores + (size_t)f * n;
    int segcount = 0;
    unsigned kmin = 0xFFFFFFFFu, kmax = 0u;
    if (b1 >= 0) {
        for (long long r0 = e0; r0 < e1; r0 += 4096) {             // block-uniform trip count
            const long long i4 = r0 + (long long)t * 4;
            float4 v = make_float4(-1.f, -1.f, -1.f, -1.f);
            if (i4 < e1) v = *reinterpret_cast<const float4 *>(sc + i4);     // n and per are multiples of 4
            const float vs[4] = {v.x, v.y, v.z, v.w};
            unsigned key[4];
            int cntA = 0, cntB = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                key[j] = tk_key(vs[j], valid_min, base_bits);
                const int bin = key[j] ? tk_bin(key[j], shift) : -1;
                cntA += bin > b1;
                cntB += bin == b1;
            }
            // keys above the bin: selected outright, unordered — gathered in LDS first: every wave-round of a frame adding to ONE
            // global counter (1 500 same-line atomics per frame, all frames' counters in one line) measured 226 us per launch
            const int incA = wave_incl_scan(cntA);
            const int totA = __shfl(incA, 63, 64);
            if (totA) {
                int base = 0;
                if (l == 0) base = atomicAdd(&s_na, totA);
                int pos = __shfl(base, 0, 64) + incA - cntA;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (key[j] && tk_bin(key[j], shift) > b1) {
                        const tk_u64 e = tk_entry(key[j], (unsigned)(i4 + j));
                        if (pos < TK_LA) s_a[pos] = e;
                        else {                                     // (more than TK_LA of a frame's < k selected keys in ONE range)
                            const int g = atomicAdd(&w.nA[f * TK_NA_STRIDE], 1);
                            if (g < TK_KMAX) listA[g] = e;
                        }
                        ++pos;
                    }
            }
            // keys in the bin: to this workgroup's segment, in index order (wave scan + scan over the 16 waves)
            const int incB = wave_incl_scan(cntB);
            if (l == 63) s_wb[wv] = incB;
            __syncthreads();
            int pos = segcount + incB - cntB, round = 0;
            for (int q = 0; q < 16; ++q) {
                const int c = s_wb[q];
                if (q < wv) pos += c;
                round += c;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (key[j] && tk_bin(key[j], shift) == b1) {
                    seg[pos++] = tk_entry(key[j], (unsigned)(i4 + j));
                    kmin = min(kmin, key[j]);
                    kmax = max(kmax, key[j]);
                }
            segcount += round;
            __syncthreads();                                       // s_wb is rewritten next round
        }
    }
    if (kmax) {
        atomicMin(&s_min, kmin);
        atomicMax(&s_max, kmax);
    }
    __syncthreads();
    const int na = min(s_na, TK_LA);                               // this workgroup's share of list A: ONE global add, then a copy
    if (t == 0 && na) s_w[0] = atomicAdd(&w.nA[f * TK_NA_STRIDE], na);
    __syncthreads();
    for (int q = t; q < na; q += 1024)
        if (s_w[0] + q < TK_KMAX) listA[s_w[0] + q] = s_a[q];
    if (t == 0) {
        w.cB[(size_t)f * TK_W + wg] = segcount;
        w.mm[((size_t)f * TK_W + wg) * 2] = s_min;
        w.mm[((size_t)f * TK_W + wg) * 2 + 1] = s_max;
    }
}

// ------------------------------------------------------------------ launch 3: finalize
// Bitonic sort, descending, of s[0 .. 1024 * EPT) (LDS), 1024 threads, the elements held in REGISTERS: thread t owns elements
// EPT * t .. EPT * t + EPT - 1.  A stage whose partner distance is below EPT is a register compare-exchange inside the thread;
// below 64 * EPT the partner sits in another lane of the same wave (two 32-bit lane exchanges per element); only the stages
// with a partner in another wave (10 of the 78 at 4096 elements) go through LDS — written in a lane-contiguous layout, one
// barrier pair each.  (The first version kept the list in LDS and did one read-compare-write round trip per stage and pair:
// 930 cycles per stage, 72 k cycles for the 4096-element sort.)
__device__ __forceinline__ tk_u64 tk_shfl_xor64(tk_u64 v, int lane_mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, lane_mask, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), lane_mask, 64);
    return ((tk_u64)hi << 32) | lo;
}
// An element at stage (size, stride) keeps the larger of (own, partner) when its subsequence runs descending and it is the lower
// index of the pair, or the subsequence runs ascending and it is the upper one.  For partners in other threads (stride >= EPT)
// that verdict is the same for all EPT elements of a thread: computed once per stage (`want_max`), which leaves one 64-bit
// compare and two selects per element — the sort is bound by the instruction rate of the ONE CU a frame's workgroup runs on
// (4 waves per SIMD: 15 instructions per element and stage were 1 000 cycles per stage).
__device__ __forceinline__ tk_u64 tk_keep(tk_u64 own, tk_u64 other, bool want_max) { return ((own > other) == want_max) ? own : other; }
template <int EPT, int S>
__device__ __forceinline__ void tk_local_stage(tk_u64 (&v)[EPT], int size, int e0) {
#pragma unroll
    for (int r = 0; r < EPT; ++r)
        if ((r & S) == 0) {
            const tk_u64 a = v[r], b = v[r | S];
            const bool desc = size < EPT ? ((r & size) == 0) : ((e0 & size) == 0);
            const bool sw = (a > b) != desc;               // exchange when the pair is not in the subsequence's order
            v[r] = sw ? b : a;
            v[r | S] = sw ? a : b;
        }
}
template <int EPT>
__device__ __forceinline__ void tk_sort_desc_regs(tk_u64 *s, int t) {
    constexpr int n = 1024 * EPT;
    tk_u64 v[EPT];
    const int e0 = EPT * t;
    __syncthreads();                                               // (other threads wrote the list)
#pragma unroll
    for (int r = 0; r < EPT; ++r) v[r] = s[e0 + r];
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64 * EPT) {                              // partner in another wave
                __syncthreads();                                   // (the last readers of the exchange buffer are done)
#pragma unroll
                for (int r = 0; r < EPT; ++r) s[r * 1024 + t] = v[r];
                __syncthreads();
                const int tp = t ^ (stride / EPT);
                const bool want_max = ((e0 & size) == 0) == ((e0 & stride) == 0);
#pragma unroll
                for (int r = 0; r < EPT; ++r) v[r] = tk_keep(v[r], s[r * 1024 + tp], want_max);
            } else if (stride >= EPT) {                            // partner in another lane
                const int lm = stride / EPT;
                const bool want_max = ((e0 & size) == 0) == ((e0 & stride) == 0);
#pragma unroll
                for (int r = 0; r < EPT; ++r) v[r] = tk_keep(v[r], tk_shfl_xor64(v[r], lm), want_max);
            } else if (EPT > 1 && stride == 1) tk_local_stage<EPT, 1>(v, size, e0);
            else if (EPT > 2 && stride == 2) tk_local_stage<EPT, 2>(v, size, e0);
            else if (EPT > 4 && stride == 4) tk_local_stage<EPT, 4>(v, size, e0);
        }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) s[e0 + r] = v[r];
    __syncthreads();
}

__global__ __launch_bounds__(1024) void tk_finalize_kernel(long long n, int k, unsigned base_bits, float fill, TkWs w, float *__restrict__ top_scores,
                                                           long long *__restrict__ top_idx, int *__restrict__ counts) {
    __shared__ tk_u64 s_list[TK_KMAX];
    __shared__ tk_u64 s_b[TK_LB];
    __shared__ int s_h[TK_BINS];
    __shared__ int s_w[16], s_wb[16], s_sel[3];
    __shared__ int s_pref[TK_W + 1];
    __shared__ int s_cnt, s_digit, s_above;
    __shared__ unsigned s_min, s_max;
    const int t = threadIdx.x, l = t & 63, wv = t >> 6, f = blockIdx.x;
#ifdef TK_STAMPS   // tools/topk_phase_probe.py: shader-clock stamps of frame 0's phases, over frame 1's (already consumed) min/max words
#define TK_STAMP(k) do { if (f == 0 && t == 0) w.mm[TK_W * 2 + (k)] = (unsigned)clock64(); } while (0)
#else
#define TK_STAMP(k) do { } while (0)
#endif
    TK_STAMP(0);
    for (int q = t; q < TK_BINS; q += 1024) {
        s_h[q] = w.hist[(size_t)f * TK_BINS + q];
        w.hist[(size_t)f * TK_BINS + q] = 0;                       // clean for the next call
    }
    if (t < TK_W) {                                                // one lane per collect workgroup: its count and key range
        const int c = w.cB[(size_t)f * TK_W + t];
        const unsigned mn = w.mm[((size_t)f * TK_W + t) * 2], mx = w.mm[((size_t)f * TK_W + t) * 2 + 1];
        const int inc = wave_incl_scan(c);                         // (TK_W <= 64: one wave)
        s_pref[t] = inc - c;
        if (t == TK_W - 1) s_pref[TK_W] = inc;
        unsigned lo = mn, hi = mx;
#pragma unroll
        for (int d = 1; d < TK_W; d <<= 1) {
            lo = min(lo, (unsigned)__shfl_xor((int)lo, d, 64));
            hi = max(hi, (unsigned)__shfl_xor((int)hi, d, 64));
        }
        if (t == 0) {
            s_min = lo; s_max = hi;
            w.nA[f * TK_NA_STRIDE] = 0;                            // clean for the next call (its value equals `above`)
        }
    }
    __syncthreads();
    TK_STAMP(1);
    tk_select_bin(s_h, s_w, s_sel, k, t);
    TK_STAMP(2);
    const int above = s_sel[0] >= 0 ? s_sel[1] : 0, kk = s_sel[2];
    const int need = kk - above, totalB = s_pref[TK_W];
    const tk_u64 *segs = w.seg + (size_t)f * TK_W * w.per;
    for (int q = t; q < TK_KMAX; q += 1024) s_list[q] = (q < above) ? w.A[(size_t)f * TK_KMAX + q] : 0ull;
    __syncthreads();
    TK_STAMP(3);
    if (need > 0) {
        if (totalB <= TK_LB) {
            // ---- the usual case: the whole bin fits in LDS — sort it, keep its `need` best
            int np2 = 1024;
            while (np2 < totalB) np2 <<= 1;
            for (int q = t; q < np2; q += 1024) {
                tk_u64 e = 0ull;
                if (q < totalB) {
                    int wq = 0;
                    for (int step = TK_W >> 1; step > 0; step >>= 1)
                        if (s_pref[wq + step] <= q) wq += step;
                    e = segs[(size_t)wq * w.per + (q - s_pref[wq])];
                }
                s_b[q] = e;
            }
            TK_STAMP(4);
            if (np2 == 1024) tk_sort_desc_regs<1>(s_b, t);
            else if (np2 == 2048) tk_sort_desc_regs<2>(s_b, t);
            else if (np2 == 4096) tk_sort_desc_regs<4>(s_b, t);
            else tk_sort_desc_regs<8>(s_b, t);
            TK_STAMP(5);
            for (int q = t; q < need; q += 1024) s_list[above + q] = s_b[q];
        } else if (s_min == s_max) {
            // ---- one tie mass: every key of the bin is equal — the first `need` in index order (segments are index-ordered)
            for (int q = t; q < need; q += 1024) {
                int wq = 0;
                for (int step = TK_W >> 1; step > 0; step >>= 1)
                    if (s_pref[wq + step] <= q) wq += step;
                s_list[above + q] = segs[(size_t)wq * w.per + (q - s_pref[wq])];
            }
        } else {
            // ---- a large bin of distinct keys (never seen on BEV maps; kept exact): 32-bit radix select over the segments,
            // three digits of 11 / 11 / 10 bits, then everything above the threshold key plus the first equals in index order
            unsigned prefix = 0u;
            int remaining = need;
            const int dshift[3] = {21, 10, 0}, dbits[3] = {11, 11, 10};
            for (int lvl = 0; lvl < 3; ++lvl) {
                for (int q = t; q < TK_BINS; q += 1024) s_h[q] = 0;
                __syncthreads();
                const unsigned himask = lvl == 0 ? 0u : ~0u << (dshift[lvl] + dbits[lvl]);
                for (int q = t; q < totalB; q += 1024) {
                    int wq = 0;
                    for (int step = TK_W >> 1; step > 0; step >>= 1)
                        if (s_pref[wq + step] <= q) wq += step;
                    const unsigned key = (unsigned)(segs[(size_t)wq * w.per + (q - s_pref[wq])] >> 32);
                    if ((key & himask) == prefix) atomicAdd(&s_h[(key >> dshift[lvl]) & ((1u << dbits[lvl]) - 1u)], 1);
                }
                __syncthreads();
                if (t == 0) {                                      // (2 048 bins, once per level: serial is fine on this path)
                    int acc = 0, d = (1 << dbits[lvl]) - 1;
                    for (; d > 0; --d) {
                        if (acc + s_h[d] >= remaining) break;
                        acc += s_h[d];
                    }
                    s_digit = d;
                    s_above = acc;
                }
                __syncthreads();
                prefix |= (unsigned)s_digit << dshift[lvl];
                remaining -= s_above;
                __syncthreads();
            }
            const unsigned T = prefix;                             // the need-th best key; `remaining` of the keys equal to it are taken
            if (t == 0) s_cnt = 0;
            __syncthreads();
            int taken_eq = 0;
            for (int r0 = 0; r0 < totalB; r0 += 1024) {            // block-uniform
                const int q = r0 + t;
                tk_u64 e = 0ull;
                if (q < totalB) {
                    int wq = 0;
                    for (int step = TK_W >> 1; step > 0; step >>= 1)
                        if (s_pref[wq + step] <= q) wq += step;
                    e = segs[(size_t)wq * w.per + (q - s_pref[wq])];
                }
                const unsigned key = (unsigned)(e >> 32);
                if (q < totalB && key > T) s_list[above + atomicAdd(&s_cnt, 1)] = e;           // fewer than need - remaining of them
                const int iseq = (q < totalB && key == T) ? 1 : 0;
                const unsigned long long bal = __ballot(iseq);
                if (l == 0) s_wb[wv] = __popcll(bal);
                __syncthreads();
                int rank = taken_eq + __popcll(bal & lanemask_lt()), round = 0;
                for (int x = 0; x < 16; ++x) {
                    const int c = s_wb[x];
                    if (x < wv) rank += c;
                    round += c;
                }
                if (iseq && rank < remaining) s_list[above + (need - remaining) + rank] = e;
                taken_eq += round;
                __syncthreads();
            }
        }
    }
    TK_STAMP(6);
    tk_sort_desc_regs<TK_KMAX / 1024>(s_list, t);
    TK_STAMP(7);
    for (int q = t; q < k; q += 1024) {
        const tk_u64 e = s_list[q];
        const bool ok = q < kk;
        top_scores[(size_t)f * k + q] = ok ? __uint_as_float(tk_unord((unsigned)(e >> 32) - 1u + base_bits)) : fill;
        top_idx[(size_t)f * k + q] = ok ? (long long)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull)) : 0ll;
    }
    if (t == 0) counts[f] = kk;
    TK_STAMP(8);
}

__global__ void tk_zero_kernel(int *p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// ------------------------------------------------------------------ C ABI
static int tk_shift_of(float valid_min, float score_max) {
    unsigned a, b;
    memcpy(&a, &valid_min, 4);
    memcpy(&b, &score_max, 4);
    const unsigned long long span64 = (unsigned long long)tk_ord(b) - (unsigned long long)tk_ord(a) + 2ull;   // largest key of a score <= score_max, + 1
    const unsigned span = span64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)span64;
    int bits = 0;
    while ((span >> bits) != 0u && bits < 32) ++bits;
    return bits > 11 ? bits - 11 : 0;
}

LIDAR_EXPORT size_t lidar_topk_workspace_bytes(int batch, long long n) {
    if (batch <= 0 || n <= 0) return 0;
    return tk_carve(nullptr, batch, n, nullptr);
}

// once per workspace buffer (and after any failed call): the histograms / counters start out zero
LIDAR_EXPORT int lidar_topk_workspace_init(void *ws, size_t ws_bytes, int batch, long long n, void *stream) {
    TkWs w;
    if (!ws || batch <= 0 || n <= 0 || tk_carve(ws, batch, n, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    const long long cells = (long long)batch * TK_BINS;
    hipLaunchKernelGGL(tk_zero_kernel, dim3(divup(cells, 256)), dim3(256), 0, (hipStream_t)stream, w.hist, cells);
    hipLaunchKernelGGL(tk_zero_kernel, dim3(divup((long long)batch * TK_NA_STRIDE, 256)), dim3(256), 0, (hipStream_t)stream, w.nA,
                       (long long)batch * TK_NA_STRIDE);
    return lidar_check_launch("lidar_topk_workspace_init");
}

LIDAR_EXPORT int lidar_anchor_scores_hist(const float *head, int batch, long long locs_per_frame, int row_stride, int cls_off,
                                          int anchors_per_loc, int num_class, float score_thresh, float *scores,
                                          unsigned char *labels, void *ws, size_t ws_bytes, void *stream) {
    if (batch <= 0 || locs_per_frame <= 0 || anchors_per_loc <= 0 || num_class <= 0 || cls_off < 0 ||
        cls_off + anchors_per_loc * num_class > row_stride || !(score_thresh > 0.f))
        return LIDAR_ERR_ARG;
    if (!head || !scores || !labels || !ws) return LIDAR_ERR_ARG;
    const long long n = locs_per_frame * anchors_per_loc;
    TkWs w;
    if (tk_carve(ws, batch, n, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    const long long blocks = (n + 256 * TK_AS_ITER - 1) / (256 * TK_AS_ITER);
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(anchor_scores_hist_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, (hipStream_t)stream, head, n, row_stride,
                       cls_off, anchors_per_loc, num_class, score_thresh, tk_shift_of(score_thresh, 1.0f), scores, labels, w.hist);
    return lidar_check_launch("lidar_anchor_scores_hist");
}

// scores (batch, n) f32 -> top_scores (batch, k) descending, top_idx (batch, k) i64, counts (batch) = entries >= valid_min.
// Candidates are the scores >= valid_min (any sign, -inf = every non-NaN score; NaNs are never candidates); slots past counts[b] hold
// (-1, 0) when valid_min > 0 (the masked-score convention of class_agnostic_nms's callers) and (-inf, 0) otherwise.  score_max: an
// upper bound of the scores (it only shapes the histogram bins: results are exact for any input).  hist_ready != 0: the workspace
// histogram was filled by lidar_anchor_scores_hist with score_thresh == valid_min (> 0) and score_max == 1 (sigmoid outputs).
LIDAR_EXPORT int lidar_topk_desc(const float *scores, int batch, long long n, int k, float valid_min, float score_max, int hist_ready,
                                 float *top_scores, long long *top_idx, int *counts, void *ws, size_t ws_bytes, void *stream) {
    if (batch <= 0 || n <= 0 || k <= 0 || k > TK_KMAX || (n & 3) || n > 0x7fffffffll || !(score_max >= valid_min) ||
        (hist_ready && !(valid_min > 0.f)))
        return LIDAR_ERR_ARG;
    if (!scores || !top_scores || !top_idx || !counts || !ws || (reinterpret_cast<uintptr_t>(scores) & 15)) return LIDAR_ERR_ARG;
    TkWs w;
    if (tk_carve(ws, batch, n, &w) > ws_bytes) return LIDAR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned base_bits;
    const float vm0 = valid_min + 0.0f;                   // a threshold of -0.0 is +0.0
    memcpy(&base_bits, &vm0, 4);
    base_bits = tk_ord(base_bits);
    const float fill = valid_min > 0.f ? -1.0f : -INFINITY;
    const int shift = tk_shift_of(valid_min, score_max);
    if (!hist_ready)
        hipLaunchKernelGGL(tk_hist_kernel, dim3(256, batch), dim3(256), 0, s, scores, n, valid_min, base_bits, shift, w.hist);
    hipLaunchKernelGGL(tk_collect_kernel, dim3(TK_W, batch), dim3(1024), 0, s, scores, n, k, valid_min, base_bits, shift, w);
    hipLaunchKernelGGL(tk_finalize_kernel, dim3(batch), dim3(1024), 0, s, n, k, base_bits, fill, w, top_scores, top_idx, counts);
    return lidar_check_launch("lidar_topk_desc");
}
