// 3x3 / stride 1 / pad 1 fp32 convolution on NHWC maps as Winograd F(2x2, 3x3) on the fp32 matrix cores, with the folded
// BatchNorm shift + ReLU in its epilogue and the output written at a channel offset of a wider NHWC map — the stride-1 layers of
// BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:34-45: Conv2d(c, c, 3, padding=1, bias=False) / BatchNorm2d /
// ReLU, LAYER_NUMS of them per block), which are 87 % of the PointPillar step's FLOPs and 80 % of SECOND's.
//
// Why Winograd here: the fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense) is the slowest matrix instruction of the part, so a
// direct implicit GEMM is bound by it (the library's asm kernel reaches 0.8 of that peak) while every other pipe idles.  F(2x2, 3x3)
// needs 16 multiplies per 2x2 output tile and (cin, cout) pair instead of 36 — 2.25x fewer MFMA cycles — and pays with VALU adds
// (the input / output transforms), which run in the MFMA's shadow.  Error: the transforms use only +-1 and +-1/2, |error| ~ 1e-6
// of the output scale (asserted against the direct convolution at 1e-4, the north_star tolerance, in tests/test_gpu_wino.py).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      g: 3x3 filter, d: 4x4 input patch of a 2x2 output tile (origin = tile - pad)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// For each of the 16 transform positions xi the sum over cin is a plain GEMM  M_xi[tile, cout] = V_xi[tile, cin] @ U_xi[cin, cout].
// A WAVE owns 32 tiles (4 x 8 tiles = 8 x 16 output pixels; "tall": 8 x 4) x 32 output channels x all 16 positions: 16 accumulator
// tiles of v_mfma_f32_32x32x2_f32 = 256 accumulator registers — one wave per SIMD, 512 registers each.  In the accumulator layout
// the 16 positions of one (tile, cout) sit in the SAME lane at the same register index of the 16 tiles, so the output transform is
// lane-local arithmetic.  Workgroup = 4 waves = MW x NW (tile blocks x groups of 32 output channels): 2 x 2 for 64 output channels,
// 1 x 4 from 128.  Channels are consumed in chunks of 4 (two MFMA K-steps, 32 MFMAs = 2 048 matrix-pipe cycles per wave).
//   A operand: lane (i = lane & 31, h = lane >> 5) supplies tile i, input channel 4c + 2h + s of chunk c, K-step s.  The input
//              transform of a tile block is computed ONCE per workgroup: the NW waves that share a block each take 4 / NW of the four
//              transform rows (raw pixels from the LDS image of the region, 8 reads + 8 adds + 4 writes per row) and leave V in LDS
//              in A-operand lane order; every wave then reads its 16 operands as 8-byte lane-linear loads.
//   B operand: the transformed filters are packed once per weight update (lidar_wino_pack_weights) in exactly the order the lanes
//              consume them — [chunk][cout / 32][xi / 2][lane][xi & 1][s] — a wave's load is one contiguous 1 KB (dwordx4 per lane).
//   LDS image of the input region ((2 TY MW + 2) x (2 TX + 2) pixels x 4 channels per chunk): filled by LDS-DMA
//              (global_load_lds_dwordx4, one pixel per lane; zero padding = out-of-image lanes read a zero word).
// Pipeline (r04 v3), per chunk c: MFMAs(c) | A operands of c + 1 read from V | transform of c + 2 | DMA of c + 3, ONE barrier per
// chunk, two-buffer rings for the raw images and for V.  One wave per SIMD has nobody to hide behind, so (measured with the
// -DWINO_PROBE ablations, tools/wino_probe.py):
//   * the issue order inside a chunk is written out MFMA by MFMA with scheduling fences (the compiler's own order, or
//     sched_group_barrier hints, cost 7-12 %): a vector-memory instruction costs 30-60 issue cycles here, so the 8 filter loads go one
//     per MFMA gap; the transform's reads, arithmetic and writes follow; the second half of a chunk is bare MFMAs, so the barrier's
//     wait finds everything complete;
//   * the kernel is PERSISTENT and the pipeline never drains: a workgroup walks its list of tile blocks and the DMA / transform /
//     operand reads of the next block's first chunks run under the last chunks of the current one (the per-block prologue — two
//     dependent DMA round trips and two barriers — was 6-12 % of a block); a block's first MFMAs take a zero C operand instead of
//     cleared accumulators;
//   * the 2 x 2 output tiles go through a per-wave LDS staging tile and leave as 16-byte stores of 4 channels (v1: one 4-byte
//     store per value, 23 % of the 64-channel layer).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WINO_PROBE                       // timing probes (tools/wino_probe.py; WRONG results): bit 0 no input DMA in the loop, bit 1 no
#define WINO_PROBE 0                     // filter loads in the loop, bit 2 no patch reads / transform in the loop, bit 3 no epilogue,
#endif                                   // bit 4 no barrier in the loop
#ifndef WINO_PERSIST                     // A/B: 0 = one tile block per workgroup launch slot (the pipeline drains at every block)
#define WINO_PERSIST 1
#endif

__device__ float4 g_wino_zero = {0.f, 0.f, 0.f, 0.f};      // what an out-of-image pixel reads (zero padding)

// ------------------------------------------------------------------ filter transform + packing
// w: (Cout, Cin, 3, 3) contiguous.
// upk[((((c * NB + nb) * 8 + xi / 2) * 64 + lane) * 2 + (xi & 1)) * 2 + s] = U_xi[cin = 4c + 2 (lane >> 5) + s][cout = 32 nb + (lane & 31)]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float *__restrict__ w, int Cin, int Cout, float *__restrict__ upk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int cout = idx / Cin, cin = idx - cout * Cin;
    const float *g = w + ((size_t)cout * Cin + cin) * 9;
    float t[4][3];                       // G g
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const float g0 = g[x], g1 = g[3 + x], g2 = g[6 + x];
        t[0][x] = g0;
        t[1][x] = 0.5f * (g0 + g1 + g2);
        t[2][x] = 0.5f * (g0 - g1 + g2);
        t[3][x] = g2;
    }
    const int c = cin >> 2, h = (cin >> 1) & 1, s = cin & 1, nb = cout >> 5, j = cout & 31, NB = Cout >> 5;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const float u[4] = {t[y][0], 0.5f * (t[y][0] + t[y][1] + t[y][2]), 0.5f * (t[y][0] - t[y][1] + t[y][2]), t[y][2]};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int xi = y * 4 + x;
            upk[(((((size_t)c * NB + nb) * 8 + (xi >> 1)) * 64 + (h * 32 + j)) * 2 + (xi & 1)) * 2 + s] = u[x];
        }
    }
}

// ------------------------------------------------------------------ the convolution
struct WinoArgs {
    const float *in;         // (B, H, W, Cin)
    const float *upk;        // packed transformed filters
    const float *bias;       // (Cout) or null
    float *out;              // (B, H, W, out_C), this layer's channels at [out_off, out_off + Cout)
    int B, H, W, Cin, Cout, out_C, out_off, relu;
    const int *grp_cout, *grp_ooff;                  // grouped + compact output: real output channels and first output channel of every group (null: 32 each, at 32 g)
    int in_C, in_goff;                               // pixel pitch of `in` (floats) and input-channel offset per 32-output-channel block (grouped: its group's first channel; else 0)
    int blocks_y, blocks_x, n_groups, n_blocks;       // grid decomposition (n_blocks = B * blocks_y * blocks_x * n_groups)
};

#define WINO_STG_PITCH 36                 // floats per staged output pixel (32 channels + 4: 16-byte aligned rows, spread over banks)

// LDS bytes of wino_f23_kernel<NW, TALL>: raw ring [2][RP] float4 + V ring [2][MW][16][64] float2 + output staging [4][128][pitch]
static constexpr int wino_rp(int NW, bool TALL) {
    return ((((TALL ? 16 : 8) * (4 / NW) + 2) * ((TALL ? 8 : 16) + 2) + 63) / 64) * 64;
}
static constexpr size_t wino_lds_bytes(int NW, bool TALL) {
    return (size_t)2 * wino_rp(NW, TALL) * 16 + (size_t)2 * (4 / NW) * 16 * 64 * 8 + (size_t)4 * 128 * WINO_STG_PITCH * 4;
}

template <int NW, bool TALL>
__global__ __launch_bounds__(256) void wino_f23_kernel(const WinoArgs a) {
    constexpr int MW = 4 / NW;
    constexpr int TYW = TALL ? 8 : 4, TXW = TALL ? 4 : 8;   // tiles per wave: rows, columns
    constexpr int RW = 2 * TXW + 2;                       // region width (pixels)
    constexpr int RH = 2 * TYW * MW + 2;                  // region height
    constexpr int RP = wino_rp(NW, TALL);                 // region pixels, padded to whole DMA wave-instructions
    constexpr int NQ = RP / 64;                           // DMA wave-instructions per chunk and workgroup
    constexpr int QW = (NQ + 3) / 4;                      // ... per wave
    constexpr int RPW = 4 / NW;                           // transform rows per wave
    extern __shared__ float4 s_mem4[];
    float4 *s_raw = s_mem4;                                                   // [2][RP]: [pixel slot][4 channels of the chunk]
    f32x2 *s_v = reinterpret_cast<f32x2 *>(s_mem4 + 2 * RP);                  // [2][MW][16][64]: V in A-operand lane order
    float *s_stg = reinterpret_cast<float *>(s_v + 2 * MW * 16 * 64);         // [4 waves][128 pixels][WINO_STG_PITCH]

    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);     // (wave-uniform, and known to be)
    const int mw = wv / NW, nw = wv % NW;
    const int i = l & 31, h = l >> 5;
    const int ty = TALL ? (i >> 2) : (i >> 3), tx = TALL ? (i & 3) : (i & 7);
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int NB = a.Cout >> 5;
    const size_t bstride = (size_t)NB * 8 * 64;          // float4 per chunk

    // ---- the list of tile blocks of this workgroup.  XCD-aware: the hardware deals consecutive workgroup ids round-robin over
    // the 8 XCDs; blocks that share input (the channel groups of one spatial block, horizontally adjacent blocks: halo) should
    // share an L2, so XCD x owns the LOGICAL blocks [x * nb8, (x + 1) * nb8) and its workgroups stride through them
    const int nb8 = (a.n_blocks + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int blk_end = min((xcd + 1) * nb8, a.n_blocks);
    int blk = xcd * nb8 + slot;
    if (blk >= blk_end) return;

    struct Tile {
        const float *src[QW];            // this lane's DMA sources (region slots q * 64 + l for q = wv, wv + 4, ...), chunk 0
        bool ok[QW];                     // ... inside the image (else: the zero word, never advanced)
        const f32x4 *b;                  // this lane's packed filters, chunk 0
        int bidx, by, bx, nb;
    };
    auto make_tile = [&](int blk_) {
        Tile tl;
        const int ng = blk_ % a.n_groups;
        blk_ /= a.n_groups;
        tl.bx = blk_ % a.blocks_x;
        blk_ /= a.blocks_x;
        tl.by = blk_ % a.blocks_y;
        tl.bidx = blk_ / a.blocks_y;
        tl.nb = ng * NW + nw;
        tl.b = reinterpret_cast<const f32x4 *>(a.upk) + ((size_t)tl.nb * 8) * 64 + l;
        const int R0 = 2 * TYW * MW * tl.by - 1, C0 = 2 * TXW * tl.bx - 1;      // image coordinates of region pixel (0, 0)
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int q = wv + 4 * k;
            const int p = q * 64 + l;
            const int ry = p / RW, rx = p - ry * RW;
            const int gy = R0 + ry, gx = C0 + rx;
            tl.ok[k] = (q < NQ) && (p < RH * RW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            tl.src[k] = tl.ok[k] ? a.in + (((size_t)tl.bidx * H + gy) * W + gx) * a.in_C + (size_t)tl.nb * a.in_goff : (const float *)&g_wino_zero;
        }
        return tl;
    };
    auto dma = [&](const Tile &tl, int c, int buf) {
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int q = wv + 4 * k;
            if (q < NQ) {                                 // wave-uniform
                const float *g = tl.ok[k] ? tl.src[k] + 4 * c : tl.src[k];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(s_raw + buf * RP + q * 64), 16, 0, 0);
            }
        }
    };
    // ---- input transform, this wave's share: rows xy = nw * RPW + rr of V = B^T d B for the tile block mw, lane = (tile i, channel
    // pair h) as in the A operand.  Row xy of B^T d is d[ra] + sg * d[rb] (sg = +-1: the fma is exact)
    const int slot0 = (2 * TYW * mw + 2 * ty) * RW + 2 * tx;
    int txy[RPW], tra[RPW], trb[RPW];
    float tsg[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int xy = nw * RPW + rr;                     // wave-uniform
        txy[rr] = xy;
        tra[rr] = xy == 0 ? 0 : (xy == 2 ? 2 : 1);
        trb[rr] = xy < 2 ? 2 : (xy == 2 ? 1 : 3);
        tsg[rr] = xy == 1 ? 1.f : -1.f;
    }
    auto transform = [&](int rbuf, int vbuf) {
        const f32x2 *base = reinterpret_cast<const f32x2 *>(s_raw + rbuf * RP + slot0) + h;
        f32x2 *vdst = s_v + ((size_t)(vbuf * MW + mw) * 16) * 64 + l;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            f32x2 tr[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x2 da = base[(tra[rr] * RW + x) * 2], db = base[(trb[rr] * RW + x) * 2];
                tr[x][0] = __builtin_fmaf(tsg[rr], db[0], da[0]);
                tr[x][1] = __builtin_fmaf(tsg[rr], db[1], da[1]);
            }
            vdst[(txy[rr] * 4 + 0) * 64] = tr[0] - tr[2];
            vdst[(txy[rr] * 4 + 1) * 64] = tr[1] + tr[2];
            vdst[(txy[rr] * 4 + 2) * 64] = tr[2] - tr[1];
            vdst[(txy[rr] * 4 + 3) * 64] = tr[1] - tr[3];
        }
    };
    auto load_v = [&](int vbuf, f32x2 (&v)[16]) {
        const f32x2 *vsrc = s_v + ((size_t)(vbuf * MW + mw) * 16) * 64 + l;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) v[xi] = vsrc[xi * 64];
    };
    auto load_b = [&](const Tile &tl, int c, f32x4 (&bb)[8]) {
        const f32x4 *p = tl.b + (size_t)c * bstride;
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = p[e * 64];
    };

    f32x16 acc[16];
    const int NC = Cin >> 2;
    f32x4 b0[8], b1[8];
    f32x2 v0[16], v1[16];
    Tile cur = make_tile(blk), nxt = cur;

    // ---- pipeline fill (first block only): raw(0..2), V(0), V(1), the A operands and filters of chunk 0
    dma(cur, 0, 0);
    dma(cur, 1, 1);
    load_b(cur, 0, b0);
    __syncthreads();                                      // raw(0), raw(1) landed (vmcnt(0) + barrier)
    transform(0, 0);
    __syncthreads();                                      // V(0) visible; raw[0] free
    if (2 < NC) dma(cur, 2, 0);
    transform(1, 1);
    load_v(0, v0);
    __syncthreads();                                      // V(1) visible, raw(2) landed; raw[1] free
#if WINO_PROBE
    load_b(cur, 0, b1);
    load_v(1, v1);
#endif

    // chunk c (parity P = c & 1, NC even): DMA raw(c + 3) -> raw[!P]; A operands of c + 1 from V[!P]; transform raw(c + 2) in raw[P]
    // -> V[P]; filters of c + 1.  Past the end of the block, "c + k" means chunk c + k - NC of the NEXT block (same parities).
    // FIRST: the block's first chunk — its K-step 0 starts the accumulators from a zero C operand.
#define WINO_CHUNK(c, P, FIRST, VC, BC, VN, BN)                                                                                    \
    {                                                                                                                              \
        if (!(WINO_PROBE & 16)) {                                                                                                  \
            if ((c) > 0) __syncthreads();                                                                                          \
            else if (!first_block) {       /* block boundary: everything this barrier orders is LDS traffic or was drained before  \
                                              the epilogue; do NOT wait for the epilogue's stores here (vmcnt untouched) */       \
                __builtin_amdgcn_s_waitcnt(0xC07F);        /* lgkmcnt(0) */                                                        \
                __builtin_amdgcn_s_barrier();                                                                                      \
            }                                                                                                                      \
        }                                                                                                                          \
        if (!(WINO_PROBE & 1)) {                                                                                                   \
            if ((c) + 3 < NC) dma(cur, (c) + 3, 1 - (P));                                                                          \
            else if (has_next) dma(nxt, (c) + 3 - NC, 1 - (P));                                                                    \
        }                                                                                                                          \
        const f32x4 *bp_ = ((c) + 1 < NC) ? cur.b + (size_t)((c) + 1) * bstride : (has_next ? nxt.b : cur.b);                      \
        const f32x2 *rb_ = reinterpret_cast<const f32x2 *>(s_raw + (P) * RP + slot0) + h;                                          \
        const f32x2 *vs_ = s_v + ((size_t)((1 - (P)) * MW + mw) * 16) * 64 + l;                                                    \
        f32x2 *vd_ = s_v + ((size_t)((P) * MW + mw) * 16) * 64 + l;                                                                \
        f32x2 da_[RPW][4], db_[RPW][4], tr_[RPW][4];                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                                           \
            const float bop_ = BC[(k & 15) >> 1][((k & 1) << 1) + (k >> 4)];                                                       \
            if (FIRST && k < 16) {                                                                                                 \
                f32x16 z_;                                                                                                         \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) z_[r_] = 0.f;                                                    \
                acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(VC[k & 15][k >> 4], bop_, z_, 0, 0, 0);                         \
            } else {                                                                                                               \
                acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(VC[k & 15][k >> 4], bop_, acc[k & 15], 0, 0, 0);                \
            }                                                                                                                      \
            if (k < 8 && !(WINO_PROBE & 2)) BN[k] = bp_[k * 64];                                                                   \
            if (k < 4 && !(WINO_PROBE & 4)) {                                                                                      \
                _Pragma("unroll") for (int rr = 0; rr < RPW; ++rr) {                                                               \
                    da_[rr][k] = rb_[(tra[rr] * RW + k) * 2];                                                                      \
                    db_[rr][k] = rb_[(trb[rr] * RW + k) * 2];                                                                      \
                }                                                                                                                  \
            }                                                                                                                      \
            if (k >= 4 && k < 8 && !(WINO_PROBE & 4)) {                                                                            \
                _Pragma("unroll") for (int rr = 0; rr < RPW; ++rr) {                                                               \
                    tr_[rr][k - 4][0] = __builtin_fmaf(tsg[rr], db_[rr][k - 4][0], da_[rr][k - 4][0]);                             \
                    tr_[rr][k - 4][1] = __builtin_fmaf(tsg[rr], db_[rr][k - 4][1], da_[rr][k - 4][1]);                             \
                }                                                                                                                  \
            }                                                                                                                      \
            if (k >= 8 && k < 8 + RPW && !(WINO_PROBE & 4)) {                                                                      \
                const int rr = k - 8;                                                                                              \
                vd_[(txy[rr] * 4 + 0) * 64] = tr_[rr][0] - tr_[rr][2];                                                             \
                vd_[(txy[rr] * 4 + 1) * 64] = tr_[rr][1] + tr_[rr][2];                                                             \
                vd_[(txy[rr] * 4 + 2) * 64] = tr_[rr][2] - tr_[rr][1];                                                             \
                vd_[(txy[rr] * 4 + 3) * 64] = tr_[rr][1] - tr_[rr][3];                                                             \
            }                                                                                                                      \
            if (k >= 12 && k < 20 && !(WINO_PROBE & 4)) { VN[2 * (k - 12)] = vs_[(2 * (k - 12)) * 64]; VN[2 * (k - 12) + 1] = vs_[(2 * (k - 12) + 1) * 64]; } \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
    }

    bool first_block = true;
    for (;;) {
        const int blk_next = blk + nslots;
        const bool has_next = WINO_PERSIST && blk_next < blk_end;
        if (has_next) nxt = make_tile(blk_next);
        WINO_CHUNK(0, 0, true, v0, b0, v1, b1)
        WINO_CHUNK(1, 1, false, v1, b1, v0, b0)
        for (int c = 2; c < NC; c += 2) {
            WINO_CHUNK(c, 0, false, v0, b0, v1, b1)
            WINO_CHUNK(c + 1, 1, false, v1, b1, v0, b0)
        }
#if WINO_PROBE & 8
        {
            float sum = 0.f;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[xi][r];
            if (sum == 12345.678f) a.out[0] = sum;
        }
#else
        // ---- epilogue: Y = A^T M A per (tile, cout), + shift, ReLU -> this wave's LDS staging tile [pixel = tile * 4 + (y, x)][32
        // channels] -> 16-byte stores (8 lanes = the 128 contiguous bytes of one pixel's 32 channels).
        // acc[xi][r]: tile 8 (r / 4) + 4 h + (r % 4) of the wave, channel l & 31
        {
            // the DMAs / filter loads issued during the last chunk (the next block's) are drained HERE, before the stores below are
            // issued, so that the next block's first barrier need not wait on the vector-memory counter (= on these stores)
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
            int lo = l;                                    // opaque copy: keeps the per-lane store addresses from being hoisted out
            asm volatile("" : "+v"(lo));                   // of the block loop (they were spilled across the main loop: 70-100 VGPRs)
            const int ch0 = 32 * cur.nb;
            const float bv = a.bias ? a.bias[ch0 + i] : 0.f;
            const bool relu = a.relu != 0;
            float *stg = s_stg + (size_t)wv * 128 * WINO_STG_PITCH;
            // xi-major: the compiler moves an accumulator tile to VGPRs as a whole (16 registers) the moment one element feeds a
            // VALU op, so walking r-major (all 16 tiles live per row) pulled all 256 accumulators into VGPRs at once and spilled the
            // next block's operands.  Here one tile is live at a time and adds straight into the 4 x 16 output values:
            // y[2 i + j][r] = sum_xi At[i][xi >> 2] At[j][xi & 3] acc[xi][r], At = [1 1 1 0; 0 1 -1 -1] (36 adds per (tile, cout))
            float yv[4][16];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[p][r] = bv;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                float m[16];                               // element-wise AGPR reads (an "a"-constrained asm operand stays in its AGPR;
#pragma unroll                                             //  a plain acc[xi][r] use copies the whole tile, see above)
                for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[r]) : "a"(acc[xi][r]));
                const int cy[2] = {(xi >> 2) < 3 ? 1 : 0, (xi >> 2) == 0 ? 0 : ((xi >> 2) == 1 ? 1 : -1)};
                const int cx[2] = {(xi & 3) < 3 ? 1 : 0, (xi & 3) == 0 ? 0 : ((xi & 3) == 1 ? 1 : -1)};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int coef = cy[p >> 1] * cx[p & 1];
                    if (coef == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) yv[p][r] += m[r];
                    } else if (coef == -1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) yv[p][r] -= m[r];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int it = 8 * (r >> 2) + 4 * h + (r & 3);
#pragma unroll
                for (int p = 0; p < 4; ++p) stg[(it * 4 + p) * WINO_STG_PITCH + i] = relu ? fmaxf(yv[p][r], 0.f) : yv[p][r];
            }
            // (only this wave reads its staging tile back: its own LDS operations are ordered, no barrier)
            const int tile_y0 = TYW * (MW * cur.by + mw), tile_x0 = TXW * cur.bx;
            if (a.grp_cout) {
                // compact grouped output: this block's group keeps only its cg real channels, at channel offset oo of the map
                // (any alignment: 4-byte stores; the padded channels are never written)
                const int cg = a.grp_cout[cur.nb], oo = a.grp_ooff[cur.nb];
                float *ob = a.out + (size_t)cur.bidx * H * W * a.out_C + a.out_off + oo + 4 * (lo & 7);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int pix = k * 8 + (lo >> 3);
                    const int it = pix >> 2, py = (pix >> 1) & 1, px = pix & 1;
                    const int oy = 2 * (tile_y0 + (TALL ? (it >> 2) : (it >> 3))) + py, ox = 2 * (tile_x0 + (TALL ? (it & 3) : (it & 7))) + px;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + pix * WINO_STG_PITCH + 4 * (lo & 7));
                    if (oy < H && ox < W) {
                        float *o = ob + ((size_t)oy * W + ox) * a.out_C;
                        const int c = 4 * (lo & 7);
                        if (c < cg) o[0] = v.x;
                        if (c + 1 < cg) o[1] = v.y;
                        if (c + 2 < cg) o[2] = v.z;
                        if (c + 3 < cg) o[3] = v.w;
                    }
                    if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float *obase = a.out + (size_t)cur.bidx * H * W * a.out_C + a.out_off + ch0 + 4 * (lo & 7);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int pix = k * 8 + (lo >> 3);    // staged pixel: tile pix >> 2, position pix & 3
                    const int it = pix >> 2, py = (pix >> 1) & 1, px = pix & 1;
                    const int oy = 2 * (tile_y0 + (TALL ? (it >> 2) : (it >> 3))) + py, ox = 2 * (tile_x0 + (TALL ? (it & 3) : (it & 7))) + px;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + pix * WINO_STG_PITCH + 4 * (lo & 7));
                    if (oy < H && ox < W) *reinterpret_cast<float4 *>(obase + ((size_t)oy * W + ox) * a.out_C) = v;
                    if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four stores in flight at a time: all 16 at once spill
                }
            }
        }
#endif
        if (!has_next) break;
        cur = nxt;
        blk = blk_next;
        first_block = false;
    }
#undef WINO_CHUNK
}

// ------------------------------------------------------------------ v4: two waves per SIMD, the 16 positions split between a wave pair
// The kernel above runs ONE wave per SIMD (256 accumulator registers), so every instruction that is not an MFMA — a filter load
// (30-60 issue cycles each), a wait, the barrier — idles the matrix pipe (ablations: 0.53-0.58 of the fp32 MFMA peak against 0.62-0.68
// with all of them removed).  Here a (tile block, channel group) is shared by TWO waves that sit on the same SIMD: wave ph = 0 owns
// transform rows xi_y in {0, 1} (positions 0..7), wave ph = 1 rows {2, 3} — 128 accumulator registers each, 256 registers per wave,
// two waves per SIMD — and each wave's stalls are covered by its partner's MFMAs.  Workgroup = 8 waves: wave w -> (mw, nw) = w & 3
// split as above, ph = w >> 2 (waves w and w + 4 share SIMD w & 3).  Everything else is the v3 pipeline: same LDS images, same
// packed filters (a wave reads the four 16-byte pieces of its half), one barrier per chunk, persistent over tile blocks.
// The output transform is linear in the positions: each wave forms the partial 2 x 2 outputs of its half, the pair meets in an LDS
// staging tile (two rounds of 16 tiles), and each wave finishes (adds the partner's partials, shift, ReLU) and stores half the pixels.
static constexpr size_t wino2_lds_bytes(int NW, bool TALL) {
    return (size_t)2 * wino_rp(NW, TALL) * 16 + (size_t)2 * (4 / NW) * 16 * 64 * 8 + (size_t)4 * 2 * 64 * 32 * 4;
}

template <int NW, bool TALL>
__global__ __launch_bounds__(512) void wino_f23x2_kernel(const WinoArgs a) {
    constexpr int MW = 4 / NW;
    constexpr int TYW = TALL ? 8 : 4, TXW = TALL ? 4 : 8;
    constexpr int RW = 2 * TXW + 2;
    constexpr int RH = 2 * TYW * MW + 2;
    constexpr int RP = wino_rp(NW, TALL);
    constexpr int NQ = RP / 64;                           // DMA wave-instructions per chunk and workgroup
    constexpr int QW = (NQ + 7) / 8;                      // ... per wave (8 waves)
    extern __shared__ float4 s_mem4[];
    float4 *s_raw = s_mem4;                                                   // [2][RP]
    f32x2 *s_v = reinterpret_cast<f32x2 *>(s_mem4 + 2 * RP);                  // [2][MW][16][64]
    float *s_stg = reinterpret_cast<float *>(s_v + 2 * MW * 16 * 64);         // [4 pairs][2 halves][64 pixels][32 channels]

    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pr = wv & 3, ph = wv >> 2;                  // wave pair (= SIMD), position half
    const int mw = pr / NW, nw = pr % NW;
    const int i = l & 31, h = l >> 5;
    const int ty = TALL ? (i >> 2) : (i >> 3), tx = TALL ? (i & 3) : (i & 7);
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int NB = a.Cout >> 5;
    const size_t bstride = (size_t)NB * 8 * 64;          // float4 per chunk

    const int nb8 = (a.n_blocks + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int blk_end = min((xcd + 1) * nb8, a.n_blocks);
    int blk = xcd * nb8 + slot;
    if (blk >= blk_end) return;

    struct Tile {
        const float *src[QW];
        bool ok[QW];
        const f32x4 *b;                  // this wave's half of the packed filters (pieces 4 ph .. 4 ph + 3), chunk 0
        int bidx, by, bx, nb;
    };
    auto make_tile = [&](int blk_) {
        Tile tl;
        const int ng = blk_ % a.n_groups;
        blk_ /= a.n_groups;
        tl.bx = blk_ % a.blocks_x;
        blk_ /= a.blocks_x;
        tl.by = blk_ % a.blocks_y;
        tl.bidx = blk_ / a.blocks_y;
        tl.nb = ng * NW + nw;
        tl.b = reinterpret_cast<const f32x4 *>(a.upk) + ((size_t)tl.nb * 8 + 4 * ph) * 64 + l;
        const int R0 = 2 * TYW * MW * tl.by - 1, C0 = 2 * TXW * tl.bx - 1;
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int q = wv + 8 * k;
            const int p = q * 64 + l;
            const int ry = p / RW, rx = p - ry * RW;
            const int gy = R0 + ry, gx = C0 + rx;
            tl.ok[k] = (q < NQ) && (p < RH * RW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            tl.src[k] = tl.ok[k] ? a.in + (((size_t)tl.bidx * H + gy) * W + gx) * a.in_C + (size_t)tl.nb * a.in_goff : (const float *)&g_wino_zero;
        }
        return tl;
    };
    auto dma = [&](const Tile &tl, int c, int buf) {
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int q = wv + 8 * k;
            if (q < NQ) {
                const float *g = tl.ok[k] ? tl.src[k] + 4 * c : tl.src[k];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(s_raw + buf * RP + q * 64), 16, 0, 0);
            }
        }
    };
    // transform rows: the 2 NW waves of a tile block share its four rows — wave u = nw + NW * ph takes row u (NW = 4: the ph = 0 waves;
    // NW = 2: one row per wave)
    const int slot0 = (2 * TYW * mw + 2 * ty) * RW + 2 * tx;
    // (NW = 4: the ph = 1 waves recompute their partner's row and store the same values to the same place — cheaper than a branch
    // in every step of the loop, which also split the scheduling regions and spilled)
    const int xy = (nw + NW * ph) & 3;                    // wave-uniform
    const int tra = xy == 0 ? 0 : (xy == 2 ? 2 : 1), trb = xy < 2 ? 2 : (xy == 2 ? 1 : 3);
    const float tsg = xy == 1 ? 1.f : -1.f;
    auto transform = [&](int rbuf, int vbuf) {
        const f32x2 *base = reinterpret_cast<const f32x2 *>(s_raw + rbuf * RP + slot0) + h;
        f32x2 *vdst = s_v + ((size_t)(vbuf * MW + mw) * 16) * 64 + l;
        f32x2 tr[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x2 da = base[(tra * RW + x) * 2], db = base[(trb * RW + x) * 2];
            tr[x][0] = __builtin_fmaf(tsg, db[0], da[0]);
            tr[x][1] = __builtin_fmaf(tsg, db[1], da[1]);
        }
        vdst[(xy * 4 + 0) * 64] = tr[0] - tr[2];
        vdst[(xy * 4 + 1) * 64] = tr[1] + tr[2];
        vdst[(xy * 4 + 2) * 64] = tr[2] - tr[1];
        vdst[(xy * 4 + 3) * 64] = tr[1] - tr[3];
    };
    auto load_v = [&](int vbuf, f32x2 (&v)[8]) {          // this wave's positions 8 ph .. 8 ph + 7
        const f32x2 *vsrc = s_v + ((size_t)(vbuf * MW + mw) * 16 + 8 * ph) * 64 + l;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vsrc[e * 64];
    };
    auto load_b = [&](const Tile &tl, int c, f32x4 (&bb)[4]) {
        const f32x4 *p = tl.b + (size_t)c * bstride;
#pragma unroll
        for (int e = 0; e < 4; ++e) bb[e] = p[e * 64];
    };

    f32x16 acc[8];
    const int NC = Cin >> 2;
    f32x4 b0[4], b1[4];
    f32x2 vv[8];
    Tile cur = make_tile(blk), nxt = cur;

    dma(cur, 0, 0);
    dma(cur, 1, 1);
    load_b(cur, 0, b0);
    __syncthreads();
    transform(0, 0);
    __syncthreads();
    if (2 < NC) dma(cur, 2, 0);
    transform(1, 1);
    load_v(0, vv);
    __syncthreads();

    // chunk c (parity P): as in the one-wave kernel, 16 MFMAs per wave.  position e (0..7) of this wave, K-step s: B piece e >> 1,
    // component 2 (e & 1) + s
#define WINO2_CHUNK(c, P, FIRST, BC, BN)                                                                                           \
    {                                                                                                                              \
        if ((c) > 0) __syncthreads();                                                                                              \
        else if (!first_block) {                                                                                                   \
            __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                    \
            __builtin_amdgcn_s_barrier();                                                                                          \
        }                                                                                                                          \
        if ((c) + 3 < NC) dma(cur, (c) + 3, 1 - (P));                                                                              \
        else if (has_next) dma(nxt, (c) + 3 - NC, 1 - (P));                                                                        \
        const f32x4 *bp_ = ((c) + 1 < NC) ? cur.b + (size_t)((c) + 1) * bstride : (has_next ? nxt.b : cur.b);                      \
        const f32x2 *rb_ = reinterpret_cast<const f32x2 *>(s_raw + (P) * RP + slot0) + h;                                          \
        const f32x2 *vs_ = s_v + ((size_t)((1 - (P)) * MW + mw) * 16 + 8 * ph) * 64 + l;                                           \
        f32x2 *vd_ = s_v + ((size_t)((P) * MW + mw) * 16) * 64 + l;                                                                \
        f32x2 da_[4], db_[4];                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                                           \
            const float bop_ = BC[(k & 7) >> 1][((k & 1) << 1) + (k >> 3)];                                                        \
            if (FIRST && k < 8) {                                                                                                  \
                f32x16 z_;                                                                                                         \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) z_[r_] = 0.f;                                                    \
                acc[k & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[k & 7][k >> 3], bop_, z_, 0, 0, 0);                           \
            } else {                                                                                                               \
                acc[k & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[k & 7][k >> 3], bop_, acc[k & 7], 0, 0, 0);                   \
            }                                                                                                                      \
            if (k < 4) BN[k] = bp_[k * 64];                                                                                        \
            if (k < 4) { da_[k] = rb_[(tra * RW + k) * 2]; db_[k] = rb_[(trb * RW + k) * 2]; }                                     \
            if (k >= 4 && k < 8) {                                                                                                 \
                da_[k - 4][0] = __builtin_fmaf(tsg, db_[k - 4][0], da_[k - 4][0]);                                                 \
                da_[k - 4][1] = __builtin_fmaf(tsg, db_[k - 4][1], da_[k - 4][1]);                                                 \
            }                                                                                                                      \
            if (k == 8) {                                                                                                          \
                vd_[(xy * 4 + 0) * 64] = da_[0] - da_[2];                                                                          \
                vd_[(xy * 4 + 1) * 64] = da_[1] + da_[2];                                                                          \
                vd_[(xy * 4 + 2) * 64] = da_[2] - da_[1];                                                                          \
                vd_[(xy * 4 + 3) * 64] = da_[1] - da_[3];                                                                          \
            }                                                                                                                      \
            /* the next chunk's A operands roll into the registers this chunk has finished with: position e is last read by MFMA  \
               e + 8, so its successor is loaded at step e + 9 (one register set instead of two: 128 VGPRs is all a wave has) */   \
            if (k >= 9) vv[k - 9] = vs_[(k - 9) * 64];                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
        vv[7] = vs_[7 * 64];                                                                                                       \
    }

    bool first_block = true;
    for (;;) {
        const int blk_next = blk + nslots;
        const bool has_next = blk_next < blk_end;
        if (has_next) nxt = make_tile(blk_next);
        WINO2_CHUNK(0, 0, true, b0, b1)
        WINO2_CHUNK(1, 1, false, b1, b0)
        for (int c = 2; c < NC; c += 2) {
            WINO2_CHUNK(c, 0, false, b0, b1)
            WINO2_CHUNK(c + 1, 1, false, b1, b0)
        }
        // ---- epilogue.  Partial outputs of this wave's half: y[2 i + j][r] += At[i][xi_y] At[j][xi_x] acc[e][r], xi = 8 ph + e,
        // At = [1 1 1 0; 0 1 -1 -1]; two rounds of 8 accumulator rows (= 16 tiles = 64 pixels) through the pair's staging tile
        {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): drain the next block's DMA / filter loads before the stores
            int lo = l;
            asm volatile("" : "+v"(lo));
            const int ch0 = 32 * cur.nb;
            const float4 bv4 = a.bias ? *reinterpret_cast<const float4 *>(a.bias + ch0 + 4 * (lo & 7)) : make_float4(0.f, 0.f, 0.f, 0.f);
            const bool relu = a.relu != 0;
            float *stg = s_stg + (size_t)pr * 2 * 64 * 32;
            float *obase = a.out + (size_t)cur.bidx * H * W * a.out_C + a.out_off + ch0 + 4 * (lo & 7);
            const int tile_y0 = TYW * (MW * cur.by + mw), tile_x0 = TXW * cur.bx;
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                float yv[4][8];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int r = 0; r < 8; ++r) yv[p][r] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float m[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[r]) : "a"(acc[e][8 * rnd + r]));
                    // xi_y = 2 ph + (e >> 2), xi_x = e & 3.  ph is wave-uniform but not a compile-time constant: both cases written out
                    const int cx[2] = {(e & 3) < 3 ? 1 : 0, (e & 3) == 0 ? 0 : ((e & 3) == 1 ? 1 : -1)};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        // At[i][xi_y]: ph = 0: xi_y 0 -> (1, 0), 1 -> (1, 1);  ph = 1: xi_y 2 -> (1, -1), 3 -> (0, -1)
                        const int ii = p >> 1;
                        const int cy0 = (e >> 2) == 0 ? (ii == 0 ? 1 : 0) : 1;            // ph = 0
                        const int cy1 = (e >> 2) == 0 ? (ii == 0 ? 1 : -1) : (ii == 0 ? 0 : -1);   // ph = 1
                        const int c0 = cy0 * cx[p & 1], c1 = cy1 * cx[p & 1];
                        if (c0 == c1) {
                            if (c0 == 1) { _Pragma("unroll") for (int r = 0; r < 8; ++r) yv[p][r] += m[r]; }
                            else if (c0 == -1) { _Pragma("unroll") for (int r = 0; r < 8; ++r) yv[p][r] -= m[r]; }
                        } else {
                            const float f = ph == 0 ? (float)c0 : (float)c1;             // wave-uniform, in {-1, 0, 1}: exact
                            _Pragma("unroll") for (int r = 0; r < 8; ++r) yv[p][r] = __builtin_fmaf(f, m[r], yv[p][r]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // rows r = 8 rnd + j: tile it = 8 (r >> 2) + 4 h + (r & 3) -> local tile (it - 16 rnd) in 0..15
                float *mine = stg + (size_t)ph * 64 * 32;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int itl = 8 * (j >> 2) + 4 * h + (j & 3);
#pragma unroll
                    for (int p = 0; p < 4; ++p) mine[(itl * 4 + p) * 32 + i] = yv[p][j];
                }
                __syncthreads();                           // both halves of every pair are staged
                // this wave finishes pixels [32 ph, 32 ph + 32) of the round: 8 lanes per pixel, 4 store instructions
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pix = 32 * ph + k * 8 + (lo >> 3);
                    const float4 pa = *reinterpret_cast<const float4 *>(stg + pix * 32 + 4 * (lo & 7));
                    const float4 pb = *reinterpret_cast<const float4 *>(stg + 64 * 32 + pix * 32 + 4 * (lo & 7));
                    float4 v = make_float4(pa.x + pb.x + bv4.x, pa.y + pb.y + bv4.y, pa.z + pb.z + bv4.z, pa.w + pb.w + bv4.w);
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    const int it = 16 * rnd + (pix >> 2), py = (pix >> 1) & 1, px = pix & 1;
                    const int oy = 2 * (tile_y0 + (TALL ? (it >> 2) : (it >> 3))) + py, ox = 2 * (tile_x0 + (TALL ? (it & 3) : (it & 7))) + px;
                    if (oy < H && ox < W) *reinterpret_cast<float4 *>(obase + ((size_t)oy * W + ox) * a.out_C) = v;
                }
                if (rnd == 0) __syncthreads();             // the staging tile is rewritten by round 1
            }
        }
        if (!has_next) break;
        cur = nxt;
        blk = blk_next;
        first_block = false;
    }
#undef WINO2_CHUNK
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_wino_packed_floats(int Cin, int Cout) {
    if (Cin < 16 || Cout <= 0 || (Cin & 7) || (Cout & 31)) return 0;       // two 4-channel chunks per loop iteration; the input DMA runs three
                                                                           // chunks ahead, across tile blocks: a block must hold at least four
    return (size_t)16 * Cin * Cout;
}

LIDAR_EXPORT int lidar_wino_supported(int Cin, int Cout) { return lidar_wino_packed_floats(Cin, Cout) != 0; }

// w: (Cout, Cin, 3, 3) contiguous fp32 (the folded convolution weight) -> packed: lidar_wino_packed_floats(Cin, Cout) floats
LIDAR_EXPORT int lidar_wino_pack_weights(const float *w, int Cin, int Cout, float *packed, void *stream) {
    if (!w || !packed || !lidar_wino_supported(Cin, Cout)) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(divup((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, packed);
    return lidar_check_launch("lidar_wino_pack_weights");
}

static int wino_cu_count() {
    static int cus[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (cus[dev] == 0) cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cus[dev];
}

// out[b][y][x][out_off + co] = act(sum_{ky,kx,ci} in[b][y+ky-1][x+kx-1][ci] * w[co][ci][ky][kx] + bias[co])   (zero padding)
// grouped (in_goff > 0): 32-output-channel block nb reads input channels [nb * in_goff, nb * in_goff + Cin) of `in` (pixel pitch in_C)
static int wino_launch(const float *in, int B, int H, int W, int Cin, int in_C, int in_goff, const float *packed, const float *bias, int relu,
                       int Cout, float *out, int out_C, int out_off, void *stream, const int *grp_cout = nullptr, const int *grp_ooff = nullptr) {
    const bool compact = grp_cout != nullptr;
    if (!in || !packed || !out || B <= 0 || H <= 0 || W <= 0 || !lidar_wino_supported(Cin, Cout) || out_off < 0 ||
        (!compact && out_off + Cout > out_C) || (compact && (!grp_ooff || in_goff <= 0)))
        return LIDAR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15) || (in_C & 3) || (in_goff & 3)) return LIDAR_ERR_ARG;
    if (!compact && ((reinterpret_cast<uintptr_t>(out) & 15) || (out_C & 3) || (out_off & 3))) return LIDAR_ERR_ARG;   // 16-byte output stores
    WinoArgs a;
    a.in = in; a.upk = packed; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.out_C = out_C; a.out_off = out_off; a.relu = relu;
    a.in_C = in_C; a.in_goff = in_goff; a.grp_cout = grp_cout; a.grp_ooff = grp_ooff;
    // grouped: every 32-channel block has its own input slice, so the waves of a workgroup (which share ONE region image) must
    // share the channel block: four tile blocks x one block of 32 output channels
    const int NW = in_goff > 0 ? 1 : (Cout % 128 == 0) ? 4 : (Cout % 64 == 0) ? 2 : 1, MW = 4 / NW;
    const int tiles_y = (H + 1) / 2, tiles_x = (W + 1) / 2;
    // wave tile 4 x 8 tiles (wide) or 8 x 4 (tall): whichever covers the map with fewer tile blocks (62 x 54: 28 instead of 32)
    const long long wide = (long long)divup(tiles_y, 4 * MW) * divup(tiles_x, 8), tall = (long long)divup(tiles_y, 8 * MW) * divup(tiles_x, 4);
    const bool use_tall = tall < wide;
    a.blocks_y = divup(tiles_y, (use_tall ? 8 : 4) * MW);
    a.blocks_x = divup(tiles_x, use_tall ? 4 : 8);
    a.n_groups = Cout / (32 * NW);
    const long long nblk = (long long)B * a.blocks_y * a.blocks_x * a.n_groups;
    if (nblk > 0x7ffffff0ll) return LIDAR_ERR_ARG;
    a.n_blocks = (int)nblk;
    // persistent: one workgroup per CU (512 registers per wave: nothing else fits), each walks its XCD's share of the blocks
    long long want = ((nblk + 7) / 8) * 8;
#if WINO_PERSIST
    const long long cap = ((long long)wino_cu_count() / 8) * 8;
    if (cap >= 8 && want > cap) want = cap;
#endif
    const dim3 grid((unsigned)want), blk(256);
    hipStream_t s = (hipStream_t)stream;
    int dev_id = 0;                                       // the LDS opt-in is a per-device function attribute: one flag per device
    (void)hipGetDevice(&dev_id);
    dev_id &= 63;
#define WINO_LAUNCH(NWV, TALLV) do {                                                                                               \
        static bool attr_set[64] = {};                                                                                            \
        if (!attr_set[dev_id]) {                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_f23_kernel<NWV, TALLV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)wino_lds_bytes(NWV, TALLV));                                                            \
            attr_set[dev_id] = true;                                                                                              \
        }                                                                                                                         \
        hipLaunchKernelGGL((wino_f23_kernel<NWV, TALLV>), grid, blk, wino_lds_bytes(NWV, TALLV), s, a);                           \
    } while (0)
#define WINO2_LAUNCH(NWV, TALLV) do {                                                                                              \
        static bool attr_set2[64] = {};                                                                                           \
        if (!attr_set2[dev_id]) {                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_f23x2_kernel<NWV, TALLV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)wino2_lds_bytes(NWV, TALLV));                                                           \
            attr_set2[dev_id] = true;                                                                                             \
        }                                                                                                                         \
        hipLaunchKernelGGL((wino_f23x2_kernel<NWV, TALLV>), grid, dim3(512), wino2_lds_bytes(NWV, TALLV), s, a);                  \
    } while (0)
    // A/B switch: LIDAR_WINO_X2=1 selects the two-waves-per-SIMD kernel.  Default OFF — measured on one box, same process order
    // (profiles/r04/wino_x2_vs_v3.log): 64 ch 0.389 vs 0.401 ms, 128 ch 0.343-0.349 vs 0.328, 256 ch 0.357-0.362 vs 0.335, SECOND
    // shapes 0.83 / 0.86 vs 0.82 / 0.84: doubling the occupancy does not recover the 20 % the one-wave kernel loses against its
    // MFMA-only ablation, i.e. that loss is not per-wave latency (which a partner wave would hide) but issue time the SIMD spends on
    // the chunk's vector-memory / LDS instructions whichever wave they belong to.
    static const bool two_waves_env = getenv("LIDAR_WINO_X2") && atoi(getenv("LIDAR_WINO_X2")) != 0;
    const bool two_waves = two_waves_env && in_goff == 0;
    if (NW == 4 && two_waves) { if (use_tall) WINO2_LAUNCH(4, true); else WINO2_LAUNCH(4, false); }
    else if (NW == 2 && two_waves) { if (use_tall) WINO2_LAUNCH(2, true); else WINO2_LAUNCH(2, false); }
    else if (NW == 4) { if (use_tall) WINO_LAUNCH(4, true); else WINO_LAUNCH(4, false); }
    else if (NW == 2) { if (use_tall) WINO_LAUNCH(2, true); else WINO_LAUNCH(2, false); }
    else { if (use_tall) WINO_LAUNCH(1, true); else WINO_LAUNCH(1, false); }
#undef WINO_LAUNCH
#undef WINO2_LAUNCH
    return lidar_check_launch("lidar_wino_conv3x3_nhwc");
}

LIDAR_EXPORT int lidar_wino_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, const float *packed, const float *bias, int relu,
                                         int Cout, float *out, int out_C, int out_off, void *stream) {
    return wino_launch(in, B, H, W, Cin, Cin, 0, packed, bias, relu, Cout, out, out_C, out_off, stream);
}

// Grouped form: n_groups independent 3x3 convolutions, group g: input channels [g * group_cin, (g + 1) * group_cin) of the (B, H, W, in_C)
// map -> output channels [32 g, 32 g + 32) (a group with fewer than 32 real output channels pads its filters with zeros): the 36
// second-layer branch convolutions of AnchorHeadMulti's separate heads (pcdet/models/dense_heads/anchor_head_multi.py:60-110) in ONE
// launch.  packed = lidar_wino_pack_weights of the (32 n_groups, group_cin, 3, 3) stacked filters.
LIDAR_EXPORT int lidar_wino_conv3x3_grouped_nhwc(const float *in, int B, int H, int W, int in_C, int group_cin, int n_groups, const float *packed,
                                                 const float *bias, int relu, float *out, int out_C, int out_off, void *stream) {
    if (n_groups <= 0 || group_cin <= 0 || (long long)group_cin * n_groups > in_C) return LIDAR_ERR_ARG;
    return wino_launch(in, B, H, W, group_cin, in_C, group_cin, packed, bias, relu, 32 * n_groups, out, out_C, out_off, stream);
}

// The grouped form with a COMPACT output: group g keeps only its grp_cout[g] (<= 32) real output channels and writes them at channels
// [out_off + grp_ooff[g], + grp_cout[g]) of the (B, H, W, out_C) map (device int arrays of n_groups entries; the caller guarantees the
// ranges fit and do not overlap) — the padded channels never reach memory (AnchorHeadMulti: 236 real channels of 36 x 32).
LIDAR_EXPORT int lidar_wino_conv3x3_grouped_compact_nhwc(const float *in, int B, int H, int W, int in_C, int group_cin, int n_groups,
                                                         const float *packed, const float *bias, int relu, const int *grp_cout,
                                                         const int *grp_ooff, float *out, int out_C, int out_off, void *stream) {
    if (n_groups <= 0 || group_cin <= 0 || (long long)group_cin * n_groups > in_C || !grp_cout || !grp_ooff) return LIDAR_ERR_ARG;
    return wino_launch(in, B, H, W, group_cin, in_C, group_cin, packed, bias, relu, 32 * n_groups, out, out_C, out_off, stream, grp_cout, grp_ooff);
}
