// 3x3 / stride 1 / pad 1 fp32 convolution on NHWC maps as Winograd F(4x4, 3x3) on the fp32 matrix cores — the second generation of
// csrc/wino_conv.hip (F(2x2, 3x3)) for the same layers: the stride-1 Conv2d / BatchNorm2d / ReLU stacks of BaseBEVBackbone
// (pcdet/models/backbones_2d/base_bev_backbone.py:34-45).  Folded BatchNorm shift + ReLU in the epilogue, output written at a channel
// offset of a wider NHWC map.
//
// Why: every dense kernel of the step already runs at the clock-limited ceiling of the fp32 MFMA (0.60-0.72 of 157 TFLOP/s,
// profiles/r04), so only fewer multiplies make it faster.  F(4x4, 3x3) needs 36 multiplies per 4 x 4 output tile and (cin, cout) pair
// where the direct form needs 144 and F(2x2, 3x3) 64: 1.78x fewer MFMA cycles than wino_conv.hip.  It pays with larger transform
// constants, i.e. rounding error.  Interpolation points {0, 1, -1, 1/2, -2, inf} (not the textbook {0, +-1, +-2, inf}: 3x smaller
// maximum error, Barabasz et al. 2020), rows of B^T scaled to small integers with the inverse factors folded into the filter
// transform (evaluated in fp64):
//   B^T = [2 -3 -4 3 2 0; 0 -2 1 5 2 0; 0 -2 5 -1 -2 0; 0 2 1 -2 -1 0; 0 1 -2 -1 2 0; 0 2 -3 -4 3 2]
//   G   = [1/2 0 0; 1/6 1/6 1/6; 1/6 -1/6 1/6; 16/15 8/15 4/15; 1/30 -1/15 2/15; 0 0 1/2]
//   A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]          Y = A^T [ (G g G^T) .* (B^T d B) ] A
// Measured |error| <= 2e-5 for outputs of magnitude 5 at 64-256 input channels (the direct fp32 convolution: 1e-6, F(2x2): 2e-6);
// asserted against the fp64 convolution at the north_star tolerance 1e-4 in tests/test_gpu_wino.py.
//
// Shape of the computation.  For each of the 36 transform positions p = 6 xy + x the sum over cin is a GEMM
// M_p[tile, cout] = V_p[tile, cin] @ U_p[cin, cout].  A WAVE owns 16 tiles (a TY x TX group: 4 x 4, 2 x 8 or 8 x 2 tiles = 256
// output pixels) x 32 output channels x all 36 positions as 72 accumulator tiles of v_mfma_f32_16x16x4_f32 (4 registers each): 288
// registers — the 256 AGPRs hold positions 0..31, positions 32..35 accumulate in 32 VGPRs (the MFMAs are inline assembly with the
// register class spelled out: given the choice, hipcc parks the overflow in VGPRs and copies it through AGPRs around every use).  In
// the accumulator layout all 36 positions of one (tile, cout) sit in the same lane at the same element index, so the output
// transform is lane-local.  One wave per SIMD, 512 registers.  Workgroup = 4 waves = 2 tile groups (stacked in y) x 2 blocks of 32
// output channels; Cout / 64 channel groups are separate logical blocks.
//   A operand (16 tiles x 4 channels of the chunk): lane (m = lane & 15, kq = lane >> 4) supplies tile m, channel 4c + kq.  The
//              input transform of a tile group is computed once per workgroup: its two waves take transform rows xy 0..2 and 3..5
//              (raw pixels from the LDS image of the region -> B^T d -> (B^T d) B) and leave V in LDS in A-operand lane order, two
//              positions per 8-byte word.
//   B operand: transformed filters packed once per weight update in the order the lanes consume them —
//              [chunk][cout / 32][position pair e][lane][(p & 1) * 2 + cout half]: one contiguous 1 KB wave load per pair.
//   LDS image of the region ((8 TY + 2) x (4 TX + 2) pixels x 4 channels): LDS-DMA (global_load_lds_dwordx4), pixels stored TILE-MAJOR
//              ([row in tile][column in tile][tile]) so that the 16 tiles x 4 channels a wave reads per patch position spread over the
//              banks (pixel-major: 4-way conflicts).  Zero padding = out-of-image lanes read a zero word.
// Pipeline per chunk c of 4 channels (72 MFMAs = 2 304 matrix-pipe cycles per wave), the v3 scheme of wino_conv.hip: MFMAs(c) |
// operands of c + 1 | transform of c + 2 | DMA of c + 3, one barrier per chunk, persistent workgroups that never drain the pipeline
// across tile blocks, issue order written out slot by slot.  Differences: the operand registers ROLL (a pair's registers are reloaded
// for the next chunk right after its four MFMAs issue: one register set instead of two; the last two pairs keep two sets so that no
// LDS read is young when the barrier waits for lgkmcnt(0)), and the barrier waits for the DMA only — s_waitcnt vmcnt(18) leaves the 18
// filter loads of the next chunk in flight (loads return in order, they were issued after the DMA).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WINO43_PROBE                     // timing probes (WRONG results): bit 0 no DMA in the loop, bit 1 no filter loads, bit 2 no
#define WINO43_PROBE 0                   // transform, bit 3 no epilogue, bit 4 no barrier
#endif

#define W43_OOB 0x80000000u               // a byte offset no map reaches (the launcher keeps maps below 2^31 bytes): the DMA reads zeros

// ------------------------------------------------------------------ filter transform + packing
// w: (Cout, Cin, 3, 3) contiguous.  float4 index ((c * NB + nb) * 18 + e) * 64 + lane, component (p & 1) * 2 + hf:
// U_p[cin = 4 c + (lane >> 4)][cout = 32 nb + 2 (lane & 15) + hf], p = 2 e + (component >> 1)   (a lane's two output channels are
// neighbours: the epilogue stages them as one 8-byte word)
__global__ __launch_bounds__(256) void wino43_pack_kernel(const float *__restrict__ w, int Cin, int Cout, float *__restrict__ upk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int cout = idx / Cin, cin = idx - cout * Cin;
    const float *g = w + ((size_t)cout * Cin + cin) * 9;
    const double G[6][3] = {{0.5, 0., 0.}, {1. / 6, 1. / 6, 1. / 6}, {1. / 6, -1. / 6, 1. / 6},
                            {16. / 15, 8. / 15, 4. / 15}, {1. / 30, -2. / 30, 4. / 30}, {0., 0., 0.5}};
    double t[6][3];                      // G g
#pragma unroll
    for (int y = 0; y < 6; ++y)
#pragma unroll
        for (int x = 0; x < 3; ++x) t[y][x] = G[y][0] * (double)g[x] + G[y][1] * (double)g[3 + x] + G[y][2] * (double)g[6 + x];
    const int c = cin >> 2, kq = cin & 3, nb = cout >> 5, hf = cout & 1, n = (cout >> 1) & 15, NB = Cout >> 5;
    const int lane = kq * 16 + n;
#pragma unroll
    for (int y = 0; y < 6; ++y)
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            const double u = t[y][0] * G[x][0] + t[y][1] * G[x][1] + t[y][2] * G[x][2];
            const int p = y * 6 + x, e = p >> 1;
            upk[(((((size_t)c * NB + nb) * 18 + e) * 64 + lane) << 2) + ((p & 1) << 1) + hf] = (float)u;
        }
}

// ------------------------------------------------------------------ the convolution
struct Wino43Args {
    const float *in;         // (B, H, W, in_C), the layer reads channels [0, Cin)
    const float *upk;        // packed transformed filters
    const float *bias;       // (Cout) or null
    float *out;              // (B, H, W, out_C), this layer's channels at [out_off, out_off + Cout)
    int B, H, W, Cin, Cout, in_C, out_C, out_off, relu;
    int blocks_y, blocks_x, n_groups, n_blocks;       // n_blocks = B * blocks_y * blocks_x * n_groups (n_groups = Cout / 64)
};

#define W43_STG_PITCH 36                  // floats per staged output pixel (32 channels + 4: 16-byte aligned rows, spread over banks)

static constexpr int w43_nt(int TY, int TX) { return (2 * TY + 1) * (TX + 1); }              // region tiles (incl. the half tiles)
static constexpr int w43_rp(int TY, int TX) { return ((16 * w43_nt(TY, TX) + 255) / 256) * 256; }  // region slots: every wave issues the same number of DMAs
static constexpr size_t w43_lds_bytes(int TY, int TX) {
    return (size_t)2 * w43_rp(TY, TX) * 16 + (size_t)2 * 2 * 18 * 64 * 8 + (size_t)4 * 64 * W43_STG_PITCH * 4;
}

#define W43_MFMA_A(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_V(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_A0(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_V0(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(ACC) : "v"(VA), "v"(VB))

template <int TY, int TX>
__global__ __launch_bounds__(256) void wino_f43_kernel(const Wino43Args a) {
    static_assert(TY * TX == 16, "a wave owns 16 tiles");
    constexpr int TR = 2 * TY + 1, TC = TX + 1, NT = TR * TC;   // region tile rows / columns (the last of each: 2 pixels of halo)
    constexpr int RH = 8 * TY + 2, RW = 4 * TX + 2;             // region pixels
    constexpr int RP = w43_rp(TY, TX);
    constexpr int NQ = RP / 64;                                 // DMA wave-instructions per chunk and workgroup
    constexpr int QW = NQ / 4;                                  // ... per wave (RP is a multiple of 256)
    extern __shared__ float4 s_mem4[];
    float4 *s_raw = s_mem4;                                                   // [2][RP]: [slot][4 channels of the chunk]
    f32x2 *s_v = reinterpret_cast<f32x2 *>(s_mem4 + 2 * RP);                  // [2][2 tile groups][18 pairs][64 lanes]
    float *s_stg = reinterpret_cast<float *>(s_v + 2 * 2 * 18 * 64);          // [4 waves][64 pixels][W43_STG_PITCH]

    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mw = wv >> 1, nw = wv & 1;
    const int m = l & 15, kq = l >> 4;
    const int ty = m / TX, tx = m % TX;
    const int H = a.H, W = a.W;
    const int NB = a.Cout >> 5;
    const unsigned bstride = (unsigned)(NB * 18 * 64);   // float4 per chunk
    const f32x4 *upk4 = reinterpret_cast<const f32x4 *>(a.upk);

    // ---- logical blocks of this workgroup (XCD-aware, as wino_conv.hip: XCD x owns blocks [x nb8, (x + 1) nb8))
    const int nb8 = (a.n_blocks + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int blk_end = min((xcd + 1) * nb8, a.n_blocks);
    int blk = xcd * nb8 + slot;
    if (blk >= blk_end) return;

    // ---- this lane's DMA slots: region pixel of slot s = q * 64 + l  (slot = (row-in-tile * 4 + column-in-tile) * NT + tile), decoded in
    // make_tile once per block (kept in registers across the main loop it cost spills, and a scratch reload is a vmcnt(0))
    struct Tile {
        unsigned off[QW];                // byte offset of this lane's DMA source pixels, chunk 0 (W43_OOB: outside the image -> zeros)
        unsigned b;                      // float4 index of this lane's packed filters, chunk 0 (a.upk + 32-bit offset: scalar base addressing)
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
        tl.nb = ng * 2 + nw;
        tl.b = (unsigned)(tl.nb * 18 * 64 + l);
        const int R0 = 8 * TY * tl.by - 1, C0 = 4 * TX * tl.bx - 1;          // image coordinates of region pixel (0, 0)
        int lq = l;
        asm volatile("" : "+v"(lq));                      // opaque: the slot decode below must not be hoisted out of the block loop
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int sl = (wv + 4 * k) * 64 + lq;
            const int c0 = sl / NT, tidx = sl - c0 * NT;
            const int tyy = tidx / TC, txx = tidx - tyy * TC;
            const int ry = 4 * tyy + (c0 >> 2), rx = 4 * txx + (c0 & 3);
            const int gy = R0 + ry, gx = C0 + rx;
            const bool ok = sl < 16 * NT && ry < RH && rx < RW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            tl.off[k] = ok ? (unsigned)(((tl.bidx * H + gy) * W + gx) * a.in_C) * 4u : W43_OOB;   // (< 2^31 bytes: checked by the launcher)
        }
        return tl;
    };
    // LDS-DMA as inline assembly, through a buffer descriptor of the input map: (1) with the builtin in flight hipcc treats the
    // vector-memory counter as unordered and turns every wait into vmcnt(0) / lgkmcnt(0) — one memory round trip per chunk; hidden from
    // its bookkeeping, its waits for the rolling filter loads stay exact (17 younger loads allowed; in fact QW more are in flight, so
    // the wait is slightly stricter than needed, never looser) and the barriers below wait for the DMA explicitly; (2) the buffer form
    // takes a 32-bit byte offset per lane (one add per chunk) and returns ZEROS for an offset beyond the map: that is the zero padding.
    // ALWAYS issued, QW per wave and chunk (the count is what "vmcnt(18)" relies on).  The DMA runs as a STREAM three chunks ahead of
    // the MFMAs: after the last chunk of a block it continues with chunk 0 of the workgroup's next block (W43_OOB when there is none).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rsrc;
    {
        const unsigned long long base = (unsigned long long)a.in;
        rsrc[0] = (int)(unsigned)base;
        rsrc[1] = (int)((unsigned)(base >> 32) & 0xffffu);                       // stride 0: raw buffer
        rsrc[2] = (int)((unsigned)a.B * (unsigned)H * (unsigned)W * (unsigned)a.in_C * 4u);   // bytes; offsets beyond read as zero
        rsrc[3] = 0x00020000;
    }
    const unsigned raw_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char *)s_raw);
    unsigned doff[QW];                    // the stream's next chunk, this lane's pixels
    auto dma = [&](int buf) {
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const unsigned dst = raw_lds + (unsigned)((buf * RP + (wv + 4 * k) * 64) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(doff[k]), "s"(rsrc), "s"(dst) : "memory", "m0");
        }
    };
    // ---- input transform, this wave's share: rows xy = 3 nw + r (r = 0..2) of V = B^T d B for tile group mw, lane = (tile m, channel
    // kq) as in the A operand.  Row xy of B^T d is a combination of the five raw rows nw .. nw + 4 (wave-uniform coefficients, the
    // same instructions for both waves); the second stage is the full B^T per row.
    // float index of region pixel (dy, dx) of this lane's patch: ((dy & 3) * 4 + (dx & 3)) * NT * 4 + ((dy >> 2) * TC + (dx >> 2)) * 4 + base
    const int rbase = ((mw * TY + ty) * TC + tx) * 4 + kq;
    int rrow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int dy = nw + j;
        rrow[j] = rbase + ((dy & 3) * 4 * NT + (dy >> 2) * TC) * 4;
    }
    float cf[3][5];
    {
        const float c0[3][5] = {{2.f, -3.f, -4.f, 3.f, 2.f}, {0.f, -2.f, 1.f, 5.f, 2.f}, {0.f, -2.f, 5.f, -1.f, -2.f}};    // rows 0..2 over raw rows 0..4
        const float c1[3][5] = {{2.f, 1.f, -2.f, -1.f, 0.f}, {1.f, -2.f, -1.f, 2.f, 0.f}, {2.f, -3.f, -4.f, 3.f, 2.f}};    // rows 3..5 over raw rows 1..5
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 5; ++j) cf[r][j] = nw ? c1[r][j] : c0[r][j];
    }
#define W43_COL(x) ((((x) & 3) * NT + ((x) >> 2)) * 4)                    /* float offset of patch column x */
    // second stage: o = B^T w for one row (6 -> 6)
#define W43_S2_0(w_, o_) o_[0] = __builtin_fmaf(2.f, w_[0] + w_[4], __builtin_fmaf(-3.f, w_[1] - w_[3], -4.f * w_[2]));
#define W43_S2_1(w_, o_) o_[1] = __builtin_fmaf(-2.f, w_[1], w_[2]) + __builtin_fmaf(5.f, w_[3], 2.f * w_[4]);
#define W43_S2_2(w_, o_) o_[2] = __builtin_fmaf(-2.f, w_[1] + w_[4], __builtin_fmaf(5.f, w_[2], -w_[3]));
#define W43_S2_34(w_, o_)                                                                                                          \
    {                                                                                                                              \
        const float a_ = w_[1] - w_[3], b_ = w_[2] - w_[4];                                                                        \
        o_[3] = __builtin_fmaf(2.f, a_, b_);                                                                                       \
        o_[4] = __builtin_fmaf(-2.f, b_, a_);                                                                                      \
    }
#define W43_S2_5(w_, o_) o_[5] = __builtin_fmaf(2.f, w_[1] + w_[5], __builtin_fmaf(-3.f, w_[2] - w_[4], -4.f * w_[3]));
#define W43_STAGE2(w_, o_) { W43_S2_0(w_, o_) W43_S2_1(w_, o_) W43_S2_2(w_, o_) W43_S2_34(w_, o_) W43_S2_5(w_, o_) }
    auto transform = [&](int rbuf, int vbuf) {            // (pipeline fill only: the loop below carries its own interleaved copy)
        const float *raw = reinterpret_cast<const float *>(s_raw + rbuf * RP);
        f32x2 *vd = s_v + ((size_t)(vbuf * 2 + mw) * 18 + 9 * nw) * 64 + l;
        float w_[3][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float d[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) d[j] = raw[rrow[j] + W43_COL(x)];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float s = cf[r][0] * d[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) s = __builtin_fmaf(cf[r][j], d[j], s);
                w_[r][x] = s;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float o[6];
            W43_STAGE2(w_[r], o)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) vd[(r * 3 + jj) * 64] = (f32x2){o[2 * jj], o[2 * jj + 1]};
        }
    };

    f32x4 acc[72];                        // acc[2 p + hf]: position p, cout half hf; element i: tile 4 kq + i of the group, cout (l & 15)
    f32x4 bb[18];                         // filters of the current chunk (rolling)
    f32x2 va[16], vx[2][2];               // A operands: pairs 0..15 rolling, pairs 16, 17 one set per chunk parity
    const int NC = a.Cin >> 2;
    Tile cur = make_tile(blk), nxt = cur;
#pragma unroll
    for (int k = 0; k < QW; ++k) doff[k] = cur.off[k];
    auto dma_advance = [&](bool cross, bool to_next, const Tile &tn) {       // cross: the chunk just fetched was the block's last one
#pragma unroll
        for (int k = 0; k < QW; ++k) doff[k] = cross ? (to_next ? tn.off[k] : W43_OOB) : doff[k] + 16u;
    };

    // ---- pipeline fill (first block only): raw(0..2), V(0), V(1), the operands of chunk 0     (NC >= 4: no block end in here)
    dma(0);
    dma_advance(false, false, cur);
    dma(1);
    dma_advance(false, false, cur);
#pragma unroll
    for (int e = 0; e < 18; ++e) bb[e] = upk4[cur.b + e * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMAs are invisible to the compiler's own wait)
    __syncthreads();                                      // raw(0), raw(1) landed
    transform(0, 0);
    __syncthreads();                                      // V(0) visible; raw[0] free
    dma(0);
    dma_advance(false, false, cur);
    transform(1, 1);
    {
        const f32x2 *vs = s_v + ((size_t)(0 * 2 + mw) * 18) * 64 + l;
#pragma unroll
        for (int e = 0; e < 16; ++e) va[e] = vs[e * 64];
        vx[0][0] = vs[16 * 64];
        vx[0][1] = vs[17 * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                      // V(1) visible, raw(2) landed; raw[1] free

    // chunk c (parity P = c & 1, NC even): DMA raw(c + 3) -> raw[!P]; operands of c + 1 (A from V[!P], filters from global) into the
    // registers the MFMAs have just read; transform raw(c + 2) in raw[P] -> V[P].  Past the end of the block, "c + k" means chunk
    // c + k - NC of the NEXT block (same parities).  FIRST: the block's first chunk starts the accumulators from a zero C operand.
    // Slot k = 0..71 <-> MFMA (pair e = k >> 2, position p = 2 e + ((k >> 1) & 1), half k & 1).
#define W43_CHUNK(c, P, FIRST)                                                                                                     \
    {                                                                                                                              \
        if (!(WINO43_PROBE & 16)) {                                                                                                \
            if ((c) > 0) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                 \
            else if (!first_block) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                 \
        }                                                                                                                          \
        if (!(WINO43_PROBE & 1)) {                                                                                                 \
            dma(1 - (P));                              /* stream element c + 3 */                                                  \
            dma_advance((c) + 4 == NC, has_next, nxt);                                                                             \
        }                                                                                                                          \
        const unsigned bp_ = ((c) + 1 < NC) ? cur.b + (unsigned)((c) + 1) * bstride : (has_next ? nxt.b : cur.b);                  \
        const float *raw_ = reinterpret_cast<const float *>(s_raw + (P) * RP);                                                     \
        const f32x2 *vs_ = s_v + ((size_t)((1 - (P)) * 2 + mw) * 18) * 64 + l;                                                     \
        f32x2 *vd_ = s_v + ((size_t)((P) * 2 + mw) * 18 + 9 * nw) * 64 + l;                                                        \
        float d_[6][5], w_[3][6], o_[3][6];                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 72; ++k) {                                                                           \
            const int e = k >> 2, p = 2 * e + ((k >> 1) & 1), hf = k & 1;                                                          \
            const float av_ = e < 16 ? va[e < 16 ? e : 0][p & 1] : vx[P][e - 16 < 0 ? 0 : e - 16][p & 1];                          \
            const float bv_ = bb[e][((p & 1) << 1) + hf];                                                                          \
            if (FIRST) {                                                                                                           \
                if (p < 32) W43_MFMA_A0(acc[2 * p + hf], av_, bv_); else W43_MFMA_V0(acc[2 * p + hf], av_, bv_);                   \
            } else {                                                                                                               \
                if (p < 32) W43_MFMA_A(acc[2 * p + hf], av_, bv_); else W43_MFMA_V(acc[2 * p + hf], av_, bv_);                     \
            }                                                                                                                      \
            /* operands of the next chunk into the registers this pair has just released */                                       \
            if ((k & 3) == 3) {                                                                                                    \
                if (!(WINO43_PROBE & 2)) bb[e] = upk4[bp_ + e * 64];                                                               \
                if (e < 16) va[e < 16 ? e : 0] = vs_[e * 64];                                                                      \
            }                                                                                                                      \
            if (k == 41) vx[1 - (P)][0] = vs_[16 * 64];                                                                            \
            if (k == 45) vx[1 - (P)][1] = vs_[17 * 64];                                                                            \
            /* input transform of chunk c + 2, spread over the chunk (one wave per SIMD: whatever does not fit into the 28 free    \
               issue cycles behind an MFMA idles the matrix pipe): one patch read per slot (column-major) in slots 0..29, first      \
               stage item (x, r) at slot 8 + 2 (3 x + r) (column x is complete at slot 5 x + 4), second stage in slots 44..58 */     \
            if (!(WINO43_PROBE & 4)) {                                                                                             \
                if (k < 30) d_[k / 5][k % 5] = raw_[rrow[k % 5] + W43_COL(k / 5)];                                                 \
                if (k >= 8 && k < 44 && !(k & 1)) {                                                                                \
                    const int x = ((k - 8) >> 1) / 3, r = ((k - 8) >> 1) % 3;                                                      \
                    float s = cf[r][0] * d_[x][0];                                                                                 \
                    _Pragma("unroll") for (int j = 1; j < 5; ++j) s = __builtin_fmaf(cf[r][j], d_[x][j], s);                       \
                    w_[r][x] = s;                                                                                                  \
                }                                                                                                                  \
                if (k >= 44 && k < 59) {                                                                                           \
                    const int r = (k - 44) / 5, part = (k - 44) % 5;                                                               \
                    if (part == 0) { W43_S2_0(w_[r], o_[r]) }                                                                      \
                    if (part == 1) { W43_S2_1(w_[r], o_[r]) }                                                                      \
                    if (part == 2) { W43_S2_2(w_[r], o_[r]) vd_[(r * 3 + 0) * 64] = (f32x2){o_[r][0], o_[r][1]}; }                 \
                    if (part == 3) { W43_S2_34(w_[r], o_[r]) vd_[(r * 3 + 1) * 64] = (f32x2){o_[r][2], o_[r][3]}; }                \
                    if (part == 4) { W43_S2_5(w_[r], o_[r]) vd_[(r * 3 + 2) * 64] = (f32x2){o_[r][4], o_[r][5]}; }                 \
                }                                                                                                                  \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
    }

    bool first_block = true;
    for (;;) {
        const int blk_next = blk + nslots;
        const bool has_next = blk_next < blk_end;
        if (has_next) nxt = make_tile(blk_next);
        W43_CHUNK(0, 0, true)
        W43_CHUNK(1, 1, false)
        for (int c = 2; c < NC; c += 2) {
            W43_CHUNK(c, 0, false)
            W43_CHUNK(c + 1, 1, false)
        }
#if !(WINO43_PROBE & 8)
        // ---- epilogue: Y = A^T M A per (tile, cout), + shift, ReLU.  Four rounds (element i of the accumulator tiles = tile 4 kq + i):
        // both output channels of the lane -> this wave's LDS staging tile [pixel-in-tile * 4 + kq][32 channels] -> 16-byte stores (8 lanes = the
        // 128 contiguous bytes of one pixel's 32 channels)
        {
            // the inline-assembly MFMAs are invisible to the compiler's hazard recogniser: let the last of them retire before the
            // first accumulator read; and drain the DMAs / filter loads of the next block before the stores below are issued, so
            // that the next block's first barrier need not wait on the vector-memory counter (= on these stores)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0) — the builtin, so that the compiler's own bookkeeping sees it too
            int lo = l;                                    // opaque copy: keeps the per-lane store addresses inside the block loop
            asm volatile("" : "+v"(lo));
            const int ch0 = 32 * cur.nb;
            const bool relu = a.relu != 0;
            float *stg = s_stg + (size_t)wv * 64 * W43_STG_PITCH;
            // this lane's two output channels are ch0 + 2 (l & 15) + {0, 1} (the packing interleaves the cout halves): every value
            // below is the PAIR (hf 0, hf 1) — packed fp32 arithmetic (v_pk_*), half the VALU instructions, and one 8-byte staging write
            const f32x2 bv = a.bias ? *reinterpret_cast<const f32x2 *>(a.bias + ch0 + 2 * (lo & 15)) : (f32x2){0.f, 0.f};
            const int gy0 = 4 * TY * (2 * cur.by + mw), gx0 = 4 * TX * cur.bx;       // first output pixel of this wave's tile group
            float *obase = a.out + (size_t)cur.bidx * H * W * a.out_C + a.out_off + ch0 + 4 * (lo & 7);
#define W43_PKFMA(a_, b_, c_) __builtin_elementwise_fma((f32x2){a_, a_}, b_, c_)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 y[4][4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[ii][j] = bv;
#pragma unroll
                for (int xy = 0; xy < 6; ++xy) {
                    f32x2 mm[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) {
                        const int p = 6 * xy + x;
                        if (p < 32) {
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(mm[x][0]) : "a"(acc[2 * p][i]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(mm[x][1]) : "a"(acc[2 * p + 1][i]));
                        } else {
                            mm[x] = (f32x2){acc[2 * p][i], acc[2 * p + 1][i]};
                        }
                    }
                    const f32x2 s = mm[1] + mm[2], d = mm[1] - mm[2];
                    f32x2 tt[4];
                    tt[0] = (mm[0] + s) + (mm[3] + mm[4]);
                    tt[1] = W43_PKFMA(0.5f, mm[3], W43_PKFMA(-2.f, mm[4], d));
                    tt[2] = W43_PKFMA(0.25f, mm[3], W43_PKFMA(4.f, mm[4], s));
                    tt[3] = W43_PKFMA(0.125f, mm[3], W43_PKFMA(-8.f, mm[4], d)) + mm[5];
                    const float at[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f},
                                            {1.f, 0.5f, 0.25f, 0.125f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float cfa = at[xy][ii];
                        if (cfa == 0.f) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (cfa == 1.f) y[ii][j] += tt[j];
                            else if (cfa == -1.f) y[ii][j] -= tt[j];
                            else y[ii][j] = W43_PKFMA(cfa, tt[j], y[ii][j]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 v = relu ? (f32x2){fmaxf(y[ii][j][0], 0.f), fmaxf(y[ii][j][1], 0.f)} : y[ii][j];
                        *reinterpret_cast<f32x2 *>(stg + ((ii * 4 + j) * 4 + (lo >> 4)) * W43_STG_PITCH + 2 * (lo & 15)) = v;
                    }
                // (only this wave reads its staging tile back: its own LDS operations are ordered, no barrier)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = k * 8 + (lo >> 3);      // staged pixel: pixel-in-tile q >> 2, lane group q & 3
                    const int mt = 4 * (q & 3) + i;       // tile of the group
                    const int oy = gy0 + 4 * (mt / TX) + (q >> 4), ox = gx0 + 4 * (mt % TX) + ((q >> 2) & 3);
                    const float4 v = *reinterpret_cast<const float4 *>(stg + q * W43_STG_PITCH + 4 * (lo & 7));
                    if (oy < H && ox < W) *reinterpret_cast<float4 *>(obase + ((size_t)oy * W + ox) * a.out_C) = v;
                    if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#else
        {
            float sum = 0.f;
#pragma unroll
            for (int p = 32; p < 36; ++p) sum += acc[2 * p][0] + acc[2 * p + 1][1];
            if (sum == 12345.678f) a.out[0] = sum;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        if (!has_next) break;
        cur = nxt;
        blk = blk_next;
        first_block = false;
    }
#undef W43_CHUNK
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_wino43_packed_floats(int Cin, int Cout) {
    if (Cin < 16 || Cout <= 0 || (Cin & 7) || (Cout & 63)) return 0;       // two 4-channel chunks per loop iteration, the input stream runs
                                                                           // three chunks ahead (>= 4 per block); 64 channels per workgroup
    return (size_t)36 * Cin * Cout;
}

LIDAR_EXPORT int lidar_wino43_supported(int Cin, int Cout) { return lidar_wino43_packed_floats(Cin, Cout) != 0; }

// w: (Cout, Cin, 3, 3) contiguous fp32 (the folded convolution weight) -> packed: lidar_wino43_packed_floats(Cin, Cout) floats
LIDAR_EXPORT int lidar_wino43_pack_weights(const float *w, int Cin, int Cout, float *packed, void *stream) {
    if (!w || !packed || !lidar_wino43_supported(Cin, Cout)) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(wino43_pack_kernel, dim3(divup((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, packed);
    return lidar_check_launch("lidar_wino43_pack_weights");
}

static int w43_cu_count() {
    static int cus[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (cus[dev] == 0) cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cus[dev];
}

// out[b][y][x][out_off + co] = act(sum_{ky,kx,ci} in[b][y+ky-1][x+kx-1][ci] * w[co][ci][ky][kx] + bias[co])   (zero padding)
// `in`: (B, H, W, in_C) NHWC, channels [0, Cin) are read (in_C >= Cin, both multiples of 4); `out`: (B, H, W, out_C).
LIDAR_EXPORT int lidar_wino43_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, int in_C, const float *packed, const float *bias, int relu,
                                           int Cout, float *out, int out_C, int out_off, void *stream) {
    if (!in || !packed || !out || B <= 0 || H <= 0 || W <= 0 || !lidar_wino43_supported(Cin, Cout) || out_off < 0 || out_off + Cout > out_C ||
        in_C < Cin)
        return LIDAR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15) || (in_C & 3)) return LIDAR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (out_C & 3) || (out_off & 3)) return LIDAR_ERR_ARG;     // 16-byte output stores
    if ((long long)B * H * W * in_C * 4 >= 0x7fffffffll) return LIDAR_ERR_ARG;                              // 32-bit byte offsets into the map
    Wino43Args a;
    a.in = in; a.upk = packed; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.in_C = in_C; a.out_C = out_C; a.out_off = out_off; a.relu = relu;
    // tile group shape: 4 x 4, 2 x 8 or 8 x 2 tiles (two groups stacked in y per workgroup) — whichever covers the map with the
    // fewest workgroup blocks (62 x 54: 2 x 7 blocks of 64 x 8 pixels instead of 4 x 4 of 32 x 16)
    const int tiles_y = (H + 3) / 4, tiles_x = (W + 3) / 4;
    const int shapes[3][2] = {{4, 4}, {2, 8}, {8, 2}};
    int best = 0;
    long long best_n = -1;
    for (int s = 0; s < 3; ++s) {
        const long long n = (long long)divup(tiles_y, 2 * shapes[s][0]) * divup(tiles_x, shapes[s][1]);
        if (best_n < 0 || n < best_n) { best_n = n; best = s; }
    }
    a.blocks_y = divup(tiles_y, 2 * shapes[best][0]);
    a.blocks_x = divup(tiles_x, shapes[best][1]);
    a.n_groups = Cout / 64;
    const long long nblk = (long long)B * a.blocks_y * a.blocks_x * a.n_groups;
    if (nblk > 0x7ffffff0ll) return LIDAR_ERR_ARG;
    a.n_blocks = (int)nblk;
    long long want = ((nblk + 7) / 8) * 8;                // persistent: one workgroup per CU, each walks its XCD's share of the blocks
    const long long cap = ((long long)w43_cu_count() / 8) * 8;
    if (cap >= 8 && want > cap) want = cap;
    const dim3 grid((unsigned)want), blk(256);
    hipStream_t s = (hipStream_t)stream;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    dev_id &= 63;
#define W43_LAUNCH(TYV, TXV) do {                                                                                                  \
        static bool attr_set[64] = {};                                                                                            \
        if (!attr_set[dev_id]) {                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_f43_kernel<TYV, TXV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)w43_lds_bytes(TYV, TXV));                                                               \
            attr_set[dev_id] = true;                                                                                              \
        }                                                                                                                         \
        hipLaunchKernelGGL((wino_f43_kernel<TYV, TXV>), grid, blk, w43_lds_bytes(TYV, TXV), s, a);                                \
    } while (0)
    if (best == 0) W43_LAUNCH(4, 4);
    else if (best == 1) W43_LAUNCH(2, 8);
    else W43_LAUNCH(8, 2);
#undef W43_LAUNCH
    return lidar_check_launch("lidar_wino43_conv3x3_nhwc");
}
