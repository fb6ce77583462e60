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
// output channels; Cout / 64 channel groups are separate logical blocks.  A chunk is 8 input channels = two K-steps = 144 MFMAs.
//   A operand: lane (m = lane & 15, kq = lane >> 4) supplies tile m, channels 8 C + 2 kq + s in K-step s.  The input transform of a
//              tile group is computed once per workgroup: its two waves take transform rows xy 0..2 and 3..5; a lane owns (tile,
//              channel PAIR): one 8-byte word of the region image -> B^T d B in packed fp32 (v_pk_*) -> V in LDS, the pair being
//              the A operands of the two K-steps.  Read back just in time through a 4-position register ring.
//   B operand: transformed filters packed once per weight update in the order the lanes consume them —
//              [chunk][cout / 32][position in the wave's order][lane][2 s + cout parity]: one contiguous 1 KB wave load per position,
//              a rolling window of 18 positions in registers, addresses on a scalar base.
//   LDS image of the region ((8 TY + 2) x (4 TX + 2) pixels x 8 channels): LDS-DMA (buffer_load_dwordx4 ... lds), pixels stored
//              TILE-MAJOR ([row in tile][column in tile][tile]) so that the 16 tiles a wave reads per patch position spread over the
//              banks.  Zero padding = byte offsets beyond the map read zeros (buffer bounds).
// Pipeline per chunk C: MFMAs(C) | filters of C / C + 1 | transform of C + 1 | DMA of C + 2, one barrier per chunk, persistent
// workgroups that never drain the pipeline across tile blocks.  The barrier waits for the DMA only — s_waitcnt vmcnt(36) leaves the
// 36 filter loads issued behind it in flight (loads return in order).
// Issue model (tools/ubench/mfma_issue.hip): on this part VALU instructions do NOT overlap the MFMAs of their SIMD (4.3 cycles each
// + 8.5 per MFMA -> VALU switch), LDS reads and vector-memory loads do.  Hence: arithmetic in few large bursts, addresses on the scalar
// unit, LDS reads one per MFMA slot, and as few VALU instructions as possible (channel pairs, packed fp32).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WINO43_PROBE                     // timing probes (WRONG results): bit 0 no DMA in the loop, bit 1 no filter loads, bit 2 no
#define WINO43_PROBE 0                   // transform, bit 3 no epilogue, bit 4 no barrier, bit 5 no transform arithmetic / V writes (reads stay), bit 7 no patch reads, bit 8 no global stores, bit 9 no load drain before the stores, bit 10 odd workgroups start late
#endif

#define W43_OOB 0x80000000u               // a byte offset no map reaches (the launcher keeps maps below 2^31 bytes): the DMA reads zeros

// ------------------------------------------------------------------ filter transform + packing
// w: (Cout, Cin, 3, 3) contiguous.  float4 index ((C * NB + nb) * 36 + q) * 64 + lane, component 2 s + hf:
// U_p[cin = 8 C + 2 (lane >> 4) + s][cout = 32 nb + 2 (lane & 15) + hf], position p = (q + 18 (nb & 1)) % 36 — block nb is consumed by
// the waves nw = nb & 1 of a workgroup, which walk the positions starting with the 18 whose V they computed themselves (see below).
// A lane's two input channels (K-steps s = 0, 1) and two output channels (hf) are neighbours in memory.
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
    const int C = cin >> 3, kq = (cin >> 1) & 3, sk = cin & 1, nb = cout >> 5, hf = cout & 1, n = (cout >> 1) & 15, NB = Cout >> 5;
    const int lane = kq * 16 + n;
#pragma unroll
    for (int y = 0; y < 6; ++y)
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            const double u = t[y][0] * G[x][0] + t[y][1] * G[x][1] + t[y][2] * G[x][2];
            const int p = y * 6 + x, q = (p + 36 - 18 * (nb & 1)) % 36;
            upk[(((((size_t)C * NB + nb) * 36 + q) * 64 + lane) << 2) + (sk << 1) + hf] = (float)u;
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
#define W43_OOB 0x80000000u               // a byte offset no map reaches (the launcher keeps maps below 2^31 bytes): the DMA reads zeros
#ifndef W43_B1                          // slots of the two transform bursts (A/B: tools/wino43_probe.py "0:W43_B1=24,W43_B2=48")
#define W43_B1 20
#endif
#ifndef W43_B2
#define W43_B2 40
#endif
#ifndef W43_STORE_AUX                   // cache policy bits of the output stores.  2 = nt (streaming): alone, a 64-channel layer gains 7 %
#define W43_STORE_AUX 0                 // (0.237 -> 0.221 ms: the 33 MB all workgroups write within microseconds stop evicting the filters
#endif                                  // from L2); inside the model the NEXT layer then misses them: 3 180 vs 3 203 frames/s, same box -> 0
#ifndef W43_R
#define W43_R 4                           // positions the A operands are read ahead of their MFMAs (divides 36)
#endif

static constexpr int w43_nt(int TY, int TX) { return (2 * TY + 1) * (TX + 1); }              // region tiles (incl. the half tiles)
static constexpr int w43_rp(int TY, int TX) { return ((32 * w43_nt(TY, TX) + 255) / 256) * 256; }  // DMA lanes (16 bytes) per chunk: every wave issues the same number
static constexpr size_t w43_lds_bytes(int TY, int TX) {
    return (size_t)2 * w43_rp(TY, TX) * 16 + (size_t)2 * 2 * 36 * 64 * 8 + (size_t)4 * 32 * W43_STG_PITCH * 4;
}

#define W43_MFMA_A(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_V(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_A0(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(ACC) : "v"(VA), "v"(VB))
#define W43_MFMA_V0(ACC, VA, VB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(ACC) : "v"(VA), "v"(VB))
#define W43_PK(c_) ((f32x2){c_, c_})
#define W43_FMA(a_, b_, c_) __builtin_elementwise_fma(a_, b_, c_)

template <int TY, int TX>
__global__ __launch_bounds__(256) void wino_f43_kernel(const Wino43Args a) {
    static_assert(TY * TX == 16, "a wave owns 16 tiles");
    constexpr int TR = 2 * TY + 1, TC = TX + 1, NT = TR * TC;   // region tile rows / columns (the last of each: 2 pixels of halo)
    constexpr int RH = 8 * TY + 2, RW = 4 * TX + 2;             // region pixels
    constexpr int RP = w43_rp(TY, TX);                          // 16-byte DMA lanes per chunk (two per pixel: 8 channels)
    constexpr int QW = RP / 256;                                // DMA wave-instructions per wave and chunk
    constexpr int VB = 2 * 36 * 64;                             // f32x2 per V buffer
    constexpr int R = W43_R;
    extern __shared__ float4 s_mem4[];
    float4 *s_raw = s_mem4;                                                   // [2][RP]: [pixel slot][8 channels of the chunk]
    f32x2 *s_v = reinterpret_cast<f32x2 *>(s_mem4 + 2 * RP);                  // [2][2 tile groups][36 positions][64 lanes]{K-step 0, 1}
    float *s_stg = reinterpret_cast<float *>(s_v + 2 * VB);                   // [4 waves][32 pixels][W43_STG_PITCH]

    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mw = wv >> 1, nw = wv & 1;
    const int m = l & 15, kq = l >> 4;
    const int ty = m / TX, tx = m % TX;
    const int H = a.H, W = a.W;
    const int NB = a.Cout >> 5;
    const int bstride = NB * 36 * 64;                    // float4 per chunk

    // ---- logical blocks of this workgroup (XCD-aware, as wino_conv.hip: XCD x owns blocks [x nb8, (x + 1) nb8))
    const int nb8 = (a.n_blocks + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int blk_end = min((xcd + 1) * nb8, a.n_blocks);
    int blk = xcd * nb8 + slot;
    if (blk >= blk_end) return;

    // ---- this lane's DMA lanes: lane sl = q * 64 + l of the chunk image holds the 16-byte half (sl & 1) of region pixel slot sl >> 1
    // (slot = (row-in-tile * 4 + column-in-tile) * NT + tile), decoded in make_tile once per block (kept in registers across the
    // main loop it cost spills, and a scratch reload is a vmcnt(0))
    struct Tile {
        unsigned off[QW];                // byte offset of this lane's DMA sources, chunk 0 (W43_OOB: outside the image -> zeros)
        int b;                           // float4 index of the wave's packed filters, chunk 0 (wave-uniform: scalar address arithmetic)
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
        tl.b = tl.nb * 36 * 64;
        const int R0 = 8 * TY * tl.by - 1, C0 = 4 * TX * tl.bx - 1;          // image coordinates of region pixel (0, 0)
        int lq = l;
        asm volatile("" : "+v"(lq));                      // opaque: the slot decode below must not be hoisted out of the block loop
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int sl = (wv + 4 * k) * 64 + lq;
            const int ps = sl >> 1;
            const int c0 = ps / NT, tidx = ps - c0 * NT;
            const int tyy = tidx / TC, txx = tidx - tyy * TC;
            const int ry = 4 * tyy + (c0 >> 2), rx = 4 * txx + (c0 & 3);
            const int gy = R0 + ry, gx = C0 + rx;
            const bool ok = ps < 16 * NT && ry < RH && rx < RW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            tl.off[k] = ok ? (unsigned)(((tl.bidx * H + gy) * W + gx) * a.in_C) * 4u + 16u * (unsigned)(sl & 1) : W43_OOB;
        }
        return tl;
    };
    // LDS-DMA as inline assembly, through a buffer descriptor of the input map: (1) with the builtin in flight hipcc treats the
    // vector-memory counter as unordered and turns every wait into vmcnt(0) / lgkmcnt(0) — one memory round trip per chunk; hidden from
    // its bookkeeping, its waits for the rolling filter loads stay exact (17 younger loads allowed; in fact QW more are in flight, so
    // the wait is slightly stricter than needed, never looser) and the barriers below wait for the DMA explicitly; (2) the buffer form
    // takes a 32-bit byte offset per lane (one add per chunk) and returns ZEROS for an offset beyond the map: that is the zero padding.
    // ALWAYS issued, QW per wave and chunk (the count is what "vmcnt(36)" relies on).  The DMA runs as a STREAM two chunks ahead of
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
    // the packed filters through a second descriptor: wave-uniform offset in a scalar register, lane * 16 in one vector register,
    // 32-bit addressing (half the address traffic of a 64-bit global load, and no VALU)
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.upk), (short)0, (int)(36u * (unsigned)a.Cin * (unsigned)a.Cout * 4u), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)((unsigned)a.B * (unsigned)H * (unsigned)W * (unsigned)a.out_C * 4u), 0x00020000);
    const int l16 = l * 16;
    auto ldf = [&](int f4_index) {                        // float4 number f4_index (wave-uniform) + lane
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frsrc, l16, f4_index * 16, 0));
    };
    const unsigned raw_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char *)s_raw);
    unsigned doff[QW];                    // the stream's next chunk, this lane's sources
    auto dma = [&](int buf) {
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const unsigned dst = raw_lds + (unsigned)((buf * RP + (wv + 4 * k) * 64) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(doff[k]), "s"(rsrc), "s"(dst) : "memory", "m0");
        }
    };

    // ---- input transform, this wave's share: rows xy = 3 nw + r (r = 0..2) of V = B^T d B for tile group mw.  Lane = (tile m, channel
    // PAIR kq): channels 8 C + 2 kq + {0, 1} are the lane's A operands of the chunk's two K-steps, sit in one 8-byte word of the region
    // image and go through the transform as one packed-fp32 value (v_pk_*: half the VALU instructions of a scalar transform).  Row xy
    // of B^T d is a combination of the five raw rows nw .. nw + 4 (wave-uniform coefficients, the same instructions for both waves);
    // the second stage is the full B^T per row.  The wave's 18 positions p = 18 nw + 6 r + x land at V[.][mw][p][lane].
    // float index of region pixel (dy, dx) of this lane's patch: (((dy & 3) * 4 + (dx & 3)) * NT + (dy >> 2) * TC + (dx >> 2)) * 8 + base
    const int rbase = ((mw * TY + ty) * TC + tx) * 8 + 2 * kq;
    int rrow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int dy = nw + j;
        rrow[j] = rbase + ((dy & 3) * 4 * NT + (dy >> 2) * TC) * 8;
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
#define W43_COL(x) ((((x) & 3) * NT + ((x) >> 2)) * 8)                    /* float offset of patch column x */
    // second stage: o = B^T w for one row (6 -> 6), packed
#define W43_S2_0(w_, o_) o_[0] = W43_FMA(W43_PK(2.f), w_[0] + w_[4], W43_FMA(W43_PK(-3.f), w_[1] - w_[3], W43_PK(-4.f) * w_[2]));
#define W43_S2_1(w_, o_) o_[1] = W43_FMA(W43_PK(-2.f), w_[1], w_[2]) + W43_FMA(W43_PK(5.f), w_[3], W43_PK(2.f) * w_[4]);
#define W43_S2_2(w_, o_) o_[2] = W43_FMA(W43_PK(-2.f), w_[1] + w_[4], W43_FMA(W43_PK(5.f), w_[2], -w_[3]));
#define W43_S2_34(w_, o_)                                                                                                          \
    {                                                                                                                              \
        const f32x2 a_ = w_[1] - w_[3], b_ = w_[2] - w_[4];                                                                        \
        o_[3] = W43_FMA(W43_PK(2.f), a_, b_);                                                                                      \
        o_[4] = W43_FMA(W43_PK(-2.f), b_, a_);                                                                                     \
    }
#define W43_S2_5(w_, o_) o_[5] = W43_FMA(W43_PK(2.f), w_[1] + w_[5], W43_FMA(W43_PK(-3.f), w_[2] - w_[4], W43_PK(-4.f) * w_[3]));
    f32x2 *const v_own = s_v + (size_t)(mw * 36 + 18 * nw) * 64 + l;           // + buffer * VB: the positions this wave computes (q = 0..17)
    const f32x2 *const v_prt = s_v + (size_t)(mw * 36 + 18 * (1 - nw)) * 64 + l;  // ... its partner's (q = 18..35)
    auto transform = [&](int rbuf, int vbuf) {            // (pipeline fill only: the loop below carries its own interleaved copy)
        const float *raw = reinterpret_cast<const float *>(s_raw + rbuf * RP);
        f32x2 *vd = v_own + vbuf * VB;
        f32x2 w_[3][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            f32x2 d[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) d[j] = *reinterpret_cast<const f32x2 *>(raw + rrow[j] + W43_COL(x));
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x2 sacc = W43_PK(cf[r][0]) * d[0];
#pragma unroll
                for (int j = 1; j < 5; ++j) sacc = W43_FMA(W43_PK(cf[r][j]), d[j], sacc);
                w_[r][x] = sacc;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x2 o[6];
            W43_S2_0(w_[r], o) W43_S2_1(w_[r], o) W43_S2_2(w_[r], o) W43_S2_34(w_[r], o) W43_S2_5(w_[r], o)
#pragma unroll
            for (int x = 0; x < 6; ++x) vd[(r * 6 + x) * 64] = o[x];
        }
    };

    f32x4 acc[72];                        // acc[2 q + hf]: the wave's q-th position, output channel 2 (l & 15) + hf; element i: tile 4 kq + i
    f32x4 bb[18];                         // filters, a rolling window of 18 positions: {K-step 0: hf 0, 1; K-step 1: hf 0, 1}
    f32x2 va[R];                          // A operands {K-step 0, K-step 1}, a ring R positions ahead of the MFMAs
    const int NC = a.Cin >> 3;            // chunks of 8 channels
    if ((WINO43_PROBE & 1024) && (slot & 1)) {            // probe: every other workgroup starts ~16 us late (de-synchronised epilogues)
        for (int z = 0; z < 4; ++z) asm volatile("s_sleep 127");
    }
    Tile cur = make_tile(blk), nxt = cur;
#pragma unroll
    for (int k = 0; k < QW; ++k) doff[k] = cur.off[k];
    auto dma_advance = [&](bool cross, bool to_next, const Tile &tn) {       // cross: the chunk just fetched was the block's last one
#pragma unroll
        for (int k = 0; k < QW; ++k) doff[k] = cross ? (to_next ? tn.off[k] : W43_OOB) : doff[k] + 32u;
    };

    // ---- pipeline fill (first block only): raw(0), raw(1), V(0), the filters of positions 0..17 and the first R A operands of chunk 0
    dma(0);
    dma_advance(false, false, cur);                       // (NC >= 4: no block end in here)
    dma(1);
    dma_advance(false, false, cur);
#pragma unroll
    for (int e = 0; e < 18; ++e) bb[e] = ldf(cur.b + e * 64);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the DMAs are invisible to the compiler's own wait; the builtin, so that its
    asm volatile("" ::: "memory");                        // bookkeeping knows the filters have landed (else chunk 0 waits for the previous block's stores)
    __syncthreads();                                      // raw(0), raw(1) landed
    transform(0, 0);
    __syncthreads();                                      // V(0) visible; raw[0] free
#pragma unroll
    for (int e = 0; e < R; ++e) va[e] = v_own[e * 64];

    // chunk C of 8 channels (parity P = C & 1, NC even), 144 slots: slot k <-> MFMA (position q = k >> 2, K-step (k >> 1) & 1, output
    // channel k & 1).  Alongside: DMA raw(C + 2) -> raw[P]; transform raw(C + 1) in raw[!P] -> V[!P]; after a position's four MFMAs its
    // filter registers take position q + 18 (of this chunk, then of chunk C + 1) and its A ring slot takes position q + R — of V[P],
    // then (q + R >= 36) the first positions of chunk C + 1 in V[!P]: those are the wave's OWN rows, written by itself earlier in this
    // chunk, so they need no barrier.  Past the end of the block, "C + 1" / "C + 2" mean chunks 0 / 1 of the NEXT block.
#define W43_CHUNK(C, P, FIRST)                                                                                                     \
    {                                                                                                                              \
        if (!(WINO43_PROBE & 16)) {                                                                                                \
            if ((C) > 0) asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                 \
            else if (!first_block) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                 \
        }                                                                                                                          \
        if (!(WINO43_PROBE & 1)) {                                                                                                 \
            dma(P);                                    /* stream element C + 2 */                                                  \
            dma_advance((C) + 3 == NC, has_next, nxt);                                                                             \
        }                                                                                                                          \
        /* filter addresses = wave-uniform base (scalar registers, advanced by scalar adds: free) + lane * 16: no VALU */         \
        const int bc_ = cur.b + (C) * bstride;                                                                                     \
        const int bn_ = ((C) + 1 < NC) ? cur.b + ((C) + 1) * bstride : (has_next ? nxt.b : cur.b);                                 \
        const float *raw_ = reinterpret_cast<const float *>(s_raw + (1 - (P)) * RP);                                               \
        const f32x2 *vo_ = v_own + (P) * VB, *vp_ = v_prt + (P) * VB;                                                              \
        f32x2 *vd_ = v_own + (1 - (P)) * VB;                                                                                       \
        f32x2 d_[6][5], w_[3][6];                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 144; ++k) {                                                                          \
            const int q = k >> 2, sk = (k >> 1) & 1, hf = k & 1;                                                                   \
            const float av_ = va[q % R][sk];                                                                                       \
            const float bv_ = bb[q % 18][(sk << 1) + hf];                                                                          \
            if (FIRST && sk == 0) {                                                                                                \
                if (q < 32) W43_MFMA_A0(acc[2 * q + hf], av_, bv_); else W43_MFMA_V0(acc[2 * q + hf], av_, bv_);                   \
            } else {                                                                                                               \
                if (q < 32) W43_MFMA_A(acc[2 * q + hf], av_, bv_); else W43_MFMA_V(acc[2 * q + hf], av_, bv_);                     \
            }                                                                                                                      \
            if ((k & 3) == 3) {                                                                                                    \
                if (!(WINO43_PROBE & 2)) bb[q % 18] = q < 18 ? ldf(bc_ + (q + 18) * 64) : ldf(bn_ + (q - 18 < 0 ? 0 : q - 18) * 64);      \
                const int qa = q + R;                                                                                              \
                va[q % R] = qa < 18 ? vo_[(qa < 18 ? qa : 0) * 64] : (qa < 36 ? vp_[(qa >= 18 && qa < 36 ? qa - 18 : 0) * 64]       \
                                                                               : vd_[(qa >= 36 ? qa - 36 : 0) * 64]);              \
            }                                                                                                                      \
            /* input transform of chunk C + 1.  VALU instructions do NOT overlap the MFMAs of their SIMD (tools/ubench/mfma_issue.hip:   \
               one v_fma behind every MFMA costs 12.7 cycles, the same instructions in a burst 4.9 each), LDS reads do (up to two per    \
               MFMA and wave).  So: the 30 patch reads one per slot, and the arithmetic in TWO bursts — first stage of columns 0..2      \
               at slot 20 (their reads: slots 0..14), first stage of columns 3..5 (reads: slots 20..34) + the whole second stage with    \
               its 18 V writes at slot 40; the own-row A reads of the next chunk start at slot 4 (36 - R) + 3 */                          \
            if (!(WINO43_PROBE & 4)) {                                                                                             \
                if (k < 15 || (k >= W43_B1 && k < W43_B1 + 15)) {                                                                               \
                    const int rk = k < 15 ? k : k - W43_B1 + 15;                                                                             \
                    if (WINO43_PROBE & 128) d_[rk / 5][rk % 5] = va[0];                                                            \
                    else d_[rk / 5][rk % 5] = *reinterpret_cast<const f32x2 *>(raw_ + rrow[rk % 5] + W43_COL(rk / 5));             \
                }                                                                                                                  \
                if ((k == W43_B1 || k == W43_B2) && (WINO43_PROBE & 32)) {     /* keep the reads alive */                                  \
                    _Pragma("unroll") for (int x = (k == W43_B1 ? 0 : 3); x < (k == W43_B1 ? 3 : 6); ++x)                                  \
                        _Pragma("unroll") for (int j = 0; j < 5; ++j) asm volatile("" : : "v"(d_[x][j]));                          \
                }                                                                                                                  \
                if ((k == W43_B1 || k == W43_B2) && !(WINO43_PROBE & 32)) {                                                                \
                    _Pragma("unroll") for (int x = (k == W43_B1 ? 0 : 3); x < (k == W43_B1 ? 3 : 6); ++x)                                  \
                        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                            \
                            f32x2 s_ = W43_PK(cf[r][0]) * d_[x][0];                                                                \
                            _Pragma("unroll") for (int j = 1; j < 5; ++j) s_ = W43_FMA(W43_PK(cf[r][j]), d_[x][j], s_);            \
                            w_[r][x] = s_;                                                                                         \
                        }                                                                                                          \
                }                                                                                                                  \
                if (k == W43_B2 && !(WINO43_PROBE & 32)) {                                                                             \
                    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                                \
                        f32x2 o_[6];                                                                                               \
                        W43_S2_0(w_[r], o_) W43_S2_1(w_[r], o_) W43_S2_2(w_[r], o_) W43_S2_34(w_[r], o_) W43_S2_5(w_[r], o_)       \
                        _Pragma("unroll") for (int x = 0; x < 6; ++x) vd_[(r * 6 + x) * 64] = o_[x];                               \
                    }                                                                                                              \
                }                                                                                                                  \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
    }

    // ---- epilogue: Y = A^T M A per (tile, cout), + shift, ReLU.  Four rounds (element i of the accumulator tiles = tile 4 kq + i) of two
    // half rounds (output rows 0, 1 / 2, 3 of the tile): this wave's LDS staging tile [32 pixels][32 channels] -> 16-byte stores (8 lanes =
    // the 128 contiguous bytes of one pixel's 32 channels).  The accumulators are stored in the wave's position order: stored row qr
    // holds transform row xy = (qr + 3 nw) % 6 — one copy of the code per nw (compile-time coefficients; a wave-uniform branch picks).
    //  * the inline-assembly MFMAs are invisible to the compiler's hazard recogniser: two s_nop 15 let the last of them retire before
    //    the first accumulator read;
    //  * `lo`: an opaque copy of the lane id keeps the per-lane store addresses inside the block loop;
    //  * this lane's two output channels are ch0 + 2 (l & 15) + {0, 1}: every value is the PAIR (hf 0, hf 1) — packed fp32 arithmetic
    //    and one 8-byte staging write; only this wave reads its staging tile back (its own LDS operations are ordered, no barrier);
    //  * the DMAs / filter loads of the next block (issued during the last chunk) are drained right before the first store, so that
    //    the next block's first barrier need not wait on the vector-memory counter (= on these stores); through the builtin, so that
    //    the compiler's own bookkeeping sees it too.
#define W43_EPILOGUE(NWV)                                                                                                          \
        {                                                                                                                          \
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                     \
            int lo = l;                                                                                                            \
            asm volatile("" : "+v"(lo));                                                                                           \
            const int ch0 = 32 * cur.nb;                                                                                           \
            const bool relu = a.relu != 0;                                                                                         \
            float *stg = s_stg + (size_t)wv * 32 * W43_STG_PITCH;                                                                  \
            const f32x2 bv = bias2;                                                                                                \
            const int gy0 = 4 * TY * (2 * cur.by + mw), gx0 = 4 * TX * cur.bx;                                                     \
            /* output addressing through a buffer descriptor: per round i a 32-bit byte offset per lane (tile of the lane's staged    \
               pixel), the (row, column) step of each store as a scalar offset, pixels outside the map get an offset beyond the      \
               descriptor's size and are dropped by the hardware; blocks inside the map skip the test (wave-uniform) */              \
            const bool inside = gy0 + 4 * TY <= H && gx0 + 4 * TX <= W;                                                            \
            int lane_ry[4], lane_cx[4];                                                                                            \
            unsigned lane_off[4];                                                                                                  \
_Pragma("unroll")                                                                                                                  \
            for (int i = 0; i < 4; ++i) {                                                                                          \
                const int mt = 4 * ((lo >> 3) & 3) + i;                                                                            \
                lane_ry[i] = gy0 + 4 * (mt / TX);                                                                                  \
                lane_cx[i] = gx0 + 4 * (mt % TX) + (lo >> 5);                                                                      \
                lane_off[i] = (unsigned)(((cur.bidx * H + lane_ry[i]) * W + lane_cx[i]) * a.out_C + a.out_off + ch0 + 4 * (lo & 7)) * 4u; \
            }                                                                                                                      \
            const float at[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f},                            \
                                    {1.f, 0.5f, 0.25f, 0.125f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};                     \
_Pragma("unroll")                                                                                                                  \
            for (int i = 0; i < 4; ++i) {                                                                                          \
                f32x2 y[4][4];                                                                                                     \
_Pragma("unroll")                                                                                                                  \
                for (int ii = 0; ii < 4; ++ii)                                                                                     \
_Pragma("unroll")                                                                                                                  \
                    for (int j = 0; j < 4; ++j) y[ii][j] = bv;                                                                     \
_Pragma("unroll")                                                                                                                  \
                for (int qr = 0; qr < 6; ++qr) {                                                                                   \
                    f32x2 mm[6];                                                                                                   \
_Pragma("unroll")                                                                                                                  \
                    for (int x = 0; x < 6; ++x) {                                                                                  \
                        const int q = 6 * qr + x;                                                                                  \
                        if (q < 32) {                                                                                              \
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(mm[x][0]) : "a"(acc[2 * q][i]));                       \
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(mm[x][1]) : "a"(acc[2 * q + 1][i]));                   \
                        } else {                                                                                                   \
                            mm[x] = (f32x2){acc[2 * q][i], acc[2 * q + 1][i]};                                                     \
                        }                                                                                                          \
                    }                                                                                                              \
                    const f32x2 s = mm[1] + mm[2], d = mm[1] - mm[2];                                                              \
                    f32x2 tt[4];                                                                                                   \
                    tt[0] = (mm[0] + s) + (mm[3] + mm[4]);                                                                         \
                    tt[1] = W43_FMA(W43_PK(0.5f), mm[3], W43_FMA(W43_PK(-2.f), mm[4], d));                                         \
                    tt[2] = W43_FMA(W43_PK(0.25f), mm[3], W43_FMA(W43_PK(4.f), mm[4], s));                                         \
                    tt[3] = W43_FMA(W43_PK(0.125f), mm[3], W43_FMA(W43_PK(-8.f), mm[4], d)) + mm[5];                               \
_Pragma("unroll")                                                                                                                  \
                    for (int ii = 0; ii < 4; ++ii) {                                                                               \
                        const float c_own = at[qr][ii], c_oth = at[(qr + 3) % 6][ii];                                              \
                        if ((NWV ? c_oth : c_own) == 0.f) continue;                                                                \
                        const float cfa = NWV ? c_oth : c_own;                                                                     \
_Pragma("unroll")                                                                                                                  \
                        for (int j = 0; j < 4; ++j) {                                                                              \
                            if (cfa == 1.f) y[ii][j] += tt[j];                                                                     \
                            else if (cfa == -1.f) y[ii][j] -= tt[j];                                                               \
                            else y[ii][j] = W43_FMA(W43_PK(cfa), tt[j], y[ii][j]);                                                 \
                        }                                                                                                          \
                    }                                                                                                              \
                    __builtin_amdgcn_sched_barrier(0);                                                                             \
                }                                                                                                                  \
_Pragma("unroll")                                                                                                                  \
                for (int hh = 0; hh < 2; ++hh) {                                                                                   \
_Pragma("unroll")                                                                                                                  \
                    for (int ii = 0; ii < 2; ++ii)                                                                                 \
_Pragma("unroll")                                                                                                                  \
                        for (int j = 0; j < 4; ++j) {                                                                              \
                            const f32x2 yv = y[2 * hh + ii][j];                                                                    \
                            const f32x2 v = relu ? (f32x2){fmaxf(yv[0], 0.f), fmaxf(yv[1], 0.f)} : yv;                             \
                            *reinterpret_cast<f32x2 *>(stg + ((ii * 4 + j) * 4 + (lo >> 4)) * W43_STG_PITCH + 2 * (lo & 15)) = v;  \
                        }                                                                                                          \
                    if (i == 0 && hh == 0 && !(WINO43_PROBE & 512)) __builtin_amdgcn_s_waitcnt(0x0F70);                                                     \
                    float4 sv_[4];                                                                                                 \
_Pragma("unroll")                                                                                                                  \
                    for (int k = 0; k < 4; ++k) sv_[k] = *reinterpret_cast<const float4 *>(stg + (k * 8 + (lo >> 3)) * W43_STG_PITCH + 4 * (lo & 7)); \
_Pragma("unroll")                                                                                                                  \
                    for (int k = 0; k < 4; ++k) {                                                                                  \
                        const int dyk = 2 * hh + (k >> 1), dxk = 2 * (k & 1);                                                      \
                        unsigned vo_ = lane_off[i];                                                                                \
                        if (!inside) vo_ = (lane_ry[i] + dyk < H && lane_cx[i] + dxk < W) ? vo_ : W43_OOB;                        \
                        if (!(WINO43_PROBE & 256))                                                                                 \
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv_[k]), orsrc, (int)vo_, (dyk * W + dxk) * a.out_C * 4, W43_STORE_AUX); \
                    }                                                                                                              \
                    __builtin_amdgcn_sched_barrier(0);                                                                             \
                }                                                                                                                  \
            }                                                                                                                      \
        }                                                                                                                          \

    bool first_block = true;
    for (;;) {
        const int blk_next = blk + nslots;
        const bool has_next = blk_next < blk_end;
        if (has_next) nxt = make_tile(blk_next);
        // this block's shift values, fetched now: by the epilogue they have long arrived (fetched there, the wait for them was a wait
        // for every filter load and DMA of the next block issued just before)
        const f32x2 bias2 = a.bias ? *reinterpret_cast<const f32x2 *>(a.bias + 32 * cur.nb + 2 * (l & 15)) : (f32x2){0.f, 0.f};
        // a no-op for the hardware (the epilogue drained every load before its 32 stores; + the shift load above = 33 operations in
        // flight at most), but hipcc's wait bookkeeping loses the epilogue's vmcnt(0) across the loop's back edge and would make chunk 0
        // wait for the filter registers with "vmcnt(17)" = for the previous block's STORES; this tells it what is known
        __builtin_amdgcn_s_waitcnt(0x8F71);                // vmcnt(33)
        W43_CHUNK(0, 0, true)
        W43_CHUNK(1, 1, false)
        for (int c = 2; c < NC; c += 2) {
            W43_CHUNK(c, 0, false)
            W43_CHUNK(c + 1, 1, false)
        }
#if !(WINO43_PROBE & 8)
        if (nw == 0) W43_EPILOGUE(0) else W43_EPILOGUE(1)
#else
        {
            float sum = 0.f;
#pragma unroll
            for (int q = 32; q < 36; ++q) sum += acc[2 * q][0] + acc[2 * q + 1][1];
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
#undef W43_EPILOGUE
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_wino43_packed_floats(int Cin, int Cout) {
    if (Cin < 32 || Cout <= 0 || (Cin & 15) || (Cout & 63)) return 0;      // two 8-channel chunks per loop iteration, the input stream runs
                                                                           // two chunks ahead (>= 4 per block); 64 channels per workgroup
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
    if ((long long)B * H * W * out_C * 4 >= 0x7fffffffll) return LIDAR_ERR_ARG;                             // ... and into the output map (W43_OOB + a store's scalar step must not wrap)
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
