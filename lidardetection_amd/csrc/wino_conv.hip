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
// A WAVE owns 32 tiles (4 x 8 tiles = 8 x 16 output pixels) x 32 output channels x all 16 positions: 16 accumulator tiles of
// v_mfma_f32_32x32x2_f32 = 256 accumulator registers — one wave per SIMD, 512 registers each.  In the accumulator layout the 16
// positions of one (tile, cout) sit in the SAME lane at the same register index of the 16 tiles, so the output transform is
// lane-local arithmetic: no shuffles, no LDS.
//   A operand: lane (i = lane & 31, h = lane >> 5) supplies tile i, input channel 4c + 2h + s of chunk c, K-step s — it reads the
//              16 raw pixels of its patch (two channels each, ds_read_b64) from the workgroup's LDS image of the input region and
//              transforms them in registers (32 packed adds per chunk).
//   B operand: the transformed filters are packed ON THE HOST SIDE (lidar_wino_pack_weights, once per weight update) in exactly
//              the order the lanes consume them — [chunk][cout / 32][xi][lane][s] — so a wave's load is one contiguous 512 B.
//   LDS image of the input region ((8 MW + 2) x 18 pixels x 4 channels per chunk): filled by LDS-DMA (global_load_lds_dwordx4,
//              one pixel per lane, zero padding by pointing out-of-image lanes at a zero word), three buffers in a ring, ONE
//              workgroup barrier per chunk: chunk c + 2 is in flight and chunk c + 1 is being read while chunk c is multiplied.
// Workgroup = 4 waves = MW x NW (rows of tiles x groups of 32 output channels): 2 x 2 for 64 output channels, 1 x 4 from 128.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WINO_RW 18                       // region width in pixels: 8 tiles x 2 + 2 halo

__device__ float4 g_wino_zero = {0.f, 0.f, 0.f, 0.f};      // what an out-of-image pixel reads (zero padding)

// ------------------------------------------------------------------ filter transform + packing
// w: (Cout, Cin, 3, 3) contiguous.  upk[(((c * NB + nb) * 16 + xi) * 64 + lane) * 2 + s] = U_xi[cin = 4c + 2 (lane >> 5) + s][cout = 32 nb + (lane & 31)]
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
        for (int x = 0; x < 4; ++x)
            upk[((((size_t)c * NB + nb) * 16 + (y * 4 + x)) * 64 + (h * 32 + j)) * 2 + s] = u[x];
    }
}

// ------------------------------------------------------------------ the convolution
struct WinoArgs {
    const float *in;         // (B, H, W, Cin)
    const float *upk;        // packed transformed filters
    const float *bias;       // (Cout) or null
    float *out;              // (B, H, W, out_C), this layer's channels at [out_off, out_off + Cout)
    int B, H, W, Cin, Cout, out_C, out_off, relu;
    int blocks_y, blocks_x, n_groups, n_blocks;       // grid decomposition (n_blocks = B * blocks_y * blocks_x * n_groups)
};

template <int NW>
__global__ __launch_bounds__(256) void wino_f23_kernel(const WinoArgs a) {
    constexpr int MW = 4 / NW;
    constexpr int RH = 8 * MW + 2;                       // region rows
    constexpr int RP = ((RH * WINO_RW + 63) / 64) * 64;  // region pixels, padded to whole DMA wave-instructions
    constexpr int NQ = RP / 64;                          // DMA wave-instructions per chunk and workgroup
    constexpr int QW = (NQ + 3) / 4;                     // ... per wave
    __shared__ float4 s_a[3][RP];                        // ring of region images: [pixel slot][4 channels of the chunk]

    // XCD-aware block order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs; blocks that share input
    // (the channel groups of one spatial block, horizontally adjacent blocks: halo) should share an L2, so consecutive LOGICAL
    // blocks are given to the same XCD
    const int nb8 = (a.n_blocks + 7) >> 3;
    int blk = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
    if (blk >= a.n_blocks) return;
    const int ng = blk % a.n_groups;
    blk /= a.n_groups;
    const int bx = blk % a.blocks_x;
    blk /= a.blocks_x;
    const int by = blk % a.blocks_y;
    const int b = blk / a.blocks_y;

    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);     // (wave-uniform, and known to be)
    const int mw = wv / NW, nw = wv % NW;
    const int i = l & 31, h = l >> 5;
    const int ty = i >> 3, tx = i & 7;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int R0 = 8 * MW * by - 1, C0 = 16 * bx - 1;    // image coordinates of region pixel (0, 0)

    // ---- this lane's DMA sources: region slots p = q * 64 + l for q = wv, wv + 4, ...
    const float *src[QW];
    bool src_ok[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = wv + 4 * k;
        const int p = q * 64 + l;
        const int ry = p / WINO_RW, rx = p - ry * WINO_RW;
        const int gy = R0 + ry, gx = C0 + rx;
        src_ok[k] = (q < NQ) && (p < RH * WINO_RW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        src[k] = src_ok[k] ? a.in + (((size_t)b * H + gy) * W + gx) * Cin : (const float *)&g_wino_zero;
    }
    auto dma = [&](int c, int buf) {
#pragma unroll
        for (int k = 0; k < QW; ++k) {
            const int q = wv + 4 * k;
            if (q < NQ) {                                 // wave-uniform
                const float *g = src_ok[k] ? src[k] + 4 * c : src[k];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(&s_a[buf][q * 64]), 16, 0, 0);
            }
        }
    };
    // ---- B operand: 16 x float2 per chunk, contiguous per wave
    const int NB = a.Cout >> 5, nb = ng * NW + nw;
    const f32x2 *bsrc = reinterpret_cast<const f32x2 *>(a.upk) + ((size_t)nb * 16) * 64 + l;
    const size_t bstride = (size_t)NB * 16 * 64;         // float2 per chunk
    auto load_b = [&](int c, f32x2 (&bb)[16]) {
        const f32x2 *p = bsrc + (size_t)c * bstride;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) bb[xi] = p[xi * 64];
    };
    // ---- A operand: raw patch from the LDS image, transformed in registers (B^T d B, two channels packed per register pair)
    const int slot0 = (8 * mw + 2 * ty) * WINO_RW + 2 * tx;
    auto load_v = [&](int buf, f32x2 (&v)[16]) {
        const f32x2 *base = reinterpret_cast<const f32x2 *>(&s_a[buf][slot0]) + h;
        f32x2 d[4][4];
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) d[y][x] = base[(y * WINO_RW + x) * 2];
#pragma unroll
        for (int x = 0; x < 4; ++x) {                    // columns: B^T d
            const f32x2 t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
            d[0][x] = t0; d[1][x] = t1; d[2][x] = t2; d[3][x] = t3;
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {                    // rows: (.) B
            v[y * 4 + 0] = d[y][0] - d[y][2];
            v[y * 4 + 1] = d[y][1] + d[y][2];
            v[y * 4 + 2] = d[y][2] - d[y][1];
            v[y * 4 + 3] = d[y][1] - d[y][3];
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    // ---- main loop.  Chunk c multiplies (vc, bc) while (vn, bn) of chunk c + 1 are fetched / transformed and the DMA of chunk
    // c + 2 is in flight; two chunks per iteration so that the register sets swap roles without copies.  One basic block per
    // chunk (clamped instead of conditional loads), with the issue order pinned: the LDS reads of the next patch behind the first
    // MFMAs, its transform (32 packed adds) and the next filter loads spread under the rest — an MFMA occupies the matrix pipe for
    // 64 cycles, the wave issues two or three other instructions in that time.
    const int NC = Cin >> 2;
    f32x2 b0[16], b1[16], v0[16], v1[16];
    dma(0, 0);
    dma(1, 1);
    load_b(0, b0);
    __syncthreads();                                      // (drains the DMA: vmcnt(0) + barrier)
    load_v(0, v0);
    int ring = 2;                                         // buffer that receives chunk c + 2; chunk c + 1 sits in (ring + 2) % 3
#define WINO_CHUNK(c, VC, BC, VN, BN)                                                                                              \
    {                                                                                                                              \
        if ((c) > 0) __syncthreads();      /* image c + 1 has landed for every wave; image c - 1 is no longer read */             \
        if ((c) + 2 < NC) dma((c) + 2, ring);                                                                                      \
        const int nxt = ring == 0 ? 2 : ring - 1;                                                                                  \
        ring = ring == 2 ? 0 : ring + 1;                                                                                           \
        load_b(min((c) + 1, NC - 1), BN);                                                                                          \
        _Pragma("unroll") for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(VC[xi][0], BC[xi][0], acc[xi], 0, 0, 0); \
        load_v(nxt, VN);                                                                                                           \
        _Pragma("unroll") for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(VC[xi][1], BC[xi][1], acc[xi], 0, 0, 0); \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     /* MFMA */                                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     /* DS read */                                                  \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     /* VMEM read */                                                \
        }                                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     /* VALU */                                                     \
        }                                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                     \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                                     \
        }                                                                                                                          \
    }
    for (int c = 0; c < NC; c += 2) {
        WINO_CHUNK(c, v0, b0, v1, b1)
        WINO_CHUNK(c + 1, v1, b1, v0, b0)
    }
#undef WINO_CHUNK

    // ---- epilogue: Y = A^T M A per (tile, cout), + shift, ReLU, store.  acc[xi][r]: tile row 8 (r / 4) + 4 h + (r % 4), cout l & 31
    const int ch = 32 * nb + i;
    const float bv = a.bias ? a.bias[ch] : 0.f;
    const bool relu = a.relu != 0;
    float *obase = a.out + (size_t)b * H * W * a.out_C + a.out_off + ch;
    const int tile_y0 = (4 * MW * by + 4 * mw), tile_x0 = 8 * bx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int it = 8 * (r >> 2) + 4 * h + (r & 3);
        const int oy = 2 * (tile_y0 + (it >> 3)), ox = 2 * (tile_x0 + (it & 7));
        float tt[2][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            tt[0][x] = acc[0 + x][r] + acc[4 + x][r] + acc[8 + x][r];
            tt[1][x] = acc[4 + x][r] - acc[8 + x][r] - acc[12 + x][r];
        }
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            float y0 = tt[yy][0] + tt[yy][1] + tt[yy][2] + bv;
            float y1 = tt[yy][1] - tt[yy][2] - tt[yy][3] + bv;
            if (relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            if (oy + yy < H) {
                float *o = obase + ((size_t)(oy + yy) * W + ox) * a.out_C;
                if (ox < W) o[0] = y0;
                if (ox + 1 < W) o[a.out_C] = y1;
            }
        }
    }
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_wino_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 31)) return 0;       // two 4-channel chunks per loop iteration
    return (size_t)16 * Cin * Cout;
}

LIDAR_EXPORT int lidar_wino_supported(int Cin, int Cout) { return lidar_wino_packed_floats(Cin, Cout) != 0; }

// w: (Cout, Cin, 3, 3) contiguous fp32 (the folded convolution weight) -> packed: lidar_wino_packed_floats(Cin, Cout) floats
LIDAR_EXPORT int lidar_wino_pack_weights(const float *w, int Cin, int Cout, float *packed, void *stream) {
    if (!w || !packed || !lidar_wino_supported(Cin, Cout)) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(divup((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, packed);
    return lidar_check_launch("lidar_wino_pack_weights");
}

// out[b][y][x][out_off + co] = act(sum_{ky,kx,ci} in[b][y+ky-1][x+kx-1][ci] * w[co][ci][ky][kx] + bias[co])   (zero padding)
LIDAR_EXPORT int lidar_wino_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, const float *packed, const float *bias, int relu,
                                         int Cout, float *out, int out_C, int out_off, void *stream) {
    if (!in || !packed || !out || B <= 0 || H <= 0 || W <= 0 || !lidar_wino_supported(Cin, Cout) || out_off < 0 || out_off + Cout > out_C)
        return LIDAR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(packed) & 7)) return LIDAR_ERR_ARG;
    WinoArgs a;
    a.in = in; a.upk = packed; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.out_C = out_C; a.out_off = out_off; a.relu = relu;
    const int NW = (Cout % 128 == 0) ? 4 : (Cout % 64 == 0) ? 2 : 1, MW = 4 / NW;
    const int tiles_y = (H + 1) / 2, tiles_x = (W + 1) / 2;
    a.blocks_y = divup(tiles_y, 4 * MW);
    a.blocks_x = divup(tiles_x, 8);
    a.n_groups = Cout / (32 * NW);
    const long long nblk = (long long)B * a.blocks_y * a.blocks_x * a.n_groups;
    if (nblk > 0x7ffffff0ll) return LIDAR_ERR_ARG;
    a.n_blocks = (int)nblk;
    const dim3 grid((unsigned)(((nblk + 7) / 8) * 8)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (NW == 4) hipLaunchKernelGGL(wino_f23_kernel<4>, grid, blk, 0, s, a);
    else if (NW == 2) hipLaunchKernelGGL(wino_f23_kernel<2>, grid, blk, 0, s, a);
    else hipLaunchKernelGGL(wino_f23_kernel<1>, grid, blk, 0, s, a);
    return lidar_check_launch("lidar_wino_conv3x3_nhwc");
}
