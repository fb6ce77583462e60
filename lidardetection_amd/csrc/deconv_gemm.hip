// ConvTranspose2d with kernel == stride == s (every deblock of BaseBEVBackbone: pcdet/models/backbones_2d/base_bev_backbone.py:51-57,
// UPSAMPLE_STRIDES 1 / 2 / 4) + folded BatchNorm shift + ReLU + the write into the layer's channel slice of the concatenated map
// (base_bev_backbone.py:103 torch.cat(ups, dim=1)) as ONE kernel on the fp32 matrix cores.
//
// With kernel == stride every input pixel owns its own s x s output patch, so the layer is a plain GEMM on the NHWC map:
//   Y[p][(ky, kx, c)] = sum_k X[p][k] W[k][(ky, kx, c)],   out[b][s y + ky][s x + kx][off + c] = act(Y + shift[c])       p = (b, y, x)
// r03 ran it as a library GEMM into a temporary (P x s^2 C: 438 MB for both PointPillar deblocks) followed by this repo's pixel-
// shuffle pass (read 438 MB, write 438 MB); the GEMMs ran at 53-67 TFLOP/s and the passes were pure HBM time — together 2 of the
// step's 6.6 ms (profiles/r04/bench_step_timeline.txt).  Here the accumulators go straight to their place in the concatenated map:
// the temporary and the pass disappear, every output byte is written once, in 512-byte runs.
//
// Same machinery as csrc/wino_conv.hip (one wave per SIMD, 256 accumulator registers = a 32-pixel x 512-column tile per wave,
// LDS-DMA ring, one barrier per 4-channel chunk, issue order written out MFMA by MFMA, persistent over tile blocks) without the
// transforms: the four waves of a workgroup take four consecutive 32-pixel row blocks and use the same B operand — a chunk's 4 x 512
// weights (8 KB, packed once per weight update in lane order): eight 16-byte buffer loads per wave straight into the next chunk's
// registers (end of r04; before: DMA'd to LDS once per workgroup and read back); the A operand is one 8-byte LDS load per lane and
// chunk from the LDS-DMA image of the block's pixels.  Per chunk and wave: 32 MFMAs against 8 buffer loads, 1 LDS read, <= 1 DMA.
#include "common.h"
#ifndef DC_PROBE                         // timing probe (WRONG results): bit 0 = the chunk barrier does not wait for the DMA
#define DC_PROBE 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DC_OOB 0x80000000u                // a byte offset no input map reaches (the launcher keeps it below 2^31 bytes): the DMA reads zeros

// W (K, N) row-major, columns ordered (ky, kx, c).
// packed[((((c * NG + g) * 8 + e) * 64 + lane) * 4 + (t & 1) * 2 + s] = W[4 c + 2 (lane >> 5) + s][512 g + 32 (2 e + (t & 1)) + (lane & 31)]
//   (t = column tile 0..15 of the group, e = t >> 1)
__global__ __launch_bounds__(256) void dc_pack_kernel(const float *__restrict__ W, int K, int N, float *__restrict__ P) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)K * N) return;
    const int k = (int)(idx / N), col = (int)(idx - (long long)k * N);
    const int c = k >> 2, h = (k >> 1) & 1, s = k & 1, g = col >> 9, t = (col >> 5) & 15, j = col & 31, NG = N >> 9;
    P[(((((size_t)c * NG + g) * 8 + (t >> 1)) * 64 + (h * 32 + j)) * 2 + (t & 1)) * 2 + s] = W[idx];
}

struct DcArgs {
    const float *in;         // (P, K) = the NHWC map (B, h, w, K)
    const float *pk;         // packed weights
    const float *bias;       // (C_up) or null
    float *out;              // (B, s h, s w, out_C), channels [out_off, out_off + C_up)
    int P, K, N, h, w, s, C_up, out_C, out_off, relu;
    int n_pblocks, n_groups, n_blocks;       // pixel blocks of 128, column groups of 512, n_blocks = n_pblocks * n_groups
    unsigned out_bytes;                      // size of the output map (< 2^31: 32-bit store offsets)
};

#define DC_STG_PITCH 132                 // floats per staged pixel: 128 channels + 4 (16-byte aligned rows, shifted banks)
static constexpr size_t dc_lds_bytes() { return (size_t)2 * 128 * 16 + (size_t)4 * 32 * DC_STG_PITCH * 4; }

__global__ __launch_bounds__(256) void dc_gemm_kernel(const DcArgs a) {
    extern __shared__ float4 s_mem4[];
    float4 *s_a = s_mem4;                                 // [2][128]: A images, [pixel of the block][4 channels of the chunk]
    float *s_stg = reinterpret_cast<float *>(s_mem4 + 2 * 128);    // [4 waves][32 pixels][DC_STG_PITCH]
    const int t = threadIdx.x, l = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = l & 31, h = l >> 5;
    const int K = a.K, NC = K >> 2;
    const int bstride = a.n_groups * 8 * 64;             // float4 per chunk

    const int nb8 = (a.n_blocks + 7) >> 3;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)a.out_bytes, 0x00020000);
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int blk_end = min((xcd + 1) * nb8, a.n_blocks);
    int blk = xcd * nb8 + slot;
    if (blk >= blk_end) return;

    struct Tile {
        unsigned aoff;                   // byte offset of this lane's A DMA source (waves 0 / 1: pixels 64 wv + l of the block), chunk 0; DC_OOB: no pixel
        int b;                           // float4 index of the group's packed weights, chunk 0 (wave-uniform)
        int p0, g;
    };
    auto make_tile = [&](int blk_) {
        Tile tl;
        tl.g = blk_ % a.n_groups;                         // column groups of one pixel block are neighbours: they share its A rows in L2
        const int pb = blk_ / a.n_groups;
        tl.p0 = pb * 128;
        const int p = tl.p0 + 64 * (wv & 1) + l;
        tl.aoff = p < a.P ? (unsigned)p * (unsigned)K * 4u : DC_OOB;
        tl.b = tl.g * 8 * 64;
        return tl;
    };
    // r04 (end): the issue model of csrc/wino43_conv.hip applied.  B — the same 8 KB for every wave — is no longer DMA'd to LDS and read
    // back (2 DMA instructions per wave at ~80 issue cycles each + 8 LDS reads) but loaded straight into the next chunk's registers
    // through a buffer descriptor (scalar offset + lane * 16: ~7 issue cycles each, no VALU); A alone goes through LDS, DMA'd by waves
    // 0 / 1 as inline assembly through a descriptor of the input map (32-bit running offset, pixels past the end read zeros) so that
    // hipcc's wait bookkeeping stays exact, and the chunk barrier waits "vmcnt(8)": for the DMA, not for the 8 loads behind it.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 arsrc;
    {
        const unsigned long long base = (unsigned long long)a.in;
        arsrc[0] = (int)(unsigned)base;
        arsrc[1] = (int)((unsigned)(base >> 32) & 0xffffu);
        arsrc[2] = (int)((unsigned)a.P * (unsigned)K * 4u);
        arsrc[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.pk), (short)0, (int)((unsigned)K * (unsigned)a.N * 4u), 0x00020000);
    const int l16 = l * 16;
    auto ldb = [&](int f4_index) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, l16, f4_index * 16, 0)); };
    const unsigned a_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char *)s_a);
    unsigned doff = 0;                    // the A stream's next chunk (waves 0 / 1)
    auto dma = [&](int buf) {
        if (wv < 2) {
            const unsigned dst = a_lds + (unsigned)((buf * 128 + 64 * wv) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(doff), "s"(arsrc), "s"(dst) : "memory", "m0");
        }
    };
    auto load_a = [&](int buf, f32x2 &av) { av = reinterpret_cast<const f32x2 *>(s_a + buf * 128 + 32 * wv + i)[h]; };

    f32x16 acc[16];
    f32x2 a0, a1;
    f32x4 b0[8], b1[8];
    Tile cur = make_tile(blk), nxt = cur;
    doff = cur.aoff;
    dma(0);
    doff += 16u;
    dma(1);
    doff += 16u;
#pragma unroll
    for (int e = 0; e < 8; ++e) b0[e] = ldb(cur.b + e * 64);
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the DMAs are invisible to the compiler's own wait
    asm volatile("" ::: "memory");
    __syncthreads();                                      // chunks 0, 1 landed
    load_a(0, a0);
    __syncthreads();                                      // every wave has read image 0: chunk 0's DMA of chunk 2 may overwrite it

    // chunk c (parity P = c & 1, NC even): A of chunk c + 1 is read from image !P and B of chunk c + 1 loaded from memory into the
    // other register set, the DMA of chunk c + 2 goes to image P (chunk c's data there was read during chunk c - 1; every wave has
    // passed this chunk's barrier since).  Past the end of the block "c + k" is chunk c + k - NC of the NEXT block (same parities).
    // Column tile n, K-step s: B piece n >> 1, component 2 (n & 1) + s.  FIRST: the block's first chunk starts from a zero C operand.
#define DC_CHUNK(c, P, FIRST, AC, BC, AN, BN)                                                                                      \
    {                                                                                                                              \
        if ((c) > 0) {                                                                                                             \
            if (DC_PROBE & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                      \
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
        } else if (!first_block) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                   \
        dma(P);                                        /* A stream element c + 2 */                                                \
        doff = ((c) + 3 == NC) ? (has_next ? nxt.aoff : DC_OOB) : doff + 16u;                                                      \
        const int bn_ = ((c) + 1 < NC) ? cur.b + ((c) + 1) * bstride : (has_next ? nxt.b : cur.b);                                 \
        const f32x2 *as_ = reinterpret_cast<const f32x2 *>(s_a + (1 - (P)) * 128 + 32 * wv + i) + h;                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                                           \
            const float bop_ = BC[(k & 15) >> 1][((k & 1) << 1) + (k >> 4)];                                                       \
            if (FIRST && k < 16) {                                                                                                 \
                f32x16 z_;                                                                                                         \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) z_[r_] = 0.f;                                                    \
                acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[k >> 4], bop_, z_, 0, 0, 0);                                 \
            } else {                                                                                                               \
                acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[k >> 4], bop_, acc[k & 15], 0, 0, 0);                        \
            }                                                                                                                      \
            if (k < 8) BN[k] = ldb(bn_ + k * 64);                                                                                  \
            if (k == 2) AN = as_[0];                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
    }

    bool first_block = true;
    for (;;) {
        const int blk_next = blk + nslots;
        const bool has_next = blk_next < blk_end;
        if (has_next) nxt = make_tile(blk_next);
        DC_CHUNK(0, 0, true, a0, b0, a1, b1)
        DC_CHUNK(1, 1, false, a1, b1, a0, b0)
        for (int c = 2; c < NC; c += 2) {
            DC_CHUNK(c, 0, false, a0, b0, a1, b1)
            DC_CHUNK(c + 1, 1, false, a1, b1, a0, b0)
        }
        // ---- epilogue: four rounds of four column tiles = 128 consecutive output channels of ONE (ky, kx) = 512 contiguous bytes
        // per pixel; staged per wave, stored as 16-byte pieces (32 lanes = one pixel's run, two pixels per instruction)
        {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the next block's DMAs are drained before the stores are issued
            int lo = l;
            asm volatile("" : "+v"(lo));
            const bool relu = a.relu != 0;
            float *stg = s_stg + (size_t)wv * 32 * DC_STG_PITCH;
            const int hw = a.h * a.w, OW = a.s * a.w;
            // output addressing through a buffer descriptor: the byte offset of each of this lane's 16 staged pixels (rows lo >> 5,
            // + 2, ...) is computed ONCE per block — one division pair, then steps with carries; per pixel and round that was two
            // integer divisions, ~50 VALU instructions each, a fifth of the 128-channel deblock — and the (ky, kx, channel) part of a
            // round is a scalar offset.  Pixels past the end get an offset beyond the descriptor's size: the store is dropped.
            unsigned pix[16];
            {
                const int p = cur.p0 + 32 * wv + (lo >> 5);
                int bb = p / hw;
                const int rem = p - bb * hw;
                int y = rem / a.w, x = rem - y * a.w;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    pix[k] = (p + 2 * k < a.P) ? (unsigned)(((bb * a.s * a.h + a.s * y) * OW + a.s * x) * a.out_C + a.out_off + 4 * (lo & 31)) * 4u : 0x80000000u;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {      // two pixels on
                        x += 1;
                        if (x >= a.w) { x = 0; y += 1; }
                        if (y >= a.h) { y = 0; bb += 1; }
                    }
                }
            }
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) {
                const int col0 = 512 * cur.g + 128 * rnd;                     // first column of the round
                const int kk = col0 / a.C_up, c0 = col0 - kk * a.C_up;        // (ky, kx) index and first channel
                const int ky = kk / a.s, kx = kk - ky * a.s;
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const float bvv = a.bias ? a.bias[c0 + 32 * nn + i] : 0.f;
                    float m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[r]) : "a"(acc[4 * rnd + nn][r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 8 * (r >> 2) + 4 * h + (r & 3);
                        const float v = m[r] + bvv;
                        stg[row * DC_STG_PITCH + 32 * nn + i] = relu ? fmaxf(v, 0.f) : v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int soff = ((ky * OW + kx) * a.out_C + c0) * 4;          // wave-uniform
#pragma unroll
                for (int k4 = 0; k4 < 16; k4 += 4) {      // four staged rows at a time: the LDS reads first, then the stores (a read right
                    f32x4 sv[4];                          // before its store exposes the LDS latency sixteen times per round)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[k] = *reinterpret_cast<const f32x4 *>(stg + (2 * (k4 + k) + (lo >> 5)) * DC_STG_PITCH + 4 * (lo & 31));
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[k]), orsrc, (int)pix[k4 + k], soff, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        blk = blk_next;
        first_block = false;
    }
#undef DC_CHUNK
}

// ------------------------------------------------------------------ C ABI
LIDAR_EXPORT size_t lidar_deconv_packed_floats(int K, int N) {
    if (K < 16 || N <= 0 || (K & 7) || (N & 511)) return 0;            // two 4-channel chunks per loop iteration (the A stream runs two chunks
                                                                       // ahead, across blocks: at least four per block), 512-column groups
    return (size_t)K * N;
}

LIDAR_EXPORT int lidar_deconv_supported(int K, int s, int C_up) {
    if (s <= 0 || C_up <= 0 || (C_up & 127)) return 0;
    return lidar_deconv_packed_floats(K, s * s * C_up) != 0;
}

// W: (K, s * s * C_up) row-major, columns ordered (ky, kx, c) — the folded ConvTranspose2d weight as FoldedBEVBackbone holds it
LIDAR_EXPORT int lidar_deconv_pack_weights(const float *W, int K, int N, float *packed, void *stream) {
    if (!W || !packed || lidar_deconv_packed_floats(K, N) == 0) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(dc_pack_kernel, dim3(divup((long long)K * N, 256)), dim3(256), 0, (hipStream_t)stream, W, K, N, packed);
    return lidar_check_launch("lidar_deconv_pack_weights");
}

static int dc_cu_count() {
    static int cus[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (cus[dev] == 0) cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cus[dev];
}

// out[b][s y + ky][s x + kx][out_off + c] = act(sum_k in[b][y][x][k] W[k][(ky, kx, c)] + bias[c])
LIDAR_EXPORT int lidar_deconv_gemm_nhwc(const float *in, int B, int h, int w, int K, const float *packed, const float *bias, int relu, int s,
                                        int C_up, float *out, int out_C, int out_off, void *stream) {
    if (!in || !packed || !out || B <= 0 || h <= 0 || w <= 0 || !lidar_deconv_supported(K, s, C_up) || out_off < 0 || out_off + C_up > out_C)
        return LIDAR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
        (out_C & 3) || (out_off & 3) || (long long)B * h * w > 0x7fffffffll || (long long)B * h * w * s * s * out_C * 4 >= 0x7fffffffll ||
        (long long)B * h * w * K * 4 >= 0x7fffffffll || (long long)K * s * s * C_up * 4 >= 0x7fffffffll)      // 32-bit byte offsets: output, input, weights
        return LIDAR_ERR_ARG;
    DcArgs a;
    a.in = in; a.pk = packed; a.bias = bias; a.out = out;
    a.P = B * h * w; a.K = K; a.N = s * s * C_up; a.h = h; a.w = w; a.s = s; a.C_up = C_up; a.out_C = out_C; a.out_off = out_off; a.relu = relu;
    a.out_bytes = (unsigned)((long long)B * h * w * s * s * out_C * 4);
    a.n_pblocks = divup(a.P, 128);
    a.n_groups = a.N / 512;
    const long long nblk = (long long)a.n_pblocks * a.n_groups;
    if (nblk > 0x7ffffff0ll) return LIDAR_ERR_ARG;
    a.n_blocks = (int)nblk;
    long long want = ((nblk + 7) / 8) * 8;
    const long long cap = ((long long)dc_cu_count() / 8) * 8;
    if (cap >= 8 && want > cap) want = cap;
    static bool attr_set[64] = {};                        // the LDS opt-in is a per-device function attribute
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    dev_id &= 63;
    if (!attr_set[dev_id]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dc_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dc_lds_bytes());
        attr_set[dev_id] = true;
    }
    hipLaunchKernelGGL(dc_gemm_kernel, dim3((unsigned)want), dim3(256), dc_lds_bytes(), (hipStream_t)stream, a);
    return lidar_check_launch("lidar_deconv_gemm_nhwc");
}
