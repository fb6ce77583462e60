// Inference epilogue of the dense BEV backbone: the eval-mode BatchNorm2d + ReLU that follows every
// Conv2d / ConvTranspose2d of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:34-45,51-57) is folded
// into the convolution weights (scale) and this single pass (shift + ReLU), and the pass writes straight into the
// channel slice of the concatenated feature map (base_bev_backbone.py:103 torch.cat(ups, dim=1)) instead of a
// temporary + cat.  HBM-bound: reads the conv output once, writes once (3 passes + cat before).
#include "common.h"

// in: (n_pix, C) rows (NHWC), out: rows of out_C channels, this tensor occupies channels [out_off, out_off + C)
template <bool RELU>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(const float4 *__restrict__ in,
                                                            const float4 *__restrict__ bias4, long long n4, int C4,
                                                            float4 *__restrict__ out, int out_C4, int out_off4) {
    constexpr int UN = 4;
    const long long base = ((long long)blockIdx.x * UN) * 256 + threadIdx.x;
    float4 v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        const long long i = base + 256ll * k;
        if (i < n4) v[k] = in[i];
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        const long long i = base + 256ll * k;
        if (i < n4) {
            const long long pix = i / C4;
            const int c = (int)(i - pix * C4);
            const float4 b = bias4[c];
            float4 r = make_float4(v[k].x + b.x, v[k].y + b.y, v[k].z + b.z, v[k].w + b.w);
            if (RELU) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
            out[pix * out_C4 + out_off4 + c] = r;
        }
    }
}

// ConvTranspose2d with kernel == stride == s run as a plain GEMM: in row = input pixel (b, y, x), in column = (ky, kx, c).
// The epilogue adds the shift, applies ReLU and does the pixel shuffle: out[b][s*y + ky][s*x + kx][out_off + c].
template <bool RELU>
__global__ __launch_bounds__(256) void bias_act_upsample_nhwc_kernel(const float4 *__restrict__ in,
                                                                     const float4 *__restrict__ bias4, long long n4,
                                                                     int h, int w, int s, int C4,
                                                                     float4 *__restrict__ out, int out_C4, int out_off4) {
    constexpr int UN = 4;
    const long long base = ((long long)blockIdx.x * UN) * 256 + threadIdx.x;
    float4 v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        const long long i = base + 256ll * k;
        if (i < n4) v[k] = in[i];
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        const long long i = base + 256ll * k;
        if (i < n4) {
            long long r = i / C4;
            const int c = (int)(i - r * C4);
            const int kx = (int)(r % s); r /= s;
            const int ky = (int)(r % s); r /= s;
            const int x = (int)(r % w); r /= w;
            const int y = (int)(r % h);
            const long long b = r / h;
            const long long opix = (b * (h * s) + (long long)y * s + ky) * ((long long)w * s) + (long long)x * s + kx;
            const float4 bb = bias4[c];
            float4 o = make_float4(v[k].x + bb.x, v[k].y + bb.y, v[k].z + bb.z, v[k].w + bb.w);
            if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            out[opix * out_C4 + out_off4 + c] = o;
        }
    }
}

LIDAR_EXPORT int lidar_bias_act_upsample_nhwc(const float *in, const float *bias, int batch, int h, int w, int s, int C,
                                              int relu, float *out, int out_C, int out_off, void *stream) {
    if (batch < 0 || h <= 0 || w <= 0 || s <= 0 || C <= 0 || (C & 3) || (out_C & 3) || (out_off & 3) || out_off < 0 ||
        out_off + C > out_C)
        return LIDAR_ERR_ARG;
    if (batch == 0) return LIDAR_OK;
    if (!in || !bias || !out) return LIDAR_ERR_ARG;
    const long long n4 = (long long)batch * h * w * s * s * (C / 4);
    const long long blocks = (n4 + 1023) / 1024;
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (relu)
        hipLaunchKernelGGL(bias_act_upsample_nhwc_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float4 *)in, (const float4 *)bias, n4, h, w, s, C / 4, (float4 *)out, out_C / 4,
                           out_off / 4);
    else
        hipLaunchKernelGGL(bias_act_upsample_nhwc_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float4 *)in, (const float4 *)bias, n4, h, w, s, C / 4, (float4 *)out, out_C / 4,
                           out_off / 4);
    return lidar_check_launch("lidar_bias_act_upsample_nhwc");
}

LIDAR_EXPORT int lidar_bias_act_nhwc(const float *in, const float *bias, long long n_pix, int C, int relu, float *out,
                                     int out_C, int out_off, void *stream) {
    if (n_pix < 0 || C <= 0 || (C & 3) || (out_C & 3) || (out_off & 3) || out_off < 0 || out_off + C > out_C)
        return LIDAR_ERR_ARG;
    if (n_pix == 0) return LIDAR_OK;
    if (!in || !bias || !out) return LIDAR_ERR_ARG;
    const long long n4 = n_pix * (C / 4);
    const long long blocks = (n4 + 1023) / 1024;
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (relu)
        hipLaunchKernelGGL(bias_act_nhwc_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)in,
                           (const float4 *)bias, n4, C / 4, (float4 *)out, out_C / 4, out_off / 4);
    else
        hipLaunchKernelGGL(bias_act_nhwc_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)in,
                           (const float4 *)bias, n4, C / 4, (float4 *)out, out_C / 4, out_off / 4);
    return lidar_check_launch("lidar_bias_act_nhwc");
}
