// Anchor-head post-processing feeding NMS, batched and sync-free, reading the merged head output
// (B, H*W, C_head) in place (no per-head reshape copies):
//   * class scores: sigmoid + max over classes + SCORE_THRESH mask — post_processing
//     (pcdet/models/detectors/detector3d_template.py:205-230) and class_agnostic_nms' score mask
//     (pcdet/models/model_utils/model_nms_utils.py:6-10);
//   * box decode of the top-k survivors only: ResidualCoder.decode_torch (pcdet/utils/box_coder_utils.py:45-77) + the
//     direction-bin correction of generate_predicted_boxes (pcdet/models/dense_heads/anchor_head_template.py:253-266,
//     limit_period pcdet/utils/common_utils.py:52-55).  Decode is per anchor, so decoding after the top-k selection gives
//     the same boxes as decoding all 321 408 anchors first.
// Every fp32 operation is kept in the order the reference's torch ops evaluate it (no FMA contraction; the division by
// the period is the multiply by its fp32 reciprocal that torch's tensor/scalar division performs).
#include "common.h"

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void anchor_scores_kernel(const float *__restrict__ head, long long n_anchor, int row_stride,
                                                            int cls_off, int A, int ncls, float thresh,
                                                            float *__restrict__ scores, unsigned char *__restrict__ labels) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_anchor) return;
    const long long loc = i / A;
    const int a = (int)(i - loc * A);
    const float *p = head + loc * row_stride + cls_off + a * ncls;
    float best = sigmoid_ref(p[0]);
    int bl = 0;
    for (int c = 1; c < ncls; ++c) {
        const float s = sigmoid_ref(p[c]);
        if (s > best) { best = s; bl = c; }       // first maximum wins, as torch.max
    }
    scores[i] = (best >= thresh) ? best : -1.0f;
    labels[i] = (unsigned char)bl;
}

__global__ __launch_bounds__(256) void decode_topk_kernel(const float *__restrict__ head, int batch, long long locs_per_frame,
                                                          int row_stride, int box_off, int dir_off, int A, int nbins,
                                                          const long long *__restrict__ top_idx, int k,
                                                          const float *__restrict__ anchors, float dir_offset,
                                                          float dir_limit_offset, float period, float *__restrict__ boxes) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= (long long)batch * k) return;
    const int b = (int)(j / k);
    const long long idx = top_idx[j];
    const long long loc = idx / A;
    const int a = (int)(idx - loc * A);
    const float *row = head + ((long long)b * locs_per_frame + loc) * row_stride;
    const float *t = row + box_off + a * 7;
    const float *an = anchors + idx * 7;
    const float xa = an[0], ya = an[1], za = an[2], dxa = an[3], dya = an[4], dza = an[5], ra = an[6];
    const float diagonal = sqrtf(dxa * dxa + dya * dya);
    float *o = boxes + j * 7;
    o[0] = t[0] * diagonal + xa;
    o[1] = t[1] * diagonal + ya;
    o[2] = t[2] * dza + za;
    o[3] = expf(t[3]) * dxa;
    o[4] = expf(t[4]) * dya;
    o[5] = expf(t[5]) * dza;
    float rg = t[6] + ra;
    if (nbins > 0) {
        const float *d = row + dir_off + a * nbins;
        int lab = 0;
        float bestd = d[0];
        for (int c = 1; c < nbins; ++c)
            if (d[c] > bestd) { bestd = d[c]; lab = c; }
        const float inv_period = 1.0f / period;                 // torch: tensor / python scalar = tensor * (1 / scalar)
        const float val = rg - dir_offset;
        const float fl = floorf(val * inv_period + dir_limit_offset);
        const float dir_rot = val - fl * period;
        rg = dir_rot + dir_offset + period * (float)lab;
    }
    o[6] = rg;
}

LIDAR_EXPORT int lidar_anchor_scores(const float *head, long long n_loc, int row_stride, int cls_off, int anchors_per_loc,
                                     int num_class, float score_thresh, float *scores, unsigned char *labels, void *stream) {
    if (n_loc < 0 || anchors_per_loc <= 0 || num_class <= 0 || cls_off < 0 ||
        cls_off + anchors_per_loc * num_class > row_stride)
        return LIDAR_ERR_ARG;
    if (n_loc == 0) return LIDAR_OK;
    if (!head || !scores || !labels) return LIDAR_ERR_ARG;
    const long long n = n_loc * anchors_per_loc;
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(anchor_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, head, n, row_stride, cls_off,
                       anchors_per_loc, num_class, score_thresh, scores, labels);
    return lidar_check_launch("lidar_anchor_scores");
}

LIDAR_EXPORT int lidar_decode_topk(const float *head, int batch, long long locs_per_frame, int row_stride, int box_off,
                                   int dir_off, int anchors_per_loc, int num_dir_bins, const long long *top_idx, int k,
                                   const float *anchors, float dir_offset, float dir_limit_offset, float period, float *boxes,
                                   void *stream) {
    if (batch < 0 || k < 0 || locs_per_frame < 0 || anchors_per_loc <= 0 || box_off < 0 ||
        box_off + anchors_per_loc * 7 > row_stride || num_dir_bins < 0 ||
        (num_dir_bins > 0 && (dir_off < 0 || dir_off + anchors_per_loc * num_dir_bins > row_stride)))
        return LIDAR_ERR_ARG;
    if (batch == 0 || k == 0) return LIDAR_OK;
    if (!head || !top_idx || !anchors || !boxes) return LIDAR_ERR_ARG;
    const long long n = (long long)batch * k;
    hipLaunchKernelGGL(decode_topk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head, batch,
                       locs_per_frame, row_stride, box_off, dir_off, anchors_per_loc, num_dir_bins, top_idx, k, anchors, dir_offset,
                       dir_limit_offset, period, boxes);
    return lidar_check_launch("lidar_decode_topk");
}

// Everything between the NMS keep lists and the detector's output in ONE launch (the reference does it per sample with index
// chains: selected = keep[:NMS_POST_MAXSIZE]; final_boxes = box_preds[selected]; final_scores, final_labels likewise —
// model_nms_utils.py:19-25, detector3d_template.py:236-262): out_boxes (B, post, 7), out_scores (B, post), out_labels (B, post)
// int64 = class + 1, out_num (B) = min(num_keep, post).  Slots past out_num repeat candidate 0 of the frame, as the batched torch
// gathers this replaces did (callers read only the first out_num).
__global__ __launch_bounds__(256) void post_nms_gather_kernel(const float *__restrict__ boxes, const float *__restrict__ top_scores,
                                                              const long long *__restrict__ top_idx, const unsigned char *__restrict__ labels,
                                                              const long long *__restrict__ keep, const int *__restrict__ num_keep,
                                                              int batch, int k, long long n, int keep_stride, int post,
                                                              float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                              long long *__restrict__ out_labels, int *__restrict__ out_num) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= batch * post) return;
    const int b = j / post, s = j - b * post;
    const int nk = min(num_keep[b], post);
    if (s == 0) out_num[b] = nk;
    long long c = 0;
    if (s < nk) c = min(max(keep[(size_t)b * keep_stride + s], 0ll), (long long)k - 1);
    const float *src = boxes + ((size_t)b * k + c) * 7;
    float *dst = out_boxes + (size_t)j * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) dst[q] = src[q];
    out_scores[j] = top_scores[(size_t)b * k + c];
    const long long a = min(max(top_idx[(size_t)b * k + c], 0ll), n - 1);
    out_labels[j] = (long long)labels[(size_t)b * n + a] + 1;
}

LIDAR_EXPORT int lidar_post_nms_gather(const float *boxes, const float *top_scores, const long long *top_idx, const unsigned char *labels,
                                       const long long *keep, const int *num_keep, int batch, int k, long long n, int keep_stride,
                                       int post, float *out_boxes, float *out_scores, long long *out_labels, int *out_num, void *stream) {
    if (batch < 0 || k <= 0 || n <= 0 || post <= 0 || keep_stride < post) return LIDAR_ERR_ARG;
    if (batch == 0) return LIDAR_OK;
    if (!boxes || !top_scores || !top_idx || !labels || !keep || !num_keep || !out_boxes || !out_scores || !out_labels || !out_num)
        return LIDAR_ERR_ARG;
    hipLaunchKernelGGL(post_nms_gather_kernel, dim3((unsigned)((batch * post + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                       top_scores, top_idx, labels, keep, num_keep, batch, k, n, keep_stride, post, out_boxes, out_scores, out_labels,
                       out_num);
    return lidar_check_launch("lidar_post_nms_gather");
}
