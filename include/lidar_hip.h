/*
 * lidar_hip.h — C ABI of liblidar_hip.so (hand-written gfx950 / CDNA4 kernels).
 *
 * This is the drop-in boundary for the reference's per-frame hot path.  Every entry point takes
 * raw DEVICE pointers (unless a parameter says "host"), explicit sizes and a hipStream_t passed as
 * void*; nothing allocates, frees or synchronises inside (scratch comes from a caller-provided
 * workspace whose size is returned by the matching *_workspace_bytes query), so every call is
 * capturable into a hipGraph.  Return value: 0 = ok, <0 = error (LIDAR_ERR_*), never exit().
 *
 * Each declaration cites the reference interface it replaces (paths under /root/reference/).
 * The Python-side binding a maintainer adds is shown in INTEGRATION.md.
 */
#ifndef LIDAR_HIP_H
#define LIDAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIDAR_OK 0
#define LIDAR_ERR_ARG (-1)
#define LIDAR_ERR_LAUNCH (-2)
#define LIDAR_ERR_WORKSPACE (-3)
#define LIDAR_ERR_UNSUPPORTED (-4) /* an optional library-backed path is not available: keep the other path */

/* ------------------------------------------------------------------ voxelisation
 * Replaces spconv.utils.VoxelGeneratorV2.generate (external, un-vendored; call site
 * pcdet/datasets/processor/data_processor.py:48-80) for a whole batch, and the voxel part of
 * DatasetTemplate.collate_batch (pcdet/datasets/dataset.py:153-185: concatenation + batch index
 * column).  Results are identical to running the sequential scan frame by frame.
 *
 *   points        (sum N_f, C) f32, frames concatenated
 *   point_offsets (batch+1) i32 DEVICE, exclusive prefix of N_f
 *   n_max         host upper bound of max_f N_f (sizes the launch and the workspace)
 *   range6/voxel_size3/grid3  HOST arrays: [x0,y0,z0,x1,y1,z1], [vx,vy,vz], [nx,ny,nz]
 *   compact       1: frame f's rows start at sum_{g<f} V_g (the collate_batch layout)
 *                 0: frame f's rows start at f*max_voxels
 *   algo          0: auto (3 when it applies, else 2); 3 (1 is accepted as an alias): LDS-binned, 2 launches — bin + zero-fill
 *                 roles in one launch, then emit; a frame's points are hash-partitioned by pillar into n_max / 1280 bins, one
 *                 workgroup each (a bin with more than 3072 points — zero-padded clouds — is handled exactly by a streaming
 *                 variant); needs n_max <= 32768, max_points < 16384 and an x / y grid below 2^24 cells; no global atomics
 *                 on the data path; more distinct voxels (4096) / list cells (3072) in ONE hash bin than its LDS holds
 *                 (adversarial input only) sets the sticky flag read by lidar_voxelize_error_flag / mirrored to the host by
 *                 lidar_voxelize_set_error_mirror; 2: global hash table (any n_max);
 *                 4: as 3 with a RESIDENT output buffer (compact layout): the caller passes the SAME voxels / num_points
 *                 buffers call after call and does not write to them in between; the zero padding then survives from call
 *                 to call and only the slots the previous call filled are re-zeroed (~30 MB of HBM traffic per 16 KITTI
 *                 frames instead of 150 MB), results unchanged.  The workspace remembers the buffer addresses: another
 *                 buffer (or the first call) is cleared in full, so only an in-place modification by the caller between
 *                 calls can break the contract
 *   voxels        (batch*max_voxels, max_points, C) f32; rows [0, total) fully written (zero padded)
 *   coords        (batch*max_voxels, 4) i32 [b, z, y, x]
 *   num_points    (batch*max_voxels) i32
 *   voxel_offsets (batch+1) i32: first row of each frame; [batch] = total rows (compact) */
size_t lidar_voxelize_workspace_bytes(int batch, int n_max, int max_voxels);
/* call once after allocating the workspace (and again after any failed call) */
int lidar_voxelize_workspace_init(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels, void *stream);
int lidar_voxelize(const float *points, const int *point_offsets, int batch, int n_max, int num_features,
                   const float *range6, const float *voxel_size3, const int *grid3, int max_points,
                   int max_voxels, int compact, int algo, float *voxels, int *coords, int *num_points,
                   int *voxel_offsets, void *ws, size_t ws_bytes, void *stream);
/* lidar_voxelize for a caller that also knows the frame offsets on the HOST — the reference's collate_batch does
 * (pcdet/datasets/dataset.py:153-185 builds the batch index from the per-sample lengths): host_offsets = batch + 1 ints in host
 * memory with the same contents as point_offsets (nullptr = none).  Up to 64 frames they travel as kernel arguments, and the
 * LDS-binned launches (algo 3 / 4) issue their first point read without a dependent load of the offsets; results identical. */
int lidar_voxelize_hostoff(const float *points, const int *point_offsets, const int *host_offsets, int batch, int n_max,
                           int num_features, const float *range6, const float *voxel_size3, const int *grid3, int max_points,
                           int max_voxels, int compact, int algo, float *voxels, int *coords, int *num_points,
                           int *voxel_offsets, void *ws, size_t ws_bytes, void *stream);
/* Measurement only: a timer = four HIP events carried by the launches themselves (hipExtLaunchKernel's start / stop events):
 * the time from the start of the call's first kernel to the end of its last one, as a kernel trace would report it, without the
 * marker / queue overhead of a hipEventRecord bracket around the call.  lidar_voxelize_time_next arms `timer` for the calling
 * thread's next lidar_voxelize(_hostoff) call: the LDS-binned path records into it, any other path (or an argument error)
 * disarms it unrecorded.  lidar_timer_elapsed_ms waits for the stop event (< 0: nothing was recorded);
 * lidar_timer_parts_ms splits the same span into {first launch, gap between the launches, last launch}. */
void *lidar_timer_create(void);
void lidar_timer_destroy(void *timer);
void lidar_voxelize_time_next(void *timer);
float lidar_timer_elapsed_ms(void *timer);
int lidar_timer_parts_ms(void *timer, float *out3);
/* optional: a device-visible HOST int (pinned + mapped memory) that receives the same error bits, so the caller can poll
 * the flag without a copy or a synchronisation (nullptr unregisters).  Cleared by the caller. */
int lidar_voxelize_set_error_mirror(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels, int *host_flag,
                                    void *stream);
/* host-synchronous read of the sticky overflow flag of algo 3 / 4 (0 = fine); not for use inside captures */
int lidar_voxelize_error_flag(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels);

/* HOST voxel generator: spconv.utils.VoxelGeneratorV2.generate for ONE frame on a CPU core, for the reference's real call
 * site — DataProcessor.transform_points_to_voxels inside forked DataLoader workers (pcdet/datasets/processor/
 * data_processor.py:48-80, pcdet/datasets/__init__.py:73), which must not touch the GPU.  Touches no HIP state.
 *   points (n, C) f32 HOST; voxels (max_voxels, max_points, C), coords_zyx (max_voxels, 3) [z, y, x], num_points (max_voxels)
 *   caller-allocated and UNinitialised: rows [0, return value) are fully written (zero padded), the rest is left alone.
 *   scratch: lidar_voxelize_cpu_scratch_bytes(n) bytes.  Returns the voxel count (>= 0) or a negative LIDAR_ERR_*. */
size_t lidar_voxelize_cpu_scratch_bytes(int n);
int lidar_voxelize_cpu(const float *points, int n, int num_features, const float *range6, const float *voxel_size3,
                       const int *grid3, int max_points, int max_voxels, float *voxels, int *coords_zyx, int *num_points,
                       void *scratch, size_t scratch_bytes);

/* ------------------------------------------------------------------ PillarVFE (one PFN layer, eval)
 * Replaces PillarVFE.forward + PFNLayer.forward (pcdet/models/backbones_3d/vfe/pillar_vfe.py:94-123,
 * :29-49) with BatchNorm1d(eps=1e-3) in eval mode folded by the caller:
 *   scale = gamma / sqrt(var + eps), shift = beta - mean*scale  (both (cout)); no-norm: scale=1, shift=bias.
 *   weight (cout, C+6[+1]) row-major = nn.Linear.weight; feature order [point C | xyz-mean | xyz-centre | dist]
 *   coords/num_points may be the f32 tensors the reference's load_data_to_gpu produces
 *   (pcdet/models/__init__.py:22) — set *_are_float — or the i32 outputs of lidar_voxelize.
 *   num_voxels_dev optional device int (e.g. &voxel_offsets[batch]); rows >= it are skipped. */
int lidar_pillar_vfe(const float *voxels, const void *num_points, const void *coords, int num_voxels,
                     const int *num_voxels_dev, int max_points, int num_features, const float *weight,
                     const float *scale, const float *shift, int cout, const float *voxel_size3,
                     const float *range6, int with_distance, int coords_are_float, int num_are_float,
                     float *out, void *stream);

/* MeanVFE.forward (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31): out (V, C) */
int lidar_mean_vfe(const float *voxels, const void *num_points, int num_voxels, int max_points,
                   int num_features, int num_are_float, float *out, void *stream);

/* PointPillarScatter.forward (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37), nz == 1:
 * canvas (batch, channels, ny, nx) f32, every element written exactly once. channels in {32,64,128}.
 * channels_last = 1 writes the same logical tensor with NHWC strides (torch.channels_last), which MIOpen's fp32
 * convolutions consume without layout transposes. */
/* Resident-canvas variant: ONE persistent channels-last canvas (B, ny, nx, C); a call clears the cells the previous call wrote
 * (prev_cells / prev_count: caller-owned state, num_voxels ints + 1 int; start with a zero canvas and *prev_count = 0) and
 * writes the new pillars -- ~2*V*C*4 bytes instead of the whole canvas; the canvas equals a fresh PointPillarScatter output. */
int lidar_pillar_scatter_update(const float *pillar_features, const void *coords, int coords_are_float, int num_voxels,
                                const int *num_voxels_dev, int channels, int batch, int nx, int ny, float *canvas,
                                int *prev_cells, int *prev_count, void *stream);
/* The backbone's first convolution straight from the pillars (PointPillarScatter pointpillar_scatter.py:14-37 + the first
 * ZeroPad2d / Conv2d / BatchNorm / ReLU of BaseBEVBackbone, base_bev_backbone.py:34-45, fused): the k x k / stride / pad
 * convolution's neighbour table over ALL output pixels in NHWC map order, nbr[(b * OH + oy) * OW + ox][ky * k + kx] = pillar row
 * at input cell (oy * stride - pad + ky, ox * stride - pad + kx) or -1.  lidar_spconv_implicit_gemm_sorted over this table with
 * weight (k * k, Cin, Cout), the folded shift as bias and ReLU writes the layer's dense NHWC output (rows without taps get
 * act(bias)); the > 90 %-zero canvas is never built.  ws: lidar_pillar_conv_table_workspace_bytes (inverse cell -> pillar map). */
size_t lidar_pillar_conv_table_workspace_bytes(int batch, int nx, int ny);
int lidar_pillar_conv_table(const void *coords, int coords_are_float, int num_voxels, const int *num_voxels_dev, int batch, int nx,
                            int ny, int k, int stride, int pad, int *nbr, void *ws, size_t ws_bytes, void *stream);
size_t lidar_pillar_scatter_workspace_bytes(int batch, int nx, int ny);
int lidar_pillar_scatter(const float *pillar_features, const void *coords, int coords_are_float, int num_voxels,
                         const int *num_voxels_dev, int channels, int batch, int nx, int ny, int channels_last,
                         float *canvas, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ iou3d_nms
 * boxes are (N,7) f32 [x, y, z, dx, dy, dz, heading].
 * mode 0 = boxes_overlap_bev_gpu (pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:49-68, kernel.cu:236-249)
 * mode 1 = boxes_iou_bev_gpu     (iou3d_nms.cpp:70-88, kernel.cu:251-265);  out (n_a, n_b) f32 */
size_t lidar_iou_workspace_bytes(int n_a, int n_b);
int lidar_boxes_pairwise_bev(const float *boxes_a, int n_a, const float *boxes_b, int n_b, int mode, float *out,
                             void *ws, size_t ws_bytes, void *stream);

/* nms_gpu (iou3d_nms.cpp:90-136) / nms_normal_gpu (:139-186), batched, greedy reduce on the device.
 *   boxes (batch, n_max, 7) already sorted by descending score; counts (batch) i32 device or NULL
 *   keep (batch, n_max) i64 device: positions of kept boxes, ascending; num_keep (batch) i32 device */
size_t lidar_nms_workspace_bytes(int batch, int n_max);
int lidar_nms_batch(const float *boxes, const int *counts, int batch, int n_max, float thresh, int normal,
                    long long *keep, int *num_keep, void *ws, size_t ws_bytes, void *stream);
/* same, but only the first max_keep survivors of each frame are wanted (the caller truncates to NMS_POST_MAXSIZE anyway,
 * pcdet/models/model_utils/model_nms_utils.py:19-21): the greedy pass stops early; num_keep <= max_keep */
int lidar_nms_batch_limited(const float *boxes, const int *counts, int batch, int n_max, float thresh, int normal,
                            int max_keep, long long *keep, int *num_keep, void *ws, size_t ws_bytes, void *stream);
/* test hook: device pointer to the (batch, n_max, ceil(n_max/64)) u64 suppression mask inside ws */
const void *lidar_nms_mask_ptr(void *ws, int batch, int n_max);

/* ------------------------------------------------------------------ pointnet2_stack (stacked-batch layout)
 * Every *_batch_cnt is a (B) i32 DEVICE array of per-sample counts.  Index outputs are i32. */
/* ball_query_wrapper_stack (pcdet/ops/pointnet2/pointnet2_stack/src/ball_query.cpp:31-47, ball_query_gpu.cu:16-66):
 * idx (M, nsample) zero-filled by the caller; an empty ball gets idx[.,0] = -1 */
int lidar_ball_query_stack(int B, int M, float radius, int nsample, const float *new_xyz, const int *new_xyz_batch_cnt,
                           const float *xyz, const int *xyz_batch_cnt, int *idx, void *stream);
/* two radii over the same centres / candidates in one pass (the scales of one StackSAModuleMSG): each idx_x equals what
 * lidar_ball_query_stack returns for (radius_x, nsample_x); every squared distance is computed once */
int lidar_ball_query_stack2(int B, int M, float radius_a, int nsample_a, float radius_b, int nsample_b, const float *new_xyz,
                            const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt, int *idx_a, int *idx_b,
                            void *stream);
/* The same lists through a cell grid over the candidates (csrc/pointnet2.hip: one binning pass per batch element, then three cell
 * rows per centre instead of every candidate; hits are put back into index order at the end).  N = rows of xyz; idx_b == NULL or
 * radius_b <= 0: one radius; nsample <= 64.  ws: lidar_ball_query_grid_workspace_bytes(B, N), uninitialised. */
size_t lidar_ball_query_grid_workspace_bytes(int B, int N);
int lidar_ball_query_stack_grid(int B, int M, int N, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                const float *new_xyz, const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt,
                                int *idx_a, int *idx_b, void *ws, size_t ws_bytes, void *stream);
/* group_points_wrapper_stack (group_points.cpp:31-69 fwd, group_points_gpu.cu:71-102): out (M, C, nsample) */
int lidar_group_points_stack(int B, int M, int C, int nsample, const float *features, const int *features_batch_cnt,
                             const int *idx, const int *idx_batch_cnt, float *out, void *stream);
/* group_points_grad_wrapper_stack (group_points_gpu.cu:15-45): grad_features (N, C) zero-filled by the caller */
/* Row-major grouping for the inference path of StackSAModuleMSG (pointnet2_stack/pointnet2_modules.py:58-92): out (M, nsample,
 * stride) with row (m, s) = [xyz[idx] - new_xyz[m] if use_xyz | features[idx] | zeros up to stride]; idx is the RAW result of
 * lidar_ball_query_stack (-1 in column 0 = empty ball -> zero rows, as QueryAndGroup zeroes them, pointnet2_utils.py:146-152). */
int lidar_group_rows_stack(int B, int M, int C, int nsample, int use_xyz, int stride, const float *xyz, const float *new_xyz,
                           const float *features, const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                           float *out, void *stream);
/* The same layer with the first linear map moved in front of the gather (it commutes with it): table (N, H) =
 * [xyz | features] @ W + b per SOURCE point, query_term (M, H) = new_xyz @ W_xyz per query (or null), and
 * out (M, nsample, H) = relu(table[idx] - query_term[m]); an empty ball gives empty_row (H) = relu(b).  H % 4 == 0. */
int lidar_group_rows_affine_stack(int B, int M, int H, int nsample, const float *table, const float *query_term,
                                  const float *empty_row, const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                                  float *out, void *stream);
/* A two-layer scale in one kernel: the gather above, the second layer on the matrix cores (W2 (H1, H2) row-major, b2) and the max
 * over the samples — out (M, H2) = max_s relu(relu(table[idx] - query_term[m]) @ W2 + b2).  H1 in {16, 32, 64}, H2 <= 128,
 * nsample in {8, 16, 32} (lidar_sa_layer2_max_supported). */
int lidar_sa_layer2_max_supported(int H1, int H2, int nsample);
int lidar_sa_layer2_max_stack(int B, int M, int H1, int H2, int nsample, const float *table, const float *query_term,
                              const float *empty_row, const float *W2, const float *b2, const int *features_batch_cnt,
                              const int *idx, const int *idx_batch_cnt, float *out, void *stream);
int lidar_group_points_grad_stack(int B, int M, int C, int N, int nsample, const float *grad_out, const int *idx,
                                  const int *idx_batch_cnt, const int *features_batch_cnt, float *grad_features,
                                  void *stream);
/* furthest_point_sampling_wrapper (sampling.cpp, sampling_gpu.cu:24-140; same kernel in pointnet2_batch):
 * points (b, n, 3), temp (b, n) = 1e10 on entry, idx (b, m) */
int lidar_furthest_point_sampling(int b, int n, int m, const float *points, float *temp, int *idx, void *stream);
/* three_nn_wrapper_stack (interpolate_gpu.cu:16-75): dist2 (N,3) squared distances, idx (N,3) global indices */
int lidar_three_nn_stack(int B, int N, const float *unknown, const int *unknown_batch_cnt, const float *known,
                         const int *known_batch_cnt, float *dist2, int *idx, void *stream);
/* three_interpolate_wrapper_stack (:107-126) / _grad_ (:151-172): features (M, C), out (N, C) */
int lidar_three_interpolate_stack(int N, int C, const float *features, const int *idx, const float *weight, float *out,
                                  void *stream);
int lidar_three_interpolate_grad_stack(int N, int C, const float *grad_out, const int *idx, const float *weight,
                                       float *grad_features, void *stream);

/* ------------------------------------------------------------------ pointnet2_batch (dense, channel-major)
 * pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24 */
int lidar_ball_query_batch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                           int *idx, void *stream);                       /* ball_query_gpu.cu:15-51 */
int lidar_group_points_batch(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                             float *out, void *stream);                   /* group_points_gpu.cu:53-72 */
int lidar_group_points_grad_batch(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                  float *grad_points, void *stream);      /* group_points_gpu.cu:14-31 */
int lidar_gather_points_batch(int b, int c, int n, int npoints, const float *points, const int *idx, float *out,
                              void *stream);                              /* sampling_gpu.cu:15-31 */
int lidar_gather_points_grad_batch(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                   float *grad_points, void *stream);     /* sampling_gpu.cu:53-70 */
int lidar_three_nn_batch(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                         void *stream);                                   /* interpolate_gpu.cu:16-59 */
int lidar_three_interpolate_batch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                  float *out, void *stream);              /* interpolate_gpu.cu:84-104 */
int lidar_three_interpolate_grad_batch(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                       const float *weight, float *grad_points, void *stream);  /* :127-149 */

/* ------------------------------------------------------------------ roiaware_pool3d / roipoint_pool3d
 * roiaware_pool3d_gpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:29-66): argmax (R,x,y,z,C) i32,
 * pts_idx_of_voxels (R,x,y,z,max_pts) i32 (slot 0 = count) and pooled (R,x,y,z,C) f32 zero-filled by the
 * caller; pool_method 0 = max, 1 = avg; out_x/y/z < 256.  No (boxes x points) scratch matrix is needed. */
int lidar_roiaware_pool3d_forward(int boxes_num, int pts_num, int channels, int max_pts_each_voxel, int out_x, int out_y,
                                  int out_z, const float *rois, const float *pts, const float *pts_feature, int *argmax,
                                  int *pts_idx_of_voxels, float *pooled_features, int pool_method, void *stream);
/* roiaware_pool3d_gpu_backward (roiaware_pool3d.cpp:68-96): grad_in (P, C) zero-filled by the caller */
int lidar_roiaware_pool3d_backward(int boxes_num, int out_x, int out_y, int out_z, int channels, int max_pts_each_voxel,
                                   const int *pts_idx_of_voxels, const int *argmax, const float *grad_out,
                                   float *grad_in, int pool_method, void *stream);
/* points_in_boxes_gpu (roiaware_pool3d.cpp:98-118): boxes (B,T,7), pts (B,P,3), box_idx_of_points (B,P) pre-filled -1 */
int lidar_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts,
                          int *box_idx_of_points, void *stream);
/* roipool3d_gpu (pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp:23-54): xyz (B,N,3), boxes3d (B,M,7) already
 * enlarged, pts_feature (B,N,C) -> pooled (B,M,S,3+C), empty_flag (B,M); both zero-filled by the caller; S <= 1024 */
int lidar_roipoint_pool3d_forward(int batch, int pts_num, int boxes_num, int feature_len, int sampled_pts_num,
                                  const float *xyz, const float *boxes3d, const float *pts_feature,
                                  float *pooled_features, int *pooled_empty_flag, void *stream);

/* ------------------------------------------------------------------ sparse 3D convolution (spconv v1.x semantics)
 * Replaces the external, un-vendored `spconv` package (docs/INSTALL.md:9,28-29; call sites
 * pcdet/models/backbones_3d/spconv_backbone.py:3-26,76-116, spconv_unet.py, roi_heads/partA2_head.py).
 * indices are (N,4) i32 [b,z,y,x]; spatial shape (D,H,W); kernel offsets enumerate (kz,ky,kx) row-major.
 * Rulebooks are neighbour tables nbr (N_out, K) i32: nbr[j][k] = input row feeding output row j through
 * offset k, or -1 (spconv.ops.get_indice_pairs gives the same relation as per-offset pair lists). */
size_t lidar_spconv_hash_capacity(int n);            /* slots; table memory = capacity * 12 bytes */
int lidar_spconv_build_hash(const int *indices, int n, int D, int H, int W, void *table, size_t capacity, void *stream);
/* SubMConv3d rulebook: outputs == inputs (same row order) */
int lidar_spconv_subm_table(const int *indices, int n, int D, int H, int W, int kD, int kH, int kW, const void *table,
                            size_t capacity, int *nbr, void *stream);
/* SparseConv3d rulebook in two phases (the caller reads *num_out in between to size nbr):
 *   phase 1 -> out_indices (out_cap >= n*prod ceil(k/s) rows, 4), *num_out; output rows in first-touch order
 *   phase 2 -> nbr (num_out, K) and nbr_t (n, K) = transposed table (inverse conv / input gradient) */
size_t lidar_spconv_conv_table_workspace_bytes(int n, int kD, int kH, int kW, int sD, int sH, int sW);
int lidar_spconv_conv_outputs(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW, int sD,
                              int sH, int sW, int pD, int pH, int pW, int *out_indices, int out_cap, int *num_out,
                              void *ws, size_t ws_bytes, void *stream);
int lidar_spconv_conv_tables(int n, int kD, int kH, int kW, int sD, int sH, int sW, int num_out, int *nbr, int *nbr_t,
                             void *ws, size_t ws_bytes, void *stream);
/* The same rulebooks through DENSE index grids (csrc/rulebook_grid.hip): one persistent (batch, D, H, W) int32 grid per resolution
 * level, owned by the caller, every cell "empty" (0x7FFFFFFF) between uses; a coordinate lookup is one load instead of a hash probe
 * sequence, and nothing is built or cleared per table.  Tables and output numbering are identical to the hash builder's.
 *   lidar_spconv_grid_init     once per grid
 *   lidar_spconv_grid_rows     mode 0: rows of a tensor -> grid (atomicMin: duplicates keep the lowest row); 1: the same cells
 *                              back to empty; 2: plain store (output rows after lidar_spconv_grid_outputs); n_dev optional
 *   lidar_spconv_grid_table    nbr (n_out, K): SubM (stride 1, padding k / 2, out_indices = the input rows) or regular forward
 *   lidar_spconv_grid_table_t  nbr_t (n, K) of a regular convolution, from the OUTPUT level's grid
 *   lidar_spconv_grid_outputs  unique output sites in first-touch order -> out_indices, *num_out (device); grid_out (empty on
 *                              entry) is left holding candidate ids: follow with lidar_spconv_grid_rows(out_indices, mode 2)
 * Capacity-sized tensors (no host read-back of *num_out): a row whose batch index is negative is a PADDING row — no neighbours
 * (tables hold -1), reaches no output, never scattered.  lidar_spconv_grid_pad_rows turns rows [min(*num_dev, cap), cap) of
 * out_indices into padding rows; `limit` of the table builders = row count of the tensor the looked-up grid holds (grid values
 * >= limit read as "no row"; <= 0: no limit), so a capacity that turns out too small yields wrong tables, never wild row ids.
 * `batch` = frames the grids were allocated for: a row whose batch index lies outside [0, batch) is treated like a padding row
 * (skipped), never an out-of-bounds access.
 *   lidar_spconv_transpose_table  nbr_t (n_in, K), pre-filled with -1 by the caller, from nbr (n_out, K) alone:
 *                              nbr[j][k] == i  <=>  nbr_t[i][k] == j (equals grid_table_t when input coordinates are unique) */
int lidar_spconv_grid_init(int *grid, size_t cells, void *stream);
int lidar_spconv_grid_rows(const int *indices, int n, const int *n_dev, int batch, int D, int H, int W, int *grid, int mode, void *stream);
int lidar_spconv_grid_table(const int *out_indices, int n_out, int batch, int D, int H, int W, int kD, int kH, int kW, int sD, int sH, int sW,
                            int pD, int pH, int pW, const int *grid_in, int limit, int *nbr, void *stream);
int lidar_spconv_grid_table_t(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW, int sD, int sH, int sW, int pD,
                              int pH, int pW, const int *grid_out, int limit, int *nbr_t, void *stream);
int lidar_spconv_grid_pad_rows(int *out_indices, const int *num_dev, int cap, void *stream);
int lidar_spconv_transpose_table(const int *nbr, int n_out, int K, int n_in, int *nbr_t, void *stream);
size_t lidar_spconv_grid_outputs_workspace_bytes(int n, int K);
int lidar_spconv_grid_outputs(const int *indices, int n, int batch, int D, int H, int W, int kD, int kH, int kW, int sD, int sH, int sW, int pD,
                              int pH, int pW, int *grid_out, int *out_indices, int *num_out, void *ws, size_t ws_bytes,
                              void *stream);
/* indice_conv forward (and input gradient with the transposed table + transposed weights):
 * out (n_out, Cout) = sum_k in[nbr[., k]] @ weight[k] (+ bias); weight (K, Cin, Cout); fp32 MFMA; Cin, Cout <= 128 */
int lidar_spconv_implicit_gemm(const float *in_features, const int *nbr, int n_out, int K, int Cin, int Cout,
                               const float *weight, const float *bias, float *out_features, void *stream);
/* the same GEMM with the inference epilogue of the reference's SubMConv3d/SparseConv3d + BatchNorm1d + ReLU triplets
 * (pcdet/models/backbones_3d/spconv_backbone.py:20-26; BN scale folded into `weight`, shift passed as `bias`) and of
 * SparseBasicBlock (spconv_backbone.py:49-63): out = act(gemm + bias + residual); residual (n_out, Cout) or NULL */
int lidar_spconv_implicit_gemm_fused(const float *in_features, const int *nbr, int n_out, int K, int Cin, int Cout,
                                     const float *weight, const float *bias, const float *residual, int relu,
                                     float *out_features, void *stream);
/* Mask-sorted execution of the same GEMM: on LiDAR occupancy a site has 3-13 of its 27 neighbours, so in table order most
 * MFMA rows of a 32-row tile are padding.  lidar_spconv_row_masks gives each row's offset bit mask (K <= 32); the caller
 * argsorts the masks (any device sort) and passes the order: workgroup row i computes table / output row order[i], so
 * rows with equal masks share MFMA tiles and offsets unused by a whole workgroup are skipped entirely.  Every output
 * still sums the same products in the same (offset) order: results are bit-identical to the table-order call. */
int lidar_spconv_row_masks(const int *nbr, int n_out, int K, int *masks, void *stream);
/* Row order for lidar_spconv_implicit_gemm_sorted WITHOUT a sort: every row finds its mask's group in an open-addressing table
 * (a table holds only a few thousand distinct masks) and takes a rank in it; groups are laid out by their top 12 mask bits (measured:
 * as good for the GEMM as a full sort), a row's position = first position of its group + its rank.  masks (n_out): bit k =
 * nbr[row][k] >= 0; order (n_out): table row visited i-th — equal masks contiguous, groups ascending in their top 12 bits.
 * 3 launches, no host sync.  The workspace keeps the table, which must be empty when a call starts (lidar_spconv_mask_group_init
 * once per buffer; every call leaves it empty) and whose layout depends on the buffer size only: always pass the same (ws,
 * ws_bytes) pair, sized for the largest table, and do not share one workspace between streams.  K <= 31. */
size_t lidar_spconv_mask_group_workspace_bytes(int n_out);
int lidar_spconv_mask_group_init(void *ws, size_t ws_bytes, void *stream);
int lidar_spconv_mask_group(const int *nbr, int n_out, int K, int *masks, int *order, void *ws, size_t ws_bytes, void *stream);
int lidar_spconv_sorted_gemm_supported(int K, int Cin, int Cout);
int lidar_spconv_implicit_gemm_sorted(const float *in_features, const int *nbr, const int *row_mask, const int *order,
                                      int n_out, int K, int Cin, int Cout, const float *weight, const float *bias,
                                      const float *residual, int relu, float *out_features, void *stream);
/* The same GEMM reading the weights in PACKED form: lidar_spconv_pack_weights lays the folded (K, Cin, Cout) weights out in the
 * order the MFMA lanes consume them (once per weight update; lidar_spconv_packed_floats floats, 0 = shape not supported:
 * Cin in {16, 32, 64, 128}, Cout % 4 == 0, K <= 32), so that a stage goes global -> LDS by LDS-DMA and the B operands of four
 * MFMA steps are one 128-bit LDS read.  Same products in the same order: bit-identical to lidar_spconv_implicit_gemm_sorted. */
size_t lidar_spconv_packed_floats(int K, int Cin, int Cout);
int lidar_spconv_pack_weights(const float *weight, int K, int Cin, int Cout, float *packed, void *stream);
int lidar_spconv_implicit_gemm_sorted_packed(const float *in_features, const int *nbr, const int *row_mask, const int *order,
                                             int n_out, int K, int Cin, int Cout, const float *packed, const float *bias,
                                             const float *residual, int relu, float *out_features, void *stream);
/* weight gradient: grad_weight (K, Cin, Cout) += sum_j in[nbr[j][k]]^T (x) grad_out[j]; zero-filled by the caller */
int lidar_spconv_wgrad(const float *in_features, const float *grad_out, const int *nbr, int n_out, int K, int Cin, int Cout,
                       float *grad_weight, void *stream);
/* weight gradient on the matrix cores: same contraction as lidar_spconv_wgrad, grad_weight overwritten (no zero fill),
 * deterministic (per-chunk partials in `ws`, summed in order); order = mask order of nbr's rows (lidar_spconv_row_masks +
 * argsort) or NULL; Cin, Cout in {16, 32, 64, 128} (lidar_spconv_wgrad_mfma_supported). */
int lidar_spconv_wgrad_mfma_supported(int K, int Cin, int Cout);
size_t lidar_spconv_wgrad_workspace_bytes(int n_out, int K, int Cin, int Cout);
int lidar_spconv_wgrad_mfma(const float *in_features, const float *grad_out, const int *nbr, const int *order, int n_out, int K,
                            int Cin, int Cout, float *grad_weight, void *ws, size_t ws_bytes, void *stream);
/* SparseConvTensor.dense(): (N, C) rows at (N,4) indices -> (B, C, D, H, W), every element written once */
size_t lidar_sparse_to_dense_workspace_bytes(int batch, int D, int H, int W);
int lidar_sparse_to_dense(const float *features, const int *indices, int n, int channels, int batch, int D, int H, int W,
                          float *out, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ dense BEV backbone epilogue
 * Eval-mode BatchNorm2d + ReLU after every Conv2d / ConvTranspose2d of BaseBEVBackbone
 * (pcdet/models/backbones_2d/base_bev_backbone.py:34-45,51-57) with the BN scale folded into the weights: one pass
 * out[pix][out_off + c] = act(in[pix][c] + bias[c]) over NHWC rows; out may alias in (out_C == C, out_off == 0) or be
 * the channel slice of the concatenated map (base_bev_backbone.py:103).  C, out_C, out_off multiples of 4. */
int lidar_bias_act_nhwc(const float *in, const float *bias, long long n_pix, int C, int relu, float *out, int out_C,
                        int out_off, void *stream);
/* Same epilogue for a ConvTranspose2d with kernel == stride == s (base_bev_backbone.py:51-55) evaluated as a GEMM:
 * in (batch*h*w, s*s*C) with column order (ky, kx, c); out[b][s*y+ky][s*x+kx][out_off + c] = act(in + bias[c]). */
int lidar_bias_act_upsample_nhwc(const float *in, const float *bias, int batch, int h, int w, int s, int C, int relu,
                                 float *out, int out_C, int out_off, void *stream);
/* A ConvTranspose2d / Conv2d with kernel == stride == 1 (the first deblock, base_bev_backbone.py:58-77) + folded BatchNorm +
 * ReLU as ONE library GEMM (hipBLASLt, RELU_BIAS epilogue) writing with a leading dimension:
 * D[m][0..N) = act(A (M, K) @ W (K, N) + bias (N)), row m of D at D + m * ldd — i.e. straight into the layer's channel slice
 * of the concatenated NHWC map (base_bev_backbone.py:103).  ws: scratch for the library (may be null / 0).  Returns
 * LIDAR_ERR_UNSUPPORTED when the library is not loadable or has no kernel for the problem. */
int lidar_dense_gemm_bias_act(const float *A, long long M, int K, const float *W, int N, const float *bias, int relu,
                              float *D, int ldd, void *ws, size_t ws_bytes, void *stream);
/* The library offers several kernels per shape and the wrapper keeps the one that timed fastest at the first call.  For N ranks
 * to run the SAME kernel (the reference's DDP ranks all run cuDNN's deterministic heuristic pick: pcdet/utils/common_utils.py:170-184
 * launches identical processes), rank 0's picks are exported — 7 ints per shape — and imported by the others before their first
 * call.  export returns the number of plans (writes at most cap). */
int lidar_dense_gemm_export_choices(int *out7, int cap);
int lidar_dense_gemm_import_choices(const int *in7, int n);

/* 3x3 / stride 1 / padding 1 fp32 convolution + folded BatchNorm shift + ReLU on NHWC maps — the LAYER_NUMS stride-1 layers of
 * every BaseBEVBackbone block (pcdet/models/backbones_2d/base_bev_backbone.py:34-45) — as Winograd F(2x2, 3x3) on the fp32 matrix
 * cores (csrc/wino_conv.hip): 2.25x fewer MFMA cycles than the direct form, |error| ~ 1e-6 of the output scale.
 * lidar_wino_pack_weights: w (Cout, Cin, 3, 3) contiguous (BatchNorm scale already folded in) -> `packed`, lidar_wino_packed_floats
 * (Cin, Cout) floats (0: shape not supported — Cin % 8 == 0, Cin >= 16 and Cout % 32 == 0 are); once per weight update.
 * lidar_wino_conv3x3_nhwc: out[b][y][x][out_off + co] = act(conv(in, w)[b][y][x][co] + bias[co]), `in` (B, H, W, Cin), `out`
 * (B, H, W, out_C) — out_off / out_C as in lidar_bias_act_nhwc (a slice of the concatenated map); bias may be null. */
size_t lidar_wino_packed_floats(int Cin, int Cout);
int lidar_wino_supported(int Cin, int Cout);
int lidar_wino_pack_weights(const float *w, int Cin, int Cout, float *packed, void *stream);
int lidar_wino_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, const float *packed, const float *bias, int relu, int Cout,
                            float *out, int out_C, int out_off, void *stream);
/* Grouped form — n_groups independent 3x3 convolutions in one launch: group g reads input channels [g * group_cin, (g + 1) *
 * group_cin) of the (B, H, W, in_C) map and writes output channels [32 g, 32 g + 32) (fewer real outputs: zero-padded filters) — the
 * second-layer branch convolutions of AnchorHeadMulti's SEPARATE_MULTIHEAD heads (pcdet/models/dense_heads/anchor_head_multi.py:60-110).
 * packed = lidar_wino_pack_weights of the stacked (32 n_groups, group_cin, 3, 3) filters; bias: 32 n_groups values or null. */
int lidar_wino_conv3x3_grouped_nhwc(const float *in, int B, int H, int W, int in_C, int group_cin, int n_groups, const float *packed,
                                    const float *bias, int relu, float *out, int out_C, int out_off, void *stream);
/* ... with a COMPACT output: group g keeps its grp_cout[g] (<= 32) real channels at [out_off + grp_ooff[g], + grp_cout[g]) of the
 * (B, H, W, out_C) map (device int arrays; ranges must fit and not overlap); the padded channels never reach memory. */
int lidar_wino_conv3x3_grouped_compact_nhwc(const float *in, int B, int H, int W, int in_C, int group_cin, int n_groups,
                                            const float *packed, const float *bias, int relu, const int *grp_cout, const int *grp_ooff,
                                            float *out, int out_C, int out_off, void *stream);

/* The same stride-1 3x3 layers as Winograd F(4x4, 3x3) (csrc/wino43_conv.hip): 4x fewer MFMA cycles than the direct form (F(2x2):
 * 2.25x), interpolation points {0, 1, -1, 1/2, -2, inf}, filter transform in fp64; |error| ~ 1e-5 of the output scale (asserted at
 * 1e-4 against the fp64 convolution).  Supported: Cin % 16 == 0, Cin >= 32, Cout % 64 == 0 (lidar_wino43_packed_floats = 36 Cin Cout,
 * 0 = unsupported).  `in` is (B, H, W, in_C) and the layer reads channels [0, Cin); the input and output maps stay below 2^31 bytes each.  Replaces the same reference
 * layers as lidar_wino_conv3x3_nhwc (pcdet/models/backbones_2d/base_bev_backbone.py:34-45). */
size_t lidar_wino43_packed_floats(int Cin, int Cout);
int lidar_wino43_supported(int Cin, int Cout);
int lidar_wino43_pack_weights(const float *w, int Cin, int Cout, float *packed, void *stream);
int lidar_wino43_conv3x3_nhwc(const float *in, int B, int H, int W, int Cin, int in_C, const float *packed, const float *bias, int relu,
                              int Cout, float *out, int out_C, int out_off, void *stream);

/* ConvTranspose2d with kernel == stride == s (the deblocks of BaseBEVBackbone, base_bev_backbone.py:51-57) + folded BatchNorm shift
 * + ReLU + the write into the layer's channel slice of the concatenated map (base_bev_backbone.py:103) as ONE fp32-MFMA kernel
 * (csrc/deconv_gemm.hip): out[b][s y + ky][s x + kx][out_off + c] = act(sum_k in[b][y][x][k] W[k][(ky, kx, c)] + bias[c]).
 * W: (K, s * s * C_up) row-major, columns ordered (ky, kx, c); packed once per weight update (lidar_deconv_pack_weights,
 * lidar_deconv_packed_floats floats; 0 = unsupported: K % 8 == 0, K >= 16, C_up % 128 == 0 and s * s * C_up % 512 == 0 are supported). */
size_t lidar_deconv_packed_floats(int K, int N);
int lidar_deconv_supported(int K, int s, int C_up);
int lidar_deconv_pack_weights(const float *W, int K, int N, float *packed, void *stream);
int lidar_deconv_gemm_nhwc(const float *in, int B, int h, int w, int K, const float *packed, const float *bias, int relu, int s, int C_up,
                           float *out, int out_C, int out_off, void *stream);

/* ------------------------------------------------------------------ anchor-head post-processing feeding NMS (8f rank 1)
 * head: (n_loc = B*H*W, row_stride) rows of the merged head output [cls | box | dir] as the 1x1 heads emit it
 * (pcdet/models/dense_heads/anchor_head_single.py:45-55).  Anchor index = loc * anchors_per_loc + a, class logit
 * channel = cls_off + a * num_class + c (the view(batch, num_anchors, -1) of anchor_head_template.py:247-250).
 * lidar_anchor_scores: scores[i] = max_c sigmoid(logit) if >= score_thresh else -1, labels[i] = argmax_c
 *   (detector3d_template.py:205-230 + model_nms_utils.py:6-10).
 * lidar_decode_topk: boxes (batch, k, 7) = ResidualCoder.decode_torch (box_coder_utils.py:45-77) + direction-bin
 *   correction (anchor_head_template.py:253-266) of the anchors picked by top_idx (batch, k) int64 (per-frame anchor ids);
 *   anchors (n_anchor_per_frame, 7). */
int lidar_anchor_scores(const float *head, long long n_loc, int row_stride, int cls_off, int anchors_per_loc, int num_class,
                        float score_thresh, float *scores, unsigned char *labels, void *stream);
int lidar_decode_topk(const float *head, int batch, long long locs_per_frame, int row_stride, int box_off, int dir_off,
                      int anchors_per_loc, int num_dir_bins, const long long *top_idx, int k, const float *anchors,
                      float dir_offset, float dir_limit_offset, float period, float *boxes, void *stream);
/* Exact top-k of the masked scores that feed NMS (reference: torch.topk(box_scores, k = min(NMS_PRE_MAXSIZE, n)) in
 * class_agnostic_nms, pcdet/models/model_utils/model_nms_utils.py:9-11) for a whole batch, deterministic — descending score, ties by
 * ascending anchor index (torch.topk leaves ties unspecified) — in 3 launches, no host synchronisation (csrc/topk.hip).
 *   lidar_topk_workspace_bytes / _init   workspace for (batch, n) scores; init once per buffer (histograms start at zero)
 *   lidar_anchor_scores_hist             = lidar_anchor_scores on (batch, locs_per_frame, row_stride) head rows, and counts every score
 *                                        >= score_thresh into the workspace's per-frame histogram (launch 1 of the 3)
 *   lidar_topk_desc                      scores (batch, n) f32, n % 4 == 0, rows 16-B aligned; k <= 4096; candidates = scores >=
 *                                        valid_min (> 0); hist_ready: the histogram was filled by lidar_anchor_scores_hist with
 *                                        score_thresh == valid_min (scores <= score_max); else one more pass builds it.
 *                                        -> top_scores (batch, k), top_idx (batch, k) i64, counts (batch) i32 = candidates kept;
 *                                        slots past counts[b] hold (-1, 0) */
size_t lidar_topk_workspace_bytes(int batch, long long n);
int lidar_topk_workspace_init(void *ws, size_t ws_bytes, int batch, long long n, void *stream);
int lidar_anchor_scores_hist(const float *head, int batch, long long locs_per_frame, int row_stride, int cls_off, int anchors_per_loc,
                             int num_class, float score_thresh, float *scores, unsigned char *labels, void *ws, size_t ws_bytes,
                             void *stream);
int lidar_topk_desc(const float *scores, int batch, long long n, int k, float valid_min, float score_max, int hist_ready,
                    float *top_scores, long long *top_idx, int *counts, void *ws, size_t ws_bytes, void *stream);
/* Everything between the NMS keep lists and the detector's output, batched, one launch (reference, per sample:
 * selected = keep[:NMS_POST_MAXSIZE]; final boxes / scores / labels by index chains — pcdet/models/model_utils/
 * model_nms_utils.py:19-25, pcdet/models/detectors/detector3d_template.py:236-262).  boxes (batch, k, 7), top_scores (batch, k),
 * top_idx (batch, k) i64 anchor ids, labels (batch, n) u8 class ids, keep (batch, keep_stride) i64 candidate positions, num_keep
 * (batch) -> out_boxes (batch, post, 7), out_scores (batch, post), out_labels (batch, post) i64 = class + 1, out_num (batch) =
 * min(num_keep, post); slots past out_num repeat candidate 0. */
int lidar_post_nms_gather(const float *boxes, const float *top_scores, const long long *top_idx, const unsigned char *labels,
                          const long long *keep, const int *num_keep, int batch, int k, long long n, int keep_stride, int post,
                          float *out_boxes, float *out_scores, long long *out_labels, int *out_num, void *stream);

/* HeightCompression in one pass (pcdet/models/backbones_2d/map_to_bev/height_compression.py:21-24): the (N, C*D, H, W) BEV
 * map of a sparse tensor written directly channels-last: out[b][h][w][c*D + d]; D <= 4, channels % 4 == 0; same workspace
 * as lidar_sparse_to_dense. */
int lidar_sparse_to_bev_nhwc(const float *features, const int *indices, int n, int channels, int batch, int D, int H, int W,
                             float *out, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ KITTI-eval rotated BEV IoU (8f rank 4)
 * rotate_iou_gpu_eval (pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py:290-330; numba.cuda in the reference):
 * boxes (n, 5) / query_boxes (k, 5) as (x, y, w, l, angle) -> iou (n, k); criterion -1: IoU, 0: inter / area(query),
 * 1: inter / area(box), 2: intersection area. */
int lidar_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion, float *iou, void *stream);

/* ------------------------------------------------------------------ CPU entry points (HOST pointers, no GPU touched)
 * Called by the reference from DataLoader workers (augmentation / database creation). */
/* boxes_iou_bev_cpu (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252): out (n_a, n_b) rotated BEV IoU */
int lidar_boxes_iou_bev_cpu(const float *boxes_a, int n_a, const float *boxes_b, int n_b, float *out);
/* points_in_boxes_cpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:143-168): out (n_boxes, n_pts) 0/1, margin 1e-2 */
int lidar_points_in_boxes_cpu(const float *boxes, int n_boxes, const float *pts, int n_pts, int *out);

#ifdef __cplusplus
}
#endif
#endif /* LIDAR_HIP_H */
