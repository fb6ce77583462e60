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
 *   algo          0: auto; 1: LDS-binned hashing (n_max <= 32768; no global atomics on the critical
 *                 path; a hash-bin overflow — only reachable with adversarial inputs — sets the sticky
 *                 flag read by lidar_voxelize_error_flag); 2: global hash table (any n_max)
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
/* host-synchronous read of the sticky overflow flag of algo 1 (0 = fine); not for use inside captures */
int lidar_voxelize_error_flag(void *ws, size_t ws_bytes, int batch, int n_max, int max_voxels);

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
 * canvas (batch, channels, ny, nx) f32, every element written exactly once. channels in {32,64,128}. */
size_t lidar_pillar_scatter_workspace_bytes(int batch, int nx, int ny);
int lidar_pillar_scatter(const float *pillar_features, const void *coords, int coords_are_float, int num_voxels,
                         const int *num_voxels_dev, int channels, int batch, int nx, int ny, float *canvas,
                         void *ws, size_t ws_bytes, void *stream);

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
/* test hook: device pointer to the (batch, n_max, ceil(n_max/64)) u64 suppression mask inside ws */
const void *lidar_nms_mask_ptr(void *ws, int batch, int n_max);

#ifdef __cplusplus
}
#endif
#endif /* LIDAR_HIP_H */
