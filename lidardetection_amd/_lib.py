"""ctypes binding of csrc/liblidar_hip.so — the only way the Python host side reaches the kernels.

Deliberately no fallback: if the shared library is missing or a symbol is absent we raise, so a
silent eager/PyTorch path can never stand in for the HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LIDAR_HIP_SO") or os.path.join(_HERE, "csrc", "liblidar_hip.so")   # env override: A/B builds (tools/)

vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must cover every function declared in include/lidar_hip.h
SIGNATURES = {
    "lidar_voxelize_workspace_bytes": (sz, [i32, i32, i32]),
    "lidar_voxelize_workspace_init": (i32, [vp, sz, i32, i32, i32, vp]),
    "lidar_voxelize": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "lidar_voxelize_hostoff": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "lidar_voxelize_error_flag": (i32, [vp, sz, i32, i32, i32]),
    "lidar_timer_create": (vp, []),
    "lidar_timer_destroy": (None, [vp]),
    "lidar_voxelize_time_next": (None, [vp]),
    "lidar_timer_elapsed_ms": (C.c_float, [vp]),
    "lidar_timer_parts_ms": (i32, [vp, vp]),
    "lidar_voxelize_set_error_mirror": (i32, [vp, sz, i32, i32, i32, vp, vp]),
    "lidar_pillar_vfe": (i32, [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp]),
    "lidar_mean_vfe": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "lidar_pillar_scatter_workspace_bytes": (sz, [i32, i32, i32]),
    "lidar_pillar_scatter": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "lidar_pillar_scatter_update": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_pillar_conv_table_workspace_bytes": (sz, [i32, i32, i32]),
    "lidar_pillar_conv_table": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "lidar_iou_workspace_bytes": (sz, [i32, i32]),
    "lidar_boxes_pairwise_bev": (i32, [vp, i32, vp, i32, i32, vp, vp, sz, vp]),
    "lidar_nms_workspace_bytes": (sz, [i32, i32]),
    "lidar_nms_batch": (i32, [vp, vp, i32, i32, f32, i32, vp, vp, vp, sz, vp]),
    "lidar_nms_batch_limited": (i32, [vp, vp, i32, i32, f32, i32, i32, vp, vp, vp, sz, vp]),
    "lidar_nms_mask_ptr": (vp, [vp, i32, i32]),
    "lidar_ball_query_stack": (i32, [i32, i32, f32, i32, vp, vp, vp, vp, vp, vp]),
    "lidar_ball_query_stack2": (i32, [i32, i32, f32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "lidar_ball_query_grid_workspace_bytes": (sz, [i32, i32]),
    "lidar_ball_query_stack_grid": (i32, [i32, i32, i32, f32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "lidar_group_points_stack": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "lidar_group_rows_stack": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "lidar_group_rows_affine_stack": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "lidar_sa_layer2_max_supported": (i32, [i32, i32, i32]),
    "lidar_sa_layer2_max_stack": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "lidar_group_points_grad_stack": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "lidar_furthest_point_sampling": (i32, [i32, i32, i32, vp, vp, vp, vp]),
    "lidar_three_nn_stack": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "lidar_three_interpolate_stack": (i32, [i32, i32, vp, vp, vp, vp, vp]),
    "lidar_three_interpolate_grad_stack": (i32, [i32, i32, vp, vp, vp, vp, vp]),
    "lidar_ball_query_batch": (i32, [i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "lidar_group_points_batch": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_group_points_grad_batch": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_gather_points_batch": (i32, [i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_gather_points_grad_batch": (i32, [i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_three_nn_batch": (i32, [i32, i32, i32, vp, vp, vp, vp, vp]),
    "lidar_three_interpolate_batch": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "lidar_three_interpolate_grad_batch": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "lidar_roiaware_pool3d_forward": (i32, [i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "lidar_roiaware_pool3d_backward": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "lidar_points_in_boxes": (i32, [i32, i32, i32, vp, vp, vp, vp]),
    "lidar_roipoint_pool3d_forward": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "lidar_spconv_hash_capacity": (sz, [i32]),
    "lidar_spconv_build_hash": (i32, [vp, i32, i32, i32, i32, vp, sz, vp]),
    "lidar_spconv_subm_table": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
    "lidar_spconv_conv_table_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "lidar_spconv_conv_outputs": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, sz, vp]),
    "lidar_spconv_conv_tables": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "lidar_spconv_grid_init": (i32, [vp, sz, vp]),
    "lidar_spconv_grid_rows": (i32, [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp]),
    "lidar_spconv_grid_table": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "lidar_spconv_grid_table_t": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "lidar_spconv_grid_pad_rows": (i32, [vp, vp, i32, vp]),
    "lidar_spconv_transpose_table": (i32, [vp, i32, i32, i32, vp, vp]),
    "lidar_spconv_grid_outputs_workspace_bytes": (sz, [i32, i32]),
    "lidar_spconv_grid_outputs": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "lidar_spconv_implicit_gemm": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "lidar_spconv_implicit_gemm_fused": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    "lidar_spconv_row_masks": (i32, [vp, i32, i32, vp, vp]),
    "lidar_spconv_mask_group_workspace_bytes": (sz, [i32]),
    "lidar_spconv_mask_group_init": (i32, [vp, sz, vp]),
    "lidar_spconv_mask_group": (i32, [vp, i32, i32, vp, vp, vp, sz, vp]),
    "lidar_spconv_sorted_gemm_supported": (i32, [i32, i32, i32]),
    "lidar_spconv_implicit_gemm_sorted": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    "lidar_spconv_packed_floats": (sz, [i32, i32, i32]),
    "lidar_spconv_pack_weights": (i32, [vp, i32, i32, i32, vp, vp]),
    "lidar_spconv_implicit_gemm_sorted_packed": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    "lidar_spconv_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "lidar_spconv_wgrad_mfma_supported": (i32, [i32, i32, i32]),
    "lidar_spconv_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "lidar_spconv_wgrad_mfma": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "lidar_sparse_to_dense_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "lidar_sparse_to_dense": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "lidar_sparse_to_bev_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
    "lidar_bias_act_nhwc": (i32, [vp, vp, C.c_longlong, i32, i32, vp, i32, i32, vp]),
    "lidar_bias_act_upsample_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "lidar_dense_gemm_bias_act": (i32, [vp, C.c_longlong, i32, vp, i32, vp, i32, vp, i32, vp, sz, vp]),
    "lidar_dense_gemm_export_choices": (i32, [vp, i32]),
    "lidar_dense_gemm_import_choices": (i32, [vp, i32]),
    "lidar_wino_packed_floats": (sz, [i32, i32]),
    "lidar_wino_supported": (i32, [i32, i32]),
    "lidar_wino_pack_weights": (i32, [vp, i32, i32, vp, vp]),
    "lidar_wino_conv3x3_nhwc": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, i32, i32, vp]),
    "lidar_wino43_packed_floats": (sz, [i32, i32]),
    "lidar_wino43_supported": (i32, [i32, i32]),
    "lidar_wino43_pack_weights": (i32, [vp, i32, i32, vp, vp]),
    "lidar_wino43_conv3x3_nhwc": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, i32, i32, vp]),
    "lidar_wino_conv3x3_grouped_nhwc": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp]),
    "lidar_wino_conv3x3_grouped_compact_nhwc": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp]),
    "lidar_deconv_packed_floats": (sz, [i32, i32]),
    "lidar_deconv_supported": (i32, [i32, i32, i32]),
    "lidar_deconv_pack_weights": (i32, [vp, i32, i32, vp, vp]),
    "lidar_deconv_gemm_nhwc": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, i32, vp]),
    "lidar_anchor_scores": (i32, [vp, C.c_longlong, i32, i32, i32, i32, f32, vp, vp, vp]),
    "lidar_decode_topk": (i32, [vp, i32, C.c_longlong, i32, i32, i32, i32, i32, vp, i32, vp, f32, f32, f32, vp, vp]),
    "lidar_topk_workspace_bytes": (sz, [i32, C.c_longlong]),
    "lidar_topk_workspace_init": (i32, [vp, sz, i32, C.c_longlong, vp]),
    "lidar_anchor_scores_hist": (i32, [vp, i32, C.c_longlong, i32, i32, i32, i32, f32, vp, vp, vp, sz, vp]),
    "lidar_topk_desc": (i32, [vp, i32, C.c_longlong, i32, f32, f32, i32, vp, vp, vp, vp, sz, vp]),
    "lidar_post_nms_gather": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, C.c_longlong, i32, i32, vp, vp, vp, vp, vp]),
    "lidar_rotate_iou_eval": (i32, [vp, i32, vp, i32, i32, vp, vp]),
    "lidar_boxes_iou_bev_cpu": (i32, [vp, i32, vp, i32, vp]),
    "lidar_points_in_boxes_cpu": (i32, [vp, i32, vp, i32, vp]),
    "lidar_voxelize_cpu_scratch_bytes": (sz, [i32]),
    "lidar_voxelize_cpu": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, sz]),
}

_lib = None


class LidarHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LidarHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
    return _lib


def load_variant(path):
    """A second, independently loaded build of the library with the same signatures (tests: A/B variants from
    csrc/build.py:VARIANTS).  The product path never calls this."""
    if not os.path.exists(path):
        raise LidarHipError(f"{path} is missing: build it with lidardetection_amd/csrc/build.py (build_variants)")
    v = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(v, name)
        fn.restype = res
        fn.argtypes = args
    return v


def check(status, what):
    if status != 0:
        raise LidarHipError(f"{what} failed with status {status}")


def ptr(t):
    """device (or host) address of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    """Raw hipStream_t of torch's current stream on the current device (the private fast accessors avoid the
    ~0.1 ms torch.cuda.current_stream() spends in availability checks on every call)."""
    import torch
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:  # pragma: no cover - older/newer torch without the private accessors
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_f32(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def host_i32(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def require_cuda(*tensors, allow=()):
    """Every tensor handed to the C ABI: on the device, contiguous, and float32 or int32 (the only element types the kernels
    read; the reference assumes them without checking — a float64 / int64 tensor would be reinterpreted bit-wise).  `allow`:
    further dtypes a particular entry point takes (e.g. torch.int64 keep lists)."""
    import torch
    ok = (torch.float32, torch.int32) + tuple(allow)
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LidarHipError("expected a CUDA (ROCm) tensor; this library has no CPU path")
        if not t.is_contiguous():
            raise LidarHipError("expected a contiguous tensor")
        if t.dtype not in ok:
            raise LidarHipError(f"expected a float32 / int32 tensor, got {t.dtype} (shape {tuple(t.shape)})")


def require_last(t, k, what):
    """last dimension of `t` must be k (boxes: 7, points: 3, ...)"""
    if t is not None and (t.dim() == 0 or t.shape[-1] != k):
        raise LidarHipError(f"{what}: expected (..., {k}), got {tuple(t.shape)}")
