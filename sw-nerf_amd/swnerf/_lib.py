"""ctypes binding of libswnerf_hip.so (include/swnerf.h).  There is NO fallback: if the
shared library is missing or a GPU is absent, every op raises - the product path never
routes through a CPU implementation."""
import ctypes
import os
from ctypes import c_int, c_int64, c_double, c_void_p, c_size_t, c_char_p, POINTER, Structure

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libswnerf_hip.so")

NET_CANON, NET_DNERF, NET_NOVIEW = 0, 1, 2

EXPORTS = ["swnerf_version", "swnerf_last_error", "swnerf_packed_floats", "swnerf_pack_net", "swnerf_pack_net_noview", "swnerf_mlp_forward_noview",
           "swnerf_get_rays", "swnerf_ndc_rays", "swnerf_pack_ray_batch", "swnerf_raw2outputs", "swnerf_raw2outputs_backward",
           "swnerf_sample_pdf", "swnerf_sample_coarse", "swnerf_embed", "swnerf_mlp_forward", "swnerf_query_points", "swnerf_render_pass",
           "swnerf_packed_bwd_floats", "swnerf_act_floats_per_row", "swnerf_mask_floats", "swnerf_mlp_forward_train", "swnerf_pack_net_bwd",
           "swnerf_mlp_backward_dx", "swnerf_gemm_tn", "swnerf_gemm_tn_fused", "swnerf_gemm_tn_group", "swnerf_feature_finish", "swnerf_canon_narrow_grads", "swnerf_deform_narrow_grads", "swnerf_noview_narrow_grads",
           "swnerf_packed_bwd_floats_kind", "swnerf_pack_net_bwd_kind", "swnerf_deform_forward_train",
           "swnerf_mlp_backward_dx_pts", "swnerf_deform_backward_dx",
           "swnerf_train_rows", "swnerf_xs_floats_per_row", "swnerf_render_pass_train", "swnerf_render_pass_backward", "swnerf_unslot_grad",
           "swnerf_render_pass_train_dnerf", "swnerf_render_pass_backward_dnerf", "swnerf_unslot_grad_time",
           "swnerf_packed_bwd_noview_floats", "swnerf_pack_net_bwd_noview", "swnerf_render_pass_backward_noview",
           "swnerf_linear", "swnerf_gemm_nn", "swnerf_relu_mask",
           "swnerf_packed_x3_floats", "swnerf_pack_net_x3", "swnerf_render_pass_x3",
           "swnerf_packed_x3_floats_kind", "swnerf_pack_net_x3_kind"]
BWD_CANON, BWD_CANON_INPUT_GRAD, BWD_DEFORM, BWD_DNERF_FUSED = 0, 1, 2, 3


class GemmItem(Structure):
    """struct swnerf_gemm_item (include/swnerf.h)"""
    _fields_ = [("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int), ("C", c_void_p), ("ldc", c_int), ("bias", c_void_p),
                ("B2", c_void_p), ("ldb2", c_int), ("Ni2", c_int), ("C2", c_void_p), ("ldc2", c_int),
                ("A2", c_void_p), ("lda2", c_int), ("No2", c_int), ("C3", c_void_p), ("ldc3", c_int), ("bias3", c_void_p)]


class PassArgs(Structure):
    """struct swnerf_pass_args (include/swnerf.h)"""
    _fields_ = [
        ("ray_batch", c_void_p), ("n_rays", c_int64), ("cols", c_int), ("kind", c_int),
        ("packed", c_void_p), ("run_deform", c_int), ("L_pos", c_int), ("L_dir", c_int), ("L_time", c_int),
        ("n_samples", c_int), ("z_vals", c_void_p), ("lindisp", c_int), ("t_rand", c_void_p),
        ("noise", c_void_p), ("white_bkgd", c_int),
        ("rgb_map", c_void_p), ("disp_map", c_void_p), ("acc_map", c_void_p), ("depth_map", c_void_p),
        ("weights", c_void_p), ("raw", c_void_p), ("dx", c_void_p), ("z_out", c_void_p),
        ("n_importance", c_int), ("u", c_void_p), ("z_fine", c_void_p), ("z_std", c_void_p),
        ("out_ch", c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"swnerf: {LIB_PATH} not found - build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the render path")
    # torch bundles its own libamdhip64/libhsa-runtime64; ours must bind to THAT copy (same SONAME), not
    # pull /opt/rocm's into the process first - two HSA runtimes in one process cannot both own the GPU
    # ("no ROCm-capable device is detected").  So: torch first, always.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.swnerf_version.restype = c_int
    L.swnerf_last_error.restype = c_char_p
    L.swnerf_packed_floats.restype = c_size_t
    L.swnerf_packed_floats.argtypes = [c_int]
    L.swnerf_pack_net.argtypes = [c_int, POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_void_p]
    L.swnerf_pack_net_noview.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]
    L.swnerf_mlp_forward_noview.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]
    L.swnerf_get_rays.argtypes = [c_int, c_int, c_double, c_double, c_double, c_double, c_int,
                                  POINTER(ctypes.c_float), c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    L.swnerf_ndc_rays.argtypes = [c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_void_p]
    L.swnerf_pack_ray_batch.argtypes = [c_void_p, c_void_p, c_int64, c_double, c_double, c_int, c_double,
                                        c_int, c_int, c_int, c_double, c_void_p, c_void_p]
    L.swnerf_raw2outputs.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_raw2outputs_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_sample_pdf.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    L.swnerf_sample_coarse.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_embed.argtypes = [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]
    L.swnerf_mlp_forward.argtypes = [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p]
    L.swnerf_query_points.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]
    L.swnerf_render_pass.argtypes = [POINTER(PassArgs), c_void_p]
    L.swnerf_packed_bwd_floats.restype = c_size_t
    L.swnerf_packed_bwd_floats.argtypes = []
    L.swnerf_act_floats_per_row.restype = c_size_t
    L.swnerf_act_floats_per_row.argtypes = []
    L.swnerf_mask_floats.restype = c_size_t
    L.swnerf_mask_floats.argtypes = [c_int64]
    L.swnerf_mlp_forward_train.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_pack_net_bwd.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]
    L.swnerf_mlp_backward_dx.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    L.swnerf_gemm_tn.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]
    L.swnerf_gemm_tn_fused.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p,
                                       c_void_p, c_int, c_int, c_void_p, c_int,
                                       c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]
    L.swnerf_gemm_tn_group.argtypes = [POINTER(GemmItem), c_int, c_int64, c_void_p]
    L.swnerf_canon_narrow_grads.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64] + [c_void_p] * 10
    L.swnerf_deform_narrow_grads.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64] + [c_void_p] * 6
    L.swnerf_noview_narrow_grads.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int64] + [c_void_p] * 5
    L.swnerf_feature_finish.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_packed_bwd_floats_kind.restype = c_size_t
    L.swnerf_packed_bwd_floats_kind.argtypes = [c_int]
    L.swnerf_pack_net_bwd_kind.argtypes = [c_int, POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]
    L.swnerf_deform_forward_train.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_mlp_backward_dx_pts.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]
    L.swnerf_deform_backward_dx.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    L.swnerf_train_rows.restype = c_int64
    L.swnerf_train_rows.argtypes = [c_int64, c_int]
    L.swnerf_xs_floats_per_row.argtypes = []
    L.swnerf_render_pass_train.argtypes = [POINTER(PassArgs), c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_render_pass_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_packed_bwd_noview_floats.restype = c_size_t
    L.swnerf_packed_bwd_noview_floats.argtypes = []
    L.swnerf_pack_net_bwd_noview.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]
    L.swnerf_render_pass_backward_noview.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int,
                                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.swnerf_render_pass_train_dnerf.argtypes = [POINTER(PassArgs)] + [c_void_p] * 7
    L.swnerf_render_pass_backward_dnerf.argtypes = [c_void_p] * 6 + [c_int] + [c_void_p] * 3 + [c_int64, c_int, c_int, c_int] + [c_void_p] * 9
    L.swnerf_unslot_grad_time.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]
    L.swnerf_unslot_grad.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]
    L.swnerf_linear.argtypes = [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]
    L.swnerf_gemm_nn.argtypes = [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]
    L.swnerf_relu_mask.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    L.swnerf_packed_x3_floats.restype = c_size_t
    L.swnerf_packed_x3_floats.argtypes = []
    L.swnerf_pack_net_x3.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p, c_void_p]
    L.swnerf_render_pass_x3.argtypes = [POINTER(PassArgs), c_int, c_void_p]
    L.swnerf_packed_x3_floats_kind.restype = c_size_t
    L.swnerf_packed_x3_floats_kind.argtypes = [c_int]
    L.swnerf_pack_net_x3_kind.argtypes = [c_int, POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    for name in EXPORTS:
        if name not in ("swnerf_last_error", "swnerf_packed_floats", "swnerf_packed_bwd_floats", "swnerf_act_floats_per_row",
                        "swnerf_packed_bwd_floats_kind", "swnerf_mask_floats", "swnerf_train_rows", "swnerf_packed_bwd_noview_floats"):
            getattr(L, name).restype = c_int
    if L.swnerf_version() != 110:
        raise RuntimeError(f"swnerf: {LIB_PATH} has version {L.swnerf_version()}, expected 110 - rebuild it "
                           "(python __graft_entry__.py)")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().swnerf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"swnerf.{what} failed (code {rc}): {msg}")


def stream_of(t):
    import torch
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def dev_f32(t, name, shape_last=None):
    """Validate a device operand the way the C ABI requires (fp32, contiguous, cuda)."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"swnerf: {name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"swnerf: {name} must live on the GPU (got device {t.device}); "
                           "the HIP render path has no CPU implementation")
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    if shape_last is not None and t.shape[-1] != shape_last:
        raise ValueError(f"swnerf: {name} last dim must be {shape_last}, got {tuple(t.shape)}")
    return t
