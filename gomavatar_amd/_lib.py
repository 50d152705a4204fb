"""ctypes binding of libgom_hip.so (C ABI: include/gom_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to
load, every product entry point raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p, POINTER

import torch  # noqa: F401  -- imported first so that libgom_hip.so binds to the SAME libamdhip64 torch uses

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOM_HIP_LIB") or os.path.join(_HERE, "libgom_hip.so")  # env override: experiment builds

GOM_ABI_VERSION = 11
GOM_FWD_REUSE_BINNING = 1
GOM_BWD_RECOMPUTE_FORWARD = 1
GOM_LOSS_BLOCKS = 256
(BUF_DEPTH, BUF_XY, BUF_CONIC_OPACITY, BUF_TILES_TOUCHED, BUF_RECT, BUF_TILE_BASE, BUF_KEYS, BUF_POINT_LIST, BUF_FINAL_T,
 BUF_N_CONTRIB, BUF_STATUS) = range(11)
OPT_SORT_CAP, OPT_PAIR_CAPACITY, OPT_PROFILE, OPT_SEG_SHIFT, OPT_TASK_GRID_PCT, OPT_SORT_MODE, OPT_BWD_MODE, OPT_FUSE_FACE, OPT_FUSE_LOSS = 0, 1, 2, 3, 4, 5, 6, 7, 8
SORT_AUTO, SORT_TILE_MERGE, SORT_DEPTH_RANK = 0, 1, 2
KERNEL_NAMES = ("preprocess", "scan_tiles", "emit", "sort", "seg_T", "seg_fwd", "combine", "seg_bwd", "preprocess_bwd", "depth_hist", "depth_rank")


class GomCamera(ctypes.Structure):
    _fields_ = [("H", c_int32), ("W", c_int32), ("tanfovx", c_float), ("tanfovy", c_float),
                ("view", c_float * 16), ("proj", c_float * 16), ("bg", c_float * 4)]


class GomFrame(ctypes.Structure):
    _fields_ = ([("N", c_int32), ("F", c_int32), ("H", c_int32), ("W", c_int32), ("sigma", c_float), ("c_rgb", c_float),
                 ("c_mask", c_float), ("cam", GomCamera)] +
                [(n, c_void_p) for n in (
                    "faces", "csr_off", "csr_idx", "lbs_weights", "vertices", "so3", "scale", "appearance", "cnl_gtfms", "dst_Rs",
                    "dst_Ts", "gt_rgb", "gt_mask", "bgcolor", "image", "loss_partials", "g_vertices", "g_so3", "g_scale",
                    "g_appearance", "work_RT", "work_fk", "work_vobs", "work_xyz", "work_cov6", "work_feat", "work_opacity",
                    "work_dimage", "work_dxyz", "work_dcov6", "work_dfeat", "work_dopacity", "work_dcorner", "work_radii")])


GOM_FRAME_FORWARD_ONLY = 1
GOM_FRAME_USE_GRAPH = 2
GOM_FRAME_BACKWARD_ONLY = 4

# name -> (restype, argtypes); every symbol include/gom_hip.h declares
SIGNATURES = {
    "gom_last_error": (c_char_p, []),
    "gom_abi_version": (c_int, []),
    "gom_state_create": (c_void_p, []),
    "gom_state_destroy": (None, [c_void_p]),
    "gom_state_set_option": (c_int, [c_void_p, c_int, c_int64]),
    "gom_state_poll": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int32), c_void_p]),
    "gom_state_kernel_times": (c_int, [c_void_p, POINTER(c_float)]),
    "gom_state_export": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "gom_raster_forward": (c_int, [c_void_p, POINTER(GomCamera), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_uint32, c_void_p]),
    "gom_raster_backward": (c_int, [c_void_p, POINTER(GomCamera), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "gom_raster_forward_dcam": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_uint32, c_void_p]),
    "gom_raster_backward_dcam": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "gom_linear_wgrad_slices": (c_int, []),
    "gom_linear_wgrad": (c_int, [c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_compose_forward": (c_int, [c_int, c_int] + [c_void_p] * 6),
    "gom_compose_backward": (c_int, [c_int, c_int] + [c_void_p] * 8),
    "gom_unpack_forward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 5),
    "gom_unpack_backward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 7),
    "gom_l1_terms_forward": (c_int, [c_int, c_int] + [c_void_p] * 5 + [c_int] + [c_void_p] * 3),
    "gom_loss_tail": (c_int, [c_int] + [c_void_p] * 10),
    "gom_camera_update_device": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "gom_l1_terms_backward": (c_int, [c_int, c_int] + [c_void_p] * 5 + [c_int] + [c_void_p] * 5),
    "gom_mlp3_forward": (c_int, [c_int64, c_int, c_int] + [c_void_p] * 14),
    "gom_mlp3_wgrad": (c_int, [c_int64, c_int, c_int] + [c_void_p] * 18),
    "gom_mlp3_backward": (c_int, [c_int64, c_int, c_int] + [c_void_p] * 15),
    "gom_posenc_forward": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "gom_posenc_backward": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_fk_forward": (c_int, [c_void_p] * 6),
    "gom_fk_backward": (c_int, [c_void_p] * 7),
    "gom_lbs_forward": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_fk_lbs_forward": (c_int, [c_int] + [c_void_p] * 9),
    "gom_face_forward": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "gom_face_backward": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_vertex_backward": (c_int, [c_int, c_int] + [c_void_p] * 11),
    "gom_conv3x3_bf16": (c_int, [c_int] * 5 + [c_void_p] * 5 + [c_uint32, c_void_p]),
    "gom_conv3x3_splits": (c_int, [c_int] * 5),
    "gom_conv3x3_bf16_splitk": (c_int, [c_int] * 5 + [c_void_p] * 5 + [c_uint32, c_int, c_void_p, c_void_p]),
    "gom_maxpool2x2_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "gom_maxpool2x2_backward_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_int, c_void_p]),
    "gom_lpips_prepare_bf16": (c_int, [c_int] * 3 + [c_void_p] * 3),
    "gom_lpips_unprepare_bf16": (c_int, [c_int] * 4 + [c_void_p] * 3),
    "gom_lpips_layer_forward_nhwc_bf16": (c_int, [c_int] * 3 + [c_void_p] * 5),
    "gom_lpips_layer_backward_nhwc_bf16": (c_int, [c_int] * 3 + [c_void_p] * 6),
    "gom_lpips_vgg_create": (c_void_p, [c_void_p] * 6),
    "gom_lpips_vgg_destroy": (None, [c_void_p]),
    "gom_lpips_vgg_set_first_layer": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gom_lpips_vgg_set_precision": (c_int, [c_void_p, c_int32]),
    "gom_lpips_vgg_value_and_grad": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_uint32, c_void_p]),
    "gom_lpips_vgg_target_features": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gom_mesh_raster_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "gom_mesh_raster_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 7),
    "gom_mesh_pix_to_face": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gom_ndc_from_world_forward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 5),
    "gom_ndc_from_world_backward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 6),
    "gom_vertex_normals_forward": (c_int, [c_int, c_int] + [c_void_p] * 8),
    "gom_vertex_normals_backward": (c_int, [c_int, c_int] + [c_void_p] * 10),
    "gom_mesh_laplacian": (c_int, [c_int] + [c_void_p] * 6),
    "gom_mesh_laplacian_backward": (c_int, [c_int] + [c_void_p] * 6),
    "gom_mesh_normal_consistency": (c_int, [c_int] + [c_void_p] * 6),
    "gom_mesh_normal_consistency_backward": (c_int, [c_int, c_int, c_int] + [c_void_p] * 11),
    "gom_mesh_color_consistency": (c_int, [c_int, c_int] + [c_void_p] * 5),
    "gom_mesh_color_consistency_backward": (c_int, [c_int, c_int] + [c_void_p] * 6),
    "gom_ssim": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p]),
    "gom_lpips_layer_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_lpips_layer_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_frame_loss_slots": (c_int, [c_int, c_int]),
    "gom_frame_forward_backward": (c_int, [c_void_p, POINTER(GomFrame), c_uint32, c_void_p]),
    "gom_batch_forward_backward": (c_int, [c_void_p, POINTER(GomFrame), c_int32, c_void_p, c_uint32, c_void_p]),
    "gom_split_forward_backward": (c_int, [POINTER(c_void_p), POINTER(GomFrame), c_int32, POINTER(c_int32), POINTER(c_void_p), c_uint32, c_void_p]),
    "gom_adam_flat": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int64), POINTER(c_float), POINTER(c_int64), c_int64, c_float, c_float,
                              c_float, c_float, c_void_p]),
    "gom_adam_flat_graphable": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int64), POINTER(c_float), POINTER(c_int64), c_int64, c_void_p, c_float,
                                        c_float, c_float, c_float, c_float, c_void_p]),
    "gom_shade_workspace_ints": (c_int, [c_int64]),
    "gom_shade_select": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_mlp3_pack_elems": (c_int, []),
    "gom_mlp3_forward_rows": (c_int, [c_int64, c_void_p, c_int, c_int] + [c_void_p] * 15),
    "gom_mlp3_backward_rows": (c_int, [c_int64, c_void_p, c_int, c_int] + [c_void_p] * 16),
    "gom_mlp3_wgrad_rows": (c_int, [c_int64, c_void_p, c_int, c_int] + [c_void_p] * 18),
    "gom_shade_scatter": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "gom_shade_backward_gather": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "gom_shade_backward_scatter": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gom_adam_multi": (c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_float), c_int64, c_void_p,
                               c_double, c_double, c_double, c_void_p]),
    "gom_peer_reduce_create": (c_void_p, [c_int32, c_int32, c_int64]),
    "gom_peer_reduce_handle": (c_int, [c_void_p, c_void_p]),
    "gom_peer_reduce_connect": (c_int, [c_void_p, c_void_p]),
    "gom_peer_reduce_buffer": (c_void_p, [c_void_p]),
    "gom_peer_reduce_run": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "gom_peer_reduce_run_adam": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int64), POINTER(c_float), POINTER(c_int64), c_int64, c_float, c_float,
                                         c_float, c_void_p]),
    "gom_peer_reduce_run_zero1": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int64), POINTER(c_float), POINTER(c_int64), c_int64, c_float, c_float,
                                          c_float, c_void_p]),
    "gom_peer_reduce_set_timeout": (c_int, [c_void_p, ctypes.c_double]),
    "gom_peer_reduce_poll": (c_int, [c_void_p]),
    "gom_peer_reduce_reset": (c_int, [c_void_p, ctypes.c_uint32]),
    "gom_peer_reduce_epoch": (ctypes.c_uint32, [c_void_p]),
    "gom_peer_reduce_status": (c_int, [c_void_p]),
    "gom_peer_reduce_destroy": (None, [c_void_p]),
    "gom_l1_loss": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


# include/gom_hip_lab.h: present only in a library built with -DGOM_LAB (experiments that were measured and not adopted)
LAB_SIGNATURES = {
    "gom_state_set_frame_optimizer": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int64), POINTER(c_float), c_void_p, c_float,
                                              c_float, c_float, c_float, c_float]),
}


def has_lab() -> bool:
    """True when the loaded library is a -DGOM_LAB build (GOM_HIP_LIB=.../libgom_hip_lab.so)."""
    return hasattr(load(), "gom_state_set_frame_optimizer")


def load() -> ctypes.CDLL:
    """Load libgom_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m gomavatar_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in LAB_SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gom_abi_version() != GOM_ABI_VERSION:
        raise RuntimeError("libgom_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().gom_last_error()
        raise RuntimeError(f"libgom_hip error {rc}: {msg.decode() if msg else '?'}")


def stream_ptr() -> int:
    """The hipStream_t torch is currently enqueuing on."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL).
    The caller must keep `t` referenced until the launch that uses the pointer has been
    enqueued: torch's caching allocator may hand a dead temporary's block to the next
    allocation, whose producer kernel could then run BEFORE ours on the same stream."""
    if t is None:
        return 0
    assert t.is_cuda and t.is_contiguous(), "libgom_hip needs contiguous device tensors"
    return t.data_ptr()


def make_camera(H: int, W: int, tanfovx: float, tanfovy: float, view16, proj16, bg) -> GomCamera:
    c = GomCamera()
    c.H, c.W = int(H), int(W)
    c.tanfovx, c.tanfovy = float(tanfovx), float(tanfovy)
    for i in range(16):
        c.view[i] = float(view16[i])
        c.proj[i] = float(proj16[i])
    for i in range(4):
        c.bg[i] = float(bg[i]) if i < len(bg) else 0.0
    return c
