"""ctypes binding of libodhip.so (include/odhip.h).  There is NO fallback: a missing library is a hard error.

This is the binding a pytoolkit maintainer would add for the path below `ObjectDetector.predict`
(reference voc_validate.py:27) -- see INTEGRATION.md.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libodhip.so"

OD_ACT_LINEAR, OD_ACT_LEAKY, OD_ACT_ELU = 0, 1, 2
OD_RES_NONE, OD_RES_SAME, OD_RES_UP2 = 0, 1, 2
OD_DT_F16, OD_DT_F32, OD_DT_BF16 = 0, 1, 2
OD_OP_CONV, OD_OP_CONV_FIRST, OD_OP_BNECK, OD_OP_STEM, OD_OP_WIDE = 1, 2, 3, 4, 5

ACT_ENUM = {None: OD_ACT_LINEAR, "linear": OD_ACT_LINEAR, "leaky": OD_ACT_LEAKY, "elu": OD_ACT_ELU}


class ConvDesc(C.Structure):
    C_NAME = "od_conv_desc"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
        ("res", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
        ("res_mode", C.c_int32), ("out_dtype", C.c_int32),
        ("out_batch_stride", C.c_int64), ("out_pix_stride", C.c_int64),
        ("tile_cfg", C.c_int32), ("transposed", C.c_int32), ("splitk", C.c_int32),
        ("splitk_workspace", C.c_void_p), ("splitk_workspace_bytes", C.c_int64),
        ("bn_partials", C.c_void_p), ("bn_partials_bytes", C.c_int64),
        # the pointwise layer that consumes this launch's output (None = no second layer)
        ("w2", C.c_void_p), ("scale2", C.c_void_p), ("bias2", C.c_void_p), ("out2", C.c_void_p),
        ("Cout2", C.c_int32), ("act2", C.c_int32), ("alpha2", C.c_float), ("pad2_", C.c_int32),
        # grouped launch: the same layer over nseg <= 3 maps of different sizes (0 / 1 = an ordinary launch)
        ("nseg", C.c_int32), ("pad3_", C.c_int32), ("seg_x", C.c_void_p * 3), ("seg_out", C.c_void_p * 3),
        ("seg_H", C.c_int32 * 3), ("seg_W", C.c_int32 * 3),
    ]


class AugParams(C.Structure):
    C_NAME = "od_aug_params"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("src_offset", C.c_int64), ("src_h", C.c_int32), ("src_w", C.c_int32),
                ("crop_x1", C.c_float), ("crop_y1", C.c_float), ("crop_x2", C.c_float), ("crop_y2", C.c_float),
                ("flip", C.c_int32), ("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float),
                ("n_erase", C.c_int32), ("erase", (C.c_float * 4) * 3), ("erase_rgb", (C.c_uint8 * 4) * 3)]


class BneckDesc(C.Structure):
    C_NAME = "od_bneck_desc"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [
        ("x", C.c_void_p), ("w1", C.c_void_p), ("scale1", C.c_void_p), ("bias1", C.c_void_p),
        ("w3", C.c_void_p), ("scale3", C.c_void_p), ("bias3", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
    ]


class StemDesc(C.Structure):
    C_NAME = "od_stem_desc"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [
        ("x", C.c_void_p), ("w0", C.c_void_p), ("scale0", C.c_void_p), ("bias0", C.c_void_p),
        ("w3", C.c_void_p), ("scale3", C.c_void_p), ("bias3", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
    ]


class SgdSeg(C.Structure):
    C_NAME = "od_sgd_seg"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("offset", C.c_int64), ("count", C.c_int64), ("lr", C.c_float), ("weight_decay", C.c_float)]


class WgradRed(C.Structure):
    C_NAME = "od_wgrad_red"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("dw_offset", C.c_int64), ("count", C.c_int64), ("slabs", C.c_void_p), ("nslabs", C.c_int32),
                ("pad_", C.c_int32)]


class PackLayer(C.Structure):
    C_NAME = "od_pack_layer"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("w_offset", C.c_int64), ("w_fwd", C.c_void_p), ("w_bwd", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("ksize", C.c_int32), ("pad_", C.c_int32)]


class WideDesc(C.Structure):
    C_NAME = "od_wide_desc"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("y", C.c_void_p), ("res", C.c_void_p), ("out32", C.c_void_p), ("out16", C.c_void_p),
                ("out_hilo", C.c_void_p), ("M", C.c_int64), ("C", C.c_int32), ("res_f32", C.c_int32),
                ("res_up2", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("pad_", C.c_int32)]


class PlanOp(C.Structure):
    C_NAME = "od_plan_op"  # the struct of include/odhip.h this mirrors (layout checked by tests/test_host_logic.py)
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("conv", ConvDesc), ("bneck", BneckDesc), ("stem", StemDesc),
                ("wide", WideDesc)]


STRUCTS = (ConvDesc, AugParams, BneckDesc, StemDesc, SgdSeg, WgradRed, PackLayer, WideDesc, PlanOp)


class OdError(RuntimeError):
    pass


_lib = None

_PROTOS = {
    # name: (restype, argtypes)
    "od_last_error": (C.c_char_p, []),
    "od_version": (C.c_int, []),
    "od_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "od_ctx_destroy": (C.c_int, [C.c_void_p]),
    "od_sizeof": (C.c_long, [C.c_char_p]),
    "od_offsetof": (C.c_long, [C.c_char_p, C.c_char_p]),
    "od_struct_fields": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "od_stream_create_cu_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "od_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "od_conv_weight_dims": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "od_conv_num_tile_cfgs": (C.c_int, []),
    "od_conv2d_fwd": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_void_p]),
    "od_conv2d_fwd_bn_rows": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc)]),
    "od_stem_supported": (C.c_int, [C.c_int, C.c_int]),
    "od_stem_fwd": (C.c_int, [C.c_void_p, C.POINTER(StemDesc), C.c_void_p]),
    "od_wide_add": (C.c_int, [C.c_void_p, C.POINTER(WideDesc), C.c_void_p]),
    "od_bottleneck_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "od_bottleneck_fwd": (C.c_int, [C.c_void_p, C.POINTER(BneckDesc), C.c_void_p]),
    "od_conv_first_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "od_upsample2x_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "od_head_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "od_decode_locs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                 C.c_int, C.c_void_p]),
    "od_topk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "od_topk_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_nms_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "od_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_gather_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "od_detect_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "od_detect_workspace_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "od_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                            C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_gather_detections_pred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "od_assign_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "od_assign_anchors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "od_loss_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_conv2d_bwd_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "od_bn_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                             C.c_int, C.c_void_p]),
    "od_bn_workspace_bytes": (C.c_size_t, [C.c_longlong, C.c_int]),
    "od_bn_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_bn_stats_from_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_void_p]),
    "od_scale_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "od_bn_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_conv2d_bwd_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "od_down2_sum_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "od_add_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "od_pred_grad_to_level": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_void_p]),
    "od_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float,
                              C.c_float, C.c_float, C.c_void_p]),
    "od_conv2d_bwd_weight_splits": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "od_conv2d_bwd_weight_slabs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "od_wgrad_reduce_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "od_sgd_step_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                    C.c_float, C.c_void_p, C.c_void_p]),
    "od_grad_nonfinite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "od_copy_if_nonzero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "od_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "od_cast_bf16_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "od_pack_weights_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "od_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    "od_conv_first_bwd_weight_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "od_conv_first_bwd_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "od_comm_unique_id_bytes": (C.c_int, []),
    "od_comm_get_unique_id": (C.c_int, [C.c_void_p, C.c_int]),
    "od_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "od_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "od_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "od_comm_destroy": (C.c_int, [C.c_void_p]),
    "od_aug_params_bytes": (C.c_int, []),
    "od_augment_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "od_plan_create": (C.c_int, [C.c_void_p, C.POINTER(PlanOp), C.c_int, C.POINTER(C.c_void_p)]),
    "od_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "od_plan_capture": (C.c_int, [C.c_void_p, C.c_void_p]),
    "od_plan_replay": (C.c_int, [C.c_void_p, C.c_void_p]),
    "od_plan_destroy": (C.c_int, [C.c_void_p]),
    "od_plan_time_ops": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "od_plan_op_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def load(path: os.PathLike | None = None):
    """dlopen libodhip.so and declare every prototype.  Raises OdError when the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = pathlib.Path(path) if path else LIB_PATH
    if not p.exists():
        raise OdError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C object_detector_amd/csrc`). There is no CPU fallback for this path.")
    # torch first: libodhip.so must bind to the HIP runtime torch already loaded (the one that owns the tensors whose
    # pointers cross the C ABI).  Loaded the other way round, a second runtime comes up and sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(str(p))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().od_last_error()
        raise OdError(f"{what or 'libodhip'} failed (rc={rc}): {msg.decode() if msg else '?'}")


def conv_weight_dims(cout: int, cin: int, ksize: int):
    a, b = C.c_int(), C.c_int()
    check(load().od_conv_weight_dims(cout, cin, ksize, C.byref(a), C.byref(b)), "od_conv_weight_dims")
    return a.value, b.value
