"""ctypes binding of libeae.so (C ABI declared in include/eae.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EAE_LIB_PATH") or os.path.join(HERE, "libeae.so")     # EAE_LIB_PATH: diagnostic build variants

c_float_p = C.POINTER(C.c_float)
c_ll_p = C.POINTER(C.c_longlong)
vp = C.c_void_p


class EaeConfig(C.Structure):
    _fields_ = [("latent_dim", C.c_int), ("num_classes", C.c_int), ("image_h", C.c_int), ("image_w", C.c_int),
                ("max_batch", C.c_int), ("quant", C.c_int), ("side_streams", C.c_int)]


class EaeStepIO(C.Structure):
    _fields_ = [("x", vp), ("labels", vp), ("B", C.c_int), ("train", C.c_int), ("head", C.c_int), ("alpha", C.c_float),
                ("x_hat", vp), ("logits", vp), ("z", vp), ("loss_accum", vp), ("loss_last", vp)]


SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p)


class EaeSrc(C.Structure):
    _fields_ = [("p0", vp), ("p1", vp), ("coef", vp), ("mode", C.c_int)]


_PROTOS = {
    "eae_last_error": (C.c_char_p, []),
    "eae_version": (C.c_int, []),
    "eae_ae_layout": (C.c_int, [C.POINTER(EaeConfig), c_ll_p, c_ll_p]),
    "eae_create": (C.c_int, [C.POINTER(EaeConfig), C.POINTER(vp)]),
    "eae_destroy": (C.c_int, [vp]),
    "eae_bind": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "eae_gate_timeouts": (C.c_longlong, [vp]),
    "eae_gate_timeouts_clear": (C.c_int, [vp]),
    "eae_gate_timeouts_nosync": (C.c_longlong, [vp]),
    "eae_encoder_backward": (C.c_int, [vp, vp, C.c_longlong, vp, vp]),
    "eae_decoder_backward": (C.c_int, [vp, vp, C.c_longlong, vp, vp, vp]),
    "eae_fp8_calibrate": (C.c_int, [vp, vp, vp, C.c_int]),
    "eae_fp8_scales": (C.c_int, [vp, vp]),
    "eae_params_changed": (C.c_int, [vp]),
    "eae_set_graph": (C.c_int, [vp, C.c_int]),
    "eae_set_adam_step": (C.c_int, [vp, C.c_longlong]),
    "eae_get_adam_step": (C.c_longlong, [vp]),
    "eae_ae_forward": (C.c_int, [vp, vp, C.POINTER(EaeStepIO)]),
    "eae_forward_generation": (C.c_longlong, [vp]),
    "eae_ae_backward": (C.c_int, [vp, vp, C.c_longlong, vp, vp, vp, vp, vp]),
    "eae_ae_grad_step": (C.c_int, [vp, vp, C.POINTER(EaeStepIO)]),
    "eae_adam_step": (C.c_int, [vp, vp, C.c_float, C.c_float]),
    "eae_ae_grad_step_begin": (C.c_int, [vp, vp, C.POINTER(EaeStepIO)]),
    "eae_ae_grad_step_end": (C.c_int, [vp, vp]),
    "eae_side_stream": (vp, [vp]),
    "eae_debug_read": (C.c_longlong, [vp, C.c_int, C.c_int, vp, C.c_longlong]),
    "eae_dp_stream": (vp, [vp, C.c_int]),
    "eae_adam_step_scaled": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float]),
    "eae_dp_local_bad": (C.c_int, [vp, vp, vp]),
    "eae_adam_step_dp": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float, vp]),
    "eae_dp_unique_id": (C.c_int, [vp]),
    "eae_dp_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "eae_dp_world": (C.c_int, [vp]),
    "eae_dp_destroy": (C.c_int, [vp]),
    "eae_dp_allreduce_bucket": (C.c_int, [vp, vp, C.c_longlong, C.c_longlong]),
    "eae_dp_broadcast": (C.c_int, [vp, vp, vp, C.c_longlong, C.c_int]),
    "eae_ae_dp_train_step": (C.c_int, [vp, vp, C.POINTER(EaeStepIO), C.c_float, C.c_int]),
    "eae_ae_train_step": (C.c_int, [vp, vp, C.POINTER(EaeStepIO), C.c_float]),
    "eae_group_train_step": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp, C.POINTER(EaeStepIO), C.POINTER(C.c_float)]),
    "eae_group_forward": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp, C.POINTER(EaeStepIO)]),
    "eae_set_geometry_mult": (C.c_int, [C.c_int]),
    "eae_streams_share_queue": (C.c_int, [vp, vp]),
    "eae_reserve_stream": (C.c_int, [vp, C.c_int]),
    "eae_sync_bn_acc_elems": (C.c_longlong, [vp]),
    "eae_set_sync_bn": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "eae_encoder_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "eae_decoder_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "eae_debug_copy": (C.c_int, [vp, C.c_int, vp, C.c_longlong]),
    "eae_profile_enable": (C.c_int, [vp, C.c_int]),
    "eae_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), c_ll_p]),
    "eae_profile_read2": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), c_ll_p]),
    "eae_op_conv_s2": (C.c_int, [vp, C.c_int, EaeSrc, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]),
    "eae_op_conv_s2_ntiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "eae_op_edge_conv": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]),
    "eae_op_edge_wgrad": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, EaeSrc, vp, C.c_longlong, vp]),
    "eae_op_deconv4_loss": (C.c_int, [vp, EaeSrc, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_float, vp, vp, vp]),
    "eae_op_wgrad_s2": (C.c_int, [vp, EaeSrc, EaeSrc, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_longlong, vp]),
    "eae_op_wgrad_s2_fp8": (C.c_int, [vp, EaeSrc, EaeSrc, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_longlong, vp, vp]),
    "eae_op_conv_s2_fp8": (C.c_int, [vp, C.c_int, EaeSrc, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]),
    "eae_op_bn_finalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_longlong, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp]),
    "eae_op_bn_eval_coef": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_float, vp]),
    "eae_op_bn_bwd_finalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_longlong, vp, vp, vp, vp, vp]),
    "eae_op_pack3x3": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "eae_op_fc_splitk": (C.c_int, [vp, EaeSrc, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_longlong, vp]),
    "eae_op_fc_bias_bf16": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "eae_op_fc_wgrad": (C.c_int, [vp, C.c_int, EaeSrc, EaeSrc, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "eae_op_head_scratch_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "eae_op_head_ce": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_longlong]),
    "eae_op_sigmoid_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "eae_augment": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_ulonglong, C.c_ulonglong, vp, vp]),
    "eae_op_adam": (C.c_int, [vp, vp, vp, vp, vp, C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_longlong]),
    "eae_mlp_layout": (C.c_int, [C.c_int, C.c_int, c_ll_p, c_ll_p]),
    "eae_mlp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "eae_mlp_destroy": (C.c_int, [vp]),
    "eae_mlp_bind": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "eae_mlp_set_adam_step": (C.c_int, [vp, C.c_longlong]),
    "eae_mlp_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_ulonglong, vp, vp]),
    "eae_mlp_backward": (C.c_int, [vp, vp, vp, C.c_int, C.c_ulonglong, vp, vp]),
    "eae_mlp_train_step": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_ulonglong, vp, vp, vp]),
    "eae_mlp_eval_step": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp]),
}

EXPORTS = tuple(_PROTOS.keys())
_lib = None


class EaeError(RuntimeError):
    pass


def load():
    """Load libeae.so (built by build.py).  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EaeError(f"{LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py build, "
                       "or eae_amd.build.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().eae_last_error()
        raise EaeError(f"libeae error {rc}: {msg.decode() if msg else '?'}")
