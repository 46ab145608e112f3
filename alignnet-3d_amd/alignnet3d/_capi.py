"""ctypes declarations for include/alignnet_hip.h (kept field-for-field in sync with it)."""
import ctypes as C
import os

MAX_WIDTHS = 8
ABI_VERSION = 1


class Widths(C.Structure):
    _fields_ = [("n", C.c_int32), ("w", C.c_int32 * MAX_WIDTHS)]

    @staticmethod
    def of(seq):
        seq = list(seq)
        if len(seq) > MAX_WIDTHS:
            raise ValueError(f"at most {MAX_WIDTHS} widths per list, got {len(seq)}")
        w = Widths()
        w.n = len(seq)
        for i, v in enumerate(seq):
            w.w[i] = int(v)
        return w


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("num_points", C.c_int32), ("num_channels", C.c_int32),
        ("num_bins", C.c_int32), ("backbone", C.c_int32),
        ("s1_conv", Widths), ("s1_fc", Widths), ("s2_conv", Widths), ("s2_fc", Widths), ("emb_conv", Widths), ("rem_fc", Widths),
        ("s1_keep", C.c_float), ("s2_keep", C.c_float), ("rem_keep", C.c_float),
        ("angle_factor", C.c_float), ("early_stage_factor", C.c_float), ("accept_inverted_angle", C.c_int32),
        ("batch_size", C.c_int32), ("ntrain", C.c_int32), ("learning_rate", C.c_float), ("lr_step", C.c_int32),
        ("lr_rate", C.c_float), ("lr_per_epoch", C.c_int32), ("bn_init", C.c_float), ("bn_rate", C.c_float),
        ("bn_clip", C.c_float), ("bn_step", C.c_int32), ("bn_per_epoch", C.c_int32), ("optimizer", C.c_int32),
        ("momentum", C.c_float), ("seed", C.c_uint64),
    ]


FP = C.POINTER(C.c_float)


class Outputs(C.Structure):
    _fields_ = [(n, FP) for n in (
        "pred_translations", "pred_remaining_angle_logits", "pred_s1_pc1centers", "pred_s1_pc2centers",
        "pred_s2_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits")]


class Labels(C.Structure):
    _fields_ = [(n, FP) for n in ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")]


class StepResult(C.Structure):
    _fields_ = [("step", C.c_int64), ("loss", C.c_float), ("learning_rate", C.c_float), ("bn_decay", C.c_float),
                ("summaries", C.c_float * 16)]


class State(C.Structure):
    _fields_ = [("step", C.c_int64), ("learning_rate", C.c_float), ("bn_decay", C.c_float)]


H = C.c_void_p

# every symbol include/alignnet_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "alignnet_create": (C.c_int, [C.POINTER(Config), C.POINTER(H)]),
    "alignnet_destroy": (None, [H]),
    "alignnet_last_error": (C.c_char_p, [H]),
    "alignnet_abi_version": (C.c_int, []),
    "alignnet_init_params": (C.c_int, [H, C.c_uint64]),
    "alignnet_num_params": (C.c_int, [H]),
    "alignnet_param_info": (C.c_int, [H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "alignnet_get_param": (C.c_int, [H, C.c_char_p, FP, C.c_size_t]),
    "alignnet_set_param": (C.c_int, [H, C.c_char_p, FP, C.c_size_t]),
    "alignnet_forward": (C.c_int, [H, FP, FP, C.c_int32, C.POINTER(Outputs)]),
    "alignnet_forward_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Outputs)]),
    "alignnet_forward_submit": (C.c_int, [H, FP, FP, C.c_int32, C.POINTER(Outputs)]),
    "alignnet_forward_wait": (C.c_int, [H]),
    "alignnet_eval_loss": (C.c_int, [H, C.POINTER(Labels), C.c_int32, FP, FP]),
    "alignnet_synchronize": (C.c_int, [H]),
    "alignnet_train_step": (C.c_int, [H, FP, FP, C.POINTER(Labels), C.c_int32, FP, C.POINTER(StepResult), C.POINTER(Outputs)]),
    "alignnet_train_step_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(Labels), C.c_int32, C.POINTER(StepResult)]),
    "alignnet_train_forward_backward": (C.c_int, [H, FP, FP, C.POINTER(Labels), C.c_int32, FP, C.POINTER(StepResult), C.POINTER(Outputs)]),
    "alignnet_grad_buffer": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "alignnet_apply_gradients": (C.c_int, [H, C.c_float]),
    "alignnet_get_grad": (C.c_int, [H, C.c_char_p, FP, C.c_size_t]),
    "alignnet_debug_dropout_uniforms": (C.c_int, [H, C.c_int32, FP, C.c_size_t]),
    "alignnet_debug_knn_graph": (C.c_int, [H, C.POINTER(C.c_int32), C.c_size_t]),
    "alignnet_debug_train_decisions": (C.c_int, [H, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_size_t]),
    "alignnet_debug_train_relu_mask": (C.c_int, [H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_size_t]),
    "alignnet_debug_train_rounded": (C.c_int, [H, C.c_int32, C.c_int32, C.POINTER(C.c_uint16), C.c_size_t]),
    "alignnet_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "alignnet_comm_loopback_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "alignnet_comm_init": (C.c_int, [H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "alignnet_comm_init_grad": (C.c_int, [H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "alignnet_comm_allreduce_grads": (C.c_int, [H]),
    "alignnet_comm_average_shadows": (C.c_int, [H]),
    "alignnet_get_state": (C.c_int, [H, C.POINTER(State)]),
    "alignnet_set_step": (C.c_int, [H, C.c_int64]),
    "alignnet_save": (C.c_int, [H, C.c_char_p]),
    "alignnet_load": (C.c_int, [H, C.c_char_p, C.c_int32]),
    "alignnet_profile_enable": (C.c_int, [H, C.c_int32]),
    "alignnet_dataset_upload": (C.c_int, [H, FP, FP, C.POINTER(C.c_int64), FP, C.c_int64]),
    "alignnet_dataset_free": (C.c_int, [H]),
    "alignnet_dataset_sample": (C.c_int, [H, C.POINTER(C.c_int32), C.c_int32, C.c_uint64, C.c_float, C.c_float]),
    "alignnet_dataset_batch": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(Labels)]),
    "alignnet_train_step_dataset": (C.c_int, [H, C.POINTER(C.c_int32), C.c_int32, C.c_uint64, C.c_float, C.c_float, C.POINTER(StepResult)]),
    "alignnet_forward_dataset": (C.c_int, [H, C.POINTER(C.c_int32), C.c_int32, C.c_uint64, C.POINTER(Outputs)]),
    "alignnet_icp_refine": (C.c_int, [H, FP, FP, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_double), C.c_double, C.c_int32,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "alignnet_icp_refine_dataset": (C.c_int, [H, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double), C.c_double, C.c_int32,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "alignnet_set_option": (C.c_int, [H, C.c_char_p, C.c_int64]),
    "alignnet_get_option": (C.c_int, [H, C.c_char_p, C.POINTER(C.c_int64)]),
    "alignnet_profile_read_kernel": (C.c_int, [H, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "alignnet_profile_read": (C.c_int, [H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int32]),
}

_lib = None


def library_path():
    """In-tree library; ALIGNNET_HIP_LIB names another build of the same sources (A/B kernel comparisons on one box)."""
    return os.environ.get("ALIGNNET_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libalignnet_hip.so")


def load_library():
    """dlopen libalignnet_hip.so and bind every declared symbol.  Raises if the library was
    not built (run `python __graft_entry__.py build` or `make -C alignnet-3d_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `make -C alignnet-3d_amd/csrc` (hipcc, gfx950). "
                      "There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            # an A/B build of an EARLIER revision (ALIGNNET_HIP_LIB, tools/ab_build.sh) may predate a test hook; the in-tree library must export everything
            if os.environ.get("ALIGNNET_HIP_LIB") and name.startswith("alignnet_debug_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.alignnet_abi_version() != ABI_VERSION:
        raise OSError("libalignnet_hip.so ABI version mismatch")
    _lib = lib
    return lib
