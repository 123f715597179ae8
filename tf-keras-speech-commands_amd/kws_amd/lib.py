"""ctypes bindings of include/kws.h."""
import ctypes
import os

from .build import LIB_PATH


ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_WORKSPACE, ERR_COMM = -1, -2, -3, -4, -5, -6   # include/kws.h


class KwsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("kws error %d: %s" % (code, message))
        self.code = code


class KwsParams(ctypes.Structure):
    _fields_ = [("buffer_t", ctypes.c_double), ("window_t", ctypes.c_double), ("hop_t", ctypes.c_double),
                ("sample_rate", ctypes.c_int32), ("sample_depth", ctypes.c_int32), ("n_fft", ctypes.c_int32),
                ("n_filt", ctypes.c_int32), ("n_mfcc", ctypes.c_int32), ("use_delta", ctypes.c_int32)]


class KwsGeometry(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("window_samples", "hop_samples", "max_samples", "buffer_samples", "n_features", "feature_size")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KwsTensorInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int32 * 4),
                ("trainable", ctypes.c_int32), ("offset", ctypes.c_int64), ("size", ctypes.c_int64)]


OVERLAP_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p)     # kws_train_args.overlap_callback


class KwsTrainArgs(ctypes.Structure):
    _fields_ = [("feat", ctypes.c_void_p), ("labels", ctypes.c_void_p), ("class_weights", ctypes.c_void_p),
                ("B", ctypes.c_int32), ("ignore_index", ctypes.c_int32), ("params", ctypes.c_void_p),
                ("state", ctypes.c_void_p), ("grads", ctypes.c_void_p), ("ws", ctypes.c_void_p),
                ("ws_bytes", ctypes.c_size_t), ("dropout_seed", ctypes.c_uint64), ("grad_scale", ctypes.c_float),
                ("probs", ctypes.c_void_p), ("stats", ctypes.c_void_p), ("bucket_event", ctypes.c_void_p),
                ("forward_event", ctypes.c_void_p), ("overlap_event", ctypes.c_void_p),
                ("overlap_callback", OVERLAP_CB), ("overlap_user", ctypes.c_void_p), ("comm", ctypes.c_void_p),
                ("comm_state_weight", ctypes.c_float), ("feat_moments", ctypes.c_void_p)]


MODEL_KINDS = {"simple_cnn": 0, "simple_cnn_lite": 1, "simple_gru": 2, "simple_lstm": 3}
BANK_MEL, BANK_BARK = 0, 1
WAV_F32, WAV_I16 = 0, 1
RAW_F64, RAW_F32 = 0, 1

_lib = None


def get_lib():
    """Load libkws_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libkws_hip.so is missing at %s: build it with `python -m kws_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and the host side of this package uses torch for device memory and
    # streams, so torch is loaded FIRST and libkws_hip.so binds to the runtime it brought.  Loaded the other way round (this library, then
    # torch) the process holds two runtimes and the second one sees no device (__graft_entry__: build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass                    # a torch-free host (tests of the C ABI alone): the system runtime the library links against
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    L.kws_version.restype = ctypes.c_char_p
    L.kws_last_error.restype = ctypes.c_char_p
    L.kws_build_id.restype = ctypes.c_char_p
    L.kws_device_count.restype = i32
    L.kws_params_default.argtypes = [ctypes.POINTER(KwsParams)]
    L.kws_params_default.restype = None
    L.kws_params_derive.argtypes = [ctypes.POINTER(KwsParams), ctypes.POINTER(KwsGeometry)]
    L.kws_featurizer_create.argtypes = [ctypes.POINTER(KwsParams), i32, ctypes.POINTER(vp)]
    L.kws_featurizer_destroy.argtypes = [vp]
    L.kws_featurizer_destroy.restype = None
    L.kws_featurizer_geometry.argtypes = [vp, ctypes.POINTER(KwsGeometry)]
    L.kws_featurizer_bank.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
    L.kws_featurize.argtypes = [vp, vp, i32, i32, i64, vp, fp, vp]
    L.kws_featurize_gather.argtypes = [vp, vp, i32, vp, i32, i64, vp, fp, vp]
    L.kws_featurizer_set_cu_share.argtypes = [vp, i32]
    L.kws_featurize_raw.argtypes = [vp, vp, i32, i32, i64, i32, fp, vp]
    u64, f32 = ctypes.c_uint64, ctypes.c_float
    L.kws_model_create.argtypes = [i32, i32, i32, i32, ctypes.POINTER(vp)]
    L.kws_model_destroy.argtypes = [vp]
    L.kws_model_destroy.restype = None
    for n in ("kws_model_param_count", "kws_model_state_count"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = i64
    L.kws_model_num_tensors.argtypes = [vp]
    L.kws_model_tensor_info.argtypes = [vp, i32, ctypes.POINTER(KwsTensorInfo)]
    L.kws_model_workspace_bytes.argtypes = [vp, i32, i32]
    L.kws_model_workspace_bytes.restype = i64
    L.kws_model_forward.argtypes = [vp, vp, i32, vp, vp, vp, ctypes.c_size_t, vp, vp, vp]
    L.kws_model_train_fwd_bwd.argtypes = [vp, ctypes.POINTER(KwsTrainArgs), vp]
    L.kws_model_bind_device.argtypes = [vp]
    L.kws_model_prepare_inference.argtypes = [vp, i32, vp, vp, vp, ctypes.c_size_t, vp]
    L.kws_model_invalidate_prepared.argtypes = [vp]
    L.kws_feature_moments_workspace_bytes.argtypes = [i32]
    L.kws_feature_moments_workspace_bytes.restype = i64
    L.kws_feature_moments.argtypes = [vp, i32, i32, i32, vp, vp, ctypes.c_size_t, vp]
    L.kws_model_grad_split.argtypes = [vp]
    L.kws_model_grad_split.restype = i64
    L.kws_loss_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.kws_confusion_counts.argtypes = [vp, vp, i32, i32, vp, vp]
    L.kws_sgd_step.argtypes = [vp, vp, i64, f32, f32, vp]
    L.kws_rmsprop_step.argtypes = [vp, vp, vp, i64, f32, f32, f32, f32, vp]
    L.kws_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, f32, vp]
    f64 = ctypes.c_double
    L.kws_featurizer_occupancy.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_size_t)]
    L.kws_decoder_create.argtypes = [ctypes.POINTER(f64), i32, f64, i32, f64, f64, ctypes.POINTER(vp)]
    L.kws_decoder_destroy.argtypes = [vp]
    L.kws_decoder_destroy.restype = None
    L.kws_decoder_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(i64)]
    L.kws_decoder_table.argtypes = [vp, ctypes.POINTER(f64), ctypes.c_size_t]
    L.kws_decoder_decode.argtypes = [vp, vp, i32, vp, i64, vp]
    L.kws_decoder_encode.argtypes = [vp, f64, ctypes.POINTER(f64)]
    L.kws_stream_push_rows.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.kws_trigger_update.argtypes = [vp, vp, i32, i32, f64, i32, i32, vp, vp, vp]
    L.kws_stream_postprocess.argtypes = [vp, vp, i32, i32, i32, f64, i32, i32, vp, vp, vp, vp, vp]
    L.kws_set_matrix_precision.argtypes = [i32]
    L.kws_get_matrix_precision.restype = i32
    L.kws_set_inference_precision.argtypes = [i32]
    L.kws_get_inference_precision.restype = i32
    L.kws_model_set_precision.argtypes = [vp, i32, i32]
    L.kws_model_get_precision.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.kws_model_set_deterministic.argtypes = [vp, i32]
    L.kws_model_set_overlap_point.argtypes = [vp, i32]
    L.kws_comm_unique_id.argtypes = [vp]
    L.kws_comm_init.argtypes = [i32, i32, vp, ctypes.POINTER(vp)]
    L.kws_comm_destroy.argtypes = [vp]
    L.kws_comm_destroy.restype = None
    L.kws_comm_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.kws_allreduce_grads.argtypes = [vp, vp, i64, i64, vp, i64, f32, vp]
    L.kws_comm_allreduce.argtypes = [vp, vp, i64, i32, i32, vp]
    L.kws_comm_broadcast.argtypes = [vp, vp, i64, i32, vp]
    L.kws_comm_timing.argtypes = [vp, i32]
    L.kws_comm_last_us.argtypes = [vp, ctypes.POINTER(f32), ctypes.POINTER(f32)]
    L.kws_prof_enable.argtypes = [i32]
    L.kws_prof_report.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.kws_prof_report.restype = i64
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise KwsError(rc, get_lib().kws_last_error().decode("utf-8", "replace"))


def version():
    return get_lib().kws_version().decode()


def build_id():
    """{source file: sha1 prefix} the loaded library was built from"""
    out = {}
    for item in get_lib().kws_build_id().decode().split(";"):
        if item:
            k, v = item.split(":")
            out[k] = v
    return out


def device_count():
    return get_lib().kws_device_count()


MATRIX_FP32, MATRIX_BF16X6 = 0, 1
INFER_FP32, INFER_FP16 = 0, 1
FEATURE_MOMENTS = 100
COMM_ID_BYTES = 128
DT_F32, DT_F64, DT_I32, DT_I64 = 0, 1, 2, 3
OP_SUM, OP_MAX, OP_AVG = 0, 1, 2


def set_matrix_precision(mode):
    """Library-wide DEFAULT (a model follows it until DeviceModel.set_precision gives it its own): MATRIX_BF16X6
    (three-way bf16 split on the matrix cores, fp32-level error) or MATRIX_FP32 (exact fp32 MFMA)."""
    check(get_lib().kws_set_matrix_precision(int(mode)))


def get_matrix_precision():
    return get_lib().kws_get_matrix_precision()


def set_inference_precision(mode):
    """INFER_FP32 (default) or INFER_FP16: simple_cnn_lite inference with fp16 activations and matrix operands, fp32
    accumulation (BASELINE configs[4]); other model kinds ignore the switch."""
    check(get_lib().kws_set_inference_precision(int(mode)))


def get_inference_precision():
    return get_lib().kws_get_inference_precision()


def prof_enable(on=True):
    check(get_lib().kws_prof_enable(1 if on else 0))


def prof_report():
    """{kernel name: {"count": n, "total_ms": t}} for the launches since prof_enable(True)"""
    import json
    L = get_lib()
    n = L.kws_prof_report(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    L.kws_prof_report(buf, len(buf))
    return json.loads(buf.value.decode())
