"""oracle/featurizer_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes access to oracle/libkws_oracle.so (the C restatement, kws_oracle.c) and,
where it was built, to oracle/_ref/libmfcc_ref.so (the reference's own
inference/tflite/mfcc.h compiled from /root/reference).  Also a small
independent numpy restatement (`numpy_mfcc`) used to cross-check the C one.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
import this module.  Pinning status: see kws_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULT_PARAMS = dict(buffer_t=1.0, window_t=0.064, hop_t=0.032, sample_rate=16000, sample_depth=2,
                      n_fft=1024, n_filt=20, n_mfcc=20, use_delta=False)  # classifier/params.py:99-103


class OracleParams(ctypes.Structure):
    _fields_ = [("buffer_t", ctypes.c_double), ("window_t", ctypes.c_double), ("hop_t", ctypes.c_double),
                ("sample_rate", ctypes.c_int), ("sample_depth", ctypes.c_int), ("n_fft", ctypes.c_int),
                ("n_filt", ctypes.c_int), ("n_mfcc", ctypes.c_int), ("use_delta", ctypes.c_int)]


def make_params(**kw):
    d = dict(DEFAULT_PARAMS)
    d.update(kw)
    return OracleParams(d["buffer_t"], d["window_t"], d["hop_t"], int(d["sample_rate"]), int(d["sample_depth"]),
                        int(d["n_fft"]), int(d["n_filt"]), int(d["n_mfcc"]), int(bool(d["use_delta"])))


def build(force=False):
    """Compile the checkers (gcc/g++ via oracle/Makefile).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libkws_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "kws_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libkws_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "libkws_oracle.so"))
        P = ctypes.POINTER(OracleParams)
        dp = ctypes.POINTER(ctypes.c_double)
        for name in ("window_samples", "hop_samples", "max_samples", "buffer_samples", "n_features", "feature_size"):
            f = getattr(L, "oracle_" + name)
            f.argtypes, f.restype = [P], ctypes.c_int
        L.oracle_mel_points.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)]
        L.oracle_mel_bank.argtypes = [ctypes.c_int] * 3 + [dp]
        L.oracle_bark_bank.argtypes = [ctypes.c_int] * 3 + [dp]
        L.oracle_power_spec.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
        L.oracle_mfcc_spec.argtypes = [dp, ctypes.c_int, P, ctypes.c_int, dp]
        L.oracle_audio_to_feature.argtypes = [dp, ctypes.c_int, P, ctypes.c_int, dp]
        L.oracle_featurize_batch_f32.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_int), P, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_float)]
        _lib = L
    return _lib


def ref_lib():
    """The compiled reference (oracle/_ref), or None when it was not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libmfcc_ref.so")
        if not os.path.exists(path):
            return None
        R = ctypes.CDLL(path)
        R.ref_mfcc_f32.argtypes = [ctypes.POINTER(ctypes.c_float)] + [ctypes.c_int] * 10 + [ctypes.POINTER(ctypes.c_float)]
        R.ref_mfcc_f64.argtypes = [ctypes.POINTER(ctypes.c_double)] + [ctypes.c_int] * 10 + [ctypes.POINTER(ctypes.c_double)]
        _ref = R
    return _ref


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def geometry(**kw):
    p = make_params(**kw)
    L = lib()
    return {k: getattr(L, "oracle_" + k)(ctypes.byref(p)) for k in
            ("window_samples", "hop_samples", "max_samples", "buffer_samples", "n_features", "feature_size")}


def mel_points(sample_rate=16000, n_fft=1024, n_filt=20):
    pts = (ctypes.c_int * (n_filt + 2))()
    rc = lib().oracle_mel_points(sample_rate, n_fft, n_filt, pts)
    if rc:
        raise ValueError("mel grid not strictly increasing")
    return list(pts)


def bank(kind="mel", sample_rate=16000, n_fft=1024, n_filt=20):
    out = np.zeros((n_filt, n_fft // 2 + 1), np.float64)
    f = lib().oracle_mel_bank if kind == "mel" else lib().oracle_bark_bank
    rc = f(sample_rate, n_fft, n_filt, _dptr(out))
    if rc:
        raise ValueError("cannot build %s bank (rc=%d)" % (kind, rc))
    return out


def power_spec(audio, window, hop, n_fft):
    a = np.ascontiguousarray(audio, np.float64)
    n = lib().oracle_power_spec(_dptr(a), len(a), window, hop, n_fft, None)
    out = np.zeros((n, n_fft // 2 + 1), np.float64)
    lib().oracle_power_spec(_dptr(a), len(a), window, hop, n_fft, _dptr(out))
    return out


def mfcc_spec(audio, kind="mel", **kw):
    """sonopy.mfcc_spec as vectorize_raw calls it (common/data_utils.py:61-70)."""
    p = make_params(**kw)
    a = np.ascontiguousarray(audio, np.float64)
    g = geometry(**kw)
    n_frames = max(0, (len(a) - g["window_samples"]) // g["hop_samples"] + 1) if len(a) >= g["window_samples"] else 0
    n_out = min(p.n_filt, p.n_mfcc)
    out = np.zeros((n_frames, n_out), np.float64)
    rc = lib().oracle_mfcc_spec(_dptr(a), len(a), ctypes.byref(p), 0 if kind == "mel" else 1, _dptr(out))
    if rc < 0:
        raise ValueError("oracle_mfcc_spec rc=%d" % rc)
    return out


def audio_to_feature(audio, kind="mel", **kw):
    """common/data_utils.py:73-86 (truncate head / left-pad / mfcc / optional deltas)."""
    p = make_params(**kw)
    a = np.ascontiguousarray(audio, np.float64)
    g = geometry(**kw)
    n_frames = (g["max_samples"] - g["window_samples"]) // g["hop_samples"] + 1
    out = np.zeros((n_frames, g["feature_size"]), np.float64)
    rc = lib().oracle_audio_to_feature(_dptr(a), len(a), ctypes.byref(p), 0 if kind == "mel" else 1, _dptr(out))
    if rc < 0:
        raise ValueError("oracle_audio_to_feature rc=%d" % rc)
    return out


def featurize_batch(wav, valid_len=None, kind="mel", **kw):
    """(B, stride) float32 -> (B, n_features, feature_size) float32; threads = OMP_NUM_THREADS."""
    p = make_params(**kw)
    g = geometry(**kw)
    w = np.ascontiguousarray(wav, np.float32)
    B, stride = w.shape
    out = np.zeros((B, g["n_features"], g["feature_size"]), np.float32)
    vl = None
    if valid_len is not None:
        vl_arr = np.ascontiguousarray(valid_len, np.int32)
        vl = vl_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rc = lib().oracle_featurize_batch_f32(w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, stride, vl,
                                          ctypes.byref(p), 0 if kind == "mel" else 1,
                                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc:
        raise ValueError("oracle_featurize_batch_f32 rc=%d" % rc)
    return out


def ref_mfcc(audio, dtype=np.float32, **kw):
    """Run the compiled REFERENCE mfcc.h exactly as speech_commands.h:293-316 drives it."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libmfcc_ref.so not built (reference absent)")
    d = dict(DEFAULT_PARAMS)
    d.update(kw)
    g = geometry(**kw)
    a = np.ascontiguousarray(audio, dtype)
    n_frames = (len(a) - g["window_samples"]) // g["hop_samples"] + 1
    out = np.zeros((n_frames, g["feature_size"]), dtype)
    args = [len(a), d["sample_rate"], g["window_samples"], g["hop_samples"], d["n_fft"], d["n_mfcc"], d["n_filt"],
            0, d["sample_rate"], int(bool(d["use_delta"]))]
    if dtype == np.float32:
        n = R.ref_mfcc_f32(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), *args,
                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    else:
        n = R.ref_mfcc_f64(_dptr(a), *args, _dptr(out))
    assert n == n_frames
    return out


# ---------------------------------------------------------------------------
# Independent numpy restatement (vectorised), used only to cross-check the C.
# Follows sonopy.mfcc_spec as corroborated by common/bark_feature.py:75-89,
# 156-175 and inference/tflite/mfcc.h:230-264.
# ---------------------------------------------------------------------------
def numpy_mel_bank(sample_rate, n_fft, n_filt):
    n_bins = n_fft // 2 + 1
    mels = np.linspace(1127.0 * np.log(1 + 0 / 700.0), 1127.0 * np.log(1 + sample_rate / 700.0), n_filt + 2, True)
    hz = 700.0 * (np.exp(mels / 1127.0) - 1)
    idx = (hz * n_bins / sample_rate).astype(int)
    banks = np.zeros((n_filt, n_bins))
    for i in range(n_filt):
        l, m, r = idx[i], idx[i + 1], idx[i + 2]
        banks[i, l:m] = np.linspace(0.0, 1.0, m - l, False)
        banks[i, m:r] = np.linspace(1.0, 0.0, r - m, False)
    return banks


def numpy_mfcc(audio, bank_matrix=None, **kw):
    from scipy.fftpack import dct
    d = dict(DEFAULT_PARAMS)
    d.update(kw)
    window = int(d["sample_rate"] * d["window_t"] + 0.5)
    hop = int(d["sample_rate"] * d["hop_t"] + 0.5)
    a = np.asarray(audio, np.float64)
    frames = [a[i - window:i] for i in range(window, len(a) + 1, hop)]
    if not frames:
        return np.empty((0, min(d["n_filt"], d["n_mfcc"])))
    fft = np.fft.rfft(np.asarray(frames), n=d["n_fft"])
    powers = (fft.real ** 2 + fft.imag ** 2) / d["n_fft"]
    if bank_matrix is None:
        bank_matrix = numpy_mel_bank(d["sample_rate"], d["n_fft"], d["n_filt"])
    eps = np.finfo(float).eps
    mels = np.log(np.clip(powers @ bank_matrix.T, eps, None))
    out = dct(mels, norm="ortho")[:, :d["n_mfcc"]]
    out[:, 0] = np.log(np.clip(powers.sum(1), eps, None))
    return out
