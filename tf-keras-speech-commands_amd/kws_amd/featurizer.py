"""Device featurizer: a thin owner of a `kws_featurizer` handle (include/kws.h)."""
import ctypes

import numpy as np

from . import lib as _l


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _l.KwsError(-3, "no HIP device visible to torch: the featurizer has no CPU fallback")
    return torch


def params_struct(pr):
    return _l.KwsParams(float(pr.buffer_t), float(pr.window_t), float(pr.hop_t), int(pr.sample_rate),
                        int(pr.sample_depth), int(pr.n_fft), int(pr.n_filt), int(pr.n_mfcc), int(bool(pr.use_delta)))


def params_key(pr):
    return (float(pr.buffer_t), float(pr.window_t), float(pr.hop_t), int(pr.sample_rate), int(pr.sample_depth),
            int(pr.n_fft), int(pr.n_filt), int(pr.n_mfcc), bool(pr.use_delta))


def derive_geometry(pr):
    """classifier/params.py derived properties computed by the C ABI (host only, no GPU needed)."""
    g = _l.KwsGeometry()
    p = params_struct(pr)
    _l.check(_l.get_lib().kws_params_derive(ctypes.byref(p), ctypes.byref(g)))
    return g.as_dict()


class Featurizer(object):
    """Batched waveform -> (B, n_features, feature_size) features on the current HIP device."""

    def __init__(self, pr, bank="mel"):
        self._L = _l.get_lib()
        self._h = ctypes.c_void_p()
        self.bank_kind = {"mel": _l.BANK_MEL, "bark": _l.BANK_BARK}[bank]
        p = params_struct(pr)
        _l.check(self._L.kws_featurizer_create(ctypes.byref(p), self.bank_kind, ctypes.byref(self._h)))
        g = _l.KwsGeometry()
        _l.check(self._L.kws_featurizer_geometry(self._h, ctypes.byref(g)))
        self.geometry = g.as_dict()
        self.n_filt, self.n_fft, self.n_mfcc = int(pr.n_filt), int(pr.n_fft), int(pr.n_mfcc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.kws_featurizer_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cu_share(self, blocks_per_cu):
        """2 (default): a launch may fill every CU's LDS (fastest alone); 1: half of it, leaving room for kernels of other
        streams (featurizing beside a train step: kws_amd.pipeline.FeaturePipeline sets this)."""
        _l.check(self._L.kws_featurizer_set_cu_share(self._h, int(blocks_per_cu)))

    def occupancy(self):
        """(resident clips per compute unit, LDS bytes per clip) the HIP runtime reports for this featurizer's kernel."""
        nb, lds = ctypes.c_int(0), ctypes.c_size_t(0)
        _l.check(self._L.kws_featurizer_occupancy(self._h, ctypes.byref(nb), ctypes.byref(lds)))
        return nb.value, lds.value

    def bank(self):
        out = np.zeros((self.n_filt, self.n_fft // 2 + 1), np.float32)
        _l.check(self._L.kws_featurizer_bank(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), out.size))
        return out

    @staticmethod
    def _dtype_code(t):
        torch = _torch()
        if t.dtype == torch.float32:
            return _l.WAV_F32
        if t.dtype == torch.int16:
            return _l.WAV_I16
        raise TypeError("waveforms must be float32 or int16, got %s" % t.dtype)

    def __call__(self, wav, valid_len=None, out=None, index=None):
        """wav: CUDA tensor (rows, stride) float32|int16; valid_len: optional CUDA int32 (rows,).  index: optional CUDA int32 (B,): featurize
        the B rows wav[index[b]] in place of all rows (kws_featurize_gather: a shuffled minibatch of a device-resident dataset, no copy)."""
        torch = _torch()
        if not wav.is_cuda or wav.dim() != 2 or not wav.is_contiguous():
            raise ValueError("wav must be a contiguous CUDA tensor of shape (B, stride)")
        rows, stride = wav.shape
        B = rows
        ix = 0
        if index is not None:
            if index.dtype != torch.int32 or not index.is_cuda or index.dim() != 1 or not index.is_contiguous():
                raise ValueError("index must be a contiguous CUDA int32 vector")
            B, ix = index.numel(), index.data_ptr()
        g = self.geometry
        if out is None:
            out = torch.empty((B, g["n_features"], g["feature_size"]), dtype=torch.float32, device=wav.device)
        elif out.numel() < B * g["n_features"] * g["feature_size"] or not out.is_contiguous():
            raise ValueError("out is too small for %d clips" % B)
        vl = 0
        if valid_len is not None:
            if valid_len.dtype != torch.int32 or not valid_len.is_cuda or valid_len.numel() != rows:
                raise ValueError("valid_len must be a CUDA int32 tensor with one element per row of wav")
            vl = valid_len.data_ptr()
        _l.check(self._L.kws_featurize_gather(self._h, wav.data_ptr(), self._dtype_code(wav), ix, B, stride, vl, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
        return out

    def raw(self, wav, n_samples=None):
        """vectorize_raw semantics: (B, n) -> (B, n_frames, n_mfcc), no padding / clipping / deltas."""
        torch = _torch()
        if not wav.is_cuda or wav.dim() != 2 or not wav.is_contiguous():
            raise ValueError("wav must be a contiguous CUDA tensor of shape (B, n)")
        B, stride = wav.shape
        n = stride if n_samples is None else int(n_samples)
        nf = self._L.kws_featurize_raw_frames(self._h, n)
        out = torch.empty((B, nf, self.n_mfcc), dtype=torch.float32, device=wav.device)
        if nf:
            _l.check(self._L.kws_featurize_raw(self._h, wav.data_ptr(), self._dtype_code(wav), B, stride, n,
                                               out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out
