"""Streaming post-processing on the device (include/kws.h, csrc/kws_stream.hip).

Mirrors the two helper classes of the reference's listen.py with the same constructor arguments, attributes and
methods -- `ThresholdDecoder` (listen.py:452-522) and `TriggerDetector` (listen.py:525-559) -- and adds `StreamBatch`,
which runs the whole per-chunk loop of listen.py:350-375 (`update_vectors`, predict, argmax / max, decode, detector
update) for S audio streams at once.  All arithmetic runs in the HIP library; there is no host fallback.
"""
import ctypes

import numpy as np

from . import lib as _l
from .featurizer import Featurizer


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("kws_amd.stream needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


def _stream():
    return _torch().cuda.current_stream().cuda_stream


class ThresholdDecoder(object):
    """listen.py:452-522.  `cd`, `min_out`, `max_out`, `out_range`, `center` as in the reference."""

    def __init__(self, mu_stds, center=0.5, resolution=200, min_z=-4, max_z=4):
        _torch()
        self._L = _l.get_lib()
        pairs = np.ascontiguousarray(np.asarray(mu_stds, dtype=np.float64).reshape(-1, 2))
        self._h = ctypes.c_void_p()
        _l.check(self._L.kws_decoder_create(pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), pairs.shape[0], float(center),
                                            int(resolution), float(min_z), float(max_z), ctypes.byref(self._h)))
        mn, rg, n = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0)
        _l.check(self._L.kws_decoder_info(self._h, ctypes.byref(mn), ctypes.byref(rg), ctypes.byref(n)))
        self.min_out, self.out_range, self.max_out = mn.value, rg.value, mn.value + rg.value
        self.center = float(center)
        self.cd = np.empty(n.value, dtype=np.float64)
        _l.check(self._L.kws_decoder_table(self._h, self.cd.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self.cd.size))

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.kws_decoder_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_device(self, raw):
        """raw: CUDA tensor, float32 (live-loop semantics) or float64 (Python-float semantics) -> float64 CUDA tensor."""
        torch = _torch()
        if not raw.is_cuda or raw.dtype not in (torch.float32, torch.float64):
            raise ValueError("raw must be a CUDA float32 or float64 tensor")
        raw = raw.contiguous()
        out = torch.empty(raw.shape, dtype=torch.float64, device=raw.device)
        code = _l.RAW_F32 if raw.dtype == torch.float32 else _l.RAW_F64
        _l.check(self._L.kws_decoder_decode(self._h, raw.data_ptr(), code, out.data_ptr(), raw.numel(), _stream()))
        return out

    def decode(self, raw_output):
        """Scalar or array in, same kind out.  float32 inputs follow the live loop (listen.py:361-367), where numpy
        evaluates 1/x - 1 in float32; Python floats / float64 follow the scalar path."""
        torch = _torch()
        if isinstance(raw_output, torch.Tensor):
            return self.decode_device(raw_output)
        a = np.asarray(raw_output)
        dt = np.float32 if a.dtype == np.float32 else np.float64
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt).reshape(-1)).cuda()
        res = self.decode_device(t).cpu().numpy().reshape(a.shape)
        return float(res) if a.ndim == 0 else res

    def encode(self, threshold):
        out = ctypes.c_double(0.0)
        _l.check(self._L.kws_decoder_encode(self._h, float(threshold), ctypes.byref(out)))
        return out.value


class TriggerDetector(object):
    """listen.py:525-559 for one stream (state lives on the device; `StreamBatch` is the many-stream form)."""

    def __init__(self, chunk_size, class_names, sensitivity=0.5, trigger_level=3):
        torch = _torch()
        self._L = _l.get_lib()
        self.chunk_size = int(chunk_size)
        self.class_names = class_names
        self.sensitivity = sensitivity
        self.trigger_level = trigger_level
        self._background = [i for i, n in enumerate(class_names) if n == 'background']
        self._state = torch.tensor([[0, -1]], dtype=torch.int32, device="cuda")
        self._fired = torch.zeros(1, dtype=torch.int32, device="cuda")

    @property
    def activation(self):
        return int(self._state[0, 0].item())

    @property
    def record_index(self):
        v = int(self._state[0, 1].item())
        return None if v < 0 else v

    def update(self, index, score):
        """Returns whether the new prediction caused an activation"""
        torch = _torch()
        index = int(np.asarray(index).reshape(-1)[0])
        score = float(np.asarray(score).reshape(-1)[0])
        # several names may be 'background' only in a malformed list; the kernel takes one index, -1 = none
        bg = index if index in self._background else (self._background[0] if self._background else -1)
        idx = torch.tensor([index], dtype=torch.int32, device="cuda")
        sc = torch.tensor([score], dtype=torch.float64, device="cuda")
        _l.check(self._L.kws_trigger_update(idx.data_ptr(), sc.data_ptr(), 1, bg, float(self.sensitivity), int(self.trigger_level),
                                            self.chunk_size, self._state.data_ptr(), self._fired.data_ptr(), _stream()))
        return bool(self._fired.item())


class StreamBatch(object):
    """S lock-stepped audio streams through update_vectors -> model -> decode -> trigger (listen.py:96-114, 350-375).

    push(chunks) takes one chunk of int16 PCM per stream -- (S, n) int16 array / CUDA tensor, or a list of S `bytes`
    objects as PyAudio / wave.readframes deliver them -- and returns (index, score, fired) as CUDA tensors of length S.
    """

    def __init__(self, pr, device_model, n_streams, chunk_size=1024, class_names=None, sensitivity=0.5, trigger_level=3,
                 decoder=None, featurizer=None, background_index=0):
        torch = _torch()
        if pr.use_delta:
            # listen.py:111-112 re-applies add_deltas to the whole matrix on every chunk, which doubles its width and
            # makes the next np.concatenate raise; streaming with deltas never worked in the reference
            raise ValueError("streaming with use_delta=True is not usable in the reference (listen.py:111-112) and is not offered")
        self._L = _l.get_lib()
        self.pr, self.model = pr, device_model
        self.S, self.chunk_size = int(n_streams), int(chunk_size)
        self.sensitivity, self.trigger_level = float(sensitivity), int(trigger_level)
        self.background_index = int(background_index)
        if class_names is not None:
            assert class_names[0] == 'background', '1st class should be background.'      # listen.py:66
        self.class_names = class_names
        self.featurizer = featurizer if featurizer is not None else Featurizer(pr)
        self.decoder = decoder if decoder is not None else ThresholdDecoder(pr.threshold_config, pr.threshold_center)
        self.window_samples, self.hop_samples = pr.window_samples, pr.hop_samples
        self.F, self.D = pr.n_features, pr.n_mfcc
        dev = device_model.device
        self.cap = self.window_samples + self.chunk_size
        self.win = torch.zeros((self.S, self.cap), dtype=torch.int16, device=dev)         # carried + new samples
        self.n_win = 0
        self.mfccs = torch.zeros((self.S, self.F, self.D), dtype=torch.float32, device=dev)
        self.state = torch.zeros((self.S, 2), dtype=torch.int32, device=dev)
        self.state[:, 1] = -1
        self.index = torch.zeros(self.S, dtype=torch.int32, device=dev)
        self.score = torch.zeros(self.S, dtype=torch.float64, device=dev)
        self.fired = torch.zeros(self.S, dtype=torch.int32, device=dev)
        self.probs = None

    def _chunk_tensor(self, chunks):
        torch = _torch()
        if isinstance(chunks, torch.Tensor):
            t = chunks
        else:
            if isinstance(chunks, (list, tuple)) and chunks and isinstance(chunks[0], (bytes, bytearray, memoryview)):
                a = np.stack([np.frombuffer(c, dtype='<i2') for c in chunks])             # buffer_to_audio's view, data_utils.py:19-21
            else:
                a = np.asarray(chunks)
            if a.dtype != np.int16:
                raise ValueError("chunks must be int16 PCM (the 1/32768 scaling of buffer_to_audio happens on the device)")
            t = torch.from_numpy(np.ascontiguousarray(a))
        if t.dim() != 2 or t.shape[0] != self.S or t.dtype != torch.int16:
            raise ValueError("expected %d int16 chunks of equal length" % self.S)
        if t.shape[1] > self.chunk_size:
            raise ValueError("chunk of %d samples exceeds chunk_size=%d" % (t.shape[1], self.chunk_size))
        return t.to(self.win.device, non_blocking=True)

    def update_vectors(self, chunks):
        """listen.py:96-114 for all streams; returns the (S, n_features, n_mfcc) feature tensor (device, updated in place)."""
        t = self._chunk_tensor(chunks)
        n = t.shape[1]
        self.win[:, self.n_win:self.n_win + n] = t
        self.n_win += n
        if self.n_win >= self.window_samples:
            rows = self.featurizer.raw(self.win, n_samples=self.n_win)                    # (S, n_new, D)
            n_new = rows.shape[1]
            _l.check(self._L.kws_stream_push_rows(self.mfccs.data_ptr(), rows.data_ptr(), self.S, self.F, self.D, n_new, _stream()))
            used = n_new * self.hop_samples
            keep = self.n_win - used
            if keep > 0:
                self.win[:, :keep] = self.win[:, used:self.n_win].clone()
            self.n_win = keep
        return self.mfccs

    def push(self, chunks):
        """One step of the loop listen.py:350-375 for every stream."""
        feats = self.update_vectors(chunks)
        self.probs, _ = self.model.forward(feats, want_probs=True, want_argmax=False)
        _l.check(self._L.kws_stream_postprocess(self.decoder.handle, self.probs.data_ptr(), self.S, self.probs.shape[1],
                                                self.background_index, self.sensitivity, self.trigger_level, self.chunk_size,
                                                self.state.data_ptr(), self.index.data_ptr(), self.score.data_ptr(),
                                                self.fired.data_ptr(), _stream()))
        return self.index, self.score, self.fired
