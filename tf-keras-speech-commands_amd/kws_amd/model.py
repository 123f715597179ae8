"""Device model: owner of a `kws_model` handle plus the flat device buffers it works on (include/kws.h).

torch is used for device memory and the stream only; every computation goes through the C ABI."""
import ctypes

import numpy as np

from . import lib as _l


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _l.KwsError(-3, "no HIP device visible to torch: the model has no CPU fallback")
    return torch


class ModelSpec(object):
    """Host-only descriptor (no GPU needed): tensor table in Keras get_weights() order."""

    def __init__(self, model_type, num_classes, n_features, feature_size):
        if model_type not in _l.MODEL_KINDS:
            raise ValueError('Unsupported model type')            # classifier/model.py:32
        self._L = _l.get_lib()
        self._h = ctypes.c_void_p()
        self.model_type, self.num_classes = model_type, int(num_classes)
        self.n_features, self.feature_size = int(n_features), int(feature_size)
        _l.check(self._L.kws_model_create(_l.MODEL_KINDS[model_type], self.num_classes, self.n_features,
                                          self.feature_size, ctypes.byref(self._h)))
        self.param_count = int(self._L.kws_model_param_count(self._h))
        self.state_count = int(self._L.kws_model_state_count(self._h))
        self.tensors = []
        info = _l.KwsTensorInfo()
        for i in range(self._L.kws_model_num_tensors(self._h)):
            _l.check(self._L.kws_model_tensor_info(self._h, i, ctypes.byref(info)))
            self.tensors.append(dict(name=info.name.decode(), shape=tuple(info.shape[:info.ndim]),
                                     trainable=bool(info.trainable), offset=int(info.offset), size=int(info.size)))

    @property
    def handle(self):
        return self._h

    def trainable_count(self):
        return sum(t["size"] for t in self.tensors if t["trainable"])

    def non_trainable_count(self):
        return sum(t["size"] for t in self.tensors if not t["trainable"])

    def workspace_bytes(self, batch, training):
        return int(self._L.kws_model_workspace_bytes(self._h, int(batch), int(bool(training))))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.kws_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FeatureMoments(object):
    """kws_feature_moments: the 10 x 10 second-moment matrix of a feature batch (include/kws.h), on the current stream."""

    def __init__(self, n_features, feature_size, device=None):
        torch = _torch()
        self._L = _l.get_lib()
        self.n_features, self.feature_size = int(n_features), int(feature_size)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        need = int(self._L.kws_feature_moments_workspace_bytes(1))
        self._ws = torch.empty((need + 256,), dtype=torch.uint8, device=dev)
        self._dev = dev

    def __call__(self, feat, out=None):
        torch = _torch()
        n = self.n_features * self.feature_size
        B = feat.numel() // n
        if out is None:
            out = torch.empty((_l.FEATURE_MOMENTS,), dtype=torch.float64, device=self._dev)
        base = self._ws.data_ptr()
        aligned = (base + 255) & ~255
        _l.check(self._L.kws_feature_moments(feat.data_ptr(), B, self.n_features, self.feature_size, out.data_ptr(), aligned,
                                             self._ws.numel() - (aligned - base), torch.cuda.current_stream().cuda_stream))
        return out


class DeviceModel(object):
    """Flat parameter / state / gradient / Adam buffers on the current HIP device + the compute entry points."""

    def __init__(self, spec, device=None):
        torch = _torch()
        self.spec = spec
        self._L = _l.get_lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        z = lambda n: torch.zeros((max(int(n), 4),), dtype=torch.float32, device=self.device)
        self.params, self.state = z(spec.param_count), z(spec.state_count)
        self.grads, self.adam_m, self.adam_v = z(spec.param_count), z(spec.param_count), z(spec.param_count)
        self.stats = torch.zeros((2,), dtype=torch.float32, device=self.device)
        self.step_count = 0
        # the model's side stream exists from here on: before any communicator / pipeline creates streams of its own (kws_model_bind_device)
        _l.check(self._L.kws_model_bind_device(spec.handle))
        self._matrix = self._infer = None      # per-model precision attributes (None: library default)
        self.weights_version = 0               # bumped whenever params / state may have changed (invalidate_prepared)
        self._ws = None
        self._ws_key = None

    # ---- weights ---------------------------------------------------------------------------------------------
    def set_weights(self, weights):
        """weights: list of arrays in Keras get_weights() order."""
        torch = _torch()
        if len(weights) != len(self.spec.tensors):
            raise ValueError("expected %d weight arrays, got %d" % (len(self.spec.tensors), len(weights)))
        p = np.zeros((self.params.numel(),), np.float32)
        s = np.zeros((self.state.numel(),), np.float32)
        for t, w in zip(self.spec.tensors, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != t["shape"]:
                raise ValueError("%s: expected shape %s, got %s" % (t["name"], t["shape"], w.shape))
            (p if t["trainable"] else s)[t["offset"]:t["offset"] + t["size"]] = w.reshape(-1)
        self.params.copy_(torch.from_numpy(p))
        self.state.copy_(torch.from_numpy(s))
        self.invalidate_prepared()

    def _split(self, flat_trainable, flat_state):
        out = []
        for t in self.spec.tensors:
            src = flat_trainable if t["trainable"] else flat_state
            out.append(src[t["offset"]:t["offset"] + t["size"]].reshape(t["shape"]).copy())
        return out

    def get_weights(self):
        return self._split(self.params.cpu().numpy(), self.state.cpu().numpy())

    def get_grads(self):
        """gradients of the trainable tensors, Keras trainable_weights order"""
        g = self.grads.cpu().numpy()
        return [g[t["offset"]:t["offset"] + t["size"]].reshape(t["shape"]).copy() for t in self.spec.tensors if t["trainable"]]

    # ---- per-model switches ----------------------------------------------------------------------------------
    def set_precision(self, matrix="keep", infer="keep"):
        """This model's own arithmetic (kws_model_set_precision): matrix in {MATRIX_FP32, MATRIX_BF16X6}, infer in
        {INFER_FP32, INFER_FP16}; None = follow the library-wide default again; "keep" leaves the attribute as it is."""
        if matrix != "keep":
            self._matrix = None if matrix is None else int(matrix)
        if infer != "keep":
            self._infer = None if infer is None else int(infer)
        _l.check(self._L.kws_model_set_precision(self.spec.handle, -1 if self._matrix is None else self._matrix,
                                                 -1 if self._infer is None else self._infer))

    def get_precision(self):
        """effective (matrix, infer) precision of this model"""
        m, i = ctypes.c_int(), ctypes.c_int()
        _l.check(self._L.kws_model_get_precision(self.spec.handle, ctypes.byref(m), ctypes.byref(i)))
        return m.value, i.value

    def set_deterministic(self, on=True):
        """fixed-order weight-gradient reductions (bit-identical gradients run to run; for parity tests, slow)"""
        _l.check(self._L.kws_model_set_deterministic(self.spec.handle, 1 if on else 0))

    def set_overlap_point(self, point=-1):
        """where the simple_cnn train step records overlap_event / calls overlap_callback (kws_model_set_overlap_point; scheduling only)"""
        _l.check(self._L.kws_model_set_overlap_point(self.spec.handle, int(point)))

    # ---- compute ---------------------------------------------------------------------------------------------
    def new_workspace(self, batch, training=False):
        """a private workspace tensor for `batch` (InferenceSession keeps one, so that the tables prepared in it are not shared
        with any other caller of this model)"""
        torch = _torch()
        return torch.empty((self.spec.workspace_bytes(batch, training) + 256,), dtype=torch.uint8, device=self.device)

    @staticmethod
    def _aligned(buf):
        base = buf.data_ptr()
        aligned = (base + 255) & ~255
        return aligned, buf.numel() - (aligned - base)

    def _workspace(self, batch, training, workspace=None):
        torch = _torch()
        if workspace is not None:
            return self._aligned(workspace)
        need = self.spec.workspace_bytes(batch, training)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need + 256,), dtype=torch.uint8, device=self.device)
        base = self._ws.data_ptr()
        aligned = (base + 255) & ~255
        return aligned, self._ws.numel() - (aligned - base)

    def _check_feat(self, feat):
        torch = _torch()
        n = self.spec.n_features * self.spec.feature_size
        if not feat.is_cuda or feat.dtype != torch.float32 or not feat.is_contiguous() or feat.numel() % n:
            raise ValueError("features must be a contiguous float32 CUDA tensor of shape (B, %d, %d[, 1])"
                             % (self.spec.n_features, self.spec.feature_size))
        return feat.numel() // n

    def prepare_inference(self, batch, workspace=None):
        """kws_model_prepare_inference for `batch` clips: later forward() calls of that batch size skip the weight-derived work
        until the weights change (set_weights and the optimizer steps of this object drop the prepared state; whoever writes
        self.params / self.state directly must call invalidate_prepared())."""
        torch = _torch()
        ws, nbytes = self._workspace(int(batch), False, workspace)
        _l.check(self._L.kws_model_prepare_inference(self.spec.handle, int(batch), self.params.data_ptr(), self.state.data_ptr(), ws, nbytes,
                                                     torch.cuda.current_stream().cuda_stream))

    def invalidate_prepared(self):
        self.weights_version += 1
        _l.check(self._L.kws_model_invalidate_prepared(self.spec.handle))

    def forward(self, feat, want_probs=True, want_argmax=True, workspace=None):
        torch = _torch()
        B = self._check_feat(feat)
        ws, nbytes = self._workspace(B, False, workspace)
        probs = torch.empty((B, self.spec.num_classes), dtype=torch.float32, device=self.device) if want_probs else None
        am = torch.empty((B,), dtype=torch.int32, device=self.device) if want_argmax else None
        _l.check(self._L.kws_model_forward(self.spec.handle, feat.data_ptr(), B, self.params.data_ptr(), self.state.data_ptr(),
                                           ws, nbytes, probs.data_ptr() if want_probs else 0,
                                           am.data_ptr() if want_argmax else 0, torch.cuda.current_stream().cuda_stream))
        return probs, am

    def train_fwd_bwd(self, feat, labels, class_weights=None, dropout_seed=0, grad_scale=1.0, want_probs=False,
                      ignore_index=0, bucket_event=None, forward_event=None, overlap_event=None, overlap_callback=None,
                      feat_moments=None, comm=None, comm_state_weight=1.0, stats_out=None):
        """labels: CUDA int32 (B,); class_weights: CUDA float32 (C,) or None.  Leaves grads in self.grads and
        {sum of losses, top-1 hits} in self.stats (device).  bucket_event: torch.cuda.Event recorded when the early
        gradient bucket [grad_split, P) is final.  feat_moments: the float64 CUDA tensor feature_moments(feat) returned (optional;
        simple_cnn then skips its own moment pass at the head of the step).  comm: a kws_amd.parallel.KwsComm -- the step then sums its
        gradients (and the BatchNormalization statistics times comm_state_weight) over the ranks itself, the early bucket under
        the rest of the backward pass; pass grad_scale = comm_state_weight = local clips / global clips.  stats_out: 2 contiguous CUDA
        floats that receive {sum of losses, hits} instead of self.stats (a fit loop gives every step its own row and sums once per epoch)."""
        torch = _torch()
        B = self._check_feat(feat)
        if labels.dtype != torch.int32 or not labels.is_cuda or labels.numel() != B:
            raise ValueError("labels must be a CUDA int32 tensor with B elements")
        ws, nbytes = self._workspace(B, True)
        probs = torch.empty((B, self.spec.num_classes), dtype=torch.float32, device=self.device) if want_probs else None
        a = _l.KwsTrainArgs()
        a.feat, a.labels = feat.data_ptr(), labels.data_ptr()
        a.class_weights = class_weights.data_ptr() if class_weights is not None else None
        a.B, a.ignore_index = B, int(ignore_index or 0)
        a.params, a.state, a.grads = self.params.data_ptr(), self.state.data_ptr(), self.grads.data_ptr()
        a.ws, a.ws_bytes = ws, nbytes
        a.dropout_seed, a.grad_scale = int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, float(grad_scale)
        a.probs = probs.data_ptr() if want_probs else None
        if stats_out is not None and (stats_out.dtype != torch.float32 or not stats_out.is_cuda or stats_out.numel() != 2 or not stats_out.is_contiguous()):
            raise ValueError("stats_out must be 2 contiguous CUDA float32 values")
        a.stats = (self.stats if stats_out is None else stats_out).data_ptr()
        if comm is not None:
            a.comm = comm.handle
            a.comm_state_weight = float(comm_state_weight)
        if feat_moments is not None:
            if feat_moments.dtype != torch.float64 or not feat_moments.is_cuda or feat_moments.numel() != _l.FEATURE_MOMENTS:
                raise ValueError("feat_moments must be the %d float64 values of feature_moments()" % _l.FEATURE_MOMENTS)
            a.feat_moments = feat_moments.data_ptr()
        for ev in (bucket_event, forward_event, overlap_event):
            if ev is not None and not ev.cuda_event:
                ev.record()                     # torch creates the hipEvent_t lazily; an un-recorded Event has no handle yet
        a.bucket_event = bucket_event.cuda_event if bucket_event is not None else None
        a.forward_event = forward_event.cuda_event if forward_event is not None else None
        a.overlap_event = overlap_event.cuda_event if overlap_event is not None else None
        raised = []
        if overlap_callback is not None:           # host function called at the step's overlap point (see include/kws.h)
            def _cb(_user):
                try:
                    overlap_callback()
                except BaseException as e:         # an exception must not unwind through the C frame
                    raised.append(e)
            cb = _l.OVERLAP_CB(_cb)
            a.overlap_callback = cb
        _l.check(self._L.kws_model_train_fwd_bwd(self.spec.handle, ctypes.byref(a), torch.cuda.current_stream().cuda_stream))
        if raised:
            raise raised[0]
        return probs

    @property
    def grad_split(self):
        return int(self._L.kws_model_grad_split(self.spec.handle))

    def adam_step(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        torch = _torch()
        self.step_count += 1
        self.invalidate_prepared()
        _l.check(self._L.kws_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                       self.adam_v.data_ptr(), self.params.numel(), float(lr), float(beta1), float(beta2),
                                       float(eps), self.step_count, float(grad_scale), torch.cuda.current_stream().cuda_stream))

    def sgd_step(self, lr, grad_scale=1.0):
        torch = _torch()
        self.step_count += 1
        self.invalidate_prepared()
        _l.check(self._L.kws_sgd_step(self.params.data_ptr(), self.grads.data_ptr(), self.params.numel(), float(lr),
                                      float(grad_scale), torch.cuda.current_stream().cuda_stream))

    def rmsprop_step(self, lr, rho=0.9, eps=1e-7, grad_scale=1.0):
        torch = _torch()
        self.step_count += 1
        self.invalidate_prepared()
        _l.check(self._L.kws_rmsprop_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_v.data_ptr(),
                                          self.params.numel(), float(lr), float(rho), float(eps), float(grad_scale),
                                          torch.cuda.current_stream().cuda_stream))
