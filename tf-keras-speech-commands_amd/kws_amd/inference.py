"""Graph-captured inference: featurize + forward replayed as ONE hipGraph (BASELINE.json configs[4] structure).

The capture goes through torch.cuda.CUDAGraph on a side stream: every kernel the C ABI enqueues on torch's current
stream (kws_featurize, kws_model_forward) becomes a graph node, so a replay costs one launch instead of ~25."""
from . import lib as _l


class InferenceSession(object):
    """Static-shape streaming inference: copy (or write) a batch into `self.wav`, call run(), read `self.probs`/`argmax`.

    The session owns its arithmetic: `fp16` applies to ITS forward passes only (the model's precision attribute is set around each
    eager run / capture / refresh and restored afterwards), so sessions of different precisions and plain dm.forward calls coexist on
    one DeviceModel.  It also follows the model's weights: set_weights and every optimizer step bump DeviceModel.weights_version, and
    run() re-derives the weight tables (refresh) by itself when the version it prepared for is stale."""

    def __init__(self, device_model, featurizer, batch, samples=None, wav_dtype=None, use_graph=True, fp16=False):
        import torch
        if not torch.cuda.is_available():
            raise _l.KwsError(-3, "no HIP device: inference has no CPU fallback")
        self.dm, self.feat, self.batch = device_model, featurizer, int(batch)
        # fp16=True: simple_cnn_lite forward with fp16 activations / matrix operands, fp32 accumulation (BASELINE configs[4])
        self.precision = _l.INFER_FP16 if fp16 else _l.INFER_FP32
        g = featurizer.geometry
        samples = int(samples or g["max_samples"])
        wav_dtype = wav_dtype or torch.float32
        dev = device_model.device
        self.wav = torch.zeros((self.batch, samples), dtype=wav_dtype, device=dev)
        self.features = torch.empty((self.batch, g["n_features"], g["feature_size"]), dtype=torch.float32, device=dev)
        self.probs = self.argmax = None
        self._graph = None
        self._version = None
        self._ws = device_model.new_workspace(self.batch)          # private: the prepared weight tables live in it
        self._eager()                                   # warm-up: allocations (workspace, outputs) happen outside the capture
        self.refresh()                                  # derive the weight tables once, not per batch
        torch.cuda.synchronize()
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with self._precision():
                with torch.cuda.graph(graph):
                    self._forward()
            self._graph = graph

    def _precision(self):
        """context: the model computes at THIS session's precision inside, at whatever it had before outside"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = self.dm._infer
            self.dm.set_precision(infer=self.precision)
            try:
                yield
            finally:
                self.dm.set_precision(infer=prev)
        return scope()

    def refresh(self):
        """Re-derives the weight tables the forward reads (kws_model_prepare_inference) from the model's CURRENT weights; the captured
        graph stays valid, its kernels read the same buffers.  run() calls it by itself after set_weights / optimizer steps of the
        DeviceModel; call it yourself only after writing dm.params / dm.state directly."""
        with self._precision():
            self.dm.prepare_inference(self.batch, workspace=self._ws)
        self._version = self.dm.weights_version

    def _forward(self):
        self.feat(self.wav, out=self.features)
        self.probs, self.argmax = self.dm.forward(self.features, workspace=self._ws)

    def _eager(self):
        with self._precision():
            self._forward()

    def run(self):
        if self._version != self.dm.weights_version and self._version is not None:
            self.refresh()
        if self._graph is not None:
            self._graph.replay()
        else:
            self._eager()
        return self.probs, self.argmax
