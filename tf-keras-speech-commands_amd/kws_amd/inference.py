"""Graph-captured inference: featurize + forward replayed as ONE hipGraph (BASELINE.json configs[4] structure).

The capture goes through torch.cuda.CUDAGraph on a side stream: every kernel the C ABI enqueues on torch's current
stream (kws_featurize, kws_model_forward) becomes a graph node, so a replay costs one launch instead of ~25."""
from . import lib as _l


class InferenceSession(object):
    """Static-shape streaming inference: copy (or write) a batch into `self.wav`, call run(), read `self.probs`/`argmax`."""

    def __init__(self, device_model, featurizer, batch, samples=None, wav_dtype=None, use_graph=True, fp16=False):
        import torch
        if not torch.cuda.is_available():
            raise _l.KwsError(-3, "no HIP device: inference has no CPU fallback")
        self.dm, self.feat, self.batch = device_model, featurizer, int(batch)
        # fp16=True: simple_cnn_lite forward with fp16 activations / matrix operands, fp32 accumulation (BASELINE configs[4]);
        # a per-model attribute (kws_model_set_precision), so sessions of different precisions coexist in one process
        self.precision = _l.INFER_FP16 if fp16 else _l.INFER_FP32
        device_model.set_precision(infer=self.precision)
        g = featurizer.geometry
        samples = int(samples or g["max_samples"])
        wav_dtype = wav_dtype or torch.float32
        dev = device_model.device
        self.wav = torch.zeros((self.batch, samples), dtype=wav_dtype, device=dev)
        self.features = torch.empty((self.batch, g["n_features"], g["feature_size"]), dtype=torch.float32, device=dev)
        self.probs = self.argmax = None
        self._graph = None
        self._ws = device_model.new_workspace(self.batch)          # private: the prepared weight tables live in it
        self._eager()                                   # warm-up: allocations (workspace, outputs) happen outside the capture
        self.refresh()                                  # weights are fixed from here on: derive their tables once, not per batch
        torch.cuda.synchronize()
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._eager()
            self._graph = graph

    def refresh(self):
        """Call after the model's weights changed (set_weights / training): re-derives the weight tables the captured forward
        reads (kws_model_prepare_inference); the graph itself stays valid, its kernels read the same buffers."""
        self.dm.prepare_inference(self.batch, workspace=self._ws)

    def _eager(self):
        self.feat(self.wav, out=self.features)
        self.probs, self.argmax = self.dm.forward(self.features, workspace=self._ws)

    def run(self):
        if self._graph is not None:
            self._graph.replay()
        else:
            self._eager()
        return self.probs, self.argmax
