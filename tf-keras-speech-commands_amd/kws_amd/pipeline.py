"""Input pipelining for the train step: features of batch k+1 are computed on a side stream while batch k trains.

The featurizer is vector-ALU / LDS work and the model's backward pass is matrix-core work, so the two share the chip well.
Double buffered; a buffer is rewritten only after the train step that read it has finished (its backward pass recomputes
conv1 from the features, so the read extends to the end of the step).
"""


def _torch():
    import torch
    return torch


class FeaturePipeline(object):
    def __init__(self, featurizer, batch, n_features, feature_size, device="cuda", moments=False, cu_share=1):
        torch = _torch()
        self.featurizer = featurizer
        # the featurizer shares the chip with the train step here: half of each CU's LDS (kws_featurizer_set_cu_share), so the
        # step's kernels still find room on every CU.  Give the pipeline its own Featurizer object if the same parameters are
        # also used for stand-alone (inference) featurization.
        # cu_share=2 keeps the whole-chip configuration: right for a light step (simple_gru at B = 2048: 0.286 against 0.306 ms per
        # step), whose kernels need little LDS and few registers
        featurizer.set_cu_share(cu_share)
        self.side = torch.cuda.Stream(device=device)
        self.bufs = [torch.empty((batch, n_features, feature_size), dtype=torch.float32, device=device) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]      # features of the buffer are complete (recorded on side)
        self.free = [None, None]                                  # the step that read the buffer is complete (recorded on main)
        self.n_submitted = 0
        self.n_taken = 0
        # moments=True: the second moments simple_cnn's first layer needs (kws_feature_moments) are computed right behind the
        # featurizer on the side stream; take() then returns (features, moments) for DeviceModel.train_fwd_bwd(feat_moments=...)
        self.moments = None
        if moments:
            from .model import FeatureMoments
            self.moments = FeatureMoments(n_features, feature_size)
            self.mom_bufs = [torch.empty((100,), dtype=torch.float64, device=device) for _ in range(2)]

    def submit(self, wav, valid_len=None, after=None):
        """Enqueue the featurization of one batch on the side stream (returns at once).  `after`: an event on the main
        stream to start behind -- best the running step's overlap_event (DeviceModel.train_fwd_bwd(overlap_event=...),
        recorded at the point of the step that a sweep found best for this -- for simple_cnn behind the last
        BatchNormalization's activation kernel, include/kws.h); call submit from train_fwd_bwd's overlap_callback so that the launch also
        sits there in host order; default: everything enqueued so far."""
        torch = _torch()
        i = self.n_submitted % 2
        if after is not None:
            self.side.wait_event(after)
        else:
            self.side.wait_stream(torch.cuda.current_stream())
        if self.free[i] is not None:
            self.side.wait_event(self.free[i])
        with torch.cuda.stream(self.side):
            self.featurizer(wav, valid_len=valid_len, out=self.bufs[i])
            if self.moments is not None:
                self.moments(self.bufs[i], out=self.mom_bufs[i])
            self.ready[i].record(self.side)
        self.n_submitted += 1

    def take(self):
        """Features of the oldest submitted batch; the current stream waits for them."""
        torch = _torch()
        if self.n_taken >= self.n_submitted:
            raise RuntimeError("take() without a matching submit()")
        i = self.n_taken % 2
        torch.cuda.current_stream().wait_event(self.ready[i])
        self.n_taken += 1
        if self.moments is not None:
            return self.bufs[i], self.mom_bufs[i]
        return self.bufs[i]

    def release(self):
        """Call after the work that reads the most recently taken buffer has been enqueued on the current stream."""
        torch = _torch()
        i = (self.n_taken - 1) % 2
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free[i] = ev
