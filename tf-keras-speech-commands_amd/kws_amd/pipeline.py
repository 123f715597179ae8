"""Input pipelining for the train step: the inputs of batch k+1 are prepared on a side stream while batch k trains.

"Prepared" = featurized from raw audio (the featurizer is vector-ALU / LDS work, the model's backward pass matrix-core work, so the
two share the chip well) or gathered from a device-resident feature set, the labels gathered, and -- for simple_cnn -- the second moments
of the features computed (kws_feature_moments), all behind one event.  Double buffered; a buffer is rewritten only after the train step
that read it has finished (its backward pass recomputes conv1 from the features, so the read extends to the end of the step).
"""


def _torch():
    import torch
    return torch


class FeaturePipeline(object):
    def __init__(self, featurizer, batch, n_features, feature_size, device="cuda", moments=False, cu_share=1, labels=False):
        torch = _torch()
        self.featurizer = featurizer
        # the featurizer shares the chip with the train step here: one persistent block per CU (kws_featurizer_set_cu_share), so the
        # step's kernels still find wave slots and LDS on every CU.  Give the pipeline its own Featurizer object if the same parameters
        # are also used for stand-alone (inference) featurization.
        # cu_share=2 keeps the whole-chip configuration: right for a light step (simple_gru at B = 2048), whose kernels need little
        # LDS and few registers
        if featurizer is not None:
            featurizer.set_cu_share(cu_share)
        self.batch = int(batch)
        self.side = torch.cuda.Stream(device=device)
        self.bufs = [torch.empty((batch, n_features, feature_size), dtype=torch.float32, device=device) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]      # the inputs of the buffer are complete (recorded on side)
        self.free = [None, None]                                  # the step that read the buffer is complete (recorded on main)
        self.count = [batch, batch]                               # clips in the buffer
        self.n_submitted = 0
        self.n_taken = 0
        # moments=True: the second moments simple_cnn's first layer needs (kws_feature_moments) are computed right behind the
        # features on the side stream; take() then returns (features, moments) for DeviceModel.train_fwd_bwd(feat_moments=...)
        self.moments = None
        if moments:
            from .model import FeatureMoments
            self.moments = FeatureMoments(n_features, feature_size)
            self.mom_bufs = [torch.empty((100,), dtype=torch.float64, device=device) for _ in range(2)]
        # labels=True: submit(labels=..., index=...) also gathers the batch's labels; take() then returns them as the last element
        self.lab_bufs = [torch.empty((batch,), dtype=torch.int32, device=device) for _ in range(2)] if labels else None

    def submit(self, wav=None, valid_len=None, after=None, index=None, features=None, labels=None):
        """Enqueue the preparation of one batch on the side stream (returns at once).  Source: `wav` (rows, samples) raw audio,
        featurized here, or `features` (rows, n_features, feature_size); `index` (CUDA int32) picks the batch's rows from either
        (default: every row); `labels` (rows,) int32 are gathered with the same index.  `after`: an event on the main stream to
        start behind -- best the running step's overlap_event (DeviceModel.train_fwd_bwd(overlap_event=...), recorded at the point
        of the step that a sweep found best for this -- for simple_cnn behind conv3's forward kernel (round 3's sweep),
        include/kws.h); call submit from train_fwd_bwd's overlap_callback so that the launch also sits there in host order;
        default: everything enqueued so far."""
        torch = _torch()
        i = self.n_submitted % 2
        if after is not None:
            self.side.wait_event(after)
        else:
            self.side.wait_stream(torch.cuda.current_stream())
        if self.free[i] is not None:
            self.side.wait_event(self.free[i])
        src = wav if features is None else features
        n = int(index.numel()) if index is not None else int(src.shape[0])
        if n > self.batch:
            raise ValueError("batch of %d clips in a pipeline built for %d" % (n, self.batch))
        out = self.bufs[i] if n == self.batch else self.bufs[i][:n]
        with torch.cuda.stream(self.side):
            if features is None:
                self.featurizer(wav, valid_len=valid_len, out=out, index=index)
            elif index is not None:
                torch.index_select(features.reshape(features.shape[0], out.shape[1], out.shape[2]), 0, index, out=out)
            else:
                out.copy_(features.reshape(out.shape))
            if labels is not None:
                if self.lab_bufs is None:
                    raise ValueError("the pipeline was built without label buffers")
                lab = self.lab_bufs[i][:n]
                if index is not None:
                    torch.index_select(labels, 0, index, out=lab)
                else:
                    lab.copy_(labels)
            if self.moments is not None and n > 0:
                self.moments(out, out=self.mom_bufs[i])
            self.ready[i].record(self.side)
        self.count[i] = n
        self.n_submitted += 1

    def take(self):
        """Inputs of the oldest submitted batch; the current stream waits for them.  -> features, or (features, moments) with
        moments=True; the gathered labels are appended when the pipeline was built with labels=True."""
        torch = _torch()
        if self.n_taken >= self.n_submitted:
            raise RuntimeError("take() without a matching submit()")
        i = self.n_taken % 2
        torch.cuda.current_stream().wait_event(self.ready[i])
        self.n_taken += 1
        n = self.count[i]
        feat = self.bufs[i] if n == self.batch else self.bufs[i][:n]
        if self.moments is None and self.lab_bufs is None:
            return feat
        out = (feat,)
        if self.moments is not None:
            out += (self.mom_bufs[i],)
        if self.lab_bufs is not None:
            out += (self.lab_bufs[i][:n],)
        return out

    def release(self):
        """Call after the work that reads the most recently taken buffer has been enqueued on the current stream."""
        torch = _torch()
        i = (self.n_taken - 1) % 2
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free[i] = ev
