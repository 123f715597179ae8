"""Model factory (host mirror of the reference's classifier/model.py:14-46).

`get_model(model_type, num_classes, batch_size=None, weights_path=None)` returns an object with the part of the
tf.keras.Model API the reference's callers use (train.py:75-95, eval.py:29, listen.py:139): compile / fit / predict /
evaluate / summary / save / load_weights / get_weights / set_weights.  Every batch of `fit` and `predict` runs in the
HIP kernels behind include/kws.h; there is no CPU path (using the model without a GPU raises kws_amd.KwsError).

Beyond the reference: `fit` / `predict` / `evaluate` also accept raw audio (N, samples) and featurize it on the GPU in
front of the network, and under torch.distributed `fit` runs data-parallel (kws_amd.parallel)."""
import math
import os
import time

import numpy as np

from classifier.models.cnn import SimpleCNN, SimpleCNNLite
from classifier.models.rnn import SimpleGRU, SimpleLSTM
from classifier.params import pr
from kws_amd import lib as _l
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec


class History(object):
    def __init__(self):
        self.history = {}
        self.epoch = []

    def append(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class KWSModel(object):
    """A compiled-once keyword-spotting classifier living on one HIP device."""

    def __init__(self, model_type, num_classes, batch_size=None, seed=None):
        cnn = model_type in ('simple_cnn', 'simple_cnn_lite')
        # RNN models take 2-D input per clip, CNNs 3-D (reference :16-20)
        self.input_shape = (pr.n_features, pr.feature_size, 1) if cnn else (pr.n_features, pr.feature_size)
        self.batch_size = batch_size
        self.spec = ModelSpec(model_type, num_classes, pr.n_features, pr.feature_size)   # ValueError('Unsupported model type')
        if model_type == 'simple_cnn':
            self.layers = SimpleCNN(input_shape=self.input_shape, feature_size=128)
        elif model_type == 'simple_cnn_lite':
            self.layers = SimpleCNNLite(input_shape=self.input_shape, feature_size=128)
        elif model_type == 'simple_gru':
            self.layers = SimpleGRU(input_shape=self.input_shape, recurrent_units=48)
        else:
            self.layers = SimpleLSTM(input_shape=self.input_shape, recurrent_units=48)
        feat = self.layers[-1]["output_shape"][-1]
        self.layers.append(dict(name='score_predict', type='Dense', output_shape=(num_classes,),
                                params=feat * num_classes + num_classes, activation='softmax'))
        self.model_type, self.num_classes = model_type, num_classes
        self.name = 'model'
        self.input_names, self.output_names = ['feature_input'], ['score_predict']
        self.optimizer = self.loss = None
        self.metrics_names = ['loss']
        self.stop_training = False
        self.history = None
        self._host_weights = init_weights(self.spec, seed)
        self._dm = None
        self._featurizer = None
        self._dropout_base = int(np.random.default_rng(seed).integers(1, 2 ** 31))
        self._class_weights_dev = None

    # ---- device state ------------------------------------------------------------------------------------------
    def _device(self):
        if self._dm is None:
            self._dm = DeviceModel(self.spec)
            self._dm.set_weights(self._host_weights)
        return self._dm

    def _get_featurizer(self):
        from common.data_utils import get_featurizer
        return get_featurizer()

    # ---- weights -----------------------------------------------------------------------------------------------
    def get_weights(self):
        return self._dm.get_weights() if self._dm is not None else [w.copy() for w in self._host_weights]

    def set_weights(self, weights):
        if len(weights) != len(self.spec.tensors):
            raise ValueError("expected %d weight arrays, got %d" % (len(self.spec.tensors), len(weights)))
        ws = []
        for t, w in zip(self.spec.tensors, weights):
            w = np.asarray(w, np.float32)
            if tuple(w.shape) != t["shape"]:
                raise ValueError("%s: expected shape %s, got %s" % (t["name"], t["shape"], w.shape))
            ws.append(w.copy())
        self._host_weights = ws
        if self._dm is not None:
            self._dm.set_weights(ws)

    @property
    def weight_names(self):
        return [t["name"] for t in self.spec.tensors]

    def count_params(self):
        return self.spec.trainable_count() + self.spec.non_trainable_count()

    def save_weights(self, filepath):
        """Keras-ordered arrays in an .npz (keys = Keras weight names).  HDF5 needs h5py, which this image lacks."""
        if str(filepath).endswith(('.h5', '.hdf5', '.keras')):
            try:
                import h5py  # noqa: F401
            except ImportError:
                raise ImportError("h5py is not installed: save to a .npz path instead (same arrays, Keras order)")
            raise NotImplementedError("HDF5 export is not implemented; use .npz")
        arrays = {n: w for n, w in zip(self.weight_names, self.get_weights())}
        arrays["__meta__"] = np.array([self.model_type, str(self.num_classes), str(pr.n_features), str(pr.feature_size)])
        arrays["__order__"] = np.array(self.weight_names)
        np.savez(filepath, **arrays)

    save = save_weights   # model.save(path) (train.py:95): the optimizer state is not part of the file

    def load_weights(self, filepath, by_name=False, skip_mismatch=False):
        path = filepath if os.path.exists(filepath) else filepath + '.npz'
        z = np.load(path, allow_pickle=False)
        order = [str(n) for n in z["__order__"]] if "__order__" in z.files else [n for n in z.files if not n.startswith("__")]
        if by_name:
            cur = dict(zip(self.weight_names, self.get_weights()))
            for n in order:
                if n in cur and (cur[n].shape == z[n].shape or not skip_mismatch):
                    cur[n] = z[n]
            self.set_weights([cur[n] for n in self.weight_names])
        else:
            self.set_weights([z[n] for n in order])

    # ---- Keras surface -----------------------------------------------------------------------------------------
    def compile(self, optimizer, loss, metrics=None):
        from classifier.loss import SparseCategoricalCrossEntropy, WeightedSparseCategoricalCrossEntropy
        from common.model_utils import Optimizer, get_optimizer
        if isinstance(optimizer, str):
            optimizer = get_optimizer(optimizer, 1e-3, decay_type=None)
        if not isinstance(optimizer, Optimizer):
            raise TypeError("optimizer must come from common.model_utils.get_optimizer")
        if not isinstance(loss, (SparseCategoricalCrossEntropy, WeightedSparseCategoricalCrossEntropy)):
            raise TypeError("loss must be a classifier.loss object")
        if getattr(loss, "from_logits", False):
            raise ValueError("the model ends in a softmax; from_logits=True does not apply")
        if isinstance(loss, WeightedSparseCategoricalCrossEntropy) and len(loss.weights) != self.num_classes:
            raise ValueError("%d class weights for %d classes" % (len(loss.weights), self.num_classes))
        self.optimizer, self.loss = optimizer, loss
        self.metrics_names = ['loss'] + [('accuracy' if m in ('accuracy', 'acc') else str(m)) for m in (metrics or [])]
        self._class_weights_dev = None

    def summary(self, print_fn=print):
        line = "_" * 65
        print_fn('Model: "%s"' % self.name)
        print_fn(line)
        print_fn("%-37s%-20s%s" % ("Layer (type)", "Output Shape", "Param #"))
        print_fn("=" * 65)
        bs = self.batch_size
        print_fn("%-37s%-20s%d" % ("feature_input (InputLayer)", str((bs,) + tuple(self.input_shape)), 0))
        for l in self.layers:
            print_fn("%-37s%-20s%d" % ("%s (%s)" % (l["name"], l["type"]), str((bs,) + tuple(l["output_shape"])), l["params"]))
        print_fn("=" * 65)
        tr, nt = self.spec.trainable_count(), self.spec.non_trainable_count()
        print_fn("Total params: {:,}".format(tr + nt))
        print_fn("Trainable params: {:,}".format(tr))
        print_fn("Non-trainable params: {:,}".format(nt))
        print_fn(line)

    # ---- batching helpers --------------------------------------------------------------------------------------
    def _to_device_inputs(self, x):
        """-> (tensor on the GPU, is_audio).  Features: (N, n_features, feature_size[, 1]); audio: (N, samples)."""
        import torch
        if not torch.cuda.is_available():
            raise _l.KwsError(-3, "no HIP device: the model has no CPU fallback")
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        nf, fs = pr.n_features, pr.feature_size
        if t.dim() == 4 and tuple(t.shape[1:]) == (nf, fs, 1) or t.dim() == 3 and tuple(t.shape[1:]) == (nf, fs):
            return t.reshape(t.shape[0], nf * fs).to(torch.float32).cuda(), False
        if t.dim() == 2:
            if t.dtype not in (torch.int16, torch.float32):
                t = t.to(torch.float32)
            return t.cuda().contiguous(), True
        raise ValueError("expected features (N, %d, %d[, 1]) or audio (N, samples), got %s" % (nf, fs, tuple(t.shape)))

    def _labels(self, y, n):
        import torch
        t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y)))
        t = t.reshape(-1).to(torch.int32).cuda()
        if t.numel() != n:
            raise ValueError("%d labels for %d samples" % (t.numel(), n))
        return t

    def _features_of(self, xb, is_audio):
        if is_audio:
            return self._get_featurizer()(xb.contiguous())
        return xb.reshape(xb.shape[0], pr.n_features, pr.feature_size)

    def _loss_args(self):
        import torch
        from classifier.loss import WeightedSparseCategoricalCrossEntropy
        if self.loss is None:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
        if isinstance(self.loss, WeightedSparseCategoricalCrossEntropy):
            if self._class_weights_dev is None:
                self._class_weights_dev = torch.from_numpy(np.asarray(self.loss.weights, np.float32)).cuda()
            return self._class_weights_dev, int(self.loss.ignore_index or 0)
        return None, int(self.loss.ignore_index or 0)

    def _apply_optimizer(self, dm):
        opt = self.optimizer
        lr = opt.current_lr()
        if opt.kind == 'adam':
            dm.adam_step(lr, opt.beta_1, opt.beta_2, opt.epsilon)
        elif opt.kind == 'rmsprop':
            dm.rmsprop_step(lr, opt.rho, opt.epsilon)
        else:
            dm.sgd_step(lr)
        opt.iterations += 1

    def _train_batch(self, dm, dp, xb, yb, is_audio, cw, ignore_index, seed, weight=None):
        """One step.  Data parallel: `weight` = this rank's share of the global batch (DataParallel.shard_plan): the
        gradient of the local mean loss is scaled by it, so the summed gradient is the global-batch mean the reference
        computes (train.py:81-92) even when the last batch splits unevenly; a rank whose shard is empty contributes
        zeros.  The BatchNormalization moving statistics are averaged with the same weights in the same exchange."""
        import torch
        if dp is not None and dp.active:
            weight = dp.grad_scale if weight is None else weight
            state = dm.state if self.spec.state_count > 0 else None
            if dp.comm is not None:
                # the step exchanges its gradients itself (kws_train_args.comm: early bucket under the rest of the backward pass); a rank
                # without clips joins the same collectives with cleared gradients and weight 0
                if xb.shape[0] > 0:
                    dm.train_fwd_bwd(self._features_of(xb, is_audio), yb, cw, dropout_seed=seed, grad_scale=weight, ignore_index=ignore_index,
                                     comm=dp.comm, comm_state_weight=weight)
                    stats = dm.stats
                else:
                    dm.grads.zero_()
                    stats = torch.zeros_like(dm.stats)
                    dp.comm.allreduce_grads(dm.grads, dm.grad_split, state, weight)
            else:
                ev = self._bucket_event
                if xb.shape[0] > 0:
                    dm.train_fwd_bwd(self._features_of(xb, is_audio), yb, cw, dropout_seed=seed, grad_scale=weight, ignore_index=ignore_index,
                                     bucket_event=ev)
                    stats = dm.stats
                else:
                    dm.grads.zero_()
                    ev.record()
                    stats = torch.zeros_like(dm.stats)
                dp.sync_grads(dm.grads, dm.grad_split, ev, state=state, state_weight=weight)
        else:
            dm.train_fwd_bwd(self._features_of(xb, is_audio), yb, cw, dropout_seed=seed, ignore_index=ignore_index)
            stats = dm.stats
        self._apply_optimizer(dm)
        return stats

    def train_on_batch(self, x, y):
        dm = self._device()
        xd, is_audio = self._to_device_inputs(x)
        yd = self._labels(y, xd.shape[0])
        cw, ig = self._loss_args()
        self._global_step = getattr(self, "_global_step", 0) + 1
        st = self._train_batch(dm, None, xd, yd, is_audio, cw, ig, (self._dropout_base << 20) + self._global_step).cpu().numpy()
        return [float(st[0]) / xd.shape[0], float(st[1]) / xd.shape[0]]

    def fit(self, x, y, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None, shuffle=True,
            initial_epoch=0, validation_freq=1, **kwargs):
        """Keras-style training loop (train.py:81-92).  The dataset is placed in HBM once; every epoch draws a fresh
        permutation on the device, the partial last batch is kept, per-epoch loss/accuracy are the sample-weighted
        means of the batches.  Under torch.distributed each rank trains on its slice of every global batch and the
        gradients / BatchNormalization statistics are exchanged through the C ABI (kws_amd.parallel, csrc/kws_comm.hip).

        The step that runs is the pipelined one bench.py measures (kws_amd.pipeline.FeaturePipeline): while batch k trains, batch k+1 is
        drawn from the resident dataset on a side stream -- raw audio is featurized IN PLACE from the rows the permutation names
        (kws_featurize_gather), features are gathered -- together with its labels and, for simple_cnn, the second moments of its features
        (kws_train_args.feat_moments); the side stream starts at the step's overlap point (overlap_event / overlap_callback).  Nothing
        on the host waits for the device inside an epoch.  `pipeline=False` runs the same arithmetic step by step on one stream
        (bit-identical results in the deterministic gradient mode: tests/test_host_api_gpu.py)."""
        import torch
        from kws_amd.parallel import DataParallel
        from kws_amd.pipeline import FeaturePipeline
        dm = self._device()
        cw, ig = self._loss_args()
        dp = kwargs.pop("data_parallel", None) or DataParallel.for_device()    # injectable for tests (a forced one-rank world)
        pipelined = bool(kwargs.pop("pipeline", True))
        if dp.active:
            dp.broadcast_(dm.params)
            dp.broadcast_(dm.state)
            self._bucket_event = torch.cuda.Event()
            self._bucket_event.record()
        batch_size = int(batch_size or 32)
        xd, is_audio = self._to_device_inputs(x)
        n = xd.shape[0]
        yd = self._labels(y, n)
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.set_model(self)
            cb.on_train_begin()
        self.history = History()
        self.stop_training = False
        steps = max(1, math.ceil(n / batch_size))
        self._global_step = getattr(self, "_global_step", 0)
        nf, fs = pr.n_features, pr.feature_size
        local_max = dp.shard(batch_size)[1] - dp.shard(batch_size)[0] if dp.active else batch_size
        pipe = None
        if pipelined:
            key = (local_max, is_audio, nf, fs)
            if getattr(self, "_pipe_key", None) != key:
                from kws_amd.featurizer import Featurizer
                # its own featurizer object (same bits in every configuration): the whole-chip configuration, as bench.py's pipeline -- at
                # the step's overlap point the featurizer's blocks and the step's kernels exclude each other from the CUs anyway
                self._pipe = FeaturePipeline(Featurizer(pr) if is_audio else None, max(1, local_max), nf, fs, device=xd.device,
                                             moments=self.model_type == 'simple_cnn', labels=True, cu_share=2)
                self._pipe_key = key
            pipe = self._pipe
        stats_all = torch.zeros((steps, 2), dtype=torch.float32, device=xd.device)     # one row per step: no per-step accumulation kernels
        overlap_ev = torch.cuda.Event()
        state = dm.state if self.spec.state_count > 0 else None
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(epoch)
            t0 = time.time()
            if shuffle:
                perm = torch.randperm(n, device=xd.device)
                if dp.active:
                    dp.broadcast_(perm)
            else:
                perm = torch.arange(n, device=xd.device)
            perm = perm.to(torch.int32)
            stats_all.zero_()
            seen = 0

            def shard_of(i):
                idx = perm[i * batch_size:(i + 1) * batch_size]
                if dp.active:
                    lo, hi, weight = dp.shard_plan(idx.numel())     # an empty shard (weight 0) still joins the collectives
                    return idx[lo:hi], weight
                return idx, None

            def submit(i, after=None):
                idx, _ = shard_of(i)
                if idx.numel() > 0:
                    if is_audio:
                        pipe.submit(wav=xd, index=idx, labels=yd, after=after)
                    else:
                        pipe.submit(features=xd, index=idx, labels=yd, after=after)

            if pipe is not None:
                submit(0)
            for i in range(steps):
                idx, weight = shard_of(i)
                nloc = idx.numel()
                self._global_step += 1
                seed = (self._dropout_base << 20) + self._global_step * 64 + dp.rank
                more = pipe is not None and i + 1 < steps
                if nloc > 0:
                    mom = None
                    if pipe is not None:
                        got = pipe.take()
                        feat, yb = got[0], got[-1]
                        mom = got[1] if len(got) == 3 else None
                    else:
                        xb = xd.index_select(0, idx)
                        feat, yb = self._features_of(xb, is_audio), yd.index_select(0, idx)
                    kw = dict(dropout_seed=seed, ignore_index=ig, feat_moments=mom, stats_out=stats_all[i])
                    if more:
                        kw.update(overlap_event=overlap_ev, overlap_callback=lambda j=i + 1: submit(j, after=overlap_ev))
                    if dp.active:
                        w = dp.grad_scale if weight is None else weight
                        if dp.comm is not None:
                            dm.train_fwd_bwd(feat, yb, cw, grad_scale=w, comm=dp.comm, comm_state_weight=w, **kw)
                        else:
                            dm.train_fwd_bwd(feat, yb, cw, grad_scale=w, bucket_event=self._bucket_event, **kw)
                            dp.sync_grads(dm.grads, dm.grad_split, self._bucket_event, state=state, state_weight=w)
                    else:
                        dm.train_fwd_bwd(feat, yb, cw, **kw)
                else:
                    # this rank's shard of a partial batch is empty: cleared gradients, weight 0, the same collectives
                    dm.grads.zero_()
                    if dp.comm is not None:
                        dp.comm.allreduce_grads(dm.grads, dm.grad_split, state, weight)
                    else:
                        self._bucket_event.record()
                        dp.sync_grads(dm.grads, dm.grad_split, self._bucket_event, state=state, state_weight=weight)
                    if more:
                        submit(i + 1)
                self._apply_optimizer(dm)
                seen += nloc
            tot = torch.cat([stats_all.double().sum(0), torch.tensor([float(seen)], dtype=torch.float64, device=xd.device)])
            if dp.active:
                dp.sum_(tot)                     # (the BatchNormalization moving statistics are averaged every step)
            tot = tot.cpu().numpy()
            dt = time.time() - t0                # the device is idle again here (tot.cpu() waited for the last step)
            logs = {'loss': float(tot[0] / tot[2]), 'accuracy': float(tot[1] / tot[2])}
            if validation_data is not None and (epoch + 1) % validation_freq == 0:
                vl, va = self.evaluate(validation_data[0], validation_data[1], batch_size=batch_size, verbose=0)
                logs['val_loss'], logs['val_accuracy'] = vl, va
            logs['lr'] = self.optimizer.current_lr()
            logs['clips_per_sec'] = float(tot[2] / dt) if dt > 0 else 0.0
            if verbose and dp.rank == 0:
                print('Epoch %d/%d - %.1fs - %s' % (epoch + 1, epochs, dt, ' - '.join(
                    '%s: %.4f' % (k, v) for k, v in logs.items() if k not in ('lr', 'clips_per_sec'))))
            self.history.append(epoch, logs)
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end()
        return self.history

    def predict(self, x, batch_size=None, verbose=0, **kwargs):
        """-> (N, num_classes) softmax scores (numpy), inference mode (moving BN statistics, no dropout)"""
        import torch
        dm = self._device()
        xd, is_audio = self._to_device_inputs(x)
        batch_size = int(batch_size or 4096)
        out = []
        for i in range(0, xd.shape[0], batch_size):
            probs, _ = dm.forward(self._features_of(xd[i:i + batch_size], is_audio).contiguous(), True, False)
            out.append(probs)
        if not out:
            return np.zeros((0, self.num_classes), np.float32)
        return torch.cat(out).cpu().numpy()

    __call__ = predict

    def evaluate(self, x, y, batch_size=None, verbose=0, **kwargs):
        """-> [loss, accuracy] in inference mode with the compiled loss"""
        import torch
        dm = self._device()
        xd, is_audio = self._to_device_inputs(x)
        n = xd.shape[0]
        yd = self._labels(y, n)
        batch_size = int(batch_size or 4096)
        loss_sum = torch.zeros((), dtype=torch.float64, device=xd.device)
        hits = torch.zeros((), dtype=torch.float64, device=xd.device)
        for i in range(0, n, batch_size):
            probs, am = dm.forward(self._features_of(xd[i:i + batch_size], is_audio).contiguous(), True, True)
            yb = yd[i:i + batch_size]
            loss_sum += self.loss(yb, probs).double().sum()
            hits += (am == yb).double().sum()
        return [float(loss_sum.item()) / max(n, 1), float(hits.item()) / max(n, 1)]

    def test_on_batch(self, x, y):
        return self.evaluate(x, y, batch_size=len(x))


def get_model(model_type, num_classes, batch_size=None, weights_path=None):
    """simple_cnn / simple_cnn_lite (4-D input) or simple_gru / simple_lstm (3-D input) + Dense softmax 'score_predict'"""
    model = KWSModel(model_type, num_classes, batch_size=batch_size)

    if weights_path:
        model.load_weights(weights_path, by_name=False)
        print('Load weights {}.'.format(weights_path))

    return model
