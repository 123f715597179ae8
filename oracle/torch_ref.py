"""oracle/torch_ref.py -- TEST / BASELINE INFRASTRUCTURE, never imported by the product path.

A torch-CPU restatement of the reference's train step, built from torch's own operators (conv2d, batch_norm,
max_pool2d, linear, autograd) instead of the hand-written numpy layers of model_oracle.py:

  * topology            classifier/models/cnn.py:27-66 (SimpleCNN), classifier/models/rnn.py:28-35 (SimpleGRU),
                        head classifier/model.py:37
  * loss                classifier/loss.py:21-42 (clipped CE on probabilities), :55-77 (weighted, unclipped)
  * optimizer           common/model_utils.py:47 -> keras Adam(epsilon=1e-7), update with epsilon OUTSIDE the bias correction
  * fit arguments       train.py:81-92 (batch mean of the per-sample losses)

Two uses:
  1. tests/test_oracle_model.py pins model_oracle.py against it (forward, gradients, BatchNormalization moving-statistic
     updates through F.batch_norm's own running-stat code, multi-step Keras-Adam trajectories);
  2. bench.py's `cpu_baseline` times it on the host cores as the stand-in for "TF-Keras CPU" (TensorFlow
     is not installed in this image), at the reference's default batch 512 / 5 classes and at the bench batch.

tf.keras itself is absent here, so this is still a restatement: parity of the MODEL half stays "unpinned" in the sense of
the task statement; what this file adds is a second, independent implementation that shares no code with model_oracle.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # keras momentum; torch's `momentum` argument is 1 - this


def same_pad(n, k, s):
    """TF 'SAME': out = ceil(n / s), total pad = max((out - 1) s + k - n, 0), the extra element at the END."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def tf_same_conv(x, w, stride, groups=1):
    """x NHWC, w HWIO -> NHWC"""
    _, H, W, _ = x.shape
    kh, kw = w.shape[:2]
    _, pt, pb = same_pad(H, kh, stride)
    _, pl, pr = same_pad(W, kw, stride)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    wt = w.permute(3, 2, 0, 1) if groups == 1 else w.permute(2, 3, 0, 1)
    return F.conv2d(xt, wt, stride=stride, groups=groups).permute(0, 2, 3, 1)


def forward(model_type, ws, x, training, drop_mask=None, new_state=None):
    """Logits.  ws: torch tensors in Keras get_weights() order; x (B, 30, 20[, 1]).  Training uses batch statistics; when
    `new_state` is a list, the updated moving statistics (computed by F.batch_norm itself) are appended to it in order."""
    it = iter(ws)
    nxt = lambda: next(it)
    if model_type in ("simple_cnn", "simple_cnn_lite"):
        lite = model_type == "simple_cnn_lite"
        h = x[..., None] if x.dim() == 3 else x
        cfg = [(1, 16, 1, False, True), (16, 32, 1, False, True), (32, 64, 2, lite, False), (64, 128, 1, True, True)]
        for cin, cout, s, relu, pool in cfg:
            if lite:
                dw, pw, b = nxt(), nxt(), nxt()
                h = tf_same_conv(h, dw, s, groups=cin)
                h = F.conv2d(h.permute(0, 3, 1, 2), pw.permute(3, 2, 0, 1)).permute(0, 2, 3, 1) + b
            else:
                h = tf_same_conv(h, nxt(), s)
            if relu:
                h = F.relu(h)                                  # Conv2D(activation='relu') in front of BN (cnn.py:55,113,122)
            g, bt, mm, mv = nxt(), nxt(), nxt(), nxt()
            rm, rv = mm.detach().clone(), mv.detach().clone()
            h = F.batch_norm(h.permute(0, 3, 1, 2), rm, rv, g, bt, training=training, momentum=1.0 - BN_MOMENTUM, eps=BN_EPS)
            if training and new_state is not None:
                new_state += [rm, rv]                          # torch updated them in place (unbiased variance, like Keras' fused op)
            h = torch.clamp(h, 0, 6)                           # ReLU(6.)
            if pool:
                h = F.max_pool2d(h, 2)
            h = h.permute(0, 2, 3, 1)
        h = h.reshape(h.shape[0], -1)                          # Flatten in NHWC order
        if training and drop_mask is not None:
            h = h * drop_mask
        k, b = nxt(), nxt()
        h = torch.clamp(h @ k + b, 0, 6)
    elif model_type == "simple_gru":
        k, rk, b = nxt(), nxt(), nxt()
        u = rk.shape[0]
        xx = x[..., 0] if x.dim() == 4 else x
        if training and drop_mask is not None:
            xx = xx * drop_mask[:, None, :]
        h = torch.zeros((xx.shape[0], u), dtype=xx.dtype)
        mx_all = xx @ k + b[0]                                  # input projection of all steps at once
        for t in range(xx.shape[1]):
            mx = mx_all[:, t]
            mh = h @ rk + b[1]                                  # reset_after=True: recurrent bias inside the reset product
            z = torch.sigmoid(mx[:, :u] + mh[:, :u])
            r = torch.sigmoid(mx[:, u:2 * u] + mh[:, u:2 * u])
            hh = mx[:, 2 * u:] + r * mh[:, 2 * u:]              # activation='linear' (rnn.py:34): no tanh
            h = z * h + (1 - z) * hh
    elif model_type == "simple_lstm":
        k, rk, b = nxt(), nxt(), nxt()
        u = rk.shape[0]
        xx = x[..., 0] if x.dim() == 4 else x
        if training and drop_mask is not None:
            xx = xx * drop_mask[:, None, :]
        h = torch.zeros((xx.shape[0], u), dtype=xx.dtype)
        c = torch.zeros((xx.shape[0], u), dtype=xx.dtype)
        for t in range(xx.shape[1]):
            a = xx[:, t] @ k + h @ rk + b
            i, f, g, o = torch.sigmoid(a[:, :u]), torch.sigmoid(a[:, u:2 * u]), torch.tanh(a[:, 2 * u:3 * u]), torch.sigmoid(a[:, 3 * u:])
            c = f * c + i * g
            h = o * torch.tanh(c)
    else:
        raise ValueError('Unsupported model type')
    k, b = nxt(), nxt()
    return h @ k + b


def loss_of(logits, labels, class_weights=None):
    """mean over the batch of classifier/loss.py's per-sample losses (on softmax probabilities)"""
    p = torch.softmax(logits, -1)
    py = p[torch.arange(p.shape[0]), labels]
    if class_weights is not None:
        return (-torch.log(py) * class_weights[labels]).mean(), p          # loss.py:67-71, no clipping
    return (-torch.log(torch.clamp(py / p.sum(-1), 1e-7, 1 - 1e-7))).mean(), p   # K.categorical_crossentropy on probabilities


class KerasAdam(object):
    """keras.optimizers.Adam: w -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)   (eps outside the correction)"""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0
        self.m = self.v = None

    @torch.no_grad()
    def step(self, params, grads):
        if self.m is None:
            self.m = [torch.zeros_like(p) for p in params]
            self.v = [torch.zeros_like(p) for p in params]
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))


class TorchModel(object):
    """Weights as torch tensors in Keras order + one train step; dtype float64 for the oracle cross-checks, float32 for the
    CPU baseline."""

    def __init__(self, model_type, weights, trainable_flags, dtype=torch.float64):
        self.model_type = model_type
        self.flags = list(trainable_flags)
        self.ws = [torch.tensor(np.asarray(w), dtype=dtype).requires_grad_(bool(t)) for w, t in zip(weights, self.flags)]
        self.dtype = dtype

    def trainable(self):
        return [w for w, t in zip(self.ws, self.flags) if t]

    def get_weights(self):
        return [w.detach().numpy().copy() for w in self.ws]

    def train_step(self, opt, x, labels, class_weights=None, drop_mask=None):
        x = torch.as_tensor(x, dtype=self.dtype)
        labels = torch.as_tensor(np.asarray(labels), dtype=torch.long)
        cw = None if class_weights is None else torch.as_tensor(class_weights, dtype=self.dtype)
        dm = None if drop_mask is None else torch.as_tensor(drop_mask, dtype=self.dtype)
        for w in self.trainable():
            w.grad = None
        new_state = []
        loss, p = loss_of(forward(self.model_type, self.ws, x, True, dm, new_state), labels, cw)
        loss.backward()
        tr = self.trainable()
        grads = [w.grad for w in tr]
        if opt is not None:
            opt.step(tr, grads)
        it = iter(new_state)
        with torch.no_grad():
            for w, t in zip(self.ws, self.flags):
                if not t:
                    w.copy_(next(it))
        return float(loss.item()), p.detach(), grads

    @torch.no_grad()
    def predict(self, x):
        return torch.softmax(forward(self.model_type, self.ws, torch.as_tensor(x, dtype=self.dtype), False), -1).numpy()
