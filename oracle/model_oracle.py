"""oracle/model_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement (float64 by default) of the model half of the reference's train step:
  topology   classifier/models/cnn.py:27-66 (SimpleCNN), :93-133 (SimpleCNNLite),
             classifier/models/rnn.py:28-35 (SimpleGRU), :64-71 (SimpleLSTM), head classifier/model.py:17-40
  loss       classifier/loss.py:21-42 (plain), :55-77 (class-weighted)
  optimizer  common/model_utils.py:47 (keras Adam, defaults)
  fit        train.py:75-92 (mean over the batch, sparse top-1 accuracy)
The layer arithmetic itself lives in tf.keras, a third-party dependency that is UNPINNED
(requirements.txt:4) and absent from /root/reference and from this container, and the reference ships no
weights, logits or tests for it.  PARITY UNPINNED for this half: the Keras semantics restated here are the
published ones (SURVEY.md section 7 lists them: TF 'SAME' padding, BatchNormalization momentum 0.99 /
epsilon 1e-3 / biased variance for normalisation and unbiased for the moving average, 2x2 'valid' max-pool,
ReLU(6), inverted dropout, GRU reset_after=True with gate order z,r,h and a LINEAR candidate, Keras Adam
with epsilon outside the bias correction, probability-space cross-entropy clipped to [1e-7, 1-1e-7]).
tests/test_oracle_model.py cross-checks every layer and the full gradients against torch-CPU autograd.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
import numpy as np

BN_MOMENTUM = 0.99
BN_EPS = 1e-3
CE_EPS = 1e-7


# ----------------------------------------------------------------------------------------------
# geometry helpers
# ----------------------------------------------------------------------------------------------
def same_pad(n_in, k, s):
    """TF 'SAME': out = ceil(in/s); total pad = max((out-1)*s + k - in, 0); extra goes to the END."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def im2col(x, kh, kw, stride):
    """x (B,H,W,C) -> cols (B,Ho,Wo,kh*kw*C) with k index = (i*kw + j)*C + c  (HWIO flattening)."""
    B, H, W, C = x.shape
    Ho, pt, pb = same_pad(H, kh, stride)
    Wo, pl, pr_ = same_pad(W, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr_), (0, 0)))
    cols = np.empty((B, Ho, Wo, kh * kw * C), x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[..., (i * kw + j) * C:(i * kw + j + 1) * C] = xp[:, i:i + (Ho - 1) * stride + 1:stride,
                                                                  j:j + (Wo - 1) * stride + 1:stride, :]
    return cols, (pt, pl, Ho, Wo)


def col2im(dcols, x_shape, kh, kw, stride):
    B, H, W, C = x_shape
    Ho, pt, pb = same_pad(H, kh, stride)
    Wo, pl, pr_ = same_pad(W, kw, stride)
    dxp = np.zeros((B, H + pt + pb, W + pl + pr_, C), dcols.dtype)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] += \
                dcols[..., (i * kw + j) * C:(i * kw + j + 1) * C]
    return dxp[:, pt:pt + H, pl:pl + W, :]


# ----------------------------------------------------------------------------------------------
# layers: each has forward(x, training) and backward(dy) -> dx, accumulating self.grads
# ----------------------------------------------------------------------------------------------
class Layer(object):
    params = ()      # names of trainable arrays, Keras get_weights() order
    state = ()       # names of non-trainable arrays

    def weights(self):
        return [(n, getattr(self, n), True) for n in self.params] + [(n, getattr(self, n), False) for n in self.state]


class Conv2D(Layer):
    """Conv2D(filters, 3, strides, padding='same', use_bias=False[, activation='relu'])  cnn.py:27-31,53-58"""
    params = ("kernel",)

    def __init__(self, cin, cout, stride=1, relu=False, k=3):
        self.cin, self.cout, self.stride, self.relu, self.k = cin, cout, stride, relu, k
        self.kernel = np.zeros((k, k, cin, cout))

    def forward(self, x, training):
        cols, _ = im2col(x, self.k, self.k, self.stride)
        y = cols @ self.kernel.reshape(-1, self.cout)
        if self.relu:
            y = np.maximum(y, 0)
        self.cache = (x.shape, cols, y)
        return y

    def backward(self, dy):
        x_shape, cols, y = self.cache
        if self.relu:
            dy = dy * (y > 0)
        K = self.kernel.reshape(-1, self.cout)
        self.grads = {"kernel": (cols.reshape(-1, K.shape[0]).T @ dy.reshape(-1, self.cout)).reshape(self.kernel.shape)}
        return col2im(dy @ K.T, x_shape, self.k, self.k, self.stride)


class SeparableConv2D(Layer):
    """SeparableConv2D(filters, 3, strides, 'same', use_bias=True[, activation='relu'])  cnn.py:93-125.
    depthwise 3x3 (multiplier 1) then pointwise 1x1 + bias."""
    params = ("depthwise_kernel", "pointwise_kernel", "bias")

    def __init__(self, cin, cout, stride=1, relu=False, k=3):
        self.cin, self.cout, self.stride, self.relu, self.k = cin, cout, stride, relu, k
        self.depthwise_kernel = np.zeros((k, k, cin, 1))
        self.pointwise_kernel = np.zeros((1, 1, cin, cout))
        self.bias = np.zeros((cout,))

    def forward(self, x, training):
        cols, _ = im2col(x, self.k, self.k, self.stride)           # (B,Ho,Wo,k*k*C)
        B, Ho, Wo, _ = cols.shape
        c5 = cols.reshape(B, Ho, Wo, self.k * self.k, self.cin)
        dw = np.einsum("bhwtc,tc->bhwc", c5, self.depthwise_kernel.reshape(self.k * self.k, self.cin))
        y = dw @ self.pointwise_kernel.reshape(self.cin, self.cout) + self.bias
        if self.relu:
            y = np.maximum(y, 0)
        self.cache = (x.shape, c5, dw, y)
        return y

    def backward(self, dy):
        x_shape, c5, dw, y = self.cache
        if self.relu:
            dy = dy * (y > 0)
        P = self.pointwise_kernel.reshape(self.cin, self.cout)
        D = self.depthwise_kernel.reshape(self.k * self.k, self.cin)
        ddw = dy @ P.T
        self.grads = {
            "bias": dy.reshape(-1, self.cout).sum(0),
            "pointwise_kernel": (dw.reshape(-1, self.cin).T @ dy.reshape(-1, self.cout)).reshape(self.pointwise_kernel.shape),
            "depthwise_kernel": np.einsum("bhwtc,bhwc->tc", c5, ddw).reshape(self.depthwise_kernel.shape),
        }
        dcols = (ddw[:, :, :, None, :] * D[None, None, None]).reshape(c5.shape[0], c5.shape[1], c5.shape[2], -1)
        return col2im(dcols, x_shape, self.k, self.k, self.stride)


class BatchNorm(Layer):
    """BatchNormalization() defaults: axis=-1, momentum 0.99, epsilon 1e-3."""
    params = ("gamma", "beta")
    state = ("moving_mean", "moving_variance")

    def __init__(self, c):
        self.c = c
        self.gamma, self.beta = np.ones((c,)), np.zeros((c,))
        self.moving_mean, self.moving_variance = np.zeros((c,)), np.ones((c,))

    def forward(self, x, training):
        if training:
            flat = x.reshape(-1, self.c)
            n = flat.shape[0]
            mean = flat.mean(0)
            var = flat.var(0)                                      # biased, used to normalise
            self.moving_mean = self.moving_mean * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
            unbiased = var * (n / max(n - 1, 1))                   # fused-BN moving update uses Bessel's correction
            self.moving_variance = self.moving_variance * BN_MOMENTUM + unbiased * (1 - BN_MOMENTUM)
        else:
            mean, var = self.moving_mean, self.moving_variance
        inv = 1.0 / np.sqrt(var + BN_EPS)
        xhat = (x - mean) * inv
        self.cache = (xhat, inv, training)
        return xhat * self.gamma + self.beta

    def backward(self, dy):
        xhat, inv, training = self.cache
        flat_dy, flat_xh = dy.reshape(-1, self.c), xhat.reshape(-1, self.c)
        self.grads = {"gamma": (flat_dy * flat_xh).sum(0), "beta": flat_dy.sum(0)}
        if not training:
            return dy * self.gamma * inv
        n = flat_dy.shape[0]
        return (self.gamma * inv) * (dy - self.grads["beta"] / n - xhat * self.grads["gamma"] / n)


class ReLU6(Layer):
    def forward(self, x, training):
        self.mask = (x > 0) & (x < 6)
        return np.minimum(np.maximum(x, 0), 6)

    def backward(self, dy):
        return dy * self.mask


class MaxPool2(Layer):
    """MaxPooling2D(): 2x2 window, stride 2, 'valid' (odd sizes floor).  The gradient goes to the FIRST maximum in
    row-major window order (ties happen after ReLU6 saturates at 0 or 6)."""

    def forward(self, x, training):
        B, H, W, C = x.shape
        Ho, Wo = H // 2, W // 2
        win = np.stack([x[:, 0:2 * Ho:2, 0:2 * Wo:2], x[:, 0:2 * Ho:2, 1:2 * Wo:2],
                        x[:, 1:2 * Ho:2, 0:2 * Wo:2], x[:, 1:2 * Ho:2, 1:2 * Wo:2]], 0)
        self.arg = win.argmax(0)
        self.shape = x.shape
        return win.max(0)

    def backward(self, dy):
        B, H, W, C = self.shape
        Ho, Wo = H // 2, W // 2
        dx = np.zeros(self.shape, dy.dtype)
        for t, (i, j) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            dx[:, i:2 * Ho:2, j:2 * Wo:2] = dy * (self.arg == t)
        return dx


class Flatten(Layer):
    def forward(self, x, training):
        self.shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, dy):
        return dy.reshape(self.shape)


def dropout_keep(seed, n, rate):
    """Counter-based keep decision shared bit-for-bit with the HIP kernels (csrc/kws_rng.h):
    h = fmix32(index ^ seed_lo) mixed with seed_hi; keep iff (h >> 8) * 2^-24 >= rate."""
    idx = np.arange(n, dtype=np.uint64)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    h = (idx ^ lo) & M
    h = (h + hi * np.uint64(0x9E3779B9)) & M
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    u = (h >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)
    return u >= rate


class Dropout(Layer):
    """Dropout(rate): inverted dropout, training only.  `mask` may be injected (values 0 or 1/(1-rate))."""

    def __init__(self, rate):
        self.rate = rate
        self.seed = None
        self.mask = None

    def forward(self, x, training):
        if not training or self.rate == 0 or (self.seed is None and self.mask is None):
            self.m = None
            return x
        if self.mask is not None:
            self.m = self.mask.reshape(x.shape)
        else:
            self.m = dropout_keep(self.seed, x.size, self.rate).reshape(x.shape) / (1.0 - self.rate)
        return x * self.m

    def backward(self, dy):
        return dy if self.m is None else dy * self.m


class Dense(Layer):
    params = ("kernel", "bias")

    def __init__(self, cin, cout):
        self.kernel, self.bias = np.zeros((cin, cout)), np.zeros((cout,))

    def forward(self, x, training):
        self.x = x
        return x @ self.kernel + self.bias

    def backward(self, dy):
        self.grads = {"kernel": self.x.T @ dy, "bias": dy.sum(0)}
        return dy @ self.kernel.T


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class GRU(Layer):
    """GRU(units, activation='linear', dropout=rate), rnn.py:34-35; Keras v2 defaults: reset_after=True (bias (2,3u)),
    recurrent_activation sigmoid, gate order z,r,h, last state returned, one input-dropout mask per sample shared
    by all timesteps (implementation=2)."""
    params = ("kernel", "recurrent_kernel", "bias")

    def __init__(self, cin, units, dropout=0.2):
        self.cin, self.u, self.rate = cin, units, dropout
        self.kernel = np.zeros((cin, 3 * units))
        self.recurrent_kernel = np.zeros((units, 3 * units))
        self.bias = np.zeros((2, 3 * units))
        self.seed = None
        self.mask = None

    def forward(self, x, training):
        B, T, _ = x.shape
        u = self.u
        m = None
        if training and self.rate > 0 and (self.seed is not None or self.mask is not None):
            m = self.mask if self.mask is not None else \
                dropout_keep(self.seed, B * self.cin, self.rate).reshape(B, self.cin) / (1.0 - self.rate)
            x = x * m[:, None, :]
        h = np.zeros((B, u), x.dtype)
        steps = []
        for t in range(T):
            mx = x[:, t] @ self.kernel + self.bias[0]
            mh = h @ self.recurrent_kernel + self.bias[1]
            z = sigmoid(mx[:, :u] + mh[:, :u])
            r = sigmoid(mx[:, u:2 * u] + mh[:, u:2 * u])
            hh = mx[:, 2 * u:] + r * mh[:, 2 * u:]            # activation='linear': no tanh
            h_new = z * h + (1 - z) * hh
            steps.append((h, z, r, hh, mh[:, 2 * u:]))
            h = h_new
        self.cache = (x, m, steps)
        return h

    def backward(self, dh):
        x, m, steps = self.cache
        B, T, _ = x.shape
        u = self.u
        gk, gr, gb = np.zeros_like(self.kernel), np.zeros_like(self.recurrent_kernel), np.zeros_like(self.bias)
        dx = np.zeros_like(x)
        for t in range(T - 1, -1, -1):
            h_prev, z, r, hh, mh_h = steps[t]
            dz = dh * (h_prev - hh)
            dhh = dh * (1 - z)
            dh_prev = dh * z
            dr = dhh * mh_h
            dmh_h = dhh * r
            dz_pre = dz * z * (1 - z)
            dr_pre = dr * r * (1 - r)
            dmx = np.concatenate([dz_pre, dr_pre, dhh], 1)
            dmh = np.concatenate([dz_pre, dr_pre, dmh_h], 1)
            gk += x[:, t].T @ dmx
            gr += h_prev.T @ dmh
            gb[0] += dmx.sum(0)
            gb[1] += dmh.sum(0)
            dx[:, t] = dmx @ self.kernel.T
            dh = dh_prev + dmh @ self.recurrent_kernel.T
        self.grads = {"kernel": gk, "recurrent_kernel": gr, "bias": gb}
        return dx if m is None else dx * m[:, None, :]


class LSTM(Layer):
    """LSTM(units, activation='tanh', dropout=rate), rnn.py:70-71; gate order i,f,c,o; recurrent sigmoid."""
    params = ("kernel", "recurrent_kernel", "bias")

    def __init__(self, cin, units, dropout=0.2):
        self.cin, self.u, self.rate = cin, units, dropout
        self.kernel = np.zeros((cin, 4 * units))
        self.recurrent_kernel = np.zeros((units, 4 * units))
        self.bias = np.zeros((4 * units,))
        self.seed = None
        self.mask = None

    def forward(self, x, training):
        B, T, _ = x.shape
        u = self.u
        m = None
        if training and self.rate > 0 and (self.seed is not None or self.mask is not None):
            m = self.mask if self.mask is not None else \
                dropout_keep(self.seed, B * self.cin, self.rate).reshape(B, self.cin) / (1.0 - self.rate)
            x = x * m[:, None, :]
        h = np.zeros((B, u), x.dtype)
        c = np.zeros((B, u), x.dtype)
        steps = []
        for t in range(T):
            a = x[:, t] @ self.kernel + h @ self.recurrent_kernel + self.bias
            i, f, g, o = sigmoid(a[:, :u]), sigmoid(a[:, u:2 * u]), np.tanh(a[:, 2 * u:3 * u]), sigmoid(a[:, 3 * u:])
            c_new = f * c + i * g
            tc = np.tanh(c_new)
            steps.append((h, c, i, f, g, o, tc))
            h, c = o * tc, c_new
        self.cache = (x, m, steps)
        return h

    def backward(self, dh):
        x, m, steps = self.cache
        B, T, _ = x.shape
        gk, gr, gb = np.zeros_like(self.kernel), np.zeros_like(self.recurrent_kernel), np.zeros_like(self.bias)
        dx = np.zeros_like(x)
        dc = np.zeros_like(dh)
        for t in range(T - 1, -1, -1):
            h_prev, c_prev, i, f, g, o, tc = steps[t]
            do = dh * tc
            dc = dc + dh * o * (1 - tc * tc)
            da = np.concatenate([dc * g * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - g * g), do * o * (1 - o)], 1)
            gk += x[:, t].T @ da
            gr += h_prev.T @ da
            gb += da.sum(0)
            dx[:, t] = da @ self.kernel.T
            dh = da @ self.recurrent_kernel.T
            dc = dc * f
        self.grads = {"kernel": gk, "recurrent_kernel": gr, "bias": gb}
        return dx if m is None else dx * m[:, None, :]


# ----------------------------------------------------------------------------------------------
# models (classifier/model.py:14-46)
# ----------------------------------------------------------------------------------------------
class Model(object):
    def __init__(self, model_type, num_classes, n_features=30, feature_size=20, dtype=np.float64):
        self.model_type, self.num_classes = model_type, num_classes
        self.n_features, self.feature_size, self.dtype = n_features, feature_size, dtype
        L = []
        if model_type in ("simple_cnn", "simple_cnn_lite"):
            conv = Conv2D if model_type == "simple_cnn" else SeparableConv2D
            lite = model_type == "simple_cnn_lite"
            h, w = n_features, feature_size
            L += [conv(1, 16), BatchNorm(16), ReLU6(), MaxPool2()]
            h, w = h // 2, w // 2
            L += [conv(16, 32), BatchNorm(32), ReLU6(), MaxPool2()]
            h, w = h // 2, w // 2
            L += [conv(32, 64, stride=2, relu=lite), BatchNorm(64), ReLU6()]   # lite: activation='relu' on sepconv3 too
            h, w = -(-h // 2), -(-w // 2)
            L += [conv(64, 128, relu=True), BatchNorm(128), ReLU6(), MaxPool2()]
            h, w = h // 2, w // 2
            L += [Flatten(), Dropout(0.5), Dense(h * w * 128, 128), ReLU6()]
            feat = 128
            self.input_rank = 4
        elif model_type == "simple_gru":
            L += [GRU(feature_size, 48, 0.2)]
            feat = 48
            self.input_rank = 3
        elif model_type == "simple_lstm":
            L += [LSTM(feature_size, 48, 0.2)]
            feat = 48
            self.input_rank = 3
        else:
            raise ValueError("Unsupported model type")                       # classifier/model.py:32
        L += [Dense(feat, num_classes)]                                      # 'score_predict', softmax applied in loss
        self.layers = L

    # ---- weights in Keras get_weights() order --------------------------------------------------
    def weight_list(self):
        out = []
        for li, l in enumerate(self.layers):
            for n in l.params:
                out.append((li, n, True))
            for n in l.state:
                out.append((li, n, False))
        # Keras orders BN as gamma, beta, moving_mean, moving_variance: params then state, as above
        return out

    def get_weights(self):
        return [np.array(getattr(self.layers[li], n)) for li, n, _ in self.weight_list()]

    def set_weights(self, ws):
        wl = self.weight_list()
        assert len(ws) == len(wl)
        for (li, n, _), w in zip(wl, ws):
            cur = getattr(self.layers[li], n)
            assert cur.shape == tuple(np.shape(w)), (li, n, cur.shape, np.shape(w))
            setattr(self.layers[li], n, np.asarray(w, self.dtype).copy())

    def init_weights(self, seed=0):
        """glorot-uniform kernels, orthogonal recurrent kernels, zero biases (Keras defaults; LSTM forget bias 1)."""
        rng = np.random.default_rng(seed)
        ws = []
        for li, n, trainable in self.weight_list():
            cur = getattr(self.layers[li], n)
            if n in ("kernel", "depthwise_kernel", "pointwise_kernel"):
                shp = cur.shape
                rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
                fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                ws.append(rng.uniform(-lim, lim, shp))
            elif n == "recurrent_kernel":
                u = cur.shape[0]
                blocks = []
                for _ in range(cur.shape[1] // u):
                    q, r = np.linalg.qr(rng.standard_normal((u, u)))
                    blocks.append(q * np.sign(np.diag(r)))
                ws.append(np.concatenate(blocks, 1))
            elif n == "bias" and isinstance(self.layers[li], LSTM):
                b = np.zeros(cur.shape)
                b[self.layers[li].u:2 * self.layers[li].u] = 1.0
                ws.append(b)
            elif n in ("gamma", "moving_variance"):
                ws.append(np.ones(cur.shape))
            else:
                ws.append(np.zeros(cur.shape))
        self.set_weights(ws)
        return self

    def trainable_count(self):
        return int(sum(getattr(self.layers[li], n).size for li, n, t in self.weight_list() if t))

    # ---- compute -------------------------------------------------------------------------------
    def set_dropout(self, seed=None, mask=None):
        for l in self.layers:
            if isinstance(l, (Dropout, GRU, LSTM)):
                l.seed, l.mask = seed, mask

    def logits(self, x, training=False):
        x = np.asarray(x, self.dtype)
        if self.input_rank == 4 and x.ndim == 3:
            x = x[..., None]
        if self.input_rank == 3 and x.ndim == 4:
            x = x[..., 0]
        for l in self.layers:
            x = l.forward(x, training)
        return x

    def predict(self, x):
        return softmax(self.logits(x, False))

    def backward(self, dlogits):
        d = dlogits
        for l in reversed(self.layers):
            d = l.backward(d)
        return d

    def grad_list(self):
        """gradients of the trainable arrays, Keras trainable_weights order"""
        return [self.layers[li].grads[n] for li, n, t in self.weight_list() if t]

    def trainable_list(self):
        return [getattr(self.layers[li], n) for li, n, t in self.weight_list() if t]

    def set_trainable(self, arrays):
        it = iter(arrays)
        for li, n, t in self.weight_list():
            if t:
                setattr(self.layers[li], n, np.asarray(next(it), self.dtype))


def softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def loss_and_grad(probs, labels, class_weights=None):
    """Per-sample losses and d(mean loss)/d(logits).
    plain   : K.categorical_crossentropy on probabilities (loss.py:36): renormalise, clip to [1e-7, 1-1e-7], -log
    weighted: -log(p[label]) * w[label], no clipping (loss.py:67-71)
    Keras reduces the per-sample vector with a batch mean (train.py:75-77)."""
    B, C = probs.shape
    labels = np.asarray(labels).reshape(-1).astype(np.int64)
    onehot = np.eye(C, dtype=probs.dtype)[labels]
    py = probs[np.arange(B), labels]
    if class_weights is None:
        # float32 semantics of the clip bounds as TF applies them
        lo, hi = probs.dtype.type(CE_EPS), probs.dtype.type(1.0) - probs.dtype.type(CE_EPS)
        clipped = np.clip(py, lo, hi)
        losses = -np.log(clipped)
        live = ((py >= lo) & (py <= hi)).astype(probs.dtype)      # clip passes gradient only inside the bounds
        dlogits = (probs - onehot) * live[:, None] / B
    else:
        w = np.asarray(class_weights, probs.dtype)[labels]
        losses = -np.log(py) * w
        dlogits = (probs - onehot) * w[:, None] / B
    return losses, dlogits


def train_forward_backward(model, x, labels, class_weights=None, dropout_seed=None, dropout_mask=None):
    """One training forward + backward.  Returns (mean loss, accuracy, probs); grads via model.grad_list()."""
    model.set_dropout(dropout_seed, dropout_mask)
    z = model.logits(x, training=True)
    p = softmax(z)
    losses, dlogits = loss_and_grad(p, labels, class_weights)
    model.backward(dlogits)
    acc = float((p.argmax(-1) == np.asarray(labels).reshape(-1)).mean())
    return float(losses.mean()), acc, p


class Adam(object):
    """keras.optimizers.Adam(learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False):
       lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= lr_t m/(sqrt(v)+eps)."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0
        self.m = self.v = None

    def step(self, params, grads, lr=None):
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
            self.v = [np.zeros_like(p) for p in params]
        self.t += 1
        lr = self.lr if lr is None else lr
        lr_t = lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
            out.append(p - lr_t * self.m[i] / (np.sqrt(self.v[i]) + self.eps))
        return out


def train_step(model, opt, x, labels, class_weights=None, dropout_seed=None, lr=None):
    loss, acc, _ = train_forward_backward(model, x, labels, class_weights, dropout_seed)
    model.set_trainable(opt.step(model.trainable_list(), model.grad_list(), lr))
    return loss, acc
