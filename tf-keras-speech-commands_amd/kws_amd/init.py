"""Host-side weight initialisation with the Keras defaults the reference relies on (it passes no initialisers:
classifier/models/cnn.py, rnn.py): glorot_uniform kernels, orthogonal recurrent kernels, zero biases (LSTM forget
bias 1, unit_forget_bias=True), BatchNormalization gamma 1 / beta 0 / moving_mean 0 / moving_variance 1."""
import numpy as np


def init_weights(spec, seed=None):
    """Arrays for every tensor of `spec` (kws_amd.model.ModelSpec) in Keras get_weights() order."""
    rng = np.random.default_rng(seed)
    out = []
    for t in spec.tensors:
        name, shape = t["name"].split("/")[-1], t["shape"]
        layer = t["name"].split("/")[0]
        if name in ("kernel", "depthwise_kernel", "pointwise_kernel"):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, shape)
        elif name == "recurrent_kernel":
            u = shape[0]
            blocks = []
            for _ in range(shape[1] // u):
                q, r = np.linalg.qr(rng.standard_normal((u, u)))
                blocks.append(q * np.sign(np.diag(r)))
            w = np.concatenate(blocks, 1)
        elif name == "bias" and layer.startswith("lstm"):
            w = np.zeros(shape)
            u = shape[-1] // 4
            w[u:2 * u] = 1.0
        elif name in ("gamma", "moving_variance"):
            w = np.ones(shape)
        else:
            w = np.zeros(shape)
        out.append(w.astype(np.float32))
    return out
