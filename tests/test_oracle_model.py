"""CPU tests: the numpy model oracle (oracle/model_oracle.py) against torch-CPU ops + autograd in float64.
torch supplies conv2d / batch_norm / max_pool2d / linear and the gradients; the GRU/LSTM cells are written by hand
in torch (nn.GRU forces tanh and another gate order) and differentiated by autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model_oracle as mo



def tf_same_conv(x, w, stride):
    """x NHWC, w HWIO -> NHWC, TF SAME padding (extra pad at bottom/right)."""
    B, H, W, C = x.shape
    kh, kw = w.shape[:2]
    _, pt, pb = mo.same_pad(H, kh, stride)
    _, pl, pr = mo.same_pad(W, kw, stride)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xt, w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)


def torch_forward(model_type, ws, x, training, drop_mask=None):
    """ws: list of torch tensors in Keras get_weights() order."""
    it = iter(ws)
    nxt = lambda: next(it)
    if model_type in ("simple_cnn", "simple_cnn_lite"):
        lite = model_type == "simple_cnn_lite"
        h = x[..., None] if x.dim() == 3 else x
        cfg = [(1, 16, 1, False, True), (16, 32, 1, False, True), (32, 64, 2, lite, False), (64, 128, 1, True, True)]
        for cin, cout, s, relu, pool in cfg:
            if lite:
                dw, pw, b = nxt(), nxt(), nxt()
                B, H, W, C = h.shape
                _, pt, pb = mo.same_pad(H, 3, s)
                _, pl, pr = mo.same_pad(W, 3, s)
                ht = F.pad(h.permute(0, 3, 1, 2), (pl, pr, pt, pb))
                ht = F.conv2d(ht, dw.permute(2, 3, 0, 1), stride=s, groups=cin)
                h = F.conv2d(ht, pw.permute(3, 2, 0, 1)).permute(0, 2, 3, 1) + b
            else:
                h = tf_same_conv(h, nxt(), s)
            if relu:
                h = F.relu(h)
            g, bt, mm, mv = nxt(), nxt(), nxt(), nxt()
            if training:
                flat = h.reshape(-1, cout)
                mean, var = flat.mean(0), flat.var(0, unbiased=False)
            else:
                mean, var = mm, mv
            h = (h - mean) / torch.sqrt(var + 1e-3) * g + bt
            h = torch.clamp(h, 0, 6)
            if pool:
                h = F.max_pool2d(h.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        h = h.reshape(h.shape[0], -1)
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
        h = torch.zeros((xx.shape[0], u), dtype=torch.float64)
        for t in range(xx.shape[1]):
            mx = xx[:, t] @ k + b[0]
            mh = h @ rk + b[1]
            z = torch.sigmoid(mx[:, :u] + mh[:, :u])
            r = torch.sigmoid(mx[:, u:2 * u] + mh[:, u:2 * u])
            hh = mx[:, 2 * u:] + r * mh[:, 2 * u:]
            h = z * h + (1 - z) * hh
    elif model_type == "simple_lstm":
        k, rk, b = nxt(), nxt(), nxt()
        u = rk.shape[0]
        xx = x[..., 0] if x.dim() == 4 else x
        if training and drop_mask is not None:
            xx = xx * drop_mask[:, None, :]
        h = torch.zeros((xx.shape[0], u), dtype=torch.float64)
        c = torch.zeros((xx.shape[0], u), dtype=torch.float64)
        for t in range(xx.shape[1]):
            a = xx[:, t] @ k + h @ rk + b
            i, f, g, o = torch.sigmoid(a[:, :u]), torch.sigmoid(a[:, u:2 * u]), torch.tanh(a[:, 2 * u:3 * u]), torch.sigmoid(a[:, 3 * u:])
            c = f * c + i * g
            h = o * torch.tanh(c)
    k, b = nxt(), nxt()
    return h @ k + b


def make(model_type, C=6, seed=0, B=8):
    m = mo.Model(model_type, C).init_weights(seed)
    rng = np.random.default_rng(seed + 1)
    # perturb BN params / biases away from their trivial init so their gradients are exercised
    ws = m.get_weights()
    for i, (li, n, t) in enumerate(m.weight_list()):
        if n in ("gamma", "moving_variance"):
            ws[i] = ws[i] * rng.uniform(0.5, 1.5, ws[i].shape)
        elif n in ("beta", "bias", "moving_mean"):
            ws[i] = ws[i] + 0.1 * rng.standard_normal(ws[i].shape)
    m.set_weights(ws)
    x = rng.standard_normal((B, 30, 20)) * 3.0
    y = rng.integers(0, C, B)
    return m, x, y


def test_param_counts_match_survey():
    assert mo.Model("simple_cnn", 36).trainable_count() == 134932
    assert mo.Model("simple_cnn", 5).trainable_count() == 130933
    assert mo.Model("simple_cnn_lite", 36).trainable_count() == 50045
    assert mo.Model("simple_gru", 36).trainable_count() == 11844
    with pytest.raises(ValueError, match="Unsupported model type"):
        mo.Model("resnet", 3)


def test_same_padding_table():
    assert mo.same_pad(30, 3, 1) == (30, 1, 1) and mo.same_pad(7, 3, 2) == (4, 1, 1) and mo.same_pad(5, 3, 2) == (3, 1, 1)
    assert mo.same_pad(8, 3, 2) == (4, 0, 1)  # even input, stride 2: the extra pad goes to the END


@pytest.mark.parametrize("model_type", ["simple_cnn", "simple_cnn_lite", "simple_gru", "simple_lstm"])
@pytest.mark.parametrize("weighted", [False, True])
def test_forward_loss_and_gradients_vs_torch_autograd(model_type, weighted):
    C = 6
    m, x, y = make(model_type, C)
    rate = 0.5 if "cnn" in model_type else 0.2
    width = 256 if "cnn" in model_type else 20
    rng = np.random.default_rng(9)
    mask = (rng.uniform(size=(len(x), width)) >= rate) / (1 - rate)
    cw = np.array([0.3] + [0.7 / (C - 1)] * (C - 1)) if weighted else None

    ws_np = m.get_weights()
    loss, acc, p = mo.train_forward_backward(m, x, y, cw, dropout_mask=mask)
    grads = m.grad_list()

    ws_t = [torch.tensor(w, requires_grad=True) for w in ws_np]
    z = torch_forward(model_type, ws_t, torch.tensor(x), True, torch.tensor(mask))
    pt = torch.softmax(z, -1)
    py = pt[torch.arange(len(y)), torch.tensor(y)]
    if weighted:
        lt = (-torch.log(py) * torch.tensor(cw)[torch.tensor(y)]).mean()
    else:
        lt = (-torch.log(torch.clamp(py / pt.sum(-1), 1e-7, 1 - 1e-7))).mean()
    lt.backward()
    np.testing.assert_allclose(p, pt.detach().numpy(), atol=1e-10)
    assert abs(loss - lt.item()) < 1e-10
    tg = [w.grad.numpy() for w, (_, _, t) in zip(ws_t, m.weight_list()) if t]
    assert len(tg) == len(grads)
    for g, t, (li, n, _) in zip(grads, tg, [w for w in m.weight_list() if w[2]]):
        np.testing.assert_allclose(g, t, atol=1e-9, rtol=1e-8, err_msg="layer %d %s" % (li, n))


def test_inference_forward_uses_moving_stats():
    m, x, y = make("simple_cnn", 5)
    ws = [torch.tensor(w) for w in m.get_weights()]
    want = torch.softmax(torch_forward("simple_cnn", ws, torch.tensor(x), False), -1).numpy()
    np.testing.assert_allclose(m.predict(x), want, atol=1e-12)


def test_bn_moving_stats_update():
    m, x, y = make("simple_cnn", 5)
    bn = m.layers[1]
    mm0, mv0 = bn.moving_mean.copy(), bn.moving_variance.copy()
    z1 = m.layers[0].forward(x[..., None], True)
    mo.train_forward_backward(m, x, y)
    flat = z1.reshape(-1, 16)
    n = flat.shape[0]
    np.testing.assert_allclose(bn.moving_mean, mm0 * 0.99 + flat.mean(0) * 0.01, atol=1e-12)
    np.testing.assert_allclose(bn.moving_variance, mv0 * 0.99 + flat.var(0) * n / (n - 1) * 0.01, atol=1e-12)


def test_cross_entropy_clip_edges():
    p = np.array([[1.0 - 1e-9, 1e-9], [1e-9, 1.0 - 1e-9], [0.3, 0.7]])
    losses, d = mo.loss_and_grad(p, [0, 0, 0])
    np.testing.assert_allclose(losses, [-np.log(1 - 1e-7), -np.log(1e-7), -np.log(0.3)], rtol=1e-12)
    assert np.all(d[0] == 0) and np.all(d[1] == 0)                    # outside the clip range: no gradient
    np.testing.assert_allclose(d[2], np.array([0.3 - 1, 0.7]) / 3)


def test_adam_is_the_keras_form():
    rng = np.random.default_rng(0)
    p, opt = [rng.standard_normal(5)], mo.Adam(1e-3)
    m = v = np.zeros(5)
    w = p[0].copy()
    for t in range(1, 4):
        g = rng.standard_normal(5)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w = w - 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)  # eps OUTSIDE the correction
        p = opt.step(p, [g])
    np.testing.assert_allclose(p[0], w, atol=1e-15)
    # differs from the PyTorch form (eps inside) when gradients are tiny
    g = np.full(5, 1e-9)
    o1 = mo.Adam(1e-3).step([np.zeros(5)], [g])[0]
    assert abs(o1[0]) < 1e-3 * 0.5


def test_dropout_hash_statistics_and_determinism():
    k = mo.dropout_keep(0x1234567ABCDEF, 1 << 16, 0.5)
    assert abs(k.mean() - 0.5) < 0.01
    assert np.array_equal(k, mo.dropout_keep(0x1234567ABCDEF, 1 << 16, 0.5))
    assert not np.array_equal(k, mo.dropout_keep(0x1234567ABCDEE, 1 << 16, 0.5))
    assert abs(mo.dropout_keep(7, 1 << 16, 0.2).mean() - 0.8) < 0.01


def test_training_reduces_loss_on_separable_task():
    rng = np.random.default_rng(3)
    C, B = 4, 64
    protos = rng.standard_normal((C, 30, 20)) * 2
    y = rng.integers(0, C, B)
    x = protos[y] + 0.3 * rng.standard_normal((B, 30, 20))
    for mt in ("simple_cnn", "simple_gru"):
        m = mo.Model(mt, C).init_weights(1)
        opt = mo.Adam(1e-2 if mt == "simple_gru" else 1e-3)
        l0 = None
        for it in range(25):
            loss, acc = mo.train_step(m, opt, x, y, dropout_seed=it)
            l0 = loss if l0 is None else l0
        assert loss < 0.6 * l0, (mt, l0, loss)
