"""CPU tests: the numpy model oracle (oracle/model_oracle.py) against torch-CPU ops + autograd in float64.
torch supplies conv2d / batch_norm / max_pool2d / linear and the gradients; the GRU/LSTM cells are written by hand
in torch (nn.GRU forces tanh and another gate order) and differentiated by autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model_oracle as mo
from oracle import torch_ref as tr



def torch_forward(model_type, ws, x, training, drop_mask=None):
    """the torch-operator restatement (oracle/torch_ref.py: conv2d, F.batch_norm, max_pool2d, hand-written GRU / LSTM cells)"""
    return tr.forward(model_type, ws, x, training, drop_mask)


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


@pytest.mark.parametrize("model_type,weighted", [("simple_cnn", False), ("simple_cnn", True), ("simple_cnn_lite", False), ("simple_gru", False),
                                                 ("simple_lstm", False)])
def test_multi_step_training_vs_torch_reference(model_type, weighted):
    """Six optimizer steps of the numpy oracle against oracle/torch_ref.py, which shares no code with it: torch's own
    F.batch_norm updates the moving statistics (momentum 0.99, epsilon 1e-3, unbiased variance), autograd supplies the
    gradients and a hand-written Keras-form Adam (epsilon outside the bias correction) the update.  Loss per step, every
    trainable tensor and every moving statistic agree to float64 rounding."""
    C, B = 6, 12
    m, _, _ = make(model_type, C, seed=4, B=B)
    flags = [t for _, _, t in m.weight_list()]
    tm = tr.TorchModel(model_type, m.get_weights(), flags)
    opt_np, opt_t = mo.Adam(1e-3), tr.KerasAdam(1e-3)
    rng = np.random.default_rng(21)
    cw = np.array([0.3] + [0.7 / (C - 1)] * (C - 1)) if weighted else None
    rate, width = (0.5, 256) if "cnn" in model_type else (0.2, 20)
    for step in range(6):
        x = rng.standard_normal((B, 30, 20)) * 3.0
        y = rng.integers(0, C, B)
        mask = (rng.uniform(size=(B, width)) >= rate) / (1 - rate)
        loss_np, _, p_np = mo.train_forward_backward(m, x, y, cw, dropout_mask=mask)
        m.set_trainable(opt_np.step(m.trainable_list(), m.grad_list()))
        loss_t, p_t, _ = tm.train_step(opt_t, x, y, cw, mask)
        assert abs(loss_np - loss_t) < 1e-10, (step, loss_np, loss_t)
        np.testing.assert_allclose(p_np, p_t.numpy(), atol=1e-10)
    for a, b, (li, n, t) in zip(m.get_weights(), tm.get_weights(), m.weight_list()):
        np.testing.assert_allclose(a, b, atol=1e-9, rtol=1e-9, err_msg="layer %d %s after 6 steps" % (li, n))
    x = rng.standard_normal((B, 30, 20)) * 3.0
    np.testing.assert_allclose(m.predict(x), tm.predict(x), atol=1e-10)     # inference with the updated moving statistics


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


def test_tie_aware_matching_finds_a_flipped_decision():
    """tests/tie_aware.py (used by the GPU parity tests): a gradient computed with ONE near-threshold decision resolved the other
    way is matched by the corresponding alternative and by no other, and reported; an arbitrary perturbation matches nothing."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tie_aware import TieAwareOracle
    m, x, y = make("simple_cnn", 6, seed=3, B=16)
    tao = TieAwareOracle(m, x, y, eps=2e-3, max_candidates=6)          # a wide eps so that this small batch has candidates
    assert len(tao.candidates) >= 2 and tao.n_near_ties >= len(tao.candidates)
    alts = list(tao.alternatives())
    assert alts[0][0] == "baseline"
    ok, label, err, base = tao.match(tao.base, 1e-9)
    assert ok and label == "baseline" and err == 0.0
    # (a gate under a pool window's losing element carries no gradient: flipping it changes nothing; take one that does)
    change = [max(np.abs(a - b).max() for a, b in zip(g, tao.base)) for _, g in alts]
    k = next(i for i in range(1, len(alts)) if change[i] > 1e-6)
    ok, label, err, base = tao.match(alts[k][1], 1e-9)
    assert ok and base > 1e-6 and label != "baseline"
    assert max(np.abs(a - b).max() for a, b in zip(alts[k][1], dict(alts)[label])) == 0.0
    bad = [g * 1.01 for g in tao.base]
    ok, label, err, base = tao.match(bad, 1e-4)
    assert not ok
    # the oracle is left with its baseline gradients
    for g, b in zip(m.grad_list(), tao.base):
        np.testing.assert_array_equal(g, b)
