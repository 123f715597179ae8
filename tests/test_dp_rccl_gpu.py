"""GPU tests of the data-parallel exchange behind the C ABI (csrc/kws_comm.hip: RCCL, two buckets, the early one enqueued by the
train step itself on its side stream, under the rest of the backward pass -- kws_train_args.comm).  A one-GPU box can only form a ONE-rank communicator, which runs exactly
the code path of N ranks (RCCL's one-rank all-reduce is the identity), so what is asserted is: the overlapped exchange
with HIP-computed gradients leaves bit-identical gradients, BatchNormalization statistics and post-Adam weights compared
with the plain single-process step, and `KWSModel.fit` through the exchange trains exactly like `fit` without it.
The N-rank arithmetic (sharding, weights, reduction) is covered by tests/test_dp_gloo.py on CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def comm(torch):
    from kws_amd.parallel import KwsComm
    c = KwsComm.single()
    yield c
    c.close()


def _model(model_type, C, seed=0):
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec
    spec = ModelSpec(model_type, C, 30, 20)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=seed))
    return dm


def _batch(B, C, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, 30, 20)) * 3.0).astype(np.float32)
    x[..., 0] -= 10.0
    return x, rng.integers(0, C, B).astype(np.int32)


def test_comm_binds_rccl_and_reduces(torch, comm):
    assert comm.world == 1 and comm.rank == 0
    assert comm.rccl_version >= 21000                     # an NCCL-style version code of the bound librccl
    for dtype in (torch.float32, torch.float64, torch.int32, torch.int64):
        t = torch.arange(1000, device="cuda").to(dtype)
        for op in ("sum", "max", "avg"):
            if op == "avg" and not dtype.is_floating_point:
                continue
            comm.allreduce(t, op)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000).to(dtype))
    from kws_amd import lib as L
    with pytest.raises(L.KwsError):                        # unknown dtype code
        L.check(L.get_lib().kws_comm_allreduce(comm._h, t.data_ptr(), 4, 99, 0, None))


@pytest.mark.parametrize("model_type", ["simple_cnn", "simple_cnn_lite", "simple_gru"])
def test_overlapped_exchange_equals_plain_step(torch, comm, model_type):
    """three train steps: (a) plain, (b) with the exchange inside the step (kws_train_args.comm: early bucket on the model's side
    stream behind conv4's weight gradient, late bucket + BN statistics grouped on the caller's stream), (c) plain step followed by
    the exchange as a call of its own (kws_allreduce_grads, the form a rank with an empty shard uses): same bits everywhere"""
    C, B = 12, 96
    a, b, c = _model(model_type, C, 3), _model(model_type, C, 3), _model(model_type, C, 3)
    det = model_type != "simple_gru"                       # the recurrent models have no fixed-order mode: compare to 1e-5
    if det:
        for m in (a, b, c):
            m.set_deterministic(True)
    split = b.grad_split
    assert (split > 0) == (model_type != "simple_gru")
    comm.timing(True)
    for step in range(3):
        x, y = _batch(B, C, 10 + step)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        a.train_fwd_bwd(xd, yd, dropout_seed=77 + step)
        ga = a.grads.clone()
        a.adam_step(1e-3)
        b.train_fwd_bwd(xd, yd, dropout_seed=77 + step, grad_scale=1.0, comm=comm, comm_state_weight=1.0)
        gb = b.grads.clone()                                # stream-ordered behind the exchange
        b.adam_step(1e-3)
        torch.cuda.synchronize()
        early, late = comm.last_us()
        assert late is not None and late > 0 and ((early is not None and early > 0) == (split > 0))
        c.train_fwd_bwd(xd, yd, dropout_seed=77 + step)
        comm.allreduce_grads(c.grads, split, c.state if c.spec.state_count else None, 1.0)
        gc = c.grads.clone()
        c.adam_step(1e-3)
        torch.cuda.synchronize()
        if det:
            assert torch.equal(ga, gb) and torch.equal(ga, gc), "gradients differ at step %d" % step
            assert torch.equal(a.params, b.params) and torch.equal(a.state, b.state)
            assert torch.equal(a.params, c.params) and torch.equal(a.state, c.state)
        else:
            scale = float(ga.abs().max())
            assert float((ga - gb).abs().max()) <= 1e-5 * scale and float((ga - gc).abs().max()) <= 1e-5 * scale
    comm.timing(False)
    if det:
        assert float(a.stats[0]) == float(b.stats[0])
    else:
        assert abs(float(a.stats[0]) - float(b.stats[0])) <= 1e-5 * abs(float(a.stats[0]))


def test_state_weight_scales_the_moving_statistics(torch, comm):
    """the late bucket carries state * weight (the batch-weighted mean of the replicas' BatchNormalization statistics)"""
    g = torch.randn(1000, device="cuda")
    st = torch.arange(1, 9, device="cuda", dtype=torch.float32)
    g0, st0 = g.clone(), st.clone()
    comm.allreduce_grads(g, 0, st, 0.25)
    torch.cuda.synchronize()
    assert torch.equal(g, g0) and torch.equal(st, st0 * 0.25)
    from kws_amd import lib as L
    with pytest.raises(L.KwsError):
        L.check(L.get_lib().kws_allreduce_grads(comm._h, g.data_ptr(), 10, 11, None, 0, 1.0, None))   # split > n


def test_fit_through_the_exchange_equals_plain_fit(torch, comm):
    """KWSModel.fit with a forced one-rank DataParallel (bucket event, weights from shard_plan, per-step statistics
    exchange, partial last batch) reproduces plain fit bit for bit in the deterministic gradient mode."""
    from classifier.loss import SparseCategoricalCrossEntropy
    from classifier.model import KWSModel
    from common.model_utils import get_optimizer
    from kws_amd.parallel import DataParallel
    C, N = 4, 150
    rng = np.random.default_rng(5)
    protos = rng.standard_normal((C, 30, 20)) * 2
    y = rng.integers(0, C, N)
    x = (protos[y] + 0.5 * rng.standard_normal((N, 30, 20))).astype(np.float32)[..., None]
    hist, weights = [], []
    for dp in (None, DataParallel(comm=comm, force=True)):
        torch.manual_seed(1234)
        m = KWSModel("simple_cnn", C, seed=3)
        m._device().set_deterministic(True)
        m.compile(optimizer=get_optimizer("adam", 1e-3), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
        h = m.fit(x, y, batch_size=64, epochs=2, verbose=0, shuffle=True, data_parallel=dp)
        hist.append(h.history["loss"])
        weights.append(m.get_weights())
    assert hist[0] == hist[1]
    for wa, wb in zip(*weights):
        np.testing.assert_array_equal(wa, wb)
