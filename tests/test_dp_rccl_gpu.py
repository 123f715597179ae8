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


def test_comm_broadcast_one_rank(torch, comm):
    """kws_comm_broadcast (what fit() ships the initial weights and the epoch permutation with): byte-wise, any dtype; bad roots refused"""
    from kws_amd import lib as L
    for dtype in (torch.float32, torch.int64, torch.uint8):
        t = (torch.arange(1001, device="cuda") % 251).to(dtype)
        want = t.clone()
        comm.broadcast(t, 0)
        torch.cuda.synchronize()
        assert torch.equal(t, want)
    with pytest.raises(L.KwsError):
        comm.broadcast(t, 1)                                 # root outside the world


def _two_rank_worker(rank, world, port, out_dir):
    """one process per GPU: the real N > 1 path (kws_comm over RCCL) with uneven shards incl. an empty one"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tf-keras-speech-commands_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # control plane only; the data path is kws_comm
    from kws_amd.parallel import DataParallel, KwsComm
    comm = KwsComm.from_torch_group()
    dp = DataParallel(comm=comm)
    C = 12
    dm = _model("simple_cnn", C, 3 + rank)                  # different initial weights per rank: the broadcast must equalise them
    dm.set_deterministic(True)
    dp.broadcast_(dm.params)
    dp.broadcast_(dm.state)
    res = {}
    for step, n_global in enumerate((96, 7, 1)):            # equal shards, uneven shards, one clip (rank 1's shard is empty)
        x, y = _batch(n_global, C, 20 + step)
        lo, hi, wgt = dp.shard_plan(n_global)
        if hi > lo:
            dm.train_fwd_bwd(torch.from_numpy(x[lo:hi]).cuda(), torch.from_numpy(y[lo:hi]).cuda(), dropout_seed=0, grad_scale=wgt, comm=comm,
                             comm_state_weight=wgt)
        else:
            dm.grads.zero_()
            comm.allreduce_grads(dm.grads, dm.grad_split, dm.state, wgt)
        res["g%d" % step] = dm.grads.cpu().numpy().copy()
        dm.adam_step(1e-3)
    res["params"], res["state"] = dm.params.cpu().numpy(), dm.state.cpu().numpy()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), **res)
    torch.cuda.synchronize()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_rccl(torch, tmp_path):
    """Needs two GPUs (the driver's multi-GPU node; skipped on a one-GPU box).  Two processes, one communicator each, uneven and empty
    shards: every rank ends every step with identical summed gradients, and identical weights / moving statistics after Adam."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.abs(a["g0"]).max() > 0 and np.abs(a["g2"]).max() > 0
