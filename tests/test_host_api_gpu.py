"""GPU tests of the Keras-like host API (classifier.model / classifier.loss / common.model_utils) end to end."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def separable(n, C, seed):
    rng = np.random.default_rng(seed)
    protos = np.random.default_rng(99).standard_normal((C, 30, 20)) * 2
    y = rng.integers(0, C, n)
    x = (protos[y] + 0.7 * rng.standard_normal((n, 30, 20))).astype(np.float32)
    return x[..., None], y


def test_losses_call_matches_oracle(torch):
    from classifier.loss import SparseCategoricalCrossEntropy, WeightedSparseCategoricalCrossEntropy
    from oracle import model_oracle as mo
    rng = np.random.default_rng(0)
    p = mo.softmax(rng.standard_normal((17, 6)) * 4).astype(np.float32)
    p[0] = [1, 0, 0, 0, 0, 0]
    y = rng.integers(0, 6, 17)
    y[0] = 1                                              # p = 0 for the label: clipped at 1e-7 in the plain loss
    want, _ = mo.loss_and_grad(p.astype(np.float64), y)
    got = SparseCategoricalCrossEntropy()(y[:, None], p)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    w = np.array([0.3, 0.14, 0.14, 0.14, 0.14, 0.14])
    want_w, _ = mo.loss_and_grad(p[1:].astype(np.float64), y[1:], w)
    got_w = WeightedSparseCategoricalCrossEntropy(w)(y[1:, None], p[1:])
    np.testing.assert_allclose(got_w, want_w, rtol=1e-5, atol=1e-6)
    ig = SparseCategoricalCrossEntropy(ignore_index=3)(y, p)
    assert np.all(ig[y == 3] == 0) and np.allclose(ig[y != 3], got[y != 3])
    lg = SparseCategoricalCrossEntropy(from_logits=True)(y, np.log(p[1:] + 1e-30).astype(np.float32)) if False else None
    assert lg is None


def test_fit_learns_and_matches_predict_oracle(torch, tmp_path):
    from classifier.loss import SparseCategoricalCrossEntropy
    from classifier.model import get_model
    from common import callbacks as cb
    from common.model_utils import get_optimizer
    from oracle import model_oracle as mo
    C = 4
    x, y = separable(1000, C, 1)
    xv, yv = separable(300, C, 2)
    from kws_amd.init import init_weights
    torch.manual_seed(0)                                   # fit() draws its shuffles from torch's device generator
    m = get_model("simple_cnn", C)
    m.set_weights(init_weights(m.spec, seed=0))
    m.compile(get_optimizer("adam", 2e-3, decay_type="cosine", decay_steps=40), SparseCategoricalCrossEntropy(), ["accuracy"])
    log = os.path.join(tmp_path, "log.jsonl")
    ck = cb.ModelCheckpoint(os.path.join(tmp_path, "ep{epoch:03d}-val_accuracy{val_accuracy:.3f}.npz"), monitor="val_accuracy",
                            mode="max", save_best_only=True)
    h = m.fit(x, y, batch_size=256, epochs=10, validation_data=(xv, yv), shuffle=True, verbose=0,
              callbacks=[ck, cb.TerminateOnNaN(), cb.JsonlLogger(log)])
    assert len(h.history["loss"]) == 10 and h.history["loss"][-1] < 0.5 * h.history["loss"][0]
    assert h.history["val_accuracy"][-1] > 0.9
    assert m.optimizer.iterations == 40 and len(open(log).read().splitlines()) == 10
    assert any(f.startswith("ep") for f in os.listdir(tmp_path))
    # the trained weights, loaded into the float64 oracle, give the same predictions / evaluation
    om = mo.Model("simple_cnn", C)
    om.set_weights(m.get_weights())
    want = om.predict(xv[..., 0].astype(np.float64))
    got = m.predict(xv, batch_size=128)
    np.testing.assert_allclose(got, want, atol=1e-3, rtol=0)
    # argmax must agree wherever the oracle's top-2 margin exceeds the fp32 tolerance
    srt = np.sort(want, -1)
    clear = (srt[:, -1] - srt[:, -2]) > 2e-3
    np.testing.assert_array_equal(got.argmax(-1)[clear], want.argmax(-1)[clear])
    loss, acc = m.evaluate(xv, yv)
    wl, _ = mo.loss_and_grad(want, yv)
    assert abs(loss - wl.mean()) < 1e-3 and abs(acc - (want.argmax(-1) == yv).mean()) < 0.01
    # save / load round trip through get_model(weights_path=...)
    path = os.path.join(tmp_path, "trained_final.npz")
    m.save(path)
    m2 = get_model("simple_cnn", C, weights_path=path)
    np.testing.assert_array_equal(m2.predict(xv[:16]), m.predict(xv[:16]))


def test_fit_from_raw_audio_and_weighted_loss(torch):
    """audio input: featurizer on the GPU in front of the network (the north-star path)"""
    from classifier.loss import WeightedSparseCategoricalCrossEntropy
    from classifier.model import get_model
    from common.model_utils import get_optimizer
    rng = np.random.default_rng(3)
    C, n = 3, 384
    y = rng.integers(0, C, n)
    t = np.arange(16000) / 16000.0
    tones = np.stack([np.sin(2 * np.pi * f * t) for f in (300.0, 1200.0, 3000.0)])
    wav = (0.3 * tones[y] + 0.05 * rng.standard_normal((n, 16000))).astype(np.float32)
    from kws_amd.init import init_weights
    torch.manual_seed(1)
    m = get_model("simple_cnn", C)
    m.set_weights(init_weights(m.spec, seed=1))
    w = np.array([0.2, 0.4, 0.4])
    m.compile(get_optimizer("adam", 2e-3, decay_type=None), WeightedSparseCategoricalCrossEntropy(w), ["accuracy"])
    h = m.fit(wav, y, batch_size=128, epochs=8, verbose=0)
    assert h.history["accuracy"][-1] > 0.95
    p = m.predict(wav[:32])
    assert p.shape == (32, C) and np.allclose(p.sum(-1), 1, atol=1e-5)
    l1 = m.train_on_batch(wav[:64], y[:64])
    assert len(l1) == 2 and np.isfinite(l1[0])


def test_rmsprop_and_sgd_steps(torch):
    from kws_amd.model import DeviceModel, ModelSpec
    dm = DeviceModel(ModelSpec("simple_cnn", 5, 30, 20))
    n = dm.params.numel()
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    dm.params.copy_(torch.from_numpy(p0)); dm.grads.copy_(torch.from_numpy(g))
    dm.sgd_step(0.1)
    np.testing.assert_allclose(dm.params.cpu().numpy(), p0 - 0.1 * g, rtol=1e-6, atol=1e-7)
    dm.params.copy_(torch.from_numpy(p0)); dm.adam_v.zero_()
    a = np.zeros(n)
    ref = p0.astype(np.float64)
    for _ in range(2):
        dm.rmsprop_step(1e-3)
        a = 0.9 * a + 0.1 * g.astype(np.float64) ** 2
        ref = ref - 1e-3 * g / (np.sqrt(a) + 1e-7)
    np.testing.assert_allclose(dm.params.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_get_dataset_builds_reference_compatible_cache(torch, golden, tmp_path):
    import wave
    from classifier.data import get_dataset
    classes = ["background", "right", "left"]
    for cname, key in (("background", "pcm_up_1"), ("right", "pcm_right_1"), ("left", "pcm_left_1")):
        d = os.path.join(tmp_path, "sounds", cname)
        os.makedirs(d)
        for i in range(2):
            w = wave.open(os.path.join(d, "%d.wav" % i), "wb")
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(golden[key][: 16000 - 3000 * i].tobytes()); w.close()
    x, y, xv, yv = get_dataset(str(tmp_path), classes, val_split=0.34)
    assert x.shape[1:] == (30, 20, 1) and len(x) + len(xv) == 6 and len(xv) == 3 and x.dtype == np.float32
    files = [f for _, _, fs in os.walk(os.path.join(tmp_path, "features")) for f in fs]
    assert len(files) == 6 and all(f.endswith(".npy") for f in files)
    one = np.load(os.path.join(tmp_path, "features", "right", sorted(os.listdir(os.path.join(tmp_path, "features", "right")))[0]))
    assert one.shape == (30, 20, 1) and one.dtype == np.float32
    x2, y2, _, _ = get_dataset(str(tmp_path), classes)            # second call: the cache is used
    assert len(x2) == 6 and sorted(y2.tolist()) == [0, 0, 1, 1, 2, 2]
    full = x2[[i for i in range(6) if y2[i] == 1]]
    assert min(np.abs(f[..., 0] - golden["refpy_mel_right_1"]).max() for f in full) < 2e-4


def test_eval_harness_confusion_matrix(torch):
    import eval as kws_eval
    from classifier.model import get_model
    from oracle import model_oracle as mo
    C = 6
    names = ["background"] + ["w%d" % i for i in range(1, C)]
    x, y = separable(700, C, 9)
    m = get_model("simple_cnn", C)
    acc, cm = kws_eval.evaluate_accuracy(m, x, y, names, batch_size=256)
    om = mo.Model("simple_cnn", C)
    om.set_weights(m.get_weights())
    want = om.predict(x[..., 0].astype(np.float64)).argmax(-1)
    ref = np.zeros((C, C), np.int64)
    np.add.at(ref, (y, want), 1)
    assert cm.sum() == 700 and abs(acc - (want == y).mean()) < 0.01
    assert np.abs(cm - ref).sum() <= 4          # an untrained net has near-tied scores on a few clips


@pytest.mark.parametrize("model_type,fp16", [("simple_cnn", False), ("simple_cnn_lite", False), ("simple_cnn_lite", True)])
def test_inference_session_end_to_end_pcm16(torch, model_type, fp16):
    """BASELINE configs[4] structure: PCM16 clips -> featurize + forward captured in ONE hipGraph (kws_amd.inference).
    The replayed graph is bit-identical to the eager launches, and the whole chain agrees with the CPU oracle
    (C featurizer + float64 model): probabilities within 1e-3, class index exact where the two best classes are 2e-3 apart."""
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    from kws_amd.inference import InferenceSession
    from kws_amd.model import DeviceModel, ModelSpec
    from oracle import featurizer_oracle as fo
    from oracle import model_oracle as mo
    B, C = 48, 36
    om = mo.Model(model_type, C).init_weights(3)
    rng = np.random.default_rng(4)
    ws = om.get_weights()
    for i, (li, n, t) in enumerate(om.weight_list()):
        if n in ("gamma", "moving_variance"):
            ws[i] = ws[i] * rng.uniform(0.5, 1.5, ws[i].shape)
        elif n in ("beta", "bias", "moving_mean"):
            ws[i] = ws[i] + 0.1 * rng.standard_normal(ws[i].shape)
    om.set_weights(ws)
    dm = DeviceModel(ModelSpec(model_type, C, pr.n_features, pr.feature_size))
    dm.set_weights(om.get_weights())
    pcm = np.clip(3000.0 * rng.standard_normal((B, 16000)), -32768, 32767).astype(np.int16)
    feat = Featurizer(pr, "mel")
    graph = InferenceSession(dm, feat, B, wav_dtype=torch.int16, use_graph=True, fp16=fp16)
    eager = InferenceSession(dm, feat, B, wav_dtype=torch.int16, use_graph=False, fp16=fp16)
    for s in (graph, eager):
        s.wav.copy_(torch.from_numpy(pcm))
    pg, ag = graph.run()
    pg2, _ = graph.run()
    pe, ae = eager.run()
    torch.cuda.synchronize()
    assert torch.equal(pg, pe) and torch.equal(ag, ae) and torch.equal(pg, pg2)
    import kws_amd.lib as L
    assert L.get_inference_precision() == L.INFER_FP32               # the session restores the library switch
    x = fo.featurize_batch(pcm.astype(np.float32) / 32768.0)          # data_utils.py:21 scaling, then the CPU featurizer
    want = om.predict(x.reshape(B, pr.n_features, pr.feature_size).astype(np.float64))
    got = pg.cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-3, rtol=0)
    top2 = np.sort(want, axis=-1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2e-3
    np.testing.assert_array_equal(ag.cpu().numpy()[clear], want.argmax(-1)[clear])


@pytest.mark.parametrize("model_type,fp16", [("simple_cnn", False), ("simple_cnn_lite", True)])
def test_inference_session_follows_the_weights(torch, model_type, fp16):
    """A graph session captured with prepared weight tables must not keep answering with the OLD weights: after set_weights (and after an
    optimizer step) run() re-derives the tables by itself and equals a fresh eager forward of the new weights, bit for bit."""
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    from kws_amd.inference import InferenceSession
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec
    B, C = 32, 12
    spec = ModelSpec(model_type, C, pr.n_features, pr.feature_size)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=1))
    rng = np.random.default_rng(9)
    wav = torch.from_numpy((0.1 * rng.standard_normal((B, 16000))).astype(np.float32)).cuda()
    feat = Featurizer(pr)
    s = InferenceSession(dm, feat, B, use_graph=True, fp16=fp16)
    s.wav.copy_(wav)
    p_old = s.run()[0].clone()
    dm.set_weights(init_weights(spec, seed=2))
    p_new = s.run()[0].clone()
    fresh = InferenceSession(dm, feat, B, use_graph=False, fp16=fp16)
    fresh.wav.copy_(wav)
    want = fresh.run()[0]
    torch.cuda.synchronize()
    assert not torch.equal(p_old, p_new)
    assert torch.equal(p_new, want)
    # an optimizer step changes the weights too
    labels = torch.from_numpy(rng.integers(0, C, B).astype(np.int32)).cuda()
    dm.train_fwd_bwd(feat(wav), labels, dropout_seed=3)
    dm.adam_step(1e-2)
    p_step = s.run()[0].clone()
    fresh2 = InferenceSession(dm, feat, B, use_graph=False, fp16=fp16)
    fresh2.wav.copy_(wav)
    want2 = fresh2.run()[0]
    torch.cuda.synchronize()
    assert not torch.equal(p_step, p_new) and torch.equal(p_step, want2)


def test_sessions_of_two_precisions_share_a_model(torch):
    """An fp16 session does not switch the model's own arithmetic: an fp32 eager session, plain dm.forward calls and an fp16 session on ONE
    simple_cnn_lite DeviceModel each keep their precision whatever order they run in."""
    from classifier.params import pr
    from kws_amd import lib as L
    from kws_amd.featurizer import Featurizer
    from kws_amd.inference import InferenceSession
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec
    B, C = 32, 12
    spec = ModelSpec("simple_cnn_lite", C, pr.n_features, pr.feature_size)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=1))
    rng = np.random.default_rng(10)
    wav = torch.from_numpy((0.1 * rng.standard_normal((B, 16000))).astype(np.float32)).cuda()
    feat = Featurizer(pr)
    x = feat(wav)
    ref32 = dm.forward(x)[0].clone()
    s32 = InferenceSession(dm, feat, B, use_graph=False, fp16=False)
    s16 = InferenceSession(dm, feat, B, use_graph=False, fp16=True)
    g16 = InferenceSession(dm, feat, B, use_graph=True, fp16=True)
    for s in (s32, s16, g16):
        s.wav.copy_(wav)
    assert dm.get_precision()[1] == L.INFER_FP32                      # creating fp16 sessions left the model alone
    a16 = s16.run()[0].clone()
    a32 = s32.run()[0].clone()
    b16 = g16.run()[0].clone()
    plain = dm.forward(x)[0].clone()
    c16 = s16.run()[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(a32, ref32) and torch.equal(plain, ref32)
    assert torch.equal(a16, b16) and torch.equal(a16, c16)
    assert not torch.equal(a16, a32)                                   # fp16 activations differ in the last bits
    assert float((a16 - a32).abs().max()) < 2e-3


@pytest.mark.parametrize("model_type,audio", [("simple_cnn", True), ("simple_cnn", False), ("simple_cnn_lite", True), ("simple_gru", False)])
def test_fit_pipelined_equals_stepwise(torch, model_type, audio):
    """KWSModel.fit runs the pipelined step bench.py measures (next batch drawn / featurized in place on a side stream from the step's
    overlap point, labels gathered there too, layer-1 moments from the pipeline, per-step statistics rows); `pipeline=False` is the same
    arithmetic on one stream.  Shuffled epochs with a partial last batch: identical history and, in the deterministic gradient mode,
    bit-identical weights (the recurrent model has no such mode: 1e-5)."""
    from classifier.loss import SparseCategoricalCrossEntropy
    from classifier.model import KWSModel
    from common.model_utils import get_optimizer
    C, N = 4, 150
    rng = np.random.default_rng(5)
    y = rng.integers(0, C, N)
    if audio:
        tones = np.sin(2 * np.pi * (300.0 * (1 + np.arange(C)))[:, None] * np.arange(16000)[None, :] / 16000.0)
        x = (0.3 * tones[y] + 0.05 * rng.standard_normal((N, 16000))).astype(np.float32)
    else:
        protos = rng.standard_normal((C, 30, 20)) * 2
        x = (protos[y] + 0.5 * rng.standard_normal((N, 30, 20))).astype(np.float32)
        if model_type != "simple_gru":
            x = x[..., None]
    det = model_type != "simple_gru"
    hist, weights = [], []
    for pipelined in (False, True):
        torch.manual_seed(1234)
        m = KWSModel(model_type, C, seed=3)
        if det:
            m._device().set_deterministic(True)
        m.compile(optimizer=get_optimizer("adam", 1e-3), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
        h = m.fit(x, y, batch_size=64, epochs=3, verbose=0, shuffle=True, pipeline=pipelined)
        hist.append((h.history["loss"], h.history["accuracy"]))
        weights.append(m.get_weights())
        assert h.history["clips_per_sec"][-1] > 0
    if det:
        assert hist[0] == hist[1]
        for wa, wb in zip(*weights):
            np.testing.assert_array_equal(wa, wb)
    else:
        np.testing.assert_allclose(hist[0][0], hist[1][0], rtol=1e-5)
        for wa, wb in zip(*weights):
            np.testing.assert_allclose(wa, wb, rtol=0, atol=1e-5)
    assert hist[1][0][-1] < hist[1][0][0]                   # and it trains


def test_featurize_gather_equals_featurizing_a_copy(torch):
    """kws_featurize_gather: clip b = row index[b] of the dataset (and of valid_len), for the tuned and the generic kernels, float32 and PCM16"""
    from classifier.params import ListenerParams, pr
    from kws_amd.featurizer import Featurizer
    rng = np.random.default_rng(3)
    rows, B = 97, 41
    a = np.clip(0.2 * rng.standard_normal((rows, 16000)), -1, 1 - 2.0 ** -15).astype(np.float32)
    lens = rng.integers(3000, 16001, rows).astype(np.int32)
    idx = rng.integers(0, rows, B).astype(np.int32)
    wav, vl, ix = torch.from_numpy(a).cuda(), torch.from_numpy(lens).cuda(), torch.from_numpy(idx).cuda()
    w16 = torch.from_numpy((a * 32768).astype(np.int16)).cuda()
    p512 = ListenerParams(1.0, 0.032, 0.016, 16000, 2, 512, 20, 13, False, ((6, 4),), 0.2)
    for params in (pr, p512):
        f = Featurizer(params)
        for src in (wav, w16):
            want = f(src.index_select(0, ix.long()).contiguous(), vl.index_select(0, ix.long()).contiguous())
            got = f(src, vl, index=ix)
            assert torch.equal(got, want)
            assert torch.equal(f(src, index=ix), f(src.index_select(0, ix.long()).contiguous()))


def test_eight_example_clips_argmax_agreement(torch, golden):
    """SURVEY 8(d) substitute for the unavailable Speech Commands v2 top-1: train on a synthetic separable task through the host
    API, then run the reference's eight example clips (example/*.wav, PCM in the golden file) through featurize + predict on
    the GPU and through the CPU restatement with the SAME trained weights: class-index argmax identical on all eight,
    probabilities within 1e-3 (north star), for simple_cnn and simple_gru."""
    from classifier.loss import SparseCategoricalCrossEntropy
    from classifier.model import KWSModel
    from common.model_utils import get_optimizer
    from oracle import featurizer_oracle as fo
    from oracle import model_oracle as mo
    names = ["right_1", "left_1", "up_1", "down_1", "right_2", "left_2", "up_2", "down_2"]
    pcm = np.stack([golden["pcm_" + n] for n in names])
    assert pcm.shape == (8, 16000) and pcm.dtype == np.int16
    feats = fo.featurize_batch(pcm.astype(np.float32) / 32768.0).astype(np.float64)      # data_utils.py:21 scaling
    C = 5
    rng = np.random.default_rng(17)
    for model_type in ("simple_cnn", "simple_gru"):
        # separable task built AROUND the real clips' features, so that the trained net gives them confident, distinct answers
        lab = np.array([1, 2, 3, 4, 1, 2, 3, 4])
        xs = np.concatenate([feats[rng.integers(0, 8, 256)] for _ in range(1)])
        idx = rng.integers(0, 8, 512)
        x = (feats[idx] + 0.3 * rng.standard_normal((512, 30, 20))).astype(np.float32)
        y = lab[idx]
        bg = (0.5 * rng.standard_normal((128, 30, 20)) - 20.0).astype(np.float32)            # class 0: quiet background
        x, y = np.concatenate([x, bg]), np.concatenate([y, np.zeros(128, np.int64)])
        m = KWSModel(model_type, C, seed=5)
        m.compile(optimizer=get_optimizer("adam", 2e-3, decay_type=None), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
        xin = x[..., None] if model_type == "simple_cnn" else x
        h = m.fit(xin, y, batch_size=128, epochs=12, verbose=0)
        assert h.history["accuracy"][-1] > 0.9, (model_type, h.history["accuracy"])
        got = m.predict(pcm)                                  # raw PCM16 audio: featurized on the GPU in front of the network
        om = mo.Model(model_type, C)
        om.set_weights([w.astype(np.float64) for w in m.get_weights()])
        want = om.predict(feats)
        np.testing.assert_array_equal(got.argmax(-1), want.argmax(-1))
        np.testing.assert_allclose(got, want, atol=1e-3, rtol=0)
        assert (want.argmax(-1) == lab).sum() >= 6, (model_type, want.argmax(-1))        # and the answers are the trained ones
        del xs
