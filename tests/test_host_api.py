"""CPU tests of the host mirror of the reference API: model factory errors and topology, optimizer / schedule
factory, losses' constructors, callbacks, class-list loader, weight files.  No compute runs here (no GPU)."""
import math
import os

import numpy as np
import pytest


def test_get_model_unsupported_type_raises_like_the_reference():
    from classifier.model import get_model
    with pytest.raises(ValueError, match="Unsupported model type"):
        get_model("resnet", 5)


def test_simple_cnn_topology_and_summary():
    from classifier.model import get_model
    m = get_model("simple_cnn", 36)
    assert m.input_shape == (30, 20, 1) and m.input_names == ["feature_input"] and m.output_names == ["score_predict"]
    assert m.count_params() == 134932 + 480
    assert sum(l["params"] for l in m.layers) == m.count_params()
    names = [l["name"] for l in m.layers]
    assert names[:4] == ["conv2d", "batch_normalization", "re_lu", "max_pooling2d"] and names[-1] == "score_predict"
    assert [l["output_shape"] for l in m.layers if l["type"] == "MaxPooling2D"] == [(15, 10, 16), (7, 5, 32), (2, 1, 128)]
    lines = []
    m.summary(print_fn=lines.append)
    text = "\n".join(lines)
    assert "Total params: 135,412" in text and "Trainable params: 134,932" in text and "Non-trainable params: 480" in text
    assert m.weight_names[0] == "conv2d/kernel" and m.weight_names[-1] == "score_predict/bias"
    w = m.get_weights()
    assert [a.shape for a in w][:5] == [(3, 3, 1, 16), (16,), (16,), (16,), (16,)]
    lim = math.sqrt(6.0 / (9 * 1 + 9 * 16))
    assert np.abs(w[0]).max() <= lim and np.all(w[1] == 1) and np.all(w[2] == 0) and np.all(w[4] == 1)


def test_weights_roundtrip_npz(tmp_path):
    from classifier.model import get_model
    m = get_model("simple_cnn", 5)
    path = os.path.join(tmp_path, "trained_final.npz")
    m.save(path)
    m2 = get_model("simple_cnn", 5, weights_path=path)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        get_model("simple_cnn", 6).load_weights(path)
    with pytest.raises(ImportError):
        m.save(os.path.join(tmp_path, "x.h5"))


def test_compile_checks_and_no_cpu_fallback():
    import kws_amd
    from classifier.loss import SparseCategoricalCrossEntropy, WeightedSparseCategoricalCrossEntropy
    from classifier.model import get_model
    from common.model_utils import get_optimizer
    m = get_model("simple_cnn", 5)
    m.compile(optimizer=get_optimizer("adam", 1e-3, decay_type=None), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
    assert m.metrics_names == ["loss", "accuracy"] and m.loss.__name__ == "sparse_categorical_crossentropy"
    with pytest.raises(ValueError):
        m.compile(get_optimizer("adam", 1e-3, decay_type=None), WeightedSparseCategoricalCrossEntropy([0.5, 0.5]))
    with pytest.raises(TypeError):
        m.compile("adam", "mse")
    if kws_amd.device_count() == 0:
        with pytest.raises(kws_amd.KwsError):
            m.predict(np.zeros((2, 30, 20, 1), np.float32))


def test_background_bias_weights_formula():
    # train.py:65-68
    from classifier.loss import WeightedSparseCategoricalCrossEntropy
    C, bias = 5, 0.3
    w = WeightedSparseCategoricalCrossEntropy(np.array([bias] + [(1.0 - bias) / (C - 1)] * (C - 1)))
    assert w.weights.dtype == np.float32 and abs(w.weights.sum() - 1) < 1e-6 and w.__name__ == "weighted_sparse_categorical_crossentropy"


def test_optimizer_factory_and_schedules():
    from common import model_utils as mu
    assert mu.get_optimizer("Adam", 1e-3, decay_type=None).kind == "adam"
    assert mu.get_optimizer("rmsprop", 1e-3, decay_type=None).rho == 0.9
    assert mu.get_optimizer("sgd", 1e-2, decay_type=None).current_lr() == 1e-2
    with pytest.raises(ValueError, match="Unsupported optimizer type"):
        mu.get_optimizer("lamb", 1e-3)
    with pytest.raises(ValueError, match="Unsupported lr decay type"):
        mu.get_lr_scheduler(1e-3, "linear", 10)
    with pytest.raises(ValueError, match="Unsupported average type"):
        mu.get_optimizer("adam", 1e-3, average_type="ema", decay_type=None)
    S = 1000
    cos = mu.get_lr_scheduler(1e-3, "cosine", S)                      # CosineDecay(alpha=0.2)
    assert cos(0) == pytest.approx(1e-3) and cos(S) == pytest.approx(2e-4) and cos(5 * S) == pytest.approx(2e-4)
    assert cos(S // 2) == pytest.approx(1e-3 * (0.8 * 0.5 + 0.2))
    ex = mu.get_lr_scheduler(1e-3, "exponential", S)
    assert ex(S) == pytest.approx(0.9e-3) and ex(S / 2) == pytest.approx(1e-3 * 0.9 ** 0.5)
    po = mu.get_lr_scheduler(1e-3, "polynomial", S)
    assert po(0) == pytest.approx(1e-3) and po(S) == pytest.approx(1e-5) and po(S // 2) == pytest.approx(0.5 * (1e-3 + 1e-5))
    pc = mu.get_lr_scheduler(5e-3, "piecewise_constant", S)            # [500, 900, 1000] -> [1e-3, lr, lr/10, lr/100]
    assert [pc(s) for s in (0, 500, 501, 900, 901, 1000, 1001)] == [1e-3, 1e-3, 5e-3, 5e-3, 5e-4, 5e-4, 5e-5]
    opt = mu.get_optimizer("adam", 1e-3, decay_type="cosine", decay_steps=S)
    opt.iterations = S
    assert opt.current_lr() == pytest.approx(2e-4)
    with pytest.raises(TypeError):
        opt.set_lr(1.0)


def test_callbacks_logic(tmp_path):
    from common import callbacks as cb
    from common.model_utils import get_optimizer

    class Fake(object):
        stop_training = False
        optimizer = get_optimizer("adam", 1e-3, decay_type=None)
        saved = []

        def save(self, path):
            self.saved.append(path)
            open(path, "w").write("x")

    m = Fake()
    r = cb.ReduceLROnPlateau(monitor="val_accuracy", factor=0.5, mode="max", patience=2, min_lr=1e-10)
    e = cb.EarlyStopping(monitor="val_accuracy", patience=3, mode="max")
    c = cb.ModelCheckpoint(os.path.join(tmp_path, "ep{epoch:03d}-val_accuracy{val_accuracy:.3f}.npz"), monitor="val_accuracy",
                           mode="max", save_best_only=True)
    k = cb.CheckpointCleanCallBack(str(tmp_path), max_keep=1)
    t = cb.TerminateOnNaN()
    for x in (r, e, c, k, t):
        x.set_model(m)
    accs = [0.5, 0.6, 0.6, 0.6, 0.6]
    for ep, a in enumerate(accs):
        for x in (r, e, c, k, t):
            x.on_epoch_end(ep, {"val_accuracy": a, "loss": 1.0})
    assert m.optimizer.current_lr() == pytest.approx(5e-4)   # one reduction after 2 stale epochs
    assert m.stop_training                                    # 3 stale epochs
    assert len(m.saved) == 2 and len(os.listdir(tmp_path)) == 1
    m.stop_training = False
    t.on_epoch_end(9, {"loss": float("nan")})
    assert m.stop_training


def test_get_classes(tmp_path):
    from common.utils import get_classes
    p = os.path.join(tmp_path, "classes.txt")
    open(p, "w").write("background\n yes \nno\n")
    assert get_classes(p) == ["background", "yes", "no"]


def test_parallel_helper_single_process():
    import torch
    from kws_amd.parallel import DataParallel
    dp = DataParallel()
    assert not dp.active and dp.world == 1 and dp.grad_scale == 1.0 and dp.shard(10) == (0, 10)
    g = torch.ones(8)
    dp.sync_grads(g)
    assert torch.equal(g, torch.ones(8))


def _write_wav(path, data, rate, width=2):
    import wave
    w = wave.open(str(path), "wb")
    w.setnchannels(data.shape[1] if data.ndim == 2 else 1)
    w.setsampwidth(width)
    w.setframerate(rate)
    w.writeframes(data.tobytes())
    w.close()


def test_load_wav_follows_the_librosa_load_contract(tmp_path):
    """common/data_utils.py:93 loads with librosa.load(sr=16000, mono=True): float32 in [-1, 1), channels averaged, other sample rates
    resampled to 16 kHz (length ceil(n sr / rate)).  host only: no GPU involved."""
    from common.data_utils import load_wav
    rng = np.random.default_rng(0)
    # 16 kHz PCM16 mono: exact int16 / 32768
    pcm = rng.integers(-20000, 20000, 16000).astype("<i2")
    _write_wav(tmp_path / "a.wav", pcm, 16000)
    a = load_wav(str(tmp_path / "a.wav"))
    assert a.dtype == np.float32 and a.shape == (16000,)
    np.testing.assert_array_equal(a, pcm.astype(np.float32) / 32768.0)
    # stereo: the mean of the channels
    st = rng.integers(-20000, 20000, (8000, 2)).astype("<i2")
    _write_wav(tmp_path / "s.wav", st, 16000)
    np.testing.assert_allclose(load_wav(str(tmp_path / "s.wav")), st.astype(np.float32).mean(1) / 32768.0, atol=1e-7)
    # 44.1 kHz and 8 kHz tones come back at 16 kHz with the right length, frequency and amplitude
    for rate, f0 in ((44100, 1000.0), (8000, 440.0), (48000, 3000.0)):
        n = rate                                           # one second
        t = np.arange(n) / float(rate)
        tone = np.round(0.5 * np.sin(2 * np.pi * f0 * t) * 32767).astype("<i2")
        _write_wav(tmp_path / ("t%d.wav" % rate), tone, rate)
        got = load_wav(str(tmp_path / ("t%d.wav" % rate)))
        assert got.dtype == np.float32 and got.shape == (16000,)
        want = 0.5 * np.sin(2 * np.pi * f0 * np.arange(16000) / 16000.0)
        mid = slice(200, -200)                             # away from the filter's edge transients
        assert np.abs(got[mid] - want[mid]).max() < 2e-3, (rate, np.abs(got[mid] - want[mid]).max())
    # content above the new Nyquist frequency is removed, not aliased
    t = np.arange(44100) / 44100.0
    hi = np.round(0.5 * np.sin(2 * np.pi * 12000.0 * t) * 32767).astype("<i2")
    _write_wav(tmp_path / "hi.wav", hi, 44100)
    assert np.abs(load_wav(str(tmp_path / "hi.wav"))[200:-200]).max() < 1e-3
    # 8-bit and 32-bit PCM
    u8 = rng.integers(0, 256, 4000).astype(np.uint8)
    _write_wav(tmp_path / "u8.wav", u8, 16000, width=1)
    np.testing.assert_allclose(load_wav(str(tmp_path / "u8.wav")), (u8.astype(np.float32) - 128.0) / 128.0)
    i32 = rng.integers(-2 ** 30, 2 ** 30, 4000).astype("<i4")
    _write_wav(tmp_path / "i32.wav", i32, 16000, width=4)
    np.testing.assert_allclose(load_wav(str(tmp_path / "i32.wav")), i32 / 2147483648.0, atol=1e-7)
