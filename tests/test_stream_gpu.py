"""GPU parity of the streaming post-processing (csrc/kws_stream.hip through kws_amd/stream.py) against
  * the vectors produced by the reference's own ThresholdDecoder / TriggerDetector (tests/golden/stream_golden.npz), and
  * oracle/stream_oracle.py on seeded random streams (the many-stream path the reference does not have).
Integer results (class index, activation counter, fired flag) must be identical; float64 scores agree to 1e-12 relative
(libm vs device log / exp differ in the last bit at most, and the table index is computed in double on both sides)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_golden.npz")
NAMES = ["default", "two", "narrow", "flat"]


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def make_decoder(g, name):
    from kws_amd.stream import ThresholdDecoder
    return ThresholdDecoder([tuple(r) for r in g["dec_%s_mu_stds" % name]], float(g["dec_%s_center" % name]))


@pytest.mark.parametrize("name", NAMES)
def test_decoder_table_matches_reference(torch, g, name):
    d = make_decoder(g, name)
    assert d.min_out == int(g["dec_%s_min_out" % name]) and d.max_out == int(g["dec_%s_max_out" % name])
    assert d.out_range == d.max_out - d.min_out
    ref = g["dec_%s_cd" % name]
    assert d.cd.shape == ref.shape
    np.testing.assert_allclose(d.cd, ref, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("name", NAMES)
def test_decode_python_float_path(torch, g, name):
    d = make_decoder(g, name)
    got = d.decode(g["dec_raw"])                                 # float64 array -> scalar semantics of listen.py:496-508
    np.testing.assert_allclose(got, g["dec_%s_decoded" % name], rtol=1e-12, atol=1e-15)
    assert d.decode(0.0) == 0.0 and d.decode(1.0) == 1.0        # passed through untouched
    assert isinstance(d.decode(0.5), float)


@pytest.mark.parametrize("name", NAMES)
def test_decode_float32_network_output_path(torch, g, name):
    d = make_decoder(g, name)
    got = d.decode(g["dec_raw"].astype(np.float32))              # what np.max(output, axis=-1) hands to decode
    np.testing.assert_allclose(got, g["dec_%s_decoded_f32in" % name], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", ["default", "two", "narrow"])
def test_encode_matches_reference(torch, g, name):
    d = make_decoder(g, name)
    got = [d.encode(float(t)) for t in g["dec_%s_encode_in" % name]]
    np.testing.assert_allclose(got, g["dec_%s_encoded" % name], rtol=1e-12)


def test_decoder_rejects_bad_config(torch):
    from kws_amd.lib import KwsError
    from kws_amd.stream import ThresholdDecoder
    with pytest.raises(KwsError):
        ThresholdDecoder(((6, 4),), 0.2, resolution=0)
    with pytest.raises(KwsError):
        ThresholdDecoder(((1e9, 4),), 0.2)


def test_trigger_detector_mirror_replays_reference_sequences(torch, g):
    from kws_amd.stream import TriggerDetector
    names = [str(n) for n in g["trig_class_names"]]
    for ci in range(int(g["trig_n_cases"])):
        chunk, sens, level = g["trig%d_cfg" % ci]
        det = TriggerDetector(int(chunk), names, float(sens), int(level))
        assert det.activation == 0 and det.record_index is None
        n = 200 if ci else 600                                   # the first case in full, the others a prefix (one launch per step)
        fired, act = [], []
        for idx, sc in zip(g["trig%d_index" % ci][:n], g["trig%d_score" % ci][:n]):
            fired.append(1 if det.update(int(idx), float(sc)) else 0)
            act.append(det.activation)
        np.testing.assert_array_equal(fired, g["trig%d_fired" % ci][:n])
        np.testing.assert_array_equal(act, g["trig%d_activation" % ci][:n])


def test_trigger_update_many_streams_matches_oracle(torch):
    import ctypes
    from kws_amd import lib as L
    from oracle import stream_oracle as so
    lib = L.get_lib()
    S, T, C = 777, 120, 6                                       # S not a multiple of the block size
    rng = np.random.default_rng(5)
    idx = rng.integers(0, C, (T, S)).astype(np.int32)
    hold = rng.random((T, S)) < 0.8                             # mostly repeat the previous class so that streaks build up
    for t in range(1, T):
        idx[t] = np.where(hold[t], idx[t - 1], idx[t])
    score = np.clip(rng.normal(0.6, 0.25, (T, S)), 0, 1)
    state = torch.zeros((S, 2), dtype=torch.int32, device="cuda")
    state[:, 1] = -1
    fired = torch.zeros(S, dtype=torch.int32, device="cuda")
    oracle = [so.TriggerState() for _ in range(S)]
    for t in range(T):
        ti = torch.from_numpy(idx[t]).cuda()
        ts = torch.from_numpy(score[t]).cuda()
        L.check(lib.kws_trigger_update(ti.data_ptr(), ts.data_ptr(), S, 0, 0.5, 3, 1024, state.data_ptr(), fired.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
        want = [1 if o.update(int(idx[t, s]), float(score[t, s]), idx[t, s] == 0, 0.5, 3, 1024) else 0 for s, o in enumerate(oracle)]
        np.testing.assert_array_equal(fired.cpu().numpy(), want)
        np.testing.assert_array_equal(state[:, 0].cpu().numpy(), [o.activation for o in oracle])
    assert sum(o.activation < 0 for o in oracle) > 0            # some streams are inside their refractory period


@pytest.mark.parametrize("n_rows", [1, 2, 7, 30, 45])
def test_push_rows_is_the_sliding_concatenate(torch, n_rows):
    from kws_amd import lib as L
    lib = L.get_lib()
    rng = np.random.default_rng(n_rows)
    S, F, D = 9, 30, 20
    feat = rng.standard_normal((S, F, D)).astype(np.float32)
    rows = rng.standard_normal((S, n_rows, D)).astype(np.float32)
    tf, tr = torch.from_numpy(feat).cuda(), torch.from_numpy(rows).cuda()
    L.check(lib.kws_stream_push_rows(tf.data_ptr(), tr.data_ptr(), S, F, D, n_rows, torch.cuda.current_stream().cuda_stream))
    new = rows[:, -F:] if n_rows > F else rows                   # listen.py:107-109
    want = np.concatenate((feat[:, new.shape[1]:], new), axis=1)
    np.testing.assert_array_equal(tf.cpu().numpy(), want)


def test_push_rows_large_matrix_multi_pass(torch):
    from kws_amd import lib as L
    lib = L.get_lib()
    rng = np.random.default_rng(0)
    S, F, D, n_rows = 3, 101, 40, 5                              # F*D > 2048: several passes of the in-place shift
    feat = rng.standard_normal((S, F, D)).astype(np.float32)
    rows = rng.standard_normal((S, n_rows, D)).astype(np.float32)
    tf, tr = torch.from_numpy(feat).cuda(), torch.from_numpy(rows).cuda()
    L.check(lib.kws_stream_push_rows(tf.data_ptr(), tr.data_ptr(), S, F, D, n_rows, torch.cuda.current_stream().cuda_stream))
    np.testing.assert_array_equal(tf.cpu().numpy(), np.concatenate((feat[:, n_rows:], rows), axis=1))


def _stream_batch(torch, S, chunk, seed=0):
    from classifier.params import pr
    from kws_amd.model import DeviceModel, ModelSpec
    from kws_amd.stream import StreamBatch
    from oracle import model_oracle as mo
    C = 5
    om = mo.Model("simple_cnn", C).init_weights(seed)
    dm = DeviceModel(ModelSpec("simple_cnn", C, pr.n_features, pr.n_mfcc))
    dm.set_weights(om.get_weights())
    names = ["background", "up", "down", "left", "right"]
    return pr, om, StreamBatch(pr, dm, S, chunk_size=chunk, class_names=names, sensitivity=0.5, trigger_level=3)


@pytest.mark.parametrize("chunk", [1024, 800])
def test_stream_batch_matches_oracle_loop(torch, chunk):
    """update_vectors + postprocess for S streams against the oracle's per-stream Python loop (listen.py:96-114, 350-375)."""
    from oracle import featurizer_oracle as fo
    from oracle import stream_oracle as so
    S, T = 5, 40
    pr, om, sb = _stream_batch(torch, S, chunk)
    rng = np.random.default_rng(chunk)
    pcm = np.clip(rng.normal(0, 3000, (T, S, chunk)), -32768, 32767).astype(np.int16)
    pcm[:6, 1] = 0                                                # a stream that starts in silence: log floor on all-zero frames
    dec = so.decoder_table(pr.threshold_config)
    states = [so.StreamState(pr.n_features, pr.n_mfcc, pr.window_samples, pr.hop_samples, fo.mfcc_spec) for _ in range(S)]
    trig = [so.TriggerState() for _ in range(S)]
    for t in range(T):
        chunks = [pcm[t, s].tobytes() for s in range(S)] if t % 2 else pcm[t]      # both input forms
        index, score, fired = sb.push(chunks)
        feats = sb.mfccs.cpu().numpy()
        probs = sb.probs.cpu().numpy()
        for s in range(S):
            want_feat = states[s].push(pcm[t, s].astype(np.float64) / 32768.0)     # buffer_to_audio, data_utils.py:19-21
            np.testing.assert_allclose(feats[s], want_feat, rtol=0, atol=3e-4)
            # post-processing is checked on the probabilities the device produced, so its parity is exact
            wi, ws, wf = so.postprocess(probs[s], 0, dec, pr.threshold_center, trig[s], 0.5, 3, chunk)
            assert int(index[s]) == wi and int(fired[s]) == int(wf)
            assert abs(float(score[s]) - ws) <= 1e-12 * max(1.0, abs(ws))
            assert int(sb.state[s, 0]) == trig[s].activation
        # and the probabilities themselves follow the oracle model on the oracle features (float32 forward pass)
        want_probs = om.predict(np.stack([st.mfccs for st in states])[..., None])
        np.testing.assert_allclose(probs, want_probs, rtol=0, atol=2e-4)


def test_stream_batch_fires_on_a_confident_streak(torch):
    """Force a confident non-background class for several chunks: the detector fires once, then rests (refractory)."""
    from kws_amd import lib as L
    pr, om, sb = _stream_batch(torch, 3, 1024)
    lib = L.get_lib()
    probs = torch.tensor([[0.01, 0.97, 0.01, 0.005, 0.005], [0.9, 0.05, 0.03, 0.01, 0.01], [0.005, 0.005, 0.005, 0.98, 0.005]],
                         dtype=torch.float32, device="cuda")
    fired_at = []
    for t in range(12):
        L.check(lib.kws_stream_postprocess(sb.decoder.handle, probs.data_ptr(), 3, 5, 0, 0.5, 3, 1024, sb.state.data_ptr(),
                                           sb.index.data_ptr(), sb.score.data_ptr(), sb.fired.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        fired_at.append(sb.fired.cpu().numpy().copy())
    fired_at = np.array(fired_at)
    # first prediction only records the class; activations 1..4 follow; the 4th exceeds trigger_level = 3
    assert list(np.nonzero(fired_at[:, 0])[0]) == [4] and list(np.nonzero(fired_at[:, 2])[0]) == [4]
    assert fired_at[:, 1].sum() == 0                              # background never fires
    assert int(sb.state[0, 0]) == -16 + 7                         # -(8*2048)//1024, then +1 per later chunk
    assert sb.index.cpu().tolist() == [1, 0, 3]
    assert float(sb.score[1]) == pytest.approx(0.9)              # background score is not decoded (listen.py:366)


def test_stream_batch_rejects_deltas_and_bad_chunks(torch):
    from classifier.params import ListenerParams, pr
    from kws_amd.stream import StreamBatch
    pr2, om, sb = _stream_batch(torch, 2, 1024)
    with pytest.raises(ValueError):
        sb.push(np.zeros((2, 2048), np.int16))                   # longer than chunk_size
    with pytest.raises(ValueError):
        sb.push(np.zeros((2, 1024), np.float32))                 # not PCM
    d = dict((k, getattr(pr, k)) for k in ("buffer_t", "window_t", "hop_t", "sample_rate", "sample_depth", "n_fft", "n_filt", "n_mfcc",
                                            "threshold_config", "threshold_center"))
    with pytest.raises(ValueError):
        StreamBatch(ListenerParams(use_delta=True, **d), sb.model, 2)


def test_listener_run_wav_mirrors_listen_py(torch, tmp_path):
    """listen.py's Listener API on the device path: run_wav over a synthetic PCM file, chunk by chunk."""
    import wave
    from classifier.model import get_model
    from kws_amd.init import init_weights
    from listen import Listener
    classes = tmp_path / "classes.txt"
    classes.write_text("background\nup\ndown\nleft\nright\n")
    rng = np.random.default_rng(3)
    pcm = np.clip(rng.normal(0, 4000, 16000 + 700), -32768, 32767).astype(np.int16)
    wav_path = str(tmp_path / "in.wav")
    with wave.open(wav_path, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    m = get_model("simple_cnn", 5)
    m.set_weights(init_weights(m.spec, seed=4))
    lis = Listener(model=m, classes_path=str(classes), input_wav=wav_path, chunk_size=1024)
    assert Listener.get_defaults("trigger_level") == 3 and "Unrecognized" in Listener.get_defaults("nope")
    res = lis.run_wav(quiet=True)
    assert len(res) == -(-len(pcm) // 1024)                      # the short last chunk is still processed (listen.py:403-428)
    # same stream through a fresh StreamBatch: identical decisions
    sb = lis.batch(1)
    for t, (index, score, fired) in enumerate(res):
        i2, s2, f2 = sb.push([pcm[t * 1024:(t + 1) * 1024].tobytes()])
        assert (int(i2[0]), float(s2[0]), bool(f2[0])) == (index, score, fired)
    # update_vectors keeps the reference's return shape, and predict() takes what it returns
    lis2 = Listener(model=m, classes_path=str(classes), chunk_size=1024)
    feats = lis2.update_vectors(pcm[:1024].tobytes())
    assert feats.shape == (30, 20, 1) and np.all(feats[:-1] == 0) and np.any(feats[-1] != 0)
    out = lis2.predict(np.expand_dims(feats, 0))
    assert out.shape == (1, 5) and abs(float(out.sum()) - 1.0) < 1e-5
    with pytest.raises(RuntimeError):
        lis2.run_microphone()
