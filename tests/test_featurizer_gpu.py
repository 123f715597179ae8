"""GPU parity tests of the HIP featurizer (through the C ABI, include/kws.h) against the CPU oracle and the golden
vectors produced by the reference.  Tolerance: north_star allows 1e-3 for fp32 results; features are O(1..40) and the
kernel computes in fp32 against a float64 oracle, so 2e-4 absolute is asserted for speech/noise-like input."""
import os
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 2e-4
NAMES = ["right_1", "left_1", "up_1", "down_1", "right_2", "left_2", "up_2", "down_2"]   # all eight clips of the reference's example/


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def feat(torch):
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    return Featurizer(pr, "mel")


def _oracle():
    from oracle import featurizer_oracle as fo
    return fo


def test_library_is_the_hip_one(torch):
    import kws_amd
    assert kws_amd.device_count() >= 1
    assert "gfx950" in kws_amd.version()


@pytest.mark.parametrize("name", NAMES)
def test_golden_reference_vectors_f32_and_i16(torch, feat, golden, name):
    pcm = golden["pcm_" + name]
    a32 = torch.from_numpy(pcm.astype(np.float32) / 32768.0).cuda()[None]
    got = feat(a32)[0].cpu().numpy()
    assert got.shape == (30, 20)
    np.testing.assert_allclose(got, golden["refcpp_f64_" + name], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got, golden["refpy_mel_" + name], atol=ATOL, rtol=0)
    got16 = feat(torch.from_numpy(pcm.copy()).cuda()[None])[0].cpu().numpy()
    np.testing.assert_array_equal(got16, got)  # int16 -> float is exact, same arithmetic afterwards


def test_random_batch_with_ragged_lengths(torch, feat):
    fo = _oracle()
    rng = np.random.default_rng(1234)
    lens = np.array([16000, 0, 1, 2, 511, 1023, 1024, 1025, 9000, 9001, 15999, 16000, 17000, 20000, 8000, 12345],
                    np.int32)
    stride = 20000
    wav = np.clip(0.1 * rng.standard_normal((len(lens), stride)), -1, 1).astype(np.float32)
    wav = np.round(wav * 32768) / 32768
    got = feat(torch.from_numpy(wav.astype(np.float32)).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
    for b, n in enumerate(lens):
        want = fo.audio_to_feature(wav[b, :n].astype(np.float64))
        np.testing.assert_allclose(got[b], want, atol=ATOL, rtol=0, err_msg="clip %d len %d" % (b, n))
    # all-padding clip: every frame at the log floor, exactly (Python behaviour is canonical)
    assert np.all(got[1][:, 0] == np.float32(np.log(np.finfo(float).eps)))
    assert np.all(got[1][:, 1:] == 0) or np.abs(got[1][:, 1:]).max() < 1e-5


def test_short_and_long_goldens(torch, feat, golden):
    s = golden["syn_short_audio"].astype(np.float32)
    got = feat(torch.from_numpy(s).cuda()[None], torch.tensor([len(s)], dtype=torch.int32).cuda())[0].cpu().numpy()
    np.testing.assert_allclose(got, golden["refpy_mel_syn_short"], atol=ATOL, rtol=0)
    l = golden["syn_long_audio"].astype(np.float32)
    got = feat(torch.from_numpy(l).cuda()[None])[0].cpu().numpy()  # stride 20000 > max_samples: head kept
    np.testing.assert_allclose(got, golden["refpy_mel_syn_long"], atol=ATOL, rtol=0)
    z = feat(torch.zeros((1, 16000), device="cuda"))[0].cpu().numpy()
    np.testing.assert_allclose(z, golden["refpy_mel_silence"], atol=1e-5, rtol=0)


def test_tone_high_dynamic_range(torch, feat, golden):
    """A pure 1 kHz tone at half scale: one band carries the tone, its two neighbours sit 95-110 dB below it (level -22 .. -25 in log
    power), three more 110-130 dB below, the rest at the eps clip.  The fp32 transform (the precision the reference's own numpy >= 2
    float32 path has, SURVEY 7 'dtype of the Python path') resolves all of that: c0 (log energy) within 2e-4 like every other clip, every
    coefficient within 5e-3 (measured 1.8e-3; round 2 accepted 0.5), and -- in the BAND domain, where an error can be priced against the
    band's own level -- every band down to 108 dB below the tone within 2.5e-3 of the float64 reference (measured 1.3e-3), the clipped
    bands exactly at log(eps)."""
    fo = _oracle()
    t = golden["syn_tone_audio"].astype(np.float32)
    got = feat(torch.from_numpy(t).cuda()[None])[0].cpu().numpy().astype(np.float64)
    want = golden["refpy_mel_syn_tone"]
    np.testing.assert_allclose(got[:, 0], want[:, 0], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got, want, atol=5e-3, rtol=0)
    # band domain: the ortho DCT-II is orthogonal, and c0 (overwritten by the log energy) only carries a per-frame constant over the bands,
    # so the band logs are known up to that constant; it is fixed on the band that carries the tone
    N = 20
    n = np.arange(N)
    D = np.cos(np.pi * (n[None, :] + 0.5) * n[:, None] / N) * np.where(n[:, None] == 0, np.sqrt(1.0 / N), np.sqrt(2.0 / N))     # D[k][band]

    def bands(c):
        c = c.copy()
        c[:, 0] = 0.0
        return c @ D
    level = np.log(np.clip(fo.power_spec(t.astype(np.float64), 1024, 512, 1024) @ fo.bank().T, 2.220446049250313e-16, None))    # absolute band logs
    delta = bands(got) - bands(want)
    delta -= delta[np.arange(delta.shape[0]), level.argmax(1)][:, None]
    rel = level - level.max(1, keepdims=True)
    resolved = rel >= -25.0                                # 108 dB below the tone
    assert resolved.sum() == 3 * level.shape[0]            # the tone's band and its two neighbours, in every frame
    assert np.abs(delta[resolved]).max() < 2.5e-3
    assert np.abs(delta).max() < 5e-3


@pytest.mark.parametrize("name", NAMES[:2])
def test_bark_bank_goldens(torch, golden, name):
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    fb = Featurizer(pr, "bark")
    np.testing.assert_allclose(fb.bank(), golden["bark_bank_20x513"], atol=1e-6)
    a = torch.from_numpy(golden["pcm_" + name].astype(np.float32) / 32768.0).cuda()[None]
    np.testing.assert_allclose(fb(a)[0].cpu().numpy(), golden["refpy_bark_" + name], atol=ATOL, rtol=0)
    s = golden["syn_short_audio"].astype(np.float32)
    got = fb(torch.from_numpy(s).cuda()[None], torch.tensor([len(s)], dtype=torch.int32).cuda())[0].cpu().numpy()
    np.testing.assert_allclose(got, golden["refpy_bark_syn_short"], atol=ATOL, rtol=0)


def test_use_delta(torch):
    from classifier.params import ListenerParams
    from kws_amd.featurizer import Featurizer
    fo = _oracle()
    p = ListenerParams(1.0, 0.064, 0.032, 16000, 2, 1024, 20, 20, True, ((6, 4),), 0.2)
    f = Featurizer(p)
    rng = np.random.default_rng(5)
    a = (0.1 * rng.standard_normal((3, 16000))).astype(np.float32)
    got = f(torch.from_numpy(a).cuda()).cpu().numpy()
    assert got.shape == (3, 30, 40)
    for b in range(3):
        np.testing.assert_allclose(got[b], fo.audio_to_feature(a[b].astype(np.float64), use_delta=True), atol=2 * ATOL)


def test_other_filter_counts(torch):
    from classifier.params import ListenerParams
    from kws_amd.featurizer import Featurizer
    fo = _oracle()
    rng = np.random.default_rng(6)
    a = (0.1 * rng.standard_normal((2, 16000))).astype(np.float32)
    for n_filt, n_mfcc, win, hop in [(40, 13, 0.064, 0.032), (26, 13, 0.025, 0.010), (20, 20, 0.064, 0.016)]:
        p = ListenerParams(1.0, win, hop, 16000, 2, 1024, n_filt, n_mfcc, False, ((6, 4),), 0.2)
        got = Featurizer(p)(torch.from_numpy(a).cuda()).cpu().numpy()
        for b in range(2):
            want = fo.audio_to_feature(a[b].astype(np.float64), n_filt=n_filt, n_mfcc=n_mfcc, window_t=win, hop_t=hop)
            np.testing.assert_allclose(got[b], want, atol=ATOL, rtol=0)


def test_vectorize_raw_lengths(torch, feat):
    fo = _oracle()
    rng = np.random.default_rng(8)
    for n in (1023, 1024, 1535, 1536, 5000, 16000, 48000):
        a = (0.1 * rng.standard_normal((2, n))).astype(np.float32)
        got = feat.raw(torch.from_numpy(a).cuda()).cpu().numpy()
        want = fo.mfcc_spec(a[1].astype(np.float64))
        assert got.shape[1:] == want.shape
        if want.size:
            np.testing.assert_allclose(got[1], want, atol=ATOL, rtol=0)


def test_unsupported_and_invalid_params_fail_loudly(torch):
    from classifier.params import ListenerParams
    from kws_amd import KwsError
    from kws_amd.featurizer import Featurizer
    with pytest.raises(KwsError):
        Featurizer(ListenerParams(1.0, 0.064, 0.032, 16000, 2, 1024, 20, 30, False, ((6, 4),), 0.2))  # n_mfcc > n_filt
    with pytest.raises(KwsError):
        Featurizer(ListenerParams(1.0, 0.064, 0.032, 16000, 2, 64, 40, 13, False, ((6, 4),), 0.2))   # repeated grid bins


def test_data_utils_api(torch, golden, tmp_path):
    from common import data_utils as du
    pcm = golden["pcm_right_1"]
    path = os.path.join(tmp_path, "right_1.wav")
    w = wave.open(path, "wb")
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes()); w.close()
    f = du.get_mfcc_feature(path)
    assert f.shape == (30, 20, 1)
    np.testing.assert_allclose(f[..., 0], golden["refpy_mel_right_1"], atol=ATOL, rtol=0)
    a = du.buffer_to_audio(pcm.tobytes())
    np.testing.assert_array_equal(a, pcm.astype(np.float32) / 32768.0)
    np.testing.assert_array_equal(np.frombuffer(du.audio_to_buffer(a), "<i2"), pcm)
    v = du.vectorize_raw(a[:4096])
    assert v.shape == (7, 20)
    with pytest.raises(du.InvalidAudio):
        du.vectorize_raw(np.zeros(0))
    np.testing.assert_allclose(du.audio_to_feature(a[:9000]),
                               _oracle().audio_to_feature(a[:9000].astype(np.float64)), atol=ATOL)


def test_full_size_properties_b4096(torch, feat):
    """BASELINE config size: batch invariance (bit-exact), run-to-run determinism, and the gain law
    x -> 2x  =>  c0 += ln 4 and c1.. unchanged (a constant shift of every log band only moves the DC term)."""
    g = torch.Generator(device="cuda").manual_seed(1234)
    wav = (0.1 * torch.randn((4096, 16000), generator=g, device="cuda")).clamp_(-1, 1 - 2.0 ** -15)
    wav = torch.round(wav * 32768) / 32768
    out = feat(wav)
    out2 = feat(wav)
    assert torch.equal(out, out2)
    idx = [0, 1, 777, 2048, 4095]
    sub = feat(wav[idx].contiguous())
    assert torch.equal(sub, out[idx])
    out_x2 = feat(wav * 2)
    d = (out_x2 - out).cpu().numpy()
    np.testing.assert_allclose(d[..., 0], np.log(4.0), atol=1e-4)
    np.testing.assert_allclose(d[..., 1:], 0.0, atol=1e-4)
    fo = _oracle()
    for b in idx:
        np.testing.assert_allclose(out[b].cpu().numpy(), fo.audio_to_feature(wav[b].double().cpu().numpy()), atol=ATOL)


def test_other_fft_sizes_generic_kernel(torch):
    """n_fft != 1024 runs the generic radix-2 kernel: cropped window (window > n_fft), exact fit, zero padding, deltas,
    bark bank, int16 input and ragged lengths."""
    from classifier.params import ListenerParams
    from kws_amd.featurizer import Featurizer
    fo = _oracle()
    rng = np.random.default_rng(12)
    lens = np.array([16000, 5000, 0, 16000], np.int32)
    a = np.round(np.clip(0.1 * rng.standard_normal((4, 16000)), -1, 1) * 32768) / 32768
    cases = [dict(n_fft=512, window_t=0.032, hop_t=0.016, n_filt=20, n_mfcc=13),       # window == n_fft
             dict(n_fft=2048, window_t=0.064, hop_t=0.032, n_filt=20, n_mfcc=20),      # zero padded to 2048
             dict(n_fft=512, window_t=0.064, hop_t=0.032, n_filt=20, n_mfcc=20),       # window 1024 cropped to 512
             dict(n_fft=256, window_t=0.016, hop_t=0.008, n_filt=13, n_mfcc=13, use_delta=True)]
    for kw in cases:
        p = ListenerParams(1.0, kw["window_t"], kw["hop_t"], 16000, 2, kw["n_fft"], kw["n_filt"], kw["n_mfcc"],
                           kw.get("use_delta", False), ((6, 4),), 0.2)
        for bank in ("mel", "bark") if kw["n_fft"] >= 512 else ("mel",):
            f = Featurizer(p, bank)
            got = f(torch.from_numpy(a.astype(np.float32)).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
            for b in range(4):
                want = fo.audio_to_feature(a[b, :lens[b]], kind=bank, **kw)
                np.testing.assert_allclose(got[b], want, atol=ATOL if not kw.get("use_delta") else 2 * ATOL, rtol=0,
                                           err_msg="%s %s clip %d" % (kw, bank, b))
        got16 = Featurizer(p)(torch.from_numpy((a * 32768).astype(np.int16)).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
        np.testing.assert_allclose(got16, Featurizer(p)(torch.from_numpy(a.astype(np.float32)).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy(), atol=1e-6)


def test_feature_pipeline_matches_direct_calls(torch):
    """kws_amd/pipeline.py: features computed on the side stream, double buffered, equal the direct call bit for bit, and a
    buffer is not rewritten before the step that read it was released."""
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    from kws_amd.pipeline import FeaturePipeline
    f = Featurizer(pr)
    B = 64
    g = torch.Generator(device="cuda").manual_seed(11)
    wavs = [0.1 * torch.randn((B, 16000), device="cuda", generator=g) for _ in range(5)]
    want = [f(w).clone() for w in wavs]
    pipe = FeaturePipeline(f, B, pr.n_features, pr.feature_size)
    with pytest.raises(RuntimeError):
        pipe.take()
    seen = []
    pipe.submit(wavs[0])
    for i in range(5):
        feat = pipe.take()
        if i + 1 < 5:
            pipe.submit(wavs[i + 1])
        # a long-running consumer of `feat` on the main stream: the pipeline may not overwrite the buffer under it
        acc = feat.clone()
        for _ in range(20):
            acc = acc + 0.0 * feat
        seen.append(acc)
        pipe.release()
    torch.cuda.synchronize()
    for got, w in zip(seen, want):
        assert torch.equal(got, w)


@pytest.mark.parametrize("bank", ["mel", "bark"])
def test_shared_mode_kernel_equals_standalone(torch, bank):
    """kws_featurizer_set_cu_share(f, 1) selects the form of the tuned kernel that keeps its twiddles in registers (one block per CU beside
    a train step): same arithmetic, so the features equal the stand-alone form bit for bit -- float32 and PCM16 input, ragged lengths
    (left-padded clips take the masked load path), a batch that does not divide over the grid's waves."""
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    B = 333
    rng = np.random.default_rng(21)
    a = np.clip(0.2 * rng.standard_normal((B, 16000)), -1, 1 - 2.0 ** -15).astype(np.float32)
    lens = rng.integers(3000, 16001, B).astype(np.int32)
    lens[::7] = 16000
    wav, vl = torch.from_numpy(a).cuda(), torch.from_numpy(lens).cuda()
    w16 = torch.from_numpy((a * 32768).astype(np.int16)).cuda()
    alone, shared = Featurizer(pr, bank), Featurizer(pr, bank)
    shared.set_cu_share(1)
    for x in (wav, w16):
        assert torch.equal(alone(x, vl), shared(x, vl))
        assert torch.equal(alone(x), shared(x))
