"""CPU tests: the C oracle (oracle/kws_oracle.c) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py) and against an independent numpy restatement."""
import json
import os

import numpy as np
import pytest

from oracle import featurizer_oracle as fo

NAMES = ["right_1", "left_1", "up_1", "down_1", "right_2", "left_2", "up_2", "down_2"]   # all eight clips of the reference's example/
HERE = os.path.dirname(os.path.abspath(__file__))


def test_geometry_matches_reference_params():
    # derived values written by the reference's own classifier/params.py (params_defaults.json)
    with open(os.path.join(HERE, "golden", "params_defaults.json")) as f:
        ref = json.load(f)["derived"]
    assert fo.geometry() == ref
    assert ref["n_features"] == 30 and ref["feature_size"] == 20


def test_mel_grid_points():
    assert fo.mel_points() == [0, 3, 7, 12, 18, 25, 33, 42, 52, 64, 79, 95, 115, 137, 163, 193, 229, 270, 317, 373,
                               437, 513]
    bank = fo.bank("mel")
    assert bank.shape == (20, 513) and np.count_nonzero(bank) == 947 - 20  # first tap of every rising edge is 0.0
    np.testing.assert_allclose(bank, fo.numpy_mel_bank(16000, 1024, 20), rtol=1e-14, atol=1e-16)  # linspace ulp


@pytest.mark.parametrize("sr,n_fft,n_filt", [(16000, 512, 20), (16000, 1024, 40), (8000, 512, 13), (16000, 2048, 26),
                                              (22050, 1024, 20)])
def test_mel_grid_other_configs_match_numpy(sr, n_fft, n_filt):
    np.testing.assert_allclose(fo.bank("mel", sr, n_fft, n_filt), fo.numpy_mel_bank(sr, n_fft, n_filt), rtol=1e-14,
                               atol=1e-16)


def test_mel_grid_refuses_duplicate_points():
    with pytest.raises(ValueError):
        fo.mel_points(16000, 64, 40)


def test_bark_bank_matches_reference(golden):
    bank = fo.bank("bark")
    np.testing.assert_allclose(bank, golden["bark_bank_20x513"], rtol=1e-12, atol=1e-15)
    assert np.count_nonzero(bank) == 715


def test_power_spec_matches_reference(golden):
    a = golden["pcm_right_1"].astype(np.float64) / 32768.0
    p = fo.power_spec(a, 1024, 512, 1024)
    np.testing.assert_allclose(p, golden["power_spec_right_1"], rtol=1e-9, atol=1e-18)


@pytest.mark.parametrize("name", NAMES)
def test_mfcc_matches_reference_cpp_and_python(golden, name):
    a = golden["pcm_" + name].astype(np.float64) / 32768.0
    got = fo.audio_to_feature(a)
    assert got.shape == (30, 20)
    np.testing.assert_allclose(got, golden["refcpp_f64_" + name], atol=1e-9, rtol=0)
    np.testing.assert_allclose(got, golden["refpy_mel_" + name], atol=1e-9, rtol=0)
    # the float instantiation of the reference C++ differs only by its float32 output rounding
    np.testing.assert_allclose(got, golden["refcpp_f32_" + name], atol=5e-6, rtol=0)


@pytest.mark.parametrize("name", NAMES)
def test_bfcc_matches_reference_python(golden, name):
    a = golden["pcm_" + name].astype(np.float64) / 32768.0
    got = fo.audio_to_feature(a, kind="bark")
    np.testing.assert_allclose(got, golden["refpy_bark_" + name], atol=1e-9, rtol=0)


def test_short_clip_left_pad_and_log_floor(golden):
    got = fo.audio_to_feature(golden["syn_short_audio"])
    np.testing.assert_allclose(got, golden["refpy_mel_syn_short"], atol=1e-9, rtol=0)
    # 7000 leading zeros -> frames 0..11 are all-zero: c0 = log(eps), c1.. = 0 (Python behaviour is canonical)
    assert np.allclose(got[:12, 0], np.log(np.finfo(float).eps))
    assert np.abs(got[:12, 1:]).max() < 1e-12
    got_b = fo.audio_to_feature(golden["syn_short_audio"], kind="bark")
    np.testing.assert_allclose(got_b, golden["refpy_bark_syn_short"], atol=1e-9, rtol=0)


def test_long_clip_keeps_head(golden):
    got = fo.audio_to_feature(golden["syn_long_audio"])
    np.testing.assert_allclose(got, golden["refpy_mel_syn_long"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(got, golden["refcpp_f64_syn_long"], atol=1e-9, rtol=0)


def test_tone_and_silence(golden):
    np.testing.assert_allclose(fo.audio_to_feature(golden["syn_tone_audio"]), golden["refpy_mel_syn_tone"], atol=1e-8)
    np.testing.assert_allclose(fo.audio_to_feature(np.zeros(16000)), golden["refpy_mel_silence"], atol=1e-12)
    np.testing.assert_allclose(fo.audio_to_feature(np.zeros(0)), golden["refpy_mel_silence"], atol=1e-12)


def test_live_reference_build_agrees_when_present():
    if fo.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference absent on this machine)")
    rng = np.random.default_rng(7)
    a = np.round(np.clip(0.1 * rng.standard_normal(16000), -1, 1) * 32768) / 32768
    np.testing.assert_allclose(fo.audio_to_feature(a), fo.ref_mfcc(a, np.float64), atol=1e-9, rtol=0)


def test_numpy_restatement_agrees_on_random_and_other_geometry():
    rng = np.random.default_rng(3)
    a = 0.3 * rng.standard_normal(16000)
    np.testing.assert_allclose(fo.audio_to_feature(a), fo.numpy_mfcc(a), atol=1e-9)
    kw = dict(window_t=0.032, hop_t=0.016, n_fft=512, n_filt=26, n_mfcc=13)
    np.testing.assert_allclose(fo.mfcc_spec(a, **kw), fo.numpy_mfcc(a, **kw), atol=1e-9)
    kw = dict(window_t=0.025, hop_t=0.010, n_fft=512, n_filt=20, n_mfcc=13)  # window 400 < n_fft: zero padded
    np.testing.assert_allclose(fo.mfcc_spec(a, **kw), fo.numpy_mfcc(a, **kw), atol=1e-9)


def test_deltas_follow_add_deltas():
    # common/data_utils.py:50-58: delta[0] = 0, delta[i] = f[i] - f[i-1], concatenated on the last axis
    rng = np.random.default_rng(5)
    a = 0.1 * rng.standard_normal(16000)
    base = fo.audio_to_feature(a)
    d = fo.audio_to_feature(a, use_delta=True)
    assert d.shape == (30, 40)
    np.testing.assert_array_equal(d[:, :20], base)
    np.testing.assert_array_equal(d[0, 20:], 0)
    np.testing.assert_allclose(d[1:, 20:], base[1:] - base[:-1], atol=0)


def test_batch_entry_matches_single():
    rng = np.random.default_rng(11)
    wav = (0.1 * rng.standard_normal((5, 16000))).astype(np.float32)
    vl = np.array([16000, 9000, 0, 1023, 16000], np.int32)
    out = fo.featurize_batch(wav, vl)
    for b in range(5):
        np.testing.assert_allclose(out[b], fo.audio_to_feature(wav[b, :vl[b]].astype(np.float64)), atol=2e-5)
