"""Pins oracle/stream_oracle.py against vectors produced by the reference's own ThresholdDecoder / TriggerDetector
(tests/golden/make_golden_stream.py executed the class definitions of /root/reference/listen.py)."""
import os

import numpy as np
import pytest

from oracle import stream_oracle as so

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stream_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


@pytest.mark.parametrize("name", ["default", "two", "narrow", "flat"])
def test_decoder_table_matches_reference(g, name):
    mn, rng, cd = so.decoder_table(g["dec_%s_mu_stds" % name])
    assert mn == int(g["dec_%s_min_out" % name])
    assert mn + rng == int(g["dec_%s_max_out" % name])
    ref = g["dec_%s_cd" % name]
    assert cd.shape == ref.shape
    np.testing.assert_allclose(cd, ref, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("name", ["default", "two", "narrow", "flat"])
def test_decode_matches_reference(g, name):
    mn, rng, cd = so.decoder_table(g["dec_%s_mu_stds" % name])
    got = so.decode(g["dec_raw"], mn, rng, cd, float(g["dec_%s_center" % name]))
    np.testing.assert_allclose(got, g["dec_%s_decoded" % name], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", ["default", "two", "narrow", "flat"])
def test_decode_float32_input_matches_reference(g, name):
    mn, rng, cd = so.decoder_table(g["dec_%s_mu_stds" % name])
    got = so.decode(g["dec_raw"], mn, rng, cd, float(g["dec_%s_center" % name]), f32_input=True)
    np.testing.assert_allclose(got, g["dec_%s_decoded_f32in" % name], rtol=1e-12, atol=1e-15)
    # and the two paths really differ near 1 (float32 rounding of 1/x - 1), which is why both are recorded
    if name == "default":
        assert np.max(np.abs(g["dec_default_decoded_f32in"] - g["dec_default_decoded"])) > 1e-4


@pytest.mark.parametrize("name", ["default", "two", "narrow"])
def test_encode_matches_reference(g, name):
    mn, rng, cd = so.decoder_table(g["dec_%s_mu_stds" % name])
    c = float(g["dec_%s_center" % name])
    got = [so.encode(float(t), mn, rng, cd, c) for t in g["dec_%s_encode_in" % name]]
    np.testing.assert_allclose(got, g["dec_%s_encoded" % name], rtol=1e-12)


def test_decode_edge_values_pass_through(g):
    mn, rng, cd = so.decoder_table(((6, 4),))
    out = so.decode(np.array([0.0, 1.0]), mn, rng, cd, 0.2)
    assert out[0] == 0.0 and out[1] == 1.0


def test_trigger_sequences_match_reference(g):
    for ci in range(int(g["trig_n_cases"])):
        chunk, sens, level = g["trig%d_cfg" % ci]
        st = so.TriggerState()
        fired, act = [], []
        for idx, sc in zip(g["trig%d_index" % ci], g["trig%d_score" % ci]):
            fired.append(1 if st.update(int(idx), float(sc), int(idx) == 0, float(sens), int(level), int(chunk)) else 0)
            act.append(st.activation)
        np.testing.assert_array_equal(fired, g["trig%d_fired" % ci])
        np.testing.assert_array_equal(act, g["trig%d_activation" % ci])


def test_stream_state_two_frames_per_chunk():
    # listen.py:96-114 with window 1024 / hop 512 / chunk 1024: 1 row after the first chunk, then 2 per chunk, 512 carried
    calls = []

    def fake_featurize(audio):
        n = (len(audio) - 1024) // 512 + 1
        calls.append(len(audio))
        return np.full((n, 3), float(len(calls)))

    st = so.StreamState(5, 3, 1024, 512, fake_featurize)
    m = st.push(np.zeros(1024))
    assert calls == [1024] and len(st.window_audio) == 512 and m[-1, 0] == 1.0 and m[-2, 0] == 0.0
    m = st.push(np.zeros(1024))
    assert calls == [1024, 1536] and len(st.window_audio) == 512
    assert list(m[:, 0]) == [0.0, 0.0, 1.0, 2.0, 2.0]
    m = st.push(np.zeros(300))                      # not enough for a new frame: nothing changes
    assert len(calls) == 2 and len(st.window_audio) == 812
    m = st.push(np.zeros(5000))                     # more new rows than the matrix holds: only the last 5 are kept
    assert m.shape == (5, 3) and np.all(m == 3.0)
