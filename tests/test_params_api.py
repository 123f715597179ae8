"""CPU tests of the host mirror of classifier/params.py (reference :16-121)."""
import json
import os

import pytest


def test_defaults_and_derived():
    from classifier.params import pr
    assert (pr.buffer_t, pr.window_t, pr.hop_t, pr.sample_rate, pr.sample_depth) == (1.0, 0.064, 0.032, 16000, 2)
    assert (pr.n_fft, pr.n_filt, pr.n_mfcc, pr.use_delta) == (1024, 20, 20, False)
    assert pr.threshold_config == ((6, 4),) and pr.threshold_center == 0.2
    assert (pr.window_samples, pr.hop_samples, pr.max_samples, pr.buffer_samples) == (1024, 512, 16000, 15872)
    assert (pr.n_features, pr.feature_size) == (30, 20)


def test_frozen_and_inject_roundtrip(tmp_path):
    from classifier import params as P
    with pytest.raises(AttributeError):
        P.pr.n_fft = 512
    saved = dict(P.pr.__dict__)
    try:
        path = os.path.join(tmp_path, "params.json")
        P.save_params(path)
        d = json.load(open(path))
        assert set(d) == set(saved) and d["threshold_config"] == [[6, 4]]
        d["use_delta"] = True
        d["n_mfcc"] = 13
        json.dump(d, open(path, "w"))
        assert P.inject_params(path) is P.pr
        assert P.pr.feature_size == 26 and P.pr.threshold_config == [[6, 4]]  # JSON lists replace tuples
    finally:
        P.pr.__dict__.clear()
        P.pr.__dict__.update(saved)


def test_inject_never_raises(tmp_path, capsys):
    from classifier import params as P
    saved = dict(P.pr.__dict__)
    assert P.inject_params(os.path.join(tmp_path, "missing.json")) is P.pr
    assert capsys.readouterr().out == ""
    bad = os.path.join(tmp_path, "bad.json")
    open(bad, "w").write("{not json")
    P.inject_params(bad)
    assert "Warning: Failed to load parameters" in capsys.readouterr().out
    assert P.pr.__dict__ == saved
