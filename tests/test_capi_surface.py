"""CPU tests of the C-ABI surface: the library loads without a GPU, exports every symbol include/kws.h declares,
host-only entry points work, and device entry points fail loudly (no CPU fallback)."""
import ctypes
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kws.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kws_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from kws_amd import get_lib
    L = get_lib()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), "libkws_hip.so does not export " + n


def test_geometry_is_the_references():
    from classifier.params import pr
    from kws_amd.featurizer import derive_geometry
    with open(os.path.join(ROOT, "tests", "golden", "params_defaults.json")) as f:
        ref = json.load(f)["derived"]
    assert derive_geometry(pr) == ref
    for k, v in ref.items():
        assert getattr(pr, k) == v


def test_no_cpu_fallback_without_device():
    import kws_amd
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    if kws_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(kws_amd.KwsError) as e:
        Featurizer(pr)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_invalid_params_are_reported_host_side():
    import kws_amd
    from kws_amd import lib as l
    p = l.KwsParams(1.0, 0.0, 0.032, 16000, 2, 1024, 20, 20, 0)
    g = l.KwsGeometry()
    rc = l.get_lib().kws_params_derive(ctypes.byref(p), ctypes.byref(g))
    assert rc == -1 and b"positive" in l.get_lib().kws_last_error()
    with pytest.raises(kws_amd.KwsError):
        l.check(rc)
