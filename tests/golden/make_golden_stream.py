"""Golden vectors for the streaming post-processing, produced by RUNNING the reference.

`/root/reference/listen.py` cannot be imported as a module here (its top level imports pyaudio / tensorflow, which are
not installed), but its two post-processing classes are plain numpy/math code.  This script parses the file, takes the
`ThresholdDecoder` and `TriggerDetector` class definitions out of the syntax tree, executes exactly those definitions and
records their inputs and outputs.  Only the resulting arrays are committed (tests/golden/stream_golden.npz); no reference
source text is stored in this repository.

    python tests/golden/make_golden_stream.py        # needs /root/reference (not available on the GPU box)
"""
import ast
import math
import os
import warnings

import numpy as np

REF = "/root/reference/listen.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stream_golden.npz")


def load_reference_classes():
    tree = ast.parse(open(REF).read(), REF)
    wanted = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("ThresholdDecoder", "TriggerDetector")]
    assert len(wanted) == 2, "reference layout changed"
    ns = {"np": np, "math": math}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)
    return ns["ThresholdDecoder"], ns["TriggerDetector"]


def main():
    ThresholdDecoder, TriggerDetector = load_reference_classes()
    rng = np.random.default_rng(20240607)
    out = {}

    # ---- ThresholdDecoder: tables and decode() on a spread of raw network outputs ----
    configs = {
        "default": (((6, 4),), 0.2),                    # classifier/params.py:102
        "two": (((6, 4), (2, 1.5)), 0.5),
        "narrow": (((0.5, 0.25),), 0.35),
        "flat": (((0, 0),), 0.2),                       # out_range == 0 branch
    }
    raw = np.concatenate([
        rng.uniform(0.0, 1.0, 1500),
        1.0 / (1.0 + np.exp(-rng.normal(3.0, 4.0, 1500))),                      # logit-normal, like a softmax maximum
        rng.uniform(0.0, 1.0, 500).astype(np.float32).astype(np.float64),       # exactly representable float32 values
        np.array([0.0, 1.0, 0.5, 1e-12, 1.0 - 1e-12, 1e-300, np.float64(np.float32(0.99999994)), 0.2, 0.8]),
    ])
    out["dec_raw"] = raw
    for name, (mu_stds, center) in configs.items():
        d = ThresholdDecoder(mu_stds, center)
        out["dec_%s_mu_stds" % name] = np.asarray(mu_stds, dtype=np.float64)
        out["dec_%s_center" % name] = np.float64(center)
        out["dec_%s_min_out" % name] = np.int64(d.min_out)
        out["dec_%s_max_out" % name] = np.int64(d.max_out)
        out["dec_%s_cd" % name] = np.asarray(d.cd, dtype=np.float64)
        out["dec_%s_decoded" % name] = np.array([float(d.decode(float(x))) for x in raw], dtype=np.float64)
        # the live loop (listen.py:361-367) passes the float32 array np.max(output, axis=-1) of shape (1,): numpy then
        # evaluates `1 / x - 1` in float32 before math.log widens it -- recorded separately from the Python-float path
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            out["dec_%s_decoded_f32in" % name] = np.array(
                [float(np.asarray(d.decode(np.array([x], dtype=np.float32))).reshape(-1)[0]) for x in raw], dtype=np.float64)
        if name != "flat":
            thr = np.linspace(0.02, 0.98, 25)
            out["dec_%s_encode_in" % name] = thr
            out["dec_%s_encoded" % name] = np.array([float(d.encode(float(t))) for t in thr], dtype=np.float64)

    # ---- TriggerDetector: prediction sequences with runs of the same class above / below the sensitivity ----
    class_names = ["background", "up", "down", "left", "right"]
    cases = [(1024, 0.5, 3), (2048, 0.5, 3), (512, 0.35, 1), (1024, 0.8, 5), (4096, 0.5, 0)]
    for ci, (chunk, sens, level) in enumerate(cases):
        n = 600
        idx = np.empty(n, dtype=np.int64)
        i = 0
        while i < n:                                     # runs of 1..12 identical predictions
            run = int(rng.integers(1, 13))
            idx[i:i + run] = int(rng.integers(0, len(class_names)))
            i += run
        score = np.clip(rng.normal(0.62, 0.25, n), 0.0, 1.0)
        score[rng.integers(0, n, 20)] = sens             # equality with the sensitivity must NOT activate (strict >)
        det = TriggerDetector(chunk, class_names, sens, level)
        fired = np.zeros(n, dtype=np.int64)
        act = np.zeros(n, dtype=np.int64)
        for t in range(n):
            fired[t] = 1 if det.update(int(idx[t]), float(score[t])) else 0
            act[t] = det.activation
        out["trig%d_cfg" % ci] = np.array([chunk, sens, level], dtype=np.float64)
        out["trig%d_index" % ci] = idx
        out["trig%d_score" % ci] = score
        out["trig%d_fired" % ci] = fired
        out["trig%d_activation" % ci] = act
    out["trig_n_cases"] = np.int64(len(cases))
    out["trig_class_names"] = np.array(class_names)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
