#!/usr/bin/env python3
"""bench.py -- the keyword-spotting train step on N MI355X GPUs (one process per GPU, RCCL for the gradient exchange).

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM:
    featurize (B x 16000 f32 waveforms -> B x 30 x 20 MFCC)  ->  simple_cnn forward (batch-stat BN, dropout)
    -> loss -> backward -> [two-bucket sum-all-reduce of the flat gradient buffer when N > 1, early bucket overlapped with the
    rest of the backward pass, through the C ABI: kws_allreduce_grads] -> Keras-form Adam.
Workload at N = 1: BASELINE.json configs[1] (simple_cnn, 36 logits = background + 35 Speech Commands v2 words,
batch 4096, HIP featurizer + fwd/bwd + Adam).  Weak scaling: every rank keeps B = 4096.

Output: ONE JSON line on rank 0 (contract in the task statement) with
  roofline      the step's dominant kernel, timed live with HIP events on its launch stream (kws_prof_*)
  cpu_baseline  the same step on the host cores: torch-CPU operators (stand-in for TF-Keras CPU, which this image lacks) behind
                the C featurizer oracle, plus the rows SURVEY.md 8(d) lists (reference-plumbing numpy featurizer, bs 512 / 5 classes)
  extra         the other SURVEY.md 8(d) workloads measured in the same run (N = 1 only): featurize + simple_cnn inference forward
                (the north star's HBM-roofline target), simple_gru train step, simple_cnn_lite fp16 hipGraph forward, the train
                step with exact-fp32 MFMA products, and the dense head's matrix-core figures.
"""
import argparse
import json
import os
import sys
import time

# HIP deals streams to hardware queues (default 4 per device).  The step runs on three streams (caller's, the model's side
# stream, the feature pipeline's); with an RCCL communicator alive in the process the default dealt them so that the step's
# fork/join branches serialised (1.2 ms/step instead of 0.72, measured).  Two queues give the same step
# time with and without a communicator.  Must be in the environment before the first HIP call; a caller's own setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tf-keras-speech-commands_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

METRIC = "1 s@16 kHz clips/sec train-step, simple_cnn bs4096, 1/2/4/8 MI355X; eval top-1"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak
N_CLASSES = 36
FWD_BYTES_PER_CLIP = 64144.0   # 64 000 B waveform in + 36 x 4 B probabilities out (features stay on chip)
FEAT_BYTES_PER_CLIP = 66400.0  # featurizer alone, 64 000 in + 2 400 out
PMC_RECORD = os.path.join(ROOT, "profiles", "r03_pmc.json")     # tools/r03_records.py: counters + pipe floors per workload and kernel
# the source files a workload's kernels live in: a record measured on other sources of THESE files is stale and not quoted
PMC_SOURCES = {"feat": ("kws_featurize.hip", "kws_featurize_v3.h"), "feat_shared": ("kws_featurize.hip", "kws_featurize_v3.h"),
               "gru": ("kws_rnn.hip", "kws_gru.h"), "lstm": ("kws_rnn.hip", "kws_lstm.h"),
               "lite16": ("kws_lite.h", "kws_lite_f16.h", "kws_featurize.hip", "kws_featurize_v3.h"),
               "infer": ("kws_model.hip", "kws_conv.h", "kws_layers.h", "kws_layer1.h", "kws_infer_fused.h", "kws_featurize.hip", "kws_featurize_v3.h"),
               "step": ("kws_model.hip", "kws_conv.h", "kws_conv_group.h", "kws_dense_head.h", "kws_layers.h", "kws_layer1.h", "kws_layer1_fast.h",
                        "kws_layer1_moments.h", "kws_device.h")}
_pmc_cache = {}


def pmc_record(workload, kernel, build_id):
    """-> (derived figures + raw counters of `kernel` in `workload` from profiles/r03_pmc.json, note).  None when there is no record or
    it was measured on different sources of the workload's files than the loaded library was built from."""
    if "rec" not in _pmc_cache:
        try:
            with open(PMC_RECORD) as f:
                _pmc_cache["rec"] = json.load(f)
        except (OSError, ValueError):
            _pmc_cache["rec"] = {}
    w = _pmc_cache["rec"].get(workload)
    if not w or kernel not in w.get("kernels", {}):
        return None, "no PMC record for this kernel"
    sha = w.get("source_sha1", {})
    stale = {fn: (sha.get(fn), build_id.get(fn)) for fn in PMC_SOURCES.get(workload, ()) if fn in build_id and sha.get(fn) != build_id.get(fn)}
    if stale:
        return None, "PMC record is stale (measured on / library built from: %s)" % stale
    return w["kernels"][kernel], "profiles/%s [%s]" % (os.path.basename(PMC_RECORD), workload)


def floors_of(rec, launch_ms):
    """the pipe floors of a kernel next to its HBM fraction ('report both'): time the counted work needs on each pipe at its
    measured peak rate (tools/r03_records.py states the formulas), the largest = compute_floor_ms, and its share of the launch"""
    if not rec:
        return {}
    d = rec.get("derived", {})
    fl = {k: d[k] for k in ("valu_floor_ms", "lds_floor_ms", "mfma_floor_ms", "hbm_floor_ms") if k in d}
    comp = {k: v for k, v in fl.items() if k != "hbm_floor_ms"}
    out = dict(fl)
    if comp:
        top = max(comp, key=comp.get)
        out["compute_floor_ms"] = comp[top]
        out["compute_floor_pipe"] = top[:-len("_floor_ms")]
        if launch_ms:
            out["compute_frac"] = round(comp[top] / launch_ms, 4)
    if "mfma_busy" in d:
        out["mfma_busy_pmc"] = d["mfma_busy"]
    return out


def synthetic_batch(B, rank, n_classes):
    """N(0, 0.1) noise on the int16 grid, 25 % of the clips with a leading run of zeros."""
    rng = np.random.default_rng(1234 + rank)
    wav = np.clip(0.1 * rng.standard_normal((B, 16000), dtype=np.float32), -1, 1 - 2.0 ** -15)
    wav = (np.round(wav * 32768) / 32768).astype(np.float32)
    lead = rng.integers(512, 8001, B)
    sel = rng.uniform(size=B) < 0.25
    for b in np.nonzero(sel)[0]:
        wav[b, :lead[b]] = 0.0
    labels = rng.integers(0, n_classes, B).astype(np.int32)
    return wav, labels


def kernel_models(B, C):
    """Algorithmic work per launch of every kernel of the step (DESIGN.md section 'Kernels'):
    name -> (bound, amount) with bytes for HBM-bound kernels and flops for MFMA-bound ones."""
    f = 4
    z = [30 * 20 * 16, 15 * 10 * 32, 4 * 3 * 64, 4 * 3 * 128]          # pre-BN conv outputs per clip
    a = [15 * 10 * 16, 7 * 5 * 32, 4 * 3 * 64, 256]                    # activations per clip
    m = {}
    m["featurize_fft1024_f32"] = ("hbm", B * FEAT_BYTES_PER_CLIP)
    m["l1_moments_kernel"] = ("hbm", B * 600.0 * f)
    m["l1m_bwd_onepass_kernel"] = ("hbm", B * (600 + a[0]) * f)
    # layer 1 is recomputed from the feature map (kws_layer1.h): bytes = features (+ a1 / da1), never a z1-sized tensor
    for k, v in (("stats", 600.0), ("act_pool", 600 + a[0]), ("bwd_reduce", 600 + a[0]), ("bwd_wgrad", 600 + a[0])):
        m["l1_%s_kernel" % k] = m["l1m_%s_kernel" % k] = ("hbm", B * v * f)
    convs = {"16,32": (150, 9 * 16 * 32), "32,64": (12, 9 * 32 * 64), "64,128": (12, 9 * 64 * 128), "128,128": (1, 256 * 128)}
    for k, (pix, kn) in convs.items():
        m["conv_gemm_fwd<%s>" % k] = ("mfma", 2.0 * B * pix * kn)
        m["conv_wgrad<%s>" % k] = ("mfma", 2.0 * B * pix * kn)
    # conv3 / conv4 / dense run as three-way bf16 splits on the bf16 matrix cores (kws_conv.h: conv_bf16_kernel); the
    # algorithmic work stays the layer's fp32 MACs (the six partial products per MAC are an implementation detail)
    for k, fl in (("conv_bf16_fwd<32,64>", 2.0 * B * 12 * 9 * 32 * 64), ("conv_bf16_fwd<64,128>", 2.0 * B * 12 * 9 * 64 * 128),
                  ("conv_bf16_fwd<128,128>", 2.0 * B * 256 * 128), ("conv_bf16_dgrad<128,64>", 2.0 * B * 12 * 9 * 64 * 128),
                  ("conv_bf16_dgrad<128,128>", 2.0 * B * 256 * 128)):
        m[k] = ("mfma", fl)
    # dgrad kernels are named <reduced channels, produced channels>; algorithmic = the useful MACs of the layer
    m["conv_dgrad<32,16>"] = ("mfma", 2.0 * B * 150 * 9 * 16 * 32)
    m["conv_dgrad<64,32>"] = ("mfma", 2.0 * B * 12 * 9 * 32 * 64)         # one launch over the four stride-2 parity classes
    for k in ("conv_fwd_clip<16,32>", "conv_wgrad_clip<16,32>", "conv_dgrad_clip<32,16>", "conv_fwd_clip_bf16<16,32>",
              "conv_wgrad_clip_bf16<16,32>", "conv_dgrad_clip_bf16<32,16>"):
        m[k] = ("mfma", 2.0 * B * 150 * 9 * 16 * 32)
    m["conv_wgrad_bf16<32,64>"] = ("mfma", 2.0 * B * 12 * 9 * 32 * 64)
    m["conv_wgrad_bf16<64,128>"] = ("mfma", 2.0 * B * 12 * 9 * 64 * 128)
    m["conv_dgrad<128,64>"] = ("mfma", 2.0 * B * 12 * 9 * 64 * 128)
    m["conv_dgrad<128,128>"] = ("mfma", 2.0 * B * 256 * 128)
    # Dense forward + head forward / backward + Dense data gradient in one kernel (kws_dense_head.h)
    m["dense_head_fused_kernel"] = ("mfma", 2.0 * B * (256 * 128 + 3 * 128 * C + 128 * 256))
    m["head_fwd_bwd_kernel"] = ("mfma", 3.0 * 2.0 * B * 128 * C)           # logits, dW2 and dx in one kernel (the fused head of the train step)
    m["head_fwd_kernel"] = ("mfma", 2.0 * B * 128 * C)
    m["head_bwd_kernel"] = ("mfma", 2.0 * 2.0 * B * 128 * C)              # dW2 and dx
    for l in range(1, 4):
        L = ".L%d" % (l + 1)
        m["channel_stats_kernel" + L] = ("hbm", B * z[l] * f)
        m["bn_act_pool_kernel" + L] = ("hbm", B * (z[l] + a[l]) * f)
        m["bn_bwd_reduce_kernel" + L] = ("hbm", B * (2 * z[l] + a[l]) * f)
        m["bn_bwd_reduce_pool_kernel" + L] = ("hbm", B * (2 * z[l] + a[l]) * f)
        m["bn_bwd_apply_kernel" + L] = ("hbm", B * 3 * z[l] * f)
    m["adam_kernel"] = ("hbm", 134932 * 7.0 * f)
    return m


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (host cores of the GPU box; bounded samples).  oracle/ is used here as the thing TIMED BESIDE the GPU path,
# never inside it.
# ---------------------------------------------------------------------------------------------------------------------
def log(msg):
    """progress on stderr (the JSON line on stdout stays the only stdout output)"""
    sys.stderr.write("[bench %s] %s\n" % (time.strftime("%H:%M:%S"), msg))
    sys.stderr.flush()


def usable_cores():
    """CPUs this process may actually run on: the affinity mask capped by the cgroup CPU quota (a GPU box hands each GPU slot a
    share of the host; os.cpu_count() reports the whole machine and oversubscribes thread pools badly)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = os.environ.get("KWS_BENCH_CPU_THREADS")
    if cap:
        n = max(1, int(cap))
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s):
    import torch
    from oracle import featurizer_oracle as fo
    from oracle import model_oracle as mo
    from oracle import torch_ref as tr
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    log("cpu_baseline: %d usable cores of %d (%s)" % (cores, os.cpu_count() or 1, cpu_model()))
    rows = {}

    def time_steps(fn, max_steps, budget):
        fn(0)                                                    # warm-up (thread pools, allocator)
        t0 = time.time()
        n = 0
        while n < max_steps and (n == 0 or time.time() - t0 < budget):
            fn(n + 1)
            n += 1
        return n, time.time() - t0

    def torch_train_row(model_type, B, C, with_featurizer):
        wav, labels = synthetic_batch(B, 0, C)
        base = mo.Model(model_type, C).init_weights(0)
        flags = [t for _, _, t in base.weight_list()]
        tm = tr.TorchModel(model_type, base.get_weights(), flags, dtype=torch.float32)
        opt = tr.KerasAdam(1e-3)
        feat0 = fo.featurize_batch(wav).astype(np.float32)
        width = 256 if "cnn" in model_type else 20
        rate = 0.5 if "cnn" in model_type else 0.2
        rng = np.random.default_rng(7)

        def step(i):
            feat = fo.featurize_batch(wav).astype(np.float32) if with_featurizer else feat0
            mask = ((rng.uniform(size=(B, width)) >= rate) / (1 - rate)).astype(np.float32)
            tm.train_step(opt, feat, labels, None, mask)

        n, dt = time_steps(step, 6, budget_s / 4.0)
        log("cpu_baseline: %s B=%d C=%d featurizer=%s: %d steps in %.1f s" % (model_type, B, C, with_featurizer, n, dt))
        return {"value": round(B * n / dt, 1), "unit": "clips/s", "kind": "port", "threads": cores,
                "what": "%s train step (torch-CPU conv2d / batch_norm / max_pool2d / autograd, Keras-form Adam; oracle/torch_ref.py)%s, "
                        "batch %d, %d classes, float32, %d steps" % (model_type, " behind the C featurizer oracle (OpenMP over clips)"
                                                                      if with_featurizer else " on precomputed features (what the reference's fit consumes)",
                                                                      B, C, n)}

    # C3: the identical train step with torch-CPU operators, all cores -- the stand-in for "TF-Keras CPU"
    rows["torch_cpu_step_bs4096_c36_with_featurizer"] = torch_train_row("simple_cnn", 4096, N_CLASSES, True)
    rows["torch_cpu_step_bs4096_c36"] = torch_train_row("simple_cnn", 4096, N_CLASSES, False)
    rows["torch_cpu_step_bs512_c5"] = torch_train_row("simple_cnn", 512, 5, False)          # BASELINE configs[0]
    rows["torch_cpu_gru_step_bs2048_c36"] = torch_train_row("simple_gru", 2048, N_CLASSES, False)
    # C2: single-thread per-clip numpy featurizer with the reference's plumbing (classifier/data.py:39-44 ->
    # common/data_utils.py:61-86: keep-head / left-pad, list-comprehension framing, rfft, dense bank product rebuilt per call, scipy DCT)
    wav, _ = synthetic_batch(256, 0, N_CLASSES)
    torch.set_num_threads(1)

    def numpy_clip(i):
        a = wav[i % 256]
        fo.numpy_mfcc(a[:16000])

    n, dt = time_steps(numpy_clip, 2000, budget_s / 6.0)
    torch.set_num_threads(cores)
    rows["numpy_featurizer_per_clip_1thread"] = {"value": round(n / dt, 1), "unit": "clips/s", "kind": "port", "threads": 1,
                                                 "what": "per-clip Python loop over oracle.featurizer_oracle.numpy_mfcc (the reference's plumbing: framing by list "
                                                         "comprehension, np.fft.rfft, dense (30,513)x(513,20) bank product with the bank rebuilt per call, scipy DCT), %d clips" % n}
    # C1: the C featurizer oracle, OpenMP over clips, all cores
    wav4, _ = synthetic_batch(4096, 0, N_CLASSES)
    n, dt = time_steps(lambda i: fo.featurize_batch(wav4), 4, budget_s / 6.0)
    rows["c_featurizer_oracle_allcores"] = {"value": round(4096 * n / dt, 1), "unit": "clips/s", "kind": "port", "threads": cores,
                                            "what": "oracle/kws_oracle.c (float64, OpenMP over clips), %d batches of 4096" % n}
    head = rows["torch_cpu_step_bs4096_c36_with_featurizer"]
    return {"value": head["value"], "unit": "clips/s", "cores": cores, "kind": "port", "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": "headline row = torch_cpu_step_bs4096_c36_with_featurizer: " + head["what"] + " (same generator as the GPU run; "
                      "TensorFlow is not installed, torch-CPU operators stand in for TF-Keras CPU)", "rows": rows}


# ---------------------------------------------------------------------------------------------------------------------
# the other SURVEY 8(d) workloads, measured in the same run (N = 1)
# ---------------------------------------------------------------------------------------------------------------------
def time_graph(session, reps, torch):
    for _ in range(3):
        session.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        session.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def kernel_roofline(workload, rep, n_launch, alg, L):
    """roofline objects of a side workload's kernels: rep = kws_prof_report() over n_launch iterations, alg = {kernel: (bound, algorithmic
    bytes or flops per launch)}.  -> {"dominant": name, kernel: {bound, achieved, peak, unit, frac, avg_launch_ms, floors from the PMC record}}"""
    out = {}
    for k, (bound, amount) in alg.items():
        if k not in rep:
            continue
        ms = rep[k]["total_ms"] / rep[k]["count"]
        if bound == "mfma":
            ach, peak, unit = amount / (ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = amount / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        prec, note = pmc_record(workload, k, L.build_id())
        e = {"bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "avg_launch_ms": round(ms, 5),
             "ms_per_iteration": round(rep[k]["total_ms"] / n_launch, 5), "traffic": (prec or {}).get("derived", {}).get("hbm_bytes"), "pmc_note": note}
        e.update(floors_of(prec, ms))
        out[k] = e
    if out:
        out["dominant"] = max((k for k in out), key=lambda k: out[k]["ms_per_iteration"])
    return out


def extra_workloads(torch, pr, feat_fn, reps, which="all"):
    import kws_amd.lib as L
    from kws_amd.inference import InferenceSession
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec
    from kws_amd.pipeline import FeaturePipeline
    out = {}
    # `which`: one of fit / infer / gru / lite (bench.py runs every workload in a child process of its own: behind another workload in the same
    # process the pipelined simple_gru step measured 0.210 ms against 0.168 ms alone -- the featurizer did not overlap the recurrent kernels)
    if which in ("all", "fit"):
        # (v, run first: a fit() is what a fresh process of a user of the reference API does) the headline step behind the REFERENCE API: classifier.model.get_model(...).compile(...).fit(raw audio, batch_size=4096) -- the same
        # pipelined step (kws_amd.pipeline: in-place gather + featurize of the next batch on a side stream), shuffled epochs over a resident set
        from classifier.loss import SparseCategoricalCrossEntropy
        from classifier.model import get_model
        from common.model_utils import get_optimizer
        nb = 48                                    # batches per epoch: 12.6 GB of float32 audio resident in HBM
        wav_np, lab_np = synthetic_batch(4096, 0, N_CLASSES)
        xs = torch.from_numpy(wav_np).cuda().repeat(nb, 1)
        ys = torch.from_numpy(lab_np).cuda().repeat(nb)
        m = get_model("simple_cnn", N_CLASSES)
        m.compile(optimizer=get_optimizer("adam", 1e-3, decay_type=None), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
        h = m.fit(xs, ys, batch_size=4096, epochs=3, verbose=0, shuffle=True)
        cps = max(h.history["clips_per_sec"][1:])
        log("extra: fit() %.4f ms/step" % (4096.0 / cps * 1e3))
        out["fit_api_step"] = {"workload": "classifier.model.get_model('simple_cnn', 36).fit(raw audio (%d, 16000) resident in HBM, batch_size=4096, shuffle=True): "
                                           "best epoch of 2 after a warm-up epoch, %d steps per epoch, wall clock incl. the epoch's host sync" % (xs.shape[0], nb),
                               "ms_per_step": round(4096.0 / cps * 1e3, 4), "clips_per_s": round(cps, 1),
                               "epochs_clips_per_s": [round(v, 1) for v in h.history["clips_per_sec"]]}
        del xs, ys, m
        torch.cuda.empty_cache()
    if which in ("all", "infer"):
        # (i) featurize + simple_cnn inference forward, B = 4096: the north star's ">= 60 % of the HBM roofline" workload
        wav_np, lab_np = synthetic_batch(4096, 0, N_CLASSES)
        spec = ModelSpec("simple_cnn", N_CLASSES, pr.n_features, pr.feature_size)
        dm = DeviceModel(spec)
        dm.set_weights(init_weights(spec, seed=0))
        s = InferenceSession(dm, feat_fn, 4096, use_graph=True)
        s.wav.copy_(torch.from_numpy(wav_np))
        ms = time_graph(s, reps, torch)
        cps = 4096 / ms * 1e3
        eager = InferenceSession(dm, feat_fn, 4096, use_graph=False)      # per-kernel times of the same forward, eager
        eager.wav.copy_(s.wav)
        L.prof_enable(True)
        for _ in range(5):
            eager.run()
        rep = L.prof_report()
        L.prof_enable(False)
        log("extra: fwd_infer %.4f ms" % ms)
        out["fwd_infer"] = {"workload": "featurize (f32 in) + simple_cnn inference forward, B = 4096, one hipGraph replay per batch", "ms": round(ms, 4),
                            "clips_per_s": round(cps, 1), "hbm_roofline_clips_per_s": round(HBM_PEAK_GBS * 1e9 / FWD_BYTES_PER_CLIP, 1),
                            "hbm_roofline_frac": round(cps * FWD_BYTES_PER_CLIP / (HBM_PEAK_GBS * 1e9), 4),
                            "kernel_ms": {k: round(v["total_ms"] / 5, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}}
        # the FLOP-side ceiling next to the HBM fraction: the summed pipe floors of the forward's kernels from the PMC record of this workload
        fl = {}
        for k in rep:
            r, _ = pmc_record("infer" if not k.startswith("featurize") else "feat", k, L.build_id())
            for n, v in floors_of(r, None).items():
                if n.endswith("_floor_ms"):
                    fl[n] = round(fl.get(n, 0.0) + v, 5)
        if fl:
            out["fwd_infer"]["floors_ms_sum_over_kernels"] = fl
            out["fwd_infer"]["compute_ceiling_clips_per_s"] = round(4096 / max(fl.get("compute_floor_ms", 0.0), 1e-9) * 1e3, 1)
            out["fwd_infer"]["compute_ceiling_note"] = "sum over the forward's kernels of each kernel's largest pipe floor (vector ALU / LDS / matrix): what the counted work needs with perfect overlap inside every kernel and none between kernels"
        del s, eager, dm
    if which in ("all", "gru"):
        # (iii) simple_gru train step, B = 2048 (BASELINE configs[2]): featurize + fwd + bwd + Adam, pipelined like the headline step
        B = 2048
        wav_np, lab_np = synthetic_batch(B, 0, N_CLASSES)
        wav, labels = torch.from_numpy(wav_np).cuda(), torch.from_numpy(lab_np).cuda()
        spec = ModelSpec("simple_gru", N_CLASSES, pr.n_features, pr.feature_size)
        dm = DeviceModel(spec)
        dm.set_weights(init_weights(spec, seed=0))
        from kws_amd.featurizer import Featurizer
        # the recurrent step is light on LDS and registers: its pipeline keeps the featurizer's whole-chip configuration (same-box 0.306 ms
        # per step with the shared-mode featurizer, 0.286 with this one)
        pipe = FeaturePipeline(Featurizer(pr), B, pr.n_features, pr.feature_size, cu_share=2)
        ev = torch.cuda.Event()

        def gru_steps(n, k0):
            pipe.submit(wav)
            for i in range(n):
                feat = pipe.take()
                dm.train_fwd_bwd(feat, labels, dropout_seed=k0 + i + 1, overlap_event=ev,
                                 overlap_callback=(lambda: pipe.submit(wav, after=ev)) if i + 1 < n else None)
                dm.adam_step(1e-3)

        gru_steps(10, 0)
        torch.cuda.synchronize()
        nrep = 4 * reps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        gru_steps(nrep, 100)
        e1.record()
        host_ms = (time.perf_counter() - t0) / nrep * 1e3
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / nrep * 1e3
        log("extra: gru host enqueue %.4f ms/step, device %.4f ms/step" % (host_ms, e0.elapsed_time(e1) / nrep))
        log("extra: gru_train %.4f ms" % ms)
        out["gru_train"] = {"workload": "configs[2]: featurize + simple_gru fwd + bwd + Adam, B = 2048, 36 classes", "ms_per_step": round(ms, 4),
                            "clips_per_s": round(B / ms * 1e3, 1), "final_loss": round(float(dm.stats[0].item()) / B, 4)}
        # per-kernel times of the same pipelined execution, and the roofline of its kernels: the recurrence is 30 DEPENDENT steps per clip block
        # (16 clips per block: 128 blocks at B = 2048, i.e. half of the 256 CUs), so the kernels are latency-bound; the figures say how far
        L.prof_enable(True)
        gru_steps(10, 1000)
        torch.cuda.synchronize()
        rep = L.prof_report()
        L.prof_enable(False)
        T, F, H = pr.n_features, pr.feature_size, 48
        fwd_flops = 2.0 * B * T * (F * 3 * H + H * 3 * H)
        out["gru_train"]["kernel_ms"] = {k: round(v["total_ms"] / 10, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}
        out["gru_train"]["roofline"] = kernel_roofline("gru", rep, 10, {"gru_fwd_kernel": ("mfma", fwd_flops), "gru_bwd_kernel": ("mfma", 2.0 * fwd_flops + 2.0 * B * T * H * 3 * H),
                                                                      "featurize_fft1024_f32": ("hbm", B * FEAT_BYTES_PER_CLIP)}, L)
        del pipe, dm
    if which in ("all", "lite"):
        # (iv) simple_cnn_lite fp16 inference, B = 16 384, hipGraph-captured featurize + forward (BASELINE configs[4])
        B = 16384
        spec = ModelSpec("simple_cnn_lite", N_CLASSES, pr.n_features, pr.feature_size)
        lite = {}
        wav_np, _ = synthetic_batch(B, 0, N_CLASSES)
        for name, dt, nbytes in (("f32_in", torch.float32, 64144.0), ("pcm16_in", torch.int16, 32144.0)):
            dm = DeviceModel(spec)
            dm.set_weights(init_weights(spec, seed=0))
            s = InferenceSession(dm, feat_fn, B, wav_dtype=dt, use_graph=True, fp16=True)
            src = torch.from_numpy(wav_np)
            s.wav.copy_(src if dt == torch.float32 else (src * 32768.0).to(torch.int16))
            ms = time_graph(s, max(5, reps // 2), torch)
            log("extra: lite fp16 %s %.4f ms" % (name, ms))
            lite[name] = {"ms": round(ms, 4), "clips_per_s": round(B / ms * 1e3, 1), "bytes_per_clip": nbytes,
                          "hbm_roofline_frac": round(B / ms * 1e3 * nbytes / (HBM_PEAK_GBS * 1e9), 4)}
            if name == "f32_in":
                eager = InferenceSession(dm, feat_fn, B, wav_dtype=dt, use_graph=False, fp16=True)      # per-kernel times of the same forward, eager
                eager.wav.copy_(s.wav)
                L.prof_enable(True)
                for _ in range(5):
                    eager.run()
                torch.cuda.synchronize()
                rep = L.prof_report()
                L.prof_enable(False)
                lite["kernel_ms"] = {k: round(v["total_ms"] / 5, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}
                # algorithmic bytes: featurizer 64 000 in + 2 400 out; front kernel 2 400 in + 2 240 out (a2 as fp16); back kernel 2 240 in + 4 C out
                lite["roofline"] = kernel_roofline("lite16", rep, 5, {"featurize_fft1024_f32": ("hbm", B * FEAT_BYTES_PER_CLIP),
                                                                     "lite_front_infer_kernel": ("hbm", B * (2400.0 + 2240.0)),
                                                                     "lite_back_f16_kernel": ("hbm", B * (2240.0 + 4.0 * N_CLASSES))}, L)
                del eager
            del s, dm
        lite["workload"] = "configs[4]: featurize + simple_cnn_lite forward, fp16 activations / matrix operands with fp32 accumulation, B = 16384, one hipGraph replay per batch"
        out["lite_fp16_graph"] = lite
    torch.cuda.empty_cache()
    return out


def self_launch(n, argv):
    """Spawn `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv>` and relay it: rank 0's JSON line is the
    only thing the ranks write to stdout, so the child's stdout IS this process's stdout; the exit code is the child's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("self-launch: %s" % " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the rank processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """The control plane of an N-rank run without a GPU: process group (gloo), shard plan of one global batch (equal, uneven and
    short batches), one all-reduce over the ranks.  Rank 0 prints one JSON line; a failed check is a non-zero exit code."""
    import torch
    import torch.distributed as dist
    from kws_amd.parallel import DataParallel
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        return 2
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = DataParallel()
    ok = dp.world == world and dp.rank == rank
    plans = {}
    for n in (args.batch * world, args.batch * world - 1, world - 1, 0):
        lo, hi, w = dp.shard_plan(n)
        t = torch.tensor([float(hi - lo), w, float(lo)], dtype=torch.float64)
        rows = [torch.zeros_like(t) for _ in range(world)]
        if world > 1:
            dist.all_gather(rows, t)
        else:
            rows = [t]
        cover = sum(int(r[0]) for r in rows) == n and abs(sum(float(r[1]) for r in rows) - (1.0 if n else 0.0)) < 1e-12
        contiguous = all(int(rows[i][2]) + int(rows[i][0]) == (int(rows[i + 1][2]) if i + 1 < world else n) for i in range(world))
        ok = ok and cover and contiguous
        plans[str(n)] = [[int(r[2]), int(r[2]) + int(r[0])] for r in rows]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "n_gpus": world, "ok": bool(ok), "shard_plans": plans,
                          "config": {"parallelism": "dp%d" % world, "global_batch": args.batch * world}}))
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (weak scaling)")
    ap.add_argument("--cpu-budget", type=float, default=16.0, help="seconds of CPU work for the cpu_baseline rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other SURVEY 8(d) workloads (extra.*)")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--extra-only", nargs="?", const="all", default=None, choices=("all", "fit", "infer", "gru", "lite"),
                    help="(internal) run the SURVEY 8(d) side workloads (or one of them) and print their JSON")
    ap.add_argument("--force-comm", action="store_true",
                    help="run the RCCL exchange (kws_allreduce_grads) even in a one-rank world: rehearsal of the N > 1 code path on one GPU")
    ap.add_argument("--overlap-point", type=int, default=-1,
                    help="(tuning) where in the step the next batch's featurizer starts: kws_model_set_overlap_point, -1 = the library's choice")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: bring up the ranks (gloo), check the process group and the shard plan of the global batch, print a JSON line, exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as FRESH child processes, before this process has
        # imported torch or touched HIP (a process that initialised the GPU must never fork / exec into ranks)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.dry_run:
        raise SystemExit(dry_run(args))

    import torch
    if args.extra_only:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
        torch.cuda.set_device(0)
        from classifier.params import pr
        from kws_amd.featurizer import Featurizer
        print(json.dumps(extra_workloads(torch, pr, Featurizer(pr), 30, args.extra_only)))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world)     # control plane: barriers, the max over ranks, the RCCL id

    import kws_amd
    from classifier.params import pr
    from kws_amd import lib as L
    from kws_amd.featurizer import Featurizer
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec
    from kws_amd.parallel import KwsComm

    B = args.batch
    feat_fn = Featurizer(pr)
    spec = ModelSpec("simple_cnn", N_CLASSES, pr.n_features, pr.feature_size)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=0))   # glorot-uniform kernels, default_rng(0)
    if args.overlap_point >= 0:
        dm.set_overlap_point(args.overlap_point)
    wav_np, lab_np = synthetic_batch(B, rank, N_CLASSES)
    wav = torch.from_numpy(wav_np).cuda()
    labels = torch.from_numpy(lab_np).cuda()
    split = dm.grad_split
    from kws_amd.pipeline import FeaturePipeline
    # the pipeline's featurizer shares the chip with the train step (half of each CU's LDS, FeaturePipeline sets it), so it is
    # its own object: feat_fn keeps the whole chip for the stand-alone workloads under `extra`
    # cu_share=2 (the whole-chip configuration of the featurizer) since round 3's overlap point: behind conv3's forward the featurizer does
    # not really SHARE the CUs -- its blocks (336 of a SIMD's 512 registers) exclude conv4's forward (352) and the Dense + head kernel (224),
    # device-clock stamps show conv4 forward -> featurizer -> Dense + head one after the other -- so it may as well run in its faster
    # configuration (0.088 against 0.096 ms): same-box 0.5412 -> 0.5375 ms per step
    pipe = FeaturePipeline(Featurizer(pr), B, pr.n_features, pr.feature_size, moments=True, cu_share=2)   # + kws_feature_moments behind the featurizer
    # data path collective: the C ABI's RCCL communicator (csrc/kws_comm.hip), bootstrapped over the torch group's store.  Created
    # AFTER the model (whose side stream exists since DeviceModel()) and the pipeline's stream: HIP deals streams to hardware queues
    # in creation order and RCCL creates its own (include/kws.h: kws_model_bind_device)
    comm = None
    if world > 1:
        comm = KwsComm.from_torch_group()
    elif args.force_comm:
        comm = KwsComm.from_torch_group() if dist is not None else KwsComm.single()      # a one-rank world runs the same code
    step_no = [0]
    overlap_ev = torch.cuda.Event()

    def submit_next():
        pipe.submit(wav, after=overlap_ev)

    def run_steps(n):
        # n complete train steps = n featurizations + n (fwd + bwd + all-reduce + Adam), all enqueued inside this call.
        # The features of batch k+1 are computed on a side stream while batch k trains (kws_amd/pipeline.py); the first
        # batch of every call is not overlapped with anything.
        if n <= 0:
            return
        pipe.submit(wav)
        for i in range(n):
            step_no[0] += 1
            feat, mom = pipe.take()
            # next batch's features on the side stream, started at the step's overlap point (behind conv3's forward kernel, include/kws.h): the library
            # records overlap_ev there and calls back, so the featurizer launch also sits at that point in HOST order
            # data parallel: the step exchanges its gradients itself (kws_train_args.comm): the early bucket (conv4 + dense + head,
            # 82 % of the bytes) on the model's side stream right behind conv4's weight gradient, the late bucket + BatchNormalization
            # statistics behind the backward pass on this stream; Adam follows in stream order
            dm.train_fwd_bwd(feat, labels, dropout_seed=step_no[0], grad_scale=1.0 / world, overlap_event=overlap_ev,
                             overlap_callback=submit_next if i + 1 < n else None, feat_moments=mom, comm=comm, comm_state_weight=1.0 / world)
            # no pipe.release() here: the featurizer that rewrites this step's feature buffer (batch k+2) is ordered behind the NEXT
            # step's overlap event on this stream, i.e. behind every kernel of this step
            dm.adam_step(1e-3)

    def fence():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    log("warm-up: %d steps" % args.warmup)
    run_steps(args.warmup)
    fence()
    log("timed region: %d steps" % args.steps)
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(dm.stats[0].item()) / B
    log("timed region done: %.4f ms/step" % (elapsed / args.steps * 1e3))

    # per-kernel timing on the launch stream (HIP events inside the library), separate from the timed region
    # (every rank runs these steps -- they contain the all-reduce -- but only rank 0 records and reports)
    roofline, breakdown, breakdown_serial, allreduce_us = None, {}, {}, None
    if args.profile_steps > 0:
        if rank == 0:
            L.prof_enable(True)
        run_steps(args.profile_steps)          # the same pipelined execution as the timed region
        fence()
        if rank == 0:
            rep = L.prof_report()
            L.prof_enable(False)
            L.prof_enable(True)
        if comm is not None:
            comm.timing(True)
        ar = []
        for _ in range(args.profile_steps):    # and once more step by step, so that the featurizer runs alone: the
            run_steps(1)                       # per-kernel times of this pass are not stretched by sharing the chip
            if comm is not None:
                torch.cuda.synchronize()
                ar.append(comm.last_us())
        fence()
        if comm is not None:
            comm.timing(False)
            allreduce_us = {"early_bucket": round(float(np.mean([a[0] for a in ar if a[0] is not None] or [0.0])), 2),
                            "late_bucket_with_bn_statistics": round(float(np.mean([a[1] for a in ar if a[1] is not None] or [0.0])), 2),
                            "early_bucket_floats": int(dm.params.numel() - split), "late_bucket_floats": int(split),
                            "how": "HIP events around each RCCL launch on the stream it is enqueued on, mean over %d serial steps on rank 0" % len(ar)}
        if rank == 0:
            rep_serial = L.prof_report()
            L.prof_enable(False)
            for k, v in sorted(rep_serial.items(), key=lambda kv: -kv[1]["total_ms"]):
                breakdown_serial[k] = round(v["total_ms"] / args.profile_steps, 4)
    dense_head = None
    if rank == 0 and args.profile_steps > 0:
        models = kernel_models(B, N_CLASSES)
        tot = sum(v["total_ms"] for v in rep.values())
        name = max(rep, key=lambda k: rep[k]["total_ms"])
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
            breakdown[k] = round(v["total_ms"] / args.profile_steps, 4)
        avg_ms = rep[name]["total_ms"] / rep[name]["count"]
        bound, amount = models.get(name, ("hbm", 0.0))
        if bound == "mfma":
            ach, peak, unit = amount / (avg_ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = amount / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        # HBM bytes per launch and the pipe floors from the PMC passes recorded in profiles/ (bench.py cannot run rocprofv3 around itself); a
        # record is only used if it was measured on the SAME sources of the kernel's files as the loaded library was built from
        prec, traffic_note = pmc_record("feat_shared" if name.startswith("featurize") else "step", name, L.build_id())
        traffic = (prec or {}).get("derived", {}).get("hbm_bytes")
        alone_ms = rep_serial[name]["total_ms"] / rep_serial[name]["count"]
        roofline = {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_note, "avg_launch_ms": round(avg_ms, 5),
                    "share_of_step_kernel_time": round(rep[name]["total_ms"] / tot, 3),
                    # the same kernel when nothing else shares the chip (second profile pass): the pipelined step runs it
                    # next to the model kernels, which stretches its launch but shortens the step
                    "alone_launch_ms": round(alone_ms, 5), "alone_frac": round(ach * avg_ms / alone_ms / peak, 4)}
        # the FLOP-side ceiling next to the HBM fraction: what the counted vector-ALU / LDS / matrix work of a launch needs at the pipes' peak rates
        roofline.update(floors_of(prec, avg_ms))
        # the dense head (Dense(256->128) as a 2x1 'valid' conv + Dense(C) softmax head), forward and backward: the only dense
        # contractions of the model -> matrix-core figures (north star: "MFMA utilisation for the dense head")
        dh = {}
        fl_tot, ms_tot = 0.0, 0.0
        for k in ("dense_head_fused_kernel", "conv_bf16_fwd<128,128>", "conv_gemm_fwd<128,128>", "conv_bf16_dgrad<128,128>", "conv_dgrad<128,128>", "conv_wgrad<128,128>",
                  "head_fwd_bwd_kernel", "head_fwd_kernel", "head_bwd_kernel"):
            if k in rep_serial and k in models:
                ms_k = rep_serial[k]["total_ms"] / rep_serial[k]["count"]
                dh[k] = {"ms": round(ms_k, 5), "algorithmic_tflops": round(models[k][1] / (ms_k * 1e-3) / 1e12, 3)}
                fl_tot += models[k][1]
                ms_tot += ms_k
        pmc = {}
        for k in dh:
            r, _ = pmc_record("step", k, L.build_id())
            if r:
                pmc[k] = {n: r["derived"].get(n) for n in ("mfma_busy", "mfma_floor_ms", "valu_floor_ms", "lds_floor_ms", "hbm_read_MB", "hbm_write_MB")}
        pmc = pmc or None
        dense_head = {"kernels": dh, "algorithmic_tflops": round(fl_tot / (ms_tot * 1e-3) / 1e12, 3) if ms_tot else None,
                      "frac_of_fp32_matrix_peak": round(fl_tot / (ms_tot * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if ms_tot else None,
                      "note": "serial-pass HIP-event times; split-precision kernels issue 6 bf16 MFMA partial products per algorithmic MAC, so their "
                              "matrix-pipe rate is 6x the algorithmic one (peak %.0f TFLOP/s bf16)" % MFMA_BF16_PEAK_TFLOPS,
                      "mfma_busy_pmc": pmc}

    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        # the headline step with exact-fp32 MFMA products everywhere (KWS_MATRIX_FP32), same pipeline
        log("extra: fp32 MFMA step")
        dm.set_precision(matrix=L.MATRIX_FP32)
        run_steps(10)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(50)
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t1) / 50 * 1e3
        dm.set_precision(matrix=None)
        # The other workloads run in a CHILD process on the same GPU (this one idles meanwhile): measured in this process behind the
        # headline run, the pipelined simple_gru step took 0.42-0.54 ms against 0.29 ms in a process of its own (tools/grupipe.py; every
        # kernel of it ~2x slower, not the clock: a 3 s pause changed nothing) -- state the headline run leaves behind (its streams /
        # hardware-queue assignment) that a user of that workload would not have.
        log("extra workloads (child process)")
        import subprocess
        extra = {}
        for which in ("fit", "infer", "gru", "lite"):          # one process per workload (see extra_workloads)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--extra-only", which], stdout=subprocess.PIPE, stderr=sys.stderr,
                                   timeout=240, text=True)
                if r.returncode == 0 and r.stdout.strip():
                    extra.update(json.loads(r.stdout.strip().splitlines()[-1]))
                else:
                    extra["error_" + which] = "child exited %d" % r.returncode
            except (subprocess.TimeoutExpired, ValueError) as e:
                extra["error_" + which] = repr(e)
        extra["fp32_mfma_step"] = {"workload": "the headline train step with every matrix product on v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chains)",
                                   "ms_per_step": round(ms32, 4), "clips_per_s": round(B / ms32 * 1e3, 1)}
        extra["dense_head_mfma"] = dense_head
    elif rank == 0:
        extra = {"dense_head_mfma": dense_head}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_budget)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {"metric": METRIC, "value": round(value, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: simple_cnn train step = featurize(f32 1 s @16 kHz) + fwd + bwd + Adam, "
                                      "36 logits (background + 35 words), batch %d per GPU" % B,
                          "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(loss, 4),
                          "matrix_precision": "conv2/conv3/conv4/dense products as three-way bf16 splits on the bf16 matrix cores "
                                              "with fp32 accumulation (fp32-level error, kws_model_set_precision); conv1, conv3 "
                                              "data gradient, dense weight gradient and everything else fp32; extra.fp32_mfma_step is the all-fp32 number",
                          "input_pipeline": "features of batch k+1 (and their second moments for layer 1, kws_feature_moments) computed on a side stream during step k, started behind conv3's forward kernel (kws_train_args.overlap_event, best point of eleven swept, profiles/r03_overlap_sweep.txt; all K featurizations inside the timed region)",
                          "gradient_exchange": ("kws_train_args.comm (RCCL behind the C ABI): early bucket grads[%d:] on the model's side stream behind conv4's weight gradient, "
                                                "late bucket + BN moving statistics grouped on the main stream behind the backward pass" % split) if comm is not None else "none (one rank)",
                          "allreduce_us": allreduce_us, "rccl_ranks": comm.world if comm is not None else 0,
                          "rccl_version": comm.rccl_version if comm is not None else None,
                          "rccl_env": {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                                  "GPU_MAX_HW_QUEUES") if k in os.environ},
                          "train_step_hbm_roofline_frac": round(value / world * 229.8e3 / (HBM_PEAK_GBS * 1e9), 4),
                          "library_build": L.build_id().get("kws_featurize.hip")},
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra, "kernel_ms_per_step": breakdown,
               "kernel_ms_per_step_serial": breakdown_serial}
        print(json.dumps(out))
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
