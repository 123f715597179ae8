#!/usr/bin/env python3
"""bench.py -- the keyword-spotting train step on N MI355X GPUs (one process per GPU, RCCL for the gradient exchange).

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM:
    featurize (B x 16000 f32 waveforms -> B x 30 x 20 MFCC)  ->  simple_cnn forward (batch-stat BN, dropout)
    -> loss -> backward -> [sum-all-reduce of the flat gradient buffer when N > 1] -> Keras-form Adam.
Workload at N = 1: BASELINE.json configs[1] (simple_cnn, 36 logits = background + 35 Speech Commands v2 words,
batch 4096, HIP featurizer + fwd/bwd + Adam).  Weak scaling: every rank keeps B = 4096.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel, measured
live with HIP events on the launch stream (kws_prof_*), and `cpu_baseline` = the CPU oracle timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tf-keras-speech-commands_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

METRIC = "1 s@16 kHz clips/sec train-step, simple_cnn bs4096, 1/2/4/8 MI355X; eval top-1"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
N_CLASSES = 36


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
    name -> (bound, amount, unit) with bytes for HBM-bound kernels and flops for MFMA-bound ones."""
    f = 4
    z = [30 * 20 * 16, 15 * 10 * 32, 4 * 3 * 64, 4 * 3 * 128]          # pre-BN conv outputs per clip
    a = [15 * 10 * 16, 7 * 5 * 32, 4 * 3 * 64, 256]                    # activations per clip
    m = {}
    m["featurize_fft1024_f32"] = ("hbm", B * 66400.0)                   # 64000 in + 2400 out
    # layer 1 is recomputed from the feature map (kws_layer1.h): bytes = features (+ a1 / da1), never a z1-sized tensor
    m["l1_stats_kernel"] = ("hbm", B * 600.0 * f)
    m["l1_act_pool_kernel"] = ("hbm", B * (600 + a[0]) * f)
    m["l1_bwd_reduce_kernel"] = ("hbm", B * (600 + a[0]) * f)
    m["l1_bwd_wgrad_kernel"] = ("hbm", B * (600 + a[0]) * f)
    m["l1m_stats_kernel"] = m["l1_stats_kernel"]                      # the MFMA, wave-per-clip forms of the same passes
    m["l1m_act_pool_kernel"] = m["l1_act_pool_kernel"]
    m["l1m_bwd_reduce_kernel"] = m["l1_bwd_reduce_kernel"]
    m["l1m_bwd_wgrad_kernel"] = m["l1_bwd_wgrad_kernel"]
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
    # conv2 runs in the clip-resident LDS kernels (kws_conv.h: conv_fwd_clip / conv_dgrad_clip / conv_wgrad_clip)
    # (fp32 MFMA) and their three-way bf16 split forms (default precision)
    for k in ("conv_fwd_clip<16,32>", "conv_wgrad_clip<16,32>", "conv_dgrad_clip<32,16>", "conv_fwd_clip_bf16<16,32>",
              "conv_wgrad_clip_bf16<16,32>", "conv_dgrad_clip_bf16<32,16>"):
        m[k] = ("mfma", 2.0 * B * 150 * 9 * 16 * 32)
    m["conv_wgrad_bf16<64,128>"] = ("mfma", 2.0 * B * 12 * 9 * 64 * 128)
    m["conv_dgrad<128,64>"] = ("mfma", 2.0 * B * 12 * 9 * 64 * 128)
    m["conv_dgrad<128,128>"] = ("mfma", 2.0 * B * 256 * 128)
    for l in range(1, 4):
        L = ".L%d" % (l + 1)
        m["channel_stats_kernel" + L] = ("hbm", B * z[l] * f)
        m["bn_act_pool_kernel" + L] = ("hbm", B * (z[l] + a[l]) * f)
        m["bn_bwd_reduce_kernel" + L] = ("hbm", B * (2 * z[l] + a[l]) * f)
        m["bn_bwd_reduce_pool_kernel" + L] = ("hbm", B * (2 * z[l] + a[l]) * f)
        m["bn_bwd_apply_kernel" + L] = ("hbm", B * 3 * z[l] * f)
    m["adam_kernel"] = ("hbm", 134932 * 7.0 * f)
    return m


def cpu_baseline(sample_clips, steps, n_classes):
    """The CPU oracle (kind 'port'): C featurizer (OpenMP over clips) + numpy float32 model step, on the host cores."""
    from oracle import featurizer_oracle as fo
    from oracle import model_oracle as mo
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    wav, labels = synthetic_batch(sample_clips, 0, n_classes)
    model = mo.Model("simple_cnn", n_classes, dtype=np.float32).init_weights(0)
    opt = mo.Adam(1e-3)

    def step(i):
        feat = fo.featurize_batch(wav)
        mo.train_step(model, opt, feat, labels, dropout_seed=i + 1)

    step(0)
    t0 = time.time()
    for i in range(steps):
        step(i + 1)
    dt = time.time() - t0
    return {"value": sample_clips * steps / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "%d steps of %d synthetic clips (same generator as the GPU run) through oracle/: C featurizer with "
                      "OpenMP over clips + numpy float32 simple_cnn fwd/bwd/Adam (BLAS threads = cores)" % (steps, sample_clips)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (weak scaling)")
    ap.add_argument("--cpu-clips", type=int, default=512)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=10)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import kws_amd
    from classifier.params import pr
    from kws_amd.featurizer import Featurizer
    from kws_amd.init import init_weights
    from kws_amd.model import DeviceModel, ModelSpec

    B = args.batch
    feat_fn = Featurizer(pr)
    spec = ModelSpec("simple_cnn", N_CLASSES, pr.n_features, pr.feature_size)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=0))   # glorot-uniform kernels, default_rng(0)
    wav_np, lab_np = synthetic_batch(B, rank, N_CLASSES)
    wav = torch.from_numpy(wav_np).cuda()
    labels = torch.from_numpy(lab_np).cuda()
    from kws_amd.pipeline import FeaturePipeline
    pipe = FeaturePipeline(feat_fn, B, pr.n_features, pr.feature_size)
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
            feat = pipe.take()
            # next batch's features on the side stream, started behind this step's last forward convolution: the library
            # records overlap_ev there and calls back, so the featurizer launch also sits at that point in HOST order
            # (from there the main chain is small kernels, then matrix-bound ones)
            dm.train_fwd_bwd(feat, labels, dropout_seed=step_no[0], grad_scale=1.0 / world, overlap_event=overlap_ev,
                             overlap_callback=submit_next if i + 1 < n else None)
            # no pipe.release() here: the featurizer that rewrites this step's feature buffer (batch k+2) is ordered behind the NEXT
            # step's overlap event on this stream, i.e. behind every kernel of this step -- the extra event would cost 6 us per step
            if dist is not None:
                dist.all_reduce(dm.grads)      # RCCL sum over xGMI; 540 KB flat buffer
            dm.adam_step(1e-3)

    def fence():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(dm.stats[0].item()) / B

    # per-kernel timing on the launch stream (HIP events inside the library), separate from the timed region
    # (every rank runs these steps -- they contain the all-reduce -- but only rank 0 records and reports)
    roofline, breakdown, breakdown_serial = None, {}, {}
    if args.profile_steps > 0:
        if rank == 0:
            kws_amd.lib.prof_enable(True)
        run_steps(args.profile_steps)          # the same pipelined execution as the timed region
        fence()
        if rank == 0:
            rep = kws_amd.lib.prof_report()
            kws_amd.lib.prof_enable(False)
            kws_amd.lib.prof_enable(True)
        for _ in range(args.profile_steps):    # and once more step by step, so that the featurizer runs alone: the
            run_steps(1)                       # per-kernel times of this pass are not stretched by sharing the chip
        fence()
        if rank == 0:
            rep_serial = kws_amd.lib.prof_report()
            kws_amd.lib.prof_enable(False)
            for k, v in sorted(rep_serial.items(), key=lambda kv: -kv[1]["total_ms"]):
                breakdown_serial[k] = round(v["total_ms"] / args.profile_steps, 4)
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
        traffic = None       # HBM bytes per launch from the PMC passes recorded in profiles/ (bench.py cannot run rocprofv3 itself)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                rec = json.load(f).get(name)
            if rec and rec.get("batch") == B:
                traffic = rec["traffic_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        alone_ms = rep_serial[name]["total_ms"] / rep_serial[name]["count"]
        roofline = {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 5),
                    "share_of_step_kernel_time": round(rep[name]["total_ms"] / tot, 3),
                    # the same kernel when nothing else shares the chip (second profile pass): the pipelined step runs it
                    # next to the model kernels, which stretches its launch but shortens the step
                    "alone_launch_ms": round(alone_ms, 5), "alone_frac": round(ach * avg_ms / alone_ms / peak, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_clips, args.cpu_steps, N_CLASSES)

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
                                              "with fp32 accumulation (fp32-level error, kws_set_matrix_precision); conv1, conv3 "
                                              "data/weight gradients, dense weight gradient and everything else fp32",
                          "input_pipeline": "features of batch k+1 computed on a side stream during step k, started behind the last forward convolution (kws_train_args.overlap_event; all K featurizations inside the timed region)",
                          "train_step_hbm_roofline_frac": round(value / world * 229.8e3 / (HBM_PEAK_GBS * 1e9), 4)},
               "roofline": roofline, "cpu_baseline": cpu, "kernel_ms_per_step": breakdown,
               "kernel_ms_per_step_serial": breakdown_serial}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
