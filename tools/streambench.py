"""Throughput of the streaming path: S lock-stepped streams, one 1024-sample chunk each per step
(update_vectors -> model forward -> decode -> trigger; kws_amd.stream.StreamBatch)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
import kws_amd.lib as L
from classifier.params import pr
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.stream import StreamBatch
for mt in ("simple_cnn_lite", "simple_cnn"):
    for S in (1024, 16384):
        spec = ModelSpec(mt, 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
        names = ["background"] + ["w%d" % i for i in range(35)]
        sb = StreamBatch(pr, dm, S, chunk_size=1024, class_names=names)
        chunk = (torch.randn((S, 1024), device="cuda") * 3000).to(torch.int16)
        for _ in range(35): sb.push(chunk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n): sb.push(chunk)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        print("%-16s S=%5d  %.3f ms per chunk step  %.2f M chunks/s  = %.0f x real time per stream (64 ms of audio per chunk)" % (mt, S, ms, S / ms / 1e3, 64.0 / ms))
        if S == 16384:
            L.prof_enable(True)
            for _ in range(10): sb.push(chunk)
            rep = L.prof_report(); L.prof_enable(False)
            for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
                print("      %-34s %.4f ms" % (k, v["total_ms"] / 10))
