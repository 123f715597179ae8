import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.inference import InferenceSession
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
feat = Featurizer(pr)
for mt, B in (("simple_cnn_lite", 16384), ("simple_cnn", 4096), ("simple_gru", 16384)):
    spec = ModelSpec(mt, 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
    for graph in (False, True):
        s = InferenceSession(dm, feat, B, use_graph=graph)
        s.wav.copy_(0.1 * torch.randn((B, 16000), device="cuda"))
        for _ in range(3): s.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): s.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        p1 = s.probs.clone()
        print("%-16s B=%5d graph=%-5s %.3f ms  %.2f Mclips/s  (%.1f%% of the 8 TB/s roofline at 64144 B/clip)" % (mt, B, graph, ms, B / ms / 1e3, B / ms / 1e3 * 64144 / 8e6 * 100))
    eager = InferenceSession(dm, feat, B, use_graph=False); eager.wav.copy_(s.wav); eager.run()
    print("   graph == eager:", torch.equal(eager.probs, p1))
# per-kernel breakdown of the simple_cnn B=4096 inference (eager, HIP-event profiler of the library)
import kws_amd.lib as L
spec = ModelSpec("simple_cnn", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
s = InferenceSession(dm, feat, 4096, use_graph=False)
s.wav.copy_(0.1 * torch.randn((4096, 16000), device="cuda"))
for _ in range(3): s.run()
L.prof_enable(True)
for _ in range(10): s.run()
rep = L.prof_report(); L.prof_enable(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("   %-34s %.4f ms" % (k, v["total_ms"] / 10))
# PCM16 input (the reference's native sample format, buffer_to_audio: int16 / 32768): 34 400 B per clip instead of 66 400
for mt, B in (("simple_cnn_lite", 16384), ("simple_cnn", 4096)):
    spec = ModelSpec(mt, 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
    s = InferenceSession(dm, feat, B, wav_dtype=torch.int16, use_graph=True)
    s.wav.copy_((3000 * torch.randn((B, 16000), device="cuda")).to(torch.int16))
    for _ in range(3): s.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): s.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-16s B=%5d PCM16 in, hipGraph  %.3f ms  %.2f Mclips/s  (%.1f%% of the 8 TB/s roofline at 32144 B/clip)" % (mt, B, ms, B / ms / 1e3, B / ms / 1e3 * 32144 / 8e6 * 100))
# BASELINE configs[4]: simple_cnn_lite streaming inference, batch 16384, hipGraph-captured featurize + forward, fp16
# (fp16 activations / matrix operands behind the featurizer, fp32 accumulation; kws_set_inference_precision)
spec = ModelSpec("simple_cnn_lite", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
for wd, scale, nbytes in ((torch.float32, 0.1, 64144), (torch.int16, 3000, 32144)):
    res = {}
    for fp16 in (False, True):
        s = InferenceSession(dm, feat, 16384, wav_dtype=wd, use_graph=True, fp16=fp16)
        torch.manual_seed(0)
        s.wav.copy_((scale * torch.randn((16384, 16000), device="cuda")).to(wd))
        for _ in range(3): s.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): s.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[fp16] = (s.probs.clone(), s.argmax.clone())
        print("simple_cnn_lite  B=16384 %s in, hipGraph, %s  %.3f ms  %.2f Mclips/s  (%.1f%% of the 8 TB/s roofline at %d B/clip)" % (
            "PCM16" if wd == torch.int16 else "f32", "fp16" if fp16 else "fp32", ms, 16384 / ms / 1e3, 16384 / ms / 1e3 * nbytes / 8e6 * 100, nbytes))
    print("   fp16 vs fp32: max |dp| %.2e, argmax agreement %.4f" % (float((res[True][0] - res[False][0]).abs().max()),
                                                                  float((res[True][1] == res[False][1]).float().mean())))
L.prof_enable(True)
s = InferenceSession(dm, feat, 16384, use_graph=False, fp16=True)
s.wav.copy_(0.1 * torch.randn((16384, 16000), device="cuda"))
L.prof_report()
for _ in range(10): s.run()
rep = L.prof_report(); L.prof_enable(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("   %-34s %.4f ms" % (k, v["total_ms"] / 10))
