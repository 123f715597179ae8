import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.inference import InferenceSession
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
feat = Featurizer(pr)
for mt, B in (("simple_cnn", 4096), ("simple_cnn_lite", 16384)):
    for dt in (torch.float32, torch.int16):
        spec = ModelSpec(mt, 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
        s = InferenceSession(dm, feat, B, wav_dtype=dt, use_graph=True)
        x = 0.1 * torch.randn((B, 16000), device="cuda")
        s.wav.copy_(x if dt == torch.float32 else (x * 32768).to(torch.int16))
        for _ in range(5): s.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): s.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print("%-16s B=%5d %-14s hipGraph %.3f ms  %.2f Mclips/s" % (mt, B, str(dt), ms, B / ms / 1e3))
import kws_amd.lib as L
for dt in (torch.float32, torch.int16):
    spec = ModelSpec("simple_cnn", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
    s = InferenceSession(dm, feat, 4096, wav_dtype=dt, use_graph=False)
    x = 0.1 * torch.randn((4096, 16000), device="cuda")
    s.wav.copy_(x if dt == torch.float32 else (x * 32768).to(torch.int16))
    for _ in range(3): s.run()
    L.prof_enable(True)
    for _ in range(10): s.run()
    rep = L.prof_report(); L.prof_enable(False)
    print(dt, [(k, round(v["total_ms"] / 10, 4)) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])[:7]])
    print("   feature stats: mean %.4f std %.4f min %.3f max %.3f" % (float(s.features.mean()), float(s.features.std()), float(s.features.min()), float(s.features.max())))
