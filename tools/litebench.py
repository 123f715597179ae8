import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
import kws_amd.lib as L
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.inference import InferenceSession
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
feat = Featurizer(pr)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
spec = ModelSpec("simple_cnn_lite", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
s = InferenceSession(dm, feat, B, use_graph=False)
s.wav.copy_(0.1 * torch.randn((B, 16000), device="cuda"))
for _ in range(3): s.run()
L.prof_enable(True)
for _ in range(10): s.run()
rep = L.prof_report(); L.prof_enable(False)
tot = 0
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"]):
    tot += v["total_ms"] / 10
    print("   %-34s %.4f ms" % (k, v["total_ms"] / 10))
print("sum %.4f ms" % tot)
