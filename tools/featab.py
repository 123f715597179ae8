"""A/B the featurizer of two library builds: python tools/featab.py <lib.so | -> [B]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-keras-speech-commands_amd"))
import torch
import kws_amd.lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from classifier.params import pr
from kws_amd.featurizer import Featurizer
f = Featurizer(pr)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wav = (0.1 * torch.randn((B, 16000), device="cuda")).contiguous()
out = torch.empty((B, 30, 20), device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    for _ in range(5): f(wav, out=out)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): f(wav, out=out)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
print("%s B=%d best %.4f ms" % (L.LIB_PATH.split("/")[-1], B, best))
