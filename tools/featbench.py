import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
f = Featurizer(pr)
B = 4096
wav = (0.1 * torch.randn((B, 16000), device="cuda")).contiguous()
out = torch.empty((B, 30, 20), device="cuda")
for _ in range(5): f(wav, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 50
for _ in range(N): f(wav, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("featurize B=%d: %.3f ms  %.2f Mclips/s  %.1f GB/s algorithmic" % (B, ms, B / ms / 1e3, B * 66400 / ms / 1e6))
w16 = (wav * 32768).to(torch.int16)
for _ in range(5): f(w16, out=out)
torch.cuda.synchronize(); e0.record()
for _ in range(N): f(w16, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("featurize i16 B=%d: %.3f ms  %.2f Mclips/s" % (B, ms, B / ms / 1e3))
