"""Featurizer time vs. number of clips (blocks): the steps show how many blocks a CU really holds at once."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
f = Featurizer(pr)
wav = (0.1 * torch.randn((8192, 16000), device="cuda")).contiguous()
out = torch.empty((8192, 30, 20), device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in [64, 128, 256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192]:
    for _ in range(3): f(wav[:B], out=out[:B])
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f(wav[:B], out=out[:B])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("B=%5d  %.4f ms  %.2f us per 256 clips" % (B, ms, ms * 1e3 / max(1, B / 256)))
