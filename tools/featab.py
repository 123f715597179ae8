"""A/B the featurizer of two library builds: python tools/featab.py <lib.so | -> [B] [cu_share] [bench]
cu_share 2 = the chip to itself (two persistent blocks per CU), 1 = the configuration beside a train step; `bench` = bench.py's batch
(25 % of the clips left-padded) instead of plain noise.  Prints the best of 5 x 50 launches and a checksum of the features."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
sys.path.insert(0, ROOT)
import torch
import kws_amd.lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from classifier.params import pr
from kws_amd.featurizer import Featurizer
f = Featurizer(pr)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
share = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f.set_cu_share(share)
if len(sys.argv) > 4 and sys.argv[4] == "bench":
    import bench
    wav = torch.from_numpy(bench.synthetic_batch(B, 0, 36)[0]).cuda()
else:
    torch.manual_seed(0)
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
print("%s B=%d share=%d best %.4f ms  checksum %.9g  absmax %.6g" % (L.LIB_PATH.split("/")[-1], B, share, best, float(out.double().sum()), float(out.abs().max())))
