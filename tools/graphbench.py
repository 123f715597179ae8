"""How much would a hipGraph of the train step save?  Eager vs replay of train_fwd_bwd (+ Adam), features precomputed."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
from classifier.params import pr
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
B = 4096
spec = ModelSpec("simple_cnn", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
feat = torch.randn((B, 30, 20), device="cuda") * 3
labels = torch.randint(0, 36, (B,), device="cuda", dtype=torch.int32)
def step():
    dm.train_fwd_bwd(feat, labels, dropout_seed=7)
    dm.adam_step(1e-3)
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager           %.4f ms" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("graph replay    %.4f ms" % timeit(g.replay))
print("eager again     %.4f ms" % timeit(step))
