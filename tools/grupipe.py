"""simple_gru train step at B = 2048 with the featurizer (a) in-stream, (b) pipelined with 1 block per CU, (c) pipelined with 2"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.pipeline import FeaturePipeline
import bench
B = 2048
wav_np, lab_np = bench.synthetic_batch(B, 0, 36)
wav, labels = torch.from_numpy(wav_np).cuda(), torch.from_numpy(lab_np).cuda()
spec = ModelSpec("simple_gru", 36, 30, 20)
dm = DeviceModel(spec); dm.set_weights(init_weights(spec, seed=0))
feat = Featurizer(pr)
def timeit(fn, n=100):
    fn(10); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def model_only(n):
    x = feat(wav)
    for i in range(n): dm.train_fwd_bwd(x, labels, dropout_seed=i + 1); dm.adam_step(1e-3)
def instream(n):
    for i in range(n):
        x = feat(wav); dm.train_fwd_bwd(x, labels, dropout_seed=i + 1); dm.adam_step(1e-3)
print("model only %.4f ms, featurizer in-stream %.4f ms" % (timeit(model_only), timeit(instream)))
for share in (1, 2):
    pipe = FeaturePipeline(Featurizer(pr), B, 30, 20)
    pipe.featurizer.set_cu_share(share)
    ev = torch.cuda.Event()
    def piped(n):
        pipe.submit(wav)
        for i in range(n):
            x = pipe.take()
            dm.train_fwd_bwd(x, labels, dropout_seed=i + 1, overlap_event=ev, overlap_callback=(lambda: pipe.submit(wav, after=ev)) if i + 1 < n else None)
            dm.adam_step(1e-3)
    print("pipelined, %d block(s) per CU: %.4f ms" % (share, timeit(piped)))
