import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
import bench
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.pipeline import FeaturePipeline
B = 4096
feat_fn = Featurizer(pr)
spec = ModelSpec("simple_cnn", 36, pr.n_features, pr.feature_size)
dm = DeviceModel(spec); dm.set_weights(init_weights(spec, seed=0))
wav_np, lab_np = bench.synthetic_batch(B, 0, 36)
wav = torch.from_numpy(wav_np).cuda(); labels = torch.from_numpy(lab_np).cuda()
pipe = FeaturePipeline(feat_fn, B, pr.n_features, pr.feature_size)
def run(n):
    pipe.submit(wav)
    for i in range(n):
        feat = pipe.take()
        if i + 1 < n: pipe.submit(wav)
        dm.train_fwd_bwd(feat, labels, dropout_seed=i + 1, grad_scale=1.0)
        pipe.release()
        dm.adam_step(1e-3)
run(20); torch.cuda.synchronize()
for n in (300,):
    t0 = time.perf_counter(); run(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("steps %d: host enqueue %.3f ms/step, total %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# model only (features precomputed)
feat = pipe.bufs[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(300):
    dm.train_fwd_bwd(feat, labels, dropout_seed=i + 1, grad_scale=1.0); dm.adam_step(1e-3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("model only: host enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
