"""Per-step device time of the first N pipelined train steps after start-up (events behind every Adam step) and the host's enqueue time."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
import bench
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.pipeline import FeaturePipeline
B, N = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 80
spec = ModelSpec("simple_cnn", 36, pr.n_features, pr.feature_size)
dm = DeviceModel(spec); dm.set_weights(init_weights(spec, seed=0))
wav_np, lab_np = bench.synthetic_batch(B, 0, 36)
wav, labels = torch.from_numpy(wav_np).cuda(), torch.from_numpy(lab_np).cuda()
pipe = FeaturePipeline(Featurizer(pr), B, pr.n_features, pr.feature_size, moments=True)
ov = torch.cuda.Event()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
torch.cuda.synchronize()
pipe.submit(wav)
evs[0].record()
for i in range(N):
    t0 = time.perf_counter()
    feat, mom = pipe.take()
    dm.train_fwd_bwd(feat, labels, dropout_seed=i + 1, grad_scale=1.0, overlap_event=ov,
                     overlap_callback=(lambda: pipe.submit(wav, after=ov)) if i + 1 < N else None, feat_moments=mom)
    dm.adam_step(float(os.environ.get("KWS_EXP_LR", "1e-3")))
    evs[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
if os.environ.get("KWS_EXP_PAUSE"):          # idle, then the same N steps again: does the slow start come back?
    first = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    print("first pass: steps 0-4 %.4f, last five %.4f ms/step" % (np.mean(first[:5]), np.mean(first[-5:])))
    time.sleep(float(os.environ["KWS_EXP_PAUSE"]) * 1e-3)
    pipe.submit(wav)
    evs[0].record()
    for i in range(N):
        feat, mom = pipe.take()
        dm.train_fwd_bwd(feat, labels, dropout_seed=i + 1, grad_scale=1.0, overlap_event=ov,
                         overlap_callback=(lambda: pipe.submit(wav, after=ov)) if i + 1 < N else None, feat_moments=mom)
        dm.adam_step(1e-3)
        evs[i + 1].record()
    torch.cuda.synchronize()
dev = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
for i in range(0, N, 5):
    print("steps %3d-%3d  device %.4f ms/step   host enqueue %.4f ms/step" % (i, i + 4, np.mean(dev[i:i + 5]), np.mean(host[i:i + 5])))
