"""The pipelined simple_gru train step of bench.py's extra.gru_train, alone (for rocprofv3 --kernel-trace + tools/timeline.py):
python3 tools/gru_pipelined.py [steps] [cu_share]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import time
import torch
import bench
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.pipeline import FeaturePipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
share = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 2048
wav_np, lab_np = bench.synthetic_batch(B, 0, 36)
wav, labels = torch.from_numpy(wav_np).cuda(), torch.from_numpy(lab_np).cuda()
spec = ModelSpec(sys.argv[3] if len(sys.argv) > 3 else "simple_gru", 36, pr.n_features, pr.feature_size)
dm = DeviceModel(spec); dm.set_weights(init_weights(spec, seed=0))
pipe = FeaturePipeline(Featurizer(pr), B, pr.n_features, pr.feature_size, cu_share=share)
ev = torch.cuda.Event()
def steps(k, k0):
    pipe.submit(wav)
    for i in range(k):
        feat = pipe.take()
        dm.train_fwd_bwd(feat, labels, dropout_seed=k0 + i + 1, overlap_event=ev,
                         overlap_callback=(lambda: pipe.submit(wav, after=ev)) if i + 1 < k else None)
        dm.adam_step(1e-3)
steps(10, 0); torch.cuda.synchronize()
t0 = time.perf_counter(); steps(n, 100); torch.cuda.synchronize()
print("%.4f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
