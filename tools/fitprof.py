"""Where the host's time goes in KWSModel.fit at B = 4096 (raw audio): python tools/fitprof.py [batches per epoch]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import torch
import bench
from classifier.loss import SparseCategoricalCrossEntropy
from classifier.model import get_model
from common.model_utils import get_optimizer
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
wav_np, lab_np = bench.synthetic_batch(4096, 0, 36)
xs = torch.from_numpy(wav_np).cuda().repeat(nb, 1)
ys = torch.from_numpy(lab_np).cuda().repeat(nb)
m = get_model("simple_cnn", 36)
m.compile(optimizer=get_optimizer("adam", 1e-3, decay_type=None), loss=SparseCategoricalCrossEntropy(), metrics=["accuracy"])
h = m.fit(xs, ys, batch_size=4096, epochs=3, verbose=0, shuffle=True)
print("clips/s per epoch:", [round(v) for v in h.history["clips_per_sec"]], "-> ms/step", [round(4096e3 / v, 4) for v in h.history["clips_per_sec"]])
pr = cProfile.Profile()
pr.enable()
h = m.fit(xs, ys, batch_size=4096, epochs=2, verbose=0, shuffle=True)
pr.disable()
print("profiled epochs ms/step", [round(4096e3 / v, 4) for v in h.history["clips_per_sec"]])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:4500])
