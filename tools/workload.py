"""One measured workload, a few iterations, nothing else -- the command rocprofv3 wraps (tools/r03_records.py):
    python3 tools/workload.py feat | feat_shared | gru | lstm | lite16 | infer | step
All use bench.py's synthetic batches at the BASELINE.json batch of the workload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from classifier.params import pr  # noqa: E402
from kws_amd.featurizer import Featurizer  # noqa: E402
from kws_amd.inference import InferenceSession  # noqa: E402
from kws_amd.init import init_weights  # noqa: E402
from kws_amd.model import DeviceModel, ModelSpec  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C = bench.N_CLASSES


def model(kind):
    spec = ModelSpec(kind, C, pr.n_features, pr.feature_size)
    dm = DeviceModel(spec)
    dm.set_weights(init_weights(spec, seed=0))
    return dm


if what in ("feat", "feat_shared"):
    f = Featurizer(pr)
    f.set_cu_share(1 if what == "feat_shared" else 2)
    wav = torch.from_numpy(bench.synthetic_batch(4096, 0, C)[0]).cuda()
    out = torch.empty((4096, pr.n_features, pr.feature_size), device="cuda")
    for _ in range(iters):
        f(wav, out=out)
elif what in ("gru", "lstm"):
    B = 2048
    dm = model("simple_gru" if what == "gru" else "simple_lstm")
    wav_np, lab_np = bench.synthetic_batch(B, 0, C)
    f = Featurizer(pr)
    wav, y = torch.from_numpy(wav_np).cuda(), torch.from_numpy(lab_np).cuda()
    for i in range(iters):
        x = f(wav)
        dm.train_fwd_bwd(x, y, dropout_seed=i + 1)
        dm.adam_step(1e-3)
elif what in ("lite16", "infer"):
    B = 16384 if what == "lite16" else 4096
    dm = model("simple_cnn_lite" if what == "lite16" else "simple_cnn")
    s = InferenceSession(dm, Featurizer(pr), B, use_graph=False, fp16=(what == "lite16"))
    s.wav.copy_(torch.from_numpy(bench.synthetic_batch(B, 0, C)[0]))
    for _ in range(iters):
        s.run()
elif what == "step":
    sys.argv = ["bench.py", "--steps", str(iters), "--warmup", "2", "--no-cpu-baseline", "--no-extra", "--profile-steps", "0"]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        bench.main()
else:
    raise SystemExit("unknown workload " + what)
torch.cuda.synchronize()
