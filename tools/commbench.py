"""Host and device cost of kws_allreduce_grads (one-rank RCCL communicator) inside the train step."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import torch
from kws_amd.init import init_weights
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.parallel import KwsComm
B = 4096
spec = ModelSpec("simple_cnn", 36, 30, 20)
dm = DeviceModel(spec); dm.set_weights(init_weights(spec, seed=0))
x = torch.randn((B, 30, 20), device="cuda"); y = torch.randint(0, 36, (B,), device="cuda", dtype=torch.int32)
if os.environ.get("COMM_LATE"):
    for i in range(5): dm.train_fwd_bwd(x, y, dropout_seed=i + 1); dm.adam_step(1e-3)
    torch.cuda.synchronize()
comm = KwsComm.single() if not os.environ.get("NO_COMM") else None
ev = torch.cuda.Event()
split = dm.grad_split
def run(n, mode):
    th = 0.0
    for i in range(n):
        dm.train_fwd_bwd(x, y, dropout_seed=i + 1, comm=comm if mode == 1 else None)
        if mode == 2:
            t0 = time.perf_counter(); comm.allreduce_grads(dm.grads, split, dm.state, 1.0); th += time.perf_counter() - t0
        dm.adam_step(1e-3)
    return th / n * 1e3
for mode, name in (((0, "no exchange"),) if comm is None else ((0, "no exchange"), (1, "exchange inside the step"), (2, "exchange as its own call"), (0, "no exchange"))):
    run(20, mode); torch.cuda.synchronize()
    t0 = time.perf_counter(); th = run(200, mode); tq = (time.perf_counter() - t0) / 200 * 1e3
    torch.cuda.synchronize(); tt = (time.perf_counter() - t0) / 200 * 1e3
    print("%-28s step %.4f ms (host enqueue %.4f ms/step, of which the exchange call %.4f ms)" % (name, tt, tq, th))
