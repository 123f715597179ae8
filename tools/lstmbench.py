import sys, os
sys.path.insert(0, "/root/repo/tf-keras-speech-commands_amd")
import torch, kws_amd
from kws_amd.model import DeviceModel, ModelSpec
from kws_amd.init import init_weights
spec = ModelSpec("simple_lstm", 36, 30, 20); dm = DeviceModel(spec); dm.set_weights(init_weights(spec, 0))
B = 2048
x = torch.randn((B, 30, 20), device="cuda"); y = torch.randint(0, 36, (B,), device="cuda", dtype=torch.int32)
for _ in range(5): dm.train_fwd_bwd(x, y, dropout_seed=3); dm.adam_step()
torch.cuda.synchronize()
kws_amd.lib.prof_enable(True)
for i in range(10): dm.train_fwd_bwd(x, y, dropout_seed=i + 1); dm.adam_step()
torch.cuda.synchronize()
for k, v in sorted(kws_amd.lib.prof_report().items(), key=lambda kv: -kv[1]["total_ms"])[:3]: print("  %-28s %.4f ms" % (k, v["total_ms"] / 10))
