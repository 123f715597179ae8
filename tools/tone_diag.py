import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tf-keras-speech-commands_amd")
import numpy as np, torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
from oracle import featurizer_oracle as fo
g = np.load("/root/repo/tests/golden/featurizer_golden.npz")
t = g["syn_tone_audio"].astype(np.float32)
got = Featurizer(pr)(torch.from_numpy(t).cuda()[None])[0].cpu().numpy().astype(np.float64)
want = g["refpy_mel_syn_tone"]
N = 20
n = np.arange(N)
D = np.cos(np.pi * (n[None, :] + 0.5) * n[:, None] / N) * np.where(n[:, None] == 0, np.sqrt(1.0 / N), np.sqrt(2.0 / N))   # D[k][n]
def bands(c):
    c = c.copy(); c[:, 0] = 0.0
    return c @ D
level = np.log(np.clip(fo.power_spec(t.astype(np.float64), 1024, 512, 1024) @ fo.bank().T, 2.220446049250313e-16, None))
delta = bands(got) - bands(want)
top = level.argmax(1)
delta = delta - delta[np.arange(30), top][:, None]
rel = level - level.max(1, keepdims=True)
for lo, hi in [(-5, 0.1), (-10, -5), (-15, -10), (-18, -15), (-20, -18), (-22, -20), (-25, -22), (-30, -25), (-40, -30)]:
    m = (rel >= lo) & (rel < hi)
    print("level in [%5.1f, %5.1f): %4d band-frames, max |delta| %.3e" % (lo, hi, m.sum(), np.abs(delta[m]).max() if m.any() else 0))
print("c0 err", np.abs(got[:, 0] - want[:, 0]).max(), "overall", np.abs(got - want).max())
