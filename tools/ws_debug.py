"""Pull intermediate tensors of a simple_cnn train step out of the device workspace (replica of carve_cnn in
csrc/kws_model.hip) and compare them with the oracle's captured intermediates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import numpy as np, torch
from test_model_gpu import build, rel_err
from oracle import model_oracle as mo
C, B = 5, 64
om, dm = build("simple_cnn", C, seed=2, perturb=False)
rng = np.random.default_rng(11)
protos = rng.standard_normal((C, 30, 20)) * 2
y2 = rng.integers(0, C, B); x2 = (protos[y2] + 0.5 * rng.standard_normal((B, 30, 20))).astype(np.float32)
ws = [w.copy() for w in om.get_weights()]
ws[0] = ws[0] + 1e-3 * np.sign(np.random.default_rng(1).standard_normal(ws[0].shape))
om.set_weights([w.astype(np.float64) for w in ws]); dm.set_weights([w.astype(np.float32) for w in ws])

def carve(B, C, training=True):
    al = lambda x: (x + 255) & ~255
    off = 0; out = {}
    def take(name, n):
        nonlocal off
        out[name] = (off, n); off = al(off + n * 4)
    H = [30, 15, 7, 4]; W = [20, 10, 5, 3]
    zs = [30 * 20 * 16, 15 * 10 * 32, 4 * 3 * 64, 4 * 3 * 128]; as_ = [15 * 10 * 16, 7 * 5 * 32, 4 * 3 * 64, 256]
    for i in range(4):
        take("z%d" % i, 0 if i == 0 else zs[i] * B); take("a%d" % i, as_[i] * B)
    for i in range(4): take("dwo%d" % i, 0)
    take("d1", B * 128); take("loss_i", B); take("correct_i", B)
    for i in range(4): take("coef%d" % i, 6 * 128)
    wn = [9 * 32 * 64, 9 * 64 * 128, 256 * 128]
    for t in range(3):
        for q in range(6): take("wsp%d_%d" % (t, q), (wn[t] + 1) // 2)
    take("partial", 1024 * 9 * 64 * 2)
    if training:
        take("dlogits", B * C); take("dd1", B * 128); take("da4", B * 256)
        for i in range(4): take("gz%d" % i, 0 if i == 0 else zs[i] * B)
        for i in range(4): take("ddw%d" % i, 0)
        for i in range(3): take("da%d" % i, as_[i] * B)
    return out, off

# oracle with captured backward signals
cap = {}
for i, L in enumerate(om.layers):
    def wrap(L=L, i=i, bw=L.backward):
        def f(dy):
            cap["dy%d" % i] = np.array(dy); dx = bw(dy); cap["dx%d" % i] = None if dx is None else np.array(dx); return dx
        return f
    L.backward = wrap()
    def wrapf(L=L, i=i, fw=L.forward):
        def f(x, training):
            y = fw(x, training); cap["y%d" % i] = np.array(y); return y
        return f
    L.forward = wrapf()
lo, _, _ = mo.train_forward_backward(om, x2.astype(np.float64), y2, dropout_seed=None)
dm.train_fwd_bwd(torch.from_numpy(x2).cuda(), torch.from_numpy(y2.astype(np.int32)).cuda(), dropout_seed=0)
torch.cuda.synchronize()
layout, total = carve(B, C)
need = dm.spec.workspace_bytes(B, True)
print("carve replica bytes", total, "library", need)
base = dm._ws.data_ptr(); aligned = (base + 255) & ~255
raw = dm._ws[aligned - base:].cpu().numpy()
def get(name, shape):
    off, n = layout[name]
    return raw[off:off + n * 4].view(np.float32).reshape(shape)
checks = [("a1 (pool1 out)", get("a0", (B, 15, 10, 16)), cap["y3"]), ("z2 (conv2 out)", get("z1", (B, 15, 10, 32)), cap["y4"]),
          ("a2 (pool2 out)", get("a1", (B, 7, 5, 32)), cap["y7"]),
          ("dz2 (into conv2.backward)", get("gz1", (B, 15, 10, 32)), cap["dy4"]), ("da1 (out of conv2.backward)", get("da0", (B, 15, 10, 16)), cap["dx4"]),
          ("da2 (into pool2.backward)", get("da1", (B, 7, 5, 32)), cap["dy7"]), ("dz3", get("gz2", (B, 4, 3, 64)), cap["dy8"])]
for name, g, w in checks:
    w = w.reshape(g.shape)
    e = np.abs(g - w)
    print("%-30s rel err %.2e   worst at %s  gpu %.5g oracle %.5g" % (name, e.max() / (np.abs(w).max() + 1e-30), np.unravel_index(e.argmax(), e.shape), g.ravel()[e.argmax()], w.ravel()[e.argmax()]))
dz = get("gz1", (B, 15, 10, 32)); wz = cap["dy4"].reshape(dz.shape)
e = np.abs(dz - wz) / (np.abs(wz).max())
print("dz2 error by row (max over b, x, c):", np.round(e.max(axis=(0, 2, 3)), 6))
print("dz2 error by col (max over b, y, c):", np.round(e.max(axis=(0, 1, 3)), 6))
