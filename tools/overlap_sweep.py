"""Same-box sweep of the step's overlap point (kws_model_set_overlap_point): python tools/overlap_sweep.py [rounds]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for r in range(rounds):
    for pt in (-1, 8, 9, 10, 0, 6, 1, 2, 3, 4, 5):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "300", "--no-cpu-baseline", "--no-extra", "--profile-steps", "0",
                              "--overlap-point", str(pt)], capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print("point %2d  %.4f ms/step" % (pt, d["ms_per_step"]), flush=True)
        except Exception:
            print("point %2d failed: %s" % (pt, out.stderr[-300:]), flush=True)
