"""Same-box A/B of an environment switch: python tools/ab_env.py VAR [rounds] -- alternates bench.py runs without / with VAR=1."""
import json, os, subprocess, sys
var, rounds = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rounds):
    for on in (False, True):
        env = dict(os.environ)
        env.pop(var, None)
        if on:
            env[var] = "1"
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "300", "--no-cpu-baseline"], env=env,
                             capture_output=True, text=True, timeout=300).stdout
        d = json.loads(out.strip().splitlines()[-1])
        sel = {k: v for k, v in d["kernel_ms_per_step_serial"].items() if len(sys.argv) > 3 and sys.argv[3] in k}
        print("%s=%d  %.4f ms/step  %s" % (var, on, d["ms_per_step"], sel), flush=True)
