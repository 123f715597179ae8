"""Same-box A/B of library builds: python tools/ab_libs.py <rounds> <lib | -> <lib | -> ... [-- substring of kernel names to print]
Alternates `bench.py --steps 300 --no-cpu-baseline --no-extra` runs (one process each) over the given builds of libkws_hip.so
("-" = the in-tree one) and prints ms/step of every run; box-to-box spread is ~2 %, run-to-run on one box ~0.3 %."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sel = None
if "--" in args:
    sel = args[args.index("--") + 1]
    args = args[:args.index("--")]
rounds, libs = int(args[0]), args[1:]
for r in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "benchab.py"), lib, "--steps", "300", "--no-cpu-baseline", "--no-extra"],
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, KWS_AB_ROWS="40"))
        lines = out.stdout.strip().splitlines()
        head = [l for l in lines if not l.startswith("   ")]
        rows = [l.strip() for l in lines if l.startswith("   ") and sel and sel in l]
        print(head[-1] if head else out.stderr[-500:], " | ".join(rows), flush=True)
