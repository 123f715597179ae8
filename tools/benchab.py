"""Run bench.py against another build of the library: python tools/benchab.py <lib | -> [bench args]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
import kws_amd.lib as L
lib = sys.argv[1]
if lib != "-":
    L.LIB_PATH = os.path.abspath(lib)
sys.argv = ["bench.py"] + sys.argv[2:]
import json, io, contextlib
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(lib, d["value"], d["ms_per_step"])
for k, v in list(d.get("kernel_ms_per_step_serial", d["kernel_ms_per_step"]).items())[:int(os.environ.get("KWS_AB_ROWS", "12"))]:
    print("   %-32s %.4f" % (k, v))
