python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py 2>/dev/null | tail -1 > gpurun_out/r3_bench_v3.json
python bench.py --steps 20 --warmup 5 --no-extra 2>/dev/null | tail -1 > gpurun_out/r3_bench_v3_driver.json
python -c "
import json
for f in ('gpurun_out/r3_bench_v3.json','gpurun_out/r3_bench_v3_driver.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('extra',{}).get('fit_api_step',{}))
"
