"""Round-3 measurement records (run on the GPU box):   python3 tools/r03_records.py [workload ...] [post]

For every workload of tools/workload.py (feat, feat_shared, step, infer, gru, lstm, lite16):
  * `rocprofv3 --kernel-trace --stats` -> gpurun_out/r03_<workload>_rocprofv3_kernel_stats.csv (per-kernel average duration)
  * tools/pmc.py: one rocprofv3 --pmc pass per counter set (FETCH_SIZE, WRITE_SIZE, the wave-life, pipe, LDS and matrix sets), --kernel-trace only
and one summary gpurun_out/r03_pmc.json: per workload and kernel the counters, the derived figures, and the pipe FLOORS bench.py prints next
to every HBM fraction:
    valu_floor_ms  = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz)       2 cycles per wave64 instruction on a SIMD-32 with >= 2 waves per SIMD
                                                                              (tools/valu_calib.hip on this GPU: 2.0 at the clock the loop sustains)
    mfma_floor_ms  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz)
    lds_floor_ms   = SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz)                  LDS-array cycles; the address / data transfer of a store is on top
    hbm_floor_ms   = HBM bytes by counters / 8 TB/s
Copy the files into profiles/ to publish.  Every record carries the sha1 of the library's sources (kws_build_id); bench.py refuses a stale one.
FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (section HBM) prescribes for gfx950 wide streaming reads."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
CLOCK_HZ, SIMDS, CUS, HBM = 2.4e9, 1024, 256, 8.0e12
SETS = {"feat": "fetch,write,sqa,sqb,lds", "feat_shared": "fetch,write,sqa,sqb,lds", "step": "fetch,write,mfma,lds,sqa", "infer": "fetch,write,mfma,lds,sqa",
        "gru": "fetch,write,mfma,lds,sqa", "lstm": "fetch,write,mfma,lds,sqa", "lite16": "fetch,write,mfma,lds,sqa"}
# short names (the ones kws_prof_report / bench.py use) of the kernels the bench line carries roofline objects for
SHORT = [("featurize_fft1024_v3_kernel<float", "featurize_fft1024_f32"), ("featurize_fft1024_v3_kernel<short", "featurize_fft1024_i16"),
         ("gru_fwd_kernel", "gru_fwd_kernel"), ("gru_bwd_kernel", "gru_bwd_kernel"), ("lstm_fwd_kernel", "lstm_fwd_kernel"),
         ("lstm_bwd_kernel", "lstm_bwd_kernel"), ("lite_front_infer_kernel", "lite_front_infer_kernel"), ("lite_back_f16_kernel", "lite_back_f16_kernel"),
         # the train step's matrix kernels under the names kws_prof_report gives them
         ("conv_wgrad_clip_bf16_kernel", "conv_wgrad_clip_bf16<16,32>"), ("conv_dgrad_clip_bf16_kernel", "conv_dgrad_clip_bf16<32,16>"),
         ("conv_fwd_clip_bf16_kernel<true", "conv_fwd_clip_bf16<16,32>"), ("conv_fwd_clip_bf16_kernel<false", "conv_fwd_clip_pool_bf16<16,32>"),
         ("conv_wgrad_bf16_kernel<64, 128", "conv_wgrad_bf16<64,128>"), ("conv_wgrad_bf16_kernel<32, 64", "conv_wgrad_bf16<32,64>"),
         ("conv_bf16_kernel<32, 64, 0, 0", "conv_bf16_fwd<32,64>"), ("conv_bf16_kernel<32, 64, 0", "conv_bf16_fwd_bn<32,64>"),
         ("conv_bf16_kernel<64, 128, 0", "conv_bf16_fwd<64,128>"), ("conv_bf16_kernel<128, 128, 0", "conv_bf16_fwd<128,128>"),
         ("conv_bf16_kernel<128, 64, 1", "conv_bf16_dgrad<128,64>"), ("conv_bf16_kernel<128, 128, 1", "conv_bf16_dgrad<128,128>"),
         ("conv_dgrad_direct_kernel<64, 32", "conv_dgrad<64,32>"), ("conv_wgrad_direct_kernel<128, 128", "conv_wgrad<128,128>"),
         ("head_bwd_mfma_kernel<true", "head_fwd_bwd_kernel"), ("l1f_bwd_onepass_kernel", "l1m_bwd_onepass_kernel"),
         ("l1m_act_pool_moments_kernel", "l1m_act_pool_kernel"), ("l1_moments_kernel", "l1_moments_kernel"), ("infer_tail_kernel", "infer_tail_kernel"),
         ("conv3_group_fwd_kernel", "conv_group_fwd<32,64>"), ("conv4_group_fwd_kernel", "conv_group_fwd<64,128>"),
         ("conv4_group_dgrad_kernel", "conv_group_dgrad<128,64>"), ("conv3_group_dgrad_kernel", "conv_group_dgrad<64,32>"),
         ("dense_head_fused_kernel", "dense_head_fused_kernel")]
ITERS = {"step": 5}          # iterations the workload command runs (workload.py: default 3; step = 3 timed + 2 warm-up bench steps)


def derived(c, launches_note=None):
    d = {}
    if "FETCH_SIZE" in c:
        d["hbm_read_MB"] = round(2.0 * c["FETCH_SIZE"] * 1024 / 1e6, 3)
    if "WRITE_SIZE" in c:
        d["hbm_write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 3)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes"] = int(2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
        d["hbm_floor_ms"] = round(d["hbm_bytes"] / HBM * 1e3, 5)
    if "SQ_INSTS_VALU" in c:
        d["valu_floor_ms"] = round(c["SQ_INSTS_VALU"] * 2.0 / (SIMDS * CLOCK_HZ) * 1e3, 5)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_floor_ms"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * CLOCK_HZ) * 1e3, 5)
    if "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_floor_ms"] = round(c["SQ_LDS_IDX_ACTIVE"] / (CUS * CLOCK_HZ) * 1e3, 5)
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 3)
        d["cu_busy_ms"] = round(c["SQ_BUSY_CU_CYCLES"] / (CUS * CLOCK_HZ) * 1e3, 5)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_per_active_cycle"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    floors = {k: v for k, v in d.items() if k.endswith("_floor_ms")}
    if floors:
        d["bound_by"] = max(floors, key=floors.get)[:-len("_floor_ms")]
    return d


def kernel_stats(name):
    """rocprofv3 --kernel-trace --stats around the workload; returns {kernel: avg ms} and leaves the CSV in gpurun_out/"""
    d = os.path.join(OUT, "r03_stats_" + name)
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "s", "--", "python3",
                        os.path.join(ROOT, "tools", "workload.py"), name, "6"], cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, TMPDIR="/tmp"))
    if r.returncode != 0:
        print(r.stdout[-2000:])
        raise SystemExit("rocprofv3 --stats failed for " + name)
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    avg = {}
    if files:
        dst = os.path.join(OUT, "r03_%s_rocprofv3_kernel_stats.csv" % name)
        shutil.copy(files[0], dst)
        with open(dst) as f:
            for row in csv.DictReader(f):
                avg[row["Name"]] = float(row["AverageNs"]) * 1e-6
    return avg


def main():
    args = [a for a in sys.argv[1:] if a != "post"]
    post = "post" in sys.argv[1:]
    names = args or list(SETS)
    path = os.path.join(OUT, "r03_pmc.json")
    out = {}
    # start from the published records (gpurun_out/ is empty on a fresh GPU box): a run over some workloads keeps the others
    for src in (path, os.path.join(ROOT, "profiles", "r03_pmc.json")):
        if os.path.exists(src):
            with open(src) as f:
                out = json.load(f)
            break
    from kws_amd import lib
    bid = lib.build_id()
    out["_how"] = __doc__
    for name in names:
        if post:
            with open(os.path.join(OUT, "pmc_r03%s.json" % name)) as f:
                rec = json.load(f)
            avg = (out.get(name) or {}).get("_avg_ms", {})
        else:
            avg = kernel_stats(name)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc.py"), "r03" + name, "", SETS[name], "--",
                                os.path.join(ROOT, "tools", "workload.py"), name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                print(r.stdout[-3000:])
                raise SystemExit("pmc.py failed for " + name)
            with open(os.path.join(OUT, "pmc_r03%s.json" % name)) as f:
                rec = json.load(f)
        kernels = {}
        for k, c in rec.items():
            if (k.startswith("_") and not k.startswith("_Z")) or "at::native" in k or "rocclr" in k or k.startswith("void at::") or "elementwise" in k:
                continue
            e = {n: v for n, v in c.items() if not n.startswith("_")}
            e["derived"] = derived(c)
            launches = max([v for n, v in c.items() if n.startswith("_launches_")] or [0])
            e["launches_per_iteration"] = round(launches / float(ITERS.get(name, 3)), 3)
            if k in avg:
                e["rocprofv3_avg_ms"] = round(avg[k], 5)
            short = [s for pat, s in SHORT if pat in k]
            kernels[short[0] if short else k] = dict(e, symbol=k)
        tot = sum(v["derived"].get("hbm_bytes", 0) * v["launches_per_iteration"] for v in kernels.values())
        out[name] = {"kernels": kernels, "_avg_ms": avg, "command": "python3 tools/workload.py " + name, "source_sha1": bid,
                     "hbm_bytes_per_iteration": int(tot),
                     "hbm_bytes_per_iteration_note": "sum over the workload's kernels of (HBM bytes per launch by counters) x (launches per iteration)"}
        print(name, {k: (v.get("rocprofv3_avg_ms"), v["derived"].get("bound_by")) for k, v in kernels.items()}, flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
