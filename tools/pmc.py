"""Collect rocprofv3 PMC counters for the kernels a python script launches and aggregate them per kernel.

    python tools/pmc.py <tag> <filter-substring> <set>[,<set>...] -- script.py [args]

Every counter SET is its own rocprofv3 pass (--pmc ... --kernel-trace, csv) as MI355X_MICROARCH.md prescribes (TCC FETCH_SIZE and
WRITE_SIZE never share a pass; at most 8 SQ counters per pass).  Output: gpurun_out/pmc_<tag>.json with the mean counter value
per launch of every kernel whose name contains the filter, plus the library build id.  Named sets:
    sqa   wave life: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
    sqb   pipes:     SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
    lds   LDS:       SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
    mfma  matrix:    SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
    fetch FETCH_SIZE        write WRITE_SIZE
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {
    "sqa": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU",
    "sqb": "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD",
    "lds": "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE",
    "mfma": "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}


def main():
    tag, filt, sets = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    cmd = sys.argv[sys.argv.index("--") + 1:]
    out = {"_command": " ".join(cmd), "_how": "rocprofv3 --pmc <set> --kernel-trace, one pass per set; mean per launch"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    for name in sets:
        d = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, name))
        r = subprocess.run(["rocprofv3", "--pmc"] + SETS[name].split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                           "python3"] + cmd, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("rocprofv3 pass %s failed" % name)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        agg = {}
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    if filt not in k:
                        continue
                    a = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
                    a[0] += float(row["Counter_Value"])
                    a[1].add(row["Dispatch_Id"])
        for k, cs in agg.items():
            e = out.setdefault(k, {})
            for cn, (tot, disp) in cs.items():
                e[cn] = tot / max(1, len(disp))
            e["_launches_" + name] = max(len(d2) for _, d2 in cs.values())
        print("pass %s: %d kernels matched" % (name, len(agg)), flush=True)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tf-keras-speech-commands_amd"))
        from kws_amd import lib
        out["_build_id"] = lib.build_id()
    except Exception as e:          # the record is still useful without it
        out["_build_id"] = str(e)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_%s.json" % tag), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main()
