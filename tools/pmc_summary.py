#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch."""
import csv, glob, sys, collections, re
def load(pattern):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"qc_fock_(tier|bm)_kernel<(\d+), (\d+)>", name)
            key = ("%s<%s,%s>" % m.groups() + " g=" + r.get("Grid_Size", "")) if m else name.split("(")[0][:40]
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
if __name__ == "__main__":
    tag = sys.argv[1]
    allc = collections.defaultdict(dict)
    for sub in ("sq", "sq2", "fetch", "write"):
        d = load("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, sub))
        for k, cs in d.items():
            for c, v in cs.items():
                allc[k][c] = sum(v) / len(v)
                allc[k]["_n"] = len(v)
    keys = sorted(allc, key=lambda k: -allc[k].get("SQ_BUSY_CYCLES", 0))
    cols = ["_n", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU",
            "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
            "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum"]
    print("kernel".ljust(22) + "".join(c.replace("SQ_", "")[:13].rjust(14) for c in cols))
    for k in keys[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
        print(k.ljust(22) + "".join(("%.4g" % allc[k].get(c, float("nan"))).rjust(14) for c in cols))
