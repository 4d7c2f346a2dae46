#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from rocprofv3 --pmc passes (tools/run_pmc.sh <workload> <tag>).

Per Fock kernel (qc_fock_tier_kernel<LAB, TIER>, qc_fock_bm_kernel<LCD, HI>): HBM traffic per launch from FETCH_SIZE / WRITE_SIZE (separate passes, KB units).  The same kernel
name is also launched with a single class bucket by bench.py's per-class profile, so dispatches are grouped by grid size
and the largest grid (= the full tier launch of a real build) is kept.  MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads (x2 correction); this kernel's reads are 8-32 B per lane gathers, which the guide
lists as uncalibrated, so both the raw and the doubled figure are given and `hbm_bytes` uses the doubled one (upper
bound).  WRITE_SIZE is exact for the f64 atomics that make up all of this kernel's writes."""
import collections, csv, glob, json, re, sys

def groups(tag, sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, sub)):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(qc_fock_tier_kernel|qc_fock_bm_kernel)<(\d+), (\d+)>", r["Kernel_Name"])
            if not m:
                continue
            out["%s<%s, %s>" % m.groups()][(int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out

def main(tag, workload):
    res = {"_workload": workload, "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE TCC_EA0_ATOMIC_sum, tools/run_pmc.sh", "_units": "bytes per launch"}
    fetch, write = groups(tag, "fetch"), groups(tag, "write")
    for k in sorted(fetch):
        gmax = max(g for g, _ in fetch[k])
        mean = lambda d, c: (sum(d[k][(gmax, c)]) / len(d[k][(gmax, c)])) if (gmax, c) in d[k] else None
        fs, wsz, at = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE"), mean(write, "TCC_EA0_ATOMIC_sum")
        res[k] = {"grid_threads": gmax, "fetch_bytes_raw": fs * 1024.0, "fetch_bytes_x2": 2048.0 * fs,
                  "write_bytes": None if wsz is None else wsz * 1024.0, "atomic_requests": at,
                  "hbm_bytes": 2048.0 * fs + (0 if wsz is None else wsz * 1024.0)}
    json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
    for k, v in res.items():
        print(k, v)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
