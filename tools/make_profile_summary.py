#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/run_profiles.sh into the summaries committed under profiles/ and re-derive the roofline
fraction from them, so that a reader can check the bench line against the profiler:

    python tools/make_profile_summary.py <round tag> [gpurun_out/prof_<tag>]

Per workload it writes
  profiles/<tag>_<workload>_bench.json           the bench line of the profiled command
  profiles/<tag>_<workload>_kernel_stats.csv     per kernel, STEADY STATE ONLY: the dispatches of the timed SCF passes (the last
                                                 `steps` builds; warm-up run, stream tuner and Schwarz pass excluded), with count,
                                                 mean / min / max duration, and the same over all dispatches for comparison
  profiles/<tag>_pmc_<workload>.json             counters per kernel (mean per dispatch, steady state): HBM bytes of one Fock build
                                                 (2 x FETCH_SIZE + WRITE_SIZE, KB units -> bytes: MI355X_MICROARCH.md, HBM section),
                                                 SQ busy / wait split, f64 MFMA operations and busy cycles of the linear algebra
and prints roofline.frac recomputed from the kernel trace (mean span of a build = first Fock kernel start to last Fock kernel end)
and the algorithmic flops/bytes the bench line reports (qc_work_stats)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

FP64_PEAK_TF, HBM_PEAK_GBS, CLOCK_GHZ, N_SIMD = 78.6, 8000.0, 2.4, 256 * 4
FOCK = ("qc_fock_tier_kernel", "qc_fock_bm_kernel", "qc_fock_tier1_low_kernel")       # (the last: merged wide-ket launches of round 3)


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def read(pattern):
    rows = []
    for f in glob.glob(pattern):
        rows += list(csv.DictReader(open(f)))
    return rows


def steady_start(rows, steps, key_id, key_name):
    """dispatch id of the first kernel of the timed region: the earliest among the last `steps` launches of every Fock kernel name"""
    by = collections.defaultdict(list)
    for r in rows:
        if short(r[key_name]).startswith(FOCK):
            by[(short(r[key_name]), r.get("Grid_Size", r.get("Grid_Size_X")))].append(int(r[key_id]))
    full = [sorted(v) for v in by.values() if len(v) >= steps + 5]          # (the Schwarz pass launches other grids)
    return min(v[-steps] for v in full)


def main(tag, src):
    os.makedirs("profiles", exist_ok=True)
    for wl in ("h2o_ccpvtz", "c6h6_ccpvdz"):
        bj = os.path.join(src, wl + "_bench.json")
        if not os.path.exists(bj):
            continue
        line = json.loads(open(bj).read().strip().splitlines()[-1])
        steps = line["steps"]
        shutil.copy(bj, "profiles/%s_%s_bench.json" % (tag, wl))
        # ---- kernel trace -> steady-state stats
        tr = read(os.path.join(src, wl + "_trace", "*", "*kernel_trace.csv"))
        d0 = steady_start(tr, steps, "Dispatch_Id", "Kernel_Name")
        allk, st = collections.defaultdict(list), collections.defaultdict(list)
        builds = collections.defaultdict(lambda: [None, None])
        per_name_seen = collections.Counter()
        for r in sorted(tr, key=lambda r: int(r["Dispatch_Id"])):
            nm = short(r["Kernel_Name"]); t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            allk[nm].append((t1 - t0) / 1e3)
            if int(r["Dispatch_Id"]) >= d0:
                st[nm].append((t1 - t0) / 1e3)
                if nm.startswith(FOCK):
                    b = per_name_seen[nm]; per_name_seen[nm] += 1       # the k-th steady launch of this kernel belongs to build k
                    lo, hi = builds[b]
                    builds[b] = [t0 if lo is None else min(lo, t0), t1 if hi is None else max(hi, t1)]
        with open("profiles/%s_%s_kernel_stats.csv" % (tag, wl), "w") as f:
            f.write("# rocprofv3 --kernel-trace of `bench.py --workload %s --no-extras --no-cpu-baseline --steps %d --warmup 3`; steady = the timed SCF passes only\n" % (wl, steps))
            f.write("kernel,steady_calls,steady_mean_us,steady_min_us,steady_max_us,steady_total_us,all_calls,all_mean_us\n")
            for nm in sorted(st, key=lambda k: -sum(st[k])):
                v, a = st[nm], allk[nm]
                f.write("\"%s\",%d,%.2f,%.2f,%.2f,%.1f,%d,%.2f\n" % (nm, len(v), sum(v) / len(v), min(v), max(v), sum(v), len(a), sum(a) / len(a)))
        spans = [(hi - lo) / 1e3 for lo, hi in builds.values() if lo is not None]
        span_us = sum(spans) / len(spans)
        rf = line["roofline"]
        tf = rf["kernel_alg_flops"] / (span_us * 1e-6) / 1e12
        gbs = rf["kernel_alg_bytes"] / (span_us * 1e-6) / 1e9
        print("%s: %d steady builds, mean span of the Fock kernels %.1f us (bench line: kernel_ms %.1f us incl. memset / scale / fold / symmetrise)"
              % (wl, len(spans), span_us, rf["kernel_ms"] * 1e3))
        print("   recomputed from the trace: %.2f TFLOP/s = frac %.4f of the FP64 roof (bench line %.4f); %.0f GB/s = frac %.4f of the HBM roof"
              % (tf, tf / FP64_PEAK_TF, rf["frac"] if rf["bound"] == "fp64" else rf["other_roof"]["frac"], gbs, gbs / HBM_PEAK_GBS))
        # ---- PMC passes
        pmc = {"_workload": wl, "_source": "rocprofv3 --pmc passes of tools/run_profiles.sh (separate runs), steady-state dispatches only, mean per dispatch",
               "_trace_check": {"steady_builds": len(spans), "fock_kernels_span_us": span_us, "fp64_frac_from_trace": tf / FP64_PEAK_TF}}
        ctr = collections.defaultdict(lambda: collections.defaultdict(list))
        for sub in ("sq", "mfma", "fetch", "write"):
            rows = read(os.path.join(src, wl + "_" + sub, "*", "*counter_collection.csv"))
            if not rows:
                continue
            d0p = steady_start(rows, steps, "Dispatch_Id", "Kernel_Name")
            for r in rows:
                if int(r["Dispatch_Id"]) >= d0p:
                    ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in ctr.items()}
        calls = {k: max(len(v) for v in cs.values()) for k, cs in ctr.items()}
        fock, hbm = {}, 0.0
        for k, c in mean.items():
            if not k.startswith(FOCK):
                continue
            fs, ws = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
            ent = {"fetch_bytes_raw": None if fs is None else fs * 1024.0, "fetch_bytes_x2": None if fs is None else fs * 2048.0,
                   "write_bytes": None if ws is None else ws * 1024.0, "atomic_requests": c.get("TCC_EA0_ATOMIC_sum")}
            for s in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"):
                if s in c:
                    ent[s] = c[s]
            if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
                ent["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
                ent["valu_active_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
            fock[k] = ent
            if fs is not None:
                hbm += fs * 2048.0 + (ws or 0.0) * 1024.0
        pmc["fock_build"] = fock
        pmc["fock_build_hbm_bytes"] = hbm
        pmc["fock_build_hbm_note"] = ("sum over the build's launches of 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes).  FETCH_SIZE halves wide coalesced reads on gfx950 "
                                      "(MI355X_MICROARCH.md); these kernels' reads are 8-32 B gathers, for which the guide gives no calibration, so the doubled figure is an upper bound")
        eig = {}
        dur = {nm: sum(v) / len(v) for nm, v in st.items()}
        for k, c in mean.items():
            if k.startswith(FOCK) or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
                continue
            busy, mops = c["SQ_VALU_MFMA_BUSY_CYCLES"], c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
            if busy <= 0 and mops <= 0:
                continue
            d_us = dur.get(k)
            ent = {"calls_per_run": calls[k], "mean_us": d_us, "SQ_INSTS_VALU_MFMA_MOPS_F64": mops, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES")}
            if d_us:
                ent["mfma_util_of_chip"] = busy / (d_us * 1e-6 * CLOCK_GHZ * 1e9 * N_SIMD)
                ent["mfma_util_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs): the gfx94x MfmaUtil formula with the kernel's own duration"
            eig[k] = ent
        pmc["eigensolve"] = eig
        json.dump(pmc, open("profiles/%s_pmc_%s.json" % (tag, wl), "w"), indent=1)
        print("   HBM bytes per build (PMC, upper bound) %.1f MB vs algorithmic %.1f MB; MFMA kernels: %s"
              % (hbm / 1e6, rf["kernel_alg_bytes"] / 1e6, ", ".join("%s util %.2e" % (k, v.get("mfma_util_of_chip", 0)) for k, v in eig.items())))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    main(tag, sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_" + tag)
