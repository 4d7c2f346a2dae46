#!/usr/bin/env python3
"""bench.py - SCF-iteration time and ERI shell-quartet throughput of the MI355X Hartree-Fock path.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A *step* is one pass of the SCF loop body (rhf.rs:67-88): direct-SCF Fock build over every unique shell quartet
(ERI evaluation + J/K digestion), F = H + G, commutator, DIIS, F' = X^T F X, eigensolve, new density, energy, rms -
through the step-wise C ABI (`qc_scf_iterate`), all operands resident in HBM before the timed region starts.

Workloads (BASELINE.json configs; inputs are the fixture files under data/, nothing is random):
  N = 1  -> H2O / cc-pVTZ RHF (configs[2], the configuration the metric is quoted on; 58 bf, 32 131 unique quartets)
  N > 1  -> C6H6 / cc-pVDZ RHF (configs[4]): the class-sorted quartet list is dealt across the ranks, every rank digests
            its shard and the partial Fock matrices are summed by one RCCL all-reduce per build (strong scaling).
            The N = 1 line carries the same workload's single-GPU numbers under "scaling_reference".
`value` = unique shell quartets enumerated per step x K / elapsed (max over ranks), whole job.
`ms_per_step` = SCF-iteration time.

Extra objects on the JSON line: "roofline" (dominant ERI class kernel, hipEvent-timed on the library's stream) and
"cpu_baseline" (the oracle - a CPU restatement of the reference algorithm - timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "h2o_ccpvtz": ("water", "cc-pVTZ", "H2O/cc-pVTZ RHF"),
    "c6h6_ccpvdz": ("benzene", "cc-pVDZ", "C6H6/cc-pVDZ RHF"),
    "h2o_sto3g": ("water", "STO-3G", "H2O/STO-3G RHF"),
    "c6h6_631gss": ("benzene", "6-31G_st_st", "C6H6/6-31G** RHF"),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TF = 78.6    # MI355X FP64 vector peak (datasheet)


def load(q, key):
    mol, basis, _ = WORKLOADS[key]
    b = q.BasisSet.load(os.path.join(ROOT, "data", "basis", basis + ".json"))
    return q.MolecularSystem.load(os.path.join(ROOT, "data", "mol", mol + ".json"), b)


def timed_steps(torch, dist, stepper, steps, warmup, world):
    for _ in range(warmup):
        stepper.iterate()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tm0 = stepper.timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.iterate()          # synchronises the library's stream at the end of every pass
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm1 = stepper.timings()
    stepper.timed = {k: (tm1[k] - tm0[k]) / steps for k in ("fock", "linalg")}   # hipEvent ms per timed step
    return dt


def roofline(torch, sysh, D_host, reps):
    """hipEvent timing (on the library's stream) of the kernels a Fock build launches.

    A build = at most 18 launches: `qc_fock_tier_kernel<LAB, TIER>` (all class buckets of bra class LAB with LCD <= 3 /
    LCD >= 4) and `qc_fock_bm_kernel<LCD, HI>` (bra-major kernels for ss / ps kets), concurrent on side streams.  The roofline entry is the tier kernel that takes the longest when
    each is launched alone; `fock_build` gives the same ratios for the whole (overlapped) build, `classes` the per-bucket
    view.  Algorithmic bytes/flops: SURVEY.md 8(d) per-quartet model summed over the quartets a launch processes."""
    import qchem_rs_amd as q
    dD = torch.from_numpy(D_host).cuda()
    dG = torch.zeros_like(dD)
    torch.cuda.synchronize()
    tp = sysh.fock_profile_tiers(dD.data_ptr(), dG.data_ptr(), reps)
    prof = sysh.fock_profile(dD.data_ptr(), dG.data_ptr(), max(1, reps // 2))
    k = int(tp["unit_ms"].argmax())
    ws = sysh.work_stats()
    cname = lambda c: ("bm<%d, %d>" % ((int(c) >> 8) & 15, (int(c) >> 4) & 15)) if int(c) >> 12 else "<%d, %d, %d>" % (int(c) >> 8, (int(c) >> 4) & 15, int(c) & 15)
    order = prof["class_ms"].argsort()[::-1][:10]
    top = [{"class": cname(prof["class_id"][i]), "ms": float(prof["class_ms"][i]),
            "quartets": int(prof["quartets"][i]), "GF": float(prof["flops"][i]) / 1e9} for i in order]
    tiers = [{"kernel": q.hf.unit_name(u), "ms_alone": float(tp["unit_ms"][u]), "quartets": int(tp["quartets"][u]),
              "alg_MB": float(tp["bytes"][u]) / 1e6, "alg_GF": float(tp["flops"][u]) / 1e9,
              "GBs_alone": float(tp["bytes"][u]) / (float(tp["unit_ms"][u]) * 1e-3) / 1e9,
              "TFLOPs_alone": float(tp["flops"][u]) / (float(tp["unit_ms"][u]) * 1e-3) / 1e12} for u in range(len(tp["unit_ms"])) if tp["quartets"][u] > 0]
    tot_ms = float(tp["total_ms"])
    gbs = float(ws.bytes_alg) / (tot_ms * 1e-3) / 1e9
    tfs = float(ws.flops_alg) / (tot_ms * 1e-3) / 1e12
    # The dominant operation is the Fock build: one kernel template (qc_fock_tier_kernel<LAB, TIER>), launched once per
    # non-empty (bra class, tier) - the launches overlap on side streams, so no single instantiation "owns" the time.
    # The roofline entry therefore prices the whole build: algorithmic bytes/flops of all unique quartets / build time
    # (hipEvents on the library's stream around the concurrent launches).  `tiers_alone` lists every instantiation timed
    # by itself; profiles/ holds the rocprofv3 --stats summary with the same kernels.
    return {
        "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
        "kernel": "qc_fock_tier_kernel<LAB, TIER> + qc_fock_bm_kernel<LCD, HI>: %d concurrent launches = one Fock build" % len(tiers), "kernel_ms": tot_ms,
        "kernel_quartets": int(ws.quartets), "kernel_alg_bytes": float(ws.bytes_alg), "kernel_alg_flops": float(ws.flops_alg),
        "note": "f64 gather-compute-scatter on the FP64 ridge (SURVEY 8d): both roofs are given; the binding one is fp64_valu",
        "fp64_valu": {"achieved": tfs, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tfs / FP64_VALU_PEAK_TF},
        "fock_build": {"ms": tot_ms, "sum_tier_kernels_serial_ms": float(tp["unit_ms"].sum()), "launches": len(tiers),
                       "quartets_per_s": float(ws.quartets) / (tot_ms * 1e-3),
                       "slowest_tier_alone": q.hf.unit_name(k),
                       "tiers_alone": tiers, "top_classes_serial": top},
    }


def cpu_baseline(mol, budget_s=20.0):
    """The oracle (CPU restatement of the reference's conventional SCF) on this host, one thread like the reference."""
    from oracle.oracle import Oracle
    o = Oracle(mol)
    nq = o.n_unique_quartets()
    # probe ~1/50 of the quartets to size the sample
    t0 = time.perf_counter(); _, c = o.eri_strided(0, 50); probe = time.perf_counter() - t0
    est_full = probe * nq / max(c, 1)
    stride = max(1, int(est_full / budget_s + 0.999))
    t0 = time.perf_counter(); I, c = o.eri_strided(0, stride); dt = time.perf_counter() - t0
    out = {"value": c / dt, "unit": "shell-quartets/s", "cores": 1, "kind": "port",
           "sample": "every %d-th of the %d unique shell quartets of the same workload (%d quartets, %.2f s), "
                     "oracle/qc_oracle.c orc_eri_full_strided" % (stride, nq, c, dt)}
    if stride == 1 and o.n <= 64:
        # whole reference-style SCF on the CPU: n^4 contraction per iteration (rhf.rs:152-167) on the stored tensor
        t0 = time.perf_counter(); r = o.rhf(100, 1e-10, eri=I); dt2 = time.perf_counter() - t0
        out["scf_iter_ms"] = dt2 * 1e3 / (r["iterations"] + 1)
        out["scf_note"] = "conventional SCF iteration (dense n^4 contraction + Jacobi eigensolve), tensor precomputed"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-reference", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import qchem_rs_amd as q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available() or not q.device_ready():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    key = args.workload if args.workload != "auto" else ("h2o_ccpvtz" if world == 1 else "c6h6_ccpvdz")
    mol = load(q, key)
    sysh = q.System(mol)
    nq_total = sysh.n_quartets()
    if world > 1:
        uid = [q.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        sysh.comm_init(uid[0], rank, world)        # shard the quartet list, RCCL communicator for the partial-Fock all-reduce

    stepper = q.ScfStepper(sysh)
    dt = timed_steps(torch, dist, stepper, args.steps, args.warmup, world)
    D = stepper.density(0)
    tm = stepper.timed
    line = {
        "metric": "ERI shell-quartets/sec through one SCF iteration (ms_per_step = SCF iter time), RHF",
        "value": nq_total * args.steps / dt, "unit": "shell-quartets/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "fixture molecule + basis files under data/ (no randomness); densities are the SCF's own iterates",
        "config": {"workload": WORKLOADS[key][2] + " direct-SCF iteration", "n_basis": sysh.n, "unique_quartets": int(nq_total),
                   "parallelism": "1 GPU" if world == 1 else "quartet shards over %d GPUs + 1 RCCL all-reduce of G per build" % world},
        "iter_breakdown_ms": {"fock_build": tm["fock"], "diis_eig_density": tm["linalg"]},
    }
    rf = roofline(torch, sysh, D, reps=5)      # collective when sharded: every rank calls it
    stepper.close()
    if rank == 0:
        line["roofline"] = rf
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and key == "h2o_ccpvtz":   # HBM bytes per build from committed rocprofv3 --pmc passes (same workload)
            try:
                recs = [v for k2, v in json.load(open(pmc)).items() if k2.startswith("qc_fock_")]
                line["roofline"]["traffic"] = {
                    "hbm_bytes": sum(r["hbm_bytes"] for r in recs), "fetch_bytes_x2": sum(r["fetch_bytes_x2"] for r in recs),
                    "write_bytes": sum(r["write_bytes"] or 0 for r in recs), "atomic_requests": sum(r["atomic_requests"] or 0 for r in recs),
                    "source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, summed over the build's launches)"}
            except Exception:
                pass
    if world == 1 and rank == 0:
        if not args.no_scaling_reference and key == "h2o_ccpvtz":
            m2 = load(q, "c6h6_ccpvdz")
            s2 = q.System(m2)
            st2 = q.ScfStepper(s2)
            k2 = max(3, min(args.steps, 5))
            dt2 = timed_steps(torch, dist, st2, k2, 1, 1)
            tm2 = st2.timed
            st2.close()
            line["scaling_reference"] = {"workload": WORKLOADS["c6h6_ccpvdz"][2] + " direct-SCF iteration", "n_gpus": 1,
                                         "value": s2.n_quartets() * k2 / dt2, "ms_per_step": dt2 * 1e3 / k2, "steps": k2,
                                         "fock_build_ms": tm2["fock"], "unique_quartets": int(s2.n_quartets())}
            if not args.no_cpu_baseline:      # the oracle on a bounded sample of the same 1.1 M quartets (~10 s of one host core)
                line["scaling_reference"]["cpu_baseline"] = cpu_baseline(m2, budget_s=10.0)
            s2.close()
        if key in ("h2o_ccpvtz", "h2o_sto3g", "c6h6_ccpvdz"):
            # the reference's own (conventional) algorithm on the same GPU: tensor resident in HBM, one streaming GEMV per pass
            s3 = q.System(mol); s3.set_fock_mode("stored")
            st3 = q.ScfStepper(s3)
            dt3 = timed_steps(torch, dist, st3, args.steps, args.warmup, 1)
            tm3 = st3.timed; n3 = s3.n
            gemv_bytes = 8.0 * (n3 * (n3 + 1) // 2) * n3 * n3
            fock_ms3 = tm3["fock"]
            line["stored_mode"] = {"ms_per_step": dt3 * 1e3 / args.steps, "tensor_build_ms": st3.tensor_ms(),
                                   "tensor_build_quartets_per_s": nq_total / (st3.tensor_ms() * 1e-3),
                                   "gemv_ms": fock_ms3, "gemv_alg_bytes": gemv_bytes,
                                   "gemv_GBs": gemv_bytes / (fock_ms3 * 1e-3) / 1e9, "gemv_frac_of_8TBs": gemv_bytes / (fock_ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "rhf.rs:45,58-62,152-167 as written: molint::eri once, electron_terms, dense n^4 contraction per pass; "
                                           "the headline value above is the direct-SCF path the north star asks for"}
            st3.close(); s3.close()
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(mol)
    if rank == 0:
        print(json.dumps(line), flush=True)
    sysh.close()                                    # releases the RCCL communicator before torch tears its own down
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
