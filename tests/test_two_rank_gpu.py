"""Two-rank GPU test of the multi-rank SCF path: the integer all-reduce of the hi / lo planes, the max all-reduce of the
pass scalars with their complements and the run-time binding of RCCL have only ever run with a 1-rank communicator on the one-GPU boxes
this repository is developed on.  With two visible devices this test runs an RHF and a closed-shell UHF SCF on two ranks (one process per
GPU, started by a parent that has made no GPU call - torch.cuda.device_count() does not initialise the runtime on this image) and asserts
that both ranks end in the bit-identical state (integer all-reduce + replicated deterministic linear algebra) and that this state is
the single-GPU run's to 1e-10 Eh with the same pass count (across shard layouts the partial sums agree to ~1e-13, not bit for bit:
include/qchem_hip.h, qc_set_accumulation).  Skipped when fewer than two devices are visible."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import qchem_rs_amd as q
from conftest import load_system
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("gloo")
box = [q.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
out = {{}}
# (QC_EIG_WARM_RMS in the environment of the second invocation: the perturbative refinement starts while the density still moves, asks for
# rotations, and the pass takes its repeat branch - the one place where a pass issues a second collective, scf_iterate)
redo = os.environ.get("QC_EIG_WARM_RMS") is not None
for mol, basis, uhf in ((("ethylene", "cc-pVDZ", False),) if redo else (("water", "cc-pVDZ", False), ("water", "STO-3G", True), ("benzene", "STO-3G", False))):
    m = load_system(mol, basis)
    ref = q.System(m)                                    # this rank alone: the whole quartet list
    run = q.unrestricted_hartree_fock if uhf else q.restricted_hartree_fock
    r1 = run(ref, q.HartreeFockConfig(100, 1e-9))
    s = q.System(m)
    s.comm_init(box[0], rank, world)
    r2 = run(s, q.HartreeFockConfig(100, 1e-9))
    if redo:
        st = q.ScfStepper(s)
        for k in range(12):
            st.iterate()
        c = st.counters(); st.close()
        assert c["redos"] >= 1, c
    out["%s/%s/%s" % (mol, basis, "uhf" if uhf else "rhf")] = dict(
        same_energy=abs(r1.electronic_energy - r2.electronic_energy) < 1e-10, same_iterations=r1.iterations == r2.iterations,
        same_orbitals=max(abs(x - y) for x, y in zip(r1.orbital_energies, r2.orbital_energies)) < 1e-9,
        e=r2.electronic_energy.hex(), w0=float(r2.orbital_energies[0]).hex())
    s.close(); ref.close()
    dist.barrier()
print("RANK%d " % rank + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_two_ranks_agree_bitwise_and_reproduce_the_single_gpu_scf(tmp_path):
    if _ndev() < 2:
        pytest.skip("needs two visible GPUs (one-GPU leases: the 1-rank communicator test covers what can run here)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for redo in (False, True):
        if redo:
            env["QC_EIG_WARM_RMS"] = "0.5"                   # the repeat branch of scf_iterate (second collective of a pass) on two ranks
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        seen = {}
        for ln in p.stdout.splitlines():
            if ln.startswith("RANK"):
                seen[int(ln[4])] = json.loads(ln.split(" ", 1)[1])
        assert sorted(seen) == [0, 1]
        for rank, res in seen.items():
            for key, r in res.items():
                # (with the repeat branch forced, single-GPU and two-rank runs take other eigensolvers per pass: same fixed point, the
                # pass count may differ by one)
                assert r["same_energy"] and r["same_orbitals"] and (redo or r["same_iterations"]), (rank, key, r)
        assert seen[0] == seen[1]                            # every rank holds the same state, bit for bit (hex digits compared)
