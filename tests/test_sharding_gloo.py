"""world_size-2 `gloo` test of the multi-GPU data path on CPU: each rank takes its shard of the unique shell quartets
from the product's plan (qc_plan_shard_quartets, host-only), the oracle digests exactly those quartets, and the partial
Fock matrices are summed with all_reduce - the collective the GPU path issues through RCCL (qc_fock_build_device).
The sum must equal the reference's dense contraction (rhf.rs:152-167) of the full tensor."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    import qchem_rs_amd as q
    from conftest import load_system
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = load_system("water", "6-31G_st_st")
    s, o = q.System(m), Oracle(m)
    D = np.random.default_rng(11).standard_normal((s.n, s.n)); D = D + D.T          # same seed on every rank
    shard = s.plan_shard_quartets(rank, world)
    G = torch.from_numpy(o.g_rhf_quartets(D, shard))
    dist.all_reduce(G, op=dist.ReduceOp.SUM)                                          # partial Fock -> full
    counts = torch.tensor([len(shard)])
    dist.all_reduce(counts)
    if rank == 0:
        G_ref = o.g_rhf(D, o.eri())
        out.put((float(np.abs(G.numpy() - G_ref).max()), int(counts.item()), int(s.n_quartets())))
    dist.barrier()
    dist.destroy_process_group()


def test_partial_fock_allreduce_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    err, counted, total = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert counted == total
    assert err < 1e-11
