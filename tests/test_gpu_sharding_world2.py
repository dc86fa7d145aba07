"""GPU, world size 2 as two PROCESSES on the box's one GPU: compute on the HIP kernels (each rank multiplies and reduces ITS shard),
gather of one partial per rank over torch.distributed (gloo through host memory - RCCL refuses two ranks on one device), final sum on
the HIP kernels again.  Checks what the gloo-only CPU test cannot: with real kernels on both sides of the collective the global sum is
identical on both ranks, independent of the shard count, and equal to the oracle's world-size-1 result (reference collective:
/root/reference/src/core/distributed/distributed_context.cpp:97-122)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deeppowers_amd.params import FheParams
from oracle.cbind import Oracle

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    from deeppowers_amd.sharding import allgather_partials, shard_bounds
    p = FheParams.n4096_l4()
    orc = Oracle.from_params(p)
    a = orc.fill(total * 2, 15).reshape(total, 2, p.n_limbs, p.n)      # the GLOBAL batch, a function of the seed only
    b = orc.fill(total * 2, 16).reshape(total, 2, p.n_limbs, p.n)
    lo, hi = shard_bounds(total, world, rank)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    local = ev.multiply(Ciphertext(to_device(np.ascontiguousarray(a[lo:hi]), ctx.device)), Ciphertext(to_device(np.ascontiguousarray(b[lo:hi]), ctx.device)))
    partial = ev.reduce_sum(local)                                                       # shard-local: one partial ciphertext
    torch.cuda.synchronize()
    gathered = allgather_partials(partial.data.cpu())                                    # the one collective (gloo, host memory)
    assert gathered.shape[0] == world and torch.equal(gathered[rank], partial.data.cpu())
    total_ct = ev.reduce_sum(Ciphertext(gathered.to(ctx.device)))
    torch.cuda.synchronize()
    # SURVEY.md 8(e)'s alternative exchange: 64-bit sum all-reduce of the partials (gloo, host memory) + ONE mod-q pass on the device - the same words
    lazy = partial.data.cpu().clone()
    dist.all_reduce(lazy, op=dist.ReduceOp.SUM)
    ar = ev.canonicalize_sum_(lazy.to(ctx.device))
    torch.cuda.synchronize()
    assert torch.equal(ar, total_ct.data), "all-reduce + mod-q pass differs from all-gather + local sum"
    np.save(os.path.join(out_dir, f"r{rank}.npy"), to_host(total_ct.data))
    np.save(os.path.join(out_dir, f"p{rank}.npy"), to_host(partial.data))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 2])
def test_world2_hip_compute_gather_hip_sum_matches_world1(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    p = FheParams.n4096_l4()
    orc = Oracle.from_params(p)
    a = orc.fill(total * 2, 15).reshape(total, 2, p.n_limbs, p.n)
    b = orc.fill(total * 2, 16).reshape(total, 2, p.n_limbs, p.n)
    want = orc.reduce_sum(orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b), threads=0).ravel(), 3)   # world-size-1 answer
    got = [np.load(os.path.join(tmp_path, f"r{r}.npy")) for r in range(world)]
    assert np.array_equal(got[0], got[1]), "ranks disagree after the all-gather"
    assert np.array_equal(got[0].reshape(want.shape), want), "the sharded sum differs from the world-size-1 result"
    parts = [np.load(os.path.join(tmp_path, f"p{r}.npy")) for r in range(world)]
    assert not np.array_equal(parts[0], parts[1])       # the ranks really worked on different shards
