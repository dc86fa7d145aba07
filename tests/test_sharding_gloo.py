"""CPU, world_size 2 over gloo: the N>1 host path (partition -> shard-local reduce -> all-gather of one
partial per rank -> local modular sum).  Compute is stood in by the oracle here (tests may use it);
on GPUs the same host code runs the HIP kernels and RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deeppowers_amd.params import FheParams, PRIMES_60
from deeppowers_amd.sharding import allgather_partials, shard_bounds
from oracle.cbind import Oracle


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(65536, 8, 3) == (3 * 8192, 4 * 8192)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _params():
    n = 256
    qs = (PRIMES_60[0][0], PRIMES_60[1][0])
    return FheParams(8, qs, tuple(pow(PRIMES_60[i][2], 8192 // n, qs[i]) for i in range(2)))


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _params()
    orc = Oracle.from_params(p)
    a = orc.fill(total * 2, 5).reshape(total, 2, p.n_limbs, p.n)
    b = orc.fill(total * 2, 6).reshape(total, 2, p.n_limbs, p.n)
    lo, hi = shard_bounds(total, world, rank)
    local = orc.ct_mul(np.ascontiguousarray(a[lo:hi]), np.ascontiguousarray(b[lo:hi])) if hi > lo else np.zeros((0, 3, p.n_limbs, p.n), np.uint64)
    partial = orc.reduce_sum(local.ravel(), 3) if hi > lo else np.zeros((3, p.n_limbs, p.n), np.uint64)
    gathered = allgather_partials(torch.from_numpy(partial.view(np.int64)))
    assert gathered.shape == (world, 3, p.n_limbs, p.n)
    assert np.array_equal(gathered[rank].numpy().view(np.uint64), partial)
    total_ct = orc.reduce_sum(gathered.numpy().view(np.uint64).ravel(), 3)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), total_ct)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total", [7, 2])
def test_world2_allgather_reduce_matches_single_process(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    p = _params()
    orc = Oracle.from_params(p)
    a = orc.fill(total * 2, 5).reshape(total, 2, p.n_limbs, p.n)
    b = orc.fill(total * 2, 6).reshape(total, 2, p.n_limbs, p.n)
    want = orc.reduce_sum(orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b)).ravel(), 3)  # world = 1 answer
    got = [np.load(os.path.join(tmp_path, f"r{r}.npy")) for r in range(world)]
    assert np.array_equal(got[0], got[1]), "ranks disagree after the all-gather"
    assert np.array_equal(got[0], want), "result depends on the shard count"


def test_allgather_without_process_group_is_identity():
    t = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    assert torch.equal(allgather_partials(t), t.unsqueeze(0))


def _rendezvous_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeppowers_amd.sharding import NativeComm
    r, w, uid = NativeComm.rendezvous()
    assert (r, w) == (rank, world) and isinstance(uid, bytes) and len(uid) == 128
    with open(os.path.join(out_dir, f"id{rank}"), "wb") as f:
        f.write(uid)
    dist.barrier()
    dist.destroy_process_group()


def test_native_comm_rendezvous_ships_rank0s_rccl_id_at_world2(tmp_path):
    """The id-shipping half of the library's own communicator (dpfhe_comm_unique_id on rank 0 -> 128 bytes to every rank) at world
    size 2: needs no device, so it runs here; dpfhe_comm_create / dpfhe_comm_allgather are exercised on the GPU box."""
    world = 2
    mp.spawn(_rendezvous_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ids = [open(os.path.join(tmp_path, f"id{r}"), "rb").read() for r in range(world)]
    assert ids[0] == ids[1] and len(ids[0]) == 128 and any(ids[0])
