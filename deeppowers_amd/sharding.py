"""Multi-GPU sharding of a ciphertext batch.

One process per GPU.  Ciphertexts of a batch are independent, so the batch is cut into contiguous
ranges, one per rank, with NO communication during compute.  The single exchange step is the one the
north star keeps: each rank reduces its shard's outputs to ONE partial ciphertext (3*L*N words,
384 KiB at N=4096/L=4) and the partials are all-gathered (RCCL over xGMI on GPUs; the reference's only
kept collective, /root/reference/src/core/distributed/distributed_context.cpp:97-122), after which every
rank sums the `world` partials mod q locally.  Payload is latency-bound, not bandwidth-bound.

The collective goes through torch.distributed (backend "nccl" IS RCCL on ROCm; "gloo" in the CPU
tests); libdpfhe_hip.so also exports the same step natively (dpfhe_comm_allgather).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced partition: the first (total % world) ranks get one extra item."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_partials(partial: torch.Tensor, group=None) -> torch.Tensor:
    """[...] -> [world, ...]; rank r's partial lands at index r on every rank."""
    if not dist.is_initialized():
        return partial.unsqueeze(0)
    world = dist.get_world_size(group)
    flat = partial.contiguous().view(-1)  # flat concatenation: the one layout every backend accepts
    out = torch.empty(world * flat.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.view((world,) + tuple(partial.shape))


def sharded_multiply_reduce(ev, a, b, group=None, stream=None):
    """Encrypted-logits style reduction: sum_i a_i (x) b_i over the GLOBAL batch, given this rank's shard.

    a, b: Ciphertext shards (leading batch dim).  Returns (local_outputs, global_sum) where global_sum is
    a 3-component Ciphertext identical on every rank."""
    from .evaluator import Ciphertext

    local = ev.multiply(a, b, stream=stream)
    partial = ev.reduce_sum(local, stream=stream)
    gathered = allgather_partials(partial.data, group)
    total = ev.reduce_sum(Ciphertext(gathered, local.is_ntt), stream=stream)
    return local, total
