"""Multi-GPU sharding of a ciphertext batch.

One process per GPU.  Ciphertexts of a batch are independent, so the batch is cut into contiguous
ranges, one per rank, with NO communication during compute.  The single exchange step is the one the
north star keeps: each rank reduces its shard's outputs to ONE partial ciphertext (3*L*N words,
384 KiB at N=4096/L=4) and the partials are all-gathered (RCCL over xGMI on GPUs; the reference's only
kept collective, /root/reference/src/core/distributed/distributed_context.cpp:97-122), after which every
rank sums the `world` partials mod q locally.  Payload is latency-bound, not bandwidth-bound.

Two transports for the one collective:
  * torch.distributed (backend "nccl" IS RCCL on ROCm; "gloo" in the CPU tests) - `allgather_partials(t, group)`;
  * the library's own communicator, dpfhe_comm_* (include/dpfhe.h): RCCL called from the C ABI on the caller's HIP
    stream, no torch.distributed on the data path - `NativeComm`.  The 128-byte RCCL id is shipped by whatever
    rendezvous the host program has (here: a torch.distributed broadcast, or a file for the C++ example
    examples/sharded_ct_mul.cpp).

`ShardedMultiplyReduce` is the step bench.py times and tests/test_gpu_parity.py checks at BASELINE configs[3]'s
per-GPU shard: multiply on the main stream, shard-local reduce -> all-gather -> final sum on a side stream,
double-buffered so that the reduce of step i overlaps the multiply of step i+1.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced partition: the first (total % world) ranks get one extra item."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class NativeComm:
    """dpfhe_comm_* (librccl behind the C ABI).  Rank 0 creates the id; `exchange_id` ships it to the other ranks."""

    def __init__(self, rank: int, world: int, device_id: int, unique_id: bytes):
        from . import _cabi
        self._lib = _cabi.load()
        self.rank, self.world, self.device_id = int(rank), int(world), int(device_id)
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        uid = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _cabi.check(self._lib.dpfhe_comm_create(C.byref(h), uid, self.rank, self.world, self.device_id), "dpfhe_comm_create")
        self._h = h

    @staticmethod
    def new_unique_id() -> bytes:
        from . import _cabi
        uid = (C.c_uint8 * 128)()
        _cabi.check(_cabi.load().dpfhe_comm_unique_id(uid), "dpfhe_comm_unique_id")
        return bytes(uid)

    @classmethod
    def rendezvous(cls, group=None) -> tuple[int, int, bytes]:
        """(rank, world, id): rank 0 creates the RCCL id, every rank of the torch.distributed group (any backend) receives its 128
        bytes.  No device needed: tests/test_sharding_gloo.py runs it at world size 2 over gloo."""
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        box = [cls.new_unique_id() if rank == 0 else None]
        if dist.is_initialized() and world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return rank, world, box[0]

    @classmethod
    def from_process_group(cls, device_id: int, group=None) -> "NativeComm":
        """Rendezvous over an existing torch.distributed group (any backend), then dpfhe_comm_create on `device_id`."""
        rank, world, uid = cls.rendezvous(group)
        return cls(rank, world, device_id, uid)

    def allgather(self, partial: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """[...] int64 CUDA tensor -> [world, ...], enqueued on `stream` (default: the current stream of the tensor's device)."""
        from . import _cabi
        if partial.dtype != torch.int64 or not partial.is_cuda or not partial.is_contiguous():
            raise _cabi.DpfheError(2000, "allgather: contiguous int64 CUDA tensor expected")
        s = torch.cuda.current_stream(partial.device) if stream is None else stream
        if out is None:
            with torch.cuda.stream(s):
                out = torch.empty((self.world,) + tuple(partial.shape), dtype=partial.dtype, device=partial.device)
        _cabi.check(self._lib.dpfhe_comm_allgather(self._h, out.data_ptr(), partial.data_ptr(), partial.numel(), s.cuda_stream), "dpfhe_comm_allgather")
        return out

    def allreduce_sum(self, ctx, partial: torch.Tensor, stream=None) -> torch.Tensor:
        """in place: every rank's `partial` ([..., L, N] canonical residues of `ctx`) becomes the sum of all ranks' partials mod q - ncclAllReduce(u64, sum)
        + one mod-q pass (dpfhe_comm_allreduce_sum; world <= 15).  The words all-gather + local sum give."""
        from . import _cabi
        if partial.dtype != torch.int64 or not partial.is_cuda or not partial.is_contiguous():
            raise _cabi.DpfheError(2000, "allreduce_sum: contiguous int64 CUDA tensor expected")
        s = torch.cuda.current_stream(partial.device) if stream is None else stream
        n_rns = partial.numel() // (ctx.params.n_limbs * ctx.params.n)
        _cabi.check(self._lib.dpfhe_comm_allreduce_sum(self._h, ctx.handle, partial.data_ptr(), n_rns, s.cuda_stream), "dpfhe_comm_allreduce_sum")
        return partial

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dpfhe_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_partials(partial: torch.Tensor, group=None, comm: NativeComm | None = None, stream=None) -> torch.Tensor:
    """[...] -> [world, ...]; rank r's partial lands at index r on every rank.  `comm` selects the native transport."""
    if comm is not None:
        return comm.allgather(partial, stream=stream)
    if not dist.is_initialized():
        return partial.unsqueeze(0)
    world = dist.get_world_size(group)
    flat = partial.contiguous().view(-1)  # flat concatenation: the one layout every backend accepts
    out = torch.empty(world * flat.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.view((world,) + tuple(partial.shape))


def sharded_multiply_reduce(ev, a, b, group=None, stream=None, comm: NativeComm | None = None):
    """Encrypted-logits style reduction: sum_i a_i (x) b_i over the GLOBAL batch, given this rank's shard.

    a, b: Ciphertext shards (leading batch dim).  Returns (local_outputs, global_sum) where global_sum is
    a 3-component Ciphertext identical on every rank.  Everything is enqueued on one stream (see
    ShardedMultiplyReduce for the overlapped, buffer-reusing form)."""
    from .evaluator import Ciphertext

    local = ev.multiply(a, b, stream=stream)
    partial = ev.reduce_sum(local, stream=stream)
    with ev._on(stream):
        gathered = allgather_partials(partial.data, group, comm, stream)
    total = ev.reduce_sum(Ciphertext(gathered, local.is_ntt), stream=stream)
    return local, total


class ShardedMultiplyReduce:
    """The BASELINE configs[3] step for one rank: multiply its shard, reduce to one partial, all-gather, sum.

    Two output buffers and two HIP streams: the (VALU-bound) multiply of step i+1 runs on `main` while the (HBM-bound)
    shard-local reduce + all-gather + final sum of step i run on `side`.  No allocation after construction."""

    def __init__(self, ev, batch: int, group=None, comm: NativeComm | None = None, main=None, collective: str = "allgather", outs=None, sum_in_ntt_domain: bool = False):
        """collective: "allgather" (one partial per rank gathered, summed locally - the north star's exchange) or "allreduce" (SURVEY.md 8(e)'s alternative:
        a 64-bit sum all-reduce of the partials in place + one mod-q pass; world <= 15).  Same words either way.
        sum_in_ntt_domain: for a caller who needs ONLY the sum - the inverse transform is linear, so the products are left in the NTT domain
        (dpfhe_ct_mul with DPFHE_OUT_NTT: 4 forward transforms + the tensor step per pair), summed there, and the total alone is transformed back: the
        same total, 3 of 7 transforms per pair never run.  NOT the metric op (whose products are coefficient-domain ciphertexts): outs[k] then holds
        NTT-domain products."""
        if collective not in ("allgather", "allreduce"):
            raise ValueError("collective must be 'allgather' or 'allreduce'")
        self.ev, self.group, self.comm, self.collective, self.sum_in_ntt_domain = ev, group, comm, collective, bool(sum_in_ntt_domain)
        ctx = ev.ctx
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        p = ctx.params
        self.outs = list(outs) if outs is not None else [ctx.empty(batch, components=3) for _ in range(2)]   # (outs: two caller buffers [batch][3][L][N] to reuse)
        self.partials = [ctx.empty(components=3) for _ in range(2)]
        self.gathered = [torch.empty((self.world, 3, p.n_limbs, p.n), dtype=torch.int64, device=ctx.device) for _ in range(2)]
        self.totals = [ctx.empty(components=3) for _ in range(2)]
        self.main = torch.cuda.current_stream(ctx.device) if main is None else main
        self.side = torch.cuda.Stream(device=ctx.device)
        self.mul_done = [torch.cuda.Event() for _ in range(2)]
        self.red_done = [torch.cuda.Event() for _ in range(2)]
        self.gather_events = None   # optional (start, end) timing events of the collective, set by the caller
        self._count = 0

    def step(self, a, b, timing=None) -> int:
        """Enqueues one step; returns the buffer index k: outs[k] / totals[k] hold its results once `red_done[k]` fires.
        timing: optional (start_event, end_event) recorded around the multiply on the main stream."""
        from .evaluator import Ciphertext
        ev, k = self.ev, self._count & 1
        self._count += 1
        self.main.wait_event(self.red_done[k])              # the reduce that read outs[k] two steps ago has finished
        if timing is not None:
            timing[0].record(self.main)
        c = ev.multiply(a, b, out=self.outs[k], out_ntt=True if self.sum_in_ntt_domain else None, stream=self.main)
        if timing is not None:
            timing[1].record(self.main)
        self.mul_done[k].record(self.main)
        self.side.wait_event(self.mul_done[k])
        with torch.cuda.stream(self.side):
            part = ev.reduce_sum(c, out=self.partials[k], stream=self.side)
            if self.gather_events is not None:
                self.gather_events[0].record(self.side)
            if self.collective == "allreduce" and self.world > 1:
                # the lazy sum of the ranks' partials (each < q < 2^60, world <= 15: no 64-bit wrap), then one pass to canonical residues
                self.totals[k].copy_(part.data)
                if self.comm is not None:
                    self.comm.allreduce_sum(ev.ctx, self.totals[k], stream=self.side)
                else:
                    dist.all_reduce(self.totals[k], op=dist.ReduceOp.SUM, group=self.group)
                    ev.canonicalize_sum_(self.totals[k], stream=self.side)
                if self.gather_events is not None:
                    self.gather_events[1].record(self.side)
                if self.sum_in_ntt_domain:
                    ev.ntt_inverse_(self.totals[k], stream=self.side)
                self.red_done[k].record(self.side)
                return k
            if self.comm is not None:
                g = self.comm.allgather(part.data, out=self.gathered[k], stream=self.side)
            elif dist.is_initialized():
                dist.all_gather_into_tensor(self.gathered[k].view(-1), part.data.view(-1), group=self.group)
                g = self.gathered[k]
            else:
                g = part.data.unsqueeze(0)
            if self.gather_events is not None:
                self.gather_events[1].record(self.side)
            ev.reduce_sum(Ciphertext(g, c.is_ntt), out=self.totals[k], stream=self.side)
            if self.sum_in_ntt_domain:
                ev.ntt_inverse_(self.totals[k], stream=self.side)   # the ONE inverse transform of the step: 3 L residue polynomials
        self.red_done[k].record(self.side)
        return k
