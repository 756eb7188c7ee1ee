"""Row sharding over the GPUs of one node and the rendezvous around it.

The estimators are sums over the draw index r of products of values that share
r (varsens/saltelli.py:577,591-596,612-619), so the n draws split into
contiguous row ranges, one per rank (one process per GPU).  Each rank
generates, evaluates and reduces its own rows; the only exchange is ONE sum
all-reduce of the accumulator record over RCCL (vs_comm_allreduce).  The
reference has no counterpart: its scale-out is PBS array jobs over exported
text files (varsens/cluster/accre-job.sh:13).

Process-group plumbing (who am I, shipping the 128-byte RCCL id, barriers and
max-over-ranks timing) rides on torch.distributed's CPU ``gloo`` group when
the job was started by ``torch.distributed.run``; nothing on the data path
touches torch.
"""
import os
import sys

import numpy


def shard_rows(n, world, rank):
    """Contiguous rows [begin, begin+count) of rank `rank`; counts differ by at most 2 and all
    but the last are even (the streaming reduction reads double2 pairs)."""
    n, world, rank = int(n), int(world), int(rank)
    if not (0 <= rank < world):
        raise Exception("rank %d not in [0,%d)" % (rank, world))
    pairs = n // 2
    base, extra = divmod(pairs, world)
    begin = 2 * (rank * base + min(rank, extra))
    count = 2 * (base + (1 if rank < extra else 0))
    if rank == world - 1:
        count = n - begin
    return begin, count


class HostGroup(object):
    """The CPU side of a multi-process job (rank/world from the launcher's environment)."""

    def __init__(self, backend="gloo"):
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if not dist.is_initialized():
                # gloo announces its connections on the C-level stdout; a bench line must stay alone there
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
                    dist.barrier()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        """Rank `src` passes `payload` (bytes); every rank returns the same bytes."""
        if self._dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def max(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def gather_objects(self, obj):
        """Every rank's `obj` (picklable), in rank order, on every rank."""
        if self._dist is None:
            return [obj]
        box = [None] * self.world
        self._dist.all_gather_object(box, obj)
        return box

    def sum_array(self, array):
        """Sum a host float64 array over ranks (host-side checks and CPU tests; the GPU data
        path uses RcclComm.allreduce on device memory instead)."""
        a = numpy.ascontiguousarray(array, dtype=numpy.float64)
        if self._dist is None:
            return a.copy()
        import torch
        t = torch.from_numpy(a.copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
        self._dist = None


class RcclComm(object):
    """``comm`` object for Varsens: world, rank and allreduce(device_buffer, count)."""

    def __init__(self, ctx, group):
        self.ctx = ctx
        self.group = group
        self.world = group.world
        self.rank = group.rank
        if self.world > 1:
            uid = ctx.comm_unique_id() if group.rank == 0 else None
            uid = group.broadcast_bytes(uid, src=0)
            ctx.comm_init(uid, self.world, self.rank)
            # RCCL builds its rings/connections on the first collective: pay that here, not in a timed pass
            warm = ctx.upload(numpy.zeros(8))
            ctx.comm_allreduce(warm, 8)
            ctx.sync()
            warm.free()

    def allreduce(self, buf, count):
        if self.world > 1:
            self.ctx.comm_allreduce(buf, count)

    def max(self, value):
        """Largest `value` over the ranks (host side; planning decisions that must agree on every rank)."""
        return self.group.max(value)

    def close(self):
        if self.world > 1:
            self.ctx.comm_destroy()


class HostExchange(object):
    """Same interface as RcclComm, but the record crosses the host: download, gloo sum, upload.
    Only a visible, flagged fallback for bench.py when RCCL cannot initialise; never chosen silently."""

    def __init__(self, ctx, group):
        self.ctx, self.group = ctx, group
        self.world, self.rank = group.world, group.rank

    def allreduce(self, buf, count):
        if self.world > 1:
            self.ctx.sync()
            total = self.group.sum_array(self.ctx.download(buf, (int(count),)))
            self.ctx.upload_into(buf, total)

    def max(self, value):
        return self.group.max(value)

    def close(self):
        pass
