"""Row sharding over the GPUs of one node and the rendezvous around it.

The estimators are sums over the draw index r of products of values that share
r (varsens/saltelli.py:577,591-596,612-619), so the n draws split into
contiguous row ranges, one per rank (one process per GPU).  Each rank
generates, evaluates and reduces its own rows; the only exchange is ONE sum
all-reduce of the accumulator record over RCCL (vs_comm_allreduce).  The
reference has no counterpart: its scale-out is PBS array jobs over exported
text files (varsens/cluster/accre-job.sh:13, accre-submit.sh:27).

Process-group plumbing (who am I, shipping the 128-byte RCCL id, barriers and
max-over-ranks timing) needs no torch: ``TcpRendezvous`` is a small store on
MASTER_ADDR:MASTER_PORT kept by rank 0 (every collective is one all-gather of
pickled values through it).  A job started by ``torch.distributed.run`` finds
that port taken by the launcher's own store, so there the same plumbing rides
on torch.distributed's CPU ``gloo`` group.  ``VARSENS_RENDEZVOUS=tcp|gloo``
overrides the choice (tcp under torchrun listens on MASTER_PORT+1 unless
``VARSENS_RENDEZVOUS_PORT`` names the port; ``VARSENS_RENDEZVOUS_FD`` on rank 0
is a listening socket the launcher bound and passed down, see bench.py).
Nothing on the data path touches either.
"""
import os
import pickle
import socket
import struct
import sys
import threading
import time

import numpy


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """Libraries that announce themselves on the C-level stdout (gloo's connection notes, RCCL's version banner at
    communicator creation) must not reach a stdout that carries one JSON line: file descriptor 1 points at stderr
    for the duration."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


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


def _send_msg(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, count):
    chunks = []
    while count:
        piece = sock.recv(min(count, 1 << 20))
        if not piece:
            raise EOFError("rendezvous peer closed the connection")
        chunks.append(piece)
        count -= len(piece)
    return b"".join(chunks)


def _recv_msg(sock):
    (size,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, size))


class TcpRendezvous(object):
    """All-gather of small picklable values over one TCP connection per rank to a store thread on
    rank 0.  Every rank must take part in every round, in the same order (as with any collective)."""

    def __init__(self, rank, world, addr, port, timeout=300.0, listener=None):
        """`listener`: a socket already bound to (addr, port) and listening, handed to rank 0 by whoever
        started the ranks (bench.py's parent binds port 0 and passes the descriptor on), so that the port
        is never free between its choice and its use."""
        self.rank, self.world = rank, world
        self._server = None
        if rank == 0:
            if listener is None:
                listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                listener.bind((addr, port))
                listener.listen(world)
            listener.settimeout(timeout)
            self._server = threading.Thread(target=self._serve, args=(listener, timeout), daemon=True)
            self._server.start()
        deadline = time.time() + timeout
        while True:
            try:
                self._sock = socket.create_connection((addr, port), timeout=5.0)
                break
            except OSError:
                if time.time() > deadline:
                    raise Exception("rank %d: no rendezvous store at %s:%d after %.0f s" % (rank, addr, port, timeout))
                time.sleep(0.05)
        self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._sock.settimeout(timeout)
        _send_msg(self._sock, rank)

    def _serve(self, listener, timeout):
        peers = [None] * self.world
        try:
            for _ in range(self.world):
                conn, _ = listener.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(None)
                r = _recv_msg(conn)
                if not (isinstance(r, int) and 0 <= r < self.world and peers[r] is None):
                    raise Exception("rendezvous: unexpected rank announcement %r" % (r,))
                peers[r] = conn
            while True:
                # a rank that leaves sends no further round: EOFError ends the store
                values = [_recv_msg(c) for c in peers]
                for c in peers:
                    _send_msg(c, values)
        except (EOFError, OSError):
            pass
        finally:
            listener.close()
            for c in peers:
                if c is not None:
                    c.close()

    def all_gather(self, value):
        _send_msg(self._sock, value)
        return _recv_msg(self._sock)

    def close(self):
        try:
            self.all_gather(None)       # nobody leaves while a peer still waits for a round
        except Exception:
            pass
        self._sock.close()
        if self._server is not None:
            self._server.join(timeout=5.0)


def _rendezvous_backend():
    """'tcp' (no torch) unless torch.distributed.run started the job, whose own store sits on MASTER_PORT."""
    choice = os.environ.get("VARSENS_RENDEZVOUS", "").strip().lower()
    if choice in ("tcp", "gloo"):
        return choice
    if choice:
        raise Exception("VARSENS_RENDEZVOUS=%r: expected 'tcp' or 'gloo'" % choice)
    return "gloo" if "TORCHELASTIC_RUN_ID" in os.environ else "tcp"


class HostGroup(object):
    """The CPU side of a multi-process job (rank/world from the launcher's environment)."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        self.backend = None
        self._dist = None
        self._tcp = None
        if self.world > 1:
            self.backend = backend or _rendezvous_backend()
            addr = os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            port = int(os.environ.setdefault("MASTER_PORT", "29500"))
            if self.backend == "tcp":
                if "VARSENS_RENDEZVOUS_PORT" in os.environ:
                    port = int(os.environ["VARSENS_RENDEZVOUS_PORT"])
                elif "TORCHELASTIC_RUN_ID" in os.environ:
                    port += 1
                listener = None
                if self.rank == 0 and "VARSENS_RENDEZVOUS_FD" in os.environ:
                    # the launcher's bound, listening socket (inherited descriptor): take it over
                    listener = socket.socket(fileno=int(os.environ.pop("VARSENS_RENDEZVOUS_FD")))
                    port = listener.getsockname()[1]
                self._tcp = TcpRendezvous(self.rank, self.world, addr, port, listener=listener)
                self._tcp.all_gather(None)
            elif self.backend == "gloo":
                self._init_gloo()
            else:
                raise Exception("unknown rendezvous back end %r" % (self.backend,))

    def _init_gloo(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            with stdout_to_stderr():   # gloo announces its connections on the C-level stdout
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
                dist.barrier()
        self._dist = dist

    def barrier(self):
        if self._tcp is not None:
            self._tcp.all_gather(None)
        elif self._dist is not None:
            self._dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        """Rank `src` passes `payload` (bytes); every rank returns the same bytes."""
        if self._tcp is not None:
            return self._tcp.all_gather(payload if self.rank == src else None)[src]
        if self._dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def max(self, value):
        if self._tcp is not None:
            return max(self._tcp.all_gather(float(value)))
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def gather_objects(self, obj):
        """Every rank's `obj` (picklable), in rank order, on every rank."""
        if self._tcp is not None:
            return self._tcp.all_gather(obj)
        if self._dist is None:
            return [obj]
        box = [None] * self.world
        self._dist.all_gather_object(box, obj)
        return box

    def sum_array(self, array):
        """Sum a host float64 array over ranks (host-side checks and CPU tests; the GPU data
        path uses RcclComm.allreduce on device memory instead)."""
        a = numpy.ascontiguousarray(array, dtype=numpy.float64)
        if self._tcp is not None:
            total = numpy.zeros_like(a)
            for part in self._tcp.all_gather(a):     # rank order on every rank: the same bits everywhere
                total += part
            return total
        if self._dist is None:
            return a.copy()
        import torch
        t = torch.from_numpy(a.copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        if self._tcp is not None:
            self._tcp.close()
            self._tcp = None
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
        self._dist = None


def device_for_rank(local_rank, shared_device=False):
    """The device ordinal a rank opens: its LOCAL_RANK where the process sees every GPU of the node (torchrun, bench.py's
    own launcher), ordinal LOCAL_RANK mod visible-count where the launcher has narrowed each rank's view to its own
    card(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank: every rank then sees ONE device, number 0).  Two ranks
    that end up on one card are caught where it matters: RCCL refuses duplicate devices."""
    from . import _native
    if shared_device:
        return 0
    visible = _native.device_count()
    if visible < 1:
        raise Exception("rank with LOCAL_RANK=%d sees no GPU" % local_rank)
    return local_rank % visible


def exchange_check(group, before, after):
    """Does an all-reduce result hold the sum over ALL ranks?  `before`: this rank's record as it
    went into the exchange, `after`: what came out (host float64 arrays of equal length on every
    rank).  Every rank's `before` crosses the rendezvous and is summed in rank order on the host;
    the report says how many ranks the record counts (entry [2], 1 per rank going in), how far the
    exchanged record is from that host sum -- relative to the sum of magnitudes, so that a
    cancelling entry cannot hide a missing addend -- and whether every rank holds the same bits.
    A collective: every rank calls it and gets the same report."""
    before = numpy.ascontiguousarray(before, dtype=numpy.float64)
    after = numpy.ascontiguousarray(after, dtype=numpy.float64)
    parts = group.gather_objects(before)
    outs = group.gather_objects(after)
    want = numpy.zeros_like(parts[0])
    size = numpy.zeros_like(parts[0])
    for part in parts:
        want += part
        size += numpy.abs(part)
    diff = numpy.abs(outs[0] - want)
    rel = float(numpy.max(numpy.where(size > 0.0, diff / numpy.where(size > 0.0, size, 1.0), diff))) if want.size else 0.0
    same = all(numpy.array_equal(o, outs[0]) for o in outs[1:])
    ranks = float(outs[0][2]) if want.size > 2 else float("nan")
    return {"ranks_in_record": ranks, "ranks_expected": group.world, "max_rel_diff_vs_host_sum": rel,
            "identical_on_all_ranks": bool(same), "doubles": int(want.size),
            # `identical_on_all_ranks` is reported, not required: a collective library may in principle sum in another
            # order on another rank; what must hold is that every rank's record is in the sum
            "ok": bool(ranks == group.world and rel <= 1e-12 and
                       all(float(numpy.max(numpy.abs(o - outs[0]))) <= 1e-12 * max(1.0, float(numpy.max(size)))
                           for o in outs[1:]))}


class RcclComm(object):
    """``comm`` object for Varsens: world, rank and allreduce(device_buffer, count)."""

    def __init__(self, ctx, group):
        self.ctx = ctx
        self.group = group
        self.world = group.world
        self.rank = group.rank
        if self.world > 1:
            # RCCL prints a version banner on the C-level stdout when the library or a communicator comes up
            # (seen on the GPU boxes: "RCCL version : ..." five lines): keep it off a stdout that carries one JSON line
            with stdout_to_stderr():
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
