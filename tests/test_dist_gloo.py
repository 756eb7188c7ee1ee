"""N>1 host path on CPU: two processes, world_size 2, over both rendezvous back ends
(the torch-free TCP store and torch.distributed gloo).

Covers what runs on the host in a multi-GPU job: the launcher environment -> HostGroup,
shipping the 128-byte communicator id, barrier / max-over-ranks timing, the row split, and
the accumulator-record contract (each rank fills its own variance slot, the SUM over ranks
finishes to the single-pass result).  The per-rank records are built by the oracle here
(test-only): the kernels that build them on a GPU are covered by the -m gpu tests.
"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from varsens_amd import dist, _native
    from oracle import saltelli_oracle as orc

    g = dist.HostGroup()
    assert g.world == 2 and g.backend == os.environ["VARSENS_RENDEZVOUS"]
    if g.backend == "tcp":
        assert "torch" not in sys.modules, "the tcp rendezvous must not import torch"
    assert [o["r"] for o in g.gather_objects({"r": g.rank})] == [0, 1]
    uid = bytes(range(128)) if g.rank == 0 else None
    uid = g.broadcast_bytes(uid, src=0)
    assert uid == bytes(range(128))
    g.barrier()
    assert g.max(10.0 + g.rank) == 11.0

    k, n, l = 5, 1001, 2
    rng = np.random.RandomState(7)
    blocks = 3.0 + rng.randn(2 * k + 2, n, l)
    blocks[:, :400] += 2.0
    begin, count = dist.shard_rows(n, g.world, g.rank)
    mine = blocks[:, begin:begin + count]
    ex = orc.exact_accumulators(mine[0], mine[1], mine[2:2 + k], mine[2 + k:])

    # lay the shard's sums out as the engine's accumulator record (include/varsens_hip.h)
    per, slot = _native.acc_per_output(k), _native.acc_slot(l)
    rec = np.zeros(_native.acc_len(k, l, g.world))
    rec[0], rec[2] = count, 1.0
    for o in range(l):
        body = rec[4 + o * per: 4 + (o + 1) * per]
        body[0], body[1], body[2] = ex["sab"][o], ex["sa"][o], ex["sb"][o]
        for q, name in enumerate(("ac", "bd", "ad", "bc")):
            body[3 + q * k: 3 + (q + 1) * k] = ex[name][:, o]
    s = rec[4 + l * per + g.rank * slot:][:slot]
    s[0] = 2 * count
    s[1:1 + l] = ex["mean"]
    s[1 + l:] = ex["m2"]

    total = g.sum_array(rec)
    assert total[0] == n and total[2] == 2

    # the check bench.py runs before it times anything (dist.exchange_check): a true sum passes; an exchange that
    # leaves a rank out, one that returns different bits per rank and one that is off in a single entry do not
    good = dist.exchange_check(g, rec, total)
    assert good["ok"] and good["ranks_in_record"] == 2 and good["max_rel_diff_vs_host_sum"] <= 1e-15
    assert good["identical_on_all_ranks"] and good["doubles"] == rec.size
    alone = dist.exchange_check(g, rec, rec)                 # "all-reduce" that did nothing: one rank in the record
    assert not alone["ok"] and alone["ranks_in_record"] == 1
    skew = total.copy()
    skew[5] *= 1.0 + 1e-9 * g.rank                           # rank 1 ends with other bits than rank 0
    assert not dist.exchange_check(g, rec, skew)["identical_on_all_ranks"]
    off = total.copy()
    off[7] += 1e-6 * abs(off[7]) + 1e-9
    bad = dist.exchange_check(g, rec, off)
    assert not bad["ok"] and bad["max_rel_diff_vs_host_sum"] > 1e-8 and bad["ranks_in_record"] == 2

    # finish exactly as finish_first_kernel does (pooled variance over the rank slots)
    slots = total[4 + l * per:].reshape(g.world, slot)
    cnt = slots[:, 0].sum()
    mu = (slots[:, :1] * slots[:, 1:1 + l]).sum(axis=0) / cnt
    m2 = (slots[:, 1 + l:] + slots[:, :1] * (slots[:, 1:1 + l] - mu) ** 2).sum(axis=0)
    var_y = m2 / (cnt - 1)
    whole = orc.compute_varsens(blocks[0], blocks[1], blocks[2:2 + k], blocks[2 + k:], n, second_order=False)
    np.testing.assert_allclose(var_y, whole["var_y"], rtol=1e-12)
    body = total[4:4 + l * per].reshape(l, per)
    E_2 = body[:, 0] / n
    U_j = (body[:, 3:3 + k] / (n - 1) + body[:, 3 + k:3 + 2 * k] / (n - 1)) / 2
    np.testing.assert_allclose(E_2, whole["E_2"], rtol=1e-13)
    np.testing.assert_allclose(U_j.T, whole["U_j"], rtol=1e-13)
    sens = (U_j.T - E_2) / var_y
    np.testing.assert_allclose(sens, whole["sens"], rtol=1e-9, atol=1e-12)
    g.close()
    print("rank", g.rank, "ok")
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("backend", ["tcp", "gloo"])
def test_two_rank_host_path(tmp_path, backend):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VARSENS_RENDEZVOUS=backend)
        env.pop("TORCHELASTIC_RUN_ID", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out


FAILING = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    from varsens_amd import dist
    g = dist.HostGroup()
    g.barrier()
    if g.rank == 1:
        os._exit(7)                      # dies without a word, in the middle of the job
    t0 = time.time()
    try:
        g.max(1.0)                       # a collective the dead rank never joins
    except Exception as exc:
        print("rank 0 noticed after %%.1f s: %%s" %% (time.time() - t0, type(exc).__name__))
        sys.exit(5)
    print("rank 0 went on alone")
""") % ROOT


def test_a_dead_rank_is_noticed_not_waited_for(tmp_path):
    """Failure detection on the torch-free rendezvous: when a rank dies, the store on rank 0 sees its connection
    close and ends the job for everybody -- the survivors get an exception at their next collective instead of
    hanging in it (bench.py's parent then kills what is left, tests/test_bench_launch.py)."""
    script = tmp_path / "failing.py"
    script.write_text(FAILING)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VARSENS_RENDEZVOUS="tcp")
        env.pop("TORCHELASTIC_RUN_ID", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    out0, _ = procs[0].communicate(timeout=60)
    procs[1].communicate(timeout=60)
    assert procs[1].returncode == 7
    assert procs[0].returncode == 5 and "noticed" in out0, out0


def test_device_for_rank_follows_the_visible_devices(monkeypatch):
    """LOCAL_RANK is the ordinal where a rank sees the whole node; a launcher that narrows each rank's view to its own
    card leaves one visible device, number 0 -- and a rank that sees none says so."""
    from varsens_amd import _native, dist
    monkeypatch.setattr(_native, "device_count", lambda: 8)
    assert [dist.device_for_rank(r) for r in range(8)] == list(range(8))
    monkeypatch.setattr(_native, "device_count", lambda: 1)
    assert [dist.device_for_rank(r) for r in range(8)] == [0] * 8
    monkeypatch.setattr(_native, "device_count", lambda: 2)
    assert [dist.device_for_rank(r) for r in range(4)] == [0, 1, 0, 1]
    assert dist.device_for_rank(5, shared_device=True) == 0
    monkeypatch.setattr(_native, "device_count", lambda: 0)
    with pytest.raises(Exception, match="sees no GPU"):
        dist.device_for_rank(0)
