"""`python bench.py --gpus N` without a launcher starts its own ranks (the reference's fan-out
is the PBS array job of varsens/cluster/accre-submit.sh:27).  CPU side: the spawn / relay / exit-code logic and
the torch-free rendezvous; the GPU side (two ranks sharing the one card) is in test_gpu_api.py."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = dict(os.environ)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                 "VARSENS_RENDEZVOUS"):
        env.pop(name, None)
    return env


def test_bench_spawns_its_own_ranks_and_relays_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--rendezvous-only"], env=clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["backend"] == "tcp"
    assert [r["rank"] for r in line["ranks"]] == [0, 1, 2]
    assert len({r["pid"] for r in line["ranks"]}) == 3 and os.getpid() not in {r["pid"] for r in line["ranks"]}


def test_parent_never_loads_the_engine():
    """The spawning parent must not initialise a GPU: it may not even import the binding."""
    code = textwrap.dedent("""
        import sys
        sys.argv = ["bench.py", "--gpus", "2", "--rendezvous-only"]
        sys.path.insert(0, %r)
        import bench
        try:
            bench.main()
        except SystemExit as e:
            assert e.code == 0, e.code
        bad = [m for m in sys.modules if m.startswith("varsens_amd") or m == "torch"]
        assert not bad, bad
    """) % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=240)
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip())["n_gpus"] == 2


def test_failed_rank_gives_no_result_line_and_the_worst_status(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import os, sys
        rank = int(os.environ["RANK"])
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert os.environ["LOCAL_RANK"] == str(rank) and "TORCHELASTIC_RUN_ID" not in os.environ
        if rank == 0:
            print('{"value": 1}', flush=True)
        sys.exit(3 if rank == 2 else 0)
    """))
    code = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(3, [], script=%r))" \
        % (ROOT, str(child))
    p = subprocess.run([sys.executable, "-c", code], env=clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=240)
    assert p.returncode == 3
    assert p.stdout.strip() == "", "a failed run must not put a result line on stdout"
    assert "[0, 0, 3]" in p.stderr


def test_job_slice_is_what_a_rank_and_its_rehearsal_share():
    """bench.job_slice: the slice arithmetic of a rank of an N-GPU job -- used by the real rank and by its one-GPU
    rehearsal (`--rehearse-rank r --of N`, tools/rehearse_weak_scaling.py).  Weak scaling: 2^log2n rows per GPU, global
    n = N * 2^log2n, rank r owns [r 2^log2n, (r+1) 2^log2n); strong: dist.shard_rows of 2^log2n rows in total.  Either
    way the slices tile [0, n) in rank order, and the shuffle entries a rank uploads are those of its own rows."""
    import sys as _sys
    _sys.path.insert(0, ROOT)
    import bench
    from varsens_amd import dist
    from oracle import saltelli_oracle as orc
    for strong in (False, True):
        for world in (1, 2, 3, 8):
            for log2n in (4, 10):
                at, total = 0, None
                for rank in range(world):
                    n, begin, rows = bench.job_slice(log2n, world, rank, strong)
                    total = n if total is None else total
                    assert n == total == ((1 << log2n) if strong else world << log2n)
                    assert begin == at and rows >= 0
                    if strong:
                        assert (begin, rows) == dist.shard_rows(n, world, rank)
                    else:
                        assert rows == 1 << log2n
                    at += rows
                assert at == total
    # the rehearsal of rank 7 of 8 reads the shuffle entries of rows [7 * 16, 8 * 16) of the 128-row design
    n, begin, rows = bench.job_slice(4, 8, 7, False)
    assert (n, begin, rows) == (128, 112, 16)
    perm = orc.seed1_permutation(n)
    assert sorted(perm) == list(range(n)) and len(perm[begin:begin + rows]) == 16
    import pytest
    with pytest.raises(Exception, match="not in"):
        bench.job_slice(4, 8, 8, False)


def test_rehearsal_arguments_are_checked():
    p = subprocess.run([sys.executable, BENCH, "--rehearse-rank", "8", "--of", "8"], env=clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "0 <= R < N" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-rank", "0", "--of", "8"], env=clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "one GPU" in p.stderr


def test_committed_counter_traffic_prices_only_the_library_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` comes from the committed PMC passes (profiles/rNN/bench_pmc.json).  The file is
    stamped with the build id of the library it was collected on (tools/pmc_run.py); a run that loads another build
    gets `traffic: null` and the reason, never a stale number."""
    import sys as _sys
    _sys.path.insert(0, ROOT)
    import bench
    from varsens_amd import _native
    kernels = {"void reduce_first_kernel<2, 4, false>(vs_reduce_args)": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 24.0}}
    path = tmp_path / "bench_pmc.json"
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(path))
    got, why = bench.pmc_summary(32, 22)
    assert got is None and "missing" in why
    path.write_text(json.dumps({"library_build_id": "0123456789abcdef", "kernels": kernels}))
    got, why = bench.pmc_summary(32, 22)
    assert got is None and "0123456789abcdef" in why and _native.build_id() in why
    assert bench.pmc_traffic(got, "void reduce_first_kernel") is None
    path.write_text(json.dumps({"kernels": kernels}))                       # a summary from before the stamp existed
    assert bench.pmc_summary(32, 22)[0] is None
    path.write_text(json.dumps({"library_build_id": _native.build_id(), "kernels": kernels}))
    got, why = bench.pmc_summary(32, 22)
    assert got == kernels and _native.build_id() in why
    assert bench.pmc_traffic(got, "void reduce_first_kernel") == (2 * 1000.0 + 24.0) * 1024.0
    assert bench.pmc_summary(16, 22)[0] is None                              # counters of another workload
