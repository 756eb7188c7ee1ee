"""CPU-only checks: the C-ABI library loads and exports what include/varsens_hip.h declares,
the ctypes binding covers it, host-side logic (scaling helpers, permutation, sharding,
argument errors) behaves like the reference.  No compute call is made without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as entry
    entry.build()
    from varsens_amd import _native
    return _native


def header_symbols():
    text = open(os.path.join(ROOT, "include", "varsens_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds(native):
    syms = header_symbols()
    assert len(syms) >= 30
    assert sorted(native.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(native):
    lib = native.load_library()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.vs_abi_version() == native.ABI_VERSION


def test_library_says_what_it_was_built_from(native):
    """vs_build_id() = varsens_amd.build.source_id(): sha256 prefix over flags + sources.  bench.py quotes the committed
    counter traffic only when the summary carries this id (tools/pmc_run.py stamps it)."""
    from varsens_amd import build as b
    assert re.fullmatch(r"[0-9a-f]{16}", native.build_id())
    assert native.build_id() == b.source_id()
    assert b.source_id(extra_cflags=["-DVS_SOMETHING"]) != b.source_id(extra_cflags=[])


def test_no_host_fallback(native):
    """Without a GPU the engine refuses to create a context (and says why); with one it works."""
    if native.device_count() == 0:
        with pytest.raises(native.VarsensHipError, match="no HIP device|no host fallback"):
            native.Context(0)
    else:
        native.Context(0).close()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "varsens_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


def test_acc_layout_helpers(native):
    assert native.acc_len(32, 1, 1) == 4 + 131 + 3
    assert native.acc_len(6, 2, 8) == 4 + 2 * 27 + 8 * 5
    assert native.first_order_len(6, 2) == 52
    assert native.gram_len(12) == 2 * 12 * 12     # S S^T and T T^T (include/varsens_hip.h: VS_GRAM_LEN)
    header = open(os.path.join(ROOT, "include", "varsens_hip.h")).read()
    assert "#define VS_GRAM_LEN(K) (2 * (size_t)(K) * (size_t)(K))" in header
    assert "#define VS_ABI_VERSION %d" % native.ABI_VERSION in header


# ---- scale.py (varsens/tests/test_scaling.py restated) --------------------------------

def test_scale_functions_match_reference(golden):
    from varsens_amd import scale
    g = golden("g9_scale")
    pts = g["pts"]
    assert np.array_equal(scale.linear(pts, np.array([-100.0] * 5), np.array([100.0] * 5)), g["linear"])
    assert np.array_equal(scale.power(pts, np.ones(5), np.full(5, 100.0)), g["power"])
    assert np.array_equal(scale.percentage(pts, np.array([-10.0, -1, 0, 1, 20]), 33.0), g["percentage"])
    assert np.array_equal(scale.magnitude(pts, np.array([1, 10, 100, 1000, 1e4])), g["magnitude"])
    for i, want in enumerate([-100.0, -50.0, 0.0, 50.0, 100.0]):                  # test_scaling.py:9
        assert abs(g["linear"][i] - want) < 1e-7
    for i, want in enumerate([-6.7, -0.835, 0.0, 1.165, 26.6]):                   # test_scaling.py:53
        assert abs(scale.percentage(pts, np.array([-10.0, -1, 0, 1, 20]), 33.0)[i] - want) < 1e-7
    lo, hi = np.array([1.0, 10, 1000]), np.array([100.0, 20, 2000])
    assert np.allclose(scale.power(np.zeros(3), lo, hi), lo) and np.allclose(scale.power(np.ones(3), lo, hi), hi)
    assert np.allclose(scale.linear(np.zeros(3), lo, hi), lo) and np.allclose(scale.linear(np.ones(3), lo, hi), hi)


def test_scale_specs_are_callable_and_describe_themselves():
    from varsens_amd import scale, _native
    pts = np.linspace(0, 1, 7)[:, None] * np.ones((1, 3))
    lo, hi = np.array([1.0, 2, 3]), np.array([2.0, 4, 8])
    assert np.array_equal(scale.Linear(lo, hi)(pts), scale.linear(pts, lo, hi))
    assert np.array_equal(scale.Power(lo, hi)(pts), scale.power(pts, lo, hi))
    assert np.array_equal(scale.Percentage(lo, 10.0)(pts), scale.percentage(pts, lo, 10.0))
    assert np.array_equal(scale.Magnitude(lo, 2.0)(pts), scale.magnitude(pts, lo, 2.0))
    mode, l, h = scale.Percentage(lo, 10.0).device_spec(3)
    assert mode == _native.SCALE_LINEAR and np.array_equal(l, lo - 10.0 * lo / 100.0)
    mode, l, h = scale.Linear(-1.0, 1.0).device_spec(4)
    assert l.shape == (4,) and h[3] == 1.0
    with pytest.raises(Exception, match="positive"):
        scale.Power([-1.0, 1, 1], hi).device_spec(3)
    assert scale.Identity().device_spec(5)[0] == _native.SCALE_IDENTITY


def test_scale_doctests():
    import doctest
    from varsens_amd import scale
    assert doctest.testmod(scale).failed == 0


def test_seed1_permutation_matches_reference(golden):
    from varsens_amd.saltelli import seed1_permutation
    g = golden("g7_seed1_perm")
    state = np.random.get_state()[1].copy()
    for n in (1, 2, 5, 13, 1024):
        assert np.array_equal(seed1_permutation(n), g["n%d" % n])
    p = seed1_permutation(100003)
    assert np.array_equal(np.concatenate([p[:64], p[-64:]]), g["n100003"]) and p.dtype == np.uint32
    assert np.array_equal(np.random.get_state()[1], state), "the global numpy RNG must be left alone"


def test_opt_in_numpy_reseed_reproduces_the_references_side_effect():
    """varsens/saltelli.py:100-101 reseeds numpy's GLOBAL generator and shuffles the (n, k) array M_2 with it; code
    that draws from numpy.random afterwards sees a definite stream.  Sample(..., reseed_numpy=True) /
    VARSENS_RESEED_NUMPY=1 leave the generator in exactly that state (the default leaves it alone)."""
    from varsens_amd.saltelli import reseed_numpy_like_the_reference
    saved = np.random.get_state()
    try:
        for n, k in ((1, 3), (2, 1), (13, 5), (1024, 6), (5000, 2)):
            np.random.seed(1)                       # what the reference does ...
            np.random.shuffle(np.zeros((n, k)))
            want = np.random.random(4)              # ... and what its caller then draws
            np.random.seed(12345)
            reseed_numpy_like_the_reference(n)
            assert np.array_equal(np.random.random(4), want), (n, k)
    finally:
        np.random.set_state(saved)


def test_native_permutation_is_numpys_legacy_shuffle(native):
    """vs_seed1_permutation (C restatement of MT19937 + legacy shuffle) == numpy RandomState(1).shuffle."""
    import time
    # 2048 = the batch of the swap loop
    for n in (0, 1, 2, 3, 5, 13, 624, 625, 1024, 2047, 2048, 2049, 2050, 4096, 4097, 6145, 100003, 1 << 20):
        want = np.arange(n, dtype=np.uint32)
        np.random.RandomState(1).shuffle(want)
        assert np.array_equal(native.seed1_permutation(n), want), n
    t0 = time.perf_counter(); native.seed1_permutation(1 << 22); t = time.perf_counter() - t0
    assert t < 2.0


def test_shard_rows_partition():
    from varsens_amd.dist import shard_rows
    for n, w in ((10, 3), (1 << 22, 8), (7, 2), (1, 1), (5, 4), (1000001, 6)):
        parts = [shard_rows(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert all(c % 2 == 0 for _, c in parts[:-1])
    with pytest.raises(Exception):
        shard_rows(10, 2, 2)


def test_objective_specs():
    from varsens_amd import objectives, _native
    g = objectives.GFunction([[0, 1, 2], [2, 1, 0]])
    assert g.outputs(3) == 2 and g.params(3).shape == (2, 3) and g.kind == _native.OBJ_G_FUNCTION
    assert objectives.Linear([1, 2, 3]).outputs(3) == 1
    with pytest.raises(Exception, match="k=4"):
        objectives.Polynomial([1, 2, 3]).params(4)
    with pytest.raises(Exception, match="k=3"):
        objectives.Ishigami().params(5)
    assert objectives.Invert().outputs(9) == 9


def test_bench_cpu_baseline_leg():
    """bench.py's cpu_baseline (oracle timed on host cores) on a tiny input: keys and sanity."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    k, n = 32, 256
    rng = np.random.RandomState(0)
    blocks = 1.0 + rng.rand(2 * k + 2, n, 1)
    cpu, R = bench.cpu_baseline(k, blocks, n, 8)
    for key in ("value", "unit", "cores", "kind", "sample", "eval_loop_evals_per_s", "reduce_only_gsamples_per_s"):
        assert key in cpu
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "GSamples/s" and cpu["value"] > 0
    assert R["sens"].shape == (k, 1)
    # the second-order leg (four tensordots, saltelli.py:612-619) on numpy's default BLAS threads, count stated
    so = cpu["second_order"]
    assert so["blas_threads"] >= 1 and so["cores"] == so["blas_threads"] and so["gsamples_per_s"] > 0
    assert so["compute_varsens_with_second_order_s"] >= so["tensordot_leg_s"] >= 0
    assert R["sens_2"].shape == (k, 1, k, 1) and cpu["host_cores_available"] == 8
    kernels, source = bench.pmc_summary(32, 22)    # the committed counter passes, if they are of this build
    traffic = bench.pmc_traffic(kernels, "void reduce_first_kernel")
    assert (kernels is None and source.startswith("none:") and traffic is None) or traffic > 2.2e9


def test_header_is_plain_c_and_a_c_program_links(native, tmp_path):
    """include/varsens_hip.h compiles as ISO C99; examples/c_abi_demo.c (no Python, no C++) links against
    the library.  Without a GPU the program must stop at vs_context_create with the no-fallback message."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    libdir = os.path.dirname(os.environ.get("VARSENS_HIP_LIB") or native.LIB_PATH)
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", libdir, "-lvarsens_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    if r.returncode != 0:   # no GPU here: the only acceptable failure is the loud one at context creation
        assert r.returncode == 1 and "vs_context_create" in r.stderr and "no host fallback" in r.stderr


# ---- streamed mode: every decision comes from the LARGEST shard, so all ranks agree -----------

def test_stream_plan_is_rank_invariant():
    from varsens_amd import dist
    from varsens_amd.saltelli import stream_passes, stream_rows_per_pass
    # shards differing by two rows: a per-rank decision would split the ranks for R between the sizes
    n, world = 1003, 4
    shards = [dist.shard_rows(n, world, r)[1] for r in range(world)]
    assert sum(shards) == n and max(shards) - min(shards) <= 3
    biggest = max(shards)
    for R in (min(shards), min(shards) + 1, biggest - 1):
        assert stream_rows_per_pass(R, biggest, 100, None) == R          # everybody streams ...
        passes = stream_passes(biggest, R)
        assert all(passes * R >= rows for rows in shards)                # ... and covers its rows
    assert stream_rows_per_pass(biggest, biggest, 100, None) == 0        # one pass holds the largest shard: resident
    assert stream_rows_per_pass(0, biggest, 100, None) == 0 and stream_rows_per_pass(None, biggest, 100, None) == 0
    # 'auto': resident while the largest shard takes at most half of HBM, else <= 8 GiB ranges of whole 64-row blocks
    hbm = 288 << 30
    assert stream_rows_per_pass('auto', 1 << 22, 8 * 66, hbm) == 0
    r = stream_rows_per_pass('auto', 1 << 27, 8 * 258, hbm)
    assert r % 64 == 0 and 0 < r * 8 * 258 <= (8 << 30)


def test_varsens_takes_the_stream_plan_from_the_largest_shard():
    """Varsens._max_rows_over_ranks: communicator max() when there is one, else the standard split."""
    from varsens_amd import dist
    from varsens_amd.saltelli import Varsens

    class Comm(object):
        world, rank = 4, 1

    v = Varsens.__new__(Varsens)
    v._comm, v.n = Comm(), 1003
    mine = dist.shard_rows(1003, 4, 1)[1]
    assert v._max_rows_over_ranks(mine) == max(dist.shard_rows(1003, 4, r)[1] for r in range(4)) > mine

    class CommMax(Comm):
        def max(self, value):
            return 777.0

    v._comm = CommMax()
    assert v._max_rows_over_ranks(mine) == 777
    v._comm = None
    assert v._max_rows_over_ranks(mine) == mine


# ------------------------------------------------------------------ what the compiler made of the kernels

# Kernels allowed to keep a few spilled registers: two fallback kernels of vs_second.hip only (the NaN-masked pass and
# odd row counts beyond 64 index rows, more than 128 index rows: three to ten dwords of row pointers, stored once and
# reloaded in front of each fetch), each with the most it may have -- so that a regression shows.  Every other kernel of the
# library -- the north-star reduction, every scan / fused kernel, the Gram kernels of all BASELINE configurations --
# must have none (DESIGN said "no spills" where the code objects said 11 .. 67).
SPILL_ALLOWED = {"gram_macro_st_kernel": 10}    # the masked / odd-row fallback of 65 .. 128 index rows


def test_hot_kernels_do_not_spill():
    """tools/spill_report.py over the objects of the build (code-object metadata: .vgpr_spill_count,
    .private_segment_fixed_size): no spilled vector register and no scratch in any product-path kernel."""
    import glob
    import importlib.util
    import shutil
    from varsens_amd import build as b
    if not os.path.isdir("/opt/rocm/lib/llvm/bin") or shutil.which("c++filt") is None:
        pytest.skip("needs the ROCm LLVM tools")
    objs = glob.glob(os.path.join(ROOT, "build", "obj", "*.o"))
    if len(objs) < len(b.SOURCES) or b.needs_build():
        b.build(force=True, verbose=False)
    spec = importlib.util.spec_from_file_location("spill_report", os.path.join(ROOT, "tools", "spill_report.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    rows = sr.report()
    assert len(rows) > 400, "the report lost kernels: %d" % len(rows)
    names = " ".join(r["kernel"] for r in rows)
    for must in ("reduce_first_kernel", "scan_kernel<3, 8, 4, true", "scan_gram_kernel", "scan_gramn_kernel",
                 "gram_st_kernel<8, 8, true, true, true>", "gram_diag_wide_kernel", "fold_first_kernel",
                 "gram_dma_kernel<4, true, false, true>", "gram_dma_kernel<4, true, true, true>",
                 "gram_dma_kernel<4, false, true, true>", "gram_dma_kernel<3, false, false, false>"):
        assert must in names, must
    bad = []
    for r in rows:
        base = re.sub(r"^void (\(anonymous namespace\)::)?", "", r["kernel"]).split("<")[0].split("(")[0]
        limit = SPILL_ALLOWED.get(base, 0)
        if r["vgpr_spill"] > limit or (limit == 0 and r["scratch_bytes"] > 0):
            bad.append("%s: %d spilled VGPRs, %d B scratch (allowed %d)" % (r["kernel"], r["vgpr_spill"],
                                                                           r["scratch_bytes"], limit))
    assert not bad, "\n".join(bad)


def test_roofline_table_reads_the_committed_summaries():
    """tools/roofline_table.py over profiles/r05: every row it knows finds its kernel in the committed rocprofv3 summary,
    and the north-star row says what the bench line says (the reduction kernel within 0.85 .. 0.92 of the HBM peak)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), os.path.join(ROOT, "profiles", "r05")],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    rows = [line for line in out.splitlines() if line.startswith("| ") and not line.startswith("| kernel") and not line.startswith("|---")]
    assert len(rows) == 16, len(rows)
    north = [r for r in rows if "north star" in r][0].split("|")
    assert 0.85 <= float(north[6]) <= 0.92, north
    dma = [r for r in rows if r.startswith("| second order (gram_dma_kernel)")][0].split("|")
    before = [r for r in rows if "before gram_dma_kernel" in r][0].split("|")
    assert float(dma[3].split()[0]) < float(before[3].split()[0])


def test_host_entry_points_under_sanitizers():
    """SURVEY.md section 5 (sanitizers: CPU build only on this pool): the host-only entry points of the C ABI --
    vs_seed1_permutation (varsens/saltelli.py:100-101), vs_design_desc_validate (every descriptor check of
    vs_design_create incl. a rank's shuffle slice and VS_PERM_EMPTY), the error store -- built from
    varsens_amd/csrc/vs_host.cpp with g++ -fsanitize=address,undefined and driven by tests/host_asan_main.cpp:
    exit status 0, no sanitizer report, and the permutations are the oracle's."""
    import shutil
    import subprocess
    from varsens_amd import build as b
    from oracle import saltelli_oracle as orc
    if shutil.which("g++") is None and shutil.which("clang++") is None:
        pytest.skip("no host C++ compiler")
    exe = b.build_asan_host(verbose=False)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "host checks ok" in p.stdout and "FAILED" not in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    # "no C++ exception crosses the boundary": the driver replaces `operator new` by one that throws on the i-th
    # allocation; the helpers that allocate came back with VS_ERR_NOMEM every time, nothing leaked, nothing terminated
    assert re.search(r"seed-1 cache: [1-9]\d* allocation failures refused with VS_ERR_NOMEM", p.stdout), p.stdout[-2000:]
    assert re.search(r"sobol tables: [1-9]\d* allocation failures refused with VS_ERR_NOMEM", p.stdout)
    assert re.search(r"text write: [1-9]\d* allocation failures refused with VS_ERR_NOMEM", p.stdout)
    assert re.search(r"text read: [1-9]\d* allocation failures refused with VS_ERR_NOMEM", p.stdout)

    def fnv(a):
        h = 1469598103934665603
        for byte in np.asarray(a, dtype="<u4").tobytes():
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    seen = {}
    for line in p.stdout.splitlines():
        if line.startswith("perm "):
            _, n, h = line.split()
            seen[int(n)] = int(h, 16)
    assert set(seen) >= {0, 1, 2, 5, 13, 1024, 2048, 2049, 100003}
    for n in (0, 1, 2, 3, 5, 13, 1024, 2047, 2048, 2049, 2050, 4097, 100003):
        assert seen[n] == fnv(orc.seed1_permutation(n)), n


def _awkward_values(rows, cols, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((rows, cols)) * 10.0 ** rng.integers(-300, 300, (rows, cols))
    flat = a.reshape(-1)
    special = [np.nan, np.inf, -np.inf, -0.0, 0.0, 5e-324, -5e-324, np.finfo(float).max, np.finfo(float).tiny,
               1.0, -1.0, 0.1, 1e22, 1e23, 9007199254740993.0, 2.2250738585072011e-308]
    flat[:len(special)] = special[:flat.size]
    flat[-1] = np.frombuffer(np.uint64(0xfff8000000000001).tobytes(), dtype=np.float64)[0]   # a negative NaN
    return a


@pytest.mark.parametrize("delimiter", ["\t", " ", ","])
@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (9, 1), (37, 5), (5000, 7)])
def test_text_files_are_numpys(native, tmp_path, delimiter, shape):
    """The text wire format of Sample.export / Objective.export (varsens/saltelli.py:187,192,407,412: numpy.savetxt,
    '%.18e') and of .load (:217,439: numpy.loadtxt): vs_text_write gives numpy.savetxt's bytes, vs_text_read
    numpy.loadtxt's values -- NaN of either sign, infinities, -0.0, denormals, the extremes and values whose 19th digit
    needs correct rounding; 5000 x 7 is past the size where the work is split over threads."""
    a = _awkward_values(*shape, seed=shape[0] * 31 + shape[1])
    ours, theirs = str(tmp_path / "ours.txt"), str(tmp_path / "numpy.txt")
    np.savetxt(theirs, a, delimiter=delimiter)
    native.text_write(ours, a, delimiter)
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    want = np.loadtxt(theirs, delimiter=delimiter, ndmin=2)
    got = native.text_read(theirs, delimiter)
    assert got.dtype == np.float64 and got.shape == want.shape == shape
    assert np.array_equal(got, want, equal_nan=True) and np.array_equal(np.signbit(got), np.signbit(want))
    finite = np.isfinite(a)
    assert np.array_equal(got[finite], a[finite])          # 19 significant digits round-trip a double


def test_text_files_edge_cases(native, tmp_path):
    """What numpy.loadtxt accepts and refuses, line by line: '#' comments, blank lines, CRLF, a last line without its
    newline, a leading '+', any run of blanks as the default delimiter, 1-D arrays one value per line; ragged rows,
    words and missing files are errors that name the file (numpy raises ValueError / FileNotFoundError there)."""
    VarsensHipError = native.VarsensHipError
    path = str(tmp_path / "edge.txt")
    open(path, "w").write("# header\n1 2 3\n\n  \n4\t 5   6 # tail\r\n+7 8e0 .9")
    assert np.array_equal(native.text_read(path), np.loadtxt(path))
    v = np.array([1.5, np.nan, -np.inf])
    native.text_write(path, v)
    np.savetxt(str(tmp_path / "np.txt"), v)
    assert open(path, "rb").read() == open(str(tmp_path / "np.txt"), "rb").read()
    assert native.text_read(path).shape == (3, 1)
    native.text_write(path, np.empty((0, 4)))
    assert os.path.getsize(path) == 0 and native.text_read(path).shape == (0, 0)
    for bad in ("1 2 3\n4 5\n", "1 2 3\n4 5 6 7\n", "1 2 3\n4 five 6\n", "1,2,3\n"):
        open(path, "w").write(bad)
        with pytest.raises(VarsensHipError, match="edge.txt"):
            native.text_read(path, " " if "," not in bad else None)
        with pytest.raises(ValueError):
            np.loadtxt(path)
    with pytest.raises(VarsensHipError, match="cannot open"):
        native.text_read(str(tmp_path / "missing.txt"))
    with pytest.raises(VarsensHipError, match="cannot open"):
        native.text_write(str(tmp_path / "no_such_dir" / "x.txt"), v)
    with pytest.raises(VarsensHipError):
        native.text_write(path, np.zeros((2, 2, 2)))


def test_every_entry_point_is_guarded():
    """include/varsens_hip.h: "no C++ exception crosses the boundary".  Every `extern "C"` function of the library that
    returns a status opens with VS_GUARD_BEGIN and closes with VS_GUARD_END (csrc/vs_guard.h: std::bad_alloc ->
    VS_ERR_NOMEM, anything else -> VS_ERR_STATE); the two that cannot fail (vs_abi_version, vs_last_error) are one-line
    returns.  The sources are the evidence here -- the device entry points cannot run without a GPU; the host-only ones
    run under a failing `operator new` in test_host_entry_points_under_sanitizers."""
    csrc = os.path.join(ROOT, "varsens_amd", "csrc")
    declared = set(header_symbols())
    defined = {}
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".cpp")):
            continue
        lines = open(os.path.join(csrc, name)).read().split("\n")
        for i, line in enumerate(lines):
            m = re.match(r'extern "C" (?:const char\*|int) (vs_\w+)\(', line)
            if not m:
                continue
            j = i
            while not lines[j].rstrip().endswith(("{", "}")):
                j += 1
            if lines[j].rstrip().endswith("}"):      # one-liner
                defined[m.group(1)] = "trivial" if re.search(r"\{ return [A-Za-z_]+; \}$", lines[j]) else "unguarded"
                continue
            end = j + 1
            while lines[end] != "}":
                end += 1
            body = [b.strip() for b in lines[j + 1:end] if b.strip()]
            defined[m.group(1)] = "guarded" if body[0] == "VS_GUARD_BEGIN" and body[-1] == "VS_GUARD_END" else "unguarded"
    assert set(defined) == declared, sorted(set(defined) ^ declared)
    assert sorted(n for n, how in defined.items() if how == "trivial") == ["vs_abi_version", "vs_build_id", "vs_last_error"]
    assert not [n for n, how in defined.items() if how == "unguarded"]
    guard = open(os.path.join(csrc, "vs_guard.h")).read()
    assert "catch (const std::bad_alloc&)" in guard and "VS_ERR_NOMEM" in guard and "catch (...)" in guard


def test_descriptor_checks_through_the_binding(native):
    """vs_design_desc_validate over ctypes (the product library, no GPU call): what Sample(...) relies on."""
    import ctypes as C
    lib = native.load_library()
    d = native.DesignDesc()
    d.generator, d.k, d.n, d.first_index, d.scale_mode = native.GEN_HALTON, 4, 50, 81, native.SCALE_IDENTITY
    entries = C.c_uint64(7)
    assert lib.vs_design_desc_validate(C.byref(d), C.byref(entries)) == 0 and entries.value == 0
    d.k = 0
    assert lib.vs_design_desc_validate(C.byref(d), None) == -1
    assert b"out of range" in lib.vs_last_error()
    # ABI 7: the library's own seed-1 shuffle (VS_DESIGN_SEED1_SHUFFLE) -- nothing read from the host, slice checked,
    # a perm pointer beside the flag and unknown flags refused
    d.k, d.flags = 4, native.DESIGN_SEED1_SHUFFLE
    assert lib.vs_design_desc_validate(C.byref(d), C.byref(entries)) == 0 and entries.value == 0
    d.perm_begin, d.perm_count = 10, 40
    assert lib.vs_design_desc_validate(C.byref(d), C.byref(entries)) == 0 and entries.value == 0
    d.perm_count = 41
    assert lib.vs_design_desc_validate(C.byref(d), None) == -1 and b"exceeds n" in lib.vs_last_error()
    d.perm_begin, d.perm_count = 50, native.PERM_EMPTY
    assert lib.vs_design_desc_validate(C.byref(d), None) == 0
    keep = np.zeros(50, np.uint32)
    d.perm_begin, d.perm_count, d.perm = 0, 0, keep.ctypes.data
    assert lib.vs_design_desc_validate(C.byref(d), None) == -1 and b"perm must be NULL" in lib.vs_last_error()
    d.perm, d.flags = None, 6
    assert lib.vs_design_desc_validate(C.byref(d), None) == -1 and b"unknown design flags" in lib.vs_last_error()
