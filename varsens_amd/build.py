"""Build libvarsens_hip.so in-tree with hipcc for gfx950.

    python -m varsens_amd.build [--force]
    python -m varsens_amd.build --asan-host     the host-only part (csrc/vs_host.cpp) under AddressSanitizer and
                                                UBSan with the driver tests/host_asan_main.cpp: build/asan/host_asan
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvarsens_hip.so")
SOURCES = ["vs_host.cpp", "vs_api.hip", "vs_generate.hip", "vs_scan.hip", "vs_scan_g.hip", "vs_scan_poly.hip", "vs_scan_lin.hip", "vs_scan_gram.hip", "vs_scan_gram8.hip", "vs_scan_gram8_g.hip", "vs_scan_gram8_poly.hip", "vs_scan_gram8_lin.hip", "vs_scan_gram4.hip", "vs_scan_gramm_g.hip", "vs_scan_gramm_poly.hip", "vs_scan_gramm_lin.hip", "vs_reduce.hip", "vs_values.hip", "vs_second.hip", "vs_gram_st.hip", "vs_gram_dma.hip", "vs_comm.hip", "vs_rtc.hip"]
# device sources hiprtc needs at run time, embedded into the library as C++ raw strings
EMBED = {"vs_device_gen.h": "vs_rtc_embed_gen.inc", "vs_device_scan.h": "vs_rtc_embed_scan.inc",
         "vs_rtc_kernel.h": "vs_rtc_embed_kernel.inc", "vs_rtc_kernel_sep.h": "vs_rtc_embed_kernel_sep.inc"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # numpy-exact arithmetic: no implicit fma contraction; fmas are written out
         "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libvarsens_hip.so cannot be built")


def source_id(extra_cflags=None):
    """Identity of what the library is built FROM: sha256 over the compiler flags and every source the build reads
    (csrc/* but the generated .inc files, include/*.h), first 16 hex digits.  It is compiled into the library
    (vs_build_id()) and stamped into the counter summaries tools/pmc_run.py writes, so that bench.py can tell whether a
    committed `roofline.traffic` figure was measured on the kernels it is about to price -- and say `traffic: null`
    when it was not.  Independent of the compiler's output being reproducible."""
    import hashlib
    h = hashlib.sha256()
    flags = list(FLAGS) + (os.environ.get("VS_EXTRA_CFLAGS", "").split() if extra_cflags is None else list(extra_cflags))
    h.update(" ".join(flags).encode())
    files = [f for f in glob.glob(os.path.join(CSRC, "*")) if not f.endswith(".inc")] + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    for f in sorted(files, key=os.path.basename):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [d for d in glob.glob(os.path.join(CSRC, "*")) if not d.endswith(".inc")] + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def embed_sources():
    for src, dst in EMBED.items():
        text = open(os.path.join(CSRC, src)).read()
        assert ')VSRTC"' not in text
        out = 'R"VSRTC(' + text + ')VSRTC"\n'
        path = os.path.join(CSRC, dst)
        if not os.path.isfile(path) or open(path).read() != out:
            with open(path, "w") as fh:
                fh.write(out)


def _deps_newer(target):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, extra_cflags, verbose=False):
    """A/B tooling (tools/ab_*.sh): the library built with extra compiler flags into build_variants/libvarsens_hip_<tag>.so
    (objects under build/obj_<tag>); load it with VARSENS_HIP_LIB.  Never used by the product."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(ROOT, "build", "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "build_variants"), exist_ok=True)
    out = os.path.join(ROOT, "build_variants", "libvarsens_hip_%s.so" % tag)
    embed_sources()
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_cflags)
    bid = source_id(extra_cflags=extra_cflags)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o").replace(".cpp", ".o"))
        cmd = [hipcc] + cflags + ["-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src), "-o", obj]
        if src == "vs_host.cpp":
            cmd.insert(1, '-DVS_BUILD_ID="%s"' % bid)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L/opt/rocm/lib", "-lhiprtc"])
    return out


def build(force=False, verbose=True):
    """One object per source (compiled side by side), then one link.  The library is written to a
    temporary name and renamed into place under a file lock, so that the ranks of a multi-process
    job that all find it missing never load a half-written file."""
    if not force and not needs_build():
        return OUT
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():     # another process built it while this one waited
            return OUT
        embed_sources()
        hipcc = _hipcc()
        # VS_EXTRA_CFLAGS: diagnostic builds of the tools/ scripts (e.g. -DVS_SCAN_STAMPS); never set by the product
        cflags = [f for f in FLAGS if f != "-shared"] + os.environ.get("VS_EXTRA_CFLAGS", "").split()
        build_id = source_id()

        again = [False]   # a second round: only what an edit during the first one made stale, and vs_host.o for the id

        def compile_one(src):
            obj = os.path.join(objdir, src.replace(".hip", ".o").replace(".cpp", ".o"))
            if (force and not again[0]) or _deps_newer(obj) or (again[0] and src == "vs_host.cpp"):
                cmd = [hipcc] + cflags + ["-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src), "-o", obj]
                if src == "vs_host.cpp":    # vs_build_id()
                    cmd.insert(1, '-DVS_BUILD_ID="%s"' % build_id)
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
            return obj

        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        # A source edited while the objects were being compiled would leave vs_build_id() naming sources the library was
        # not (wholly) built from: go round again -- the objects older than the edit and vs_host.o with the new id
        for _ in range(3):
            now = source_id()
            if now == build_id:
                break
            build_id = now
            again[0] = True
            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
                objs = list(pool.map(compile_one, SOURCES))
        tmp = OUT + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-L/opt/rocm/lib", "-lhiprtc"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    return OUT


def build_asan_host(verbose=True):
    """The sanitizer leg (GPU sanitizers are not available on the pool): everything of the library
    that runs without a GPU -- the error store, the seed-1 shuffle, the descriptor checks -- is one plain C++ file;
    g++ builds it with -fsanitize=address,undefined together with the driver program.  Returns the executable."""
    gxx = shutil.which("g++") or shutil.which("clang++")
    if gxx is None:
        raise RuntimeError("no host C++ compiler for the sanitizer build")
    outdir = os.path.join(ROOT, "build", "asan")
    os.makedirs(outdir, exist_ok=True)
    exe = os.path.join(outdir, "host_asan")
    srcs = [os.path.join(CSRC, "vs_host.cpp"), os.path.join(ROOT, "tests", "host_asan_main.cpp")]
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(CSRC, h) for h in ("vs_guard.h", "vs_host.h")]
    if os.path.isfile(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
        return exe
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + srcs + ["-o", exe]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    if "--variant" in sys.argv:     # python -m varsens_amd.build --variant TAG -DFLAG ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--asan-host" in sys.argv:
        print(build_asan_host())
    else:
        build(force="--force" in sys.argv)
