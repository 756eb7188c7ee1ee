"""Build libvarsens_hip.so in-tree with hipcc for gfx950.

    python -m varsens_amd.build [--force]
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
SOURCES = ["vs_api.hip", "vs_generate.hip", "vs_reduce.hip", "vs_second.hip", "vs_comm.hip", "vs_rtc.hip"]
# device sources hiprtc needs at run time, embedded into the library as C++ raw strings
EMBED = {"vs_device_gen.h": "vs_rtc_embed_gen.inc", "vs_rtc_kernel.h": "vs_rtc_embed_kernel.inc",
         "vs_rtc_kernel_sep.h": "vs_rtc_embed_kernel_sep.inc"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # numpy-exact arithmetic: no implicit fma contraction; fmas are written out
         "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libvarsens_hip.so cannot be built")


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


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    embed_sources()
    cmd = [_hipcc()] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-o", OUT] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-L/opt/rocm/lib", "-lhiprtc"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
