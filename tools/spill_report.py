#!/usr/bin/env python3
"""Registers, spills and scratch of every gfx950 kernel in the build, read off the code objects.

    python3 tools/spill_report.py [--csv OUT.csv] [--spilled-only] [OBJECT_OR_LIBRARY ...]

For each object under build/obj (default) or each file named: the `.hip_fatbin` section is dumped
(llvm-objcopy), the gfx950 code object unbundled (clang-offload-bundler) and its AMDGPU metadata note
read (llvm-readelf --notes): .vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count,
.sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size
(static LDS), .max_flat_workgroup_size.  Kernel names are demangled (c++filt).  No GPU needed:
this is what the compiler decided, the same facts rocprofv3 reports per dispatch.

`report(paths)` returns the rows for tests/test_abi_and_host.py::test_hot_kernels_do_not_spill.
"""
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("VS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def tool(name):
    path = os.path.join(LLVM, name)
    if not os.path.isfile(path):
        raise RuntimeError("%s not found (set VS_LLVM_BIN)" % path)
    return path


def code_object_notes(path, work):
    """The metadata note text of the gfx950 code object inside `path` (.o, .so or a bare code object)."""
    fat = os.path.join(work, "fat.bin")
    co = os.path.join(work, "k.co")
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    rc = subprocess.call([tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(work, "x")],
                         stderr=subprocess.DEVNULL)
    if rc != 0 or not os.path.isfile(fat) or os.path.getsize(fat) == 0:
        co = path           # no fat binary inside: taken to be a code object itself
    else:
        subprocess.check_call([tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=" + TARGET, "--output=" + co])
    return subprocess.run([tool("llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          text=True).stdout


def parse_notes(text):
    """One dict per kernel from the YAML-like metadata dump (each kernel entry opens with '  - ' at depth 1)."""
    kernels, cur = [], None
    in_kernels = False
    for line in text.splitlines():
        if line.startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if in_kernels and re.match(r"^[A-Za-z]", line):
            in_kernels = False
        if not in_kernels:
            continue
        if line.startswith("  - "):
            cur = {}
            kernels.append(cur)
            line = "    " + line[4:]
        m = re.match(r"^    (\.[a-z_]+):\s+(\S.*)$", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip()
    return [k for k in kernels if ".name" in k]


def demangle(names):
    if not names:
        return []
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.isfile(filt):
        return list(names)
    out = subprocess.run([filt], input="\n".join(names) + "\n", stdout=subprocess.PIPE, text=True).stdout
    return out.splitlines()


def report(paths=None):
    """Rows {object, kernel, mangled, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_bytes,
    max_workgroup} for every kernel in `paths` (default: build/obj/*.o)."""
    if not paths:
        objdir = os.path.join(ROOT, "build", "obj")
        paths = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o"))
    rows = []
    with tempfile.TemporaryDirectory() as work:
        for path in paths:
            ks = parse_notes(code_object_notes(path, work))
            pretty = demangle([k[".name"] for k in ks])
            for k, name in zip(ks, pretty):
                v = [int(k.get(f, 0) or 0) for f in FIELDS]
                rows.append(dict(object=os.path.basename(path), kernel=name, mangled=k[".name"], vgpr=v[0], agpr=v[1],
                                 sgpr=v[2], vgpr_spill=v[3], sgpr_spill=v[4], scratch_bytes=v[5], lds_bytes=v[6],
                                 max_workgroup=v[7]))
    return rows


def main():
    args = sys.argv[1:]
    out = None
    if "--csv" in args:
        i = args.index("--csv")
        out = args[i + 1]
        del args[i:i + 2]
    spilled_only = "--spilled-only" in args
    args = [a for a in args if a != "--spilled-only"]
    rows = report(args)
    cols = ["object", "kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes", "lds_bytes",
            "max_workgroup"]
    if out:
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(cols)
            for r in rows:
                w.writerow([r[c] for c in cols])
    shown = [r for r in rows if not spilled_only or r["vgpr_spill"] or r["scratch_bytes"]]
    for r in shown:
        print("%-22s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d B  lds %6d  wg %4d  %s"
              % (r["object"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"],
                 r["lds_bytes"], r["max_workgroup"], r["kernel"][:150]))
    n_spill = sum(1 for r in rows if r["vgpr_spill"])
    print("%d kernels, %d with spilled vector registers, %d with scratch" %
          (len(rows), n_spill, sum(1 for r in rows if r["scratch_bytes"])))


if __name__ == "__main__":
    main()
