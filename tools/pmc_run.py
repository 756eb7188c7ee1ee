#!/usr/bin/env python3
"""rocprofv3 counter passes over one command, one counter group per pass, per-kernel averages as JSON.

    python3 tools/pmc_run.py NAME [--kernels SUBSTR,SUBSTR] [--groups a,b,...] -- python3 tools/gram_only.py

Writes gpurun_out/pmc/NAME.json (and NAME_kernel_stats.csv from a plain --kernel-trace --stats pass).  This
process never touches the GPU; every pass is `rocprofv3 ... -- <the program itself>` (no env / shell hop: the
profiler's preload initialises the GPU before the program starts).  Counters and kernel-trace only -- never
together with --sys-trace or the hip/hsa trace domains.

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE counts half the bytes of 16-B-per-lane
loads on gfx950 (doubled in `hbm_bytes`); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES are cycles; GRBM_GUI_ACTIVE sums the 8 XCDs.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = {
    "issue": "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
             "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
    "lds": "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS "
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE",
    "mem": "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE "
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    opts, cmd = argv[:cut], argv[cut + 1:]
    name = opts[0]
    wanted = None
    groups = list(GROUPS)
    i = 1
    while i < len(opts):
        if opts[i] == "--kernels":
            wanted = opts[i + 1].split(",")
        elif opts[i] == "--groups":
            groups = opts[i + 1].split(",")
        i += 2
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the passes run from /tmp: name the program's files by their place in the repository
    cmd = [os.path.join(root, a) if not os.path.isabs(a) and os.path.isfile(os.path.join(root, a)) else a for a in cmd]
    out = os.path.join(root, "gpurun_out", "pmc")
    work = os.path.join(out, name + "_work")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    env = dict(os.environ, TMPDIR="/tmp")
    pmc, failed = {}, []

    def keep(kname):
        return not kname.startswith("__amd_rocclr") and (wanted is None or any(w in kname for w in wanted))

    # plain kernel-trace + stats: the durations the counters are read beside
    d = os.path.join(work, "stats")
    rc = subprocess.call(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + cmd,
                         cwd="/tmp", env=env, stdout=open(os.path.join(work, "stats.log"), "w"),
                         stderr=subprocess.STDOUT)
    found = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if rc == 0 and found:
        shutil.copy(found[0], os.path.join(out, name + "_kernel_stats.csv"))
    else:
        failed.append("stats")
    for g in groups:
        d = os.path.join(work, g)
        rc = subprocess.call(["rocprofv3", "--pmc"] + GROUPS[g].split() +
                             ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd,
                             cwd="/tmp", env=env, stdout=open(os.path.join(work, g + ".log"), "w"),
                             stderr=subprocess.STDOUT)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if rc != 0 or not files:
            failed.append(g)
            shutil.copy(os.path.join(work, g + ".log"), os.path.join(out, "%s_%s_FAILED.log" % (name, g)))
            continue
        for f in files:
            rows = list(csv.DictReader(open(f)))
            biggest = collections.defaultdict(int)      # full-size launches only (largest grid of each kernel)
            for row in rows:
                biggest[row["Kernel_Name"]] = max(biggest[row["Kernel_Name"]], int(row["Grid_Size"]))
            agg = collections.defaultdict(list)
            for row in rows:
                if keep(row["Kernel_Name"]) and int(row["Grid_Size"]) == biggest[row["Kernel_Name"]]:
                    agg[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
                    pmc.setdefault(row["Kernel_Name"], {}).update(
                        grid_size=int(row["Grid_Size"]), workgroup_size=int(row.get("Workgroup_Size", 0) or 0),
                        rocprof_vgpr_count=int(row.get("VGPR_Count", 0) or 0),
                        rocprof_accum_vgpr_count=int(row.get("Accum_VGPR_Count", 0) or 0),
                        rocprof_sgpr_count=int(row.get("SGPR_Count", 0) or 0),
                        lds_bytes=int(row.get("LDS_Block_Size", 0) or 0),
                        scratch_bytes=int(row.get("Scratch_Size", 0) or 0))
            for (kname, c), v in agg.items():
                pmc[kname][c] = sum(v) / len(v)
                pmc[kname]["launches"] = len(v)
    for kname, c in pmc.items():
        if c.get("GRBM_GUI_ACTIVE"):
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            c["gpu_cycles"] = cycles
            if "SQ_ACTIVE_INST_VALU" in c:
                c["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * 1024.0)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            for part in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if part in c:
                    c[part + "_frac_of_wave_cycles"] = c[part] / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_LDS" in c and c.get("SQ_LDS_BANK_CONFLICT") is not None and c["SQ_ACTIVE_INST_LDS"] > 0:
            c["lds_conflict_per_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
        if "FETCH_SIZE" in c:
            c["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
    units = {"rocprof_vgpr_count": "rocprofv3's VGPR_Count column as it stands; on gfx950 it reads HALF the allocated "
                                   "vector registers per lane (60 for a kernel whose code object says .vgpr_count 119, "
                                   "allocated in blocks of 8 = 120) -- the registers, spills and scratch of a kernel are "
                                   "read off the code object by tools/spill_report.py, not from here",
             "rocprof_accum_vgpr_count": "rocprofv3's Accum_VGPR_Count column, same unit",
             "rocprof_sgpr_count": "rocprofv3's SGPR_Count column", "lds_bytes": "bytes per workgroup",
             "scratch_bytes": "bytes per lane", "FETCH_SIZE, WRITE_SIZE": "KB", "hbm_bytes": "bytes per launch",
             "SQ_*": "per launch, summed over the chip (see the docstring of tools/pmc_run.py)"}
    sys.path.insert(0, root)
    from varsens_amd import build as vs_build   # (pure Python: this process still never touches the GPU)
    json.dump({"command": cmd, "units": units, "library_build_id": vs_build.source_id(), "env": {k: os.environ[k] for k in ("K", "LOG2N") if k in os.environ},
               "failed_groups": failed, "kernels": pmc}, open(os.path.join(out, name + ".json"), "w"), indent=1)
    shutil.rmtree(work, ignore_errors=True)
    print("pmc_run %s: %d kernels, failed groups: %s" % (name, len(pmc), failed or "none"), flush=True)


if __name__ == "__main__":
    main()
