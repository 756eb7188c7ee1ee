"""Predict the 1 -> N GPU weak-scaling efficiency of bench.py on ONE GPU (no N > 1 curve has been
measured, and the per-rank work is not constant in N).

    python3 tools/rehearse_weak_scaling.py [--of 8] [--ranks 0,3,7] [--out DIR]

Runs, one after the other on this GPU: the N = 1 step loop of bench.py (the denominator of a weak-scaling curve), then
`bench.py --rehearse-rank r --of N` for each r -- the same loop over rank r's rows of the N-times-longer global design,
without the exchange -- and the N = 1 loop once more (box drift).  Writes every line to DIR and one summary:

    predicted_weak_efficiency = ms_per_step(N = 1) / max_r ms_per_step(rank r of N)

What the prediction leaves out: the all-reduce itself (1 KB, issued on the auxiliary stream beside the next evaluation;
`allreduce_ms` of a real run prices it) and anything the ranks of a real node share (host memory, PCIe, power).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    if p.returncode != 0:
        raise SystemExit("bench.py %s failed:\n%s" % (" ".join(argv), p.stderr[-3000:]))
    return json.loads(p.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--of", type=int, default=8)
    ap.add_argument("--ranks", default="0,3,7")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05", "rehearsal"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    common = ["--k", str(args.k), "--log2n", str(args.log2n), "--steps", str(args.steps), "--warmup", "20"]
    base_argv = common + ["--no-cpu-baseline", "--no-host-routes"]
    lines = {"n1_before": run(base_argv)}
    print("N=1: %.4f ms per step" % lines["n1_before"]["ms_per_step"], flush=True)
    for r in [int(x) for x in args.ranks.split(",")]:
        lines["rank%d_of_%d" % (r, args.of)] = run(common + ["--rehearse-rank", str(r), "--of", str(args.of)])
        print("rank %d of %d: %.4f ms per step, stages %s" % (r, args.of, lines["rank%d_of_%d" % (r, args.of)]["ms_per_step"],
                                                            lines["rank%d_of_%d" % (r, args.of)]["stages_ms"]), flush=True)
    lines["n1_after"] = run(base_argv)
    print("N=1 again: %.4f ms per step" % lines["n1_after"]["ms_per_step"], flush=True)
    for name, line in lines.items():
        with open(os.path.join(args.out, "bench_%s.json" % name), "w") as fh:
            fh.write(json.dumps(line) + "\n")
    t1 = min(lines["n1_before"]["ms_per_step"], lines["n1_after"]["ms_per_step"])
    ranks = {name: line for name, line in lines.items() if name.startswith("rank")}
    worst = max(ranks, key=lambda nm: ranks[nm]["ms_per_step"])
    stages1 = lines["n1_after"]["stages_ms"]
    summary = {
        "workload": "bench.py step, k=%d, 2^%d rows per GPU, weak scaling to N=%d" % (args.k, args.log2n, args.of),
        "ms_per_step_n1": {"before": lines["n1_before"]["ms_per_step"], "after": lines["n1_after"]["ms_per_step"]},
        "ms_per_step_rank_of_n": {name: line["ms_per_step"] for name, line in ranks.items()},
        "stages_ms_n1": stages1,
        "stages_ms_rank_of_n": {name: line["stages_ms"] for name, line in ranks.items()},
        "stage_growth_worst_rank": {st: ranks[worst]["stages_ms"].get(st, float("nan")) / stages1[st]
                                    for st in stages1 if stages1[st] > 0 and st in ranks[worst]["stages_ms"]},
        "fused_kernel_ms": {"n1": (lines["n1_after"].get("fused_path") or {}).get("kernel_ms"),
                            **{name: (line.get("fused_path") or {}).get("kernel_ms") for name, line in ranks.items()}},
        "worst_rank": worst,
        "predicted_weak_efficiency": t1 / ranks[worst]["ms_per_step"],
        "not_in_the_prediction": "the all-reduce (1 KB per step, on the auxiliary stream beside the next evaluation) and "
                                 "whatever the GPUs of a real node share; NO N > 1 run has been measured",
    }
    with open(os.path.join(args.out, "predicted_weak_scaling.json"), "w") as fh:
        fh.write(json.dumps(summary, indent=1) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
