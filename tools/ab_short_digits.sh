#!/bin/bash
# A/B inside the sustained evaluate -> reduce step (tools/eval_in_step.py) and in bench.py: the product against
# build_variants/libvarsens_hip_shortall.so (python -m varsens_amd.build --variant shortall -DVS_SCAN_SHORT_ALL: the
# three-instruction Halton digit step and the two-word quotient also in the evaluation kernels of up to 96 columns).
# Round 4 measured the short form SLOWER there in the planar layout; asked again for the row-blocked one.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rnd in 1 2; do
for lib in "" $R/build_variants/libvarsens_hip_shortall.so; do
  echo "== lib: ${lib:-product}"
  VARSENS_HIP_LIB=$lib K="16 32 48 64 96" python3 $R/tools/eval_in_step.py
  VARSENS_HIP_LIB=$lib python3 $R/bench.py --no-host-routes --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench: value %.1f step %.4f eval %.4f reduce %.4f' % (d['value'], d['ms_per_step'], d['stages_ms']['eval'], d['stages_ms']['reduce_first']))"
done
done
