"""Wall time of the two routes a reference user runs with an OPAQUE model:

  (c) pre-computed values: Objective(k, n, objective_vals=v) -> Varsens(o)   (varsens/saltelli.py:297-298, 415-495)
  (b) host callable:       Sample -> Objective(k, n, s, f) -> Varsens(o)     (varsens/saltelli.py:308-355)

beside the oracle's load_objective + compute_varsens on the same data (the reference's CPU path, restated) and the raw
PCIe copy of the same bytes.  The legs themselves live in bench.py (`precomputed_route`, `host_callable_route`: its
result line carries them); this is their command line for other shapes.

    python3 tools/host_routes.py [k log2n l ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


from bench import host_callable_route, precomputed_route, synthetic_values  # noqa: E402,F401  (bench.py holds the legs)


if __name__ == "__main__":
    from varsens_amd import _native
    ctx = _native.Context(0)
    args = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(32, 20, 1), (32, 20, 3)]
    # first touch of everything (library load, first allocation, clocks)
    precomputed_route(ctx, 6, 10, 1, cpu=False)
    for k, log2n, l in shapes:
        print(json.dumps({"precomputed_route": precomputed_route(ctx, k, log2n, l)}), flush=True)
    if os.environ.get("HOST_CALLABLE", "1") != "0":
        host_callable_route(ctx, 6, 10, cpu=False)
        print(json.dumps({"host_callable_route": host_callable_route(ctx, 32, int(os.environ.get("HC_LOG2N", 18)))}), flush=True)
