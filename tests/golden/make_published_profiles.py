#!/usr/bin/env python3
"""The reference's published accuracy profiles as a fixture: the numeric rows of
varsens/profile/error-profile-dim{6,12,24,48,96,192}.csv (written by varsens/profile/parameters.py:77-83,147:
n, mean, sd, lower CI, upper CI, max of sum_i (V_i - S_i var_y)^2 over b random g-functions).  Data only.

    python tests/golden/make_published_profiles.py        (needs /root/reference; writes published_error_profiles.json)
"""
import json
import os

import numpy as np

REF = "/root/reference/varsens/profile"
HERE = os.path.dirname(os.path.abspath(__file__))

out = {"columns": ["n", "mean", "sd", "lci", "uci", "max"], "source": "varsens/profile/error-profile-dim<k>.csv", "dims": {}}
for k in (6, 12, 24, 48, 96, 192):
    rows = np.loadtxt(os.path.join(REF, "error-profile-dim%d.csv" % k), delimiter=",")
    out["dims"][str(k)] = [[float("%.17g" % v) for v in row] for row in rows]
with open(os.path.join(HERE, "published_error_profiles.json"), "w") as fh:
    json.dump(out, fh, indent=0)
print("wrote", {k: len(v) for k, v in out["dims"].items()})
