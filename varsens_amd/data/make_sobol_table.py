"""Regenerate sobol_joe_kuo_1024x32.npy: left-justified 32-bit Joe-Kuo direction
integers v[c][j] for the first 1024 dimensions, taken from scipy's own
initialiser (new-joe-kuo-6.21201 data shipped with scipy)."""
import os

import numpy as np
from scipy.stats import _sobol

d, bits = 1024, 32
sv = np.zeros((d, bits), dtype=np.uint64)
_sobol._initialize_v(sv, dim=d, bits=bits)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sobol_joe_kuo_1024x32.npy"),
        sv.astype(np.uint32))
