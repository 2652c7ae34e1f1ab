"""A few MLII evaluations at one size (for rocprofv3 --kernel-trace + tools/trace_summary.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y, _ = O.synthetic_problem(n, 8, 20240001, m=1)
th = np.log([np.sqrt(8.0), 1e-2])
with GPR(kernel="rbf") as gp:
    gp.set_data(X, y)
    for _ in range(4):
        gp.nlml(th, grad="exact")
