"""BASELINE configs[4] at its stated size (n = 32768, d = 32, Matern-5/2, sn~ = 0.1) through the ORACLE
(oracle.gp_oracle.fit_predict_lean = the ref_idiom=False statements of fit_predict in row blocks + the blocked Cholesky):
tests/golden/config4_oracle.npz.  ~3 min on 8 cores, 16 GiB; the GPU tier compares the fp32 + refinement engine and the fp64
engine with these numbers instead of recomputing them in every run (SIGP_LIVE_ORACLE=1 recomputes).  The kernel FUNCTION
(Matern-5/2) has no counterpart in the reference: this fixture is an oracle output, not a reference capture.
    python tests/golden/make_golden_config4.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as O  # noqa: E402

N, D, SEED, M = 32768, 32, 20240004, 2
ELL, SN = float(np.sqrt(D)), 1e-1

if __name__ == "__main__":
    X, y, Xs = O.synthetic_problem(N, D, SEED, m=M)
    r = O.fit_predict_lean(X, y, Xs, ELL, SN, kind="matern52", threads=8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config4_oracle.npz"), n=N, d=D, seed=SEED, m=M, ell=ELL, sn=SN,
                        fmean=r["fmean"], fvar=r["fvar"], sigma_f=r["sigma_f"], sigma_n=r["sigma_n"], nlml=r["nlml"], A_tilde=r["A_tilde"][:, 0],
                        numpy=np.__version__)
    print("wrote config4_oracle.npz", r["fmean"], r["fvar"], r["sigma_f"], r["nlml"])
