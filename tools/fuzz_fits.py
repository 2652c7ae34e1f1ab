"""Randomised single fits and small lockstep batches against the oracle: random order n (block-boundary cases included), feature
count, ride rows, kernel, precision, outer panel width and schedule options.  Prints the worst relative errors; exits non-zero on a
tolerance violation.  usage: fuzz_fits.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

def run(cases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  edge = [1, 2, 15, 16, 17, 127, 128, 129, 255, 256, 257, 383, 384, 385, 1023, 1024, 1025, 1151, 1152, 2047, 2048, 2049]
  worst = {"mean": 0.0, "var": 0.0, "nlml": 0.0, "sigma_f": 0.0}
  t0 = time.time()
  fails = 0
  for case in range(cases):
      n = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 2600))
      d = int(rng.integers(1, 41)); m = int(rng.integers(0, 5))
      kind = str(rng.choice(["rbf", "matern52"]))
      dtype = "f32" if (rng.random() < 0.25 and d <= 64 and m <= 3) else "f64"
      W = int(rng.choice([1, 2, 3, 4, 8, 16]))
      opts = {"panel_chain": int(rng.choice([0, 1, 3])), "first_on_panel": int(rng.choice([0, 1, 2])), "lookahead": int(rng.choice([0, 1, 1])),
              "chain_rows": int(rng.choice([0, 8, 80, 1000]))}
      X, y, Xs = O.synthetic_problem(n, d, 1000 + case, m=max(m, 1))
      Xs = Xs[:m] if m else None
      ell = float(np.sqrt(d) * 10 ** rng.uniform(-0.5, 0.5)); sn = float(10 ** rng.uniform(-2, 0))
      ref = O.fit_predict(X, y, Xs if m else X[:1], ell, sn, kind=kind, ref_idiom=False)
      with GPR(kernel=kind, outer_blocks=W, dtype=dtype) as gp:
          for k, v in opts.items():
              gp.set_option(k, v)
          gp.fit(X, y, ell, sn, Xs=Xs)
          got = {"nlml": gp.nlml_, "sigma_f": gp.sigma_f_}
          if m:
              mu, var = gp.predict(Xs)
              got["mean"], got["var"] = mu, var
      tol = {"mean": 1e-8, "var": 1e-8, "nlml": 1e-9, "sigma_f": 1e-8} if dtype == "f64" else {"mean": 1e-6, "var": 1e-5, "nlml": 5e-5, "sigma_f": 1e-6}   # fp32 factor: its log-determinant carries fp32 rounding
      refd = {"mean": ref["fmean"][:m], "var": ref["fvar"][:m], "nlml": ref["nlml"], "sigma_f": ref["sigma_f"]}
      for k in got:
          a, b = np.atleast_1d(np.asarray(got[k], dtype=float)), np.atleast_1d(np.asarray(refd[k], dtype=float))
          e = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
          if dtype == "f64":
              worst[k] = max(worst[k], e)
          if not (e <= tol[k]):
              fails += 1
              print("FAIL case %d: n=%d d=%d m=%d %s %s W=%d %s : %s rel err %.3e" % (case, n, d, m, kind, dtype, W, opts, k, e), flush=True)
  if verbose:
    print("%d cases in %.1f s, %d failures; worst fp64 relative errors: %s" % (cases, time.time() - t0, fails, {k: "%.1e" % v for k, v in worst.items()}))
  return fails, worst


if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    sys.exit(1 if f else 0)
