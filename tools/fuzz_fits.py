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
      d = int(rng.integers(1, 41)); m = int(rng.integers(0, 5)) if rng.random() < 0.7 else int(rng.choice([14, 15, 16, 20]))     # (16 rows in use is where the ride-row tile form ends)
      kind = str(rng.choice(["rbf", "matern52"]))
      dtype = "f32" if (rng.random() < 0.25 and d <= 64 and m <= 3) else "f64"
      W = int(rng.choice([1, 2, 3, 4, 8, 16]))
      opts = {"panel_chain": int(rng.choice([0, 1, 3, 5, 7, 8, 15, 15])), "first_on_panel": int(rng.choice([0, 1, 2])), "lookahead": int(rng.choice([0, 1, 1])),
              "chain_rows": int(rng.choice([0, 8, 80, 1000])), "link_rows": int(rng.choice([0, 16, 256, 4096])), "kbuild_mfma": int(rng.choice([0, 1, 2])),
              "refine_sym": int(rng.choice([0, 1, 1])),
              # the 128-tile update kernel (and with it the diagonal / ride-row tile forms) and the strip solve at sizes that would not reach them by default
              "small_tile_threshold": int(rng.choice([1, 320])), "tiny_tile_threshold": int(rng.choice([1, 256])), "panel_mode": int(rng.choice([0, 1, 2])),
              "diag_tiles": int(rng.choice([0, 1, 1])), "ride_tiles": int(rng.choice([0, 1, 1])), "strip_tri": int(rng.choice([0, 1, 1]))}
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
          if k == "nlml" and dtype == "f32":      # the fp32 factor's log-determinant: an absolute error of ~1e-7 per order, whatever nlML's own size
              e = min(e, float(np.max(np.abs(a - b))) / (1e-6 * max(n, 1)) * tol[k])
          if not (e <= tol[k]):
              fails += 1
              print("FAIL case %d: n=%d d=%d m=%d %s %s W=%d %s : %s rel err %.3e" % (case, n, d, m, kind, dtype, W, opts, k, e), flush=True)
  if verbose:
    print("%d cases in %.1f s, %d failures; worst fp64 relative errors: %s" % (cases, time.time() - t0, fails, {k: "%.1e" % v for k, v in worst.items()}))
  return fails, worst


def run_batches(cases, seed, verbose=True):
  """Lockstep batches: B data sets x F fits in groups of G (ragged last group), strips forced on small shapes through strip_min,
  chain / recursion top blocks, one member made singular now and then (its info > 0, the others untouched)."""
  rng = np.random.default_rng(seed)
  fails = 0; worst = 0.0; t0 = time.time()
  for case in range(cases):
      n = int(rng.integers(100, 1700)); d = int(rng.integers(1, 12)); B = int(rng.integers(1, 5)); F = int(rng.integers(2, 11))
      G = int(rng.integers(1, F + 1)); W = int(rng.choice([2, 4, 8]))
      kind = str(rng.choice(["rbf", "matern52"]))
      Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
      for b in range(B):
          Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 5000 + 10 * case + b, m=1)
      ell = np.sqrt(d) * 10 ** rng.uniform(-0.4, 0.4, F); sn = 10 ** rng.uniform(-2, 0, F)
      bad = int(rng.integers(0, F)) if (rng.random() < 0.3 and n > 40) else -1
      if bad >= 0:
          Xb[bad % B, n // 2] = Xb[bad % B, n // 3]; sn[bad] = 0.0      # duplicate row + no noise: K~ singular for every fit of that data set with sn~ = 0
      opts = {"panel_chain": int(rng.choice([0, 1, 3])), "strip_min": int(rng.choice([1, 8, 512])), "first_on_panel": int(rng.choice([0, 1, 2])),
              "small_tile_threshold": int(rng.choice([1, 320])), "strips_after_update": int(rng.choice([0, 1])),
              "diag_tiles": int(rng.choice([0, 1, 1])), "ride_tiles": int(rng.choice([0, 1, 1])), "strip_tri": int(rng.choice([0, 1, 1]))}
      with GPR(kernel=kind, outer_blocks=W, panel_mode=str(rng.choice(["auto", "strips", "recursive"]))) as gp:
          for k, v in opts.items():
              gp.set_option(k, v)
          r = gp.fit_batch(Xb, yb, Xsb, ell, sn, concurrency=int(rng.choice([1, 2])), group=G)
      for i in range(F):
          b = i % B
          if i == bad:
              continue        # singular to rounding: either outcome (info > 0 or a huge condition number) is legitimate
          ref = O.fit_predict(Xb[b], yb[b], Xsb[b], float(ell[i]), float(sn[i]), kind=kind, ref_idiom=False)
          errs = [abs(r["mean"][i, 0] - ref["fmean"][0]) / abs(ref["fmean"][0]), abs(r["var"][i, 0] - ref["fvar"][0]) / abs(ref["fvar"][0]),
                  abs(r["nlml"][i] - ref["nlml"]) / abs(ref["nlml"])]
          cond_ok = sn[i] > 0 or bad < 0 or (i % B) != (bad % B)
          e = max(errs)
          if cond_ok:
              worst = max(worst, e)
          if r["info"][i] != 0 or not (e <= (1e-8 if cond_ok else 1e-2)):
              if cond_ok or r["info"][i] == 0:
                  fails += 1
                  print("FAIL batch case %d fit %d: n=%d d=%d B=%d F=%d G=%d W=%d %s %s info=%d err %.3e" % (case, i, n, d, B, F, G, W, kind, opts, r["info"][i], e), flush=True)
  if verbose:
    print("%d batch cases in %.1f s, %d failures; worst relative error %.1e" % (cases, time.time() - t0, fails, worst))
  return fails, worst


if __name__ == "__main__":
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    f, _ = run(nc, sd)
    f2, _ = run_batches(max(1, nc // 5), sd + 1)
    sys.exit(1 if (f or f2) else 0)
