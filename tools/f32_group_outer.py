"""configs[4]'s shape in a lockstep group of 4 under outer panel widths 8 / 12 / 16 (ms per fit, results compared with width 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.gp_oracle import synthetic_problem
from seaiceextentforecasting_amd import GPR
n4, d4, G4 = 32768, 32, 4
Xb = np.zeros((G4, n4, d4)); yb = np.zeros((G4, n4)); Xsb = np.zeros((G4, 1, d4))
for b in range(G4):
    Xb[b], yb[b], Xsb[b] = synthetic_problem(n4, d4, 20240004 + b, m=1)
e4 = np.full(G4, np.sqrt(d4)); s4 = np.full(G4, 1e-1)
base = None
for W in (8, 16, 12, 8):
    with GPR(kernel="matern52", dtype="f32", outer_blocks=W) as g:
        g.upload_batch(Xb, yb, Xsb, group=G4, concurrency=1)
        g.run_batch(0, G4, e4, s4, concurrency=1, group=G4)
        g.synchronize(); t0 = time.perf_counter()
        r4 = g.run_batch(0, G4, e4, s4, concurrency=1, group=G4)
        g.synchronize(); dt = (time.perf_counter() - t0) / G4
    if base is None:
        base = r4
    print("outer %2d: %.2f ms per fit; nlml rel diff vs width 8 %.1e, mean %.1e" % (W, 1e3 * dt, np.max(np.abs(r4["nlml"] - base["nlml"]) / np.abs(base["nlml"])),
          np.max(np.abs(r4["mean"] - base["mean"]) / np.abs(base["mean"]))), flush=True)
