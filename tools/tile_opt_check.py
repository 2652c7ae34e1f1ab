"""An engine option that only changes the trailing update's tile shape / walk must leave every result bit-identical:
    python tools/tile_opt_check.py name value_fp64 value_fp32"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
name, v64, v32 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ok = True
for dtype, kern, val in (("f64", "rbf", v64), ("f32", "matern52", v32)):
    for n, W in ((4100, 2), (2300, 4), (6000, 8)):
        X, y, Xs = O.synthetic_problem(n, 8, 7 + n, m=1)
        res = []
        for v in (0, val):
            with GPR(kernel=kern, outer_blocks=W, dtype=dtype) as gp:
                gp.set_option("small_tile_threshold", 0)
                gp.set_option(name, v)
                gp.fit(X, y, np.sqrt(8.0), 1e-1, Xs=Xs)
                mu, var = gp.predict(Xs)
                res.append((gp.nlml_, gp.sigma_f_, mu, var))
        same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
        ok &= same
        print(dtype, n, W, "bit-identical:", same)
sys.exit(0 if ok else 1)
