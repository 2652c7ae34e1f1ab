import sys, os, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
from conftest import load_golden
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
g = load_golden(sys.argv[1] if len(sys.argv) > 1 else "north_July")
for r in g["records"][:2]:
    gp = GPR(kernel="netdiffusion")
    gp.set_data(r["X"], r["y"], M=r["M"])
    for th, nl, gr in zip(r["mlii_theta"], r["mlii_nlml"], r["mlii_grad"]):
        t = time.time(); print("theta", th, "ref", nl, gr, flush=True)
        val, grad = gp.nlml(th, grad="ref")
        print("   ->", val, grad, "%.2fs" % (time.time() - t), flush=True)
    gp.close()
