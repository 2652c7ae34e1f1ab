import os
import sys

# the boxes report 256 logical CPUs but schedule far fewer: 64 spinning OpenBLAS threads on tiny matrices only add
# latency (and are a suspect for a rare multi-minute stall); the oracle's matrices here are small
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
os.environ.setdefault("OMP_NUM_THREADS", "8")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Rebuild the reference-side inputs and captured records from tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    keys = list(z.keys())
    g = dict(script=str(z["meta/script"]), kind=str(z["meta/kind"]), args=[int(a) for a in z["meta/args"]],
             SIC={}, SST=None, SIEs_dt={}, SIEs_trend={}, records=[], GPR={})
    for prefix in ("SIC", "SST"):
        dd = {}
        for k in keys:
            if k.startswith(prefix + "/") and k.endswith("/ids"):
                sub = k.split("/")[1]
                ids = z[k]
                data = z["%s/%s/data" % (prefix, sub)]
                dd[sub] = {int(i): data[j].copy() for j, i in enumerate(ids)}
        if dd:
            g[prefix] = dd
    for k in keys:
        if k.startswith("SIEs_dt/"):
            g["SIEs_dt"][k.split("/", 1)[1]] = z[k]
        elif k.startswith("SIEs_trend/"):
            g["SIEs_trend"][k.split("/", 1)[1]] = z[k]
        elif k.startswith("GPR/"):
            g["GPR"][k.split("/", 1)[1]] = z[k]
    for j in range(int(z["meta/nrec"])):
        pre = "rec%02d/" % j
        g["records"].append({k[len(pre):]: z[k] for k in keys if k.startswith(pre)})
    return g


GOLDEN_NAMES = sorted(f[:-4] for f in os.listdir(GOLDEN)
                      if f.endswith(".npz") and f != "callers.npz" and not f.startswith(("networks_", "config")))


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)
