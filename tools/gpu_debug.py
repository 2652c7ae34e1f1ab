"""First-light diagnostics on the GPU box: stage-by-stage errors against the oracle (test infrastructure)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR, LinAlgError

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))

def stage(n, d, kind, m=3, seed=1, outer=None, la=None):
    X, y, Xs = O.synthetic_problem(n, d, seed, m=m)
    ell, sn = np.sqrt(d), 1e-2
    ref = O.fit_predict(X, y, Xs, ell, sn, kind=kind, M=None, ref_idiom=False)
    gp = GPR(kernel=kind, outer_blocks=outer, lookahead=la)
    gp.set_data(X, y, Xs=Xs)
    K = gp.kernel_matrix(ell, sn)
    eK = rel(K, np.tril(ref["K_tilde"]))
    try:
        gp.refit(ell, sn)
    except LinAlgError as e:
        print("n=%d %s: NOT SPD info=%s  eK=%.2e" % (n, kind, e.info, eK)); gp.close(); return
    Lt = gp.L_tilde_
    eL = rel(Lt, ref["L_tilde"])
    res = rel(Lt @ Lt.T, ref["K_tilde"])
    mu, var = gp.predict(Xs)
    Xs2 = np.random.default_rng(5).standard_normal((5, d))
    ref2 = O.fit_predict(X, y, Xs2, ell, sn, kind=kind, M=ref["M"], ref_idiom=False)
    mu2, var2 = gp.predict(Xs2)
    al = gp.alpha_
    print("n=%5d d=%2d %-12s outer=%s la=%s | K %.1e  L %.1e  LLt %.1e | sf %.1e nlml %.1e | ride mean %.1e var %.1e | gen mean %.1e var %.1e | alpha %.1e"
          % (n, d, kind, outer, la, eK, eL, res, rel(gp.sigma_f_, ref["sigma_f"]), rel(gp.nlml_, ref["nlml"]),
             rel(mu, ref["fmean"]), rel(var, ref["fvar"]), rel(mu2, ref2["fmean"]), rel(var2, ref2["fvar"]), rel(al, ref["alpha"])))
    gp.close()

def batch_check(n=640, d=8, B=5, F=11, group=4, conc=2):
    Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 2, d))
    for b in range(B):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 100 + b, m=2)
    ell = np.sqrt(d) * np.logspace(-0.3, 0.3, F); sn = np.logspace(-3, -1, F)
    gp = GPR(kernel="rbf")
    r = gp.fit_batch(Xb, yb, Xsb, ell, sn, concurrency=conc, group=group)
    worst = 0
    for i in range(F):
        b = i % B
        ref = O.fit_predict(Xb[b], yb[b], Xsb[b], ell[i], sn[i], kind="rbf", ref_idiom=False)
        worst = max(worst, rel(r["mean"][i], ref["fmean"]), rel(r["var"][i], ref["fvar"]), rel(r["nlml"][i], ref["nlml"]), rel(r["sigma_f"][i], ref["sigma_f"]))
    print("batch n=%d B=%d F=%d group=%d conc=%d: worst rel err %.2e  info %s" % (n, B, F, group, conc, worst, r["info"].tolist()))
    gp.close()

if __name__ == "__main__":
    if "--batch" in sys.argv:
        batch_check(); batch_check(group=1, conc=3); batch_check(n=300, B=3, F=20, group=8, conc=2); batch_check(group=16, conc=1)
        sys.exit(0)
    for n in (1, 7, 64, 128, 129, 257, 640):
        for kind in ("rbf", "matern52", "netdiffusion"):
            if kind == "netdiffusion" and n < 7: continue
            stage(n, 4 if n < 64 else 8, kind)
    for outer, la in ((1, 0), (2, 0), (2, 1), (4, 1), (3, 1)):
        stage(1024, 8, "rbf", outer=outer, la=la)
    t = time.time(); stage(2048, 8, "rbf"); print("2048 wall %.1fs" % (time.time() - t))
