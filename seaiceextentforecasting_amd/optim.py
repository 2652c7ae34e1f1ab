"""Lockstep quasi-Newton driver for the ``MLII`` contract (north/June1st.py:235-262): B independent 2-parameter problems
(theta = (log l, log sn~) per (region, year) data set) advance TOGETHER -- every round asks the caller for ONE trial point per
unfinished data set, i.e. one device call for the lot, whatever the individual line searches do.

The reference's call (`minimize(MLII, x0, method='CG', jac=True)`, commented out at :259-262) started from its table entries; many of
those sit on plateaus of the profiled likelihood (sn~ -> inf: K~ -> sn~ I, gradient ~ 1e-6), so the first step of a data set is a unit
step along -g (as L-BFGS-B's first step is) and the inverse Hessian is scaled by s.y / y.y before its first update.
"""
import numpy as np


def bfgs_lockstep(evaluate, theta0, maxiter=50, gtol=1e-5, ftol=1e-10, max_step=2.0):
    """``evaluate(theta [B, 2]) -> (f [B], g [B, 2])`` (+inf where the fit fails).  Returns dict(x, fun, jac, nit, nfev, converged).
    BFGS on the 2-vector per data set, Armijo backtracking; finished data sets ride along at their optimum so that every round is
    one call of the same shape."""
    th = np.array(theta0, dtype=np.float64, copy=True)
    B = th.shape[0]
    f, g = evaluate(th)
    f, g = np.array(f, dtype=np.float64), np.array(g, dtype=np.float64)
    nfev = 1
    H = np.tile(np.eye(2), (B, 1, 1))
    fresh = np.ones(B, dtype=bool)                 # H is still the identity: unit first step, scale before the first update
    done = ~np.isfinite(f) | (np.max(np.abs(g), axis=1) <= gtol)
    nit = np.zeros(B, dtype=np.int64)
    small = np.zeros(B, dtype=np.int64)            # consecutive accepted steps with a negligible decrease
    step = np.ones(B)
    direction = np.zeros((B, 2))
    trial = th.copy()
    new_dir = np.ones(B, dtype=bool)
    for _ in range(maxiter * 8):
        act = np.flatnonzero(~done)
        if act.size == 0:
            break
        for b in act:
            if new_dir[b]:
                p = -H[b] @ g[b]
                if p @ g[b] >= 0:                     # not a descent direction: reset the inverse Hessian
                    H[b] = np.eye(2); p = -g[b]; fresh[b] = True
                nrm = np.linalg.norm(p)
                if fresh[b] and nrm > 0:
                    p = p / nrm                       # first step of this data set: unit length in log space
                elif nrm > max_step:                  # log-space steps of at most max_step
                    p = p * (max_step / nrm)
                direction[b] = p; step[b] = 1.0; new_dir[b] = False
            trial[b] = th[b] + step[b] * direction[b]
        trial[done] = th[done]
        fa, ga = evaluate(trial)
        nfev += 1
        for b in act:
            slope = g[b] @ direction[b]
            if np.isfinite(fa[b]) and fa[b] <= f[b] + 1e-4 * step[b] * slope:
                s_ = trial[b] - th[b]; yv = ga[b] - g[b]
                df = f[b] - fa[b]
                th[b], f[b], g[b] = trial[b].copy(), fa[b], ga[b]
                nit[b] += 1
                sy = s_ @ yv
                if sy > 1e-14 * max(1.0, np.linalg.norm(s_) * np.linalg.norm(yv)):
                    if fresh[b]:
                        H[b] = np.eye(2) * (sy / (yv @ yv)); fresh[b] = False
                    rho = 1.0 / sy
                    V = np.eye(2) - rho * np.outer(s_, yv)
                    H[b] = V @ H[b] @ V.T + rho * np.outer(s_, s_)
                new_dir[b] = True
                small[b] = small[b] + 1 if df <= ftol * max(1.0, abs(f[b])) else 0
                if np.max(np.abs(g[b])) <= gtol or small[b] >= 2 or nit[b] >= maxiter:
                    done[b] = True
            else:
                step[b] *= 0.5
                if step[b] < 1e-8:
                    done[b] = True
    conv = np.array([np.isfinite(f[b]) and np.max(np.abs(g[b])) <= max(gtol, 1e-3 * max(1.0, abs(f[b]))) for b in range(B)])
    return dict(x=th, fun=f, nit=nit, nfev=nfev, converged=conv, jac=g)


def newton_lockstep(evaluate, theta0, maxiter=40, gtol=1e-5, ftol=1e-12, max_step=2.0, alphas=(1.0, 2.0, 0.5, 0.125), h=1e-4):
    """Lockstep Newton for evaluators whose extra points are free (the one-workgroup-per-fit kernel: a round of 12 points per data set
    is still ONE launch).  Every round evaluates, per unfinished data set, ``len(alphas)`` step lengths along the current direction and,
    next to each candidate, its two forward-difference neighbours theta_c + h e_i -- so the accepted candidate arrives with its exact
    gradient AND a finite-difference Hessian of exact gradients.  Direction: modified Newton (the 2 x 2 Hessian's eigenvalues replaced by
    their magnitudes, floored), capped at ``max_step`` in log space; acceptance: the lowest candidate that satisfies Armijo; none ->
    the candidate set shrinks by 16.  ``evaluate(theta [P, 2], owner [P]) -> (f [P], g [P, 2])`` with owner[i] = data set of point i.
    Returns dict(x, fun, jac, nit, nfev = rounds, converged)."""
    th = np.array(theta0, dtype=np.float64, copy=True)
    B = th.shape[0]
    E = np.eye(2) * h
    A = len(alphas)

    def probe(points, owners):
        """f, g, Hessian at each point (three evaluations per point, one call)."""
        P = np.concatenate([points, points + E[0], points + E[1]])
        fo, go = evaluate(P, np.concatenate([owners, owners, owners]))
        n = len(points)
        f, g = np.array(fo[:n], dtype=np.float64), np.array(go[:n], dtype=np.float64)
        Hs = np.stack([(go[n:2 * n] - g) / h, (go[2 * n:] - g) / h], axis=2)      # column i = d grad / d theta_i
        return f, g, 0.5 * (Hs + np.transpose(Hs, (0, 2, 1)))

    f, g, Hs = probe(th, np.arange(B))
    nfev = 1
    done = ~np.isfinite(f) | (np.max(np.abs(g), axis=1) <= gtol)
    nit = np.zeros(B, dtype=np.int64)
    shrink = np.ones(B)
    small = np.zeros(B, dtype=np.int64)
    for _ in range(maxiter):
        act = np.flatnonzero(~done)
        if act.size == 0:
            break
        dirs = np.zeros((B, 2))
        for b in act:
            Hb = Hs[b]
            if np.all(np.isfinite(Hb)):
                w, V = np.linalg.eigh(Hb)
                w = np.maximum(np.abs(w), 1e-8 * max(1.0, np.max(np.abs(w))))
                p = -(V / w) @ (V.T @ g[b])
            else:
                p = -g[b]
            nrm = np.linalg.norm(p)
            if nrm > max_step:
                p = p * (max_step / nrm)
            dirs[b] = p
        pts = np.concatenate([th[act] + a * shrink[act, None] * dirs[act] for a in alphas])
        fc, gc, Hc = probe(pts, np.tile(act, A))
        nfev += 1
        for k, b in enumerate(act):
            slope = g[b] @ dirs[b]
            best = -1
            for a in range(A):
                i = a * len(act) + k
                if np.isfinite(fc[i]) and np.all(np.isfinite(gc[i])) and fc[i] <= f[b] + 1e-4 * alphas[a] * shrink[b] * slope and (best < 0 or fc[i] < fc[best]):
                    best = i
            if best < 0:
                shrink[b] /= 16.0
                if shrink[b] < 1e-9:
                    done[b] = True
                continue
            df = f[b] - fc[best]
            th[b], f[b], g[b], Hs[b] = pts[best], fc[best], gc[best], Hc[best]
            nit[b] += 1
            shrink[b] = min(1.0, shrink[b] * 4.0)
            small[b] = small[b] + 1 if df <= ftol * max(1.0, abs(f[b])) else 0
            if np.max(np.abs(g[b])) <= gtol or small[b] >= 2:
                done[b] = True
    conv = np.array([np.isfinite(f[b]) and np.max(np.abs(g[b])) <= max(gtol, 1e-3 * max(1.0, abs(f[b]))) for b in range(B)])
    return dict(x=th, fun=f, nit=nit, nfev=nfev, converged=conv, jac=g)
