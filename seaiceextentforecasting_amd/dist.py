"""Multi-GPU sharding of independent fits: the retrospective (year x grid-point) list is dealt
round-robin over ranks, every rank factorises its share on its own GPU with no data-path collective, and the
few floats per fit (sigma_f, nlml, info, mean, var) are all-gathered at the end over torch.distributed
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
import numpy as np


def shard_indices(n_items, rank, world):
    """Round-robin share of rank `rank`: items rank, rank+world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return np.arange(rank, n_items, world)


def gather_results(local, n_items, rank, world, dist=None):
    """All-gather per-fit result arrays.  ``local`` maps name -> array whose leading axis follows
    shard_indices(n_items, rank, world).  Returns name -> array over all n_items (on every rank)."""
    idx = shard_indices(n_items, rank, world)
    out = {}
    if world == 1 or dist is None:
        for k, v in local.items():
            full = np.zeros((n_items,) + np.asarray(v).shape[1:], dtype=np.asarray(v).dtype)
            full[idx] = v
            out[k] = full
        return out
    import torch
    per = (n_items + world - 1) // world            # pad every rank's share to the same length
    for k in sorted(local):
        v = np.asarray(local[k])
        pad = np.zeros((per,) + v.shape[1:], dtype=np.float64)
        pad[:len(idx)] = v
        t = torch.from_numpy(pad)
        backend = dist.get_backend()
        if backend == "nccl":
            t = t.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        full = np.zeros((n_items,) + v.shape[1:], dtype=np.float64)
        for r in range(world):
            ridx = shard_indices(n_items, r, world)
            full[ridx] = parts[r].cpu().numpy()[:len(ridx)]
        out[k] = full.astype(v.dtype) if v.dtype != np.float64 else full
    return out


def fit_batch_sharded(engine, X, y, Xs, ell, sn_tilde, rank, world, dist=None):
    """Run the fits (fit i uses data set i % B) sharded over ranks.  ``engine(Xb, yb, Xsb, ell, sn)`` is the
    per-rank batch engine (``GPR.fit_batch`` bound to this rank's GPU) and returns the dict GPR.fit_batch returns."""
    X = np.asarray(X); y = np.asarray(y)
    ell = np.atleast_1d(ell); sn_tilde = np.atleast_1d(sn_tilde)
    F = len(ell)
    B = X.shape[0] if X.ndim == 3 else 1
    mine = shard_indices(F, rank, world)
    if X.ndim == 3:      # per-fit data sets: gather this rank's data sets explicitly so fit j of the shard uses set j
        ds = mine % B
        Xl, yl = X[ds], y[ds]
        Xsl = None if Xs is None else np.asarray(Xs)[ds]
    else:
        Xl, yl, Xsl = X, y, Xs
    r = engine(Xl, yl, Xsl, ell[mine], sn_tilde[mine]) if len(mine) else dict(
        sigma_f=np.zeros(0), nlml=np.zeros(0), info=np.zeros(0, np.int64), sigma_n=np.zeros(0),
        mean=np.zeros((0, 0 if Xs is None else np.asarray(Xs).shape[-2])), var=np.zeros((0, 0 if Xs is None else np.asarray(Xs).shape[-2])))
    return gather_results(r, F, rank, world, dist)
