"""Multi-GPU sharding of independent fits: the retrospective (year x grid-point) list is dealt
round-robin over ranks, every rank factorises its share on its own GPU with no data-path collective, and the
few floats per fit (sigma_f, nlml, info, mean, var) are all-gathered at the end over torch.distributed
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
import ctypes as C

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


class DistributedGPR:
    """One large fit sharded over the GPUs of a node (BASELINE configs[3]): 1-D block-cyclic ownership
    of outer panels (``outer_blocks`` x 128 columns).  Per panel: the owner factors it (diagonal blocks, panel
    solve, panel-internal updates), the factored panel -- rows below it, the ride-along rows and the inverse
    diagonal blocks -- is broadcast (``torch.distributed``: RCCL over xGMI with backend "nccl"), and every rank
    applies the rank-K update to the panels it owns.  After the last panel every rank holds the whole factor and
    the solved ride rows, so sigma_f / nlML / predictions are formed locally with no further collective.

    ``dtype="f32"`` shards the fp32 factorisation of the mixed-precision engine the same way (panels travel as fp32);
    the fp64 iterative refinement then runs replicated on every rank against its complete copy of the factor.

    Same call sites as ``GPR``: ``fit`` (north/June1st.py:264-271) and ``predict`` (:272-277)."""

    def __init__(self, kernel, rank, world, dist, device=0, outer_blocks=4, lookahead=True, dtype="f64", owner_only=False):
        from .gpr import GPR
        import torch
        self._torch = torch
        self.rank, self.world, self.dist = int(rank), int(world), dist
        self.W = int(outer_blocks)
        self.lookahead = bool(lookahead)
        self.dtype = dtype                     # "f32": fp32 factor sharded the same way, fp64 refinement replicated on every rank (configs[4])
        # owner_only: a rank allocates, builds and updates only the block columns of its own panels (per-rank matrix bytes
        # ~ 1/world; received panels are applied straight out of the receive buffer; ride-row reductions are all-reduced).
        # The factor then stays spread over the ranks: predictions exist for the ride-along points only.
        self.owner_only = bool(owner_only)
        if self.owner_only and dtype != "f64":
            raise ValueError("owner_only sharding is the fp64 engine's (the fp32 refinement needs the whole factor on every rank)")
        self.gp = GPR(kernel=kernel, device=device, dtype=dtype)
        if self.owner_only:
            self.gp.set_option("owner_only", 1)
        self.device = device
        self._bufs = [None, None]

    def close(self):
        self.gp.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def owner(self, panel_index):
        return panel_index % self.world

    def fit(self, X, y, ell, sn_tilde, M=None, Xs=None):
        """With ``lookahead`` (default) the owner of panel p+1 updates that panel's columns first, factors it and
        starts its broadcast while every rank is still applying panel p to the rest of its columns, so the xGMI
        transfer and the owner's latency chain hide behind the trailing update.  Without it the steps
        run strictly one after the other.  Both orders do the same arithmetic: results are bit-identical."""
        if self.owner_only:
            return self._fit_owner_only(X, y, ell, sn_tilde, M, Xs)
        from . import _lib as L
        from .gpr import LinAlgError
        torch, gp, lib = self._torch, self.gp, self.gp._lib
        gp.set_data(X, y, M=M, Xs=Xs)          # X, y replicated on every rank (n*d*8 bytes)
        gp.build(ell, sn_tilde)                # every rank builds K~; it only ever updates the panels it owns
        gp._check(lib.sigp_dist_begin(gp._h), "dist_begin")
        gp.set_option("dist_async", 1 if self.lookahead else 0)
        T = int(lib.sigp_num_blocks(gp._h))
        panels = [(J, min(self.W, T - J)) for J in range(0, T, self.W)]
        P = len(panels)
        nelem = [int(lib.sigp_dist_panel_elems(gp._h, J, Wc)) for J, Wc in panels]
        nmax = max(nelem) + 1
        for k in range(2):                     # two broadcast buffers: panel p+1 is received while panel p is in use
            if self._bufs[k] is None or self._bufs[k].numel() < nmax:
                self._bufs[k] = torch.empty(nmax, dtype=torch.float64 if self.dtype == "f64" else torch.float32, device="cuda:%d" % self.device)

        def view(p):
            return self._bufs[p % 2][:nelem[p] + 1]

        def factor_and_pack(p):                # owner: panel p is up to date -> factor, pack [panel | dinv | info]
            J, Wc = panels[p]
            pinfo = C.c_int64(0)
            gp._check(lib.sigp_dist_panel_factor(gp._h, J, Wc, C.byref(pinfo)), "dist_panel_factor")
            gp._check(lib.sigp_dist_sync(gp._h, 1), "dist_sync")          # the buffer's previous panel is unpacked
            gp._check(lib.sigp_dist_panel_pack(gp._h, J, Wc, C.c_void_p(view(p).data_ptr())), "dist_panel_pack")
            view(p)[nelem[p]] = float(pinfo.value)

        def start_bcast(p):                    # the block-row panel broadcast (RCCL over xGMI with backend "nccl")
            if self.world == 1:
                return None
            return self.dist.broadcast(view(p), src=self.owner(p), async_op=True)

        def wait_bcast(work):
            if work is not None:
                work.wait()
            torch.cuda.synchronize(self.device) if not self.lookahead else torch.cuda.current_stream(self.device).synchronize()

        def update(p, q):                      # columns of panel q -= (panel p)(panel p)^T rows
            J, Wc = panels[p]
            Jq, Wq = panels[q]
            c0 = Jq - (J + Wc)
            gp._check(lib.sigp_dist_update(gp._h, J, Wc, c0, c0 + Wq), "dist_update")

        info = 0
        try:
            if self.rank == self.owner(0):
                factor_and_pack(0)
            work = start_bcast(0)
            for p in range(P):
                J, Wc = panels[p]
                wait_bcast(work)
                work = None
                info = int(view(p)[nelem[p]].item())
                if info != 0:
                    break
                if self.rank != self.owner(p):
                    gp._check(lib.sigp_dist_panel_unpack(gp._h, J, Wc, C.c_void_p(view(p).data_ptr())), "dist_panel_unpack")
                mine = [q for q in range(p + 1, P) if self.owner(q) == self.rank]
                nxt = p + 1
                if self.lookahead and nxt < P:
                    if self.owner(nxt) == self.rank:
                        update(p, nxt)                         # the next panel's own columns first ...
                        gp._check(lib.sigp_dist_mark(gp._h), "dist_mark")
                        mine.remove(nxt)
                        for q in mine:                         # ... the rest of this rank's updates run on the update stream
                            update(p, q)
                        mine = []
                        factor_and_pack(nxt)                   # ... while the panel stream factors and packs panel p+1
                    else:
                        gp._check(lib.sigp_dist_sync(gp._h, 1), "dist_sync")   # buffer (p+1)%2 free: panel p-1 unpacked
                    work = start_bcast(nxt)
                for q in mine:
                    update(p, q)
                if not self.lookahead and nxt < P:
                    if self.owner(nxt) == self.rank:
                        factor_and_pack(nxt)
                    work = start_bcast(nxt)
            if work is not None:                   # non-SPD exit with a broadcast in flight: complete the collective
                wait_bcast(work)
        finally:
            gp._check(lib.sigp_dist_sync(gp._h, 0), "dist_sync")
            gp.set_option("dist_async", 0)
        out = np.zeros(4)
        m = 0 if gp._ride is None else gp._ride.shape[0]
        mean, var = np.zeros(max(m, 1)), np.zeros(max(m, 1))
        rc = lib.sigp_dist_finish(gp._h, info, L.ptr(out), L.ptr(mean), L.ptr(var))
        gp.info_ = info
        if rc == L.NOT_SPD:
            raise LinAlgError("Matrix is not positive definite (pivot %d)" % info, info)
        gp._check(rc, "dist_finish")
        gp.sigma_f_, gp.nlml_, gp.sigma_n_ = float(out[0]), float(out[1]), float(out[3])
        gp._ride_mean, gp._ride_var = mean[:m].copy(), var[:m].copy()
        gp._fitted = True
        self.sigma_f_, self.nlml_, self.sigma_n_ = gp.sigma_f_, gp.nlml_, gp.sigma_n_
        return self

    def _fit_owner_only(self, X, y, ell, sn_tilde, M, Xs):
        """The same panel loop on owner-only storage (include/sigp.h: sigp_dist_local_*)."""
        from . import _lib as L
        from .gpr import LinAlgError
        torch, gp, lib = self._torch, self.gp, self.gp._lib
        if Xs is None:
            raise ValueError("owner_only: pass the test points to fit(Xs=...) -- they ride along the factorisation; the factor itself stays sharded")
        gp.set_data(X, y, M=M, Xs=Xs)
        if gp._ride is None:
            raise ValueError("owner_only: at most %d ride-along test points" % L.MAX_RIDE)
        gp._check(lib.sigp_dist_local_begin(gp._h, self.W, self.world, self.rank), "dist_local_begin")
        Sig = None
        if gp.kernel == "netdiffusion":
            Sig = L.f64(gp._sigma(float(ell)), 2)
        gp._check(lib.sigp_dist_local_build(gp._h, gp._kid, float(ell), float(sn_tilde), L.ptr(Sig), 0 if Sig is None else Sig.shape[1]), "dist_local_build")
        gp.set_option("dist_async", 1 if self.lookahead else 0)
        T = int(lib.sigp_num_blocks(gp._h))
        panels = [(J, min(self.W, T - J)) for J in range(0, T, self.W)]
        P = len(panels)
        nelem = [int(lib.sigp_dist_panel_elems(gp._h, J, Wc)) for J, Wc in panels]
        nmax = max(nelem) + 1
        for k in range(2):
            if self._bufs[k] is None or self._bufs[k].numel() < nmax:
                self._bufs[k] = torch.empty(nmax, dtype=torch.float64, device="cuda:%d" % self.device)

        def view(p):
            return self._bufs[p % 2][:nelem[p] + 1]

        def factor_and_pack(p):
            pinfo = C.c_int64(0)
            gp._check(lib.sigp_dist_local_buffer_wait(gp._h, p % 2), "dist_local_buffer_wait")    # updates with panel p-2 are done with the buffer
            gp._check(lib.sigp_dist_local_factor(gp._h, p, C.c_void_p(view(p).data_ptr()), C.byref(pinfo)), "dist_local_factor")
            view(p)[nelem[p]] = float(pinfo.value)

        def start_bcast(p):
            if self.world == 1:
                return None
            if self.rank != self.owner(p):
                gp._check(lib.sigp_dist_local_buffer_wait(gp._h, p % 2), "dist_local_buffer_wait")
            return self.dist.broadcast(view(p), src=self.owner(p), async_op=True)

        def wait_bcast(work):
            if work is not None:
                work.wait()
            torch.cuda.synchronize(self.device) if not self.lookahead else torch.cuda.current_stream(self.device).synchronize()

        def update(p, q):
            gp._check(lib.sigp_dist_local_update(gp._h, p, C.c_void_p(view(p).data_ptr()), q, p % 2), "dist_local_update")

        info = 0
        try:
            if self.rank == self.owner(0):
                factor_and_pack(0)
            work = start_bcast(0)
            for p in range(P):
                wait_bcast(work)
                work = None
                info = int(view(p)[nelem[p]].item())
                if info != 0:
                    break
                mine = [q for q in range(p + 1, P) if self.owner(q) == self.rank]
                nxt = p + 1
                if self.lookahead and nxt < P:
                    if self.owner(nxt) == self.rank:
                        update(p, nxt)                         # the next panel's own columns first ...
                        gp._check(lib.sigp_dist_mark(gp._h), "dist_mark")
                        mine.remove(nxt)
                        for q in mine:
                            update(p, q)
                        mine = []
                        factor_and_pack(nxt)                   # ... the panel stream factors and packs panel p+1 meanwhile
                    work = start_bcast(nxt)
                for q in mine:
                    update(p, q)
                if not self.lookahead and nxt < P:
                    if self.owner(nxt) == self.rank:
                        factor_and_pack(nxt)
                    work = start_bcast(nxt)
            if work is not None:
                wait_bcast(work)
        finally:
            gp._check(lib.sigp_dist_sync(gp._h, 0), "dist_sync")
            gp.set_option("dist_async", 0)
        res = np.zeros(512)
        gp._check(lib.sigp_dist_local_reduce(gp._h, L.ptr(res)), "dist_local_reduce")
        if self.world > 1:                                      # the one exchange besides the panel broadcast: 512 doubles
            t = torch.from_numpy(res)
            if self.dist.get_backend() == "nccl":
                t = t.cuda(self.device)
            self.dist.all_reduce(t)
            res = t.cpu().numpy().copy()
        out = np.zeros(4)
        m = gp._ride.shape[0]
        mean, var = np.zeros(max(m, 1)), np.zeros(max(m, 1))
        rc = lib.sigp_dist_local_results(gp._h, L.ptr(res), info, L.ptr(out), L.ptr(mean), L.ptr(var))
        gp.info_ = info
        if rc == L.NOT_SPD:
            raise LinAlgError("Matrix is not positive definite (pivot %d)" % info, info)
        gp._check(rc, "dist_local_results")
        gp.sigma_f_, gp.nlml_, gp.sigma_n_ = float(out[0]), float(out[1]), float(out[3])
        gp._ride_mean, gp._ride_var = mean[:m].copy(), var[:m].copy()
        gp._fitted = False
        self._own_ride = (gp._ride.copy(), gp._ride_mean, gp._ride_var)
        self.sigma_f_, self.nlml_, self.sigma_n_ = gp.sigma_f_, gp.nlml_, gp.sigma_n_
        return self

    @property
    def matrix_bytes_(self):
        """Device bytes this rank holds in matrix / factor buffers (owner_only: ~ 1/world of the replicated form)."""
        return self.gp.matrix_bytes_

    def predict(self, Xs):
        if self.owner_only:
            ride, mu, var = self._own_ride
            Xs = np.atleast_2d(np.asarray(Xs, dtype=np.float64))
            if Xs.shape != ride.shape or not np.array_equal(Xs, ride):
                raise RuntimeError("owner_only: the factor is spread over the ranks; predictions exist for the points passed to fit(Xs=...)")
            return mu.copy(), var.copy()
        return self.gp.predict(Xs)
