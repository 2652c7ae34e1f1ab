"""Multi-GPU.  Independent fits: the retrospective (year x grid-point) list is dealt round-robin over
ranks, every rank factorises its share on its own GPU with no data-path collective, and the few floats per fit
(sigma_f, nlml, info, mean, var) are all-gathered at the end over torch.distributed (backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests).  One large fit: ``DistributedGPR`` over the library's own sharded fit (sigp_dist_*)."""
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


def tcp_exchange_id(rank, world, addr="127.0.0.1", port=29555, make_id=None, timeout=120.0):
    """Torch-free rendezvous for the 128-byte ncclUniqueId of ``sigp_dist_init``: rank 0 creates it (``make_id()``) and
    serves it on (addr, port); every other rank connects and reads it.  Any other channel (MPI, a file, torch.distributed's
    store) does as well -- the library only needs the same 128 bytes on every rank."""
    import socket
    import time
    if rank == 0:
        uid = make_id()
        with socket.socket() as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                c, _a = srv.accept()
                with c:
                    c.sendall(uid)
        return uid
    t0 = time.time()
    while True:
        try:
            with socket.create_connection((addr, port), timeout=timeout) as c:
                buf = b""
                while len(buf) < 128:
                    chunk = c.recv(128 - len(buf))
                    if not chunk:
                        raise ConnectionError("rendezvous: short read")
                    buf += chunk
                return buf
        except (ConnectionRefusedError, ConnectionError):
            if time.time() - t0 > timeout:
                raise
            time.sleep(0.05)


class DistributedGPR:
    """One large fit sharded over the GPUs of a node (BASELINE configs[3] fp64, configs[4] fp32 + fp64 refinement):
    a thin wrapper over the library's own sharded fit (include/sigp.h: ``sigp_dist_init`` / ``sigp_dist_fit``) -- 1-D
    block-cyclic ownership of outer panels (``outer_blocks`` x 128 columns), owner-only storage (per-rank matrix bytes ~
    1/world), the block-row panel broadcast on the library's RCCL communicator with look-ahead, stream-ordered with events
    (no host synchronisation per panel).  ``dtype="f32"``: the fp32 factor sharded the same way, triangular solves on the
    distributed factor and the fp64 refinement residual sharded by rows.

    ``dist`` selects how the ranks find each other and what moves the panels:

    * ``None`` with ``world == 1``: nothing to move.
    * an initialised ``torch.distributed`` with backend "nccl": the LIBRARY's RCCL communicator (one device per rank); torch only
      carries the 128-byte unique id to the other ranks.
    * an initialised ``torch.distributed`` with another backend (gloo): a host-pointer transport (``sigp_dist_init_transport``) --
      the library stages panels through pinned memory and calls back into ``dist.broadcast`` / ``dist.all_reduce``.  This is
      how several ranks rehearse on ONE GPU (RCCL needs a device per rank); same panel loop, same arithmetic.
    * a callable ``exchange(uid_or_None) -> uid``: the library's RCCL communicator with any other rendezvous, e.g.
      ``lambda mk: tcp_exchange_id(rank, world, port=..., make_id=mk)`` (no torch in the process).

    Same call sites as ``GPR``: ``fit`` (north/June1st.py:264-271) and ``predict`` (:272-277); the factor stays spread over
    the ranks, so predictions exist for the points passed to ``fit(Xs=...)`` (they ride along the factorisation)."""

    def __init__(self, kernel, rank, world, dist, device=0, outer_blocks=8, lookahead=True, dtype="f64", owner_only=True, stats=False, force_rccl=False,
                 panel_split=False):
        from .gpr import GPR
        self.rank, self.world, self.dist = int(rank), int(world), dist
        self.W = int(outer_blocks)
        self.lookahead = bool(lookahead)
        self.dtype = dtype
        self.owner_only = True                 # storage is always owner-only now (the replicated form of rounds 1-2 is gone)
        self.gp = GPR(kernel=kernel, device=device, dtype=dtype)
        self.gp.set_option("owner_only", 1)
        if stats:
            self.gp.set_option("dist_stats", 1)
        if panel_split:                        # panel exchange by row pieces + all-gather (sigp.h: dist_panel_split); bit-identical results
            self.gp.set_option("dist_panel_split", 1)
        self.device = device
        self.transport = "none"
        self._force_rccl = bool(force_rccl)    # world == 1 only: open a one-rank RCCL communicator anyway (exercises the RCCL path on one GPU)
        self._cb = None
        self._own_ride = None
        self._init_transport()

    # ---- transports ------------------------------------------------------------------------------------------------
    def _init_transport(self):
        from . import _lib as L
        gp, lib = self.gp, self.gp._lib
        if self.world == 1:
            if self._force_rccl:
                self._rccl_init(self._make_id())
            else:
                gp._check(lib.sigp_dist_init(gp._h, 1, 0, None), "dist_init")
            return
        dist = self.dist
        if dist is None:
            raise ValueError("DistributedGPR: world > 1 needs `dist` (torch.distributed, or a callable that exchanges the unique id)")
        if callable(dist) and not hasattr(dist, "broadcast"):
            uid = dist(self._make_id if self.rank == 0 else None)
            self._rccl_init(uid)
            return
        backend = dist.get_backend()
        if backend == "nccl":
            import torch
            t = torch.zeros(128, dtype=torch.uint8, device="cuda:%d" % self.device)
            if self.rank == 0:
                t.copy_(torch.frombuffer(bytearray(self._make_id()), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            self._rccl_init(bytes(t.cpu().numpy().tobytes()))
            return
        self._host_transport(dist)

    def _make_id(self):
        buf = C.create_string_buffer(128)
        rc = self.gp._lib.sigp_dist_unique_id(buf)
        if rc != 0:
            from . import _lib as L
            raise L.SigpError("sigp_dist_unique_id failed (rc=%d): librccl could not be bound" % rc)
        return buf.raw

    def _rccl_init(self, uid):
        gp = self.gp
        buf = C.create_string_buffer(bytes(uid), 128)
        gp._check(gp._lib.sigp_dist_init(gp._h, self.world, self.rank, buf), "dist_init")
        self.transport = "rccl"

    def _host_transport(self, dist):
        import torch
        from . import _lib as L

        def as_tensor(buf, count, np_dtype):
            arr = np.ctypeslib.as_array((C.c_uint8 * (count * np.dtype(np_dtype).itemsize)).from_address(buf)).view(np_dtype)
            return torch.from_numpy(arr)

        def bcast(ctx, buf, nbytes, root, stream):
            try:
                dist.broadcast(as_tensor(buf, int(nbytes), np.uint8), src=int(root))
                return 0
            except Exception:            # noqa: BLE001 -- an exception must not unwind through the C frames
                return 1

        def allreduce(ctx, buf, count, is_f32, op, stream):
            try:
                t = as_tensor(buf, int(count), np.float32 if is_f32 else np.float64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)
                return 0
            except Exception:            # noqa: BLE001
                return 1

        world, rank = self.world, self.rank

        def scatter(ctx, buf, chunk, root, stream):          # in place: the root's chunk r -> rank r at buf + r chunk
            try:
                full = as_tensor(buf, int(chunk) * world, np.uint8)
                parts = [full[r * int(chunk):(r + 1) * int(chunk)] for r in range(world)]
                mine = torch.empty(int(chunk), dtype=torch.uint8)
                dist.scatter(mine, [q.clone() for q in parts] if rank == int(root) else None, src=int(root))
                if rank != int(root):
                    parts[rank].copy_(mine)
                return 0
            except Exception:            # noqa: BLE001
                return 1

        def allgather(ctx, buf, chunk, stream):              # in place: chunk r from rank r
            try:
                full = as_tensor(buf, int(chunk) * world, np.uint8)
                parts = [full[r * int(chunk):(r + 1) * int(chunk)] for r in range(world)]
                got = [torch.empty(int(chunk), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(got, parts[rank].clone())
                for r in range(world):
                    if r != rank:
                        parts[r].copy_(got[r])
                return 0
            except Exception:            # noqa: BLE001
                return 1

        self._cb = (L.BCAST_FN(bcast), L.ALLREDUCE_FN(allreduce), L.SCATTER_FN(scatter), L.ALLGATHER_FN(allgather))
        self._tr = L.Transport(None, 0, self._cb[0], self._cb[1], self._cb[2], self._cb[3])
        gp = self.gp
        gp._check(gp._lib.sigp_dist_init_transport2(gp._h, self.world, self.rank, C.byref(self._tr), C.sizeof(self._tr)), "dist_init_transport")
        self.transport = "host"

    def close(self):
        if getattr(self, "gp", None) is not None and getattr(self.gp, "_h", None):
            self.gp._lib.sigp_dist_shutdown(self.gp._h)
        self.gp.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def owner(self, panel_index):
        return panel_index % self.world

    def fit(self, X, y, ell, sn_tilde, M=None, Xs=None):
        """One call into ``sigp_dist_fit``: build of the own block columns, the panel loop with its broadcasts, the reductions
        (all-reduced), and for the fp32 engine the sharded solves + refinement.  Results are identical on every rank; with and
        without ``lookahead`` the arithmetic is the same (bit-identical results)."""
        from . import _lib as L
        from .gpr import LinAlgError
        gp, lib = self.gp, self.gp._lib
        gp.set_data(X, y, M=M, Xs=Xs)          # X, y replicated on every rank (n*d*8 bytes)
        if Xs is not None and gp._ride is None:
            raise ValueError("sharded fit: at most %d ride-along test points" % (L.MAX_RIDE if self.dtype == "f64" else 3))
        return self.refit(ell, sn_tilde)

    def refit(self, ell, sn_tilde):
        """The sharded fit again on the staged data with new hyper-parameters (every rank calls it)."""
        from . import _lib as L
        from .gpr import LinAlgError
        gp, lib = self.gp, self.gp._lib
        if not gp._has_data:
            raise RuntimeError("refit: no data staged; call fit() first")
        Sig = None
        if gp.kernel == "netdiffusion":
            Sig = L.f64(gp._sigma(float(ell)), 2)
        out = np.zeros(4)
        m = 0 if gp._ride is None else gp._ride.shape[0]
        mean, var = np.zeros(max(m, 1)), np.zeros(max(m, 1))
        rc = lib.sigp_dist_fit(gp._h, gp._kid, float(ell), float(sn_tilde), L.ptr(Sig), 0 if Sig is None else Sig.shape[1], self.W,
                               int(self.lookahead), L.ptr(out), L.ptr(mean), L.ptr(var))
        gp.info_ = int(out[2]) if rc in (L.OK, L.NOT_SPD) else -1
        if rc == L.NOT_SPD:
            raise LinAlgError("Matrix is not positive definite (pivot %d)" % gp.info_, gp.info_)
        gp._check(rc, "dist_fit")
        gp.sigma_f_, gp.nlml_, gp.sigma_n_ = float(out[0]), float(out[1]), float(out[3])
        gp._ride_mean, gp._ride_var = mean[:m].copy(), var[:m].copy()
        gp._fitted = False
        self._own_ride = (None if gp._ride is None else gp._ride.copy(), gp._ride_mean, gp._ride_var)
        self.sigma_f_, self.nlml_, self.sigma_n_ = gp.sigma_f_, gp.nlml_, gp.sigma_n_
        return self

    @property
    def matrix_bytes_(self):
        """Device bytes this rank holds in matrix / factor buffers (~ 1/world of the single-GPU matrix)."""
        return self.gp.matrix_bytes_

    @property
    def refine_residual_(self):
        return self.gp.refine_residual_

    def stats(self):
        """``dist_*`` statistics of the last fit (meaningful with ``stats=True``): ms and bytes on THIS rank."""
        return {k: self.gp._stat("dist_" + k) for k in ("fit_ms", "factor_ms", "bcast_bytes", "comm_ms", "stall_ms", "solve_ms", "collectives", "comm_ranks", "host_comm_ms", "enqueue_ms",
                                                                "link_bytes", "owner_ms", "split_panels", "link_panel_max")}

    def predict(self, Xs):
        """(fmean [m], fvar [m]) as ``GPR.predict``.  The points passed to ``fit(Xs=...)`` rode along the factorisation and cost nothing;
        any other points go through ``sigp_dist_predict`` (solves on the distributed factor) -- a COLLECTIVE call: every rank must
        make it with the same ``Xs`` (RBF / Matern)."""
        from . import _lib as L
        if self._own_ride is None:
            raise RuntimeError("predict: call fit() first")
        ride, mu, var = self._own_ride
        Xs = L.f64(np.atleast_2d(np.asarray(Xs, dtype=np.float64)), 2)
        if ride is not None and Xs.shape == ride.shape and np.array_equal(Xs, ride):
            return mu.copy(), var.copy()
        if self.gp.kernel == "netdiffusion":
            raise RuntimeError("sharded fit with the reference kernel: predictions exist for the points passed to fit(Xs=...)")
        if Xs.shape[1] != self.gp.d:
            raise ValueError("Xs must have %d columns" % self.gp.d)
        m = Xs.shape[0]
        mean, v = np.zeros(m), np.zeros(m)
        self.gp._check(self.gp._lib.sigp_dist_predict(self.gp._h, L.ptr(Xs), m, Xs.shape[1], L.ptr(mean), L.ptr(v)), "dist_predict")
        return mean, v
