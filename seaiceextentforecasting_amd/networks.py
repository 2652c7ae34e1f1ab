"""Complex-network feature pipeline -- the caller that produces the GP's design matrix.

Own restatement of ``ComplexNetworks.Network`` (reference ``ComplexNetworks.py:11-326``) with the same outputs
(``tau``, ``V``, ``anomaly``, ``links``, ``strength``, ``strengthmap``) and the same call convention the forecast
scripts use (``Network.tau(net, 0.01)``, ``Network.area_level(net, latlon_grid=...)``,
``Network.intra_links(net, area=...)`` -- unbound, because the instance attribute ``tau`` shadows the method,
SURVEY App. C-2).  The greedy algorithms are the reference's, decision for decision (same candidate order, same
first-maximum tie-breaks, same NaN sentinel), but cell membership tests use sets and every correlation average is
taken from the N x N cell-correlation matrix instead of Python list scans, so realistic grids (57x57) take seconds
instead of minutes.  Parity is pinned by tests/golden/networks_*.npz (the reference module itself, imported in the
authoring container, on synthetic fields).
"""
import itertools
import operator
import warnings

import numpy as np
from scipy import stats


class Network:
    def __init__(self, data):
        """``data``: de-trended anomalies [x, y, t]; NaN = land / never-ice (ComplexNetworks.py:12-29)."""
        self.data = data
        self.dimX, self.dimY, self.dimT = self.data.shape
        self.V = {}
        self.A = {}
        self._R = None
        self._active = None
        self.tau = 0            # shadows the method on instances, exactly like the reference (:23)
        self.nodes = []
        self.unavail = []
        self.anomaly = {}
        self.links = {}
        self.strength = {}
        self.strengthmap = []

    # ---- threshold (behaviour of ComplexNetworks.py:31-47) ------------------------------------------------
    def tau(self, significance=0.01, engine=None):
        """Correlation threshold of the network: the mean of the cell-to-cell correlations that are positive and
        significant under a one-sided t-test with T-2 degrees of freedom.

        Active cells are those with any non-zero anomaly (``|nanmax| > 0``), taken in row-major order; ``nodes`` holds
        their flat indices.  Only the N x N matrix of pairwise correlations is kept (``area_level`` reads nothing
        else); the reference's N x dimX x dimY scatter of it is available on demand as ``corrs``.
        ``engine``: an optional ``GPR`` handle -- the correlation matrix and the thresholded mean are then formed on
        the GPU (``sigp_corr_tau``: one fp64 MFMA product of the standardised series + a fused reduction)."""
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                   # all-NaN (land) cells
            active = np.abs(np.nanmax(self.data, axis=2)) > 0
        rows, cols = np.nonzero(active)
        self._active = (rows, cols)
        self.nodes = np.atleast_2d(rows * self.dimY + cols)
        self._node_index = {int(flat): k for k, flat in enumerate(self.nodes[0])}
        series = self.data[rows, cols, :]                     # [N, T]
        dof = self.dimT - 2
        if engine is not None:
            self._R, self.tau = engine.corr_tau(series, dof, significance)
            return
        R = np.corrcoef(series)
        np.fill_diagonal(R, np.nan)                           # a cell is not its own neighbour
        self._R = R
        self.tau = significant_positive_mean(R, dof, significance)

    @property
    def corrs(self):
        """[N, dimX, dimY]: map of every active cell's correlation with every other cell, NaN elsewhere
        (the attribute the reference materialises in ``tau``; built here only when somebody asks for it)."""
        if self._R is None:
            return []
        maps = np.full((self._R.shape[0], self.dimX, self.dimY), np.nan)
        maps[:, self._active[0], self._active[1]] = self._R
        return maps

    # ---- ComplexNetworks.py:49-281 ------------------------------------------------------------------------
    def area_level(self, latlon_grid=False, native=True):
        """Areas of the network: greedy region growing from the threshold ``tau`` and the cell correlations, then merging of
        neighbouring areas (reference ComplexNetworks.py:49-281; sets ``V`` / ``A`` / ``unavail``).
        ``native=True`` (default): ``sigp_area_level`` of libsigp.so -- the same algorithm in C++ on flat arrays, decision for
        decision and with NumPy's own summation order in every mean, so the areas are identical; ~60x faster (host code, no
        GPU involved).  ``native=False``: the Python restatement below, which is what the reference's goldens pin."""
        if native:
            return self._area_level_native(latlon_grid)
        ids = np.where(np.isnan(self.data))
        i_nan, j_nan = int(ids[0][0]), int(ids[1][0])          # first NaN cell = out-of-bounds sentinel (:50-51)
        dimX, dimY, tau = self.dimX, self.dimY, self.tau
        node_index = self._node_index
        R = self._R
        unavail = set()                                        # the reference's list self.unavail, as a set
        unavail_list = []

        def inb(i, j):
            return 0 <= i <= dimX - 1 and 0 <= j <= dimY - 1

        def cell_neighbours(i, j):                             # gen_cell_neighbours (:53-79)
            out = []
            for (a, b), wrap in (((i - 1, j), None), ((i + 1, j), None), ((i, j - 1), (i, dimY - 1)), ((i, j + 1), (i, 0))):
                if (a, b) in unavail:
                    out.append((i_nan, j_nan))
                elif inb(a, b):
                    out.append((a, b))
                elif latlon_grid and wrap is not None:
                    out.append(wrap)
                else:
                    out.append((i_nan, j_nan))
            return out

        def corr_cell(ID, cell):                               # self.corrs[ID, cell]
            n = node_index.get(cell[0] * dimY + cell[1])
            return np.nan if n is None else R[ID, n]

        def expand(cells):                                     # expand / gen_area_neighbours / area_max_correlation (:81-149)
            while True:
                cand, seen = [], set()
                for di, dj in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    for (ci, cj) in cells:
                        a, b = ci + di, cj + dj
                        if (a, b) in unavail:
                            continue
                        nb = (a, b) if inb(a, b) else (i_nan, j_nan)
                        if nb not in seen:                     # duplicates never change the first maximum
                            seen.add(nb)
                            cand.append(nb)
                if not cand:
                    return
                idx = np.array([node_index[c[0] * dimY + c[1]] for c in cells])
                X, R_mean = [], []
                for nb in cand:
                    n = node_index.get(nb[0] * dimY + nb[1])
                    if n is None:
                        continue
                    X.append(nb)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        R_mean.append(np.nanmean(R[n, idx]))
                if not R_mean:
                    return
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    Rmax = np.nanmax(R_mean)
                if not (Rmax > tau):
                    return
                m = X[int(np.where(np.asarray(R_mean) == Rmax)[0][0])]
                cells.append(m)                                # `m not in self.unavail` is always true in the reference (:139)
                unavail.add(m)
                unavail_list.append([m[0], m[1]])

        # S T E P  1: create areas (:151-196)
        A = {}
        k = 0
        for i, j in itertools.product(range(dimX), range(dimY)):
            ID = node_index.get(i * dimY + j)
            if ID is None or (i, j) in unavail:
                continue
            nei = cell_neighbours(i, j)
            nei_corrs = [corr_cell(ID, c) for c in nei]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                nei_max = np.nanmax(nei_corrs)
            if not (nei_max > tau):
                continue
            best = nei[int(np.where(np.asarray(nei_corrs) == nei_max)[0][0])]
            if best in unavail:
                continue
            cells = [(i, j), best]
            A[k] = cells
            for c in cells:
                unavail.add(c)
                unavail_list.append([c[0], c[1]])
            expand(cells)
            k += 1
        V = A

        # S T E P  2: minimise the number of areas (:198-265)
        unavail = set()
        unavail_list = []
        while True:
            num_cells = {kk: (len(V[kk]) if V[kk][0] not in unavail else 0) for kk in V}
            if not num_cells:
                break
            max_ID = max(num_cells.items(), key=operator.itemgetter(1))[0]
            if num_cells[max_ID] == 0:
                break
            big = V[max_ID]
            big_set = set(big)
            owner = {}
            for kk in V:
                for c in V[kk]:
                    owner.setdefault(c, kk)
            Anei_Rs = {}
            unavail_neis = set()
            for Xc in big:
                nei_list = cell_neighbours(Xc[0], Xc[1])
                present = {owner[nb] for nb in nei_list if nb in owner and nb not in big_set}
                if not present:
                    continue
                for kk in V:                                   # the reference scans areas in dict order for every cell
                    if kk not in present:
                        continue
                    for nb in nei_list:
                        if nb in big_set or nb in unavail_neis or owner.get(nb) != kk:
                            continue
                        unavail_neis.update(V[kk])
                        hyp = big + V[kk]
                        idx = np.array([node_index[c[0] * dimY + c[1]] for c in hyp])
                        sub = R[np.ix_(idx, idx)]
                        R_mean = []
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            for a in range(len(hyp)):
                                R_mean.append(np.nanmean(sub[a, a + 1:]))      # pairs (cell, later cells) (:236-244)
                            if kk not in Anei_Rs:
                                Anei_Rs[kk] = np.nanmean(R_mean)
            merged = False
            if Anei_Rs:
                best_k = max(Anei_Rs.items(), key=operator.itemgetter(1))[0]
                if Anei_Rs[best_k] > tau:
                    V[max_ID] = big + V.pop(best_k)
                    merged = True
            if not merged:
                for c in big:
                    unavail.add(c)
                    unavail_list.append([c[0], c[1]])
        # the reference ends by looking up the two largest areas and raises ValueError if there is only one (:269-279)
        sizes = {kk: len(V[kk]) for kk in V}
        max_ID = max(sizes.items(), key=operator.itemgetter(1))[0]
        rest = {kk: v for kk, v in sizes.items() if kk != max_ID}
        max(rest.items(), key=operator.itemgetter(1))
        self.V = {kk: [[c[0], c[1]] for c in V[kk]] for kk in V}
        self.A = self.V
        self.unavail = unavail_list

    def _area_level_native(self, latlon_grid):
        import ctypes as C

        from . import _lib as L
        lib = L.load()
        nan_cells = np.where(np.isnan(self.data))
        cell_nan = int(nan_cells[0][0]) * self.dimY + int(nan_cells[1][0])     # first NaN cell = out-of-bounds sentinel (:50-51)
        R = np.ascontiguousarray(self._R, dtype=np.float64)
        N = R.shape[0]
        node = np.full(self.dimX * self.dimY, -1, dtype=np.int32)
        node[np.asarray(self.nodes[0], dtype=np.int64)] = np.arange(N, dtype=np.int32)
        cells = np.zeros(N, dtype=np.int32); offs = np.zeros(N + 1, dtype=np.int64); ids = np.zeros(N, dtype=np.int32)
        na = C.c_int64(0); nu = C.c_int64(0)
        ip32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        cap = 2 * N + 64
        while True:       # the `unavail` list can hold a cell more than once on a lat-lon grid: retry with the length it reports
            un = np.zeros(cap, dtype=np.int32)
            rc = lib.sigp_area_level(L.ptr(R), N, ip32(node), self.dimX, self.dimY, cell_nan, float(self.tau), int(bool(latlon_grid)), ip32(cells),
                                     offs.ctypes.data_as(C.POINTER(C.c_int64)), ip32(ids), C.byref(na), ip32(un), cap, C.byref(nu))
            if rc != L.OK:
                raise ValueError("area_level: bad argument (rc=%d)" % rc)
            if nu.value <= cap:
                break
            cap = int(nu.value)
        V = {}
        for a in range(na.value):
            cs = cells[offs[a]:offs[a + 1]]
            V[int(ids[a])] = [[int(c) // self.dimY, int(c) % self.dimY] for c in cs]
        # the reference ends by looking up the two largest areas and raises ValueError if there is only one (:269-279)
        sizes = {kk: len(v) for kk, v in V.items()}
        max_ID = max(sizes.items(), key=operator.itemgetter(1))[0]
        rest = {kk: v for kk, v in sizes.items() if kk != max_ID}
        max(rest.items(), key=operator.itemgetter(1))
        self.V = V
        self.A = self.V
        self.unavail = [[int(c) // self.dimY, int(c) % self.dimY] for c in un[:nu.value]]

    # ---- area series and links (behaviour of ComplexNetworks.py:283-326) -----------------------------------
    def intra_links(self, area=None, lat=None, engine=None):
        """Per-area anomaly series (the GP's features): the sum over an area's cells of the cell series weighted by
        sqrt(cell area) (or sqrt(cos lat) on a lat-lon grid); ``links`` = covariance between area series (0 on the
        diagonal), ``strength`` = sum of |links|, painted onto the cells as ``strengthmap``.
        ``engine``: a ``GPR`` handle -- the area sums are then formed on the GPU (``sigp_area_sums``, same additions in the
        same order: bit-identical)."""
        if lat is not None:
            weight = np.sqrt(np.cos(np.radians(lat)))
        elif area is not None:
            weight = np.sqrt(area)
        else:
            weight = np.ones((self.dimX, self.dimY))
        ids = list(self.V)
        series = np.zeros((len(ids), self.dimT))
        if engine is not None and ids:
            label = np.full((self.dimX, self.dimY), -1, dtype=np.int32)
            for row, A in enumerate(ids):
                cells = np.asarray(self.V[A], dtype=np.int64)
                label[cells[:, 0], cells[:, 1]] = row
            series = engine.area_sums(self.data, weight, label, len(ids))
        for row, A in enumerate(ids if engine is None else []):
            # unique cells in row-major order, accumulated one after the other: the same additions in the same order as
            # a NaN-padded cube reduced over its two leading axes, without building that cube per area
            flat = np.unique([c[0] * self.dimY + c[1] for c in self.V[A]])
            cx, cy = np.divmod(flat, self.dimY)
            contrib = self.data[cx, cy, :] * weight[cx, cy][:, None]
            series[row] = np.add.reduce(np.where(np.isnan(contrib), 0.0, contrib), axis=0)
        self.anomaly = {A: series[row].copy() for row, A in enumerate(ids)}
        if len(ids) > 1:
            # pearson r x sd_a x sd_b (population sd) is the population covariance
            cov = np.atleast_2d(np.cov(series, bias=True))
        else:
            cov = np.zeros((len(ids), len(ids)))
        # the reference forms a link as pearsonr(a, b)[0] * sd_a * sd_b (ComplexNetworks.py:311-318: stats.pearsonr x sdA x sdA2): for an area whose series is
        # constant that is NaN (0/0 inside pearsonr), which nansum then ignores in the strength; keep the NaN so that ``links``
        # itself is the reference's output too
        # (constancy as scipy detects it -- every value equal to the first: np.std of 0.1 repeated can be 1e-17, not 0)
        const = np.all(series == series[:, :1], axis=1)
        if np.any(const) and len(ids) > 1:
            cov = cov.copy()
            cov[const, :] = np.nan
            cov[:, const] = np.nan
        np.fill_diagonal(cov, 0.0)
        self.links = {A: [0 if j == row else cov[row, j] for j in range(len(ids))] for row, A in enumerate(ids)}
        self.strength = {A: np.nansum(np.abs(cov[row])) for row, A in enumerate(ids)}
        self.strengthmap = np.full((self.dimX, self.dimY), np.nan)
        for A in ids:
            cells = np.asarray(self.V[A], dtype=np.int64)
            self.strengthmap[cells[:, 0], cells[:, 1]] = self.strength[A]


def significant_positive_mean(R, dof, significance):
    """Mean of the entries of R that are >= 0 and whose one-sided p-value (Student t, ``dof`` degrees of freedom,
    t = r sqrt(dof / (1 - r^2))) is below ``significance``.  NaN entries (the diagonal) never qualify."""
    with np.errstate(invalid="ignore", divide="ignore"):
        positive = R[R >= 0]
        t_stat = positive * np.sqrt(dof / (1 - positive ** 2))
    significant = stats.t.sf(t_stat, dof) < significance
    return np.mean(positive[significant])


def networks(dataset, latlon=False, area_key="psar", lat_key="lat", significance=0.01, engine=None):
    """The scripts' ``networks()`` driver (north/June1st.py:196-206): sets ``dataset['nodes']`` and
    ``dataset['anoms']`` from ``dataset['dt']``.  ``engine``: a ``GPR`` handle -- the correlation matrix / threshold and the area
    sums are then formed on the GPU (``Network.tau(engine=)``, ``Network.intra_links(engine=)``); the areas come from the C++
    ``sigp_area_level`` either way."""
    net = Network(data=dataset["dt"])
    Network.tau(net, significance, engine=engine)
    Network.area_level(net, latlon_grid=latlon)
    if latlon:
        Network.intra_links(net, lat=dataset[lat_key], engine=engine)
    else:
        Network.intra_links(net, area=dataset[area_key], engine=engine)
    dataset["nodes"] = net.V
    dataset["anoms"] = net.anomaly
    return net


def networks_retro(dataset, fmin, fmax, latlon=False, area_key="psar", lat_key="lat", significance=0.01, engine=None):
    """The retro scripts' ``networks(dataset, fmin, fmax)`` (north/retrospective_forecasts/September1st_retro.py:161-169): one
    network per forecast year on ``dataset['dt_YYYY']`` -> ``dataset['nodes_YYYY']``, ``dataset['anoms_YYYY']``."""
    for year in range(fmin, fmax + 1):
        net = Network(data=dataset["dt_%d" % year])
        Network.tau(net, significance, engine=engine)
        Network.area_level(net, latlon_grid=latlon)
        if latlon:
            Network.intra_links(net, lat=dataset[lat_key], engine=engine)
        else:
            Network.intra_links(net, area=dataset[area_key], engine=engine)
        dataset["nodes_%d" % year] = net.V
        dataset["anoms_%d" % year] = net.anomaly
    return dataset
