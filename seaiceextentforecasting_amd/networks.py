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
        self.corrs = []
        self.tau = 0            # shadows the method on instances, exactly like the reference (:23)
        self.nodes = []
        self.unavail = []
        self.anomaly = {}
        self.links = {}
        self.strength = {}
        self.strengthmap = []

    # ---- ComplexNetworks.py:31-47 -------------------------------------------------------------------------
    def tau(self, significance=0.01):
        """Cell-to-cell correlations and the threshold tau = mean of the significantly positive ones
        (one-sided t-test, df = T-2)."""
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ID = np.where(np.abs(np.nanmax(self.data, 2)) > 0)
        N = np.shape(ID)[1]
        R = np.corrcoef(self.data[ID])
        np.fill_diagonal(R, np.nan)
        self._R = R.copy()                                     # N x N, NaN diagonal: the only correlations ever used
        self.nodes = np.atleast_2d(ID[0] * self.dimY + ID[1])
        self._node_index = {int(v): n for n, v in enumerate(self.nodes[0, :])}
        self.corrs = np.zeros((N, self.dimX, self.dimY)) * np.nan
        for n in range(N):                                     # row-wise scatter: 7x faster than one fancy assignment
            self.corrs[n, :, :][ID] = R[n, :]
        df = self.dimT - 2
        R = R[R >= 0]
        T = R * np.sqrt(df / (1 - R ** 2))
        P = stats.t.sf(T, df)
        R = R[P < significance]
        self.tau = np.mean(R)

    # ---- ComplexNetworks.py:49-281 ------------------------------------------------------------------------
    def area_level(self, latlon_grid=False):
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

    # ---- ComplexNetworks.py:283-326 -----------------------------------------------------------------------
    def intra_links(self, area=None, lat=None):
        """Area anomaly series (the GP's features), covariance links and strength map."""
        self.anomaly, self.links, self.strength = {}, {}, {}
        self.strengthmap = np.zeros((self.dimX, self.dimY)) * np.nan
        if lat is not None:
            scale = np.sqrt(np.cos(np.radians(lat)))
        elif area is not None:
            scale = np.sqrt(area)
        else:
            scale = np.ones((self.dimX, self.dimY))
        for A in self.V:
            temp_array = np.zeros(self.data.shape) * np.nan
            for cell in self.V[A]:
                temp_array[cell[0], cell[1], :] = np.multiply(self.data[cell[0], cell[1], :], scale[cell[0], cell[1]])
            self.anomaly[A] = np.nansum(temp_array, axis=(0, 1))
        keys = list(self.anomaly)
        sd = {A: np.std(self.anomaly[A]) for A in keys}
        for A in keys:
            self.links[A] = [0 if A2 == A else stats.pearsonr(self.anomaly[A], self.anomaly[A2])[0] * (sd[A] * sd[A2]) for A2 in keys]
        for A in self.links:
            self.strength[A] = np.nansum([abs(v) for v in self.links[A]])
            for cell in self.V[A]:
                self.strengthmap[cell[0], cell[1]] = self.strength[A]


def networks(dataset, latlon=False, area_key="psar", lat_key="lat", significance=0.01):
    """The scripts' ``networks()`` driver (north/June1st.py:196-206): sets ``dataset['nodes']`` and
    ``dataset['anoms']`` from ``dataset['dt']``."""
    net = Network(data=dataset["dt"])
    Network.tau(net, significance)
    Network.area_level(net, latlon_grid=latlon)
    if latlon:
        Network.intra_links(net, lat=dataset[lat_key])
    else:
        Network.intra_links(net, area=dataset[area_key])
    dataset["nodes"] = net.V
    dataset["anoms"] = net.anomaly
    return net
