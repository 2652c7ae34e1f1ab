#!/usr/bin/env python3
"""Goldens for the ComplexNetworks restatement: the reference module itself
(/root/reference/ComplexNetworks.py, imported in the authoring container only) on synthetic anomaly fields."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, "/root/reference")
import ComplexNetworks as CN  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def field(seed, X, Y, T, nblobs, land_frac):
    """Spatially coherent anomalies: a few smooth 'modes' with random time series + noise; NaN land cells."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
    data = 0.6 * rng.standard_normal((X, Y, T))
    for _ in range(nblobs):
        ci, cj, w = rng.uniform(0, X), rng.uniform(0, Y), rng.uniform(1.5, 3.5)
        pattern = np.exp(-((ii - ci) ** 2 + (jj - cj) ** 2) / (2 * w * w))
        data += pattern[:, :, None] * rng.standard_normal(T) * 2.0
    data -= data.mean(axis=2, keepdims=True)
    land = rng.random((X, Y)) < land_frac
    land[0, 0] = True                         # area_level needs at least one NaN cell (sentinel)
    data[land] = np.nan
    return data


def main():
    warnings.filterwarnings("ignore")
    cases = [("a", 11, 12, 10, 41, 4, 0.10, False), ("b", 12, 14, 9, 35, 5, 0.15, False), ("c", 13, 10, 12, 41, 3, 0.05, True),
             ("d", 14, 16, 15, 38, 6, 0.12, False)]
    for name, seed, X, Y, T, nb, lf, latlon in cases:
        data = field(seed, X, Y, T, nb, lf)
        net = CN.Network(data=data.copy())
        CN.Network.tau(net, 0.01)
        CN.Network.area_level(net, latlon_grid=latlon)
        rng = np.random.default_rng(seed + 100)
        if latlon:
            lat = np.tile(np.linspace(-60, 60, Y), (X, 1))
            CN.Network.intra_links(net, lat=lat)
            aux = lat
        else:
            area = 600 + 50 * rng.random((X, Y))
            CN.Network.intra_links(net, area=area)
            aux = area
        out = {"data": data, "aux": aux, "latlon": np.array(int(latlon)), "tau": np.array(net.tau), "strengthmap": net.strengthmap,
               "area_ids": np.array(list(net.V.keys()), dtype=np.int64)}
        for k in net.V:
            out["V/%d" % k] = np.array(net.V[k], dtype=np.int64)
            out["anomaly/%d" % k] = net.anomaly[k]
            out["links/%d" % k] = np.array(net.links[k], dtype=np.float64)
            out["strength/%d" % k] = np.array(net.strength[k])
        np.savez_compressed(os.path.join(OUT, "networks_%s.npz" % name), **out)
        print(name, "grid %dx%dx%d" % (X, Y, T), "tau %.4f" % net.tau, "areas", len(net.V), "sizes", sorted((len(v) for v in net.V.values()), reverse=True)[:8])


if __name__ == "__main__":
    main()
