"""MI355X-native Gaussian-process engine for the GPR hot path of SeaIceExtentForecasting.

Host side of the drop-in boundary: ``GPR.fit / predict / nlml / fit_batch / nlml_grid``
over hand-written HIP kernels behind a C ABI (include/sigp.h, libsigp.so).  There is no CPU fallback.
"""
from .gpr import GPR, LinAlgError  # noqa: F401
from .features import (SCRIPT_TABLE, LGRID, SGRID, select_features, design_matrix, laplacian_M,  # noqa: F401
                       sigma_tilde, SigmaEigh)
from .retro import retro_forecast, operational_forecast, retro_grid_search, retro_optimise  # noqa: F401
from .smallbatch import SmallBatch  # noqa: F401
from .callers import detrend, detrend_cube, skill, forecast_tables  # noqa: F401
from .networks import Network, networks_retro  # noqa: F401  (the scripts' networks() driver lives at seaiceextentforecasting_amd.networks.networks)
from .dist import shard_indices, gather_results, fit_batch_sharded, DistributedGPR  # noqa: F401

__all__ = ["GPR", "LinAlgError", "SCRIPT_TABLE", "LGRID", "SGRID", "select_features", "design_matrix",
           "laplacian_M", "sigma_tilde", "SigmaEigh", "retro_forecast", "operational_forecast", "retro_grid_search", "retro_optimise"]
