"""Host-side AddressSanitizer run of the C shim (make -C seaiceextentforecasting_amd/csrc asan): every entry point of
include/sigp.h called with arguments that must be rejected BEFORE any device work (null handle, bad sizes), plus the
create-without-GPU path.  Run as:
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_negative_paths.py
GPU AddressSanitizer is unavailable on the pool, so this covers the host code only (argument validation, error strings)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = C.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libsigp_asan.so"))
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
assert lib.sigp_version() >= 100
h = C.c_void_p()
rc = lib.sigp_create(C.byref(h), 0, 7)
assert rc == L.BAD_ARG, rc                       # bad dtype is rejected before the device is touched
rc = lib.sigp_create(C.byref(h), 10 ** 6, 0)
assert rc in (L.HIP_ERROR,), rc                  # no such device (or no GPU at all): loud failure, no handle
assert not h.value
assert lib.sigp_last_error(None) == b"null handle"
null = C.c_void_p(None)
calls = 0
for name, (res, args) in L.SIGNATURES.items():
    if name in ("sigp_version", "sigp_create", "sigp_last_error"):
        continue
    argv = []
    for a in args:
        if a is L._h or a is C.c_void_p or a is C.c_char_p or hasattr(a, "contents"):
            argv.append(None)
        elif a is C.c_double:
            argv.append(0.0)
        else:
            argv.append(0)
    r = getattr(lib, name)(*argv)
    calls += 1
    assert r != 0 or name in ("sigp_num_blocks",), (name, r)   # null handle never reports success
# sigp_area_level is host code: run it for real under the sanitizer (random fields, plain and lat-lon grids)
import numpy as np
from seaiceextentforecasting_amd import networks as NW
runs = 0
for seed in range(6):
    r = np.random.default_rng(seed)
    dimX, dimY, T = 9 + seed, 8 + 2 * seed, 24
    base = r.standard_normal((4, T))
    f = np.full((dimX, dimY, T), np.nan)
    for i in range(dimX):
        for j in range(dimY):
            if r.random() > 0.2:
                f[i, j] = base[(i * 2 // dimX) * 2 + (j * 2 // dimY)] + r.uniform(0.3, 1.2) * r.standard_normal(T)
    net = NW.Network(data=f); NW.Network.tau(net, 0.05)
    nanc = np.where(np.isnan(f)); cell_nan = int(nanc[0][0]) * dimY + int(nanc[1][0])
    R = np.ascontiguousarray(net._R); N = R.shape[0]
    node = np.full(dimX * dimY, -1, dtype=np.int32); node[np.asarray(net.nodes[0], dtype=np.int64)] = np.arange(N, dtype=np.int32)
    cells = np.zeros(N, dtype=np.int32); offs = np.zeros(N + 1, dtype=np.int64); ids = np.zeros(N, dtype=np.int32); un = np.zeros(N, dtype=np.int32)   # un: deliberately short
    na = C.c_int64(0); nu = C.c_int64(0)
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    for latlon in (0, 1):
        rc = lib.sigp_area_level(L.ptr(R), N, p32(node), dimX, dimY, cell_nan, float(net.tau), latlon, p32(cells), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                 p32(ids), C.byref(na), p32(un), N, C.byref(nu))
        assert rc == 0 and 0 <= na.value <= N and offs[na.value] <= N and nu.value >= 0, (rc, na.value)
        runs += 1
print("asan negative paths ok: %d entry points, %d area_level runs" % (calls, runs))
