"""Does hipLaunchHostFunc hold back the stream's later work (it should), and does hipStreamWaitValue32 wait on pinned memory?"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so.7")
vp = C.c_void_p
HOSTFN = C.CFUNCTYPE(None, vp)
hip.hipLaunchHostFunc.argtypes = [vp, HOSTFN, vp]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]; hip.hipEventRecord.argtypes = [vp, vp]; hip.hipEventQuery.argtypes = [vp]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]; hip.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
st = vp(); print("stream", hip.hipStreamCreateWithFlags(C.byref(st), 1))
buf = vp(); hip.hipMalloc(C.byref(buf), 1 << 20)
ev = vp(); hip.hipEventCreateWithFlags(C.byref(ev), 2)
cb = HOSTFN(lambda _p: time.sleep(1.0))
t0 = time.perf_counter()
print("launch", hip.hipLaunchHostFunc(st, cb, None))
hip.hipMemsetAsync(buf, 0, 1 << 20, st)
hip.hipEventRecord(ev, st)
while hip.hipEventQuery(ev) != 0:
    time.sleep(0.01)
print("event behind a 1 s host function completed after %.3f s" % (time.perf_counter() - t0))
