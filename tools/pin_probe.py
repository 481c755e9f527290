"""Host -> HBM paths for Arrow buffers (row g of the review: pinned, overlapped streaming): what do the pieces cost here?
pageable hipMemcpy vs hipHostRegister (+ async copy + unregister) vs CPU memcpy into a pinned staging buffer."""
import ctypes, time
import numpy as np
hip = ctypes.CDLL("libamdhip64.so")
def chk(rc, what):
    assert rc == 0, (what, rc)
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
N = 256 << 20
dev = ctypes.c_void_p(); chk(hip.hipMalloc(ctypes.byref(dev), N), "malloc")
for trial in range(3):
    a = np.random.default_rng(trial).integers(0, 255, N, dtype=np.uint8)  # fresh pageable buffer, touched
    t0 = time.perf_counter(); chk(hip.hipMemcpy(dev, a.ctypes.data, N, 1), "h2d"); t1 = time.perf_counter()
    print(f"pageable hipMemcpy      : {(t1-t0)*1e3:7.2f} ms = {N/(t1-t0)/1e9:6.1f} GB/s")
    t0 = time.perf_counter(); chk(hip.hipHostRegister(a.ctypes.data, N, 0), "register"); t1 = time.perf_counter()
    chk(hip.hipMemcpyAsync(dev, a.ctypes.data, N, 1, None), "h2d async"); chk(hip.hipDeviceSynchronize(), "sync"); t2 = time.perf_counter()
    chk(hip.hipHostUnregister(a.ctypes.data), "unregister"); t3 = time.perf_counter()
    print(f"register {(t1-t0)*1e3:7.2f} ms, copy from registered {(t2-t1)*1e3:7.2f} ms = {N/(t2-t1)/1e9:6.1f} GB/s, unregister {(t3-t2)*1e3:7.2f} ms; all {N/(t3-t0)/1e9:6.1f} GB/s")
    t0 = time.perf_counter(); chk(hip.hipHostRegister(a.ctypes.data, N, 0), "register"); t1 = time.perf_counter()
    print(f"register again (pages resident) {(t1-t0)*1e3:7.2f} ms"); chk(hip.hipHostUnregister(a.ctypes.data), "unregister")
pin = ctypes.c_void_p(); chk(hip.hipHostMalloc(ctypes.byref(pin), N, 0), "hostmalloc")
pv = np.ctypeslib.as_array((ctypes.c_uint8 * N).from_address(pin.value))
for trial in range(2):
    t0 = time.perf_counter(); pv[:] = a; t1 = time.perf_counter()
    chk(hip.hipMemcpyAsync(dev, pin, N, 1, None), "h2d"); chk(hip.hipDeviceSynchronize(), "sync"); t2 = time.perf_counter()
    print(f"CPU memcpy into pinned (1 thread) {(t1-t0)*1e3:7.2f} ms = {N/(t1-t0)/1e9:6.1f} GB/s; pinned -> HBM {(t2-t1)*1e3:7.2f} ms = {N/(t2-t1)/1e9:6.1f} GB/s")
