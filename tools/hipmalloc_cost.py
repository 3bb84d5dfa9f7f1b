"""hipMalloc / hipFree wall time by size on this box (fresh process; second round = after a free of the same size)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipDeviceSynchronize()
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for rnd in range(2):
    for gb in (0.25, 1.0, 3.3):
        n = int(gb * (1 << 30))
        t0 = time.perf_counter()
        rc = hip.hipMalloc(C.byref(p), n)
        t1 = time.perf_counter()
        hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipFree(p)
        t3 = time.perf_counter()
        print(f"round {rnd}: {gb} GiB: hipMalloc {1e3 * (t1 - t0):.2f} ms (rc {rc}), memset+sync {1e3 * (t2 - t1):.2f} ms, hipFree {1e3 * (t3 - t2):.2f} ms", flush=True)
