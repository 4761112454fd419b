import time, numpy as np, torch, ctypes as C
hip = C.CDLL("libamdhip64.so")
x = torch.zeros(8, device="cuda", dtype=torch.float64); torch.cuda.synchronize()
buf = (C.c_ulonglong * 4)()
def d2h(tag):
    t = time.perf_counter(); rc = hip.hipMemcpy(buf, C.c_void_p(x.data_ptr()), 32, 2); dt = (time.perf_counter() - t) * 1e3
    print("%-40s %.3f ms rc=%d" % (tag, dt, rc))
d2h("first"); d2h("second")
big = torch.zeros(30_000_000, device="cuda", dtype=torch.float64)
for kind in ("zeros", "empty", "zeros-touched"):
    arrs = [np.zeros(3_000_000, np.float64) if kind != "empty" else np.empty(3_000_000, np.float64) for _ in range(8)]
    if kind == "zeros-touched":
        for a in arrs: a[::512] = 1
    d2h("after 8 x 24MB np.%s" % kind); d2h("again")
    # large D2H into those arrays (ROCm pins user pages)
    t = time.perf_counter()
    for a in arrs: hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), C.c_void_p(big.data_ptr()), a.nbytes, 2)
    print("  8 x 24MB D2H %.3f ms" % ((time.perf_counter() - t) * 1e3))
    d2h("after big D2H")
    del arrs
    d2h("after del arrs"); d2h("again")
