"""Experiment (GPU box): how late is the first small operation after a host-only pause that follows a burst of device work?  Plain torch, no library code:
a burst of memory-bound kernels on two streams for ~0.3 s, a pause of P ms in which the host computes, then a 1 MB upload from page-locked memory + a tiny kernel +
stream synchronisation, timed.  Variants: nothing during the pause / a light kernel loop on another stream during the pause (keeps the device busy)."""
import time, sys, torch
dev = torch.device("cuda", 0)
x = torch.empty(1 << 28, dtype=torch.float32, device=dev); y = torch.empty_like(x)
h = torch.empty(1 << 18, dtype=torch.float32).pin_memory(); d = torch.empty(1 << 18, dtype=torch.float32, device=dev)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def burst(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        with torch.cuda.stream(s1): y.copy_(x)
        with torch.cuda.stream(s2): x.mul_(1.0001)
    torch.cuda.synchronize()
def spin(ms):
    t0 = time.perf_counter(); a = 0
    while (time.perf_counter() - t0) * 1e3 < ms: a += 1
def first_op():
    t0 = time.perf_counter()
    with torch.cuda.stream(s3):
        d.copy_(h, non_blocking=True); d.add_(1.0)
    s3.synchronize()
    return (time.perf_counter() - t0) * 1e3
burst(200); first_op()
for warm in (0, 1):
    for pause in (0, 1, 3, 6, 10, 20, 50):
        res = []
        for rep in range(6):
            burst(300)
            if warm:
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e3 < pause:
                    with torch.cuda.stream(s1): y[: 1 << 20].add_(1.0)
            else: spin(pause)
            res.append(first_op())
        print("keep-busy %d  pause %2d ms: first op + sync %s ms" % (warm, pause, " ".join("%.2f" % r for r in res)), flush=True)
