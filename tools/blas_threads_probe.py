"""host BLAS threading on the GPU box: cores visible, the BLAS pool numpy created, and the time of the sweep loop's small host
operations (a psi-sized dot, eigh of sector-sized matrices) under different pool sizes"""
import os
import time

import numpy as np
from threadpoolctl import threadpool_info, threadpool_limits

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for fn in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(fn):
        print(fn, open(fn).read().strip())
for i in threadpool_info():
    print(i)
rng = np.random.default_rng(0)
x = rng.standard_normal(600000)
mats = [rng.standard_normal((n, n)) for n in (20, 60, 150, 300) for _ in range(8)]
mats = [m + m.T for m in mats]
a, b = rng.standard_normal((300, 2000)), rng.standard_normal((2000, 300))
for lim in (None, 16, 8, 4, 1):
    with threadpool_limits(limits=lim, user_api="blas"):
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                x @ x
            t1 = time.perf_counter()
            for m in mats:
                np.linalg.eigh(m)
            t2 = time.perf_counter()
            for _ in range(10):
                a @ b
            t3 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1, t3 - t2))
        print("limit %-5s  20 dots %.4f..%.4f s   32 eigh %.4f..%.4f s   10 gemm %.4f..%.4f s" % (
            lim, min(t[0] for t in ts), max(t[0] for t in ts), min(t[1] for t in ts), max(t[1] for t in ts),
            min(t[2] for t in ts), max(t[2] for t in ts)))
