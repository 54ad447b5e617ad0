"""device memory and host RSS after repeated replays of one chain in ONE process (nothing may grow from pass to pass once the
caches are warm): tools/sweep_leak_check.py <prefix> <su2|sz> <n_sweeps> [passes]"""
import os
import resource
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (hipMemGetInfo through torch)

from block2_preview_amd import capi  # noqa: E402
from block2_preview_amd.sweep import DMRG, ChainFixture  # noqa: E402

prefix, sym, n_sw = sys.argv[1], sys.argv[2], int(sys.argv[3])
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
capi.device_init(0)
conv = 1e-18 if "cut9" in prefix else 1e-13
for p in range(passes):
    fx = ChainFixture(prefix).preload()
    dm = DMRG(fx, sym, conv_thrd=conv)
    dm.init_environments()
    for isw in range(n_sw):
        dm.sweep(isw, isw % 2 == 0)
    for t in list(dm.L.values()) + list(dm.R.values()) + [dm.EL, dm.ER]:
        if t is not None:
            t.close()
    del dm, fx
    capi.device_sync()
    free, total = torch.cuda.mem_get_info(0)
    print("pass %d: device memory in use %.3f GB (library caches included), host RSS high-water %.2f GB, plan cache %s" % (
        p, (total - free) / 1e9, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, capi.plan_cache_stats()), flush=True)
released = capi.trim()
free, total = torch.cuda.mem_get_info(0)
print("after b2x_trim (%s bytes released): device memory in use %.3f GB" % (released, (total - free) / 1e9))
