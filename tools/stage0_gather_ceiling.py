"""Ceiling of a 'gathered-column' stage 0 (the review's proposal: all stage-0 products of one psi slice X as ONE product
X . [op(Y_1) | op(Y_2) | ...]).  The distinct stage-0 products X . op(Y_j) of a captured H.psi plan run (a) as they are — one small
GEMM each, the shape stage 0 has today — and (b) grouped by X with the Y_j of a group CONCATENATED in memory into one k x sum(n_j)
operand: one wide GEMM per group, same MACs, same kernel (b2x_gemm_plan).  (b) is what a gathered stage 0 could at best reach:
the real thing would still fetch its columns from the scattered Y_j.   usage: stage0_gather_ceiling.py [scale ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import GEMM_DTYPE, read_struct_npz


def timed(plan, vin, vout, reps=20):
    for _ in range(3):
        plan.execute_device(vin.ptr, vout.ptr, 1.0)
    capi.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute_device(vin.ptr, vout.ptr, 1.0)
    capi.device_sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    capi.device_init(0)
    base = read_struct_npz(os.path.join(ROOT, "tests", "golden", "cr2_su2_m250_sw1_site20.struct.npz"))
    for scale in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        pf = synth.scale_plan(base, scale) if scale != 1 else base
        p = pf.pairs
        key = np.stack([p[f].astype(np.int64) for f in ("x_off", "y_off", "m0", "n0", "k0", "lda0", "ldb0", "tb0")], 1)
        key = key[p["ta0"] == 0]
        uniq = np.unique(key, axis=0)
        rng = np.random.default_rng(0)
        psi = capi.DeviceBuffer(pf.psi_len, rng.standard_normal(pf.psi_len))
        # (a) one GEMM per product; (b) one GEMM per (X, k0): the group's operands copied side by side into a new arena
        ga = np.zeros(len(uniq), GEMM_DTYPE)
        groups = {}
        for r in uniq:
            groups.setdefault((int(r[0]), int(r[2]), int(r[4]), int(r[5])), []).append(r)
        gb = np.zeros(len(groups), GEMM_DTYPE)
        cat_len = sum(int(r[3]) * int(r[4]) for r in uniq)
        out_len = sum(int(r[2]) * int(r[3]) for r in uniq)
        c_off = b_off = 0
        ia = 0
        macs = 0
        for gi, ((x_off, m0, k0, lda0), rs) in enumerate(groups.items()):
            nsum = sum(int(r[3]) for r in rs)
            gb[gi] = (m0, nsum, k0, lda0, nsum, nsum, 0, 0, 1, 0, 0, 1.0, x_off, b_off, c_off)
            for r in rs:
                n0 = int(r[3])
                ga[ia] = (m0, n0, k0, lda0, int(r[6]), n0, 0, int(r[7]), 1, 0, 0, 1.0, x_off, int(r[1]), c_off)
                c_off += m0 * n0
                ia += 1
                macs += m0 * n0 * k0
            b_off += k0 * nsum
        buf_a = capi.DeviceBuffer(pf.arena_len, rng.standard_normal(pf.arena_len))
        buf_b = capi.DeviceBuffer(cat_len, rng.standard_normal(cat_len))
        arena_a = capi.Arena.adopt_device(buf_a.ptr, pf.arena_len, keep=buf_a)
        arena_b = capi.Arena.adopt_device(buf_b.ptr, cat_len, keep=buf_b)
        out = capi.DeviceBuffer(out_len)
        pa, pb = capi.GemmPlan(arena_a, ga, pf.psi_len, out_len), capi.GemmPlan(arena_b, gb, pf.psi_len, out_len)
        ta, tb = timed(pa, psi, out), timed(pb, psi, out)
        widths = np.array([g["n"] for g in gb])
        print("x%d (M=%d): %d distinct stage-0 products (mean %.1f x %.1f x k %.1f, %.2f GMAC) in %d groups (mean width %.0f): "
              "one GEMM each %.3f ms (%.1f TFLOP/s), one concatenated GEMM per group %.3f ms (%.1f TFLOP/s): x%.2f" % (
                  scale, 250 * scale, len(uniq), uniq[:, 2].mean(), uniq[:, 3].mean(), uniq[:, 4].mean(), macs / 1e9, len(groups),
                  widths.mean(), ta, 2 * macs / ta / 1e9, tb, 2 * macs / tb / 1e9, ta / tb), flush=True)
        for o in (pa, pb, arena_a, arena_b, out, psi):
            o.close()


if __name__ == "__main__":
    main()
