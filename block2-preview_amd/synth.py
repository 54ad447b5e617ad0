"""Synthetic GEMM-pair plans (structure only; data is filled by the caller).

* ``random_rotate_plan``   — the shape family of the reference's own executor test
  (unit_test/test_batch_gemm.cpp:88-143: random dims 1..100, 1..30 outputs x 1..30 inputs, random
  transposes), extended with row/column slices as BatchGEMMSeq::three_rotate produces them
  (src/core/batch_gemm.hpp:952-1022).
* ``scale_plan``           — a captured plan structure with every sector dimension multiplied by an
  integer factor ("synthetic plans with the measured shape histograms scaled to M").
"""
import numpy as np

from .planfile import PAIR_DTYPE, PlanFile


def _pair(m0, n0, k0, lda0, ldb0, m1, lda1, ldc1, tb0, ta1, a0, a1, x, y, z, v):
    p = np.zeros((), PAIR_DTYPE)
    p["m0"], p["n0"], p["k0"], p["lda0"], p["ldb0"] = m0, n0, k0, lda0, ldb0
    p["m1"], p["n1"], p["k1"], p["lda1"], p["ldc1"] = m1, n0, m0, lda1, ldc1
    p["tb0"], p["ta1"], p["alpha0"], p["alpha1"] = tb0, ta1, a0, a1
    p["x_off"], p["y_off"], p["z_off"], p["v_off"] = x, y, z, v
    return p


def random_rotate_plan(rng, n_sectors=4, max_dim=100, max_terms=12, slices=True):
    """Random plan over ``n_sectors`` psi sectors / psi' sectors.  Returns a PlanFile without data."""
    src = [(int(rng.integers(1, max_dim + 1)), int(rng.integers(1, max_dim + 1))) for _ in range(n_sectors)]
    dst = [(int(rng.integers(1, max_dim + 1)), int(rng.integers(1, max_dim + 1))) for _ in range(n_sectors)]
    src_off = np.concatenate([[0], np.cumsum([a * b for a, b in src])]).astype(np.int64)
    dst_off = np.concatenate([[0], np.cumsum([a * b for a, b in dst])]).astype(np.int64)
    pairs, arena_len = [], 0
    for iv in range(n_sectors):
        vm, vn = dst[iv]
        for _ in range(int(rng.integers(1, max_terms + 1))):
            ic = int(rng.integers(0, n_sectors))
            cm, cn = src[ic]
            # optional slices: rows [ar0, ar0+am) of X -> rows [vr0, vr0+cmv) of V, or the column analogue
            mode = int(rng.integers(0, 3)) if slices else 0
            ar0, am, ac0, an = 0, cm, 0, cn
            vr0, vmm, vc0, vnn = 0, vm, 0, vn
            if mode == 1:
                am = int(rng.integers(1, cm + 1)); ar0 = int(rng.integers(0, cm - am + 1))
                vmm = int(rng.integers(1, vm + 1)); vr0 = int(rng.integers(0, vm - vmm + 1))
            elif mode == 2:
                an = int(rng.integers(1, cn + 1)); ac0 = int(rng.integers(0, cn - an + 1))
                vnn = int(rng.integers(1, vn + 1)); vc0 = int(rng.integers(0, vn - vnn + 1))
            tb0, ta1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            # Y: op(Y) is an x vnn ; Z: op(Z) is vmm x am ; both dense blocks
            y_off, arena_len = arena_len, arena_len + an * vnn
            z_off, arena_len = arena_len, arena_len + vmm * am
            pairs.append(_pair(
                am, vnn, an, cn, (an if tb0 else vnn), vmm, (vmm if ta1 else am), vn, tb0, ta1,
                float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)),
                src_off[ic] + ar0 * cn + ac0, y_off, z_off, dst_off[iv] + vr0 * vn + vc0))
    pf = PlanFile()
    pf.pairs = np.array(pairs, PAIR_DTYPE)
    pf.psi_len, pf.sigma_len, pf.arena_len = int(src_off[-1]), int(dst_off[-1]), int(arena_len)
    pf.max_work = int((pf.pairs["m0"].astype(np.int64) * pf.pairs["n0"]).max())
    return pf


def operator_product_plan(rng, n_row=3, n_col=3, max_dim=60, n_left=4, n_right=4, n_terms=10):
    """Plan of H = sum_t  L_{a(t)} (x) R_{b(t)}  on a psi with n_row x n_col sectors: every operator has a dense block for
    every (sector, sector) it connects, a term contributes one pair per (row block of L) x (column block of R).  Terms
    that reuse a right operator share the stage-0 product X.op(Y) of their pairs, terms that reuse a left operator share
    op(Z).X — the structure the plan compiler exploits.  Returns a PlanFile without data."""
    rd = [int(rng.integers(1, max_dim + 1)) for _ in range(n_row)]
    cd = [int(rng.integers(1, max_dim + 1)) for _ in range(n_col)]
    off = np.zeros((n_row, n_col), np.int64)
    tot = 0
    for r in range(n_row):
        for c in range(n_col):
            off[r, c], tot = tot, tot + rd[r] * cd[c]
    arena_len = 0

    def make_op(dims):
        nonlocal arena_len
        blocks = {}
        for i in range(len(dims)):
            for j in range(len(dims)):
                if rng.random() < 0.6:
                    tr = int(rng.integers(0, 2))
                    blocks[(i, j)] = (arena_len, tr)  # op block maps sector j -> sector i: op(.) is dims[i] x dims[j]
                    arena_len += dims[i] * dims[j]
        return blocks

    lops = [make_op(rd) for _ in range(n_left)]
    rops = [make_op(cd) for _ in range(n_right)]
    pairs = []
    for _ in range(n_terms):
        a, b = int(rng.integers(n_left)), int(rng.integers(n_right))
        alpha = float(rng.uniform(-1, 1))
        for (r, rp), (z_off, ta1) in lops[a].items():
            for (c, cp), (y_off, tb) in rops[b].items():
                # V(r, c) += alpha * op(Z)(r <- rp) . X(rp, cp) . op(Y)^T(cp -> c);  op(Y) as stored is cd[c] x cd[cp]:
                # stage 0 needs the (cd[cp] x cd[c]) matrix: tb0 = 1 reads the stored block transposed
                tb0 = 1 - tb  # stored (c x cp) row-major read transposed, or stored (cp x c) read plain
                ldb0 = cd[cp] if tb0 else cd[c]
                lda1 = rd[r] if ta1 else rd[rp]
                pairs.append(_pair(rd[rp], cd[c], cd[cp], cd[cp], ldb0, rd[r], lda1, cd[c], tb0, ta1, 1.0, alpha,
                                   off[rp, cp], y_off, z_off, off[r, c]))
    pf = PlanFile()
    pf.pairs = np.array(pairs, PAIR_DTYPE)
    pf.psi_len = pf.sigma_len = int(tot)
    pf.arena_len = int(arena_len)
    pf.max_work = int((pf.pairs["m0"].astype(np.int64) * pf.pairs["n0"]).max()) if len(pairs) else 0
    return pf


def _column_offsets(off, m, n, ld):
    """Column index (inside its sector row) of the first element of each window: windows whose address
    ranges overlap belong to one sector; the sector's rows start where a window with col 0 starts."""
    order = np.argsort(off, kind="stable")
    col = np.zeros(len(off), np.int64)
    i = 0
    while i < len(order):
        a = order[i]
        end = off[a] + (m[a] - 1) * ld[a] + n[a]
        j = i + 1
        while j < len(order) and off[order[j]] < end:
            b = order[j]
            end = max(end, off[b] + (m[b] - 1) * ld[b] + n[b])
            j += 1
        grp = order[i:j]
        multi = [g for g in grp if m[g] > 1]
        L = int(ld[multi[0]]) if multi else int(max(off[g] - off[grp[0]] + n[g] for g in grp))
        base = int(off[grp[0]])
        c0 = 0
        for g in grp:
            if (int(off[g]) - base + c0) % L + int(n[g]) > L:
                c0 = (L - (int(off[g]) - base) % L) % L
        for g in grp:
            col[g] = (int(off[g]) - base + c0) % L
        i = j
    return col


def scale_plan(pf, f):
    """Every sector dimension x f (offsets follow: f*f on whole rows, f on the in-row column offset)."""
    f = int(f)
    p = pf.pairs.copy()
    i64 = lambda a: a.astype(np.int64)
    xc = _column_offsets(i64(p["x_off"]), i64(p["m0"]), i64(p["k0"]), i64(p["lda0"]))
    vc = _column_offsets(i64(p["v_off"]), i64(p["m1"]), i64(p["n1"]), i64(p["ldc1"]))
    q = p.copy()
    for nm in ("m0", "n0", "k0", "lda0", "ldb0", "m1", "n1", "k1", "lda1", "ldc1"):
        q[nm] = p[nm] * f
    q["x_off"] = (f * f * (i64(p["x_off"]) - xc) + f * xc).astype(np.uint64)
    q["v_off"] = (f * f * (i64(p["v_off"]) - vc) + f * vc).astype(np.uint64)
    q["y_off"] = (f * f * i64(p["y_off"])).astype(np.uint64)
    q["z_off"] = (f * f * i64(p["z_off"])).astype(np.uint64)
    out = PlanFile()
    out.pairs = q
    out.psi_len, out.sigma_len, out.arena_len = pf.psi_len * f * f, pf.sigma_len * f * f, pf.arena_len * f * f
    out.max_work = pf.max_work * f * f
    out.meta = pf.meta.copy()
    return out


def scale_gemm_list(gl, f):
    """Single-GEMM list with every sector dimension x f (same rule as scale_plan: f*f on whole rows, f on the in-row
    column offset; operator operands are whole blocks)."""
    from .planfile import GemmList

    f = int(f)
    g = gl.gemms
    i64 = lambda a: a.astype(np.int64)
    m, n, k = i64(g["m"]), i64(g["n"]), i64(g["k"])
    # (rows, cols) as stored in memory
    ar, ac = np.where(g["ta"] == 1, k, m), np.where(g["ta"] == 1, m, k)
    br, bc = np.where(g["tb"] == 1, n, k), np.where(g["tb"] == 1, k, n)
    q = g.copy()
    for nm in ("m", "n", "k", "lda", "ldb", "ldc"):
        q[nm] = g[nm] * f
    # input-vector operands (A and B together: they slice the same sectors)
    ia, ib = np.nonzero(g["a_src"] == 1)[0], np.nonzero(g["b_src"] == 1)[0]
    off = np.concatenate([i64(g["a_off"])[ia], i64(g["b_off"])[ib]])
    rows = np.concatenate([ar[ia], br[ib]])
    cols = np.concatenate([ac[ia], bc[ib]])
    ld = np.concatenate([i64(g["lda"])[ia], i64(g["ldb"])[ib]])
    col = _column_offsets(off, rows, cols, ld) if len(off) else np.zeros(0, np.int64)
    new = f * f * (off - col) + f * col
    q["a_off"] = (f * f * i64(g["a_off"])).astype(np.uint64)
    q["b_off"] = (f * f * i64(g["b_off"])).astype(np.uint64)
    q["a_off"][ia] = new[:len(ia)].astype(np.uint64)
    q["b_off"][ib] = new[len(ia):].astype(np.uint64)
    cc = _column_offsets(i64(g["c_off"]), m, n, i64(g["ldc"]))
    q["c_off"] = (f * f * (i64(g["c_off"]) - cc) + f * cc).astype(np.uint64)
    out = GemmList()
    out.gemms = q
    out.arena_len, out.in_len, out.out_len = gl.arena_len * f * f, gl.in_len * f * f, gl.out_len * f * f
    out.macs, out.forward = gl.macs * f ** 3, gl.forward
    out.out_offsets, out.out_lens = gl.out_offsets * (f * f), gl.out_lens * (f * f)
    return out


def shard_pairs(pairs, rank, world):
    """sum-MPO style sharding of one plan: every operator TERM (here: distinct stage-1 left operator
    block z_off, i.e. one left-block operator of the MPO bond) is owned by exactly one rank, as
    ParallelRuleSimple::index_prefactor assigns integrals (src/dmrg/parallel_simple.hpp:56-99).
    The partial sigma of all ranks sums to the full H psi.  Owners are chosen by MAC weight (longest processing time
    first, ties by block order) so that the ranks' H.psi times match; every rank computes the same assignment."""
    if world == 1:
        return pairs
    _, inv = np.unique(pairs["z_off"], return_inverse=True)
    macs = (pairs["m0"].astype(np.int64) * pairs["n0"] * pairs["k0"] +
            pairs["m1"].astype(np.int64) * pairs["n1"] * pairs["k1"]).astype(np.float64)
    wgt = np.bincount(inv, weights=macs)
    order = np.argsort(-wgt, kind="stable")
    owner = np.zeros(len(wgt), np.int64)
    load = np.zeros(world)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += wgt[i]
    return pairs[owner[inv] == rank]


def compact_arena(pairs, return_runs=False):
    """Renumber y_off / z_off so that only the operator blocks these pairs reference remain, packed
    back to back (what one sum-MPO rank actually holds).  Returns (pairs, arena_len), and with return_runs the map of
    the packing as well: (old_start, new_start, length) of every contiguous run that was kept."""
    p = pairs.copy()
    ey = np.where(p["tb0"] == 1, (p["n0"].astype(np.int64) - 1) * p["ldb0"] + p["k0"],
                  (p["k0"].astype(np.int64) - 1) * p["ldb0"] + p["n0"])
    ez = np.where(p["ta1"] == 1, (p["k1"].astype(np.int64) - 1) * p["lda1"] + p["m1"],
                  (p["m1"].astype(np.int64) - 1) * p["lda1"] + p["k1"])
    start = np.concatenate([p["y_off"].astype(np.int64), p["z_off"].astype(np.int64)])
    end = start + np.concatenate([ey, ez])
    order = np.argsort(start, kind="stable")
    s, e = start[order], end[order]
    # merge overlapping ranges
    run_end = np.maximum.accumulate(e)
    new_run = np.ones(len(s), bool)
    new_run[1:] = s[1:] >= run_end[:-1]
    run_id = np.cumsum(new_run) - 1
    run_start = s[new_run]
    run_stop = np.zeros(run_id[-1] + 1 if len(s) else 0, np.int64)
    np.maximum.at(run_stop, run_id, e)
    run_len = run_stop - run_start
    run_new = np.concatenate([[0], np.cumsum(run_len)])
    new_start = np.empty(len(s), np.int64)
    new_start[order] = run_new[run_id] + (s - run_start[run_id])
    n = len(p)
    p["y_off"] = new_start[:n].astype(np.uint64)
    p["z_off"] = new_start[n:].astype(np.uint64)
    if return_runs:
        return p, int(run_new[-1]), (run_start, run_new[:-1], run_len)
    return p, int(run_new[-1])
