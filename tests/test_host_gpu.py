"""The C++ host mirror driving the MI355X path the way block2's own tests drive block2:
* TestRotateTasked     (unit_test/test_batch_gemm.cpp:88-143)  — record with rotate, replay with operator()
* three_rotate         (unit_test/test_matrix.cpp:394-462)     — sliced pair == rotate with the embedded block
* EffectiveHamiltonian::eigs / davidson — site energies of the golden reference runs (tests/golden/*.log)
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_plan_files
from block2_preview_amd.planfile import read_plan
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host(gpu):
    from block2_preview_amd import b2x_host

    b2x_host.device_init(0)
    return b2x_host


def test_rotate_tasked(host):
    rng = np.random.default_rng(1969)
    thrd = 1e-10
    for _ in range(25):
        ma, na, mc, nc = (int(x) for x in rng.integers(1, 101, 4))
        ncbatch, nbatch = (int(x) for x in rng.integers(1, 31, 2))
        a = rng.random((nbatch, ma, na))
        c = np.zeros((ncbatch, mc, nc))
        d = rng.random(ncbatch)
        l, r = rng.random((mc, ma)), rng.random((na, nc))
        conjl, conjr = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        seq = host.BatchGEMMSeq(1 << 24, host.SeqTypes.Device)
        for ic in range(ncbatch):
            for ii in range(nbatch):
                seq.rotate((ii * ma * na, ma, na), (ic * mc * nc, mc, nc), l.reshape(ma, mc) if conjl else l, conjl,
                           r.reshape(nc, na) if conjr else r, conjr, d[ic])
        cf = c.ravel()
        seq(a.ravel(), cf)
        seq.deallocate()
        seq.clear()
        L = l.reshape(ma, mc).T if conjl else l
        R = r.reshape(nc, na).T if conjr else r
        for ic in range(ncbatch):
            std = d[ic] * sum(L @ a[ii] @ R for ii in range(nbatch))
            assert np.allclose(cf.reshape(ncbatch, mc, nc)[ic], std, rtol=thrd, atol=thrd)


@pytest.mark.parametrize("dleft", [True, False])
def test_three_rotate_equals_rotate_with_embedded_block(host, dleft):
    """bra (or ket) = (1x1 site block) x (block-operator sub-block) placed at `stride` inside the enlarged operator"""
    rng = np.random.default_rng(7 if dleft else 8)
    for _ in range(12):
        am, an, cm, cn = (int(x) for x in rng.integers(2, 40, 4))
        a, c = rng.random((am, an)), np.zeros((cm, cn))
        conj_bra, conj_ket, dconj = (bool(x) for x in rng.integers(0, 2, 3))
        scal = np.array([[rng.uniform(-1, 1)]])
        scale = float(rng.uniform(-1, 1))
        if dleft:
            # enlarged bra maps a-rows (am) to c-rows (cm); the sub-block couples rows [a0,a0+sa) -> [c0,c0+sc)
            sa, sc = int(rng.integers(1, am + 1)), int(rng.integers(1, cm + 1))
            a0, c0 = int(rng.integers(0, am - sa + 1)), int(rng.integers(0, cm - sc + 1))
            big_op = rng.random((sc, sa))  # op(big), sc x sa
            eff = dconj ^ conj_bra
            big = big_op.T.copy() if eff else big_op
            bra_full = np.zeros((cm, am))
            bra_full[c0:c0 + sc, a0:a0 + sa] = scal[0, 0] * big_op
            bra_store = bra_full.T.copy() if conj_bra else bra_full  # stored so that op(bra) = bra_full
            ket = rng.random((cn, an)) if conj_ket else rng.random((an, cn))
            stride = (a0 * bra_store.shape[1] + c0) if conj_bra else (c0 * bra_store.shape[1] + a0)
            seq = host.BatchGEMMSeq()
            seq.three_rotate((0, am, an), (0, cm, cn), bra_store, conj_bra, ket, conj_ket, scal, False, big, dconj, True,
                             scale, stride)
            K = ket.T if conj_ket else ket
            std = scale * bra_full @ a @ K
        else:
            sa, sc = int(rng.integers(1, an + 1)), int(rng.integers(1, cn + 1))
            a0, c0 = int(rng.integers(0, an - sa + 1)), int(rng.integers(0, cn - sc + 1))
            big_op = rng.random((sa, sc))  # op(big): a-cols -> c-cols
            eff = dconj ^ conj_ket
            # stage 0 uses conj flag 1 (transpose) when the effective conj is set, else the block as stored
            big = big_op.T.copy() if eff else big_op
            ket_full = np.zeros((an, cn))
            ket_full[a0:a0 + sa, c0:c0 + sc] = scal[0, 0] * big_op
            ket_store = ket_full.T.copy() if conj_ket else ket_full
            bra = rng.random((am, cm)) if conj_bra else rng.random((cm, am))
            stride = (c0 * ket_store.shape[1] + a0) if conj_ket else (a0 * ket_store.shape[1] + c0)
            seq = host.BatchGEMMSeq()
            seq.three_rotate((0, am, an), (0, cm, cn), bra, conj_bra, ket_store, conj_ket, scal, False, big, dconj,
                             False, scale, stride)
            B = bra.T if conj_bra else bra
            std = scale * B @ a @ ket_full
        cf = c.ravel()
        seq(a.ravel(), cf)
        assert np.allclose(cf.reshape(cm, cn), std, rtol=1e-10, atol=1e-10)


def _site_energy(log, sweep, site):
    for line in open(os.path.join(GOLDEN, log)):
        t = line.split()
        if t and t[0] == "SITE_ENERGY" and int(t[1]) == sweep and int(t[2]) == site:
            return float(t[3])
    raise KeyError((log, sweep, site))


CASES = [("n2su2.sw0.site4.plan", "n2su2.log"), ("n2su2.sw2.site5.plan", "n2su2.log"),
         ("n2sz.sw2.site4.plan", "n2sz.log"), ("h10szm50.sw2.site4.plan", "h10szm50.log"),
         ("h10szm50.sw1.site5.plan", "h10szm50.log")]


@pytest.mark.parametrize("plan,log", CASES)
def test_eigs_reproduces_reference_site_energy(host, plan, log):
    """EffectiveHamiltonian::eigs (device-resident Davidson) from the reference's own initial guess reaches the
    energy the reference's Davidson reported for that (sweep, site): E = eig + const_e."""
    pf = read_plan(os.path.join(GOLDEN, plan))
    sweep, site, const_e = int(pf.meta[0]), int(pf.meta[1]), float(pf.meta[2])
    seq = host.BatchGEMMSeq()
    seq.load_pairs(pf.pairs, pf.arena)
    h = host.EffectiveHamiltonian(seq, pf.diag.tolist())
    e, ndav, nflop, tdav, ket = h.eigs(pf.psi.tolist(), conv_thrd=1e-12, max_iter=500)
    h.post_precompute()
    e_ref = _site_energy(log, sweep, site)
    # the reference stopped at its sweep threshold (squared residual 1e-9 .. 1e-7): its value is above ours by
    # at most the residual; 1e-7 Ha is the in-tree test tolerance (unit_test/test_dmrg_n2_sto3g.cpp:126)
    assert abs((e + const_e) - e_ref) < 1e-7, (e + const_e, e_ref, ndav)
    assert nflop == ndav * pf.macs
    assert abs(np.linalg.norm(ket) - 1.0) < 1e-10


def test_eigs_matches_dense_diagonalisation(host):
    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw2.site5.plan"))
    H = oracle.dense(pf.pairs, pf.arena, pf.psi_len, pf.sigma_len)
    assert np.abs(H - H.T).max() < 1e-10
    w = np.linalg.eigvalsh(0.5 * (H + H.T))
    assert np.allclose(np.diag(H), pf.diag, atol=1e-10)  # the reference's diag really is the plan's diagonal
    seq = host.BatchGEMMSeq()
    seq.load_pairs(pf.pairs, pf.arena)
    h = host.EffectiveHamiltonian(seq, pf.diag.tolist())
    e, ndav, _, _, ket = h.eigs(pf.psi.tolist(), conv_thrd=1e-14, max_iter=300)
    assert abs(e - w[0]) < 1e-10
    assert np.abs(H @ ket - e * ket).max() < 1e-6


def _dense_case(host):
    from block2_preview_amd import capi

    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw2.site5.plan"))
    H = oracle.dense(pf.pairs, pf.arena, pf.psi_len, pf.sigma_len)
    H = 0.5 * (H + H.T)
    w, v = np.linalg.eigh(H)
    return capi, pf, H, w, v


def test_davidson_types_shift_and_rel_thrd(host):
    """davidson_type / shift select the roots as the reference orders them (iterative_matrix_functions.hpp:1024-1049):
    CloseTo -> nearest to shift, LessThan -> largest eigenvalue <= shift, GreaterThan -> smallest >= shift; the
    DavidsonPrecond / NoPrecond variants (:1084-1087) reach the same ground state; rel_conv_thrd loosens the stop test."""
    capi, pf, H, w, v = _dense_case(host)
    DT = host.DavidsonTypes
    seq = host.BatchGEMMSeq()
    seq.load_pairs(pf.pairs, pf.arena)
    h = host.EffectiveHamiltonian(seq, pf.diag.tolist())
    rng = np.random.default_rng(3)
    shift = 0.5 * (w[3] + w[4]) + 0.1 * (w[4] - w[3])  # between the 4th and 5th level, nearer to the 5th
    for typ, want in ((DT.CloseTo, w[4]), (DT.LessThan, w[3]), (DT.GreaterThan, w[4])):
        # guess = target eigenvector + noise (interior roots from a random guess need many iterations)
        tgt = v[:, int(np.argmin(np.abs(w - want)))]
        g = tgt + 0.02 * rng.standard_normal(len(tgt)) / np.sqrt(len(tgt))  # Ritz value of the guess on the right side of shift
        e, ndav, _, _, ket = h.eigs(g.tolist(), conv_thrd=1e-12, max_iter=2000, davidson_type=typ, shift=shift)
        assert abs(e - want) < 1e-8, (typ, e, want, ndav)
        assert np.abs(H @ ket - e * ket).max() < 1e-5
    nd = {}
    for typ in (DT.Normal, DT.DavidsonPrecond, DT.NoPrecond):
        e, nd[typ], _, _, ket = h.eigs(pf.psi.tolist(), conv_thrd=1e-12, max_iter=2000, davidson_type=typ)
        assert abs(e - w[0]) < 1e-9, (typ, e, w[0])
    g0 = (v[:, 0] + 0.3 * rng.standard_normal(len(w)) / np.sqrt(len(w))).tolist()
    e_tight, nd_tight, _, _, _ = h.eigs(g0, conv_thrd=1e-14, max_iter=2000)
    e_loose, nd_loose, _, _, _ = h.eigs(g0, conv_thrd=1e-14, rel_conv_thrd=1e-4, max_iter=2000)
    assert nd_loose < nd_tight and abs(e_loose - w[0]) < 1e-4 and abs(e_tight - w[0]) < 1e-10  # |r|^2 < thrd + (E rel)^2
    with pytest.raises(RuntimeError):
        h.eigs(pf.psi.tolist(), davidson_type=DT.Harmonic)
    h.post_precompute()


def test_davidson_projects_out_ortho_states(host):
    """ors / proj_weights (iterative_matrix_functions.hpp:888-893, 946-959, 975-977): with the ground state projected
    out ((1 - |v><v|)) or shifted up (H + w |v><v|) Davidson returns the first excited state."""
    capi, pf, H, w, v = _dense_case(host)
    seq = host.BatchGEMMSeq()
    seq.load_pairs(pf.pairs, pf.arena)
    h = host.EffectiveHamiltonian(seq, pf.diag.tolist())
    gs = v[:, 0]
    rng = np.random.default_rng(5)
    g = (v[:, 1] + 0.1 * rng.standard_normal(len(gs)) / np.sqrt(len(gs))).tolist()
    e, ndav, _, _, ket = h.eigs(g, conv_thrd=1e-12, max_iter=2000, ortho_bra=[(2.5 * gs).tolist()])
    assert abs(e - w[1]) < 1e-8 and abs(ket @ gs) < 1e-8
    e2, _, _, _, ket2 = h.eigs(g, conv_thrd=1e-12, max_iter=2000, ortho_bra=[gs.tolist()],
                               projection_weights=[w[-1] - w[0] + 1.0])
    assert abs(e2 - w[1]) < 1e-8 and abs(ket2 @ gs) < 1e-6
    h.post_precompute()


def test_vec_precondition_is_the_reference_formula(host):
    """davidson_precondition (iterative_matrix_functions.hpp:66-72): q[i] /= ld - aa[i] where |ld - aa[i]| > 1e-12."""
    from block2_preview_amd import capi

    rng = np.random.default_rng(11)
    n = 10007
    q, aa = rng.standard_normal(n), rng.standard_normal(n)
    ld = 0.3
    aa[5] = ld  # untouched element
    import ctypes as C

    dq, da = capi.DeviceBuffer(n, q), capi.DeviceBuffer(n, aa)
    capi.check(capi.lib().b2x_vec_precondition(C.c_void_p(dq.ptr), C.c_void_p(da.ptr), C.c_double(ld), C.c_size_t(n), None))
    ref = q.copy()
    m = np.abs(ld - aa) > 1e-12
    ref[m] /= ld - aa[m]
    out = dq.download()
    assert out[5] == q[5] and np.abs(out - ref).max() <= 4e-16 * np.abs(ref).max()


def test_davidson_over_a_sum_of_plans(host):
    """davidson_device(plan, ..., more_plans=[...]): H = sum of several plans accumulating into one sigma (the sum-MPO
    Hamiltonian H = sum_r H_r with the ranks' plans held by one process) gives the eigenpair of the undivided plan"""
    from block2_preview_amd import capi

    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw2.site5.plan"))
    arena = capi.Arena.from_host([pf.arena])
    n = pf.psi_len
    diag = capi.DeviceBuffer(n, pf.diag)
    full = capi.Plan(arena, pf.pairs, n, n)
    ket = capi.DeviceBuffer(n, pf.psi)
    e_full, nd_full = host.davidson_device(full._h.value, diag.ptr, ket.ptr, n, 1e-12, 500)
    v_full = ket.download()
    half = len(pf.pairs) // 2
    pa = capi.Plan(arena, pf.pairs[:half], n, n)
    pb = capi.Plan(arena, pf.pairs[half:], n, n)
    ket.upload(pf.psi)
    e_sum, nd_sum = host.davidson_device(pa._h.value, diag.ptr, ket.ptr, n, 1e-12, 500, more_plans=[pb._h.value])
    v_sum = ket.download()
    assert abs(e_sum - e_full) < 1e-10 and abs(abs(v_sum @ v_full) - 1.0) < 1e-8
    sa, sf = capi.DeviceBuffer(n), capi.DeviceBuffer(n)  # (half of the pairs alone is another operator)
    pa.execute_device(ket.ptr, sa.ptr, 1.0)
    full.execute_device(ket.ptr, sf.ptr, 1.0)
    capi.device_sync()
    assert np.abs(sa.download() - sf.download()).max() > 1e-3
    sa.close(), sf.close()
    for x in (full, pa, pb, arena, diag, ket):
        x.close()
