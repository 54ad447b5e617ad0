"""Reader/writer for the B2XPLAN1 container (layout: oracle/planfile.h) and the numpy view of
``b2x_pair`` (include/b2x.h).  Pure data plumbing — no arithmetic lives here."""
import numpy as np

PAIR_DTYPE = np.dtype(
    [
        ("m0", "<i4"), ("n0", "<i4"), ("k0", "<i4"),
        ("lda0", "<i4"), ("ldb0", "<i4"),
        ("m1", "<i4"), ("n1", "<i4"), ("k1", "<i4"),
        ("lda1", "<i4"), ("ldc1", "<i4"),
        ("ta0", "u1"), ("tb0", "u1"), ("ta1", "u1"), ("tb1", "u1"),
        ("reserved", "<u4"),
        ("alpha0", "<f8"), ("alpha1", "<f8"),
        ("x_off", "<u8"), ("y_off", "<u8"), ("z_off", "<u8"), ("v_off", "<u8"),
    ],
    align=False,
)
assert PAIR_DTYPE.itemsize == 96

# numpy view of b2x_diag_term (include/b2x.h)
DIAG_TERM_DTYPE = np.dtype([("m", "<i4"), ("n", "<i4"), ("a_stride", "<i4"), ("b_stride", "<i4"), ("ldc", "<i4"),
                            ("reserved", "<i4"), ("alpha", "<f8"), ("a_off", "<u8"), ("b_off", "<u8"), ("c_off", "<u8")])
assert DIAG_TERM_DTYPE.itemsize == 56

# numpy view of b2x_gemm (include/b2x.h): one single-GEMM record C += alpha op(A) op(B)
GEMM_DTYPE = np.dtype([("m", "<i4"), ("n", "<i4"), ("k", "<i4"), ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4"),
                       ("ta", "u1"), ("tb", "u1"), ("a_src", "u1"), ("b_src", "u1"), ("reserved", "<u4"),
                       ("alpha", "<f8"), ("a_off", "<u8"), ("b_off", "<u8"), ("c_off", "<u8")])
assert GEMM_DTYPE.itemsize == 64

# numpy view of b2x_outer_term (include/b2x.h): C[r][c] += alpha * A[a_off + r*a_rs + c*a_cs] * B[b_off + r*b_rs + c*b_cs]
OUTER_TERM_DTYPE = np.dtype([("m", "<i4"), ("n", "<i4"), ("a_rs", "<i4"), ("a_cs", "<i4"), ("b_rs", "<i4"), ("b_cs", "<i4"),
                             ("ldc", "<i4"), ("a_src", "u1"), ("b_src", "u1"), ("reserved", "u1", 2), ("alpha", "<f8"),
                             ("a_off", "<u8"), ("b_off", "<u8"), ("c_off", "<u8")])
assert OUTER_TERM_DTYPE.itemsize == 64

F_ARENA, F_PSI, F_SIGMA, F_DIAG, F_PSIOUT = 1, 2, 4, 8, 16


class PlanFile:
    """In-memory image of one plan file."""

    def __init__(self):
        self.pairs = np.zeros(0, PAIR_DTYPE)
        self.psi_len = self.sigma_len = self.max_work = self.arena_len = 0
        self.ranges = np.zeros((0, 2), np.uint64)
        self.meta = np.zeros(0)
        self.arena = self.psi = self.sigma_ref = self.diag = self.psi_out = None

    @property
    def n_pairs(self):
        return len(self.pairs)

    @property
    def macs(self):
        p = self.pairs
        return int(
            (p["m0"].astype(np.int64) * p["n0"] * p["k0"]).sum()
            + (p["m1"].astype(np.int64) * p["n1"] * p["k1"]).sum()
        )


def read_plan(fn):
    with open(fn, "rb") as f:
        raw = f.read()
    if raw[:8] != b"B2XPLAN1":
        raise ValueError("%s: not a B2XPLAN1 file" % fn)
    hdr = np.frombuffer(raw, "<u8", 8, 8)
    n_pairs, psi_len, sigma_len, max_work, arena_len, n_ranges, flags, n_meta = (int(x) for x in hdr)
    pf = PlanFile()
    pos = 72
    pf.pairs = np.frombuffer(raw, PAIR_DTYPE, n_pairs, pos).copy()
    pos += 96 * n_pairs
    pf.ranges = np.frombuffer(raw, "<u8", 2 * n_ranges, pos).reshape(-1, 2).copy()
    pos += 16 * n_ranges
    pf.meta = np.frombuffer(raw, "<f8", n_meta, pos).copy()
    pos += 8 * n_meta
    pf.psi_len, pf.sigma_len, pf.max_work, pf.arena_len = psi_len, sigma_len, max_work, arena_len

    def take(n):
        nonlocal pos
        a = np.frombuffer(raw, "<f8", n, pos).copy()
        pos += 8 * n
        return a

    if flags & F_ARENA:
        pf.arena = take(arena_len)
    if flags & F_PSI:
        pf.psi = take(psi_len)
    if flags & F_SIGMA:
        pf.sigma_ref = take(sigma_len)
    if flags & F_DIAG:
        pf.diag = take(psi_len)
    if flags & F_PSIOUT:
        pf.psi_out = take(psi_len)
    return pf


def write_plan(fn, pf):
    flags = 0
    for bit, arr in ((F_ARENA, pf.arena), (F_PSI, pf.psi), (F_SIGMA, pf.sigma_ref), (F_DIAG, pf.diag),
                     (F_PSIOUT, pf.psi_out)):
        if arr is not None:
            flags |= bit
    hdr = np.array([pf.n_pairs, pf.psi_len, pf.sigma_len, pf.max_work, pf.arena_len, len(pf.ranges), flags,
                    len(pf.meta)], "<u8")
    with open(fn, "wb") as f:
        f.write(b"B2XPLAN1")
        f.write(hdr.tobytes())
        f.write(np.ascontiguousarray(pf.pairs, PAIR_DTYPE).tobytes())
        f.write(np.ascontiguousarray(pf.ranges, "<u8").tobytes())
        f.write(np.ascontiguousarray(pf.meta, "<f8").tobytes())
        for arr in (pf.arena, pf.psi, pf.sigma_ref, pf.diag, pf.psi_out):
            if arr is not None:
                f.write(np.ascontiguousarray(arr, "<f8").tobytes())


def write_struct_npz(fn, pf):
    """Structure-only plan (no operator / psi data) as compressed columns — a 100k-pair plan is < 1 MB."""
    cols = {n: pf.pairs[n] for n in PAIR_DTYPE.names if n != "reserved"}
    np.savez_compressed(fn, psi_len=pf.psi_len, sigma_len=pf.sigma_len, arena_len=pf.arena_len,
                        max_work=pf.max_work, meta=pf.meta, **cols)


def read_struct_npz(fn):
    z = np.load(fn, allow_pickle=False)
    pf = PlanFile()
    n = len(z["m0"])
    pf.pairs = np.zeros(n, PAIR_DTYPE)
    for name in PAIR_DTYPE.names:
        if name != "reserved":
            pf.pairs[name] = z[name]
    pf.psi_len, pf.sigma_len = int(z["psi_len"]), int(z["sigma_len"])
    pf.arena_len, pf.max_work = int(z["arena_len"]), int(z["max_work"])
    pf.meta = z["meta"].copy()
    return pf


_ARR_DT = {0: "<u8", 1: "<i8", 2: "<f8", 3: "<u4", 4: "u1"}


def read_arrays(fn):
    """Named-array container B2XARR01 (oracle/ref_dump.cpp: effective-Hamiltonian level fixtures) -> dict."""
    raw = open(fn, "rb").read()
    if raw[:8] != b"B2XARR01":
        raise ValueError("%s: not a B2XARR01 file" % fn)
    pos, out = 8, {}
    while pos < len(raw):
        ln = int(np.frombuffer(raw, "<u4", 1, pos)[0])
        pos += 4
        name = raw[pos:pos + ln].decode()
        pos += ln
        dt = _ARR_DT[raw[pos]]
        pos += 1
        cnt = int(np.frombuffer(raw, "<u8", 1, pos)[0])
        pos += 8
        out[name] = np.frombuffer(raw, dt, cnt, pos).copy()
        pos += cnt * np.dtype(dt).itemsize
    return out


class GemmList:
    """A captured single-GEMM list (perturbative noise, oracle/ref_dump.cpp capture_pnoise)."""
    gemms = None
    arena_len = in_len = out_len = 0
    macs = 0
    forward = 0
    out_offsets = out_lens = None
    arena = vin = out_ref = None


def read_gemm_list(fn):
    """.pnoise (B2XARR01 with data) or .pnoise_struct.npz (structure only) -> GemmList"""
    gl = GemmList()
    if fn.endswith(".npz"):
        z = np.load(fn, allow_pickle=False)
        gl.gemms = np.zeros(len(z["m"]), GEMM_DTYPE)
        for name in GEMM_DTYPE.names:
            if name != "reserved":
                gl.gemms[name] = z[name]
        lens = z["lens"]
        gl.out_offsets, gl.out_lens = z["out_offsets"].copy(), z["out_lens"].copy()
    else:
        d = read_arrays(fn)
        gl.gemms = np.frombuffer(d["gemms"].tobytes(), GEMM_DTYPE).copy()
        lens = d["lens"]
        gl.out_offsets, gl.out_lens = d["out.offsets"], d["out.lens"]
        gl.arena, gl.vin, gl.out_ref = d.get("arena"), d.get("in"), d.get("out_ref")
    assert int(lens[0]) == len(gl.gemms)
    gl.arena_len, gl.in_len, gl.out_len = int(lens[1]), int(lens[2]), int(lens[3])
    gl.macs, gl.forward = int(lens[5]), int(lens[6])
    return gl


def write_gemm_struct_npz(fn, gl):
    cols = {n: gl.gemms[n] for n in GEMM_DTYPE.names if n != "reserved"}
    lens = np.array([len(gl.gemms), gl.arena_len, gl.in_len, gl.out_len, len(gl.out_lens), gl.macs, gl.forward], "<u8")
    np.savez_compressed(fn, lens=lens, out_offsets=gl.out_offsets, out_lens=gl.out_lens, **cols)


def write_outer_struct_npz(fn, terms, lens):
    """blocking term list without data (structure of a Cr2-size blocking step) as compressed columns"""
    cols = {n: terms[n] for n in OUTER_TERM_DTYPE.names if n != "reserved"}
    np.savez_compressed(fn, lens=np.asarray(lens, "<u8"), **cols)


def read_outer_struct_npz(fn):
    z = np.load(fn, allow_pickle=False)
    t = np.zeros(len(z["m"]), OUTER_TERM_DTYPE)
    for name in OUTER_TERM_DTYPE.names:
        if name != "reserved":
            t[name] = z[name]
    return t, z["lens"].copy()


def partition_filename(save_dir, prefix_distri, tag, left, i, info=False):
    """scratch file of a renormalised block (MovingEnvironment::get_left/right_partition_filename,
    src/dmrg/moving_environment.hpp:857-880): <save_dir>/<prefix_distri>.PART.[INFO.]<tag>.LEFT|RIGHT.<i>; prefix_distri is
    "F<rank>" under a parallel rule (src/core/parallel_rule.hpp:340), "F0" otherwise"""
    return "%s/%s.PART.%s%s.%s.%d" % (save_dir, prefix_distri, "INFO." if info else "", tag, "LEFT" if left else "RIGHT", i)


def mps_tensor_filename(save_dir, prefix, tag, i):
    """scratch file of MPS tensor i (MPS::get_filename, src/dmrg/mps.hpp): <save_dir>/<prefix>.MPS.<tag>.<i>; i = -1 holds
    the canonical form and center"""
    return "%s/%s.MPS.%s.%d" % (save_dir, prefix, tag, i)


def read_partition_file(fn, fp_prec=None, fp_chunk=1024):
    """content of a partition file of MovingEnvironment (frame_->save_data(1, get_left/right_partition_filename(i)),
    src/dmrg/moving_environment.hpp:428-440; DataFrame::save_data_to / load_data_from, src/core/allocator.hpp:518-530, 580-592):
    the frame's two stacks as they stand —
        size_t  words used of the integer stack      size_t  elements used of the double stack
        uint32 x that many   (the SparseMatrixInfo arrays of the block's operators, in allocation order)
        double x that many   (the renormalised operator blocks, in allocation order; with DataFrame::fp_codec set: the
                              FPCodec::write_array stream of them instead, src/core/fp_codec.hpp)
    Returns (integer stack, double stack)."""
    import struct

    raw = open(fn, "rb").read()
    if len(raw) < 16:
        raise ValueError("%s: too short for a partition file" % fn)
    iused, dused = struct.unpack_from("<QQ", raw, 0)
    off = 16 + 4 * iused
    if off > len(raw):
        raise ValueError("%s: integer stack of %d words does not fit the file" % (fn, iused))
    istack = np.frombuffer(raw, np.uint32, iused, 16).copy()
    if fp_prec is None:
        if len(raw) != off + 8 * dused:
            raise ValueError("%s: %d bytes, header says %d" % (fn, len(raw), off + 8 * dused))
        dstack = np.frombuffer(raw, np.float64, dused, off).copy()
    else:
        from . import b2x_host

        dstack = np.asarray(b2x_host.fpcodec_decode(raw[off:], int(dused)), np.float64)
    return istack, dstack


def write_partition_file(fn, istack, dstack, fp_prec=None, fp_chunk=1024):
    """the inverse of read_partition_file: byte-identical to what the reference writes for the same stacks"""
    import struct

    istack = np.ascontiguousarray(istack, np.uint32)
    dstack = np.ascontiguousarray(dstack, np.float64)
    with open(fn, "wb") as f:
        f.write(struct.pack("<QQ", istack.size, dstack.size))
        f.write(istack.tobytes())
        if fp_prec is None:
            f.write(dstack.tobytes())
        else:
            from . import b2x_host

            f.write(bytes(b2x_host.fpcodec_encode(dstack, fp_prec, fp_chunk)))
