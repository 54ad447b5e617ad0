// b2x_host.hpp — C++ host-side mirror of block2's interface for the H·psi path, above the C ABI.
//
// Same names, argument meaning and error behaviour as the reference classes this path sits behind, so a
// block2 user (and the parity tests) can drive the MI355X path the way they drive block2:
//
//   GMatrix                      src/core/matrix.hpp:92-107          row-major view {data, m, n}
//   SeqTypes                     src/core/threading.hpp:105-135
//   BatchGEMMSeq::rotate         src/core/batch_gemm.hpp:893-902     records one GEMM pair
//   BatchGEMMSeq::three_rotate   src/core/batch_gemm.hpp:952-1022    records one sliced pair
//   BatchGEMMSeq::operator()     src/core/batch_gemm.hpp:1563-1684   replays the plan: v += scale * H c
//   BatchGEMMSeq::rotate + rotate_perform        environment rotation c = bra^T a ket of operator blocks: the pairs that
//                                OperatorFunctions::tensor_rotate records in SeqTypes::Auto (operator_functions.hpp:175-210)
//   BatchGEMMSeq::tensor_product / iadd + outer_perform   blocking: c = a (x) b block products and operator sums
//                                src/core/batch_gemm.hpp:872-885, 1130-1136 -> AdvancedGEMM::tensor_product :431-505
//   BatchGEMMSeq::multiply / three_rotate_tr_left / three_rotate_tr_right / auto_perform(v)
//                                src/core/batch_gemm.hpp:887-891, 1025-1109, 1410-1455   single-GEMM lists (noise)
//   IterativeMatrixFunctions::davidson          src/core/iterative_matrix_functions.hpp:864-1173
//   IterativeMatrixFunctions::olsen_precondition                     :93-108
//   EffectiveHamiltonian::{precompute, operator(), eigs, post_precompute}
//                                src/dmrg/effective_hamiltonian.hpp:224-251, 449-467, 470-558
//
// Everything numerical happens on the device through include/b2x.h; a non-zero return becomes
// std::runtime_error (the reference throws at this boundary).  Header-only, like the reference.
#pragma once
#include "../../../include/b2x.h"
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace b2xh {

inline void check(int rc) {
    if (rc != 0)
        throw std::runtime_error(std::string("b2x: ") + b2x_last_error());
}

struct GMatrix {
    double *data;
    int m, n;
    GMatrix(double *data, int m, int n) : data(data), m(m), n(n) {}
    size_t size() const { return (size_t)m * n; }
    GMatrix flip_dims() const { return GMatrix(data, n, m); }
    GMatrix shift_ptr(size_t l) const { return GMatrix(data + l, m, n); }
    double &operator()(int i, int j) const { return data[(size_t)i * n + j]; }
};

enum struct SeqTypes : uint8_t { None = 0, Simple = 1, Auto = 2, Tasked = 4, SimpleTasked = 5, Device = 8 };

// Plan recorder + device executor.  As in the reference, psi / psi' operands are recorded as OFFSETS
// ("pointers from null": precompute() sets cmat->data = vmat->data = 0), operator operands as absolute
// host pointers; the operator ranges are uploaded once, when the plan is first executed.
// Threading (src/core/threading.hpp:105-135, 137-): the one field of the global threading scheme this path reads —
// OperatorFunctions takes the mode of its sequence from it (src/core/operator_functions.hpp:73-75).
struct Threading {
    SeqTypes seq_type = SeqTypes::Device;
};
inline std::shared_ptr<Threading> &threading_() {
    static std::shared_ptr<Threading> t = std::make_shared<Threading>();
    return t;
}

// ParallelCommunicator<S> (src/core/parallel_rule.hpp:38-308): size / rank / root, tcomm, and the three collectives the
// H.psi path uses, on DEVICE-resident fp64 vectors.  The base class is the serial communicator of the reference: its
// collectives must not be called (the reference asserts false, :56-307).
struct ParallelCommunicator {
    int size, rank, root;
    double tcomm = 0.0; // seconds spent in collectives
    ParallelCommunicator(int size = 1, int rank = 0, int root = 0) : size(size), rank(rank), root(root) {}
    virtual ~ParallelCommunicator() = default;
    bool is_root() const { return rank == root; }
    virtual void allreduce_sum(double *, size_t) { throw std::runtime_error("ParallelCommunicator::allreduce_sum: serial communicator"); }
    virtual void broadcast(double *, size_t, int) { throw std::runtime_error("ParallelCommunicator::broadcast: serial communicator"); }
    virtual void barrier() { throw std::runtime_error("ParallelCommunicator::barrier: serial communicator"); }
    virtual b2x_comm *handle() const { return nullptr; }
};
// the MPICommunicator of this build (src/core/parallel_mpi.hpp:300-309, 133-141, 125-132): RCCL over xGMI through the C ABI
struct RCCLCommunicator : ParallelCommunicator {
    b2x_comm *comm = nullptr;
    RCCLCommunicator(int rank, int size, const std::string &id_file, int root = 0) : ParallelCommunicator(size, rank, root) {
        check(b2x_comm_init(&comm, rank, size, id_file.c_str()));
    }
    ~RCCLCommunicator() override {
        if (comm)
            b2x_comm_destroy(comm);
    }
    RCCLCommunicator(const RCCLCommunicator &) = delete;
    RCCLCommunicator &operator=(const RCCLCommunicator &) = delete;
    template <typename F> void timed(F f) {
        auto t0 = std::chrono::steady_clock::now();
        f();
        tcomm += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void allreduce_sum(double *dev, size_t n) override {
        timed([&]() { check(b2x_allreduce_sum(comm, dev, n, nullptr)); });
    }
    void broadcast(double *dev, size_t n, int owner) override {
        timed([&]() { check(b2x_broadcast(comm, dev, n, owner, nullptr)); });
    }
    void barrier() override {
        timed([&]() { check(b2x_barrier(comm)); });
    }
    b2x_comm *handle() const override { return comm; }
};

struct BatchGEMMSeq {
    SeqTypes mode = SeqTypes::Device;
    std::vector<b2x_pair> pairs;
    std::vector<const double *> y_ptr, z_ptr; // absolute operator pointers of stage 0 / stage 1
    size_t max_work = 0;
    size_t nflop = 0, cumulative_nflop = 0;
    b2x_arena *arena = nullptr;
    b2x_plan *plan = nullptr;
    size_t psi_len = 0, sigma_len = 0;
    BatchGEMMSeq(size_t /*max_batch_flops*/ = 1LL << 24, SeqTypes mode = threading_()->seq_type) : mode(mode) {}
    ~BatchGEMMSeq() { deallocate(); }
    BatchGEMMSeq(const BatchGEMMSeq &) = delete;
    BatchGEMMSeq &operator=(const BatchGEMMSeq &) = delete;

    void push(int ta0, int tb0, int m0, int n0, int k0, double alpha0, const double *a0, int lda0, const double *b0,
              int ldb0, int ta1, int m1, int k1, double alpha1, const double *a1, int lda1, double *c1, int ldc1) {
        if (plan != nullptr)
            throw std::runtime_error("BatchGEMMSeq: plan already uploaded; call clear() first");
        b2x_pair p{};
        p.m0 = m0, p.n0 = n0, p.k0 = k0, p.lda0 = lda0, p.ldb0 = ldb0;
        p.m1 = m1, p.n1 = n0, p.k1 = k1, p.lda1 = lda1, p.ldc1 = ldc1;
        p.ta0 = (uint8_t)ta0, p.tb0 = (uint8_t)tb0, p.ta1 = (uint8_t)ta1, p.tb1 = 0;
        p.alpha0 = alpha0, p.alpha1 = alpha1;
        p.x_off = (uint64_t)(a0 - (const double *)0);
        p.v_off = (uint64_t)(c1 - (double *)0);
        pairs.push_back(p);
        y_ptr.push_back(b0), z_ptr.push_back(a1);
        max_work = std::max(max_work, (size_t)m0 * n0);
        nflop += (size_t)m0 * n0 * k0 + (size_t)m1 * n0 * k1;
    }
    // [c] += scale * op(bra) x [a] x op(ket)      conj & 1 == transpose (real double)
    void rotate(const GMatrix &a, const GMatrix &c, const GMatrix &bra, uint8_t conj_bra, const GMatrix &ket,
                uint8_t conj_ket, double scale) {
        const int tb0 = conj_ket & 1, ta1 = conj_bra & 1;
        const int wn = tb0 ? ket.m : ket.n, k0 = tb0 ? ket.n : ket.m;
        push(0, tb0, a.m, wn, k0, 1.0, a.data, a.n, ket.data, ket.n, ta1, c.m, a.m, scale, bra.data, bra.n, c.data,
             c.n);
    }
    //  dleft: [c] = scale * [bra] (= [da] x [db]) * [a] * [ket]
    // !dleft: [c] = scale * [bra] * [a] * [ket] (= [da] x [db])     one of da / db is 1 x 1
    void three_rotate(const GMatrix &a, const GMatrix &c, const GMatrix &bra, bool conj_bra, const GMatrix &ket,
                      bool conj_ket, const GMatrix &da, bool dconja, const GMatrix &db, bool dconjb, bool dleft,
                      double scale, uint64_t stride) {
        const bool a_scalar = da.m == 1 && da.n == 1, b_scalar = db.m == 1 && db.n == 1;
        if (!a_scalar && !b_scalar)
            throw std::runtime_error("three_rotate: one factor of the delayed operator must be 1 x 1");
        if (dleft) {
            dconja ^= conj_bra, dconjb ^= conj_bra;
            int am = (dconja ? da.m : da.n) * (dconjb ? db.m : db.n);
            int cm = (dconja ? da.n : da.m) * (dconjb ? db.n : db.m);
            uint32_t ast = (uint32_t)(conj_bra ? stride / bra.n : stride % bra.n);
            uint32_t cst = (uint32_t)(conj_bra ? stride % bra.n : stride / bra.n);
            const int tb0 = conj_ket ? 1 : 0;
            const int wn = conj_ket ? ket.m : ket.n, k0 = conj_ket ? ket.n : ket.m;
            const GMatrix &big = a_scalar ? db : da;
            const bool bconj = a_scalar ? dconjb : dconja;
            const double sc = a_scalar ? *da.data : *db.data;
            push(0, tb0, am, wn, k0, scale, a.data + (size_t)ast * a.n, a.n, ket.data, ket.n, bconj ? 1 : 0, cm, am, sc,
                 big.data, big.n, c.data + (size_t)cst * c.n, c.n);
        } else {
            dconja ^= conj_ket, dconjb ^= conj_ket;
            int kn = (dconja ? da.m : da.n) * (dconjb ? db.m : db.n);
            int km = (dconja ? da.n : da.m) * (dconjb ? db.n : db.m);
            uint32_t ast = (uint32_t)(conj_ket ? stride % ket.n : stride / ket.n);
            uint32_t cst = (uint32_t)(conj_ket ? stride / ket.n : stride % ket.n);
            const GMatrix &big = a_scalar ? db : da;
            const bool bconj = a_scalar ? dconjb : dconja;
            const double sc = a_scalar ? *da.data : *db.data;
            // work = a[:, ast:ast+km] * op(big): conj flag 1 = transpose, 2 = plain (batch_gemm.hpp:989-1003)
            push(0, bconj ? 1 : 0, a.m, kn, km, sc, a.data + ast, a.n, big.data, big.n, conj_bra ? 1 : 0, c.m, a.m,
                 scale, bra.data, bra.n, c.data + cst, c.n);
        }
    }
    // ---- diagonal of H_eff: rank-1 products of operator-block diagonals -------------------------------------
    // c[i][j] += scale * a[i][i] * b[j][j]     (AdvancedGEMM::tensor_product_diagonal, batch_gemm.hpp:507-524)
    std::vector<b2x_diag_term> diag_terms;
    std::vector<const double *> da_ptr, db_ptr;
    void push_diag(const double *a, int a_stride, int m, const double *b, int b_stride, int n, double *c, int ldc,
                   double alpha) {
        b2x_diag_term t{};
        t.m = m, t.n = n, t.a_stride = a_stride, t.b_stride = b_stride, t.ldc = ldc, t.alpha = alpha;
        t.c_off = (uint64_t)(c - (double *)0);
        diag_terms.push_back(t);
        da_ptr.push_back(a), db_ptr.push_back(b);
    }
    void tensor_product_diagonal(uint8_t /*conj*/, const GMatrix &a, const GMatrix &b, const GMatrix &c, double scale) {
        if (a.m != a.n || b.m != b.n || c.m != a.n || c.n != b.n)
            throw std::runtime_error("tensor_product_diagonal: shape mismatch");
        push_diag(a.data, a.n + 1, a.n, b.data, b.n + 1, b.n, c.data, c.n, scale);
    }
    // diagonal of (da x db) x b  or  a x (da x db), one of da / db being 1 x 1 (matrix_functions.hpp:1189-1240):
    // only a diagonal sub-block (row offset == column offset of `stride`) contributes
    void three_tensor_product_diagonal(uint8_t /*conj*/, const GMatrix &a, const GMatrix &b, const GMatrix &c,
                                       const GMatrix &da, bool /*dconja*/, const GMatrix &db, bool /*dconjb*/, bool dleft,
                                       double scale, uint64_t stride) {
        const int dstrm = (int)(stride / (uint64_t)(dleft ? a.m : b.m)), dstrn = (int)(stride % (uint64_t)(dleft ? a.m : b.m));
        if (dstrn != dstrm)
            return;
        const bool a_scalar = da.m == 1 && da.n == 1;
        if (!a_scalar && !(db.m == 1 && db.n == 1))
            throw std::runtime_error("three_tensor_product_diagonal: one factor must be 1 x 1");
        const GMatrix &big = a_scalar ? db : da;
        const double sc = scale * (a_scalar ? *da.data : *db.data);
        if (dleft) // rows [dstr, dstr + big.n) of c
            push_diag(big.data, big.n + 1, big.n, b.data, b.n + 1, b.n, c.data + (size_t)dstrn * c.n, c.n, sc);
        else // columns [dstr, dstr + big.n) of c
            push_diag(a.data, a.n + 1, a.n, big.data, big.n + 1, big.n, c.data + dstrn, c.n, sc);
    }
    // execute the recorded diagonal terms: diag (host vector) += terms
    void diag_perform(double *diag, size_t len) {
        if (diag_terms.empty())
            return;
        std::vector<std::pair<const double *, size_t>> ext;
        for (size_t i = 0; i < diag_terms.size(); i++) {
            ext.emplace_back(da_ptr[i], (size_t)(diag_terms[i].m - 1) * diag_terms[i].a_stride + 1);
            ext.emplace_back(db_ptr[i], (size_t)(diag_terms[i].n - 1) * diag_terms[i].b_stride + 1);
        }
        std::sort(ext.begin(), ext.end());
        std::vector<const double *> bases;
        std::vector<size_t> lens;
        for (auto &e : ext) {
            if (!bases.empty() && e.first <= bases.back() + lens.back())
                lens.back() = std::max(lens.back(), (size_t)(e.first - bases.back()) + e.second);
            else
                bases.push_back(e.first), lens.push_back(e.second);
        }
        b2x_arena *ar = nullptr;
        check(b2x_arena_create(&ar, bases.size(), bases.data(), lens.data()));
        std::vector<b2x_diag_term> terms = diag_terms;
        for (size_t i = 0; i < terms.size(); i++) {
            check(b2x_arena_resolve(ar, da_ptr[i], &terms[i].a_off));
            check(b2x_arena_resolve(ar, db_ptr[i], &terms[i].b_off));
        }
        int rc = b2x_diag_build(ar, terms.size(), terms.data(), len, diag, 0, nullptr);
        b2x_arena_destroy(ar);
        check(rc);
        diag_terms.clear(), da_ptr.clear(), db_ptr.clear();
    }
    // Replay recorded rotate() calls whose a / c operands are ABSOLUTE host blocks (environment rotation: a = blocks of
    // the enlarged operators, c = blocks of the rotated operators; the reference's Auto-mode auto_perform(),
    // batch_gemm.hpp:1411-1416).  The a blocks are packed into one device vector (the plan's "psi"), the c blocks into
    // another ("psi'"), bra / ket blocks form the arena; c += result (beta = 1, as recorded), then the list is cleared.
    void rotate_perform(const std::vector<std::pair<const double *, size_t>> &a_blocks,
                        const std::vector<std::pair<double *, size_t>> &c_blocks) {
        if (pairs.empty())
            return;
        struct Packed {
            std::vector<const double *> starts;
            std::vector<size_t> lens, offs;
            size_t tot = 0;
            void build(std::vector<std::pair<const double *, size_t>> r) {
                std::sort(r.begin(), r.end());
                for (auto &e : r) {
                    if (!starts.empty() && e.first < starts.back() + lens.back()) {
                        if (e.first + e.second > starts.back() + lens.back())
                            throw std::runtime_error("rotate_perform: partially overlapping blocks");
                        continue;
                    }
                    starts.push_back(e.first), lens.push_back(e.second), offs.push_back(tot), tot += e.second;
                }
            }
            uint64_t resolve(const double *p) const {
                size_t r = std::upper_bound(starts.begin(), starts.end(), p) - starts.begin();
                if (r == 0 || p >= starts[r - 1] + lens[r - 1])
                    throw std::runtime_error("rotate_perform: operand outside the given blocks");
                return offs[r - 1] + (uint64_t)(p - starts[r - 1]);
            }
        } X, V;
        X.build(a_blocks);
        {
            std::vector<std::pair<const double *, size_t>> cb;
            for (auto &e : c_blocks)
                cb.emplace_back(e.first, e.second);
            V.build(cb);
        }
        for (b2x_pair &p : pairs) {
            p.x_off = X.resolve((const double *)0 + p.x_off);
            p.v_off = V.resolve((const double *)0 + p.v_off);
        }
        std::vector<double> x(X.tot), v(V.tot);
        for (size_t r = 0; r < X.starts.size(); r++)
            std::copy(X.starts[r], X.starts[r] + X.lens[r], x.begin() + X.offs[r]);
        for (size_t r = 0; r < V.starts.size(); r++)
            std::copy(V.starts[r], V.starts[r] + V.lens[r], v.begin() + V.offs[r]);
        (*this)(GMatrix(x.data(), (int)x.size(), 1), GMatrix(v.data(), (int)v.size(), 1), 1.0);
        for (size_t r = 0; r < V.starts.size(); r++)
            std::copy(v.begin() + V.offs[r], v.begin() + V.offs[r] + V.lens[r], const_cast<double *>(V.starts[r]));
        clear();
    }
    // ---- element-wise block products (blocking of the environments) -----------------------------------------------
    std::vector<b2x_outer_term> outer_terms;
    std::vector<const double *> oa_ptr, ob_ptr;
    std::vector<double *> oc_ptr;
    // C[r][c] += alpha * A[r*a_rs + c*a_cs] * B[r*b_rs + c*b_cs]; a null operand pointer means the constant 1.0
    void push_outer(int m, int n, const double *a, int a_rs, int a_cs, const double *b, int b_rs, int b_cs, double *c,
                    int ldc, double alpha) {
        b2x_outer_term t{};
        t.m = m, t.n = n, t.a_rs = a_rs, t.a_cs = a_cs, t.b_rs = b_rs, t.b_cs = b_cs, t.ldc = ldc, t.alpha = alpha;
        t.a_src = a ? 0 : 2, t.b_src = b ? 0 : 2;
        outer_terms.push_back(t);
        oa_ptr.push_back(a), ob_ptr.push_back(b), oc_ptr.push_back(c);
    }
    // [c + stride] += scale * op(a) (x) op(b)   (BatchGEMMSeq::tensor_product, batch_gemm.hpp:1130-1136; the window of
    // c starts `stride` elements after c(0, 0) and keeps c's leading dimension).  conj = transpose (real double).
    void tensor_product(const GMatrix &a, bool conja, const GMatrix &b, bool conjb, const GMatrix &c, double scale,
                        uint64_t stride) {
        double *cw = c.data + stride;
        const int am = conja ? a.n : a.m, an = conja ? a.m : a.n; // shape of op(a)
        const int bm = conjb ? b.n : b.m, bn = conjb ? b.m : b.n;
        const int a_rs = conja ? 1 : a.n, a_cs = conja ? a.n : 1; // op(a)[i][j] = a.data[i*a_rs + j*a_cs]
        const int b_rs = conjb ? 1 : b.n, b_cs = conjb ? b.n : 1;
        if (a.m == 1 && a.n == 1)
            push_outer(bm, bn, b.data, b_rs, b_cs, a.data, 0, 0, cw, c.n, scale);
        else if (b.m == 1 && b.n == 1)
            push_outer(am, an, a.data, a_rs, a_cs, b.data, 0, 0, cw, c.n, scale);
        else // general Kronecker product: one term per element of op(a), the block op(b) scaled by it
            for (int i = 0; i < am; i++)
                for (int j = 0; j < an; j++)
                    push_outer(bm, bn, b.data, b_rs, b_cs, a.data + (size_t)i * a_rs + (size_t)j * a_cs, 0, 0,
                               cw + (size_t)i * bm * c.n + (size_t)j * bn, c.n, scale);
    }
    // [a] += scale * op(b)   (BatchGEMMSeq::iadd, batch_gemm.hpp:872-881; cfactor must be 1 on this path)
    void iadd(const GMatrix &a, const GMatrix &b, double scale = 1.0, bool conj = false, double cfactor = 1.0) {
        if (cfactor != 1.0)
            throw std::runtime_error("BatchGEMMSeq::iadd: only accumulation (cfactor == 1) is recorded on this path");
        if (!conj) // (recorded with the block's own shape: plain and transposed sums into one block then share a row grid)
            push_outer(a.m, a.n, b.data, a.n, 1, nullptr, 0, 0, a.data, a.n, scale);
        else
            push_outer(b.n, b.m, b.data, 1, b.n, nullptr, 0, 0, a.data, a.n, scale);
    }
    // Execute the recorded block products: operands are absolute host blocks (block operators, site operators), the
    // outputs lie inside `c_blocks` (the enlarged operators), which are packed into one device vector, accumulated
    // into and copied back.  Clears the list.
    void outer_perform(const std::vector<std::pair<double *, size_t>> &c_blocks) {
        if (outer_terms.empty())
            return;
        std::vector<std::pair<const double *, size_t>> ext;
        for (size_t i = 0; i < outer_terms.size(); i++) {
            const b2x_outer_term &t = outer_terms[i];
            if (oa_ptr[i])
                ext.emplace_back(oa_ptr[i], (size_t)(t.m - 1) * t.a_rs + (size_t)(t.n - 1) * t.a_cs + 1);
            if (ob_ptr[i])
                ext.emplace_back(ob_ptr[i], (size_t)(t.m - 1) * t.b_rs + (size_t)(t.n - 1) * t.b_cs + 1);
        }
        std::sort(ext.begin(), ext.end());
        std::vector<const double *> bases;
        std::vector<size_t> lens;
        for (auto &e : ext) {
            if (!bases.empty() && e.first <= bases.back() + lens.back())
                lens.back() = std::max(lens.back(), (size_t)(e.first - bases.back()) + e.second);
            else
                bases.push_back(e.first), lens.push_back(e.second);
        }
        static const double zero = 0.0;
        if (bases.empty())
            bases.push_back(&zero), lens.push_back(1);
        // pack the output blocks
        std::vector<std::pair<double *, size_t>> cb = c_blocks;
        std::sort(cb.begin(), cb.end());
        std::vector<size_t> coff(cb.size());
        size_t ctot = 0;
        for (size_t r = 0; r < cb.size(); r++)
            coff[r] = ctot, ctot += cb[r].second;
        auto cres = [&](double *p) -> uint64_t {
            size_t r = std::upper_bound(cb.begin(), cb.end(), std::make_pair(p, (size_t)-1)) - cb.begin();
            if (r == 0 || p >= cb[r - 1].first + cb[r - 1].second)
                throw std::runtime_error("outer_perform: output outside the given blocks");
            return coff[r - 1] + (uint64_t)(p - cb[r - 1].first);
        };
        b2x_arena *ar = nullptr;
        check(b2x_arena_create(&ar, bases.size(), bases.data(), lens.data()));
        int rc = 0;
        try {
            for (size_t i = 0; i < outer_terms.size() && rc == 0; i++) {
                if (oa_ptr[i])
                    rc = b2x_arena_resolve(ar, oa_ptr[i], &outer_terms[i].a_off);
                if (ob_ptr[i] && rc == 0)
                    rc = b2x_arena_resolve(ar, ob_ptr[i], &outer_terms[i].b_off);
                outer_terms[i].c_off = cres(oc_ptr[i]);
            }
        } catch (...) {
            b2x_arena_destroy(ar);
            throw;
        }
        std::vector<double> v(ctot);
        for (size_t r = 0; r < cb.size(); r++)
            std::copy(cb[r].first, cb[r].first + cb[r].second, v.begin() + coff[r]);
        if (rc == 0)
            rc = b2x_outer_build(ar, outer_terms.size(), outer_terms.data(), nullptr, 0, v.size(), v.data(), 0, nullptr);
        b2x_arena_destroy(ar);
        check(rc);
        for (size_t r = 0; r < cb.size(); r++)
            std::copy(v.begin() + coff[r], v.begin() + coff[r] + cb[r].second, cb[r].first);
        outer_terms.clear(), oa_ptr.clear(), ob_ptr.clear(), oc_ptr.clear();
    }
    // ---- single-GEMM lists (perturbative noise): batch[1]-only records, replayed by auto_perform(v) -----------
    std::vector<b2x_gemm> gemms;
    std::vector<const double *> ga_ptr, gb_ptr;
    std::vector<double *> gc_ptr;
    size_t gemm_nflop = 0;
    // one xgemm slot (batch_gemm.hpp:313-320): c(m x n) += alpha op(a) op(b)
    void push_gemm(int ta, int tb, int m, int n, int k, double alpha, const double *a, int lda, const double *b, int ldb,
                   double *c, int ldc) {
        b2x_gemm g{};
        g.m = m, g.n = n, g.k = k, g.lda = lda, g.ldb = ldb, g.ldc = ldc;
        g.ta = (uint8_t)ta, g.tb = (uint8_t)tb, g.alpha = alpha;
        gemms.push_back(g);
        ga_ptr.push_back(a), gb_ptr.push_back(b), gc_ptr.push_back(c);
        gemm_nflop += (size_t)m * n * k;
    }
    // [c] = scale * op(a) x op(b) + cfactor * [c]      conj & 1 == transpose (BatchGEMM::multiply, :329-337)
    void multiply(const GMatrix &a, uint8_t conja, const GMatrix &b, uint8_t conjb, const GMatrix &c, double scale,
                  double cfactor) {
        if (cfactor != 1.0)
            throw std::runtime_error("BatchGEMMSeq::multiply: only accumulation (cfactor == 1) is recorded on this path");
        push_gemm(conja & 1, conjb & 1, c.m, (conjb & 1) ? b.m : b.n, (conjb & 1) ? b.n : b.m, scale, a.data, a.n, b.data,
                  b.n, c.data, c.n);
    }
    // [c] = scale * [a] x op(ket)   with the bra side traced out;  dleft: the row slice (ast, am) of a -> (cst, cm) of c;
    // !dleft: ket = da x db with one 1 x 1 factor, column slices (batch_gemm.hpp:1025-1064)
    void three_rotate_tr_left(const GMatrix &a, const GMatrix &c, const GMatrix &bra, bool conj_bra, const GMatrix &ket,
                              bool conj_ket, const GMatrix &da, bool dconja, const GMatrix &db, bool dconjb, bool dleft,
                              double scale, uint64_t stride) {
        if (dleft) {
            dconja ^= conj_bra, dconjb ^= conj_bra;
            int am = (dconja ? da.m : da.n) * (dconjb ? db.m : db.n);
            int cm = (dconja ? da.n : da.m) * (dconjb ? db.n : db.m);
            uint32_t ast = (uint32_t)(conj_bra ? stride / bra.n : stride % bra.n);
            uint32_t cst = (uint32_t)(conj_bra ? stride % bra.n : stride / bra.n);
            multiply(GMatrix(a.data + (size_t)ast * a.n, am, a.n), false, ket, conj_ket ? 1 : 2,
                     GMatrix(c.data + (size_t)cst * c.n, cm, c.n), scale, 1.0);
        } else {
            dconja ^= conj_ket, dconjb ^= conj_ket;
            uint32_t ast = (uint32_t)(conj_ket ? stride % ket.n : stride / ket.n);
            uint32_t cst = (uint32_t)(conj_ket ? stride / ket.n : stride % ket.n);
            const bool a_scalar = da.m == 1 && da.n == 1, b_scalar = db.m == 1 && db.n == 1;
            if (!a_scalar && !b_scalar)
                throw std::runtime_error("three_rotate_tr_left: one factor of the delayed operator must be 1 x 1");
            const GMatrix &big = a_scalar ? db : da;
            const bool bconj = a_scalar ? dconjb : dconja;
            const double sc = a_scalar ? *da.data : *db.data;
            // c[:, cst:] += (scalar * scale) * a[:, ast:] * op(big); the slices keep the parents' leading dimensions
            push_gemm(0, bconj ? 1 : 0, c.m, bconj ? big.m : big.n, bconj ? big.n : big.m, sc * scale, a.data + ast, a.n,
                      big.data, big.n, c.data + cst, c.n);
        }
    }
    //  dleft: [c] = scale * [bra] (= [da] x [db]) * [a];  !dleft: [c] = scale * [bra] * [a]  (batch_gemm.hpp:1066-1109)
    void three_rotate_tr_right(const GMatrix &a, const GMatrix &c, const GMatrix &bra, bool conj_bra, const GMatrix &ket,
                               bool conj_ket, const GMatrix &da, bool dconja, const GMatrix &db, bool dconjb, bool dleft,
                               double scale, uint64_t stride) {
        if (dleft) {
            dconja ^= conj_bra, dconjb ^= conj_bra;
            int am = (dconja ? da.m : da.n) * (dconjb ? db.m : db.n);
            int cm = (dconja ? da.n : da.m) * (dconjb ? db.n : db.m);
            uint32_t ast = (uint32_t)(conj_bra ? stride / bra.n : stride % bra.n);
            uint32_t cst = (uint32_t)(conj_bra ? stride % bra.n : stride / bra.n);
            const bool a_scalar = da.m == 1 && da.n == 1, b_scalar = db.m == 1 && db.n == 1;
            if (!a_scalar && !b_scalar)
                throw std::runtime_error("three_rotate_tr_right: one factor of the delayed operator must be 1 x 1");
            const GMatrix &big = a_scalar ? db : da;
            const bool bconj = a_scalar ? dconjb : dconja;
            const double sc = a_scalar ? *da.data : *db.data;
            multiply(big, bconj ? 3 : 0, GMatrix(a.data + (size_t)ast * a.n, am, a.n), false,
                     GMatrix(c.data + (size_t)cst * c.n, cm, c.n), scale * sc, 1.0);
        } else {
            dconja ^= conj_ket, dconjb ^= conj_ket;
            int kn = (dconja ? da.m : da.n) * (dconjb ? db.m : db.n);
            uint32_t ast = (uint32_t)(conj_ket ? stride % ket.n : stride / ket.n);
            uint32_t cst = (uint32_t)(conj_ket ? stride / ket.n : stride % ket.n);
            push_gemm(conj_bra ? 1 : 0, 0, c.m, kn, a.m, scale, bra.data, bra.n, a.data + ast, a.n, c.data + cst, c.n);
        }
    }
    // Replay the recorded single-GEMM list: v += records (BatchGEMMSeq::auto_perform(v), Tasked branch), then clear.
    // Operands inside `in` (the wavefunction) are read from the input vector; all others are operator blocks.
    void auto_perform(const GMatrix &v, const GMatrix &in = GMatrix(nullptr, 0, 0)) {
        if (gemms.empty())
            return;
        const double *i0 = in.data, *i1 = in.data ? in.data + in.size() : nullptr;
        std::vector<std::pair<const double *, size_t>> ext;
        for (size_t i = 0; i < gemms.size(); i++) {
            b2x_gemm &g = gemms[i];
            size_t ea = g.ta ? (size_t)(g.k - 1) * g.lda + g.m : (size_t)(g.m - 1) * g.lda + g.k;
            size_t eb = g.tb ? (size_t)(g.n - 1) * g.ldb + g.k : (size_t)(g.k - 1) * g.ldb + g.n;
            g.a_src = i0 && ga_ptr[i] >= i0 && ga_ptr[i] < i1, g.b_src = i0 && gb_ptr[i] >= i0 && gb_ptr[i] < i1;
            if (!g.a_src)
                ext.emplace_back(ga_ptr[i], ea);
            if (!g.b_src)
                ext.emplace_back(gb_ptr[i], eb);
            if (gc_ptr[i] < v.data || gc_ptr[i] >= v.data + v.size())
                throw std::runtime_error("BatchGEMMSeq::auto_perform: output pointer outside v");
            g.c_off = (uint64_t)(gc_ptr[i] - v.data);
        }
        std::sort(ext.begin(), ext.end());
        std::vector<const double *> bases;
        std::vector<size_t> lens;
        for (auto &e : ext) {
            if (!bases.empty() && e.first <= bases.back() + lens.back())
                lens.back() = std::max(lens.back(), (size_t)(e.first - bases.back()) + e.second);
            else
                bases.push_back(e.first), lens.push_back(e.second);
        }
        static const double zero = 0.0;
        if (bases.empty())
            bases.push_back(&zero), lens.push_back(1);
        b2x_arena *ar = nullptr;
        check(b2x_arena_create(&ar, bases.size(), bases.data(), lens.data()));
        int rc = 0;
        for (size_t i = 0; i < gemms.size() && rc == 0; i++) {
            if (gemms[i].a_src)
                gemms[i].a_off = (uint64_t)(ga_ptr[i] - i0);
            else
                rc = b2x_arena_resolve(ar, ga_ptr[i], &gemms[i].a_off);
            if (gemms[i].b_src)
                gemms[i].b_off = (uint64_t)(gb_ptr[i] - i0);
            else if (rc == 0)
                rc = b2x_arena_resolve(ar, gb_ptr[i], &gemms[i].b_off);
        }
        b2x_plan *gp = nullptr;
        if (rc == 0)
            rc = b2x_gemm_plan_create(&gp, ar, gemms.size(), gemms.data(), in.size(), v.size(), nullptr);
        if (rc == 0)
            rc = b2x_plan_execute(gp, in.data ? in.data : &zero, v.data, 1.0, 0, nullptr);
        if (gp)
            b2x_plan_destroy(gp);
        b2x_arena_destroy(ar);
        check(rc);
        cumulative_nflop += gemm_nflop;
        gemms.clear(), ga_ptr.clear(), gb_ptr.clear(), gc_ptr.clear();
        gemm_nflop = 0;
    }
    // upload the operator ranges + compile the device plan (lazily, on first execution)
    void prepare(size_t psi_len_, size_t sigma_len_) {
        if (plan != nullptr)
            return;
        psi_len = psi_len_, sigma_len = sigma_len_;
        std::vector<std::pair<const double *, size_t>> ext;
        for (size_t i = 0; i < pairs.size(); i++) {
            const b2x_pair &p = pairs[i];
            size_t ey = p.tb0 ? (size_t)(p.n0 - 1) * p.ldb0 + p.k0 : (size_t)(p.k0 - 1) * p.ldb0 + p.n0;
            size_t ez = p.ta1 ? (size_t)(p.k1 - 1) * p.lda1 + p.m1 : (size_t)(p.m1 - 1) * p.lda1 + p.k1;
            ext.emplace_back(y_ptr[i], ey), ext.emplace_back(z_ptr[i], ez);
        }
        std::sort(ext.begin(), ext.end());
        std::vector<const double *> bases;
        std::vector<size_t> lens;
        for (auto &e : ext) {
            if (!bases.empty() && e.first <= bases.back() + lens.back())
                lens.back() = std::max(lens.back(), (size_t)(e.first - bases.back()) + e.second);
            else
                bases.push_back(e.first), lens.push_back(e.second);
        }
        check(b2x_arena_create(&arena, bases.size(), bases.data(), lens.data()));
        for (size_t i = 0; i < pairs.size(); i++) {
            check(b2x_arena_resolve(arena, y_ptr[i], &pairs[i].y_off));
            check(b2x_arena_resolve(arena, z_ptr[i], &pairs[i].z_off));
        }
        check(b2x_plan_create(&plan, arena, pairs.size(), pairs.data(), psi_len, sigma_len, nullptr));
    }
    // Matrix multiply vector (c) => vector (v):  v += scale * H c   (host pointers)
    void operator()(const GMatrix &c, const GMatrix &v, double scale = 1.0) {
        if (pairs.empty())
            return;
        prepare(c.size(), v.size());
        if (c.size() != psi_len || v.size() != sigma_len)
            throw std::runtime_error("BatchGEMMSeq::operator(): vector length differs from the recorded plan");
        check(b2x_plan_execute(plan, c.data, v.data, scale, 0, nullptr));
        cumulative_nflop += nflop;
    }
    // same with device-resident vectors (what Davidson uses)
    void apply_device(const double *c_dev, double *v_dev, double scale, void *stream = nullptr) {
        if (plan == nullptr)
            throw std::runtime_error("BatchGEMMSeq::apply_device: call prepare() first");
        check(b2x_plan_execute(plan, c_dev, v_dev, scale, 1, stream));
        cumulative_nflop += nflop;
    }
    void deallocate() {
        if (plan)
            b2x_plan_destroy(plan), plan = nullptr;
        if (arena)
            b2x_arena_destroy(arena), arena = nullptr;
    }
    void clear() {
        deallocate();
        pairs.clear(), y_ptr.clear(), z_ptr.clear();
        gemms.clear(), ga_ptr.clear(), gb_ptr.clear(), gc_ptr.clear();
        outer_terms.clear(), oa_ptr.clear(), ob_ptr.clear(), oc_ptr.clear();
        max_work = 0, nflop = 0, gemm_nflop = 0;
    }
};

// device buffer (RAII)
struct DeviceVector {
    double *p = nullptr;
    size_t n = 0;
    explicit DeviceVector(size_t n) : n(n) { check(b2x_device_alloc((void **)&p, std::max<size_t>(n, 1) * 8)); }
    ~DeviceVector() {
        if (p)
            b2x_device_free(p);
    }
    DeviceVector(const DeviceVector &) = delete;
    void upload(const double *h) { check(b2x_memcpy_h2d(p, h, n * 8)); }
    void download(double *h) const { check(b2x_memcpy_d2h(h, p, n * 8)); }
};

// symmetric eigenproblem of the (<= 63 x 63) subspace matrix (the reference calls LAPACK dsyev here): Householder
// reduction to tridiagonal form, then implicit-shift QL; eigenvalues ascending, row j of `a` = eigenvector j on return
// (the layout davidson uses).  Only the lower triangle of `a` is read.  (A cyclic Jacobi solver stood here first: at
// m ~ 30 it cost 0.8 ms per Davidson iteration, more than the whole vector algebra.)
inline void small_eigs(std::vector<double> &a, std::vector<double> &w, int m) {
    const int n = m;
    std::vector<double> V((size_t)n * n), d(n), e(n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++)
            V[(size_t)i * n + j] = V[(size_t)j * n + i] = a[(size_t)i * n + j];
    auto v = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
    // ---- Householder tridiagonalisation (V accumulates the transformation) ----
    for (int j = 0; j < n; j++)
        d[j] = v(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++)
            scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++)
                d[j] = v(i - 1, j), v(i, j) = 0.0, v(j, i) = 0.0;
        } else {
            for (int k = 0; k < i; k++)
                d[k] /= scale, h += d[k] * d[k];
            double f = d[i - 1], g = std::sqrt(h);
            if (f > 0)
                g = -g;
            e[i] = scale * g, h -= f * g, d[i - 1] = f - g;
            for (int j = 0; j < i; j++)
                e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j], v(j, i) = f, g = e[j] + v(j, j) * f;
                for (int k = j + 1; k <= i - 1; k++)
                    g += v(k, j) * d[k], e[k] += v(k, j) * f;
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++)
                e[j] /= h, f += e[j] * d[j];
            const double hh = f / (h + h);
            for (int j = 0; j < i; j++)
                e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j], g = e[j];
                for (int k = j; k <= i - 1; k++)
                    v(k, j) -= (f * e[k] + g * d[k]);
                d[j] = v(i - 1, j), v(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        v(n - 1, i) = v(i, i), v(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++)
                d[k] = v(k, i + 1) / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++)
                    g += v(k, i + 1) * v(k, j);
                for (int k = 0; k <= i; k++)
                    v(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++)
            v(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; j++)
        d[j] = v(n - 1, j), v(n - 1, j) = 0.0;
    if (n > 0)
        v(n - 1, n - 1) = 1.0, e[0] = 0.0;
    // ---- implicit-shift QL on the tridiagonal matrix (d, e) ----
    for (int i = 1; i < n; i++)
        e[i - 1] = e[i];
    if (n > 0)
        e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int mm = l;
        while (mm < n) {
            if (std::fabs(e[mm]) <= eps * tst1)
                break;
            mm++;
        }
        if (mm > l) {
            int iter = 0;
            do {
                iter++;
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = std::hypot(p, 1.0);
                if (p < 0)
                    r = -r;
                d[l] = e[l] / (p + r), d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++)
                    d[i] -= h;
                f += h;
                p = d[mm];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = mm - 1; i >= l; i--) {
                    c3 = c2, c2 = c, s2 = s;
                    g = c * e[i], h = c * p, r = std::hypot(p, e[i]);
                    e[i + 1] = s * r, s = e[i] / r, c = p / r, p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) {
                        h = v(k, i + 1);
                        v(k, i + 1) = s * v(k, i) + c * h, v(k, i) = c * v(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p, d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] += f, e[l] = 0.0;
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++)
        idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return d[x] < d[y]; });
    w.resize(n);
    std::vector<double> out((size_t)n * n);
    for (int j = 0; j < n; j++) {
        w[j] = d[idx[j]];
        for (int i = 0; i < n; i++)
            out[(size_t)j * n + i] = v(i, idx[j]);
    }
    a.swap(out);
}

// DavidsonTypes (src/core/matrix_functions.hpp:273-293): the same names and bit values (they are interface).
enum struct DavidsonTypes : uint16_t {
    Normal = 0,
    GreaterThan = 1,
    LessThan = 2,
    CloseTo = 4,
    Harmonic = 16,
    HarmonicGreaterThan = 16 | 1,
    HarmonicLessThan = 16 | 2,
    HarmonicCloseTo = 16 | 4,
    DavidsonPrecond = 32,
    NoPrecond = 64,
    NonHermitian = 128,
    Exact = 256,
    LeftEigen = 512,
    ElementProj = 1024
};
inline bool operator&(DavidsonTypes a, DavidsonTypes b) { return ((uint16_t)a & (uint16_t)b) != 0; }
inline DavidsonTypes operator|(DavidsonTypes a, DavidsonTypes b) { return DavidsonTypes((uint16_t)a | (uint16_t)b); }

// The PComm argument of the reference's davidson: rank / root and a broadcast of DEVICE vectors (b2x_broadcast, RCCL).
// (bcast_fn / sum_fn: a transport supplied by the caller instead of the library's RCCL communicator — ranks that share one
// card, which RCCL refuses, go through the host mirror's gloo transport in tests and rehearsals)
struct DeviceComm {
    b2x_comm *comm = nullptr;
    int rank = 0, size = 1, root = 0;
    std::function<void(double *, size_t)> bcast_fn, sum_fn;
    void broadcast(double *dev, size_t n) const {
        if (bcast_fn)
            bcast_fn(dev, n);
        else
            check(b2x_broadcast(comm, dev, n, root, nullptr));
    }
    void allreduce_sum(double *dev, size_t n) const {
        if (sum_fn)
            sum_fn(dev, n);
        else
            check(b2x_allreduce_sum(comm, dev, n, nullptr));
    }
};

struct IterativeMatrixFunctions {
    // Davidson for k eigenpairs, every psi-sized vector resident on the device.
    //   op(b_dev, sigma_dev): sigma += H b   (sigma arrives zeroed, as in the reference :972-973)
    //   aa_dev: diagonal of H (preconditioner), vs_dev: k initial guesses, overwritten with eigenvectors.
    // Same control flow as the reference (:864-1173): orthonormalise guesses (and project out ors), expand by one
    // preconditioned residual per iteration, full Rayleigh-Ritz rotation of (b, sigma) each iteration, roots ordered by
    // davidson_type / shift, converge on |r|^2 < conv_thrd + (ld rel_conv_thrd)^2, collapse to deflation_min_size
    // vectors at deflation_max_size.  ors: states projected out ((1 - |v><v|), or H + w |v><v| when proj_weights is
    // given, :888-893).  Sum-MPO (pcomm != nullptr): op() ends in the all-reduce of sigma, so every rank holds the same
    // (b, sigma) and runs the SAME subspace step (bitwise: RCCL's all-reduce returns identical sums on every rank, the
    // vec_* kernels reduce in a fixed order) instead of the reference's root-only step; the reference's broadcast of the
    // new basis vectors (:968-970) is kept, the scalar broadcasts (:1094-1097) are not needed.
    // Not built: Harmonic / NonHermitian / Exact / LeftEigen variants (other solvers of the reference).
    static std::vector<double> davidson(const std::function<void(const double *, double *)> &op, const double *aa_dev,
                                        std::vector<double *> &vs_dev, size_t n, double shift,
                                        DavidsonTypes davidson_type, int &ndav, bool iprint = false,
                                        const DeviceComm *pcomm = nullptr, double conv_thrd = 5E-6,
                                        double rel_conv_thrd = 0.0, int max_iter = 5000, int soft_max_iter = -1,
                                        int deflation_min_size = 2, int deflation_max_size = 50,
                                        const std::vector<double *> &ors = std::vector<double *>(),
                                        const std::vector<double> &proj_weights = std::vector<double>()) {
        if ((davidson_type & DavidsonTypes::Harmonic) || (davidson_type & DavidsonTypes::NonHermitian) ||
            (davidson_type & DavidsonTypes::Exact) || (davidson_type & DavidsonTypes::LeftEigen))
            throw std::runtime_error("davidson: Harmonic / NonHermitian / Exact / LeftEigen types are not built");
        const int k = (int)vs_dev.size();
        int nor = (int)ors.size(), nwg = 0;
        if (davidson_type & DavidsonTypes::ElementProj)
            ;
        else if (proj_weights.size() != 0) {
            if (proj_weights.size() != ors.size())
                throw std::runtime_error("davidson: proj_weights and ors differ in length");
            nwg = (int)ors.size(), nor = 0;
        }
        if (deflation_min_size < k)
            deflation_min_size = k;
        if (deflation_max_size < k + k / 2)
            deflation_max_size = k + k / 2;
        if (deflation_max_size > 63)
            deflation_max_size = 63;
        const int M = deflation_max_size;
        // psi-sized work vectors come from slabs of 8 (one hipMalloc per slab: a device allocation per new subspace vector
        // costs more than the vector algebra of an iteration at small |psi|)
        std::vector<std::unique_ptr<DeviceVector>> store;
        const size_t n_pad = (n + 31) & ~(size_t)31;
        size_t slab_left = 0;
        double *slab_next = nullptr;
        auto mk = [&]() {
            if (slab_left == 0) {
                store.emplace_back(new DeviceVector(8 * n_pad));
                slab_next = store.back()->p, slab_left = 8;
            }
            double *r = slab_next;
            slab_next += n_pad, slab_left--;
            return r;
        };
        // the subspace vectors are allocated as the subspace grows (the reference holds 2 M vectors up front, :902-907;
        // typical runs converge with m << M)
        std::vector<double *> bs(M, nullptr), sg(M, nullptr);
        auto ensure = [&](int i) {
            if (bs[i] == nullptr)
                bs[i] = mk(), sg[i] = mk();
        };
        for (int i = 0; i < k; i++)
            ensure(i);
        double *q = mk(), *t = mk(), *x = mk(); // residual / work vector / current Ritz vector
        double *q2 = mk();                      // preconditioned residual of the fused step (below)
        std::vector<double *> dfl_b, dfl_s;     // work vectors of a deflation step
        auto dot = [&](const double *u, const double *v) {
            double r;
            check(b2x_vec_dot(u, v, n, &r, nullptr));
            return r;
        };
        std::vector<double> or_normsqs(nor);
        for (int i = 0; i < nor; i++) { // :912-918 (the ors are orthogonalised in place, as the reference does)
            for (int j = 0; j < i; j++)
                if (std::fabs(or_normsqs[j]) > 1E-14)
                    check(b2x_vec_axpy(-dot(ors[j], ors[i]) / or_normsqs[j], ors[j], ors[i], n, nullptr));
            or_normsqs[i] = dot(ors[i], ors[i]);
        }
        auto project_ors = [&](double *v) {
            for (int j = 0; j < nor; j++)
                if (std::fabs(or_normsqs[j]) > 1E-14)
                    check(b2x_vec_axpy(-dot(ors[j], v) / or_normsqs[j], ors[j], v, n, nullptr));
        };
        // v -= sum_{j < m} <b_j, v> b_j: ONE multi-dot and ONE linear combination per pass (the reference's loop of m
        // dot / axpy pairs, :1143-1144, costs m launches and m host round trips on a device), two passes for the
        // orthogonality a one-pass classical Gram-Schmidt loses.  Result in `v` (the work vector w is swapped in).
        std::vector<double> gs_c(M + 2);
        auto orthogonalise = [&](double *&v, double *&w, int m_) {
            for (int pass = 0; pass < 2 && m_ > 0; pass++) {
                std::vector<const double *> ptrs(bs.begin(), bs.begin() + m_);
                check(b2x_vec_multi_dot(ptrs.data(), m_, v, n, gs_c.data(), nullptr));
                for (int j = 0; j < m_; j++)
                    gs_c[j] = -gs_c[j];
                gs_c[m_] = 1.0;
                ptrs.push_back(v);
                check(b2x_vec_lincomb(ptrs.data(), m_ + 1, gs_c.data(), w, n, nullptr));
                std::swap(v, w);
            }
        };
        for (int i = 0; i < k; i++)
            check(b2x_vec_copy(vs_dev[i], bs[i], n, nullptr));
        int m = k;
        for (int i = 0; i < k; i++) {
            for (int j = 0; j < i; j++)
                check(b2x_vec_axpy(-dot(bs[j], bs[i]), bs[j], bs[i], n, nullptr));
            double nrm = std::sqrt(dot(bs[i], bs[i]));
            if (nrm * nrm < 1E-14 && i > 0) {
                m = i;
                break;
            }
            if (nrm * nrm < 1E-14)
                throw std::runtime_error("Cannot generate initial guess 0 for Davidson (zero norm)!");
            check(b2x_vec_scal(1.0 / nrm, bs[i], n, nullptr));
        }
        for (int i = 0; i < m && nor != 0; i++) {
            project_ors(bs[i]);
            double nrm = std::sqrt(dot(bs[i], bs[i]));
            if (nrm * nrm < 1E-14)
                throw std::runtime_error("Cannot generate initial guess " + std::to_string(i) +
                                         " for Davidson unitary to all given states!");
            check(b2x_vec_scal(1.0 / nrm, bs[i], n, nullptr));
        }
        // The reference rotates ALL basis vectors and their sigmas into the Ritz basis in every iteration (:1000-1022:
        // 2 m^2 axpy-like passes over psi-sized vectors).  The Ritz pairs depend on the subspace only, so the basis stays
        // as it is here and the projected matrix H_ij = <b_i, sigma_j> is kept on the host (one new column per new
        // vector); the Ritz vector and the residual of the root in work are two linear combinations.  Same subspace, same
        // Ritz values and residuals up to rounding; per iteration O(m) instead of O(m^2) vector passes and a fixed
        // number of launches (tools/davidson_overhead.py: 2.5 -> ms of overhead per iteration at |psi| = 36 k).
        std::vector<double> H((size_t)M * M, 0.0), alpha, ld, row(M), coef(M + 1);
        std::vector<double> eigvals(k);
        std::vector<int> idx(M);
        int ck = 0, msig = 0, xiter = 0;
        double qq = 0;
        auto thrd = [&](double e) { return conv_thrd + e * e * rel_conv_thrd * rel_conv_thrd; };
        // x = Ritz vector of root r (coefficients alpha row r), q = H x - theta x
        auto ritz_residual = [&](int r) {
            std::vector<const double *> pb(bs.begin(), bs.begin() + m), ps(sg.begin(), sg.begin() + m);
            check(b2x_vec_lincomb(pb.data(), m, &alpha[(size_t)r * m], x, n, nullptr));
            for (int j = 0; j < m; j++)
                coef[j] = alpha[(size_t)r * m + j];
            coef[m] = -ld[r];
            ps.push_back(x);
            check(b2x_vec_lincomb(ps.data(), m + 1, coef.data(), q, n, nullptr));
        };
        static const bool prof = getenv("B2X_DAV_PROFILE") != nullptr;
        const char *fused_env = getenv("B2X_DAV_FUSED");
        const bool fused_ok = nor == 0 && nwg == 0 && !(davidson_type & DavidsonTypes::DavidsonPrecond) &&
                              !(davidson_type & DavidsonTypes::NoPrecond) && !(fused_env && fused_env[0] == '0');
        std::vector<const double *> fu, fv; // pairs of the fused dot products, and their results
        std::vector<double> fr;
        int gs_pending = -1; // basis vector whose device-side normalisation has not been confirmed yet
        if (fused_ok) {      // (a flag left by an earlier solve that ended before looking at it)
            int stale = 0;
            check(b2x_vec_gs_status(&stale, 1));
        }
        double t_op = 0, t_eig = 0;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double>(b - a).count();
        };
        const auto t_begin = now();
        while (xiter < max_iter && (soft_max_iter == -1 || xiter < soft_max_iter)) {
            xiter++;
            if (pcomm != nullptr && xiter != 1)
                for (int i = msig; i < m; i++)
                    pcomm->broadcast(bs[i], n);
            for (int i = msig; i < m; i++, msig++) {
                check(b2x_vec_zero(sg[i], n, nullptr));
                const auto t0 = now();
                op(bs[i], sg[i]);
                if (prof) {
                    check(b2x_device_sync());
                    t_op += secs(t0, now());
                }
                for (int j = 0; j < nwg; j++)
                    check(b2x_vec_axpy(dot(ors[j], bs[i]) * proj_weights[j], ors[j], sg[i], n, nullptr));
                // new column of the projected matrix: H(j, i) = <b_j, sigma_i>, j <= i (H is symmetric)
                std::vector<const double *> ptrs(bs.begin(), bs.begin() + i + 1);
                check(b2x_vec_multi_dot(ptrs.data(), i + 1, sg[i], n, row.data(), nullptr));
                if (gs_pending == i) { // (b_i came from the device-side Gram-Schmidt finish: its status is known now)
                    gs_pending = -1;
                    int degenerate = 0;
                    check(b2x_vec_gs_status(&degenerate, 1));
                    if (degenerate) { // the careful way: two more passes against b_0..b_{i-1}, explicit norm, sigma again
                        double *v = bs[i], *w = t;
                        orthogonalise(v, w, i);
                        if (v != bs[i])
                            check(b2x_vec_copy(v, bs[i], n, nullptr));
                        check(b2x_vec_scal(1.0 / std::sqrt(dot(bs[i], bs[i])), bs[i], n, nullptr));
                        if (pcomm != nullptr)
                            pcomm->broadcast(bs[i], n);
                        check(b2x_vec_zero(sg[i], n, nullptr));
                        op(bs[i], sg[i]);
                        check(b2x_vec_multi_dot(ptrs.data(), i + 1, sg[i], n, row.data(), nullptr));
                    }
                }
                for (int j = 0; j <= i; j++)
                    H[(size_t)j * M + i] = H[(size_t)i * M + j] = row[j];
            }
            // Rayleigh-Ritz in the current basis
            alpha.assign((size_t)m * m, 0.0);
            for (int i = 0; i < m; i++)
                for (int j = 0; j <= i; j++)
                    alpha[(size_t)i * m + j] = H[(size_t)i * M + j];
            {
                const auto t0 = now();
                small_eigs(alpha, ld, m); // row r of alpha = eigenvector r (ascending eigenvalues)
                t_eig += secs(t0, now());
            }
            for (int i = 0; i < m; i++)
                idx[i] = i;
            if (davidson_type & DavidsonTypes::CloseTo) // the root order of :1024-1049
                std::sort(idx.begin(), idx.begin() + m,
                          [&](int i, int j) { return std::fabs(ld[i] - shift) < std::fabs(ld[j] - shift); });
            else if (davidson_type & DavidsonTypes::LessThan)
                std::sort(idx.begin(), idx.begin() + m, [&](int i, int j) {
                    if ((shift >= ld[i]) != (shift >= ld[j]))
                        return shift >= ld[i];
                    return shift >= ld[i] ? shift - ld[i] < shift - ld[j] : ld[i] - shift > ld[j] - shift;
                });
            else if (davidson_type & DavidsonTypes::GreaterThan)
                std::sort(idx.begin(), idx.begin() + m, [&](int i, int j) {
                    if ((shift > ld[i]) != (shift > ld[j]))
                        return shift > ld[j];
                    return shift > ld[i] ? shift - ld[i] > shift - ld[j] : ld[i] - shift < ld[j] - shift;
                });
            for (int i = 0; i < ck; i++) { // re-check the roots already counted as converged
                ritz_residual(idx[i]);
                if (std::fabs(dot(q, q)) >= thrd(ld[idx[i]])) {
                    ck = i;
                    break;
                }
            }
            const int ick = idx[ck];
            // The usual step (Olsen preconditioner, no projected-out states, no collapse due) in TWO host round trips instead of
            // six: the residual norm, the two Olsen products and the projections of the preconditioned residual q2 and of t on
            // the basis are ONE b2x_vec_pair_dots; the Olsen correction and the first Gram-Schmidt pass are then ONE linear
            // combination (both are linear in q2 and t: <b_j, q2 - g t> = <b_j, q2> - g <b_j, t>); the second pass and the
            // normalisation stay on the device (b2x_vec_gs_finish).  Same subspace as the step-by-step form below up to rounding
            // (B2X_DAV_FUSED=0 selects that form; tools/davidson_overhead.py).
            const bool fused = fused_ok && m < deflation_max_size && m + 2 <= 64 && 2 * m + 3 <= 128;
            if (fused) {
                std::vector<const double *> pb(bs.begin(), bs.begin() + m), ps(sg.begin(), sg.begin() + m);
                check(b2x_vec_ritz_olsen(pb.data(), ps.data(), m, &alpha[(size_t)ick * m], ld[ick], aa_dev, x, q, q2, t, n, nullptr));
                fu.assign({q, q2, t}), fv.assign({q, x, x});
                for (int j = 0; j < m; j++)
                    fu.push_back(bs[j]), fv.push_back(q2);
                for (int j = 0; j < m; j++)
                    fu.push_back(bs[j]), fv.push_back(t);
                fr.resize(3 + 2 * (size_t)m);
                check(b2x_vec_pair_dots(fu.data(), fv.data(), 3 + 2 * m, n, fr.data(), nullptr));
                qq = fr[0];
            } else {
                ritz_residual(ick);
                project_ors(q);
                qq = dot(q, q);
            }
            if (iprint)
                printf("%6d%6d%6d%15.8f%13.2e\n", xiter, m, ck, ld[ick], std::fabs(qq));
            if (fused) {
            } else if (davidson_type & DavidsonTypes::DavidsonPrecond) // davidson_precondition (:66-72)
                check(b2x_vec_precondition(q, aa_dev, ld[ick], n, nullptr));
            else if (!(davidson_type & DavidsonTypes::NoPrecond)) { // olsen_precondition (:93-108)
                check(b2x_vec_olsen_prepare(q, t, x, aa_dev, ld[ick], n, nullptr));
                const double *qt[2] = {q, t};
                double cqt[2];
                check(b2x_vec_multi_dot(qt, 2, x, n, cqt, nullptr));
                check(b2x_vec_axpy(-cqt[0] / cqt[1], t, q, n, nullptr));
            }
            eigvals.resize(ck + 1);
            for (int i = 0; i <= ck; i++)
                eigvals[i] = ld[idx[i]];
            if (std::fabs(qq) < thrd(eigvals[ck]) && m >= k) {
                ck++;
                if (ck == k)
                    break;
            } else {
                if (m >= deflation_max_size) { // collapse to the leading Ritz vectors (:1108-1141)
                    const int keep = deflation_min_size;
                    std::vector<const double *> pb(bs.begin(), bs.begin() + m), ps(sg.begin(), sg.begin() + m);
                    if ((int)dfl_b.size() < keep)
                        for (int j = (int)dfl_b.size(); j < keep; j++)
                            dfl_b.push_back(mk()), dfl_s.push_back(mk());
                    for (int j = 0; j < keep; j++) {
                        check(b2x_vec_lincomb(pb.data(), m, &alpha[(size_t)idx[j] * m], dfl_b[j], n, nullptr));
                        check(b2x_vec_lincomb(ps.data(), m, &alpha[(size_t)idx[j] * m], dfl_s[j], n, nullptr));
                    }
                    std::vector<double> th(keep);
                    for (int j = 0; j < keep; j++) {
                        th[j] = ld[idx[j]];
                        check(b2x_vec_copy(dfl_b[j], bs[j], n, nullptr));
                        check(b2x_vec_copy(dfl_s[j], sg[j], n, nullptr));
                    }
                    // in the new basis the projected matrix is diag(theta) and the Ritz vectors are the basis vectors
                    std::fill(H.begin(), H.end(), 0.0);
                    alpha.assign((size_t)keep * keep, 0.0);
                    ld.assign(th.begin(), th.end());
                    for (int j = 0; j < keep; j++)
                        H[(size_t)j * M + j] = th[j], alpha[(size_t)j * keep + j] = 1.0, idx[j] = j;
                    m = msig = keep;
                }
                bool appended = false;
                if (fused) {
                    // v1 = q2 - g t - sum_j (<b_j, q2> - g <b_j, t>) b_j   (the residual q is free: v1 goes there)
                    const double g = fr[1] / fr[2];
                    std::vector<const double *> ptrs(bs.begin(), bs.begin() + m);
                    for (int j = 0; j < m; j++)
                        gs_c[j] = -(fr[3 + j] - g * fr[3 + m + j]);
                    ptrs.push_back(q2), ptrs.push_back(t);
                    gs_c[m] = 1.0, gs_c[m + 1] = -g;
                    check(b2x_vec_lincomb(ptrs.data(), m + 2, gs_c.data(), q, n, nullptr));
                    // second pass and normalisation WITHOUT a host round trip: <b_j, v1> and <v1, v1> stay on the device and the
                    // kernel that forms b_m = (v1 - sum c_j b_j) / |.| reads them there (|.|^2 = <v1, v1> - sum c_j^2 for an
                    // orthonormal basis; the c_j are rounding-sized after the first pass: no cancellation).  Should the new
                    // direction lie in the subspace to rounding, the kernel raises a flag that is looked at after the next wait
                    // (the projected-matrix column of b_m, below) and b_m is redone the careful way there.
                    ptrs.assign(bs.begin(), bs.begin() + m);
                    ensure(m);
                    check(b2x_vec_gs_finish(ptrs.data(), m, q, bs[m], n, nullptr));
                    gs_pending = m;
                    m++, appended = true;
                }
                if (!appended) {
                    orthogonalise(q, t, m);
                    project_ors(q);
                    check(b2x_vec_scal(1.0 / std::sqrt(dot(q, q)), q, n, nullptr));
                    ensure(m);
                    check(b2x_vec_copy(q, bs[m], n, nullptr));
                    m++;
                }
            }
            if (xiter == soft_max_iter)
                break;
        }
        if (xiter == soft_max_iter)
            eigvals.resize(k, 0);
        if (xiter == max_iter)
            throw std::runtime_error("Davidson: only " + std::to_string(ck) + " converged!");
        if (!alpha.empty() && (int)ld.size() >= k) { // eigenvectors = Ritz vectors of the last Rayleigh-Ritz step
            const int mr = (int)ld.size();
            std::vector<const double *> pb(bs.begin(), bs.begin() + mr);
            std::vector<double *> outv(k);
            for (int i = 0; i < k; i++) {
                outv[i] = mk();
                check(b2x_vec_lincomb(pb.data(), mr, &alpha[(size_t)idx[i] * mr], outv[i], n, nullptr));
            }
            for (int i = 0; i < k; i++)
                check(b2x_vec_copy(outv[i], vs_dev[i], n, nullptr));
        }
        if (pcomm != nullptr)
            for (int i = 0; i < k; i++)
                pcomm->broadcast(vs_dev[i], n);
        check(b2x_device_sync());
        if (prof)
            fprintf(stderr, "davidson: %d iterations %.3f ms each: H.psi (synchronised) %.3f, host eigensolver %.3f, vector algebra %.3f\n",
                    xiter, secs(t_begin, now()) / xiter * 1e3, t_op / xiter * 1e3, t_eig / xiter * 1e3,
                    (secs(t_begin, now()) - t_op - t_eig) / xiter * 1e3);
        ndav = xiter;
        return eigvals;
    }
    // the entry point EffectiveHamiltonian::eigs calls (:1181-1194): Harmonic types have their own solver in the
    // reference, every other type is davidson()
    static std::vector<double> harmonic_davidson(const std::function<void(const double *, double *)> &op,
                                                 const double *aa_dev, std::vector<double *> &vs_dev, size_t n,
                                                 double shift, DavidsonTypes davidson_type, int &ndav,
                                                 bool iprint = false, const DeviceComm *pcomm = nullptr,
                                                 double conv_thrd = 5E-6, double rel_conv_thrd = 0.0,
                                                 int max_iter = 5000, int soft_max_iter = -1,
                                                 int deflation_min_size = 2, int deflation_max_size = 50,
                                                 const std::vector<double *> &ors = std::vector<double *>(),
                                                 const std::vector<double> &proj_weights = std::vector<double>()) {
        return davidson(op, aa_dev, vs_dev, n, shift, davidson_type, ndav, iprint, pcomm, conv_thrd, rel_conv_thrd,
                        max_iter, soft_max_iter, deflation_min_size, deflation_max_size, ors, proj_weights);
    }
    // the round-1 call form (Normal type, no shift, no projection)
    static std::vector<double> davidson(const std::function<void(const double *, double *)> &op, const double *aa_dev,
                                        std::vector<double *> &vs_dev, size_t n, int &ndav, double conv_thrd = 5E-6,
                                        int max_iter = 5000, int soft_max_iter = -1, int deflation_min_size = 2,
                                        int deflation_max_size = 50, bool iprint = false) {
        return davidson(op, aa_dev, vs_dev, n, 0.0, DavidsonTypes::Normal, ndav, iprint, nullptr, conv_thrd, 0.0,
                        max_iter, soft_max_iter, deflation_min_size, deflation_max_size);
    }
};

// The local problem of one site at the level this path sees it: a recorded plan (what precompute() builds),
// the diagonal and the wavefunction.  eigs() == EffectiveHamiltonian::eigs (effective_hamiltonian.hpp:470-558):
// returns (energy, ndav, nflop, tdav); ket is overwritten with the eigenvector.
// EffectiveKernel<FL> (effective_hamiltonian.hpp:81-91): a user hook AROUND the matrix-vector product f of eigs —
// compute(beta, f, a, b, xs) must leave b += beta * H a (it may call f any number of times).  Here a, b (and xs) are
// DEVICE vectors of |psi| doubles: Davidson never brings them to the host.
struct EffectiveKernel {
    typedef std::function<void(const double *, double *, double)> MatMul;
    virtual ~EffectiveKernel() = default;
    virtual void compute(double beta, const MatMul &f, const double *a, double *b, const std::vector<const double *> &xs) const {
        (void)xs;
        f(a, b, beta);
    }
};

struct EffectiveHamiltonian {
    std::shared_ptr<BatchGEMMSeq> seq;
    std::vector<double> diag;
    size_t n;
    std::shared_ptr<EffectiveKernel> eff_kernel; // (effective_hamiltonian.hpp:121, used at :515-519)
    EffectiveHamiltonian(const std::shared_ptr<BatchGEMMSeq> &seq, const std::vector<double> &diag)
        : seq(seq), diag(diag), n(diag.size()) {}
    void precompute() { seq->prepare(n, n); }
    void post_precompute() { seq->deallocate(); }
    // [c] += factor * [H_eff] x [b]    (host vectors)
    void operator()(const GMatrix &b, const GMatrix &c, double factor = 1.0) {
        precompute();
        (*seq)(b, c, factor);
    }
    // eigs (effective_hamiltonian.hpp:470-558), arguments in the reference's order after ket (the reference takes ket
    // from the object): ortho_bra = states projected out, projection_weights as in davidson; pcomm = para_rule->comm.
    std::tuple<double, int, size_t, double> eigs(std::vector<double> &ket, bool iprint = false, double conv_thrd = 5E-6,
                                                 double rel_conv_thrd = 0.0, int max_iter = 5000, int soft_max_iter = -1,
                                                 int deflation_min_size = 2, int deflation_max_size = 50,
                                                 DavidsonTypes davidson_type = DavidsonTypes::Normal, double shift = 0,
                                                 const DeviceComm *pcomm = nullptr,
                                                 const std::vector<std::vector<double>> &ortho_bra =
                                                     std::vector<std::vector<double>>(),
                                                 const std::vector<double> &projection_weights = std::vector<double>()) {
        if (ket.size() != n)
            throw std::runtime_error("EffectiveHamiltonian::eigs: ket length differs from diag");
        precompute();
        seq->cumulative_nflop = 0;
        DeviceVector dk(n), dd(n);
        dk.upload(ket.data()), dd.upload(diag.data());
        std::vector<std::unique_ptr<DeviceVector>> dors;
        std::vector<double *> ors;
        for (auto &o : ortho_bra) {
            if (o.size() != n)
                throw std::runtime_error("EffectiveHamiltonian::eigs: ortho_bra length differs from diag");
            dors.emplace_back(new DeviceVector(n));
            dors.back()->upload(o.data());
            ors.push_back(dors.back()->p);
        }
        std::vector<double *> vs{dk.p};
        int ndav = 0;
        auto t0 = std::chrono::steady_clock::now();
        EffectiveKernel::MatMul f3 = [this](const double *a, double *b, double scale) { seq->apply_device(a, b, scale); };
        auto f = [this, &f3](const double *b, double *s) {
            if (eff_kernel == nullptr)
                f3(b, s, 1.0);
            else
                eff_kernel->compute(1.0, f3, b, s, std::vector<const double *>());
        };
        std::vector<double> eners = IterativeMatrixFunctions::harmonic_davidson(
            f, dd.p, vs, n, shift, davidson_type, ndav, iprint, pcomm, conv_thrd, rel_conv_thrd, max_iter, soft_max_iter,
            deflation_min_size, deflation_max_size, ors, projection_weights);
        double tdav = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        dk.download(ket.data());
        size_t nf = seq->cumulative_nflop;
        seq->cumulative_nflop = 0;
        return std::make_tuple(eners[0], ndav, nf, tdav);
    }
};

} // namespace b2xh
