// b2x_symbolic_py.hpp — pybind glue for the symbolic -> numeric host layer (b2x_symbolic.hpp).
// A SymbolicEffectiveHamiltonian is assembled from the named arrays of an effective-Hamiltonian fixture
// (oracle/ref_dump.cpp `eham=`; read with planfile.read_arrays): operator infos, operator tensors, the term list
// of H_eff, wavefunction infos and the operator data arena.
#pragma once
#include "b2x_symbolic.hpp"
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <chrono>
#include <sstream>

namespace b2xh {
namespace py = pybind11;

struct SymEHBase {
    virtual ~SymEHBase() = default;
    virtual py::dict wfn_cinfo() const = 0;
    virtual void precompute() = 0;
    virtual void post_precompute() = 0;
    virtual size_t n_pairs() const = 0;
    virtual size_t nflop() const = 0;
    virtual py::array pairs() const = 0;
    virtual void record() = 0;
    virtual py::array_t<double> compute_diag() = 0;
    virtual py::array diag_terms() = 0;
    virtual void apply(py::array_t<double, py::array::c_style> b, py::array_t<double, py::array::c_style> c,
                       double factor) = 0;
    virtual py::tuple eigs(std::vector<double> ket, double conv_thrd, int max_iter) = 0;
    virtual py::tuple perturbative_noise(const py::dict &d, bool execute) = 0;
};

template <typename S> struct SymEH : SymEHBase {
    typedef SparseMatrixInfo<S> Info;
    std::shared_ptr<SymbolicEffectiveHamiltonian<S>> h;
    py::array_t<double> arena; // keeps the operator data alive
    template <typename T> static const T *arr(const py::dict &d, const std::string &k, size_t *n = nullptr) {
        if (!d.contains(k.c_str()))
            throw std::runtime_error("fixture lacks array '" + k + "'");
        py::array a = py::array::ensure(d[k.c_str()]);
        if (a.itemsize() != (py::ssize_t)sizeof(T))
            throw std::runtime_error("array '" + k + "' has the wrong element size");
        if (n)
            *n = (size_t)a.size();
        return (const T *)a.data();
    }
    static std::shared_ptr<Info> info(const py::dict &d, int id, std::map<int, std::shared_ptr<Info>> &cache) {
        auto it = cache.find(id);
        if (it != cache.end())
            return it->second;
        const std::string pre = "info." + std::to_string(id);
        size_t n;
        const uint64_t *q = arr<uint64_t>(d, pre + ".quanta", &n);
        const uint32_t *nb = arr<uint32_t>(d, pre + ".nbra"), *nk = arr<uint32_t>(d, pre + ".nket"),
                       *nt = arr<uint32_t>(d, pre + ".ntot");
        const uint64_t *meta = arr<uint64_t>(d, pre + ".meta");
        auto r = std::make_shared<Info>();
        r->n = (int)n;
        for (size_t i = 0; i < n; i++) {
            r->quanta.push_back(S(q[i]));
            r->n_states_bra.push_back(nb[i]), r->n_states_ket.push_back(nk[i]), r->n_states_total.push_back(nt[i]);
        }
        r->delta_quantum = S(meta[0]), r->is_fermion = meta[1] != 0, r->is_wavefunction = meta[2] != 0;
        // tensor-product connection info of a delayed enlarged operator (built by the blocking layer, taken as data)
        if (d.contains((pre + ".cinfo.n").c_str())) {
            auto ci = std::make_shared<typename Info::ConnectionInfo>();
            const int64_t *nn = arr<int64_t>(d, pre + ".cinfo.n");
            for (int i = 0; i < 5; i++)
                ci->n[i] = (int)nn[i];
            ci->nc = (int)nn[5];
            size_t nq, nc;
            const uint64_t *cq = arr<uint64_t>(d, pre + ".cinfo.quanta", &nq);
            const uint32_t *ix = arr<uint32_t>(d, pre + ".cinfo.idx");
            const uint64_t *st = arr<uint64_t>(d, pre + ".cinfo.stride", &nc);
            const double *f = arr<double>(d, pre + ".cinfo.factor");
            const uint32_t *ia = arr<uint32_t>(d, pre + ".cinfo.ia"), *ib = arr<uint32_t>(d, pre + ".cinfo.ib"),
                           *ic = arr<uint32_t>(d, pre + ".cinfo.ic");
            for (size_t i = 0; i < nq; i++)
                ci->quanta.push_back(S(cq[i])), ci->idx.push_back(ix[i]);
            for (size_t i = 0; i < nc; i++) {
                ci->stride.push_back(st[i]), ci->factor.push_back(f[i]);
                ci->ia.push_back(ia[i]), ci->ib.push_back(ib[i]), ci->ic.push_back(ic[i]);
            }
            r->cinfo = ci;
        }
        cache[id] = r;
        return r;
    }
    std::shared_ptr<OperatorTensor<S>> tensor(const py::dict &d, const std::string &pre,
                                              std::map<int, std::shared_ptr<Info>> &cache) {
        auto t = std::make_shared<OperatorTensor<S>>();
        size_t n;
        const int64_t *iid = arr<int64_t>(d, pre + ".info", &n), *len = arr<int64_t>(d, pre + ".len"),
                      *off = arr<int64_t>(d, pre + ".off");
        const double *fac = arr<double>(d, pre + ".factor");
        for (size_t i = 0; i < n; i++) {
            auto m = std::make_shared<SparseMatrix<S>>();
            m->info = info(d, (int)iid[i], cache);
            m->factor = fac[i];
            m->total_memory = len[i] < 0 ? 0 : (size_t)len[i];
            m->data = off[i] < 0 ? nullptr : arena.mutable_data() + off[i];
            t->ops.push_back(m);
        }
        return t;
    }
    explicit SymEH(const py::dict &d) {
        arena = py::array_t<double>(py::array::ensure(d["arena"]));
        std::map<int, std::shared_ptr<Info>> cache;
        const uint64_t *lab = arr<uint64_t>(d, "labels");
        std::vector<std::pair<S, std::shared_ptr<Info>>> li, ri;
        size_t nl, nr;
        const uint64_t *ll = arr<uint64_t>(d, "linfos.label", &nl), *lid = arr<uint64_t>(d, "linfos.info");
        const uint64_t *rl = arr<uint64_t>(d, "rinfos.label", &nr), *rid = arr<uint64_t>(d, "rinfos.info");
        for (size_t i = 0; i < nl; i++)
            li.emplace_back(S(ll[i]), info(d, (int)lid[i], cache));
        for (size_t i = 0; i < nr; i++)
            ri.emplace_back(S(rl[i]), info(d, (int)rid[i], cache));
        auto ket = info(d, (int)arr<uint64_t>(d, "ket.info")[0], cache);
        auto bra = info(d, (int)arr<uint64_t>(d, "bra.info")[0], cache);
        if (ket == bra) { // the reference shares one info object; the wavefunction connection info hangs off ket
            bra = std::make_shared<Info>(*ket);
        }
        auto lopt = tensor(d, "lopt", cache), ropt = tensor(d, "ropt", cache);
        const uint64_t *del = arr<uint64_t>(d, "tensor.delayed");
        if (del[0] || del[1]) {
            auto &dt = del[0] ? lopt : ropt;
            dt->type = OperatorTensorTypes::Delayed;
            dt->lopt = tensor(d, "dopt.l", cache), dt->ropt = tensor(d, "dopt.r", cache);
        }
        size_t nt;
        const int64_t *ty = arr<int64_t>(d, "expr.type", &nt), *cj = arr<int64_t>(d, "expr.conj"),
                      *ea = arr<int64_t>(d, "expr.a"), *eb = arr<int64_t>(d, "expr.b"), *d0 = arr<int64_t>(d, "expr.d0"),
                      *d1 = arr<int64_t>(d, "expr.d1"), *dc = arr<int64_t>(d, "expr.dconj");
        const double *ef = arr<double>(d, "expr.factor");
        std::vector<OpTerm> expr(nt);
        for (size_t i = 0; i < nt; i++) {
            expr[i].type = ty[i] ? OpTypes::SumProd : OpTypes::Prod;
            expr[i].factor = ef[i], expr[i].conj = (uint8_t)cj[i];
            expr[i].a = (int)ea[i], expr[i].b = (int)eb[i], expr[i].d0 = (int)d0[i], expr[i].d1 = (int)d1[i];
            expr[i].dconj = (uint8_t)dc[i];
        }
        // operator sub-labels (conj flag, combined delta quantum): from the reference's get_uniq_sub_labels
        // (partition.hpp, MPO layer) — taken as data, stored with the fixture as the keys of its connection info
        const int64_t *nn = arr<int64_t>(d, "wfn_cinfo.n");
        const uint64_t *sq = arr<uint64_t>(d, "wfn_cinfo.quanta");
        std::vector<std::pair<uint8_t, S>> subdq;
        for (int k = 0; k < (int)nn[4]; k++) {
            uint8_t c = 0;
            while (c < 3 && k >= (int)nn[c + 1])
                c++;
            subdq.emplace_back(c, S(sq[k]));
        }
        std::vector<double> diag; // (a noise fixture carries no diagonal: only operator(), not eigs, is used on it)
        if (d.contains("diag")) {
            size_t nd;
            const double *dg = arr<double>(d, "diag", &nd);
            diag.assign(dg, dg + nd);
        }
        h = std::make_shared<SymbolicEffectiveHamiltonian<S>>(li, ri, lopt, ropt, expr, ket, bra, S(lab[2]), subdq, diag);
    }
    py::dict wfn_cinfo() const override {
        const auto &c = *h->wfn_info;
        py::dict r;
        std::vector<uint64_t> q;
        for (auto x : c.quanta)
            q.push_back(x.data);
        r["n"] = std::vector<int64_t>{c.n[0], c.n[1], c.n[2], c.n[3], c.n[4], c.nc};
        r["quanta"] = q, r["idx"] = c.idx, r["stride"] = c.stride, r["factor"] = c.factor;
        r["ia"] = c.ia, r["ib"] = c.ib, r["ic"] = c.ic;
        return r;
    }
    py::array_t<double> compute_diag() override {
        std::vector<double> d = h->compute_diag(h->subdq);
        return py::array_t<double>(d.size(), d.data());
    }
    // record the diagonal terms only (initialize_diag + tensor_product_diagonal), offsets relative to the arena
    py::array diag_terms() override {
        typedef SparseMatrixInfo<S> I;
        auto dinfo = std::make_shared<I>(*h->ket_info);
        dinfo->cinfo = std::make_shared<typename I::ConnectionInfo>();
        dinfo->cinfo->initialize_diag(h->ket_info->delta_quantum, h->opdq, h->subdq, h->left_op_infos,
                                      h->right_op_infos, dinfo, h->tf->opf->cg);
        SparseMatrix<S> dmat;
        dmat.info = dinfo, dmat.data = (double *)0, dmat.factor = 1.0;
        auto &sq = *h->tf->opf->seq;
        sq.diag_terms.clear(), sq.da_ptr.clear(), sq.db_ptr.clear();
        h->tf->tensor_product_diagonal(h->expr, *h->lopt, *h->ropt, dmat, h->opdq);
        std::vector<b2x_diag_term> t = sq.diag_terms;
        for (size_t i = 0; i < t.size(); i++) {
            t[i].a_off = (uint64_t)(sq.da_ptr[i] - arena.data());
            t[i].b_off = (uint64_t)(sq.db_ptr[i] - arena.data());
        }
        sq.diag_terms.clear(), sq.da_ptr.clear(), sq.db_ptr.clear();
        py::array_t<uint8_t> a(t.size() * sizeof(b2x_diag_term));
        std::memcpy(a.mutable_data(), t.data(), t.size() * sizeof(b2x_diag_term));
        return a;
    }
    void precompute() override { h->precompute(); }
    void post_precompute() override { h->post_precompute(); }
    size_t n_pairs() const override { return h->tf->opf->seq->pairs.size(); }
    size_t nflop() const override { return h->tf->opf->seq->nflop; }
    // the recorded pairs with operator pointers expressed as offsets into the fixture's arena (no device needed)
    py::array pairs() const override {
        const auto &sq = *h->tf->opf->seq;
        std::vector<b2x_pair> p = sq.pairs;
        for (size_t i = 0; i < p.size(); i++) {
            p[i].y_off = (uint64_t)(sq.y_ptr[i] - arena.data());
            p[i].z_off = (uint64_t)(sq.z_ptr[i] - arena.data());
        }
        py::array_t<uint8_t> a(p.size() * sizeof(b2x_pair));
        std::memcpy(a.mutable_data(), p.data(), p.size() * sizeof(b2x_pair));
        return a;
    }
    // record the plan only (tensor_product_multiply with null-based wavefunctions), no upload
    void record() override {
        auto seq = h->tf->opf->seq;
        if (!seq->pairs.empty())
            return;
        SparseMatrix<S> cmat, vmat;
        cmat.info = h->ket_info, vmat.info = h->bra_info;
        cmat.data = vmat.data = (double *)0;
        py::gil_scoped_release nogil; // (the walk touches no Python object: other threads of the sweep loop may run)
        h->tf->tensor_product_multiply(h->expr, *h->lopt, *h->ropt, cmat, vmat, h->opdq);
    }
    void apply(py::array_t<double, py::array::c_style> b, py::array_t<double, py::array::c_style> c,
               double factor) override {
        (*h)(GMatrix(b.mutable_data(), (int)b.size(), 1), GMatrix(c.mutable_data(), (int)c.size(), 1), factor);
    }
    // perturbative noise from the `noise.*` arrays of an .enoise fixture: returns (b2x_gemm records with operands as
    // offsets into (arena, psi) and outputs as offsets into the perturbed-wavefunction vector, that vector); with
    // execute the list also runs on the device (BatchGEMMSeq::auto_perform)
    py::tuple perturbative_noise(const py::dict &d, bool execute) override {
        std::map<int, std::shared_ptr<Info>> cache;
        const int64_t *args = arr<int64_t>(d, "noise.args"), *iop = arr<int64_t>(d, "noise.iop");
        const bool trace_right = args[0] != 0;
        size_t np, nv;
        const int64_t *pc = arr<int64_t>(d, "noise.psubsl.conj", &np);
        const uint64_t *pl = arr<uint64_t>(d, "noise.psubsl.label");
        std::vector<std::pair<uint8_t, S>> psubsl;
        for (size_t i = 0; i < np; i++)
            psubsl.emplace_back((uint8_t)pc[i], S(pl[i]));
        const uint64_t *vd = arr<uint64_t>(d, "noise.vdqs", &nv), *vi = arr<uint64_t>(d, "noise.vinfo"),
                       *vo = arr<uint64_t>(d, "noise.voff");
        std::vector<S> vdqs;
        std::vector<std::shared_ptr<Info>> vinfos;
        std::vector<uint64_t> voffs;
        for (size_t i = 0; i < nv; i++)
            vdqs.push_back(S(vd[i])), vinfos.push_back(info(d, (int)vi[i], cache)), voffs.push_back(vo[i]);
        typename TensorFunctions<S>::IdentityOps id;
        id.l = (int)iop[0], id.r = (int)iop[1], id.dl = (int)iop[2], id.dr = (int)iop[3];
        py::array_t<double> psi = py::array_t<double>(py::array::ensure(d["psi"]));
        py::array_t<double> v((py::ssize_t)args[5]);
        std::fill(v.mutable_data(), v.mutable_data() + v.size(), 0.0);
        auto seq = h->tf->opf->seq;
        seq->gemms.clear(), seq->ga_ptr.clear(), seq->gb_ptr.clear(), seq->gc_ptr.clear(), seq->gemm_nflop = 0;
        h->record_perturbative_noise(trace_right, psubsl, vdqs, vinfos, voffs, S(arr<uint64_t>(d, "noise.vacuum")[0]), id,
                                     psi.mutable_data(), v.mutable_data());
        std::vector<b2x_gemm> g = seq->gemms;
        for (size_t i = 0; i < g.size(); i++) {
            auto cls = [&](const double *p, uint8_t &src, uint64_t &off) {
                if (p >= psi.data() && p < psi.data() + psi.size())
                    src = 1, off = (uint64_t)(p - psi.data());
                else
                    src = 0, off = (uint64_t)(p - arena.data());
            };
            cls(seq->ga_ptr[i], g[i].a_src, g[i].a_off), cls(seq->gb_ptr[i], g[i].b_src, g[i].b_off);
            g[i].c_off = (uint64_t)(seq->gc_ptr[i] - v.data());
        }
        py::array_t<uint8_t> ga(g.size() * sizeof(b2x_gemm));
        std::memcpy(ga.mutable_data(), g.data(), g.size() * sizeof(b2x_gemm));
        if (execute)
            seq->auto_perform(GMatrix(v.mutable_data(), (int)v.size(), 1), GMatrix(psi.mutable_data(), (int)psi.size(), 1));
        else
            seq->gemms.clear(), seq->ga_ptr.clear(), seq->gb_ptr.clear(), seq->gc_ptr.clear(), seq->gemm_nflop = 0;
        return py::make_tuple(ga, v);
    }
    py::tuple eigs(std::vector<double> ket, double conv_thrd, int max_iter) override {
        auto r = h->eigs(ket, conv_thrd, max_iter);
        return py::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r),
                              py::array_t<double>(ket.size(), ket.data()));
    }
};

// Environment rotation from a symbolic-level fixture (oracle/ref_dump.cpp `erot=`): operator infos of the enlarged and
// of the rotated block, MPS tensor infos, data.  TensorFunctions::left_rotate / right_rotate records the pairs through
// OperatorFunctions::tensor_rotate; execute = false returns them (offsets relative to x / v / arena, no device needed),
// execute = true also runs them on the device (BatchGEMMSeq::rotate_perform) and returns the rotated blocks.
template <typename S> py::tuple sym_rotate(const py::dict &d, bool execute) {
    typedef SparseMatrixInfo<S> Info;
    std::map<int, std::shared_ptr<Info>> cache;
    py::array_t<double> x = py::array_t<double>(py::array::ensure(d["x"]));
    py::array_t<double> arena = py::array_t<double>(py::array::ensure(d["arena"]));
    const uint64_t *meta = SymEH<S>::template arr<uint64_t>(d, "meta");
    const bool right = meta[2] != 0;
    py::array_t<double> v((py::ssize_t)meta[6]);
    if (execute) // (recording needs the ADDRESSES of the rotated blocks only: the pages of v are not touched, and v is returned uninitialised)
        std::fill(v.mutable_data(), v.mutable_data() + v.size(), 0.0);
    size_t n;
    const int64_t *ai = SymEH<S>::template arr<int64_t>(d, "a.info", &n), *ci = SymEH<S>::template arr<int64_t>(d, "c.info"),
                  *ao = SymEH<S>::template arr<int64_t>(d, "a.off"), *co = SymEH<S>::template arr<int64_t>(d, "c.off");
    const double *af = SymEH<S>::template arr<double>(d, "a.factor");
    OperatorTensor<S> a, c;
    std::vector<std::pair<const double *, size_t>> ab;
    std::vector<std::pair<double *, size_t>> cb;
    for (size_t i = 0; i < n; i++) {
        auto am = std::make_shared<SparseMatrix<S>>(), cm = std::make_shared<SparseMatrix<S>>();
        am->info = SymEH<S>::info(d, (int)ai[i], cache), cm->info = SymEH<S>::info(d, (int)ci[i], cache);
        am->factor = af[i], cm->factor = 1.0;
        am->data = x.mutable_data() + ao[i], cm->data = v.mutable_data() + co[i];
        am->total_memory = am->info->get_total_memory(), cm->total_memory = cm->info->get_total_memory();
        a.ops.push_back(am), c.ops.push_back(cm);
        ab.emplace_back(am->data, am->total_memory), cb.emplace_back(cm->data, cm->total_memory);
    }
    const int64_t *mi = SymEH<S>::template arr<int64_t>(d, "mps.info"), *mo = SymEH<S>::template arr<int64_t>(d, "mps.off");
    const double *mf = SymEH<S>::template arr<double>(d, "mps.factor");
    SparseMatrix<S> bra, ket;
    bra.info = SymEH<S>::info(d, (int)mi[0], cache), ket.info = SymEH<S>::info(d, (int)mi[1], cache);
    bra.data = arena.mutable_data() + mo[0], ket.data = arena.mutable_data() + mo[1];
    bra.factor = mf[0], ket.factor = mf[1];
    auto seq = std::make_shared<BatchGEMMSeq>();
    TensorFunctions<S> tf(std::make_shared<OperatorFunctions<S>>(seq));
    {
        py::gil_scoped_release nogil; // (the walk touches no Python object)
        if (right)
            tf.right_rotate(a, bra, ket, c);
        else
            tf.left_rotate(a, bra, ket, c);
    }
    std::vector<b2x_pair> p = seq->pairs;
    for (size_t i = 0; i < p.size(); i++) {
        p[i].x_off = (uint64_t)(((const double *)0 + p[i].x_off) - x.data());
        p[i].v_off = (uint64_t)(((const double *)0 + p[i].v_off) - v.data());
        p[i].y_off = (uint64_t)(seq->y_ptr[i] - arena.data());
        p[i].z_off = (uint64_t)(seq->z_ptr[i] - arena.data());
    }
    py::array_t<uint8_t> pa(p.size() * sizeof(b2x_pair));
    std::memcpy(pa.mutable_data(), p.data(), p.size() * sizeof(b2x_pair));
    if (execute)
        seq->rotate_perform(ab, cb);
    return py::make_tuple(pa, v);
}

// Blocking from a symbolic-level fixture (oracle/ref_dump.cpp `eblk=`): operator infos (the enlarged ones with their
// tensor-product connection info), the flattened expression of every enlarged operator, data.  TensorFunctions::contract
// records the block products through OperatorFunctions::tensor_product; execute = false returns them as b2x_outer_term
// records (block operators = input vector, site operators = arena; no device needed), execute = true also runs them
// (BatchGEMMSeq::outer_perform) and returns the enlarged operators.
template <typename S> py::tuple sym_blocking(const py::dict &d, bool execute) {
    typedef SparseMatrixInfo<S> Info;
    static const bool dbg_clock = getenv("B2X_PLAN_DEBUG") != nullptr; // (development aid: phases of the walk on stderr)
    const auto t_start = std::chrono::steady_clock::now();
    auto t_last = t_start;
    auto lap = [&](const char *what) {
        if (!dbg_clock)
            return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[b2x blocking] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t_last).count());
        t_last = n;
    };
    std::map<int, std::shared_ptr<Info>> cache;
    py::array_t<double> x = py::array_t<double>(py::array::ensure(d["x"]));
    py::array_t<double> site = py::array_t<double>(py::array::ensure(d["site"]));
    const uint64_t *meta = SymEH<S>::template arr<uint64_t>(d, "meta");
    const bool right = meta[2] != 0;
    if (!meta[3])
        throw std::runtime_error("fixture holds expression forms this mirror does not cover (operator sums without an intermediate)");
    py::array_t<double> v((py::ssize_t)meta[7]);
    if (execute) // (as in sym_rotate: a recording walk never reads or writes v — an enlarged block is tens of MB that would be faulted in)
        std::fill(v.mutable_data(), v.mutable_data() + v.size(), 0.0);
    // lop / rop as TensorFunctions::tensor_product receives them: left blocking (block, site), right blocking (site, block)
    auto load = [&](const std::string &pre, double *base) {
        OperatorTensor<S> t;
        size_t n;
        const int64_t *iid = SymEH<S>::template arr<int64_t>(d, pre + ".info", &n), *off = SymEH<S>::template arr<int64_t>(d, pre + ".off");
        const double *fac = SymEH<S>::template arr<double>(d, pre + ".factor");
        for (size_t i = 0; i < n; i++) {
            auto m = std::make_shared<SparseMatrix<S>>();
            m->info = SymEH<S>::info(d, (int)iid[i], cache);
            // (an operator without data has no blocks or is absent on this side: it contributes no connection)
            m->factor = fac[i], m->data = off[i] < 0 ? nullptr : base + off[i];
            m->total_memory = off[i] < 0 ? 0 : m->info->get_total_memory();
            t.ops.push_back(m);
        }
        return t;
    };
    OperatorTensor<S> lop = load("lop", right ? site.mutable_data() : x.mutable_data());
    OperatorTensor<S> rop = load("rop", right ? x.mutable_data() : site.mutable_data());
    std::vector<int> tmp_pos_l, tmp_pos_r; // temporary t -> its position in lop.ops / rop.ops (-1: other side)
    size_t tmp_nl = lop.ops.size(), tmp_nr = rop.ops.size();
    size_t nc;
    const int64_t *ci = SymEH<S>::template arr<int64_t>(d, "c.info", &nc), *co = SymEH<S>::template arr<int64_t>(d, "c.off"),
                  *tb = SymEH<S>::template arr<int64_t>(d, "c.term_begin");
    const int64_t *ty = SymEH<S>::template arr<int64_t>(d, "term.type"), *cj = SymEH<S>::template arr<int64_t>(d, "term.conj"),
                  *ta = SymEH<S>::template arr<int64_t>(d, "term.a"), *tbi = SymEH<S>::template arr<int64_t>(d, "term.b");
    const double *tf = SymEH<S>::template arr<double>(d, "term.factor");
    // Temporaries (sum-MPO fixtures: operator sums with transposed members, no stored intermediate): tmp_t = sum_k factor_k *
    // op_k (or its transpose) with the info of member 0, TensorFunctions::tensor_product's SumProd branch
    // (tensor_functions.hpp:2236-2261).  They live behind x in the input vector's address space; their sums are recorded
    // into a sequence of their own (executed BEFORE the products), and a term refers to temporary t as operator number
    // (operators of that side) + t.
    std::vector<double> tmpbuf;
    auto seq_tmp = std::make_shared<BatchGEMMSeq>();
    if (d.contains("tmp.side")) {
        size_t nt;
        const int64_t *tside = SymEH<S>::template arr<int64_t>(d, "tmp.side", &nt), *tbeg = SymEH<S>::template arr<int64_t>(d, "tmp.begin"),
                      *top = SymEH<S>::template arr<int64_t>(d, "tmp.op"), *tcj = SymEH<S>::template arr<int64_t>(d, "tmp.conj");
        const double *tfac = SymEH<S>::template arr<double>(d, "tmp.factor");
        size_t tot = 0;
        std::vector<size_t> toff(nt);
        for (size_t t = 0; t < nt; t++) {
            const OperatorTensor<S> &side = tside[t] ? lop : rop;
            if (tbeg[t + 1] <= tbeg[t] || top[tbeg[t]] < 0 || top[tbeg[t]] >= (int64_t)side.ops.size())
                throw std::runtime_error("symbolic_blocking: malformed temporary");
            toff[t] = tot, tot += side.ops[top[tbeg[t]]]->info->get_total_memory();
        }
        tmpbuf.assign(tot + 1, 0.0);
        OperatorFunctions<S> opf_tmp(seq_tmp);
        const size_t nl = lop.ops.size(), nr = rop.ops.size();
        for (size_t t = 0; t < nt; t++) {
            OperatorTensor<S> &side = tside[t] ? lop : rop;
            const size_t n_side = tside[t] ? nl : nr;
            auto m = std::make_shared<SparseMatrix<S>>();
            m->info = side.ops[top[tbeg[t]]]->info;
            m->factor = 1.0, m->data = tmpbuf.data() + toff[t], m->total_memory = m->info->get_total_memory();
            for (int64_t k = tbeg[t]; k < tbeg[t + 1]; k++) {
                if (top[k] < 0 || top[k] >= (int64_t)n_side)
                    throw std::runtime_error("symbolic_blocking: temporary refers to an unknown operator");
                if (side.ops[top[k]]->data) // (a member without data is zero on this rank)
                    opf_tmp.iadd(*m, *side.ops[top[k]], tfac[k], tcj[k] != 0);
            }
            side.ops.push_back(m); // index n_side + (number of temporaries of this side so far): the generator counts per fixture
        }
        // the generator numbers temporaries over BOTH sides (t = running index): re-index so that term.a / term.b find them
        // (operators of the side + t  ->  position in the side's list)
        std::vector<int> pos_l, pos_r;
        for (size_t t = 0, il = nl, ir = nr; t < nt; t++)
            if (tside[t])
                pos_l.push_back((int)il++), pos_r.push_back(-1);
            else
                pos_r.push_back((int)ir++), pos_l.push_back(-1);
        tmp_pos_l = pos_l, tmp_pos_r = pos_r, tmp_nl = nl, tmp_nr = nr;
    }
    OperatorTensor<S> c;
    std::vector<std::vector<OpTerm>> exprs(nc);
    for (size_t i = 0; i < nc; i++) {
        auto m = std::make_shared<SparseMatrix<S>>();
        m->info = SymEH<S>::info(d, (int)ci[i], cache);
        m->factor = 1.0, m->data = v.mutable_data() + co[i], m->total_memory = m->info->get_total_memory();
        c.ops.push_back(m);
        for (int64_t k = tb[i]; k < tb[i + 1]; k++) {
            OpTerm t;
            t.factor = tf[k], t.conj = (uint8_t)cj[k], t.a = (int)ta[k], t.b = (int)tbi[k];
            if (ty[k] == 2) { // product with a temporary
                if (t.a >= (int)tmp_nl && t.a - (int)tmp_nl < (int)tmp_pos_l.size() && tmp_pos_l[t.a - tmp_nl] >= 0)
                    t.a = tmp_pos_l[t.a - tmp_nl];
                else if (t.b >= (int)tmp_nr && t.b - (int)tmp_nr < (int)tmp_pos_r.size() && tmp_pos_r[t.b - tmp_nr] >= 0)
                    t.b = tmp_pos_r[t.b - tmp_nr];
                else
                    throw std::runtime_error("symbolic_blocking: term refers to an unknown temporary");
            }
            exprs[i].push_back(t);
        }
    }
    lap("infos + expressions");
    auto seq = std::make_shared<BatchGEMMSeq>();
    TensorFunctions<S> tfn(std::make_shared<OperatorFunctions<S>>(seq));
    {
        py::gil_scoped_release nogil; // (the walk touches no Python object)
        tfn.contract(lop, rop, c, exprs);
    }
    lap("walk");
    // the records go straight into the array that is returned (one pass over 64 bytes per term: a blocking list of 4e5 terms
    // was copied three times through freshly mapped memory, which cost more than the walk itself)
    const size_t nt = seq->outer_terms.size();
    py::array_t<uint8_t> pa(nt * sizeof(b2x_outer_term));
    b2x_outer_term *t = reinterpret_cast<b2x_outer_term *>(pa.mutable_data());
    const double *x0 = x.data(), *x1 = x0 + x.size(), *s0 = site.data(), *s1 = s0 + site.size();
    const double *m0 = tmpbuf.empty() ? nullptr : tmpbuf.data(), *m1 = tmpbuf.empty() ? nullptr : m0 + tmpbuf.size();
    const uint64_t x_len = (uint64_t)x.size();
    auto classify = [&](const double *p, uint8_t &src, uint64_t &off) {
        if (p >= x0 && p < x1)
            src = 1, off = (uint64_t)(p - x0);
        else if (p >= s0 && p < s1)
            src = 0, off = (uint64_t)(p - s0);
        else if (m0 && p >= m0 && p < m1)
            src = 1, off = x_len + (uint64_t)(p - m0); // temporaries: behind x in the input vector
        else
            throw std::runtime_error("symbolic_blocking: operand outside the fixture's data");
    };
    const double *v0 = v.data();
    lap("array");
    for (size_t i = 0; i < nt; i++) {
        t[i] = seq->outer_terms[i];
        classify(seq->oa_ptr[i], t[i].a_src, t[i].a_off);
        classify(seq->ob_ptr[i], t[i].b_src, t[i].b_off);
        t[i].c_off = (uint64_t)(seq->oc_ptr[i] - v0);
    }
    lap("offsets + copy");
    if (tmpbuf.empty()) {
        if (execute)
            seq->outer_perform({{v.mutable_data(), (size_t)v.size()}});
        return py::make_tuple(pa, v);
    }
    // with temporaries: (terms, v, sum terms of the temporaries, length of the temporaries' area).  The sum terms read x /
    // the site operators and write the area behind x: input and output vector of that pass are the SAME extended vector.
    if (execute)
        throw std::runtime_error("symbolic_blocking: execute = true is not built for fixtures with temporaries");
    std::vector<b2x_outer_term> ts = seq_tmp->outer_terms;
    for (size_t i = 0; i < ts.size(); i++) {
        classify(seq_tmp->oa_ptr[i], ts[i].a_src, ts[i].a_off);
        ts[i].b_src = 2, ts[i].b_off = 0;
        ts[i].c_off = (uint64_t)x.size() + (uint64_t)(seq_tmp->oc_ptr[i] - tmpbuf.data());
    }
    py::array_t<uint8_t> ps(ts.size() * sizeof(b2x_outer_term));
    std::memcpy(ps.mutable_data(), ts.data(), ts.size() * sizeof(b2x_outer_term));
    return py::make_tuple(pa, v, ps, (size_t)(tmpbuf.size() - 1));
}

// Numerical transform from a symbolic-level fixture (oracle/ref_dump.cpp chain mode, `.entr`): the operators of one block
// (infos, offsets into one vector) and, per new operator, the terms  new += factor * op  /  factor * op^T.
// TensorFunctions::numerical_transform walks them through OperatorFunctions::iadd; the recorded element-wise terms read
// and write the SAME vector (inputs and outputs are different operators); returned as b2x_outer_term records.
template <typename S> py::array sym_transform(const py::dict &d) {
    typedef SparseMatrixInfo<S> Info;
    std::map<int, std::shared_ptr<Info>> cache;
    const uint64_t *meta = SymEH<S>::template arr<uint64_t>(d, "meta");
    std::unique_ptr<double[]> t_mem(new double[(size_t)meta[3] + 1]); // address space of the block's vector: never read, never
    double *const t_base = t_mem.get();                               // written, so its pages are never touched
    size_t n;
    const int64_t *iid = SymEH<S>::template arr<int64_t>(d, "t.info", &n), *off = SymEH<S>::template arr<int64_t>(d, "t.off");
    const double *fac = SymEH<S>::template arr<double>(d, "t.factor");
    std::vector<std::shared_ptr<SparseMatrix<S>>> ops;
    for (size_t i = 0; i < n; i++) {
        auto m = std::make_shared<SparseMatrix<S>>();
        m->info = SymEH<S>::info(d, (int)iid[i], cache);
        m->factor = fac[i], m->data = off[i] < 0 ? nullptr : t_base + off[i];
        m->total_memory = off[i] < 0 ? 0 : m->info->get_total_memory();
        ops.push_back(m);
    }
    size_t nn;
    const int64_t *nop = SymEH<S>::template arr<int64_t>(d, "new.op", &nn), *tb = SymEH<S>::template arr<int64_t>(d, "new.term_begin");
    const int64_t *top = SymEH<S>::template arr<int64_t>(d, "term.op"), *tcj = SymEH<S>::template arr<int64_t>(d, "term.conj");
    const double *tf = SymEH<S>::template arr<double>(d, "term.factor");
    auto seq = std::make_shared<BatchGEMMSeq>();
    OperatorFunctions<S> opf(seq);
    for (size_t k = 0; k < nn; k++)
        for (int64_t j = tb[k]; j < tb[k + 1]; j++) {
            if (nop[k] < 0 || top[j] < 0)
                throw std::runtime_error("symbolic_transform: unknown operator");
            // (an operator without data is identically zero here — in a sum-MPO run every rank holds a part of the integrals
            // and some of its operators vanish; the reference's iadd skips them, operator_functions.hpp:135-174)
            if (!ops[nop[k]]->data || !ops[top[j]]->data)
                continue;
            opf.iadd(*ops[nop[k]], *ops[top[j]], tf[j], tcj[j] != 0);
        }
    std::vector<b2x_outer_term> terms = seq->outer_terms;
    for (size_t i = 0; i < terms.size(); i++) {
        terms[i].a_src = 1, terms[i].a_off = (uint64_t)(seq->oa_ptr[i] - t_base);
        terms[i].b_src = 2, terms[i].b_off = 0;
        terms[i].c_off = (uint64_t)(seq->oc_ptr[i] - t_base);
    }
    py::array_t<uint8_t> pa(terms.size() * sizeof(b2x_outer_term));
    std::memcpy(pa.mutable_data(), terms.data(), terms.size() * sizeof(b2x_outer_term));
    return pa;
}

// SparseMatrix on-disk format of the reference (MPS tensors, operator blocks): load -> dict of arrays, and save
template <typename S> py::dict sm_load(const std::string &fn) {
    SparseMatrix<S> m;
    m.load_data(fn, true);
    py::dict r;
    std::vector<uint64_t> q;
    for (auto x : m.info->quanta)
        q.push_back(x.data);
    r["quanta"] = q, r["nbra"] = m.info->n_states_bra, r["nket"] = m.info->n_states_ket, r["ntot"] = m.info->n_states_total;
    r["meta"] = std::vector<uint64_t>{m.info->delta_quantum.data, (uint64_t)m.info->is_fermion,
                                      (uint64_t)m.info->is_wavefunction, (uint64_t)m.total_memory};
    r["factor"] = m.factor;
    r["data"] = py::array_t<double>(m.total_memory, m.data);
    return r;
}
template <typename S>
void sm_save(const std::string &fn, const std::vector<uint64_t> &q, const std::vector<uint32_t> &nb,
             const std::vector<uint32_t> &nk, const std::vector<uint32_t> &nt, const std::vector<uint64_t> &meta, double factor,
             py::array_t<double, py::array::c_style> data, double fp_prec = 0.0, size_t fp_chunk = 4096) {
    SparseMatrix<S> m;
    m.info = std::make_shared<SparseMatrixInfo<S>>();
    m.info->n = (int)q.size();
    for (auto x : q)
        m.info->quanta.push_back(S(x));
    m.info->n_states_bra = nb, m.info->n_states_ket = nk, m.info->n_states_total = nt;
    m.info->delta_quantum = S(meta[0]), m.info->is_fermion = meta[1] != 0, m.info->is_wavefunction = meta[2] != 0;
    m.factor = factor, m.total_memory = (size_t)data.size(), m.data = data.mutable_data();
    if (fp_prec > 0.0) {
        FPCodec codec(fp_prec, fp_chunk);
        m.save_data(fn, true, &codec);
    } else
        m.save_data(fn, true);
}

inline void bind_symbolic(py::module_ &m) {
    m.def("sparse_matrix_load", [](const std::string &sym, const std::string &fn) {
        return sym == "su2" ? sm_load<SU2>(fn) : sm_load<SZ>(fn);
    });
    m.def("sparse_matrix_save", [](const std::string &sym, const std::string &fn, std::vector<uint64_t> q, std::vector<uint32_t> nb,
                                   std::vector<uint32_t> nk, std::vector<uint32_t> nt, std::vector<uint64_t> meta, double factor,
                                   py::array_t<double, py::array::c_style> data, double fp_prec, size_t fp_chunk) {
        if (sym == "su2")
            sm_save<SU2>(fn, q, nb, nk, nt, meta, factor, data, fp_prec, fp_chunk);
        else
            sm_save<SZ>(fn, q, nb, nk, nt, meta, factor, data, fp_prec, fp_chunk);
    }, py::arg("sym"), py::arg("filename"), py::arg("quanta"), py::arg("nbra"), py::arg("nket"), py::arg("ntot"), py::arg("meta"),
       py::arg("factor"), py::arg("data"), py::arg("fp_prec") = 0.0, py::arg("fp_chunk") = (size_t)4096);
    // FPCodec of the reference's scratch files on plain arrays: encode -> the byte stream write_array produces
    m.def("fpcodec_encode", [](py::array_t<double, py::array::c_style> a, double prec, size_t chunk) {
        FPCodec c(prec, chunk);
        std::ostringstream os;
        c.write_array(os, a.data(), (size_t)a.size());
        return py::bytes(os.str());
    }, py::arg("data"), py::arg("prec"), py::arg("chunk") = (size_t)4096);
    m.def("fpcodec_decode", [](py::bytes b, size_t len) {
        std::istringstream is((std::string)b);
        py::array_t<double> out((py::ssize_t)len);
        FPCodec::read_array(is, out.mutable_data(), len);
        return out;
    });
    m.def("symbolic_blocking", [](const std::string &sym, const py::dict &d, bool execute) {
        if (sym == "sz")
            return sym_blocking<SZ>(d, execute);
        if (sym == "su2")
            return sym_blocking<SU2>(d, execute);
        throw std::runtime_error("symmetry must be 'sz' or 'su2'");
    }, py::arg("sym"), py::arg("fixture"), py::arg("execute") = false);
    m.def("symbolic_rotate", [](const std::string &sym, const py::dict &d, bool execute) {
        if (sym == "sz")
            return sym_rotate<SZ>(d, execute);
        if (sym == "su2")
            return sym_rotate<SU2>(d, execute);
        throw std::runtime_error("symmetry must be 'sz' or 'su2'");
    }, py::arg("sym"), py::arg("fixture"), py::arg("execute") = false);
    m.def("symbolic_transform", [](const std::string &sym, const py::dict &d) {
        if (sym == "sz")
            return sym_transform<SZ>(d);
        if (sym == "su2")
            return sym_transform<SU2>(d);
        throw std::runtime_error("symmetry must be 'sz' or 'su2'");
    }, py::arg("sym"), py::arg("fixture"));
    py::class_<SymEHBase, std::shared_ptr<SymEHBase>>(m, "SymbolicEffectiveHamiltonian")
        .def(py::init([](const std::string &sym, const py::dict &d) -> std::shared_ptr<SymEHBase> {
            if (sym == "sz")
                return std::make_shared<SymEH<SZ>>(d);
            if (sym == "su2")
                return std::make_shared<SymEH<SU2>>(d);
            throw std::runtime_error("symmetry must be 'sz' or 'su2'");
        }))
        .def("wfn_cinfo", &SymEHBase::wfn_cinfo)
        .def("precompute", &SymEHBase::precompute)
        .def("post_precompute", &SymEHBase::post_precompute)
        .def_property_readonly("n_pairs", &SymEHBase::n_pairs)
        .def_property_readonly("nflop", &SymEHBase::nflop)
        .def("pairs", &SymEHBase::pairs)
        .def("record", &SymEHBase::record)
        .def("compute_diag", &SymEHBase::compute_diag)
        .def("diag_terms", &SymEHBase::diag_terms)
        .def("__call__", &SymEHBase::apply, py::arg("b"), py::arg("c"), py::arg("factor") = 1.0)
        .def("perturbative_noise", &SymEHBase::perturbative_noise, py::arg("fixture"), py::arg("execute") = false)
        .def("eigs", &SymEHBase::eigs, py::arg("ket"), py::arg("conv_thrd") = 5E-6, py::arg("max_iter") = 5000);
    // label algebra exposed for unit tests (packed 64-bit in / out)
    m.def("su2_add", [](uint64_t a, uint64_t b) { return (SU2(a) + SU2(b)).data; });
    m.def("su2_combine", [](uint64_t dq, uint64_t bra, uint64_t ket) { return SU2(dq).combine(SU2(bra), SU2(ket)).data; });
    m.def("su2_get_bra", [](uint64_t a, uint64_t dq) { return SU2(a).get_bra(SU2(dq)).data; });
    m.def("su2_make", [](int n, int tl, int t, int pg) { return SU2(n, tl, t, pg).data; });
    m.def("sz_make", [](int n, int t, int pg) { return SZ(n, t, pg).data; });
    m.def("sz_add", [](uint64_t a, uint64_t b) { return (SZ(a) + SZ(b)).data; });
    m.def("sz_neg", [](uint64_t a) { return (-SZ(a)).data; });
    m.def("wigner_6j", [](int a, int b, int c, int d, int e, int f) { return (double)CG<SU2>().wigner_6j(a, b, c, d, e, f); });
    m.def("wigner_9j", [](int a, int b, int c, int d, int e, int f, int g, int h, int i) {
        return (double)CG<SU2>().wigner_9j_2j(a, b, c, d, e, f, g, h, i);
    });
}

} // namespace b2xh
