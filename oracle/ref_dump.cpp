/*
 * ref_dump.cpp — fixture generator.  TEST INFRASTRUCTURE, built ONLY in the authoring
 * container against the read-only reference headers (see oracle/Makefile, target _ref/ref_dump);
 * nothing of the reference is copied: this file only *calls* block2's public API, the way
 * block2's own unit tests do (unit_test/test_dmrg_n2_sto3g.cpp:51-148).
 *
 * It runs the reference two-site DMRG on an FCIDUMP and, at the stage callback
 * "DMRG::sweep::iter.eff_ham" (src/dmrg/sweep_algorithm.hpp:1235-1237), captures for selected
 * (sweep, site) the GEMM-pair plan that EffectiveHamiltonian::precompute() records
 * (src/dmrg/effective_hamiltonian.hpp:224-244), the operator blocks it points into, psi, diag,
 * and the reference sigma = H psi computed by TensorFunctions::operator()
 * (src/core/tensor_functions.hpp:59-62 -> BatchGEMMSeq::operator(), src/core/batch_gemm.hpp:1563).
 *
 * usage: ref_dump <fcidump|hubbard:L:t:U> <su2|sz> <M> <n_sweeps> <outprefix> [key=value ...]
 *   dump=<sweep>:<site>[,<sweep>:<site>...]  plans to capture with data
 *   struct=<sweep>:<site>[,...]              plans to capture WITHOUT data (structure only)
 *   eham=<sweep>:<site>[,...]                effective-Hamiltonian level fixtures (infos, tensors, expression)
 *   pnoise=<sweep>:<site>[,...]              single-GEMM list of the perturbative noise (with data + reference result)
 *   pnoise_struct=<sweep>:<site>[,...]       the same list without data
 *   enoise=<sweep>:<site>[,...]              the perturbative noise at the symbolic level (eham content + sub-labels,
 *                                            perturbed-wavefunction infos, reference result)
 *   rot=<sweep>:<center>[,...]               environment rotation (TensorFunctions::left_rotate / right_rotate called
 *                                            while MovingEnvironment::center == <center>): GEMM-pair plan + data + result
 *   rot_struct=<sweep>:<center>[,...]        the same plan without data
 *   erot=<sweep>:<center>[,...]              the rotation at the symbolic level (operator infos, MPS tensor infos, data)
 *   blk=<sweep>:<center>[,...]               blocking (TensorFunctions::left_contract / right_contract called while
 *                                            center == <center>): element-wise block-product terms + data + result
 *   blk_struct=<sweep>:<center>[,...]        the same terms without data
 *   eblk=<sweep>:<center>[,...]              blocking at the symbolic level (operator infos incl. the tensor-product
 *                                            connection infos, the expression of every enlarged operator, data)
 *   tensor_file=<i>[,<j>...]                 after the run: MPS tensor i written by the reference's own
 *                                            SparseMatrix::save_data(file, true) (src/core/sparse_matrix.hpp:957-971) next to
 *                                            its content as named arrays (on-disk format fixture)
 *   fp_prec=<x> [fp_chunk=<n>]               (with tensor_file=) also write the tensor in compressed storage, <file>.fpc
 *   occ=<file>   nthreads=<n>   seed=<n>   noise=<a,b,c>   tol=<x>   dav_iter=<n>  pg=<d2h|c1>
 *   stack_gb=<n>  main_stack=1  keep_part=1   size of the double stack; operators on the frame stack (block2's default);
 *                                            partition files left in the scratch directory
 *   cutoff=<x>                               DMRG::cutoff (smallest density-matrix weight that may be kept; default 1e-14)
 *   dav_thrd=<x>                             Davidson threshold of EVERY sweep (the default loosens it to noise / 10 in noisy sweeps)
 *   prefactors=1                             (with para=) also write ParallelRuleSimple::index_prefactor of every (i,j), (i,j,k,l)
 *   para=i|ij                                sum-MPO parallel rule (ParallelRuleSimple I / IJ); under mpirun with the
 *                                            _HAS_MPI build every rank writes <outprefix>.r<rank>of<size>.*
 *   chain=<last sweep>  nodelay=1  nocache=1  every blocking / rotation / effective Hamiltonian from the initial environments
 *                                            up to that sweep at the symbolic level, numbered in call order, without bulk data
 *                                            (fixtures of the site-to-site chain; needs nodelay=1)
 *                                            with noise != 0 the perturbative-noise step of every site is an event of its own
 *                                            (<kind> = enoise: the eham content + sub-labels + perturbed-ket infos, no bulk data)
 *   spectra=1                                log, per site, the FULL density-matrix spectrum the reference truncated
 *                                            (DMRG::store_wfn_spectra: sqrt of every eigenvalue, sector by sector, before the cut)
 *                                            and the discarded weight: "SPECTRA <sweep> <site> <error> <n> <values...>"
 *   (always)                                 "SWEEP_TIME <sweep> <wall s> teff teig tprt tblk tmve tdm tsplt tsvd ndav nflop":
 *                                            the reference's own per-sweep timers (sweep_algorithm.hpp:3208-3217)
 *   stop_after=<sweep>:<site>                leave the run right after the captures of that site (large-M structure runs)
 */
#include "block2_core.hpp"
#include "block2_dmrg.hpp"
#include "planfile.h"
#include <execinfo.h>
#include <map>
#include <set>
#include <signal.h>
#include <unistd.h>

using namespace block2;
using namespace std;

struct DumpSpec {
    set<pair<int, int>> with_data, structure, eham, pnoise, pnoise_struct, enoise, rot, rot_struct, erot, blk, blk_struct, eblk;
    string prefix;
    // chain=<last sweep>: EVERY blocking, rotation and effective Hamiltonian of the run up to that sweep (the initial
    // environments included) is written at the symbolic level, numbered in call order (<prefix>.ev<NNN>.<kind>), without
    // the bulk data a replay produces itself (operators, wavefunctions): what remains are the infos, the expressions,
    // operator keys, the site operators and — for the initial environments — the MPS tensors of the starting state
    int chain = -2;
    mutable int ev = 0;
    mutable vector<string> *evlog = nullptr;
    bool lite() const { return chain > -2; }
    string next_event(const string &kind, int isw, int center) const {
        char buf[64];
        snprintf(buf, sizeof(buf), ".ev%03d.", ev);
        stringstream ss;
        ss << "EVENT " << ev << " " << kind << " " << isw << " " << center;
        if (evlog)
            evlog->push_back(ss.str());
        ev++;
        return prefix + buf + kind;
    }
};

// a key that identifies an operator across tensors and fixtures: hash of its printed name (e.g. "R[ 3 ]", "H")
template <typename S> static uint64_t op_key(const shared_ptr<OpExpr<S>> &x) {
    stringstream ss;
    ss << abs_value(x);
    return (uint64_t)std::hash<string>()(ss.str());
}


// ---- named-array container (B2XARR01) for effective-Hamiltonian level fixtures ------------------------
// record: u32 name_len, name, u8 dtype (0 = u64, 1 = i64, 2 = f64, 3 = u32, 4 = u8), u64 count, data
struct ArrayFile {
    FILE *f;
    explicit ArrayFile(const string &fn) : f(fopen(fn.c_str(), "wb")) { fwrite("B2XARR01", 1, 8, f); }
    ~ArrayFile() { fclose(f); }
    void put(const string &name, uint8_t dt, size_t esz, const void *data, uint64_t n) {
        uint32_t l = (uint32_t)name.size();
        fwrite(&l, 4, 1, f), fwrite(name.data(), 1, l, f), fwrite(&dt, 1, 1, f), fwrite(&n, 8, 1, f);
        if (n)
            fwrite(data, esz, n, f);
    }
    void u64(const string &n, const vector<uint64_t> &v) { put(n, 0, 8, v.data(), v.size()); }
    void i64(const string &n, const vector<int64_t> &v) { put(n, 1, 8, v.data(), v.size()); }
    void f64(const string &n, const vector<double> &v) { put(n, 2, 8, v.data(), v.size()); }
    void f64(const string &n, const double *p, size_t c) { put(n, 2, 8, p, c); }
    // bulk data a chain replay computes itself: written as an empty array in chain mode (the length goes to "<name>.len")
    bool lite = false;
    void bulk(const string &n, const double *p, size_t c) {
        const uint64_t cc = c;
        put(n + ".len", 0, 8, &cc, 1);
        put(n, 2, 8, p, lite ? 0 : c);
    }
    void bulk(const string &n, const vector<double> &v) { bulk(n, v.data(), v.size()); }
    void u32(const string &n, const vector<uint32_t> &v) { put(n, 3, 4, v.data(), v.size()); }
    void u8(const string &n, const vector<uint8_t> &v) { put(n, 4, 1, v.data(), v.size()); }
};

// Everything the symbolic -> numeric layer of one EffectiveHamiltonian consumes and produces:
// operator infos, operator tensors (incl. the delayed left/right enlarged operators), the expression of
// H_eff, the wavefunction infos, and the reference's own ConnectionInfo / plan / sigma for them.
template <typename S> struct EhamDump {
    typedef double FL;
    ArrayFile af;
    map<const SparseMatrixInfo<S> *, int> info_ids;
    set<int> cinfo_done;
    vector<pair<const double *, size_t>> ranges; // operator data ranges
    explicit EhamDump(const string &fn) : af(fn) {}
    template <typename CI> void put_cinfo(const string &pre, const shared_ptr<CI> &ci) {
        vector<int64_t> nn(ci->n, ci->n + 5);
        nn.push_back(ci->nc);
        af.i64(pre + ".n", nn);
        vector<uint64_t> q(ci->n[4]);
        vector<uint32_t> idx(ci->n[4]);
        for (int i = 0; i < ci->n[4]; i++)
            q[i] = ci->quanta[i].data, idx[i] = ci->idx[i];
        af.u64(pre + ".quanta", q), af.u32(pre + ".idx", idx);
        af.u64(pre + ".stride", vector<uint64_t>(ci->stride, ci->stride + ci->nc));
        af.f64(pre + ".factor", ci->factor, ci->nc);
        af.u32(pre + ".ia", vector<uint32_t>(ci->ia, ci->ia + ci->nc));
        af.u32(pre + ".ib", vector<uint32_t>(ci->ib, ci->ib + ci->nc));
        af.u32(pre + ".ic", vector<uint32_t>(ci->ic, ci->ic + ci->nc));
    }
    int info_id(const shared_ptr<SparseMatrixInfo<S>> &info, bool with_cinfo) {
        auto it = info_ids.find(info.get());
        if (it != info_ids.end()) {
            if (with_cinfo && info->cinfo != nullptr && !cinfo_done.count(it->second)) {
                cinfo_done.insert(it->second);
                put_cinfo("info." + Parsing::to_string(it->second) + ".cinfo", info->cinfo);
            }
            return it->second;
        }
        int id = (int)info_ids.size();
        info_ids[info.get()] = id;
        string pre = "info." + Parsing::to_string(id);
        vector<uint64_t> q(info->n);
        vector<uint32_t> nb(info->n), nk(info->n), nt(info->n);
        for (int i = 0; i < info->n; i++)
            q[i] = info->quanta[i].data, nb[i] = info->n_states_bra[i], nk[i] = info->n_states_ket[i],
            nt[i] = info->n_states_total[i];
        af.u64(pre + ".quanta", q), af.u32(pre + ".nbra", nb), af.u32(pre + ".nket", nk), af.u32(pre + ".ntot", nt);
        af.u64(pre + ".meta", vector<uint64_t>{info->delta_quantum.data, (uint64_t)info->is_fermion,
                                               (uint64_t)info->is_wavefunction, (uint64_t)info->get_total_memory()});
        if (with_cinfo && info->cinfo != nullptr) {
            cinfo_done.insert(id);
            put_cinfo(pre + ".cinfo", info->cinfo);
        }
        return id;
    }
    // one operator tensor: per op (info id, factor, has data) + data range; returns the op order
    vector<shared_ptr<OpExpr<S>>> put_tensor(const string &pre, const shared_ptr<OperatorTensor<S, FL>> &t,
                                             bool with_cinfo, vector<const double *> &ptrs) {
        vector<shared_ptr<OpExpr<S>>> order;
        vector<int64_t> iid, has;
        vector<double> fac;
        vector<uint64_t> keys;
        string names; // (printed names, one per line: for people reading a fixture)
        for (auto &kv : t->ops) {
            keys.push_back(op_key<S>(kv.first));
            stringstream ss;
            ss << abs_value(kv.first);
            names += ss.str() + "\n";
        }
        af.u64(pre + ".key", keys);
        af.put(pre + ".names", 4, 1, names.data(), names.size());
        for (auto &kv : t->ops) {
            order.push_back(kv.first);
            iid.push_back(info_id(kv.second->info, with_cinfo));
            fac.push_back(kv.second->factor);
            bool hd = kv.second->data != nullptr && kv.second->total_memory != 0;
            has.push_back(hd ? (int64_t)kv.second->total_memory : -1);
            ptrs.push_back(hd ? kv.second->data : nullptr);
            if (hd)
                ranges.push_back(make_pair((const double *)kv.second->data, (size_t)kv.second->total_memory));
        }
        af.i64(pre + ".info", iid), af.f64(pre + ".factor", fac), af.i64(pre + ".len", has);
        return order;
    }
    // position of symbol x in the dumped op order of tensor t (lookup by the tensor's own hash / equality)
    static int find_op(const shared_ptr<OperatorTensor<S, FL>> &t, const vector<shared_ptr<OpExpr<S>>> &order,
                       const shared_ptr<OpExpr<S>> &x) {
        auto it = t->ops.find(x);
        if (it == t->ops.end())
            return -1;
        for (size_t i = 0; i < order.size(); i++)
            if (order[i].get() == it->first.get())
                return (int)i;
        return -1;
    }
};

// TensorFunctions whose partial multiply (virtual, src/core/tensor_functions.hpp:366) keeps a copy of the
// batch[1] list it recorded, so that the list EffectiveHamiltonian::perturbative_noise replays right afterwards
// (auto_perform(v), src/dmrg/effective_hamiltonian.hpp:397-402) can be written out next to the reference result.
template <typename S> struct CapTF : TensorFunctions<S, double> {
    typedef double FL;
    mutable vector<b2x_gemm> recs;
    mutable vector<const double *> pa, pb;
    mutable vector<double *> pc;
    mutable const double *out_base = nullptr;
    mutable size_t out_len = 0;
    // arguments of the top-level call (what the symbolic walk consumes)
    mutable vector<pair<uint8_t, S>> a_psubsl;
    mutable vector<S> a_vdqs;
    mutable shared_ptr<SparseMatrixGroup<S, FL>> a_vmats;
    mutable bool a_trace_right = false;
    mutable int a_vidx = 0, a_tvidx = 0;
    CapTF(const shared_ptr<OperatorFunctions<S, FL>> &opf) : TensorFunctions<S, FL>(opf) {}
    void tensor_product_partial_multiply(
        const shared_ptr<OpExpr<S>> &expr, const shared_ptr<OpExpr<S>> &xexpr,
        const shared_ptr<OperatorTensor<S, FL>> &lopt, const shared_ptr<OperatorTensor<S, FL>> &ropt, bool trace_right,
        const shared_ptr<SparseMatrix<S, FL>> &cmat, const vector<pair<uint8_t, S>> &psubsl,
        const vector<vector<shared_ptr<typename SparseMatrixInfo<S>::ConnectionInfo>>> &cinfos, const vector<S> &vdqs,
        const shared_ptr<SparseMatrixGroup<S, FL>> &vmats, int &vidx, int tvidx, bool do_reduce) const override {
        a_psubsl = psubsl, a_vdqs = vdqs, a_vmats = vmats, a_trace_right = trace_right, a_vidx = vidx, a_tvidx = tvidx;
        // a parallel MPO hands over its LOCAL expression wrapped in an OpExprRef (ParallelTensorFunctions unwraps it the same
        // way, src/core/parallel_tensor_functions.hpp:173-185)
        auto unwrap = [](const shared_ptr<OpExpr<S>> &e) -> shared_ptr<OpExpr<S>> {
            return e != nullptr && e->get_type() == OpTypes::ExprRef ? dynamic_pointer_cast<OpExprRef<S>>(e)->op : e;
        };
        TensorFunctions<S, FL>::tensor_product_partial_multiply(unwrap(expr), unwrap(xexpr), lopt, ropt, trace_right, cmat, psubsl,
                                                                cinfos, vdqs, vmats, vidx, tvidx, false);
        auto b0 = this->opf->seq->batch[0], b1 = this->opf->seq->batch[1];
        assert(b0->c.size() == 0 && b1->acidxs.size() == 0);
        size_t n = b1->c.size();
        assert(b1->gp.size() == n);
        recs.resize(n), pa.resize(n), pb.resize(n), pc.resize(n);
        for (size_t i = 0; i < n; i++) {
            b2x_gemm &g = recs[i];
            memset(&g, 0, sizeof(g));
            g.m = b1->m[i], g.n = b1->n[i], g.k = b1->k[i];
            g.lda = b1->lda[i], g.ldb = b1->ldb[i], g.ldc = b1->ldc[i];
            g.ta = (b1->ta[i] == CblasTrans || b1->ta[i] == CblasConjTrans);
            g.tb = (b1->tb[i] == CblasTrans || b1->tb[i] == CblasConjTrans);
            g.alpha = b1->alpha[i];
            assert(b1->beta[i] == 1.0 && b1->gp[i] == 1);
            pa[i] = b1->a[i], pb[i] = b1->b[i], pc[i] = b1->c[i];
        }
        out_base = vmats->data, out_len = vmats->total_memory;
    }
};

// Environment rotation: TensorFunctions::left_rotate / right_rotate
// (src/core/tensor_functions.hpp:2365-2403) rotate every operator of the enlarged block into the truncated basis,
// c[op] = bra^T a[op] ket block by block (OperatorFunctions::tensor_rotate, src/core/operator_functions.hpp:175-210).
// The override lets the reference compute the result, then asks the reference's OWN tensor_rotate to record the same
// work into a BatchGEMMSeq in SeqTypes::Auto mode (seq->rotate, :202-204) and writes that pair list as a plan file:
//   psi   := the operator blocks of a  (stage-0 A operands),   arena := the MPS tensor(s) (bra / ket),
//   sigma := the operator blocks of c  (stage-1 outputs), sigma_ref = what the reference computed.
// Base = TensorFunctions (serial) or ParallelTensorFunctions (sum-MPO run under mpirun: every rank records ITS events)
template <typename S, typename Base = TensorFunctions<S, double>> struct RotTFT : Base {
    using Base::opf;
    typedef double FL;
    DMRG<S, FL, FL> *dmrg = nullptr;
    MovingEnvironment<S, FL, FL> *me = nullptr; // (chain mode: the initial environments are built before a DMRG exists)
    const DumpSpec *spec = nullptr;
    mutable vector<string> *log = nullptr;
    template <typename... Args> RotTFT(Args &&...args) : Base(std::forward<Args>(args)...) {}
    bool in_chain() const { return spec != nullptr && spec->lite() && (dmrg == nullptr || dmrg->isweep <= spec->chain); }
    int cur_sweep() const { return dmrg == nullptr ? -1 : dmrg->isweep; }
    int cur_center() const { return dmrg != nullptr ? dmrg->me->center : (me != nullptr ? me->center : -1); }
    void left_rotate(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<SparseMatrix<S, FL>> &mpst_bra,
                     const shared_ptr<SparseMatrix<S, FL>> &mpst_ket, shared_ptr<OperatorTensor<S, FL>> &c) const override {
        Base::left_rotate(a, mpst_bra, mpst_ket, c);
        maybe_capture(a, mpst_bra, mpst_ket, c, false);
    }
    void right_rotate(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<SparseMatrix<S, FL>> &mpst_bra,
                      const shared_ptr<SparseMatrix<S, FL>> &mpst_ket, shared_ptr<OperatorTensor<S, FL>> &c) const override {
        Base::right_rotate(a, mpst_bra, mpst_ket, c);
        maybe_capture(a, mpst_bra, mpst_ket, c, true);
    }
    // ---- blocking: c = a (x) b for every operator of the enlarged block (tensor_functions.hpp:2842-2885, 2941-2983)
    void left_contract(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<OperatorTensor<S, FL>> &b,
                       shared_ptr<OperatorTensor<S, FL>> &c, const shared_ptr<Symbolic<S>> &cexprs = nullptr,
                       OpNamesSet delayed = OpNamesSet()) const override {
        Base::left_contract(a, b, c, cexprs, delayed);
        capture_blocking(a, b, c, cexprs, delayed, false);
    }
    void right_contract(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<OperatorTensor<S, FL>> &b,
                        shared_ptr<OperatorTensor<S, FL>> &c, const shared_ptr<Symbolic<S>> &cexprs = nullptr,
                        OpNamesSet delayed = OpNamesSet()) const override {
        Base::right_contract(a, b, c, cexprs, delayed);
        capture_blocking(a, b, c, cexprs, delayed, true);
    }
    // NC -> CN switch of the conventional MPO near the middle site: new (complementary) operators are linear combinations
    // of operators of the same block, new += factor * op or its transpose (TensorFunctions::numerical_transform,
    // src/core/tensor_functions.hpp:2462-2517 -> OperatorFunctions::iadd, operator_functions.hpp:135-174)
    void numerical_transform(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<Symbolic<S>> &names,
                             const shared_ptr<Symbolic<S>> &exprs) const override {
        // (written AFTER the transform has run: the serial reference has every new operator allocated beforehand, the
        // sum-MPO one allocates them inside — parallel_tensor_functions.hpp:1062-1070 —, and a replay needs the layout that
        // holds them; in chain mode no data are written, only the layout)
        Base::numerical_transform(a, names, exprs);
        if (in_chain()) {
            const bool right = a->lmat == nullptr;
            EhamDump<S> ed(spec->next_event(right ? "rntr" : "lntr", cur_sweep(), cur_center()) + ".entr");
            vector<const double *> ptrs;
            auto order = ed.put_tensor("t", a, false, ptrs);
            Packed2 T;
            T.build(ed.ranges);
            vector<int64_t> off, nop, tb, top, tcj;
            vector<double> tf;
            for (auto p : ptrs) {
                uint64_t o = 0;
                off.push_back(p != nullptr && T.resolve(p, o) ? (int64_t)o : -1);
            }
            for (size_t k = 0; k < names->data.size(); k++) {
                shared_ptr<OpExpr<S>> e0 = exprs->data[k];
                if (e0->get_type() == OpTypes::ExprRef) // sum-MPO: the rank's local expression
                    e0 = dynamic_pointer_cast<OpExprRef<S>>(e0)->op;
                if (e0->get_type() == OpTypes::Zero)
                    continue;
                auto expr = e0 * (1.0 / dynamic_pointer_cast<OpElement<S, FL>>(names->data[k])->factor);
                if (expr->get_type() != OpTypes::Sum)
                    continue;
                nop.push_back(EhamDump<S>::find_op(a, order, abs_value(names->data[k])));
                tb.push_back((int64_t)top.size());
                for (auto &x : dynamic_pointer_cast<OpSum<S, FL>>(expr)->strings) {
                    top.push_back(EhamDump<S>::find_op(a, order, x->get_op()));
                    tf.push_back(x->factor), tcj.push_back(x->conj != 0);
                }
            }
            tb.push_back((int64_t)top.size());
            ed.af.i64("t.off", off), ed.af.i64("new.op", nop), ed.af.i64("new.term_begin", tb);
            ed.af.i64("term.op", top), ed.af.f64("term.factor", tf), ed.af.i64("term.conj", tcj);
            ed.af.u64("meta", vector<uint64_t>{(uint64_t)(cur_sweep() + 1), (uint64_t)cur_center(), (uint64_t)right, T.tot});
        }
    }
    // intermediates of the NEXT blocking, formed right after a rotation (moving_environment.hpp:415, 643): operator sums
    // TEMP = sum_k factor_k * op_k (or its transpose) of the rotated block (TensorFunctions::intermediates,
    // src/core/tensor_functions.hpp:2404-2459); written in the format of the numerical transform
    void intermediates(const shared_ptr<Symbolic<S>> &names, const shared_ptr<Symbolic<S>> &exprs,
                       const shared_ptr<OperatorTensor<S, FL>> &a, bool left) const override {
        vector<shared_ptr<OpSumProd<S, FL>>> made;
        if (in_chain()) { // which sums the call below creates (same selection as the reference's loop)
            auto seen = a->ops;
            for (size_t i = 0; i < exprs->data.size(); i++) {
                shared_ptr<OpExpr<S>> ei = exprs->data[i];
                if (ei != nullptr && ei->get_type() == OpTypes::ExprRef) // sum-MPO: the rank's local expression
                    ei = dynamic_pointer_cast<OpExprRef<S>>(ei)->op;
                if (ei != nullptr && ei->get_type() == OpTypes::Sum)
                    for (auto &x : dynamic_pointer_cast<OpSum<S, FL>>(ei)->strings)
                        if (x->get_type() == OpTypes::SumProd) {
                            auto ex = dynamic_pointer_cast<OpSumProd<S, FL>>(x);
                            if ((left && ex->b == nullptr) || (!left && ex->a == nullptr) || ex->c == nullptr)
                                continue;
                            if (seen.count(ex->c) != 0)
                                continue;
                            seen[ex->c] = nullptr;
                            made.push_back(ex);
                        }
            }
        }
        Base::intermediates(names, exprs, a, left);
        if (!in_chain() || made.empty())
            return;
        EhamDump<S> ed(spec->next_event(left ? "lint" : "rint", cur_sweep(), cur_center()) + ".entr");
        vector<const double *> ptrs;
        auto order = ed.put_tensor("t", a, false, ptrs);
        Packed2 T;
        T.build(ed.ranges);
        vector<int64_t> off, nop, tb, top, tcj;
        vector<double> tf;
        for (auto p : ptrs) {
            uint64_t o = 0;
            off.push_back(p != nullptr && T.resolve(p, o) ? (int64_t)o : -1);
        }
        for (auto &ex : made) {
            nop.push_back(EhamDump<S>::find_op(a, order, ex->c));
            tb.push_back((int64_t)top.size());
            for (size_t k = 0; k < ex->ops.size(); k++) {
                top.push_back(EhamDump<S>::find_op(a, order, abs_value((shared_ptr<OpExpr<S>>)ex->ops[k])));
                tf.push_back(ex->ops[k]->factor), tcj.push_back(ex->conjs[k] ? 1 : 0);
            }
        }
        tb.push_back((int64_t)top.size());
        ed.af.i64("t.off", off), ed.af.i64("new.op", nop), ed.af.i64("new.term_begin", tb);
        ed.af.i64("term.op", top), ed.af.f64("term.factor", tf), ed.af.i64("term.conj", tcj);
        ed.af.u64("meta", vector<uint64_t>{(uint64_t)(cur_sweep() + 1), (uint64_t)cur_center(), (uint64_t)!left, T.tot});
    }
    struct Packed2 {
        vector<const double *> starts;
        vector<uint64_t> offs, lens;
        uint64_t tot = 0;
        void build(vector<pair<const double *, size_t>> r) {
            sort(r.begin(), r.end());
            for (auto &e : r) {
                if (e.second == 0 || (!starts.empty() && e.first < starts.back() + lens.back()))
                    continue;
                starts.push_back(e.first), offs.push_back(tot), lens.push_back(e.second), tot += e.second;
            }
        }
        bool resolve(const double *p, uint64_t &off) const {
            size_t r = upper_bound(starts.begin(), starts.end(), p) - starts.begin();
            if (r == 0 || p >= starts[r - 1] + lens[r - 1])
                return false;
            off = offs[r - 1] + (uint64_t)(p - starts[r - 1]);
            return true;
        }
        vector<double> gather() const {
            vector<double> d(tot);
            for (size_t r = 0; r < starts.size(); r++)
                memcpy(d.data() + offs[r], starts[r], lens[r] * 8);
            return d;
        }
    };
    // The reference's own TensorFunctions::tensor_product (public) is asked to record the contraction of every
    // operator into a private Auto-mode sequence: batch[1] then holds the k = 1 GEMM groups of
    // AdvancedGEMM::tensor_product (src/core/batch_gemm.hpp:431-505).  Runs of group members with constant pointer
    // steps are written as one 2-D term  C[r][c] += alpha * A[r*a_rs + c*a_cs] * B[0]  (a mechanical re-grouping).
    void capture_blocking(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<OperatorTensor<S, FL>> &b,
                          const shared_ptr<OperatorTensor<S, FL>> &c, const shared_ptr<Symbolic<S>> &cexprs,
                          OpNamesSet delayed, bool right) const {
        const bool chain = in_chain();
        if (chain && !delayed.empty()) {
            cerr << "chain capture needs nodelay=1 (delayed contraction leaves enlarged blocks uncontracted)" << endl;
            abort();
        }
        if (chain && a == nullptr) { // the first block of a chain: c = the site operators (left_assign / right_assign)
            EhamDump<S> ed(spec->next_event(right ? "rasg" : "lasg", cur_sweep(), cur_center()) + ".easg");
            vector<const double *> bp;
            ed.put_tensor("b", b, false, bp);
            Packed2 Sx, V;
            vector<pair<const double *, size_t>> sr, vr;
            for (auto &kv : b->ops)
                if (kv.second->data != nullptr)
                    sr.push_back(make_pair((const double *)kv.second->data, (size_t)kv.second->total_memory));
            vector<uint64_t> ck;
            vector<int64_t> ci, co, bo;
            for (auto &kv : c->ops)
                if (kv.second->data != nullptr)
                    vr.push_back(make_pair((const double *)kv.second->data, (size_t)kv.second->total_memory));
            Sx.build(sr), V.build(vr);
            for (auto p : bp) {
                uint64_t off = 0;
                bo.push_back(p != nullptr && Sx.resolve(p, off) ? (int64_t)off : -1);
            }
            for (auto &kv : c->ops) {
                uint64_t off = 0;
                ck.push_back(op_key<S>(kv.first)), ci.push_back(ed.info_id(kv.second->info, false));
                co.push_back(kv.second->data != nullptr && V.resolve(kv.second->data, off) ? (int64_t)off : -1);
            }
            ed.af.i64("b.off", bo), ed.af.u64("c.key", ck), ed.af.i64("c.info", ci), ed.af.i64("c.off", co);
            ed.af.u64("meta", vector<uint64_t>{(uint64_t)(cur_sweep() + 1), (uint64_t)cur_center(), (uint64_t)right, Sx.tot, V.tot});
            ed.af.f64("site", Sx.gather());
            return;
        }
        if (!chain && (dmrg == nullptr || a == nullptr || !delayed.empty()))
            return;
        if (spec == nullptr)
            return;
        pair<int, int> key(cur_sweep(), cur_center());
        const bool with_data = !chain && spec->blk.count(key), sym = chain || spec->eblk.count(key);
        if (!with_data && !spec->blk_struct.count(key) && !sym)
            return;
        if (!chain) {
            static set<string> done; // first call per (sweep, center, side) only
            stringstream id;
            id << key.first << ":" << key.second << ":" << right;
            if (!done.insert(id.str()).second)
                return;
        }
        auto opf_cap = make_shared<OperatorFunctions<S, FL>>(this->opf->cg);
        opf_cap->seq = make_shared<BatchGEMMSeq<FL>>(0, SeqTypes::Auto);
        auto tf_cap = make_shared<TensorFunctions<S, FL>>(opf_cap);
        shared_ptr<Symbolic<S>> exprs = cexprs != nullptr ? cexprs : (right ? b->rmat * a->rmat : a->lmat * b->lmat);
        const auto &names = right ? c->rmat->data : c->lmat->data;
        assert(exprs->data.size() == names.size());
        set<const void *> seen;
        struct CBlock {
            const double *p;
            int rows, cols;
        };
        vector<CBlock> cblocks;
        vector<pair<const double *, size_t>> xr, sr, vr;
        for (size_t i = 0; i < names.size(); i++) {
            auto cop = dynamic_pointer_cast<OpElement<S, FL>>(names[i]);
            auto op = abs_value(names[i]);
            auto cm = c->ops.at(op);
            if (!seen.insert(cm.get()).second)
                continue;
            auto expr = exprs->data[i] * (1.0 / cop->factor);
            if (chain) // (only the symbolic level is written: nothing to record)
                ;
            else if (right)
                tf_cap->tensor_product(expr, b->ops, a->ops, cm);
            else
                tf_cap->tensor_product(expr, a->ops, b->ops, cm);
            vr.push_back(make_pair((const double *)cm->data, (size_t)cm->total_memory));
            for (int k = 0; k < cm->info->n; k++)
                cblocks.push_back(CBlock{cm->data + cm->info->n_states_total[k], (int)cm->info->n_states_bra[k],
                                         (int)cm->info->n_states_ket[k]});
        }
        for (auto &kv : a->ops)
            if (kv.second->data != nullptr)
                xr.push_back(make_pair((const double *)kv.second->data, (size_t)kv.second->total_memory));
        for (auto &kv : b->ops)
            if (kv.second->data != nullptr)
                sr.push_back(make_pair((const double *)kv.second->data, (size_t)kv.second->total_memory));
        sort(cblocks.begin(), cblocks.end(), [](const CBlock &x, const CBlock &y) { return x.p < y.p; });
        Packed2 X, Sx, V;
        X.build(xr), Sx.build(sr), V.build(vr);
        auto b0 = opf_cap->seq->batch[0], b1 = opf_cap->seq->batch[1];
        stringstream fn;
        if (chain)
            fn << spec->next_event(right ? "rblk" : "lblk", key.first, key.second);
        else
            fn << spec->prefix << ".sw" << key.first << ".c" << key.second << (right ? ".rblk" : ".lblk");
        if (b0->c.size() != 0) {
            cerr << "BLK " << fn.str() << " skipped: two-stage records present" << endl;
            return;
        }
        if (b1->acc_gp.size() != b1->gp.size())
            b1->build_acc_gp();
        vector<b2x_outer_term> terms;
        bool ok = true;
        uint64_t n_members = 0;
        for (size_t g = 0; g < b1->gp.size() && ok && !chain; g++) {
            const size_t kz = b1->acc_gp[g], gc = b1->gp[g];
            ok = b1->n[g] == 1 && b1->k[g] == 1 && b1->beta[g] == 1.0 && b1->ta[g] == CblasNoTrans;
            n_members += gc;
            size_t k = kz;
            while (k < kz + gc && ok) {
                // maximal run with constant steps and one scalar
                size_t e = k + 1;
                ptrdiff_t da = 0, dc = 0;
                if (e < kz + gc && b1->b[e] == b1->b[k]) {
                    da = b1->a[e] - b1->a[k], dc = b1->c[e] - b1->c[k];
                    if (da >= 0 && dc >= (ptrdiff_t)b1->m[g]) {
                        e++;
                        while (e < kz + gc && b1->b[e] == b1->b[k] && b1->a[e] - b1->a[e - 1] == da &&
                               b1->c[e] - b1->c[e - 1] == dc)
                            e++;
                    } else
                        da = dc = 0;
                }
                b2x_outer_term t;
                memset(&t, 0, sizeof(t));
                t.m = (int)(e - k), t.n = b1->m[g], t.a_rs = (int)da, t.a_cs = b1->lda[g];
                t.ldc = t.m > 1 ? (int)dc : t.n;
                t.alpha = b1->alpha[g];
                // operands: the vector comes from the block operators (input) or from the site operators (arena)
                uint64_t off;
                if (X.resolve(b1->a[k], off))
                    t.a_src = 1, t.a_off = off;
                else if (Sx.resolve(b1->a[k], off))
                    t.a_src = 0, t.a_off = off + (1ull << 40); // site arena follows; fixed up below
                else
                    ok = false;
                if (ok && X.resolve(b1->b[k], off))
                    t.b_src = 1, t.b_off = off;
                else if (ok && Sx.resolve(b1->b[k], off))
                    t.b_src = 0, t.b_off = off + (1ull << 40);
                else
                    ok = false;
                if (ok && !V.resolve(b1->c[k], off))
                    ok = false;
                t.c_off = off;
                if (ok && t.m == 1) {
                    // a single contiguous member may span several rows of its block: give it the block's shape
                    auto it = upper_bound(cblocks.begin(), cblocks.end(), b1->c[k],
                                          [](const double *p, const CBlock &x) { return p < x.p; });
                    assert(it != cblocks.begin());
                    --it;
                    const int cols = it->cols;
                    const ptrdiff_t rel = b1->c[k] - it->p;
                    if (cols > 0 && rel % cols == 0 && t.n > cols && t.n % cols == 0) {
                        t.m = t.n / cols, t.n = cols, t.ldc = cols, t.a_rs = cols * t.a_cs;
                    }
                }
                terms.push_back(t);
                k = e;
            }
        }
        if (!ok) { // (operator sums formed in temporaries: no record-level list; the symbolic level below still applies)
            cerr << "BLK " << fn.str() << ": record outside the operator ranges (temporaries), record-level list skipped" << endl;
            opf_cap->seq->clear();
            if (!sym)
                return;
            terms.clear();
        }
        for (auto &t : terms) {
            if (t.a_src == 0)
                t.a_off -= (1ull << 40);
            if (t.b_src == 0)
                t.b_off -= (1ull << 40);
        }
        if (sym) {
            // what TensorFunctions::tensor_product(expr, lop, rop, mat) consumes: per enlarged operator the flattened
            // expression (Prod terms; SumProd terms whose operator sum is an existing intermediate), the operator infos
            // (the enlarged ones with their tensor-product connection info) and the data
            EhamDump<S> ed(fn.str() + ".eblk");
            ArrayFile &af = ed.af;
            af.lite = chain;
            const auto &lt = right ? b : a, &rt = right ? a : b; // (lop, rop) as tensor_product receives them
            vector<const double *> lp, rp;
            auto lorder = ed.put_tensor("lop", lt, false, lp), rorder = ed.put_tensor("rop", rt, false, rp);
            auto offs_in = [&](const vector<const double *> &ptrs, bool is_block) {
                vector<int64_t> o;
                for (auto p : ptrs) {
                    uint64_t off = 0;
                    if (p == nullptr || !(is_block ? X.resolve(p, off) : Sx.resolve(p, off)))
                        o.push_back(-1);
                    else
                        o.push_back((int64_t)off);
                }
                return o;
            };
            af.i64("lop.off", offs_in(lp, !right)), af.i64("rop.off", offs_in(rp, right));
            vector<int64_t> c_info, c_off, t_begin, ty, cj, ta, tb;
            vector<int64_t> tmp_side, tmp_begin{0}, tmp_op, tmp_cj; // temporaries (operator sums with transposed members)
            vector<double> tmp_fac;
            vector<uint64_t> c_key;
            vector<double> tf;
            bool supported = true;
            set<const void *> seen2;
            function<void(const shared_ptr<OpExpr<S>> &)> flat = [&](const shared_ptr<OpExpr<S>> &e) {
                if (e->get_type() == OpTypes::ExprRef) { // sum-MPO: the rank's local expression (ParallelMPO)
                    flat(dynamic_pointer_cast<OpExprRef<S>>(e)->op);
                    return;
                }
                if (e->get_type() == OpTypes::Prod) {
                    auto op = dynamic_pointer_cast<OpProduct<S, FL>>(e);
                    if (op->b == nullptr) {
                        supported = false;
                        return;
                    }
                    ty.push_back(0), cj.push_back(op->conj), tf.push_back(op->factor);
                    ta.push_back(EhamDump<S>::find_op(lt, lorder, op->a)), tb.push_back(EhamDump<S>::find_op(rt, rorder, op->b));
                } else if (e->get_type() == OpTypes::SumProd) {
                    auto op = dynamic_pointer_cast<OpSumProd<S, FL>>(e);
                    const bool inter = op->c != nullptr && ((op->b == nullptr && rt->ops.count(op->c)) ||
                                                            (op->a == nullptr && lt->ops.count(op->c)));
                    bool any_tr = false;
                    for (size_t k = 0; k < op->ops.size(); k++)
                        any_tr = any_tr || op->conjs[k];
                    if (!inter && any_tr) {
                        // no stored intermediate and TRANSPOSED members (the sum-MPO MPOs have them): the product is taken
                        // under ONE sub-label of the connection info, that of (op->conj, ops[0]) — a transposed member's own
                        // product has no entry there — so the temporary is written as what it is: tmp = sum_k factor_k *
                        // op_k (or its transpose), info of ops[0] (tensor_functions.hpp:2236-2261), and ONE product with it.
                        // The temporary is operator number (operators of its side) + t in the term's a / b field.
                        const bool on_r = op->b == nullptr;
                        const int t_idx = (int)tmp_side.size();
                        tmp_side.push_back(on_r ? 0 : 1);
                        for (size_t k = 0; k < op->ops.size(); k++) {
                            auto ok = abs_value((shared_ptr<OpExpr<S>>)op->ops[k]);
                            tmp_op.push_back(on_r ? EhamDump<S>::find_op(rt, rorder, ok) : EhamDump<S>::find_op(lt, lorder, ok));
                            tmp_fac.push_back(op->ops[k]->factor), tmp_cj.push_back(op->conjs[k] ? 1 : 0);
                        }
                        tmp_begin.push_back((int64_t)tmp_op.size());
                        ty.push_back(2), cj.push_back(op->conj), tf.push_back(op->factor);
                        if (on_r)
                            ta.push_back(EhamDump<S>::find_op(lt, lorder, op->a)), tb.push_back((int64_t)rorder.size() + t_idx);
                        else
                            ta.push_back((int64_t)lorder.size() + t_idx), tb.push_back(EhamDump<S>::find_op(rt, rorder, op->b));
                        return;
                    }
                    if (!inter) {
                        // no stored intermediate: the reference sums the operators into a temporary (iadd with the
                        // member's factor and transposition flag, tensor_functions.hpp:2236-2261) and takes ONE product;
                        // by linearity that is the sum of the members' products, which is what is written here
                        for (size_t k = 0; k < op->ops.size(); k++) {
                            auto ok = abs_value((shared_ptr<OpExpr<S>>)op->ops[k]);
                            ty.push_back(0), tf.push_back(op->factor * op->ops[k]->factor);
                            if (op->b == nullptr) {
                                cj.push_back(op->conj ^ (op->conjs[k] ? 2 : 0));
                                ta.push_back(EhamDump<S>::find_op(lt, lorder, op->a)), tb.push_back(EhamDump<S>::find_op(rt, rorder, ok));
                            } else {
                                cj.push_back(op->conj ^ (op->conjs[k] ? 1 : 0));
                                ta.push_back(EhamDump<S>::find_op(lt, lorder, ok)), tb.push_back(EhamDump<S>::find_op(rt, rorder, op->b));
                            }
                        }
                        return;
                    }
                    ty.push_back(1), cj.push_back(op->conj), tf.push_back(op->factor);
                    if (op->b == nullptr)
                        ta.push_back(EhamDump<S>::find_op(lt, lorder, op->a)), tb.push_back(EhamDump<S>::find_op(rt, rorder, op->c));
                    else
                        ta.push_back(EhamDump<S>::find_op(lt, lorder, op->c)), tb.push_back(EhamDump<S>::find_op(rt, rorder, op->b));
                } else if (e->get_type() == OpTypes::Sum) {
                    for (auto &x : dynamic_pointer_cast<OpSum<S, FL>>(e)->strings)
                        flat(x);
                } else if (e->get_type() != OpTypes::Zero)
                    supported = false;
            };
            for (size_t i = 0; i < names.size(); i++) {
                auto cop = dynamic_pointer_cast<OpElement<S, FL>>(names[i]);
                auto cm = c->ops.at(abs_value(names[i]));
                if (!seen2.insert(cm.get()).second)
                    continue;
                uint64_t off = 0;
                if (cm->total_memory)
                    V.resolve(cm->data, off);
                c_info.push_back(ed.info_id(cm->info, true)), c_off.push_back((int64_t)off);
                c_key.push_back(op_key<S>(c->ops.find(abs_value(names[i]))->first));
                t_begin.push_back((int64_t)ty.size());
                flat(exprs->data[i] * (1.0 / cop->factor));
            }
            t_begin.push_back((int64_t)ty.size());
            af.i64("c.info", c_info), af.i64("c.off", c_off), af.i64("c.term_begin", t_begin), af.u64("c.key", c_key);
            af.i64("term.type", ty), af.i64("term.conj", cj), af.f64("term.factor", tf), af.i64("term.a", ta), af.i64("term.b", tb);
            if (!tmp_side.empty()) // (only sum-MPO runs have them: serial fixtures keep their format)
                af.i64("tmp.side", tmp_side), af.i64("tmp.begin", tmp_begin), af.i64("tmp.op", tmp_op), af.f64("tmp.factor", tmp_fac),
                    af.i64("tmp.conj", tmp_cj);
            af.u64("meta", vector<uint64_t>{(uint64_t)key.first, (uint64_t)key.second, (uint64_t)right, (uint64_t)supported,
                                            (uint64_t)terms.size(), X.tot, Sx.tot, V.tot});
            af.bulk("x", X.gather()), af.f64("site", Sx.gather()), af.bulk("v_ref", V.gather());
            cerr << "EBLK " << fn.str() << ".eblk ops=" << c_info.size() << " terms=" << ty.size()
                 << " supported=" << supported << endl;
        }
        if (ok && (with_data || spec->blk_struct.count(key))) {
            ArrayFile af(fn.str() + ".blk");
            af.put("terms", 4, 1, terms.data(), terms.size() * sizeof(b2x_outer_term));
            af.u64("lens", vector<uint64_t>{(uint64_t)terms.size(), Sx.tot, X.tot, V.tot, (uint64_t)right, n_members,
                                            (uint64_t)b1->gp.size(), (uint64_t)seen.size()});
            if (with_data)
                af.f64("arena", Sx.gather()), af.f64("in", X.gather()), af.f64("out_ref", V.gather());
        }
        stringstream ss;
        ss << "BLK " << fn.str() << ".blk terms=" << terms.size() << " groups=" << b1->gp.size() << " members=" << n_members
           << " ops=" << seen.size() << " in=" << X.tot << " site=" << Sx.tot << " out=" << V.tot;
        if (log)
            log->push_back(ss.str());
        cerr << ss.str() << endl;
        opf_cap->seq->clear();
    }
    void maybe_capture(const shared_ptr<OperatorTensor<S, FL>> &a, const shared_ptr<SparseMatrix<S, FL>> &bra,
                       const shared_ptr<SparseMatrix<S, FL>> &ket, const shared_ptr<OperatorTensor<S, FL>> &c,
                       bool right) const {
        const bool chain = in_chain();
        if (spec == nullptr || (!chain && dmrg == nullptr))
            return;
        pair<int, int> key(cur_sweep(), cur_center());
        bool wd = !chain && spec->rot.count(key), st = !chain && spec->rot_struct.count(key), er = chain || spec->erot.count(key);
        if (!wd && !st && !er)
            return;
        // record with the reference's tensor_rotate into a private Auto-mode sequence (nothing is executed)
        auto opf_cap = make_shared<OperatorFunctions<S, FL>>(this->opf->cg);
        opf_cap->seq = make_shared<BatchGEMMSeq<FL>>(0, SeqTypes::Auto);
        const auto &names = right ? a->rmat->data : a->lmat->data;
        vector<pair<const double *, size_t>> xr, vr, opr;
        for (size_t i = 0; i < names.size(); i++)
            if (names[i]->get_type() != OpTypes::Zero) {
                auto pa = abs_value(names[i]);
                auto am = a->ops.at(pa), cm = c->ops.at(pa);
                opf_cap->tensor_rotate(am, cm, bra, ket, right);
                xr.push_back(make_pair((const double *)am->data, (size_t)am->total_memory));
                vr.push_back(make_pair((const double *)cm->data, (size_t)cm->total_memory));
            }
        opr.push_back(make_pair((const double *)bra->data, (size_t)bra->total_memory));
        opr.push_back(make_pair((const double *)ket->data, (size_t)ket->total_memory));
        auto b0 = opf_cap->seq->batch[0], b1 = opf_cap->seq->batch[1];
        size_t n = b0->c.size();
        assert(b1->c.size() == n && b0->acidxs.size() == 0);
        // sorted, de-duplicated ranges -> packed offsets
        struct Packed {
            vector<const double *> starts;
            vector<uint64_t> offs, lens;
            uint64_t tot = 0;
            void build(vector<pair<const double *, size_t>> r) {
                sort(r.begin(), r.end());
                r.erase(unique(r.begin(), r.end()), r.end());
                for (auto &e : r) {
                    if (!starts.empty() && e.first < starts.back() + lens.back())
                        continue; // (identical or nested range)
                    starts.push_back(e.first), offs.push_back(tot), lens.push_back(e.second), tot += e.second;
                }
            }
            uint64_t resolve(const double *p) const {
                size_t r = upper_bound(starts.begin(), starts.end(), p) - starts.begin() - 1;
                assert(p >= starts[r] && p < starts[r] + lens[r]);
                return offs[r] + (uint64_t)(p - starts[r]);
            }
            vector<double> gather() const {
                vector<double> d(tot);
                for (size_t r = 0; r < starts.size(); r++)
                    memcpy(d.data() + offs[r], starts[r], lens[r] * 8);
                return d;
            }
        } X, V, A;
        X.build(xr), V.build(vr), A.build(opr);
        vector<b2x_pair> pairs(n);
        uint64_t macs = 0;
        for (size_t i = 0; i < n; i++) {
            b2x_pair &p = pairs[i];
            memset(&p, 0, sizeof(p));
            p.m0 = b0->m[i], p.n0 = b0->n[i], p.k0 = b0->k[i], p.lda0 = b0->lda[i], p.ldb0 = b0->ldb[i];
            p.m1 = b1->m[i], p.n1 = b1->n[i], p.k1 = b1->k[i], p.lda1 = b1->lda[i], p.ldc1 = b1->ldc[i];
            p.ta0 = b0->ta[i] != CblasNoTrans, p.tb0 = (b0->tb[i] == CblasTrans || b0->tb[i] == CblasConjTrans);
            p.ta1 = (b1->ta[i] == CblasTrans || b1->ta[i] == CblasConjTrans), p.tb1 = b1->tb[i] != CblasNoTrans;
            p.alpha0 = b0->alpha[i], p.alpha1 = b1->alpha[i];
            assert(b0->beta[i] == 0.0 && b1->beta[i] == 1.0 && !p.ta0 && !p.tb1);
            p.x_off = X.resolve(b0->a[i]), p.y_off = A.resolve(b0->b[i]);
            p.z_off = A.resolve(b1->a[i]), p.v_off = V.resolve(b1->c[i]);
            macs += (uint64_t)p.m0 * p.n0 * p.k0 + (uint64_t)p.m1 * p.n1 * p.k1;
        }
        if (er) { // symbolic-level fixture: what OperatorFunctions::tensor_rotate consumes (infos) and produces
            stringstream efn;
            if (chain)
                efn << spec->next_event(right ? "rrot" : "lrot", key.first, key.second) << ".erot";
            else
                efn << spec->prefix << ".sw" << key.first << ".c" << key.second << (right ? ".rrot" : ".lrot") << ".erot";
            EhamDump<S> ed(efn.str());
            ArrayFile &af = ed.af;
            af.lite = chain;
            vector<int64_t> ai, ci, ao, co;
            vector<uint64_t> akey, ckey;
            vector<double> afac;
            for (size_t i = 0; i < names.size(); i++)
                if (names[i]->get_type() != OpTypes::Zero) {
                    auto pa = abs_value(names[i]);
                    auto am = a->ops.at(pa), cm = c->ops.at(pa);
                    ai.push_back(ed.info_id(am->info, false)), ci.push_back(ed.info_id(cm->info, false));
                    ao.push_back(am->total_memory ? (int64_t)X.resolve(am->data) : 0);
                    co.push_back(cm->total_memory ? (int64_t)V.resolve(cm->data) : 0);
                    afac.push_back(am->factor);
                    // (each tensor's own key objects: equal operators may print differently in different tensors)
                    akey.push_back(op_key<S>(a->ops.find(pa)->first)), ckey.push_back(op_key<S>(c->ops.find(pa)->first));
                }
            af.i64("a.info", ai), af.i64("c.info", ci), af.i64("a.off", ao), af.i64("c.off", co), af.f64("a.factor", afac);
            af.u64("a.key", akey), af.u64("c.key", ckey);
            if (chain) { // the whole rotated tensor as the next step will see it (operators the rotation does not write included)
                vector<const double *> cp;
                ed.put_tensor("ct", c, false, cp);
            }
            af.i64("mps.info", vector<int64_t>{ed.info_id(bra->info, false), ed.info_id(ket->info, false)});
            af.i64("mps.off", vector<int64_t>{(int64_t)A.resolve(bra->data), (int64_t)A.resolve(ket->data)});
            af.f64("mps.factor", vector<double>{bra->factor, ket->factor});
            af.u64("meta", vector<uint64_t>{(uint64_t)key.first, (uint64_t)key.second, (uint64_t)right, (uint64_t)n, macs,
                                            X.tot, V.tot, A.tot});
            // (the MPS tensor is data of the starting state while the initial environments are built; later the replay
            //  rotates with the tensors it obtained from its own decomposition)
            af.bulk("x", X.gather());
            if (chain && dmrg != nullptr)
                af.bulk("arena", A.gather());
            else
                af.f64("arena", A.gather());
            af.bulk("v_ref", V.gather());
            cerr << "EROT " << efn.str() << " ops=" << ai.size() << endl;
        }
        if (!wd && !st)
            return;
        b2x_planfile pf;
        memset(&pf, 0, sizeof(pf));
        pf.n_pairs = n, pf.psi_len = X.tot, pf.sigma_len = V.tot, pf.arena_len = A.tot;
        pf.max_work = opf_cap->seq->max_work;
        vector<uint64_t> ranges;
        for (size_t r = 0; r < A.starts.size(); r++)
            ranges.push_back(A.offs[r]), ranges.push_back(A.lens[r]);
        pf.n_ranges = A.starts.size(), pf.ranges = ranges.data(), pf.pairs = pairs.data();
        double meta[8] = {(double)key.first, (double)key.second, 0.0, (double)right, 0.0, (double)macs,
                          (double)dmrg->me->n_sites, (double)names.size()};
        pf.meta = meta, pf.n_meta = 8;
        vector<double> arena, psi, sigma;
        if (wd) {
            arena = A.gather(), psi = X.gather(), sigma = V.gather();
            pf.arena = arena.data(), pf.psi = psi.data(), pf.sigma_ref = sigma.data();
            pf.flags = B2XPF_ARENA | B2XPF_PSI | B2XPF_SIGMA;
        }
        stringstream fn;
        fn << spec->prefix << ".sw" << key.first << ".c" << key.second << (right ? ".rrot" : ".lrot") << (wd ? ".plan" : ".struct");
        b2x_planfile_write(fn.str().c_str(), &pf);
        stringstream ss;
        ss << "ROT " << fn.str() << " pairs=" << n << " ops=" << xr.size() << " x=" << X.tot << " v=" << V.tot
           << " mps=" << A.tot << " macs=" << macs;
        if (log)
            log->push_back(ss.str());
        cerr << ss.str() << endl;
        opf_cap->seq->clear();
    }
};
template <typename S> using RotTF = RotTFT<S>;
template <typename S, typename R>
static void setup_rtf(const shared_ptr<R> &rtf, MovingEnvironment<S, double, double> *me, DMRG<S, double, double> *dm,
                      const DumpSpec *spec, vector<string> *log, const shared_ptr<MPO<S, double>> &mpo,
                      std::function<void(DMRG<S, double, double> *)> &set_dmrg) {
    rtf->me = me, rtf->dmrg = dm, rtf->spec = spec, rtf->log = log;
    mpo->tf = rtf; // MovingEnvironment rotates through mpo->tf (src/dmrg/moving_environment.hpp:360)
    set_dmrg = [rtf](DMRG<S, double, double> *d) { rtf->dmrg = d; };
}


template <typename S> struct Dumper : CallbackKernel {
    typedef double FL;
    DMRG<S, FL, FL> *dmrg = nullptr;
    DumpSpec spec;
    bool start_forward = true; // DMRG::forward is only written when solve() returns: sweep isw runs forward iff
                               // (isw even) == start_forward (src/dmrg/sweep_algorithm.hpp:3076-3101)
    mutable vector<string> log;
    pair<int, int> stop_after = make_pair(-1, -1);
    bool spectra = false;
    mutable Timer sweep_timer;
    mutable map<pair<int, int>, string> pending; // file awaiting psi_out / energy
    void compute(const string &name, int iprint) const override {
        if (dmrg == nullptr)
            return;
        int isw = dmrg->isweep, site = dmrg->me->center;
        pair<int, int> key(isw, site);
        if (name == "DMRG::sweep::iter.eff_ham") {
            bool wd = spec.with_data.count(key), st = spec.structure.count(key);
            if (wd || st)
                capture(isw, site, wd);
            if (spec.eham.count(key))
                capture_eham(isw, site);
            if (spec.lite() && isw <= spec.chain) { // chain mode: infos, tensors (keys), expression; no bulk data
                EhamDump<S> ed(spec.next_event("eham", isw, site) + ".eham");
                ed.af.lite = true;
                write_eham_common(ed, dmrg->current_eff_ham);
                ed.af.u64("chain.meta", vector<uint64_t>{(uint64_t)isw, (uint64_t)site, (uint64_t)dmrg->me->n_sites,
                                                         (uint64_t)((isw % 2 == 0) == start_forward)});
            }
            if (stop_after == key) { // structure captures at large M: the rest of the sweep is not needed
                ofstream lf((spec.prefix + ".log").c_str());
                for (auto &l : log)
                    lf << l << endl;
                lf << "STOPPED_AFTER " << isw << " " << site << endl;
                lf.close();
                cout << "STOPPED_AFTER " << isw << " " << site << endl;
                cout.flush();
                _exit(0); // the run is abandoned on purpose (no teardown)
            }
        } else if (name == "DMRG::sweep::iter.eff_ham.end") {
            bool wd = spec.pnoise.count(key), st = spec.pnoise_struct.count(key), en = spec.enoise.count(key);
            if (wd || st || en)
                capture_pnoise(isw, site, wd, st, en);
            // chain mode, noisy sweep: the perturbative-noise step of this site as an event of its own (symbolic level only)
            if (spec.lite() && isw <= spec.chain && (dmrg->noise_type & NoiseTypes::Perturbative) &&
                isw < (int)dmrg->noises.size() && dmrg->noises[isw] != 0)
                capture_pnoise(isw, site, false, false, true, spec.next_event("enoise", isw, site) + ".enoise", true);
        } else if (name == "DMRG::sweep::iter.end") {
            stringstream ss;
            ss.precision(15);
            ss << "SITE_ENERGY " << isw << " " << site << " " << dmrg->sweep_energies.back()[0];
            log.push_back(ss.str());
            if (spectra) { // the spectrum split_density_matrix truncated at this site (all eigenvalues, sector by sector)
                stringstream sp;
                sp.precision(17);
                sp << "SPECTRA " << isw << " " << site << " " << dmrg->sweep_discarded_weights.back() << " "
                   << dmrg->wfn_spectra.size();
                for (auto x : dmrg->wfn_spectra)
                    sp << " " << x;
                log.push_back(sp.str());
            }
        } else if (name == "DMRG::sweep.start") {
            sweep_timer.get_time();
        } else if (name == "DMRG::sweep.end") {
            stringstream ss;
            ss.precision(9);
            ss << "SWEEP_TIME " << isw << " " << sweep_timer.get_time() << " " << dmrg->teff << " " << dmrg->teig << " "
               << dmrg->tprt << " " << dmrg->tblk << " " << dmrg->tmve << " " << dmrg->tdm << " " << dmrg->tsplt << " "
               << dmrg->tsvd << " " << dmrg->me->trot << " " << dmrg->me->tctr << " " << (double)dmrg->sweep_cumulative_nflop;
            log.push_back(ss.str());
        }
    }

    struct EhamOrders {
        vector<shared_ptr<OpExpr<S>>> l, r, dl, dr;
        uint64_t arena_len = 0;
        size_t n_terms = 0;
    };
    // everything the symbolic -> numeric layer of one EffectiveHamiltonian consumes (infos, tensors, arena, expression)
    EhamOrders write_eham_common(EhamDump<S> &ed, const shared_ptr<EffectiveHamiltonian<S, FL>> &h) const {
        EhamOrders eo;
        ArrayFile &af = ed.af;
        S cdq = h->ket->info->delta_quantum, vdq = h->bra->info->delta_quantum;
        af.u64("labels", vector<uint64_t>{cdq.data, vdq.data, h->opdq.data, h->hop_left_vacuum.data});
        af.f64("const_e", vector<double>{(double)dmrg->me->mpo->const_e});
        // operator infos
        vector<uint64_t> ll, li, rl, ri;
        for (auto &p : h->left_op_infos)
            ll.push_back(p.first.data), li.push_back((uint64_t)ed.info_id(p.second, false));
        for (auto &p : h->right_op_infos)
            rl.push_back(p.first.data), ri.push_back((uint64_t)ed.info_id(p.second, false));
        af.u64("linfos.label", ll), af.u64("linfos.info", li), af.u64("rinfos.label", rl), af.u64("rinfos.info", ri);
        af.u64("ket.info", vector<uint64_t>{(uint64_t)ed.info_id(h->ket->info, false)});
        af.u64("bra.info", vector<uint64_t>{(uint64_t)ed.info_id(h->bra->info, false)});
        // the reference's own connection info of the wavefunction (what initialize_wfn must reproduce)
        ed.put_cinfo("wfn_cinfo", h->wfn_infos[0]);
        // operator tensors
        auto lopt = h->op->lopt, ropt = h->op->ropt;
        const bool dl = lopt->get_type() == OperatorTensorTypes::Delayed,
                   dr = ropt->get_type() == OperatorTensorTypes::Delayed;
        af.u64("tensor.delayed", vector<uint64_t>{(uint64_t)dl, (uint64_t)dr});
        vector<const double *> lp, rp, dlp, drp;
        auto &lorder = eo.l, &rorder = eo.r, &dlorder = eo.dl, &drorder = eo.dr;
        lorder = ed.put_tensor("lopt", lopt, true, lp);
        rorder = ed.put_tensor("ropt", ropt, true, rp);
        shared_ptr<DelayedOperatorTensor<S, FL>> dopt = nullptr;
        if (dl || dr) {
            dopt = dynamic_pointer_cast<DelayedOperatorTensor<S, FL>>(dl ? lopt : ropt);
            dlorder = ed.put_tensor("dopt.l", dopt->lopt, false, dlp);
            drorder = ed.put_tensor("dopt.r", dopt->ropt, false, drp);
        }
        // arena of all operator data ranges
        auto rg = ed.ranges;
        sort(rg.begin(), rg.end());
        rg.erase(unique(rg.begin(), rg.end()), rg.end());
        vector<const double *> starts;
        vector<uint64_t> offs;
        uint64_t tot = 0;
        for (auto &r : rg)
            starts.push_back(r.first), offs.push_back(tot), tot += r.second;
        vector<double> arena(tot);
        for (size_t i = 0; i < rg.size(); i++)
            memcpy(arena.data() + offs[i], rg[i].first, rg[i].second * 8);
        af.bulk("arena", arena);
        auto offs_of = [&](const vector<const double *> &ptrs) {
            vector<int64_t> o;
            for (auto p : ptrs) {
                if (p == nullptr) {
                    o.push_back(-1);
                    continue;
                }
                size_t k = lower_bound(starts.begin(), starts.end(), p) - starts.begin();
                o.push_back((int64_t)offs[k]);
            }
            return o;
        };
        af.i64("lopt.off", offs_of(lp)), af.i64("ropt.off", offs_of(rp));
        af.i64("dopt.l.off", offs_of(dlp)), af.i64("dopt.r.off", offs_of(drp));
        // expression of H_eff: a sum of Prod / SumProd terms
        vector<int64_t> ty, cj, ia, ib, d0, d1, dcj;
        vector<double> fac;
        auto add_term = [&](const shared_ptr<OpExpr<S>> &e) {
            if (e->get_type() == OpTypes::SumProd) {
                auto op = dynamic_pointer_cast<OpSumProd<S, FL>>(e);
                ty.push_back(1), cj.push_back(op->conj), fac.push_back(op->factor);
                ia.push_back(EhamDump<S>::find_op(lopt, lorder, op->a)), ib.push_back(EhamDump<S>::find_op(ropt, rorder, op->b));
                d0.push_back(EhamDump<S>::find_op(dopt->lopt, dlorder, op->ops[0]));
                d1.push_back(EhamDump<S>::find_op(dopt->ropt, drorder, op->ops[1]));
                dcj.push_back((int64_t)op->conjs[0] | ((int64_t)op->conjs[1] << 1));
            } else if (e->get_type() == OpTypes::Prod) {
                auto op = dynamic_pointer_cast<OpProduct<S, FL>>(e);
                ty.push_back(0), cj.push_back(op->conj), fac.push_back(op->factor);
                ia.push_back(EhamDump<S>::find_op(lopt, lorder, op->a)), ib.push_back(EhamDump<S>::find_op(ropt, rorder, op->b));
                d0.push_back(-1), d1.push_back(-1), dcj.push_back(0);
            } else
                assert(false);
        };
        auto ex = h->op->mat->data[0];
        if (ex->get_type() == OpTypes::ExprRef) // sum-MPO: the rank's local expression (ParallelMPO)
            ex = dynamic_pointer_cast<OpExprRef<S>>(ex)->op;
        if (ex->get_type() == OpTypes::Sum)
            for (auto &t : dynamic_pointer_cast<OpSum<S, FL>>(ex)->strings)
                add_term(t);
        else
            add_term(ex);
        af.i64("expr.type", ty), af.i64("expr.conj", cj), af.f64("expr.factor", fac), af.i64("expr.a", ia);
        af.i64("expr.b", ib), af.i64("expr.d0", d0), af.i64("expr.d1", d1), af.i64("expr.dconj", dcj);
        eo.arena_len = tot, eo.n_terms = ty.size();
        return eo;
    }
    void capture_eham(int isw, int site) const {
        auto h = dmrg->current_eff_ham;
        stringstream fn;
        fn << spec.prefix << ".sw" << isw << ".site" << site << ".eham";
        EhamDump<S> ed(fn.str());
        ArrayFile &af = ed.af;
        EhamOrders eo = write_eham_common(ed, h);
        uint64_t tot = eo.arena_len;
        struct { size_t n; size_t size() const { return n; } } ty{eo.n_terms};
        // data: psi, diag, reference sigma
        size_t n = h->ket->total_memory;
        af.f64("psi", h->ket->data, n), af.f64("diag", h->diag->data, n);
        vector<double> sigma(h->bra->total_memory, 0.0);
        h->precompute();
        (*h->tf)(GMatrix<double>(h->ket->data, (MKL_INT)n, 1), GMatrix<double>(sigma.data(), (MKL_INT)sigma.size(), 1), 1.0);
        af.u64("n_pairs", vector<uint64_t>{(uint64_t)h->tf->opf->seq->batch[0]->c.size(),
                                           (uint64_t)(h->tf->opf->seq->batch[0]->nflop + h->tf->opf->seq->batch[1]->nflop)});
        h->post_precompute();
        af.f64("sigma_ref", sigma);
        cerr << "EHAM " << fn.str() << " terms=" << ty.size() << " arena=" << tot << " psi=" << n << endl;
    }
    // perturbative noise: run the reference's own EffectiveHamiltonian::perturbative_noise with the capturing
    // TensorFunctions swapped in, exactly as DMRG::update_two_dot calls it (src/dmrg/sweep_algorithm.hpp:799-802)
    void capture_pnoise(int isw, int site, bool with_data, bool structure, bool symbolic, const string &sym_file = "",
                        bool lite = false) const {
        auto h = dmrg->current_eff_ham;
        auto cap = make_shared<CapTF<S>>(h->tf->opf);
        auto old_tf = h->tf;
        h->tf = cap;
        const bool forward = (isw % 2 == 0) == start_forward;
        // (under MPI every rank is here at the same site: the rule makes the perturbed labels the union over the ranks, as in
        // the real step, so that all ranks record the same layout of the perturbed wavefunctions)
        auto pket = h->perturbative_noise(forward, site, site + 1, FuseTypes::FuseLR, dmrg->me->ket->info,
                                          dmrg->noise_type, dmrg->me->para_rule);
        h->tf = old_tf;
        size_t n = cap->recs.size();
        const double *ket0 = h->ket->data, *ket1 = ket0 + h->ket->total_memory;
        vector<pair<const double *, size_t>> ext;
        for (size_t i = 0; i < n; i++) {
            b2x_gemm &g = cap->recs[i];
            size_t ea = g.ta ? (size_t)(g.k - 1) * g.lda + g.m : (size_t)(g.m - 1) * g.lda + g.k;
            size_t eb = g.tb ? (size_t)(g.n - 1) * g.ldb + g.k : (size_t)(g.k - 1) * g.ldb + g.n;
            g.a_src = cap->pa[i] >= ket0 && cap->pa[i] < ket1, g.b_src = cap->pb[i] >= ket0 && cap->pb[i] < ket1;
            if (!g.a_src)
                ext.push_back(make_pair(cap->pa[i], ea));
            if (!g.b_src)
                ext.push_back(make_pair(cap->pb[i], eb));
            assert(cap->pc[i] >= cap->out_base && cap->pc[i] < cap->out_base + cap->out_len);
            g.c_off = (uint64_t)(cap->pc[i] - cap->out_base);
        }
        sort(ext.begin(), ext.end());
        vector<pair<const double *, size_t>> rg;
        for (auto &e : ext) {
            if (!rg.empty() && e.first <= rg.back().first + rg.back().second)
                rg.back().second = max(rg.back().second, (size_t)(e.first - rg.back().first) + e.second);
            else
                rg.push_back(e);
        }
        vector<const double *> starts(rg.size());
        vector<uint64_t> offs(rg.size());
        uint64_t tot = 0;
        for (size_t r = 0; r < rg.size(); r++)
            starts[r] = rg[r].first, offs[r] = tot, tot += rg[r].second;
        auto resolve = [&](const double *p) -> uint64_t {
            size_t r = upper_bound(starts.begin(), starts.end(), p) - starts.begin() - 1;
            return offs[r] + (uint64_t)(p - starts[r]);
        };
        uint64_t macs = 0;
        for (size_t i = 0; i < n; i++) {
            b2x_gemm &g = cap->recs[i];
            g.a_off = g.a_src ? (uint64_t)(cap->pa[i] - ket0) : resolve(cap->pa[i]);
            g.b_off = g.b_src ? (uint64_t)(cap->pb[i] - ket0) : resolve(cap->pb[i]);
            macs += (uint64_t)g.m * g.n * g.k;
        }
        if (symbolic) {
            stringstream efn;
            if (sym_file.empty())
                efn << spec.prefix << ".sw" << isw << ".site" << site << ".enoise";
            else
                efn << sym_file;
            EhamDump<S> ed(efn.str());
            ArrayFile &af = ed.af;
            af.lite = lite;
            EhamOrders eo = write_eham_common(ed, h);
            if (lite)
                af.u64("chain.meta", vector<uint64_t>{(uint64_t)isw, (uint64_t)site, (uint64_t)dmrg->me->n_sites,
                                                      (uint64_t)forward});
            auto lopt = h->op->lopt, ropt = h->op->ropt;
            shared_ptr<OpExpr<S>> i_op = make_shared<OpElement<S, FL>>(OpNames::I, SiteIndex(), S());
            vector<int64_t> iop{EhamDump<S>::find_op(lopt, eo.l, i_op), EhamDump<S>::find_op(ropt, eo.r, i_op), -1, -1};
            if (lopt->get_type() == OperatorTensorTypes::Delayed || ropt->get_type() == OperatorTensorTypes::Delayed) {
                auto dopt = dynamic_pointer_cast<DelayedOperatorTensor<S, FL>>(
                    lopt->get_type() == OperatorTensorTypes::Delayed ? lopt : ropt);
                iop[2] = EhamDump<S>::find_op(dopt->lopt, eo.dl, i_op), iop[3] = EhamDump<S>::find_op(dopt->ropt, eo.dr, i_op);
            }
            af.i64("noise.iop", iop);
            vector<int64_t> pc;
            vector<uint64_t> pl, vd, vi, vo;
            for (auto &x : cap->a_psubsl)
                pc.push_back(x.first), pl.push_back(x.second.data);
            for (auto &x : cap->a_vdqs)
                vd.push_back(x.data);
            for (int j = 0; j < cap->a_vmats->n; j++)
                vi.push_back((uint64_t)ed.info_id((*cap->a_vmats)[j]->info, false)), vo.push_back(cap->a_vmats->offsets[j]);
            af.i64("noise.psubsl.conj", pc), af.u64("noise.psubsl.label", pl), af.u64("noise.vdqs", vd);
            af.u64("noise.vinfo", vi), af.u64("noise.voff", vo);
            af.i64("noise.args", vector<int64_t>{(int64_t)cap->a_trace_right, (int64_t)cap->a_vidx, (int64_t)cap->a_tvidx,
                                                 (int64_t)n, (int64_t)macs, (int64_t)cap->out_len});
            af.u64("noise.vacuum", vector<uint64_t>{dmrg->me->ket->info->vacuum.data});
            af.f64("noise.value", vector<double>{isw < (int)dmrg->noises.size() ? (double)dmrg->noises[isw] : 0.0});
            af.bulk("psi", h->ket->data, h->ket->total_memory);
            af.bulk("out_ref", pket->data, pket->total_memory);
            if (!lite)
                cerr << "ENOISE " << efn.str() << " gemms=" << n << endl;
        }
        if (with_data || structure) {
        stringstream fn;
        fn << spec.prefix << ".sw" << isw << ".site" << site << (with_data ? ".pnoise" : ".pnoise_struct");
        {
            ArrayFile af(fn.str());
            af.put("gemms", 4, 1, cap->recs.data(), n * sizeof(b2x_gemm));
            af.u64("lens", vector<uint64_t>{(uint64_t)n, tot, (uint64_t)h->ket->total_memory, (uint64_t)cap->out_len,
                                            (uint64_t)pket->n, macs, (uint64_t)forward});
            vector<uint64_t> po(pket->n), pl(pket->n);
            for (int j = 0; j < pket->n; j++)
                po[j] = pket->offsets[j], pl[j] = (*pket)[j]->total_memory;
            af.u64("out.offsets", po), af.u64("out.lens", pl);
            if (with_data) {
                vector<double> arena(tot);
                for (size_t r = 0; r < rg.size(); r++)
                    memcpy(arena.data() + offs[r], rg[r].first, rg[r].second * 8);
                af.f64("arena", arena);
                af.f64("in", h->ket->data, h->ket->total_memory);
                af.f64("out_ref", pket->data, pket->total_memory);
            }
        }
        stringstream ss;
        ss << "PNOISE " << fn.str() << " gemms=" << n << " in=" << h->ket->total_memory << " out=" << cap->out_len
           << " kets=" << pket->n << " arena=" << tot << " macs=" << macs;
        log.push_back(ss.str());
        cerr << ss.str() << endl;
        }
        pket->deallocate_infos();
        pket->deallocate();
    }
    void capture(int isw, int site, bool with_data) const {
        auto h = dmrg->current_eff_ham;
        h->precompute();
        auto seq = h->tf->opf->seq;
        auto b0 = seq->batch[0], b1 = seq->batch[1];
        size_t n = b0->c.size();
        assert(b1->c.size() == n && b0->acidxs.size() == 0);
        b2x_planfile pf;
        memset(&pf, 0, sizeof(pf));
        pf.n_pairs = n;
        pf.psi_len = h->ket->total_memory;
        pf.sigma_len = h->bra->total_memory;
        pf.max_work = seq->max_work;
        vector<b2x_pair> pairs(n);
        // operator pointer extents -> merged ranges
        vector<pair<const double *, size_t>> ext;
        ext.reserve(2 * n);
        for (size_t i = 0; i < n; i++) {
            b2x_pair &p = pairs[i];
            memset(&p, 0, sizeof(p));
            p.m0 = b0->m[i], p.n0 = b0->n[i], p.k0 = b0->k[i];
            p.lda0 = b0->lda[i], p.ldb0 = b0->ldb[i];
            p.m1 = b1->m[i], p.n1 = b1->n[i], p.k1 = b1->k[i];
            p.lda1 = b1->lda[i], p.ldc1 = b1->ldc[i];
            p.ta0 = b0->ta[i] != CblasNoTrans, p.tb0 = (b0->tb[i] == CblasTrans || b0->tb[i] == CblasConjTrans);
            p.ta1 = (b1->ta[i] == CblasTrans || b1->ta[i] == CblasConjTrans), p.tb1 = b1->tb[i] != CblasNoTrans;
            p.alpha0 = b0->alpha[i], p.alpha1 = b1->alpha[i];
            assert(b0->beta[i] == 0.0 && b1->beta[i] == 1.0);
            assert(b0->ldc[i] == p.n0 && b1->ldb[i] == p.n0 && p.k1 == p.m0 && p.n1 == p.n0);
            assert(!p.ta0 && !p.tb1);
            p.x_off = (uint64_t)(b0->a[i] - (const double *)0);
            p.v_off = (uint64_t)(b1->c[i] - (double *)0);
            size_t ey = p.tb0 ? (size_t)(p.n0 - 1) * p.ldb0 + p.k0 : (size_t)(p.k0 - 1) * p.ldb0 + p.n0;
            size_t ez = p.ta1 ? (size_t)(p.k1 - 1) * p.lda1 + p.m1 : (size_t)(p.m1 - 1) * p.lda1 + p.k1;
            ext.push_back(make_pair(b0->b[i], ey));
            ext.push_back(make_pair(b1->a[i], ez));
        }
        vector<pair<const double *, size_t>> srt = ext;
        sort(srt.begin(), srt.end());
        vector<pair<const double *, size_t>> rg; // (start, len)
        for (auto &e : srt) {
            if (!rg.empty() && e.first <= rg.back().first + rg.back().second) {
                size_t end = max(rg.back().second, (size_t)(e.first - rg.back().first) + e.second);
                rg.back().second = end;
            } else
                rg.push_back(e);
        }
        vector<uint64_t> ranges(rg.size() * 2);
        vector<const double *> starts(rg.size());
        uint64_t tot = 0;
        for (size_t r = 0; r < rg.size(); r++) {
            starts[r] = rg[r].first;
            ranges[2 * r] = tot, ranges[2 * r + 1] = rg[r].second;
            tot += rg[r].second;
        }
        auto resolve = [&](const double *p) -> uint64_t {
            size_t r = upper_bound(starts.begin(), starts.end(), p) - starts.begin() - 1;
            return ranges[2 * r] + (uint64_t)(p - starts[r]);
        };
        for (size_t i = 0; i < n; i++) {
            pairs[i].y_off = resolve(b0->b[i]);
            pairs[i].z_off = resolve(b1->a[i]);
        }
        pf.arena_len = tot;
        pf.n_ranges = rg.size();
        pf.pairs = pairs.data();
        pf.ranges = ranges.data();
        double meta[8] = {(double)isw,
                          (double)site,
                          (double)dmrg->me->mpo->const_e,
                          0.0,
                          (double)dmrg->forward,
                          (double)seq->batch[0]->nflop + (double)seq->batch[1]->nflop,
                          (double)dmrg->me->n_sites,
                          0.0};
        pf.meta = meta, pf.n_meta = 8;
        vector<double> arena, sigma;
        if (with_data) {
            arena.resize(tot);
            for (size_t r = 0; r < rg.size(); r++)
                memcpy(arena.data() + ranges[2 * r], rg[r].first, rg[r].second * 8);
            sigma.assign(pf.sigma_len, 0.0);
            // reference replay: sigma = 1.0 * H * psi  (Tasked executor)
            (*h->tf)(GMatrix<double>(h->ket->data, (MKL_INT)pf.psi_len, 1),
                     GMatrix<double>(sigma.data(), (MKL_INT)pf.sigma_len, 1), 1.0);
            pf.arena = arena.data();
            pf.psi = h->ket->data;
            pf.sigma_ref = sigma.data();
            pf.diag = h->diag->data;
            pf.flags = B2XPF_ARENA | B2XPF_PSI | B2XPF_SIGMA | B2XPF_DIAG;
        }
        h->post_precompute();
        stringstream fn;
        fn << spec.prefix << ".sw" << isw << ".site" << site << (with_data ? ".plan" : ".struct");
        b2x_planfile_write(fn.str().c_str(), &pf);
        stringstream ss;
        ss << "DUMP " << fn.str() << " pairs=" << n << " psi=" << pf.psi_len << " arena=" << tot
           << " ranges=" << rg.size() << " max_work=" << pf.max_work
           << " macs=" << (uint64_t)meta[5];
        log.push_back(ss.str());
        cerr << ss.str() << endl;
    }
};

static set<pair<int, int>> parse_pairs(const string &s) {
    set<pair<int, int>> r;
    for (auto &tok : Parsing::split(s, ",", true)) {
        auto ab = Parsing::split(tok, ":", true);
        r.insert(make_pair(Parsing::to_int(ab[0]), Parsing::to_int(ab[1])));
    }
    return r;
}

template <typename S>
int run(const string &fd, int M, int n_sweeps, const string &prefix, map<string, string> &kv) {
    typedef double FL;
    size_t isize = 1LL << 28, dsize = 1LL << 33;
    if (kv.count("stack_gb")) // (bytes of the double stack; large-M structure runs need more than the 8 GB default)
        dsize = (size_t)Parsing::to_int(kv["stack_gb"]) << 30;
    int nth = kv.count("nthreads") ? Parsing::to_int(kv["nthreads"]) : 8;
    frame_<double>() = make_shared<DataFrame<double>>(isize, dsize, kv.count("scratch") ? kv["scratch"] : "/tmp/b2x_ref_scratch");
    frame_<double>()->use_main_stack = kv.count("main_stack") != 0; // (main_stack=1: block2's default — renormalised operators
                                                                    //  live on the frame's stack and go into the partition files)
    frame_<double>()->minimal_disk_usage = true;
    threading_() = make_shared<Threading>(ThreadingTypes::OperatorBatchedGEMM | ThreadingTypes::Global, nth, nth, 1);
    threading_()->seq_type = SeqTypes::Tasked;
    shared_ptr<FCIDUMP<FL>> fcidump;
    PGTypes pg = PGTypes::D2H;
    if (kv.count("pg") && kv["pg"] == "c1")
        pg = PGTypes::C1;
    if (fd.substr(0, 8) == "hubbard:") {
        auto t = Parsing::split(fd, ":", true);
        fcidump = make_shared<HubbardFCIDUMP>((uint16_t)Parsing::to_int(t[1]), Parsing::to_double(t[2]), Parsing::to_double(t[3]), false);
        pg = PGTypes::C1;
    } else {
        fcidump = make_shared<FCIDUMP<FL>>();
        fcidump->read(fd);
    }
    fcidump->rescale();
    vector<uint8_t> orbsym = fcidump->template orb_sym<uint8_t>();
    transform(orbsym.begin(), orbsym.end(), orbsym.begin(), PointGroup::swap_pg(pg));
    S vacuum(0);
    int norb = fcidump->n_sites();
    S target(fcidump->n_elec(), fcidump->twos(), PointGroup::swap_pg(pg)(fcidump->isym()));
    // para=i|ij: the sum-MPO parallel rule: integrals masked by ParallelRuleSimple::index_prefactor
    // (src/dmrg/parallel_simple.hpp:56-99; wrapped as in unit_test/mpi/test_sum_mpo_n2_sto3g.cpp:183-190 and
    // src/main.cpp:233-238), every rank builds the MPO of ITS integrals, ParallelMPO around it.  Built with -D_HAS_MPI and started under mpirun every rank runs its own H_r; the
    // captured plan is the rank's, sigma_ref is what ParallelTensorFunctions::operator() returned: the ALL-REDUCED H psi.
    shared_ptr<ParallelRuleSimple<S, FL>> para_rule;
    string rank_tag;
    if (kv.count("para")) {
#ifdef _HAS_MPI
        shared_ptr<ParallelCommunicator<S>> comm = make_shared<MPICommunicator<S>>();
#else
        shared_ptr<ParallelCommunicator<S>> comm = make_shared<ParallelCommunicator<S>>(1, 0, 0);
#endif
        para_rule = make_shared<ParallelRuleSimple<S, FL>>(
            kv["para"] == "ij" ? ParallelSimpleTypes::IJ : ParallelSimpleTypes::I, comm);
        fcidump = make_shared<ParallelFCIDUMP<S, FL>>(fcidump, para_rule);
        if (comm->size > 1)
            rank_tag = ".r" + Parsing::to_string(comm->rank) + "of" + Parsing::to_string(comm->size);
        if (kv.count("prefactors")) { // this rank's share of every integral index (what the partition rule must reproduce)
            ArrayFile af(prefix + rank_tag + ".prefactors");
            vector<double> pij((size_t)norb * norb), pijkl((size_t)norb * norb * norb * norb);
            for (int i = 0; i < norb; i++)
                for (int j = 0; j < norb; j++) {
                    pij[(size_t)i * norb + j] = para_rule->index_prefactor(i, j);
                    for (int k = 0; k < norb; k++)
                        for (int l = 0; l < norb; l++)
                            pijkl[(((size_t)i * norb + j) * norb + k) * norb + l] = para_rule->index_prefactor(i, j, k, l);
                }
            af.u64("meta", vector<uint64_t>{(uint64_t)norb, (uint64_t)comm->rank, (uint64_t)comm->size,
                                            (uint64_t)(kv["para"] == "ij" ? 3 : 1)});
            af.f64("ij", pij), af.f64("ijkl", pijkl);
        }
    }
    auto hamil = make_shared<HamiltonianQC<S, FL>>(vacuum, norb, orbsym, fcidump);
    shared_ptr<MPO<S, FL>> mpo;
    if (para_rule != nullptr) {
        // as src/main.cpp:312-320, 463-470 ("simple_parallel"): the ordinary QC MPO of the masked integrals, the
        // NC -> CN switch moved to trans_center = norb / (1 + 1/sqrt(size)), ParallelMPO around it
        int trans_center = (int)(norb / (1 + 1.0 / sqrt((double)para_rule->comm->size)));
        QCTypes qct = trans_center >= norb - 2 ? QCTypes::NC : QCTypes::Conventional;
        mpo = make_shared<MPOQC<S, FL>>(hamil, qct, "HQC", trans_center);
        mpo = make_shared<SimplifiedMPO<S, FL>>(mpo, make_shared<RuleQC<S, FL>>(), true);
        mpo = make_shared<ParallelMPO<S, FL>>(mpo, para_rule);
    } else {
        mpo = make_shared<MPOQC<S, FL>>(hamil, QCTypes::Conventional);
        mpo = make_shared<SimplifiedMPO<S, FL>>(mpo, make_shared<RuleQC<S, FL>>(), true, true,
                                                 OpNamesSet({OpNames::R, OpNames::RD}));
    }
    Random::rand_seed(kv.count("seed") ? (unsigned)Parsing::to_int(kv["seed"]) : 1234u);
    auto mps_info = make_shared<MPSInfo<S>>(norb, vacuum, target, hamil->basis);
    if (kv.count("occ")) {
        vector<double> occs = read_occ(kv["occ"]);
        mps_info->set_bond_dimension_using_occ((ubond_t)M, occs, 1);
    } else
        mps_info->set_bond_dimension((ubond_t)M);
    cout << "LEFT_DIMS";
    for (int i = 0; i <= norb; i++)
        cout << " " << mps_info->left_dims[i]->n_states_total;
    cout << endl;
    if (kv.count("info_only")) {
        // print per-bond sector tables and stop (no tensors are allocated)
        for (int i = 0; i <= norb; i++) {
            cout << "LEFT " << i << " n=" << mps_info->left_dims[i]->n << " :";
            for (int k = 0; k < mps_info->left_dims[i]->n; k++)
                cout << " " << mps_info->left_dims[i]->n_states[k];
            cout << endl;
            cout << "RIGHT " << i << " n=" << mps_info->right_dims[i]->n << " :";
            for (int k = 0; k < mps_info->right_dims[i]->n; k++)
                cout << " " << mps_info->right_dims[i]->n_states[k];
            cout << endl;
        }
        return 0;
    }
    auto mps = make_shared<MPS<S, FL>>(norb, 0, 2);
    mps->initialize(mps_info);
    mps->random_canonicalize();
    mps->save_mutable();
    mps->deallocate();
    mps_info->save_mutable();
    mps_info->deallocate_mutable();
    auto me = make_shared<MovingEnvironment<S, FL, FL>>(mpo, mps, mps, "DMRG");
    auto dumper = make_shared<Dumper<S>>();
    dumper->spec.prefix = prefix + rank_tag;
    // the recording TensorFunctions: over the serial base, or — sum-MPO run — over ParallelTensorFunctions with the rule
    bool rtf_installed = false;
    std::function<void(DMRG<S, FL, FL> *)> rtf_set_dmrg = [](DMRG<S, FL, FL> *) {};
    auto install_rtf = [&](DMRG<S, FL, FL> *dm) {
        if (para_rule != nullptr)
            setup_rtf<S>(make_shared<RotTFT<S, ParallelTensorFunctions<S, FL>>>(mpo->tf->opf, para_rule), me.get(), dm,
                         &dumper->spec, &dumper->log, mpo, rtf_set_dmrg);
        else
            setup_rtf<S>(make_shared<RotTF<S>>(mpo->tf->opf), me.get(), dm, &dumper->spec, &dumper->log, mpo, rtf_set_dmrg);
        rtf_installed = true;
    };
    if (kv.count("chain")) { // every blocking / rotation from the very first one: hook in before the environments exist
        dumper->spec.chain = Parsing::to_int(kv["chain"]);
        dumper->spec.evlog = &dumper->log;
        install_rtf(nullptr);
    }
    me->init_environments(false);
    if (para_rule == nullptr && !kv.count("nodelay")) // (the reference's sum-MPO test keeps the default: no delayed contraction)
        me->delayed_contraction = OpNamesSet::normal_ops();
    me->cached_contraction = !kv.count("nocache");
    vector<ubond_t> bdims = {(ubond_t)M};
    vector<double> noises = {1E-8, 1E-9, 0.0};
    if (kv.count("noise")) {
        noises.clear();
        for (auto &x : Parsing::split(kv["noise"], ",", true))
            noises.push_back(Parsing::to_double(x));
    }
    auto dmrg = make_shared<DMRG<S, FL, FL>>(me, bdims, noises);
    dmrg->iprint = kv.count("iprint") ? Parsing::to_int(kv["iprint"]) : 1;
    dmrg->noise_type = NoiseTypes::ReducedPerturbative;
    dmrg->decomp_type = DecompositionTypes::DensityMatrix;
    dmrg->davidson_soft_max_iter = kv.count("dav_iter") ? Parsing::to_int(kv["dav_iter"]) : 4000;
    if (kv.count("cutoff")) // DMRG::cutoff: density-matrix weights below it are never kept (default 1e-14)
        dmrg->cutoff = Parsing::to_double(kv["cutoff"]);
    if (kv.count("dav_thrd")) // (default: noise * 0.1 in a noisy sweep, tol * 0.1 otherwise, sweep_algorithm.hpp:3038-3047)
        dmrg->davidson_conv_thrds = vector<double>(max(n_sweeps, 1), Parsing::to_double(kv["dav_thrd"]));
    dumper->dmrg = dmrg.get();
    dumper->start_forward = mps->center == 0;
    rtf_set_dmrg(dmrg.get());
    if (kv.count("stop_after"))
        dumper->stop_after = *parse_pairs(kv["stop_after"]).begin();
    if (kv.count("spectra"))
        dumper->spectra = true, dmrg->store_wfn_spectra = true;
    if (kv.count("dump"))
        dumper->spec.with_data = parse_pairs(kv["dump"]);
    if (kv.count("struct"))
        dumper->spec.structure = parse_pairs(kv["struct"]);
    if (kv.count("eham"))
        dumper->spec.eham = parse_pairs(kv["eham"]);
    if (kv.count("pnoise"))
        dumper->spec.pnoise = parse_pairs(kv["pnoise"]);
    if (kv.count("pnoise_struct"))
        dumper->spec.pnoise_struct = parse_pairs(kv["pnoise_struct"]);
    if (kv.count("enoise"))
        dumper->spec.enoise = parse_pairs(kv["enoise"]);
    if (kv.count("rot"))
        dumper->spec.rot = parse_pairs(kv["rot"]);
    if (kv.count("rot_struct"))
        dumper->spec.rot_struct = parse_pairs(kv["rot_struct"]);
    if (kv.count("erot"))
        dumper->spec.erot = parse_pairs(kv["erot"]);
    if (kv.count("blk"))
        dumper->spec.blk = parse_pairs(kv["blk"]);
    if (kv.count("blk_struct"))
        dumper->spec.blk_struct = parse_pairs(kv["blk_struct"]);
    if (kv.count("eblk"))
        dumper->spec.eblk = parse_pairs(kv["eblk"]);
    if (!rtf_installed &&
        (!dumper->spec.rot.empty() || !dumper->spec.rot_struct.empty() || !dumper->spec.erot.empty() ||
         !dumper->spec.blk.empty() || !dumper->spec.blk_struct.empty() || !dumper->spec.eblk.empty()))
        install_rtf(dmrg.get());
    callback_() = dumper;
    double tol = kv.count("tol") ? Parsing::to_double(kv["tol"]) : 1E-8;
    Timer t;
    t.get_time();
    double energy = dmrg->solve(n_sweeps, mps->center == 0, tol);
    double tt = t.get_time();
    callback_() = make_shared<CallbackKernel>();
    if (kv.count("tensor_file"))
        for (auto &tok : Parsing::split(kv["tensor_file"], ",", true)) {
            int i = Parsing::to_int(tok);
            mps->load_tensor(i);
            auto t = mps->tensors[i];
            string fn = prefix + ".mps" + tok + ".tensor";
            t->save_data(fn, true); // info + factor + total_memory + data, as MPS::save_tensor writes it (mps.hpp:2573-2578)
            if (kv.count("fp_prec")) { // the same tensor in compressed storage (FPCodec, src/core/fp_codec.hpp:158-)
                frame_<double>()->fp_codec = make_shared<FPCodec<double>>(Parsing::to_double(kv["fp_prec"]),
                                                                            kv.count("fp_chunk") ? (size_t)Parsing::to_int(kv["fp_chunk"]) : 1024);
                frame_<double>()->compressed_sparse_tensor_storage = true;
                t->save_data(fn + ".fpc", true);
                frame_<double>()->compressed_sparse_tensor_storage = false;
                frame_<double>()->fp_codec = nullptr;
            }
            EhamDump<S> ed(fn + ".arr");
            int id = ed.info_id(t->info, false);
            ed.af.f64("factor", vector<double>{t->factor});
            ed.af.u64("info", vector<uint64_t>{(uint64_t)id, (uint64_t)t->total_memory});
            ed.af.f64("data", t->data, t->total_memory);
            cerr << "TENSOR " << fn << " n=" << t->info->n << " len=" << t->total_memory << endl;
            mps->unload_tensor(i);
        }
    ofstream lf((prefix + rank_tag + ".log").c_str());
    lf.precision(15);
    for (auto &l : dumper->log)
        lf << l << endl;
    lf << "FINAL_ENERGY " << energy << endl;
    lf << "TOTAL_TIME " << tt << endl;
    for (size_t i = 0; i < dmrg->energies.size(); i++)
        lf << "SWEEP_ENERGY " << i << " " << dmrg->energies[i][0] << endl; // (DMRG::sweep_time is only filled by the
                                                                             // parallel-site sweep: indexing it here was
                                                                             // the crash at the end of every earlier run)
    cout.precision(15);
    cout << "FINAL_ENERGY " << energy << " T = " << tt << endl;
    // teardown in the order the reference's own tests use (unit_test/test_dmrg_n2_sto3g.cpp:121-122, 142, 236-237, 39-43):
    // persistent stack memory is released explicitly, newest first, before the frame goes away
    mps_info->deallocate();
    if (!kv.count("keep_part")) // keep_part=1: the partition files stay in scratch= (on-disk format fixtures)
        me->remove_partition_files();
    mpo->deallocate();
    hamil->deallocate();
    fcidump->deallocate();
    dumper->dmrg = nullptr;
    frame_<double>()->activate(0);
    frame_<double>() = nullptr;
    return 0;
}

static void on_fault(int sig) { // a crashing generator must not pass for a finished one: say where, exit non-zero
    void *bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "ref_dump: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}

int main(int argc, char **argv) {
    signal(SIGSEGV, on_fault), signal(SIGABRT, on_fault);
    if (argc < 6) {
        cerr << "usage: ref_dump <fcidump|hubbard:L:t:U> <su2|sz> <M> <n_sweeps> <outprefix> [key=value ...]" << endl;
        return 2;
    }
    map<string, string> kv;
    for (int i = 6; i < argc; i++) {
        string a = argv[i];
        size_t e = a.find('=');
        kv[a.substr(0, e)] = e == string::npos ? "1" : a.substr(e + 1);
    }
    string sym = argv[2];
    if (sym == "su2")
        return run<SU2>(argv[1], atoi(argv[3]), atoi(argv[4]), argv[5], kv);
    else
        return run<SZ>(argv[1], atoi(argv[3]), atoi(argv[4]), argv[5], kv);
}
