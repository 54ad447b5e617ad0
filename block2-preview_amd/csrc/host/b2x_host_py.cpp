// b2x_host_py.cpp — pybind11 surface of the C++ host mirror (module `b2x_host`), the analogue of block2's
// bindings for these classes: BatchGEMMSeq (src/pybind/pybind_core.hpp:3793-3830), EffectiveHamiltonian
// incl. __call__ / eigs (src/pybind/pybind_dmrg.hpp:588-640), SeqTypes (pybind_core.hpp:2063-2071).
// GMatrix arguments are numpy arrays (operators: 2-D float64, C-contiguous) or integer element offsets
// (psi / psi' operands, which block2 records "from null").
#include "b2x_host.hpp"
#include "b2x_symbolic_py.hpp"
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;
using namespace b2xh;
typedef py::array_t<double, py::array::c_style | py::array::forcecast> arr;

static GMatrix gm(const arr &a) {
    if (a.ndim() != 2)
        throw std::runtime_error("expected a 2-D float64 array");
    return GMatrix(const_cast<double *>(a.data()), (int)a.shape(0), (int)a.shape(1));
}
static GMatrix off(uint64_t o, int m, int n) { return GMatrix((double *)0 + o, m, n); }

// operand of a single-GEMM record: a 2-D float64 array (operator block) or (flat float64 array, element offset, m, n)
// for a block of the wavefunction / the perturbed wavefunctions (row length n at flat + offset)
static GMatrix gmx(const py::object &o, std::vector<py::object> &keep) {
    keep.push_back(o);
    if (py::isinstance<py::tuple>(o)) {
        py::tuple t = o.cast<py::tuple>();
        auto flat = t[0].cast<py::array_t<double, py::array::c_style>>();
        uint64_t off = t[1].cast<uint64_t>();
        int m = t[2].cast<int>(), n = t[3].cast<int>();
        if (off + (uint64_t)m * n > (uint64_t)flat.size())
            throw std::runtime_error("block runs past the end of the flat vector");
        return GMatrix(const_cast<double *>(flat.data()) + off, m, n);
    }
    auto a = o.cast<py::array_t<double, py::array::c_style>>();
    if (a.ndim() != 2)
        throw std::runtime_error("expected a 2-D float64 array");
    return GMatrix(const_cast<double *>(a.data()), (int)a.shape(0), (int)a.shape(1));
}

struct PySeq : BatchGEMMSeq {
    std::vector<py::object> keep; // operator arrays must outlive the plan upload
    using BatchGEMMSeq::BatchGEMMSeq;
};

// Python-overridable EffectiveKernel (src/pybind/pybind_dmrg.hpp:1107-1165): compute(beta, f, a, b, xs) with a, b, xs the
// DEVICE addresses (int) of |psi| doubles and f(a, b, scale) the matrix-vector product b += scale * H a
struct PyEffectiveKernel : EffectiveKernel {
    void compute(double beta, const MatMul &f, const double *a, double *b, const std::vector<const double *> &xs) const override {
        py::gil_scoped_acquire gil;
        py::function over = py::get_override(static_cast<const EffectiveKernel *>(this), "compute");
        if (!over) {
            EffectiveKernel::compute(beta, f, a, b, xs);
            return;
        }
        py::cpp_function pf([&f](uintptr_t pa, uintptr_t pb, double scale) { f((const double *)pa, (double *)pb, scale); });
        std::vector<uintptr_t> px;
        for (const double *x : xs)
            px.push_back((uintptr_t)x);
        over(beta, pf, (uintptr_t)a, (uintptr_t)b, px);
    }
};

// Named classes of the hot path as block2's pybind exposes them per symmetry (src/pybind/pybind_core.hpp:1100-1340):
// OperatorFunctions owns the sequence (mode from Global.threading.seq_type), TensorFunctions forwards operator() to it
// (tensor_functions.hpp:59-62) and carries the symbolic walks; their inputs are the fixture dictionaries of
// oracle/ref_dump.cpp, because OperatorTensor / SparseMatrixInfo objects are not bound as classes.
template <typename S> struct PyOpf {
    std::shared_ptr<PySeq> seq;
};
template <typename S> struct PyTf {
    std::shared_ptr<PyOpf<S>> opf;
    std::shared_ptr<ParallelCommunicator> comm; // ParallelTensorFunctions only
    static bool is_right(const py::dict &d) { return SymEH<S>::template arr<uint64_t>(d, "meta")[2] != 0; }
};
template <typename S> static void bind_functions(py::module_ &m, const char *doc) {
    py::module_ sm = m.def_submodule(std::is_same<S, SU2>::value ? "su2" : "sz", doc);
    py::class_<PyOpf<S>, std::shared_ptr<PyOpf<S>>>(sm, "OperatorFunctions")
        .def(py::init([](py::object seq) {
                 auto o = std::make_shared<PyOpf<S>>();
                 o->seq = seq.is_none() ? std::make_shared<PySeq>((size_t)1 << 24, threading_()->seq_type)
                                        : seq.cast<std::shared_ptr<PySeq>>();
                 return o;
             }),
             py::arg("seq") = py::none())
        .def_readwrite("seq", &PyOpf<S>::seq);
    auto call = [](PyTf<S> &t, py::array_t<double, py::array::c_style> b, py::array_t<double, py::array::c_style> c, double scale) {
        BatchGEMMSeq &q = *t.opf->seq;
        q.prepare((size_t)b.size(), (size_t)c.size());
        if (t.comm == nullptr) { // TensorFunctions::operator() (tensor_functions.hpp:59-62)
            q(GMatrix(b.mutable_data(), (int)b.size(), 1), GMatrix(c.mutable_data(), (int)c.size(), 1), scale);
            return;
        }
        // ParallelTensorFunctions::operator() (parallel_tensor_functions.hpp:51-55): local H_r psi, then the all-reduce
        DeviceVector db((size_t)b.size()), dc((size_t)c.size());
        db.upload(b.data());
        check(b2x_vec_zero(dc.p, (size_t)c.size(), nullptr));
        q.apply_device(db.p, dc.p, scale);
        t.comm->allreduce_sum(dc.p, (size_t)c.size());
        std::vector<double> h((size_t)c.size());
        dc.download(h.data());
        for (py::ssize_t i = 0; i < c.size(); i++)
            c.mutable_data()[i] += h[(size_t)i];
    };
    auto rot = [](bool want_right) {
        return [want_right](PyTf<S> &, const py::dict &d, bool execute) {
            if (PyTf<S>::is_right(d) != want_right)
                throw std::runtime_error(want_right ? "right_rotate: the fixture is a left rotation" : "left_rotate: the fixture is a right rotation");
            return sym_rotate<S>(d, execute);
        };
    };
    py::class_<PyTf<S>, std::shared_ptr<PyTf<S>>>(sm, "TensorFunctions")
        .def(py::init([](std::shared_ptr<PyOpf<S>> opf) {
            auto t = std::make_shared<PyTf<S>>();
            t->opf = opf;
            return t;
        }))
        .def_readwrite("opf", &PyTf<S>::opf)
        .def_readonly("comm", &PyTf<S>::comm)
        .def("__call__", call, py::arg("b"), py::arg("c"), py::arg("scale") = 1.0)
        .def("left_rotate", rot(false), py::arg("fixture"), py::arg("execute") = false)
        .def("right_rotate", rot(true), py::arg("fixture"), py::arg("execute") = false)
        .def("left_contract", [](PyTf<S> &, const py::dict &d, bool execute) { return sym_blocking<S>(d, execute); },
             py::arg("fixture"), py::arg("execute") = false)
        .def("right_contract", [](PyTf<S> &, const py::dict &d, bool execute) { return sym_blocking<S>(d, execute); },
             py::arg("fixture"), py::arg("execute") = false)
        .def("numerical_transform", [](PyTf<S> &, const py::dict &d) { return sym_transform<S>(d); }, py::arg("fixture"));
    // ParallelTensorFunctions(opf, comm): the same object with a communicator
    sm.def("ParallelTensorFunctions", [](std::shared_ptr<PyOpf<S>> opf, std::shared_ptr<ParallelCommunicator> comm) {
        auto t = std::make_shared<PyTf<S>>();
        t->opf = opf, t->comm = comm;
        return t;
    });
}

PYBIND11_MODULE(b2x_host, m) {
    m.doc() = "C++ host mirror of block2's H.psi interface over the MI355X C ABI (include/b2x.h)";
    py::enum_<DavidsonTypes>(m, "DavidsonTypes", py::arithmetic()) // pybind_core.hpp (same names / values)
        .value("Normal", DavidsonTypes::Normal)
        .value("GreaterThan", DavidsonTypes::GreaterThan)
        .value("LessThan", DavidsonTypes::LessThan)
        .value("CloseTo", DavidsonTypes::CloseTo)
        .value("Harmonic", DavidsonTypes::Harmonic)
        .value("HarmonicGreaterThan", DavidsonTypes::HarmonicGreaterThan)
        .value("HarmonicLessThan", DavidsonTypes::HarmonicLessThan)
        .value("HarmonicCloseTo", DavidsonTypes::HarmonicCloseTo)
        .value("DavidsonPrecond", DavidsonTypes::DavidsonPrecond)
        .value("NoPrecond", DavidsonTypes::NoPrecond)
        .value("NonHermitian", DavidsonTypes::NonHermitian)
        .value("Exact", DavidsonTypes::Exact)
        .value("LeftEigen", DavidsonTypes::LeftEigen)
        .value("ElementProj", DavidsonTypes::ElementProj)
        .def("__or__", [](DavidsonTypes a, DavidsonTypes b) { return a | b; });
    py::enum_<SeqTypes>(m, "SeqTypes", py::arithmetic())
        .value("Nothing", SeqTypes::None)
        .value("Simple", SeqTypes::Simple)
        .value("Auto", SeqTypes::Auto)
        .value("Tasked", SeqTypes::Tasked)
        .value("SimpleTasked", SeqTypes::SimpleTasked)
        .value("Device", SeqTypes::Device);
    py::class_<PySeq, std::shared_ptr<PySeq>>(m, "BatchGEMMSeq")
        .def(py::init([](size_t max_batch_flops, SeqTypes mode) { return std::make_shared<PySeq>(max_batch_flops, mode); }),
             py::arg("max_batch_flops") = (size_t)1 << 24, py::arg("mode") = SeqTypes::Device)
        .def_readwrite("mode", &PySeq::mode)
        .def_readonly("max_work", &PySeq::max_work)
        .def_readonly("nflop", &PySeq::nflop)
        .def_readwrite("cumulative_nflop", &PySeq::cumulative_nflop)
        .def_property_readonly("n_pairs", [](const PySeq &s) { return s.pairs.size(); })
        // rotate(a=(offset, m, n), c=(offset, m, n), bra, conj_bra, ket, conj_ket, scale)
        .def("rotate",
             [](PySeq &s, std::tuple<uint64_t, int, int> a, std::tuple<uint64_t, int, int> c, arr bra, uint8_t conj_bra,
                arr ket, uint8_t conj_ket, double scale) {
                 s.keep.push_back(bra), s.keep.push_back(ket);
                 s.rotate(off(std::get<0>(a), std::get<1>(a), std::get<2>(a)),
                          off(std::get<0>(c), std::get<1>(c), std::get<2>(c)), gm(bra), conj_bra, gm(ket), conj_ket,
                          scale);
             })
        .def("three_rotate",
             [](PySeq &s, std::tuple<uint64_t, int, int> a, std::tuple<uint64_t, int, int> c, arr bra, bool conj_bra,
                arr ket, bool conj_ket, arr da, bool dconja, arr db, bool dconjb, bool dleft, double scale,
                uint64_t stride) {
                 s.keep.push_back(bra), s.keep.push_back(ket), s.keep.push_back(da), s.keep.push_back(db);
                 s.three_rotate(off(std::get<0>(a), std::get<1>(a), std::get<2>(a)),
                                off(std::get<0>(c), std::get<1>(c), std::get<2>(c)), gm(bra), conj_bra, gm(ket),
                                conj_ket, gm(da), dconja, gm(db), dconjb, dleft, scale, stride);
             })
        // load a recorded plan: b2x_pair records (structured array viewed as bytes) + the operator arena
        .def("load_pairs",
             [](PySeq &s, py::array pairs, arr arena) {
                 if (pairs.itemsize() != (py::ssize_t)sizeof(b2x_pair))
                     throw std::runtime_error("pairs: itemsize must equal sizeof(b2x_pair)");
                 s.keep.push_back(arena);
                 const b2x_pair *p = (const b2x_pair *)pairs.data();
                 for (py::ssize_t i = 0; i < pairs.shape(0); i++) {
                     const b2x_pair &q = p[i];
                     s.push(q.ta0, q.tb0, q.m0, q.n0, q.k0, q.alpha0, (const double *)0 + q.x_off, q.lda0,
                            arena.data() + q.y_off, q.ldb0, q.ta1, q.m1, q.k1, q.alpha1, arena.data() + q.z_off, q.lda1,
                            (double *)0 + q.v_off, q.ldc1);
                 }
             })
        // ---- element-wise block products (blocking) ----
        .def_property_readonly("n_outer", [](const PySeq &s) { return s.outer_terms.size(); })
        .def("outer_dims",
             [](const PySeq &s) { // (m, n, a_rs, a_cs, b_rs, b_cs, ldc, alpha) per recorded term
                 std::vector<std::tuple<int, int, int, int, int, int, int, double>> r;
                 for (const b2x_outer_term &t : s.outer_terms)
                     r.emplace_back(t.m, t.n, t.a_rs, t.a_cs, t.b_rs, t.b_cs, t.ldc, t.alpha);
                 return r;
             })
        .def("tensor_product",
             [](PySeq &s, py::object a, bool conja, py::object b, bool conjb, py::object c, double scale, uint64_t stride) {
                 s.tensor_product(gmx(a, s.keep), conja, gmx(b, s.keep), conjb, gmx(c, s.keep), scale, stride);
             })
        .def("iadd",
             [](PySeq &s, py::object a, py::object b, double scale, bool conj, double cfactor) {
                 s.iadd(gmx(a, s.keep), gmx(b, s.keep), scale, conj, cfactor);
             },
             py::arg("a"), py::arg("b"), py::arg("scale") = 1.0, py::arg("conj") = false, py::arg("cfactor") = 1.0)
        // load recorded b2x_outer_term records: operands resolved against (arena, vin), outputs against vout
        .def("load_outer",
             [](PySeq &s, py::array terms, arr arena, py::array_t<double, py::array::c_style> vin,
                py::array_t<double, py::array::c_style> vout) {
                 if (terms.itemsize() != (py::ssize_t)sizeof(b2x_outer_term))
                     throw std::runtime_error("terms: itemsize must equal sizeof(b2x_outer_term)");
                 s.keep.push_back(arena), s.keep.push_back(vin), s.keep.push_back(vout);
                 const b2x_outer_term *t = (const b2x_outer_term *)terms.data();
                 for (py::ssize_t i = 0; i < terms.shape(0); i++) {
                     const b2x_outer_term &q = t[i];
                     const double *a = q.a_src == 2 ? nullptr : (q.a_src ? vin.data() : arena.data()) + q.a_off;
                     const double *b = q.b_src == 2 ? nullptr : (q.b_src ? vin.data() : arena.data()) + q.b_off;
                     s.push_outer(q.m, q.n, a, q.a_rs, q.a_cs, b, q.b_rs, q.b_cs, vout.mutable_data() + q.c_off, q.ldc,
                                  q.alpha);
                 }
             })
        // outer_perform(v): v (flat vector holding every output block) += recorded block products
        .def("outer_perform",
             [](PySeq &s, py::array_t<double, py::array::c_style> v) {
                 s.outer_perform({{v.mutable_data(), (size_t)v.size()}});
                 s.keep.clear();
             })
        // ---- single-GEMM lists (perturbative noise) ----
        .def_property_readonly("n_gemms", [](const PySeq &s) { return s.gemms.size(); })
        .def("gemm_dims",
             [](const PySeq &s) { // (ta, tb, m, n, k, lda, ldb, ldc, alpha) per recorded slot
                 std::vector<std::tuple<int, int, int, int, int, int, int, int, double>> r;
                 for (const b2x_gemm &g : s.gemms)
                     r.emplace_back(g.ta, g.tb, g.m, g.n, g.k, g.lda, g.ldb, g.ldc, g.alpha);
                 return r;
             })
        .def("multiply",
             [](PySeq &s, py::object a, uint8_t conja, py::object b, uint8_t conjb, py::object c, double scale,
                double cfactor) {
                 s.multiply(gmx(a, s.keep), conja, gmx(b, s.keep), conjb, gmx(c, s.keep), scale, cfactor);
             })
        .def("three_rotate_tr_left",
             [](PySeq &s, py::object a, py::object c, py::object bra, bool conj_bra, py::object ket, bool conj_ket,
                py::object da, bool dconja, py::object db, bool dconjb, bool dleft, double scale, uint64_t stride) {
                 s.three_rotate_tr_left(gmx(a, s.keep), gmx(c, s.keep), gmx(bra, s.keep), conj_bra, gmx(ket, s.keep),
                                        conj_ket, gmx(da, s.keep), dconja, gmx(db, s.keep), dconjb, dleft, scale, stride);
             })
        .def("three_rotate_tr_right",
             [](PySeq &s, py::object a, py::object c, py::object bra, bool conj_bra, py::object ket, bool conj_ket,
                py::object da, bool dconja, py::object db, bool dconjb, bool dleft, double scale, uint64_t stride) {
                 s.three_rotate_tr_right(gmx(a, s.keep), gmx(c, s.keep), gmx(bra, s.keep), conj_bra, gmx(ket, s.keep),
                                         conj_ket, gmx(da, s.keep), dconja, gmx(db, s.keep), dconjb, dleft, scale, stride);
             })
        // load recorded b2x_gemm records: operands resolved against (arena, vin), outputs against vout
        .def("load_gemms",
             [](PySeq &s, py::array gemms, arr arena, py::array_t<double, py::array::c_style> vin,
                py::array_t<double, py::array::c_style> vout) {
                 if (gemms.itemsize() != (py::ssize_t)sizeof(b2x_gemm))
                     throw std::runtime_error("gemms: itemsize must equal sizeof(b2x_gemm)");
                 s.keep.push_back(arena), s.keep.push_back(vin), s.keep.push_back(vout);
                 const b2x_gemm *g = (const b2x_gemm *)gemms.data();
                 for (py::ssize_t i = 0; i < gemms.shape(0); i++) {
                     const b2x_gemm &q = g[i];
                     s.push_gemm(q.ta, q.tb, q.m, q.n, q.k, q.alpha, (q.a_src ? vin.data() : arena.data()) + q.a_off, q.lda,
                                 (q.b_src ? vin.data() : arena.data()) + q.b_off, q.ldb, vout.mutable_data() + q.c_off,
                                 q.ldc);
                 }
             })
        // auto_perform(v [, wavefunction]): v += recorded list; operands inside `wavefunction` use the input vector
        .def("auto_perform",
             [](PySeq &s, py::array_t<double, py::array::c_style> v, py::object in) {
                 GMatrix gin(nullptr, 0, 0);
                 if (!in.is_none()) {
                     auto a = in.cast<py::array_t<double, py::array::c_style>>();
                     gin = GMatrix(const_cast<double *>(a.data()), (int)a.size(), 1);
                 }
                 s.auto_perform(GMatrix(v.mutable_data(), (int)v.size(), 1), gin);
                 s.keep.clear();
             },
             py::arg("v"), py::arg("wavefunction") = py::none())
        .def("__call__",
             [](PySeq &s, py::array_t<double, py::array::c_style> c, py::array_t<double, py::array::c_style> v,
                double scale) {
                 s(GMatrix(c.mutable_data(), (int)c.size(), 1), GMatrix(v.mutable_data(), (int)v.size(), 1), scale);
             },
             py::arg("c"), py::arg("v"), py::arg("scale") = 1.0)
        .def("deallocate", &PySeq::deallocate)
        .def("clear", [](PySeq &s) {
            s.clear();
            s.keep.clear();
        });
    py::class_<EffectiveHamiltonian>(m, "EffectiveHamiltonian", py::dynamic_attr())
        .def(py::init([](std::shared_ptr<PySeq> seq, std::vector<double> diag) {
            return new EffectiveHamiltonian(std::static_pointer_cast<BatchGEMMSeq>(seq), diag);
        }))
        // (the Python object is kept next to the C++ pointer: an override lives in the Python instance)
        .def_property(
            "eff_kernel", [](py::object self) -> py::object { return py::getattr(self, "_eff_kernel", py::none()); },
            [](py::object self, py::object k) {
                self.cast<EffectiveHamiltonian &>().eff_kernel =
                    k.is_none() ? nullptr : k.cast<std::shared_ptr<EffectiveKernel>>();
                py::setattr(self, "_eff_kernel", k);
            })
        .def("precompute", &EffectiveHamiltonian::precompute)
        .def("post_precompute", &EffectiveHamiltonian::post_precompute)
        .def("__call__",
             [](EffectiveHamiltonian &h, py::array_t<double, py::array::c_style> b,
                py::array_t<double, py::array::c_style> c, double factor) {
                 h(GMatrix(b.mutable_data(), (int)b.size(), 1), GMatrix(c.mutable_data(), (int)c.size(), 1), factor);
             },
             py::arg("b"), py::arg("c"), py::arg("factor") = 1.0)
        // eigs(ket, ...) -> (energy, ndav, nflop, tdav, ket_out); keywords as the reference's eigs
        // (effective_hamiltonian.hpp:471-480); ortho_bra = host vectors of |psi| doubles
        .def("eigs",
             [](EffectiveHamiltonian &h, std::vector<double> ket, double conv_thrd, int max_iter, int soft_max_iter,
                int deflation_min_size, int deflation_max_size, bool iprint, double rel_conv_thrd,
                DavidsonTypes davidson_type, double shift, std::vector<std::vector<double>> ortho_bra,
                std::vector<double> projection_weights) {
                 auto r = h.eigs(ket, iprint, conv_thrd, rel_conv_thrd, max_iter, soft_max_iter, deflation_min_size,
                                 deflation_max_size, davidson_type, shift, nullptr, ortho_bra, projection_weights);
                 return py::make_tuple(std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r),
                                       py::array_t<double>(ket.size(), ket.data()));
             },
             py::arg("ket"), py::arg("conv_thrd") = 5E-6, py::arg("max_iter") = 5000, py::arg("soft_max_iter") = -1,
             py::arg("deflation_min_size") = 2, py::arg("deflation_max_size") = 50, py::arg("iprint") = false,
             py::arg("rel_conv_thrd") = 0.0, py::arg("davidson_type") = DavidsonTypes::Normal, py::arg("shift") = 0.0,
             py::arg("ortho_bra") = std::vector<std::vector<double>>(),
             py::arg("projection_weights") = std::vector<double>());
    // Davidson on a plan that is already resident on the device (a b2x_plan* from the C ABI, e.g. capi.Plan._h.value):
    // diag, kets and ors are device addresses of n doubles; the kets are overwritten with the eigenvectors.  Nothing but
    // the Rayleigh-Ritz scalars crosses PCIe.  comm = (b2x_comm*, rank, size, root) of the sum-MPO communicator: sigma is
    // all-reduced after every H.psi and new basis vectors are broadcast from root, as the reference's davidson does
    // with its pcomm.  -> (eigenvalue[s], number of H.psi applications)
    m.def("davidson_device",
          [](uintptr_t plan, uintptr_t diag_dev, py::object ket_dev, size_t n, double conv_thrd, int max_iter, int soft_max_iter,
             int deflation_min_size, int deflation_max_size, bool iprint, double rel_conv_thrd, DavidsonTypes davidson_type,
             double shift, std::vector<uintptr_t> ors, std::vector<double> proj_weights, py::object comm,
             std::vector<uintptr_t> more_plans) -> py::object {
              b2x_plan *p = (b2x_plan *)plan;
              DeviceComm dc;
              const bool para = !comm.is_none();
              if (para && py::isinstance<py::tuple>(comm)) { // (b2x_comm handle, rank, size, root): RCCL through the C ABI
                  py::tuple t = comm.cast<py::tuple>();
                  dc.comm = (b2x_comm *)t[0].cast<uintptr_t>();
                  dc.rank = t[1].cast<int>(), dc.size = t[2].cast<int>(), dc.root = t[3].cast<int>();
              } else if (para) { // an object with rank / size / root, allreduce_device(ptr, n), broadcast_device(ptr, n, owner)
                  dc.rank = comm.attr("rank").cast<int>(), dc.size = comm.attr("size").cast<int>();
                  dc.root = comm.attr("root").cast<int>();
                  const int root = dc.root;
                  dc.sum_fn = [comm](double *p_, size_t n_) {
                      py::gil_scoped_acquire gil; // (the solve runs without the GIL)
                      comm.attr("allreduce_device")((uintptr_t)p_, n_);
                  };
                  dc.bcast_fn = [comm, root](double *p_, size_t n_) {
                      py::gil_scoped_acquire gil;
                      comm.attr("broadcast_device")((uintptr_t)p_, n_, root);
                  };
              }
              size_t slen = n;
              auto f = [p, para, dc, slen, more_plans](const double *b, double *s) {
                  check(b2x_plan_execute(p, b, s, 1.0, 1, nullptr));
                  for (uintptr_t q : more_plans) // H = sum of several plans (sum-MPO ranks held by one process)
                      check(b2x_plan_execute((b2x_plan *)q, b, s, 1.0, 1, nullptr));
                  if (para) // ParallelTensorFunctions::operator() (parallel_tensor_functions.hpp:51-55)
                      dc.allreduce_sum(s, slen);
              };
              const bool many = py::isinstance<py::sequence>(ket_dev);
              std::vector<double *> vs;
              if (many)
                  for (auto h : ket_dev.cast<py::sequence>())
                      vs.push_back((double *)h.cast<uintptr_t>());
              else
                  vs.push_back((double *)ket_dev.cast<uintptr_t>());
              std::vector<double *> o;
              for (auto x : ors)
                  o.push_back((double *)x);
              int ndav = 0;
              std::vector<double> e;
              const std::function<void(const double *, double *)> fop = f; // (made and destroyed with the GIL held: it may own Python references)
              {
                  // no Python object is touched from here to the end of the solve (a caller-supplied transport takes the GIL for
                  // its own calls): other Python threads may run — the sweep loop prepares the next site on them while the device
                  // iterates, sweep.DMRG._prefetch_next
                  py::gil_scoped_release nogil;
                  e = IterativeMatrixFunctions::harmonic_davidson(
                      fop, (const double *)diag_dev, vs, n, shift, davidson_type, ndav, iprint, para ? &dc : nullptr, conv_thrd,
                      rel_conv_thrd, max_iter, soft_max_iter, deflation_min_size, deflation_max_size, o, proj_weights);
              }
              if (many)
                  return py::make_tuple(e, ndav);
              return py::make_tuple(e[0], ndav);
          },
          py::arg("plan"), py::arg("diag_dev"), py::arg("ket_dev"), py::arg("n"), py::arg("conv_thrd") = 5E-6,
          py::arg("max_iter") = 5000, py::arg("soft_max_iter") = -1, py::arg("deflation_min_size") = 2,
          py::arg("deflation_max_size") = 50, py::arg("iprint") = false, py::arg("rel_conv_thrd") = 0.0,
          py::arg("davidson_type") = DavidsonTypes::Normal, py::arg("shift") = 0.0,
          py::arg("ors") = std::vector<uintptr_t>(), py::arg("proj_weights") = std::vector<double>(),
          py::arg("comm") = py::none(), py::arg("more_plans") = std::vector<uintptr_t>());
    // Threading.seq_type + the process-wide instance (pybind_core.hpp:1726-1731, Global.threading)
    py::class_<Threading, std::shared_ptr<Threading>>(m, "Threading")
        .def(py::init<>())
        .def_readwrite("seq_type", &Threading::seq_type);
    struct GlobalNS {};
    py::class_<GlobalNS>(m, "Global")
        .def_property_static(
            "threading", [](py::object) { return threading_(); },
            [](py::object, std::shared_ptr<Threading> t) { threading_() = t; });
    // ParallelCommunicator / the RCCL communicator (pybind_core.hpp:1267-1340: ParallelCommunicator, MPICommunicator);
    // vectors are DEVICE addresses of fp64 data
    py::class_<ParallelCommunicator, std::shared_ptr<ParallelCommunicator>>(m, "ParallelCommunicator")
        .def(py::init<int, int, int>(), py::arg("size") = 1, py::arg("rank") = 0, py::arg("root") = 0)
        .def_readwrite("size", &ParallelCommunicator::size)
        .def_readwrite("rank", &ParallelCommunicator::rank)
        .def_readwrite("root", &ParallelCommunicator::root)
        .def_readwrite("tcomm", &ParallelCommunicator::tcomm)
        .def("is_root", &ParallelCommunicator::is_root)
        .def("allreduce_sum", [](ParallelCommunicator &c, uintptr_t dev, size_t n) { c.allreduce_sum((double *)dev, n); })
        .def("broadcast", [](ParallelCommunicator &c, uintptr_t dev, size_t n, int owner) { c.broadcast((double *)dev, n, owner); })
        .def("barrier", &ParallelCommunicator::barrier)
        .def_property_readonly("handle", [](const ParallelCommunicator &c) { return (uintptr_t)c.handle(); });
    py::class_<RCCLCommunicator, std::shared_ptr<RCCLCommunicator>, ParallelCommunicator>(m, "RCCLCommunicator")
        .def(py::init<int, int, const std::string &, int>(), py::arg("rank"), py::arg("size"), py::arg("id_file"),
             py::arg("root") = 0);
    py::class_<EffectiveKernel, PyEffectiveKernel, std::shared_ptr<EffectiveKernel>>(m, "EffectiveKernel")
        .def(py::init<>())
        .def("compute", [](const EffectiveKernel &k, double beta, py::function f, uintptr_t a, uintptr_t b, std::vector<uintptr_t> xs) {
            std::vector<const double *> px;
            for (uintptr_t x : xs)
                px.push_back((const double *)x);
            k.EffectiveKernel::compute(beta, [&f](const double *pa, double *pb, double s) { f((uintptr_t)pa, (uintptr_t)pb, s); },
                                       (const double *)a, (double *)b, px);
        });
    bind_functions<SU2>(m, "SU2 instantiation of the hot-path classes");
    bind_functions<SZ>(m, "SZ instantiation of the hot-path classes");
    b2xh::bind_symbolic(m);
    m.def("device_init", [](int ordinal) { check(b2x_device_init(ordinal)); }, py::arg("ordinal") = 0);
    m.def("small_eigs", [](std::vector<double> a, int n) {
        std::vector<double> w;
        small_eigs(a, w, n);
        return py::make_tuple(w, a);
    });
}
