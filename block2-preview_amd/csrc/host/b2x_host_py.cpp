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
    py::class_<EffectiveHamiltonian>(m, "EffectiveHamiltonian")
        .def(py::init([](std::shared_ptr<PySeq> seq, std::vector<double> diag) {
            return new EffectiveHamiltonian(std::static_pointer_cast<BatchGEMMSeq>(seq), diag);
        }))
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
             double shift, std::vector<uintptr_t> ors, std::vector<double> proj_weights, py::object comm) -> py::object {
              b2x_plan *p = (b2x_plan *)plan;
              DeviceComm dc;
              const bool para = !comm.is_none();
              if (para) {
                  py::tuple t = comm.cast<py::tuple>();
                  dc.comm = (b2x_comm *)t[0].cast<uintptr_t>();
                  dc.rank = t[1].cast<int>(), dc.size = t[2].cast<int>(), dc.root = t[3].cast<int>();
              }
              size_t slen = n;
              auto f = [p, para, dc, slen](const double *b, double *s) {
                  check(b2x_plan_execute(p, b, s, 1.0, 1, nullptr));
                  if (para) // ParallelTensorFunctions::operator() (parallel_tensor_functions.hpp:51-55)
                      check(b2x_allreduce_sum(dc.comm, s, slen, nullptr));
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
              std::vector<double> e = IterativeMatrixFunctions::harmonic_davidson(
                  f, (const double *)diag_dev, vs, n, shift, davidson_type, ndav, iprint, para ? &dc : nullptr, conv_thrd,
                  rel_conv_thrd, max_iter, soft_max_iter, deflation_min_size, deflation_max_size, o, proj_weights);
              if (many)
                  return py::make_tuple(e, ndav);
              return py::make_tuple(e[0], ndav);
          },
          py::arg("plan"), py::arg("diag_dev"), py::arg("ket_dev"), py::arg("n"), py::arg("conv_thrd") = 5E-6,
          py::arg("max_iter") = 5000, py::arg("soft_max_iter") = -1, py::arg("deflation_min_size") = 2,
          py::arg("deflation_max_size") = 50, py::arg("iprint") = false, py::arg("rel_conv_thrd") = 0.0,
          py::arg("davidson_type") = DavidsonTypes::Normal, py::arg("shift") = 0.0,
          py::arg("ors") = std::vector<uintptr_t>(), py::arg("proj_weights") = std::vector<double>(),
          py::arg("comm") = py::none());
    b2xh::bind_symbolic(m);
    m.def("device_init", [](int ordinal) { check(b2x_device_init(ordinal)); }, py::arg("ordinal") = 0);
    m.def("small_eigs", [](std::vector<double> a, int n) {
        std::vector<double> w;
        small_eigs(a, w, n);
        return py::make_tuple(w, a);
    });
}
