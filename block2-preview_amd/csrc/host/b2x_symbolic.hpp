// b2x_symbolic.hpp — host mirror of the symbolic -> numeric layer that BUILDS the GEMM-pair plan:
//
//   SZ / SU2 labels            src/core/symmetry.hpp:654-731 (SZLongLong), :1183-1306 (SU2LongLong)
//   CG<S>                      src/core/clebsch_gordan.hpp:36-57 (trivial), :58-214 (SU(2) 6j / 9j / transpose)
//   SparseMatrixInfo<S>        src/core/sparse_matrix.hpp:47-60, 713-809   sector table, find_state
//   ConnectionInfo::initialize_wfn   src/core/sparse_matrix.hpp:161-289   (iv, ia, ib, ic, factor) lists
//   SparseMatrix<S>            src/core/sparse_matrix.hpp:876-1050         one data pointer + per-sector offsets
//   OperatorTensor / DelayedOperatorTensor    src/core/operator_tensor.hpp:41-58, 209-268
//   OperatorFunctions::tensor_product_multiply        src/core/operator_functions.hpp:474-542
//   OperatorFunctions::three_tensor_product_multiply  src/core/operator_functions.hpp:543-671
//   TensorFunctions::tensor_product_multiply / operator()   src/core/tensor_functions.hpp:1880-2025, 59-62
//   EffectiveHamiltonian ctor / precompute             src/dmrg/effective_hamiltonian.hpp:124-244
//
// Labels keep the reference's 64-bit packing (it defines the sector ORDER, hence every offset in psi and in
// the operators: SURVEY.md Appendix C); the arithmetic on them is restated field by field.  The symbolic
// expression of H_eff is a flat list of terms (Prod / SumProd) over integer operator ids instead of block2's
// OpExpr tree — the tree belongs to the MPO layer, which is out of scope.  Host-only integer / 9j work; the
// numerics run on the device through BatchGEMMSeq (b2x_host.hpp).
#pragma once
#include "b2x_fpcodec.hpp"
#include <fstream>
#include <istream>
#include <ostream>
#include "b2x_host.hpp"
#include <cstring>
#include <map>

namespace b2xh {

static const uint64_t kInvalidLabel = 0xFFFFFFFFFFFFFFFFULL;

// ---- SZ: (N, 2Sz, point group) ----------------------------------------------------------------------
struct SZ {
    uint64_t data;
    SZ() : data(0) {}
    explicit SZ(uint64_t d) : data(d) {}
    SZ(int n, int twos, int pg)
        : data(((uint64_t)(uint16_t)(int16_t)n << 48) | ((uint64_t)(uint16_t)(int16_t)twos << 16) | (uint64_t)(uint16_t)pg) {}
    int n() const { return (int16_t)(data >> 48); }
    int twos() const { return (int16_t)((data >> 16) & 0xFFFF); }
    int pg() const { return (int)(data & 0xFFFF); }
    int multiplicity() const { return 1; }
    bool is_fermion() const { return twos() & 1; }
    int count() const { return 1; }
    SZ operator[](int) const { return *this; }
    bool operator==(SZ o) const { return data == o.data; }
    bool operator!=(SZ o) const { return data != o.data; }
    bool operator<(SZ o) const { return data < o.data; }
    SZ operator-() const { return SZ(-n(), -twos(), pg()); }
    SZ operator+(SZ o) const { return SZ(n() + o.n(), twos() + o.twos(), pg() ^ o.pg()); }
    SZ operator-(SZ o) const { return *this + (-o); }
    SZ get_ket() const { return *this; }
    SZ get_bra(SZ dq) const { return *this + dq; }
    // dq.combine(bra, ket): the ket label if ket + dq == bra, else invalid
    SZ combine(SZ bra, SZ ket) const { return (ket + *this == bra) ? ket : SZ(kInvalidLabel); }
};

// ---- SU2: (N, 2S_low, 2S, point group); a label may carry a spin RANGE (sum of two labels) or a
// bra/ket PAIR (twos_low = bra spin, twos = ket spin) ------------------------------------------------
struct SU2 {
    uint64_t data;
    SU2() : data(0) {}
    explicit SU2(uint64_t d) : data(d) {}
    SU2(int n, int twos_low, int twos, int pg)
        : data(((uint64_t)(uint16_t)(int16_t)n << 48) | ((uint64_t)(uint16_t)twos_low << 32) |
               ((uint64_t)(uint16_t)twos << 16) | (uint64_t)(uint16_t)pg) {}
    SU2(int n, int twos, int pg) : SU2(n, twos, twos, pg) {}
    int n() const { return (int16_t)(data >> 48); }
    int twos() const { return (int)((data >> 16) & 0xFFFF); }
    int twos_low() const { return (int)((data >> 32) & 0xFFFF); }
    int pg() const { return (int)(data & 0xFFFF); }
    int multiplicity() const { return twos() + 1; }
    bool is_fermion() const { return twos() & 1; }
    int count() const { return ((twos() - twos_low()) >> 1) + 1; }
    SU2 operator[](int i) const { return SU2(n(), twos_low() + 2 * i, twos_low() + 2 * i, pg()); }
    bool operator==(SU2 o) const { return data == o.data; }
    bool operator!=(SU2 o) const { return data != o.data; }
    bool operator<(SU2 o) const { return data < o.data; }
    SU2 operator-() const { return SU2(-n(), twos_low(), twos(), pg()); }
    // coupled label: spins from |2S_a - 2S_b| .. 2S_a + 2S_b
    SU2 operator+(SU2 o) const {
        int lo1 = twos() - o.twos_low(), lo2 = o.twos() - twos_low();
        int lo = (lo1 < 0) ? lo2 : (lo2 < 0 ? lo1 : std::min(lo1, lo2));
        return SU2(n() + o.n(), lo, twos() + o.twos(), pg() ^ o.pg());
    }
    SU2 operator-(SU2 o) const { return *this + (-o); }
    SU2 get_ket() const { return SU2(n(), twos(), twos(), pg()); }
    SU2 get_bra(SU2 dq) const { return SU2(n() + dq.n(), twos_low(), twos_low(), pg() ^ dq.pg()); }
    static bool triangle(int a, int b, int c) { return !((a + b + c) & 1) && c <= a + b && c >= std::abs(a - b); }
    // dq.combine(bra, ket): ket with twos_low := bra spin, if the quantum numbers and the triangle rule allow it
    SU2 combine(SU2 bra, SU2 ket) const {
        SU2 k(ket.n(), bra.twos(), ket.twos(), ket.pg());
        if (k.get_bra(*this) != bra || !triangle(ket.twos(), twos(), bra.twos()))
            return SU2(kInvalidLabel);
        return k;
    }
};

// ---- coupling coefficients ----------------------------------------------------------------------------
template <typename S> struct CG;
template <> struct CG<SZ> {
    double wigner_9j(SZ, SZ, SZ, SZ, SZ, SZ, SZ, SZ, SZ) const { return 1.0; }
    double transpose_cg(SZ, SZ, SZ) const { return 1.0; }
};
template <> struct CG<SU2> {
    std::vector<long double> lf; // log factorials
    explicit CG(int n = 400) : lf(n + 1, 0.0L) {
        for (int i = 1; i <= n; i++)
            lf[i] = lf[i - 1] + logl((long double)i);
    }
    // all arguments are 2j
    long double log_delta(int a, int b, int c) const {
        return 0.5L * (lf[(a + b - c) / 2] + lf[(a - b + c) / 2] + lf[(-a + b + c) / 2] - lf[(a + b + c) / 2 + 1]);
    }
    long double wigner_6j(int a, int b, int c, int d, int e, int f) const {
        if (!SU2::triangle(a, b, c) || !SU2::triangle(a, e, f) || !SU2::triangle(d, b, f) || !SU2::triangle(d, e, c))
            return 0.0L;
        const int a1 = (a + b + c) / 2, a2 = (a + e + f) / 2, a3 = (d + b + f) / 2, a4 = (d + e + c) / 2;
        const int b1 = (a + b + d + e) / 2, b2 = (b + c + e + f) / 2, b3 = (a + c + d + f) / 2;
        const int lo = std::max(std::max(a1, a2), std::max(a3, a4)), hi = std::min(b1, std::min(b2, b3));
        const long double ld = log_delta(a, b, c) + log_delta(a, e, f) + log_delta(d, b, f) + log_delta(d, e, c);
        long double r = 0.0L;
        for (int t = lo; t <= hi; t++) { // Racah's single sum
            long double term = lf[t + 1] - lf[t - a1] - lf[t - a2] - lf[t - a3] - lf[t - a4] - lf[b1 - t] -
                               lf[b2 - t] - lf[b3 - t];
            r += ((t & 1) ? -1.0L : 1.0L) * expl(term + ld);
        }
        return r;
    }
    long double wigner_9j_2j(int a, int b, int c, int d, int e, int f, int g, int h, int i) const {
        if (!SU2::triangle(a, b, c) || !SU2::triangle(d, e, f) || !SU2::triangle(g, h, i) || !SU2::triangle(a, d, g) ||
            !SU2::triangle(b, e, h) || !SU2::triangle(c, f, i))
            return 0.0L;
        const int lo = std::max(std::abs(a - i), std::max(std::abs(h - d), std::abs(b - f)));
        const int hi = std::min(a + i, std::min(h + d, b + f));
        long double r = 0.0L;
        for (int x = lo; x <= hi; x += 2)
            r += (x + 1) * ((x & 1) ? -1.0L : 1.0L) * wigner_6j(a, b, c, f, i, x) * wigner_6j(d, e, f, b, x, h) *
                 wigner_6j(g, h, i, x, a, d);
        return r;
    }
    double wigner_9j(SU2 a, SU2 b, SU2 c, SU2 d, SU2 e, SU2 f, SU2 g, SU2 h, SU2 i) const {
        return (double)wigner_9j_2j(a.twos(), b.twos(), c.twos(), d.twos(), e.twos(), f.twos(), g.twos(), h.twos(),
                                    i.twos());
    }
    // transpose factor of an operator with 2S = td between row / column spins tl / tr
    double transpose_cg(SU2 d, SU2 l, SU2 r) const {
        const int td = d.twos(), tl = l.twos(), tr = r.twos();
        return (double)((((td + tl - tr) & 2) ? -1.0L : 1.0L) * sqrtl((long double)(tr + 1)) /
                        sqrtl((long double)(tl + 1)));
    }
};

// ---- block-sparse matrix description -------------------------------------------------------------------
template <typename S> struct SparseMatrixInfo {
    struct ConnectionInfo {
        std::vector<S> quanta;
        std::vector<uint32_t> idx;
        std::vector<uint64_t> stride; // iv for wavefunction infos; sub-block position for tensor-product infos
        std::vector<double> factor;
        std::vector<uint32_t> ia, ib, ic;
        int n[5] = {0, 0, 0, 0, 0}, nc = 0;
        // all (iv, ia, ib, ic) with non-zero coupling, per operator sub-label (conj flag, combined delta quantum)
        void initialize_wfn(S cdq, S vdq, S opdq, const std::vector<std::pair<uint8_t, S>> &subdq,
                            const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &ainfos,
                            const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &binfos,
                            const std::shared_ptr<SparseMatrixInfo> &cinfo,
                            const std::shared_ptr<SparseMatrixInfo> &vinfo, const CG<S> &cg) {
            quanta.clear(), idx.clear(), stride.clear(), factor.clear(), ia.clear(), ib.clear(), ic.clear();
            if (ainfos.empty() || binfos.empty()) {
                n[4] = nc = 0;
                return;
            }
            for (int i = 0; i < 5; i++)
                n[i] = -1;
            auto find_info = [](const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &infos, S q) {
                auto it = std::lower_bound(infos.begin(), infos.end(), q,
                                           [](const std::pair<S, std::shared_ptr<SparseMatrixInfo>> &p, S x) {
                                               return p.first < x;
                                           });
                if (it == infos.end() || it->first != q)
                    throw std::runtime_error("initialize_wfn: operator delta quantum without an info");
                return it->second;
            };
            struct Ent {
                double f;
                uint64_t iv;
                uint32_t ia, ib, ic;
            };
            for (size_t k = 0; k < subdq.size(); k++) {
                const uint8_t cj = subdq[k].first;
                if (n[cj] == -1)
                    n[cj] = (int)k;
                const bool cja = cj & 1, cjb = (cj & 2) >> 1;
                idx.push_back((uint32_t)stride.size());
                quanta.push_back(subdq[k].second);
                S adq = cja ? -subdq[k].second.get_bra(opdq) : subdq[k].second.get_bra(opdq);
                S bdq = cjb ? subdq[k].second.get_ket() : -subdq[k].second.get_ket();
                auto ainfo = find_info(ainfos, adq), binfo = find_info(binfos, bdq);
                // the reference orders entries by the position ip of the connection inside its psi' sector
                // first, by sector second (sparse_matrix.hpp:172-176, 257-266)
                std::vector<std::vector<Ent>> pv;
                for (int iv = 0; iv < vinfo->n; iv++) {
                    size_t ip = 0;
                    S lq = vinfo->quanta[iv].get_bra(vdq), rq = -vinfo->quanta[iv].get_ket();
                    S rqprimes = cjb ? rq + bdq : rq - bdq;
                    for (int r = 0; r < rqprimes.count(); r++) {
                        S rqprime = rqprimes[r];
                        int jb = binfo->find_state(cjb ? bdq.combine(rqprime, rq) : bdq.combine(rq, rqprime));
                        if (jb == -1)
                            continue;
                        S lqprimes = cdq - rqprime;
                        for (int l = 0; l < lqprimes.count(); l++) {
                            S lqprime = lqprimes[l];
                            int ja = ainfo->find_state(cja ? adq.combine(lqprime, lq) : adq.combine(lq, lqprime));
                            int jc = cinfo->find_state(cdq.combine(lqprime, -rqprime));
                            if (ja == -1 || jc == -1)
                                continue;
                            double f = std::sqrt((double)cdq.multiplicity() * opdq.multiplicity() * lq.multiplicity() *
                                                 rq.multiplicity()) *
                                       cg.wigner_9j(lqprime, rqprime, cdq, adq, bdq, opdq, lq, rq, vdq);
                            f *= (binfo->is_fermion && lqprime.is_fermion()) ? -1 : 1;
                            if (cja)
                                f *= cg.transpose_cg(adq, lq, lqprime);
                            if (cjb)
                                f *= cg.transpose_cg(bdq, rq, rqprime);
                            if (std::fabs(f) >= 1E-20) {
                                if (pv.size() <= ip)
                                    pv.emplace_back();
                                pv[ip].push_back(Ent{f, (uint64_t)iv, (uint32_t)ja, (uint32_t)jb, (uint32_t)jc});
                                ip++;
                            }
                        }
                    }
                }
                for (auto &row : pv)
                    for (auto &e : row) {
                        factor.push_back(e.f), stride.push_back(e.iv);
                        ia.push_back(e.ia), ib.push_back(e.ib), ic.push_back(e.ic);
                    }
            }
            n[4] = (int)subdq.size();
            for (int i = 3; i >= 0; i--)
                if (n[i] == -1)
                    n[i] = n[i + 1];
            nc = (int)stride.size();
        }
        // connections of the DIAGONAL of H_eff: (ia, ib, ic) with a[ia], b[ib] diagonal blocks of psi sector ic
        // (sparse_matrix.hpp:80-159); sub-labels whose operators cannot be diagonal get an empty range
        void initialize_diag(S cdq, S opdq, const std::vector<std::pair<uint8_t, S>> &subdq,
                             const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &ainfos,
                             const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &binfos,
                             const std::shared_ptr<SparseMatrixInfo> &cinfo, const CG<S> &cg) {
            quanta.clear(), idx.clear(), stride.clear(), factor.clear(), ia.clear(), ib.clear(), ic.clear();
            if (ainfos.empty() || binfos.empty()) {
                n[4] = nc = 0;
                return;
            }
            for (int i = 0; i < 5; i++)
                n[i] = -1;
            auto find_info = [](const std::vector<std::pair<S, std::shared_ptr<SparseMatrixInfo>>> &infos, S q) {
                auto it = std::lower_bound(infos.begin(), infos.end(), q,
                                           [](const std::pair<S, std::shared_ptr<SparseMatrixInfo>> &p, S x) {
                                               return p.first < x;
                                           });
                if (it == infos.end() || it->first != q)
                    throw std::runtime_error("initialize_diag: operator delta quantum without an info");
                return it->second;
            };
            for (size_t k = 0; k < subdq.size(); k++) {
                const uint8_t cj = subdq[k].first;
                if (n[cj] == -1)
                    n[cj] = (int)k;
                const bool cja = cj & 1, cjb = (cj & 2) >> 1;
                idx.push_back((uint32_t)ic.size());
                quanta.push_back(subdq[k].second);
                S adq = cja ? -subdq[k].second.get_bra(opdq) : subdq[k].second.get_bra(opdq);
                S bdq = cjb ? subdq[k].second.get_ket() : -subdq[k].second.get_ket();
                if ((adq + bdq)[0] != (adq - adq)[0]) // the pair of operators changes the quantum numbers
                    continue;
                auto ainfo = find_info(ainfos, adq), binfo = find_info(binfos, bdq);
                for (int jc = 0; jc < cinfo->n; jc++) {
                    S aq = cinfo->quanta[jc].get_bra(cdq), bq = -cinfo->quanta[jc].get_ket();
                    int ja = ainfo->find_state(aq), jb = binfo->find_state(bq);
                    if (ja == -1 || jb == -1 || aq != aq.get_bra(adq) || bq != bq.get_bra(bdq))
                        continue;
                    double f = std::sqrt((double)cdq.multiplicity() * opdq.multiplicity() * aq.multiplicity() *
                                         bq.multiplicity()) *
                               cg.wigner_9j(aq, bq, cdq, adq, bdq, opdq, aq, bq, cdq);
                    if (cja)
                        f *= cg.transpose_cg(adq, aq, aq);
                    if (cjb)
                        f *= cg.transpose_cg(bdq, bq, bq);
                    f *= (binfo->is_fermion && aq.is_fermion()) ? -1 : 1;
                    if (std::fabs(f) >= 1E-20)
                        ia.push_back((uint32_t)ja), ib.push_back((uint32_t)jb), ic.push_back((uint32_t)jc),
                            factor.push_back(f), stride.push_back(0);
                }
            }
            n[4] = (int)subdq.size();
            for (int i = 3; i >= 0; i--)
                if (n[i] == -1)
                    n[i] = n[i + 1];
            nc = (int)ic.size();
        }
    };
    std::vector<S> quanta; // sorted ascending by packed value
    std::vector<uint32_t> n_states_bra, n_states_ket, n_states_total;
    S delta_quantum;
    bool is_fermion = false, is_wavefunction = false;
    int n = 0;
    std::shared_ptr<ConnectionInfo> cinfo;
    // on-disk layout of the reference (src/core/sparse_matrix.hpp:511-566, ubond_t = uint16_t): delta quantum (8 B), n
    // (int32), then one uint32 array [quanta: 2 words each | n_states_bra, n_states_ket: uint16 each | n_states_total:
    // uint32 each], then is_fermion and is_wavefunction (1 B each).  Byte-compatible in both directions.
    void load_data(std::istream &ifs) {
        uint64_t dq;
        int32_t nn;
        ifs.read((char *)&dq, sizeof(dq)), ifs.read((char *)&nn, sizeof(nn));
        if (!ifs.good() || nn < 0)
            throw std::runtime_error("SparseMatrixInfo::load_data failed.");
        delta_quantum = S(dq), n = nn;
        std::vector<uint32_t> buf((size_t)n * 4);
        ifs.read((char *)buf.data(), (std::streamsize)(buf.size() * 4));
        quanta.resize(n), n_states_bra.resize(n), n_states_ket.resize(n), n_states_total.resize(n);
        const uint16_t *bk = (const uint16_t *)(buf.data() + (size_t)n * 2);
        for (int i = 0; i < n; i++) {
            quanta[i] = S((uint64_t)buf[2 * i] | ((uint64_t)buf[2 * i + 1] << 32));
            n_states_bra[i] = bk[i], n_states_ket[i] = bk[n + i];
            n_states_total[i] = buf[(size_t)n * 3 + i];
        }
        uint8_t f, w;
        ifs.read((char *)&f, 1), ifs.read((char *)&w, 1);
        if (ifs.fail())
            throw std::runtime_error("SparseMatrixInfo::load_data failed.");
        is_fermion = f != 0, is_wavefunction = w != 0;
        cinfo = nullptr;
    }
    void save_data(std::ostream &ofs) const {
        uint64_t dq = delta_quantum.data;
        int32_t nn = n;
        ofs.write((const char *)&dq, sizeof(dq)), ofs.write((const char *)&nn, sizeof(nn));
        std::vector<uint32_t> buf((size_t)n * 4, 0);
        uint16_t *bk = (uint16_t *)(buf.data() + (size_t)n * 2);
        for (int i = 0; i < n; i++) {
            buf[2 * i] = (uint32_t)(quanta[i].data & 0xFFFFFFFFu), buf[2 * i + 1] = (uint32_t)(quanta[i].data >> 32);
            if (n_states_bra[i] > 0xFFFF || n_states_ket[i] > 0xFFFF)
                throw std::runtime_error("SparseMatrixInfo::save_data: bond dimension exceeds ubond_t (uint16)");
            bk[i] = (uint16_t)n_states_bra[i], bk[n + i] = (uint16_t)n_states_ket[i];
            buf[(size_t)n * 3 + i] = n_states_total[i];
        }
        ofs.write((const char *)buf.data(), (std::streamsize)(buf.size() * 4));
        uint8_t f = is_fermion, w = is_wavefunction;
        ofs.write((const char *)&f, 1), ofs.write((const char *)&w, 1);
    }
    int find_state(S q) const {
        auto it = std::lower_bound(quanta.begin(), quanta.end(), q);
        return (it == quanta.end() || *it != q) ? -1 : (int)(it - quanta.begin());
    }
    size_t get_total_memory() const {
        return n == 0 ? 0 : (size_t)n_states_total[n - 1] + (size_t)n_states_bra[n - 1] * n_states_ket[n - 1];
    }
};

template <typename S> struct SparseMatrix {
    std::shared_ptr<SparseMatrixInfo<S>> info;
    double *data = nullptr;
    double factor = 1.0;
    size_t total_memory = 0;
    std::vector<double> storage; // owns the data of a matrix loaded from disk
    // on-disk layout of the reference (src/core/sparse_matrix.hpp:896-971): [info if load_info] factor (f64),
    // total_memory (size_t), data; in compressed storage: factor, SIZE_MAX, total_memory, FPCodec array (b2x_fpcodec.hpp)
    void load_data(const std::string &filename, bool load_info = false) {
        std::ifstream ifs(filename.c_str(), std::ios::binary);
        if (!ifs.good())
            throw std::runtime_error("SparseMatrix:load_data on '" + filename + "' failed.");
        if (load_info) {
            info = std::make_shared<SparseMatrixInfo<S>>();
            info->load_data(ifs);
        }
        uint64_t tm;
        ifs.read((char *)&factor, sizeof(factor)), ifs.read((char *)&tm, sizeof(tm));
        const bool coded = tm == ~(uint64_t)0;
        if (coded)
            ifs.read((char *)&tm, sizeof(tm));
        total_memory = (size_t)tm;
        storage.resize(total_memory);
        if (coded)
            FPCodec::read_array(ifs, storage.data(), total_memory);
        else
            ifs.read((char *)storage.data(), (std::streamsize)(total_memory * sizeof(double)));
        if (ifs.fail() || ifs.bad())
            throw std::runtime_error("SparseMatrix:load_data on '" + filename + "' failed.");
        data = storage.data();
    }
    // codec != nullptr: compressed storage (frame->compressed_sparse_tensor_storage with frame->fp_codec in the reference)
    void save_data(const std::string &filename, bool save_info = false, const FPCodec *codec = nullptr) const {
        std::ofstream ofs(filename.c_str(), std::ios::binary);
        if (!ofs.good())
            throw std::runtime_error("SparseMatrix:save_data on '" + filename + "' failed.");
        if (save_info)
            info->save_data(ofs);
        uint64_t tm = total_memory;
        const uint64_t flag = ~(uint64_t)0;
        ofs.write((const char *)&factor, sizeof(factor));
        if (codec)
            ofs.write((const char *)&flag, sizeof(flag));
        ofs.write((const char *)&tm, sizeof(tm));
        if (codec)
            codec->write_array(ofs, data, total_memory);
        else
            ofs.write((const char *)data, (std::streamsize)(total_memory * sizeof(double)));
        if (!ofs.good())
            throw std::runtime_error("SparseMatrix:save_data on '" + filename + "' failed.");
    }
    GMatrix operator[](int i) const {
        return GMatrix(data + info->n_states_total[i], (int)info->n_states_bra[i], (int)info->n_states_ket[i]);
    }
};

enum struct OperatorTensorTypes : uint8_t { Normal, Delayed };

// operator symbols are integer ids into `ops`
template <typename S> struct OperatorTensor {
    OperatorTensorTypes type = OperatorTensorTypes::Normal;
    std::vector<std::shared_ptr<SparseMatrix<S>>> ops;
    // for a delayed (not yet contracted) enlarged block: the block operators and the site operators
    std::shared_ptr<OperatorTensor> lopt, ropt;
    OperatorTensorTypes get_type() const { return type; }
};

enum struct OpTypes : uint8_t { Prod = 0, SumProd = 1 };
// one term of H_eff:  factor * a (x) b          (Prod:    a in lopt, b in ropt)
//                     factor * (d0 (x) d1) (x) b or a (x) (d0 (x) d1)   (SumProd: the delayed factor expanded)
struct OpTerm {
    OpTypes type = OpTypes::Prod;
    double factor = 1.0;
    uint8_t conj = 0;
    int a = -1, b = -1;
    int d0 = -1, d1 = -1;
    uint8_t dconj = 0;
};

// which side of op(a) (x) op(b) is the identity and is dropped (perturbative noise; operator_functions.hpp:48)
enum struct TraceTypes : uint8_t { None = 0, Left = 1, Right = 2 };

template <typename S> struct OperatorFunctions {
    std::shared_ptr<BatchGEMMSeq> seq;
    CG<S> cg;
    explicit OperatorFunctions(const std::shared_ptr<BatchGEMMSeq> &seq) : seq(seq) {}
    // c += scale * factor * op(a[ia]) (x) op(b[ib]) for every connection of the matching sub-label (blocking; reference
    // operator_functions.hpp:672-711).  The connection info of c (ConnectionInfo::initialize_tp, built by the reference's
    // Partition layer) lists (ia, ib, ic, stride, factor): block ic of the enlarged operator receives the Kronecker
    // product in the window that starts `stride` elements after its first element.
    void tensor_product(uint8_t conj, const SparseMatrix<S> &a, const SparseMatrix<S> &b, const SparseMatrix<S> &c,
                        double scale = 1.0) const {
        scale = scale * a.factor * b.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        S adq = a.info->delta_quantum, bdq = b.info->delta_quantum, cdq = c.info->delta_quantum;
        if (!c.info->cinfo)
            throw std::runtime_error("tensor_product: missing connection info");
        const auto &ci = *c.info->cinfo;
        S abdq = cdq.combine((conj & 1) ? -adq : adq, (conj & 2) ? bdq : -bdq);
        int ik = (int)(std::lower_bound(ci.quanta.begin() + ci.n[conj], ci.quanta.begin() + ci.n[conj + 1], abdq) -
                       ci.quanta.begin());
        if (ik >= ci.n[conj + 1] || ci.quanta[ik] != abdq)
            throw std::runtime_error("tensor_product: sub-label not in the connection info (conj " + std::to_string((int)conj) +
                                     ", a.dq " + std::to_string(adq.data) + ", b.dq " + std::to_string(bdq.data) + ", c.dq " +
                                     std::to_string(cdq.data) + ", sub-labels of this conj: " +
                                     std::to_string(ci.n[conj + 1] - ci.n[conj]) + ")");
        int ixa = (int)ci.idx[ik], ixb = ik == ci.n[4] - 1 ? ci.nc : (int)ci.idx[ik + 1];
        for (int il = ixa; il < ixb; il++)
            seq->tensor_product(a[(int)ci.ia[il]], conj & 1, b[(int)ci.ib[il]], (conj & 2) >> 1, c[(int)ci.ic[il]],
                                scale * ci.factor[il], ci.stride[il]);
    }
    // a += scale * op(b) sector by sector (OperatorFunctions::iadd, operator_functions.hpp:135-174): conj reads the
    // transposed block of b and carries the transposition factor of the coupling (the sums of operators that
    // TensorFunctions::numerical_transform forms at the NC -> CN switch of the conventional MPO)
    void iadd(const SparseMatrix<S> &a, const SparseMatrix<S> &b, double scale = 1.0, bool conj = false) const {
        if (std::fabs(b.factor * scale) < 1E-20)
            return;
        const S adq = a.info->delta_quantum, bdq = b.info->delta_quantum;
        for (int ia = 0; ia < a.info->n; ia++) {
            const S bra = a.info->quanta[ia].get_bra(adq), ket = a.info->quanta[ia].get_ket();
            const S bq = conj ? bdq.combine(ket, bra) : bdq.combine(bra, ket);
            if (bq.data == kInvalidLabel)
                continue;
            const int ib = b.info->find_state(bq);
            if (ib < 0)
                continue;
            double factor = scale * b.factor;
            if (conj)
                factor *= cg.transpose_cg(bdq, bra, ket);
            seq->iadd(a[ia], b[ib], factor, conj, 1.0);
        }
    }
    // c[ic] += scale * op(rot_bra[cq]) a[ia] op(rot_ket[cq']) for every sector of c (operator_functions.hpp:175-210):
    // a is the operator in the enlarged basis (more sectors than c), the MPS tensor blocks are looked up by the bra /
    // ket labels of the c sector; trans = false: bra^T . a . ket (left blocks), true: bra . a . ket^T (right blocks)
    void tensor_rotate(const SparseMatrix<S> &a, const SparseMatrix<S> &c, const SparseMatrix<S> &rot_bra,
                       const SparseMatrix<S> &rot_ket, bool trans, double scale = 1.0) const {
        scale = scale * a.factor * rot_bra.factor * rot_ket.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        S adq = a.info->delta_quantum, cdq = c.info->delta_quantum;
        if (adq != cdq || a.info->n < c.info->n)
            throw std::runtime_error("tensor_rotate: operator infos do not match");
        for (int ic = 0, ia = 0; ic < c.info->n; ia++, ic++) {
            while (ia < a.info->n && a.info->quanta[ia] != c.info->quanta[ic])
                ia++;
            if (ia >= a.info->n)
                throw std::runtime_error("tensor_rotate: sector of c missing in a");
            S cq = c.info->quanta[ic].get_bra(cdq), cqprime = c.info->quanta[ic].get_ket();
            int ibra = rot_bra.info->find_state(cq), iket = rot_ket.info->find_state(cqprime);
            if (ibra < 0 || iket < 0)
                throw std::runtime_error("tensor_rotate: MPS tensor block not found");
            seq->rotate(a[ia], c[ic], rot_bra[ibra], (uint8_t)((int)!trans | 2), rot_ket[iket], (uint8_t)trans, scale);
        }
    }
    // v[iv] += scale * factor * op(a[ia]) c[ic] op(b[ib])^T for every connection of the matching sub-label
    void tensor_product_multiply(uint8_t conj, const SparseMatrix<S> &a, const SparseMatrix<S> &b,
                                 const SparseMatrix<S> &c, const SparseMatrix<S> &v, S opdq, double scale = 1.0,
                                 TraceTypes tt = TraceTypes::None) const {
        scale = scale * a.factor * b.factor * c.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        S adq = a.info->delta_quantum, bdq = b.info->delta_quantum;
        if (!c.info->cinfo)
            throw std::runtime_error("tensor_product_multiply: missing connection info");
        const auto &ci = *c.info->cinfo;
        S abdq = opdq.combine((conj & 1) ? -adq : adq, (conj & 2) ? bdq : -bdq);
        int ik = (int)(std::lower_bound(ci.quanta.begin() + ci.n[conj], ci.quanta.begin() + ci.n[conj + 1], abdq) -
                       ci.quanta.begin());
        if (ik >= ci.n[conj + 1] || ci.quanta[ik] != abdq)
            throw std::runtime_error("tensor_product_multiply: sub-label not in the connection info");
        int ixa = (int)ci.idx[ik], ixb = ik == ci.n[4] - 1 ? ci.nc : (int)ci.idx[ik + 1];
        for (int il = ixa; il < ixb; il++) {
            const GMatrix cm = c[(int)ci.ic[il]], vm = v[(int)ci.stride[il]];
            if (tt == TraceTypes::None)
                seq->rotate(cm, vm, a[(int)ci.ia[il]], (conj & 1) ? 3 : 0, b[(int)ci.ib[il]], (conj & 2) ? 2 : 1,
                            scale * ci.factor[il]);
            else if (tt == TraceTypes::Left) // v += c . op(b)^T   (operator_functions.hpp:518-526)
                seq->multiply(cm, false, b[(int)ci.ib[il]], (conj & 2) ? 2 : 1, vm, scale * ci.factor[il], 1.0);
            else // v += op(a) . c   (:527-535)
                seq->multiply(a[(int)ci.ia[il]], (conj & 1) ? 3 : 0, cm, false, vm, scale * ci.factor[il], 1.0);
        }
    }
    // diag(c)[ic] += scale * factor * diag(a[ia]) (x) diag(b[ib])       (operator_functions.hpp:211-245)
    void tensor_product_diagonal(uint8_t conj, const SparseMatrix<S> &a, const SparseMatrix<S> &b,
                                 const SparseMatrix<S> &c, S opdq, double scale = 1.0) const {
        scale = scale * a.factor * b.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        S adq = a.info->delta_quantum, bdq = b.info->delta_quantum;
        const auto &ci = *c.info->cinfo;
        S abdq = opdq.combine((conj & 1) ? -adq : adq, (conj & 2) ? bdq : -bdq);
        int ik = (int)(std::lower_bound(ci.quanta.begin() + ci.n[conj], ci.quanta.begin() + ci.n[conj + 1], abdq) -
                       ci.quanta.begin());
        if (ik >= ci.n[conj + 1] || ci.quanta[ik] != abdq)
            throw std::runtime_error("tensor_product_diagonal: sub-label not in the connection info");
        int ixa = (int)ci.idx[ik], ixb = ik == ci.n[4] - 1 ? ci.nc : (int)ci.idx[ik + 1];
        for (int il = ixa; il < ixb; il++)
            seq->tensor_product_diagonal(conj, a[(int)ci.ia[il]], b[(int)ci.ib[il]], c[(int)ci.ic[il]],
                                         scale * ci.factor[il]);
    }
    // the delayed variant (operator_functions.hpp:246-328)
    void three_tensor_product_diagonal(uint8_t conj, const SparseMatrix<S> &a, const SparseMatrix<S> &b,
                                       const SparseMatrix<S> &c, uint8_t dconj, const SparseMatrix<S> &da,
                                       const SparseMatrix<S> &db, bool dleft, S opdq, double scale = 1.0) const {
        scale = scale * a.factor * b.factor * da.factor * db.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        const SparseMatrix<S> &dc = dleft ? a : b;
        S adq = a.info->delta_quantum, bdq = b.info->delta_quantum;
        S abdq = opdq.combine((conj & 1) ? -adq : adq, (conj & 2) ? bdq : -bdq);
        S dadq = da.info->delta_quantum, dbdq = db.info->delta_quantum, dcdq = dc.info->delta_quantum;
        S dabdq = dcdq.combine((dconj & 1) ? -dadq : dadq, (dconj & 2) ? dbdq : -dbdq);
        if (!c.info->cinfo || !dc.info->cinfo)
            throw std::runtime_error("three_tensor_product_diagonal: missing connection info");
        const auto &ci = *c.info->cinfo, &di = *dc.info->cinfo;
        int ik = (int)(std::lower_bound(ci.quanta.begin() + ci.n[conj], ci.quanta.begin() + ci.n[conj + 1], abdq) -
                       ci.quanta.begin());
        if (ik >= ci.n[conj + 1] || ci.quanta[ik] != abdq)
            throw std::runtime_error("three_tensor_product_diagonal: sub-label not in the connection info");
        int ixa = (int)ci.idx[ik], ixb = ik == ci.n[4] - 1 ? ci.nc : (int)ci.idx[ik + 1];
        int idk = (int)(std::lower_bound(di.quanta.begin() + di.n[dconj], di.quanta.begin() + di.n[dconj + 1], dabdq) -
                        di.quanta.begin());
        if (idk >= di.n[dconj + 1] || di.quanta[idk] != dabdq)
            throw std::runtime_error("three_tensor_product_diagonal: delayed sub-label not in its connection info");
        int idxa = (int)di.idx[idk], idxb = idk == di.n[4] - 1 ? di.nc : (int)di.idx[idk + 1];
        for (int il = ixa; il < ixb; il++) {
            int ja = (int)ci.ia[il], jb = (int)ci.ib[il], jc = (int)ci.ic[il];
            uint32_t idc = (uint32_t)(dleft ? ja : jb);
            int idl = (int)(std::lower_bound(di.ic.begin() + idxa, di.ic.begin() + idxb, idc) - di.ic.begin());
            for (; idl < idxb && di.ic[idl] == idc; idl++)
                seq->three_tensor_product_diagonal(conj, a[ja], b[jb], c[jc], da[(int)di.ia[idl]], dconj & 1,
                                                   db[(int)di.ib[idl]], (dconj & 2) >> 1, dleft,
                                                   scale * ci.factor[il] * di.factor[idl], di.stride[idl]);
        }
    }
    // same with the left (dleft) or right operator still delayed as da (x) db: nested connection lists
    void three_tensor_product_multiply(uint8_t conj, const SparseMatrix<S> &a, const SparseMatrix<S> &b,
                                       const SparseMatrix<S> &c, const SparseMatrix<S> &v, uint8_t dconj,
                                       const SparseMatrix<S> &da, const SparseMatrix<S> &db, bool dleft, S opdq,
                                       double scale = 1.0, TraceTypes tt = TraceTypes::None) const {
        scale = scale * a.factor * b.factor * c.factor * da.factor * db.factor;
        if (std::fabs(scale) < 1E-20)
            return;
        const SparseMatrix<S> &dc = dleft ? a : b;
        S adq = a.info->delta_quantum, bdq = b.info->delta_quantum;
        S abdq = opdq.combine((conj & 1) ? -adq : adq, (conj & 2) ? bdq : -bdq);
        S dadq = da.info->delta_quantum, dbdq = db.info->delta_quantum, dcdq = dc.info->delta_quantum;
        S dabdq = dcdq.combine((dconj & 1) ? -dadq : dadq, (dconj & 2) ? dbdq : -dbdq);
        if (!c.info->cinfo || !dc.info->cinfo)
            throw std::runtime_error("three_tensor_product_multiply: missing connection info");
        const auto &ci = *c.info->cinfo, &di = *dc.info->cinfo;
        int ik = (int)(std::lower_bound(ci.quanta.begin() + ci.n[conj], ci.quanta.begin() + ci.n[conj + 1], abdq) -
                       ci.quanta.begin());
        if (ik >= ci.n[conj + 1] || ci.quanta[ik] != abdq)
            throw std::runtime_error("three_tensor_product_multiply: sub-label not in the connection info");
        int ixa = (int)ci.idx[ik], ixb = ik == ci.n[4] - 1 ? ci.nc : (int)ci.idx[ik + 1];
        int idk = (int)(std::lower_bound(di.quanta.begin() + di.n[dconj], di.quanta.begin() + di.n[dconj + 1], dabdq) -
                        di.quanta.begin());
        if (idk >= di.n[dconj + 1] || di.quanta[idk] != dabdq)
            throw std::runtime_error("three_tensor_product_multiply: delayed sub-label not in its connection info");
        int idxa = (int)di.idx[idk], idxb = idk == di.n[4] - 1 ? di.nc : (int)di.idx[idk + 1];
        for (int il = ixa; il < ixb; il++) {
            int ja = (int)ci.ia[il], jb = (int)ci.ib[il], jc = (int)ci.ic[il], jv = (int)ci.stride[il];
            uint32_t idc = (uint32_t)(dleft ? ja : jb);
            int idl = (int)(std::lower_bound(di.ic.begin() + idxa, di.ic.begin() + idxb, idc) - di.ic.begin());
            for (; idl < idxb && di.ic[idl] == idc; idl++) {
                const double f = scale * ci.factor[il] * di.factor[idl];
                const GMatrix dam = da[(int)di.ia[idl]], dbm = db[(int)di.ib[idl]];
                if (tt == TraceTypes::None)
                    seq->three_rotate(c[jc], v[jv], a[ja], conj & 1, b[jb], !(conj & 2), dam, dconj & 1, dbm,
                                      (dconj & 2) >> 1, dleft, f, di.stride[idl]);
                else if (tt == TraceTypes::Left)
                    seq->three_rotate_tr_left(c[jc], v[jv], a[ja], conj & 1, b[jb], !(conj & 2), dam, dconj & 1, dbm,
                                              (dconj & 2) >> 1, dleft, f, di.stride[idl]);
                else
                    seq->three_rotate_tr_right(c[jc], v[jv], a[ja], conj & 1, b[jb], !(conj & 2), dam, dconj & 1, dbm,
                                               (dconj & 2) >> 1, dleft, f, di.stride[idl]);
            }
        }
    }
};

template <typename S> struct TensorFunctions {
    std::shared_ptr<OperatorFunctions<S>> opf;
    explicit TensorFunctions(const std::shared_ptr<OperatorFunctions<S>> &opf) : opf(opf) {}
    // mat += eval(expr): a sum of products  factor * lop[a] (x) rop[b]  (tensor_functions.hpp:2185-2286; a SumProd term
    // whose operator sum exists as an intermediate is the product with that intermediate)
    void tensor_product(const std::vector<OpTerm> &expr, const OperatorTensor<S> &lop, const OperatorTensor<S> &rop,
                        const SparseMatrix<S> &mat) const {
        for (const OpTerm &t : expr) {
            if (t.a < 0 || t.b < 0 || t.a >= (int)lop.ops.size() || t.b >= (int)rop.ops.size() || !lop.ops[t.a] ||
                !rop.ops[t.b])
                throw std::runtime_error("tensor_product: term refers to an unknown operator");
            opf->tensor_product(t.conj, *lop.ops[t.a], *rop.ops[t.b], mat, t.factor);
        }
    }
    // c[i] = eval(exprs[i]) over (a, b) = (block operators, site operators): left_contract (tensor_functions.hpp:
    // 2842-2885); right_contract (:2941-2983) passes the site operators as the left factor.  Records the block products;
    // BatchGEMMSeq::outer_perform executes them.
    void contract(const OperatorTensor<S> &lop, const OperatorTensor<S> &rop, const OperatorTensor<S> &c,
                  const std::vector<std::vector<OpTerm>> &exprs) const {
        if (exprs.size() != c.ops.size())
            throw std::runtime_error("contract: one expression per enlarged operator expected");
        for (size_t i = 0; i < c.ops.size(); i++)
            if (c.ops[i])
                tensor_product(exprs[i], lop, rop, *c.ops[i]);
    }
    // c = mpst_bra^T x a x mpst_ket for every operator of the enlarged block (tensor_functions.hpp:2365-2383);
    // right_rotate: c = mpst_bra x a x mpst_ket^T (:2385-2403).  Operators absent from a (null) are skipped.  The pairs
    // are recorded; BatchGEMMSeq::rotate_perform executes them.
    void rotate(const OperatorTensor<S> &a, const SparseMatrix<S> &mpst_bra, const SparseMatrix<S> &mpst_ket,
                const OperatorTensor<S> &c, bool right) const {
        if (a.ops.size() != c.ops.size())
            throw std::runtime_error("rotate: operator tensors differ in size");
        for (size_t i = 0; i < a.ops.size(); i++)
            if (a.ops[i] && c.ops[i])
                opf->tensor_rotate(*a.ops[i], *c.ops[i], mpst_bra, mpst_ket, right);
    }
    void left_rotate(const OperatorTensor<S> &a, const SparseMatrix<S> &mpst_bra, const SparseMatrix<S> &mpst_ket,
                     const OperatorTensor<S> &c) const {
        rotate(a, mpst_bra, mpst_ket, c, false);
    }
    void right_rotate(const OperatorTensor<S> &a, const SparseMatrix<S> &mpst_bra, const SparseMatrix<S> &mpst_ket,
                      const OperatorTensor<S> &c) const {
        rotate(a, mpst_bra, mpst_ket, c, true);
    }
    // vmat += expr x cmat: walks the sum of terms (the reference fans this out over threads and merges the
    // per-thread plans; the recorded order is the same)
    void tensor_product_multiply(const std::vector<OpTerm> &expr, const OperatorTensor<S> &lopt,
                                 const OperatorTensor<S> &ropt, const SparseMatrix<S> &cmat,
                                 const SparseMatrix<S> &vmat, S opdq) const {
        for (const OpTerm &t : expr) {
            if (t.a < 0 || t.b < 0 || t.a >= (int)lopt.ops.size() || t.b >= (int)ropt.ops.size())
                throw std::runtime_error("tensor_product_multiply: term refers to an unknown operator");
            if (t.type == OpTypes::SumProd) {
                const bool dleft = lopt.get_type() == OperatorTensorTypes::Delayed;
                const OperatorTensor<S> &dopt = dleft ? lopt : ropt;
                if (dopt.get_type() != OperatorTensorTypes::Delayed || !dopt.lopt || !dopt.ropt)
                    throw std::runtime_error("tensor_product_multiply: SumProd term without a delayed operator tensor");
                opf->three_tensor_product_multiply(t.conj, *lopt.ops[t.a], *ropt.ops[t.b], cmat, vmat, t.dconj,
                                                   *dopt.lopt->ops[t.d0], *dopt.ropt->ops[t.d1], dleft, opdq,
                                                   t.factor);
            } else
                opf->tensor_product_multiply(t.conj, *lopt.ops[t.a], *ropt.ops[t.b], cmat, vmat, opdq, t.factor);
        }
    }
    // positions of the identity operator in lopt / ropt and in the two factors of the delayed tensor (-1 = absent)
    struct IdentityOps {
        int l = -1, r = -1, dl = -1, dr = -1;
    };
    // vmats += (expr with one side traced out) x cmat: the perturbative-noise walk (tensor_functions.hpp:366-803, real
    // double, no stacked MPO, "reduced" noise: one perturbed wavefunction per target label).  For every term only the
    // operator of the kept side acts on cmat; the other side must be the identity.  cinfos[j][k] is the connection
    // info of (psubsl[j], k-th label of cmat.dq + psubsl[j].label), vdqs the sorted target labels of vmats.
    void tensor_product_partial_multiply(const std::vector<OpTerm> &expr, const OperatorTensor<S> &lopt,
                                         const OperatorTensor<S> &ropt, bool trace_right, const SparseMatrix<S> &cmat,
                                         const std::vector<std::pair<uint8_t, S>> &psubsl,
                                         const std::vector<std::vector<std::shared_ptr<typename SparseMatrixInfo<S>::ConnectionInfo>>> &cinfos,
                                         const std::vector<S> &vdqs, const std::vector<SparseMatrix<S>> &vmats,
                                         const IdentityOps &id) const {
        // no identity on the traced side: the site is not optimised, the noise is skipped (:383-391)
        if ((!trace_right && id.l < 0) || (trace_right && id.r < 0))
            return;
        const bool ldel = lopt.get_type() == OperatorTensorTypes::Delayed, rdel = ropt.get_type() == OperatorTensorTypes::Delayed;
        const TraceTypes tt = trace_right ? TraceTypes::Right : TraceTypes::Left;
        for (const OpTerm &t : expr) {
            const SparseMatrix<S> *dlmat = nullptr, *drmat = nullptr;
            uint8_t dconj = 0;
            const bool dleft = ldel; // which tensor is delayed (at most one is)
            if (ldel || rdel) {
                const OperatorTensor<S> &dopt = ldel ? lopt : ropt;
                if (t.type == OpTypes::SumProd && dleft == trace_right) { // the kept operator is the delayed product
                    dlmat = dopt.lopt->ops[t.d0].get(), drmat = dopt.ropt->ops[t.d1].get(), dconj = t.dconj;
                } else if (dleft != trace_right) { // the traced identity lives in the delayed tensor
                    const auto &iop = trace_right ? ropt.ops[id.r] : lopt.ops[id.l];
                    if (iop->data == nullptr) {
                        if (id.dl < 0 || id.dr < 0)
                            throw std::runtime_error("partial multiply: delayed identity factors missing");
                        dlmat = dopt.lopt->ops[id.dl].get(), drmat = dopt.ropt->ops[id.dr].get();
                    }
                }
            }
            const SparseMatrix<S> &lmat = trace_right ? *lopt.ops[t.a] : *lopt.ops[id.l];
            const SparseMatrix<S> &rmat = trace_right ? *ropt.ops[id.r] : *ropt.ops[t.b];
            const uint8_t cj = trace_right ? (t.conj & 1) : (t.conj & 2);
            const S q = trace_right ? lmat.info->delta_quantum : rmat.info->delta_quantum;
            const S opdq = cj ? -q : q;
            const S pks = cmat.info->delta_quantum + opdq;
            const std::pair<uint8_t, S> key((uint8_t)(cj ? 1 : 0), opdq);
            const int ij = (int)(std::lower_bound(psubsl.begin(), psubsl.end(), key,
                                                  [](const std::pair<uint8_t, S> &x, const std::pair<uint8_t, S> &y) {
                                                      return x.first != y.first ? x.first < y.first : x.second < y.second;
                                                  }) -
                                 psubsl.begin());
            if (ij >= (int)psubsl.size() || psubsl[ij].first != key.first || psubsl[ij].second != key.second)
                throw std::runtime_error("partial multiply: operator sub-label not in psubsl");
            for (int k = 0; k < pks.count(); k++) {
                const int iv = (int)(std::lower_bound(vdqs.begin(), vdqs.end(), pks[k]) - vdqs.begin());
                if (iv >= (int)vdqs.size() || vdqs[iv] != pks[k])
                    throw std::runtime_error("partial multiply: target label not in vdqs");
                SparseMatrix<S> cm = cmat; // the wavefunction seen through this (sub-label, target) connection info
                auto cinfo_holder = std::make_shared<SparseMatrixInfo<S>>(*cmat.info);
                cinfo_holder->cinfo = cinfos[ij][k];
                cm.info = cinfo_holder;
                if (dlmat != nullptr)
                    opf->three_tensor_product_multiply(cj, lmat, rmat, cm, vmats[iv], dconj, *dlmat, *drmat, dleft, opdq,
                                                       t.factor, tt);
                else
                    opf->tensor_product_multiply(cj, lmat, rmat, cm, vmats[iv], opdq, t.factor, tt);
            }
        }
    }
    // diag(mat) += diagonal of expr   (tensor_functions.hpp:2027-2182)
    void tensor_product_diagonal(const std::vector<OpTerm> &expr, const OperatorTensor<S> &lopt,
                                 const OperatorTensor<S> &ropt, const SparseMatrix<S> &mat, S opdq) const {
        for (const OpTerm &t : expr) {
            if (t.type == OpTypes::SumProd) {
                const bool dleft = lopt.get_type() == OperatorTensorTypes::Delayed;
                const OperatorTensor<S> &dopt = dleft ? lopt : ropt;
                opf->three_tensor_product_diagonal(t.conj, *lopt.ops[t.a], *ropt.ops[t.b], mat, t.dconj,
                                                   *dopt.lopt->ops[t.d0], *dopt.ropt->ops[t.d1], dleft, opdq,
                                                   t.factor);
            } else
                opf->tensor_product_diagonal(t.conj, *lopt.ops[t.a], *ropt.ops[t.b], mat, opdq, t.factor);
        }
    }
    // c += scale * H b : replay of the recorded plan (tensor_functions.hpp:59-62)
    void operator()(const GMatrix &b, const GMatrix &c, double scale = 1.0) { (*opf->seq)(b, c, scale); }
};

// Builds the wavefunction connection info and the plan of one site from the symbolic description, then
// behaves as EffectiveHamiltonian (b2x_host.hpp) for operator() / eigs.
template <typename S> struct SymbolicEffectiveHamiltonian {
    typedef SparseMatrixInfo<S> Info;
    std::vector<std::pair<S, std::shared_ptr<Info>>> left_op_infos, right_op_infos;
    std::shared_ptr<OperatorTensor<S>> lopt, ropt;
    std::vector<OpTerm> expr;
    std::shared_ptr<Info> ket_info, bra_info;
    S opdq;
    std::vector<std::pair<uint8_t, S>> subdq;
    std::shared_ptr<typename Info::ConnectionInfo> wfn_info;
    std::shared_ptr<TensorFunctions<S>> tf;
    std::vector<double> diag;
    SymbolicEffectiveHamiltonian(const std::vector<std::pair<S, std::shared_ptr<Info>>> &linfos,
                                 const std::vector<std::pair<S, std::shared_ptr<Info>>> &rinfos,
                                 const std::shared_ptr<OperatorTensor<S>> &lopt,
                                 const std::shared_ptr<OperatorTensor<S>> &ropt, const std::vector<OpTerm> &expr,
                                 const std::shared_ptr<Info> &ket_info, const std::shared_ptr<Info> &bra_info, S opdq,
                                 const std::vector<std::pair<uint8_t, S>> &subdq, const std::vector<double> &diag)
        : left_op_infos(linfos), right_op_infos(rinfos), lopt(lopt), ropt(ropt), expr(expr), ket_info(ket_info),
          bra_info(bra_info), opdq(opdq), subdq(subdq), diag(diag) {
        auto seq = std::make_shared<BatchGEMMSeq>();
        tf = std::make_shared<TensorFunctions<S>>(std::make_shared<OperatorFunctions<S>>(seq));
        wfn_info = std::make_shared<typename Info::ConnectionInfo>();
        wfn_info->initialize_wfn(ket_info->delta_quantum, bra_info->delta_quantum, opdq, subdq, left_op_infos,
                                 right_op_infos, ket_info, bra_info, tf->opf->cg);
        ket_info->cinfo = wfn_info;
    }
    // diagonal of H_eff as the reference's constructor builds it (effective_hamiltonian.hpp:189-200):
    // initialize_diag + tensor_product_diagonal, evaluated on the device
    std::vector<double> compute_diag(const std::vector<std::pair<uint8_t, S>> &subdq) {
        auto dinfo = std::make_shared<Info>(*ket_info);
        dinfo->cinfo = std::make_shared<typename Info::ConnectionInfo>();
        dinfo->cinfo->initialize_diag(ket_info->delta_quantum, opdq, subdq, left_op_infos, right_op_infos, dinfo,
                                      tf->opf->cg);
        SparseMatrix<S> dmat;
        dmat.info = dinfo, dmat.data = (double *)0, dmat.factor = 1.0;
        tf->tensor_product_diagonal(expr, *lopt, *ropt, dmat, opdq);
        std::vector<double> d(ket_info->get_total_memory(), 0.0);
        tf->opf->seq->diag_perform(d.data(), d.size());
        return d;
    }
    // Perturbative noise (effective_hamiltonian.hpp:252-423, NoiseTypes::ReducedPerturbative): one perturbed
    // wavefunction per target label vdqs[i] with info vinfos[i]; builds the connection infos of every
    // (operator sub-label, target) pair with initialize_wfn (:353-373) and records the single-GEMM list of the partial
    // multiply into tf->opf->seq.  ket / vdata are the absolute host buffers of psi and of the perturbed
    // wavefunctions (vmats[i] starts at vdata + voffs[i]); execute with seq->auto_perform(v, ket).
    void record_perturbative_noise(bool trace_right, const std::vector<std::pair<uint8_t, S>> &psubsl,
                                   const std::vector<S> &vdqs, const std::vector<std::shared_ptr<Info>> &vinfos,
                                   const std::vector<uint64_t> &voffs, S vacuum,
                                   const typename TensorFunctions<S>::IdentityOps &id, double *ket, double *vdata) {
        const S ket_label = ket_info->delta_quantum, idq = vacuum;
        std::vector<std::vector<std::shared_ptr<typename Info::ConnectionInfo>>> cinfos(psubsl.size());
        for (size_t j = 0; j < psubsl.size(); j++) {
            const S pks = ket_label + psubsl[j].second;
            cinfos[j].resize(pks.count());
            for (int k = 0; k < pks.count(); k++) {
                const int ib = (int)(std::lower_bound(vdqs.begin(), vdqs.end(), pks[k]) - vdqs.begin());
                if (ib >= (int)vdqs.size() || vdqs[ib] != pks[k])
                    throw std::runtime_error("perturbative_noise: target label missing");
                const S odq = psubsl[j].second;
                std::vector<std::pair<uint8_t, S>> subdq = {
                    trace_right ? std::make_pair(psubsl[j].first, odq.combine(odq, -idq))
                                : std::make_pair((uint8_t)(psubsl[j].first << 1), odq.combine(idq, -odq))};
                cinfos[j][k] = std::make_shared<typename Info::ConnectionInfo>();
                cinfos[j][k]->initialize_wfn(ket_label, pks[k], odq, subdq, left_op_infos, right_op_infos, ket_info,
                                             vinfos[ib], tf->opf->cg);
            }
        }
        SparseMatrix<S> cmat;
        cmat.info = ket_info, cmat.data = ket, cmat.factor = 1.0;
        std::vector<SparseMatrix<S>> vmats(vinfos.size());
        for (size_t i = 0; i < vinfos.size(); i++)
            vmats[i].info = vinfos[i], vmats[i].data = vdata + voffs[i], vmats[i].factor = 1.0;
        tf->tensor_product_partial_multiply(expr, *lopt, *ropt, trace_right, cmat, psubsl, cinfos, vdqs, vmats, id);
    }
    // record the plan with null-based wavefunctions (every psi / psi' address becomes an element offset)
    void precompute() {
        auto seq = tf->opf->seq;
        if (!seq->pairs.empty())
            return;
        SparseMatrix<S> cmat, vmat;
        cmat.info = ket_info, vmat.info = bra_info;
        cmat.data = vmat.data = (double *)0;
        cmat.factor = vmat.factor = 1.0;
        tf->tensor_product_multiply(expr, *lopt, *ropt, cmat, vmat, opdq);
        seq->prepare(ket_info->get_total_memory(), bra_info->get_total_memory());
    }
    void post_precompute() { tf->opf->seq->clear(); }
    void operator()(const GMatrix &b, const GMatrix &c, double factor = 1.0) {
        precompute();
        (*tf)(b, c, factor);
    }
    std::tuple<double, int, size_t, double> eigs(std::vector<double> &ket, double conv_thrd = 5E-6, int max_iter = 5000,
                                                 int soft_max_iter = -1, int deflation_min_size = 2,
                                                 int deflation_max_size = 50) {
        precompute();
        EffectiveHamiltonian h(tf->opf->seq, diag);
        return h.eigs(ket, false, conv_thrd, 0.0, max_iter, soft_max_iter, deflation_min_size, deflation_max_size);
    }
};

} // namespace b2xh
