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
 *   occ=<file>   nthreads=<n>   seed=<n>   noise=<a,b,c>   tol=<x>   dav_iter=<n>  pg=<d2h|c1>
 */
#include "block2_core.hpp"
#include "block2_dmrg.hpp"
#include "planfile.h"
#include <map>
#include <set>

using namespace block2;
using namespace std;

struct DumpSpec {
    set<pair<int, int>> with_data, structure;
    string prefix;
};

template <typename S> struct Dumper : CallbackKernel {
    typedef double FL;
    DMRG<S, FL, FL> *dmrg = nullptr;
    DumpSpec spec;
    mutable vector<string> log;
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
        } else if (name == "DMRG::sweep::iter.end") {
            stringstream ss;
            ss.precision(15);
            ss << "SITE_ENERGY " << isw << " " << site << " " << dmrg->sweep_energies.back()[0];
            log.push_back(ss.str());
        }
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
    int nth = kv.count("nthreads") ? Parsing::to_int(kv["nthreads"]) : 8;
    frame_<double>() = make_shared<DataFrame<double>>(isize, dsize, kv.count("scratch") ? kv["scratch"] : "/tmp/b2x_ref_scratch");
    frame_<double>()->use_main_stack = false;
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
    auto hamil = make_shared<HamiltonianQC<S, FL>>(vacuum, norb, orbsym, fcidump);
    shared_ptr<MPO<S, FL>> mpo = make_shared<MPOQC<S, FL>>(hamil, QCTypes::Conventional);
    mpo = make_shared<SimplifiedMPO<S, FL>>(mpo, make_shared<RuleQC<S, FL>>(), true, true,
                                             OpNamesSet({OpNames::R, OpNames::RD}));
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
    me->init_environments(false);
    me->delayed_contraction = OpNamesSet::normal_ops();
    me->cached_contraction = true;
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
    auto dumper = make_shared<Dumper<S>>();
    dumper->dmrg = dmrg.get();
    dumper->spec.prefix = prefix;
    if (kv.count("dump"))
        dumper->spec.with_data = parse_pairs(kv["dump"]);
    if (kv.count("struct"))
        dumper->spec.structure = parse_pairs(kv["struct"]);
    callback_() = dumper;
    double tol = kv.count("tol") ? Parsing::to_double(kv["tol"]) : 1E-8;
    Timer t;
    t.get_time();
    double energy = dmrg->solve(n_sweeps, mps->center == 0, tol);
    double tt = t.get_time();
    callback_() = make_shared<CallbackKernel>();
    ofstream lf((prefix + ".log").c_str());
    lf.precision(15);
    for (auto &l : dumper->log)
        lf << l << endl;
    lf << "FINAL_ENERGY " << energy << endl;
    lf << "TOTAL_TIME " << tt << endl;
    for (size_t i = 0; i < dmrg->energies.size(); i++)
        lf << "SWEEP_ENERGY " << i << " " << dmrg->energies[i][0] << " T " << dmrg->sweep_time[i] << endl;
    cout.precision(15);
    cout << "FINAL_ENERGY " << energy << " T = " << tt << endl;
    me->remove_partition_files();
    return 0;
}

int main(int argc, char **argv) {
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
