/*
 * ref_replay.cpp — replays a B2XPLAN1 file with the REFERENCE executor.  TEST/BASELINE
 * INFRASTRUCTURE, built against the read-only reference headers into oracle/_ref/ (see
 * oracle/Makefile); the product never links or calls it.
 *
 * Each pair is recorded exactly as AdvancedGEMM<double>::multiply does for the rotate pair
 * (src/core/batch_gemm.hpp:329-337, 564-575): stage 0 into batch[0] with the W offset as C,
 * stage 1 into batch[1]; then BatchGEMMSeq<double>::operator() (Tasked branch,
 * src/core/batch_gemm.hpp:1606-1682) executes it: OpenMP static split over pairs,
 * thread-private psi', tree reduction.
 *
 * usage: ref_replay <plan> [out=<sigma.bin>] [reps=<n>] [threads=<n>] [scale=<x>] [seed=<n>]
 * prints: REP <i> sec=<s> per replay, then REPLAY pairs=<n> macs=<n> threads=<t> reps=<r> sec_per_replay=<s> gmacs=<x>
 */
#include "block2_core.hpp"
#include "planfile.h"
#include <map>

using namespace block2;
using namespace std;

int main(int argc, char **argv) {
    if (argc < 2) {
        cerr << "usage: ref_replay <plan> [out=..] [reps=..] [threads=..] [scale=..]" << endl;
        return 2;
    }
    map<string, string> kv;
    for (int i = 2; i < argc; i++) {
        string a = argv[i];
        size_t e = a.find('=');
        kv[a.substr(0, e)] = e == string::npos ? "1" : a.substr(e + 1);
    }
    int nth = kv.count("threads") ? atoi(kv["threads"].c_str()) : 8;
    int reps = kv.count("reps") ? atoi(kv["reps"].c_str()) : 1;
    double scale = kv.count("scale") ? atof(kv["scale"].c_str()) : 1.0;
    b2x_planfile pf;
    if (b2x_planfile_read(argv[1], &pf) != 0) {
        cerr << "cannot read plan " << argv[1] << endl;
        return 1;
    }
    frame_<double>() = make_shared<DataFrame<double>>(1 << 20, 1 << 20, "/tmp/b2x_ref_replay");
    frame_<double>()->use_main_stack = false;
    threading_() = make_shared<Threading>(ThreadingTypes::OperatorBatchedGEMM | ThreadingTypes::Global, nth, nth, 1);
    threading_()->seq_type = SeqTypes::Tasked;
    Random::rand_seed(kv.count("seed") ? (unsigned)atoi(kv["seed"].c_str()) : 1969u);
    vector<double> arena_own, psi_own;
    if (pf.arena == nullptr) {
        arena_own.resize(pf.arena_len);
        Random::fill<double>(arena_own.data(), arena_own.size());
        pf.arena = arena_own.data();
    }
    if (pf.psi == nullptr) {
        psi_own.resize(pf.psi_len);
        Random::fill<double>(psi_own.data(), psi_own.size());
        pf.psi = psi_own.data();
    }
    auto seq = make_shared<BatchGEMMSeq<double>>(1 << 24, SeqTypes::Tasked);
    for (uint64_t i = 0; i < pf.n_pairs; i++) {
        const b2x_pair &p = pf.pairs[i];
        size_t wsz = (size_t)p.m0 * p.n0;
        double *w = (double *)0 + seq->batch[0]->work;
        seq->batch[0]->xgemm(p.ta0, p.tb0, p.m0, p.n0, p.k0, p.alpha0, (const double *)0 + p.x_off, p.lda0,
                             pf.arena + p.y_off, p.ldb0, 0.0, w, p.n0);
        seq->batch[1]->xgemm(p.ta1, p.tb1, p.m1, p.n1, p.k1, p.alpha1, pf.arena + p.z_off, p.lda1, w, p.n0, 1.0,
                             (double *)0 + p.v_off, p.ldc1);
        seq->max_work = max(seq->max_work, wsz);
        seq->batch[0]->work += wsz;
        seq->batch[1]->work += wsz;
    }
    uint64_t macs = seq->batch[0]->nflop + seq->batch[1]->nflop;
    vector<double> sigma(pf.sigma_len, 0.0);
    Timer t;
    t.get_time();
    double total = 0;
    cout.precision(9);
    for (int r = 0; r < reps; r++) {
        if (r == reps - 1)
            fill(sigma.begin(), sigma.end(), 0.0);
        (*seq)(GMatrix<double>(pf.psi, (MKL_INT)pf.psi_len, 1), GMatrix<double>(sigma.data(), (MKL_INT)pf.sigma_len, 1),
               scale);
        const double dt = t.get_time();
        total += dt;
        cout << "REP " << r << " sec=" << dt << endl; // (the first replay also pays MKL's thread start and first touches)
    }
    double tt = total / reps;
    cout << "REPLAY pairs=" << pf.n_pairs << " macs=" << macs << " threads=" << nth << " reps=" << reps
         << " sec_per_replay=" << tt << " gmacs=" << (double)macs / tt * 1e-9 << endl;
    if (kv.count("out")) {
        FILE *f = fopen(kv["out"].c_str(), "wb");
        fwrite(sigma.data(), 8, sigma.size(), f);
        fclose(f);
    }
    if (pf.sigma_ref != nullptr && scale == 1.0) {
        double md = 0, mx = 0;
        for (size_t i = 0; i < sigma.size(); i++)
            md = max(md, fabs(sigma[i] - pf.sigma_ref[i])), mx = max(mx, fabs(pf.sigma_ref[i]));
        cout << "CHECK max_abs_diff=" << md << " max_abs_ref=" << mx << endl;
    }
    if (arena_own.size())
        pf.arena = nullptr;
    if (psi_own.size())
        pf.psi = nullptr;
    return 0;
}
