// ASan/UBSan driver for the plan compiler (host code only; GPU sanitizers are not available on the pool):
//   cd block2-preview_amd/csrc && g++ -std=c++17 -g -O1 -fsanitize=address,undefined -o /tmp/asan_plan ../../tools/asan_plan_check.cpp b2x_plan.cpp
//   ASAN_OPTIONS=detect_leaks=0 /tmp/asan_plan tests/golden/*.plan      (leaks: the file reader of this driver never frees)
// Every golden plan is compiled in four modes (fused, two-stage, reference order, pre-sums + 1 MiB scratch) and evaluated
// with the host emulation against the reference's sigma.  TEST TOOL: includes oracle/planfile.h for the file format only.
#include "../tests/native/b2x_emulate.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
extern "C" {
#include "../oracle/planfile.h"
}
using namespace b2x;
using b2x_test::emulate_plan_host;
int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        b2x_planfile pf;
        if (b2x_planfile_read(argv[a], &pf)) { printf("cannot read %s\n", argv[a]); return 1; }
        for (int mode = 0; mode < 4; mode++) {
            b2x_plan_options opt{};
            opt.two_stage = mode >= 1; opt.keep_order = mode == 2; opt.presum = mode == 3; opt.scratch_mb = mode == 3 ? 1 : 0;
            CompiledPlan cp; std::string err;
            int rc = compile_plan(pf.n_pairs, pf.pairs, pf.psi_len, pf.sigma_len, pf.arena_len, pf.arena_len, &opt, cp, err);
            if (rc) { printf("compile failed: %s\n", err.c_str()); return 1; }
            std::vector<double> sig(pf.sigma_len, 0.0);
            if (!cp.fallback && pf.arena && pf.psi) emulate_plan_host(cp, pf.arena, pf.psi, sig.data(), 1.0);
            double e = 0, m = 0;
            if (pf.sigma_ref) for (size_t i = 0; i < sig.size(); i++) { e = std::max(e, std::abs(sig[i] - pf.sigma_ref[i])); m = std::max(m, std::abs(pf.sigma_ref[i])); }
            printf("%s mode %d: err %.2e / %.2e executed/alg %.3f\n", argv[a], mode, e, m, (double)cp.stats.macs_executed / cp.stats.macs);
        }
    }
    return 0;
}
