// b2x_testhooks.cpp — TEST-ONLY library (tests/native/libb2x_testhooks.so): evaluates the work lists the plan compiler
// produces with plain host loops, so that the non-GPU test-suite can verify the PLAN COMPILER (segmentation into tiles /
// parts / items, the algebraic rewrites) against the oracle.  It links the host-only compiler source
// (block2-preview_amd/csrc/b2x_plan.cpp) and nothing of the device path; it is NOT part of libb2x.so and nothing in the
// product loads it.
#include <cstdio>
#include <cstdlib>
#include "b2x_emulate.hpp"
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

using namespace b2x;
using b2x_test::emulate_outer_host;
using b2x_test::emulate_plan_host;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

extern "C" {

const char *b2x_test_last_error(void) { return g_err.c_str(); }

// test hook: segmentation of diagonal terms only (no device)
int b2x_debug_compile_diag(size_t n_terms, const b2x_diag_term *terms, size_t diag_len, uint64_t arena_len,
                           uint64_t *n_comps) {
    std::vector<DiagComp> comps;
    std::vector<DiagTermD> dterms;
    std::string err;
    int rc = compile_diag(n_terms, terms, diag_len, arena_len, comps, dterms, err);
    if (rc != B2X_OK)
        return fail(rc, err);
    if (n_comps)
        *n_comps = comps.size();
    return B2X_OK;
}

// ---------------------------------------------------------------------------------- test hook
// Compiles a plan and evaluates the compiled work list with plain host loops.  Exists so the
// non-GPU test-suite can verify the PLAN COMPILER (segmentation into tiles/parts/items) against
// the oracle; it is not declared in include/b2x.h and nothing in the product calls it.
int b2x_debug_compile_and_emulate(size_t n_pairs, const b2x_pair *pairs, size_t psi_len, size_t sigma_len,
                                  uint64_t arena_len, const double *arena, const double *psi, double *sigma,
                                  double scale, const b2x_plan_options *opt, b2x_plan_stats *stats, int *fallback) {
    CompiledPlan cp;
    std::string err;
    int rc = compile_plan(n_pairs, pairs, psi_len, sigma_len, arena_len, arena_len, opt, cp, err);
    if (rc != B2X_OK)
        return fail(rc, err);
    if (stats)
        *stats = cp.stats;
    if (fallback)
        *fallback = cp.fallback ? 1 : 0;
    // launch contract of the 1-wave workgroups: their class holds only tiles of <= kGGNarrowFrags fragments x kGGNarrowN columns
    if (cp.short_narrow)
        for (const SuperStep &ss : cp.steps)
            for (int stg = 0; stg < 2; stg++) {
                const uint32_t *v = stg ? ss.s1_v : ss.s0_v;
                for (uint32_t i = v[1]; i < v[kGGVariants]; i++) {
                    const GItem &it = cp.gitems[i];
                    bool ok = it.rows >= 1 && it.rows <= kGGNarrowFrags * kGGRowUnit && it.cols >= 1 && it.cols <= kGGNarrowN;
                    for (uint32_t sg = it.seg_begin; sg < it.seg_end && ok; sg++)
                        ok = cp.gsegs[sg].mr == it.rows && cp.gsegs[sg].tc0 >= 0 && cp.gsegs[sg].tc0 + cp.gsegs[sg].nc <= it.cols;
                    if (!ok)
                        return fail(B2X_ERR_INVALID, "narrow-class item outside the 1-wave kernel's contract");
                }
                for (uint32_t i = v[0]; i < v[1]; i++)
                    if (cp.gitems[i].rows <= kGGNarrowFrags * kGGRowUnit)
                        return fail(B2X_ERR_INVALID, "short tile in the tall class of a narrow plan");
            }
    if (!cp.fallback && arena && psi && sigma)
        emulate_plan_host(cp, arena, psi, sigma, scale);
    if (getenv("B2X_TEST_TRAFFIC_MODEL")) { // operand bytes the grouped-GEMM work list requests, per stage and source
        double by[2][2][3] = {}, outb[2] = {};
        uint64_t nit[2] = {}, nseg[2] = {}, chunks[2] = {};
        for (const SuperStep &ss : cp.steps)
            for (int stg = 0; stg < 2; stg++) {
                const uint32_t *v = stg ? ss.s1_v : ss.s0_v;
                for (uint32_t i = v[0]; i < v[kGGVariants]; i++) {
                    const GItem &it = cp.gitems[i];
                    nit[stg]++, outb[stg] += 8.0 * it.rows * it.cols;
                    for (uint32_t sg = it.seg_begin; sg < it.seg_end; sg++) {
                        const GSeg &g = cp.gsegs[sg];
                        nseg[stg]++, chunks[stg] += (g.K + 15) / 16;
                        by[stg][0][g.a_src] += 8.0 * it.rows * g.K;
                        by[stg][1][g.b_src] += 8.0 * g.K * it.cols;
                    }
                }
            }
        for (int stg = 0; stg < 2; stg++) {
            uint64_t n32 = 0, n64 = 0, nw = 0, c32 = 0, call = 0;
            for (const SuperStep &ss : cp.steps) {
                const uint32_t *v = stg ? ss.s1_v : ss.s0_v;
                for (uint32_t i = v[0]; i < v[kGGVariants]; i++) {
                    const GItem &it = cp.gitems[i];
                    uint64_t ch = 0;
                    for (uint32_t sg = it.seg_begin; sg < it.seg_end; sg++)
                        ch += (cp.gsegs[sg].K + 15) / 16;
                    (it.cols <= 32 ? n32 : it.cols <= 64 ? n64 : nw)++;
                    call += ch;
                    if (it.cols <= 32)
                        c32 += ch;
                }
            }
            fprintf(stderr, "stage %d: items with cols <= 32: %llu (chunks %llu of %llu), <= 64: %llu, wider: %llu\n", stg,
                    (unsigned long long)n32, (unsigned long long)c32, (unsigned long long)call, (unsigned long long)n64,
                    (unsigned long long)nw);
        }
        {
            double byf[2][17] = {};
            for (const SuperStep &ss : cp.steps)
                for (int stg = 0; stg < 2; stg++) {
                    const uint32_t *v = stg ? ss.s1_v : ss.s0_v;
                    for (uint32_t i = v[0]; i < v[kGGVariants]; i++) {
                        const GItem &it = cp.gitems[i];
                        const int fr = (it.rows + 15) / 16;
                        for (uint32_t sg = it.seg_begin; sg < it.seg_end; sg++)
                            byf[stg][std::min(fr, 16)] += (double)fr * ((cp.gsegs[sg].K + 15) / 16);
                    }
                }
            for (int stg = 0; stg < 2; stg++) {
                double tot = 0;
                for (int f = 1; f <= 16; f++)
                    tot += byf[stg][f];
                fprintf(stderr, "stage %d issue-slot share by row fragments:", stg);
                for (int f = 1; f <= 8; f++)
                    fprintf(stderr, " %d:%.3f", f, tot > 0 ? byf[stg][f] / tot : 0.0);
                fprintf(stderr, "\n");
            }
        }
        for (int stg = 0; stg < 2; stg++)
            fprintf(stderr, "stage %d: items %llu segs %llu chunks %llu | A arena %.3f psi %.3f scratch %.3f | B arena %.3f psi %.3f "
                    "scratch %.3f | out %.3f GB\n", stg, (unsigned long long)nit[stg], (unsigned long long)nseg[stg],
                    (unsigned long long)chunks[stg], by[stg][0][0] / 1e9, by[stg][0][1] / 1e9, by[stg][0][2] / 1e9,
                    by[stg][1][0] / 1e9, by[stg][1][1] / 1e9, by[stg][1][2] / 1e9, outb[stg] / 1e9);
    }
    return B2X_OK;
}

int b2x_debug_compile_and_emulate_gemms(size_t n_gemms, const b2x_gemm *gemms, size_t in_len, size_t out_len,
                                        uint64_t arena_len, const double *arena, const double *in, double *out,
                                        double scale, const b2x_plan_options *opt, b2x_plan_stats *stats) {
    CompiledPlan cp;
    std::string err;
    int rc = compile_gemm_list(n_gemms, gemms, in_len, out_len, arena_len, arena_len, opt, cp, err);
    if (rc != B2X_OK)
        return fail(rc, err);
    if (stats)
        *stats = cp.stats;
    if (arena && in && out)
        emulate_plan_host(cp, arena, in, out, scale);
    return B2X_OK;
}

int b2x_debug_compile_and_emulate_outer(size_t n_terms, const b2x_outer_term *terms, size_t in_len, size_t out_len,
                                        uint64_t arena_len, const double *arena, const double *in, double *out,
                                        uint64_t *n_work, uint64_t *n_entries) {
    std::vector<OWork> work;
    std::vector<OEntry> entries;
    std::string err;
    int rc = compile_outer(n_terms, terms, in_len, out_len, arena_len, work, entries, err);
    if (rc != B2X_OK)
        return fail(rc, err);
    if (n_work)
        *n_work = work.size();
    if (n_entries)
        *n_entries = entries.size();
    if (arena && out)
        emulate_outer_host(work, entries, arena, in, out);
    return B2X_OK;
}


} // extern "C"
