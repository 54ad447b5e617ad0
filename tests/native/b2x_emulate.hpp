// b2x_emulate.hpp — TEST-ONLY host evaluation of exactly what the device kernels compute from a CompiledPlan (plain
// loops, no MFMA).  Used by tests/native/b2x_testhooks.cpp and tools/asan_plan_check.cpp; never part of the product.
#pragma once
#include "../../block2-preview_amd/csrc/b2x_plan.hpp"
#include <cstring>
#include <vector>

namespace b2x_test {
using namespace b2x;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

inline void emulate_outer_host(const std::vector<OWork> &work, const std::vector<OEntry> &entries, const double *arena,
                        const double *in, double *out) {
    static const double one = 1.0;
    for (const OWork &w : work) {
        const uint32_t nseg = (uint32_t)ceil_div(w.cols, kOuterTileCols);
        for (uint32_t tile = w.t_begin; tile < w.t_end; tile++) {
            const int r0 = (int)(tile / nseg) * w.rpt, c0 = (int)(tile % nseg) * kOuterTileCols;
            for (int r = r0; r < std::min(w.rows, r0 + w.rpt); r++)
                for (int c = c0; c < std::min(w.cols, c0 + kOuterTileCols); c++) {
                    double sum = 0.0;
                    for (uint32_t k = w.entry_begin; k < w.entry_end; k++) {
                        const OEntry &t = entries[k];
                        const double a = t.a_src == 2 ? one : (t.a_src ? in : arena)[t.a_off + (uint64_t)r * t.a_rs + (uint64_t)c * t.a_cs];
                        const double b = t.b_src == 2 ? one : (t.b_src ? in : arena)[t.b_off + (uint64_t)r * t.b_rs + (uint64_t)c * t.b_cs];
                        sum += t.alpha * a * b;
                    }
                    if (w.ld < 0)
                        out[w.out_off + (uint64_t)r * (uint64_t)(-w.ld) + c] = sum;
                    else
                        out[w.out_off + (uint64_t)r * w.ld + c] += sum;
                }
        }
    }
}


inline void emulate_plan_host(const CompiledPlan &cp, const double *arena, const double *psi, double *sigma, double scale) {
    std::vector<double> slabs(cp.slab_elems, 0.0);
    for (int k = 0; k < kNumClasses; k++) {
        const ClassWork &cw = cp.cls[k];
        for (const DItem &it : cw.items) {
            double *acc = slabs.data() + it.slab_off;
            for (uint32_t pi = it.part_begin; pi < it.part_end; pi++) {
                const DPart &P = cw.parts[pi];
                std::vector<double> w((size_t)P.k1 * P.nc);
                for (int r = 0; r < P.k1; r++)
                    for (int c = 0; c < P.nc; c++) {
                        double s = 0;
                        for (int k2 = 0; k2 < P.k0; k2++)
                            s += psi[P.x_off + (uint64_t)r * P.ldx + k2] *
                                 arena[P.y_off + (uint64_t)k2 * P.sky + (uint64_t)c * P.scy];
                        w[(size_t)r * P.nc + c] = s * P.alpha;
                    }
                for (int r = 0; r < P.mr; r++)
                    for (int c = 0; c < P.nc; c++) {
                        double s = 0;
                        for (int k2 = 0; k2 < P.k1; k2++)
                            s += arena[P.z_off + (uint64_t)r * P.srz + (uint64_t)k2 * P.skz] * w[(size_t)k2 * P.nc + c];
                        acc[(size_t)(P.tr0 + r) * it.cols + P.tc0 + c] += s;
                    }
            }
        }
    }
    // two-stage path
    std::vector<double> scratch(cp.scratch_elems, 0.0), gslabs(cp.gslab_elems, 0.0);
    for (const StageCopy &sc : cp.stage) // operands staged behind the W slots (arena: at upload; input: per execute)
        memcpy(scratch.data() + sc.dst_off, (sc.src == 0 ? arena : psi) + sc.src_off, sc.len * sizeof(double));
    if (!cp.aux_work.empty()) // operator pre-sums (done once at plan creation on the device)
        emulate_outer_host(cp.aux_work, cp.aux_entries, arena, scratch.data(), scratch.data());
    auto run_item = [&](const GItem &it) {
        std::vector<double> acc((size_t)it.rows * it.cols, 0.0);
        for (uint32_t si = it.seg_begin; si < it.seg_end; si++) {
            const GSeg &g = cp.gsegs[si];
            const double *A = g.a_src == 0 ? arena : (g.a_src == 1 ? psi : scratch.data());
            const double *B = g.b_src == 0 ? arena : (g.b_src == 1 ? psi : scratch.data());
            for (int r = 0; r < g.mr; r++)
                for (int c = 0; c < g.nc; c++) {
                    double s = 0;
                    for (int k = 0; k < g.K; k++)
                        s += A[g.a_off + (uint64_t)r * g.a_sr + (uint64_t)k * g.a_sk] *
                             B[g.b_off + (uint64_t)k * g.b_sk + (uint64_t)c * g.b_sc];
                    acc[(size_t)r * it.cols + g.tc0 + c] += g.alpha * s;
                }
        }
        double *o = (it.out_kind ? scratch.data() : gslabs.data()) + it.out_off;
        for (int r = 0; r < it.rows; r++)
            for (int c = 0; c < it.cols; c++)
                o[(size_t)r * it.out_ld + c] = it.alpha * acc[(size_t)r * it.cols + c];
    };
    for (const SuperStep &ss : cp.steps) {
        for (uint32_t i = ss.s0_v[0]; i < ss.s0_v[kGGVariants]; i++)
            run_item(cp.gitems[i]);
        if (ss.sum_end > ss.sum_begin) { // S = sum_i alpha_i W_i (scratch -> scratch)
            std::vector<OWork> wk(cp.sum_work.begin() + ss.sum_begin, cp.sum_work.begin() + ss.sum_end);
            std::vector<double> src = scratch;
            emulate_outer_host(wk, cp.sum_entries, arena, src.data(), scratch.data());
        }
        for (uint32_t i = ss.s1_v[0]; i < ss.s1_v[kGGVariants]; i++)
            run_item(cp.gitems[i]);
        for (uint32_t ti = ss.tile_begin; ti < ss.tile_end; ti++) {
            const DTile &t = cp.gtiles[ti];
            for (int r = 0; r < t.rows; r++)
                for (int c = 0; c < t.cols; c++) {
                    double s = 0;
                    for (int i = 0; i < t.n_items; i++)
                        s += gslabs[t.slab_off + (uint64_t)i * t.rows * t.cols + (uint64_t)r * t.cols + c];
                    sigma[t.sigma_off + (uint64_t)r * t.ld + c] += scale * s;
                }
        }
    }
    for (const DTile &t : cp.tiles)
        for (int r = 0; r < t.rows; r++)
            for (int c = 0; c < t.cols; c++) {
                double s = 0;
                for (int i = 0; i < t.n_items; i++)
                    s += slabs[t.slab_off + (uint64_t)i * t.rows * t.cols + (uint64_t)r * t.cols + c];
                sigma[t.sigma_off + (uint64_t)r * t.ld + c] += scale * s;
            }
}



} // namespace b2x_test
