// b2x_kernels.h — launch entry points of b2x_kernels.hip (internal; the public ABI is include/b2x.h)
#pragma once
#include "b2x_plan.hpp"
#include <hip/hip_runtime_api.h>

namespace b2x {

// index of the psi / sigma / scale argument of a kernel this file launches (-1: the kernel has none); false for a kernel
// function the launchers have not seen
struct KernelArgSlots {
    int psi, sigma, scale;
};
bool kernel_arg_slots(const void *func, KernelArgSlots *out);

hipError_t launch_main(int cls, const DPart *parts, const DItem *items, uint32_t n_items, const double *arena,
                       const double *psi, double *slabs, hipStream_t st);
hipError_t launch_gg(const GSeg *segs, const GItem *items, const uint32_t *v_begin, const double *arena,
                     const double *psi, double *scratch, double *slabs, bool seg_scaled, int tile_n, hipStream_t st, int which = 0,
                     bool short_narrow = false, int short_frags = kGGShortFrags);
hipError_t launch_outer(const OWork *work, uint32_t n_work, const OEntry *entries, const double *arena, const double *in,
                        double *out, int rows_in_flight, hipStream_t st);
// max_elems / max_items: elements of the largest tile / slabs of the tile with the most slabs (0: unknown) — they pick the grid
// and, for tiles with many slabs, the kernel that splits the slabs of an element over several threads
hipError_t launch_reduce(const DTile *tiles, uint32_t n_tiles, const double *slabs, double *sigma, double scale,
                         hipStream_t st, uint32_t max_elems = 0, uint32_t max_items = 0);
hipError_t launch_generic(const b2x_pair *pairs, uint32_t n_pairs, const double *arena, const double *psi,
                          double *sigma, double scale, hipStream_t st);
hipError_t launch_diag(const DiagComp *comps, uint32_t n_comps, const DiagTermD *terms, const double *arena, double *diag,
                       hipStream_t st);
hipError_t launch_axpy(double a, const double *x, double *y, size_t n, hipStream_t st);
hipError_t launch_scal(double a, double *x, size_t n, hipStream_t st);
hipError_t launch_precond(double *q, const double *diag, double shift, size_t n, hipStream_t st);
hipError_t launch_olsen(const double *q, double *q_out, double *t, const double *c, const double *diag, double ld, size_t n,
                        hipStream_t st);
// x = sum a_j b_j, q = sum a_j s_j - theta x, q2 = q / (theta - diag), t = x / (theta - diag) in one pass (m <= 64)
hipError_t launch_ritz_olsen(const double *const *bs, const double *const *ss, int m, const double *alpha, double theta,
                             const double *diag, double *x, double *q, double *q2, double *t, size_t n, hipStream_t st);
// second Gram-Schmidt pass + normalisation without a host round trip (m <= 63): dots / partial = device scratch of the dot products
hipError_t launch_gs_finish(const double *const *bs, int m, const double *v, double *partial, double *dots, double *out, size_t n,
                            int *flag, hipStream_t st);
// descs: device array of n {dst offset, src offset, length} triples (uint64 each, elements)
hipError_t launch_gather(const void *descs, uint32_t n, double *dst, const double *src, hipStream_t st);
hipError_t launch_pairdot(const double *const *us, const double *const *vs, int np, size_t n, double *partial, double *out,
                          hipStream_t st);
hipError_t launch_lincomb(const double *const *vs, const double *coef, int nv, double *y, size_t n, hipStream_t st);
int multidot_blocks(size_t n);
hipError_t launch_multidot(const double *const *vs, int nv, const double *x, size_t n, double *partial, double *out,
                           hipStream_t st);

} // namespace b2x
