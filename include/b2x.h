/*
 * b2x.h — C ABI of the MI355X-native H·psi (effective-Hamiltonian contraction) path.
 *
 * This is the drop-in boundary for block2's GEMM-pair plan replay.  Every entry point
 * names the reference interface it replaces (paths relative to the block2 source tree):
 *
 *   plan build    <- BatchGEMMSeq<double>::rotate / three_rotate   src/core/batch_gemm.hpp:893-902, 952-1022
 *                    (what EffectiveHamiltonian::precompute() records, src/dmrg/effective_hamiltonian.hpp:224-244)
 *   plan execute  <- BatchGEMMSeq<double>::operator()(c, v, scale) src/core/batch_gemm.hpp:1563-1684 (Tasked branch)
 *                    reached from TensorFunctions::operator()      src/core/tensor_functions.hpp:59-62
 *   plan destroy  <- EffectiveHamiltonian::post_precompute()       src/dmrg/effective_hamiltonian.hpp:245-251
 *   all-reduce    <- ParallelCommunicator::allreduce_sum(double*, size_t)  src/core/parallel_rule.hpp:55
 *                    (MPI body: src/core/parallel_mpi.hpp:300-309), called by
 *                    ParallelTensorFunctions::operator()           src/core/parallel_tensor_functions.hpp:51-55
 *   vector ops    <- the BLAS-1 calls of IterativeMatrixFunctions::davidson
 *                    src/core/iterative_matrix_functions.hpp:864-1173 (ddot/daxpy/dscal/dcopy on |psi| vectors)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * non-zero code on failure (b2x_last_error() gives the text).  The C++ host wrapper turns a
 * non-zero return into std::runtime_error, matching the reference's error convention.
 * All matrices are ROW-MAJOR with explicit leading dimension, as GMatrix (src/core/matrix.hpp:92-107).
 */
#ifndef B2X_H
#define B2X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B2X_OK 0
#define B2X_ERR_INVALID 1  /* bad argument / unsupported descriptor */
#define B2X_ERR_DEVICE 2   /* HIP runtime failure */
#define B2X_ERR_NOMEM 3
#define B2X_ERR_STATE 4

/*
 * One GEMM pair of the plan (real double: exactly one stage-0 and one stage-1 descriptor per
 * H·psi term; batch[0] / batch[1] slots i of BatchGEMMSeq, src/core/batch_gemm.hpp:237-247).
 *
 *   stage 0:  W (m0 x n0)  = alpha0 * opA0(X)(m0 x k0) * opB0(Y)(k0 x n0)          beta = 0
 *   stage 1:  V (m1 x n1) += alpha1 * scale * opA1(Z)(m1 x k1) * opB1(W)(k1 x n1)  beta = 1
 *
 * X is a slice of psi (x_off = batch[0]->a[i], an element offset "from null"), Y and Z are
 * operator blocks (absolute host pointers in the reference; arena element offsets here),
 * V is a slice of psi' (v_off = batch[1]->c[i]).  W is pair-local (never leaves the chip).
 * On this path ta0 = tb1 = 0, k1 = m0, n1 = n0, ldc0 = ldb1 = n0.
 */
typedef struct b2x_pair {
    int32_t m0, n0, k0;
    int32_t lda0, ldb0;
    int32_t m1, n1, k1;
    int32_t lda1, ldc1;
    uint8_t ta0, tb0, ta1, tb1; /* 0 = no transpose, 1 = transpose */
    uint32_t reserved;
    double alpha0, alpha1;
    uint64_t x_off; /* psi   element offset */
    uint64_t y_off; /* arena element offset (stage-0 B operand) */
    uint64_t z_off; /* arena element offset (stage-1 A operand) */
    uint64_t v_off; /* psi'  element offset */
} b2x_pair;

typedef struct b2x_arena b2x_arena;
typedef struct b2x_plan b2x_plan;

typedef struct b2x_plan_stats {
    uint64_t n_pairs;
    uint64_t macs;            /* sum m0*n0*k0 + m1*n1*k1 == reference nflop (batch_gemm.hpp:307) */
    uint64_t op_elems_unique; /* distinct operator elements referenced by the plan */
    uint64_t psi_len, sigma_len;
    uint64_t n_targets;       /* disjoint psi' output regions after overlap merging */
    uint64_t n_tiles;         /* output tiles */
    uint64_t n_items;         /* work items (tile x pair-chunk) launched per execute */
    uint64_t n_parts;         /* (pair x tile) parts */
    uint64_t device_bytes;    /* plan metadata + partial-sum slabs resident in HBM */
    uint64_t macs_executed;   /* MACs the kernels really execute: > macs where the fused kernel recomputes stage 0 per row
                                 tile, < macs where the cheaper association of a pair is taken (keep_order = 0) */
    uint64_t dominant_class;  /* kernel class that carries most MACs */
    uint64_t macs_dominant;   /* MACs executed by the dominant kernel */
    uint64_t macs_alg_dominant; /* reference-count MACs (the pairs' m0 n0 k0 + m1 n1 k1) of the work in the dominant kernel */
    uint64_t n_launches;      /* launches of the dominant kernel per execute */
    uint64_t macs_issued;     /* MFMA issue slots x 1024 of the two-stage path incl. tile padding (0 if unused) */
    uint64_t fallback;        /* 1: the output windows could not be segmented, the plan runs on the per-pair atomic kernel
                                 (not bitwise reproducible); 0 for every plan the reference's DMRG records */
    uint64_t n_staged;        /* operands copied into plan-owned memory because a 16-byte fetch would otherwise touch the
                                 element behind the caller's buffer (operands that end exactly at the end of psi or of an
                                 adopted arena): arena operands once at plan creation, psi operands per execute */
    /* the three rewrites of the pair list (all zero with keep_order = 1) */
    uint64_t n_flipped;         /* pairs taken in the other association, (op(Z).X).op(Y) */
    uint64_t n_shared_products; /* pairs that reuse the stage-0 product of another pair */
    uint64_t n_merged_groups;   /* groups of pairs whose scaled stage-0 products are summed before ONE stage-1 product */
    uint64_t n_merged_members;  /* pairs in those groups */
} b2x_plan_stats;

/* tuning knobs; pass NULL for defaults */
typedef struct b2x_plan_options {
    int32_t tile_m, tile_n;       /* 0 = auto */
    int32_t kernel;               /* 0 = auto, 1 = scalar reference kernel, 2 = MFMA kernel */
    int64_t item_macs;            /* target MACs per work item (0 = auto) */
    int32_t two_stage;            /* 0 = auto (sectors taller than one fused tile), 1 = always, -1 = never */
    int32_t scratch_mb;           /* W scratch budget of the two-stage path in MiB (0 = 16384) */
    int32_t keep_order;           /* 1 = always form X.op(Y) first, as the reference does; 0 = per pair the cheaper of
                                     (op(Z).X).op(Y) and op(Z).(X.op(Y)) (same result up to rounding) */
    int32_t presum;               /* 1 = also pre-sum, at plan creation, the second operators of pairs that share a stage-0
                                     product and a psi' window (costs plan-owned memory) */
    int32_t reserved[4];
} b2x_plan_options;

const char *b2x_last_error(void);
const char *b2x_version(void);

/* device ------------------------------------------------------------------------------- */
int b2x_device_count(int *n);
int b2x_device_init(int ordinal);                 /* hipSetDevice; fails if no gfx950 device */
int b2x_device_sync(void);
int b2x_device_alloc(void **dptr, size_t bytes);  /* device vector of at least `bytes` (size classes of 1/8 octave; freed vectors are kept for the next request of the class: B2X_VEC_CACHE_MB, b2x_trim) */
int b2x_device_free(void *dptr);
int b2x_memcpy_h2d(void *dst, const void *src, size_t bytes);
int b2x_memcpy_d2h(void *dst, const void *src, size_t bytes);

/* operator arena ---------------------------------------------------------------------------
 * Replaces: the heap-resident operator blocks OperatorTensor::ops[*]->data that the reference
 * plan points into (src/core/operator_tensor.hpp:47; immutable for one EffectiveHamiltonian). */
int b2x_arena_create(b2x_arena **out, size_t n_ranges, const double *const *host_bases,
                     const size_t *lens);               /* concatenates + uploads */
int b2x_arena_adopt_device(b2x_arena **out, double *dev_base, size_t len); /* no copy, not owned; exactly len
                                                                              elements are ever read (no slack needed) */
int b2x_arena_resolve(const b2x_arena *a, const double *host_ptr, uint64_t *off);
int b2x_arena_len(const b2x_arena *a, uint64_t *len);
int b2x_arena_device_ptr(const b2x_arena *a, double **dev_base);
int b2x_arena_destroy(b2x_arena *a);

/* plan -------------------------------------------------------------------------------------- */
int b2x_plan_create(b2x_plan **out, const b2x_arena *arena, size_t n_pairs, const b2x_pair *pairs,
                    size_t psi_len, size_t sigma_len, const b2x_plan_options *opt);
/* sigma += scale * H * psi.  on_device != 0: psi/sigma are device pointers (no copies);
 * stream: hipStream_t (NULL = default stream).  Asynchronous when on_device != 0.
 * Exactly psi_len elements of psi, sigma_len of sigma and the arena's len elements are accessed: buffers need no
 * slack (the kernels fetch 16-byte granules; an operand at the very end of a buffer is read from a copy the plan
 * stages in its own memory, see b2x_plan_stats.n_staged). */
int b2x_plan_execute(b2x_plan *p, const double *psi, double *sigma, double scale, int on_device,
                     void *stream);
int b2x_plan_get_stats(const b2x_plan *p, b2x_plan_stats *st);
/* time n launches of the dominant kernel with HIP events on `stream`; returns average ms */
int b2x_plan_time_kernel(b2x_plan *p, const double *psi_dev, double *sigma_dev, int n, void *stream,
                         double *avg_ms_main, double *avg_ms_total);
int b2x_plan_destroy(b2x_plan *p);
/* Compiled plans are cached: b2x_plan_destroy keeps the plan's work lists in HBM (its scratch and slabs go back to a buffer
 * pool), and a later b2x_plan_create / b2x_gemm_plan_create with byte-identical records, lengths, options and arena extent
 * takes the plan back instead of compiling again — the case of every site of a sweep once the bond dimensions have settled
 * (the reference rebuilds its plan in precompute() every time, effective_hamiltonian.hpp:224-251).  LRU, bounded by
 * B2X_PLAN_CACHE_MB MiB of work lists (default 8192; 0 disables).  Counters since process start; clear() frees the cache. */
int b2x_plan_cache_stats(uint64_t *hits, uint64_t *misses, uint64_t *plans, uint64_t *bytes);
int b2x_plan_cache_clear(void);
/* Idle device memory the library holds — the parked plans of the cache above and the pool of recycled scratch / slab buffers
 * (B2X_POOL_MB, default a quarter of the card) — is given back to the driver: automatically whenever one of the library's
 * own allocations fails (the allocation is then retried), and on request by b2x_trim, for other allocators of the same
 * process (the counterpart of the reference returning a site's stack memory, src/core/allocator.hpp:175-214).
 * bytes_released (may be NULL): growth of free device memory over the call. */
int b2x_trim(uint64_t *bytes_released);

/* single-GEMM lists (perturbative noise; partial multiplies) ------------------------------------------------
 * Replaces: the batch[1]-only lists that BatchGEMMSeq::multiply / three_rotate_tr_left / three_rotate_tr_right
 * record (src/core/batch_gemm.hpp:887-891, 1025-1109) for OperatorFunctions::tensor_product_multiply with
 * TraceTypes::Left / Right (src/core/operator_functions.hpp:518-535), replayed by BatchGEMMSeq::auto_perform(v)
 * (batch_gemm.hpp:1410-1455) inside EffectiveHamiltonian::perturbative_noise (effective_hamiltonian.hpp:252-423).
 * One record is one xgemm slot (batch_gemm.hpp:289-320):
 *     C (m x n) += alpha * opA(A)(m x k) * opB(B)(k x n)                beta = 1
 * A and B each live either in the operator arena or in the INPUT vector (the wavefunction); C is a window of the
 * OUTPUT vector (the perturbed wavefunctions).  The compiled plan is executed with b2x_plan_execute(plan, input,
 * output, scale, ...):  output += scale * sum of records. */
typedef struct b2x_gemm {
    int32_t m, n, k;
    int32_t lda, ldb, ldc;
    uint8_t ta, tb;       /* 0 = no transpose, 1 = transpose */
    uint8_t a_src, b_src; /* 0 = operator arena, 1 = input vector */
    uint32_t reserved;
    double alpha;
    uint64_t a_off, b_off; /* element offsets into the buffer a_src / b_src selects */
    uint64_t c_off;        /* element offset into the output vector */
} b2x_gemm;
int b2x_gemm_plan_create(b2x_plan **out, const b2x_arena *arena, size_t n_gemms, const b2x_gemm *gemms,
                         size_t in_len, size_t out_len, const b2x_plan_options *opt);

/* diagonal of H_eff (Olsen preconditioner of Davidson) --------------------------------------------------
 * Replaces: the rank-1 products recorded by BatchGEMMSeq / AdvancedGEMM::tensor_product_diagonal and
 * three_tensor_product_diagonal (src/core/batch_gemm.hpp:507-563, src/core/matrix_functions.hpp:1179-1240),
 * driven by OperatorFunctions::tensor_product_diagonal (src/core/operator_functions.hpp:211-328).
 * One term:  C[r][c] += alpha * A[a_off + r*a_stride] * B[b_off + c*b_stride],  r < m, c < n, where C is a window
 * (c_off, ldc) of the diagonal vector and A / B walk the diagonals of two operator blocks (stride = ld + 1). */
typedef struct b2x_diag_term {
    int32_t m, n;
    int32_t a_stride, b_stride;
    int32_t ldc, reserved;
    double alpha;
    uint64_t a_off, b_off; /* arena element offsets */
    uint64_t c_off;        /* element offset into the diagonal vector */
} b2x_diag_term;
/* diag += sum of terms.  on_device != 0: diag is a device pointer.  Deterministic (no atomics). */
int b2x_diag_build(const b2x_arena *arena, size_t n_terms, const b2x_diag_term *terms, size_t diag_len, double *diag,
                   int on_device, void *stream);

/* element-wise block products (blocking of the environments; operator sums) ----------------------------------
 * Replaces: the k = 1 GEMM groups AdvancedGEMM::tensor_product records for c = a (x) b when one factor is 1 x 1 (the
 * site operators of a quantum-chemistry MPO) or block by block otherwise (src/core/batch_gemm.hpp:431-505, executed by
 * GMatrixFunctions::tensor_product, src/core/matrix_functions.hpp:1117-1177, under OperatorFunctions::tensor_product,
 * src/core/operator_functions.hpp:672-711 and TensorFunctions::left_contract / right_contract,
 * src/core/tensor_functions.hpp:2842-2885), and BatchGEMM::iadd (batch_gemm.hpp:313-317; OperatorFunctions::iadd,
 * operator_functions.hpp:126-174).  One term:
 *     C[r][c] += alpha * A[a_off + r*a_rs + c*a_cs] * B[b_off + r*b_rs + c*b_cs]        r < m, c < n
 * where C is a window (c_off, ldc) of the OUTPUT vector (the enlarged operators).  A scalar factor has rs = cs = 0, a
 * transposed block swaps rs and cs.  Sources: 0 = operator arena, 1 = INPUT vector, 2 = the constant 1.0. */
typedef struct b2x_outer_term {
    int32_t m, n;
    int32_t a_rs, a_cs;
    int32_t b_rs, b_cs;
    int32_t ldc;
    uint8_t a_src, b_src;
    uint8_t reserved[2];
    double alpha;
    uint64_t a_off, b_off;
    uint64_t c_off;
} b2x_outer_term;
/* The same in two steps, for a list whose STRUCTURE is known before its data are (the blocking of the next site, compiled while
 * the device is busy with this one): create compiles the list for an arena of `arena_len` elements and uploads the work list;
 * execute launches out_dev += sum of terms on device vectors (asynchronous on `stream`); destroy waits for the device. */
typedef struct b2x_outer_plan b2x_outer_plan;
int b2x_outer_plan_create(b2x_outer_plan **out, uint64_t arena_len, size_t n_terms, const b2x_outer_term *terms, size_t in_len,
                          size_t out_len);
int b2x_outer_plan_execute(const b2x_outer_plan *plan, const b2x_arena *arena, const double *in_dev, double *out_dev,
                           void *stream);
int b2x_outer_plan_destroy(b2x_outer_plan *plan);
/* out += sum of terms.  on_device != 0: in / out are device pointers.  Deterministic (no atomics). */
int b2x_outer_build(const b2x_arena *arena, size_t n_terms, const b2x_outer_term *terms, const double *in, size_t in_len,
                    size_t out_len, double *out, int on_device, void *stream);

/* sum-MPO communicator ------------------------------------------------------------------------------------------
 * Replaces: ParallelCommunicator<S>::allreduce_sum(double*, size_t), broadcast(double*, size_t, int), barrier()
 * (src/core/parallel_rule.hpp:55, 74, 128; MPI bodies src/core/parallel_mpi.hpp:300-309, 133-141, 125-132) for the
 * device-resident vectors of this path: ParallelTensorFunctions::operator() sums the partial H.psi of the ranks
 * (src/core/parallel_tensor_functions.hpp:51-55), the diagonal is summed once per site (:853), Davidson broadcasts from
 * the root.  Transport: RCCL (over xGMI inside a node); one process per GPU; the communicator binds to the device that
 * is current when it is created (call b2x_device_init first).  Rendezvous without MPI (b2x_comm_init_session): rank 0
 * removes whatever is at `id_file` (a path every rank can read — any local directory on one node), writes {magic, nonce,
 * 128-byte RCCL id} there and removes the file again once the communicator exists; the other ranks wait for a file that
 * carries THEIR `nonce` (B2X_COMM_TIMEOUT_S, default 120 s) — the file a crashed earlier run left behind is rejected, not
 * joined.  `nonce` = any number all ranks of one launch share and earlier launches did not (the launcher's run id, a
 * time stamp broadcast by its control plane); 0 accepts any file: b2x_comm_init(id_file) is that form, safe only where
 * no earlier file can exist at the path.  A launcher with its own broadcast uses b2x_comm_unique_id + b2x_comm_init_id.
 * Collectives are asynchronous and ordered on `stream` (hipStream_t, NULL = default stream): they run on the
 * communicator's own stream once the work queued on `stream` so far is done, and `stream` resumes after them. */
typedef struct b2x_comm b2x_comm;
int b2x_comm_init(b2x_comm **out, int rank, int size, const char *id_file);
int b2x_comm_init_session(b2x_comm **out, int rank, int size, const char *id_file, uint64_t nonce);
int b2x_comm_unique_id(void *id128);                                   /* rank 0: a fresh 128-byte id */
int b2x_comm_init_id(b2x_comm **out, int rank, int size, const void *id128);
int b2x_comm_rank(const b2x_comm *c, int *rank, int *size);
int b2x_allreduce_sum(b2x_comm *c, double *dev, size_t n, void *stream); /* in place, SUM over ranks */
int b2x_broadcast(b2x_comm *c, double *dev, size_t n, int root, void *stream);
int b2x_barrier(b2x_comm *c);                                          /* returns when every rank has arrived */
int b2x_comm_destroy(b2x_comm *c);

/* device-resident vector algebra for Davidson (all pointers are device pointers) ------------ */
int b2x_vec_dot(const double *x, const double *y, size_t n, double *host_result, void *stream);
int b2x_vec_axpy(double a, const double *x, double *y, size_t n, void *stream);   /* y += a x */
int b2x_vec_scal(double a, double *x, size_t n, void *stream);
int b2x_vec_copy(const double *x, double *y, size_t n, void *stream);
int b2x_vec_zero(double *x, size_t n, void *stream);
/* davidson_precondition (iterative_matrix_functions.hpp:66-72, used at :1084-1085 for DavidsonTypes::DavidsonPrecond):
 * q[i] /= (shift - diag[i]) when |shift - diag[i]| > 1e-12   (shift = the current Ritz value ld) */
int b2x_vec_precondition(double *q, const double *diag, double shift, size_t n, void *stream);
/* first half of olsen_precondition (iterative_matrix_functions.hpp:93-108): t = c; then, where
 * |ld - diag[i]| > 1e-12:  t[i] /= ld - diag[i],  q[i] /= ld - diag[i].  (The caller finishes with
 * q += -(c.q)/(c.t) * t using b2x_vec_multi_dot + b2x_vec_axpy.) */
int b2x_vec_olsen_prepare(double *q, double *t, const double *c, const double *diag, double ld, size_t n,
                          void *stream);
/* dst[dst_off[i] + k] = src[src_off[i] + k], k < len[i], for i < n_ranges (offsets and lengths in elements; ranges must not
 * overlap in dst): one launch instead of n_ranges copies — the operator blocks of the enlarged blocks gathered into the arena
 * of an effective Hamiltonian.  Returns when the copy is done. */
int b2x_vec_gather(double *dst, const double *src, size_t n_ranges, const uint64_t *dst_off, const uint64_t *src_off,
                   const uint64_t *len, void *stream);
/* the same with the scaled residual written to q_out and q left as it is (q_out == q: b2x_vec_olsen_prepare): lets the
 * Davidson step take |q|^2, c.q_out, c.t and the projections of q_out and t on the basis from ONE b2x_vec_pair_dots */
int b2x_vec_olsen_prepare_to(const double *q, double *q_out, double *t, const double *c, const double *diag, double ld,
                             size_t n, void *stream);
/* result[j] = <us[j], vs[j]> for j < n_pairs <= 128 (host arrays of device pointers): every dot product of a Davidson step
 * in one launch and one host round trip (the reference's loop of dot calls, iterative_matrix_functions.hpp:1143-1144,
 * would be one device synchronisation each) */
int b2x_vec_pair_dots(const double *const *us, const double *const *vs, int n_pairs, size_t n, double *host_result,
                      void *stream);
/* One pass over a Davidson basis b_j and its images s_j (m <= 64): the Ritz vector x = sum alpha_j b_j, its residual
 * q = sum alpha_j s_j - theta x, and the first half of olsen_precondition on both (as b2x_vec_olsen_prepare_to, c = x, ld = theta):
 * q2 = q / (theta - diag), t = x / (theta - diag). */
int b2x_vec_ritz_olsen(const double *const *bs, const double *const *ss, int m, const double *alpha, double theta,
                       const double *diag, double *x, double *q, double *q2, double *t, size_t n, void *stream);
/* out = (v - sum_{j<m} <b_j, v> b_j) / norm for an orthonormal set b (m <= 63): the second Gram-Schmidt pass and the
 * normalisation of a new Davidson basis vector with the coefficients kept on the device — no host round trip; asynchronous on
 * `stream`.  When the norm is not safely positive (v in the span of b to rounding) the unnormalised difference is written and a
 * flag raised that b2x_vec_gs_status reports after the caller's next wait on the stream (reset != 0 clears it). */
int b2x_vec_gs_finish(const double *const *bs, int m, const double *v, double *out, size_t n, void *stream);
int b2x_vec_gs_status(int *degenerate, int reset);
/* gram[j] = <vs[j], x> for j < nv; vs = nv device pointers (host array of device pointers) */
int b2x_vec_multi_dot(const double *const *vs, int nv, const double *x, size_t n, double *host_result,
                      void *stream);
/* y = sum_j coef[j] * vs[j] */
int b2x_vec_lincomb(const double *const *vs, int nv, const double *coef, double *y, size_t n,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif
