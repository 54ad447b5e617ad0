// b2x_plan.hpp — host-side compiler from the reference's GEMM-pair list to the device work list.
//
// The reference replays pairs one by one with thread-private copies of psi' and a tree reduction
// (BatchGEMMSeq::operator(), src/core/batch_gemm.hpp:1606-1682).  On MI355X the list is instead
// re-segmented by OUTPUT: psi' is cut into tiles, every pair is split into the parts that touch a
// tile, each tile's part list is chopped into work items of similar cost, and one wave / workgroup per
// item accumulates its parts in registers and writes a partial slab; a second kernel sums the
// slabs of a tile in fixed order into psi'.  No atomics, bitwise reproducible.
// Two execution paths: the fused wave kernel (small sectors: W never leaves the registers) and the two-stage
// grouped-GEMM path (W through an HBM scratch, in super-steps).  On the second path the compiler also uses the algebra
// of V += alpha op(Z) X op(Y) — association per pair, stage-0 products shared between pairs, sums of products before a
// common factor — unless b2x_plan_options.keep_order asks for the reference's order.
#pragma once
#include "../../include/b2x.h"
#include <cstdint>
#include <string>
#include <vector>

namespace b2x {

// One (pair x tile) part.  Everything the kernel needs, 72 bytes.
struct DPart {
    uint64_t x_off; // psi   offset of X[k1lo][0]
    uint64_t y_off; // arena offset of op(Y)[0][c_lo]
    uint64_t z_off; // arena offset of op(Z)[r_lo][k1lo]
    double alpha;   // alpha0 * alpha1
    int32_t ldx;    // X leading dimension
    int32_t sky, scy; // op(Y)[k][c] = Y[k*sky + c*scy]
    int32_t srz, skz; // op(Z)[r][k] = Z[r*srz + k*skz]
    int32_t k0;       // stage-0 inner dimension
    int32_t k1;       // rows of X = inner dimension of stage 1 (the kernel walks it in chunks of 16*K1F)
    int16_t mr, nc;   // output rows / cols of this part
    int16_t tr0, tc0; // offset of the part's output window inside the tile
};
static_assert(sizeof(DPart) == 72, "DPart layout");

struct DItem {
    uint32_t part_begin, part_end;
    uint64_t slab_off; // element offset of this item's partial slab
    int32_t rows, cols; // tile dims (slab is rows x cols, ld = cols)
};
static_assert(sizeof(DItem) == 24, "DItem layout");

struct DTile {
    uint64_t sigma_off; // psi' offset of the tile's top-left element
    uint64_t slab_off;  // first slab of this tile; slabs of the tile are consecutive
    int32_t ld;         // psi' leading dimension of the enclosing sector
    int32_t rows, cols;
    int32_t n_items;
};
static_assert(sizeof(DTile) == 32, "DTile layout");

// ---- two-stage path (large sectors): both stages are plain grouped GEMMs -------------------------
// One K-segment of a tile's accumulation:  C[0:mr, tc0:tc0+nc] += alpha * A(mr x K) * B(K x nc)
// (row tiles are cut at every window boundary, so a segment always starts at the tile's first row)
struct GSeg {
    uint64_t a_off, b_off; // element offsets into the buffer selected by a_src / b_src
    int32_t a_sr, a_sk;    // A[r][k] = bufA[a_off + r*a_sr + k*a_sk]
    int32_t b_sk, b_sc;    // B[k][c] = bufB[b_off + k*b_sk + c*b_sc]
    int32_t K;
    int32_t mr, nc, tc0;
    int32_t a_src, b_src; // 0 = operator arena, 1 = psi / input vector, 2 = W scratch
    double alpha;         // applied to the B fragments by the SCALED kernel variant only (single-GEMM lists);
                          // the H.psi plan keeps 1.0 here and scales once per tile (GItem::alpha)
};                        // all fields 32/64-bit so the descriptor is fetched with scalar loads
static_assert(sizeof(GSeg) == 64, "GSeg layout");

struct GItem {
    uint32_t seg_begin, seg_end;
    uint64_t out_off; // element offset in the slab buffer (out_kind 0) or the W scratch (out_kind 1)
    double alpha;     // factor applied when the tile is stored
    int32_t out_ld, rows, cols;
    uint32_t out_kind;
};
static_assert(sizeof(GItem) == 40, "GItem layout");

// ---- element-wise block products (blocking) ---------------------------------------------------------------------
// The windows of one output sector are cut along all their row / column boundaries into CELLS; a cell lists the terms
// that cover it with operand offsets already moved to the cell origin, so the kernel walks a cell's list without any
// window test.  A cell is covered by TILES of 64 columns x rpt rows (tile id = strip * n_col_segments + segment); a
// work unit is a range of tiles of one cell and is executed by ONE WAVE (four units per workgroup).
struct OEntry {
    uint64_t a_off, b_off;
    double alpha;
    int32_t a_rs, a_cs, b_rs, b_cs;
    int32_t a_src, b_src;
};
static_assert(sizeof(OEntry) == 48, "OEntry layout");
struct OWork {
    uint64_t out_off; // output offset of the cell's (0, 0)
    int32_t ld, rows, cols;
    int32_t rpt;                   // rows per tile
    uint32_t t_begin, t_end;       // tile range of the cell handled by this unit
    uint32_t entry_begin, entry_end;
};
static_assert(sizeof(OWork) == 40, "OWork layout");
static const int kOuterTileCols = 64;

// The grouped-GEMM kernel fetches A operands as 16-byte granules and may reach ONE element behind an operand
// (gg_body::lane_offsets).  Inside a buffer that element is neighbouring data that meets a zeroed B lane.  Where it
// would lie OUTSIDE the source buffer (an operand that ends exactly at the end of an adopted arena or of the caller's
// psi), the operand is copied into plan-owned memory that has a zero behind it, and the segment reads the copy: the
// caller's buffers need no slack.
struct StageCopy {
    uint32_t src;     // 0 = operator arena, 1 = input vector
    uint64_t src_off; // element offset in the source buffer
    uint64_t dst_off; // element offset in the scratch
    uint64_t len;
};

// stage 0 of a batch of pairs -> W scratch; stage 1 consumes it; reduce adds the slabs into psi'.
// All items of a stage share one launch (range [s?_v[0], s?_v[kGGVariants])); every workgroup picks the kernel body
// for its tile height (rows rounded up to 16).
static const int kGGVariants = 4;
struct SuperStep {
    uint32_t s0_v[kGGVariants + 1];
    uint32_t s1_v[kGGVariants + 1];
    uint32_t tile_begin, tile_end; // DTile range (two-stage tile list)
    uint32_t sum_begin, sum_end;   // OWork range of the sum pass between the stages (CompiledPlan::sum_work)
};

// tile of one workgroup of the grouped-GEMM kernel: kGGTileN / (16 * kGGCF) waves, each kGGCF column fragments wide
static const int kGGTileM = 128, kGGTileN = 128;
static const int kGGCF = 2;
static const int kGGRowUnit = 16; // tile heights are multiples of one MFMA row fragment
static const int kGGNarrowFrags = 2, kGGNarrowN = 32; // 1-wave workgroups: tiles of up to kGGNarrowFrags row fragments x kGGNarrowN columns
// largest leading dimension of any operand (elements): per-lane byte offsets inside a tile are 32-bit, products 24-bit
static const int kMaxLeadingDim = 1 << 22;
static const int kGGShortFrags = 3, kGGMidFrags = 5; // the short class holds tiles of up to 3 row fragments (4 waves/SIMD) or, for plans of
                                                      // mid-height sectors, up to 5 (3 waves/SIMD): CompiledPlan::short_frags // tiles of up to this many row fragments run on the low-register kernel instantiation

// kernel classes of the fused path.  nw = tile width / 16, tmf = tile height / 16, k1f = k1 chunk / 16.
// All fused classes run hpsi_wave — one WAVE per work item, operands straight from L2 into MFMA fragments, no LDS,
// no barriers (the wavefront-level grouped GEMM for the many small symmetry blocks).  Sectors taller than 128 rows
// (or wider / deeper than one tile handles well) go to the two-stage grouped-GEMM path instead.
struct KClass {
    int nw, tmf, k1f;
    bool wave;
};
static const KClass kClasses[] = {{2, 2, 2, true}, {2, 4, 2, true}, {2, 8, 2, true}};
static const int kNumClasses = 3;

struct ClassWork {
    std::vector<DPart> parts;
    std::vector<DItem> items;
};

struct CompiledPlan {
    ClassWork cls[kNumClasses];
    std::vector<DTile> tiles;
    uint64_t slab_elems = 0;
    uint64_t cls_macs[kNumClasses] = {0, 0, 0}; // MACs the kernels of each class execute
    b2x_plan_stats stats{};
    // two-stage path
    std::vector<GSeg> gsegs;
    std::vector<GItem> gitems;
    std::vector<DTile> gtiles;
    std::vector<SuperStep> steps;
    uint64_t scratch_elems = 0, gslab_elems = 0;
    bool fallback = false; // windows could not be segmented -> generic atomic kernel
    std::string fallback_reason;
    int cap_units = 0; // tallest tile in row fragments when the compiler caps it below kGGTileM / 16 (0: no cap)
    int short_frags = kGGShortFrags; // tallest tile (row fragments) of the short class of this plan: kGGShortFrags or kGGMidFrags
    bool short_narrow = false; // the short tile class ([v[1], v[last]) of every stage) holds tiles of <= 32 x 32: 1-wave workgroups
    bool seg_scaled = false; // single-GEMM list: segments carry their own alpha (gg_kernel SCALED variant)
    int gg_tile_n = kGGTileN; // column width of the grouped-GEMM tiles of this plan: 128 (4 waves) or 64 (2 waves)
    // sum pass of the two-stage path: S = sum_i alpha_i W_i for pairs that multiply the same operator block into the same
    // psi' window (scratch -> scratch, element-wise; OWork::ld < 0 marks "assign" instead of "accumulate")
    std::vector<OWork> sum_work;
    std::vector<OEntry> sum_entries;
    // operator pre-sums, built once when the plan is created (arena -> the first elements of the scratch)
    std::vector<OWork> aux_work;
    std::vector<OEntry> aux_entries;
    // operands staged in the tail of the scratch (stage_residual_reads): arena sources are copied once when the plan is
    // uploaded, input-vector sources at the start of every execute
    std::vector<StageCopy> stage;
    std::vector<uint64_t> scratch_pads; // the padding element behind every scratch slot (only the B2X_DEBUG_POISON knob needs the list)
};

// ---- diagonal build ----------------------------------------------------------------------------------------
struct DiagTermD {
    uint64_t a_off, b_off;
    double alpha;
    int32_t row0, col0, m, n; // window inside the component
    int32_t a_stride, b_stride;
};
struct DiagComp {
    uint64_t base; // diag offset of the component's (0, 0)
    int32_t ld, rows, cols;
    uint32_t term_begin, term_end;
};
int compile_outer(size_t n_terms, const b2x_outer_term *terms, size_t in_len, size_t out_len, uint64_t arena_len,
                  std::vector<OWork> &work, std::vector<OEntry> &entries, std::string &err);

// groups the terms by output sector (overlapping windows), keeping plan order inside a sector
int compile_diag(size_t n_terms, const b2x_diag_term *terms, size_t diag_len, uint64_t arena_len,
                 std::vector<DiagComp> &comps, std::vector<DiagTermD> &dterms, std::string &err);

// returns 0 / B2X_ERR_INVALID (err filled)
// arena_cap: elements of the arena buffer that may be READ (>= arena_len: an owned arena has slack behind it)
int compile_plan(size_t n_pairs, const b2x_pair *pairs, size_t psi_len, size_t sigma_len, uint64_t arena_len,
                 uint64_t arena_cap, const b2x_plan_options *opt, CompiledPlan &out, std::string &err);

// single-GEMM list (b2x_gemm records) -> one super-step of stage-1-shaped items (no stage 0, no W scratch)
int compile_gemm_list(size_t n_gemms, const b2x_gemm *gemms, size_t in_len, size_t out_len, uint64_t arena_len,
                      uint64_t arena_cap, const b2x_plan_options *opt, CompiledPlan &out, std::string &err);


} // namespace b2x
