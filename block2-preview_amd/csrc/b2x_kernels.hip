// b2x_kernels.hip — CDNA4 (gfx950) kernels of the H·psi plan replay.
//
//   hpsi_wave<TMF,K1F,CF>   fused pair kernel, one WAVE per work item (sectors up to 128 rows): stage 0
//                           W = alpha X op(Y) and stage 1 V += op(Z) W on v_mfma_f64_16x16x4_f64; the stage-0
//                           accumulator IS the stage-1 B operand (C/D layout row = (lane>>4) + 4*reg, col = lane&15
//                           equals the B-fragment layout of k-step `reg`), so W never leaves the register file
//                           (the reference round-trips it through a per-thread work array, batch_gemm.hpp:1630-1635)
//   gg_kernel<CF,NW,KC,SB>  grouped GEMM of the two-stage path (tall sectors) and of single-GEMM lists: LDS-DMA staged A,
//                           register B; shipped as <2,4,16>: 128x128 tiles, 4 waves, two workgroups per CU
//   hpsi_reduce             psi'[tile] += scale * sum of the tile's partial slabs, fixed order (no atomics)
//   hpsi_generic            per-pair atomic kernel: on-device cross-check / fallback for unsegmentable plans
//   diag_build_k, vec_*     diagonal of H_eff and BLAS-1 for the device-resident Davidson
#include "b2x_kernels.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace b2x {

// Which argument of a kernel is psi / sigma / scale: recorded by the launchers per kernel function, read back by the HIP-graph
// replay of b2x_capi.cpp, which patches exactly those arguments of its kernel nodes (never guessed from pointer values:
// psi is argument 3 of gg_kernel but argument 4 of hpsi_wave).
namespace {
std::mutex g_slots_mu;
std::unordered_map<const void *, KernelArgSlots> g_slots;
inline void note_slots(const void *func, int psi, int sigma, int scale) {
    std::lock_guard<std::mutex> lk(g_slots_mu);
    g_slots[func] = KernelArgSlots{psi, sigma, scale};
}
} // namespace
bool kernel_arg_slots(const void *func, KernelArgSlots *out) {
    std::lock_guard<std::mutex> lk(g_slots_mu);
    auto it = g_slots.find(func);
    if (it == g_slots.end())
        return false;
    *out = it->second;
    return true;
}

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d4u __attribute__((ext_vector_type(4), aligned(8))); // 4 consecutive doubles, 8-byte aligned

// ------------------------------------------------------------------------------------------------
// hpsi_wave<TMF, K1F, CF>: the wavefront-level grouped GEMM for small symmetry blocks.  ONE WAVE per work
// item (a slice of the part list of one (TMF*16) x (CF*16) psi' tile); four independent waves share a
// workgroup only for dispatch.  No LDS and no barriers: every MFMA operand is loaded from L2 straight into
// its fragment register (A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15]), W = alpha X op(Y)
// is chained from the stage-0 accumulator into the stage-1 B operand inside the register file, and the
// memory latency of these short K loops is hidden by the other waves of the SIMD (<= 128 VGPRs).
template <int TMF, int K1F, int CF>
__global__ __launch_bounds__(256, 2) void hpsi_wave(const DPart *__restrict__ parts, const DItem *__restrict__ items,
                                                  uint32_t n_items, const double *__restrict__ arena,
                                                  const double *__restrict__ psi, double *__restrict__ slabs) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    // wave-uniform item index in an SGPR: descriptors are fetched with scalar loads, branches stay uniform
    const uint32_t it = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (it >= n_items)
        return;
    const DItem item = items[it];

    v4d acc[TMF][CF];
#pragma unroll
    for (int f = 0; f < TMF; f++)
#pragma unroll
        for (int q = 0; q < CF; q++)
            acc[f][q] = v4d{0.0, 0.0, 0.0, 0.0};

    DPart Pn = parts[item.part_begin]; // descriptors are prefetched one part ahead (scalar loads)
    for (uint32_t pi = item.part_begin; pi < item.part_end; pi++) {
        const DPart P = Pn;
        Pn = parts[min(pi + 1, item.part_end - 1)];
        const int k0 = P.k0, mr = P.mr, nc = P.nc, tr0 = P.tr0, tc0 = P.tc0;
        const int f_lo = tr0 >> 4, f_hi = (tr0 + mr + 15) >> 4;
        // per-lane column of every column fragment (clamped offset + validity)
        uint32_t yoff[CF];
        bool cok[CF];
#pragma unroll
        for (int q = 0; q < CF; q++) {
            int cc = q * 16 + c - tc0;
            cok[q] = cc >= 0 && cc < nc;
            yoff[q] = (uint32_t)min(max(cc, 0), nc - 1) * (uint32_t)P.scy;
        }
        const double *Y = arena + P.y_off;
        for (int k1lo = 0; k1lo < P.k1; k1lo += K1F * 16) {
            const int k1c = min(K1F * 16, P.k1 - k1lo);
            const double *X = psi + P.x_off + (int64_t)k1lo * P.ldx;
            const double *Z = arena + P.z_off + (int64_t)k1lo * P.skz;
            v4d w[K1F][CF];
#pragma unroll
            for (int f = 0; f < K1F; f++)
#pragma unroll
                for (int q = 0; q < CF; q++)
                    w[f][q] = v4d{0.0, 0.0, 0.0, 0.0};
            // MFMA operand lanes are loaded in a permuted order so that every lane reads CONTIGUOUS k:
            //  * stage 0, k-step s of a 16-k trip: lane group g supplies k = kb + 4g + s (A and B alike), i.e. a lane
            //    reads X[row][kb+4g .. +3] (32 B) instead of four 8-byte gathers;
            //  * the X row fed to MFMA row i is f*16 + pi(i), pi(i) = 4*(i&3) + (i>>2), which makes accumulator
            //    register r of lane group g hold W row f*16 + 4g + r — so in stage 1 lane group g needs
            //    op(Z)[row][f1*16 + 4g .. +3], again 32 contiguous bytes.
            // Rows >= k1c read a valid row; their W rows are neutralised in stage 1 (op(Z) zeroed for k >= k1c).
            const int pi_c = 4 * (c & 3) + (c >> 2);
            uint32_t xoff[K1F];
#pragma unroll
            for (int f = 0; f < K1F; f++)
                xoff[f] = (uint32_t)min(f * 16 + pi_c, k1c - 1) * (uint32_t)P.ldx;
            const bool ywide = (P.sky == 1); // op(Y) stored k-contiguous
            // ---------------- stage 0: W = X * op(Y), 16 k per trip: all loads of the trip first ----------------
            for (int kb = 0; kb < k0; kb += 16) {
                double b[4][CF], a[4][K1F];
                if (kb + 16 <= k0) { // full trip: wide loads, no k masking
                    const uint32_t kk = (uint32_t)(kb + 4 * g);
#pragma unroll
                    for (int f = 0; f < K1F; f++) {
                        const d4u v = *(const d4u *)(X + xoff[f] + kk);
#pragma unroll
                        for (int s = 0; s < 4; s++)
                            a[s][f] = v[s];
                    }
                    if (ywide) {
#pragma unroll
                        for (int q = 0; q < CF; q++) {
                            const d4u v = *(const d4u *)(Y + yoff[q] + kk);
#pragma unroll
                            for (int s = 0; s < 4; s++)
                                b[s][q] = cok[q] ? v[s] : 0.0;
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; s++)
#pragma unroll
                            for (int q = 0; q < CF; q++) {
                                double v = Y[(kk + s) * (uint32_t)P.sky + yoff[q]];
                                b[s][q] = cok[q] ? v : 0.0;
                            }
                    }
                } else { // k tail: clamped 8-byte loads, k >= k0 zeroed on the B side
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const int k = kb + 4 * g + s;
                        const uint32_t kc = (uint32_t)min(k, k0 - 1);
#pragma unroll
                        for (int q = 0; q < CF; q++) {
                            double v = Y[kc * (uint32_t)P.sky + yoff[q]];
                            b[s][q] = (cok[q] && k < k0) ? v : 0.0;
                        }
#pragma unroll
                        for (int f = 0; f < K1F; f++)
                            a[s][f] = X[xoff[f] + kc];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int f = 0; f < K1F; f++)
#pragma unroll
                        for (int q = 0; q < CF; q++)
                            w[f][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][f], b[s][q], w[f][q], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < K1F; f++)
#pragma unroll
                for (int q = 0; q < CF; q++)
                    w[f][q] *= P.alpha;
            // ---------------- stage 1: V += op(Z) * W ----------------
            // per-lane op(Z) row of every row fragment (clamped) and its validity
            uint32_t zoff[TMF];
            bool rok[TMF];
#pragma unroll
            for (int f = 0; f < TMF; f++) {
                const int rr = f * 16 + c - tr0;
                rok[f] = rr >= 0 && rr < mr;
                zoff[f] = (uint32_t)min(max(rr, 0), mr - 1) * (uint32_t)P.srz;
            }
            const bool zwide = (P.skz == 1); // op(Z) stored k-contiguous
#pragma unroll
            for (int f1 = 0; f1 < K1F; f1++)
                if (f1 * 16 < k1c) {
                    // a[r] = op(Z)[row f*16+c][f1*16 + 4g + r], one row fragment at a time (8 VGPRs in flight per f)
                    const bool zfull = zwide && f1 * 16 + 16 <= k1c;
#pragma unroll
                    for (int f = 0; f < TMF; f++)
                        if (f >= f_lo && f < f_hi) {
                            double a[4];
                            if (zfull) {
                                const d4u v = *(const d4u *)(Z + zoff[f] + (uint32_t)(f1 * 16 + 4 * g));
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    a[r] = rok[f] ? v[r] : 0.0;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int k = f1 * 16 + 4 * g + r;
                                    double v = Z[zoff[f] + (uint32_t)min(k, k1c - 1) * (uint32_t)P.skz];
                                    a[r] = (rok[f] && k < k1c) ? v : 0.0;
                                }
                            }
#pragma unroll
                            for (int r = 0; r < 4; r++)
#pragma unroll
                                for (int q = 0; q < CF; q++)
                                    acc[f][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], w[f1][q][r], acc[f][q], 0, 0, 0);
                        }
                }
        }
    }
    double *slab = slabs + item.slab_off;
#pragma unroll
    for (int q = 0; q < CF; q++) {
        const int col = q * 16 + c;
        if (col < item.cols) {
#pragma unroll
            for (int f = 0; f < TMF; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int row = f * 16 + 4 * r + g;
                    if (row < item.rows)
                        slab[(int64_t)row * item.cols + col] = acc[f][q][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gg_body<TMF, CF, NW, KC, SB>: grouped GEMM of the two-stage path (large sectors) and of single-GEMM lists.  One
// workgroup of NW waves owns a (TMF*16) x (NW*CF*16) output tile and walks a list of K-segments
//     C[window] += A(mr x K) * B(K x nc).
//   stage 0 items: one segment, A = X (psi), B = op(Y) (arena), tile stored (x alpha) into the W scratch
//   stage 1 items: many segments, A = op(Z) (arena), B = W (scratch), tile stored into a partial slab
// Wave w owns CF column fragments x all TMF row fragments (acc = TMF*CF*8 VGPRs, in VGPRs: no AGPR traffic).
// A chunks (TM x 16 k) go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no staging registers,
// no ds_write), double-buffered, one barrier per chunk; the DMA destination is lane-linear, so the bank
// swizzle is applied on the per-lane SOURCE address and again on the ds_read address.  B fragments are
// private to a wave and are prefetched one chunk ahead straight into registers.  Row tiles are cut at the
// row-slice boundaries of the sector, so a segment's rows always cover its tile: only columns (B side)
// and the k tail need masking, and both are applied to the B registers.
template <int TMF, int CF, int NW, int KC, bool SB, int CW = CF> // CW = column fragments per wave in the tile LAYOUT, CF <= CW of them active
__device__ __forceinline__ void gg_body(const GItem &item, const GItem *item_ptr, double *lds, const GSeg *__restrict__ segs,
                                        const double *__restrict__ arena, const double *__restrict__ psi,
                                        double *__restrict__ scratch, double *__restrict__ slabs) {
    constexpr int TM = TMF * 16, NT = NW * 64;
    constexpr int KS = KC / 4;           // k-steps (MFMA K = 4) per chunk
    constexpr int ABUF = TM * KC;        // doubles per LDS buffer (unpadded: the DMA image is lane-linear)
    constexpr int NG = TM * KC / 2;      // 16-byte granules per chunk
    constexpr int NI = (NG + NT - 1) / NT; // DMA instructions per thread per chunk

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // in an SGPR: the LDS-DMA destinations (M0) stay scalar
    const int g = lane >> 4, c = lane & 15;

    v4d acc[TMF][CF];
#pragma unroll
    for (int f = 0; f < TMF; f++)
#pragma unroll
        for (int q = 0; q < CF; q++)
            acc[f][q] = v4d{0.0, 0.0, 0.0, 0.0};

    double bnxt[CF][KS], bcur[CF][KS];
    // ---- per-segment state ---------------------------------------------------------------------------
    // The MFMA f64 pipe and the vector ALU of a SIMD do not run side by side: every VALU instruction between two MFMAs
    // costs matrix time (probe: the same MFMA stream ran at 74.5 TFLOP/s with no per-chunk VALU work, 68 with the
    // copies and control of this loop, 62 with per-load address arithmetic).  A load address is therefore a SCALAR base
    // that advances with the chunk (SALU) plus a per-lane byte offset (va, vb) that is computed when a segment is
    // entered and again only for a segment's last, partial chunk, whose k tail needs clamping.
    const double *sA = arena, *sB = arena;
    uint32_t va[NI];     // per DMA instruction: byte offset of this lane's granule from the chunk's A base
    uint32_t vb[CF][KS]; // byte offset of this lane's B element from the k-step's B base
    uint32_t astep = 1, bstep = 1;
    uint32_t bmask = 0;  // B mask of the chunk that is fetched next (columns; k tail)
    bool tail = false;   // the offsets in va / vb are those of the partial last chunk
    double salpha = 1.0; // SB (single-GEMM lists): per-segment factor, folded into the B fragments
    bool s_kmaj = false, cols_full = false, bmasked = false;

    // A 16-byte A granule may reach ONE element behind its operand: a k-contiguous operand whose K is not a multiple of
    // the chunk depth is read up to element K of every row, a row-contiguous one whose row count is not a multiple of 16
    // up to row mr of every k.  Inside a buffer that element is the caller's own neighbouring (finite) data and meets a
    // zeroed B lane; where it would lie OUTSIDE the buffer — the operand ends exactly at the end of an adopted arena or
    // of psi — the plan compiler hands the kernel a staged copy instead (stage_residual_reads, b2x_plan.cpp), so no
    // buffer needs slack.  (An exact variant of these offsets — last granule fetched one element earlier, B slots and
    // the tile store following it — was measured: +13 % time at M=250, +6 % at M=500, because this per-segment
    // arithmetic IS the bottleneck of short segments; profiles/README.md.)
    // Per-lane offsets of the chunk at k offset kb of segment S: unclamped for a full chunk (valid for every full chunk
    // of the segment), clamped to the last k for the partial one.  This arithmetic runs once or twice per SEGMENT, and for
    // the short segments of small bond dimensions (one or two chunks of 8-16 MFMAs) it used to cost more vector-ALU time
    // than the MFMAs themselves (~80 VALU per call + ~50 in a masked commit; the fp64 MFMA pipe and the VALU do not
    // overlap).  What depends on the lane alone is therefore computed once per ITEM (lane_consts; kept in registers for
    // tiles up to 64 rows, recomputed for the taller ones whose accumulators leave no room), the k clamp is a scalar
    // (min with 0xFFFF for a full chunk: no select), products are 24-bit multiply-adds, and the B mask is a bit count.
    constexpr bool HOIST = NI <= 2; // (4 NI registers)
    auto lane_consts = [&](int j, uint32_t &row_c, uint32_t &k_ofs, uint32_t &rk, uint32_t &kl) __attribute__((always_inline)) {
        // granule G of the LDS image is written by lane `lane` of DMA instruction (wave, j)
        const int G = (wave * NI + j) * 64 + lane;
        // image [TM rows][KC k]; granule slot gs of a row holds k = 2*(gs ^ swz(row)), +1
        const int row = G / (KC / 2), gs = G % (KC / 2);
        const int swz = KC == 16 ? ((row >> 1) & 7) : (row & 15);
        row_c = (uint32_t)min(row, item.rows - 1), k_ofs = (uint32_t)(2 * (gs ^ swz));
        // fragment-major image [f][k][16 rows]: granule = rows (f*16 + 2p, +1) of one k
        const int f = G / (KC * 8), p2 = G & 7;
        rk = (uint32_t)min(f * 16 + 2 * p2, item.rows - 1), kl = (uint32_t)((G % (KC * 8)) >> 3);
    };
    uint32_t h_rowc[HOIST ? NI : 1], h_kofs[HOIST ? NI : 1], h_rk[HOIST ? NI : 1], h_kl[HOIST ? NI : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < NI; j++)
            lane_consts(j, h_rowc[j], h_kofs[j], h_rk[j], h_kl[j]);
    }
    const int cbase0 = wave * (CW * 16) + c; // this lane's column in the tile (first column fragment)
    auto lane_offsets = [&](const GSeg &S, int kb, bool part) __attribute__((always_inline)) {
        const int n = S.K - kb;                                 // valid k of this chunk (part: n < KC)
        const uint32_t kcl = part ? (uint32_t)(n - 1) : 0xFFFFu; // largest k offset that may be fetched
#pragma unroll
        for (int j = 0; j < NI; j++) {
            uint32_t row_c, k_ofs, rk, kl;
            if constexpr (HOIST)
                row_c = h_rowc[j], k_ofs = h_kofs[j], rk = h_rk[j], kl = h_kl[j];
            else
                lane_consts(j, row_c, k_ofs, rk, kl);
            // (a segment's rows always cover its tile: S.mr == item.rows, which lane_consts clamps to)
            va[j] = (s_kmaj ? __umul24(min(kl, kcl), astep) + rk : __umul24(row_c, (uint32_t)S.a_sr) + min(k_ofs, kcl)) << 3;
        }
        uint32_t kt[KS]; // this lane's k of every k-step, times the k stride of B
#pragma unroll
        for (int s = 0; s < KS; s++) // full chunk: the k-step's 4 s rows are in the scalar base; partial: base = chunk start
            kt[s] = __umul24(min((uint32_t)g + (part ? 4u * s : 0u), kcl), bstep);
        // k-steps in which this lane's k is valid: 4 s + g < n
        const uint32_t kbits = part ? ((1u << ((n - g + 3) >> 2)) - 1u) : ((1u << KS) - 1u);
        bmask = 0;
#pragma unroll
        for (int q = 0; q < CF; q++) {
            const int cc = cbase0 + q * 16 - S.tc0;
            const bool in = (uint32_t)cc < (uint32_t)S.nc;
            const uint32_t co = __umul24((uint32_t)min(max(cc, 0), S.nc - 1), (uint32_t)S.b_sc);
#pragma unroll
            for (int s = 0; s < KS; s++)
                vb[q][s] = (co + kt[s]) << 3;
            bmask |= in ? kbits << (q * KS) : 0u;
        }
        tail = part;
        bmasked = part || !cols_full;
    };
    auto enter = [&, arena, psi, scratch](const GSeg &S) __attribute__((always_inline)) {
        sA = (S.a_src == 0 ? arena : (S.a_src == 1 ? psi : scratch)) + S.a_off;
        sB = (S.b_src == 0 ? arena : (S.b_src == 1 ? psi : scratch)) + S.b_off;
        s_kmaj = (S.a_sk != 1);
        cols_full = (S.tc0 == 0 && S.nc >= item.cols);
        astep = (uint32_t)S.a_sk, bstep = (uint32_t)S.b_sk;
        if (SB)
            salpha = S.alpha;
        lane_offsets(S, 0, S.K < KC);
    };
    // the A half of a fetch: this wave's share of the LDS-DMA of the chunk at k offset kb into LDS buffer `As`
    auto fetch_dma = [&](int kb, double *As) __attribute__((always_inline)) {
        const char *ba = (const char *)(sA + (uint64_t)((uint32_t)kb * astep));
#ifdef B2X_PROBE_L2_WINDOW // (timing probe, wrong results: every operand fetch lands in the first MiB of the arena: all L2 hits)
        ba = (const char *)arena + ((uint64_t)(uintptr_t)ba & 0xFFFF8ull);
#endif
#ifdef B2X_PROBE_NO_LOADS // (timing probe, wrong results)
        if (kb != 0x7fffffff)
            return;
#endif
#pragma unroll
        for (int j = 0; j < NI; j++)
            if (NG % NT == 0 || (wave * NI + j) * 64 < NG) // whole DMA instruction inside the image
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ba + va[j]),
                                                 (__attribute__((address_space(3))) void *)(As + (wave * NI + j) * 128),
                                                 16, 0, 0);
    };
    // ... and the B fragments of the same chunk into bnxt
    auto fetch = [&](int kb, double *As) __attribute__((always_inline)) {
        fetch_dma(kb, As);
        const char *bb = (const char *)(sB + (uint64_t)((uint32_t)kb * bstep));
#ifdef B2X_PROBE_L2_WINDOW
        bb = (const char *)arena + ((uint64_t)(uintptr_t)bb & 0xFFFF8ull);
#endif
        const uint32_t b_s4 = tail ? 0u : bstep * 32u; // bytes per k-step (a partial chunk keeps its k offsets per lane)
#pragma unroll
        for (int s = 0; s < KS; s++) {
#pragma unroll
            for (int q = 0; q < CF; q++)
#ifdef B2X_PROBE_NO_LOADS
                bnxt[q][s] = (double)(vb[q][s] & 1);
#else
                bnxt[q][s] = *(const double *)(bb + vb[q][s]);
#endif
            bb += b_s4;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        if (!bmasked) { // (wave-uniform) the common case: a register copy, no mask arithmetic
#pragma unroll
            for (int q = 0; q < CF; q++)
#pragma unroll
                for (int s = 0; s < KS; s++)
                    bcur[q][s] = SB ? bnxt[q][s] * salpha : bnxt[q][s];
            return;
        }
#pragma unroll
        for (int q = 0; q < CF; q++)
#pragma unroll
            for (int s = 0; s < KS; s++) { // all-ones / zero word from the lane's mask bit, then two ANDs
                const int32_t m = (int32_t)(bmask << (31 - (q * KS + s))) >> 31;
                const double v = SB ? bnxt[q][s] * salpha : bnxt[q][s];
                const uint64_t u = (uint64_t)__double_as_longlong(v) & (uint64_t)(int64_t)m;
                bcur[q][s] = __longlong_as_double((long long)u);
            }
    };
    // the MFMA block: all TMF x CF fragments, branch-free
    // Pin the issue order inside the MFMA block: LDS reads run LEAD fragments ahead of the MFMAs that consume
    // them.  Left alone, hipcc hoists all 4*TMF ds_reads to the top of the block (2 VGPRs each), which at
    // TMF = 16 exceeds the 256-VGPR budget of two waves per SIMD and spills inside the loop.
    auto pin_schedule = [&](auto nks) __attribute__((always_inline)) {
        if constexpr (TMF <= 5 && CF == 2) {
            // Bodies of the register-capped classes (<= 168 / 128 VGPRs): hipcc merges the reads of two neighbouring
            // fragments into one ds_read2st64_b64, so a k-step is (TMF + 1) / 2 LDS instructions, and under the cap it gave
            // ALL of them the same four registers — read, wait, four MFMAs, read, wait, ... with the LDS latency exposed
            // twelve times per chunk.  Pinned instead: two LDS instructions in flight, the next one issued before the
            // MFMAs of the previous pair (eight registers for A).
            constexpr int NDS = decltype(nks)::value * ((TMF + 1) / 2), LEADI = NDS < 2 ? NDS : 2;
            __builtin_amdgcn_sched_group_barrier(0x100, LEADI, 0);
#pragma unroll
            for (int i = 0; i < NDS - LEADI; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * CF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, LEADI * 2 * CF, 0);
        } else {
            constexpr int NRD = decltype(nks)::value * TMF, LEAD = NRD < 6 ? NRD : 6;
            __builtin_amdgcn_sched_group_barrier(0x100, LEAD, 0); // DS reads
#pragma unroll
            for (int i = 0; i < NRD - LEAD; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, CF, 0); // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, LEAD * CF, 0);
        }
    };
    // The MFMA block: all TMF x CF fragments in ONE basic block for both image layouts (a branch over the
    // layouts would give every accumulator two definitions and hipcc then keeps two copies of the tile).
    // Both images are fragment-major with the same fragment stride (16*KC doubles), so the ds_read of fragment f
    // is base(lane, k-step) + an immediate:
    //   rowmaj image: (row, k) at row*KC + 2*((k>>1) ^ swz(row)) + (k&1),  swz = (row>>1)&7 (KC 16) or row&15 (KC 32)
    //   kmaj   image: (row, k) at (row>>4)*16*KC + k*16 + (row&15)         (conflict-free as it stands)
    // with row = f*16 + c, k = 4s + g.
    // k-steps [S0, S0 + NS) of the chunk
    // (the lane's element offset of k-step s inside a fragment for the layout of the CURRENT chunk's segment: kept in KS
    //  registers and rewritten only when the layout changes — both layouts' offsets held side by side and selected per
    //  chunk cost 2 KS registers and 2 KS VALU instructions per chunk, in bodies that have 4 registers for A fragments)
    int aoff[KS];
    auto set_layout = [&](bool kmaj) __attribute__((always_inline)) {
        const int sw = KC == 16 ? ((c >> 1) & 7) : c;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int o_row = c * KC + 2 * ((2 * s + (g >> 1)) ^ sw) + (g & 1);
            const int o_k = (4 * s + g) * 16 + c;
            aoff[s] = kmaj ? o_k : o_row;
        }
    };
    auto compute = [&](const double *As, bool kmaj, auto s0c, auto nsc) __attribute__((always_inline)) {
        constexpr int S0 = decltype(s0c)::value, NS = decltype(nsc)::value;
        (void)kmaj;
        const double *pb[NS];
#pragma unroll
        for (int s = 0; s < NS; s++)
            pb[s] = As + aoff[S0 + s];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int f = 0; f < TMF; f++) {
#ifdef B2X_PROBE_ONE_FRAG // (timing probe, wrong results: `make probes`, tools/probe_ab.sh)
                if (f > 0)
                    continue;
#endif
                double a = pb[s][f * 16 * KC];
#pragma unroll
                for (int q = 0; q < CF; q++)
                    acc[f][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[q][S0 + s], acc[f][q], 0, 0, 0);
            }
        pin_schedule(nsc);
    };

    // A wave whose columns all lie beyond the tile's width (the last column tile of a sector is cut at 32-column wave
    // granularity) only helps staging the A chunks: same segment walk and the same barriers as its workgroup, but no B
    // loads and no MFMAs, so its SIMD is left to the wave of the other workgroup on the CU.  The branch is taken once,
    // before the main loop, so the MFMA path below stays one straight loop.
    if (wave * (CW * 16) >= item.cols) {
        uint32_t hi = item.seg_begin;
        if (hi < item.seg_end) {
            GSeg S = segs[hi];
            int kb = 0, buf = 0;
            enter(S);
            fetch_dma(0, lds);
            __syncthreads();
            while (true) {
                uint32_t nsi = hi;
                int nkb = kb + KC;
                if (nkb >= S.K)
                    nsi = hi + 1, nkb = 0;
                const bool more = nsi < item.seg_end;
                if (more && nsi != hi) {
                    S = segs[nsi];
                    enter(S);
                } else if (more && nkb + KC > S.K)
                    lane_offsets(S, nkb, true); // the segment's partial last chunk
                if (!more)
                    break;
                fetch_dma(nkb, lds + (buf ^ 1) * ABUF);
                buf ^= 1;
                __syncthreads();
                hi = nsi, kb = nkb;
            }
        }
        return;
    }
    uint32_t si = item.seg_begin;
    if (si < item.seg_end) {
        GSeg S = segs[si];
        // The NEXT segment's descriptor (64 bytes = one scalar-cache line) is only touched one segment ahead, through a
        // single dword that stays in one SGPR: holding the whole prefetched descriptor (16 SGPRs) next to the current
        // one left the chunk loop short of scalar registers, and what it spilled it reloaded with VALU lane reads in every
        // chunk, between the MFMAs.  The load at the switch then hits the scalar cache.
        auto touch = [&](uint32_t i) __attribute__((always_inline)) {
            return *(const uint32_t *)(segs + min(i, item.seg_end - 1));
        };
        uint32_t warm = touch(si + 1);
        int kb = 0, buf = 0;
        enter(S);
        fetch(0, lds);
        commit();
        bool cur_kmaj = s_kmaj;
        set_layout(cur_kmaj);
        // A chunk with at most KC/2 valid k (the tail of a segment whose K is not a multiple of the chunk depth; a whole
        // segment at tiny K) runs the first half of its k-steps only: the B fragments of the other half are zero.
        bool cur_short = S.K <= KC / 2;
        __syncthreads(); // drains the DMA (vmcnt(0)) and publishes the image
        // SIMD partners must not run in lockstep (same program, same barrier: both would issue loads, then both MFMAs,
        // leaving the matrix pipe idle in the load phase of both).  With 4-wave workgroups (the shipped configuration)
        // a SIMD's two waves belong to two DIFFERENT workgroups with independent barriers and drift apart by
        // themselves.  With 8-wave workgroups (waves w and w + 4 share a SIMD) the second half of the waves issues the
        // next chunk's loads in the MIDDLE of the MFMA block instead of before it (MI355X_MICROARCH.md, stagger note).
        const bool early = NW < 8 || wave < NW / 2;
        constexpr int KH = KS / 2;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, KH>;
        using NH0 = std::integral_constant<int, KH>;
        using NH1 = std::integral_constant<int, KS - KH>;
        while (true) {
            uint32_t nsi = si;
            int nkb = kb + KC;
            if (nkb >= S.K)
                nsi = si + 1, nkb = 0;
            const bool more = nsi < item.seg_end;
            if (early) {
                if (more && nsi != si) {
                    asm volatile("" ::"s"(warm)); // (the touch of this descriptor has landed by now)
                    S = segs[nsi];
                    warm = touch(nsi + 1);
                    enter(S);
                } else if (more && nkb + KC > S.K)
                    lane_offsets(S, nkb, true); // the segment's partial last chunk
                if (more)
                    fetch(nkb, lds + (buf ^ 1) * ABUF);
            }
            __builtin_amdgcn_sched_barrier(0); // loads issued; nothing below may move above them
            compute(lds + buf * ABUF, cur_kmaj, H0{}, NH0{});
            __builtin_amdgcn_sched_barrier(0);
            if (!early) {
                if (more && nsi != si) {
                    asm volatile("" ::"s"(warm));
                    S = segs[nsi];
                    warm = touch(nsi + 1);
                    enter(S);
                } else if (more && nkb + KC > S.K)
                    lane_offsets(S, nkb, true);
                if (more)
                    fetch(nkb, lds + (buf ^ 1) * ABUF);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!cur_short)
                compute(lds + buf * ABUF, cur_kmaj, H1{}, NH1{});
            __builtin_amdgcn_sched_barrier(0);
            if (!more)
                break;
            buf ^= 1;
            commit();
            if (cur_kmaj != s_kmaj) { // (wave-uniform; only at a switch between segments of different A layouts)
                cur_kmaj = s_kmaj;
                set_layout(cur_kmaj);
            }
            cur_short = S.K - nkb <= KC / 2; // (S is the segment of the chunk at nkb by now)
            __syncthreads();
            si = nsi, kb = nkb;
        }
    }
    // ---- store the tile ----
    // The item descriptor is read AGAIN here (through a pointer the compiler cannot trace) instead of staying in scalar
    // registers across the main loop: the loop is short of SGPRs, and what it spills it reloads with VALU lane reads in
    // every chunk, next to the MFMAs.
    const GItem *ip_opaque = item_ptr;
    asm volatile("" : "+s"(ip_opaque));
    const GItem fin = *ip_opaque;
    double *out = (fin.out_kind ? scratch : slabs) + fin.out_off;
#pragma unroll
    for (int q = 0; q < CF; q++) {
        const int col = wave * (CW * 16) + q * 16 + c;
        if (col < fin.cols) {
            // rows g, g + 4, g + 8, ...: a running pointer (one add per store, no 64-bit multiply); only the tile's last
            // row fragment can be cut short (rows > 16 (TMF - 1) by construction), so only it tests the row
            double *po = out + (int64_t)g * fin.out_ld + col;
            const int64_t step = (int64_t)4 * fin.out_ld;
#pragma unroll
            for (int f = 0; f < TMF; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (f < TMF - 1 || f * 16 + 4 * r + g < fin.rows)
                        *po = fin.alpha * acc[f][q][r];
                    po += step;
                }
        }
    }
}

// One launch per stage: every workgroup picks the body specialised for its item's tile height (one body per number
// of 16-row fragments, 1..kGGTileM/16), so tiles of all heights share a grid (no per-variant launch tails) and one LDS
// allocation, and no MFMA is issued on padding rows beyond the last fragment.
// Shipped configuration: 128 x 128 tiles, 4 waves x (8 row fragments x 2 column fragments) = 128 accumulator VGPRs per
// wave, 32 KB of LDS: TWO workgroups per CU (one wave of each per SIMD).  Against the earlier 256 x 128 / 8-wave /
// CF = 1 workgroup (one per CU) this halves the ds_reads per MFMA (an A fragment feeds two MFMAs), gives the SIMD
// partners independent barriers, and wastes less on the many sectors shorter than 256 rows:
// M=1000 26.2 -> 29.6, M=4000 52.3 -> 55.2 TFLOP/s on the bench plan, 62.3 on uniform 1024^3 pairs.
// TMAX = tallest tile (in 16-row fragments) this instantiation serves.  The register count of a kernel is that of its
// tallest body, so the short tiles (<= 16 kGGShortFrags = 48 rows, or <= 80 rows at 3 waves per SIMD for plans of mid-height
// sectors: CompiledPlan::short_frags; the sectors of small bond dimensions, where a segment
// is one or two chunks long and the exposed load latency, not the MFMA pipe, sets the pace) get an instantiation of
// their own with half the registers (<= 128: no spills up to three row fragments) and 3/8 of the LDS: four waves per
// SIMD instead of two hide that latency.
#ifndef B2X_NARROW_WAVES
#define B2X_NARROW_WAVES 4
#endif
static constexpr int kNarrowWaves = B2X_NARROW_WAVES; // waves per SIMD the 1-wave workgroups of the smallest tiles are compiled for
template <int CF, int NW, int KC, bool SB, int TMAX>
__global__ __launch_bounds__(NW * 64, (NW == 1 && TMAX <= 2) ? kNarrowWaves : (TMAX <= 3 ? 4 : (TMAX <= 5 ? 3 : 2))) void gg_kernel(const GSeg *__restrict__ segs, const GItem *__restrict__ items,
                                                         const double *__restrict__ arena,
                                                         const double *__restrict__ psi, double *__restrict__ scratch,
                                                         double *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) double lds[2 * TMAX * 16 * KC];
    const GItem item = items[blockIdx.x];
    // A wave whose share of the tile is at most 16 columns wide (narrow sectors, the last column tile of a sector) runs the
    // body with ONE active column fragment: half the MFMAs and half the B loads of the padded two (same segment walk and
    // barriers as the other waves of its workgroup, same register budget).
    const int wcols = item.cols - (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (CF * 16);
    const bool half = CF == 2 && wcols > 0 && wcols <= 16;
#define B2X_GG_CASE(T)                                                                                                 \
    case T:                                                                                                            \
        if constexpr (T <= TMAX) {                                                                                     \
            if (half)                                                                                                  \
                gg_body<T, (CF == 2 ? 1 : CF), NW, KC, SB, CF>(item, items + blockIdx.x, lds, segs, arena, psi, scratch, slabs); \
            else                                                                                                       \
                gg_body<T, CF, NW, KC, SB>(item, items + blockIdx.x, lds, segs, arena, psi, scratch, slabs);           \
        }                                                                                                              \
        break;
    switch ((item.rows + 15) >> 4) { // row fragments of the tile
        B2X_GG_CASE(1) B2X_GG_CASE(2) B2X_GG_CASE(3) B2X_GG_CASE(4) B2X_GG_CASE(5) B2X_GG_CASE(6) B2X_GG_CASE(7)
    default:
        if (half)
            gg_body<TMAX, (CF == 2 ? 1 : CF), NW, KC, SB, CF>(item, items + blockIdx.x, lds, segs, arena, psi, scratch, slabs);
        else
            gg_body<TMAX, CF, NW, KC, SB>(item, items + blockIdx.x, lds, segs, arena, psi, scratch, slabs);
    }
#undef B2X_GG_CASE
}

// psi'[tile] += scale * sum_i slab_i[tile]   (fixed order i = 0..n_items-1)
__global__ __launch_bounds__(256) void hpsi_reduce(const DTile *__restrict__ tiles, const double *__restrict__ slabs,
                                                    double *__restrict__ sigma, double scale) {
    const DTile t = tiles[blockIdx.x];
    const int n = t.rows * t.cols;
    if (t.n_items == 0)
        return;
    // blockIdx.y strides over the tile's elements: enough workgroups to stream the slabs at HBM rate
    for (int e = blockIdx.y * 256 + threadIdx.x; e < n; e += gridDim.y * 256) {
        const double *s = slabs + t.slab_off + e;
        double sum = 0.0;
        for (int i = 0; i < t.n_items; i++)
            sum += s[(int64_t)i * n];
        int r = e / t.cols, cc = e - r * t.cols;
        sigma[t.sigma_off + (int64_t)r * t.ld + cc] += scale * sum;
    }
}

// The same sum for tiles with MANY slabs (small psi' and a long contraction: a site near the ends of a chain has one or two tiles
// that all work items of the plan accumulate into, hundreds to thousands of slabs each; the tiles of one plan differ by orders of
// magnitude).  hpsi_reduce walks the slabs of an element one after the other in one thread; here IY threads share an element —
// thread iy takes slabs iy, iy + IY, ... — and their partial sums are added in a fixed binary tree through LDS: still no atomics,
// still the same bits on every run.  IY is chosen PER TILE from its slab count (a power of two, 1 .. 64: about four slabs per
// thread); a workgroup pass covers EX = 256 / IY consecutive elements.
__global__ __launch_bounds__(256) void hpsi_reduce_split(const DTile *__restrict__ tiles, const double *__restrict__ slabs,
                                                          double *__restrict__ sigma, double scale) {
    __shared__ double part[256];
    const DTile t = tiles[blockIdx.x];
    const int n = t.rows * t.cols;
    if (t.n_items == 0)
        return;
    int lg = 0; // IY = 2^lg
    while (lg < 6 && (4 << lg) < t.n_items)
        lg++;
    const int IY = 1 << lg, EX = 256 >> lg;
    const int ex = threadIdx.x & (EX - 1), iy = threadIdx.x >> (8 - lg);
    for (int e0 = blockIdx.y * EX; e0 < n; e0 += gridDim.y * EX) { // (uniform per workgroup: the barriers below are safe)
        const int e = e0 + ex;
        double sum = 0.0;
        if (e < n) {
            const double *s = slabs + t.slab_off + e;
            for (int i = iy; i < t.n_items; i += IY)
                sum += s[(int64_t)i * n];
        }
        part[threadIdx.x] = sum; // [iy][ex]
        __syncthreads();
        for (int h = IY >> 1; h > 0; h >>= 1) {
            if (iy < h)
                part[threadIdx.x] += part[threadIdx.x + h * EX];
            __syncthreads();
        }
        if (iy == 0 && e < n) {
            int r = e / t.cols, cc = e - r * t.cols;
            sigma[t.sigma_off + (int64_t)r * t.ld + cc] += scale * part[ex];
        }
        __syncthreads();
    }
}

// ------------- generic fallback / on-device cross-check (any plan; atomics, not reproducible) -----
// one workgroup per pair: W chunk (16 rows of W at a time) in LDS, scalar FMAs, atomicAdd into psi'.
__global__ __launch_bounds__(256) void hpsi_generic(const b2x_pair *__restrict__ pairs, const double *__restrict__ arena,
                                                     const double *__restrict__ psi, double *__restrict__ sigma,
                                                     double scale) {
    const b2x_pair p = pairs[blockIdx.x];
    constexpr int WC = 2048; // doubles of W kept in LDS per chunk
    __shared__ double Ws[WC];
    const int n = p.n0;
    const double *X = psi + p.x_off, *Y = arena + p.y_off, *Z = arena + p.z_off;
    double *V = sigma + p.v_off;
    // n may exceed WC: then process column chunks too
    for (int cb = 0; cb < n; cb += WC) {
        const int ncol = min(WC, n - cb);
        const int rp = max(1, WC / ncol) < p.m0 ? max(1, WC / ncol) : p.m0;
        for (int rb = 0; rb < p.m0; rb += rp) {
            const int nr = min(rp, p.m0 - rb);
            __syncthreads();
            for (int idx = threadIdx.x; idx < nr * ncol; idx += 256) {
                int r = idx / ncol, cidx = idx - r * ncol + cb;
                double s = 0.0;
                for (int k = 0; k < p.k0; k++) {
                    double y = p.tb0 ? Y[(int64_t)cidx * p.ldb0 + k] : Y[(int64_t)k * p.ldb0 + cidx];
                    s += X[(int64_t)(rb + r) * p.lda0 + k] * y;
                }
                Ws[r * ncol + (cidx - cb)] = s * p.alpha0;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < p.m1 * ncol; idx += 256) {
                int i = idx / ncol, cidx = idx - i * ncol;
                double s = 0.0;
                for (int r = 0; r < nr; r++) {
                    double z = p.ta1 ? Z[(int64_t)(rb + r) * p.lda1 + i] : Z[(int64_t)i * p.lda1 + rb + r];
                    s += z * Ws[r * ncol + cidx];
                }
                atomicAdd(&V[(int64_t)i * p.ldc1 + cb + cidx], s * p.alpha1 * scale);
            }
        }
    }
}

// ------------------------------------ diagonal of H_eff ------------------------------------------
// diag[window] += alpha * a_diag (x) b_diag for every term; one workgroup column per sector (component), every
// thread owns elements and walks the sector's terms in plan order: deterministic, no atomics (HBM/L2-bound, one-off
// per site).
__global__ __launch_bounds__(256) void diag_build_k(const DiagComp *__restrict__ comps, const DiagTermD *__restrict__ terms,
                                                     const double *__restrict__ arena, double *__restrict__ diag) {
    const DiagComp C = comps[blockIdx.x];
    const int n = C.rows * C.cols;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < n; e += gridDim.y * 256) {
        const int r = e / C.cols, c = e - r * C.cols;
        double sum = 0.0;
        for (uint32_t t = C.term_begin; t < C.term_end; t++) {
            const DiagTermD T = terms[t];
            const int rr = r - T.row0, cc = c - T.col0;
            if (rr >= 0 && rr < T.m && cc >= 0 && cc < T.n)
                sum += T.alpha * arena[T.a_off + (int64_t)rr * T.a_stride] * arena[T.b_off + (int64_t)cc * T.b_stride];
        }
        diag[C.base + (int64_t)r * C.ld + c] += sum;
    }
}
hipError_t launch_diag(const DiagComp *comps, uint32_t n_comps, const DiagTermD *terms, const double *arena, double *diag,
                       hipStream_t st) {
    if (n_comps == 0)
        return hipSuccess;
    hipLaunchKernelGGL(diag_build_k, dim3(n_comps, 32), dim3(256), 0, st, comps, terms, arena, diag);
    return hipGetLastError();
}

// out[cell] += sum over the cell's entries of alpha * A[r][c] * B[r][c]: element-wise block products of the blocking step
// (a (x) scalar site operator, operator sums).  ONE WAVE per work unit (a few 64-column x rpt-row tiles of one cell; the
// many small symmetry blocks give tens of thousands of units, so four independent waves share a workgroup only for
// dispatch).  Lane = column, the wave walks the tile's rows four at a time (4 x entries loads in flight) and the
// entry list in plan order with no window test: deterministic, no atomics.  HBM-bound: non-transposed blocks are read
// as 512-byte row segments; a transposed block is read with one cache line per lane that the next rows reuse from L1/L2.
template <int RU, bool HASB> // HASB = false: every B operand is the constant 1 (operator sums); RU = rows in flight per entry visit: 4 (blocking: short tiles), 16 (operator sums: a transposed block is read one
                   // cache line per lane, and 16 rows consume every element of the line before the next entry evicts it)
__global__ __launch_bounds__(256) void outer_build_k(const OWork *__restrict__ work, uint32_t n_work,
                                                      const OEntry *__restrict__ entries, const double *__restrict__ arena,
                                                      const double *__restrict__ in, double *__restrict__ out) {
    const uint32_t unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (unit >= n_work)
        return;
    const int lane = threadIdx.x & 63;
    const OWork W = work[unit];
    const uint32_t nseg = (uint32_t)(W.cols + 63) >> 6;
    for (uint32_t tile = W.t_begin; tile < W.t_end; tile++) {
        const int r0 = (int)(tile / nseg) * W.rpt, c = (int)(tile % nseg) * 64 + lane;
        const int r1 = min(W.rows, r0 + W.rpt);
        const bool live = c < W.cols;
        const uint64_t cc = (uint64_t)min(c, W.cols - 1); // clamped: dead lanes load a valid element and drop it
        for (int r = r0; r < r1; r += RU) {
            double sum[RU];
            uint64_t rr[RU];
#pragma unroll
            for (int u = 0; u < RU; u++)
                sum[u] = 0.0, rr[u] = (uint64_t)min(r + u, r1 - 1);
            const bool full_rows = r + RU <= r1; // no clamped (repeated) row among the RU in flight
            // software pipeline over the entry list: the loads of entry k + 1 are in flight while entry k is consumed
            // (descriptor by scalar loads one entry ahead, operands in a second register set)
            double a_n[RU], b_n[RU];
            OEntry Tn = entries[W.entry_begin];
            auto issue = [&](const OEntry &T) __attribute__((always_inline)) {
                const double *pa = (T.a_src == 1 ? in : arena) + T.a_off + cc * (uint64_t)T.a_cs;
                if (RU >= 4 && T.a_rs == 1 && full_rows) { // transposed block: the lane's rows are contiguous, 32-byte loads
#pragma unroll
                    for (int u = 0; u < RU; u += 4) {
                        const d4u v = *(const d4u *)(pa + rr[u]);
                        a_n[u] = v[0], a_n[u + 1] = v[1], a_n[u + 2] = v[2], a_n[u + 3] = v[3];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < RU; u++)
                        a_n[u] = pa[rr[u] * (uint64_t)T.a_rs];
                }
                if (HASB) {
                    if (T.b_src != 2) { // (wave-uniform: the descriptor lives in SGPRs)
                        const double *pb = (T.b_src == 1 ? in : arena) + T.b_off + cc * (uint64_t)T.b_cs;
#pragma unroll
                        for (int u = 0; u < RU; u++)
                            b_n[u] = pb[rr[u] * (uint64_t)T.b_rs];
                    } else {
#pragma unroll
                        for (int u = 0; u < RU; u++)
                            b_n[u] = 1.0;
                    }
                }
            };
            issue(Tn);
            for (uint32_t k = W.entry_begin; k < W.entry_end; k++) {
                const OEntry T = Tn;
                double a[RU], b[RU];
#pragma unroll
                for (int u = 0; u < RU; u++)
                    a[u] = a_n[u], b[u] = HASB ? b_n[u] : 1.0;
                if (k + 1 < W.entry_end) {
                    Tn = entries[k + 1];
                    issue(Tn);
                }
#pragma unroll
                for (int u = 0; u < RU; u++)
                    sum[u] += T.alpha * (T.a_src == 2 ? 1.0 : a[u]) * b[u];
            }
            if (live) {
                const bool assign = W.ld < 0; // sum pass of the two-stage path: S = ..., not S += ...
                const uint64_t ld = (uint64_t)(assign ? -W.ld : W.ld);
#pragma unroll
                for (int u = 0; u < RU; u++)
                    if (r + u < r1) {
                        double *o = out + W.out_off + (uint64_t)(r + u) * ld + c;
                        *o = assign ? sum[u] : *o + sum[u];
                    }
            }
        }
    }
}
hipError_t launch_outer(const OWork *work, uint32_t n_work, const OEntry *entries, const double *arena, const double *in,
                        double *out, int rows_in_flight, hipStream_t st) {
    if (n_work == 0)
        return hipSuccess;
    // the sum passes: sixteen rows in flight (the HASB = false specialisation needs fewer registers, runs more waves per
    // SIMD and is SLOWER: 127 vs 113 ms on the M=4000 noise list — the transposed blocks live on L1/L2 reuse)
    note_slots((const void *)outer_build_k<16, true>, -1, -1, -1), note_slots((const void *)outer_build_k<4, true>, -1, -1, -1);
    if (rows_in_flight >= 16)
        hipLaunchKernelGGL((outer_build_k<16, true>), dim3((n_work + 3) / 4), dim3(256), 0, st, work, n_work, entries, arena, in, out);
    else
        hipLaunchKernelGGL((outer_build_k<4, true>), dim3((n_work + 3) / 4), dim3(256), 0, st, work, n_work, entries, arena, in, out);
    return hipGetLastError();
}

// ------------------------------------ vector kernels ------------------------------------------
__global__ void vec_axpy_k(double a, const double *x, double *y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
__global__ void vec_scal_k(double a, double *x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] *= a;
}
__global__ void vec_precond_k(double *q, const double *diag, double shift, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double d = shift - diag[i]; // davidson_precondition: q /= ld - aa[i]
        if (fabs(d) > 1e-12)
            q[i] /= d;
    }
}
// (qo == q: in place, the reference's form; qo != q leaves the residual itself intact for the fused dot products of the
// Davidson step)
__global__ void vec_olsen_k(const double *q, double *qo, double *t, const double *c, const double *diag, double ld, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double d = ld - diag[i], tv = c[i], qv = q[i];
        if (fabs(d) > 1e-12)
            tv /= d, qv /= d;
        t[i] = tv, qo[i] = qv;
    }
}
struct VecPtrs {
    const double *p[64];
    double coef[64];
};
// y = sum_j coef[j] * vs[j]
__global__ void vec_lincomb_k(VecPtrs vp, int nv, double *y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int j = 0; j < nv; j++)
            s += vp.coef[j] * vp.p[j][i];
        y[i] = s;
    }
}
// partial[j * nblk + b] = sum over block b's grid-stride share of vs[j][i] * x[i]; fixed-order two-pass
// (grid = blocks x vectors: the vectors of a Gram column are reduced side by side, not one after the other in a block)
__global__ __launch_bounds__(256) void vec_multidot_k(VecPtrs vp, int nv, const double *x, size_t n, double *partial) {
    __shared__ double sh[256];
    const int j = blockIdx.y;
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        s += vp.p[j][i] * x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        partial[(size_t)j * gridDim.x + blockIdx.x] = sh[0];
}
// Ritz vector, residual and the first half of the Olsen preconditioner of a Davidson step in ONE pass over the basis and its
// images: x = sum_j a_j b_j, q = sum_j a_j s_j - theta x, and with d = theta - diag (where |d| > 1e-12, else 1): q2 = q / d,
// t = x / d.  (Two linear combinations and vec_olsen_k before: three launches, x and q read back from memory.)
struct RitzPtrs {
    const double *b[64];
    const double *s[64];
    double a[64];
};
__global__ void vec_ritz_olsen_k(RitzPtrs rp, int m, double theta, const double *__restrict__ diag, double *__restrict__ x,
                                 double *__restrict__ q, double *__restrict__ q2, double *__restrict__ t, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double xv = 0.0, sv = 0.0;
        for (int j = 0; j < m; j++)
            xv += rp.a[j] * rp.b[j][i], sv += rp.a[j] * rp.s[j][i];
        const double qv = sv - theta * xv;
        const double d = theta - diag[i];
        const bool ok = fabs(d) > 1e-12;
        x[i] = xv, q[i] = qv;
        q2[i] = ok ? qv / d : qv, t[i] = ok ? xv / d : xv;
    }
}
// Second Gram-Schmidt pass + normalisation with the coefficients still ON THE DEVICE (dots[j] = <b_j, v>, dots[m] = <v, v>, as
// vec_pairdot_k + vec_multidot_final_k left them): out = (v - sum_j dots[j] b_j) / sqrt(<v, v> - sum_j dots[j]^2).  The norm of
// the result follows from the dots for an orthonormal basis; when it is not safely positive (the new direction lies in the span of
// the basis to rounding) the unnormalised difference is written and *flag set: the host looks at the flag after its next wait.
__global__ void vec_gs_finish_k(VecPtrs vp, int m, const double *__restrict__ v, const double *__restrict__ dots,
                                double *__restrict__ out, size_t n, int *flag) {
    double nrm2 = dots[m];
    for (int j = 0; j < m; j++)
        nrm2 -= dots[j] * dots[j];
    const bool ok = nrm2 > 1e-24 * fabs(dots[m]) && nrm2 > 0.0;
    const double inv = ok ? 1.0 / sqrt(nrm2) : 1.0;
    if (!ok && blockIdx.x == 0 && threadIdx.x == 0)
        *flag = 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double s = v[i];
        for (int j = 0; j < m; j++)
            s -= dots[j] * vp.p[j][i];
        out[i] = s * inv;
    }
}
// dst[d.dst + i] = src[d.src + i], i < d.len, for every descriptor: the operator blocks of an enlarged block gathered into
// the arena of the next step in ONE launch (a site has ~1e3 of them; one hipMemcpyAsync each was 3 us of host time apiece).
// One workgroup per descriptor; the host cuts long ranges into pieces of kCopyPiece elements.
struct CopyDesc {
    uint64_t dst, src, len;
};
__global__ __launch_bounds__(256) void vec_gather_k(const CopyDesc *__restrict__ ds, double *__restrict__ dst,
                                                    const double *__restrict__ src) {
    const CopyDesc d = ds[blockIdx.x];
    double *o = dst + d.dst;
    const double *s = src + d.src;
    for (uint64_t i = threadIdx.x; i < d.len; i += 256)
        o[i] = s[i];
}
// the same for independent pairs: partial[j * nblk + b] = share of <u_j, v_j> (one launch and one host round trip for all
// the dot products of a Davidson step; pointers only: 2 KB of kernel arguments for 128 pairs)
struct PairPtrs {
    const double *u[128];
    const double *v[128];
};
__global__ __launch_bounds__(256) void vec_pairdot_k(PairPtrs pp, size_t n, double *partial) {
    __shared__ double sh[256];
    const int j = blockIdx.y;
    const double *u = pp.u[j], *v = pp.v[j];
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        s += u[i] * v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        partial[(size_t)j * gridDim.x + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void vec_multidot_final_k(const double *partial, int nblk, double *out) {
    __shared__ double sh[256];
    int j = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256)
        s += partial[(size_t)j * nblk + b];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[j] = sh[0];
}

// ------------------------------------ launchers -----------------------------------------------
template <int TMF, int K1F, int CF>
static hipError_t launch_wave_t(const DPart *parts, const DItem *items, uint32_t n_items, const double *arena,
                                const double *psi, double *slabs, hipStream_t st) {
    note_slots((const void *)hpsi_wave<TMF, K1F, CF>, 4, -1, -1);
    hipLaunchKernelGGL((hpsi_wave<TMF, K1F, CF>), dim3((n_items + 3) / 4), dim3(256), 0, st, parts, items, n_items,
                       arena, psi, slabs);
    return hipGetLastError();
}

hipError_t launch_main(int cls, const DPart *parts, const DItem *items, uint32_t n_items, const double *arena,
                       const double *psi, double *slabs, hipStream_t st) {
    if (n_items == 0)
        return hipSuccess;
    switch (cls) { // (tmf, k1f, tile width / 16) of kClasses[cls]
    case 0:
        return launch_wave_t<2, 2, 2>(parts, items, n_items, arena, psi, slabs, st);
    case 1:
        return launch_wave_t<4, 2, 2>(parts, items, n_items, arena, psi, slabs, st);
    case 2:
        return launch_wave_t<8, 2, 2>(parts, items, n_items, arena, psi, slabs, st);
    }
    return hipErrorInvalidValue;
}

// items of one stage: [v_begin[0], v_begin[1]) tall tiles (> 16 kGGShortFrags rows), [v_begin[1], v_begin[kGGVariants]) short ones;
// each class is one launch of the instantiation that serves it
// which: 0 = both classes on `st`, 1 = the tall class only, 2 = the short class only
hipError_t launch_gg(const GSeg *segs, const GItem *items, const uint32_t *v_begin, const double *arena,
                     const double *psi, double *scratch, double *slabs, bool seg_scaled, int tile_n, hipStream_t st, int which,
                     bool short_narrow, int short_frags) {
    // chunk depth 16: a 32-deep chunk halves the barriers (+2 % on uniform 1024^3 pairs) but pads every K to 32 and
    // spills at 256 VGPRs (-2 % on the M=4000 plan)
#define B2X_GG_LAUNCH(NWV, SBV, TMAXV, B, E)                                                                           \
    if ((E) > (B)) {                                                                                                   \
        note_slots((const void *)gg_kernel<kGGCF, NWV, 16, SBV, TMAXV>, 3, -1, -1);                                    \
        hipLaunchKernelGGL((gg_kernel<kGGCF, NWV, 16, SBV, TMAXV>), dim3((E) - (B)), dim3(NWV * 64), 0, st, segs,      \
                           items + (B), arena, psi, scratch, slabs);                                                   \
    }
    const int nw = tile_n / (16 * kGGCF); // waves per workgroup: 4 (128-column tiles) or 2 (narrow sectors)
    const uint32_t b0 = which == 2 ? v_begin[1] : v_begin[0], b1 = v_begin[1];
    uint32_t b2 = which == 1 ? v_begin[1] : v_begin[kGGVariants];
    if (short_narrow && b2 > b1) { // the short class as 1-wave workgroups of 32 columns (H.psi plans: the SCALED variant)
        if (!seg_scaled)
            return hipErrorInvalidValue;
        B2X_GG_LAUNCH(1, true, kGGNarrowFrags, b1, b2);
        b2 = b1;
    }
    if (seg_scaled && short_frags == kGGMidFrags && b2 > b1) { // short class of up to 5 row fragments (3 waves / SIMD)
        if (nw >= 4) {
            B2X_GG_LAUNCH(4, true, kGGMidFrags, b1, b2);
        } else {
            B2X_GG_LAUNCH(2, true, kGGMidFrags, b1, b2);
        }
        b2 = b1;
    }
    if (nw >= 4) {
        if (seg_scaled) {
            B2X_GG_LAUNCH(4, true, 8, b0, b1);
            B2X_GG_LAUNCH(4, true, kGGShortFrags, b1, b2);
        } else {
            B2X_GG_LAUNCH(4, false, 8, b0, b1);
            B2X_GG_LAUNCH(4, false, kGGShortFrags, b1, b2);
        }
    } else { // (the plan compiler picks 128- or 64-column tiles)
        if (seg_scaled) {
            B2X_GG_LAUNCH(2, true, 8, b0, b1);
            B2X_GG_LAUNCH(2, true, kGGShortFrags, b1, b2);
        } else {
            B2X_GG_LAUNCH(2, false, 8, b0, b1);
            B2X_GG_LAUNCH(2, false, kGGShortFrags, b1, b2);
        }
    }
#undef B2X_GG_LAUNCH
    return hipGetLastError();
}

// max_elems = rows x columns of the largest tile of the list: small tiles get fewer workgroups each (a 32 x 32 tile is four
// blocks of 256 elements; sixteen per tile, twelve of them idle, made the reduce 9 % of an H.psi at M=250)
hipError_t launch_reduce(const DTile *tiles, uint32_t n_tiles, const double *slabs, double *sigma, double scale,
                         hipStream_t st, uint32_t max_elems, uint32_t max_items) {
    if (n_tiles == 0)
        return hipSuccess;
    // many slabs per tile: the slabs of an element are split over up to 64 threads (hpsi_reduce_split; the grid covers the
    // narrowest element block, EX = 4, of the largest tile, capped: tiles with fewer slabs use wider blocks and fewer passes)
    // (large tiles with a few hundred slabs — the plans of M >= 2000 — stream better through the one-thread-per-element kernel:
    // 0.75 against 1.1 ms per launch at M=4000)
    if (max_items >= 16 && (max_elems <= 4096 || max_items >= 2048)) {
        const uint32_t elems = max_elems == 0 ? 16384u : max_elems;
        uint32_t ex = 64;
        while (ex > 4 && 4u * (256u / ex) < max_items)
            ex >>= 1;
        const uint32_t gy = std::min(64u, std::max(1u, (elems + ex - 1) / ex));
        note_slots((const void *)hpsi_reduce_split, -1, 2, 3);
        hipLaunchKernelGGL(hpsi_reduce_split, dim3(n_tiles, gy), dim3(256), 0, st, tiles, slabs, sigma, scale);
        return hipGetLastError();
    }
    const uint32_t gy = max_elems == 0 ? 16u : std::min(16u, std::max(1u, (max_elems + 255u) / 256u));
    note_slots((const void *)hpsi_reduce, -1, 2, 3);
    hipLaunchKernelGGL(hpsi_reduce, dim3(n_tiles, gy), dim3(256), 0, st, tiles, slabs, sigma, scale);
    return hipGetLastError();
}

hipError_t launch_generic(const b2x_pair *pairs, uint32_t n_pairs, const double *arena, const double *psi,
                          double *sigma, double scale, hipStream_t st) {
    if (n_pairs == 0)
        return hipSuccess;
    hipLaunchKernelGGL(hpsi_generic, dim3(n_pairs), dim3(256), 0, st, pairs, arena, psi, sigma, scale);
    return hipGetLastError();
}

static inline int vec_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
hipError_t launch_axpy(double a, const double *x, double *y, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(vec_axpy_k, dim3(vec_grid(n)), dim3(256), 0, st, a, x, y, n);
    return hipGetLastError();
}
hipError_t launch_scal(double a, double *x, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(vec_scal_k, dim3(vec_grid(n)), dim3(256), 0, st, a, x, n);
    return hipGetLastError();
}
hipError_t launch_precond(double *q, const double *diag, double shift, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(vec_precond_k, dim3(vec_grid(n)), dim3(256), 0, st, q, diag, shift, n);
    return hipGetLastError();
}
hipError_t launch_olsen(const double *q, double *q_out, double *t, const double *c, const double *diag, double ld, size_t n,
                        hipStream_t st) {
    hipLaunchKernelGGL(vec_olsen_k, dim3(vec_grid(n)), dim3(256), 0, st, q, q_out, t, c, diag, ld, n);
    return hipGetLastError();
}
hipError_t launch_lincomb(const double *const *vs, const double *coef, int nv, double *y, size_t n, hipStream_t st) {
    VecPtrs vp;
    for (int j = 0; j < nv; j++)
        vp.p[j] = vs[j], vp.coef[j] = coef[j];
    hipLaunchKernelGGL(vec_lincomb_k, dim3(vec_grid(n)), dim3(256), 0, st, vp, nv, y, n);
    return hipGetLastError();
}
int multidot_blocks(size_t n) {
    size_t b = (n + 256 * 8 - 1) / (256 * 8);
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
hipError_t launch_multidot(const double *const *vs, int nv, const double *x, size_t n, double *partial, double *out,
                           hipStream_t st) {
    VecPtrs vp;
    for (int j = 0; j < nv; j++)
        vp.p[j] = vs[j], vp.coef[j] = 0.0;
    int nb = multidot_blocks(n);
    hipLaunchKernelGGL(vec_multidot_k, dim3(nb, nv), dim3(256), 0, st, vp, nv, x, n, partial);
    hipLaunchKernelGGL(vec_multidot_final_k, dim3(nv), dim3(256), 0, st, partial, nb, out);
    return hipGetLastError();
}

hipError_t launch_ritz_olsen(const double *const *bs, const double *const *ss, int m, const double *alpha, double theta,
                             const double *diag, double *x, double *q, double *q2, double *t, size_t n, hipStream_t st) {
    RitzPtrs rp;
    for (int j = 0; j < m; j++)
        rp.b[j] = bs[j], rp.s[j] = ss[j], rp.a[j] = alpha[j];
    hipLaunchKernelGGL(vec_ritz_olsen_k, dim3(vec_grid(n)), dim3(256), 0, st, rp, m, theta, diag, x, q, q2, t, n);
    return hipGetLastError();
}
hipError_t launch_gs_finish(const double *const *bs, int m, const double *v, double *partial, double *dots, double *out, size_t n,
                            int *flag, hipStream_t st) {
    PairPtrs pp;
    VecPtrs vp;
    for (int j = 0; j < m; j++)
        pp.u[j] = bs[j], pp.v[j] = v, vp.p[j] = bs[j], vp.coef[j] = 0.0;
    pp.u[m] = v, pp.v[m] = v;
    int nb = multidot_blocks(n);
    hipLaunchKernelGGL(vec_pairdot_k, dim3(nb, m + 1), dim3(256), 0, st, pp, n, partial);
    hipLaunchKernelGGL(vec_multidot_final_k, dim3(m + 1), dim3(256), 0, st, partial, nb, dots);
    hipLaunchKernelGGL(vec_gs_finish_k, dim3(vec_grid(n)), dim3(256), 0, st, vp, m, v, dots, out, n, flag);
    return hipGetLastError();
}
hipError_t launch_gather(const void *descs, uint32_t n, double *dst, const double *src, hipStream_t st) {
    hipLaunchKernelGGL(vec_gather_k, dim3(n), dim3(256), 0, st, (const CopyDesc *)descs, dst, src);
    return hipGetLastError();
}
hipError_t launch_pairdot(const double *const *us, const double *const *vs, int np, size_t n, double *partial, double *out,
                          hipStream_t st) {
    PairPtrs pp;
    for (int j = 0; j < np; j++)
        pp.u[j] = us[j], pp.v[j] = vs[j];
    int nb = multidot_blocks(n);
    hipLaunchKernelGGL(vec_pairdot_k, dim3(nb, np), dim3(256), 0, st, pp, n, partial);
    hipLaunchKernelGGL(vec_multidot_final_k, dim3(np), dim3(256), 0, st, partial, nb, out);
    return hipGetLastError();
}

} // namespace b2x
