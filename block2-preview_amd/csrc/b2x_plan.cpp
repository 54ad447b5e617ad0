// b2x_plan.cpp — plan compiler (host).  See b2x_plan.hpp for the scheme.
#include "b2x_plan.hpp"
#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <string>
#include <cmath>
#include <map>
#include <memory_resource>
#include <sys/mman.h>
#include <cstring>
#include <deque>
#include <cstdlib>
#include <numeric>
#include <unordered_map>

namespace b2x {

namespace {

struct Window {
    uint64_t off;
    int32_t m, n, ld;
    uint32_t pair;
};

struct Component {
    uint64_t base; // psi' offset of local (0, 0)
    int32_t ld, rows, cols;
    uint32_t w_begin, w_end; // windows (sorted array) of this component
};

// Memory of the plan compiler's temporaries.  A plan of 10^5 pairs allocates (and frees) some hundred MB of short-lived
// lists per compilation; through malloc every large one is a fresh mmap whose pages are faulted in one by one — a
// quarter of the compile time (Cr2 M=250 plan 163 -> 118 ms, x16 plan 441 -> 318 ms with a retained heap).  The
// temporaries therefore come from a per-thread bump arena that is rewound, not released, when a compilation ends
// (2 MB-aligned anonymous mappings, MADV_HUGEPAGE where the kernel offers it; given back above kKeepBytes).
class ScratchArena : public std::pmr::memory_resource {
    struct Chunk {
        char *p;
        size_t cap;
    };
    std::vector<Chunk> chunks;
    size_t cur = 0, off = 0;
    static constexpr size_t kChunk = (size_t)64 << 20, kKeepBytes = (size_t)1 << 30;
    void *do_allocate(size_t bytes, size_t align) override {
        for (;; cur++, off = 0) {
            if (cur == chunks.size()) {
                const size_t cap = (std::max(bytes + align, kChunk) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
                void *m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (m == MAP_FAILED)
                    throw std::bad_alloc();
                (void)madvise(m, cap, MADV_HUGEPAGE);
                chunks.push_back(Chunk{(char *)m, cap});
            }
            const size_t o = (off + align - 1) & ~(align - 1);
            if (o + bytes <= chunks[cur].cap) {
                off = o + bytes;
                return chunks[cur].p + o;
            }
        }
    }
    void do_deallocate(void *, size_t, size_t) override {}
    bool do_is_equal(const std::pmr::memory_resource &o) const noexcept override { return this == &o; }

  public:
    size_t capacity() const {
        size_t t = 0;
        for (const Chunk &c : chunks)
            t += c.cap;
        return t;
    }
    void rewind() {
        if (capacity() > kKeepBytes) {
            for (const Chunk &c : chunks)
                munmap(c.p, c.cap);
            chunks.clear();
        }
        cur = 0, off = 0;
    }
    ~ScratchArena() override {
        for (const Chunk &c : chunks)
            munmap(c.p, c.cap);
    }
};
struct ArenaScope { // rewinds the arena when the compilation ends, however it ends
    ScratchArena &a;
    ~ArenaScope() { a.rewind(); }
};
template <class T> using tvec = std::pmr::vector<T>;
// the reserved (still untouched) storage of a large output list: ask for 2 MB pages, 512 x fewer first-touch faults
inline void advise_huge(const void *p, size_t bytes) {
    const uintptr_t two_mb = (uintptr_t)2 << 20, b = ((uintptr_t)p + two_mb - 1) & ~(two_mb - 1), e = ((uintptr_t)p + bytes) & ~(two_mb - 1);
    if (e > b)
        (void)madvise((void *)b, (size_t)(e - b), MADV_HUGEPAGE);
}

// v reordered as std::stable_sort by the NK key words T::k would order it, for lists that fall into FEW groups of equal
// keys (the 155 728 records of the Cr2 noise list: 1 400 groups): one hashing pass assigns groups, only the group
// representatives are sorted, the members follow in their original order.
template <int NK, class T> void group_sort(std::vector<T> &v, std::pmr::memory_resource *mem) {
    const size_t n = v.size();
    if (n < 2)
        return;
    struct Slot {
        uint64_t h;
        uint32_t rep, id; // rep = first member (+1; 0 = empty)
    };
    size_t cap = 64;
    while (cap < 2 * n)
        cap <<= 1;
    tvec<Slot> tab(cap, Slot{0, 0, 0}, mem);
    tvec<uint32_t> gid(n, mem), reps(mem), count(mem);
    for (size_t x = 0; x < n; x++) {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < NK; k++)
            h = (h ^ v[x].k[k]) * 0xBF58476D1CE4E5B9ull, h ^= h >> 29;
        size_t i = (size_t)h & (cap - 1);
        while (tab[i].rep != 0 && !(tab[i].h == h && std::equal(v[x].k, v[x].k + NK, v[tab[i].rep - 1].k)))
            i = (i + 1) & (cap - 1);
        if (tab[i].rep == 0) {
            tab[i] = Slot{h, (uint32_t)x + 1, (uint32_t)reps.size()};
            reps.push_back((uint32_t)x), count.push_back(0);
        }
        gid[x] = tab[i].id, count[tab[i].id]++;
    }
    tvec<uint32_t> order(reps.size(), mem); // groups by key
    for (size_t g = 0; g < order.size(); g++)
        order[g] = (uint32_t)g;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return std::lexicographical_compare(v[reps[a]].k, v[reps[a]].k + NK, v[reps[b]].k, v[reps[b]].k + NK);
    });
    tvec<size_t> start(reps.size(), mem);
    size_t pos = 0;
    for (uint32_t g : order)
        start[g] = pos, pos += count[g];
    std::vector<T> out(n);
    for (size_t x = 0; x < n; x++)
        out[start[gid[x]]++] = v[x];
    v.swap(out);
}

// (development aid, B2X_PLAN_DEBUG=1) wall time of the phases of compile_plan, on stderr
struct PhaseClock {
    bool on = getenv("B2X_PLAN_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::map<std::string, double> acc;
    void lap(const char *name) {
        if (!on)
            return;
        const auto n = std::chrono::steady_clock::now();
        acc[name] += std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
    }
    void report() {
        if (!on)
            return;
        for (auto &kv : acc)
            fprintf(stderr, "[b2x plan] phase %-22s %8.2f ms\n", kv.first.c_str(), kv.second);
    }
};
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
// columns on which a tile of `cols` columns issues MFMAs: a wave covers 16 kGGCF columns; the wave that holds the last
// 1..16 columns runs the body with one active column fragment (gg_kernel), every other one both
// k depth on which a segment of depth K issues MFMAs: whole 16-deep chunks, and a tail of 1..8 runs half a chunk
inline int issued_k(int K) {
    const int rem = K % 16;
    return K - rem + (rem == 0 ? 0 : (rem <= 8 ? 8 : 16));
}
inline int issued_cols(int cols) {
    const int w = 16 * kGGCF, rem = cols % w;
    return cols - rem + (rem == 0 ? 0 : (rem <= 16 ? 16 : w));
}

// cut `total` into n nearly equal pieces that are multiples of 16 (except the last)
std::vector<int> balanced_cuts(int total, int max_piece) {
    int n = ceil_div(total, max_piece);
    int piece = round_up(ceil_div(total, n), 16);
    if (piece > max_piece)
        piece = max_piece;
    std::vector<int> cuts;
    for (int s = 0; s < total; s += piece)
        cuts.push_back(s);
    cuts.push_back(total);
    return cuts;
}

// Column cuts for the grouped-GEMM kernel: full tiles plus one remainder tile.  A wave covers kGGTileN / NW columns and a
// wave without columns issues no MFMA, so the remainder tile costs only the waves it fills (balanced tiles would keep
// more waves busy on padding).
std::vector<int> wave_cuts(int total, int tile) {
    std::vector<int> cuts;
    for (int s = 0; s < total; s += tile)
        cuts.push_back(s);
    cuts.push_back(total);
    return cuts;
}

// Row cuts for the grouped-GEMM kernel, which has a body for every tile height that is a multiple of 16 rows
// (<= 256): split `total` into ceil(U/16) pieces of near-equal size in units of 16 (U = ceil(total/16)), so the
// issued rows are 16*U (the minimum) and no piece is needlessly small.
std::vector<int> unit_cuts(int total, int max_units = kGGTileM / kGGRowUnit) {
    const int U = ceil_div(total, kGGRowUnit), n = ceil_div(U, max_units);
    std::vector<int> cuts{0};
    int pos = 0;
    for (int i = 0; i < n; i++) {
        int units = U / n + (i < U % n ? 1 : 0);
        pos = std::min(total, pos + units * kGGRowUnit);
        cuts.push_back(pos);
    }
    cuts.back() = total;
    return cuts;
}

// The cut lists are asked for once or twice per PAIR and the dimensions repeat heavily (a few hundred distinct values per
// plan): memoised, so that a plan of 10^5 pairs does not pay 3 x 10^5 small vector allocations.
// (a table per parameter value — three or four of them occur — indexed by `total`: a hash lookup per call was a tenth of the
// stage-0 list build)
template <class F> const std::vector<int> &cuts_memo(int total, int param, F make) {
    struct Tab {
        int param;
        std::deque<std::vector<int>> by_total; // (a deque: growing it leaves the references handed out earlier valid)
    };
    static thread_local std::vector<Tab> tabs;
    Tab *t = nullptr;
    for (Tab &x : tabs)
        if (x.param == param)
            t = &x;
    if (!t) {
        tabs.push_back(Tab{param, {}});
        t = &tabs.back();
    }
    if (total < 0 || total >= (1 << 20)) { // (no such dimension in practice: computed, not kept)
        static thread_local std::vector<int> big;
        big = make(total, param);
        return big;
    }
    if ((size_t)total >= t->by_total.size())
        t->by_total.resize((size_t)total + 1);
    std::vector<int> &v = t->by_total[total];
    if (v.empty())
        v = make(total, param);
    return v;
}
const std::vector<int> &unit_cuts_m(int total, int max_units = kGGTileM / kGGRowUnit) {
    return cuts_memo(total, max_units, [](int t, int p) { return unit_cuts(t, p); });
}
const std::vector<int> &wave_cuts_m(int total, int tile) {
    return cuts_memo(total, tile, [](int t, int p) { return wave_cuts(t, p); });
}

// Sort the output windows and merge overlapping ones into disjoint components (sectors of the output vector with a
// common leading dimension).  `fallback` is set when the windows cannot be laid on a common grid.
std::vector<Component> build_components(std::vector<Window> &win, bool &fallback, std::string &reason) {
    // stable order by offset (windows of equal offset stay in plan order).  Long lists (the 1e5..1e6 windows of a blocking or
    // H.psi list) go through a least-significant-digit radix pass over the offsets — the same order as std::stable_sort gives,
    // a third of its time
    if (win.size() < 4096) {
        std::stable_sort(win.begin(), win.end(), [](const Window &a, const Window &b) { return a.off < b.off; }); // (mostly pre-sorted input: the merge sort wins)
    } else {
        uint64_t top = 0;
        for (const Window &w : win)
            top |= w.off;
        std::vector<Window> tmp(win.size());
        std::vector<uint32_t> cnt;
        const int kBits = 11;
        for (int shift = 0; shift < 64 && (top >> shift) != 0; shift += kBits) {
            cnt.assign(((size_t)1 << kBits) + 1, 0);
            for (const Window &w : win)
                cnt[((w.off >> shift) & (((uint64_t)1 << kBits) - 1)) + 1]++;
            for (size_t d = 0; d < ((size_t)1 << kBits); d++)
                cnt[d + 1] += cnt[d];
            for (const Window &w : win)
                tmp[cnt[(w.off >> shift) & (((uint64_t)1 << kBits) - 1)]++] = w;
            win.swap(tmp);
        }
    }
    std::vector<Component> comps;
    size_t i = 0;
    while (i < win.size()) {
        uint64_t end = win[i].off + (uint64_t)(win[i].m - 1) * win[i].ld + win[i].n;
        size_t j = i + 1;
        while (j < win.size() && win[j].off < end) {
            end = std::max(end, win[j].off + (uint64_t)(win[j].m - 1) * win[j].ld + win[j].n);
            j++;
        }
        Component c;
        c.base = win[i].off, c.w_begin = (uint32_t)i, c.w_end = (uint32_t)j;
        // a single-row window may carry any ld; take ld from a multi-row window if there is one
        int ld = 0;
        for (size_t k = i; k < j; k++)
            if (win[k].m > 1) {
                if (ld == 0)
                    ld = win[k].ld;
                else if (ld != win[k].ld) {
                    fallback = true;
                    reason = "overlapping output windows with different leading dimensions";
                }
            }
        if (ld == 0) {
            ld = 1;
            for (size_t k = i; k < j; k++)
                ld = std::max(ld, (int)(win[k].off - c.base) + win[k].n);
        }
        c.ld = ld, c.rows = 0, c.cols = 0;
        // column alignment: find shift c0 so that no window wraps around a row
        int c0 = 0;
        for (int attempt = 0; attempt < 2 && !fallback; attempt++) {
            bool ok = true;
            for (size_t k = i; k < j && ok; k++) {
                uint64_t rel = win[k].off - c.base + (uint64_t)c0;
                // (offsets inside a component nearly always fit 32 bits: a 32-bit remainder costs a third of a 64-bit one, and
                // this loop and the next run once per window of every list)
                const int col = rel <= 0xFFFFFFFFull ? (int)((uint32_t)rel % (uint32_t)ld) : (int)(rel % (uint64_t)ld);
                if (col + win[k].n > ld)
                    ok = false, c0 = (int)((uint64_t)ld - (win[k].off - c.base) % (uint64_t)ld) % ld;
            }
            if (ok)
                break;
            if (attempt == 1) {
                fallback = true;
                reason = "output windows do not share a row alignment";
            }
        }
        if ((uint64_t)c0 > c.base) {
            fallback = true;
            reason = "output window alignment precedes psi'";
        }
        if (!fallback) {
            c.base -= (uint64_t)c0;
            for (size_t k = i; k < j; k++) {
                uint64_t rel = win[k].off - c.base;
                int r0, cc;
                if (rel <= 0xFFFFFFFFull)
                    r0 = (int)((uint32_t)rel / (uint32_t)ld), cc = (int)((uint32_t)rel - (uint32_t)r0 * (uint32_t)ld);
                else
                    r0 = (int)(rel / (uint64_t)ld), cc = (int)(rel % (uint64_t)ld);
                c.rows = std::max(c.rows, r0 + win[k].m);
                c.cols = std::max(c.cols, cc + win[k].n);
            }
        }
        comps.push_back(c);
        i = j;
    }
    return comps;
}

// Every slot of the scratch (a W product, a sum, a staged operand) is followed by at least one padding element that no
// kernel ever writes (it stays zero from the allocation on): the one element a degenerate K = 1 / one-row operand
// fetch may touch behind the slot.  Returns the slot's footprint, a multiple of 16 elements: every slot starts on a
// 128-byte line, so the rows of a W whose width is a multiple of 16 never straddle cache lines.
inline uint64_t slot_elems(CompiledPlan &out, uint64_t off, uint64_t size) {
    out.scratch_pads.push_back(off + size);
    return (size + 16) & ~(uint64_t)15;
}

// see StageCopy (b2x_plan.hpp): called once all segments of a plan exist.  The kernel's A fetch (gg_body::lane_offsets)
// reaches one element behind an operand whose K (k-contiguous layout) is not a multiple of the 16-deep chunk, or whose row
// count (row-contiguous layout) is not a multiple of 16.
void stage_residual_reads(CompiledPlan &out, uint64_t arena_cap, uint64_t in_cap) {
    std::map<std::pair<uint64_t, uint64_t>, uint64_t> done[2]; // (offset, extent) -> scratch offset, per source
    for (GSeg &g : out.gsegs) {
        if (g.a_src > 1)
            continue; // the scratch is plan-owned: every slot is followed by a zero, and it only holds finite values
        const bool kmaj = g.a_sk != 1;
        if ((kmaj ? g.mr : g.K) % 16 == 0)
            continue; // whole granules only
        const uint64_t ext = (uint64_t)(g.mr - 1) * (uint64_t)g.a_sr + (uint64_t)(g.K - 1) * (uint64_t)g.a_sk + 1;
        if (g.a_off + ext + 1 <= (g.a_src == 0 ? arena_cap : in_cap))
            continue; // the element behind the operand is inside the buffer
        auto key = std::make_pair(g.a_off, ext);
        auto it = done[g.a_src].find(key);
        if (it == done[g.a_src].end()) {
            const uint64_t dst = out.scratch_elems;
            out.scratch_elems += slot_elems(out, dst, ext);
            out.stage.push_back(StageCopy{(uint32_t)g.a_src, g.a_off, dst, ext});
            it = done[g.a_src].emplace(key, dst).first;
        }
        g.a_src = 2, g.a_off = it->second;
    }
    out.stats.n_staged = out.stage.size();
}

} // namespace

int compile_plan(size_t n_pairs, const b2x_pair *pairs, size_t psi_len, size_t sigma_len, uint64_t arena_len,
                 uint64_t arena_cap, const b2x_plan_options *opt, CompiledPlan &out, std::string &err) {
    out = CompiledPlan();
    b2x_plan_stats &st = out.stats;
    st.n_pairs = n_pairs, st.psi_len = psi_len, st.sigma_len = sigma_len;
    PhaseClock pc;
    static thread_local ScratchArena mem_arena;
    ScratchArena *const mem = &mem_arena;
    ArenaScope mem_scope{mem_arena};
    // ---- validation + statistics -------------------------------------------------------------
    tvec<std::pair<uint64_t, uint64_t>> opext(mem);
    opext.reserve(2 * n_pairs);
    for (size_t i = 0; i < n_pairs; i++) {
        const b2x_pair &p = pairs[i];
        if (p.ta0 != 0 || p.tb1 != 0 || p.tb0 > 1 || p.ta1 > 1) {
            err = "pair " + std::to_string(i) + ": unsupported transpose flags (this path has ta0 = tb1 = 0)";
            return B2X_ERR_INVALID;
        }
        if (p.m0 <= 0 || p.n0 <= 0 || p.k0 <= 0 || p.m1 <= 0 || p.k1 != p.m0 || p.n1 != p.n0) {
            err = "pair " + std::to_string(i) + ": inconsistent dimensions (need k1 == m0, n1 == n0, all > 0)";
            return B2X_ERR_INVALID;
        }
        if (p.lda0 < p.k0 || p.ldb0 < (p.tb0 ? p.k0 : p.n0) || p.lda1 < (p.ta1 ? p.m1 : p.k1) || p.ldc1 < p.n1) {
            err = "pair " + std::to_string(i) + ": leading dimension smaller than row length";
            return B2X_ERR_INVALID;
        }
        if (std::max(std::max(p.lda0, p.ldb0), std::max(p.lda1, p.ldc1)) >= kMaxLeadingDim) {
            // the kernels form per-lane BYTE offsets inside a tile in 32 bits with 24-bit multiplies (row < 128, k < 16)
            err = "pair " + std::to_string(i) + ": leading dimension >= 2^22 elements is not supported";
            return B2X_ERR_INVALID;
        }
        uint64_t ex = (uint64_t)(p.m0 - 1) * p.lda0 + p.k0;
        uint64_t ey = p.tb0 ? (uint64_t)(p.n0 - 1) * p.ldb0 + p.k0 : (uint64_t)(p.k0 - 1) * p.ldb0 + p.n0;
        uint64_t ez = p.ta1 ? (uint64_t)(p.k1 - 1) * p.lda1 + p.m1 : (uint64_t)(p.m1 - 1) * p.lda1 + p.k1;
        uint64_t ev = (uint64_t)(p.m1 - 1) * p.ldc1 + p.n1;
        if (p.x_off + ex > psi_len || p.v_off + ev > sigma_len || p.y_off + ey > arena_len ||
            p.z_off + ez > arena_len) {
            err = "pair " + std::to_string(i) + ": operand runs past the end of psi / psi' / arena";
            return B2X_ERR_INVALID;
        }
        st.macs += (uint64_t)p.m0 * p.n0 * p.k0 + (uint64_t)p.m1 * p.n1 * p.k1;
        opext.emplace_back(p.y_off, ey);
        opext.emplace_back(p.z_off, ez);
    }
    { // (an operator block is named by many pairs: distinct (offset, extent) first — a hash set —, then the sort)
        size_t cap = 1024;
        while (cap < 2 * opext.size())
            cap <<= 1;
        tvec<std::pair<uint64_t, uint64_t>> set(cap, std::make_pair(~(uint64_t)0, (uint64_t)0), mem), uniq(mem);
        for (const auto &e : opext) {
            size_t i = (size_t)((e.first * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (set[i].first != ~(uint64_t)0 && set[i] != e)
                i = (i + 1) & (cap - 1);
            if (set[i].first == ~(uint64_t)0)
                set[i] = e, uniq.push_back(e);
        }
        opext.swap(uniq);
    }
    std::sort(opext.begin(), opext.end());
    {
        uint64_t cur_b = 0, cur_e = 0;
        for (auto &e : opext) {
            if (e.first >= cur_e) {
                st.op_elems_unique += cur_e - cur_b;
                cur_b = e.first, cur_e = e.first + e.second;
            } else
                cur_e = std::max(cur_e, e.first + e.second);
        }
        st.op_elems_unique += cur_e - cur_b;
    }
    if (n_pairs == 0)
        return B2X_OK;
    pc.lap("0 validate");
    // ---- output windows -> disjoint components -----------------------------------------------
    std::vector<Window> win(n_pairs);
    for (size_t i = 0; i < n_pairs; i++)
        win[i] = Window{pairs[i].v_off, pairs[i].m1, pairs[i].n1, pairs[i].ldc1, (uint32_t)i};
    std::vector<Component> comps = build_components(win, out.fallback, out.fallback_reason);
    st.n_targets = comps.size();
    st.fallback = out.fallback;
    if (out.fallback)
        return B2X_OK;
    pc.lap("1 components");
    // ---- tiles, parts, items -----------------------------------------------------------------
    struct HostTile {
        int cls;
        std::vector<DPart> parts;
        std::vector<double> cost;
    };
    int64_t total_cost = 0;
    uint64_t cls_macs[kNumClasses] = {0, 0, 0};
    double cls_alg[kNumClasses] = {0, 0, 0};
    std::vector<HostTile> htiles;
    std::vector<const Component *> big; // components routed to the two-stage path
    const int two_stage = opt ? opt->two_stage : 0;
    struct SectorShape {
        bool must;
        double macs;
        int max_k;
    };
    std::vector<SectorShape> shapes(comps.size());
    bool any_must = false;
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const Component &c = comps[ci];
        int max_k0 = 0, max_k = 0;
        double sector_macs = 0;
        for (uint32_t wi = c.w_begin; wi < c.w_end; wi++) {
            const b2x_pair &p = pairs[win[wi].pair];
            max_k0 = std::max(max_k0, (int)p.k0);
            max_k = std::max(max_k, std::max((int)p.k0, (int)p.k1));
            sector_macs += (double)p.m0 * p.n0 * p.k0 + (double)p.m1 * p.n1 * p.k1;
        }
        shapes[ci].must = c.rows > 128 || c.cols > 128 || max_k0 > 512;
        shapes[ci].macs = sector_macs, shapes[ci].max_k = max_k;
        any_must = any_must || shapes[ci].must;
    }
    // Since the grouped-GEMM kernel has a low-register instantiation for short tiles (four waves per SIMD) and its own
    // cheap per-segment arithmetic, it beats the fused wave kernel on every plan measured, small ones included (golden
    // N2 / H10 / Hubbard plans 1.0-2.8x, H10 at M=500 2.6x: the fused path pays three launches, recomputes stage 0 per
    // row tile and reads every partial slab back; profiles/r02_route_probe.txt): auto routing sends every plan there.
    // The fused wave kernel stays available (b2x_plan_options.two_stage = -1) and tested.
    any_must = true;
    for (const Component &c : comps) {
        // auto routing: the fused kernel recomputes stage 0 per row tile and keeps W in registers, which
        // pays for sectors that fit one tile; tall / wide / deep ones go to the grouped-GEMM path
        // Routing is per PLAN: as soon as one sector is taller / wider / deeper than a fused tile, or the plan has more
        // than 1e9 MACs, EVERY sector takes the grouped-GEMM path: the plan pays that path's fixed costs (two more
        // launches, the W round trip) once, its kernel runs the MFMAs better, and only there the algebra of DESIGN.md 4.5
        // applies.  Measured on the Cr2 plan: M=250 1.74 ms fused -> 1.30 ms; M=500 8.0 -> 6.9 ms when the small sectors
        // followed the large ones; mixed plans (some sectors fused) were never faster than all-grouped.  Small plans
        // (the N2 / H10 test sizes) stay on the fused wave kernel.
        const bool large = any_must && !(opt && opt->tile_n > 0); // (a forced fused-tile class, as the tests ask for, keeps the wave kernel)
        if (two_stage > 0 || (two_stage == 0 && large)) {
            big.push_back(&c);
            continue;
        }
        // class by shape: columns decide the number of waves, rows the fragment count
        int cls;
        if (opt && opt->tile_n > 0) { // forced class (tests): 16 -> 32x32 wave, 32 -> 128x32 wave, 64 -> 64x32 wave
            cls = opt->tile_n <= 16 ? 0 : (opt->tile_n <= 32 ? 2 : 1);
        } else {
            if (c.cols <= 32 && c.rows <= 32)
                cls = 0;
            else if (c.rows <= 64)
                cls = 1;
            else
                cls = 2; // (sectors taller than 128 rows only get here when the two-stage path is disabled)
        }
        const KClass &K = kClasses[cls];
        const int TM = K.tmf * 16, TN = K.nw * 16, K1C = K.k1f * 16;
        std::vector<int> rc = balanced_cuts(c.rows, TM), cc = balanced_cuts(c.cols, TN);
        int nrt = (int)rc.size() - 1, nct = (int)cc.size() - 1;
        size_t t0 = htiles.size();
        htiles.resize(t0 + (size_t)nrt * nct);
        for (int a = 0; a < nrt; a++)
            for (int b = 0; b < nct; b++) {
                DTile t{};
                t.sigma_off = c.base + (uint64_t)rc[a] * c.ld + cc[b];
                t.ld = c.ld, t.rows = rc[a + 1] - rc[a], t.cols = cc[b + 1] - cc[b];
                out.tiles.push_back(t);
                htiles[t0 + (size_t)a * nct + b].cls = cls;
            }
        for (uint32_t wi = c.w_begin; wi < c.w_end; wi++) {
            const Window &w = win[wi];
            const b2x_pair &p = pairs[w.pair];
            uint64_t rel = w.off - c.base;
            int row0 = (int)(rel / (uint64_t)c.ld), col0 = (int)(rel % (uint64_t)c.ld);
            int a0 = (int)(std::upper_bound(rc.begin(), rc.end(), row0) - rc.begin()) - 1;
            int b0 = (int)(std::upper_bound(cc.begin(), cc.end(), col0) - cc.begin()) - 1;
            for (int a = a0; a < nrt && rc[a] < row0 + w.m; a++)
                for (int b = b0; b < nct && cc[b] < col0 + w.n; b++) {
                    int ra = std::max(row0, rc[a]), rb = std::min(row0 + w.m, rc[a + 1]);
                    int ca = std::max(col0, cc[b]), cb = std::min(col0 + w.n, cc[b + 1]);
                    HostTile &ht = htiles[t0 + (size_t)a * nct + b];
                    {
                        DPart d{};
                        int r_lo = ra - row0, c_lo = ca - col0;
                        d.x_off = p.x_off;
                        d.ldx = p.lda0;
                        if (p.tb0) // op(Y)[k][c] = Y[c][k]
                            d.y_off = p.y_off + (uint64_t)c_lo * p.ldb0, d.sky = 1, d.scy = p.ldb0;
                        else
                            d.y_off = p.y_off + (uint64_t)c_lo, d.sky = p.ldb0, d.scy = 1;
                        if (p.ta1) // op(Z)[r][k] = Z[k][r]
                            d.z_off = p.z_off + (uint64_t)r_lo, d.srz = 1, d.skz = p.lda1;
                        else
                            d.z_off = p.z_off + (uint64_t)r_lo * p.lda1, d.srz = p.lda1, d.skz = 1;
                        d.alpha = p.alpha0 * p.alpha1;
                        d.k0 = p.k0, d.k1 = p.k1;
                        d.mr = (int16_t)(rb - ra), d.nc = (int16_t)(cb - ca);
                        d.tr0 = (int16_t)(ra - rc[a]), d.tc0 = (int16_t)(ca - cc[b]);
                        // cost model: MFMA issue slots (16x16x4 granules) of both stages
                        double c0 = (double)round_up(p.k1, 16) * round_up(d.nc, 16) * round_up(p.k0, 4);
                        double c1 = (double)round_up(d.mr, 16) * round_up(d.nc, 16) * round_up(p.k1, 4);
                        double cst = c0 + c1 + 4096.0 * ceil_div(p.k1, K1C);
                        ht.parts.push_back(d);
                        ht.cost.push_back(cst);
                        total_cost += (int64_t)cst;
                        cls_macs[cls] += (uint64_t)p.k1 * d.nc * p.k0 + (uint64_t)d.mr * d.nc * p.k1;
                        // algorithmic share: stage 0 of a pair counted once over its row tiles
                        cls_alg[cls] += (double)p.k1 * d.nc * p.k0 * ((double)d.mr / p.m1) + (double)d.mr * d.nc * p.k1;
                    }
                }
        }
    }
    st.n_tiles = out.tiles.size();
    // work-item size: aim for ~16 items per CU, but never below ~0.5 MMAC-equivalents
    // Work-item sizes.  Every item writes a partial slab of its tile that hpsi_reduce has to read back, so more items
    // than the chip can run at once only buy tail balance at the price of reduce traffic (at M=250 the reduce was 38 %
    // of an H.psi with 32 768 wave items: 2.2 ms; ~10 000 items: 1.75 ms; ~4 000: 2.6 ms, too few to balance).  Fused
    // classes: about three wave items per wave slot (256 CUs x 4 SIMDs x 3-4 waves).  Grouped GEMM: enough workgroup items for ~8 rounds over the 512 workgroup
    // slots, between 2e6 (small plans need the parallelism) and 3e7 (large plans: fewer, longer items amortise the
    // tile store) MFMA-slot units.
    double gg_macs_total = 0;
    for (size_t ci = 0; ci < comps.size(); ci++)
        if (two_stage > 0 || (two_stage == 0 && any_must))
            gg_macs_total += shapes[ci].macs;
    const bool forced = opt && opt->item_macs > 0;
    const double item_cost = forced ? (double)opt->item_macs : std::max((double)total_cost / 10240.0, 131072.0);
    // (upper bound 3e7: at M=1000 6e7 left too few items per launch, 12.6 -> 11.5 ms; no difference at M >= 2000)
    // (plans below ~1.5 GMAC — H10/STO-6G at M=500: 0.48 GMAC in 50 psi' tiles — are latency-bound launches of a few hundred
    //  items that walk ~50 segments each: items of 3e5 instead of 2e6 slot units fill the chip, stage 1 0.123 -> 0.084 ms and the
    //  whole H.psi 0.132 -> 0.113 ms (more partial slabs for the reduce); above that size smaller items lose to the reduce:
    //  profiles/r03_item_size_small_plans_ab.txt)
    const double step_item_lb = gg_macs_total < 1.5e9 ? 3.0e5 : 2.0e6;
    const double step_item_cost = forced ? (double)opt->item_macs : std::min(3.0e7, std::max(step_item_lb, gg_macs_total / 4096.0));
    uint64_t slab = 0;
    for (size_t t = 0; t < htiles.size(); t++) {
        HostTile &ht = htiles[t];
        DTile &dt = out.tiles[t];
        ClassWork &cw = out.cls[ht.cls];
        dt.slab_off = slab;
        dt.n_items = 0;
        if (ht.parts.empty())
            continue;
        double tile_cost = std::accumulate(ht.cost.begin(), ht.cost.end(), 0.0);
        const double ic = forced ? std::max(item_cost / 8.0, 65536.0) : item_cost; // (forced: test knob, 1/8 per wave item)
        int n_it = std::max(1, (int)std::lround(tile_cost / ic));
        double per = tile_cost / n_it, acc = 0;
        uint32_t pb = (uint32_t)cw.parts.size(), begin = pb;
        int made = 0;
        for (size_t k = 0; k < ht.parts.size(); k++) {
            cw.parts.push_back(ht.parts[k]);
            acc += ht.cost[k];
            bool last = k + 1 == ht.parts.size();
            if (last || (acc >= per * (made + 1) && made + 1 < n_it)) {
                DItem it{};
                it.part_begin = begin, it.part_end = (uint32_t)cw.parts.size();
                it.slab_off = slab, it.rows = dt.rows, it.cols = dt.cols;
                cw.items.push_back(it);
                slab += (uint64_t)dt.rows * dt.cols;
                begin = it.part_end;
                made++;
            }
        }
        dt.n_items = made;
        st.n_parts += ht.parts.size();
        std::vector<DPart>().swap(ht.parts);
    }
    out.slab_elems = slab;
    // longest items first (tail balance); stable so equal items keep plan order
    for (int k = 0; k < kNumClasses; k++) {
        ClassWork &cw = out.cls[k];
        std::stable_sort(cw.items.begin(), cw.items.end(), [](const DItem &a, const DItem &b) {
            return (a.part_end - a.part_begin) > (b.part_end - b.part_begin);
        });
        st.n_items += cw.items.size();
    }
    pc.lap("2 fused tiles");
    // ---- two-stage path: W = alpha X op(Y) to scratch, then psi' tiles += op(Z) W ---------------
    uint64_t gg_macs = 0;
    if (!big.empty()) {
        // narrow psi' sectors (small bond dimensions) waste most of a 128-column tile: such plans use 2-wave workgroups
        // (MAC-weighted mean output width of the pairs; measured on the Cr2 plan: 64-column tiles / 2-wave workgroups win
        // below ~300 — M=250 1.30 -> 0.84 ms, M=500 3.5 -> 2.6 ms —, tie at 240 (M=1000), 128 columns win above)
        double wsum = 0, wn = 0;
        for (const Component *c : big)
            for (uint32_t wi = c->w_begin; wi < c->w_end; wi++) {
                const b2x_pair &p = pairs[win[wi].pair];
                const double w = (double)p.m0 * p.n0 * p.k0 + (double)p.m1 * p.n1 * p.k1;
                wsum += w, wn += w * p.n0;
            }
        const int TN = (wsum > 0 && wn / wsum < 300.0) ? 64 : kGGTileN;
        out.gg_tile_n = TN;
        // Height of the short tile class.  Its kernel instantiation serves tiles of up to 3 row fragments with <= 128 VGPRs
        // (4 waves per SIMD), or of up to 5 with <= 168 (3 waves per SIMD; the tall instantiation: 2).  Plans whose work
        // sits in 4-5-fragment tiles (the M=1000-2000 class: 66 % / 44 % of the MFMA issue slots of the Cr2 plan) take the
        // second (M=1000 -3 %, M=2000 -2 % on one box), plans with a real share of <= 3-fragment tiles keep the first
        // (uniform-M Cr2 M=1000: +3 % with 5).  Shares estimated from the pairs' row counts, MAC-weighted.
        {
            double w3 = 0, w5 = 0, w7 = 0, wt = 0;
            for (const Component *c : big)
                for (uint32_t wi = c->w_begin; wi < c->w_end; wi++) {
                    const b2x_pair &p = pairs[win[wi].pair];
                    const double w = (double)p.m0 * p.n0 * p.k0 + (double)p.m1 * p.n1 * p.k1;
                    for (int rows : {p.m1, p.k1}) {
                        const std::vector<int> &rc = unit_cuts_m(rows);
                        for (size_t a = 0; a + 1 < rc.size(); a++) {
                            const int fr = ceil_div(rc[a + 1] - rc[a], kGGRowUnit);
                            const double ws = 0.5 * w * (rc[a + 1] - rc[a]) / rows;
                            wt += ws;
                            if (fr <= kGGShortFrags)
                                w3 += ws;
                            else if (fr <= kGGMidFrags)
                                w5 += ws;
                            else if (fr >= 7)
                                w7 += ws;
                        }
                    }
                }
            static const int sf_env = getenv("B2X_SHORT_FRAGS") ? atoi(getenv("B2X_SHORT_FRAGS")) : 0;
            out.short_frags = sf_env == kGGShortFrags || sf_env == kGGMidFrags
                                  ? sf_env
                                  : (wt > 0 && w3 / wt < 0.25 && w5 / wt > 0.3 ? kGGMidFrags : kGGShortFrags);
            if (pc.on)
                fprintf(stderr, "[b2x plan] row-tile shares (MAC-weighted): <=3 frags %.3f, 4-5 %.3f, 6 %.3f, 7-8 %.3f\n", w3 / wt,
                        w5 / wt, (wt - w3 - w5 - w7) / wt, w7 / wt);
            // Tile-height cap.  Plans whose work sits in tiles of 4-6 row fragments rather than in full 8-fragment ones (the
            // TRUE Cr2 structures at M=2000-4000: 34-60 % of the work in 7-8-fragment tiles, against 74 % for the x16 plan)
            // run faster when NO tile is taller than 5 fragments: the whole plan then runs on the <= 168-VGPR instantiation
            // at three waves per SIMD (true M=2000: +7 %, its x2: +3.5 %, x16: +-0; same-box sweeps in
            // profiles/r03_tile_height_cap_true_structures.txt).  It requests more operand bytes (5-fragment tiles: 6.1 MAC
            // per byte against 8), which costs nothing: the kernel does not wait for the fabric
            // (profiles/r03_l2_window_probe.txt).  Plans with a real share of <= 3-fragment tiles keep the four-waves-per-SIMD
            // class for them (true M=1000: -1.5 % with the cap).  B2X_MAX_UNITS overrides.
            out.cap_units = (wt > 0 && w3 / wt < 0.25 && w7 / wt < 0.65) ? kGGMidFrags : 0;
            if (out.cap_units && !(sf_env == kGGShortFrags || sf_env == kGGMidFrags))
                out.short_frags = kGGMidFrags;
        }
        pc.lap("3 class estimates");
        const int short_frags = out.short_frags;
        // Short tiles of plans with narrow sectors run on ONE-WAVE workgroups of 32 columns (gg_kernel<CF, NW = 1, ...,
        // TMAX = kGGNarrowFrags>: 116 VGPRs, 8 KB of LDS, 16 independent workgroups per CU): in a 2-wave workgroup the
        // second wave of a tile of <= 32 columns only helps staging A (at M=250 that is 59 % of the stage-0 chunks), yet
        // holds a wave slot.  Row ranges of up to 48 rows are cut into tiles of at most kGGNarrowFrags fragments, and the
        // columns of such a row tile at 32 instead of TN.  B2X_NARROW=0 keeps the 2-wave short-tile instantiation.
        static const int narrow_env = getenv("B2X_NARROW") ? atoi(getenv("B2X_NARROW")) : 1;
        // Only where segments are short as well (MAC-weighted mean output width < 90: the M=250 class; measured on one box:
        // Cr2 M=250 kernel time -11 %, M=500 +22 %, M=1000 +3 %, uniform-M Cr2 M=1000 +17 % — with longer K the two waves
        // of a 64-column tile share the A staging to better effect than two 1-wave workgroups that stage it twice).
        static const double narrow_w = getenv("B2X_NARROW_W") ? atof(getenv("B2X_NARROW_W")) : 90.0;
        const bool use_narrow = narrow_env != 0 && TN == 64 && wsum > 0 && wn / wsum < narrow_w;
        out.short_narrow = use_narrow;
        const int short_rows = (use_narrow ? kGGNarrowFrags : short_frags) * kGGRowUnit;
        static const int max_units_env = getenv("B2X_MAX_UNITS") ? atoi(getenv("B2X_MAX_UNITS")) : 0; // (probe)
        const int max_units = max_units_env > 0 ? max_units_env : (out.cap_units ? out.cap_units : kGGTileM / kGGRowUnit);
        auto row_cuts = [&](int total) -> const std::vector<int> & {
            return use_narrow && total <= kGGShortFrags * kGGRowUnit ? unit_cuts_m(total, kGGNarrowFrags) : unit_cuts_m(total, max_units);
        };
        auto col_tile = [&](int rows) { return use_narrow && rows <= short_rows ? kGGNarrowN : TN; };
        // effective pairs of this path: an operator pre-sum (below) replaces the second operator of a merged pair by a
        // block in the plan's own buffer (source 2 = scratch, whose first aux_len elements persist across executions)
        // The segment list of a large plan is 10^6-10^7 records of 64 bytes: grown by doubling it is copied (and its fresh
        // pages faulted in) three times over.  An upper estimate of its length is reserved instead (address space only;
        // x16 Cr2 plan: compile 669 -> 594 ms): tiles of a pair's windows in both stages, plus one cut per window for
        // the row ranges other windows of the sector split.
        {
            size_t est = 0;
            for (size_t i = 0; i < n_pairs; i++) {
                const b2x_pair &p = pairs[i];
                const size_t ct = (size_t)ceil_div(p.n0, use_narrow ? kGGNarrowN : TN) + 1;
                est += ((size_t)ceil_div(p.m1, kGGTileM) + 2) * ct + ((size_t)ceil_div(std::max(p.k1, p.m1), kGGTileM) + 1) * ct;
            }
            try { // (an optimisation only: a host that refuses the address space — strict overcommit — grows the lists as before)
                out.gsegs.reserve(est), out.gitems.reserve(est);
                advise_huge(out.gsegs.data(), est * sizeof(GSeg)), advise_huge(out.gitems.data(), est * sizeof(GItem));
            } catch (const std::bad_alloc &) {
            }
        }
        tvec<b2x_pair> ep(pairs, pairs + n_pairs, mem);
        tvec<uint8_t> zsrc(n_pairs, 0, mem), ysrc(n_pairs, 0, mem);
        uint64_t aux_len = 0;
        const uint64_t budget = (uint64_t)(opt && opt->scratch_mb > 0 ? opt->scratch_mb : 16384) * (1u << 17);
        // work in component order; a super-step closes when the W scratch budget is reached
        // A pair is V += alpha * op(Z) . X . op(Y).  The reference always forms W = X . op(Y) first; the product is
        // associative, so per pair the cheaper order is taken: (op(Z) . X) . op(Y) costs m1 k1 k0 + m1 k0 n MACs against
        // k1 k0 n + m1 k1 n (on the Cr2 mid-chain plan 42 % of the pairs flip and 24 % of the MACs disappear; the result
        // differs from the reference's by rounding only).  flip: stage 0 writes W' = alpha op(Z) X (m1 x k0) to the scratch,
        // stage 1 accumulates W' . op(Y).
        // Pairs that share the stage-0 product (same X slice and same right operator block for W, same left operator
        // block and same X slice for W') compute it ONCE per super-step: the shared W is stored unscaled and every
        // pair's alpha is applied per segment in stage 1 (the SCALED kernel variant).  On the Cr2 mid-chain plan 29 % of
        // the pairs reuse a product of another pair (H = sum of left (x) right operator products: many left operators meet
        // the same right operator), another 13 % of the MACs.  To bring the sharers into one super-step the pairs are
        // processed in the order of their stage-0 key, not of their psi' sector (hpsi_reduce adds the slabs of every
        // super-step into psi', so a sector may be spread over super-steps).
        struct PW {
            const Component *c;
            uint32_t wi;
            uint64_t w_off;
            bool flip;
            bool owner; // generates the stage-0 items of its W
        };
        const bool allow_flip = !(opt && opt->keep_order == 1);
        tvec<PW> cur(mem);
        uint64_t used = 0;
        auto flush = [&]() {
            if (cur.empty())
                return;
            pc.lap("4b super-step fill");
            SuperStep ss{};
            ss.sum_begin = (uint32_t)out.sum_work.size();
            uint64_t sum_extra = 0; // scratch taken by the sums of this step, behind its W slots
            const uint32_t s0_begin = (uint32_t)out.gitems.size();
            // Locality key of a stage-0 item (s0_order below).  The products that read one B operand — X_i . op(Y) for the
            // psi slices X_i that meet one right-operator block Y, op(Z_i) . X for the left-operator blocks that meet one
            // psi slice X — are the row blocks of ONE tall product [X_1; X_2; ...] . op(Y): its tiles of one column read the
            // same B tile, those of one row the same A rows.  key = (B operand, 4 x 4 block of tiles of that tall product).
            tvec<uint64_t> s0_key(mem);
            struct BGroup {
                uint32_t id, row_tiles;
            };
            std::pmr::unordered_map<uint64_t, BGroup> bgroups(mem);
            bgroups.reserve(cur.size());
            auto s0_group = [&](uint64_t b_id, size_t n_row_tiles) -> BGroup {
                auto it = bgroups.find(b_id);
                if (it == bgroups.end())
                    it = bgroups.emplace(b_id, BGroup{(uint32_t)bgroups.size(), 0u}).first;
                BGroup r = it->second;
                it->second.row_tiles += (uint32_t)n_row_tiles;
                return r;
            };
            auto s0_push_key = [&](const BGroup &bg, size_t a, size_t b) {
                static const int sb = getenv("B2X_XCD_BLK") ? atoi(getenv("B2X_XCD_BLK")) : 4; // (probe: block edge)
                s0_key.push_back(((uint64_t)bg.id << 24) | ((uint64_t)(((bg.row_tiles + a) / sb) & 0xFFF) << 12) | (uint64_t)((b / sb) & 0xFFF));
            };
            // stage 0: tiles of every W
            for (const PW &pw : cur) {
                const b2x_pair &p = ep[win[pw.wi].pair];
                if (!pw.owner)
                    continue;
                if (pw.flip) { // W'(m1 x k0) = op(Z)(m1 x k1) . X(k1 x k0)
                    const std::vector<int> &rc = row_cuts(p.m1);
                    const BGroup bg = s0_group(((uint64_t)1 << 63) | p.x_off, rc.size() - 1);
                    for (size_t a = 0; a + 1 < rc.size(); a++) {
                        const std::vector<int> &cc = wave_cuts_m(p.k0, col_tile(rc[a + 1] - rc[a]));
                        for (size_t b = 0; b + 1 < cc.size(); b++) {
                            s0_push_key(bg, a, b);
                            GSeg g{};
                            g.a_src = zsrc[win[pw.wi].pair];
                            if (p.ta1)
                                g.a_off = p.z_off + (uint64_t)rc[a], g.a_sr = 1, g.a_sk = p.lda1;
                            else
                                g.a_off = p.z_off + (uint64_t)rc[a] * p.lda1, g.a_sr = p.lda1, g.a_sk = 1;
                            g.b_src = 1, g.b_off = p.x_off + (uint64_t)cc[b], g.b_sk = p.lda0, g.b_sc = 1;
                            g.K = p.k1, g.alpha = 1.0;
                            g.mr = rc[a + 1] - rc[a], g.nc = cc[b + 1] - cc[b];
                            GItem it{};
                            it.seg_begin = (uint32_t)out.gsegs.size(), it.seg_end = it.seg_begin + 1;
                            it.out_off = pw.w_off + (uint64_t)rc[a] * p.k0 + cc[b], it.out_ld = p.k0;
                            it.rows = g.mr, it.cols = g.nc, it.alpha = 1.0, it.out_kind = 1;
                            out.gsegs.push_back(g);
                            out.gitems.push_back(it);
                            gg_macs += (uint64_t)g.mr * g.nc * g.K;
                        }
                    }
                    continue;
                }
                const std::vector<int> &rc = row_cuts(p.k1);
                const BGroup bg = s0_group(((uint64_t)ysrc[win[pw.wi].pair] << 62) | p.y_off, rc.size() - 1);
                for (size_t a = 0; a + 1 < rc.size(); a++) {
                    const std::vector<int> &cc = wave_cuts_m(p.n0, col_tile(rc[a + 1] - rc[a]));
                    for (size_t b = 0; b + 1 < cc.size(); b++) {
                        s0_push_key(bg, a, b);
                        GSeg g{};
                        g.a_src = 1, g.a_off = p.x_off + (uint64_t)rc[a] * p.lda0, g.a_sr = p.lda0, g.a_sk = 1;
                        g.b_src = ysrc[win[pw.wi].pair];
                        if (p.tb0)
                            g.b_off = p.y_off + (uint64_t)cc[b] * p.ldb0, g.b_sk = 1, g.b_sc = p.ldb0;
                        else
                            g.b_off = p.y_off + (uint64_t)cc[b], g.b_sk = p.ldb0, g.b_sc = 1;
                        g.K = p.k0, g.alpha = 1.0;
                        g.mr = rc[a + 1] - rc[a], g.nc = cc[b + 1] - cc[b];
                        GItem it{};
                        it.seg_begin = (uint32_t)out.gsegs.size(), it.seg_end = it.seg_begin + 1;
                        it.out_off = pw.w_off + (uint64_t)rc[a] * p.n0 + cc[b], it.out_ld = p.n0;
                        it.rows = g.mr, it.cols = g.nc, it.alpha = 1.0, it.out_kind = 1;
                        out.gsegs.push_back(g);
                        out.gitems.push_back(it);
                        gg_macs += (uint64_t)g.mr * g.nc * g.K;
                    }
                }
            }
            pc.lap("5 stage-0 items");
            const uint32_t s1_begin = (uint32_t)out.gitems.size();
            tvec<uint32_t> tile_of_item(mem); // stage-1 items in creation order -> running number of their tile
            uint32_t tile_seq = 0, tile_sector = 0;
            ss.tile_begin = (uint32_t)out.gtiles.size();
            // stage 1: per component, per psi' tile, the segments of this step's pairs
            uint64_t slab = 0;
            // (the step's pairs arrive in stage-0 key order: regroup them by psi' sector, plan order inside a sector)
            // (components are runs of the offset-sorted window array, and a window belongs to one pair: window order IS
            // sector order; one pass over a window-indexed table instead of a comparison sort)
            {
                tvec<uint32_t> at(n_pairs, ~0u, mem);
                for (size_t k = 0; k < cur.size(); k++)
                    at[cur[k].wi] = (uint32_t)k;
                tvec<PW> byw(mem);
                byw.reserve(cur.size());
                for (size_t w = 0; w < n_pairs; w++)
                    if (at[w] != ~0u)
                        byw.push_back(cur[at[w]]);
                cur.swap(byw);
            }
            tvec<GSeg> fsegs(mem);
            tvec<uint32_t> ftile(mem);
            const uint32_t per_step = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, (used >> 10) / 262144));
            size_t i = 0;
            while (i < cur.size()) {
                size_t j = i;
                while (j < cur.size() && cur[j].c == cur[i].c)
                    j++;
                const Component &c = *cur[i].c;
                // rows: cut first at the boundaries of the row slices (three_rotate windows stack the sector
                // from a few row ranges), then balance each range into tiles; columns: balanced tiles
                std::vector<int> bounds{0, c.rows};
                for (size_t q = i; q < j; q++) {
                    const Window &w = win[cur[q].wi];
                    int row0 = (int)((w.off - c.base) / (uint64_t)c.ld);
                    bounds.push_back(row0), bounds.push_back(row0 + w.m);
                }
                std::sort(bounds.begin(), bounds.end());
                bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
                std::vector<int> rc;
                for (size_t bi = 0; bi + 1 < bounds.size(); bi++) {
                    const std::vector<int> &sub = row_cuts(bounds[bi + 1] - bounds[bi]);
                    for (size_t k = 0; k + 1 < sub.size(); k++)
                        rc.push_back(bounds[bi] + sub[k]);
                }
                rc.push_back(c.rows);
                // column cuts per ROW TILE: short row tiles (1-wave workgroups) are cut at 32 columns, the others at TN
                const std::vector<int> cc_wide = wave_cuts(c.cols, TN), cc_narrow = wave_cuts(c.cols, kGGNarrowN);
                int nrt = (int)rc.size() - 1;
                std::vector<const std::vector<int> *> ccs(nrt);
                std::vector<size_t> tbase(nrt + 1, 0);
                int nct_max = 1;
                for (int a = 0; a < nrt; a++) {
                    ccs[a] = col_tile(rc[a + 1] - rc[a]) == TN ? &cc_wide : &cc_narrow;
                    tbase[a + 1] = tbase[a] + (ccs[a]->size() - 1);
                    nct_max = std::max(nct_max, (int)ccs[a]->size() - 1);
                }
                // segments of the sector in creation order (fsegs, with their tile in ftile), then scattered tile by tile
                // into the plan's list (a list per tile cost an allocation per tile and a second copy of every segment)
                fsegs.clear(), ftile.clear();
                tvec<uint32_t> tstart(tbase[nrt] + 1, 0, mem);
                tvec<double> tcost(tbase[nrt], 0.0, mem);
                pc.lap("6a cuts");
                // Distributive law: pairs of this sector that multiply the SAME operator block into the SAME window
                // (op(Z) . W_i with one Z, or W'_i . op(Y) with one Y) first sum their scaled stage-0 products,
                // S = sum_i alpha_i W_i, in an element-wise pass between the stages, then take ONE stage-1 product.
                // Worth it only where the saved MACs outweigh the pass at HBM speed: (g - 1)/(g + 1) x (rows of Z, or
                // columns of op(Y)) > 64 MAC per element moved.
                tvec<uint64_t> merged_off(j - i, ~(uint64_t)0, mem); // S offset of a group's first member
                tvec<uint8_t> merged_skip(j - i, 0, mem);             // later members: no segment of their own
                if (allow_flip && j - i > 1 && !getenv("B2X_NO_MERGE")) {
                    struct MK {
                        uint64_t k[6];
                        uint32_t q;
                    };
                    tvec<MK> mk(mem);
                    mk.reserve(j - i);
                    for (size_t q = i; q < j; q++) {
                        const Window &w = win[cur[q].wi];
                        const b2x_pair &p = ep[w.pair];
                        // (a group is only summed where dim (g - 1)/(g + 1) > 64, below: pairs whose dim is <= 64 — every pair
                        // of a small-M plan — never qualify and need not be keyed and sorted)
                        if ((cur[q].flip ? p.n0 : p.m1) <= 64)
                            continue;
                        mk.emplace_back();
                        MK &m = mk.back();
                        m.q = (uint32_t)q;
                        m.k[0] = cur[q].flip, m.k[1] = cur[q].flip ? p.y_off : p.z_off, m.k[2] = w.off;
                        m.k[3] = ((uint64_t)p.m1 << 32) | (uint32_t)p.n0, m.k[4] = cur[q].flip ? (uint64_t)p.k0 : (uint64_t)p.k1;
                        m.k[5] = cur[q].flip ? (((uint64_t)p.ldb0 << 8) | p.tb0 | ((uint64_t)ysrc[w.pair] << 60))
                                             : (((uint64_t)p.lda1 << 8) | p.ta1 | ((uint64_t)zsrc[w.pair] << 60));
                    }
                    std::sort(mk.begin(), mk.end(), [](const MK &x, const MK &y) {
                        for (int k = 0; k < 6; k++)
                            if (x.k[k] != y.k[k])
                                return x.k[k] < y.k[k];
                        return x.q < y.q;
                    });
                    for (size_t a = 0; a < mk.size();) {
                        size_t b = a + 1;
                        while (b < mk.size() && std::equal(mk[a].k, mk[a].k + 6, mk[b].k))
                            b++;
                        const size_t gsz = b - a;
                        const b2x_pair &p0 = ep[win[cur[mk[a].q].wi].pair];
                        const bool fl = cur[mk[a].q].flip;
                        const double dim = fl ? (double)p0.n0 : (double)p0.m1;
                        if (gsz > 1 && dim * (double)(gsz - 1) / (double)(gsz + 1) > 64.0) {
                            const int srows = fl ? p0.m1 : p0.k1, scols = fl ? p0.k0 : p0.n0;
                            const uint64_t s_off = used + sum_extra;
                            sum_extra += slot_elems(out, s_off, (uint64_t)srows * scols);
                            const uint32_t eb = (uint32_t)out.sum_entries.size();
                            for (size_t x = a; x < b; x++) {
                                const b2x_pair &px = ep[win[cur[mk[x].q].wi].pair];
                                OEntry e{};
                                e.a_off = cur[mk[x].q].w_off, e.b_off = 0, e.alpha = px.alpha0 * px.alpha1;
                                e.a_rs = scols, e.a_cs = 1, e.b_rs = e.b_cs = 0, e.a_src = 1, e.b_src = 2;
                                out.sum_entries.push_back(e);
                                if (x > a)
                                    merged_skip[mk[x].q - i] = 1;
                            }
                            merged_off[mk[a].q - i] = s_off;
                            out.stats.n_merged_groups++, out.stats.n_merged_members += gsz;
                            OWork wk{};
                            wk.out_off = s_off, wk.ld = -scols, wk.rows = srows, wk.cols = scols; // ld < 0: assign
                            wk.rpt = 16; // outer_build_k<16>
                            wk.entry_begin = eb, wk.entry_end = (uint32_t)out.sum_entries.size();
                            const uint32_t nseg = (uint32_t)ceil_div(scols, kOuterTileCols), nstrip = (uint32_t)ceil_div(srows, wk.rpt);
                            // tiles per work unit (one wave): more for large groups, and for steps whose W products add up to
                            // millions of tiles (x16 Cr2 plan: 2.5 M units of 40 bytes took a third of the compile time)
                            const uint32_t ntile = nseg * nstrip, per = std::max<uint32_t>(ntile >= 4096 ? 4 : (ntile >= 1024 ? 2 : 1), per_step);
                            for (uint32_t t0 = 0; t0 < ntile; t0 += per) {
                                wk.t_begin = t0, wk.t_end = std::min(ntile, t0 + per);
                                out.sum_work.push_back(wk);
                            }
                        }
                        a = b;
                    }
                }
                pc.lap("6b merge groups");
                for (size_t q = i; q < j; q++) {
                    if (merged_skip[q - i])
                        continue;
                    const bool msum = merged_off[q - i] != ~(uint64_t)0; // reads the group's sum, already scaled
                    const uint64_t src_off = msum ? merged_off[q - i] : cur[q].w_off;
                    const Window &w = win[cur[q].wi];
                    const b2x_pair &p = ep[w.pair];
                    uint64_t rel = w.off - c.base;
                    int row0 = (int)(rel / (uint64_t)c.ld), col0 = (int)(rel % (uint64_t)c.ld);
                    int a0 = (int)(std::upper_bound(rc.begin(), rc.end(), row0) - rc.begin()) - 1;
                    for (int a = a0; a < nrt && rc[a] < row0 + w.m; a++) {
                        const std::vector<int> &cc = *ccs[a];
                        const int nct = (int)cc.size() - 1;
                        int b0 = (int)(std::upper_bound(cc.begin(), cc.end(), col0) - cc.begin()) - 1;
                        for (int b = b0; b < nct && cc[b] < col0 + w.n; b++) {
                            int ra = std::max(row0, rc[a]), rb = std::min(row0 + w.m, rc[a + 1]);
                            int ca = std::max(col0, cc[b]), cb = std::min(col0 + w.n, cc[b + 1]);
                            int r_lo = ra - row0, c_lo = ca - col0;
                            GSeg g{};
                            if (cur[q].flip) { // V[window] += W'(m1 x k0) . op(Y)(k0 x n)
                                g.a_src = 2, g.a_off = src_off + (uint64_t)r_lo * p.k0, g.a_sr = p.k0, g.a_sk = 1;
                                g.b_src = ysrc[w.pair];
                                if (p.tb0)
                                    g.b_off = p.y_off + (uint64_t)c_lo * p.ldb0, g.b_sk = 1, g.b_sc = p.ldb0;
                                else
                                    g.b_off = p.y_off + (uint64_t)c_lo, g.b_sk = p.ldb0, g.b_sc = 1;
                                g.K = p.k0, g.alpha = msum ? 1.0 : p.alpha0 * p.alpha1;
                            } else {
                            g.a_src = zsrc[w.pair];
                            if (p.ta1)
                                g.a_off = p.z_off + (uint64_t)r_lo, g.a_sr = 1, g.a_sk = p.lda1;
                            else
                                g.a_off = p.z_off + (uint64_t)r_lo * p.lda1, g.a_sr = p.lda1, g.a_sk = 1;
                            g.b_src = 2, g.b_off = src_off + (uint64_t)c_lo, g.b_sk = p.n0, g.b_sc = 1;
                            g.K = p.k1, g.alpha = msum ? 1.0 : p.alpha0 * p.alpha1;
                            }
                            g.mr = rb - ra, g.nc = cb - ca;
                            g.tc0 = ca - cc[b]; // (ra == rc[a]: rows are cut at every window boundary)
                            size_t t = tbase[a] + b;
                            fsegs.push_back(g), ftile.push_back((uint32_t)t), tstart[t + 1]++;
                            tcost[t] += (double)round_up(g.mr, 16) * round_up(g.nc, 16) * round_up(g.K, 16) + 65536.0;
                            gg_macs += (uint64_t)g.mr * g.nc * g.K;
                        }
                    }
                }
                pc.lap("6c segments");
                const size_t seg_base = out.gsegs.size();
                {
                    for (size_t t = 0; t < tbase[nrt]; t++)
                        tstart[t + 1] += tstart[t];
                    tvec<uint32_t> pos(tstart.begin(), tstart.end() - 1, mem);
                    out.gsegs.resize(seg_base + fsegs.size());
                    for (size_t k = 0; k < fsegs.size(); k++)
                        out.gsegs[seg_base + pos[ftile[k]]++] = fsegs[k];
                }
                for (int a = 0; a < nrt; a++) {
                    const std::vector<int> &cc = *ccs[a];
                    const int nct = (int)cc.size() - 1;
                    for (int b = 0; b < nct; b++) {
                        size_t t = tbase[a] + b;
                        if (tstart[t + 1] == tstart[t])
                            continue;
                        DTile dt{};
                        dt.sigma_off = c.base + (uint64_t)rc[a] * c.ld + cc[b];
                        dt.ld = c.ld, dt.rows = rc[a + 1] - rc[a], dt.cols = cc[b + 1] - cc[b];
                        dt.slab_off = slab;
                        // items of ~equal cost; target: a few thousand items per super-step
                        double per_item = step_item_cost;
                        int n_it = std::max(1, (int)std::lround(tcost[t] / per_item));
                        double per = tcost[t] / n_it, acc = 0;
                        int made = 0;
                        uint32_t begin = (uint32_t)(seg_base + tstart[t]);
                        const uint32_t t_end = (uint32_t)(seg_base + tstart[t + 1]);
                        for (uint32_t k = begin; k < t_end; k++) {
                            const GSeg &g = out.gsegs[k];
                            acc += (double)round_up(g.mr, 16) * round_up(g.nc, 16) * round_up(g.K, 16) + 65536.0;
                            bool last = k + 1 == t_end;
                            if (last || (acc >= per * (made + 1) && made + 1 < n_it)) {
                                GItem it{};
                                it.seg_begin = begin, it.seg_end = k + 1;
                                it.out_off = slab, it.out_ld = dt.cols, it.rows = dt.rows, it.cols = dt.cols;
                                it.alpha = 1.0, it.out_kind = 0;
                                out.gitems.push_back(it);
                                tile_of_item.push_back(tile_seq);
                                slab += (uint64_t)dt.rows * dt.cols;
                                begin = it.seg_end;
                                made++;
                            }
                        }
                        dt.n_items = made;
                        out.gtiles.push_back(dt);
                        // sibling order: 4 x 4 blocks of tiles (a tile shares its A operands with the tiles of its row and
                        // its B operands with those of its column: a square block shares both)
                        static const int sb = getenv("B2X_XCD_BLK") ? atoi(getenv("B2X_XCD_BLK")) : 4; // (probe: block edge)
                        const uint32_t blk = (uint32_t)((a / sb) * ceil_div(nct_max, sb) + b / sb);
                        const uint32_t key = (tile_sector << 20) | ((blk & 0x3FFFu) << 6) | (uint32_t)((a % sb) * sb + b % sb);
                        for (int m2 = 0; m2 < made; m2++)
                            tile_of_item[tile_of_item.size() - 1 - m2] = key;
                        tile_seq++;
                    }
                }
                pc.lap("6d items");
                tile_sector++;
                i = j;
            }
            pc.lap("6 stage-1 items");
            const uint32_t s1_end = (uint32_t)out.gitems.size();
            ss.tile_end = (uint32_t)out.gtiles.size();
            // group by tile-height variant; inside a variant, longest items first
            auto variant = [](const GItem &x) { return (x.rows - 1) / kGGRowUnit; }; // row fragments - 1
            // one grid per stage holds the items of all tile heights: longest (by MFMA issue slots) first
            auto icost = [&](const GItem &x) {
                uint64_t k = 0;
                for (uint32_t q = x.seg_begin; q < x.seg_end; q++)
                    k += (uint64_t)round_up(out.gsegs[q].K, 16);
                return k * (uint64_t)(variant(x) + 1);
            };
            // tall tiles first, short ones (<= kGGShortFrags row fragments) behind them: each class is one launch of the
            // kernel instantiation that serves it (launch_gg); the cost order holds inside a class
            // The short class gets its own launch (on the plan's auxiliary stream, beside the tall one: launch_stage in
            // b2x_capi.cpp) where it carries at least 30 % of the stage's MFMA issue slots: M=250 -20 % time, M=500 -5 %,
            // the uniform-bond-dimension Cr2 plan at M=1000 -11 %; below that share the few short tiles ride along in the
            // tall launch (split regardless: M=1000 +1.5 %).  B2X_SPLIT_THR overrides (2 = never split).
            auto short_share = [&](uint32_t b, uint32_t e) {
                double tot = 0, sh = 0;
                for (uint32_t ii = b; ii < e; ii++) {
                    const double c = (double)icost(out.gitems[ii]);
                    tot += c;
                    if (out.gitems[ii].rows <= short_rows)
                        sh += c;
                }
                return tot > 0 ? sh / tot : 0.0;
            };
            static const double thr = getenv("B2X_SPLIT_THR") ? atof(getenv("B2X_SPLIT_THR")) : 0.3;
            // (the 1-wave workgroups are a launch of their own whenever a stage has short tiles at all)
            const double thr_eff = use_narrow ? 1e-30 : thr;
            const bool split0 = short_share(s0_begin, s1_begin) >= thr_eff, split1 = short_share(s1_begin, s1_end) >= thr_eff;
            auto tall = [short_rows](const GItem &x) { return x.rows > short_rows; };
            // Stage 0: one workgroup per W tile.  Workgroup ids i, i + 8, i + 16, ... run on one XCD (one L2, whose lines
            // live a few microseconds under this traffic): the items of a locality group (s0_key) are handed to ONE XCD as
            // consecutive workgroups, so that all but the first find their B tile / A rows in that L2.  Groups go
            // longest first to the XCD queue with the least work so far (tail balance as before); B2X_S0_LOCAL=0 restores
            // the plain longest-first order.
            uint32_t s0_mid = s1_begin;
            {
                const uint32_t n0 = s1_begin - s0_begin;
                static const int s0_local = getenv("B2X_S0_LOCAL") ? atoi(getenv("B2X_S0_LOCAL")) : 1;
                struct IK {
                    uint64_t cost, grp;
                    uint32_t idx;
                    bool tall;
                };
                tvec<IK> ik(n0, mem);
                for (uint32_t q = 0; q < n0; q++) {
                    const GItem &it = out.gitems[s0_begin + q];
                    ik[q] = IK{icost(it), s0_key[q], s0_begin + q, !split0 || tall(it)};
                }
                tvec<GItem> sorted(mem);
                sorted.reserve(n0);
                for (int cl = 0; cl < 2; cl++) { // tall class, then the short one (its own launch when split0)
                    tvec<IK> v(mem);
                    for (const IK &k : ik)
                        if (k.tall == (cl == 0))
                            v.push_back(k);
                    if (cl == 0)
                        s0_mid = s0_begin + (uint32_t)v.size();
                    if (!s0_local || v.size() < 64) {
                        std::sort(v.begin(), v.end(), [](const IK &x, const IK &y) { return x.cost != y.cost ? x.cost > y.cost : x.idx < y.idx; });
                        for (const IK &k : v)
                            sorted.push_back(out.gitems[k.idx]);
                        continue;
                    }
                    // groups: members together (longest first inside), groups by their longest member
                    std::sort(v.begin(), v.end(), [](const IK &x, const IK &y) {
                        return x.grp != y.grp ? x.grp < y.grp : (x.cost != y.cost ? x.cost > y.cost : x.idx < y.idx);
                    });
                    struct Grp {
                        uint64_t maxc, sum;
                        uint32_t b, e;
                    };
                    tvec<Grp> gs(mem);
                    for (uint32_t a = 0; a < v.size();) {
                        uint32_t b = a;
                        uint64_t sum = 0;
                        while (b < v.size() && v[b].grp == v[a].grp)
                            sum += v[b].cost + 64, b++; // (+64: a workgroup's fixed cost in the same MFMA-slot unit)
                        gs.push_back(Grp{v[a].cost, sum, a, b});
                        a = b;
                    }
                    std::sort(gs.begin(), gs.end(), [](const Grp &x, const Grp &y) { return x.maxc != y.maxc ? x.maxc > y.maxc : x.b < y.b; });
                    const size_t N = v.size();
                    tvec<tvec<uint32_t>> qu(8, mem);
                    size_t cap[8];
                    uint64_t load[8] = {};
                    for (size_t x = 0; x < 8; x++)
                        cap[x] = (N - x + 7) / 8, qu[x].reserve(cap[x]);
                    for (const Grp &g : gs) {
                        uint32_t a = g.b;
                        while (a < g.e) { // least-loaded queue with room; a group larger than the room is split
                            int best = -1;
                            for (int x = 0; x < 8; x++)
                                if (qu[x].size() < cap[x] && (best < 0 || load[x] < load[best]))
                                    best = x;
                            const uint32_t take = (uint32_t)std::min<size_t>(g.e - a, cap[best] - qu[best].size());
                            for (uint32_t q = a; q < a + take; q++)
                                qu[best].push_back(v[q].idx), load[best] += v[q].cost + 64;
                            a += take;
                        }
                    }
                    for (size_t i = 0; i < N; i++)
                        sorted.push_back(out.gitems[qu[i % 8][i / 8]]);
                }
                std::copy(sorted.begin(), sorted.end(), out.gitems.begin() + s0_begin);
            }
            pc.lap("7 stage-0 order");
            uint32_t s1_mid = s1_begin;
            // Stage 1: longest first ACROSS sibling groups, siblings together inside a group.  Items of equal cost that
            // take the same K range (item index j) of neighbouring tiles of one sector walk the same operands: the tiles of
            // a row share their A blocks, the tiles of a column their B blocks.  Workgroup ids i, i + 8, i + 16, ... run on
            // one XCD (one L2), so after the cost order every window of 8 g items is transposed: XCD x receives g
            // consecutive siblings instead of every eighth one.  The longest-first order is kept to within one window.
            {
                struct Key {
                    uint64_t cost;
                    uint32_t j, seq, idx;
                };
                tvec<Key> keys(mem);
                keys.reserve(s1_end - s1_begin);
                uint32_t seq = 0, j = 0;
                for (uint32_t ii = s1_begin; ii < s1_end; ii++) { // (creation order: tile by tile, items of a tile in a row)
                    const GItem &it = out.gitems[ii];
                    const bool same_tile = ii > s1_begin && tile_of_item[ii - s1_begin] == tile_of_item[ii - 1 - s1_begin];
                    j = same_tile ? j + 1 : 0;
                    seq = tile_of_item[ii - s1_begin];
                    keys.push_back(Key{icost(it), j, seq, ii});
                }
                std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) {
                    if (x.cost != y.cost)
                        return x.cost > y.cost;
                    if (x.j != y.j)
                        return x.j < y.j;
                    return x.seq != y.seq ? x.seq < y.seq : x.idx < y.idx;
                });
                // g = 16 siblings per XCD and window (measured on the M=4000 plan: FETCH_SIZE -12 %, time -1 %; g = 4 ... 64
                // within 2 % of each other; B2X_XCD_G overrides, 0 = plain cost order)
                static const int xg = getenv("B2X_XCD_G") ? atoi(getenv("B2X_XCD_G")) : 16;
                tvec<GItem> sorted(mem);
                sorted.reserve(keys.size());
                for (const Key &k : keys)
                    sorted.push_back(out.gitems[k.idx]);
                const size_t n_tall = !split1 ? sorted.size() : (size_t)(std::stable_partition(sorted.begin(), sorted.end(), tall) - sorted.begin());
                s1_mid = s1_begin + (uint32_t)n_tall;
                if (xg > 1) {
                    const size_t W = (size_t)8 * xg;
                    tvec<GItem> tmp(W, mem);
                    const size_t lim[3] = {0, n_tall, sorted.size()};
                    for (int cl = 0; cl < 2; cl++) // (a window never straddles the two launches)
                        for (size_t b = lim[cl]; b + W <= lim[cl + 1]; b += W) {
                            for (size_t x = 0; x < 8; x++)
                                for (size_t k = 0; k < (size_t)xg; k++)
                                    tmp[8 * k + x] = sorted[b + x * xg + k];
                            std::copy(tmp.begin(), tmp.end(), sorted.begin() + b);
                        }
                }
                std::copy(sorted.begin(), sorted.end(), out.gitems.begin() + s1_begin);
            }
            pc.lap("8 stage-1 order");
            auto fill = [&](uint32_t b, uint32_t mid, uint32_t e, uint32_t *v) { // [v[0], v[1]) tall, [v[1], v[last]) short tiles
                v[0] = b, v[1] = mid;
                for (int k = 2; k <= kGGVariants; k++)
                    v[k] = e;
            };
            fill(s0_begin, s0_mid, s1_begin, ss.s0_v);
            fill(s1_begin, s1_mid, s1_end, ss.s1_v);
            for (uint32_t ii = s0_begin; ii < s1_end; ii++) {
                const GItem &it = out.gitems[ii];
                uint64_t tm = (uint64_t)kGGRowUnit * (uint64_t)(variant(it) + 1);
                for (uint32_t k = it.seg_begin; k < it.seg_end; k++)
                    st.macs_issued += tm * (uint64_t)issued_cols(it.cols) * (uint64_t)issued_k(out.gsegs[k].K);
            }
            ss.sum_end = (uint32_t)out.sum_work.size();
            pc.lap("9 issue count");
            if (pc.on) { // work-list shape of this super-step, per launch
                const uint32_t lim[5] = {s0_begin, s0_mid, s1_begin, s1_mid, s1_end};
                static const char *nm[4] = {"s0 tall", "s0 short", "s1 tall", "s1 short"};
                for (int l = 0; l < 4; l++) {
                    uint64_t ni = lim[l + 1] - lim[l], ns = 0, nch = 0, frag_ch = 0, rows = 0, cols = 0;
                    for (uint32_t ii = lim[l]; ii < lim[l + 1]; ii++) {
                        const GItem &it = out.gitems[ii];
                        rows += it.rows, cols += it.cols;
                        for (uint32_t k = it.seg_begin; k < it.seg_end; k++)
                            ns++, nch += ceil_div(out.gsegs[k].K, 16), frag_ch += (uint64_t)(variant(it) + 1) * ceil_div(out.gsegs[k].K, 16);
                    }
                    if (ni)
                        fprintf(stderr, "[b2x plan] %-8s items %8llu  segs/item %6.2f  chunks/seg %5.2f  row frags/chunk %4.2f  rows %5.1f cols %5.1f\n",
                                nm[l], (unsigned long long)ni, (double)ns / ni, (double)nch / ns, (double)frag_ch / nch, (double)rows / ni, (double)cols / ni);
                }
            }
            out.steps.push_back(ss);
            out.scratch_elems = std::max(out.scratch_elems, used + sum_extra);
            out.gslab_elems = std::max(out.gslab_elems, slab);
            cur.clear();
            used = aux_len;
        };
        struct Cand {
            const Component *c;
            uint32_t wi;
            bool flip;
            uint64_t key[7]; // identifies the stage-0 product
        };
        auto make_key = [&](const b2x_pair &p, bool fl, uint64_t *key) {
            if (fl) // W' = op(Z) X: left operator block + psi slice
                key[0] = 1, key[1] = p.x_off, key[2] = p.z_off, key[3] = ((uint64_t)p.m1 << 32) | (uint32_t)p.k1,
                key[4] = ((uint64_t)p.k0 << 32) | (uint32_t)p.lda0, key[5] = ((uint64_t)p.lda1 << 8) | p.ta1, key[6] = 0;
            else // W = X op(Y): psi slice + right operator block
                key[0] = 0, key[1] = p.x_off, key[2] = p.y_off, key[3] = ((uint64_t)p.k1 << 32) | (uint32_t)p.k0,
                key[4] = ((uint64_t)p.n0 << 32) | (uint32_t)p.lda0, key[5] = ((uint64_t)p.ldb0 << 8) | p.tb0, key[6] = 0;
        };
        tvec<Cand> cand(mem);
        {
            size_t nc = 0;
            for (const Component *c : big)
                nc += c->w_end - c->w_begin;
            cand.reserve(nc);
        }
        for (const Component *c : big)
            for (uint32_t wi = c->w_begin; wi < c->w_end; wi++) {
                Cand cd{};
                cd.c = c, cd.wi = wi, cd.flip = false;
                cand.push_back(cd);
            }
        pc.lap("4 cand");
        if (allow_flip) {
            // order of each pair: stage-0 cost amortised over the pairs that would share the product + its own stage 1
            // (the keys of both orders, in candidate order: the passes below read them in sequence and confirm a table hit with
            // ONE access, not through candidate -> window -> pair)
            tvec<std::array<uint64_t, 7>> keys[2] = {tvec<std::array<uint64_t, 7>>(mem), tvec<std::array<uint64_t, 7>>(mem)};
            for (int fl = 0; fl < 2; fl++) {
                keys[fl].resize(cand.size());
                for (size_t q = 0; q < cand.size(); q++)
                    make_key(ep[win[cand[q].wi].pair], fl != 0, keys[fl][q].data());
            }
            auto group_sizes = [&](bool fl) { // size of every candidate's sharing group: hash table on the 7-word key
                const tvec<std::array<uint64_t, 7>> &kk = keys[fl ? 1 : 0];
                struct Slot { // 16 bytes: the table of a 10^5-pair plan stays in the last-level cache
                    uint64_t h;
                    uint32_t count, first; // first = candidate that opened the slot (+1; 0 = empty)
                };
                size_t cap = 64;
                while (cap < 2 * cand.size())
                    cap <<= 1;
                tvec<Slot> tab(cap, Slot{0, 0, 0}, mem);
                tvec<uint32_t> slot_of(cand.size(), mem);
                for (size_t q = 0; q < cand.size(); q++) {
                    const uint64_t *key = kk[q].data();
                    uint64_t h = 0x9E3779B97F4A7C15ull;
                    for (int k = 0; k < 7; k++)
                        h = (h ^ key[k]) * 0xBF58476D1CE4E5B9ull, h ^= h >> 29;
                    size_t i = (size_t)h & (cap - 1);
                    while (tab[i].first != 0) {
                        if (tab[i].h == h && kk[tab[i].first - 1] == kk[q]) // (confirmed on the full key of the slot's first member)
                            break;
                        i = (i + 1) & (cap - 1);
                    }
                    if (tab[i].first == 0)
                        tab[i].h = h, tab[i].first = (uint32_t)q + 1;
                    tab[i].count++, slot_of[q] = (uint32_t)i;
                }
                tvec<uint32_t> sz(cand.size(), mem);
                for (size_t q = 0; q < cand.size(); q++)
                    sz[q] = tab[slot_of[q]].count;
                return sz;
            };
            const tvec<uint32_t> g0 = group_sizes(false), g1 = group_sizes(true);
            for (size_t q = 0; q < cand.size(); q++) {
                const b2x_pair &p = ep[win[cand[q].wi].pair];
                const double cur_c = (double)p.k1 * p.k0 * p.n0 / g0[q] + (double)p.m1 * p.k1 * p.n0;
                const double alt_c = (double)p.m1 * p.k1 * p.k0 / g1[q] + (double)p.m1 * p.k0 * p.n0;
                cand[q].flip = alt_c < 0.95 * cur_c;
                std::copy(keys[cand[q].flip ? 1 : 0][q].begin(), keys[cand[q].flip ? 1 : 0][q].end(), cand[q].key);
            }
        } else
            for (Cand &cd : cand)
                make_key(ep[win[cd.wi].pair], cd.flip, cd.key);
        pc.lap("4 group sizes");
        if (allow_flip && opt && opt->presum == 1) {
            // Operator pre-sums (off by default: +2 % at M=4000, +8 % at M=500, -2 % at M=1000 on the Cr2 plan, for 36 GB
            // of plan-owned memory at M=4000; the per-step sums of W products above catch most of the same redundancy).  Pairs with the SAME stage-0 product and the SAME psi' window differ only in their second
            // operator: op(Z_i) for W = X op(Y) (same X, Y), op(Y_i) for W' = op(Z) X (same Z, X).  Their sum
            // S = sum_i alpha_i op(.)_i depends on the operators alone, so it is formed ONCE, when the plan is created
            // (element-wise kernel into the persistent head of the plan's scratch), and the group becomes one pair.
            struct PK {
                uint64_t k[10];
                uint32_t q;
            };
            tvec<PK> pk(cand.size(), mem);
            for (size_t q = 0; q < cand.size(); q++) {
                const Window &w = win[cand[q].wi];
                const b2x_pair &p = ep[w.pair];
                std::copy(cand[q].key, cand[q].key + 7, pk[q].k);
                pk[q].k[7] = w.off, pk[q].k[8] = ((uint64_t)p.m1 << 32) | (uint32_t)p.n0, pk[q].k[9] = (uint64_t)p.ldc1;
                pk[q].q = (uint32_t)q;
            }
            std::stable_sort(pk.begin(), pk.end(), [](const PK &x, const PK &y) {
                for (int k = 0; k < 10; k++)
                    if (x.k[k] != y.k[k])
                        return x.k[k] < y.k[k];
                return false;
            });
            tvec<uint8_t> dead(cand.size(), 0, mem);
            for (size_t a2 = 0; a2 < pk.size();) {
                size_t b2 = a2 + 1;
                while (b2 < pk.size() && std::equal(pk[a2].k, pk[a2].k + 10, pk[b2].k))
                    b2++;
                if (b2 - a2 > 1) {
                    const uint32_t lead = win[cand[pk[a2].q].wi].pair;
                    const bool fl = cand[pk[a2].q].flip;
                    const b2x_pair p0 = ep[lead];
                    const int srows = fl ? p0.k0 : p0.m1, scols = fl ? p0.n0 : p0.k1; // op(Y) is k0 x n0, op(Z) is m1 x k1
                    const uint64_t s_off = aux_len;
                    aux_len += slot_elems(out, s_off, (uint64_t)srows * scols);
                    const uint32_t eb = (uint32_t)out.aux_entries.size();
                    for (size_t x = a2; x < b2; x++) {
                        const b2x_pair &px = ep[win[cand[pk[x].q].wi].pair];
                        OEntry e{};
                        const bool tr = fl ? px.tb0 : px.ta1;
                        const int ld = fl ? px.ldb0 : px.lda1;
                        e.a_off = fl ? px.y_off : px.z_off, e.alpha = px.alpha0 * px.alpha1;
                        e.a_rs = tr ? 1 : ld, e.a_cs = tr ? ld : 1, e.a_src = 0, e.b_src = 2;
                        out.aux_entries.push_back(e);
                        if (x > a2)
                            dead[pk[x].q] = 1;
                    }
                    OWork wk{};
                    wk.out_off = s_off, wk.ld = -scols, wk.rows = srows, wk.cols = scols, wk.rpt = 16;
                    wk.entry_begin = eb, wk.entry_end = (uint32_t)out.aux_entries.size();
                    const uint32_t ntile = (uint32_t)ceil_div(scols, kOuterTileCols) * (uint32_t)ceil_div(srows, 16);
                    for (uint32_t t0 = 0; t0 < ntile; t0++) {
                        wk.t_begin = t0, wk.t_end = t0 + 1;
                        out.aux_work.push_back(wk);
                    }
                    b2x_pair &pm = ep[lead];
                    pm.alpha0 = 1.0, pm.alpha1 = 1.0;
                    if (fl)
                        pm.y_off = s_off, pm.tb0 = 0, pm.ldb0 = scols, ysrc[lead] = 2;
                    else
                        pm.z_off = s_off, pm.ta1 = 0, pm.lda1 = scols, zsrc[lead] = 2;
                }
                a2 = b2;
            }
            if (aux_len) {
                tvec<Cand> keep(mem);
                for (size_t q = 0; q < cand.size(); q++)
                    if (!dead[q])
                        keep.push_back(cand[q]);
                cand.swap(keep);
                used = aux_len;
            }
        }
        pc.lap("4 keys+presum");
        auto key_less = [](const Cand &x, const Cand &y) {
            for (int k = 0; k < 7; k++)
                if (x.key[k] != y.key[k])
                    return x.key[k] < y.key[k];
            return false;
        };
        if (allow_flip) { // (keep_order = 1 replays the reference pair by pair: sector order, no sharing)
            // the order of std::stable_sort(cand, key_less), from a sort of 32-byte records (leading key words + position)
            struct SK {
                uint64_t a, b, c;
                uint32_t idx;
            };
            tvec<SK> sk(cand.size(), mem);
            for (size_t q = 0; q < cand.size(); q++)
                sk[q] = SK{cand[q].key[0], cand[q].key[1], cand[q].key[2], (uint32_t)q};
            std::sort(sk.begin(), sk.end(), [&](const SK &x, const SK &y) {
                if (x.a != y.a)
                    return x.a < y.a;
                if (x.b != y.b)
                    return x.b < y.b;
                if (x.c != y.c)
                    return x.c < y.c;
                if (key_less(cand[x.idx], cand[y.idx]))
                    return true;
                if (key_less(cand[y.idx], cand[x.idx]))
                    return false;
                return x.idx < y.idx;
            });
            tvec<Cand> sorted(cand.size(), mem);
            for (size_t q = 0; q < cand.size(); q++)
                sorted[q] = cand[sk[q].idx];
            cand.swap(sorted);
        }
        pc.lap("4 pair order + keys");
        uint64_t last_off = 0;
        for (size_t q = 0; q < cand.size(); q++) {
            const Cand &cd = cand[q];
            const b2x_pair &p = ep[win[cd.wi].pair];
            const bool same = allow_flip && q > 0 && !cur.empty() && !key_less(cand[q - 1], cd) && !key_less(cd, cand[q - 1]);
            out.stats.n_flipped += cd.flip ? 1 : 0;
            if (same) { // shares the product of the previous pair (still in this super-step)
                out.stats.n_shared_products++;
                cur.push_back(PW{cd.c, cd.wi, last_off, cd.flip, false});
                continue;
            }
            const uint64_t wexact = cd.flip ? (uint64_t)p.m1 * p.k0 : (uint64_t)p.k1 * p.n0;
            if (used - aux_len + wexact + 16 > budget && !cur.empty()) // (the budget bounds the per-step part of the scratch)
                flush();
            const uint64_t wsz = slot_elems(out, used, wexact);
            cur.push_back(PW{cd.c, cd.wi, used, cd.flip, true});
            last_off = used;
            used += wsz;
        }
        flush();
        st.n_tiles += out.gtiles.size();
        st.n_items += out.gitems.size();
        st.n_parts += out.gsegs.size();
    }
    for (int k = 0; k < kNumClasses; k++) {
        out.cls_macs[k] = cls_macs[k];
        if (cls_macs[k] > cls_macs[st.dominant_class])
            st.dominant_class = k;
    }
    out.seg_scaled = true; // stage-1 segments carry their pair's alpha (shared stage-0 products are stored unscaled)
    st.macs_executed = cls_macs[0] + cls_macs[1] + cls_macs[2] + gg_macs;
    st.macs_dominant = cls_macs[st.dominant_class];
    st.macs_alg_dominant = (uint64_t)(cls_alg[st.dominant_class] + 0.5);
    st.n_launches = 1;
    if (gg_macs > st.macs_dominant) { // the two-stage grouped-GEMM kernel carries the plan
        st.dominant_class = kNumClasses, st.macs_dominant = gg_macs;
        st.macs_alg_dominant = (uint64_t)(gg_macs_total + 0.5); // reference count of the pairs on this path
        st.n_launches = 0;
        for (const SuperStep &ss : out.steps)
            st.n_launches += (ss.s0_v[1] > ss.s0_v[0]) + (ss.s0_v[kGGVariants] > ss.s0_v[1]) + (ss.s1_v[1] > ss.s1_v[0]) +
                             (ss.s1_v[kGGVariants] > ss.s1_v[1]);
    }
    stage_residual_reads(out, std::max(arena_cap, arena_len), psi_len);
    st.device_bytes = (out.scratch_elems + out.gslab_elems) * 8 + out.gsegs.size() * sizeof(GSeg) +
                      out.gitems.size() * sizeof(GItem) + out.gtiles.size() * sizeof(DTile) + slab * 8 + st.n_parts * sizeof(DPart) + st.n_items * sizeof(DItem) + st.n_tiles * sizeof(DTile);
    if (pc.on) {
        fprintf(stderr, "[b2x plan] segments %zu (capacity %zu), items %zu (capacity %zu), sum work %zu entries %zu, pads %zu\n", out.gsegs.size(), out.gsegs.capacity(), out.gitems.size(), out.gitems.capacity(), out.sum_work.size(), out.sum_entries.size(), out.scratch_pads.size());
        fprintf(stderr, "[b2x plan] scratch arena of the compiler: %.0f MB mapped\n", (double)mem_arena.capacity() / 1048576.0);
    }
    pc.lap("a closing");
    pc.report();
    return B2X_OK;
}

int compile_gemm_list(size_t n_gemms, const b2x_gemm *gemms, size_t in_len, size_t out_len, uint64_t arena_len,
                      uint64_t arena_cap, const b2x_plan_options *opt, CompiledPlan &out, std::string &err) {
    out = CompiledPlan();
    out.seg_scaled = true;
    b2x_plan_stats &st = out.stats;
    st.n_pairs = n_gemms, st.psi_len = in_len, st.sigma_len = out_len;
    static thread_local ScratchArena mem_arena; // (temporaries, see compile_plan)
    ScratchArena *const mem = &mem_arena;
    ArenaScope mem_scope{mem_arena};
    std::vector<std::pair<uint64_t, uint64_t>> opext;
    for (size_t i = 0; i < n_gemms; i++) {
        const b2x_gemm &g = gemms[i];
        auto id = [i]() { return "gemm " + std::to_string(i); }; // (built on error only: 10^5 records pass through here)
        if (g.ta > 1 || g.tb > 1 || g.a_src > 1 || g.b_src > 1) {
            err = id() + ": unsupported transpose / source flags";
            return B2X_ERR_INVALID;
        }
        if (g.m <= 0 || g.n <= 0 || g.k <= 0) {
            err = id() + ": dimensions must be positive";
            return B2X_ERR_INVALID;
        }
        if (g.lda < (g.ta ? g.m : g.k) || g.ldb < (g.tb ? g.k : g.n) || g.ldc < g.n) {
            err = id() + ": leading dimension smaller than row length";
            return B2X_ERR_INVALID;
        }
        if (std::max(g.lda, std::max(g.ldb, g.ldc)) >= kMaxLeadingDim) { // (see compile_plan)
            err = id() + ": leading dimension >= 2^22 elements is not supported";
            return B2X_ERR_INVALID;
        }
        uint64_t ea = g.ta ? (uint64_t)(g.k - 1) * g.lda + g.m : (uint64_t)(g.m - 1) * g.lda + g.k;
        uint64_t eb = g.tb ? (uint64_t)(g.n - 1) * g.ldb + g.k : (uint64_t)(g.k - 1) * g.ldb + g.n;
        uint64_t ec = (uint64_t)(g.m - 1) * g.ldc + g.n;
        if (g.a_off + ea > (g.a_src ? (uint64_t)in_len : arena_len) || g.b_off + eb > (g.b_src ? (uint64_t)in_len : arena_len) ||
            g.c_off + ec > out_len) {
            err = id() + ": operand runs past the end of the input / output vector or the arena";
            return B2X_ERR_INVALID;
        }
        st.macs += (uint64_t)g.m * g.n * g.k;
        if (!g.a_src)
            opext.emplace_back(g.a_off, ea);
        if (!g.b_src)
            opext.emplace_back(g.b_off, eb);
    }
    { // (distinct (offset, extent) first, as in compile_plan)
        size_t cap = 1024;
        while (cap < 2 * opext.size())
            cap <<= 1;
        tvec<std::pair<uint64_t, uint64_t>> set(cap, std::make_pair(~(uint64_t)0, (uint64_t)0), mem);
        std::vector<std::pair<uint64_t, uint64_t>> uniq;
        for (const auto &e : opext) {
            size_t i = (size_t)((e.first * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (set[i].first != ~(uint64_t)0 && set[i] != e)
                i = (i + 1) & (cap - 1);
            if (set[i].first == ~(uint64_t)0)
                set[i] = e, uniq.push_back(e);
        }
        opext.swap(uniq);
    }
    std::sort(opext.begin(), opext.end());
    {
        uint64_t cur_b = 0, cur_e = 0;
        for (auto &e : opext) {
            if (e.first >= cur_e) {
                st.op_elems_unique += cur_e - cur_b;
                cur_b = e.first, cur_e = e.first + e.second;
            } else
                cur_e = std::max(cur_e, e.first + e.second);
        }
        st.op_elems_unique += cur_e - cur_b;
    }
    if (n_gemms == 0)
        return B2X_OK;
    // ---- distributive law ------------------------------------------------------------------------------------------
    // Records that multiply the SAME block of the input vector into the SAME output window,
    //     C += sum_i alpha_i op(A_i) . X      or      C += X . sum_i alpha_i op(B_i),
    // (the reduced perturbative noise applies every left / right operator of the Hamiltonian to the same psi blocks:
    // on the Cr2 M=250 list 155 728 records fall into 1 400 such groups) first sum their operator blocks,
    // S = sum_i alpha_i op(A_i), in an element-wise pass (outer_build_k into the scratch; transposed and plain members form
    // separate groups), then take ONE product.  Worth it where the product saved per member outweighs reading its operator
    // block once more: (g - 1)/(g + 1) x (columns of X, or rows of X) > 8.  keep_order = 1 replays record by record.
    std::vector<b2x_gemm> eff_store;
    const b2x_gemm *eff = gemms;
    size_t n_eff = n_gemms;
    if (!(opt && opt->keep_order == 1)) {
        struct MK {
            uint64_t k[7];
            uint32_t i;
        };
        std::vector<MK> mk;
        std::vector<uint8_t> taken(n_gemms, 0);
        for (size_t i = 0; i < n_gemms; i++) {
            const b2x_gemm &g = gemms[i];
            const bool right = g.a_src == 0 && g.b_src == 1, left = g.a_src == 1 && g.b_src == 0;
            if (!right && !left)
                continue;
            MK m{};
            m.i = (uint32_t)i;
            m.k[0] = right, m.k[1] = right ? g.b_off : g.a_off, m.k[2] = g.c_off;
            m.k[3] = ((uint64_t)g.m << 32) | (uint32_t)g.n, m.k[4] = ((uint64_t)g.k << 32) | (uint32_t)g.ldc;
            m.k[5] = right ? (((uint64_t)g.ldb << 8) | g.tb) : (((uint64_t)g.lda << 8) | g.ta);
            m.k[6] = right ? g.ta : g.tb; // members of a group are stored in the same orientation (S is built in it)
            mk.push_back(m);
        }
        group_sort<7>(mk, mem);
        uint64_t s_used = 0;
        std::vector<b2x_gemm> merged;
        // Sums already formed, by member set: the same operator blocks meet several psi blocks (one per quantum number of
        // the other index), and their coefficient vectors are proportional (the ratio is the coupling factor of the psi
        // block).  Such groups share ONE sum, S = sum_i alpha_i A_i, and carry the ratio in their product's alpha:
        // on the Cr2 M=250 noise list 1 400 groups need 448 sums, 0.50 GB of operator reads instead of 1.81 GB.
        struct Formed {
            std::vector<double> alpha;
            uint64_t s_off;
        };
        std::map<std::vector<uint64_t>, std::vector<Formed>> formed;
        for (size_t a = 0; a < mk.size();) {
            size_t b = a + 1;
            while (b < mk.size() && std::equal(mk[a].k, mk[a].k + 7, mk[b].k))
                b++;
            const size_t gsz = b - a;
            const b2x_gemm &g0 = gemms[mk[a].i];
            const bool right = g0.a_src == 0;
            const double dim = right ? (double)g0.n : (double)g0.m;
            if (gsz > 1) {
                // S = sum alpha_i A_i in the members' STORED orientation (the key holds their transposition flag, so a
                // group is homogeneous): every member is read along its contiguous dimension (lane = stored column) —
                // reading transposed members in the output's layout fetched every cache line ~3 times (PMC, r01) — and
                // the one remaining product takes S with the members' flag
                const bool tr0 = right ? g0.ta : g0.tb;
                const int orows = right ? g0.m : g0.k, ocols = right ? g0.k : g0.n; // op(A_i) or op(B_i)
                const int srows = tr0 ? ocols : orows, scols = tr0 ? orows : ocols;
                std::vector<uint32_t> mem; // members by operator offset: the canonical order of a member set
                for (size_t x = a; x < b; x++)
                    mem.push_back(mk[x].i);
                auto op_off = [&](uint32_t i) { return right ? gemms[i].a_off : gemms[i].b_off; };
                auto op_ld = [&](uint32_t i) { return (uint64_t)(right ? gemms[i].lda : gemms[i].ldb); };
                std::stable_sort(mem.begin(), mem.end(), [&](uint32_t x, uint32_t y) { return op_off(x) < op_off(y); });
                std::vector<uint64_t> sig{(uint64_t)srows, (uint64_t)scols};
                std::vector<double> al;
                bool nonzero = true;
                for (uint32_t i : mem) {
                    sig.push_back(op_off(i)), sig.push_back(op_ld(i)), al.push_back(gemms[i].alpha);
                    nonzero = nonzero && gemms[i].alpha != 0.0 && std::isfinite(gemms[i].alpha);
                }
                uint64_t s_off = ~(uint64_t)0;
                double ratio = 1.0;
                std::vector<Formed> &cand = formed[sig];
                if (nonzero)
                    for (const Formed &f : cand) { // alpha = c * f.alpha to a few ulp?
                        const double c = al[0] / f.alpha[0];
                        bool ok = std::isfinite(c) && c != 0.0;
                        for (size_t j = 0; ok && j < al.size(); j++)
                            ok = std::fabs(al[j] - c * f.alpha[j]) <= 4.0 * DBL_EPSILON * std::fabs(al[j]);
                        if (ok) {
                            s_off = f.s_off, ratio = c;
                            break;
                        }
                    }
                if (s_off == ~(uint64_t)0) {
                    if (!(dim * (double)(gsz - 1) / (double)(gsz + 1) > 8.0)) { // not worth a sum of its own
                        a = b;
                        continue;
                    }
                    s_off = s_used;
                    s_used += slot_elems(out, s_off, (uint64_t)srows * scols);
                    const uint32_t eb = (uint32_t)out.sum_entries.size();
                    for (uint32_t i : mem) {
                        OEntry e{};
                        e.a_off = op_off(i), e.alpha = gemms[i].alpha;
                        e.a_rs = (int32_t)op_ld(i), e.a_cs = 1, e.a_src = 0, e.b_src = 2;
                        out.sum_entries.push_back(e);
                    }
                    OWork wk{};
                    wk.out_off = s_off, wk.ld = -scols, wk.rows = srows, wk.cols = scols; // ld < 0: assign
                    wk.rpt = 16; // outer_build_k<16>: sixteen rows in flight per entry visit
                    wk.entry_begin = eb, wk.entry_end = (uint32_t)out.sum_entries.size();
                    const uint32_t nseg = (uint32_t)ceil_div(scols, kOuterTileCols), nstrip = (uint32_t)ceil_div(srows, wk.rpt);
                    const uint32_t ntile = nseg * nstrip;
                    for (uint32_t t0 = 0; t0 < ntile; t0++) { // one tile per wave: the entry lists are long
                        wk.t_begin = t0, wk.t_end = t0 + 1;
                        out.sum_work.push_back(wk);
                    }
                    if (nonzero)
                        cand.push_back(Formed{al, s_off});
                }
                for (uint32_t i : mem)
                    taken[i] = 1;
                b2x_gemm r = g0; // one product with the summed operator (source 2 = scratch, internal)
                r.alpha = ratio;
                if (right)
                    r.a_src = 2, r.a_off = s_off, r.lda = scols, r.ta = tr0;
                else
                    r.b_src = 2, r.b_off = s_off, r.ldb = scols, r.tb = tr0;
                merged.push_back(r);
            }
            a = b;
        }
        if (!merged.empty()) {
            for (size_t i = 0; i < n_gemms; i++)
                if (!taken[i])
                    eff_store.push_back(gemms[i]);
            eff_store.insert(eff_store.end(), merged.begin(), merged.end());
            eff = eff_store.data(), n_eff = eff_store.size();
            out.scratch_elems = s_used;
        }
    }
    std::vector<Window> win(n_eff);
    for (size_t i = 0; i < n_eff; i++)
        win[i] = Window{eff[i].c_off, eff[i].m, eff[i].n, eff[i].ldc, (uint32_t)i};
    bool bad = false;
    std::string reason;
    std::vector<Component> comps = build_components(win, bad, reason);
    if (bad) { // (there is no atomic fallback for single-GEMM lists)
        err = reason;
        return B2X_ERR_INVALID;
    }
    st.n_targets = comps.size();
    double eff_macs = 0;
    for (size_t i = 0; i < n_eff; i++)
        eff_macs += (double)eff[i].m * eff[i].n * eff[i].k;
    double wsum = 0, wn = 0; // tile width by the MAC-weighted mean output width of the records, as in compile_plan
    for (size_t i = 0; i < n_eff; i++) {
        const double w = (double)eff[i].m * eff[i].n * eff[i].k;
        wsum += w, wn += w * eff[i].n;
    }
    const int TN = (wsum > 0 && wn / wsum < 300.0) ? 64 : kGGTileN;
    out.gg_tile_n = TN;
    // item size as in compile_plan: ~8 rounds over the workgroup slots, between 2e6 and 6e7 MFMA-slot units
    const double per_item = opt && opt->item_macs > 0 ? (double)opt->item_macs : std::min(6.0e7, std::max(2.0e6, eff_macs / 4096.0));
    SuperStep ss{};
    uint64_t slab = 0, gg_macs = 0;
    for (const Component &c : comps) {
        std::vector<int> bounds{0, c.rows};
        for (uint32_t wi = c.w_begin; wi < c.w_end; wi++) {
            int row0 = (int)((win[wi].off - c.base) / (uint64_t)c.ld);
            bounds.push_back(row0), bounds.push_back(row0 + win[wi].m);
        }
        std::sort(bounds.begin(), bounds.end());
        bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
        std::vector<int> rc;
        for (size_t bi = 0; bi + 1 < bounds.size(); bi++) {
            std::vector<int> sub = unit_cuts(bounds[bi + 1] - bounds[bi]);
            for (size_t k = 0; k + 1 < sub.size(); k++)
                rc.push_back(bounds[bi] + sub[k]);
        }
        rc.push_back(c.rows);
        std::vector<int> cc = wave_cuts(c.cols, TN);
        const int nrt = (int)rc.size() - 1, nct = (int)cc.size() - 1;
        std::vector<std::vector<GSeg>> tsegs((size_t)nrt * nct);
        std::vector<double> tcost((size_t)nrt * nct, 0.0);
        // plan order inside a sector (the window array is sorted by offset; replay order = record order)
        std::vector<uint32_t> members;
        for (uint32_t wi = c.w_begin; wi < c.w_end; wi++)
            members.push_back(wi);
        std::sort(members.begin(), members.end(), [&](uint32_t a, uint32_t b) { return win[a].pair < win[b].pair; });
        for (uint32_t wi : members) {
            const Window &w = win[wi];
            const b2x_gemm &p = eff[w.pair];
            uint64_t rel = w.off - c.base;
            int row0 = (int)(rel / (uint64_t)c.ld), col0 = (int)(rel % (uint64_t)c.ld);
            int a0 = (int)(std::upper_bound(rc.begin(), rc.end(), row0) - rc.begin()) - 1;
            int b0 = (int)(std::upper_bound(cc.begin(), cc.end(), col0) - cc.begin()) - 1;
            for (int a = a0; a < nrt && rc[a] < row0 + w.m; a++)
                for (int b = b0; b < nct && cc[b] < col0 + w.n; b++) {
                    int ra = std::max(row0, rc[a]), rb = std::min(row0 + w.m, rc[a + 1]);
                    int ca = std::max(col0, cc[b]), cb = std::min(col0 + w.n, cc[b + 1]);
                    int r_lo = ra - row0, c_lo = ca - col0;
                    GSeg g{};
                    g.a_src = p.a_src, g.b_src = p.b_src;
                    if (p.ta) // op(A)[r][k] = A[k][r]
                        g.a_off = p.a_off + (uint64_t)r_lo, g.a_sr = 1, g.a_sk = p.lda;
                    else
                        g.a_off = p.a_off + (uint64_t)r_lo * p.lda, g.a_sr = p.lda, g.a_sk = 1;
                    if (p.tb) // op(B)[k][c] = B[c][k]
                        g.b_off = p.b_off + (uint64_t)c_lo * p.ldb, g.b_sk = 1, g.b_sc = p.ldb;
                    else
                        g.b_off = p.b_off + (uint64_t)c_lo, g.b_sk = p.ldb, g.b_sc = 1;
                    g.K = p.k, g.alpha = p.alpha;
                    g.mr = rb - ra, g.nc = cb - ca, g.tc0 = ca - cc[b];
                    size_t t = (size_t)a * nct + b;
                    tsegs[t].push_back(g);
                    tcost[t] += (double)round_up(g.mr, 16) * round_up(g.nc, 16) * round_up(g.K, 16) + 65536.0;
                    gg_macs += (uint64_t)g.mr * g.nc * g.K;
                }
        }
        for (int a = 0; a < nrt; a++)
            for (int b = 0; b < nct; b++) {
                size_t t = (size_t)a * nct + b;
                if (tsegs[t].empty())
                    continue;
                DTile dt{};
                dt.sigma_off = c.base + (uint64_t)rc[a] * c.ld + cc[b];
                dt.ld = c.ld, dt.rows = rc[a + 1] - rc[a], dt.cols = cc[b + 1] - cc[b];
                dt.slab_off = slab;
                int n_it = std::max(1, (int)std::lround(tcost[t] / per_item));
                double per = tcost[t] / n_it, acc = 0;
                int made = 0;
                uint32_t begin = (uint32_t)out.gsegs.size();
                for (size_t k = 0; k < tsegs[t].size(); k++) {
                    const GSeg &g = tsegs[t][k];
                    out.gsegs.push_back(g);
                    acc += (double)round_up(g.mr, 16) * round_up(g.nc, 16) * round_up(g.K, 16) + 65536.0;
                    bool last = k + 1 == tsegs[t].size();
                    if (last || (acc >= per * (made + 1) && made + 1 < n_it)) {
                        GItem it{};
                        it.seg_begin = begin, it.seg_end = (uint32_t)out.gsegs.size();
                        it.out_off = slab, it.out_ld = dt.cols, it.rows = dt.rows, it.cols = dt.cols;
                        it.alpha = 1.0, it.out_kind = 0;
                        out.gitems.push_back(it);
                        slab += (uint64_t)dt.rows * dt.cols;
                        begin = it.seg_end;
                        made++;
                    }
                }
                dt.n_items = made;
                out.gtiles.push_back(dt);
            }
    }
    auto variant = [](const GItem &x) { return (x.rows - 1) / kGGRowUnit; }; // row fragments - 1
    auto icost = [&](const GItem &x) {
        uint64_t k = 0;
        for (uint32_t q = x.seg_begin; q < x.seg_end; q++)
            k += (uint64_t)round_up(out.gsegs[q].K, 16);
        return k * (uint64_t)(variant(x) + 1);
    };
    std::stable_sort(out.gitems.begin(), out.gitems.end(), [&](const GItem &x, const GItem &y) { return icost(x) > icost(y); });
    double c_tot = 0, c_short = 0; // (same rule as compile_plan: a launch of its own where the short tiles carry enough of the list)
    for (const GItem &x : out.gitems) {
        const double c = (double)icost(x);
        c_tot += c;
        if (x.rows <= kGGShortFrags * kGGRowUnit)
            c_short += c;
    }
    const uint32_t n_tall = !(c_tot > 0 && c_short / c_tot >= 0.3)
                                ? (uint32_t)out.gitems.size()
                                : (uint32_t)(std::stable_partition(out.gitems.begin(), out.gitems.end(), [](const GItem &x) {
                                                 return x.rows > kGGShortFrags * kGGRowUnit;
                                             }) - out.gitems.begin());
    for (int k = 0; k <= kGGVariants; k++)
        ss.s0_v[k] = 0, ss.s1_v[k] = (uint32_t)out.gitems.size();
    ss.s1_v[0] = 0, ss.s1_v[1] = n_tall; // [0, n_tall) tall tiles, [n_tall, end) short ones (launch_gg)
    ss.tile_begin = 0, ss.tile_end = (uint32_t)out.gtiles.size();
    ss.sum_begin = 0, ss.sum_end = (uint32_t)out.sum_work.size();
    for (const GItem &it : out.gitems) {
        uint64_t tm = (uint64_t)kGGRowUnit * (uint64_t)(variant(it) + 1);
        for (uint32_t k = it.seg_begin; k < it.seg_end; k++)
            st.macs_issued += tm * (uint64_t)issued_cols(it.cols) * (uint64_t)issued_k(out.gsegs[k].K);
    }
    out.steps.push_back(ss);
    out.gslab_elems = slab;
    stage_residual_reads(out, std::max(arena_cap, arena_len), in_len);
    st.n_tiles = out.gtiles.size(), st.n_items = out.gitems.size(), st.n_parts = out.gsegs.size();
    st.macs_executed = gg_macs;
    st.dominant_class = kNumClasses, st.macs_dominant = gg_macs, st.macs_alg_dominant = st.macs;
    st.n_launches = (n_tall > 0) + (out.gitems.size() > n_tall);
    st.device_bytes = out.gslab_elems * 8 + out.gsegs.size() * sizeof(GSeg) + out.gitems.size() * sizeof(GItem) +
                      out.gtiles.size() * sizeof(DTile) + out.scratch_elems * 8 + out.sum_work.size() * sizeof(OWork) +
                      out.sum_entries.size() * sizeof(OEntry);
    return B2X_OK;
}

int compile_outer(size_t n_terms, const b2x_outer_term *terms, size_t in_len, size_t out_len, uint64_t arena_len,
                  std::vector<OWork> &work, std::vector<OEntry> &entries, std::string &err) {
    work.clear(), entries.clear();
    PhaseClock pc;
    auto span = [](int64_t m, int64_t n, int64_t rs, int64_t cs, int64_t &lo, int64_t &hi) {
        // offsets touched by r*rs + c*cs, r < m, c < n (strides may be 0, never negative)
        lo = 0, hi = (m - 1) * rs + (n - 1) * cs;
    };
    for (size_t i = 0; i < n_terms; i++) {
        const b2x_outer_term &t = terms[i];
        auto id = [i]() { return "outer term " + std::to_string(i); }; // (built on error only)
        if (t.m <= 0 || t.n <= 0 || t.ldc < t.n || t.a_src > 2 || t.b_src > 2 || t.a_rs < 0 || t.a_cs < 0 || t.b_rs < 0 ||
            t.b_cs < 0) {
            err = id() + ": bad dimensions, strides or source flags";
            return B2X_ERR_INVALID;
        }
        int64_t lo, hi;
        span(t.m, t.n, t.a_rs, t.a_cs, lo, hi);
        if (t.a_src < 2 && t.a_off + (uint64_t)hi >= (t.a_src ? (uint64_t)in_len : arena_len)) {
            err = id() + ": A operand out of range";
            return B2X_ERR_INVALID;
        }
        span(t.m, t.n, t.b_rs, t.b_cs, lo, hi);
        if (t.b_src < 2 && t.b_off + (uint64_t)hi >= (t.b_src ? (uint64_t)in_len : arena_len)) {
            err = id() + ": B operand out of range";
            return B2X_ERR_INVALID;
        }
        if (t.c_off + (uint64_t)(t.m - 1) * t.ldc + t.n > out_len) {
            err = id() + ": output window out of range";
            return B2X_ERR_INVALID;
        }
    }
    if (n_terms == 0)
        return B2X_OK;
    pc.lap("o0 validate");
    std::vector<Window> win(n_terms);
    for (size_t i = 0; i < n_terms; i++)
        win[i] = Window{terms[i].c_off, terms[i].m, terms[i].n, terms[i].ldc, (uint32_t)i};
    bool bad = false;
    std::string reason;
    std::vector<Component> comps = build_components(win, bad, reason);
    if (bad) {
        err = reason;
        return B2X_ERR_INVALID;
    }
    pc.lap("o1 components");
    // cells of a component: the grid cut at every window boundary.  The terms of a cell are gathered by a counting pass into
    // one flat list (a blocking list has ~1e5 windows of a few dozen elements each, three terms per cell: a vector per cell
    // was an allocation per cell and most of this function's time)
    std::vector<int> rb, cb;
    std::vector<uint32_t> members, cell_begin, cell_fill, flat;
    struct Span {
        uint32_t a0, a1, b0, b1;
    };
    std::vector<Span> spans;
    entries.reserve(n_terms);
    work.reserve(n_terms / 2 + 16);
    std::vector<int> wr0, wc0; // row / column of every window's origin inside its component (one division per window)
    auto emit_cell = [&](const Component &c, int r_lo, int r_hi, int c_lo, int c_hi, uint32_t eb, int T) -> bool {
        OWork w{};
        w.out_off = c.base + (uint64_t)r_lo * c.ld + c_lo;
        w.ld = c.ld, w.rows = r_hi - r_lo, w.cols = c_hi - c_lo;
        w.entry_begin = eb, w.entry_end = (uint32_t)entries.size();
        // tiles: 64 columns x rpt rows, rpt sized so that a tile carries a few thousand element-term products;
        // a unit (one wave) takes up to 4 consecutive tiles
        w.rpt = std::max(4, std::min(64, 256 / std::max(1, T)));
        const uint32_t nseg = (uint32_t)ceil_div(w.cols, kOuterTileCols), nstrip = (uint32_t)ceil_div(w.rows, w.rpt);
        const uint64_t ntile = (uint64_t)nseg * nstrip;
        if (ntile > 0xFFFFFFFFull) {
            err = "outer cell with more than 2^32 tiles";
            return false;
        }
        const uint32_t per = ntile >= 4096 ? 4 : (ntile >= 1024 ? 2 : 1);
        for (uint64_t t0 = 0; t0 < ntile; t0 += per) {
            w.t_begin = (uint32_t)t0, w.t_end = (uint32_t)std::min<uint64_t>(ntile, t0 + per);
            work.push_back(w);
        }
        return true;
    };
    auto make_entry = [&](const b2x_outer_term &t, int64_t dr, int64_t dc) {
        OEntry e{};
        e.a_off = t.a_off + (uint64_t)(dr * t.a_rs + dc * t.a_cs);
        e.b_off = t.b_off + (uint64_t)(dr * t.b_rs + dc * t.b_cs);
        e.alpha = t.alpha, e.a_rs = t.a_rs, e.a_cs = t.a_cs, e.b_rs = t.b_rs, e.b_cs = t.b_cs;
        e.a_src = t.a_src, e.b_src = t.b_src;
        if (t.a_src == 2) // the constant 1.0: the kernel still forms an address, keep it inside the arena
            e.a_off = 0, e.a_rs = e.a_cs = 0;
        if (t.b_src == 2)
            e.b_off = 0, e.b_rs = e.b_cs = 0;
        entries.push_back(e);
    };
    for (const Component &c : comps) {
        const uint32_t nw = c.w_end - c.w_begin;
        // The usual component of a blocking list: a handful of terms that all write the SAME window (one sub-block of an
        // enlarged operator) — one cell, the terms in plan order (the sort by offset is stable), none of the grid machinery.
        bool same = true;
        for (uint32_t wi = c.w_begin + 1; wi < c.w_end && same; wi++)
            same = win[wi].off == win[c.w_begin].off && win[wi].m == win[c.w_begin].m && win[wi].n == win[c.w_begin].n;
        if (same && win[c.w_begin].off == c.base) {
            const uint32_t eb = (uint32_t)entries.size();
            for (uint32_t wi = c.w_begin; wi < c.w_end; wi++)
                make_entry(terms[win[wi].pair], 0, 0);
            if (!emit_cell(c, 0, win[c.w_begin].m, 0, win[c.w_begin].n, eb, (int)nw))
                return B2X_ERR_INVALID;
            continue;
        }
        rb.assign({0, c.rows}), cb.assign({0, c.cols});
        wr0.resize(nw), wc0.resize(nw);
        for (uint32_t wi = c.w_begin; wi < c.w_end; wi++) {
            const uint64_t rel = win[wi].off - c.base;
            int r0, c0;
            if (rel <= 0xFFFFFFFFull) // (a 32-bit division costs a third of a 64-bit one)
                r0 = (int)((uint32_t)rel / (uint32_t)c.ld), c0 = (int)((uint32_t)rel - (uint32_t)r0 * (uint32_t)c.ld);
            else
                r0 = (int)(rel / (uint64_t)c.ld), c0 = (int)(rel % (uint64_t)c.ld);
            wr0[wi - c.w_begin] = r0, wc0[wi - c.w_begin] = c0;
            rb.push_back(r0), rb.push_back(r0 + win[wi].m), cb.push_back(c0), cb.push_back(c0 + win[wi].n);
        }
        std::sort(rb.begin(), rb.end()), rb.erase(std::unique(rb.begin(), rb.end()), rb.end());
        std::sort(cb.begin(), cb.end()), cb.erase(std::unique(cb.begin(), cb.end()), cb.end());
        const size_t nr = rb.size() - 1, ncl = cb.size() - 1;
        // plan order inside a cell
        members.resize(nw);
        for (uint32_t k = 0; k < nw; k++)
            members[k] = c.w_begin + k;
        auto by_pair = [&](uint32_t a, uint32_t b) { return win[a].pair < win[b].pair; };
        if (!std::is_sorted(members.begin(), members.end(), by_pair))
            std::sort(members.begin(), members.end(), by_pair);
        spans.resize(nw);
        cell_begin.assign(nr * ncl + 1, 0);
        for (uint32_t k = 0; k < nw; k++) {
            const Window &w = win[members[k]];
            const int r0 = wr0[members[k] - c.w_begin], c0 = wc0[members[k] - c.w_begin];
            Span sp;
            sp.a0 = (uint32_t)(std::lower_bound(rb.begin(), rb.end(), r0) - rb.begin());
            sp.b0 = (uint32_t)(std::lower_bound(cb.begin(), cb.end(), c0) - cb.begin());
            sp.a1 = sp.a0, sp.b1 = sp.b0;
            while (sp.a1 < nr && rb[sp.a1] < r0 + w.m)
                sp.a1++;
            while (sp.b1 < ncl && cb[sp.b1] < c0 + w.n)
                sp.b1++;
            spans[k] = sp;
            for (uint32_t a = sp.a0; a < sp.a1; a++)
                for (uint32_t b = sp.b0; b < sp.b1; b++)
                    cell_begin[a * ncl + b + 1]++;
        }
        for (size_t i = 0; i < nr * ncl; i++)
            cell_begin[i + 1] += cell_begin[i];
        flat.resize(cell_begin[nr * ncl]);
        cell_fill.assign(cell_begin.begin(), cell_begin.end() - 1);
        for (uint32_t k = 0; k < nw; k++) { // (members in plan order: every cell's list comes out in plan order)
            const Span &sp = spans[k];
            for (uint32_t a = sp.a0; a < sp.a1; a++)
                for (uint32_t b = sp.b0; b < sp.b1; b++)
                    flat[cell_fill[a * ncl + b]++] = members[k];
        }
        for (size_t a = 0; a < nr; a++)
            for (size_t b = 0; b < ncl; b++) {
                const uint32_t *lst = flat.data() + cell_begin[a * ncl + b];
                const int T = (int)(cell_begin[a * ncl + b + 1] - cell_begin[a * ncl + b]);
                if (T == 0)
                    continue;
                const uint32_t eb = (uint32_t)entries.size();
                for (int li = 0; li < T; li++) {
                    const uint32_t wi = lst[li];
                    // cell origin inside the term window
                    make_entry(terms[win[wi].pair], rb[a] - wr0[wi - c.w_begin], cb[b] - wc0[wi - c.w_begin]);
                }
                if (!emit_cell(c, rb[a], rb[a + 1], cb[b], cb[b + 1], eb, T))
                    return B2X_ERR_INVALID;
            }
    }
    pc.lap("o2 cells");
    pc.report();
    return B2X_OK;
}

int compile_diag(size_t n_terms, const b2x_diag_term *terms, size_t diag_len, uint64_t arena_len,
                 std::vector<DiagComp> &comps, std::vector<DiagTermD> &dterms, std::string &err) {
    comps.clear(), dterms.clear();
    std::vector<uint32_t> order(n_terms);
    for (size_t i = 0; i < n_terms; i++) {
        const b2x_diag_term &t = terms[i];
        order[i] = (uint32_t)i;
        if (t.m <= 0 || t.n <= 0 || t.ldc < t.n || t.a_stride <= 0 || t.b_stride <= 0 ||
            t.c_off + (uint64_t)(t.m - 1) * t.ldc + t.n > diag_len ||
            t.a_off + (uint64_t)(t.m - 1) * t.a_stride >= arena_len ||
            t.b_off + (uint64_t)(t.n - 1) * t.b_stride >= arena_len) {
            err = "diag term " + std::to_string(i) + ": window or operand out of range";
            return B2X_ERR_INVALID;
        }
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return terms[a].c_off < terms[b].c_off; });
    size_t i = 0;
    while (i < n_terms) {
        const b2x_diag_term &t0 = terms[order[i]];
        uint64_t end = t0.c_off + (uint64_t)(t0.m - 1) * t0.ldc + t0.n;
        size_t j = i + 1;
        while (j < n_terms && terms[order[j]].c_off < end) {
            const b2x_diag_term &t = terms[order[j]];
            end = std::max(end, t.c_off + (uint64_t)(t.m - 1) * t.ldc + t.n);
            j++;
        }
        DiagComp c{};
        c.base = t0.c_off, c.ld = 0, c.rows = 0, c.cols = 0;
        for (size_t k = i; k < j; k++)
            if (terms[order[k]].m > 1)
                c.ld = terms[order[k]].ldc; // (a one-column sector legitimately has ld == 1)
        if (c.ld == 0) { // only single-row windows: any ld that holds them all
            c.ld = 1;
            for (size_t k = i; k < j; k++)
                c.ld = std::max(c.ld, (int)(terms[order[k]].c_off - c.base) + terms[order[k]].n);
        }
        // column alignment as in compile_plan: a window must not wrap around a row
        int c0 = 0;
        for (size_t k = i; k < j; k++) {
            uint64_t rel = terms[order[k]].c_off - c.base;
            if ((int)(rel % (uint64_t)c.ld) + terms[order[k]].n > c.ld)
                c0 = (int)((uint64_t)c.ld - rel % (uint64_t)c.ld) % c.ld;
        }
        if ((uint64_t)c0 > c.base) {
            err = "diag windows do not share a row alignment";
            return B2X_ERR_INVALID;
        }
        c.base -= (uint64_t)c0;
        c.term_begin = (uint32_t)dterms.size();
        std::vector<uint32_t> members(order.begin() + i, order.begin() + j);
        std::sort(members.begin(), members.end()); // plan order inside the sector
        for (uint32_t id : members) {
            const b2x_diag_term &t = terms[id];
            if (t.m > 1 && t.ldc != c.ld) {
                err = "overlapping diag windows with different leading dimensions";
                return B2X_ERR_INVALID;
            }
            uint64_t rel = t.c_off - c.base;
            DiagTermD d{};
            d.a_off = t.a_off, d.b_off = t.b_off, d.alpha = t.alpha;
            d.row0 = (int)(rel / (uint64_t)c.ld), d.col0 = (int)(rel % (uint64_t)c.ld);
            if (d.col0 + t.n > c.ld) {
                err = "diag window wraps around a row";
                return B2X_ERR_INVALID;
            }
            d.m = t.m, d.n = t.n, d.a_stride = t.a_stride, d.b_stride = t.b_stride;
            c.rows = std::max(c.rows, d.row0 + d.m), c.cols = std::max(c.cols, d.col0 + d.n);
            dterms.push_back(d);
        }
        c.term_end = (uint32_t)dterms.size();
        comps.push_back(c);
        i = j;
    }
    return B2X_OK;
}

} // namespace b2x
