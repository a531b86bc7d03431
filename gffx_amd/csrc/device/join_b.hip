// join_b.hip -- Join B: "does ANY region of the line's seqid satisfy the predicate" for every GFF
// line of the hit blocks (reference: commands/intersect.rs:500-521, the numeric core of
// gff_line_overlaps_queries; called for every non-comment line of every hit block, :284-321).
//
// The reference scans all regions of the seqid linearly per line: O(lines x regions_of_seqid),
// its dominant cost.  Here the regions are radix-sorted ON THE DEVICE by (seqid, start) once per run (radix_sort.hpp,
// stable: ties keep the BED order) and two monotone helper columns make each mode one or two directory lookups.  With
// (s, e) the RAW column-4/5 integers of the line (1-based closed, no swap: intersect.rs:475-489) and the regions' raw
// (qs, qe) (no s<e check: intersect.rs:223-225):
//   Contained       exists q: s >= qs && e <= qe   <=>  PM(s) >= e      PM(x) = max{qe : qs <= x}
//   ContainsRegion  exists q: s <= qs && e >= qe   <=>  SM(s) <= e      SM(x) = min{qe : qs >= x}
//   Overlap  (qs<=s<=qe) || (qs<=e<=qe) || (s<=qs<=e) || (s<=qe<=e)     (intersect.rs:512-515)
//     a line with s <= e:  for a region with qs <= qe the four clauses are exactly  qs <= e && qe >= s, so over those
//                          regions the answer is  PM(e) >= s  -- ONE lookup.  A DEGENERATE region (qs > qe; a zero-length
//                          BED row is one) can only match clauses 3 and 4, i.e. when one of its two ends lies in [s, e];
//                          if it passes the PM test (qs <= e && qe >= s) both ends do, so PM over ALL regions gives no
//                          false positive, and the degenerate ones are completed by
//                              CD(first qs > e) - CD(first qs >= s) > 0      CD(i) = degenerate regions before position i
//                              || some degenerate qe in [s, e]               DE = their ends, sorted per seqid
//                          (both skipped when the run has no degenerate region: the usual case).
//     a line with s > e:   clauses 3 and 4 are empty, the rest is  PM(s) >= s || PM(e) >= e.
// Each clause is a conjunction of two comparisons on one region, so the rewrite is exact for every input; tests check it
// against the oracle's literal scan.  Every lookup first narrows to one bin of a per-seqid directory over the sorted
// starts (~2 bins per region), so it touches the directory pair and one or two 16-byte records instead of a 17-step
// binary search.  No invert here: intersect.rs:232-240 has no such parameter.
//
// HBM layout: line table SoA {seq, start, end} u32 x n_lines, file order (neighbouring lanes = neighbouring lines =
// nearby coordinates -> the lookups of a wave walk the same cache lines); per run: T[i] = {qs, PM, SM, CD} (16 B, one
// load) for the regions in (seqid, start) order + the sentinel T[n], dir (u32 per bin), SeqMeta (32 B per seqid, staged
// in LDS), DE.  One thread per line, one byte out.  Roofline bound: HBM; algorithmic bytes per line: 12 in + 1 out.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <memory>
#include <numeric>
#include <vector>

#include "gffx_device.hpp"
#include "radix_sort.hpp"
#include "regions_store.hpp"

namespace gffx {

struct LinesView {
    const uint32_t *seq, *start, *end;
    unsigned long long n;
};
// per seqid: its regions are T[q_lo, q_hi); its directory is dir[d_base .. d_base + nb]: dir[d_base + b] = first
// position whose start >= b << shift (entry nb = q_hi); its degenerate ends are DE[dq_lo, dq_hi)
struct SeqMeta {
    uint32_t q_lo, q_hi, shift, nb;
    uint32_t d_base, dq_lo, dq_hi, pad;
};
struct RegionsView {
    const uint4 *T;        // n + 1 records {qs, pm, sm, cd}
    const uint32_t *dir;
    const SeqMeta *meta;   // n_seq
    const uint32_t *de;    // sorted {seqid, qe} pairs of the degenerate regions (word 2 p + 1 = the end)
    const uint32_t *dird;  // their directory, same bins as `dir`: dird[d_base + b] = first pair of the seqid whose end >= b << shift
    uint32_t n_seq, n_deg;
};
#ifndef GFFX_META_LDS_MAX
#define GFFX_META_LDS_MAX 256
#endif
constexpr uint32_t kMetaLds = GFFX_META_LDS_MAX;  // seqids whose SeqMeta a block stages in LDS (more: read through the caches)

// first position in [lo, hi) whose start is > x (UPPER) / >= x
template <bool UPPER>
__device__ __forceinline__ uint32_t bound_qs(const uint4 *T, uint32_t lo, uint32_t hi, uint32_t x) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = T[mid].x;
        if (UPPER ? v <= x : v < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// PM(x) = max{qe : qs <= x} over the seqid's regions; false: no region starts at or before x.  *pos = first position with
// start > x.  The directory pair gives the bin [a, c); T[a - 1] and T[a] are loaded together, which settles bins of 0 or
// 1 regions (most of them, at 2 bins per region) without a further step.
__device__ __forceinline__ bool pm_at(const RegionsView &R, const SeqMeta &m, uint32_t x, uint32_t &pm, uint32_t *pos = nullptr) {
    const uint32_t b = x >> m.shift;
    uint32_t u;
    if (b >= m.nb) {
        u = m.q_hi;  // beyond the largest start of the seqid
    } else {
        const uint32_t a = min(max(R.dir[m.d_base + b], m.q_lo), m.q_hi), c = min(max(R.dir[m.d_base + b + 1], a), m.q_hi);  // (clamped: see k_b_carry)
        const uint4 t0 = R.T[a], tm = R.T[a > m.q_lo ? a - 1 : a];
        if (c == a || t0.x > x) {
            if (pos) *pos = a;
            pm = tm.y;
            return a > m.q_lo;
        }
        if (c == a + 1) {
            if (pos) *pos = c;
            pm = t0.y;
            return true;
        }
        u = bound_qs<true>(R.T, a + 1, c, x);
    }
    if (pos) *pos = u;
    if (u == m.q_lo) return false;
    pm = R.T[u - 1].y;
    return true;
}

// first position of the seqid whose start is >= x (q_hi: none); *t = its record when there is one
__device__ __forceinline__ uint32_t lower_qs(const RegionsView &R, const SeqMeta &m, uint32_t x, uint4 *t = nullptr) {
    const uint32_t b = x >> m.shift;
    if (b >= m.nb) return m.q_hi;
    const uint32_t a = min(max(R.dir[m.d_base + b], m.q_lo), m.q_hi - 1), c = min(max(R.dir[m.d_base + b + 1], a), m.q_hi);  // (clamped: see k_b_carry)
    const uint4 t0 = R.T[a], t1 = R.T[a + 1];  // (T has n + 1 records)
    uint32_t l;
    if (c == a || t0.x >= x) {
        if (t) *t = t0;
        return a;  // (an empty bin below the seqid's last one: a < q_hi)
    }
    if (c == a + 1) {
        if (t) *t = t1;
        return c;
    }
    l = bound_qs<false>(R.T, a + 1, c, x);
    if (t && l < m.q_hi) *t = R.T[l];
    return l;
}

// some degenerate end in [s, e] (the seqid's bins: ends are below their starts, so below the largest start)
__device__ __forceinline__ bool deg_end_in(const RegionsView &R, const SeqMeta &m, uint32_t s, uint32_t e) {
    if (m.dq_hi == m.dq_lo) return false;
    const uint32_t b = s >> m.shift;
    if (b >= m.nb) return false;
    uint32_t lo = min(max(R.dird[m.d_base + b], m.dq_lo), m.dq_hi), hi = min(max(R.dird[m.d_base + b + 1], lo), m.dq_hi);
    while (hi - lo > 2) {  // first end >= s inside the bin (a bin holds ~0.5 ends; clustered ones are halved down first)
        const uint32_t mid = (lo + hi) >> 1;
        if (R.de[2 * mid + 1] < s)
            lo = mid + 1;
        else
            hi = mid;
    }
    while (lo < hi && R.de[2 * lo + 1] < s) ++lo;
    return lo < m.dq_hi && R.de[2 * lo + 1] <= e;
}

// DEG: the run has regions with start > end (Overlap only): their clauses are evaluated for the lines the PM test did not
// keep (for every line, side by side with the PM lookup, was faster at 1 M regions -- 35 vs 40 us -- and twice slower at 10 M,
// where PM keeps every line).
template <int MODE, bool META_LDS, bool DEG>
__global__ __launch_bounds__(256) void k_lines_exists(LinesView L, RegionsView R, uint8_t *keep) {
    __shared__ uint4 s_meta[META_LDS && kMetaLds ? 2 * kMetaLds : 2];
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < L.n;
    const uint32_t seq = live ? L.seq[i] : 0xFFFFFFFFu;
    const uint32_t s = live ? L.start[i] : 0u, e = live ? L.end[i] : 0u;
    if (META_LDS) {
        for (uint32_t t = threadIdx.x; t < 2 * R.n_seq; t += 256) s_meta[t] = reinterpret_cast<const uint4 *>(R.meta)[t];
        __syncthreads();
    }
    if (!live) return;
    uint8_t k = 0;
    if (seq < R.n_seq) {
        SeqMeta m;
        if (META_LDS) {
            const uint4 m0 = s_meta[2 * seq], m1 = s_meta[2 * seq + 1];
            m = SeqMeta{m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        } else {
            m = R.meta[seq];
        }
        if (m.q_hi > m.q_lo) {  // a seqid without regions has no map entry (intersect.rs:495-498)
            uint32_t pm = 0;
            if (MODE == GFFX_MODE_CONTAINED) {
                k = (pm_at(R, m, s, pm) && pm >= e) ? 1 : 0;
            } else if (MODE == GFFX_MODE_CONTAINS_REGION) {
                uint4 t = make_uint4(0, 0, 0, 0);
                const uint32_t l = lower_qs(R, m, s, &t);
                k = (l < m.q_hi && t.z <= e) ? 1 : 0;
            } else if (s <= e) {
                uint32_t u = 0;
                bool any = pm_at(R, m, e, pm, &u) && pm >= s;
                if (DEG && !any) {  // the degenerate regions (qs > qe): one of their two ends in [s, e]
                    const uint32_t l = lower_qs(R, m, s);
                    const bool d4 = deg_end_in(R, m, s, e);  // (independent of the line above: their loads travel together)
                    any = R.T[u].w != R.T[l].w || d4;
                }
                k = any ? 1 : 0;
            } else {
                bool any = pm_at(R, m, s, pm) && pm >= s;  // qs <= s <= qe
                if (!any) any = pm_at(R, m, e, pm) && pm >= e;  // qs <= e <= qe
                k = any ? 1 : 0;
            }
        }
    }
    keep[i] = k;
}

// ---- the usual run (no degenerate regions): kLinesPerThread lines per thread, their lookups issued side by side -------------
// One line per thread is a chain of three dependent loads (SeqMeta from LDS aside): line -> directory pair -> T pair, and the
// 53 k waves of a 3.4 M-line table pass through the chip in ~6.5 generations of that chain (20 us).  Here a thread probes
// its lines TOGETHER -- the probe is branch-free up to the T loads, so the directory pairs of all of them are in flight at
// once, then their T pairs -- and resolves them afterwards (bins of 0 / 1 regions from the pair at hand; longer bins by the
// binary search, rare).
#ifndef GFFX_LINES_PER_THREAD
#define GFFX_LINES_PER_THREAD 2
#endif
constexpr int kLinesPerThread = GFFX_LINES_PER_THREAD;
struct Probe {
    uint32_t a, c;  // the bin's regions are T[a, c)
    uint4 t0, t1;   // T[a]; T[a - 1] (running-max lookups) or T[a + 1] (running-min lookups)
    bool in;        // x's bin exists (x is not beyond the seqid's largest start)
};
template <bool SM>
__device__ __forceinline__ Probe probe(const RegionsView &R, const SeqMeta &m, uint32_t x) {
    Probe p;
    const uint32_t b = x >> m.shift;
    p.in = b < m.nb;
    const uint32_t bi = m.d_base + (p.in ? b : 0u);  // (out of range, or a seqid without regions: any valid directory word)
    const uint32_t a = min(max(R.dir[bi], m.q_lo), SM ? max(m.q_hi, 1u) - 1u : m.q_hi);
    p.a = a;
    p.c = min(max(R.dir[bi + 1], a), m.q_hi);
    p.t0 = R.T[a];
    p.t1 = SM ? R.T[a + 1] : R.T[a > m.q_lo ? a - 1 : a];  // (T has n + 1 records)
    return p;
}
// PM(x) = max{qe : qs <= x}; false: no region of the seqid starts at or before x
__device__ __forceinline__ bool resolve_pm(const RegionsView &R, const SeqMeta &m, uint32_t x, const Probe &p, uint32_t &pm) {
    if (!p.in) {  // beyond the largest start
        if (m.q_hi == m.q_lo) return false;
        pm = R.T[m.q_hi - 1].y;
        return true;
    }
    if (p.c == p.a || p.t0.x > x) {
        pm = p.t1.y;
        return p.a > m.q_lo;
    }
    if (p.c == p.a + 1) {
        pm = p.t0.y;
        return true;
    }
    pm = R.T[bound_qs<true>(R.T, p.a + 1, p.c, x) - 1].y;
    return true;
}
// SM(x) = min{qe : qs >= x}; false: no region of the seqid starts at or after x
__device__ __forceinline__ bool resolve_sm(const RegionsView &R, const SeqMeta &m, uint32_t x, const Probe &p, uint32_t &sm) {
    if (!p.in || m.q_hi == m.q_lo) return false;
    if (p.c == p.a || p.t0.x >= x) {
        sm = p.t0.z;
        return true;  // (an empty bin below the seqid's last one: a < q_hi)
    }
    uint32_t l = p.c;
    uint4 t = p.t1;
    if (p.c != p.a + 1) {
        l = bound_qs<false>(R.T, p.a + 1, p.c, x);
        if (l < m.q_hi) t = R.T[l];
    }
    sm = t.z;
    return l < m.q_hi;
}

template <int MODE, bool META_LDS>
__global__ __launch_bounds__(256) void k_lines_exists2(LinesView L, RegionsView R, uint8_t *keep) {
    __shared__ uint4 s_meta[META_LDS && kMetaLds ? 2 * kMetaLds : 2];
    const unsigned long long i0 = (unsigned long long)blockIdx.x * (256 * kLinesPerThread) + threadIdx.x;
    uint32_t seq[kLinesPerThread], s[kLinesPerThread], e[kLinesPerThread];
#pragma unroll
    for (int r = 0; r < kLinesPerThread; ++r) {
        const unsigned long long i = i0 + 256ull * r;
        const bool live = i < L.n;
        seq[r] = live ? L.seq[i] : 0xFFFFFFFFu, s[r] = live ? L.start[i] : 0u, e[r] = live ? L.end[i] : 0u;
    }
    if (META_LDS) {
        for (uint32_t t = threadIdx.x; t < 2 * R.n_seq; t += 256) s_meta[t] = reinterpret_cast<const uint4 *>(R.meta)[t];
        __syncthreads();
    }
    SeqMeta m[kLinesPerThread];
    Probe p[kLinesPerThread];
#pragma unroll
    for (int r = 0; r < kLinesPerThread; ++r) {
        m[r] = SeqMeta{0, 0, 0, 0, 0, 0, 0, 0};  // (a line of no seqid: a seqid without regions)
        if (seq[r] < R.n_seq) {
            if (META_LDS) {
                const uint4 m0 = s_meta[2 * seq[r]], m1 = s_meta[2 * seq[r] + 1];
                m[r] = SeqMeta{m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            } else {
                m[r] = R.meta[seq[r]];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kLinesPerThread; ++r)  // Contained / ContainsRegion look s up; Overlap (s <= e) looks e up
        p[r] = probe<MODE == GFFX_MODE_CONTAINS_REGION>(R, m[r], MODE == GFFX_MODE_OVERLAP ? e[r] : s[r]);
#pragma unroll
    for (int r = 0; r < kLinesPerThread; ++r) {
        const unsigned long long i = i0 + 256ull * r;
        if (i >= L.n) continue;
        uint32_t v = 0;
        bool k;
        if (MODE == GFFX_MODE_CONTAINED) {
            k = resolve_pm(R, m[r], s[r], p[r], v) && v >= e[r];
        } else if (MODE == GFFX_MODE_CONTAINS_REGION) {
            k = resolve_sm(R, m[r], s[r], p[r], v) && v <= e[r];
        } else if (s[r] <= e[r]) {
            k = resolve_pm(R, m[r], e[r], p[r], v) && v >= s[r];
        } else {  // s > e: clauses 3 / 4 are empty (rare: one lookup more)
            k = resolve_pm(R, m[r], e[r], p[r], v) && v >= e[r];  // qs <= e <= qe
            if (!k) k = m[r].q_hi > m[r].q_lo && pm_at(R, m[r], s[r], v) && v >= s[r];  // qs <= s <= qe
        }
        keep[i] = (k && m[r].q_hi > m[r].q_lo) ? 1 : 0;
    }
}

// ---- region tables on the device (what the reference builds per run as `query_ivmap`, intersect.rs:621-633) --------------
// The records {seqid, qs, qe} are radix-sorted by (seqid, qs) (radix_sort.hpp).  Three kernels turn the sorted records into
// the tables: k_b_local (per 1024-region block: the T records with block-local running max / min / degenerate count, the
// per-seqid offsets, the block's carries), k_b_carry (ONE block: scans the carries of all blocks, lays out the directories),
// k_b_finish (folds the carries in, fills the directory, writes the degenerate regions' {seqid, qe} out in order).
constexpr int kScanBlock = 1024;
constexpr int kScanWaves = kScanBlock / 64;

struct BlockCarry {          // per 1024-region block, arrays of n_blocks (+ 1) entries
    uint32_t *max_out, *min_out;    // k_b_local: running max at the block's last region / running min at its first
    uint32_t *head_any;             // bit0: a seqid starts inside the block, bit1: a seqid ends inside it
    uint32_t *first_head, *last_tail;  // thread index of the first region that starts a seqid (1024: none) / the last that ends one (-1)
    uint32_t *deg;                  // degenerate regions in the block
    uint32_t *max_in, *min_in, *deg_in;  // k_b_carry: what the block's open seqid brings in from the left / right; degenerate regions before it
};

template <bool MAXOP>
__device__ __forceinline__ uint32_t scan_op(uint32_t a, uint32_t b) {
    return MAXOP ? max(a, b) : min(a, b);
}
// segmented inclusive scan over the 1024 threads of a block, in thread order; f = 1 starts a segment.  Returns the scanned
// value; f becomes "a segment start at or before me inside the block".  s_v / s_f: 16 words each.
template <bool MAXOP>
__device__ __forceinline__ uint32_t block_seg_scan(uint32_t v, uint32_t &f, uint32_t *s_v, uint32_t *s_f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t pv = __shfl_up(v, o, 64), pf = __shfl_up(f, o, 64);
        if (lane >= o) {
            if (!f) v = scan_op<MAXOP>(pv, v);
            f |= pf;
        }
    }
    if (lane == 63) s_v[wave] = v, s_f[wave] = f;
    __syncthreads();
    // carry of the waves before me: the value of the segment still open where this wave starts
    uint32_t cv = 0, cf = 0;
    for (int w = 0; w < wave; ++w) {
        cv = (w == 0 || s_f[w]) ? s_v[w] : scan_op<MAXOP>(cv, s_v[w]);
        cf |= s_f[w];
    }
    if (wave > 0 && !f) v = scan_op<MAXOP>(cv, v);
    f |= cf;
    __syncthreads();  // (s_v / s_f are reused by the next scan)
    return v;
}
// exclusive sum over the 1024 threads; *total = the block's sum
__device__ __forceinline__ uint32_t block_sum_scan(uint32_t v, uint32_t *s_v, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_v[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int w = 0; w < kScanWaves; ++w) {
        base += w < wave ? s_v[w] : 0u;
        all += s_v[w];
    }
    *total = all;
    __syncthreads();
    return base + inc - v;
}

// 256 threads x 4 consecutive regions = one block of 1024: a thread scans its four regions in registers (three 16-byte loads
// in, four 16-byte records out), the threads' aggregates are scanned with wave shuffles, the four waves meet at ONE barrier.
constexpr int kLocalItems = 4, kLocalThreads = kScanBlock / kLocalItems, kLocalWaves = kLocalThreads / 64;
__global__ __launch_bounds__(kLocalThreads) void k_b_local(const uint32_t *rec, unsigned long long n, uint32_t n_seq, uint4 *T, uint32_t *q_off,
                                                           BlockCarry C) {
    __shared__ uint32_t s_v[kLocalWaves], s_f[kLocalWaves], s_c[kLocalWaves], s_bv[kLocalWaves], s_bf[kLocalWaves];
    __shared__ uint32_t s_first, s_last;
    if (threadIdx.x == 0) s_first = kScanBlock, s_last = 0xFFFFFFFFu;
    const unsigned long long i0 = (unsigned long long)blockIdx.x * kScanBlock + (unsigned long long)threadIdx.x * kLocalItems;
    uint32_t seq[kLocalItems], qs[kLocalItems], qe[kLocalItems];
    if (i0 + kLocalItems <= n) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(rec + 3 * i0);  // (i0 is a multiple of 4: 48-byte steps, 16-byte aligned)
        const uint4 a = r4[0], b = r4[1], c = r4[2];
        seq[0] = a.x, qs[0] = a.y, qe[0] = a.z, seq[1] = a.w, qs[1] = b.x, qe[1] = b.y;
        seq[2] = b.z, qs[2] = b.w, qe[2] = c.x, seq[3] = c.y, qs[3] = c.z, qe[3] = c.w;
    } else {
#pragma unroll
        for (int k = 0; k < kLocalItems; ++k) {
            const bool live = i0 + k < n;
            seq[k] = live ? rec[3 * (i0 + k)] : 0xFFFFFFFFu, qs[k] = live ? rec[3 * (i0 + k) + 1] : 0u, qe[k] = live ? rec[3 * (i0 + k) + 2] : 0u;
        }
    }
    const uint32_t seq_before = (i0 > 0 && i0 < n) ? rec[3 * (i0 - 1)] : 0xFFFFFFFEu;
    const uint32_t seq_after = i0 + kLocalItems < n ? rec[3 * (i0 + kLocalItems)] : 0xFFFFFFFEu;
    __syncthreads();
    // a region past the end is its own segment (head and tail) with neutral values
    bool head[kLocalItems], tail[kLocalItems];
    uint32_t pm[kLocalItems], sm[kLocalItems], cd[kLocalItems];
    uint32_t run = 0, tf = 0, n_deg = 0;
    int first_head = kLocalItems, last_tail = -1;
#pragma unroll
    for (int k = 0; k < kLocalItems; ++k) {
        const bool live = i0 + k < n;
        head[k] = !live || seq[k] != (k ? seq[k - 1] : seq_before);
        tail[k] = !live || i0 + k + 1 == n || seq[k] != (k + 1 < kLocalItems ? seq[k + 1] : seq_after);
        if (live && head[k]) {  // q_off[c] = first sorted position of seqid c: every boundary fills the seqids it skips over
            const uint32_t c1 = min(seq[k], n_seq);  // (a seqid out of range is reported by the sort's histogram kernel)
            const long long c0 = i0 + k ? (long long)min(k ? seq[k - 1] : seq_before, n_seq) : -1;
            for (long long c = c0 + 1; c <= (long long)c1; ++c) q_off[c] = (uint32_t)(i0 + k);
            if (first_head == kLocalItems) first_head = k;
        }
        if (live && i0 + k + 1 == n)
            for (uint32_t c = min(seq[k], n_seq) + 1; c <= n_seq; ++c) q_off[c] = (uint32_t)n;
        if (live && tail[k]) last_tail = k;
        run = head[k] ? (live ? qe[k] : 0u) : max(run, qe[k]);
        tf |= head[k] ? 1u : 0u;
        pm[k] = run;
        cd[k] = n_deg;
        n_deg += (live && qs[k] > qe[k]) ? 1u : 0u;
    }
    uint32_t brun = 0xFFFFFFFFu, tb = 0;
#pragma unroll
    for (int k = kLocalItems - 1; k >= 0; --k) {
        const bool live = i0 + k < n;
        brun = tail[k] ? (live ? qe[k] : 0xFFFFFFFFu) : min(brun, qe[k]);
        tb |= tail[k] ? 1u : 0u;
        sm[k] = brun;
    }
    if (first_head < kLocalItems) atomicMin(&s_first, threadIdx.x * kLocalItems + first_head);  // (read after the barrier below)
    if (last_tail >= 0) atomicMax((int *)&s_last, (int)(threadIdx.x * kLocalItems + last_tail));
    // inclusive scans of the threads' aggregates inside the wave: running max from the left (restarts at a head), running
    // min from the right (restarts at a tail), degenerate regions
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t iv = run, ff = tf, bv = brun, fb = tb, ic = n_deg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t pv = __shfl_up(iv, o, 64), pf = __shfl_up(ff, o, 64), nv = __shfl_down(bv, o, 64), nf = __shfl_down(fb, o, 64),
                       pc = __shfl_up(ic, o, 64);
        if (lane >= o) {
            if (!ff) iv = max(pv, iv);
            ff |= pf;
            ic += pc;
        }
        if (lane + o < 64) {
            if (!fb) bv = min(nv, bv);
            fb |= nf;
        }
    }
    if (lane == 63) s_v[wave] = iv, s_f[wave] = ff, s_c[wave] = ic;
    if (lane == 0) s_bv[wave] = bv, s_bf[wave] = fb;
    __syncthreads();
    // what the thread's open segments bring in: from the lanes / waves before me, and from those after me
    uint32_t cv = 0, cb = 0xFFFFFFFFu, cbase = 0, block_deg = 0;
#pragma unroll
    for (int w = 0; w < kLocalWaves; ++w) {
        if (w < wave) cv = s_f[w] ? s_v[w] : max(cv, s_v[w]);
        cbase += w < wave ? s_c[w] : 0u;
        block_deg += s_c[w];
    }
#pragma unroll
    for (int w = kLocalWaves - 1; w >= 0; --w)
        if (w > wave) cb = s_bf[w] ? s_bv[w] : min(cb, s_bv[w]);
    uint32_t ev = __shfl_up(iv, 1, 64), ef = __shfl_up(ff, 1, 64), nbv = __shfl_down(bv, 1, 64), nbf = __shfl_down(fb, 1, 64);
    if (lane == 0) ev = 0u, ef = 0u;
    if (lane == 63) nbv = 0xFFFFFFFFu, nbf = 0u;
    const uint32_t in_max = ef ? ev : max(cv, ev), in_min = nbf ? nbv : min(cb, nbv);
    const uint32_t cd0 = cbase + ic - n_deg;
#pragma unroll
    for (int k = 0; k < kLocalItems; ++k) {
        if (k < first_head && !(head[k])) pm[k] = max(pm[k], in_max);
        if (k > last_tail && !(tail[k])) sm[k] = min(sm[k], in_min);
        if (i0 + k < n) T[i0 + k] = make_uint4(qs[k], pm[k], sm[k], cd0 + cd[k]);
    }
    if (threadIdx.x == kLocalThreads - 1) C.max_out[blockIdx.x] = pm[kLocalItems - 1];  // (past the end: the last block's is never read)
    if (threadIdx.x == 0) {
        C.min_out[blockIdx.x] = sm[0];
        C.deg[blockIdx.x] = block_deg;
        C.first_head[blockIdx.x] = s_first;
        C.last_tail[blockIdx.x] = s_last;
        C.head_any[blockIdx.x] = (s_first < kScanBlock ? 1u : 0u) | (s_last != 0xFFFFFFFFu ? 2u : 0u);
    }
}

// ONE block.  (1) the carries of all region blocks: max_in[b] = running max of the seqid open at the end of block b - 1,
// min_in[b] = running min of the seqid open at the start of block b + 1, deg_in[b] = degenerate regions before block b;
// (2) per seqid the directory geometry {shift, nb} over its largest start (~2 bins per region, >= 16) and its place d_base.
// cnt[0] = degenerate regions in total, cnt[1] = directory words in total.
__global__ __launch_bounds__(kScanBlock) void k_b_carry(uint32_t n_blocks, BlockCarry C, const uint32_t *q_off, const uint4 *T, uint32_t n, uint32_t n_seq,
                                                        SeqMeta *meta, uint32_t *cnt, unsigned long long cap_dir) {
    __shared__ uint32_t s_v[kScanWaves], s_f[kScanWaves];
    __shared__ uint32_t s_run;
    // forward: running max
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += kScanBlock) {
        const uint32_t b = b0 + threadIdx.x;
        const bool live = b < n_blocks;
        uint32_t f = live ? (C.head_any[b] & 1u) : 1u;
        uint32_t v = block_seg_scan<true>(live ? C.max_out[b] : 0u, f, s_v, s_f);
        if (!f) v = max(v, s_run);  // the chunks before this one
        if (live) C.max_in[b + 1] = v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_run = v;
        __syncthreads();
    }
    // backward: running min (thread x of a chunk = block n_blocks - 1 - x0 - x)
    if (threadIdx.x == 0) s_run = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t x0 = 0; x0 < n_blocks; x0 += kScanBlock) {
        const uint32_t x = x0 + threadIdx.x;
        const bool live = x < n_blocks;
        const uint32_t b = live ? n_blocks - 1 - x : 0;
        uint32_t f = live ? ((C.head_any[b] >> 1) & 1u) : 1u;
        uint32_t v = block_seg_scan<false>(live ? C.min_out[b] : 0xFFFFFFFFu, f, s_v, s_f);
        if (!f) v = min(v, s_run);
        if (live && b > 0) C.min_in[b - 1] = v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_run = v;
        __syncthreads();
    }
    // degenerate regions before every block
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += kScanBlock) {
        const uint32_t b = b0 + threadIdx.x;
        uint32_t total = 0;
        const uint32_t ex = block_sum_scan(b < n_blocks ? C.deg[b] : 0u, s_v, &total);
        if (b < n_blocks) C.deg_in[b] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) cnt[0] = run;
    // directories
    run = 0;
    for (uint32_t c0 = 0; c0 < n_seq; c0 += kScanBlock) {
        const uint32_t c = c0 + threadIdx.x;
        SeqMeta m{0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t size = 0;
        if (c < n_seq) {
            // (a run with a seqid out of range fails on the host, but its kernels still run to the end: whatever the broken
            //  order left in q_off, no range may leave [0, n])
            m.q_lo = min(q_off[c], n), m.q_hi = min(q_off[c + 1], n);
            if (m.q_hi > m.q_lo) {
                const uint32_t vmax = T[m.q_hi - 1].x;
                const unsigned long long budget = max(2ull * (m.q_hi - m.q_lo), 16ull);
                uint32_t shift = 0;
                while ((((unsigned long long)vmax >> shift) + 1) > budget) shift++;
                m.shift = shift, m.nb = (vmax >> shift) + 1;
                size = m.nb + 1;
            }
        }
        uint32_t total = 0;
        const uint32_t ex = block_sum_scan(size, s_v, &total);
        if (c < n_seq) {
            m.d_base = run + ex;
            if ((unsigned long long)m.d_base + m.nb + 1 > cap_dir)  // (ranges that overlap: only the broken order of a failing run)
                m.q_hi = m.q_lo, m.nb = 0, m.shift = 0, m.d_base = 0;
            meta[c] = m;
        }
        run += total;
    }
    if (threadIdx.x == 0) cnt[1] = run;
}

// Folds the carries into T, fills the directory, writes the sentinel T[n] and the degenerate regions' {seqid, qe} (in
// (seqid, start) order, position = CD) for the sort by (seqid, qe).  Same shape as k_b_local: a block = one block of 1024
// regions (its carries are five scalar loads), a thread = four consecutive regions.
__global__ __launch_bounds__(kLocalThreads) void k_b_finish(const uint32_t *rec, unsigned long long n, uint32_t n_seq, uint4 *T, BlockCarry C,
                                                            const SeqMeta *meta, uint32_t *dir, uint32_t *deg_out, const uint32_t *cnt) {
    const uint32_t blk = blockIdx.x;
    const unsigned long long i0 = (unsigned long long)blk * kScanBlock + (unsigned long long)threadIdx.x * kLocalItems;
    if (i0 >= n) return;
    const uint32_t first_head = C.first_head[blk], max_in = C.max_in[blk], min_in = C.min_in[blk], deg_in = C.deg_in[blk];
    const int last_tail = (int)C.last_tail[blk];
    uint4 t[kLocalItems];
    uint32_t seq[kLocalItems], qe[kLocalItems];
    if (i0 + kLocalItems <= n) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(rec + 3 * i0);
        const uint4 a = r4[0], b = r4[1], c = r4[2];
        seq[0] = a.x, qe[0] = a.z, seq[1] = a.w, qe[1] = b.y, seq[2] = b.z, qe[2] = c.x, seq[3] = c.y, qe[3] = c.w;
#pragma unroll
        for (int k = 0; k < kLocalItems; ++k) t[k] = T[i0 + k];
    } else {
#pragma unroll
        for (int k = 0; k < kLocalItems; ++k) {
            const bool live = i0 + k < n;
            seq[k] = live ? rec[3 * (i0 + k)] : 0xFFFFFFFFu, qe[k] = live ? rec[3 * (i0 + k) + 2] : 0u;
            t[k] = live ? T[i0 + k] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    const uint32_t x_before = i0 ? T[i0 - 1].x : 0u;  // (the start never changes: no race with the thread that rewrites T[i0 - 1])
    uint4 ma[kLocalItems], mb[kLocalItems];  // SeqMeta; every load before the first store: the stores below may alias as far as the compiler knows
#pragma unroll
    for (int k = 0; k < kLocalItems; ++k) {
        const uint32_t c = min(seq[k], n_seq ? n_seq - 1 : 0u);
        ma[k] = reinterpret_cast<const uint4 *>(meta)[2 * (size_t)c], mb[k] = reinterpret_cast<const uint4 *>(meta)[2 * (size_t)c + 1];
    }
    const uint32_t total_deg = cnt[0];
#pragma unroll
    for (int k = 0; k < kLocalItems; ++k) {
        const unsigned long long i = i0 + k;
        if (i >= n) break;
        const uint32_t tid = threadIdx.x * kLocalItems + k;
        if (tid < first_head) t[k].y = max(t[k].y, max_in);  // (block 0 starts with a head)
        if ((int)tid > last_tail) t[k].z = min(t[k].z, min_in);
        t[k].w += deg_in;
        T[i] = t[k];
        if (t[k].x > qe[k]) deg_out[2 * (size_t)t[k].w] = seq[k], deg_out[2 * (size_t)t[k].w + 1] = qe[k];
        if (i + 1 == n) T[n] = make_uint4(0u, 0u, 0u, total_deg);
        if (seq[k] >= n_seq) continue;  // (reported by the sort's histogram kernel)
        const uint32_t q_lo = ma[k].x, q_hi = ma[k].y, shift = ma[k].z, nb = ma[k].w;
        uint32_t *d = dir + mb[k].x;
        const uint32_t b = min(t[k].x >> shift, nb);  // (min: only a run that fails anyway -- a seqid out of range breaks the order -- gets there)
        const long long bprev = i > q_lo ? (long long)min((k ? t[k - 1].x : x_before) >> shift, nb) : -1;
        for (long long x = bprev + 1; x <= (long long)b; ++x) d[x] = (uint32_t)i;
        if (i + 1 == q_hi)
            for (uint32_t x = b + 1; x <= nb; ++x) d[x] = q_hi;
    }
}

// the sorted {seqid, qe} pairs of the degenerate regions -> every seqid's range [dq_lo, dq_hi) in SeqMeta and its directory
// (the bins of the seqid's start directory: an end is below its start, so below the largest start)
__global__ __launch_bounds__(256) void k_b_deg_ranges(const uint32_t *pairs, uint32_t n, uint32_t n_seq, SeqMeta *meta, uint32_t *dird) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c1 = min(pairs[2 * i], n_seq), qe = pairs[2 * i + 1];
    const long long c0 = i ? (long long)min(pairs[2 * (i - 1)], n_seq) : -1;
    const bool last_of_seq = i + 1 == n || pairs[2 * (i + 1)] != pairs[2 * i];
    // seqids (c0, c1] start at i; seqids [c0, c1) end at i
    for (long long c = c0 + 1; c <= (long long)c1 && c < (long long)n_seq; ++c) meta[c].dq_lo = i;
    for (long long c = max(c0, 0ll); c < (long long)c1; ++c) meta[c].dq_hi = i;
    if (i + 1 == n) {
        if (c1 < n_seq) meta[c1].dq_hi = n;
        for (uint32_t c = c1 + 1; c < n_seq; ++c) meta[c].dq_lo = meta[c].dq_hi = n;
    }
    if (c1 >= n_seq) return;  // (reported by the sort's histogram kernel)
    const uint32_t shift = meta[c1].shift, nb = meta[c1].nb;  // (k_b_carry wrote them; this kernel only writes dq_lo / dq_hi)
    uint32_t *d = dird + meta[c1].d_base;
    const uint32_t b = min(qe >> shift, nb);
    const long long bprev = c0 == (long long)c1 ? (long long)min(pairs[2 * (i - 1) + 1] >> shift, nb) : -1;
    for (long long x = bprev + 1; x <= (long long)b; ++x) d[x] = i;
    if (last_of_seq)
        for (uint32_t x = b + 1; x <= nb; ++x) d[x] = i + 1;
}

}  // namespace gffx

using namespace gffx;

constexpr int kSortPassesMax = 4 + 4;  // the coordinate's four bytes + up to four of the seqid
constexpr size_t kWorkFront = 64;      // words of d_work in front of the sort's work space (d_err, d_cnt)

struct gffx_hip_lines {
    int device = 0;
    uint64_t n = 0;
    uint32_t *d_seq = nullptr, *d_start = nullptr, *d_end = nullptr;
    uint8_t *d_keep = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;  // bracket k_lines_exists of the last _test
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;  // ... and the device preparation of the region tables
    uint32_t note_seq = 0;                        // h_cnt = {degenerate regions, note_seq}, posted by the sort's histogram kernel
    int last_sort_passes = 0;                     // radix passes of the last run's region sort (4 with the top digit, else 4 + seqid bytes)
    double last_kernel_ms = 0.0, last_prep_ms = 0.0;
    // region tables of the last _test (grow-only device buffers)
    uint64_t cap_q = 0, cap_seq = 0, cap_work = 0, cap_dir = 0, cap_blocks = 0;
    uint32_t *d_rec_a = nullptr, *d_rec_b = nullptr;  // 3 u32 per region: input / sort ping-pong; later the degenerate pairs
    uint4 *d_T = nullptr;            // cap_q + 1
    uint32_t *d_carry = nullptr;     // BlockCarry: 9 arrays of cap_blocks + 1
    uint32_t *d_work = nullptr;      // sort work space
    uint32_t *d_qoff = nullptr;      // cap_seq + 1
    SeqMeta *d_meta = nullptr;       // cap_seq
    uint32_t *d_dir = nullptr;       // cap_dir
    uint32_t *d_dird = nullptr;      // cap_dird: the directory of the degenerate ends (allocated by the first run that has any)
    uint64_t cap_dird = 0;
    uint32_t *d_err = nullptr;       // 4 words at the front of d_work
    uint32_t *d_cnt = nullptr;       // d_work + 4: {degenerate regions (k_b_carry), directory words, degenerate regions (the sort's histogram kernel)}
    uint32_t *h_cnt = nullptr;       // pinned, coherent: the kernel writes it
    const uint32_t *d_de = nullptr;  // the sorted degenerate pairs of the last Overlap _test (inside d_rec_a / d_rec_b)
    uint64_t last_nq = 0;
    uint32_t last_n_seq = 0, last_n_deg = 0;
    bool last_deg_known = false;
};

template <typename T>
static int dalloc(T **p, size_t n) {
    *p = nullptr;
    GFFX_HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GFFX_OK;
}
template <typename T>
static int regrow(T **p, size_t n) {
    if (*p) GFFX_HIP_TRY(hipFree(*p));
    return dalloc(p, n);
}

extern "C" void gffx_hip_lines_destroy(gffx_hip_lines *L) {
    if (!L) return;
    (void)hipSetDevice(L->device);
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    for (void *p : {(void *)L->d_seq, (void *)L->d_start, (void *)L->d_end, (void *)L->d_keep, (void *)L->d_rec_a, (void *)L->d_rec_b,
                    (void *)L->d_T, (void *)L->d_carry, (void *)L->d_work, (void *)L->d_qoff, (void *)L->d_meta, (void *)L->d_dir, (void *)L->d_dird})
        (void)hipFree(p);
    if (L->h_cnt) (void)hipHostFree(L->h_cnt);
    for (hipEvent_t e : {L->ev_a, L->ev_b, L->ev_p0, L->ev_p1})
        if (e) (void)hipEventDestroy(e);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
}

extern "C" int gffx_hip_lines_create(int device, uint64_t n_lines, const uint32_t *seq,
                                     const uint32_t *start, const uint32_t *end,
                                     gffx_hip_lines **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_lines_create: out is NULL");
    *out = nullptr;
    if (n_lines && (!seq || !start || !end)) return fail(GFFX_E_INVALID, "gffx_hip_lines_create: NULL array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        ndev = 0;
    }
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<gffx_hip_lines, void (*)(gffx_hip_lines *)> L(new gffx_hip_lines, gffx_hip_lines_destroy);
    L->device = device;
    L->n = n_lines;
    int rc;
    if ((rc = dalloc(&L->d_seq, n_lines)) || (rc = dalloc(&L->d_start, n_lines)) ||
        (rc = dalloc(&L->d_end, n_lines)) || (rc = dalloc(&L->d_keep, n_lines)) || (rc = dalloc(&L->d_work, kWorkFront)) ||
        (rc = dalloc(&L->d_T, 2)) || (rc = dalloc(&L->d_dir, 4)))  // (k_lines_exists2 probes T[0..1] / dir[0..1] even for a run without regions)
        return rc;
    GFFX_HIP_TRY(hipMemset(L->d_T, 0, 2 * sizeof(uint4)));
    GFFX_HIP_TRY(hipMemset(L->d_dir, 0, 16));
    L->cap_work = kWorkFront;
    L->d_err = L->d_work, L->d_cnt = L->d_work + 4;
    GFFX_HIP_TRY(hipHostMalloc((void **)&L->h_cnt, 64, hipHostMallocCoherent | hipHostMallocMapped));
    L->h_cnt[0] = L->h_cnt[1] = L->h_cnt[2] = L->h_cnt[3] = 0;
    GFFX_HIP_TRY(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&L->ev_a, &L->ev_b, &L->ev_p0, &L->ev_p1}) GFFX_HIP_TRY(hipEventCreate(e));
    if (n_lines) {
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_seq, seq, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_start, start, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_end, end, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipStreamSynchronize(L->stream));
    }
    *out = L.release();
    return GFFX_OK;
}


static int lines_reserve(gffx_hip_lines *L, uint64_t nq, uint32_t n_seq) {
    int rc;
    if (nq > L->cap_q) {
        const uint64_t cap = nq + nq / 8 + 1024;
        const size_t blocks = (size_t)((cap + kScanBlock - 1) / kScanBlock) + 2;
        if ((rc = regrow(&L->d_rec_a, 3 * cap)) || (rc = regrow(&L->d_rec_b, 3 * cap)) || (rc = regrow(&L->d_T, cap + 1)) ||
            (rc = regrow(&L->d_carry, 9 * (blocks + 1))))
            return rc;
        L->cap_q = cap;
        L->cap_blocks = blocks;
        L->cap_dir = 0;
    }
    // <= 2 bins per region (>= 16 per seqid, + 1 closing word) and the power-of-two rounding of the bin width never adds
    const uint64_t want_dir = 2 * L->cap_q + 17ull * (n_seq + 1) + 64;
    if (want_dir > L->cap_dir) {
        if ((rc = regrow(&L->d_dir, want_dir))) return rc;
        L->cap_dir = want_dir;
    }
    if (n_seq + 1 > L->cap_seq) {
        const uint64_t cap = n_seq + 1 + 64;
        if ((rc = regrow(&L->d_qoff, cap + 1)) || (rc = regrow(&L->d_meta, cap))) return rc;
        L->cap_seq = cap;
    }
    // d_err (4 words) and d_cnt (4) sit in front of the sort's work space: ONE memset clears them and the sort's head
    const uint64_t want_work = kWorkFront + DeviceSort::work_words(L->cap_q, kSortPassesMax);
    if (want_work > L->cap_work) {
        if ((rc = regrow(&L->d_work, want_work))) return rc;
        L->cap_work = want_work;
        L->d_err = L->d_work, L->d_cnt = L->d_work + 4;
    }
    return GFFX_OK;
}

template <int MODE>
static void launch_lines(gffx_hip_lines *L, const RegionsView &rv) {
    const LinesView lv{L->d_seq, L->d_start, L->d_end, (unsigned long long)L->n};
    const unsigned blocks = (unsigned)((L->n + 255) / 256);
    if constexpr (MODE == GFFX_MODE_OVERLAP) {
        if (rv.n_deg) {  // regions with start > end: the one-line-per-thread kernel with their clauses
            if (rv.n_seq <= kMetaLds)
                hipLaunchKernelGGL((k_lines_exists<MODE, true, true>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
            else
                hipLaunchKernelGGL((k_lines_exists<MODE, false, true>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
            return;
        }
    }
    {
        const unsigned blocks2 = (unsigned)((L->n + 256 * kLinesPerThread - 1) / (256 * kLinesPerThread));
        if (rv.n_seq <= kMetaLds)
            hipLaunchKernelGGL((k_lines_exists2<MODE, true>), dim3(blocks2), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        else
            hipLaunchKernelGGL((k_lines_exists2<MODE, false>), dim3(blocks2), dim3(256), 0, L->stream, lv, rv, L->d_keep);
    }
}

// Region tables from the records in d_rec_a (AoS {seqid, qs, qe}, any order), then k_lines_exists.
static int lines_run(gffx_hip_lines *L, uint64_t nq, uint32_t n_seq, int mode, uint8_t *keep_host) {
    const unsigned long long n = nq;
    int seq_bytes = 1;
    while (seq_bytes < 4 && (n_seq > (1u << (8 * seq_bytes)))) seq_bytes++;
    SortPlan p1{}, p2{};  // (seqid, start) for the records; (seqid, end) for the degenerate pairs {seqid, end}
    for (int b = 0; b < 4; ++b) p1.word[p1.n_passes] = 1, p1.shift[p1.n_passes++] = (uint8_t)(8 * b);
    for (int b = 0; b < seq_bytes; ++b) p1.word[p1.n_passes] = 0, p1.shift[p1.n_passes++] = (uint8_t)(8 * b);
    p2 = p1;
    GFFX_HIP_TRY(hipEventRecord(L->ev_p0, L->stream));
    GFFX_HIP_TRY(hipMemsetAsync(L->d_work, 0, (kWorkFront + (nq ? DeviceSort::head_words(p1.n_passes) : 0)) * 4, L->stream));  // d_err, d_cnt, the sort's head
    L->d_de = nullptr;
    L->last_n_deg = 0;
    L->last_deg_known = false;
    if (nq) {
        const uint32_t n_blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
        const size_t stride = L->cap_blocks + 1;
        uint32_t *c = L->d_carry;
        const BlockCarry C{c, c + stride, c + 2 * stride, c + 3 * stride, c + 4 * stride, c + 5 * stride, c + 6 * stride, c + 7 * stride, c + 8 * stride};
        uint32_t *s1 = nullptr;
        // Only Overlap looks at the ends of the regions with start > end, and how many there are decides what is launched
        // after the table kernels: the histogram kernel counts them, and the host reads the count while the passes run.
        const bool want_deg = mode == GFFX_MODE_OVERLAP;
        const uint32_t note_seq = ++L->note_seq;
        // (h_cnt + 2: the sort's own note -- does the mixed-radix top digit of (seqid, start >> 24) fit 256 values?  Then the sort is
        //  four passes instead of five: radix_sort.hpp)
        int rc = DeviceSort::run<3>(L->stream, L->d_rec_a, L->d_rec_b, n, p1, n_seq, L->d_work + kWorkFront, L->d_err, &s1,
                                    want_deg ? SortNote{L->d_cnt + 2, L->h_cnt, note_seq} : SortNote{nullptr, nullptr, 0}, true, L->h_cnt + 2, note_seq,
                                    &L->last_sort_passes);
        if (rc) return rc;
        uint32_t *other = s1 == L->d_rec_a ? L->d_rec_b : L->d_rec_a;
        hipLaunchKernelGGL(k_b_local, dim3(n_blocks), dim3(kLocalThreads), 0, L->stream, s1, n, n_seq, L->d_T, L->d_qoff, C);
        hipLaunchKernelGGL(k_b_carry, dim3(1), dim3(kScanBlock), 0, L->stream, n_blocks, C, L->d_qoff, L->d_T, (uint32_t)n, n_seq, L->d_meta, L->d_cnt, (unsigned long long)L->cap_dir);
        hipLaunchKernelGGL(k_b_finish, dim3(n_blocks), dim3(kLocalThreads), 0, L->stream, s1, n, n_seq, L->d_T, C, L->d_meta, L->d_dir, other, L->d_cnt);
        GFFX_HIP_TRY(hipGetLastError());
        if (want_deg) {
            // (posted by the histogram kernel ~0.1 ms ago; should the note never arrive, the stream itself is the clock)
            const auto t0 = std::chrono::steady_clock::now();
            bool posted = false;
            unsigned long long word = 0;
            while (!(posted = (uint32_t)((word = __atomic_load_n(reinterpret_cast<unsigned long long *>(L->h_cnt), __ATOMIC_ACQUIRE)) >> 32) == note_seq) &&
                   std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2))
                std::this_thread::sleep_for(std::chrono::microseconds(20));  // (a queued stream would otherwise burn a quota CPU)
            uint32_t n_deg = (uint32_t)word;
            if (!posted) {
                GFFX_HIP_TRY(hipStreamSynchronize(L->stream));
                GFFX_HIP_TRY(hipMemcpy(&n_deg, L->d_cnt + 2, 4, hipMemcpyDeviceToHost));
            }
            L->last_n_deg = n_deg;
            L->last_deg_known = true;
            if (n_deg) {
                uint32_t *sd = nullptr;
                if ((rc = DeviceSort::run<2>(L->stream, other, s1, n_deg, p2, n_seq, L->d_work + kWorkFront, L->d_err, &sd))) return rc;
                if (L->cap_dird < L->cap_dir) {  // (only a run with degenerate regions ever pays for their directory)
                    GFFX_HIP_TRY(hipStreamSynchronize(L->stream));
                    if ((rc = regrow(&L->d_dird, L->cap_dir))) return rc;
                    L->cap_dird = L->cap_dir;
                }
                hipLaunchKernelGGL(k_b_deg_ranges, dim3((n_deg + 255) / 256), dim3(256), 0, L->stream, sd, n_deg, n_seq, L->d_meta, L->d_dird);
                GFFX_HIP_TRY(hipGetLastError());
                L->d_de = sd;
            }
        }
    } else {
        GFFX_HIP_TRY(hipMemsetAsync(L->d_meta, 0, (size_t)std::max<uint32_t>(n_seq, 1) * sizeof(SeqMeta), L->stream));
        GFFX_HIP_TRY(hipMemsetAsync(L->d_qoff, 0, ((size_t)n_seq + 1) * 4, L->stream));
    }
    GFFX_HIP_TRY(hipEventRecord(L->ev_p1, L->stream));
    if (L->n) {
        const RegionsView rv{L->d_T, L->d_dir, L->d_meta, L->d_de, L->d_dird, n_seq, L->last_n_deg};
        GFFX_HIP_TRY(hipEventRecord(L->ev_a, L->stream));
        if (mode == GFFX_MODE_CONTAINED)
            launch_lines<GFFX_MODE_CONTAINED>(L, rv);
        else if (mode == GFFX_MODE_CONTAINS_REGION)
            launch_lines<GFFX_MODE_CONTAINS_REGION>(L, rv);
        else
            launch_lines<GFFX_MODE_OVERLAP>(L, rv);
        GFFX_HIP_TRY(hipGetLastError());
        GFFX_HIP_TRY(hipEventRecord(L->ev_b, L->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(keep_host, L->d_keep, L->n, hipMemcpyDeviceToHost, L->stream));
    }
    uint32_t h_err[4] = {0, 0, 0, 0};
    GFFX_HIP_TRY(hipMemcpyAsync(h_err, L->d_err, 16, hipMemcpyDeviceToHost, L->stream));
    GFFX_HIP_TRY(hipStreamSynchronize(L->stream));
    float ms = 0.f;
    if (L->n && hipEventElapsedTime(&ms, L->ev_a, L->ev_b) == hipSuccess) L->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, L->ev_p0, L->ev_p1) == hipSuccess) L->last_prep_ms = ms;
    L->last_nq = nq;
    L->last_n_seq = n_seq;
    if (h_err[0] & 2u)
        return fail(GFFX_E_CHR_RANGE, "gffx_hip_lines_test: a region has chr >= %u", n_seq);
    if (h_err[0] & 4u) return fail(GFFX_E_HIP, "gffx_hip_lines_test: the device sort timed out waiting for an earlier tile");
    return GFFX_OK;
}

static int lines_check(gffx_hip_lines *L, const void *regions, uint64_t nq, int mode, const uint8_t *keep_host, const char *who) {
    if (!L) return fail(GFFX_E_INVALID, "%s: lines is NULL", who);
    if (mode < 0 || mode > 2) return fail(GFFX_E_INVALID, "%s: bad mode %d", who, mode);
    if (nq && !regions) return fail(GFFX_E_INVALID, "%s: regions is NULL", who);
    if (L->n && !keep_host) return fail(GFFX_E_INVALID, "%s: keep_host is NULL", who);
    return GFFX_OK;
}

extern "C" int gffx_hip_lines_test(gffx_hip_lines *L, const uint32_t *regions, uint64_t nq,
                                   uint32_t n_seq, int mode, uint8_t *keep_host) {
    int rc = lines_check(L, regions, nq, mode, keep_host, "gffx_hip_lines_test");
    if (rc) return rc;
    GFFX_HIP_TRY(hipSetDevice(L->device));
    if ((rc = lines_reserve(L, nq, n_seq))) return rc;
    if (nq) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, regions, nq * 12, hipMemcpyHostToDevice, L->stream));
    return lines_run(L, nq, n_seq, mode, keep_host);
}

// the same with the regions already in HBM (AoS triples, e.g. the batch Join A uploaded: gffx_hip_batch_device_regions)
extern "C" int gffx_hip_lines_test_device(gffx_hip_lines *L, const uint32_t *d_regions, uint64_t nq,
                                          uint32_t n_seq, int mode, uint8_t *keep_host) {
    int rc = lines_check(L, d_regions, nq, mode, keep_host, "gffx_hip_lines_test_device");
    if (rc) return rc;
    GFFX_HIP_TRY(hipSetDevice(L->device));
    if ((rc = lines_reserve(L, nq, n_seq))) return rc;
    if (nq) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, d_regions, nq * 12, hipMemcpyDeviceToDevice, L->stream));
    return lines_run(L, nq, n_seq, mode, keep_host);
}

extern "C" int gffx_hip_lines_test_store(gffx_hip_lines *L, const gffx_hip_regions *R, uint32_t n_seq, int mode, uint8_t *keep_host) {
    if (!R || !R->keep_all) return fail(GFFX_E_INVALID, "gffx_hip_lines_test_store: needs a keep_all region store");
    const uint64_t nq = R->rows;
    int rc = lines_check(L, R->d, nq, mode, keep_host, "gffx_hip_lines_test_store");
    if (rc) return rc;
    if (R->device != L->device) return fail(GFFX_E_INVALID, "gffx_hip_lines_test_store: store and line table on different devices");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    if ((rc = lines_reserve(L, nq, n_seq))) return rc;
    for (int k = 0; k < 2; ++k)  // every append has to have landed
        if (R->pending[k]) GFFX_HIP_TRY(hipStreamWaitEvent(L->stream, R->copied[k], 0));
    if (nq) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, R->d, nq * 12, hipMemcpyDeviceToDevice, L->stream));
    return lines_run(L, nq, n_seq, mode, keep_host);
}

// The region tables of the last _test, for parity tests: q_off (n_seq + 1), then QS, PM, SM, CD (nq each).
extern "C" int gffx_hip_lines_copy_tables(gffx_hip_lines *L, uint64_t *q_off, uint32_t *qs, uint32_t *pm, uint32_t *sm, uint32_t *cd) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_tables: lines is NULL");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    const uint64_t nq = L->last_nq;
    if (q_off) {
        std::vector<uint32_t> off(L->last_n_seq + 1, 0);
        GFFX_HIP_TRY(hipMemcpy(off.data(), L->d_qoff, off.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < off.size(); i++) q_off[i] = off[i];
    }
    if (nq && (qs || pm || sm || cd)) {
        std::vector<uint4> t(nq);
        GFFX_HIP_TRY(hipMemcpy(t.data(), L->d_T, nq * sizeof(uint4), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < nq; i++) {
            if (qs) qs[i] = t[i].x;
            if (pm) pm[i] = t[i].y;
            if (sm) sm[i] = t[i].z;
            if (cd) cd[i] = t[i].w;
        }
    }
    return GFFX_OK;
}
// ... the bin directory over QS: d_off (n_seq + 1), shift_nb (2 per seqid), dir_qs (d_off[n_seq] words) ...
extern "C" int gffx_hip_lines_copy_dirs(gffx_hip_lines *L, uint64_t *d_off, uint32_t *shift_nb, uint32_t *dir_qs) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_dirs: lines is NULL");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    std::vector<SeqMeta> meta(L->last_n_seq);
    if (L->last_n_seq) GFFX_HIP_TRY(hipMemcpy(meta.data(), L->d_meta, meta.size() * sizeof(SeqMeta), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t c = 0; c < L->last_n_seq; c++) {
        if (d_off) d_off[c] = meta[c].d_base;
        if (shift_nb) shift_nb[2 * c] = meta[c].shift, shift_nb[2 * c + 1] = meta[c].nb;
        total = meta[c].d_base + (meta[c].q_hi > meta[c].q_lo ? meta[c].nb + 1 : 0);
    }
    if (!L->last_nq) total = 0;
    if (d_off) d_off[L->last_n_seq] = total;
    if (dir_qs && total) GFFX_HIP_TRY(hipMemcpy(dir_qs, L->d_dir, total * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
// ... and, after an Overlap-mode _test, the degenerate regions (start > end): *n_deg of them, dq_off (n_seq + 1) and their
// ends sorted per seqid (de: *n_deg words; call once with de = NULL to learn the size)
extern "C" int gffx_hip_lines_copy_degenerate(gffx_hip_lines *L, uint64_t *n_deg, uint64_t *dq_off, uint32_t *de) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_degenerate: lines is NULL");
    if (!L->last_deg_known && L->last_nq) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_degenerate: the last _test was not in Overlap mode");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    if (n_deg) *n_deg = L->last_n_deg;
    if (dq_off) {
        std::vector<SeqMeta> meta(L->last_n_seq);
        if (L->last_n_seq && L->last_nq) GFFX_HIP_TRY(hipMemcpy(meta.data(), L->d_meta, meta.size() * sizeof(SeqMeta), hipMemcpyDeviceToHost));
        uint64_t run = 0;
        for (uint32_t c = 0; c < L->last_n_seq; c++) {
            dq_off[c] = run;
            if (L->last_n_deg && L->last_nq) run += meta[c].dq_hi - meta[c].dq_lo;
        }
        dq_off[L->last_n_seq] = run;
    }
    if (de && L->last_n_deg) {
        std::vector<uint32_t> pairs(2 * (size_t)L->last_n_deg);
        GFFX_HIP_TRY(hipMemcpy(pairs.data(), L->d_de, pairs.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < L->last_n_deg; i++) de[i] = pairs[2 * (size_t)i + 1];
    }
    return GFFX_OK;
}
extern "C" double gffx_hip_lines_last_prep_ms(const gffx_hip_lines *L) { return L ? L->last_prep_ms : 0.0; }
extern "C" int gffx_hip_lines_last_sort_passes(const gffx_hip_lines *L) { return L ? L->last_sort_passes : 0; }

extern "C" double gffx_hip_lines_last_kernel_ms(const gffx_hip_lines *L) { return L ? L->last_kernel_ms : 0.0; }
