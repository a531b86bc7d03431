// join_b.hip -- Join B: "does ANY region of the line's seqid satisfy the predicate" for every GFF
// line of the hit blocks (reference: commands/intersect.rs:500-521, the numeric core of
// gff_line_overlaps_queries; called for every non-comment line of every hit block, :284-321).
//
// The reference scans all regions of the seqid linearly per line: O(lines x regions_of_seqid),
// its dominant cost.  Here the regions of a seqid are sorted by start once per run and three
// monotone helper arrays make each mode a couple of binary searches.  With (s, e) the RAW
// column-4/5 integers of the line (1-based closed, no swap: intersect.rs:475-489) and the regions'
// raw (qs, qe) (no s<e check: intersect.rs:223-225):
//   Contained       exists q: s >= qs && e <= qe   <=>  PM(s) >= e      PM(x) = max{qe : qs <= x}
//   ContainsRegion  exists q: s <= qs && e >= qe   <=>  SM(s) <= e      SM(x) = min{qe : qs >= x}
//   Overlap  (qs<=s<=qe) || (qs<=e<=qe) || (s<=qs<=e) || (s<=qe<=e)     (intersect.rs:512-515)
//                                                  <=>  PM(s) >= s || PM(e) >= e
//                                                       || some qs in [s,e] || some qe in [s,e]
// Each clause is a conjunction of two comparisons on one region, so the rewrite is exact for every
// input, including degenerate regions (qs > qe) and lines (s > e); tests check it against the
// oracle's literal scan.  Every search first narrows to one bin of a per-seqid directory over the
// sorted array (built with the sort, ~2 bins per region), so it touches ~3 words instead of ~17.  No invert here: intersect.rs:232-240 has no such parameter.
//
// HBM layout: line table SoA {seq, start, end} u32 x n_lines, file order (neighbouring lanes =
// neighbouring lines = nearby coordinates -> the searches of a wave walk the same cache lines);
// per run: q_off[n_seq+1], QS (sorted starts), PM (prefix max of ends), SM (suffix min of ends),
// QE (sorted ends), each u32 x n_regions.  One thread per line, one byte out.
// Roofline bound: HBM; algorithmic bytes per line: 12 in + 1 out.
#include <algorithm>
#include <atomic>
#include <thread>
#include <memory>
#include <numeric>
#include <vector>

#include "gffx_device.hpp"

namespace gffx {

struct LinesView {
    const uint32_t *seq, *start, *end;
    unsigned long long n;
};
struct RegionsView {
    const unsigned long long *q_off;  // n_seq + 1
    const uint32_t *qs, *pm, *sm, *qe;
    // Directories over the sorted starts and the sorted ends of every seqid: dir[d_off[c] + b] = first position
    // whose value >= b << shift(c), for b = 0..nb(c) (the last one = the seqid's end).  A search for x only has
    // to look inside [dir[b], dir[b+1]) with b = x >> shift -- usually zero or one element instead of a 17-step
    // binary search over all regions of the seqid.  nullptr: no directory (plain binary search).
    const uint32_t *dir_qs, *dir_qe;
    const unsigned long long *d_off;  // n_seq + 1
    const uint2 *d_meta;              // per seqid {shift, nb}
    uint32_t n_seq;
};

// narrow [lo, hi) to the directory bin of x
__device__ __forceinline__ void dir_narrow(const uint32_t *dir, const RegionsView &R, uint32_t seq, uint32_t x,
                                           unsigned long long &lo, unsigned long long &hi) {
    if (!dir) return;
    const uint2 m = R.d_meta[seq];
    const uint32_t b = x >> m.x;
    if (b >= m.y) {
        lo = hi;  // beyond the largest value of the seqid
        return;
    }
    const uint32_t *d = dir + R.d_off[seq] + b;
    lo = d[0];
    hi = d[1];
}

// first index in [lo, hi) with a[i] >= x
__device__ __forceinline__ unsigned long long lower_bound_u32(const uint32_t *a, unsigned long long lo,
                                                              unsigned long long hi, uint32_t x) {
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (a[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
// first index in [lo, hi) with a[i] > x
__device__ __forceinline__ unsigned long long upper_bound_u32(const uint32_t *a, unsigned long long lo,
                                                              unsigned long long hi, uint32_t x) {
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (a[mid] <= x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_lines_exists(LinesView L, RegionsView R, uint8_t *keep) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n) return;
    const uint32_t seq = L.seq[i];
    uint8_t k = 0;
    if (seq < R.n_seq) {
        const unsigned long long lo = R.q_off[seq], hi = R.q_off[seq + 1];
        if (hi > lo) {  // a seqid without regions has no map entry (intersect.rs:495-498)
            const uint32_t s = L.start[i], e = L.end[i];
            auto ub_qs = [&](uint32_t x) {
                unsigned long long a = lo, b = hi;
                dir_narrow(R.dir_qs, R, seq, x, a, b);
                return upper_bound_u32(R.qs, a, b, x);
            };
            auto lb_qs = [&](uint32_t x) {
                unsigned long long a = lo, b = hi;
                dir_narrow(R.dir_qs, R, seq, x, a, b);
                return lower_bound_u32(R.qs, a, b, x);
            };
            auto lb_qe = [&](uint32_t x) {
                unsigned long long a = lo, b = hi;
                dir_narrow(R.dir_qe, R, seq, x, a, b);
                return lower_bound_u32(R.qe, a, b, x);
            };
            auto ub_qe = [&](uint32_t x) {
                unsigned long long a = lo, b = hi;
                dir_narrow(R.dir_qe, R, seq, x, a, b);
                return upper_bound_u32(R.qe, a, b, x);
            };
            if (MODE == GFFX_MODE_CONTAINED) {
                const unsigned long long u = ub_qs(s);  // regions with qs <= s
                k = (u > lo && R.pm[u - 1] >= e) ? 1 : 0;
            } else if (MODE == GFFX_MODE_CONTAINS_REGION) {
                const unsigned long long l = lb_qs(s);  // regions with qs >= s
                k = (l < hi && R.sm[l] <= e) ? 1 : 0;
            } else {
                const unsigned long long us = ub_qs(s);
                bool any = us > lo && R.pm[us - 1] >= s;  // qs <= s <= qe
                if (!any) {
                    const unsigned long long ue = ub_qs(e);
                    any = ue > lo && R.pm[ue - 1] >= e;  // qs <= e <= qe
                    if (!any && s <= e) {
                        const unsigned long long ls = lb_qs(s);
                        any = ls < ue;  // some qs in [s, e]
                        if (!any) {
                            const unsigned long long a = lb_qe(s);
                            const unsigned long long b = ub_qe(e);
                            any = a < b;  // some qe in [s, e]
                        }
                    }
                }
                k = any ? 1 : 0;
            }
        }
    }
    keep[i] = k;
}

}  // namespace gffx

using namespace gffx;

struct gffx_hip_lines {
    int device = 0;
    uint64_t n = 0;
    uint32_t *d_seq = nullptr, *d_start = nullptr, *d_end = nullptr;
    uint8_t *d_keep = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;  // bracket k_lines_exists of the last _test
    double last_kernel_ms = 0.0;
};

template <typename T>
static int dalloc(T **p, size_t n) {
    *p = nullptr;
    GFFX_HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GFFX_OK;
}

extern "C" void gffx_hip_lines_destroy(gffx_hip_lines *L) {
    if (!L) return;
    (void)hipSetDevice(L->device);
    (void)hipFree(L->d_seq);
    (void)hipFree(L->d_start);
    (void)hipFree(L->d_end);
    (void)hipFree(L->d_keep);
    if (L->ev_a) (void)hipEventDestroy(L->ev_a);
    if (L->ev_b) (void)hipEventDestroy(L->ev_b);
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
        (rc = dalloc(&L->d_end, n_lines)) || (rc = dalloc(&L->d_keep, n_lines)))
        return rc;
    GFFX_HIP_TRY(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
    GFFX_HIP_TRY(hipEventCreate(&L->ev_a));
    GFFX_HIP_TRY(hipEventCreate(&L->ev_b));
    if (n_lines) {
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_seq, seq, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_start, start, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_end, end, n_lines * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_HIP_TRY(hipStreamSynchronize(L->stream));
    }
    *out = L.release();
    return GFFX_OK;
}

extern "C" int gffx_hip_lines_test(gffx_hip_lines *L, const uint32_t *regions, uint64_t nq,
                                   uint32_t n_seq, int mode, uint8_t *keep_host) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_test: lines is NULL");
    if (mode < 0 || mode > 2) return fail(GFFX_E_INVALID, "gffx_hip_lines_test: bad mode %d", mode);
    if (nq && !regions) return fail(GFFX_E_INVALID, "gffx_hip_lines_test: regions is NULL");
    if (L->n && !keep_host) return fail(GFFX_E_INVALID, "gffx_hip_lines_test: keep_host is NULL");
    for (uint64_t i = 0; i < nq; i++)
        if (regions[3 * i] >= n_seq)
            return fail(GFFX_E_CHR_RANGE, "gffx_hip_lines_test: region %llu has chr %u >= %u",
                        (unsigned long long)i, regions[3 * i], n_seq);
    GFFX_HIP_TRY(hipSetDevice(L->device));
    // Region preparation: bucket by seqid, sort by start, prefix max / suffix min of the ends,
    // sorted ends.  Host-side for now (std::sort); the device radix sort replaces it.
    std::vector<unsigned long long> q_off(n_seq + 1, 0);
    for (uint64_t i = 0; i < nq; i++) q_off[regions[3 * i] + 1]++;
    for (uint32_t c = 0; c < n_seq; c++) q_off[c + 1] += q_off[c];
    std::vector<uint64_t> key(nq);  // (qs << 32 | qe) grouped by seqid
    {
        std::vector<unsigned long long> cur(q_off.begin(), q_off.end() - 1);
        for (uint64_t i = 0; i < nq; i++)
            key[cur[regions[3 * i]]++] = ((uint64_t)regions[3 * i + 1] << 32) | regions[3 * i + 2];
    }
    std::vector<uint32_t> qs(nq), pm(nq), sm(nq), qe(nq);
    auto prep_seq = [&](uint32_t c) {
        const uint64_t lo = q_off[c], hi = q_off[c + 1];
        if (hi == lo) return;
        std::sort(key.begin() + lo, key.begin() + hi);
        uint32_t m = 0;
        for (uint64_t i = lo; i < hi; i++) {
            qs[i] = (uint32_t)(key[i] >> 32);
            qe[i] = (uint32_t)key[i];
            m = std::max(m, qe[i]);
            pm[i] = m;
        }
        m = UINT32_MAX;
        for (uint64_t i = hi; i-- > lo;) {
            m = std::min(m, qe[i]);
            sm[i] = m;
        }
        std::sort(qe.begin() + lo, qe.begin() + hi);
    };
    {  // the seqids are independent: host threads take them largest first (80 ms -> ~10 ms per 1 M regions)
        std::vector<uint32_t> by_size(n_seq);
        for (uint32_t c = 0; c < n_seq; c++) by_size[c] = c;
        std::sort(by_size.begin(), by_size.end(),
                  [&](uint32_t a, uint32_t b) { return q_off[a + 1] - q_off[a] > q_off[b + 1] - q_off[b]; });
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned n_thr = nq < 50000 ? 1u : std::max(1u, std::min({hw ? hw : 1u, 16u, n_seq}));
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (;;) {
                const uint32_t k = next.fetch_add(1);
                if (k >= n_seq) return;
                prep_seq(by_size[k]);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_thr; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    // directories over the sorted starts / ends (see RegionsView); positions are u32: skipped for >= 2^32 regions
    std::vector<unsigned long long> dir_off(n_seq + 1, 0);
    std::vector<uint2> dir_meta(n_seq, make_uint2(0, 0));
    std::vector<uint32_t> dir_qs, dir_qe;
    const bool use_dir = nq > 0 && nq < 0xFFFFFFFFull;
    if (use_dir) {
        for (uint32_t c = 0; c < n_seq; c++) {
            const uint64_t lo = q_off[c], hi = q_off[c + 1];
            dir_off[c + 1] = dir_off[c];
            if (hi == lo) continue;
            const uint32_t vmax = std::max(qs[hi - 1], qe[hi - 1]);
            const uint64_t budget = std::max<uint64_t>(2 * (hi - lo), 16);
            uint32_t shift = 0;
            while ((((uint64_t)vmax >> shift) + 1) > budget) shift++;
            const uint32_t nb = (vmax >> shift) + 1;
            dir_meta[c] = make_uint2(shift, nb);
            uint64_t ps = lo, pe = lo;
            for (uint32_t b = 0; b < nb; b++) {
                const uint64_t edge = (uint64_t)b << shift;
                while (ps < hi && qs[ps] < edge) ps++;
                while (pe < hi && qe[pe] < edge) pe++;
                dir_qs.push_back((uint32_t)ps);
                dir_qe.push_back((uint32_t)pe);
            }
            dir_qs.push_back((uint32_t)hi);
            dir_qe.push_back((uint32_t)hi);
            dir_off[c + 1] = dir_qs.size();
        }
    }
    unsigned long long *d_off = nullptr, *d_doff = nullptr;
    uint32_t *d_q = nullptr, *d_dir = nullptr;
    uint2 *d_dmeta = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_off);
        (void)hipFree(d_q);
        (void)hipFree(d_doff);
        (void)hipFree(d_dir);
        (void)hipFree(d_dmeta);
    };
    int rc;
    if ((rc = dalloc(&d_off, n_seq + 1)) || (rc = dalloc(&d_q, 4 * nq)) || (rc = dalloc(&d_doff, n_seq + 1)) ||
        (rc = dalloc(&d_dir, 2 * dir_qs.size())) || (rc = dalloc(&d_dmeta, n_seq))) {
        cleanup();
        return rc;
    }
#define GFFX_TRY_C(expr)                                                                       \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            cleanup();                                                                         \
            return fail(GFFX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));            \
        }                                                                                      \
    } while (0)
    GFFX_TRY_C(hipMemcpyAsync(d_off, q_off.data(), (n_seq + 1) * 8, hipMemcpyHostToDevice, L->stream));
    if (nq) {
        GFFX_TRY_C(hipMemcpyAsync(d_q, qs.data(), nq * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_q + nq, pm.data(), nq * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_q + 2 * nq, sm.data(), nq * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_q + 3 * nq, qe.data(), nq * 4, hipMemcpyHostToDevice, L->stream));
    }
    if (use_dir) {
        GFFX_TRY_C(hipMemcpyAsync(d_doff, dir_off.data(), (n_seq + 1) * 8, hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_dmeta, dir_meta.data(), n_seq * sizeof(uint2), hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_dir, dir_qs.data(), dir_qs.size() * 4, hipMemcpyHostToDevice, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(d_dir + dir_qs.size(), dir_qe.data(), dir_qe.size() * 4, hipMemcpyHostToDevice, L->stream));
    }
    if (L->n) {
        LinesView lv{L->d_seq, L->d_start, L->d_end, (unsigned long long)L->n};
        RegionsView rv{d_off, d_q, d_q + nq, d_q + 2 * nq, d_q + 3 * nq,
                       use_dir ? d_dir : nullptr, use_dir ? d_dir + dir_qs.size() : nullptr, d_doff, d_dmeta, n_seq};
        const unsigned blocks = (unsigned)((L->n + 255) / 256);
        GFFX_TRY_C(hipEventRecord(L->ev_a, L->stream));
        if (mode == GFFX_MODE_CONTAINED)
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_CONTAINED>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        else if (mode == GFFX_MODE_CONTAINS_REGION)
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_CONTAINS_REGION>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        else
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_OVERLAP>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        GFFX_TRY_C(hipGetLastError());
        GFFX_TRY_C(hipEventRecord(L->ev_b, L->stream));
        GFFX_TRY_C(hipMemcpyAsync(keep_host, L->d_keep, L->n, hipMemcpyDeviceToHost, L->stream));
    }
    GFFX_TRY_C(hipStreamSynchronize(L->stream));
    if (L->n) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, L->ev_a, L->ev_b) == hipSuccess) L->last_kernel_ms = ms;
    }
#undef GFFX_TRY_C
    cleanup();
    return GFFX_OK;
}

extern "C" double gffx_hip_lines_last_kernel_ms(const gffx_hip_lines *L) { return L ? L->last_kernel_ms : 0.0; }
