// join_b.hip -- Join B: "does ANY region of the line's seqid satisfy the predicate" for every GFF
// line of the hit blocks (reference: commands/intersect.rs:500-521, the numeric core of
// gff_line_overlaps_queries; called for every non-comment line of every hit block, :284-321).
//
// The reference scans all regions of the seqid linearly per line: O(lines x regions_of_seqid),
// its dominant cost.  Here the regions are radix-sorted ON THE DEVICE by (seqid, start, end) once per run (radix_sort.hpp) and three
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
#include "radix_sort.hpp"
#include "regions_store.hpp"

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


// ---- region tables on the device (what the reference builds per run as `query_ivmap`, intersect.rs:621-633) --------------
// The records {seqid, qs, qe} are radix-sorted by (seqid, qs, qe) (radix_sort.hpp); q_off, QS, PM, SM come from that order,
// QE from a second sort by (seqid, qe); the two bin directories are filled from the sorted arrays.  All kernels below are
// one thread per region.

// q_off[c] = first sorted position of seqid c (n_seq + 1 entries): every boundary thread fills the seqids it skips over
__global__ __launch_bounds__(256) void k_b_offsets(const uint32_t *rec, unsigned long long n, uint32_t n_seq, unsigned long long *q_off) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = min(rec[3 * i], n_seq);  // (a seqid out of range is reported by the sort's histogram kernel)
    const long long prev = i ? (long long)min(rec[3 * (i - 1)], n_seq) : -1;
    for (long long c = prev + 1; c <= (long long)s; ++c) q_off[c] = i;
    if (i + 1 == n)
        for (uint32_t c = s + 1; c <= n_seq; ++c) q_off[c] = n;
}

// column w of the sorted records; head[i] = 1 where a seqid's run starts (forward) / ends (backward scans)
__global__ __launch_bounds__(256) void k_b_columns(const uint32_t *rec, unsigned long long n, uint32_t *qs, uint32_t *qe_by_qs,
                                                   uint8_t *head_fwd, uint8_t *head_bwd) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = rec[3 * i];
    qs[i] = rec[3 * i + 1];
    qe_by_qs[i] = rec[3 * i + 2];
    head_fwd[i] = (i == 0 || rec[3 * (i - 1)] != s) ? 1 : 0;
    head_bwd[i] = (i + 1 == n || rec[3 * (i + 1)] != s) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_b_column2(const uint32_t *rec, unsigned long long n, uint32_t *out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rec[3 * i + 2];
}

// Segmented inclusive scan, one level: MAXOP ? running max : running min; BACKWARD scans from the end (element x of the
// scan is array element n-1-x).  head[x'] = 1 starts a new segment.  Blocks of 1024 elements; the block's last value and
// "has a head" flag go to agg / agg_head (the next level scans those), k_b_scan_apply folds the carries back in.
constexpr int kScanBlock = 1024;
template <bool MAXOP>
__device__ __forceinline__ uint32_t scan_op(uint32_t a, uint32_t b) {
    return MAXOP ? max(a, b) : min(a, b);
}
template <bool MAXOP, bool BACKWARD>
__global__ __launch_bounds__(kScanBlock) void k_b_scan_local(const uint32_t *val, const uint8_t *head, unsigned long long n, uint32_t *out,
                                                             uint32_t *agg, uint8_t *agg_head) {
    __shared__ uint32_t s_v[kScanBlock / 64];
    __shared__ uint32_t s_f[kScanBlock / 64];
    const unsigned long long x = (unsigned long long)blockIdx.x * kScanBlock + threadIdx.x;
    const bool live = x < n;
    const unsigned long long i = BACKWARD ? n - 1 - (live ? x : 0) : x;
    uint32_t v = live ? val[i] : (MAXOP ? 0u : 0xFFFFFFFFu);
    uint32_t f = live ? head[i] : 1u;
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
    // carry of the preceding waves of the block: the value of the segment still open where this wave starts
    // (serial over <= 15 wave totals; a wave that holds a head restarts it)
    uint32_t cv = 0, cf = 0;
    for (int w = 0; w < wave; ++w) {
        cv = (w == 0 || s_f[w]) ? s_v[w] : scan_op<MAXOP>(cv, s_v[w]);
        cf |= s_f[w];
    }
    if (wave > 0 && !f) v = scan_op<MAXOP>(cv, v);
    f |= cf;
    if (live) out[i] = v;
    if (threadIdx.x == kScanBlock - 1 || x + 1 == n) {
        if (live) agg[blockIdx.x] = v, agg_head[blockIdx.x] = (uint8_t)(f ? 1 : 0);
    }
}
// fold the scanned block carries in: an element before the first head of its block continues the previous blocks' segment
template <bool MAXOP, bool BACKWARD>
__global__ __launch_bounds__(kScanBlock) void k_b_scan_apply(const uint8_t *head, unsigned long long n, uint32_t *out, const uint32_t *agg_scanned) {
    __shared__ uint32_t s_any[kScanBlock / 64];
    if (blockIdx.x == 0) return;
    const unsigned long long x = (unsigned long long)blockIdx.x * kScanBlock + threadIdx.x;
    const bool live = x < n;
    const unsigned long long i = BACKWARD ? n - 1 - (live ? x : 0) : x;
    const uint32_t f = live ? head[i] : 1u;
    // "a head at or before me inside the block": inclusive OR-scan
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(f != 0);
    const bool before_in_wave = (m & ((2ull << lane) - 1ull)) != 0;
    if (lane == 0) s_any[wave] = m != 0;
    __syncthreads();
    bool seen = before_in_wave;
    for (int w = 0; w < wave; ++w) seen |= s_any[w] != 0;
    if (live && !seen) out[i] = scan_op<MAXOP>(agg_scanned[blockIdx.x - 1], out[i]);
}

// per seqid: the directory geometry {shift, nb} over max(largest start, largest end) and the directory's place d_off
__global__ __launch_bounds__(256) void k_b_dir_meta(const unsigned long long *q_off, const uint32_t *qs, const uint32_t *qe, uint32_t n_seq,
                                                    uint2 *d_meta, unsigned long long *d_off) {
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    // seqids in chunks of 256: sizes, then a serial prefix by thread 0 (n_seq is small)
    for (uint32_t c0 = 0; c0 < n_seq; c0 += 256) {
        const uint32_t c = c0 + threadIdx.x;
        uint32_t size = 0;
        if (c < n_seq) {
            const unsigned long long lo = q_off[c], hi = q_off[c + 1];
            uint2 m = make_uint2(0, 0);
            if (hi > lo) {
                const uint32_t vmax = max(qs[hi - 1], qe[hi - 1]);
                const unsigned long long budget = max(2ull * (hi - lo), 16ull);
                uint32_t shift = 0;
                while ((((unsigned long long)vmax >> shift) + 1) > budget) shift++;
                m = make_uint2(shift, (vmax >> shift) + 1);
                size = m.y + 1;
            }
            d_meta[c] = m;
        }
        __shared__ uint32_t s_size[256];
        s_size[threadIdx.x] = size;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = s_run;
            for (uint32_t k = 0; k < 256 && c0 + k < n_seq; ++k) {
                d_off[c0 + k] = run;
                run += s_size[k];
            }
            s_run = run;
            if (c0 + 256 >= n_seq) d_off[n_seq] = run;
        }
        __syncthreads();
    }
    if (n_seq == 0 && threadIdx.x == 0) d_off[0] = 0;
}

// dir[d_off[c] + b] = first sorted position of seqid c whose value >= b << shift, b = 0..nb (nb: the seqid's end)
__global__ __launch_bounds__(256) void k_b_dir_fill(const uint32_t *val, const uint32_t *rec /* seqid of element i */, unsigned long long n,
                                                    uint32_t n_seq, const unsigned long long *q_off, const unsigned long long *d_off, const uint2 *d_meta,
                                                    uint32_t *dir) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = rec[3 * i];
    if (c >= n_seq) return;  // (reported by the sort's histogram kernel)
    const unsigned long long lo = q_off[c], hi = q_off[c + 1];
    const uint2 m = d_meta[c];
    uint32_t *d = dir + d_off[c];
    const uint32_t b = val[i] >> m.x;
    const long long bprev = i > lo ? (long long)(val[i - 1] >> m.x) : -1;
    for (long long x = bprev + 1; x <= (long long)b; ++x) d[x] = (uint32_t)i;
    if (i + 1 == hi)
        for (uint32_t x = b + 1; x <= m.y; ++x) d[x] = (uint32_t)hi;
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
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;  // ... and the device preparation of the region tables
    double last_kernel_ms = 0.0, last_prep_ms = 0.0;
    // region tables of the last _test (grow-only device buffers)
    uint64_t cap_q = 0, cap_seq = 0, cap_work = 0, cap_dir = 0;
    uint32_t *d_rec_a = nullptr, *d_rec_b = nullptr, *d_rec_c = nullptr;  // 3 u32 per region: input / sort ping-pong / kept (seqid, qs, qe) order
    uint32_t *d_tab = nullptr;       // QS, PM, SM, QE, E (qe in qs order): 5 x cap_q
    uint8_t *d_head = nullptr;       // 2 x cap_q
    uint32_t *d_scan = nullptr;      // carries of the scan levels
    uint8_t *d_scan_head = nullptr;
    uint32_t *d_work = nullptr;      // sort work space
    unsigned long long *d_qoff = nullptr, *d_doff = nullptr;  // cap_seq + 1
    uint2 *d_dmeta = nullptr;
    uint32_t *d_dir = nullptr;       // 2 x cap_dir
    uint32_t *d_err = nullptr;
    uint64_t last_nq = 0;
    uint32_t last_n_seq = 0;
    bool last_dir = false;
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
    (void)hipFree(L->d_seq);
    (void)hipFree(L->d_start);
    (void)hipFree(L->d_end);
    (void)hipFree(L->d_keep);
    (void)hipFree(L->d_rec_a);
    (void)hipFree(L->d_rec_b);
    (void)hipFree(L->d_rec_c);
    (void)hipFree(L->d_tab);
    (void)hipFree(L->d_head);
    (void)hipFree(L->d_scan);
    (void)hipFree(L->d_scan_head);
    (void)hipFree(L->d_work);
    (void)hipFree(L->d_qoff);
    (void)hipFree(L->d_doff);
    (void)hipFree(L->d_dmeta);
    (void)hipFree(L->d_dir);
    (void)hipFree(L->d_err);
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
        (rc = dalloc(&L->d_end, n_lines)) || (rc = dalloc(&L->d_keep, n_lines)) || (rc = dalloc(&L->d_err, 4)))
        return rc;
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

static int lines_reserve(gffx_hip_lines *L, uint64_t nq, uint32_t n_seq, int sort_passes) {
    int rc;
    if (nq > L->cap_q) {
        const uint64_t cap = nq + nq / 8 + 1024;
        const size_t blocks = (size_t)((cap + kScanBlock - 1) / kScanBlock) + 2;
        if ((rc = regrow(&L->d_rec_a, 3 * cap)) || (rc = regrow(&L->d_rec_b, 3 * cap)) || (rc = regrow(&L->d_rec_c, 3 * cap)) ||
            (rc = regrow(&L->d_tab, 5 * cap)) || (rc = regrow(&L->d_head, 2 * cap)) || (rc = regrow(&L->d_scan, 4 * blocks)) ||
            (rc = regrow(&L->d_scan_head, 2 * blocks)))
            return rc;
        L->cap_q = cap;
        L->cap_dir = 0;
    }
    const uint64_t want_dir = 2 * L->cap_q + 17ull * (n_seq + 1) + 64;
    if (want_dir > L->cap_dir) {
        if ((rc = regrow(&L->d_dir, 2 * want_dir))) return rc;
        L->cap_dir = want_dir;
    }
    if (n_seq + 1 > L->cap_seq) {
        const uint64_t cap = n_seq + 1 + 64;
        if ((rc = regrow(&L->d_qoff, cap + 1)) || (rc = regrow(&L->d_doff, cap + 1)) || (rc = regrow(&L->d_dmeta, cap))) return rc;
        L->cap_seq = cap;
    }
    const uint64_t want_work = DeviceSort::work_words(L->cap_q, sort_passes);
    if (want_work > L->cap_work) {
        if ((rc = regrow(&L->d_work, want_work))) return rc;
        L->cap_work = want_work;
    }
    return GFFX_OK;
}

// one direction of the segmented scans: local blocks, the carries (recursively), fold back
template <bool MAXOP, bool BACKWARD>
static int seg_scan(gffx_hip_lines *L, const uint32_t *val, const uint8_t *head, uint64_t n, uint32_t *out) {
    // level 0 over the elements; level 1 over the block carries (forward from here on: the carries already are in scan order)
    const uint32_t b0 = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
    uint32_t *agg0 = L->d_scan, *agg0s = L->d_scan + b0 + 1;
    uint8_t *h0 = L->d_scan_head;
    hipLaunchKernelGGL((k_b_scan_local<MAXOP, BACKWARD>), dim3(b0), dim3(kScanBlock), 0, L->stream, val, head, (unsigned long long)n, out, agg0, h0);
    if (b0 > 1) {
        const uint32_t b1 = (b0 + kScanBlock - 1) / kScanBlock;
        uint32_t *agg1 = agg0s + b0 + 1, *agg1s = agg1 + b1 + 1;
        uint8_t *h1 = h0 + b0 + 1;
        hipLaunchKernelGGL((k_b_scan_local<MAXOP, false>), dim3(b1), dim3(kScanBlock), 0, L->stream, agg0, h0, (unsigned long long)b0, agg0s, agg1, h1);
        if (b1 > 1) {  // > 1 M blocks = > 10^9 regions never happens (the sort refuses 2^30), two carry levels are enough up to 2^30
            hipLaunchKernelGGL((k_b_scan_local<MAXOP, false>), dim3(1), dim3(kScanBlock), 0, L->stream, agg1, h1, (unsigned long long)b1, agg1s,
                               agg1s + b1 + 1, h1 + b1 + 1);
            hipLaunchKernelGGL((k_b_scan_apply<MAXOP, false>), dim3(b1), dim3(kScanBlock), 0, L->stream, h0, (unsigned long long)b0, agg0s, agg1s);
        }
        hipLaunchKernelGGL((k_b_scan_apply<MAXOP, BACKWARD>), dim3(b0), dim3(kScanBlock), 0, L->stream, head, (unsigned long long)n, out, agg0s);
    }
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

// Region tables from the records in d_rec_a (AoS {seqid, qs, qe}, any order), then k_lines_exists.
static int lines_run(gffx_hip_lines *L, uint64_t nq, uint32_t n_seq, int mode, uint8_t *keep_host) {
    const unsigned long long n = nq;
    const uint32_t g256 = (uint32_t)((n + 255) / 256);
    int seq_bytes = 1;
    while (seq_bytes < 4 && (n_seq > (1u << (8 * seq_bytes)))) seq_bytes++;
    SortPlan p1{}, p2{};
    for (int b = 0; b < 4; ++b) p1.word[p1.n_passes] = 2, p1.shift[p1.n_passes++] = (uint8_t)(8 * b);
    for (int b = 0; b < 4; ++b) p1.word[p1.n_passes] = 1, p1.shift[p1.n_passes++] = (uint8_t)(8 * b);
    for (int b = 0; b < seq_bytes; ++b) p1.word[p1.n_passes] = 0, p1.shift[p1.n_passes++] = (uint8_t)(8 * b);
    for (int b = 0; b < 4; ++b) p2.word[p2.n_passes] = 2, p2.shift[p2.n_passes++] = (uint8_t)(8 * b);
    for (int b = 0; b < seq_bytes; ++b) p2.word[p2.n_passes] = 0, p2.shift[p2.n_passes++] = (uint8_t)(8 * b);
    uint32_t *qs = L->d_tab, *pm = qs + L->cap_q, *sm = pm + L->cap_q, *qe = sm + L->cap_q, *eq = qe + L->cap_q;
    const bool use_dir = nq > 0;
    GFFX_HIP_TRY(hipEventRecord(L->ev_p0, L->stream));
    GFFX_HIP_TRY(hipMemsetAsync(L->d_err, 0, 16, L->stream));
    GFFX_HIP_TRY(hipMemsetAsync(L->d_qoff, 0, (n_seq + 1) * 8, L->stream));
    GFFX_HIP_TRY(hipMemsetAsync(L->d_doff, 0, (n_seq + 1) * 8, L->stream));
    if (nq) {
        uint32_t *s1 = nullptr, *s2 = nullptr;
        int rc = DeviceSort::run(L->stream, L->d_rec_a, L->d_rec_b, n, p1, n_seq, L->d_work, L->d_err, &s1);
        if (rc) return rc;
        // keep the (seqid, qs, qe) order: the second sort needs both ping-pong buffers
        GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_c, s1, n * 12, hipMemcpyDeviceToDevice, L->stream));
        hipLaunchKernelGGL(k_b_offsets, dim3(g256), dim3(256), 0, L->stream, L->d_rec_c, n, n_seq, L->d_qoff);
        hipLaunchKernelGGL(k_b_columns, dim3(g256), dim3(256), 0, L->stream, L->d_rec_c, n, qs, eq, L->d_head, L->d_head + L->cap_q);
        if ((rc = seg_scan<true, false>(L, eq, L->d_head, nq, pm))) return rc;               // PM: running max of the ends
        if ((rc = seg_scan<false, true>(L, eq, L->d_head + L->cap_q, nq, sm))) return rc;     // SM: running min from the right
        if (s1 != L->d_rec_a) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, s1, n * 12, hipMemcpyDeviceToDevice, L->stream));
        if ((rc = DeviceSort::run(L->stream, L->d_rec_a, L->d_rec_b, n, p2, n_seq, L->d_work, L->d_err, &s2))) return rc;
        hipLaunchKernelGGL(k_b_column2, dim3(g256), dim3(256), 0, L->stream, s2, n, qe);
        hipLaunchKernelGGL(k_b_dir_meta, dim3(1), dim3(256), 0, L->stream, L->d_qoff, qs, qe, n_seq, L->d_dmeta, L->d_doff);
        hipLaunchKernelGGL(k_b_dir_fill, dim3(g256), dim3(256), 0, L->stream, qs, L->d_rec_c, n, n_seq, L->d_qoff, L->d_doff, L->d_dmeta, L->d_dir);
        hipLaunchKernelGGL(k_b_dir_fill, dim3(g256), dim3(256), 0, L->stream, qe, s2, n, n_seq, L->d_qoff, L->d_doff, L->d_dmeta, L->d_dir + L->cap_dir);
        GFFX_HIP_TRY(hipGetLastError());
    }
    GFFX_HIP_TRY(hipEventRecord(L->ev_p1, L->stream));
    if (L->n) {
        LinesView lv{L->d_seq, L->d_start, L->d_end, (unsigned long long)L->n};
        RegionsView rv{L->d_qoff, qs, pm, sm, qe, use_dir ? L->d_dir : nullptr, use_dir ? L->d_dir + L->cap_dir : nullptr,
                       L->d_doff, L->d_dmeta, n_seq};
        const unsigned blocks = (unsigned)((L->n + 255) / 256);
        GFFX_HIP_TRY(hipEventRecord(L->ev_a, L->stream));
        if (mode == GFFX_MODE_CONTAINED)
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_CONTAINED>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        else if (mode == GFFX_MODE_CONTAINS_REGION)
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_CONTAINS_REGION>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
        else
            hipLaunchKernelGGL((k_lines_exists<GFFX_MODE_OVERLAP>), dim3(blocks), dim3(256), 0, L->stream, lv, rv, L->d_keep);
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
    L->last_dir = use_dir;
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
    if ((rc = lines_reserve(L, nq, n_seq, 9 + 3))) return rc;
    if (nq) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, regions, nq * 12, hipMemcpyHostToDevice, L->stream));
    return lines_run(L, nq, n_seq, mode, keep_host);
}

// the same with the regions already in HBM (AoS triples, e.g. the batch Join A uploaded: gffx_hip_batch_device_regions)
extern "C" int gffx_hip_lines_test_device(gffx_hip_lines *L, const uint32_t *d_regions, uint64_t nq,
                                          uint32_t n_seq, int mode, uint8_t *keep_host) {
    int rc = lines_check(L, d_regions, nq, mode, keep_host, "gffx_hip_lines_test_device");
    if (rc) return rc;
    GFFX_HIP_TRY(hipSetDevice(L->device));
    if ((rc = lines_reserve(L, nq, n_seq, 9 + 3))) return rc;
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
    if ((rc = lines_reserve(L, nq, n_seq, 9 + 3))) return rc;
    for (int k = 0; k < 2; ++k)  // every append has to have landed
        if (R->pending[k]) GFFX_HIP_TRY(hipStreamWaitEvent(L->stream, R->copied[k], 0));
    if (nq) GFFX_HIP_TRY(hipMemcpyAsync(L->d_rec_a, R->d, nq * 12, hipMemcpyDeviceToDevice, L->stream));
    return lines_run(L, nq, n_seq, mode, keep_host);
}

// The region tables of the last _test, for parity tests: q_off (n_seq + 1), then QS, PM, SM, QE (nq each).
extern "C" int gffx_hip_lines_copy_tables(gffx_hip_lines *L, uint64_t *q_off, uint32_t *qs, uint32_t *pm, uint32_t *sm, uint32_t *qe) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_tables: lines is NULL");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    const uint64_t nq = L->last_nq;
    if (q_off) GFFX_HIP_TRY(hipMemcpy(q_off, L->d_qoff, (L->last_n_seq + 1) * 8, hipMemcpyDeviceToHost));
    uint32_t *host[4] = {qs, pm, sm, qe};
    for (int t = 0; t < 4; ++t)
        if (host[t] && nq) GFFX_HIP_TRY(hipMemcpy(host[t], L->d_tab + (size_t)t * L->cap_q, nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
// ... and the two bin directories: d_off (n_seq + 1), shift_nb (2 per seqid), dir_qs / dir_qe (d_off[n_seq] each)
extern "C" int gffx_hip_lines_copy_dirs(gffx_hip_lines *L, uint64_t *d_off, uint32_t *shift_nb, uint32_t *dir_qs, uint32_t *dir_qe) {
    if (!L) return fail(GFFX_E_INVALID, "gffx_hip_lines_copy_dirs: lines is NULL");
    GFFX_HIP_TRY(hipSetDevice(L->device));
    std::vector<unsigned long long> off(L->last_n_seq + 1, 0);
    GFFX_HIP_TRY(hipMemcpy(off.data(), L->d_doff, off.size() * 8, hipMemcpyDeviceToHost));
    if (d_off)
        for (size_t i = 0; i < off.size(); i++) d_off[i] = off[i];
    if (shift_nb && L->last_n_seq) GFFX_HIP_TRY(hipMemcpy(shift_nb, L->d_dmeta, (size_t)L->last_n_seq * 8, hipMemcpyDeviceToHost));
    const uint64_t total = off.back();
    if (dir_qs && total) GFFX_HIP_TRY(hipMemcpy(dir_qs, L->d_dir, total * 4, hipMemcpyDeviceToHost));
    if (dir_qe && total) GFFX_HIP_TRY(hipMemcpy(dir_qe, L->d_dir + L->cap_dir, total * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" double gffx_hip_lines_last_prep_ms(const gffx_hip_lines *L) { return L ? L->last_prep_ms : 0.0; }

extern "C" double gffx_hip_lines_last_kernel_ms(const gffx_hip_lines *L) { return L ? L->last_kernel_ms : 0.0; }
