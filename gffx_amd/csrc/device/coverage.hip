// coverage.hip -- `gffx coverage` (BED source): covered bases of feature segments under the union of the regions
// of their seqid (reference: commands/coverage.rs:92-124 merge_intervals / union_len, :339-364 the two-pointer walk).
//
// The reference merges, per root, the regions that hit the root and walks every line of the root's block against
// that list.  A region that overlaps a segment lying inside the root's interval necessarily hits the root, so for
// such segments the per-root list can be replaced by ONE list per seqid: the union U of all regions (sorted,
// touching spans merged like merge_intervals).  With P[k] = total length of U[0..k) the covered bases below x are
//      F(x) = P[k-1] + min(x, U[k-1].end) - U[k-1].start,   k = #{U.start < x}   (0 if k == 0)
// and a segment [a, b) holds F(b) - F(a) covered bases: two searches per segment, each narrowed to one bin of a
// directory over U.start (as in join_b.hip).  Segments that stick out of their root's interval are handled on
// the host (exactly, they are rare).  One thread per segment; 12 B in, 4 B out.  Roofline bound: HBM.
#include <algorithm>
#include <vector>

#include "gffx_device.hpp"

namespace gffx {

struct UnionView {
    const unsigned long long *u_off;  // n_seq + 1: spans of seqid c are [u_off[c], u_off[c+1])
    const uint32_t *us, *ue;          // sorted, disjoint, non-touching
    const unsigned long long *pb;     // covered bases of the seqid's spans before this one
    const uint32_t *dir;              // directory over us (see join_b.hip)
    const unsigned long long *d_off;  // n_seq + 1
    const uint2 *d_meta;              // per seqid {shift, nb}
    uint32_t n_seq;
};

__device__ __forceinline__ unsigned long long covered_below(const UnionView &U, uint32_t seq, unsigned long long lo,
                                                            unsigned long long hi, uint32_t x) {
    unsigned long long a = lo, b = hi;
    const uint2 m = U.d_meta[seq];
    const uint32_t bin = x >> m.x;
    if (bin >= m.y) {
        a = b = hi;  // beyond the largest start: every span starts below x
    } else {
        const uint32_t *d = U.dir + U.d_off[seq] + bin;
        a = d[0];
        b = d[1];
    }
    while (a < b) {  // first span with start >= x
        const unsigned long long mid = (a + b) >> 1;
        if (U.us[mid] < x)
            a = mid + 1;
        else
            b = mid;
    }
    if (a == lo) return 0ull;
    const unsigned long long k = a - 1;
    return U.pb[k] + (min(x, U.ue[k]) - U.us[k]);
}

__global__ __launch_bounds__(256) void k_segments_covered(UnionView U, unsigned long long n_seg, const uint32_t *seg_seq,
                                                          const uint32_t *seg_start, const uint32_t *seg_end,
                                                          uint32_t *covered) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_seg) return;
    const uint32_t seq = seg_seq[i], a = seg_start[i], b = seg_end[i];
    uint32_t c = 0;
    if (seq < U.n_seq && a < b) {
        const unsigned long long lo = U.u_off[seq], hi = U.u_off[seq + 1];
        if (hi > lo) c = (uint32_t)(covered_below(U, seq, lo, hi, b) - covered_below(U, seq, lo, hi, a));
    }
    covered[i] = c;
}

}  // namespace gffx

using namespace gffx;

extern "C" int gffx_hip_segments_covered(int device, uint64_t n_seg, const uint32_t *seg_seq, const uint32_t *seg_start,
                                         const uint32_t *seg_end, const uint32_t *regions, uint64_t nq, uint32_t n_seq,
                                         uint32_t *covered_out) {
    if (n_seg && (!seg_seq || !seg_start || !seg_end || !covered_out))
        return fail(GFFX_E_INVALID, "gffx_hip_segments_covered: NULL segment array");
    if (nq && !regions) return fail(GFFX_E_INVALID, "gffx_hip_segments_covered: regions is NULL");
    for (uint64_t i = 0; i < nq; i++)
        if (regions[3 * i] >= n_seq)
            return fail(GFFX_E_CHR_RANGE, "gffx_hip_segments_covered: region %llu has chr %u >= %u", (unsigned long long)i,
                        regions[3 * i], n_seq);
    for (uint64_t i = 0; i < n_seg; i++)
        if (seg_seq[i] >= n_seq)
            return fail(GFFX_E_CHR_RANGE, "gffx_hip_segments_covered: segment %llu has chr %u >= %u", (unsigned long long)i,
                        seg_seq[i], n_seq);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        ndev = 0;
    }
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    if (n_seg == 0) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(device));
    // per seqid: sort by start, merge while s <= current end (coverage.rs:92-109), prefix of the lengths, directory
    std::vector<unsigned long long> q_off(n_seq + 1, 0);
    for (uint64_t i = 0; i < nq; i++) q_off[regions[3 * i] + 1]++;
    for (uint32_t c = 0; c < n_seq; c++) q_off[c + 1] += q_off[c];
    std::vector<uint64_t> key(nq);
    {
        std::vector<unsigned long long> cur(q_off.begin(), q_off.end() - 1);
        for (uint64_t i = 0; i < nq; i++) key[cur[regions[3 * i]]++] = ((uint64_t)regions[3 * i + 1] << 32) | regions[3 * i + 2];
    }
    std::vector<unsigned long long> u_off(n_seq + 1, 0), pb, d_off(n_seq + 1, 0);
    std::vector<uint32_t> us, ue, dir;
    std::vector<uint2> d_meta(n_seq, make_uint2(0, 0));
    for (uint32_t c = 0; c < n_seq; c++) {
        const uint64_t lo = q_off[c], hi = q_off[c + 1];
        const size_t u0 = us.size();
        if (hi > lo) {
            std::sort(key.begin() + lo, key.begin() + hi);
            uint32_t cs = (uint32_t)(key[lo] >> 32), ce = (uint32_t)key[lo];
            if (ce < cs) ce = cs;  // (rows with s >= e never get here: coverage.rs:251)
            unsigned long long acc = 0;
            for (uint64_t i = lo + 1; i < hi; i++) {
                const uint32_t s = (uint32_t)(key[i] >> 32), e = (uint32_t)key[i];
                if (s <= ce) {
                    ce = std::max(ce, e);
                } else {
                    us.push_back(cs);
                    ue.push_back(ce);
                    pb.push_back(acc);
                    acc += ce - cs;
                    cs = s;
                    ce = e;
                }
            }
            us.push_back(cs);
            ue.push_back(ce);
            pb.push_back(acc);
        }
        u_off[c + 1] = us.size();
        d_off[c + 1] = d_off[c];
        const size_t n_u = us.size() - u0;
        if (n_u) {
            const uint32_t vmax = us.back();
            const uint64_t budget = std::max<uint64_t>(2 * n_u, 16);
            uint32_t shift = 0;
            while ((((uint64_t)vmax >> shift) + 1) > budget) shift++;
            const uint32_t nb = (vmax >> shift) + 1;
            d_meta[c] = make_uint2(shift, nb);
            size_t p = u0;
            for (uint32_t b = 0; b < nb; b++) {
                const uint64_t edge = (uint64_t)b << shift;
                while (p < us.size() && us[p] < edge) p++;
                dir.push_back((uint32_t)p);
            }
            dir.push_back((uint32_t)us.size());
            d_off[c + 1] = dir.size();
        }
    }
    if (us.size() >= 0xFFFFFFFFull) return fail(GFFX_E_INVALID, "gffx_hip_segments_covered: too many spans");
    void *bufs[12] = {nullptr};
    int nb_ = 0;
    auto cleanup = [&]() {
        for (int i = 0; i < nb_; i++) (void)hipFree(bufs[i]);
    };
    auto up = [&](const void *src, size_t bytes, void **dst) -> int {
        *dst = nullptr;
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GFFX_E_OOM : GFFX_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        bufs[nb_++] = *dst;
        if (bytes && src) {
            e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(GFFX_E_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
        }
        return GFFX_OK;
    };
    void *d_uoff, *d_us, *d_ue, *d_pb, *d_dir, *d_doff, *d_dmeta, *d_sseq, *d_ss, *d_se, *d_cov;
    int rc;
    if ((rc = up(u_off.data(), u_off.size() * 8, &d_uoff)) || (rc = up(us.data(), us.size() * 4, &d_us)) ||
        (rc = up(ue.data(), ue.size() * 4, &d_ue)) || (rc = up(pb.data(), pb.size() * 8, &d_pb)) ||
        (rc = up(dir.data(), dir.size() * 4, &d_dir)) || (rc = up(d_off.data(), d_off.size() * 8, &d_doff)) ||
        (rc = up(d_meta.data(), d_meta.size() * sizeof(uint2), &d_dmeta)) || (rc = up(seg_seq, n_seg * 4, &d_sseq)) ||
        (rc = up(seg_start, n_seg * 4, &d_ss)) || (rc = up(seg_end, n_seg * 4, &d_se)) || (rc = up(nullptr, n_seg * 4, &d_cov))) {
        cleanup();
        return rc;
    }
    const UnionView U{(const unsigned long long *)d_uoff, (const uint32_t *)d_us, (const uint32_t *)d_ue,
                      (const unsigned long long *)d_pb,   (const uint32_t *)d_dir, (const unsigned long long *)d_doff,
                      (const uint2 *)d_dmeta,             n_seq};
    hipLaunchKernelGGL(k_segments_covered, dim3((uint32_t)((n_seg + 255) / 256)), dim3(256), 0, 0, U,
                       (unsigned long long)n_seg, (const uint32_t *)d_sseq, (const uint32_t *)d_ss, (const uint32_t *)d_se,
                       (uint32_t *)d_cov);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(covered_out, d_cov, n_seg * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(GFFX_E_HIP, "k_segments_covered failed: %s", hipGetErrorString(e));
    return GFFX_OK;
}
