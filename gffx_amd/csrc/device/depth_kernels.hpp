// depth_kernels.hpp -- `gffx depth` (BED source), the join after Join A (commands/depth.rs:121-217).
//
// For every (region, root) pair of Join A -- deduped per region by root_fid (depth.rs:241) -- the
// reference re-parses the root's byte block and, per region, collects the set of feature IDs with
// a line overlapping the region (half-open, 0-based: depth.rs:78-82,146-147), adds 1 to each of
// them and tracks min start / max end over the overlapped lines.  Here the blocks are parsed ONCE
// on the host into a line table {start, end, group}: the lines of a block are sorted by ID and a
// GROUP is one (block, ID) -- the unit the reference dedups on.  A wave serves 64 regions: the pairs'
// fids and line ranges are fetched lane-parallel, then the wave walks the (region, block) items 64
// lines at a time (coalesced 12 B/line), two items per step; a ballot finds, for every run of equal
// groups, the first overlapping lane, which adds 1 to depth[group]; overlapping lanes flag their line,
// and the group extents (min start / max end over overlapped lines) are taken from the flags when the
// results are read.  Per-group results are merged to IDs on the host
// (min / max / sum -- the merges of depth.rs:264-291 and :501-508).
// Roofline bound: HBM.  Algorithmic bytes: 12 B per (pair, block line) in + the group atomics.
#pragma once
#include "gffx_device.hpp"

namespace gffx {

struct DepthTableView {
    const uint32_t *line_start, *line_end, *line_group;
    const uint2 *fid_lines;  // n_fid entries: {first line, number of lines} of the fid's block; y = 0xFFFFFFFF: no
                             // usable block (depth.rs:242-243)
    uint32_t n_fid;
};

struct DepthAcc {
    unsigned long long *depth;  // per group: regions with an overlapping line
    uint8_t *line_hit;          // per line: overlapped by at least one counted region (min start / max end of a
                                // group are taken over these lines afterwards: k_depth_extent -- a plain coalesced
                                // store here instead of two random atomics per overlapping line)
};

// One wave serves 64 consecutive regions.  Step j of the wave: every lane whose region has a j-th pair
// fetches that pair's root_fid and the line range of its block (lane-parallel gathers), drops it if an
// earlier pair of the region carries the same fid (a region counts a root once, depth.rs:241); then the
// wave walks the surviving (region, block) items one after the other, two at a time so that the line
// loads of one item are in flight while the other is evaluated: 64 lines per chunk, coalesced.
__device__ __forceinline__ void depth_item_chunk(const DepthTableView &T, const DepthAcc &acc, int lane, uint32_t qs,
                                                 uint32_t qe, unsigned long long base, unsigned long long le, bool valid,
                                                 uint32_t s, uint32_t e, uint32_t g, uint32_t &carry_group,
                                                 bool &carry_hit) {
    const bool hit = valid && max(s, qs) < min(e, qe);  // depth.rs:78-82
    const unsigned long long hitmask = __ballot(hit);
    uint32_t prev = __shfl_up(g, 1, 64);
    if (lane == 0) prev = carry_group;
    const bool same = valid && g == prev;
    const unsigned long long startmask = __ballot(!same);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // bits <= lane
    const unsigned long long below = startmask & upto;
    const int rs = below ? 63 - __clzll(below) : -1;  // first lane of my run; -1: it continues the carry run
    const unsigned long long before = (1ull << lane) - 1ull;
    const unsigned long long run_lo = rs > 0 ? ~((1ull << rs) - 1ull) : ~0ull;
    const bool earlier = (hitmask & before & run_lo) != 0ull || (rs < 0 && carry_hit);
    if (hit) {
        acc.line_hit[base + lane] = 1;
        if (!earlier) atomicAdd(&acc.depth[g], 1ull);  // the first overlapping line of a (block, ID) group
    }
    const int lastl = (int)min(63ull, le - base - 1ull);
    carry_group = __shfl(g, lastl, 64);
    carry_hit = __shfl((int)(hit || earlier), lastl, 64) != 0;
}

__global__ __launch_bounds__(256) void k_depth_regions(DepthTableView T, QueryView q, unsigned long long nq,
                                                       const uint32_t *counts, const unsigned long long *offsets,
                                                       const uint32_t *fids, DepthAcc acc) {
    const unsigned long long wave0 = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (wave0 >= nq) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long i = wave0 + lane;
    uint32_t cnt = 0, qs = 0, qe = 0;
    unsigned long long off = 0;
    if (i < nq) {
        cnt = counts[i];
        if (cnt) {
            off = offsets[i];
            if (q.aos) {
                qs = q.aos[3 * i + 1];
                qe = q.aos[3 * i + 2];
            } else {
                qs = q.start[i];
                qe = q.end[i];
            }
        }
    }
    uint32_t maxcnt = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, (uint32_t)__shfl_xor(maxcnt, o, 64));
    for (uint32_t j = 0; j < maxcnt; ++j) {
        // ---- lane-parallel: the j-th pair of my region
        uint32_t lb = 0, n = 0;
        if (j < cnt) {
            const uint32_t fid = fids[off + j];
            bool dup = false;
            for (uint32_t j2 = 0; j2 < j; ++j2) dup |= fids[off + j2] == fid;
            if (!dup && fid < T.n_fid) {
                const uint2 r = T.fid_lines[fid];
                if (r.y != 0xFFFFFFFFu) {
                    lb = r.x;
                    n = r.y;
                }
            }
        }
        // ---- the wave walks the items, two per step
        unsigned long long todo = __ballot(n > 0);
        while (todo) {
            const int la = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int lbn = todo ? __ffsll((long long)todo) - 1 : -1;
            if (lbn >= 0) todo &= todo - 1;
            const uint32_t a_qs = __shfl(qs, la, 64), a_qe = __shfl(qe, la, 64), a_lb = __shfl(lb, la, 64),
                           a_n = __shfl(n, la, 64);
            const int lb2 = lbn >= 0 ? lbn : la;
            const uint32_t b_qs = __shfl(qs, lb2, 64), b_qe = __shfl(qe, lb2, 64), b_lb = __shfl(lb, lb2, 64),
                           b_n = lbn >= 0 ? __shfl(n, lb2, 64) : 0u;
            const unsigned long long a_le = (unsigned long long)a_lb + a_n, b_le = (unsigned long long)b_lb + b_n;
            uint32_t a_cg = 0xFFFFFFFFu, b_cg = 0xFFFFFFFFu;
            bool a_ch = false, b_ch = false;
            unsigned long long ab = a_lb, bb = b_lb;
            while (ab < a_le || bb < b_le) {
                const bool a_on = ab < a_le, b_on = bb < b_le;
                const bool av = a_on && ab + lane < a_le, bv = b_on && bb + lane < b_le;
                uint32_t as = 0, ae = 0, ag = 0xFFFFFFFEu, bs = 0, be = 0, bg = 0xFFFFFFFEu;
                if (av) {
                    as = T.line_start[ab + lane];
                    ae = T.line_end[ab + lane];
                    ag = T.line_group[ab + lane];
                }
                if (bv) {
                    bs = T.line_start[bb + lane];
                    be = T.line_end[bb + lane];
                    bg = T.line_group[bb + lane];
                }
                if (a_on) {
                    depth_item_chunk(T, acc, lane, a_qs, a_qe, ab, a_le, av, as, ae, ag, a_cg, a_ch);
                    ab += 64;
                }
                if (b_on) {
                    depth_item_chunk(T, acc, lane, b_qs, b_qe, bb, b_le, bv, bs, be, bg, b_cg, b_ch);
                    bb += 64;
                }
            }
        }
    }
}

// min start / max end per group over the lines that were overlapped (depth.rs:198-199), once per result read
__global__ __launch_bounds__(256) void k_depth_extent(unsigned long long n_lines, const uint8_t *line_hit,
                                                      const uint32_t *line_start, const uint32_t *line_end,
                                                      const uint32_t *line_group, uint32_t *min_start, uint32_t *max_end) {
    const unsigned long long l = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_lines || !line_hit[l]) return;
    atomicMin(&min_start[line_group[l]], line_start[l]);
    atomicMax(&max_end[line_group[l]], line_end[l]);
}

}  // namespace gffx
