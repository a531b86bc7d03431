// depth_kernels.hpp -- `gffx depth` (BED source), the join after Join A (commands/depth.rs:121-217).
//
// For every (region, root) pair of Join A -- deduped per region by root_fid (depth.rs:241) -- the
// reference re-parses the root's byte block and, per region, collects the set of feature IDs with
// a line overlapping the region (half-open, 0-based: depth.rs:78-82,146-147), adds 1 to each of
// them and tracks min start / max end over the overlapped lines.  Here the blocks are parsed ONCE
// on the host into a line table {start, end, group}: the lines of a block are sorted by ID and a
// GROUP is one (block, ID) -- the unit the reference dedups on.  One wave per region walks the
// blocks of its pairs 64 lines at a time (coalesced 12 B/line); a ballot finds, for every run of
// equal groups, the first overlapping lane, which adds 1 to depth[group]; overlapping lanes
// atomicMin / atomicMax the group's extent.  Per-group results are merged to IDs on the host
// (min / max / sum -- the merges of depth.rs:264-291 and :501-508).
// Roofline bound: HBM.  Algorithmic bytes: 12 B per (pair, block line) in + the group atomics.
#pragma once
#include "gffx_device.hpp"

namespace gffx {

struct DepthTableView {
    const unsigned long long *block_off;  // n_blocks + 1: lines of block b are [block_off[b], block_off[b+1])
    const uint32_t *line_start, *line_end, *line_group;
    const uint32_t *block_of_fid;         // n_fid entries, UINT32_MAX = the fid has no (valid) block
    uint32_t n_fid;
};

struct DepthAcc {
    unsigned long long *depth;  // per group
    uint32_t *min_start, *max_end;
};

__global__ __launch_bounds__(256) void k_depth_regions(DepthTableView T, QueryView q, unsigned long long nq,
                                                       const uint32_t *counts, const unsigned long long *offsets,
                                                       const uint32_t *fids, DepthAcc acc) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq) return;
    const int lane = threadIdx.x & 63;
    const uint32_t cnt = counts[i];
    if (cnt == 0) return;
    uint32_t qs, qe;
    if (q.aos) {
        qs = q.aos[3 * i + 1];
        qe = q.aos[3 * i + 2];
    } else {
        qs = q.start[i];
        qe = q.end[i];
    }
    const unsigned long long off = offsets[i];
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t fid = fids[off + j];
        bool dup = false;  // a region counts a root once even if two tree intervals carry its fid (depth.rs:241)
        for (uint32_t j2 = 0; j2 < j; ++j2) dup |= fids[off + j2] == fid;
        if (dup || fid >= T.n_fid) continue;
        const uint32_t blk = T.block_of_fid[fid];
        if (blk == 0xFFFFFFFFu) continue;  // depth.rs:242-243
        const unsigned long long lb = T.block_off[blk], le = T.block_off[blk + 1];
        uint32_t carry_group = 0xFFFFFFFFu;  // group of the previous chunk's last line, and whether its run already hit
        bool carry_hit = false;
        for (unsigned long long base = lb; base < le; base += 64) {
            const unsigned long long l = base + lane;
            const bool valid = l < le;
            uint32_t s = 0, e = 0, g = 0xFFFFFFFEu;
            if (valid) {
                s = T.line_start[l];
                e = T.line_end[l];
                g = T.line_group[l];
            }
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
                atomicMin(&acc.min_start[g], s);
                atomicMax(&acc.max_end[g], e);
                if (!earlier) atomicAdd(&acc.depth[g], 1ull);
            }
            // carry for the next chunk: the last line's group and whether its run has hit so far
            const int lastl = (int)min(63ull, le - base - 1ull);
            const bool run_hit = hit || earlier;
            carry_group = __shfl(g, lastl, 64);
            // the run of the last lane: any hit among its lanes (the last lane's own `hit || earlier` covers them)
            carry_hit = __shfl((int)run_hit, lastl, 64) != 0;
        }
    }
}

}  // namespace gffx
