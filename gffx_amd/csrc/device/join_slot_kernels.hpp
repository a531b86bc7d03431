// join_slot_kernels.hpp -- Join A over the SLOT index ("slots" strategy, the default): queries in input order,
// count + emit in one kernel, ONE gather per query in the usual case.  Same result set as join_a_kernels.hpp
// (utils/tree.rs:110 + intersect.rs:145-161).
//
// The sweep of join_fused_kernels.hpp costs a chain of dependent gathers per query (bin record -> refine ->
// aux[] steps); a wave waits for the longest chain among its lanes (4-6 gathers of ~1.4 us under load), and the
// chip's L1s carry 2.1 misses per query.  The slot index trades that chain for a precomputed candidate list:
// every seqid is cut into windows of 2^shift bp (~1 per entry); the SLOT of window b lists every entry that can
// overlap a query of width <= wmax whose last base lies in the window:  start < (b+1) << shift  and
// end + wmax > b << shift.  A 32-byte slot carries the list length, the first two entries {start, end,
// root_fid} and where the rest of the list sits (16-byte {start, end, root_fid, position} records, `spill`).
// Query (qs, qe) with 0 < qe - qs <= wmax: read slot (qe-1) >> shift, test its entries exactly
// (start < qe && end > qs, then the mode predicate) -- 68 % of GENCODE-shaped queries end after that one access,
// the others read their spill entries from one more line.  Wider or empty-width queries, and windows whose list
// is longer than kSlotMaxList (dense clusters: the list stops being output-sensitive), take the exact skip-link
// sweep of join_a_kernels.hpp in their lane; the host switches a batch that is mostly such queries to the
// sweep kernel.
// A thread serves 4 CONSECUTIVE queries (three 16-byte loads bring them in, one 16-byte store per output
// array); kept root_fids wait in registers for the round's one pair reservation (same scheme as the fused
// kernel), so there is no LDS queue.  Roofline bound: HBM.  Algorithmic bytes per query: 12 in + 4 + 4*h out.
#pragma once
#include "join_fused_kernels.hpp"

#ifndef GFFX_SLOT_THREADS
#define GFFX_SLOT_THREADS 512
#endif
#ifndef GFFX_SLOT_MIN_WAVES
#define GFFX_SLOT_MIN_WAVES 4
#endif

namespace gffx {

constexpr int kSlotThreads = GFFX_SLOT_THREADS;
constexpr int kSlotItems = 4;  // fixed: one uint4 of every query column per thread
constexpr uint32_t kSlotChunk = kSlotThreads * kSlotItems;
constexpr uint32_t kSlotStage = 4096;  // root_fids of a round staged in LDS (16 KB) so that they leave in full lines
constexpr uint32_t kSlotExtras = 6;    // per thread: root_fids kept from list entries 4.. wait here (LDS) for the emit
static_assert(kSlotMaxList <= 16, "entries 0-3 live in registers, the rest is read in batches of four");

// 16-byte nontemporal accesses (the builtins want clang vector types, not HIP's struct uint4)
typedef uint32_t gffx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long gffx_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 nt_load4(const uint32_t *p) {
    const gffx_u32x4 v = GFFX_NT_LOAD(reinterpret_cast<const gffx_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    gffx_u32x4 v;
    v.x = a, v.y = b, v.z = c, v.w = d;
    GFFX_NT_STORE(v, reinterpret_cast<gffx_u32x4 *>(p));
}
__device__ __forceinline__ void nt_store2(unsigned long long *p, unsigned long long a, unsigned long long b) {
    gffx_u64x2 v;
    v.x = a, v.y = b;
    GFFX_NT_STORE(v, reinterpret_cast<gffx_u64x2 *>(p));
}

// index gathers: plain by default (tools/kbench.hip times the nontemporal variant with -DGFFX_SLOT_NT_GATHER)
__device__ __forceinline__ uint4 gather4(const uint4 *p) {
#if defined(GFFX_SLOT_NT_GATHER)
    const gffx_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gffx_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

template <int MODE, bool INVERT>
__device__ __forceinline__ bool slot_test(uint32_t s, uint32_t e, uint32_t qs, uint32_t qe) {
    return s < qe && e > qs && keep_pair<MODE, INVERT>(s, e, qs, qe);
}

// the kept pairs of one query, the slow way (exact for every query): f(position, start, aux)
template <int MODE, bool INVERT, typename F>
__device__ __forceinline__ void slot_slow(const IndexView &ix, uint32_t chr, uint32_t qs, uint32_t qe, F &&f) {
    for_each_kept<MODE, INVERT>(ix, ix.chr_meta[chr], qs, qe, f);
}

// OUT: 0 = counts (+ offsets) only, 1 = root_fids as the only pair output (bench / depth), 2 = anything (triples, bitmap)
template <int MODE, bool INVERT, bool AOS, bool META_LDS, int OUT>
__global__ __launch_bounds__(kSlotThreads, GFFX_SLOT_MIN_WAVES) void k_join_slots(IndexView ix, QueryView q, unsigned long long nq,
                                                                                   FusedOut out, int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_scratch = reinterpret_cast<uint32_t *>(smem);                       // 64 B
    unsigned long long *s_base = reinterpret_cast<unsigned long long *>(smem + 64);  // 8 B
    uint32_t *s_fids = reinterpret_cast<uint32_t *>(smem + 80);                      // kSlotStage x 4 B
    uint32_t *s_extra = s_fids + kSlotStage + kSlotExtras * threadIdx.x;             // this thread's kSlotExtras x 4 B
    // the 4 consecutive queries of this thread in round r: three 16-byte loads (scalar ones for an unaligned column or
    // the batch's ragged end)
    uint32_t qc[4], qs[4], qe[4];
    auto load_round = [&](unsigned long long r) {
        const unsigned long long i0 = r * kSlotChunk + 4ull * threadIdx.x;
        if (vec_ok && i0 + 4 <= nq) {
            if (AOS) {
                const uint32_t *p = q.aos + 3ull * i0;
                const uint4 a = nt_load4(p), b = nt_load4(p + 4), c = nt_load4(p + 8);
                qc[0] = a.x, qs[0] = a.y, qe[0] = a.z;
                qc[1] = a.w, qs[1] = b.x, qe[1] = b.y;
                qc[2] = b.z, qs[2] = b.w, qe[2] = c.x;
                qc[3] = c.y, qs[3] = c.z, qe[3] = c.w;
            } else {
                const uint4 c = nt_load4(q.chr + i0), s = nt_load4(q.start + i0), e = nt_load4(q.end + i0);
                qc[0] = c.x, qc[1] = c.y, qc[2] = c.z, qc[3] = c.w;
                qs[0] = s.x, qs[1] = s.y, qs[2] = s.z, qs[3] = s.w;
                qe[0] = e.x, qe[1] = e.y, qe[2] = e.z, qe[3] = e.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qc[k] = 0xFFFFFFFFu;  // "no query"
                qs[k] = qe[k] = 0;
                if (i0 + k < nq) load_query<AOS>(q, i0 + k, qc[k], qs[k], qe[k]);
            }
        }
    };
    const unsigned long long n_rounds = (nq + kSlotChunk - 1) / kSlotChunk;
    if (blockIdx.x < n_rounds) load_round(blockIdx.x);  // in flight while the seqid table is staged
    const uint4 *cm;  // seqid -> (first slot, n_slots, shift, wmax)
    if (META_LDS) {
        uint4 *m = reinterpret_cast<uint4 *>(smem + 80 + 4 * kSlotStage + 4 * kSlotExtras * kSlotThreads);
        for (uint32_t i = threadIdx.x; i < ix.n_chr; i += blockDim.x) m[i] = ix.slot_meta[i];
        __syncthreads();
        cm = m;
    } else {
        cm = ix.slot_meta;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out.pair_cursor_next = 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool want_pairs = OUT != 0;
    constexpr bool fids_only = OUT == 1;
    bool bad = false;
    uint32_t n_slow = 0;

    GFFX_STAMP(3, 0);
    for (unsigned long long r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        const unsigned long long i0 = r * kSlotChunk + 4ull * threadIdx.x;  // this thread's 4 consecutive queries
        if (r != blockIdx.x) load_round(r);
        const uint32_t nv = i0 + 4 <= nq ? 4u : (i0 < nq ? (uint32_t)(nq - i0) : 0u);  // this thread's queries in the batch
        // ---- slot record (one 32-byte gather per query, all four in flight together)
        uint32_t kind[4];  // 0 nothing to do, 1 slot list, 2 slow (exact sweep in this lane)
        uint32_t sidx[4];
        uint4 h0[4], h1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            kind[k] = 0;
            sidx[k] = 0;
            h0[k] = h1[k] = make_uint4(0, 0, 0xFFFFFFFFu, 0);
            h1[k].y = 0xFFFFFFFFu;
            if ((uint32_t)k >= nv) continue;
            if (qc[k] >= ix.n_chr) {
                bad = true;
                continue;
            }
            if (MODE == GFFX_MODE_OVERLAP && INVERT) continue;  // invert ^ true: nothing is ever kept
            const uint4 m = cm[qc[k]];
            if (m.y == 0) continue;  // seqid without roots
            if (qe[k] > qs[k] && qe[k] - qs[k] <= m.w) {
                const uint32_t b = (qe[k] - 1) >> m.z;
                if (b < m.y) {  // beyond the last window nothing can reach the query
                    kind[k] = 1;
                    sidx[k] = m.x + b;
                    h0[k] = gather4(ix.slots + 2ull * sidx[k]);
                    h1[k] = gather4(ix.slots + 2ull * sidx[k] + 1);
                }
            } else {
                kind[k] = 2;
            }
        }
        if (r == blockIdx.x) GFFX_STAMP(3, 1);
        // ---- the first two spill entries of the longer lists (one more line, all four items together)
        uint4 sp0[4], sp1[4];
        uint32_t n[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            n[k] = 0;
            sp0[k] = sp1[k] = make_uint4(0xFFFFFFFFu, 0, 0, 0);  // start = max: never < qe
            if (kind[k] != 1) continue;
            n[k] = h0[k].x & 255u;
            if (n[k] == 255u) {  // dense window
                kind[k] = 2;
                n[k] = 0;
                continue;
            }
            const uint32_t off = h0[k].x >> 8;
            if (n[k] > 2) sp0[k] = gather4(ix.spill + off);
            if (n[k] > 3) sp1[k] = gather4(ix.spill + off + 1);
        }
        // ---- count; the root_fids of the kept entries among the first four stay in registers
        uint32_t cnt[4], mask[4], f0[4], f1[4], f2[4], f3[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cnt[k] = mask[k] = 0;
            f0[k] = h1[k].x, f1[k] = h1[k].w, f2[k] = sp0[k].z, f3[k] = sp1[k].z;
            if (kind[k] == 1) {
                uint32_t mk = 0;
                // (an absent entry reads start = 0xFFFFFFFF: never < qe)
                if (slot_test<MODE, INVERT>(h0[k].z, h0[k].w, qs[k], qe[k])) mk |= 1u;
                if (slot_test<MODE, INVERT>(h1[k].y, h1[k].z, qs[k], qe[k])) mk |= 2u;
                if (slot_test<MODE, INVERT>(sp0[k].x, sp0[k].y, qs[k], qe[k])) mk |= 4u;
                if (slot_test<MODE, INVERT>(sp1[k].x, sp1[k].y, qs[k], qe[k])) mk |= 8u;
                mask[k] = mk;
                cnt[k] = __popc(mk);
            } else if (kind[k] == 2) {
                uint32_t c = 0;
                slot_slow<MODE, INVERT>(ix, qc[k], qs[k], qe[k], [&](uint32_t, uint32_t, const uint4 &) {
                    ++c;
                    return true;
                });
                cnt[k] = c;
                ++n_slow;
            }
        }
        // ---- list entries 4..7 (2 % of the queries, but some lane of nearly every wave): {start, end} of all of
        // them in ONE batch, the root_fid of a kept one from the line that read just brought in; they wait in LDS
        uint32_t xc = 0, xn = 0, xf = 0;  // extras stashed per item (4 bits each) / in total; items that walk again at emit
        uint32_t nmax = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) nmax = max(nmax, kind[k] == 1 ? n[k] : 0u);
        for (uint32_t jb = 4; jb < nmax; jb += 4) {  // (a second trip: 0.05 % of the queries)
            uint2 se[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 *sp = ix.spill + (h0[k].x >> 8) + (jb - 2);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    se[k][t] = make_uint2(0xFFFFFFFFu, 0);
                    if (kind[k] == 1 && n[k] > jb + t) se[k][t] = *reinterpret_cast<const uint2 *>(sp + t);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(kind[k] == 1 && n[k] > jb)) continue;
                const uint4 *sp = ix.spill + (h0[k].x >> 8) + (jb - 2);
                uint32_t km = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (n[k] > jb + t && slot_test<MODE, INVERT>(se[k][t].x, se[k][t].y, qs[k], qe[k])) km |= 1u << t;
                const uint32_t ce = __popc(km);
                if (ce == 0) continue;
                cnt[k] += ce;
                // (the emit reads the extras item after item: a later item must not have stashed before this one)
                if (!(xf >> k & 1u) && xn + ce <= kSlotExtras && (k == 3 || (xc >> (4 * (k + 1))) == 0)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (km >> t & 1u) s_extra[xn++] = sp[t].z;
                    xc += ce << (4 * k);
                } else {
                    xf |= 1u << k;
                }
            }
        }
        if (r == blockIdx.x) GFFX_STAMP(3, 2);
        // ---- reserve the round's pair segment: block scan + ONE returning atomicAdd
        const uint32_t mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        __syncthreads();  // s_scratch / s_base of the previous round are no longer read
        if (lane == 63) s_scratch[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, btotal = 0;
#pragma unroll
        for (int x = 0; x < kSlotThreads / 64; ++x) {
            const uint32_t v = s_scratch[x];
            if (x < wave) wbase += v;
            btotal += v;
        }
        if (threadIdx.x == 0) s_base[0] = btotal ? atomicAdd(out.pair_cursor, (unsigned long long)btotal) : 0ull;
        __syncthreads();
        unsigned long long pos = s_base[0] + wbase + inc - mine;
        if (r == blockIdx.x) GFFX_STAMP(3, 3);
        // ---- results out: counts / offsets in input order, 16 bytes per thread and array
        if (i0 + 4 <= nq) {
            nt_store4(out.counts + i0, cnt[0], cnt[1], cnt[2], cnt[3]);
            if (out.offsets) {
                nt_store2(out.offsets + i0, pos, pos + cnt[0]);
                nt_store2(out.offsets + i0 + 2, pos + cnt[0] + cnt[1], pos + cnt[0] + cnt[1] + cnt[2]);
            }
        } else {
            unsigned long long o = pos;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k < nq) {
                    out.counts[i0 + k] = cnt[k];
                    if (out.offsets) out.offsets[i0 + k] = o;
                }
                o += cnt[k];
            }
        }
        const unsigned long long seg = s_base[0];
        const bool staged = fids_only && btotal <= kSlotStage;  // block-uniform
        uint32_t xr = 0;  // read cursor into this thread's extras
        if (OUT == 1 && staged) {  // the timed path: LDS positions are 32-bit, nothing but LDS traffic per kept pair
            uint32_t lp = (uint32_t)(pos - seg);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (cnt[k] == 0) continue;
                if (kind[k] == 1) {
                    const uint32_t m = mask[k];
                    if (m & 1u) s_fids[lp] = f0[k];
                    if (m & 2u) s_fids[lp + (m & 1u)] = f1[k];
                    if (m & 4u) s_fids[lp + __popc(m & 3u)] = f2[k];
                    if (m & 8u) s_fids[lp + __popc(m & 7u)] = f3[k];
                    uint32_t e = lp + __popc(m);
                    const uint32_t xk = (xc >> (4 * k)) & 15u;
                    if (xf >> k & 1u) {  // did not fit the thread's LDS share: read entries 4.. again
                        xr += xk;
                        const uint32_t off = ix.slots[2ull * sidx[k]].x >> 8;
                        for (uint32_t j = 4; j < n[k]; ++j) {
                            const uint4 x = ix.spill[off + j - 2];
                            if (slot_test<MODE, INVERT>(x.x, x.y, qs[k], qe[k])) s_fids[e++] = x.z;
                        }
                    } else {
                        for (uint32_t x = 0; x < xk; ++x) s_fids[e++] = s_extra[xr++];
                    }
                } else {  // slow lane: walk the chain again
                    uint32_t e = lp;
                    slot_slow<MODE, INVERT>(ix, qc[k], qs[k], qe[k], [&](uint32_t, uint32_t, const uint4 &a) {
                        s_fids[e++] = a.w;
                        return true;
                    });
                }
                lp += cnt[k];
            }
        } else if (want_pairs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned long long o = pos;
                pos += cnt[k];
                if (cnt[k] == 0) continue;
                if (kind[k] == 1) {
                    if (fids_only) {  // the bench / depth path: straight from registers
                        auto put = [&](uint32_t f) {
                            if (staged)
                                s_fids[(uint32_t)(o - seg)] = f;
                            else if (o < out.capacity)
                                out.fids[o] = f;
                            ++o;
                        };
                        if (mask[k] & 1u) put(f0[k]);
                        if (mask[k] & 2u) put(f1[k]);
                        if (mask[k] & 4u) put(f2[k]);
                        if (mask[k] & 8u) put(f3[k]);
                        const uint32_t xk = (xc >> (4 * k)) & 15u;
                        if (xf >> k & 1u) {  // did not fit the thread's LDS share: read entries 4.. again
                            xr += xk;
                            const uint32_t off = ix.slots[2ull * sidx[k]].x >> 8;
                            for (uint32_t j = 4; j < n[k]; ++j) {
                                const uint4 e = ix.spill[off + j - 2];
                                if (slot_test<MODE, INVERT>(e.x, e.y, qs[k], qe[k])) put(e.z);
                            }
                        } else {
                            for (uint32_t x = 0; x < xk; ++x) put(s_extra[xr++]);
                        }
                    } else {  // triples / root bitmap: walk the (cache-warm) list again
                        const uint4 a = ix.slots[2ull * sidx[k]], b = ix.slots[2ull * sidx[k] + 1];
                        const uint32_t off = a.x >> 8;
                        for (uint32_t j = 0; j < n[k]; ++j) {
                            uint32_t s, e, f, p;
                            if (j == 0) {
                                s = a.z, e = a.w, f = b.x, p = out.bitmap ? ix.slot_pos[2ull * sidx[k]] : 0u;
                            } else if (j == 1) {
                                s = b.y, e = b.z, f = b.w, p = out.bitmap ? ix.slot_pos[2ull * sidx[k] + 1] : 0u;
                            } else {
                                const uint4 x = ix.spill[off + j - 2];
                                s = x.x, e = x.y, f = x.z, p = x.w;
                            }
                            if (!slot_test<MODE, INVERT>(s, e, qs[k], qe[k])) continue;
                            if (o < out.capacity) {
                                if (out.fids) out.fids[o] = f;
                                if (out.triples) {
                                    uint32_t *tr = out.triples + 3ull * o;
                                    tr[0] = f, tr[1] = s, tr[2] = e;
                                }
                                if (out.bitmap) atomicOr(&out.bitmap[p >> 5], 1u << (p & 31));
                            }
                            ++o;
                        }
                    }
                } else {  // slow lane: walk the chain again
                    slot_slow<MODE, INVERT>(ix, qc[k], qs[k], qe[k], [&](uint32_t j, uint32_t s, const uint4 &e) {
                        if (staged) {
                            s_fids[(uint32_t)(o - seg)] = e.w;
                        } else if (o < out.capacity) {
                            if (out.fids) out.fids[o] = e.w;
                            if (out.triples) {
                                uint32_t *tr = out.triples + 3ull * o;
                                tr[0] = e.w;
                                tr[1] = MODE == GFFX_MODE_OVERLAP ? ix.start[j] : s;
                                tr[2] = e.x;
                            }
                            if (out.bitmap) atomicOr(&out.bitmap[j >> 5], 1u << (j & 31));
                        }
                        ++o;
                        return true;
                    });
                }
            }
        }
        if (staged) {  // the round's root_fids leave as full lines
            __syncthreads();
            for (uint32_t x = threadIdx.x; x < btotal; x += kSlotThreads)
                if (seg + x < out.capacity) GFFX_NT_STORE(s_fids[x], out.fids + seg + x);
        }
        if (r == blockIdx.x) GFFX_STAMP(3, 4);
    }
    if (bad) atomicOr(out.err, 1u);
    // how many queries took the slow lane (the host moves a batch that is mostly such queries to the sweep kernel)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_slow += __shfl_xor(n_slow, o, 64);
    if (lane == 0 && n_slow) atomicAdd(out.err + 1, n_slow);
}

}  // namespace gffx
