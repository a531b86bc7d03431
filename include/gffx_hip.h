/*
 * gffx_hip.h -- C ABI of the MI355X (gfx950) engine for the `gffx intersect` hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  It is what a
 * Rust `gffx-hip` FFI crate would bind with `extern "C"` (binding shown in INTEGRATION.md).
 * Reference anchors are Baohua-Chen/GFFx v0.4.0, paths relative to its src/.
 *
 *   reference item (file:line)                          entry point here
 *   --------------------------------------------------  ---------------------------------------
 *   TreeIndexData / IntervalTree<u32> per seqid         gffx_hip_index_create / _destroy
 *     (utils/tree_index.rs:12-16, utils/tree.rs:5-23)     (device-resident sorted SoA + AoS)
 *   IntervalTree::query_interval (utils/tree.rs:98-121) \
 *   query_features + OverlapMode predicates + invert     > gffx_hip_batch_run / _wait and the
 *     (commands/intersect.rs:105-169, :73-78, :145-161) /   one-shot gffx_hip_query_features
 *   unique-root collection (commands/intersect.rs:598-615)  GFFX_OUT_ROOT_BITMAP
 *   gff_line_overlaps_queries, numeric part             gffx_hip_lines_* (Join B)
 *     (commands/intersect.rs:500-521)
 *   compute_hit_depth / compute_root_depth              gffx_hip_depth_* (`gffx depth`, BED source)
 *     (commands/depth.rs:121-293)
 *   merge_intervals + the two-pointer walk              gffx_hip_segments_covered (`gffx coverage`, BED source)
 *     (commands/coverage.rs:92-124, :339-364)
 *
 * Semantics (bit-exact with the reference):
 *   a root interval iv of the query's seqid is a HIT iff  iv.start < q.end && iv.end > q.start
 *   (strict, half-open; u32 compares; q.start >= q.end rows are legal and evaluated as-is);
 *   keep = mode predicate(iv, q);  a (query, root) pair is emitted iff  invert ^ keep.
 * Output order: the pairs of one query are contiguous (its segment); WHERE the segments lie and the order inside a segment
 * depend on the strategy, and only the multiset of pairs per query is the contract (the reference's own order is that of an
 * FxHashMap walk plus a tree DFS and is unspecified as well):
 *   DIRECT                 segments in INPUT order (CSR: offsets = exclusive prefix of the counts); inside a segment
 *                          descending position in the seqid's start-sorted list (ties in start: later builder order first)
 *   WINDOWS / FUSED        segments in the order in which rounds (2048 or 4096 regions) reserve them (differs from run to run);
 *                          inside a round the runs of its groups of 256 regions in the order the waves arrive, the regions
 *                          of a group in input order; every query's segment start is given explicitly (GFFX_OUT_OFFSETS /
 *                          GFFX_OUT_OFFSETS32) or per group (GFFX_OUT_SEGBASE) and the segments tile [0, pairs) exactly.  Inside a segment:
 *                          WINDOWS ascending list order (start, ties in builder order) for regions answered from the
 *                          window's candidate list (the wide form: the inline entries of the line of the region's first
 *                          base, then the run of roots that start inside the region, then the entries of that line's
 *                          list that continue in the spill records -- these start at or before the region, so a wide-form
 *                          segment is NOT ascending by start), descending for regions that took the exact sweep; FUSED descending
 *   SORTED (partitioned)   segments in the order genome tiles were served, offsets explicit, inside a segment descending
 *
 * Threading: an index is immutable after creation and may be shared by threads; a batch owns
 * one HIP stream plus its buffers and must not be used from two threads at once.
 * Errors: every function returns GFFX_OK (0) or a negative gffx_status; the message of the
 * last failure on the calling thread is in gffx_hip_last_error().  Nothing throws or aborts
 * across this boundary.  There is NO CPU fallback: without a HIP device every compute entry
 * point fails with GFFX_E_NO_DEVICE.
 *
 * Tuning knobs (results never depend on any of them).  Each is an environment variable that is read ONCE per object -- the
 * index builders' when gffx_hip_index_create runs (a clone inherits them), the passes' when gffx_hip_batch_create runs -- and a
 * batch's can be changed afterwards with gffx_hip_batch_set_option(batch, "WIN_THREADS", 512).  No launch reads the environment.
 * gffx_hip_index_options / gffx_hip_batch_options report the values that differ from the defaults as a JSON object
 * (`gffx ... --stats-json` and bench.py's `config.knobs` carry it).
 *   passes (batch):
 *   GFFX_HIP_WIN_THREADS=0|512|1024 block width of the window kernels (0 = the engine's choice: 1024 for a pass of >= 500 000 regions
 *                                   that runs alone, 512 otherwise).  Batches sorted by position handed over in groups on ONE stream
 *                                   (GFFX_HIP_GROUP=1) are 25 % faster at 1024: their gene-dense wave rounds overflow the 512-thread
 *                                   kernel's smaller strips (on two streams: 1.5 %; profiles/r06_block_width_on_sorted_batches.txt)
 *   GFFX_HIP_WIN_WIDE=0|1|2         the mixed form of the window kernels: never / AUTO's choice for batches with wide rows (default) /
 *                                   every eligible pass of the windows strategy
 *   GFFX_HIP_WIDTH_SAMPLE=0         no width sample of the rows the host hands over (AUTO then learns from a first waited pass)
 *   GFFX_HIP_GROUP=0|1|2|3          gffx_hip_batches_run_n with four batches or more: one launch per GROUP of batches, the groups on 1 / 2
 *                                   (default) / 3 streams of the index; 0: always pass by pass; 1 also groups two or three batches
 *   GFFX_HIP_TICKETS=0..4           how a launch's blocks get their rounds: 0 a fixed stride, 1 the launch's tail by ticket, 2 every round by
 *                                   ticket, 3 only the rounds beyond the last full stride, 4 (default) the engine's choice (1 for a launch that
 *                                   serves one batch with eight or more rounds per block, else 0)
 *   GFFX_HIP_AUTO_STRATEGY=n        what GFFX_STRATEGY_AUTO resolves to (0: the engine's choice)
 *   launch sizes (0 = the engine's choice): GFFX_HIP_FUSED_BLOCKS, GFFX_HIP_BITMAP_BLOCKS; GFFX_HIP_JOIN_BLOCKS, GFFX_HIP_MAX_BLOCKS;
 *   GFFX_HIP_PARTITION_BUDGET_MB (record buffers of the partitioned strategy)
 *   index build: GFFX_HIP_SLOT_WMAX (widest region a window line answers, 16384), GFFX_HIP_WIN_PER_ENTRY, GFFX_HIP_WIN_SPLIT,
 *   GFFX_HIP_WIN_FILTER_KB, GFFX_HIP_WIN_MAX_LINES, GFFX_HIP_BINS_PER_ENTRY
 */
#ifndef GFFX_HIP_H
#define GFFX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFFX_HIP_ABI_VERSION 1

/* commands/intersect.rs:73-78 OverlapMode */
enum gffx_mode { GFFX_MODE_CONTAINED = 0, GFFX_MODE_CONTAINS_REGION = 1, GFFX_MODE_OVERLAP = 2 };

enum gffx_status {
    GFFX_OK = 0,
    GFFX_E_INVALID = -1,   /* bad argument */
    GFFX_E_NO_DEVICE = -2, /* no HIP device / device index out of range */
    GFFX_E_HIP = -3,       /* a HIP runtime call failed */
    GFFX_E_OOM = -4,       /* host or device allocation failed */
    GFFX_E_CHR_RANGE = -5, /* a query's chr >= n_chr (the reference panics: intersect.rs:117) */
    GFFX_E_STATE = -6      /* call order violated (e.g. results read before _wait) */
};

/* what a batch run materialises on the device */
enum gffx_out {
    GFFX_OUT_COUNTS = 1,      /* u32 kept pairs per query, input order (always produced) */
    GFFX_OUT_FIDS = 2,        /* u32 root_fid per kept pair, CSR order */
    GFFX_OUT_TRIPLES = 4,     /* (root_fid, iv.start, iv.end) per kept pair == the reference's
                                 Vec<(u32,u32,u32)> (intersect.rs:162) */
    GFFX_OUT_ROOT_BITMAP = 8, /* one bit per root of the index: set iff the root is in >=1 kept pair */
    GFFX_OUT_OFFSETS = 16,    /* u64 start of every query's pair segment (nq+1 entries; [nq] = number of pairs).
                                 Direct strategy: the exclusive prefix of the counts (CSR in input order).
                                 Partitioned strategy: segments follow the order in which genome tiles were
                                 served, so the offsets are explicit -- they still tile [0, pairs) exactly. */
    GFFX_OUT_OFFSETS32 = 64,  /* the same segment starts as u32 (nq entries), for callers that know the pass keeps
                                 fewer than 2^32 pairs (the engine checks: _wait fails with GFFX_E_INVALID otherwise);
                                 halves the bytes the offsets cost.  Windows strategy only (AUTO picks it). */
    GFFX_OUT_BITMAP_KEEP = 128, /* with GFFX_OUT_ROOT_BITMAP: do not clear the batch's bitmap first -- a caller that streams a
                                 BED file chunk by chunk through one batch accumulates the unique roots of all chunks
                                 (commands/intersect.rs:598-615 dedups over the whole file) */
    GFFX_OUT_SEGBASE = 256,   /* windows strategy (AUTO picks it): per GROUP of 256 consecutive regions (regions 256 g .. 256 g + 255,
                                 a wave's share of a round) the u64 start of the group's run of pairs -- ceil(nq / 256)
                                 entries, 8 bytes per 256 regions.  The segments of a group's regions follow each other in
                                 input order, so region i's segment starts at segbase[i / 256] + the counts of the group's
                                 regions before i: a consumer that reads the counts anyway needs no per-region offsets
                                 (they are not part of the algorithmic bytes).  The runs tile [0, pairs) exactly. */
    GFFX_OUT_NO_COUNTS = 512, /* with GFFX_OUT_ROOT_BITMAP alone (windows strategy): the per-region counts need not be written -- a
                                 caller that only wants the unique roots (the CLI: commands/intersect.rs:598-615) saves their 4 bytes
                                 per region */
    GFFX_OUT_EMIT_ORDER = 32  /* partitioned strategy: leave the per-query results in emission order
                                 ({input row, count, offset} records: gffx_hip_batch_copy_query_records) and
                                 skip the scatter into input-order arrays; _copy_counts / _copy_offsets then
                                 run that scatter on demand */
};

enum gffx_strategy {
    GFFX_STRATEGY_AUTO = 0,   /* the engine picks: windows; fused for a batch of mostly wide regions */
    GFFX_STRATEGY_DIRECT = 1, /* queries in input order; bin directory + gathers from the L2-resident index */
    GFFX_STRATEGY_SORTED = 2, /* "partitioned": one-pass device radix partition of the batch by genome window,
                                 then a fused count+emit join served from LDS-staged index tiles */
    GFFX_STRATEGY_FUSED = 3,  /* queries in input order, ONE kernel: interleaved gathers from the L2-resident
                                 index, count + emit per block round; counts / offsets in input order, pair
                                 segments in the order rounds reserve them (offsets explicit) */
    /* 4 was GFFX_STRATEGY_SLOTS (round 1's slot index, superseded by WINDOWS and removed in round 3: _run rejects it) */
    GFFX_STRATEGY_WINDOWS = 5, /* AUTO's choice.  One 32-byte index LINE per region: the line of the genome window
                                 (<= 2^15 bp) the region ends in lists up to 4 candidate roots {start, end as 16-bit
                                 window-relative coordinates, root_fid}; longer lists, dense windows, wide and
                                 empty-width regions are deferred inside the round and served exactly (list tail /
                                 skip-link sweep).  Root-bitmap passes set bits in an LDS-private bitmap.  Output
                                 contract as FUSED.  Domain: end >= start for every root (an interval with end < start
                                 never leaves the reference's IntervalTree::build, utils/tree.rs:48-50). */
};

enum gffx_kernel_id { /* for gffx_hip_batch_kernel_ms */
    GFFX_K_JOIN_COUNT = 0,
    GFFX_K_JOIN_EMIT = 1,
    GFFX_K_SORT = 2,
    GFFX_K_LINES = 3,
    GFFX_K_FUSED = 4,
    GFFX_K_UNPERMUTE = 5,
    GFFX_K_FUSED_DIRECT = 6,
    GFFX_K_DEPTH = 7,
    GFFX_K_SLOTS = 8, /* (retired with the slots strategy; the number stays reserved) */
    GFFX_K_WINDOWS = 9,    /* k_join_roots: the windows strategy's root passes */
    GFFX_K_BITMAP_OR = 10, /* (k_bitmap_fold runs at gffx_hip_batch_wait, outside the profiled passes; the number stays reserved) */
    GFFX_K_WAVE = 11,      /* k_join_pairs: the windows strategy's pair passes (counts + root_fids / positions) */
    GFFX_K__COUNT = 12
};

typedef struct gffx_hip_index gffx_hip_index;
typedef struct gffx_hip_batch gffx_hip_batch;
typedef struct gffx_hip_lines gffx_hip_lines;
typedef struct gffx_hip_regions gffx_hip_regions;
typedef struct gffx_hip_depth gffx_hip_depth;

int gffx_hip_abi_version(void);
/* number of visible HIP devices (0 when none / no driver); never fails */
int gffx_hip_device_count(void);
/* optional: pay the process's one-off HIP costs (runtime and context creation, code-object load) now -- a host
 * can call it on a side thread while it still parses its inputs.  Nothing depends on it. */
int gffx_hip_warmup(int device);
/* thread-local, valid until the next failing call on this thread */
const char *gffx_hip_last_error(void);

/* ---- index: replaces TreeIndexData.chr_entries (utils/tree_index.rs:12-16) ----------------
 * chr_offsets has n_chr+1 entries delimiting each seqid's root intervals in start/end/root_fid
 * (0-based half-open coordinates as the builder stores them, index_builder/core.rs:108-109;
 * any order inside a seqid -- the builder's file order is fine).  The library sorts each seqid
 * stably by start, adds the running maximum of `end`, builds the bin directory and uploads
 * everything to `device` once. */
int gffx_hip_index_create(uint32_t n_chr, const uint32_t *chr_offsets, const uint32_t *start,
                          const uint32_t *end, const uint32_t *root_fid, int device,
                          gffx_hip_index **out);
void gffx_hip_index_destroy(gffx_hip_index *);
uint32_t gffx_hip_index_n_chr(const gffx_hip_index *);
/* the index builders' knobs that are not at their defaults, as a JSON object ("{}": all defaults).  snprintf's contract: returns the
 * length the text needs; at most cap - 1 bytes and a NUL are written (buf may be NULL with cap 0). */
int gffx_hip_index_options(const gffx_hip_index *, char *buf, size_t cap);
uint64_t gffx_hip_index_n_roots(const gffx_hip_index *);
int gffx_hip_index_device(const gffx_hip_index *);
/* root_fid of the i-th bit of the root bitmap (i < n_roots), host array owned by the index */
const uint32_t *gffx_hip_index_sorted_fids(const gffx_hip_index *);

/* The same index on another device of the node (multi-GPU hosts replicate the index: the host-side build runs once,
 * the device arrays are copied GPU to GPU). */
int gffx_hip_index_clone(const gffx_hip_index *, int device, gffx_hip_index **out);

/* ---- region stores: streaming BED ingestion (commands/intersect.rs:201-230 parses the whole file into one Vec; a host
 * that parses chunk by chunk hands every chunk over through one of two PINNED staging buffers while it parses the next) --
 * A store keeps regions in HBM as AoS triples (chr, start, end).  keep_all = 1: every appended chunk stays (capacity_rows
 * in total; Join B needs all regions of the run: gffx_hip_lines_test_store); keep_all = 0: a ring of two chunk slots, the
 * chunk appended from staging buffer k overwrites slot k (the caller has waited for the batch that read it). */
int gffx_hip_regions_create(int device, uint64_t capacity_rows, uint64_t chunk_rows, int keep_all, gffx_hip_regions **out);
void gffx_hip_regions_destroy(gffx_hip_regions *);
/* pinned host buffer k (0 or 1), chunk_rows rows of 3 u32 */
uint32_t *gffx_hip_regions_staging(gffx_hip_regions *, int k);
/* block until the last append from staging buffer k has left the host buffer (it may then be refilled) */
int gffx_hip_regions_wait_staging(gffx_hip_regions *, int k);
/* asynchronous H2D copy of the first n_rows rows of staging buffer k to the store */
int gffx_hip_regions_append(gffx_hip_regions *, int k, uint64_t n_rows);
/* the same for a chunk that sits in staging buffer k in n_parts pieces (rows stage_first[p] .. + n_rows[p]: parser threads
 * fill disjoint parts of the buffer); the pieces land back to back in the store and count as ONE append */
int gffx_hip_regions_append_parts(gffx_hip_regions *, int k, uint32_t n_parts, const uint64_t *stage_first, const uint64_t *n_rows);
uint64_t gffx_hip_regions_rows(const gffx_hip_regions *); /* keep_all stores: rows appended so far */
/* The batch borrows rows [first, first + n_rows) of the chunk last appended from staging buffer k (no copy; the batch's
 * stream waits for that append).  They must stay untouched until the batch's pass has finished. */
int gffx_hip_batch_set_regions_store(gffx_hip_batch *, const gffx_hip_regions *, int k, uint64_t first, uint64_t n_rows);

/* ---- multi-GPU exchange step: all-gather of per-device {regions, kept pairs} over RCCL (xGMI), one rank per device,
 * single process (ncclCommInitAll).  counts_in: 2 values per device; counts_out: n_dev x (2 x n_dev) -- what every
 * device holds after the collective (all rows are equal).  `devices` must be distinct. */
int gffx_hip_allgather_counts(int n_dev, const int *devices, const uint64_t *counts_in, uint64_t *counts_out);

/* ---- query batches: replaces query_features (commands/intersect.rs:105-169) --------------- */
int gffx_hip_batch_create(const gffx_hip_index *, uint64_t max_queries, gffx_hip_batch **out);
void gffx_hip_batch_destroy(gffx_hip_batch *);

/* regions = nq AoS triples (chr, start, end): exactly the reference's &[(u32,u32,u32)]
 * (intersect.rs:107).  Copied host -> device on the batch's stream. */
int gffx_hip_batch_set_regions_host(gffx_hip_batch *, const uint32_t *regions, uint64_t nq);
/* same, from three host arrays */
int gffx_hip_batch_set_regions_soa_host(gffx_hip_batch *, const uint32_t *chr,
                                        const uint32_t *start, const uint32_t *end, uint64_t nq);
/* borrow three DEVICE arrays (SoA) already resident in HBM; they must stay valid until _wait */
int gffx_hip_batch_set_regions_device(gffx_hip_batch *, const uint32_t *d_chr,
                                      const uint32_t *d_start, const uint32_t *d_end, uint64_t nq);

/* Enqueue one pass of Join A over the batch on its stream (asynchronous). */
int gffx_hip_batch_run(gffx_hip_batch *, int mode, int invert, uint32_t out_flags, int strategy);
/* Block until the pass finished; grows the output buffers and replays the emit step when the
 * kept-pair count exceeded their capacity.  Returns GFFX_E_CHR_RANGE if a chr was out of range. */
int gffx_hip_batch_wait(gffx_hip_batch *);
/* Cheaper than _wait for pipelines: only checks that the stream drained (no capacity replay). */
int gffx_hip_batch_sync(gffx_hip_batch *);

uint64_t gffx_hip_batch_n_queries(const gffx_hip_batch *);
/* valid after _wait */
uint64_t gffx_hip_batch_total_hits(const gffx_hip_batch *);
/* After _wait: the kept pairs of ALL root passes (GFFX_OUT_ROOT_BITMAP alone, windows strategy) since the batch's last pass without
 * GFFX_OUT_BITMAP_KEEP -- what a caller that streams a BED file chunk by chunk through one batch reports per device
 * (commands/intersect.rs:105-169 returns the pairs of the whole file; the hit-count exchange of `gffx intersect --gpus N` carries
 * this number).  For any other last pass: the pass's own total (= gffx_hip_batch_total_hits). */
int gffx_hip_batch_kept_pairs_accumulated(gffx_hip_batch *, uint64_t *out);
int gffx_hip_batch_copy_counts(gffx_hip_batch *, uint32_t *host /* nq */);
int gffx_hip_batch_copy_offsets(gffx_hip_batch *, uint64_t *host /* nq+1 */);
int gffx_hip_batch_copy_offsets32(gffx_hip_batch *, uint32_t *host /* nq */); /* GFFX_OUT_OFFSETS32 */
int gffx_hip_batch_copy_segbase(gffx_hip_batch *, uint64_t *host /* ceil(nq / 256) */); /* GFFX_OUT_SEGBASE */
/* per-query records in emission order: rows[i] = input row, counts[i] = kept pairs, offsets[i] = start of
 * its segment in fids / triples (needs GFFX_OUT_OFFSETS); any pointer may be NULL; nq entries each */
int gffx_hip_batch_copy_query_records(gffx_hip_batch *, uint32_t *rows, uint32_t *counts, uint64_t *offsets);
int gffx_hip_batch_copy_fids(gffx_hip_batch *, uint32_t *host /* total_hits */);
int gffx_hip_batch_copy_triples(gffx_hip_batch *, uint32_t *host /* 3*total_hits */);
/* n_words = ceil(n_roots/64); bit i <-> gffx_hip_index_sorted_fids()[i] */
int gffx_hip_batch_copy_root_bitmap(gffx_hip_batch *, uint64_t *host, uint64_t n_words);
/* device views of the same buffers (NULL if not produced), for zero-copy consumers */
const uint32_t *gffx_hip_batch_device_counts(const gffx_hip_batch *);
const uint32_t *gffx_hip_batch_device_fids(const gffx_hip_batch *);
const uint64_t *gffx_hip_batch_device_offsets(const gffx_hip_batch *);
const uint32_t *gffx_hip_batch_device_offsets32(const gffx_hip_batch *);
const uint64_t *gffx_hip_batch_device_segbase(const gffx_hip_batch *);
const uint32_t *gffx_hip_batch_device_triples(const gffx_hip_batch *);
/* the batch's own device copy of the regions as AoS triples (after _set_regions_host; NULL for SoA / borrowed regions) */
const uint32_t *gffx_hip_batch_device_regions(const gffx_hip_batch *);
/* pre-size the pair buffers (pairs); avoids the capacity replay on the first run */
int gffx_hip_batch_reserve_hits(gffx_hip_batch *, uint64_t n_pairs);
/* Tuning knobs of the batch's passes ("Tuning knobs" above): set one by its name (with or without the GFFX_HIP_ prefix, any
 * case); GFFX_E_INVALID for an unknown name or a value outside the knob's range.  _options: the non-default ones as JSON. */
int gffx_hip_batch_set_option(gffx_hip_batch *, const char *name, long value);
int gffx_hip_batch_options(const gffx_hip_batch *, char *buf, size_t cap);

/* HIP-event timing of the kernels on the batch's own stream.  While enabled every launch is
 * bracketed by events; _kernel_ms returns the accumulated milliseconds and launch count of
 * one kernel since the last _reset_profile (events are resolved by _wait/_sync). */
int gffx_hip_batch_set_profiling(gffx_hip_batch *, int enabled);
int gffx_hip_batch_kernel_ms(gffx_hip_batch *, int kernel_id, double *total_ms, uint64_t *launches);
int gffx_hip_batch_reset_profile(gffx_hip_batch *);
/* n passes back to back on the batch's stream between ONE pair of HIP events (blocking): total_ms / n = the average
 * launch-to-launch duration of a pass without an event pair per launch */
int gffx_hip_batch_timed_runs(gffx_hip_batch *, int mode, int invert, uint32_t out_flags, int strategy, uint32_t n,
                              double *total_ms);
/* n_passes passes enqueued round-robin over n_batches batches (batch i % n_batches takes pass i) in one call: a host that
 * keeps several batches in flight pays one FFI crossing for the lot (bench.py's timed region: the launch loop runs in C).
 * Round 6: consecutive passes over DISTINCT batches of one index that resolve to the same pass of the windows strategy are served
 * by ONE launch per group of up to 8 batches (every batch a share of the launch's blocks, its rounds handed out by ticket): the
 * index lines are fetched into the L2s once per launch instead of once per batch, ramp and drain are paid once.  The launch runs on
 * a stream of the index; a batch's own stream joins it whenever the batch is used on its own again (_run, _wait, _sync, a new set of
 * regions): per-batch order is what it always was.  Up to three batches run pass by pass (their launches share the chip as in round
 * 5); from four on the batches are cut into groups of equal size (<= 8), half of them on each of two such streams, so that one group's
 * drain overlaps the next one's ramp.  Results are exactly those of n_passes single _run calls.  Knob GFFX_HIP_GROUP of batches[0]
 * ("Tuning knobs" above). */
int gffx_hip_batches_run_n(gffx_hip_batch *const *batches, uint32_t n_batches, int mode, int invert, uint32_t out_flags,
                           int strategy, uint64_t n_passes);
/* How _batches_run_n cuts passes over these batches into launches: *groups per walk over the batches (0: pass by pass), the *largest
 * group's size, the *streams the groups alternate between (any pointer may be NULL). */
int gffx_hip_batches_plan(gffx_hip_batch *const *batches, uint32_t n_batches, uint32_t *groups, uint32_t *largest, uint32_t *streams);
/* n_launches times ONE pass over each of the n_batches (<= 8) batches -- the launch _batches_run_n issues for such a group --, back
 * to back between one pair of HIP events on the stream they run on (blocking): total_ms / n_launches = the duration of the launch.
 * *grouped (may be NULL) = 1 when the passes ran as one launch, 0 when they did not group (then: pass after pass on batches[0]'s
 * stream is NOT what was timed -- the figure is meaningless). */
int gffx_hip_batches_timed_runs(gffx_hip_batch *const *batches, uint32_t n_batches, int mode, int invert, uint32_t out_flags,
                                int strategy, uint32_t n_launches, double *total_ms, uint32_t *grouped);
/* Threads per block of the last windows-strategy pair pass of this batch (512 or 1024; 0: none ran).  The engine takes
 * 1024-thread blocks (one per CU, rounds of 4096 regions) for a batch of 500 000 regions or more while NO other batch of the
 * index has passes in flight, 512-thread blocks (two per CU: kernels of two batches share the CUs) otherwise;
 * The knob GFFX_HIP_WIN_THREADS (512 / 1024) forces one. */
uint32_t gffx_hip_batch_block_threads(const gffx_hip_batch *);
/* Blocks of that launch: as many as the device has slots for (256 blocks of 1024 threads, 512 of 512 threads), fewer for a small
 * batch (a block per round of 4 x threads regions) -- and 256 blocks of 512 threads (one per CU) for a pair pass over at most ~2 M
 * regions launched while TWO OR MORE other batches of the index have passes in flight: kernels of different streams run side by side only when each leaves
 * slots free, and that is where three batches in flight gain (measured; with one other batch in flight the full grid is better).
 * A root pass (GFFX_OUT_ROOT_BITMAP alone) takes one block per CU as soon as ONE other batch is in flight (measured likewise).
 * The knob GFFX_HIP_FUSED_BLOCKS forces a count (GFFX_HIP_BITMAP_BLOCKS: the root passes'). */
uint32_t gffx_hip_batch_block_count(const gffx_hip_batch *);
/* 1 when the last run's passes took the MIXED form of the window kernels (every mode, inverted or not; round 4's "wide form" is
 * its all-wide Overlap case): every region is served its own way in one launch -- a region the index lines answer (up to 16 Ki bases by default) from ONE
 * line as in the narrow form, a wider one from two index lines and two rank words, no sweep.  AUTO chooses it for a batch with
 * more than one wide row in 128: found by a sample of the rows gffx_hip_batch_set_regions_host / _soa_host are given (each judged
 * against its own seqid's line width), or -- regions already on the device -- by a previous waited pass of the narrow form, which
 * counts the rows its lines did not answer.  What a wide region keeps in each mode: join_pairs_kernels.hpp, pair_locate_mixed
 * (Contained: its run of roots filtered by their ends; ContainsRegion: the roots over its first base that reach its end).  The knob
 * GFFX_HIP_WIN_WIDE (0 never / 1 AUTO / 2 every eligible pass of the windows strategy) steers it. */
int gffx_hip_batch_wide_form(const gffx_hip_batch *);

/* One-shot drop-in for query_features (commands/intersect.rs:105-111): host regions in, host
 * triples out (malloc'd by the library, release with gffx_hip_free_host). */
int gffx_hip_query_features(const gffx_hip_index *, const uint32_t *regions, uint64_t nq, int mode,
                            int invert, uint32_t **triples_out, uint64_t *n_triples);
void gffx_hip_free_host(void *);

/* ---- Join B: the numeric core of gff_line_overlaps_queries (commands/intersect.rs:500-521) -
 * A line table is the (seqid number, raw column-4 start, raw column-5 end) of GFF lines,
 * in file order, uploaded once.  seq == UINT32_MAX marks a line that can never match (seqid
 * without queries / unparsable columns).  _test evaluates, for every line, whether ANY region
 * of the line's seqid satisfies the literal closed-interval predicate of the given mode
 * (no invert: intersect.rs:232-240 has no such parameter) and writes one byte per line. */
int gffx_hip_lines_create(int device, uint64_t n_lines, const uint32_t *seq, const uint32_t *start,
                          const uint32_t *end, gffx_hip_lines **out);
void gffx_hip_lines_destroy(gffx_hip_lines *);
/* regions = AoS triples as above (all regions of the run, BED order: intersect.rs:621-633);
 * n_seq = number of seqids; keep_host receives n_lines bytes (0/1). */
int gffx_hip_lines_test(gffx_hip_lines *, const uint32_t *regions, uint64_t nq, uint32_t n_seq,
                        int mode, uint8_t *keep_host);
/* the same with the regions already in HBM as AoS triples (e.g. gffx_hip_batch_device_regions of the batch Join A ran on) */
int gffx_hip_lines_test_device(gffx_hip_lines *, const uint32_t *d_regions, uint64_t nq, uint32_t n_seq,
                               int mode, uint8_t *keep_host);
/* ... or in a region store (keep_all = 1): all rows appended so far */
int gffx_hip_lines_test_store(gffx_hip_lines *, const gffx_hip_regions *, uint32_t n_seq, int mode, uint8_t *keep_host);
/* HIP-event duration (ms) of the line kernel (k_lines_exists2; k_lines_exists when regions with start > end exist) in the last
 * _test call, on the table's own stream */
double gffx_hip_lines_last_kernel_ms(const gffx_hip_lines *);
/* ... and of the device preparation of the region tables before it: radix sort by (seqid, start), running max / min of the
 * ends, the bin directory and -- Overlap mode, when the run has regions with start > end -- the sort of those regions' ends
 * (the reference builds `query_ivmap` on the CPU: intersect.rs:621-633) */
double gffx_hip_lines_last_prep_ms(const gffx_hip_lines *);
/* radix passes of the last call's region sort by (seqid, start): 4 when the mixed-radix top digit (seqid, start >> 24) fits 256 values
 * (GRCh38-scale BED files), else 4 + one per byte of the seqid count */
int gffx_hip_lines_last_sort_passes(const gffx_hip_lines *);
/* The region tables of the last _test, for parity checks: q_off (n_seq + 1 entries), then per region in (seqid, start)
 * order (stable: equal starts keep the order of `regions`): QS = start, PM = running max of `end` inside the seqid, SM =
 * running min of `end` from the seqid's last region backwards, CD = regions with start > end before this position (all
 * seqids).  Any pointer may be NULL. */
int gffx_hip_lines_copy_tables(gffx_hip_lines *, uint64_t *q_off, uint32_t *qs, uint32_t *pm, uint32_t *sm, uint32_t *cd);
/* ... the bin directory over QS: d_off (n_seq + 1), shift_nb ({shift, bins} per seqid), dir_qs (d_off[n_seq] entries;
 * dir[d_off[c] + b] = first position of seqid c whose start >= b << shift, entry `bins` = the seqid's end) ... */
int gffx_hip_lines_copy_dirs(gffx_hip_lines *, uint64_t *d_off, uint32_t *shift_nb, uint32_t *dir_qs);
/* ... and, after an Overlap-mode _test, the regions with start > end: their number, dq_off (n_seq + 1) and their ends
 * sorted per seqid (de: *n_deg entries; call with de = NULL first to learn the size). */
int gffx_hip_lines_copy_degenerate(gffx_hip_lines *, uint64_t *n_deg, uint64_t *dq_off, uint32_t *de);

/* ---- `gffx depth` with a BED source: compute_hit_depth / compute_root_depth (commands/depth.rs:121-293) --
 * The host parses every root BLOCK once (the byte range of a .gof record; for a root_fid with several
 * records the LAST one, index_loader/gof.rs:32-37) into the lines that carry an ID, with the 0-based
 * half-open coordinates of depth.rs:145-147.  Inside a block the lines are ordered by ID; a GROUP is one
 * (block, ID) -- the unit depth.rs:202-206 dedups on -- numbered globally.  block_of_fid[root_fid] is
 * the block (UINT32_MAX: no usable record, depth.rs:242-243).
 * _accumulate adds the regions of a finished Join A pass (Overlap, no invert, GFFX_OUT_FIDS |
 * GFFX_OUT_OFFSETS): per group the number of regions with an overlapping line (a region counts a root
 * once, depth.rs:241) and the min start / max end of the overlapped lines (UINT32_MAX / 0 when none). */
int gffx_hip_depth_create(int device, uint32_t n_groups, uint32_t n_blocks, const uint64_t *block_line_off /* n_blocks+1 */,
                          const uint32_t *line_start, const uint32_t *line_end, const uint32_t *line_group,
                          uint32_t n_fid, const uint32_t *block_of_fid, gffx_hip_depth **out);
void gffx_hip_depth_destroy(gffx_hip_depth *);
int gffx_hip_depth_accumulate(gffx_hip_depth *, gffx_hip_batch *);
int gffx_hip_depth_reset(gffx_hip_depth *);
int gffx_hip_depth_copy(gffx_hip_depth *, uint64_t *depth, uint32_t *min_start, uint32_t *max_end /* n_groups each */);

/* ---- `gffx coverage` with a BED source: the covered bases of feature segments (commands/coverage.rs:339-364) --
 * covered_out[i] = |[seg_start[i], seg_end[i])  ∩  union of the regions of seqid seg_seq[i]|  (regions = AoS
 * (chr, start, end) rows with start < end; touching regions merge like merge_intervals, coverage.rs:92-109).
 * For a segment inside its root's interval this equals the reference's per-root figure: a region that overlaps
 * the segment then hits the root.  Segments sticking out of their root are the caller's to handle. */
int gffx_hip_segments_covered(int device, uint64_t n_seg, const uint32_t *seg_seq, const uint32_t *seg_start,
                              const uint32_t *seg_end, const uint32_t *regions, uint64_t nq, uint32_t n_seq,
                              uint32_t *covered_out);

#ifdef __cplusplus
}
#endif
#endif /* GFFX_HIP_H */
