/*
 * gffx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, CPU-only restatement of the reference algorithm for the
 * `gffx intersect` hot path (Baohua-Chen/GFFx v0.4.0).  It exists so that the
 * HIP path can be checked against something that follows the reference source
 * text line by line.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * (no #[test], no tests/ dir, benchmark/ submodule empty) and cannot be built
 * here (Rust; no cargo/rustc in the image).  The pins this oracle has are
 * (i) the hand-derived known-answer table in tests/golden/appendix_e.json,
 * (ii) an independent Python restatement (oracle/gffx_oracle_py.py) and
 * (iii) a brute-force O(R*Q) evaluation of the predicates.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src).
 */
#ifndef GFFX_ORACLE_H
#define GFFX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* commands/intersect.rs:73-78 */
enum { ORACLE_MODE_CONTAINED = 0, ORACLE_MODE_CONTAINS_REGION = 1, ORACLE_MODE_OVERLAP = 2 };

typedef struct oracle_index oracle_index; /* utils/tree_index.rs:12-16 TreeIndexData */

/* Build per-seqid centered interval trees (utils/tree.rs:30-64) from the
 * tree inputs in builder order (index_builder/core.rs:177-180).
 * chr_offsets has n_chr+1 entries into start/end/fid. */
oracle_index *oracle_index_from_roots(uint32_t n_chr, const uint32_t *chr_offsets,
                                      const uint32_t *start, const uint32_t *end,
                                      const uint32_t *fid);
void oracle_index_free(oracle_index *);
uint32_t oracle_index_n_chr(const oracle_index *);
uint64_t oracle_index_n_roots(const oracle_index *);

/* commands/intersect.rs:105-169 query_features.  regions = nq AoS (chr,start,end).
 * triples_out (malloc'd, free with oracle_free) = (root_fid, iv.start, iv.end) per
 * kept pair, seqids ascending, BED order inside a seqid, tree DFS order inside a
 * query.  counts (optional, nq entries, input order) = kept pairs per query.
 * Returns 0, or -1 if a chr is out of range (the reference panics there). */
int oracle_query_features(const oracle_index *, const uint32_t *regions, uint64_t nq, int mode,
                          int invert, uint32_t **triples_out, uint64_t *n_triples,
                          uint32_t *counts);

/* Brute force O(R*Q) evaluation of tree.rs:110 + intersect.rs:145-161; same outputs
 * (triples in index order inside a query).  Independent of the tree. */
int oracle_query_features_brute(uint32_t n_chr, const uint32_t *chr_offsets, const uint32_t *start,
                                const uint32_t *end, const uint32_t *fid, const uint32_t *regions,
                                uint64_t nq, int mode, int invert, uint32_t **triples_out,
                                uint64_t *n_triples, uint32_t *counts);

/* commands/intersect.rs:441-523 gff_line_overlaps_queries on one raw line
 * (no trailing '\n').  Queries of the line's own seqid are found through
 * (seq_names[i] -> qoff[i]..qoff[i+1]) into qs/qe. */
int oracle_gff_line_overlaps_queries(const uint8_t *line, size_t len, uint32_t n_seq,
                                     const char *const *seq_names, const uint64_t *qoff,
                                     const uint32_t *qs, const uint32_t *qe, int mode);

/* The numeric core of intersect.rs:500-521 for one (start,end) against a query list. */
int oracle_line_predicate(uint32_t start, uint32_t end, const uint32_t *qs, const uint32_t *qe,
                          uint64_t nq, int mode);

/* index_builder/core.rs:41-242 build_index: writes <gff>.{fts,prt,a2f,atn,sqs,gof,rit,rix}.
 * Returns 0 or -1 with a message in err. */
int oracle_build_index(const char *gff_path, const char *attr_key, const char *skip_types,
                       int verbose, char *err, size_t errlen);

/* utils/tree_index.rs:21-34 load_tree_index, through the bypass of SURVEY App. A.3
 * (.sqs + .gof + the root lines of the GFF) -- 1:1 with the builder's tree inputs. */
int oracle_load_tree_index(const char *gff_path, oracle_index **out, char *err, size_t errlen);
/* Same, but by reading .rit/.rix in the (hypothesised, unpinned) bincode layout. */
int oracle_load_tree_index_rit(const char *gff_path, oracle_index **out, char *err, size_t errlen);
/* Flatten the index back to arrays (builder order). Caller frees with oracle_free. */
int oracle_index_export(const oracle_index *, uint32_t **chr_offsets, uint32_t **start,
                        uint32_t **end, uint32_t **fid);
const char *oracle_index_seq_name(const oracle_index *, uint32_t i);

/* commands/intersect.rs:201-230 parse_bed_file / :172-198 parse_region.
 * regions_out = malloc'd AoS triples. */
int oracle_parse_bed_file(const char *bed_path, const oracle_index *, uint32_t **regions_out,
                          uint64_t *nq, char *err, size_t errlen);
int oracle_parse_region(const char *region, const oracle_index *, uint32_t out[3], char *err,
                        size_t errlen);

/* commands/intersect.rs:541-655 run: the whole command; output bytes go to out_path.
 * Exactly one of region / bed_path is non-NULL.  types may be NULL.
 * Returns 0, or 1 (the reference's `Error: ...` exit code) with the message in err. */
int oracle_intersect_run(const char *gff_path, const char *region, const char *bed_path, int mode,
                         int invert, int entire_group, const char *types, const char *out_path,
                         char *err, size_t errlen);

/* commands/depth.rs:429-495: the BED rows `gffx depth` keeps (its rules differ from intersect's:
 * malformed rows, s >= e and unknown seqids are dropped silently). */
int oracle_depth_parse_bed(const char *bed_path, const oracle_index *, uint32_t **regions_out,
                           uint64_t *nq, char *err, size_t errlen);
/* commands/depth.rs:548-635 run with a .bed source: per feature ID the number of regions overlapping
 * a line of that ID (deduped per region inside a root's block), with min start / max end of the
 * overlapped lines.  Rows sorted by id (the reference's order is a hash-map walk). */
int oracle_depth_run(const char *gff_path, const char *bed_path, const char *out_path, char *err,
                     size_t errlen);

/* commands/coverage.rs:487-582 run with a .bed source: per feature ID of every root block hit by a region the
 * covered bases (breadth) = |union of the regions that hit the root  ∩  union of the ID's lines|, the extent
 * over ALL lines of the ID in those blocks, and breadth / extent with six decimals.  Rows sorted by id. */
int oracle_coverage_run(const char *gff_path, const char *bed_path, const char *out_path, char *err,
                        size_t errlen);

void oracle_free(void *);

#ifdef __cplusplus
}
#endif
#endif
