/*
 * gffx_host.h -- C shims over the C++ host side (gffx_amd/csrc/host) so that the host logic of
 * the intersect path -- index building, side-car loading, BED/region parsing, block lookup, the
 * entire-group writer -- can be exercised without a GPU (tests/test_host_cpu.py).  None of these
 * touch the device; the joins live behind include/gffx_hip.h.
 * Every function returns 0, or -1 with a message in err (the text `gffx` prints after "Error: ").
 * Arrays returned through pointers are malloc'd; release them with gffx_host_free.
 */
#ifndef GFFX_HOST_H
#define GFFX_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* index_builder/core.rs:41-242 */
int gffx_host_build_index(const char *gff, const char *attr_key, const char *skip_types, int verbose,
                          char *err, size_t errlen);
/* utils/tree_index.rs:21-34: names = '\n'-joined seqid names; interval arrays in builder order */
int gffx_host_load_tree_index(const char *gff, uint32_t *n_chr, uint32_t **chr_offsets, uint32_t **start,
                              uint32_t **end, uint32_t **root_fid, char **names, char *err, size_t errlen);
/* commands/intersect.rs:201-230 / :172-198 (the seqid map comes from <gff>.sqs) */
int gffx_host_parse_bed_file(const char *gff, const char *bed, uint32_t **regions, uint64_t *n_regions,
                             char *err, size_t errlen);
/* the same rows through the chunked parser of the streaming CLI (chunks of chunk_bytes cut at line starts, four pieces per
 * thread on persistent workers, recycled row buffers): host/intersect.cpp::stream_unique_roots' producer, without a device */
int gffx_host_parse_bed_file_chunked(const char *gff, const char *bed, uint32_t threads, uint64_t chunk_bytes, uint32_t **regions,
                                     uint64_t *n_regions, char *err, size_t errlen);
/* The host half of `gffx intersect --gpus n_dev` without a device (stream_unique_roots' parser pool + per-chunk bucket scatter):
 * rows = device 0's rows, then device 1's, ... (dev_rows[d] rows of 3 words each); keep_all: device 0 receives every row. */
int gffx_host_shard_bed_file(const char *gff, const char *bed, uint32_t threads, uint64_t chunk_bytes, uint32_t n_dev, int keep_all,
                             uint32_t **rows, uint64_t *dev_rows /* n_dev */, char *err, size_t errlen);
int gffx_host_parse_region(const char *gff, const char *region, uint32_t out[3], char *err, size_t errlen);
/* index_loader/gof.rs:54-128: offsets[2*i], offsets[2*i+1] = block of roots[i] (UINT64_MAX = missing) */
int gffx_host_roots_to_offsets(const char *gff, const uint32_t *roots, uint64_t n, uint64_t *offsets,
                               char *err, size_t errlen);
/* utils/common.rs:188-287: blocks = n x (fid, start, end) as u64 triples */
int gffx_host_write_gff_output(const char *gff, const uint64_t *blocks, uint64_t n, const char *out_path,
                               char *err, size_t errlen);
/* commands/intersect.rs:80-102 with types = comma list; returns 0/1 */
int gffx_host_gff_type_allowed(const char *line, size_t len, const char *types);
/* commands/intersect.rs:446-494: returns 1 and fills seq offsets/len + raw start/end, else 0 */
int gffx_host_split_line(const char *line, size_t len, size_t *seq_len, uint32_t *start, uint32_t *end);
/* commands/depth.rs:450-495: the BED rows `gffx depth` keeps (seqid map from <gff>.sqs) */
int gffx_host_depth_parse_bed(const char *gff, const char *bed, uint32_t **regions, uint64_t *n_regions,
                              char *err, size_t errlen);
/* commands/depth.rs:131-152 over every root block: the device line table of include/gffx_hip.h
 * ("gffx depth") plus, per group, the id (index into ids, '\n'-joined) and the chrom text ('\n'-joined) */
int gffx_host_depth_block_table(const char *gff, uint32_t *n_blocks, uint64_t **block_line_off, uint64_t *n_lines,
                                uint32_t **line_start, uint32_t **line_end, uint32_t **line_group, uint32_t *n_fid,
                                uint32_t **block_of_fid, uint32_t *n_groups, uint32_t **group_id, char **group_chrom,
                                char **ids, char *err, size_t errlen);
/* the all-line SoA image `<gff>.lsoa` written by `gffx index` ("cached GPU-SoA side-car"):
 * 1 = it loads and equals a fresh parse on `threads` host threads, 0 = not usable (absent / stale / does not
 * validate; the reason in err), < 0 = error or mismatch */
int gffx_host_line_table_check(const char *gff, uint32_t threads, char *err, size_t errlen);
/* the all-line table `<gff>.lall` written by `gffx index` for intersect's per-line mode (replaces the text
 * walk of write_gff_match_only_by_coords, commands/intersect.rs:266-329): 1 = it loads, equals a fresh build on `threads`
 * host threads, and gives for every block of the index the very lines, columns and -T decisions (types: comma list or NULL)
 * the text walk gives; 0 = not usable (absent / stale; the reason in err); < 0 = error or mismatch.  *n_lines = lines checked. */
int gffx_host_all_lines_check(const char *gff, const char *types, uint32_t threads, uint64_t *n_lines, char *err, size_t errlen);
/* The chromosome-bucket plan of `gffx intersect --gpus N` (LPT with splitting over the per-seqid region counts of a BED
 * chunk; the reference buckets by seqid first, commands/intersect.rs:114-120).  slices = malloc'd (rank, chr, lo, hi) u64
 * quadruples, ranks ascending, a rank's slices sorted by (chr, lo). */
int gffx_host_plan_shards(const uint64_t *bucket_sizes, uint32_t n_chr, uint32_t n_ranks, uint64_t **slices, uint64_t *n_slices);
/* the `gffx` command line in-process (main.rs); returns the exit code */
int gffx_host_cli(int argc, char **argv);
void gffx_host_free(void *);

#ifdef __cplusplus
}
#endif
#endif
