/*
 * gffx_oracle.c -- TEST INFRASTRUCTURE ONLY (see gffx_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's `gffx intersect` path.
 * PARITY UNPINNED (the reference has no tests / golden vectors and cannot be
 * built in this image); pinned only by the hand-derived table in
 * tests/golden/appendix_e.json, by oracle/gffx_oracle_py.py and by brute force.
 *
 * Reference citations are relative to /root/reference/src.
 */
#define _GNU_SOURCE
#include "gffx_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define MISSING UINT64_MAX /* index_loader/gof.rs:7, commands/intersect.rs:18 */

/* ------------------------------------------------------------------ utils */

static void set_err(char *err, size_t errlen, const char *fmt, ...) {
    if (!err || !errlen) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

void oracle_free(void *p) { free(p); }

typedef struct {
    uint8_t *p;
    size_t n, cap;
} buf_t;

static void buf_push(buf_t *b, const void *src, size_t n) {
    if (b->n + n > b->cap) {
        size_t c = b->cap ? b->cap * 2 : 4096;
        while (c < b->n + n) c *= 2;
        b->p = (uint8_t *)realloc(b->p, c);
        b->cap = c;
    }
    memcpy(b->p + b->n, src, n);
    b->n += n;
}

typedef struct {
    const uint8_t *p;
    size_t n;
    int fd;
    int mapped;
} map_t;

static int map_file(const char *path, map_t *m) {
    m->p = NULL;
    m->n = 0;
    m->mapped = 0;
    m->fd = open(path, O_RDONLY);
    if (m->fd < 0) return -1;
    struct stat st;
    if (fstat(m->fd, &st) != 0) {
        close(m->fd);
        return -1;
    }
    m->n = (size_t)st.st_size;
    if (m->n == 0) {
        m->p = (const uint8_t *)"";
        return 0;
    }
    void *p = mmap(NULL, m->n, PROT_READ, MAP_PRIVATE, m->fd, 0);
    if (p == MAP_FAILED) {
        close(m->fd);
        return -1;
    }
    m->p = (const uint8_t *)p;
    m->mapped = 1;
    return 0;
}

static void unmap_file(map_t *m) {
    if (m->mapped) munmap((void *)m->p, m->n);
    if (m->fd >= 0) close(m->fd);
    m->mapped = 0;
    m->fd = -1;
}

/* `<gff filename><suffix>` next to the GFF: utils/common.rs:123-127 append_suffix */
static char *append_suffix(const char *path, const char *suffix) {
    size_t a = strlen(path), b = strlen(suffix);
    char *r = (char *)malloc(a + b + 1);
    memcpy(r, path, a);
    memcpy(r + a, suffix, b + 1);
    return r;
}

/* Rust std::str::from_utf8 validity (index_builder/core.rs:82, intersect.rs:215). */
static int utf8_valid(const uint8_t *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) {
            i++;
        } else if (c >= 0xC2 && c <= 0xDF) {
            if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return 0;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (i + 2 >= n) return 0;
            uint8_t c1 = s[i + 1], c2 = s[i + 2];
            if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80) return 0;
            if (c == 0xE0 && c1 < 0xA0) return 0;
            if (c == 0xED && c1 > 0x9F) return 0;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (i + 3 >= n) return 0;
            uint8_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
            if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return 0;
            if (c == 0xF0 && c1 < 0x90) return 0;
            if (c == 0xF4 && c1 > 0x8F) return 0;
            i += 4;
        } else {
            return 0;
        }
    }
    return 1;
}

/* Byte length of a Unicode White_Space char at s (char::is_whitespace, regex \s), else 0. */
static size_t ws_at(const uint8_t *s, size_t n) {
    if (n == 0) return 0;
    uint8_t c = s[0];
    if ((c >= 0x09 && c <= 0x0D) || c == 0x20) return 1;
    if (c == 0xC2 && n >= 2 && (s[1] == 0x85 || s[1] == 0xA0)) return 2;
    if (n >= 3) {
        if (c == 0xE1 && s[1] == 0x9A && s[2] == 0x80) return 3;                 /* U+1680 */
        if (c == 0xE2 && s[1] == 0x80 && s[2] >= 0x80 && s[2] <= 0x8A) return 3; /* U+2000-200A */
        if (c == 0xE2 && s[1] == 0x80 && (s[2] == 0xA8 || s[2] == 0xA9 || s[2] == 0xAF)) return 3;
        if (c == 0xE2 && s[1] == 0x81 && s[2] == 0x9F) return 3; /* U+205F */
        if (c == 0xE3 && s[1] == 0x80 && s[2] == 0x80) return 3; /* U+3000 */
    }
    return 0;
}

/* str::trim on valid UTF-8 */
static void trim_ws(const uint8_t **ps, size_t *pn) {
    const uint8_t *s = *ps;
    size_t n = *pn;
    for (;;) {
        size_t w = ws_at(s, n);
        if (!w) break;
        s += w;
        n -= w;
    }
    for (;;) {
        if (n == 0) break;
        size_t w = 0;
        if (ws_at(s + n - 1, 1) == 1)
            w = 1;
        else if (n >= 2 && ws_at(s + n - 2, 2) == 2)
            w = 2;
        else if (n >= 3 && ws_at(s + n - 3, 3) == 3)
            w = 3;
        if (!w) break;
        n -= w;
    }
    *ps = s;
    *pn = n;
}

/* Rust `str::parse::<u32>()`: optional '+', >=1 ASCII digits, no overflow. */
static int parse_u32_rust(const uint8_t *s, size_t n, uint32_t *out) {
    if (n == 0) return 0;
    size_t i = 0;
    if (s[0] == '+') {
        i = 1;
        if (n == 1) return 0;
    }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > UINT32_MAX) return 0;
    }
    *out = (uint32_t)v;
    return 1;
}

/* commands/intersect.rs:526-538 parse_u32_ascii: digits only, checked arithmetic. */
static int parse_u32_ascii(const uint8_t *s, size_t n, uint32_t *out) {
    if (n == 0) return 0;
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > UINT32_MAX) return 0;
    }
    *out = (uint32_t)v;
    return 1;
}

/* ------------------------------------------------------- string hash map */

typedef struct {
    char **keys;
    uint32_t *klen;
    uint32_t *vals;
    size_t cap, n;
} smap_t;

static uint64_t fnv1a(const uint8_t *s, size_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= s[i];
        h *= 1099511628211ULL;
    }
    return h;
}

static void smap_init(smap_t *m, size_t cap) {
    size_t c = 16;
    while (c < cap * 2) c *= 2;
    m->cap = c;
    m->n = 0;
    m->keys = (char **)calloc(c, sizeof(char *));
    m->klen = (uint32_t *)calloc(c, sizeof(uint32_t));
    m->vals = (uint32_t *)calloc(c, sizeof(uint32_t));
}

static void smap_free(smap_t *m) {
    for (size_t i = 0; i < m->cap; i++) free(m->keys[i]);
    free(m->keys);
    free(m->klen);
    free(m->vals);
}

static size_t smap_slot(const smap_t *m, const uint8_t *k, size_t n) {
    size_t i = (size_t)fnv1a(k, n) & (m->cap - 1);
    while (m->keys[i] && !(m->klen[i] == n && memcmp(m->keys[i], k, n) == 0))
        i = (i + 1) & (m->cap - 1);
    return i;
}

static void smap_grow(smap_t *m);

/* insert or overwrite; returns 1 if the key was new */
static int smap_put(smap_t *m, const uint8_t *k, size_t n, uint32_t v) {
    if ((m->n + 1) * 2 > m->cap) smap_grow(m);
    size_t i = smap_slot(m, k, n);
    if (m->keys[i]) {
        m->vals[i] = v;
        return 0;
    }
    m->keys[i] = (char *)malloc(n + 1);
    memcpy(m->keys[i], k, n);
    m->keys[i][n] = 0;
    m->klen[i] = (uint32_t)n;
    m->vals[i] = v;
    m->n++;
    return 1;
}

static int smap_get(const smap_t *m, const uint8_t *k, size_t n, uint32_t *v) {
    size_t i = smap_slot(m, k, n);
    if (!m->keys[i]) return 0;
    *v = m->vals[i];
    return 1;
}

static void smap_grow(smap_t *m) {
    smap_t o = *m;
    m->cap = o.cap * 2;
    m->n = 0;
    m->keys = (char **)calloc(m->cap, sizeof(char *));
    m->klen = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
    m->vals = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
    for (size_t i = 0; i < o.cap; i++) {
        if (!o.keys[i]) continue;
        size_t j = (size_t)fnv1a((const uint8_t *)o.keys[i], o.klen[i]) & (m->cap - 1);
        while (m->keys[j]) j = (j + 1) & (m->cap - 1);
        m->keys[j] = o.keys[i];
        m->klen[j] = o.klen[i];
        m->vals[j] = o.vals[i];
        m->n++;
    }
    free(o.keys);
    free(o.klen);
    free(o.vals);
}

/* ------------------------------------------------ interval tree (tree.rs) */

typedef struct { /* utils/tree.rs:5-10 Interval<u32> */
    uint32_t start, end, root_fid;
} ivl_t;

typedef struct node { /* utils/tree.rs:17-23 Node */
    uint32_t center;
    ivl_t *ivs;
    size_t n;
    struct node *left, *right;
} node_t;

/* stable sort by start == Vec::sort_by_key(|iv| iv.start)  (tree.rs:40) */
static void msort_start(ivl_t *a, ivl_t *tmp, size_t n) {
    if (n < 2) return;
    if (n <= 16) {
        for (size_t i = 1; i < n; i++) {
            ivl_t x = a[i];
            size_t j = i;
            while (j > 0 && a[j - 1].start > x.start) {
                a[j] = a[j - 1];
                j--;
            }
            a[j] = x;
        }
        return;
    }
    size_t h = n / 2;
    msort_start(a, tmp, h);
    msort_start(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].start < a[i].start) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(ivl_t));
}

/* utils/tree.rs:35-64 IntervalTree::build; consumes `ivs` (malloc'd) */
static node_t *tree_build(ivl_t *ivs, size_t n) {
    if (n == 0) { /* :36-38 */
        free(ivs);
        return NULL;
    }
    ivl_t *tmp = (ivl_t *)malloc(n * sizeof(ivl_t));
    msort_start(ivs, tmp, n); /* :40 */
    free(tmp);
    size_t mid = n / 2;                /* :41 */
    uint32_t center = ivs[mid].start; /* :42 */
    ivl_t *l = (ivl_t *)malloc(n * sizeof(ivl_t));
    ivl_t *r = (ivl_t *)malloc(n * sizeof(ivl_t));
    ivl_t *c = (ivl_t *)malloc(n * sizeof(ivl_t));
    size_t nl = 0, nr = 0, nc = 0;
    for (size_t i = 0; i < n; i++) { /* :48-56 */
        if (ivs[i].end < center)
            l[nl++] = ivs[i];
        else if (ivs[i].start > center)
            r[nr++] = ivs[i];
        else
            c[nc++] = ivs[i];
    }
    free(ivs);
    node_t *nd = (node_t *)malloc(sizeof(node_t)); /* :58-63 */
    nd->center = center;
    nd->ivs = c;
    nd->n = nc;
    nd->left = tree_build(l, nl);
    nd->right = tree_build(r, nr);
    return nd;
}

static void tree_free(node_t *n) {
    if (!n) return;
    tree_free(n->left);
    tree_free(n->right);
    free(n->ivs);
    free(n);
}

typedef struct {
    const ivl_t **p;
    size_t n, cap;
} hits_t;

static void hits_push(hits_t *h, const ivl_t *iv) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->p = (const ivl_t **)realloc(h->p, h->cap * sizeof(*h->p));
    }
    h->p[h->n++] = iv;
}

/* utils/tree.rs:102-121 query_interval_rec */
static void tree_query(const node_t *n, uint32_t start, uint32_t end, hits_t *out) {
    if (!n) return;
    for (size_t i = 0; i < n->n; i++) { /* :109-113 */
        const ivl_t *iv = &n->ivs[i];
        if (iv->start < end && iv->end > start) hits_push(out, iv);
    }
    if (start < n->center) tree_query(n->left, start, end, out); /* :114-116 */
    if (end > n->center) tree_query(n->right, start, end, out);  /* :117-119 */
}

struct oracle_index {      /* utils/tree_index.rs:12-16 */
    uint32_t n_chr;        /* chr_entries: tree i <-> seqid_num i */
    node_t **trees;
    uint32_t *chr_offsets; /* builder-order copies, for export */
    ivl_t *flat;
    uint32_t n_names;      /* num_to_seqid / seqid_to_num (.sqs lines) */
    char **names;
};

oracle_index *oracle_index_from_roots(uint32_t n_chr, const uint32_t *chr_offsets,
                                      const uint32_t *start, const uint32_t *end,
                                      const uint32_t *fid) {
    oracle_index *ix = (oracle_index *)calloc(1, sizeof(*ix));
    ix->n_chr = n_chr;
    ix->trees = (node_t **)calloc(n_chr ? n_chr : 1, sizeof(node_t *));
    ix->chr_offsets = (uint32_t *)malloc((n_chr + 1) * sizeof(uint32_t));
    memcpy(ix->chr_offsets, chr_offsets, (n_chr + 1) * sizeof(uint32_t));
    uint32_t tot = chr_offsets[n_chr];
    ix->flat = (ivl_t *)malloc((tot ? tot : 1) * sizeof(ivl_t));
    for (uint32_t i = 0; i < tot; i++) {
        ix->flat[i].start = start[i];
        ix->flat[i].end = end[i];
        ix->flat[i].root_fid = fid[i];
    }
    for (uint32_t c = 0; c < n_chr; c++) { /* index_builder/core.rs:206-218 */
        size_t n = chr_offsets[c + 1] - chr_offsets[c];
        ivl_t *cp = (ivl_t *)malloc((n ? n : 1) * sizeof(ivl_t));
        memcpy(cp, ix->flat + chr_offsets[c], n * sizeof(ivl_t));
        ix->trees[c] = tree_build(cp, n);
    }
    ix->n_names = 0;
    ix->names = NULL;
    return ix;
}

void oracle_index_free(oracle_index *ix) {
    if (!ix) return;
    for (uint32_t c = 0; c < ix->n_chr; c++) tree_free(ix->trees[c]);
    free(ix->trees);
    free(ix->chr_offsets);
    free(ix->flat);
    for (uint32_t i = 0; i < ix->n_names; i++) free(ix->names[i]);
    free(ix->names);
    free(ix);
}

uint32_t oracle_index_n_chr(const oracle_index *ix) { return ix->n_chr; }
uint64_t oracle_index_n_roots(const oracle_index *ix) { return ix->chr_offsets[ix->n_chr]; }
const char *oracle_index_seq_name(const oracle_index *ix, uint32_t i) {
    return i < ix->n_names ? ix->names[i] : NULL;
}

int oracle_index_export(const oracle_index *ix, uint32_t **chr_offsets, uint32_t **start,
                        uint32_t **end, uint32_t **fid) {
    uint32_t tot = ix->chr_offsets[ix->n_chr];
    *chr_offsets = (uint32_t *)malloc((ix->n_chr + 1) * sizeof(uint32_t));
    memcpy(*chr_offsets, ix->chr_offsets, (ix->n_chr + 1) * sizeof(uint32_t));
    *start = (uint32_t *)malloc((tot ? tot : 1) * 4);
    *end = (uint32_t *)malloc((tot ? tot : 1) * 4);
    *fid = (uint32_t *)malloc((tot ? tot : 1) * 4);
    for (uint32_t i = 0; i < tot; i++) {
        (*start)[i] = ix->flat[i].start;
        (*end)[i] = ix->flat[i].end;
        (*fid)[i] = ix->flat[i].root_fid;
    }
    return 0;
}

/* --------------------------------------------- Join A (query_features) */

static inline int mode_keep(int mode, uint32_t ivs, uint32_t ive, uint32_t rs, uint32_t re) {
    switch (mode) { /* commands/intersect.rs:145-158 */
    case ORACLE_MODE_CONTAINED: return ivs >= rs && ive <= re;
    case ORACLE_MODE_CONTAINS_REGION: return ivs <= rs && ive >= re;
    default: return 1;
    }
}

/* commands/intersect.rs:105-169 */
int oracle_query_features(const oracle_index *ix, const uint32_t *regions, uint64_t nq, int mode,
                          int invert, uint32_t **triples_out, uint64_t *n_triples,
                          uint32_t *counts) {
    /* :114-120 bucket regions by chromosome (index panics when chr out of range).
     * The bucket vector has seqid_to_num.len() slots; with a consistent index that
     * equals the number of trees. */
    uint32_t nb = ix->n_names ? ix->n_names : ix->n_chr;
    uint64_t *bcnt = (uint64_t *)calloc((size_t)nb + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < nq; i++) {
        uint32_t chr = regions[3 * i];
        if (chr >= nb) {
            free(bcnt);
            return -1;
        }
        bcnt[chr + 1]++;
    }
    for (uint32_t c = 0; c < nb; c++) bcnt[c + 1] += bcnt[c];
    uint64_t *order = (uint64_t *)malloc((nq ? nq : 1) * sizeof(uint64_t));
    {
        uint64_t *cur = (uint64_t *)malloc(((size_t)nb + 1) * sizeof(uint64_t));
        memcpy(cur, bcnt, ((size_t)nb + 1) * sizeof(uint64_t));
        for (uint64_t i = 0; i < nq; i++) order[cur[regions[3 * i]]++] = i;
        free(cur);
    }
    buf_t res = {0};
    hits_t hits = {0};
    if (counts) memset(counts, 0, nq * sizeof(uint32_t));
    /* :124 `for (&seq_num, tree) in &index_data.chr_entries` -- FxHashMap order is
     * unspecified; ascending seq_num is used here. */
    for (uint32_t c = 0; c < ix->n_chr && c < nb; c++) {
        if (bcnt[c + 1] == bcnt[c]) continue; /* :126-128 */
        for (uint64_t k = bcnt[c]; k < bcnt[c + 1]; k++) { /* :139 */
            uint64_t qi = order[k];
            uint32_t rstart = regions[3 * qi + 1], rend = regions[3 * qi + 2];
            hits.n = 0;                                     /* :140 */
            tree_query(ix->trees[c], rstart, rend, &hits); /* :141 */
            for (size_t h = 0; h < hits.n; h++) {          /* :143 */
                const ivl_t *iv = hits.p[h];
                int keep = mode_keep(mode, iv->start, iv->end, rstart, rend);
                if ((invert != 0) ^ keep) { /* :161 */
                    uint32_t t[3] = {iv->root_fid, iv->start, iv->end};
                    buf_push(&res, t, sizeof t); /* :162 */
                    if (counts) counts[qi]++;
                }
            }
        }
    }
    free(hits.p);
    free(order);
    free(bcnt);
    *n_triples = res.n / 12;
    *triples_out = (uint32_t *)res.p;
    if (!res.p) *triples_out = (uint32_t *)malloc(4);
    return 0;
}

int oracle_query_features_brute(uint32_t n_chr, const uint32_t *chr_offsets, const uint32_t *start,
                                const uint32_t *end, const uint32_t *fid, const uint32_t *regions,
                                uint64_t nq, int mode, int invert, uint32_t **triples_out,
                                uint64_t *n_triples, uint32_t *counts) {
    buf_t res = {0};
    if (counts) memset(counts, 0, nq * sizeof(uint32_t));
    for (uint64_t i = 0; i < nq; i++)
        if (regions[3 * i] >= n_chr) return -1;
    for (uint32_t c = 0; c < n_chr; c++) {
        for (uint64_t qi = 0; qi < nq; qi++) {
            if (regions[3 * qi] != c) continue;
            uint32_t rs = regions[3 * qi + 1], re = regions[3 * qi + 2];
            for (uint32_t j = chr_offsets[c]; j < chr_offsets[c + 1]; j++) {
                if (!(start[j] < re && end[j] > rs)) continue; /* tree.rs:110 */
                int keep = mode_keep(mode, start[j], end[j], rs, re);
                if ((invert != 0) ^ keep) {
                    uint32_t t[3] = {fid[j], start[j], end[j]};
                    buf_push(&res, t, sizeof t);
                    if (counts) counts[qi]++;
                }
            }
        }
    }
    *n_triples = res.n / 12;
    *triples_out = res.p ? (uint32_t *)res.p : (uint32_t *)malloc(4);
    return 0;
}

/* ----------------------------------------------- Join B (line predicate) */

/* commands/intersect.rs:500-521 */
int oracle_line_predicate(uint32_t start, uint32_t end, const uint32_t *qs, const uint32_t *qe,
                          uint64_t nq, int mode) {
    for (uint64_t i = 0; i < nq; i++) {
        uint32_t s = qs[i], e = qe[i];
        int keep;
        switch (mode) {
        case ORACLE_MODE_CONTAINED: keep = start >= s && end <= e; break;        /* :504 */
        case ORACLE_MODE_CONTAINS_REGION: keep = start <= s && end >= e; break;  /* :508 */
        default:
            keep = (s <= start && start <= e) || (s <= end && end <= e) ||       /* :512-513 */
                   (start <= s && s <= end) || (start <= e && e <= end);         /* :514-515 */
        }
        if (keep) return 1;
    }
    return 0;
}

static const uint8_t *find_tab(const uint8_t *p, const uint8_t *e) {
    return (const uint8_t *)memchr(p, '\t', (size_t)(e - p));
}

/* commands/intersect.rs:441-523; outputs the parsed pieces for reuse */
static int line_cols(const uint8_t *line, size_t len, const uint8_t **seq, size_t *seqlen,
                     uint32_t *start, uint32_t *end) {
    const uint8_t *e = line + len, *p = line;
    const uint8_t *i1 = find_tab(p, e); /* :449-452 */
    if (!i1) return 0;
    *seq = p;
    *seqlen = (size_t)(i1 - p);
    p = i1 + 1;
    const uint8_t *i2 = find_tab(p, e); /* :457-460 skip source */
    if (!i2) return 0;
    p = i2 + 1;
    const uint8_t *i3 = find_tab(p, e); /* :464-467 skip type */
    if (!i3) return 0;
    p = i3 + 1;
    const uint8_t *i4 = find_tab(p, e); /* :471-478 */
    if (!i4) return 0;
    if (!parse_u32_ascii(p, (size_t)(i4 - p), start)) return 0;
    p = i4 + 1;
    const uint8_t *i5 = find_tab(p, e); /* :482-489 */
    if (!i5) return 0;
    if (!parse_u32_ascii(p, (size_t)(i5 - p), end)) return 0;
    if (!utf8_valid(*seq, *seqlen)) return 0; /* :491-494 */
    return 1;
}

int oracle_gff_line_overlaps_queries(const uint8_t *line, size_t len, uint32_t n_seq,
                                     const char *const *seq_names, const uint64_t *qoff,
                                     const uint32_t *qs, const uint32_t *qe, int mode) {
    const uint8_t *seq;
    size_t seqlen;
    uint32_t start, end;
    if (!line_cols(line, len, &seq, &seqlen, &start, &end)) return 0;
    for (uint32_t i = 0; i < n_seq; i++) { /* :495-498 ivmap.get(seq_str) */
        if (strlen(seq_names[i]) == seqlen && memcmp(seq_names[i], seq, seqlen) == 0) {
            if (qoff[i + 1] == qoff[i]) return 0; /* no entry in the map */
            return oracle_line_predicate(start, end, qs + qoff[i], qe + qoff[i],
                                         qoff[i + 1] - qoff[i], mode);
        }
    }
    return 0;
}

/* commands/intersect.rs:80-102 gff_type_allowed; allow = list of names */
static int gff_type_allowed(const uint8_t *line, size_t len, char **allow, size_t n_allow) {
    const uint8_t *e = line + len, *p = line;
    for (int tabs = 0; tabs < 2; tabs++) { /* :84-92 */
        const uint8_t *t = find_tab(p, e);
        if (!t) return 0;
        p = t + 1;
    }
    const uint8_t *i2 = find_tab(p, e); /* :93-96 */
    if (!i2) return 0;
    size_t n = (size_t)(i2 - p);
    if (!utf8_valid(p, n)) return 0; /* :98-101 */
    for (size_t i = 0; i < n_allow; i++)
        if (strlen(allow[i]) == n && memcmp(allow[i], p, n) == 0) return 1;
    return 0;
}

/* ------------------------------------------ index builder (core.rs:41-242) */

typedef struct { /* index_builder/core.rs:59-67 RawFeature */
    char *seqid;
    uint32_t start, end;
    uint64_t line_offset;
    char *id;
    char *parent; /* NULL = None */
    char *attr;   /* NULL = None */
} rawf_t;

static char *dupn(const uint8_t *s, size_t n) {
    char *r = (char *)malloc(n + 1);
    memcpy(r, s, n);
    r[n] = 0;
    return r;
}

/* regex `<key>=([^;\s]+)` (stop_ws=1) or `<key>=([^;]+)` (stop_ws=0): leftmost match */
static int regex_capture(const uint8_t *line, size_t n, const char *key, int stop_ws,
                         const uint8_t **cap, size_t *caplen) {
    size_t kl = strlen(key);
    for (size_t p = 0; p + kl + 1 <= n; p++) {
        if (memcmp(line + p, key, kl) != 0 || line[p + kl] != '=') continue;
        size_t q = p + kl + 1, q0 = q;
        while (q < n) {
            if (line[q] == ';') break;
            if (stop_ws && ws_at(line + q, n - q)) break;
            q++;
        }
        if (q > q0) {
            *cap = line + q0;
            *caplen = q - q0;
            return 1;
        }
    }
    return 0;
}

static int write_file(const char *path, const void *p, size_t n) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    if (n && fwrite(p, 1, n, f) != n) {
        fclose(f);
        return -1;
    }
    return fclose(f);
}

static void put_u32(buf_t *b, uint32_t v) {
    uint8_t x[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
    buf_push(b, x, 4);
}
static void put_u64(buf_t *b, uint64_t v) {
    uint8_t x[8];
    for (int i = 0; i < 8; i++) x[i] = (uint8_t)(v >> (8 * i));
    buf_push(b, x, 8);
}

/* .rit image of one tree: bincode 1.x default config (hypothesis):
 * Option tag u8, center u32, Vec len u64, Interval = 3 x u32, left, right. */
static void rit_write(buf_t *b, const node_t *n) {
    uint8_t tag = n ? 1 : 0;
    buf_push(b, &tag, 1);
    if (!n) return;
    put_u32(b, n->center);
    put_u64(b, n->n);
    for (size_t i = 0; i < n->n; i++) {
        put_u32(b, n->ivs[i].start);
        put_u32(b, n->ivs[i].end);
        put_u32(b, n->ivs[i].root_fid);
    }
    rit_write(b, n->left);
    rit_write(b, n->right);
}

typedef struct {
    rawf_t *p;
    size_t n, cap;
} rawv_t;

static void rawv_free(rawv_t *v) {
    for (size_t i = 0; i < v->n; i++) {
        free(v->p[i].seqid);
        free(v->p[i].id);
        free(v->p[i].parent);
        free(v->p[i].attr);
    }
    free(v->p);
}

typedef struct { /* everything build_index derives, kept in memory */
    rawv_t raw;
    uint32_t *fid;     /* per raw feature */
    uint32_t *prt;     /* .prt */
    uint32_t *a2f;     /* .a2f */
    char **atn;        /* .atn values */
    size_t n_atn;
    char **seqids;     /* .sqs */
    uint32_t n_seq;
    buf_t gof;         /* .gof bytes */
    /* trees_input per seqid, builder order */
    ivl_t **tin;
    size_t *tin_n;
    size_t *tin_cap;
} built_t;

static void built_free(built_t *b) {
    rawv_free(&b->raw);
    free(b->fid);
    free(b->prt);
    free(b->a2f);
    for (size_t i = 0; i < b->n_atn; i++) free(b->atn[i]);
    free(b->atn);
    for (uint32_t i = 0; i < b->n_seq; i++) {
        free(b->seqids[i]);
        free(b->tin[i]);
    }
    free(b->seqids);
    free(b->tin);
    free(b->tin_n);
    free(b->tin_cap);
    free(b->gof.p);
}

static int build_in_memory(const uint8_t *data, size_t len, const char *attr_key,
                           const char *skip_types, int verbose, built_t *B, char *err,
                           size_t errlen) {
    memset(B, 0, sizeof(*B));
    /* :47 skip_types.split(',') -> set (no trimming, empty strings allowed) */
    char **skip = NULL;
    size_t n_skip = 0;
    {
        const char *s = skip_types;
        for (;;) {
            const char *c = strchr(s, ',');
            size_t n = c ? (size_t)(c - s) : strlen(s);
            skip = (char **)realloc(skip, (n_skip + 1) * sizeof(char *));
            skip[n_skip++] = dupn((const uint8_t *)s, n);
            if (!c) break;
            s = c + 1;
        }
    }
    int rc = -1;
    size_t offset = 0;
    while (offset < len) { /* :71 */
        const uint8_t *nl = (const uint8_t *)memchr(data + offset, '\n', len - offset);
        size_t nl_pos = nl ? (size_t)(nl - data) : len; /* :72-74 */
        const uint8_t *lb = data + offset;
        size_t ln = nl_pos - offset;
        uint64_t line_offset = offset; /* :76 */
        offset = nl_pos + 1;           /* :77 */
        if (ln == 0 || lb[0] == '#') continue; /* :79-81 */
        if (!utf8_valid(lb, ln)) {             /* :82 */
            set_err(err, errlen, "invalid utf-8 sequence in GFF line at byte %llu",
                    (unsigned long long)line_offset);
            goto done;
        }
        const uint8_t *line = lb;
        size_t n = ln;
        trim_ws(&line, &n);
        if (n == 0) continue; /* :83-85 */
        /* :87-90 split('\t') must give exactly 9 fields */
        const uint8_t *f[10];
        size_t fl[10];
        size_t nf = 0;
        {
            const uint8_t *p = line, *e = line + n;
            for (;;) {
                const uint8_t *t = find_tab(p, e);
                if (nf < 10) {
                    f[nf] = p;
                    fl[nf] = (size_t)((t ? t : e) - p);
                }
                nf++;
                if (!t) break;
                p = t + 1;
            }
        }
        if (nf != 9) {
            set_err(err, errlen, "Invalid GFF line (expected 9 columns): %.*s", (int)n, line);
            goto done;
        }
        int skipped = 0; /* :95-100 */
        for (size_t i = 0; i < n_skip; i++)
            if (strlen(skip[i]) == fl[2] && memcmp(skip[i], f[2], fl[2]) == 0) skipped = 1;
        if (skipped) {
            if (verbose) printf("skip comment feature: %.*s\n", (int)fl[2], f[2]);
            continue;
        }
        uint32_t s1, e1;
        if (!parse_u32_rust(f[3], fl[3], &s1) || !parse_u32_rust(f[4], fl[4], &e1)) { /* :102-103 */
            set_err(err, errlen, "invalid digit found in string");
            goto done;
        }
        if (e1 == 0) continue; /* :104-106 */
        if (s1 > e1) {         /* :107 */
            uint32_t t = s1;
            s1 = e1;
            e1 = t;
        }
        uint32_t start = s1 ? s1 - 1 : 0; /* :108 saturating_sub(1) */
        uint32_t end = e1;               /* :109 */
        const uint8_t *cap;
        size_t capn;
        if (!regex_capture(line, n, "ID", 1, &cap, &capn)) { /* :112-115 */
            set_err(err, errlen, "Missing ID in feature: %.*s", (int)n, line);
            goto done;
        }
        rawf_t rf;
        rf.seqid = dupn(f[0], fl[0]);
        rf.start = start;
        rf.end = end;
        rf.line_offset = line_offset;
        rf.id = dupn(cap, capn);
        rf.parent = regex_capture(line, n, "Parent", 1, &cap, &capn) ? dupn(cap, capn) : NULL; /* :117 */
        rf.attr = regex_capture(line, n, attr_key, 0, &cap, &capn) ? dupn(cap, capn) : NULL;   /* :119-127 */
        if (rf.attr && (strchr(rf.attr, ' ') || strchr(rf.attr, ';') || strchr(rf.attr, ',')))
            fprintf(stderr,
                    "[WARN] Attribute value contains invalid chars (.,;) (should be URL-encoded): in '%s'\n",
                    rf.attr);
        if (B->raw.n == B->raw.cap) {
            B->raw.cap = B->raw.cap ? B->raw.cap * 2 : 1024;
            B->raw.p = (rawf_t *)realloc(B->raw.p, B->raw.cap * sizeof(rawf_t));
        }
        B->raw.p[B->raw.n++] = rf;
    }
    {
        size_t nr = B->raw.n;
        smap_t fmap; /* :141-144 feature_map: later duplicates overwrite */
        smap_init(&fmap, nr + 1);
        for (size_t i = 0; i < nr; i++)
            smap_put(&fmap, (const uint8_t *)B->raw.p[i].id, strlen(B->raw.p[i].id), (uint32_t)i);
        B->fid = (uint32_t *)malloc((nr ? nr : 1) * 4);
        B->prt = (uint32_t *)malloc((nr ? nr : 1) * 4);
        B->a2f = (uint32_t *)malloc((nr ? nr : 1) * 4);
        smap_t amap, seqmap;
        smap_init(&amap, 1024);
        smap_init(&seqmap, 64);
        int have_root = 0; /* :156 current_root */
        uint32_t cur_id = 0, cur_seq = 0;
        uint64_t cur_off = 0;
        for (size_t i = 0; i < nr; i++) { /* :159 */
            rawf_t *rf = &B->raw.p[i];
            uint32_t fid = 0;
            smap_get(&fmap, (const uint8_t *)rf->id, strlen(rf->id), &fid); /* :160 */
            B->fid[i] = fid;
            uint32_t parent_id = fid; /* :163-167 */
            if (rf->parent) {
                uint32_t pv;
                if (smap_get(&fmap, (const uint8_t *)rf->parent, strlen(rf->parent), &pv))
                    parent_id = pv;
            }
            B->prt[i] = parent_id; /* :168 */
            if (parent_id == fid) { /* :170 */
                uint32_t seqnum;
                if (!smap_get(&seqmap, (const uint8_t *)rf->seqid, strlen(rf->seqid), &seqnum)) {
                    seqnum = B->n_seq; /* :171-175 */
                    smap_put(&seqmap, (const uint8_t *)rf->seqid, strlen(rf->seqid), seqnum);
                    B->seqids = (char **)realloc(B->seqids, (B->n_seq + 1) * sizeof(char *));
                    B->tin = (ivl_t **)realloc(B->tin, (B->n_seq + 1) * sizeof(ivl_t *));
                    B->tin_n = (size_t *)realloc(B->tin_n, (B->n_seq + 1) * sizeof(size_t));
                    B->tin_cap = (size_t *)realloc(B->tin_cap, (B->n_seq + 1) * sizeof(size_t));
                    B->tin_cap[B->n_seq] = 0;
                    B->seqids[B->n_seq] = strdup(rf->seqid);
                    B->tin[B->n_seq] = NULL;
                    B->tin_n[B->n_seq] = 0;
                    B->n_seq++;
                }
                size_t k = B->tin_n[seqnum]; /* :177-180 */
                if (k == B->tin_cap[seqnum]) {
                    B->tin_cap[seqnum] = k ? k * 2 : 16;
                    B->tin[seqnum] =
                        (ivl_t *)realloc(B->tin[seqnum], B->tin_cap[seqnum] * sizeof(ivl_t));
                }
                B->tin[seqnum][k].start = rf->start;
                B->tin[seqnum][k].end = rf->end;
                B->tin[seqnum][k].root_fid = fid;
                B->tin_n[seqnum] = k + 1;
                if (have_root) { /* :182-184, write_gof :32-38 */
                    put_u32(&B->gof, cur_id);
                    put_u32(&B->gof, cur_seq);
                    put_u64(&B->gof, cur_off);
                    put_u64(&B->gof, rf->line_offset);
                }
                have_root = 1; /* :185 */
                cur_id = fid;
                cur_off = rf->line_offset;
                cur_seq = seqnum;
            }
            if (rf->attr) { /* :189-198 */
                uint32_t aid;
                if (!smap_get(&amap, (const uint8_t *)rf->attr, strlen(rf->attr), &aid)) {
                    aid = (uint32_t)B->n_atn;
                    smap_put(&amap, (const uint8_t *)rf->attr, strlen(rf->attr), aid);
                    B->atn = (char **)realloc(B->atn, (B->n_atn + 1) * sizeof(char *));
                    B->atn[B->n_atn++] = strdup(rf->attr);
                }
                B->a2f[i] = aid;
            } else {
                B->a2f[i] = UINT32_MAX;
            }
        }
        if (have_root) { /* :201-203 */
            put_u32(&B->gof, cur_id);
            put_u32(&B->gof, cur_seq);
            put_u64(&B->gof, cur_off);
            put_u64(&B->gof, (uint64_t)len);
        }
        smap_free(&fmap);
        smap_free(&amap);
        smap_free(&seqmap);
    }
    rc = 0;
done:
    for (size_t i = 0; i < n_skip; i++) free(skip[i]);
    free(skip);
    if (rc != 0) built_free(B);
    return rc;
}

int oracle_build_index(const char *gff_path, const char *attr_key, const char *skip_types,
                       int verbose, char *err, size_t errlen) {
    map_t m;
    if (verbose) fprintf(stderr, "Building index for %s ...\n", gff_path); /* :49-51 */
    if (map_file(gff_path, &m) != 0) {
        set_err(err, errlen, "No such file or directory (os error 2)");
        return -1;
    }
    built_t B;
    if (build_in_memory(m.p, m.n, attr_key, skip_types, verbose, &B, err, errlen) != 0) {
        unmap_file(&m);
        return -1;
    }
    int rc = 0;
    char *path;
    buf_t b = {0};
    /* .fts :161 */
    for (size_t i = 0; i < B.raw.n; i++) {
        buf_push(&b, B.raw.p[i].id, strlen(B.raw.p[i].id));
        buf_push(&b, "\n", 1);
    }
    path = append_suffix(gff_path, ".fts");
    rc |= write_file(path, b.p, b.n);
    free(path);
    /* .gof */
    path = append_suffix(gff_path, ".gof");
    rc |= write_file(path, B.gof.p, B.gof.n);
    free(path);
    /* .rit / .rix :206-224, tree_io.rs:37-63 */
    {
        buf_t rit = {0}, rix = {0};
        buf_push(&rix, "[", 1);
        for (uint32_t c = 0; c < B.n_seq; c++) {
            char num[32];
            int k = snprintf(num, sizeof num, "%s%llu", c ? "," : "", (unsigned long long)rit.n);
            buf_push(&rix, num, (size_t)k);
            ivl_t *cp = (ivl_t *)malloc((B.tin_n[c] ? B.tin_n[c] : 1) * sizeof(ivl_t));
            memcpy(cp, B.tin[c], B.tin_n[c] * sizeof(ivl_t));
            node_t *t = tree_build(cp, B.tin_n[c]);
            rit_write(&rit, t);
            tree_free(t);
        }
        buf_push(&rix, "]", 1);
        path = append_suffix(gff_path, ".rit");
        rc |= write_file(path, rit.p, rit.n);
        free(path);
        path = append_suffix(gff_path, ".rix");
        rc |= write_file(path, rix.p, rix.n);
        free(path);
        free(rit.p);
        free(rix.p);
    }
    /* .sqs :227-228 */
    b.n = 0;
    for (uint32_t c = 0; c < B.n_seq; c++) {
        buf_push(&b, B.seqids[c], strlen(B.seqids[c]));
        buf_push(&b, "\n", 1);
    }
    path = append_suffix(gff_path, ".sqs");
    rc |= write_file(path, b.p, b.n);
    free(path);
    /* .atn :231-234 */
    b.n = 0;
    buf_push(&b, "#attribute=", 11);
    buf_push(&b, attr_key, strlen(attr_key));
    buf_push(&b, "\n", 1);
    for (size_t i = 0; i < B.n_atn; i++) {
        buf_push(&b, B.atn[i], strlen(B.atn[i]));
        buf_push(&b, "\n", 1);
    }
    path = append_suffix(gff_path, ".atn");
    rc |= write_file(path, b.p, b.n);
    free(path);
    /* .a2f / .prt :235-236 */
    b.n = 0;
    for (size_t i = 0; i < B.raw.n; i++) put_u32(&b, B.a2f[i]);
    path = append_suffix(gff_path, ".a2f");
    rc |= write_file(path, b.p, b.n);
    free(path);
    b.n = 0;
    for (size_t i = 0; i < B.raw.n; i++) put_u32(&b, B.prt[i]);
    path = append_suffix(gff_path, ".prt");
    rc |= write_file(path, b.p, b.n);
    free(path);
    free(b.p);
    built_free(&B);
    unmap_file(&m);
    if (rc != 0) set_err(err, errlen, "failed to write index files next to %s", gff_path);
    if (verbose && rc == 0) fprintf(stderr, "Index built successfully for %s\n", gff_path);
    return rc ? -1 : 0;
}

/* ------------------------------------------------------ index loaders */

/* index_loader/core.rs:19-34 load_sqs: BufRead::lines (strips "\n" and "\r\n") */
static int load_sqs(const char *gff_path, char ***names, uint32_t *n, char *err, size_t errlen) {
    char *p = append_suffix(gff_path, ".sqs");
    map_t m;
    if (map_file(p, &m) != 0) {
        set_err(err, errlen, "Failed to open SQS file: \"%s\"", p);
        free(p);
        return -1;
    }
    free(p);
    *names = NULL;
    *n = 0;
    size_t off = 0;
    while (off < m.n) {
        const uint8_t *nl = (const uint8_t *)memchr(m.p + off, '\n', m.n - off);
        size_t e = nl ? (size_t)(nl - m.p) : m.n;
        size_t ln = e - off;
        if (nl && ln > 0 && m.p[e - 1] == '\r') ln--;
        *names = (char **)realloc(*names, (*n + 1) * sizeof(char *));
        (*names)[(*n)++] = dupn(m.p + off, ln);
        off = e + 1;
    }
    unmap_file(&m);
    return 0;
}

typedef struct { /* index_loader/gof.rs:10-15 GofEntry */
    uint32_t fid, seq;
    uint64_t s, e;
} gofe_t;

/* index_loader/gof.rs:95-128 load_gof */
static int load_gof(const char *gff_path, gofe_t **ents, size_t *n, char *err, size_t errlen) {
    char *p = append_suffix(gff_path, ".gof");
    map_t m;
    if (map_file(p, &m) != 0) {
        set_err(err, errlen, "Failed to mmap %s", p);
        free(p);
        return -1;
    }
    if (m.n % 24 != 0) { /* :103-110 */
        set_err(err, errlen, "Corrupted GOF (%s): length %zu not multiple of 24", p, m.n);
        free(p);
        unmap_file(&m);
        return -1;
    }
    free(p);
    *n = m.n / 24;
    *ents = (gofe_t *)malloc((*n ? *n : 1) * sizeof(gofe_t));
    for (size_t i = 0; i < *n; i++) {
        const uint8_t *r = m.p + 24 * i;
        gofe_t g;
        g.fid = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
        g.seq = (uint32_t)r[4] | (uint32_t)r[5] << 8 | (uint32_t)r[6] << 16 | (uint32_t)r[7] << 24;
        g.s = 0;
        g.e = 0;
        for (int k = 0; k < 8; k++) {
            g.s |= (uint64_t)r[8 + k] << (8 * k);
            g.e |= (uint64_t)r[16 + k] << (8 * k);
        }
        (*ents)[i] = g;
    }
    unmap_file(&m);
    return 0;
}

static oracle_index *index_from_tin(uint32_t n_seq, ivl_t **tin, size_t *tin_n, char **names,
                                    uint32_t n_names) {
    uint32_t *off = (uint32_t *)malloc((n_seq + 1) * 4);
    off[0] = 0;
    for (uint32_t c = 0; c < n_seq; c++) off[c + 1] = off[c] + (uint32_t)tin_n[c];
    uint32_t tot = off[n_seq];
    uint32_t *s = (uint32_t *)malloc((tot ? tot : 1) * 4), *e = (uint32_t *)malloc((tot ? tot : 1) * 4),
             *f = (uint32_t *)malloc((tot ? tot : 1) * 4);
    for (uint32_t c = 0; c < n_seq; c++)
        for (size_t k = 0; k < tin_n[c]; k++) {
            s[off[c] + k] = tin[c][k].start;
            e[off[c] + k] = tin[c][k].end;
            f[off[c] + k] = tin[c][k].root_fid;
        }
    oracle_index *ix = oracle_index_from_roots(n_seq, off, s, e, f);
    free(off);
    free(s);
    free(e);
    free(f);
    ix->n_names = n_names;
    ix->names = (char **)malloc((n_names ? n_names : 1) * sizeof(char *));
    for (uint32_t i = 0; i < n_names; i++) ix->names[i] = strdup(names[i]);
    return ix;
}

/* tree inputs == one (start,end,fid) per .gof record, parsed from the
 * root's own line with the builder's coordinate rules (index_builder/core.rs:102-109). */
int oracle_load_tree_index(const char *gff_path, oracle_index **out, char *err, size_t errlen) {
    char **names;
    uint32_t n_names;
    if (load_sqs(gff_path, &names, &n_names, err, errlen) != 0) return -1;
    gofe_t *g;
    size_t ng;
    if (load_gof(gff_path, &g, &ng, err, errlen) != 0) return -1;
    map_t m;
    if (map_file(gff_path, &m) != 0) {
        set_err(err, errlen, "Cannot open GFF: \"%s\"", gff_path);
        return -1;
    }
    uint32_t n_seq = n_names;
    ivl_t **tin = (ivl_t **)calloc(n_seq ? n_seq : 1, sizeof(ivl_t *));
    size_t *tn = (size_t *)calloc(n_seq ? n_seq : 1, sizeof(size_t));
    size_t *tc = (size_t *)calloc(n_seq ? n_seq : 1, sizeof(size_t));
    int rc = 0;
    for (size_t i = 0; i < ng && rc == 0; i++) {
        if (g[i].seq >= n_seq || g[i].s >= m.n) {
            set_err(err, errlen, "GOF record %zu out of range", i);
            rc = -1;
            break;
        }
        const uint8_t *lb = m.p + g[i].s;
        const uint8_t *nl = (const uint8_t *)memchr(lb, '\n', m.n - g[i].s);
        size_t ln = nl ? (size_t)(nl - lb) : m.n - g[i].s;
        trim_ws(&lb, &ln);
        const uint8_t *p = lb, *e = lb + ln;
        const uint8_t *f3 = NULL, *f4 = NULL;
        size_t l3 = 0, l4 = 0;
        for (int col = 0; col < 5; col++) {
            const uint8_t *t = find_tab(p, e);
            if (!t) {
                rc = -1;
                break;
            }
            if (col == 3) {
                f3 = p;
                l3 = (size_t)(t - p);
            }
            if (col == 4) {
                f4 = p;
                l4 = (size_t)(t - p);
            }
            p = t + 1;
        }
        uint32_t s1 = 0, e1 = 0;
        if (rc != 0 || !parse_u32_rust(f3, l3, &s1) || !parse_u32_rust(f4, l4, &e1)) {
            set_err(err, errlen, "cannot parse root line of GOF record %zu", i);
            rc = -1;
            break;
        }
        if (s1 > e1) {
            uint32_t t = s1;
            s1 = e1;
            e1 = t;
        }
        uint32_t c = g[i].seq;
        if (tn[c] == tc[c]) {
            tc[c] = tc[c] ? tc[c] * 2 : 16;
            tin[c] = (ivl_t *)realloc(tin[c], tc[c] * sizeof(ivl_t));
        }
        tin[c][tn[c]].start = s1 ? s1 - 1 : 0;
        tin[c][tn[c]].end = e1;
        tin[c][tn[c]].root_fid = g[i].fid;
        tn[c]++;
    }
    if (rc == 0) *out = index_from_tin(n_seq, tin, tn, names, n_names);
    for (uint32_t c = 0; c < n_seq; c++) free(tin[c]);
    free(tin);
    free(tn);
    free(tc);
    free(g);
    for (uint32_t i = 0; i < n_names; i++) free(names[i]);
    free(names);
    unmap_file(&m);
    return rc;
}

typedef struct {
    const uint8_t *p;
    size_t n, off;
    int bad;
} rd_t;
static uint32_t rd_u32(rd_t *r) {
    if (r->off + 4 > r->n) {
        r->bad = 1;
        return 0;
    }
    const uint8_t *x = r->p + r->off;
    r->off += 4;
    return (uint32_t)x[0] | (uint32_t)x[1] << 8 | (uint32_t)x[2] << 16 | (uint32_t)x[3] << 24;
}
static uint64_t rd_u64(rd_t *r) {
    uint64_t lo = rd_u32(r), hi = rd_u32(r);
    return lo | hi << 32;
}
typedef struct {
    ivl_t *p;
    size_t n, cap;
} ivv_t;
/* collects the intervals of a serialised tree (any order; the tree is rebuilt) */
static void rit_read(rd_t *r, ivv_t *v, int depth) {
    if (r->bad || depth > 200) {
        r->bad = 1;
        return;
    }
    if (r->off >= r->n) {
        r->bad = 1;
        return;
    }
    uint8_t tag = r->p[r->off++];
    if (tag == 0) return;
    if (tag != 1) {
        r->bad = 1;
        return;
    }
    (void)rd_u32(r);
    uint64_t n = rd_u64(r);
    if (r->bad || n > (r->n - r->off) / 12) {
        r->bad = 1;
        return;
    }
    for (uint64_t i = 0; i < n; i++) {
        if (v->n == v->cap) {
            v->cap = v->cap ? v->cap * 2 : 64;
            v->p = (ivl_t *)realloc(v->p, v->cap * sizeof(ivl_t));
        }
        v->p[v->n].start = rd_u32(r);
        v->p[v->n].end = rd_u32(r);
        v->p[v->n].root_fid = rd_u32(r);
        v->n++;
    }
    rit_read(r, v, depth + 1);
    rit_read(r, v, depth + 1);
}

/* utils/tree_index.rs:36-82 load_region_index (bincode layout = hypothesis, unpinned) */
int oracle_load_tree_index_rit(const char *gff_path, oracle_index **out, char *err,
                               size_t errlen) {
    char **names;
    uint32_t n_names;
    if (load_sqs(gff_path, &names, &n_names, err, errlen) != 0) return -1;
    char *rp = append_suffix(gff_path, ".rit"), *xp = append_suffix(gff_path, ".rix");
    map_t rit, rix;
    int rc = -1;
    if (map_file(rp, &rit) != 0) {
        set_err(err, errlen, "open %s", rp);
        goto out0;
    }
    if (map_file(xp, &rix) != 0) {
        set_err(err, errlen, "open %s", xp);
        unmap_file(&rit);
        goto out0;
    }
    {
        /* JSON array of u64 */
        uint64_t *offs = NULL;
        size_t no = 0;
        size_t i = 0;
        while (i < rix.n && rix.p[i] != '[') i++;
        i++;
        while (i < rix.n) {
            while (i < rix.n && (rix.p[i] == ' ' || rix.p[i] == ',' || rix.p[i] == '\n')) i++;
            if (i >= rix.n || rix.p[i] == ']') break;
            uint64_t v = 0;
            size_t d = 0;
            while (i < rix.n && rix.p[i] >= '0' && rix.p[i] <= '9') {
                v = v * 10 + (uint64_t)(rix.p[i] - '0');
                i++;
                d++;
            }
            if (!d) {
                set_err(err, errlen, "parse json %s", xp);
                free(offs);
                goto out1;
            }
            offs = (uint64_t *)realloc(offs, (no + 1) * 8);
            offs[no++] = v;
        }
        for (size_t k = 0; k + 1 < no; k++) /* :54-58 */
            if (offs[k] > offs[k + 1]) {
                set_err(err, errlen, "offsets not sorted ascending: %llu > %llu",
                        (unsigned long long)offs[k], (unsigned long long)offs[k + 1]);
                free(offs);
                goto out1;
            }
        if (no && offs[no - 1] > rit.n) { /* :59-62 */
            set_err(err, errlen, "last offset %llu out of file size %zu",
                    (unsigned long long)offs[no - 1], rit.n);
            free(offs);
            goto out1;
        }
        ivl_t **tin = (ivl_t **)calloc(no ? no : 1, sizeof(ivl_t *));
        size_t *tn = (size_t *)calloc(no ? no : 1, sizeof(size_t));
        int bad = 0;
        for (size_t k = 0; k < no; k++) {
            size_t s = (size_t)offs[k], e = k + 1 < no ? (size_t)offs[k + 1] : rit.n;
            rd_t r = {rit.p + s, e - s, 0, 0};
            ivv_t v = {0};
            rit_read(&r, &v, 0);
            if (r.bad) {
                set_err(err, errlen, "bincode2 deserialize tree #%zu (%zu..%zu)", k, s, e);
                bad = 1;
            }
            tin[k] = v.p;
            tn[k] = v.n;
        }
        if (!bad) {
            *out = index_from_tin((uint32_t)no, tin, tn, names, n_names);
            rc = 0;
        }
        for (size_t k = 0; k < no; k++) free(tin[k]);
        free(tin);
        free(tn);
        free(offs);
    }
out1:
    unmap_file(&rit);
    unmap_file(&rix);
out0:
    free(rp);
    free(xp);
    for (uint32_t i = 0; i < n_names; i++) free(names[i]);
    free(names);
    return rc;
}

/* ------------------------------------------------------- region parsing */

static int seq_lookup(const oracle_index *ix, const uint8_t *s, size_t n, uint32_t *out) {
    for (uint32_t i = 0; i < ix->n_names; i++)
        if (strlen(ix->names[i]) == n && memcmp(ix->names[i], s, n) == 0) {
            /* FxHashMap built by collect(): a later duplicate name overwrites (core.rs:28-32) */
            *out = i;
            for (uint32_t j = i + 1; j < ix->n_names; j++)
                if (strlen(ix->names[j]) == n && memcmp(ix->names[j], s, n) == 0) *out = j;
            return 1;
        }
    return 0;
}

/* commands/intersect.rs:172-198 */
int oracle_parse_region(const char *region, const oracle_index *ix, uint32_t out[3], char *err,
                        size_t errlen) {
    const char *colon = strchr(region, ':'); /* :177-179 split_once(':') */
    if (!colon) {
        set_err(err, errlen, "Invalid region format, expected 'chr:start-end'");
        return -1;
    }
    const char *range = colon + 1;
    const char *dash = strchr(range, '-'); /* :180-182 */
    if (!dash) {
        set_err(err, errlen, "Invalid range format, expected 'start-end'");
        return -1;
    }
    uint32_t s, e;
    if (!parse_u32_rust((const uint8_t *)range, (size_t)(dash - range), &s) || /* :183-184 */
        !parse_u32_rust((const uint8_t *)dash + 1, strlen(dash + 1), &e)) {
        set_err(err, errlen, "invalid digit found in string");
        return -1;
    }
    uint32_t chr;
    if (!seq_lookup(ix, (const uint8_t *)region, (size_t)(colon - region), &chr)) { /* :185-187 */
        set_err(err, errlen, "Sequence ID not found: %.*s", (int)(colon - region), region);
        return -1;
    }
    if (s >= e) { /* :188-190 */
        set_err(err, errlen, "Region start must be less than end (%u >= %u)", s, e);
        return -1;
    }
    out[0] = chr;
    out[1] = s;
    out[2] = e;
    return 0;
}

static int is_ascii_ws(uint8_t c) { /* u8::is_ascii_whitespace: SP \t \n \x0C \r (no \x0B) */
    return c == ' ' || c == '\t' || c == '\n' || c == '\x0C' || c == '\r';
}

/* commands/intersect.rs:201-230.  lexical_core::parse::<u32> (1.0.5, un-vendored): taken
 * as optional '+' then >=1 digits, complete, no overflow -- unpinned, see DESIGN.md. */
int oracle_parse_bed_file(const char *bed_path, const oracle_index *ix, uint32_t **regions_out,
                          uint64_t *nq, char *err, size_t errlen) {
    map_t m;
    if (map_file(bed_path, &m) != 0) {
        set_err(err, errlen, "No such file or directory (os error 2)");
        return -1;
    }
    /* name -> number through a hash map so big BEDs stay fast */
    smap_t sm;
    smap_init(&sm, ix->n_names + 1);
    for (uint32_t i = 0; i < ix->n_names; i++)
        smap_put(&sm, (const uint8_t *)ix->names[i], strlen(ix->names[i]), i);
    buf_t out = {0};
    int rc = 0;
    size_t off = 0;
    /* :211 mmap.split(|b| b == '\n'): a final empty piece after a trailing '\n' is skipped by :212 */
    while (off <= m.n) {
        const uint8_t *nl = off < m.n ? (const uint8_t *)memchr(m.p + off, '\n', m.n - off) : NULL;
        size_t e = nl ? (size_t)(nl - m.p) : m.n;
        const uint8_t *line = m.p + off;
        size_t ln = e - off;
        off = e + 1;
        if (ln == 0 || line[0] == '#') continue; /* :212-214 */
        if (!utf8_valid(line, ln)) {             /* :215 */
            set_err(err, errlen, "invalid utf-8 sequence in BED line");
            rc = -1;
            break;
        }
        const uint8_t *f[3];
        size_t fl[3];
        int nf = 0;
        size_t i = 0;
        while (i < ln && nf < 3) { /* :216-219 split_ascii_whitespace */
            while (i < ln && is_ascii_ws(line[i])) i++;
            if (i >= ln) break;
            size_t j = i;
            while (j < ln && !is_ascii_ws(line[j])) j++;
            f[nf] = line + i;
            fl[nf] = j - i;
            nf++;
            i = j;
        }
        if (nf < 3) continue;
        uint32_t chr;
        if (!smap_get(&sm, f[0], fl[0], &chr)) continue; /* :220-222 */
        uint32_t t[3];
        t[0] = chr;
        if (!parse_u32_rust(f[1], fl[1], &t[1]) || !parse_u32_rust(f[2], fl[2], &t[2])) { /* :223-224 */
            set_err(err, errlen, "lexical parse error: invalid BED coordinate");
            rc = -1;
            break;
        }
        buf_push(&out, t, sizeof t); /* :225 */
    }
    smap_free(&sm);
    unmap_file(&m);
    if (rc != 0) {
        free(out.p);
        return -1;
    }
    *nq = out.n / 12;
    *regions_out = out.p ? (uint32_t *)out.p : (uint32_t *)malloc(4);
    return 0;
}

/* ------------------------------------------------------------ writers */

typedef struct {
    uint32_t fid;
    uint64_t s, e;
} block_t;

static int cmp_block_start(const void *a, const void *b) {
    uint64_t x = ((const block_t *)a)->s, y = ((const block_t *)b)->s;
    return x < y ? -1 : x > y;
}

/* utils/common.rs:188-287 write_gff_output */
static void write_gff_output(const uint8_t *gff, size_t file_len, const block_t *blocks,
                             size_t nb, buf_t *out) {
    block_t *v = (block_t *)malloc((nb ? nb : 1) * sizeof(block_t));
    size_t n = 0;
    for (size_t i = 0; i < nb; i++) { /* :200-208 */
        if (blocks[i].s == MISSING) {
            fprintf(stderr, "[WARN] skipped fid=%u due to sentinel start offset\n", blocks[i].fid);
            continue;
        }
        v[n++] = blocks[i];
    }
    qsort(v, n, sizeof(block_t), cmp_block_start); /* :210 */
    block_t *mg = (block_t *)malloc((n ? n : 1) * sizeof(block_t));
    size_t nm = 0;
    if (n) { /* :212-229 */
        uint64_t cs = v[0].s, ce = v[0].e;
        for (size_t i = 1; i < n; i++) {
            if (v[i].s <= ce) {
                if (v[i].e > ce) ce = v[i].e;
            } else {
                if (cs < ce) {
                    mg[nm].s = cs;
                    mg[nm].e = ce;
                    nm++;
                }
                cs = v[i].s;
                ce = v[i].e;
            }
        }
        if (cs < ce) {
            mg[nm].s = cs;
            mg[nm].e = ce;
            nm++;
        }
    }
    for (size_t i = 0; i < nm; i++) { /* :232-242 */
        if (mg[i].s >= mg[i].e) continue;
        if (mg[i].e > file_len) continue;
        buf_push(out, gff + mg[i].s, (size_t)(mg[i].e - mg[i].s));
    }
    free(v);
    free(mg);
}

/* commands/intersect.rs:232-438 write_gff_match_only_by_coords */
static void write_gff_match_only_by_coords(const uint8_t *gff, size_t file_len,
                                           const block_t *blocks, size_t nb, uint32_t n_seq,
                                           const char *const *seq_names, const uint64_t *qoff,
                                           const uint32_t *qs, const uint32_t *qe,
                                           const char *types_filter, int mode, buf_t *out) {
    char **allow = NULL; /* :252-259 */
    size_t n_allow = 0;
    if (types_filter) {
        const char *s = types_filter;
        for (;;) {
            const char *c = strchr(s, ',');
            const uint8_t *t = (const uint8_t *)s;
            size_t n = c ? (size_t)(c - s) : strlen(s);
            trim_ws(&t, &n);
            if (n) {
                allow = (char **)realloc(allow, (n_allow + 1) * sizeof(char *));
                allow[n_allow++] = dupn(t, n);
            }
            if (!c) break;
            s = c + 1;
        }
    }
    block_t *v = (block_t *)malloc((nb ? nb : 1) * sizeof(block_t));
    memcpy(v, blocks, nb * sizeof(block_t));
    qsort(v, nb, sizeof(block_t), cmp_block_start); /* :335 (order of kept parts) */
    for (size_t b = 0; b < nb; b++) {               /* :266-329 */
        if (v[b].s == MISSING) {
            fprintf(stderr, "[WARN] skipped fid=%u due to sentinel start offset\n", v[b].fid);
            continue;
        }
        size_t s = (size_t)v[b].s;
        size_t e = v[b].e < (uint64_t)file_len ? (size_t)v[b].e : file_len; /* :274 */
        if (s >= e || e > file_len) continue;                               /* :275-277 */
        const uint8_t *src = gff + s;
        size_t slen = e - s, pos = 0;
        while (pos < slen) { /* :284 */
            const uint8_t *nlp = (const uint8_t *)memchr(src + pos, '\n', slen - pos);
            size_t nl = nlp ? (size_t)(nlp - src) + 1 : slen; /* :286-289 */
            const uint8_t *line = src + pos;
            size_t ln = nl - pos;
            if (ln && line[ln - 1] == '\n') ln--; /* :293-297 */
            if (ln && line[0] != '#') {           /* :299 */
                int pass = 1;
                if (types_filter && !gff_type_allowed(line, ln, allow, n_allow)) pass = 0; /* :302-306 */
                if (pass && oracle_gff_line_overlaps_queries(line, ln, n_seq, seq_names, qoff, qs,
                                                             qe, mode))
                    buf_push(out, src + pos, nl - pos); /* :309-312 incl. '\n' */
            }
            pos = nl; /* :320 */
        }
    }
    free(v);
    for (size_t i = 0; i < n_allow; i++) free(allow[i]);
    free(allow);
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* commands/intersect.rs:541-655 run */
int oracle_intersect_run(const char *gff_path, const char *region, const char *bed_path, int mode,
                         int invert, int entire_group, const char *types, const char *out_path,
                         char *err, size_t errlen) {
    oracle_index *ix = NULL;
    if (oracle_load_tree_index(gff_path, &ix, err, errlen) != 0) return 1; /* :561 */
    uint32_t *regions = NULL;
    uint64_t nq = 0;
    if (bed_path) { /* :564-572 */
        if (oracle_parse_bed_file(bed_path, ix, &regions, &nq, err, errlen) != 0) {
            oracle_index_free(ix);
            return 1;
        }
    } else if (region) {
        regions = (uint32_t *)malloc(12);
        nq = 1;
        if (oracle_parse_region(region, ix, regions, err, errlen) != 0) {
            free(regions);
            oracle_index_free(ix);
            return 1;
        }
    } else {
        set_err(err, errlen, "No region specified");
        oracle_index_free(ix);
        return 1;
    }
    uint32_t *feats = NULL;
    uint64_t nf = 0;
    if (oracle_query_features(ix, regions, nq, mode, invert, &feats, &nf, NULL) != 0) { /* :586-594 */
        set_err(err, errlen, "panic: index out of bounds (chr)");
        free(regions);
        oracle_index_free(ix);
        return 101;
    }
    gofe_t *g = NULL; /* :597 */
    size_t ng = 0;
    if (load_gof(gff_path, &g, &ng, err, errlen) != 0) {
        free(feats);
        free(regions);
        oracle_index_free(ix);
        return 1;
    }
    /* :598-615 unique roots (hash order irrelevant: both writers sort by offset) */
    uint32_t *roots = (uint32_t *)malloc((nf ? nf : 1) * 4);
    for (uint64_t i = 0; i < nf; i++) roots[i] = feats[3 * i];
    qsort(roots, nf, 4, cmp_u32);
    size_t nr = 0;
    for (uint64_t i = 0; i < nf; i++)
        if (i == 0 || roots[i] != roots[i - 1]) roots[nr++] = roots[i];
    /* :617 roots_to_offsets (gof.rs:54-84): fid -> offsets, later GOF duplicates win (gof.rs:32-37) */
    block_t *blocks = (block_t *)malloc((nr ? nr : 1) * sizeof(block_t));
    for (size_t i = 0; i < nr; i++) {
        blocks[i].fid = roots[i];
        blocks[i].s = MISSING;
        blocks[i].e = MISSING;
        for (size_t k = 0; k < ng; k++)
            if (g[k].fid == roots[i]) {
                blocks[i].s = g[k].s;
                blocks[i].e = g[k].e;
            }
    }
    map_t m;
    int rc = 0;
    if (map_file(gff_path, &m) != 0) {
        set_err(err, errlen, "Cannot open GFF: \"%s\"", gff_path);
        rc = 1;
    } else {
        buf_t out = {0};
        if (!entire_group || types) { /* :619 */
            /* :621-633 query_ivmap: seq name -> all regions of that seq, BED order */
            uint32_t ns = ix->n_names;
            uint64_t *qoff = (uint64_t *)calloc((size_t)ns + 1, 8);
            for (uint64_t i = 0; i < nq; i++) qoff[regions[3 * i] + 1]++;
            for (uint32_t c = 0; c < ns; c++) qoff[c + 1] += qoff[c];
            uint32_t *qs = (uint32_t *)malloc((nq ? nq : 1) * 4), *qe = (uint32_t *)malloc((nq ? nq : 1) * 4);
            uint64_t *cur = (uint64_t *)malloc(((size_t)ns + 1) * 8);
            memcpy(cur, qoff, ((size_t)ns + 1) * 8);
            for (uint64_t i = 0; i < nq; i++) {
                uint64_t k = cur[regions[3 * i]]++;
                qs[k] = regions[3 * i + 1];
                qe[k] = regions[3 * i + 2];
            }
            free(cur);
            write_gff_match_only_by_coords(m.p, m.n, blocks, nr, ns, (const char *const *)ix->names,
                                           qoff, qs, qe, types, mode, &out); /* :636-644 */
            free(qoff);
            free(qs);
            free(qe);
        } else {
            write_gff_output(m.p, m.n, blocks, nr, &out); /* :647-652 */
        }
        if (out_path) {
            if (write_file(out_path, out.p, out.n) != 0) {
                set_err(err, errlen, "cannot write %s", out_path);
                rc = 1;
            }
        } else {
            fwrite(out.p, 1, out.n, stdout);
            fflush(stdout);
        }
        free(out.p);
        unmap_file(&m);
    }
    free(blocks);
    free(roots);
    free(g);
    free(feats);
    free(regions);
    oracle_index_free(ix);
    return rc;
}

/* ------------------------------------------- depth (commands/depth.rs, BED source) */

/* depth.rs:86-97 parse_u32_fast: non-empty, ASCII digits only, checked arithmetic
 * (identical to intersect.rs's parse_u32_ascii) */
#define parse_u32_fast parse_u32_ascii

/* depth.rs:105-120 fast_id: the FIRST occurrence of the bytes "ID=" anywhere in the attribute
 * column (so "geneID=" matches too), value up to the next ';', ' ' or '\t' (a trailing '\r' of a
 * CRLF file stays part of the value).  Returns 0 if there is none. */
static int fast_id(const uint8_t *a, size_t n, const uint8_t **id, size_t *idlen) {
    size_t i = 0;
    while (i + 2 < n) {
        if (a[i] == 'I' && a[i + 1] == 'D' && a[i + 2] == '=') {
            size_t j = i + 3;
            while (j < n && a[j] != ';' && a[j] != ' ' && a[j] != '\t') j++;
            *id = a + i + 3;
            *idlen = j - (i + 3);
            return 1;
        }
        i++;
    }
    return 0;
}

typedef struct {
    uint32_t start, end, id; /* depth.rs:102-103 FeatureInst (0-based half-open), id = GLOBAL id index */
    const uint8_t *seq;      /* the line's seqid column (for id_chrom, :151) */
    size_t seqlen;
} dfeat_t;

typedef struct {
    smap_t ids;          /* id string -> index */
    char **id_str;       /* index -> id (owned by ids) */
    char **chrom;        /* index -> seqid of the first root block (file order) that gave the ID a depth */
    uint32_t *min_s, *max_e, *stamp;
    uint64_t *depth;
    size_t n, cap;
    uint32_t serial;     /* root block being processed */
} dres_t;

static uint32_t dres_id(dres_t *r, const uint8_t *id, size_t idlen) {
    uint32_t v;
    if (smap_get(&r->ids, id, idlen, &v)) return v;
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 1024;
        r->id_str = (char **)realloc(r->id_str, r->cap * sizeof(char *));
        r->chrom = (char **)realloc(r->chrom, r->cap * sizeof(char *));
        r->min_s = (uint32_t *)realloc(r->min_s, r->cap * 4);
        r->max_e = (uint32_t *)realloc(r->max_e, r->cap * 4);
        r->stamp = (uint32_t *)realloc(r->stamp, r->cap * 4);
        r->depth = (uint64_t *)realloc(r->depth, r->cap * 8);
    }
    v = (uint32_t)r->n++;
    smap_put(&r->ids, id, idlen, v);
    r->id_str[v] = dupn(id, idlen);
    r->chrom[v] = NULL;
    r->stamp[v] = 0;
    r->min_s[v] = UINT32_MAX;
    r->max_e[v] = 0;
    r->depth[v] = 0;
    return v;
}

/* depth.rs:121-217 compute_root_depth on one root's byte block and the regions that hit the root,
 * merged straight into the global result (the merges of :264-291 and :501-508 are min / max / sum,
 * hence order-independent; `chrom` comes from the first root block that contributes the ID -- its
 * id_chrom (:151) is the seqid of the block's first line with that ID.  "First" is hash order in the
 * reference and file order here; it only matters if one ID sits on two seqids).
 * The reference gathers candidates through 2^bin_shift bins first (:163-197); a feature and a region
 * that overlap share the bin of any common position, and bins only ever ADD candidates that the
 * exact test (:196) then rejects, so the binning cannot change the result and is not restated. */
static void depth_one_root(const uint8_t *slice, size_t n, const uint32_t *regs /* (s,e) pairs */, size_t nregs,
                           dres_t *res) {
    if (!utf8_valid(slice, n) || nregs == 0) return; /* :132 `if let Ok(text)`, :153 */
    dfeat_t *f = NULL;
    size_t nf = 0, cf = 0;
    size_t pos = 0;
    while (pos < n) { /* :133 split_terminator('\n') */
        const uint8_t *lb = slice + pos;
        const uint8_t *nl = (const uint8_t *)memchr(lb, '\n', n - pos);
        size_t ll = nl ? (size_t)(nl - lb) : n - pos;
        pos += ll + 1;
        if (ll == 0 || lb[0] == '#') continue; /* :134 */
        /* :135-141 splitn(9, '\t'): 8 tabs needed; column 9 is the remainder */
        const uint8_t *col[9];
        size_t len[9];
        const uint8_t *p = lb, *e = lb + ll;
        int k = 0;
        for (; k < 8; k++) {
            const uint8_t *t = find_tab(p, e);
            if (!t) break;
            col[k] = p;
            len[k] = (size_t)(t - p);
            p = t + 1;
        }
        if (k < 8) continue;
        col[8] = p;
        len[8] = (size_t)(e - p);
        uint32_t s1, e1;
        if (!parse_u32_fast(col[3], len[3], &s1) || !parse_u32_fast(col[4], len[4], &e1)) continue; /* :143 */
        if (e1 == 0) continue;                                                                      /* :144 */
        if (s1 > e1) { uint32_t t = s1; s1 = e1; e1 = t; }                                          /* :145 */
        const uint8_t *id;
        size_t idlen;
        if (!fast_id(col[8], len[8], &id, &idlen)) continue; /* :149 */
        if (nf == cf) {
            cf = cf ? cf * 2 : 64;
            f = (dfeat_t *)realloc(f, cf * sizeof(dfeat_t));
        }
        f[nf].start = s1 ? s1 - 1 : 0; /* :146 saturating_sub(1) */
        f[nf].end = e1;               /* :147 */
        f[nf].id = dres_id(res, id, idlen);
        f[nf].seq = col[0];
        f[nf].seqlen = len[0];
        nf++;
    }
    /* :181-208 per region: the set of IDs with an overlapping instance; min/max over those instances */
    uint32_t *hit = (uint32_t *)malloc((nf ? nf : 1) * 4);
    for (size_t r = 0; r < nregs; r++) {
        const uint32_t rs = regs[2 * r], re = regs[2 * r + 1];
        size_t nh = 0;
        for (size_t i = 0; i < nf; i++) {
            const uint32_t l = f[i].start > rs ? f[i].start : rs, rr = f[i].end < re ? f[i].end : re;
            if (l < rr) { /* :78-82 overlaps */
                hit[nh++] = f[i].id;
                if (f[i].start < res->min_s[f[i].id]) res->min_s[f[i].id] = f[i].start;
                if (f[i].end > res->max_e[f[i].id]) res->max_e[f[i].id] = f[i].end;
            }
        }
        qsort(hit, nh, 4, cmp_u32); /* :202-203 sort + dedup */
        for (size_t i = 0; i < nh; i++)
            if (i == 0 || hit[i] != hit[i - 1]) { /* :204-206 */
                res->depth[hit[i]]++;
                res->stamp[hit[i]] = res->serial;
            }
    }
    for (size_t i = 0; i < nf; i++) /* :210-216 + the or_insert of the merges: chrom of a new ID */
        if (res->stamp[f[i].id] == res->serial && !res->chrom[f[i].id]) res->chrom[f[i].id] = dupn(f[i].seq, f[i].seqlen);
    free(hit);
    free(f);
}

/* depth.rs:429-495: BED rows of `depth`.  Lines are cut at '\n' and KEEP it; fields split on tab or
 * space, empty fields dropped; < 3 fields, '#' lines, a non-UTF-8 or unparsable field, s >= e or an
 * unknown seqid drop the row silently.  Column 3 is trim_end()ed before parsing (:485). */
int oracle_depth_parse_bed(const char *bed_path, const oracle_index *ix, uint32_t **regions_out, uint64_t *nq,
                           char *err, size_t errlen) {
    map_t m;
    if (map_file(bed_path, &m) != 0) {
        set_err(err, errlen, "No such file or directory (os error 2)");
        return -1;
    }
    buf_t out = {0};
    uint64_t n = 0;
    size_t pos = 0;
    while (pos < m.n) {
        const uint8_t *lb = m.p + pos;
        const uint8_t *nl = (const uint8_t *)memchr(lb, '\n', m.n - pos);
        size_t ll = nl ? (size_t)(nl - lb) + 1 : m.n - pos; /* the window includes the '\n' */
        pos += ll;
        if (ll == 0 || lb[0] == '#') continue;
        const uint8_t *fld[3];
        size_t fl[3];
        int nfld = 0;
        size_t i = 0;
        while (i < ll) {
            while (i < ll && (lb[i] == '\t' || lb[i] == ' ')) i++;
            size_t j = i;
            while (j < ll && lb[j] != '\t' && lb[j] != ' ') j++;
            if (j > i) {
                if (nfld < 3) {
                    fld[nfld] = lb + i;
                    fl[nfld] = j - i;
                }
                nfld++;
            }
            i = j;
        }
        if (nfld < 3) continue;
        if (!utf8_valid(fld[0], fl[0]) || !utf8_valid(fld[1], fl[1]) || !utf8_valid(fld[2], fl[2])) continue;
        uint32_t s, e, chr;
        if (!parse_u32_rust(fld[1], fl[1], &s)) continue;
        const uint8_t *f2 = fld[2];
        size_t l2 = fl[2];
        for (;;) { /* str::trim_end(): trailing Unicode whitespace */
            size_t k = 0, w = 0;
            /* find a whitespace char that ends exactly at l2 (ASCII fast path, then multi-byte) */
            if (l2 && (f2[l2 - 1] == ' ' || (f2[l2 - 1] >= 0x09 && f2[l2 - 1] <= 0x0D))) {
                l2--;
                continue;
            }
            for (k = 2; k <= 3 && k <= l2; k++)
                if ((w = ws_at(f2 + l2 - k, k)) == k) break;
            if (k <= 3 && k <= l2 && w == k) {
                l2 -= k;
                continue;
            }
            break;
        }
        if (!parse_u32_rust(f2, l2, &e)) continue;
        if (s >= e) continue;
        if (!seq_lookup(ix, fld[0], fl[0], &chr)) continue;
        uint32_t t[3] = {chr, s, e};
        buf_push(&out, t, 12);
        n++;
    }
    unmap_file(&m);
    *regions_out = (uint32_t *)out.p;
    if (!out.p) *regions_out = (uint32_t *)malloc(4);
    *nq = n;
    return 0;
}

static int cmp_cstr_idx(const void *a, const void *b, void *ctx) {
    char **s = (char **)ctx;
    return strcmp(s[*(const uint32_t *)a], s[*(const uint32_t *)b]);
}

/* depth.rs:548-635 run with a .bed source: "id\tchr\tstart\tend\tdepth" rows for every ID with depth > 0.
 * The reference writes them in FxHashMap order (unspecified); here they are sorted by id bytes so that two
 * outputs can be compared.  Batching by 100 000 BED lines (:462) only bounds memory: every merge is min /
 * max / sum, so one pass over all regions gives the same rows.  Returns 0, or 1 with the message in err. */
int oracle_depth_run(const char *gff_path, const char *bed_path, const char *out_path, char *err, size_t errlen) {
    gofe_t *g = NULL;
    size_t ng = 0;
    if (load_gof(gff_path, &g, &ng, err, errlen) != 0) return 1; /* :563 */
    map_t m;
    if (map_file(gff_path, &m) != 0) { /* :564-565 */
        set_err(err, errlen, "Cannot open GFF file: \"%s\"", gff_path);
        free(g);
        return 1;
    }
    oracle_index *ix = NULL;
    if (oracle_load_tree_index(gff_path, &ix, err, errlen) != 0) { /* :573 */
        unmap_file(&m);
        free(g);
        return 1;
    }
    uint32_t *regions = NULL;
    uint64_t nq = 0;
    if (oracle_depth_parse_bed(bed_path, ix, &regions, &nq, err, errlen) != 0) {
        oracle_index_free(ix);
        unmap_file(&m);
        free(g);
        return 1;
    }
    /* gof.index_cached(): fid -> (start, end), later duplicates win (gof.rs:32-37) */
    uint32_t max_fid = 0;
    for (size_t k = 0; k < ng; k++)
        if (g[k].fid > max_fid) max_fid = g[k].fid;
    uint32_t *rec_of = (uint32_t *)malloc(((size_t)max_fid + 2) * 4);
    memset(rec_of, 0xFF, ((size_t)max_fid + 2) * 4);
    for (size_t k = 0; k < ng; k++) rec_of[g[k].fid] = (uint32_t)k;
    /* depth.rs:222-249 compute_hit_depth, first half: regions per root (deduped per region by root_fid) */
    buf_t *by_root = (buf_t *)calloc(ng ? ng : 1, sizeof(buf_t));
    hits_t hits = {0};
    uint32_t seen[64];
    for (uint64_t i = 0; i < nq; i++) {
        const uint32_t chr = regions[3 * i], rs = regions[3 * i + 1], re = regions[3 * i + 2];
        if (chr >= ix->n_chr) continue; /* chr_entries.get(&chr) */
        hits.n = 0;
        tree_query(ix->trees[chr], rs, re, &hits);
        size_t ns = 0;
        uint32_t *big = NULL;
        for (size_t h = 0; h < hits.n; h++) {
            const uint32_t fid = hits.p[h]->root_fid;
            int dup = 0;
            const uint32_t *sv = big ? big : seen;
            for (size_t x = 0; x < ns; x++)
                if (sv[x] == fid) dup = 1;
            if (dup) continue; /* :241 seen_in_region */
            if (!big && ns == 64) {
                big = (uint32_t *)malloc((hits.n + 1) * 4);
                memcpy(big, seen, 64 * 4);
            }
            (big ? big : seen)[ns++] = fid;
            if (fid > max_fid || rec_of[fid] == UINT32_MAX) continue; /* :242 idx.get */
            const gofe_t *ge = &g[rec_of[fid]];
            if (ge->s == MISSING || ge->e == MISSING || ge->e <= ge->s) continue; /* :243 */
            uint32_t se[2] = {rs, re};
            buf_push(&by_root[rec_of[fid]], se, 8); /* :244 */
        }
        free(big);
    }
    free(hits.p);
    dres_t res;
    memset(&res, 0, sizeof res);
    smap_init(&res.ids, 1024);
    for (size_t k = 0; k < ng; k++) { /* :251-292, roots in file order instead of hash order */
        if (!by_root[k].n) continue;
        res.serial++;
        if (g[k].e <= m.n) depth_one_root(m.p + g[k].s, (size_t)(g[k].e - g[k].s), (const uint32_t *)by_root[k].p, by_root[k].n / 8, &res);
        free(by_root[k].p);
    }
    free(by_root);
    /* :515-546 write_depth_results */
    uint32_t *order = (uint32_t *)malloc((res.n ? res.n : 1) * 4);
    size_t no = 0;
    for (size_t i = 0; i < res.n; i++)
        if (res.depth[i] > 0) order[no++] = (uint32_t)i;
    qsort_r(order, no, 4, cmp_cstr_idx, res.id_str);
    buf_t out = {0};
    const char *hdr = "id\tchr\tstart\tend\tdepth\n";
    buf_push(&out, hdr, strlen(hdr));
    for (size_t x = 0; x < no; x++) {
        const uint32_t i = order[x];
        char num[96];
        buf_push(&out, res.id_str[i], strlen(res.id_str[i]));
        buf_push(&out, "\t", 1);
        buf_push(&out, res.chrom[i], strlen(res.chrom[i]));
        int nn = snprintf(num, sizeof num, "\t%u\t%u\t%llu\n", res.min_s[i] == UINT32_MAX ? 0u : res.min_s[i], res.max_e[i],
                          (unsigned long long)res.depth[i]);
        buf_push(&out, num, (size_t)nn);
    }
    int rc = 0;
    if (out_path) {
        if (write_file(out_path, out.p, out.n) != 0) {
            set_err(err, errlen, "cannot write %s", out_path);
            rc = 1;
        }
    } else {
        fwrite(out.p, 1, out.n, stdout);
        fflush(stdout);
    }
    free(out.p);
    free(order);
    for (size_t i = 0; i < res.n; i++) {
        free(res.id_str[i]);
        free(res.chrom[i]);
    }
    free(res.id_str);
    free(res.chrom);
    free(res.min_s);
    free(res.max_e);
    free(res.stamp);
    free(res.depth);
    smap_free(&res.ids);
    free(rec_of);
    free(regions);
    oracle_index_free(ix);
    unmap_file(&m);
    free(g);
    return rc;
}

/* ---------------------------------------- coverage (commands/coverage.rs, BED source) */

typedef struct {
    uint32_t s, e;
} cspan_t;

static int cmp_cspan(const void *a, const void *b) {
    const cspan_t *x = (const cspan_t *)a, *y = (const cspan_t *)b;
    return x->s < y->s ? -1 : x->s > y->s;
}

/* coverage.rs:92-109 merge_intervals: sort by start, merge while s <= current end (touching spans merge) */
static size_t merge_spans(cspan_t *v, size_t n) {
    if (!n) return 0;
    qsort(v, n, sizeof *v, cmp_cspan);
    size_t o = 0;
    uint32_t cs = v[0].s, ce = v[0].e;
    for (size_t i = 1; i < n; i++) {
        if (v[i].s <= ce) {
            if (v[i].e > ce) ce = v[i].e;
        } else {
            v[o].s = cs;
            v[o].e = ce;
            o++;
            cs = v[i].s;
            ce = v[i].e;
        }
    }
    v[o].s = cs;
    v[o].e = ce;
    return o + 1;
}

/* coverage.rs:112-124 union_len */
static uint64_t union_len(cspan_t *v, size_t n) {
    if (!n) return 0;
    qsort(v, n, sizeof *v, cmp_cspan);
    uint64_t total = 0;
    uint32_t cs = v[0].s, ce = v[0].e;
    for (size_t i = 1; i < n; i++) {
        if (v[i].s <= ce) {
            if (v[i].e > ce) ce = v[i].e;
        } else {
            total += ce - cs;
            cs = v[i].s;
            ce = v[i].e;
        }
    }
    return total + (ce - cs);
}

typedef struct {
    smap_t ids;
    char **id_str, **chrom;
    uint32_t *min_s, *max_e, *stamp;
    uint64_t *breadth;
    uint8_t *present;  /* the ID has a row (coverage.rs:372: length > 0 || breadth > 0) */
    size_t n, cap;
    uint32_t serial;
} cres_t;

static uint32_t cres_id(cres_t *r, const uint8_t *id, size_t idlen) {
    uint32_t v;
    if (smap_get(&r->ids, id, idlen, &v)) return v;
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 1024;
        r->id_str = (char **)realloc(r->id_str, r->cap * sizeof(char *));
        r->chrom = (char **)realloc(r->chrom, r->cap * sizeof(char *));
        r->min_s = (uint32_t *)realloc(r->min_s, r->cap * 4);
        r->max_e = (uint32_t *)realloc(r->max_e, r->cap * 4);
        r->stamp = (uint32_t *)realloc(r->stamp, r->cap * 4);
        r->breadth = (uint64_t *)realloc(r->breadth, r->cap * 8);
        r->present = (uint8_t *)realloc(r->present, r->cap);
    }
    v = (uint32_t)r->n++;
    smap_put(&r->ids, id, idlen, v);
    r->id_str[v] = dupn(id, idlen);
    r->chrom[v] = NULL;
    r->stamp[v] = 0;
    r->min_s[v] = UINT32_MAX;
    r->max_e[v] = 0;
    r->breadth[v] = 0;
    r->present[v] = 0;
    return v;
}

typedef struct {
    uint32_t start, end, id;
    const uint8_t *seq;
    size_t seqlen;
} cline_t;

static int cmp_cline(const void *a, const void *b) {
    const cline_t *x = (const cline_t *)a, *y = (const cline_t *)b;
    return x->start < y->start ? -1 : x->start > y->start;
}

/* coverage.rs:277-378 compute_breadth_for_root on one root's block and its merged coverage, merged into the
 * global map with the rule of :417-428 (min start, max end, breadth summed; chrom of the first contribution --
 * hash order in the reference, file order of the blocks here). */
static void breadth_one_root(const uint8_t *slice, size_t n, const cspan_t *cov, size_t ncov, cres_t *res) {
    cline_t *ln = NULL;
    size_t nl = 0, cl = 0;
    if (utf8_valid(slice, n)) { /* :296 */
        size_t pos = 0;
        while (pos < n) {
            const uint8_t *lb = slice + pos;
            const uint8_t *nlp = (const uint8_t *)memchr(lb, '\n', n - pos);
            size_t ll = nlp ? (size_t)(nlp - lb) : n - pos;
            pos += ll + 1;
            if (ll == 0 || lb[0] == '#') continue; /* :298 */
            const uint8_t *col[9];
            size_t len[9];
            const uint8_t *p = lb, *e = lb + ll;
            int k = 0;
            for (; k < 8; k++) {
                const uint8_t *t = find_tab(p, e);
                if (!t) break;
                col[k] = p;
                len[k] = (size_t)(t - p);
                p = t + 1;
            }
            if (k < 8) continue;
            col[8] = p;
            len[8] = (size_t)(e - p);
            uint32_t s1, e1;
            if (!parse_u32_ascii(col[3], len[3], &s1) || !parse_u32_ascii(col[4], len[4], &e1)) continue; /* :306 */
            if (e1 == 0) continue;                                                                       /* :307 */
            if (s1 > e1) { uint32_t t = s1; s1 = e1; e1 = t; }                                           /* :308 */
            const uint8_t *id;
            size_t idlen;
            if (!fast_id(col[8], len[8], &id, &idlen)) continue; /* :313 */
            if (nl == cl) {
                cl = cl ? cl * 2 : 64;
                ln = (cline_t *)realloc(ln, cl * sizeof(cline_t));
            }
            ln[nl].start = s1 ? s1 - 1 : 0;
            ln[nl].end = e1;
            ln[nl].id = cres_id(res, id, idlen);
            ln[nl].seq = col[0];
            ln[nl].seqlen = len[0];
            nl++;
        }
    }
    if (nl == 0 || ncov == 0) { /* :325-327 */
        free(ln);
        return;
    }
    /* per-block accumulators, folded into the global map at the end (the per-root map of :369-377) */
    size_t nid = res->n;
    uint32_t *bmin = (uint32_t *)malloc(nid * 4), *bmax = (uint32_t *)malloc(nid * 4);
    size_t *pcs = (size_t *)calloc(nid + 1, sizeof(size_t));
    for (size_t i = 0; i < nid; i++) {
        bmin[i] = UINT32_MAX;
        bmax[i] = 0;
    }
    /* chrom of an ID inside this block = seqid of its first line in FILE order (:316-319), taken before the sort */
    for (size_t i = 0; i < nl; i++)
        if (res->stamp[ln[i].id] != res->serial) {
            res->stamp[ln[i].id] = res->serial;
            if (!res->chrom[ln[i].id]) res->chrom[ln[i].id] = dupn(ln[i].seq, ln[i].seqlen);
        }
    qsort(ln, nl, sizeof *ln, cmp_cline); /* :330 sort_unstable_by_key(start0) */
    cspan_t *pieces = (cspan_t *)malloc((nl * 2 + ncov * 2 + 4) * sizeof(cspan_t));
    uint32_t *piece_id = (uint32_t *)malloc((nl * 2 + ncov * 2 + 4) * 4);
    size_t np = 0, cap_p = nl * 2 + ncov * 2 + 4;
    size_t j = 0;
    for (size_t i = 0; i < nl; i++) { /* :339-364 two-pointer walk */
        const cline_t *fl = &ln[i];
        while (j < ncov && cov[j].e <= fl->start) j++;
        if (fl->start < bmin[fl->id]) bmin[fl->id] = fl->start;
        if (fl->end > bmax[fl->id]) bmax[fl->id] = fl->end;
        size_t k = j;
        while (k < ncov && cov[k].s < fl->end) {
            const uint32_t s = fl->start > cov[k].s ? fl->start : cov[k].s;
            const uint32_t e = fl->end < cov[k].e ? fl->end : cov[k].e;
            if (e > s) {
                if (np == cap_p) {
                    cap_p *= 2;
                    pieces = (cspan_t *)realloc(pieces, cap_p * sizeof(cspan_t));
                    piece_id = (uint32_t *)realloc(piece_id, cap_p * 4);
                }
                pieces[np].s = s;
                pieces[np].e = e;
                piece_id[np] = fl->id;
                np++;
            }
            if (cov[k].e <= fl->end)
                k++;
            else
                break;
        }
    }
    /* :367-377 per ID: union of its pieces; a row exists if length > 0 || breadth > 0 */
    for (size_t x = 0; x < np; x++) pcs[piece_id[x] + 1]++;
    for (size_t i = 0; i < nid; i++) pcs[i + 1] += pcs[i];
    cspan_t *byid = (cspan_t *)malloc((np ? np : 1) * sizeof(cspan_t));
    size_t *cur = (size_t *)malloc((nid + 1) * sizeof(size_t));
    memcpy(cur, pcs, (nid + 1) * sizeof(size_t));
    for (size_t x = 0; x < np; x++) byid[cur[piece_id[x]]++] = pieces[x];
    for (size_t i = 0; i < nid; i++) {
        if (bmin[i] == UINT32_MAX && bmax[i] == 0) continue; /* no line of this ID in the block */
        const uint64_t length = bmax[i] > bmin[i] ? (uint64_t)(bmax[i] - bmin[i]) : 0;
        const uint64_t br = union_len(byid + pcs[i], pcs[i + 1] - pcs[i]);
        if (length > 0 || br > 0) {
            res->present[i] = 1;
            if (bmin[i] < res->min_s[i]) res->min_s[i] = bmin[i];
            if (bmax[i] > res->max_e[i]) res->max_e[i] = bmax[i];
            res->breadth[i] += br;
        }
    }
    free(cur);
    free(byid);
    free(piece_id);
    free(pieces);
    free(pcs);
    free(bmin);
    free(bmax);
    free(ln);
}

/* coverage.rs:487-582 run with a .bed source: "id\tchr\tstart\tend\tbreadth\tfraction" rows (fraction with six
 * decimals, :466-472) for every ID of every root block that at least one region hits; rows sorted by id here
 * (hash-map order in the reference).  BED rows follow coverage.rs:230-256 (the same rules as depth's). */
int oracle_coverage_run(const char *gff_path, const char *bed_path, const char *out_path, char *err, size_t errlen) {
    gofe_t *g = NULL;
    size_t ng = 0;
    if (load_gof(gff_path, &g, &ng, err, errlen) != 0) return 1; /* :501 */
    map_t m;
    if (map_file(gff_path, &m) != 0) {
        set_err(err, errlen, "Cannot open GFF file: \"%s\"", gff_path);
        free(g);
        return 1;
    }
    oracle_index *ix = NULL;
    if (oracle_load_tree_index(gff_path, &ix, err, errlen) != 0) { /* :511 */
        unmap_file(&m);
        free(g);
        return 1;
    }
    uint32_t *regions = NULL;
    uint64_t nq = 0;
    if (oracle_depth_parse_bed(bed_path, ix, &regions, &nq, err, errlen) != 0) { /* :208-256: same row rules */
        oracle_index_free(ix);
        unmap_file(&m);
        free(g);
        return 1;
    }
    uint32_t max_fid = 0;
    for (size_t k = 0; k < ng; k++)
        if (g[k].fid > max_fid) max_fid = g[k].fid;
    uint32_t *rec_of = (uint32_t *)malloc(((size_t)max_fid + 2) * 4);
    memset(rec_of, 0xFF, ((size_t)max_fid + 2) * 4);
    for (size_t k = 0; k < ng; k++) rec_of[g[k].fid] = (uint32_t)k; /* index_cached(): last record wins */
    /* :258-268 by_root[root_fid].push((s, e)), a region once per root_fid */
    buf_t *by_root = (buf_t *)calloc((size_t)max_fid + 2, sizeof(buf_t));
    hits_t hits = {0};
    for (uint64_t i = 0; i < nq; i++) {
        const uint32_t chr = regions[3 * i], rs = regions[3 * i + 1], re = regions[3 * i + 2];
        if (chr >= ix->n_chr) continue;
        hits.n = 0;
        tree_query(ix->trees[chr], rs, re, &hits);
        for (size_t h = 0; h < hits.n; h++) {
            const uint32_t fid = hits.p[h]->root_fid;
            int dup = 0;
            for (size_t x = 0; x < h; x++)
                if (hits.p[x]->root_fid == fid) dup = 1;
            if (dup || fid > max_fid + 1u) continue;
            cspan_t sp = {rs, re};
            buf_push(&by_root[fid], &sp, sizeof sp);
        }
    }
    free(hits.p);
    cres_t res;
    memset(&res, 0, sizeof res);
    smap_init(&res.ids, 1024);
    /* :383-431 finalize_compute_breadth, roots in file order of their blocks */
    for (size_t k = 0; k < ng; k++) {
        const uint32_t fid = g[k].fid;
        if (rec_of[fid] != k || !by_root[fid].n) continue;
        const size_t ncov = merge_spans((cspan_t *)by_root[fid].p, by_root[fid].n / sizeof(cspan_t)); /* :401 */
        if (g[k].s == MISSING || g[k].e == MISSING || g[k].e <= g[k].s || g[k].e > m.n) continue;       /* :403 */
        res.serial++;
        breadth_one_root(m.p + g[k].s, (size_t)(g[k].e - g[k].s), (const cspan_t *)by_root[fid].p, ncov, &res);
    }
    for (size_t f = 0; f <= (size_t)max_fid + 1; f++) free(by_root[f].p);
    free(by_root);
    uint32_t *order = (uint32_t *)malloc((res.n ? res.n : 1) * 4);
    size_t no = 0;
    for (size_t i = 0; i < res.n; i++)
        if (res.present[i]) order[no++] = (uint32_t)i;
    qsort_r(order, no, 4, cmp_cstr_idx, res.id_str);
    buf_t out = {0};
    const char *hdr = "id\tchr\tstart\tend\tbreadth\tfraction\n";
    buf_push(&out, hdr, strlen(hdr));
    for (size_t x = 0; x < no; x++) {
        const uint32_t i = order[x];
        char num[160];
        const uint64_t length = res.max_e[i] > res.min_s[i] ? (uint64_t)(res.max_e[i] - res.min_s[i]) : 0; /* saturating_sub */
        const double frac = length > 0 ? (double)res.breadth[i] / (double)length : 0.0;
        buf_push(&out, res.id_str[i], strlen(res.id_str[i]));
        buf_push(&out, "\t", 1);
        buf_push(&out, res.chrom[i] ? res.chrom[i] : "", res.chrom[i] ? strlen(res.chrom[i]) : 0);
        int nn = snprintf(num, sizeof num, "\t%u\t%u\t%llu\t%.6f\n", res.min_s[i], res.max_e[i],
                          (unsigned long long)res.breadth[i], frac);
        buf_push(&out, num, (size_t)nn);
    }
    int rc = 0;
    if (out_path) {
        if (write_file(out_path, out.p, out.n) != 0) {
            set_err(err, errlen, "cannot write %s", out_path);
            rc = 1;
        }
    } else {
        fwrite(out.p, 1, out.n, stdout);
        fflush(stdout);
    }
    free(out.p);
    free(order);
    for (size_t i = 0; i < res.n; i++) {
        free(res.id_str[i]);
        free(res.chrom[i]);
    }
    free(res.id_str);
    free(res.chrom);
    free(res.min_s);
    free(res.max_e);
    free(res.stamp);
    free(res.breadth);
    free(res.present);
    smap_free(&res.ids);
    free(rec_of);
    free(regions);
    oracle_index_free(ix);
    unmap_file(&m);
    free(g);
    return rc;
}
