/* synth_text.c -- fast writers for the synthetic inputs of bench.py and the full-size tests (test infrastructure, not
 * product code): a GENCODE-shaped GFF3 around given root genes, and a BED file from u32 region triples.
 * Python's per-line writers take minutes at 3.4 M GFF lines / 100 M BED rows; this takes seconds.
 *
 *   synth_text gff <roots.bin> <names.txt> <out.gff> <tx_per_gene> <exons_per_tx> <seed>
 *       roots.bin: u32 n_chr, u32 n, chr_offsets[n_chr+1], start[n] (0-based), end[n] (exclusive)   -- native endian
 *       text: gene line (ID=gene%06u), per transcript an mRNA line (ID=<gene>.t<k>;Parent=<gene>) and per exon an exon
 *       line plus, with probability 0.6, a CDS line (children directly after their parent, file sorted by (seqid, gene
 *       start) because the roots are); 1-based closed coordinates.
 *   synth_text bed <regions.bin> <names.txt> <out.bed>
 *       regions.bin: u64 n, then n x {chr, start, end} u32; one "name\tstart\tend\n" row each.
 */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t g_state;
static uint64_t rnd(void) { /* splitmix64 */
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rnd01(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }
static uint32_t poisson_min1(double mean) { /* Knuth; mean is small */
    double l = 1.0, limit = __builtin_exp(-mean);
    uint32_t k = 0;
    do {
        k++;
        l *= rnd01();
    } while (l > limit);
    return k - 1 ? k - 1 : 1;
}

static char **read_names(const char *path, uint32_t *n_out) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    char **names = NULL, line[4096];
    uint32_t n = 0;
    while (fgets(line, sizeof line, f)) {
        size_t l = strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        names = (char **)realloc(names, (n + 1) * sizeof *names);
        names[n++] = strdup(line);
    }
    fclose(f);
    *n_out = n;
    return names;
}

static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

static int do_gff(int argc, char **argv) {
    if (argc < 8) return 2;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return perror(argv[2]), 1;
    uint32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 1;
    const uint32_t n_chr = hdr[0], n = hdr[1];
    uint32_t *co = (uint32_t *)malloc((n_chr + 1) * 4), *st = (uint32_t *)malloc((size_t)n * 4 + 4), *en = (uint32_t *)malloc((size_t)n * 4 + 4);
    if (fread(co, 4, n_chr + 1, f) != n_chr + 1 || fread(st, 4, n, f) != n || fread(en, 4, n, f) != n) return 1;
    fclose(f);
    uint32_t n_names = 0;
    char **names = read_names(argv[3], &n_names);
    if (!names || n_names < n_chr) return fprintf(stderr, "names: need %u\n", n_chr), 1;
    const double tx = atof(argv[5]), ex = atof(argv[6]);
    g_state = strtoull(argv[7], NULL, 10);
    FILE *o = fopen(argv[4], "wb");
    if (!o) return perror(argv[4]), 1;
    static char buf[1 << 22];
    setvbuf(o, buf, _IOFBF, sizeof buf);
    uint64_t lines = 0;
    fputs("##gff-version 3\n", o), lines++;
    uint32_t cuts[512];
    for (uint32_t c = 0; c < n_chr; c++) {
        for (uint32_t j = co[c]; j < co[c + 1]; j++) {
            const uint32_t gs = st[j] + 1, ge = en[j] < gs ? gs : en[j];
            const char strand = (rnd() & 1) ? '+' : '-';
            fprintf(o, "%s\tsynth\tgene\t%u\t%u\t.\t%c\t.\tID=gene%06u;gene_name=G%u;gene_type=protein_coding\n", names[c], gs, ge, strand, j, j);
            lines++;
            const uint32_t ntx = poisson_min1(tx);
            for (uint32_t t = 0; t < ntx; t++) {
                const uint32_t q = (ge - gs) / 4 + 1;
                const uint32_t ts = gs + (uint32_t)(rnd() % q);
                uint32_t te = ge - (uint32_t)(rnd() % q);
                if (te < ts) te = ts;
                fprintf(o, "%s\tsynth\tmRNA\t%u\t%u\t.\t%c\t.\tID=gene%06u.t%u;Parent=gene%06u;gene_name=G%u\n", names[c], ts, te, strand, j, t, j, j);
                lines++;
                uint32_t nex = poisson_min1(ex);
                if (nex > 256) nex = 256;
                for (uint32_t x = 0; x < 2 * nex; x++) cuts[x] = ts + (uint32_t)(rnd() % (te - ts + 1));
                qsort(cuts, 2 * nex, 4, cmp_u32);
                for (uint32_t x = 0; x < nex; x++) {
                    fprintf(o, "%s\tsynth\texon\t%u\t%u\t.\t%c\t.\tID=gene%06u.t%u.e%u;Parent=gene%06u.t%u\n", names[c], cuts[2 * x], cuts[2 * x + 1],
                            strand, j, t, x, j, t);
                    lines++;
                    if (rnd01() < 0.6) {
                        fprintf(o, "%s\tsynth\tCDS\t%u\t%u\t.\t%c\t0\tID=gene%06u.t%u.c%u;Parent=gene%06u.t%u\n", names[c], cuts[2 * x],
                                cuts[2 * x + 1], strand, j, t, x, j, t);
                        lines++;
                    }
                }
            }
        }
    }
    if (fclose(o)) return perror("close"), 1;
    printf("%" PRIu64 "\n", lines);
    return 0;
}

static int do_bed(int argc, char **argv) {
    if (argc < 5) return 2;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return perror(argv[2]), 1;
    uint64_t n;
    if (fread(&n, 8, 1, f) != 1) return 1;
    uint32_t n_names = 0;
    char **names = read_names(argv[3], &n_names);
    if (!names) return 1;
    size_t *nl = (size_t *)malloc(n_names * sizeof *nl);
    for (uint32_t i = 0; i < n_names; i++) nl[i] = strlen(names[i]);
    FILE *o = fopen(argv[4], "wb");
    if (!o) return perror(argv[4]), 1;
    static char buf[1 << 22];
    setvbuf(o, buf, _IOFBF, sizeof buf);
    enum { CH = 1 << 16 };
    uint32_t *r = (uint32_t *)malloc((size_t)CH * 12);
    char line[4200];
    for (uint64_t done = 0; done < n;) {
        const size_t k = (size_t)((n - done) < CH ? (n - done) : CH);
        if (fread(r, 12, k, f) != k) return fprintf(stderr, "short read\n"), 1;
        for (size_t i = 0; i < k; i++) {
            const uint32_t c = r[3 * i];
            if (c >= n_names) return fprintf(stderr, "chr %u out of range\n", c), 1;
            memcpy(line, names[c], nl[c]);
            char *p = line + nl[c];
            char tmp[12];
            for (int w = 1; w <= 2; w++) {
                *p++ = '\t';
                uint32_t v = r[3 * i + w];
                int d = 0;
                do tmp[d++] = (char)('0' + v % 10), v /= 10;
                while (v);
                while (d) *p++ = tmp[--d];
            }
            *p++ = '\n';
            fwrite(line, 1, (size_t)(p - line), o);
        }
        done += k;
    }
    fclose(f);
    if (fclose(o)) return perror("close"), 1;
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "gff")) return do_gff(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "bed")) return do_bed(argc, argv);
    fprintf(stderr, "usage: synth_text gff <roots.bin> <names.txt> <out.gff> <tx_per_gene> <exons_per_tx> <seed>\n"
                    "       synth_text bed <regions.bin> <names.txt> <out.bed>\n");
    return 2;
}
