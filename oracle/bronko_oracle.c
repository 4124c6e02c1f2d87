/*
 * bronko_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See bronko_oracle.h.
 *
 * Literal, single-threaded restatement of treangenlab/bronko v0.1.0 (reference paths are relative to
 * /root/reference).  Deliberately simple: no SIMD, no threads, the same loop structure as the reference so
 * that each block can be checked against the cited lines.  PARITY UNPINNED for the `call` side (header).
 */
#include "bronko_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static char g_err[512];
static void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* orc_last_error(void) { return g_err; }
void orc_free(void* p) { free(p); }

static void* xmalloc(size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); abort(); }
    return p;
}
static void* xcalloc(size_t n, size_t s) {
    void* p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void* xrealloc(void* q, size_t n) {
    void* p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); abort(); }
    return p;
}
static char* xstrdup(const char* s) {
    size_t n = strlen(s) + 1;
    char* p = (char*)xmalloc(n);
    memcpy(p, s, n);
    return p;
}

/* ======================================================================== src/lcb.rs */

/* lcb.rs:47-55 -- anything that is not ACGT/acgt encodes as 0 (= A) */
uint8_t orc_nt_to_bits(uint8_t nt) {
    switch (nt) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

/* lcb.rs:67-74 -- MSB first: the first base ends up in the highest 2 bits of the 2k-bit value */
uint64_t orc_kmer_to_u64(const uint8_t* kmer, int k) {
    uint64_t val = 0;
    for (int i = 0; i < k; i++) {
        val <<= 2;
        val |= (uint64_t)orc_nt_to_bits(kmer[i]);
    }
    return val;
}

/* lcb.rs:76-85 */
uint64_t orc_reverse_complement_u64(uint64_t kmer_val, int k) {
    uint64_t rc = 0;
    for (int i = 0; i < k; i++) {
        uint64_t two_bits = (kmer_val >> (2 * i)) & 3u;
        uint64_t comp = 3u ^ two_bits;
        rc <<= 2;
        rc |= comp;
    }
    return rc;
}

/* lcb.rs:87-95 -- (fwd,false) iff fwd < rev, else (rev,true) */
uint64_t orc_canonical_kmer(const uint8_t* kmer, int k, int* was_rc) {
    uint64_t fwd = orc_kmer_to_u64(kmer, k);
    uint64_t rev = orc_reverse_complement_u64(fwd, k);
    if (fwd < rev) { *was_rc = 0; return fwd; }
    *was_rc = 1;
    return rev;
}

/* lcb.rs:1-45 -- all arithmetic is u64 and wraps (release build), which matters for k = 31 */
void orc_assign_buckets(uint64_t kmer, int k, uint64_t* buckets) {
    uint64_t num_a[32] = {0}, val[32] = {0}, mu[32] = {0};
    uint64_t mask = 3ull << ((k - 1) * 2);
    uint64_t p = 1ull << ((k - 1) * 2);
    uint64_t cur = kmer & mask;

    val[0] = kmer - cur;
    mu[0] = (cur != 0) ? p + ((cur >> 2) * ((uint64_t)k - 1)) : val[0];
    uint64_t sum_mu = mu[0];

    for (int i = 1; i < k; i++) {
        num_a[i] = num_a[i - 1] + ((cur == 0) ? 1 : 0);
        mask >>= 2;
        cur = kmer & mask;
        p >>= 2;
        val[i] = val[i - 1] - cur;
        mu[i] = (cur != 0) ? p + ((cur >> 2) * ((uint64_t)k - (uint64_t)i - 1)) : val[i];
        sum_mu += mu[i];
    }

    mask = 3ull << ((k - 1) * 2);
    for (int i = 0; i < k; i++) {
        cur = kmer & mask;
        mask >>= 2;
        buckets[i] = sum_mu - mu[i] + val[i] - num_a[i] * cur + 1 + num_a[i];
    }
}

/* ======================================================================== index (src/build.rs) */

typedef struct { char* name; uint64_t len; uint8_t* seq; } seq_meta;   /* build.rs:31-36 */
typedef struct { char* name; int n_seq; seq_meta* seqs; } file_meta;   /* build.rs:39-43 */

struct orc_index {
    int k;                      /* BronkoIndex.k   build.rs:25 */
    uint64_t n_buckets;
    uint64_t* ids;              /* sorted ascending */
    uint64_t* off;              /* n_buckets + 1 */
    orc_bucket_info* entries;
    uint64_t n_entries;
    uint64_t hmask;             /* open-addressing lookup over ids */
    uint32_t* htab;
    int n_files;                /* ViralMetadata build.rs:46-50 */
    file_meta* files;
    int meta_k;
    uint64_t total_cells;
    uint64_t** cell_off;        /* [file][seq] */
};

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

typedef struct { uint64_t id; uint64_t ord; orc_bucket_info e; } pair_t;
static int pair_cmp(const void* a, const void* b) {
    const pair_t* x = (const pair_t*)a; const pair_t* y = (const pair_t*)b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    if (x->ord != y->ord) return x->ord < y->ord ? -1 : 1;
    return 0;
}

static void index_finish(orc_index* ix, pair_t* pairs, uint64_t n) {
    qsort(pairs, n, sizeof(pair_t), pair_cmp);
    uint64_t nb = 0;
    for (uint64_t i = 0; i < n; i++) if (i == 0 || pairs[i].id != pairs[i - 1].id) nb++;
    ix->n_buckets = nb;
    ix->n_entries = n;
    ix->ids = (uint64_t*)xmalloc(nb * sizeof(uint64_t));
    ix->off = (uint64_t*)xmalloc((nb + 1) * sizeof(uint64_t));
    ix->entries = (orc_bucket_info*)xmalloc(n * sizeof(orc_bucket_info));
    uint64_t b = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (i == 0 || pairs[i].id != pairs[i - 1].id) { ix->ids[b] = pairs[i].id; ix->off[b] = i; b++; }
        ix->entries[i] = pairs[i].e;
    }
    ix->off[nb] = n;
    uint64_t cap = 16;
    while (cap < nb * 2) cap <<= 1;
    ix->hmask = cap - 1;
    ix->htab = (uint32_t*)xmalloc(cap * sizeof(uint32_t));
    memset(ix->htab, 0xff, cap * sizeof(uint32_t));
    for (uint64_t i = 0; i < nb; i++) {
        uint64_t h = mix64(ix->ids[i]) & ix->hmask;
        while (ix->htab[h] != 0xffffffffu) h = (h + 1) & ix->hmask;
        ix->htab[h] = (uint32_t)i;
    }
    ix->total_cells = 0;
    ix->cell_off = (uint64_t**)xcalloc(ix->n_files, sizeof(uint64_t*));
    for (int f = 0; f < ix->n_files; f++) {
        ix->cell_off[f] = (uint64_t*)xcalloc(ix->files[f].n_seq, sizeof(uint64_t));
        for (int s = 0; s < ix->files[f].n_seq; s++) {
            ix->cell_off[f][s] = ix->total_cells;
            ix->total_cells += ix->files[f].seqs[s].len;
        }
    }
}

uint64_t orc_index_lookup(const orc_index* ix, uint64_t id, uint64_t* first) {
    uint64_t h = mix64(id) & ix->hmask;
    for (;;) {
        uint32_t b = ix->htab[h];
        if (b == 0xffffffffu) return 0;
        if (ix->ids[b] == id) { *first = ix->off[b]; return ix->off[b + 1] - ix->off[b]; }
        h = (h + 1) & ix->hmask;
    }
}

/* FASTA(.gz) reader with needletail 0.6 semantics as used at build.rs:156-189: record id = header line
 * without '>', sequence = all sequence lines with line terminators removed, case preserved. */
typedef struct { char* header; uint8_t* seq; uint64_t len; } fa_rec;

/* one text line of arbitrary length (terminator included); returns 0 at EOF */
static int gz_getline(gzFile g, char** line, size_t* cap, size_t* len_out) {
    size_t len = 0;
    for (;;) {
        if (!gzgets(g, *line + len, (int)(*cap - len))) { if (len == 0) return 0; break; }
        len += strlen(*line + len);
        if (len && (*line)[len - 1] == '\n') break;
        if (len + 1 < *cap) break;                 /* EOF without a trailing newline */
        *cap *= 2; *line = (char*)xrealloc(*line, *cap);
    }
    *len_out = len;
    return 1;
}

static int read_fasta(const char* path, fa_rec** out, int* n_out) {
    gzFile g = gzopen(path, "rb");
    if (!g) { set_err("Failed to parse fasta file: %s", path); return -1; }
    gzbuffer(g, 1 << 20);
    size_t cap = 1 << 16, len;
    char* line = (char*)xmalloc(cap);
    fa_rec* recs = NULL; int n = 0; uint64_t scap = 0;
    while (gz_getline(g, &line, &cap, &len)) {
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (len && line[0] == '>') {
            recs = (fa_rec*)xrealloc(recs, (n + 1) * sizeof(fa_rec));
            recs[n].header = xstrdup(line + 1);
            recs[n].seq = NULL; recs[n].len = 0; scap = 0;
            n++;
        } else if (n > 0 && len) {
            fa_rec* r = &recs[n - 1];
            if (r->len + len > scap) { scap = (r->len + len) * 2; r->seq = (uint8_t*)xrealloc(r->seq, scap); }
            memcpy(r->seq + r->len, line, len);
            r->len += len;
        }
    }
    free(line);
    gzclose(g);
    *out = recs; *n_out = n;
    return 0;
}

/* Path::file_stem (build.rs:161-165): file name without its final extension */
static char* file_stem(const char* path) {
    const char* base = strrchr(path, '/');
    base = base ? base + 1 : path;
    char* s = xstrdup(base);
    char* dot = strrchr(s, '.');
    if (dot && dot != s) *dot = 0;
    return s;
}

/* first whitespace-delimited token (build.rs:178-182) */
static char* first_token(const char* h) {
    while (*h == ' ' || *h == '\t') h++;
    size_t n = strcspn(h, " \t\r\n\v\f");
    char* s = (char*)xmalloc(n + 1);
    memcpy(s, h, n); s[n] = 0;
    return s;
}

static orc_index* build_from_files(int k, int n_files, file_meta* files) {
    orc_index* ix = (orc_index*)xcalloc(1, sizeof(orc_index));
    ix->k = k; ix->meta_k = k; ix->n_files = n_files; ix->files = files;
    uint64_t total = 0;
    for (int f = 0; f < n_files; f++)
        for (int s = 0; s < files[f].n_seq; s++)
            if (files[f].seqs[s].len >= (uint64_t)k) total += (files[f].seqs[s].len - k + 1) * (uint64_t)k;
    pair_t* pairs = (pair_t*)xmalloc(total * sizeof(pair_t));
    uint64_t n = 0;
    uint64_t buckets[32];
    /* build.rs:152-217 per file, then merged in file order (build.rs:223-228) == one pass in file order */
    for (int f = 0; f < n_files; f++) {
        for (int s = 0; s < files[f].n_seq; s++) {            /* seq_id: u8 counter build.rs:170,207 */
            const seq_meta* sm = &files[f].seqs[s];
            if (sm->len < (uint64_t)k) continue;              /* upstream would panic slicing seq[0..k] */
            for (uint64_t i = 0; i + k <= sm->len; i++) {     /* build.rs:191 */
                int canonical;
                uint64_t kb = orc_canonical_kmer(sm->seq + i, k, &canonical);  /* build.rs:193 */
                orc_assign_buckets(kb, k, buckets);                           /* build.rs:194 */
                for (int j = 0; j < k; j++) {                                 /* build.rs:196-204 */
                    pair_t* p = &pairs[n];
                    p->id = buckets[j]; p->ord = n;
                    memset(&p->e, 0, sizeof p->e);
                    p->e.file_id = (uint16_t)f; p->e.seq_id = (uint8_t)s; p->e.location = (uint32_t)i;
                    p->e.idx = (uint8_t)j; p->e.canonical = (uint8_t)canonical;
                    n++;
                }
            }
        }
    }
    index_finish(ix, pairs, n);
    free(pairs);
    return ix;
}

orc_index* orc_index_build(int k, const char* const* paths, int n_files) {
    file_meta* files = (file_meta*)xcalloc(n_files, sizeof(file_meta));
    for (int f = 0; f < n_files; f++) {
        fa_rec* recs; int nr;
        if (read_fasta(paths[f], &recs, &nr) != 0) return NULL;
        files[f].name = file_stem(paths[f]);
        files[f].n_seq = nr;
        files[f].seqs = (seq_meta*)xcalloc(nr, sizeof(seq_meta));
        for (int s = 0; s < nr; s++) {
            files[f].seqs[s].name = first_token(recs[s].header);
            files[f].seqs[s].len = recs[s].len;
            files[f].seqs[s].seq = recs[s].seq ? recs[s].seq : (uint8_t*)xmalloc(1);
            free(recs[s].header);
        }
        free(recs);
    }
    return build_from_files(k, n_files, files);
}

orc_index* orc_index_build_mem(int k, int n_files, const char* const* file_names, const int* n_seqs,
                               const char* const* seq_names, const uint8_t* const* seqs, const uint64_t* seq_lens) {
    file_meta* files = (file_meta*)xcalloc(n_files, sizeof(file_meta));
    int q = 0;
    for (int f = 0; f < n_files; f++) {
        files[f].name = xstrdup(file_names[f]);
        files[f].n_seq = n_seqs[f];
        files[f].seqs = (seq_meta*)xcalloc(n_seqs[f], sizeof(seq_meta));
        for (int s = 0; s < n_seqs[f]; s++, q++) {
            files[f].seqs[s].name = xstrdup(seq_names[q]);
            files[f].seqs[s].len = seq_lens[q];
            files[f].seqs[s].seq = (uint8_t*)xmalloc(seq_lens[q] + 1);
            memcpy(files[f].seqs[s].seq, seqs[q], seq_lens[q]);
        }
    }
    return build_from_files(k, n_files, files);
}

void orc_index_free(orc_index* ix) {
    if (!ix) return;
    for (int f = 0; f < ix->n_files; f++) {
        for (int s = 0; s < ix->files[f].n_seq; s++) { free(ix->files[f].seqs[s].name); free(ix->files[f].seqs[s].seq); }
        free(ix->files[f].seqs); free(ix->files[f].name);
        if (ix->cell_off) free(ix->cell_off[f]);
    }
    free(ix->cell_off); free(ix->files); free(ix->ids); free(ix->off); free(ix->entries); free(ix->htab); free(ix);
}

int orc_index_k(const orc_index* ix) { return ix->k; }
int orc_index_meta_k(const orc_index* ix) { return ix->meta_k; }
uint64_t orc_index_n_buckets(const orc_index* ix) { return ix->n_buckets; }
uint64_t orc_index_n_entries(const orc_index* ix) { return ix->n_entries; }
const uint64_t* orc_index_bucket_ids(const orc_index* ix) { return ix->ids; }
const uint64_t* orc_index_bucket_off(const orc_index* ix) { return ix->off; }
const orc_bucket_info* orc_index_entries(const orc_index* ix) { return ix->entries; }
int orc_index_n_files(const orc_index* ix) { return ix->n_files; }
const char* orc_index_file_name(const orc_index* ix, int f) { return ix->files[f].name; }
int orc_index_n_seqs(const orc_index* ix, int f) { return ix->files[f].n_seq; }
const char* orc_index_seq_name(const orc_index* ix, int f, int s) { return ix->files[f].seqs[s].name; }
uint64_t orc_index_seq_len(const orc_index* ix, int f, int s) { return ix->files[f].seqs[s].len; }
const uint8_t* orc_index_seq(const orc_index* ix, int f, int s) { return ix->files[f].seqs[s].seq; }
uint64_t orc_index_total_cells(const orc_index* ix) { return ix->total_cells; }
uint64_t orc_index_cell_offset(const orc_index* ix, int f, int s) { return ix->cell_off[f][s]; }

/* ------------------------------------------------------------------ .bkdb codec
 * bincode 2.0.1 `config::standard()` (build.rs:140-141, call.rs:187-188): little-endian, varint ints:
 * v < 251 -> 1 byte; 0xFB + u16; 0xFC + u32; 0xFD + u64.  u8/bool are one raw byte.  Struct fields in
 * declaration order (build.rs:23-60): BronkoIndex{k, global_index, metadata}. */
typedef struct { const uint8_t* p; const uint8_t* end; int bad; } rd_t;
static uint64_t rd_varint(rd_t* r) {
    if (r->p >= r->end) { r->bad = 1; return 0; }
    uint8_t b = *r->p++;
    if (b < 251) return b;
    int nb = b == 251 ? 2 : b == 252 ? 4 : b == 253 ? 8 : -1;
    if (nb < 0 || r->end - r->p < nb) { r->bad = 1; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < nb; i++) v |= (uint64_t)r->p[i] << (8 * i);
    r->p += nb;
    return v;
}
static uint8_t rd_u8(rd_t* r) {
    if (r->p >= r->end) { r->bad = 1; return 0; }
    return *r->p++;
}
static char* rd_string(rd_t* r) {
    uint64_t n = rd_varint(r);
    if (r->bad || (uint64_t)(r->end - r->p) < n) { r->bad = 1; return xstrdup(""); }
    char* s = (char*)xmalloc(n + 1);
    memcpy(s, r->p, n); s[n] = 0; r->p += n;
    return s;
}

orc_index* orc_bkdb_load(const char* path) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { set_err("Failed to open file '%s'", path); return NULL; }
    fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)xmalloc(sz);
    if (fread(buf, 1, sz, fp) != (size_t)sz) { fclose(fp); free(buf); set_err("short read"); return NULL; }
    fclose(fp);
    rd_t r = {buf, buf + sz, 0};
    orc_index* ix = (orc_index*)xcalloc(1, sizeof(orc_index));
    ix->k = (int)rd_varint(&r);
    uint64_t map_len = rd_varint(&r);
    uint64_t cap = 1 << 20, n = 0;
    pair_t* pairs = (pair_t*)xmalloc(cap * sizeof(pair_t));
    for (uint64_t m = 0; m < map_len && !r.bad; m++) {
        uint64_t key = rd_varint(&r);
        uint64_t cnt = rd_varint(&r);
        for (uint64_t e = 0; e < cnt && !r.bad; e++) {
            if (n == cap) { cap *= 2; pairs = (pair_t*)xrealloc(pairs, cap * sizeof(pair_t)); }
            pair_t* p = &pairs[n];
            p->id = key; p->ord = n;
            memset(&p->e, 0, sizeof p->e);
            p->e.file_id = (uint16_t)rd_varint(&r);
            p->e.seq_id = rd_u8(&r);
            p->e.location = (uint32_t)rd_varint(&r);
            p->e.idx = rd_u8(&r);
            p->e.canonical = rd_u8(&r);
            n++;
        }
    }
    ix->n_files = (int)rd_varint(&r);
    if (r.bad) { free(pairs); free(buf); free(ix); set_err("Failed to read Bronko Index from '%s'", path); return NULL; }
    ix->files = (file_meta*)xcalloc(ix->n_files, sizeof(file_meta));
    for (int f = 0; f < ix->n_files && !r.bad; f++) {
        ix->files[f].name = rd_string(&r);
        ix->files[f].n_seq = (int)rd_varint(&r);
        ix->files[f].seqs = (seq_meta*)xcalloc(ix->files[f].n_seq, sizeof(seq_meta));
        for (int s = 0; s < ix->files[f].n_seq && !r.bad; s++) {
            seq_meta* sm = &ix->files[f].seqs[s];
            sm->name = rd_string(&r);
            sm->len = rd_varint(&r);
            uint64_t vl = rd_varint(&r);
            if (r.bad || (uint64_t)(r.end - r.p) < vl) { r.bad = 1; break; }
            sm->seq = (uint8_t*)xmalloc(vl + 1);
            memcpy(sm->seq, r.p, vl); r.p += vl;
            if (vl != sm->len) sm->len = vl < sm->len ? vl : sm->len; /* defensive: never index past seq */
        }
    }
    ix->meta_k = (int)rd_varint(&r);
    if (r.bad || r.p != r.end) {
        set_err("Failed to read Bronko Index from '%s' (%s)", path, r.bad ? "truncated" : "trailing bytes");
        free(pairs); free(buf);
        return NULL;
    }
    index_finish(ix, pairs, n);
    free(pairs); free(buf);
    return ix;
}

static void wr_varint(FILE* fp, uint64_t v) {
    uint8_t b[9]; int n;
    if (v < 251) { b[0] = (uint8_t)v; n = 1; }
    else if (v <= 0xffffu) { b[0] = 251; b[1] = v & 0xff; b[2] = (v >> 8) & 0xff; n = 3; }
    else if (v <= 0xffffffffull) { b[0] = 252; for (int i = 0; i < 4; i++) b[1 + i] = (v >> (8 * i)) & 0xff; n = 5; }
    else { b[0] = 253; for (int i = 0; i < 8; i++) b[1 + i] = (v >> (8 * i)) & 0xff; n = 9; }
    fwrite(b, 1, n, fp);
}

/* build.rs:122-143.  Map entries are written in ascending id order (upstream order is hashbrown
 * iteration order, which no reader may depend on): decode-equality, not byte-equality. */
int orc_bkdb_save(const orc_index* ix, const char* path) {
    FILE* fp = fopen(path, "wb");
    if (!fp) { set_err("File path %s not valid", path); return -1; }
    wr_varint(fp, (uint64_t)ix->k);
    wr_varint(fp, ix->n_buckets);
    for (uint64_t b = 0; b < ix->n_buckets; b++) {
        wr_varint(fp, ix->ids[b]);
        wr_varint(fp, ix->off[b + 1] - ix->off[b]);
        for (uint64_t e = ix->off[b]; e < ix->off[b + 1]; e++) {
            const orc_bucket_info* x = &ix->entries[e];
            wr_varint(fp, x->file_id); fputc(x->seq_id, fp); wr_varint(fp, x->location);
            fputc(x->idx, fp); fputc(x->canonical ? 1 : 0, fp);
        }
    }
    wr_varint(fp, (uint64_t)ix->n_files);
    for (int f = 0; f < ix->n_files; f++) {
        size_t nl = strlen(ix->files[f].name);
        wr_varint(fp, nl); fwrite(ix->files[f].name, 1, nl, fp);
        wr_varint(fp, (uint64_t)ix->files[f].n_seq);
        for (int s = 0; s < ix->files[f].n_seq; s++) {
            const seq_meta* sm = &ix->files[f].seqs[s];
            nl = strlen(sm->name);
            wr_varint(fp, nl); fwrite(sm->name, 1, nl, fp);
            wr_varint(fp, sm->len);
            wr_varint(fp, sm->len); fwrite(sm->seq, 1, sm->len, fp);
        }
    }
    wr_varint(fp, (uint64_t)ix->meta_k);
    int bad = ferror(fp);
    fclose(fp);
    return bad ? -1 : 0;
}

/* ======================================================================== KMC3 contract */
/* call.rs:1166-1181: `kmc -k{k} -m2 -t{t} -b -ci{min_kmers} -cs1000000 <fastq> ...`, then
 * `kmc_tools transform ... dump` (call.rs:1203-1211) and load_kmers (call.rs:1241-1255).
 * KMC3 itself is an external C++ binary that is not under /root/reference (version unpinned, README.md:28):
 * this block restates its published behaviour for those flags. */
struct orc_kmer_counter {
    int k;
    uint64_t cap, n;          /* open addressing, EMPTY = ~0 */
    uint64_t* keys;
    uint64_t* vals;
    uint64_t total_reads, total_kmers;
    uint64_t n_out;
    uint64_t* out_kmers;
    uint64_t* out_counts;
};

orc_kmer_counter* orc_counter_new(int k) {
    orc_kmer_counter* c = (orc_kmer_counter*)xcalloc(1, sizeof *c);
    c->k = k; c->cap = 1 << 16;
    c->keys = (uint64_t*)xmalloc(c->cap * 8);
    memset(c->keys, 0xff, c->cap * 8);
    c->vals = (uint64_t*)xcalloc(c->cap, 8);
    return c;
}
void orc_counter_free(orc_kmer_counter* c) {
    if (!c) return;
    free(c->keys); free(c->vals); free(c->out_kmers); free(c->out_counts); free(c);
}
static void counter_grow(orc_kmer_counter* c) {
    uint64_t ocap = c->cap; uint64_t* ok = c->keys; uint64_t* ov = c->vals;
    c->cap = ocap * 2;
    c->keys = (uint64_t*)xmalloc(c->cap * 8);
    memset(c->keys, 0xff, c->cap * 8);
    c->vals = (uint64_t*)xcalloc(c->cap, 8);
    uint64_t m = c->cap - 1;
    for (uint64_t i = 0; i < ocap; i++) {
        if (ok[i] == ~0ull) continue;
        uint64_t h = mix64(ok[i]) & m;
        while (c->keys[h] != ~0ull) h = (h + 1) & m;
        c->keys[h] = ok[i]; c->vals[h] = ov[i];
    }
    free(ok); free(ov);
}
static inline void counter_inc(orc_kmer_counter* c, uint64_t key) {
    if ((c->n + 1) * 10 > c->cap * 6) counter_grow(c);
    uint64_t m = c->cap - 1, h = mix64(key) & m;
    for (;;) {
        if (c->keys[h] == key) { c->vals[h]++; return; }
        if (c->keys[h] == ~0ull) { c->keys[h] = key; c->vals[h] = 1; c->n++; return; }
        h = (h + 1) & m;
    }
}

/* One read: every window of k consecutive ACGT/acgt symbols is one k-mer occurrence, counted exactly as it
 * appears on the read strand (-b = no canonicalisation); any other symbol breaks the run. */
void orc_counter_add_read(orc_kmer_counter* c, const uint8_t* seq, uint64_t len) {
    const int k = c->k;
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t cur = 0; int valid = 0;
    c->total_reads++;
    for (uint64_t i = 0; i < len; i++) {
        int b;
        switch (seq[i]) {
            case 'A': case 'a': b = 0; break;
            case 'C': case 'c': b = 1; break;
            case 'G': case 'g': b = 2; break;
            case 'T': case 't': b = 3; break;
            default: b = -1;
        }
        if (b < 0) { valid = 0; cur = 0; continue; }
        cur = ((cur << 2) | (uint64_t)b) & mask;
        if (++valid >= k) { counter_inc(c, cur); c->total_kmers++; }
    }
}

uint64_t orc_counter_add_fastq(orc_kmer_counter* c, const char* path) {
    gzFile g = gzopen(path, "rb");
    if (!g) { set_err("cannot open %s", path); return (uint64_t)-1; }
    gzbuffer(g, 1 << 20);
    size_t cap = 1 << 16, len;
    char* line = (char*)xmalloc(cap);
    uint64_t n = 0; int ln = 0;
    while (gz_getline(g, &line, &cap, &len)) {   /* 4-line FASTQ records: @id / sequence / + / quality */
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if ((ln & 3) == 1) { orc_counter_add_read(c, (const uint8_t*)line, len); n++; }
        ln++;
    }
    free(line);
    gzclose(g);
    return n;
}

uint64_t orc_counter_finish(orc_kmer_counter* c, uint64_t ci, uint64_t cs, uint64_t cx, uint64_t* stats4) {
    free(c->out_kmers); free(c->out_counts);
    c->out_kmers = (uint64_t*)xmalloc(c->n * 8);
    c->out_counts = (uint64_t*)xmalloc(c->n * 8);
    uint64_t m = 0;
    for (uint64_t i = 0; i < c->cap; i++) {
        if (c->keys[i] == ~0ull) continue;
        uint64_t cnt = c->vals[i];
        if (cnt >= ci && cnt <= cx) {             /* -ci / -cx on the true count            */
            c->out_kmers[m] = c->keys[i];
            c->out_counts[m] = cnt > cs ? cs : cnt; /* -cs: stored counter saturates          */
            m++;
        }
    }
    c->n_out = m;
    if (stats4) { stats4[0] = c->total_reads; stats4[1] = c->total_kmers; stats4[2] = c->n; stats4[3] = m; }
    return m;
}
const uint64_t* orc_counter_kmers(const orc_kmer_counter* c) { return c->out_kmers; }
const uint64_t* orc_counter_counts(const orc_kmer_counter* c) { return c->out_counts; }

/* ======================================================================== map_kmers (call.rs:1257-1434) */
void orc_map_kmers(const orc_index* ix, const uint64_t* kmers, const uint64_t* counts, uint64_t n_kmers,
                   int n_fixed, int use_full_kmer,
                   uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                   uint64_t* stats, uint8_t* present) {
    const int k = ix->k;
    uint64_t buckets[32];
    uint64_t* hits = (uint64_t*)xcalloc(ix->n_files, 8);     /* per_genome_bucket_hits call.rs:1303 */
    int* touched = (int*)xmalloc(sizeof(int) * (ix->n_files ? ix->n_files : 1));

    /* window slice call.rs:1291-1300 */
    int w0, w1;
    if (use_full_kmer) { w0 = 0; w1 = k; }
    else if (n_fixed * 2 + 1 >= k) { w0 = 0; w1 = 0; }
    else { w0 = n_fixed; w1 = k - n_fixed - 1; }
    const uint64_t num_buckets_perfect = (uint64_t)(w1 - w0);   /* call.rs:1302 */

    for (uint64_t t = 0; t < n_kmers; t++) {
        /* the dumped k-mer string re-encoded by canonical_kmer (call.rs:1288): kmers[t] is kmer_to_u64 of it */
        const uint64_t fwd = kmers[t];
        const uint64_t n = counts[t];
        const uint64_t rev = orc_reverse_complement_u64(fwd, k);
        uint64_t kmer_bin; int rc;
        if (fwd < rev) { kmer_bin = fwd; rc = 0; } else { kmer_bin = rev; rc = 1; }   /* lcb.rs:90-94 */
        orc_assign_buckets(kmer_bin, k, buckets);                                     /* call.rs:1289 */

        int n_touched = 0;
        for (int j = w0; j < w1; j++) {                                               /* call.rs:1305 */
            uint64_t first;
            uint64_t cnt = orc_index_lookup(ix, buckets[j], &first);                  /* call.rs:1307 */
            for (uint64_t e = 0; e < cnt; e++) {                                      /* call.rs:1309 */
                const orc_bucket_info* info = &ix->entries[first + e];
                if (hits[info->file_id]++ == 0) touched[n_touched++] = info->file_id; /* call.rs:1316-1318 */
                /* output_maps.get(&file_id) / get_mut(seq): silently skipped when absent (call.rs:1324,1337) */
                if ((int)info->file_id >= ix->n_files || (int)info->seq_id >= ix->files[info->file_id].n_seq) continue;
                const uint64_t genome_pos = info->location;                           /* call.rs:1328 */
                const uint64_t nuc_x = info->idx;                                     /* call.rs:1329 */
                const uint64_t idx = genome_pos + nuc_x;                              /* call.rs:1334,1361 */
                if (idx >= ix->files[info->file_id].seqs[info->seq_id].len) continue; /* upstream would panic */
                const uint64_t cell = (ix->cell_off[info->file_id][info->seq_id] + idx) * 4;
                uint64_t bit_idx; int forward;
                if (info->canonical) {
                    const uint64_t pos = (uint64_t)k - nuc_x - 1;                     /* call.rs:1332 */
                    bit_idx = ((kmer_bin >> (2 * ((uint64_t)k - pos - 1))) & 3u) ^ 3u; /* call.rs:1333 */
                    forward = rc ? 1 : 0;                                             /* call.rs:1336-1357 */
                } else {
                    const uint64_t pos = nuc_x;                                       /* call.rs:1359 */
                    bit_idx = (kmer_bin >> (2 * ((uint64_t)k - pos - 1))) & 3u;       /* call.rs:1360 */
                    forward = rc ? 0 : 1;                                             /* call.rs:1363-1383 */
                }
                uint64_t* nk = forward ? fwd_nk : rev_nk;
                uint64_t* dp = forward ? fwd_depth : rev_depth;
                nk[cell + bit_idx] += 1;
                if (dp[cell + bit_idx] < n) dp[cell + bit_idx] = n;
            }
        }

        /* call.rs:1390-1418 */
        int n_perfect = 0, uniq = -1;
        for (int q = 0; q < n_touched; q++)
            if (hits[touched[q]] == num_buckets_perfect) { n_perfect++; uniq = touched[q]; }
        for (int q = 0; q < n_touched; q++) {
            const int f = touched[q];
            present[f] = 1;
            if (hits[f] == num_buckets_perfect) stats[f * 3 + 0] += 1;
            else if (hits[f] > 0) stats[f * 3 + 1] += 1;
            hits[f] = 0;
        }
        if (n_perfect == 1) { stats[uniq * 3 + 2] += 1; present[uniq] = 1; }
    }
    free(hits); free(touched);
}

/* call.rs:422-450 (single) and :452-502 (paired: caller passes the summed stats) */
int orc_pick_best_genome(const orc_index* ix, const uint64_t* stats, const uint8_t* present) {
    int best = -1; double best_score = 0.0;
    for (int f = 0; f < ix->n_files; f++) {
        if (!present[f]) continue;
        uint64_t genome_len = 0;
        for (int s = 0; s < ix->files[f].n_seq; s++) genome_len += ix->files[f].seqs[s].len;
        double score = (double)stats[f * 3 + 0] / (double)genome_len / 2.0;   /* call.rs:435 */
        if (score > best_score) { best_score = score; best = f; }             /* call.rs:443 */
    }
    return best;
}

/* ======================================================================== noise + calling */
void orc_call_params_default(orc_call_params* p) {   /* consts.rs:2-21 */
    p->k = 21; p->min_af = 0.03; p->no_end_filter = 0; p->no_strand_filter = 0; p->no_strand_balance_filter = 0;
    p->strand_balance_ratio = 0.1; p->n_per_strand = 2; p->strand_odds_max = 6.0; p->min_depth = 300;
    p->min_variant_depth = 3; p->variant_multiplier = 1.5;
}
void orc_map_params_default(orc_map_params* p) { p->n_fixed = 2; p->use_full_kmer = 0; p->ci = 3; p->cs = 1000000; p->cx = 1000000000ull; }

static const double k_tcrit[298] = {
#include "tcrit_table.inc"
};
/* StudentsT::new(0,1,n-2).inverse_cdf(1 - 0.001/n) (call.rs:924-925), tabulated: see gen_tcrit.py */
static double t_crit(uint64_t n) { return (n >= 3 && n <= 300) ? k_tcrit[n - 3] : NAN; }

static int u64_desc(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? 1 : x > y ? -1 : 0;
}

/* call.rs:799-967 */
void orc_baseline_noise(const uint64_t* fwd4, const uint64_t* rev4, uint64_t len, double* nmax, double* nmean, double* nstd) {
    enum { window_size = 100, max_table_len = window_size / 10, half_window = window_size / 2 };
    const double alpha = 0.001; (void)alpha;
    for (uint64_t i = 0; i < len; i++) { nmax[i] = 0.0; nmean[i] = 0.0; nstd[i] = 0.0; }       /* call.rs:809 */
    /* upstream sizes these len*3 but only indexes (i % 100)*3 + {0,1,2} (call.rs:813-814,828,850) */
    double window_counts[window_size * 3] = {0};
    int in_max[window_size * 3] = {0};
    double maxes[max_table_len] = {0};
    uint64_t n = 0; double s = 0.0, s2 = 0.0, mu, var;

    for (uint64_t i = 0; i < len + half_window; i++) {
        const uint64_t base_pos = (i % window_size) * 3;
        double freqs[4] = {0, 0, 0, 0};
        if (i < len) {                                                             /* call.rs:831-845 */
            uint64_t counts[4];
            for (int b = 0; b < 4; b++) counts[b] = fwd4[i * 4 + b] + rev4[i * 4 + b];
            qsort(counts, 4, sizeof(uint64_t), u64_desc);
            uint64_t total_depth = counts[0] + counts[1] + counts[2] + counts[3];
            if (total_depth != 0)
                for (int b = 0; b < 4; b++) freqs[b] = (double)counts[b] / (double)total_depth;
        }
        for (int j = 1; j < 4; j++) {                                              /* call.rs:848 */
            const uint64_t idx = base_pos + (j - 1);
            const double old = window_counts[idx];
            if (old > 0.0) {                                                       /* call.rs:854-870 */
                n -= 1; s -= old; s2 -= old * old;
                if (in_max[idx] == 1) {
                    for (int pos = 0; pos < max_table_len; pos++) {
                        if (fabs(maxes[pos] - old) < 1e-12) {
                            for (int q = pos; q < max_table_len - 1; q++) maxes[q] = maxes[q + 1];
                            maxes[max_table_len - 1] = 0.0;
                            break;
                        }
                    }
                    in_max[idx] = 0;
                }
            }
            const double maf = freqs[j];                                           /* call.rs:873 */
            if (maf > 0.0) {
                n += 1; s += maf; s2 += maf * maf;
                for (int q = max_table_len - 1; q >= 0; q--) {                     /* call.rs:880-889 */
                    if (maf > maxes[q]) {
                        if (q + 1 < max_table_len) maxes[q + 1] = maxes[q];
                        maxes[q] = maf;
                    } else break;
                }
                in_max[idx] = 1;
            } else {
                in_max[idx] = 0;
                window_counts[idx] = 0.0;
            }
            window_counts[idx] = maf;                                              /* call.rs:896 */
        }
        if (n != 0) { mu = s / (double)n; var = (s2 / (double)n) - mu * mu; }      /* call.rs:901-907 */
        else { mu = 0.0; var = 0.0; }

        int curr_max_idx = 0; uint64_t curr_n = n;
        double curr_s = s, curr_s2 = s2, curr_mu = mu, curr_var = var;
        while (curr_max_idx < max_table_len && maxes[curr_max_idx] != 0.0) {       /* call.rs:917 */
            const double candidate = maxes[curr_max_idx];
            const double sd = sqrt(curr_var);
            double tau;
            if (curr_n > 2) {                                                      /* call.rs:922-929 */
                const double tc = t_crit(curr_n);
                tau = (tc * ((double)curr_n - 1.0)) / (sqrt((double)curr_n) * sqrt((double)curr_n - 2.0 + tc * tc));
            } else tau = INFINITY;
            if (fabs(candidate - curr_mu) > tau * sd) {                            /* call.rs:934 */
                curr_s -= candidate;
                curr_s2 -= candidate;               /* sic: not candidate^2 (call.rs:936) */
                curr_n -= 1;
                if (curr_n > 0) { curr_mu = curr_s / (double)curr_n; curr_var = (curr_s2 / (double)curr_n) - curr_mu * curr_mu; }
                else { curr_mu = 0.0; curr_var = 0.0; }
                curr_max_idx += 1;
            } else break;
        }
        if (i >= half_window) {                                                    /* call.rs:953-962 */
            const uint64_t w = i - half_window;
            if (w < len) {
                /* curr_max_idx == 10 would index out of bounds upstream (panic); report 0.0 instead */
                nmax[w] = curr_max_idx < max_table_len ? maxes[curr_max_idx] : 0.0;
                nmean[w] = curr_mu;
                nstd[w] = sqrt(curr_var);
            }
        }
    }
}

/* call.rs:969-1150; sequences visited in metadata order (upstream: DashMap iteration order, call.rs:995) */
uint64_t orc_call_variants(const orc_index* ix, int file_id, const orc_call_params* p,
                           const uint64_t* fwd_depth, const uint64_t* rev_depth,
                           const uint64_t* fwd_nk, const uint64_t* rev_nk,
                           orc_vcf_record** out, uint64_t* n_major, uint64_t* n_minor,
                           double* breadth, double* depth_cov) {
    uint64_t cap = 64, nrec = 0;
    orc_vcf_record* recs = (orc_vcf_record*)xmalloc(cap * sizeof *recs);
    uint64_t num_minor = 0, num_major = 0, positions_covered = 0, total_positions = 0, total_coverage = 0;
    const file_meta* fm = &ix->files[file_id];
    for (int s = 0; s < fm->n_seq; s++) {
        const uint64_t len = fm->seqs[s].len;
        const uint64_t c0 = ix->cell_off[file_id][s] * 4;
        const uint64_t* row_f = fwd_depth + c0; const uint64_t* row_r = rev_depth + c0;
        const uint64_t* cnt_f = fwd_nk + c0;   const uint64_t* cnt_r = rev_nk + c0;
        double* nmax = (double*)xmalloc(len * 8); double* nmean = (double*)xmalloc(len * 8); double* nstd = (double*)xmalloc(len * 8);
        orc_baseline_noise(row_f, row_r, len, nmax, nmean, nstd);                  /* call.rs:1002 */
        int64_t start = 0, end = (int64_t)len;
        if (!p->no_end_filter) { start = p->k; end = (int64_t)len - p->k; }        /* call.rs:1013-1016 */
        total_positions += len;                                                    /* call.rs:1019 */
        for (int64_t i = start; i < end; i++) {
            const uint64_t* row = row_f + i * 4; const uint64_t* row_rev = row_r + i * 4;
            const uint64_t* count = cnt_f + i * 4; const uint64_t* count_rev = cnt_r + i * 4;
            const uint8_t ref_base = orc_nt_to_bits(fm->seqs[s].seq[i]);           /* call.rs:1029-1030 */
            uint64_t row_total[4], total_depth = 0;
            for (int b = 0; b < 4; b++) { row_total[b] = row[b] + row_rev[b]; total_depth += row_total[b]; }
            if (total_depth == 0) continue;                                        /* call.rs:1044 */
            positions_covered += 1; total_coverage += total_depth;
            for (int alt = 0; alt < 4; alt++) {
                if (alt == ref_base || row_total[alt] == 0) continue;              /* call.rs:1053 */
                double sor = p->strand_odds_max + 1.0;                             /* call.rs:1058 */
                if (!p->no_strand_filter) {
                    const double a = (double)row[ref_base] + 1.0, b = (double)row_rev[ref_base] + 1.0;
                    const double c = (double)row[alt] + 1.0, d = (double)row_rev[alt] + 1.0;
                    const double ref_total = a + b + c + d;
                    const double min_strand_depth = fmin(a + c, b + d);
                    const double min_strand_percent = min_strand_depth / ref_total;
                    if ((!p->no_strand_balance_filter) | (p->no_strand_balance_filter & (min_strand_percent >= p->strand_balance_ratio))) {
                        const double r = (a * d) / (b * c);                        /* call.rs:1075-1079 */
                        const double ref_ratio = fmin(a, b) / fmax(a, b);
                        const double alt_ratio = fmin(c, d) / fmax(c, d);
                        sor = log(r + (1.0 / r)) + log(ref_ratio) - log(alt_ratio);
                        if (sor > p->strand_odds_max) continue;                    /* call.rs:1082 */
                        const uint64_t c_k = count[alt], d_k = count_rev[alt];     /* call.rs:1087-1092 */
                        if (c_k < p->n_per_strand && d_k < p->n_per_strand) continue;
                    } else sor = -1.0;                                             /* call.rs:1094 */
                }
                const uint64_t alt_count = row_total[alt];
                const double af = (double)alt_count / (double)total_depth;         /* call.rs:1100 */
                const double y0 = p->variant_multiplier, p0 = 0.5, a0 = 0.03;
                const double factor = y0 + p0 * pow(a0, 100.0 * af);               /* call.rs:1105 */
                if (af < p->min_af || af < (fmax(factor, y0) * nmax[i])) continue; /* call.rs:1107 */
                if (af >= 0.5) num_major += 1;
                else {
                    if (total_depth < p->min_depth) continue;                      /* call.rs:1116 */
                    if (alt_count < p->min_variant_depth) continue;                /* call.rs:1119 */
                    num_minor += 1;
                }
                if (nrec == cap) { cap *= 2; recs = (orc_vcf_record*)xrealloc(recs, cap * sizeof *recs); }
                orc_vcf_record* rec = &recs[nrec++];
                rec->seq_id = s; rec->pos = (uint64_t)i + 1; rec->ref_base = ref_base; rec->alt_base = (uint8_t)alt;
                rec->fwd_ref = row[ref_base]; rec->rev_ref = row_rev[ref_base]; rec->fwd_alt = row[alt]; rec->rev_alt = row_rev[alt];
                rec->depth = total_depth; rec->af = af; rec->sor = sor;
            }
        }
        free(nmax); free(nmean); free(nstd);
    }
    *out = recs; *n_major = num_major; *n_minor = num_minor;
    *breadth = (double)positions_covered / (double)total_positions;                /* call.rs:1144 */
    *depth_cov = (double)total_coverage / (double)positions_covered;               /* call.rs:1145 */
    return nrec;
}

/* ======================================================================== names + writers */
static int ends_with(const char* s, const char* suf) {
    size_t n = strlen(s), m = strlen(suf);
    return n >= m && memcmp(s + n - m, suf, m) == 0;
}
/* util.rs:30-50 (trim_end_matches strips the suffix repeatedly) */
void orc_clean_sample_id(const char* path, char* buf, size_t buflen) {
    static const char* suffixes[] = {".fastq.gz", ".fasta.gz", "fna.gz", "fnq.gz", ".fq.gz", ".fastq", ".fasta", ".fnq", ".fna", ".fa", ".fq"};
    const char* base = strrchr(path, '/');
    base = base ? base + 1 : path;
    char* fn = xstrdup(base);
    for (size_t i = 0; i < sizeof suffixes / sizeof *suffixes; i++) {
        if (ends_with(fn, suffixes[i])) {
            size_t m = strlen(suffixes[i]);
            while (ends_with(fn, suffixes[i])) fn[strlen(fn) - m] = 0;
            snprintf(buf, buflen, "%s", fn);
            free(fn);
            return;
        }
    }
    char* dot = strrchr(fn, '.');
    if (dot && dot != fn) *dot = 0;
    snprintf(buf, buflen, "%s", fn);
    free(fn);
}

static char bits_to_char(unsigned b) { return b == 0 ? 'A' : b == 1 ? 'C' : b == 2 ? 'G' : b == 3 ? 'T' : 'N'; } /* lcb.rs:57-65 */

/* Rust `{:.3}` on f64: correctly rounded decimal; NaN prints "NaN", infinities "inf"/"-inf" */
static void fmt_f(char* buf, size_t n, double v, int prec) {
    if (isnan(v)) snprintf(buf, n, "NaN");
    else if (isinf(v)) snprintf(buf, n, v > 0 ? "inf" : "-inf");
    else snprintf(buf, n, "%.*f", prec, v);
}

/* call.rs:735-774 */
int orc_write_vcf(const char* out_path, const char* reads_path, const orc_index* ix, int file_id, const orc_vcf_record* recs, uint64_t n) {
    FILE* fp = fopen(out_path, "w");
    if (!fp) { set_err("Failed to create vcf output file"); return -1; }
    fprintf(fp, "##fileformat=VCFv4.5\n##source=bronko-v0.1.0\n##reference=file://%s\n", reads_path);
    const file_meta* fm = &ix->files[file_id];
    for (int s = 0; s < fm->n_seq; s++) {
        char* tok = first_token(fm->seqs[s].name);
        fprintf(fp, "##contig=<ID=%s,length=%llu>\n", tok, (unsigned long long)fm->seqs[s].len);
        free(tok);
    }
    fprintf(fp, "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total Depth\">\n");
    fprintf(fp, "##INFO=<ID=AF,Number=1,Type=Float,Description=\"Allele Frequency\">\n");
    fprintf(fp, "##INFO=<ID=DP4,Number=4,Type=Integer,Description=\"Fwd_ref,Rev_ref,Fwd_alt,Rev_alt\">\n");
    fprintf(fp, "##INFO=<ID=SOR,Number=4,Type=Float,Description=\"SOR\">\n");
    fprintf(fp, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    for (uint64_t i = 0; i < n; i++) {
        const orc_vcf_record* v = &recs[i];
        char* tok = first_token(fm->seqs[v->seq_id].name);
        char af[64], sor[64];
        fmt_f(af, sizeof af, v->af, 3); fmt_f(sor, sizeof sor, v->sor, 3);
        fprintf(fp, "%s\t%llu\t.\t%c\t%c\t.\tPASS\tDP=%llu;AF=%s;DP4=%llu,%llu,%llu,%llu;SOR=%s\n", tok,
                (unsigned long long)v->pos, bits_to_char(v->ref_base), bits_to_char(v->alt_base),
                (unsigned long long)v->depth, af, (unsigned long long)v->fwd_ref, (unsigned long long)v->rev_ref,
                (unsigned long long)v->fwd_alt, (unsigned long long)v->rev_alt, sor);
        free(tok);
    }
    fclose(fp);
    return 0;
}

/* call.rs:648-695 */
int orc_write_pileup(const char* out_path, const orc_index* ix, int file_id, const uint64_t* fwd_depth, const uint64_t* rev_depth) {
    FILE* fp = fopen(out_path, "w");
    if (!fp) { set_err("Failed to create tsv pileup file"); return -1; }
    fprintf(fp, "reference\tindex\tref\tA\tC\tG\tT\ta\tc\tg\tt\n");
    const file_meta* fm = &ix->files[file_id];
    for (int s = 0; s < fm->n_seq; s++) {
        const uint64_t c0 = ix->cell_off[file_id][s] * 4;
        for (uint64_t i = 0; i < fm->seqs[s].len; i++) {
            const uint64_t* f = fwd_depth + c0 + i * 4; const uint64_t* r = rev_depth + c0 + i * 4;
            fprintf(fp, "%s\t%llu\t%c\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\n", fm->seqs[s].name,
                    (unsigned long long)(i + 1), (char)fm->seqs[s].seq[i],
                    (unsigned long long)f[0], (unsigned long long)f[1], (unsigned long long)f[2], (unsigned long long)f[3],
                    (unsigned long long)r[0], (unsigned long long)r[1], (unsigned long long)r[2], (unsigned long long)r[3]);
        }
    }
    fclose(fp);
    return 0;
}

/* ======================================================================== orchestration (call.rs:212-387) */
void orc_sample_pileup(const orc_index* ix, const orc_map_params* mp, int n_mates,
                       const uint8_t* const* reads, const uint64_t* read_lens, const uint64_t* mate_off,
                       uint64_t* fwd_depth, uint64_t* rev_depth, uint64_t* fwd_nk, uint64_t* rev_nk,
                       uint64_t* stats, uint8_t* present, uint64_t* kmc_stats) {
    /* one KMC run per mate file (call.rs:301-307), then map R1, map R2 into the SAME arrays (call.rs:316-317) */
    for (int m = 0; m < n_mates; m++) {
        orc_kmer_counter* c = orc_counter_new(ix->k);
        for (uint64_t r = mate_off[m]; r < mate_off[m + 1]; r++) orc_counter_add_read(c, reads[r], read_lens[r]);
        uint64_t nk = orc_counter_finish(c, mp->ci, mp->cs, mp->cx, kmc_stats + 4 * m);
        orc_map_kmers(ix, orc_counter_kmers(c), orc_counter_counts(c), nk, mp->n_fixed, mp->use_full_kmer,
                      fwd_depth, rev_depth, fwd_nk, rev_nk, stats + (size_t)m * ix->n_files * 3, present + (size_t)m * ix->n_files);
        orc_counter_free(c);
    }
}
