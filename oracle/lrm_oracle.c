/*
 * lrm_oracle.c -- CPU restatement of the accaln hot path.  TEST INFRASTRUCTURE
 * ONLY (see lrm_oracle.h).  Plain C11 + OpenMP; every function cites the
 * reference file:line (relative to /root/reference) it restates.
 *
 * Nothing here is copied from the reference: the functions are re-derived from
 * its behaviour, including the quirks listed in SURVEY.md Appendix A.
 */
#define _GNU_SOURCE
#include "lrm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NEG (-(1 << 28))

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */

/* A/a=0 C/c=1 G/g=2 T/t=3 (lchash.c:38-42, fmidx.c:278-282).  Other bytes are
 * UB in the reference (uninitialised mapper[]); fenced here to -1.         */
static inline int dna_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

static inline void ct_add(orc_counters *dst, const orc_counters *src) {
    if (!dst) return;
    dst->n_lc += src->n_lc;       dst->n_fmi += src->n_fmi;
    dst->n_occ += src->n_occ;     dst->bwt_bytes += src->bwt_bytes;
    dst->n_sa += src->n_sa;       dst->n_seeds += src->n_seeds;
    dst->n_phases += src->n_phases; dst->cells += src->cells;
    dst->tiles += src->tiles;     dst->read_bases += src->read_bases;
    dst->cigar_ops += src->cigar_ops;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* index construction                                                        */
/* ------------------------------------------------------------------------ */

/* asindex.c:78-116 (create_meta): per record the upper-cased sequence, then
 * its reverse complement, offsets accumulated; one '$' at the very end.
 * N->random (asindex.c:53-60, srand48(time)) is not reproducible: fenced.  */
int orc_cat_build(const char *const *seqs, const uint64_t *lens, int nseq,
                  char **cat_out, uint64_t *cat_len, orc_mta *mta_out) {
    uint64_t total = 1;
    for (int i = 0; i < nseq; ++i) total += 2 * lens[i];
    char *cat = (char *) malloc(total + 1);
    if (!cat) return -2;
    uint64_t off = 0;
    for (int i = 0; i < nseq; ++i) {
        uint64_t n = lens[i];
        mta_out[i].offset = off;
        mta_out[i].seq_len = n;
        for (uint64_t p = 0; p < n; ++p) {
            char c = seqs[i][p];
            if (c > 0x60) c -= 0x20;                 /* asindex.c:63-68 */
            if (dna_code(c) < 0) { free(cat); return -1; }
            cat[off + p] = c;
        }
        for (uint64_t p = 0; p < n; ++p) {           /* asindex.c:70-75,100-102 */
            char c = cat[off + n - 1 - p];
            cat[off + n + p] = "TGCA"[dna_code(c)];
        }
        off += 2 * n;
    }
    cat[off++] = '$';                                /* asindex.c:109-110 */
    cat[off] = '\0';
    *cat_out = cat;
    *cat_len = off;
    return 0;
}

static const char *g_sa_text;
static uint64_t g_sa_len;
static int sa_cmp(const void *pa, const void *pb) {
    uint64_t a = *(const uint64_t *) pa, b = *(const uint64_t *) pb;
    if (a == b) return 0;
    const unsigned char *t = (const unsigned char *) g_sa_text;
    uint64_t n = g_sa_len - (a > b ? a : b);
    int r = memcmp(t + a, t + b, n);
    if (r) return r;
    /* one suffix is a prefix of the other: cannot happen with a unique '$'
     * terminator, but order the shorter one first to stay total.           */
    return a > b ? -1 : 1;
}

/* What pSAscan writes to .sa5 (psascan/sa_use.cc:8-18): the suffix array of
 * the whole file.  The text ends in '$' (0x24), smaller than A/C/G/T and
 * unique, so the SA is unique and any correct sorter is byte-identical.     */
int orc_sa_build(const char *text, uint64_t L, uint64_t *sa_out) {
    for (uint64_t i = 0; i < L; ++i) sa_out[i] = i;
    g_sa_text = text;
    g_sa_len = L;
    qsort(sa_out, L, sizeof(uint64_t), sa_cmp);
    return 0;
}

/* fmidx.c:101-125: counts over text[0..L-2] (the final '$' is excluded),
 * then exclusive prefix sums over all 256 byte values.                     */
static void build_c_table(const char *text, uint64_t L, uint64_t *tab) {
    memset(tab, 0, 256 * sizeof(uint64_t));
    for (uint64_t i = 0; i + 1 < L; ++i) tab[(unsigned char) text[i]]++;
    uint64_t sum = 0;
    for (int i = 0; i < 256; ++i) { uint64_t t = sum + tab[i]; tab[i] = sum; sum = t; }
}

static char *seq_from_num(uint64_t num, int hlen, char *buf) {
    for (int i = 0; i < hlen; ++i)                  /* lchash.c:19-33 */
        buf[hlen - 1 - i] = "ACGT"[(num >> (2 * i)) & 3];
    buf[hlen] = 0;
    return buf;
}

int orc_index_build(const char *cat, uint64_t L, const orc_mta *mta, int mta_len,
                    int o_ratio, int hlen, orc_index *out) {
    memset(out, 0, sizeof(*out));
    out->content = (char *) malloc(L + 1);
    memcpy(out->content, cat, L);
    out->content[L] = 0;
    out->con_len = L;
    out->mta_len = mta_len;
    out->mta = (orc_mta *) malloc(sizeof(orc_mta) * (mta_len > 0 ? mta_len : 1));
    memcpy(out->mta, mta, sizeof(orc_mta) * mta_len);

    out->sa = (uint64_t *) malloc(sizeof(uint64_t) * L);
    out->sa_len = L;
    orc_sa_build(cat, L, out->sa);

    orc_fmi *f = &out->fmi;
    f->c = (uint64_t *) malloc(256 * sizeof(uint64_t));
    build_c_table(cat, L, f->c);

    /* fmidx.c:76-98: bwt[i] = SA[i]==0 ? '$' : text[SA[i]-1] */
    f->length = L;
    f->bwt = (char *) malloc(L + 1);
    for (uint64_t i = 0; i < L; ++i)
        f->bwt[i] = out->sa[i] == 0 ? '$' : cat[out->sa[i] - 1];
    f->bwt[L] = 0;

    /* fmidx.c:128-150,186-190: o[4*(i/ratio)+sym] = #sym in bwt[0..i-1] at
     * i%ratio==0; '$' and NUL are not counted.                              */
    f->o_ratio = o_ratio;
    f->o_len = 4 * (L / (uint64_t) o_ratio + 1);
    f->o = (uint64_t *) calloc(f->o_len, sizeof(uint64_t));
    uint64_t tmp[4] = {0, 0, 0, 0};
    for (uint64_t i = 0; i < L; ++i) {
        if (i % (uint64_t) o_ratio == 0) {
            uint64_t id = i / (uint64_t) o_ratio;
            for (int s = 0; s < 4; ++s) f->o[4 * id + s] = tmp[s];
        }
        char c = f->bwt[i];
        if (c != '\0' && c != '$') tmp[dna_code(c)]++;
    }

    /* fmidx.c:153-163,194: csa[i] = SA[4i], csa_len = L/4+1 */
    f->csa_ratio = 4;
    f->csa_len = L / 4 + 1;
    f->csa = (uint64_t *) calloc(f->csa_len, sizeof(uint64_t));
    for (uint64_t i = 0; i < f->csa_len; ++i) {
        uint64_t r = i * 4;
        f->csa[i] = r < L ? out->sa[r] : 0;   /* sa_access past the end: fenced to 0 */
    }

    /* lchash.c:52-73: one backward search per hlen-mer, (0,0) when absent */
    uint64_t upper = (uint64_t) (1U << (2 * hlen));   /* lchash.c:75-77 */
    out->lch.hlen = hlen;
    out->lch.len = 2 * upper;
    out->lch.lc = (uint64_t *) malloc(sizeof(uint64_t) * 2 * upper);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < upper; ++i) {
        char buf[40];
        uint64_t k = 1, l = f->length - 1;
        uint64_t r = orc_fmi_aln(f, seq_from_num(i, hlen, buf), hlen, &k, &l, NULL);
        if (r == 0) k = l = 0;
        out->lch.lc[2 * i] = k;
        out->lch.lc[2 * i + 1] = l;
    }
    return 0;
}

void orc_index_free(orc_index *idx) {
    free(idx->fmi.c); free(idx->fmi.o); free(idx->fmi.csa); free(idx->fmi.bwt);
    free(idx->lch.lc); free(idx->sa); free(idx->content); free(idx->mta);
    memset(idx, 0, sizeof(*idx));
}

void orc_index_adopt(orc_index *out, const uint64_t *c256, const uint64_t *o,
                     uint64_t o_len, int o_ratio, const char *bwt, uint64_t length,
                     const uint64_t *lc, uint64_t lc_len, int hlen,
                     const uint64_t *sa, uint64_t sa_len, const char *content,
                     uint64_t con_len, const orc_mta *mta, int mta_len) {
    memset(out, 0, sizeof(*out));
    out->fmi.c = (uint64_t *) c256;   out->fmi.o = (uint64_t *) o;
    out->fmi.o_len = o_len;           out->fmi.o_ratio = o_ratio;
    out->fmi.bwt = (char *) bwt;      out->fmi.length = length;
    out->fmi.csa_ratio = 4;
    out->lch.lc = (uint64_t *) lc;    out->lch.len = lc_len;  out->lch.hlen = hlen;
    out->sa = (uint64_t *) sa;        out->sa_len = sa_len;
    out->content = (char *) content;  out->con_len = con_len;
    out->mta = (orc_mta *) mta;       out->mta_len = mta_len;
}

/* ------------------------------------------------------------------------ */
/* seed side                                                                 */
/* ------------------------------------------------------------------------ */

/* fmidx.c:277-293: rank of c in bwt[0..loc] = sampled count + scan of the
 * sample block up to and including loc (exact byte compare).               */
uint64_t orc_occ_access(const orc_fmi *idx, char c, uint64_t loc, orc_counters *ct) {
    uint64_t ratio = (uint64_t) idx->o_ratio;
    uint64_t id = loc / ratio;
    uint64_t count = 0;
    for (uint64_t i = id * ratio; i <= loc; ++i)
        if (idx->bwt[i] == c) count++;
    if (ct) { ct->n_occ++; ct->bwt_bytes += loc - id * ratio + 1; }
    return idx->o[4 * id + (uint64_t) dna_code(c)] + count;
}

/* fmidx.c:295-313: backward search over qry[len-1..0]; early exit when the
 * interval empties; k,l are written back even on failure.                  */
uint64_t orc_fmi_aln(const orc_fmi *idx, const char *qry, int len,
                     uint64_t *k, uint64_t *l, orc_counters *ct) {
    uint64_t kk = *k, ll = *l;
    if (ct && len > 0) ct->n_fmi++;
    for (int i = len - 1; i >= 0; --i) {
        char c = qry[i];
        kk = idx->c[(unsigned char) c] + orc_occ_access(idx, c, kk - 1, ct) + 1;
        ll = idx->c[(unsigned char) c] + orc_occ_access(idx, c, ll, ct);
        if (kk > ll) break;
    }
    *k = kk;
    *l = ll;
    return kk > ll ? 0 : ll - kk + 1;
}

/* lchash.c:36-49: 2-bit pack, first char most significant */
uint64_t orc_num_from_seq(const char *seq, int hlen) {
    uint64_t sum = 0;
    for (int i = 0; i < hlen; ++i) {
        sum += (uint64_t) dna_code(seq[i]);
        sum <<= 2;
    }
    return sum >> 2;
}

/* lchash.c:89-104 (+ lc_access :12-16) */
uint64_t orc_lc_aln(const char *qry, int qlen, uint64_t *k, uint64_t *l,
                    const orc_fmi *fmi, const orc_lch *h, orc_counters *ct) {
    int left = qlen - h->hlen;
    if (ct) ct->n_seeds++;
    if (qlen >= h->hlen) {
        uint64_t num = orc_num_from_seq(qry + left, h->hlen);
        *k = h->lc[2 * num];
        *l = h->lc[2 * num + 1];
        if (ct) ct->n_lc++;
    } else {
        *k = 1;
        *l = fmi->length - 1;
    }
    if (*k == 0 && *l == 0) return 0;
    return orc_fmi_aln(fmi, qry, left, k, l, ct);
}

/* fmidx.c:18-33 with sa_buf resident: loc > len -> 0 */
uint64_t orc_sa_access(const orc_index *idx, uint64_t loc) {
    if (loc > idx->sa_len) return 0;
    if (loc == idx->sa_len) return 0;   /* reference reads one past the end here: fenced */
    return idx->sa[loc];
}

/* fmidx.c:315-331 (unused by the aligner; kept for the "next" rows) */
uint64_t orc_csa_access(const orc_fmi *fmi, uint64_t loc) {
    uint64_t ratio = (uint64_t) fmi->csa_ratio;
    int counter = 0;
    while (loc % ratio != 0) {
        char c = fmi->bwt[loc];
        if (c == '$') return (uint64_t) counter;
        loc = fmi->c[(unsigned char) c] + orc_occ_access(fmi, c, loc, NULL) - 1;
        counter++;
        if (counter > 5 * fmi->csa_ratio) return 0;
    }
    return fmi->csa[loc / ratio] + (uint64_t) counter;
}

/* histo.c:9-24 */
orc_histo *orc_histo_init(uint32_t cap) {
    orc_histo *h = (orc_histo *) malloc(sizeof(orc_histo));
    h->cap = cap ? cap : 1;
    h->size = 0;
    h->entries = (orc_entry *) malloc(sizeof(orc_entry) * h->cap);
    return h;
}

void orc_histo_destroy(orc_histo *h) {
    free(h->entries);
    free(h);
}

/* histo.c:42-56 (+ push :30-40, key_hash :26-28): linear scan; on a bucket
 * hit val++ and key=min; otherwise append {key,1,key>>4}.                   */
void orc_histo_add(orc_histo *h, uint64_t key) {
    uint64_t bucket = key >> 4;
    int found = 0;
    for (uint32_t i = 0; i < h->size; ++i) {
        if (h->entries[i].bucket == bucket) {
            found = 1;
            h->entries[i].val += 1;
            if (key < h->entries[i].key) h->entries[i].key = key;
        }
    }
    if (!found) {
        if (h->size + 1 >= h->cap) {          /* grow; capacity is not observable */
            h->cap *= 2;
            h->entries = (orc_entry *) realloc(h->entries, sizeof(orc_entry) * h->cap);
        }
        orc_entry e = {key, 1, bucket};
        h->entries[h->size++] = e;
    }
}

/* histo.c:84-96: stable top-2 by val (strict >), returns top1.val+top2.val */
uint64_t orc_histo_find_2_max(orc_histo *h, orc_entry *store) {
    memset(store, 0, 2 * sizeof(orc_entry));
    for (uint32_t i = 0; i < h->size; ++i) {
        orc_entry e = h->entries[i];
        if (store[1].val < e.val && store[0].val < e.val) {
            store[1] = store[0];
            store[0] = e;
        } else if (store[1].val < e.val && store[0].val >= e.val) {
            store[1] = e;
        }
    }
    return store[0].val + store[1].val;
}

static void trace_seed(orc_trace *t, int j, uint64_t rr, uint64_t k, uint64_t l) {
    if (!t) return;
    if (t->n_seeds == t->cap_seeds) {
        t->cap_seeds = t->cap_seeds ? 2 * t->cap_seeds : 1024;
        t->seeds = (orc_seed_rec *) realloc(t->seeds, t->cap_seeds * sizeof(orc_seed_rec));
    }
    orc_seed_rec r = {j, rr, k, l};
    t->seeds[t->n_seeds++] = r;
}

static void trace_phase(orc_trace *t, int iter, const orc_entry *cand, uint64_t v, int decided) {
    if (!t) return;
    if (t->n_phases == t->cap_phases) {
        t->cap_phases = t->cap_phases ? 2 * t->cap_phases : 64;
        t->phases = (orc_phase_rec *) realloc(t->phases, t->cap_phases * sizeof(orc_phase_rec));
    }
    orc_phase_rec r;
    r.iter = iter; r.top1 = cand[0]; r.top2 = cand[1]; r.v = v; r.decided = decided;
    t->phases[t->n_phases++] = r;
}

/* alnmain.c:333-405 for one read */
int orc_seed_read(const orc_index *idx, const char *read, uint32_t len,
                  uint32_t seed_len, uint32_t thres, orc_entry *best,
                  orc_trace *trace, orc_counters *ct) {
    const int sl = (int) seed_len;
    const int gl = 1;                                   /* alnmain.c:342 */
    orc_histo *ot = orc_histo_init(thres);
    orc_entry cand[2];
    memset(cand, 0, sizeof(cand));
    memset(best, 0, sizeof(*best));
    int iter;
    int phases = 0;
    /* alnmain.c:353 compares j with the unsigned len - sl; len < sl wraps in
     * the reference (out-of-bounds reads).  Fenced: no seeds.               */
    const uint32_t jlimit = len > (uint32_t) sl ? len - (uint32_t) sl : 0;

    for (iter = 0; iter < sl + gl; ++iter) {
        orc_histo *in = orc_histo_init(thres);
        phases++;
        if (ct) ct->n_phases++;
        for (int j = iter; (uint32_t) j < jlimit; j += sl + gl) {
            uint64_t kk = 1, ll = idx->fmi.length - 1, rr;
            rr = orc_lc_aln(read + j, sl, &kk, &ll, &idx->fmi, &idx->lch, ct);
            trace_seed(trace, j, rr, kk, ll);
            if (rr > 0 && rr < (uint64_t) thres) {
                for (uint64_t k = kk; k <= ll; ++k) {
                    uint64_t l = orc_sa_access(idx, k) - (uint64_t) (int64_t) j;
                    if (ct) ct->n_sa++;
                    orc_histo_add(in, l);
                }
            }
        }
        int num_seeds = (int) (len / (uint32_t) (sl + gl));     /* alnmain.c:371 */
        if (num_seeds > 0) {
            uint64_t v = orc_histo_find_2_max(in, cand);
            double score = (double) v / num_seeds;
            if (score > 0.6) {                                  /* alnmain.c:378 */
                *best = cand[0];
                trace_phase(trace, iter, cand, v, 1);
                orc_histo_destroy(in);
                break;
            } else {
                trace_phase(trace, iter, cand, v, 0);
                if (cand[0].val != 0) orc_histo_add(ot, cand[0].key);
            }
        }
        orc_histo_destroy(in);
    }
    if (iter >= sl + gl - 1) {                                  /* alnmain.c:400-403 */
        orc_histo_find_2_max(ot, cand);
        *best = cand[0];
    }
    orc_histo_destroy(ot);
    if (ct) ct->read_bases += len;
    return phases;
}

/* ------------------------------------------------------------------------ */
/* extend side                                                               */
/* ------------------------------------------------------------------------ */

/* alnmain.c:151-176, u64 arithmetic reproduced as written */
int orc_seq_lookup(const orc_mta *table, int len, uint64_t loc, uint32_t qlen,
                   orc_seq_meta *result) {
    for (int i = 0; i < len; ++i) {
        uint64_t sl = table[i].seq_len;
        uint64_t start = table[i].offset;
        uint64_t end = start + sl * 2;
        if (loc >= start && loc + qlen <= start + sl) {
            result->strand = 0;
            result->seq_id = i;
            result->loc = loc;
            result->off = loc - start;
            return 1;
        } else if (loc >= start + sl && loc + qlen <= end) {
            result->strand = 1;
            result->seq_id = i;
            result->off = end - loc - qlen;
            result->loc = result->off + start;
            return 1;
        }
    }
    return 0;
}

/* alnmain.c:27-60 */
void orc_rev_comp_in_place(char *seq, uint32_t len) {
    for (uint32_t i = 0; i < len; ++i) {
        int code = dna_code(seq[i]);
        seq[i] = code < 0 ? 'N' : "TGCA"[code];
    }
    for (uint32_t i = 0; i < len / 2; ++i) {
        char c = seq[i];
        seq[i] = seq[len - 1 - i];
        seq[len - 1 - i] = c;
    }
}

/*
 * docs/GACT_SPEC.md -- tiled, banded extension from the anchor (0,0).
 * The reference's simple_gact (mutils.c:97-103 call site; submodule source
 * absent) is restated from the published Darwin GACT algorithm: fixed tile
 * T, overlap O, the traceback of a non-final tile is kept only for its first
 * T-O bases and the next tile starts where it stopped.
 *
 * Tile DP on lattice points (a,b), 0<=a<=tq, 0<=b<=tt, band -W/2 <= b-a < W/2:
 *   R[a][b] = 0                                     if a==tq or b==tt   (free exit)
 *   R[a][b] = max( R[a+1][b+1] + (q[a]==d[b] ? +1 : -1),     DIAG
 *                  R[a+1][b]   - 1,                          INS  (query base)
 *                  R[a][b+1]   - 1 )                         DEL  (target base)
 * out-of-band neighbours count as -inf; ties: DIAG, then INS, then DEL.
 */
int orc_gact(const char *q, int n, const char *d, int m, orc_gact_params gp,
             uint8_t *ops, int *n_ops, orc_counters *ct) {
    const int T = gp.T, O = gp.O, W = gp.W;
    *n_ops = 0;
    if (T <= 0 || O < 0 || O >= T || W < 2 || (W & 1) || n < 0 || m < 0) return -1;
    const int hw = W / 2;
    const int cap = T - O;
    const int pitch = T + 2;
    int *R = (int *) malloc(sizeof(int) * (size_t) (T + 2) * (size_t) pitch);
    uint8_t *P = (uint8_t *) malloc((size_t) (T + 2) * (size_t) pitch);
    if (!R || !P) { free(R); free(P); return -1; }
    int i = 0, j = 0, nops = 0, score = 0;
    uint64_t cells = 0, tiles = 0;

    while (i < n && j < m) {
        const int tq = (n - i) < T ? (n - i) : T;
        const int tt = (m - j) < T ? (m - j) : T;
        const int last = (i + tq == n);
        tiles++;
        for (int a = tq; a >= 0; --a) {
            int blo = a - hw; if (blo < 0) blo = 0;
            int bhi = a + hw - 1; if (bhi > tt) bhi = tt;
            for (int b = bhi; b >= blo; --b) {
                int *r = &R[a * pitch + b];
                if (a == tq || b == tt) { *r = 0; continue; }
                /* (a+1,b+1) shares the diagonal: always in band */
                int cd = R[(a + 1) * pitch + b + 1] + (q[i + a] == d[j + b] ? 1 : -1);
                /* (a+1,b): diagonal b-a-1 >= -hw ? */
                int ci = (b - a - 1 >= -hw) ? R[(a + 1) * pitch + b] - 1 : ORC_NEG;
                /* (a,b+1): diagonal b-a+1 <= hw-1 ? */
                int cdl = (b - a + 1 <= hw - 1) ? R[a * pitch + b + 1] - 1 : ORC_NEG;
                int best = cd; uint8_t p = 0;
                if (ci > best || cdl > best) {
                    if (ci >= cdl) { best = ci; p = 1; } else { best = cdl; p = 2; }
                }
                *r = best;
                P[a * pitch + b] = p;
                cells++;
            }
        }
        int a = 0, b = 0;
        /* the walk keeps at most T-O bases of either sequence; in the tile that holds the read's
         * end it may run on to the edge, but never past anti-diagonal 2(T-O), so the pointers a
         * walk can touch are bounded the same way in every tile                              */
        while (a < tq && b < tt && (last ? (a + b < 2 * cap) : (a < cap && b < cap))) {
            uint8_t p = P[a * pitch + b];
            if (p == 0) {
                int eq = q[i + a] == d[j + b];
                ops[nops++] = eq ? '=' : 'X';
                score += !eq;
                a++; b++;
            } else if (p == 1) {
                ops[nops++] = 'I'; score++; a++;
            } else {
                ops[nops++] = 'D'; score++; b++;
            }
        }
        i += a;
        j += b;
    }
    while (i < n) { ops[nops++] = 'I'; score++; i++; }   /* target exhausted */
    free(R);
    free(P);
    *n_ops = nops;
    if (ct) { ct->cells += cells; ct->tiles += tiles; ct->cigar_ops += (uint64_t) nops; }
    return score;
}

/* alnmain.c:425-451 for one read (cigar_align = mutils.c:94-105) */
int orc_extend_read(const orc_index *idx, char *read, uint32_t len,
                    const orc_entry *best, orc_gact_params gp,
                    uint8_t *ops, int *n_ops, int *score, orc_seq_meta *meta,
                    orc_counters *ct) {
    uint64_t loc = best->key;                                   /* alnmain.c:427 */
    memset(meta, 0, sizeof(*meta));
    meta->seq_id = -1;
    *n_ops = 0;
    *score = -1;
    int mr = orc_seq_lookup(idx->mta, idx->mta_len, loc, len, meta);
    /* Reference: on mr==0 the meta struct is uninitialised yet still used
     * (alnmain.c:430-444).  Fenced: no extension, score -1.  Also fenced: a
     * window that leaves the text (only reachable through u64 wrap).        */
    if (mr == 0 || len == 0 || meta->loc >= idx->con_len ||
        (uint64_t) len > idx->con_len - meta->loc) {
        if (mr) { mr = 0; memset(meta, 0, sizeof(*meta)); meta->seq_id = -1; }
        return mr;
    }
    if (meta->strand == 1) orc_rev_comp_in_place(read, len);    /* alnmain.c:433-438 */
    *score = orc_gact(read, (int) len, idx->content + meta->loc, (int) len, gp,
                      ops, n_ops, ct);
    return mr;
}

int orc_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen) {
    int w = 0;
    if (n_ops == 0) {
        if (buflen > 1) { buf[0] = '*'; buf[1] = 0; return 1; }
        return 0;
    }
    int i = 0;
    while (i < n_ops) {
        /* '=' and 'X' both print as M (SAM alignment match) */
        char o = (ops[i] == '=' || ops[i] == 'X') ? 'M' : (char) ops[i];
        int run = 0;
        while (i < n_ops) {
            char o2 = (ops[i] == '=' || ops[i] == 'X') ? 'M' : (char) ops[i];
            if (o2 != o) break;
            run++; i++;
        }
        int r = snprintf(buf + w, (size_t) (buflen - w), "%d%c", run, o);
        if (r < 0 || r >= buflen - w) return -1;
        w += r;
    }
    return w;
}

/* ------------------------------------------------------------------------ */
/* batch drivers                                                             */
/* ------------------------------------------------------------------------ */

int orc_seed_batch(const orc_index *idx, const char *reads, uint64_t stride,
                   const uint32_t *lens, uint64_t n, uint32_t seed_len,
                   uint32_t thres, orc_entry *best, int32_t *phases_out,
                   orc_counters *ct, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    orc_counters total;
    memset(&total, 0, sizeof(total));
#pragma omp parallel num_threads(nthreads)
    {
        orc_counters local;
        memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 4)
        for (uint64_t i = 0; i < n; ++i) {
            int ph = orc_seed_read(idx, reads + i * stride, lens[i], seed_len, thres,
                                   &best[i], NULL, ct ? &local : NULL);
            if (phases_out) phases_out[i] = ph;
        }
#pragma omp critical
        ct_add(&total, &local);
    }
    if (ct) ct_add(ct, &total);
    return 0;
}

int orc_extend_batch(const orc_index *idx, char *reads, uint64_t stride,
                     const uint32_t *lens, uint64_t n, const orc_entry *best,
                     orc_gact_params gp, uint8_t *store, uint64_t store_stride,
                     int32_t *n_ops, int32_t *score, orc_seq_meta *meta,
                     int32_t *meta_r, orc_counters *ct, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    orc_counters total;
    memset(&total, 0, sizeof(total));
    int bad = 0;
#pragma omp parallel num_threads(nthreads)
    {
        orc_counters local;
        memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 4)
        for (uint64_t i = 0; i < n; ++i) {
            if ((uint64_t) lens[i] * 2 > store_stride) {
#pragma omp atomic write
                bad = 1;
                continue;
            }
            int no = 0, sc = -1;
            meta_r[i] = orc_extend_read(idx, reads + i * stride, lens[i], &best[i], gp,
                                        store + i * store_stride, &no, &sc, &meta[i],
                                        ct ? &local : NULL);
            n_ops[i] = no;
            score[i] = sc;
        }
#pragma omp critical
        ct_add(&total, &local);
    }
    if (ct) ct_add(ct, &total);
    return bad ? -1 : 0;
}
