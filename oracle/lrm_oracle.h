/*
 * lrm_oracle.h -- CPU restatement of the seed-and-extend hot path of
 * lisanhu/LongReadMapper (accaln).
 *
 * *** TEST INFRASTRUCTURE ONLY. ***
 * This is the parity oracle and the "port" CPU baseline.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (longreadmapper_amd/, include/) never links, imports or calls
 * anything in oracle/.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - voting (histo_*)          : PINNED against the reference's own histo.c,
 *                                 compiled unmodified into oracle/_ref/.
 *   - ui40 SA element format    : PINNED against the reference's sa_use.h
 *                                 inline functions (oracle/_ref harness).
 *   - FM index / lchash / seeds : pinned by the known-answer vector recorded
 *                                 in SURVEY.md section 4 (captured from a compiled
 *                                 reference during the survey) + the recipes
 *                                 of the reference's own tests
 *                                 (test/test-fmidx.cc, test/test-lchash.cc).
 *                                 fmidx.c / lchash.c themselves cannot be
 *                                 compiled here: they include gact/gact.h and
 *                                 mlog/logger.h, which are un-vendored
 *                                 submodules.
 *   - GACT extension / CIGAR    : PARITY UNPINNED.  The reference's GACT
 *                                 source (submodule lisanhu/GACT, pin unknown)
 *                                 is absent; orc_gact() restates the published
 *                                 Darwin GACT algorithm per docs/GACT_SPEC.md.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef LRM_ORACLE_H
#define LRM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* histo/histo.h:21-29 */
typedef struct { uint64_t key, val, bucket; } orc_entry;
typedef struct { orc_entry *entries; uint32_t cap, size; } orc_histo;

/* fmidx/fmidx.h:16-21 (same field order as dna_fmi) */
typedef struct {
    uint64_t length, o_len, csa_len;
    uint64_t *c, *o, *csa;
    int o_ratio, csa_ratio;
    char *bwt;
} orc_fmi;

/* lchash/lchash.h:16-20 */
typedef struct { uint64_t *lc; uint64_t len; int hlen; } orc_lch;

/* accaln.h:67-71 without the name string (names are host-side only) */
typedef struct { uint64_t offset, seq_len; } orc_mta;

/* alnmain.c:143-148 (g_name replaced by the index of the mta entry) */
typedef struct { uint64_t loc, off; int32_t seq_id; uint8_t strand; } orc_seq_meta;

typedef struct {
    orc_fmi fmi;
    orc_lch lch;
    uint64_t *sa;          /* SA values as returned by sa_access (fmidx.c:18-33) */
    uint64_t sa_len;
    char *content;         /* the .cat text, con_len bytes (+NUL) */
    uint64_t con_len;
    orc_mta *mta;
    int mta_len;
} orc_index;

/* work counters: the exact algorithmic bytes of a run */
typedef struct {
    uint64_t n_lc;        /* lc_access calls               (16 B each)            */
    uint64_t n_fmi;       /* fmi_aln calls that ran >=1 step                       */
    uint64_t n_occ;       /* _occ_access calls              (8 B o-sample each)    */
    uint64_t bwt_bytes;   /* bwt bytes scanned by _occ_access                      */
    uint64_t n_sa;        /* sa_access calls                (8 B each)             */
    uint64_t n_seeds;     /* lc_aln calls                                          */
    uint64_t n_phases;    /* seeding phases executed                               */
    uint64_t cells;       /* GACT DP cells evaluated                               */
    uint64_t tiles;       /* GACT tiles                                            */
    uint64_t read_bases;
    uint64_t cigar_ops;
} orc_counters;

/* GACT parameters (docs/GACT_SPEC.md) */
typedef struct { int T, O, W; } orc_gact_params;

/* ---- index construction (asindex.c, fmidx.c:76-198, lchash.c:52-73) ---- */
/* .cat text from nseq sequences: per record fwd + revcomp, upper-case, final '$'
 * (asindex.c:78-116).  N is fenced (returns -1 if any non-ACGT base).       */
int orc_cat_build(const char *const *seqs, const uint64_t *lens, int nseq,
                  char **cat_out, uint64_t *cat_len, orc_mta *mta_out);
/* plain comparison-sort suffix array of text[0..L) whose last byte is a
 * unique minimal '$' (what pSAscan produces for the .cat file).            */
int orc_sa_build(const char *text, uint64_t L, uint64_t *sa_out);
int orc_index_build(const char *cat, uint64_t L, const orc_mta *mta, int mta_len,
                    int o_ratio, int hlen, orc_index *out);
void orc_index_free(orc_index *idx);
/* borrow externally built arrays (reference in-memory layouts); nothing is
 * copied and orc_index_free must NOT be called on the result.              */
void orc_index_adopt(orc_index *out, const uint64_t *c256, const uint64_t *o,
                     uint64_t o_len, int o_ratio, const char *bwt, uint64_t length,
                     const uint64_t *lc, uint64_t lc_len, int hlen,
                     const uint64_t *sa, uint64_t sa_len, const char *content,
                     uint64_t con_len, const orc_mta *mta, int mta_len);

/* ---- seed side ---- */
uint64_t orc_occ_access(const orc_fmi *idx, char c, uint64_t loc, orc_counters *ct);
uint64_t orc_fmi_aln(const orc_fmi *idx, const char *qry, int len,
                     uint64_t *k, uint64_t *l, orc_counters *ct);
uint64_t orc_num_from_seq(const char *seq, int hlen);
uint64_t orc_lc_aln(const char *qry, int qlen, uint64_t *k, uint64_t *l,
                    const orc_fmi *fmi, const orc_lch *h, orc_counters *ct);
uint64_t orc_sa_access(const orc_index *idx, uint64_t loc);
uint64_t orc_csa_access(const orc_fmi *fmi, uint64_t loc);

orc_histo *orc_histo_init(uint32_t cap);
void orc_histo_destroy(orc_histo *h);
void orc_histo_add(orc_histo *h, uint64_t key);
uint64_t orc_histo_find_2_max(orc_histo *h, orc_entry *store);

/* per-seed / per-phase trace of one read (optional, for fixtures) */
typedef struct { int32_t j; uint64_t rr, k, l; } orc_seed_rec;
typedef struct { int32_t iter; orc_entry top1, top2; uint64_t v; int32_t decided; } orc_phase_rec;
typedef struct {
    orc_seed_rec *seeds; uint64_t n_seeds, cap_seeds;
    orc_phase_rec *phases; uint64_t n_phases, cap_phases;
} orc_trace;

/* PART 1 of alnmain.c:333-405 for one read.  Returns the number of phases
 * executed; *best is the candidate entry.  Reads with len <= seed_len are
 * fenced to "no seeds" (the reference's unsigned wrap at :353 is UB).      */
int orc_seed_read(const orc_index *idx, const char *read, uint32_t len,
                  uint32_t seed_len, uint32_t thres, orc_entry *best,
                  orc_trace *trace, orc_counters *ct);

/* ---- extend side ---- */
int orc_seq_lookup(const orc_mta *table, int len, uint64_t loc, uint32_t qlen,
                   orc_seq_meta *result);
void orc_rev_comp_in_place(char *seq, uint32_t len);
/* docs/GACT_SPEC.md.  ops buffer must hold n+m bytes.  Returns score
 * (edit distance, >=0) or -1.                                              */
int orc_gact(const char *q, int n, const char *d, int m, orc_gact_params gp,
             uint8_t *ops, int *n_ops, orc_counters *ct);
/* PART 2 of alnmain.c:425-451 for one read (read is rev-comped in place when
 * strand==1).  Returns meta_r.  score = -1 and n_ops = 0 when the lookup
 * fails or the window leaves the text (fenced).             */
int orc_extend_read(const orc_index *idx, char *read, uint32_t len,
                    const orc_entry *best, orc_gact_params gp,
                    uint8_t *ops, int *n_ops, int *score, orc_seq_meta *meta,
                    orc_counters *ct);
/* run-length SAM CIGAR text from op bytes; returns strlen */
int orc_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen);

/* ---- batch drivers (OpenMP over reads; the CPU baseline) ---- */
int orc_seed_batch(const orc_index *idx, const char *reads, uint64_t stride,
                   const uint32_t *lens, uint64_t n, uint32_t seed_len,
                   uint32_t thres, orc_entry *best, int32_t *phases_out,
                   orc_counters *ct, int nthreads);
int orc_extend_batch(const orc_index *idx, char *reads, uint64_t stride,
                     const uint32_t *lens, uint64_t n, const orc_entry *best,
                     orc_gact_params gp, uint8_t *store, uint64_t store_stride,
                     int32_t *n_ops, int32_t *score, orc_seq_meta *meta,
                     int32_t *meta_r, orc_counters *ct, int nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
