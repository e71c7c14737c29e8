/*
 * lrm_index_host.h -- CPU-side index construction and the on-disk formats
 * (part of liblrm_accel.so; index construction stays on the CPU by design).
 *
 * Replaces, with byte-identical outputs for N-free references:
 *   create_meta   asindex.c:78-116      -> lrm_cat_from_seqs / lrm_create_meta_fasta
 *   sa_build      psascan/sa_use.cc:8-18 -> lrm_sa_build   (own SA-IS; the SA of a text that ends
 *                                                          in a unique minimal '$' is unique)
 *   fmi_build     fmidx.c:166-198       -> lrm_host_index_build  (C, BWT, O, CSA)
 *   lc_build      lchash.c:52-73        -> lrm_host_index_build  (one pass over the SA instead of
 *                                                          4^hlen backward searches; same table)
 *   fmi_write/read fmidx.c:221-275, lc_write/read lchash.c:106-127,
 *   ui40_fread sa_use.h:31-46, load_mta alnmain.c:125-140, mstring_write/read mutils.c:53-68
 *
 * The structs are the reference's in-memory layouts (lrm_accel.h), so a host
 * index built here can be handed to lrm_index_upload() or to the reference's
 * own CPU code unchanged.
 */
#ifndef LRM_INDEX_HOST_H
#define LRM_INDEX_HOST_H

#include "lrm_accel.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrm_host_index {
    lrm_dna_fmi fmi;
    lrm_lc_hash lch;
    lrm_sa_mem sa;
    char *content;          /* .cat text, con_len bytes + NUL */
    uint64_t con_len;
    lrm_mta_entry *mta;
    int mta_len;
} lrm_host_index;

/* .cat text + .mta table from in-memory sequences (names may be NULL).
 * Bases are upper-cased; N/n is replaced by a seeded pseudo-random base
 * (the reference uses srand48(time), asindex.c:37-60,125 -- not reproducible). */
int lrm_cat_from_seqs(const char *const *names, const char *const *seqs, const uint64_t *lens,
                      int nseq, uint64_t n_seed, char **cat_out, uint64_t *cat_len,
                      lrm_mta_entry **mta_out);

/* Suffix array of text[0..L) (last byte: unique minimal '$') as ui40 in RAM. */
int lrm_sa_build(const char *text, uint64_t L, lrm_ui40 *out);

/* fmi_build + lc_build.  The text is copied into out->content. */
int lrm_host_index_build(const char *cat, uint64_t L, const lrm_mta_entry *mta, int mta_len,
                         int o_ratio, int hlen, lrm_host_index *out);
void lrm_host_index_free(lrm_host_index *idx);

/* On-disk formats (Notes.txt:6-29).  `genome` is the FASTA path the reference
 * derives its file names from: genome.mta, genome.cat, genome.cat.mfi,
 * genome.cat.lch, genome.cat.sa5. */
int lrm_host_index_write(const lrm_host_index *idx, const char *genome);
int lrm_host_index_read(const char *genome, lrm_host_index *out);

int lrm_fmi_write(const lrm_dna_fmi *fmi, const char *prefix);      /* prefix + ".mfi" */
int lrm_fmi_read(lrm_dna_fmi *fmi, const char *prefix);
int lrm_lc_write(const char *path, const lrm_lc_hash *h);
int lrm_lc_read(const char *path, lrm_lc_hash *h);
int lrm_sa5_write(const char *path, const lrm_ui40 *mem, uint64_t n);
/* returns the number of entries read (ui40_fread semantics), <0 on error */
int64_t lrm_sa5_read(const char *path, lrm_ui40 *mem, uint64_t nitems);
int lrm_mta_write(const char *path, const lrm_mta_entry *mta, int n);
int lrm_mta_read(const char *path, lrm_mta_entry **mta_out);        /* returns count, <0 on error */
void lrm_mta_free(lrm_mta_entry *mta, int n);

/* FASTA (plain text) -> genome.mta + genome.cat, then the full index files:
 * what `accidx genome` does (asindex.c:129-153). */
int lrm_accidx(const char *genome, int o_ratio, int hlen, uint64_t n_seed);

#ifdef __cplusplus
}
#endif
#endif
