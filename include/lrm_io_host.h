/*
 * lrm_io_host.h -- the steps either side of the hot path, host C++ in
 * liblrm_accel.so: FASTA/FASTQ batch loader, SAM header / records, run-length CIGAR text, and the
 * whole `accaln ref.fa reads.fq` flow on the GPU path.
 *
 *   reads_load + refactor_reads_seq   accaln.c:45-58, alnmain.c:87-103   -> lrm_reader_*
 *   gen_sam_header                    alnmain.c:62-75                    -> lrm_sam_header
 *   SAM record printing               alnmain.c:485-527                  -> lrm_sam_format
 *   parse_cigar (gact submodule, source absent; PARITY UNPINNED)         -> lrm_parse_cigar
 *   single_end                        alnmain.c:277-551                  -> lrm_accaln
 */
#ifndef LRM_IO_HOST_H
#define LRM_IO_HOST_H

#include "lrm_accel.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One batch in the layout refactor_reads_seq builds: read i at seqs + i*stride, NUL padded,
 * stride = max_len + 1.  quals[i] is NULL for FASTA records. */
typedef struct lrm_read_batch {
    uint64_t n, stride;
    uint32_t max_len;
    char *seqs;
    uint32_t *lens;
    char **names;
    char **quals;
    /* storage behind names[] / quals[] when the batch came from the parallel parser (NULL: one allocation per entry),
     * and whether seqs is the caller's buffer (lrm_reader_next_into); lrm_read_batch_free looks at them */
    char *name_arena, *qual_arena;
    int seqs_borrowed;
} lrm_read_batch;

typedef struct lrm_reader lrm_reader;

/* FASTA / FASTQ, plain or gzip (zlib), multi-line records, name = header up to the first blank. */
int lrm_reader_open(lrm_reader **out, const char *path);
/* Loads up to batch_size records; returns the number loaded (0 at end of file), <0 on error
 * (-2: quality string of a different length, like kseq). */
int64_t lrm_reader_next(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out);
/* The same with the sequences written into a buffer of the caller (e.g. pinned memory from lrm_host_alloc, so that the
 * batch can be handed to lrm_map_batch_submit without a staging copy) when n * (max_len + 1) <= seq_cap; otherwise the
 * library allocates as lrm_reader_next does.  out->seqs_borrowed tells which. */
int64_t lrm_reader_next_into(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out, void *seq_buf, uint64_t seq_cap);
void lrm_read_batch_free(lrm_read_batch *b);
void lrm_reader_close(lrm_reader *r);

/* Run-length SAM CIGAR from op bytes ('=' and 'X' print as M); "*" for an empty alignment.
 * Returns the text length, <0 if buf is too small. */
int lrm_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen);

/* @SQ per sequence, @RG with ID accaln<rg_id> (the reference uses time(NULL)), @PG.
 * Returns malloc'd text (free with lrm_free). */
char *lrm_sam_header(const lrm_mta_entry *mta, int mta_len, long rg_id, uint64_t *len_out);

/* One SAM line per read, alnmain.c:500-525 field for field.  Unmapped reads (meta_r == 0 or
 * score == -1; the reference prints an uninitialised struct there) print RNAME "*", POS 0, CIGAR "*".
 * Returns malloc'd text (free with lrm_free). */
char *lrm_sam_format(const lrm_read_batch *reads, const lrm_mta_entry *mta, int mta_len,
                     const lrm_cigar *cig, const int *score, const lrm_seq_meta *meta,
                     const int *meta_r, uint64_t n, uint64_t *len_out);
void lrm_free(void *p);

/* `accaln genome reads [batch seed_len thres]` on the GPU path: loads the index files next to
 * `genome`, maps `reads_path` batch by batch, writes SAM to `sam_path`.  total/valid are the
 * reference's "Sensitivity: valid/total" counters (alnmain.c:541). */
int lrm_accaln(const char *genome, const char *reads_path, const char *sam_path, lrm_params p,
               lrm_gact_params gp, int device, long rg_id, uint64_t *total, uint64_t *valid);

#ifdef __cplusplus
}
#endif
#endif
