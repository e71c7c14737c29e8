/*
 * lrm_accel.h -- C-ABI of liblrm_accel.so, the MI355X (gfx950) implementation
 * of the seed-and-extend hot path of lisanhu/LongReadMapper (accaln).
 *
 * Everything here is extern "C", plain pointers and sizes.  The entry points
 * are what a maintainer of the reference binds in alnmain.c instead of the
 * per-read CPU loops (see INTEGRATION.md for the patch):
 *
 *   reference (file:line, relative to the reference tree)      replaced by
 *   ---------------------------------------------------------  ------------------
 *   init(): index arrays resident in host RAM                   lrm_index_upload
 *     alnmain.c:179-256, fmidx.h:16-28, lchash.h:16-20
 *   PART 1 seed + vote loop, alnmain.c:333-405                  lrm_seed_batch
 *     lc_aln lchash.h:25-27, fmi_aln fmidx.h:37-38,
 *     sa_access fmidx.h:30, histo_* histo.h:32-37
 *   PART 2 locus resolve + rev-comp + extension,                lrm_extend_batch
 *     alnmain.c:408-451, cigar_align mutils.h:57-58
 *   PART 1 + PART 2 in one device pass (one upload per batch)   lrm_map_batch
 *   the batch loop with its planned copy clauses,                lrm_map_batch_submit / lrm_map_batch_wait
 *     alnmain.c:302-330, 420-424 (batches overlap on the device)
 *   params (run-time options), alnmain.h:10-13,                  lrm_index_options / lrm_map_options
 *     alnmain.c:574-588
 *   PART 3 result flags, alnmain.c:458-477                      lrm_result_flags
 *   context_destroy(), accaln.c:7-43                            lrm_index_free
 *   host batch loop over devices, alnmain.c:302-330             lrm_index_upload_multi (+ the same batch calls)
 *   pair_end(), alnmain.c:554-557 (unimplemented, returns -1)   lrm_pair_end
 *
 * Struct mirrors keep the reference's field order so the reference's own
 * objects can be passed by pointer cast (dna_fmi*, lc_hash*, entry*, params).
 *
 * There is NO CPU fallback: every batch call fails (negative return,
 * lrm_last_error()) when no gfx950 device / code object is available.
 *
 * Return convention: 0 = ok, <0 = error (message via lrm_last_error()).
 * Data-level conventions of the reference are preserved: 0 seed hits, score
 * -1 = alignment failure, meta_r 0 = locus outside every sequence.
 */
#ifndef LRM_ACCEL_H
#define LRM_ACCEL_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRM_ABI_VERSION 3

/* histo/histo.h:21-23 `entry` */
typedef struct lrm_entry { uint64_t key, val, bucket; } lrm_entry;

/* alnmain.h:10-13 `params` */
typedef struct lrm_params { uint64_t batch_size; uint32_t seed_len, thres; } lrm_params;

/* fmidx/fmidx.h:16-21 `dna_fmi` (identical field order) */
typedef struct lrm_dna_fmi {
    uint64_t length, o_len, csa_len;
    uint64_t *c, *o, *csa;
    int o_ratio, csa_ratio;
    char *bwt;
} lrm_dna_fmi;

/* lchash/lchash.h:16-20 `lc_hash` */
typedef struct lrm_lc_hash { uint64_t *lc; uint64_t len; int hlen; } lrm_lc_hash;

/* psascan/sa_use.h:17-20 `ui40_t` as it sits in RAM (8 bytes with padding) and
 * fmidx/fmidx.h:23-26 `sa_mem` */
typedef struct lrm_ui40 { uint32_t low; uint8_t high; } lrm_ui40;
typedef struct lrm_sa_mem { uint64_t start, len; lrm_ui40 *mem; } lrm_sa_mem;

/* accaln.h:67-71 `mta_entry` flattened: {mstring seq_name{l,s,own}; offset; seq_len} */
typedef struct lrm_mta_entry {
    uint64_t name_len; char *name; int name_own;
    uint64_t offset; size_t seq_len;
} lrm_mta_entry;

/* alnmain.c:143-148 `seq_meta`; g_name is returned as the index of the mta entry */
typedef struct lrm_seq_meta { uint64_t loc, off; int32_t seq_id; uint8_t strand; } lrm_seq_meta;

/* gact `cigar` as used at mutils.c:97-103 / alnmain.c:314-325: one op byte per
 * alignment column ('=' 'X' 'I' 'D') in a caller-owned buffer */
typedef struct lrm_cigar { uint8_t *cigar; int n_cigar_op; int score; } lrm_cigar;

/* GACT tile / overlap / band (docs/GACT_SPEC.md); {0,0,0} selects the defaults */
typedef struct lrm_gact_params { int T, O, W; } lrm_gact_params;
#define LRM_GACT_T_DEFAULT 320
#define LRM_GACT_O_DEFAULT 120
#define LRM_GACT_W_DEFAULT 128

typedef struct lrm_index lrm_index;       /* opaque: device-resident index */

/* ---------------------------------------------------------------------------
 * Run-time options (the reference's run-time surface is `params`, alnmain.h:10-13, filled from argv at
 * alnmain.c:574-588; what is a choice of THIS implementation lives in the two structs below).
 * Initialise with lrm_*_options_init (sets struct_size and the automatic choices), change fields, pass by pointer;
 * NULL means "all automatic".  LRM_* environment variables remain as OVERRIDES for tuning sessions only: they are
 * read once, when a handle is created, never per call.
 * ------------------------------------------------------------------------- */
typedef struct lrm_index_options {
    uint32_t struct_size;      /* sizeof(lrm_index_options) of the caller's header */
    int32_t sa_sampled;        /* 0 / 1: the full suffix array (sa_access, fmidx.c:18-33); r = 2..64, a power of two: only
                                  rows i*r stay on the device -- r = 4 is the reference's csa table, fmidx.c:153-163 -- and the
                                  kernels locate the other rows by LF steps (csa_access, fmidx.c:315-331): GRCh38 49.6 -> 12.4 GB */
    int32_t lc_long;           /* k-mer length of the long seed table derived on the device: -1 automatic (the longest that
                                  leaves room in HBM), 0 none (lchash alone, lchash.c:89-104), 13..17 */
    int32_t lc_long_max;       /* cap on the automatic choice (0: none): the 64 GiB 16-mer table costs 0.7-2 s at upload, a
                                  caller that knows its run is short says 15 */
    int32_t lc_pair;           /* layout of the long table: -1 automatic (pair-line), 0 plain, 1 pair-line */
    uint32_t lcx_threshold;    /* 0: default (2^24 - 1); lchash intervals of at least this many rows go through the side
                                  table (tests) */
    uint32_t lc_entry_bytes;   /* entries of the long table: 0 automatic, 8, or 5 (pair-line layout only: 40 bytes per
                                  (k-1)-mer instead of 64 -- pair-line 17-mers of a GRCh38-sized text in 160 GiB; counts that
                                  do not fit the 40 - ceil(log2(rows)) count bits go through a side hash table) */
    int32_t lc_core;           /* core table (texts of up to 2^25 rows, next to pair-line 16-mers): one 64-byte line per 13-mer holds
                                  the entries of the 16-mers around it, the lookups of FOUR neighbouring read positions share it
                                  (4 GiB): -1 automatic, 0 off, 1 on */
    uint32_t lc_count_bits;    /* tests: count bits of the 5-byte entries (0: 40 - ceil(log2(rows))); a small value sends
                                  ordinary repeats through the side hash table */
    int32_t seed_table;        /* SEED table: a hash table of the text's seed_table_len-mers -- (first row, count) of every distinct one,
                                  the exact result of lc_aln + fmi_aln (lchash.c:89-104, fmidx.c:295-313) for it -- in 64-byte lines
                                  that the seeds of 4 (or 2) neighbouring read positions share: a seed of that length costs a quarter
                                  (half) of a memory line and no backward step; a 20-mer that occurs once carries its text position
                                  (sa_access of its row, fmidx.c:18-33), so the vote stage gathers nothing for it.  -1 automatic (pure ACGT texts, when HBM allows: 2 GiB
                                  for an E. coli-sized text, 64 GiB chr1-sized, 128 GiB GRCh38-sized), 0 off, 1 on.  Seeds of any
                                  other length go through the tables above. */
    uint32_t seed_table_len;   /* seed length the table is built for: 0 = 20, the reference's default (alnmain.c:577-580); 16..24 */
    uint32_t seed_table_share; /* read positions per line: 0 automatic, 4 (8-byte slots, eight per line) or 2 (6-byte slots, ten) */
    uint32_t seed_table_bits;  /* tests: log2 of the number of lines (0: automatic) -- few lines force the overflow paths */
    uint32_t seed_table_count_bits; /* tests: count bits of a slot (0: all that are left) -- few send repeats to the side table */
    uint32_t reserved[2];
} lrm_index_options;
void lrm_index_options_init(lrm_index_options *o);

typedef struct lrm_map_options {
    uint32_t struct_size;      /* sizeof(lrm_map_options) of the caller's header */
    int32_t dense_results;     /* 0: cig_out[i].cigar = store_mem + i*store_stride (alnmain.c:322-325).
                                  1: DENSE -- cig_out[i].cigar = store_mem + off[i], the used op bytes of consecutive reads
                                  packed back to back (16-byte aligned) inside the region of store_mem their rows would
                                  occupy.  The convention of mutils.c:97-103 is kept (caller-owned buffer, callee-set
                                  pointer); with store_mem pinned the device image of the op bytes is DMA'd straight into
                                  it and the library spends no host CPU on the results.  Needs store_stride % 16 == 0. */
    int32_t gact_impl;         /* extension kernel: 0 automatic, 1 one read per wavefront, 3 two reads per wavefront,
                                  4 bit-sliced lane per read whenever it applies */
    int32_t seed_rounds;       /* 0 automatic (phase 0 first unless the previous batch decided < 2 % there), 1, 2 */
    int32_t vote_exact_only;   /* 1: every (read, phase) item goes through the exact vote kernel (the fast kernel in front of it
                                  is skipped); results are identical either way -- tests and A/B timing */
    uint32_t slice_reads;      /* host pipeline shape, 0 = automatic: reads per device pass, */
    uint32_t sub_batches;      /*   seed sub-batches per pass, */
    uint32_t group_subs;       /*   sub-batches per extension group */
    uint32_t bs_waves;         /* tests: cap on the resident wavefronts of the bit-sliced kernel (forces lane refills) */
    uint32_t cigar_text;       /* 1 (implies dense_results): cig_out[i].cigar points to the NUL-terminated run-length CIGAR TEXT
                                  parse_cigar would print from the op bytes (alnmain.c:497-498; '=' and 'X' columns as M, "*"
                                  for a read without an alignment) instead of the op bytes themselves -- the run-length pass runs
                                  on the device and ~0.45 instead of 1.1 bytes per read base cross the link.  n_cigar_op stays
                                  the number of alignment columns. */
    uint32_t copy_threads;     /* memcpy team of the PAGEABLE paths (upload staging, result placement): 0 automatic (the host's CPU
                                  share / replicas, at most 8 -- what 24 Gbp/s through pageable buffers needs), else 1..16; a caller
                                  whose own threads need the cores (lrm_accaln's parser and formatter) says 1 or 2 */
    uint32_t keep_reads;       /* 1: reads_buf is left exactly as the caller gave it -- the reverse-complemented copies of the
                                  reverse-strand reads (alnmain.c:437 does that in place, before the extension) stay on the device,
                                  0.5 bytes per read base less cross the link and nothing is placed on the host.  A caller that
                                  prints such a read (SAM SEQ) applies meta_out[i].strand itself: the read was reverse-complemented
                                  iff meta_r[i] != 0 && meta_out[i].strand == 1 (lrm_accaln's formatter does).  Host-buffer
                                  batch calls only. */
    uint32_t reserved[7];
} lrm_map_options;
void lrm_map_options_init(lrm_map_options *o);

const char *lrm_last_error(void);
int lrm_abi_version(void);
/* number of visible HIP devices (<=0: none). Does not fail loudly: probing only. */
int lrm_device_count(void);

/* ---------------------------------------------------------------------------
 * Index: host arrays in the reference's in-memory layout -> device image
 * ------------------------------------------------------------------------- */

/* Size in bytes of the device image ("blob") for an index of these dimensions. */
uint64_t lrm_index_blob_bytes(uint64_t length, int hlen, int mta_len);

/* Serialise the reference-layout arrays into the device image, in host memory
 * (blob must hold lrm_index_blob_bytes()).  Pure CPU; usable without a GPU.
 * sa: sa_len elements of ui40 in RAM (8-byte stride), as filled by ui40_fread. */
int lrm_index_pack_blob(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch,
                        const lrm_sa_mem *sa, const char *content, uint64_t con_len,
                        const lrm_mta_entry *mta, int mta_len, void *blob, uint64_t blob_bytes);

/* One-call upload used behind alnmain.c:init(): the image is packed piece by piece into pinned chunks whose
 * DMA overlaps the packing of the next piece -- no host copy of the image. */
int lrm_index_upload(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch,
                     const lrm_sa_mem *sa, const char *content, uint64_t con_len,
                     const lrm_mta_entry *mta, int mta_len, int device);

/* pack straight into device memory the caller owns (blob_bytes >= lrm_index_blob_bytes()): no host copy of the
 * image; the buffer can then be broadcast and adopted (lrm_index_adopt_device) on every rank. */
int lrm_index_pack_device(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                          const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                          void *d_blob, uint64_t blob_bytes, int device);

/* Multi-GPU group: the image is packed once, uploaded to devices[0] and replicated to the other devices over
 * xGMI (one RCCL broadcast; hipMemcpyPeer if librccl cannot be loaded).  devices == NULL means 0..ngpus-1.  The
 * returned handle is used with the same batch calls: lrm_seed_batch / lrm_extend_batch / lrm_map_batch split
 * the batch into ngpus contiguous slices balanced by bases and run one host thread per device, every device
 * writing its slice of the caller's arrays in place (results do not depend on ngpus).  A device may be listed
 * more than once (logical replicas on one GPU).  The *_dev entry points address one replica:
 * lrm_index_replica(). */
int lrm_index_upload_multi(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch,
                           const lrm_sa_mem *sa, const char *content, uint64_t con_len,
                           const lrm_mta_entry *mta, int mta_len, const int *devices, int ngpus);
int lrm_index_replicas(const lrm_index *idx);
lrm_index *lrm_index_replica(lrm_index *idx, int r);     /* borrowed; replica 0 is the handle itself */

/* The same entry points with options (NULL = automatic).  lrm_index_upload_opt covers one device (ngpus == 1,
 * devices[0]; devices == NULL means device 0..ngpus-1) and the multi-GPU group. */
uint64_t lrm_index_blob_bytes_opt(uint64_t length, int hlen, int mta_len, const lrm_index_options *opt);
int lrm_index_pack_blob_opt(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa, const char *content,
                            uint64_t con_len, const lrm_mta_entry *mta, int mta_len, void *blob, uint64_t blob_bytes,
                            const lrm_index_options *opt);
int lrm_index_pack_device_opt(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa, const char *content,
                              uint64_t con_len, const lrm_mta_entry *mta, int mta_len, void *d_blob, uint64_t blob_bytes,
                              int device, const lrm_index_options *opt);
int lrm_index_upload_opt(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                         const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                         const int *devices, int ngpus, const lrm_index_options *opt);
int lrm_index_adopt_device_opt(lrm_index **out, void *d_blob, uint64_t blob_bytes, int device, const lrm_index_options *opt);
int lrm_index_upload_blob_opt(lrm_index **out, const void *blob, uint64_t blob_bytes, int device, const lrm_index_options *opt);

/* Default options of the batch calls on this handle (every replica of a group): lrm_seed_batch, lrm_extend_batch,
 * lrm_map_batch and the *_dev calls use them; lrm_map_batch_submit takes its own.  NULL restores the automatic choices. */
int lrm_index_set_map_options(lrm_index *idx, const lrm_map_options *opt);

/* Which derived seed tables a handle (replica 0 of a group) ended up with -- the automatic choices depend on the text
 * and on the HBM that was free at upload.  Results never depend on them. */
typedef struct lrm_index_tables {
    int32_t lc_long;             /* k-mer length of the long table (0: none) */
    int32_t lc_pair;             /* pair-line layout */
    int32_t lc_entry_bytes;      /* 8 or 5 */
    int32_t lc_core;             /* core table present */
    int32_t seed_table_len;      /* seed length of the seed table (0: none) */
    int32_t seed_table_share;    /* read positions per line: 4 or 2 */
    int32_t seed_table_bits;     /* log2 of its lines (64 bytes each) */
    int32_t seed_table_slot_bytes;
    int32_t seed_table_count_bits;
    int32_t reserved0;
    uint64_t seed_table_side_entries;   /* entries of crowded lines / counts beyond the slot's bits, in the side hash table */
    uint64_t derived_bytes;      /* HBM held by all derived tables (long table, core table, seed table, side tables) */
    uint64_t reserved[4];
} lrm_index_tables;
int lrm_index_get_tables(const lrm_index *idx, lrm_index_tables *out);

/* Adopt a blob that already sits in device memory (e.g. the destination of an
 * RCCL broadcast).  The blob is borrowed: it must outlive the handle. */
int lrm_index_adopt_device(lrm_index **out, void *d_blob, uint64_t blob_bytes, int device);

/* Upload a host blob (copy). */
int lrm_index_upload_blob(lrm_index **out, const void *blob, uint64_t blob_bytes, int device);

void lrm_index_free(lrm_index *idx);

/* ---------------------------------------------------------------------------
 * Batch entry points with HOST buffers (the drop-in boundary)
 * ------------------------------------------------------------------------- */

/* PART 1.  reads_buf: dense buffer, read i at reads_buf + i*stride, lens[i]
 * bases, upper-case ACGT (alnmain.c:87-103 layout, stride = max_read_len+1).
 * best_out[i] = the candidate entry the reference leaves in best[chunk_i]. */
int lrm_seed_batch(lrm_index *idx, const char *reads_buf, uint64_t stride,
                   const uint32_t *lens, uint64_t n, lrm_params p, lrm_entry *best_out);

/* PART 2.  Reads whose locus resolves to the reverse strand are reverse-
 * complemented IN PLACE in reads_buf (alnmain.c:433-438).  cig_out[i].cigar is
 * set to store_mem + i*store_stride (store_stride >= 2*lens[i]);
 * score_out[i] = cig_out[i].score (the value the reference writes to limit[]). */
int lrm_extend_batch(lrm_index *idx, char *reads_buf, uint64_t stride,
                     const uint32_t *lens, uint64_t n, const lrm_entry *best,
                     lrm_gact_params gp, lrm_cigar *cig_out, uint8_t *store_mem,
                     uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out,
                     int *meta_r_out);

/* PART 1 + PART 2 for one batch in a single device pass: the reads cross the link once, best[] stays on the
 * device between the two parts.  Same outputs as lrm_seed_batch followed by lrm_extend_batch. */
int lrm_map_batch(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                  lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out,
                  uint8_t *store_mem, uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out,
                  int *meta_r_out);

/* The same, asynchronous: lrm_map_batch_submit queues the batch and returns; lrm_map_batch_wait blocks until its
 * results are in the caller's arrays, returns the batch's status and frees the ticket.  Up to TWO batches per device
 * are in flight on the device at once -- the upload and the seeds of batch k+1 run under the extension tail and the
 * result download of batch k, which is the chain that bounds a single call (alnmain.c:302-330 with the copy clauses
 * planned at :420-424) -- further submissions queue.  Every argument array must stay valid, and untouched by the
 * caller, until the wait returns; batches complete in submission order.  lrm_map_batch == submit + wait.
 * opt == NULL: the handle's default options. */
typedef struct lrm_ticket lrm_ticket;
int lrm_map_batch_submit(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                         lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out,
                         uint8_t *store_mem, uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out,
                         int *meta_r_out, const lrm_map_options *opt, lrm_ticket **ticket_out);
int lrm_map_batch_wait(lrm_ticket *ticket);

/* Pinned host memory for the batch buffers (reads_buf, store_mem): the DMA engines then read / write the
 * caller's memory directly; pageable buffers work too and are staged through pinned chunks.
 * lrm_host_register pins memory the caller already owns (what alnmain.c:297-320 mallocs). */
void *lrm_host_alloc(uint64_t bytes);
void lrm_host_free(void *p);
int lrm_host_register(void *p, uint64_t bytes);
int lrm_host_unregister(void *p);

/* alnmain.c:554-557 `pair_end`: declared, unimplemented ("todo") and returning -1 in the reference; kept so
 * that the call surface is complete. */
int lrm_pair_end(int argc, const char *argv[]);

/* PART 3 flag/mapq/valid assembly (alnmain.c:460-474); pure host arithmetic. */
void lrm_result_flags(const int *score, const int *meta_r, const lrm_seq_meta *meta,
                      uint64_t n, int *flag_out, int *mapq_out, int *valid_out);

/* ---------------------------------------------------------------------------
 * Batch entry points with DEVICE buffers (inputs/outputs resident in HBM;
 * asynchronous on `stream`, a hipStream_t passed as void*, may be NULL).
 * ------------------------------------------------------------------------- */

typedef struct lrm_workspace lrm_workspace;   /* opaque: device scratch for a batch shape */

/* Scratch for batches of up to n_max reads of up to max_len bases.
 * Preconditions of the *_dev calls (not checked on the device): d_lens[i] <= max_len <= stride, and
 * store_stride >= 2*max_len.  A kernel-side error of a batch (vote table overflow) is sticky: the NEXT *_dev
 * call on the workspace, or lrm_workspace_stats, returns -2 and clears it. */
int lrm_workspace_create(lrm_workspace **out, lrm_index *idx, uint64_t n_max,
                         uint32_t max_len, uint32_t seed_len, uint32_t thres);
void lrm_workspace_free(lrm_workspace *ws);
uint64_t lrm_workspace_bytes(const lrm_workspace *ws);

int lrm_seed_batch_dev(lrm_index *idx, lrm_workspace *ws, const char *d_reads,
                       uint64_t stride, const uint32_t *d_lens, uint64_t n,
                       uint32_t max_len, lrm_params p, lrm_entry *d_best, void *stream);

int lrm_extend_batch_dev(lrm_index *idx, lrm_workspace *ws, char *d_reads, uint64_t stride,
                         const uint32_t *d_lens, uint64_t n, uint32_t max_len,
                         const lrm_entry *d_best, lrm_gact_params gp, uint8_t *d_store,
                         uint64_t store_stride, int32_t *d_n_ops, int32_t *d_score,
                         lrm_seq_meta *d_meta, int32_t *d_meta_r, void *stream);

/* Counters of the last *_dev call on this workspace (device->host copy, syncs
 * the stream): vote items per table tier, reads decided in
 * phase 0, GACT tiles.  For tests / bench bookkeeping only. */
typedef struct lrm_stats {
    uint64_t vote_tier2_items;      /* (read,phase) items voted by a whole workgroup in one pass (192 < hits <= 1152) */
    uint64_t vote_tier3_items;      /* ... in several passes over the workgroup table (hits > 1152) */
    uint64_t reads_decided_phase0;  /* reads whose vote passed 0.6 in phase 0 */
    uint64_t gact_tiles;
    /* with lrm_workspace_set_counting(ws, 1): the memory requests the DEVICE layout made in the last seed call */
    uint64_t seeds_evaluated;       /* seed positions searched */
    uint64_t seed_table_lookups;    /* 8-byte entries of the seed tables (lchash image or long table) read */
    uint64_t seed_rank_requests;    /* 16-byte {prefix, mask} pairs of the occ blocks read (1 or 2 per backward step) */
    uint64_t vote_redo_items;       /* (read, phase) items the fast vote kernel left to the exact one */
} lrm_stats;
/* Counting build of the seed kernel for the NEXT calls on this workspace (bench bookkeeping: slower, never timed). */
int lrm_workspace_set_counting(lrm_workspace *ws, int enable);
int lrm_workspace_stats(lrm_workspace *ws, lrm_stats *out, void *stream);

/* Per-kernel timing with HIP events recorded on the launch stream (bench bookkeeping).
 * Kernel order: pack2bit, seed_search, vote, decide, locus_resolve, revcomp, gact (byte kernels),
 * bs_pack_reads, gact_bs (bit-sliced kernel + expansion).
 * lrm_workspace_timing synchronises the stream, ADDS the elapsed milliseconds and launch counts
 * of everything recorded since the last call into ms[LRM_N_KERNELS] / launches[LRM_N_KERNELS], and
 * resets the record. */
#define LRM_N_KERNELS 9
int lrm_workspace_set_timing(lrm_workspace *ws, int enable);
int lrm_workspace_timing(lrm_workspace *ws, double *ms, uint64_t *launches, void *stream);
const char *lrm_kernel_name(int kernel);

/* Debug/parity taps (tests only): per-seed search results of one read as the
 * seed kernel produced them, in (phase, ordinal) order: j, rr, k, l. */
int lrm_debug_seed_search(lrm_index *idx, const char *read, uint32_t len, uint32_t seed_len,
                          uint32_t thres, int32_t *j_out, uint64_t *rr_out, uint64_t *k_out,
                          uint64_t *l_out, uint64_t cap, uint64_t *n_out);

/* Test-only: force the multi-pass vote tier of this handle's batches into overflow (a pass limit above the table size
 * and a small table); 0, 0 restores the defaults. */
int lrm_debug_set_vote_limits(lrm_index *idx, uint32_t t3_limit, uint32_t t3_slots);

/* Tuning sessions only (tools/): re-read the LRM_* overrides for the batch calls of this handle (normally they are
 * read once, when the handle is created). */
int lrm_debug_reload_env(lrm_index *idx);

/* RCCL self-test (tests only): dlopen + ncclCommInitAll + a 1-rank grouped ncclBroadcast of `bytes` bytes on
 * `device`.  0 ok, 1 librccl not loadable (lrm_index_upload_multi then uses hipMemcpyPeer), <0 error. */
int lrm_debug_rccl_selftest(int device, uint64_t bytes);

/* Direct kernel tap (tests only): simple_gact on one (q, d) pair. */
int lrm_debug_gact(const char *q, int n, const char *d, int m, lrm_gact_params gp,
                   uint8_t *ops, int *n_ops, int *score, int device);
/* ... with the kernel chosen by the caller (lrm_map_options.gact_impl values) */
int lrm_debug_gact_impl(const char *q, int n, const char *d, int m, lrm_gact_params gp, int impl,
                        uint8_t *ops, int *n_ops, int *score, int device);

#ifdef __cplusplus
}
#endif
#endif /* LRM_ACCEL_H */
