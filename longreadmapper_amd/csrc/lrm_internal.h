// lrm_internal.h -- device image layout + internal declarations of liblrm_accel.so
//
// Device image ("blob"), designed for gfx950 gathers rather than copied from the
// reference's in-memory arrays:
//
//   [BlobHeader 256 B]
//   [occ blocks]   one 64-byte block per 64 BWT rows: for each symbol A,C,G,T a 16-byte pair
//                    u64 cnt    # of the symbol in bwt[0 .. 64*blk)       ('$' not counted)
//                    u64 mask   bit r set <=> bwt[64*blk + r] is the symbol
//                  -> one rank query == ONE aligned 16-B gather + one 64-bit popcount (the
//                     reference layout, fmidx.c:277-293, needs an 8-B o[] read plus a <=32-B
//                     byte-compare scan of bwt[] in a different array).  The '$' row sets no
//                     mask bit, so it needs no special case.
//   [lc table]     4^hlen entries of 8 bytes, k | (l-k+1) << 40 (0 = absent), indexed by the
//                  LSB-first 2-bit code of the hlen-mer (base i at bits 2i..2i+1) so the kernel
//                  extracts the index as one bit field of the packed read.  Half the footprint of
//                  the reference's {u64 k, u64 l} pairs (lchash.c:12-16): 128 MiB for hlen 12, which
//                  stays resident in the 256 MiB Infinity Cache.  Intervals of >= 2^24-1 rows are
//                  marked 0xFFFFFF and resolved through a small sorted side table [lcx].
//                  (reference order: first base most significant, lchash.c:36-49; the packer permutes.)
//   [sa]           u64 per row (values of sa_access, fmidx.c:18-33); with LRM_SA_SAMPLED=r only rows i*r
//                  (csa, fmidx.c:153-163) and the kernels locate the others by LF steps (csa_access, fmidx.c:315-331)
//   [content]      the .cat text, 1 byte per base (GACT target side)
//   [mta]          {u64 offset, u64 seq_len} per sequence (accaln.h:67-71 without names)
//
// All sections are 256-byte aligned.
#pragma once
#include <stdint.h>
#include "../../include/lrm_accel.h"

#define LRM_BLOB_MAGIC 0x4c524d424c4f4232ull   // "LRMBLOB2" (2: C[] folded into the occ prefixes)
#define LRM_OCC_ROWS 64

// One LF step = ONE 16-byte gather + one 64-bit popcount: per 64 BWT rows and per symbol a
// {C[sym] + rank prefix, occurrence bitmask} pair; the four symbols of a block share a 64-byte line.
struct LrmOccEntry {
    uint64_t cnt;       // C[sym] + # of this symbol in bwt[0 .. 64*blk): an LF step is cnt + popcount(mask bits 0..r)
    uint64_t mask;      // bit r set <=> bwt[64*blk + r] is this symbol ('$' sets no bit)
};
struct LrmOccBlock { LrmOccEntry sym[4]; };
static_assert(sizeof(LrmOccBlock) == 64, "occ block must be one 64-B line");

struct LrmMtaDev { uint64_t offset, seq_len; };

struct LrmBlobHeader {
    uint64_t magic, version;
    uint64_t length;        // L = bwt rows = text bytes
    uint64_t c4[4];         // C[] for A,C,G,T  (fmidx.c:101-125)
    uint64_t dollar_row;    // row whose bwt char is '$'
    uint64_t n_blocks;
    uint64_t lc_entries;    // 4^hlen
    uint64_t sa_len, con_len;
    uint64_t off_occ, off_lc, off_sa, off_content, off_mta;
    uint64_t total_bytes;
    int32_t hlen, mta_len;
    uint64_t off_lcx, n_lcx;   // side table of {code, k, l} for intervals too long for 24 bits
    uint64_t sa_ratio;         // 0/1: [sa] holds every row; r > 1: rows i*r only (the reference's csa, fmidx.c:153-163)
    uint64_t reserved[10];
};
static_assert(sizeof(LrmBlobHeader) == 256, "header is 256 B");

// What kernels receive by value.
struct LrmIndexView {
    const LrmOccBlock *occ;
    const uint64_t *lc;       // 8-byte entries
    const uint64_t *lcx;      // {code, k, l} triples, sorted by code
    uint64_t n_lcx;
    const uint64_t *sa;
    const char *content;
    const LrmMtaDev *mta;
    uint64_t length, dollar_row, sa_len, con_len;
    uint64_t c4[4];
    int32_t hlen, mta_len;
    const uint64_t *lcl;      // optional LONG table (hl-mers, hl > hlen), built on the device from lc + FM steps; null = unused
    int32_t hl;
    int32_t sa_shift;         // log2 of the SA sampling ratio (0: every row is stored)
    int32_t lcl_pair;         // long table in PAIR-LINE layout (seed_kernels.hip): the lookups of two neighbouring seeds share a 64-byte line
    int32_t lcl_kbits;        // 0: 8-byte entries (k | count << 40).  > 0: 5-BYTE entries, k in the low lcl_kbits bits, count above
                              // (all ones = look the hl-mer up in the side hash table lclx)
    const uint64_t *core;     // optional CORE table of small texts (seed_kernels.hip): one 64-byte line per 13-mer holds the entries of
                              // the 16-mers around it, so that the lookups of FOUR neighbouring read positions share a line; null = unused
    const uint64_t *lclx;     // side hash table of the 5-byte layout: {hl-mer code + 1, k | count << 40} pairs, open addressing
    uint64_t lclx_mask;       // slots - 1 (a power of two)
    // optional SEED table (seed_kernels.hip): (first row, count) of every distinct sd_len-mer of the text in 64-byte lines shared
    // by the seeds of sd_f neighbouring read positions; null = unused
    const uint64_t *sd;
    const uint64_t *sdx;      // its side hash table {code + 1, k | count << 40}: entries of crowded lines, counts beyond the slot's bits
    uint64_t sdx_mask;
    int32_t sd_len;           // seed length it holds
    int32_t sd_f;             // read positions per line: 4 or 2
    int32_t sd_bits;          // log2 lines
    int32_t sd_kbits;         // bits of k in a slot
    int32_t sd_slot;          // slot bytes: 8 (eight per line, overflow flag in slot 0) or 6 (ten per line, entry count in the last 4 bytes)
    int32_t sd_cbits;         // count bits of a slot (all ones: the side table has the count)
};

// ---- resolved choices of a handle ------------------------------------------------------------------------------
// Options come from the caller (lrm_index_options / lrm_map_options); LRM_* environment variables are tuning
// OVERRIDES, read ONCE when a handle is created (LrmEnv) and never per call or per launch.
struct LrmEnv {                     // the LRM_* variables as they stood when the handle was created
    static constexpr int MAXV = 32;
    const char *name[MAXV];
    long long val[MAXV];
    int n;
    bool get(const char *key, long long *out) const;
};
void lrm_env_snapshot(LrmEnv *e);
struct LrmIndexTune {
    int sa_ratio;                   // 1: full SA; 2..64: sampled
    int lc_long, lc_long_max, lc_pair, lc_entry_bytes, lc_count_bits, lc_core;
    int sd, sd_len, sd_f, sd_bits, sd_cbits;      // seed table: -1 / 0 / 1, seed length, positions per line (0 automatic), tests: lines, count bits
    uint64_t lcx_threshold;
};
struct LrmMapTune {
    int dense, gact_impl, seed_rounds, cigar_text, keep_reads;
    uint32_t slice_reads, sub_batches, group_subs, bs_waves, copy_threads;
    uint32_t ss_items, ss_lds_pad, vote_vg, vote_t1, vote_u, vote_load, vote_fast;      // kernel tuning (environment only; measured defaults)
    uint32_t t3_limit, t3_slots;                                 // lrm_debug_set_vote_limits (tests)
    int ext_streams, seed_streams, verbose;
};
void lrm_resolve_index_tune(const lrm_index_options *opt, const LrmEnv &env, LrmIndexTune *out);
void lrm_resolve_map_tune(const lrm_map_options *opt, const LrmEnv &env, LrmMapTune *out);

struct LrmHostCtx;            // lrm_host.hip: per-handle state of the host-buffer entry points
struct lrm_index {
    void *d_blob;
    uint64_t blob_bytes;
    int owns_blob;
    int device;
    LrmBlobHeader hdr;
    LrmIndexView view;
    uint64_t *d_lcl;          // long lc table (owned; may be null)
    uint64_t *d_lclx;         // its side hash table in the 5-byte layout (owned; may be null)
    uint64_t *d_core;         // core table (owned; may be null)
    uint64_t *d_sd, *d_sdx;   // seed table and its side hash table (owned; may be null)
    uint64_t sd_side_entries; // entries of the side table (stats)
    uint64_t *d_cpl;          // planar 2-bit copy of the text for the bit-sliced GACT kernel (owned; may be null)
    int cpl_ok;               // text is pure ACGT (otherwise the byte kernels are used)
    uint64_t *d_sas;          // sampled-SA locate mode (csa_access, fmidx.c:315-331): SA rows i*sa_ratio (owned; may be null)
    LrmHostCtx *host;         // workspace, device mirrors, pinned staging and streams of the host-buffer calls (owned, lazy)
    // multi-GPU group (lrm_index_upload_multi): replica r lives on its own device; peers[0] == this.  A batch call
    // on the group handle partitions the reads by bases and runs one host thread per replica.
    int n_peers;
    lrm_index **peers;
    LrmEnv env;               // LRM_* overrides as read at creation
    LrmIndexTune itune;
    LrmMapTune mtune;         // default options of the batch calls on this handle (lrm_index_set_map_options)
    uint32_t dbg_t3_limit, dbg_t3_slots;   // lrm_debug_set_vote_limits (0 = default)
};

// Per-(read,phase) vote result written by the vote kernels.
struct LrmPhaseRes {
    uint64_t key1, val1, bucket1;
    uint64_t key2, val2, bucket2;
};

// Device counters block (one per workspace).
struct LrmDevCounters {
    unsigned long long reserved[8];
    unsigned long long decided_phase0;
    unsigned long long gact_tiles;
    unsigned long long pad[2];
    unsigned long long vote_fast_ticket[2];   // per seeding round: ticket of vote_fast_kernel,
    unsigned long long vote_redo_n[2];        //   items it left to the exact kernel
    unsigned long long vote_big_n[2];         //   items it left to its workgroup form, and that kernel's ticket
    unsigned long long vote_big_ticket[2];
};
// Error word of a workspace: ONE dword of host-coherent pinned memory that kernels set with a plain store (bit 0:
// vote table overflow in the multi-pass tier).  It is never cleared by a launch, so an error raised by any
// sub-batch survives until the host reads it: lrm_workspace_stats and every *_dev entry point check it (the
// next call after the faulty batch fails), the host-buffer entry points check it per sub-batch.
#define LRM_ERR_VOTE_OVERFLOW 1u
// vote tiers (seed_kernels.hip): hits per (read, phase) item up to which one wavefront / one workgroup pass suffices
#define LRM_VOTE_T1_LIMIT 192
#ifndef LRM_VOTE_T3_SLOTS
#define LRM_VOTE_T3_SLOTS 1024
#endif
#define LRM_VOTE_T3_LIMIT (LRM_VOTE_T3_SLOTS * 3 / 4)
#define LRM_VOTE_GRID 1536        // resident workgroups of the vote kernel (6 per CU)
#define LRM_VOTE_FAST_GRID 1280   // ... of the fast vote kernel (5 per CU)
#define LRM_VOTE_KC_CAP 16384     // hits per workgroup whose keys the multi-pass items keep between passes (12 B each)

enum LrmKernelId { LRM_K_PACK2BIT = 0, LRM_K_SEED_SEARCH, LRM_K_VOTE, LRM_K_DECIDE,
                   LRM_K_LOCUS, LRM_K_REVCOMP, LRM_K_GACT, LRM_K_PACK_PLANAR, LRM_K_GACT_BS,
                   LRM_K_COUNT };
#define LRM_MAX_TIMED 4096

#define LRM_WS_SEED 1
#define LRM_WS_EXTEND 2
struct lrm_workspace {
    lrm_index *idx;
    int parts;               // LRM_WS_SEED | LRM_WS_EXTEND: which scratch this workspace owns
    // optional per-kernel timing (HIP events recorded on the launch stream)
    int timing;
    int counting;            // seed_search runs its counting build (lrm_workspace_set_counting)
    int n_timed;
    void *ev_start[LRM_MAX_TIMED], *ev_stop[LRM_MAX_TIMED];
    int ev_kernel[LRM_MAX_TIMED];
    int device;
    uint64_t n_max;
    uint64_t n_last;         // reads in the last seed call (for stats)
    uint32_t max_len, seed_len, thres;
    uint32_t P;              // seed_len + 1 phases
    uint32_t cap_q;          // seeds per phase capacity
    uint64_t words_per_read; // u64 words of the 2-bit packed read (+1 guard)
    uint64_t bytes;
    // device buffers
    uint64_t *d_reads2;      // packed reads
    uint64_t *d_rec;         // survivor records k | rr << 40, compact per (read, phase): n_max * P * cap_q capacity
    uint32_t *d_recq;        // seed ordinal q of every survivor record
    uint32_t *d_cnt;         // survivors per (read, phase)
    uint64_t *d_kc_key;      // vote kernel: per-workgroup scratch of the keys of multi-pass items (LRM_VOTE_GRID x LRM_VOTE_KC_CAP)
    uint32_t *d_kc_ord;      //              ... and their order keys
    uint64_t *d_redo;        // items the fast vote kernels left to the exact one (n_max * P)
    uint64_t *d_big;         // items the wavefront form left to the workgroup form (n_max * P)
    uint64_t *d_gtab;        // pool of global-memory vote tables for items beyond the key scratch: g_slices x {key[g_slots], cf[g_slots]}
    uint32_t *d_glock;       // one lock word per slice
    uint32_t g_slices, g_slots;
    LrmPhaseRes *d_phase;    // n_max * P
    uint8_t *d_decided;      // n_max
    uint32_t *d_hcount;      // SA hits (sum of rr) per (read, phase): routes an item to its vote-table tier
    LrmDevCounters *d_counters;
    uint64_t hist_n;         // reads of the last seed launch (with the phase-0 decision count at h_err + 2: round policy)
    volatile uint32_t *h_err;   // error word (pinned host memory, 64 B block) and its device alias
    uint32_t *d_err;
    // bit-sliced GACT: planar reads (wpr words per read) and per-read "byte other than ACGT" flags
    uint64_t *d_qpl;
    uint64_t qpl_wpr;
    uint32_t *d_rflags;
    uint32_t *d_ckpt;        // checkpoint scratch of the bit-sliced kernel
    uint64_t *d_codes;       // 2-bit CIGAR codes (expanded to bytes by bs_expand_kernel)
    uint64_t codes_cw;
    int32_t *d_ncodes;
};

#define LRM_BS_PADW 24       // planar words of padding on either side of a packed sequence
#define LRM_BS_MIN_READS 16384
struct LrmBsArgs {
    const uint64_t *qpl; uint64_t wpr;     // planar reads
    const uint32_t *flags;                 // per read: holds a byte other than ACGT
    const uint64_t *cpl;                   // planar text
    uint32_t *ckpt;                        // checkpoint scratch, lrm_bs_ckpt_words(n) words
    uint64_t *codes; uint64_t cw;          // 2-bit CIGAR codes, cw words per read
    int32_t *ncodes;                       // codes per read (the rest of n_ops is the 'I' tail)
};
uint64_t lrm_bs_planar_words(uint64_t len);
uint64_t lrm_bs_code_words(uint32_t max_len);
uint64_t lrm_bs_ckpt_words(uint64_t n);
bool lrm_bs_wanted(lrm_gact_params gp, uint64_t n, int gact_impl);      // W <= 128 and (impl 4, or automatic with a large batch)
int lrm_bs_pack_reads(const char *d_reads, uint64_t stride, const uint32_t *d_lens, uint64_t n, uint32_t max_len,
                      uint64_t *d_qpl, uint64_t wpr, uint32_t *d_flags, void *stream);
int lrm_bs_pack_text(const char *d_text, uint64_t len, uint64_t *d_out, uint32_t *d_flag, void *stream);
int lrm_bs_launch(const LrmBsArgs *bs, const uint32_t *d_lens, const lrm_seq_meta *d_meta, const int32_t *d_meta_r,
                  const uint32_t *d_tlens, uint64_t n, int T, int O, int W, uint8_t *d_store, uint64_t store_stride,
                  int32_t *d_n_ops, int32_t *d_score, LrmDevCounters *counters, uint32_t max_waves, void *stream);
int lrm_bs_prepare_index(lrm_index *idx);
int lrm_lcl_prepare_index(lrm_index *idx);       // seed_kernels.hip: the long seed table
void lrm_bs_free_index(lrm_index *idx);

void lrm_set_error(const char *fmt, ...);
// OpenMP team size for the library's host loops: omp_get_max_threads() capped by the CPUs this process may really use
// (affinity mask, cgroup CPU quota) -- a GPU box hands a job 16 of its 256 hardware threads, and a team of 256 on a
// quota of 16 is throttled to a crawl.  OMP_NUM_THREADS still lowers it.
int lrm_host_threads(void);
int lrm_require_device(int device);
int lrm_workspace_create_parts(lrm_workspace **out, lrm_index *idx, uint64_t n_max, uint32_t max_len, uint32_t seed_len,
                               uint32_t thres, int parts);
int lrm_ws_take_error(lrm_workspace *ws);        // -2 + message if a kernel raised the workspace's sticky error word
void lrm_host_ctx_free(lrm_index *idx);          // lrm_host.hip            // hipSetDevice + "no CPU fallback" error
void lrm_time_begin(lrm_workspace *ws, int kernel, void *stream);
void lrm_time_end(lrm_workspace *ws, void *stream);

// launchers implemented in the .hip files (all asynchronous on `stream`)
int lrm_launch_seed(lrm_index *idx, lrm_workspace *ws, const char *d_reads, uint64_t stride,
                    const uint32_t *d_lens, uint64_t n, uint32_t max_len, uint32_t seed_len,
                    uint32_t thres, lrm_entry *d_best, const LrmMapTune &mt, void *stream);
int lrm_launch_extend(lrm_index *idx, lrm_workspace *ws, char *d_reads, uint64_t stride,
                      const uint32_t *d_lens, uint64_t n, uint32_t max_len,
                      const lrm_entry *d_best, lrm_gact_params gp, uint8_t *d_store,
                      uint64_t store_stride, int32_t *d_n_ops, int32_t *d_score,
                      lrm_seq_meta *d_meta, int32_t *d_meta_r, const LrmMapTune &mt, void *stream);
int lrm_launch_debug_seed(lrm_index *idx, const char *d_read, uint32_t len, uint32_t seed_len,
                          uint64_t *d_reads2, uint64_t words, int32_t *d_j, uint64_t *d_rr,
                          uint64_t *d_k, uint64_t *d_l, uint64_t cap, void *stream);
