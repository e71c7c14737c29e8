// lrm_api.hip -- C-ABI glue of liblrm_accel.so (include/lrm_accel.h):
// device image packing, upload/adopt, workspaces, host-buffer and device-buffer batch calls.
// No CPU fallback anywhere: without a HIP device every batch entry point returns an error.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <algorithm>
#include <dlfcn.h>
#include <sched.h>
#include <omp.h>
#include "lrm_internal.h"

static thread_local char g_err[512] = "";

void lrm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

extern "C" const char *lrm_last_error(void) { return g_err; }
int lrm_host_threads(void) {
    // the CPUs this process may use (affinity mask, cgroup quota) are read once; the caller's OpenMP thread limit is
    // followed on every call (a host program may lower it between calls)
    static const int cap = []() {
        int m = 1 << 20;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && c < m) m = c; }
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[64];
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {                                                                    // cgroup v1
            FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {} else quota = -1;
            if (fq) fclose(fq);
            if (fp) fclose(fp);
        }
        if (quota > 0 && period > 0) { const int c = (int) (quota / period); if (c >= 1 && c < m) m = c; }
        return m < 1 ? 1 : m;
    }();
    const int m = omp_get_max_threads();
    const int n = m < cap ? m : cap;
    return n < 1 ? 1 : n;
}

extern "C" int lrm_abi_version(void) { return LRM_ABI_VERSION; }

extern "C" int lrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }
#define LRM_LCX_MAX 4096    // capacity of the long-interval side table

// ------------------------------------------------------------------------------------------
// options: the caller's structs, then the LRM_* environment overrides as they stood when the handle was created
// ------------------------------------------------------------------------------------------
static const char *const k_env_names[] = {
    "LRM_SA_SAMPLED", "LRM_LC_LONG", "LRM_LC_PAIR", "LRM_LC_BYTES", "LRM_LC_CORE", "LRM_LCX_THRESHOLD", "LRM_SD", "LRM_SD_SHARE", "LRM_SD_BITS",   // index
    "LRM_GACT_IMPL", "LRM_SEED_ROUNDS", "LRM_HOST_DENSE", "LRM_HOST_SLICE", "LRM_HOST_SUBS",
    "LRM_HOST_GROUP", "LRM_BS_WAVES", "LRM_SS_ITEMS", "LRM_VOTE_VG", "LRM_VOTE_T1", "LRM_VOTE_U", "LRM_VOTE_LOAD",
    "LRM_HOST_EXT_STREAMS", "LRM_HOST_SEED_STREAMS", "LRM_HOST_VERBOSE", "LRM_VOTE_FAST", "LRM_HOST_SLOTS", "LRM_SS_PAD"};
static_assert(sizeof(k_env_names) / sizeof(k_env_names[0]) <= LrmEnv::MAXV, "LrmEnv too small");

void lrm_env_snapshot(LrmEnv *e) {
    e->n = 0;
    for (const char *name : k_env_names)
        if (const char *v = getenv(name)) {
            e->name[e->n] = name;
            e->val[e->n] = *v ? strtoll(v, nullptr, 0) : 1;            // a variable set to the empty string counts as 1
            e->n++;
        }
}
bool LrmEnv::get(const char *key, long long *out) const {
    for (int i = 0; i < n; ++i) if (strcmp(name[i], key) == 0) { *out = val[i]; return true; }
    return false;
}

extern "C" void lrm_index_options_init(lrm_index_options *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t) sizeof(*o);
    o->lc_long = -1;
    o->lc_pair = -1;
    o->lc_core = -1;
    o->seed_table = -1;
}
extern "C" void lrm_map_options_init(lrm_map_options *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t) sizeof(*o);
}

static inline bool valid_sa_ratio(long long r) { return r >= 2 && r <= 64 && (r & (r - 1)) == 0; }

// sa_sampled = r (a power of two, 2..64): keep only SA rows i*r in the image -- the reference's `csa` table
// (fmidx.c:153-163, csa_ratio 4) -- and locate the other rows by LF steps on the device (csa_access,
// fmidx.c:315-331: expected r steps per row, no fixed bound: the walk ends at a stored row or at the '$' row).
void lrm_resolve_index_tune(const lrm_index_options *opt, const LrmEnv &env, LrmIndexTune *t) {
    lrm_index_options o;
    lrm_index_options_init(&o);
    if (opt) memcpy(&o, opt, opt->struct_size && opt->struct_size < sizeof(o) ? opt->struct_size : sizeof(o));
    t->sa_ratio = valid_sa_ratio(o.sa_sampled) ? o.sa_sampled : 1;
    t->lc_long = o.lc_long; t->lc_long_max = o.lc_long_max; t->lc_pair = o.lc_pair;
    t->lc_entry_bytes = o.lc_entry_bytes == 5 || o.lc_entry_bytes == 8 ? (int) o.lc_entry_bytes : 0;
    t->lc_count_bits = o.lc_count_bits >= 2 && o.lc_count_bits <= 24 ? (int) o.lc_count_bits : 0;
    t->lc_core = o.lc_core;
    t->sd = o.seed_table;
    t->sd_len = o.seed_table_len >= 16 && o.seed_table_len <= 24 ? (int) o.seed_table_len : 20;        // alnmain.c:577-580: seed_len 20
    t->sd_f = o.seed_table_share == 2 || o.seed_table_share == 4 ? (int) o.seed_table_share : 0;
    t->sd_bits = o.seed_table_bits >= 4 && o.seed_table_bits <= 32 ? (int) o.seed_table_bits : 0;
    t->sd_cbits = o.seed_table_count_bits >= 2 && o.seed_table_count_bits <= 24 ? (int) o.seed_table_count_bits : 0;
    t->lcx_threshold = o.lcx_threshold >= 1 && o.lcx_threshold <= 0xFFFFFFu ? o.lcx_threshold : 0xFFFFFFull;
    long long v;
    if (env.get("LRM_SA_SAMPLED", &v)) t->sa_ratio = valid_sa_ratio(v) ? (int) v : 1;
    if (env.get("LRM_LC_LONG", &v)) t->lc_long = (int) v;
    if (env.get("LRM_LC_PAIR", &v)) t->lc_pair = v != 0;
    if (env.get("LRM_LC_BYTES", &v) && (v == 5 || v == 8)) t->lc_entry_bytes = (int) v;
    if (env.get("LRM_LC_CORE", &v)) t->lc_core = v != 0;
    if (env.get("LRM_SD", &v)) t->sd = v != 0;
    if (env.get("LRM_SD_SHARE", &v) && (v == 2 || v == 4)) t->sd_f = (int) v;
    if (env.get("LRM_SD_BITS", &v) && v >= 4 && v <= 32) t->sd_bits = (int) v;
    if (env.get("LRM_LCX_THRESHOLD", &v) && v >= 1 && v <= 0xFFFFFFll) t->lcx_threshold = (uint64_t) v;
}

void lrm_resolve_map_tune(const lrm_map_options *opt, const LrmEnv &env, LrmMapTune *t) {
    lrm_map_options o;
    lrm_map_options_init(&o);
    if (opt) memcpy(&o, opt, opt->struct_size && opt->struct_size < sizeof(o) ? opt->struct_size : sizeof(o));
    memset(t, 0, sizeof(*t));
    t->cigar_text = o.cigar_text != 0;
    t->dense = o.dense_results != 0 || t->cigar_text;
    t->gact_impl = o.gact_impl; t->seed_rounds = o.seed_rounds;
    t->slice_reads = o.slice_reads; t->sub_batches = o.sub_batches; t->group_subs = o.group_subs; t->bs_waves = o.bs_waves;
    t->copy_threads = o.copy_threads <= 16 ? o.copy_threads : 16;
    t->keep_reads = o.keep_reads != 0;
    // measured defaults of the kernel knobs (tools/seed_probe.py sweeps them through the environment)
    t->ss_items = 2048; t->vote_vg = 16; t->vote_t1 = LRM_VOTE_T1_LIMIT; t->vote_u = 2; t->vote_load = 50; t->vote_fast = o.vote_exact_only ? 0 : 1;
    t->ext_streams = 2; t->seed_streams = 2;
    long long v;
    if (env.get("LRM_GACT_IMPL", &v)) t->gact_impl = (int) v;
    if (env.get("LRM_SEED_ROUNDS", &v)) t->seed_rounds = (int) v;
    if (env.get("LRM_HOST_DENSE", &v)) t->dense = v != 0 || t->cigar_text;
    if (env.get("LRM_HOST_SLICE", &v) && v >= 1) t->slice_reads = (uint32_t) v;
    if (env.get("LRM_HOST_SUBS", &v) && v >= 1) t->sub_batches = (uint32_t) v;
    if (env.get("LRM_HOST_GROUP", &v) && v >= 1) t->group_subs = (uint32_t) v;
    if (env.get("LRM_BS_WAVES", &v) && v >= 1) t->bs_waves = (uint32_t) v;
    if (env.get("LRM_SS_ITEMS", &v) && (v == 1024 || v == 2048 || v == 4096)) t->ss_items = (uint32_t) v;
    if (env.get("LRM_SS_PAD", &v) && v >= 0 && v <= 60000) t->ss_lds_pad = (uint32_t) v;
    if (env.get("LRM_VOTE_VG", &v) && v >= 1 && v <= 64) t->vote_vg = (uint32_t) v;
    if (env.get("LRM_VOTE_T1", &v) && v >= 0 && v <= LRM_VOTE_T1_LIMIT) t->vote_t1 = (uint32_t) v;
    if (env.get("LRM_VOTE_U", &v)) t->vote_u = (uint32_t) v;
    if (env.get("LRM_VOTE_LOAD", &v) && v >= 10 && v <= 95) t->vote_load = (uint32_t) v;
    if (env.get("LRM_VOTE_FAST", &v)) t->vote_fast = v != 0;
    if (env.get("LRM_HOST_EXT_STREAMS", &v) && v >= 1 && v <= 4) t->ext_streams = (int) v;
    if (env.get("LRM_HOST_SEED_STREAMS", &v) && v >= 1 && v <= 3) t->seed_streams = (int) v;
    if (env.get("LRM_HOST_VERBOSE", &v)) t->verbose = v != 0;
}

static void blob_layout(uint64_t length, int hlen, int mta_len, int sa_ratio, LrmBlobHeader *h) {
    memset(h, 0, sizeof(*h));
    h->magic = LRM_BLOB_MAGIC;
    h->version = LRM_ABI_VERSION;
    h->length = length;
    h->hlen = hlen;
    h->mta_len = mta_len;
    h->n_blocks = (length + LRM_OCC_ROWS - 1) / LRM_OCC_ROWS + 1;   // +1: rank(loc) may touch the block of L-1 only; spare block keeps gathers in bounds
    h->lc_entries = 1ull << (2 * hlen);
    h->sa_ratio = (uint64_t) sa_ratio;
    h->sa_len = sa_ratio > 1 ? (length + sa_ratio - 1) / sa_ratio : length;
    h->con_len = length;
    uint64_t off = sizeof(LrmBlobHeader);
    h->off_occ = off;       off = align256(off + h->n_blocks * sizeof(LrmOccBlock));
    h->off_lc = off;        off = align256(off + h->lc_entries * 8);
    h->off_lcx = off;       off = align256(off + LRM_LCX_MAX * 24);
    h->off_sa = off;        off = align256(off + h->sa_len * 8);
    h->off_content = off;   off = align256(off + h->con_len + 1);
    h->off_mta = off;       off = align256(off + (uint64_t) (mta_len > 0 ? mta_len : 1) * sizeof(LrmMtaDev));
    h->total_bytes = off;
}

static void tune_of(const lrm_index_options *opt, LrmIndexTune *t) {
    LrmEnv env;
    lrm_env_snapshot(&env);
    lrm_resolve_index_tune(opt, env, t);
}

extern "C" uint64_t lrm_index_blob_bytes_opt(uint64_t length, int hlen, int mta_len, const lrm_index_options *opt) {
    LrmIndexTune t;
    tune_of(opt, &t);
    LrmBlobHeader h;
    blob_layout(length, hlen, mta_len, t.sa_ratio, &h);
    return h.total_bytes;
}
extern "C" uint64_t lrm_index_blob_bytes(uint64_t length, int hlen, int mta_len) {
    return lrm_index_blob_bytes_opt(length, hlen, mta_len, nullptr);
}

static inline int code_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// The image is produced SECTION BY SECTION in pieces of <= LRM_PACK_PIECE bytes, every piece by all host
// threads, so that the same code fills a host blob (lrm_index_pack_blob) or a pair of pinned chunks whose DMA
// overlaps the packing of the next piece (lrm_index_upload: no host copy of the image -- GRCh38 is a 63 GB
// image next to 75 GB of reference-layout arrays).
#define LRM_PACK_PIECE (64ull << 20)
struct BlobPacker {
    const lrm_dna_fmi *fmi; const lrm_lc_hash *lch; const lrm_sa_mem *sa; const char *content;
    const lrm_mta_entry *mta; int mta_len;
    LrmBlobHeader h;
    uint64_t L;
    static constexpr uint64_t SEG = 1ull << 20;          // rows per segment of the bwt prefix counts
    std::vector<uint64_t> seg_cnt;                       // [seg][4]: # of A,C,G,T in bwt[0 .. seg*SEG)
    uint64_t total[4];
    std::vector<uint64_t> lcx;                           // side table, sorted {code, k, l}

    int init(const lrm_dna_fmi *fmi_, const lrm_lc_hash *lch_, const lrm_sa_mem *sa_, const char *content_,
             uint64_t con_len, const lrm_mta_entry *mta_, int mta_len_, const lrm_index_options *opt) {
        LrmIndexTune tune;
        tune_of(opt, &tune);
        fmi = fmi_; lch = lch_; sa = sa_; content = content_; mta = mta_; mta_len = mta_len_;
        if (!fmi || !lch || !sa || !content) { lrm_set_error("null argument"); return -1; }
        L = fmi->length;
        if (L < 2 || L >= (1ull << 40)) { lrm_set_error("text length %llu outside [2, 2^40)", (unsigned long long) L); return -1; }
        if (con_len != L) { lrm_set_error("content length %llu != fm length %llu", (unsigned long long) con_len, (unsigned long long) L); return -1; }
        if (sa->len < L) { lrm_set_error("suffix array has %llu rows, need %llu", (unsigned long long) sa->len, (unsigned long long) L); return -1; }
        if (lch->hlen < 1 || lch->hlen > 15) { lrm_set_error("hlen %d outside [1,15] (lchash.c:75-77)", lch->hlen); return -1; }
        if (lch->len != 2ull << (2 * lch->hlen)) { lrm_set_error("lc table length %llu != 2*4^hlen", (unsigned long long) lch->len); return -1; }
        if (mta_len < 0 || (mta_len > 0 && !mta)) { lrm_set_error("bad mta"); return -1; }
        blob_layout(L, lch->hlen, mta_len, tune.sa_ratio, &h);
        h.c4[0] = fmi->c[(unsigned char) 'A']; h.c4[1] = fmi->c[(unsigned char) 'C'];
        h.c4[2] = fmi->c[(unsigned char) 'G']; h.c4[3] = fmi->c[(unsigned char) 'T'];

        // pass 1 over the bwt: per-segment symbol counts (parallel), the '$' row, alphabet check
        const uint64_t nseg = (L + SEG - 1) / SEG;
        seg_cnt.assign((nseg + 1) * 4, 0);
        uint64_t dollar = ~0ull, n_dollar = 0, bad_row = ~0ull;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(dynamic, 4) reduction(+ : n_dollar) reduction(min : dollar, bad_row)
        for (uint64_t sg = 0; sg < nseg; ++sg) {
            const uint64_t lo = sg * SEG, hi = lo + SEG < L ? lo + SEG : L;
            uint64_t c[4] = {0, 0, 0, 0};
            for (uint64_t i = lo; i < hi; ++i) {
                const char ch = fmi->bwt[i];
                const int code = code_of(ch);
                if (code >= 0) c[code]++;
                else if (ch == '$') { n_dollar++; if (i < dollar) dollar = i; }
                else if (i < bad_row) bad_row = i;
            }
            for (int x = 0; x < 4; ++x) seg_cnt[(sg + 1) * 4 + x] = c[x];
        }
        if (bad_row != ~0ull || n_dollar > 1) {
            const uint64_t r = bad_row != ~0ull ? bad_row : dollar;
            lrm_set_error("bwt row %llu holds byte 0x%02x (only upper-case ACGT and one '$' supported)", (unsigned long long) r, (unsigned) (unsigned char) fmi->bwt[r]);
            return -1;
        }
        if (n_dollar == 0) { lrm_set_error("bwt has no '$' row"); return -1; }
        h.dollar_row = dollar;
        for (uint64_t sg = 1; sg <= nseg; ++sg)
            for (int x = 0; x < 4; ++x) seg_cnt[sg * 4 + x] += seg_cnt[(sg - 1) * 4 + x];
        for (int x = 0; x < 4; ++x) total[x] = seg_cnt[nseg * 4 + x];

        // side table of the lc intervals that do not fit 24 bits of length
        const uint64_t long_thr = tune.lcx_threshold;           // (tests send shorter intervals through the side table too)
        lcx_thr = long_thr;
        std::vector<uint64_t> over;
        const uint64_t ne = h.lc_entries;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t num = 0; num < ne; ++num) {
            const uint64_t k = lch->lc[2 * num], l = lch->lc[2 * num + 1];
            if (k == 0 && l == 0) continue;
            const uint64_t cnt = l >= k ? l - k + 1 : 0;
            if (cnt == 0 || cnt >= long_thr || k >= (1ull << 40)) {
                const uint64_t code = rev_groups(num, lch->hlen);
#pragma omp critical
                { over.push_back(code); over.push_back(k); over.push_back(l); }
            }
        }
        if (over.size() / 3 > LRM_LCX_MAX) { lrm_set_error("too many long lchash intervals (%zu)", over.size() / 3); return -1; }
        std::vector<size_t> ord(over.size() / 3);
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return over[3 * a] < over[3 * b]; });
        lcx.assign(3 * (size_t) LRM_LCX_MAX, ~0ull);
        for (size_t i = 0; i < ord.size(); ++i)
            for (int f = 0; f < 3; ++f) lcx[3 * i + f] = over[3 * ord[i] + f];
        h.n_lcx = ord.size();
        return 0;
    }

    uint64_t lcx_thr = 0xFFFFFFull;
    static inline uint64_t rev_groups(uint64_t v, int hl) {          // reverse the order of the 2-bit groups
        uint64_t code = 0;
        for (int i = 0; i < hl; ++i) { code = (code << 2) | (v & 3); v >>= 2; }
        return code;
    }

    // occ blocks [b0, b0 + nb): one {C[sym] + prefix count, occurrence mask} pair per symbol and 64 bwt rows;
    // cross-checked against the reference's sampled O table (fmidx.c:128-150)
    int fill_occ(uint64_t b0, uint64_t nb, LrmOccBlock *dst) const {
        const uint64_t ratio = (uint64_t) fmi->o_ratio;
        const uint64_t bps = SEG / LRM_OCC_ROWS;           // blocks per segment
        uint64_t bad = ~0ull;
        const uint64_t s0 = b0 / bps, s1 = (b0 + nb + bps - 1) / bps;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(dynamic, 1) reduction(min : bad)
        for (uint64_t sg = s0; sg < s1; ++sg) {
            uint64_t run[4];
            const uint64_t nseg = (L + SEG - 1) / SEG;
            const uint64_t sgc = sg < nseg ? sg : nseg;
            for (int x = 0; x < 4; ++x) run[x] = seg_cnt[sgc * 4 + x];
            const uint64_t blo = sg * bps > b0 ? sg * bps : b0, bhi = (sg + 1) * bps < b0 + nb ? (sg + 1) * bps : b0 + nb;
            // rows of the segment before blo (a piece boundary inside a segment): count them
            for (uint64_t i = sg * SEG; i < blo * LRM_OCC_ROWS && i < L; ++i) { const int c = code_of(fmi->bwt[i]); if (c >= 0) run[c]++; }
            for (uint64_t b = blo; b < bhi; ++b) {
                LrmOccBlock blk;
                for (int x = 0; x < 4; ++x) { blk.sym[x].cnt = h.c4[x] + run[x]; blk.sym[x].mask = 0; }
                const uint64_t r0 = b * LRM_OCC_ROWS, r1 = r0 + LRM_OCC_ROWS < L ? r0 + LRM_OCC_ROWS : L;
                for (uint64_t i = r0; i < r1; ++i) {
                    if (fmi->o && ratio > 0 && i % ratio == 0) {
                        const uint64_t *o = fmi->o + 4 * (i / ratio);
                        if ((o[0] != run[0] || o[1] != run[1] || o[2] != run[2] || o[3] != run[3]) && i < bad) bad = i;
                    }
                    const int c = code_of(fmi->bwt[i]);
                    if (c >= 0) { run[c]++; blk.sym[c].mask |= 1ull << (i & 63); }
                }
                dst[b - b0] = blk;
            }
        }
        if (bad != ~0ull) { lrm_set_error("O table disagrees with bwt at row %llu", (unsigned long long) bad); return -1; }
        return 0;
    }

    // lc entries [c0, c0 + n) in device order (LSB-first code): gathered from the reference's table
    void fill_lc(uint64_t c0, uint64_t n, uint64_t *dst) const {
        const int hl = lch->hlen;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t num = rev_groups(c0 + i, hl);               // the permutation is an involution
            const uint64_t k = lch->lc[2 * num], l = lch->lc[2 * num + 1];
            uint64_t e = 0;
            if (!(k == 0 && l == 0)) {
                uint64_t cnt = l >= k ? l - k + 1 : 0;
                if (cnt == 0 || cnt >= lcx_thr || k >= (1ull << 40)) cnt = 0xFFFFFFull;
                e = (k & ((1ull << 40) - 1ull)) | (cnt << 40);
            }
            dst[i] = e;
        }
    }

    // SA entries [e0, e0 + n) of the image: rows e*sa_ratio, as u64 (sa_use.h:27-29)
    void fill_sa(uint64_t e0, uint64_t n, uint64_t *dst) const {
        const uint64_t r = h.sa_ratio;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t i = 0; i < n; ++i) {
            const lrm_ui40 &v = sa->mem[(e0 + i) * r];
            dst[i] = ((uint64_t) v.high << 32) | (uint64_t) v.low;
        }
    }

    // Emits the image in order as (offset, bytes) pieces through `sink`, which may consume the buffer
    // asynchronously: next_buf() hands out the buffer for the next piece (>= LRM_PACK_PIECE bytes).
    template <typename NextBuf, typename Sink>
    int emit(NextBuf next_buf, Sink sink) const {
        {   // header
            uint8_t *b = next_buf();
            if (!b) return -1;
            memcpy(b, &h, sizeof(h));
            if (sink(0, sizeof(h), b)) return -1;
        }
        const uint64_t bpp = LRM_PACK_PIECE / sizeof(LrmOccBlock);
        for (uint64_t b0 = 0; b0 < h.n_blocks; b0 += bpp) {
            const uint64_t nb = h.n_blocks - b0 < bpp ? h.n_blocks - b0 : bpp;
            uint8_t *b = next_buf();
            if (!b) return -1;
            if (fill_occ(b0, nb, (LrmOccBlock *) b)) return -1;
            if (sink(h.off_occ + b0 * sizeof(LrmOccBlock), nb * sizeof(LrmOccBlock), b)) return -1;
        }
        const uint64_t epp = LRM_PACK_PIECE / 8;
        for (uint64_t c0 = 0; c0 < h.lc_entries; c0 += epp) {
            const uint64_t n = h.lc_entries - c0 < epp ? h.lc_entries - c0 : epp;
            uint8_t *b = next_buf();
            if (!b) return -1;
            fill_lc(c0, n, (uint64_t *) b);
            if (sink(h.off_lc + c0 * 8, n * 8, b)) return -1;
        }
        {
            uint8_t *b = next_buf();
            if (!b) return -1;
            memcpy(b, lcx.data(), (size_t) LRM_LCX_MAX * 24);
            if (sink(h.off_lcx, (uint64_t) LRM_LCX_MAX * 24, b)) return -1;
        }
        for (uint64_t e0 = 0; e0 < h.sa_len; e0 += epp) {
            const uint64_t n = h.sa_len - e0 < epp ? h.sa_len - e0 : epp;
            uint8_t *b = next_buf();
            if (!b) return -1;
            fill_sa(e0, n, (uint64_t *) b);
            if (sink(h.off_sa + e0 * 8, n * 8, b)) return -1;
        }
        for (uint64_t o = 0; o < L + 1; o += LRM_PACK_PIECE) {
            const uint64_t n = L + 1 - o < LRM_PACK_PIECE ? L + 1 - o : LRM_PACK_PIECE;
            uint8_t *b = next_buf();
            if (!b) return -1;
            const uint64_t nc = o + n > L ? L - o : n;                 // the byte after the text is a NUL
            const uint64_t piece = 1ull << 20, np = (nc + piece - 1) / piece;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
            for (uint64_t i = 0; i < np; ++i) {
                const uint64_t po = i * piece, pl = nc - po < piece ? nc - po : piece;
                memcpy(b + po, content + o + po, pl);
            }
            if (nc < n) b[nc] = 0;
            if (sink(h.off_content + o, n, b)) return -1;
        }
        {
            uint8_t *b = next_buf();
            if (!b) return -1;
            LrmMtaDev *md = (LrmMtaDev *) b;
            for (int i = 0; i < mta_len; ++i) { md[i].offset = mta[i].offset; md[i].seq_len = (uint64_t) mta[i].seq_len; }
            const uint64_t bytes = (uint64_t) (mta_len > 0 ? mta_len : 1) * sizeof(LrmMtaDev);
            if (mta_len == 0) memset(b, 0, bytes);
            if (sink(h.off_mta, bytes, b)) return -1;
        }
        return 0;
    }
};

extern "C" int lrm_index_pack_blob(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                   const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                   void *blob, uint64_t blob_bytes) {
    return lrm_index_pack_blob_opt(fmi, lch, sa, content, con_len, mta, mta_len, blob, blob_bytes, nullptr);
}
extern "C" int lrm_index_pack_blob_opt(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                       const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                       void *blob, uint64_t blob_bytes, const lrm_index_options *opt) {
    if (!blob) { lrm_set_error("null argument"); return -1; }
    BlobPacker pk;
    if (pk.init(fmi, lch, sa, content, con_len, mta, mta_len, opt)) return -1;
    if (mta_len > 0 && (uint64_t) mta_len * sizeof(LrmMtaDev) > LRM_PACK_PIECE) { lrm_set_error("too many sequences"); return -1; }
    if (blob_bytes < pk.h.total_bytes) { lrm_set_error("blob buffer too small"); return -1; }
    // pieces are written in place: the "next buffer" is the piece's own position in the blob, so the gaps between
    // the 256-byte aligned sections are cleared first
    uint8_t *base = (uint8_t *) blob;
    // the emit order is fixed, so the destination of every piece is known in advance: replay the layout
    struct Cursor { const BlobPacker *pk; uint64_t sec, pos; } cur = {&pk, 0, 0};
    auto piece_offset = [&]() -> uint64_t {
        const LrmBlobHeader &h = pk.h;
        const uint64_t secs[7][3] = {{0, sizeof(LrmBlobHeader), sizeof(LrmBlobHeader)},
                                      {h.off_occ, h.n_blocks * sizeof(LrmOccBlock), LRM_PACK_PIECE},
                                      {h.off_lc, h.lc_entries * 8, LRM_PACK_PIECE},
                                      {h.off_lcx, (uint64_t) LRM_LCX_MAX * 24, (uint64_t) LRM_LCX_MAX * 24},
                                      {h.off_sa, h.sa_len * 8, LRM_PACK_PIECE},
                                      {h.off_content, pk.L + 1, LRM_PACK_PIECE},
                                      {h.off_mta, 1, 1}};
        while (cur.sec < 7 && cur.pos >= secs[cur.sec][1]) { cur.sec++; cur.pos = 0; }
        const uint64_t off = secs[cur.sec][0] + cur.pos;
        cur.pos += secs[cur.sec][2];
        return off;
    };
    {   // clear alignment gaps (and the spare tail) so that images of equal inputs are byte-identical
        const LrmBlobHeader &h = pk.h;
        const uint64_t ends[6][2] = {{h.off_occ + h.n_blocks * sizeof(LrmOccBlock), h.off_lc}, {h.off_lc + h.lc_entries * 8, h.off_lcx},
                                      {h.off_lcx + (uint64_t) LRM_LCX_MAX * 24, h.off_sa}, {h.off_sa + h.sa_len * 8, h.off_content},
                                      {h.off_content + pk.L + 1, h.off_mta},
                                      {h.off_mta + (uint64_t) (mta_len > 0 ? mta_len : 1) * sizeof(LrmMtaDev), h.total_bytes}};
        for (auto &e : ends) if (e[1] > e[0]) memset(base + e[0], 0, e[1] - e[0]);
    }
    return pk.emit([&]() -> uint8_t * { return base + piece_offset(); },
                   [&](uint64_t off, uint64_t bytes, const uint8_t *buf) -> int {
                       if (buf != base + off) { lrm_set_error("internal: piece order"); return -1; }
                       (void) bytes;
                       return 0;
                   });
}

static int make_handle(lrm_index **out, void *d_blob, uint64_t bytes, int device, int owns, const LrmBlobHeader &h,
                       const lrm_index_options *opt) {
    if (h.magic != LRM_BLOB_MAGIC || h.version != LRM_ABI_VERSION) { lrm_set_error("not an lrm index image (magic/version)"); return -1; }
    if (h.total_bytes > bytes) { lrm_set_error("index image truncated"); return -1; }
    lrm_index *ix = new (std::nothrow) lrm_index;
    if (!ix) { lrm_set_error("out of memory"); return -1; }
    memset(ix, 0, sizeof(*ix));
    ix->d_blob = d_blob; ix->blob_bytes = bytes; ix->owns_blob = owns; ix->device = device; ix->hdr = h;
    uint8_t *b = (uint8_t *) d_blob;
    ix->view.occ = (const LrmOccBlock *) (b + h.off_occ);
    ix->view.lc = (const uint64_t *) (b + h.off_lc);
    ix->view.lcx = (const uint64_t *) (b + h.off_lcx);
    ix->view.n_lcx = h.n_lcx;
    ix->view.sa = (const uint64_t *) (b + h.off_sa);
    ix->view.content = (const char *) (b + h.off_content);
    ix->view.mta = (const LrmMtaDev *) (b + h.off_mta);
    ix->view.length = h.length; ix->view.dollar_row = h.dollar_row;
    ix->view.sa_len = h.sa_len; ix->view.con_len = h.con_len;
    for (int i = 0; i < 4; ++i) ix->view.c4[i] = h.c4[i];
    ix->view.hlen = h.hlen; ix->view.mta_len = h.mta_len;
    ix->view.core = nullptr;
    ix->view.sd = nullptr; ix->view.sdx = nullptr; ix->view.sdx_mask = 0;
    ix->view.sd_len = ix->view.sd_f = ix->view.sd_bits = ix->view.sd_kbits = ix->view.sd_slot = ix->view.sd_cbits = 0;
    ix->view.lcl = nullptr; ix->view.hl = 0; ix->view.lcl_pair = 0; ix->view.lcl_kbits = 0; ix->view.lclx = nullptr; ix->view.lclx_mask = 0;
    ix->view.sa_shift = 0;
    for (uint64_t r = h.sa_ratio > 1 ? h.sa_ratio : 1; r > 1; r >>= 1) ix->view.sa_shift++;
    ix->n_peers = 1;
    lrm_env_snapshot(&ix->env);                          // the LRM_* overrides are read here, once per handle
    lrm_resolve_index_tune(opt, ix->env, &ix->itune);
    lrm_resolve_map_tune(nullptr, ix->env, &ix->mtune);
    if (lrm_bs_prepare_index(ix)) { delete ix; return -1; }
    if (lrm_lcl_prepare_index(ix)) { lrm_bs_free_index(ix); delete ix; return -1; }
    *out = ix;
    return 0;
}

int lrm_require_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        lrm_set_error("no HIP device available (%s): liblrm_accel has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return -1;
    }
    if (device < 0 || device >= n) { lrm_set_error("device %d out of range (have %d)", device, n); return -1; }
    HIPCHK(hipSetDevice(device));
    return 0;
}
#define require_device lrm_require_device

extern "C" int lrm_index_upload_blob(lrm_index **out, const void *blob, uint64_t blob_bytes, int device) {
    return lrm_index_upload_blob_opt(out, blob, blob_bytes, device, nullptr);
}
extern "C" int lrm_index_upload_blob_opt(lrm_index **out, const void *blob, uint64_t blob_bytes, int device,
                                         const lrm_index_options *opt) {
    if (!out || !blob || blob_bytes < sizeof(LrmBlobHeader)) { lrm_set_error("bad blob"); return -1; }
    if (require_device(device)) return -1;
    LrmBlobHeader h;
    memcpy(&h, blob, sizeof(h));
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, blob_bytes));
    if (hipMemcpy(d, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(d); lrm_set_error("index upload failed"); return -1; }
    if (make_handle(out, d, blob_bytes, device, 1, h, opt)) { (void) hipFree(d); return -1; }
    return 0;
}

extern "C" int lrm_index_adopt_device(lrm_index **out, void *d_blob, uint64_t blob_bytes, int device) {
    return lrm_index_adopt_device_opt(out, d_blob, blob_bytes, device, nullptr);
}
extern "C" int lrm_index_adopt_device_opt(lrm_index **out, void *d_blob, uint64_t blob_bytes, int device,
                                          const lrm_index_options *opt) {
    if (!out || !d_blob || blob_bytes < sizeof(LrmBlobHeader)) { lrm_set_error("bad blob"); return -1; }
    if (require_device(device)) return -1;
    LrmBlobHeader h;
    HIPCHK(hipMemcpy(&h, d_blob, sizeof(h), hipMemcpyDeviceToHost));
    return make_handle(out, d_blob, blob_bytes, device, 0, h, opt);
}

// pack + upload without a host copy of the image: two pinned chunks, the DMA of one overlaps the packing of the other
static int stream_image(const BlobPacker &pk, void *d_dst) {
    struct Res {
        void *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t st = nullptr;
        ~Res() {
            if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
            for (int i = 0; i < 2; ++i) { if (pin[i]) (void) hipHostFree(pin[i]); if (ev[i]) (void) hipEventDestroy(ev[i]); }
        }
    } r;
    HIPCHK(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipHostMalloc(&r.pin[i], LRM_PACK_PIECE, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
    HIPCHK(hipMemsetAsync(d_dst, 0, pk.h.total_bytes, r.st));          // alignment gaps: images of equal inputs are byte-identical
    uint64_t seq = 0;
    bool used[2] = {false, false};
    int rc = pk.emit(
        [&]() -> uint8_t * {
            const int b = (int) (seq & 1);
            if (used[b] && hipEventSynchronize(r.ev[b]) != hipSuccess) { lrm_set_error("index upload: event wait failed"); return nullptr; }
            return (uint8_t *) r.pin[b];
        },
        [&](uint64_t off, uint64_t n, const uint8_t *buf) -> int {
            const int b = (int) (seq & 1);
            HIPCHK(hipMemcpyAsync((uint8_t *) d_dst + off, buf, n, hipMemcpyHostToDevice, r.st));
            HIPCHK(hipEventRecord(r.ev[b], r.st));
            used[b] = true;
            ++seq;
            return 0;
        });
    if (rc) return -1;
    HIPCHK(hipStreamSynchronize(r.st));
    return 0;
}

static int upload_one(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                      const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len, int device,
                      const lrm_index_options *opt) {
    if (!out || !fmi || !lch) { lrm_set_error("null argument"); return -1; }
    if (require_device(device)) return -1;
    BlobPacker pk;
    if (pk.init(fmi, lch, sa, content, con_len, mta, mta_len, opt)) return -1;
    if (mta_len > 0 && (uint64_t) mta_len * sizeof(LrmMtaDev) > LRM_PACK_PIECE) { lrm_set_error("too many sequences"); return -1; }
    const uint64_t bytes = pk.h.total_bytes;
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { (void) hipGetLastError(); lrm_set_error("hipMalloc of the %llu-byte index image failed", (unsigned long long) bytes); return -1; }
    if (stream_image(pk, d) || make_handle(out, d, bytes, device, 1, pk.h, opt)) { (void) hipFree(d); return -1; }
    return 0;
}

extern "C" int lrm_index_upload(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len, int device) {
    return upload_one(out, fmi, lch, sa, content, con_len, mta, mta_len, device, nullptr);
}

// the same into device memory the caller owns (e.g. a buffer that is then broadcast to the other ranks and
// adopted with lrm_index_adopt_device on every rank)
extern "C" int lrm_index_pack_device(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                     const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                     void *d_blob, uint64_t blob_bytes, int device) {
    return lrm_index_pack_device_opt(fmi, lch, sa, content, con_len, mta, mta_len, d_blob, blob_bytes, device, nullptr);
}
extern "C" int lrm_index_pack_device_opt(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                         const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                         void *d_blob, uint64_t blob_bytes, int device, const lrm_index_options *opt) {
    if (!fmi || !lch || !d_blob) { lrm_set_error("null argument"); return -1; }
    if (require_device(device)) return -1;
    BlobPacker pk;
    if (pk.init(fmi, lch, sa, content, con_len, mta, mta_len, opt)) return -1;
    if (mta_len > 0 && (uint64_t) mta_len * sizeof(LrmMtaDev) > LRM_PACK_PIECE) { lrm_set_error("too many sequences"); return -1; }
    if (blob_bytes < pk.h.total_bytes) { lrm_set_error("device buffer too small for the image"); return -1; }
    return stream_image(pk, d_blob);
}

// ------------------------------------------------------------------------------------------
// multi-GPU group: the image is packed and uploaded once (device devices[0]) and replicated to the other
// devices over xGMI -- one RCCL broadcast when the devices are distinct and librccl is loadable, else
// hipMemcpyPeer (or a plain device copy when a device is listed twice: a logical replica, used by tests on a
// one-GPU box).  Every replica derives its own planar text / long seed table on its device.
// ------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    typedef int (*init_all_t)(void **, int, const int *);
    typedef int (*bcast_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*group_t)(void);
    typedef int (*destroy_t)(void *);
    typedef const char *(*errstr_t)(int);
    void *lib = nullptr;
    init_all_t init_all = nullptr; bcast_t bcast = nullptr; group_t gstart = nullptr, gend = nullptr; destroy_t destroy = nullptr;
    errstr_t errstr = nullptr;
    bool load() {
        if (getenv("LRM_NO_RCCL")) return false;
        for (const char *name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) return false;
        init_all = (init_all_t) dlsym(lib, "ncclCommInitAll"); bcast = (bcast_t) dlsym(lib, "ncclBroadcast");
        gstart = (group_t) dlsym(lib, "ncclGroupStart"); gend = (group_t) dlsym(lib, "ncclGroupEnd");
        destroy = (destroy_t) dlsym(lib, "ncclCommDestroy"); errstr = (errstr_t) dlsym(lib, "ncclGetErrorString");
        return init_all && bcast && gstart && gend && destroy;
    }
};

// ncclBroadcast of the image from devs[0] into bufs[1..] (rccl.h:591; ncclUint8 = 1), in pieces of 256 MiB so
// that RCCL pipelines across the xGMI links.  Returns 1 if RCCL is unavailable (caller falls back), -1 on error.
int rccl_broadcast(const std::vector<int> &devs, const std::vector<void *> &bufs, uint64_t bytes) {
    Rccl r;
    if (!r.load()) return 1;
    const int n = (int) devs.size();
    std::vector<void *> comms((size_t) n, nullptr);
    int rc = r.init_all(comms.data(), n, devs.data());
    if (rc != 0) { lrm_set_error("ncclCommInitAll failed: %s", r.errstr ? r.errstr(rc) : "?"); return -1; }
    std::vector<hipStream_t> st((size_t) n, nullptr);
    int out = 0;
    for (int i = 0; i < n && !out; ++i)
        if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) out = -1;
    const uint64_t piece = 256ull << 20;
    for (uint64_t o = 0; o < bytes && !out; o += piece) {
        const uint64_t l = bytes - o < piece ? bytes - o : piece;
        r.gstart();
        for (int i = 0; i < n; ++i) {
            rc = r.bcast((const char *) bufs[i] + o, (char *) bufs[i] + o, (size_t) l, 1 /* ncclUint8 */, 0, comms[i], st[i]);
            if (rc != 0) out = -1;
        }
        rc = r.gend();
        if (rc != 0) out = -1;
    }
    for (int i = 0; i < n; ++i) {
        if (st[i]) { (void) hipSetDevice(devs[i]); if (hipStreamSynchronize(st[i]) != hipSuccess) out = -1; (void) hipStreamDestroy(st[i]); }
        if (comms[i]) r.destroy(comms[i]);
    }
    if (out) lrm_set_error("RCCL broadcast of the index image failed%s%s", rc ? ": " : "", rc && r.errstr ? r.errstr(rc) : "");
    return out;
}
}  // namespace

extern "C" int lrm_index_upload_multi(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                      const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                      const int *devices, int ngpus) {
    return lrm_index_upload_opt(out, fmi, lch, sa, content, con_len, mta, mta_len, devices, ngpus, nullptr);
}
extern "C" int lrm_index_upload_opt(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                    const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                    const int *devices, int ngpus, const lrm_index_options *opt) {
    if (!out || ngpus < 1 || ngpus > 64) { lrm_set_error("bad argument (1 <= ngpus <= 64)"); return -1; }
    std::vector<int> devs((size_t) ngpus);
    for (int i = 0; i < ngpus; ++i) devs[i] = devices ? devices[i] : i;
    lrm_index *root = nullptr;
    if (upload_one(&root, fmi, lch, sa, content, con_len, mta, mta_len, devs[0], opt)) return -1;
    if (ngpus == 1) { *out = root; return 0; }
    const uint64_t bytes = root->blob_bytes;
    std::vector<void *> bufs((size_t) ngpus, nullptr);
    bufs[0] = root->d_blob;
    bool distinct = true;
    for (int i = 0; i < ngpus; ++i) for (int k = 0; k < i; ++k) distinct &= devs[i] != devs[k];
    auto cleanup = [&](int upto) { for (int i = 1; i < upto; ++i) if (bufs[i]) { (void) hipSetDevice(devs[i]); (void) hipFree(bufs[i]); } lrm_index_free(root); };
    for (int i = 1; i < ngpus; ++i) {
        if (lrm_require_device(devs[i]) || hipMalloc(&bufs[i], bytes) != hipSuccess) {
            if (!bufs[i]) lrm_set_error("device %d: cannot allocate the %llu-byte index image", devs[i], (unsigned long long) bytes);
            cleanup(i + 1);
            return -1;
        }
    }
    int rc = distinct ? rccl_broadcast(devs, bufs, bytes) : 1;
    if (rc == 1) {                                     // no RCCL (or logical replicas on one device): peer copies
        rc = 0;
        for (int i = 1; i < ngpus && !rc; ++i) {
            (void) hipSetDevice(devs[i]);
            const hipError_t e = devs[i] == devs[0] ? hipMemcpy(bufs[i], bufs[0], bytes, hipMemcpyDeviceToDevice)
                                                     : hipMemcpyPeer(bufs[i], devs[i], bufs[0], devs[0], bytes);
            if (e != hipSuccess) { lrm_set_error("replication of the index image to device %d failed: %s", devs[i], hipGetErrorString(e)); rc = -1; }
        }
    }
    if (rc) { cleanup(ngpus); return -1; }
    root->peers = new (std::nothrow) lrm_index *[(size_t) ngpus];
    if (!root->peers) { cleanup(ngpus); lrm_set_error("out of memory"); return -1; }
    root->peers[0] = root;
    root->n_peers = 1;
    for (int i = 1; i < ngpus; ++i) {
        lrm_index *rep = nullptr;
        if (lrm_require_device(devs[i]) || make_handle(&rep, bufs[i], bytes, devs[i], 1, root->hdr, opt)) {
            for (int k = i; k < ngpus; ++k) { (void) hipSetDevice(devs[k]); (void) hipFree(bufs[k]); }
            lrm_index_free(root);                  // frees the replicas made so far
            return -1;
        }
        root->peers[i] = rep;
        root->n_peers = i + 1;
    }
    *out = root;
    return 0;
}

// Test tap: the RCCL path of lrm_index_upload_multi on ONE device -- dlopen of librccl, ncclCommInitAll, a grouped
// ncclBroadcast of `bytes` bytes on a 1-rank communicator, teardown.  A one-GPU box cannot run the multi-device
// broadcast itself; this checks that the library loads and that the calls are bound with the right signatures.
// Returns 0 ok, 1 RCCL not loadable (the multi-GPU upload then falls back to hipMemcpyPeer), -1 error.
extern "C" int lrm_debug_rccl_selftest(int device, uint64_t bytes) {
    if (lrm_require_device(device)) return -1;
    void *d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) { lrm_set_error("hipMalloc failed"); return -1; }
    std::vector<uint8_t> h(bytes ? bytes : 1);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t) (i * 131u + 7u);
    int rc = hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    if (rc == 0) rc = rccl_broadcast(std::vector<int>{device}, std::vector<void *>{d}, (uint64_t) h.size());
    std::vector<uint8_t> back(h.size());
    if (rc == 0 && hipMemcpy(back.data(), d, h.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    if (rc == 0 && back != h) { lrm_set_error("RCCL self-test: buffer changed by a 1-rank broadcast"); rc = -1; }
    (void) hipFree(d);
    return rc;
}

extern "C" int lrm_index_set_map_options(lrm_index *idx, const lrm_map_options *opt) {
    if (!idx) { lrm_set_error("null argument"); return -1; }
    for (int r = 0; r < idx->n_peers; ++r) {
        lrm_index *ix = idx->peers ? idx->peers[r] : idx;
        lrm_resolve_map_tune(opt, ix->env, &ix->mtune);
        ix->mtune.t3_limit = ix->dbg_t3_limit; ix->mtune.t3_slots = ix->dbg_t3_slots;
    }
    return 0;
}
// tuning sessions (tools/*_probe.py): take the LRM_* variables as they stand NOW for the batch calls of this handle
extern "C" int lrm_debug_reload_env(lrm_index *idx) {
    if (!idx) { lrm_set_error("null argument"); return -1; }
    for (int r = 0; r < idx->n_peers; ++r) {
        lrm_index *ix = idx->peers ? idx->peers[r] : idx;
        lrm_env_snapshot(&ix->env);
        lrm_resolve_map_tune(nullptr, ix->env, &ix->mtune);
        ix->mtune.t3_limit = ix->dbg_t3_limit; ix->mtune.t3_slots = ix->dbg_t3_slots;
    }
    return 0;
}
extern "C" int lrm_debug_set_vote_limits(lrm_index *idx, uint32_t t3_limit, uint32_t t3_slots) {
    if (!idx) { lrm_set_error("null argument"); return -1; }
    if (t3_slots && (t3_slots < 8 || t3_slots > LRM_VOTE_T3_SLOTS)) { lrm_set_error("t3_slots outside [8, %d]", LRM_VOTE_T3_SLOTS); return -1; }
    for (int r = 0; r < idx->n_peers; ++r) {
        lrm_index *ix = idx->peers ? idx->peers[r] : idx;
        ix->dbg_t3_limit = ix->mtune.t3_limit = t3_limit;
        ix->dbg_t3_slots = ix->mtune.t3_slots = t3_slots;
    }
    return 0;
}

extern "C" int lrm_index_get_tables(const lrm_index *idx, lrm_index_tables *out) {
    if (!idx || !out) { lrm_set_error("lrm_index_get_tables: null argument"); return -1; }
    memset(out, 0, sizeof(*out));
    const LrmIndexView &v = idx->view;
    uint64_t bytes = 0;
    if (v.lcl) {
        out->lc_long = v.hl; out->lc_pair = v.lcl_pair; out->lc_entry_bytes = v.lcl_kbits ? 5 : 8;
        bytes += ((v.lcl_pair ? 2ull : 1ull) << (2 * v.hl)) * (uint64_t) out->lc_entry_bytes;
        if (v.lclx) bytes += (v.lclx_mask + 1) * 16;
    }
    if (v.core) { out->lc_core = 1; bytes += 64ull << 26; }
    if (v.sd) {
        out->seed_table_len = v.sd_len; out->seed_table_share = v.sd_f; out->seed_table_bits = v.sd_bits;
        out->seed_table_slot_bytes = v.sd_slot; out->seed_table_count_bits = v.sd_cbits;
        out->seed_table_side_entries = idx->sd_side_entries;
        bytes += (64ull << v.sd_bits) + (v.sdx_mask + 1) * 16;
    }
    out->derived_bytes = bytes;
    return 0;
}

extern "C" int lrm_index_replicas(const lrm_index *idx) { return idx ? idx->n_peers : 0; }
extern "C" lrm_index *lrm_index_replica(lrm_index *idx, int r) {
    if (!idx || r < 0 || r >= idx->n_peers) return nullptr;
    return idx->peers ? idx->peers[r] : idx;
}

// alnmain.c:554-557: the paired-end entry is declared and unimplemented in the reference ("todo"); it returns -1.
extern "C" int lrm_pair_end(int argc, const char *argv[]) { (void) argc; (void) argv; return -1; }

extern "C" void lrm_index_free(lrm_index *idx) {
    if (!idx) return;
    for (int r = 1; r < idx->n_peers && idx->peers; ++r) lrm_index_free(idx->peers[r]);     // replicas of a multi-GPU group
    delete[] idx->peers;
    (void) hipSetDevice(idx->device);
    lrm_host_ctx_free(idx);                      // workspace, device mirrors, pinned staging, streams of the host-buffer calls
    lrm_bs_free_index(idx);
    if (idx->d_lcl) (void) hipFree(idx->d_lcl);
    if (idx->d_lclx) (void) hipFree(idx->d_lclx);
    if (idx->d_core) (void) hipFree(idx->d_core);
    if (idx->d_sd) (void) hipFree(idx->d_sd);
    if (idx->d_sdx) (void) hipFree(idx->d_sdx);
    if (idx->owns_blob && idx->d_blob) (void) hipFree(idx->d_blob);
    delete idx;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
extern "C" void lrm_workspace_free(lrm_workspace *ws) {
    if (!ws) return;
    (void) hipSetDevice(ws->device);
    (void) hipFree(ws->d_reads2); (void) hipFree(ws->d_rec); (void) hipFree(ws->d_phase); (void) hipFree(ws->d_decided);
    (void) hipFree(ws->d_hcount); (void) hipFree(ws->d_counters); (void) hipFree(ws->d_recq); (void) hipFree(ws->d_cnt); (void) hipFree(ws->d_kc_key); (void) hipFree(ws->d_kc_ord); (void) hipFree(ws->d_redo); (void) hipFree(ws->d_big); (void) hipFree(ws->d_gtab); (void) hipFree(ws->d_glock);
    (void) hipFree(ws->d_qpl); (void) hipFree(ws->d_rflags);
    (void) hipFree(ws->d_ckpt); (void) hipFree(ws->d_codes); (void) hipFree(ws->d_ncodes);
    if (ws->h_err) (void) hipHostFree((void *) ws->h_err);
    for (int i = 0; i < LRM_MAX_TIMED; ++i) {
        if (ws->ev_start[i]) (void) hipEventDestroy((hipEvent_t) ws->ev_start[i]);
        if (ws->ev_stop[i]) (void) hipEventDestroy((hipEvent_t) ws->ev_stop[i]);
    }
    delete ws;
}

extern "C" uint64_t lrm_workspace_bytes(const lrm_workspace *ws) { return ws ? ws->bytes : 0; }

extern "C" int lrm_workspace_create(lrm_workspace **out, lrm_index *idx, uint64_t n_max, uint32_t max_len,
                                    uint32_t seed_len, uint32_t thres) {
    return lrm_workspace_create_parts(out, idx, n_max, max_len, seed_len, thres, LRM_WS_SEED | LRM_WS_EXTEND);
}

// parts: LRM_WS_SEED (packed reads, survivor lists, phase results), LRM_WS_EXTEND (planar reads, checkpoints, codes);
// the host pipeline seeds in small sub-batches and extends in larger groups, each with the scratch it needs
int lrm_workspace_create_parts(lrm_workspace **out, lrm_index *idx, uint64_t n_max, uint32_t max_len,
                               uint32_t seed_len, uint32_t thres, int parts) {
    if (!out || !idx) { lrm_set_error("null argument"); return -1; }
    if (seed_len < 1 || seed_len > 32) { lrm_set_error("seed_len %u outside [1,32]", seed_len); return -1; }
    if (thres >= (1u << 24)) { lrm_set_error("thres %u >= 2^24 unsupported", thres); return -1; }
    if (n_max == 0) n_max = 1;
    if (require_device(idx->device)) return -1;
    lrm_workspace *ws = new (std::nothrow) lrm_workspace;
    if (!ws) { lrm_set_error("out of memory"); return -1; }
    memset(ws, 0, sizeof(*ws));
    ws->idx = idx; ws->device = idx->device; ws->n_max = n_max; ws->max_len = max_len;
    ws->seed_len = seed_len; ws->thres = thres;
    ws->P = seed_len + 1;
    uint32_t jl = max_len > seed_len ? max_len - seed_len : 0;
    ws->cap_q = (jl + ws->P - 1) / ws->P;
    if (ws->cap_q == 0) ws->cap_q = 1;
    ws->words_per_read = (uint64_t) max_len / 32 + 2;
    ws->qpl_wpr = lrm_bs_planar_words(max_len);
    ws->codes_cw = lrm_bs_code_words(max_len);
    ws->parts = parts;
    {   // pool of global vote tables: a slice holds 2^k >= 2 x the most hits one (read, phase) item can have
        const uint64_t hmax = (uint64_t) ws->cap_q * (thres > 1 ? thres - 1 : 1);
        uint64_t gs = 1024;
        while (gs < 2 * hmax && gs < (1ull << 26)) gs <<= 1;
        uint64_t nsl = (256ull << 20) / (gs * 16);
        ws->g_slots = (uint32_t) gs;
        ws->g_slices = (uint32_t) (nsl < 2 ? 2 : nsl > 32 ? 32 : nsl);
    }
    struct { void **p; uint64_t bytes; int part; } allocs[] = {
        {(void **) &ws->d_reads2, n_max * ws->words_per_read * 8 + 128, LRM_WS_SEED},   // + slack: seed_search's scalar window loads reach 6 words

        {(void **) &ws->d_rec, n_max * (uint64_t) ws->P * ws->cap_q * 8, LRM_WS_SEED},
        {(void **) &ws->d_recq, n_max * (uint64_t) ws->P * ws->cap_q * 4, LRM_WS_SEED},
        {(void **) &ws->d_cnt, n_max * (uint64_t) ws->P * 4, LRM_WS_SEED},
        {(void **) &ws->d_kc_key, (uint64_t) LRM_VOTE_GRID * LRM_VOTE_KC_CAP * 8, LRM_WS_SEED},
        {(void **) &ws->d_kc_ord, (uint64_t) LRM_VOTE_GRID * LRM_VOTE_KC_CAP * 4, LRM_WS_SEED},
        {(void **) &ws->d_redo, n_max * (uint64_t) ws->P * 8, LRM_WS_SEED},
        {(void **) &ws->d_big, n_max * (uint64_t) ws->P * 8, LRM_WS_SEED},
        {(void **) &ws->d_gtab, (uint64_t) ws->g_slices * ws->g_slots * 16, LRM_WS_SEED},
        {(void **) &ws->d_glock, 64 * 4, LRM_WS_SEED},
        {(void **) &ws->d_phase, n_max * (uint64_t) ws->P * sizeof(LrmPhaseRes), LRM_WS_SEED},
        {(void **) &ws->d_decided, n_max, LRM_WS_SEED},
        {(void **) &ws->d_hcount, n_max * (uint64_t) ws->P * 4, LRM_WS_SEED},
        {(void **) &ws->d_counters, sizeof(LrmDevCounters), LRM_WS_SEED | LRM_WS_EXTEND},
        {(void **) &ws->d_qpl, n_max * ws->qpl_wpr * 8 + 16, LRM_WS_EXTEND},
        {(void **) &ws->d_rflags, n_max * 4, LRM_WS_EXTEND},
        {(void **) &ws->d_ckpt, lrm_bs_ckpt_words(n_max) * 4, LRM_WS_EXTEND},
        {(void **) &ws->d_codes, n_max * ws->codes_cw * 8, LRM_WS_EXTEND},
        {(void **) &ws->d_ncodes, n_max * 4, LRM_WS_EXTEND},
    };
    for (auto &a : allocs) {
        if (!(a.part & parts)) continue;
        if (hipMalloc(a.p, a.bytes) != hipSuccess) {
            lrm_set_error("hipMalloc of %llu workspace bytes failed", (unsigned long long) a.bytes);
            lrm_workspace_free(ws);
            return -1;
        }
        ws->bytes += a.bytes;
    }
    if (hipMemset(ws->d_counters, 0, sizeof(LrmDevCounters)) != hipSuccess) { lrm_workspace_free(ws); lrm_set_error("memset failed"); return -1; }
    if (ws->d_glock && hipMemset(ws->d_glock, 0, 64 * 4) != hipSuccess) { lrm_workspace_free(ws); lrm_set_error("memset failed"); return -1; }
    {   // error word: host-coherent pinned memory the kernels store to (never reset by a launch)
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            if (h) (void) hipHostFree(h);
            lrm_workspace_free(ws);
            lrm_set_error("allocation of the workspace error word failed");
            return -1;
        }
        memset(h, 0, 64);
        ws->h_err = (volatile uint32_t *) h;
        ws->d_err = (uint32_t *) d;
    }
    *out = ws;
    return 0;
}

// Reads and clears the sticky error word.  Kernels that raised it have completed only if the caller has
// synchronised with them; a later call sees the rest ("the next call after the faulty batch fails").
int lrm_ws_take_error(lrm_workspace *ws) {
    if (!ws || !ws->h_err) return 0;
    const uint32_t e = *ws->h_err;
    if (!e) return 0;
    *ws->h_err = 0;
    if (e & LRM_ERR_VOTE_OVERFLOW)
        lrm_set_error("vote table overflow in the multi-pass tier: results of some phases of an earlier batch on this workspace are invalid");
    else lrm_set_error("device error word 0x%x", e);
    return -2;
}

static int check_ws(lrm_workspace *ws, lrm_index *idx, uint64_t n, uint32_t max_len, uint32_t seed_len, uint32_t thres) {
    if (!ws || ws->idx != idx) { lrm_set_error("workspace does not belong to this index"); return -1; }
    if (n > ws->n_max || max_len > ws->max_len || seed_len != ws->seed_len || thres > ws->thres) {
        lrm_set_error("workspace too small: have n=%llu len=%u seed=%u thres=%u, need n=%llu len=%u seed=%u thres=%u",
                      (unsigned long long) ws->n_max, ws->max_len, ws->seed_len, ws->thres,
                      (unsigned long long) n, max_len, seed_len, thres);
        return -1;
    }
    return 0;
}

extern "C" int lrm_seed_batch_dev(lrm_index *idx, lrm_workspace *ws, const char *d_reads, uint64_t stride,
                                  const uint32_t *d_lens, uint64_t n, uint32_t max_len, lrm_params p,
                                  lrm_entry *d_best, void *stream) {
    if (!idx || !d_reads || !d_lens || !d_best) { lrm_set_error("null argument"); return -1; }
    if (check_ws(ws, idx, n, max_len, p.seed_len, p.thres)) return -1;
    if (!(ws->parts & LRM_WS_SEED)) { lrm_set_error("workspace has no seed-stage scratch"); return -1; }
    if (stride < max_len) { lrm_set_error("stride %llu < max_len %u", (unsigned long long) stride, max_len); return -1; }
    if (lrm_ws_take_error(ws)) return -2;
    HIPCHK(hipSetDevice(idx->device));
    return lrm_launch_seed(idx, ws, d_reads, stride, d_lens, n, max_len, p.seed_len, p.thres, d_best, idx->mtune, stream);
}

extern "C" int lrm_extend_batch_dev(lrm_index *idx, lrm_workspace *ws, char *d_reads, uint64_t stride,
                                    const uint32_t *d_lens, uint64_t n, uint32_t max_len, const lrm_entry *d_best,
                                    lrm_gact_params gp, uint8_t *d_store, uint64_t store_stride, int32_t *d_n_ops,
                                    int32_t *d_score, lrm_seq_meta *d_meta, int32_t *d_meta_r, void *stream) {
    if (!idx || !ws || !d_reads || !d_lens || !d_best || !d_store || !d_n_ops || !d_score || !d_meta || !d_meta_r) {
        lrm_set_error("null argument");
        return -1;
    }
    if (ws->idx != idx) { lrm_set_error("workspace does not belong to this index"); return -1; }
    if (lrm_ws_take_error(ws)) return -2;
    HIPCHK(hipSetDevice(idx->device));
    return lrm_launch_extend(idx, ws, d_reads, stride, d_lens, n, max_len, d_best, gp, d_store, store_stride,
                             d_n_ops, d_score, d_meta, d_meta_r, idx->mtune, stream);
}

extern "C" int lrm_workspace_stats(lrm_workspace *ws, lrm_stats *out, void *stream) {
    if (!ws || !out) { lrm_set_error("null argument"); return -1; }
    HIPCHK(hipSetDevice(ws->device));
    LrmDevCounters c;
    HIPCHK(hipMemcpyAsync(&c, ws->d_counters, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t) stream));
    HIPCHK(hipStreamSynchronize((hipStream_t) stream));
    {   // tier occupancy from the per-(read,phase) hit counts of the last seed call (host-side count)
        std::vector<uint32_t> hc((size_t) ws->n_last * ws->P);
        if (!hc.empty() && ws->d_hcount) HIPCHK(hipMemcpy(hc.data(), ws->d_hcount, hc.size() * 4, hipMemcpyDeviceToHost));
        else hc.clear();
        uint64_t t2 = 0, t3 = 0;
        for (uint32_t h : hc) { t2 += (h > LRM_VOTE_T1_LIMIT && h <= LRM_VOTE_T3_LIMIT); t3 += (h > LRM_VOTE_T3_LIMIT); }
        out->vote_tier2_items = t2;
        out->vote_tier3_items = t3;
    }
    out->reads_decided_phase0 = c.decided_phase0;
    out->gact_tiles = c.gact_tiles;
    out->vote_redo_items = c.vote_redo_n[0] + c.vote_redo_n[1];
    out->seeds_evaluated = c.reserved[3];
    out->seed_table_lookups = c.reserved[4];
    out->seed_rank_requests = c.reserved[5];
    return lrm_ws_take_error(ws);
}

// ------------------------------------------------------------------------------------------
// per-kernel timing
// ------------------------------------------------------------------------------------------
void lrm_time_begin(lrm_workspace *ws, int kernel, void *stream) {
    if (!ws || !ws->timing || ws->n_timed >= LRM_MAX_TIMED) return;
    int i = ws->n_timed;
    if (!ws->ev_start[i]) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ws->ev_start[i] = a; ws->ev_stop[i] = b;
    }
    ws->ev_kernel[i] = kernel;
    (void) hipEventRecord((hipEvent_t) ws->ev_start[i], (hipStream_t) stream);
}

void lrm_time_end(lrm_workspace *ws, void *stream) {
    if (!ws || !ws->timing || ws->n_timed >= LRM_MAX_TIMED || !ws->ev_stop[ws->n_timed]) return;
    (void) hipEventRecord((hipEvent_t) ws->ev_stop[ws->n_timed], (hipStream_t) stream);
    ws->n_timed++;
}

extern "C" int lrm_workspace_set_counting(lrm_workspace *ws, int enable) {
    if (!ws) { lrm_set_error("null argument"); return -1; }
    ws->counting = enable ? 1 : 0;
    return 0;
}

extern "C" int lrm_workspace_set_timing(lrm_workspace *ws, int enable) {
    if (!ws) { lrm_set_error("null argument"); return -1; }
    ws->timing = enable ? 1 : 0;
    ws->n_timed = 0;
    return 0;
}

extern "C" int lrm_workspace_timing(lrm_workspace *ws, double *ms, uint64_t *launches, void *stream) {
    if (!ws || !ms || !launches) { lrm_set_error("null argument"); return -1; }
    HIPCHK(hipSetDevice(ws->device));
    HIPCHK(hipStreamSynchronize((hipStream_t) stream));
    for (int i = 0; i < ws->n_timed; ++i) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, (hipEvent_t) ws->ev_start[i], (hipEvent_t) ws->ev_stop[i]));
        ms[ws->ev_kernel[i]] += (double) t;
        launches[ws->ev_kernel[i]] += 1;
    }
    ws->n_timed = 0;
    return 0;
}

extern "C" const char *lrm_kernel_name(int k) {
    static const char *names[LRM_K_COUNT] = {"pack2bit_kernel", "seed_search_kernel", "vote_kernel", "decide_kernel",
                                             "locus_resolve_kernel", "revcomp_kernel", "gact_kernel",
                                             "bs_pack_reads_kernel", "gact_bs_kernel"};
    return k >= 0 && k < LRM_K_COUNT ? names[k] : "?";
}

extern "C" void lrm_result_flags(const int *score, const int *meta_r, const lrm_seq_meta *meta, uint64_t n,
                                 int *flag_out, int *mapq_out, int *valid_out) {
    for (uint64_t i = 0; i < n; ++i) {                     // alnmain.c:460-474
        int flag = 0, mapq = 255, valid = score[i] >= 0;
        if (meta_r[i] == 0 || score[i] == -1) { valid = 0; flag += 0x4; mapq = 0; }
        else if (meta[i].strand == 1) flag += 16;
        flag_out[i] = flag; mapq_out[i] = mapq; valid_out[i] = valid;
    }
}

// ------------------------------------------------------------------------------------------
// debug taps (tests only)
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    int alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};

extern "C" int lrm_debug_seed_search(lrm_index *idx, const char *read, uint32_t len, uint32_t seed_len, uint32_t thres,
                                     int32_t *j_out, uint64_t *rr_out, uint64_t *k_out, uint64_t *l_out, uint64_t cap,
                                     uint64_t *n_out) {
    (void) thres;
    if (!idx || !read || !n_out) { lrm_set_error("null argument"); return -1; }
    if (seed_len < 1 || seed_len > 32) { lrm_set_error("seed_len %u outside [1,32]", seed_len); return -1; }
    if (require_device(idx->device)) return -1;
    uint64_t words = (uint64_t) len / 32 + 2;
    DevBuf d_read, d_r2, d_j, d_rr, d_k, d_l;
    if (d_read.alloc(len + 1) || d_r2.alloc((words + 1) * 8) || d_j.alloc(cap * 4) || d_rr.alloc(cap * 8) ||
        d_k.alloc(cap * 8) || d_l.alloc(cap * 8)) { lrm_set_error("device allocation failed"); return -1; }
    HIPCHK(hipMemcpy(d_read.p, read, len, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_j.p, 0xff, cap * 4));
    int cap_q = lrm_launch_debug_seed(idx, (const char *) d_read.p, len, seed_len, (uint64_t *) d_r2.p, words,
                                      (int32_t *) d_j.p, (uint64_t *) d_rr.p, (uint64_t *) d_k.p, (uint64_t *) d_l.p,
                                      cap, nullptr);
    if (cap_q < 0) return -1;
    HIPCHK(hipDeviceSynchronize());
    uint64_t total = (uint64_t) cap_q * (seed_len + 1);
    HIPCHK(hipMemcpy(j_out, d_j.p, total * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rr_out, d_rr.p, total * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(k_out, d_k.p, total * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(l_out, d_l.p, total * 8, hipMemcpyDeviceToHost));
    *n_out = total;
    return cap_q;
}
