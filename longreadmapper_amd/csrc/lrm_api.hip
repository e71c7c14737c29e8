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
extern "C" int lrm_abi_version(void) { return LRM_ABI_VERSION; }

extern "C" int lrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }
#define LRM_LCX_MAX 4096    // capacity of the long-interval side table

static void blob_layout(uint64_t length, int hlen, int mta_len, LrmBlobHeader *h) {
    memset(h, 0, sizeof(*h));
    h->magic = LRM_BLOB_MAGIC;
    h->version = LRM_ABI_VERSION;
    h->length = length;
    h->hlen = hlen;
    h->mta_len = mta_len;
    h->n_blocks = (length + LRM_OCC_ROWS - 1) / LRM_OCC_ROWS + 1;   // +1: rank(loc) may touch the block of L-1 only; spare block keeps gathers in bounds
    h->lc_entries = 1ull << (2 * hlen);
    h->sa_len = length;
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

extern "C" uint64_t lrm_index_blob_bytes(uint64_t length, int hlen, int mta_len) {
    LrmBlobHeader h;
    blob_layout(length, hlen, mta_len, &h);
    return h.total_bytes;
}

static inline int code_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

extern "C" int lrm_index_pack_blob(const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                   const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len,
                                   void *blob, uint64_t blob_bytes) {
    if (!fmi || !lch || !sa || !content || !blob) { lrm_set_error("null argument"); return -1; }
    const uint64_t L = fmi->length;
    if (L < 2 || L >= (1ull << 40)) { lrm_set_error("text length %llu outside [2, 2^40)", (unsigned long long) L); return -1; }
    if (con_len != L) { lrm_set_error("content length %llu != fm length %llu", (unsigned long long) con_len, (unsigned long long) L); return -1; }
    if (sa->len < L) { lrm_set_error("suffix array has %llu rows, need %llu", (unsigned long long) sa->len, (unsigned long long) L); return -1; }
    if (lch->hlen < 1 || lch->hlen > 15) { lrm_set_error("hlen %d outside [1,15] (lchash.c:75-77)", lch->hlen); return -1; }
    if (lch->len != 2ull << (2 * lch->hlen)) { lrm_set_error("lc table length %llu != 2*4^hlen", (unsigned long long) lch->len); return -1; }
    if (mta_len < 0 || (mta_len > 0 && !mta)) { lrm_set_error("bad mta"); return -1; }
    LrmBlobHeader h;
    blob_layout(L, lch->hlen, mta_len, &h);
    if (blob_bytes < h.total_bytes) { lrm_set_error("blob buffer too small"); return -1; }
    h.c4[0] = fmi->c[(unsigned char) 'A']; h.c4[1] = fmi->c[(unsigned char) 'C'];
    h.c4[2] = fmi->c[(unsigned char) 'G']; h.c4[3] = fmi->c[(unsigned char) 'T'];
    uint8_t *base = (uint8_t *) blob;
    memset(base, 0, sizeof(LrmBlobHeader));

    // occ blocks from the bwt; cross-checked against the reference's sampled O table
    LrmOccBlock *occ = (LrmOccBlock *) (base + h.off_occ);
    memset(occ, 0, h.n_blocks * sizeof(LrmOccBlock));
    uint64_t run[4] = {0, 0, 0, 0};
    uint64_t dollar = ~0ull;
    const uint64_t ratio = (uint64_t) fmi->o_ratio;
    for (uint64_t i = 0; i < L; ++i) {
        if ((i & (LRM_OCC_ROWS - 1)) == 0) {
            LrmOccBlock *b = &occ[i >> 6];
            for (int c = 0; c < 4; ++c) b->sym[c].cnt = run[c];
        }
        if (fmi->o && ratio > 0 && i % ratio == 0) {
            const uint64_t *o = fmi->o + 4 * (i / ratio);
            if (o[0] != run[0] || o[1] != run[1] || o[2] != run[2] || o[3] != run[3]) {
                lrm_set_error("O table disagrees with bwt at row %llu", (unsigned long long) i);
                return -1;
            }
        }
        char ch = fmi->bwt[i];
        int code = code_of(ch);
        if (code < 0) {
            if (ch == '$' && dollar == ~0ull) { dollar = i; }
            else { lrm_set_error("bwt row %llu holds byte 0x%02x (only upper-case ACGT and one '$' supported)", (unsigned long long) i, (unsigned) (unsigned char) ch); return -1; }
        } else {
            run[code]++;
            occ[i >> 6].sym[code].mask |= 1ull << (i & 63);
        }
    }
    if (dollar == ~0ull) { lrm_set_error("bwt has no '$' row"); return -1; }
    h.dollar_row = dollar;
    for (uint64_t b = (L + LRM_OCC_ROWS - 1) / LRM_OCC_ROWS; b < h.n_blocks; ++b)     // spare block(s): final counts
        for (int c = 0; c < 4; ++c) occ[b].sym[c].cnt = run[c];

    // lc table, permuted to the LSB-first code, 8 bytes per entry
    uint64_t *lc = (uint64_t *) (base + h.off_lc);
    const int hl = lch->hlen;
    const uint64_t ne = h.lc_entries;
    std::vector<uint64_t> over;
    uint64_t long_thr = 0xFFFFFFull;           // testing knob: send shorter intervals through the side table too
    if (const char *e = getenv("LRM_LCX_THRESHOLD")) { long_thr = strtoull(e, nullptr, 0); if (long_thr < 1 || long_thr > 0xFFFFFFull) long_thr = 0xFFFFFFull; }
#pragma omp parallel for schedule(static)
    for (uint64_t num = 0; num < ne; ++num) {
        uint64_t code = 0, t = num;
        for (int i = 0; i < hl; ++i) { code = (code << 2) | (t & 3); t >>= 2; }   // reverse the 2-bit groups
        const uint64_t k = lch->lc[2 * num], l = lch->lc[2 * num + 1];
        uint64_t e = 0;
        if (!(k == 0 && l == 0)) {
            uint64_t cnt = l >= k ? l - k + 1 : 0;
            if (cnt == 0 || cnt >= long_thr || k >= (1ull << 40)) {
                cnt = 0xFFFFFFull;
#pragma omp critical
                { over.push_back(code); over.push_back(k); over.push_back(l); }
            }
            e = (k & ((1ull << 40) - 1ull)) | (cnt << 40);
        }
        lc[code] = e;
    }
    if (over.size() / 3 > LRM_LCX_MAX) { lrm_set_error("too many long lchash intervals (%zu)", over.size() / 3); return -1; }
    {
        uint64_t *lcx = (uint64_t *) (base + h.off_lcx);
        memset(lcx, 0xff, LRM_LCX_MAX * 24);
        std::vector<size_t> ord(over.size() / 3);
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return over[3 * a] < over[3 * b]; });
        for (size_t i = 0; i < ord.size(); ++i)
            for (int f = 0; f < 3; ++f) lcx[3 * i + f] = over[3 * ord[i] + f];
        h.n_lcx = ord.size();
    }

    uint64_t *sav = (uint64_t *) (base + h.off_sa);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < L; ++i)
        sav[i] = ((uint64_t) sa->mem[i].high << 32) | (uint64_t) sa->mem[i].low;   // sa_use.h:27-29

    memcpy(base + h.off_content, content, L);
    base[h.off_content + L] = 0;
    LrmMtaDev *md = (LrmMtaDev *) (base + h.off_mta);
    for (int i = 0; i < mta_len; ++i) { md[i].offset = mta[i].offset; md[i].seq_len = (uint64_t) mta[i].seq_len; }
    memcpy(base, &h, sizeof(h));
    return 0;
}

static int make_handle(lrm_index **out, void *d_blob, uint64_t bytes, int device, int owns, const LrmBlobHeader &h) {
    if (h.magic != LRM_BLOB_MAGIC || h.version != LRM_ABI_VERSION) { lrm_set_error("not an lrm index image (magic/version)"); return -1; }
    if (h.total_bytes > bytes) { lrm_set_error("index image truncated"); return -1; }
    lrm_index *ix = new (std::nothrow) lrm_index;
    if (!ix) { lrm_set_error("out of memory"); return -1; }
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
    ix->view.lcl = nullptr; ix->view.hl = 0; ix->view.pad_ = 0; ix->d_lcl = nullptr;
    if (lrm_bs_prepare_index(ix)) { delete ix; return -1; }
    if (lrm_lcl_prepare_index(ix)) { lrm_bs_free_index(ix); delete ix; return -1; }
    *out = ix;
    return 0;
}

static int require_device(int device) {
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

extern "C" int lrm_index_upload_blob(lrm_index **out, const void *blob, uint64_t blob_bytes, int device) {
    if (!out || !blob || blob_bytes < sizeof(LrmBlobHeader)) { lrm_set_error("bad blob"); return -1; }
    if (require_device(device)) return -1;
    LrmBlobHeader h;
    memcpy(&h, blob, sizeof(h));
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, blob_bytes));
    if (hipMemcpy(d, blob, blob_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(d); lrm_set_error("index upload failed"); return -1; }
    if (make_handle(out, d, blob_bytes, device, 1, h)) { (void) hipFree(d); return -1; }
    return 0;
}

extern "C" int lrm_index_adopt_device(lrm_index **out, void *d_blob, uint64_t blob_bytes, int device) {
    if (!out || !d_blob || blob_bytes < sizeof(LrmBlobHeader)) { lrm_set_error("bad blob"); return -1; }
    if (require_device(device)) return -1;
    LrmBlobHeader h;
    HIPCHK(hipMemcpy(&h, d_blob, sizeof(h), hipMemcpyDeviceToHost));
    return make_handle(out, d_blob, blob_bytes, device, 0, h);
}

extern "C" int lrm_index_upload(lrm_index **out, const lrm_dna_fmi *fmi, const lrm_lc_hash *lch, const lrm_sa_mem *sa,
                                const char *content, uint64_t con_len, const lrm_mta_entry *mta, int mta_len, int device) {
    if (!out || !fmi || !lch) { lrm_set_error("null argument"); return -1; }
    if (require_device(device)) return -1;
    uint64_t bytes = lrm_index_blob_bytes(fmi->length, lch->hlen, mta_len);
    void *blob = malloc(bytes);
    if (!blob) { lrm_set_error("out of host memory for a %llu-byte index image", (unsigned long long) bytes); return -1; }
    int rc = lrm_index_pack_blob(fmi, lch, sa, content, con_len, mta, mta_len, blob, bytes);
    if (rc == 0) rc = lrm_index_upload_blob(out, blob, bytes, device);
    free(blob);
    return rc;
}

// Per-thread state of the host-buffer entry points: the workspace, device mirrors of the caller's arrays and two
// pinned staging chunks.  The caller's buffers are pageable (alnmain.c mallocs them): a plain hipMemcpy stages
// them through one driver thread at a fraction of the link rate, so the copies here run chunk-wise through
// pinned memory, the host-side half of each chunk being an OpenMP memcpy that overlaps the DMA of the previous one.
struct DevSlot {
    void *p = nullptr; uint64_t cap = 0;
    int ensure(uint64_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void) hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return -1; }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};
#define LRM_STAGE_CHUNK (32ull << 20)
struct DevSet { DevSlot reads, lens, best, store, nops, score, meta, mr; };
struct HostCache {
    lrm_workspace *ws;
    DevSet set[2];                       // a batch goes through the device in sub-batches, alternating between the sets
    void *pin_up[2], *pin_dn[2];         // pinned staging chunks per direction
    hipStream_t up, down, comp;          // upload DMA, download DMA, kernels
    hipEvent_t ev_pin_up[2], ev_pin_dn[2], ev_up[2], ev_done[2];
    int pin_up_used[2];
    uint64_t up_seq;
    int staging_ready;
};
static thread_local HostCache g_cache = {};

static int staging_init(HostCache &c) {
    if (c.staging_ready) return 0;
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc(&c.pin_up[b], LRM_STAGE_CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&c.pin_dn[b], LRM_STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) { lrm_set_error("pinned staging allocation failed"); return -1; }
        if (hipEventCreateWithFlags(&c.ev_pin_up[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_pin_dn[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_up[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_done[b], hipEventDisableTiming) != hipSuccess) { lrm_set_error("event creation failed"); return -1; }
        c.pin_up_used[b] = 0;
    }
    if (hipStreamCreateWithFlags(&c.up, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c.down, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c.comp, hipStreamNonBlocking) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    c.up_seq = 0;
    c.staging_ready = 1;
    return 0;
}
static void staging_release(HostCache &c) {
    if (c.staging_ready) {
        (void) hipDeviceSynchronize();
        for (int b = 0; b < 2; ++b) {
            (void) hipHostFree(c.pin_up[b]); (void) hipHostFree(c.pin_dn[b]);
            (void) hipEventDestroy(c.ev_pin_up[b]); (void) hipEventDestroy(c.ev_pin_dn[b]);
            (void) hipEventDestroy(c.ev_up[b]); (void) hipEventDestroy(c.ev_done[b]);
        }
        (void) hipStreamDestroy(c.up); (void) hipStreamDestroy(c.down); (void) hipStreamDestroy(c.comp);
        c.staging_ready = 0;
    }
    for (int b = 0; b < 2; ++b) {
        DevSet &d = c.set[b];
        d.reads.release(); d.lens.release(); d.best.release(); d.store.release();
        d.nops.release(); d.score.release(); d.meta.release(); d.mr.release();
    }
}
#define LRM_COPY_THREADS 8        // enough to outrun the link; a library must not fan out over every core of its host
static void par_memcpy(void *dst, const void *src, uint64_t bytes) {
    const uint64_t piece = 1ull << 20, np = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(static) num_threads(LRM_COPY_THREADS)
    for (uint64_t i = 0; i < np; ++i) {
        const uint64_t o = i * piece, l = bytes - o < piece ? bytes - o : piece;
        memcpy((char *) dst + o, (const char *) src + o, l);
    }
}
// host -> device through the pinned chunks, enqueued on the upload stream; returns when the last chunk has been
// handed to the DMA engine (not when it has landed: order later work behind the upload stream)
static int stage_h2d(HostCache &c, void *d_dst, const void *h_src, uint64_t bytes) {
    if (staging_init(c)) return -1;
    for (uint64_t o = 0; o < bytes; o += LRM_STAGE_CHUNK, ++c.up_seq) {
        const int b = (int) (c.up_seq & 1);
        const uint64_t l = bytes - o < LRM_STAGE_CHUNK ? bytes - o : LRM_STAGE_CHUNK;
        if (c.pin_up_used[b]) HIPCHK(hipEventSynchronize(c.ev_pin_up[b]));     // the chunk's previous DMA has drained
        par_memcpy(c.pin_up[b], (const char *) h_src + o, l);
        HIPCHK(hipMemcpyAsync((char *) d_dst + o, c.pin_up[b], l, hipMemcpyHostToDevice, c.up));
        HIPCHK(hipEventRecord(c.ev_pin_up[b], c.up));
        c.pin_up_used[b] = 1;
    }
    return 0;
}
// small host array -> device behind the staged upload (pageable source: the call returns once it is staged)
static int small_h2d(HostCache &c, void *d_dst, const void *h_src, uint64_t bytes) {
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c.up));
    return 0;
}
// device -> host, `rows` rows of `width` bytes (device pitch spitch, host pitch dpitch); rows == 1 is a flat copy
static int stage_d2h(HostCache &c, void *h_dst, uint64_t dpitch, const void *d_src, uint64_t spitch, uint64_t width,
                     uint64_t rows) {
    if (staging_init(c)) return -1;
    if (width == 0 || rows == 0) return 0;
    const bool flat = rows == 1;
    const uint64_t unit = flat ? LRM_STAGE_CHUNK : (LRM_STAGE_CHUNK / width ? LRM_STAGE_CHUNK / width : 0);
    if (unit == 0) { lrm_set_error("row wider than a staging chunk"); return -1; }
    const uint64_t total = flat ? width : rows;                      // bytes (flat) or rows
    uint64_t k = 0, o = 0, prev_o = 0, prev_l = 0;
    while (true) {
        const int b = (int) (k & 1);
        const uint64_t l = o < total ? (total - o < unit ? total - o : unit) : 0;
        if (l) {
            if (flat) HIPCHK(hipMemcpyAsync(c.pin_dn[b], (const char *) d_src + o, l, hipMemcpyDeviceToHost, c.down));
            else HIPCHK(hipMemcpy2DAsync(c.pin_dn[b], width, (const char *) d_src + o * spitch, spitch, width, l,
                                         hipMemcpyDeviceToHost, c.down));
            HIPCHK(hipEventRecord(c.ev_pin_dn[b], c.down));
        }
        if (prev_l) {                                                 // unpack the previous chunk while this one flies
            const int pb = (int) ((k - 1) & 1);
            HIPCHK(hipEventSynchronize(c.ev_pin_dn[pb]));
            if (flat) par_memcpy((char *) h_dst + prev_o, c.pin_dn[pb], prev_l);
            else {
#pragma omp parallel for schedule(static) num_threads(LRM_COPY_THREADS)
                for (uint64_t r = 0; r < prev_l; ++r)
                    memcpy((char *) h_dst + (prev_o + r) * dpitch, (const char *) c.pin_dn[pb] + r * width, width);
            }
        }
        if (l == 0) break;
        prev_o = o; prev_l = l; o += l; ++k;
    }
    return 0;
}

extern "C" void lrm_index_free(lrm_index *idx) {
    if (!idx) return;
    (void) hipSetDevice(idx->device);
    if (g_cache.ws && g_cache.ws->idx == idx) { lrm_workspace_free(g_cache.ws); g_cache.ws = nullptr; staging_release(g_cache); }
    lrm_bs_free_index(idx);
    if (idx->d_lcl) (void) hipFree(idx->d_lcl);
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
    (void) hipFree(ws->d_hcount); (void) hipFree(ws->d_counters);
    (void) hipFree(ws->d_qpl); (void) hipFree(ws->d_rflags);
    (void) hipFree(ws->d_ckpt); (void) hipFree(ws->d_codes); (void) hipFree(ws->d_ncodes);
    for (int i = 0; i < LRM_MAX_TIMED; ++i) {
        if (ws->ev_start[i]) (void) hipEventDestroy((hipEvent_t) ws->ev_start[i]);
        if (ws->ev_stop[i]) (void) hipEventDestroy((hipEvent_t) ws->ev_stop[i]);
    }
    delete ws;
}

extern "C" uint64_t lrm_workspace_bytes(const lrm_workspace *ws) { return ws ? ws->bytes : 0; }

extern "C" int lrm_workspace_create(lrm_workspace **out, lrm_index *idx, uint64_t n_max, uint32_t max_len,
                                    uint32_t seed_len, uint32_t thres) {
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
    struct { void **p; uint64_t bytes; } allocs[] = {
        {(void **) &ws->d_reads2, n_max * ws->words_per_read * 8},
        {(void **) &ws->d_rec, n_max * (uint64_t) ws->P * ws->cap_q * 8},
        {(void **) &ws->d_phase, n_max * (uint64_t) ws->P * sizeof(LrmPhaseRes)},
        {(void **) &ws->d_decided, n_max},
        {(void **) &ws->d_hcount, n_max * (uint64_t) ws->P * 4},
        {(void **) &ws->d_counters, sizeof(LrmDevCounters)},
        {(void **) &ws->d_qpl, n_max * ws->qpl_wpr * 8 + 16},
        {(void **) &ws->d_rflags, n_max * 4},
        {(void **) &ws->d_ckpt, lrm_bs_ckpt_words(n_max) * 4},
        {(void **) &ws->d_codes, n_max * ws->codes_cw * 8},
        {(void **) &ws->d_ncodes, n_max * 4},
    };
    for (auto &a : allocs) {
        if (hipMalloc(a.p, a.bytes) != hipSuccess) {
            lrm_set_error("hipMalloc of %llu workspace bytes failed", (unsigned long long) a.bytes);
            lrm_workspace_free(ws);
            return -1;
        }
        ws->bytes += a.bytes;
    }
    if (hipMemset(ws->d_counters, 0, sizeof(LrmDevCounters)) != hipSuccess) { lrm_workspace_free(ws); lrm_set_error("memset failed"); return -1; }
    *out = ws;
    return 0;
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
    if (stride < max_len) { lrm_set_error("stride %llu < max_len %u", (unsigned long long) stride, max_len); return -1; }
    HIPCHK(hipSetDevice(idx->device));
    return lrm_launch_seed(idx, ws, d_reads, stride, d_lens, n, max_len, p.seed_len, p.thres, d_best, stream);
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
    HIPCHK(hipSetDevice(idx->device));
    return lrm_launch_extend(idx, ws, d_reads, stride, d_lens, n, max_len, d_best, gp, d_store, store_stride,
                             d_n_ops, d_score, d_meta, d_meta_r, stream);
}

extern "C" int lrm_workspace_stats(lrm_workspace *ws, lrm_stats *out, void *stream) {
    if (!ws || !out) { lrm_set_error("null argument"); return -1; }
    HIPCHK(hipSetDevice(ws->device));
    LrmDevCounters c;
    HIPCHK(hipMemcpyAsync(&c, ws->d_counters, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t) stream));
    HIPCHK(hipStreamSynchronize((hipStream_t) stream));
    {   // tier occupancy from the per-(read,phase) hit counts of the last seed call (host-side count)
        std::vector<uint32_t> hc((size_t) ws->n_last * ws->P);
        if (!hc.empty()) HIPCHK(hipMemcpy(hc.data(), ws->d_hcount, hc.size() * 4, hipMemcpyDeviceToHost));
        uint64_t t2 = 0, t3 = 0;
        for (uint32_t h : hc) { t2 += (h > 192 && h <= 768); t3 += (h > 768); }
        out->vote_tier2_items = t2;
        out->vote_tier3_items = t3;
    }
    out->reads_decided_phase0 = c.decided_phase0;
    out->gact_tiles = c.gact_tiles;
    if (c.error_flags & 1ull) { lrm_set_error("vote table overflow in the multi-pass tier: results of some phases are invalid"); return -2; }
    return 0;
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
    static const char *names[LRM_K_COUNT] = {"pack2bit_kernel", "seed_search_kernel", "vote_wave_kernel",
                                             "vote_wave2_kernel", "decide_kernel", "locus_resolve_kernel",
                                             "revcomp_kernel", "gact_kernel", "vote_block_kernel",
                                             "bs_pack_reads_kernel", "gact_bs_kernel"};
    return k >= 0 && k < LRM_K_COUNT ? names[k] : "?";
}

// ------------------------------------------------------------------------------------------
// host-buffer entry points (drop-in boundary): stage through device buffers
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    int alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};

static int get_cached_ws(lrm_index *idx, uint64_t n, uint32_t max_len, uint32_t seed_len, uint32_t thres, lrm_workspace **out) {
    lrm_workspace *ws = g_cache.ws;
    if (ws && ws->idx == idx && n <= ws->n_max && max_len <= ws->max_len && seed_len == ws->seed_len && thres <= ws->thres) {
        *out = ws;
        return 0;
    }
    if (ws) { lrm_workspace_free(ws); g_cache.ws = nullptr; }
    if (lrm_workspace_create(&ws, idx, n, max_len, seed_len, thres)) return -1;
    g_cache.ws = ws;
    *out = ws;
    return 0;
}

static uint32_t max_of(const uint32_t *lens, uint64_t n) {
    uint32_t m = 0;
    for (uint64_t i = 0; i < n; ++i) m = lens[i] > m ? lens[i] : m;
    return m;
}

// Reads per device pass of the host-buffer entry points: the per-batch scratch is ~13 bytes per read base (seed
// records 8, op bytes 2, codes, packed copies), so very large caller batches (the reference's sweeps use up to
// 1 M reads, gen-sbatch-scripts.py:74) go through the device in slices of ~32 GB of scratch.  Results do not
// depend on the slicing: there is no cross-read state.
static uint64_t host_slice_reads(uint32_t max_len) {
    if (const char *e = getenv("LRM_HOST_SLICE")) { const long long v = atoll(e); if (v >= 1) return (uint64_t) v; }   // test knob
    const uint64_t per_read = 13ull * (max_len ? max_len : 1) + 4096;
    uint64_t r = (32ull << 30) / per_read;
    return r < 16384 ? 16384 : r;
}

// Sub-batches of one device pass: the upload of sub-batch k+1 (host memcpy into pinned chunks + DMA) and the
// download of sub-batch k-1 run while the kernels of sub-batch k execute on their own stream.
#define LRM_PIPE_MIN_READS 16384
static uint64_t pipe_subs(uint64_t n) {
    if (const char *e = getenv("LRM_HOST_SUBS")) { const long long v = atoll(e); if (v >= 1) return (uint64_t) v < n ? (uint64_t) v : n; }   // test knob
    const uint64_t k = n / LRM_PIPE_MIN_READS;
    return k < 2 ? 1 : (k > 4 ? 4 : k);
}

static int seed_slice(lrm_index *idx, const char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                      uint32_t max_len, lrm_params p, lrm_entry *best_out) {
    HostCache &hc = g_cache;
    if (staging_init(hc)) return -1;
    const uint64_t nsub = pipe_subs(n), sub = (n + nsub - 1) / nsub;
    lrm_workspace *ws;
    if (get_cached_ws(idx, sub, max_len, p.seed_len, p.thres, &ws)) return -1;
    uint64_t prev_off = 0, prev_m = 0;
    for (uint64_t k = 0, off = 0; off < n; ++k, off += sub) {
        const int b = (int) (k & 1);
        const uint64_t m = n - off < sub ? n - off : sub;
        DevSet &d = hc.set[b];
        if (d.reads.ensure(m * stride) || d.lens.ensure(m * 4) || d.best.ensure(m * sizeof(lrm_entry))) { lrm_set_error("device allocation failed"); return -1; }
        if (stage_h2d(hc, d.reads.p, reads_buf + off * stride, m * stride)) return -1;
        if (small_h2d(hc, d.lens.p, lens + off, m * 4)) return -1;
        HIPCHK(hipEventRecord(hc.ev_up[b], hc.up));
        HIPCHK(hipStreamWaitEvent(hc.comp, hc.ev_up[b], 0));
        if (lrm_launch_seed(idx, ws, (const char *) d.reads.p, stride, (const uint32_t *) d.lens.p, m, max_len, p.seed_len,
                            p.thres, (lrm_entry *) d.best.p, hc.comp)) return -1;
        HIPCHK(hipEventRecord(hc.ev_done[b], hc.comp));
        if (prev_m) {                                              // results of the previous sub-batch
            const int pb = (int) ((k - 1) & 1);
            HIPCHK(hipEventSynchronize(hc.ev_done[pb]));
            HIPCHK(hipMemcpy(best_out + prev_off, hc.set[pb].best.p, prev_m * sizeof(lrm_entry), hipMemcpyDeviceToHost));
        }
        prev_off = off; prev_m = m;
    }
    {
        const int pb = (int) (((n + sub - 1) / sub - 1) & 1);
        HIPCHK(hipEventSynchronize(hc.ev_done[pb]));
        HIPCHK(hipMemcpy(best_out + prev_off, hc.set[pb].best.p, prev_m * sizeof(lrm_entry), hipMemcpyDeviceToHost));
    }
    lrm_stats st;
    if (lrm_workspace_stats(ws, &st, hc.comp)) return -1;
    return 0;
}

extern "C" int lrm_seed_batch(lrm_index *idx, const char *reads_buf, uint64_t stride, const uint32_t *lens,
                              uint64_t n, lrm_params p, lrm_entry *best_out) {
    if (!idx || !reads_buf || !lens || !best_out) { lrm_set_error("null argument"); return -1; }
    if (n == 0) return 0;
    if (require_device(idx->device)) return -1;
    const uint32_t max_len = max_of(lens, n);
    if (stride < max_len) { lrm_set_error("stride < longest read"); return -1; }
    const uint64_t slice = host_slice_reads(max_len);
    for (uint64_t o = 0; o < n; o += slice) {
        const uint64_t m = n - o < slice ? n - o : slice;
        if (seed_slice(idx, reads_buf + o * stride, stride, lens + o, m, max_len, p, best_out + o)) return -1;
    }
    return 0;
}

static int extend_collect(HostCache &hc, int b, uint64_t m, char *reads_buf, uint64_t stride, lrm_cigar *cig_out,
                          uint8_t *store_mem, uint64_t store_stride, uint64_t dstride, int *score_out,
                          lrm_seq_meta *meta_out, int *meta_r_out) {
    DevSet &d = hc.set[b];
    HIPCHK(hipEventSynchronize(hc.ev_done[b]));
    std::vector<int32_t> nops(m);
    HIPCHK(hipMemcpy(nops.data(), d.nops.p, m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(score_out, d.score.p, m * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(meta_out, d.meta.p, m * sizeof(lrm_seq_meta), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(meta_r_out, d.mr.p, m * 4, hipMemcpyDeviceToHost));
    if (stage_d2h(hc, reads_buf, 0, d.reads.p, 0, m * stride, 1)) return -1;          // rev-comped reads travel back
    int32_t mx = 0;                                                  // only the columns some read uses cross the link
    for (uint64_t i = 0; i < m; ++i) mx = nops[i] > mx ? nops[i] : mx;
    uint64_t width = ((uint64_t) mx + 63) & ~63ull;
    if (width > store_stride) width = store_stride;
    if (stage_d2h(hc, store_mem, store_stride, d.store.p, dstride, width, m)) return -1;
    for (uint64_t i = 0; i < m; ++i) {                               // alnmain.c:322-325, mutils.c:99-104
        cig_out[i].cigar = store_mem + i * store_stride;
        cig_out[i].n_cigar_op = nops[i];
        cig_out[i].score = score_out[i];
    }
    return 0;
}

static int extend_slice(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                        uint32_t max_len, const lrm_entry *best, lrm_gact_params gp, lrm_cigar *cig_out,
                        uint8_t *store_mem, uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out,
                        int *meta_r_out) {
    HostCache &hc = g_cache;
    if (staging_init(hc)) return -1;
    const uint64_t nsub = pipe_subs(n), sub = (n + nsub - 1) / nsub;
    lrm_workspace *ws = hc.ws;
    if (!ws || ws->idx != idx) {
        if (get_cached_ws(idx, sub, max_len, 20, 300, &ws)) return -1;   // extend only needs the counters block and the GACT scratch
    } else if (sub > ws->n_max || max_len > ws->max_len) {
        const uint32_t sl = ws->seed_len, th = ws->thres;
        if (get_cached_ws(idx, sub, max_len, sl, th, &ws)) return -1;
    }
    const uint64_t dstride = (store_stride + 3) & ~3ull;     // the bit-sliced kernel stores CIGAR bytes four at a time
    uint64_t prev_off = 0, prev_m = 0;
    uint64_t k = 0;
    for (uint64_t off = 0; off < n; ++k, off += sub) {
        const int b = (int) (k & 1);
        const uint64_t m = n - off < sub ? n - off : sub;
        DevSet &d = hc.set[b];
        if (d.reads.ensure(m * stride) || d.lens.ensure(m * 4) || d.best.ensure(m * sizeof(lrm_entry)) ||
            d.store.ensure(m * dstride) || d.nops.ensure(m * 4) || d.score.ensure(m * 4) ||
            d.meta.ensure(m * sizeof(lrm_seq_meta)) || d.mr.ensure(m * 4)) { lrm_set_error("device allocation failed"); return -1; }
        if (stage_h2d(hc, d.reads.p, reads_buf + off * stride, m * stride)) return -1;
        if (small_h2d(hc, d.lens.p, lens + off, m * 4)) return -1;
        if (small_h2d(hc, d.best.p, best + off, m * sizeof(lrm_entry))) return -1;
        HIPCHK(hipEventRecord(hc.ev_up[b], hc.up));
        HIPCHK(hipStreamWaitEvent(hc.comp, hc.ev_up[b], 0));
        if (lrm_launch_extend(idx, ws, (char *) d.reads.p, stride, (const uint32_t *) d.lens.p, m, max_len,
                              (const lrm_entry *) d.best.p, gp, (uint8_t *) d.store.p, dstride, (int32_t *) d.nops.p,
                              (int32_t *) d.score.p, (lrm_seq_meta *) d.meta.p, (int32_t *) d.mr.p, hc.comp)) return -1;
        HIPCHK(hipEventRecord(hc.ev_done[b], hc.comp));
        if (prev_m && extend_collect(hc, (int) ((k - 1) & 1), prev_m, reads_buf + prev_off * stride, stride,
                                     cig_out + prev_off, store_mem + prev_off * store_stride, store_stride, dstride,
                                     score_out + prev_off, meta_out + prev_off, meta_r_out + prev_off)) return -1;
        prev_off = off; prev_m = m;
    }
    return extend_collect(hc, (int) ((k - 1) & 1), prev_m, reads_buf + prev_off * stride, stride, cig_out + prev_off,
                          store_mem + prev_off * store_stride, store_stride, dstride, score_out + prev_off,
                          meta_out + prev_off, meta_r_out + prev_off);
}

extern "C" int lrm_extend_batch(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                                const lrm_entry *best, lrm_gact_params gp, lrm_cigar *cig_out, uint8_t *store_mem,
                                uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out, int *meta_r_out) {
    if (!idx || !reads_buf || !lens || !best || !cig_out || !store_mem || !score_out || !meta_out || !meta_r_out) {
        lrm_set_error("null argument");
        return -1;
    }
    if (n == 0) return 0;
    if (require_device(idx->device)) return -1;
    const uint32_t max_len = max_of(lens, n);
    if (stride < max_len) { lrm_set_error("stride < longest read"); return -1; }
    const uint64_t slice = host_slice_reads(max_len);
    for (uint64_t o = 0; o < n; o += slice) {
        const uint64_t m = n - o < slice ? n - o : slice;
        if (extend_slice(idx, reads_buf + o * stride, stride, lens + o, m, max_len, best + o, gp, cig_out + o,
                         store_mem + o * store_stride, store_stride, score_out + o, meta_out + o, meta_r_out + o)) return -1;
    }
    return 0;
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
