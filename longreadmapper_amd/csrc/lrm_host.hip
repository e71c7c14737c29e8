// lrm_host.hip -- the host-buffer entry points of liblrm_accel.so (the drop-in boundary):
//   lrm_map_batch     PART 1 + PART 2 of single_end() for one batch in ONE device pass (alnmain.c:333-451)
//   lrm_seed_batch    PART 1 alone (alnmain.c:333-405)
//   lrm_extend_batch  PART 2 alone (alnmain.c:408-451)
// on one device or on a multi-GPU group handle (lrm_index_upload_multi: reads partitioned by bases, one host
// thread per replica, every replica writes its slice of the caller's arrays in place -- SURVEY 8(b)/(e); the
// host loop this replaces is alnmain.c:302-330).
//
// How a slice of a batch goes through a device: its arrays are mirrored whole in HBM; the reads are uploaded and
// SEEDED in sub-batches (three seed streams, round robin: the first kernels start a few milliseconds into the
// call and the uploads hide behind them), the EXTENSION runs over groups of sub-batches on its own stream as soon
// as their seeds are done, two groups at a time on two streams (the bit-sliced kernel carries one read per lane;
// the memory-latency-bound seed kernels of later sub-batches overlap its VALU-bound work), and a second host
// thread downloads every group while the next one is extended.  Results leave the device DENSE: a pack kernel
// gathers the used part of every CIGAR row and the reverse-complemented reads (the only rows of reads_buf that
// changed) into one contiguous buffer, which crosses the link at the full DMA rate (a strided hipMemcpy2D of the
// same rows measured 6 GB/s against 57 GB/s flat) and is scattered into the caller's rows by a multi-threaded
// memcpy.  Reads that are pinned (lrm_host_alloc / lrm_host_register) are uploaded by the DMA engines directly;
// pageable ones (what alnmain.c mallocs) are staged chunk-wise through pinned memory, the host half of every chunk
// overlapping the DMA of the previous chunk.
// No CPU fallback: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "lrm_internal.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

namespace {

// LRM_HOST_VERBOSE=1: stage times of the host-buffer pipeline on stderr (tuning aid)
struct HostClock {
    bool on = getenv("LRM_HOST_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

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
constexpr uint64_t STAGE_CHUNK = 32ull << 20;
constexpr int N_SEED_STREAMS = 3;
constexpr int N_EXT_STREAMS = 4;        // upper bound; run_slice uses n_ext_streams of them (LRM_HOST_EXT_STREAMS)
constexpr int COPY_THREADS = 8;       // enough to outrun the link; a library must not fan out over every core of its host
constexpr int N_DOWN = 2;             // download lanes available (run_slice uses one unless LRM_HOST_DOWN=2)
struct DevSet { DevSlot reads, lens, best, store, nops, score, meta, mr; };
// one download lane: its stream, its pair of pinned chunks, its dense result buffer + offset table on the device
struct DownLane {
    hipStream_t st = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    DevSlot dense, offs;
};

}  // namespace

struct LrmHostCtx {
    std::mutex mu;                       // one host-buffer call at a time per replica (re-entrant per handle otherwise)
    lrm_workspace *ws_seed[N_SEED_STREAMS] = {};   // seed-stage scratch, one per seed stream (sub-batch sized)
    lrm_workspace *ws_ext[N_EXT_STREAMS] = {};     // extension scratch (group sized), one per extension stream
    DevSet dev;                                    // device mirrors of the caller's arrays for one slice
    void *pin_up[2] = {nullptr, nullptr};
    DownLane dn[N_DOWN];
    std::mutex err_mu;                   // the lanes poll the workspaces' sticky error words
    hipStream_t up = nullptr, seed[N_SEED_STREAMS] = {}, ext[N_EXT_STREAMS] = {};
    hipEvent_t ev_pin_up[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_up, ev_seed, ev_ext;   // per sub-batch / per extension group, grown on demand
    bool pin_up_used[2] = {false, false};
    uint64_t up_seq = 0;
    bool ready = false;
};

namespace {

int ctx_init(LrmHostCtx &c) {
    if (c.ready) return 0;
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc(&c.pin_up[b], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) { lrm_set_error("pinned staging allocation failed"); return -1; }
        if (hipEventCreateWithFlags(&c.ev_pin_up[b], hipEventDisableTiming) != hipSuccess) { lrm_set_error("event creation failed"); return -1; }
        for (int l = 0; l < N_DOWN; ++l) {
            if (hipHostMalloc(&c.dn[l].pin[b], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) { lrm_set_error("pinned staging allocation failed"); return -1; }
            if (hipEventCreateWithFlags(&c.dn[l].ev[b], hipEventDisableTiming) != hipSuccess) { lrm_set_error("event creation failed"); return -1; }
        }
    }
    // Priorities: the result path first (pack kernels + downloads), then the extension of a finished group, then
    // the seed kernels of later sub-batches -- otherwise every group's extension finishes at the very end, behind
    // all the seed work, and the downloads of all but the first group run after the compute instead of under it.
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // numerically lower = higher priority
    const int p_seed = prio_lo, p_ext = prio_hi < prio_lo ? prio_lo - 1 : prio_lo, p_down = prio_hi;
    if (hipStreamCreateWithPriority(&c.up, hipStreamNonBlocking, p_down) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    for (int l = 0; l < N_DOWN; ++l)
        if (hipStreamCreateWithPriority(&c.dn[l].st, hipStreamNonBlocking, p_down) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    for (int s = 0; s < N_EXT_STREAMS; ++s)
        if (hipStreamCreateWithPriority(&c.ext[s], hipStreamNonBlocking, p_ext) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    for (int s = 0; s < N_SEED_STREAMS; ++s)
        if (hipStreamCreateWithPriority(&c.seed[s], hipStreamNonBlocking, p_seed) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    c.ready = true;
    return 0;
}

int ensure_events(std::vector<hipEvent_t> &v, size_t n) {
    while (v.size() < n) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { lrm_set_error("event creation failed"); return -1; }
        v.push_back(e);
    }
    return 0;
}

void par_memcpy(void *dst, const void *src, uint64_t bytes) {
    const uint64_t piece = 1ull << 20, np = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(static) num_threads(COPY_THREADS)
    for (uint64_t i = 0; i < np; ++i) {
        const uint64_t o = i * piece, l = bytes - o < piece ? bytes - o : piece;
        memcpy((char *) dst + o, (const char *) src + o, l);
    }
}

// pinned (hipHostMalloc / hipHostRegister) memory can be handed to the DMA engines as it is, and kernels can
// write it through its device alias (*dev_alias)
bool is_pinned(const void *p, void **dev_alias = nullptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void) hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    if (dev_alias) *dev_alias = a.devicePointer;
    return true;
}

// host -> device on the upload stream; returns when the last byte has been handed to the DMA engine (not when
// it has landed: later work is ordered behind the upload stream)
int h2d(LrmHostCtx &c, void *d_dst, const void *h_src, uint64_t bytes, bool pinned) {
    if (bytes == 0) return 0;
    if (pinned) { HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c.up)); return 0; }
    for (uint64_t o = 0; o < bytes; o += STAGE_CHUNK, ++c.up_seq) {
        const int b = (int) (c.up_seq & 1);
        const uint64_t l = bytes - o < STAGE_CHUNK ? bytes - o : STAGE_CHUNK;
        if (c.pin_up_used[b]) HIPCHK(hipEventSynchronize(c.ev_pin_up[b]));     // the chunk's previous DMA has drained
        par_memcpy(c.pin_up[b], (const char *) h_src + o, l);
        HIPCHK(hipMemcpyAsync((char *) d_dst + o, c.pin_up[b], l, hipMemcpyHostToDevice, c.up));
        HIPCHK(hipEventRecord(c.ev_pin_up[b], c.up));
        c.pin_up_used[b] = true;
    }
    return 0;
}

// pack kernel: row i of a pitched device array (len[i] bytes; 0 = skip) -> dense[off[i] ..), 16 bytes per lane.
// Rows start at any byte (the hardware takes the unaligned dwords); dense offsets are 16-byte aligned.
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *__restrict__ src, uint64_t pitch,
                                                        const uint32_t *__restrict__ len, const uint64_t *__restrict__ off,
                                                        uint8_t *__restrict__ dense, uint64_t rows) {
    const uint64_t row = blockIdx.x;
    if (row >= rows) return;
    const uint32_t l = len[row];
    const uint8_t *s = src + row * pitch;
    uint8_t *d = dense + off[row];
    for (uint32_t o = (blockIdx.y * 256 + threadIdx.x) * 16; o < l; o += gridDim.y * 256 * 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (o + 16 <= l) __builtin_memcpy(w, s + o, 16);
        else for (uint32_t e = 0; o + e < l; ++e) w[e >> 2] |= (uint32_t) s[o + e] << (8 * (e & 3));
        *reinterpret_cast<uint4 *>(d + o) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// rows of a pitched device array -> rows of another pitched array (len[i] bytes of row i; 0 = skip).  The
// destination may be the device alias of PINNED HOST memory: the stores then cross the link as posted writes,
// 1 KiB per wavefront instruction, and the caller's rows are filled with no staging copy and no host work.
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint8_t *__restrict__ src, uint64_t spitch,
                                                        uint8_t *__restrict__ dst, uint64_t dpitch,
                                                        const uint32_t *__restrict__ len, uint64_t rows) {
    const uint64_t row = blockIdx.x;
    if (row >= rows) return;
    const uint32_t l = len[row];
    const uint8_t *s = src + row * spitch;
    uint8_t *d = dst + row * dpitch;
    const uint32_t head = (uint32_t) ((16u - ((uintptr_t) d & 15u)) & 15u);         // bytes up to the first aligned 16
    if (blockIdx.y == 0) for (uint32_t o = threadIdx.x; o < head && o < l; o += 256) d[o] = s[o];
    for (uint32_t o = head + (blockIdx.y * 256 + threadIdx.x) * 16; o < l; o += gridDim.y * 256 * 16) {
        if (o + 16 <= l) {
            uint32_t w[4];
            __builtin_memcpy(w, s + o, 16);
            *reinterpret_cast<uint4 *>(d + o) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint32_t e = 0; o + e < l; ++e) d[o + e] = s[o + e];
        }
    }
}

// dense device buffer -> rows of the caller's arrays: contiguous DMA through the lane's pinned chunks, every chunk
// scattered into the caller's rows by COPY_THREADS threads while the next one flies.
// off[i] (16-byte aligned, ascending) / len[i]: position and length of entry i in the dense buffer; dst[i]: where
// its bytes go.
int d2h_dense(DownLane &L, const uint8_t *d_dense, uint64_t total, const uint64_t *off, const uint32_t *len,
              uint8_t *const *dst, uint64_t rows) {
    if (total == 0) return 0;
    uint64_t k = 0, o = 0, prev_o = 0, prev_l = 0;
    uint64_t row_lo = 0;                                              // first entry that may still have bytes at or after prev_o
    while (true) {
        const int b = (int) (k & 1);
        const uint64_t l = o < total ? (total - o < STAGE_CHUNK ? total - o : STAGE_CHUNK) : 0;
        if (l) {
            HIPCHK(hipMemcpyAsync(L.pin[b], d_dense + o, l, hipMemcpyDeviceToHost, L.st));
            HIPCHK(hipEventRecord(L.ev[b], L.st));
        }
        if (prev_l) {                                                 // scatter the previous chunk while this one flies
            const int pb = (int) ((k - 1) & 1);
            HIPCHK(hipEventSynchronize(L.ev[pb]));
            const uint8_t *chunk = (const uint8_t *) L.pin[pb];
            const uint64_t c0 = prev_o, c1 = prev_o + prev_l;
            while (row_lo < rows && off[row_lo] + len[row_lo] <= c0) ++row_lo;
            uint64_t row_hi = row_lo;
            while (row_hi < rows && off[row_hi] < c1) ++row_hi;
#pragma omp parallel for schedule(static) num_threads(COPY_THREADS)
            for (uint64_t r = row_lo; r < row_hi; ++r) {
                const uint64_t a = off[r] > c0 ? off[r] : c0, e = off[r] + len[r] < c1 ? off[r] + len[r] : c1;
                if (e > a) memcpy(dst[r] + (a - off[r]), chunk + (a - c0), e - a);
            }
        }
        if (l == 0) break;
        prev_o = o; prev_l = l; o += l; ++k;
    }
    return 0;
}

uint32_t max_of(const uint32_t *lens, uint64_t n) {
    uint32_t m = 0;
    for (uint64_t i = 0; i < n; ++i) m = lens[i] > m ? lens[i] : m;
    return m;
}

// Reads per device pass: the per-batch scratch is ~13 bytes per read base (seed records, op bytes, codes, packed
// copies), so very large caller batches (the reference's sweeps use up to 1 M reads, gen-sbatch-scripts.py:74) go
// through the device in slices of ~32 GB of scratch.  Results do not depend on the slicing: there is no
// cross-read state.
uint64_t host_slice_reads(uint32_t max_len) {
    if (const char *e = getenv("LRM_HOST_SLICE")) { const long long v = atoll(e); if (v >= 1) return (uint64_t) v; }   // test knob
    const uint64_t per_read = 13ull * (max_len ? max_len : 1) + 4096;
    uint64_t r = (32ull << 30) / per_read;
    return r < 16384 ? 16384 : r;
}

// Seed sub-batches of one device pass (see the header comment): small enough that the first kernels start a few
// milliseconds after the call and the uploads hide behind them.
constexpr uint64_t PIPE_MIN_READS = 8192;
uint64_t pipe_subs(uint64_t n) {
    if (const char *e = getenv("LRM_HOST_SUBS")) { const long long v = atoll(e); if (v >= 1) return (uint64_t) v < n ? (uint64_t) v : n; }   // test knob
    const uint64_t k = n / PIPE_MIN_READS;
    return k < 2 ? 1 : (k > 12 ? 12 : k);
}
// Sub-batches per extension group: the bit-sliced kernel carries one read per LANE, so it wants >= 32 k reads
// per launch for decent SIMD coverage; two groups are in extension at once (two streams).  Measured per 100 k-read
// batch: groups of 17 k reads 69 ms, 25 k 73 ms, 33 k 78 ms.
constexpr uint64_t EXT_GROUP_READS = 16384;
uint64_t ext_group_subs(uint64_t sub, uint64_t nsub) {
    if (const char *e = getenv("LRM_HOST_GROUP")) { const long long v = atoll(e); if (v >= 1) return (uint64_t) v < nsub ? (uint64_t) v : nsub; }   // test knob
    const uint64_t g = (EXT_GROUP_READS + sub - 1) / (sub ? sub : 1);
    return g < 1 ? 1 : (g > nsub ? nsub : g);
}

enum { DO_SEED = 1, DO_EXTEND = 2 };
struct MapJob {
    int mode;
    char *reads; uint64_t stride; const uint32_t *lens; uint64_t n;
    lrm_params p; lrm_gact_params gp;
    const lrm_entry *best_in; lrm_entry *best_out;
    lrm_cigar *cig; uint8_t *store_mem; uint64_t store_stride; int *score; lrm_seq_meta *meta; int *meta_r;
    MapJob slice(uint64_t o, uint64_t m) const {
        MapJob j = *this;
        j.reads = reads + o * stride; j.lens = lens + o; j.n = m;
        if (best_in) j.best_in = best_in + o;
        if (best_out) j.best_out = best_out + o;
        if (cig) { j.cig = cig + o; j.store_mem = store_mem + o * store_stride; j.score = score + o; j.meta = meta + o; j.meta_r = meta_r + o; }
        return j;
    }
};

int get_ws(lrm_workspace *&ws, lrm_index *idx, uint64_t n, uint32_t max_len, uint32_t seed_len, uint32_t thres, int parts) {
    if (ws && n <= ws->n_max && max_len <= ws->max_len && (!(parts & LRM_WS_SEED) || (seed_len == ws->seed_len && thres <= ws->thres))) return 0;
    if (ws) { lrm_workspace_free(ws); ws = nullptr; }
    return lrm_workspace_create_parts(&ws, idx, n, max_len, seed_len, thres, parts);
}

// hand-off between the issuing thread and the download thread
struct Pipe {
    std::mutex m;
    std::condition_variable cv;
    uint64_t issued = 0;              // units (extension groups, or seed sub-batches in seed-only mode) handed to the device
    bool stop = false;
    int rc = 0;
    char err[512] = "";
    void fail(int code) {
        std::lock_guard<std::mutex> g(m);
        if (!rc) { rc = code; snprintf(err, sizeof(err), "%s", lrm_last_error()); }
        cv.notify_all();
    }
};

struct Range { uint64_t off, m; };

int take_errors(LrmHostCtx &c) {
    std::lock_guard<std::mutex> g(c.err_mu);
    int rc = 0;
    for (int s = 0; s < N_SEED_STREAMS; ++s) if (lrm_ws_take_error(c.ws_seed[s])) rc = -2;
    for (int s = 0; s < N_EXT_STREAMS; ++s) if (lrm_ws_take_error(c.ws_ext[s])) rc = -2;
    return rc;
}

// download of one unit [off, off + m) of the slice (runs on the download thread once `done` has fired)
int collect(LrmHostCtx &c, DownLane &L, const MapJob &j, const Range &u, hipEvent_t done, uint64_t dstride, const HostClock &clk) {
    DevSet &d = c.dev;
    const double t_in = clk.ms();
    HIPCHK(hipEventSynchronize(done));
    const double t_done = clk.ms();
    if (take_errors(c)) return -2;                                   // raised by this or an earlier unit: never lost
    const uint64_t m = u.m, o = u.off;
    if (j.mode & DO_SEED) HIPCHK(hipMemcpyAsync(j.best_out + o, (const lrm_entry *) d.best.p + o, m * sizeof(lrm_entry), hipMemcpyDeviceToHost, L.st));
    if (!(j.mode & DO_EXTEND)) { HIPCHK(hipStreamSynchronize(L.st)); return 0; }
    std::vector<int32_t> nops(m);
    HIPCHK(hipMemcpyAsync(nops.data(), (const int32_t *) d.nops.p + o, m * 4, hipMemcpyDeviceToHost, L.st));
    HIPCHK(hipMemcpyAsync(j.score + o, (const int32_t *) d.score.p + o, m * 4, hipMemcpyDeviceToHost, L.st));
    HIPCHK(hipMemcpyAsync(j.meta + o, (const lrm_seq_meta *) d.meta.p + o, m * sizeof(lrm_seq_meta), hipMemcpyDeviceToHost, L.st));
    HIPCHK(hipMemcpyAsync(j.meta_r + o, (const int32_t *) d.mr.p + o, m * 4, hipMemcpyDeviceToHost, L.st));
    HIPCHK(hipStreamSynchronize(L.st));
    // dense layout: the used part of every CIGAR row, then the reads that were reverse-complemented in place
    // (alnmain.c:437; the other rows of reads_buf did not change)
    std::vector<uint64_t> off(2 * m);
    std::vector<uint32_t> len(2 * m);
    uint64_t total = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t cap = j.store_stride;
        len[i] = nops[i] > 0 ? (uint32_t) ((uint64_t) nops[i] < cap ? (uint64_t) nops[i] : cap) : 0u;
        off[i] = total;
        total += ((uint64_t) len[i] + 15) & ~15ull;
    }
    for (uint64_t i = 0; i < m; ++i) {
        const bool rev = j.meta_r[o + i] != 0 && j.meta[o + i].strand == 1;
        len[m + i] = rev ? j.lens[o + i] : 0u;
        off[m + i] = total;
        total += ((uint64_t) len[m + i] + 15) & ~15ull;
    }
    const uint8_t *d_store = (const uint8_t *) d.store.p + o * dstride, *d_reads = (const uint8_t *) d.reads.p + o * j.stride;
    void *store_alias = nullptr, *reads_alias = nullptr;
    const bool direct = getenv("LRM_HOST_DIRECT") != nullptr &&
                        is_pinned(j.store_mem + o * j.store_stride, &store_alias) && store_alias &&
                        is_pinned(j.reads + o * j.stride, &reads_alias) && reads_alias;
    if (total) {
        if (L.offs.ensure(2 * m * 12)) { lrm_set_error("device allocation failed"); return -1; }
        uint64_t *d_off = (uint64_t *) L.offs.p;
        uint32_t *d_len = (uint32_t *) ((uint8_t *) L.offs.p + 2 * m * 8);
        HIPCHK(hipMemcpyAsync(d_len, len.data(), 2 * m * 4, hipMemcpyHostToDevice, L.st));
        const uint32_t gy_ops = (uint32_t) ((j.store_stride + 4095) / 4096), gy_rd = (uint32_t) ((j.stride + 4095) / 4096);
        if (direct) {
            // LRM_HOST_DIRECT=1 with pinned caller buffers: the device writes the rows straight into them as posted
            // writes.  Measured slower than the dense DMA + host scatter on this platform (23 GB/s against 57), so off
            // by default.
            hipLaunchKernelGGL(copy_rows_kernel, dim3((uint32_t) m, gy_ops ? gy_ops : 1), dim3(256), 0, L.st, d_store, dstride,
                               (uint8_t *) store_alias, j.store_stride, d_len, m);
            hipLaunchKernelGGL(copy_rows_kernel, dim3((uint32_t) m, gy_rd ? gy_rd : 1), dim3(256), 0, L.st, d_reads, j.stride,
                               (uint8_t *) reads_alias, j.stride, d_len + m, m);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(L.st));
        } else {
            if (L.dense.ensure(total)) { lrm_set_error("device allocation failed"); return -1; }
            HIPCHK(hipMemcpyAsync(d_off, off.data(), 2 * m * 8, hipMemcpyHostToDevice, L.st));
            hipLaunchKernelGGL(pack_rows_kernel, dim3((uint32_t) m, gy_ops ? gy_ops : 1), dim3(256), 0, L.st, d_store, dstride,
                               d_len, d_off, (uint8_t *) L.dense.p, m);
            hipLaunchKernelGGL(pack_rows_kernel, dim3((uint32_t) m, gy_rd ? gy_rd : 1), dim3(256), 0, L.st, d_reads, j.stride,
                               d_len + m, d_off + m, (uint8_t *) L.dense.p, m);
            HIPCHK(hipGetLastError());
            std::vector<uint8_t *> dst(2 * m);
            for (uint64_t i = 0; i < m; ++i) {
                dst[i] = j.store_mem + (o + i) * j.store_stride;
                dst[m + i] = (uint8_t *) j.reads + (o + i) * j.stride;
            }
            if (d2h_dense(L, (const uint8_t *) L.dense.p, total, off.data(), len.data(), dst.data(), 2 * m)) return -1;
        }
    }
    for (uint64_t i = 0; i < m; ++i) {                               // alnmain.c:322-325, mutils.c:99-104
        j.cig[o + i].cigar = j.store_mem + (o + i) * j.store_stride;
        j.cig[o + i].n_cigar_op = nops[i];
        j.cig[o + i].score = j.score[o + i];
    }
    if (clk.on) fprintf(stderr, "[lrm host] collect off=%llu m=%llu: wait-from %.1f kernels-done %.1f collected %.1f ms (%s, %.0f MB)\n",
                        (unsigned long long) o, (unsigned long long) m, t_in, t_done, clk.ms(), direct ? "direct" : "dense", total / 1e6);
    return 0;
}

// One device pass over a slice of the job.  The slice's arrays are mirrored whole on the device; the reads are
// uploaded and SEEDED in sub-batches (seed streams, round robin), the EXTENSION runs over groups of sub-batches on
// its own stream as soon as their seeds are done, and the download thread collects every group while the next one
// is still being extended.
int run_slice(lrm_index *idx, LrmHostCtx &c, const MapJob &j, uint32_t max_len) {
    const uint64_t n = j.n, nsub = pipe_subs(n), sub = (n + nsub - 1) / nsub;
    const uint64_t dstride = (j.store_stride + 3) & ~3ull;           // the bit-sliced kernel stores CIGAR bytes four at a time
    const bool pin_reads = is_pinned(j.reads);
    std::vector<Range> subs, units;
    for (uint64_t off = 0; off < n; off += sub) subs.push_back({off, n - off < sub ? n - off : sub});
    const uint64_t gsub = (j.mode & DO_EXTEND) ? ext_group_subs(sub, subs.size()) : 1;
    // unit boundaries (in sub-batches).  (Cutting the last group once more, so that less is left to download after
    // the last kernel, was measured: no gain -- the call is bound by its kernels.)
    std::vector<size_t> ends;
    for (size_t k = gsub; k < subs.size(); k += gsub) ends.push_back(k);
    ends.push_back(subs.size());
    std::vector<size_t> unit_of(subs.size());
    for (size_t g = 0, k0 = 0; g < ends.size(); k0 = ends[g], ++g) {
        units.push_back({subs[k0].off, subs[ends[g] - 1].off + subs[ends[g] - 1].m - subs[k0].off});
        for (size_t k = k0; k < ends[g]; ++k) unit_of[k] = g;
    }
    // groups in extension at once.  The lane-per-read kernel has a fixed latency per group (a lane walks its read's
    // tiles one after the other: ~7.6 ms for 10 kbp), and a 17 k-read group fills a quarter of the SIMDs.
    int n_ext_streams = 2;
    if (const char *e = getenv("LRM_HOST_EXT_STREAMS")) { const int v = atoi(e); if (v >= 1 && v <= N_EXT_STREAMS) n_ext_streams = v; }   // tuning knob
    uint64_t unit_max = 0;
    for (auto &u : units) unit_max = u.m > unit_max ? u.m : unit_max;
    if (n > 0x7fffffffull) { lrm_set_error("batch too large"); return -1; }
    if (j.mode & DO_SEED)
        for (int s = 0; s < N_SEED_STREAMS && (size_t) s < subs.size(); ++s)
            if (get_ws(c.ws_seed[s], idx, sub, max_len, j.p.seed_len, j.p.thres, LRM_WS_SEED)) return -1;
    if (j.mode & DO_EXTEND)
        for (int s = 0; s < n_ext_streams && (size_t) s < units.size(); ++s)
            if (get_ws(c.ws_ext[s], idx, unit_max, max_len, 20, 300, LRM_WS_EXTEND)) return -1;
    if (ensure_events(c.ev_up, subs.size()) || ensure_events(c.ev_seed, subs.size()) || ensure_events(c.ev_ext, units.size())) return -1;
    DevSet &d = c.dev;
    if (d.reads.ensure(n * j.stride) || d.lens.ensure(n * 4) || d.best.ensure(n * sizeof(lrm_entry))) { lrm_set_error("device allocation failed"); return -1; }
    // (the dense result buffer and its offset table at their worst-case size for a unit, so that the download thread
    //  never reallocates -- a hipFree would drain the whole device -- while other units are in flight)
    if ((j.mode & DO_EXTEND) && (d.store.ensure(n * dstride) || d.nops.ensure(n * 4) || d.score.ensure(n * 4) ||
                                  d.meta.ensure(n * sizeof(lrm_seq_meta)) || d.mr.ensure(n * 4) ||
                                  false)) { lrm_set_error("device allocation failed"); return -1; }
    int n_down = 1;
    if (const char *e = getenv("LRM_HOST_DOWN")) { const int v = atoi(e); if (v >= 1 && v <= N_DOWN && (size_t) v <= units.size()) n_down = v; }   // tuning knob
    if (j.mode & DO_EXTEND)
        for (int l = 0; l < n_down; ++l)
            if (c.dn[l].dense.ensure(unit_max * (dstride + j.stride + 32)) || c.dn[l].offs.ensure(unit_max * 2 * 12)) { lrm_set_error("device allocation failed"); return -1; }

    Pipe pipe;
    HostClock clk;
    const int device = idx->device;
    const bool seed_only = !(j.mode & DO_EXTEND);
    // Download lanes: unit g is collected by lane g % n_down.  One lane by default.  A second lane was meant to hide
    // the fixed costs of a unit (small copies, offset table, pack kernels, the first chunk's flight and the last
    // chunk's scatter: ~1.7 of 6.1 ms per 256 MB unit) behind the other lane's DMA; measured on a 16-core host it
    // is slower (pinned 71 vs 70 ms per 100 k x 10 kbp batch, pageable 99 vs 73: two scatter teams plus the upload
    // staging oversubscribe the cores, and the two DMA queues share one link), so LRM_HOST_DOWN=2 is opt-in.
    // Likewise three or four extension streams instead of two (LRM_HOST_EXT_STREAMS): 72-74 ms against 70-71.
    auto lane = [&](int ln) {
        if (hipSetDevice(device) != hipSuccess) { lrm_set_error("hipSetDevice failed on the download thread"); pipe.fail(-1); return; }
        for (uint64_t g = (uint64_t) ln; g < units.size(); g += (uint64_t) n_down) {
            {
                std::unique_lock<std::mutex> lk(pipe.m);
                pipe.cv.wait(lk, [&] { return pipe.issued > g || pipe.stop || pipe.rc; });
                if (pipe.rc || pipe.issued <= g) return;
            }
            // seed-only: a unit is done when the seeds of its last sub-batch are (sub-batches of a seed stream are ordered)
            hipEvent_t done = seed_only ? c.ev_seed[ends[g] - 1] : c.ev_ext[g];
            int rc;
            try { rc = collect(c, c.dn[ln], j, units[g], done, dstride, clk); }
            catch (const std::exception &e) { lrm_set_error("download thread: %s", e.what()); rc = -1; }
            if (rc) { pipe.fail(rc); return; }
        }
    };
    std::vector<std::thread> downloaders;
    for (int l = 0; l < n_down; ++l) downloaders.emplace_back(lane, l);

    int rc = 0;
    int n_seed_streams = 2;
    if (const char *e = getenv("LRM_HOST_SEED_STREAMS")) { const int v = atoi(e); if (v >= 1 && v <= N_SEED_STREAMS) n_seed_streams = v; }   // tuning knob
    for (uint64_t k = 0; k < subs.size() && !rc; ++k) {
        const int s = (int) (k % (uint64_t) n_seed_streams);
        const uint64_t m = subs[k].m, off = subs[k].off;
        { std::lock_guard<std::mutex> lk(pipe.m); if (pipe.rc) break; }
        auto issue = [&]() -> int {
            char *dr = (char *) d.reads.p + off * j.stride;
            if (h2d(c, dr, j.reads + off * j.stride, m * j.stride, pin_reads)) return -1;
            HIPCHK(hipMemcpyAsync((uint32_t *) d.lens.p + off, j.lens + off, m * 4, hipMemcpyHostToDevice, c.up));
            if (!(j.mode & DO_SEED)) HIPCHK(hipMemcpyAsync((lrm_entry *) d.best.p + off, j.best_in + off, m * sizeof(lrm_entry), hipMemcpyHostToDevice, c.up));
            HIPCHK(hipEventRecord(c.ev_up[k], c.up));
            if (j.mode & DO_SEED) {
                HIPCHK(hipStreamWaitEvent(c.seed[s], c.ev_up[k], 0));
                if (lrm_launch_seed(idx, c.ws_seed[s], dr, j.stride, (const uint32_t *) d.lens.p + off, m, max_len, j.p.seed_len, j.p.thres,
                                    (lrm_entry *) d.best.p + off, c.seed[s])) return -1;
                HIPCHK(hipEventRecord(c.ev_seed[k], c.seed[s]));
            }
            const uint64_t g = unit_of[k];
            const bool closes = k + 1 == ends[g];
            if (closes && (j.mode & DO_EXTEND)) {                                  // the group's extension, behind its seeds / uploads
                const int xs = (int) (g % (uint64_t) n_ext_streams);
                for (uint64_t x = g ? ends[g - 1] : 0; x <= k; ++x) HIPCHK(hipStreamWaitEvent(c.ext[xs], (j.mode & DO_SEED) ? c.ev_seed[x] : c.ev_up[x], 0));
                const Range &u = units[g];
                if (lrm_launch_extend(idx, c.ws_ext[xs], (char *) d.reads.p + u.off * j.stride, j.stride, (const uint32_t *) d.lens.p + u.off, u.m,
                                      max_len, (const lrm_entry *) d.best.p + u.off, j.gp, (uint8_t *) d.store.p + u.off * dstride, dstride,
                                      (int32_t *) d.nops.p + u.off, (int32_t *) d.score.p + u.off, (lrm_seq_meta *) d.meta.p + u.off,
                                      (int32_t *) d.mr.p + u.off, c.ext[xs])) return -1;
                HIPCHK(hipEventRecord(c.ev_ext[g], c.ext[xs]));
            }
            if (closes) {
                { std::lock_guard<std::mutex> lk(pipe.m); pipe.issued = g + 1; }
                pipe.cv.notify_all();
            }
            return 0;
        };
        const double t_i0 = clk.ms();
        rc = issue();
        if (clk.on) fprintf(stderr, "[lrm host] issue   off=%llu m=%llu: %.1f -> %.1f ms\n", (unsigned long long) off, (unsigned long long) m, t_i0, clk.ms());
        if (rc) { pipe.fail(rc); break; }
    }
    { std::lock_guard<std::mutex> lk(pipe.m); pipe.stop = true; }
    pipe.cv.notify_all();
    for (auto &t : downloaders) t.join();
    if (clk.on) fprintf(stderr, "[lrm host] slice of %llu reads, %zu seed sub-batches, %zu units: %.1f ms\n", (unsigned long long) n, subs.size(), units.size(), clk.ms());
    if (pipe.rc) {
        for (int s = 0; s < N_SEED_STREAMS; ++s) (void) hipStreamSynchronize(c.seed[s]);
        for (int s = 0; s < N_EXT_STREAMS; ++s) (void) hipStreamSynchronize(c.ext[s]);
        (void) hipStreamSynchronize(c.up);
        for (int l = 0; l < N_DOWN; ++l) (void) hipStreamSynchronize(c.dn[l].st);
        (void) take_errors(c);                                        // reported now: do not fail the next call
        lrm_set_error("%s", pipe.err);
        return pipe.rc;
    }
    // the device mirrors are reused by the next slice: everything must have drained (it has: every unit was collected)
    return 0;
}

std::mutex g_host_ctx_init;             // creation of a handle's host context (the context's own mutex lives inside it)

int run_replica(lrm_index *idx, const MapJob &j) {
    if (lrm_require_device(idx->device)) return -1;
    {
        std::lock_guard<std::mutex> g(g_host_ctx_init);
        if (!idx->host) {
            idx->host = new (std::nothrow) LrmHostCtx;
            if (!idx->host) { lrm_set_error("out of memory"); return -1; }
        }
    }
    LrmHostCtx &c = *idx->host;
    std::lock_guard<std::mutex> g(c.mu);
    if (ctx_init(c)) return -1;
    const uint32_t max_len = max_of(j.lens, j.n);
    if (j.stride < max_len) { lrm_set_error("stride < longest read"); return -1; }
    if ((j.mode & DO_EXTEND) && j.store_stride < 2ull * max_len) { lrm_set_error("store_stride < 2 * longest read (alnmain.c:316-320)"); return -1; }
    const uint64_t slice = host_slice_reads(max_len);
    for (uint64_t o = 0; o < j.n; o += slice) {
        const MapJob s = j.slice(o, j.n - o < slice ? j.n - o : slice);
        const int rc = run_slice(idx, c, s, max_len);
        if (rc) return rc;
    }
    return 0;
}

// contiguous slices balanced by cumulative bases, not by read count (100 kbp reads next to 1 kbp ones)
void partition_by_bases(const uint32_t *lens, uint64_t n, int parts, std::vector<uint64_t> &cuts) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) total += lens[i];
    cuts.assign((size_t) parts + 1, n);
    cuts[0] = 0;
    uint64_t acc = 0, i = 0;
    for (int r = 1; r < parts; ++r) {
        const uint64_t target = (uint64_t) ((__uint128_t) total * (uint64_t) r / (uint64_t) parts);
        while (i < n && acc < target) acc += lens[i++];
        cuts[r] = i;
    }
}

int run_job_impl(lrm_index *idx, const MapJob &j);
// C ABI: no C++ exception may leave the library (allocation failures of the host-side bookkeeping, thread creation)
int run_job(lrm_index *idx, const MapJob &j) {
    try {
        return run_job_impl(idx, j);
    } catch (const std::exception &e) {
        lrm_set_error("host-side failure: %s", e.what());
        return -1;
    } catch (...) {
        lrm_set_error("host-side failure");
        return -1;
    }
}

int run_job_impl(lrm_index *idx, const MapJob &j) {
    if (j.n == 0) return 0;
    if (idx->n_peers <= 1 || !idx->peers) return run_replica(idx, j);
    const int np = idx->n_peers;
    std::vector<uint64_t> cuts;
    partition_by_bases(j.lens, j.n, np, cuts);
    std::vector<int> rcs((size_t) np, 0);
    std::vector<std::string> errs((size_t) np);
    std::vector<std::thread> th;
    for (int r = 0; r < np; ++r) {
        th.emplace_back([&, r]() {                                   // one host thread per replica / device
            const uint64_t lo = cuts[r], hi = cuts[r + 1];
            if (hi <= lo) return;
            rcs[r] = run_replica(idx->peers[r], j.slice(lo, hi - lo));
            if (rcs[r]) errs[r] = lrm_last_error();
        });
    }
    for (auto &t : th) t.join();
    for (int r = 0; r < np; ++r)
        if (rcs[r]) { lrm_set_error("replica %d (device %d): %s", r, idx->peers[r]->device, errs[r].c_str()); return rcs[r]; }
    return 0;
}

}  // namespace

void lrm_host_ctx_free(lrm_index *idx) {
    LrmHostCtx *c = idx->host;
    if (!c) return;
    idx->host = nullptr;
    if (c->ready) {
        for (int s = 0; s < N_SEED_STREAMS; ++s) (void) hipStreamSynchronize(c->seed[s]);
        for (int s = 0; s < N_EXT_STREAMS; ++s) (void) hipStreamSynchronize(c->ext[s]);
        (void) hipStreamSynchronize(c->up);
        for (int l = 0; l < N_DOWN; ++l) (void) hipStreamSynchronize(c->dn[l].st);
        for (int b = 0; b < 2; ++b) {
            (void) hipHostFree(c->pin_up[b]);
            (void) hipEventDestroy(c->ev_pin_up[b]);
            for (int l = 0; l < N_DOWN; ++l) { (void) hipHostFree(c->dn[l].pin[b]); (void) hipEventDestroy(c->dn[l].ev[b]); }
        }
        (void) hipStreamDestroy(c->up);
        for (int l = 0; l < N_DOWN; ++l) (void) hipStreamDestroy(c->dn[l].st);
        for (int s = 0; s < N_EXT_STREAMS; ++s) (void) hipStreamDestroy(c->ext[s]);
        for (int s = 0; s < N_SEED_STREAMS; ++s) (void) hipStreamDestroy(c->seed[s]);
    }
    for (auto *v : {&c->ev_up, &c->ev_seed, &c->ev_ext}) for (hipEvent_t e : *v) (void) hipEventDestroy(e);
    for (int s = 0; s < N_SEED_STREAMS; ++s) if (c->ws_seed[s]) lrm_workspace_free(c->ws_seed[s]);
    for (int s = 0; s < N_EXT_STREAMS; ++s) if (c->ws_ext[s]) lrm_workspace_free(c->ws_ext[s]);
    DevSet &d = c->dev;
    d.reads.release(); d.lens.release(); d.best.release(); d.store.release();
    d.nops.release(); d.score.release(); d.meta.release(); d.mr.release();
    for (int l = 0; l < N_DOWN; ++l) { c->dn[l].dense.release(); c->dn[l].offs.release(); }
    delete c;
}

extern "C" int lrm_seed_batch(lrm_index *idx, const char *reads_buf, uint64_t stride, const uint32_t *lens,
                              uint64_t n, lrm_params p, lrm_entry *best_out) {
    if (!idx || !reads_buf || !lens || !best_out) { lrm_set_error("null argument"); return -1; }
    MapJob j = {};
    j.mode = DO_SEED; j.reads = const_cast<char *>(reads_buf); j.stride = stride; j.lens = lens; j.n = n; j.p = p; j.best_out = best_out;
    return run_job(idx, j);
}

extern "C" int lrm_extend_batch(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                                const lrm_entry *best, lrm_gact_params gp, lrm_cigar *cig_out, uint8_t *store_mem,
                                uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out, int *meta_r_out) {
    if (!idx || !reads_buf || !lens || !best || !cig_out || !store_mem || !score_out || !meta_out || !meta_r_out) {
        lrm_set_error("null argument");
        return -1;
    }
    MapJob j = {};
    j.mode = DO_EXTEND; j.reads = reads_buf; j.stride = stride; j.lens = lens; j.n = n; j.gp = gp; j.best_in = best;
    j.p.seed_len = 20; j.p.thres = 300;                              // only sizes the workspace when none is cached yet
    j.cig = cig_out; j.store_mem = store_mem; j.store_stride = store_stride; j.score = score_out; j.meta = meta_out; j.meta_r = meta_r_out;
    return run_job(idx, j);
}

extern "C" int lrm_map_batch(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                             lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out, uint8_t *store_mem,
                             uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out, int *meta_r_out) {
    if (!idx || !reads_buf || !lens || !best_out || !cig_out || !store_mem || !score_out || !meta_out || !meta_r_out) {
        lrm_set_error("null argument");
        return -1;
    }
    MapJob j = {};
    j.mode = DO_SEED | DO_EXTEND; j.reads = reads_buf; j.stride = stride; j.lens = lens; j.n = n; j.p = p; j.gp = gp;
    j.best_out = best_out;
    j.cig = cig_out; j.store_mem = store_mem; j.store_stride = store_stride; j.score = score_out; j.meta = meta_out; j.meta_r = meta_r_out;
    return run_job(idx, j);
}

// Pinned host memory for the caller's batch buffers (reads_buf, store_mem): the DMA engines read and write it
// directly, no staging copy.  lrm_host_register pins memory the caller already owns (malloc'd at alnmain.c:297-320).
extern "C" void *lrm_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError(); lrm_set_error("hipHostMalloc of %llu bytes failed", (unsigned long long) bytes); return nullptr; }
    return p;
}
extern "C" void lrm_host_free(void *p) { if (p) (void) hipHostFree(p); }
extern "C" int lrm_host_register(void *p, uint64_t bytes) {
    if (!p || !bytes) { lrm_set_error("bad argument"); return -1; }
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return 0;
}
extern "C" int lrm_host_unregister(void *p) {
    if (!p) return 0;
    HIPCHK(hipHostUnregister(p));
    return 0;
}
