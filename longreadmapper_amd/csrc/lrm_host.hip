// lrm_host.hip -- the host-buffer entry points of liblrm_accel.so (the drop-in boundary):
//   lrm_map_batch          PART 1 + PART 2 of single_end() for one batch in ONE device pass (alnmain.c:333-451)
//   lrm_map_batch_submit   the same, asynchronous: up to two batches per device in flight (the batch loop
//   lrm_map_batch_wait       alnmain.c:302-330 with the copy clauses the reference planned at :420-424)
//   lrm_seed_batch         PART 1 alone (alnmain.c:333-405)
//   lrm_extend_batch       PART 2 alone (alnmain.c:408-451)
// on one device or on a multi-GPU group handle (reads partitioned by bases, every replica writes its slice of the
// caller's arrays in place -- SURVEY 8(b)/(e)).
//
// Every replica owns TWO long-lived host threads and two SLOTS of device resources (mirrors of the caller's arrays,
// workspaces, dense result buffers):
//   issuer     takes the next slice of a submitted batch, waits for a free slot, and hands the whole slice to the
//              device without waiting for anything: the reads are uploaded and SEEDED in sub-batches (two seed streams),
//              the EXTENSION runs over groups of sub-batches on two extension streams as soon as their seeds are done;
//   collector  follows the extension groups in order: small result arrays, then the op bytes and the
//              reverse-complemented reads (the only rows of reads_buf that changed, alnmain.c:437).
// With two slots the upload and the seeds of batch k+1 run under the extension tail and the result download of
// batch k -- the serial chain that bounds a single call.  Stream priorities: results > extension > seeds.
//
// How results reach the caller (lrm_map_options): always as a DENSE image packed on the device (the used part of every
// CIGAR row, the reverse-complemented reads) that crosses the link by DMA at its full rate -- a strided hipMemcpy2D of
// the same rows does 6 GB/s, and a kernel writing the caller's pinned memory itself collapses to 2-9 GB/s as soon as
// compute kernels own the chip (tools/d2h_under_load.hip).
//   dense_results   the op bytes stay dense: ONE DMA per group straight into the caller's (pinned) store_mem, and
//                   cig[i].cigar points into it (the convention of mutils.c:97-103 kept).  The reverse-complemented
//                   reads are the only rows left to place: through a ring of pinned chunks, by the collector alone.
//   rows (default)  cig[i].cigar = store_mem + i*store_stride as in alnmain.c:322-325: the whole image comes down
//                   through the ring and a small memcpy team scatters it.
//
// Host CPU: every wait for the device is a sleep-poll on an event (hipEventSynchronize and hipStreamSynchronize spin
// a core for the whole wait on this platform, blocking-sync events included: tools/hostlink_bench.hip).
// No CPU fallback: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "lrm_internal.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

// completion of one submitted batch: `pending` slices (over all replicas) still to be collected
struct lrm_ticket {
    std::mutex m;
    std::condition_variable cv;
    int pending = 0;
    int rc = 0;
    std::string err;
    void part_done(int code, const std::string &msg) {
        std::lock_guard<std::mutex> g(m);
        if (code && !rc) { rc = code; err = msg; }
        --pending;
        cv.notify_all();
    }
};

namespace {

struct HostClock {
    bool on = false;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct DevSlot {
    void *p = nullptr; uint64_t cap = 0;
    int ensure(uint64_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void) hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void) hipGetLastError(); p = nullptr; return -1; }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};
struct PinSlot {
    void *p = nullptr; uint64_t cap = 0;
    int ensure(uint64_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void) hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); p = nullptr; return -1; }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void) hipHostFree(p); p = nullptr; cap = 0; }
};

constexpr uint64_t STAGE_CHUNK = 32ull << 20;      // pinned chunks of the pageable upload staging
constexpr uint64_t RING_CHUNK = 16ull << 20;       // ... and of the download ring
constexpr int N_RING = 16;          // 256 MiB of pinned chunks per replica: a unit's reverse-complemented reads (0.25 GB) fit whole
constexpr int N_SEED_STREAMS = 3;
constexpr int N_EXT_STREAMS = 4;
constexpr int N_SLOTS = 3;            // upper bound on the batches (slices) in flight per replica; n_slots of them are used
struct DevSet { DevSlot reads, lens, best, store, nops, score, meta, mr, tlen; };

// device-side resources of one slice in flight
struct Slot {
    lrm_workspace *ws_seed[N_SEED_STREAMS] = {};   // seed-stage scratch, one per seed stream (sub-batch sized)
    lrm_workspace *ws_ext[N_EXT_STREAMS] = {};     // extension scratch (group sized), one per extension stream
    DevSet dev;                                    // device mirrors of the caller's arrays
    DevSlot dense[2], offs[2];                     // dense result image + offset table, alternating over the groups
    hipEvent_t ev_dense[2] = {nullptr, nullptr};   // the last transfer out of dense[b] has drained
    bool dense_used[2] = {false, false};
    PinSlot h_small;                               // pinned staging of the small result arrays and offset tables (per read)
    std::vector<hipEvent_t> ev_up, ev_seed, ev_ext;   // per sub-batch / per extension group, grown on demand
    bool busy = false;
};

// Waits for an event WITHOUT spinning: hipEventSynchronize / hipStreamSynchronize burn a core for the whole wait
// (measured, also for hipEventBlockingSync events), and 8 replicas x 2 threads of that is the host's whole CPU share.
int wait_event(hipEvent_t ev) {
    useconds_t nap = 20;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return 0;
        (void) hipGetLastError();
        if (e != hipErrorNotReady) { lrm_set_error("hipEventQuery failed: %s", hipGetErrorString(e)); return -1; }
        usleep(nap);
        if (nap < 200) nap += 20;
    }
}

}  // namespace

struct MapJob;
struct SliceJob;

struct LrmHostCtx {
    lrm_index *idx = nullptr;
    int copy_threads = 4;                // memcpy team of the pageable paths (staging upload, result scatter)
    int n_slots = 2;                     // slices in flight (LRM_HOST_SLOTS)
    // --- queues (mu) ---
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<SliceJob>> q_issue, q_collect;
    bool stop = false;
    int n_active = 0;                    // slices queued or in flight (lrm_host_ctx_free drains them)
    Slot slots[N_SLOTS];
    std::thread issuer, collector;
    bool threads_up = false;
    // --- device objects ---
    hipStream_t up = nullptr, down = nullptr, seed[N_SEED_STREAMS] = {}, ext[N_EXT_STREAMS] = {};
    // issuer only: staging of pageable uploads
    void *pin_up[2] = {nullptr, nullptr};
    hipEvent_t ev_pin_up[2] = {nullptr, nullptr};
    bool pin_up_used[2] = {false, false};
    uint64_t up_seq = 0;
    // collector only: chunks of the scatter path, event of the small copies
    void *pin_dn[N_RING] = {};
    hipEvent_t ev_pin_dn[N_RING] = {};
    hipEvent_t ev_small = nullptr, ev_tail = nullptr;
    bool ready = false;
};

namespace {

hipEvent_t new_event() {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return e;
}

int ctx_init(LrmHostCtx &c) {
    if (c.ready) return 0;
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc(&c.pin_up[b], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) { lrm_set_error("pinned staging allocation failed"); return -1; }
        if (!(c.ev_pin_up[b] = new_event())) { lrm_set_error("event creation failed"); return -1; }
        for (int s = 0; s < N_SLOTS; ++s) if (!(c.slots[s].ev_dense[b] = new_event())) { lrm_set_error("event creation failed"); return -1; }
    }
    for (int b = 0; b < N_RING; ++b) {
        if (hipHostMalloc(&c.pin_dn[b], RING_CHUNK, hipHostMallocDefault) != hipSuccess) { lrm_set_error("pinned staging allocation failed"); return -1; }
        if (!(c.ev_pin_dn[b] = new_event())) { lrm_set_error("event creation failed"); return -1; }
    }
    if (!(c.ev_small = new_event()) || !(c.ev_tail = new_event())) { lrm_set_error("event creation failed"); return -1; }
    // Priorities: the result path first (pack kernels + downloads), then the extension of a finished group, then
    // the seed kernels of later sub-batches -- otherwise every group's extension finishes at the very end, behind
    // all the seed work, and the downloads of all but the first group run after the compute instead of under it.
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // numerically lower = higher priority
    const int p_seed = prio_lo, p_ext = prio_hi < prio_lo ? prio_lo - 1 : prio_lo, p_down = prio_hi;
    if (hipStreamCreateWithPriority(&c.up, hipStreamNonBlocking, p_down) != hipSuccess ||
        hipStreamCreateWithPriority(&c.down, hipStreamNonBlocking, p_down) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    for (int s = 0; s < N_EXT_STREAMS; ++s)
        if (hipStreamCreateWithPriority(&c.ext[s], hipStreamNonBlocking, p_ext) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    for (int s = 0; s < N_SEED_STREAMS; ++s)
        if (hipStreamCreateWithPriority(&c.seed[s], hipStreamNonBlocking, p_seed) != hipSuccess) { lrm_set_error("stream creation failed"); return -1; }
    c.ready = true;
    return 0;
}

int ensure_events(std::vector<hipEvent_t> &v, size_t n) {
    while (v.size() < n) {
        hipEvent_t e = new_event();
        if (!e) { lrm_set_error("event creation failed"); return -1; }
        v.push_back(e);
    }
    return 0;
}

void par_memcpy(void *dst, const void *src, uint64_t bytes, int threads) {
    if (threads <= 1 || bytes < (4ull << 20)) { memcpy(dst, src, bytes); return; }
    const uint64_t piece = 1ull << 20, np = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(static) num_threads(threads)
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
int h2d(LrmHostCtx &c, void *d_dst, const void *h_src, uint64_t bytes, bool pinned, int threads) {
    if (bytes == 0) return 0;
    if (pinned) { HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c.up)); return 0; }
    for (uint64_t o = 0; o < bytes; o += STAGE_CHUNK, ++c.up_seq) {
        const int b = (int) (c.up_seq & 1);
        const uint64_t l = bytes - o < STAGE_CHUNK ? bytes - o : STAGE_CHUNK;
        if (c.pin_up_used[b] && wait_event(c.ev_pin_up[b])) return -1;         // the chunk's previous DMA has drained
        par_memcpy(c.pin_up[b], (const char *) h_src + o, l, threads);
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

// Run-length CIGAR text on the device (what parse_cigar prints, alnmain.c:497-498: '=' and 'X' columns as M): one
// workgroup per read walks the op bytes 4096 columns at a time; a run is printed where it ENDS, its start comes from a
// prefix maximum of the run starts, its place in the text from a prefix sum of the bytes the earlier runs print.
// WRITE = false: only the text length (tlen[row]); WRITE = true: the text at dense + off[row], NUL-terminated.
// Reads without an alignment (no ops, locus outside every sequence, score -1) print "*".
__device__ __forceinline__ uint32_t op_class(uint32_t b) { return (b == '=' || b == 'X') ? (uint32_t) 'M' : b; }
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    return v < 10 ? 1u : v < 100 ? 2u : v < 1000 ? 3u : v < 10000 ? 4u : v < 100000 ? 5u : v < 1000000 ? 6u : v < 10000000 ? 7u : 10u;
}
template <bool IS_MAX>
__device__ __forceinline__ int block_excl_scan(int v, int *s_w, int *total) {       // exclusive scan over 256 threads (max with -1 / sum with 0)
    const int lane = (int) (threadIdx.x & 63u), wave = (int) (threadIdx.x >> 6);
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x = IS_MAX ? (y > x ? y : x) : x + y;
    }
    __syncthreads();                                               // s_w of the previous scan has been read
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    int before = IS_MAX ? -1 : 0, all = IS_MAX ? -1 : 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = s_w[w];
        all = IS_MAX ? (t > all ? t : all) : all + t;
        if (w < wave) before = IS_MAX ? (t > before ? t : before) : before + t;
    }
    int excl = __shfl_up(x, 1, 64);
    if (lane == 0) excl = IS_MAX ? -1 : 0;
    *total = all;
    return IS_MAX ? (excl > before ? excl : before) : excl + before;
}
template <bool WRITE>
__global__ __launch_bounds__(256) void cigar_text_kernel(const uint8_t *__restrict__ store, uint64_t pitch, const int32_t *__restrict__ n_ops,
                                                         const int32_t *__restrict__ score, const int32_t *__restrict__ meta_r,
                                                         uint32_t *__restrict__ tlen, const uint64_t *__restrict__ off,
                                                         uint8_t *__restrict__ dense, uint64_t rows) {
    __shared__ int s_w[4];
    const uint64_t row = blockIdx.x;
    if (row >= rows) return;
    const int n = n_ops[row];
    const bool none = n <= 0 || meta_r[row] == 0 || score[row] == -1;
    uint8_t *out = WRITE ? dense + off[row] : nullptr;
    if (none) {
        if (threadIdx.x == 0) { if (WRITE) { out[0] = '*'; out[1] = 0; } else tlen[row] = 1; }
        return;
    }
    const uint8_t *ops = store + row * pitch;
    int carry_start = 0, carry_out = 0;
    for (int base = 0; base < n; base += 4096) {
        const int c0 = base + (int) threadIdx.x * 16;
        uint32_t cl[18];                                           // classes of columns c0 - 1 .. c0 + 16 (0 = outside the read)
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int col = c0 - 1 + k;
            cl[k] = col >= 0 && col < n ? op_class(ops[col]) : 0u;
        }
        int last_start = -1;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (c0 + k < n && cl[k + 1] != cl[k]) last_start = c0 + k;
        int any_start;
        const int before = block_excl_scan<true>(last_start, s_w, &any_start);
        const int open = before >= 0 ? before : carry_start;       // start of the run that is open at my first column
        int bytes = 0, cs = open;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int col = c0 + k;
            if (col < n) {
                if (cl[k + 1] != cl[k]) cs = col;
                if (cl[k + 2] != cl[k + 1]) bytes += (int) dec_digits((uint32_t) (col - cs + 1)) + 1;
            }
        }
        int chunk_bytes;
        int o = carry_out + block_excl_scan<false>(bytes, s_w, &chunk_bytes);
        if (WRITE) {
            cs = open;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int col = c0 + k;
                if (col < n) {
                    if (cl[k + 1] != cl[k]) cs = col;
                    if (cl[k + 2] != cl[k + 1]) {
                        uint32_t len = (uint32_t) (col - cs + 1);
                        const int nd = (int) dec_digits(len);
                        for (int d = nd - 1; d >= 0; --d) { out[o + d] = (uint8_t) ('0' + len % 10u); len /= 10u; }
                        out[o + nd] = (uint8_t) cl[k + 1];
                        o += nd + 1;
                    }
                }
            }
        }
        carry_out += chunk_bytes;
        if (any_start >= 0) carry_start = any_start;
    }
    if (threadIdx.x == 0) { if (WRITE) out[carry_out] = 0; else tlen[row] = (uint32_t) carry_out; }
}

// dense device buffer -> the caller's memory through the context's ring of pinned chunks: contiguous DMA pieces, every
// piece copied into place while the next ones fly.  Entries i with off[i] (16-byte aligned, ascending) / len[i] in the
// dense buffer go to dst[i]; dst == nullptr: the image is copied as it is to `flat`.
// `threads` = 1: the collector copies alone with plain memcpy -- no OpenMP team, whose idle threads spin between the
// pieces (8 threads spinning through every download was 0.35 CPU-s per Gbp); the row layout of the op bytes (1.1 GB per
// Gbp to scatter) needs the team.
// `after_issue` runs once, as soon as the last piece has been handed to the DMA engine (before the ring is drained):
// whatever it queues flies while this thread still copies.
template <typename F>
int d2h_ring(LrmHostCtx &c, const uint8_t *d_dense, uint64_t total, const uint64_t *off, const uint32_t *len,
             uint8_t *const *dst, uint64_t rows, uint8_t *flat, int threads, F after_issue) {
    if (total == 0) return after_issue();
    const uint64_t np = (total + RING_CHUNK - 1) / RING_CHUNK;
    uint64_t row_lo = 0;                                              // first entry that may still have bytes at or after the piece
    for (uint64_t k = 0; k < np + N_RING - 1; ++k) {
        if (k < np) {                                                 // issue piece k (its chunk was drained N_RING pieces ago)
            const int b = (int) (k % N_RING);
            const uint64_t o = k * RING_CHUNK, l = total - o < RING_CHUNK ? total - o : RING_CHUNK;
            HIPCHK(hipMemcpyAsync(c.pin_dn[b], d_dense + o, l, hipMemcpyDeviceToHost, c.down));
            HIPCHK(hipEventRecord(c.ev_pin_dn[b], c.down));
            if (k + 1 == np && after_issue()) return -1;
        }
        if (k + 1 < N_RING) continue;
        const uint64_t p = k + 1 - N_RING;                            // drain piece p while the later ones fly
        if (p >= np) break;
        const int pb = (int) (p % N_RING);
        if (wait_event(c.ev_pin_dn[pb])) return -1;
        const uint8_t *chunk = (const uint8_t *) c.pin_dn[pb];
        const uint64_t c0 = p * RING_CHUNK, c1 = c0 + (total - c0 < RING_CHUNK ? total - c0 : RING_CHUNK);
        if (!dst) { par_memcpy(flat + c0, chunk, c1 - c0, threads); continue; }
        while (row_lo < rows && off[row_lo] + len[row_lo] <= c0) ++row_lo;
        uint64_t row_hi = row_lo;
        while (row_hi < rows && off[row_hi] < c1) ++row_hi;
        if (threads <= 1) {
            for (uint64_t r = row_lo; r < row_hi; ++r) {
                const uint64_t a = off[r] > c0 ? off[r] : c0, e = off[r] + len[r] < c1 ? off[r] + len[r] : c1;
                if (e > a) memcpy(dst[r] + (a - off[r]), chunk + (a - c0), e - a);
            }
        } else {
#pragma omp parallel for schedule(static) num_threads(threads)
            for (uint64_t r = row_lo; r < row_hi; ++r) {
                const uint64_t a = off[r] > c0 ? off[r] : c0, e = off[r] + len[r] < c1 ? off[r] + len[r] : c1;
                if (e > a) memcpy(dst[r] + (a - off[r]), chunk + (a - c0), e - a);
            }
        }
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
// through the device in slices of ~32 GB of scratch, two of them in flight.  Results do not depend on the slicing:
// there is no cross-read state.
uint64_t host_slice_reads(uint32_t max_len, const LrmMapTune &mt) {
    if (mt.slice_reads >= 1) return mt.slice_reads;
    const uint64_t per_read = 13ull * (max_len ? max_len : 1) + 4096;
    uint64_t r = (32ull << 30) / per_read;
    return r < 16384 ? 16384 : r;
}

// Seed sub-batches of one device pass: small enough that the first kernels start a few milliseconds after the
// upload begins and the uploads hide behind them.
constexpr uint64_t PIPE_MIN_READS = 8192;
uint64_t pipe_subs(uint64_t n, const LrmMapTune &mt) {
    if (mt.sub_batches >= 1) return mt.sub_batches < n ? mt.sub_batches : n;
    const uint64_t k = n / PIPE_MIN_READS;
    return k < 2 ? 1 : (k > 12 ? 12 : k);
}
// Sub-batches per extension group: the bit-sliced kernel carries one read per LANE, so it wants >= 32 k reads
// per launch for decent SIMD coverage; two groups are in extension at once (two streams).  Measured per 100 k-read
// batch [r2, one batch at a time]: groups of 17 k reads 69 ms, 25 k 73 ms, 33 k 78 ms.
// With ANOTHER slice in flight on the device the chain inside one slice no longer matters, the fill of the chip does:
// groups of ~50 k reads (two per 100 k-read batch: 41.4 ms per batch with two in flight against 48.3 with groups of
// 17 k [r3]; a single call prefers the small groups: 58.0 against 62.0).
constexpr uint64_t EXT_GROUP_READS = 16384, EXT_GROUP_READS_BUSY = 49152;
uint64_t ext_group_subs(uint64_t sub, uint64_t nsub, const LrmMapTune &mt, bool busy) {
    if (mt.group_subs >= 1) return mt.group_subs < nsub ? mt.group_subs : nsub;
    const uint64_t want = busy ? EXT_GROUP_READS_BUSY : EXT_GROUP_READS;
    const uint64_t g = (want + sub - 1) / (sub ? sub : 1);
    return g < 1 ? 1 : (g > nsub ? nsub : g);
}

}  // namespace

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

struct Range { uint64_t off, m; };

// one slice of a submitted batch on one replica, from the issuer's queue to its collection
struct SliceJob {
    MapJob j;
    LrmMapTune mt;
    lrm_ticket *ticket = nullptr;
    uint32_t max_len = 0;
    Slot *slot = nullptr;
    bool busy = false;                // another slice was queued or in flight when this one was issued
    // plan (made by the issuer)
    std::vector<Range> subs, units;
    std::vector<size_t> ends, unit_of;
    uint64_t dstride = 0;
    bool seed_only = false;
    HostClock clk;
    // issuer -> collector hand-off
    std::mutex m;
    std::condition_variable cv;
    uint64_t issued = 0;              // units handed to the device
    bool issue_done = false;
    int rc = 0;
    std::string err;
    std::atomic<bool> failed{false};
    void fail(int code) {
        std::lock_guard<std::mutex> g(m);
        if (!rc) { rc = code; err = lrm_last_error(); }
        failed.store(true);
        cv.notify_all();
    }
};

namespace {

int get_ws(lrm_workspace *&ws, lrm_index *idx, uint64_t n, uint32_t max_len, uint32_t seed_len, uint32_t thres, int parts) {
    if (ws && n <= ws->n_max && max_len <= ws->max_len && (!(parts & LRM_WS_SEED) || (seed_len == ws->seed_len && thres <= ws->thres))) return 0;
    if (ws) { lrm_workspace_free(ws); ws = nullptr; }
    return lrm_workspace_create_parts(&ws, idx, n, max_len, seed_len, thres, parts);
}

int take_errors(Slot &s) {
    int rc = 0;
    for (int k = 0; k < N_SEED_STREAMS; ++k) if (lrm_ws_take_error(s.ws_seed[k])) rc = -2;
    for (int k = 0; k < N_EXT_STREAMS; ++k) if (lrm_ws_take_error(s.ws_ext[k])) rc = -2;
    return rc;
}

// ---- issuer: plan a slice and hand all of it to the device (no waits but for pageable staging chunks) -------------------------
int plan_and_issue(LrmHostCtx &c, SliceJob &sj) {
    lrm_index *idx = c.idx;
    const MapJob &j = sj.j;
    const LrmMapTune &mt = sj.mt;
    Slot &S = *sj.slot;
    const uint64_t n = j.n, nsub = pipe_subs(n, mt), sub = (n + nsub - 1) / nsub;
    sj.dstride = (j.store_stride + 3) & ~3ull;           // the bit-sliced kernel stores CIGAR bytes four at a time
    sj.seed_only = !(j.mode & DO_EXTEND);
    const uint64_t dstride = sj.dstride;
    const bool pin_reads = is_pinned(j.reads);
    std::vector<Range> &subs = sj.subs, &units = sj.units;
    std::vector<size_t> &ends = sj.ends, &unit_of = sj.unit_of;
    for (uint64_t off = 0; off < n; off += sub) subs.push_back({off, n - off < sub ? n - off : sub});
    const uint64_t gsub = (j.mode & DO_EXTEND) ? ext_group_subs(sub, subs.size(), mt, sj.busy) : 1;
    for (size_t k = gsub; k < subs.size(); k += gsub) ends.push_back(k);
    ends.push_back(subs.size());
    unit_of.resize(subs.size());
    for (size_t g = 0, k0 = 0; g < ends.size(); k0 = ends[g], ++g) {
        units.push_back({subs[k0].off, subs[ends[g] - 1].off + subs[ends[g] - 1].m - subs[k0].off});
        for (size_t k = k0; k < ends[g]; ++k) unit_of[k] = g;
    }
    const int n_ext_streams = mt.ext_streams >= 1 && mt.ext_streams <= N_EXT_STREAMS ? mt.ext_streams : 2;
    const int n_seed_streams = mt.seed_streams >= 1 && mt.seed_streams <= N_SEED_STREAMS ? mt.seed_streams : 2;
    uint64_t unit_max = 0;
    for (auto &u : units) unit_max = u.m > unit_max ? u.m : unit_max;
    if (n > 0x7fffffffull) { lrm_set_error("batch too large"); return -1; }
    if (j.mode & DO_SEED)
        for (int s = 0; s < n_seed_streams && (size_t) s < subs.size(); ++s)
            if (get_ws(S.ws_seed[s], idx, sub, sj.max_len, j.p.seed_len, j.p.thres, LRM_WS_SEED)) return -1;
    if (j.mode & DO_EXTEND)
        for (int s = 0; s < n_ext_streams && (size_t) s < units.size(); ++s)
            if (get_ws(S.ws_ext[s], idx, unit_max, sj.max_len, 20, 300, LRM_WS_EXTEND)) return -1;
    if (ensure_events(S.ev_up, subs.size()) || ensure_events(S.ev_seed, subs.size()) || ensure_events(S.ev_ext, units.size())) return -1;
    DevSet &d = S.dev;
    if (d.reads.ensure(n * j.stride) || d.lens.ensure(n * 4) || d.best.ensure(n * sizeof(lrm_entry))) { lrm_set_error("device allocation failed"); return -1; }
    if ((j.mode & DO_EXTEND) && (d.store.ensure(n * dstride) || d.nops.ensure(n * 4) || d.score.ensure(n * 4) ||
                                  d.meta.ensure(n * sizeof(lrm_seq_meta)) || d.mr.ensure(n * 4) ||
                                  (mt.cigar_text && d.tlen.ensure(n * 4)))) { lrm_set_error("device allocation failed"); return -1; }
    // (the dense result buffers and offset tables at their worst-case size for a unit, so that the collector never
    //  reallocates -- a hipFree would drain the whole device -- while other work is in flight)
    if (j.mode & DO_EXTEND)
        for (int b = 0; b < 2; ++b)
            if (S.dense[b].ensure(unit_max * (dstride + j.stride + 32)) || S.offs[b].ensure(unit_max * 2 * 12)) { lrm_set_error("device allocation failed"); return -1; }
    if (S.h_small.ensure(n * 96 + 4096)) { lrm_set_error("pinned staging allocation failed"); return -1; }
    S.dense_used[0] = S.dense_used[1] = false;

    for (uint64_t k = 0; k < subs.size(); ++k) {
        if (sj.failed.load()) return 0;                                            // the collector hit an error: stop feeding the device
        const int s = (int) (k % (uint64_t) n_seed_streams);
        const uint64_t m = subs[k].m, off = subs[k].off;
        const double t_i0 = sj.clk.ms();
        char *dr = (char *) d.reads.p + off * j.stride;
        if (h2d(c, dr, j.reads + off * j.stride, m * j.stride, pin_reads, mt.copy_threads ? (int) mt.copy_threads : c.copy_threads)) return -1;
        HIPCHK(hipMemcpyAsync((uint32_t *) d.lens.p + off, j.lens + off, m * 4, hipMemcpyHostToDevice, c.up));
        if (!(j.mode & DO_SEED)) HIPCHK(hipMemcpyAsync((lrm_entry *) d.best.p + off, j.best_in + off, m * sizeof(lrm_entry), hipMemcpyHostToDevice, c.up));
        HIPCHK(hipEventRecord(S.ev_up[k], c.up));
        if (j.mode & DO_SEED) {
            HIPCHK(hipStreamWaitEvent(c.seed[s], S.ev_up[k], 0));
            if (lrm_launch_seed(idx, S.ws_seed[s], dr, j.stride, (const uint32_t *) d.lens.p + off, m, sj.max_len, j.p.seed_len, j.p.thres,
                                (lrm_entry *) d.best.p + off, mt, c.seed[s])) return -1;
            HIPCHK(hipEventRecord(S.ev_seed[k], c.seed[s]));
        }
        const uint64_t g = unit_of[k];
        const bool closes = k + 1 == ends[g];
        if (closes && (j.mode & DO_EXTEND)) {                                  // the group's extension, behind its seeds / uploads
            const int xs = (int) (g % (uint64_t) n_ext_streams);
            for (uint64_t x = g ? ends[g - 1] : 0; x <= k; ++x) HIPCHK(hipStreamWaitEvent(c.ext[xs], (j.mode & DO_SEED) ? S.ev_seed[x] : S.ev_up[x], 0));
            const Range &u = units[g];
            if (lrm_launch_extend(idx, S.ws_ext[xs], (char *) d.reads.p + u.off * j.stride, j.stride, (const uint32_t *) d.lens.p + u.off, u.m,
                                  sj.max_len, (const lrm_entry *) d.best.p + u.off, j.gp, (uint8_t *) d.store.p + u.off * dstride, dstride,
                                  (int32_t *) d.nops.p + u.off, (int32_t *) d.score.p + u.off, (lrm_seq_meta *) d.meta.p + u.off,
                                  (int32_t *) d.mr.p + u.off, mt, c.ext[xs])) return -1;
            if (mt.cigar_text) {                                               // length of every read's run-length CIGAR text
                hipLaunchKernelGGL(cigar_text_kernel<false>, dim3((uint32_t) u.m), dim3(256), 0, c.ext[xs], (const uint8_t *) d.store.p + u.off * dstride, dstride,
                                   (const int32_t *) d.nops.p + u.off, (const int32_t *) d.score.p + u.off, (const int32_t *) d.mr.p + u.off,
                                   (uint32_t *) d.tlen.p + u.off, (const uint64_t *) nullptr, (uint8_t *) nullptr, u.m);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(S.ev_ext[g], c.ext[xs]));
        }
        if (closes) {
            { std::lock_guard<std::mutex> lk(sj.m); sj.issued = g + 1; }
            sj.cv.notify_all();
        }
        if (sj.clk.on) fprintf(stderr, "[lrm host] issue   off=%llu m=%llu: %.1f -> %.1f ms\n", (unsigned long long) off, (unsigned long long) m, t_i0, sj.clk.ms());
    }
    if (sj.clk.on) {
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        fprintf(stderr, "[lrm host] slice issued at %.1f ms (issuer thread CPU so far %.1f ms)\n", sj.clk.ms(), ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6);
    }
    return 0;
}

// ---- collector: one unit [off, off + m) of the slice, once `done` has fired ------------------------------------------------
int collect(LrmHostCtx &c, SliceJob &sj, size_t g) {
    const MapJob &j = sj.j;
    Slot &S = *sj.slot;
    DevSet &d = S.dev;
    const Range &u = sj.units[g];
    hipEvent_t done = sj.seed_only ? S.ev_seed[sj.ends[g] - 1] : S.ev_ext[g];
    const double t_in = sj.clk.ms();
    if (wait_event(done)) return -1;
    const double t_done = sj.clk.ms();
    if (take_errors(S)) return -2;                                   // raised by this or an earlier unit: never lost
    const uint64_t m = u.m, o = u.off, dstride = sj.dstride;
    // small arrays: device -> this unit's region of the pinned staging -> the caller's arrays
    uint8_t *hs = (uint8_t *) S.h_small.p + o * 96;
    lrm_entry *h_best = (lrm_entry *) hs;                            // 24 B per read
    lrm_seq_meta *h_meta = (lrm_seq_meta *) (hs + m * 24);           // 24
    int32_t *h_nops = (int32_t *) (hs + m * 48), *h_score = h_nops + m, *h_mr = h_score + m;   // 3 x 4
    uint32_t *h_len = (uint32_t *) (h_mr + m);                       // 2 x 4
    uint64_t *h_off = (uint64_t *) (hs + m * 72);                    // 2 x 8  (8-byte aligned: o*96 + m*72)
    uint32_t *h_tlen = (uint32_t *) (hs + m * 88);                   // 4  (cigar_text)
    const bool text = sj.mt.cigar_text != 0 && (j.mode & DO_EXTEND);
    if (j.mode & DO_SEED) HIPCHK(hipMemcpyAsync(h_best, (const lrm_entry *) d.best.p + o, m * sizeof(lrm_entry), hipMemcpyDeviceToHost, c.down));
    if (j.mode & DO_EXTEND) {
        HIPCHK(hipMemcpyAsync(h_nops, (const int32_t *) d.nops.p + o, m * 4, hipMemcpyDeviceToHost, c.down));
        HIPCHK(hipMemcpyAsync(h_score, (const int32_t *) d.score.p + o, m * 4, hipMemcpyDeviceToHost, c.down));
        HIPCHK(hipMemcpyAsync(h_meta, (const lrm_seq_meta *) d.meta.p + o, m * sizeof(lrm_seq_meta), hipMemcpyDeviceToHost, c.down));
        HIPCHK(hipMemcpyAsync(h_mr, (const int32_t *) d.mr.p + o, m * 4, hipMemcpyDeviceToHost, c.down));
        if (text) HIPCHK(hipMemcpyAsync(h_tlen, (const uint32_t *) d.tlen.p + o, m * 4, hipMemcpyDeviceToHost, c.down));
    }
    HIPCHK(hipEventRecord(c.ev_small, c.down));
    if (wait_event(c.ev_small)) return -1;
    if (j.mode & DO_SEED) memcpy(j.best_out + o, h_best, m * sizeof(lrm_entry));
    if (!(j.mode & DO_EXTEND)) return 0;
    memcpy(j.score + o, h_score, m * 4);
    memcpy(j.meta + o, h_meta, m * sizeof(lrm_seq_meta));
    memcpy(j.meta_r + o, h_mr, m * 4);

    // Dense image of the unit on the device: the used part of every CIGAR row, then the reads that were
    // reverse-complemented in place (alnmain.c:437; the other rows of reads_buf did not change).  Everything crosses
    // the link by DMA (hipMemcpyAsync): a hand-written kernel that writes the caller's pinned memory runs at the link
    // rate on an idle chip and at 2-9 GB/s once the compute kernels of the batches in flight own the wave slots, stream
    // priority or not, while the DMA keeps 50-57 GB/s (tools/d2h_under_load.hip, profiles/r3/probes).
    uint8_t *h_store = j.store_mem + o * j.store_stride;
    const bool pin_store = is_pinned(h_store);
    const bool dense = sj.mt.dense != 0;
    const int copy_threads = sj.mt.copy_threads ? (int) sj.mt.copy_threads : c.copy_threads;
    uint64_t total_ops = 0, total = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t cap = j.store_stride;
        if (text) h_len[i] = h_tlen[i] + 1u;                                  // the text and its NUL
        else h_len[i] = h_nops[i] > 0 ? (uint32_t) ((uint64_t) h_nops[i] < cap ? (uint64_t) h_nops[i] : cap) : 0u;
        h_off[i] = total_ops;
        total_ops += ((uint64_t) h_len[i] + 15) & ~15ull;
    }
    if (text && total_ops > m * j.store_stride) {
        lrm_set_error("run-length CIGAR text of a group (%llu bytes) does not fit the %llu bytes of its rows in store_mem",
                      (unsigned long long) total_ops, (unsigned long long) (m * j.store_stride));
        return -3;
    }
    total = total_ops;
    uint64_t n_rev = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const bool rev = !sj.mt.keep_reads && h_mr[i] != 0 && h_meta[i].strand == 1;   // (keep_reads: the caller's buffer stays as it is)
        h_len[m + i] = rev ? j.lens[o + i] : 0u;
        h_off[m + i] = total;
        total += ((uint64_t) h_len[m + i] + 15) & ~15ull;
        n_rev += rev;
    }
    const int b = (int) (g & 1);
    if (S.dense_used[b] && wait_event(S.ev_dense[b])) return -1;     // the transfer two units ago has left the buffer
    const uint8_t *d_store = (const uint8_t *) d.store.p + o * dstride, *d_reads = (const uint8_t *) d.reads.p + o * j.stride;
    uint64_t *d_off = (uint64_t *) S.offs[b].p;
    uint32_t *d_len = (uint32_t *) ((uint8_t *) S.offs[b].p + 2 * m * 8);
    uint8_t *dn = (uint8_t *) S.dense[b].p;
    if (total) {
        HIPCHK(hipMemcpyAsync(d_len, h_len, 2 * m * 4, hipMemcpyHostToDevice, c.down));
        HIPCHK(hipMemcpyAsync(d_off, h_off, 2 * m * 8, hipMemcpyHostToDevice, c.down));
        const uint32_t gy_ops = (uint32_t) ((j.store_stride + 4095) / 4096), gy_rd = (uint32_t) ((j.stride + 4095) / 4096);
        if (n_rev)
            hipLaunchKernelGGL(pack_rows_kernel, dim3((uint32_t) m, gy_rd ? gy_rd : 1), dim3(256), 0, c.down, d_reads, j.stride,
                               d_len + m, d_off + m, dn, m);
        if (total_ops && text)
            hipLaunchKernelGGL(cigar_text_kernel<true>, dim3((uint32_t) m), dim3(256), 0, c.down, d_store, dstride, (const int32_t *) d.nops.p + o,
                               (const int32_t *) d.score.p + o, (const int32_t *) d.mr.p + o, (uint32_t *) nullptr, (const uint64_t *) d_off, dn, m);
        else if (total_ops)
            hipLaunchKernelGGL(pack_rows_kernel, dim3((uint32_t) m, gy_ops ? gy_ops : 1), dim3(256), 0, c.down, d_store, dstride,
                               d_len, d_off, dn, m);
        HIPCHK(hipGetLastError());
        std::vector<uint8_t *> dst(2 * m);
        for (uint64_t i = 0; i < m; ++i) {
            dst[i] = j.store_mem + (o + i) * j.store_stride;
            dst[m + i] = (uint8_t *) j.reads + (o + i) * j.stride;
        }
        if (dense) {
            // The reverse-complemented reads are the only rows the host has to place: through the chunk ring, copied by
            // this thread alone.  Then the op bytes: ONE DMA straight into the region of the caller's pinned store_mem
            // the unit's rows would occupy (sum of the 16-aligned lengths <= m * store_stride because
            // store_stride % 16 == 0) -- it flies while this thread goes on to the next unit.
            // The ring is deep enough (N_RING chunks) for every piece of a unit's reads to be handed to the DMA engine before
            // the first is drained, so the op bytes follow right behind them on the same stream and fly while this thread
            // places the reads.  (With four chunks the op bytes waited until the collector had copied nearly all of
            // the reads through them: 48 ms per batch on a box with a slow host memcpy against 32 with keep_reads, whose op
            // bytes leave at once; the op bytes on a second stream next to the ring: 38-40 ms -- two blit copies at a time
            // share the link badly; helper threads for the placement: no difference.)
            auto ops_dma = [&]() -> int {
                if (total_ops && pin_store) HIPCHK(hipMemcpyAsync(h_store, dn, total_ops, hipMemcpyDeviceToHost, c.down));
                return 0;
            };
            std::vector<uint64_t> roff(m);
            for (uint64_t i = 0; i < m; ++i) roff[i] = h_off[m + i] - total_ops;
            if (d2h_ring(c, dn + total_ops, total - total_ops, roff.data(), h_len + m, dst.data() + m, m, nullptr, 1, ops_dma)) return -1;
            if (total_ops && !pin_store && d2h_ring(c, dn, total_ops, nullptr, nullptr, nullptr, 0, h_store, copy_threads, []() { return 0; })) return -1;
        } else {
            // row layout (alnmain.c:322-325): every used CIGAR row and every reverse-complemented read is placed by the
            // host's memcpy team
            if (d2h_ring(c, dn, total, h_off, h_len, dst.data(), 2 * m, nullptr, copy_threads, []() { return 0; })) return -1;
        }
        HIPCHK(hipEventRecord(S.ev_dense[b], c.down));
        S.dense_used[b] = true;
    }
    for (uint64_t i = 0; i < m; ++i) {                               // alnmain.c:322-325, mutils.c:99-104
        j.cig[o + i].cigar = dense ? h_store + h_off[i] : j.store_mem + (o + i) * j.store_stride;
        j.cig[o + i].n_cigar_op = h_nops[i];
        j.cig[o + i].score = h_score[i];
    }
    if (sj.clk.on) fprintf(stderr, "[lrm host] collect off=%llu m=%llu: wait-from %.1f kernels-done %.1f issued %.1f ms (%s, %.0f MB)\n",
                           (unsigned long long) o, (unsigned long long) m, t_in, t_done, sj.clk.ms(), dense ? (pin_store ? "dense, DMA into store_mem" : "dense, staged") : "rows", total / 1e6);
    return 0;
}

void issuer_main(LrmHostCtx *cp) {
    LrmHostCtx &c = *cp;
    if (hipSetDevice(c.idx->device) != hipSuccess) { (void) hipGetLastError(); }
    for (;;) {
        std::unique_ptr<SliceJob> job;
        Slot *slot = nullptr;
        {
            std::unique_lock<std::mutex> lk(c.mu);
            c.cv.wait(lk, [&] {
                if (c.stop) return true;
                if (c.q_issue.empty()) return false;
                for (int k = 0; k < c.n_slots; ++k) if (!c.slots[k].busy) return true;
                return false;
            });
            if (c.stop && c.q_issue.empty()) return;
            if (c.q_issue.empty()) continue;
            for (int k = 0; k < c.n_slots; ++k) if (!c.slots[k].busy) { slot = &c.slots[k]; break; }
            if (!slot) continue;
            slot->busy = true;
            job = std::move(c.q_issue.front());
            c.q_issue.pop_front();
            job->busy = c.n_active > 1;
        }
        SliceJob *sj = job.get();
        sj->slot = slot;
        sj->clk.on = sj->mt.verbose != 0;
        {   // the collector follows the slice from now on
            std::lock_guard<std::mutex> lk(c.mu);
            c.q_collect.push_back(std::move(job));
        }
        c.cv.notify_all();
        int rc;
        try { rc = plan_and_issue(c, *sj); }
        catch (const std::exception &e) { lrm_set_error("issuer thread: %s", e.what()); rc = -1; }
        if (rc) sj->fail(rc);
        {   // (notified under the lock: the collector deletes the slice once it has seen issue_done)
            std::lock_guard<std::mutex> lk(sj->m);
            sj->issue_done = true;
            sj->cv.notify_all();
        }
    }
}

void collector_main(LrmHostCtx *cp) {
    LrmHostCtx &c = *cp;
    if (hipSetDevice(c.idx->device) != hipSuccess) { (void) hipGetLastError(); }
    for (;;) {
        std::unique_ptr<SliceJob> job;
        {
            std::unique_lock<std::mutex> lk(c.mu);
            c.cv.wait(lk, [&] { return c.stop || !c.q_collect.empty(); });
            if (c.q_collect.empty()) { if (c.stop) return; continue; }
            job = std::move(c.q_collect.front());
            c.q_collect.pop_front();
        }
        SliceJob &sj = *job;
        Slot &S = *sj.slot;
        for (size_t g = 0;; ++g) {
            {
                std::unique_lock<std::mutex> lk(sj.m);
                sj.cv.wait(lk, [&] { return sj.issued > g || sj.issue_done || sj.rc; });
                if (sj.rc || sj.issued <= g) break;                       // failed, or every unit has been collected
            }
            int rc;
            try { rc = collect(c, sj, g); }
            catch (const std::exception &e) { lrm_set_error("collector thread: %s", e.what()); rc = -1; }
            if (rc) { sj.fail(rc); break; }
        }
        { std::unique_lock<std::mutex> lk(sj.m); sj.cv.wait(lk, [&] { return sj.issue_done; }); }
        int rc = sj.rc;
        std::string err = sj.err;
        if (!rc) {
            // the last transfers into the caller's memory (DMA and device row writes are ordered on the download stream)
            if (hipEventRecord(c.ev_tail, c.down) != hipSuccess || wait_event(c.ev_tail)) { rc = -1; err = lrm_last_error(); }
        }
        if (rc) {                                                        // error path: let everything issued for this slot drain
            for (int s = 0; s < N_SEED_STREAMS; ++s) (void) hipStreamSynchronize(c.seed[s]);
            for (int s = 0; s < N_EXT_STREAMS; ++s) (void) hipStreamSynchronize(c.ext[s]);
            (void) hipStreamSynchronize(c.up);
            (void) hipStreamSynchronize(c.down);
            (void) take_errors(S);                                        // reported now: do not fail the next batch
        }
        if (sj.clk.on) {
            timespec ts;
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
            fprintf(stderr, "[lrm host] slice of %llu reads, %zu seed sub-batches, %zu units: %.1f ms (collector thread CPU so far %.1f ms)\n", (unsigned long long) sj.j.n,
                    sj.subs.size(), sj.units.size(), sj.clk.ms(), ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6);
        }
        lrm_ticket *t = sj.ticket;
        job.reset();
        {
            std::lock_guard<std::mutex> lk(c.mu);
            S.busy = false;
            --c.n_active;
        }
        c.cv.notify_all();
        t->part_done(rc, err);
    }
}

std::mutex g_host_ctx_init;             // creation of a handle's host context (the context's own mutex lives inside it)

int ensure_ctx(lrm_index *idx, int group_size) {
    if (lrm_require_device(idx->device)) return -1;
    std::lock_guard<std::mutex> g(g_host_ctx_init);
    if (!idx->host) {
        idx->host = new (std::nothrow) LrmHostCtx;
        if (!idx->host) { lrm_set_error("out of memory"); return -1; }
        idx->host->idx = idx;
    }
    LrmHostCtx &c = *idx->host;
    if (ctx_init(c)) return -1;
    // the memcpy team of the pageable paths: a library must not fan out over every core of its host, and the
    // replicas of a group share the host's CPU share
    int ct = lrm_host_threads() / (group_size > 0 ? group_size : 1);
    c.copy_threads = ct < 1 ? 1 : (ct > 8 ? 8 : ct);
    { long long v; if (idx->env.get("LRM_HOST_SLOTS", &v) && v >= 1 && v <= N_SLOTS) c.n_slots = (int) v; }
    if (!c.threads_up) {
        try {
            c.issuer = std::thread(issuer_main, &c);
            try { c.collector = std::thread(collector_main, &c); }
            catch (...) {
                { std::lock_guard<std::mutex> lk(c.mu); c.stop = true; }
                c.cv.notify_all();
                c.issuer.join();
                c.stop = false;
                throw;
            }
        } catch (const std::exception &e) {
            lrm_set_error("cannot start the host pipeline threads: %s", e.what());
            return -1;
        }
        c.threads_up = true;
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

// queue the slices of one replica's share of the batch
int submit_replica(lrm_index *ix, const MapJob &j, const LrmMapTune &mt, lrm_ticket *t, int group_size) {
    if (j.n == 0) return 0;
    if (ensure_ctx(ix, group_size)) return -1;
    LrmHostCtx &c = *ix->host;
    const uint32_t max_len = max_of(j.lens, j.n);
    const uint64_t slice = host_slice_reads(max_len, mt);
    std::vector<std::unique_ptr<SliceJob>> jobs;
    for (uint64_t o = 0; o < j.n; o += slice) {
        std::unique_ptr<SliceJob> sj(new SliceJob);
        sj->j = j.slice(o, j.n - o < slice ? j.n - o : slice);
        sj->mt = mt;
        sj->ticket = t;
        sj->max_len = max_len;
        jobs.push_back(std::move(sj));
    }
    { std::lock_guard<std::mutex> g(t->m); t->pending += (int) jobs.size(); }
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (auto &sj : jobs) { c.q_issue.push_back(std::move(sj)); ++c.n_active; }
    }
    c.cv.notify_all();
    return 0;
}

int submit_impl(lrm_index *idx, const MapJob &j, const lrm_map_options *opt, lrm_ticket **out) {
    LrmMapTune mt = idx->mtune;
    if (opt) {
        lrm_resolve_map_tune(opt, idx->env, &mt);
        mt.t3_limit = idx->dbg_t3_limit; mt.t3_slots = idx->dbg_t3_slots;
    }
    const uint32_t max_len = max_of(j.lens, j.n);
    if (j.stride < max_len) { lrm_set_error("stride < longest read"); return -1; }
    if ((j.mode & DO_EXTEND) && j.store_stride < 2ull * max_len) { lrm_set_error("store_stride < 2 * longest read (alnmain.c:316-320)"); return -1; }
    if ((j.mode & DO_EXTEND) && mt.dense && (j.store_stride & 15u)) { lrm_set_error("dense results need store_stride to be a multiple of 16"); return -1; }
    std::unique_ptr<lrm_ticket> t(new lrm_ticket);
    if (j.n) {
        if (idx->n_peers <= 1 || !idx->peers) {
            if (submit_replica(idx, j, mt, t.get(), 1)) return -1;
        } else {
            const int np = idx->n_peers;
            std::vector<uint64_t> cuts;
            partition_by_bases(j.lens, j.n, np, cuts);
            for (int r = 0; r < np; ++r) {
                const uint64_t lo = cuts[r], hi = cuts[r + 1];
                if (hi <= lo) continue;
                if (submit_replica(idx->peers[r], j.slice(lo, hi - lo), mt, t.get(), np)) {
                    // the replicas queued so far run to completion before the caller gets its buffers back
                    const std::string msg = lrm_last_error();
                    { std::unique_lock<std::mutex> lk(t->m); t->cv.wait(lk, [&] { return t->pending == 0; }); }
                    lrm_set_error("replica %d (device %d): %s", r, idx->peers[r]->device, msg.c_str());
                    return -1;
                }
            }
        }
    }
    *out = t.release();
    return 0;
}

int wait_impl(lrm_ticket *t) {
    int rc;
    {
        std::unique_lock<std::mutex> lk(t->m);
        t->cv.wait(lk, [&] { return t->pending == 0; });
        rc = t->rc;
        if (rc) lrm_set_error("%s", t->err.c_str());
    }
    delete t;
    return rc;
}

// C ABI: no C++ exception may leave the library (allocation failures of the host-side bookkeeping, thread creation)
int run_job(lrm_index *idx, const MapJob &j, const lrm_map_options *opt, lrm_ticket **ticket_out) {
    try {
        lrm_ticket *t = nullptr;
        if (submit_impl(idx, j, opt, &t)) return -1;
        if (ticket_out) { *ticket_out = t; return 0; }
        return wait_impl(t);
    } catch (const std::exception &e) {
        lrm_set_error("host-side failure: %s", e.what());
        return -1;
    } catch (...) {
        lrm_set_error("host-side failure");
        return -1;
    }
}

}  // namespace

void lrm_host_ctx_free(lrm_index *idx) {
    LrmHostCtx *c = idx->host;
    if (!c) return;
    idx->host = nullptr;
    if (c->threads_up) {
        {   // batches still queued or in flight run to completion first (their tickets stay valid)
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->n_active == 0; });
            c->stop = true;
        }
        c->cv.notify_all();
        c->issuer.join();
        c->collector.join();
    }
    if (c->ready) {
        for (int s = 0; s < N_SEED_STREAMS; ++s) (void) hipStreamSynchronize(c->seed[s]);
        for (int s = 0; s < N_EXT_STREAMS; ++s) (void) hipStreamSynchronize(c->ext[s]);
        (void) hipStreamSynchronize(c->up);
        (void) hipStreamSynchronize(c->down);
    }
    for (int b = 0; b < 2; ++b) {
        if (c->pin_up[b]) (void) hipHostFree(c->pin_up[b]);
        if (c->ev_pin_up[b]) (void) hipEventDestroy(c->ev_pin_up[b]);
    }
    for (int b = 0; b < N_RING; ++b) {
        if (c->pin_dn[b]) (void) hipHostFree(c->pin_dn[b]);
        if (c->ev_pin_dn[b]) (void) hipEventDestroy(c->ev_pin_dn[b]);
    }
    if (c->ev_small) (void) hipEventDestroy(c->ev_small);
    if (c->ev_tail) (void) hipEventDestroy(c->ev_tail);
    if (c->up) (void) hipStreamDestroy(c->up);
    if (c->down) (void) hipStreamDestroy(c->down);
    for (int s = 0; s < N_EXT_STREAMS; ++s) if (c->ext[s]) (void) hipStreamDestroy(c->ext[s]);
    for (int s = 0; s < N_SEED_STREAMS; ++s) if (c->seed[s]) (void) hipStreamDestroy(c->seed[s]);
    for (Slot &S : c->slots) {
        for (auto *v : {&S.ev_up, &S.ev_seed, &S.ev_ext}) for (hipEvent_t e : *v) (void) hipEventDestroy(e);
        for (int s = 0; s < N_SEED_STREAMS; ++s) if (S.ws_seed[s]) lrm_workspace_free(S.ws_seed[s]);
        for (int s = 0; s < N_EXT_STREAMS; ++s) if (S.ws_ext[s]) lrm_workspace_free(S.ws_ext[s]);
        DevSet &d = S.dev;
        d.reads.release(); d.lens.release(); d.best.release(); d.store.release();
        d.nops.release(); d.score.release(); d.meta.release(); d.mr.release(); d.tlen.release();
        for (int b = 0; b < 2; ++b) { S.dense[b].release(); S.offs[b].release(); if (S.ev_dense[b]) (void) hipEventDestroy(S.ev_dense[b]); }
        S.h_small.release();
    }
    delete c;
}

extern "C" int lrm_seed_batch(lrm_index *idx, const char *reads_buf, uint64_t stride, const uint32_t *lens,
                              uint64_t n, lrm_params p, lrm_entry *best_out) {
    if (!idx || !reads_buf || !lens || !best_out) { lrm_set_error("null argument"); return -1; }
    MapJob j = {};
    j.mode = DO_SEED; j.reads = const_cast<char *>(reads_buf); j.stride = stride; j.lens = lens; j.n = n; j.p = p; j.best_out = best_out;
    return run_job(idx, j, nullptr, nullptr);
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
    return run_job(idx, j, nullptr, nullptr);
}

static int map_job_of(MapJob &j, lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                      lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out, uint8_t *store_mem,
                      uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out, int *meta_r_out) {
    if (!idx || !reads_buf || !lens || !best_out || !cig_out || !store_mem || !score_out || !meta_out || !meta_r_out) {
        lrm_set_error("null argument");
        return -1;
    }
    j = MapJob{};
    j.mode = DO_SEED | DO_EXTEND; j.reads = reads_buf; j.stride = stride; j.lens = lens; j.n = n; j.p = p; j.gp = gp;
    j.best_out = best_out;
    j.cig = cig_out; j.store_mem = store_mem; j.store_stride = store_stride; j.score = score_out; j.meta = meta_out; j.meta_r = meta_r_out;
    return 0;
}

extern "C" int lrm_map_batch(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                             lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out, uint8_t *store_mem,
                             uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out, int *meta_r_out) {
    MapJob j;
    if (map_job_of(j, idx, reads_buf, stride, lens, n, p, gp, best_out, cig_out, store_mem, store_stride, score_out, meta_out, meta_r_out)) return -1;
    return run_job(idx, j, nullptr, nullptr);
}

extern "C" int lrm_map_batch_submit(lrm_index *idx, char *reads_buf, uint64_t stride, const uint32_t *lens, uint64_t n,
                                    lrm_params p, lrm_gact_params gp, lrm_entry *best_out, lrm_cigar *cig_out,
                                    uint8_t *store_mem, uint64_t store_stride, int *score_out, lrm_seq_meta *meta_out,
                                    int *meta_r_out, const lrm_map_options *opt, lrm_ticket **ticket_out) {
    if (!ticket_out) { lrm_set_error("null argument"); return -1; }
    MapJob j;
    if (map_job_of(j, idx, reads_buf, stride, lens, n, p, gp, best_out, cig_out, store_mem, store_stride, score_out, meta_out, meta_r_out)) return -1;
    return run_job(idx, j, opt, ticket_out);
}

extern "C" int lrm_map_batch_wait(lrm_ticket *ticket) {
    if (!ticket) { lrm_set_error("null ticket"); return -1; }
    try { return wait_impl(ticket); }
    catch (...) { lrm_set_error("host-side failure"); return -1; }
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
