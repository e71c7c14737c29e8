// hostlink_bench.hip -- what the host-buffer boundary (lrm_host.hip) can expect from the link and the host:
//   A. multi-threaded memcpy into pinned (hipHostMalloc) vs pageable (malloc) destinations and from either source
//      (the result scatter / upload staging of the round-2 pipeline; "pinned slower than pageable" in BENCH_r02)
//   B. flat DMA: D2H, H2D, and both at once (the link is full duplex)
//   C. a kernel that writes rows straight into pinned host memory (posted writes), alone and next to a D2H DMA
//      (the reverse-complemented reads are the one result that has to land in strided caller rows)
//   D. process CPU time spent WAITING for the device: hipEventSynchronize on a default event (spins) vs an event
//      created with hipEventBlockingSync (sleeps)
//   hipcc --offload-arch=gfx950 -O3 -fopenmp -o tools/_bin/hostlink_bench tools/hostlink_bench.hip
#include <hip/hip_runtime.h>
#include <omp.h>
#include <sys/resource.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_s() {
    rusage u;
    getrusage(RUSAGE_SELF, &u);
    return u.ru_utime.tv_sec + u.ru_utime.tv_usec * 1e-6 + u.ru_stime.tv_sec + u.ru_stime.tv_usec * 1e-6;
}

static void par_memcpy(void *dst, const void *src, uint64_t bytes, int threads) {
    const uint64_t piece = 1ull << 20, np = (bytes + piece - 1) / piece;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (uint64_t i = 0; i < np; ++i) {
        const uint64_t o = i * piece, l = bytes - o < piece ? bytes - o : piece;
        memcpy((char *) dst + o, (const char *) src + o, l);
    }
}

// rows of a pitched device array -> rows of a pitched (host-pinned) array, 16 bytes per lane
__global__ __launch_bounds__(256) void copy_rows(const uint8_t *__restrict__ src, uint64_t spitch, uint8_t *__restrict__ dst,
                                                 uint64_t dpitch, uint32_t len, uint64_t rows, uint32_t row_step) {
    const uint64_t row = (uint64_t) blockIdx.x * row_step;
    if (row >= rows) return;
    const uint8_t *s = src + row * spitch;
    uint8_t *d = dst + row * dpitch;
    for (uint32_t o = (blockIdx.y * 256 + threadIdx.x) * 16; o + 16 <= len; o += gridDim.y * 256 * 16)
        *reinterpret_cast<uint4 *>(d + o) = *reinterpret_cast<const uint4 *>(s + o);
}

__global__ void spin_kernel(uint64_t cycles, uint64_t *out) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = t0;
}

int main(int argc, char **argv) {
    const uint64_t GB = 1ull << 30;
    const uint64_t bytes = (argc > 1 ? (uint64_t) atoll(argv[1]) : 1ull) * GB;
    const int maxthr = argc > 2 ? atoi(argv[2]) : 8;
    uint8_t *pin_a = nullptr, *pin_b = nullptr, *dev_a = nullptr, *dev_b = nullptr;
    CHECK(hipHostMalloc((void **) &pin_a, bytes, hipHostMallocPortable | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **) &pin_b, bytes, hipHostMallocDefault));
    uint8_t *pg_a = (uint8_t *) malloc(bytes), *pg_b = (uint8_t *) malloc(bytes);
    CHECK(hipMalloc((void **) &dev_a, bytes));
    CHECK(hipMalloc((void **) &dev_b, bytes));
    par_memcpy(pg_a, pin_a, bytes, maxthr); memset(pg_b, 1, bytes); memset(pin_a, 2, bytes); memset(pin_b, 3, bytes);
    CHECK(hipMemset(dev_a, 4, bytes)); CHECK(hipMemset(dev_b, 5, bytes));
    CHECK(hipDeviceSynchronize());

    // ---- A ----
    struct { const char *name; uint8_t *dst, *src; } cps[] = {
        {"pageable <- pageable", pg_a, pg_b}, {"pageable <- pinned", pg_a, pin_b},
        {"pinned(mapped) <- pageable", pin_a, pg_b}, {"pinned(mapped) <- pinned", pin_a, pin_b},
        {"pinned(default) <- pageable", pin_b, pg_a}};
    for (auto &c : cps)
        for (int thr : {1, maxthr}) {
            double best = 1e9, cpu = 0;
            for (int rep = 0; rep < 3; ++rep) {
                const double c0 = cpu_s(), t0 = now();
                par_memcpy(c.dst, c.src, bytes, thr);
                const double t = now() - t0;
                if (t < best) { best = t; cpu = cpu_s() - c0; }
            }
            printf("{\"test\": \"memcpy\", \"case\": \"%s\", \"threads\": %d, \"GBps\": %.1f, \"cpu_s_per_GB\": %.3f}\n", c.name, thr,
                   bytes / best / 1e9, cpu / (bytes / 1e9));
            fflush(stdout);
        }

    // ---- B ----
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t eb1, eb2;
    CHECK(hipEventCreateWithFlags(&eb1, hipEventBlockingSync | hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&eb2, hipEventBlockingSync | hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            if (mode != 1) CHECK(hipMemcpyAsync(pin_a, dev_a, bytes, hipMemcpyDeviceToHost, s1));
            if (mode != 0) CHECK(hipMemcpyAsync(dev_b, pin_b, bytes, hipMemcpyHostToDevice, s2));
            CHECK(hipEventRecord(eb1, s1)); CHECK(hipEventRecord(eb2, s2));
            CHECK(hipEventSynchronize(eb1)); CHECK(hipEventSynchronize(eb2));
            const double t = now() - t0;
            if (t < best) best = t;
        }
        printf("{\"test\": \"dma\", \"case\": \"%s\", \"GBps_each_direction\": %.1f}\n", mode == 0 ? "D2H" : mode == 1 ? "H2D" : "D2H + H2D at once",
               bytes / best / 1e9);
        fflush(stdout);
    }

    // ---- C: rows of 10 kB (pitch 10001 like reads_buf: rows start at any byte), half of the rows written ----
    {
        const uint64_t pitch = 10001, len = 10000, rows = bytes / pitch;
        for (int gy : {1, 3}) for (int with_dma = 0; with_dma < 2; ++with_dma) {
            double best = 1e9, best_dma = 0;
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now();
                double t_dma = 0;
                hipLaunchKernelGGL(copy_rows, dim3((uint32_t) (rows / 2), gy), dim3(256), 0, s1, dev_a, pitch, pin_a, pitch, (uint32_t) len, rows, 2u);
                CHECK(hipEventRecord(eb1, s1));
                if (with_dma) { CHECK(hipMemcpyAsync(pin_b, dev_b, bytes, hipMemcpyDeviceToHost, s2)); CHECK(hipEventRecord(eb2, s2)); }
                if (with_dma) { CHECK(hipEventSynchronize(eb2)); t_dma = now() - t0; }
                CHECK(hipEventSynchronize(eb1));
                const double t = now() - t0;
                if (t < best) { best = t; best_dma = t_dma; }
            }
            printf("{\"test\": \"kernel rows -> pinned host\", \"grid_y\": %d, \"concurrent_D2H_dma\": %d, \"row_GBps\": %.1f, \"dma_GBps\": %.1f}\n", gy, with_dma,
                   (rows / 2) * len / best / 1e9, with_dma ? bytes / best_dma / 1e9 : 0.0);
            fflush(stdout);
        }
    }

    // ---- D ----
    {
        int clock_khz = 100000;                                  // wall_clock64 ticks at 100 MHz on gfx9
        (void) hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0);
        const uint64_t cyc = (uint64_t) clock_khz * 200;          // 200 ms
        hipEvent_t e_spin, e_block;
        CHECK(hipEventCreateWithFlags(&e_spin, hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&e_block, hipEventBlockingSync | hipEventDisableTiming));
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, cyc, (uint64_t *) nullptr);
            const double c0 = cpu_s(), t0 = now();
            if (mode == 0) { CHECK(hipEventRecord(e_spin, s1)); CHECK(hipEventSynchronize(e_spin)); }
            else if (mode == 1) { CHECK(hipEventRecord(e_block, s1)); CHECK(hipEventSynchronize(e_block)); }
            else if (mode == 2) CHECK(hipStreamSynchronize(s1));
            else {                                               // what lrm_host.hip does: query, sleep 20..200 us, query
                CHECK(hipEventRecord(e_spin, s1));
                useconds_t nap = 20;
                while (hipEventQuery(e_spin) == hipErrorNotReady) { (void) hipGetLastError(); usleep(nap); if (nap < 200) nap += 20; }
            }
            printf("{\"test\": \"wait for a 200 ms kernel\", \"how\": \"%s\", \"wall_ms\": %.1f, \"cpu_ms\": %.1f}\n",
                   mode == 0 ? "hipEventSynchronize, default event" : mode == 1 ? "hipEventSynchronize, hipEventBlockingSync event" :
                   mode == 2 ? "hipStreamSynchronize" : "hipEventQuery + usleep(20..200 us)",
                   (now() - t0) * 1e3, (cpu_s() - c0) * 1e3);
            fflush(stdout);
        }
    }
    return 0;
}
