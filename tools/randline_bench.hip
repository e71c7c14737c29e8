// randline_bench.hip -- how many independent random 64-byte lines per second does one MI355X deliver?
//
// seed_search's time is its L2 misses divided by ~50 G random lines per second on every text and table layout
// measured in round 2 -- but that ceiling was only ever seen through seed_search itself.  This microbenchmark pins
// it independently: every lane issues loads of 8 or 16 bytes at uniformly random 64-byte lines of a large buffer
// (no two lanes of a wavefront share a line, nothing is reused: every load is an L2 and Infinity-Cache miss once the
// buffer is much larger than 256 MiB), with 1 / 2 / 4 / 8 loads in flight per lane, at several occupancies, either
// with addresses that do not depend on loaded data (pure throughput) or as a dependent chain (the shape of an FM
// backward search: the next address comes out of the previous load).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/randline_bench tools/randline_bench.hip
//   tools/_bin/randline_bench [GiB=64] [iters=64]
// Prints one JSON object per configuration and a final summary line.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// ILP loads in flight per lane and iteration; BYTES 8 or 16 per load; DEP: the next round's addresses depend on the
// data of this round (chain), else on a counter only
template <int ILP, int BYTES, bool DEP>
__global__ __launch_bounds__(256) void rand_lines(const uint8_t *__restrict__ buf, uint64_t line_mask, int iters,
                                                  uint64_t seed, uint64_t *__restrict__ out) {
    // (dynamic LDS passed at launch only limits the resident workgroups)
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = mix(seed + tid * 0x9e3779b97f4a7c15ull);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            s = mix(s + 0x9e3779b97f4a7c15ull * (uint64_t) (u + 1));
            const uint64_t line = s & line_mask;
            const uint8_t *p = buf + line * 64 + ((s >> 58) & 3u) * 16;   // one of the four 16-byte quarters of the line
            if (BYTES == 16) {
                const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(p);
                v[u] = x.x ^ x.y;
            } else {
                v[u] = *reinterpret_cast<const uint64_t *>(p);
            }
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) acc ^= v[u];
        if (DEP) s ^= acc;                                       // the next addresses wait for these loads
    }
    out[tid] = acc;
}

struct Cfg { int ilp, bytes, dep, blocks_per_cu; };

template <int ILP, int BYTES, bool DEP>
static int run(const uint8_t *buf, uint64_t line_mask, int iters, uint64_t *out, int blocks, int lds, float *ms) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {                         // first launch warms the TLBs / clocks
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((rand_lines<ILP, BYTES, DEP>), dim3(blocks), dim3(256), lds, 0, buf, line_mask, iters, 1234ull + rep, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(ms, a, b));
    }
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return 0;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 64.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint64_t lines = 1;
    while ((lines * 2) * 64 <= (uint64_t) (gib * 1073741824.0)) lines *= 2;       // power of two: the mask is uniform
    const uint64_t bytes = lines * 64;
    uint8_t *buf = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 0x5a, bytes));
    const int max_blocks = cus * 8;
    uint64_t *out = nullptr;
    CHECK(hipMalloc(&out, (uint64_t) max_blocks * 256 * 8 * 4));
    CHECK(hipDeviceSynchronize());
    fprintf(stderr, "device %s, %d CUs, buffer %.1f GiB (%llu lines)\n", prop.name, cus, bytes / 1073741824.0, (unsigned long long) lines);
    // resident workgroups per CU are limited through dynamic LDS (160 KB per CU): 8 -> 20000 B each, 4 -> 40000, 2 -> 80000
    const Cfg cfgs[] = {
        {1, 16, 0, 8}, {2, 16, 0, 8}, {4, 16, 0, 8}, {8, 16, 0, 8},
        {1, 8, 0, 8}, {4, 8, 0, 8},
        {1, 16, 0, 4}, {4, 16, 0, 4}, {1, 16, 0, 2}, {4, 16, 0, 2}, {8, 16, 0, 2},
        {1, 16, 1, 8}, {2, 16, 1, 8}, {4, 16, 1, 8}, {1, 16, 1, 6}, {1, 16, 1, 4},
    };
    double best = 0;
    for (const Cfg &c : cfgs) {
        const int lds = c.blocks_per_cu >= 8 ? 0 : (160 * 1024 / c.blocks_per_cu) - 2048;
        const int blocks = cus * c.blocks_per_cu * 4;             // four rounds of resident workgroups
        float ms = 0;
        int rc = 1;
#define RUN(I, B, D) if (c.ilp == I && c.bytes == B && c.dep == D) rc = run<I, B, D != 0>(buf, lines - 1, iters, out, blocks, lds, &ms)
        RUN(1, 16, 0); RUN(2, 16, 0); RUN(4, 16, 0); RUN(8, 16, 0); RUN(1, 8, 0); RUN(4, 8, 0);
        RUN(1, 16, 1); RUN(2, 16, 1); RUN(4, 16, 1);
#undef RUN
        if (rc) return 1;
        const double n = (double) blocks * 256.0 * iters * c.ilp;
        const double glines = n / (ms * 1e-3) / 1e9;
        if (glines > best) best = glines;
        printf("{\"buffer_gib\": %.2f, \"loads_in_flight_per_lane\": %d, \"bytes_per_load\": %d, \"dependent_chain\": %d, "
               "\"workgroups_per_cu\": %d, \"waves_per_cu\": %d, \"ms\": %.3f, \"G_lines_per_s\": %.2f, \"TB_per_s_of_64B_lines\": %.3f}\n",
               bytes / 1073741824.0, c.ilp, c.bytes, c.dep, c.blocks_per_cu, c.blocks_per_cu * 4, ms, glines, glines * 64e-3);
        fflush(stdout);
    }
    printf("{\"summary\": \"best\", \"buffer_gib\": %.2f, \"G_lines_per_s\": %.2f}\n", bytes / 1073741824.0, best);
    (void) hipFree(buf); (void) hipFree(out);
    return 0;
}
