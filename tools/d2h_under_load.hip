// d2h_under_load.hip -- device-to-pinned-host transfers while the chip is busy: the runtime's hipMemcpyAsync (a
// 256-workgroup blit kernel on this platform) against hand-written writers, with nothing / an integer-VALU kernel /
// a random-gather kernel running at lower stream priority.  Question behind it: the host pipeline's own row writer
// ran at a third of the rate of the runtime's copy once two batches were in flight -- is that the kernel's shape?
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/d2h_under_load tools/d2h_under_load.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the library's rows_out_kernel: persistent grid, one row at a time per workgroup
__global__ __launch_bounds__(256) void rows_out(const uint8_t *__restrict__ src, uint64_t spitch, uint8_t *__restrict__ dst, uint64_t dpitch,
                                                const uint32_t *__restrict__ len, uint64_t rows) {
    for (uint64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint32_t l = len[row];
        if (l == 0) continue;
        const uint8_t *s = src + row * spitch;
        uint8_t *d = dst + row * dpitch;
        const uint32_t head = (uint32_t) ((16u - ((uintptr_t) d & 15u)) & 15u);
        for (uint32_t o = threadIdx.x; o < head && o < l; o += 256) d[o] = s[o];
        for (uint32_t o = head + threadIdx.x * 16; o < l; o += 256 * 16) {
            if (o + 16 <= l) { uint32_t w[4]; __builtin_memcpy(w, s + o, 16); *reinterpret_cast<uint4 *>(d + o) = make_uint4(w[0], w[1], w[2], w[3]); }
            else for (uint32_t e = 0; o + e < l; ++e) d[o + e] = s[o + e];
        }
    }
}
// flat copy, U 16-byte pieces per lane in flight
template <int U>
__global__ __launch_bounds__(512) void flat_out(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16) {
    const uint64_t gs = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gs * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * gs < n16) v[u] = src[i + u * gs];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * gs < n16) dst[i + u * gs] = v[u];
    }
}
// rows, but every lane keeps U rows' pieces in flight: workgroup takes U consecutive rows at a time
template <int U>
__global__ __launch_bounds__(256) void rows_out_u(const uint8_t *__restrict__ src, uint64_t spitch, uint8_t *__restrict__ dst, uint64_t dpitch,
                                                  const uint32_t *__restrict__ len, uint64_t rows) {
    for (uint64_t r0 = (uint64_t) blockIdx.x * U; r0 < rows; r0 += (uint64_t) gridDim.x * U) {
        uint32_t l[U];
#pragma unroll
        for (int u = 0; u < U; ++u) l[u] = r0 + u < rows ? len[r0 + u] : 0u;
        uint32_t lmax = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) lmax = l[u] > lmax ? l[u] : lmax;
        for (uint32_t o = threadIdx.x * 16; o < lmax; o += 256 * 16) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) if (o + 16 <= l[u]) __builtin_memcpy(&v[u], src + (r0 + u) * spitch + o, 16);
#pragma unroll
            for (int u = 0; u < U; ++u) if (o + 16 <= l[u]) __builtin_memcpy(dst + (r0 + u) * dpitch + o, &v[u], 16);
        }
    }
}

__global__ __launch_bounds__(256) void valu_load(uint64_t iters, uint32_t *out) {
    uint32_t a = threadIdx.x, b = blockIdx.x, c = 7;
    for (uint64_t i = 0; i < iters; ++i) { a = a * 1664525u + b; b = (b ^ a) + c; c = c * 22695477u + a; }
    if (a == 0x12345u) out[0] = a + b + c;
}
__global__ __launch_bounds__(256) void gather_load(const uint64_t *buf, uint64_t mask, uint64_t iters, uint64_t *out) {
    uint64_t s = blockIdx.x * 256ull + threadIdx.x, acc = 0;
    for (uint64_t i = 0; i < iters; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; acc ^= buf[((s >> 20) & mask) * 8]; s ^= acc; }
    if (acc == 0x12345ull) out[0] = acc;
}

int main() {
    const uint64_t pitch = 20000, len = 11000, rows = 50000, bytes = rows * pitch;        // op-byte rows of a 10 kbp batch
    uint8_t *d_src, *h_dst; uint32_t *d_len; uint64_t *d_g; uint32_t *d_o;
    CHECK(hipMalloc((void **) &d_src, bytes)); CHECK(hipMemset(d_src, 3, bytes));
    CHECK(hipHostMalloc((void **) &h_dst, bytes, hipHostMallocPortable | hipHostMallocMapped));
    CHECK(hipMalloc((void **) &d_len, rows * 4));
    { uint32_t *l = (uint32_t *) malloc(rows * 4); for (uint64_t i = 0; i < rows; ++i) l[i] = (uint32_t) len; CHECK(hipMemcpy(d_len, l, rows * 4, hipMemcpyHostToDevice)); free(l); }
    const uint64_t glines = 1ull << 27;                           // 8 GiB gather buffer
    CHECK(hipMalloc((void **) &d_g, glines * 64)); CHECK(hipMemset(d_g, 1, glines * 64));
    CHECK(hipMalloc((void **) &d_o, 64));
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sd, sb1, sb2;
    CHECK(hipStreamCreateWithPriority(&sd, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&sb1, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&sb2, hipStreamNonBlocking, lo));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint64_t payload = rows * len;
    for (int load = 0; load < 4; ++load) {
        const char *lname = load == 0 ? "idle chip" : load == 1 ? "VALU kernel" : load == 2 ? "random-gather kernel" : "VALU + gather kernels";
        for (int v = 0; v < 9; ++v) {
            // background load long enough to cover the transfer (~200 ms)
            if (load & 1) hipLaunchKernelGGL(valu_load, dim3(256 * 16), dim3(256), 0, sb1, 3000000ull, d_o);
            if (load & 2) hipLaunchKernelGGL(gather_load, dim3(256 * 16), dim3(256), 0, sb2, d_g, glines - 1, 6000ull, (uint64_t *) d_o);
            usleep(3000);
            CHECK(hipEventRecord(e0, sd));
            const char *name = "";
            switch (v) {
            case 0: name = "hipMemcpyAsync, flat (dense image)"; CHECK(hipMemcpyAsync(h_dst, d_src, payload, hipMemcpyDeviceToHost, sd)); break;
            case 1: name = "flat_out<1>, 256 x 512"; hipLaunchKernelGGL(flat_out<1>, dim3(256), dim3(512), 0, sd, (const uint4 *) d_src, (uint4 *) h_dst, payload / 16); break;
            case 2: name = "flat_out<4>, 256 x 512"; hipLaunchKernelGGL(flat_out<4>, dim3(256), dim3(512), 0, sd, (const uint4 *) d_src, (uint4 *) h_dst, payload / 16); break;
            case 3: name = "flat_out<8>, 1024 x 512"; hipLaunchKernelGGL(flat_out<8>, dim3(1024), dim3(512), 0, sd, (const uint4 *) d_src, (uint4 *) h_dst, payload / 16); break;
            case 4: name = "rows_out (library), 256 wg"; hipLaunchKernelGGL(rows_out, dim3(256), dim3(256), 0, sd, d_src, pitch, h_dst, pitch, d_len, rows); break;
            case 5: name = "rows_out (library), 2048 wg"; hipLaunchKernelGGL(rows_out, dim3(2048), dim3(256), 0, sd, d_src, pitch, h_dst, pitch, d_len, rows); break;
            case 6: name = "rows_out_u<4>, 256 wg"; hipLaunchKernelGGL(rows_out_u<4>, dim3(256), dim3(256), 0, sd, d_src, pitch, h_dst, pitch, d_len, rows); break;
            case 7: name = "rows_out_u<8>, 512 wg"; hipLaunchKernelGGL(rows_out_u<8>, dim3(512), dim3(256), 0, sd, d_src, pitch, h_dst, pitch, d_len, rows); break;
            case 8: name = "rows_out_u<4>, 1024 wg"; hipLaunchKernelGGL(rows_out_u<4>, dim3(1024), dim3(256), 0, sd, d_src, pitch, h_dst, pitch, d_len, rows); break;
            }
            CHECK(hipEventRecord(e1, sd));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipDeviceSynchronize());
            printf("{\"background\": \"%s\", \"transfer\": \"%s\", \"ms\": %.2f, \"GBps\": %.1f}\n", lname, name, ms, payload / (ms * 1e-3) / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
