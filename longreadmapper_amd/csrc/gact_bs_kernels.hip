// SPDX-License-Identifier: MIT
// Bit-sliced GACT for the default band (W = 128): ONE LANE PER READ, 64 lattice points per 64-bit word.
// Replaces the reference's per-read simple_gact call (mutils.c:97-103) for large batches; bit-exact
// against oracle/lrm_oracle.c:orc_gact (docs/GACT_SPEC.md).  tests/models/gact_bitslice_model.c is the
// CPU model of exactly this sequence of operations.
//
// Why: the score kernels (gact3_kernel) are bound by VALU issue at ~16 instructions per anti-diagonal
// for 128 lattice points.  The recurrence only needs DIFFERENCES of neighbouring scores, and with the
// +1/-1/-1 scheme those lie in [-1, 2]:
//     V(a,b) = R[a][b] - R[a+1][b],   H(a,b) = R[a][b] - R[a][b+1]          (2-bit code = value + 1)
//     u = H(a+1,b), w = V(a,b+1), s = +-1:   X = max(s, u-1, w-1),  V = X - u,  H = X - w
//     DIAG iff s >= u-1 and s >= w-1, else INS iff u >= w, else DEL          (the spec's tie order)
// so a whole anti-diagonal of the band -- 64 lattice points of one parity -- is four 64-bit bit-planes and
// one step is ~25 boolean operations on them, independent of the neighbouring lanes.  A wavefront
// advances 64 reads x 64 lattice points per step.
//
//   anti-diagonal s, bit t: diagonal d = 2t - 64 (+1 when s is odd), a = A0 - t, b = B0 + t,
//   A0 = (s + 64 - (s&1)) >> 1, B0 = s - A0.  even s: u = H_prev << 1, w = V_prev; odd s: u = H_prev,
//   w = V_prev >> 1.  The zero shifted in is code 0 (-1): it can never win, which is the band's -inf.
//   Free-exit points (a == tq or b == tt) are forced to V = H = 0 through one-hot "sentinel" planes that
//   travel with the sequence planes.
//
// Sequences are read as bit-planes (planar 2-bit packing: 32 bases = {lo word, hi word}); the query
// plane of a step is a 64-bit window of the bit-reversed read that slides by one base every second
// step, the target plane a window of the reference sliding the other way -- two v_alignbit per plane,
// with wave-uniform shift amounts.
//
// Traceback without storing 2 x 64 bits per step and read: pass 1 runs the tile from its far corner
// to the anchor and keeps the four difference planes at every 32nd anti-diagonal below 2(T-O) in LDS
// (32 B per lane and checkpoint).  Pass 2 takes the 32-step blocks in walk order: recompute the block
// from its checkpoint with the decision planes kept in registers, then walk through it -- every lane
// follows its own path with the steps predicated on "my path is on this anti-diagonal".
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "lrm_internal.h"

#define BS_K 32
#define BS_PADW LRM_BS_PADW

struct __attribute__((aligned(8))) BsPair { uint64_t a, b; };

__device__ __forceinline__ uint32_t bs_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}
__device__ __forceinline__ uint32_t bs_onehot(int x) { return (uint32_t) x < 32u ? (1u << x) : 0u; }

// ----------------------------------------------------------------------------------------
// planar packing: one lane per base, the two code bits of 64 bases are two ballots
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ void bs_pack_group(const uint8_t *src, uint64_t len, uint64_t g, int lane,
                                              uint64_t *out, uint32_t *flag) {
    const uint64_t p = g * 64 + (uint64_t) lane;
    const uint32_t c = p < len ? src[p] : (uint32_t) 'A';          // bases past the end pack as A, unflagged
    const uint32_t code = ((c >> 1) ^ (c >> 2)) & 3u;                      // A 0, C 1, T 2, G 3 (any bijection works)
    const bool bad = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');     // bytes compare by equality in the spec:
    const uint64_t lo = __ballot(code & 1u), hi = __ballot(code >> 1);    // anything else goes to the byte kernels
    const uint64_t nb = __ballot(bad);
    if (lane == 0) {
        out[2 * g] = (lo & 0xffffffffull) | (hi << 32);
        out[2 * g + 1] = (lo >> 32) | (hi & 0xffffffff00000000ull);
        if (nb && flag) atomicOr(flag, 1u);
    }
}

// reads: word w of read r at out + r*wpr + BS_PADW + w; padding words are zeroed
__global__ __launch_bounds__(256) void bs_pack_reads_kernel(const char *__restrict__ reads, uint64_t stride,
                                                            const uint32_t *__restrict__ lens,
                                                            uint64_t *__restrict__ out, uint64_t wpr,
                                                            uint32_t *__restrict__ flags, uint64_t n,
                                                            uint32_t groups_per_read, uint32_t waves_per_read) {
    const uint64_t wave_id = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t r = wave_id / waves_per_read;
    const uint32_t part = (uint32_t) (wave_id % waves_per_read);
    if (r >= n) return;
    const uint64_t len = lens[r];
    uint64_t *o = out + r * wpr;
    if (part == 0) {
        for (uint64_t w = lane; w < BS_PADW; w += 64) o[w] = 0;
        for (uint64_t w = BS_PADW + 2ull * groups_per_read + lane; w < wpr; w += 64) o[w] = 0;
    }
    const uint8_t *src = reinterpret_cast<const uint8_t *>(reads) + r * stride;
    for (uint32_t g = part; g < groups_per_read; g += waves_per_read)
        bs_pack_group(src, len, g, lane, o + BS_PADW, flags + r);
}

// reference text: one wavefront per 64 KiB
__global__ __launch_bounds__(256) void bs_pack_content_kernel(const char *__restrict__ content, uint64_t len,
                                                              uint64_t *__restrict__ out, uint64_t groups,
                                                              uint32_t *__restrict__ flag) {
    const uint64_t wave_id = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t g0 = wave_id * 1024;
    for (uint64_t g = g0; g < g0 + 1024 && g < groups; ++g)
        bs_pack_group(reinterpret_cast<const uint8_t *>(content), len, g, lane, out, flag);
}

// ----------------------------------------------------------------------------------------
// per-lane tile context and sequence streams
// ----------------------------------------------------------------------------------------
struct BsTile {
    const uint64_t *qpl;      // planar read, word 0
    const uint64_t *dpl;      // planar reference text, word 0
    int64_t dpos;             // loc + j: text position of the tile's anchor
    int32_t i;                // read position of the tile's anchor
    int32_t tq, tt;
};

struct BsWord { uint32_t lo, hi, sent; };

// stream word whose bit b holds read base a_hi - b (the read runs backwards through the planes)
__device__ __forceinline__ BsWord bs_q_word(const BsTile &t, int a_hi) {
    const int pos = t.i + a_hi - 31;
    const int k = pos >> 5;
    const uint32_t sh = (uint32_t) pos & 31u;
    BsPair p;
    __builtin_memcpy(&p, t.qpl + k, sizeof(p));
    BsWord w;
    w.lo = __builtin_bitreverse32(bs_alignbit((uint32_t) p.b, (uint32_t) p.a, sh));
    w.hi = __builtin_bitreverse32(bs_alignbit((uint32_t) (p.b >> 32), (uint32_t) (p.a >> 32), sh));
    w.sent = bs_onehot(a_hi - t.tq);
    return w;
}
// word whose bit b holds text base b_lo + b
__device__ __forceinline__ BsWord bs_d_word(const BsTile &t, int b_lo) {
    const int64_t pos = t.dpos + b_lo;
    const int64_t k = pos >> 5;
    const uint32_t sh = (uint32_t) pos & 31u;
    BsPair p;
    __builtin_memcpy(&p, t.dpl + k, sizeof(p));
    BsWord w;
    w.lo = bs_alignbit((uint32_t) p.b, (uint32_t) p.a, sh);
    w.hi = bs_alignbit((uint32_t) (p.b >> 32), (uint32_t) (p.a >> 32), sh);
    w.sent = bs_onehot(t.tt - b_lo);
    return w;
}

struct BsStream {
    BsWord q0, q1, q2, d0, d1, d2;     // three consecutive stream words each
    uint64_t Qlo, Qhi, Qs, Dlo, Dhi, Ds;
    int qnext, dnext;                  // wave-uniform: a of bit 0 of the next query word / b of the next (lower) text word
};

__device__ __forceinline__ uint64_t bs_win(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh) {
    return (uint64_t) bs_alignbit(w1, w0, sh) | ((uint64_t) bs_alignbit(w2, w1, sh) << 32);
}
__device__ __forceinline__ void bs_extract_q(BsStream &st, uint32_t shq) {
    st.Qlo = bs_win(st.q0.lo, st.q1.lo, st.q2.lo, shq);
    st.Qhi = bs_win(st.q0.hi, st.q1.hi, st.q2.hi, shq);
    st.Qs = bs_win(st.q0.sent, st.q1.sent, st.q2.sent, shq);
}
__device__ __forceinline__ void bs_extract_d(BsStream &st, uint32_t shd) {
    st.Dlo = bs_win(st.d0.lo, st.d1.lo, st.d2.lo, shd);
    st.Dhi = bs_win(st.d0.hi, st.d1.hi, st.d2.hi, shd);
    st.Ds = bs_win(st.d0.sent, st.d1.sent, st.d2.sent, shd);
}
// windows for anti-diagonal s; the text window starts room_d bases into its words
__device__ __forceinline__ void bs_stream_init(BsStream &st, const BsTile &t, int s, int room_d) {
    const int A0 = (s + 64 - (s & 1)) >> 1, B0 = s - A0;
    st.q0 = bs_q_word(t, A0);
    st.q1 = bs_q_word(t, A0 - 32);
    st.q2 = bs_q_word(t, A0 - 64);
    st.d0 = bs_d_word(t, B0 - room_d);
    st.d1 = bs_d_word(t, B0 - room_d + 32);
    st.d2 = bs_d_word(t, B0 - room_d + 64);
    st.qnext = A0 - 96;
    st.dnext = B0 - room_d - 32;
    bs_extract_q(st, 0);
    bs_extract_d(st, (uint32_t) room_d);
}

struct BsState { uint64_t V1, V0, H1, H0; };

// one anti-diagonal.  TRACK: also produce the decision planes (N: not diagonal; G: deletion if N, else mismatch)
template <bool ODD, bool TRACK>
__device__ __forceinline__ void bs_step(BsState &x, const BsStream &st, uint64_t &N, uint64_t &G) {
    uint64_t u1, u0, w1, w0;
    if (!ODD) { u1 = x.H1 << 1; u0 = x.H0 << 1; w1 = x.V1; w0 = x.V0; }
    else      { u1 = x.H1; u0 = x.H0; w1 = x.V1 >> 1; w0 = x.V0 >> 1; }
    const uint64_t m = ~((st.Qlo ^ st.Dlo) | (st.Qhi ^ st.Dhi));
    const uint64_t d0 = u0 ^ w0, b0 = ~u0 & w0, t1 = u1 ^ w1, d1 = t1 ^ b0;
    const uint64_t lt = (~u1 & w1) | (~t1 & b0);
    const uint64_t big = u1 | w1, nd = ~m & big, del = nd & lt, ins = nd & ~lt, n1 = d1 ^ d0;
    const uint64_t V1 = (m & ~u1) | (del & n1), V0 = (~nd & ~u0) | (del & d0);
    const uint64_t H1 = (m & ~w1) | (ins & d1), H0 = (~nd & ~w0) | (ins & d0);
    const uint64_t Bm = st.Qs | st.Ds;
    x.V1 = V1 & ~Bm; x.V0 = V0 | Bm; x.H1 = H1 & ~Bm; x.H0 = H0 | Bm;
    if (TRACK) { N = nd; G = del | ~(m | big); }
}

__device__ __forceinline__ int bs_wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

// ----------------------------------------------------------------------------------------
// the kernel: one wavefront per workgroup, lane = read
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gact_bs_kernel(const uint64_t *__restrict__ qpl, uint64_t wpr,
                                                     const uint32_t *__restrict__ lens,
                                                     const lrm_seq_meta *__restrict__ meta,
                                                     const int32_t *__restrict__ meta_r,
                                                     const uint64_t *__restrict__ cpl,
                                                     const uint32_t *__restrict__ tlens,
                                                     const uint32_t *__restrict__ rflags, uint64_t n_reads,
                                                     int T, int O, uint8_t *__restrict__ store,
                                                     uint64_t store_stride, int32_t *__restrict__ n_ops_out,
                                                     int32_t *__restrict__ score_out, LrmDevCounters *counters) {
    extern __shared__ __attribute__((aligned(16))) uint64_t ck[];      // [checkpoint][plane][lane]
    const int lane = threadIdx.x;
    const uint64_t r = (uint64_t) blockIdx.x * 64 + (uint64_t) lane;
    const uint64_t rs = r < n_reads ? r : 0;
    const bool fenced = r < n_reads && meta_r[rs] == 0;
    bool alive = r < n_reads && !fenced && !(rflags && rflags[rs]);
    if (fenced) { n_ops_out[r] = 0; score_out[r] = -1; }
    const int n = alive ? (int) lens[rs] : 0;
    const int m = alive ? (tlens ? (int) tlens[rs] : n) : 0;
    const int64_t loc = alive ? (int64_t) meta[rs].loc : 0;
    uint8_t *out = store + rs * store_stride;
    BsTile t;
    t.qpl = qpl + rs * wpr + BS_PADW;
    t.dpl = cpl;

    const int cap = T - O, lim2 = 2 * cap;
    const int nblk = (lim2 + BS_K - 1) / BS_K;
    int i = 0, j = 0, cnt = 0, score = 0;
    uint32_t wbuf = 0;
    unsigned tiles = 0;

#define BS_EMIT(OP) do {                                                             \
        wbuf |= (uint32_t) (OP) << ((cnt & 3) * 8);                                      \
        cnt++;                                                                           \
        if ((cnt & 3) == 0) { *reinterpret_cast<uint32_t *>(out + cnt - 4) = wbuf; wbuf = 0; } \
    } while (0)

    while (true) {
        if (alive && (score < 0 || !(i < n && j < m))) {          // this lane's read is finished: write it out
            if (score >= 0) while (i < n) { BS_EMIT('I'); score++; i++; }     // text exhausted
            const int tail = cnt & 3;
            for (int e = 0; e < tail; ++e) out[cnt - tail + e] = (uint8_t) (wbuf >> (8 * e));
            n_ops_out[r] = score >= 0 ? cnt : 0;
            score_out[r] = score;
            alive = false;
        }
        const bool act = alive;
        const uint64_t actmask = __ballot(act);
        if (actmask == 0) break;
        tiles += (unsigned) __popcll(actmask);
        t.i = i;
        t.dpos = loc + j;
        t.tq = act ? min(T, n - i) : 0;
        t.tt = act ? min(T, m - j) : 0;
        const bool last = i + t.tq == n;
        const int S0 = (bs_wave_max(t.tq + t.tt) + BS_K - 1) & ~(BS_K - 1);
        const int nb = min(nblk, S0 / BS_K);

        BsState x = {0ull, ~0ull, 0ull, ~0ull};
        BsStream st;
        uint64_t nN, nG;
        // ---- pass 1: differences only, checkpoints at the block boundaries ----
        bs_stream_init(st, t, S0, 31);
        uint32_t shq = 0, shd = 31;
        for (int s = S0;; s -= 2) {
            bs_step<false, false>(x, st, nN, nG);
            if ((s & (BS_K - 1)) == 0 && (s >> 5) <= nb) {
                uint64_t *c = ck + (size_t) (s >> 5) * 256 + lane;
                c[0] = x.V1; c[64] = x.V0; c[128] = x.H1; c[192] = x.H0;
            }
            if (s == BS_K) break;
            if (++shq == 32) {
                st.q0 = st.q1; st.q1 = st.q2; st.q2 = bs_q_word(t, st.qnext);
                st.qnext -= 32;
                shq = 0;
            }
            bs_extract_q(st, shq);
            bs_step<true, false>(x, st, nN, nG);
            if (shd == 0) {
                st.d2 = st.d1; st.d1 = st.d0; st.d0 = bs_d_word(t, st.dnext);
                st.dnext -= 32;
                shd = 32;
            }
            --shd;
            bs_extract_d(st, shd);
        }
        // ---- pass 2: per block recompute with decision planes, then walk through the block ----
        int a = 0, b = 0;
        bool running = act;
        for (int c = 0; c < nb; ++c) {
            if (__ballot(running) == 0) break;
            {
                const uint64_t *cp = ck + (size_t) (c + 1) * 256 + lane;
                x.V1 = cp[0]; x.V0 = cp[64]; x.H1 = cp[128]; x.H0 = cp[192];
            }
            bs_stream_init(st, t, BS_K * c + BS_K - 1, 16);
            uint64_t N[BS_K], G[BS_K];
#pragma unroll
            for (int k = BS_K - 1; k >= 1; k -= 2) {
                bs_step<true, true>(x, st, N[k], G[k]);
                bs_extract_d(st, (uint32_t) ((k - 1) >> 1));                 // 15 .. 0
                bs_step<false, true>(x, st, N[k - 1], G[k - 1]);
                if (k > 1) bs_extract_q(st, (uint32_t) (16 - ((k - 1) >> 1)));   // 1 .. 15
            }
            int sw = a + b - BS_K * c;
#pragma unroll
            for (int k = 0; k < BS_K; ++k) {
                if (running && sw == k) {
                    const uint32_t tpos = (uint32_t) (b - a + 64) >> 1;
                    const uint32_t nbit = (uint32_t) (N[k] >> tpos) & 1u, gbit = (uint32_t) (G[k] >> tpos) & 1u;
                    const uint32_t op = nbit ? (gbit ? 'D' : 'I') : (gbit ? 'X' : '=');
                    BS_EMIT(op);
                    score += (int) (nbit | gbit);
                    a += (int) (1u ^ (nbit & gbit));
                    b += (int) (1u ^ (nbit & (gbit ^ 1u)));
                    sw = a + b - BS_K * c;
                    running = a < t.tq && b < t.tt && (last ? (a + b < lim2) : (a < cap && b < cap));
                }
            }
        }
        if (act) {
            i += a;
            j += b;
            if (a + b == 0) score = -1;                           // cannot happen (every walk moves); never spin
        }
    }
#undef BS_EMIT
    if (lane == 0 && tiles) atomicAdd(&counters->gact_tiles, (unsigned long long) tiles);
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

uint64_t lrm_bs_planar_words(uint64_t len) { return 2 * ((len + 63) / 64) + 2 * (uint64_t) BS_PADW; }

bool lrm_bs_wanted(lrm_gact_params gp, uint64_t n) {
    const char *e = getenv("LRM_GACT_IMPL");
    const int impl = e ? atoi(e) : 0;
    return gp.W == 128 && (impl == 4 || (impl == 0 && n >= LRM_BS_MIN_READS));
}

// planar text into a caller-provided buffer of lrm_bs_planar_words(len) words (+ a flag word)
int lrm_bs_pack_text(const char *d_text, uint64_t len, uint64_t *d_out, uint32_t *d_flag, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    const uint64_t words = lrm_bs_planar_words(len), groups = (len + 63) / 64;
    HIPCHK(hipMemsetAsync(d_out, 0, words * 8, stream));
    HIPCHK(hipMemsetAsync(d_flag, 0, 4, stream));
    if (groups == 0) return 0;
    const uint64_t waves = (groups + 1023) / 1024;
    hipLaunchKernelGGL(bs_pack_content_kernel, dim3((uint32_t) ((waves + 3) / 4)), dim3(256), 0, stream, d_text, len,
                       d_out + BS_PADW, groups, d_flag);
    HIPCHK(hipGetLastError());
    return 0;
}

// planar copy of the reference text, owned by the index handle (device memory, +25 % of the text)
int lrm_bs_prepare_index(lrm_index *idx) {
    idx->d_cpl = nullptr;
    idx->cpl_ok = 0;
    // the text ends with the FM terminator (accaln.c: content length == fmi length); no window reaches it
    const uint64_t len = idx->view.con_len > 0 ? idx->view.con_len - 1 : 0;
    if (len == 0) return 0;
    const uint64_t words = lrm_bs_planar_words(len), groups = (len + 63) / 64;
    uint64_t *d = nullptr;
    uint32_t *flag = nullptr;
    HIPCHK(hipMalloc(&d, words * 8 + 16));
    HIPCHK(hipMalloc(&flag, 16));
    (void) groups;
    if (lrm_bs_pack_text(idx->view.content, len, d, flag, nullptr)) return -1;
    uint32_t h = 0;
    HIPCHK(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
    (void) hipFree(flag);
    idx->d_cpl = d;
    idx->cpl_ok = h == 0;          // a text with bytes other than ACGT keeps the byte kernels
    return 0;
}

void lrm_bs_free_index(lrm_index *idx) {
    if (idx->d_cpl) (void) hipFree(idx->d_cpl);
    idx->d_cpl = nullptr;
}

// planar reads + per-read "has a byte other than ACGT" flags into caller-provided buffers
int lrm_bs_pack_reads(const char *d_reads, uint64_t stride, const uint32_t *d_lens, uint64_t n, uint32_t max_len,
                      uint64_t *d_qpl, uint64_t wpr, uint32_t *d_flags, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    const uint32_t gpr = (max_len + 63) / 64;
    uint32_t wv = (gpr + 15) / 16;                     // ~16 groups (1 KiB of read) per wavefront
    if (wv == 0) wv = 1;
    const uint64_t waves = n * wv;
    if ((waves + 3) / 4 > 0x7fffffffull) { lrm_set_error("planar pack grid too large: split the batch"); return -1; }
    HIPCHK(hipMemsetAsync(d_flags, 0, n * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(bs_pack_reads_kernel, dim3((uint32_t) ((waves + 3) / 4)), dim3(256), 0, stream, d_reads, stride,
                       d_lens, d_qpl, wpr, d_flags, n, gpr, wv);
    return 0;
}

int lrm_bs_launch(const uint64_t *d_qpl, uint64_t wpr, const uint32_t *d_lens, const lrm_seq_meta *d_meta,
                  const int32_t *d_meta_r, const uint64_t *d_cpl, const uint32_t *d_tlens, const uint32_t *d_flags,
                  uint64_t n, int T, int O, uint8_t *d_store, uint64_t store_stride, int32_t *d_n_ops,
                  int32_t *d_score, LrmDevCounters *counters, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    const int nblk = (2 * (T - O) + BS_K - 1) / BS_K;
    const size_t shmem = (size_t) (nblk + 1) * 256 * 8;
    const uint64_t blocks = (n + 63) / 64;
    if (blocks > 0x7fffffffull) { lrm_set_error("gact grid too large: split the batch"); return -1; }
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gact_bs_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem);
        if (e != hipSuccess) { lrm_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", shmem, hipGetErrorString(e)); return -1; }
    }
    hipLaunchKernelGGL(gact_bs_kernel, dim3((uint32_t) blocks), dim3(64), shmem, stream, d_qpl, wpr, d_lens, d_meta,
                       d_meta_r, d_cpl + BS_PADW, d_tlens, d_flags, n, T, O, d_store, store_stride, d_n_ops, d_score,
                       counters);
    return 0;
}
