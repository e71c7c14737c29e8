// SPDX-License-Identifier: MIT
// Bit-sliced GACT for bands of up to 128 diagonals (the default): ONE LANE PER READ, 64 lattice points per word.
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

#ifndef BS_K
#define BS_K 32                 // anti-diagonals per traceback block (even, <= 32)
#endif
#define BS_H (BS_K / 2)
#define LRM_BS_MAX_WAVES 2048ull  // 2 per SIMD on 256 CUs
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

// reads: word w of read r at out + r*wpr + BS_PADW + w; padding words are zeroed.
// One wavefront per BS_PACK_G groups of 64 bases (1 KiB of a read): every lane takes 16 bases with two ALIGNED
// 16-byte loads (rows start at any byte: stride = max_read_len + 1) and v_alignbyte, turns them into 16 low and 16
// high plane bits with multiplies (no ballots), and pairs of lanes assemble the 64-bit words.  (The first version
// loaded one byte per lane and built the planes with 48 ballots per KiB: 0.72 ms per Gbp [r2].)
#define BS_PACK_G 16
__device__ __forceinline__ uint32_t bs_bytes_equal(uint32_t x, uint32_t c4) {    // 0x80 in every byte of x equal to c4's
    const uint32_t z = x ^ c4;
    return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);
}
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
    const uint32_t len = lens[r];
    uint64_t *o = out + r * wpr;
    if (part == 0) {
        for (uint64_t w = lane; w < BS_PADW; w += 64) o[w] = 0;
        for (uint64_t w = BS_PADW + 2ull * groups_per_read + lane; w < wpr; w += 64) o[w] = 0;
    }
    const uint8_t *row = reinterpret_cast<const uint8_t *>(reads) + r * stride;
    const uint32_t g0 = part * BS_PACK_G;
    const uint32_t p0 = g0 * 64 + 16 * (uint32_t) lane;               // this lane's 16 bases: [p0, p0 + 16)
    uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint8_t *src = row + p0;
    const uint32_t sh = (uint32_t) ((uintptr_t) src & 15u);           // the same in every lane of the wavefront
    if (p0 < len) {                                                   // both loads touch a 16-byte line that holds a base of the read
        const uint4 q0 = *reinterpret_cast<const uint4 *>(src - sh);
        d[0] = q0.x; d[1] = q0.y; d[2] = q0.z; d[3] = q0.w;
        if (sh && p0 + (16 - sh) < len) {
            const uint4 q1 = *reinterpret_cast<const uint4 *>(src - sh + 16);
            d[4] = q1.x; d[5] = q1.y; d[6] = q1.z; d[7] = q1.w;
        }
    }
    const uint32_t ws = sh >> 2, bs = sh & 3;                          // dword and byte part of the shift (wave-uniform)
    uint32_t lo = 0, hi = 0, bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t a_ = d[k], b_ = d[k + 1];
        if (ws == 1) { a_ = d[k + 1]; b_ = d[k + 2]; }
        else if (ws == 2) { a_ = d[k + 2]; b_ = d[k + 3]; }
        else if (ws == 3) { a_ = d[k + 3]; b_ = d[k + 4]; }
        uint32_t x = __builtin_amdgcn_alignbyte(b_, a_, bs);           // bases p0 + 4k .. p0 + 4k + 3
        const uint32_t pk = p0 + 4 * (uint32_t) k;
        const uint32_t nv = len > pk ? (len - pk < 4 ? len - pk : 4) : 0;          // bases of this dword inside the read
        const uint32_t keep = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
        x = (x & keep) | (0x41414141u & ~keep);                        // bases past the end pack as A, unflagged
        const uint32_t code = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        lo |= ((((code & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * k);
        hi |= (((((code >> 1) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * k);
        const uint32_t ok = bs_bytes_equal(x, 0x41414141u) | bs_bytes_equal(x, 0x43434343u) |
                            bs_bytes_equal(x, 0x47474747u) | bs_bytes_equal(x, 0x54545454u);
        bad |= ~ok & 0x80808080u;
    }
    const uint32_t mine = lo | (hi << 16);
    const uint32_t other = (uint32_t) __shfl_xor((int) mine, 1);
    if (!(lane & 1) && g0 + (uint32_t) (lane >> 2) < groups_per_read) {
        // word (lane >> 1) of the wavefront's 32: low planes of 32 bases in the low half, high planes in the high half
        const uint64_t w = (uint64_t) (mine & 0xFFFFu) | ((uint64_t) (other & 0xFFFFu) << 16) |
                           ((uint64_t) (mine >> 16) << 32) | ((uint64_t) (other >> 16) << 48);
        o[BS_PADW + 2ull * g0 + (uint32_t) (lane >> 1)] = w;
    }
    if (__ballot(bad != 0) && lane == 0) atomicOr(flags + r, 1u);
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

// stream word whose bit b holds read base a_hi - b (the read runs backwards through the planes):
// a 16-byte gather of two adjacent planar words, then a funnel shift by the lane's own alignment
__device__ __forceinline__ BsPair bs_q_raw(const BsTile &t, int a_hi) {
    BsPair p;
    __builtin_memcpy(&p, t.qpl + ((t.i + a_hi - 31) >> 5), sizeof(p));
    return p;
}
__device__ __forceinline__ BsWord bs_q_conv(const BsTile &t, int a_hi, const BsPair &p) {
    const uint32_t sh = (uint32_t) (t.i + a_hi - 31) & 31u;
    BsWord w;
    w.lo = __builtin_bitreverse32(bs_alignbit((uint32_t) p.b, (uint32_t) p.a, sh));
    w.hi = __builtin_bitreverse32(bs_alignbit((uint32_t) (p.b >> 32), (uint32_t) (p.a >> 32), sh));
    w.sent = bs_onehot(a_hi - t.tq);
    return w;
}
__device__ __forceinline__ BsWord bs_q_word(const BsTile &t, int a_hi) { return bs_q_conv(t, a_hi, bs_q_raw(t, a_hi)); }
// word whose bit b holds text base b_lo + b
__device__ __forceinline__ BsPair bs_d_raw(const BsTile &t, int b_lo) {
    BsPair p;
    __builtin_memcpy(&p, t.dpl + ((t.dpos + b_lo) >> 5), sizeof(p));
    return p;
}
__device__ __forceinline__ BsWord bs_d_conv(const BsTile &t, int b_lo, const BsPair &p) {
    const uint32_t sh = (uint32_t) (t.dpos + b_lo) & 31u;
    BsWord w;
    w.lo = bs_alignbit((uint32_t) p.b, (uint32_t) p.a, sh);
    w.hi = bs_alignbit((uint32_t) (p.b >> 32), (uint32_t) (p.a >> 32), sh);
    w.sent = bs_onehot(t.tt - b_lo);
    return w;
}
__device__ __forceinline__ BsWord bs_d_word(const BsTile &t, int b_lo) { return bs_d_conv(t, b_lo, bs_d_raw(t, b_lo)); }

struct BsPl { uint32_t lo, hi; };                    // one bit-plane of an anti-diagonal: 64 lattice points

struct BsStream {
    BsWord q0, q1, q2, d0, d1, d2;     // three consecutive stream words each
    BsPl Qlo, Qhi, Qs, Dlo, Dhi, Ds;
    BsPl bandE, bandO;                 // lattice points inside the band on even / odd anti-diagonals (all ones for W = 128)
    bool narrow;                       // W < 128: every step takes the masked (BOUND) form
    int qnext, dnext;                  // wave-uniform: a of bit 0 of the next query word / b of the next (lower) text word
};

template <bool BOUND>
__device__ __forceinline__ void bs_extract_q(BsStream &st, uint32_t shq) {
    st.Qlo.lo = bs_alignbit(st.q1.lo, st.q0.lo, shq);  st.Qlo.hi = bs_alignbit(st.q2.lo, st.q1.lo, shq);
    st.Qhi.lo = bs_alignbit(st.q1.hi, st.q0.hi, shq);  st.Qhi.hi = bs_alignbit(st.q2.hi, st.q1.hi, shq);
    if (BOUND) { st.Qs.lo = bs_alignbit(st.q1.sent, st.q0.sent, shq); st.Qs.hi = bs_alignbit(st.q2.sent, st.q1.sent, shq); }
}
template <bool BOUND>
__device__ __forceinline__ void bs_extract_d(BsStream &st, uint32_t shd) {
    st.Dlo.lo = bs_alignbit(st.d1.lo, st.d0.lo, shd);  st.Dlo.hi = bs_alignbit(st.d2.lo, st.d1.lo, shd);
    st.Dhi.lo = bs_alignbit(st.d1.hi, st.d0.hi, shd);  st.Dhi.hi = bs_alignbit(st.d2.hi, st.d1.hi, shd);
    if (BOUND) { st.Ds.lo = bs_alignbit(st.d1.sent, st.d0.sent, shd); st.Ds.hi = bs_alignbit(st.d2.sent, st.d1.sent, shd); }
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
    bs_extract_q<true>(st, 0);
    bs_extract_d<true>(st, (uint32_t) room_d);
}
// does any lane hold a free-exit point in its current stream words?  (wave-uniform)
__device__ __forceinline__ bool bs_any_sentinel(const BsStream &st) {
    if (st.narrow) return true;
    return __ballot((st.q0.sent | st.q1.sent | st.q2.sent | st.d0.sent | st.d1.sent | st.d2.sent) != 0u) != 0ull;
}

struct BsState { BsPl V1, V0, H1, H0; };

// gfx950 v_bitop3_b32: any boolean function of three words in one instruction; the table is the
// function evaluated on A = 0xF0, B = 0xCC, C = 0xAA
enum : uint32_t { TA = 0xF0u, TB = 0xCCu, TC = 0xAAu };
#define BS_LOP3(a, b, c, EXPR) __builtin_amdgcn_bitop3_b32((a), (b), (c), (uint32_t) (EXPR) & 0xFFu)

// one 32-bit half of an anti-diagonal: u = H of the lower neighbour, w = V of the upper one
template <bool BOUND, bool TRACK>
__device__ __forceinline__ void bs_half(uint32_t u1, uint32_t u0, uint32_t w1, uint32_t w0, uint32_t ql, uint32_t qh,
                                        uint32_t dl, uint32_t dh, uint32_t bm, uint32_t band, uint32_t &V1, uint32_t &V0,
                                        uint32_t &H1, uint32_t &H0, uint32_t &N, uint32_t &G) {
    const uint32_t e1 = ql ^ dl;
    const uint32_t m = BS_LOP3(e1, qh, dh, ~TA & ~(TB ^ TC));              // bases equal
    const uint32_t b0 = BS_LOP3(u0, w0, w0, ~TA & TB);                     // borrow of the low code bit
    const uint32_t lt = BS_LOP3(u1, w1, b0, ((TA ^ TB) & TB) | (~(TA ^ TB) & TC));   // u < w
    const uint32_t nd = BS_LOP3(m, u1, w1, ~TA & (TB | TC));               // a gap beats the mismatch diagonal
    const uint32_t d0 = u0 ^ w0;
    const uint32_t d1 = BS_LOP3(u1, w1, b0, TA ^ TB ^ TC);                 // (u - w) mod 4 = d1 d0
    // high bit of w - u for u < w: w - u >= 2 needs w >= 1 (code >= 2), i.e. w1 set, so the lattice point is a gap point
    // (nd) unless the bases match -- `nd & lt` need not be formed: the match bit gates it in the next instruction
    const uint32_t a = BS_LOP3(lt, d1, d0, TA & (TB ^ TC));
    uint32_t v1 = BS_LOP3(a, m, u1, (TA & ~TB) | (TB & ~TC));               // match ? ~u1 : a
    const uint32_t hh = BS_LOP3(nd, lt, d1, TA & ~TB & TC);
    uint32_t h1 = BS_LOP3(hh, m, w1, TA | (TB & ~TC));
    const uint32_t x = BS_LOP3(nd, lt, d0, TA & TB & TC);
    uint32_t v0 = BS_LOP3(x, nd, u0, TA | (~(TB | TC) & 0xFFu));
    const uint32_t xx = BS_LOP3(nd, lt, d0, TA & ~TB & TC);
    uint32_t h0 = BS_LOP3(xx, nd, w0, TA | (~(TB | TC) & 0xFFu));
    if (BOUND) {                 // free-exit points: V = H = 0 (code 1); lattice points outside a band narrower
                                 // than the 128 diagonals of the planes: code 0 (-1), the value that never wins
        v1 = BS_LOP3(v1, bm, band, TA & ~TB & TC);  v0 = BS_LOP3(v0, bm, band, (TA | TB) & TC);
        h1 = BS_LOP3(h1, bm, band, TA & ~TB & TC);  h0 = BS_LOP3(h0, bm, band, (TA | TB) & TC);
    }
    V1 = v1; V0 = v0; H1 = h1; H0 = h0;
    if (TRACK) { N = nd; G = BS_LOP3(nd, lt, m, (TA & TB) | (~TA & TC)); }   // G: deletion if N, else MATCH
}

// one anti-diagonal.  TRACK: also produce the decision planes
template <bool ODD, bool BOUND, bool TRACK>
__device__ __forceinline__ void bs_step(BsState &x, const BsStream &st, BsPl &N, BsPl &G) {
    BsPl u1, u0, w1, w0;
    if (!ODD) {                        // u = H << 1 (the lowest lattice point has no insertion neighbour in the band)
        u1.lo = x.H1.lo << 1; u1.hi = bs_alignbit(x.H1.hi, x.H1.lo, 31);
        u0.lo = x.H0.lo << 1; u0.hi = bs_alignbit(x.H0.hi, x.H0.lo, 31);
        w1 = x.V1; w0 = x.V0;
    } else {                           // w = V >> 1 (the highest one has no deletion neighbour)
        u1 = x.H1; u0 = x.H0;
        w1.lo = bs_alignbit(x.V1.hi, x.V1.lo, 1); w1.hi = x.V1.hi >> 1;
        w0.lo = bs_alignbit(x.V0.hi, x.V0.lo, 1); w0.hi = x.V0.hi >> 1;
    }
    const uint32_t bl = BOUND ? (st.Qs.lo | st.Ds.lo) : 0u, bh = BOUND ? (st.Qs.hi | st.Ds.hi) : 0u;
    const BsPl &band = ODD ? st.bandO : st.bandE;
    bs_half<BOUND, TRACK>(u1.lo, u0.lo, w1.lo, w0.lo, st.Qlo.lo, st.Qhi.lo, st.Dlo.lo, st.Dhi.lo, bl, band.lo,
                          x.V1.lo, x.V0.lo, x.H1.lo, x.H0.lo, N.lo, G.lo);
    bs_half<BOUND, TRACK>(u1.hi, u0.hi, w1.hi, w0.hi, st.Qlo.hi, st.Qhi.hi, st.Dlo.hi, st.Dhi.hi, bh, band.hi,
                          x.V1.hi, x.V0.hi, x.H1.hi, x.H0.hi, N.hi, G.hi);
}

__device__ __forceinline__ uint64_t bs_bit_range(int lo, int hi) {       // bits lo..hi of a 64-bit word, 0 <= lo <= hi <= 63
    return (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
}

__device__ __forceinline__ int bs_wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

// ----------------------------------------------------------------------------------------
// the kernel: one wavefront per workgroup, lane = read
// ----------------------------------------------------------------------------------------
// everything a traceback block needs from memory, fetched one block ahead (during the previous walk)
struct BsBlockRaw { BsPair q0, q1, q2, d0, d1, d2; uint32_t ck[8]; };

__device__ __forceinline__ void bs_block_prefetch(BsBlockRaw &raw, const BsTile &t, int c, const uint32_t *ckw, int lane) {
    const int A0 = BS_H * (c + 1) + 31, b_lo = BS_H * c - 32;      // anti-diagonal K(c+1)-1: A0, and B0 - K/2
    raw.q0 = bs_q_raw(t, A0); raw.q1 = bs_q_raw(t, A0 - 32); raw.q2 = bs_q_raw(t, A0 - 64);
    raw.d0 = bs_d_raw(t, b_lo); raw.d1 = bs_d_raw(t, b_lo + 32); raw.d2 = bs_d_raw(t, b_lo + 64);
    const uint32_t *cp = ckw + (size_t) c * 512 + lane;      // checkpoint c+1 = state after anti-diagonal K(c+1)
#pragma unroll
    for (int e = 0; e < 8; ++e) raw.ck[e] = cp[64 * e];
}

__global__ __launch_bounds__(64) void gact_bs_kernel(const uint64_t *__restrict__ qpl, uint64_t wpr,
                                                     const uint32_t *__restrict__ lens,
                                                     const lrm_seq_meta *__restrict__ meta,
                                                     const int32_t *__restrict__ meta_r,
                                                     const uint64_t *__restrict__ cpl,
                                                     const uint32_t *__restrict__ tlens,
                                                     const uint32_t *__restrict__ rflags, uint64_t n_reads,
                                                     int T, int O, int W, uint32_t *__restrict__ ckpt,
                                                     uint64_t *__restrict__ codes, uint64_t cw,
                                                     int32_t *__restrict__ n_codes_out,
                                                     int32_t *__restrict__ n_ops_out,
                                                     int32_t *__restrict__ score_out, LrmDevCounters *counters) {
    const int lane = threadIdx.x;
    // Lanes take reads from a queue (one atomic per wavefront and refill): reads of very different
    // lengths share a wavefront without the short ones idling behind the longest, and a grid of a few
    // resident wavefronts per SIMD serves any batch size.
    unsigned long long *queue = &counters->reserved[0];
    uint64_t r = 0;
    bool alive = false, exhausted = false;
    int n = 0, m = 0;
    int64_t loc = 0;
    uint64_t *cout = codes;                                     // 2-bit CIGAR codes, 32 per word
    BsTile t;
    t.qpl = qpl + BS_PADW;
    t.dpl = cpl;

    const int cap = T - O, lim2 = 2 * cap;
    const int nblk = (lim2 + BS_K - 1) / BS_K;
    uint32_t *ckw = ckpt + (size_t) blockIdx.x * (size_t) nblk * 512;   // this wavefront's checkpoints (L2-resident scratch)
    // band -W/2 <= b - a <= W/2 - 1 as bit ranges of the planes (even: d = 2t - 64, odd: d = 2t - 63)
    const int hw = W / 2;
    const uint64_t bandE64 = bs_bit_range((65 - hw) >> 1, (63 + hw) >> 1);
    const uint64_t bandO64 = bs_bit_range((64 - hw) >> 1, (62 + hw) >> 1);
    int i = 0, j = 0, cnt = 0, score = 0;
    uint64_t sb = 0;               // code stream buffer: `fill` bits used
    int fill = 0, widx = 0;
    uint64_t pend = 0;             // a full code word whose store is delayed past the next block's loads
    bool has_pend = false;
    unsigned tiles = 0;

    while (true) {
        if (alive && (score < 0 || !(i < n && j < m))) {          // this lane's read is finished: write it out
            if (has_pend) { cout[widx++] = pend; has_pend = false; }
            if (fill > 0) cout[widx] = sb;
            const int rest = score >= 0 ? n - i : 0;               // text exhausted: the expansion appends 'I's
            n_codes_out[r] = score >= 0 ? cnt : 0;
            n_ops_out[r] = score >= 0 ? cnt + rest : 0;
            score_out[r] = score >= 0 ? score + rest : score;
            alive = false;
        }
        // refill idle lanes from the queue
        while (true) {
            const bool need = !alive && !exhausted;
            const uint64_t needmask = __ballot(need);
            if (needmask == 0) break;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(queue, (unsigned long long) __popcll(needmask));
            base = __shfl(base, 0);
            if (need) {
                r = base + (uint64_t) __popcll(needmask & ((1ull << lane) - 1ull));
                if (r >= n_reads) {
                    exhausted = true;
                } else if (meta_r[r] == 0) {                       // fenced: no extension
                    n_ops_out[r] = 0; score_out[r] = -1; n_codes_out[r] = 0;
                } else if (!(rflags && rflags[r])) {               // flagged reads belong to the byte kernel
                    n = (int) lens[r];
                    m = tlens ? (int) tlens[r] : n;
                    loc = (int64_t) meta[r].loc;
                    cout = codes + r * cw;
                    t.qpl = qpl + r * wpr + BS_PADW;
                    i = j = cnt = score = 0;
                    sb = 0; fill = 0; widx = 0;
                    alive = true;
                    if (!(n > 0 && m > 0)) {                       // nothing to align: only the 'I' tail
                        n_codes_out[r] = 0; n_ops_out[r] = n; score_out[r] = n;
                        alive = false;
                    }
                }
            }
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
        const int S0 = ((bs_wave_max(t.tq + t.tt) + BS_K - 1) / BS_K) * BS_K;
        const int nb = min(nblk, S0 / BS_K);

        BsState x = {{0u, 0u}, {~0u, ~0u}, {0u, 0u}, {~0u, ~0u}};
        BsStream st;
        st.bandE.lo = (uint32_t) bandE64; st.bandE.hi = (uint32_t) (bandE64 >> 32);
        st.bandO.lo = (uint32_t) bandO64; st.bandO.hi = (uint32_t) (bandO64 >> 32);
        st.narrow = W < 128;
        BsPl nN, nG;
        // ---- pass 1: differences only, checkpoints at the block boundaries ----
        bs_stream_init(st, t, S0, 31);
        BsPair pq = bs_q_raw(t, st.qnext), pd = bs_d_raw(t, st.dnext);      // next stream words, one refill ahead
        uint32_t shq = 0, shd = 31;
        bool hb = bs_any_sentinel(st);
        int ck_s = nb * BS_K;                                      // next anti-diagonal whose state is kept
        for (int s = S0;; s -= 2) {
            if (hb) bs_step<false, true, false>(x, st, nN, nG); else bs_step<false, false, false>(x, st, nN, nG);
            if (s == ck_s) {
                uint32_t *c_ = ckw + (size_t) (s / BS_K - 1) * 512 + lane;
                c_[0] = x.V1.lo; c_[64] = x.V1.hi; c_[128] = x.V0.lo; c_[192] = x.V0.hi;
                c_[256] = x.H1.lo; c_[320] = x.H1.hi; c_[384] = x.H0.lo; c_[448] = x.H0.hi;
                ck_s -= BS_K;
            }
            if (s == BS_K) break;
            if (++shq == 32) {
                st.q0 = st.q1; st.q1 = st.q2; st.q2 = bs_q_conv(t, st.qnext, pq);
                st.qnext -= 32;
                pq = bs_q_raw(t, st.qnext);
                shq = 0;
                hb = bs_any_sentinel(st);
                if (hb) bs_extract_d<true>(st, shd);          // the text sentinel window may not be current
            }
            if (hb) {
                bs_extract_q<true>(st, shq);
                bs_step<true, true, false>(x, st, nN, nG);
            } else {
                bs_extract_q<false>(st, shq);
                bs_step<true, false, false>(x, st, nN, nG);
            }
            if (shd == 0) {
                st.d2 = st.d1; st.d1 = st.d0; st.d0 = bs_d_conv(t, st.dnext, pd);
                st.dnext -= 32;
                pd = bs_d_raw(t, st.dnext);
                shd = 32;
                hb = bs_any_sentinel(st);
                if (hb) bs_extract_q<true>(st, shq);          // the query sentinel window may not be current
            }
            --shd;
            if (hb) bs_extract_d<true>(st, shd); else bs_extract_d<false>(st, shd);
        }
        // ---- pass 2: per block recompute with decision planes, then walk through the block ----
        int a = 0, b = 0;
        bool running = act;
        // the walk keeps at most T-O bases of either sequence; in the read's last tile it may run on to the edge
        // but not past anti-diagonal 2(T-O) (docs/GACT_SPEC.md)
        const int amax = last ? t.tq : min(t.tq, cap), bmax = last ? t.tt : min(t.tt, cap);
        const int smax = last ? lim2 : 0x7fffffff;
        BsBlockRaw raw;
        bs_block_prefetch(raw, t, 0, ckw, lane);
        for (int c = 0; c < nb; ++c) {
            if (__ballot(running) == 0) break;
            {
                const int A0 = BS_H * (c + 1) + 31, b_lo = BS_H * c - 32;
                st.q0 = bs_q_conv(t, A0, raw.q0); st.q1 = bs_q_conv(t, A0 - 32, raw.q1); st.q2 = bs_q_conv(t, A0 - 64, raw.q2);
                st.d0 = bs_d_conv(t, b_lo, raw.d0); st.d1 = bs_d_conv(t, b_lo + 32, raw.d1); st.d2 = bs_d_conv(t, b_lo + 64, raw.d2);
                x.V1.lo = raw.ck[0]; x.V1.hi = raw.ck[1]; x.V0.lo = raw.ck[2]; x.V0.hi = raw.ck[3];
                x.H1.lo = raw.ck[4]; x.H1.hi = raw.ck[5]; x.H0.lo = raw.ck[6]; x.H0.hi = raw.ck[7];
                bs_extract_q<true>(st, 0);
                bs_extract_d<true>(st, BS_H);
            }
            // the next block's words and checkpoint are fetched a whole block ahead (248 VGPRs: still two wavefronts per SIMD;
            // behind the recompute, with only the walk in between: 11.72 against 11.57 ms)
            if (c + 1 < nb) bs_block_prefetch(raw, t, c + 1, ckw, lane);
            // the previous block's full code word goes out here, behind the loads it must not delay
            if (has_pend) { cout[widx++] = pend; has_pend = false; }
            BsPl N[BS_K], G[BS_K];
            if (bs_any_sentinel(st)) {
#pragma unroll
                for (int k = BS_K - 1; k >= 1; k -= 2) {
                    bs_step<true, true, true>(x, st, N[k], G[k]);
                    bs_extract_d<true>(st, (uint32_t) ((k - 1) >> 1));                 // K/2-1 .. 0
                    bs_step<false, true, true>(x, st, N[k - 1], G[k - 1]);
                    if (k > 1) bs_extract_q<true>(st, (uint32_t) (BS_H - ((k - 1) >> 1)));   // 1 .. K/2-1
                }
            } else {
#pragma unroll
                for (int k = BS_K - 1; k >= 1; k -= 2) {
                    bs_step<true, false, true>(x, st, N[k], G[k]);
                    bs_extract_d<false>(st, (uint32_t) ((k - 1) >> 1));
                    bs_step<false, false, true>(x, st, N[k - 1], G[k - 1]);
                    if (k > 1) bs_extract_q<false>(st, (uint32_t) (BS_H - ((k - 1) >> 1)));
                }
            }
            // walk: the lane's path crosses each anti-diagonal at most once; codes 0 X, 1 =, 2 I, 3 D.
            // Branch-free: every step runs in all lanes, `on` (0/1) gates its effects.
            const int sbase = BS_K * c;
            uint64_t bw = 0;
            uint32_t e2 = 0;
#pragma unroll
            for (int k = 0; k < BS_K; ++k) {
                const uint32_t on = (running && a + b == sbase + k) ? 1u : 0u;
                const uint32_t dd = (uint32_t) (b - a + 64);
                const bool up = dd >= 64u;
                const uint32_t nbit = __builtin_amdgcn_ubfe(up ? N[k].hi : N[k].lo, dd >> 1, 1u);   // v_bfe takes the offset mod 32
                const uint32_t gbit = __builtin_amdgcn_ubfe(up ? G[k].hi : G[k].lo, dd >> 1, 1u);
                const uint32_t code = (nbit << 1) | gbit;
                bw |= (uint64_t) (on ? code : 0u) << e2;
                e2 += on << 1;
                score += (int) BS_LOP3(on, nbit, gbit, TA & ~(~TB & TC));          // every column but '='
                a += (int) BS_LOP3(on, nbit, gbit, TA & ~(TB & TC));               // every column but 'D'
                b += (int) BS_LOP3(on, nbit, gbit, TA & ~(TB & ~TC));              // every column but 'I'
                running = running && a < amax && b < bmax && a + b < smax;
            }
            // append the block's codes (at most 32) to the lane's code stream
            sb |= bw << fill;
            cnt += (int) (e2 >> 1);
            if (fill + (int) e2 >= 64) {
                pend = sb; has_pend = true;
                sb = (bw >> 1) >> (63 - fill);
                fill -= 64;
            }
            fill += (int) e2;
        }
        if (act) {
            i += a;
            j += b;
            if (a + b == 0) score = -1;                           // cannot happen (every walk moves); never spin
        }
    }
    if (lane == 0 && tiles) atomicAdd(&counters->gact_tiles, (unsigned long long) tiles);
}

// codes -> CIGAR bytes ('=' 'X' 'I' 'D', one per alignment column): one thread per 16 columns (one 16-byte store
// when the row is 16-byte aligned, four 4-byte stores otherwise)
__device__ __forceinline__ uint32_t bs_ops4(uint32_t c8, int o, int nc) {
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t op = o + e < nc ? __builtin_amdgcn_ubfe(0x44493d58u, ((c8 >> (2 * e)) & 3u) * 8u, 8u) : (uint32_t) 'I';
        w |= op << (8 * e);
    }
    return w;
}
__global__ __launch_bounds__(256) void bs_expand_kernel(const uint64_t *__restrict__ codes, uint64_t cw,
                                                        const int32_t *__restrict__ n_codes,
                                                        const int32_t *__restrict__ n_ops,
                                                        const uint32_t *__restrict__ rflags,
                                                        const int32_t *__restrict__ meta_r, uint64_t n_reads,
                                                        uint32_t blocks_per_read, uint8_t *__restrict__ store,
                                                        uint64_t store_stride) {
    const uint64_t r = blockIdx.x / blocks_per_read;
    if (r >= n_reads) return;
    if (meta_r[r] == 0 || (rflags && rflags[r])) return;         // fenced, or written by the byte kernel
    const int no = n_ops[r], nc = n_codes[r];
    const int o = (int) ((blockIdx.x % blocks_per_read) * 256 + threadIdx.x) * 16;
    if (o >= no) return;
    const uint32_t c32 = (uint32_t) (codes[r * cw + (uint64_t) (o >> 5)] >> ((o & 31) * 2));
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = bs_ops4((c32 >> (8 * q)) & 0xffu, o + 4 * q, nc);
    uint8_t *out = store + r * store_stride + o;
    if (o + 16 <= no && (((uintptr_t) out) & 15u) == 0) {
        *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oq = o + 4 * q;
            if (oq + 4 <= no) *reinterpret_cast<uint32_t *>(out + 4 * q) = w[q];
            else for (int e = 0; oq + e < no; ++e) out[4 * q + e] = (uint8_t) (w[q] >> (8 * e));
        }
    }
}

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

uint64_t lrm_bs_planar_words(uint64_t len) { return 2 * ((len + 63) / 64) + 2 * (uint64_t) BS_PADW; }

bool lrm_bs_wanted(lrm_gact_params gp, uint64_t n, int impl) {
    return gp.W <= 128 && (impl == 4 || (impl == 0 && n >= LRM_BS_MIN_READS));
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
    (void) groups;
    uint32_t h = 0;
    const bool ok = hipMalloc(&d, words * 8 + 16) == hipSuccess && hipMalloc(&flag, 16) == hipSuccess &&
                    lrm_bs_pack_text(idx->view.content, len, d, flag, nullptr) == 0 &&
                    hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (flag) (void) hipFree(flag);
    if (!ok) {                                   // nothing leaks on a failed allocation / pack / copy
        if (d) (void) hipFree(d);
        lrm_set_error("planar text for the bit-sliced extension: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
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
    uint32_t wv = (gpr + BS_PACK_G - 1) / BS_PACK_G;   // 16 groups (1 KiB of read) per wavefront
    if (wv == 0) wv = 1;
    const uint64_t waves = n * wv;
    if ((waves + 3) / 4 > 0x7fffffffull) { lrm_set_error("planar pack grid too large: split the batch"); return -1; }
    HIPCHK(hipMemsetAsync(d_flags, 0, n * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(bs_pack_reads_kernel, dim3((uint32_t) ((waves + 3) / 4)), dim3(256), 0, stream, d_reads, stride,
                       d_lens, d_qpl, wpr, d_flags, n, gpr, wv);
    return 0;
}

uint64_t lrm_bs_code_words(uint32_t max_len) { return (2ull * max_len + 31) / 32 + 2; }
uint64_t lrm_bs_ckpt_words(uint64_t n) {                     // T - O <= 512: at most 1024/K blocks per wavefront
    uint64_t waves = (n + 63) / 64;
    if (waves > LRM_BS_MAX_WAVES) waves = LRM_BS_MAX_WAVES;
    return waves * (uint64_t) (1024 / BS_K) * 512ull;
}

int lrm_bs_launch(const LrmBsArgs *bs, const uint32_t *d_lens, const lrm_seq_meta *d_meta, const int32_t *d_meta_r,
                  const uint32_t *d_tlens, uint64_t n, int T, int O, int W, uint8_t *d_store, uint64_t store_stride,
                  int32_t *d_n_ops, int32_t *d_score, LrmDevCounters *counters, uint32_t max_waves, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    uint64_t blocks = (n + 63) / 64;
    if (blocks > LRM_BS_MAX_WAVES) blocks = LRM_BS_MAX_WAVES;              // resident wavefronts; lanes refill from the queue
    if (max_waves >= 1 && max_waves < blocks) blocks = max_waves;          // (tests: a small grid forces refills)
    HIPCHK(hipMemsetAsync(&counters->reserved[0], 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(gact_bs_kernel, dim3((uint32_t) blocks), dim3(64), 0, stream, bs->qpl, bs->wpr, d_lens, d_meta,
                       d_meta_r, bs->cpl + BS_PADW, d_tlens, bs->flags, n, T, O, W, bs->ckpt, bs->codes, bs->cw,
                       bs->ncodes, d_n_ops, d_score, counters);
    const uint32_t bpr = (uint32_t) ((bs->cw * 32 + 4095) / 4096);          // 256 threads x 16 columns per block
    if (n * bpr > 0x7fffffffull) { lrm_set_error("expand grid too large: split the batch"); return -1; }
    hipLaunchKernelGGL(bs_expand_kernel, dim3((uint32_t) (n * bpr)), dim3(256), 0, stream, bs->codes, bs->cw, bs->ncodes,
                       d_n_ops, bs->flags, d_meta_r, n, bpr, d_store, store_stride);
    return 0;
}
