// gact_kernels.hip -- gfx950 kernels for PART 2 of the accaln hot path
// (reference: alnmain.c:408-451 -- seq_lookup :151-176, _rev_comp_in_place :27-60,
//  cigar_align mutils.c:94-105 -> simple_gact [gact submodule, source absent]).
//
//   locus_resolve  one lane per read: seq_lookup with the reference's u64 arithmetic
//   revcomp        reverse-complement reads that resolved to the reverse strand, in place
//   gact3          TWO READS PER WAVEFRONT (packed 16-bit scores, traceback planes in registers): small batches
//   gact_wide      ONE READ PER WAVEFRONT (32-bit scores, traceback in LDS): bands wider than 128 diagonals, and the
//                  general fallback (reads with bytes other than ACGT, T - O > 256)
//                  Tiles of a read are walked in sequence (tile i+1 starts where tile i's traceback stopped),
//                  docs/GACT_SPEC.md.  Large batches run on the bit-sliced lane-per-read kernel (gact_bs_kernels.hip).
//
// GACT tile on a 64-lane wavefront: the band (<=128 diagonals) is laid across the lanes and
// the wavefront sweeps anti-diagonals s = a+b from the far corner down to the anchor.  On an
// even s lane L owns diagonal d = 2L-64, on an odd s diagonal d = 2L-63, so every lane computes
// one lattice point per step (no idle parity) and the three neighbours are: own lane two steps
// ago (DIAG), own lane / lane-1 (INS) and own lane / lane+1 (DEL) one step ago -- a single DPP
// wave shift per step, no LDS traffic for scores.  Traceback pointers (2 bit/point) are packed
// 16 steps per dword and parked in LDS; the traceback walk reads them back.  Integer max/add
// recurrences: nothing here is a contraction, MFMA does not apply.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "lrm_internal.h"

#define GACT_NEG (-(1 << 28))
#define DPP_WAVE_SHL1 0x130  // lane L <- lane L+1
#define DPP_WAVE_SHR1 0x138  // lane L <- lane L-1

// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void locus_resolve_kernel(LrmIndexView ix, const lrm_entry *__restrict__ best,
                                                            const uint32_t *__restrict__ lens, uint64_t n,
                                                            lrm_seq_meta *__restrict__ meta,
                                                            int32_t *__restrict__ meta_r) {
    uint64_t read = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (read >= n) return;
    const uint64_t loc = best[read].key;                       // alnmain.c:427
    const uint32_t qlen = lens[read];
    lrm_seq_meta m;
    m.loc = 0; m.off = 0; m.seq_id = -1; m.strand = 0;
    int mr = 0;
    for (int i = 0; i < ix.mta_len; ++i) {                     // alnmain.c:155-174
        uint64_t sl = ix.mta[i].seq_len;
        uint64_t start = ix.mta[i].offset;
        uint64_t end = start + sl * 2;
        if (loc >= start && loc + qlen <= start + sl) {
            m.strand = 0; m.seq_id = i; m.loc = loc; m.off = loc - start;
            mr = 1;
            break;
        } else if (loc >= start + sl && loc + qlen <= end) {
            m.strand = 1; m.seq_id = i; m.off = end - loc - qlen; m.loc = m.off + start;
            mr = 1;
            break;
        }
    }
    // Fences: the reference consumes an uninitialised struct when the lookup fails,
    // and a wrapped u64 locus can pass the test while pointing outside the text.
    if (mr && (qlen == 0 || m.loc >= ix.con_len || (uint64_t) qlen > ix.con_len - m.loc)) mr = 0;
    if (!mr) { m.loc = 0; m.off = 0; m.seq_id = -1; m.strand = 0; }
    meta[read] = m;
    meta_r[read] = mr;
}

// alnmain.c:31-52: A/a -> T, C/c -> G, G/g -> C, T/t -> A, anything else -> N.  Branch-free on purpose: a `switch`
// compiles to a cascade of divergent branches per byte (the first revcomp kernel spent 1.5 ms per Gbp in them).
__device__ __forceinline__ char comp_base(char c) {
    const uint32_t u = (uint32_t) (uint8_t) c & 0xDFu;            // fold case
    uint32_t r = 'N';
    r = u == 'A' ? 'T' : r;
    r = u == 'C' ? 'G' : r;
    r = u == 'G' ? 'C' : r;
    r = u == 'T' ? 'A' : r;
    return (char) r;
}

__device__ __forceinline__ uint32_t bytes_equal(uint32_t x, uint32_t c4) {     // 0xFF in every byte of x equal to c4's
    const uint32_t z = x ^ c4;
    const uint32_t t = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);  // 0x80 where the byte of z is zero
    return (t >> 7) * 0xFFu;
}

__device__ __forceinline__ uint32_t revcomp4(uint32_t w) {       // 4 bases: reversed and complemented
    const uint32_t x = w & 0xDFDFDFDFu;
    const uint32_t a = bytes_equal(x, 0x41414141u), c = bytes_equal(x, 0x43434343u);
    const uint32_t g = bytes_equal(x, 0x47474747u), t = bytes_equal(x, 0x54545454u);
    const uint32_t o = (a & 0x54545454u) | (c & 0x47474747u) | (g & 0x43434343u) | (t & 0x41414141u) |
                       (~(a | c | g | t) & 0x4E4E4E4Eu);
    return __builtin_bswap32(o);
}

// In place, with ALIGNED 16-byte accesses only (rows start at any byte: stride = max_read_len + 1).  A workgroup owns
// `seg` bases of the front half of a read and their mirror bases; it copies both spans into LDS with aligned
// 16-byte loads (the first and last chunk reach a few bytes outside the span: loaded, never used), and after the
// barrier every thread builds whole aligned 16-byte chunks of each span from the other one (five LDS dwords,
// v_alignbyte, byte swap + complement).  Chunks that straddle a span's ends are written bytewise.  Measured [r2]:
// one unaligned dword per thread 1.5 ms per Gbp, one unaligned 16-byte access per thread 2.2 ms, this version
// see profiles/r2.
#define RC_SEG 4096
__device__ __forceinline__ void revcomp_span(const uint8_t *__restrict__ src, uint32_t src_off, char *gdst,
                                             uint32_t dst_off, uint32_t cnt, uint32_t tid) {
    // destination bytes i in [0, cnt) live at gdst_aligned + dst_off + i; byte i = comp(src[src_off + cnt-1-i])
    const uint32_t nchunk = (dst_off + cnt + 15) / 16;
    const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
    for (uint32_t c = tid; c < nchunk; c += 256) {
        const int32_t i0 = (int32_t) (16 * c) - (int32_t) dst_off;           // first destination byte of the chunk
        if (i0 >= 0 && (uint32_t) i0 + 16 <= cnt) {
            const uint32_t lo = src_off + cnt - 16 - (uint32_t) i0;          // source bytes [lo, lo + 16)
            const uint32_t q = lo >> 2, sh = lo & 3;
            uint32_t d[5], o[4];
#pragma unroll
            for (int e = 0; e < 5; ++e) d[e] = src32[q + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[3 - e] = revcomp4(__builtin_amdgcn_alignbyte(d[e + 1], d[e], sh));
            *reinterpret_cast<uint4 *>(gdst + 16 * c) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            for (int k = 0; k < 16; ++k) {
                const int32_t i = i0 + k;
                if (i >= 0 && (uint32_t) i < cnt) gdst[16 * c + k] = comp_base((char) src[src_off + cnt - 1 - (uint32_t) i]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void revcomp_kernel(char *__restrict__ reads, uint64_t stride,
                                                      const uint32_t *__restrict__ lens,
                                                      const lrm_seq_meta *__restrict__ meta,
                                                      const int32_t *__restrict__ meta_r, uint64_t n,
                                                      uint32_t chunks_per_read, uint32_t seg) {
    __shared__ __attribute__((aligned(16))) uint8_t s_a[RC_SEG + 48], s_b[RC_SEG + 48];
    const uint64_t read = blockIdx.x / chunks_per_read;
    const uint32_t chunk = blockIdx.x % chunks_per_read;
    if (read >= n) return;
    if (!meta_r[read] || meta[read].strand != 1) return;       // alnmain.c:433
    const uint32_t len = lens[read], half = len / 2;
    char *r = reads + read * stride;
    const uint32_t tid = threadIdx.x;
    if (chunk == 0 && tid == 0 && (len & 1)) r[half] = comp_base(r[half]);
    const uint32_t s = chunk * seg;
    if (s >= half) return;
    const uint32_t e = s + seg < half ? s + seg : half, cnt = e - s;          // front [s, e), mirror [len-e, len-s)
    char *ga = r + s, *gb = r + (len - e);
    const uint32_t off_a = (uint32_t) ((uintptr_t) ga & 15), off_b = (uint32_t) ((uintptr_t) gb & 15);
    ga -= off_a;
    gb -= off_b;
    for (uint32_t c = tid; 16 * c < off_a + cnt; c += 256)
        *reinterpret_cast<uint4 *>(s_a + 16 * c) = *reinterpret_cast<const uint4 *>(ga + 16 * c);
    for (uint32_t c = tid; 16 * c < off_b + cnt; c += 256)
        *reinterpret_cast<uint4 *>(s_b + 16 * c) = *reinterpret_cast<const uint4 *>(gb + 16 * c);
    __syncthreads();
    revcomp_span(s_b, off_b, ga, off_a, cnt, tid);
    revcomp_span(s_a, off_a, gb, off_b, cnt, tid);
}

// ----------------------------------------------------------------------------------------
// GACT
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_from_lower(int v, int fill) {   // lane L <- lane L-1, lane 0 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_upper(int v, int fill) {   // lane L <- lane L+1, lane 63 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHL1, 0xf, 0xf, false);
}

// one DP step for this lane's lattice point.  TRACK: also append the 2-bit pointer to acc.
template <bool TRACK, bool FULLBAND>
__device__ __forceinline__ int gact_cell(int r_diag, int r_ins, int r_del, uint32_t qc, uint32_t dc,
                                         bool is_exit, bool inband, uint32_t &acc) {
    const bool eq = qc == dc;
    int cd = r_diag + (eq ? 1 : -1);
    int m1 = (r_ins >= r_del ? r_ins : r_del) - 1;
    int best = cd >= m1 ? cd : m1;
    if (TRACK) {
        // 2-bit pointer: 0 DIAG/match, 3 DIAG/mismatch, 1 INS, 2 DEL -- the walk needs no sequence reads
        uint32_t p1 = r_ins >= r_del ? 1u : 2u;        // INS before DEL on ties
        uint32_t pd = eq ? 0u : 3u;
        uint32_t p = cd >= m1 ? pd : p1;               // DIAG wins ties
        acc = (acc << 2) | p;
    }
    best = is_exit ? 0 : best;
    if (!FULLBAND) best = inband ? best : GACT_NEG;
    return best;
}

// ----------------------------------------------------------------------------------------
// GACT v2: TWO reads per wavefront, packed 16-bit.
//
// Every lane register holds the same lattice point of two independent reads (read A in bits 0-15,
// read B in bits 16-31) and the recurrences run on v_pk_* instructions, so one instruction stream
// advances two tiles.  To make that work without per-half compares/selects:
//   * scores are kept as V = 2R + s + BIAS (s = a+b of the point).  All three candidates of a point
//     share s, so every max / tie decision is unchanged, and the score terms become
//     DIAG: V - 4*[mismatch]   GAP: V - 3   EXIT: s + BIAS   -- no compare needed: the mismatch
//     flag is pk_min_u16(q ^ d, 1);
//   * valid V are >= 1 and "no such neighbour" is 0, so the DPP wave shift runs with bound_ctrl
//     (out-of-wave lanes read 0) and needs no fill register;
//   * traceback decisions are two bit-planes per read, N = "not DIAG" and G = "DEL rather than INS"
//     (sign bits of two packed subtractions), 16 steps per 16-bit half; match/mismatch is
//     recovered during the walk from the staged sequences, a whole DIAG run per instruction.
// ----------------------------------------------------------------------------------------
#define G2_BIAS 2560
#define G2_PAD 40            // guard positions on both sides of the staged sequences (>= 33)

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    v2s x = __builtin_bit_cast(v2s, a), y = __builtin_bit_cast(v2s, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    v2u x = __builtin_bit_cast(v2u, a), y = __builtin_bit_cast(v2u, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
}
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) {
    v2u x = __builtin_bit_cast(v2u, a), y = __builtin_bit_cast(v2u, b);
    return __builtin_bit_cast(uint32_t, (v2u) (x + y));
}
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
    v2u x = __builtin_bit_cast(v2u, a), y = __builtin_bit_cast(v2u, b);
    return __builtin_bit_cast(uint32_t, (v2u) (x - y));
}
__device__ __forceinline__ uint32_t pk_mul16(uint32_t a, uint32_t b) {
    v2u x = __builtin_bit_cast(v2u, a), y = __builtin_bit_cast(v2u, b);
    return __builtin_bit_cast(uint32_t, (v2u) (x * y));
}
__device__ __forceinline__ uint32_t pk_lshr16(uint32_t a, int n) {
    v2u x = __builtin_bit_cast(v2u, a);
    return __builtin_bit_cast(uint32_t, (v2u) (x >> (unsigned short) n));
}
__device__ __forceinline__ uint32_t pk_ashr16(uint32_t a, int n) {
    v2s x = __builtin_bit_cast(v2s, a);
    return __builtin_bit_cast(uint32_t, (v2s) (x >> (short) n));
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t) lo & 0xffffu) | ((uint32_t) hi << 16); }

// The three packed instructions LLVM will not form from C (it rewrites min(x,1)*4 into two
// compares + selects + a byte permute): written out.  All inputs/outputs are plain VGPRs and the
// results feed only non-DPP VALU instructions, so no manual wait states are needed.
__device__ __forceinline__ uint32_t pk_mismatch(uint32_t x) {         // per half: x != 0 ? 1 : 0
    uint32_t z;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(z) : "v"(x));
    return z;
}
__device__ __forceinline__ uint32_t pk_mad_m4(uint32_t z, uint32_t v) { // per half: v - 4*z
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, -4, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(z), "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t pk_shift_in(uint32_t acc, uint32_t bit) {  // per half: acc*2 + bit
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(acc), "v"(bit));
    return r;
}

template <bool TRACK, bool EXIT, bool FULLBAND>
__device__ __forceinline__ uint32_t gact2_cell(uint32_t v_diag, uint32_t v_ins, uint32_t v_del, uint32_t qc,
                                               uint32_t dc, uint32_t e_pk, uint32_t s_pk, uint32_t sb_pk,
                                               bool inband, uint32_t &accN, uint32_t &accG) {
    const uint32_t cd = pk_mad_m4(pk_mismatch(qc ^ dc), v_diag);        // DIAG: -4 on a mismatch
    const uint32_t m1 = pk_sub16(pk_max_i16(v_ins, v_del), 0x00030003u); // best gap: -3
    uint32_t best = pk_max_i16(cd, m1);
    if (TRACK) {
        accN = pk_shift_in(accN, pk_lshr16(pk_sub16(cd, m1), 15));       // cd < m1: not DIAG
        accG = pk_shift_in(accG, pk_lshr16(pk_sub16(v_ins, v_del), 15)); // ins < del: DEL
    }
    if (EXIT) {          // only anti-diagonals that touch the tile's far boundary pay for this
        const uint32_t interior = pk_ashr16(pk_sub16(s_pk, e_pk), 15);  // 0xFFFF where s < e
        best = (interior & best) | (~interior & sb_pk);                 // exit / outside: V = s + BIAS
    }
    if (!FULLBAND) best = inband ? best : 0u;
    return best;
}

// ----------------------------------------------------------------------------------------
// GACT v3 = v2 with the traceback bit-planes kept in REGISTERS.
// With the planes in LDS a wavefront needs 16 KiB and only 2 wavefronts fit a SIMD, which leaves
// the dependent chain of the sweep exposed.  The walk's anti-diagonal index only ever grows, and a
// plane word is one VGPR per 16-step block, so: the sweep moves each finished block into a
// statically named register (switch on the block number, one v_mov per plane), and the
// wave-uniform walk fetches the word of the lane it needs with v_readlane from the register the
// same switch names.  LDS then holds only the staged sequences (3.1 KiB per wavefront) and
// occupancy is set by registers (~5 wavefronts = 10 reads per SIMD).  The walk looks two blocks
// ahead, so a DIAG run only ends at an indel or after >= 16 moves.
// ----------------------------------------------------------------------------------------
// (block, next block) pairs; the planes are plain local variables tbnK / tbgK (K = 0..32) of the kernel --
// not an array or struct: LLVM sinks the stores of a switch over a[K] / s.mK into one dynamically
// addressed store, which sends the whole object to scratch; distinct allocas are left alone.
#define G3_CASES(X) X(0,1) X(1,2) X(2,3) X(3,4) X(4,5) X(5,6) X(6,7) X(7,8) X(8,9) X(9,10) X(10,11) X(11,12) X(12,13) X(13,14) X(14,15) X(15,16) X(16,17) X(17,18) X(18,19) X(19,20) X(20,21) X(21,22) X(22,23) X(23,24) X(24,25) X(25,26) X(26,27) X(27,28) X(28,29) X(29,30) X(30,31) X(31,32)

template <bool FULLBAND, int NB>
__global__ __launch_bounds__(256) void gact3_kernel(const char *__restrict__ reads, uint64_t stride,
                                                    const uint32_t *__restrict__ lens,
                                                    const lrm_seq_meta *__restrict__ meta,
                                                    const int32_t *__restrict__ meta_r,
                                                    const char *__restrict__ content,
                                                    const uint32_t *__restrict__ tlens, uint64_t n_reads,
                                                    int T, int O, int W, uint8_t *__restrict__ store,
                                                    uint64_t store_stride, int32_t *__restrict__ n_ops_out,
                                                    int32_t *__restrict__ score_out, LrmDevCounters *counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint64_t r0 = ((uint64_t) blockIdx.x * 4 + (uint64_t) wave) * 2;
    if (r0 >= n_reads) return;

    int n[2], m[2], i[2] = {0, 0}, j[2] = {0, 0}, nops[2] = {0, 0}, score[2] = {0, 0};
    const uint8_t *q[2], *d[2];
    uint8_t *ops_out[2];
    bool ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t r = r0 + (uint64_t) h;
        const uint64_t rs = r < n_reads ? r : r0;
        ok[h] = r < n_reads && meta_r[rs] != 0;
        n[h] = ok[h] ? __builtin_amdgcn_readfirstlane((int) lens[rs]) : 0;
        m[h] = ok[h] ? (tlens ? __builtin_amdgcn_readfirstlane((int) tlens[rs]) : n[h]) : 0;
        q[h] = reinterpret_cast<const uint8_t *>(reads) + rs * stride;
        d[h] = reinterpret_cast<const uint8_t *>(content) + (ok[h] ? meta[rs].loc : 0);
        ops_out[h] = store + rs * store_stride;
        if (r < n_reads && !ok[h] && lane == 0) { n_ops_out[r] = 0; score_out[r] = -1; }   // fenced
    }
    if (!ok[0] && !ok[1]) return;

    const uint32_t seq_words = (uint32_t) T + 2 * G2_PAD;
    uint32_t *qbuf = reinterpret_cast<uint32_t *>(smem) + (size_t) wave * 2 * seq_words + G2_PAD;
    uint32_t *dbuf = qbuf + seq_words;

    const int hw = W / 2;
    const int dE = 2 * lane - 64, dO = 2 * lane - 63;
    const bool inE = dE >= -hw && dE < hw, inO = dO >= -hw && dO < hw;
    const int cap = T - O, lim2 = 2 * cap;
    const int nblk = ((lim2 - 1) >> 4) + 1;
    const int s_hi = nblk * 16 - 1;
    unsigned tiles = 0;
#define X(K, K1) uint32_t tbn##K = 0, tbg##K = 0;
    G3_CASES(X)
#undef X
    const uint32_t tbn32 = 0, tbg32 = 0;
#define TB_PUT(k_, an_, ag_) do { switch (k_) {                                   \
        G3_CASES(TB_PUT_CASE)                                                         \
        default: break; } } while (0)
#define TB_PUT_CASE(K, K1) case K: if (K < NB) { tbn##K = accN; tbg##K = accG; } break;
#define TB_GET_CASE(K, K1) case K: if (K < NB) {                                  \
            n0 = (uint32_t) __builtin_amdgcn_readlane((int) tbn##K, ln_);             \
            g0 = (uint32_t) __builtin_amdgcn_readlane((int) tbg##K, ln_);             \
            n1 = (K1 < NB) ? (uint32_t) __builtin_amdgcn_readlane((int) tbn##K1, ln_) : 0u; \
            g1 = (K1 < NB) ? (uint32_t) __builtin_amdgcn_readlane((int) tbg##K1, ln_) : 0u; \
        } break;

    while ((i[0] < n[0] && j[0] < m[0]) || (i[1] < n[1] && j[1] < m[1])) {
        int tq[2], tt[2];
        bool act[2], last[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            act[h] = i[h] < n[h] && j[h] < m[h];
            tq[h] = act[h] ? min(T, n[h] - i[h]) : 0;
            tt[h] = act[h] ? min(T, m[h] - j[h]) : 0;
            last[h] = i[h] + tq[h] == n[h];
            tiles += act[h] ? 1u : 0u;
        }
        for (int x = lane; x < T; x += 64) {
            uint32_t qa = x < tq[0] ? q[0][i[0] + x] : 0u, qb = x < tq[1] ? q[1][i[1] + x] : 0u;
            uint32_t da = x < tt[0] ? d[0][j[0] + x] : 0u, db = x < tt[1] ? d[1][j[1] + x] : 0u;
            qbuf[x] = qa | (qb << 16);
            dbuf[x] = da | (db << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint32_t eE = pack16(min(2 * tq[0] + dE, 2 * tt[0] - dE), min(2 * tq[1] + dE, 2 * tt[1] - dE));
        const uint32_t eO = pack16(min(2 * tq[0] + dO, 2 * tt[0] - dO), min(2 * tq[1] + dO, 2 * tt[1] - dO));
        uint32_t r1 = 0, r2 = 0, accN = 0, accG = 0;
        int s = max(tq[0] + tt[0], tq[1] + tt[1]);
        const uint32_t *qp, *dp;
        uint32_t qc = 0, dc = 0, qn, dn;
        if ((s & 1) == 0) {
            qp = qbuf + (s / 2 + 32 - lane);
            dp = dbuf + (s / 2 - 32 + lane) + 1;
            qc = *qp;
        } else {
            qp = qbuf + ((s - 1) / 2 + 32 - lane) + 1;
            dp = dbuf + ((s - 1) / 2 - 31 + lane);
            dc = *dp;
        }
        qn = qp[-1];
        dn = dp[-1];

        // Below s_free every lane of every ACTIVE read is strictly inside its tile, so the exit test
        // (3 VALU + 3 SALU per step) is only paid on the first ~64 anti-diagonals of a tile.
        int s_free = 0x7fffffff;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (act[h]) s_free = min(s_free, 2 * min(tq[h], tt[h]) - 64);

#define G2_SPK(S) ((uint32_t) (S) * 0x00010001u)
#define G2_ODD(TRACK, EXIT, S) do {                                                              \
            qc = qn; qp -= 1; qn = qp[-1];                                                           \
            uint32_t del_ = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) r1, DPP_WAVE_SHL1, 0xf, 0xf, true); \
            uint32_t r0_ = gact2_cell<TRACK, EXIT, FULLBAND>(r2, r1, del_, qc, dc, eO, G2_SPK(S),    \
                                                             G2_SPK((S) + G2_BIAS), inO, accN, accG); \
            r2 = r1; r1 = r0_;                                                                       \
        } while (0)
#define G2_EVEN(TRACK, EXIT, S) do {                                                             \
            dc = dn; dp -= 1; dn = dp[-1];                                                           \
            uint32_t ins_ = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) r1, DPP_WAVE_SHR1, 0xf, 0xf, true); \
            uint32_t r0_ = gact2_cell<TRACK, EXIT, FULLBAND>(r2, ins_, r1, qc, dc, eE, G2_SPK(S),    \
                                                             G2_SPK((S) + G2_BIAS), inE, accN, accG); \
            r2 = r1; r1 = r0_;                                                                       \
        } while (0)
#define G2_FLUSH(S) do { if (((S) & 15) == 0) TB_PUT((S) >> 4, accN, accG); } while (0)
        // generic segment: anti-diagonals s .. LIM+1 (any parity at either end)
#define G2_SEG(LIM, TRACK, EXIT) do {                                                            \
            const int lim_ = (LIM);                                                                  \
            if (s > lim_ && (s & 1) == 0) { G2_EVEN(TRACK, EXIT, s); if (TRACK) G2_FLUSH(s); s--; }  \
            for (; s - 1 > lim_; s -= 2) {                                                           \
                G2_ODD(TRACK, EXIT, s); G2_EVEN(TRACK, EXIT, s - 1); if (TRACK) G2_FLUSH(s - 1);     \
            }                                                                                        \
            if (s > lim_) { G2_ODD(TRACK, EXIT, s); s--; }                                           \
        } while (0)

        // phase A (no walk can reach these anti-diagonals: scores only), with then without exit test
        G2_SEG(max(s_hi, s_free - 1), false, true);
        if (s > s_hi && (s & 1) == 0) { G2_EVEN(false, false, s); s--; }      // the unrolled loops start on an odd s
        for (; s - 7 > s_hi; s -= 8) {
            G2_ODD(false, false, s);     G2_EVEN(false, false, s - 1); G2_ODD(false, false, s - 2); G2_EVEN(false, false, s - 3);
            G2_ODD(false, false, s - 4); G2_EVEN(false, false, s - 5); G2_ODD(false, false, s - 6); G2_EVEN(false, false, s - 7);
        }
        G2_SEG(s_hi, false, false);
        // phase B (scores + traceback bit-planes): exit test only for small tiles; then down to a
        // 16-step block boundary; then whole blocks -- 16 steps, one flush, no per-step bookkeeping
        G2_SEG(max(-1, s_free - 1), true, true);
        if (s >= 0 && (s & 1) == 0) { G2_EVEN(true, false, s); G2_FLUSH(s); s--; }
        for (; s >= 1 && (s & 15) != 15; s -= 2) {
            G2_ODD(true, false, s); G2_EVEN(true, false, s - 1); G2_FLUSH(s - 1);
        }
        for (; s >= 15; s -= 16) {
            G2_ODD(true, false, s);      G2_EVEN(true, false, s - 1);  G2_ODD(true, false, s - 2);  G2_EVEN(true, false, s - 3);
            G2_ODD(true, false, s - 4);  G2_EVEN(true, false, s - 5);  G2_ODD(true, false, s - 6);  G2_EVEN(true, false, s - 7);
            G2_ODD(true, false, s - 8);  G2_EVEN(true, false, s - 9);  G2_ODD(true, false, s - 10); G2_EVEN(true, false, s - 11);
            G2_ODD(true, false, s - 12); G2_EVEN(true, false, s - 13); G2_ODD(true, false, s - 14); G2_EVEN(true, false, s - 15);
            TB_PUT(s >> 4, accN, accG);
        }
#undef G2_SEG
#undef G2_FLUSH
#undef G2_ODD
#undef G2_EVEN
#undef G2_SPK

        // traceback walks (wave-uniform), a 32-step window of the lane's bit-planes per iteration
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!act[h]) continue;
            const int sh = 16 * h;
            int a = 0, b = 0, cnt = 0, sc = 0;
            uint8_t *out = ops_out[h] + nops[h];
            while (a < tq[h] && b < tt[h] && (last[h] ? (a + b < lim2) : (a < cap && b < cap))) {
                const int sw = a + b, dd = b - a, e0 = sw & 15;
                uint32_t n0 = 0, g0 = 0, n1 = 0, g1 = 0;
                {
                    const int ln_ = (dd + 64) >> 1;
                    switch (sw >> 4) {
                        G3_CASES(TB_GET_CASE)
                        default: break;
                    }
                }
                const uint32_t wN = ((n0 >> sh) & 0xffffu) | (((n1 >> sh) & 0xffffu) << 16);   // steps e0 .. 31
                const uint32_t wG = ((g0 >> sh) & 0xffffu) | (((g1 >> sh) & 0xffffu) << 16);
                const uint32_t par = (e0 & 1) ? 0xAAAAAAAAu : 0x55555555u;
                const uint32_t gm = wN & par & (0xffffffffu << e0);
                const int egap = gm ? __builtin_ctz(gm) : 32 + (e0 & 1);
                const int r = (egap - e0) >> 1;
                int rmax = min(tq[h] - a, tt[h] - b);
                rmax = last[h] ? min(rmax, (lim2 - sw + 1) >> 1) : min(rmax, min(cap - a, cap - b));
                const int rr = min(r, rmax);
                bool neq = false;
                if (lane < rr) {
                    neq = (((qbuf[a + lane] ^ dbuf[b + lane]) >> sh) & 0xffffu) != 0u;
                    out[cnt + lane] = neq ? 'X' : '=';
                }
                const int mism = __builtin_popcountll(__ballot(neq));
                const int a2 = a + rr, b2 = b + rr;
                const bool more = a2 < tq[h] && b2 < tt[h] && (last[h] ? (a2 + b2 < lim2) : (a2 < cap && b2 < cap));
                const bool do_gap = gm != 0u && rr == r && more;
                const bool is_del = ((wG >> (egap & 31)) & 1u) != 0u;
                if (do_gap && lane == 0) out[cnt + rr] = is_del ? 'D' : 'I';
                sc += mism + (do_gap ? 1 : 0);
                cnt += rr + (do_gap ? 1 : 0);
                a = a2 + ((do_gap && !is_del) ? 1 : 0);
                b = b2 + ((do_gap && is_del) ? 1 : 0);
            }
            nops[h] += cnt;
            score[h] += sc;
            i[h] += a;
            j[h] += b;
            if (a + b == 0) { n[h] = 0; score[h] = -1; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (!ok[h]) continue;
        const uint64_t r = r0 + (uint64_t) h;
        if (score[h] >= 0 && i[h] < n[h]) {
            int rest = n[h] - i[h];
            for (int x = lane; x < rest; x += 64) ops_out[h][nops[h] + x] = 'I';
            nops[h] += rest;
            score[h] += rest;
        }
        if (lane == 0) {
            n_ops_out[r] = score[h] >= 0 ? nops[h] : 0;
            score_out[r] = score[h];
        }
    }
    if (lane == 0) atomicAdd(&counters->gact_tiles, (unsigned long long) tiles);
}

// ----------------------------------------------------------------------------------------
// GACT, one read per wavefront, 32-bit scores: wide bands (128 < W <= 1024, e.g. the full-tile band W = T of the
// config-5 sweep; DPL = 2, 4, 8) and, with DPL = 1, the general fallback for what the fast kernels do not take:
// reads holding a byte other than ACGT behind the bit-sliced kernel, and tiles with T - O > 256 in small batches.
// The wavefront sweeps anti-diagonals from the tile's far corner to the anchor; every lane owns DPL diagonal pairs: diagonal index x = lane + 64*g
// (g < DPL), d = 2x - 64*DPL (+1 on odd anti-diagonals).  The neighbour of a group's edge lane is the
// opposite edge lane of the adjacent group (one v_readlane per group and step).  One wavefront per
// workgroup: the traceback of a wide band needs up to 128 KiB of LDS.  Simple form (32-bit scores,
// one read per wavefront); the packed two-read kernel covers the default band.
// ----------------------------------------------------------------------------------------
template <int DPL>
__global__ __launch_bounds__(64) void gact_wide_kernel(const char *__restrict__ reads, uint64_t stride,
                                                       const uint32_t *__restrict__ lens,
                                                       const lrm_seq_meta *__restrict__ meta,
                                                       const int32_t *__restrict__ meta_r,
                                                       const char *__restrict__ content,
                                                       const uint32_t *__restrict__ tlens, uint64_t n_reads,
                                                       int T, int O, int W, uint8_t *__restrict__ store,
                                                       uint64_t store_stride, int32_t *__restrict__ n_ops_out,
                                                       int32_t *__restrict__ score_out, LrmDevCounters *counters,
                                                       const uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NX = 64 * DPL;                   // diagonal indices per parity
    constexpr int HWX = NX;                        // d = 2x - HWX (+1)
    constexpr int PADW = NX / 2 + 40;              // guard bytes on both sides of the staged sequences
    const int lane = threadIdx.x & 63;
    const uint64_t read = blockIdx.x;
    if (read >= n_reads) return;
    if (flags && !flags[read]) return;             // second launch behind the bit-sliced kernel: flagged reads only
    if (!meta_r[read]) {
        if (lane == 0) { n_ops_out[read] = 0; score_out[read] = -1; }
        return;
    }
    const int cap = T - O, lim2 = 2 * cap;
    const int tb_words = ((lim2 - 1) >> 4) + 1;
    const int seq_bytes = (T + 2 * PADW + 15) & ~15;
    uint32_t *tb = reinterpret_cast<uint32_t *>(smem);
    uint8_t *qbuf = smem + (size_t) tb_words * NX * 4 + PADW;
    uint8_t *dbuf = qbuf + seq_bytes;
    uint8_t *opsbuf = smem + (size_t) tb_words * NX * 4 + 2 * (size_t) seq_bytes;

    const int n = __builtin_amdgcn_readfirstlane((int) lens[read]);
    const int m = tlens ? __builtin_amdgcn_readfirstlane((int) tlens[read]) : n;
    const uint8_t *q = reinterpret_cast<const uint8_t *>(reads) + read * stride;
    const uint8_t *d = reinterpret_cast<const uint8_t *>(content) + meta[read].loc;
    uint8_t *ops_out = store + read * store_stride;
    const int hw = W / 2;
    const int s_hi = tb_words * 16 - 1;

    int i = 0, j = 0, nops = 0, score = 0;
    unsigned tiles = 0;
    while (i < n && j < m) {
        const int tq = (n - i) < T ? (n - i) : T;
        const int tt = (m - j) < T ? (m - j) : T;
        const bool last = (i + tq == n);
        tiles++;
        for (int x = lane; x < tq; x += 64) qbuf[x] = q[i + x];
        for (int x = lane; x < tt; x += 64) dbuf[x] = d[j + x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        int r1[DPL], r2[DPL], eE[DPL], eO[DPL];
        bool inE[DPL], inO[DPL];
        uint32_t acc[DPL];
#pragma unroll
        for (int g = 0; g < DPL; ++g) {
            const int x = lane + 64 * g, dE = 2 * x - HWX, dO = dE + 1;
            r1[g] = GACT_NEG; r2[g] = GACT_NEG; acc[g] = 0;
            eE[g] = min(2 * tq + dE, 2 * tt - dE);
            eO[g] = min(2 * tq + dO, 2 * tt - dO);
            inE[g] = dE >= -hw && dE < hw;
            inO[g] = dO >= -hw && dO < hw;
        }
        for (int s = tq + tt; s >= 0; --s) {
            const bool odd = s & 1;
            const bool track = s <= s_hi;
            int nr[DPL];
#pragma unroll
            for (int g = 0; g < DPL; ++g) {
                const int x = lane + 64 * g;
                // even: a = s/2 + HWX/2 - x, b = s/2 - HWX/2 + x ; odd: a = (s-1)/2 + HWX/2 - x, b = (s-1)/2 - HWX/2 + 1 + x
                const int h = odd ? (s - 1) / 2 : s / 2;
                const int a = h + HWX / 2 - x, b = h - HWX / 2 + x + (odd ? 1 : 0);
                const uint32_t qc = qbuf[a], dc = dbuf[b];
                int ins, del;
                if (!odd) {      // INS neighbour = diagonal index x-1 one step ago, DEL = own
                    const int fill = g > 0 ? __builtin_amdgcn_readlane(r1[g > 0 ? g - 1 : 0], 63) : GACT_NEG;
                    ins = dpp_from_lower(r1[g], fill);
                    del = r1[g];
                } else {         // INS = own, DEL = diagonal index x+1 one step ago
                    const int fill = g + 1 < DPL ? __builtin_amdgcn_readlane(r1[g + 1 < DPL ? g + 1 : g], 0) : GACT_NEG;
                    ins = r1[g];
                    del = dpp_from_upper(r1[g], fill);
                }
                uint32_t a2 = acc[g];
                nr[g] = gact_cell<true, false>(r2[g], ins, del, qc, dc, s >= (odd ? eO[g] : eE[g]),
                                               odd ? inO[g] : inE[g], a2);
                if (track) acc[g] = a2;
            }
#pragma unroll
            for (int g = 0; g < DPL; ++g) {
                r2[g] = r1[g]; r1[g] = nr[g];
                if (track && (s & 15) == 0) tb[(s >> 4) * NX + lane + 64 * g] = acc[g];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        int a = 0, b = 0, cnt = 0;
        while (a < tq && b < tt && (last ? (a + b < lim2) : (a < cap && b < cap))) {
            const int sw = a + b, dd = b - a;
            const uint32_t word = __builtin_amdgcn_readfirstlane(tb[(sw >> 4) * NX + ((dd + HWX) >> 1)]);
            const uint32_t p = (word >> (2 * (sw & 15))) & 3u;
            uint8_t op;
            if (p == 0u || p == 3u) { op = p == 0u ? '=' : 'X'; score += p == 3u ? 1 : 0; a++; b++; }
            else if (p == 1u) { op = 'I'; score++; a++; }
            else { op = 'D'; score++; b++; }
            if (lane == 0) opsbuf[cnt] = op;
            cnt++;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int x = lane; x < cnt; x += 64) ops_out[nops + x] = opsbuf[x];
        nops += cnt;
        i += a;
        j += b;
        if (a + b == 0) { score = -1; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (score >= 0 && i < n) {
        int rest = n - i;
        for (int x = lane; x < rest; x += 64) ops_out[nops + x] = 'I';
        nops += rest;
        score += rest;
    }
    if (lane == 0) {
        n_ops_out[read] = score >= 0 ? nops : 0;
        score_out[read] = score;
        atomicAdd(&counters->gact_tiles, (unsigned long long) tiles);
    }
}

typedef void (*gact1_fn_t)(const char *, uint64_t, const uint32_t *, const lrm_seq_meta *, const int32_t *,
                           const char *, const uint32_t *, uint64_t, int, int, int, uint8_t *, uint64_t, int32_t *,
                           int32_t *, LrmDevCounters *, const uint32_t *);
typedef void (*gact2_fn_t)(const char *, uint64_t, const uint32_t *, const lrm_seq_meta *, const int32_t *,
                           const char *, const uint32_t *, uint64_t, int, int, int, uint8_t *, uint64_t, int32_t *,
                           int32_t *, LrmDevCounters *);

// one read per wavefront (gact_wide_kernel): DPL diagonal pairs per lane for the band W; flags != null: flagged reads only
static int launch_one_per_wave(lrm_gact_params gp, uint64_t n, hipStream_t stream, const char *reads, uint64_t stride,
                               const uint32_t *lens, const lrm_seq_meta *meta, const int32_t *meta_r, const char *content,
                               const uint32_t *tlens, uint8_t *store, uint64_t store_stride, int32_t *n_ops, int32_t *score,
                               LrmDevCounters *counters, const uint32_t *flags) {
    const int dpl = gp.W <= 128 ? 1 : gp.W <= 256 ? 2 : gp.W <= 512 ? 4 : 8;
    const int nx = 64 * dpl, padw = nx / 2 + 40;
    const int tbw = ((2 * (gp.T - gp.O) - 1) >> 4) + 1;
    const int seqb = (gp.T + 2 * padw + 15) & ~15;
    size_t shw = (size_t) tbw * nx * 4 + 2 * (size_t) seqb + (((size_t) 2 * (gp.T - gp.O) + 15) & ~(size_t) 15);
    if (shw > 160 * 1024) { lrm_set_error("GACT T=%d O=%d W=%d needs %zu B of LDS (> 160 KiB)", gp.T, gp.O, gp.W, shw); return -1; }
    gact1_fn_t fw = dpl == 1 ? gact_wide_kernel<1> : dpl == 2 ? gact_wide_kernel<2> : dpl == 4 ? gact_wide_kernel<4> : gact_wide_kernel<8>;
    if (shw > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fw),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) shw);
        if (e != hipSuccess) { lrm_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", shw, hipGetErrorString(e)); return -1; }
    }
    if (n > 0x7fffffffull) { lrm_set_error("gact grid too large: split the batch"); return -1; }
    hipLaunchKernelGGL(fw, dim3((uint32_t) n), dim3(64), shw, stream, reads, stride, lens, meta, meta_r, content,
                       tlens, n, gp.T, gp.O, gp.W, store, store_stride, n_ops, score, counters, flags);
    return 0;
}

static int gact_launch(lrm_gact_params gp, uint64_t n, hipStream_t stream, const char *reads, uint64_t stride,
                       const uint32_t *lens, const lrm_seq_meta *meta, const int32_t *meta_r, const char *content,
                       const uint32_t *tlens, uint8_t *store, uint64_t store_stride, int32_t *n_ops, int32_t *score,
                       LrmDevCounters *counters, const LrmBsArgs *bs, int impl, uint32_t bs_waves) {
    if (gp.W > 128)
        return launch_one_per_wave(gp, n, stream, reads, stride, lens, meta, meta_r, content, tlens, store, store_stride,
                                   n_ops, score, counters, nullptr);
    // impl (lrm_map_options.gact_impl): 0 = automatic, 1 = one read per wavefront, 3 = packed two reads per wavefront,
    // 4 = bit-sliced lane per read whenever it applies (W <= 128, pure ACGT text, 4-byte aligned CIGAR store;
    // otherwise as 0)
    const int nblk = ((2 * (gp.T - gp.O) - 1) >> 4) + 1;
    // Bit-sliced kernel: a wavefront carries 64 reads, so it needs a large batch to fill the chip
    // (below ~16 k reads the two-reads-per-wavefront kernel finishes first).
    const bool bs_ok = bs && bs->cpl && gp.W <= 128 && (((uintptr_t) store | (uintptr_t) store_stride) & 3u) == 0;
    if (bs_ok && (impl == 4 || (impl == 0 && n >= LRM_BS_MIN_READS))) {
        if (lrm_bs_launch(bs, lens, meta, meta_r, tlens, n, gp.T, gp.O, gp.W, store, store_stride, n_ops, score, counters,
                          bs_waves, stream)) return -1;
        // reads holding a byte other than ACGT (rare): one read per wavefront, flagged reads only
        return launch_one_per_wave(gp, n, stream, reads, stride, lens, meta, meta_r, content, tlens, store, store_stride,
                                   n_ops, score, counters, bs->flags);
    }
    if (impl != 1 && nblk <= 32) {
        size_t shmem3 = (size_t) 4 * 2 * ((size_t) gp.T + 2 * G2_PAD) * 4;
        gact2_fn_t fn3;
        if (nblk <= 26) fn3 = gp.W >= 128 ? gact3_kernel<true, 26> : gact3_kernel<false, 26>;
        else fn3 = gp.W >= 128 ? gact3_kernel<true, 32> : gact3_kernel<false, 32>;
        uint64_t blocks3 = (n + 7) / 8;
        hipLaunchKernelGGL(fn3, dim3((uint32_t) blocks3), dim3(256), shmem3, stream, reads, stride, lens, meta, meta_r,
                           content, tlens, n, gp.T, gp.O, gp.W, store, store_stride, n_ops, score, counters);
        return 0;
    }
    // T - O > 256 (more traceback planes than the packed kernel keeps in registers) or LRM_GACT_IMPL=1
    return launch_one_per_wave(gp, n, stream, reads, stride, lens, meta, meta_r, content, tlens, store, store_stride,
                               n_ops, score, counters, nullptr);
}

// ----------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

int lrm_launch_extend(lrm_index *idx, lrm_workspace *ws, char *d_reads, uint64_t stride,
                      const uint32_t *d_lens, uint64_t n, uint32_t max_len,
                      const lrm_entry *d_best, lrm_gact_params gp, uint8_t *d_store,
                      uint64_t store_stride, int32_t *d_n_ops, int32_t *d_score,
                      lrm_seq_meta *d_meta, int32_t *d_meta_r, const LrmMapTune &mt, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    if (n == 0) return 0;
    if (gp.T == 0 && gp.O == 0 && gp.W == 0) {
        gp.T = LRM_GACT_T_DEFAULT; gp.O = LRM_GACT_O_DEFAULT; gp.W = LRM_GACT_W_DEFAULT;
    }
    if (gp.T < 16 || gp.T > 512 || gp.O < 0 || gp.O >= gp.T || gp.W < 2 || (gp.W & 1) || gp.W > 1024) {
        lrm_set_error("unsupported GACT parameters T=%d O=%d W=%d (need 16<=T<=512, 0<=O<T, even 2<=W<=1024)",
                      gp.T, gp.O, gp.W);
        return -1;
    }
    if (store_stride < 2ull * max_len) {
        lrm_set_error("store_stride %llu < 2*max_len %u", (unsigned long long) store_stride, max_len);
        return -1;
    }
    lrm_time_begin(ws, LRM_K_LOCUS, stream);
    hipLaunchKernelGGL(locus_resolve_kernel, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, stream,
                       idx->view, d_best, d_lens, n, d_meta, d_meta_r);
    lrm_time_end(ws, stream);
    {
        const uint32_t half = max_len / 2 + 1;
        const uint32_t cpr = (half + RC_SEG - 1) / RC_SEG;                                  // workgroups per read
        const uint32_t seg = ((half + cpr - 1) / cpr + 15) & ~15u;                           // <= RC_SEG bases each
        uint64_t blocks = n * cpr;
        if (blocks > 0x7fffffffull) { lrm_set_error("revcomp grid too large: split the batch"); return -1; }
        lrm_time_begin(ws, LRM_K_REVCOMP, stream);
        hipLaunchKernelGGL(revcomp_kernel, dim3((uint32_t) blocks), dim3(256), 0, stream, d_reads, stride,
                           d_lens, d_meta, d_meta_r, n, cpr, seg);
        lrm_time_end(ws, stream);
    }
    LrmBsArgs bs = {};
    const bool want_bs = lrm_bs_wanted(gp, n, mt.gact_impl) && idx->d_cpl && idx->cpl_ok && ws->d_qpl && n <= ws->n_max &&
                         max_len <= ws->max_len;
    if (want_bs) {
        lrm_time_begin(ws, LRM_K_PACK_PLANAR, stream);
        if (lrm_bs_pack_reads(d_reads, stride, d_lens, n, max_len, ws->d_qpl, ws->qpl_wpr, ws->d_rflags, stream)) return -1;
        lrm_time_end(ws, stream);
        bs.qpl = ws->d_qpl; bs.wpr = ws->qpl_wpr; bs.flags = ws->d_rflags; bs.cpl = idx->d_cpl;
        bs.ckpt = ws->d_ckpt; bs.codes = ws->d_codes; bs.cw = ws->codes_cw; bs.ncodes = ws->d_ncodes;
    }
    const bool runs_bs = want_bs && (((uintptr_t) d_store | (uintptr_t) store_stride) & 3u) == 0;
    lrm_time_begin(ws, runs_bs ? LRM_K_GACT_BS : LRM_K_GACT, stream);
    if (gact_launch(gp, n, stream, d_reads, stride, d_lens, d_meta, d_meta_r, idx->view.content,
                    (const uint32_t *) nullptr, d_store, store_stride, d_n_ops, d_score, ws->d_counters,
                    want_bs ? &bs : nullptr, mt.gact_impl, mt.bs_waves)) return -1;
    lrm_time_end(ws, stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// direct kernel tap (tests only): simple_gact on one (q, d) pair, m may differ from n
extern "C" int lrm_debug_gact(const char *q, int n, const char *d, int m, lrm_gact_params gp, uint8_t *ops,
                              int *n_ops, int *score, int device) {
    LrmEnv env;                                  // a tap without a handle: LRM_GACT_IMPL as it stands now
    lrm_env_snapshot(&env);
    long long impl = 0;
    (void) env.get("LRM_GACT_IMPL", &impl);
    return lrm_debug_gact_impl(q, n, d, m, gp, (int) impl, ops, n_ops, score, device);
}

extern "C" int lrm_debug_gact_impl(const char *q, int n, const char *d, int m, lrm_gact_params gp, int impl, uint8_t *ops,
                                   int *n_ops, int *score, int device) {
    if (!q || !d || !ops || !n_ops || !score || n < 0 || m < 0) { lrm_set_error("bad argument"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        lrm_set_error("no HIP device available: liblrm_accel has no CPU fallback");
        return -1;
    }
    HIPCHK(hipSetDevice(device));
    if (gp.T == 0 && gp.O == 0 && gp.W == 0) {
        gp.T = LRM_GACT_T_DEFAULT; gp.O = LRM_GACT_O_DEFAULT; gp.W = LRM_GACT_W_DEFAULT;
    }
    if (gp.T < 16 || gp.T > 512 || gp.O < 0 || gp.O >= gp.T || gp.W < 2 || (gp.W & 1) || gp.W > 1024) {
        lrm_set_error("unsupported GACT parameters T=%d O=%d W=%d", gp.T, gp.O, gp.W);
        return -1;
    }
    // every device buffer is owned by a guard: an early HIPCHK return frees them all
    struct Buf {
        void *p = nullptr;
        ~Buf() { if (p) (void) hipFree(p); }
        int alloc(size_t bytes) { return hipMalloc(&p, bytes) == hipSuccess ? 0 : -1; }
    };
    Buf bq, bd, bops, bl, bm, br, bc, bqpl, bcpl, bfl, bcodes, bck;
    if (bq.alloc((size_t) n + 16) || bd.alloc((size_t) m + 16) || bops.alloc((size_t) n + m + 16) || bl.alloc(16) ||
        bm.alloc(sizeof(lrm_seq_meta)) || br.alloc(16) || bc.alloc(sizeof(LrmDevCounters))) { lrm_set_error("device allocation failed"); return -1; }
    char *dq = (char *) bq.p, *dd = (char *) bd.p;
    uint8_t *dops = (uint8_t *) bops.p;
    uint32_t *dl = (uint32_t *) bl.p;
    lrm_seq_meta *dm = (lrm_seq_meta *) bm.p;
    int32_t *dr = (int32_t *) br.p;
    LrmDevCounters *dc = (LrmDevCounters *) bc.p;
    HIPCHK(hipMemset(dc, 0, sizeof(LrmDevCounters)));
    uint32_t hl[2] = {(uint32_t) n, (uint32_t) m};
    lrm_seq_meta hm; hm.loc = 0; hm.off = 0; hm.seq_id = 0; hm.strand = 0;
    int32_t hr[3] = {1, 0, 0};
    HIPCHK(hipMemcpy(dq, q, (size_t) n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dd, d, (size_t) m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dl, hl, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dm, &hm, sizeof(hm), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr, hr, 12, hipMemcpyHostToDevice));
    LrmBsArgs bs = {};
    if (lrm_bs_wanted(gp, 1, impl)) {
        const uint64_t wq = lrm_bs_planar_words((uint64_t) n), wd = lrm_bs_planar_words((uint64_t) m);
        bs.cw = lrm_bs_code_words((uint32_t) (n > m ? n : m));
        if (bqpl.alloc(wq * 8 + 16) || bcpl.alloc(wd * 8 + 16) || bfl.alloc(16) || bcodes.alloc(bs.cw * 8 + 16) ||
            bck.alloc(lrm_bs_ckpt_words(1) * 4)) { lrm_set_error("device allocation failed"); return -1; }
        uint32_t *dfl = (uint32_t *) bfl.p;
        if (lrm_bs_pack_text(dd, (uint64_t) m, (uint64_t *) bcpl.p, dfl + 1, nullptr)) return -1;
        if (lrm_bs_pack_reads(dq, 0, dl, 1, (uint32_t) n, (uint64_t *) bqpl.p, wq, dfl, nullptr)) return -1;
        uint32_t tf = 0;
        HIPCHK(hipMemcpy(&tf, dfl + 1, 4, hipMemcpyDeviceToHost));
        bs.qpl = (uint64_t *) bqpl.p; bs.wpr = wq; bs.flags = dfl; bs.cpl = tf ? nullptr : (uint64_t *) bcpl.p;
        bs.codes = (uint64_t *) bcodes.p; bs.ckpt = (uint32_t *) bck.p; bs.ncodes = (int32_t *) (dfl + 2);
    }
    if (gact_launch(gp, 1, 0, dq, 0, dl, dm, dr, dd, dl + 1, dops, 0, dr + 1, dr + 2, dc, bs.qpl ? &bs : nullptr, impl, 0)) return -1;
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hr, dr, 12, hipMemcpyDeviceToHost));
    *n_ops = hr[1];
    *score = hr[2];
    if (hr[1] > 0) HIPCHK(hipMemcpy(ops, dops, (size_t) hr[1], hipMemcpyDeviceToHost));
    return 0;
}
