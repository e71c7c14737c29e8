// seed_kernels.hip -- gfx950 kernels for PART 1 of the accaln hot path
// (reference: alnmain.c:333-405; lchash.c:12-16,36-49,89-104; fmidx.c:18-33,277-313;
//  histo.c:26-56,84-96).  Integer gathers + LDS voting; no MFMA (nothing here is a
// contraction).
//
// Decomposition (MI355X-first, not the reference's per-read loop nest):
//   pack2bit      reads (1 B/base) -> 2 bit/base stream, so a seed is ONE bit-field window
//   seed_search   one lane per seed: lc lookup + FM backward extension; compact survivor lists   (K1, HBM gathers)
//   vote          one wavefront or workgroup per (read, phase): flat hit expansion, SA gather, LDS vote table (K2)
//   decide        one lane per read: the phase state machine of alnmain.c:371-403
//
// The reference evaluates phases one after another and stops at the first phase whose
// top-2 vote passes 0.6.  Phases are independent computations, so evaluating them
// speculatively and replaying the decisions in order is exact.  To avoid 21x waste on
// clean reads the host launches phase 0 first and phases 1..s only for undecided reads.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include "lrm_internal.h"


// A/a=0 C/c=1 G/g=2 T/t=3 ; other bytes are fenced (UB in the reference, lchash.c:38-44)
__device__ __forceinline__ uint32_t base_code(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }

// ----------------------------------------------------------------------------------------
// pack2bit: one thread per output dword (16 bases: one 16-byte load -- rows start at any byte, the hardware
// takes the unaligned dwords -- and one 4-byte store).  Bases past the read end pack as 0.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack2bit_kernel(const char *__restrict__ reads, uint64_t stride,
                                                       const uint32_t *__restrict__ lens,
                                                       uint8_t *__restrict__ out, uint64_t bytes_per_read,
                                                       uint32_t chunks_per_read, uint64_t n) {
    uint64_t read = blockIdx.x / chunks_per_read;
    uint32_t chunk = blockIdx.x % chunks_per_read;
    if (read >= n) return;
    uint64_t ow = (uint64_t) chunk * 256 + threadIdx.x;
    if (ow * 4 >= bytes_per_read) return;
    uint32_t len = lens[read];
    const uint8_t *r = (const uint8_t *) reads + read * stride;
    uint64_t p = ow * 16;
    uint32_t v = 0;
    if (p + 16 <= len) {
        uint32_t w[4];
        __builtin_memcpy(w, r + p, 16);
#pragma unroll
        for (int t = 0; t < 16; ++t) v |= base_code((w[t >> 2] >> (8 * (t & 3))) & 0xffu) << (2 * t);
    } else {
        for (int t = 0; t < 16; ++t) {
            uint32_t code = (p + t < len) ? base_code(r[p + t]) : 0u;
            v |= code << (2 * t);
        }
    }
    *reinterpret_cast<uint32_t *>(out + read * bytes_per_read + ow * 4) = v;
}

// ----------------------------------------------------------------------------------------
// FM LF-mapping: lf(c, loc) = C[c] + rank(c, loc), rank = # of c in bwt[0..loc] == _occ_access (fmidx.c:277-293)
// and C[] as fmi_aln adds it (fmidx.c:305-311).  One 16-byte gather {C[c] + prefix, mask}, one shift and one
// popcount.  The kernel is bound by the number of per-lane memory requests, then by its 64-bit index arithmetic,
// so the layout is built to make an LF step ONE request and the packer folds C[c] into the stored prefix (the
// first version selected C[c] from four scalar pairs in every step: 14 of its ~63 vector instructions).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t occ_lf_of(const ulonglong2 e, uint64_t loc) {
    return e.x + (uint64_t) __popcll(e.y << (63u - ((uint32_t) loc & 63u)));          // bits 0 .. loc % 64 of the mask
}

// the two LF values of one backward step; after the table lookup most intervals are a handful of rows,
// so k-1 and l usually fall into the same 64-row block and ONE 16-byte request serves both
// (returns the number of 16-byte requests it made: 1 or 2 -- only the counting build of seed_search looks at it)
__device__ __forceinline__ uint32_t occ_lf2(const LrmIndexView &ix, uint32_t c, uint64_t loc_a, uint64_t loc_b,
                                            uint64_t &ra, uint64_t &rb) {
    const ulonglong2 eb = *reinterpret_cast<const ulonglong2 *>(&ix.occ[loc_b >> 6].sym[c]);
    ulonglong2 ea = eb;
    const bool two = (loc_a >> 6) != (loc_b >> 6);
    if (two) ea = *reinterpret_cast<const ulonglong2 *>(&ix.occ[loc_a >> 6].sym[c]);
    ra = occ_lf_of(ea, loc_a);
    rb = occ_lf_of(eb, loc_b);
    return two ? 2u : 1u;
}

// SA[row].  Full SA: one 8-byte gather (sa_access, fmidx.c:18-33).  Sampled SA (LRM_SA_SAMPLED=r): only rows
// i*r are stored -- the reference's csa table (fmidx.c:153-163) -- and the other rows walk LF steps until
// they reach a stored row or the '$' row: SA[row] = SA[LF^t(row)] + t (csa_access, fmidx.c:315-331).  The
// bwt symbol of a row is the symbol whose occurrence mask holds the row's bit; the masks of the four symbols
// of a block share one 64-byte line.  The reference's own LF step there subtracts one row too many
// (fmidx.c:323, `- 1` on top of the inclusive rank: its walk leaves the text order and gives up after 5*ratio
// steps); this is the textbook LF, so that the locate equals sa_access on every row -- the two modes of this
// library give identical results, and csa_access itself is never called on the reference's hot path.
__device__ __forceinline__ uint64_t sa_locate(const LrmIndexView &ix, uint64_t row) {
    if (ix.sa_shift == 0) return ix.sa[row];
    const uint64_t rmask = (1ull << ix.sa_shift) - 1ull;
    uint64_t t = 0;
    while (row & rmask) {
        if (row == ix.dollar_row) return t;                             // SA[row] == 0
        const LrmOccBlock *b = &ix.occ[row >> 6];
        const uint32_t r = (uint32_t) row & 63u;
        const ulonglong2 e0 = *reinterpret_cast<const ulonglong2 *>(&b->sym[0]);
        const ulonglong2 e1 = *reinterpret_cast<const ulonglong2 *>(&b->sym[1]);
        const ulonglong2 e2 = *reinterpret_cast<const ulonglong2 *>(&b->sym[2]);
        const ulonglong2 e3 = *reinterpret_cast<const ulonglong2 *>(&b->sym[3]);
        const uint32_t c = (uint32_t) ((e1.y >> r) & 1ull) | ((uint32_t) ((e2.y >> r) & 1ull) << 1) | ((uint32_t) ((e3.y >> r) & 1ull) * 3u);
        const ulonglong2 e = c == 0 ? e0 : c == 1 ? e1 : c == 2 ? e2 : e3;
        row = occ_lf_of(e, row);                                        // LF(row) = C[c] + rank(c, row)
        ++t;
    }
    return ix.sa[row >> ix.sa_shift] + t;
}

// A survivor record's row field (40 bits) with bit 39 set holds the TEXT POSITION of a unique seed instead of its row: the
// seed table stores SA[k] next to such a seed (count code 0), so the vote stage has nothing to gather for it.  (Rows and
// positions stay below 2^39: 288 GB of HBM hold no longer text.)
#define LRM_LOCATED_BIT (1ull << 39)
__device__ __forceinline__ uint64_t sa_of_unique(const LrmIndexView &ix, uint64_t rec) {
    const uint64_t kk = rec & ((1ull << 40) - 1ull);
    return (kk & LRM_LOCATED_BIT) ? (kk & (LRM_LOCATED_BIT - 1ull)) : sa_locate(ix, kk);
}

// lc_access (lchash.c:12-16) on the 8-byte device entries
__device__ __forceinline__ void lc_lookup(const LrmIndexView &ix, uint64_t code, uint64_t &k, uint64_t &l) {
    const uint64_t e = ix.lc[code];
    k = e & ((1ull << 40) - 1ull);
    const uint64_t cnt = e >> 40;
    l = k + cnt - 1;
    if (e == 0) { k = 0; l = 0; }                                // absent hlen-mer
    else if (cnt == 0xFFFFFFull) {                               // interval too long for 24 bits: side table
        uint64_t lo = 0, hi = ix.n_lcx;
        while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ix.lcx[3 * mid] < code) lo = mid + 1; else hi = mid; }
        k = ix.lcx[3 * lo + 1];
        l = ix.lcx[3 * lo + 2];
    }
}

// ----------------------------------------------------------------------------------------
// SEED table: the whole of lc_aln + fmi_aln for a seed of sd_len bases in ONE memory line that the seeds of sd_f
// neighbouring read positions share.
//   The seeds at read positions p0 .. p0 + F - 1 (p0 a multiple of F = sd_f) share the CORE [p0 + F - 1, p0 + S): S - F + 1
//   bases.  A bijective hash of the core gives the line (its top sd_bits bits) and a residue (the rest); a slot of the line
//   is { k, count, tag } with tag = the seed's role r = p - p0, its F - 1 bases outside the core and the residue -- so a tag
//   names one S-mer exactly, and a lookup that finds no slot with its tag has proved the S-mer absent from the text
//   (rr = 0), which is how most seeds of a noisy read end.  Every distinct S-mer of the text is entered once per role with
//   the (k, l - k + 1) a real search gave for it (sd_build_kernel), so the table IS the reference's result, also where the
//   reference has a quirk (the '$' row).
//   8-byte slots (eight per line): k in the low sd_kbits bits, the count above, bit 63 - tagbits of slot 0 = "line
//   overflowed", the tag in the top bits.  6-byte slots (ten per line, texts of >= 2^32 rows): k | count << kbits | tag
//   << (48 - tagbits), the last four bytes of the line count its entries (> 10: overflowed).
//   Entries that found no room, and counts of all ones, are in a side hash table keyed by the S-mer.
// ----------------------------------------------------------------------------------------
struct SdKey { uint64_t line; uint64_t tag; uint32_t tb; };
// (The kernel that looks seeds up here is bound by its VECTOR INSTRUCTIONS once a seed costs a quarter of a line -- 3.95 G
//  wave-instructions per Gbp in 8.0 ms with a 64-bit multiplicative hash, 64-bit tag compares and a division per seed --
//  so the hash is ONE 32-bit multiply: the low 32 bits of the core times an odd constant (a bijection of those bits whose
//  TOP bits depend on all of them: they index the line), the bits of the core above 32 xor-ed with low bits of the product.)
__device__ __forceinline__ SdKey sd_key_of(const LrmIndexView &ix, uint64_t code, uint32_t r) {
    const uint32_t lf = ix.sd_f == 4 ? 2u : 1u, F = 1u << lf, lo_n = F - 1u - r;
    const uint32_t CL2 = 2u * ((uint32_t) ix.sd_len - F + 1u), wlo = CL2 < 32u ? CL2 : 32u;
    const uint64_t core = (code >> (2u * lo_n)) & ((1ull << CL2) - 1ull);
    const uint32_t extra = (uint32_t) (code & ((1ull << (2u * lo_n)) - 1ull)) | ((uint32_t) (code >> (2u * lo_n + CL2)) << (2u * lo_n));
    uint32_t m = (uint32_t) core * 0x9E3779B1u;
    if (wlo < 32u) m &= (1u << wlo) - 1u;
    const uint32_t chi = ((uint32_t) (core >> 32) ^ m) & ((1u << (CL2 - wlo)) - 1u);          // (0 when the core has <= 32 bits)
    const uint32_t rb = CL2 - (uint32_t) ix.sd_bits;                                          // < 32: sd_plan keeps sd_bits > CL2 - 32
    SdKey key;
    key.line = ((uint64_t) chi << (wlo - rb)) | (uint64_t) (m >> rb);
    key.tb = lf + 2u * (F - 1u) + rb;
    key.tag = (uint64_t) (r | (extra << lf)) | ((uint64_t) (m & ((1u << rb) - 1u)) << (lf + 2u * (F - 1u)));
    return key;
}
__device__ __forceinline__ uint32_t sd_filter_bit(uint32_t tag) { return (((tag * 0x9E3779B1u) >> 27) * 24u) >> 5; }   // 0 .. 23
// side table: true + entry (k | count << 40) when the S-mer is there
__device__ __forceinline__ bool sd_side_lookup(const LrmIndexView &ix, uint64_t code, uint64_t &e) {
    uint64_t slot = (code * 0x9E3779B97F4A7C15ull) >> 20 & ix.sdx_mask;
    for (;;) {
        const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(ix.sdx + 2 * slot);
        if (x.x == code + 1) { e = x.y; return true; }
        if (x.x == 0) return false;
        slot = (slot + 1) & ix.sdx_mask;
    }
}
// search of a fetched line.  0: absent (rr = 0); 1: k, c set; 2: take the other tables (a count beyond 24 bits)
// Slots fill from the front and an empty slot is all zeros, so the search runs from the LAST slot to the first with
// `e = match ? slot : e`: an empty slot can only "match" a tag of zero, a real entry before it overrides it, and e == 0 in
// the end means "not there".  The tag sits in the top bits of a slot: 32-bit compares on the high dword (8-byte slots, tags
// of <= 32 bits) or on the third halfword (6-byte slots, tags of <= 16 bits: the 2^31-line table of a GRCh38-sized text).
__device__ __forceinline__ int sd_search(const LrmIndexView &ix, const SdKey &key, uint64_t code, const uint64_t (&W)[8],
                                         uint64_t &k, uint64_t &c, uint32_t *cnt) {
    uint64_t e = 0;
    bool ovf;
    if (ix.sd_slot == 8) {
        if (key.tb <= 32u) {
            const uint32_t sh = 32u - key.tb, t32 = (uint32_t) key.tag;
            uint32_t elo = 0, ehi = 0;
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                const uint32_t hi = (uint32_t) (W[i] >> 32);
                const bool m = (hi >> sh) == t32;
                elo = m ? (uint32_t) W[i] : elo;
                ehi = m ? hi : ehi;
            }
            e = (uint64_t) elo | ((uint64_t) ehi << 32);
        } else {
#pragma unroll
            for (int i = 7; i >= 0; --i)
                if ((W[i] >> (64u - key.tb)) == key.tag) e = W[i];
        }
        ovf = (W[0] >> (63u - key.tb)) & 1ull;
    } else {
        uint32_t D[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { D[2 * i] = (uint32_t) W[i]; D[2 * i + 1] = (uint32_t) (W[i] >> 32); }
        if (key.tb <= 16u) {
            const uint32_t sh = 16u - key.tb, t32 = (uint32_t) key.tag;
            uint32_t elo = 0, ehi = 0;
#pragma unroll
            for (int i = 9; i >= 0; --i) {
                const int h2 = 3 * i + 2;                                              // the slot's third halfword: tag on top
                const uint32_t hw = (h2 & 1) ? D[h2 >> 1] >> 16 : D[h2 >> 1] & 0xFFFFu;
                const uint32_t lo = (i & 1) ? __builtin_amdgcn_alignbit(D[(3 * i + 1) >> 1], D[(3 * i) >> 1], 16) : D[(3 * i) >> 1];
                const bool m = (hw >> sh) == t32;
                elo = m ? lo : elo;
                ehi = m ? hw : ehi;
            }
            e = (uint64_t) elo | ((uint64_t) ehi << 32);
        } else {
#pragma unroll
            for (int i = 9; i >= 0; --i) {
                const int w = (48 * i) >> 6, off = (48 * i) & 63;
                uint64_t v = W[w] >> off;
                if (off > 16) v |= W[w + 1] << (64 - off);
                v &= (1ull << 48) - 1ull;
                if ((v >> (48u - key.tb)) == key.tag) e = v;
            }
        }
        // the line's last word: entry count in the low byte; above it a 24-bit filter of the tags that found no room -- a seed
        // the text does not hold (most seeds of a noisy read) goes on to the side table only when its filter bit is set
        ovf = (D[15] & 0xFFu) > 10u && ((D[15] >> (8u + sd_filter_bit((uint32_t) key.tag))) & 1u);
    }
    const uint64_t cmax = (1ull << ix.sd_cbits) - 1ull;
    if (e != 0) {
        k = e & ((1ull << ix.sd_kbits) - 1ull);
        c = (e >> ix.sd_kbits) & cmax;
        if (c == 0) { k |= LRM_LOCATED_BIT; c = 1; return 1; }        // a unique S-mer: the field is SA[k], not k (sa_of_unique)
        if (c != cmax) return 1;
    } else if (!ovf) {
        return 0;
    }
    uint64_t se;
    if (cnt) cnt[0] += 2;
    if (!sd_side_lookup(ix, code, se)) return e != 0 ? 2 : 0;           // (a saturated count without a side entry: never)
    if ((se >> 40) == 0xFFFFFFull) return 2;
    k = se & ((1ull << 40) - 1ull);
    c = se >> 40;
    return 1;
}
// a lane on its own: the whole line in ONE round trip (four independent 16-byte requests), searched in registers
__device__ __forceinline__ int sd_lookup(const LrmIndexView &ix, uint64_t win, uint32_t jpar, uint64_t &k, uint64_t &c, uint32_t *cnt) {
    const uint64_t code = win & ((1ull << (2 * ix.sd_len)) - 1ull);
    const SdKey key = sd_key_of(ix, code, jpar & (uint32_t) (ix.sd_f - 1));
    const uint64_t *line = ix.sd + key.line * 8;
    const ulonglong2 x0 = *reinterpret_cast<const ulonglong2 *>(line), x1 = *reinterpret_cast<const ulonglong2 *>(line + 2);
    const ulonglong2 x2 = *reinterpret_cast<const ulonglong2 *>(line + 4), x3 = *reinterpret_cast<const ulonglong2 *>(line + 6);
    if (cnt) cnt[0] += 1;
    const uint64_t W[8] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y, x3.x, x3.y};
    return sd_search(ix, key, code, W, k, c, cnt);
}
template <int CTRL>
__device__ __forceinline__ uint64_t quad_perm64(uint64_t v) {
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) v, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (v >> 32), CTRL, 0xf, 0xf, true);
    return (uint64_t) lo | ((uint64_t) hi << 32);
}
// The lanes of a wavefront hold CONSECUTIVE read positions (lane id == position, mod sd_f): the sd_f lanes that share a line
// fetch it TOGETHER -- every lane one quarter (half) of the line of the group's first lane, a contiguous 64-byte request per
// group instead of four 16-byte requests per lane -- and pass the pieces around inside the quad (DPP quad_perm: no LDS).
// Every lane of the wavefront must be here (lanes without a seed come with win = 0 and ignore the answer); a lane whose
// group's first lane has no seed has none either (positions grow with the lane id).
// sd_issue_shared starts the fetch (the lane's piece: a, and b with two positions per line), sd_finish_shared gathers the
// line and searches it.  (Keeping the fetches of 2 / 4 / 8 of a lane's seeds in flight between the two: 7.85 / 8.56 / 10.3 ms
// per Gbp on the bench workload against 7.63 for one -- the kernel is not short of requests in flight.)
__device__ __forceinline__ void sd_issue_shared(const LrmIndexView &ix, const SdKey &key, ulonglong2 &a, ulonglong2 &b) {
    const uint32_t lane = __lane_id();
    if (ix.sd_f == 4) {
        const uint64_t line = quad_perm64<0x00>(key.line);                                    // quad_perm [0,0,0,0]
        a = *reinterpret_cast<const ulonglong2 *>(ix.sd + line * 8 + (lane & 3u) * 2);
        b = a;
    } else {
        const uint64_t line = quad_perm64<0xA0>(key.line);                                    // [0,0,2,2]
        const uint64_t *src = ix.sd + line * 8 + (lane & 1u) * 4;
        a = *reinterpret_cast<const ulonglong2 *>(src);
        b = *reinterpret_cast<const ulonglong2 *>(src + 2);
    }
}
__device__ __forceinline__ int sd_finish_shared(const LrmIndexView &ix, const SdKey &key, uint64_t code, const ulonglong2 &a, const ulonglong2 &b,
                                                uint64_t &k, uint64_t &c, uint32_t *cnt) {
    uint64_t W[8];
    if (ix.sd_f == 4) {
        W[0] = quad_perm64<0x00>(a.x); W[1] = quad_perm64<0x00>(a.y);
        W[2] = quad_perm64<0x55>(a.x); W[3] = quad_perm64<0x55>(a.y);                         // [1,1,1,1]
        W[4] = quad_perm64<0xAA>(a.x); W[5] = quad_perm64<0xAA>(a.y);                         // [2,2,2,2]
        W[6] = quad_perm64<0xFF>(a.x); W[7] = quad_perm64<0xFF>(a.y);                         // [3,3,3,3]
    } else {
        const uint64_t pa = quad_perm64<0xB1>(a.x), pb = quad_perm64<0xB1>(a.y);              // [1,0,3,2]: the partner's half
        const uint64_t pc = quad_perm64<0xB1>(b.x), pd = quad_perm64<0xB1>(b.y);
        const bool odd = __lane_id() & 1u;
        W[0] = odd ? pa : a.x; W[1] = odd ? pb : a.y; W[2] = odd ? pc : b.x; W[3] = odd ? pd : b.y;
        W[4] = odd ? a.x : pa; W[5] = odd ? a.y : pb; W[6] = odd ? b.x : pc; W[7] = odd ? b.y : pd;
    }
    if (cnt) cnt[0] += 1;
    return sd_search(ix, key, code, W, k, c, cnt);
}

// lc_aln (lchash.c:89-104) + fmi_aln (fmidx.c:295-313) on the packed read.
// win: bases j.. of the read, 2 bits each, LSB first.  Returns rr; k,l as the reference
// leaves them (also on failure).
// jpar: parity of the seed's read position (only the pair-line layout of the long table looks at it).
// cnt (counting build only): cnt[0] += 8-byte table lookups, cnt[1] += 16-byte rank requests of this seed.
__device__ __forceinline__ uint64_t seed_one(const LrmIndexView &ix, uint64_t win, int seed_len, uint32_t jpar,
                                             uint64_t &k, uint64_t &l, uint32_t *cnt = nullptr) {
    int left = seed_len - ix.hlen;
    bool looked_up = false;
    if (ix.sd && seed_len == ix.sd_len) {
        uint64_t c;
        const int st = sd_lookup(ix, win, jpar, k, c, cnt);
        if (st == 0) { k = 0; l = 0; return 0; }                          // the text does not hold this seed
        if (st == 1) { l = k + c - 1; return c; }
    }
    if (ix.core && seed_len >= 16) {
        // CORE table (small texts): the 16-mers of read positions p0 .. p0 + 3 (p0 a multiple of 4) share the 13 bases
        // [p0 + 3, p0 + 16) of their windows; the line of that 13-mer holds the entries of the text's 16-mers around it
        // (eight 8-byte slots: k | count << 40 | tag << 56, tag = the window's role r = p & 3 and its 3 bases outside
        // the core), so the four lanes read ONE line.  Slots fill from a tag-dependent home pair onwards (no deletions:
        // an empty slot ends the search); a line that would need more than eight slots is all ones: such 16-mers take
        // the pair-line table below.
        const int left2 = seed_len - 16;
        const uint64_t W = (win >> (2 * left2)) & 0xFFFFFFFFull;                        // the seed's last 16 bases, first base lowest
        const uint32_t r = jpar & 3u;
        const uint64_t corec = (W >> (2 * (3 - r))) & ((1ull << 26) - 1ull);
        const uint32_t extra = (uint32_t) (W & ((1ull << (2 * (3 - r))) - 1ull)) | ((uint32_t) (W >> (2 * (16 - r))) << (2 * (3 - r)));
        const uint32_t tag = r | (extra << 2);
        const uint64_t *line = ix.core + corec * 8;
        // The whole line in ONE round trip (four independent 16-byte requests to one 64-byte line), searched in registers.
        // (Measured on the bench workload, ms per Gbp: no core table 15.4-15.7; a search walking the line pair by pair
        //  from a tag-dependent home pair 17.3 -- every step is a dependent round trip for the whole wavefront; the first
        //  half of the line, the second only when the first is full of other 16-mers 14.2 -- the repeat family's lines
        //  are, and some lane of nearly every wavefront sits in one; the whole line at once 13.2.)
        const ulonglong2 x0 = *reinterpret_cast<const ulonglong2 *>(line), x1 = *reinterpret_cast<const ulonglong2 *>(line + 2);
        const ulonglong2 x2 = *reinterpret_cast<const ulonglong2 *>(line + 4), x3 = *reinterpret_cast<const ulonglong2 *>(line + 6);
        if (cnt) cnt[0] += 1;
        const uint64_t sl[8] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y, x3.x, x3.y};
        uint64_t e = 0;
        int state = x0.x == ~0ull ? 3 : 2;                               // (an overflowed line is all ones)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (sl[i] != 0 && sl[i] != ~0ull && (uint32_t) (sl[i] >> 56) == tag) { e = sl[i]; state = 1; }
        if (state != 3) {
            if (state != 1) { k = 0; l = 0; return 0; }                  // dead by its 16th base
            k = e & ((1ull << 40) - 1ull);
            l = k + ((e >> 40) & 0xFFFFull) - 1;
            left = left2;
            looked_up = true;
        }
    }
    if (!looked_up && ix.lcl && seed_len >= ix.hl) {
        // Long table: entry[hl-mer] = lc[hlen-mer] followed by hl - hlen backward steps, precomputed on the device
        // (lcl_build_kernel) -- the same (k, l) the reference reaches after those steps, for one memory request
        // instead of 1 + 2(hl - hlen).  The kernel is bound by the number of requests, and most seeds of a noisy
        // read die inside their last hl bases.
        const int left2 = seed_len - ix.hl;
        uint64_t at = (win >> (2 * left2)) & ((1ull << (2 * ix.hl)) - 1ull);          // the seed's last hl bases, first base lowest
        if (ix.lcl_pair) {
            // PAIR-LINE layout: one 64-byte line per (hl-1)-mer S holds the entries of its four left extensions a.S and
            // of its four right extensions S.b.  The seed at an even read position j looks its hl-mer up as a.S, the seed
            // at j + 1 as S.b with the SAME S (its hl-mer without its last base = the hl-mer of j without its first):
            // the two lanes of neighbouring positions read one line, and the texture path merges them into one request.
            const uint64_t smask = (1ull << (2 * (ix.hl - 1))) - 1ull;
            at = (jpar & 1u) ? ((at & smask) << 3) + 4u + (at >> (2 * (ix.hl - 1)))
                             : ((at >> 2) << 3) + (at & 3u);
        }
        uint64_t e;
        if (ix.lcl_kbits) {
            // 5-byte entries (pair-line layout only: 40 bytes per (hl-1)-mer): k in the low kbits bits, the count above;
            // a count of all ones sends the hl-mer to the side hash table (an 8-byte entry per such hl-mer)
            uint64_t v;
            __builtin_memcpy(&v, reinterpret_cast<const uint8_t *>(ix.lcl) + at * 5, 8);     // one unaligned 8-byte request
            v &= (1ull << 40) - 1ull;
            const uint64_t c5 = v >> ix.lcl_kbits, cmax = (1ull << (40 - ix.lcl_kbits)) - 1ull;
            e = v == 0 ? 0ull : (v & ((1ull << ix.lcl_kbits) - 1ull)) | (c5 << 40);
            if (c5 == cmax) {
                const uint64_t code = (win >> (2 * left2)) & ((1ull << (2 * ix.hl)) - 1ull);
                uint64_t slot = (code * 0x9E3779B97F4A7C15ull) >> 20 & ix.lclx_mask;
                for (;;) {                                                // the hl-mer is in the table: the packer put it there
                    const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(ix.lclx + 2 * slot);
                    if (x.x == code + 1) { e = x.y; break; }
                    if (x.x == 0) { e = 0xFFFFFFull << 40; break; }       // (never: defensive, takes the reference's path)
                    slot = (slot + 1) & ix.lclx_mask;
                }
                if (cnt) cnt[0] += 1;
            }
        } else {
            e = ix.lcl[at];
        }
        if (cnt) cnt[0] += 1;
        if ((e >> 40) != 0xFFFFFFull) {                               // (marker: interval too long for 24 bits)
            if (e == 0) { k = 0; l = 0; return 0; }                   // dead by its hl-th base; k, l are dead values then
            k = e & ((1ull << 40) - 1ull);
            l = k + (e >> 40) - 1;
            left = left2;
            looked_up = true;
        }
    }
    if (!looked_up) {
        if (left >= 0) {
            lc_lookup(ix, (win >> (2 * left)) & ((1ull << (2 * ix.hlen)) - 1ull), k, l);
            if (cnt) cnt[0] += 1;
        } else {
            k = 1;
            l = ix.length - 1;
        }
        if (k == 0 && l == 0) return 0;
    }
    for (int i = left - 1; i >= 0; --i) {
        uint32_t c = (uint32_t) (win >> (2 * i)) & 3u;
        uint64_t ra, rb;
        const uint32_t nreq = occ_lf2(ix, c, k - 1, l, ra, rb);
        if (cnt) cnt[1] += nreq;
        k = ra + 1;
        l = rb;
        if (k > l) break;
    }
    return k > l ? 0 : l - k + 1;
}

// entry of the long table for one hl-mer (code: first base lowest): the lc entry of its last hlen bases followed by
// hl - hlen backward steps -- the (k, l) the reference reaches after those steps -- as k | count << 40, 0 = absent,
// count 0xFFFFFF = "too long for 24 bits: take the reference's path"
__device__ __forceinline__ uint64_t lcl_entry(const LrmIndexView &ix, int hl, uint64_t code) {
    const int ext = hl - ix.hlen;
    uint64_t k, l;
    lc_lookup(ix, code >> (2 * ext), k, l);
    if (k == 0 && l == 0) return 0;
    for (int i = ext - 1; i >= 0 && k <= l; --i) {
        const uint32_t c = (uint32_t) (code >> (2 * i)) & 3u;
        uint64_t ra, rb;
        occ_lf2(ix, c, k - 1, l, ra, rb);
        k = ra + 1;
        l = rb;
    }
    if (k > l) return 0;
    const uint64_t cnt = l - k + 1;
    return cnt >= 0xFFFFFFull ? (0xFFFFFFull << 40) : (k | (cnt << 40));
}

// Long table, one lane per slot.  PLAIN layout: slot = hl-mer code.  PAIR-LINE layout (see seed_one): line S (an
// (hl-1)-mer), slot a < 4: the entry of a.S; slot 4 + b: the entry of S.b -- every hl-mer is stored twice (once as a
// left, once as a right extension of an (hl-1)-mer): 16 bytes per hl-mer, or 10 with 5-byte entries (kbits > 0).
// 5-byte entries whose count does not fit go to `ovf` ({code, entry} pairs, appended once per hl-mer: from its
// left-extension slot) for the side hash table.
__global__ __launch_bounds__(256) void lcl_build_kernel(LrmIndexView ix, int hl, int pair, int kbits, uint64_t *__restrict__ out,
                                                        uint64_t slot0, uint64_t *__restrict__ ovf, uint64_t ovf_cap,
                                                        unsigned long long *__restrict__ n_ovf) {
    const uint64_t slot = slot0 + (uint64_t) blockIdx.x * 256 + threadIdx.x;
    uint64_t code = slot;
    uint32_t w = 0;
    if (pair) {
        const uint64_t S = slot >> 3;
        if (S >= (1ull << (2 * (hl - 1)))) return;
        w = (uint32_t) slot & 7u;
        code = w < 4 ? ((S << 2) | w) : (S | ((uint64_t) (w - 4) << (2 * (hl - 1))));
    } else if (code >= (1ull << (2 * hl))) {
        return;
    }
    const uint64_t e = lcl_entry(ix, hl, code);
    if (!kbits) { out[slot] = e; return; }
    const uint64_t cmax = (1ull << (40 - kbits)) - 1ull, c = e >> 40;
    uint64_t v = e & ((1ull << 40) - 1ull);                            // k (< 2^kbits)
    if (e != 0) {
        if (c < cmax) v |= c << kbits;
        else {
            v |= cmax << kbits;
            if (w < 4) {
                const unsigned long long at = atomicAdd(n_ovf, 1ull);
                if (at < ovf_cap) { ovf[2 * at] = code; ovf[2 * at + 1] = e; }
            }
        }
    }
    uint8_t *p = reinterpret_cast<uint8_t *>(out) + slot * 5;
    const uint32_t lo = (uint32_t) v;
    __builtin_memcpy(p, &lo, 4);
    p[4] = (uint8_t) (v >> 32);
}

__global__ __launch_bounds__(256) void lclx_build_kernel(const uint64_t *__restrict__ ovf, uint64_t n, uint64_t *__restrict__ table, uint64_t mask) {
    const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = ovf[2 * i], e = ovf[2 * i + 1];
    uint64_t slot = (code * 0x9E3779B97F4A7C15ull) >> 20 & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long *) &table[2 * slot], 0ull, (unsigned long long) (code + 1));
        if (prev == 0ull || prev == code + 1) { table[2 * slot + 1] = e; return; }
        slot = (slot + 1) & mask;
    }
}

// Core table (see seed_one): every 16-mer X the text holds (a non-zero entry of the pair-line table `pl`) enters the lines
// of its four cores, once per role.  A line that cannot take an entry (all eight slots taken, or a count beyond 16 bits)
// goes on the overflow list and is set to all ones afterwards.
__device__ __forceinline__ void core_insert(uint64_t *core, uint64_t corec, uint32_t tag, uint64_t entry, bool fits, uint64_t *ovf,
                                            uint64_t ovf_cap, unsigned long long *n_ovf) {
    uint64_t *line = core + corec * 8;
    (void) tag;
    if (fits)
        for (int sl = 0; sl < 8; ++sl)                                     // slots fill from the front
            if (atomicCAS((unsigned long long *) &line[sl], 0ull, (unsigned long long) entry) == 0ull) return;
    const unsigned long long at = atomicAdd(n_ovf, 1ull);
    if (at < ovf_cap) ovf[at] = corec;
}
__global__ __launch_bounds__(256) void core_build_kernel(const uint64_t *__restrict__ pl, uint64_t *__restrict__ core, uint64_t x0,
                                                         uint64_t *__restrict__ ovf, uint64_t ovf_cap, unsigned long long *n_ovf) {
    const uint64_t X = x0 + (uint64_t) blockIdx.x * 256 + threadIdx.x;                  // a 16-mer, first base lowest
    if (X >> 32) return;
    const uint64_t e = pl[((X >> 2) << 3) + (X & 3u)];                                  // its entry as the left extension of its last 15 bases
    if (e == 0) return;
    const uint64_t c = e >> 40;
    const bool fits = c < 0xFFFFull;
    const uint64_t body = (e & ((1ull << 40) - 1ull)) | ((c & 0xFFFFull) << 40);
    for (uint32_t r = 0; r < 4; ++r) {
        const uint64_t corec = (X >> (2 * (3 - r))) & ((1ull << 26) - 1ull);
        const uint32_t extra = (uint32_t) (X & ((1ull << (2 * (3 - r))) - 1ull)) | ((uint32_t) (X >> (2 * (16 - r))) << (2 * (3 - r)));
        const uint32_t tag = r | (extra << 2);
        core_insert(core, corec, tag, body | ((uint64_t) tag << 56), fits, ovf, ovf_cap, n_ovf);
    }
}
__global__ __launch_bounds__(256) void core_ovf_kernel(uint64_t *__restrict__ core, const uint64_t *__restrict__ ovf, uint64_t n) {
    const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 8) return;
    core[ovf[i >> 3] * 8 + (i & 7)] = ~0ull;
}

// Seed table build, one lane per text position p: the S-mer at p is searched as a seed would be (through whatever
// tables the handle already has); the first lane to claim the interval's first row k in a bitmap over the rows (distinct
// S-mers have disjoint intervals) enters it -- once per distinct S-mer -- into the lines of its F roles.  What does not
// fit (a full line, a count of cmax or more) goes on the list for the side hash table.  The suffix array is read only
// for the text position a unique S-mer takes along (deduplication goes by the claimed rows, not by SA values).
__global__ __launch_bounds__(256) void sd_build_kernel(LrmIndexView ix, LrmIndexView sdv, uint64_t *__restrict__ sd, uint64_t p0,
                                                       uint32_t *__restrict__ claimed,
                                                       uint64_t *__restrict__ ovf, uint64_t ovf_cap, unsigned long long *__restrict__ n_ovf) {
    const uint64_t p = p0 + (uint64_t) blockIdx.x * 256 + threadIdx.x;
    const int S = sdv.sd_len;
    if (ix.con_len < (uint64_t) S + 1 || p > ix.con_len - 1 - (uint64_t) S) return;          // content[con_len - 1] is the '$'
    uint64_t code = 0;
    for (int i = 0; i < S; ++i) code |= (uint64_t) base_code((uint8_t) ix.content[p + i]) << (2 * i);
    uint64_t k, l;
    const uint64_t rr = seed_one(ix, code, S, (uint32_t) p, k, l);                            // (ix.sd is null here)
    if (rr == 0) return;
    if (atomicOr(&claimed[k >> 5], 1u << (k & 31u)) & (1u << (k & 31u))) return;               // another occurrence entered this S-mer
    const uint64_t cmax = (1ull << sdv.sd_cbits) - 1ull;
    uint64_t c = rr < cmax ? rr : cmax;
    uint64_t side = k | ((rr < 0xFFFFFFull ? rr : 0xFFFFFFull) << 40);
    if (rr == 1 && sdv.sd_kbits <= 38) {
        // a unique S-mer takes its text position along (count code 0): most hits of a read come from unique seeds, and every
        // one of them was a random 64-byte line of the suffix array in the vote stage
        const uint64_t pos = sa_locate(ix, k);
        if (pos >= 1 && pos < (1ull << sdv.sd_kbits)) { k = pos; c = 0; side = pos | LRM_LOCATED_BIT | (1ull << 40); }
    }
    bool to_side = rr >= cmax;
    for (uint32_t r = 0; r < (uint32_t) sdv.sd_f; ++r) {
        const SdKey key = sd_key_of(sdv, code, r);
        uint64_t *line = sd + key.line * 8;
        bool placed = false;
        if (sdv.sd_slot == 8) {
            const uint64_t v = k | (c << sdv.sd_kbits) | (key.tag << (64u - key.tb));
            for (int sl = 0; sl < 8 && !placed; ++sl)
                placed = atomicCAS((unsigned long long *) &line[sl], 0ull, (unsigned long long) v) == 0ull;
            if (!placed) atomicOr((unsigned long long *) &line[0], 1ull << (63u - key.tb));
        } else {
            const uint64_t v = k | (c << sdv.sd_kbits) | (key.tag << (48u - key.tb));
            const uint32_t at = atomicAdd(reinterpret_cast<uint32_t *>(line) + 15, 1u) & 0xFFu;
            if (at >= 250u) atomicAdd(n_ovf, 1ull << 40);                                     // (the count byte would run into the filter: give the table up)
            if (at >= 10u) atomicOr(reinterpret_cast<uint32_t *>(line) + 15, 1u << (8u + sd_filter_bit((uint32_t) key.tag)));
            if (at < 10u) {
                uint16_t *h = reinterpret_cast<uint16_t *>(line) + 3 * at;                   // three 2-byte stores: slots are 6 bytes apart
                h[0] = (uint16_t) v; h[1] = (uint16_t) (v >> 16); h[2] = (uint16_t) (v >> 32);
                placed = true;
            }
        }
        if (!placed) to_side = true;
    }
    if (to_side) {
        const unsigned long long at = atomicAdd(n_ovf, 1ull);
        if (at < ovf_cap) { ovf[2 * at] = code; ovf[2 * at + 1] = side; }
    }
}

__global__ __launch_bounds__(256) void sd_clear_kernel(ulonglong2 *__restrict__ p, uint64_t n16) {
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * 256) p[i] = make_ulonglong2(0, 0);
}

// Geometry and build of the seed table (see sd_lookup).  Lines: the smallest power of two that keeps the average line
// at <= 2.2 entries of 8 (four positions per line: 4 entries per distinct S-mer) or, where that does not fit, at <= 6 of
// 10 (two positions per line, 6-byte slots) -- an E. coli-sized text: 2 GiB; chr1-sized: 64 GiB; GRCh38-sized (6.2 G
// rows): 128 GiB, 3.5 % of the lines overflow into a side table of ~0.13 G entries.
struct SdPlan { int f, bits, slot, kbits, cbits; uint64_t bytes; };
static bool sd_plan(const lrm_index *idx, uint64_t free_b, SdPlan *pl) {
    const LrmIndexTune &tu = idx->itune;
    const uint64_t L = idx->view.length;
    const int S = tu.sd_len;
    if (tu.sd == 0 || !idx->cpl_ok || L < 64 || idx->view.con_len != L) return false;
    int kbits = 1;
    while ((1ull << kbits) < L) ++kbits;
    for (int f = 4; f >= 2; f -= 2) {
        if (tu.sd_f && tu.sd_f != f) continue;
        const int slot = f == 4 ? 8 : 6, lf = f == 4 ? 2 : 1;
        const double target = f == 4 ? 2.2 : 6.0;
        int bits = 10;
        while ((double) f * (double) L / (double) (1ull << bits) > target && bits < 34) ++bits;
        if (tu.sd_bits) bits = tu.sd_bits;
        const int CL2 = 2 * (S - f + 1);
        if (bits > CL2) bits = CL2;
        if (bits < 11) bits = 11;
        if (bits < CL2 - 31) bits = CL2 - 31;                                    // (sd_key_of: fewer than 32 residue bits)
        const int tb = lf + 2 * (f - 1) + (CL2 - bits);
        int cbits = (slot == 8 ? 63 : 48) - tb - kbits;
        if (cbits < (tu.sd_bits ? 2 : 4) || tb > 40) continue;                  // (tests force few lines: long tags)
        if (cbits > 24) cbits = 24;
        if (tu.sd_cbits && tu.sd_cbits < cbits) cbits = tu.sd_cbits;
        const uint64_t bytes = 64ull << bits;
        // room: the table, its side table (<= 1/8 of it) and what the batch workspaces need afterwards
        const uint64_t spare = bytes >= (32ull << 30) ? (40ull << 30) : (8ull << 30);
        if (tu.sd < 0 && ((uint64_t) free_b < bytes + bytes / 8 + spare || (tu.lc_long_max >= 13 && bytes > (16ull << 30)))) continue;
        pl->f = f; pl->bits = bits; pl->slot = slot; pl->kbits = kbits; pl->cbits = cbits; pl->bytes = bytes;
        return true;
    }
    return false;
}
static int sd_build(lrm_index *idx, const SdPlan &pl) {
    uint64_t *d = nullptr, *ovf = nullptr, *tab = nullptr;
    uint32_t *claimed = nullptr;
    unsigned long long *n_ovf = nullptr;
    const char *why = "";
    unsigned long long n = 0;
    auto give_up = [&]() {
        if (idx->mtune.verbose) fprintf(stderr, "[lrm] seed table (share %d, 2^%d lines, %d-byte slots) not built: %s (side entries %llu)\n", pl.f, pl.bits, pl.slot, why, n);
        if (d) (void) hipFree(d); if (ovf) (void) hipFree(ovf); if (tab) (void) hipFree(tab); if (n_ovf) (void) hipFree(n_ovf);
        if (claimed) (void) hipFree(claimed);
        (void) hipGetLastError(); return 0; };
    const uint64_t L = idx->view.length, lines = 1ull << pl.bits;
    // (a core that occurs once in the text brings one entry PER ROLE to its line, so a line holds F x Poisson entries: with two
    //  positions per line and 2.9 cores per line on average 7 % of the lines of a GRCh38-sized text need more than ten slots)
    uint64_t ovf_cap = lines / 4 + 4096;
    if (ovf_cap > (1ull << 30)) ovf_cap = 1ull << 30;
    if (idx->itune.sd_bits) ovf_cap = (uint64_t) pl.f * L + 4096;                 // (tests force crowded lines)
    why = "no room for the table";
    if (hipMalloc(&d, pl.bytes) != hipSuccess) { d = nullptr; return give_up(); }
    why = "no room for the overflow list";
    if (hipMalloc(&ovf, ovf_cap * 16) != hipSuccess) { ovf = nullptr; return give_up(); }
    if (hipMalloc(&n_ovf, 8) != hipSuccess) { n_ovf = nullptr; return give_up(); }
    const uint64_t cl_bytes = ((L + 31) / 32 + 1) * 4;
    if (hipMalloc(&claimed, cl_bytes) != hipSuccess) { claimed = nullptr; return give_up(); }
    why = "memset failed";
    if (hipMemset(claimed, 0, cl_bytes) != hipSuccess) return give_up();
    hipLaunchKernelGGL(sd_clear_kernel, dim3(256 * 64), dim3(256), 0, 0, reinterpret_cast<ulonglong2 *>(d), pl.bytes / 16);     // (128 GiB: not a hipMemset)
    if (hipGetLastError() != hipSuccess || hipMemset(n_ovf, 0, 8) != hipSuccess) return give_up();
    LrmIndexView sdv = idx->view;
    sdv.sd_len = idx->itune.sd_len; sdv.sd_f = pl.f; sdv.sd_bits = pl.bits; sdv.sd_kbits = pl.kbits; sdv.sd_slot = pl.slot; sdv.sd_cbits = pl.cbits;
    const uint64_t chunk = 1ull << 22;
    for (uint64_t b0 = 0, blocks = (L + 255) / 256; b0 < blocks; b0 += chunk) {
        const uint64_t nb = blocks - b0 < chunk ? blocks - b0 : chunk;
        hipLaunchKernelGGL(sd_build_kernel, dim3((uint32_t) nb), dim3(256), 0, 0, idx->view, sdv, d, b0 * 256, claimed, ovf, ovf_cap, n_ovf);
    }
    why = "build kernel failed";
    if (hipDeviceSynchronize() != hipSuccess) { give_up(); lrm_set_error("seed table build failed"); return -1; }
    why = "too many entries beside their lines";
    if (hipMemcpy(&n, n_ovf, 8, hipMemcpyDeviceToHost) != hipSuccess || n > ovf_cap) return give_up();       // too crowded: the other tables alone
    uint64_t tslots = 1024;
    while (tslots < 2 * n) tslots <<= 1;
    why = "no room for the side table";
    if (hipMalloc(&tab, tslots * 16) != hipSuccess) { tab = nullptr; return give_up(); }
    if (hipMemset(tab, 0, tslots * 16) != hipSuccess) return give_up();
    if (n) hipLaunchKernelGGL(lclx_build_kernel, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, 0, ovf, (uint64_t) n, tab, tslots - 1);
    if (hipDeviceSynchronize() != hipSuccess) { give_up(); lrm_set_error("seed table side build failed"); return -1; }
    (void) hipFree(ovf); (void) hipFree(n_ovf); (void) hipFree(claimed);
    idx->d_sd = d; idx->d_sdx = tab;
    idx->view.sd = d; idx->view.sdx = tab; idx->view.sdx_mask = tslots - 1;
    idx->view.sd_len = sdv.sd_len; idx->view.sd_f = pl.f; idx->view.sd_bits = pl.bits; idx->view.sd_kbits = pl.kbits; idx->view.sd_slot = pl.slot;
    idx->view.sd_cbits = pl.cbits;
    idx->sd_side_entries = n;
    if (idx->mtune.verbose) fprintf(stderr, "[lrm] seed table: %d positions per line, 2^%d lines, %d-byte slots, %d count bits, %llu side entries\n", pl.f, pl.bits, pl.slot, pl.cbits, n);
    return 0;
}

// The long seed table.  seed_search's time is its L2 misses divided by ~50 G random 64-byte lines per second
// (tools/randline_bench.hip pins that rate independently), and the first lookup of a seed is a miss whatever the text,
// so the table is (a) as long as HBM allows -- the longer the k-mer, the more noisy seeds die in the lookup instead of one
// random step later -- and (b) in the pair-line layout, where the lookups of two neighbouring read positions share a
// line.  Measured on 100 k x 10 kbp ONT reads, ms per Gbp [r2]: E. coli-sized text plain 13-mers 24.5, pair-line
// 13 / 14 / 15 / 16-mers 19.2 / 18.5 / 17.6 / 15.6; chr1-sized text plain 16 28.4, pair-line 16 20.3; GRCh38-sized text
// plain 16 40.6, plain 17 (128 GiB) 32.4, pair-line 16 (64 GiB) 30.2.
// Automatic choice: texts of >= 2^32 rows (every 16-mer occurs: the lookup decides nothing there) take pair-line 17-mers
// with 5-BYTE entries (160 GiB) when that leaves 40 GiB of HBM free; otherwise pair-line 16-mers with 8-byte entries
// (64 GiB) when that leaves 64 GiB free, else 15 (16 GiB, leaving 32), 14 (4 GiB, leaving 8), 13 (1 GiB).
// lrm_index_options lc_long = 0 (off) | 13..17, lc_pair = 0 | 1, lc_entry_bytes = 5 | 8 override.  A table that cannot
// be allocated is skipped: results never depend on it.  Cost at upload [r2]: 16 GiB and below ~10 ms, the 64 GiB
// table 0.65 s (2 s when the memory was freed a moment ago) -- repaid after a few hundred Gbp of reads, so callers that
// know their run is short cap the length (lrm_index_options.lc_long_max; lrm_accaln does it from the size of the reads
// file).
static int lcl_prepare_tables(lrm_index *idx, size_t free_b, bool have_sd);
int lrm_lcl_prepare_index(lrm_index *idx) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); free_b = 0; }
    // The seed table (seeds of the usual length) is planned first and built last, through the tables made here for the seeds
    // of any other length; those make do with the HBM it leaves.
    SdPlan sdp;
    const bool want_sd = sd_plan(idx, free_b, &sdp);
    if (want_sd) { const uint64_t need = sdp.bytes + sdp.bytes / 8; free_b = free_b > need ? free_b - need : 0; }
    const int rc_lcl = lcl_prepare_tables(idx, free_b, want_sd);
    if (rc_lcl) return rc_lcl;
    return want_sd ? sd_build(idx, sdp) : 0;
}

static int lcl_prepare_tables(lrm_index *idx, size_t free_b, bool have_sd) {
    const uint64_t L = idx->view.length;
    int hl = 13, pair = 1, ebytes = 8;
    int kbits = 1;
    while ((1ull << kbits) < L) ++kbits;
    static const struct { int hl, ebytes; uint64_t spare, min_rows; } ladder[] = {
        {17, 5, 40ull << 30, 1ull << 32}, {16, 8, 64ull << 30, 0}, {15, 8, 32ull << 30, 0}, {14, 8, 8ull << 30, 0}};
    for (const auto &c : ladder)
        if (L >= c.min_rows && kbits <= 36 && !(have_sd && c.hl > 16) && (uint64_t) free_b >= (2ull * c.ebytes << (2 * c.hl)) + c.spare) { hl = c.hl; ebytes = c.ebytes; break; }
    const LrmIndexTune &tu = idx->itune;
    if (tu.lc_long_max >= 13 && hl > tu.lc_long_max) { hl = tu.lc_long_max; ebytes = 8; }      // the caller expects a short run
    if (tu.lc_long >= 0) { if (tu.lc_long != hl) ebytes = 8; hl = tu.lc_long; }
    if (tu.lc_pair >= 0) pair = tu.lc_pair != 0;
    if (tu.lc_entry_bytes) ebytes = tu.lc_entry_bytes;
    if (!pair || kbits > 36) ebytes = 8;                               // (>= 4 count bits; the plain layout keeps aligned 8-byte entries)
    if (tu.lc_count_bits && 40 - tu.lc_count_bits >= kbits) kbits = 40 - tu.lc_count_bits;       // (tests: few count bits force the side table)
    if (hl <= idx->view.hlen || hl > 17 || L < 2) return 0;
    uint64_t *d = nullptr, *ovf = nullptr, *tab = nullptr;
    unsigned long long *n_ovf = nullptr;
    const uint64_t slots = (pair ? 2ull : 1ull) << (2 * hl);
    const uint64_t ovf_cap = ebytes == 5 ? (slots / 64 < (64ull << 20) ? slots / 64 + 1024 : (64ull << 20)) : 0;
    auto give_up = [&]() { if (d) (void) hipFree(d); if (ovf) (void) hipFree(ovf); if (tab) (void) hipFree(tab); if (n_ovf) (void) hipFree(n_ovf); (void) hipGetLastError(); };
    if (hipMalloc(&d, slots * (uint64_t) ebytes + 16) != hipSuccess) { d = nullptr; give_up(); return 0; }     // no room: the reference's table alone
    if (ebytes == 5 && (hipMalloc(&ovf, ovf_cap * 16) != hipSuccess || hipMalloc(&n_ovf, 8) != hipSuccess || hipMemset(n_ovf, 0, 8) != hipSuccess)) { give_up(); return 0; }
    const uint64_t chunk = 1ull << 22;                                // 2^30 threads per launch (grid limit 2^32)
    for (uint64_t b0 = 0, blocks = slots / 256; b0 < blocks; b0 += chunk) {
        const uint64_t nb = blocks - b0 < chunk ? blocks - b0 : chunk;
        hipLaunchKernelGGL(lcl_build_kernel, dim3((uint32_t) nb), dim3(256), 0, 0, idx->view, hl, pair, ebytes == 5 ? kbits : 0, d, b0 * 256,
                           ovf, ovf_cap, n_ovf);
    }
    if (hipDeviceSynchronize() != hipSuccess) { give_up(); lrm_set_error("long lc table build failed"); return -1; }
    uint64_t mask = 0;
    if (ebytes == 5) {
        unsigned long long n = 0;
        if (hipMemcpy(&n, n_ovf, 8, hipMemcpyDeviceToHost) != hipSuccess || n > ovf_cap) { give_up(); return 0; }   // (too many: the lchash image alone)
        uint64_t tslots = 1024;
        while (tslots < 2 * n) tslots <<= 1;
        mask = tslots - 1;
        if (hipMalloc(&tab, tslots * 16) != hipSuccess || hipMemset(tab, 0, tslots * 16) != hipSuccess) { give_up(); return 0; }
        if (n) hipLaunchKernelGGL(lclx_build_kernel, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, 0, ovf, (uint64_t) n, tab, mask);
        if (hipDeviceSynchronize() != hipSuccess) { give_up(); lrm_set_error("long lc side table build failed"); return -1; }
        (void) hipFree(ovf); (void) hipFree(n_ovf);
    }
    idx->d_lcl = d;
    idx->d_lclx = tab;
    idx->view.lcl = d;
    idx->view.hl = hl;
    idx->view.lcl_pair = pair;
    idx->view.lcl_kbits = ebytes == 5 ? kbits : 0;
    idx->view.lclx = tab;
    idx->view.lclx_mask = mask;
    // Core table on top of pair-line 16-mers with 8-byte entries, for texts small enough that a 13-mer's line holds the
    // 16-mers around it (4 L / 4^13 entries per line on average: 0.55 for an E. coli-sized text, 2 at 2^25 rows).
    const bool core_auto = L <= (1ull << 25);
    if (hl == 16 && pair && ebytes == 8 && (tu.lc_core > 0 || (tu.lc_core < 0 && core_auto))) {
        uint64_t *dc = nullptr, *covf = nullptr;
        unsigned long long *cn = nullptr;
        const uint64_t lines = 1ull << 26, ocap = 16ull << 20;
        bool ok = hipMalloc(&dc, lines * 64) == hipSuccess && hipMalloc(&covf, ocap * 8) == hipSuccess && hipMalloc(&cn, 8) == hipSuccess &&
                  hipMemset(dc, 0, lines * 64) == hipSuccess && hipMemset(cn, 0, 8) == hipSuccess;
        if (ok) {
            for (uint64_t b0 = 0, blocks = (1ull << 32) / 256; b0 < blocks; b0 += chunk) {
                const uint64_t nb = blocks - b0 < chunk ? blocks - b0 : chunk;
                hipLaunchKernelGGL(core_build_kernel, dim3((uint32_t) nb), dim3(256), 0, 0, (const uint64_t *) d, dc, b0 * 256, covf, ocap, cn);
            }
            unsigned long long n = 0;
            ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&n, cn, 8, hipMemcpyDeviceToHost) == hipSuccess && n <= ocap;
            if (ok && n) {
                hipLaunchKernelGGL(core_ovf_kernel, dim3((uint32_t) ((n * 8 + 255) / 256)), dim3(256), 0, 0, dc, (const uint64_t *) covf, (uint64_t) n);
                ok = hipDeviceSynchronize() == hipSuccess;
            }
        }
        if (covf) (void) hipFree(covf);
        if (cn) (void) hipFree(cn);
        if (ok) { idx->d_core = dc; idx->view.core = dc; }
        else { if (dc) (void) hipFree(dc); (void) hipGetLastError(); }      // no room or too many crowded lines: the pair-line table alone
    }
    return 0;
}

struct __attribute__((aligned(8))) WordPair { uint64_t a, b; };

__device__ __forceinline__ uint64_t read_window(const uint64_t *__restrict__ words, uint32_t j) {
    uint32_t wi = j >> 5, sh = (j & 31) * 2;
    WordPair w;                                           // ONE 16-byte request; branch-free, so that the compiler
    __builtin_memcpy(&w, words + wi, sizeof(w));          // cannot split it into a load plus a conditional second load
    return (w.a >> sh) | ((w.b << 1) << (63 - sh));       // sh == 0: the second term shifts out entirely
}

// ----------------------------------------------------------------------------------------
// K1 seed_search: one lane per seed, SS_ITEMS seeds per workgroup (4 or 8 per thread).  Work items of a read are
// (q, iter) with iter fastest, so the 64 lanes of a wavefront hold 64 CONSECUTIVE read positions.
// Output: the SURVIVORS only (0 < rr < thres, ~25 % of the seeds of a noisy read), compact per (read, phase):
//   rec [id][0 .. cnt[id])   k | rr << 40         recq[id][..]  seed ordinal q        (id = read*P + phase)
//   cnt [id]  survivors      hits[id]  sum of rr  (= SA rows the vote will gather; routes the item to its tier)
// A workgroup appends its survivors to per-phase lists in LDS (LDS atomics), reserves room in the global lists with
// ONE global atomic per phase and workgroup, and copies every list segment out with contiguous stores.  (The first
// version stored an 8-byte record per seed POSITION, 75 % zeros, with scattered stores: 16.7 GB written per Gbp,
// and every vote tier re-read all of it.)  The order of a list does not matter: the first-seen order key of a hit,
// (q << tbits) | t, is a property of the hit.
// ----------------------------------------------------------------------------------------
// SS_ITEMS seeds per workgroup (LDS lists of 12 B per seed; LRM_SS_ITEMS=1024|2048|4096 overrides the default 2048).
// Measured per 1-Gbp step [r2]: E. coli-sized text 512 / 1024 / 2048 / 4096 seeds: 26.2 / 25.5 / 24.9 / 31.6 ms;
// GRCh38-sized text 1024 / 2048 / 4096: 44.0 / 40.9 / 42.7 ms.  More resident wavefronts (1024: eight workgroups per
// CU instead of six) do NOT help: the kernel is bound by the memory system's random-request rate, not by latency.
// COUNT: the counting build (lrm_workspace_set_counting; bench bookkeeping, never in a timed region) adds up the
// memory requests the device layout really makes -- seeds evaluated, 8-byte table lookups, 16-byte rank requests --
// into counters->reserved[3..5].
template <int SS_ITEMS, bool COUNT>
__global__ __launch_bounds__(256) void seed_search_kernel(LrmIndexView ix, const uint64_t *__restrict__ reads2,
                                                          uint64_t words_per_read,
                                                          const uint32_t *__restrict__ lens,
                                                          const uint8_t *__restrict__ decided, uint64_t n,
                                                          int seed_len, uint32_t thres, int phase_lo, int phase_hi,
                                                          uint32_t cap_q, uint32_t blocks_per_read,
                                                          uint64_t *__restrict__ rec, uint32_t *__restrict__ recq,
                                                          uint32_t *__restrict__ gcnt, uint32_t *__restrict__ ghits,
                                                          LrmDevCounters *counters) {
    __shared__ uint64_t s_rec[SS_ITEMS + 64];
    __shared__ uint32_t s_traffic[3];
    uint32_t my_cnt[2] = {0, 0}, my_seeds = 0;
    if (COUNT && threadIdx.x < 3) s_traffic[threadIdx.x] = 0;
    __shared__ uint32_t s_q[SS_ITEMS + 64];
    __shared__ uint32_t s_cnt[64], s_hits[64], s_base[64];
    const uint64_t read = blockIdx.x / blocks_per_read;
    const uint32_t chunk = blockIdx.x % blocks_per_read;
    if (read >= n) return;
    if (decided && decided[read]) return;
    const int P = seed_len + 1;
    const uint32_t np = (uint32_t) (phase_hi - phase_lo + 1);
    const uint32_t cap_pp = SS_ITEMS / np + 1;                 // survivors of one phase in one workgroup
    const uint32_t tid = threadIdx.x;
    if (tid < np) { s_cnt[tid] = 0; s_hits[tid] = 0; }
    __syncthreads();
    // iter fastest: neighbouring seeds overlap, so they share their fate (a sequencing error kills ~20 consecutive
    // seeds, a clean stretch lets all of them run the full backward extension): wavefronts diverge little.
    // Measured alternatives that lost: q fastest (+26 %), lane refill from a work chunk (+13 %), packing the
    // survivors of the table lookup into fewer wavefronts (+8 %), and searching 2 / 4 of the lane's seeds TOGETHER
    // (independent chains per lane, all rank gathers of a step in flight at once: 43.4 / 56.7 ms per step against
    // 29.0 [r2]): a wavefront of one-seed lanes stops as soon as its 64 neighbouring seeds are dead, a wavefront of
    // interleaved chains runs until its longest chain ends.  What binds the kernel is the memory system's rate of
    // random requests (56 G 64-byte L2 misses per second on the small text): neither more resident wavefronts (see
    // SS_ITEMS) nor fewer vector instructions (7.1 -> 5.0 G per Gbp with C[] folded into the occ prefixes and the
    // scalar window loads: -2.5 %) move it much.  Also measured [r2]: a
    // 2 MiB presence bitmap of the 12-mers in front of the table lookup (57 % of the seeds die on an L2 hit instead of
    // fetching a table line): 27.3 vs 27.5 ms, not kept; non-temporal loads for the one-touch table lines: 31.7 ms.
    const uint32_t len = lens[read];
    const uint32_t jl = len > (uint32_t) seed_len ? len - (uint32_t) seed_len : 0;   // alnmain.c:353 (fenced for len<s)
    const uint64_t *words = reads2 + read * words_per_read;
    // When (nearly) all phases run in one launch the 64 lanes of a wavefront hold (nearly) consecutive read positions:
    // their 32-base windows lie inside six consecutive words of the packed read, which the wavefront fetches with
    // SCALAR loads (one request per wavefront through the scalar cache instead of 64 lane requests through the
    // texture path) and every lane cuts its window out with selects and a funnel shift.
    const bool dense_lanes = (uint32_t) P - np <= 1u;
    const bool shared_lines = ix.sd && seed_len == ix.sd_len && np == (uint32_t) P && phase_lo == 0;
    LrmIndexView ix_nosd = ix;
    ix_nosd.sd = nullptr;
    if (shared_lines) {
        // All phases in this launch: a lane's item IS its read position j, lane id == j mod sd_f, and the lanes that share a
        // line of the seed table fetch it together (sd_issue_shared): every lane of the wavefront stays in until the line is
        // in registers.
        const uint32_t lim = jl < cap_q * np ? jl : cap_q * np;                                     // positions with a seed
        const uint32_t np_inv = 0xFFFFFFFFu / np;
        const uint64_t smask = (1ull << (2 * seed_len)) - 1ull;
#pragma unroll 1
        for (uint32_t it = 0; it < SS_ITEMS / 256; ++it) {
            const uint32_t j = chunk * SS_ITEMS + it * 256 + tid;
            const bool have = j < lim;
            const uint32_t j0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) (j & ~63u));        // lane 0's position: a multiple of 64
            if (j0 >= lim) break;                                                                  // (positions only grow)
            const uint64_t *wp = words + (j0 >> 5);                                                // wave-uniform address: scalar loads
            const uint64_t W0 = wp[0], W1 = wp[1], W2 = wp[2];
            const uint32_t rel = j - j0, sh = (rel & 31u) * 2u;
            const uint64_t lo = rel < 32u ? W0 : W1, hi = rel < 32u ? W1 : W2;
            const uint64_t code = have ? ((lo >> sh) | ((hi << 1) << (63 - sh))) & smask : 0ull;
            const SdKey key = sd_key_of(ix, code, j & (uint32_t) (ix.sd_f - 1));
            ulonglong2 xa, xb;
            sd_issue_shared(ix, key, xa, xb);
            uint64_t k = 0, l = 0, c = 0, rr;
            const int st = sd_finish_shared(ix, key, code, xa, xb, k, c, COUNT ? my_cnt : nullptr);
            if (!have) continue;
            if (st == 1) rr = c;
            else if (st == 0) rr = 0;
            else rr = seed_one(ix_nosd, code, seed_len, j, k, l, COUNT ? my_cnt : nullptr);           // (a count beyond 24 bits)
            if (COUNT) my_seeds++;
            if (rr > 0 && rr < (uint64_t) thres) {
                uint32_t q = __umulhi(j, np_inv), ph = j - q * np;                                  // j / np, j % np without a division
                if (ph >= np) { ++q; ph -= np; }
                if (ph >= np) { ++q; ph -= np; }
                const uint32_t slot = atomicAdd(&s_cnt[ph], 1u);
                atomicAdd(&s_hits[ph], (uint32_t) rr);
                s_rec[ph * cap_pp + slot] = k | (rr << 40);
                s_q[ph * cap_pp + slot] = q;
            }
        }
    } else {
    const uint32_t np_inv_g = 0xFFFFFFFFu / np;
#pragma unroll 1
    for (uint32_t it = 0; it < SS_ITEMS / 256; ++it) {
        const uint32_t item = chunk * SS_ITEMS + it * 256 + tid;
        uint32_t q = __umulhi(item, np_inv_g), ph = item - q * np;             // item / np, item % np without a division
        if (ph >= np) { ++q; ph -= np; }
        if (ph >= np) { ++q; ph -= np; }
        uint64_t W[6] = {0, 0, 0, 0, 0, 0};
        uint32_t jw = 0;                                                       // first base of W[0]
        if (dense_lanes) {
            const uint32_t item0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) (item & ~63u));   // lane 0's item
            const uint32_t q0 = item0 / np;
            if (q0 < cap_q) {
                const uint32_t j0 = (uint32_t) phase_lo + item0 % np + q0 * (uint32_t) P;
                const uint64_t *wp = words + (j0 >> 5);                        // wave-uniform address
                jw = j0 & ~31u;
#pragma unroll
                for (int e = 0; e < 6; ++e) W[e] = wp[e];
            }
        }
        if (q >= cap_q) break;
        const uint32_t j = (uint32_t) phase_lo + ph + q * (uint32_t) P;        // < 2^32: cap_q * P <= max_len + P
        if (j >= jl) continue;
        uint64_t win, k, l;
        if (dense_lanes) {
            const uint32_t rel = j - jw, wi = rel >> 5, sh = (rel & 31u) * 2u;  // rel <= 31 + 63 + 4: wi in 0..3
            const uint64_t lo = wi == 0 ? W[0] : wi == 1 ? W[1] : wi == 2 ? W[2] : wi == 3 ? W[3] : W[4];
            const uint64_t hi = wi == 0 ? W[1] : wi == 1 ? W[2] : wi == 2 ? W[3] : wi == 3 ? W[4] : W[5];
            win = (lo >> sh) | ((hi << 1) << (63 - sh));
        } else {
            win = read_window(words, j);
        }
        const uint64_t rr = seed_one(ix, win, seed_len, j, k, l, COUNT ? my_cnt : nullptr);
        if (COUNT) my_seeds++;
        if (rr > 0 && rr < (uint64_t) thres) {
            const uint32_t slot = atomicAdd(&s_cnt[ph], 1u);
            atomicAdd(&s_hits[ph], (uint32_t) rr);
            s_rec[ph * cap_pp + slot] = k | (rr << 40);
            s_q[ph * cap_pp + slot] = q;
        }
    }
    }
    if (COUNT) { atomicAdd(&s_traffic[0], my_seeds); atomicAdd(&s_traffic[1], my_cnt[0]); atomicAdd(&s_traffic[2], my_cnt[1]); }
    __syncthreads();
    if (COUNT && tid < 3) atomicAdd(&counters->reserved[3 + tid], (unsigned long long) s_traffic[tid]);
    if (tid < np) {
        const uint32_t c = s_cnt[tid];
        if (c) {
            const uint64_t id = read * (uint64_t) P + (uint64_t) (phase_lo + (int) tid);
            s_base[tid] = atomicAdd(&gcnt[id], c);
            atomicAdd(&ghits[id], s_hits[tid]);
        }
    }
    __syncthreads();
    const uint32_t cpp_inv = 0xFFFFFFFFu / cap_pp;                         // e / cap_pp without a division (e < 2^16: exact after one fix-up)
    for (uint32_t e = tid; e < np * cap_pp; e += 256) {
        uint32_t ph = __umulhi(e, cpp_inv), sl = e - ph * cap_pp;
        if (sl >= cap_pp) { ++ph; sl -= cap_pp; }
        if (sl < s_cnt[ph]) {
            const uint64_t o = (read * (uint64_t) P + (uint64_t) (phase_lo + (int) ph)) * cap_q + s_base[ph] + sl;
            rec[o] = s_rec[e];
            recq[o] = s_q[e];
        }
    }
}

// debug tap: full (j, rr, k, l) per seed of one read, in (iter, q) order
__global__ __launch_bounds__(256) void seed_search_debug_kernel(LrmIndexView ix, const uint64_t *__restrict__ words,
                                                                uint32_t len, int seed_len, uint32_t cap_q,
                                                                int32_t *j_out, uint64_t *rr_out, uint64_t *k_out,
                                                                uint64_t *l_out) {
    const int P = seed_len + 1;
    uint32_t item = blockIdx.x * 256 + threadIdx.x;
    uint32_t iter = item / cap_q, q = item % cap_q;
    if (iter >= (uint32_t) P) return;
    uint32_t jl = len > (uint32_t) seed_len ? len - (uint32_t) seed_len : 0;
    uint64_t j = (uint64_t) iter + (uint64_t) q * (uint64_t) P;
    uint64_t o = (uint64_t) iter * cap_q + q;
    if (j >= jl) { j_out[o] = -1; return; }
    uint64_t win = read_window(words, (uint32_t) j);
    uint64_t k, l;
    uint64_t rr = seed_one(ix, win, seed_len, (uint32_t) j, k, l);
    j_out[o] = (int32_t) j; rr_out[o] = rr; k_out[o] = k; l_out[o] = l;
}

// ----------------------------------------------------------------------------------------
// K2 vote.
// histo_add / histo_find_2_max (histo.c:42-56, 84-96) order entries by insertion; the stable top-2 is "val
// descending, first-seen ascending".  First-seen order of a bucket is the order of (seed ordinal q, SA offset t)
// of its first hit, which is intrinsic to the hit -- so the table can be filled in any order: every slot keeps the
// count, the low 4 bits of the minimum key and the minimum order key (q << tbits) | t.
//
// (Measured and not kept [r2]: a second kernel with a 4096-slot table (70 KB of LDS) for items that need three or more
//  passes -- ultra-long reads, ~3500 hits per item: 45.0 vs 46.1 ms per 2 Gbp, 10.9 vs 11.2 on the bench workload.)
// The vote table always lives in LDS.  ONE kernel votes every (read, phase) item; a 256-thread workgroup owns
// VG consecutive items and routes each by its hit count H (an upper bound on its distinct buckets, left by
// seed_search next to the survivor list):
//   H == 0            the zero result
//   H <= T1_LIMIT     one WAVEFRONT per item (the four wavefronts work on different items), 256-slot table
//   H >  T1_LIMIT     the whole WORKGROUP on one item after the other, T3_SLOTS-slot table, ceil(H / T3_LIMIT)
//                     passes: pass p admits only buckets with hash % passes == p and the per-pass top-2 are merged
//                     (buckets of different passes are disjoint, so the merge is exact)
// Hits are expanded FLAT: the survivors' hit counts are prefix-summed into LDS, and hit h of the item finds its
// seed by a binary search over the prefix -- every lane gathers one SA row per step whatever the shape of the
// item, and all gathers of a step (up to 4 per lane) are in flight before the first vote is cast.  (The first
// version walked repeat seeds two at a time, one memory latency per pair: an item with 24 repeat seeds took 12
// dependent round trips, now 1-2.)
// ----------------------------------------------------------------------------------------
#define VG 16                // items per workgroup (default; LRM_VOTE_VG)
#define VG_MAX 64
#define T1_SLOTS 256
#define T1_LIMIT LRM_VOTE_T1_LIMIT
#define T3_SLOTS LRM_VOTE_T3_SLOTS
#define T3_LIMIT LRM_VOTE_T3_LIMIT
#define T3_CHUNK 256                  // survivors per prefix chunk of the workgroup tier: one per thread
#define EMPTY32 0xFFFFFFFFu

// never a vote key: keys are SA - j (u64 wrap) with SA < 2^40 and j < 2^32, i.e. in [0, 2^40) or [2^64 - 2^32, 2^64)
#define EMPTY_KEY 0x8000000000000000ull

__device__ __forceinline__ uint32_t bucket_hash(uint64_t bucket) {
    uint32_t x = (uint32_t) bucket ^ (uint32_t) (bucket >> 29);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    return x * 0x85EBCA6Bu;
}

// One slot = {min key of the bucket, count, min order key}: the bucket is key >> 4 (histo.c:26-28) and the entry's
// key is the minimum key added to it (histo.c:45-49), so the smallest key IS the slot's identity and its payload.
// Count and order key share one 8-byte word, `count << 32 | ~first` -- the slot's rank in the stable top-2 as it
// stands: the count is a 32-bit atomic add on the high dword, the first-seen order a 32-bit atomic max on the low
// one, and clearing or scanning a slot is one 8-byte LDS access instead of two 4-byte ones.
struct VoteTable {
    uint64_t *key;
    uint64_t *cf;
    uint32_t slots;
};

// Returns false only if the table is full (never in the wavefront tier, where H <= 0.75*slots; in the multi-pass
// tier only under a pathological hash skew) -- the probe loop is bounded so a wave can never spin.
__device__ __forceinline__ bool vote_insert(const VoteTable &t, uint64_t key, uint32_t order, uint32_t hash, uint32_t n = 1u) {
    const uint64_t bucket = key >> 4;
    uint32_t slot = (uint32_t) (((uint64_t) hash * t.slots) >> 32);
    for (uint32_t probe = 0; probe < t.slots; ++probe) {
        // (a plain read before the compare-and-swap, to step over occupied slots cheaply, measured SLOWER: 15.5 vs
        //  13.4 ms per Gbp [r2] -- the extra dependent LDS round trip costs more than the CAS it saves)
        const unsigned long long prev = atomicCAS((unsigned long long *) &t.key[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || (prev >> 4) == bucket) {
            if (prev != EMPTY_KEY && key < prev) atomicMin((unsigned long long *) &t.key[slot], (unsigned long long) key);
            uint32_t *cf = reinterpret_cast<uint32_t *>(&t.cf[slot]);
            atomicAdd(cf + 1, n);                              // count
            atomicMax(cf, 0xFFFFFFFFu - order);                // ~(min order key)
            return true;
        }
        slot = slot + 1 == t.slots ? 0 : slot + 1;
    }
    return false;
}

__device__ __forceinline__ bool vote_admit(const VoteTable &t, uint64_t key, uint32_t order, uint32_t passes, uint32_t pass) {
    const uint32_t hash = bucket_hash(key >> 4);
    if (passes == 1 || (hash >> 16) % passes == pass) return vote_insert(t, key, order, hash);
    return true;
}

struct PhaseTop { uint64_t key1, bucket1, key2, bucket2; uint32_t val1, first1, val2, first2; };

__device__ __forceinline__ void write_phase(LrmPhaseRes *out, const PhaseTop &p) {
    LrmPhaseRes res = {0, 0, 0, 0, 0, 0};
    if (p.val1) { res.key1 = p.key1; res.val1 = p.val1; res.bucket1 = p.bucket1; }
    if (p.val2) { res.key2 = p.key2; res.val2 = p.val2; res.bucket2 = p.bucket2; }
    *out = res;
}

// inclusive prefix sum over the 64 lanes on the DPP network: four row shifts inside the rows of 16, then the
// row totals are broadcast to the following rows (row_bcast:15 / row_bcast:31) -- six v_add_u32_dpp, no LDS
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// # of set bits of a wave mask below this lane
__device__ __forceinline__ uint32_t mask_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}

// Top-2 of a vote table, "count descending, first-seen ascending" (histo.c:84-96 with the insertion order carried
// by the order key): one u64 per slot, count << 32 | ~first, is unique among the filled slots (every hit has its own
// order key), so the stable top-2 is the two largest keys.  Every lane scans its slots, then two max-reductions.
struct Top2 { uint64_t k1, k2; uint32_t s1, s2; };

// max over the 64 lanes, returned in every lane: the prefix-max runs on the DPP network like wave_incl_scan (row
// shifts inside the rows of 16, then row_bcast:15 / row_bcast:31), lane 63 ends up with the maximum and two readlanes
// broadcast it.  (The first version was a butterfly of __shfl_xor: twelve ds_bpermute per reduction, a third of the
// LDS instructions of a small vote item.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_max_step(uint64_t v) {
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) v, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (v >> 32), CTRL, ROW_MASK, 0xf, false);
    const uint64_t o = ((uint64_t) hi << 32) | lo;                 // 0 where the lane has no source: the identity of max
    return o > v ? o : v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
    v = dpp_max_step<0x111, 0xf>(v);      // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);      // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);      // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);      // row_shr:8
    v = dpp_max_step<0x142, 0xa>(v);      // row_bcast:15 -> rows 1, 3
    v = dpp_max_step<0x143, 0xc>(v);      // row_bcast:31 -> rows 2, 3
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, 63);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), 63);
    return ((uint64_t) hi << 32) | lo;
}

template <int NT>
__device__ __forceinline__ Top2 table_top2(const VoteTable &t, uint32_t tid) {
    uint64_t k1 = 0, k2 = 0;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t s = tid; s < t.slots; s += NT) {
        const uint64_t k = t.cf[s];
        if (k > k1) { k2 = k1; s2 = s1; k1 = k; s1 = s; }
        else if (k > k2) { k2 = k; s2 = s; }
    }
    // wave-level: the largest key, then the largest of what is left
    const uint64_t m1 = wave_max_u64(k1);
    const bool win = k1 == m1 && (m1 >> 32) != 0;
    const uint64_t m2 = wave_max_u64(win ? k2 : k1);
    Top2 r;
    r.k1 = (m1 >> 32) ? m1 : 0; r.k2 = (m2 >> 32) ? m2 : 0; r.s1 = 0; r.s2 = 0;
    if (r.k1) {
        const unsigned long long b = __ballot(k1 == m1);
        r.s1 = (uint32_t) __builtin_amdgcn_readlane((int) s1, (int) __builtin_ctzll(b));
    }
    if (r.k2) {
        const bool has = (win ? k2 : k1) == m2;
        const unsigned long long b = __ballot(has);
        const int src = (int) __builtin_ctzll(b);
        r.s2 = (uint32_t) __builtin_amdgcn_readlane((int) (win ? s2 : s1), src);
    }
    return r;
}

// the same over a table in GLOBAL memory (the one-pass path of very large items): the counts were written by atomics
// through L2, and a slice of the pool is reused by later items, so the scan reads past the L1 (agent-scope loads)
__device__ __forceinline__ Top2 table_top2_global(const VoteTable &t, uint32_t tid) {
    uint64_t k1 = 0, k2 = 0;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t s = tid; s < t.slots; s += 256) {
        const uint64_t k = __hip_atomic_load(&t.cf[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k > k1) { k2 = k1; s2 = s1; k1 = k; s1 = s; }
        else if (k > k2) { k2 = k; s2 = s; }
    }
    const uint64_t m1 = wave_max_u64(k1);
    const bool win = k1 == m1 && (m1 >> 32) != 0;
    const uint64_t m2 = wave_max_u64(win ? k2 : k1);
    Top2 r;
    r.k1 = (m1 >> 32) ? m1 : 0; r.k2 = (m2 >> 32) ? m2 : 0; r.s1 = 0; r.s2 = 0;
    if (r.k1) {
        const unsigned long long b = __ballot(k1 == m1);
        r.s1 = (uint32_t) __builtin_amdgcn_readlane((int) s1, (int) __builtin_ctzll(b));
    }
    if (r.k2) {
        const bool has = (win ? k2 : k1) == m2;
        const unsigned long long b = __ballot(has);
        const int src = (int) __builtin_ctzll(b);
        r.s2 = (uint32_t) __builtin_amdgcn_readlane((int) (win ? s2 : s1), src);
    }
    return r;
}

// survivor s of the hit h: off[s] <= h < off[s + 1]  (off: exclusive prefix of the staged survivors' hit counts,
// strictly increasing because every survivor has at least one hit; cnt >= 1)
// (Measured alternative for the wavefront tier [r2]: a marker byte where the hits of each staged seed begin + a DPP
//  prefix maximum over the 64 consecutive hits of the lanes, i.e. one LDS read instead of seven dependent ones:
//  9.80 vs 9.76 ms per Gbp -- the search is not what the tier waits for.)
__device__ __forceinline__ uint32_t find_seed(const uint32_t *off, uint32_t cnt, uint32_t h) {
    uint32_t lo = 0, hi = cnt;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= h) lo = mid; else hi = mid;
    }
    return lo;
}

// Survivors come in two kinds.  UNIQUE seeds (rr == 1: the read's true locus, ~3/4 of the survivors of a noisy
// read) are voted by the lane that loaded them: one SA gather, no staging.  REPEAT seeds (rr > 1) are compacted
// into LDS with the prefix sums of their hit counts and their hits are expanded flat: hit h finds its seed by a
// binary search over the (few) staged repeat seeds.
// The hits [0, total) of the staged survivors (off / srec / sq), voted by NT threads (tid of NT).
template <int NT, int VOTE_U>               // VOTE_U: SA gathers in flight per lane
__device__ __forceinline__ bool vote_hits(const LrmIndexView &ix, const VoteTable &t, const uint32_t *off,
                                          const uint64_t *srec, const uint32_t *sq, uint32_t cnt, uint32_t total,
                                          uint32_t iter, uint32_t P, uint32_t tbits, uint32_t tid, uint32_t passes,
                                          uint32_t pass, uint64_t *kc_key = nullptr, uint32_t *kc_ord = nullptr) {
    bool ok = true;
    for (uint32_t hb = 0; hb < total; hb += NT * VOTE_U) {
        uint64_t v[VOTE_U];
        uint32_t ss[VOTE_U], tt[VOTE_U];
#pragma unroll
        for (int u = 0; u < VOTE_U; ++u) {
            const uint32_t h = hb + (uint32_t) u * NT + tid;
            v[u] = 0; ss[u] = 0; tt[u] = 0;
            if (h < total) {
                const uint32_t s = find_seed(off, cnt, h);
                ss[u] = s;
                tt[u] = h - off[s];
                v[u] = sa_locate(ix, (srec[s] & ((1ull << 40) - 1ull)) + tt[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < VOTE_U; ++u) {
            const uint32_t h = hb + (uint32_t) u * NT + tid;
            if (h < total) {
                const uint32_t q = sq[ss[u]];
                const uint64_t key = v[u] - (uint64_t) (iter + q * P);              // alnmain.c:363-365 (u64 wrap kept); j < 2^32
                if (kc_key) { kc_key[h] = key; kc_ord[h] = (q << tbits) | tt[u]; }   // multi-pass items: keys kept for the later passes
                ok &= vote_admit(t, key, (q << tbits) | tt[u], passes, pass);
            }
        }
    }
    return ok;
}

// ---- wavefront tier: H <= T1_LIMIT, so at most T1_LIMIT survivors -------------------------------------------
struct WaveLds {
    uint64_t key[T1_SLOTS];
    uint64_t srec[T1_LIMIT / 2];             // repeat seeds have >= 2 hits each
    uint64_t cf[T1_SLOTS];
    uint32_t off[T1_LIMIT / 2 + 4];
    uint32_t sq[T1_LIMIT / 2];
};

template <int VOTE_U>
__device__ __forceinline__ void vote_item_wave(const LrmIndexView &ix, const uint64_t *__restrict__ rec,
                                               const uint32_t *__restrict__ recq, uint32_t cnt, uint32_t H,
                                               uint32_t iter, uint32_t P, uint32_t tbits, int lane, WaveLds &L,
                                               LrmPhaseRes *out, uint32_t load) {
    VoteTable t = {L.key, L.cf, 0};
    {   // clear / scan only as much of the table as this item can fill (<= 75 % load)
        const uint32_t eff = H * 100u / load + 64;
        t.slots = eff < (uint32_t) T1_SLOTS ? eff : (uint32_t) T1_SLOTS;
    }
    constexpr int NU = (T1_LIMIT + 63) / 64;
    uint64_t e[NU], sv[NU];
    uint32_t qq[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                                   // survivor loads first, table clear behind them
        const uint32_t s = (uint32_t) u * 64 + (uint32_t) lane;
        e[u] = s < cnt ? rec[s] : 0ull;
        qq[u] = s < cnt ? recq[s] : 0u;
    }
    for (uint32_t s = lane; s < t.slots; s += 64) { t.key[s] = EMPTY_KEY; t.cf[s] = 0; }
    uint32_t run = 0, nbig = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const uint32_t rr = (uint32_t) (e[u] >> 40);
        sv[u] = rr == 1 ? sa_of_unique(ix, e[u]) : 0ull;       // unique seeds: gather at once (or nothing to gather)
        const bool big = rr > 1;
        const unsigned long long bm = __ballot(big);
        const uint32_t incl = wave_incl_scan(big ? rr : 0u);
        if (big) {
            const uint32_t idx = nbig + mask_rank(bm);
            L.off[idx] = run + incl - rr; L.srec[idx] = e[u]; L.sq[idx] = qq[u];
        }
        run += (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
        nbig += (uint32_t) __popcll(bm);
    }
    if (lane == 0) L.off[nbig] = run;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (peeling the most frequent buckets of a batch off with ballots + wave reductions, so that one lane adds a whole
    //  group of equal votes, measured SLOWER: 18.8 vs 13.4 ms per Gbp [r2] -- same-slot contention is not the cost)
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if ((uint32_t) (e[u] >> 40) == 1) vote_admit(t, sv[u] - (uint64_t) (iter + qq[u] * P), qq[u] << tbits, 1u, 0u);
    if (nbig) vote_hits<64, VOTE_U>(ix, t, L.off, L.srec, L.sq, nbig, run, iter, P, tbits, (uint32_t) lane, 1u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const Top2 w = table_top2<64>(t, (uint32_t) lane);
    if (lane == 0) {
        PhaseTop p = {};
        if (w.k1) { p.val1 = (uint32_t) (w.k1 >> 32); p.key1 = t.key[w.s1]; p.bucket1 = p.key1 >> 4; }
        if (w.k2) { p.val2 = (uint32_t) (w.k2 >> 32); p.key2 = t.key[w.s2]; p.bucket2 = p.key2 >> 4; }
        write_phase(out, p);
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- workgroup tier -------------------------------------------------------------------------------------------
struct BlockLds {
    uint64_t key[T3_SLOTS];
    uint64_t srec[T3_CHUNK];
    uint64_t cf[T3_SLOTS];
    uint32_t off[T3_CHUNK + 4];
    uint32_t sq[T3_CHUNK];
};
union VoteLds { WaveLds w[4]; BlockLds b; };

template <int VOTE_U>
__device__ __forceinline__ void vote_item_block(const LrmIndexView &ix, const uint64_t *__restrict__ rec,
                                                const uint32_t *__restrict__ recq, uint32_t cnt, uint32_t H,
                                                uint32_t iter, uint32_t P, uint32_t tbits, uint32_t slots, uint32_t limit,
                                                BlockLds &L, uint32_t *s_wsum, Top2 *s_top, LrmPhaseRes *out,
                                                uint32_t *err_word, uint32_t load, uint64_t *kc_key, uint32_t *kc_ord,
                                                uint32_t kc_cap, uint64_t *gtab, uint32_t *glock, uint32_t g_slices,
                                                uint32_t g_slots, uint32_t *s_slice) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const int lane = (int) (tid & 63);
    VoteTable t = {L.key, L.cf, slots};
    uint32_t passes = (H + limit - 1) / limit;
    // Items with more hits than the key scratch holds (a read made of a 100-299-copy repeat family: up to
    // cap_q * (thres - 1) hits per phase, 142 k for a 10 kbp read) would take H / 768 passes of H gathers each over
    // the 1024-slot LDS table -- quadratic, minutes for a batch of such reads.  They vote in ONE pass into a table in
    // global memory instead: the workgroup takes one of a few slices of a pool (a spin on a lock word: holders never
    // wait for anyone, so it always comes free), clears 2^k >= 2 H slots, inserts with the same compare-and-swap
    // protocol through L2, scans, gives the slice back.
    const bool big = gtab != nullptr && H > kc_cap && 2ull * H <= (uint64_t) g_slots;
    if (big) {
        if (tid == 0) {
            uint32_t got = 0xFFFFFFFFu;
            for (uint32_t spin = 0; got == 0xFFFFFFFFu; ++spin) {
                const uint32_t sl = (blockIdx.x + spin) % g_slices;
                if (atomicCAS(&glock[sl], 0u, 1u) == 0u) got = sl;
                else __builtin_amdgcn_s_sleep(32);
            }
            *s_slice = got;
        }
        __syncthreads();
        uint32_t gs = 1024;
        while (gs < 2 * H) gs <<= 1;
        uint64_t *base = gtab + (uint64_t) (*s_slice) * 2ull * g_slots;
        t.key = base; t.cf = base + g_slots; t.slots = gs;
        passes = 1;
    }
    // Items that need several passes (ultra-long reads: ~3500 hits, five passes): the first pass writes every hit's
    // {key, order key} to this workgroup's slice of a global scratch, and the later passes stream them back
    // (12 coalesced bytes per hit) instead of searching, gathering and subtracting again.
    // (from three passes on: with two, writing and re-reading 12 B per hit costs as much traffic as it saves)
    const bool cache = passes > 2 && H <= kc_cap;
    if (!big) {
        const uint32_t per_pass = passes > 1 ? limit : H;
        const uint32_t eff = per_pass * 100u / load + 64;
        t.slots = eff < slots ? eff : slots;
    }
    PhaseTop best = {};
    for (uint32_t pass = 0; pass < passes; ++pass) {
        for (uint32_t s = tid; s < t.slots; s += 256) { t.key[s] = EMPTY_KEY; t.cf[s] = 0; }
        if (big) { __threadfence(); __syncthreads(); }             // the cleared slots are in L2 before the first atomic of another thread
        bool ok = true;
        if (cache && pass > 0) {
            __syncthreads();                                   // table cleared
            for (uint32_t i = tid; i < H; i += 256) ok &= vote_admit(t, kc_key[i], kc_ord[i], passes, pass);
        } else {
        uint32_t kbase = 0;                                    // hits of the chunks before this one
        for (uint32_t c0 = 0; c0 < cnt; c0 += T3_CHUNK) {
            const uint32_t nc = cnt - c0 < (uint32_t) T3_CHUNK ? cnt - c0 : (uint32_t) T3_CHUNK;
            // one survivor per thread; unique seeds gather at once, repeat seeds are compacted into LDS
            const uint64_t e0 = tid < nc ? rec[c0 + tid] : 0ull;
            const uint32_t q0 = tid < nc ? recq[c0 + tid] : 0u;
            const uint32_t r0 = (uint32_t) (e0 >> 40);
            const uint64_t v0 = r0 == 1 ? sa_of_unique(ix, e0) : 0ull;
            const uint32_t b0 = r0 > 1 ? 1u : 0u, h0 = b0 ? r0 : 0u;
            const unsigned long long bm = __ballot(b0 != 0), um = __ballot(r0 == 1);
            const uint32_t incl_h = wave_incl_scan(h0);
            __syncthreads();                                   // the previous chunk's (or pass's) staging is no longer read
            if (lane == 63) { s_wsum[wave] = incl_h; s_wsum[4 + wave] = (uint32_t) __popcll(bm); s_wsum[8 + wave] = (uint32_t) __popcll(um); }
            __syncthreads();
            uint32_t woff_h = 0, total = 0, woff_n = 0, nbig = 0, woff_u = 0, nuni = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t x = s_wsum[w], y = s_wsum[4 + w], z = s_wsum[8 + w];
                total += x; nbig += y; nuni += z;
                if (w < wave) { woff_h += x; woff_n += y; woff_u += z; }
            }
            if (b0) {
                const uint32_t idx = woff_n + mask_rank(bm);
                L.off[idx] = woff_h + incl_h - h0; L.srec[idx] = e0; L.sq[idx] = q0;
            }
            if (tid == 0) L.off[nbig] = total;
            __syncthreads();
            if (r0 == 1) {
                const uint64_t key = v0 - (uint64_t) (iter + q0 * P);
                if (cache) { const uint32_t i = kbase + woff_u + mask_rank(um); kc_key[i] = key; kc_ord[i] = q0 << tbits; }
                ok &= vote_admit(t, key, q0 << tbits, passes, pass);
            }
            if (nbig) ok &= vote_hits<256, VOTE_U>(ix, t, L.off, L.srec, L.sq, nbig, total, iter, P, tbits, tid, passes, pass,
                                                   cache ? kc_key + kbase + nuni : nullptr, cache ? kc_ord + kbase + nuni : nullptr);
            kbase += nuni + total;
        }
        if (cache) __threadfence_block();                      // the scratch is read back by other threads of the workgroup
        }
        if (!ok) *(volatile uint32_t *) err_word = LRM_ERR_VOTE_OVERFLOW;   // host-coherent, sticky
        if (big) __threadfence();
        __syncthreads();
        const Top2 w = big ? table_top2_global(t, tid) : table_top2<256>(t, tid);                       // this wavefront's share of the table
        if (lane == 0) s_top[wave] = w;
        __syncthreads();
        if (tid == 0) {
            // the pass's top-2 = the two largest of the four wavefronts' pairs; merged into the running top-2 of
            // the earlier passes (disjoint bucket sets).  Keys compare as (count, first-seen) pairs.
            uint64_t ck[2] = {0, 0};
            uint32_t cslot[2] = {0, 0};
            for (int x = 0; x < 4; ++x) {
                const Top2 c = s_top[x];
                const uint64_t ks[2] = {c.k1, c.k2};
                const uint32_t ss[2] = {c.s1, c.s2};
                for (int y = 0; y < 2; ++y) {
                    if (ks[y] > ck[0]) { ck[1] = ck[0]; cslot[1] = cslot[0]; ck[0] = ks[y]; cslot[0] = ss[y]; }
                    else if (ks[y] > ck[1]) { ck[1] = ks[y]; cslot[1] = ss[y]; }
                }
            }
            uint64_t r1k = best.val1 ? ((uint64_t) best.val1 << 32) | (0xFFFFFFFFu - best.first1) : 0;
            uint64_t r2k = best.val2 ? ((uint64_t) best.val2 << 32) | (0xFFFFFFFFu - best.first2) : 0;
            PhaseTop nb = best;
            for (int x = 0; x < 2; ++x) {
                const uint64_t c = ck[x];
                if (!c) continue;
                const uint32_t cv = (uint32_t) (c >> 32), cf = 0xFFFFFFFFu - (uint32_t) c;
                const uint64_t ky = big ? __hip_atomic_load(&t.key[cslot[x]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t.key[cslot[x]], bk = ky >> 4;
                if (c > r1k) {
                    nb.key2 = nb.key1; nb.bucket2 = nb.bucket1; nb.val2 = nb.val1; nb.first2 = nb.first1; r2k = r1k;
                    nb.key1 = ky; nb.bucket1 = bk; nb.val1 = cv; nb.first1 = cf; r1k = c;
                } else if (c > r2k) {
                    nb.key2 = ky; nb.bucket2 = bk; nb.val2 = cv; nb.first2 = cf; r2k = c;
                }
            }
            best = nb;
        }
        __syncthreads();
    }
    if (tid == 0) write_phase(out, best);
    if (big) {                                                     // the slice goes back to the pool
        __syncthreads();
        if (tid == 0) { __threadfence(); atomicExch(&glock[*s_slice], 0u); }
    }
}

// ---- fast path: one wavefront per item, repeat-only buckets never enter the table ------------------------------------------
// On a text with interspersed repeats most hits of an item come from a few REPEAT seeds (rr up to thres - 1 hits each)
// and land in buckets of their own, one or two votes each: on the bench workload 85 % of all hits, and what pushes an
// item from the 256-slot wavefront table into the workgroup tier and its passes.  They cannot win.  The vote's output is
// the top entry and the COUNT of the second (alnmain.c:374-388 reads cand[0] and cand[1].val only), so:
//   A  the hits of the UNIQUE seeds (rr == 1; at most one per survivor) are inserted as before -- table T;
//   B  a hit of a repeat seed is looked up in T with plain reads: present -> counted (count, min key, first-seen
//      order: exactly what an insert would have done); absent -> it belongs to a bucket made of repeat hits only, and
//      only a 16-bit counter of a small SKETCH (indexed by a hash of the bucket) is incremented -- no compare-and-swap,
//      no probe chain, no table space;
//   C  with t2 = the second-highest count in T and M = the largest sketch counter (>= the count of every repeat-only
//      bucket): if M < t2 no repeat-only bucket reaches the top two, and the top two of T are the item's result, bit for
//      bit.  Otherwise (few true hits, or a read made of repeats) the item goes on a list for the exact kernel above.
// The table only ever holds buckets of unique seeds (<= survivors <= T1_LIMIT), so an item needs one pass whatever its
// hit count, and a wavefront stages its repeat seeds 64 survivors at a time.  Items with more than T1_LIMIT survivors go
// to the exact kernel as well.
#define FAST_SK_WORDS 512                    // 1024 16-bit counters per wavefront
#define FB_LIMIT 1536                        // survivors up to which the workgroup form takes an item (75 % of its 2048 slots)
struct FastLds {
    uint64_t key[T1_SLOTS];
    uint64_t cf[T1_SLOTS];
    uint64_t srec[64];
    uint32_t off[64 + 4];
    uint32_t sq[64];
    uint32_t sketch[FAST_SK_WORDS];
};

// B: returns true if the bucket is in the table (and has been counted)
__device__ __forceinline__ bool vote_count_if_present(const VoteTable &t, uint64_t key, uint32_t order, uint32_t hash) {
    const uint64_t bucket = key >> 4;
    uint32_t slot = (uint32_t) (((uint64_t) hash * t.slots) >> 32);
    for (uint32_t probe = 0; probe < t.slots; ++probe) {
        const uint64_t prev = t.key[slot];
        if (prev == EMPTY_KEY) return false;                              // (no deletions: an empty slot ends the chain)
        if ((prev >> 4) == bucket) {
            if (key < prev) atomicMin((unsigned long long *) &t.key[slot], (unsigned long long) key);
            uint32_t *cf = reinterpret_cast<uint32_t *>(&t.cf[slot]);
            atomicAdd(cf + 1, 1u);
            atomicMax(cf, 0xFFFFFFFFu - order);
            return true;
        }
        slot = slot + 1 == t.slots ? 0 : slot + 1;
    }
    return false;
}

template <int NT, int VOTE_U>
__device__ __forceinline__ void fast_hits(const LrmIndexView &ix, const VoteTable &t, const uint32_t *off, const uint64_t *srec,
                                          const uint32_t *sq, uint32_t *sketch, uint32_t sk_mask, uint32_t cnt, uint32_t total,
                                          uint32_t iter, uint32_t P, uint32_t tbits, uint32_t lane) {
    for (uint32_t hb = 0; hb < total; hb += NT * VOTE_U) {
        uint64_t v[VOTE_U];
        uint32_t ss[VOTE_U], tt[VOTE_U];
#pragma unroll
        for (int u = 0; u < VOTE_U; ++u) {
            const uint32_t h = hb + (uint32_t) u * NT + lane;
            v[u] = 0; ss[u] = 0; tt[u] = 0;
            if (h < total) {
                const uint32_t s = find_seed(off, cnt, h);
                ss[u] = s;
                tt[u] = h - off[s];
                v[u] = sa_locate(ix, (srec[s] & ((1ull << 40) - 1ull)) + tt[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < VOTE_U; ++u) {
            const uint32_t h = hb + (uint32_t) u * NT + lane;
            if (h < total) {
                const uint32_t q = sq[ss[u]];
                const uint64_t key = v[u] - (uint64_t) (iter + q * P);              // alnmain.c:363-365 (u64 wrap kept)
                const uint32_t hash = bucket_hash(key >> 4);
                if (!vote_count_if_present(t, key, (q << tbits) | tt[u], hash)) {
                    const uint32_t c = (hash >> 5) & sk_mask;
                    atomicAdd(&sketch[c >> 1], 1u << (16 * (c & 1)));              // < 2^16 hits per bucket: 16 per seed at most
                }
            }
        }
    }
}

#define FAST_CH 16                           // items per ticket of a wavefront
template <int VOTE_U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
void vote_fast_kernel(LrmIndexView ix, const uint64_t *__restrict__ rec, const uint32_t *__restrict__ recq,
                      const uint32_t *__restrict__ gcnt, const uint32_t *__restrict__ ghits,
                      const uint8_t *__restrict__ decided, uint64_t n, int seed_len, int phase_lo, int phase_hi,
                      uint32_t cap_q, uint32_t tbits, uint32_t load, unsigned long long *ticket,
                      LrmPhaseRes *__restrict__ phase_res, uint64_t *__restrict__ redo, unsigned long long *redo_n,
                      uint64_t *__restrict__ big, unsigned long long *big_n) {
    __shared__ FastLds lds[4];
    const uint32_t lane = threadIdx.x & 63u;
    FastLds &L = lds[threadIdx.x >> 6];
    const uint32_t P = (uint32_t) seed_len + 1;
    const uint32_t np = (uint32_t) (phase_hi - phase_lo + 1);
    const uint64_t n_items = n * (uint64_t) np;
    constexpr int NU = (T1_LIMIT + 63) / 64;
    for (;;) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(ticket, (unsigned long long) FAST_CH);
        base = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (base >> 32)) << 32) |
               (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) base);
        if (base >= n_items) break;
        uint64_t read = base / np;
        uint32_t pi = (uint32_t) (base - read * np);
        for (uint32_t it = 0; it < FAST_CH && base + it < n_items; ++it, ++pi) {
            if (pi == np) { pi = 0; ++read; }
            const uint64_t item = base + it;
            const uint32_t iter = (uint32_t) phase_lo + pi;
            const uint64_t id = read * (uint64_t) P + iter;
            if (decided && decided[read]) continue;
            const uint32_t H = ghits[id], cnt = gcnt[id];
            if (H == 0) {
                if (lane == 0) { LrmPhaseRes z = {0, 0, 0, 0, 0, 0}; phase_res[id] = z; }
                continue;
            }
            if (cnt > (uint32_t) T1_LIMIT) {                                   // more survivors than the wavefront table is sized for:
                if (lane == 0) {                                               // the workgroup form of this kernel, or the exact kernel
                    if (cnt <= (uint32_t) FB_LIMIT) big[atomicAdd(big_n, 1ull)] = item;
                    else redo[atomicAdd(redo_n, 1ull)] = item;
                }
                continue;
            }
            VoteTable t = {L.key, L.cf, 0};
            {
                const uint32_t eff = cnt * 100u / load + 64;
                t.slots = eff < (uint32_t) T1_SLOTS ? eff : (uint32_t) T1_SLOTS;
            }
            const uint64_t *irec = rec + id * cap_q;
            const uint32_t *iq = recq + id * cap_q;
            uint64_t e[NU], sv[NU];
            uint32_t qq[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {                                   // survivor loads first, clears behind them
                const uint32_t s = (uint32_t) u * 64 + lane;
                e[u] = s < cnt ? irec[s] : 0ull;
                qq[u] = s < cnt ? iq[s] : 0u;
            }
            for (uint32_t s = lane; s < t.slots; s += 64) { t.key[s] = EMPTY_KEY; t.cf[s] = 0; }
#pragma unroll
            for (uint32_t s = 0; s < FAST_SK_WORDS / 64; ++s) L.sketch[s * 64 + lane] = 0;
            unsigned long long any_big = 0;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const uint32_t rr = (uint32_t) (e[u] >> 40);
                sv[u] = rr == 1 ? sa_of_unique(ix, e[u]) : 0ull;       // unique seeds: gather at once (or nothing to gather)
                any_big |= __ballot(rr > 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // A: the unique seeds' hits make the table
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if ((uint32_t) (e[u] >> 40) == 1) vote_admit(t, sv[u] - (uint64_t) (iter + qq[u] * P), qq[u] << tbits, 1u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // B: the repeat seeds' hits, 64 survivors at a time
            if (any_big) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const uint32_t rr = (uint32_t) (e[u] >> 40);
                    const bool big = rr > 1;
                    const unsigned long long bm = __ballot(big);
                    if (bm == 0) continue;
                    const uint32_t incl = wave_incl_scan(big ? rr : 0u);
                    if (big) {
                        const uint32_t idx = mask_rank(bm);
                        L.off[idx] = incl - rr; L.srec[idx] = e[u]; L.sq[idx] = qq[u];
                    }
                    const uint32_t nb = (uint32_t) __popcll(bm);
                    const uint32_t run = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
                    if (lane == 0) L.off[nb] = run;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    fast_hits<64, VOTE_U>(ix, t, L.off, L.srec, L.sq, L.sketch, 2 * FAST_SK_WORDS - 1, nb, run, iter, P, tbits, lane);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // C
            const Top2 w = table_top2<64>(t, lane);
            uint32_t m = 0;
#pragma unroll
            for (uint32_t s = 0; s < FAST_SK_WORDS / 64; ++s) {
                const uint32_t x = L.sketch[s * 64 + lane];
                const uint32_t a = x & 0xffffu, b = x >> 16;
                m = a > m ? a : m;
                m = b > m ? b : m;
            }
            const uint32_t M = (uint32_t) wave_max_u64((uint64_t) m);
            const bool settled = any_big == 0 || M < (uint32_t) (w.k2 >> 32);
            if (lane == 0) {
                if (settled) {
                    PhaseTop p = {};
                    if (w.k1) { p.val1 = (uint32_t) (w.k1 >> 32); p.key1 = t.key[w.s1]; p.bucket1 = p.key1 >> 4; }
                    if (w.k2) { p.val2 = (uint32_t) (w.k2 >> 32); p.key2 = t.key[w.s2]; p.bucket2 = p.key2 >> 4; }
                    write_phase(&phase_res[id], p);
                } else {
                    redo[atomicAdd(redo_n, 1ull)] = item;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same for items with up to FB_LIMIT survivors (reads of 100 kbp: ~1200 per item), one WORKGROUP per item: a
// 2048-slot table for the unique seeds' buckets, a 4096-counter sketch, repeat seeds staged 256 survivors at a time.
// One pass whatever the hit count (the exact kernel takes ceil(hits / 768) passes over its 1024-slot table: five on
// such reads).  Works through the list the wavefront kernel leaves (`big`).
#define FB_SLOTS 2048
#define FB_SK_WORDS 2048
struct FastBlockLds {
    uint64_t key[FB_SLOTS];
    uint64_t cf[FB_SLOTS];
    uint64_t srec[T3_CHUNK];
    uint32_t off[T3_CHUNK + 4];
    uint32_t sq[T3_CHUNK];
    uint32_t sketch[FB_SK_WORDS];
};
template <int VOTE_U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void vote_fast_block_kernel(LrmIndexView ix, const uint64_t *__restrict__ rec, const uint32_t *__restrict__ recq,
                            const uint32_t *__restrict__ gcnt, const uint32_t *__restrict__ ghits, int seed_len,
                            int phase_lo, int phase_hi, uint32_t cap_q, uint32_t tbits, uint32_t load,
                            unsigned long long *ticket, LrmPhaseRes *__restrict__ phase_res,
                            const uint64_t *__restrict__ big, const unsigned long long *__restrict__ big_n,
                            uint64_t *__restrict__ redo, unsigned long long *redo_n) {
    __shared__ FastBlockLds L;
    __shared__ uint32_t s_wsum[8], s_m[4];
    __shared__ Top2 s_top[4];
    __shared__ unsigned long long s_at;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t P = (uint32_t) seed_len + 1;
    const uint32_t np = (uint32_t) (phase_hi - phase_lo + 1);
    const uint64_t n_big = (uint64_t) *big_n;
    for (;;) {
        if (tid == 0) s_at = atomicAdd(ticket, 1ull);
        __syncthreads();
        const uint64_t at = s_at;
        if (at >= n_big) break;
        const uint64_t item = big[at];
        const uint64_t read = item / np;
        const uint32_t iter = (uint32_t) phase_lo + (uint32_t) (item - read * np);
        const uint64_t id = read * (uint64_t) P + iter;
        const uint32_t cnt = gcnt[id];
        const uint64_t *irec = rec + id * cap_q;
        const uint32_t *iq = recq + id * cap_q;
        VoteTable t = {L.key, L.cf, 0};
        {
            const uint32_t eff = cnt * 100u / load + 64;
            t.slots = eff < (uint32_t) FB_SLOTS ? eff : (uint32_t) FB_SLOTS;
        }
        for (uint32_t s = tid; s < t.slots; s += 256) { t.key[s] = EMPTY_KEY; t.cf[s] = 0; }
        for (uint32_t s = tid; s < FB_SK_WORDS; s += 256) L.sketch[s] = 0;
        __syncthreads();
        // A: the unique seeds' hits make the table
        bool ok = true;
        uint32_t any_big = 0;
        for (uint32_t c0 = 0; c0 < cnt; c0 += 256) {
            const uint64_t e0 = c0 + tid < cnt ? irec[c0 + tid] : 0ull;
            const uint32_t r0 = (uint32_t) (e0 >> 40);
            any_big |= r0 > 1 ? 1u : 0u;
            if (r0 == 1) {
                const uint32_t q0 = iq[c0 + tid];
                const uint64_t v0 = sa_of_unique(ix, e0);
                ok &= vote_admit(t, v0 - (uint64_t) (iter + q0 * P), q0 << tbits, 1u, 0u);
            }
        }
        const bool block_big = __syncthreads_or((int) any_big) != 0;
        // B: the repeat seeds' hits, 256 survivors at a time
        if (block_big) {
            for (uint32_t c0 = 0; c0 < cnt; c0 += 256) {
                const uint64_t e0 = c0 + tid < cnt ? irec[c0 + tid] : 0ull;
                const uint32_t q0 = c0 + tid < cnt ? iq[c0 + tid] : 0u;
                const uint32_t r0 = (uint32_t) (e0 >> 40);
                const bool b0 = r0 > 1;
                const unsigned long long bm = __ballot(b0);
                const uint32_t incl_h = wave_incl_scan(b0 ? r0 : 0u);
                if (lane == 63) { s_wsum[wave] = incl_h; s_wsum[4 + wave] = (uint32_t) __popcll(bm); }
                __syncthreads();
                uint32_t woff_h = 0, total = 0, woff_n = 0, nbig = 0;
#pragma unroll
                for (uint32_t w = 0; w < 4; ++w) {
                    const uint32_t x = s_wsum[w], y = s_wsum[4 + w];
                    total += x; nbig += y;
                    if (w < wave) { woff_h += x; woff_n += y; }
                }
                if (b0) {
                    const uint32_t idx = woff_n + mask_rank(bm);
                    L.off[idx] = woff_h + incl_h - r0; L.srec[idx] = e0; L.sq[idx] = q0;
                }
                if (tid == 0) L.off[nbig] = total;
                __syncthreads();
                if (nbig) fast_hits<256, VOTE_U>(ix, t, L.off, L.srec, L.sq, L.sketch, 2 * FB_SK_WORDS - 1, nbig, total, iter, P, tbits, tid);
                __syncthreads();                                   // the staging is rewritten by the next chunk
            }
        }
        // C
        const Top2 w = table_top2<256>(t, tid);
        uint32_t m = 0;
        for (uint32_t s = tid; s < FB_SK_WORDS; s += 256) {
            const uint32_t x = L.sketch[s];
            const uint32_t a = x & 0xffffu, b = x >> 16;
            m = a > m ? a : m;
            m = b > m ? b : m;
        }
        m = (uint32_t) wave_max_u64((uint64_t) m);
        if (lane == 0) { s_top[wave] = w; s_m[wave] = m; }
        const bool all_ok = __syncthreads_and((int) ok) != 0;
        if (tid == 0) {
            uint64_t ck[2] = {0, 0};
            uint32_t cslot[2] = {0, 0}, M = 0;
            for (int x = 0; x < 4; ++x) {
                const Top2 c = s_top[x];
                const uint64_t ks[2] = {c.k1, c.k2};
                const uint32_t ss[2] = {c.s1, c.s2};
                for (int y = 0; y < 2; ++y) {
                    if (ks[y] > ck[0]) { ck[1] = ck[0]; cslot[1] = cslot[0]; ck[0] = ks[y]; cslot[0] = ss[y]; }
                    else if (ks[y] > ck[1]) { ck[1] = ks[y]; cslot[1] = ss[y]; }
                }
                M = s_m[x] > M ? s_m[x] : M;
            }
            const bool settled = all_ok && (!block_big || M < (uint32_t) (ck[1] >> 32));
            if (settled) {
                PhaseTop p = {};
                if (ck[0]) { p.val1 = (uint32_t) (ck[0] >> 32); p.key1 = t.key[cslot[0]]; p.bucket1 = p.key1 >> 4; }
                if (ck[1]) { p.val2 = (uint32_t) (ck[1] >> 32); p.key2 = t.key[cslot[1]]; p.bucket2 = p.key2 >> 4; }
                write_phase(&phase_res[id], p);
            } else {
                redo[atomicAdd(redo_n, 1ull)] = item;
            }
        }
        __syncthreads();                                           // s_at, the table and s_top are rewritten by the next item
    }
}

#ifndef LRM_VOTE_WAVES_PER_EU
#define LRM_VOTE_WAVES_PER_EU 6     // 79 VGPRs and 23.7 KB of LDS per workgroup: six workgroups per CU
#endif
template <int VOTE_U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LRM_VOTE_WAVES_PER_EU, LRM_VOTE_WAVES_PER_EU)))
void vote_kernel(LrmIndexView ix, const uint64_t *__restrict__ rec,
                                                   const uint32_t *__restrict__ recq,
                                                   const uint32_t *__restrict__ gcnt,
                                                   const uint32_t *__restrict__ ghits,
                                                   const uint8_t *__restrict__ decided, uint64_t n, int seed_len,
                                                   int phase_lo, int phase_hi, uint32_t cap_q, uint32_t tbits,
                                                   uint32_t slots3, uint32_t limit3, uint32_t vg, uint32_t limit1, uint32_t load,
                                                   unsigned long long *ticket, uint64_t *__restrict__ kc_key_all,
                                                   uint32_t *__restrict__ kc_ord_all, uint32_t kc_cap,
                                                   LrmPhaseRes *__restrict__ phase_res, uint32_t *err_word,
                                                   const uint64_t *__restrict__ list, const unsigned long long *__restrict__ list_n,
                                                   uint64_t *gtab, uint32_t *glock, uint32_t g_slices, uint32_t g_slots) {
    __shared__ VoteLds lds;
    __shared__ uint32_t g_H[VG_MAX], g_cnt[VG_MAX], g_ph[VG_MAX];
    __shared__ uint64_t g_id[VG_MAX];
    __shared__ uint32_t s_wsum[12];
    __shared__ Top2 s_top[4];
    __shared__ unsigned long long s_grp;
    __shared__ uint32_t s_slice;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const int lane = (int) (tid & 63);
    const uint32_t P = (uint32_t) seed_len + 1;
    const uint32_t np = (uint32_t) (phase_hi - phase_lo + 1);
    // list mode: the items the fast kernel could not settle (vote_fast_kernel), by their item numbers
    const uint64_t n_items = list ? (uint64_t) *list_n : n * (uint64_t) np;
    const uint64_t n_groups = (n_items + vg - 1) / vg;
    uint64_t *kc_key = kc_key_all + (uint64_t) blockIdx.x * kc_cap;      // this workgroup's slice of the key scratch
    uint32_t *kc_ord = kc_ord_all + (uint64_t) blockIdx.x * kc_cap;
    // A fixed grid of resident workgroups takes groups of items from a ticket counter (dynamic balance, and one
    // scratch slice per workgroup); every workgroup ends with a ticket beyond the last group.
    for (;;) {
    if (tid == 0) s_grp = atomicAdd(ticket, 1ull);
    __syncthreads();
    const uint64_t grp = s_grp;
    if (grp >= n_groups) break;
    if (tid < vg) {
        const uint64_t li = grp * vg + tid;
        const uint64_t item = list && li < n_items ? list[li] : li;
        uint32_t H = 0, c = 0, ph = 0;
        uint64_t id = 0;
        if (li < n_items) {
            const uint64_t read = item / np;
            ph = (uint32_t) phase_lo + (uint32_t) (item - read * np);        // the item's phase (kept: a 64-bit modulo per item and wavefront is ~150 instructions)
            id = read * (uint64_t) P + (uint64_t) ph;
            if (!(decided && decided[read])) {
                H = ghits[id];
                c = gcnt[id];
                if (H == 0) { LrmPhaseRes z = {0, 0, 0, 0, 0, 0}; phase_res[id] = z; }
            }
        }
        g_H[tid] = H; g_cnt[tid] = c; g_id[tid] = id; g_ph[tid] = ph;
    }
    __syncthreads();
    for (uint32_t g = wave; g < vg; g += 4) {                 // wavefront tier: four items at a time
        const uint32_t H = g_H[g];
        if (H == 0 || H > limit1) continue;
        const uint64_t id = g_id[g];
        vote_item_wave<VOTE_U>(ix, rec + id * cap_q, recq + id * cap_q, g_cnt[g], H, g_ph[g], P, tbits, lane,
                       lds.w[wave], &phase_res[id], load);
    }
    __syncthreads();
    for (uint32_t g = 0; g < vg; ++g) {                       // workgroup tier: one item after the other
        const uint32_t H = g_H[g];
        if (H <= limit1) continue;
        const uint64_t id = g_id[g];
        vote_item_block<VOTE_U>(ix, rec + id * cap_q, recq + id * cap_q, g_cnt[g], H, g_ph[g], P, tbits, slots3,
                        limit3, lds.b, s_wsum, s_top, &phase_res[id], err_word, load, kc_key, kc_ord, kc_cap, gtab, glock, g_slices,
                        g_slots, &s_slice);
        __syncthreads();
    }
    __syncthreads();                                          // s_grp, g_* are rewritten by the next round
    }
}

// ----------------------------------------------------------------------------------------
// decide: replay of alnmain.c:371-403 over the per-phase vote results.
//   mode 0 : phase 0 only -- mark reads whose phase-0 vote passes (they are final)
//   mode 1 : all phases, for reads not marked in mode 0
//   mode 2 : all phases for every read (single-round launches); counts the reads mode 0 would have marked
// (double)v/num_seeds > 0.6  <=>  5v > 3*num_seeds for every feasible size.
// ----------------------------------------------------------------------------------------
#define MAX_PHASES 40
__global__ __launch_bounds__(256) void decide_kernel(const LrmPhaseRes *__restrict__ phase_res,
                                                     const uint32_t *__restrict__ lens, uint64_t n,
                                                     int seed_len, int mode, uint8_t *__restrict__ decided,
                                                     lrm_entry *__restrict__ best, LrmDevCounters *counters) {
    uint64_t read = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (read >= n) return;
    const int P = seed_len + 1;
    const uint64_t num_seeds = lens[read] / (uint32_t) P;          // alnmain.c:371
    const LrmPhaseRes *pr = phase_res + read * (uint64_t) P;
    if (mode == 0) {
        uint8_t d = 0;
        if (num_seeds > 0 && P > 1) {       // a break on the LAST phase is undone (alnmain.c:400): P==1 never final here
            LrmPhaseRes r0 = pr[0];
            uint64_t v = r0.val1 + r0.val2;
            if (5 * v > 3 * num_seeds) {
                d = 1;
                lrm_entry e = {r0.key1, r0.val1, r0.bucket1};
                best[read] = e;
                atomicAdd(&counters->decided_phase0, 1ull);
            }
        }
        decided[read] = d;
        return;
    }
    if (mode == 1 && decided[read]) return;
    // ot_iter_histo (alnmain.c:340,386-388,400-403): at most P entries
    uint64_t ot_key[MAX_PHASES], ot_val[MAX_PHASES], ot_bucket[MAX_PHASES];
    int ot_n = 0;
    lrm_entry out = {0, 0, 0};
    int iter;
    for (iter = 0; iter < P; ++iter) {
        if (num_seeds > 0) {
            LrmPhaseRes r = pr[iter];
            uint64_t v = r.val1 + r.val2;
            if (5 * v > 3 * num_seeds) {
                out.key = r.key1; out.val = r.val1; out.bucket = r.bucket1;
                if (mode == 2 && iter == 0 && P > 1) atomicAdd(&counters->decided_phase0, 1ull);
                break;
            } else if (r.val1 != 0) {
                uint64_t key = r.key1, bucket = key >> 4;
                bool found = false;
                for (int t = 0; t < ot_n; ++t) {
                    if (ot_bucket[t] == bucket) {
                        found = true;
                        ot_val[t] += 1;
                        if (key < ot_key[t]) ot_key[t] = key;
                    }
                }
                if (!found) { ot_key[ot_n] = key; ot_val[ot_n] = 1; ot_bucket[ot_n] = bucket; ot_n++; }
            }
        }
    }
    if (iter >= P - 1) {       // ran out, or broke on the last phase: winner comes from ot_iter_histo
        lrm_entry t1 = {0, 0, 0};
        for (int t = 0; t < ot_n; ++t)
            if (t1.val < ot_val[t]) { t1.key = ot_key[t]; t1.val = ot_val[t]; t1.bucket = ot_bucket[t]; }
        out = t1;
    }
    best[read] = out;
}

// ----------------------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    lrm_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

int lrm_launch_seed(lrm_index *idx, lrm_workspace *ws, const char *d_reads, uint64_t stride,
                    const uint32_t *d_lens, uint64_t n, uint32_t max_len, uint32_t seed_len,
                    uint32_t thres, lrm_entry *d_best, const LrmMapTune &mt, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    if (n == 0) return 0;
    const int P = (int) seed_len + 1;
    const uint32_t cap_q = ws->cap_q;
    const uint64_t wpr = ws->words_per_read;
    (void) max_len; (void) thres;

    HIPCHK(hipMemsetAsync(ws->d_counters, 0, sizeof(LrmDevCounters), stream));
    HIPCHK(hipMemsetAsync(ws->d_hcount, 0, n * (uint64_t) P * 4, stream));
    HIPCHK(hipMemsetAsync(ws->d_cnt, 0, n * (uint64_t) P * 4, stream));
    ws->n_last = n;
    {
        uint64_t bpr = wpr * 8;
        uint32_t cpr = (uint32_t) ((bpr / 4 + 255) / 256);
        uint64_t blocks = n * cpr;
        if (blocks > 0x7fffffffull) { lrm_set_error("pack2bit grid too large"); return -1; }
        lrm_time_begin(ws, LRM_K_PACK2BIT, stream);
        hipLaunchKernelGGL(pack2bit_kernel, dim3((uint32_t) blocks), dim3(256), 0, stream, d_reads, stride,
                           d_lens, (uint8_t *) ws->d_reads2, bpr, cpr, n);
        lrm_time_end(ws, stream);
    }
    uint32_t tbits = 1;
    while ((1u << tbits) < thres && tbits < 31) tbits++;
    if (((uint64_t) cap_q << tbits) > 0xffffffffull) {
        lrm_set_error("read too long for the vote order key: cap_q %u << %u bits exceeds 32 bits", cap_q, tbits);
        return -1;
    }
    // (tests force overflows of the multi-pass tier with a pass limit above the table size and a small table:
    //  lrm_debug_set_vote_limits)
    const uint32_t t3_limit = mt.t3_limit ? mt.t3_limit : (uint32_t) T3_LIMIT;
    const uint32_t t3_slots = mt.t3_slots >= 8 && mt.t3_slots <= T3_SLOTS ? mt.t3_slots : (uint32_t) T3_SLOTS;
    // tuning knobs (measured defaults; tools/seed_probe.py sweeps them through the environment, read at handle creation)
    const uint32_t vg = mt.vote_vg >= 1 && mt.vote_vg <= VG_MAX ? mt.vote_vg : VG;
    const uint32_t t1_limit = mt.vote_t1 <= T1_LIMIT ? mt.vote_t1 : (uint32_t) T1_LIMIT;
    const int vote_u = (int) mt.vote_u;
    const uint32_t vote_load = mt.vote_load;        // percent of the table slots an item is sized for (when the table allows): at 75 % the
                                                    // linear probes of the slowest lane cost +1.7 ms per Gbp [r2], at 90 % +4.4 ms
    // Rounds.  Phase 0 alone first, then phases 1..s for the reads it did not decide, saves 20/21 of the work on clean
    // reads; on noisy reads phase 0 decides nothing and the split only costs a second set of launches whose phase-0
    // wavefronts hold seeds 21 positions apart (no shared fate).  The workspace remembers how many reads the previous
    // batch decided in phase 0 (copied back asynchronously, never waited for): below 2 % the next batch runs ALL phases
    // in one round.  Speculative evaluation is exact, so the results do not depend on the choice.  lrm_map_options.seed_rounds.
    bool single = false;
    {
        const volatile uint64_t *hist = reinterpret_cast<const volatile uint64_t *>(ws->h_err + 2);
        const uint64_t d0 = *hist;
        if (ws->hist_n >= 64 && d0 * 50 < ws->hist_n) single = true;
        if (mt.seed_rounds == 1 || mt.seed_rounds == 2) single = mt.seed_rounds == 1;
        if (P == 1) single = true;
    }
    for (int round = single ? 1 : 0; round < 2; ++round) {
        int lo = round == 0 || single ? 0 : 1;
        int hi = round == 0 ? 0 : P - 1;
        if (lo > hi) break;
        int np = hi - lo + 1;
        const uint8_t *dec = round == 0 || single ? nullptr : ws->d_decided;
        const uint32_t ss_items = mt.ss_items;
        uint32_t bpr = (uint32_t) (((uint64_t) np * cap_q + ss_items - 1) / ss_items);
        uint64_t blocks = n * bpr;
        if (blocks > 0x7fffffffull) { lrm_set_error("seed_search grid too large: split the batch"); return -1; }
        lrm_time_begin(ws, LRM_K_SEED_SEARCH, stream);
        auto sk = ws->counting ? seed_search_kernel<2048, true>
                               : ss_items == 1024u ? seed_search_kernel<1024, false> : ss_items == 4096u ? seed_search_kernel<4096, false> : seed_search_kernel<2048, false>;
        if (ws->counting) bpr = (uint32_t) (((uint64_t) np * cap_q + 2047) / 2048);
        if (ws->counting) blocks = n * bpr;
        // (mt.ss_lds_pad: extra dynamic LDS per workgroup, i.e. fewer resident workgroups per CU -- see DESIGN 5)
        hipLaunchKernelGGL(sk, dim3((uint32_t) blocks), dim3(256), mt.ss_lds_pad, stream, idx->view,
                           ws->d_reads2, wpr, d_lens, dec, n, (int) seed_len, thres, lo, hi, cap_q, bpr,
                           ws->d_rec, ws->d_recq, ws->d_cnt, ws->d_hcount, ws->d_counters);
        lrm_time_end(ws, stream);
        uint64_t items = n * (uint64_t) np;
        uint64_t vblocks = (items + vg - 1) / vg;
        if (vblocks > LRM_VOTE_GRID) vblocks = LRM_VOTE_GRID;          // resident workgroups; groups of items go by ticket
        lrm_time_begin(ws, LRM_K_VOTE, stream);
        auto vk = vote_u == 2 ? vote_kernel<2> : vote_u == 8 ? vote_kernel<8> : vote_kernel<4>;
        uint64_t *big_tab = mt.t3_limit ? nullptr : ws->d_gtab;      // (the overflow-forcing test knobs keep the LDS passes)
        if (mt.vote_fast) {
            // fast kernel over all items, then the exact kernel over the items it could not settle (its list)
            uint64_t fblocks = (items + 4 * FAST_CH - 1) / (4 * FAST_CH);
            if (fblocks > LRM_VOTE_FAST_GRID) fblocks = LRM_VOTE_FAST_GRID;
            auto fk = vote_u == 2 ? vote_fast_kernel<2> : vote_u == 8 ? vote_fast_kernel<8> : vote_fast_kernel<4>;
            hipLaunchKernelGGL(fk, dim3((uint32_t) fblocks), dim3(256), 0, stream, idx->view, ws->d_rec, ws->d_recq, ws->d_cnt,
                               ws->d_hcount, dec, n, (int) seed_len, lo, hi, cap_q, tbits, vote_load,
                               &ws->d_counters->vote_fast_ticket[round], ws->d_phase, ws->d_redo, &ws->d_counters->vote_redo_n[round],
                               ws->d_big, &ws->d_counters->vote_big_n[round]);
            auto fbk = vote_u == 2 ? vote_fast_block_kernel<2> : vote_u == 8 ? vote_fast_block_kernel<8> : vote_fast_block_kernel<4>;
            uint64_t bblocks = items < 768 ? items : 768;                     // three workgroups per CU
            hipLaunchKernelGGL(fbk, dim3((uint32_t) bblocks), dim3(256), 0, stream, idx->view, ws->d_rec, ws->d_recq, ws->d_cnt,
                               ws->d_hcount, (int) seed_len, lo, hi, cap_q, tbits, vote_load, &ws->d_counters->vote_big_ticket[round],
                               ws->d_phase, (const uint64_t *) ws->d_big, (const unsigned long long *) &ws->d_counters->vote_big_n[round],
                               ws->d_redo, &ws->d_counters->vote_redo_n[round]);
            hipLaunchKernelGGL(vk, dim3((uint32_t) vblocks), dim3(256), 0, stream, idx->view, ws->d_rec, ws->d_recq,
                               ws->d_cnt, ws->d_hcount, dec, n, (int) seed_len, lo, hi, cap_q, tbits, t3_slots, t3_limit,
                               vg, t1_limit, vote_load, &ws->d_counters->reserved[1 + round], ws->d_kc_key, ws->d_kc_ord,
                               (uint32_t) LRM_VOTE_KC_CAP, ws->d_phase, ws->d_err, (const uint64_t *) ws->d_redo,
                               (const unsigned long long *) &ws->d_counters->vote_redo_n[round], big_tab, ws->d_glock, ws->g_slices, ws->g_slots);
        } else {
            hipLaunchKernelGGL(vk, dim3((uint32_t) vblocks), dim3(256), 0, stream, idx->view, ws->d_rec, ws->d_recq,
                               ws->d_cnt, ws->d_hcount, dec, n, (int) seed_len, lo, hi, cap_q, tbits, t3_slots, t3_limit,
                               vg, t1_limit, vote_load, &ws->d_counters->reserved[1 + round], ws->d_kc_key, ws->d_kc_ord,
                               (uint32_t) LRM_VOTE_KC_CAP, ws->d_phase, ws->d_err, (const uint64_t *) nullptr,
                               (const unsigned long long *) nullptr, big_tab, ws->d_glock, ws->g_slices, ws->g_slots);
        }
        lrm_time_end(ws, stream);
        lrm_time_begin(ws, LRM_K_DECIDE, stream);
        hipLaunchKernelGGL(decide_kernel, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, stream, ws->d_phase,
                           d_lens, n, (int) seed_len, single ? 2 : round, ws->d_decided, d_best, ws->d_counters);
        lrm_time_end(ws, stream);
    }
    HIPCHK(hipMemcpyAsync((void *) (ws->h_err + 2), &ws->d_counters->decided_phase0, 8, hipMemcpyDeviceToHost, stream));
    ws->hist_n = n;
    HIPCHK(hipGetLastError());
    return 0;
}

int lrm_launch_debug_seed(lrm_index *idx, const char *d_read, uint32_t len, uint32_t seed_len,
                          uint64_t *d_reads2, uint64_t words, int32_t *d_j, uint64_t *d_rr,
                          uint64_t *d_k, uint64_t *d_l, uint64_t cap, void *stream_) {
    hipStream_t stream = (hipStream_t) stream_;
    const uint32_t P = seed_len + 1;
    uint32_t jl = len > seed_len ? len - seed_len : 0;
    uint32_t cap_q = (jl + P - 1) / P;
    if (cap_q == 0) cap_q = 1;
    if ((uint64_t) cap_q * P > cap) { lrm_set_error("debug seed buffer too small"); return -1; }
    uint64_t bpr = words * 8;
    uint32_t cpr = (uint32_t) ((bpr / 4 + 255) / 256);
    uint32_t *d_len1 = (uint32_t *) (d_reads2 + words);     // caller reserves one extra word for the length
    HIPCHK(hipMemcpyAsync(d_len1, &len, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(pack2bit_kernel, dim3(cpr), dim3(256), 0, stream, d_read, (uint64_t) 0, d_len1,
                       (uint8_t *) d_reads2, bpr, cpr, (uint64_t) 1);
    uint32_t items = cap_q * P;
    hipLaunchKernelGGL(seed_search_debug_kernel, dim3((items + 255) / 256), dim3(256), 0, stream, idx->view,
                       d_reads2, len, (int) seed_len, cap_q, d_j, d_rr, d_k, d_l);
    HIPCHK(hipGetLastError());
    return (int) cap_q;
}
