// index_host.cpp -- CPU index construction + on-disk formats (include/lrm_index_host.h).
// Index construction stays on the CPU (BASELINE north_star); this is the product's own
// builder: an SA-IS suffix sorter instead of the vendored pSAscan/libdivsufsort, and the
// lchash table derived in one pass over the suffix array instead of 4^hlen backward searches.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include <omp.h>
#include "../../include/lrm_index_host.h"
#include "lrm_internal.h"

// ------------------------------------------------------------------------------------------
// SA-IS (Nong, Zhang, Chan: "Two efficient algorithms for linear time suffix array
// construction").  s[n-1] must be the unique smallest symbol.
// ------------------------------------------------------------------------------------------
namespace {

// LRM_BUILD_VERBOSE=1: stage times of the index builder on stderr
struct StageTimer {
    double t0; bool on;
    StageTimer() : t0(omp_get_wtime()), on(getenv("LRM_BUILD_VERBOSE") != nullptr) {}
    void lap(const char *what) { if (on) { const double t = omp_get_wtime(); fprintf(stderr, "[lrm build] %-28s %8.2f s\n", what, t - t0); t0 = t; } }
};

template <typename I>
struct SaIs {
    static inline bool tget(const uint8_t *t, I i) { return (t[i >> 3] >> (i & 7)) & 1; }
    static inline void tset(uint8_t *t, I i, bool b) {
        if (b) t[i >> 3] |= (uint8_t) (1u << (i & 7)); else t[i >> 3] &= (uint8_t) ~(1u << (i & 7));
    }
    static inline bool is_lms(const uint8_t *t, I i) { return i > 0 && tget(t, i) && !tget(t, i - 1); }

    template <typename C>
    static void buckets(const C *s, I *bkt, I n, I K, bool end) {
        for (I i = 0; i <= K; ++i) bkt[i] = 0;
        for (I i = 0; i < n; ++i) bkt[(I) s[i]]++;
        I sum = 0;
        for (I i = 0; i <= K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
    }
    template <typename C>
    static void induce_l(const uint8_t *t, I *SA, const C *s, I *bkt, I n, I K) {
        buckets(s, bkt, n, K, false);
        for (I i = 0; i < n; ++i) {
            I j = SA[i] - 1;
            if (SA[i] > 0 && !tget(t, j)) SA[bkt[(I) s[j]]++] = j;
        }
    }
    template <typename C>
    static void induce_s(const uint8_t *t, I *SA, const C *s, I *bkt, I n, I K) {
        buckets(s, bkt, n, K, true);
        for (I i = n - 1; i >= 0; --i) {
            I j = SA[i] - 1;
            if (SA[i] > 0 && tget(t, j)) SA[--bkt[(I) s[j]]] = j;
        }
    }

    template <typename C>
    static void run(const C *s, I *SA, I n, I K) {
        if (n == 1) { SA[0] = 0; return; }
        std::vector<uint8_t> tv((size_t) n / 8 + 1, 0);
        uint8_t *t = tv.data();
        tset(t, n - 2, false);
        tset(t, n - 1, true);
        for (I i = n - 3; i >= 0; --i)
            tset(t, i, s[i] < s[i + 1] || (s[i] == s[i + 1] && tget(t, i + 1)));
        std::vector<I> bv((size_t) K + 1);
        I *bkt = bv.data();
        // stage 1: sort the LMS substrings
        buckets(s, bkt, n, K, true);
        for (I i = 0; i < n; ++i) SA[i] = -1;
        for (I i = 1; i < n; ++i)
            if (is_lms(t, i)) SA[--bkt[(I) s[i]]] = i;
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
        I n1 = 0;
        for (I i = 0; i < n; ++i)
            if (is_lms(t, SA[i])) SA[n1++] = SA[i];
        for (I i = n1; i < n; ++i) SA[i] = -1;
        I name = 0, prev = -1;
        for (I i = 0; i < n1; ++i) {
            I pos = SA[i];
            bool diff = false;
            for (I d = 0; d < n; ++d) {
                if (prev == -1 || s[pos + d] != s[prev + d] || tget(t, pos + d) != tget(t, prev + d)) { diff = true; break; }
                else if (d > 0 && (is_lms(t, pos + d) || is_lms(t, prev + d))) break;
            }
            if (diff) { name++; prev = pos; }
            SA[n1 + pos / 2] = name - 1;
        }
        for (I i = n - 1, j = n - 1; i >= n1; --i)
            if (SA[i] >= 0) SA[j--] = SA[i];
        // stage 2: solve the reduced problem
        I *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) run<I>(s1, SA1, n1, name - 1);
        else for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
        // stage 3: induce the final order
        buckets(s, bkt, n, K, true);
        for (I i = 1, j = 0; i < n; ++i)
            if (is_lms(t, i)) s1[j++] = i;
        for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
        for (I i = n1; i < n; ++i) SA[i] = -1;
        for (I i = n1 - 1; i >= 0; --i) {
            I j = SA[i];
            SA[i] = -1;
            SA[--bkt[(I) s[j]]] = j;
        }
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
    }
};

inline int dna_code(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return -1; }
}

inline uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Parallel suffix sorter for nucleotide texts (the builder GRCh38-sized references need: 6.2 G rows in
// minutes on the host cores next to the GPU; the reference uses the parallel external-memory pSAscan,
// psascan/sa_use.cc:8-18, asindex.c:138).  The suffix array of a text that ends in a unique minimal '$' is
// unique, so any correct sorter reproduces the reference's .sa5 byte for byte.
//
//   1. the text is packed to 2 bits per base, first base most significant, so that the integer order of a
//      64-bit window is the lexicographic order of 32 bases; positions past the last base read as 'A' (0)
//   2. suffixes are distributed by their first PB bases into 4^PB buckets (histogram, prefix sums)
//   3. the buckets are processed in groups that fit a bounded scratch: one parallel scan of the text collects
//      {next 32 bases, position} of every suffix of the group, each bucket is sorted by that key in cache,
//      and only runs of equal keys (40+ common bases) are compared through the packed text
//   4. the sorted positions go straight into the caller's ui40 array (8-byte slots, padding zeroed)
// A suffix that runs into '$' compares as if padded with 'A' and loses ties to longer suffixes, which is the
// order '$' < 'A' gives.  Texts with very long exact repeats would make step 3 quadratic: the work spent on
// ties is counted and the build falls back to the linear-time SA-IS above when it exceeds a budget.
// ------------------------------------------------------------------------------------------
namespace {

struct PackedText {
    std::vector<uint64_t> w;      // 32 bases per word, first base in bits 63..62
    uint64_t n = 0;               // bases (text length without '$')
    inline uint64_t window(uint64_t p) const {             // bases p .. p+31, zero padded
        const uint64_t i = p >> 5, sh = (p & 31) * 2;
        const uint64_t a = w[i], b = w[i + 1];
        return sh ? ((a << sh) | (b >> (64 - sh))) : a;
    }
    inline uint64_t roll(uint64_t win, uint64_t p) const {  // window(p) -> window(p + 1)
        const uint64_t q = p + 32;
        return (win << 2) | ((w[q >> 5] >> (62 - 2 * (q & 31))) & 3ull);
    }
};

struct KeyPos { uint64_t key, pos; };

// exact order of the suffixes at p and q, known to share their first d bases (padded semantics, see above)
static inline bool suffix_less(const PackedText &t, uint64_t p, uint64_t q, uint64_t d, uint64_t &work) {
    while (true) {
        const int64_t lp = (int64_t) t.n - (int64_t) (p + d), lq = (int64_t) t.n - (int64_t) (q + d);
        if (lp <= 0 || lq <= 0) return p > q;                        // the one that has reached '$' is smaller
        const uint64_t wp = t.window(p + d), wq = t.window(q + d);
        ++work;
        if (wp != wq) {
            const int64_t first = __builtin_clzll(wp ^ wq) >> 1, lmin = lp < lq ? lp : lq;
            if (first >= lmin) return p > q;                          // equal up to the shorter one's '$'
            return wp < wq;
        }
        d += 32;
    }
}

template <typename F>
static void sort_ties(KeyPos *a, uint64_t n, F less) {
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && a[j].key == a[i].key) ++j;
        if (j - i > 1) std::sort(a + i, a + j, less);
        i = j;
    }
}

// 2-bit image of text[0 .. L-1) (the '$' at L-1 is not part of it); false if the text holds a byte other than
// upper-case ACGT (such texts take the generic paths)
static bool pack_text(const char *text, uint64_t L, PackedText &t) {
    const uint64_t n = L - 1;
    t.n = n;
    t.w.assign(n / 32 + 4, 0);
    int bad = 0;
    const uint64_t nwords = (n + 31) / 32;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static) reduction(| : bad)
    for (uint64_t wi = 0; wi < nwords; ++wi) {
        uint64_t v = 0;
        const uint64_t lo = wi * 32, hi = lo + 32 < n ? lo + 32 : n;
        for (uint64_t i = lo; i < hi; ++i) {
            const int c = dna_code(text[i]);
            if (c < 0 || text[i] > 'Z') bad = 1;
            v |= (uint64_t) (c & 3) << (62 - 2 * (i - lo));
        }
        t.w[wi] = v;
    }
    return !bad;
}

// top-bits distribution + std::sort of the pieces: ~3x fewer comparisons than std::sort alone on a 100 k-row bucket
static void sort_by_key(KeyPos *a, uint64_t m, std::vector<KeyPos> &tmp) {
    auto by_key = [](const KeyPos &x, const KeyPos &y) { return x.key < y.key; };
    if (m < 2048) { std::sort(a, a + m, by_key); return; }
    constexpr int RB = 11;
    uint32_t cnt[(1u << RB) + 1] = {0};
    for (uint64_t i = 0; i < m; ++i) cnt[(a[i].key >> (64 - RB)) + 1]++;
    for (uint32_t b = 0; b < (1u << RB); ++b) cnt[b + 1] += cnt[b];
    if (tmp.size() < m) tmp.resize(m);
    uint32_t cur[1u << RB];
    memcpy(cur, cnt, sizeof(cur));
    for (uint64_t i = 0; i < m; ++i) tmp[cur[a[i].key >> (64 - RB)]++] = a[i];
    memcpy(a, tmp.data(), m * sizeof(KeyPos));
    for (uint32_t b = 0; b < (1u << RB); ++b)
        if (cnt[b + 1] - cnt[b] > 1) std::sort(a + cnt[b], a + cnt[b + 1], by_key);
}

static int sa_build_bucketed(const PackedText &t, uint64_t L, lrm_ui40 *out, uint64_t tie_budget_per_row) {
    const uint64_t n = L - 1;
    StageTimer tm;

    const int PB = L > (1ull << 26) ? 8 : (L > (1ull << 18) ? 5 : 2);     // bases of the distribution prefix
    const uint64_t NB = 1ull << (2 * PB);
    const int nth = lrm_host_threads();        // the team size of the region below (the caller's OpenMP limit may differ)
    // histogram of prefixes (per-thread, merged)
    std::vector<uint64_t> cnt(NB + 1, 0);
    {
        std::vector<std::vector<uint64_t>> local((size_t) nth, std::vector<uint64_t>(NB, 0));
#pragma omp parallel num_threads(nth)
        {
            std::vector<uint64_t> &h = local[(size_t) omp_get_thread_num()];
#pragma omp for schedule(static)
            for (uint64_t blk = 0; blk < (n + 65535) / 65536; ++blk) {
                const uint64_t lo = blk * 65536, hi = lo + 65536 < n ? lo + 65536 : n;
                uint64_t win = t.window(lo);
                for (uint64_t p = lo; p < hi; ++p) {
                    h[win >> (64 - 2 * PB)]++;
                    win = t.roll(win, p);
                }
            }
        }
        for (int th = 0; th < nth; ++th) for (uint64_t bkt = 0; bkt < NB; ++bkt) cnt[bkt + 1] += local[(size_t) th][bkt];
    }
    for (uint64_t bkt = 0; bkt < NB; ++bkt) cnt[bkt + 1] += cnt[bkt];      // cnt[b] = suffixes in buckets < b
    tm.lap("sa: prefix histogram");
    uint64_t *sa64 = reinterpret_cast<uint64_t *>(out);                    // a ui40 slot is 8 bytes: value < 2^40, padding 0
    sa64[0] = n;                                                           // the suffix "$"

    // groups of consecutive buckets within the scratch bound
    uint64_t scratch_rows = (L / 6) + (1ull << 20);
    if (const char *e = getenv("LRM_SA_SCRATCH_ROWS")) { const long long v = atoll(e); if (v >= 1024) scratch_rows = (uint64_t) v; }   // test knob
    uint64_t max_bucket = 0;
    for (uint64_t bkt = 0; bkt < NB; ++bkt) max_bucket = std::max(max_bucket, cnt[bkt + 1] - cnt[bkt]);
    if (scratch_rows < max_bucket) scratch_rows = max_bucket;
    KeyPos *scratch = (KeyPos *) malloc(scratch_rows * sizeof(KeyPos));
    if (!scratch) { lrm_set_error("out of memory (suffix sorter scratch, %llu rows)", (unsigned long long) scratch_rows); return -1; }
    std::vector<uint64_t> cursor(NB);
    uint64_t work_total = 0;
    const uint64_t budget = tie_budget_per_row * L + (1ull << 24);
    bool over = false;
    for (uint64_t g0 = 0; g0 < NB && !over;) {
        uint64_t g1 = g0 + 1;
        while (g1 < NB && cnt[g1 + 1] - cnt[g0] <= scratch_rows) ++g1;
        const uint64_t base = cnt[g0], rows = cnt[g1] - cnt[g0];
        if (rows == 0) { g0 = g1; continue; }
        for (uint64_t bkt = g0; bkt < g1; ++bkt) cursor[bkt] = cnt[bkt] - base;
        // collect {key, position} of the group's suffixes: threads reserve space in small batches
#pragma omp parallel num_threads(lrm_host_threads())
        {
            constexpr int LB = 8;
            std::vector<KeyPos> lbuf((size_t) (g1 - g0) * LB);
            std::vector<uint8_t> lcnt((size_t) (g1 - g0), 0);
            auto flush = [&](uint64_t bkt) {
                const uint64_t k = bkt - g0;
                const uint8_t m = lcnt[k];
                if (!m) return;
                uint64_t at;
#pragma omp atomic capture
                { at = cursor[bkt]; cursor[bkt] += m; }
                memcpy(scratch + at, &lbuf[k * LB], (size_t) m * sizeof(KeyPos));
                lcnt[k] = 0;
            };
#pragma omp for schedule(dynamic, 4) nowait
            for (uint64_t blk = 0; blk < (n + 262143) / 262144; ++blk) {
                const uint64_t lo = blk * 262144, hi = lo + 262144 < n ? lo + 262144 : n;
                uint64_t win = t.window(lo);
                for (uint64_t p = lo; p < hi; ++p) {
                    const uint64_t bkt = win >> (64 - 2 * PB);
                    win = t.roll(win, p);
                    if (bkt < g0 || bkt >= g1) continue;
                    const uint64_t k = bkt - g0;
                    lbuf[k * LB + lcnt[k]] = KeyPos{t.window(p + (uint64_t) PB), p};
                    if (++lcnt[k] == LB) flush(bkt);
                }
            }
            for (uint64_t bkt = g0; bkt < g1; ++bkt) flush(bkt);
        }
        tm.lap("sa: collect group");
        // sort every bucket: by key in cache, ties through the text
        uint64_t work = 0;
#pragma omp parallel num_threads(lrm_host_threads()) reduction(+ : work)
        {
        std::vector<KeyPos> tmp;
#pragma omp for schedule(dynamic, 1)
        for (uint64_t bkt = g0; bkt < g1; ++bkt) {
            KeyPos *a = scratch + (cnt[bkt] - base);
            const uint64_t m = cnt[bkt + 1] - cnt[bkt];
            if (m == 0) continue;
            sort_by_key(a, m, tmp);
            uint64_t wk = 0;
            sort_ties(a, m, [&](const KeyPos &x, const KeyPos &y) { return suffix_less(t, x.pos, y.pos, (uint64_t) PB + 32, wk); });
            work += wk;
            uint64_t *dst = sa64 + 1 + cnt[bkt];
            for (uint64_t i = 0; i < m; ++i) dst[i] = a[i].pos;
        }
        }
        work_total += work;
        tm.lap("sa: sort group");
        if (work_total > budget) over = true;
        g0 = g1;
    }
    free(scratch);
    return over ? 1 : 0;           // 1: too repetitive for this scheme, the caller falls back to SA-IS
}

}  // namespace

static thread_local const PackedText *g_packed_for_build = nullptr;

// LRM_SA_ALGO = "sais" forces the linear-time sorter, "bucket" the parallel one (no fallback budget).
extern "C" int lrm_sa_build(const char *text, uint64_t L, lrm_ui40 *out) {
    if (!text || !out || L < 1) { lrm_set_error("bad argument"); return -1; }
    if (text[L - 1] != '$') { lrm_set_error("text must end in '$'"); return -1; }
    {   // '$' must not occur inside the text
        uint64_t inner = ~0ull;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static) reduction(min : inner)
        for (uint64_t i = 0; i < L - 1; ++i) if (text[i] == '$' && i < inner) inner = i;
        if (inner != ~0ull) { lrm_set_error("'$' occurs inside the text (offset %llu)", (unsigned long long) inner); return -1; }
    }
    const char *algo = getenv("LRM_SA_ALGO");
    const bool force_sais = algo && !strcmp(algo, "sais"), force_bucket = algo && !strcmp(algo, "bucket");
    if (!force_sais && L >= 2) {
        PackedText own;
        const PackedText *t = g_packed_for_build;          // lrm_host_index_build shares its packed text
        if (!t) { if (pack_text(text, L, own)) t = &own; }
        if (t) {
            const int rc = sa_build_bucketed(*t, L, out, force_bucket ? (1ull << 40) / (L ? L : 1) + 1024 : 24);
            if (rc == 0) return 0;
            if (rc < 0) return -1;
        }
        // bytes other than upper-case ACGT, or too repetitive: the generic linear-time path below
    }
    // remap to a dense alphabet, '$' -> 0 (must be unique and last)
    std::vector<uint8_t> s((size_t) L);
    int map[256];
    bool seen[256] = {false};
    for (uint64_t i = 0; i < L; ++i) seen[(unsigned char) text[i]] = true;
    int K = 0;
    for (int c = 0; c < 256; ++c) map[c] = seen[c] ? K++ : -1;
    if (map[(unsigned char) '$'] != 0) { lrm_set_error("'$' is not the smallest byte of the text"); return -1; }
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
    for (uint64_t i = 0; i < L; ++i) s[i] = (uint8_t) map[(unsigned char) text[i]];
    uint64_t *sa64 = reinterpret_cast<uint64_t *>(out);          // ui40 slots are 8 bytes: value < 2^40, padding zeroed
    if (L < (1ull << 31) - 8) {
        std::vector<int32_t> sa((size_t) L);
        SaIs<int32_t>::run<uint8_t>(s.data(), sa.data(), (int32_t) L, (int32_t) (K - 1));
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t i = 0; i < L; ++i) sa64[i] = (uint64_t) (uint32_t) sa[i];
    } else {
        // in place: SA-IS runs on the caller's 8-byte slots (int64), no second array
        static_assert(sizeof(lrm_ui40) == 8, "ui40 is 8 bytes in RAM (sa_use.h:17-20)");
        SaIs<int64_t>::run<uint8_t>(s.data(), reinterpret_cast<int64_t *>(out), (int64_t) L, (int64_t) (K - 1));
    }
    return 0;
}

extern "C" int lrm_cat_from_seqs(const char *const *names, const char *const *seqs, const uint64_t *lens, int nseq,
                                 uint64_t n_seed, char **cat_out, uint64_t *cat_len, lrm_mta_entry **mta_out) {
    if (!seqs || !lens || nseq <= 0 || !cat_out || !cat_len || !mta_out) { lrm_set_error("bad argument"); return -1; }
    uint64_t total = 1;
    for (int i = 0; i < nseq; ++i) total += 2 * lens[i];
    char *cat = (char *) malloc(total + 1);
    lrm_mta_entry *mta = (lrm_mta_entry *) calloc((size_t) nseq, sizeof(lrm_mta_entry));
    if (!cat || !mta) { free(cat); free(mta); lrm_set_error("out of memory"); return -1; }
    uint64_t off = 0, rs = n_seed;
    for (int i = 0; i < nseq; ++i) {
        const uint64_t n = lens[i];
        char nm[32];
        const char *name = names && names[i] ? names[i] : nm;
        if (!(names && names[i])) snprintf(nm, sizeof(nm), "seq%d", i);
        mta[i].name_len = strlen(name);
        mta[i].name = strdup(name);
        mta[i].name_own = 1;
        mta[i].offset = off;                          // asindex.c:89-93
        mta[i].seq_len = n;
        // N/n -> pseudo-random base (asindex.c:53-60; seeded per position here, so the result does not depend on
        // the thread count), upper-casing (asindex.c:63-68)
        uint64_t bad_pos = ~0ull;
        const uint64_t rs0 = splitmix64(rs);
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static) reduction(min : bad_pos)
        for (uint64_t p = 0; p < n; ++p) {
            char c = seqs[i][p];
            if (c == 'n' || c == 'N') { uint64_t st = rs0 ^ (p * 0x9E3779B97F4A7C15ull); c = "ACGT"[splitmix64(st) & 3]; }
            if (c > 0x60) c -= 0x20;
            if (dna_code(c) < 0 && p < bad_pos) bad_pos = p;
            cat[off + p] = c;
        }
        if (bad_pos != ~0ull) {
            lrm_set_error("sequence %d offset %llu: byte 0x%02x is not a nucleotide", i, (unsigned long long) bad_pos, (unsigned) (unsigned char) seqs[i][bad_pos]);
            free(cat); lrm_mta_free(mta, nseq);
            return -1;
        }
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t p = 0; p < n; ++p) cat[off + n + p] = "TGCA"[dna_code(cat[off + n - 1 - p])];   // asindex.c:70-75
        off += 2 * n;
    }
    cat[off++] = '$';                                  // asindex.c:109-110
    cat[off] = 0;
    *cat_out = cat; *cat_len = off; *mta_out = mta;
    return 0;
}

extern "C" void lrm_mta_free(lrm_mta_entry *mta, int n) {
    if (!mta) return;
    for (int i = 0; i < n; ++i) if (mta[i].name_own) free(mta[i].name);
    free(mta);
}

extern "C" void lrm_host_index_free(lrm_host_index *idx) {
    if (!idx) return;
    free(idx->fmi.c); free(idx->fmi.o); free(idx->fmi.csa); free(idx->fmi.bwt);
    free(idx->lch.lc); free(idx->sa.mem); free(idx->content);
    lrm_mta_free(idx->mta, idx->mta_len);
    memset(idx, 0, sizeof(*idx));
}

static inline uint64_t ui40v(const lrm_ui40 &v) { return ((uint64_t) v.high << 32) | v.low; }

extern "C" int lrm_host_index_build(const char *cat, uint64_t L, const lrm_mta_entry *mta, int mta_len, int o_ratio,
                                    int hlen, lrm_host_index *out) {
    if (!cat || !out || L < 2 || o_ratio < 1 || hlen < 1 || hlen > 15) { lrm_set_error("bad argument"); return -1; }
    memset(out, 0, sizeof(*out));
    out->content = (char *) malloc(L + 1);
    out->sa.mem = (lrm_ui40 *) malloc(sizeof(lrm_ui40) * L);
    if (!out->content || !out->sa.mem) { lrm_host_index_free(out); lrm_set_error("out of memory"); return -1; }
    {
        const uint64_t piece = 1ull << 22, np = (L + piece - 1) / piece;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t i = 0; i < np; ++i) memcpy(out->content + i * piece, cat + i * piece, L - i * piece < piece ? L - i * piece : piece);
    }
    out->content[L] = 0;
    out->con_len = L;
    out->mta_len = mta_len;
    out->mta = (lrm_mta_entry *) calloc((size_t) (mta_len > 0 ? mta_len : 1), sizeof(lrm_mta_entry));
    for (int i = 0; i < mta_len; ++i) {
        out->mta[i] = mta[i];
        out->mta[i].name = mta[i].name ? strdup(mta[i].name) : strdup("");
        out->mta[i].name_own = 1;
    }
    out->sa.start = 0;
    out->sa.len = L;
    StageTimer tm;
    PackedText pt;
    const bool pure = pack_text(cat, L, pt);            // 2-bit image: a quarter of the footprint for the random accesses below
    tm.lap("pack 2-bit");
    g_packed_for_build = pure ? &pt : nullptr;
    const int sa_rc = lrm_sa_build(cat, L, out->sa.mem);
    g_packed_for_build = nullptr;
    if (sa_rc) { lrm_host_index_free(out); return -1; }
    tm.lap("suffix array");
    const lrm_ui40 *sa = out->sa.mem;

    lrm_dna_fmi *f = &out->fmi;
    // C table: counts over text[0..L-2], exclusive prefix sums over all byte values (fmidx.c:101-125)
    f->c = (uint64_t *) calloc(256, sizeof(uint64_t));
    {
        uint64_t ca = 0, cc = 0, cg = 0, ct = 0, other = 0;
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static) reduction(+ : ca, cc, cg, ct, other)
        for (uint64_t i = 0; i < L - 1; ++i) {
            switch (cat[i]) { case 'A': ca++; break; case 'C': cc++; break; case 'G': cg++; break; case 'T': ct++; break; default: other++; }
        }
        f->c[(unsigned char) 'A'] = ca; f->c[(unsigned char) 'C'] = cc; f->c[(unsigned char) 'G'] = cg; f->c[(unsigned char) 'T'] = ct;
        if (other) {                                       // generic bytes: the plain loop
            memset(f->c, 0, 256 * sizeof(uint64_t));
            for (uint64_t i = 0; i + 1 < L; ++i) f->c[(unsigned char) cat[i]]++;
        }
    }
    { uint64_t sum = 0; for (int i = 0; i < 256; ++i) { uint64_t t = sum + f->c[i]; f->c[i] = sum; sum = t; } }
    // BWT (fmidx.c:76-98)
    f->length = L;
    f->bwt = (char *) malloc(L + 1);
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
    for (uint64_t i = 0; i < L; ++i) {
        const uint64_t v = ui40v(sa[i]);
        f->bwt[i] = v == 0 ? '$' : (pure ? "ACGT"[(pt.w[(v - 1) >> 5] >> (62 - 2 * ((v - 1) & 31))) & 3] : cat[v - 1]);
    }
    f->bwt[L] = 0;
    tm.lap("C + bwt");
    // O table (fmidx.c:128-150,186-190)
    f->o_ratio = o_ratio;
    f->o_len = 4 * (L / (uint64_t) o_ratio + 1);
    f->o = (uint64_t *) calloc(f->o_len, sizeof(uint64_t));
    {   // segments of SEG sample intervals: counts per segment first, then every segment fills its samples
        const uint64_t R = (uint64_t) o_ratio, SEG = 1ull << 15, rows_per_seg = SEG * R, nseg = (L + rows_per_seg - 1) / rows_per_seg;
        std::vector<uint64_t> segc((nseg + 1) * 4, 0);
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t sg = 0; sg < nseg; ++sg) {
            uint64_t c[4] = {0, 0, 0, 0};
            const uint64_t lo = sg * rows_per_seg, hi = lo + rows_per_seg < L ? lo + rows_per_seg : L;
            for (uint64_t i = lo; i < hi; ++i) { const int code = dna_code(f->bwt[i]); if (code >= 0) c[code]++; }
            for (int x = 0; x < 4; ++x) segc[(sg + 1) * 4 + x] = c[x];
        }
        for (uint64_t sg = 1; sg <= nseg; ++sg) for (int x = 0; x < 4; ++x) segc[sg * 4 + x] += segc[(sg - 1) * 4 + x];
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
        for (uint64_t sg = 0; sg < nseg; ++sg) {
            uint64_t run[4] = {segc[sg * 4], segc[sg * 4 + 1], segc[sg * 4 + 2], segc[sg * 4 + 3]};
            const uint64_t lo = sg * rows_per_seg, hi = lo + rows_per_seg < L ? lo + rows_per_seg : L;
            for (uint64_t i = lo; i < hi; ++i) {
                if (i % R == 0) memcpy(f->o + 4 * (i / R), run, sizeof(run));
                const int code = dna_code(f->bwt[i]);
                if (code >= 0) run[code]++;
            }
        }
        // (when L is a multiple of R the sample past the last row stays 0, as fmidx.c:135-147 leaves it)
    }
    tm.lap("O table");
    // CSA (fmidx.c:153-163,194)
    f->csa_ratio = 4;
    f->csa_len = L / 4 + 1;
    f->csa = (uint64_t *) calloc(f->csa_len, sizeof(uint64_t));
#pragma omp parallel for num_threads(lrm_host_threads()) schedule(static)
    for (uint64_t i = 0; i < f->csa_len; ++i) f->csa[i] = i * 4 < L ? ui40v(sa[i * 4]) : 0;

    tm.lap("csa");
    // lchash (lchash.c:52-73): the SA interval of every hlen-mer.  Suffixes sharing their first
    // hlen bases are contiguous in the SA, so one pass finds every interval's first/last row.
    // Quirk kept: fmi_aln starts from rows [1, L-1] (lchash.c:56), i.e. without the '$' row, so
    // the occurrence that ends on the very last base of the text (the suffix "P$", always the
    // first row of P's interval) is never found.  That row is skipped here as well.
    const uint64_t upper = 1ull << (2 * hlen);
    out->lch.hlen = hlen;
    out->lch.len = 2 * upper;
    out->lch.lc = (uint64_t *) calloc(2 * upper, sizeof(uint64_t));
    auto code_at = [&](uint64_t row, uint64_t &code) -> bool {
        uint64_t pos = ui40v(sa[row]);
        if (pos + (uint64_t) hlen >= L - 1) return false;         // runs into '$', or is the "P$" row (see above)
        if (pure) { code = pt.window(pos) >> (64 - 2 * hlen); return true; }   // first base most significant (lchash.c:36-49)
        uint64_t c = 0;
        for (int i = 0; i < hlen; ++i) c = (c << 2) | (uint64_t) dna_code(cat[pos + i]);
        code = c;
        return true;
    };
    // one text access per row: the codes of a block of rows first, then the interval boundaries inside it
    const uint64_t RB = 1ull << 16, nrb = (L + RB - 1) / RB;
#pragma omp parallel num_threads(lrm_host_threads())
    {
        std::vector<uint64_t> codes(RB + 2);
#pragma omp for schedule(dynamic, 4)
        for (uint64_t b = 0; b < nrb; ++b) {
            const uint64_t lo = b * RB, hi = lo + RB < L ? lo + RB : L;
            const uint64_t NONE = ~0ull;
            for (uint64_t r = (lo ? lo - 1 : lo); r < (hi < L ? hi + 1 : hi); ++r) {
                uint64_t c;
                codes[r + 1 - lo] = code_at(r, c) ? c : NONE;
            }
            for (uint64_t r = lo; r < hi; ++r) {
                const uint64_t cur = codes[r + 1 - lo];
                if (cur == NONE) continue;
                if (r == 0 || codes[r - lo] != cur) out->lch.lc[2 * cur] = r;
                if (r + 1 == L || codes[r + 2 - lo] != cur) out->lch.lc[2 * cur + 1] = r;
            }
        }
    }
    tm.lap("lchash");
    return 0;
}

// ------------------------------------------------------------------------------------------
// files
// ------------------------------------------------------------------------------------------
#define IOCHK(cond, what) do { if (!(cond)) { lrm_set_error("%s: %s", what, path.c_str()); if (fp) fclose(fp); return -1; } } while (0)

extern "C" int lrm_fmi_write(const lrm_dna_fmi *fmi, const char *prefix) {      // fmidx.c:221-244
    std::string path = std::string(prefix) + ".mfi";
    FILE *fp = fopen(path.c_str(), "wb");
    IOCHK(fp, "cannot create");
    IOCHK(fwrite(fmi->c, sizeof(uint64_t), 256, fp) == 256, "write failed");
    IOCHK(fwrite(&fmi->o_ratio, sizeof(int), 1, fp) == 1, "write failed");
    IOCHK(fwrite(&fmi->o_len, sizeof(uint64_t), 1, fp) == 1, "write failed");
    IOCHK(fwrite(fmi->o, sizeof(uint64_t), fmi->o_len, fp) == fmi->o_len, "write failed");
    IOCHK(fwrite(&fmi->length, sizeof(uint64_t), 1, fp) == 1, "write failed");
    IOCHK(fwrite(fmi->bwt, 1, fmi->length, fp) == fmi->length, "write failed");
    IOCHK(fwrite(&fmi->csa_ratio, sizeof(int), 1, fp) == 1, "write failed");
    IOCHK(fwrite(&fmi->csa_len, sizeof(uint64_t), 1, fp) == 1, "write failed");
    IOCHK(fwrite(fmi->csa, sizeof(uint64_t), fmi->csa_len, fp) == fmi->csa_len, "write failed");
    fclose(fp);
    return 0;
}

extern "C" int lrm_fmi_read(lrm_dna_fmi *fmi, const char *prefix) {              // fmidx.c:246-275
    std::string path = std::string(prefix) + ".mfi";
    memset(fmi, 0, sizeof(*fmi));
    FILE *fp = fopen(path.c_str(), "rb");
    IOCHK(fp, "cannot open");
    fmi->c = (uint64_t *) malloc(256 * sizeof(uint64_t));
    IOCHK(fread(fmi->c, sizeof(uint64_t), 256, fp) == 256, "short read (C)");
    IOCHK(fread(&fmi->o_ratio, sizeof(int), 1, fp) == 1, "short read");
    IOCHK(fread(&fmi->o_len, sizeof(uint64_t), 1, fp) == 1, "short read");
    fmi->o = (uint64_t *) malloc(fmi->o_len * sizeof(uint64_t));
    IOCHK(fmi->o && fread(fmi->o, sizeof(uint64_t), fmi->o_len, fp) == fmi->o_len, "short read (O)");
    IOCHK(fread(&fmi->length, sizeof(uint64_t), 1, fp) == 1, "short read");
    fmi->bwt = (char *) malloc(fmi->length + 1);
    IOCHK(fmi->bwt && fread(fmi->bwt, 1, fmi->length, fp) == fmi->length, "short read (bwt)");
    fmi->bwt[fmi->length] = 0;
    IOCHK(fread(&fmi->csa_ratio, sizeof(int), 1, fp) == 1, "short read");
    IOCHK(fread(&fmi->csa_len, sizeof(uint64_t), 1, fp) == 1, "short read");
    fmi->csa = (uint64_t *) malloc(fmi->csa_len * sizeof(uint64_t));
    IOCHK(fmi->csa && fread(fmi->csa, sizeof(uint64_t), fmi->csa_len, fp) == fmi->csa_len, "short read (csa)");
    fclose(fp);
    return 0;
}

extern "C" int lrm_lc_write(const char *p, const lrm_lc_hash *h) {                // lchash.c:106-112
    std::string path = p;
    FILE *fp = fopen(p, "wb");
    IOCHK(fp, "cannot create");
    IOCHK(fwrite(&h->hlen, sizeof(int), 1, fp) == 1, "write failed");
    IOCHK(fwrite(&h->len, sizeof(uint64_t), 1, fp) == 1, "write failed");
    IOCHK(fwrite(h->lc, sizeof(uint64_t), h->len, fp) == h->len, "write failed");
    fclose(fp);
    return 0;
}

extern "C" int lrm_lc_read(const char *p, lrm_lc_hash *h) {                       // lchash.c:114-127
    std::string path = p;
    memset(h, 0, sizeof(*h));
    FILE *fp = fopen(p, "rb");
    IOCHK(fp, "cannot open");
    IOCHK(fread(&h->hlen, sizeof(int), 1, fp) == 1, "short read");
    IOCHK(fread(&h->len, sizeof(uint64_t), 1, fp) == 1, "short read");
    h->lc = (uint64_t *) malloc(h->len * sizeof(uint64_t));
    IOCHK(h->lc && fread(h->lc, sizeof(uint64_t), h->len, fp) == h->len, "short read (lc)");
    fclose(fp);
    return 0;
}

extern "C" int lrm_sa5_write(const char *p, const lrm_ui40 *mem, uint64_t n) {    // 5-byte LE entries (uint40.h)
    std::string path = p;
    FILE *fp = fopen(p, "wb");
    IOCHK(fp, "cannot create");
    const uint64_t CH = 1 << 20;
    std::vector<uint8_t> buf(CH * 5);
    for (uint64_t i = 0; i < n; i += CH) {
        uint64_t m = n - i < CH ? n - i : CH;
        for (uint64_t j = 0; j < m; ++j) {
            memcpy(&buf[j * 5], &mem[i + j].low, 4);
            buf[j * 5 + 4] = mem[i + j].high;
        }
        IOCHK(fwrite(buf.data(), 5, m, fp) == m, "write failed");
    }
    fclose(fp);
    return 0;
}

extern "C" int64_t lrm_sa5_read(const char *p, lrm_ui40 *mem, uint64_t nitems) {  // sa_use.h:31-46
    std::string path = p;
    FILE *fp = fopen(p, "rb");
    IOCHK(fp, "cannot open");
    const uint64_t CH = 1 << 20;
    std::vector<uint8_t> buf(CH * 5);
    uint64_t got = 0;
    while (got < nitems) {
        uint64_t want = nitems - got < CH ? nitems - got : CH;
        size_t nb = fread(buf.data(), 1, want * 5, fp);
        uint64_t m = nb / 5;
        for (uint64_t j = 0; j < m; ++j) {
            uint64_t v = 0;
            memcpy(&v, &buf[j * 5], 5);                                      // little-endian: low, then high
            memcpy(&mem[got + j], &v, 8);                                    // padding bytes zeroed
        }
        got += m;
        if (m < want) break;
    }
    fclose(fp);
    return (int64_t) got;
}

extern "C" int lrm_mta_write(const char *p, const lrm_mta_entry *mta, int n) {    // asindex.c:89-93, mutils.c:53-56
    std::string path = p;
    FILE *fp = fopen(p, "wb");
    IOCHK(fp, "cannot create");
    for (int i = 0; i < n; ++i) {
        uint64_t l = mta[i].name_len;
        size_t sl = mta[i].seq_len;
        IOCHK(fwrite(&l, sizeof(uint64_t), 1, fp) == 1, "write failed");
        IOCHK(fwrite(mta[i].name, 1, l, fp) == l, "write failed");
        IOCHK(fwrite(&mta[i].offset, sizeof(uint64_t), 1, fp) == 1, "write failed");
        IOCHK(fwrite(&sl, sizeof(size_t), 1, fp) == 1, "write failed");
    }
    fclose(fp);
    return 0;
}

extern "C" int lrm_mta_read(const char *p, lrm_mta_entry **mta_out) {             // alnmain.c:125-140
    std::string path = p;
    FILE *fp = fopen(p, "rb");
    IOCHK(fp, "cannot open");
    std::vector<lrm_mta_entry> v;
    while (true) {
        uint64_t l;
        if (fread(&l, sizeof(uint64_t), 1, fp) != 1) break;
        lrm_mta_entry e;
        memset(&e, 0, sizeof(e));
        e.name_len = l;
        e.name = (char *) malloc(l + 1);
        e.name_own = 1;
        if (fread(e.name, 1, l, fp) != l) { free(e.name); break; }
        e.name[l] = 0;
        size_t sl = 0;
        if (fread(&e.offset, sizeof(uint64_t), 1, fp) != 1 || fread(&sl, sizeof(size_t), 1, fp) != 1) { free(e.name); break; }
        e.seq_len = sl;
        v.push_back(e);
        if (v.size() >= 65535) break;                                            // alnmain.c:127
    }
    fclose(fp);
    lrm_mta_entry *out = (lrm_mta_entry *) calloc(v.size() ? v.size() : 1, sizeof(lrm_mta_entry));
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    *mta_out = out;
    return (int) v.size();
}

static int write_file(const std::string &path, const char *buf, uint64_t n) {
    FILE *fp = fopen(path.c_str(), "wb");
    IOCHK(fp, "cannot create");
    IOCHK(fwrite(buf, 1, n, fp) == n, "write failed");
    fclose(fp);
    return 0;
}

extern "C" int lrm_host_index_write(const lrm_host_index *idx, const char *genome) {
    std::string g = genome, cat = g + ".cat";
    if (lrm_mta_write((g + ".mta").c_str(), idx->mta, idx->mta_len)) return -1;
    if (write_file(cat, idx->content, idx->con_len)) return -1;
    if (lrm_fmi_write(&idx->fmi, cat.c_str())) return -1;
    if (lrm_lc_write((cat + ".lch").c_str(), &idx->lch)) return -1;
    if (lrm_sa5_write((cat + ".sa5").c_str(), idx->sa.mem, idx->sa.len)) return -1;
    return 0;
}

extern "C" int lrm_host_index_read(const char *genome, lrm_host_index *out) {      // alnmain.c:179-256 (init)
    memset(out, 0, sizeof(*out));
    std::string g = genome, cat = g + ".cat";
    if (lrm_fmi_read(&out->fmi, cat.c_str())) { lrm_host_index_free(out); return -1; }
    if (lrm_lc_read((cat + ".lch").c_str(), &out->lch)) { lrm_host_index_free(out); return -1; }
    int n = lrm_mta_read((g + ".mta").c_str(), &out->mta);
    if (n < 0) { lrm_host_index_free(out); return -1; }
    out->mta_len = n;
    {
        std::string path = cat;
        FILE *fp = fopen(cat.c_str(), "rb");
        if (!fp) { lrm_set_error("cannot open: %s", cat.c_str()); lrm_host_index_free(out); return -1; }
        fseek(fp, 0, SEEK_END);
        long l = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        out->content = (char *) malloc((size_t) l + 1);
        if (fread(out->content, 1, (size_t) l, fp) != (size_t) l) { fclose(fp); lrm_set_error("short read: %s", cat.c_str()); lrm_host_index_free(out); return -1; }
        out->content[l] = 0;
        out->con_len = (uint64_t) l;
        fclose(fp);
    }
    out->sa.mem = (lrm_ui40 *) malloc(sizeof(lrm_ui40) * out->con_len);
    int64_t got = lrm_sa5_read((cat + ".sa5").c_str(), out->sa.mem, out->con_len);
    if (got < 0) { lrm_host_index_free(out); return -1; }
    out->sa.start = 0;
    out->sa.len = (uint64_t) got;
    return 0;
}

// FASTA (plain text) records -> names/sequences
static int read_fasta(const char *path_, std::vector<std::string> &names, std::vector<std::string> &seqs) {
    std::string path = path_;
    FILE *fp = fopen(path_, "rb");
    IOCHK(fp, "cannot open");
    std::vector<char> line(1 << 16);
    bool have = false;
    while (fgets(line.data(), (int) line.size(), fp)) {
        size_t l = strlen(line.data());
        bool full = l > 0 && line[l - 1] == '\n';
        while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        if (line[0] == '>') {
            std::string nm(line.data() + 1);
            size_t sp = nm.find_first_of(" \t");                      // kseq: name ends at first whitespace
            if (sp != std::string::npos) nm.resize(sp);
            names.push_back(nm);
            seqs.emplace_back();
            have = true;
            while (!full && fgets(line.data(), (int) line.size(), fp)) {  // swallow the rest of a long header
                size_t k = strlen(line.data());
                full = k > 0 && line[k - 1] == '\n';
            }
        } else if (have) {
            seqs.back().append(line.data(), l);
        }
    }
    fclose(fp);
    return 0;
}

extern "C" int lrm_accidx(const char *genome, int o_ratio, int hlen, uint64_t n_seed) {   // asindex.c:129-153
    std::vector<std::string> names, seqs;
    if (read_fasta(genome, names, seqs)) return -1;
    if (seqs.empty()) { lrm_set_error("no FASTA records in %s", genome); return -1; }
    std::vector<const char *> np, sp;
    std::vector<uint64_t> lens;
    for (size_t i = 0; i < seqs.size(); ++i) { np.push_back(names[i].c_str()); sp.push_back(seqs[i].c_str()); lens.push_back(seqs[i].size()); }
    char *cat = nullptr;
    uint64_t L = 0;
    lrm_mta_entry *mta = nullptr;
    if (lrm_cat_from_seqs(np.data(), sp.data(), lens.data(), (int) seqs.size(), n_seed, &cat, &L, &mta)) return -1;
    lrm_host_index idx;
    int rc = lrm_host_index_build(cat, L, mta, (int) seqs.size(), o_ratio, hlen, &idx);
    free(cat);
    lrm_mta_free(mta, (int) seqs.size());
    if (rc) return -1;
    rc = lrm_host_index_write(&idx, genome);
    lrm_host_index_free(&idx);
    return rc;
}
