// synth.cpp -- seeded synthetic references and long reads (bench / test tooling, liblrm_synth.so).
// The reference has no simulator (scripts/dna_txt_gen.py:1-12 is an unseeded random text
// generator); these are the workload definitions of SURVEY.md 8(d): splitmix64 streams,
// i.i.d. ACGT references with optional planted repeat families, reads with per-base
// substitution / insertion / deletion rates truncated to an exact length.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <omp.h>

static inline uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t &s) { return (double) (splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

extern "C" {

// i.i.d. upper-case ACGT of n bases.  repeat_frac > 0 plants families of `rep_len`-base
// elements, `rep_copies` copies each at `rep_div` per-base divergence, until that fraction
// of the sequence is covered.
int lrm_synth_reference(char *out, uint64_t n, uint64_t seed, double repeat_frac, uint32_t rep_len,
                        uint32_t rep_copies, double rep_div) {
    const uint64_t CH = 1 << 16;
    const int64_t nch = (int64_t) ((n + CH - 1) / CH);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nch; ++c) {
        uint64_t s = seed ^ (0xA5A5A5A5ull + (uint64_t) c * 0xD1B54A32D192ED03ull);
        uint64_t lo = (uint64_t) c * CH, hi = lo + CH < n ? lo + CH : n;
        uint64_t bits = 0; int left = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            if (!left) { bits = splitmix64(s); left = 32; }
            out[i] = "ACGT"[bits & 3]; bits >>= 2; left--;
        }
    }
    if (repeat_frac > 0 && rep_len > 0 && rep_copies > 0 && n > 4ull * rep_len) {
        uint64_t s = seed ^ 0x5EED5EEDull;
        uint64_t target = (uint64_t) (repeat_frac * (double) n), covered = 0;
        std::vector<char> elem(rep_len);
        while (covered < target) {
            for (uint32_t i = 0; i < rep_len; ++i) elem[i] = "ACGT"[splitmix64(s) & 3];
            for (uint32_t cp = 0; cp < rep_copies && covered < target; ++cp) {
                uint64_t pos = splitmix64(s) % (n - rep_len);
                for (uint32_t i = 0; i < rep_len; ++i) {
                    char b = elem[i];
                    if (u01(s) < rep_div) b = "ACGT"[((b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3) + 1 + splitmix64(s) % 3) & 3];
                    out[pos + i] = b;
                }
                covered += rep_len;
            }
        }
    }
    return 0;
}

// Reads from a set of forward sequences laid back to back in `ref` (sequence s at
// seq_off[s], seq_len[s] bases).  Read r: sequence chosen by length, start uniform, strand
// 50/50, per-reference-base edits, exactly `read_len` bases (reads_out + r*stride, NUL padded).
// truth_*: sequence id, leftmost reference base, reference bases consumed, strand.
int lrm_synth_reads(const char *ref, const uint64_t *seq_off, const uint64_t *seq_len, int nseq,
                    uint64_t n_reads, uint32_t read_len, double p_sub, double p_ins, double p_del,
                    uint64_t seed, char *reads_out, uint64_t stride, uint32_t *lens_out,
                    int32_t *truth_seq, uint64_t *truth_pos, uint32_t *truth_span, uint8_t *truth_strand,
                    int nthreads) {
    if (nseq <= 0 || stride < (uint64_t) read_len + 1) return -1;
    std::vector<double> cum(nseq);
    double tot = 0;
    for (int i = 0; i < nseq; ++i) { tot += (double) seq_len[i]; cum[i] = tot; }
    if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int64_t r = 0; r < (int64_t) n_reads; ++r) {
        uint64_t s = seed ^ (0x1234567ull + (uint64_t) r * 0x9E3779B97F4A7C15ull);
        splitmix64(s);
        double pick = u01(s) * tot;
        int sid = 0;
        while (sid + 1 < nseq && cum[sid] <= pick) sid++;
        const uint64_t sl = seq_len[sid];
        const char *sq = ref + seq_off[sid];
        uint64_t span = (uint64_t) ((double) read_len * 1.3) + 32;
        uint64_t start = sl > span ? splitmix64(s) % (sl - span) : 0;
        uint8_t strand = (uint8_t) (splitmix64(s) & 1);
        char *out = reads_out + (uint64_t) r * stride;
        uint32_t k = 0;
        uint64_t p = start;
        while (k < read_len) {
            if (p >= sl) { out[k++] = "ACGT"[splitmix64(s) & 3]; continue; }   // ran off the sequence: random tail
            char b = sq[p++];
            double x = u01(s);
            if (x < p_del) continue;
            if (x < p_del + p_sub) {
                int c = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3;
                b = "ACGT"[(c + 1 + splitmix64(s) % 3) & 3];
            }
            out[k++] = b;
            if (k < read_len && u01(s) < p_ins) out[k++] = "ACGT"[splitmix64(s) & 3];
        }
        if (strand) {
            for (uint32_t i = 0; i < read_len / 2; ++i) {
                char a = comp(out[i]), c = comp(out[read_len - 1 - i]);
                out[i] = c; out[read_len - 1 - i] = a;
            }
            if (read_len & 1) out[read_len / 2] = comp(out[read_len / 2]);
        }
        memset(out + read_len, 0, stride - read_len);
        lens_out[r] = read_len;
        if (truth_seq) truth_seq[r] = sid;
        if (truth_pos) truth_pos[r] = start;
        if (truth_span) truth_span[r] = (uint32_t) (p - start);
        if (truth_strand) truth_strand[r] = strand;
    }
    return 0;
}

int lrm_synth_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
