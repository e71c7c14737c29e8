// io_host.cpp -- FASTA/FASTQ batches, SAM text, and the accaln flow on the GPU path
// (include/lrm_io_host.h).  Host-side C++; the kernels are reached only through the C-ABI.
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <new>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <omp.h>
#include "../../include/lrm_io_host.h"
#include "../../include/lrm_index_host.h"
#include "lrm_internal.h"

// ------------------------------------------------------------------------------------------
// FASTA / FASTQ batches (reads_load + refactor_reads_seq, accaln.c:45-58, alnmain.c:87-103)
//
// Two parsers over one buffered byte source:
//   fast   4-line FASTQ records (what every long-read basecaller writes): the buffered block is cut at arbitrary byte
//          offsets into one piece per host thread, every piece resynchronises to a record boundary (a line that starts
//          with '@' whose line-after-next starts with '+': a quality line may start with '@', but then the line after
//          next is a sequence, and a sequence never starts with '+'), the pieces are indexed in parallel and the
//          sequences / names / qualities are copied into the dense batch in parallel.  Plain files are read with
//          parallel preads, gzip streams are inflated by zlib (one thread) into the same buffer.
//   slow   the general kseq-like parser (multi-line records, FASTA, CR LF): taken for the rest of the file as soon as
//          the fast parser meets a record that is not four LF-terminated lines.
// ------------------------------------------------------------------------------------------
// Large host buffers (the reader's block, a batch's sequences and its name / quality arenas) are recycled through a small
// pool instead of going back to the allocator: a freed gigabyte is unmapped by malloc and the next batch pays for its
// page faults again -- with every thread of a copy loop faulting on one address space that is seconds per gigabyte on
// some kernels (tools/io_probe.py: 0.4 GB/s the first time, 7-25 GB/s on warm pages).
namespace bigmem {
struct Hdr { size_t cap; size_t pad[7]; };
static_assert(sizeof(Hdr) == 64, "header keeps the payload 64-byte aligned");
std::mutex mu;
std::vector<Hdr *> pool;                 // free blocks (at most 12, largest kept)
void *alloc(size_t n) {
    if (n == 0) n = 1;
    {
        std::lock_guard<std::mutex> g(mu);
        size_t best = pool.size();
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i]->cap >= n && pool[i]->cap <= 2 * n + (1u << 20) && (best == pool.size() || pool[i]->cap < pool[best]->cap)) best = i;
        if (best != pool.size()) { Hdr *h = pool[best]; pool.erase(pool.begin() + (long) best); return h + 1; }
    }
    const size_t cap = n + n / 8;
    Hdr *h = (Hdr *) malloc(sizeof(Hdr) + cap);
    if (!h) return nullptr;
    h->cap = cap;
    return h + 1;
}
size_t capacity(void *p) { return p ? ((Hdr *) p - 1)->cap : 0; }
void release(void *p) {
    if (!p) return;
    Hdr *h = (Hdr *) p - 1;
    Hdr *drop = nullptr;
    {
        std::lock_guard<std::mutex> g(mu);
        pool.push_back(h);
        if (pool.size() > 12) {                                    // too many: the smallest goes back to the allocator
            size_t sm = 0;
            for (size_t i = 1; i < pool.size(); ++i) if (pool[i]->cap < pool[sm]->cap) sm = i;
            drop = pool[sm];
            pool.erase(pool.begin() + (long) sm);
        }
    }
    free(drop);
}
}  // namespace bigmem

struct lrm_reader {
    gzFile fp = nullptr;
    int fd = -1;             // plain file: read with pread (fp is null then)
    uint64_t file_off = 0;
    struct RawBuf {          // grows without initialising (a std::vector would clear every gigabyte it grows by)
        char *p = nullptr; size_t cap = 0;
        char *data() { return p; }
        const char *data() const { return p; }
        size_t size() const { return cap; }
        char &operator[](size_t i) { return p[i]; }
        void resize(size_t n, size_t keep) {        // keeps the first `keep` bytes
            if (n <= cap) return;
            char *q = (char *) bigmem::alloc(n);
            if (!q) throw std::bad_alloc();
            if (keep) memcpy(q, p, keep);
            bigmem::release(p);
            p = q; cap = bigmem::capacity(q);
        }
        ~RawBuf() { bigmem::release(p); }
    } buf;
    size_t pos = 0, end = 0;
    int last = 0;            // slow parser: header character already consumed ('>' or '@'), 0 = none
    bool eof = false;
    bool fast = true;        // the 4-line FASTQ fast parser is still viable
    size_t rec_bytes = 0;    // bytes per record seen so far (sizes the next block)

    // appends up to `want` bytes at buf[end..): returns the number read (0 at end of input)
    size_t read_more(size_t want) {
        if (eof || want == 0) return 0;
        if (buf.size() < end + want) buf.resize(end + want, end);
        size_t got = 0;
        if (fd >= 0) {
            const size_t piece = 8u << 20, np = (want + piece - 1) / piece;
            std::vector<ssize_t> gotp(np, 0);
            const int nt = (int) (np < (size_t) lrm_host_threads() ? np : (size_t) lrm_host_threads());
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt > 1 ? nt : 1)
            for (size_t i = 0; i < np; ++i) {
                const size_t o = i * piece, l = want - o < piece ? want - o : piece;
                size_t done = 0;
                while (done < l) {
                    const ssize_t k = pread(fd, buf.data() + end + o + done, l - done, (off_t) (file_off + o + done));
                    if (k <= 0) break;
                    done += (size_t) k;
                }
                gotp[i] = (ssize_t) done;
            }
            for (size_t i = 0; i < np; ++i) {
                got += (size_t) gotp[i];
                if ((size_t) gotp[i] < (want - i * piece < piece ? want - i * piece : piece)) break;      // short piece: end of file
            }
            file_off += got;
        } else {
            while (got < want) {
                const unsigned ask = (unsigned) (want - got < (1u << 30) ? want - got : (1u << 30));
                const int k = gzread(fp, buf.data() + end + got, ask);
                if (k <= 0) break;
                got += (size_t) k;
            }
        }
        if (got < want) eof = true;
        end += got;
        return got;
    }
    void compact() {
        if (pos == 0) return;
        if (end > pos) memmove(buf.data(), buf.data() + pos, end - pos);
        end -= pos;
        pos = 0;
    }
    // ---- slow parser primitives ----
    int getc_() {
        if (pos == end) {
            pos = end = 0;
            if (read_more(1u << 20) == 0) return -1;
        }
        return (unsigned char) buf[pos++];
    }
    // appends the rest of the current line (without the newline) to s; returns false at EOF before any byte.
    // Whole buffer spans at a time (memchr + one append): a 10 kbp sequence line is one or two appends.
    bool line_(std::string &s) {
        bool any = false;
        while (true) {
            if (pos == end) {
                pos = end = 0;
                if (read_more(1u << 20) == 0) return any;
            }
            any = true;
            const char *b = buf.data() + pos;
            const char *nl = (const char *) memchr(b, '\n', end - pos);
            size_t len = nl ? (size_t) (nl - b) : end - pos;
            pos += len + (nl ? 1 : 0);
            const size_t at = s.size();
            s.append(b, len);
            if (len && memchr(b, '\r', len)) {                     // CR LF files: carriage returns are dropped wherever they are
                size_t w = at;
                for (size_t i = at; i < s.size(); ++i) if (s[i] != '\r') s[w++] = s[i];
                s.resize(w);
            }
            if (nl) return true;
        }
    }
};

extern "C" int lrm_reader_open(lrm_reader **out, const char *path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { lrm_set_error("cannot open: %s", path); return -1; }
    unsigned char magic[2] = {0, 0};
    const ssize_t k = pread(fd, magic, 2, 0);
    lrm_reader *r = new lrm_reader;
    if (k == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {          // gzip: inflate through zlib
        r->fp = gzdopen(fd, "rb");
        if (!r->fp) { close(fd); delete r; lrm_set_error("cannot open: %s", path); return -1; }
        gzbuffer(r->fp, 1u << 20);
    } else {
        r->fd = fd;
    }
    *out = r;
    return 0;
}

extern "C" void lrm_reader_close(lrm_reader *r) {
    if (!r) return;
    if (r->fp) gzclose(r->fp);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

extern "C" void lrm_read_batch_free(lrm_read_batch *b) {
    if (!b) return;
    const bool pooled = b->name_arena != nullptr;              // a batch of the parallel parser: its large buffers go back to the pool
    if (!b->seqs_borrowed) { if (pooled) bigmem::release(b->seqs); else free(b->seqs); }
    free(b->lens);
    if (pooled) { bigmem::release(b->name_arena); bigmem::release(b->qual_arena); }
    else for (uint64_t i = 0; i < b->n; ++i) { free(b->names ? b->names[i] : nullptr); free(b->quals ? b->quals[i] : nullptr); }
    free(b->names); free(b->quals);
    memset(b, 0, sizeof(*b));
}

namespace {

struct FqRec { size_t name, name_len, seq, seq_len, qual; };

// One 4-line record at buf[p..e): returns the offset behind it, 0 if it is incomplete (needs more input), SIZE_MAX if it
// is not a plain 4-line record (the caller falls back to the general parser).  `final`: e is the end of the input, a
// missing last newline is fine.
inline size_t fq_record(const char *buf, size_t p, size_t e, bool final, FqRec *out) {
    if (buf[p] != '@') return SIZE_MAX;
    const char *l0 = (const char *) memchr(buf + p, '\n', e - p);
    if (!l0) return final ? SIZE_MAX : 0;
    const size_t s0 = (size_t) (l0 - buf) + 1;
    const char *l1 = s0 < e ? (const char *) memchr(buf + s0, '\n', e - s0) : nullptr;
    if (!l1) return final ? SIZE_MAX : 0;
    const size_t p0 = (size_t) (l1 - buf) + 1;
    if (p0 >= e) return final ? SIZE_MAX : 0;
    if (buf[p0] != '+') return SIZE_MAX;
    const char *l2 = (const char *) memchr(buf + p0, '\n', e - p0);
    if (!l2) return final ? SIZE_MAX : 0;
    const size_t q0 = (size_t) (l2 - buf) + 1, slen = p0 - 1 - s0;
    size_t q1;                                                       // end of the quality line
    if (q0 + slen < e) { if (buf[q0 + slen] != '\n') return SIZE_MAX; q1 = q0 + slen + 1; }
    else if (q0 + slen == e && final) q1 = e;
    else return final ? SIZE_MAX : 0;
    if ((slen && buf[s0 + slen - 1] == '\r') || buf[s0 - 2] == '\r') return SIZE_MAX;     // CR LF: general parser
    size_t nl = 0;
    while (p + 1 + nl < s0 - 1 && buf[p + 1 + nl] != ' ' && buf[p + 1 + nl] != '\t') ++nl;     // name ends at the first blank
    out->name = p + 1; out->name_len = nl; out->seq = s0; out->seq_len = slen; out->qual = q0;
    return q1;
}

// Indexes the complete 4-line records of buf[from..e) in parallel.  Returns the offset behind the last one (== from if
// none), or SIZE_MAX if the region is not 4-line FASTQ.
size_t fq_index(const char *buf, size_t from, size_t e, bool final, std::vector<FqRec> &recs) {
    const int T = lrm_host_threads();
    const size_t len = e - from;
    int nt = (int) (len / (1u << 20));
    nt = nt < 1 ? 1 : (nt > T ? T : nt);
    std::vector<size_t> start((size_t) nt + 1, e);
    start[0] = from;
    bool bad = false;
#pragma omp parallel for schedule(static, 1) num_threads(nt) reduction(|| : bad)
    for (int t = 1; t < nt; ++t) {
        // first record boundary at or after the cut: a line that starts with '@' whose line-after-next starts with '+'
        size_t c = from + len * (size_t) t / (size_t) nt;
        size_t found = e;
        for (int tries = 0; tries < 8 && c < e; ++tries) {
            const char *nl = (const char *) memchr(buf + c - 1, '\n', e - (c - 1));
            if (!nl) break;
            const size_t q = (size_t) (nl - buf) + 1;
            if (q >= e) break;
            if (buf[q] == '@') {
                const char *a = (const char *) memchr(buf + q, '\n', e - q);
                const char *b = a && (size_t) (a - buf) + 1 < e ? (const char *) memchr(a + 1, '\n', e - (size_t) (a + 1 - buf)) : nullptr;
                if (!b || (size_t) (b - buf) + 1 >= e) break;                 // runs out of the region: no boundary in this piece
                if (b[1] == '+') { found = q; break; }
            }
            c = q + 1;
        }
        start[(size_t) t] = found;
    }
    for (int t = 1; t < nt; ++t) if (start[(size_t) t] < start[(size_t) t - 1]) start[(size_t) t] = start[(size_t) t - 1];   // (pieces shorter than a record)
    std::vector<std::vector<FqRec>> part((size_t) nt);
    std::vector<size_t> stop((size_t) nt, 0);
#pragma omp parallel for schedule(static, 1) num_threads(nt) reduction(|| : bad)
    for (int t = 0; t < nt; ++t) {
        size_t p = start[(size_t) t];
        const size_t lim = start[(size_t) t + 1];
        const bool last_piece = lim == e;
        auto &v = part[(size_t) t];
        while (p < lim) {
            FqRec r;
            const size_t nx = fq_record(buf, p, e, final, &r);
            if (nx == SIZE_MAX) { bad = true; break; }
            if (nx == 0) { if (!last_piece) bad = true; break; }              // incomplete: only the tail of the region may be
            v.push_back(r);
            p = nx;
        }
        if (!bad && !last_piece && p != lim) bad = true;                      // the piece must end exactly where the next one starts
        stop[(size_t) t] = p;
    }
    if (bad) return SIZE_MAX;
    size_t behind = from;
    for (int t = 0; t < nt; ++t) {
        recs.insert(recs.end(), part[(size_t) t].begin(), part[(size_t) t].end());
        if (!part[(size_t) t].empty() || stop[(size_t) t] > behind) behind = stop[(size_t) t] > behind ? stop[(size_t) t] : behind;
    }
    return behind;
}

// the general parser (multi-line FASTA / FASTQ, CR LF): one record after the other
int64_t reader_next_slow(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out, char *seq_buf, uint64_t seq_cap) {
    std::vector<std::string> names, seqs, quals;
    std::vector<char> has_qual;
    int rc = 0;
    while (names.size() < batch_size) {                      // reads_load, accaln.c:45-58
        int c = r->last;
        if (c == 0) {
            while ((c = r->getc_()) != -1 && c != '>' && c != '@') {}
            if (c == -1) break;
        }
        r->last = 0;
        std::string header, seq, qual;
        r->line_(header);
        size_t sp = header.find_first_of(" \t");
        if (sp != std::string::npos) header.resize(sp);       // name ends at the first blank
        bool plus = false;
        while ((c = r->getc_()) != -1) {
            if (c == '>' || c == '@') { r->last = c; break; }
            if (c == '+') { plus = true; break; }
            if (c == '\n' || c == '\r') continue;
            seq.push_back((char) c);
            r->line_(seq);
        }
        if (plus) {
            std::string skip;
            r->line_(skip);
            while (qual.size() < seq.size()) { if (!r->line_(qual)) break; }
            if (qual.size() != seq.size()) { rc = -2; lrm_set_error("record %s: quality length differs from sequence length", header.c_str()); break; }
        }
        names.push_back(header); seqs.push_back(seq); quals.push_back(qual); has_qual.push_back(plus ? 1 : 0);
    }
    if (rc < 0) return rc;
    const uint64_t n = names.size();
    if (n == 0) return 0;
    uint32_t max_len = 0;                                     // refactor_reads_seq, alnmain.c:87-103
    for (auto &s : seqs) max_len = s.size() > max_len ? (uint32_t) s.size() : max_len;
    out->n = n; out->max_len = max_len; out->stride = (uint64_t) max_len + 1;
    if (seq_buf && n * out->stride <= seq_cap) { out->seqs = seq_buf; out->seqs_borrowed = 1; memset(seq_buf, 0, n * out->stride); }
    else out->seqs = (char *) calloc(n * out->stride, 1);
    out->lens = (uint32_t *) malloc(n * sizeof(uint32_t));
    out->names = (char **) calloc(n, sizeof(char *));
    out->quals = (char **) calloc(n, sizeof(char *));
    for (uint64_t i = 0; i < n; ++i) {
        memcpy(out->seqs + i * out->stride, seqs[i].data(), seqs[i].size());
        out->lens[i] = (uint32_t) seqs[i].size();
        out->names[i] = strdup(names[i].c_str());
        out->quals[i] = has_qual[i] ? strdup(quals[i].c_str()) : nullptr;
    }
    return (int64_t) n;
}

}  // namespace

static int64_t reader_next_impl(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out, void *seq_buf, uint64_t seq_cap);
extern "C" int64_t lrm_reader_next_into(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out, void *seq_buf, uint64_t seq_cap) {
    try { return reader_next_impl(r, batch_size, out, seq_buf, seq_cap); }
    catch (const std::exception &e) { lrm_set_error("reader: %s", e.what()); return -1; }
}
static int64_t reader_next_impl(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out, void *seq_buf, uint64_t seq_cap) {
    memset(out, 0, sizeof(*out));
    if (!r || batch_size == 0) return 0;
    if (!r->fast) return reader_next_slow(r, batch_size, out, (char *) seq_buf, seq_cap);
    // ---- fast parser: index at least batch_size records of the buffered block ----
    r->compact();
    std::vector<FqRec> recs;
    size_t scanned = 0;                                       // records of buf[0..scanned) are in `recs`
    for (;;) {
        if (r->end == scanned && r->eof) break;
        if (recs.size() >= batch_size) break;
        const size_t per = r->rec_bytes ? r->rec_bytes : 1024;
        size_t want = (size_t) ((batch_size - recs.size()) * (double) per * 1.05) + (1u << 20);
        if (want > (1ull << 32)) want = 1ull << 32;
        const size_t have = r->end - scanned;
        if (have < want && !r->eof) r->read_more(want - have);
        if (r->end == scanned) break;
        if (recs.empty() && scanned == 0 && r->buf[0] != '@') { r->fast = false; break; }      // FASTA or leading junk: general parser
        const size_t before = recs.size();
        const size_t behind = fq_index(r->buf.data(), scanned, r->end, r->eof, recs);
        if (behind == SIZE_MAX) { recs.resize(before); r->fast = false; break; }
        if (behind == scanned) {
            if (r->eof) { if (scanned < r->end) { r->fast = false; } break; }       // trailing bytes that are no record: let the general parser judge
            r->rec_bytes = (r->end - scanned) * 2;                                  // one record is longer than the block: read more
            continue;
        }
        scanned = behind;
        if (recs.size() > before) r->rec_bytes = (scanned) / recs.size() + 1;
    }
    if (recs.empty()) {
        if (!r->fast) return reader_next_slow(r, batch_size, out, (char *) seq_buf, seq_cap);
        return 0;
    }
    const uint64_t n = recs.size() < batch_size ? recs.size() : batch_size;
    const char *buf = r->buf.data();
    uint32_t max_len = 0;
    uint64_t name_bytes = 0, qual_bytes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (recs[i].seq_len > 0xffffffffull) { lrm_set_error("record longer than 2^32 bases"); return -1; }
        max_len = recs[i].seq_len > max_len ? (uint32_t) recs[i].seq_len : max_len;
        name_bytes += recs[i].name_len + 1;
        qual_bytes += recs[i].seq_len + 1;
    }
    out->n = n; out->max_len = max_len; out->stride = (uint64_t) max_len + 1;     // refactor_reads_seq, alnmain.c:87-103
    if (seq_buf && n * out->stride <= seq_cap) { out->seqs = (char *) seq_buf; out->seqs_borrowed = 1; }
    else out->seqs = (char *) bigmem::alloc(n * out->stride);
    out->lens = (uint32_t *) malloc(n * sizeof(uint32_t));
    out->names = (char **) malloc(n * sizeof(char *));
    out->quals = (char **) malloc(n * sizeof(char *));
    out->name_arena = (char *) bigmem::alloc(name_bytes ? name_bytes : 1);
    out->qual_arena = (char *) bigmem::alloc(qual_bytes ? qual_bytes : 1);
    if (!out->seqs || !out->lens || !out->names || !out->quals || !out->name_arena || !out->qual_arena) {
        if (!out->seqs_borrowed) bigmem::release(out->seqs);
        bigmem::release(out->name_arena); bigmem::release(out->qual_arena);
        free(out->lens); free(out->names); free(out->quals);
        memset(out, 0, sizeof(*out));
        lrm_set_error("out of memory");
        return -1;
    }
    {   // arena offsets (serial prefix sums), then the copies in parallel
        uint64_t no = 0, qo = 0;
        for (uint64_t i = 0; i < n; ++i) {
            out->names[i] = out->name_arena + no; no += recs[i].name_len + 1;
            out->quals[i] = out->qual_arena + qo; qo += recs[i].seq_len + 1;
        }
        const uint64_t stride = out->stride;
#pragma omp parallel for schedule(static) num_threads(lrm_host_threads())
        for (uint64_t i = 0; i < n; ++i) {
            const FqRec &q = recs[i];
            char *row = out->seqs + i * stride;
            memcpy(row, buf + q.seq, q.seq_len);
            memset(row + q.seq_len, 0, stride - q.seq_len);                       // NUL padded (alnmain.c:94 callocs)
            out->lens[i] = (uint32_t) q.seq_len;
            memcpy(out->names[i], buf + q.name, q.name_len); out->names[i][q.name_len] = 0;
            memcpy(out->quals[i], buf + q.qual, q.seq_len); out->quals[i][q.seq_len] = 0;
        }
    }
    // behind the last record taken: the start of the next one, or what was scanned
    r->pos = n < recs.size() ? recs[n].name - 1 : scanned;
    return (int64_t) n;
}

extern "C" int64_t lrm_reader_next(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out) {
    return lrm_reader_next_into(r, batch_size, out, nullptr, 0);
}

static inline int put_uint(char *dst, uint64_t v) {           // decimal text of v, returns its length (<= 20)
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char) ('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}

// class of an op byte for the run-length CIGAR: '=' and 'X' print as M, every other byte as itself
static inline char op_class(uint8_t o) { return (o == '=' || o == 'X') ? 'M' : (char) o; }

// run-length text of n_ops op bytes appended at dst (room for 11 bytes per run must be there: callers reserve
// 2 * n_ops + 16 -- a run of one column prints as two bytes, a longer one as fewer per column); returns the length.
// One pass, one compare per column, the decimal digits of a run written back to front into place.
static inline size_t rle_write(const uint8_t *ops, int n_ops, char *dst) {
    size_t w = 0;
    int i = 0;
    while (i < n_ops) {
        const char cls = op_class(ops[i]);
        int j = i + 1;
        if (cls == 'M') { while (j < n_ops && (ops[j] == '=' || ops[j] == 'X')) ++j; }
        else { const uint8_t o = ops[i]; while (j < n_ops && ops[j] == o) ++j; }
        uint32_t run = (uint32_t) (j - i);
        if (run < 10) { dst[w++] = (char) ('0' + run); }
        else if (run < 100) { dst[w++] = (char) ('0' + run / 10); dst[w++] = (char) ('0' + run % 10); }
        else w += (size_t) put_uint(dst + w, run);
        dst[w++] = cls;
        i = j;
    }
    return w;
}

extern "C" int lrm_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen) {
    if (n_ops <= 0) {
        if (buflen < 2) return -1;
        buf[0] = '*'; buf[1] = 0;
        return 1;
    }
    if (buflen >= 2 * n_ops + 16) {                            // room for the worst case: no checks inside the loop
        const size_t w = rle_write(ops, n_ops, buf);
        buf[w] = 0;
        return (int) w;
    }
    int w = 0, i = 0;
    while (i < n_ops) {
        const char o = op_class(ops[i]);
        int run = 0;
        if (o == 'M') { while (i < n_ops && (ops[i] == '=' || ops[i] == 'X')) { run++; i++; } }
        else { while (i < n_ops && ops[i] == (uint8_t) o) { run++; i++; } }
        if (buflen - w < 13) return -1;                        // <= 10 digits + op + NUL
        w += put_uint(buf + w, (uint64_t) run);
        buf[w++] = o;
    }
    buf[w] = 0;
    return w;
}

extern "C" void lrm_free(void *p) { free(p); }

static char *dup_out(const std::string &s, uint64_t *len_out) {
    char *p = (char *) malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    if (len_out) *len_out = s.size();
    return p;
}

extern "C" char *lrm_sam_header(const lrm_mta_entry *mta, int mta_len, long rg_id, uint64_t *len_out) {
    std::string s;
    char line[1200];
    for (int i = 0; i < mta_len; ++i) {                       // alnmain.c:66-72
        snprintf(line, sizeof(line), "@SQ\tSN:%.*s\tLN:%ld\n", (int) (mta[i].name_len < 1023 ? mta[i].name_len : 1023),
                 mta[i].name ? mta[i].name : "", (long) mta[i].seq_len);
        s += line;
    }
    snprintf(line, sizeof(line), "@RG\tID:%s%ld\tSM:SM_data\n", "accaln", rg_id);   // alnmain.c:73
    s += line;
    s += "@PG\tID:accaln\tPN:accaln\n";                                                // alnmain.c:74
    return dup_out(s, len_out);
}

// _rev_comp_in_place's base map (alnmain.c:29-52): ACGT of either case -> upper-case complement, anything else -> 'N'
static const struct CompTable { char t[256]; CompTable() { for (int c = 0; c < 256; ++c) t[c] = 'N'; t['A'] = t['a'] = 'T'; t['C'] = t['c'] = 'G'; t['G'] = t['g'] = 'C'; t['T'] = t['t'] = 'A'; } } k_comp;

// SAM lines of reads [lo, hi) appended to s (alnmain.c:500-525 field for field).  No snprintf on the hot path: a 10 kbp
// ONT read has ~2000 CIGAR runs.
// cigar_is_text: cig[i].cigar is the NUL-terminated run-length text already (lrm_map_options.cigar_text), not op bytes.
static void sam_format_range(const lrm_read_batch *reads, const lrm_mta_entry *mta, int mta_len, const lrm_cigar *cig,
                             const int *score, const lrm_seq_meta *meta, const int *meta_r, uint64_t lo, uint64_t hi,
                             std::string &s, bool cigar_is_text, bool revcomp_here) {
    uint64_t est = 0;
    for (uint64_t i = lo; i < hi; ++i) est += 2ull * reads->lens[i] + 2ull * (cig[i].n_cigar_op > 0 ? (uint64_t) cig[i].n_cigar_op : 0) + 160;
    s.clear();
    s.reserve(est);
    char num[32];
    for (uint64_t i = lo; i < hi; ++i) {
        const uint32_t len = reads->lens[i];
        const bool unmapped = meta_r[i] == 0 || score[i] == -1;      // alnmain.c:466-469
        int flag = 0, mapq = 255;
        if (unmapped) { flag += 0x4; mapq = 0; }
        else if (meta[i].strand == 1) flag += 16;
        s += reads->names[i];
        s += '\t';
        s.append(num, (size_t) put_uint(num, (uint64_t) flag));
        s += '\t';
        if (!unmapped && meta[i].seq_id >= 0 && meta[i].seq_id < mta_len) s.append(mta[meta[i].seq_id].name, mta[meta[i].seq_id].name_len);
        else s += '*';
        s += '\t';
        s.append(num, (size_t) put_uint(num, unmapped ? 0ull : (uint64_t) (meta[i].off + 1)));   // %ld of a non-negative value
        s += '\t';
        s.append(num, (size_t) put_uint(num, (uint64_t) mapq));
        s += '\t';
        if (!unmapped && cig[i].n_cigar_op > 0 && cigar_is_text) {
            s.append((const char *) cig[i].cigar);
        } else if (!unmapped && cig[i].n_cigar_op > 0) {
            const size_t at = s.size();
            s.resize(at + 2 * (size_t) cig[i].n_cigar_op + 16);            // alnmain.c:497: a 2 * qlen buffer there
            const size_t w = rle_write(cig[i].cigar, cig[i].n_cigar_op, &s[at]);
            s.resize(at + w);
        } else {
            s += '*';
        }
        s += "\t*\t0\t0\t";                                            // r_name "*", 0L, 0
        if (revcomp_here && meta_r[i] != 0 && meta[i].strand == 1) {
            // lrm_map_options.keep_reads: the batch came back as it went -- _rev_comp_in_place (alnmain.c:27-60) while copying
            const size_t at = s.size();
            s.resize(at + len);
            const char *src = reads->seqs + i * reads->stride;
            char *dst = &s[at];
            for (uint32_t x = 0; x < len; ++x) dst[x] = k_comp.t[(uint8_t) src[len - 1 - x]];
        } else {
            s.append(reads->seqs + i * reads->stride, len);           // the (possibly rev-comped) read
        }
        s += '\t';
        if (reads->quals[i]) s.append(reads->quals[i], len); else s += '*';
        s += "\tED:I:";
        if (score[i] < 0) { s += '-'; s.append(num, (size_t) put_uint(num, (uint64_t) (-(int64_t) score[i]))); }
        else s.append(num, (size_t) put_uint(num, (uint64_t) score[i]));
        s += '\n';
    }
}

// Every thread formats a contiguous range of reads into its own buffer.
static void sam_format_parts(const lrm_read_batch *reads, const lrm_mta_entry *mta, int mta_len, const lrm_cigar *cig,
                             const int *score, const lrm_seq_meta *meta, const int *meta_r, uint64_t n, int nt,
                             std::vector<std::string> &parts, bool cigar_is_text = false, bool revcomp_here = false) {
    if (nt < 1) nt = 1;
    if ((uint64_t) nt > n) nt = n ? (int) n : 1;
    parts.resize((size_t) nt);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; ++t)
        sam_format_range(reads, mta, mta_len, cig, score, meta, meta_r, n * (uint64_t) t / (uint64_t) nt, n * (uint64_t) (t + 1) / (uint64_t) nt,
                         parts[(size_t) t], cigar_is_text, revcomp_here);
}

extern "C" char *lrm_sam_format(const lrm_read_batch *reads, const lrm_mta_entry *mta, int mta_len,
                                const lrm_cigar *cig, const int *score, const lrm_seq_meta *meta,
                                const int *meta_r, uint64_t n, uint64_t *len_out) {
    std::vector<std::string> parts;
    sam_format_parts(reads, mta, mta_len, cig, score, meta, meta_r, n, lrm_host_threads(), parts);
    uint64_t total = 0;
    std::vector<uint64_t> at(parts.size() + 1, 0);
    for (size_t k = 0; k < parts.size(); ++k) { at[k + 1] = at[k] + parts[k].size(); total = at[k + 1]; }
    char *out = (char *) malloc(total + 1);
    if (!out) return nullptr;
#pragma omp parallel for schedule(static, 1) num_threads((int) parts.size())
    for (size_t k = 0; k < parts.size(); ++k) memcpy(out + at[k], parts[k].data(), parts[k].size());
    out[total] = 0;
    if (len_out) *len_out = total;
    return out;
}

namespace {

// bounded hand-off between two stages of the accaln pipeline
template <typename T>
struct StageQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap;
    bool closed = false;
    explicit StageQueue(size_t c) : cap(c) {}
    bool push(T &&v) {                                   // false: the queue was closed by the consumer (an error downstream)
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.size() < cap || closed; });
        if (closed) return false;
        q.push_back(std::move(v));
        cv.notify_all();
        return true;
    }
    bool pop(T &v) {                                     // false: closed and drained
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> lk(m); closed = true; cv.notify_all(); }
};

// one batch on its way through the pipeline, with the caller-side buffers of the hot path (pinned: the DMA engines
// read the reads and write the dense op bytes straight from / into them); recycled
struct BatchSet {
    lrm_read_batch b;
    char *reads_pin = nullptr; uint64_t reads_cap = 0;
    uint8_t *store_pin = nullptr; uint64_t store_cap = 0;
    std::atomic<bool> pin_ready{false};          // the pinner thread has given this set its pinned buffers
    uint8_t *store_pg = nullptr; uint64_t store_pg_cap = 0;      // pageable stand-in until then (or when a batch outgrows the pinned one)
    uint8_t *store = nullptr;
    std::vector<lrm_entry> best;
    std::vector<lrm_cigar> cig;
    std::vector<int> score, meta_r;
    std::vector<lrm_seq_meta> meta;
    uint64_t sstride = 0;
    lrm_ticket *ticket = nullptr;
    BatchSet() { memset(&b, 0, sizeof(b)); }
    ~BatchSet() { lrm_host_free(reads_pin); lrm_host_free(store_pin); free(store_pg); }
};

struct StageError {                                      // lrm_last_error() is thread-local: stages report through this
    std::mutex m;
    int rc = 0;
    std::string msg;
    void set(int code) {
        std::lock_guard<std::mutex> lk(m);
        if (!rc) { rc = code; msg = lrm_last_error(); }
    }
    int get() { std::lock_guard<std::mutex> lk(m); return rc; }
};

}  // namespace

// single_end() (alnmain.c:277-551) as a pipeline around the asynchronous batch calls: a loader thread parses batch k+2
// (parallel FASTQ parser, sequences straight into a pinned buffer), this thread keeps two batches in flight on the device
// (lrm_map_batch_submit / _wait, dense results DMA'd into pinned memory), and a writer thread formats batch k-1 in parallel
// and writes the parts of its SAM text with parallel pwrites.  Three sets of buffers go round.  The reference does the
// stages one after the other; at device mapping rates the text stages are the whole run time, so they have to overlap
// AND be parallel.  Output is identical: batches are written in input order.
extern "C" int lrm_accaln(const char *genome, const char *reads_path, const char *sam_path, lrm_params p,
                          lrm_gact_params gp, int device, long rg_id, uint64_t *total_out, uint64_t *valid_out) {
    const bool verbose = getenv("LRM_HOST_VERBOSE") != nullptr;            // stage times on stderr (tuning aid)
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_load = 0, t_map = 0, t_fmt = 0, t_write = 0;
    const double t_begin = now();
    lrm_host_index hi;
    if (lrm_host_index_read(genome, &hi)) return -1;                       // init(), alnmain.c:179-256
    const double t_read_idx = now();
    lrm_index *gpu = nullptr;
    lrm_index_options iopt;
    lrm_index_options_init(&iopt);
    {   // a reads file below 64 GiB (some tens of Gbp) does not repay the 0.7-2 s the 64 GiB seed table costs at upload
        struct stat st;
        if (stat(reads_path, &st) == 0 && (uint64_t) st.st_size < (64ull << 30)) iopt.lc_long_max = 15;
    }
    int rc = lrm_index_upload_opt(&gpu, &hi.fmi, &hi.lch, &hi.sa, hi.content, hi.con_len, hi.mta, hi.mta_len, &device, 1, &iopt);
    const double t_upload = now();
    int out_fd = -1;
    lrm_reader *rd = nullptr;
    uint64_t total = 0, valid = 0, out_off = 0;
    if (rc == 0) {
        out_fd = open(sam_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (out_fd < 0) { lrm_set_error("cannot create: %s", sam_path); rc = -1; }
    }
    if (rc == 0) {
        uint64_t hl = 0;
        char *h = lrm_sam_header(hi.mta, hi.mta_len, rg_id, &hl);
        if (pwrite(out_fd, h, hl, 0) != (ssize_t) hl) { lrm_set_error("cannot write: %s", sam_path); rc = -1; }
        out_off = hl;
        free(h);
        if (rc == 0) rc = lrm_reader_open(&rd, reads_path);
    }
    if (rc == 0) {
        constexpr int NSETS = 4;                                              // loading, two on the device, formatting
        std::vector<std::unique_ptr<BatchSet>> sets;
        StageQueue<BatchSet *> free_sets(NSETS), loaded(NSETS), mapped(NSETS);
        for (int k = 0; k < NSETS; ++k) { sets.emplace_back(new BatchSet); BatchSet *s = sets.back().get(); free_sets.push(std::move(s)); }
        StageError err;
        const uint64_t bs = p.batch_size ? p.batch_size : 1000;
        const int io_threads = lrm_host_threads() > 2 ? lrm_host_threads() / 2 : 1;     // loader and writer share the host
        lrm_map_options mopt;
        lrm_map_options_init(&mopt);
        mopt.cigar_text = 1;                        // parse_cigar (alnmain.c:497-498) runs on the device: the SAM CIGAR text comes back
        mopt.copy_threads = 2;                      // the parser and the formatter need the cores
        mopt.keep_reads = 1;                        // reverse-strand reads are reverse-complemented by the formatter as it copies them
        // Pinning the batch buffers (0.2 s per GB to pin and to release, and the device stalls while the runtime pins)
        // pays from a few tens of Gbp on: reads files below 16 GiB run through pageable buffers.
        bool want_pinned = false;
        { struct stat st; if (stat(reads_path, &st) == 0 && (uint64_t) st.st_size >= (16ull << 30)) want_pinned = true; }
        // Pinning memory costs ~0.2 s per GB: a thread of its own sizes the sets' pinned buffers from the first batch
        // (+ 1/8) while the first batches already run through pageable memory (staged upload, staged dense download).
        std::mutex dims_m;
        std::condition_variable dims_cv;
        uint64_t dim_reads = 0, dim_store = 0;
        bool dims_known = false, dims_stop = false, first_submitted = false;
        std::thread pinner([&]() {
            if (hipSetDevice(device) != hipSuccess) { (void) hipGetLastError(); }
            {
                std::unique_lock<std::mutex> lk(dims_m);
                // (after the first submit: that call builds the handle's host context -- streams, staging, mirrors -- and
                //  would queue behind the runtime's lock while gigabytes are being pinned here)
                dims_cv.wait(lk, [&] { return (dims_known && first_submitted) || dims_stop; });
                if (!dims_known || dims_stop || !want_pinned) return;
            }
            for (auto &sp : sets) {
                { std::lock_guard<std::mutex> lk(dims_m); if (dims_stop) return; }
                BatchSet *s = sp.get();
                s->reads_pin = (char *) lrm_host_alloc(dim_reads);
                s->store_pin = (uint8_t *) lrm_host_alloc(dim_store);
                if (!s->reads_pin || !s->store_pin) { lrm_host_free(s->reads_pin); lrm_host_free(s->store_pin); s->reads_pin = nullptr; s->store_pin = nullptr; return; }
                s->reads_cap = dim_reads; s->store_cap = dim_store;
                s->pin_ready.store(true, std::memory_order_release);
                if (verbose) fprintf(stderr, "[lrm accaln] %.3f pinned a set (%.2f + %.2f GB)\n", now() - t_upload, dim_reads / 1e9, dim_store / 1e9);
            }
        });
        std::thread loader([&]() {                                            // alnmain.c:302
            BatchSet *s = nullptr;
            while (!err.get() && free_sets.pop(s)) {
                const double t0 = now();
                const bool pinned = s->pin_ready.load(std::memory_order_acquire);
                int64_t n = lrm_reader_next_into(rd, bs, &s->b, pinned ? s->reads_pin : nullptr, pinned ? s->reads_cap : 0);
                t_load += now() - t0;
                if (verbose) fprintf(stderr, "[lrm accaln] %.3f loaded %lld reads in %.3f s (%s)\n", now() - t_upload, (long long) n, now() - t0, pinned ? "pinned" : "pageable");
                if (n < 0) { err.set(-1); break; }
                if (n == 0) { BatchSet *q = s; free_sets.push(std::move(q)); break; }
                if (!dims_known) {
                    std::lock_guard<std::mutex> lk(dims_m);
                    const uint64_t rb = s->b.n * s->b.stride, sb = s->b.n * ((((uint64_t) s->b.max_len * 2 + 15) & ~15ull) + 16);
                    dim_reads = rb + rb / 8 + 4096; dim_store = sb + sb / 8 + 4096;
                    dims_known = true;
                    dims_cv.notify_all();
                }
                if (!loaded.push(std::move(s))) break;
            }
            loaded.close();
        });
        // PART 3, alnmain.c:458-527, in two stages: the formatter turns a mapped batch into SAM text (one part per thread)
        // and gives the set back; the flusher writes the parts of the previous batch with parallel pwrites meanwhile.
        struct TextBatch { std::vector<std::string> parts; };
        StageQueue<std::unique_ptr<TextBatch>> texts(2), free_texts(3);
        for (int k = 0; k < 3; ++k) { std::unique_ptr<TextBatch> tb(new TextBatch); free_texts.push(std::move(tb)); }
        std::thread formatter([&]() {
            BatchSet *s = nullptr;
            while (mapped.pop(s)) {
                std::unique_ptr<TextBatch> tb;
                if (!err.get() && free_texts.pop(tb)) {
                    const uint64_t n = s->b.n;
                    const double t0 = now();
                    sam_format_parts(&s->b, hi.mta, hi.mta_len, s->cig.data(), s->score.data(), s->meta.data(), s->meta_r.data(), n,
                                     io_threads, tb->parts, /* cigar_is_text */ true, /* revcomp_here */ true);
                    t_fmt += now() - t0;
                    if (verbose) fprintf(stderr, "[lrm accaln] %.3f formatted %llu reads in %.3f s\n", now() - t_upload, (unsigned long long) n, now() - t0);
                    total += n;
                    for (uint64_t i = 0; i < n; ++i) valid += (s->score[i] >= 0 && s->meta_r[i] != 0) ? 1 : 0;   // alnmain.c:464-469,489-491
                    if (!texts.push(std::move(tb))) err.set(-1);
                }
                lrm_read_batch_free(&s->b);
                BatchSet *q = s;
                free_sets.push(std::move(q));
            }
            texts.close();
        });
        std::thread flusher([&]() {
            std::unique_ptr<TextBatch> tb;
            while (texts.pop(tb)) {
                if (!err.get()) {
                    const double t1 = now();
                    std::vector<std::string> &parts = tb->parts;
                    std::vector<uint64_t> at(parts.size() + 1, out_off);
                    for (size_t k = 0; k < parts.size(); ++k) at[k + 1] = at[k] + parts[k].size();
                    bool ok = true;
#pragma omp parallel for schedule(static, 1) num_threads((int) parts.size()) reduction(&& : ok)
                    for (size_t k = 0; k < parts.size(); ++k) {
                        size_t done = 0;
                        while (done < parts[k].size()) {
                            const ssize_t w = pwrite(out_fd, parts[k].data() + done, parts[k].size() - done, (off_t) (at[k] + done));
                            if (w <= 0) { ok = false; break; }
                            done += (size_t) w;
                        }
                    }
                    out_off = at[parts.size()];
                    if (!ok) { lrm_set_error("cannot write: %s", sam_path); err.set(-1); }
                    t_write += now() - t1;
                    if (verbose) fprintf(stderr, "[lrm accaln] %.3f wrote %.2f GB in %.3f s\n", now() - t_upload, (at[parts.size()] - at[0]) / 1e9, now() - t1);
                }
                if (!free_texts.push(std::move(tb))) break;
            }
            free_texts.close();
        });
        std::deque<BatchSet *> inflight;
        auto finish_oldest = [&]() {
            BatchSet *s = inflight.front();
            inflight.pop_front();
            const double t0 = now();
            const int mrc = lrm_map_batch_wait(s->ticket);
            t_map += now() - t0;
            if (verbose) fprintf(stderr, "[lrm accaln] %.3f waited %.3f s for a batch of %llu\n", now() - t_upload, now() - t0, (unsigned long long) s->b.n);
            s->ticket = nullptr;
            if (mrc) { err.set(-1); lrm_read_batch_free(&s->b); BatchSet *q = s; free_sets.push(std::move(q)); return; }
            if (!mapped.push(std::move(s))) { /* writer gone: error path */ }
        };
        BatchSet *s = nullptr;
        while (loaded.pop(s)) {
            if (err.get()) { lrm_read_batch_free(&s->b); BatchSet *q = s; free_sets.push(std::move(q)); continue; }   // drain what the loader already parsed
            const size_t n = (size_t) s->b.n;
            s->best.resize(n); s->cig.resize(n); s->score.resize(n); s->meta_r.resize(n); s->meta.resize(n);
            s->sstride = (((uint64_t) s->b.max_len * 2 + 15) & ~15ull) > 0 ? (((uint64_t) s->b.max_len * 2 + 15) & ~15ull) : 16;   // alnmain.c:316-320, a multiple of 16
            if (s->pin_ready.load(std::memory_order_acquire) && n * s->sstride <= s->store_cap) {
                s->store = s->store_pin;
            } else {
                if (n * s->sstride > s->store_pg_cap) {
                    free(s->store_pg);
                    s->store_pg_cap = n * s->sstride;
                    s->store_pg = (uint8_t *) malloc(s->store_pg_cap ? s->store_pg_cap : 1);
                    if (!s->store_pg) { s->store_pg_cap = 0; lrm_set_error("out of memory"); err.set(-1); lrm_read_batch_free(&s->b); BatchSet *q = s; free_sets.push(std::move(q)); continue; }
                }
                s->store = s->store_pg;
            }
            // PART 1 + PART 2 in one device pass, asynchronously: up to two batches on the device
            const double t0 = now();
            const int src = lrm_map_batch_submit(gpu, s->b.seqs, s->b.stride, s->b.lens, (uint64_t) n, p, gp, s->best.data(), s->cig.data(),
                                                 s->store, s->sstride, s->score.data(), s->meta.data(), s->meta_r.data(), &mopt, &s->ticket);
            t_map += now() - t0;
            if (verbose) fprintf(stderr, "[lrm accaln] %.3f submitted %zu reads (%s store) in %.3f s\n", now() - t_upload, n, s->store == s->store_pin ? "pinned" : "pageable", now() - t0);
            if (!first_submitted) { { std::lock_guard<std::mutex> lk(dims_m); first_submitted = true; } dims_cv.notify_all(); }
            if (src) { err.set(-1); lrm_read_batch_free(&s->b); BatchSet *q = s; free_sets.push(std::move(q)); continue; }
            inflight.push_back(s);
            if (inflight.size() >= 3) finish_oldest();                        // two on the device, one queued behind them
        }
        while (!inflight.empty()) finish_oldest();
        mapped.close();
        free_sets.close();
        loader.join();
        formatter.join();
        flusher.join();
        { std::lock_guard<std::mutex> lk(dims_m); dims_stop = true; }
        dims_cv.notify_all();
        pinner.join();
        rc = err.get();
        if (rc) lrm_set_error("%s", err.msg.c_str());
        const double t_done = now();
        if (gpu) { lrm_index_free(gpu); gpu = nullptr; }                      // before the pinned buffers of the sets go
        sets.clear();
        if (verbose) fprintf(stderr, "[lrm accaln] last batch written %.2f s after the upload; freeing the device image and the pinned sets %.2f s\n",
                             t_done - t_upload, now() - t_done);
    }
    if (rd) lrm_reader_close(rd);
    if (out_fd >= 0) close(out_fd);
    if (verbose)
        fprintf(stderr, "[lrm accaln] index files %.2f s, upload %.2f s, batches %.2f s wall (busy: loader %.2f, device waits %.2f, "
                        "formatter %.2f, write %.2f)\n", t_read_idx - t_begin, t_upload - t_read_idx, now() - t_upload, t_load, t_map,
                t_fmt, t_write);
    if (gpu) lrm_index_free(gpu);
    lrm_host_index_free(&hi);
    if (total_out) *total_out = total;
    if (valid_out) *valid_out = valid;
    return rc;
}
