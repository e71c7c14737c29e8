// io_host.cpp -- FASTA/FASTQ batches, SAM text, and the accaln flow on the GPU path
// (include/lrm_io_host.h).  Host-side C++; the kernels are reached only through the C-ABI.
#include <zlib.h>
#include <sys/stat.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <omp.h>
#include "../../include/lrm_io_host.h"
#include "../../include/lrm_index_host.h"
#include "lrm_internal.h"

struct lrm_reader {
    gzFile fp;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    int last = 0;            // header character already consumed ('>' or '@'), 0 = none
    bool eof = false;
    int getc_() {
        if (pos == end) {
            if (eof) return -1;
            int n = gzread(fp, buf.data(), (unsigned) buf.size());
            if (n <= 0) { eof = true; return -1; }
            pos = 0; end = (size_t) n;
        }
        return (unsigned char) buf[pos++];
    }
    // appends the rest of the current line (without the newline) to s; returns false at EOF before any byte.
    // Whole buffer spans at a time (memchr + one append): a 10 kbp sequence line is one or two appends.
    bool line_(std::string &s) {
        bool any = false;
        while (true) {
            if (pos == end) {
                if (eof) return any;
                int n = gzread(fp, buf.data(), (unsigned) buf.size());
                if (n <= 0) { eof = true; return any; }
                pos = 0; end = (size_t) n;
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
    gzFile fp = gzopen(path, "rb");
    if (!fp) { lrm_set_error("cannot open: %s", path); return -1; }
    lrm_reader *r = new lrm_reader;
    r->fp = fp;
    r->buf.resize(1 << 20);
    *out = r;
    return 0;
}

extern "C" void lrm_reader_close(lrm_reader *r) {
    if (!r) return;
    gzclose(r->fp);
    delete r;
}

extern "C" void lrm_read_batch_free(lrm_read_batch *b) {
    if (!b) return;
    free(b->seqs); free(b->lens);
    for (uint64_t i = 0; i < b->n; ++i) { free(b->names ? b->names[i] : nullptr); free(b->quals ? b->quals[i] : nullptr); }
    free(b->names); free(b->quals);
    memset(b, 0, sizeof(*b));
}

extern "C" int64_t lrm_reader_next(lrm_reader *r, uint64_t batch_size, lrm_read_batch *out) {
    memset(out, 0, sizeof(*out));
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
    out->seqs = (char *) calloc(n * out->stride, 1);
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

static inline int put_uint(char *dst, uint64_t v) {           // decimal text of v, returns its length (<= 20)
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char) ('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}

extern "C" int lrm_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen) {
    if (n_ops <= 0) {
        if (buflen < 2) return -1;
        buf[0] = '*'; buf[1] = 0;
        return 1;
    }
    int w = 0, i = 0;
    while (i < n_ops) {
        const char o = (ops[i] == '=' || ops[i] == 'X') ? 'M' : (char) ops[i];
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

extern "C" char *lrm_sam_format(const lrm_read_batch *reads, const lrm_mta_entry *mta, int mta_len,
                                const lrm_cigar *cig, const int *score, const lrm_seq_meta *meta,
                                const int *meta_r, uint64_t n, uint64_t *len_out) {
    // Every thread formats a contiguous range of reads into its own buffer (no snprintf on the hot path: a 10 kbp ONT
    // read has ~2000 CIGAR runs), then the parts are copied to their offsets of one allocation in parallel.
    const int nt = lrm_host_threads();
    std::vector<std::string> parts((size_t) nt);
    std::vector<uint64_t> at((size_t) nt + 1, 0);
    char *out = nullptr;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(), team = omp_get_num_threads();
        const uint64_t lo = n * (uint64_t) t / (uint64_t) team, hi = n * (uint64_t) (t + 1) / (uint64_t) team;
        std::string &s = parts[(size_t) t];
        uint64_t est = 0;
        for (uint64_t i = lo; i < hi; ++i) est += 3ull * reads->lens[i] + 128;
        s.reserve(est);
        std::vector<char> cbuf;
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
            if (!unmapped) {
                cbuf.resize((size_t) 2 * len + 16);                        // alnmain.c:497
                const int cl = lrm_parse_cigar(cig[i].cigar, cig[i].n_cigar_op, cbuf.data(), (int) cbuf.size());
                if (cl > 0) s.append(cbuf.data(), (size_t) cl);
            } else {
                s += '*';
            }
            s += "\t*\t0\t0\t";                                            // r_name "*", 0L, 0
            s.append(reads->seqs + i * reads->stride, len);               // the (possibly rev-comped) read
            s += '\t';
            if (reads->quals[i]) s.append(reads->quals[i], len); else s += '*';
            s += "\tED:I:";
            if (score[i] < 0) { s += '-'; s.append(num, (size_t) put_uint(num, (uint64_t) (-(int64_t) score[i]))); }
            else s.append(num, (size_t) put_uint(num, (uint64_t) score[i]));
            s += '\n';
        }
#pragma omp barrier
#pragma omp single
        {
            for (int k = 0; k < team; ++k) at[(size_t) k + 1] = at[(size_t) k] + parts[(size_t) k].size();
            out = (char *) malloc(at[(size_t) team] + 1);
            if (out) out[at[(size_t) team]] = 0;
            if (len_out) *len_out = at[(size_t) team];
        }
        if (out) memcpy(out + at[(size_t) t], s.data(), s.size());
    }
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

struct MappedBatch {
    lrm_read_batch b;
    std::vector<lrm_cigar> cig;
    std::vector<int> score, meta_r;
    std::vector<lrm_seq_meta> meta;
    std::vector<uint8_t> store;
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

// single_end() (alnmain.c:277-551) as a three-stage pipeline: a loader thread parses batch k+1 and a writer thread
// formats and writes batch k-1 while this thread maps batch k on the device.  The reference does the three one after
// the other; at device mapping rates the text stages are the whole run time (tools/io_probe.py: loader 0.4 Gbp/s,
// formatter 0.7 Gbp/s on 8 cores, mapping 14 Gbp/s), so they must at least overlap.  Output is identical: batches
// are written in input order.
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
    FILE *out = nullptr;
    lrm_reader *rd = nullptr;
    uint64_t total = 0, valid = 0;
    if (rc == 0) {
        out = fopen(sam_path, "wb");
        if (!out) { lrm_set_error("cannot create: %s", sam_path); rc = -1; }
    }
    if (rc == 0) {
        uint64_t hl = 0;
        char *h = lrm_sam_header(hi.mta, hi.mta_len, rg_id, &hl);
        fwrite(h, 1, hl, out);
        free(h);
        rc = lrm_reader_open(&rd, reads_path);
    }
    if (rc == 0) {
        StageQueue<lrm_read_batch> loaded(2);
        StageQueue<std::unique_ptr<MappedBatch>> mapped(2);
        StageError err;
        const uint64_t bs = p.batch_size ? p.batch_size : 1000;
        std::thread loader([&]() {                                            // alnmain.c:302
            while (!err.get()) {
                lrm_read_batch b;
                const double t0 = now();
                const int64_t n = lrm_reader_next(rd, bs, &b);
                t_load += now() - t0;
                if (n < 0) { err.set(-1); break; }
                if (n == 0) break;
                if (!loaded.push(std::move(b))) { lrm_read_batch_free(&b); break; }
            }
            loaded.close();
        });
        std::thread writer([&]() {                                            // PART 3, alnmain.c:458-527
            std::unique_ptr<MappedBatch> mb;
            while (mapped.pop(mb)) {
                if (!err.get()) {
                    const uint64_t n = mb->b.n;
                    uint64_t tl = 0;
                    const double t0 = now();
                    char *txt = lrm_sam_format(&mb->b, hi.mta, hi.mta_len, mb->cig.data(), mb->score.data(), mb->meta.data(),
                                               mb->meta_r.data(), n, &tl);
                    const double t1 = now();
                    if (!txt || fwrite(txt, 1, tl, out) != tl) { lrm_set_error("cannot write: %s", sam_path); err.set(-1); }
                    t_fmt += t1 - t0; t_write += now() - t1;
                    free(txt);
                    total += n;
                    for (uint64_t i = 0; i < n; ++i) valid += (mb->score[i] >= 0 && mb->meta_r[i] != 0) ? 1 : 0;   // alnmain.c:464-469,489-491
                }
                lrm_read_batch_free(&mb->b);
            }
        });
        lrm_read_batch b;
        while (loaded.pop(b)) {
            if (err.get()) { lrm_read_batch_free(&b); continue; }             // drain what the loader already parsed
            const size_t n = (size_t) b.n;
            std::unique_ptr<MappedBatch> mb(new MappedBatch);
            mb->b = b;
            std::vector<lrm_entry> best(n);
            mb->cig.resize(n); mb->score.resize(n); mb->meta_r.resize(n); mb->meta.resize(n);
            const uint64_t sstride = (uint64_t) b.max_len * 2 > 0 ? (uint64_t) b.max_len * 2 : 1;   // alnmain.c:316-320
            mb->store.resize(n * sstride);
            // PART 1 + PART 2 in one device pass (the reads cross the link once)
            const double t0 = now();
            const int mrc = lrm_map_batch(gpu, b.seqs, b.stride, b.lens, (uint64_t) n, p, gp, best.data(), mb->cig.data(),
                                          mb->store.data(), sstride, mb->score.data(), mb->meta.data(), mb->meta_r.data());
            t_map += now() - t0;
            if (mrc) {
                err.set(-1);
                lrm_read_batch_free(&mb->b);
                continue;
            }
            if (!mapped.push(std::move(mb))) break;
        }
        mapped.close();
        loader.join();
        writer.join();
        rc = err.get();
        if (rc) lrm_set_error("%s", err.msg.c_str());
    }
    if (rd) lrm_reader_close(rd);
    if (out) fclose(out);
    if (verbose)
        fprintf(stderr, "[lrm accaln] index files %.2f s, upload %.2f s, batches %.2f s wall (busy: loader %.2f, device %.2f, "
                        "formatter %.2f, write %.2f)\n", t_read_idx - t_begin, t_upload - t_read_idx, now() - t_upload, t_load, t_map,
                t_fmt, t_write);
    if (gpu) lrm_index_free(gpu);
    lrm_host_index_free(&hi);
    if (total_out) *total_out = total;
    if (valid_out) *valid_out = valid;
    return rc;
}
