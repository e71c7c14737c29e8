// io_host.cpp -- FASTA/FASTQ batches, SAM text, and the accaln flow on the GPU path
// (include/lrm_io_host.h).  Host-side C++; the kernels are reached only through the C-ABI.
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    // appends the rest of the current line (without the newline) to s; returns false at EOF before any byte
    bool line_(std::string &s) {
        int c;
        bool any = false;
        while ((c = getc_()) != -1) {
            any = true;
            if (c == '\n') break;
            if (c != '\r') s.push_back((char) c);
        }
        return any;
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

extern "C" int lrm_parse_cigar(const uint8_t *ops, int n_ops, char *buf, int buflen) {
    if (n_ops <= 0) {
        if (buflen < 2) return -1;
        buf[0] = '*'; buf[1] = 0;
        return 1;
    }
    int w = 0, i = 0;
    while (i < n_ops) {
        char o = (ops[i] == '=' || ops[i] == 'X') ? 'M' : (char) ops[i];
        int run = 0;
        while (i < n_ops && (((ops[i] == '=' || ops[i] == 'X') ? 'M' : (char) ops[i]) == o)) { run++; i++; }
        int r = snprintf(buf + w, (size_t) (buflen - w), "%d%c", run, o);
        if (r < 0 || r >= buflen - w) return -1;
        w += r;
    }
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
    const int nt = omp_get_max_threads();
    std::vector<std::string> parts((size_t) nt);
    std::vector<uint64_t> first((size_t) nt + 1, 0);
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const uint64_t lo = n * (uint64_t) t / (uint64_t) nt, hi = n * (uint64_t) (t + 1) / (uint64_t) nt;
        std::string &s = parts[(size_t) t];
        std::vector<char> cbuf;
        char num[64];
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t len = reads->lens[i];
            const bool unmapped = meta_r[i] == 0 || score[i] == -1;      // alnmain.c:466-469
            int flag = 0, mapq = 255;
            if (unmapped) { flag += 0x4; mapq = 0; }
            else if (meta[i].strand == 1) flag += 16;
            cbuf.resize((size_t) 2 * len + 16);                            // alnmain.c:497
            const char *cg = "*";
            if (!unmapped) {
                lrm_parse_cigar(cig[i].cigar, cig[i].n_cigar_op, cbuf.data(), (int) cbuf.size());
                cg = cbuf.data();
            }
            s += reads->names[i];
            snprintf(num, sizeof(num), "\t%d\t", flag); s += num;
            if (!unmapped && meta[i].seq_id >= 0 && meta[i].seq_id < mta_len) s.append(mta[meta[i].seq_id].name, mta[meta[i].seq_id].name_len);
            else s += "*";
            snprintf(num, sizeof(num), "\t%ld\t%d\t", unmapped ? 0L : (long) (meta[i].off + 1), mapq); s += num;
            s += cg;
            s += "\t*\t0\t0\t";                                            // r_name "*", 0L, 0
            s.append(reads->seqs + i * reads->stride, len);               // the (possibly rev-comped) read
            s += "\t";
            if (reads->quals[i]) s.append(reads->quals[i], len); else s += "*";
            snprintf(num, sizeof(num), "\tED:I:%d\n", score[i]); s += num;
        }
    }
    std::string all;
    size_t tot = 0;
    for (auto &p : parts) tot += p.size();
    all.reserve(tot);
    for (auto &p : parts) all += p;
    return dup_out(all, len_out);
}

extern "C" int lrm_accaln(const char *genome, const char *reads_path, const char *sam_path, lrm_params p,
                          lrm_gact_params gp, int device, long rg_id, uint64_t *total_out, uint64_t *valid_out) {
    lrm_host_index hi;
    if (lrm_host_index_read(genome, &hi)) return -1;                       // init(), alnmain.c:179-256
    lrm_index *gpu = nullptr;
    int rc = lrm_index_upload(&gpu, &hi.fmi, &hi.lch, &hi.sa, hi.content, hi.con_len, hi.mta, hi.mta_len, device);
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
    while (rc == 0) {                                                         // alnmain.c:302
        lrm_read_batch b;
        int64_t n = lrm_reader_next(rd, p.batch_size ? p.batch_size : 1000, &b);
        if (n < 0) { rc = -1; break; }
        if (n == 0) break;
        std::vector<lrm_entry> best((size_t) n);
        std::vector<lrm_cigar> cig((size_t) n);
        std::vector<int> score((size_t) n), meta_r((size_t) n);
        std::vector<lrm_seq_meta> meta((size_t) n);
        const uint64_t sstride = (uint64_t) b.max_len * 2 > 0 ? (uint64_t) b.max_len * 2 : 1;   // alnmain.c:316-320
        std::vector<uint8_t> store((size_t) n * sstride);
        // PART 1 + PART 2 in one device pass (the reads cross the link once)
        rc = lrm_map_batch(gpu, b.seqs, b.stride, b.lens, (uint64_t) n, p, gp, best.data(), cig.data(), store.data(),
                           sstride, score.data(), meta.data(), meta_r.data());
        if (rc == 0) {
            uint64_t tl = 0;
            char *txt = lrm_sam_format(&b, hi.mta, hi.mta_len, cig.data(), score.data(), meta.data(), meta_r.data(),
                                       (uint64_t) n, &tl);
            fwrite(txt, 1, tl, out);
            free(txt);
            total += (uint64_t) n;
            for (int64_t i = 0; i < n; ++i) valid += (score[(size_t) i] >= 0 && meta_r[(size_t) i] != 0) ? 1 : 0;   // alnmain.c:464-469,489-491
        }
        lrm_read_batch_free(&b);
    }
    if (rd) lrm_reader_close(rd);
    if (out) fclose(out);
    if (gpu) lrm_index_free(gpu);
    lrm_host_index_free(&hi);
    if (total_out) *total_out = total;
    if (valid_out) *valid_out = valid;
    return rc;
}
