"""The steps either side of the hot path: FASTA/FASTQ batch loader, run-length
CIGAR, SAM header and records -- CPU only, against a Python restatement of alnmain.c's formatting."""
import ctypes as C
import gzip

import numpy as np
import pytest

import orc
import sam_ref
from longreadmapper_amd import capi, mapper
from longreadmapper_amd.capi import lib


def _read_all(path, batch):
    rd = C.c_void_p()
    capi.check(lib.lrm_reader_open(C.byref(rd), str(path).encode()), "open")
    out = []
    while True:
        b = capi.ReadBatch()
        n = lib.lrm_reader_next(rd, batch, C.byref(b))
        assert n >= 0, lib.lrm_last_error()
        if n == 0:
            break
        assert b.stride == b.max_len + 1
        raw = np.ctypeslib.as_array(C.cast(b.seqs, C.POINTER(C.c_uint8)), shape=(n, b.stride)).copy()
        for i in range(n):
            ln = b.lens[i]
            assert not raw[i, ln:].any()                                   # NUL padded (calloc, alnmain.c:94)
            out.append((b.names[i].decode(), bytes(raw[i, :ln]), b.quals[i], n))
        lib.lrm_read_batch_free(C.byref(b))
    lib.lrm_reader_close(rd)
    return out


def test_fastq_fasta_reader(tmp_path):
    recs = [("r1", b"ACGTACGT", b"IIIIHHHH"), ("r2", b"A" * 130, b"#" * 130), ("r3", b"", b""), ("r4", b"GATTACA", b"@+@+@+@")]
    txt = b""
    for i, (nm, s, q) in enumerate(recs):
        body = s if i != 1 else s[:60] + b"\n" + s[60:120] + b"\n" + s[120:]    # multi-line record
        qb = q if i != 1 else q[:70] + b"\n" + q[70:]
        txt += b"@" + nm.encode() + (b" comment here" if i == 0 else b"") + b"\n" + body + b"\n+\n" + qb + b"\n"
    p = tmp_path / "r.fq"
    p.write_bytes(txt)
    got = _read_all(p, 3)
    assert [(g[0], g[1], g[2]) for g in got] == [(n, s, q) for n, s, q in recs]
    assert [g[3] for g in got] == [3, 3, 3, 1]                               # batches of 3 then 1
    gz = tmp_path / "r.fq.gz"
    gz.write_bytes(gzip.compress(txt))
    assert [(g[0], g[1], g[2]) for g in _read_all(gz, 10)] == [(n, s, q) for n, s, q in recs]
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">a desc\nACGT\nAC\n>b\nTTTT\n")
    assert [(g[0], g[1], g[2]) for g in _read_all(fa, 10)] == [("a", b"ACGTAC", None), ("b", b"TTTT", None)]
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@x\nACGT\n+\nII\n")
    rd = C.c_void_p()
    capi.check(lib.lrm_reader_open(C.byref(rd), str(bad).encode()))
    b = capi.ReadBatch()
    assert lib.lrm_reader_next(rd, 4, C.byref(b)) == -2                      # kseq: -2 truncated quality
    lib.lrm_reader_close(rd)


@pytest.mark.parametrize("batch", [1, 7, 1000, 100000])
@pytest.mark.parametrize("trailing_newline", [True, False])
def test_parallel_fastq_parser(tmp_path, batch, trailing_newline):
    """4-line FASTQ goes through the parallel parser: the block is cut at arbitrary byte offsets, every piece
    resynchronises to a record boundary (quality lines that start with '@' or '+' must not fool it), batches end at
    any record.  Compared with a line-by-line parse; also gzip-compressed, and with the sequences written into a
    caller's buffer (lrm_reader_next_into)."""
    rng = np.random.default_rng(7)
    recs = []
    for i in range(3000):
        ln = int(rng.choice([0, 1, 2, 50, 700, 3000, 9000]))
        seq = bytes(rng.choice(list(b"ACGTN"), size=ln).astype(np.uint8))
        qual = bytes(rng.choice(list(b"@+#I5!~"), size=ln).astype(np.uint8))
        if ln and i % 3 == 0:
            qual = b"@" + qual[1:]
        if ln and i % 5 == 0:
            qual = b"+" + qual[1:]
        name = b"read_%d" % i + (b" a comment @+ with\ttabs" if i % 4 == 0 else b"")
        recs.append((name, seq, qual))
    txt = b"".join(b"@" + n + b"\n" + s + b"\n+" + (n if i % 7 == 0 else b"") + b"\n" + q + b"\n" for i, (n, s, q) in enumerate(recs))
    if not trailing_newline:
        txt = txt[:-1]
    assert len(txt) > 6_000_000                                    # several pieces of >= 1 MiB
    p = tmp_path / "big.fq"
    p.write_bytes(txt)
    want = [(n.split(b" ")[0].split(b"\t")[0].decode(), s, q) for n, s, q in recs]
    got = _read_all(p, batch)
    assert [(g[0], g[1], g[2]) for g in got] == want
    if batch == 1000:
        assert [g[3] for g in got[::1000]] == [1000, 1000, 1000]
        gz = tmp_path / "big.fq.gz"
        gz.write_bytes(gzip.compress(txt, 1))
        assert [(g[0], g[1], g[2]) for g in _read_all(gz, batch)] == want
        # sequences into a caller's buffer, too small for the second call
        rd = C.c_void_p()
        capi.check(lib.lrm_reader_open(C.byref(rd), str(p).encode()))
        mine = np.full(1000 * 9001, 0x55, dtype=np.uint8)
        b = capi.ReadBatch()
        assert lib.lrm_reader_next_into(rd, 1000, C.byref(b), mine.ctypes.data, mine.nbytes) == 1000
        assert b.seqs_borrowed == 1 and b.seqs == mine.ctypes.data
        rows = mine[:1000 * b.stride].reshape(1000, b.stride)
        for i in (0, 1, 17, 999):
            assert bytes(rows[i, :b.lens[i]]) == recs[i][1] and not rows[i, b.lens[i]:].any()
        lib.lrm_read_batch_free(C.byref(b))
        assert lib.lrm_reader_next_into(rd, 1000, C.byref(b), mine.ctypes.data, 10) == 1000
        assert b.seqs_borrowed == 0 and b.seqs != mine.ctypes.data
        lib.lrm_read_batch_free(C.byref(b))
        lib.lrm_reader_close(rd)


def test_parser_falls_back_in_the_middle_of_a_file(tmp_path):
    """4-line records first (parallel parser), then a multi-line record and CR LF lines: the general parser takes over
    at that record; nothing is lost or duplicated."""
    a = b"".join(b"@a%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(50))
    b = b"@multi\nACGT\nACGT\n+\nIIII\nIIII\n@crlf\r\nGGCC\r\n+\r\n!!!!\r\n@z\nTT\n+\n##\n"
    p = tmp_path / "mixed.fq"
    p.write_bytes(a + b)
    got = _read_all(p, 20)
    assert [g[0] for g in got] == ["a%d" % i for i in range(50)] + ["multi", "crlf", "z"]
    assert got[50][1] == b"ACGTACGT" and got[50][2] == b"IIIIIIII" and got[51][1] == b"GGCC" and got[51][2] == b"!!!!"
    assert got[52][1] == b"TT"


def test_parse_cigar_rle():
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = int(rng.integers(0, 300))
        ops = bytes(rng.choice(list(b"====XID"), size=n).astype(np.uint8))
        buf = C.create_string_buffer(2 * n + 16)
        arr = np.frombuffer(ops, dtype=np.uint8) if n else np.zeros(1, dtype=np.uint8)
        ln = lib.lrm_parse_cigar(arr.ctypes.data, n, buf, len(buf))
        assert buf.value.decode() == sam_ref.rle(ops) == orc.parse_cigar(ops) and ln == len(buf.value)
    buf = C.create_string_buffer(3)
    arr = np.frombuffer(b"=" * 100 + b"I", dtype=np.uint8)
    assert lib.lrm_parse_cigar(arr.ctypes.data, 101, buf, 3) < 0             # "100M1I" does not fit


def test_sam_header_and_records(tmp_path):
    names = [b"chrA", b"contig_two"]
    mta = (capi.MtaEntry * 2)()
    for i, (nm, off, ln) in enumerate(zip(names, (0, 2000), (1000, 321))):
        mta[i].name_len, mta[i].name, mta[i].offset, mta[i].seq_len = len(nm), nm, off, ln
    ln_out = C.c_uint64()
    h = lib.lrm_sam_header(mta, 2, 1234567, C.byref(ln_out))
    txt = C.string_at(h, ln_out.value).decode()
    lib.lrm_free(h)
    pymta = [("chrA", 0, 1000), ("contig_two", 2000, 321)]
    assert txt == sam_ref.header(pymta, 1234567)

    p = tmp_path / "r.fq"
    p.write_bytes(b"@q0\nACGTAC\n+\nIIIIII\n@q1\nGGGG\n+\n####\n@q2\nTT\n+\n!!\n@q3\nACG\n+\n;;;\n")
    rd = C.c_void_p()
    capi.check(lib.lrm_reader_open(C.byref(rd), str(p).encode()))
    b = capi.ReadBatch()
    assert lib.lrm_reader_next(rd, 10, C.byref(b)) == 4
    ops = [b"==X=I=", b"=D==="[:5], b"", b"==="]
    store = np.zeros((4, 12), dtype=np.uint8)
    cig = (capi.Cigar * 4)()
    for i, o in enumerate(ops):
        store[i, :len(o)] = np.frombuffer(o, dtype=np.uint8) if o else []
        cig[i].cigar = C.cast(store[i].ctypes.data, capi.u8p)
        cig[i].n_cigar_op = len(o)
    score = np.array([2, 1, -1, 0], dtype=np.int32)
    meta_r = np.array([1, 1, 1, 0], dtype=np.int32)
    meta = np.zeros(4, dtype=mapper.META_DT)
    meta["seq_id"] = [0, 1, 0, -1]
    meta["off"] = [41, 7, 3, 0]
    meta["strand"] = [0, 1, 0, 0]
    for i, sc in enumerate(score):
        cig[i].score = int(sc)
    t = lib.lrm_sam_format(C.byref(b), mta, 2, C.cast(cig, C.c_void_p), score.ctypes.data, meta.ctypes.data,
                           meta_r.ctypes.data, 4, C.byref(ln_out))
    got = C.string_at(t, ln_out.value).decode()
    lib.lrm_free(t)
    seqs = ["ACGTAC", "GGGG", "TT", "ACG"]
    quals = ["IIIIII", "####", "!!", ";;;"]
    want = "".join(sam_ref.record("q%d" % i, seqs[i], quals[i], pymta, ops[i], int(score[i]), int(meta_r[i]),
                                  int(meta["seq_id"][i]), int(meta["off"][i]), int(meta["strand"][i])) for i in range(4))
    assert got == want
    assert got.splitlines()[0] == "q0\t0\tchrA\t42\t255\t4M1I1M\t*\t0\t0\tACGTAC\tIIIIII\tED:I:2"
    assert got.splitlines()[2].split("\t")[1:6] == ["4", "*", "0", "0", "*"]
    lib.lrm_read_batch_free(C.byref(b))
    lib.lrm_reader_close(rd)
