"""End to end on the GPU path: `accidx ref.fa` + `accaln ref.fa reads.fq` (lrm_accidx / lrm_accaln, the
single_end() flow of alnmain.c:277-551) against SAM text assembled from the CPU oracle's results."""
import ctypes as C

import numpy as np
import pytest

import orc
import sam_ref
from longreadmapper_amd import capi, index, synth
from longreadmapper_amd.capi import lib

pytestmark = pytest.mark.gpu


def test_accaln_sam_matches_oracle(gpu, tmp_path):
    seqs = [synth.reference(90_000, seed=31), synth.reference(40_000, seed=32)]
    fa = tmp_path / "ref.fa"
    with open(fa, "wb") as f:
        for nm, s in zip((b"chr1 primary", b"chr2"), seqs):
            f.write(b">" + nm + b"\n")
            b = bytes(s)
            for i in range(0, len(b), 60):
                f.write(b[i:i + 60] + b"\n")
    assert lib.lrm_accidx(str(fa).encode(), 32, 10, 1) == 0
    r = synth.reads(seqs, 150, 1200, synth.ONT, seed=5)
    lens = r["lens"].copy()
    lens[::7] = 300
    lens[3] = 15                                   # shorter than a seed
    fq = tmp_path / "reads.fq"
    with open(fq, "wb") as f:
        for i in range(len(lens)):
            s = bytes(r["reads"][i, :lens[i]])
            f.write(b"@read%d extra\n" % i + s + b"\n+\n" + bytes([33 + (i + j) % 40 for j in range(len(s))]) + b"\n")
    sam = tmp_path / "out.sam"
    total, valid = C.c_uint64(), C.c_uint64()
    p = capi.Params(64, 20, 300)                   # batch of 64: three batches with different max_len
    capi.check(lib.lrm_accaln(str(fa).encode(), str(fq).encode(), str(sam).encode(), p, capi.GactParams(0, 0, 0), gpu,
                              424242, C.byref(total), C.byref(valid)), "lrm_accaln")
    got = open(sam).read()

    hi = index.HostIndex.read(str(fa))
    oi = orc.OracleIndex.from_host_index(hi)
    mta = hi.mta()
    assert [m[0] for m in mta] == ["chr1", "chr2"]
    want = sam_ref.header(mta, 424242)
    n_valid = 0
    for lo in range(0, len(lens), 64):             # the oracle per batch, exactly like the host loop
        bl = lens[lo:lo + 64]
        ml = int(bl.max())
        reads = np.zeros((len(bl), ml + 1), dtype=np.uint8)
        for i, l in enumerate(bl):
            reads[i, :l] = r["reads"][lo + i, :l]
        best, _ = oi.seed_batch(reads, bl)
        ext = oi.extend_batch(reads, bl, best)
        for i, l in enumerate(bl):
            k = int(ext["n_ops"][i])
            qual = "".join(chr(33 + (lo + i + j) % 40) for j in range(l))
            want += sam_ref.record("read%d" % (lo + i), bytes(reads[i, :l]).decode(), qual, mta, bytes(ext["ops"][i, :k]),
                                   int(ext["score"][i]), int(ext["meta_r"][i]), int(ext["meta"]["seq_id"][i]),
                                   int(ext["meta"]["off"][i]), int(ext["meta"]["strand"][i]))
            n_valid += int(ext["score"][i] >= 0 and ext["meta_r"][i] != 0)
    assert got == want
    assert total.value == len(lens) and valid.value == n_valid
    lines = got.splitlines()[4:]
    # a read shorter than a seed has no votes: best = {0,0,0}, i.e. locus 0 -- which resolves (quirk kept)
    assert lines[3].split("\t")[1:4] == ["0", "chr1", "1"]
    assert sum(l.split("\t")[1] == "16" for l in lines) > 20
