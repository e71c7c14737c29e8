"""BASELINE.json configs other than the bench line, as parity cases at a size the CPU oracle finishes in
seconds: PacBio-CLR 15 kbp reads (config 2's read profile), 10 kbp ONT reads against a multi-sequence
reference with planted repeats (config 3's shape), and 100 kbp ultra-long reads with the GACT
tile / overlap / band sweep of config 4, including the full-tile band W = T."""
import numpy as np
import pytest

import orc
from longreadmapper_amd import capi, index, mapper, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref3(gpu):
    seqs = [synth.reference(1_500_000, seed=41, repeat_frac=0.05, rep_len=300, rep_copies=200, rep_div=0.05),
            synth.reference(700_000, seed=42), synth.reference(300_000, seed=43)]
    hi = index.HostIndex.build(seqs, names=["chrA", "chrB", "chrC"], hlen=12)
    di = index.DeviceIndex.upload(hi, gpu)
    yield seqs, hi, di, orc.OracleIndex.from_host_index(hi)
    di.close()


def _compare(di, oi, reads, lens, gact, threads=8):
    want_best, _ = oi.seed_batch(reads, lens, nthreads=threads)
    got_best = mapper.seed_batch(di, reads, lens)
    assert np.array_equal(got_best, want_best)
    rc, rg = reads.copy(), reads.copy()
    want = oi.extend_batch(rc, lens, want_best, gact, nthreads=threads)
    got = mapper.extend_batch(di, rg, lens, got_best, gact)
    assert np.array_equal(got["meta_r"], want["meta_r"]) and np.array_equal(got["score"], want["score"])
    assert np.array_equal(got["n_ops"], want["n_ops"]) and np.array_equal(rc, rg)
    for i in range(len(lens)):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), i
    return got


def test_pacbio_clr_15k(ref3):
    seqs, hi, di, oi = ref3
    r = synth.reads(seqs, 48, 15_000, synth.PACBIO_CLR, seed=13)
    got = _compare(di, oi, r["reads"], r["lens"], (320, 120, 128))
    # CLR reads insert twice as often as they delete: the read's diagonal drifts by (ins - del) * Lr ~ 675 bases
    # over 15 kbp, while the reference's scheme aligns against ONE window of the read's length at the voted
    # diagonal (alnmain.c:440-446).  The start of the read can sit hundreds of bases off that diagonal, out of
    # any band, so part of the read aligns as noise: ED rates well above the 15 % error rate are expected here
    # and are a property of the pipeline, not of the kernel (GPU == oracle is what this test pins).
    rate = got["score"][got["score"] >= 0] / 15_000
    assert 0.10 < np.median(rate) < 0.6


def test_ont_10k_multi_sequence_with_repeats(ref3):
    seqs, hi, di, oi = ref3
    r = synth.reads(seqs, 64, 10_000, synth.ONT, seed=11)
    got = _compare(di, oi, r["reads"], r["lens"], (320, 120, 128))
    near = (got["meta"]["seq_id"] == r["seq"]) & (got["meta"]["strand"] == r["strand"]) & \
           (np.abs(got["meta"]["off"].astype(np.int64) - r["pos"].astype(np.int64)) < 300)
    assert near.mean() > 0.95


@pytest.mark.parametrize("T,O,W", [(128, 32, 32), (320, 120, 128), (512, 120, 128), (256, 64, 256)])
def test_ultralong_100k_byte_kernels(ref3, T, O, W):
    """Small batches take the byte kernels (two reads per wavefront / one read per wavefront for T - O > 256 /
    wide band); the full sweep below runs the W <= 128 points on the bit-sliced kernel."""
    seqs, hi, di, oi = ref3
    r = synth.reads(seqs, 6, 100_000, synth.ONT, seed=17)
    _compare(di, oi, r["reads"], r["lens"], (T, O, W))


@pytest.mark.parametrize("T,O,W", [(320, 120, 128), (512, 64, 64)])
def test_ultralong_100k_bitsliced(ref3, map_options, T, O, W):
    """The W <= 128 points of the config-4 sweep through the lane-per-read kernel (500 tiles per read), reads of
    100 kbp next to short ones so that lanes finish at very different times."""
    seqs, hi, di, oi = ref3
    map_options(di, gact_impl=4)
    r = synth.reads(seqs, 6, 100_000, synth.ONT, seed=17)
    lens = r["lens"].copy()
    lens[1], lens[4] = 7_000, 333
    _compare(di, oi, r["reads"], lens, (T, O, W))


SWEEP = sorted({(T, O, W) for T in (128, 256, 320, 512) for O in (32, 64, 120) for W in (32, 64, 128, T)})


@pytest.fixture(scope="module")
def ultralong(ref3):
    seqs, hi, di, oi = ref3
    r = synth.reads(seqs, 4, 100_000, synth.ONT, seed=23)
    lens = r["lens"].copy()
    lens[3] = 41_000                           # reads of different lengths finish at different times
    want_best, _ = oi.seed_batch(r["reads"], lens, nthreads=8)
    got_best = mapper.seed_batch(di, r["reads"], lens)
    assert np.array_equal(got_best, want_best)
    return r["reads"], lens, want_best


@pytest.mark.parametrize("T,O,W", SWEEP, ids=["T%d-O%d-W%d" % p for p in SWEEP])
def test_config5_full_gact_sweep(ref3, ultralong, map_options, T, O, W):
    """every point of T in {128,256,320,512} x O in {32,64,120} x W in {32,64,128,T} on
    100 kbp reads (45 distinct points: W = T coincides with 128 for T = 128).  W <= 128 runs on the bit-sliced
    lane-per-read kernel (what large batches use), W > 128 on the wide-band kernel; extension results are
    compared with the oracle bit for bit (scores, op strings, rev-comped reads)."""
    seqs, hi, di, oi = ref3
    reads, lens, best = ultralong
    if W <= 128:
        map_options(di, gact_impl=4)
    rc, rg = reads.copy(), reads.copy()
    want = oi.extend_batch(rc, lens, best, (T, O, W), nthreads=8)
    got = mapper.extend_batch(di, rg, lens, best, (T, O, W))
    assert np.array_equal(got["meta_r"], want["meta_r"]) and np.array_equal(got["score"], want["score"])
    assert np.array_equal(got["n_ops"], want["n_ops"]) and np.array_equal(rc, rg)
    for i in range(len(lens)):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), i
    assert (want["score"] >= 0).all()


def test_pacbio_and_multiseq_bitsliced(ref3, map_options):
    seqs, hi, di, oi = ref3
    map_options(di, gact_impl=4)
    r = synth.reads(seqs, 48, 15_000, synth.PACBIO_CLR, seed=13)
    _compare(di, oi, r["reads"], r["lens"], (320, 120, 128))
    r = synth.reads(seqs, 64, 10_000, synth.ONT, seed=11)
    _compare(di, oi, r["reads"], r["lens"], (320, 120, 128))


def test_largest_tile_and_band(ref3):
    """T = 512, O = 0, W = 1024: the largest traceback the kernel supports (134 KiB of LDS per wavefront)."""
    seqs, hi, di, oi = ref3
    r = synth.reads(seqs, 8, 3_000, synth.ONT, seed=19)
    _compare(di, oi, r["reads"], r["lens"], (512, 0, 1024))


@pytest.mark.parametrize("alphabet,impl", [(b"ACGTacgtNRY-", None), (b"ACGTacgtNRY-", 4), (b"ACGT", 4)])
def test_revcomp_in_place_ragged_rows(ref3, map_options, alphabet, impl):
    """`_rev_comp_in_place` (alnmain.c:27-60) on rows of every alignment: lengths 1..70, around the kernel's
    4096-base span boundaries, odd and even, with lower-case and non-ACGT bytes (-> 'N'), every read placed on the
    reverse strand by its locus; forward-strand rows in between must stay untouched.  With gact_impl = 4 the same
    rows also go through the planar packer of the bit-sliced kernel (16 bases per lane from unaligned rows; reads with
    a byte other than ACGT are flagged there and fall back to the byte kernel)."""
    seqs, hi, di, oi = ref3
    if impl:
        map_options(di, gact_impl=impl)
    lens = list(range(1, 71)) + [4095, 4096, 4097, 8191, 8192, 8193, 8223, 8224, 8225, 12289, 16384, 16399, 20001]
    n, mx = len(lens), max(lens)
    rng = np.random.default_rng(5)
    reads = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=(n, mx + 1))].copy()
    lens = np.array(lens, dtype=np.uint32)
    for i in range(n):
        reads[i, lens[i]:] = 0
    best = np.zeros(n, dtype=mapper.ENTRY_DT)
    L0 = len(seqs[0])
    best["key"] = np.where(np.arange(n) % 3 == 2, 1000 + 37 * np.arange(n), L0 + 5000 + 41 * np.arange(n))   # seq 0: fwd / revcomp half
    rc, rg = reads.copy(), reads.copy()
    want = oi.extend_batch(rc, lens, best, (64, 16, 32), nthreads=8)
    got = mapper.extend_batch(di, rg, lens, best, (64, 16, 32))
    assert np.array_equal(got["meta_r"], want["meta_r"]) and (want["meta_r"] == 1).all()
    assert (got["meta"]["strand"] == (np.arange(n) % 3 != 2)).all()
    assert np.array_equal(rc, rg)
    assert not np.array_equal(rg, reads)
    assert np.array_equal(got["score"], want["score"]) and np.array_equal(got["n_ops"], want["n_ops"])
