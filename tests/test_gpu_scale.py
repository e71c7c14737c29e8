"""Size-independent properties at BASELINE configs[1] scale (E. coli sized reference, 10 kbp ONT
reads; a 4000-read slice of the 100k batch) where the oracle would take minutes: CIGAR
well-formedness and re-scoring, truth recovery, determinism, batch-split invariance, and an
oracle spot check on a few reads."""
import numpy as np
import pytest

import orc
from longreadmapper_amd import index, mapper, synth

pytestmark = pytest.mark.gpu
N_READS, LR = 4000, 10_000


@pytest.fixture(scope="module")
def world(gpu):
    ref = synth.reference(4_641_652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
    hi = index.HostIndex.build([ref], hlen=12)
    di = index.DeviceIndex.upload(hi, gpu)
    r = synth.reads([ref], N_READS, LR, synth.ONT, seed=11)
    best = mapper.seed_batch(di, r["reads"], r["lens"])
    reads = r["reads"].copy()
    ext = mapper.extend_batch(di, reads, r["lens"], best)
    yield dict(ref=ref, hi=hi, di=di, r=r, best=best, reads=reads, ext=ext)
    di.close()


def test_cigars_are_wellformed_and_rescore(world):
    ext, reads, hi = world["ext"], world["reads"], world["hi"]
    content = hi.content()
    assert (ext["meta_r"] == 1).all() and (ext["score"] >= 0).all()
    eq, X, I, D = ord("="), ord("X"), ord("I"), ord("D")
    for i in range(0, N_READS, 7):
        k = int(ext["n_ops"][i])
        ops = ext["ops"][i, :k]
        assert np.isin(ops, [eq, X, I, D]).all()
        qn = int(((ops == eq) | (ops == X) | (ops == I)).sum())
        tn = int(((ops == eq) | (ops == X) | (ops == D)).sum())
        assert qn == LR and tn <= LR                             # consumes the whole read, a prefix of the window
        assert int((ops != eq).sum()) == int(ext["score"][i])    # ED:I == X + I + D
        # replay: '=' columns really match, 'X' columns really differ
        loc = int(ext["meta"]["loc"][i])
        qi = np.cumsum((ops != D)) - 1
        ti = np.cumsum((ops != I)) - 1
        q = reads[i, :LR]
        t = content[loc:loc + LR]
        m = ops == eq
        x = ops == X
        assert (q[qi[m]] == t[ti[m]]).all() and (q[qi[x]] != t[ti[x]]).all()


def test_loci_and_error_rate_match_the_simulation(world):
    ext, r = world["ext"], world["r"]
    near = (np.abs(ext["meta"]["off"].astype(np.int64) - r["pos"].astype(np.int64)) < 300) & \
           (ext["meta"]["strand"] == r["strand"])
    assert near.mean() > 0.99
    rate = ext["score"][near].astype(np.float64) / LR
    assert 0.07 < np.median(rate) < 0.13                         # 4 % sub + 3 % ins + 3 % del


def test_deterministic_and_independent_of_batch_split(world):
    di, r = world["di"], world["r"]
    a = mapper.seed_batch(di, r["reads"][:600], r["lens"][:600])
    b = np.concatenate([mapper.seed_batch(di, r["reads"][lo:hi], r["lens"][lo:hi])
                        for lo, hi in ((0, 1), (1, 130), (130, 600))])
    assert np.array_equal(a, world["best"][:600]) and np.array_equal(b, a)
    reads = r["reads"][:300].copy()
    e = mapper.extend_batch(di, reads, r["lens"][:300], a[:300])
    assert np.array_equal(e["score"], world["ext"]["score"][:300])
    assert np.array_equal(e["n_ops"], world["ext"]["n_ops"][:300])


def test_oracle_spot_check(world):
    oi = orc.OracleIndex.from_host_index(world["hi"])
    r = world["r"]
    idx = np.arange(0, N_READS, 250)
    best, _ = oi.seed_batch(r["reads"][idx], r["lens"][idx], nthreads=8)
    assert np.array_equal(best, world["best"][idx])
    reads = np.ascontiguousarray(r["reads"][idx]).copy()
    want = oi.extend_batch(reads, r["lens"][idx], best, nthreads=8)
    assert np.array_equal(want["score"], world["ext"]["score"][idx])
    for n, i in enumerate(idx):
        k = int(want["n_ops"][n])
        assert bytes(want["ops"][n, :k]) == bytes(world["ext"]["ops"][i, :k])


def test_bitsliced_and_score_kernels_agree_at_batch_scale(world, map_options):
    """20 k reads: above the automatic switch to the lane-per-read bit-sliced kernel.  Its CIGARs must equal the
    score kernel's for every read (two independent formulations of docs/GACT_SPEC.md), and the oracle's on a few."""
    di = world["di"]
    r = synth.reads([world["ref"]], 20_000, LR, synth.ONT, seed=23)
    best = mapper.seed_batch(di, r["reads"], r["lens"])
    ra = r["reads"].copy()
    a = mapper.extend_batch(di, ra, r["lens"], best)                 # automatic: bit-sliced at this size
    map_options(di, gact_impl=3)
    rb = r["reads"].copy()
    b = mapper.extend_batch(di, rb, r["lens"], best)
    assert np.array_equal(ra, rb)
    assert np.array_equal(a["score"], b["score"]) and np.array_equal(a["n_ops"], b["n_ops"])
    cols = np.arange(a["ops"].shape[1])[None, :] < a["n_ops"][:, None]
    assert np.array_equal(np.where(cols, a["ops"], 0), np.where(cols, b["ops"], 0))
    oi = orc.OracleIndex.from_host_index(world["hi"])
    idx = np.arange(0, 20_000, 2500)
    reads = np.ascontiguousarray(r["reads"][idx]).copy()
    want = oi.extend_batch(reads, r["lens"][idx], best[idx], nthreads=8)
    assert np.array_equal(want["score"], a["score"][idx])
    for n, i in enumerate(idx):
        k = int(want["n_ops"][n])
        assert bytes(want["ops"][n, :k]) == bytes(a["ops"][i, :k])
