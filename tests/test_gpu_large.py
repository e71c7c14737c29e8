"""Rows and suffix-array values beyond 2^32 through the whole GPU path, against the oracle: a 2.2 Gbp synthetic
reference (4.4 G rows: more than 2^32, so `k`, `l`, SA values and loci all need their high bits), built by the
product's parallel suffix sorter.  Needs ~75 GB of host RAM and about one minute on the GPU box's 16 CPUs
(`profiles/r2/large_test.log`); it is skipped when the host has less than 110 GB available or with
LRM_TEST_LARGE=0.  bench.py --ref-len 3099750718 asserts the same equality on the GRCh38-sized text in every run."""
import os

import numpy as np
import pytest

import orc
from longreadmapper_amd import index, mapper, synth

pytestmark = pytest.mark.gpu


def _host_gb_available():
    gb = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                gb = int(line.split()[1]) / 1e6
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            gb = min(gb, int(lim) / 1e9)
    except (OSError, ValueError):
        pass
    return gb


@pytest.mark.skipif(os.environ.get("LRM_TEST_LARGE") == "0" or _host_gb_available() < 110,
                    reason="needs 75 GB of host RAM (LRM_TEST_LARGE=0 switches it off)")
def test_rows_beyond_2p32_end_to_end(gpu):
    n_ref = 2_200_000_000
    ref = synth.reference(n_ref, seed=5, repeat_frac=0.02, rep_len=300, rep_copies=500, rep_div=0.05)
    hi = index.HostIndex.build([ref], hlen=12)
    assert hi.length == 2 * n_ref + 1 > 1 << 32
    # half of the reads from the first 100 Mbp: their reverse-strand copies map into the top of the revcomp half of
    # the .cat text, i.e. to loci (and SA values) beyond 2^32
    ra = synth.reads([ref], 1500, 5000, synth.ONT, seed=9)
    rb = synth.reads([ref[:100_000_000]], 1500, 5000, synth.ONT, seed=10)
    r = {k: np.concatenate([ra[k], rb[k]]) for k in ("reads", "lens", "pos", "strand")}
    oi = orc.OracleIndex.from_host_index(hi)
    want, _ = oi.seed_batch(r["reads"], r["lens"], nthreads=16)
    rc = r["reads"].copy()
    wext = oi.extend_batch(rc, r["lens"], want, nthreads=16)
    assert (want["key"] >= np.uint64(1 << 32)).mean() > 0.15               # loci beyond 2^32
    di = index.DeviceIndex.upload(hi, gpu)
    try:
        rg = r["reads"].copy()
        got = mapper.map_batch(di, rg, r["lens"])
        assert np.array_equal(got["best"], want)
        assert np.array_equal(got["score"], wext["score"]) and np.array_equal(got["n_ops"], wext["n_ops"])
        assert np.array_equal(got["meta"]["loc"], wext["meta"]["loc"]) and np.array_equal(rg, rc)
        for i in range(0, 3000, 37):
            k = int(wext["n_ops"][i])
            assert bytes(got["ops"][i, :k]) == bytes(wext["ops"][i, :k]), i
        near = (np.abs(got["meta"]["off"].astype(np.int64) - r["pos"].astype(np.int64)) < 300) & (got["meta"]["strand"] == r["strand"])
        assert near.mean() > 0.97
    finally:
        di.close()
