"""GPU parity: the HIP path (through the C-ABI of liblrm_accel.so) against the CPU oracle on the
same seeded inputs.  Bit-exact: every compared quantity is an integer, byte or index."""
import ctypes as C

import numpy as np
import pytest

import orc
import workloads
from longreadmapper_amd import capi, index, mapper, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev_indexes(gpu):
    cache = {}

    def get(name):
        if name not in cache:
            sc = workloads.scenario(name)
            cache[name] = (sc, index.DeviceIndex.upload(sc["hi"], gpu), orc.OracleIndex.from_host_index(sc["hi"]))
        return cache[name]
    yield get
    for _, di, _ in cache.values():
        di.close()


def _gpu_gact(q, d, T, O, W, impl=0):
    qa = np.frombuffer(q, dtype=np.uint8)
    da = np.frombuffer(d, dtype=np.uint8)
    ops = np.zeros(len(q) + len(d) + 16, dtype=np.uint8)
    n_ops, score = C.c_int(), C.c_int()
    capi.check(capi.lib.lrm_debug_gact_impl(qa.ctypes.data, len(q), da.ctypes.data, len(d), capi.GactParams(T, O, W), impl,
                                            ops.ctypes.data, C.byref(n_ops), C.byref(score), 0), "lrm_debug_gact_impl")
    return score.value, bytes(ops[:n_ops.value])


def _mutate(rng, s, sub, ins, dele):
    out = bytearray()
    for c in s:
        x = rng.random()
        if x < dele:
            continue
        if x < dele + sub:
            c = rng.choice([b for b in b"ACGT" if b != c])
        out.append(c)
        if rng.random() < ins:
            out.append(rng.choice(list(b"ACGT")))
    return bytes(out)


GACT_PARAMS = [(320, 120, 128), (128, 32, 64), (64, 16, 32), (16, 0, 2), (320, 120, 20), (512, 120, 128),
               (100, 99, 128), (33, 7, 66), (320, 0, 128),
               (320, 120, 320), (128, 32, 256), (64, 16, 600), (200, 40, 1024)]      # wide bands: gact_wide_kernel


@pytest.mark.parametrize("T,O,W", GACT_PARAMS)
def test_gact_kernel_vs_oracle(gpu, T, O, W):
    rng = np.random.default_rng(T * 1000 + O * 10 + W)
    ref = bytes(synth.reference(20000, seed=3))
    sizes = [1, 2, 5, 31, 63, 64, 65, 127, 199, 200, 201, 319, 320, 321, 500, 1000, 2500]
    for n in sizes:
        for prof in ((0, 0, 0), (0.04, 0.03, 0.03), (0.015, 0.09, 0.045), (0.2, 0.1, 0.1)):
            p = int(rng.integers(0, len(ref) - 3 * n - 64))
            q = _mutate(rng, ref[p:p + n], *prof) or b"C"
            for m in {len(q), max(1, len(q) - 7), len(q) + 13}:
                d = ref[p:p + m]
                want = orc.gact(q, d, T, O, W)
                got = _gpu_gact(q, d, T, O, W)
                assert got == (want[0], want[1]), (n, prof, m)
    # unrelated sequences and a start offset (leading gap)
    q, d = ref[100:700], ref[9000:9600]
    assert _gpu_gact(q, d, T, O, W) == orc.gact(q, d, T, O, W)[:2]
    q, d = ref[140:900], ref[100:860]
    assert _gpu_gact(q, d, T, O, W) == orc.gact(q, d, T, O, W)[:2]


BS_PARAMS = [(320, 120, 128), (512, 120, 128), (100, 99, 128), (320, 0, 128), (64, 16, 128), (33, 7, 128), (16, 0, 128),
             (512, 0, 128),
             (320, 120, 64), (320, 120, 32), (128, 32, 64), (64, 16, 32), (16, 0, 2), (320, 120, 20), (33, 7, 66),
             (256, 120, 64), (512, 64, 64)]          # bands narrower than the 128 diagonals of the planes


@pytest.mark.parametrize("T,O,W", BS_PARAMS)
def test_gact_bitsliced_kernel_vs_oracle(gpu, T, O, W):
    """gact_impl = 4: the lane-per-read bit-sliced kernel (normally used for batches >= 16 k reads)."""
    rng = np.random.default_rng(T * 1000 + O * 10 + 7)
    ref = bytes(synth.reference(20000, seed=3))
    sizes = [1, 2, 5, 31, 63, 64, 65, 127, 199, 200, 201, 319, 320, 321, 500, 1000, 2500]
    if W != 128:
        sizes = sizes[::2]
    for n in sizes:
        for prof in ((0, 0, 0), (0.04, 0.03, 0.03), (0.015, 0.09, 0.045), (0.2, 0.1, 0.1)):
            p = int(rng.integers(0, len(ref) - 3 * n - 64))
            q = _mutate(rng, ref[p:p + n], *prof) or b"C"
            for m in {len(q), max(1, len(q) - 7), len(q) + 13}:
                d = ref[p:p + m]
                want = orc.gact(q, d, T, O, W)
                got = _gpu_gact(q, d, T, O, W, 4)
                assert got == (want[0], want[1]), (n, prof, m)
    q, d = ref[100:700], ref[9000:9600]
    assert _gpu_gact(q, d, T, O, W, 4) == orc.gact(q, d, T, O, W)[:2]
    q, d = ref[140:900], ref[100:860]
    assert _gpu_gact(q, d, T, O, W, 4) == orc.gact(q, d, T, O, W)[:2]
    # a byte other than ACGT in the read or in the text: routed to the byte kernel, same answer
    q, d = bytearray(ref[300:1300]), bytearray(ref[300:1300])
    q[500] = ord("N")
    assert _gpu_gact(bytes(q), bytes(d), T, O, W) == orc.gact(bytes(q), bytes(d), T, O, W)[:2]
    d[100] = ord("a")
    assert _gpu_gact(bytes(q), bytes(d), T, O, W) == orc.gact(bytes(q), bytes(d), T, O, W)[:2]


def test_gact_rejects_unsupported_params(gpu):
    q = np.frombuffer(b"ACGT", dtype=np.uint8)
    ops = np.zeros(16, dtype=np.uint8)
    a, b = C.c_int(), C.c_int()
    for gp in ((320, 320, 128), (320, 120, 129), (8, 0, 8), (320, 120, 1026)):
        rc = capi.lib.lrm_debug_gact(q.ctypes.data, 4, q.ctypes.data, 4, capi.GactParams(*gp), ops.ctypes.data,
                                     C.byref(a), C.byref(b), 0)
        assert rc < 0 and b"unsupported GACT" in capi.lib.lrm_last_error()


def s_len_of(sc):
    return sc["seed_len"]


@pytest.mark.parametrize("long_table", ["0", "auto", "16-nocore", "16-core", "13-plain", "14", "15-5byte", "14-5byte-side",
                                        "sd", "sd-share2", "sd-crowded", "sd-share2-crowded-counts"])
@pytest.mark.parametrize("name", ["clean-1k", "ont-2k", "pacbio-3k-h12", "ragged", "seed12", "seed32",
                                  "seed-below-hlen", "repeats-ties"])
def test_seed_search_per_seed(dev_indexes, gpu, name, long_table):
    """K1 alone: (j, rr, k, l) of every seed of a read.  With lc_long = 0 (the reference's table only) also the
    k > l pairs of failed searches are the reference's (fmidx.c:310-312); through the long seed table (automatic:
    pair-line 16-mers when 128 GiB of HBM are free, shorter k-mers otherwise; "13-plain": the plain layout) a seed that
    dies inside its last hl bases reports k = l = 0 -- rr = 0 either way, and the reference never reads k, l of such
    a seed (alnmain.c:357-366)."""
    if long_table not in ("0", "auto") and name not in ("ont-2k", "seed32", "ragged", "repeats-ties"):
        pytest.skip("explicit table variants are built for three scenarios only")
    sc, di, oi = dev_indexes(name)
    own = None
    if long_table.startswith("sd"):
        # the SEED table: (k, count) of every distinct 20-mer of the text, four (or two) read positions per 64-byte line;
        # "crowded": so few lines that most entries live in the side hash table; "counts": 2 count bits per slot
        own = di = index.DeviceIndex.upload(sc["hi"], gpu, seed_table=1, lc_long=14,
                                            seed_table_share=2 if "share2" in long_table else None,
                                            seed_table_bits=(14 if "share2" in long_table else 16) if "crowded" in long_table else None,
                                            seed_table_count_bits=2 if "counts" in long_table else None)
        t = di.tables()
        if s_len_of(sc) == 20:
            assert t["seed_table_len"] == 20 and t["seed_table_share"] == (2 if "share2" in long_table else 4), t
            assert t["seed_table_slot_bytes"] == (6 if "share2" in long_table else 8), t
            if "crowded" in long_table:
                assert t["seed_table_side_entries"] > 1000, t
    elif long_table != "auto":
        own = di = index.DeviceIndex.upload(sc["hi"], gpu, seed_table=0, lc_long=int(long_table.split("-")[0]),
                                            lc_pair=0 if long_table.endswith("-plain") else None,
                                            lc_core=1 if long_table.endswith("-core") else 0,         # four positions per line
                                            lc_entry_bytes=5 if "5byte" in long_table else None,      # 40 bytes per (k-1)-mer
                                            lc_count_bits=2 if long_table.endswith("-side") else None)  # counts >= 3: side hash table
    s = sc["seed_len"]
    sa = sc["hi"].sa()
    for i in range(0, len(sc["lens"]), 5):
        ln = int(sc["lens"][i])
        read = np.ascontiguousarray(sc["reads"][i, :max(ln, 1)])
        cap = (ln // (s + 1) + 2) * (s + 1)
        j = np.zeros(cap, dtype=np.int32)
        rr = np.zeros(cap, dtype=np.uint64)
        k = np.zeros(cap, dtype=np.uint64)
        l = np.zeros(cap, dtype=np.uint64)
        n_out = C.c_uint64()
        cap_q = capi.check(capi.lib.lrm_debug_seed_search(di.handle, read.ctypes.data, ln, s, sc["thres"],
                                                          j.ctypes.data, rr.ctypes.data, k.ctypes.data,
                                                          l.ctypes.data, cap, C.byref(n_out)), "debug_seed_search")
        got = {}
        for x in range(n_out.value):
            if j[x] >= 0:
                got[int(j[x])] = (int(rr[x]), int(k[x]), int(l[x]))
        # the oracle with thres=0 never votes, so it never breaks: all phases, every seed position
        tr = oi.seed_read(bytes(sc["reads"][i, :ln]), s, 0, trace=True)
        want = {jj: (r_, k_, l_) for jj, r_, k_, l_ in tr["seeds"]}
        if long_table != "0":
            dead = lambda d: {jj: (v if v[0] > 0 else (0, 0, 0)) for jj, v in d.items()}
            got, want = dead(got), dead(want)
            # a unique seed found in the seed table comes with its TEXT POSITION (bit 39 of the row field set) instead of
            # its row: SA[k] of the oracle's row
            LOC = 1 << 39
            for jj, v in got.items():
                if v[1] & LOC:
                    w = want[jj]
                    assert v[0] == 1 and w[0] == 1 and (v[1] & (LOC - 1)) == int(sa[w[1]]), (name, i, jj, v, w)
                    got[jj] = w
        assert got == want, (name, i)
        assert cap_q >= 1
    if own is not None:
        own.close()


@pytest.mark.parametrize("name", workloads.SEED_SCENARIOS)
def test_seed_batch_vs_oracle(dev_indexes, gpu, name):
    sc, di, oi = dev_indexes(name)
    want, phases = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    got = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    for f in ("key", "val", "bucket"):
        assert np.array_equal(got[f], want[f]), (name, f, np.nonzero(got[f] != want[f])[0][:10])
    # the same through the long seed table (what large texts use automatically)
    for tag, opts in (("long table", dict(lc_long=14, seed_table=0)),
                      ("pair-line 16-mers without the core table", dict(lc_long=16, lc_core=0, seed_table=0)),
                      ("5-byte long table", dict(lc_long=15, lc_entry_bytes=5, seed_table=0)),
                      ("5-byte long table, side hash table", dict(lc_long=14, lc_entry_bytes=5, lc_count_bits=2, seed_table=0)),
                      ("lchash alone", dict(lc_long=0, seed_table=0)),
                      ("seed table, two positions per line", dict(seed_table=1, seed_table_share=2, lc_long=13)),
                      ("seed table, crowded lines", dict(seed_table=1, seed_table_bits=16, lc_long=13)),
                      ("seed table, two positions per line, crowded, 2 count bits",
                       dict(seed_table=1, seed_table_share=2, seed_table_bits=14, seed_table_count_bits=2, lc_long=13))):
        d2 = index.DeviceIndex.upload(sc["hi"], gpu, **opts)
        got = mapper.seed_batch(d2, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        for f in ("key", "val", "bucket"):
            assert np.array_equal(got[f], want[f]), (name, f, tag)
        if "seed table" in tag:
            # all phases in one launch: the lanes that share a line of the seed table fetch it together
            d2.set_map_options(seed_rounds=1)
            got = mapper.seed_batch(d2, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
            for f in ("key", "val", "bucket"):
                assert np.array_equal(got[f], want[f]), (name, f, tag, "one round")
        d2.close()
    if name in ("seed16", "seed24"):
        # a seed table built for THIS seed length.  16: four positions per line (26-bit cores: the hash keeps fewer than 32
        # bits) and two (6-byte slots whose 15-bit tags are compared as halfwords -- the form a GRCh38-sized text takes);
        # 24: cores of 42 / 46 bits, the bits above 32 folded into the line index
        for share in (4, 2):
            # (6-byte slots have room for the 46-bit core's tag only with many lines: a small text gets them by hand)
            d2 = index.DeviceIndex.upload(sc["hi"], gpu, seed_table=1, seed_table_len=sc["seed_len"], seed_table_share=share, lc_long=13,
                                          seed_table_bits=24 if (name == "seed24" and share == 2) else None)
            t = d2.tables()
            assert t["seed_table_len"] == sc["seed_len"] and t["seed_table_share"] == share and t["seed_table_slot_bytes"] == (8 if share == 4 else 6), t
            for rounds in (0, 1):
                d2.set_map_options(seed_rounds=rounds)
                got = mapper.seed_batch(d2, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
                for f in ("key", "val", "bucket"):
                    assert np.array_equal(got[f], want[f]), (name, f, share, rounds)
            d2.close()
    if name == "clean-1k":
        assert (phases == 1).mean() > 0.7          # exercised the phase-0 early decision
    if name == "ont-2k":
        assert (phases == 21).all()


def test_vote_tiers_beyond_the_wave_table(dev_indexes, gpu):
    sc, di, oi = dev_indexes("repeats-overflow")
    import torch
    n, stride = sc["reads"].shape
    dm = mapper.DeviceMapper(di, n, stride - 1, sc["seed_len"], sc["thres"], device=gpu)
    d_reads = torch.from_numpy(sc["reads"]).cuda()
    d_lens = torch.from_numpy(sc["lens"].astype(np.int32)).cuda()
    dm.seed(d_reads, d_lens)
    st = dm.stats()
    assert st["vote_tier2_items"] > 0 and st["vote_tier3_items"] > 0
    want, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    got = dm.results(n)["best"]
    for f in ("key", "val", "bucket"):
        assert np.array_equal(got[f], want[f])
    dm.close()


@pytest.mark.parametrize("name", ["clean-1k", "ont-2k", "pacbio-3k-h12", "ragged", "repeats-ties"])
@pytest.mark.parametrize("gact", [(320, 120, 128), (128, 64, 32), (320, 120, 128, "bitsliced")])
def test_extend_batch_vs_oracle(dev_indexes, map_options, name, gact):
    sc, di, oi = dev_indexes(name)
    if len(gact) == 4:
        map_options(di, gact_impl=4)
        gact = gact[:3]
    best, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    if name == "ragged":     # also: a wrapped diagonal, a locus straddling the strand boundary, the last bases
        best = best.copy()
        N = len(sc["seqs"][0])
        best["key"][5] = (1 << 64) - 3
        best["key"][6] = N - 10
        best["key"][7] = 2 * N - 41
        best["key"][8] = 2 * N + 5
    r_cpu = sc["reads"].copy()
    want = oi.extend_batch(r_cpu, sc["lens"], best, gact)
    r_gpu = sc["reads"].copy()
    got = mapper.extend_batch(di, r_gpu, sc["lens"], best, gact)
    assert np.array_equal(got["meta_r"], want["meta_r"])
    for f in ("loc", "off", "seq_id", "strand"):
        assert np.array_equal(got["meta"][f], want["meta"][f]), f
    assert np.array_equal(got["score"], want["score"])
    assert np.array_equal(got["n_ops"], want["n_ops"])
    for i in range(len(best)):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), (name, i)
    assert np.array_equal(r_gpu, r_cpu)            # reverse-strand reads were rev-comped in place
    assert (want["meta"]["strand"] == 1).any() or name == "ragged"
    flag, mapq, valid = mapper.result_flags(got["score"], got["meta_r"], got["meta"])
    assert ((flag & 4) != 0).sum() == ((want["meta_r"] == 0) | (want["score"] == -1)).sum()


def test_device_resident_pipeline_equals_host_path(dev_indexes, gpu):
    import torch
    sc, di, oi = dev_indexes("ont-2k")
    n, stride = sc["reads"].shape
    dm = mapper.DeviceMapper(di, n, stride - 1, sc["seed_len"], sc["thres"], device=gpu)
    d_reads = torch.from_numpy(sc["reads"].copy()).cuda()
    d_lens = torch.from_numpy(sc["lens"].astype(np.int32)).cuda()
    for _ in range(2):                              # reads are rev-comped in place: reload per pass
        d_reads.copy_(torch.from_numpy(sc["reads"]))
        dm.seed(d_reads, d_lens)
        dm.extend(d_reads, d_lens)
    torch.cuda.synchronize()
    res = dm.results(n)
    best = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    assert np.array_equal(res["best"], best)
    r = sc["reads"].copy()
    ext = mapper.extend_batch(di, r, sc["lens"], best)
    assert np.array_equal(res["score"], ext["score"]) and np.array_equal(res["n_ops"], ext["n_ops"])
    assert np.array_equal(d_reads.cpu().numpy(), r)
    st = dm.stats()
    assert st["gact_tiles"] > 0
    dm.close()


@pytest.mark.parametrize("name", ["ont-2k", "ragged"])
def test_bitsliced_kernel_is_the_one_that_runs(dev_indexes, gpu, map_options, name):
    """Device-resident extend with gact_impl = 4: the launch record must show gact_bs_kernel (no silent
    route through the byte kernels), results equal the oracle's; one read carries an N (byte kernel, flagged)."""
    import torch
    sc, di, oi = dev_indexes(name)
    map_options(di, gact_impl=4)
    reads = sc["reads"].copy()
    n, stride = reads.shape
    dm = mapper.DeviceMapper(di, n, stride - 1, sc["seed_len"], sc["thres"], device=gpu)
    dm.set_timing(True)
    d_reads = torch.from_numpy(reads.copy()).cuda()
    d_lens = torch.from_numpy(sc["lens"].astype(np.int32)).cuda()
    dm.seed(d_reads, d_lens)
    best, _ = oi.seed_batch(reads, sc["lens"], sc["seed_len"], sc["thres"])
    if name == "ont-2k":           # the N goes in after seeding (a non-ACGT base is undefined in the reference's lchash)
        reads[3, 700] = ord("N")
        d_reads[3, 700] = ord("N")
    dm.extend(d_reads, d_lens)
    torch.cuda.synchronize()
    t = dm.timing()
    assert t["gact_bs_kernel"][1] == 1 and t["bs_pack_reads_kernel"][1] == 1 and t["gact_kernel"][1] == 0
    res = dm.results(n)
    assert np.array_equal(res["best"], best)
    r_cpu = reads.copy()
    want = oi.extend_batch(r_cpu, sc["lens"], best, (320, 120, 128))
    assert np.array_equal(res["score"], want["score"]) and np.array_equal(res["n_ops"], want["n_ops"])
    for i in range(n):
        k = int(want["n_ops"][i])
        assert bytes(res["ops"][i, :k]) == bytes(want["ops"][i, :k]), i
    assert np.array_equal(d_reads.cpu().numpy(), r_cpu)
    dm.close()


@pytest.mark.parametrize("waves", [1, 2])
def test_bitsliced_lane_refill(dev_indexes, map_options, waves):
    """lrm_map_options.bs_waves: a grid of one or two wavefronts for 230 reads of very different lengths -- every lane takes
    several reads from the queue in turn (what happens to every batch above 131 k reads), next to fenced reads."""
    sc, di, oi = dev_indexes("ont-2k")
    map_options(di, gact_impl=4, bs_waves=waves)
    rng = np.random.default_rng(5)
    n0, stride = sc["reads"].shape
    n = 230
    reads = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        src = int(rng.integers(0, n0))
        ln = int(rng.choice([0, 1, 30, 199, 200, 201, 500, 1000, 1500, int(sc["lens"][src])]))
        ln = min(ln, int(sc["lens"][src]))
        off = int(rng.integers(0, int(sc["lens"][src]) - ln + 1))
        reads[i, :ln] = sc["reads"][src, off:off + ln]
        lens[i] = ln
    best, _ = oi.seed_batch(reads, lens, sc["seed_len"], sc["thres"])
    best = best.copy()
    best["key"][7] = (1 << 64) - 5            # fenced: wrapped diagonal
    r_cpu = reads.copy()
    want = oi.extend_batch(r_cpu, lens, best, (320, 120, 128))
    r_gpu = reads.copy()
    got = mapper.extend_batch(di, r_gpu, lens, best, (320, 120, 128))
    assert np.array_equal(got["meta_r"], want["meta_r"]) and (want["meta_r"] == 0).any()
    assert np.array_equal(got["score"], want["score"]) and np.array_equal(got["n_ops"], want["n_ops"])
    for i in range(n):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), i
    assert np.array_equal(r_gpu, r_cpu)


def test_host_batches_are_sliced_without_changing_results(dev_indexes, map_options):
    """Caller batches above ~32 GB of device scratch go through the device in slices (slice_reads forces 7 reads
    per slice here): same best[], scores, ops, rev-comped reads as in one pass."""
    sc, di, oi = dev_indexes("ont-2k")
    best = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    r1 = sc["reads"].copy()
    e1 = mapper.extend_batch(di, r1, sc["lens"], best)
    # slices of 7 reads; then one slice cut into 5 pipelined sub-batches (upload k+1 / kernels k / download k-1)
    for env in ({"slice_reads": 7}, {"sub_batches": 5}, {"slice_reads": 23, "sub_batches": 2}):
        map_options(di, **env)
        best2 = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        r2 = sc["reads"].copy()
        e2 = mapper.extend_batch(di, r2, sc["lens"], best2)
        map_options(di)
        assert np.array_equal(best, best2) and np.array_equal(r1, r2), env
        assert np.array_equal(e1["score"], e2["score"]) and np.array_equal(e1["n_ops"], e2["n_ops"]), env
        for i in range(len(best)):
            k = int(e1["n_ops"][i])
            assert bytes(e1["ops"][i, :k]) == bytes(e2["ops"][i, :k]), (env, i)
        for f in ("loc", "off", "seq_id", "strand"):
            assert np.array_equal(e1["meta"][f], e2["meta"][f]), env


def test_blob_roundtrip_and_adopt(dev_indexes, gpu):
    """The image broadcast path: pack on the host, move as bytes, adopt on the device."""
    import torch
    sc, di, oi = dev_indexes("clean-1k")
    blob = sc["hi"].pack_blob()
    t = torch.from_numpy(blob).cuda()
    d2 = index.DeviceIndex.adopt(t, gpu)
    a = mapper.seed_batch(di, sc["reads"], sc["lens"])
    b = mapper.seed_batch(d2, sc["reads"], sc["lens"])
    assert np.array_equal(a, b)
    d2.close()



def test_long_interval_side_table(gpu):
    """lchash intervals too long for the 24-bit count of the 8-byte device entries go through a sorted
    side table; the packer's threshold knob sends ordinary repeats there so the path is exercised."""
    sc = workloads.scenario("repeats-ties")
    oi = orc.OracleIndex.from_host_index(sc["hi"])
    want, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    di = index.DeviceIndex.upload(sc["hi"], gpu, lcx_threshold=30)
    got = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    assert np.array_equal(got, want)
    di.close()


@pytest.mark.parametrize("name", ["clean-1k", "ont-2k", "last-phase-break", "ragged"])
def test_round_policy_does_not_change_results(dev_indexes, map_options, name):
    """Phase 0 first and phases 1..s for the undecided reads (two rounds), or all phases at once (what the library
    switches to when the previous batch decided almost nothing in phase 0): same best[], because evaluating phases
    speculatively and replaying alnmain.c:371-403 in order is exact."""
    sc, di, oi = dev_indexes(name)
    want, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    for rounds in (2, 1, 2):
        map_options(di, seed_rounds=rounds)
        got = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        assert np.array_equal(got, want), (name, rounds)
    map_options(di)
    for _ in range(3):                     # the adaptive policy: the second and third call see the first one's history
        assert np.array_equal(mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"]), want)


def _seed_both(hi, reads, lens, gpu, thres=300, **mopts):
    oi = orc.OracleIndex.from_host_index(hi)
    want, _ = oi.seed_batch(reads, lens, 20, thres)
    di = index.DeviceIndex.upload(hi, gpu)
    try:
        di.set_map_options(**mopts)
        got = mapper.seed_batch(di, reads, lens, 20, thres)
    finally:
        di.close()
    return oi, want, got


@pytest.mark.parametrize("exact_only", [0, 1])
def test_vote_wave_tier_at_its_limit(gpu, exact_only):
    """An item with exactly T1_LIMIT = 192 hits from exactly T1_LIMIT / 2 = 96 repeat seeds of two hits each: the
    wavefront tier's staging arrays (96 entries, sized on "a repeat seed has at least two hits") are full to the last
    slot.  96 distinct 20-mers sit twice each in the reference; the read strings them together 21 bases apart, so that
    phase 0 sees them all and nothing else."""
    rng = np.random.default_rng(11)
    kmers = [bytes(rng.choice(list(b"ACGT"), size=20).astype(np.uint8)) for _ in range(96)]
    parts = []
    for rep in range(2):
        for i, k in enumerate(kmers):
            parts.append(bytes(synth.reference(80 + (i * 7 + rep * 13) % 40, seed=1000 + 2 * i + rep)))
            parts.append(k)
    parts.append(bytes(synth.reference(500, seed=5)))
    ref = np.frombuffer(b"".join(parts), dtype=np.uint8)
    hi = index.HostIndex.build([ref], hlen=8)
    read = b"".join(k + b"ACGT"[i % 4:i % 4 + 1] for i, k in enumerate(kmers)) + bytes(synth.reference(25, seed=77))
    reads = np.zeros((3, len(read) + 1), dtype=np.uint8)
    lens = np.array([len(read), len(read), 900], dtype=np.uint32)
    reads[0, :len(read)] = np.frombuffer(read, dtype=np.uint8)
    reads[1] = reads[0]
    reads[2, :900] = ref[3000:3900]
    oi, want, got = _seed_both(hi, reads, lens, gpu, vote_exact_only=exact_only)
    tr = oi.seed_read(read, 20, 300, trace=True)
    ph0 = [(j, rr) for j, rr, _, _ in tr["seeds"] if j % 21 == 0]
    assert sum(rr for _, rr in ph0) == 192 and sum(rr == 2 for _, rr in ph0) == 96      # H == T1_LIMIT, 96 repeat seeds
    assert np.array_equal(got, want)


@pytest.mark.parametrize("exact_only", [0, 1])
def test_vote_key_scratch_at_its_capacity(gpu, exact_only):
    """An item with exactly LRM_VOTE_KC_CAP = 16384 hits (64 seeds x 256 copies of a tandem block): the multi-pass tier
    keeps the keys of such an item in its workgroup's slice of the key scratch, which is then full to the last entry
    (22 passes over the 1024-slot table).  Past the capacity (257 x 64 and 250 x 130 hits) the item votes in one pass into
    a table in global memory (a slice of the workspace's pool)."""
    flank = lambda s: bytes(synth.reference(700, seed=s))
    for copies, nseeds in ((256, 64), (257, 64), (250, 130)):       # 16384, 16448 and 32500 hits in the phase-0 item
        block = bytes(synth.reference(nseeds * 21, seed=3))
        ref = np.frombuffer(flank(1) + block * copies + flank(2), dtype=np.uint8)
        hi = index.HostIndex.build([ref], hlen=8)
        read = block + bytes(synth.reference(30, seed=9))
        nrep = 80 if copies == 257 else 1           # more such items than the pool has slices (32): slices are handed on
        reads = np.zeros((nrep + 1, len(read) + 1), dtype=np.uint8)
        reads[:nrep, :len(read)] = np.frombuffer(read, dtype=np.uint8)
        reads[nrep, :700] = np.frombuffer(flank(1), dtype=np.uint8)
        lens = np.array([len(read)] * nrep + [700], dtype=np.uint32)
        oi, want, got = _seed_both(hi, reads, lens, gpu, vote_exact_only=exact_only)
        tr = oi.seed_read(read, 20, 300, trace=True)
        assert sum(rr for j, rr, _, _ in tr["seeds"] if j % 21 == 0) == nseeds * copies
        assert np.array_equal(got, want), (copies, nseeds)


def test_exact_vote_kernel_alone_equals_the_fast_path(dev_indexes, map_options):
    """vote_exact_only = 1 skips the fast vote kernel: same best[] on every scenario (the fast kernel only ever settles
    an item when its result is the exact one)."""
    for name in workloads.SEED_SCENARIOS:
        sc, di, oi = dev_indexes(name)
        a = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        map_options(di, vote_exact_only=1)
        b = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        map_options(di)
        assert np.array_equal(a, b), name


def test_oracle_and_product_each_build_their_own_index(gpu):
    """Everywhere else the oracle ADOPTS the index the product's host builder made (OracleIndex.from_host_index), so a
    wrong suffix array / BWT / lchash would be wrong on both sides.  Here the oracle builds its own index with its own
    suffix sorter (orc_index_build) from the same multi-sequence text, the product builds its own, the two are compared
    array by array, and the mapping results are compared on top."""
    seqs = [synth.reference(700_000, seed=51, repeat_frac=0.05, rep_len=300, rep_copies=40, rep_div=0.05),
            synth.reference(350_000, seed=52), synth.reference(60_000, seed=53)]
    hi = index.HostIndex.build(seqs, names=["a", "b", "c"], hlen=10)
    oi = orc.OracleIndex.build(seqs, o_ratio=32, hlen=10)
    assert hi.length == oi.length >= 2_000_000
    assert np.array_equal(hi.sa(), oi.sa()) and np.array_equal(hi.bwt(), oi.bwt())
    assert np.array_equal(hi.c(), oi.c()) and np.array_equal(hi.o(), oi.o()) and np.array_equal(hi.lc(), oi.lc())
    r = synth.reads(seqs, 96, 3000, synth.ONT, seed=7)
    want_best, _ = oi.seed_batch(r["reads"], r["lens"], nthreads=8)
    rc = r["reads"].copy()
    want = oi.extend_batch(rc, r["lens"], want_best, nthreads=8)
    di = index.DeviceIndex.upload(hi, gpu)
    try:
        rg = r["reads"].copy()
        got = mapper.map_batch(di, rg, r["lens"])
    finally:
        di.close()
    assert np.array_equal(got["best"], want_best) and np.array_equal(got["score"], want["score"]) and np.array_equal(rg, rc)
    assert np.array_equal(got["n_ops"], want["n_ops"])
    for i in range(len(want_best)):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), i
