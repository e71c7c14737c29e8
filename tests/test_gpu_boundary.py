"""GPU tests of the drop-in boundary beyond the two per-part calls: the fused one-pass call, pinned caller
buffers, the sticky kernel-error word, multi-GPU group handles, the sampled-SA locate mode and suffix-array
values beyond 32 bits.  Everything goes through the C-ABI (ctypes) and is compared with the CPU oracle."""
import numpy as np
import pytest

import orc
import workloads
from longreadmapper_amd import capi, index, mapper, synth

pytestmark = pytest.mark.gpu


def _assert_ext_equal(got, want, n, what=""):
    assert np.array_equal(got["meta_r"], want["meta_r"]), what
    for f in ("loc", "off", "seq_id", "strand"):
        assert np.array_equal(got["meta"][f], want["meta"][f]), (what, f)
    assert np.array_equal(got["score"], want["score"]), what
    assert np.array_equal(got["n_ops"], want["n_ops"]), what
    for i in range(n):
        k = int(want["n_ops"][i])
        assert bytes(got["ops"][i, :k]) == bytes(want["ops"][i, :k]), (what, i)


@pytest.fixture(scope="module")
def ont(gpu):
    sc = workloads.scenario("ont-2k")
    di = index.DeviceIndex.upload(sc["hi"], gpu)
    oi = orc.OracleIndex.from_host_index(sc["hi"])
    best, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    r_cpu = sc["reads"].copy()
    ext = oi.extend_batch(r_cpu, sc["lens"], best)
    yield sc, di, oi, best, ext, r_cpu
    di.close()


def _dense_rows(got):
    """A dense-layout result as rows, so that _assert_ext_equal can compare it."""
    n = len(got["n_ops"])
    width = int(got["n_ops"].max()) if n else 0
    rows = np.zeros((n, max(width, 1)), dtype=np.uint8)
    for i in range(n):
        k = int(got["n_ops"][i])
        rows[i, :k] = np.frombuffer(mapper.ops_of(got, i), dtype=np.uint8)
    return dict(got, ops=rows)


@pytest.mark.parametrize("opts", [None, {}, {"sub_batches": 4}, {"slice_reads": 9, "sub_batches": 2},
                                  {"dense_results": 1}, {"dense_results": 1, "slice_reads": 13, "sub_batches": 3, "group_subs": 1},
                                  {"keep_reads": 1}, {"dense_results": 1, "keep_reads": 1, "sub_batches": 3}])
def test_map_batch_equals_the_two_calls_and_the_oracle(ont, opts):
    """lrm_map_batch = lrm_seed_batch + lrm_extend_batch in one device pass (one upload of the reads), in every result
    mode of lrm_map_options: rows / dense, any slicing."""
    sc, di, oi, best, ext, r_cpu = ont
    r = sc["reads"].copy()
    got = mapper.map_batch(di, r, sc["lens"], sc["seed_len"], sc["thres"], options=opts)
    if opts and opts.get("dense_results"):
        # cig[i].cigar points into store_mem, 16-byte aligned, ascending, inside the rows of the read's group
        off = got["ops_off"][got["n_ops"] > 0]
        assert (off % 16 == 0).all() and (np.diff(off) > 0).all()
        got = _dense_rows(got)
    assert np.array_equal(got["best"], best)
    _assert_ext_equal(got, ext, len(best), str(opts))
    if opts and opts.get("keep_reads"):
        # the caller's buffer comes back as it went; strand and meta_r say which reads the reference would have
        # reverse-complemented in place (alnmain.c:437) -- exactly the rows in which the oracle's buffer differs
        assert np.array_equal(r, sc["reads"])
        rev = (got["meta_r"] != 0) & (got["meta"]["strand"] == 1)
        changed = (r_cpu != sc["reads"]).any(axis=1)
        assert rev.any() and np.array_equal(changed, rev & (sc["lens"] > 0))
    else:
        assert np.array_equal(r, r_cpu)                  # reverse-strand reads rev-comped in the caller's buffer


@pytest.mark.parametrize("opts", [{}, {"dense_results": 1}])
@pytest.mark.parametrize("pinned", [False, True])
def test_two_batches_in_flight(ont, opts, pinned):
    """lrm_map_batch_submit / lrm_map_batch_wait: three batches submitted back to back (two in flight on the device,
    the third queued), each with its own caller buffers; every one of them equals the synchronous call and the
    oracle, whatever the order of the waits."""
    sc, di, oi, best, ext, r_cpu = ont
    n, stride = sc["reads"].shape
    bufs, pend = [], []
    try:
        for k in range(3):
            if pinned:
                r = mapper.pinned_empty((n, stride))
                st = mapper.pinned_empty((n, (2 * (stride - 1) + 15) // 16 * 16))
                st[:] = 0
            else:
                r, st = np.empty((n, stride), dtype=np.uint8), None
            r[:] = sc["reads"]
            bufs.append((r, st))
            pend.append(mapper.map_batch_submit(di, r, sc["lens"], sc["seed_len"], sc["thres"], store=st, options=opts))
        for k in (1, 0, 2):
            got = pend[k].wait()
            if opts.get("dense_results"):
                got = _dense_rows(got)
            assert np.array_equal(got["best"], best)
            _assert_ext_equal(got, ext, n, "batch %d %s" % (k, opts))
            assert np.array_equal(bufs[k][0], r_cpu)
    finally:
        for p in pend:
            if p.ticket is not None:
                p.wait()
        if pinned:
            for r, st in bufs:
                mapper.pinned_free(r)
                mapper.pinned_free(st)


@pytest.mark.parametrize("opts", [{"cigar_text": 1}, {"cigar_text": 1, "slice_reads": 11, "sub_batches": 2, "group_subs": 1}])
@pytest.mark.parametrize("pinned", [False, True])
def test_cigar_text_layout(ont, opts, pinned):
    """lrm_map_options.cigar_text: parse_cigar (alnmain.c:497-498) on the device -- cig[i].cigar points to the
    NUL-terminated run-length text of the read's op bytes ('=' and 'X' columns as M), "*" for a read without an
    alignment; everything else as in the other layouts."""
    import sam_ref
    sc, di, oi, best, ext, r_cpu = ont
    n, stride = sc["reads"].shape
    r = mapper.pinned_empty((n, stride)) if pinned else np.empty((n, stride), dtype=np.uint8)
    st = mapper.pinned_empty((n, (2 * (stride - 1) + 15) // 16 * 16)) if pinned else None
    try:
        r[:] = sc["reads"]
        got = mapper.map_batch(di, r, sc["lens"], sc["seed_len"], sc["thres"], store=st, options=opts)
        assert np.array_equal(got["best"], best) and np.array_equal(got["score"], ext["score"]) and np.array_equal(got["n_ops"], ext["n_ops"])
        assert np.array_equal(got["meta_r"], ext["meta_r"]) and np.array_equal(r, r_cpu)
        for i in range(n):
            k = int(ext["n_ops"][i])
            none = k <= 0 or ext["meta_r"][i] == 0 or ext["score"][i] == -1
            assert mapper.text_of(got, i).decode() == ("*" if none else sam_ref.rle(bytes(ext["ops"][i, :k]))), i
    finally:
        if pinned:
            mapper.pinned_free(r)
            mapper.pinned_free(st)


def test_cigar_text_of_long_and_odd_runs(gpu):
    """The device's run-length pass on op strings that cross its 4096-column chunks and its 16-column lanes with runs of
    every length: identical reads (one run of thousands of '='), a read with a long insertion, reads of 1..40 bases."""
    import sam_ref
    ref = synth.reference(60_000, seed=3)
    hi = index.HostIndex.build([ref], hlen=8)
    di = index.DeviceIndex.upload(hi, gpu)
    oi = orc.OracleIndex.from_host_index(hi)
    try:
        lens = [9000, 8191, 8192, 8193, 4096, 4097, 12000] + list(range(21, 41))
        n, mx = len(lens), max(lens)
        reads = np.zeros((n, mx + 1), dtype=np.uint8)
        rng = np.random.default_rng(1)
        for i, l in enumerate(lens):
            p0 = 1000 + 37 * i
            seq = ref[p0:p0 + l].copy()
            if i == 6:                                    # a 3000-base insertion in the middle: a long run of I
                seq[4000:7000] = rng.choice(list(b"ACGT"), size=3000)
            reads[i, :l] = seq
        lens = np.array(lens, dtype=np.uint32)
        want_best, _ = oi.seed_batch(reads, lens)
        rc = reads.copy()
        want = oi.extend_batch(rc, lens, want_best)
        rg = reads.copy()
        got = mapper.map_batch(di, rg, lens, options={"cigar_text": 1})
        assert np.array_equal(got["best"], want_best) and np.array_equal(got["n_ops"], want["n_ops"])
        for i in range(n):
            k = int(want["n_ops"][i])
            none = k <= 0 or want["meta_r"][i] == 0 or want["score"][i] == -1
            assert mapper.text_of(got, i).decode() == ("*" if none else sam_ref.rle(bytes(want["ops"][i, :k]))), i
        assert mapper.text_of(got, 0) == b"9000M"
    finally:
        di.close()


def test_dense_results_need_an_aligned_stride(ont):
    sc, di, oi, best, ext, r_cpu = ont
    r = sc["reads"].copy()
    n, stride = r.shape
    st = np.zeros((n, 2 * (stride - 1) + 6), dtype=np.uint8)
    with pytest.raises(capi.LrmError, match="multiple of 16"):
        mapper.map_batch(di, r, sc["lens"], sc["seed_len"], sc["thres"], store=st, options={"dense_results": 1})


def test_map_batch_with_pinned_caller_buffers(ont):
    """Buffers from lrm_host_alloc are handed to the DMA engines directly (no staging copy): same results in the row
    layout and in the dense one (op bytes DMA'd straight into the pinned store_mem)."""
    sc, di, oi, best, ext, r_cpu = ont
    n, stride = sc["reads"].shape
    r = mapper.pinned_empty((n, stride))
    store = mapper.pinned_empty((n, 2 * (stride - 1)))
    try:
        r[:] = sc["reads"]
        store[:] = 0
        got = mapper.map_batch(di, r, sc["lens"], sc["seed_len"], sc["thres"], store=store)
        assert np.array_equal(got["best"], best)
        _assert_ext_equal(got, ext, n, "pinned")
        assert np.array_equal(r, r_cpu)
        r[:] = sc["reads"]
        dstore = mapper.pinned_empty((n, (2 * (stride - 1) + 15) // 16 * 16))
        try:
            dstore[:] = 0
            got = mapper.map_batch(di, r, sc["lens"], sc["seed_len"], sc["thres"], store=dstore, options={"dense_results": 1})
            assert np.array_equal(got["best"], best) and np.array_equal(r, r_cpu)
            _assert_ext_equal(_dense_rows(got), ext, n, "pinned, dense")
        finally:
            mapper.pinned_free(dstore)
        # registered caller memory (what a maintainer does with the malloc'd buffers of alnmain.c:297-320)
        r2 = np.ascontiguousarray(sc["reads"].copy())
        capi.check(capi.lib.lrm_host_register(r2.ctypes.data, r2.nbytes), "lrm_host_register")
        try:
            got2 = mapper.map_batch(di, r2, sc["lens"], sc["seed_len"], sc["thres"])
        finally:
            capi.check(capi.lib.lrm_host_unregister(r2.ctypes.data), "lrm_host_unregister")
        assert np.array_equal(got2["best"], best) and np.array_equal(r2, r_cpu)
        _assert_ext_equal(got2, ext, n, "registered")
    finally:
        mapper.pinned_free(r)
        mapper.pinned_free(store)


def test_vote_overflow_is_reported_from_any_sub_batch(gpu):
    """lrm_debug_set_vote_limits: a pass limit above the table size forces the multi-pass tier into one pass and a small
    table: items with more distinct buckets than slots overflow.  The error word is sticky, so an overflow in sub-batch 0 of 4 fails the call
    (it used to be erased by the next launch), and a *_dev caller gets it on the next call."""
    import torch
    sc = dict(workloads.scenario("repeats-overflow"))
    di = index.DeviceIndex.upload(sc["hi"], gpu)
    n = len(sc["lens"])
    # only sub-batch 0 of 4 can overflow: reads 10.. are replaced by random sequences (no seed hits at all)
    reads = sc["reads"].copy()
    rnd = synth.reference(reads.shape[1] * (n - 10), seed=77).reshape(n - 10, reads.shape[1])
    for i in range(10, n):
        reads[i, :sc["lens"][i]] = rnd[i - 10, :sc["lens"][i]]
    sc["reads"] = reads
    try:
        di.set_map_options(sub_batches=4)
        di.debug_set_vote_limits(1000000, 64)
        with pytest.raises(capi.LrmError, match="vote table overflow"):
            mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        # the handle stays usable and the flag does not leak into the next (good) batch
        di.debug_set_vote_limits(0, 0)
        oi = orc.OracleIndex.from_host_index(sc["hi"])
        want, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        got = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        assert np.array_equal(got, want)
        # device-buffer API: the faulty batch returns 0 (asynchronous), the next call on the workspace fails
        di.debug_set_vote_limits(1000000, 64)
        stride = sc["reads"].shape[1]
        dm = mapper.DeviceMapper(di, n, stride - 1, sc["seed_len"], sc["thres"], device=gpu)
        d_reads = torch.from_numpy(sc["reads"]).cuda()
        d_lens = torch.from_numpy(sc["lens"].astype(np.int32)).cuda()
        dm.seed(d_reads, d_lens)
        torch.cuda.synchronize()
        di.debug_set_vote_limits(0, 0)
        with pytest.raises(capi.LrmError, match="vote table overflow"):
            dm.seed(d_reads, d_lens)
        dm.seed(d_reads, d_lens)                         # cleared by the failing call
        torch.cuda.synchronize()
        assert np.array_equal(dm.results(n)["best"], want)
        dm.close()
    finally:
        di.close()


@pytest.mark.parametrize("ngpus", [2, 3])
def test_multi_gpu_group_handle_equals_one_gpu(ont, gpu, ngpus):
    """lrm_index_upload_multi with the one visible device listed N times: N logical replicas, the batch is cut
    into N slices by bases, one host thread per replica, results written in place -- byte-identical with N=1.
    (Real multi-device runs are the driver's; this covers the code path and the partition.)"""
    sc, di, oi, best, ext, r_cpu = ont
    dg = index.DeviceIndex.upload_multi(sc["hi"], [gpu] * ngpus)
    try:
        assert dg.replicas == ngpus
        # ragged lengths so that "by bases" differs from "by reads"
        lens = sc["lens"].copy()
        lens[: len(lens) // 3] //= 4
        reads = sc["reads"].copy()
        for i, l in enumerate(lens):
            reads[i, l:] = 0
        b1 = mapper.seed_batch(di, reads, lens, sc["seed_len"], sc["thres"])
        bN = mapper.seed_batch(dg, reads, lens, sc["seed_len"], sc["thres"])
        assert np.array_equal(b1, bN)
        r1, rN, rM = reads.copy(), reads.copy(), reads.copy()
        e1 = mapper.extend_batch(di, r1, lens, b1)
        eN = mapper.extend_batch(dg, rN, lens, bN)
        _assert_ext_equal(eN, e1, len(lens), "extend on the group")
        assert np.array_equal(r1, rN)
        m = mapper.map_batch(dg, rM, lens, sc["seed_len"], sc["thres"])
        assert np.array_equal(m["best"], b1) and np.array_equal(rM, r1)
        _assert_ext_equal(m, e1, len(lens), "map on the group")
        want, _ = oi.seed_batch(reads, lens, sc["seed_len"], sc["thres"])
        assert np.array_equal(b1, want)
    finally:
        dg.close()


@pytest.mark.parametrize("name", ["ont-2k", "repeats-ties", "repeats-overflow", "ragged"])
@pytest.mark.parametrize("ratio", [4, 32])
def test_sampled_sa_locate_mode(gpu, name, ratio):
    """lrm_index_options.sa_sampled = r keeps SA rows i*r only (the reference's csa table, fmidx.c:153-163) and locates the
    rest by LF steps on the device (csa_access, fmidx.c:315-331, with the textbook LF -- see seed_kernels.hip):
    identical best[] to the full-SA mode and to the oracle, with 1/r of the SA bytes in HBM."""
    sc = workloads.scenario(name)
    oi = orc.OracleIndex.from_host_index(sc["hi"])
    want, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
    full = sc["hi"].blob_bytes()
    assert sc["hi"].blob_bytes(sa_sampled=ratio) < full
    di = index.DeviceIndex.upload(sc["hi"], gpu, sa_sampled=ratio)
    try:
        got = mapper.seed_batch(di, sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        assert np.array_equal(got, want)
    finally:
        di.close()


@pytest.mark.parametrize("opts", [{"dense_results": 1}, {"cigar_text": 1}, {}])
def test_group_handle_with_batches_in_flight(ont, gpu, opts):
    """lrm_map_batch_submit on a multi-GPU group handle (three logical replicas on the one device): every batch is cut
    by bases into one share per replica, each replica's issuer / collector threads run their share, one ticket completes
    when all have; two batches in flight plus one queued, in every result layout -- byte-identical with one GPU."""
    import sam_ref
    sc, di, oi, best, ext, r_cpu = ont
    dg = index.DeviceIndex.upload_multi(sc["hi"], [gpu] * 3)
    try:
        n = len(best)
        bufs = [sc["reads"].copy() for _ in range(3)]
        pend = [mapper.map_batch_submit(dg, b, sc["lens"], sc["seed_len"], sc["thres"], options=opts) for b in bufs]
        for k in (2, 0, 1):
            got = pend[k].wait()
            assert np.array_equal(got["best"], best) and np.array_equal(bufs[k], r_cpu)
            assert np.array_equal(got["score"], ext["score"]) and np.array_equal(got["n_ops"], ext["n_ops"])
            if opts.get("cigar_text"):
                for i in range(n):
                    kk = int(ext["n_ops"][i])
                    none = kk <= 0 or ext["meta_r"][i] == 0 or ext["score"][i] == -1
                    assert mapper.text_of(got, i).decode() == ("*" if none else sam_ref.rle(bytes(ext["ops"][i, :kk]))), (k, i)
            else:
                _assert_ext_equal(_dense_rows(got) if opts.get("dense_results") else got, ext, n, "group, batch %d %s" % (k, opts))
    finally:
        dg.close()


def test_suffix_array_values_beyond_32_bits(gpu):
    """GRCh38's .cat has 6.2 G rows: SA values need ui40.high (sa_use.h:17-29).  A small FM index whose SA values
    are all shifted by 2^33 + 2^36 pushes 37-bit values through lrm_index_upload (ui40 -> u64), the SA gathers,
    the u64 diagonals, the LDS vote keys / buckets and best[] -- compared with the oracle on the same shifted SA."""
    seqs = [synth.reference(150_000, seed=21)]
    hi = index.HostIndex.build(seqs, o_ratio=32, hlen=8)
    shift = (1 << 33) + (1 << 36)
    raw = hi.sa_raw()
    raw += np.uint64(shift)                              # ui40 {low, high}: bits 32..39 are `high`
    assert int(hi.sa().min()) >= shift
    r = synth.reads(seqs, 48, 1500, synth.ONT, seed=5)
    oi = orc.OracleIndex.from_host_index(hi)
    want, _ = oi.seed_batch(r["reads"], r["lens"], 20, 300)
    assert (want["key"] >= np.uint64(1 << 33)).sum() > 40
    di = index.DeviceIndex.upload(hi, gpu)
    try:
        got = mapper.seed_batch(di, r["reads"], r["lens"], 20, 300)
        assert np.array_equal(got, want)
    finally:
        di.close()
    # the same through the sampled-SA mode (values are added to LF step counts after the gather)
    di = index.DeviceIndex.upload(hi, gpu, sa_sampled=4)
    try:
        # sampled rows hold shifted values; unsampled rows add their step count: still the shifted SA
        got = mapper.seed_batch(di, r["reads"], r["lens"], 20, 300)
        assert np.array_equal(got, want)
    finally:
        di.close()


def test_rccl_is_loadable_and_bound(gpu):
    """The multi-device index broadcast goes through RCCL (dlopen'ed: librccl.so.1); a one-GPU box cannot run it,
    so the same code is driven on a 1-rank communicator: library load, ncclCommInitAll, grouped ncclBroadcast in
    256 MiB pieces, ncclCommDestroy."""
    rc = capi.lib.lrm_debug_rccl_selftest(gpu, (300 << 20) + 12345)
    assert rc == 0, (rc, capi.lib.lrm_last_error())


def test_concurrent_calls_on_one_handle_and_on_two(ont, gpu):
    """"re-entrant per handle, no hidden globals".  Four host threads call lrm_map_batch at the same
    time -- two on one handle (serialised by the handle's own mutex), one each on two more handles (independent) --
    and every call returns the single-threaded result."""
    import threading
    sc, di, oi, best, ext, r_cpu = ont
    others = [index.DeviceIndex.upload(sc["hi"], gpu) for _ in range(2)]
    out, errs = {}, []

    def work(tag, handle):
        try:
            for rep in range(3):
                r = sc["reads"].copy()
                got = mapper.map_batch(handle, r, sc["lens"], sc["seed_len"], sc["thres"])
                assert np.array_equal(got["best"], best) and np.array_equal(r, r_cpu)
                _assert_ext_equal(got, ext, len(best), tag)
            out[tag] = True
        except Exception as e:          # noqa: BLE001 -- reported below with the thread's tag
            errs.append((tag, repr(e)))

    th = [threading.Thread(target=work, args=(t, h)) for t, h in (("a0", di), ("a1", di), ("b", others[0]), ("c", others[1]))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in others:
        o.close()
    assert not errs, errs
    assert len(out) == 4


def test_map_batch_degenerate_and_ragged_batches(gpu):
    """The fused call on the shapes a host loop can hand over: an empty batch, a batch of one, only empty reads, the
    ragged scenario (empty, shorter than a seed, exactly seed_len, ...) in rows WIDER than max_len + 1 with op rows
    wider than 2 * max_len -- all against the oracle, bytes outside the used part of a row untouched."""
    sc = workloads.scenario("ragged")
    di = index.DeviceIndex.upload(sc["hi"], gpu)
    oi = orc.OracleIndex.from_host_index(sc["hi"])
    try:
        n, stride = sc["reads"].shape
        got = mapper.map_batch(di, np.zeros((0, 8), dtype=np.uint8), np.zeros(0, dtype=np.uint32))
        assert len(got["best"]) == 0 and len(got["score"]) == 0
        wide = np.full((n, stride + 16), 0x5A, dtype=np.uint8)           # caller rows with slack behind the NUL padding
        wide[:, :stride] = sc["reads"]
        store = np.full((n, 2 * (stride - 1) + 40), 0xA5, dtype=np.uint8)
        best, _ = oi.seed_batch(sc["reads"], sc["lens"], sc["seed_len"], sc["thres"])
        r_cpu = sc["reads"].copy()
        ext = oi.extend_batch(r_cpu, sc["lens"], best)
        got = mapper.map_batch(di, wide, sc["lens"], sc["seed_len"], sc["thres"], store=store)
        assert np.array_equal(got["best"], best)
        _assert_ext_equal(got, ext, n, "ragged, wide rows")
        assert np.array_equal(wide[:, :stride], r_cpu) and (wide[:, stride:] == 0x5A).all()
        for i in range(n):
            assert (store[i, int(ext["n_ops"][i]):] == 0xA5).all(), i      # nothing written beyond n_cigar_op
        for k in (1, 2):                                                   # a batch of one or two reads
            r1 = sc["reads"][20:20 + k].copy()
            g1 = mapper.map_batch(di, r1, sc["lens"][20:20 + k], sc["seed_len"], sc["thres"])
            assert np.array_equal(g1["best"], best[20:20 + k]) and np.array_equal(g1["score"], ext["score"][20:20 + k])
        z = np.zeros((5, 33), dtype=np.uint8)                              # only empty reads: no seeds, 'I' tails of length 0
        gz = mapper.map_batch(di, z, np.zeros(5, dtype=np.uint32), sc["seed_len"], sc["thres"])
        bz, _ = oi.seed_batch(z, np.zeros(5, dtype=np.uint32), sc["seed_len"], sc["thres"])
        ez = oi.extend_batch(z.copy(), np.zeros(5, dtype=np.uint32), bz)
        assert np.array_equal(gz["best"], bz) and np.array_equal(gz["score"], ez["score"]) and np.array_equal(gz["meta_r"], ez["meta_r"])
    finally:
        di.close()
