"""The device image ("blob") of an index, checked on the CPU against a numpy model of the layout documented in
longreadmapper_amd/csrc/lrm_internal.h: occ blocks from the bwt, the permuted 8-byte lc entries, SA values
(full / sampled, beyond 32 bits), text, mta.  No GPU involved: lrm_index_pack_blob is pure host code, and
lrm_index_upload emits exactly the same pieces."""
import numpy as np
import pytest

from longreadmapper_amd import capi, index, synth

HDR = 256


def _header(blob):
    h = np.frombuffer(blob[:HDR].tobytes(), dtype="<u8")
    f = dict(magic=h[0], version=h[1], length=h[2], c4=h[3:7], dollar_row=h[7], n_blocks=h[8], lc_entries=h[9],
             sa_len=h[10], con_len=h[11], off_occ=h[12], off_lc=h[13], off_sa=h[14], off_content=h[15], off_mta=h[16],
             total_bytes=h[17], off_lcx=h[19], n_lcx=h[20], sa_ratio=h[21])
    f["hlen"], f["mta_len"] = np.frombuffer(blob[18 * 8:19 * 8].tobytes(), dtype="<i4")
    return {k: (int(v) if np.ndim(v) == 0 else v.astype(np.uint64)) for k, v in f.items()}


def _model_occ(bwt, n_blocks, c4):
    L = len(bwt)
    out = np.zeros((n_blocks, 4, 2), dtype=np.uint64)
    pad = np.zeros(n_blocks * 64, dtype=np.uint8)
    pad[:L] = bwt
    for c, ch in enumerate(b"ACGT"):
        is_c = (pad == ch).reshape(n_blocks, 64)
        out[:, c, 0] = np.uint64(c4[c]) + np.concatenate([[0], np.cumsum(is_c.sum(axis=1))[:-1]]).astype(np.uint64)   # C[c] + prefix
        out[:, c, 1] = (is_c.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1)
    return out


def _rev_groups(v, hl):
    code = np.zeros_like(v)
    for _ in range(hl):
        code = (code << np.uint64(2)) | (v & np.uint64(3))
        v = v >> np.uint64(2)
    return code


@pytest.mark.parametrize("n,hlen", [(1000, 4), (70_000, 7), (2_200_000, 9)])     # the last spans several bwt segments
def test_image_matches_numpy_model(n, hlen):
    hi = index.HostIndex.build([synth.reference(n, seed=n % 97, repeat_frac=0.05, rep_len=100, rep_copies=20)], hlen=hlen)
    blob = hi.pack_blob()
    h = _header(blob)
    L = hi.length
    assert h["length"] == L and h["total_bytes"] == len(blob) == hi.blob_bytes()
    assert h["sa_ratio"] == 1 and h["sa_len"] == L
    bwt = hi.bwt()
    assert h["dollar_row"] == int(np.nonzero(bwt == ord("$"))[0][0])
    occ = np.frombuffer(blob[h["off_occ"]:h["off_occ"] + h["n_blocks"] * 64].tobytes(), dtype="<u8").reshape(-1, 4, 2)
    c_arr = hi.c()
    assert [int(x) for x in h["c4"]] == [int(c_arr[ord(ch)]) for ch in "ACGT"]
    assert np.array_equal(occ, _model_occ(bwt, h["n_blocks"], h["c4"]))
    # lc: entry[code] = k | cnt << 40 with code = the 2-bit groups of the reference's index reversed
    lc = np.frombuffer(blob[h["off_lc"]:h["off_lc"] + h["lc_entries"] * 8].tobytes(), dtype="<u8")
    ref = hi.lc().reshape(-1, 2)
    num = np.arange(h["lc_entries"], dtype=np.uint64)
    k, l = ref[:, 0], ref[:, 1]
    want = np.where((k == 0) & (l == 0), np.uint64(0), k | ((l - k + np.uint64(1)) << np.uint64(40)))
    assert np.array_equal(lc[_rev_groups(num, hlen)], want)
    assert np.array_equal(np.frombuffer(blob[h["off_sa"]:h["off_sa"] + L * 8].tobytes(), dtype="<u8"), hi.sa())
    assert bytes(blob[h["off_content"]:h["off_content"] + L]) == bytes(hi.content())
    mta = np.frombuffer(blob[h["off_mta"]:h["off_mta"] + 16].tobytes(), dtype="<u8")
    assert (int(mta[0]), int(mta[1])) == hi.mta()[0][1:]


def test_sa_values_beyond_32_bits_survive_packing():
    """ui40.high (sa_use.h:17-29): GRCh38's 6.2 G rows need it; here every value is shifted by 2^33 + 2^39."""
    hi = index.HostIndex.build([synth.reference(5000, seed=3)], hlen=4)
    shift = (1 << 33) + (1 << 39)
    raw = hi.sa_raw()
    before = hi.sa().copy()
    raw += np.uint64(shift)
    blob = hi.pack_blob()
    h = _header(blob)
    got = np.frombuffer(blob[h["off_sa"]:h["off_sa"] + hi.length * 8].tobytes(), dtype="<u8")
    assert np.array_equal(got, before + np.uint64(shift)) and int(got.min()) >= 1 << 33


@pytest.mark.parametrize("ratio", [2, 4, 64])
def test_sampled_sa_image(ratio):
    """lrm_index_options.sa_sampled = r: the [sa] section holds rows i*r only -- for r = 4 exactly the reference's csa table
    (fmidx.c:153-163) -- and the image shrinks accordingly."""
    hi = index.HostIndex.build([synth.reference(9001, seed=4)], hlen=5)
    full = hi.blob_bytes()
    blob = hi.pack_blob(sa_sampled=ratio)
    h = _header(blob)
    L = hi.length
    assert h["sa_ratio"] == ratio and h["sa_len"] == (L + ratio - 1) // ratio and len(blob) < full
    got = np.frombuffer(blob[h["off_sa"]:h["off_sa"] + h["sa_len"] * 8].tobytes(), dtype="<u8")
    assert np.array_equal(got, hi.sa()[::ratio])
    if ratio == 4:
        assert np.array_equal(got, hi.csa()[:len(got)])
    assert hi.blob_bytes(sa_sampled=3) == full       # not a power of two: ignored, full SA
    assert hi.blob_bytes(sa_sampled=ratio) == len(blob)


def test_pair_end_is_the_reference_stub():
    assert capi.lib.lrm_pair_end(0, None) == -1            # alnmain.c:554-557


def test_group_handle_needs_a_device():
    if capi.lib.lrm_device_count() > 0:
        pytest.skip("a GPU is visible")
    hi = index.HostIndex.build([synth.reference(400, seed=1)], hlen=3)
    with pytest.raises(capi.LrmError, match="no HIP device"):
        index.DeviceIndex.upload_multi(hi, [0, 0])
