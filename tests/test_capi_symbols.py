"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares.
No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from longreadmapper_amd import capi, _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("lrm_accel.h", "lrm_index_host.h", "lrm_io_host.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(lrm_[a-z0-9_]+)\s*\(", src))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    decl = _declared()
    assert len(decl) >= 35
    lib = C.CDLL(_build.ACCEL_LIB)
    for name in sorted(decl):
        assert hasattr(lib, name), name
    assert decl == set(capi.SYMBOLS), decl ^ set(capi.SYMBOLS)


def test_abi_version_and_struct_layouts():
    assert capi.lib.lrm_abi_version() == 3
    assert C.sizeof(capi.Entry) == 24 and C.sizeof(capi.Params) == 16
    assert C.sizeof(capi.DnaFmi) == 64 and C.sizeof(capi.LcHash) == 24 and C.sizeof(capi.SaMem) == 24
    assert C.sizeof(capi.MtaEntry) == 40 and C.sizeof(capi.SeqMeta) == 24 and C.sizeof(capi.Cigar) == 16


def test_no_cpu_fallback_without_device():
    if capi.lib.lrm_device_count() > 0:
        pytest.skip("a GPU is visible")
    from longreadmapper_amd import index, synth
    hi = index.HostIndex.build([synth.reference(400, seed=1)], hlen=3)
    with pytest.raises(capi.LrmError, match="no HIP device"):
        index.DeviceIndex.upload(hi)
    q = np.frombuffer(b"ACGTACGT", dtype=np.uint8)
    ops = np.zeros(32, dtype=np.uint8)
    n_ops, score = C.c_int(), C.c_int()
    rc = capi.lib.lrm_debug_gact(q.ctypes.data, 8, q.ctypes.data, 8, capi.GactParams(0, 0, 0), ops.ctypes.data,
                                 C.byref(n_ops), C.byref(score), 0)
    assert rc < 0 and b"no HIP device" in capi.lib.lrm_last_error()


def test_result_flags_part3():
    # alnmain.c:460-474
    from longreadmapper_amd import mapper
    meta = np.zeros(4, dtype=mapper.META_DT)
    meta["strand"] = [0, 1, 1, 0]
    flag, mapq, valid = mapper.result_flags([5, 7, -1, 3], [1, 1, 1, 0], meta)
    assert flag.tolist() == [0, 16, 4, 4] and mapq.tolist() == [255, 255, 0, 0] and valid.tolist() == [1, 1, 0, 0]
