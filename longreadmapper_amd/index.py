"""Host-side index objects (CPU construction, files) and their device upload.

Mirrors the reference's index life cycle: `accidx` (asindex.c) builds the five files,
`init()` (alnmain.c:179-256) loads them; here `HostIndex.build()/read()` produce the same
in-memory structs and `DeviceIndex` holds their MI355X image."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib


def _np_view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class HostIndex:
    """dna_fmi + lc_hash + sa_mem + content + mta, in the reference's in-memory layouts."""

    def __init__(self):
        self.h = capi.HostIndex()
        self._owned = False

    @classmethod
    def build(cls, seqs, names=None, o_ratio=32, hlen=12, n_seed=7):
        """seqs: list of byte strings / uint8 arrays (forward strands)."""
        self = cls()
        bufs = [bytes(memoryview(np.ascontiguousarray(s))) if isinstance(s, np.ndarray) else bytes(s) for s in seqs]
        n = len(bufs)
        seqp = (C.c_char_p * n)(*bufs)
        namep = (C.c_char_p * n)(*[(nm.encode() if isinstance(nm, str) else nm) for nm in names]) if names else None
        lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
        cat = C.c_void_p()
        cat_len = C.c_uint64()
        mta = C.POINTER(capi.MtaEntry)()
        check(lib.lrm_cat_from_seqs(namep, seqp, lens, n, n_seed, C.byref(cat), C.byref(cat_len), C.byref(mta)),
              "lrm_cat_from_seqs")
        try:
            check(lib.lrm_host_index_build(cat, cat_len.value, mta, n, o_ratio, hlen, C.byref(self.h)),
                  "lrm_host_index_build")
        finally:
            C.CDLL(None).free(cat)
            lib.lrm_mta_free(mta, n)
        self._owned = True
        return self

    @classmethod
    def read(cls, genome):
        self = cls()
        check(lib.lrm_host_index_read(genome.encode(), C.byref(self.h)), "lrm_host_index_read")
        self._owned = True
        return self

    def write(self, genome):
        check(lib.lrm_host_index_write(C.byref(self.h), genome.encode()), "lrm_host_index_write")

    def close(self):
        if self._owned:
            lib.lrm_host_index_free(C.byref(self.h))
            self._owned = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy views (no copies)
    @property
    def length(self):
        return int(self.h.fmi.length)

    @property
    def hlen(self):
        return int(self.h.lch.hlen)

    @property
    def mta_len(self):
        return int(self.h.mta_len)

    def c(self):
        return _np_view(self.h.fmi.c, 256, np.uint64)

    def o(self):
        return _np_view(self.h.fmi.o, int(self.h.fmi.o_len), np.uint64)

    def csa(self):
        return _np_view(self.h.fmi.csa, int(self.h.fmi.csa_len), np.uint64)

    def bwt(self):
        return _np_view(self.h.fmi.bwt, self.length, np.uint8)

    def lc(self):
        return _np_view(self.h.lch.lc, int(self.h.lch.len), np.uint64)

    def sa_raw(self):
        return _np_view(self.h.sa.mem, int(self.h.sa.len), np.uint64)

    def sa(self):
        """SA values as uint64 -- a VIEW of the ui40 slots (no copy: 50 GB for GRCh38).  The builder and the .sa5
        reader of this library zero the three padding bytes of every 8-byte slot, so the slot read as a
        little-endian u64 is the value."""
        return self.sa_raw()

    def content(self):
        return _np_view(self.h.content, int(self.h.con_len), np.uint8)

    def mta(self):
        return [(self.h.mta[i].name.decode() if self.h.mta[i].name else "", int(self.h.mta[i].offset),
                 int(self.h.mta[i].seq_len)) for i in range(self.mta_len)]

    def blob_bytes(self, **opts):
        return int(lib.lrm_index_blob_bytes_opt(self.length, self.hlen, self.mta_len, C.byref(capi.index_options(**opts))))

    def pack_device(self, device=0, **opts):
        """The image packed straight into a torch uint8 CUDA tensor (no host copy): broadcast it, then adopt.
        opts: lrm_index_options fields (sa_sampled, lcx_threshold)."""
        import torch
        n = self.blob_bytes(**opts)
        t = torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", device))
        check(lib.lrm_index_pack_device_opt(C.byref(self.h.fmi), C.byref(self.h.lch), C.byref(self.h.sa), self.h.content,
                                            self.h.con_len, self.h.mta, self.h.mta_len, t.data_ptr(), n, device,
                                            C.byref(capi.index_options(**opts))), "lrm_index_pack_device_opt")
        return t

    def pack_blob(self, out=None, **opts):
        """Serialise to the device image in host memory (numpy uint8)."""
        n = self.blob_bytes(**opts)
        if out is None:
            out = np.empty(n, dtype=np.uint8)
        check(lib.lrm_index_pack_blob_opt(C.byref(self.h.fmi), C.byref(self.h.lch), C.byref(self.h.sa), self.h.content,
                                          self.h.con_len, self.h.mta, self.h.mta_len, out.ctypes.data, n,
                                          C.byref(capi.index_options(**opts))), "lrm_index_pack_blob_opt")
        return out


class DeviceIndex:
    """Device-resident index image (lrm_index*)."""

    def __init__(self, handle, keep=None):
        self.handle = handle
        self._keep = keep      # e.g. the torch tensor that owns an adopted blob

    @classmethod
    def upload(cls, host: HostIndex, device=0, **opts):
        """opts: lrm_index_options fields (sa_sampled, lc_long, lc_long_max, lc_pair, lcx_threshold)."""
        return cls.upload_multi(host, [device], **opts)

    @classmethod
    def upload_multi(cls, host: HostIndex, devices, **opts):
        """One device, or a multi-GPU group handle (lrm_index_upload_opt): one image per listed device, replicated
        over xGMI; the batch calls shard the reads over the replicas."""
        hnd = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        check(lib.lrm_index_upload_opt(C.byref(hnd), C.byref(host.h.fmi), C.byref(host.h.lch), C.byref(host.h.sa),
                                       host.h.content, host.h.con_len, host.h.mta, host.h.mta_len, devs,
                                       len(devices), C.byref(capi.index_options(**opts))), "lrm_index_upload_opt")
        return cls(hnd)

    def set_map_options(self, **opts):
        """Default lrm_map_options of the batch calls on this handle (no arguments: the automatic choices)."""
        check(lib.lrm_index_set_map_options(self.handle, C.byref(capi.map_options(**opts))), "lrm_index_set_map_options")

    def tables(self):
        """The derived seed tables this handle ended up with (lrm_index_get_tables) as a dict."""
        t = capi.IndexTables()
        check(lib.lrm_index_get_tables(self.handle, C.byref(t)), "lrm_index_get_tables")
        return {f: int(getattr(t, f)) for f, _ in t._fields_ if not f.startswith("reserved")}

    def debug_set_vote_limits(self, t3_limit=0, t3_slots=0):
        check(lib.lrm_debug_set_vote_limits(self.handle, t3_limit, t3_slots), "lrm_debug_set_vote_limits")

    @property
    def replicas(self):
        return int(lib.lrm_index_replicas(self.handle))

    @classmethod
    def upload_blob(cls, blob: np.ndarray, device=0, **opts):
        hnd = C.c_void_p()
        check(lib.lrm_index_upload_blob_opt(C.byref(hnd), blob.ctypes.data, blob.nbytes, device,
                                            C.byref(capi.index_options(**opts))), "lrm_index_upload_blob_opt")
        return cls(hnd)

    @classmethod
    def adopt(cls, blob_tensor, device=0, **opts):
        """blob_tensor: torch uint8 CUDA tensor holding the image (e.g. after an RCCL broadcast)."""
        hnd = C.c_void_p()
        check(lib.lrm_index_adopt_device_opt(C.byref(hnd), blob_tensor.data_ptr(), blob_tensor.numel(), device,
                                             C.byref(capi.index_options(**opts))), "lrm_index_adopt_device_opt")
        return cls(hnd, keep=blob_tensor)

    def close(self):
        if self.handle:
            lib.lrm_index_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
