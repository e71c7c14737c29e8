"""ctypes binding of the parity oracle (oracle/liblrm_oracle.so) and of the compiled pieces of
the reference (oracle/_ref/).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module; the product never does."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_LIB = os.path.join(ORC_DIR, "liblrm_oracle.so")
REF_HISTO = os.path.join(ORC_DIR, "_ref", "libref_histo.so")
REF_UI40 = os.path.join(ORC_DIR, "_ref", "libref_ui40.so")

ENTRY_DT = np.dtype([("key", "<u8"), ("val", "<u8"), ("bucket", "<u8")])
META_DT = np.dtype([("loc", "<u8"), ("off", "<u8"), ("seq_id", "<i4"), ("strand", "u1"), ("_pad", "V3")])
SEEDREC_DT = np.dtype([("j", "<i4"), ("_p", "V4"), ("rr", "<u8"), ("k", "<u8"), ("l", "<u8")])
u64p = C.POINTER(C.c_uint64)


class Entry(C.Structure):
    _fields_ = [("key", C.c_uint64), ("val", C.c_uint64), ("bucket", C.c_uint64)]


class Fmi(C.Structure):
    _fields_ = [("length", C.c_uint64), ("o_len", C.c_uint64), ("csa_len", C.c_uint64),
                ("c", u64p), ("o", u64p), ("csa", u64p), ("o_ratio", C.c_int), ("csa_ratio", C.c_int),
                ("bwt", C.c_void_p)]


class Lch(C.Structure):
    _fields_ = [("lc", u64p), ("len", C.c_uint64), ("hlen", C.c_int)]


class Mta(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("seq_len", C.c_uint64)]


class SeqMeta(C.Structure):
    _fields_ = [("loc", C.c_uint64), ("off", C.c_uint64), ("seq_id", C.c_int32), ("strand", C.c_uint8)]


class Index(C.Structure):
    _fields_ = [("fmi", Fmi), ("lch", Lch), ("sa", u64p), ("sa_len", C.c_uint64), ("content", C.c_void_p),
                ("con_len", C.c_uint64), ("mta", C.POINTER(Mta)), ("mta_len", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_lc", "n_fmi", "n_occ", "bwt_bytes", "n_sa", "n_seeds", "n_phases",
                                          "cells", "tiles", "read_bases", "cigar_ops")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def seed_algorithmic_bytes(self):
        """16 B per lc pair, 8 B per o-sample + the scanned bwt bytes per
        _occ_access, 8 B per SA entry, the read once, 24 B entry out."""
        return 16 * self.n_lc + 8 * self.n_occ + self.bwt_bytes + 8 * self.n_sa + self.read_bases


class GactParams(C.Structure):
    _fields_ = [("T", C.c_int), ("O", C.c_int), ("W", C.c_int)]


class SeedRec(C.Structure):
    _fields_ = [("j", C.c_int32), ("rr", C.c_uint64), ("k", C.c_uint64), ("l", C.c_uint64)]


class PhaseRec(C.Structure):
    _fields_ = [("iter", C.c_int32), ("top1", Entry), ("top2", Entry), ("v", C.c_uint64), ("decided", C.c_int32)]


class Trace(C.Structure):
    _fields_ = [("seeds", C.POINTER(SeedRec)), ("n_seeds", C.c_uint64), ("cap_seeds", C.c_uint64),
                ("phases", C.POINTER(PhaseRec)), ("n_phases", C.c_uint64), ("cap_phases", C.c_uint64)]


def build(force=False):
    """make -C oracle (also (re)builds oracle/_ref when /root/reference is present)."""
    src = [os.path.join(ORC_DIR, f) for f in ("lrm_oracle.c", "lrm_oracle.h", "Makefile")]
    stale = force or not os.path.exists(ORC_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORC_LIB) for s in src)
    need_ref = os.path.isdir("/root/reference/histo") and not os.path.exists(REF_HISTO)
    if stale or need_ref:
        r = subprocess.run(["make", "-C", ORC_DIR] + (["-B"] if force else []), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout)
    return ORC_LIB


def _load():
    build()
    lib = C.CDLL(ORC_LIB)
    vp, i, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    sig = {
        "orc_max_threads": (i, []),
        "orc_cat_build": (i, [C.POINTER(C.c_char_p), u64p, i, C.POINTER(vp), u64p, C.POINTER(Mta)]),
        "orc_sa_build": (i, [vp, u64, vp]),
        "orc_index_build": (i, [vp, u64, C.POINTER(Mta), i, i, i, C.POINTER(Index)]),
        "orc_index_free": (None, [C.POINTER(Index)]),
        "orc_index_adopt": (None, [C.POINTER(Index), vp, vp, u64, i, vp, u64, vp, u64, i, vp, u64, vp, u64,
                                   C.POINTER(Mta), i]),
        "orc_occ_access": (u64, [C.POINTER(Fmi), C.c_char, u64, vp]),
        "orc_fmi_aln": (u64, [C.POINTER(Fmi), vp, i, u64p, u64p, vp]),
        "orc_num_from_seq": (u64, [vp, i]),
        "orc_lc_aln": (u64, [vp, i, u64p, u64p, C.POINTER(Fmi), C.POINTER(Lch), vp]),
        "orc_sa_access": (u64, [C.POINTER(Index), u64]),
        "orc_csa_access": (u64, [C.POINTER(Fmi), u64]),
        "orc_histo_init": (vp, [u32]),
        "orc_histo_destroy": (None, [vp]),
        "orc_histo_add": (None, [vp, u64]),
        "orc_histo_find_2_max": (u64, [vp, C.POINTER(Entry)]),
        "orc_seed_read": (i, [C.POINTER(Index), vp, u32, u32, u32, C.POINTER(Entry), C.POINTER(Trace),
                              C.POINTER(Counters)]),
        "orc_seq_lookup": (i, [C.POINTER(Mta), i, u64, u32, C.POINTER(SeqMeta)]),
        "orc_rev_comp_in_place": (None, [vp, u32]),
        "orc_gact": (i, [vp, i, vp, i, GactParams, vp, C.POINTER(i), C.POINTER(Counters)]),
        "orc_extend_read": (i, [C.POINTER(Index), vp, u32, C.POINTER(Entry), GactParams, vp, C.POINTER(i),
                                C.POINTER(i), C.POINTER(SeqMeta), C.POINTER(Counters)]),
        "orc_parse_cigar": (i, [vp, i, vp, i]),
        "orc_seed_batch": (i, [C.POINTER(Index), vp, u64, vp, u64, u32, u32, vp, vp, C.POINTER(Counters), i]),
        "orc_extend_batch": (i, [C.POINTER(Index), vp, u64, vp, u64, vp, GactParams, vp, u64, vp, vp, vp, vp,
                                 C.POINTER(Counters), i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
_libc = C.CDLL(None)


def _view(ptr, n, dtype):
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    if n == 0 or not addr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class OracleIndex:
    """An orc_index, either built by the oracle itself or adopting external arrays."""

    def __init__(self):
        self.ix = Index()
        self._owned = False
        self._keep = []

    @classmethod
    def build(cls, seqs, o_ratio=32, hlen=12):
        self = cls()
        bufs = [bytes(memoryview(np.ascontiguousarray(s))) if isinstance(s, np.ndarray) else bytes(s) for s in seqs]
        n = len(bufs)
        seqp = (C.c_char_p * n)(*bufs)
        lens = (C.c_uint64 * n)(*[len(b) for b in bufs])
        mta = (Mta * n)()
        cat = C.c_void_p()
        cat_len = C.c_uint64()
        rc = lib.orc_cat_build(seqp, lens, n, C.byref(cat), C.byref(cat_len), mta)
        assert rc == 0, "orc_cat_build failed (non-ACGT base?)"
        lib.orc_index_build(cat, cat_len.value, mta, n, o_ratio, hlen, C.byref(self.ix))
        _libc.free(cat)
        self._owned = True
        return self

    @classmethod
    def build_from_text(cls, cat: bytes, mta_list, o_ratio=32, hlen=12):
        self = cls()
        n = len(mta_list)
        mta = (Mta * max(n, 1))()
        for i, (off, sl) in enumerate(mta_list):
            mta[i].offset, mta[i].seq_len = off, sl
        buf = C.create_string_buffer(cat, len(cat))
        lib.orc_index_build(buf, len(cat), mta, n, o_ratio, hlen, C.byref(self.ix))
        self._owned = True
        return self

    @classmethod
    def adopt(cls, c256, o, o_ratio, bwt, lc, hlen, sa, content, mta_list):
        """Borrow numpy arrays in the reference's in-memory layouts (kept alive by this object)."""
        self = cls()
        arrs = [np.ascontiguousarray(c256, dtype=np.uint64), np.ascontiguousarray(o, dtype=np.uint64),
                np.ascontiguousarray(bwt, dtype=np.uint8), np.ascontiguousarray(lc, dtype=np.uint64),
                np.ascontiguousarray(sa, dtype=np.uint64), np.ascontiguousarray(content, dtype=np.uint8)]
        n = len(mta_list)
        mta = (Mta * max(n, 1))()
        for i, (off, sl) in enumerate(mta_list):
            mta[i].offset, mta[i].seq_len = off, sl
        self._keep = arrs + [mta]
        c_, o_, b_, l_, s_, t_ = arrs
        lib.orc_index_adopt(C.byref(self.ix), c_.ctypes.data, o_.ctypes.data, len(o_), o_ratio, b_.ctypes.data,
                            len(b_), l_.ctypes.data, len(l_), hlen, s_.ctypes.data, len(s_), t_.ctypes.data,
                            len(t_), mta, n)
        return self

    @classmethod
    def from_host_index(cls, hi):
        """Adopt the arrays of a product-built HostIndex (index construction is outside the hot
        path; equality of the two builders is tested separately)."""
        mta = [(off, sl) for _, off, sl in hi.mta()]
        self = cls.adopt(hi.c(), hi.o(), int(hi.h.fmi.o_ratio), hi.bwt(), hi.lc(), hi.hlen, hi.sa(), hi.content(),
                         mta)
        self._keep.append(hi)
        return self

    def close(self):
        if self._owned:
            lib.orc_index_free(C.byref(self.ix))
            self._owned = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # views
    @property
    def length(self):
        return int(self.ix.fmi.length)

    def c(self):
        return _view(self.ix.fmi.c, 256, np.uint64)

    def o(self):
        return _view(self.ix.fmi.o, int(self.ix.fmi.o_len), np.uint64)

    def csa(self):
        return _view(self.ix.fmi.csa, int(self.ix.fmi.csa_len), np.uint64)

    def bwt(self):
        return _view(self.ix.fmi.bwt, self.length, np.uint8)

    def lc(self):
        return _view(self.ix.lch.lc, int(self.ix.lch.len), np.uint64)

    def sa(self):
        return _view(self.ix.sa, int(self.ix.sa_len), np.uint64)

    def content(self):
        return _view(self.ix.content, int(self.ix.con_len), np.uint8)

    def mta(self):
        return [(int(self.ix.mta[i].offset), int(self.ix.mta[i].seq_len)) for i in range(int(self.ix.mta_len))]

    # scalar entry points
    def fmi_aln(self, qry: bytes, k=None, l=None):
        kk = C.c_uint64(1 if k is None else k)
        ll = C.c_uint64(self.length - 1 if l is None else l)
        r = lib.orc_fmi_aln(C.byref(self.ix.fmi), qry, len(qry), C.byref(kk), C.byref(ll), None)
        return int(r), int(kk.value), int(ll.value)

    def lc_aln(self, qry: bytes):
        kk = C.c_uint64(1)
        ll = C.c_uint64(self.length - 1)
        r = lib.orc_lc_aln(qry, len(qry), C.byref(kk), C.byref(ll), C.byref(self.ix.fmi), C.byref(self.ix.lch), None)
        return int(r), int(kk.value), int(ll.value)

    def seed_read(self, read: bytes, seed_len=20, thres=300, trace=False):
        best = Entry()
        ct = Counters()
        tr = Trace() if trace else None
        phases = lib.orc_seed_read(C.byref(self.ix), read, len(read), seed_len, thres, C.byref(best),
                                   C.byref(tr) if trace else None, C.byref(ct))
        out = dict(best=(int(best.key), int(best.val), int(best.bucket)), phases=phases, counters=ct.as_dict())
        if trace:
            out["seeds"] = [(tr.seeds[i].j, int(tr.seeds[i].rr), int(tr.seeds[i].k), int(tr.seeds[i].l))
                            for i in range(tr.n_seeds)]
            out["phase_recs"] = [dict(iter=tr.phases[i].iter,
                                      top1=(int(tr.phases[i].top1.key), int(tr.phases[i].top1.val),
                                            int(tr.phases[i].top1.bucket)),
                                      top2=(int(tr.phases[i].top2.key), int(tr.phases[i].top2.val),
                                            int(tr.phases[i].top2.bucket)),
                                      v=int(tr.phases[i].v), decided=int(tr.phases[i].decided))
                                 for i in range(tr.n_phases)]
            _libc.free(C.cast(tr.seeds, C.c_void_p))
            _libc.free(C.cast(tr.phases, C.c_void_p))
        return out

    def seed_batch(self, reads, lens, seed_len=20, thres=300, nthreads=1, counters=False):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n, stride = reads.shape
        best = np.zeros(n, dtype=ENTRY_DT)
        phases = np.zeros(n, dtype=np.int32)
        ct = Counters()
        lib.orc_seed_batch(C.byref(self.ix), reads.ctypes.data, stride, lens.ctypes.data, n, seed_len, thres,
                           best.ctypes.data, phases.ctypes.data, C.byref(ct) if counters else None, nthreads)
        return (best, phases, ct) if counters else (best, phases)

    def extend_batch(self, reads, lens, best, gact=(320, 120, 128), nthreads=1, counters=False):
        """reads is modified in place (rev-comp), as in the reference."""
        assert reads.dtype == np.uint8 and reads.flags.c_contiguous
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        best = np.ascontiguousarray(best, dtype=ENTRY_DT)
        n, stride = reads.shape
        store_stride = max(2 * int(lens.max()) if n else 1, 1)
        store = np.zeros((n, store_stride), dtype=np.uint8)
        n_ops = np.zeros(n, dtype=np.int32)
        score = np.zeros(n, dtype=np.int32)
        meta = np.zeros(n, dtype=META_DT)
        meta_r = np.zeros(n, dtype=np.int32)
        ct = Counters()
        rc = lib.orc_extend_batch(C.byref(self.ix), reads.ctypes.data, stride, lens.ctypes.data, n,
                                  best.ctypes.data, GactParams(*gact), store.ctypes.data, store_stride,
                                  n_ops.ctypes.data, score.ctypes.data, meta.ctypes.data, meta_r.ctypes.data,
                                  C.byref(ct) if counters else None, nthreads)
        assert rc == 0
        out = dict(ops=store, n_ops=n_ops, score=score, meta=meta, meta_r=meta_r)
        if counters:
            out["counters"] = ct
        return out


def gact(q: bytes, d: bytes, T=320, O=120, W=128):
    ops = np.zeros(len(q) + len(d) + 1, dtype=np.uint8)
    n_ops = C.c_int(0)
    ct = Counters()
    score = lib.orc_gact(q, len(q), d, len(d), GactParams(T, O, W), ops.ctypes.data, C.byref(n_ops), C.byref(ct))
    return score, bytes(ops[:n_ops.value]), ct.as_dict()


def parse_cigar(ops: bytes):
    buf = C.create_string_buffer(4 * len(ops) + 16)
    n = lib.orc_parse_cigar(ops, len(ops), buf, len(buf))
    assert n >= 0
    return buf.value.decode()


def seq_lookup(mta_list, loc, qlen):
    n = len(mta_list)
    mta = (Mta * max(n, 1))()
    for i, (off, sl) in enumerate(mta_list):
        mta[i].offset, mta[i].seq_len = off, sl
    m = SeqMeta()
    r = lib.orc_seq_lookup(mta, n, loc, qlen, C.byref(m))
    return r, (int(m.loc), int(m.off), int(m.seq_id), int(m.strand))


def rev_comp(s: bytes):
    buf = C.create_string_buffer(s, len(s))
    lib.orc_rev_comp_in_place(buf, len(s))
    return buf.raw


class Histo:
    def __init__(self, cap=300):
        self.h = lib.orc_histo_init(cap)

    def add(self, key):
        lib.orc_histo_add(self.h, key)

    def find_2_max(self):
        st = (Entry * 2)()
        v = lib.orc_histo_find_2_max(self.h, st)
        return int(v), [(int(e.key), int(e.val), int(e.bucket)) for e in st]

    def __del__(self):
        try:
            lib.orc_histo_destroy(self.h)
        except Exception:
            pass


# ---- compiled pieces of the reference (only present after `make -C oracle` with /root/reference) ----
def ref_histo_lib():
    if not os.path.exists(REF_HISTO):
        return None
    l = C.CDLL(REF_HISTO)
    l.histo_init.restype = C.c_void_p
    l.histo_init.argtypes = [C.c_uint32]
    l.histo_add.restype = None
    l.histo_add.argtypes = [C.c_void_p, C.c_uint64]
    l.histo_destroy.restype = None
    l.histo_destroy.argtypes = [C.c_void_p]
    l.histo_find_2_max.restype = C.c_uint64
    l.histo_find_2_max.argtypes = [C.c_void_p, C.POINTER(Entry)]
    return l


def ref_ui40_lib():
    if not os.path.exists(REF_UI40):
        return None
    l = C.CDLL(REF_UI40)
    l.ref_ui40_sizeof.restype = C.c_ulong
    l.ref_ui40_from_bytes_convert.restype = C.c_uint64
    l.ref_ui40_from_bytes_convert.argtypes = [C.c_void_p]
    l.ref_ui40_fread_path.restype = C.c_size_t
    l.ref_ui40_fread_path.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    return l
