"""Seeded synthetic workloads; thin binding of liblrm_synth.so."""
import ctypes as C
import os

import numpy as np

from . import _build

ONT = dict(p_sub=0.04, p_ins=0.03, p_del=0.03)          # ~10 % error
PACBIO_CLR = dict(p_sub=0.015, p_ins=0.09, p_del=0.045)  # ~15 % error
CLEAN = dict(p_sub=0.0, p_ins=0.0, p_del=0.0)


def _lib():
    path = _build.SYNTH_LIB
    if not os.path.exists(path):
        path = _build.build_synth()
    lib = C.CDLL(path)
    lib.lrm_synth_reference.restype = C.c_int
    lib.lrm_synth_reference.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_uint32, C.c_uint32,
                                        C.c_double]
    lib.lrm_synth_reads.restype = C.c_int
    lib.lrm_synth_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32,
                                    C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return lib


_L = None


def lib():
    global _L
    if _L is None:
        _L = _lib()
    return _L


def reference(n, seed=1, repeat_frac=0.0, rep_len=300, rep_copies=1000, rep_div=0.05):
    """i.i.d. upper-case ACGT, optionally with planted repeat families -> uint8 array."""
    out = np.empty(n, dtype=np.uint8)
    rc = lib().lrm_synth_reference(out.ctypes.data, n, seed, repeat_frac, rep_len, rep_copies, rep_div)
    assert rc == 0
    return out


def reads(seqs, n_reads, read_len, profile=ONT, seed=11, nthreads=0):
    """Reads from a list of forward sequences (uint8 arrays).

    Returns dict(reads=(n, read_len+1) uint8 NUL padded, lens, seq, pos, span, strand)."""
    if isinstance(seqs, np.ndarray):
        seqs = [seqs]
    ref = np.concatenate(seqs) if len(seqs) > 1 else np.ascontiguousarray(seqs[0])
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    stride = read_len + 1
    out = np.zeros((n_reads, stride), dtype=np.uint8)
    rl = np.zeros(n_reads, dtype=np.uint32)
    tseq = np.zeros(n_reads, dtype=np.int32)
    tpos = np.zeros(n_reads, dtype=np.uint64)
    tspan = np.zeros(n_reads, dtype=np.uint32)
    tstrand = np.zeros(n_reads, dtype=np.uint8)
    rc = lib().lrm_synth_reads(ref.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(seqs), n_reads, read_len,
                               profile["p_sub"], profile["p_ins"], profile["p_del"], seed, out.ctypes.data,
                               stride, rl.ctypes.data, tseq.ctypes.data, tpos.ctypes.data, tspan.ctypes.data,
                               tstrand.ctypes.data, nthreads)
    assert rc == 0
    return dict(reads=out, lens=rl, seq=tseq, pos=tpos, span=tspan, strand=tstrand)
