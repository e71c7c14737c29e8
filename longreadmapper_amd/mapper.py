"""Batch entry points of the hot path, host-buffer and device-buffer flavours.

`seed_batch` / `extend_batch` are PART 1 / PART 2 of single_end() (alnmain.c:333-451) for a
whole batch; names, argument meaning and data conventions follow the reference (best[],
cig[], limit[], meta_r[], m[])."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib

ENTRY_DT = np.dtype([("key", "<u8"), ("val", "<u8"), ("bucket", "<u8")])
META_DT = np.dtype([("loc", "<u8"), ("off", "<u8"), ("seq_id", "<i4"), ("strand", "u1"), ("_pad", "V3")])
assert ENTRY_DT.itemsize == 24 and META_DT.itemsize == 24

DEFAULT_SEED_LEN = 20      # alnmain.c:577-580
DEFAULT_THRES = 300
DEFAULT_GACT = (320, 120, 128)
N_KERNELS = 9              # LRM_N_KERNELS in include/lrm_accel.h


def seed_batch(index, reads, lens, seed_len=DEFAULT_SEED_LEN, thres=DEFAULT_THRES):
    """Host buffers in, best[] out (numpy structured array key/val/bucket)."""
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n, stride = reads.shape
    best = np.zeros(n, dtype=ENTRY_DT)
    p = capi.Params(n, seed_len, thres)
    check(lib.lrm_seed_batch(index.handle, reads.ctypes.data, stride, lens.ctypes.data, n, p, best.ctypes.data),
          "lrm_seed_batch")
    return best


def extend_batch(index, reads, lens, best, gact=DEFAULT_GACT):
    """Host buffers; `reads` is modified in place (reverse-strand reads are rev-comped).

    Returns dict(ops=(n, store_stride) uint8, n_ops, score, meta, meta_r)."""
    assert reads.dtype == np.uint8 and reads.flags.c_contiguous and reads.flags.writeable
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    best = np.ascontiguousarray(best, dtype=ENTRY_DT)
    n, stride = reads.shape
    max_len = int(lens.max()) if n else 0
    store_stride = max(2 * max_len, 1)                    # alnmain.c:316-320
    store = np.zeros((n, store_stride), dtype=np.uint8)
    cig = (capi.Cigar * max(n, 1))()
    score = np.zeros(n, dtype=np.int32)
    meta = np.zeros(n, dtype=META_DT)
    meta_r = np.zeros(n, dtype=np.int32)
    gp = capi.GactParams(*gact)
    check(lib.lrm_extend_batch(index.handle, reads.ctypes.data, stride, lens.ctypes.data, n, best.ctypes.data, gp,
                               C.cast(cig, C.c_void_p), store.ctypes.data, store_stride, score.ctypes.data,
                               meta.ctypes.data, meta_r.ctypes.data), "lrm_extend_batch")
    n_ops = np.ctypeslib.as_array(C.cast(cig, C.POINTER(C.c_int32)), shape=(max(n, 1), 4))[:n, 2].copy()
    return dict(ops=store, n_ops=n_ops, score=score, meta=meta, meta_r=meta_r)


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in pinned host memory (lrm_host_alloc): the DMA engines read / write it directly.
    Free with pinned_free(arr)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.lrm_host_alloc(max(n, 1))
    if not p:
        raise capi.LrmError("lrm_host_alloc: " + lib.lrm_last_error().decode(errors="replace"))
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    return arr


def pinned_free(arr):
    lib.lrm_host_free(arr.ctypes.data)


class PendingBatch:
    """A batch submitted with map_batch_submit: wait() blocks until its results are in the caller's arrays."""

    def __init__(self, ticket, keep, n, dense, text=False):
        self.ticket, self._keep, self.n, self.dense, self.text = ticket, keep, n, dense, text

    def wait(self):
        t, self.ticket = self.ticket, None
        assert t is not None, "already waited for"
        check(lib.lrm_map_batch_wait(t), "lrm_map_batch_wait")
        best, store, cig, score, meta, meta_r = self._keep[:6]
        n = self.n
        cv = np.ctypeslib.as_array(C.cast(cig, C.POINTER(C.c_int32)), shape=(max(n, 1), 4))[:n]
        out = dict(best=best, ops=store, n_ops=cv[:, 2].copy(), score=score, meta=meta, meta_r=meta_r)
        if self.dense:          # cig[i].cigar = store_mem + off[i]
            ptr = np.ctypeslib.as_array(C.cast(cig, C.POINTER(C.c_uint64)), shape=(max(n, 1), 2))[:n, 0]
            out["ops_off"] = (ptr - np.uint64(store.ctypes.data)).astype(np.int64)
        out["is_text"] = self.text   # cig[i].cigar -> NUL-terminated run-length CIGAR text: text_of(res, i)
        return out


def text_of(res, i):
    """Run-length CIGAR text of read i from a map_batch result in the cigar_text layout."""
    assert res.get("is_text")
    flat = res["ops"].reshape(-1)
    o = int(res["ops_off"][i])
    chunk = 64
    while True:
        b = bytes(flat[o:o + chunk])
        z = b.find(b"\0")
        if z >= 0:
            return b[:z]
        chunk *= 4


def ops_of(res, i):
    """Op bytes of read i from a map_batch result in either layout."""
    k = int(res["n_ops"][i])
    if "ops_off" in res:
        o = int(res["ops_off"][i])
        return bytes(res["ops"].reshape(-1)[o:o + k])
    return bytes(res["ops"][i, :k])


def map_batch_submit(index, reads, lens, seed_len=DEFAULT_SEED_LEN, thres=DEFAULT_THRES, gact=DEFAULT_GACT, store=None,
                     options=None):
    """lrm_map_batch_submit: queues the batch and returns a PendingBatch.  `reads` is modified in place like
    extend_batch once the batch runs; `store` may be a caller-provided (n, >= 2*max_len) uint8 array (e.g. pinned);
    `options`: dict of lrm_map_options fields (None: the handle's defaults)."""
    assert reads.dtype == np.uint8 and reads.flags.c_contiguous and reads.flags.writeable
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n, stride = reads.shape
    max_len = int(lens.max()) if n else 0
    if store is None:
        store = np.zeros((n, max((2 * max_len + 15) // 16 * 16, 16)), dtype=np.uint8)
    store_stride = store.shape[1]
    best = np.zeros(n, dtype=ENTRY_DT)
    cig = (capi.Cigar * max(n, 1))()
    score = np.zeros(n, dtype=np.int32)
    meta = np.zeros(n, dtype=META_DT)
    meta_r = np.zeros(n, dtype=np.int32)
    opt = capi.map_options(**options) if options is not None else None
    ticket = C.c_void_p()
    check(lib.lrm_map_batch_submit(index.handle, reads.ctypes.data, stride, lens.ctypes.data, n,
                                   capi.Params(n, seed_len, thres), capi.GactParams(*gact), best.ctypes.data,
                                   C.cast(cig, C.c_void_p), store.ctypes.data, store_stride, score.ctypes.data,
                                   meta.ctypes.data, meta_r.ctypes.data, C.byref(opt) if opt is not None else None,
                                   C.byref(ticket)), "lrm_map_batch_submit")
    text = bool(opt.cigar_text) if opt is not None else False
    dense = (bool(opt.dense_results) or text) if opt is not None else False
    return PendingBatch(ticket, (best, store, cig, score, meta, meta_r, reads, lens), n, dense, text)


def map_batch(index, reads, lens, seed_len=DEFAULT_SEED_LEN, thres=DEFAULT_THRES, gact=DEFAULT_GACT, store=None,
              options=None):
    """PART 1 + PART 2 in one device pass; `reads` is modified in place like extend_batch.
    Without `options` this is lrm_map_batch (the handle's default options), with them submit + wait."""
    if options is not None:
        return map_batch_submit(index, reads, lens, seed_len, thres, gact, store, options).wait()
    assert reads.dtype == np.uint8 and reads.flags.c_contiguous and reads.flags.writeable
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n, stride = reads.shape
    max_len = int(lens.max()) if n else 0
    if store is None:
        store = np.zeros((n, max(2 * max_len, 1)), dtype=np.uint8)
    store_stride = store.shape[1]
    best = np.zeros(n, dtype=ENTRY_DT)
    cig = (capi.Cigar * max(n, 1))()
    score = np.zeros(n, dtype=np.int32)
    meta = np.zeros(n, dtype=META_DT)
    meta_r = np.zeros(n, dtype=np.int32)
    check(lib.lrm_map_batch(index.handle, reads.ctypes.data, stride, lens.ctypes.data, n,
                            capi.Params(n, seed_len, thres), capi.GactParams(*gact), best.ctypes.data,
                            C.cast(cig, C.c_void_p), store.ctypes.data, store_stride, score.ctypes.data,
                            meta.ctypes.data, meta_r.ctypes.data), "lrm_map_batch")
    n_ops = np.ctypeslib.as_array(C.cast(cig, C.POINTER(C.c_int32)), shape=(max(n, 1), 4))[:n, 2].copy()
    return dict(best=best, ops=store, n_ops=n_ops, score=score, meta=meta, meta_r=meta_r)


def result_flags(score, meta_r, meta):
    n = len(score)
    flag = np.zeros(n, dtype=np.int32)
    mapq = np.zeros(n, dtype=np.int32)
    valid = np.zeros(n, dtype=np.int32)
    score = np.ascontiguousarray(score, dtype=np.int32)
    meta_r = np.ascontiguousarray(meta_r, dtype=np.int32)
    meta = np.ascontiguousarray(meta, dtype=META_DT)
    lib.lrm_result_flags(score.ctypes.data, meta_r.ctypes.data, meta.ctypes.data, n, flag.ctypes.data,
                         mapq.ctypes.data, valid.ctypes.data)
    return flag, mapq, valid


class DeviceMapper:
    """Device-resident batches: torch tensors own the HBM buffers, kernels run on torch's
    current stream (so torch.cuda.Event brackets them)."""

    def __init__(self, index, n_max, max_len, seed_len=DEFAULT_SEED_LEN, thres=DEFAULT_THRES, gact=DEFAULT_GACT,
                 device=0):
        import torch
        self.torch = torch
        self.index = index
        self.n_max, self.max_len = n_max, max_len
        self.seed_len, self.thres, self.gact = seed_len, thres, gact
        self.dev = torch.device("cuda", device)
        ws = C.c_void_p()
        check(lib.lrm_workspace_create(C.byref(ws), index.handle, n_max, max_len, seed_len, thres),
              "lrm_workspace_create")
        self.ws = ws
        self.store_stride = 2 * max_len
        self.best = torch.zeros((n_max, 3), dtype=torch.int64, device=self.dev)       # lrm_entry
        self.store = torch.zeros((n_max, self.store_stride), dtype=torch.uint8, device=self.dev)
        self.n_ops = torch.zeros(n_max, dtype=torch.int32, device=self.dev)
        self.score = torch.zeros(n_max, dtype=torch.int32, device=self.dev)
        self.meta = torch.zeros((n_max, 24), dtype=torch.uint8, device=self.dev)      # lrm_seq_meta
        self.meta_r = torch.zeros(n_max, dtype=torch.int32, device=self.dev)

    def workspace_bytes(self):
        return int(lib.lrm_workspace_bytes(self.ws))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def seed(self, d_reads, d_lens, n=None):
        n = d_reads.shape[0] if n is None else n
        p = capi.Params(n, self.seed_len, self.thres)
        check(lib.lrm_seed_batch_dev(self.index.handle, self.ws, d_reads.data_ptr(), d_reads.stride(0),
                                     d_lens.data_ptr(), n, self.max_len, p, self.best.data_ptr(), self._stream()),
              "lrm_seed_batch_dev")

    def extend(self, d_reads, d_lens, n=None):
        n = d_reads.shape[0] if n is None else n
        gp = capi.GactParams(*self.gact)
        check(lib.lrm_extend_batch_dev(self.index.handle, self.ws, d_reads.data_ptr(), d_reads.stride(0),
                                       d_lens.data_ptr(), n, self.max_len, self.best.data_ptr(), gp,
                                       self.store.data_ptr(), self.store_stride, self.n_ops.data_ptr(),
                                       self.score.data_ptr(), self.meta.data_ptr(), self.meta_r.data_ptr(),
                                       self._stream()), "lrm_extend_batch_dev")

    def stats(self):
        st = capi.Stats()
        check(lib.lrm_workspace_stats(self.ws, C.byref(st), self._stream()), "lrm_workspace_stats")
        return dict(vote_tier2_items=int(st.vote_tier2_items), vote_tier3_items=int(st.vote_tier3_items),
                    reads_decided_phase0=int(st.reads_decided_phase0), gact_tiles=int(st.gact_tiles),
                    seeds_evaluated=int(st.seeds_evaluated), seed_table_lookups=int(st.seed_table_lookups),
                    seed_rank_requests=int(st.seed_rank_requests), vote_redo_items=int(st.vote_redo_items))

    def set_counting(self, enable=True):
        """The next seed calls run the counting build of the seed kernel (stats(): requests of the device layout)."""
        check(lib.lrm_workspace_set_counting(self.ws, int(enable)), "lrm_workspace_set_counting")

    def set_timing(self, enable=True):
        check(lib.lrm_workspace_set_timing(self.ws, int(enable)), "lrm_workspace_set_timing")

    def timing(self):
        """-> {kernel name: (total ms, launches)} accumulated since set_timing / the last call."""
        ms = np.zeros(N_KERNELS, dtype=np.float64)
        launches = np.zeros(N_KERNELS, dtype=np.uint64)
        check(lib.lrm_workspace_timing(self.ws, ms.ctypes.data, launches.ctypes.data, self._stream()),
              "lrm_workspace_timing")
        return {lib.lrm_kernel_name(i).decode(): (float(ms[i]), int(launches[i])) for i in range(N_KERNELS)}

    def results(self, n):
        """Copy the outputs of the last seed+extend to numpy (host)."""
        best = self.best[:n].cpu().numpy().view(np.uint64).reshape(n, 3)
        out = np.zeros(n, dtype=ENTRY_DT)
        out["key"], out["val"], out["bucket"] = best[:, 0], best[:, 1], best[:, 2]
        meta = self.meta[:n].cpu().numpy().reshape(-1).view(META_DT)
        return dict(best=out, ops=self.store[:n].cpu().numpy(), n_ops=self.n_ops[:n].cpu().numpy(),
                    score=self.score[:n].cpu().numpy(), meta=meta, meta_r=self.meta_r[:n].cpu().numpy())

    def close(self):
        if self.ws:
            lib.lrm_workspace_free(self.ws)
            self.ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
