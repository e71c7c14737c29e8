"""Host I/O probe (CPU only): throughput of the FASTQ batch loader and of the SAM formatter on 10 kbp reads.
python tools/io_probe.py [n_reads]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import capi, mapper, synth

lib = capi.lib
n, Lr = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 10000
ref = synth.reference(2_000_000, seed=1)
r = synth.reads([ref], n, Lr, synth.ONT, seed=3)
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "io_probe_%d.fq" % os.getpid())
with open(path, "wb") as f:
    q = b"I" * Lr
    for i in range(n):
        f.write(b"@read%d some comment\n" % i + r["reads"][i, :Lr].tobytes() + b"\n+\n" + q + b"\n")
size = os.path.getsize(path)
rd = C.c_void_p()
capi.check(lib.lrm_reader_open(C.byref(rd), path.encode()), "open")
b = capi.ReadBatch()
t0 = time.perf_counter()
got = lib.lrm_reader_next(rd, n, C.byref(b))
t1 = time.perf_counter()
assert got == n
print("reader: %d reads, %.1f MB of FASTQ in %.3f s = %.0f MB/s = %.3f Gbp/s" % (n, size / 1e6, t1 - t0, size / 1e6 / (t1 - t0), n * Lr / (t1 - t0) / 1e9))
mta = (capi.MtaEntry * 1)()
mta[0].name_len, mta[0].name, mta[0].offset, mta[0].seq_len = 4, b"chrA", 0, len(ref)
store = np.full((n, 2 * Lr), ord("="), dtype=np.uint8)
store[:, ::7] = ord("X"); store[:, ::31] = ord("I"); store[:, 5::43] = ord("D")
cig = (capi.Cigar * n)()
for i in range(n):
    cig[i].cigar = C.cast(store[i].ctypes.data, capi.u8p)
    cig[i].n_cigar_op = 11000
    cig[i].score = 980
score = np.full(n, 980, dtype=np.int32)
meta_r = np.ones(n, dtype=np.int32)
meta = np.zeros(n, dtype=mapper.META_DT)
meta["off"] = np.arange(n) * 13
ln = C.c_uint64()
for _ in range(2):                       # the second call has the thread team and the allocator warm
    t0 = time.perf_counter()
    t = lib.lrm_sam_format(C.byref(b), mta, 1, C.cast(cig, C.c_void_p), score.ctypes.data, meta.ctypes.data, meta_r.ctypes.data, n, C.byref(ln))
    t1 = time.perf_counter()
    if _ == 0:
        lib.lrm_free(t)
print("sam_format: %.1f MB of SAM in %.3f s = %.0f MB/s = %.3f Gbp/s (%d threads)" % (ln.value / 1e6, t1 - t0, ln.value / 1e6 / (t1 - t0), n * Lr / (t1 - t0) / 1e9, os.cpu_count()))
lib.lrm_free(t)
lib.lrm_read_batch_free(C.byref(b))
lib.lrm_reader_close(rd)
os.remove(path)
