"""End-to-end probe (GPU box): `accidx` + `accaln` on an E. coli-sized reference and N x 10 kbp reads from a
FASTQ file, SAM out -- wall time of the whole flow, i.e. with the text stages either side of the hot path.
python tools/accaln_probe.py [n_reads] [batch]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import capi, synth
from longreadmapper_amd.capi import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
Lr = 10000
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "accaln_probe_%d" % os.getpid())
os.makedirs(d, exist_ok=True)
ref = synth.reference(4641652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
fa, fq, sam = os.path.join(d, "ref.fa"), os.path.join(d, "reads.fq"), os.path.join(d, "out.sam")
with open(fa, "wb") as f:
    f.write(b">synth_ref\n" + bytes(ref) + b"\n")
t0 = time.perf_counter()
assert lib.lrm_accidx(fa.encode(), 32, 12, 1) == 0
t_idx = time.perf_counter() - t0
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
q = b"I" * Lr
with open(fq, "wb") as f:
    for i in range(n):
        f.write(b"@read%d\n" % i + r["reads"][i, :Lr].tobytes() + b"\n+\n" + q + b"\n")
total, valid = C.c_uint64(), C.c_uint64()
os.environ.setdefault("LRM_HOST_VERBOSE", "1")        # stage times of lrm_accaln on stderr
for rep in range(3):
    if os.path.exists(sam):
        os.remove(sam)                      # (truncating the previous run's gigabytes would be timed otherwise)
    t0 = time.perf_counter()
    capi.check(lib.lrm_accaln(fa.encode(), fq.encode(), sam.encode(), capi.Params(batch, 20, 300), capi.GactParams(0, 0, 0), 0,
                              1, C.byref(total), C.byref(valid)), "lrm_accaln")
    t = time.perf_counter() - t0
    print("accaln run %d: %d reads x %d bp in batches of %d: %.2f s = %.3f Gbp/s end to end (index load + upload included); "
          "valid %d / %d; FASTQ %.0f MB, SAM %.0f MB; accidx %.1f s"
          % (rep, n, Lr, batch, t, n * Lr / t / 1e9, valid.value, total.value, os.path.getsize(fq) / 1e6,
             os.path.getsize(sam) / 1e6, t_idx), flush=True)
for p in os.listdir(d):
    os.remove(os.path.join(d, p))
os.rmdir(d)
