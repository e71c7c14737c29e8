"""Probe (GPU box): reads made of an interspersed repeat family of 100-299 copies -- every seed has that many hits, no
seed is unique, so no item can be settled by the fast vote kernels and every item has tens of thousands of hits.  Times
the seed stage with the one-pass global vote table of such items and with the LDS passes they took before
(lrm_debug_set_vote_limits(768, 0) keeps the default pass limit but switches the global table off).
    python tools/repeat_probe.py [copies=200] [n_reads=2000] [read_len=4000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from longreadmapper_amd import index, mapper, synth
import orc

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
Lr = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
elem = synth.reference(Lr + 500, seed=5)
parts = []
for i in range(copies):
    parts.append(synth.reference(700 + (i * 37) % 90, seed=1000 + i))
    parts.append(elem)
parts.append(synth.reference(700, seed=3))
ref = np.concatenate(parts)
hi = index.HostIndex.build([ref], hlen=10)
di = index.DeviceIndex.upload(hi, 0)
rng = np.random.default_rng(1)
reads = np.zeros((n, Lr + 1), dtype=np.uint8)
for i in range(n):
    o = int(rng.integers(0, 500))
    reads[i, :Lr] = elem[o:o + Lr]
lens = np.full(n, Lr, dtype=np.uint32)
d_reads = torch.from_numpy(reads).cuda()
d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
dm = mapper.DeviceMapper(di, n, Lr)
oi = orc.OracleIndex.from_host_index(hi)
want, _ = oi.seed_batch(reads[:8], lens[:8], nthreads=8)
for what, lim in (("one-pass global table", 0), ("LDS passes (round 2)", 768)):
    di.debug_set_vote_limits(lim, 0)
    dm.seed(d_reads, d_lens, n=min(n, 64))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dm.seed(d_reads, d_lens)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    got = dm.results(8)["best"]
    st = dm.stats()
    print("%-24s %d reads x %d bp, every seed %d hits (%d hits per item): seed stage %.1f ms = %.3f ms per read, %d items left to the exact "
          "kernel, first 8 reads equal the oracle: %s" % (what, n, Lr, copies, (Lr // 21) * copies, t * 1e3, t * 1e3 / n, st["vote_redo_items"],
                                                          bool(np.array_equal(got, want))), flush=True)
