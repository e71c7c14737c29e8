"""Tuning probe (run on the GPU box): how well do the seed stage (memory-latency bound) and the extension
(VALU bound) of DIFFERENT batches share the chip?  Times N seed passes alone, N extension passes alone, and both
loops at once on two streams, for the stream-priority combinations.   python tools/overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import index, mapper, synth

n, Lr, N = int(os.environ.get("PROBE_READS", "100000")), int(os.environ.get("PROBE_LEN", "10000")), 6
ref = synth.reference(4641652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
pristine = torch.from_numpy(r["reads"]).cuda()
d_lens = torch.from_numpy(r["lens"].astype(np.int32)).cuda()
dm1, dm2 = mapper.DeviceMapper(di, n, Lr), mapper.DeviceMapper(di, n, Lr)
r1, r2 = pristine.clone(), pristine.clone()
dm1.seed(r1, d_lens)
dm2.seed(r2, d_lens)
dm2.extend(r2, d_lens)
torch.cuda.synchronize()


def seeds(s):
    with torch.cuda.stream(s):
        for _ in range(N):
            dm1.seed(r1, d_lens)


def exts(s):
    with torch.cuda.stream(s):
        for _ in range(N):
            r2.copy_(pristine)
            dm2.extend(r2, d_lens)


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N * 1e3


for ps, pe in ((0, 0), (0, -1), (-1, 0)):
    s1, s2 = torch.cuda.Stream(priority=ps), torch.cuda.Stream(priority=pe)
    a = wall(lambda: seeds(s1))
    b = wall(lambda: exts(s2))
    c = wall(lambda: (seeds(s1), exts(s2)))
    d = wall(lambda: (exts(s2), seeds(s1)))
    print("prio seed %2d ext %2d: seed alone %.2f ms  ext alone %.2f ms  sum %.2f  both %.2f / %.2f ms per step"
          % (ps, pe, a, b, a + b, c, d), flush=True)
