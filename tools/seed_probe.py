"""Tuning probe (run on the GPU box): the seed stage of the default bench workload under a list of environment
variants, per-kernel HIP-event times.   python tools/seed_probe.py "A=1" "A=2 B=3" ...   ("" = defaults)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import capi, index, mapper, synth

n, Lr = int(os.environ.get("PROBE_READS", "100000")), int(os.environ.get("PROBE_LEN", "10000"))
ref = synth.reference(int(os.environ.get("PROBE_REF", "4641652")), seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
d_reads = torch.from_numpy(r["reads"]).cuda()
d_lens = torch.from_numpy(r["lens"].astype(np.int32)).cuda()
dm = mapper.DeviceMapper(di, n, Lr)
base = None
for var in (sys.argv[1:] or [""]):
    kv = dict(x.split("=") for x in var.split()) if var else {}
    os.environ.update(kv)
    capi.lib.lrm_debug_reload_env(di.handle)           # the overrides are normally read once per handle
    dm.seed(d_reads, d_lens)
    torch.cuda.synchronize()
    dm.set_timing(True)
    for _ in range(3):
        dm.seed(d_reads, d_lens)
    torch.cuda.synchronize()
    t = dm.timing()
    dm.set_timing(False)
    best = dm.results(n)["best"]
    if base is None:
        base = best.copy()
    same = bool(np.array_equal(best, base))
    for k in kv:
        del os.environ[k]
    print("%-40s seed_search %.2f  vote %.2f  ms per step (sum of both rounds)  same=%s"
          % (var or "(defaults)", t["seed_search_kernel"][0] / 3, t["vote_kernel"][0] / 3, same), flush=True)
