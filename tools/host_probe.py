"""Tuning probe (run on the GPU box): lrm_map_batch on caller buffers (default bench workload) under a list of
environment variants; pinned and pageable buffers.   python tools/host_probe.py "A=1" "B=2 C=3" ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import index, mapper, synth

n, Lr = int(os.environ.get("PROBE_READS", "100000")), int(os.environ.get("PROBE_LEN", "10000"))
ref = synth.reference(4641652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
bufs = {"pinned": (mapper.pinned_empty((n, Lr + 1)), mapper.pinned_empty((n, 2 * Lr))),
        "pageable": (np.empty((n, Lr + 1), dtype=np.uint8), np.empty((n, 2 * Lr), dtype=np.uint8))}
for hr, hs in bufs.values():
    hs[:] = 0
for var in (sys.argv[1:] or [""]):
    kv = dict(x.split("=") for x in var.split()) if var else {}
    os.environ.update(kv)
    for kind, (hr, hs) in bufs.items():
        hr[:] = r["reads"]
        mapper.map_batch(di, hr, r["lens"], store=hs)
        ts = []
        for _ in range(4):
            hr[:] = r["reads"]
            t0 = time.perf_counter()
            mapper.map_batch(di, hr, r["lens"], store=hs)
            ts.append(time.perf_counter() - t0)
        print("%-44s %-9s %.1f ms  (%.2f Gbp/s)  min %.1f" % (var or "(defaults)", kind, 1e3 * np.mean(ts),
              n * Lr / np.mean(ts) / 1e9, 1e3 * min(ts)), flush=True)
    for k in kv:
        del os.environ[k]
