"""Host-buffer (drop-in) entry points timed end to end, PCIe included: for DESIGN.md only."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longreadmapper_amd import index, mapper, synth
ref = synth.reference(4_641_652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
n, Lr = 20000, 10000
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
for it in range(3):
    reads = r["reads"].copy()
    t0 = time.perf_counter(); best = mapper.seed_batch(di, reads, r["lens"]); t1 = time.perf_counter()
    ext = mapper.extend_batch(di, reads, r["lens"], best); t2 = time.perf_counter()
    print("host-buffer path: seed %.1f ms, extend %.1f ms, %.2f Gbp/s PCIe-inclusive (%d x %d)" % ((t1-t0)*1e3, (t2-t1)*1e3, n*Lr/(t2-t0)/1e9, n, Lr), flush=True)
