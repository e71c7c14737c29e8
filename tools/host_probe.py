"""Tuning probe (run on the GPU box): the host-buffer boundary on caller buffers (default bench workload) under a list
of variants.  A variant is a blank-separated list of lrm_map_options fields and probe keys:
    python tools/host_probe.py "" "dense_results=1" "dense_results=1 inflight=2" "inflight=2 group_subs=6 buf=pageable"
probe keys: inflight=K (batches submitted ahead, default 1 = submit + wait), buf=pinned|pageable (default pinned),
steps=N.  Prints wall ms per batch, Gbp/s and the process CPU seconds per Gbp (all threads of the process)."""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longreadmapper_amd import capi, index, mapper, synth


def cpu_s():
    u = resource.getrusage(resource.RUSAGE_SELF)
    return u.ru_utime + u.ru_stime


n, Lr = int(os.environ.get("PROBE_READS", "100000")), int(os.environ.get("PROBE_LEN", "10000"))
ref = synth.reference(int(os.environ.get("PROBE_REF", "4641652")), seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
NBUF = 3
sstride = (2 * Lr + 15) // 16 * 16
bufs = {"pinned": [(mapper.pinned_empty((n, Lr + 1)), mapper.pinned_empty((n, sstride))) for _ in range(NBUF)]}
base = None
for var in (sys.argv[1:] or [""]):
    kv = dict(x.split("=") for x in var.split()) if var else {}
    inflight = int(kv.pop("inflight", 1))
    kind = kv.pop("buf", "pinned")
    steps = int(kv.pop("steps", 10))
    envs = {k: kv.pop(k) for k in list(kv) if k.startswith("LRM_")}          # tuning overrides, re-read for this variant
    os.environ.update(envs)
    capi.lib.lrm_debug_reload_env(di.handle)
    opts = {k: int(v) for k, v in kv.items()}
    if kind not in bufs:
        bufs[kind] = [(np.empty((n, Lr + 1), dtype=np.uint8), np.empty((n, sstride), dtype=np.uint8)) for _ in range(NBUF)]
    bb = bufs[kind]
    for hr, hs in bb:
        hr[:] = r["reads"]
    res = mapper.map_batch(di, bb[0][0], r["lens"], store=bb[0][1], options=opts)      # warm-up (mirrors, workspaces)
    p = [mapper.map_batch_submit(di, bb[k][0], r["lens"], store=bb[k][1], options=opts) for k in (1, 2, 0)]    # warm-up in flight
    [x.wait() for x in p]
    if base is None:
        base = (res["best"].copy(), res["score"].copy(), res["n_ops"].copy())
    same = np.array_equal(res["best"], base[0]) and np.array_equal(res["score"], base[1]) and np.array_equal(res["n_ops"], base[2])
    for hr, hs in bb:
        hr[:] = r["reads"]
    c0, t0 = cpu_s(), time.perf_counter()
    pend = []
    for s in range(steps):
        # (the reads of a buffer that has been through a batch are partly reverse-complemented; mapping them again is
        #  the same amount of work -- restoring 1 GB per step on the host would time the host, not the path)
        hr, hs = bb[s % NBUF]
        pend.append(mapper.map_batch_submit(di, hr, r["lens"], store=hs, options=opts))
        if len(pend) >= inflight:
            pend.pop(0).wait()
    while pend:
        pend.pop(0).wait()
    wall, cpu = time.perf_counter() - t0, cpu_s() - c0
    gbp = steps * n * Lr / 1e9
    for k in envs:
        del os.environ[k]
    print("%-52s %-8s inflight %d: %6.1f ms per batch  %5.2f Gbp/s  host CPU %.3f s per Gbp  same=%s"
          % (" ".join("%s=%s" % kv_ for kv_ in list(opts.items()) + list(envs.items())) or "(defaults)", kind, inflight, 1e3 * wall / steps, gbp / wall, cpu / gbp, same), flush=True)
