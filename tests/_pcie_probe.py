"""Host-buffer (drop-in) entry points timed end to end at the C-ABI, PCIe included, with caller-owned buffers that
are reused across batches as alnmain.c reuses buf / store_mem: for DESIGN.md only."""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longreadmapper_amd import capi, index, mapper, synth
lib = capi.lib
ref = synth.reference(4_641_652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
n, Lr = int(os.environ.get("PROBE_READS", "100000")), 10000
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
lens = np.ascontiguousarray(r["lens"], dtype=np.uint32)
stride = r["reads"].shape[1]
reads = np.empty_like(r["reads"])
best = np.zeros(n, dtype=mapper.ENTRY_DT)
store_stride = 2 * Lr
store = np.ones((n, store_stride), dtype=np.uint8)          # touched: no first-use page faults inside the timing
cig = (capi.Cigar * n)()
score = np.zeros(n, dtype=np.int32); meta = np.zeros(n, dtype=mapper.META_DT); meta_r = np.zeros(n, dtype=np.int32)
p = capi.Params(n, 20, 300)
gp = capi.GactParams(320, 120, 128)
for it in range(4):
    reads[:] = r["reads"]
    t0 = time.perf_counter()
    capi.check(lib.lrm_seed_batch(di.handle, reads.ctypes.data, stride, lens.ctypes.data, n, p, best.ctypes.data), "seed")
    t1 = time.perf_counter()
    capi.check(lib.lrm_extend_batch(di.handle, reads.ctypes.data, stride, lens.ctypes.data, n, best.ctypes.data, gp,
                                    C.cast(cig, C.c_void_p), store.ctypes.data, store_stride, score.ctypes.data,
                                    meta.ctypes.data, meta_r.ctypes.data), "extend")
    t2 = time.perf_counter()
    print("host-buffer path: seed %.1f ms, extend %.1f ms, %.2f Gbp/s PCIe-inclusive (%d x %d)"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, n * Lr / (t2 - t0) / 1e9, n, Lr), flush=True)
