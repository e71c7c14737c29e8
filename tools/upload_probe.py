import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longreadmapper_amd import index, synth
ref = synth.reference(4641652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
torch.cuda.init(); torch.zeros(1).cuda()
for var in ("0", "13", "14", "15", "16", "16", None):
    if var is None: os.environ.pop("LRM_LC_LONG", None)
    else: os.environ["LRM_LC_LONG"] = var
    t0 = time.perf_counter(); di = index.DeviceIndex.upload(hi, 0); torch.cuda.synchronize(); t1 = time.perf_counter()
    di.close(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("LRM_LC_LONG=%s upload %.3f s, free %.3f s" % (var, t1 - t0, t2 - t1), flush=True)
