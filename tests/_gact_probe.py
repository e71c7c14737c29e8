import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longreadmapper_amd import index, mapper, synth
ref = synth.reference(4_641_652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
n, Lr = 40000, 10000
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
dm = mapper.DeviceMapper(di, n, Lr)
pr = torch.from_numpy(r["reads"]).cuda(); d_reads = pr.clone(); d_lens = torch.from_numpy(r["lens"].astype(np.int32)).cuda()
dm.seed(d_reads, d_lens); torch.cuda.synchronize()
for it in range(3):
    d_reads.copy_(pr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dm.extend(d_reads, d_lens); e1.record(); torch.cuda.synchronize()
    print(os.environ.get("LRM_GACT_IMPL"), os.environ.get("LRM_GACT_DBG"), "extend ms", e0.elapsed_time(e1), flush=True)
