import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longreadmapper_amd import index, mapper, synth
ref = synth.reference(4_641_652, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
hi = index.HostIndex.build([ref], hlen=12)
di = index.DeviceIndex.upload(hi, 0)
n, Lr = 20000, 10000
r = synth.reads([ref], n, Lr, synth.ONT, seed=11)
dm = mapper.DeviceMapper(di, n, Lr)
d_reads = torch.from_numpy(r["reads"]).cuda(); d_lens = torch.from_numpy(r["lens"].astype(np.int32)).cuda()
for it in range(2):
    dm.seed(d_reads, d_lens); torch.cuda.synchronize()
print("done", dm.stats())
