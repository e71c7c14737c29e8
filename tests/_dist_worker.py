"""Worker for tests/test_dist_cpu.py: one rank of a world_size-2 gloo job (CPU only).
Exercises the multi-GPU plumbing of longreadmapper_amd.dist with the CPU oracle standing in
for the per-rank mapper (the GPU mapper itself is covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from longreadmapper_amd import dist, index, synth  # noqa: E402
import orc  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world, _ = dist.init_process_group(backend="gloo")
    assert world == 2
    ref = synth.reference(60_000, seed=5)
    r = synth.reads([ref], 40, 600, synth.ONT, seed=9)
    lens = r["lens"].copy()
    lens[::3] = 150                     # ragged, so that balancing by bases differs from by count
    reads = r["reads"].copy()
    for i, l in enumerate(lens):
        reads[i, l:] = 0

    # one-time index image broadcast: only rank 0 builds it
    blob = None
    hi = None
    if rank == 0:
        hi = index.HostIndex.build([ref], hlen=8)
        blob = torch.from_numpy(hi.pack_blob())
    dist.BCAST_CHUNK = 1 << 20          # several pieces even for this small image
    blob = dist.broadcast_blob(blob, device=torch.device("cpu"), src=0)
    digest = int(blob.to(torch.int64).sum().item()), int(blob.numel())

    # every rank maps its contiguous slice; results gathered in input order
    slices = dist.partition_by_bases(lens, world)
    lo, hi_ = slices[rank]
    oi = orc.OracleIndex.build([ref], hlen=8)     # stand-in mapper needs a host-side index on every rank
    best, _ = oi.seed_batch(reads[lo:hi_], lens[lo:hi_])
    rs = reads[lo:hi_].copy()
    ext = oi.extend_batch(rs, lens[lo:hi_], best)
    local = dict(key=best["key"].copy(), score=ext["score"], n_ops=ext["n_ops"])
    merged = dist.gather_in_order(local, slices)
    tdist.barrier()
    if rank == 0:
        full_best, _ = oi.seed_batch(reads, lens)
        rs = reads.copy()
        full = oi.extend_batch(rs, lens, full_best)
        ok = (np.array_equal(merged["key"], full_best["key"]) and np.array_equal(merged["score"], full["score"])
              and np.array_equal(merged["n_ops"], full["n_ops"]))
        with open(out_path, "w") as f:
            f.write("%d %d %d %s\n" % (digest[0], digest[1], int(ok), slices))
    else:
        with open(out_path + ".r1", "w") as f:
            f.write("%d %d\n" % digest)
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
