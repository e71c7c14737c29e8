"""Multi-GPU plumbing on CPU: world_size 2, gloo backend (the N>1 path of bench.py / dist.py)."""
import os
import subprocess
import sys

import numpy as np

from longreadmapper_amd import dist

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_by_bases_balances_and_covers():
    rng = np.random.default_rng(1)
    lens = rng.integers(100, 100_000, size=1000)
    for world in (1, 2, 3, 8):
        sl = dist.partition_by_bases(lens, world)
        assert sl[0][0] == 0 and sl[-1][1] == len(lens)
        assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
        per = [int(lens[a:b].sum()) for a, b in sl]
        assert max(per) - min(per) <= 2 * int(lens.max())
    assert dist.partition_by_bases([], 4) == [(0, 0)] * 4
    assert dist.partition_by_bases([5], 2)[-1][1] == 1


def test_two_ranks_gloo_broadcast_shard_merge(tmp_path):
    out = str(tmp_path / "res.txt")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29533", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    d0 = open(out).read().split()
    d1 = open(out + ".r1").read().split()
    assert d0[0] == d1[0] and d0[1] == d1[1]          # both ranks hold the same image bytes
    assert d0[2] == "1"                               # sharded + merged == single-process result
