#!/usr/bin/env python3
"""bench.py -- aligned Gbp/s of the seed-and-extend hot path on MI355X.

A step = one pass of the hot path (lrm_seed_batch_dev + lrm_extend_batch_dev) over one batch
of synthetic reads that already sits in HBM.  Default workload = BASELINE.json configs[1]:
E. coli K-12 sized reference (4,641,652 bp, synthetic -- no FASTA on the box), 100k x 10 kbp
ONT-profile reads, seed_len 20, thres 300 (reference defaults), GACT T=320 O=120 W=128.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, algorithmic bytes /
HIP-event time vs 8 TB/s), "cpu_baseline" (the CPU oracle, OpenMP on the host cores, bounded
sample of the same reads), "kernels" (per-kernel time and achieved algorithmic GB/s).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ECOLI_N = 4_641_652
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ref-len", type=int, default=ECOLI_N)
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--profile", default="ont", choices=["ont", "pacbio", "clean"])
    ap.add_argument("--seed-len", type=int, default=20)
    ap.add_argument("--thres", type=int, default=300)
    ap.add_argument("--gact", default="320,120,128")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0: skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as tdist
    from longreadmapper_amd import dist, index, mapper, synth

    rank, world, local = dist.init_process_group()
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d (launch N>1 with torch.distributed.run)" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gact = tuple(int(x) for x in args.gact.split(","))
    prof = {"ont": synth.ONT, "pacbio": synth.PACBIO_CLR, "clean": synth.CLEAN}[args.profile]

    # ---- index: built on the CPU by rank 0, one RCCL broadcast of the device image -----------------
    t0 = time.time()
    ref = synth.reference(args.ref_len, seed=1, repeat_frac=0.05, rep_len=300, rep_copies=1000, rep_div=0.05)
    hi = None
    blob = None
    if rank == 0:
        hi = index.HostIndex.build([ref], names=["synth_ecoli"], o_ratio=32, hlen=12)
        t_build = time.time() - t0
        blob = torch.from_numpy(hi.pack_blob()).to(dev)
        log("index: N=%d L=%d built in %.1fs, image %.1f MiB" % (args.ref_len, hi.length, t_build, blob.numel() / 2**20))
    t1 = time.time()
    blob = dist.broadcast_blob(blob, device=dev, src=0)
    torch.cuda.synchronize()
    t_bcast = time.time() - t1
    di = index.DeviceIndex.adopt(blob, local)

    # ---- reads: every rank maps its own batch (weak scaling), resident in HBM ----------------------
    n, Lr = args.reads, args.read_len
    r = synth.reads([ref], n, Lr, prof, seed=11 + 1000 * rank)
    pristine = torch.from_numpy(r["reads"]).to(dev)
    d_reads = torch.empty_like(pristine)
    d_lens = torch.from_numpy(r["lens"].astype(np.int32)).to(dev)
    dm = mapper.DeviceMapper(di, n, Lr, args.seed_len, args.thres, gact, device=local)
    bases = int(r["lens"].sum())
    if rank == 0:
        log("reads: %d x %d (%s), workspace %.2f GiB" % (n, Lr, args.profile, dm.workspace_bytes() / 2**30))

    def step():
        d_reads.copy_(pristine)      # extend rev-comps reverse-strand reads in place: restore the batch
        dm.seed(d_reads, d_lens)
        dm.extend(d_reads, d_lens)

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_kernel_timing:
        dm.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = dm.timing() if not args.no_kernel_timing else {}
    dm.set_timing(False)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        tdist.all_reduce(tt, op=tdist.ReduceOp.MAX)
    elapsed = float(tt.item())
    stats = dm.stats()

    if rank != 0:
        if world > 1:
            tdist.barrier()
        return

    # ---- sanity of the measured batch (cheap, outside the timed region) ----------------------------
    res = dm.results(min(n, 2000))
    mapped = int(((res["meta_r"] == 1) & (res["score"] >= 0)).sum())
    truth_ok = 0
    for i in range(len(res["score"])):
        if res["meta_r"][i] and abs(int(res["meta"]["off"][i]) - int(r["pos"][i])) < 300 \
                and int(res["meta"]["strand"][i]) == int(r["strand"][i]):
            truth_ok += 1
    log("sanity: %d/%d mapped, %d within 300 bp of the true locus, median ED %.0f"
        % (mapped, len(res["score"]), truth_ok, float(np.median(res["score"]))))

    # ---- CPU baseline: the oracle (CPU restatement), all host cores, bounded sample ----------------
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oi = orc.OracleIndex.from_host_index(hi)
    cores = min(orc.lib.orc_max_threads(), os.cpu_count() or 1)
    cpu = None
    per_base = None
    sample_n = min(n, 4 * cores)
    rs = np.ascontiguousarray(r["reads"][:sample_n]).copy()
    tc = time.perf_counter()
    best, _, ct_seed = oi.seed_batch(rs, r["lens"][:sample_n], args.seed_len, args.thres, nthreads=cores, counters=True)
    ext = oi.extend_batch(rs, r["lens"][:sample_n], best, gact, nthreads=cores, counters=True)
    probe = time.perf_counter() - tc
    if args.cpu_seconds > 0:
        want = int(min(n, max(sample_n, sample_n * args.cpu_seconds / max(probe, 1e-3))))
        if want > sample_n:
            sample_n = want
            rs = np.ascontiguousarray(r["reads"][:sample_n]).copy()
            tc = time.perf_counter()
            best, _, ct_seed = oi.seed_batch(rs, r["lens"][:sample_n], args.seed_len, args.thres, nthreads=cores,
                                             counters=True)
            ext = oi.extend_batch(rs, r["lens"][:sample_n], best, gact, nthreads=cores, counters=True)
            probe = time.perf_counter() - tc
    sample_bases = int(r["lens"][:sample_n].sum())
    cpu = dict(value=sample_bases / probe / 1e9, unit="Gbp/s", cores=cores, kind="port",
               sample="first %d reads of the batch (%d bases), seed+extend, %.1f s wall, OpenMP dynamic over reads"
                      % (sample_n, sample_bases, probe))
    # the oracle's output on the sample must equal the GPU's (same reads: the first sample_n of rank 0)
    gres = dm.results(min(sample_n, n))
    assert np.array_equal(gres["best"][:sample_n], best), "GPU best[] differs from the CPU oracle on the bench batch"
    assert np.array_equal(gres["score"][:sample_n], ext["score"]), "GPU scores differ from the CPU oracle"
    assert np.array_equal(gres["n_ops"][:sample_n], ext["n_ops"]), "GPU CIGAR lengths differ from the CPU oracle"
    # algorithmic bytes per read base, counted exactly on the sample
    ce = ext["counters"]
    per_base = dict(
        seed_search=(16 * ct_seed.n_lc + 8 * ct_seed.n_occ + ct_seed.bwt_bytes) / sample_bases,
        vote=(8 * ct_seed.n_sa) / sample_bases,
        pack2bit=1.0,
        gact=(2 * sample_bases + int(ce.cigar_ops) + 32 * sample_n) / sample_bases,
        cells=int(ce.cells) / sample_bases,
        seeds=ct_seed.n_seeds / sample_bases,
    )

    # ---- per-kernel table and the roofline of the dominant kernel ----------------------------------
    kernels = {}
    dominant = None
    for name, (ms, launches) in ktimes.items():
        if launches == 0:
            continue
        key = {"vote_wave": "vote", "vote_wave2": "vote", "vote_block": "vote"}.get(name.replace("_kernel", ""),
                                                                                 name.replace("_kernel", ""))
        alg = per_base.get(key) if not name.startswith("vote") else None
        per_launch_bytes = alg * bases * args.steps / launches if alg else None
        avg_ms = ms / launches
        kernels[name] = dict(ms_total=round(ms, 3), launches=launches, avg_ms=round(avg_ms, 4),
                             algorithmic_bytes_per_launch=per_launch_bytes,
                             achieved_GBps=(per_launch_bytes / (avg_ms * 1e-3) / 1e9) if per_launch_bytes else None)
        if dominant is None or ms > ktimes[dominant][0]:
            dominant = name
    roofline = None
    if dominant:
        k = kernels[dominant]
        ach = k["achieved_GBps"] or 0.0
        roofline = dict(kernel=dominant, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=ach / HBM_PEAK_GBS, traffic=None,
                        avg_launch_ms=k["avg_ms"], algorithmic_bytes_per_launch=k["algorithmic_bytes_per_launch"])
        if dominant == "gact_kernel":
            gcups = per_base["cells"] * bases * args.steps / (ktimes[dominant][0] * 1e-3) / 1e9
            roofline["note"] = ("integer DP kernel: HBM fraction is small by construction "
                                "(%.1f B/base algorithmic); %.0f GCUPS" % (per_base["gact"], gcups))
            roofline["gcups"] = gcups
    total_bases = bases * world * args.steps
    out = dict(metric="aligned Gbp/sec", value=total_bases / elapsed / 1e9, unit="Gbp/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8/u64 integer", data="synthetic",
               config=dict(workload="E. coli K-12 sized synthetic reference (%d bp, 5%% planted repeats), "
                                    "%d x %d bp %s-profile reads per GPU" % (args.ref_len, n, Lr, args.profile),
                           seed_len=args.seed_len, thres=args.thres, gact_T=gact[0], gact_O=gact[1], gact_W=gact[2],
                           reads_per_gpu=n, read_len=Lr, parallelism="reads sharded, index replicated (1 RCCL bcast)"),
               roofline=roofline, cpu_baseline=cpu, kernels=kernels,
               algorithmic_bytes_per_base=per_base, stats=stats,
               index_broadcast_s=round(t_bcast, 3), speedup_vs_cpu=(total_bases / elapsed / 1e9) / cpu["value"])
    print(json.dumps(out), flush=True)
    if world > 1:
        tdist.barrier()


if __name__ == "__main__":
    main()
