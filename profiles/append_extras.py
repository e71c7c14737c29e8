#!/usr/bin/env python3
"""Appends the sections of profiles/<tag>/README.md that do not come from profiles/collect.sh: the GRCh38-sized run
(`bench_grch38_100k_x_10kbp.json/.log`), the large-index test log and the pointer to the probe outputs.
Run after profiles/summarize_pmc.py (which rewrites the README from scratch)."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), tag)
d = json.loads(open(os.path.join(dst, "bench_grch38_100k_x_10kbp.json")).read().strip().splitlines()[-1])
ix, iso, pc, rf, cpu = d["index"], d["isolated"], d["pcie_inclusive"], d["roofline"], d["cpu_baseline"]
out = ["", "## GRCh38-sized text on one GPU (`bench_grch38_100k_x_10kbp.json`, `.log`)", "",
       "`python bench.py --ref-len 3099750718 --steps %d --warmup %d --cpu-seconds 8`: synthetic 3,099,750,718 bp reference "
       "(no FASTA on the box), L = %d rows; index built on the box's %d-CPU share in %.1f s (peak RSS %.1f GB), %.1f GiB "
       "image packed + uploaded in %.1f s, 64 GiB long seed table (pair-line 16-mers) derived on the device."
       % (d["steps"], d["warmup"], ix["rows"], ix["host_cpus"], ix["build_s"], ix["peak_rss_gb"], ix["image_bytes"] / 2**30,
          ix["pack_upload_s"]), "", "| | |", "|---|---|",
       "| `value` (HBM-resident, 3 streams) | **%.2f Gbp/s**, %.1f ms per 1-Gbp step |" % (d["value"], d["ms_per_step"]),
       "| serialized replay | %.2f Gbp/s, %.1f ms per step |" % (iso["value"], iso["ms_per_step"]),
       "| `pcie_inclusive` (`lrm_map_batch`, pinned / pageable) | %.2f / %.2f Gbp/s |" % (pc["pinned"]["value"], pc["pageable"]["value"]),
       "| `seed_search` | %.2f ms per launch (one launch per step), %.2f TB/s algorithmic = **%.2f of 8 TB/s** (%.0f B per read base in "
       "the reference layout) |" % (rf["avg_launch_ms"], rf["achieved"] / 1e3, rf["frac"], d["algorithmic_bytes_per_base"]["seed_search"]),
       "| CPU oracle, %d threads / 1 thread | %.5f / %.6f Gbp/s (x%.0f / x%.0f for `value`) |"
       % (cpu["cores"], cpu["value"], cpu["one_thread"]["value"], d["value"] / cpu["value"], d["value"] / cpu["one_thread"]["value"]),
       "| checked in the run | %s |" % pc["checked"], "", "Serialized per-kernel table of that run:", "",
       "| kernel | launches | avg ms |", "|---|---|---|"]
for k, v in iso["kernels"].items():
    out.append("| %s | %d | %.3f |" % (k, v["launches"], v["avg_ms"]))
for f, what in (("bench_ecoli_131072_reads.json", "E. coli-sized text, 131,072 x 10 kbp ONT reads per step"),
                ("bench_chr1_pacbio15k_131072_reads.json", "human-chr1-sized text, 131,072 x 15 kbp PacBio-CLR reads per step")):
    fp = os.path.join(dst, f)
    if os.path.exists(fp):
        e = json.loads(open(fp).read().strip().splitlines()[-1])
        out += ["", "`%s`: %s (two wavefronts of the lane-per-read extension on every SIMD): **%.2f Gbp/s** HBM-resident, %.2f serialized, "
                "%.2f / %.2f PCIe-inclusive (pinned / pageable), extension %.1f TCUPS."
                % (f, what, e["value"], e["isolated"]["value"], e["pcie_inclusive"]["pinned"]["value"], e["pcie_inclusive"]["pageable"]["value"],
                   e["isolated"]["roofline_gact"]["gcups"] / 1e3)]
out += ["", "`pmc_grch38_seed_vote.json`: FETCH_SIZE and TCC_HIT/MISS of `seed_search` and `vote` on the GRCh38-sized workload (own `--pmc` passes): "
        "seed_search with the pair-line 16-mer table fetches 90 GB per launch (1.44 G L2 misses x 64 B) in 29.7 ms; with the PLAIN tables that preceded it it fetches 95 GB per launch (1.52 G L2 misses x 64 B) in 32.2 ms with the 17-mer table, 128 GB (2.04 G misses) in 41.7 ms with the 16-mer table: 47-49 G random 64-byte lines per second in all three."]
out += ["", "`large_test.log`: `tests/test_gpu_large.py` (4.4 G rows: loci, rows and SA values beyond 2^32 end to end against the oracle).",
        "", "`probes/`: raw outputs of the tuning probes behind the \"measured and rejected\" notes of `DESIGN.md` (its README lists them)."]
with open(os.path.join(dst, "README.md"), "a") as f:
    f.write("\n".join(out) + "\n")
