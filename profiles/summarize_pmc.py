#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by profiles/collect.sh) into the committed summaries:
profiles/<tag>/kernel_stats.csv (rocprofv3 --kernel-trace --stats), bench_default.json,
pmc_summary.json (per kernel, per timed step of the default workload) and README.md."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)

LAUNCHES_PER_STEP = {"pack2bit_kernel": 1, "seed_search_kernel": 2, "vote_kernel": 2, "decide_kernel": 2, "locus_resolve_kernel": 1, "revcomp_kernel": 1,
                     "gact3_kernel": 1, "gact_kernel": 1, "gact_bs_kernel": 1, "bs_pack_reads_kernel": 1,
                     "bs_expand_kernel": 1}


def short(name):
    n = name.replace("void ", "").split("(")[0].split("<")[0]
    if n in ("vote_fast_kernel", "vote_fast_block_kernel"):      # the vote stage is three launches per seeding round: fast
        n = "vote_kernel"                                       # (wavefront form), fast (workgroup form), exact (the list)
    return n


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))

_b = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
for _k, _v in (_b.get("isolated") or {}).get("kernels", {}).items():       # launches per step as the library really ran them
    if _k in LAUNCHES_PER_STEP and _b.get("steps"):
        LAUNCHES_PER_STEP[_k] = max(1, round(_v["launches"] / _b["steps"]))

LAUNCHES_PER_STEP["vote_kernel"] *= 3
per_kernel = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    order = collections.defaultdict(list)
    for r in rows:
        k = short(r["Kernel_Name"])
        d = int(r["Dispatch_Id"])
        per[(k, d)][r["Counter_Name"]] += float(r["Counter_Value"])
        per[(k, d)]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d not in order[k]:
            order[k].append(d)
    for k, ds in order.items():
        if k not in LAUNCHES_PER_STEP:
            continue
        sel = ds[-LAUNCHES_PER_STEP[k]:]          # the timed step is the last one (one warm-up step before it)
        agg = collections.defaultdict(float)
        for d in sel:
            for c, v in per[(k, d)].items():
                agg[c] += v
        per_kernel[k].update(agg)
        # (bench.py brackets the three vote launches of a seeding round with ONE pair of HIP events: one "launch" there)
        per_kernel[k]["launches_per_step"] = LAUNCHES_PER_STEP[k] // 3 if k == "vote_kernel" else LAUNCHES_PER_STEP[k]

bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
summary = {"workload": bench["config"], "note": "per timed step (1 Gbp); FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them",
           "kernels": {}}
lines = ["| kernel | launches/step | ms/step (PMC run) | FETCH GB | WRITE GB | L2 hit | VALU wave-instr | SQ_WAIT_ANY / SQ_WAVE_CYCLES |",
         "|---|---|---|---|---|---|---|---|"]
for k, v in per_kernel.items():
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    e = dict(launches_per_step=int(v["launches_per_step"]), ms=v.get("ms"), fetch_bytes=v.get("FETCH_SIZE", 0) * 1024,
             write_bytes=v.get("WRITE_SIZE", 0) * 1024, l2_hit=hit, l2_miss=miss, valu_insts=v.get("SQ_INSTS_VALU"),
             salu_insts=v.get("SQ_INSTS_SALU"), vmem_rd=v.get("SQ_INSTS_VMEM_RD"), wave_cycles_q=v.get("SQ_WAVE_CYCLES"),
             wait_any_q=v.get("SQ_WAIT_ANY"), active_valu_q=v.get("SQ_ACTIVE_INST_VALU"), waves=v.get("SQ_WAVES"),
             lds_insts=v.get("SQ_INSTS_LDS"), lds_active_q=v.get("SQ_ACTIVE_INST_LDS"), lds_bank_conflict=v.get("SQ_LDS_BANK_CONFLICT"),
             lds_idx_active=v.get("SQ_LDS_IDX_ACTIVE"), lds_addr_conflict=v.get("SQ_LDS_ADDR_CONFLICT"),
             wait_inst_any_q=v.get("SQ_WAIT_INST_ANY"), active_inst_any_q=v.get("SQ_ACTIVE_INST_ANY"), busy_cycles=v.get("SQ_BUSY_CYCLES"))
    summary["kernels"][k] = e
    lines.append("| %s | %d | %.2f | %.2f | %.2f | %s | %s | %s |" % (
        k, e["launches_per_step"], e["ms"] or 0, e["fetch_bytes"] / 1e9, e["write_bytes"] / 1e9,
        ("%.0f %%" % (100 * hit / (hit + miss))) if hit + miss else "-",
        ("%.3g" % e["valu_insts"]) if e["valu_insts"] else "-",
        ("%.0f %%" % (100 * e["wait_any_q"] / e["wave_cycles_q"])) if e["wave_cycles_q"] else "-"))
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)

def stats_table(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append("| %s | %s | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
    return "| kernel | calls | avg ms | % |\n|---|---|---|---|\n" + "\n".join(rows) + "\n"


def event_table(b):
    out = "| kernel | launches | avg ms |\n|---|---|---|\n"
    for k, v in b["kernels"].items():
        out += "| %s | %d | %.3f |\n" % (k, v["launches"], v["avg_ms"])
    return out


def load(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    shutil.copy(p, os.path.join(dst, name))
    return json.loads(open(p).read().strip().splitlines()[-1])


b1 = load("bench_streams1.json")
bc = load("bench_chr1_pacbio15k.json")
bu = load("bench_ultralong_20k_x_100kbp.json")
stats1 = glob.glob(os.path.join(src, "trace_streams1", "*", "*_kernel_stats.csv"))
if stats1:
    shutil.copy(stats1[0], os.path.join(dst, "kernel_stats_streams1.csv"))
with open(os.path.join(dst, "README.md"), "w") as f:
    f.write("# Profiles %s -- default `bench.py` workload on one MI355X\n\n" % tag)
    f.write("`bench_default.json`: the JSON line of `python bench.py` (%d steps, %d warm-up, steps alternating over %d HIP "
            "streams so that kernels of different steps overlap).\n\n" % (bench["steps"], bench["warmup"], bench.get("streams", 1)))
    f.write("value = **%.2f Gbp/s**, %.1f ms per 1-Gbp step; CPU oracle on %d host cores: %.4f Gbp/s (x%.0f).\n\n"
            % (bench["value"], bench["ms_per_step"], bench["cpu_baseline"]["cores"], bench["cpu_baseline"]["value"],
               bench["speedup_vs_cpu"]))
    if b1:
        f.write("`bench_streams1.json`: `python bench.py --streams 1` (every kernel alone on the chip): **%.2f Gbp/s**, "
                "%.1f ms per step.\n\n" % (b1["value"], b1["ms_per_step"]))
    if bc:
        f.write("`bench_chr1_pacbio15k.json`: human-chr1-sized text, 50 k x 15 kbp PacBio-CLR-profile reads: **%.2f Gbp/s**, "
                "%.1f ms per 0.75-Gbp step.\n\n" % (bc["value"], bc["ms_per_step"]))
    if bu:
        f.write("`bench_ultralong_20k_x_100kbp.json`: 20 k x 100 kbp ONT-profile reads: **%.2f Gbp/s**, %.1f ms per 2-Gbp step.\n\n"
                % (bu["value"], bu["ms_per_step"]))
    if bench.get("pcie_inclusive"):
        pi = bench["pcie_inclusive"]
        f.write("PCIe-inclusive (SURVEY 8(d): caller buffers, H2D of reads + D2H of results timed):\n\n| leg | Gbp/s | ms per batch | host CPU s per Gbp |\n|---|---|---|---|\n")
        for k, v in pi.items():
            if isinstance(v, dict) and "ms_per_batch" in v:
                f.write("| %s | %.2f | %.1f | %.3f |\n" % (k, v["value"], v["ms_per_batch"], v["host_cpu_s_per_Gbp"]))
        f.write("\n")
    g = bench.get("grch38")
    if g and g.get("value"):
        f.write("GRCh38-sized leg of the same command (fresh child process, %.0f s): **%.2f Gbp/s** HBM-resident (%.1f ms per step), "
                "%.2f Gbp/s PCIe-inclusive, serialized replay %.1f ms per step (seed_search %.2f, vote %.2f, gact_bs %.2f), x%.0f the CPU oracle "
                "on %d threads.\n\n" % (g["child_wall_s"], g["value"], g["ms_per_step"], g["value_pcie_inclusive"], g["isolated"]["ms_per_step"],
                                        g["isolated"]["kernel_avg_ms"]["seed_search_kernel"], g["isolated"]["kernel_avg_ms"]["vote_kernel"],
                                        g["isolated"]["kernel_avg_ms"]["gact_bs_kernel"], g["speedup_vs_cpu"], g["cpu_baseline"]["cores"]))
    f.write("## HIP-event timing inside bench.py, default command (durations include the overlap with other steps' kernels)\n\n")
    f.write(event_table(bench))
    if bench.get("isolated"):
        f.write("\n## Serialized replay inside the same command (`isolated`: what `roofline` is taken from)\n\n")
        f.write(event_table(bench["isolated"]))
    if stats:
        f.write("\n## rocprofv3 --kernel-trace --stats of the default command (`kernel_stats.csv`)\n\n" + stats_table(stats[0]))
    if b1:
        f.write("\n## HIP-event timing, `--streams 1` (isolated kernels)\n\n" + event_table(b1))
    if stats1:
        f.write("\n## rocprofv3 --kernel-trace --stats of `bench.py --streams 1` (`kernel_stats_streams1.csv`)\n\n" + stats_table(stats1[0]))
    f.write("\n## PMC passes (`pmc_summary.json`; separate `--streams 1` runs, one timed step each)\n\n" + "\n".join(lines) + "\n")
print(open(os.path.join(dst, "README.md")).read())
