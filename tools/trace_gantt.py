"""Text Gantt chart of a rocprofv3 --kernel-trace [--memory-copy-trace] CSV pair: busy fraction of every kernel family
and copy direction per time bin over the last WINDOW ms of the run.
    python tools/trace_gantt.py DIR_WITH_CSVS [window_ms=300] [bin_ms=5]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 300e6
bin_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 5e6
K = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
mf = glob.glob(d + "/*memory_copy_trace.csv")
M = list(csv.DictReader(open(mf[0]))) if mf else []
FAM = ('seed_search', 'vote', 'decide', 'pack2bit', 'locus', 'revcomp', 'bs_pack_reads', 'gact_bs', 'bs_expand', 'pack_rows',
       'copy_rows', 'lcl_build', 'fillBuffer', 'copyBuffer', 'bs_pack_content', 'gact3', 'gact_wide')


def short(n):
    for k in FAM:
        if k in n:
            return k
    return n[:20]


ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in K]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']),
        'H2D' if 'HOST_TO' in r['Direction'] else 'D2H' if 'DEVICE_TO_HOST' in r['Direction'] else 'D2D') for r in M]
tend = max(e[1] for e in ev)
t0 = tend - window
allv = [e for e in ev if e[1] > t0]
names = [n for n in ['H2D', 'pack2bit', 'seed_search', 'vote', 'decide', 'revcomp', 'bs_pack_reads', 'gact_bs', 'bs_expand', 'pack_rows',
                     'copy_rows', 'D2H', 'fillBuffer', 'copyBuffer'] if any(e[2] == n for e in allv)]
print("t(ms)  " + " ".join("%8s" % n[:8] for n in names))
for b in range(int(window / bin_ns)):
    lo, hi = t0 + b * bin_ns, t0 + (b + 1) * bin_ns
    row = []
    for n in names:
        iv = sorted((max(s, lo), min(e, hi)) for s, e, nm in allv if nm == n and e > lo and s < hi)
        busy, cur = 0, None
        for s, e in iv:
            if cur is None:
                cur = [s, e]
            elif s <= cur[1]:
                cur[1] = max(cur[1], e)
            else:
                busy += cur[1] - cur[0]
                cur = [s, e]
        if cur:
            busy += cur[1] - cur[0]
        row.append(busy / bin_ns)
    print("%5.0f  " % (b * bin_ns / 1e6) + " ".join("%8s" % ("%.2f" % x if x > 0.005 else ".") for x in row))
tot = collections.Counter()
for s, e, n in allv:
    tot[n] += (min(e, tend) - max(s, t0)) / 1e6
print("summed durations in the window (ms):", {k: round(v, 1) for k, v in tot.most_common()})
