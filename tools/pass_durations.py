#!/usr/bin/env python3
"""Per-pass durations (fold end -> fold end) of the last N SCF passes in a rocprofv3 kernel trace, and the kernel list of the longest one.
usage: pass_durations.py <kernel_trace.csv> [N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
folds = [i for i, k in enumerate(ks) if 'qc_fold' in k[2]]
folds = folds[-(N + 1):]
d = [(ks[b][1] - ks[a][1]) / 1e3 for a, b in zip(folds[:-1], folds[1:])]
print("passes:", " ".join("%.0f" % x for x in d))
print("sum %.1f us, mean %.1f" % (sum(d), sum(d) / len(d)))
def short(n):
    m = re.search(r'qc_fock_(tier|bm)_kernel<(\d+), (\d+)>', n)
    return "%s<%s,%s>" % m.groups() if m else n.split('(')[0][:44]
w = max(range(len(d)), key=lambda i: d[i])
a, b = folds[w], folds[w + 1]
t0 = ks[a][1]
print("longest pass (%d):" % w)
for s, e, n, q in ks[a + 1:b + 1]:
    if 'join_mark' in n: continue
    print("%8.1f %8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
