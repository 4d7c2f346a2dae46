#!/usr/bin/env python3
"""Kernel timeline of SCF passes from a rocprofv3 --kernel-trace CSV: for the chosen pass (default: the middle one) every kernel with
its start / end relative to the end of the previous pass's last kernel, and over all steady passes the mean start offset of every
Fock launch inside its build (how far the host's issue ramp pushes a launch behind the first one).
usage: build_timeline.py <kernel_trace.csv> [pass index]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)


def short(n):
    m = re.search(r'qc_fock_(tier|bm)_kernel<(\d+), (\d+)>', n)
    return "%s<%s,%s>" % m.groups() if m else n.split('(')[0][:40]


folds = [i for i, k in enumerate(ks) if 'qc_fold' in k[2]]
if len(folds) < 4:
    sys.exit("no fold kernels in the trace")
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(folds) // 2
i0, i1 = folds[it - 1], folds[it]
t0 = ks[i0][1]
print("pass %d: previous fold end -> this fold end %.1f us" % (it, (ks[i1][1] - t0) / 1e3))
for s, e, n, q in ks[i0 + 1:i1 + 1]:
    print("%8.1f %8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
# steady statistics over the last half of the passes
off, dur, span, gap_join, lin = defaultdict(list), defaultdict(list), [], [], []
for a, b in zip(folds[len(folds) // 2:-1], folds[len(folds) // 2 + 1:]):
    fk = [k for k in ks[a + 1:b] if 'qc_fock_' in k[2]]
    if not fk:
        continue
    s0 = min(k[0] for k in fk)
    e1 = max(k[1] for k in fk)
    span.append((e1 - s0) / 1e3)
    gap_join.append((ks[b][0] - e1) / 1e3)
    lin.append((s0 - ks[a][1]) / 1e3)
    for s, e, n, q in fk:
        off[short(n)].append((s - s0) / 1e3)
        dur[short(n)].append((e - s) / 1e3)
if span:
    print("steady passes: %d   Fock span %.1f us   last Fock kernel -> fold start %.1f us   previous fold end -> first Fock kernel %.1f us"
          % (len(span), sum(span) / len(span), sum(gap_join) / len(gap_join), sum(lin) / len(lin)))
    for n in sorted(off, key=lambda x: sum(off[x]) / len(off[x])):
        o, d = off[n], dur[n]
        print("   %-12s start +%6.1f us   duration %6.1f us   end +%6.1f us" % (n, sum(o) / len(o), sum(d) / len(d), (sum(o) + sum(d)) / len(o)))
