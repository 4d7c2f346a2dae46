#!/usr/bin/env python3
"""Print the kernel timeline of one SCF iteration from a rocprofv3 --kernel-trace CSV (argument: path, iteration index)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Grid_Size_X', '')) for r in rows)
idx = [i for i, k in enumerate(ks) if 'reduce_replicas' in k[2]]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[it - 1], idx[it]
t0 = ks[i0][1]
print("window (us): %.1f" % ((ks[i1][0] - t0) / 1e3))
for s, e, n, q, g in ks[i0:i1 + 2]:
    m = re.search(r'qc_fock_tier_kernel<(\d+), (\d+)>', n)
    m2 = re.search(r'qc_fock_bm_kernel<(\d+), (\d+)>', n)
    nm = ("tier<%s,%s>" % m.groups()) if m else ("bm<%s,%s>" % m2.groups()) if m2 else n.split('(')[0][:34]
    if len(sys.argv) > 3 or m or m2 or 'jacobi' in n or 'reduce' in n:
        print("%8.1f %8.1f %7.1f  q%s %s grid=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, nm, g))
