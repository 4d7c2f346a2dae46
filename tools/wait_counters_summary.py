"""Per-kernel sums of the counters collected by tools/wait_counters.sh, normalised by SQ_WAVE_CYCLES where that makes sense."""
import csv, glob, collections, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "waitc")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "")[:34]
            if "qc_fock" in k:
                acc[k][p + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    wc = v.get("p1:SQ_WAVE_CYCLES", 1.0)
    print(k)
    for name, val in sorted(v.items()):
        print("    %-34s %12.4g   /wave_cycles(p1) %.3f" % (name, val, val / wc))
