# usage: bash tools/run_trace.sh <workload> <tag>
W=${1:-h2o_ccpvtz}; TAG=${2:-trace}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference --workload $W > $R/gpurun_out/${TAG}.json 2> $R/gpurun_out/${TAG}.err
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/${TAG}/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), r['TotalDurationNs'].rjust(12), ("%.1f"%float(r['AverageNs'])).rjust(12), r['Percentage'].rjust(7))
for r in rows:
    if 'gemm' in r['Name'] or 'jacobi' in r['Name'] or 'dots' in r['Name'] or 'energy' in r['Name'] or 'reduce_repl' in r['Name'] or 'symmetrize' in r['Name'] or 'lincomb' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), r['TotalDurationNs'].rjust(12), ("%.1f"%float(r['AverageNs'])).rjust(12), r['Percentage'].rjust(7))
PY
