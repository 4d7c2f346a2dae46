# A/B/C... of environment settings on ONE box: bash tools/ab_multi.sh <workload> <rounds> "VAR=a" "VAR=b" ...   (QC_AB_DUMMY=1 = default build)
W=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for X in "$@"; do
      env $X QC_BENCH_DETAIL=/tmp/ab_detail.json timeout -k 10 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 2>>$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
dd=json.load(open('/tmp/ab_detail.json')); sp=dd.get('config',{}).get('spec', '')
print('%-28s iter %.4f  build %.4f  linalg %.4f  %s' % ('$X', d['ms_per_step'], b['fock_build'], b['diis_eig_density'], sp))"
  done
done
