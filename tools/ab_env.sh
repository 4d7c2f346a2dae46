# A/B of one environment setting on ONE box, H2O/cc-pVTZ: bash tools/ab_env.sh "VAR=value" [rounds]
E=$1; R=${2:-6}
for r in $(seq 1 $R); do
  for X in "QC_AB_DUMMY=1" "$E"; do
      env $X timeout -k 10 300 python bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 30 2>>$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
dd=json.load(open('$GRAFT_REPO_ROOT/bench_detail.json')); sp=dd.get('config',{}).get('spec', '')
print('%-28s iter %.4f  build %.4f  linalg %.4f  %s' % ('$X', d['ms_per_step'], b['fock_build'], b['diis_eig_density'], sp))"
  done
done
