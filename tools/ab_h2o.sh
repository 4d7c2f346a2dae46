A=$1; B=$2; R=${3:-6}
for r in $(seq 1 $R); do
  for L in $A $B; do
      QCHEM_HIP_LIB=$L timeout -k 10 300 python bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 30 2>>$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('%-28s iter %.4f  build %.4f  linalg %.4f' % ('$L'.split('/')[-1], d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
  done
done
