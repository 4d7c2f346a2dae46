# A/B of two builds of the library on ONE box (box-to-box spread is 5-8 %, larger than most kernel changes): alternates
#   bash tools/ab_libs.sh <libA.so> <libB.so> [rounds]     (run on the GPU box through gpurun; prints per-round bench breakdowns)
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    for W in h2o_ccpvtz c6h6_ccpvdz; do
      QCHEM_HIP_LIB=$L timeout -k 10 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 2>>$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('%-28s %-12s iter %.4f  build %.4f  linalg %.4f' % ('$L'.split('/')[-1], '$W', d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
    done
  done
done
