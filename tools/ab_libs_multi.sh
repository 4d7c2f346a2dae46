# A/B/C of library builds on ONE box: bash tools/ab_libs_multi.sh <workload> <rounds> <libA.so|default> <libB.so> ...
W=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for L in "$@"; do
      if [ $L = default ]; then unset QCHEM_HIP_LIB; else export QCHEM_HIP_LIB=$GRAFT_REPO_ROOT/qchem-rs_amd/$L; fi
      QC_BENCH_DETAIL=/tmp/ab_detail.json timeout -k 10 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 2>>$GRAFT_REPO_ROOT/gpurun_out/ab_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('%-28s %-12s iter %.4f  build %.4f  linalg %.4f' % ('$L', '$W', d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
  done
done
unset QCHEM_HIP_LIB
