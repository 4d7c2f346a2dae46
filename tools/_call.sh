set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_join.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_join.log
bash tools/ab_env.sh "QC_EVENT_JOIN=1" 5 > $O/ab_join.log 2>&1
cat $O/ab_join.log
for X in QC_AB_DUMMY=1 QC_EVENT_JOIN=1 QC_AB_DUMMY=1 QC_EVENT_JOIN=1; do
env $X timeout -k 10 300 python bench.py --workload c6h6_ccpvdz --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('c6h6 %-16s iter %.4f  build %.4f  linalg %.4f' % ('$X', d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
done
tail -5 $R/gpurun_out/ab_stderr.log
