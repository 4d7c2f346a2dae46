set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_probe.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_probe.log
for W in 4 5 6 7 0; do for r in 1 2 3; do
QC_TUNE_FIXED=$W timeout -k 10 300 python bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 30 2>>$O/fixed_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('fixed W=$W iter %.4f  build %.4f  linalg %.4f' % (d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
done; done
cd /tmp && export TMPDIR=/tmp
QC_SCF_DEBUG=1 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $O/pmc_probe -- python $R/bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 5 --warmup 3 > $O/pmc_probe.json 2> $O/pmc_probe.err; echo "pmc rc=$?"; grep "event join" $O/pmc_probe.err | head -2; cut -c1-200 $O/pmc_probe.json
