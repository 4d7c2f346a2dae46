set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_seq.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_seq.log
bash tools/ab_env.sh "QC_EVENT_WAIT=1" 6 > $O/ab_seq.log 2>&1
cat $O/ab_seq.log
grep -v amdgpu.ids $R/gpurun_out/ab_stderr.log | tail -5
