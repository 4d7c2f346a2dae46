set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_pp2.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pp2.log
bash tools/ab_env.sh "QC_NO_BM_PP=1" 3 > $O/ab_pp_h2o.log 2>&1; cat $O/ab_pp_h2o.log
