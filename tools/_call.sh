set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_sw.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sw.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
