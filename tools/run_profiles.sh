# Round profile set: tests, default bench, rocprofv3 kernel stats of the same command, PMC passes (run via gpurun)
set -e
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -- python $R/bench.py > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/bench_under_rocprof.err
cd $R && bash tools/run_pmc.sh h2o_ccpvtz pmc_h2o > /dev/null
echo done
