# usage: bash tools/run_trace.sh <workload> <tag>   (kernel trace + stats of a short bench run; on the GPU box)
W=${1:-h2o_ccpvtz}; TAG=${2:-trace}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference --workload $W > $R/gpurun_out/${TAG}.json 2> $R/gpurun_out/${TAG}.err
python $R/tools/build_timeline.py $(ls $R/gpurun_out/${TAG}/*/*kernel_trace.csv | head -1) 10
