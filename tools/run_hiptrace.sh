# usage: bash tools/run_hiptrace.sh <workload> <tag>   (HIP API + kernel trace of a short bench run; on the GPU box)
W=${1:-h2o_ccpvtz}; TAG=${2:-hiptrace}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scaling-reference --workload $W > $R/gpurun_out/${TAG}.json 2> $R/gpurun_out/${TAG}.err
ls $R/gpurun_out/${TAG}/*/
