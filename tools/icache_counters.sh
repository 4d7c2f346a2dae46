R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/icache
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in h2o_ccpvtz c6h6_ccpvdz; do
  ARGS="$R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 20 --warmup 3"
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/${W} -- python $ARGS > /dev/null 2> $O/${W}.err || exit 1
done
