# What the Fock kernels' waves wait for: SQ instruction-class and wait counters per kernel (run on the GPU box through gpurun):
#   bash tools/wait_counters.sh [workload]     ->  gpurun_out/waitc/<pass>/..._counter_collection.csv ; python tools/wait_counters_summary.py
W=${1:-c6h6_ccpvdz}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/waitc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 10 --warmup 3"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p1 -- python $ARGS > /dev/null 2> $O/p1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --output-format csv -d $O/p2 -- python $ARGS > /dev/null 2> $O/p2.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/p3 -- python $ARGS > /dev/null 2> $O/p3.err || exit 1
echo done
