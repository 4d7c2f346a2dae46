# usage: bash tools/run_pmc.sh <workload> <tag>   (run on the GPU box through gpurun)
W=${1:-c6h6_ccpvdz}; TAG=${2:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-reference --workload $W > $R/gpurun_out/${TAG}_sq.json 2> $R/gpurun_out/${TAG}_sq.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-reference --workload $W > /dev/null 2> $R/gpurun_out/${TAG}_sq2.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-reference --workload $W > /dev/null 2> $R/gpurun_out/${TAG}_fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $R/gpurun_out/${TAG}_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-reference --workload $W > /dev/null 2> $R/gpurun_out/${TAG}_write.err || exit 1
ls $R/gpurun_out/${TAG}_*/*/ | head -30
