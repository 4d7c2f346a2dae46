# Round profile set (run on the GPU box through gpurun): bash tools/run_profiles.sh <round tag, e.g. r03>
# Per workload: (1) the bench line itself, (2) rocprofv3 --kernel-trace --stats of the same command, (3) PMC passes in their own
# runs (separate --pmc passes, nothing but --kernel-trace beside them).  Under rocprofv3 the program goes directly after `--`.
# Then, in the SAME invocation: the summaries (tools/make_profile_summary.py -> profiles/<tag>_*) and the default `python bench.py` line,
# which quotes its HBM traffic from the PMC summary just written - so the two cannot disagree.  Everything lands under
# gpurun_out/prof_<tag>/ (profiles/ of the box copy is mirrored into gpurun_out/prof_<tag>/profiles/ for the way back).
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in h2o_ccpvtz c6h6_ccpvdz; do
  ARGS="$R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 --warmup 3"
  python $ARGS > $O/${W}_bench.json 2> $O/${W}_bench.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_trace -- python $ARGS > $O/${W}_bench_under_rocprof.json 2> $O/${W}_trace.err || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/${W}_sq -- python $ARGS > /dev/null 2> $O/${W}_sq.err || exit 1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${W}_mfma -- python $ARGS > /dev/null 2> $O/${W}_mfma.err || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${W}_fetch -- python $ARGS > /dev/null 2> $O/${W}_fetch.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $O/${W}_write -- python $ARGS > /dev/null 2> $O/${W}_write.err || exit 1
  echo "$W done"
done
cd $R
python tools/make_profile_summary.py $TAG $O > $O/summary.txt 2>&1 || { cat $O/summary.txt; exit 1; }
cat $O/summary.txt
# the same passes without a profiler in the process: every kernel's own clock (QC_DEV_TIMELINE), and the O2 triplet's linear algebra
WARM_RUNS=40 python tools/timeline_probe.py 2> $O/device_timeline_raw.txt || exit 1
{ echo "# tools/timeline_probe.py (QC_DEV_TIMELINE=1): H2O/cc-pVTZ RHF, one SCF run after 40 warm-up runs; microseconds on the device's 100 MHz clock,"; echo "# from the end of the previous pass's last kernel; unitN = launch unit N of the build (qc_unit_of), small2 = second launch of a cold pass"; grep "timeline\]\|events of the same" $O/device_timeline_raw.txt | tail -17; } > profiles/${TAG}_device_timeline.txt
python tools/open_shell_fused_probe.py > profiles/${TAG}_o2_triplet_linear_algebra.txt 2>&1 || exit 1
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
cp bench_detail.json $O/bench_default_detail.json
cp $O/bench_default.json profiles/${TAG}_bench_default.json; cp $O/bench_default_detail.json profiles/${TAG}_bench_default_detail.json
mkdir -p $O/profiles && cp profiles/${TAG}_* $O/profiles/
wc -c $O/bench_default.json
