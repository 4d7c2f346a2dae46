# The assignment search of N fresh processes on one box: where it ends (QC_TUNE_DEBUG) and the bench line of each.
# usage: bash tools/search_probe.sh <workload> <processes> ["VAR=value"]
W=$1; N=$2; X=${3:-QC_AB_DUMMY=1}
for r in $(seq 1 $N); do
  env $X QC_TUNE_DEBUG=1 QC_BENCH_DETAIL=/tmp/sp_detail.json timeout -k 10 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 2>/tmp/sp_err.txt >/tmp/sp_out.txt
  rc=$?
  if [ $rc -ne 0 ] || [ ! -s /tmp/sp_out.txt ]; then echo "bench.py exit code $rc, stdout $(wc -c < /tmp/sp_out.txt) bytes; stderr tail:"; grep -v "^\[tune\] trial" /tmp/sp_err.txt | tail -30; continue; fi
  python -c "
import json,sys
d=json.loads(open('/tmp/sp_out.txt').readline()); b=d['iter_breakdown_ms']
print('%-24s iter %.4f  build %.4f  linalg %.4f' % ('$X', d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
  grep "search ends\|finals inside" /tmp/sp_err.txt | cut -c1-220
done
