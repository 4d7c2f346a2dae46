set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
for T in 0.2 0.05; do
QC_EIG_WARM_RMS=$T QC_SCF_DEBUG=1 timeout -k 10 120 python tools/scf_trace_probe.py water cc-pVTZ 16 > $O/warm_$T.out 2> $O/warm_$T.err
echo "warm_rms $T:"; grep -o "ctl a: [0-9 -]*\|mode [0-9] cold [0-9]" $O/warm_$T.err | paste - - | head -8
for i in 1 2 3; do
QC_EIG_WARM_RMS=$T timeout -k 10 300 python bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('warm_rms $T iter %.4f  build %.4f  linalg %.4f  %s' % (d['ms_per_step'], b['fock_build'], b['diis_eig_density'], d['config']['passes'][-25:]))"
done; done
