set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
QC_TUNE_DEBUG=1 QCHEM_HIP_LIB=$R/qchem-rs_amd/libqchem_hip_base.so timeout -k 10 300 python bench.py --workload h2o_ccpvtz --no-extras --no-cpu-baseline --steps 30 2>$O/lot_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('run $i iter %.4f  build %.4f  linalg %.4f' % (d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
grep "final cand\|online\|in-pass" $O/lot_$i.err | tr '\n' ';' ; echo
done
for r in 1 2 3; do for L in libqchem_hip_base.so libqchem_hip.so; do
QCHEM_HIP_LIB=$R/qchem-rs_amd/$L timeout -k 10 300 python bench.py --workload c6h6_ccpvdz --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['iter_breakdown_ms']
print('c6h6 %-24s iter %.4f  build %.4f  linalg %.4f' % ('$L', d['ms_per_step'], b['fock_build'], b['diis_eig_density']))"
done; done
