#!/bin/bash
# Builds qchem-rs_amd/libqchem_hip_dbg.so: the product library with the column kernels of the high bra classes compiled under
# -DQC_PHASE_TIMING (qc_fock_kernel.h): workgroups 0 and 1 of every matrix-core segment print where a slot's time goes (setup, Boys,
# R table, k-loop, step 3, digestion; 10 ns units).  Diagnostic only:
#   QCHEM_HIP_LIB=$PWD/qchem-rs_amd/libqchem_hip_dbg.so python tools/class_profile.py h2o_ccpvtz 1
set -e
cd "$(dirname "$0")/../qchem-rs_amd/csrc"
make -s -j6
objs=""
for l in 0 1 2; do objs="$objs gen/qc_fock_lab$l.o"; done
for l in 3 4 5 6; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -DQC_PHASE_TIMING -c gen/qc_fock_lab$l.hip -o /tmp/qc_dbg_lab$l.o &
  objs="$objs /tmp/qc_dbg_lab$l.o"
done
# bra-major kernels under -DQC_BM_TIMING: wave 0 of the first two workgroups of a segment prints set-up / k-loop / step 3 / digestion / flush
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -DQC_BM_TIMING -c qc_fock_bm.hip -o /tmp/qc_dbg_bm.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libqchem_hip_dbg.so qc_system.o qc_api.o qc_fock.o /tmp/qc_dbg_bm.o qc_one_electron.o qc_linalg.o qc_eig_tridiag.o qc_peaks.o qc_scf_small.o $objs -ldl -Wl,-rpath,/opt/rocm/lib
