#!/bin/bash
# A second build of the library with some translation units compiled under extra flags, for A/B runs through QCHEM_HIP_LIB:
#   bash tools/build_variant_lib.sh <name> "<flags>" <tu> [<tu> ...]      e.g.  build_variant_lib.sh w2 "-DQC_T1LOW_WAVES(V)=2" gen/qc_fock_low1
# -> qchem-rs_amd/libqchem_hip_<name>.so (git-ignored, travels to the GPU box).  The product library is built first.
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../qchem-rs_amd/csrc"
make -s -j8
ALL="qc_system qc_api qc_fock qc_fock_bm qc_one_electron qc_linalg qc_eig_tridiag qc_peaks qc_scf_small gen/qc_fock_lab0 gen/qc_fock_lab1 gen/qc_fock_lab2 gen/qc_fock_lab3 gen/qc_fock_lab4 gen/qc_fock_lab5 gen/qc_fock_lab6 gen/qc_fock_low1 gen/qc_fock_mid1 gen/qc_fock_hi1"
objs=""
for tu in $ALL; do
  hit=0; for v in "$@"; do [ "$v" = "$tu" ] && hit=1; done
  if [ $hit = 1 ]; then
    src=$tu.hip; [ -f $src ] || src=$tu.cpp
    o=/tmp/qcv_${NAME}_$(basename $tu).o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -I../../include $FLAGS -c $src -o $o &
    objs="$objs $o"
  else objs="$objs $tu.o"; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libqchem_hip_$NAME.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
echo built ../libqchem_hip_$NAME.so
