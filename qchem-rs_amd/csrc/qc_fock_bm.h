// qc_fock_bm.h - launch arguments of the bra-major kernels (qc_fock_bm.hip)
#pragma once
#include "qc_fock_kernel.h"

// One launch = the bra-major classes of one ket type (LCD = 0: ss kets, 1: ps kets) and bra range (HI = 0: LAB <= 2,
// 1: LAB >= 3); the grid is the concatenation of the classes' bundle lists ("segments"), one wave per bundle.
struct QcBmArgs {
    QcKernelArgs base;
    const double *pairdataT;
    int nseg;
    int seg_end[QC_MAXSEG];            // exclusive prefix of workgroup counts
    int seg_lab[QC_MAXSEG];
    const QcBundle *seg_bundles[QC_MAXSEG];
    const int *seg_ketlist[QC_MAXSEG];
};

int qc_launch_bm(int lcd, int hi, int grid, size_t lds, hipStream_t st, const QcBmArgs &a);
