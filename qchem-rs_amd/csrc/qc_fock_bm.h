// qc_fock_bm.h - launch arguments of the bra-major kernels (qc_fock_bm.hip)
#pragma once
#include "qc_fock_kernel.h"

// One launch = the bra-major classes of one ket type (LCD = 0: ss kets, 1: ps kets) and bra range (HI = 0: LAB <= 2,
// 1: LAB >= 3); the grid is the concatenation of the classes' bundle lists ("segments"), one wave per bundle.
// waves per workgroup (they share the LDS Boys table, nothing else); the launcher uses fewer when the I blocks are large
// (Cartesian d bras).  8 waves for the launches with 13-14 KB I blocks was measured: no gain, the ss-ket ones lose.
#ifndef QC_BM_WAVES
#define QC_BM_WAVES 4
#endif
constexpr int qc_bm_waves(int lcd, int hi) { return QC_BM_WAVES; }
constexpr int QC_BM_LDS_TABLE = ((QC_BOYS_NGRID * 9 + 1) & ~1) * 8;   // bytes of the table at the head of the workgroup's LDS
struct QcBmArgs {
    QcKernelArgs base;
    const double *pairdataT;
    const double *pspack;              // packed primitive records of the ps pairs (ket side of the LCD = 1 kernels)
    int nseg;
    int seg_end[QC_MAXSEG];            // exclusive prefix of workgroup counts
    int seg_lab[QC_MAXSEG];
    int seg_lcd[QC_MAXSEG];            // ket type of the segment (the merged launch qc_fock_bm_kernel<3, 0>: ss kets / high bras + ps kets / low bras)
    int seg_nbundles[QC_MAXSEG];
    int seg_iwords[QC_MAXSEG];         // doubles of one wave's I block (nab * ncd * 65)
    int seg_rows[QC_MAXSEG];           // most bra functions (na + nb) of the segment's bundles: rows of a wave's exchange buffer (0: none)
    int use_rowbuf;                    // exchange rows accumulate in LDS (n small enough), else global atomics per bundle
    const QcBundleDev *seg_bundles[QC_MAXSEG];
    const QcKetUnit *seg_ketlist[QC_MAXSEG];
};

int qc_ds_order_probe(hipStream_t st, double *d_out64);      // 64 sums of the same 64-lane DS add: equal bits = the order is fixed
int qc_launch_bm(int lcd, int hi, int grid, int nwaves /* <= qc_bm_waves(lcd, hi) */, size_t lds, hipStream_t st, const QcBmArgs &a);
