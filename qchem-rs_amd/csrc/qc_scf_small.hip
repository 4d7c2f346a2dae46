// qc_scf_small.hip - the Roothaan step of one spin for n <= 64 basis functions as ONE workgroup with every matrix in LDS.
//
// Replaces, for small molecules, the ~25-40 launches per SCF pass of the generic path (qc_api.cpp roothaan_enqueue + qc_linalg.hip):
// the body of the reference's loop between the Fock build and the convergence test (rhf.rs:70-88, uhf.rs:84-137), i.e.
//   pre    e = F D S - S D F (rhf.rs:71), DIIS over the sample window (diis.rs:28-59), F' = X^T F X (rhf.rs:74)
//   refine sorted_eigs(F') (rhf.rs:75, utils.rs:20-36) by Ogita-Aishima refinement from start vectors (previous pass / tridiagonal path)
//   post   C = X C' (rhf.rs:76), new density (rhf.rs:169-181), electronic energy and diagonal rms (rhf.rs:84-88)
// At n = 58 each of those launches is a 4-5 us latency kernel (16 tiles of 16 x 16, operands from L2) and the pass is bound by launch
// gaps.  Here four waves - one per SIMD of one CU, 512 registers each - own a 2 x 2 set of 16 x 16 tiles each.  A matrix is a 64 x 66
// block of LDS (34 KB; four of them in the CU's 160 KB) whose padding rows and columns are zeroed once and never written again, so a
// product is 16 k-steps of four independent f64 MFMAs per wave fed by four ds_read_b64 with no bounds test anywhere (leading dimension
// 66 = 2 mod 4 doubles: the row-strided fragments hit distinct bank pairs), all operands requested before the first matrix
// instruction; phases are separated by s_barrier instead of kernel boundaries; whatever streams in from L2 (Fock / error matrices of the
// DIIS window, one-electron matrices) is requested sixteen elements per lane at a time.
// (Round 2's single-workgroup fusion lost because its tiles read their operands through the CU's L1 from global memory, DESIGN.md 3.3;
// a first LDS form with 16 waves of one tile each was no faster than the launches it replaced: 128 registers per lane serialise the
// loads; bounds tests as branches put a wait behind every load.)  All reductions have a fixed order: results are deterministic.
#include <atomic>

#include "qc_internal.h"

typedef double qcs_d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int QCS_T = 256, QCS_W = QCS_T / 64, QCS_E = 16;           // threads, waves, matrix elements per thread (rows rr + 4 q)
constexpr int QCS_LD = 66, QCS_BUF = 64 * QCS_LD;                    // leading dimension and doubles of a matrix buffer
constexpr int QCS_SMALL_DOUBLES = 64 /*lam*/ + QCS_W * 12 /*red*/ + 12 /*dots*/ + 12 /*c*/ + 144 /*B*/ + 8 /*scalars*/ + 12 * 16 /*wave sums of the generic-order dots*/;
constexpr int QCS_SMALL_INTS = 64 /*partner*/ + 64 /*rank*/ + 16 /*flags*/;

__device__ __forceinline__ double qcs_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int qcs_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// C = alpha op(A) op(B) on the zero-padded 64 x 66 blocks.  QUAD (n > 32): wave w owns the tiles (w >> 1) + 2 a, (w & 1) + 2 b, a, b in
// {0, 1} - the partial edge tiles of a 58 x 58 matrix are spread evenly - and runs all 16 k-steps; otherwise (n <= 32) one tile per wave, 8
// k-steps.  `ks` limits the k-steps (density: only the occupied columns, KMASK).  Padding operands are zeros, padding results are not stored.
// Lane maps of v_mfma_f64_16x16x4 as in qc_gemm_tile (qc_linalg.hip); the k-steps of a tile run in ascending order, so an element is the
// same chain of fused multiply-adds as there.  C must not alias A or B; the caller separates phases by barriers.
template <bool TA, bool TB, bool QUAD, bool KMASK>
__device__ __forceinline__ void qcs_gemm_t(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C, int n, int ks,
                                           int kdim, double alpha) {
    constexpr int NT = QUAD ? 2 : 1, NK = QUAD ? 16 : 8, ld = QCS_LD;
    const int wave = qcs_wave(), lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int r0 = (wave >> 1) * 16, c0 = (wave & 1) * 16;
    const int a_s = TA ? 1 : ld, a_k = TA ? ld : 1, b_s = TB ? ld : 1, b_k = TB ? 1 : ld;
    double av[NT][NK], bv[NT][NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int kk = 4 * s + lk;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            av[t][s] = A[(r0 + 32 * t + li) * a_s + kk * a_k];
            bv[t][s] = B[(c0 + 32 * t + li) * b_s + kk * b_k];
            if constexpr (KMASK) { if (kk >= kdim) { av[t][s] = 0.0; bv[t][s] = 0.0; } }      // (inner range ends inside the matrix)
        }
    }
    qcs_d4 acc[NT][NT];
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y) acc[x][y] = qcs_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        if (s < ks) {                                      // (uniform, scalar)
#pragma unroll
            for (int x = 0; x < NT; ++x)
#pragma unroll
                for (int y = 0; y < NT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x][s], bv[y][s], acc[x][y], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int col = c0 + 32 * y + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 32 * x + lk + 4 * r;
                if (row < n && col < n) C[row * ld + col] = alpha * acc[x][y][r];
            }
        }
}
// KMASK: kdim < n (the density, C_occ C_occ^T): operands past kdim are not padding and are masked
template <bool TA, bool TB, bool KMASK = false>
__device__ __forceinline__ void qcs_gemm(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C, int n, int kdim,
                                         double alpha) {
    const int ks = (kdim + 3) >> 2;
    if (n > 32) qcs_gemm_t<TA, TB, true, KMASK>(A, B, C, n, ks, kdim, alpha);
    else qcs_gemm_t<TA, TB, false, KMASK>(A, B, C, n, ks, kdim, alpha);
}

// sum over the workgroup of NV values per thread, fixed order: lanes by shuffles, waves by thread j (nv <= NV of them are wanted)
template <int NV>
__device__ __forceinline__ void qcs_block_sums(const double (&v)[NV], int nv, double *red, double *out) {
    const int wave = qcs_wave(), lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (j < nv) {                                      // (uniform)
            double s = v[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) red[wave * 12 + j] = s;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < QCS_W; ++k) t += red[k * 12 + threadIdx.x];
        out[threadIdx.x] = t;
    }
    __syncthreads();
}

// DIIS coefficients (diis.rs:40-51): the M x M system [B 1; 1 0] c = (0, ..., 0, 1), M = window + 1, by Householder QR in the order of
// operations of nalgebra's qr().solve() (qc_qr_solve_reg, qc_linalg.hip), by ONE wave with the matrix in registers: lane j holds column j
// (lane M the right-hand side), the pivot column travels by v_readlane.  Returns false when a pivot vanishes ("DIIS failed").
template <int M>
__device__ __forceinline__ bool qcs_qr_solve(double (&col)[13], int lane, double (&x)[13]) {
    bool fail = false;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (!fail) {                                       // (uniform)
            double vk[M];
            double norm = 0.0;
#pragma unroll
            for (int i = k; i < M; ++i) { vk[i] = qcs_readlane(col[i], k); norm += vk[i] * vk[i]; }
            norm = sqrt(norm);
            if (norm == 0.0) fail = true;
            else {
                const double alpha = vk[k] > 0 ? -norm : norm;
                vk[k] -= alpha;
                double vn = 0.0;
#pragma unroll
                for (int i = k; i < M; ++i) vn += vk[i] * vk[i];
                if (vn > 0.0 && lane >= k) {               // columns k .. M-1 and the right-hand side
                    double d = 0.0;
#pragma unroll
                    for (int i = k; i < M; ++i) d += vk[i] * col[i];
                    d *= 2.0 / vn;
#pragma unroll
                    for (int i = k; i < M; ++i) col[i] -= d * vk[i];
                }
            }
        }
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        x[i] = 0.0;
        if (!fail) {                                       // (uniform)
            double s = qcs_readlane(col[i], M);
#pragma unroll
            for (int j = i + 1; j < M; ++j) s -= qcs_readlane(col[i], j) * x[j];
            const double piv = qcs_readlane(col[i], i);
            if (piv == 0.0) fail = true;
            else x[i] = s / piv;
        }
    }
    return !fail;
}

}  // namespace

constexpr double QCS_REF_TAU = 1e-3, QCS_REF_TINY = 1e-13, QCS_REF_GFLOOR = 1e-8;      // as qc_linalg.hip (QC_REF_*)

__global__ __launch_bounds__(QCS_T) void qc_scf_small_kernel(const QcSmallArgs a) {
    extern __shared__ double lds[];
    constexpr int ld = QCS_LD, BUF = QCS_BUF;
    const int n = a.n;
    double *B0 = lds, *B1 = lds + BUF, *B2 = lds + 2 * BUF, *B3 = lds + 3 * BUF;
    double *sm = lds + 4 * BUF;
    double *lam = sm, *red = lam + 64, *dots = red + QCS_W * 12, *cvec = dots + 12, *Bl = cvec + 12, *scal = Bl + 144, *gsh = scal + 8;
    int *partner = reinterpret_cast<int *>(gsh + 12 * 16), *rank_s = partner + 64, *flg = rank_s + 64;
    const int tid = threadIdx.x, rr = tid >> 6, cc = tid & 63;           // this thread's elements: rows rr + 4 q, column cc
    const bool cok = cc < n;
    if (a.tl != nullptr && tid == 0) a.tl[0] = wall_clock64();
    bool ok = true;                                                      // (uniform) the eigenvectors for `post` exist
    double *Cpv = B0;                                                    // ... and the block they are in
#ifdef QC_SMALL_TIMING
    long long tph[40] = {}; int nph = 0;
#define QCS_STAMP() do { if (tid == 0 && nph < 40) tph[nph++] = wall_clock64(); } while (0)
    QCS_STAMP();
#else
#define QCS_STAMP() do {} while (0)
#endif
    // element loops: all of a thread's loads are requested before the first use; out-of-range elements read element 0 and are not stored
#define QCS_EACH(q, i) _Pragma("unroll") for (int q = 0, i = rr; q < QCS_E; ++q, i += QCS_W)
#define QCS_IN(i) (cok && (i) < n)
#define QCS_GX(i) (QCS_IN(i) ? (i) * n + cc : 0)

    // the padding of the four blocks (and the small arrays) is zero for the whole kernel: products run over 64 x 64 without bounds tests
    for (int x = tid; x < 4 * BUF + 64; x += QCS_T) lds[x] = 0.0;
    if (tid < 64) partner[tid] = -1;
    __syncthreads();

    if (a.phases & 1) {
        // ---- e = F D S - S D F
        {
            double f[QCS_E], d[QCS_E], s[QCS_E], g[QCS_E];
            const bool haveF = a.F != nullptr;                           // (uniform)
            const double *__restrict__ fsrc = haveF ? a.F : a.H;
            QCS_EACH(q, i) {
                const int x = QCS_GX(i);
                f[q] = fsrc[x]; d[q] = a.D[x]; s[q] = a.S[x];
                g[q] = haveF ? 0.0 : a.G[x];
            }
            if (tid < a.maxlen * a.maxlen) Bl[tid] = a.Bmat[tid];
            QCS_EACH(q, i)
                if (QCS_IN(i)) {
                    double fv = f[q];
                    if (!haveF) { fv = 1.0 * f[q] + 1.0 * g[q]; a.F_out[i * n + cc] = fv; }    // the arithmetic of qc_axpby(1, H, 1, G)
                    B0[i * ld + cc] = fv; B1[i * ld + cc] = d[q]; B2[i * ld + cc] = s[q];
                }
        }
        __syncthreads();
        QCS_STAMP();
        qcs_gemm<false, false>(B0, B1, B3, n, n, 1.0);                   // F D
        __syncthreads();
        QCS_STAMP();
        qcs_gemm<false, false>(B3, B2, B0, n, n, 1.0);                   // (F D) S
        __syncthreads();
        QCS_STAMP();
        {
            double e[QCS_E], part[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) part[j] = 0.0;
            QCS_EACH(q, i) {
                const int ic = min(i, 63);
                e[q] = B0[ic * ld + cc] - B0[cc * ld + ic];              // FDS - (FDS)^T = FDS - SDF (zero in the padding)
                if (QCS_IN(i)) a.E_out[i * n + cc] = e[q];
                part[0] = fma(e[q], e[q], part[0]);
            }
            // <e_0, e_j>, diis.rs:43-45: the older error matrices three at a time (48 loads in flight)
#pragma unroll
            for (int j0 = 1; j0 < 12; j0 += 3) {
                if (j0 < a.m) {                                          // (uniform)
                    double o[3][QCS_E];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const double *__restrict__ src = a.errs[j0 + u < a.m ? j0 + u : 0];      // (past the window: any valid matrix, weight 0)
                        QCS_EACH(q, i) o[u][q] = src[QCS_GX(i)];
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        constexpr int JMAX = 11;
                        const int j = j0 + u < JMAX ? j0 + u : JMAX;
                        const double w = j0 + u < a.m ? 1.0 : 0.0;
                        QCS_EACH(q, i) part[j] = fma(e[q] * w, o[u][q], part[j]);                // (e is zero outside the matrix)
                    }
                }
            }
            qcs_block_sums<12>(part, a.m, red, dots);
            if (a.dots_generic) {
                // Open-shell runs: the dot products in the summation order of the generic launch sequence (qc_dots_kernel, qc_linalg.hip: 1024
                // threads, thread t sums the elements t + 1024 u in ascending u, lanes by the shuffle tree, the sixteen wave sums in
                // order).  The trajectories of such runs - crawls along saddles of the UHF functional, DESIGN.md 1 - depend on the last bit
                // of these numbers, and the one that is pinned against the oracle is the generic sequence's.  The error matrix goes to a flat
                // copy in B3 (free between the two products around here), the 1024 virtual threads are served in four rounds.
                const int nn = n * n;
                QCS_EACH(q, i) if (QCS_IN(i)) B3[i * n + cc] = e[q];
                __syncthreads();
                const int wave = qcs_wave(), lane = tid & 63;
                for (int j = 0; j < a.m; ++j) {
                    const double *__restrict__ yj = a.errs[j];
                    for (int r = 0; r < 4; ++r) {
                        const int t = r * QCS_T + tid;
                        double av[4], bv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int x = min(t + u * 1024, nn - 1); av[u] = B3[x]; bv[u] = j == 0 ? B3[x] : yj[x]; }
                        double sgen = 0.0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (t + u * 1024 < nn) sgen = fma(av[u], bv[u], sgen);
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) sgen += __shfl_down(sgen, o, 64);
                        if (lane == 0) gsh[j * 16 + r * QCS_W + wave] = sgen;
                    }
                }
                __syncthreads();
                if (tid < a.m) {
                    double tsum = 0.0;
                    for (int k = 0; k < 16; ++k) tsum += gsh[tid * 16 + k];
                    dots[tid] = tsum;
                }
                for (int x = tid; x < BUF; x += QCS_T) B3[x] = 0.0;      // (the block's padding is zero for the rest of the kernel)
                __syncthreads();
            }
        }
        QCS_STAMP();
        // ---- DIIS coefficients by wave 0; the other three waves fetch X meanwhile.  The first three Fock matrices of the combination are
        // requested before the solve (they do not depend on it) and arrive while it runs.
        double fo0[3][QCS_E];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const double *__restrict__ src = a.focks[u < a.m ? u : 0];
            QCS_EACH(q, i) fo0[u][q] = src[QCS_GX(i)];
        }
        if (qcs_wave() == 0) {
            const int lane = tid, m = a.m, M = m + 1, ML = a.maxlen;
            if (lane < m) {
                const double d = dots[lane];
                const int p = a.slot[0] * ML + a.slot[lane], q = a.slot[lane] * ML + a.slot[0];
                Bl[p] = d; Bl[q] = d; a.Bmat[p] = d; a.Bmat[q] = d;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double cj = lane == 0 ? 1.0 : 0.0;                           // diis.rs:33-38: the newest Fock matrix while the window is short
            if (m >= a.minlen) {
                double col[13], x[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    double v = 0.0;
                    if (i < M && lane <= M) {
                        if (lane == M) v = i == m ? 1.0 : 0.0;                                   // right-hand side
                        else if (i < m && lane < m) v = Bl[a.slot[i] * ML + a.slot[lane]];
                        else v = (i == m && lane == m) ? 0.0 : 1.0;                              // border +1, corner 0
                    }
                    col[i] = v;
                    x[i] = 0.0;
                }
                bool solved = false;
                switch (M) {
#define QCS_QR(MM) case MM: solved = qcs_qr_solve<MM>(col, lane, x); break;
                    QCS_QR(2) QCS_QR(3) QCS_QR(4) QCS_QR(5) QCS_QR(6) QCS_QR(7) QCS_QR(8) QCS_QR(9) QCS_QR(10) QCS_QR(11) QCS_QR(12) QCS_QR(13)
#undef QCS_QR
                    default: break;
                }
                if (!solved) { if (lane == 0) *a.diis_flag = 1; }        // "DIIS failed" (rhf.rs:73); c stays (1, 0, ...)
                else {
                    cj = 0.0;
#pragma unroll
                    for (int j = 0; j < 12; ++j) if (lane == j && j < m) cj = x[j];
                }
            }
            if (lane < 12) { cvec[lane] = cj; a.c_out[lane] = cj; }
        } else {
            const int w3 = qcs_wave() - 1;                               // rows w3 + 3 q
            double xr[22];
#pragma unroll
            for (int q = 0; q < 22; ++q) { const int i = w3 + 3 * q; xr[q] = a.X[(cok && i < n) ? i * n + cc : 0]; }
#pragma unroll
            for (int q = 0; q < 22; ++q) { const int i = w3 + 3 * q; if (cok && i < n) B2[i * ld + cc] = xr[q]; }
        }
        __syncthreads();
        QCS_STAMP();
        // ---- F_diis = sum_j c_j F_j (diis.rs:52-58), three Fock matrices at a time; F' = X^T F_diis X
        {
            double acc[QCS_E];
            QCS_EACH(q, i) acc[q] = 0.0;
#pragma unroll
            for (int j0 = 0; j0 < 12; j0 += 3) {
                if (j0 < a.m) {                                          // (uniform)
                    double o[3][QCS_E];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        if (j0 == 0) { QCS_EACH(q, i) o[u][q] = fo0[u][q]; }
                        else {
                            const double *__restrict__ src = a.focks[j0 + u < a.m ? j0 + u : 0];
                            QCS_EACH(q, i) o[u][q] = src[QCS_GX(i)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        if (j0 + u < a.m) {                              // (uniform)
                            const double c = cvec[j0 + u];
                            QCS_EACH(q, i) acc[q] = fma(c, o[u][q], acc[q]);
                        }
                    }
                }
            }
            QCS_EACH(q, i) if (QCS_IN(i)) B0[i * ld + cc] = acc[q];
        }
        __syncthreads();
        QCS_STAMP();
        qcs_gemm<false, false>(B0, B2, B1, n, n, 1.0);                   // F X
        __syncthreads();
        QCS_STAMP();
        qcs_gemm<true, false>(B2, B1, B3, n, n, 1.0);                    // X^T (F X)
        __syncthreads();
        QCS_STAMP();
        QCS_EACH(q, i) if (QCS_IN(i)) a.Fp[i * n + cc] = B3[i * ld + cc];
    }

    if (a.phases & 2) {
        // ---- eigenvectors of F' (B3) by refinement from V0: the passes of qc_eig_refine_async (qc_linalg.hip) back to back
        {
            double v[QCS_E], f[QCS_E];
            const bool loadA = !(a.phases & 1);                          // (uniform)
            const double *__restrict__ fsrc = loadA ? a.Fp : a.V0;
            QCS_EACH(q, i) { const int x = QCS_GX(i); v[q] = a.V0[x]; f[q] = fsrc[x]; }
            QCS_EACH(q, i) if (QCS_IN(i)) { B0[i * ld + cc] = v[q]; if (loadA) B3[i * ld + cc] = f[q]; }
        }
        if (tid < 4) flg[tid] = 0;
        __syncthreads();
        QCS_STAMP();
        double *X = B0, *Xn = B2;
        for (int pass = 0; pass < a.npass; ++pass) {
            qcs_gemm<false, false>(B3, X, B1, n, n, 1.0);                // A X
            __syncthreads();
            qcs_gemm<true, false>(X, B1, Xn, n, n, 1.0);                 // S = X^T A X
            __syncthreads();
            qcs_gemm<true, false>(X, X, B1, n, n, 1.0);                  // X^T X
            __syncthreads();
            QCS_STAMP();
            const double *Sm = Xn, *XtX = B1;
            // statistics and decisions (qc_refine_stats_kernel); everything reads the zero padding instead of testing bounds
            {
                double part[2] = {0.0, 0.0};
                double amax = 0.0;
                QCS_EACH(q, i) {
                    const int ic = min(i, 63);
                    const bool in = QCS_IN(i), dg = ic == cc;
                    const double sv = Sm[ic * ld + cc], r = ((in && dg) ? 1.0 : 0.0) - XtX[ic * ld + cc];
                    part[1] = fma(r, r, part[1]);
                    part[0] = dg ? part[0] : fma(sv, sv, part[0]);
                    if (in && dg) { const double l = sv / (1.0 - r); lam[ic] = l; amax = fmax(amax, fabs(l)); }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, 64));
                qcs_block_sums<2>(part, 2, red, scal);                   // scal[0] = ||off S||^2, scal[1] = ||R||^2
                if ((tid & 63) == 0) red[tid >> 6] = amax;
                if (tid == 0) { flg[4] = 0; flg[5] = 0; flg[6] = 0; flg[7] = 0; }      // multi, nstrong, big update, not the last update
                __syncthreads();
                if (tid == 0) { double c = 0.0; for (int k = 0; k < QCS_W; ++k) c = fmax(c, red[k]); scal[2] = c; }
                __syncthreads();
            }
            const double scale = scal[2], tiny = QCS_REF_TINY * scale, gfloor = QCS_REF_GFLOOR * scale;
            {
                double cmax = 0.0;
                int big = 0, notlast = 0;                                // some regular pair has |a| > 0.1 g / > 1e-7 g (update size, without the division)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {                         // one row per team of 16 lanes
                    const int i = (tid >> 4) + 16 * ii;
                    int cnt = 0, who = -1;
                    const double li_ = lam[i];                           // (zero past n: such rows find av = 0 everywhere)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = (tid & 15) + 16 * jj;
                        const double lj = lam[j];
                        const double r = -XtX[i * ld + j];
                        const double sij = 0.5 * (Sm[i * ld + j] + Sm[j * ld + i]);
                        const double av = fabs(sij + 0.5 * (li_ + lj) * r), g = fabs(lj - li_);
                        const bool valid = j != i && av > tiny;
                        const bool strong = valid && av > QCS_REF_TAU * fmax(g, gfloor);
                        cnt += strong ? 1 : 0;
                        who = strong ? j : who;
                        const bool weak = valid && !strong;
                        cmax = (weak && g <= gfloor) ? fmax(cmax, av) : cmax;
                        const bool reg = weak && g > gfloor;
                        big |= (reg && !(av <= 0.1 * g)) ? 1 : 0;
                        notlast |= (reg && !(av <= 1e-7 * g)) ? 1 : 0;
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 16); who = max(who, __shfl_xor(who, o, 16)); }
                    if ((tid & 15) == 0 && i < n) {
                        partner[i] = cnt == 0 ? -1 : (cnt == 1 ? who : -2);
                        if (cnt > 1) flg[4] = 1;
                        if (cnt == 1) atomicAdd(&flg[5], 1);
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cmax = fmax(cmax, __shfl_down(cmax, o, 64));
                if (big) flg[6] = 1;
                if (notlast) flg[7] = 1;
                __syncthreads();
                if ((tid & 63) == 0) red[tid >> 6] = cmax;
                if (tid < n) { const int pj = partner[tid]; if (pj >= 0 && partner[pj] != tid) flg[4] = 1; }     // a strong pair must be mutual
                __syncthreads();
                if (tid == 0) {
                    double ca = 0.0;
                    for (int k = 0; k < QCS_W; ++k) ca = fmax(ca, red[k]);
                    const double orth = sqrt(scal[1]), scl = fmax(scale, 1e-300);
                    const int multi = flg[4];
                    if (flg[6] || !(orth <= 1e-3) || multi) flg[0] = 2;                          // not perturbative: rotations needed
                    else {
                        flg[1] = (!flg[7] && orth <= 1e-7 && flg[5] == 0) ? 1 : 0;               // one more update finishes
                        flg[2] = (ca <= 1e-12 * scl) ? 1 : 0;                                    // no coupling left inside degenerate pairs
                    }
                }
                __syncthreads();
            }
            QCS_STAMP();
            if (flg[0] != 0) break;
            // M = I + E with exact rotations on the strong pairs (qc_refine_m), into B1 in place of X^T X
            {
                double mv[QCS_E];
                const double lj = lam[cc];
                QCS_EACH(q, i) {
                    const int ic = min(i, 63), j = cc;
                    const double li_ = lam[ic], xij = XtX[ic * ld + j];
                    const double sij = 0.5 * (Sm[ic * ld + j] + Sm[j * ld + ic]);
                    const double r = (ic == j ? 1.0 : 0.0) - xij;
                    const double av = sij + 0.5 * (li_ + lj) * r, g = lj - li_;
                    const int pi_ = partner[ic];
                    double mm = (fabs(av) <= tiny || fabs(g) <= gfloor) ? 0.5 * r : (sij + lj * r) / g;         // regular / negligible / degenerate
                    if (ic == j) mm = 1.0 + 0.5 * r;
                    if (pi_ >= 0 && (pi_ == j || ic == j)) {             // this index rotates with its strong partner
                        const double avp = ic == j ? 0.5 * (Sm[ic * ld + pi_] + Sm[pi_ * ld + ic]) - 0.5 * (li_ + lam[pi_]) * XtX[ic * ld + pi_] : av;
                        const double gp = ic == j ? lam[pi_] - li_ : g;
                        const double theta = gp / (2.0 * avp);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
                        mm = ic == j ? 1.0 / sqrt(fma(t, t, 1.0)) + 0.5 * r : t / sqrt(fma(t, t, 1.0));
                    }
                    mv[q] = mm;
                }
                __syncthreads();
                QCS_EACH(q, i) if (QCS_IN(i)) B1[i * ld + cc] = mv[q];
            }
            __syncthreads();
            QCS_STAMP();
            qcs_gemm<false, false>(X, B1, Xn, n, n, 1.0);                // X (I + E): error now ~ emax^2
            __syncthreads();
            QCS_STAMP();
            if (tid == 0) flg[3] += 1;                                   // passes used
            if (flg[1] && flg[2]) {                                      // final vectors: ascending eigenvalues, columns alongside
                if (tid < n) {
                    const double wi = lam[tid];
                    int r = 0;
                    for (int j = 0; j < n; ++j) r += (lam[j] < wi || (lam[j] == wi && j < tid)) ? 1 : 0;
                    rank_s[tid] = r;
                    a.w_out[r] = wi;
                }
                __syncthreads();
                const int rc = cok ? rank_s[cc] : 0;
                QCS_EACH(q, i) if (QCS_IN(i)) X[i * ld + rc] = Xn[i * ld + cc];
                if (tid == 0) flg[0] = 1;
                __syncthreads();
                break;
            }
            if (flg[1] || pass == a.npass - 1) {                         // coupling inside a degenerate cluster, or passes exhausted
                __syncthreads();
                if (tid == 0) flg[0] = 2;
                __syncthreads();
                break;
            }
            double *t = X; X = Xn; Xn = t;
            __syncthreads();
        }
        if (tid < 4) a.ctl[tid] = flg[tid];
        ok = flg[0] == 1;
        if (ok) {
            QCS_EACH(q, i) if (QCS_IN(i)) a.Cp_out[i * n + cc] = X[i * ld + cc];
            Cpv = X;
        }
        __syncthreads();
        QCS_STAMP();
    }

    if ((a.phases & 4) && ok) {
        // ---- C = X C', D = dfac C_occ C_occ^T, energy, rms
        double h[QCS_E], g[QCS_E];                                       // H and G of this thread's elements: requested now, used at the end
        double dold = 0.0;
        {
            double xv[QCS_E], cp[QCS_E];
            const bool loadC = !(a.phases & 2);                          // (uniform)
            const double *__restrict__ csrc = loadC ? a.Cp_in : a.X;
            QCS_EACH(q, i) {
                const int x = QCS_GX(i);
                xv[q] = a.X[x]; cp[q] = csrc[x];
                h[q] = a.H[x]; g[q] = a.G[x];
            }
            dold = a.Dold[tid < n ? tid * n + tid : 0];
            QCS_EACH(q, i) if (QCS_IN(i)) { B1[i * ld + cc] = xv[q]; if (loadC) B0[i * ld + cc] = cp[q]; }
        }
        __syncthreads();
        QCS_STAMP();
        double *const Cm = Cpv == B2 ? B0 : B2;                          // (the block the eigenvectors are not in)
        qcs_gemm<false, false>(B1, Cpv, Cm, n, n, 1.0);                  // C
        __syncthreads();
        if (a.nocc > 0) qcs_gemm<false, true, true>(Cm, Cm, B3, n, a.nocc, a.dfac);         // C_occ C_occ^T
        else QCS_EACH(q, i) if (QCS_IN(i)) B3[i * ld + cc] = 0.0;
        QCS_EACH(q, i) if (QCS_IN(i)) a.C_out[i * n + cc] = Cm[i * ld + cc];
        __syncthreads();
        QCS_STAMP();
        double part[3] = {0.0, 0.0, 0.0};
        QCS_EACH(q, i) {
            const int ic = min(i, 63);
            const double dn = B3[ic * ld + cc];                                                  // (zero in the padding)
            if (QCS_IN(i)) a.Dn[i * n + cc] = dn;
            part[0] = fma(QCS_IN(i) ? B3[cc * ld + ic] : 0.0, 2.0 * h[q] + g[q], part[0]);      // tr(Dn (2H + G)) = sum_ij Dn_ji (2H + G)_ij
            part[2] += fabs(dn);
        }
        if (tid < n) { const double d = B3[tid * ld + tid] - dold; part[1] = d * d; }
        qcs_block_sums<3>(part, 3, red, scal);
        if (tid == 0) {
            a.scal_out[0] = 0.5 * scal[0]; a.scal_out[1] = scal[1];
            if (a.fxs_out) {     // fixed-point unit of the build that will digest Dn (qc_fx_scale_kernel, qc_linalg.hip): 2^S (4 imax sum|D|) <= 2^60
                const double bound = 4.0 * a.imax * scal[2];
                int e = 0;
                if (bound > 0.0 && bound < 1e300) (void)frexp(bound, &e);
                else if (!(bound < 1e300)) e = 1100;
                int S = 60 - e;
                S = S > QC_FX_MAXBITS ? QC_FX_MAXBITS : (S < -900 ? -900 : S);
                a.fxs_out[0] = ldexp(1.0, S); a.fxs_out[1] = (bound < 1e300) ? ldexp(1.0, -S) : __builtin_nan("");
            }
        }
        QCS_STAMP();
    }
    if (a.ctl_all) {
        __threadfence();
        __syncthreads();
        if (tid < 16) {
            int *p = a.ctl_all + tid;
            a.ctl_out[tid] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __threadfence_system();
    if (a.fork_words) {
        // A speculative build of the next pass is queued behind this kernel: its side streams wait for the fork word, its class kernels look
        // at the cancel word (qc_fock.hip, "Device-side fork").  If this pass meets the stopping rule the host has announced (rhf.rs:94:
        // rms < epsilon, rms = sqrt(sum_i dD_ii^2 / n) as the host forms it), that build is emptied - and the host told so, in pinned memory,
        // before it can see the pass end.  When the eigensolve wants a repeat (!ok) the host discards the build; it is released all the same.
        __syncthreads();
        if (tid == 0) {
            if (a.eps > 0.0 && (a.phases & 4) && ok && sqrt(scal[1] / n) < a.eps) {
                __hip_atomic_store(a.fork_words + 2, a.fork_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.h_cancel) __hip_atomic_store(a.h_cancel, a.fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(a.fork_words + 1, a.fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.tl != nullptr && tid == 0) a.tl[1] = wall_clock64();
    if (a.seq_out) {     // the host polls this word instead of the stream's event (which the packet processor signals some microseconds later)
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.seq_out, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef QC_SMALL_TIMING
    if (tid == 0) {
        printf("[small n=%d phases %d m %d] stamps (10 ns):", n, a.phases, a.m);
        for (int k = 1; k < nph; ++k) printf(" %lld", tph[k] - tph[k - 1]);
        printf("\n");
    }
#endif
#undef QCS_EACH
#undef QCS_IN
#undef QCS_GX
#undef QCS_STAMP
}

size_t qc_scf_small_lds_bytes(int) { return (size_t)(4 * QCS_BUF + QCS_SMALL_DOUBLES) * sizeof(double) + QCS_SMALL_INTS * sizeof(int); }

int qc_scf_small_launch(hipStream_t st, const QcSmallArgs &a) {
    if (a.n < 1 || a.n > QC_SMALL_MAXN || a.m < 1 || a.m > 12) return QC_ERR_INVALID;
    const size_t lds = qc_scf_small_lds_bytes(a.n);
    static std::atomic<bool> raised{false};
    if (!raised.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(qc_scf_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return QC_ERR_HIP;
        raised.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(qc_scf_small_kernel, dim3(1), dim3(QCS_T), lds, st, a);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}
