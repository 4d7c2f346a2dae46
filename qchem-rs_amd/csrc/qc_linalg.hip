// qc_linalg.hip - dense f64 linear algebra of the SCF iteration on gfx950: MFMA GEMM, symmetric eigensolver,
// element-wise and reduction kernels.  All matrices row-major, device pointers, on the caller's stream.
//
// Replaces the nalgebra operations in the reference's loop body: the `*` products at rhf.rs:71,74,76,85,
// SymmetricEigen behind utils::sorted_eigs (hf/utils.rs:20-36), the Frobenius dots of diis.rs:43-45 and the
// trace / diagonal-rms at rhf.rs:84-88.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "qc_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// C = alpha * op(A) * op(B) + beta * C with v_mfma_f64_16x16x4_f64.  One wave per 16x16 tile of C.
// Operand lane maps (cdna_hip_programming.md section 3): lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r holds C[row = (l >> 4) + 4 r][col = l & 15].
struct QcGemmArgs { int m, n, k; double alpha; const double *A; int lda, ta; const double *B; int ldb, tb; double beta; double *C; int ldc; };

__device__ __forceinline__ void qc_gemm_tile(const QcGemmArgs &g) {
    const int m = g.m, n = g.n, k = g.k;
    const double *__restrict__ A = g.A, *__restrict__ B = g.B;
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    const int ai = row0 + li, bj = col0 + li;
    const bool aok = ai < m, bok = bj < n;
    const size_t a_i = g.ta ? (size_t)ai : (size_t)ai * g.lda, a_k = g.ta ? (size_t)g.lda : 1;
    const size_t b_j = g.tb ? (size_t)bj * g.ldb : (size_t)bj, b_k = g.tb ? 1 : (size_t)g.ldb;
    // The operands of 64 k-values (4 x 4 MFMA steps) are requested before the first MFMA of the chunk: a tile is a chain of L2
    // round trips otherwise, one per 16 k-values.
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < k; k0 += 64) {
        double av[16], bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kk = k0 + 4 * s + lk;
            av[s] = (aok && kk < k) ? A[a_i + kk * a_k] : 0.0;
            bv[s] = (bok && kk < k) ? B[b_j + kk * b_k] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (k0 + 4 * s < k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);     // (uniform)
    }
    if (bok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + lk + 4 * r;
            if (row < m) {
                double *c = &g.C[(size_t)row * g.ldc + bj];
                *c = (g.beta == 0.0) ? g.alpha * acc[r] : fma(g.alpha, acc[r], g.beta * *c);
            }
        }
    }
}

__global__ __launch_bounds__(64) void qc_gemm_kernel(const QcGemmArgs g, const int *__restrict__ skip) {
    if (skip && *skip != 0) return;                        // device-side control flow of the sync-free SCF step
    qc_gemm_tile(g);
}

// two independent products of the same shape in one launch (blockIdx.z picks the problem)
__global__ __launch_bounds__(64) void qc_gemm2_kernel(const QcGemmArgs g0, const QcGemmArgs g1, const int *__restrict__ skip) {
    if (skip && *skip != 0) return;
    qc_gemm_tile(blockIdx.z ? g1 : g0);
}

void qc_gemm(hipStream_t st, int m, int n, int k, double alpha, const double *A, int lda, bool ta, const double *B, int ldb, bool tb,
             double beta, double *C, int ldc, const int *skip) {
    dim3 grid((n + 15) / 16, (m + 15) / 16);
    const QcGemmArgs g{m, n, k, alpha, A, lda, ta ? 1 : 0, B, ldb, tb ? 1 : 0, beta, C, ldc};
    hipLaunchKernelGGL(qc_gemm_kernel, grid, dim3(64), 0, st, g, skip);
}

// C0 = op(A0) op(B0), C1 = op(A1) op(B1), all n x n
static void qc_gemm_pair(hipStream_t st, int n, const double *A0, bool ta0, const double *B0, double *C0, const double *A1, bool ta1,
                         const double *B1, double *C1, const int *skip) {
    dim3 grid((n + 15) / 16, (n + 15) / 16, 2);
    const QcGemmArgs g0{n, n, n, 1.0, A0, n, ta0 ? 1 : 0, B0, n, 0, 0.0, C0, n}, g1{n, n, n, 1.0, A1, n, ta1 ? 1 : 0, B1, n, 0, 0.0, C1, n};
    hipLaunchKernelGGL(qc_gemm2_kernel, grid, dim3(64), 0, st, g0, g1, skip);
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric eigensolver: parallel cyclic two-sided Jacobi in ONE workgroup, matrix (and, when it fits, the
// eigenvector matrix) resident in LDS.  Round-robin ("chess tournament") ordering gives m/2 disjoint rotations per
// step and m-1 steps per sweep.  Each step is two phases separated by one barrier each:
//   A. m/2 threads compute the rotations (c_k, s_k) of the step's pairs (p_k, q_k);
//   B. the similarity transform J^T A J is applied as (m/2)^2 independent 2x2 blocks - block (k,l) = rows {p_k,q_k} x
//      columns {p_l,q_l} gets J_k^T . B . J_l - and V <- V J as n x m/2 independent row pairs.
// Output: eigenvalues ascending in w, eigenvectors as the columns of Vout (utils::sorted_eigs, hf/utils.rs:20-36).
constexpr int QC_EIG_THREADS = 1024;

__device__ __forceinline__ void qc_rr_pair(int step, int k, int m, int &p, int &q) {
    if (k == 0) { p = m - 1; q = step; return; }
    p = step + k; if (p >= m - 1) p -= m - 1;
    q = step - k; if (q < 0) q += m - 1;
}

template <bool V_IN_LDS>
__global__ __launch_bounds__(QC_EIG_THREADS) void qc_jacobi_kernel(int n, const double *__restrict__ Ain, double *__restrict__ Vg,
                                                                   double *__restrict__ Vout, double *__restrict__ w, int max_sweeps, double done_tol,
                                                                   int *__restrict__ notconv) {
    extern __shared__ double sm[];
    const int m = (n + 1) & ~1, half = m / 2, ld = m | 1;
    double *A = sm;                                         // m x ld (padding row/column stay zero => identity rotations)
    double *V = V_IN_LDS ? A + (size_t)m * ld : Vg;         // m x ldv
    const int ldv = V_IN_LDS ? ld : n;
    double *cs = A + (size_t)m * ld * (V_IN_LDS ? 2 : 1);   // 2 * half
    double *red = cs + 2 * half;                            // 32
    int *rank = (int *)(red + 32);                          // m
    const int tid = threadIdx.x, nt = blockDim.x;

    double fro = 0.0;
    for (int x = tid; x < m * m; x += nt) {
        const int i = x / m, j = x - i * m;
        const double v = (i < n && j < n) ? Ain[(size_t)i * n + j] : 0.0;
        A[i * ld + j] = v;
        fro = fma(v, v, fro);
        if (V_IN_LDS) V[i * ldv + j] = (i == j) ? 1.0 : 0.0;
        else if (i < n && j < n) V[(size_t)i * ldv + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) fro += __shfl_down(fro, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    __syncthreads();
    double normF2 = 0.0;
    for (int k = 0; k < nt / 64; ++k) normF2 += red[k];
    __syncthreads();

    bool conv = false;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        double seen = 0.0;                                  // sum of a_pq^2 at the moment each pair is rotated
        for (int step = 0; step < m - 1; ++step) {
            if (tid < half) {                               // phase A
                int p, q;
                qc_rr_pair(step, tid, m, p, q);
                const double apq = A[p * ld + q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double theta = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
                    c = 1.0 / sqrt(fma(t, t, 1.0));
                    s = t * c;
                    seen = fma(apq, apq, seen);
                }
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
            }
            __syncthreads();
            for (int b = tid; b < half * half; b += nt) {   // phase B: 2x2 blocks of J^T A J
                const int k = b / half, l = b - k * half;
                const double ck = cs[2 * k], sk = cs[2 * k + 1], cl = cs[2 * l], sl = cs[2 * l + 1];
                if (sk == 0.0 && sl == 0.0) continue;
                int pk, qk, pl, ql;
                qc_rr_pair(step, k, m, pk, qk);
                qc_rr_pair(step, l, m, pl, ql);
                const double b00 = A[pk * ld + pl], b01 = A[pk * ld + ql], b10 = A[qk * ld + pl], b11 = A[qk * ld + ql];
                const double r00 = ck * b00 - sk * b10, r01 = ck * b01 - sk * b11;      // J_k^T B
                const double r10 = sk * b00 + ck * b10, r11 = sk * b01 + ck * b11;
                A[pk * ld + pl] = cl * r00 - sl * r01; A[pk * ld + ql] = sl * r00 + cl * r01;   // (.) J_l
                A[qk * ld + pl] = cl * r10 - sl * r11; A[qk * ld + ql] = sl * r10 + cl * r11;
            }
            for (int x = tid; x < n * half; x += nt) {      // V <- V J
                const int i = x / half, l = x - i * half;
                const double cl = cs[2 * l], sl = cs[2 * l + 1];
                if (sl == 0.0) continue;
                int pl, ql;
                qc_rr_pair(step, l, m, pl, ql);
                const double vp = V[(size_t)i * ldv + pl], vq = V[(size_t)i * ldv + ql];
                V[(size_t)i * ldv + pl] = cl * vp - sl * vq;
                V[(size_t)i * ldv + ql] = sl * vp + cl * vq;
            }
            __syncthreads();
        }
        // off-diagonal mass met during this sweep; the sweep itself reduced it (quadratically, once small)
        for (int o = 32; o > 0; o >>= 1) seen += __shfl_down(seen, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = seen;
        __syncthreads();
        double off2 = 0.0;
        for (int k = 0; k < nt / 64; ++k) off2 += red[k];
        __syncthreads();
        // Jacobi converges quadratically: a sweep that met a relative off-norm <= done_tol (1e-9) leaves <= ~done_tol^2 behind
        if (2.0 * off2 <= done_tol * done_tol * normF2) { conv = true; break; }
    }
    if (!conv && notconv && tid == 0) *notconv = 1;         // sweeps exhausted: the caller reports it instead of using the vectors
    // ascending order (utils.rs:28): rank sort, then permute columns
    for (int i = tid; i < n; i += nt) {
        const double wi = A[i * ld + i];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const double wj = A[j * ld + j];
            r += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
        }
        rank[i] = r;
        w[r] = wi;
    }
    __syncthreads();
    for (int x = tid; x < n * n; x += nt) {
        const int i = x / n, j = x - i * n;
        Vout[(size_t)i * n + rank[j]] = V[(size_t)i * ldv + j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi for matrices whose A and V do not both fit in LDS (98 < n <= 128).
// B = A + sigma I is made positive definite with a Gershgorin shift; the columns of G (initially B) are rotated
// pairwise to mutual orthogonality, G <- G J.  At convergence G = U Sigma: the normalised columns are the
// eigenvectors and ||g_i|| - sigma the eigenvalues, so only ONE n x n matrix has to live in LDS (column-major,
// 8 lanes per column pair, conflict-free reads).  One barrier per step.  The rotation parameters are computed by every
// lane of a team, so the kernel is bound by that scalar chain times the number of resident waves: narrow teams (fewer
// waves) and reciprocal-square-root forms (no IEEE divide / sqrt sequences) make the step 2.4x shorter than the
// 16-lane / divide version.
constexpr int QC_EIG1_TEAM = 8;
constexpr int QC_EIG1_ROWS = 128 / QC_EIG1_TEAM;       // rows per lane of the LDS variant (n <= 128)
constexpr int QC_EIG1_THREADS = 64 * QC_EIG1_TEAM;     // 64 teams: one per column pair

// 1 / x to a few ulp: hardware estimate + two Newton steps (the IEEE division sequence is twice as long, and every lane of a
// team runs this chain between the barriers of a step)
__device__ __forceinline__ double qc_fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// Jacobi rotation (c, s) that orthogonalises two columns with squared norms a, b and inner product g.  c^2 + s^2 = 1 holds to
// the accuracy of rsqrt whatever the error of t, so the reciprocal shortcuts cost nothing in orthonormality.
__device__ __forceinline__ void qc_hestenes_rotation(double a, double b, double g, double &cs, double &sn) {
    const double zeta = (b - a) * (0.5 * qc_fast_rcp(g));
    const double z2 = fma(zeta, zeta, 1.0);
    const double den = fabs(zeta) + z2 * rsqrt(z2);            // |zeta| + sqrt(zeta^2 + 1)
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) * qc_fast_rcp(den);
    cs = rsqrt(fma(t, t, 1.0));
    sn = cs * t;
}

__global__ __launch_bounds__(QC_EIG1_THREADS) void qc_jacobi1_kernel(int n, const double *__restrict__ Ain, double *__restrict__ Vout,
                                                                    double *__restrict__ w, int max_sweeps, double done_tol, int *__restrict__ notconv) {
    extern __shared__ double sm[];
    const int m = (n + 1) & ~1, half = m / 2, ld = n | 1;        // column stride (doubles)
    double *G = sm;                                               // m columns x ld (padding column stays zero)
    double *red = G + (size_t)m * ld;                             // 32
    double *nrm = red + 32;                                       // m column norms
    int *rank = (int *)(nrm + m);                                 // m
    int *flag = rank + m;                                         // rotations done in this sweep
    const int tid = threadIdx.x, nt = blockDim.x;

    // Gershgorin lower bound of the spectrum -> shift
    double low = 1e300, spread = 0.0;
    for (int i = tid; i < n; i += nt) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) if (j != i) r += fabs(Ain[(size_t)i * n + j]);
        low = fmin(low, Ain[(size_t)i * n + i] - r);
        spread = fmax(spread, Ain[(size_t)i * n + i] + r);
    }
    for (int o = 32; o > 0; o >>= 1) { low = fmin(low, __shfl_down(low, o, 64)); spread = fmax(spread, __shfl_down(spread, o, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = low; red[16 + (tid >> 6)] = spread; }
    __syncthreads();
    low = red[0]; spread = red[16];
    for (int k = 1; k < nt / 64; ++k) { low = fmin(low, red[k]); spread = fmax(spread, red[16 + k]); }
    __syncthreads();
    const double sigma = fmax(0.0, -low) + 0.05 * (spread - low) + 1e-3;
    for (int x = tid; x < m * ld; x += nt) {
        const int j = x / ld, i = x - j * ld;                     // column j, row i
        G[x] = (i < n && j < n) ? Ain[(size_t)i * n + j] + (i == j ? sigma : 0.0) : 0.0;
    }
    if (tid == 0) *flag = 0;
    __syncthreads();

    const int team = tid / QC_EIG1_TEAM, tl = tid % QC_EIG1_TEAM;
    bool conv = false;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int step = 0; step < m - 1; ++step) {
            if (team < half) {
                int p, q;
                qc_rr_pair(step, team, m, p, q);
                double *gp = G + (size_t)p * ld, *gq = G + (size_t)q * ld;
                double a = 0.0, b = 0.0, c = 0.0, xp[QC_EIG1_ROWS], xq[QC_EIG1_ROWS];
#pragma unroll
                for (int k = 0; k < QC_EIG1_ROWS; ++k) {
                    const int r = tl + k * QC_EIG1_TEAM;
                    xp[k] = (r < n) ? gp[r] : 0.0; xq[k] = (r < n) ? gq[r] : 0.0;
                    a = fma(xp[k], xp[k], a); b = fma(xq[k], xq[k], b); c = fma(xp[k], xq[k], c);
                }
#pragma unroll
                for (int o = QC_EIG1_TEAM / 2; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, QC_EIG1_TEAM); b += __shfl_xor(b, o, QC_EIG1_TEAM); c += __shfl_xor(c, o, QC_EIG1_TEAM);
                }
                if (c * c > 1e-32 * (a * b)) {                     // team-uniform: |c| / sqrt(a b) > 1e-16
                    double cs, sn;
                    qc_hestenes_rotation(a, b, c, cs, sn);
#pragma unroll
                    for (int k = 0; k < QC_EIG1_ROWS; ++k) {
                        const int r = tl + k * QC_EIG1_TEAM;
                        if (r < n) { gp[r] = cs * xp[k] - sn * xq[k]; gq[r] = sn * xp[k] + cs * xq[k]; }
                    }
                    // quadratic convergence: pairs whose relative coupling |c| / sqrt(a b) is below done_tol (1e-9) are done after this rotation
                    if (tl == 0 && c * c > done_tol * done_tol * (a * b)) *flag = 1;
                }
            }
            __syncthreads();
        }
        const int any = *flag;
        __syncthreads();
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (!any) { conv = true; break; }
    }
    if (!conv && notconv && tid == 0) *notconv = 1;
    // eigenvalues = column norms - sigma; ascending rank sort; normalised, permuted columns out
    for (int j = tid; j < n; j += nt) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = fma(G[(size_t)j * ld + i], G[(size_t)j * ld + i], s2);
        nrm[j] = sqrt(s2);
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const double wi = nrm[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (nrm[j] < wi || (nrm[j] == wi && j < i)) ? 1 : 0;
        rank[i] = r;
        w[r] = wi - sigma;
    }
    __syncthreads();
    for (int x = tid; x < n * n; x += nt) {
        const int i = x / n, j = x - i * n;
        Vout[(size_t)i * n + rank[j]] = G[(size_t)j * ld + i] / nrm[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-sided Jacobi for n > 128: same algorithm as qc_jacobi1_kernel with G in global memory (d_work, n columns of
// stride n) and the pairs of a step spread over the 64 teams of one workgroup in a loop.  Slow (global read-modify-write
// behind a workgroup barrier per step) but only the cold start and the rare fallback of the refinement use it; the SCF
// loop's eigensolves are GEMMs (qc_eig_device_refine).  Keeps every shipped basis (benzene/6-311++G**: n = 180) usable.
__global__ __launch_bounds__(QC_EIG_THREADS) void qc_jacobi1g_kernel(int n, const double *__restrict__ Ain, double *__restrict__ G,
                                                                     double *__restrict__ Vout, double *__restrict__ w, int max_sweeps, double done_tol,
                                                                     int *__restrict__ notconv) {
    extern __shared__ double sm[];
    __shared__ double red[32];
    __shared__ int flag;
    double *nrm = sm;                                       // n column norms
    int *rank = reinterpret_cast<int *>(sm + n);            // n ranks
    const int m = (n + 1) & ~1, half = m / 2;
    const int tid = threadIdx.x, nt = blockDim.x;
    double low = 1e300, spread = 0.0;
    for (int i = tid; i < n; i += nt) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) if (j != i) r += fabs(Ain[(size_t)i * n + j]);
        low = fmin(low, Ain[(size_t)i * n + i] - r);
        spread = fmax(spread, Ain[(size_t)i * n + i] + r);
    }
    for (int o = 32; o > 0; o >>= 1) { low = fmin(low, __shfl_down(low, o, 64)); spread = fmax(spread, __shfl_down(spread, o, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = low; red[16 + (tid >> 6)] = spread; }
    __syncthreads();
    low = red[0]; spread = red[16];
    for (int k = 1; k < nt / 64; ++k) { low = fmin(low, red[k]); spread = fmax(spread, red[16 + k]); }
    __syncthreads();
    const double sigma = fmax(0.0, -low) + 0.05 * (spread - low) + 1e-3;
    for (int x = tid; x < n * n; x += nt) {               // column-major copy of the (symmetric) shifted matrix
        const int j = x / n, i = x - j * n;
        G[x] = 0.5 * (Ain[(size_t)i * n + j] + Ain[(size_t)j * n + i]) + (i == j ? sigma : 0.0);
    }
    if (tid == 0) flag = 0;
    __syncthreads();
    const int team = tid / QC_EIG1_TEAM, tl = tid % QC_EIG1_TEAM, nteams = nt / QC_EIG1_TEAM;
    bool conv = false;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int step = 0; step < m - 1; ++step) {
            for (int pr = team; pr < half; pr += nteams) {
                int p, q;
                qc_rr_pair(step, pr, m, p, q);
                if (p >= n || q >= n) continue;           // padding index of an odd n
                double *gp = G + (size_t)p * n, *gq = G + (size_t)q * n;
                double a = 0.0, b = 0.0, c = 0.0;
                for (int r = tl; r < n; r += QC_EIG1_TEAM) { const double xp = gp[r], xq = gq[r]; a = fma(xp, xp, a); b = fma(xq, xq, b); c = fma(xp, xq, c); }
#pragma unroll
                for (int o = QC_EIG1_TEAM / 2; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, QC_EIG1_TEAM); b += __shfl_xor(b, o, QC_EIG1_TEAM); c += __shfl_xor(c, o, QC_EIG1_TEAM);
                }
                const double rel = fabs(c) / sqrt(a * b);
                if (rel > 1e-16) {
                    const double zeta = (b - a) / (2.0 * c);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
                    const double cs = 1.0 / sqrt(fma(t, t, 1.0)), sn = cs * t;
                    for (int r = tl; r < n; r += QC_EIG1_TEAM) { const double xp = gp[r], xq = gq[r]; gp[r] = cs * xp - sn * xq; gq[r] = sn * xp + cs * xq; }
                    if (tl == 0 && rel > done_tol) flag = 1;
                }
            }
            __syncthreads();
        }
        const int any = flag;
        __syncthreads();
        if (tid == 0) flag = 0;
        __syncthreads();
        if (!any) { conv = true; break; }
    }
    if (!conv && notconv && tid == 0) *notconv = 1;
    for (int j = tid; j < n; j += nt) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = fma(G[(size_t)j * n + i], G[(size_t)j * n + i], s2);
        nrm[j] = sqrt(s2);
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const double wi = nrm[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (nrm[j] < wi || (nrm[j] == wi && j < i)) ? 1 : 0;
        rank[i] = r;
        w[r] = wi - sigma;
    }
    __syncthreads();
    for (int x = tid; x < n * n; x += nt) {
        const int i = x / n, j = x - i * n;
        Vout[(size_t)i * n + rank[j]] = G[(size_t)j * n + i] / nrm[j];
    }
}

// dA: input (left intact), dV: sorted eigenvectors, dw: eigenvalues, d_work: n*n scratch
int qc_eig_device(hipStream_t st, int n, double *dA, double *dV, double *dw, double *d_work, int max_sweeps, double done_tol, int *notconv) {
    const int m = (n + 1) & ~1, ld = m | 1;
    const size_t tail = (2 * (m / 2) + 32) * sizeof(double) + (size_t)m * sizeof(int) + 16;
    const size_t lds2 = 2 * (size_t)m * ld * sizeof(double) + tail, lds1 = (size_t)m * ld * sizeof(double) + tail;
    const bool v_in_lds = lds2 <= 160 * 1024;
    static const bool force1 = getenv("QC_EIG_ONESIDED") != nullptr;
    if ((!v_in_lds || force1) && n <= QC_EIG1_ROWS * QC_EIG1_TEAM) {   // 98 < n <= 128: one-sided variant, a single matrix in LDS
        const size_t l1 = ((size_t)m * (n | 1) + 32 + m) * sizeof(double) + (size_t)(m + 4) * sizeof(int) + 16;
        if (l1 <= 160 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qc_jacobi1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
            if (e != hipSuccess) return QC_ERR_HIP;
            hipLaunchKernelGGL(qc_jacobi1_kernel, dim3(1), dim3(QC_EIG1_THREADS), l1, st, n, dA, dV, dw, max_sweeps, done_tol, notconv);
            return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
        }
    }
    const size_t lds = v_in_lds ? lds2 : lds1;
    if (lds > 160 * 1024 || !v_in_lds) {                // n > 128: the global-memory variant
        if (n > 3000) return QC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(qc_jacobi1g_kernel, dim3(1), dim3(QC_EIG_THREADS), (size_t)n * 12 + 16, st, n, dA, d_work, dV, dw, max_sweeps, done_tol, notconv);
        return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
    }
    auto kern = v_in_lds ? qc_jacobi_kernel<true> : qc_jacobi_kernel<false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return QC_ERR_HIP;
    }
    static const int nthreads = getenv("QC_EIG_THREADS") ? atoi(getenv("QC_EIG_THREADS")) : QC_EIG_THREADS;
    hipLaunchKernelGGL(kern, dim3(1), dim3(nthreads), lds, st, n, dA, d_work, dV, dw, max_sweeps, done_tol, notconv);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

// Warm-started variant for the SCF loop: with V0 the eigenvectors of the previous iteration's matrix,
// B = V0^T A V0 is nearly diagonal, Jacobi needs 1-3 sweeps instead of ~8, and V = V0 Q.  The three products are
// f64 MFMA GEMMs.  t1/t2: n*n scratch each.
int qc_eig_device_warm(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                       int max_sweeps, double done_tol, int *notconv) {
    qc_gemm(st, n, n, n, 1.0, dA, n, false, dV0, n, false, 0.0, t1, n);        // A V0
    qc_gemm(st, n, n, n, 1.0, dV0, n, true, t1, n, false, 0.0, t2, n);         // V0^T (A V0)
    int rc = qc_eig_device(st, n, t2, t1, dw, d_work, max_sweeps, done_tol, notconv);   // Q -> t1
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, dV0, n, false, t1, n, false, 0.0, dV, n);        // V = V0 Q
    return QC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Eigenvector refinement (Ogita & Aishima, Japan J. Indust. Appl. Math. 35 (2018) 1007): with X an approximate
// eigenvector matrix of symmetric A,  R = I - X^T X,  S = X^T A X,  lam_i = S_ii / (1 - R_ii),
//   E_ij = (S_ij + lam_j R_ij) / (lam_j - lam_i)   (|lam_i - lam_j| > delta),   E_ij = R_ij / 2   (otherwise, and i = j),
//   X <- X (I + E)   converges quadratically.  Everything O(n^3) is an f64 MFMA GEMM - this is where the matrix cores
// earn their keep in the SCF loop, whose Fock matrix moves little between iterations once DIIS has kicked in.
// Pair (i,j) classes, with  a_ij = S_ij + (lam_i + lam_j)/2 R_ij  (coupling),  g = |lam_j - lam_i|,  scale = max|lam|:
//   negligible : |a_ij| <= 1e-13 scale                       -> E_ij = R_ij / 2        (orthogonality only)
//   strong     : |a_ij| / max(g, 1e-8 scale) > 1e-3          -> exact 2x2 Jacobi rotation, provided every index has at
//                most one strong partner (near-degenerate pairs: their eigenvectors turn by finite angles under tiny
//                changes of A; the first-order formula would need many passes); otherwise the caller goes to Jacobi
//   degenerate : g <= 1e-8 scale, weak coupling              -> E_ij = R_ij / 2, coupling recorded in stats[4]
//   regular    :                                             -> E_ij = (S_ij + lam_j R_ij) / (lam_j - lam_i)
// stats: [0] ||offdiag(S)||_F  [1] ||R||_F  [2] scale  [3] #strong pairs  [4] max coupling left in degenerate pairs
//        [5] max |a_ij| / g over regular pairs (size of the first-order update)  [6] 1 if some index has > 1 strong partner
constexpr double QC_REF_TAU = 1e-3, QC_REF_TINY = 1e-13, QC_REF_GFLOOR = 1e-8;

__device__ __forceinline__ double qc_refine_m(int n, int x, const double *__restrict__ S, const double *__restrict__ XtX,
                                              const double *__restrict__ lam, double tiny, double gfloor, const int *__restrict__ partner);

// `M` (sync-free variant only): the update matrix I + E is written by this kernel as well, saving a launch per pass
__global__ __launch_bounds__(1024) void qc_refine_stats_kernel(int n, const double *__restrict__ S, const double *__restrict__ XtX,
                                                                double *__restrict__ lam, double *__restrict__ stats, int *__restrict__ partner,
                                                                int *__restrict__ ctl, double *__restrict__ M) {
    if (ctl && ctl[0] != 0) return;
    __shared__ double red[4 * 16];
    __shared__ double sh_scale;
    __shared__ int sh_multi, sh_nstrong;
    const int tid = threadIdx.x;
    double off = 0.0, rr = 0.0, amax = 0.0;
    for (int x = tid; x < n * n; x += 1024) {
        const int i = x / n, j = x - i * n;
        const double r = (i == j ? 1.0 : 0.0) - XtX[x];
        rr = fma(r, r, rr);
        if (i != j) off = fma(S[x], S[x], off);
        else { const double l = S[x] / (1.0 - r); lam[i] = l; amax = fmax(amax, fabs(l)); }
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); rr += __shfl_down(rr, o, 64); amax = fmax(amax, __shfl_down(amax, o, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = off; red[16 + (tid >> 6)] = rr; red[32 + (tid >> 6)] = amax; }
    if (tid == 0) { sh_multi = 0; sh_nstrong = 0; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < 16; ++k) { a += red[k]; b += red[16 + k]; c = fmax(c, red[32 + k]); }
        stats[0] = sqrt(a); stats[1] = sqrt(b); stats[2] = c;
        sh_scale = c;
    }
    __syncthreads();                                         // lam[] (global, written by this workgroup) is visible
    const double scale = sh_scale, tiny = QC_REF_TINY * scale, gfloor = QC_REF_GFLOOR * scale;
    double cmax = 0.0, emax = 0.0;
    for (int i = tid >> 4; i < n; i += 64) {                 // one row per team of 16 lanes
        int cnt = 0, who = -1;
        for (int j = tid & 15; j < n; j += 16) {
            if (j == i) continue;
            const double r = -XtX[(size_t)i * n + j];
            const double sij = 0.5 * (S[(size_t)i * n + j] + S[(size_t)j * n + i]);   // A is symmetric only to rounding
            const double a = fabs(sij + 0.5 * (lam[i] + lam[j]) * r), g = fabs(lam[j] - lam[i]);
            if (a <= tiny) continue;
            if (a > QC_REF_TAU * fmax(g, gfloor)) { ++cnt; who = j; }
            else if (g <= gfloor) cmax = fmax(cmax, a);
            else emax = fmax(emax, a / g);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 16); who = max(who, __shfl_xor(who, o, 16)); }
        if ((tid & 15) == 0) {
            partner[i] = cnt == 0 ? -1 : (cnt == 1 ? who : -2);
            if (cnt > 1) sh_multi = 1;
            if (cnt == 1) atomicAdd(&sh_nstrong, 1);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { cmax = fmax(cmax, __shfl_down(cmax, o, 64)); emax = fmax(emax, __shfl_down(emax, o, 64)); }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = cmax; red[16 + (tid >> 6)] = emax; }
    // a strong pair must be mutual (i's only strong partner is j and vice versa); one index per thread (partner[] was written by
    // this workgroup before the barrier above) - as a loop of thread 0 these dependent loads were half of the kernel's 18 us
    for (int i = tid; i < n; i += 1024) { const int pj = partner[i]; if (pj >= 0 && partner[pj] != i) sh_multi = 1; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 16; ++k) { a = fmax(a, red[k]); b = fmax(b, red[16 + k]); }
        const int multi = sh_multi;
        stats[3] = 0.5 * sh_nstrong; stats[4] = a; stats[5] = b; stats[6] = multi;
        if (ctl) {   // the decisions qc_eig_device_refine takes on the host, for the sync-free variant
            const double orth = stats[1], scl = fmax(stats[2], 1e-300);
            if (!(b <= 0.1) || !(orth <= 1e-3) || multi) ctl[0] = 2;             // not perturbative: rotations needed
            else {
                ctl[1] = (b <= 1e-7 && orth <= 1e-7 && sh_nstrong == 0) ? 1 : 0;   // one more update finishes
                ctl[2] = (a <= 1e-12 * scl) ? 1 : 0;                               // no coupling left inside degenerate pairs
            }
        }
    }
    if (M) {
        __syncthreads();                                     // partner[], lam[], the decision: written by this workgroup
        if (ctl[0] != 0) return;
        const double scl = sh_scale;
        for (int x = tid; x < n * n; x += 1024) M[x] = qc_refine_m(n, x, S, XtX, lam, QC_REF_TINY * scl, QC_REF_GFLOOR * scl, partner);
    }
}

// The statistics / decisions / update matrix of a refinement pass spread over the chip (round 3).  As one workgroup
// (qc_refine_stats_kernel above, kept for the synchronous variant) this took 28 us per pass at n = 114 - 74 us of benzene's 340 us Roothaan
// step: ~2000 f64-heavy instructions per wave for sixteen waves on ONE compute unit.  Now two launches of ceil(n / 8) workgroups of 256
// threads, eight rows each (32 lanes per row, columns l + 32 q):
//   A  every workgroup forms all n Rayleigh quotients itself (n loads: cheaper than a grid barrier), classifies the pairs of its rows,
//      fixes their partner[] entries and leaves its partial sums / maxima / flags in `part` (8 doubles per workgroup);
//   B  every workgroup folds the partials in a fixed order and takes the decisions itself (workgroup 0 publishes them), then writes
//      its rows of M = I + E.
// The size of the first-order update is tested as |a| <= tau g (no division; stats[5] reports the threshold exceeded).
constexpr int QC_STATS_ROWS = 8;
__global__ __launch_bounds__(256) void qc_refine_statsA_kernel(int n, const double *__restrict__ S, const double *__restrict__ XtX,
                                                                double *__restrict__ lam, int *__restrict__ partner, double *__restrict__ part,
                                                                const int *__restrict__ ctl) {
    if (ctl && ctl[0] != 0) return;
    extern __shared__ double sh_lam[];                       // n
    __shared__ double red[3][4];
    __shared__ double sh_scale;
    __shared__ int sh_multi, sh_nstrong, sh_big, sh_notlast;
    const int tid = threadIdx.x, r = tid >> 5, l = tid & 31;
    double amax = 0.0;
    for (int t = tid; t < n; t += 256) {
        const double rr_ = 1.0 - XtX[(size_t)t * n + t], lv = S[(size_t)t * n + t] / (1.0 - rr_);
        sh_lam[t] = lv; amax = fmax(amax, fabs(lv));
        if (blockIdx.x == 0) lam[t] = lv;
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_down(amax, o, 64));
    if ((tid & 63) == 0) red[0][tid >> 6] = amax;
    if (tid == 0) { sh_multi = 0; sh_nstrong = 0; sh_big = 0; sh_notlast = 0; }
    __syncthreads();
    if (tid == 0) sh_scale = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
    __syncthreads();
    const double scale = sh_scale, tiny = QC_REF_TINY * scale, gfloor = QC_REF_GFLOOR * scale;
    const int i = blockIdx.x * QC_STATS_ROWS + r;
    const bool iok = i < n;
    const double li = sh_lam[iok ? i : 0];
    double off = 0.0, rsum = 0.0, cmax = 0.0;
    int cnt = 0, who = -1, big = 0, notlast = 0;
    for (int j0 = l; j0 < n; j0 += 128) {                    // four columns of the row at a time
        double sij[4], sji[4], xij[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 32 * u;
            const bool ok = iok && j < n;
            sij[u] = S[ok ? (size_t)i * n + j : 0]; sji[u] = S[ok ? (size_t)j * n + i : 0]; xij[u] = XtX[ok ? (size_t)i * n + j : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 32 * u;
            if (iok && j < n) {
                const double rv = (i == j ? 1.0 : 0.0) - xij[u];
                rsum = fma(rv, rv, rsum);
                if (i != j) {
                    off = fma(sij[u], sij[u], off);
                    const double lj = sh_lam[j];
                    const double av = fabs(0.5 * (sij[u] + sji[u]) + 0.5 * (li + lj) * rv), g = fabs(lj - li);
                    if (av > tiny) {
                        if (av > QC_REF_TAU * fmax(g, gfloor)) { ++cnt; who = j; }
                        else if (g <= gfloor) cmax = fmax(cmax, av);
                        else { big |= !(av <= 0.1 * g) ? 1 : 0; notlast |= !(av <= 1e-7 * g) ? 1 : 0; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 32); who = max(who, __shfl_xor(who, o, 32)); }
    if (l == 0 && iok) {
        partner[i] = cnt == 0 ? -1 : (cnt == 1 ? who : -2);
        if (cnt > 1) sh_multi = 1;
        if (cnt == 1) atomicAdd(&sh_nstrong, 1);
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); rsum += __shfl_down(rsum, o, 64); cmax = fmax(cmax, __shfl_down(cmax, o, 64)); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = off; red[1][tid >> 6] = rsum; red[2][tid >> 6] = cmax; }
    if (big) sh_big = 1;
    if (notlast) sh_notlast = 1;
    __syncthreads();
    if (tid == 0) {
        double *p = part + 8 * blockIdx.x;
        p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        p[2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
        p[3] = sh_nstrong; p[4] = sh_multi; p[5] = sh_big; p[6] = sh_notlast; p[7] = scale;
    }
}
__global__ __launch_bounds__(256) void qc_refine_statsB_kernel(int n, const double *__restrict__ S, const double *__restrict__ XtX,
                                                                const double *__restrict__ lam, double *__restrict__ stats,
                                                                const int *__restrict__ partner, const double *__restrict__ part, int nwg,
                                                                int *__restrict__ ctl, double *__restrict__ M) {
    if (ctl[0] != 0) return;
    extern __shared__ double shb[];                          // lam[n], then partner[n] as ints
    double *sh_lam = shb;
    int *sh_partner = reinterpret_cast<int *>(shb + n);
    __shared__ int sh_go, sh_multi;
    __shared__ double sh_scale;
    const int tid = threadIdx.x, r = tid >> 5, l = tid & 31;
    for (int t = tid; t < n; t += 256) { sh_lam[t] = lam[t]; sh_partner[t] = partner[t]; }
    if (tid == 0) sh_multi = 0;
    __syncthreads();
    for (int t = tid; t < n; t += 256) { const int pj = sh_partner[t]; if (pj >= 0 && sh_partner[pj] != t) sh_multi = 1; }    // a strong pair must be mutual
    __syncthreads();
    if (tid == 0) {
        double off = 0.0, rs = 0.0, cm = 0.0, nstrong = 0.0;
        int multi = sh_multi, big = 0, notlast = 0;
        for (int w = 0; w < nwg; ++w) {
            const double *p = part + 8 * w;
            off += p[0]; rs += p[1]; cm = fmax(cm, p[2]); nstrong += p[3];
            multi |= p[4] != 0.0; big |= p[5] != 0.0; notlast |= p[6] != 0.0;
        }
        const double scale = part[7], orth = sqrt(rs), scl = fmax(scale, 1e-300);
        sh_scale = scale;
        int c0 = 0, c1 = 0, c2 = 0;
        if (big || !(orth <= 1e-3) || multi) c0 = 2;                          // not perturbative: rotations needed
        else {
            c1 = (!notlast && orth <= 1e-7 && nstrong == 0.0) ? 1 : 0;        // one more update finishes
            c2 = (cm <= 1e-12 * scl) ? 1 : 0;                                 // no coupling left inside degenerate pairs
        }
        sh_go = c0 == 0;
        if (blockIdx.x == 0) {
            stats[0] = sqrt(off); stats[1] = orth; stats[2] = scale; stats[3] = 0.5 * nstrong; stats[4] = cm;
            stats[5] = big ? 1.0 : (notlast ? 1e-6 : 0.0); stats[6] = multi;
            if (c0) ctl[0] = c0; else { ctl[1] = c1; ctl[2] = c2; }
        }
    }
    __syncthreads();
    if (!sh_go) return;
    const double scale = sh_scale, tiny = QC_REF_TINY * scale, gfloor = QC_REF_GFLOOR * scale;
    const int i = blockIdx.x * QC_STATS_ROWS + r;
    if (i >= n) return;
    const double li = sh_lam[i];
    const int pi_ = sh_partner[i];
    for (int j0 = l; j0 < n; j0 += 128) {
        double sij[4], sji[4], xij[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 32 * u;
            const bool ok = j < n;
            sij[u] = S[ok ? (size_t)i * n + j : 0]; sji[u] = S[ok ? (size_t)j * n + i : 0]; xij[u] = XtX[ok ? (size_t)i * n + j : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 32 * u;
            if (j < n) {
                const double lj = sh_lam[j];
                const double rv = (i == j ? 1.0 : 0.0) - xij[u];
                const double sy = 0.5 * (sij[u] + sji[u]);
                const double av = sy + 0.5 * (li + lj) * rv, g = lj - li;
                double mm;
                if (i == j) {
                    mm = 1.0 + 0.5 * rv;
                    if (pi_ >= 0) {
                        const double avp = 0.5 * (S[(size_t)i * n + pi_] + S[(size_t)pi_ * n + i]) - 0.5 * (li + sh_lam[pi_]) * XtX[(size_t)i * n + pi_];
                        const double theta = (sh_lam[pi_] - li) / (2.0 * avp);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
                        mm = 1.0 / sqrt(fma(t, t, 1.0)) + 0.5 * rv;
                    }
                } else if (pi_ == j) {
                    const double theta = g / (2.0 * av);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
                    mm = t / sqrt(fma(t, t, 1.0));
                } else if (fabs(av) <= tiny || fabs(g) <= gfloor) mm = 0.5 * rv;
                else mm = (sy + lj * rv) / g;
                M[(size_t)i * n + j] = mm;
            }
        }
    }
}
// doubles of the scratch behind lam / stats / partner that the two kernels need
size_t qc_refine_part_doubles(int n) { return 8 * (size_t)((n + QC_STATS_ROWS - 1) / QC_STATS_ROWS); }

// element x of M = I + E, with exact rotations on the strong pairs
__device__ __forceinline__ double qc_refine_m(int n, int x, const double *__restrict__ S, const double *__restrict__ XtX,
                                              const double *__restrict__ lam, double tiny, double gfloor, const int *__restrict__ partner) {
    const int i = x / n, j = x - i * n;
    const double r = (i == j ? 1.0 : 0.0) - XtX[x];
    double m;
    if (i == j) {
        m = 1.0 + 0.5 * r;
        const int pj = partner[i];
        if (pj >= 0) {                                   // cosine of this index's rotation
            const double a = 0.5 * (S[(size_t)i * n + pj] + S[(size_t)pj * n + i]) - 0.5 * (lam[i] + lam[pj]) * XtX[(size_t)i * n + pj];
            const double theta = (lam[pj] - lam[i]) / (2.0 * a);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
            m = 1.0 / sqrt(fma(t, t, 1.0)) + 0.5 * r;
        }
    } else {
        const double sij = 0.5 * (S[x] + S[(size_t)j * n + i]);      // A is symmetric only to rounding: use the symmetric part
        const double a = sij + 0.5 * (lam[i] + lam[j]) * r, g = lam[j] - lam[i];
        if (partner[i] == j) {                           // sine: x_j' = c x_j + s x_i
            const double theta = g / (2.0 * a);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
            m = t / sqrt(fma(t, t, 1.0));
        } else if (fabs(a) <= tiny || fabs(g) <= gfloor) m = 0.5 * r;
        else m = (sij + lam[j] * r) / g;
    }
    return m;
}

__global__ void qc_refine_update_kernel(int n, const double *__restrict__ S, const double *__restrict__ XtX, const double *__restrict__ lam,
                                        const double *__restrict__ stats, const int *__restrict__ partner, double *__restrict__ M,
                                        const int *__restrict__ ctl) {
    if (ctl && ctl[0] != 0) return;
    const double scale = stats[2], tiny = QC_REF_TINY * scale, gfloor = QC_REF_GFLOOR * scale;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * n; x += gridDim.x * blockDim.x) M[x] = qc_refine_m(n, x, S, XtX, lam, tiny, gfloor, partner);
}

// ascending eigenvalues + columns permuted alongside (utils.rs:28)
__global__ __launch_bounds__(1024) void qc_sort_columns_kernel(int n, const double *__restrict__ lam, const double *__restrict__ X,
                                                                double *__restrict__ w, double *__restrict__ Xs) {
    extern __shared__ int rank_s[];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) {
        const double wi = lam[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (lam[j] < wi || (lam[j] == wi && j < i)) ? 1 : 0;
        rank_s[i] = r;
        w[r] = wi;
    }
    __syncthreads();
    for (int x = tid; x < n * n; x += 1024) {
        const int i = x / n, j = x - i * n;
        Xs[(size_t)i * n + rank_s[j]] = X[x];
    }
}

// Eigen-decomposition of dA starting from the eigenvectors dV0 of a nearby matrix.
//   pass: S = X^T A X, X^T X, stats -> host (one small sync).
//     * update size max|E| > 0.1 (not perturbative), orthogonality lost, or passes exhausted -> Jacobi kernel on S
//       (nearly diagonal => few sweeps), V = X Q;
//     * max|E| <= 1e-7: one more update leaves ~1e-14 (quadratic) - unless a cluster still carries coupling, which only
//       rotations can remove -> Jacobi on S;
//     * otherwise apply X <- X (I + E) and take another pass.
// Scratch: t1..t4, d_work (n*n each), small (n + 8 doubles).
int qc_eig_device_refine(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                         double *t3, double *t4, double *small, int *notconv) {
    const size_t nn = (size_t)n * n;
    double *lam = small, *stats = small + n;
    double *X = t4;                                            // current eigenvector estimate
    if (hipMemcpyAsync(X, dV0, nn * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return QC_ERR_HIP;
    double hs[8];
    int *partner = reinterpret_cast<int *>(small + n + 8);
    if (const char *dump = getenv("QC_EIG_DUMP")) {           // debugging aid: append (n, A, V0) of every call to a file
        static int ncall = 0;
        std::vector<double> h(2 * nn);
        (void)hipMemcpy(h.data(), dA, nn * sizeof(double), hipMemcpyDeviceToHost);
        (void)hipMemcpy(h.data() + nn, dV0, nn * sizeof(double), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(dump, ncall++ ? "ab" : "wb")) { double dn = n; fwrite(&dn, 8, 1, f); fwrite(h.data(), 8, 2 * nn, f); fclose(f); }
    }
    constexpr int MAXPASS = 5;
    static const bool dbg = getenv("QC_EIG_DEBUG") != nullptr;
    for (int pass = 0; pass < MAXPASS; ++pass) {
        qc_gemm(st, n, n, n, 1.0, dA, n, false, X, n, false, 0.0, t1, n);          // A X
        qc_gemm(st, n, n, n, 1.0, X, n, true, t1, n, false, 0.0, t2, n);           // S = X^T A X
        qc_gemm(st, n, n, n, 1.0, X, n, true, X, n, false, 0.0, t3, n);            // X^T X
        hipLaunchKernelGGL(qc_refine_stats_kernel, dim3(1), dim3(1024), 0, st, n, t2, t3, lam, stats, partner, (int *)nullptr, (double *)nullptr);
        if (hipMemcpyAsync(hs, stats, 7 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return QC_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return QC_ERR_HIP;
        const double scale = fmax(hs[2], 1e-300), orth = hs[1], nstrong = hs[3], cmax = hs[4], emax = hs[5], multi = hs[6];
        if (dbg) fprintf(stderr, "[eig n=%d pass %d] off %.2e orth %.2e scale %.1f strong %.0f multi %.0f cmax %.2e emax %.2e\n", n, pass, hs[0], orth, scale, nstrong, multi, cmax, emax);
        if (!(emax <= 0.1) || !(orth <= 1e-3) || multi != 0.0 || pass == MAXPASS - 1) {
            // not perturbative (or not converging): rotations.  Always from the orthonormal start V0.
            if (pass > 0) {
                qc_gemm(st, n, n, n, 1.0, dA, n, false, dV0, n, false, 0.0, t1, n);
                qc_gemm(st, n, n, n, 1.0, dV0, n, true, t1, n, false, 0.0, t2, n);
            }
            int rc = qc_eig_device(st, n, t2, t1, dw, d_work, 40, 1e-9, notconv);      // Q -> t1 (sorted), eigenvalues -> dw
            if (rc != QC_OK) return rc;
            qc_gemm(st, n, n, n, 1.0, dV0, n, false, t1, n, false, 0.0, dV, n);    // V = V0 Q
            return QC_OK;
        }
        const bool last = emax <= 1e-7 && orth <= 1e-7 && nstrong == 0.0;
        hipLaunchKernelGGL(qc_refine_update_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, n, t2, t3, lam, stats, partner, t1, (const int *)nullptr);
        qc_gemm(st, n, n, n, 1.0, X, n, false, t1, n, false, 0.0, d_work, n);      // X (I + E): error now ~ emax^2
        if (last && cmax <= 1e-12 * scale) {   // lam is second-order accurate already; d_work holds the final vectors
            hipLaunchKernelGGL(qc_sort_columns_kernel, dim3(1), dim3(1024), n * sizeof(int), st, n, lam, d_work, dw, dV);
            return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
        }
        if (hipMemcpyAsync(X, d_work, nn * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return QC_ERR_HIP;
        if (last) {
            // converged except for coupling left inside degenerate pairs, which only rotations remove:
            // X is orthonormal to ~1e-14 now, so Jacobi on X^T A X (a handful of non-trivial rotations) finishes the job
            qc_gemm(st, n, n, n, 1.0, dA, n, false, X, n, false, 0.0, t1, n);
            qc_gemm(st, n, n, n, 1.0, X, n, true, t1, n, false, 0.0, t2, n);
            int rc = qc_eig_device(st, n, t2, t1, dw, d_work, 40, 1e-9, notconv);
            if (rc != QC_OK) return rc;
            qc_gemm(st, n, n, n, 1.0, X, n, false, t1, n, false, 0.0, dV, n);
            return QC_OK;
        }
    }
    return QC_ERR_HIP;   // not reached
}

// Sync-free variant for the SCF loop: a fixed number of passes is enqueued, every kernel looks at the device-side control
// word ctl[0] (0 running, 1 done, 2 rotations needed) and returns at once when the refinement has ended.  The caller
// reads ctl[0] together with the iteration's energy; on 2 it repeats the eigensolve with qc_eig_device_refine.
__global__ __launch_bounds__(1024) void qc_refine_finish_kernel(int n, const double *__restrict__ lam, const double *__restrict__ Xn,
                                                                 double *__restrict__ X, double *__restrict__ w, double *__restrict__ Xs,
                                                                 int *__restrict__ ctl, int final_pass) {
    if (ctl[0] != 0) return;
    extern __shared__ int rank_s[];
    const int tid = threadIdx.x;
    const int last = ctl[1], clean = ctl[2];
    __syncthreads();
    if (tid == 0) ctl[3] += 1;                              // passes used (the host sizes the next step's pipeline with it)
    if (last && clean) {                                    // Xn holds the final vectors: ascending eigenvalues, columns alongside
        double *lam_s = reinterpret_cast<double *>(rank_s + ((n + 1) & ~1));      // (the n quotients once, not n times per thread, from L2)
        for (int i = tid; i < n; i += 1024) lam_s[i] = lam[i];
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const double wi = lam_s[i];
            int r = 0;
            for (int j = 0; j < n; ++j) r += (lam_s[j] < wi || (lam_s[j] == wi && j < i)) ? 1 : 0;
            rank_s[i] = r;
            w[r] = wi;
        }
        __syncthreads();
        // thread (row group, column): eight elements requested at a time, no index division
        const int c = tid & 127, rg = tid >> 7;
        for (int c0 = 0; c0 < n; c0 += 128) {
            const int j = c0 + c;
            const int rj = j < n ? rank_s[j] : 0;
            for (int i0 = rg; i0 < n; i0 += 64) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + 8 * u; v[u] = Xn[(j < n && i < n) ? (size_t)i * n + j : 0]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + 8 * u; if (j < n && i < n) Xs[(size_t)i * n + rj] = v[u]; }
            }
        }
        __syncthreads();
        if (tid == 0) ctl[0] = 1;
    } else if (last || final_pass) {
        __syncthreads();
        if (tid == 0) ctl[0] = 2;                           // coupling inside a degenerate cluster, or passes exhausted
    }
    // (otherwise the next pass reads Xn in place: the passes alternate between two buffers, no copy)
}

int qc_eig_refine_async(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                        double *t3, double *t4, double *small, int *ctl, int npass) {
    double *lam = small, *stats = small + n;
    int *partner = reinterpret_cast<int *>(small + n + 8);
    // ctl[0..3] must be zero on entry (the SCF step clears all control words with one memset)
    const double *X = dV0;                                    // pass 0 reads the start vectors in place
    for (int pass = 0; pass < npass; ++pass) {
        qc_gemm_pair(st, n, dA, false, X, t1, X, true, X, t3, ctl);                     // A X  and  X^T X in one launch
        qc_gemm(st, n, n, n, 1.0, X, n, true, t1, n, false, 0.0, t2, n, ctl);           // S = X^T A X
        {   // statistics, decisions, M = I + E -> t1
            static const bool one_wg = getenv("QC_STATS_ONE_WG") != nullptr;      // (A/B switch: the single-workgroup kernel)
            if (one_wg) hipLaunchKernelGGL(qc_refine_stats_kernel, dim3(1), dim3(1024), 0, st, n, t2, t3, lam, stats, partner, ctl, t1);
            else {
                const int nwg = (n + QC_STATS_ROWS - 1) / QC_STATS_ROWS;
                double *part = small + 2 * n + 16;            // (behind lam[n], stats[8], partner[n]: callers allocate qc_eig_small_doubles(n))
                hipLaunchKernelGGL(qc_refine_statsA_kernel, dim3(nwg), dim3(256), n * sizeof(double), st, n, t2, t3, lam, partner, part, ctl);
                hipLaunchKernelGGL(qc_refine_statsB_kernel, dim3(nwg), dim3(256), n * sizeof(double) + n * sizeof(int) + 8, st, n, t2, t3, lam, stats, partner,
                                   part, nwg, ctl, t1);
            }
        }
        double *Xn = (pass & 1) ? t4 : d_work;                // (passes alternate between d_work and t4; pass 0 reads the start vectors in place)
        qc_gemm(st, n, n, n, 1.0, X, n, false, t1, n, false, 0.0, Xn, n, ctl);          // X (I + E)
        hipLaunchKernelGGL(qc_refine_finish_kernel, dim3(1), dim3(1024), ((n + 1) & ~1) * sizeof(int) + n * sizeof(double), st, n, lam, Xn, (double *)nullptr, dw, dV, ctl, pass == npass - 1 ? 1 : 0);
        X = Xn;
    }
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

int qc_eig_cold_async(hipStream_t st, int n, double *dA, double *dX0, double *triwork, double *dV, double *dw, double *d_work, double *t1, double *t2,
                      double *t3, double *t4, double *small, int *ctl, int npass) {
    int rc = qc_eig_tridiag_start(st, n, dA, dX0, triwork);
    if (rc != QC_OK) return rc;
    return qc_eig_refine_async(st, n, dA, dX0, dV, dw, d_work, t1, t2, t3, t4, small, ctl, npass);
}

int qc_eig_cold_sync(hipStream_t st, int n, double *dA, double *dX0, double *triwork, double *dV, double *dw, double *d_work, double *t1, double *t2,
                     double *t3, double *t4, double *small, int *ctl, int *notconv) {
    static const bool force_jacobi = getenv("QC_EIG_JACOBI") != nullptr;      // A/B switch: the single-workgroup Jacobi kernels only
    if (!qc_tri_ok(n) || force_jacobi) return qc_eig_device(st, n, dA, dV, dw, d_work, 40, 1e-9, notconv);
    if (hipMemsetAsync(ctl, 0, 4 * sizeof(int), st) != hipSuccess) return QC_ERR_HIP;
    int rc = qc_eig_cold_async(st, n, dA, dX0, triwork, dV, dw, d_work, t1, t2, t3, t4, small, ctl, 4);
    if (rc != QC_OK) return rc;
    int h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return QC_ERR_HIP;
    static const bool dbg = getenv("QC_EIG_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[eig cold n=%d] ctl %d %d %d passes %d\n", n, h[0], h[1], h[2], h[3]);
    if (h[0] == 1) return QC_OK;
    if (hipMemsetAsync(ctl, 0, 4 * sizeof(int), st) != hipSuccess) return QC_ERR_HIP;
    return qc_eig_device(st, n, dA, dV, dw, d_work, 40, 1e-9, notconv);      // rotations: the start was not good enough
}

// DIIS coefficients on the device (diis.rs:40-51): B is kept slot-indexed in HBM, the new row <e_0, e_j> arrives in
// `dots` (window order, newest first); Householder QR with the arithmetic of nalgebra's qr().solve(); one thread.
struct QcDiisArgs { int m, minlen, maxlen; int slot[12]; };
// the (m+1) x (m+1) system entirely in registers: every loop bound is a compile-time constant
template <int M>
__device__ __forceinline__ bool qc_qr_solve_reg(double (&A)[M * M], double (&y)[M], double (&x)[M]) {
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double norm = 0.0;
#pragma unroll
        for (int i = k; i < M; ++i) norm += A[i * M + k] * A[i * M + k];
        norm = sqrt(norm);
        if (norm == 0.0) return false;
        const double alpha = A[k * M + k] > 0 ? -norm : norm;
        double v[M];
        double vn = 0.0;
#pragma unroll
        for (int i = k; i < M; ++i) { v[i] = A[i * M + k] - (i == k ? alpha : 0.0); vn += v[i] * v[i]; }
        if (vn > 0.0) {
#pragma unroll
            for (int j = k; j < M; ++j) {
                double d = 0.0;
#pragma unroll
                for (int i = k; i < M; ++i) d += v[i] * A[i * M + j];
                d *= 2.0 / vn;
#pragma unroll
                for (int i = k; i < M; ++i) A[i * M + j] -= d * v[i];
            }
            double d = 0.0;
#pragma unroll
            for (int i = k; i < M; ++i) d += v[i] * y[i];
            d *= 2.0 / vn;
#pragma unroll
            for (int i = k; i < M; ++i) y[i] -= d * v[i];
        }
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) s -= A[i * M + j] * x[j];
        if (A[i * M + i] == 0.0) return false;
        x[i] = s / A[i * M + i];
    }
    return true;
}

template <int M>
__global__ void qc_diis_solve_kernel(QcDiisArgs a, const double *__restrict__ dots, double *B, double *__restrict__ c, int *__restrict__ flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    constexpr int m = M - 1;
    const int ML = a.maxlen;
#pragma unroll
    for (int j = 0; j < m; ++j) { const double d = dots[j]; B[a.slot[0] * ML + a.slot[j]] = d; B[a.slot[j] * ML + a.slot[0]] = d; }
    c[0] = 1.0;
#pragma unroll
    for (int j = 1; j < 12; ++j) c[j] = 0.0;
    if (m < a.minlen) return;                               // diis.rs:33-38: hand back the newest Fock matrix
    double A[M * M], y[M], x[M];
#pragma unroll
    for (int i = 0; i < m; ++i) {
#pragma unroll
        for (int j = 0; j < m; ++j) A[i * M + j] = B[a.slot[i] * ML + a.slot[j]];
        A[i * M + m] = 1.0; A[m * M + i] = 1.0; y[i] = 0.0;  // border +1, corner 0 (diis.rs:40-46)
    }
    A[m * M + m] = 0.0; y[m] = 1.0;
    if (!qc_qr_solve_reg<M>(A, y, x)) { *flag = 1; return; }   // "DIIS failed" (rhf.rs:73); c stays (1, 0, ...)
#pragma unroll
    for (int j = 0; j < m; ++j) c[j] = x[j];
}
void qc_diis_solve(hipStream_t st, int m, int minlen, int maxlen, const int *slots, const double *dots, double *B, double *c, int *flag) {
    QcDiisArgs a{};
    a.m = m; a.minlen = minlen; a.maxlen = maxlen;
    for (int j = 0; j < m; ++j) a.slot[j] = slots[j];
#define QC_DIIS_CASE(M) case M - 1: hipLaunchKernelGGL(qc_diis_solve_kernel<M>, dim3(1), dim3(64), 0, st, a, dots, B, c, flag); break;
    switch (m) { QC_DIIS_CASE(2) QC_DIIS_CASE(3) QC_DIIS_CASE(4) QC_DIIS_CASE(5) QC_DIIS_CASE(6) QC_DIIS_CASE(7) QC_DIIS_CASE(8) QC_DIIS_CASE(9)
                 QC_DIIS_CASE(10) QC_DIIS_CASE(11) QC_DIIS_CASE(12) default: break; }
#undef QC_DIIS_CASE
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void qc_axpby_kernel(int nn, double a, const double *x, double b, const double *y, double *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}
void qc_axpby(hipStream_t st, int n, double a, const double *x, double b, const double *y, double *out) {
    const int nn = n * n;
    hipLaunchKernelGGL(qc_axpby_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, nn, a, x, b, y, out);
}

__global__ void qc_sub_transpose_kernel(int n, const double *M, double *out) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * n; x += gridDim.x * blockDim.x) {
        const int i = x / n, j = x - i * n;
        out[x] = M[x] - M[j * n + i];
    }
}
void qc_sub_transpose(hipStream_t st, int n, const double *M, double *out) {
    hipLaunchKernelGGL(qc_sub_transpose_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, n, M, out);
}

// G = Gt + Gt^T: closes the unique-quartet digestion (fock_finalize)
// (with H and F given, F = H + G - the Fock matrix of rhf.rs:68 - leaves in the same launch)
// FX: Gt = [hi plane | lo plane] of 64-bit fixed-point integers, units fxs[1] = 2^-S and 2^-(S+32) (qc_fock_kernel.h): the two
// halves are added as integers, the planes meet in one fma when the sum becomes a double - G is symmetric bit for bit
template <bool FX>
__global__ void qc_symmetrize_add_kernel(int n, const double *Gt, size_t lo_off, double *G, const double *H, double *F, const double *fxs) {
    const double u1 = FX ? fxs[1] : 1.0, u2 = u1 * 0x1p-32;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * n; x += gridDim.x * blockDim.x) {
        const int i = x / n, j = x - i * n;
        double g;
        if constexpr (FX) {
            const long long *Gh = reinterpret_cast<const long long *>(Gt), *Gl = Gh + lo_off;
            g = fma((double)(Gl[x] + Gl[j * n + i]), u2, (double)(Gh[x] + Gh[j * n + i]) * u1);
        } else g = Gt[x] + Gt[j * n + i];
        G[x] = g;
        if (F) F[x] = 1.0 * H[x] + 1.0 * g;              // the arithmetic of qc_axpby(1, H, 1, G)
    }
}
// Replica fold and G = Gt + Gt^T (and F = H + G) of a fixed-point build in ONE launch (single-rank builds: the folded planes are not
// needed for an all-reduce): thread (i <= j) sums the `nrep` replicas of the four integers hi[ij], hi[ji], lo[ij], lo[ji] - eight replicas
// requested at a time - and writes the pair; integer sums are exact, so the result is the one of qc_reduce_replicas + qc_symmetrize_add.
// The replicas it has read are zeroed on the way: the next build finds clean accumulator planes without a 1.7 MB memset of its own.
__global__ __launch_bounds__(64) void qc_fold_symmetrize_kernel(int n, int nrep, size_t rep_stride, long long *__restrict__ Gh, size_t lo_off,
                                                                double *__restrict__ G, const double *__restrict__ H, double *__restrict__ F,
                                                                const double *__restrict__ fxs, unsigned long long *tl,
                                                                const unsigned *join_cnt, unsigned join_target, int *timeout_flag, long long limit) {
    if (tl != nullptr && threadIdx.x == 0 && blockIdx.x == 0) tl[0] = wall_clock64();
    if (join_cnt != nullptr) {
        // the device-side join of the build's side streams (qc_join_wait_kernel's loop, qc_fock.hip), by every workgroup of this launch
        // itself: it sits behind the last class kernel of the handle's own stream and goes on when the other streams' markers are in
        long long t0 = 0;
        unsigned it = 0;
        while ((int)(__hip_atomic_load(join_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - join_target) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if ((++it & 63u) == 0) {
                const long long t = wall_clock64();
                if (t0 == 0) t0 = t;
                else if (t - t0 > limit) { if (threadIdx.x == 0) __hip_atomic_store(timeout_flag, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    struct Leave { unsigned long long *tl; __device__ ~Leave() { if (tl != nullptr && threadIdx.x == 0) tl[1 + (blockIdx.x & 31)] = wall_clock64(); } } leave{tl};
    // lane = (element of this workgroup's 16, quarter of the replicas): one batch of loads per lane, the quarters meet by shuffles
    const int e = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int x = blockIdx.x * 16 + e;
    const bool in = x < n * n;
    const int i = in ? x / n : 0, j = in ? x - i * n : 0;
    const bool act = in && i <= j;
    const int xx = act ? x : 0, xt = act ? j * n + i : 0;
    long long *__restrict__ Gl = Gh + lo_off;
    const int per = (nrep + 3) / 4, r0 = part * per, r1 = min(nrep, r0 + per);
    long long h = 0, l = 0;
    for (int rb = r0; rb < r1; rb += 8) {
        long long a[8], b[8], c[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t o = (size_t)min(rb + u, nrep - 1) * rep_stride;
            a[u] = Gh[o + xx]; b[u] = Gh[o + xt]; c[u] = Gl[o + xx]; d[u] = Gl[o + xt];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (act && rb + u < r1) {
                h += a[u] + (i == j ? 0 : b[u]); l += c[u] + (i == j ? 0 : d[u]);
                const size_t o = (size_t)(rb + u) * rep_stride;
                Gh[o + xx] = 0; Gh[o + xt] = 0; Gl[o + xx] = 0; Gl[o + xt] = 0;
            }
    }
    h += __shfl_xor(h, 1, 4); l += __shfl_xor(l, 1, 4);
    h += __shfl_xor(h, 2, 4); l += __shfl_xor(l, 2, 4);
    if (!act || part != 0) return;
    if (i == j) { h *= 2; l *= 2; }                       // (diagonal: Gt_ii + Gt_ii)
    const double u1 = fxs[1], u2 = u1 * 0x1p-32;
    const double g = fma((double)l, u2, (double)h * u1);
    G[x] = g; G[xt] = g;
    if (F) { F[x] = 1.0 * H[x] + 1.0 * g; F[xt] = 1.0 * H[xt] + 1.0 * g; }
}
void qc_fold_symmetrize(hipStream_t st, int n, int nrep, size_t rep_stride, double *Gt, size_t lo_off, double *G, const double *H, double *F,
                        const double *fxs, unsigned long long *tl, const unsigned *join_cnt, unsigned join_target, int *timeout_flag, long long limit) {
    hipLaunchKernelGGL(qc_fold_symmetrize_kernel, dim3((n * n + 15) / 16), dim3(64), 0, st, n, nrep, rep_stride,
                       reinterpret_cast<long long *>(Gt), lo_off, G, H, F, fxs, tl, join_cnt, join_target, timeout_flag, limit);
}
void qc_symmetrize_add(hipStream_t st, int n, const double *Gt, size_t lo_off, double *G, const double *H, double *F, const double *fxs) {
    if (fxs) hipLaunchKernelGGL(qc_symmetrize_add_kernel<true>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, Gt, lo_off, G, H, F, fxs);
    else hipLaunchKernelGGL(qc_symmetrize_add_kernel<false>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, Gt, lo_off, G, H, F, fxs);
}

// Fixed-point scale of one Fock build.  Every |(ij|kl)| <= imax (Schwarz) and every element of Gt is a sum of terms
// f (ij|kl) D_kl with |f| <= 2 in which each D_kl occurs at most once per kind (J, K), so |Gt_ij| <= 3 imax sum|D| and
// 2^S (4 imax sum|D|) <= 2^60 keeps every partial and final sum inside 62 bits - wrap-around cannot happen.  One workgroup,
// fixed reduction order: the same densities give the same scale on every rank.
__global__ __launch_bounds__(1024) void qc_fx_scale_kernel(int nn, const double *__restrict__ Da, const double *__restrict__ Db, double imax,
                                                           double *__restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int x = threadIdx.x; x < nn; x += 1024) s += fabs(Da[x]) + (Db ? fabs(Db[x]) : 0.0);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sh[k];
        const double bound = 4.0 * imax * t;
        int e = 0;
        if (bound > 0.0 && bound < 1e300) { (void)frexp(bound, &e); }      // bound < 2^e
        else if (!(bound < 1e300)) e = 1100;                              // inf / nan densities: coarsest scale
        int S = 60 - e;
        S = S > QC_FX_MAXBITS ? QC_FX_MAXBITS : (S < -900 ? -900 : S);
        // non-finite densities must stay visible: integers cannot carry a NaN, so the unit that turns the sums back into doubles
        // (qc_symmetrize_add_kernel) becomes NaN and every element of G with it - what f64 accumulation would have produced
        out[0] = ldexp(1.0, S); out[1] = (bound < 1e300) ? ldexp(1.0, -S) : __builtin_nan("");
    }
}
void qc_fx_scale(hipStream_t st, int n, const double *Da, const double *Db, double imax, double *out) {
    hipLaunchKernelGGL(qc_fx_scale_kernel, dim3(1), dim3(1024), 0, st, n * n, Da, Db, imax, out);
}

// flag[0] = number of positions where a and b differ bitwise (spin-symmetry test of the UHF build)
__global__ void qc_count_diff_kernel(size_t count, const double *a, const double *b, int *flag) {
    int d = 0;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < count; x += (size_t)gridDim.x * blockDim.x)
        d += (__double_as_longlong(a[x]) != __double_as_longlong(b[x])) ? 1 : 0;
    if (d) atomicAdd(flag, d);
}
void qc_count_diff(hipStream_t st, size_t count, const double *a, const double *b, int *flag) {
    hipLaunchKernelGGL(qc_count_diff_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, a, b, flag);
}

// out[p][x] = sum_r Gt[p][r][x]: folds the accumulation replicas of the Fock build, plane by plane (f64: one plane; fixed
// point: hi and lo planes, `plane_stride` doubles apart, summed as integers); the sums of the planes land back to back in `out`
// (a small matrix gives only a dozen workgroups, so the kernel is as long as its chain of loads: eight replicas are requested
// at a time - four independent partial sums, combined in a fixed order)
template <typename T>
__global__ void qc_reduce_replicas_kernel(size_t count, int nrep, size_t stride, int nplanes, size_t plane_stride, const T *Gt, T *out) {
    for (size_t y = blockIdx.x * (size_t)blockDim.x + threadIdx.x; y < count * nplanes; y += (size_t)gridDim.x * blockDim.x) {
        const size_t p = y / count, x = y - p * count;
        const T *base = Gt + p * plane_stride + x;
        T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int r = 0;
        for (; r + 8 <= nrep; r += 8) {
            const T *g = base + (size_t)r * stride;
            const T a0 = g[0], a1 = g[stride], a2 = g[2 * stride], a3 = g[3 * stride];
            const T a4 = g[4 * stride], a5 = g[5 * stride], a6 = g[6 * stride], a7 = g[7 * stride];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            s0 += a4; s1 += a5; s2 += a6; s3 += a7;
        }
        for (; r < nrep; ++r) s0 += base[(size_t)r * stride];
        out[y] = (s0 + s1) + (s2 + s3);
    }
}
void qc_reduce_replicas(hipStream_t st, size_t count, int nrep, size_t stride, const double *Gt, double *out, bool fx, size_t plane_stride) {
    const int nplanes = fx ? 2 : 1;
    const unsigned grid = (unsigned)((count * nplanes + 63) / 64);
    if (fx) hipLaunchKernelGGL(qc_reduce_replicas_kernel<long long>, dim3(grid), dim3(64), 0, st, count, nrep, stride, nplanes, plane_stride,
                               reinterpret_cast<const long long *>(Gt), reinterpret_cast<long long *>(out));
    else hipLaunchKernelGGL(qc_reduce_replicas_kernel<double>, dim3(grid), dim3(64), 0, st, count, nrep, stride, nplanes, plane_stride, Gt, out);
}


// ---------------------------------------------------------------------------------------------------------------
// Stored-tensor ("conventional") Fock build: the reference's own algorithm with the n^4 tensor resident in HBM
// (288 GB hold it up to n ~ 400).  K5a: T[i,j,k,l] = I[i,j,k,l] - c I[i,k,j,l] (rhf.rs:58-62 with c = 1/2; c = 1 gives
// the exchange-permuted copy the UHF contraction of uhf.rs:216-226 needs).  K5b: G[i,j] = sum_kl D[k,l] T[i,j,k,l] for
// i <= j, mirrored (rhf.rs:152-167) - a GEMV over n(n+1)/2 rows of n^2 doubles: pure HBM streaming, 4 n^4 bytes.
__global__ void qc_permute_tensor_kernel(int n, const double *__restrict__ I, double c_direct, double c_exch, double *__restrict__ T) {
    const size_t n1 = n, n2 = n1 * n1, n3 = n2 * n1, n4 = n3 * n1;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n4; x += (size_t)gridDim.x * blockDim.x) {
        const size_t i = x / n3, r = x - i * n3, j = r / n2, r2 = r - j * n2, k = r2 / n1, l = r2 - k * n1;
        T[x] = c_direct * I[x] + c_exch * I[i * n3 + k * n2 + j * n1 + l];
    }
}
void qc_permute_tensor(hipStream_t st, int n, const double *I, double c_direct, double c_exch, double *T) {
    const size_t n4 = (size_t)n * n * n * n;
    hipLaunchKernelGGL(qc_permute_tensor_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 1u << 20)), dim3(256), 0, st, n, I, c_direct, c_exch,
                       T);
}

// Persistent workgroups, each streaming a strided set of upper-triangle rows (i <= j) of the tensor(s) against the
// density held in LDS (n^2 doubles: 27 KB at n = 58, 104 KB at n = 114; larger n read D through L2 instead).
// G[i,j] = G[j,i] = <T1[i,j,:], D1> + <T2[i,j,:], D2>   (T2/D2 optional).  16-byte loads.
constexpr int QC_GEMV_THREADS = 512;
template <bool D_IN_LDS>
__global__ __launch_bounds__(QC_GEMV_THREADS) void qc_tensor_gemv_kernel(int n, const double *__restrict__ T1, const double *__restrict__ D1,
                                                                         const double *__restrict__ T2, const double *__restrict__ D2,
                                                                         double *__restrict__ G) {
    extern __shared__ double sd[];
    __shared__ double sh[QC_GEMV_THREADS / 64];
    const size_t nn = (size_t)n * n, nv = nn / 2;
    const int tid = threadIdx.x, nrows = n * (n + 1) / 2;
    const double2 *d1 = reinterpret_cast<const double2 *>(D1), *d2 = reinterpret_cast<const double2 *>(D2);
    if (D_IN_LDS) {
        for (size_t x = tid; x < nn; x += QC_GEMV_THREADS) { sd[x] = D1[x]; if (T2) sd[nn + (nn & 1) + x] = D2[x]; }
        __syncthreads();
        d1 = reinterpret_cast<const double2 *>(sd);
        d2 = reinterpret_cast<const double2 *>(sd + nn + (nn & 1));
    }
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        int i = (int)((2.0 * n + 1.0 - sqrt((2.0 * n + 1.0) * (2.0 * n + 1.0) - 8.0 * row)) * 0.5);
        while (i > 0 && (size_t)i * n - (size_t)i * (i - 1) / 2 > (size_t)row) --i;
        while ((size_t)(i + 1) * n - (size_t)(i + 1) * i / 2 <= (size_t)row) ++i;
        const int j = i + (row - (int)((size_t)i * n - (size_t)i * (i - 1) / 2));
        const size_t base = ((size_t)i * n + j) * nn;
        double acc = 0.0;
        {
            const double2 *t1 = reinterpret_cast<const double2 *>(T1 + base);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            size_t x = tid;
            for (; x + 3 * QC_GEMV_THREADS < nv; x += 4 * QC_GEMV_THREADS) {        // 4 independent 16-byte loads in flight per lane
                const double2 p0 = t1[x], p1 = t1[x + QC_GEMV_THREADS], p2 = t1[x + 2 * QC_GEMV_THREADS], p3 = t1[x + 3 * QC_GEMV_THREADS];
                const double2 q0 = d1[x], q1 = d1[x + QC_GEMV_THREADS], q2 = d1[x + 2 * QC_GEMV_THREADS], q3 = d1[x + 3 * QC_GEMV_THREADS];
                a0 = fma(p0.x, q0.x, fma(p0.y, q0.y, a0)); a1 = fma(p1.x, q1.x, fma(p1.y, q1.y, a1));
                a2 = fma(p2.x, q2.x, fma(p2.y, q2.y, a2)); a3 = fma(p3.x, q3.x, fma(p3.y, q3.y, a3));
            }
            for (; x < nv; x += QC_GEMV_THREADS) { const double2 a = t1[x], b = d1[x]; a0 = fma(a.x, b.x, fma(a.y, b.y, a0)); }
            acc = (a0 + a1) + (a2 + a3);
            if ((nn & 1) && tid == 0) acc = fma(T1[base + nn - 1], D1[nn - 1], acc);
        }
        if (T2) {
            const double2 *t2 = reinterpret_cast<const double2 *>(T2 + base);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            size_t x = tid;
            for (; x + 3 * QC_GEMV_THREADS < nv; x += 4 * QC_GEMV_THREADS) {
                const double2 p0 = t2[x], p1 = t2[x + QC_GEMV_THREADS], p2 = t2[x + 2 * QC_GEMV_THREADS], p3 = t2[x + 3 * QC_GEMV_THREADS];
                const double2 q0 = d2[x], q1 = d2[x + QC_GEMV_THREADS], q2 = d2[x + 2 * QC_GEMV_THREADS], q3 = d2[x + 3 * QC_GEMV_THREADS];
                a0 = fma(p0.x, q0.x, fma(p0.y, q0.y, a0)); a1 = fma(p1.x, q1.x, fma(p1.y, q1.y, a1));
                a2 = fma(p2.x, q2.x, fma(p2.y, q2.y, a2)); a3 = fma(p3.x, q3.x, fma(p3.y, q3.y, a3));
            }
            for (; x < nv; x += QC_GEMV_THREADS) { const double2 a = t2[x], b = d2[x]; a0 = fma(a.x, b.x, fma(a.y, b.y, a0)); }
            acc += (a0 + a1) + (a2 + a3);
            if ((nn & 1) && tid == 0) acc = fma(T2[base + nn - 1], D2[nn - 1], acc);
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((tid & 63) == 0) sh[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            double r = 0.0;
            for (int k = 0; k < QC_GEMV_THREADS / 64; ++k) r += sh[k];
            G[(size_t)i * n + j] = r; G[(size_t)j * n + i] = r;
        }
        __syncthreads();
    }
}
void qc_tensor_gemv(hipStream_t st, int n, const double *T1, const double *D1, const double *T2, const double *D2, double *G) {
    const size_t nn = (size_t)n * n, lds = (T2 ? 2 : 1) * (nn + 1) * sizeof(double);
    const int nrows = n * (n + 1) / 2;
    if (lds <= 150 * 1024) {
        static size_t allowed = 48 * 1024;
        if (lds > allowed) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(qc_tensor_gemv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); allowed = lds; }
        const int per_cu = std::max<int>(1, std::min<int>(4, (int)(150 * 1024 / lds)));
        hipLaunchKernelGGL(qc_tensor_gemv_kernel<true>, dim3(std::min(nrows, 256 * per_cu)), dim3(QC_GEMV_THREADS), lds, st, n, T1, D1, T2, D2, G);
    } else {
        hipLaunchKernelGGL(qc_tensor_gemv_kernel<false>, dim3(std::min(nrows, 256 * 4)), dim3(QC_GEMV_THREADS), 0, st, n, T1, D1, T2, D2, G);
    }
}

// out[j] = <x, ys[j]>, one workgroup per j (diis.rs:43-45)
struct QcPtrList { const double *p[16]; };
__global__ __launch_bounds__(1024) void qc_dots_kernel(int nn, const double *x, QcPtrList ys, double *out) {
    __shared__ double sh[16];
    const double *y = ys.p[blockIdx.x];
    double s = 0.0;
    for (int i0 = threadIdx.x; i0 < nn; i0 += 4 * 1024) {       // (four elements of each array requested before the first is used: 16 -> 6 us at n = 114)
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * 1024, nn - 1); a[u] = x[i]; b[u] = y[i]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * 1024 < nn) s = fma(a[u], b[u], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sh[k];
        out[blockIdx.x] = t;
    }
}
void qc_dots(hipStream_t st, int n, const double *x, const double *const *ys, int ny, double *out) {
    QcPtrList l;
    for (int j = 0; j < ny; ++j) l.p[j] = ys[j];
    hipLaunchKernelGGL(qc_dots_kernel, dim3(ny), dim3(1024), 0, st, n * n, x, l, out);
}

// out2[0] = 0.5 tr(Dnew (2H + G)),  out2[1] = sum_i (Dnew - Dold)_ii^2      (rhf.rs:84-88)
// `out2` may be pinned host memory (the SCF pass reads its scalars without a copy); with `ctl` the control words are
// copied next to them (ctl_out) and cleared for the next pass.
// (one workgroup: the kernel is as long as a thread's chain of loads, hence 1024 threads and a separate pass over the diagonal)
__global__ __launch_bounds__(1024) void qc_energy_rms_kernel(int n, const double *Dn, const double *Do, const double *H, const double *G,
                                                             double *out2, int *ctl, int *ctl_out) {
    __shared__ double sh[2][16];
    double e = 0.0, r = 0.0;
    // tr(Dn (2H + G)) = sum_x Dn[x] (2H + G)[x^T]; H and G are symmetric bit for bit (qc_one_electron.hip, qc_symmetrize_add_kernel), so the
    // transposed elements are the elements themselves and every read is coalesced; eight elements per thread are requested at a time
    // (as a chain of thirteen strided trips to L2 this kernel took 25 us at n = 114, now 11)
    const int nn = n * n;
    for (int x0 = threadIdx.x; x0 < nn; x0 += 8 * 1024) {
        double d[8], h[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int x = min(x0 + u * 1024, nn - 1); d[u] = Dn[x]; h[u] = H[x]; g[u] = G[x]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (x0 + u * 1024 < nn) e = fma(d[u], 2.0 * h[u] + g[u], e);
    }
    for (int i = threadIdx.x; i < n; i += 1024) { const double d = Dn[(size_t)i * n + i] - Do[(size_t)i * n + i]; r = fma(d, d, r); }
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o, 64); r += __shfl_down(r, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = e; sh[1][threadIdx.x >> 6] = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        e = 0.0; r = 0.0;
        for (int k = 0; k < 16; ++k) { e += sh[0][k]; r += sh[1][k]; }
        out2[0] = 0.5 * e; out2[1] = r;
    }
    if (ctl && threadIdx.x < 16) { ctl_out[threadIdx.x] = ctl[threadIdx.x]; ctl[threadIdx.x] = 0; }
    __threadfence_system();
}
void qc_energy_rms(hipStream_t st, int n, const double *Dnew, const double *Dold, const double *H, const double *G, double *out2, int *ctl,
                   int *ctl_out) {
    hipLaunchKernelGGL(qc_energy_rms_kernel, dim3(1), dim3(1024), 0, st, n, Dnew, Dold, H, G, out2, ctl, ctl_out);
}

// w[nwords + i] = ~w[i]: the complements that let a max all-reduce over bit patterns reveal a rank whose words differ
__global__ void qc_sync_pack_kernel(unsigned long long *w, int nwords) {
    if ((int)threadIdx.x < nwords) w[nwords + threadIdx.x] = ~w[threadIdx.x];
}
void qc_sync_pack(hipStream_t st, unsigned long long *w, int nwords) { hipLaunchKernelGGL(qc_sync_pack_kernel, dim3(1), dim3(64), 0, st, w, nwords); }

// out = sum_i c[i] * Fs[i]   (diis.rs:52-58)
struct QcCoefList { double c[16]; };
__global__ void qc_lincomb_kernel(int nn, QcPtrList Fs, QcCoefList c, int m, double *out) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < nn; x += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s = fma(c.c[i], Fs.p[i][x], s);
        out[x] = s;
    }
}
__global__ void qc_lincomb_dev_kernel(int nn, QcPtrList Fs, const double *__restrict__ c, int m, double *out) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < nn; x += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s = fma(c[i], Fs.p[i][x], s);
        out[x] = s;
    }
}
void qc_lincomb_dev(hipStream_t st, int n, const double *const *Fs, const double *c_dev, int m, double *out) {
    QcPtrList l;
    for (int i = 0; i < m; ++i) l.p[i] = Fs[i];
    hipLaunchKernelGGL(qc_lincomb_dev_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, n * n, l, c_dev, m, out);
}
void qc_lincomb(hipStream_t st, int n, const double *const *Fs, const double *c, int m, double *out) {
    QcPtrList l; QcCoefList cl;
    for (int i = 0; i < m; ++i) { l.p[i] = Fs[i]; cl.c[i] = c[i]; }
    hipLaunchKernelGGL(qc_lincomb_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, n * n, l, cl, m, out);
}

// X = U diag(lam_ii^-1/2) U^T helper: out[i][k] = U[i][k] / sqrt(Lam[k][k])   (rhf.rs:128-130)
__global__ void qc_scale_cols_invsqrt_kernel(int n, const double *U, const double *Lam, double *out) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * n; x += gridDim.x * blockDim.x) {
        const int k = x % n;
        out[x] = U[x] / sqrt(Lam[k * n + k]);
    }
}
void qc_scale_cols_invsqrt(hipStream_t st, int n, const double *U, const double *Lam, double *out) {
    hipLaunchKernelGGL(qc_scale_cols_invsqrt_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, n, U, Lam, out);
}
