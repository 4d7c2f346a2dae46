// qc_linalg.hip - dense f64 linear algebra of the SCF iteration on gfx950: MFMA GEMM, symmetric eigensolver,
// element-wise and reduction kernels.  All matrices row-major, device pointers, on the caller's stream.
//
// Replaces the nalgebra operations in the reference's loop body: the `*` products at rhf.rs:71,74,76,85,
// SymmetricEigen behind utils::sorted_eigs (hf/utils.rs:20-36), the Frobenius dots of diis.rs:43-45 and the
// trace / diagonal-rms at rhf.rs:84-88.
#include "qc_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// C = alpha * op(A) * op(B) + beta * C with v_mfma_f64_16x16x4_f64.  One wave per 16x16 tile of C.
// Operand lane maps (cdna_hip_programming.md section 3): lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r holds C[row = (l >> 4) + 4 r][col = l & 15].
__global__ __launch_bounds__(64) void qc_gemm_kernel(int m, int n, int k, double alpha, const double *__restrict__ A, int lda, int ta,
                                                     const double *__restrict__ B, int ldb, int tb, double beta, double *__restrict__ C,
                                                     int ldc) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    const int ai = row0 + li, bj = col0 + li;
    const bool aok = ai < m, bok = bj < n;
    const size_t a_i = ta ? (size_t)ai : (size_t)ai * lda, a_k = ta ? (size_t)lda : 1;
    const size_t b_j = tb ? (size_t)bj * ldb : (size_t)bj, b_k = tb ? 1 : (size_t)ldb;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < k; k0 += 16) {
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = k0 + 4 * s + lk;
            av[s] = (aok && kk < k) ? A[a_i + kk * a_k] : 0.0;
            bv[s] = (bok && kk < k) ? B[b_j + kk * b_k] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    if (bok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + lk + 4 * r;
            if (row < m) {
                double *c = &C[(size_t)row * ldc + bj];
                *c = (beta == 0.0) ? alpha * acc[r] : fma(alpha, acc[r], beta * *c);
            }
        }
    }
}

void qc_gemm(hipStream_t st, int m, int n, int k, double alpha, const double *A, int lda, bool ta, const double *B, int ldb, bool tb,
             double beta, double *C, int ldc) {
    dim3 grid((n + 15) / 16, (m + 15) / 16);
    hipLaunchKernelGGL(qc_gemm_kernel, grid, dim3(64), 0, st, m, n, k, alpha, A, lda, ta ? 1 : 0, B, ldb, tb ? 1 : 0, beta, C, ldc);
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric eigensolver: parallel cyclic two-sided Jacobi, one workgroup, matrix resident in LDS.
// Round-robin ("chess tournament") ordering gives m/2 disjoint rotations per step, m-1 steps per sweep.
// Output: eigenvalues ascending in w, eigenvectors as the columns of V (sorted alongside).
constexpr int QC_EIG_THREADS = 1024;

__global__ __launch_bounds__(QC_EIG_THREADS) void qc_jacobi_kernel(int n, const double *__restrict__ Ain, double *__restrict__ V,
                                                                   double *__restrict__ Vs, double *__restrict__ w, int max_sweeps) {
    extern __shared__ double sm[];
    const int ld = n | 1;
    const int m = (n + 1) & ~1, half = m / 2;
    double *A = sm;                       // n x ld
    double *cs = A + (size_t)n * ld;      // 2 * half
    int *pq = (int *)(cs + 2 * half);     // 2 * half
    double *red = (double *)(pq + 2 * half + (half & 1 ? 0 : 0));  // 32 partials (+2)
    red = (double *)(((uintptr_t)red + 7) & ~(uintptr_t)7);
    const int tid = threadIdx.x, nt = blockDim.x;

    for (int x = tid; x < n * n; x += nt) {
        const int i = x / n, j = x - i * n;
        A[i * ld + j] = Ain[x];
        V[x] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // convergence: off-diagonal Frobenius mass relative to the whole matrix
        double off = 0.0, tot = 0.0;
        for (int x = tid; x < n * n; x += nt) {
            const int i = x / n, j = x - i * n;
            const double v = A[i * ld + j];
            tot += v * v;
            if (i != j) off += v * v;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); tot += __shfl_down(tot, o, 64); }
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = off; red[2 * (tid >> 6) + 1] = tot; }
        __syncthreads();
        if (tid == 0) {
            double so = 0.0, stt = 0.0;
            for (int k = 0; k < nt / 64; ++k) { so += red[2 * k]; stt += red[2 * k + 1]; }
            red[40] = so; red[41] = stt;
        }
        __syncthreads();
        const double soff = red[40], stot = red[41];
        __syncthreads();
        if (soff <= 1e-30 * stot || soff == 0.0) break;

        for (int step = 0; step < m - 1; ++step) {
            // phase 1: the rotations of this step
            if (tid < half) {
                int p, q;
                if (tid == 0) { p = m - 1; q = step; }
                else { p = (step + tid) % (m - 1); q = (step - tid + (m - 1)) % (m - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p * ld + q];
                    if (apq != 0.0) {
                        const double theta = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                    }
                } else {
                    p = -1;   // padding pair
                }
                pq[2 * tid] = p; pq[2 * tid + 1] = q;
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // phase 2: columns  A <- A J,  V <- V J
            for (int x = tid; x < n * half; x += nt) {
                const int i = x / half, k = x - i * half;
                const int p = pq[2 * k], q = pq[2 * k + 1];
                if (p < 0) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (s == 0.0) continue;
                const double aip = A[i * ld + p], aiq = A[i * ld + q];
                A[i * ld + p] = c * aip - s * aiq;
                A[i * ld + q] = s * aip + c * aiq;
                const double vip = V[i * n + p], viq = V[i * n + q];
                V[i * n + p] = c * vip - s * viq;
                V[i * n + q] = s * vip + c * viq;
            }
            __syncthreads();
            // phase 3: rows  A <- J^T A
            for (int x = tid; x < half * n; x += nt) {
                const int k = x / n, j = x - k * n;
                const int p = pq[2 * k], q = pq[2 * k + 1];
                if (p < 0) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (s == 0.0) continue;
                const double apj = A[p * ld + j], aqj = A[q * ld + j];
                A[p * ld + j] = c * apj - s * aqj;
                A[q * ld + j] = s * apj + c * aqj;
            }
            __syncthreads();
        }
    }
    // ascending order (utils.rs:28): rank sort, then permute columns
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const double wi = A[i * ld + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double wj = A[j * ld + j];
            rank += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
        }
        pq[i] = rank;     // pq has 2*half >= n entries
        w[rank] = wi;
    }
    __syncthreads();
    for (int x = tid; x < n * n; x += nt) {
        const int i = x / n, j = x - i * n;
        Vs[i * n + pq[j]] = V[x];
    }
}

// dA: input (left intact), dV: sorted eigenvectors, dw: eigenvalues, d_work: n*n scratch
int qc_eig_device(hipStream_t st, int n, double *dA, double *dV, double *dw, double *d_work) {
    const int ld = n | 1, m = (n + 1) & ~1;
    const size_t lds = ((size_t)n * ld + 2 * (m / 2)) * sizeof(double) + 2 * (m / 2) * sizeof(int) + 64 * sizeof(double);
    if (lds > 160 * 1024) return QC_ERR_UNSUPPORTED;   // n <= 141; larger n needs the multi-workgroup solver (next round)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qc_jacobi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return QC_ERR_HIP;
    }
    hipLaunchKernelGGL(qc_jacobi_kernel, dim3(1), dim3(QC_EIG_THREADS), lds, st, n, dA, d_work, dV, dw, 60);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
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
__global__ void qc_symmetrize_add_kernel(int n, const double *Gt, double *G) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * n; x += gridDim.x * blockDim.x) {
        const int i = x / n, j = x - i * n;
        G[x] = Gt[x] + Gt[j * n + i];
    }
}
void qc_symmetrize_add(hipStream_t st, int n, const double *Gt, double *G) {
    hipLaunchKernelGGL(qc_symmetrize_add_kernel, dim3((n * n + 255) / 256), dim3(256), 0, st, n, Gt, G);
}

__device__ __forceinline__ double block_sum_256(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// out[j] = <x, ys[j]>, one workgroup per j (diis.rs:43-45)
struct QcPtrList { const double *p[16]; };
__global__ __launch_bounds__(256) void qc_dots_kernel(int nn, const double *x, QcPtrList ys, double *out) {
    __shared__ double sh[4];
    const double *y = ys.p[blockIdx.x];
    double s = 0.0;
    for (int i = threadIdx.x; i < nn; i += 256) s = fma(x[i], y[i], s);
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
void qc_dots(hipStream_t st, int n, const double *x, const double *const *ys, int ny, double *out) {
    QcPtrList l;
    for (int j = 0; j < ny; ++j) l.p[j] = ys[j];
    hipLaunchKernelGGL(qc_dots_kernel, dim3(ny), dim3(256), 0, st, n * n, x, l, out);
}

// out2[0] = 0.5 tr(Dnew (2H + G)),  out2[1] = sum_i (Dnew - Dold)_ii^2      (rhf.rs:84-88)
__global__ __launch_bounds__(256) void qc_energy_rms_kernel(int n, const double *Dn, const double *Do, const double *H, const double *G,
                                                            double *out2) {
    __shared__ double sh[4];
    double e = 0.0, r = 0.0;
    for (int x = threadIdx.x; x < n * n; x += 256) {
        const int i = x / n, j = x - i * n;
        e = fma(Dn[x], 2.0 * H[j * n + i] + G[j * n + i], e);
        if (i == j) { const double d = Dn[x] - Do[x]; r = fma(d, d, r); }
    }
    e = block_sum_256(e, sh);
    r = block_sum_256(r, sh);
    if (threadIdx.x == 0) { out2[0] = 0.5 * e; out2[1] = r; }
}
void qc_energy_rms(hipStream_t st, int n, const double *Dnew, const double *Dold, const double *H, const double *G, double *out2) {
    hipLaunchKernelGGL(qc_energy_rms_kernel, dim3(1), dim3(256), 0, st, n, Dnew, Dold, H, G, out2);
}

// out = sum_i c[i] * Fs[i]   (diis.rs:52-58)
struct QcCoefList { double c[16]; };
__global__ void qc_lincomb_kernel(int nn, QcPtrList Fs, QcCoefList c, int m, double *out) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < nn; x += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s = fma(c.c[i], Fs.p[i][x], s);
        out[x] = s;
    }
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
