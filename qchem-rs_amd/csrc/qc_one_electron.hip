// qc_one_electron.hip - overlap, kinetic-energy and nuclear-attraction matrices on the GPU.
//
// Replaces molint::overlap / kinetic / nuclear (call sites rhf.rs:41-43, uhf.rs:52-54) - the first "next" row of the
// scope table: the last CPU-only start-up phase of an SCF run.  Same McMurchie-Davidson formulas as the
// host version (qc_host_one_electron, qc_system.cpp), which stays as the GPU-free entry point of the C ABI:
//   S_ab = (pi/p)^{3/2} E^x_0 E^y_0 E^z_0,
//   T_ab = -1/2 sum_axis <a| d^2/dx^2 |b>   (second derivative of the ket primitive: b+2, b, b-2 terms),
//   V_ab = - sum_C Z_C (2 pi / p) sum_tuv E^x_t E^y_u E^z_v R_tuv(p, P - C).
// One wave per shell pair; the lanes share out its work items - primitive pairs (S, T) or primitive pair x nucleus (V) -
// and add their Cartesian blocks into LDS (ds_add_f64); the block is then transformed to the shells' functions.
// This is set-up code, run once per geometry: it is written for clarity, not tuned (tables in scratch memory).
#include "qc_internal.h"

namespace {

constexpr int MAXC = 10;          // Cartesian components of an f shell
constexpr int EI = QC_LMAX + 1, EJ = QC_LMAX + 3, ET = 2 * QC_LMAX + 4;

struct DevShell { double A[3]; int L, nprim, ncart, nfunc, off, poff, toff, pad; };

// 1-D Hermite expansion coefficients E[i][j][t] of x_A^i x_B^j exp(-a x_A^2 - b x_B^2), i <= imax, j <= jmax
struct E1 {
    double v[EI][EJ][ET];
    __device__ double get(int i, int j, int t) const { return (t < 0 || t > i + j) ? 0.0 : v[i][j][t]; }
};
__device__ void hermite_e(E1 &E, int imax, int jmax, double a, double b, double Q) {
    const double p = a + b, h = 0.5 / p, xpa = -b / p * Q, xpb = a / p * Q;
    for (int i = 0; i <= imax; ++i) for (int j = 0; j <= jmax; ++j) for (int t = 0; t < ET; ++t) E.v[i][j][t] = 0.0;
    E.v[0][0][0] = exp(-a * b / p * Q * Q);
    for (int i = 1; i <= imax; ++i)
        for (int t = 0; t <= i; ++t) E.v[i][0][t] = h * E.get(i - 1, 0, t - 1) + xpa * E.get(i - 1, 0, t) + (t + 1) * E.get(i - 1, 0, t + 1);
    for (int j = 1; j <= jmax; ++j)
        for (int i = 0; i <= imax; ++i)
            for (int t = 0; t <= i + j; ++t) E.v[i][j][t] = h * E.get(i, j - 1, t - 1) + xpb * E.get(i, j - 1, t) + (t + 1) * E.get(i, j - 1, t + 1);
}

// F_n(x), n = 0..nmax: Kummer series at nmax + downward recursion; asymptotic form + upward recursion for large x
__device__ void boys_series(int nmax, double x, double *F) {
    const double ex = exp(-x);
    if (x < 38.0) {
        double term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 500; ++k) { term *= 2.0 * x / (2 * nmax + 2 * k + 1); sum += term; if (term < 1e-18 * sum) break; }
        F[nmax] = ex * sum;
        for (int n = nmax; n > 0; --n) F[n - 1] = (2.0 * x * F[n] + ex) / (2 * n - 1);
    } else {
        F[0] = 0.5 * sqrt(M_PI / x) * erf(sqrt(x));
        for (int n = 0; n < nmax; ++n) F[n + 1] = ((2 * n + 1) * F[n] - ex) / (2.0 * x);
    }
}

// Hermite Coulomb integrals R^0_tuv, t+u+v <= L
__device__ void hermite_r(int L, double alpha, const double *PC, double (*W)[qc_nherm(QC_LPAIR)]) {
    double F[QC_LPAIR + 1];
    boys_series(L, alpha * (PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]), F);
    double f = 1.0;
    for (int n = 0; n <= L; ++n) { W[n][0] = f * F[n]; f *= -2.0 * alpha; }
    for (int N = 1; N <= L; ++N)
        for (int n = 0; n + N <= L; ++n)
            for (int t = N; t >= 0; --t)
                for (int u = N - t; u >= 0; --u) {
                    const int v = N - t - u;
                    double val;
                    if (t) val = PC[0] * W[n + 1][qc_hidx(t - 1, u, v)] + (t > 1 ? (t - 1) * W[n + 1][qc_hidx(t - 2, u, v)] : 0.0);
                    else if (u) val = PC[1] * W[n + 1][qc_hidx(t, u - 1, v)] + (u > 1 ? (u - 1) * W[n + 1][qc_hidx(t, u - 2, v)] : 0.0);
                    else val = PC[2] * W[n + 1][qc_hidx(t, u, v - 1)] + (v > 1 ? (v - 1) * W[n + 1][qc_hidx(t, u, v - 2)] : 0.0);
                    W[n][qc_hidx(t, u, v)] = val;
                }
}

// Cartesian component x of order L: (lx, ly, lz) in the order of cart_list() (qc_system.cpp)
__device__ void cart_of(int L, int x, int *l) {
    int k = 0;
    for (int lx = L; lx >= 0; --lx)
        for (int ly = L - lx; ly >= 0; --ly, ++k)
            if (k == x) { l[0] = lx; l[1] = ly; l[2] = L - lx - ly; return; }
}

typedef __attribute__((address_space(3))) double lds_double;

__global__ __launch_bounds__(64) void qc_one_electron_kernel(int which, int nshells, const DevShell *__restrict__ sh, const double *__restrict__ exps,
                                                             const double *__restrict__ coefs, const double *__restrict__ Tm, int natoms,
                                                             const int *__restrict__ Z, const double *__restrict__ xyz, int n, double *__restrict__ out) {
    __shared__ double cart[MAXC * MAXC];
    // shell pair (a >= b) of this workgroup
    int a = 0, rem = blockIdx.x;
    while (rem > a) { rem -= a + 1; ++a; }
    const int b = rem;
    const DevShell A = sh[a], B = sh[b];
    const int lane = threadIdx.x, nca = A.ncart, ncb = B.ncart;
    for (int i = lane; i < MAXC * MAXC; i += 64) cart[i] = 0.0;
    __syncthreads();
    const int npp = A.nprim * B.nprim, nitems = (which == 2) ? npp * natoms : npp;
    for (int item = lane; item < nitems; item += 64) {
        const int pp = (which == 2) ? item / natoms : item, c = (which == 2) ? item - pp * natoms : 0;
        const int i = pp / B.nprim, j = pp - i * B.nprim;
        const double ea = exps[A.poff + i], eb = exps[B.poff + j], p = ea + eb, cc = coefs[A.poff + i] * coefs[B.poff + j];
        double P[3];
        for (int k = 0; k < 3; ++k) P[k] = (ea * A.A[k] + eb * B.A[k]) / p;
        E1 E[3];
        for (int k = 0; k < 3; ++k) hermite_e(E[k], A.L, B.L + 2, ea, eb, A.A[k] - B.A[k]);
        const double s3 = pow(M_PI / p, 1.5);
        double W[QC_LPAIR + 1][qc_nherm(QC_LPAIR)];
        double vpref = 0.0;
        if (which == 2) {
            const double PC[3] = {P[0] - xyz[3 * c], P[1] - xyz[3 * c + 1], P[2] - xyz[3 * c + 2]};
            hermite_r(A.L + B.L, p, PC, W);
            vpref = -(double)Z[c] * 2.0 * M_PI / p;
        }
        for (int x = 0; x < nca; ++x) {
            int ai[3];
            cart_of(A.L, x, ai);
            for (int y = 0; y < ncb; ++y) {
                int bi[3];
                cart_of(B.L, y, bi);
                double val;
                if (which == 0) {
                    val = s3 * E[0].get(ai[0], bi[0], 0) * E[1].get(ai[1], bi[1], 0) * E[2].get(ai[2], bi[2], 0);
                } else if (which == 1) {
                    double s1[3], t1[3];
                    for (int k = 0; k < 3; ++k) {
                        s1[k] = E[k].get(ai[k], bi[k], 0);
                        t1[k] = 4.0 * eb * eb * E[k].get(ai[k], bi[k] + 2, 0) - 2.0 * eb * (2 * bi[k] + 1) * s1[k];
                        if (bi[k] >= 2) t1[k] += bi[k] * (bi[k] - 1) * E[k].get(ai[k], bi[k] - 2, 0);
                    }
                    val = -0.5 * s3 * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
                } else {
                    double acc = 0.0;
                    for (int t = 0; t <= ai[0] + bi[0]; ++t)
                        for (int u = 0; u <= ai[1] + bi[1]; ++u)
                            for (int v = 0; v <= ai[2] + bi[2]; ++v)
                                acc += E[0].get(ai[0], bi[0], t) * E[1].get(ai[1], bi[1], u) * E[2].get(ai[2], bi[2], v) * W[0][qc_hidx(t, u, v)];
                    val = vpref * acc;
                }
                (void)__builtin_amdgcn_ds_atomic_fadd_f64((lds_double *)&cart[x * ncb + y], cc * val);
            }
        }
    }
    __syncthreads();
    // Cartesian -> the shells' functions (solid harmonics or scaled monomials), both triangles
    const double *Ta = Tm + A.toff, *Tb = Tm + B.toff;
    for (int f = lane; f < A.nfunc * B.nfunc; f += 64) {
        const int fa = f / B.nfunc, fb = f - fa * B.nfunc;
        if (a == b && fb > fa) continue;                   // diagonal blocks: one triangle, mirrored (exactly symmetric output)
        double v = 0.0;
        for (int x = 0; x < nca; ++x)
            for (int y = 0; y < ncb; ++y) v += Ta[fa * nca + x] * Tb[fb * ncb + y] * cart[x * ncb + y];
        out[(size_t)(A.off + fa) * n + B.off + fb] = v;
        out[(size_t)(B.off + fb) * n + A.off + fa] = v;
    }
}

}  // namespace

// which: 0 overlap, 1 kinetic, 2 nuclear attraction; d_out: n x n device matrix
int qc_one_electron_device(qc_system *S, int which, double *d_out) {
    if (which < 0 || which > 2) return QC_ERR_INVALID;
    if (!S->d_shells) {       // shells, primitives, transformation matrices: uploaded at the first call
        std::vector<DevShell> hs(S->nshells);
        std::vector<double> ex, co, tm;
        for (int s = 0; s < S->nshells; ++s) {
            const QcShell &q = S->shells[s];
            DevShell d{};
            for (int k = 0; k < 3; ++k) d.A[k] = q.A[k];
            d.L = q.L; d.nprim = q.nprim; d.ncart = q.ncart; d.nfunc = q.nfunc; d.off = q.off;
            d.poff = (int)ex.size(); d.toff = (int)tm.size();
            ex.insert(ex.end(), q.exps.begin(), q.exps.end());
            co.insert(co.end(), q.coefs.begin(), q.coefs.end());
            tm.insert(tm.end(), q.T.begin(), q.T.end());
            hs[s] = d;
        }
        const size_t bytes_s = hs.size() * sizeof(DevShell), bytes_p = ex.size() * sizeof(double), bytes_t = tm.size() * sizeof(double);
        char *blob = nullptr;
        const size_t off_e = (bytes_s + 63) & ~(size_t)63, off_c = off_e + ((bytes_p + 63) & ~(size_t)63), off_t = off_c + ((bytes_p + 63) & ~(size_t)63);
        const size_t off_z = off_t + ((bytes_t + 63) & ~(size_t)63), off_x = off_z + ((S->natoms * sizeof(int) + 63) & ~(size_t)63);
        const size_t total = off_x + S->natoms * 3 * sizeof(double);
        QC_HIP_CHECK(hipMalloc(&blob, total));
        QC_HIP_CHECK(hipMemcpy(blob, hs.data(), bytes_s, hipMemcpyHostToDevice));
        QC_HIP_CHECK(hipMemcpy(blob + off_e, ex.data(), bytes_p, hipMemcpyHostToDevice));
        QC_HIP_CHECK(hipMemcpy(blob + off_c, co.data(), bytes_p, hipMemcpyHostToDevice));
        QC_HIP_CHECK(hipMemcpy(blob + off_t, tm.data(), bytes_t, hipMemcpyHostToDevice));
        QC_HIP_CHECK(hipMemcpy(blob + off_z, S->Z.data(), S->natoms * sizeof(int), hipMemcpyHostToDevice));
        QC_HIP_CHECK(hipMemcpy(blob + off_x, S->xyz.data(), S->natoms * 3 * sizeof(double), hipMemcpyHostToDevice));
        S->d_shells = blob;
        S->shell_blob_off[0] = off_e; S->shell_blob_off[1] = off_c; S->shell_blob_off[2] = off_t; S->shell_blob_off[3] = off_z; S->shell_blob_off[4] = off_x;
    }
    char *blob = static_cast<char *>(S->d_shells);
    const int npairs = S->nshells * (S->nshells + 1) / 2;
    hipLaunchKernelGGL(qc_one_electron_kernel, dim3(npairs), dim3(64), 0, S->stream, which, S->nshells, reinterpret_cast<const DevShell *>(blob),
                       reinterpret_cast<const double *>(blob + S->shell_blob_off[0]), reinterpret_cast<const double *>(blob + S->shell_blob_off[1]),
                       reinterpret_cast<const double *>(blob + S->shell_blob_off[2]), S->natoms, reinterpret_cast<const int *>(blob + S->shell_blob_off[3]),
                       reinterpret_cast<const double *>(blob + S->shell_blob_off[4]), S->nbasis, d_out);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}
