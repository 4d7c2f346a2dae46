// qc_eig_tridiag.hip - start vectors for the symmetric eigensolver by Householder tridiagonalisation (gfx950).
//
// Replaces the cold / far-from-converged eigensolves behind utils::sorted_eigs (hf/utils.rs:20-36; nalgebra's SymmetricEigen
// does the same thing: tridiagonalise, then iterate on the tridiagonal matrix).  The parallel Jacobi kernels of qc_linalg.hip
// touch the whole matrix in every one of their ~6 (n-1) steps and are bound by the LDS port of the single CU they run on
// (2 n^2 doubles read and written per step: 1.8 us per step at n = 114, 1.1-1.6 ms per eigensolve).  Here
//   1. qc_tridiag_kernel      A + pert = Q T Q^T: n-2 Householder steps in ONE workgroup, matrix resident in LDS (n <= 139, else in
//                             global memory), 4 barriers per step, every access along rows (both triangles are kept);
//   2. qc_tri_eig_kernel      eigenvalues of T by 9-way multisection on division-free Sturm sequences (8 lanes per eigenvalue),
//                             eigenvectors by twisted factorisation (one lane per eigenvalue, four O(n) sweeps);
//   3. qc_backtransform_kernel  X0 = Q Z: one wave per eigenvector applies the n-2 reflectors (whole chip);
// and the result goes into the Ogita-Aishima refinement (qc_eig_refine_async: f64 MFMA GEMMs), which converges quadratically
// and therefore only needs X0 to ~1e-3: that is why none of the delicate parts of a tridiagonal eigensolver are needed -
//   * exactly degenerate eigenvalues (benzene's E-type orbital pairs) would leave the twisted vectors of a pair parallel.  A
//     diagonal pseudo-random perturbation of relative size 1e-9 splits them first (any basis of a degenerate subspace is a valid
//     answer); the refinement then works on the unperturbed matrix and treats what is left as its "strong pairs";
//   * a start that is not good enough (orthogonality > 1e-3, clusters the refinement cannot rotate) is detected by the
//     refinement's control word and answered with the Jacobi kernels - correctness never rests on this file.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "qc_internal.h"

namespace {

constexpr int TRI_THREADS = 1024;
constexpr int TRI_TEAM = 8;                         // lanes per matrix row (matvec, update) / per eigenvalue (multisection)

// Lane exchanges inside a 16-lane row through DPP moves of the two 32-bit halves (64-bit DPP exists only for row_newbcast): a few
// cycles each, where the ds_bpermute behind __shfl_xor costs an LDS round trip - and these reductions sit on the critical path of
// every Householder step.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
// sum over the 8 lanes of a team (aligned groups of 8), result in every lane
__device__ __forceinline__ double team_sum(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    return v;
}
// sum over the wave, result in every lane; fixed order
__device__ __forceinline__ double wave_sum(double v) {
    v = team_sum(v);
    v += dpp_move<DPP_ROW_MIRROR>(v);                    // 16-lane row sums
    auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return (row(0) + row(16)) + (row(32) + row(48));
}
// Workgroup barrier that waits for the LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)) - here the reflector rows a step writes for the LATER kernels, a ~2.5 us round trip per step that made
// a Householder step cost the same for n = 24 as for n = 128.  Nothing inside these kernels reads those stores back.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// deterministic pseudo-random number in [-1, 1) from an index (splitmix64)
__device__ __forceinline__ double hash_unit(unsigned long long i) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(long long)(z >> 11) * 0x1p-52 - 1.0;
}

// 1 / x to a few ulp: hardware estimate + two Newton steps (half the dependent latency of the IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// ---- 1. A + diag(pert) = Q T Q^T.  Vr[k * n + i]: Householder vector of step k (zero for i <= k, one at k + 1); tri = [d (n) | e (n) | tau (n) | e^2 (n)].
template <bool IN_LDS>
__global__ __launch_bounds__(TRI_THREADS) void qc_tridiag_kernel(int n, int ld, const double *__restrict__ Ain, double rel_pert, double *__restrict__ Awork,
                                                                  double *__restrict__ Vr, double *__restrict__ tri) {
    extern __shared__ double sm[];
    double *A = IN_LDS ? sm : Awork;                     // n rows of stride ld, both triangles
    double *p = IN_LDS ? sm + (size_t)n * ld : sm;       // n (at least 16)
    double *sc = p + (n < 16 ? 16 : n);                  // 8 scalars
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *d = tri, *e = tri + n, *tau = tri + 2 * n;

    // load (symmetric part), scale = max |a_ii| + row sums -> perturbation size
    double amax = 0.0;
    for (int x = tid; x < n * n; x += TRI_THREADS) {
        const int i = x / n, j = x - i * n;
        const double a = 0.5 * (Ain[x] + Ain[(size_t)j * n + i]);
        A[(size_t)i * ld + j] = a;
        amax = fmax(amax, fabs(a));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) p[wave] = amax;
    __syncthreads();
    if (tid == 0) {
        double m = 0.0;
        for (int k = 0; k < TRI_THREADS / 64; ++k) m = fmax(m, p[k]);
        sc[0] = m;
    }
    __syncthreads();
    const double pert = rel_pert * sc[0];
    for (int i = tid; i < n; i += TRI_THREADS) A[(size_t)i * ld + i] += pert * hash_unit((unsigned long long)i);
    __syncthreads();

    // Two barriers per step, no single-wave phases: the Householder scalars of row k + 1 are worked out (rsqrt / rcp forms - the
    // IEEE sqrt and division sequences are most of a step when every wave runs them) by the team that has just updated that
    // row, while the other waves still update theirs; v is never materialised (v_j = row_k[j] * inv) and every wave forms the
    // scalar p.v itself.
    const int tl = tid & (TRI_TEAM - 1), team = tid / TRI_TEAM;
    auto leave_row_scalars = [&](const double *row, int kk) {      // team-wide: sc[2..4] = beta, tau, 1 / (x0 - beta) of row kk
        double s2 = 0.0;
        for (int j = kk + 2 + tl; j < n; j += TRI_TEAM) s2 = fma(row[j], row[j], s2);
        s2 = team_sum(s2);
        const double x0 = row[kk + 1];
        double beta = x0, t = 0.0, inv = 0.0;
        if (s2 > 0.0) {
            const double nn2 = fma(x0, x0, s2);
            double r = __builtin_amdgcn_rsq(nn2);                   // 1 / ||x||, two Newton steps
            r = r * fma(-0.5 * nn2 * r, r, 1.5);
            r = r * fma(-0.5 * nn2 * r, r, 1.5);
            const double mu = nn2 * r;
            beta = x0 <= 0.0 ? mu : -mu;
            t = (beta - x0) * (x0 <= 0.0 ? r : -r);                 // (beta - x0) / beta
            inv = fast_rcp(x0 - beta);
        }
        if (tl == 0) { sc[2] = beta; sc[3] = t; sc[4] = inv; }
    };
    if (team == 0) leave_row_scalars(A, 0);
    __syncthreads();
    for (int k = 0; k + 2 < n; ++k) {
        const int m = n - k - 1;                         // trailing block: indices k + 1 .. n - 1
        const double *rowk = A + (size_t)k * ld;
        const double beta = sc[2], t = sc[3], inv = sc[4];
        if (tid == 0) { d[k] = rowk[k]; e[k] = beta; tau[k] = t; tri[3 * n + k] = beta * beta; }
        for (int i = tid; i < n; i += TRI_THREADS) Vr[(size_t)k * n + i] = i <= k ? 0.0 : (i == k + 1 ? 1.0 : rowk[i] * inv);
        if (t != 0.0) {                                  // (workgroup-uniform)
            // p = tau A22 v: one team of 8 lanes per row
            for (int r = team; r < m; r += TRI_THREADS / TRI_TEAM) {
                const double *row = A + (size_t)(k + 1 + r) * ld;
                double s = tl == 0 ? row[k + 1] : 0.0;   // v_{k+1} = 1
#pragma unroll 4
                for (int j = k + 2 + tl; j < n; j += TRI_TEAM) s = fma(row[j], rowk[j] * inv, s);
                s = team_sum(s);
                if (tl == 0) p[k + 1 + r] = t * s;
            }
            if (IN_LDS) lds_barrier(); else __syncthreads();
            double pv = 0.0;                             // p . v, by every wave for itself
            for (int i = k + 1 + lane; i < n; i += 64) pv = fma(p[i], i == k + 1 ? 1.0 : rowk[i] * inv, pv);
            const double K = 0.5 * t * wave_sum(pv);
            // A22 -= v w^T + w v^T,  w = p - K v
            for (int r = team; r < m; r += TRI_THREADS / TRI_TEAM) {
                const int i = k + 1 + r;
                double *row = A + (size_t)i * ld;
                const double vi = r == 0 ? 1.0 : rowk[i] * inv, wi = fma(-K, vi, p[i]);
#pragma unroll 4
                for (int j = k + 1 + tl; j < n; j += TRI_TEAM) {
                    const double vj = j == k + 1 ? 1.0 : rowk[j] * inv, wj = fma(-K, vj, p[j]);
                    row[j] -= fma(vi, wj, wi * vj);
                }
                // (LDS: the DS unit serves a wave's operations in order, so the team sees its own lanes' stores)
                if (IN_LDS && r == 0) leave_row_scalars(row, k + 1);
            }
        } else if (IN_LDS && team == 0) leave_row_scalars(A + (size_t)(k + 1) * ld, k + 1);
        if (IN_LDS) lds_barrier(); else __syncthreads();
        if (!IN_LDS) {                                    // matrix in global memory: the row is read back behind the barrier
            if (team == 0) leave_row_scalars(A + (size_t)(k + 1) * ld, k + 1);
            __syncthreads();
        }
    }
    if (tid == 0) {
        const double el = A[(size_t)(n - 2) * ld + n - 1];
        d[n - 2] = A[(size_t)(n - 2) * ld + n - 2]; e[n - 2] = el; tau[n - 2] = 0.0; tri[3 * n + n - 2] = el * el;
        d[n - 1] = A[(size_t)(n - 1) * ld + n - 1]; e[n - 1] = 0.0; tau[n - 1] = 0.0; tri[3 * n + n - 1] = 0.0;
    }
}

// ---- 1b. the same with the matrix in REGISTERS: team r (TEAM lanes) owns row r, lane l of it the column pairs (2 l, 2 l + 1) + 2 TEAM u,
// u < TR_U.  The LDS kernel above moves the trailing block through the LDS port three times per step (48 m^2 bytes) and waits for every one
// of those loads; here a step touches LDS only for vectors (the row being eliminated, v, p; double-buffered so that two barriers per step
// suffice), and the two O(n^2) phases are unpredicated FMAs on registers: columns that are already finished meet v_j = 0 (their stale
// register contents stay finite - each step adds a bounded multiple of p_j).  What limits a step is the instruction count of the waves on
// the CU's four SIMDs, hence the bare loops.  Instances (2 TEAM TR_U columns, as many rows, TEAM threads each):
//   <8, 4>  n <=  64,  512 threads: half the FMAs of a step and half the waves at its two barriers (H2O/cc-pVTZ, n = 58)
//   <8, 8>  n <= 128, 1024 threads
//   <4, 24> n <= 192,  768 threads, 48 matrix elements per lane (these sizes ran from global memory before: 2.2 ms at n = 174)
template <int TEAM> __device__ __forceinline__ double team_sum_t(double v);
template <> __device__ __forceinline__ double team_sum_t<8>(double v) { return team_sum(v); }
template <> __device__ __forceinline__ double team_sum_t<4>(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    return v;
}
template <int TEAM, int TR_U>
__global__ __launch_bounds__(2 * TEAM * TR_U * TEAM) void qc_tridiag_reg_kernel(int n, const double *__restrict__ Ain, double rel_pert, double *__restrict__ Vr,
                                                                                 double *__restrict__ tri) {
    constexpr int NCOL = 2 * TEAM * TR_U, THREADS = NCOL * TEAM, RPW = 64 / TEAM, NCH = (NCOL + 63) / 64;   // rows per wave; 64-column chunks of a row
    __shared__ double2 vbuf[2][NCH * 32], pbuf[2][NCH * 32];
    __shared__ double rowbuf[NCH * 64], sc[2][4], red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = tid & (TEAM - 1), row = tid / TEAM;
    double *d = tri, *e = tri + n, *tau = tri + 2 * n;
    double2 a[TR_U];
    double amax = 0.0;
    const double pert_unit = hash_unit((unsigned long long)row);
#pragma unroll
    for (int u = 0; u < TR_U; ++u) {
        const int j = 2 * tl + 2 * TEAM * u;
        a[u].x = (row < n && j < n) ? 0.5 * (Ain[(size_t)row * n + j] + Ain[(size_t)j * n + row]) : 0.0;
        a[u].y = (row < n && j + 1 < n) ? 0.5 * (Ain[(size_t)row * n + j + 1] + Ain[(size_t)(j + 1) * n + row]) : 0.0;
        amax = fmax(amax, fmax(fabs(a[u].x), fabs(a[u].y)));
        if constexpr (TR_U > 8) { if (u % 4 == 3) __builtin_amdgcn_sched_barrier(0); }      // (four loads per pair: not all 96 at once)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) red[wave] = amax;
    for (int i = tid; i < NCH * 32; i += THREADS) { vbuf[0][i] = vbuf[1][i] = pbuf[0][i] = pbuf[1][i] = make_double2(0.0, 0.0); }
    __syncthreads();
    amax = 0.0;
    for (int k = 0; k < THREADS / 64; ++k) amax = fmax(amax, red[k]);
    const double pert = rel_pert * amax * pert_unit;
#pragma unroll
    for (int u = 0; u < TR_U; ++u) {
        const int j = 2 * tl + 2 * TEAM * u;
        a[u].x += j == row ? pert : 0.0;
        a[u].y += j + 1 == row ? pert : 0.0;
    }

    for (int k = 0; k + 2 < n; ++k) {
        double2 *vb = vbuf[k & 1], *pb = pbuf[k & 1];
        double *scb = sc[k & 1];
        if (wave == k / RPW) {                           // (wave-uniform) the wave that holds row k builds the Householder vector
            if (row == k) {
#pragma unroll
                for (int u = 0; u < TR_U; ++u) reinterpret_cast<double2 *>(rowbuf)[tl + TEAM * u] = a[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same wave: the DS unit serves its operations in order
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double x[NCH], part = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { x[c] = rowbuf[lane + 64 * c]; part += lane + 64 * c > k + 1 ? x[c] * x[c] : 0.0; }
            const double s2 = wave_sum(part);
            const double x0 = rowbuf[k + 1], dk = rowbuf[k];
            double beta = x0, t = 0.0, inv = 0.0;
            if (s2 > 0.0) {
                const double nn2 = fma(x0, x0, s2);
                double r = __builtin_amdgcn_rsq(nn2);               // 1 / ||x||, two Newton steps
                r = r * fma(-0.5 * nn2 * r, r, 1.5);
                r = r * fma(-0.5 * nn2 * r, r, 1.5);
                const double mu = nn2 * r;
                beta = x0 <= 0.0 ? mu : -mu;
                t = (beta - x0) * (x0 <= 0.0 ? r : -r);             // (beta - x0) / beta
                inv = fast_rcp(x0 - beta);
            }
            double *vrow = Vr + (size_t)k * n;           // (for the back-transformation kernel; nothing here waits for these stores)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int i = lane + 64 * c;
                const double vv = i <= k ? 0.0 : (i == k + 1 ? 1.0 : x[c] * inv);
                reinterpret_cast<double *>(vb)[i] = vv;
                if (i < n) vrow[i] = vv;
            }
            if (lane == 0) { scb[0] = t; d[k] = dk; e[k] = beta; tau[k] = t; tri[3 * n + k] = beta * beta; }
        }
        lds_barrier();
        const double t = scb[0];
        const bool busy = t != 0.0 && RPW * wave + RPW - 1 > k;   // (wave-uniform: some row of this wave is still in the trailing block)
        constexpr bool KEEPV = TR_U <= 8;                 // v_j stays in registers between the phases, or is read again (registers)
        double2 v[KEEPV ? TR_U : 1];
        double y = 0.0;
        if (busy) {
#pragma unroll
            for (int u = 0; u < TR_U; ++u) {
                if (2 * TEAM * u + 2 * TEAM - 1 > k) {        // (uniform skip of finished columns)
                    const double2 vv = vb[tl + TEAM * u];
                    if constexpr (KEEPV) v[u] = vv;
                    y = fma(a[u].x, vv.x, fma(a[u].y, vv.y, y));
                } else if constexpr (KEEPV) v[u] = make_double2(0.0, 0.0);
                if constexpr (!KEEPV) { if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0); }   // (keeps the loads of 24 column pairs from being hoisted into 96 more registers)
            }
            y = t * team_sum_t<TEAM>(y);
            if (tl == 0 && row > k) reinterpret_cast<double *>(pb)[row] = y;
        }
        lds_barrier();
        if (busy) {
            double pv = 0.0;
#pragma unroll
            for (int u = 0; u < TR_U; ++u)
                if (2 * TEAM * u + 2 * TEAM - 1 > k) {
                    const double2 p2 = pb[tl + TEAM * u], vv = KEEPV ? v[KEEPV ? u : 0] : vb[tl + TEAM * u];
                    pv = fma(p2.x, vv.x, fma(p2.y, vv.y, pv));
                    if constexpr (!KEEPV) { if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0); }
                }
            const double K = 0.5 * t * team_sum_t<TEAM>(pv);
            const double vi = reinterpret_cast<double *>(vb)[row], wi = fma(-K, vi, y);
            if (row > k) {
#pragma unroll
                for (int u = 0; u < TR_U; ++u)
                    if (2 * TEAM * u + 2 * TEAM - 1 > k) {                // (p_j is read a second time rather than kept: registers)
                        const double2 p2 = pb[tl + TEAM * u], vv = KEEPV ? v[KEEPV ? u : 0] : vb[tl + TEAM * u];
                        a[u].x -= fma(vi, fma(-K, vv.x, p2.x), wi * vv.x);
                        a[u].y -= fma(vi, fma(-K, vv.y, p2.y), wi * vv.y);
                        if constexpr (!KEEPV) { if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0); }
                    }
            }
        }
    }
    // last 2 x 2 block
    if (row == n - 2 || row == n - 1) {
        double dk = 0.0, ek = 0.0;
#pragma unroll
        for (int u = 0; u < TR_U; ++u) {
            const int j = 2 * tl + 2 * TEAM * u;
            dk += (j == row ? a[u].x : 0.0) + (j + 1 == row ? a[u].y : 0.0);
            ek += (j == row + 1 ? a[u].x : 0.0) + (j + 1 == row + 1 ? a[u].y : 0.0);
        }
        dk = team_sum_t<TEAM>(dk); ek = team_sum_t<TEAM>(ek);
        if (tl == 0) { d[row] = dk; e[row] = row == n - 1 ? 0.0 : ek; tau[row] = 0.0; tri[3 * n + row] = row == n - 1 ? 0.0 : ek * ek; }
    }
}

// number of eigenvalues of T below x: sign changes of the Sturm sequence p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}, without divisions.
// d / e2 sit in LDS and are read with wave-uniform addresses (broadcast reads, eight steps requested at a time: as scalar loads from
// global memory every chunk waited ~500 cycles for its operands - SMEM returns out of order, so nothing could be kept in flight
// across the wait - 720 cycles per chunk against 300 now); rescaled every 8 steps.
__device__ __forceinline__ int sturm_count(int n, const double *__restrict__ d, const double *__restrict__ e2, double x) {
    double q = 1.0, p = d[0] - x;
    int cnt = p < 0.0 ? 1 : 0;
    int i = 1;
    for (; i + 8 <= n; i += 8) {
        double dd[8], ee[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { dd[u] = d[i + u]; ee[u] = e2[i + u - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // (a sign change = the sign bits differ: one xor, one shift, one add on the high words - as two compares, a mask xor and a
            // conditional add the count was half of the loop's 17 instructions per step, and the loop is bound by issue)
            const double pn = fma(dd[u] - x, p, -(ee[u] * q));
            cnt += (int)((unsigned)(__double2hiint(pn) ^ __double2hiint(p)) >> 31);
            q = p; p = pn;
        }
        const double s = fmax(fabs(p), fabs(q));
        if (s > 0x1p300) { p *= 0x1p-300; q *= 0x1p-300; }
        else if (s < 0x1p-300 && s > 0.0) { p *= 0x1p300; q *= 0x1p300; }
    }
    for (; i < n; ++i) {
        const double pn = fma(d[i] - x, p, -(e2[i - 1] * q));
        cnt += (int)((unsigned)(__double2hiint(pn) ^ __double2hiint(p)) >> 31);
        q = p; p = pn;
    }
    return cnt;
}

// ---- 2. eigenpairs of the tridiagonal matrix, TE_WAVES of them per workgroup.  Z[i * n + j] = component i of the (unnormalised)
// eigenvector of the j-th eigenvalue (ascending); out = [lambda (n) | 1 / ||z_j|| (n)].  tri = [d | e | tau | e^2].
//   eigenvalue j: one wave, 65-way multisection on Sturm counts (9 rounds: the interval shrinks 65-fold per round, the lanes'
//   verdicts meet in one ballot);
//   eigenvector j: twisted factorisation (Fernando; Parlett & Dhillon) by lanes 0 and 1 of that wave - forward pivots D+ (lane 0)
//   and backward pivots D- (lane 1) are independent recurrences and run side by side into two LDS arrays; twist index
//   r = argmin |D+_k + D-_k - (d_k - lambda)|; then z_r = 1 with the downward (lane 0) and upward (lane 1) one-term recurrences.
constexpr int TE_WAVES = 4;
__global__ __launch_bounds__(TE_WAVES * 64) void qc_tri_eig_kernel(int n, const double *__restrict__ tri, double *__restrict__ Zg, double *__restrict__ out) {
    extern __shared__ double sm[];                       // d | e | e^2 (n each, shared by the workgroup), then per wave: D+ (n) | D- (n)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * TE_WAVES + wave;
    double *d = sm, *e = sm + n, *e2 = sm + 2 * n;
    for (int i = threadIdx.x; i < n; i += TE_WAVES * 64) { d[i] = tri[i]; e[i] = tri[n + i]; e2[i] = tri[3 * n + i]; }
    __syncthreads();
    if (j >= n) return;
    double *Dp = sm + 3 * n + (size_t)wave * 2 * n, *Dm = Dp + n;
    // Gershgorin interval
    double lo = 1e300, hi = -1e300;
    for (int i = lane; i < n; i += 64) {
        const double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
        lo = fmin(lo, d[i] - r); hi = fmax(hi, d[i] + r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    { const double w = fmax(hi - lo, 1e-300); lo -= 1e-3 * w; hi += 1e-3 * w; }
    const double tiny = 1e-300 + 0x1p-52 * 0x1p-52 * fmax(fabs(lo), fabs(hi));
    // Nine rounds: the interval shrinks to 65^-9 = 5e-17 of the Gershgorin range.  (Eight - 3e-15 - are not enough: the twisted
    // factorisation needs the eigenvalue to well below the smallest gap, ~1e-10 of the scale for pairs that are only split by the
    // perturbation, and with eight rounds the doubly degenerate test spectra at n = 114 came out parallel enough to be rejected.)
    for (int round = 0; round < 9; ++round) {
        const double h = (hi - lo) * (1.0 / 65.0);
        const double x = lo + h * (lane + 1);
        const unsigned long long above = __ballot(sturm_count(n, d, e2, x) > j);     // lanes whose point lies above eigenvalue j
        const int first = above ? __builtin_ctzll(above) : 64;
        const double nlo = lo + h * first, nhi = first == 64 ? hi : lo + h * (first + 1);
        lo = nlo; hi = nhi;
        if (!(hi - lo > 0x1p-52 * fmax(fabs(lo), fabs(hi)))) break;                  // (wave-uniform)
    }
    const double l = 0.5 * (lo + hi);
    auto guard = [&](double x) { return fabs(x) < tiny ? tiny : x; };
    // The two pivot recurrences, D+_i = (d_i - l) - e_{i-1}^2 / D+_{i-1} upwards (lane 0) and D-_i = (d_i - l) - e_i^2 / D-_{i+1} downwards
    // (lane 1), as ONE instruction stream: in two branches the wave would run them one after the other.  Operands of four steps are
    // requested ahead of the dependent chain (rcp + Newton + fma per step).
    if (lane < 2) {
        const int up = lane == 0;
        double *arr = up ? Dp : Dm;
        const int i0 = up ? 0 : n - 1;
        double piv = guard(d[i0] - l);
        arr[i0] = piv;
        int s = 1;
        for (; s + 4 <= n; s += 4) {
            int ii[4]; double dv[4], fv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ii[u] = up ? s + u : n - 1 - s - u; dv[u] = d[ii[u]]; fv[u] = e2[up ? ii[u] - 1 : ii[u]]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { piv = guard(fma(-fv[u], fast_rcp(piv), dv[u] - l)); arr[ii[u]] = piv; }
        }
        for (; s < n; ++s) { const int i = up ? s : n - 1 - s; piv = guard(fma(-e2[up ? i - 1 : i], fast_rcp(piv), d[i] - l)); arr[i] = piv; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double gmin = 1e300;
    int r = 0;
    for (int i = lane; i < n; i += 64) {
        const double g = fabs(Dp[i] + Dm[i] - (d[i] - l));
        if (g < gmin) { gmin = g; r = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double g2 = __shfl_xor(gmin, o, 64);
        const int r2 = __shfl_xor(r, o, 64);
        if (g2 < gmin || (g2 == gmin && r2 < r)) { gmin = g2; r = r2; }
    }
    // The multipliers -(e_i / D+_i) and -(e_i / D-_{i+1}) do not depend on z: all lanes form them at once (in place of the pivots),
    // the two serial recurrences z_i = m_i z_{i+1} (lane 0, downwards from r) and z_{i+1} = m_i z_i (lane 1, upwards) are then bare
    // multiplications - again one instruction stream for both lanes.
    for (int i = lane; i < n; i += 64) {
        if (i < r) Dp[i] = -(e[i] * fast_rcp(Dp[i]));
        if (i >= r && i + 1 < n) Dm[i + 1] = -(e[i] * fast_rcp(Dm[i + 1]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double nrm = 0.0;
    if (lane < 2) {
        const int down = lane == 0;
        const int steps = down ? r : n - 1 - r;           // lane 0: i = r-1 .. 0;  lane 1: i = r+1 .. n-1
        double zi = 1.0;
        if (down) { nrm = 1.0; Zg[(size_t)r * n + j] = 1.0; }
        const double *mul = down ? Dp : Dm;
        for (int s = 1; s <= steps; ++s) {
            const int i = down ? r - s : r + s;
            zi *= mul[i];
            Zg[(size_t)i * n + j] = zi;
            nrm = fma(zi, zi, nrm);
        }
    }
    nrm += __shfl(nrm, 1, 64);
    if (lane == 0) { out[j] = l; out[n + j] = rsqrt(nrm); }
}

// ---- 3. X0 = Q Z, Q = H_0 H_1 ... H_{n-3}: one wave per eigenvector, its components in registers (NPL per lane).  The reflector rows come
// through LDS: the workgroup (BT_WAVES eigenvectors) loads a chunk of rows with all its threads, eight loads in flight per thread, and
// every wave then walks through the chunk.  (With the rows read from L2 one step ahead, a step - a dot product, a wave reduction, an
// axpy - waited ~0.4 us for its reflector: 24 / 34 us at n = 58 / 114.)
constexpr int BT_WAVES = 8;
constexpr int BT_LDS_BYTES = 120 * 1024;
template <int NPL>
__global__ __launch_bounds__(BT_WAVES * 64) void qc_backtransform_kernel(int n, int rows_per_chunk, const double *__restrict__ Zg, const double *__restrict__ Vr,
                                                                         const double *__restrict__ tri, const double *__restrict__ evn, double *__restrict__ X0) {
    extern __shared__ double bt_sh[];                      // [rows_per_chunk][n] reflector rows, then [rows_per_chunk] tau
    constexpr int NT = BT_WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, j = blockIdx.x * BT_WAVES + (tid >> 6);
    const bool act = j < n;                                // (no early exit: the chunk loop has barriers)
    const double *tau = tri + 2 * n;
    double *const sh_tau = bt_sh + (size_t)rows_per_chunk * n;
    const double sc = act ? evn[n + j] : 0.0;
    double x[NPL];
#pragma unroll
    for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; x[u] = (act && i < n) ? Zg[(size_t)i * n + j] * sc : 0.0; }
    for (int khi = n - 3; khi >= 0; khi -= rows_per_chunk) {
        const int klo = max(0, khi - rows_per_chunk + 1), cnt = khi - klo + 1, total = cnt * n;
        const double *__restrict__ src = Vr + (size_t)klo * n;
        __syncthreads();                                   // the previous chunk's readers are done
        for (int base = 0; base < total; base += 8 * NT) {
            double tmp[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) { const int i = base + b * NT + tid; tmp[b] = i < total ? src[i] : 0.0; }
#pragma unroll
            for (int b = 0; b < 8; ++b) { const int i = base + b * NT + tid; if (i < total) bt_sh[i] = tmp[b]; }
        }
        for (int i = tid; i < cnt; i += NT) sh_tau[i] = tau[klo + i];
        __syncthreads();
        if (act) {
            double vn[NPL];
#pragma unroll
            for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; vn[u] = i < n ? bt_sh[(size_t)(khi - klo) * n + i] : 0.0; }
            for (int k = khi; k >= klo; --k) {
                double vv[NPL];
#pragma unroll
                for (int u = 0; u < NPL; ++u) vv[u] = vn[u];
                if (k > klo) {
#pragma unroll
                    for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; vn[u] = i < n ? bt_sh[(size_t)(k - 1 - klo) * n + i] : 0.0; }   // next reflector, one step ahead
                }
                const double t = sh_tau[k - klo];
                double s = 0.0;
#pragma unroll
                for (int u = 0; u < NPL; ++u) s = fma(vv[u], x[u], s);
                s = t * wave_sum(s);
#pragma unroll
                for (int u = 0; u < NPL; ++u) x[u] = fma(-s, vv[u], x[u]);
            }
        }
    }
    if (act) {
#pragma unroll
        for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; if (i < n) X0[(size_t)i * n + j] = x[u]; }
    }
}

}  // namespace

size_t qc_eig_tridiag_work_doubles(int n) { return 2 * (size_t)n * n + 8 * (size_t)n + 16; }

// Approximate eigenvectors (columns of dX0, ascending eigenvalues, orthonormal to ~1e-6 for generic matrices) of the symmetric dA,
// which is left intact.  work: qc_eig_tridiag_work_doubles(n) doubles.  Asynchronous on `st`.
int qc_eig_tridiag_start(hipStream_t st, int n, const double *dA, double *dX0, double *work) {
    if (!qc_tri_ok(n)) return QC_ERR_UNSUPPORTED;
    double *Vr = work, *Zg = work + (size_t)n * n, *tri = Zg + (size_t)n * n, *evn = tri + 4 * (size_t)n;
    static std::atomic<bool> raised{false};
    if (!raised.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(qc_tridiag_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QC_LDS_MAX) != hipSuccess) return QC_ERR_HIP;
        raised.store(true, std::memory_order_release);
    }
    static const double pert_env = getenv("QC_TRI_PERT") ? atof(getenv("QC_TRI_PERT")) : 0.0;      // A/B switch
    const double rel_pert = pert_env > 0.0 ? pert_env : 1e-11;
    const size_t small = ((size_t)n + 32) * sizeof(double);
    // row stride: 8 mod 32 doubles, so that the eight rows a wave reads at a time fall into disjoint LDS banks (when that still fits)
    int ld = n + ((8 - n % 32) + 32) % 32;
    if ((size_t)n * ld * sizeof(double) + small > (size_t)QC_LDS_MAX - 1024) ld = n;
    const size_t lds_a = (size_t)n * ld * sizeof(double) + small;
    static const bool no_reg = getenv("QC_TRI_LDS") != nullptr;       // A/B switch
    if (n <= 64 && !no_reg)
        hipLaunchKernelGGL((qc_tridiag_reg_kernel<8, 4>), dim3(1), dim3(512), 0, st, n, dA, rel_pert, Vr, tri);
    else if (n <= 128 && !no_reg)
        hipLaunchKernelGGL((qc_tridiag_reg_kernel<8, 8>), dim3(1), dim3(1024), 0, st, n, dA, rel_pert, Vr, tri);
    else if (n <= 192 && !no_reg)
        hipLaunchKernelGGL((qc_tridiag_reg_kernel<4, 24>), dim3(1), dim3(768), 0, st, n, dA, rel_pert, Vr, tri);
    else if (lds_a <= (size_t)QC_LDS_MAX - 1024)
        hipLaunchKernelGGL(qc_tridiag_kernel<true>, dim3(1), dim3(TRI_THREADS), lds_a, st, n, ld, dA, rel_pert, (double *)nullptr, Vr, tri);
    else        // matrix in global memory (Zg's storage is free until the next kernel)
        hipLaunchKernelGGL(qc_tridiag_kernel<false>, dim3(1), dim3(TRI_THREADS), small, st, n, n, dA, rel_pert, Zg, Vr, tri);
    hipLaunchKernelGGL(qc_tri_eig_kernel, dim3((n + TE_WAVES - 1) / TE_WAVES), dim3(TE_WAVES * 64), (size_t)(TE_WAVES * 2 + 3) * n * sizeof(double), st, n, tri, Zg, evn);
    const int npl = (n + 63) / 64;
    const dim3 grid((n + BT_WAVES - 1) / BT_WAVES), block(BT_WAVES * 64);
    const int rpc = std::max(1, std::min(n - 2, (int)((BT_LDS_BYTES - 8 * n) / (8 * (size_t)n + 8))));     // reflector rows per LDS chunk
    const size_t bt_lds = ((size_t)rpc * n + rpc) * sizeof(double);
    {
        static std::atomic<bool> raised{false};             // (process-wide attribute; the flag only saves the repeated call)
        if (!raised.load(std::memory_order_acquire)) {
#define QC_BT_ATTR(N) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(qc_backtransform_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS_BYTES);
            QC_BT_ATTR(1) QC_BT_ATTR(2) QC_BT_ATTR(3) QC_BT_ATTR(4) QC_BT_ATTR(5) QC_BT_ATTR(6) QC_BT_ATTR(7) QC_BT_ATTR(8)
#undef QC_BT_ATTR
            raised.store(true, std::memory_order_release);
        }
    }
#define QC_BT_CASE(N) case N: hipLaunchKernelGGL(qc_backtransform_kernel<N>, grid, block, bt_lds, st, n, rpc, Zg, Vr, tri, evn, dX0); break;
    switch (npl) { QC_BT_CASE(1) QC_BT_CASE(2) QC_BT_CASE(3) QC_BT_CASE(4) QC_BT_CASE(5) QC_BT_CASE(6) QC_BT_CASE(7) QC_BT_CASE(8) default: return QC_ERR_UNSUPPORTED; }
#undef QC_BT_CASE
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}
