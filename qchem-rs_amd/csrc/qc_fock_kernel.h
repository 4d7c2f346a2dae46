// qc_fock_kernel.h - the direct-SCF ERI + Fock digestion kernel for one quartet class (LAB | LCD), gfx950 / wave64.
//
// Replaces molint::eri (rhf.rs:45) fused with compute_electronic_hamiltonian (rhf.rs:152-167, uhf.rs:210-227): every
// unique shell quartet (ab|cd) is evaluated by McMurchie-Davidson in the *Hermite-matrix* form
//        (ab|cd) = sum_{ij in ab} sum_{kl in cd}  E_ab,ij^T  .  Rmat(p_ij, q_kl)  .  E_cd,kl
// (E = per-primitive-pair Hermite expansion matrices prepared once per geometry, spherical transform and contraction
// coefficients folded in; Rmat[h1][h2] = (-1)^{|h2|} R_{h1+h2}) and immediately contracted with the density into the
// six J/K blocks.  f64 throughout.
//
// Work unit = "slot": one shell quartet (bra pair | ket pair) restricted to a range of its primitive quartets and,
// for the small matrix-core classes, to one 16-column tile of its ket block (deeply contracted quartets are cut into
// several slots so no lane group runs a long sequential loop; digestion is linear, every slot digests its own partial block).
// Mapping (one wave = one workgroup, grid-stride over batches of G slots of one launch bucket (LAB, LCD, C)):
//   lanes = G groups x C columns, C = 16, 32 or 64 >= n_cd (ket function pairs; wider kets take several column passes),
//   G = 64 / C; group g works on slot g of the batch, completely independently of the other groups (own LDS region, own
//   descriptors, own digestion).  A group is made of whole 16-lane rows because
//   * a lane owns one ket column cd and keeps the half-contracted W[HAB] in VGPRs, and both Hermite contractions are
//     `v_fmac_f64_dpp ... row_newbcast` FMAs: the operand common to the group (an R value in step 2, a bra expansion
//     coefficient in step 3) sits in one lane of each row and is broadcast by the instruction itself, no LDS traffic;
//   * R tables: for L <= 6 the lanes of a group evaluate several primitive quartets' tables at once in registers
//     (generated recursion, gen_step2.py) and park them in LDS; above, one table per primitive quartet is built
//     cooperatively; classes with an fd / ff ket and a bra of L >= 3 run both contractions as f64 MFMA tiles instead;
//   * the contracted block I[ab][cd] lives in the group's LDS region, density tiles are staged next to it, and the six
//     J/K block updates are flushed with global_atomic_add_f64 into one of `nrep` replicas.
// (The classes with an ss / ps ket and L <= 4 - one whole quartet per lane - are the bra-major kernels, qc_fock_bm.hip.)
#pragma once
#include "qc_internal.h"
#include <atomic>
#include <utility>

struct QcKernelArgs {
    const QcPairDesc *pairs;
    const double *pairdata;
    const double *pairdataT;  // same blocks, expansion stored [ab][h] (A operand of the MFMA step 3)
    const double *boys;
    const int2 *rplan;        // recurrence plans of the cooperative R tables, all orders (qc_plan_off)
    const uint4 *gidx;        // gather records of the matrix-core classes (qc_gidx_off)
    int n;
    const double *Dj, *Dk0, *Dk1;
    double *G0, *G1;          // replica 0 of the accumulation targets
    size_t rep_stride;        // doubles between replicas
    int nrep;                 // accumulation replicas (spreads global-atomic contention)
    double cK;
    double *eri_out;
    const double *fxs;        // accumulation mode of G0 / G1: non-null = two-limb 64-bit fixed point (order-independent, exact) with
                              // the scale 2^S of this build at fxs[0] (qc_fx_scale_kernel); null = f64 atomics
    size_t fx_lo;             // doubles from an element of the hi plane to the same element of the lo plane
    double *schwarz_out;      // if non-null: no digestion - the slots are (P|P) quartets and sqrt(max |(ab|cd)|) goes to [P]
    const unsigned *cancel;   // non-null: a speculative build (issued before the host knew that the SCF pass in front of it would not be the
    unsigned cancel_seq;      // last one) - its kernels return at once when *cancel == cancel_seq (qc_spec_release_kernel, qc_scf_small.hip)
    unsigned long long *tl;   // non-null (QC_DEV_TIMELINE): QC_TL_W clock words of this launch - [0] start of workgroup 0, [1 + (workgroup & 31)] ends
};

// Device-side timeline of an SCF pass (QC_DEV_TIMELINE=1): every kernel of the pass leaves the clock of its first start and its last end -
// what rocprofv3's kernel trace shows too, but without the profiler's own packets between the dispatches.
__device__ __forceinline__ void qc_tl_stamp(unsigned long long *tl, int end) {
    if (tl != nullptr && threadIdx.x == 0 && (end || blockIdx.x == 0)) {
        const unsigned long long t = wall_clock64();
        if (end) __hip_atomic_store(tl + 1 + (blockIdx.x & 31), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(tl, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (every class kernel starts with this: one scalar load)
__device__ __forceinline__ bool qc_build_cancelled(const QcKernelArgs &a) {
    return a.cancel != nullptr && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.cancel_seq;
}

// ---- order-independent, exact accumulation.  The digestion adds ~10^3 contributions from different waves into every
// element of Gt; with f64 atomics the sum depends on the order the memory system happens to serve them in, so two builds
// from the same density differ in the last bits - and the reference, which evaluates both spins (and every build) with one
// fixed sequence of operations (uhf.rs:80-108, 210-227), never sees such noise.  In fixed-point mode a contribution v is
// split into two 64-bit integers, hi = rint(v 2^S) and lo = rint((v 2^S - hi) 2^32), which are added to two accumulator
// planes with integer atomics (global_atomic_add_x2).  Integer addition is associative: the result is the same whatever the
// order, the replica, the stream assignment or the number of ranks (the all-reduce adds integers too).  S comes from a
// rigorous bound on |Gt| (qc_fx_scale, qc_linalg.hip), so the hi plane cannot leave 62 bits; the lo plane carries 32 more
// bits, unit 2^-(S+32) Eh - 3e-23 for benzene - so the sum is exact to far below the rounding of a single contribution:
// more accurate than f64 accumulation, not less, also when a near-singular overlap matrix blows the density up to 1e6.
__device__ __forceinline__ void qc_fx2(double v, double scale, unsigned long long &hi, unsigned long long &lo) {
    const double t = v * scale;                                             // exact (power of two)
    const double th = __builtin_rint(t);                                    // integer-valued, |th| < 2^62
    const double tl = __builtin_rint((t - th) * 0x1p32);                    // t - th exact, |tl| <= 2^31
    const double h1 = __builtin_floor(th * 0x1p-32);
    const double h0 = __builtin_fma(h1, -0x1p32, th);                       // exact, 0 <= h0 < 2^32
    hi = ((unsigned long long)(unsigned)(int)h1 << 32) | (unsigned long long)(unsigned)h0;
    lo = (unsigned long long)(long long)(int)tl;                            // (+2^31 saturates one unit of 2^-(S+32) short: harmless)
}
// p: element of the hi plane; the lo plane lies lo_off doubles behind it
__device__ __forceinline__ void qc_gadd(double *p, double v, double fxscale, size_t lo_off) {
    if (fxscale != 0.0) {
        unsigned long long hi, lo;
        qc_fx2(v, fxscale, hi, lo);
        atomicAdd(reinterpret_cast<unsigned long long *>(p), hi);
        atomicAdd(reinterpret_cast<unsigned long long *>(p + lo_off), lo);
    } else unsafeAtomicAdd(p, v);
}
// the same value into the two spins' matrices (Coulomb terms): one conversion
__device__ __forceinline__ void qc_gadd2(double *p0, double *p1, bool two, double v, double fxscale, size_t lo_off) {
    if (fxscale != 0.0) {
        unsigned long long hi, lo;
        qc_fx2(v, fxscale, hi, lo);
        atomicAdd(reinterpret_cast<unsigned long long *>(p0), hi);
        atomicAdd(reinterpret_cast<unsigned long long *>(p0 + lo_off), lo);
        if (two) {
            atomicAdd(reinterpret_cast<unsigned long long *>(p1), hi);
            atomicAdd(reinterpret_cast<unsigned long long *>(p1 + lo_off), lo);
        }
    } else {
        unsafeAtomicAdd(p0, v);
        if (two) unsafeAtomicAdd(p1, v);
    }
}

template <int L>
__device__ __forceinline__ void qc_rtab(double alpha, double X, double Y, double Z, const double (&F)[L + 1], double (&R)[qc_nherm(L)]);

// (t,u,v) of every Hermite index up to order QC_LTOT
struct QcTuvTable { unsigned char t[qc_nherm(QC_LTOT)], u[qc_nherm(QC_LTOT)], v[qc_nherm(QC_LTOT)]; };
__host__ __device__ constexpr QcTuvTable qc_make_tuv() {
    QcTuvTable T{};
    int k = 0;
    for (int N = 0; N <= QC_LTOT; ++N)
        for (int t = N; t >= 0; --t)
            for (int u = N - t; u >= 0; --u) { T.t[k] = (unsigned char)t; T.u[k] = (unsigned char)u; T.v[k] = (unsigned char)(N - t - u); ++k; }
    return T;
}

// Step 2 with the R table spread over the registers of each 16-lane row (groups of 16, 32 or 64 lanes: LGC >= 4).
// Lane l keeps R[16 k + (l & 15)] in Rd[k]; `v_fmac_f64_dpp ... row_newbcast:j` multiplies lane j's copy into every
// lane of the row at the full FP64 rate, so an FMA costs no LDS read at all (the rolled form above pays one
// 512-byte ds_read per FMA, and the four SIMDs of a CU share one 128 B/clk LDS port).  Fully unrolled: the R index
// of every (h1, h2) is a compile-time constant; the ket coefficients e[h2] arrive in batches, one batch ahead.
template <int K>
__device__ __forceinline__ void qc_fmac_bc(double &acc, double r, double e) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(r), "v"(e), "n"(K));
}
template <int H1, int H2>
__host__ __device__ constexpr int qc_ridx() {
    constexpr QcTuvTable T = qc_make_tuv();
    return qc_hidx(T.t[H1] + T.t[H2], T.u[H1] + T.u[H2], T.v[H1] + T.v[H2]);
}
template <int H>
__host__ __device__ constexpr int qc_order_of() {
    constexpr QcTuvTable T = qc_make_tuv();
    return T.t[H] + T.u[H] + T.v[H];
}
template <int LAB, int NR, int H2, int... H1>
__device__ __forceinline__ void qc_step2_dpp_row(double (&W)[qc_nherm(LAB)], const double (&Rd)[NR], double e,
                                                 std::integer_sequence<int, H1...>) {
    (qc_fmac_bc<(qc_ridx<H1, H2>() & 15)>(W[H1], Rd[qc_ridx<H1, H2>() >> 4], e), ...);
}
// 1 / sqrt(x) for finite x > 0: hardware estimate + one third-order correction (the library's rsqrt without its class check and selects:
// 6 instructions instead of 10, once or twice per primitive quartet in every kernel)
__device__ __forceinline__ double qc_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
typedef __attribute__((address_space(3))) double qc_lds_f64;
// (adds of ONE wave to its own LDS words: the DS unit serves the lanes of an instruction in a fixed order, instructions in program order,
// so such sums do not depend on timing)
__device__ __forceinline__ void qc_ds_add(double *p, double v) { (void)__builtin_amdgcn_ds_atomic_fadd_f64((qc_lds_f64 *)p, v); }
// floor(x / d) for 0 <= x < 2^16, d <= 128 from inv = 1 / d (1 ulp): three instructions instead of the ~35 of a u32 division
__device__ __forceinline__ int qc_fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// acc[j] += sum_h E_j[h] W[h] for four rows at once; row j's coefficients sit in the lanes of each 16-lane row (Er[j][h >> 4], lane h & 15)
template <int HAB, int NRE, int... H>
__device__ __forceinline__ void qc_dot4_bc(double (&acc)[4], const double (&Er)[4][NRE], const double (&W)[HAB], std::integer_sequence<int, H...>) {
    ((qc_fmac_bc<(H & 15)>(acc[0], Er[0][H >> 4], W[H]), qc_fmac_bc<(H & 15)>(acc[1], Er[1][H >> 4], W[H]),
      qc_fmac_bc<(H & 15)>(acc[2], Er[2][H >> 4], W[H]), qc_fmac_bc<(H & 15)>(acc[3], Er[3][H >> 4], W[H])), ...);
}
// ket coefficients of step 2 arrive in batches of 8, one batch ahead of the FMAs that use them.  (16 per batch for the d.d ... f.f kets -
// 35 / 56 / 84 coefficients per column, where a batch's FMAs are shorter than the next batch's trip to L2 - was measured in round 3,
// -DQC_DPPB_HI=16: H2O build 0.233 -> 0.252 ms, benzene 1.635 -> 1.652: the registers cost more than the latency gives.)
#ifndef QC_DPPB_HI
#define QC_DPPB_HI 8
#endif
__host__ __device__ constexpr int qc_dpp_batch(int LCD) { return LCD >= 4 ? QC_DPPB_HI : 8; }
template <int LAB, int NR, int H0, int NBT, int... J>
__device__ __forceinline__ void qc_step2_dpp_batch(double (&W)[qc_nherm(LAB)], const double (&Rd)[NR], const double (&e)[NBT],
                                                   double sc, std::integer_sequence<int, J...>) {
    (qc_step2_dpp_row<LAB, NR, H0 + J>(W, Rd, e[J] * ((qc_order_of<H0 + J>() & 1) ? -sc : sc),
                                       std::make_integer_sequence<int, qc_nherm(LAB)>{}), ...);
}
// Batches H0.. of one primitive quartet.  Before the FMAs of the last batch the first coefficients (eN) and the R
// registers (RdN) of the *next* primitive quartet are requested, so a step never starts by waiting for its operands.
template <int LAB, int LCD, int NR, int H0>
__device__ __forceinline__ void qc_step2_dpp_from(double (&W)[qc_nherm(LAB)], const double (&Rd)[NR], const double (&e)[qc_dpp_batch(LCD)],
                                                  const double *__restrict__ Ecd, int ncd, double sc,
                                                  const double *__restrict__ EcdN, const double *__restrict__ RwN, int l16,
                                                  double (&eN)[qc_dpp_batch(LCD)], double (&RdN)[NR]) {
    constexpr int HCD = qc_nherm(LCD), HR = qc_nherm(LAB + LCD), NB = (HCD - H0 < qc_dpp_batch(LCD)) ? HCD - H0 : qc_dpp_batch(LCD);
    if constexpr (H0 + NB < HCD) {
        double en[qc_dpp_batch(LCD)];
#pragma unroll
        for (int j = 0; j < qc_dpp_batch(LCD); ++j) en[j] = (H0 + NB + j < HCD) ? Ecd[(size_t)(H0 + NB + j) * ncd] : 0.0;
        __builtin_amdgcn_sched_barrier(0);          // the next batch is requested before this one's FMAs
        qc_step2_dpp_batch<LAB, NR, H0, qc_dpp_batch(LCD)>(W, Rd, e, sc, std::make_integer_sequence<int, NB>{});
        qc_step2_dpp_from<LAB, LCD, NR, H0 + NB>(W, Rd, en, Ecd, ncd, sc, EcdN, RwN, l16, eN, RdN);
    } else {
#pragma unroll
        for (int j = 0; j < qc_dpp_batch(LCD); ++j) eN[j] = (j < HCD) ? EcdN[(size_t)j * ncd] : 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) RdN[k] = RwN[min(16 * k + l16, HR - 1)];
        __builtin_amdgcn_sched_barrier(0);
        qc_step2_dpp_batch<LAB, NR, H0, qc_dpp_batch(LCD)>(W, Rd, e, sc, std::make_integer_sequence<int, NB>{});
    }
}
// operands of the first primitive quartet of a run
template <int LAB, int LCD, int NR>
__device__ __forceinline__ void qc_step2_dpp_first(const double *__restrict__ Ecd, int ncd, const double *__restrict__ Rw, int l16,
                                                   double (&e)[qc_dpp_batch(LCD)], double (&Rd)[NR]) {
    constexpr int HCD = qc_nherm(LCD), HR = qc_nherm(LAB + LCD);
#pragma unroll
    for (int j = 0; j < qc_dpp_batch(LCD); ++j) e[j] = (j < HCD) ? Ecd[(size_t)j * ncd] : 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) Rd[k] = Rw[min(16 * k + l16, HR - 1)];
}
// one primitive quartet on its own (cooperative-table classes: the table of the next one does not exist yet)
// (`e` = the first batch of ket coefficients, requested by the caller before it built the table)
template <int LAB, int LCD>
__device__ __forceinline__ void qc_step2_dpp(double (&W)[qc_nherm(LAB)], const double (&e)[qc_dpp_batch(LCD)], const double *__restrict__ Ecd, int ncd, double sc,
                                             const double *__restrict__ Rw, int lane) {
    constexpr int NR = (qc_nherm(LAB + LCD) + 15) / 16, HR = qc_nherm(LAB + LCD);
    double Rd[NR], eN[qc_dpp_batch(LCD)], RdN[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) Rd[k] = Rw[min(16 * k + (lane & 15), HR - 1)];
    qc_step2_dpp_from<LAB, LCD, NR, 0>(W, Rd, e, Ecd, ncd, sc, Ecd, Rw, lane & 15, eN, RdN);
}

// Boys function F_0..F_L at x.  Inside the table: Horner evaluation of the 8-term Taylor expansion of the top order
// about the nearest grid point x_k (rows hold F_{L+j}(x_k) / j!), exp(-x) = exp(-x_k) * exp(x_k - x) from the grid value
// and a degree-8 polynomial (|x_k - x| <= DX/2: remainder 5e-18), downward recursion below.  Beyond the table:
// asymptotic F_0 = sqrt(pi/x)/2 and upward recursion (the exp(-x) term, < 7e-19 there, is kept for the high orders only).
template <int L>
__device__ __forceinline__ void qc_boys(double x, const double *__restrict__ tab, double (&F)[L + 1]) {
    if (x < QC_BOYS_XMAX) {
        const int k = (int)(x * (1.0 / QC_BOYS_DX) + 0.5);
        const double d = k * QC_BOYS_DX - x;
        const double4 *row = reinterpret_cast<const double4 *>(tab + ((size_t)L * QC_BOYS_NGRID + k) * 8);   // 64-byte aligned row
        const double4 lo = row[0], hi = row[1];
        double f = hi.w;
        f = fma(f, d, hi.z);
        f = fma(f, d, hi.y);
        f = fma(f, d, hi.x);
        f = fma(f, d, lo.w);
        f = fma(f, d, lo.z);
        f = fma(f, d, lo.y);
        f = fma(f, d, lo.x);
        F[L] = f;
        if constexpr (L > 0) {
            const double exk = tab[(size_t)(QC_LTOT + 1) * QC_BOYS_NGRID * 8 + k];
            double ed = 1.0 / 40320.0;
            ed = fma(ed, d, 1.0 / 5040.0);
            ed = fma(ed, d, 1.0 / 720.0);
            ed = fma(ed, d, 1.0 / 120.0);
            ed = fma(ed, d, 1.0 / 24.0);
            ed = fma(ed, d, 1.0 / 6.0);
            ed = fma(ed, d, 0.5);
            ed = fma(ed, d, 1.0);
            ed = fma(ed, d, 1.0);
            const double ex = exk * ed, x2 = 2.0 * x;
#pragma unroll
            for (int n = L; n > 0; --n) F[n - 1] = fma(x2, F[n], ex) * (1.0 / (2 * n - 1));
        }
    } else {
        const double t = qc_rsqrt(x);
        F[0] = (0.5 * 1.7724538509055160273) * t;
        if constexpr (L > 0) {
            const double hr = 0.5 * (t * t);
            if constexpr (L <= QC_LREG) {     // exp(-x) < 7e-19: below 1e-13 relative for orders <= 4
#pragma unroll
                for (int n = 0; n < L; ++n) F[n + 1] = ((double)(2 * n + 1) * hr) * F[n];
            } else {                          // high orders are small themselves (F_12(42) ~ 4e-13): keep the term
                const double ex = exp(-x);
#pragma unroll
                for (int n = 0; n < L; ++n) F[n + 1] = fma((double)(2 * n + 1), F[n], -ex) * hr;
            }
        }
    }
}

// offset of level n inside the R work array (levels hold orders 0..L-n)
template <int L>
__host__ __device__ constexpr int qc_roff(int n) {
    int o = 0;
    for (int m = 0; m < n; ++m) o += qc_nherm(L - m);
    return o;
}

// Cooperative Hermite-Coulomb table R^0_{tuv}, t+u+v <= L, by the C lanes of a group (lane-in-group `li`), in LDS.
// On return Rw[0 .. nherm(L)) holds R^0.  Every lane of the workgroup must call this (it contains barriers).
// Stage N = t+u+v computes its (L-N+1)(N+1)(N+2)/2 entries (levels n <= L-N) from stage N-1:
//   R^n_{t,u,v} = g R^{n+1}_{(t,u,v) - e_g} + c R^{n+1}_{(t,u,v) - 2 e_g},  g = the first axis with a non-zero index, c = that index - 1.
// Which entry reads what is the same for every quartet, so it comes from a plan (host-built once per order, qc_build_rplan; staged in
// LDS by the workgroup): a record holds the byte offsets of the target and the two sources, c and the axis - decoding (n,t,u,v) from the
// flat index instead cost ~100 instructions per entry, integer multiplies at a quarter of the rate, 15 us per table for a lone wave at L = 12.
template <int L, int C>
__device__ __forceinline__ void qc_build_r(double *__restrict__ Rw, const int2 *__restrict__ plan, int li, double alpha, double X, double Y, double Z,
                                           const double (&F)[L + 1]) {
    if (li == 0) {
        double f = 1.0;
#pragma unroll
        for (int n = 0; n <= L; ++n) { Rw[qc_roff<L>(n)] = f * F[n]; f *= -2.0 * alpha; }
    }
    __syncthreads();
    char *const Rb = reinterpret_cast<char *>(Rw);
    int base = 0;
#pragma unroll
    for (int N = 1; N <= L; ++N) {
        const int total = (L - N + 1) * ((N + 1) * (N + 2) / 2);
#pragma unroll
        for (int k = 0; k * C < total; ++k) {
            const int e = li + k * C;
            const bool ok = e < total;
            const int2 pe = plan[base + (ok ? e : 0)];
            const int ax = pe.y >> 24;
            const double g = ax == 0 ? X : (ax == 1 ? Y : Z);
            const double r1 = *reinterpret_cast<const double *>(Rb + (pe.x >> 16)), r2 = *reinterpret_cast<const double *>(Rb + (pe.y & 0xffff));
            const double val = fma((double)((pe.y >> 16) & 0xff), r2, g * r1);
            if (ok) *reinterpret_cast<double *>(Rb + (pe.x & 0xffff)) = val;
        }
        base += total;
        __syncthreads();
    }
}

// `run` > 0 (bra-run mode, low-L classes): the slot list is grouped by bra pair - every batch of G slots shares one bra, padded with
// null slots (ket < 0) - and workgroup `blk` works through the `run` consecutive batches [blk run, (blk + 1) run).  With `rb_rows` > 0 the
// wave keeps the targets that belong to the bra - J_ab and the exchange rows of the bra's functions, (na + nb) x n per spin - in an
// LDS row buffer across the batches of one bra and flushes what is non-zero when the bra changes: one global atomic (pair) and one
// fixed-point conversion per touched element and bra run instead of per slot (the memory-side atomics of these classes were a third
// of a benzene build's; cf. the bra-major kernels).  run == 0: slots are independent, grid-stride over batches.
// (MFMA_OK = false: the instance of a launch that never sees f-ket classes - qc_fock_tier_kernel<LAB, 2> - keeps the VALU form of its d.d / f.p-ket
// bodies and with it its two waves per SIMD)
template <int LAB, int LCD, int LGC, bool MFMA_OK = true>
__device__ __forceinline__ void qc_fock_body(const QcKernelArgs &a, const QcSlot *__restrict__ slots, const int nslots, const int slot_words,
                                             const int blk, const int nblk, const int run = 0, const int rb_rows = 0) {
    constexpr int L = LAB + LCD, HAB = qc_nherm(LAB), HCD = qc_nherm(LCD);
    static_assert(LGC >= 4, "a lane group is made of whole 16-lane rows (DPP row broadcasts)");
    constexpr int C = 1 << LGC, G = 64 >> LGC;
    extern __shared__ double lds[];
    // (a 64-lane group is the whole workgroup: group index 0 as a constant keeps the slot and everything derived from it in scalar registers)
    const int lane = threadIdx.x, g = (LGC == 6) ? 0 : lane >> LGC, li = (LGC == 6) ? lane : lane & (C - 1);
    const double *__restrict__ pd = a.pairdata;
    const int n = a.n;
    const bool uhf = a.Dk1 != nullptr;
    const double fxscale = a.fxs ? a.fxs[0] : 0.0;          // 0: f64 atomics
    const bool digest = a.eri_out == nullptr && a.schwarz_out == nullptr;
    constexpr bool HOIST = qc_hoisted(L);
    constexpr int CH = qc_hoist_chunk(L, LGC);                  // primitive quartets per chunk of the hoisted path
    constexpr int NHP = qc_nherm(L) | 1;                    // padded table length of the hoisted path
    double *const Rw = lds + (size_t)g * slot_words;       // this group's private LDS region
    double *const Iblk = Rw + qc_region0(L, LGC);
    const size_t rep = (size_t)(blk % a.nrep) * a.rep_stride;   // accumulation replica of this workgroup
    constexpr bool MFMA = MFMA_OK && qc_use_mfma(LAB, LCD) && LGC == 6;   // contractions on the matrix cores (one slot per wave)
    // recurrence plan of this class's R tables: behind the groups' regions, shared by them (read after the first barrier below)
    int2 *const plan = reinterpret_cast<int2 *>(lds + (size_t)G * slot_words);
    if constexpr (!HOIST) {
        const int2 *__restrict__ gp = a.rplan + qc_plan_off(L);
        for (int i = lane; i < qc_nplan(L); i += 64) plan[i] = gp[i];
    }
#ifdef QC_PHASE_TIMING
    long long tph[8] = {};
#define QC_T(i) do { const long long t_ = wall_clock64(); tph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define QC_T(i) do {} while (0)
#endif
    // bra-run mode: row buffer behind the groups' regions (and the plan): exchange rows [spin][na + nb][n], then J_ab[nab]
    const bool use_rb = rb_rows > 0 && digest;
    double *const rbK = lds + (size_t)G * slot_words + (HOIST ? 0 : qc_nplan(L));
    double *const rbJ = rbK + (size_t)(uhf ? 2 : 1) * rb_rows * n;
    if (use_rb) {
        for (int i = lane; i < (uhf ? 2 : 1) * rb_rows * n + rb_rows * rb_rows; i += 64) rbK[i] = 0.0;
        __syncthreads();
    }
    int rb_bra = -1;                                       // bra whose targets the row buffer holds (wave-uniform)
    // what is non-zero in the row buffer goes to Gt: rows of the bra's functions (a first, then b), then the J_ab block
    auto rb_flush = [&](const int bra) {
        const QcPairDesc pf = a.pairs[bra];
        const int na = pf.na, nb = pf.nb, rows = na + nb;
        const float inn = __builtin_amdgcn_rcpf((float)n), inb = __builtin_amdgcn_rcpf((float)nb);
        const size_t rep = (size_t)(blk % a.nrep) * a.rep_stride;
        for (int s = 0; s < (uhf ? 2 : 1); ++s) {
            double *Gs = (s ? a.G1 : a.G0) + rep;
            double *Ks = rbK + (size_t)s * rb_rows * n;
            for (int x = lane; x < rows * n; x += 64) {
                const double v = Ks[x];
                if (v != 0.0) {
                    const int r = qc_fdiv(x, inn), col = x - r * n;
                    qc_gadd(&Gs[(size_t)(r < na ? pf.offa + r : pf.offb + r - na) * n + col], v, fxscale, a.fx_lo);
                    Ks[x] = 0.0;
                }
            }
        }
        for (int ab = lane; ab < na * nb; ab += 64) {
            const double v = rbJ[ab];
            if (v != 0.0) {
                const int r = qc_fdiv(ab, inb);
                const size_t o = (size_t)(pf.offa + r) * n + pf.offb + ab - r * nb;
                qc_gadd2(&a.G0[rep + o], &a.G1[rep + o], uhf, v, fxscale, a.fx_lo);
                rbJ[ab] = 0.0;
            }
        }
        __syncthreads();
    };
    const int nbatch = (nslots + G - 1) / G;
    const int w_begin = run > 0 ? blk * run : blk, w_end = run > 0 ? min(nbatch, (blk + 1) * run) : nbatch, w_step = run > 0 ? 1 : nblk;
    for (int wave = w_begin; wave < w_end; wave += w_step) {
#ifdef QC_PHASE_TIMING
        long long tlast = wall_clock64();
#endif
        const int slot = wave * G + g;
        const QcSlot sl = slots[min(slot, nslots - 1)];
        const bool active = slot < nslots && sl.ket >= 0;  // (null slots pad the batches of the bra-run mode)
        if (use_rb) {
            const int bra0 = __builtin_amdgcn_readfirstlane(slots[min(wave * G, nslots - 1)].bra);
            if (bra0 != rb_bra) { if (rb_bra >= 0) rb_flush(rb_bra); rb_bra = bra0; }
        }
        const QcPairDesc pb = a.pairs[sl.bra], pk = a.pairs[max(sl.ket, 0)];
        const int na = pb.na, nb = pb.nb, nc = pk.na, nd = pk.nb;
        const int nab = na * nb, ncd = nc * nd;
        const float inb = __builtin_amdgcn_rcpf((float)nb), inc = __builtin_amdgcn_rcpf((float)nc), ind = __builtin_amdgcn_rcpf((float)nd);
        double *tDj_ab = Iblk + nab * ncd, *tDj_cd = tDj_ab + nab;
        double *tK = tDj_cd + ncd;   // per spin: Dk_ac, Dk_ad, Dk_bc, Dk_bd
        const int ktile = na * nc + na * nd + nb * nc + nb * nd;
        // 64-lane groups digest through LDS accumulators of the six target blocks (see the digestion below)
        double *const aJcd = tK + 2 * ktile, *const aK = aJcd + ncd;

        if (active) {
            for (int i = li; i < nab * ncd; i += C) Iblk[i] = 0.0;
            if constexpr (LGC == 6) { if (digest) for (int i = li; i < ncd + 2 * ktile; i += C) aJcd[i] = 0.0; }
            if (digest) {   // stage the density tiles this quartet touches
                for (int i = li; i < nab; i += C) { const int r = qc_fdiv(i, inb); tDj_ab[i] = a.Dj[(size_t)(pb.offa + r) * n + pb.offb + i - r * nb]; }
                for (int i = li; i < ncd; i += C) { const int r = qc_fdiv(i, ind); tDj_cd[i] = a.Dj[(size_t)(pk.offa + r) * n + pk.offb + i - r * nd]; }
                for (int s = 0; s < (uhf ? 2 : 1); ++s) {
                    const double *Dk = s ? a.Dk1 : a.Dk0;
                    double *t0 = tK + s * ktile, *t1 = t0 + na * nc, *t2 = t1 + na * nd, *t3 = t2 + nb * nc;
                    for (int i = li; i < na * nc; i += C) { const int r = qc_fdiv(i, inc); t0[i] = Dk[(size_t)(pb.offa + r) * n + pk.offa + i - r * nc]; }
                    for (int i = li; i < na * nd; i += C) { const int r = qc_fdiv(i, ind); t1[i] = Dk[(size_t)(pb.offa + r) * n + pk.offb + i - r * nd]; }
                    for (int i = li; i < nb * nc; i += C) { const int r = qc_fdiv(i, inc); t2[i] = Dk[(size_t)(pb.offb + r) * n + pk.offa + i - r * nc]; }
                    for (int i = li; i < nb * nd; i += C) { const int r = qc_fdiv(i, ind); t3[i] = Dk[(size_t)(pb.offb + r) * n + pk.offb + i - r * nd]; }
                }
            }
        }
        QC_T(0);
        const int strideB = qc_pair_stride(LAB, nab), strideK = qc_pair_stride(LCD, ncd);
        const int len = active ? sl.hi - sl.lo : 0;
        int maxlen = len;                                   // uniform trip count: the longest slot of this wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
        const double *braBase = pd + pb.doff, *ketBase = pd + pk.doff;
        const int npass = (LGC == 6) ? (ncd + 63) / 64 : 1;  // > 1 only for ket pairs with more than 64 function pairs
        const int cbeg = active ? sl.c0 : 0, cend = active ? sl.c1 : 0;   // ket columns of this slot (all of them unless the host split the class)

        if constexpr (MFMA) {
          // ---- high-order kets (LCD >= 4): both Hermite contractions on the matrix cores.  One slot per wave.
          //   step 2  W[h1][c] += sum_h2 Rm[h1][h2] Ee[h2][c],  Rm[h1][h2] = (-1)^{|h2|} R_{h1+h2},  Ee = E_cd,kl * pref
          //   step 3  I[ab][c] += sum_h1 Et[ab][h1] W[h1][c]
          // as v_mfma_f64_16x16x4 tiles.  The VALU form reads one R value from LDS per FMA in every lane (7056 reads
          // per lane for (ff|ff)); here the gathered Rm fragment is spread over the wave (110 reads per lane) and the W
          // accumulators - result layout C[4r + (l >> 4)][l & 15] - are, register for register, the B operands
          // B[k = l >> 4][j = l & 15] of step 3's k-steps, so nothing moves between the two products.
          typedef double qc_d4 __attribute__((ext_vector_type(4)));
          constexpr int MT = (HAB + 15) / 16, KS = (HCD + 3) / 4;
          const int i16 = lane & 15, q4 = lane >> 4;
          const uint4 *__restrict__ gi = a.gidx + qc_gidx_off(LAB, LCD) + lane;         // this lane's gather records, one per k-step
          const char *const Rb = reinterpret_cast<const char *>(Rw);
          const double *__restrict__ pdT = a.pairdataT;
          const int K_cd = pk.K;
          // One pass over the slot's primitive quartets for the column tiles [col0, col0 + 16 NT); NT is a compile-time constant of the
          // instance (a run-time bound put a branch around every matrix instruction)
          auto tile = [&](auto ntc, const int col0) {
            constexpr int NT = decltype(ntc)::value;
            constexpr bool AHEAD = MT * NT <= 12;           // registers for a second set of step-3 A fragments
            qc_d4 Wacc[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Wacc[mt][nt] = qc_d4{0.0, 0.0, 0.0, 0.0};
            // A fragments of step 3, row tile `it`: Et[ab][h1], shared by the column tiles (clamped loads, masked to zero outside the block)
            auto load_e = [&](const double *__restrict__ Et, int it, double (&av)[MT][4]) {
                const int ab = 16 * it + i16;
                const double *row = Et + (size_t)min(ab, nab - 1) * HAB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int h1 = 16 * mt + 4 * r + q4;
                        const double v = row[min(h1, HAB - 1)];
                        av[mt][r] = (ab < nab && h1 < HAB) ? v : 0.0;
                    }
            };
            // step 3 for the bra primitive pair whose expansion block is Et; avE = the fragments of row tile 0 (requested by the caller
            // before the Boys function of the pair's last primitive quartet); with registers to spare the next tile's travel one ahead
            auto flush = [&](const double *__restrict__ Et, double (&avE)[MT][4]) {
                const int MI = (nab + 15) / 16;
                for (int it = 0; it < MI; ++it) {
                    double avN[AHEAD ? MT : 1][4];
                    if constexpr (AHEAD) load_e(Et, min(it + 1, MI - 1), avN);
                    else { if (it > 0) load_e(Et, it, avE); }
                    __builtin_amdgcn_sched_barrier(0);
                    qc_d4 acc[NT][2];                       // two accumulators per column tile: consecutive matrix instructions are independent
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][0] = acc[nt][1] = qc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][r & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(avE[mt][r], Wacc[mt][nt][r], acc[nt][r & 1], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int c = col0 + 16 * nt + i16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int abr = 16 * it + q4 + 4 * r;
                            if (abr < nab && c < cend) Iblk[abr * ncd + c] += acc[nt][0][r] + acc[nt][1][r];     // this lane alone owns the element
                        }
                    }
                    if constexpr (AHEAD) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) avE[mt][r] = avN[mt][r];
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) Wacc[mt][nt] = qc_d4{0.0, 0.0, 0.0, 0.0};
            };
            int ij = sl.lo / K_cd, kl = sl.lo - ij * K_cd;
            for (int itq = 0; itq < len; ++itq) {
                QC_T(4);
                const bool last = kl == K_cd - 1 || itq == len - 1;      // of this bra primitive pair: step 3 follows
                const double *__restrict__ Et = pdT + pb.doff + (size_t)ij * strideB + 4;     // [ab][h]
                const double4 cb = *reinterpret_cast<const double4 *>(braBase + (size_t)ij * strideB);
                const double *ket = ketBase + (size_t)kl * strideK;
                const double4 ck = *reinterpret_cast<const double4 *>(ket);
                double avE[MT][4];
                if (last) load_e(Et, 0, avE);
                const double p = cb.x, q = ck.x;
                const double X = cb.y - ck.y, Y = cb.z - ck.z, Z = cb.w - ck.w;
                const double pref = qc_rsqrt(p + q);
                const double alpha = p * q * (pref * pref);
                // Operand fragments of k-step ks.  B = the column tiles of the ket block (plain loads, clamped: rows past HCD meet a zero
                // A value, columns past ncd are never read back); A = the gathered, signed R values: which LDS word lane l reads for row
                // tile mt comes from a host-built record (qc_build_gidx), eight u16 per k-step and lane.
                const double *Ecd = ket + 4;
                auto load_b = [&](int ks, double (&bv)[NT]) {
                    const int h2 = min(4 * ks + q4, HCD - 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = Ecd[(size_t)h2 * ncd + min(col0 + 16 * nt + i16, ncd - 1)];
                };
                auto gather = [&](const uint4 rec, double (&raw)[MT]) {
                    const unsigned w[3] = {rec.x, rec.y, rec.z};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const unsigned off = (mt & 1) ? (w[mt >> 1] >> 16) : (w[mt >> 1] & 0xffffu);
                        raw[mt] = *reinterpret_cast<const double *>(Rb + off);
                    }
                };
                auto scale = [&](const unsigned recw, const double (&raw)[MT], double (&av)[MT]) {
                    const unsigned fl = recw >> 16;
                    const double sg = (fl & 2u) ? ((fl & 1u) ? -pref : pref) : 0.0;          // sign of the ket order, scale, validity
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = sg * raw[mt];
                };
                // B fragments and gather records travel PD k-steps ahead of their matrix instructions (a k-step is MT NT of them, 64 cycles
                // each; a ket block comes from the L2 / the memory-side cache in 500-1200): the first PD are requested before the Boys
                // function and the R table, a buffer is refilled as soon as its instructions have issued.  The R values of step ks + 1
                // are read from LDS before the instructions of step ks and scaled after them.
                constexpr int PD = MT * NT >= 12 ? 3 : (MT * NT >= 6 ? 4 : 6);
                double bv[PD][NT], avA[MT];
                uint4 gr[PD];
#pragma unroll
                for (int d = 0; d < PD; ++d) { load_b(min(d, KS - 1), bv[d]); gr[d] = gi[64 * min(d + 1, KS - 1)]; }
                const uint4 gr0 = gi[0];
                double F[L + 1];
                qc_boys<L>(alpha * (X * X + Y * Y + Z * Z), a.boys, F);
                __syncthreads();                  // previous iteration's readers of Rw are done
                QC_T(1);
                qc_build_r<L, 64>(Rw, plan, lane, alpha, X, Y, Z, F);
                QC_T(2);
                { double raw[MT]; gather(gr0, raw); scale(gr0.w, raw, avA); }
                for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
                    for (int d = 0; d < PD; ++d) {
                        const int ks = ks0 + d;
                        if (ks >= KS) break;
                        double raw[MT];
                        const unsigned recw = gr[d].w;
                        gather(gr[d], raw);       // (k-step ks + 1; past the end: the last one again, unused)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                Wacc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(avA[mt], bv[d][nt], Wacc[mt][nt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + PD < KS) load_b(ks + PD, bv[d]);
                        gr[d] = gi[64 * min(ks + PD + 1, KS - 1)];
                        scale(recw, raw, avA);
                    }
                }
                QC_T(3);
                if (last) flush(Et, avE);
                if (++kl == K_cd) { kl = 0; ++ij; }
            }
            QC_T(4);
          };
          for (int col0 = cbeg; col0 < cend; col0 += 64) {       // this slot's ket columns, 64 at a time
              switch (min(4, (cend - col0 + 15) / 16)) {
                  case 1: tile(std::integral_constant<int, 1>{}, col0); break;
                  case 2: tile(std::integral_constant<int, 2>{}, col0); break;
                  case 3: tile(std::integral_constant<int, 3>{}, col0); break;
                  default: tile(std::integral_constant<int, 4>{}, col0); break;
              }
          }
        } else {
        for (int pass = 0; pass < npass; ++pass) {
            const int col = pass * 64 + li;
            const bool colok = col < ncd;
            double W[HAB];
#pragma unroll
            for (int h = 0; h < HAB; ++h) W[h] = 0.0;
            int cur_ij = -1;

            auto flush = [&](int ij) {   // step 3: I[ab][col] += sum_h E_ab,ij[h][ab] W[h]
                // The bra expansion block is staged transposed ([ab][h]) in the group's LDS region by coalesced loads,
                // then read back as broadcasts.  Writer and reader lanes are the same lanes of one wave (this code runs
                // under a group-uniform, possibly wave-divergent condition), DS operations of a wave execute in order,
                // so wavefront-scope fences (compiler ordering only) are sufficient - no s_barrier here.
                {
                    // groups made of whole 16-lane rows: four [ab] rows of the transposed block at a time, straight from
                    // memory into the row's lanes, and the dot products by row broadcasts - no staging, no LDS reads
                    constexpr int NRE = (HAB + 15) / 16;
                    const double *__restrict__ Et = a.pairdataT + pb.doff + (size_t)ij * strideB + 4;     // [ab][h]
                    const int l16 = lane & 15;
                    double Er[4][NRE];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < NRE; ++k) Er[j][k] = Et[(size_t)min(j, nab - 1) * HAB + min(16 * k + l16, HAB - 1)];
                    for (int ab0 = 0; ab0 < nab; ab0 += 4) {
                        double En[4][NRE];                       // the next four rows are requested before these are used
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int k = 0; k < NRE; ++k) En[j][k] = Et[(size_t)min(ab0 + 4 + j, nab - 1) * HAB + min(16 * k + l16, HAB - 1)];
                        __builtin_amdgcn_sched_barrier(0);
                        double acc[4] = {0.0, 0.0, 0.0, 0.0};
                        qc_dot4_bc<HAB, NRE>(acc, Er, W, std::make_integer_sequence<int, HAB>{});
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ab0 + j < nab && colok) Iblk[(ab0 + j) * ncd + col] += acc[j];   // column `col` of this slot belongs to this lane alone
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int k = 0; k < NRE; ++k) Er[j][k] = En[j][k];
                    }
                }
            };

            const int K_cd = pk.K;
            if constexpr (HOIST) {
                // Low-L classes: Boys + R table are the bulk of a primitive quartet and do not depend on the column, so
                // the C lanes of a group evaluate C *consecutive* primitive quartets of their slot at once (phase A,
                // tables in registers, then parked in the group's LDS region) and afterwards walk through them one by
                // one, every lane contracting its own ket column (phase B).  Same-wave LDS hand-off: wavefront fences.
                double *const meta = Rw + CH * NHP;
                // this lane's primitive quartet of the current chunk: starts at sl.lo + li, advances by CH per chunk
                int ijA, klA;
                { const int pq0 = sl.lo + li; ijA = pq0 / K_cd; klA = pq0 - ijA * K_cd; }
                for (int it0 = 0; it0 < maxlen; it0 += CH) {
                    {
                        const bool vA = li < CH && it0 + li < len;
                        const int ijC = vA ? ijA : 0, klC = vA ? klA : 0;      // stay inside the pair blocks when idle
                        const double4 cb = *reinterpret_cast<const double4 *>(braBase + (size_t)ijC * strideB);
                        const double4 ck = *reinterpret_cast<const double4 *>(ketBase + (size_t)klC * strideK);
                        const double p = cb.x, q = ck.x;
                        const double X = cb.y - ck.y, Y = cb.z - ck.z, Z = cb.w - ck.w;
                        const double pref = qc_rsqrt(p + q);
                        const double alpha = p * q * (pref * pref);
                        double F[L + 1], Rr[qc_nherm(L)];
                        qc_boys<L>(alpha * (X * X + Y * Y + Z * Z), a.boys, F);
                        qc_rtab<L>(alpha, X, Y, Z, F, Rr);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // phase-B readers of the previous chunk
                        if (li < CH) {
                            double *mine = Rw + li * NHP;
#pragma unroll
                            for (int h = 0; h < qc_nherm(L); ++h) mine[h] = Rr[h];
                            meta[2 * li] = vA ? pref : 0.0;
                            reinterpret_cast<int2 *>(meta)[2 * li + 1] = make_int2(ijC, klC * strideK + 4);   // ket block offset
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        klA += CH;
                        while (klA >= K_cd) { klA -= K_cd; ++ijA; }
                    }
                    QC_T(1);
                    const int nB = min(CH, maxlen - it0);
                    {
                        // R values by row broadcasts out of registers; the operands of step s + 1 (its record, first ket
                        // coefficients and R registers) are requested while step s computes
                        constexpr int NR = (qc_nherm(L) + 15) / 16;
                        const int l16 = lane & 15, colo = colok ? col : 0;
                        double prefN = meta[0];
                        int2 ikN = reinterpret_cast<const int2 *>(meta)[1];
                        double eA[qc_dpp_batch(LCD)], RdA[NR];
                        qc_step2_dpp_first<LAB, LCD, NR>(ketBase + ikN.y + colo, ncd, Rw, l16, eA, RdA);
                        for (int s = 0; s < nB; ++s) {
                            const bool valid = it0 + s < len;
                            const double pref = prefN;
                            const int2 ik = ikN;
                            const int sn = min(s + 1, nB - 1);
                            prefN = meta[2 * sn];
                            ikN = reinterpret_cast<const int2 *>(meta)[2 * sn + 1];
                            if (valid && ik.x != cur_ij) {
                                if (cur_ij >= 0) flush(cur_ij);
#pragma unroll
                                for (int h = 0; h < HAB; ++h) W[h] = 0.0;
                                cur_ij = ik.x;
                            }
                            const double sc = (valid && colok) ? pref : 0.0;
                            double ec[qc_dpp_batch(LCD)], Rdc[NR];
#pragma unroll
                            for (int j = 0; j < qc_dpp_batch(LCD); ++j) ec[j] = eA[j];
#pragma unroll
                            for (int k = 0; k < NR; ++k) Rdc[k] = RdA[k];
                            qc_step2_dpp_from<LAB, LCD, NR, 0>(W, Rdc, ec, ketBase + ik.y + colo, ncd, sc, ketBase + ikN.y + colo,
                                                               Rw + sn * NHP, l16, eA, RdA);
                        }
                    }
                    QC_T(3);
                }
            } else {
            // primitive-quartet loop of this slot: (ij, kl) advances incrementally; the 32-byte headers [p, P] of the next
            // primitive pair are requested one iteration ahead
            int ij = sl.lo / K_cd, kl = sl.lo - ij * K_cd;
            double4 hb = *reinterpret_cast<const double4 *>(braBase + (size_t)ij * strideB);
            double4 hk = *reinterpret_cast<const double4 *>(ketBase + (size_t)kl * strideK);
            for (int it = 0; it < maxlen; ++it) {
                const bool valid = it < len;
                if (valid && ij != cur_ij) {
                    if (cur_ij >= 0) flush(cur_ij);
#pragma unroll
                    for (int h = 0; h < HAB; ++h) W[h] = 0.0;
                    cur_ij = ij;
                }
                const double4 cb = hb, ck = hk;
                const double *ket = ketBase + (size_t)kl * strideK;
                // advance to the next primitive quartet and prefetch its headers (stay in range on the last pass)
                int nij = ij, nkl = kl + 1;
                if (nkl == K_cd) { nkl = 0; ++nij; }
                if (it + 1 < len) {
                    ij = nij; kl = nkl;
                    hb = *reinterpret_cast<const double4 *>(braBase + (size_t)ij * strideB);
                    hk = *reinterpret_cast<const double4 *>(ketBase + (size_t)kl * strideK);
                }
                double e0[qc_dpp_batch(LCD)];               // first ket coefficients: on their way while the table is built
                {
                    const double *E0 = ket + 4 + (colok ? col : 0);
#pragma unroll
                    for (int j = 0; j < qc_dpp_batch(LCD); ++j) e0[j] = (j < HCD) ? E0[(size_t)j * ncd] : 0.0;
                }
                const double p = cb.x, q = ck.x;
                const double X = cb.y - ck.y, Y = cb.z - ck.z, Z = cb.w - ck.w;
                const double pq_sum = p + q;
                const double pref = qc_rsqrt(pq_sum);                 // 1 / sqrt(p + q)
                const double alpha = p * q * (pref * pref);
                double F[L + 1];
                qc_boys<L>(alpha * (X * X + Y * Y + Z * Z), a.boys, F);
                const double sc = (valid && colok) ? pref : 0.0;
                const double *Ecd = ket + 4 + (colok ? col : 0);
                __syncthreads();                  // previous iteration's readers of Rw are done
                qc_build_r<L, C>(Rw, plan, li, alpha, X, Y, Z, F);
                qc_step2_dpp<LAB, LCD>(W, e0, Ecd, ncd, sc, Rw, lane);
            }
            }
            if (cur_ij >= 0) flush(cur_ij);
        }
        }
        __syncthreads();
        QC_T(5);

        if (active) {
            const double f = (pb.shA_eq_shB ? 0.5 : 1.0) * (pk.shA_eq_shB ? 0.5 : 1.0) * (sl.bra == sl.ket ? 0.5 : 1.0);
            if (a.schwarz_out != nullptr) {
                // Schwarz factor of pair P from the complete (P|P) block (unsplit slot): by the Schwarz inequality the largest
                // |(ab|cd)| of the block sits on its diagonal, so no index matching is needed
                double m = 0.0;
                for (int x = li; x < nab * ncd; x += C) m = fmax(m, fabs(Iblk[x]));
#pragma unroll
                for (int o = C / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, C));
                if (li == 0) a.schwarz_out[sl.bra] = sqrt(m);
            } else if (a.eri_out != nullptr) {
                // materialise (ij|kl) with its 8 symmetry images: the tensor molint::eri returns (tests / plumbing only;
                // the host hands this mode unsplit slots, so plain stores are complete values)
                const size_t n1 = n, n2 = n1 * n1, n3 = n2 * n1;
                for (int x = li; x < nab * ncd; x += C) {
                    const int ab = x / ncd, cd = x - ab * ncd;
                    const size_t i = pb.offa + ab / nb, j = pb.offb + ab % nb, k = pk.offa + cd / nd, l = pk.offb + cd % nd;
                    const double v = Iblk[x];
                    double *o = a.eri_out;
                    o[i * n3 + j * n2 + k * n1 + l] = v; o[j * n3 + i * n2 + k * n1 + l] = v;
                    o[i * n3 + j * n2 + l * n1 + k] = v; o[j * n3 + i * n2 + l * n1 + k] = v;
                    o[k * n3 + l * n2 + i * n1 + j] = v; o[l * n3 + k * n2 + i * n1 + j] = v;
                    o[k * n3 + l * n2 + j * n1 + i] = v; o[l * n3 + k * n2 + j * n1 + i] = v;
                }
            } else {
                double *G0 = a.G0 + rep, *G1 = a.G1 + rep;
                const double fj = 2.0 * f, fk = -a.cK * f;
                if (LGC == 6 && nab >= 16) {
                    // One wave, one slot, a bra with enough function pairs to occupy the lanes.  A lane owns a bra function pair (i,j) and walks along the slot's ket columns (k,l), the same
                    // column in every lane: J_ab and, between two changes of k, K_ac and K_bc accumulate in registers; K_ad and K_bd go to
                    // LDS accumulators per column (lanes sharing i or j meet there: na- or nb-way), J_cd is a second, column-owned pass.
                    // Afterwards every touched accumulator leaves as one global atomic.  (The per-target loops of the narrow groups
                    // below are chains of dependent FMAs with most lanes idle at these block shapes: 14 us for a 49 x 13 block, more
                    // than its two contractions; one LDS atomic per product, tried first, queues up to 32 lanes on one word: 8 us.)
                    const double *t_ac = tK, *t_ad = t_ac + na * nc, *t_bc = t_ad + na * nd, *t_bd = t_bc + nb * nc;
                    double *k_ac = aK, *k_ad = k_ac + na * nc, *k_bc = k_ad + na * nd, *k_bd = k_bc + nb * nc;
                    const int k0 = qc_fdiv(cbeg, ind), l0 = cbeg - k0 * nd;
                    for (int ab = lane; ab < nab; ab += 64) {
                        const int i = qc_fdiv(ab, inb), j = ab - i * nb;
                        const double *Irow = Iblk + ab * ncd;
                        double jab = 0.0, aik[2] = {0.0, 0.0}, ajk[2] = {0.0, 0.0};
                        int k = k0, l = l0;
                        for (int c = cbeg; c < cend; ++c) {
                            const double v = Irow[c];
                            jab = fma(v, tDj_cd[c], jab);
                            aik[0] = fma(v, t_bd[j * nd + l], aik[0]);
                            ajk[0] = fma(v, t_ad[i * nd + l], ajk[0]);
                            qc_ds_add(&k_ad[i * nd + l], v * t_bc[j * nc + k]);
                            qc_ds_add(&k_bd[j * nd + l], v * t_ac[i * nc + k]);
                            if (uhf) {
                                aik[1] = fma(v, t_bd[ktile + j * nd + l], aik[1]);
                                ajk[1] = fma(v, t_ad[ktile + i * nd + l], ajk[1]);
                                qc_ds_add(&k_ad[ktile + i * nd + l], v * t_bc[ktile + j * nc + k]);
                                qc_ds_add(&k_bd[ktile + j * nd + l], v * t_ac[ktile + i * nc + k]);
                            }
                            if (++l == nd || c + 1 == cend) {
                                qc_ds_add(&k_ac[i * nc + k], aik[0]); qc_ds_add(&k_bc[j * nc + k], ajk[0]);
                                if (uhf) { qc_ds_add(&k_ac[ktile + i * nc + k], aik[1]); qc_ds_add(&k_bc[ktile + j * nc + k], ajk[1]); }
                                aik[0] = aik[1] = ajk[0] = ajk[1] = 0.0;
                                l = 0; ++k;
                            }
                        }
                        const size_t o = (size_t)(pb.offa + i) * n + pb.offb + j;
                        qc_gadd2(&G0[o], &G1[o], uhf, fj * jab, fxscale, a.fx_lo);
                    }
                    {   // J_cd: lane = (part of the bra range, column)
                        const int wc = cend - cbeg, parts = wc >= 64 ? 1 : 64 / wc, chunk = (nab + parts - 1) / parts;
                        const float iwc = __builtin_amdgcn_rcpf((float)wc);
                        for (int y = lane; y < wc * parts; y += 64) {
                            const int part = qc_fdiv(y, iwc), c = cbeg + y - part * wc;
                            const int ab1 = min(nab, (part + 1) * chunk);
                            double s0 = 0.0, s1 = 0.0;
                            int ab = part * chunk;
                            for (; ab + 1 < ab1; ab += 2) { s0 = fma(Iblk[ab * ncd + c], tDj_ab[ab], s0); s1 = fma(Iblk[(ab + 1) * ncd + c], tDj_ab[ab + 1], s1); }
                            if (ab < ab1) s0 = fma(Iblk[ab * ncd + c], tDj_ab[ab], s0);
                            qc_ds_add(&aJcd[c], s0 + s1);
                        }
                    }
                    __syncthreads();
                    QC_T(7);
                    for (int cd = cbeg + lane; cd < cend; cd += 64) {
                        const int r = qc_fdiv(cd, ind);
                        const size_t o = (size_t)(pk.offa + r) * n + pk.offb + cd - r * nd;
                        qc_gadd2(&G0[o], &G1[o], uhf, fj * aJcd[cd], fxscale, a.fx_lo);
                    }
                    for (int s = 0; s < (uhf ? 2 : 1); ++s) {
                        double *Gs = s ? G1 : G0;
                        const double *w_ac = aK + s * ktile, *w_ad = w_ac + na * nc, *w_bc = w_ad + na * nd, *w_bd = w_bc + nb * nc;
                        // (a column split leaves most exchange targets of a slot untouched: exact zeros stay home)
                        for (int x = lane; x < na * nc; x += 64) { const int i = qc_fdiv(x, inc); const double v = w_ac[x]; if (v != 0.0) qc_gadd(&Gs[(size_t)(pb.offa + i) * n + pk.offa + x - i * nc], fk * v, fxscale, a.fx_lo); }
                        for (int x = lane; x < na * nd; x += 64) { const int i = qc_fdiv(x, ind); const double v = w_ad[x]; if (v != 0.0) qc_gadd(&Gs[(size_t)(pb.offa + i) * n + pk.offb + x - i * nd], fk * v, fxscale, a.fx_lo); }
                        for (int x = lane; x < nb * nc; x += 64) { const int j = qc_fdiv(x, inc); const double v = w_bc[x]; if (v != 0.0) qc_gadd(&Gs[(size_t)(pb.offb + j) * n + pk.offa + x - j * nc], fk * v, fxscale, a.fx_lo); }
                        for (int x = lane; x < nb * nd; x += 64) { const int j = qc_fdiv(x, ind); const double v = w_bd[x]; if (v != 0.0) qc_gadd(&Gs[(size_t)(pb.offb + j) * n + pk.offb + x - j * nd], fk * v, fxscale, a.fx_lo); }
                    }
                } else {
                // J blocks: Gt_ab += 2f sum_cd I D_cd ; Gt_cd += 2f sum_ab I D_ab   (final G = Gt + Gt^T)
                // (integrals outside [cbeg, cend) are not this slot's: the loops run over its columns only)
                for (int ab = li; ab < nab; ab += C) {
                    double s = 0.0;
#pragma unroll 4
                    for (int cd = cbeg; cd < cend; ++cd) s = fma(Iblk[ab * ncd + cd], tDj_cd[cd], s);
                    if (use_rb) { qc_ds_add(&rbJ[ab], fj * s); continue; }
                    const int r = qc_fdiv(ab, inb);
                    const size_t o = (size_t)(pb.offa + r) * n + pb.offb + ab - r * nb;
                    qc_gadd2(&G0[o], &G1[o], uhf, fj * s, fxscale, a.fx_lo);
                }
                for (int cd = cbeg + li; cd < cend; cd += C) {
                    double s = 0.0;
#pragma unroll 4
                    for (int ab = 0; ab < nab; ++ab) s = fma(Iblk[ab * ncd + cd], tDj_ab[ab], s);
                    const int r = qc_fdiv(cd, ind);
                    const size_t o = (size_t)(pk.offa + r) * n + pk.offb + cd - r * nd;
                    qc_gadd2(&G0[o], &G1[o], uhf, fj * s, fxscale, a.fx_lo);
                }
                // K blocks: Gt_ac -= cK f sum_bd I D_bd, and the ad / bc / bd images
                for (int s = 0; s < (uhf ? 2 : 1); ++s) {
                    double *Gs = s ? G1 : G0;
                    double *const Ks = rbK + (size_t)s * rb_rows * n;     // (row buffer: rows of a, then rows of b)
                    const double *t_ac = tK + s * ktile, *t_ad = t_ac + na * nc, *t_bc = t_ad + na * nd, *t_bd = t_bc + nb * nc;
                    for (int x = li; x < na * nc; x += C) {          // (i,k) <- sum_{j,l} I[ij,kl] D[j,l]
                        const int i = qc_fdiv(x, inc), k = x - i * nc;
                        const int l0 = max(0, cbeg - k * nd), l1 = min(nd, cend - k * nd);
                        double acc = 0.0;
                        for (int j = 0; j < nb; ++j)
#pragma unroll 3
                            for (int l = l0; l < l1; ++l) acc = fma(Iblk[(i * nb + j) * ncd + k * nd + l], t_bd[j * nd + l], acc);
                        if (use_rb) qc_ds_add(&Ks[i * n + pk.offa + k], fk * acc);
                        else qc_gadd(&Gs[(size_t)(pb.offa + i) * n + pk.offa + k], fk * acc, fxscale, a.fx_lo);
                    }
                    for (int x = li; x < na * nd; x += C) {          // (i,l) <- sum_{j,k} I[ij,kl] D[j,k]
                        const int i = qc_fdiv(x, ind), l = x - i * nd;
                        const int k0 = qc_fdiv(max(0, cbeg - l + nd - 1), ind), k1 = min(nc, qc_fdiv(max(0, cend - l + nd - 1), ind));
                        double acc = 0.0;
                        for (int j = 0; j < nb; ++j)
#pragma unroll 3
                            for (int k = k0; k < k1; ++k) acc = fma(Iblk[(i * nb + j) * ncd + k * nd + l], t_bc[j * nc + k], acc);
                        if (use_rb) qc_ds_add(&Ks[i * n + pk.offb + l], fk * acc);
                        else qc_gadd(&Gs[(size_t)(pb.offa + i) * n + pk.offb + l], fk * acc, fxscale, a.fx_lo);
                    }
                    for (int x = li; x < nb * nc; x += C) {          // (j,k) <- sum_{i,l} I[ij,kl] D[i,l]
                        const int j = qc_fdiv(x, inc), k = x - j * nc;
                        const int l0 = max(0, cbeg - k * nd), l1 = min(nd, cend - k * nd);
                        double acc = 0.0;
                        for (int i = 0; i < na; ++i)
#pragma unroll 3
                            for (int l = l0; l < l1; ++l) acc = fma(Iblk[(i * nb + j) * ncd + k * nd + l], t_ad[i * nd + l], acc);
                        if (use_rb) qc_ds_add(&Ks[(na + j) * n + pk.offa + k], fk * acc);
                        else qc_gadd(&Gs[(size_t)(pb.offb + j) * n + pk.offa + k], fk * acc, fxscale, a.fx_lo);
                    }
                    for (int x = li; x < nb * nd; x += C) {          // (j,l) <- sum_{i,k} I[ij,kl] D[i,k]
                        const int j = qc_fdiv(x, ind), l = x - j * nd;
                        const int k0 = qc_fdiv(max(0, cbeg - l + nd - 1), ind), k1 = min(nc, qc_fdiv(max(0, cend - l + nd - 1), ind));
                        double acc = 0.0;
                        for (int i = 0; i < na; ++i)
#pragma unroll 3
                            for (int k = k0; k < k1; ++k) acc = fma(Iblk[(i * nb + j) * ncd + k * nd + l], t_ac[i * nc + k], acc);
                        if (use_rb) qc_ds_add(&Ks[(na + j) * n + pk.offb + l], fk * acc);
                        else qc_gadd(&Gs[(size_t)(pb.offb + j) * n + pk.offb + l], fk * acc, fxscale, a.fx_lo);
                    }
                }
                }
            }
        }
        __syncthreads();   // the slot regions are reused by the next batch of slots
        QC_T(6);
    }
    if (use_rb && rb_bra >= 0) rb_flush(rb_bra);
#ifdef QC_PHASE_TIMING
#ifdef QC_PHASE_ALL
    if (lane == 0 && blk < 2 && digest)
#else
    if (MFMA && lane == 0 && blk < 2 && digest)
#endif
        printf("[phase] <%d,%d,%d> blk %d/%d nslots %d: setup %lld  boys+hdr %lld  rtab %lld  kloop %lld  flush %lld  tail %lld  products %lld  atomics %lld  (10 ns units)\n", LAB, LCD, LGC, blk, nblk, nslots,
               tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[7], tph[6]);
#endif
#undef QC_T
}

// One launch = one "tier" of a bra class: all (LCD, LGC) buckets of LAB with LCD <= 3 (tier 0: moderate register
// footprint, several waves per SIMD) or LCD >= 4 (tier 1).  The grid is the concatenation of the buckets' block
// ranges ("segments"); a workgroup finds its segment and runs that bucket's body.  Fewer, fuller launches: the eager
// launch of ~35 separate class kernels was host-launch-bound (~10 us each).
constexpr int QC_MAXSEG = 12;
struct QcTierArgs {
    QcKernelArgs base;
    int nseg;
    int seg_lab[QC_MAXSEG];        // bra class of the segment (the merged wide-ket launch of the low bra classes, qc_fock_tier1_low_kernel)
    int seg_end[QC_MAXSEG];        // exclusive prefix of workgroup counts
    int seg_code[QC_MAXSEG];       // (LCD << 4) | LGC
    int seg_nslots[QC_MAXSEG];
    int seg_words[QC_MAXSEG];
    int seg_run[QC_MAXSEG];        // bra-run mode: batches per workgroup (0: independent slots, grid-stride)
    int seg_rbrows[QC_MAXSEG];     // ... rows of the LDS row buffer (0: none)
    const QcSlot *seg_slots[QC_MAXSEG];
};

// (TIER 2 = the tier-1 launch of a build without f-ket classes - every segment an LCD = 4 bucket, any basis without f functions: the same
// launch unit, but a kernel that does not carry the register footprint of the matrix-core bodies and so keeps two waves per SIMD:
// (ps|dd) of benzene/cc-pVDZ 0.142 -> 0.108 ms, (pp|dd) 0.099 -> 0.081 alone)
template <int LAB, int TIER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(((TIER == 0 && LAB <= 2) || TIER == 2) ? 2 : 1)))
void qc_fock_tier_kernel(const QcTierArgs a) {
    if (qc_build_cancelled(a.base)) return;
    qc_tl_stamp(a.base.tl, 0);
    // the high-L tiers are few, long, latency-bound waves: let them win issue arbitration against the many short
    // low-L waves they share a SIMD with
    if constexpr (TIER >= 1) __builtin_amdgcn_s_setprio(3);
    int s = 0;
    while (s + 1 < a.nseg && (int)blockIdx.x >= a.seg_end[s]) ++s;
    const int b0 = s ? a.seg_end[s - 1] : 0;
    const int blk = blockIdx.x - b0, nblk = a.seg_end[s] - b0;
    const QcSlot *slots = a.seg_slots[s];
    const int nslots = a.seg_nslots[s], words = a.seg_words[s];
#define QC_CASE(LCD, LGC) case ((LCD) << 4 | (LGC)): qc_fock_body<LAB, LCD, LGC>(a.base, slots, nslots, words, blk, nblk, a.seg_run[s], a.seg_rbrows[s]); break;
    if constexpr (TIER == 0) {
        switch (a.seg_code[s]) { QC_CASE(2, 4) QC_CASE(3, 4) QC_CASE(3, 5) default: break; }
    } else if constexpr (TIER == 1) {
        switch (a.seg_code[s]) { QC_CASE(4, 5) QC_CASE(4, 6) QC_CASE(5, 6) QC_CASE(6, 6) default: break; }
    } else {
#define QC_CASE_V(LCD, LGC) case ((LCD) << 4 | (LGC)): qc_fock_body<LAB, LCD, LGC, false>(a.base, slots, nslots, words, blk, nblk, a.seg_run[s], a.seg_rbrows[s]); break;
        switch (a.seg_code[s]) { QC_CASE_V(4, 5) QC_CASE_V(4, 6) default: break; }
#undef QC_CASE_V
    }
#undef QC_CASE
    qc_tl_stamp(a.base.tl, 1);
}

// (round 3) ONE launch for the wide-ket buckets (LCD >= 4) of the bra classes LAB = 0, 1 and 2 of a basis with f functions: three launches of
// 500-1100 one-wave workgroups at one wave per SIMD each, 27 + 35 + 54 us in-build on H2O/cc-pVTZ, chained on the side streams - their
// workgroups now fill the chip together.  All three kernels ran at one wave per SIMD already (256 + 6 / 12 / 70 registers), so the merged
// kernel costs no occupancy; every workgroup gets the largest segment's LDS (17 KB instead of 8-10 for the LAB <= 1 buckets: four waves per
// CU, no limit).  The per-class launches of the profiling / set-up passes keep the per-LAB kernels.
#ifndef QC_T1LOW_WAVES
#define QC_T1LOW_WAVES(V) 1
#endif
template <int V>     // (every instance lives in its own translation unit: gen/qc_fock_low1.hip, _mid1, _hi1)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QC_T1LOW_WAVES(V))))
void qc_fock_tier1_low_kernel(const QcTierArgs a) {
    if (qc_build_cancelled(a.base)) return;
    qc_tl_stamp(a.base.tl, 0);
    __builtin_amdgcn_s_setprio(3);
    int s = 0;
    while (s + 1 < a.nseg && (int)blockIdx.x >= a.seg_end[s]) ++s;
    const int b0 = s ? a.seg_end[s - 1] : 0;
    const int blk = blockIdx.x - b0, nblk = a.seg_end[s] - b0;
    const QcSlot *slots = a.seg_slots[s];
    const int nslots = a.seg_nslots[s], words = a.seg_words[s];
#define QC_CASE(LAB, LCD, LGC) case ((LAB) << 8 | (LCD) << 4 | (LGC)): qc_fock_body<LAB, LCD, LGC>(a.base, slots, nslots, words, blk, nblk, a.seg_run[s], a.seg_rbrows[s]); break;
    // V = 0: bra classes 0, 1, 2;  V = 1: 3 and 4 (994 + 350 workgroups on H2O/cc-pVTZ, 22 / 31 KB of LDS: still four per CU);  V = 2: 5 and 6
    // (46 + 4 workgroups, 42 / 56 KB) - the f.d / f.f bras stay out of the launch of classes 3 and 4 because their LDS would halve its waves
    if constexpr (V == 0) {
        switch ((a.seg_lab[s] << 8) | a.seg_code[s]) {
            QC_CASE(0, 5, 6) QC_CASE(0, 6, 6)
            QC_CASE(1, 4, 5) QC_CASE(1, 4, 6) QC_CASE(1, 5, 6) QC_CASE(1, 6, 6)
            QC_CASE(2, 4, 5) QC_CASE(2, 4, 6) QC_CASE(2, 5, 6) QC_CASE(2, 6, 6)
            default: break;
        }
    } else if constexpr (V == 1) {
        switch ((a.seg_lab[s] << 8) | a.seg_code[s]) {
            QC_CASE(3, 4, 5) QC_CASE(3, 4, 6) QC_CASE(3, 5, 6) QC_CASE(3, 6, 6)
            QC_CASE(4, 4, 5) QC_CASE(4, 4, 6) QC_CASE(4, 5, 6) QC_CASE(4, 6, 6)
            default: break;
        }
    } else {
        switch ((a.seg_lab[s] << 8) | a.seg_code[s]) {
            QC_CASE(5, 4, 5) QC_CASE(5, 4, 6) QC_CASE(5, 5, 6) QC_CASE(5, 6, 6)
            QC_CASE(6, 4, 5) QC_CASE(6, 4, 6) QC_CASE(6, 5, 6) QC_CASE(6, 6, 6)
            default: break;
        }
    }
#undef QC_CASE
    qc_tl_stamp(a.base.tl, 1);
}
template <int V>
int qc_launch_tier1_low_impl(int grid, size_t lds, hipStream_t st, const QcTierArgs &a) {
    static std::atomic<bool> raised{false};
    if (lds > 48 * 1024 && !raised.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(qc_fock_tier1_low_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, QC_LDS_MAX);
        if (e != hipSuccess) return QC_ERR_HIP;
        raised.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(qc_fock_tier1_low_kernel<V>, dim3(grid), dim3(64), lds, st, a);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

template <int LAB, int TIER>
int qc_launch_tier(int grid, size_t lds, hipStream_t st, const QcTierArgs &a) {
    auto kern = qc_fock_tier_kernel<LAB, TIER>;
    // the dynamic-LDS cap of this instantiation is raised once, to the device maximum (the attribute is process-wide, so a
    // per-handle high-water mark would let one handle lower another's cap; the flag only saves the repeated call)
    static std::atomic<bool> raised{false};
    if (lds > 48 * 1024 && !raised.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QC_LDS_MAX);
        if (e != hipSuccess) return QC_ERR_HIP;
        raised.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, a);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}
