// qc_fock_bm.hip - "bra-major" ERI + Fock digestion kernels for the quartet classes whose narrower pair is an ss or a
// ps pair and whose total Hermite order is <= QC_LREG (all tables in registers; ps kets against d.d / f.p bras reach order 5), and for the
// p.p kets of large lists against p.p / d.s bras.  gfx950 / wave64.
//
// Same mathematics as qc_fock_kernel.h (molint::eri, rhf.rs:45, fused with compute_electronic_hamiltonian,
// rhf.rs:152-167 / uhf.rs:210-227), different mapping.  The contraction inside the primitive-quartet loop costs
// ncd * HAB * HCD FMAs, the one outside (once per bra primitive pair) nab * HAB * ncd: the pair with few functions
// belongs inside.  So here the narrow pair is the ket and
//   * one wave = one bundle: ONE bra pair (restricted to a range of its primitive pairs) against up to 64 ket pairs,
//     one ket - one whole shell quartet - per lane;
//   * everything of the bra is wave-uniform: its primitive headers and its expansion block E_ab (stored [ab][h] in
//     pairdataT) come through scalar loads, the step-3 contraction is FMAs with scalar operands, no LDS staging;
//   * a lane keeps W[ncd][HAB], the R table and the Boys values in registers and its block I[ab][cd] in a private LDS
//     column (row stride 65: conflict-free both ways); there is not a single barrier in the kernel;
//   * digestion: the J_ab block is common to the wave - it is reduced over the lanes through LDS and leaves as one
//     atomic per element; the ket-side J and the four K blocks are per-lane atomics into the replica of the wave.
#include "qc_fock_kernel.h"
#include "gen/qc_rtab.h"

#include "qc_fock_bm.h"

// The Boys table of the launch's Hermite order lives in LDS (QC_BM_TROW doubles per grid point: the 8 Taylor
// coefficients and exp(-x_k)): with one quartet per lane every table access is a 64-address gather, and the texture path
// - not the VALU - was what these kernels waited for when the rows came from global memory (rocprofv3: 60 % of the
// wave-cycles waiting to issue).  Same evaluation as qc_boys<L>.
// f64 add into LDS through the DS unit (a generic pointer would make it a flat atomic)
typedef __attribute__((address_space(3))) double qc_lds_double;
// (the row buffer belongs to one wave and every add to it is an instruction of that wave: the DS unit serves the lanes of an
// instruction in a fixed order, so these sums do not depend on timing - f64 is fine here in the fixed-point mode too)
__device__ __forceinline__ void qc_lds_add(double *p, double v) { (void)__builtin_amdgcn_ds_atomic_fadd_f64((qc_lds_double *)p, v); }
// Wave-uniform reads of the bra's pair data go through the scalar unit: a pointer taken from the kernel's argument struct carries no
// no-alias information, so the compiler reads `pd[uniform]` with a 64-lane vector load per two values (10 loads and a full memory
// latency per row of step 3); the same address in the constant address space is an s_load.  (The pair data are written by the host
// before any launch.)
typedef const __attribute__((address_space(4))) double qc_cdouble;

// -DQC_BM_TIMING (tools/build_variant_lib.sh): wave 0 of the first workgroups of a segment prints where its time went (10 ns units)
#ifdef QC_BM_TIMING
#define QC_BT(i) do { const long long t_ = wall_clock64(); tph[i] += t_ - tlast; tlast = t_; } while (0)
#define QC_BT_ARGS , long long *tph, long long &tlast
#define QC_BT_PASS , tph, tlast
#else
#define QC_BT(i) do {} while (0)
#define QC_BT_ARGS
#define QC_BT_PASS
#endif
constexpr int QC_BM_TROW = 9;
constexpr int QC_BM_TWORDS = QC_BOYS_NGRID * QC_BM_TROW;

template <int L>
__device__ __forceinline__ void qc_boys_lds(double x, const double *__restrict__ T, double (&F)[L + 1]) {
    if (x < QC_BOYS_XMAX) {
        const int k = (int)(x * (1.0 / QC_BOYS_DX) + 0.5);
        const double d = k * QC_BOYS_DX - x;
        const double *__restrict__ r = T + __umul24(k, QC_BM_TROW);   // (24-bit multiply: full rate, the 32-bit one a quarter)
        double f = r[7];
        f = fma(f, d, r[6]);
        f = fma(f, d, r[5]);
        f = fma(f, d, r[4]);
        f = fma(f, d, r[3]);
        f = fma(f, d, r[2]);
        f = fma(f, d, r[1]);
        f = fma(f, d, r[0]);
        F[L] = f;
        if constexpr (L > 0) {
            double ed = 1.0 / 40320.0;
            ed = fma(ed, d, 1.0 / 5040.0);
            ed = fma(ed, d, 1.0 / 720.0);
            ed = fma(ed, d, 1.0 / 120.0);
            ed = fma(ed, d, 1.0 / 24.0);
            ed = fma(ed, d, 1.0 / 6.0);
            ed = fma(ed, d, 0.5);
            ed = fma(ed, d, 1.0);
            ed = fma(ed, d, 1.0);
            const double ex = r[8] * ed, x2 = 2.0 * x;
#pragma unroll
            for (int n = L; n > 0; --n) F[n - 1] = fma(x2, F[n], ex) * (1.0 / (2 * n - 1));
        }
    } else {
        const double t = qc_rsqrt(x);
        F[0] = (0.5 * 1.7724538509055160273) * t;
        if constexpr (L > 0) {
            const double hr = 0.5 * (t * t);
#pragma unroll
            for (int n = 0; n < L; ++n) F[n + 1] = ((double)(2 * n + 1) * hr) * F[n];
        }
    }
}

// Step 2 against a ps ket in packed form: column c is the Cartesian axis c of the p function, its Hermite expansion is
// e0[c] on the s-type function plus e1 on the first-order function of that axis - 2 FMAs per (column, bra Hermite
// index) instead of the 4 of the dense 4 x 3 block.  `me1` = -e1 (odd ket order).
template <int LAB>
__device__ __forceinline__ void qc_step2_ps(double (&W)[3][qc_nherm(LAB)], const double (&e0)[3], const double me1,
                                            const double (&R)[qc_nherm(LAB + 1)]) {
    constexpr QcTuvTable T = qc_make_tuv();
#pragma unroll
    for (int h = 0; h < qc_nherm(LAB); ++h) {
        const int t = T.t[h], u = T.u[h], v = T.v[h];
        const double r0 = R[h];
        W[0][h] = fma(me1, R[qc_hidx(t + 1, u, v)], fma(e0[0], r0, W[0][h]));
        W[1][h] = fma(me1, R[qc_hidx(t, u + 1, v)], fma(e0[1], r0, W[1][h]));
        W[2][h] = fma(me1, R[qc_hidx(t, u, v + 1)], fma(e0[2], r0, W[2][h]));
    }
}

// Step 2 against a p.p ket in packed form (round 3), for the wave's Cartesian axis CAX of the ket's FIRST function: column d is the axis d of
// the second function, and its expansion has four terms - on the s-type Hermite function, on the first-order functions of the axes CAX
// and d, and on the second-order function of CAX + d (qc_system.cpp, packed record): 4 FMAs per (column, bra Hermite index) where the
// dense 10 x 9 block would take 10.  c0[d] = A_c D_d + [c = d] kh, cc[d] = -kh D_d, cd = -hq A_c, ccd = khh (signs: odd ket orders).
template <int LAB, int CAX>
__device__ __forceinline__ void qc_step2_pp(double (&W)[3][qc_nherm(LAB)], const double (&c0)[3], const double (&cc)[3], const double cd, const double ccd,
                                            const double (&R)[qc_nherm(LAB + 2)]) {
    constexpr QcTuvTable T = qc_make_tuv();
#pragma unroll
    for (int h = 0; h < qc_nherm(LAB); ++h) {
        const int t = T.t[h] + (CAX == 0), u = T.u[h] + (CAX == 1), v = T.v[h] + (CAX == 2);     // h + e_c
        const double r0 = R[h], rc = R[qc_hidx(t, u, v)];
        W[0][h] = fma(ccd, R[qc_hidx(t + 1, u, v)], fma(cd, R[qc_hidx(T.t[h] + 1, T.u[h], T.v[h])], fma(cc[0], rc, fma(c0[0], r0, W[0][h]))));
        W[1][h] = fma(ccd, R[qc_hidx(t, u + 1, v)], fma(cd, R[qc_hidx(T.t[h], T.u[h] + 1, T.v[h])], fma(cc[1], rc, fma(c0[1], r0, W[1][h]))));
        W[2][h] = fma(ccd, R[qc_hidx(t, u, v + 1)], fma(cd, R[qc_hidx(T.t[h], T.u[h], T.v[h] + 1)], fma(cc[2], rc, fma(c0[2], r0, W[2][h]))));
    }
}

// One pass over the ket primitives for NIJ consecutive bra primitive pairs (NIJ = 2 shares every ket load between two
// primitive quartets), followed by step 3 with the wave-uniform bra blocks.
// Latencies a pass used to expose, and what hides them now (round 3; with ket chunks of <= 8 primitives a pass is only ~16 primitive
// quartets long, so they were a third to a half of a wave's life): the bra headers of the NEXT pass are requested before the
// primitive loop (`hd` carries them from pass to pass); the first ket record is loaded once per bundle (`hk0`, `ek0`); the rows of
// step 3 come in pieces of QC_BM_PIECE scalars, each requested while the previous one is being used.
constexpr int QC_BM_PIECE = 10;
#ifndef QC_BM_AHEAD
#define QC_BM_AHEAD 1
#endif
template <int LAB, int LCD, int NIJ, int CAX = 0>
__device__ __forceinline__ void qc_bm_pass(const double *__restrict__ pd, const double *__restrict__ pdT, const double *__restrict__ Tb,
                                           const int bdoff, const int strideB, const int ij, const int ij_last, const int nab,
                                           const double *__restrict__ ketBase, const double4 hk0, const double4 ek0, const double4 ek0b, const double4 ek0c,
                                           const int K_cd, const int Kc1, const int maxK, double *const I, double (&hd)[2][4] QC_BT_ARGS) {
    constexpr int L = LAB + LCD, HAB = qc_nherm(LAB), NC = (LCD == 0) ? 1 : 3;
    constexpr int strideK = (LCD == 0) ? qc_pair_stride(0, 1) : (LCD == 1 ? 8 : 16);
    constexpr int LS = 65;
    double p[NIJ], Px[NIJ], Py[NIJ], Pz[NIJ];
#pragma unroll
    for (int u = 0; u < NIJ; ++u) { p[u] = hd[u][0]; Px[u] = hd[u][1]; Py[u] = hd[u][2]; Pz[u] = hd[u][3]; }
    // headers of the bra primitive pairs after this pass (clamped: the last pass re-reads its own)
    double hn[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qc_cdouble *bh = (qc_cdouble *)(pd + bdoff + (size_t)min(ij + NIJ + u, ij_last) * strideB);
        hn[u][0] = bh[0]; hn[u][1] = bh[1]; hn[u][2] = bh[2]; hn[u][3] = bh[3];
    }
    double W[NIJ][NC][HAB];
#pragma unroll
    for (int u = 0; u < NIJ; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < HAB; ++h) W[u][c][h] = 0.0;
    // high bras: the expansion block of this bra primitive pair (step 3 below) is requested now - coalesced, NV values per lane held in
    // registers through the primitive loop - and staged in the wave's LDS afterwards
    // (LAB = 3: at most 18 function pairs x 20 = 6 values per lane; LAB = 4 would need 20 - that costs the ss-ket launch its second wave
    // per SIMD - and its bras are mostly single primitive pairs, one pass per bundle: staged after the loop, one exposed latency)
    constexpr bool STAGE = LAB >= 3 && LCD == 0;     // (ps kets, three columns per lane: measured slower staged - 17 vs 10 us per bundle on H2O/cc-pVTZ)
    constexpr int NV = (LAB == 3) ? (18 * 20 + 63) / 64 : 1;
    constexpr bool PRE = STAGE && LAB == 3;
    const int lane = threadIdx.x & 63;
    const double *__restrict__ Eg = pdT + bdoff + (size_t)ij * strideB + 4;
    const int ne = nab * HAB;
    double est[NV];
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < NV; ++k) est[k] = (k * 64 < ne) ? Eg[min(lane + k * 64, ne - 1)] : 0.0;
    }
    // The ket primitives' records are requested QC_BM_AHEAD iterations ahead (round 4).  ss kets: pair-data block [q, Q | E]; ps kets: packed
    // record [q, Q | e0x, e0y, e0z, e1] (ketBase points into pspack, stride 8).  One iteration ahead - rounds 1-3 - left every iteration
    // waiting for its record: per-wave phase timing (-DQC_BM_TIMING) shows 0.47-0.74 us per iteration for EVERY class, (ss|ss) with its
    // 130 instructions per iteration as well as (pp|ps) with 217 - the latency of an L2 hit under load, not the arithmetic (0.15-0.3 us
    // of issue).  The loop is unrolled by the depth so that the staged records rotate through fixed registers.
    struct Rec { double4 h; double e; double4 e4, e4b, e4c; };
    auto fetch = [&](const int kl) -> Rec {
        Rec r{};
        const double *__restrict__ kbn = ketBase + (size_t)min(kl, Kc1) * strideK;
        r.h = *reinterpret_cast<const double4 *>(kbn);
        if constexpr (LCD == 0) r.e = kbn[4];
        else r.e4 = *reinterpret_cast<const double4 *>(kbn + 4);
        if constexpr (LCD == 2) { r.e4b = *reinterpret_cast<const double4 *>(kbn + 8); r.e4c = *reinterpret_cast<const double4 *>(kbn + 12); }
        return r;
    };
    auto step = [&](const int kl, const Rec &rc) {
        const bool valid = kl < K_cd;
        const double4 ck = rc.h;
        const double q = ck.x;
#pragma unroll
        for (int u = 0; u < NIJ; ++u) {
            const double X = Px[u] - ck.y, Y = Py[u] - ck.z, Z = Pz[u] - ck.w;
            const double pref = qc_rsqrt(p[u] + q);
            const double alpha = p[u] * q * (pref * pref);
            double F[L + 1], Rr[qc_nherm(L)];
            qc_boys_lds<L>(alpha * (X * X + Y * Y + Z * Z), Tb, F);
            qc_rtab<L>(alpha, X, Y, Z, F, Rr);
            const double sc = valid ? pref : 0.0;
            if constexpr (LCD == 0) {
                const double e = rc.e * sc;               // ss ket: a single Hermite function, W[h] += e R_h
#pragma unroll
                for (int h = 0; h < qc_nherm(LAB); ++h) W[u][0][h] = fma(e, Rr[h], W[u][0][h]);
            } else if constexpr (LCD == 1) {
                const double e0[3] = {rc.e4.x * sc, rc.e4.y * sc, rc.e4.z * sc};
                qc_step2_ps<LAB>(W[u], e0, -(rc.e4.w * sc), Rr);
            } else {
                // packed p.p record: e4 = (A_x, A_y, A_z, kh), e4b = (D_x, D_y, D_z, khh), e4c = (hq A_x, hq A_y, hq A_z, -)
                const double Ac = (CAX == 0 ? rc.e4.x : (CAX == 1 ? rc.e4.y : rc.e4.z)) * sc, hAc = (CAX == 0 ? rc.e4c.x : (CAX == 1 ? rc.e4c.y : rc.e4c.z)) * sc;
                const double kh = rc.e4.w * sc;
                const double c0[3] = {fma(Ac, rc.e4b.x, CAX == 0 ? kh : 0.0), fma(Ac, rc.e4b.y, CAX == 1 ? kh : 0.0), fma(Ac, rc.e4b.z, CAX == 2 ? kh : 0.0)};
                const double cc[3] = {-(kh * rc.e4b.x), -(kh * rc.e4b.y), -(kh * rc.e4b.z)};
                qc_step2_pp<LAB, CAX>(W[u], c0, cc, -hAc, rc.e4b.w * sc, Rr);
            }
        }
    };
    Rec r0{};
    r0.h = hk0; r0.e = ek0.x; r0.e4 = ek0; r0.e4b = ek0b; r0.e4c = ek0c;
    // (p.p kets keep one record ahead: their 16-double records would push the kernel past 256 registers - one wave per SIMD)
    constexpr int AHEAD = (LCD == 2) ? 1 : QC_BM_AHEAD;
    if constexpr (AHEAD == 1) {
        for (int kl = 0; kl < maxK; ++kl) {
            const Rec cur = r0;
            r0 = fetch(kl + 1);
            step(kl, cur);
        }
    } else if constexpr (AHEAD == 2) {
        Rec r1 = fetch(1);
        for (int kl = 0; kl < maxK; kl += 2) {
            { const Rec cur = r0; r0 = fetch(kl + 2); step(kl, cur); }
            if (kl + 1 < maxK) { const Rec cur = r1; r1 = fetch(kl + 3); step(kl + 1, cur); }
        }
    } else {
        Rec r1 = fetch(1), r2 = fetch(2);
        for (int kl = 0; kl < maxK; kl += 3) {
            { const Rec cur = r0; r0 = fetch(kl + 3); step(kl, cur); }
            if (kl + 1 < maxK) { const Rec cur = r1; r1 = fetch(kl + 4); step(kl + 1, cur); }
            if (kl + 2 < maxK) { const Rec cur = r2; r2 = fetch(kl + 5); step(kl + 2, cur); }
        }
    }
    QC_BT(1);
#ifdef QC_BM_TIMING
    tph[6] += maxK; tph[7] += 1;      // (iterations of the primitive loop, passes)
#endif
    // step 3 with the wave-uniform bra blocks: I[ab][c] += sum_u sum_h E_ab,ij+u[ab][h] W[u][c][h]
    if constexpr (STAGE) {
        // High bras (HAB = 20 / 35, up to 36 function pairs): a row is 2-4 scalar pieces, and one piece ahead leaves the step bound by the
        // scalar-load latency (17 us of a 36 us bundle for the (dd|ss) class).  The block is staged in the wave's LDS - coalesced
        // loads, one latency - and read back as wave-uniform (broadcast) DS reads.
        static_assert(NIJ == 1, "high bras run one bra primitive pair per pass");
        double *const Es = I - lane + nab * NC * LS;            // behind the wave's I block (qc_bm_wave_words)
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (k * 64 < ne) Es[min(lane + k * 64, ne - 1)] = est[k];   // (lanes past the end rewrite the last element with its own value)
        }
        for (int x0 = lane + (PRE ? NV * 64 : 0); x0 < ne; x0 += 4 * 64) {  // what the register stage does not hold
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = Eg[min(x0 + k * 64, ne - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + k * 64 < ne) Es[x0 + k * 64] = v[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int ab = 0; ab < nab; ++ab) {
            const double *row = Es + ab * HAB;
            double acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.0;
#pragma unroll
            for (int h = 0; h < HAB; ++h) {            // (one chain in ascending h, as in the scalar form: same roundings)
                const double e = row[h];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fma(e, W[0][c][h], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) qc_lds_add(&I[(ab * NC + c) * LS], acc[c]);
        }
        // (the next pass overwrites Es only after these reads: DS operations of a wave execute in order)
    } else {
    constexpr int PL = QC_BM_PIECE, NPC = (HAB + PL - 1) / PL, NP = NIJ * NPC;
    qc_cdouble *ET = (qc_cdouble *)(pdT + bdoff + (size_t)ij * strideB + 4);
    auto load_piece = [&](double (&e)[PL], const int ab, const int u, const int pc) {
        qc_cdouble *row = ET + (size_t)u * strideB + ab * HAB + pc * PL;
#pragma unroll
        for (int k = 0; k < PL; ++k)
            if (pc * PL + k < HAB) e[k] = row[k];
    };
    double cur[PL], nxt[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) cur[k] = nxt[k] = 0.0;
    load_piece(cur, 0, 0, 0);
    for (int ab = 0; ab < nab; ++ab) {
        double acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.0;
        const int abn = min(ab + 1, nab - 1);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int u = j / NPC, pc = j % NPC;
            if (j + 1 < NP) load_piece(nxt, ab, (j + 1) / NPC, (j + 1) % NPC);
            else load_piece(nxt, abn, 0, 0);
#pragma unroll
            for (int k = 0; k < PL; ++k)
                if (pc * PL + k < HAB) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = fma(cur[k], W[u][c][pc * PL + k], acc[c]);
                }
#pragma unroll
            for (int k = 0; k < PL; ++k) cur[k] = nxt[k];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) qc_lds_add(&I[(ab * NC + c) * LS], acc[c]);
    }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) hd[u][k] = hn[u][k];
    QC_BT(2);
}

// Device records of the work lists (built by qc_bm_device_lists, qc_fock.hip): the bundle carries what the kernel needs of its bra pair,
// the unit what it needs of the lane's ket pair - a wave used to start with three dependent trips to memory (bundle -> pair descriptors
// -> first ket record); now the bundle comes through the scalar unit and both records of the NEXT bundle are requested while the current
// one is being digested (qc_bm_segment).
typedef int qc_v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) qc_v4i qc_cv4i;
struct QcBmBundleRegs { int bra, ij_lo, ij_hi, first, nket, maxK, bdoff, offa, offb, na, nb, eq, cax; };   // wave-uniform (cax: p.p kets, axis of the first function)
__device__ __forceinline__ QcBmBundleRegs qc_bm_load_bundle(const QcBundleDev *bundles, const int b) {
    const qc_cv4i *bp = (const qc_cv4i *)(bundles + b);
    const qc_v4i x = bp[0], y = bp[1], z = bp[2];
    return QcBmBundleRegs{x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y & 0xff, (z.y >> 8) & 0xff, (z.y >> 16) & 1, z.z};
}

template <int LAB, int LCD>
__device__ __forceinline__ void qc_bm_body(const QcKernelArgs &a, const double *__restrict__ pdT, const QcBmBundleRegs &bd, const QcKetUnit ue,
                                           const int blk, const double *__restrict__ Tb, double *const Iw,
                                           const double *__restrict__ pspack, double *const rowbuf, const int rowcap QC_BT_ARGS) {
    constexpr int HAB = qc_nherm(LAB), NC = (LCD == 0) ? 1 : 3;
    constexpr int LS = 65;                                    // LDS row stride (doubles)
    constexpr bool PAIRED = 2 * NC * HAB <= 24;               // two bra primitive pairs per pass when W[2][NC][HAB] fits
    const int lane = threadIdx.x & 63;
    const double *__restrict__ pd = a.pairdata;
    const int n = a.n;
    const bool uhf = a.Dk1 != nullptr;
    const double fxscale = a.fxs ? a.fxs[0] : 0.0;           // 0: f64 atomics

    const int bra = bd.bra, ij_lo = bd.ij_lo, ij_hi = bd.ij_hi, nket = bd.nket, maxK = bd.maxK;
    const int na = bd.na, nb = bd.nb, offa = bd.offa, offb = bd.offb, bdoff = bd.bdoff;
    const int nab = na * nb;
    const int strideB = qc_pair_stride(LAB, nab);

    const bool active = lane < nket;
    // unit: ket pair | offset of its (chunk of) primitive records | c0, d0 | primitives, shape - QcKetUnit
    const int ket = ue.ket;
    const int K_cd = active ? (ue.info & 0xffff) : 0, Kc1 = max(K_cd - 1, 0);
    const bool nd1 = (ue.info >> 16) & 1;                      // (nc, nd) = (1,1), (3,1) [nd1] or (1,3)
    const bool keq = (ue.info >> 17) & 1;
    const int psperm = (ue.info >> 18) & 63;
    // (p.p kets: the lane's three columns share the function of the wave's axis in the first shell - it moves into the column base)
    const int c0 = (ue.cd & 0xffff) + (LCD == 2 ? ((ue.info >> (24 + 2 * bd.cax)) & 3) : 0), d0 = (int)((unsigned)ue.cd >> 16);
    const double *__restrict__ ketBase = ((LCD == 0) ? pd : pspack) + ue.koff;
    double *const I = Iw + lane;                              // I[x * LS], x = ab * NC + col

    for (int x = 0; x < nab * NC; ++x) I[x * LS] = 0.0;

    // first ket record of the lane's chunk (the same for every pass) and the headers of the first two bra primitive pairs
    const double4 hk0 = *reinterpret_cast<const double4 *>(ketBase);
    const double4 ek0 = (LCD >= 1) ? *reinterpret_cast<const double4 *>(ketBase + 4) : double4{ketBase[4], 0.0, 0.0, 0.0};
    const double4 ek0b = (LCD == 2) ? *reinterpret_cast<const double4 *>(ketBase + 8) : double4{0.0, 0.0, 0.0, 0.0};
    const double4 ek0c = (LCD == 2) ? *reinterpret_cast<const double4 *>(ketBase + 12) : double4{0.0, 0.0, 0.0, 0.0};
    double hd[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qc_cdouble *bh = (qc_cdouble *)(pd + bdoff + (size_t)min(ij_lo + u, ij_hi - 1) * strideB);
        hd[u][0] = bh[0]; hd[u][1] = bh[1]; hd[u][2] = bh[2]; hd[u][3] = bh[3];
    }
    QC_BT(0);
    int ij = ij_lo;
    if constexpr (LCD == 2) {
        // (the axis is wave-uniform: one of three instances of the pass)
        if (bd.cax == 0) for (; ij < ij_hi; ++ij) qc_bm_pass<LAB, LCD, 1, 0>(pd, pdT, Tb, bdoff, strideB, ij, ij_hi - 1, nab, ketBase, hk0, ek0, ek0b, ek0c, K_cd, Kc1, maxK, I, hd QC_BT_PASS);
        else if (bd.cax == 1) for (; ij < ij_hi; ++ij) qc_bm_pass<LAB, LCD, 1, 1>(pd, pdT, Tb, bdoff, strideB, ij, ij_hi - 1, nab, ketBase, hk0, ek0, ek0b, ek0c, K_cd, Kc1, maxK, I, hd QC_BT_PASS);
        else for (; ij < ij_hi; ++ij) qc_bm_pass<LAB, LCD, 1, 2>(pd, pdT, Tb, bdoff, strideB, ij, ij_hi - 1, nab, ketBase, hk0, ek0, ek0b, ek0c, K_cd, Kc1, maxK, I, hd QC_BT_PASS);
    } else {
    if constexpr (PAIRED)
        for (; ij + 1 < ij_hi; ij += 2) qc_bm_pass<LAB, LCD, 2>(pd, pdT, Tb, bdoff, strideB, ij, ij_hi - 1, nab, ketBase, hk0, ek0, ek0b, ek0c, K_cd, Kc1, maxK, I, hd QC_BT_PASS);
    for (; ij < ij_hi; ++ij) qc_bm_pass<LAB, LCD, 1>(pd, pdT, Tb, bdoff, strideB, ij, ij_hi - 1, nab, ketBase, hk0, ek0, ek0b, ek0c, K_cd, Kc1, maxK, I, hd QC_BT_PASS);
    }

    if (a.schwarz_out != nullptr) {
        // Schwarz factors: the bundles are (P|P) quartets, one ket per bundle; the largest element of the block is on its diagonal
        if (active) {
            double m = 0.0;
            for (int x = 0; x < nab * NC; ++x) m = fmax(m, fabs(I[x * LS]));
            a.schwarz_out[ket] = sqrt(m);
        }
        return;
    }
    if (a.eri_out != nullptr) {
        // materialise (ij|kl) with its 8 symmetry images (tests / stored-tensor mode; bundles are not cut along ij there)
        if (active) {
            const size_t n1 = n, n2 = n1 * n1, n3 = n2 * n1;
            double *o = a.eri_out;
            for (int ab = 0; ab < nab; ++ab) {
                const size_t i = offa + ab / nb, j = offb + ab % nb;
                for (int c = 0; c < NC; ++c) {
                    const int fc = (LCD == 0) ? 0 : (psperm >> (2 * c)) & 3;    // ps kets: column c = axis c = p function fc
                    const size_t k = c0 + (nd1 ? fc : 0), l = d0 + (nd1 ? 0 : fc);
                    const double v = I[(ab * NC + c) * LS];
                    o[i * n3 + j * n2 + k * n1 + l] = v; o[j * n3 + i * n2 + k * n1 + l] = v;
                    o[i * n3 + j * n2 + l * n1 + k] = v; o[j * n3 + i * n2 + l * n1 + k] = v;
                    o[k * n3 + l * n2 + i * n1 + j] = v; o[l * n3 + k * n2 + i * n1 + j] = v;
                    o[k * n3 + l * n2 + j * n1 + i] = v; o[l * n3 + k * n2 + j * n1 + i] = v;
                }
            }
        }
        return;
    }

    const size_t rep = (size_t)(blk % a.nrep) * a.rep_stride;
    double *G0 = a.G0 + rep, *G1 = a.G1 + rep;
    const double f = active ? (bd.eq ? 0.5 : 1.0) * (keq ? 0.5 : 1.0) * (bra == ket ? 0.5 : 1.0) : 0.0;
    // column c of this lane's ket is the function pair (c0 + kc[c], d0 + lc[c])
    int kc[NC], lc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int fc = (LCD == 0) ? 0 : (psperm >> (2 * c)) & 3;     // ps kets: column c = axis c = p function fc
        kc[c] = nd1 ? fc : 0; lc[c] = nd1 ? 0 : fc;
    }
    // this lane's ket-side density elements (Coulomb part below): requested here, used after the exchange blocks
    double dcd[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dcd[c] = active ? a.Dj[(size_t)(c0 + kc[c]) * n + d0 + lc[c]] : 0.0;

    const double fk = -a.cK * f;
    // Exchange contributions land in the rows of the bra's functions (r = i, or na + j) at the columns of this lane's ket
    // functions.  With a row buffer (LDS, per wave) they are DS atomics and the rows leave as one global atomic per
    // touched element after the bundle; without one they go to global memory directly.
    auto kadd = [&](int sp, int r, int grow, int col, double v) {
        if (rowbuf) qc_lds_add(&rowbuf[((size_t)sp * rowcap + r) * n + col], v);
        else qc_gadd(&(sp ? G1 : G0)[(size_t)grow * n + col], v, fxscale, a.fx_lo);
    };
    auto kadd3 = [&](int sp, int r, int grow, const double (&accK)[NC], const double (&accL)[NC]) {
        if constexpr (NC == 1) {
            kadd(sp, r, grow, c0, fk * accK[0]);
            kadd(sp, r, grow, d0, fk * accL[0]);
        } else if (nd1) {                                     // columns differ in k, share l
            kadd(sp, r, grow, c0 + kc[0], fk * accK[0]); kadd(sp, r, grow, c0 + kc[1], fk * accK[1]); kadd(sp, r, grow, c0 + kc[2], fk * accK[2]);
            kadd(sp, r, grow, d0, fk * (accL[0] + accL[1] + accL[2]));
        } else {                                              // columns differ in l, share k
            kadd(sp, r, grow, c0, fk * (accK[0] + accK[1] + accK[2]));
            kadd(sp, r, grow, d0 + lc[0], fk * accL[0]); kadd(sp, r, grow, d0 + lc[1], fk * accL[1]); kadd(sp, r, grow, d0 + lc[2], fk * accL[2]);
        }
    };
    // ---- exchange blocks first (they read I), per spin: Gt_ik -= cK f sum_jl I D_jl and the il / jk / jl images.
    // The density rows of the functions of the OTHER bra shell are gathered first (one memory latency per half instead of one per
    // target row: the elements do not depend on the target), then every target row is a loop over LDS reads.  RB covers a whole
    // shell, so a target element is ONE sum over the other shell's functions in ascending order, as before (the O2 triplet run of
    // the tests sits on a saddle and follows these roundings).
    constexpr int RB = (LAB <= 2) ? 6 : 10;
    auto exchange_half = [&](const double *__restrict__ Dk, const int s, const bool on_a) {
        const int nt = on_a ? na : nb, no = on_a ? nb : na;              // target shell / other shell (wave-uniform)
        const int offo = on_a ? offb : offa;
        for (int o0 = 0; o0 < no; o0 += RB) {
            double dK[RB][NC], dL[RB][NC];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int c = 0; c < NC; ++c) dK[r][c] = dL[r][c] = 0.0;
                if (o0 + r < no) {
                    const double *__restrict__ Drow = Dk + (size_t)(offo + o0 + r) * n;
#pragma unroll
                    for (int c = 0; c < NC; ++c) { dK[r][c] = Drow[d0 + lc[c]]; dL[r][c] = Drow[c0 + kc[c]]; }
                }
            }
            for (int t = 0; t < nt; ++t) {
                double accK[NC], accL[NC];                   // G[target, c_k] and G[target, d_l] partial sums
#pragma unroll
                for (int c = 0; c < NC; ++c) accK[c] = accL[c] = 0.0;
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    if (o0 + r < no) {
                        const int o = o0 + r;
                        const int ab = on_a ? t * nb + o : o * nb + t;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const double v = I[(ab * NC + c) * LS];
                            accK[c] = fma(v, dK[r][c], accK[c]);
                            accL[c] = fma(v, dL[r][c], accL[c]);
                        }
                    }
                }
                kadd3(s, on_a ? t : na + t, (on_a ? offa : offb) + t, accK, accL);
            }
        }
    };
    if (active) {
        for (int s = 0; s < (uhf ? 2 : 1); ++s) {
            const double *__restrict__ Dk = s ? a.Dk1 : a.Dk0;
            exchange_half(Dk, s, true);                       // targets on the bra functions a_i: D[b_j, d_l] and D[b_j, c_k]
            exchange_half(Dk, s, false);                      // targets on the bra functions b_j: D[a_i, d_l] and D[a_i, c_k]
        }
    }

    QC_BT(3);
    // ---- Coulomb blocks: Gt_cd += 2f sum_ab I D_ab (per lane);  Gt_ab += 2f sum_cd I D_cd (reduced over the wave)
    const double fj = 2.0 * f;
    double jcd[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) jcd[c] = 0.0;
    {
        // D_ab is wave-uniform: scalar loads, one element ahead (the density is written by earlier launches only)
        qc_cdouble *Dj = (qc_cdouble *)a.Dj;
        int bi = 0, bj = 0;
        double dab = Dj[(size_t)offa * n + offb];
        for (int ab = 0; ab < nab; ++ab) {
            int bjn = bj + 1, bin = bi;
            if (bjn == nb) { bjn = 0; ++bin; }
            if (bin == na) { bin = na - 1; bjn = nb - 1; }
            const double dnx = Dj[(size_t)(offa + bin) * n + offb + bjn];
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double v = I[(ab * NC + c) * LS];
                jcd[c] = fma(v, dab, jcd[c]);
                t = fma(v, dcd[c], t);
            }
            I[(ab * NC) * LS] = fj * t;                       // own column: this lane's share of Gt_ab
            dab = dnx; bi = bin; bj = bjn;
        }
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const size_t o = (size_t)(c0 + kc[c]) * n + d0 + lc[c];
            qc_gadd2(&G0[o], &G1[o], uhf, fj * jcd[c], fxscale, a.fx_lo);
        }
    }
    // same-wave LDS hand-off (DS operations of a wave execute in order): lane ab sums row ab over the 64 columns
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ab = lane; ab < nab; ab += 64) {
        const double *row = Iw + (size_t)(ab * NC) * LS;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int j = 0; j < 64; j += 2) { s0 += row[j]; s1 += row[j + 1]; }
        const size_t o = (size_t)(offa + ab / nb) * n + offb + ab % nb;
        qc_gadd2(&G0[o], &G1[o], uhf, s0 + s1, fxscale, a.fx_lo);
    }
    QC_BT(4);
}

// Workgroup = qc_bm_waves(LCD, HI) independent waves sharing the LDS Boys table of their segment's Hermite order; every wave
// walks the segment's bundle list with the stride of the launch (bundles are sorted longest first).
template <int LAB, int LCD, int NW>
__device__ __forceinline__ void qc_bm_segment(const QcBmArgs &a, const int s, const int wg, const int nwg) {
    extern __shared__ double lds[];
    constexpr int L = LAB + LCD;
    const int tid = threadIdx.x, wave = tid >> 6;
    {
        const double *__restrict__ rows = a.base.boys + (size_t)L * QC_BOYS_NGRID * 8;
        const double *__restrict__ exk = a.base.boys + (size_t)(QC_LTOT + 1) * QC_BOYS_NGRID * 8;
        // eight loads of a thread in flight together (an L2 round trip per batch, not per element)
        const int nt = blockDim.x;
        for (int i0 = tid; i0 < QC_BM_TWORDS; i0 += 8 * nt) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = min(i0 + j * nt, QC_BM_TWORDS - 1);
                const int k = i / QC_BM_TROW, r = i - k * QC_BM_TROW;
                v[j] = (r < 8) ? rows[k * 8 + r] : exk[k];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 + j * nt < QC_BM_TWORDS) lds[i0 + j * nt] = v[j];
        }
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;                           // <= NW: fewer when the LDS blocks of NW waves would not fit
    const int n = a.base.n, lane = tid & 63;
    const int rowcap = a.seg_rows[s], nsp = a.base.Dk1 ? 2 : 1;
    const int rowwords = a.use_rowbuf ? nsp * rowcap * n : 0;
    double *const wbase = lds + ((QC_BM_TWORDS + 1) & ~1) + (size_t)wave * (a.seg_iwords[s] + rowwords);
    double *const Iw = wbase;
    double *const rowbuf = rowwords ? wbase + a.seg_iwords[s] : nullptr;
    for (int x = lane; x < rowwords; x += 64) rowbuf[x] = 0.0;
    const QcBundleDev *__restrict__ bundles = a.seg_bundles[s];
    const QcKetUnit *__restrict__ units = a.seg_ketlist[s];
    // the rows a wave has accumulated for the bundle of bra pair `bra` leave as one global atomic per touched element:
    // the 64 kets of a bundle share most of their functions, so the lanes' contributions combine 2-5x in LDS first
    const double fxscale = a.base.fxs ? a.base.fxs[0] : 0.0;
    auto flush_rows = [&](const QcBmBundleRegs &pb) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int sp = 0; sp < nsp; ++sp) {
            double *G = (sp ? a.base.G1 : a.base.G0) + (size_t)(((wg * nw + wave) % a.base.nrep)) * a.base.rep_stride;
            for (int r = 0; r < pb.na + pb.nb; ++r) {
                const int grow = r < pb.na ? pb.offa + r : pb.offb + r - pb.na;
                double *row = rowbuf + ((size_t)sp * rowcap + r) * n;
                for (int col = lane; col < n; col += 64) {
                    const double v = row[col];
                    if (v != 0.0) { qc_gadd(&G[(size_t)grow * n + col], v, fxscale, a.base.fx_lo); row[col] = 0.0; }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int nb = a.seg_nbundles[s];
#ifdef QC_BM_TIMING
    long long tph[8] = {}, tlast = wall_clock64();
    int nbun = 0;
#endif
    const int b0 = __builtin_amdgcn_readfirstlane(wg * nw + wave), bstep = nwg * nw;
    if (b0 >= nb) return;
    QcBmBundleRegs bd = qc_bm_load_bundle(bundles, b0);
    QcKetUnit ue = units[bd.first + min(lane, bd.nket - 1)];
    for (int b = b0; b < nb; b += bstep) {
        // records of the wave's next bundle: the bundle now (scalar), its unit while the rows of this one are flushed
        const QcBmBundleRegs bdn = qc_bm_load_bundle(bundles, min(b + bstep, nb - 1));
        qc_bm_body<LAB, LCD>(a.base, a.pairdataT, bd, ue, b, lds, Iw, a.pspack, rowbuf, rowcap QC_BT_PASS);
        const QcKetUnit uen = units[bdn.first + min(lane, bdn.nket - 1)];
        if (rowbuf) flush_rows(bd);
        bd = bdn; ue = uen;
        QC_BT(5);
#ifdef QC_BM_TIMING
        ++nbun;
#endif
    }
#ifdef QC_BM_TIMING
    if (lane == 0 && wave == 0 && wg < 2 && a.base.eri_out == nullptr && a.base.schwarz_out == nullptr)
        printf("bm<%d,%d> seg wg %d: %d bundles (of %d)  setup %lld  k-loop %lld  step3 %lld  digestK %lld  digestJ %lld  flush %lld  [10 ns]  k-iterations %lld in %lld passes: %.0f ns per iteration\n", LAB, LCD, wg, nbun, nb,
               tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6], tph[7], tph[6] ? 10.0 * (double)tph[1] / (double)tph[6] : 0.0);
#endif
}

template <int LCD, int HI>
__global__ __launch_bounds__(qc_bm_waves(LCD, HI) * 64) void qc_fock_bm_kernel(const QcBmArgs a) {
    if (qc_build_cancelled(a.base)) return;
    qc_tl_stamp(a.base.tl, 0);
    int s = 0;
    while (s + 1 < a.nseg && (int)blockIdx.x >= a.seg_end[s]) ++s;
    const int b0 = s ? a.seg_end[s - 1] : 0;
    const int wg = blockIdx.x - b0, nwg = a.seg_end[s] - b0;
#define QC_BM_CASE(LAB) case LAB: qc_bm_segment<LAB, LCD, qc_bm_waves(LCD, HI)>(a, s, wg, nwg); break;
    if constexpr (LCD == 0 && HI == 0) { switch (a.seg_lab[s]) { QC_BM_CASE(0) QC_BM_CASE(1) QC_BM_CASE(2) default: break; } }
    if constexpr (LCD == 0 && HI == 1) { switch (a.seg_lab[s]) { QC_BM_CASE(3) QC_BM_CASE(4) default: break; } }
    if constexpr (LCD == 1 && HI == 0) { switch (a.seg_lab[s]) { QC_BM_CASE(1) QC_BM_CASE(2) default: break; } }
    if constexpr (LCD == 1 && HI == 1) { switch (a.seg_lab[s]) { QC_BM_CASE(3) QC_BM_CASE(4) default: break; } }
    if constexpr (LCD == 2 && HI == 0) { switch (a.seg_lab[s]) { QC_BM_CASE(2) default: break; } }
#undef QC_BM_CASE
    // <3, 0>: ONE launch for the ss kets against the high bras and the ps kets against the low bras (round 3) - both run at two waves per
    // SIMD and one workgroup per CU (129 / 98 KB of LDS on H2O/cc-pVTZ), so sharing a launch costs neither; their bundles fill the
    // chip together instead of following each other on a stream (50 + 83 us in-build there)
    if constexpr (LCD == 3 && HI == 0) {
        switch ((a.seg_lcd[s] << 4) | a.seg_lab[s]) {
            case 0x03: qc_bm_segment<3, 0, qc_bm_waves(0, 1)>(a, s, wg, nwg); break;
            case 0x04: qc_bm_segment<4, 0, qc_bm_waves(0, 1)>(a, s, wg, nwg); break;
            case 0x11: qc_bm_segment<1, 1, qc_bm_waves(1, 0)>(a, s, wg, nwg); break;
            case 0x12: qc_bm_segment<2, 1, qc_bm_waves(1, 0)>(a, s, wg, nwg); break;
            default: break;
        }
    }
    qc_tl_stamp(a.base.tl, 1);
}

template <int LCD, int HI>
static int launch_bm(int grid, int nwaves, size_t lds, hipStream_t st, const QcBmArgs &a) {
    auto kern = qc_fock_bm_kernel<LCD, HI>;
    static std::atomic<bool> raised{false};       // once per instantiation, to the device maximum (see qc_launch_tier)
    if (lds > 48 * 1024 && !raised.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QC_LDS_MAX);
        if (e != hipSuccess) return QC_ERR_HIP;
        raised.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nwaves * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

// The exchange rows of a bundle are summed in LDS with f64 DS atomics, several lanes of ONE instruction adding to one address (qc_lds_add):
// bitwise reproducible builds need the DS unit to serve such lanes in an order that does not depend on timing.  No manual says so; every
// device seen does it.  This probe asks the device at hand: 64 lanes add values of very different magnitudes (the sum depends on the
// order) to one word, 64 times over, with another wave of the workgroup hammering the same LDS bank meanwhile; all 64 sums must be the same
// bits.  A device that fails is served without the row buffer (direct fixed-point global atomics: order-independent by construction).
__global__ __launch_bounds__(128) void qc_ds_order_probe_kernel(double *out) {
    __shared__ double acc[64 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 1) {                                   // traffic on the same banks, other addresses
        for (int rep = 0; rep < 4096; ++rep) qc_lds_add(&acc[64 + ((lane * 7 + rep) & 63)], 1.0);
        return;
    }
    const double v = ldexp(1.0 + 0.001 * lane, (lane * 37) % 53 - 20) * ((lane & 1) ? -1.0 : 1.0);
    for (int rep = 0; rep < 64; ++rep) {
        double *word = &acc[rep & 1];
        if (lane == 0) *word = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((lane + rep) & 3) qc_lds_add(&acc[2 + (lane & 31)], 1.0);      // (an unrelated add with a partial mask in front)
        qc_lds_add(word, v);                           // the probed instruction: 64 lanes, one address
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) out[rep] = *word;
    }
}
int qc_ds_order_probe(hipStream_t st, double *d_out64) {
    hipLaunchKernelGGL(qc_ds_order_probe_kernel, dim3(1), dim3(128), 0, st, d_out64);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

int qc_launch_bm(int lcd, int hi, int grid, int nwaves, size_t lds, hipStream_t st, const QcBmArgs &a) {
    if (lcd == 3) return hi ? QC_ERR_UNSUPPORTED : launch_bm<3, 0>(grid, nwaves, lds, st, a);
    if (lcd == 2) return hi ? QC_ERR_UNSUPPORTED : launch_bm<2, 0>(grid, nwaves, lds, st, a);
    if (lcd == 0) return hi ? launch_bm<0, 1>(grid, nwaves, lds, st, a) : launch_bm<0, 0>(grid, nwaves, lds, st, a);
    return hi ? launch_bm<1, 1>(grid, nwaves, lds, st, a) : launch_bm<1, 0>(grid, nwaves, lds, st, a);
}
