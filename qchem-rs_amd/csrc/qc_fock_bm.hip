// qc_fock_bm.hip - "bra-major" ERI + Fock digestion kernels for the quartet classes whose narrower pair is an ss or a
// ps pair and whose total Hermite order is <= QC_LREG (all tables in registers).  gfx950 / wave64.
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
#include "gen/qc_step2_lab0.h"
#include "gen/qc_step2_lab1.h"
#include "gen/qc_step2_lab2.h"
#include "gen/qc_step2_lab3.h"
#include "gen/qc_step2_lab4.h"

#include "qc_fock_bm.h"

template <int LAB, int LCD>
__device__ __forceinline__ void qc_bm_body(const QcKernelArgs &a, const double *__restrict__ pdT, const QcBundle *__restrict__ bundles,
                                           const int *__restrict__ ketlist, const int blk) {
    constexpr int L = LAB + LCD, HAB = qc_nherm(LAB), NC = (LCD == 0) ? 1 : 3;
    constexpr int strideK = qc_pair_stride(LCD, NC);
    constexpr int LS = 65;                                    // LDS row stride (doubles)
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const double *__restrict__ pd = a.pairdata;
    const int n = a.n;
    const bool uhf = a.Dk1 != nullptr;

    const QcBundle bd = bundles[blk];
    const int bra = __builtin_amdgcn_readfirstlane(bd.bra);
    const int ij_lo = __builtin_amdgcn_readfirstlane(bd.ij_lo), ij_hi = __builtin_amdgcn_readfirstlane(bd.ij_hi);
    const int first = __builtin_amdgcn_readfirstlane(bd.first), nket = __builtin_amdgcn_readfirstlane(bd.nket);
    const int maxK = __builtin_amdgcn_readfirstlane(bd.maxK);
    const QcPairDesc pb = a.pairs[bra];
    const int na = __builtin_amdgcn_readfirstlane(pb.na), nb = __builtin_amdgcn_readfirstlane(pb.nb);
    const int offa = __builtin_amdgcn_readfirstlane(pb.offa), offb = __builtin_amdgcn_readfirstlane(pb.offb);
    const int bdoff = __builtin_amdgcn_readfirstlane(pb.doff);
    const int nab = na * nb;
    const int strideB = qc_pair_stride(LAB, nab);

    const bool active = lane < nket;
    const int ket = ketlist[first + (active ? lane : 0)];
    const QcPairDesc pk = a.pairs[ket];
    const int K_cd = active ? pk.K : 0;
    const double *__restrict__ ketBase = pd + pk.doff;
    double *const I = lds + lane;                             // I[x * LS], x = ab * NC + col

    for (int x = 0; x < nab * NC; ++x) I[x * LS] = 0.0;

    for (int ij = ij_lo; ij < ij_hi; ++ij) {
        const double *__restrict__ bh = pd + bdoff + (size_t)ij * strideB;
        const double p = bh[0], Px = bh[1], Py = bh[2], Pz = bh[3];
        double W[NC][HAB];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < HAB; ++h) W[c][h] = 0.0;
        double4 hk = *reinterpret_cast<const double4 *>(ketBase);
        for (int kl = 0; kl < maxK; ++kl) {
            const bool valid = kl < K_cd;
            const double4 ck = hk;
            const double *__restrict__ kb = ketBase + (size_t)(valid ? kl : 0) * strideK;
            if (kl + 1 < K_cd) hk = *reinterpret_cast<const double4 *>(ketBase + (size_t)(kl + 1) * strideK);
            const double q = ck.x;
            const double X = Px - ck.y, Y = Py - ck.z, Z = Pz - ck.w;
            const double pref = rsqrt(p + q);
            const double alpha = p * q * (pref * pref);
            double F[L + 1], Rr[qc_nherm(L)];
            qc_boys<L>(alpha * (X * X + Y * Y + Z * Z), a.boys, F);
            qc_rtab<L>(alpha, X, Y, Z, F, Rr);
            const double sc = valid ? pref : 0.0;
            if constexpr (LCD == 0) {
                double e[1] = {kb[4] * sc};
                qc_step2<LAB, 0>(W[0], e, Rr);
            } else {
                // ket block: E[h][col], 12 consecutive doubles behind the 32-byte header
                const double4 e0 = *reinterpret_cast<const double4 *>(kb + 4), e1 = *reinterpret_cast<const double4 *>(kb + 8),
                              e2 = *reinterpret_cast<const double4 *>(kb + 12);
                const double ev[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double e[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) e[h] = ev[h * 3 + c] * sc;
                    qc_step2<LAB, 1>(W[c], e, Rr);
                }
            }
        }
        // step 3 with the wave-uniform bra block: I[ab][c] += sum_h E_ab,ij[ab][h] W[c][h]
        const double *__restrict__ ET = pdT + bdoff + (size_t)ij * strideB + 4;
        for (int ab = 0; ab < nab; ++ab) {
            const double *__restrict__ row = ET + ab * HAB;
            double acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.0;
#pragma unroll
            for (int h = 0; h < HAB; ++h) {
                const double ev = row[h];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fma(ev, W[c][h], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) I[(ab * NC + c) * LS] += acc[c];
        }
    }

    const int nd = pk.nb;                                     // (nc, nd) = (1,1), (3,1) or (1,3)
    const int c0 = pk.offa, d0 = pk.offb;
    if (a.eri_out != nullptr) {
        // materialise (ij|kl) with its 8 symmetry images (tests / stored-tensor mode; bundles are not cut along ij there)
        if (active) {
            const size_t n1 = n, n2 = n1 * n1, n3 = n2 * n1;
            double *o = a.eri_out;
            for (int ab = 0; ab < nab; ++ab) {
                const size_t i = offa + ab / nb, j = offb + ab % nb;
                for (int c = 0; c < NC; ++c) {
                    const size_t k = c0 + (nd == 1 ? c : 0), l = d0 + (nd == 1 ? 0 : c);
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
    const double f = active ? (pb.shA_eq_shB ? 0.5 : 1.0) * (pk.shA_eq_shB ? 0.5 : 1.0) * (bra == ket ? 0.5 : 1.0) : 0.0;
    // column c of this lane's ket is the function pair (c0 + kc[c], d0 + lc[c])
    int kc[NC], lc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { kc[c] = (nd == 1) ? c : 0; lc[c] = (nd == 1) ? 0 : c; }

    // ---- exchange blocks first (they read I), per spin: Gt_ik -= cK f sum_jl I D_jl and the il / jk / jl images
    const double fk = -a.cK * f;
    if (active) {
        for (int s = 0; s < (uhf ? 2 : 1); ++s) {
            const double *__restrict__ Dk = s ? a.Dk1 : a.Dk0;
            double *Gs = s ? G1 : G0;
            // targets on the bra function a_i: needs D[b_j, d_l] and D[b_j, c_k]
            for (int i = 0; i < na; ++i) {
                double accK[NC], accL[NC];                   // G[a_i, c_k] and G[a_i, d_l] partial sums
#pragma unroll
                for (int c = 0; c < NC; ++c) accK[c] = accL[c] = 0.0;
                for (int j = 0; j < nb; ++j) {
                    const double *__restrict__ Drow = Dk + (size_t)(offb + j) * n;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double v = I[((i * nb + j) * NC + c) * LS];
                        accK[c] = fma(v, Drow[d0 + lc[c]], accK[c]);
                        accL[c] = fma(v, Drow[c0 + kc[c]], accL[c]);
                    }
                }
                double *Grow = Gs + (size_t)(offa + i) * n;
                if constexpr (NC == 1) {
                    unsafeAtomicAdd(&Grow[c0], fk * accK[0]);
                    unsafeAtomicAdd(&Grow[d0], fk * accL[0]);
                } else if (nd == 1) {                         // columns differ in k, share l
                    unsafeAtomicAdd(&Grow[c0 + 0], fk * accK[0]); unsafeAtomicAdd(&Grow[c0 + 1], fk * accK[1]); unsafeAtomicAdd(&Grow[c0 + 2], fk * accK[2]);
                    unsafeAtomicAdd(&Grow[d0], fk * (accL[0] + accL[1] + accL[2]));
                } else {                                      // columns differ in l, share k
                    unsafeAtomicAdd(&Grow[c0], fk * (accK[0] + accK[1] + accK[2]));
                    unsafeAtomicAdd(&Grow[d0 + 0], fk * accL[0]); unsafeAtomicAdd(&Grow[d0 + 1], fk * accL[1]); unsafeAtomicAdd(&Grow[d0 + 2], fk * accL[2]);
                }
            }
            // targets on the bra function b_j: needs D[a_i, d_l] and D[a_i, c_k]
            for (int j = 0; j < nb; ++j) {
                double accK[NC], accL[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) accK[c] = accL[c] = 0.0;
                for (int i = 0; i < na; ++i) {
                    const double *__restrict__ Drow = Dk + (size_t)(offa + i) * n;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const double v = I[((i * nb + j) * NC + c) * LS];
                        accK[c] = fma(v, Drow[d0 + lc[c]], accK[c]);
                        accL[c] = fma(v, Drow[c0 + kc[c]], accL[c]);
                    }
                }
                double *Grow = Gs + (size_t)(offb + j) * n;
                if constexpr (NC == 1) {
                    unsafeAtomicAdd(&Grow[c0], fk * accK[0]);
                    unsafeAtomicAdd(&Grow[d0], fk * accL[0]);
                } else if (nd == 1) {
                    unsafeAtomicAdd(&Grow[c0 + 0], fk * accK[0]); unsafeAtomicAdd(&Grow[c0 + 1], fk * accK[1]); unsafeAtomicAdd(&Grow[c0 + 2], fk * accK[2]);
                    unsafeAtomicAdd(&Grow[d0], fk * (accL[0] + accL[1] + accL[2]));
                } else {
                    unsafeAtomicAdd(&Grow[c0], fk * (accK[0] + accK[1] + accK[2]));
                    unsafeAtomicAdd(&Grow[d0 + 0], fk * accL[0]); unsafeAtomicAdd(&Grow[d0 + 1], fk * accL[1]); unsafeAtomicAdd(&Grow[d0 + 2], fk * accL[2]);
                }
            }
        }
    }

    // ---- Coulomb blocks: Gt_cd += 2f sum_ab I D_ab (per lane);  Gt_ab += 2f sum_cd I D_cd (reduced over the wave)
    const double fj = 2.0 * f;
    double dcd[NC], jcd[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dcd[c] = active ? a.Dj[(size_t)(c0 + kc[c]) * n + d0 + lc[c]] : 0.0; jcd[c] = 0.0; }
    for (int ab = 0; ab < nab; ++ab) {
        const double dab = a.Dj[(size_t)(offa + ab / nb) * n + offb + ab % nb];      // wave-uniform
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double v = I[(ab * NC + c) * LS];
            jcd[c] = fma(v, dab, jcd[c]);
            t = fma(v, dcd[c], t);
        }
        I[(ab * NC) * LS] = fj * t;                           // own column: this lane's share of Gt_ab
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const size_t o = (size_t)(c0 + kc[c]) * n + d0 + lc[c];
            unsafeAtomicAdd(&G0[o], fj * jcd[c]);
            if (uhf) unsafeAtomicAdd(&G1[o], fj * jcd[c]);
        }
    }
    // same-wave LDS hand-off (DS operations of a wave execute in order): lane ab sums row ab over the 64 columns
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ab = lane; ab < nab; ab += 64) {
        const double *row = lds + (size_t)(ab * NC) * LS;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int j = 0; j < 64; j += 2) { s0 += row[j]; s1 += row[j + 1]; }
        const size_t o = (size_t)(offa + ab / nb) * n + offb + ab % nb;
        unsafeAtomicAdd(&G0[o], s0 + s1);
        if (uhf) unsafeAtomicAdd(&G1[o], s0 + s1);
    }
}

template <int LCD, int HI>
__global__ __launch_bounds__(64) void qc_fock_bm_kernel(const QcBmArgs a) {
    int s = 0;
    while (s + 1 < a.nseg && (int)blockIdx.x >= a.seg_end[s]) ++s;
    const int blk = blockIdx.x - (s ? a.seg_end[s - 1] : 0);
    const QcBundle *bundles = a.seg_bundles[s];
    const int *ketlist = a.seg_ketlist[s];
#define QC_BM_CASE(LAB) case LAB: qc_bm_body<LAB, LCD>(a.base, a.pairdataT, bundles, ketlist, blk); break;
    if constexpr (LCD == 0 && HI == 0) { switch (a.seg_lab[s]) { QC_BM_CASE(0) QC_BM_CASE(1) QC_BM_CASE(2) default: break; } }
    if constexpr (LCD == 0 && HI == 1) { switch (a.seg_lab[s]) { QC_BM_CASE(3) QC_BM_CASE(4) default: break; } }
    if constexpr (LCD == 1 && HI == 0) { switch (a.seg_lab[s]) { QC_BM_CASE(1) QC_BM_CASE(2) default: break; } }
    if constexpr (LCD == 1 && HI == 1) { switch (a.seg_lab[s]) { QC_BM_CASE(3) default: break; } }
#undef QC_BM_CASE
}

template <int LCD, int HI>
static int launch_bm(int grid, size_t lds, hipStream_t st, const QcBmArgs &a) {
    auto kern = qc_fock_bm_kernel<LCD, HI>;
    static size_t lds_allowed = 48 * 1024;
    if (lds > lds_allowed) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return QC_ERR_HIP;
        lds_allowed = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, a);
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

int qc_launch_bm(int lcd, int hi, int grid, size_t lds, hipStream_t st, const QcBmArgs &a) {
    if (lcd == 0) return hi ? launch_bm<0, 1>(grid, lds, st, a) : launch_bm<0, 0>(grid, lds, st, a);
    return hi ? launch_bm<1, 1>(grid, lds, st, a) : launch_bm<1, 0>(grid, lds, st, a);
}
