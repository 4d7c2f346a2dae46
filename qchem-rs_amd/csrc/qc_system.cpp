// qc_system.cpp - host model of a molecule + basis for the MI355X Hartree-Fock path.
//
// Replaces what the reference receives as `&MolecularSystem` from molint (main.rs:76-77; members read at
// rhf.rs:36-37) and precomputes, once per geometry, everything the HIP kernels stream: normalised shells, the
// shell-pair list with per-primitive-pair Hermite expansion matrices (spherical transform, contraction coefficients
// and the pair part of the ERI prefactor folded in), the class-sorted unique-quartet task lists and their
// per-rank shards.  Also holds the host implementation of molint::overlap/kinetic/nuclear (rhf.rs:41-43), which
// SURVEY.md 8f ranks as "next" for the GPU.  Integral formulas: McMurchie-Davidson.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

#include <cstdlib>

#include "qc_internal.h"

namespace {

double dfact(int n) { double r = 1.0; for (; n > 1; n -= 2) r *= n; return r; }
double binom(int n, int k) { if (k < 0 || k > n) return 0.0; double r = 1.0; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; }

struct Cart { int x, y, z; };
std::vector<Cart> cart_list(int L) {
    std::vector<Cart> v;
    for (int lx = L; lx >= 0; --lx) for (int ly = L - lx; ly >= 0; --ly) v.push_back({lx, ly, L - lx - ly});
    return v;
}

// real solid harmonic S_lm in Cartesian monomials, Helgaker-Jorgensen-Olsen eq. 6.4.47 (scale fixed afterwards)
std::vector<double> solid_row(int l, int m, const std::vector<Cart> &cl) {
    std::vector<double> row(cl.size(), 0.0);
    const int am = std::abs(m), neg = m < 0 ? 1 : 0;
    for (int t = 0; 2 * t <= l - am; ++t)
        for (int u = 0; u <= t; ++u)
            for (int k = 0; 2 * k + neg <= am; ++k) {          // 2v = 2k + neg
                const int tv = 2 * k + neg;
                double c = ((t + k) & 1 ? -1.0 : 1.0) * std::pow(0.25, t) * binom(l, t) * binom(l - t, am + t) * binom(t, u) * binom(am, tv);
                const int ex = 2 * t + am - 2 * u - tv, ey = 2 * u + tv, ez = l - 2 * t - am;
                if (ex < 0) continue;
                for (size_t i = 0; i < cl.size(); ++i)
                    if (cl[i].x == ex && cl[i].y == ey && cl[i].z == ez) row[i] += c;
            }
    return row;
}

double mono_overlap_1d(int n, double p) { return (n & 1) ? 0.0 : dfact(n - 1) / std::pow(2.0 * p, n / 2) * std::sqrt(M_PI / p); }

// 1-D Hermite expansion coefficients E[i][j][t], i<=imax, j<=jmax (flat, stride helpers)
struct E1 {
    int imax, jmax, tdim;
    std::vector<double> v;
    E1(int im, int jm) : imax(im), jmax(jm), tdim(im + jm + 2), v((size_t)(im + 1) * (jm + 1) * (im + jm + 2), 0.0) {}
    double &at(int i, int j, int t) { return v[((size_t)i * (jmax + 1) + j) * tdim + t]; }
    double get(int i, int j, int t) const { return (t < 0 || t > i + j) ? 0.0 : v[((size_t)i * (jmax + 1) + j) * tdim + t]; }
};
E1 hermite_e(int imax, int jmax, double a, double b, double Q) {
    E1 E(imax, jmax);
    const double p = a + b, h = 0.5 / p, xpa = -b / p * Q, xpb = a / p * Q;
    E.at(0, 0, 0) = std::exp(-a * b / p * Q * Q);
    for (int i = 1; i <= imax; ++i)
        for (int t = 0; t <= i; ++t)
            E.at(i, 0, t) = h * E.get(i - 1, 0, t - 1) + xpa * E.get(i - 1, 0, t) + (t + 1) * E.get(i - 1, 0, t + 1);
    for (int j = 1; j <= jmax; ++j)
        for (int i = 0; i <= imax; ++i)
            for (int t = 0; t <= i + j; ++t)
                E.at(i, j, t) = h * E.get(i, j - 1, t - 1) + xpb * E.get(i, j - 1, t) + (t + 1) * E.get(i, j - 1, t + 1);
    return E;
}

// Hermite Coulomb integrals R^0_tuv, t+u+v <= L (host, for the nuclear-attraction matrix)
void hermite_r_host(int L, double alpha, const double PC[3], std::vector<double> &R0) {
    std::vector<std::vector<double>> W(L + 1, std::vector<double>(qc_nherm(L), 0.0));
    std::vector<double> F(L + 1);
    qc_boys_host(L, alpha * (PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]), F.data());
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
    R0 = W[0];
}

}  // namespace

// F_n(x), n = 0..nmax: Kummer series at nmax + downward recursion; erf + upward recursion for large x.
void qc_boys_host(int nmax, double x, double *F) {
    const double ex = std::exp(-x);
    if (x < 38.0) {
        double term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 500; ++k) { term *= 2.0 * x / (2 * nmax + 2 * k + 1); sum += term; if (term < 1e-18 * sum) break; }
        F[nmax] = ex * sum;
        for (int n = nmax; n > 0; --n) F[n - 1] = (2.0 * x * F[n] + ex) / (2 * n - 1);
    } else {
        F[0] = 0.5 * std::sqrt(M_PI / x) * std::erf(std::sqrt(x));
        for (int n = 0; n < nmax; ++n) F[n + 1] = ((2 * n + 1) * F[n] - ex) / (2.0 * x);
    }
}

static void normalise_shell(QcShell &sh) {
    const int L = sh.L;
    auto cl = cart_list(L);
    sh.ncart = (int)cl.size();
    sh.nfunc = sh.pure ? 2 * L + 1 : sh.ncart;
    for (int i = 0; i < sh.nprim; ++i) {
        const double a = sh.exps[i];
        sh.coefs[i] *= std::pow(2.0 * a / M_PI, 0.75) * std::pow(4.0 * a, 0.5 * L) / std::sqrt(dfact(2 * L - 1));
    }
    sh.T.assign((size_t)sh.nfunc * sh.ncart, 0.0);
    if (sh.pure) {
        for (int m = -L; m <= L; ++m) { auto r = solid_row(L, m, cl); std::copy(r.begin(), r.end(), sh.T.begin() + (size_t)(m + L) * sh.ncart); }
    } else {
        for (int c = 0; c < sh.ncart; ++c) sh.T[(size_t)c * sh.ncart + c] = 1.0;
    }
    // contracted self-overlap of the monomials, then scale each function to unit norm
    std::vector<double> M((size_t)sh.ncart * sh.ncart, 0.0);
    for (int c1 = 0; c1 < sh.ncart; ++c1)
        for (int c2 = 0; c2 < sh.ncart; ++c2) {
            double s = 0.0;
            for (int i = 0; i < sh.nprim; ++i)
                for (int j = 0; j < sh.nprim; ++j) {
                    const double p = sh.exps[i] + sh.exps[j];
                    s += sh.coefs[i] * sh.coefs[j] * mono_overlap_1d(cl[c1].x + cl[c2].x, p) * mono_overlap_1d(cl[c1].y + cl[c2].y, p) *
                         mono_overlap_1d(cl[c1].z + cl[c2].z, p);
                }
            M[(size_t)c1 * sh.ncart + c2] = s;
        }
    for (int f = 0; f < sh.nfunc; ++f) {
        double *row = &sh.T[(size_t)f * sh.ncart];
        double s2 = 0.0;
        for (int c1 = 0; c1 < sh.ncart; ++c1) for (int c2 = 0; c2 < sh.ncart; ++c2) s2 += row[c1] * row[c2] * M[(size_t)c1 * sh.ncart + c2];
        const double sc = 1.0 / std::sqrt(s2);
        for (int c = 0; c < sh.ncart; ++c) row[c] *= sc;
    }
}

// Hermite expansion matrix of one primitive pair in basis functions: out[h * nab + (fa*nb+fb)], h < nherm(la+lb).
static void pair_hermite_matrix(const QcShell &A, const QcShell &B, int i, int j, double scale, double *out, double *p_out, double P[3]) {
    const int la = A.L, lb = B.L, nh = qc_nherm(la + lb), nab = A.nfunc * B.nfunc;
    const double a = A.exps[i], b = B.exps[j], p = a + b;
    for (int k = 0; k < 3; ++k) P[k] = (a * A.A[k] + b * B.A[k]) / p;
    *p_out = p;
    E1 Ex = hermite_e(la, lb, a, b, A.A[0] - B.A[0]), Ey = hermite_e(la, lb, a, b, A.A[1] - B.A[1]), Ez = hermite_e(la, lb, a, b, A.A[2] - B.A[2]);
    auto ca = cart_list(la), cb = cart_list(lb);
    const double cc = A.coefs[i] * B.coefs[j] * scale;
    std::vector<double> Ec((size_t)ca.size() * cb.size() * nh, 0.0);
    for (size_t x = 0; x < ca.size(); ++x)
        for (size_t y = 0; y < cb.size(); ++y) {
            double *row = &Ec[(x * cb.size() + y) * nh];
            for (int t = 0; t <= ca[x].x + cb[y].x; ++t)
                for (int u = 0; u <= ca[x].y + cb[y].y; ++u)
                    for (int v = 0; v <= ca[x].z + cb[y].z; ++v)
                        row[qc_hidx(t, u, v)] = cc * Ex.get(ca[x].x, cb[y].x, t) * Ey.get(ca[x].y, cb[y].y, u) * Ez.get(ca[x].z, cb[y].z, v);
        }
    std::fill(out, out + (size_t)nh * nab, 0.0);
    for (int fa = 0; fa < A.nfunc; ++fa)
        for (size_t x = 0; x < ca.size(); ++x) {
            const double ta = A.T[(size_t)fa * A.ncart + x];
            if (ta == 0.0) continue;
            for (int fb = 0; fb < B.nfunc; ++fb)
                for (size_t y = 0; y < cb.size(); ++y) {
                    const double tb = ta * B.T[(size_t)fb * B.ncart + y];
                    if (tb == 0.0) continue;
                    const double *row = &Ec[(x * cb.size() + y) * nh];
                    for (int h = 0; h < nh; ++h) out[(size_t)h * nab + fa * B.nfunc + fb] += tb * row[h];
                }
        }
}

void qc_build_model(qc_system *S) {
    static const bool sdbg = getenv("QC_SETUP_DEBUG") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt = tnow();
    auto lap = [&](const char *what) { if (sdbg) { const double t = tnow(); fprintf(stderr, "[model] %-28s %.3f ms\n", what, t - tt); tt = t; } };
    int off = 0;
    for (auto &sh : S->shells) {
        for (int k = 0; k < 3; ++k) sh.A[k] = S->xyz[3 * sh.atom + k];
        normalise_shell(sh);
        sh.off = off;
        off += sh.nfunc;
    }
    S->nbasis = off;
    S->nelec = std::accumulate(S->Z.begin(), S->Z.end(), 0);
    // shell pairs A >= B, with per-primitive-pair blocks [p, Px, Py, Pz, E(nherm x nab)].
    // The pair part of the ERI prefactor 2 pi^{5/2} / (p q sqrt(p+q)) is folded in as sqrt(2) pi^{5/4} / p.
    const double half_pref = std::sqrt(2.0) * std::pow(M_PI, 1.25);
    S->pairs.clear(); S->pairA.clear(); S->pairB.clear(); S->pairKfull.clear(); S->pairdata.clear(); S->pairdataT.clear(); S->pspack.clear();
    S->pp_ok = true;
    // Primitive pairs whose whole expansion block is below QC_PRIM_CUTOFF are not stored: with the Gaussian-product
    // factor exp(-mu R^2) and the coefficients folded into E, an integral is bounded by max|E_ab| max|E_cd| (times
    // O(10)), so a dropped primitive pair changes no integral by more than ~1e-16 - five orders below the 1e-10 parity
    // bar - while two-centre pairs of deeply contracted s/p shells lose a third to two thirds of their primitives.
    // (The reference evaluates them all; counts quoted as "enumerated" keep the full numbers.)
    int64_t npairs_all = 0;
    std::vector<double> blkbuf;
    for (int a = 0; a < S->nshells; ++a)
        for (int b = 0; b <= a; ++b) {
            const QcShell &A = S->shells[a], &B = S->shells[b];
            ++npairs_all;
            QcPairDesc d;
            d.doff = (int)S->pairdata.size();
            d.K = 0;
            d.na = A.nfunc; d.nb = B.nfunc; d.offa = A.off; d.offb = B.off; d.L = A.L + B.L; d.shA_eq_shB = (a == b);
            d.psoff = -1; d.psperm = 0;
            const bool is_ps = (d.L == 1);
            const bool is_pp = (A.L == 1 && B.L == 1);
            if (is_ps || is_pp) d.psoff = (int)S->pspack.size();
            int ppermA = 0, ppermB = 0;                    // pp pairs: basis function of Cartesian axis x: (perm >> 2x) & 3, per shell
            if (is_pp) {
                auto axis_perm = [&](const QcShell &sh, int &perm) {
                    bool ok = sh.nfunc == 3 && sh.ncart == 3;
                    perm = 0;
                    int seen = 0;
                    for (int ax = 0; ax < 3 && ok; ++ax) {
                        int f = 0;
                        for (int g = 1; g < 3; ++g) if (std::fabs(sh.T[(size_t)g * 3 + ax]) > std::fabs(sh.T[(size_t)f * 3 + ax])) f = g;
                        for (int g = 0; g < 3; ++g) if (g != f && sh.T[(size_t)g * 3 + ax] != 0.0) ok = false;
                        perm |= f << (2 * ax); seen |= 1 << f;
                    }
                    return ok && seen == 7;
                };
                if (!axis_perm(A, ppermA) || !axis_perm(B, ppermB)) S->pp_ok = false;
                d.psperm = ppermA | (ppermB << 6);
            }
            const int nab = d.na * d.nb, stride = qc_pair_stride(d.L, nab), ne = qc_nherm(d.L) * nab;
            blkbuf.assign(stride, 0.0);
            for (int i = 0; i < A.nprim; ++i)
                for (int j = 0; j < B.nprim; ++j) {
                    double p, P[3];
                    const double pp = A.exps[i] + B.exps[j];
                    std::fill(blkbuf.begin(), blkbuf.end(), 0.0);
                    pair_hermite_matrix(A, B, i, j, half_pref / pp, blkbuf.data() + 4, &p, P);
                    double mx = 0.0;
                    for (int k = 0; k < ne; ++k) mx = std::max(mx, std::fabs(blkbuf[4 + k]));
                    if (mx < QC_PRIM_CUTOFF) continue;
                    blkbuf[0] = p; blkbuf[1] = P[0]; blkbuf[2] = P[1]; blkbuf[3] = P[2];
                    S->pairdata.insert(S->pairdata.end(), blkbuf.begin(), blkbuf.end());
                    const int nh = qc_nherm(d.L);
                    if (is_ps) {
                        // The 4 x 3 expansion block of a p.s product has 4 distinct numbers: row 0 (the s-type Hermite
                        // function) is dense, rows 1..3 are one value on a permutation (which p function is which axis).
                        const double *E = blkbuf.data() + 4;             // E[h * 3 + col]
                        int perm = 0;
                        double e1 = 0.0, e0[3] = {0, 0, 0};
                        bool ok = true;
                        for (int ax = 0; ax < 3; ++ax) {
                            int col = 0;
                            for (int c2 = 1; c2 < 3; ++c2) if (std::fabs(E[(1 + ax) * 3 + c2]) > std::fabs(E[(1 + ax) * 3 + col])) col = c2;
                            for (int c2 = 0; c2 < 3; ++c2) if (c2 != col && std::fabs(E[(1 + ax) * 3 + c2]) > 1e-14 * std::fabs(E[(1 + ax) * 3 + col])) ok = false;
                            if (ax == 0) e1 = E[3 + col];
                            else if (std::fabs(E[(1 + ax) * 3 + col] - e1) > 1e-13 * std::fabs(e1)) ok = false;
                            perm |= col << (2 * ax);
                            e0[ax] = E[col];
                        }
                        if (((perm & 3) == ((perm >> 2) & 3)) || ((perm & 3) == ((perm >> 4) & 3)) || (((perm >> 2) & 3) == ((perm >> 4) & 3))) ok = false;
                        if (d.K > 0 && perm != d.psperm) ok = false;
                        if (!ok) S->last_error = "p shell whose functions are not the Cartesian axes (in some order)";
                        d.psperm = perm;
                        const double rec[8] = {p, P[0], P[1], P[2], e0[0], e0[1], e0[2], e1};
                        S->pspack.insert(S->pspack.end(), rec, rec + 8);
                    }
                    if (is_pp && S->pp_ok) {
                        // Packed record of a p.p primitive pair (ket side of qc_fock_bm_kernel<2, 0>), 16 doubles:
                        //   [q, Q | A_x, A_y, A_z, kh | D_x, D_y, D_z, khh | hA_x, hA_y, hA_z, 0],  A = K (Q - C), D = Q - D', hq = 1/(2q), kh = K hq,
                        //   khh = K hq^2, hA = hq A:  the expansion of the function pair (axis c of the first shell, axis d of the second) is
                        //   E_000 = A_c D_d + [c = d] kh,  E_{e_c} = kh D_d,  E_{e_d} = hA_c,  E_{e_c + e_d} = khh - everything else is zero.
                        // K is read off the block (second-order coefficient of the x.x pair), and the whole block is checked against the form.
                        const double hq = 0.5 / p;
                        const double PC[3] = {P[0] - A.A[0], P[1] - A.A[1], P[2] - A.A[2]}, PD[3] = {P[0] - B.A[0], P[1] - B.A[1], P[2] - B.A[2]};
                        auto fn = [&](int perm, int ax) { return (perm >> (2 * ax)) & 3; };
                        auto Eb = [&](int t, int u, int v, int c, int dd) { return blkbuf[4 + (size_t)qc_hidx(t, u, v) * nab + fn(ppermA, c) * 3 + fn(ppermB, dd)]; };
                        const double Kc = Eb(2, 0, 0, 0, 0) / (hq * hq);
                        const double kh = Kc * hq, khh = Kc * hq * hq;
                        double worst = 0.0, scale_e = 0.0;
                        for (int c = 0; c < 3; ++c)
                            for (int dd = 0; dd < 3; ++dd)
                                for (int t = 0; t <= 2; ++t)
                                    for (int u = 0; t + u <= 2; ++u)
                                        for (int v = 0; t + u + v <= 2; ++v) {
                                            const int tv[3] = {t, u, v};
                                            int ec[3] = {0, 0, 0}, ed[3] = {0, 0, 0}, ecd[3] = {0, 0, 0};
                                            ec[c] = 1; ed[dd] = 1; ecd[c] += 1; ecd[dd] += 1;
                                            auto is = [&](const int *e) { return tv[0] == e[0] && tv[1] == e[1] && tv[2] == e[2]; };
                                            double want = 0.0;
                                            if (t + u + v == 0) want = Kc * PC[c] * PD[dd] + (c == dd ? kh : 0.0);
                                            else if (t + u + v == 1) want = (is(ec) ? kh * PD[dd] : 0.0) + (is(ed) ? kh * PC[c] : 0.0);
                                            else if (is(ecd)) want = khh;
                                            worst = std::max(worst, std::fabs(want - Eb(t, u, v, c, dd)));
                                            scale_e = std::max(scale_e, std::fabs(want));
                                        }
                        if (!(worst <= 1e-12 * std::max(scale_e, 1e-300))) S->pp_ok = false;
                        const double rec[16] = {p, P[0], P[1], P[2], Kc * PC[0], Kc * PC[1], Kc * PC[2], kh, PD[0], PD[1], PD[2], khh,
                                                hq * Kc * PC[0], hq * Kc * PC[1], hq * Kc * PC[2], 0.0};
                        S->pspack.insert(S->pspack.end(), rec, rec + 16);
                    }
                    const size_t t0 = S->pairdataT.size();
                    S->pairdataT.insert(S->pairdataT.end(), blkbuf.begin(), blkbuf.end());
                    for (int h = 0; h < nh; ++h)
                        for (int ab = 0; ab < nab; ++ab) S->pairdataT[t0 + 4 + (size_t)ab * nh + h] = blkbuf[4 + (size_t)h * nab + ab];
                    ++d.K;
                }
            if (d.K == 0) continue;                       // the whole shell pair is negligible
            S->pairs.push_back(d); S->pairA.push_back(a); S->pairB.push_back(b); S->pairKfull.push_back(A.nprim * B.nprim);
        }
    lap("shell pairs");
    // unique quartets (pair P >= pair Q), bucketed by launch class.
    //  * bra-major classes (bm): the narrower pair is an ss or ps pair and LAB + LCD <= QC_LREG.  It becomes the ket,
    //    one lane per quartet: the per-primitive-quartet contraction costs ncd * HAB * HCD FMAs, so the pair with the
    //    fewest functions belongs on the side that is contracted inside the primitive loop, and the bra side (shared
    //    by the whole wave) is contracted once per bra primitive pair.
    //  * all others: column kernels, the wider pair is the ket (one lane per ket function pair, R table shared in LDS).
    const int np = (int)S->pairs.size();
    S->nquartets = npairs_all * (npairs_all + 1) / 2;      // enumerated (unscreened) count, as the reference would visit
    const int NB = (QC_LPAIR + 1) * (QC_LPAIR + 1) * 7 * 2;
    std::vector<std::vector<QcTask>> bucket(NB);
    // (the A/B switches of the classification, read once per system - a test switches them - and NOT per quartet: two getenv calls in this
    // loop were 0.6 us each, a third of the time a handle took to create)
    const bool no_bm_pp = getenv("QC_NO_BM_PP") != nullptr, no_bm_ps4 = getenv("QC_NO_BM_PS4") != nullptr;
    for (int P = 0; P < np; ++P)
        for (int Q = 0; Q <= P; ++Q) {
            const QcPairDesc &dp = S->pairs[P], &dq = S->pairs[Q];
            const int np_ = dp.na * dp.nb, nq_ = dq.na * dq.nb;
            const bool p_is_wide = (np_ > nq_) || (np_ == nq_ && dp.L >= dq.L);
            const int wide = p_is_wide ? P : Q, narrow = p_is_wide ? Q : P;
            const QcPairDesc &dn = S->pairs[narrow], &dw = S->pairs[wide];
            // (round 3) p.p kets against p.p / d.s bras, total order 4: bra-major too - lane-per-quartet with the packed p.p records, three
            // bundles per ket group (one per Cartesian axis of the ket's first function, three columns each); the pair that would be the
            // ket of the column kernels (the wide one) stays the ket.  QC_NO_BM_PP: the column kernels keep them (A/B switch).
            const bool w_is_pp = S->shells[S->pairA[wide]].L == 1 && S->shells[S->pairB[wide]].L == 1;
            // (round 3: ps kets against d.d / f.p bras - total order 5: the table of 56 entries next to W[3][35] in a lane, one wave per SIMD
            // like the d.p / f.s bras of that launch - in the bra-major form too.  The column kernels ran (ps|dd) at 1.7 TFLOP/s; measured,
            // alternating runs on one box: H2O/cc-pVTZ iteration 0.3481 against 0.3529 ms (twelve runs each; the LCD = 4 bucket was 944 of the
            // 1444 waves of its tier<1, 1> launch), benzene/cc-pVDZ build 1.398 against 1.390 ms (four each: one launch fewer, no change).
            // QC_NO_BM_PS4 keeps them with the column kernels.)
            const bool ps4 = !no_bm_ps4 && dn.L == 1 && dw.L == 4;
            if ((dn.L <= 1 && dn.L + dw.L <= QC_LREG) || ps4) {
                bucket[(((dw.L * (QC_LPAIR + 1) + dn.L) * 7) + 0) * 2 + 1].push_back(QcTask{wide, narrow});
            } else if (!no_bm_pp && S->pp_ok && dn.L == 2 && dw.L == 2 && w_is_pp && dw.K <= 63 && QC_LREG >= 4) {
                bucket[(((2 * (QC_LPAIR + 1) + 2) * 7) + 0) * 2 + 1].push_back(QcTask{narrow, wide});
            } else {
                bucket[(((dn.L * (QC_LPAIR + 1) + dw.L) * 7) + qc_lgc_for(dn.L, dw.L, dw.na * dw.nb)) * 2].push_back(QcTask{narrow, wide});
            }
        }
    {   // The p.p-ket bra-major class pays where its list fills the chip (benzene/cc-pVDZ: 45 480 quartets, 0.148 ms against 0.186 ms in the
        // column kernels, build -6 %); a small molecule's few hundred bundles are one more latency-bound launch (H2O/cc-pVTZ: 1526 quartets,
        // no gain measured) - those stay with the column kernels.  QC_BM_PP_MIN overrides the threshold.
        const long pp_min = getenv("QC_BM_PP_MIN") ? atol(getenv("QC_BM_PP_MIN")) : 8192;      // (read per system: a test switches it)
        auto &ppb = bucket[(((2 * (QC_LPAIR + 1) + 2) * 7) + 0) * 2 + 1];
        if (!ppb.empty() && (long)ppb.size() < pp_min) {
            auto &colb = bucket[(((2 * (QC_LPAIR + 1) + 2) * 7) + qc_lgc_for(2, 2, 9)) * 2];
            colb.insert(colb.end(), ppb.begin(), ppb.end());
            ppb.clear();
        }
    }
    {   // d.d / f.p kets against d.p ... f.f bras: a short list (a small molecule) goes to the 64-lane instance of its class, which runs the
        // contractions as matrix-core tiles with one wave per 16-column tile; a long one keeps the 32-lane VALU form with two slots per wave
        // (benzene/cc-pVDZ: the tile form cost the full chip 2.6 % in round 2).  QC_MFMA4_MAX: longest list that switches (0: never).
        const long mx = getenv("QC_MFMA4_MAX") ? atol(getenv("QC_MFMA4_MAX")) : 1024;
        // (only where the wide-ket launches are the f-capable kernels - a basis with f functions; without them the launch is the variant
        // that carries the d.d / f.p-ket bodies alone, in their VALU form, at two waves per SIMD: qc_fock_tier_kernel<LAB, 2>)
        S->has_fkets = false;
        for (int b = 0; b < NB; ++b) if (!(b & 1) && (b / 14) % (QC_LPAIR + 1) >= 5 && !bucket[b].empty()) S->has_fkets = true;
        for (int lab = 3; S->has_fkets && lab <= QC_LPAIR; ++lab) {
            auto &b5 = bucket[(((lab * (QC_LPAIR + 1) + 4) * 7) + 5) * 2], &b6 = bucket[(((lab * (QC_LPAIR + 1) + 4) * 7) + 6) * 2];
            if (!b5.empty() && (long)b5.size() <= mx) { b6.insert(b6.end(), b5.begin(), b5.end()); b5.clear(); }
        }
    }
    lap("quartets into buckets");
    S->classes.clear();
    for (int b = 0; b < NB; ++b) {
        auto &v = bucket[b];
        if (v.empty()) continue;
        QcClass c;
        c.bm = b & 1;
        c.LGC = (b / 2) % 7; c.LCD = (b / 14) % (QC_LPAIR + 1); c.LAB = b / 14 / (QC_LPAIR + 1);
        c.tasks = std::move(v);
        S->classes.push_back(std::move(c));
    }
    {   // one launch for the wide-ket buckets of the low bra classes (qc_fock_tier1_low_kernel) where there are f-ket buckets among them -
        // without f functions those launches are the two-waves-per-SIMD variants and stay apart.  QC_NO_T1_MERGE: off (A/B switch).
        int nseg[3] = {0, 0, 0}; bool fket = false;
        for (const auto &c : S->classes) if (!c.bm && c.LCD >= 4) { ++nseg[c.LAB <= 2 ? 0 : (c.LAB <= 4 ? 1 : 2)]; fket = fket || c.LCD >= 5; }
        S->merge_t1 = fket && std::max(nseg[0], std::max(nseg[1], nseg[2])) <= 12 && getenv("QC_NO_T1_MERGE") == nullptr;
        bool has01 = false, has10 = false;
        for (const auto &c : S->classes) if (c.bm) { has01 = has01 || (c.LCD == 0 && c.LAB >= 3); has10 = has10 || (c.LCD == 1 && c.LAB <= 2); }
        // (a matter of launch count, so only for small builds: one launch less is 5 % of an H2O/cc-pVTZ build (32 k quartets, 0.17 ms); in a long
        // build the merged launch is the longest chain by itself - benzene/cc-pVDZ, 1.1 M quartets: builds 1.379 ms apart against 1.405
        // merged, three alternating runs each.  QC_BM_MERGE_MAX: the limit in quartets)
        size_t q_all = 0;
        for (const auto &c : S->classes) q_all += c.tasks.size();
        static const long merge_max = getenv("QC_BM_MERGE_MAX") ? atol(getenv("QC_BM_MERGE_MAX")) : 262144;
        S->merge_bm = has01 && has10 && (long)q_all <= merge_max && getenv("QC_NO_BM_MERGE") == nullptr;
    }
    lap("classes");
    // (the lists themselves wait for the Schwarz factors of the device set-up - or for whoever asks first, qc_ensure_lists: building them
    // here as well was 1.6 ms of a 13.7 ms cold SCF of H2O/cc-pVTZ, 60 ms for benzene/cc-pVDZ)
    qc_build_shards(S, true);
    lap("class sizes");
}

// Lane-group widths the kernels are instantiated for, per ket Hermite order (the switch in qc_fock_tier_kernel).  A group is
// made of whole 16-lane rows - the Hermite contractions broadcast inside rows (DPP) - so the 5..10 columns of a ds / fs ket
// leave lanes idle; as 8-lane groups with LDS-fed contractions those classes were 20 % slower.
int qc_lgc_for(int lab, int lcd, int ncd) {
    if (qc_mfma_always(lab, lcd)) return 6;         // matrix-core classes: one slot per wave, always the full wave
    static const int allowed[QC_LPAIR + 1][4] = {{4, -1, -1, -1}, {4, -1, -1, -1}, {4, -1, -1, -1}, {4, 5, -1, -1}, {5, 6, -1, -1}, {6, -1, -1, -1}, {6, -1, -1, -1}};
    for (int i = 0; i < 4 && allowed[lcd][i] >= 0; ++i)
        if ((1 << allowed[lcd][i]) >= ncd) return allowed[lcd][i];
    return 6;   // wider than a wave: the kernel makes several column passes
}

// Cut quartets into slots of at most `itmax` primitive quartets; longest first so the slots batched into one wave
// have (nearly) equal trip counts.  With `split_cols` (matrix-core classes that cannot fill the chip) a slot is further
// cut into the 16-column tiles of the ket block: the integrals of different ket columns are independent all the way into
// the digestion, so a single (ff|ff) quartet is then worked on by four waves (each repeats the R table, none waits).
void qc_make_slots(const qc_system *S, const std::vector<QcTask> &tasks, int itmax, bool split_cols, std::vector<QcSlot> &out) {
    out.clear();
    for (const auto &t : tasks) {
        const int npq = S->pairs[t.bra].K * S->pairs[t.ket].K, ncd = S->pairs[t.ket].na * S->pairs[t.ket].nb;
        const int step = itmax > 0 ? itmax : npq;
        const int nparts = (npq + step - 1) / step;
        const int ctile = split_cols ? 16 : ncd;
        for (int s = 0; s < nparts; ++s) {   // equal-length parts
            const int lo = (int)((int64_t)npq * s / nparts), hi = (int)((int64_t)npq * (s + 1) / nparts);
            for (int c0 = 0; c0 < ncd; c0 += ctile) out.push_back(QcSlot{t.bra, t.ket, lo, hi, c0, std::min(ncd, c0 + ctile)});
        }
    }
    std::stable_sort(out.begin(), out.end(), [](const QcSlot &x, const QcSlot &y) { return x.hi - x.lo > y.hi - y.lo; });
}

// Bra-major work units: the tasks of one bra pair, kets sorted by primitive count (so the lanes of a wave run nearly
// equal trip counts), cut into bundles of at most 64 kets; with itmax > 0 a bundle is further cut along the bra primitive
// pairs so that a lane evaluates about itmax primitive quartets.
// `unit` > 0 (round 3): a lane's work unit is a CHUNK of at most `unit` primitives of a ket pair instead of the whole pair - the entry of
// `ketlist` then carries the chunk (ket | first primitive << 18 | length << 25, seven bits each; length 0 = the whole pair, as the set-up
// passes use it).  Integrals are linear in the ket primitives and the digestion is linear in the integrals, so every chunk is digested on its own.
// With whole pairs a wave runs as long as its most contracted ket (64 primitive pairs for a pair of contracted s shells) while the
// lanes of single-primitive kets idle, and a molecule with few shell pairs cannot fill 64 lanes per bra at all (H2O/cc-pVTZ: 55 ss
// pairs); chunks of equal length fill the lanes and bound the trip count.
// stable counting sort, ascending key in [0, nkeys): the list orders below have small integer keys, and a comparator merge sort over the
// two to four million entries of a large system's bra-major classes was 200 of the 250 ms its work lists took (this container's host)
template <class T, class KeyFn>
static void qc_counting_sort(std::vector<T> &v, size_t nkeys, KeyFn key) {
    if (v.size() < 2) return;
    std::vector<uint32_t> cnt(nkeys + 1, 0);
    for (const auto &x : v) ++cnt[(size_t)key(x) + 1];
    for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
    std::vector<T> out(v.size());
    for (const auto &x : v) out[cnt[(size_t)key(x)]++] = x;
    v.swap(out);
}

bool qc_make_bundles(const qc_system *S, const std::vector<QcTask> &tasks, int itmax, std::vector<QcBundle> &bundles, std::vector<int> &ketlist, int unit) {
    bundles.clear(); ketlist.clear();
    if (unit > 0 && S->pairs.size() < (1u << QC_KET_BITS)) {
        struct U { int bra, ket, kl0, len; };
        std::vector<U> us;
        us.reserve(tasks.size() * 2);
        for (const auto &t : tasks) {
            const int K = S->pairs[t.ket].K;
            if (K > 127) { us.clear(); break; }                  // (does not fit the packed entry: whole pairs below)
            const int nch = (K + unit - 1) / unit;
            for (int c = 0; c < nch; ++c) {
                const int lo = (int)((int64_t)K * c / nch), hi = (int)((int64_t)K * (c + 1) / nch);
                us.push_back(U{t.bra, t.ket, lo, hi - lo});
            }
        }
        if (!us.empty()) {
            // by bra, inside a bra the longest chunks first, stable: the minor key first, then the major one (len <= 127)
            qc_counting_sort(us, 128, [](const U &x) { return 127 - x.len; });
            qc_counting_sort(us, S->pairs.size(), [](const U &x) { return x.bra; });
            ketlist.reserve(us.size());
            for (size_t i = 0; i < us.size();) {
                size_t j = i;
                while (j < us.size() && us[j].bra == us[i].bra && j - i < 64) ++j;
                const int Kab = S->pairs[us[i].bra].K, maxK = us[i].len;
                const int first = (int)ketlist.size();
                for (size_t k = i; k < j; ++k) ketlist.push_back((int)qc_pack_ket_entry(us[k].ket, us[k].kl0, us[k].len));
                // bra primitive pairs per bundle: about max(itmax, 32) primitive quartets per lane, so that the digestion of a partial block
                // (two to three primitive quartets' worth of instructions) stays a small share
                static const int pq_env = getenv("QC_BM_PQ") ? atoi(getenv("QC_BM_PQ")) : 32;           // (A/B switch)
                static const int pq_pp_env = getenv("QC_BM_PP_PQ") ? atoi(getenv("QC_BM_PP_PQ")) : 96;  // (the p.p-ket class - large lists only - amortises its heavier digestion over three times the rows: 0.148 -> 0.131 ms alone on benzene)
                const bool ket_pp = S->pairs[us[i].ket].L == 2;
                const int rows = std::max(1, std::min(Kab, std::max(itmax, ket_pp ? pq_pp_env : pq_env) / std::max(maxK, 1)));
                const int nparts = (Kab + rows - 1) / rows;
                for (int sp = 0; sp < nparts; ++sp)
                    bundles.push_back(QcBundle{us[i].bra, (int)((int64_t)Kab * sp / nparts), (int)((int64_t)Kab * (sp + 1) / nparts), first, (int)(j - i), maxK, 0, 0});
                i = j;
            }
            {   // long bundles first (stable)
                int64_t cmax = 0;
                for (const auto &x : bundles) cmax = std::max(cmax, (int64_t)(x.ij_hi - x.ij_lo) * x.maxK);
                if (cmax < (1 << 22)) qc_counting_sort(bundles, (size_t)cmax + 1, [cmax](const QcBundle &x) { return cmax - (int64_t)(x.ij_hi - x.ij_lo) * x.maxK; });
                else std::stable_sort(bundles.begin(), bundles.end(), [](const QcBundle &x, const QcBundle &y) {
                    return (int64_t)(x.ij_hi - x.ij_lo) * x.maxK > (int64_t)(y.ij_hi - y.ij_lo) * y.maxK;
                });
            }
            return true;
        }
    }
    std::vector<QcTask> t(tasks);
    {   // by bra, inside a bra the kets with the most primitive pairs first, stable
        int kmax = 0;
        for (const auto &x : t) kmax = std::max(kmax, S->pairs[x.ket].K);
        if (kmax < (1 << 20)) {
            qc_counting_sort(t, (size_t)kmax + 1, [&](const QcTask &x) { return kmax - S->pairs[x.ket].K; });
            qc_counting_sort(t, S->pairs.size(), [](const QcTask &x) { return x.bra; });
        } else
            std::stable_sort(t.begin(), t.end(), [&](const QcTask &x, const QcTask &y) {
                if (x.bra != y.bra) return x.bra < y.bra;
                return S->pairs[x.ket].K > S->pairs[y.ket].K;
            });
    }
    for (size_t i = 0; i < t.size();) {
        size_t j = i;
        while (j < t.size() && t[j].bra == t[i].bra && j - i < 64) ++j;
        const int Kab = S->pairs[t[i].bra].K, maxK = S->pairs[t[i].ket].K;
        const int first = (int)ketlist.size();
        for (size_t k = i; k < j; ++k) ketlist.push_back(t[k].ket);
        int nparts = 1;
        if (itmax > 0) nparts = std::min<int64_t>(Kab, std::max<int64_t>(1, ((int64_t)Kab * maxK + itmax - 1) / itmax));
        for (int s = 0; s < nparts; ++s)
            bundles.push_back(QcBundle{t[i].bra, (int)((int64_t)Kab * s / nparts), (int)((int64_t)Kab * (s + 1) / nparts), first, (int)(j - i), maxK, 0, 0});
        i = j;
    }
    // long bundles first: the tail of the launch is made of short ones
    {   // long bundles first (stable): cost = bra primitive pairs x longest ket
        int64_t cmax = 0;
        for (const auto &x : bundles) cmax = std::max(cmax, (int64_t)(x.ij_hi - x.ij_lo) * x.maxK);
        if (cmax < (1 << 22)) qc_counting_sort(bundles, (size_t)cmax + 1, [cmax](const QcBundle &x) { return cmax - (int64_t)(x.ij_hi - x.ij_lo) * x.maxK; });
        else std::stable_sort(bundles.begin(), bundles.end(), [](const QcBundle &x, const QcBundle &y) {
            return (int64_t)(x.ij_hi - x.ij_lo) * x.maxK > (int64_t)(y.ij_hi - y.ij_lo) * y.maxK;
        });
    }
    return false;          // entries are plain pair indices
}

// Static shard: inside every launch class the cost-sorted quartet list is dealt to the ranks in boustrophedon order
// (0..N-1, N-1..0, ...; start rank rotated per class), so each rank holds the same mix of classes and nearly the same
// modelled cost.  Data-only; no communication.
void qc_build_shards(qc_system *S, bool meta_only) {
    S->lists_stale = meta_only;
    static const bool sdbg = getenv("QC_SETUP_DEBUG") != nullptr;
    double tacc[6] = {0, 0, 0, 0, 0, 0};
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tlast = tnow();
    auto acc = [&](int k) { if (sdbg) { const double t = tnow(); tacc[k] += t - tlast; tlast = t; } };
    // Schwarz screening (once the factors exist - they come from a device pass): |(ab|cd)| <= Q_ab Q_cd, quartets below
    // schwarz_tau are not evaluated.  Density-independent, so the work lists stay static; the reference visits every quartet
    // (its own TODO, uhf.rs:49-50), throughput figures keep counting the enumerated ones.
    const bool screen = !S->pairQ.empty() && S->schwarz_tau > 0.0;
    S->nscreened = 0;
    std::vector<QcTask> kept;
    for (size_t ci = 0; ci < S->classes.size(); ++ci) {
        auto &c = S->classes[ci];
        if (meta_only) {
            // sizes over the WHOLE task list (an upper bound of any shard's): slot words / LDS bytes of the class, of its column-kernel form
            // (p.p-ket bra-major classes), rows of the bra-major exchange buffer
            c.shard.clear(); c.slots.clear(); c.bundles.clear(); c.ketlist.clear();
            c.prim_quartets = 0; c.bytes_alg = 0; c.flops_alg = 0; c.run = 0; c.rb_rows = 0;
            int words = 0, cw = 0, mx = 0;
            c.bm_rows = 0;
            if (c.bm && c.LCD == 2) c.col_lgc = qc_lgc_for(c.LAB, c.LCD, 9);
            for (const auto &t : c.tasks) {
                const QcPairDesc &b = S->pairs[t.bra], &k = S->pairs[t.ket];
                const int ncd = k.na * k.nb, nab = b.na * b.nb, kt = b.na * k.na + b.na * k.nb + b.nb * k.na + b.nb * k.nb;
                words = std::max(words, qc_region0(b.L + k.L, c.LGC) + nab * ncd + nab + ncd + 2 * kt + (c.LGC == 6 ? ncd + 2 * kt : 0));
                if (c.bm && c.LCD == 2) cw = std::max(cw, qc_region0(b.L + k.L, c.col_lgc) + nab * ncd + nab + ncd + 2 * kt);
                if (c.bm) { mx = std::max(mx, qc_bm_wave_words(c.LAB, nab, c.LCD == 2 ? 3 : ncd)); c.bm_rows = std::max(c.bm_rows, b.na + b.nb); }
            }
            c.slot_words = words;
            c.lds_bytes = words * 8 * (64 >> c.LGC) + (qc_hoisted(c.LAB + c.LCD) ? 0 : qc_nplan(c.LAB + c.LCD) * 8);
            if (c.bm && c.LCD == 2) { c.col_slot_words = cw; c.col_lds_bytes = cw * 8 * (64 >> c.col_lgc) + (qc_hoisted(c.LAB + c.LCD) ? 0 : qc_nplan(c.LAB + c.LCD) * 8); }
            if (c.bm) { c.slot_words = mx; c.lds_bytes = mx * 8; }
            continue;
        }
        // heaviest first: the primitive-quartet count is the dominant cost inside a class.  A stable counting sort on that count (a
        // product of two primitive-pair counts: a few thousand at most) - the comparator sort with its four indirections per comparison
        // was most of the 105 ms benzene/cc-pVDZ's lists took on the GPU box's host, twice per cold handle.
        {
            int64_t kmax = 0;
            for (const auto &t : c.tasks) kmax = std::max(kmax, (int64_t)S->pairs[t.bra].K * S->pairs[t.ket].K);
            if (kmax < (1 << 20)) {
                std::vector<uint32_t> cnt((size_t)kmax + 2, 0);
                for (const auto &t : c.tasks) ++cnt[(size_t)(kmax - (int64_t)S->pairs[t.bra].K * S->pairs[t.ket].K) + 1];
                for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
                std::vector<QcTask> sorted(c.tasks.size());
                for (const auto &t : c.tasks) sorted[cnt[(size_t)(kmax - (int64_t)S->pairs[t.bra].K * S->pairs[t.ket].K)]++] = t;
                c.tasks.swap(sorted);
            } else
                std::stable_sort(c.tasks.begin(), c.tasks.end(), [&](const QcTask &x, const QcTask &y) {
                    return (int64_t)S->pairs[x.bra].K * S->pairs[x.ket].K > (int64_t)S->pairs[y.bra].K * S->pairs[y.ket].K;
                });
        }
        acc(0);
        kept.clear();
        for (const auto &t : c.tasks)
            if (!screen || S->pairQ[t.bra] * S->pairQ[t.ket] >= S->schwarz_tau) kept.push_back(t);
        S->nscreened += (int64_t)(c.tasks.size() - kept.size());
        c.shard.clear();
        std::vector<QcBundle> pb; std::vector<int> kl;
        if (c.bm && S->nranks > 1) qc_make_bundles(S, kept, 0, pb, kl);
        if (pb.size() >= (size_t)8 * S->nranks) {
            // enough of them: deal whole 64-ket bundles, not single quartets, so a rank's waves stay full
            for (size_t i = 0; i < pb.size(); ++i)
                if (qc_shard_owner(i, S->nranks, ci) == S->rank)
                    for (int k = 0; k < pb[i].nket; ++k) c.shard.push_back(QcTask{pb[i].bra, kl[pb[i].first + k]});
        } else {
            for (size_t i = 0; i < kept.size(); ++i)
                if (qc_shard_owner(i, S->nranks, ci) == S->rank) c.shard.push_back(kept[i]);
        }
        acc(1);
        // slot length: long enough to amortise the per-slot digestion, short enough that the class still fills the chip
        int64_t tot_pq = 0;
        for (const auto &t : c.shard) tot_pq += (int64_t)S->pairs[t.bra].K * S->pairs[t.ket].K;
        const int64_t want_waves = 256 * 8, G = 64 >> c.LGC;
        // (at least 8 = one hoisted chunk: shorter slots only multiply the per-slot set-up and digestion; A/B on H2O/cc-pVTZ,
        // three runs each: 2 -> 0.346, 8 -> 0.327, 16 -> 0.342, 64 -> 0.465 ms per build)
        int itmax = (int)std::min<int64_t>(QC_SLOT_ITMAX, std::max<int64_t>(8, tot_pq / (want_waves * G)));
        c.slots.clear(); c.bundles.clear(); c.ketlist.clear();
        if (c.bm) {
            c.bm_rows = 0;
            for (const auto &t : c.shard) c.bm_rows = std::max(c.bm_rows, S->pairs[t.bra].na + S->pairs[t.bra].nb);
            c.ket_packed = qc_make_bundles(S, c.shard, itmax, c.bundles, c.ketlist);
            // ket-primitive chunks instead of whole ket pairs where they save instructions of the slowest lanes (model below)
            static const int unit_env = getenv("QC_BM_UNIT") ? atoi(getenv("QC_BM_UNIT")) : 8;       // (A/B switch: 0 = whole ket pairs)
            if (unit_env > 0) {
                // Model of a list: sum over its bundles of (bra primitive pairs x longest chunk) + 6 for the digestion of the lanes' partial
                // blocks.  Measured (four alternating runs each, H2O / benzene build in ms; whole pairs 0.269 / 1.642): chunks wherever this
                // model gains 0.229 / 1.640; only in the launches of the low bras (LAB <= 2) 0.240 / 1.601; with a model that also prices
                // step 3 - paid per bra primitive pair and LANE, 4116 FMAs for an (ff| bra, whatever the chunk length - 0.233 / 1.635.  So:
                // the low-bra launches switch where the model gains (>= 20 % for lists that fill the chip, any modelled gain and up to a
                // 20 % modelled loss for lists that cannot - there a launch is as long as its longest wave); the high-bra launches only when
                // the list is tiny (H2O's 192 bundles).
                auto cost = [](const std::vector<QcBundle> &bs) { int64_t t = 0; for (const auto &b : bs) t += (int64_t)(b.ij_hi - b.ij_lo) * b.maxK + 6; return t; };
                std::vector<QcBundle> ub; std::vector<int> uk;
                static const int unit_pp_env = getenv("QC_BM_PP_UNIT") ? atoi(getenv("QC_BM_PP_UNIT")) : 8;     // (the same for the p.p-ket class)
                const bool upacked = qc_make_bundles(S, c.shard, itmax, ub, uk, c.LCD == 2 ? unit_pp_env : unit_env);
                static const int gain_env = getenv("QC_BM_GAIN") ? atoi(getenv("QC_BM_GAIN")) : 0;       // (A/B switch: percent of the old cost)
                const bool high_bra = c.LAB >= 3;
                const int gain = gain_env > 0 ? gain_env : (c.bundles.size() < 4096 ? 120 : 80);
                const bool allowed = !high_bra || c.bundles.size() < 512;
                if (!ub.empty() && allowed && cost(ub) * 100 < cost(c.bundles) * gain) { c.bundles.swap(ub); c.ketlist.swap(uk); c.ket_packed = upacked; }
            }
            if (c.LCD == 2) {   // p.p kets: every bundle three times, once per Cartesian axis of the ket's first function (QcBundle::pad0)
                std::vector<QcBundle> b3;
                b3.reserve(3 * c.bundles.size());
                for (const auto &b : c.bundles)
                    for (int ax = 0; ax < 3; ++ax) { QcBundle x = b; x.pad0 = ax; b3.push_back(x); }
                c.bundles.swap(b3);
            }
        }
        else {
            qc_make_slots(S, c.shard, itmax, false, c.slots);
            // a matrix-core class (one slot per wave) with fewer slots than the chip has SIMDs: one wave per 16-column tile
            if ((qc_mfma_always(c.LAB, c.LCD) || (S->has_fkets && qc_use_mfma(c.LAB, c.LCD))) && c.LGC == 6 && c.slots.size() * 4 <= 1024) qc_make_slots(S, c.shard, itmax, true, c.slots);
            // Low-L classes (16- / 32-lane groups: several slots per wave): bra-run mode.  Slots are grouped by bra pair - heavy bras
            // first, inside a bra the longest slots first - every batch of G slots gets one bra (null slots pad), and a workgroup takes
            // `run` consecutive batches, so that the targets that belong to the bra stay in its LDS row buffer across them.
            c.run = 0; c.rb_rows = 0;
            // MEASURED AND NOT THE DEFAULT (round 3, benzene/cc-pVDZ, classes alone): (pp|pp)-type bucket 0.274 ms with independent slots,
            // 0.65 ms in bra runs of 7-8 batches WITH OR WITHOUT the row buffer - removing most of the class's global atomics changes
            // nothing (they are not what its waves wait for), while equal-length runs of one bra lose the load balance of the
            // length-sorted grid-stride list (a heavy bra's run is 5x the average).  QC_BRA_RUN=<batches> switches it on for A/B and
            // for counting atomic requests (tools/run_pmc.sh).
            static const int run_env = getenv("QC_BRA_RUN") ? atoi(getenv("QC_BRA_RUN")) : 0;
            if (c.LGC <= 5 && c.LCD <= 3 && S->nbasis <= 128 && run_env > 0 && !c.slots.empty()) {
                const int G = 64 >> c.LGC;
                std::vector<int> order;                      // bras by first appearance in the cost-sorted slot list
                std::vector<std::vector<QcSlot>> by;
                std::vector<int> where(S->pairs.size(), -1);
                for (const auto &sl : c.slots) {
                    if (where[sl.bra] < 0) { where[sl.bra] = (int)by.size(); by.emplace_back(); }
                    by[where[sl.bra]].push_back(sl);
                }
                std::vector<QcSlot> grouped;
                for (auto &v : by) {
                    std::stable_sort(v.begin(), v.end(), [](const QcSlot &x, const QcSlot &y) { return x.hi - x.lo > y.hi - y.lo; });
                    for (const auto &sl : v) grouped.push_back(sl);
                    while (grouped.size() % G) grouped.push_back(QcSlot{v[0].bra, -1, 0, 0, 0, 0});
                    c.rb_rows = std::max(c.rb_rows, S->pairs[v[0].bra].na + S->pairs[v[0].bra].nb);
                }
                c.slots.swap(grouped);
                const int64_t nbatch = (int64_t)c.slots.size() / G;
                // enough workgroups for two per SIMD-pair of the chip, at most 16 batches per run
                (void)nbatch;
                c.run = run_env;
            }
        }
        acc(c.bm ? 2 : 3);
        c.prim_quartets = 0; c.bytes_alg = 0; c.flops_alg = 0;
        int words = 0;
        for (const auto &t : c.shard) {
            const QcPairDesc &b = S->pairs[t.bra], &k = S->pairs[t.ket];
            const QcShell &A = S->shells[S->pairA[t.bra]], &B = S->shells[S->pairB[t.bra]];
            const QcShell &C = S->shells[S->pairA[t.ket]], &D = S->shells[S->pairB[t.ket]];
            const double Kab = S->pairKfull[t.bra], Kcd = S->pairKfull[t.ket], L = b.L + k.L;   // model on enumerated primitives
            const double hab = qc_nherm(b.L), hcd = qc_nherm(k.L);
            const double na = b.na, nb = b.nb, nc = k.na, nd = k.nb;
            const double ca = A.ncart, cb = B.ncart, cc = C.ncart, cd = D.ncart;
            c.prim_quartets += (int64_t)b.K * k.K;          // evaluated (after the primitive-pair cut-off)
            // SURVEY.md 8(d) work model, verbatim
            c.bytes_alg += 8.0 * (5.0 * (Kab + Kcd) + 3.0 * (na * nb + nc * nd + na * nc + na * nd + nb * nc + nb * nd));
            // (the two Hermite -> Cartesian terms depend on which pair is transformed first; the model takes the cheaper order, so that the
            // figure is a function of the quartet and not of the orientation a launch class happens to store it in)
            const double tr_ab_first = 2.0 * ca * cb * hab * hcd + 2.0 * ca * cb * cc * cd * hcd;
            const double tr_cd_first = 2.0 * cc * cd * hcd * hab + 2.0 * cc * cd * ca * cb * hab;
            c.flops_alg += Kab * Kcd * (40.0 * (L + 1) + 3.0 * qc_rwork((int)L) + 2.0 * hab * hcd) + std::min(tr_ab_first, tr_cd_first) +
                           12.0 * na * nb * nc * nd;
            // LDS doubles one lane group needs for this quartet (layout in qc_fock_kernel.h)
            const int ncd = k.na * k.nb, nab = b.na * b.nb;
            // (the bra block is read straight from memory)
            const int kt = b.na * k.na + b.na * k.nb + b.nb * k.na + b.nb * k.nb;
            // (I block, density tiles; 64-lane groups: + accumulators of the six target blocks)
            const int w = qc_region0(b.L + k.L, c.LGC) + nab * ncd + nab + ncd + 2 * kt + (c.LGC == 6 ? ncd + 2 * kt : 0);
            words = std::max(words, w);
        }
        c.slot_words = words;
        c.lds_bytes = words * 8 * (64 >> c.LGC) + (qc_hoisted(c.LAB + c.LCD) ? 0 : qc_nplan(c.LAB + c.LCD) * 8);   // + the R recurrence plan
        if (c.bm && c.LCD == 2) {   // (what the column kernels need for this class: the set-up passes run it through them)
            c.col_lgc = qc_lgc_for(c.LAB, c.LCD, 9);
            int cw = 0;
            for (const auto &t : c.tasks) {
                const QcPairDesc &b = S->pairs[t.bra], &k = S->pairs[t.ket];
                const int ncd = k.na * k.nb, nab = b.na * b.nb, kt = b.na * k.na + b.na * k.nb + b.nb * k.na + b.nb * k.nb;
                cw = std::max(cw, qc_region0(b.L + k.L, c.col_lgc) + nab * ncd + nab + ncd + 2 * kt);
            }
            c.col_slot_words = cw;
            c.col_lds_bytes = cw * 8 * (64 >> c.col_lgc) + (qc_hoisted(c.LAB + c.LCD) ? 0 : qc_nplan(c.LAB + c.LCD) * 8);
        }
        if (c.bm) {   // I[nab * ncd][65]: one column per lane
            int mx = 0;
            c.bm_rows = 0;
            for (const auto &t : c.shard) {
                mx = std::max(mx, qc_bm_wave_words(c.LAB, S->pairs[t.bra].na * S->pairs[t.bra].nb, c.LCD == 2 ? 3 : S->pairs[t.ket].na * S->pairs[t.ket].nb));
                c.bm_rows = std::max(c.bm_rows, S->pairs[t.bra].na + S->pairs[t.bra].nb);
            }
            c.slot_words = mx;
            c.lds_bytes = mx * 8;
        }
        acc(4);
    }
    if (sdbg && !meta_only) fprintf(stderr, "[lists] order %.2f  screen + deal %.2f  bundles %.2f  slots %.2f  work model + sizes %.2f ms\n", tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
}

// ---- one-electron matrices on the host (molint::overlap / kinetic / nuclear, rhf.rs:41-43); which: 0 S, 1 T, 2 V
void qc_host_one_electron(const qc_system *S, int which, double *out) {
    const int n = S->nbasis;
    std::fill(out, out + (size_t)n * n, 0.0);
    std::vector<double> R0;
    for (int a = 0; a < S->nshells; ++a)
        for (int b = 0; b <= a; ++b) {
            const QcShell &A = S->shells[a], &B = S->shells[b];
            auto ca = cart_list(A.L), cb = cart_list(B.L);
            std::vector<double> cart(ca.size() * cb.size(), 0.0);
            for (int i = 0; i < A.nprim; ++i)
                for (int j = 0; j < B.nprim; ++j) {
                    const double ea = A.exps[i], eb = B.exps[j], p = ea + eb, cc = A.coefs[i] * B.coefs[j];
                    double P[3];
                    for (int k = 0; k < 3; ++k) P[k] = (ea * A.A[k] + eb * B.A[k]) / p;
                    E1 E[3] = {hermite_e(A.L, B.L + 2, ea, eb, A.A[0] - B.A[0]), hermite_e(A.L, B.L + 2, ea, eb, A.A[1] - B.A[1]),
                               hermite_e(A.L, B.L + 2, ea, eb, A.A[2] - B.A[2])};
                    const double s3 = std::pow(M_PI / p, 1.5);
                    for (size_t x = 0; x < ca.size(); ++x)
                        for (size_t y = 0; y < cb.size(); ++y) {
                            const int ai[3] = {ca[x].x, ca[x].y, ca[x].z}, bi[3] = {cb[y].x, cb[y].y, cb[y].z};
                            double val = 0.0;
                            if (which == 0) {
                                val = s3 * E[0].get(ai[0], bi[0], 0) * E[1].get(ai[1], bi[1], 0) * E[2].get(ai[2], bi[2], 0);
                            } else if (which == 1) {
                                // -1/2 d^2/dx^2 acting on the ket primitive, one axis at a time
                                double s1[3], t1[3];
                                for (int k = 0; k < 3; ++k) {
                                    s1[k] = E[k].get(ai[k], bi[k], 0);
                                    t1[k] = 4.0 * eb * eb * E[k].get(ai[k], bi[k] + 2, 0) - 2.0 * eb * (2 * bi[k] + 1) * s1[k];
                                    if (bi[k] >= 2) t1[k] += bi[k] * (bi[k] - 1) * E[k].get(ai[k], bi[k] - 2, 0);
                                }
                                val = -0.5 * s3 * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
                            } else {
                                for (int c = 0; c < S->natoms; ++c) {
                                    const double PC[3] = {P[0] - S->xyz[3 * c], P[1] - S->xyz[3 * c + 1], P[2] - S->xyz[3 * c + 2]};
                                    hermite_r_host(A.L + B.L, p, PC, R0);
                                    double acc = 0.0;
                                    for (int t = 0; t <= ai[0] + bi[0]; ++t)
                                        for (int u = 0; u <= ai[1] + bi[1]; ++u)
                                            for (int v = 0; v <= ai[2] + bi[2]; ++v)
                                                acc += E[0].get(ai[0], bi[0], t) * E[1].get(ai[1], bi[1], u) * E[2].get(ai[2], bi[2], v) * R0[qc_hidx(t, u, v)];
                                    val -= S->Z[c] * 2.0 * M_PI / p * acc;
                                }
                            }
                            cart[x * cb.size() + y] += cc * val;
                        }
                }
            for (int fa = 0; fa < A.nfunc; ++fa)
                for (int fb = 0; fb < B.nfunc; ++fb) {
                    double v = 0.0;
                    for (size_t x = 0; x < ca.size(); ++x)
                        for (size_t y = 0; y < cb.size(); ++y) v += A.T[(size_t)fa * A.ncart + x] * B.T[(size_t)fb * B.ncart + y] * cart[x * cb.size() + y];
                    out[(size_t)(A.off + fa) * n + B.off + fb] = v;
                    out[(size_t)(B.off + fb) * n + A.off + fa] = v;
                }
        }
}
