/*
 * qc_oracle.c - CPU restatement of the qchem-rs Hartree-Fock path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X engine in qchem-rs_amd/.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" against the Rust binary.  The reference has no tests, no golden vectors and
 * cannot be built here (no Rust toolchain; its integral crate `molint` is a path dependency outside the tree,
 * /root/reference/Cargo.toml:12).  What pins this oracle instead (tests/test_oracle_known_answers.py):
 *   - Szabo & Ostlund H2/STO-3G integrals and energy on the reference's own data/mol/hydrogen.json + STO-3G.json,
 *   - T.D. Crawford's published H2O/STO-3G SCF energy (-74.942079928192 Eh) and E_nuc,
 *   - an independent numpy/scipy implementation (tools/gen_golden.py writes tests/golden/integrals_golden.json),
 *   - structural invariants (8-fold ERI symmetry, tr(DS)=N, rotation invariance).
 *
 * Two parts:
 *   (1) the SCF drivers, DIIS and helpers: a line-by-line restatement of
 *         /root/reference/core/src/hf/rhf.rs:32-181, uhf.rs:36-241, diis.rs:28-59, hf/utils.rs:7-36
 *       (conventional SCF: full ERI tensor once, dense n^4 contraction per iteration);
 *   (2) the integrals molint supplies (overlap/kinetic/nuclear/eri, call sites rhf.rs:41-45, uhf.rs:52-55):
 *       no source exists in the reference tree, so this is textbook McMurchie-Davidson (Helgaker, Jorgensen, Olsen,
 *       "Molecular Electronic-Structure Theory", ch. 9), the scheme the reference's profile implies.
 *
 * Conventions: bohr; row-major n x n matrices (all symmetric except C, whose columns are MOs: C[i*n+k]);
 * every contracted basis function is scaled to unit self-overlap; Cartesian components in lexicographic order
 * (xx,xy,xz,yy,yz,zz); spherical components m = -l..+l (real solid harmonics).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LMAX 4                 /* highest shell angular momentum supported by the oracle (g) */
#define LTOT (4 * LMAX)        /* highest Hermite order in an ERI */
#define NHERM(L) (((L) + 1) * ((L) + 2) * ((L) + 3) / 6)
#define NCART(L) (((L) + 1) * ((L) + 2) / 2)

typedef struct {
    int atom, L, pure, nprim, ncart, nfunc, off;
    double A[3];
    double *exp;   /* nprim */
    double *coef;  /* nprim, contraction coefficient times the norm of the (L,0,0) primitive */
    double *T;     /* nfunc x ncart: function = sum_c T[f][c] * x^lx y^ly z^lz * radial, unit self-overlap */
} Shell;

typedef struct {
    int natoms, nshells, nbasis;
    int *Z;
    double *xyz;
    Shell *sh;
} Basis;

static int g_hidx[LTOT + 1][LTOT + 1][LTOT + 1];
static int g_ht[NHERM(LTOT)], g_hu[NHERM(LTOT)], g_hv[NHERM(LTOT)];
static int g_tables_ready = 0;

static void init_tables(void) {
    if (g_tables_ready) return;
    int k = 0;
    for (int N = 0; N <= LTOT; N++)
        for (int t = N; t >= 0; t--)
            for (int u = N - t; u >= 0; u--) {
                int v = N - t - u;
                g_hidx[t][u][v] = k; g_ht[k] = t; g_hu[k] = u; g_hv[k] = v; k++;
            }
    g_tables_ready = 1;
}

static void cart_components(int L, int (*lmn)[3]) {
    int k = 0;
    for (int lx = L; lx >= 0; lx--)
        for (int ly = L - lx; ly >= 0; ly--) { lmn[k][0] = lx; lmn[k][1] = ly; lmn[k][2] = L - lx - ly; k++; }
}

static double dfact(int n) { double r = 1.0; for (; n > 1; n -= 2) r *= n; return r; }  /* n!! ; (-1)!! = 1 */
static double binom(int n, int k) {
    if (k < 0 || k > n) return 0.0;
    double r = 1.0; for (int i = 1; i <= k; i++) r = r * (n - k + i) / i; return r;
}
static double fact(int n) { double r = 1.0; for (int i = 2; i <= n; i++) r *= i; return r; }

/* Real solid harmonic S_lm as a polynomial in Cartesian monomials of degree l (Helgaker et al. eq. 6.4.47).
 * out[c] = coefficient of component c (lexicographic order).  Overall scale is irrelevant: rows are renormalised. */
static void solid_harmonic_row(int l, int m, double *out) {
    int nc = NCART(l), lmn[NCART(LMAX)][3];
    cart_components(l, lmn);
    for (int c = 0; c < nc; c++) out[c] = 0.0;
    int am = abs(m);
    int two_vm = (m < 0) ? 1 : 0;               /* v_m = 0 or 1/2, kept doubled */
    for (int t = 0; t <= (l - am) / 2; t++)
        for (int u = 0; u <= t; u++) {
            int vmax2 = 2 * ((am - two_vm) / 2) + two_vm; /* 2*(floor(|m|/2 - v_m) + v_m) */
            for (int v2 = two_vm; v2 <= vmax2; v2 += 2) {
                /* exponent of (-1): t + v - v_m with v - v_m integer = (v2 - two_vm)/2 */
                int sgn = ((t + (v2 - two_vm) / 2) % 2) ? -1 : 1;
                double c = sgn * pow(0.25, t) * binom(l, t) * binom(l - t, am + t) * binom(t, u) * binom(am, v2);
                int ex = 2 * t + am - (2 * u + v2), ey = 2 * u + v2, ez = l - 2 * t - am;
                if (ex < 0 || ey < 0 || ez < 0) continue;
                for (int k = 0; k < nc; k++)
                    if (lmn[k][0] == ex && lmn[k][1] == ey && lmn[k][2] == ez) out[k] += c;
            }
        }
}

/* ------------------------------------------------------------------ Hermite expansion coefficients (1-D) */
/* E[i][j][t] for 0<=i<=imax, 0<=j<=jmax, 0<=t<=i+j; Q = A-B along this axis. */
#define EDIM (LMAX + 3)
typedef double E1D[EDIM][EDIM][2 * EDIM];

static void hermite_e(int imax, int jmax, double a, double b, double Q, E1D E) {
    double p = a + b, mu = a * b / p;
    double XPA = -b / p * Q, XPB = a / p * Q, h = 0.5 / p;
    memset(E, 0, sizeof(E1D));
    E[0][0][0] = exp(-mu * Q * Q);
    for (int i = 0; i < imax; i++)
        for (int t = 0; t <= i + 1; t++) {
            double v = XPA * E[i][0][t] + (t + 1) * E[i][0][t + 1];
            if (t > 0) v += h * E[i][0][t - 1];
            E[i + 1][0][t] = v;
        }
    for (int i = 0; i <= imax; i++)
        for (int j = 0; j < jmax; j++)
            for (int t = 0; t <= i + j + 1; t++) {
                double v = XPB * E[i][j][t] + (t + 1) * E[i][j][t + 1];
                if (t > 0) v += h * E[i][j][t - 1];
                E[i][j + 1][t] = v;
            }
}

/* ------------------------------------------------------------------ Boys function F_n(x), n = 0..nmax */
static void boys(int nmax, double x, double *F) {
    if (x < 35.0) {
        /* convergent series at n = nmax, then downward recursion */
        double ex = exp(-x), term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 400; k++) {
            term *= 2.0 * x / (2 * nmax + 2 * k + 1);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        F[nmax] = ex * sum;
        for (int n = nmax - 1; n >= 0; n--) F[n] = (2.0 * x * F[n + 1] + ex) / (2 * n + 1);
    } else {
        double ex = exp(-x);
        F[0] = 0.5 * sqrt(M_PI / x) * erf(sqrt(x));
        for (int n = 0; n < nmax; n++) F[n + 1] = ((2 * n + 1) * F[n] - ex) / (2.0 * x);
    }
}

/* Hermite Coulomb integrals R^0_{tuv}(alpha, PQ) for t+u+v <= L, into R0[NHERM(L)] */
static void hermite_r(int L, double alpha, const double PQ[3], double *R0) {
    static __thread double W[LTOT + 1][NHERM(LTOT)];
    double F[LTOT + 1];
    double r2 = PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2];
    boys(L, alpha * r2, F);
    double f = 1.0;
    for (int n = 0; n <= L; n++) { W[n][0] = f * F[n]; f *= -2.0 * alpha; }
    for (int N = 1; N <= L; N++)
        for (int n = 0; n <= L - N; n++)
            for (int t = N; t >= 0; t--)
                for (int u = N - t; u >= 0; u--) {
                    int v = N - t - u;
                    double val;
                    if (t > 0) {
                        val = PQ[0] * W[n + 1][g_hidx[t - 1][u][v]];
                        if (t > 1) val += (t - 1) * W[n + 1][g_hidx[t - 2][u][v]];
                    } else if (u > 0) {
                        val = PQ[1] * W[n + 1][g_hidx[t][u - 1][v]];
                        if (u > 1) val += (u - 1) * W[n + 1][g_hidx[t][u - 2][v]];
                    } else {
                        val = PQ[2] * W[n + 1][g_hidx[t][u][v - 1]];
                        if (v > 1) val += (v - 1) * W[n + 1][g_hidx[t][u][v - 2]];
                    }
                    W[n][g_hidx[t][u][v]] = val;
                }
    memcpy(R0, W[0], sizeof(double) * NHERM(L));
}

/* ------------------------------------------------------------------ basis construction */
static double prim_overlap_1d_monomial(int i, int j, double a, double b) {
    /* int x^(i+j) exp(-(a+b)x^2) dx, same centre */
    int n = i + j;
    if (n % 2) return 0.0;
    double p = a + b;
    return dfact(n - 1) / pow(2.0 * p, n / 2) * sqrt(M_PI / p);
}

Basis *orc_basis_create(int natoms, const int *Z, const double *xyz, int nshells, const int *sh_atom,
                        const int *sh_L, const int *sh_pure, const int *sh_nprim, const double *exps,
                        const double *coefs) {
    init_tables();
    Basis *B = (Basis *)calloc(1, sizeof(Basis));
    B->natoms = natoms; B->nshells = nshells;
    B->Z = (int *)malloc(sizeof(int) * natoms);
    B->xyz = (double *)malloc(sizeof(double) * 3 * natoms);
    memcpy(B->Z, Z, sizeof(int) * natoms);
    memcpy(B->xyz, xyz, sizeof(double) * 3 * natoms);
    B->sh = (Shell *)calloc(nshells, sizeof(Shell));
    int poff = 0, boff = 0;
    for (int s = 0; s < nshells; s++) {
        Shell *S = &B->sh[s];
        int L = sh_L[s];
        if (L > LMAX) { fprintf(stderr, "oracle: L=%d unsupported\n", L); exit(2); }
        S->atom = sh_atom[s]; S->L = L; S->pure = sh_pure[s] && L >= 2; S->nprim = sh_nprim[s];
        S->ncart = NCART(L); S->nfunc = S->pure ? 2 * L + 1 : S->ncart; S->off = boff;
        for (int k = 0; k < 3; k++) S->A[k] = xyz[3 * S->atom + k];
        S->exp = (double *)malloc(sizeof(double) * S->nprim);
        S->coef = (double *)malloc(sizeof(double) * S->nprim);
        for (int i = 0; i < S->nprim; i++) {
            double a = exps[poff + i];
            S->exp[i] = a;
            /* norm of x^L exp(-a r^2) */
            double N = pow(2.0 * a / M_PI, 0.75) * pow(4.0 * a, 0.5 * L) / sqrt(dfact(2 * L - 1));
            S->coef[i] = coefs[poff + i] * N;
        }
        poff += S->nprim;
        S->T = (double *)calloc((size_t)S->nfunc * S->ncart, sizeof(double));
        if (S->pure) for (int m = -L; m <= L; m++) solid_harmonic_row(L, m, &S->T[(m + L) * S->ncart]);
        else for (int c = 0; c < S->ncart; c++) S->T[c * S->ncart + c] = 1.0;
        /* scale every function to unit self-overlap */
        int lmn[NCART(LMAX)][3];
        cart_components(L, lmn);
        for (int f = 0; f < S->nfunc; f++) {
            double s2 = 0.0;
            for (int c1 = 0; c1 < S->ncart; c1++)
                for (int c2 = 0; c2 < S->ncart; c2++) {
                    double tt = S->T[f * S->ncart + c1] * S->T[f * S->ncart + c2];
                    if (tt == 0.0) continue;
                    for (int i = 0; i < S->nprim; i++)
                        for (int j = 0; j < S->nprim; j++) {
                            double o = 1.0;
                            for (int k = 0; k < 3; k++)
                                o *= prim_overlap_1d_monomial(lmn[c1][k], lmn[c2][k], S->exp[i], S->exp[j]);
                            s2 += tt * S->coef[i] * S->coef[j] * o;
                        }
                }
            double sc = 1.0 / sqrt(s2);
            for (int c = 0; c < S->ncart; c++) S->T[f * S->ncart + c] *= sc;
        }
        boff += S->nfunc;
    }
    B->nbasis = boff;
    return B;
}

void orc_basis_destroy(Basis *B) {
    if (!B) return;
    for (int s = 0; s < B->nshells; s++) { free(B->sh[s].exp); free(B->sh[s].coef); free(B->sh[s].T); }
    free(B->sh); free(B->Z); free(B->xyz); free(B);
}
int orc_nbasis(const Basis *B) { return B->nbasis; }
int orc_nshells(const Basis *B) { return B->nshells; }
int orc_shell_offset(const Basis *B, int s) { return B->sh[s].off; }
int orc_shell_nfunc(const Basis *B, int s) { return B->sh[s].nfunc; }
int orc_shell_L(const Basis *B, int s) { return B->sh[s].L; }
int orc_shell_nprim(const Basis *B, int s) { return B->sh[s].nprim; }

/* transform a (ncartA x ncartB) Cartesian block to functions and scatter into an n x n matrix (both triangles) */
static void scatter_pair(const Basis *B, const Shell *SA, const Shell *SB, const double *cart, double *M) {
    int n = B->nbasis;
    for (int fa = 0; fa < SA->nfunc; fa++)
        for (int fb = 0; fb < SB->nfunc; fb++) {
            double v = 0.0;
            for (int ca = 0; ca < SA->ncart; ca++) {
                double ta = SA->T[fa * SA->ncart + ca];
                if (ta == 0.0) continue;
                for (int cb = 0; cb < SB->ncart; cb++) v += ta * SB->T[fb * SB->ncart + cb] * cart[ca * SB->ncart + cb];
            }
            M[(SA->off + fa) * n + SB->off + fb] = v;
            M[(SB->off + fb) * n + SA->off + fa] = v;
        }
}

/* ------------------------------------------------------------------ one-electron integrals (molint::overlap/kinetic/nuclear) */
/* which: 0 overlap, 1 kinetic, 2 nuclear attraction.  Replaces the call sites rhf.rs:41-43 / uhf.rs:52-54. */
static void one_electron(const Basis *B, int which, double *M) {
    int n = B->nbasis;
    memset(M, 0, sizeof(double) * n * n);
    static __thread E1D Ex, Ey, Ez;
    double cart[NCART(LMAX) * NCART(LMAX)];
    double R0[NHERM(2 * LMAX)];
    for (int sa = 0; sa < B->nshells; sa++)
        for (int sb = 0; sb <= sa; sb++) {
            const Shell *SA = &B->sh[sa], *SB = &B->sh[sb];
            int la = SA->L, lb = SB->L;
            int A[NCART(LMAX)][3], Bc[NCART(LMAX)][3];
            cart_components(la, A); cart_components(lb, Bc);
            memset(cart, 0, sizeof(cart));
            for (int i = 0; i < SA->nprim; i++)
                for (int j = 0; j < SB->nprim; j++) {
                    double a = SA->exp[i], b = SB->exp[j], p = a + b;
                    double cc = SA->coef[i] * SB->coef[j];
                    double P[3];
                    for (int k = 0; k < 3; k++) P[k] = (a * SA->A[k] + b * SB->A[k]) / p;
                    hermite_e(la, lb + 2, a, b, SA->A[0] - SB->A[0], Ex);
                    hermite_e(la, lb + 2, a, b, SA->A[1] - SB->A[1], Ey);
                    hermite_e(la, lb + 2, a, b, SA->A[2] - SB->A[2], Ez);
                    double s3 = pow(M_PI / p, 1.5);
                    for (int ca = 0; ca < SA->ncart; ca++)
                        for (int cb = 0; cb < SB->ncart; cb++) {
                            int ax = A[ca][0], ay = A[ca][1], az = A[ca][2];
                            int bx = Bc[cb][0], by = Bc[cb][1], bz = Bc[cb][2];
                            double val = 0.0;
                            if (which == 0) {
                                val = Ex[ax][bx][0] * Ey[ay][by][0] * Ez[az][bz][0] * s3;
                            } else if (which == 1) {
                                /* -1/2 <a| d^2/dx^2 |b> per axis, d^2/dx^2 acting on x^j e^{-b x^2} */
                                double Sx = Ex[ax][bx][0], Sy = Ey[ay][by][0], Sz = Ez[az][bz][0];
                                double Tx = -2.0 * b * (2 * bx + 1) * Ex[ax][bx][0] + 4.0 * b * b * Ex[ax][bx + 2][0];
                                if (bx >= 2) Tx += bx * (bx - 1) * Ex[ax][bx - 2][0];
                                double Ty = -2.0 * b * (2 * by + 1) * Ey[ay][by][0] + 4.0 * b * b * Ey[ay][by + 2][0];
                                if (by >= 2) Ty += by * (by - 1) * Ey[ay][by - 2][0];
                                double Tz = -2.0 * b * (2 * bz + 1) * Ez[az][bz][0] + 4.0 * b * b * Ez[az][bz + 2][0];
                                if (bz >= 2) Tz += bz * (bz - 1) * Ez[az][bz - 2][0];
                                val = -0.5 * (Tx * Sy * Sz + Sx * Ty * Sz + Sx * Sy * Tz) * s3;
                            } else {
                                for (int c = 0; c < B->natoms; c++) {
                                    double PC[3];
                                    for (int k = 0; k < 3; k++) PC[k] = P[k] - B->xyz[3 * c + k];
                                    hermite_r(la + lb, p, PC, R0);
                                    double acc = 0.0;
                                    for (int t = 0; t <= ax + bx; t++)
                                        for (int u = 0; u <= ay + by; u++)
                                            for (int v = 0; v <= az + bz; v++)
                                                acc += Ex[ax][bx][t] * Ey[ay][by][u] * Ez[az][bz][v] * R0[g_hidx[t][u][v]];
                                    val += -B->Z[c] * 2.0 * M_PI / p * acc;
                                }
                            }
                            cart[ca * SB->ncart + cb] += cc * val;
                        }
                }
            scatter_pair(B, SA, SB, cart, M);
        }
}
void orc_overlap(const Basis *B, double *M) { one_electron(B, 0, M); }
void orc_kinetic(const Basis *B, double *M) { one_electron(B, 1, M); }
void orc_nuclear(const Basis *B, double *M) { one_electron(B, 2, M); }

/* ------------------------------------------------------------------ two-electron integrals (molint::eri, rhf.rs:45) */
/* Hermite expansion of a primitive pair, dense: Eab[comp_ab][NHERM(la+lb)], comp_ab = ca*ncartB + cb */
static void pair_hermite(const Shell *SA, const Shell *SB, int i, int j, double *Eab, double *p_out, double *P) {
    static __thread E1D Ex, Ey, Ez;
    int la = SA->L, lb = SB->L, nh = NHERM(la + lb);
    int A[NCART(LMAX)][3], Bc[NCART(LMAX)][3];
    cart_components(la, A); cart_components(lb, Bc);
    double a = SA->exp[i], b = SB->exp[j], p = a + b, cc = SA->coef[i] * SB->coef[j];
    for (int k = 0; k < 3; k++) P[k] = (a * SA->A[k] + b * SB->A[k]) / p;
    *p_out = p;
    hermite_e(la, lb, a, b, SA->A[0] - SB->A[0], Ex);
    hermite_e(la, lb, a, b, SA->A[1] - SB->A[1], Ey);
    hermite_e(la, lb, a, b, SA->A[2] - SB->A[2], Ez);
    memset(Eab, 0, sizeof(double) * SA->ncart * SB->ncart * nh);
    for (int ca = 0; ca < SA->ncart; ca++)
        for (int cb = 0; cb < SB->ncart; cb++) {
            double *row = &Eab[(ca * SB->ncart + cb) * nh];
            for (int t = 0; t <= A[ca][0] + Bc[cb][0]; t++)
                for (int u = 0; u <= A[ca][1] + Bc[cb][1]; u++)
                    for (int v = 0; v <= A[ca][2] + Bc[cb][2]; v++)
                        row[g_hidx[t][u][v]] = cc * Ex[A[ca][0]][Bc[cb][0]][t] * Ey[A[ca][1]][Bc[cb][1]][u] *
                                               Ez[A[ca][2]][Bc[cb][2]][v];
        }
}

/* (ab|cd) for one shell quartet, in basis functions, out[fa][fb][fc][fd] (row-major) */
void orc_eri_shell_quartet(const Basis *B, int sa, int sb, int sc, int sd, double *out) {
    const Shell *SA = &B->sh[sa], *SB = &B->sh[sb], *SC = &B->sh[sc], *SD = &B->sh[sd];
    int lab = SA->L + SB->L, lcd = SC->L + SD->L, L = lab + lcd;
    int nhab = NHERM(lab), nhcd = NHERM(lcd);
    int nab = SA->ncart * SB->ncart, ncd = SC->ncart * SD->ncart;
    double *Eab = (double *)malloc(sizeof(double) * nab * nhab);
    double *Ecd = (double *)malloc(sizeof(double) * ncd * nhcd);
    double *W = (double *)malloc(sizeof(double) * nhab * ncd);
    double *cart = (double *)calloc((size_t)nab * ncd, sizeof(double));
    double R0[NHERM(LTOT)];
    for (int i = 0; i < SA->nprim; i++)
        for (int j = 0; j < SB->nprim; j++) {
            double p, P[3];
            pair_hermite(SA, SB, i, j, Eab, &p, P);
            for (int k = 0; k < SC->nprim; k++)
                for (int l = 0; l < SD->nprim; l++) {
                    double q, Q[3], PQ[3];
                    pair_hermite(SC, SD, k, l, Ecd, &q, Q);
                    for (int x = 0; x < 3; x++) PQ[x] = P[x] - Q[x];
                    double alpha = p * q / (p + q);
                    hermite_r(L, alpha, PQ, R0);
                    double pref = 2.0 * pow(M_PI, 2.5) / (p * q * sqrt(p + q));
                    /* W[h][cd] = sum_h' (-1)^{|h'|} Ecd[cd][h'] R[h + h'] */
                    for (int h = 0; h < nhab; h++)
                        for (int cd = 0; cd < ncd; cd++) {
                            double acc = 0.0;
                            const double *e = &Ecd[cd * nhcd];
                            for (int h2 = 0; h2 < nhcd; h2++) {
                                if (e[h2] == 0.0) continue;
                                int sg = ((g_ht[h2] + g_hu[h2] + g_hv[h2]) & 1) ? -1 : 1;
                                acc += sg * e[h2] * R0[g_hidx[g_ht[h] + g_ht[h2]][g_hu[h] + g_hu[h2]][g_hv[h] + g_hv[h2]]];
                            }
                            W[h * ncd + cd] = acc * pref;
                        }
                    for (int ab = 0; ab < nab; ab++) {
                        const double *e = &Eab[ab * nhab];
                        for (int h = 0; h < nhab; h++) {
                            if (e[h] == 0.0) continue;
                            double eh = e[h];
                            for (int cd = 0; cd < ncd; cd++) cart[ab * ncd + cd] += eh * W[h * ncd + cd];
                        }
                    }
                }
        }
    /* Cartesian -> basis functions on all four indices */
    int na = SA->nfunc, nb = SB->nfunc, nc = SC->nfunc, nd = SD->nfunc;
    int ca = SA->ncart, cb = SB->ncart, cc = SC->ncart, cd = SD->ncart;
    double *t1 = (double *)calloc((size_t)na * cb * cc * cd, sizeof(double));
    for (int f = 0; f < na; f++) for (int c = 0; c < ca; c++) { double t = SA->T[f * ca + c]; if (t == 0.0) continue;
        for (int r = 0; r < cb * cc * cd; r++) t1[f * cb * cc * cd + r] += t * cart[c * cb * cc * cd + r]; }
    double *t2 = (double *)calloc((size_t)na * nb * cc * cd, sizeof(double));
    for (int fa = 0; fa < na; fa++) for (int f = 0; f < nb; f++) for (int c = 0; c < cb; c++) { double t = SB->T[f * cb + c]; if (t == 0.0) continue;
        for (int r = 0; r < cc * cd; r++) t2[(fa * nb + f) * cc * cd + r] += t * t1[(fa * cb + c) * cc * cd + r]; }
    double *t3 = (double *)calloc((size_t)na * nb * nc * cd, sizeof(double));
    for (int ab = 0; ab < na * nb; ab++) for (int f = 0; f < nc; f++) for (int c = 0; c < cc; c++) { double t = SC->T[f * cc + c]; if (t == 0.0) continue;
        for (int r = 0; r < cd; r++) t3[(ab * nc + f) * cd + r] += t * t2[(ab * cc + c) * cd + r]; }
    memset(out, 0, sizeof(double) * na * nb * nc * nd);
    for (int abc = 0; abc < na * nb * nc; abc++) for (int f = 0; f < nd; f++) { double acc = 0.0;
        for (int c = 0; c < cd; c++) acc += SD->T[f * cd + c] * t3[abc * cd + c];
        out[abc * nd + f] = acc; }
    free(t1); free(t2); free(t3); free(Eab); free(Ecd); free(W); free(cart);
}

/* number of unique shell quartets (A>=B, C>=D, AB>=CD) */
long orc_n_unique_quartets(const Basis *B) {
    long np = (long)B->nshells * (B->nshells + 1) / 2;
    return np * (np + 1) / 2;
}

/* Full ERI tensor, row-major (i,j,k,l), chemists' notation - what `molint::eri` hands to rhf.rs:45.
 * Computes every `stride`-th unique shell quartet starting at `first` (stride=1, first=0: all) and fills the
 * 8 symmetry-equivalent positions.  Returns the number of shell quartets computed. */
long orc_eri_full_strided(const Basis *B, double *out, long first, long stride) {
    int n = B->nbasis;
    size_t n2 = (size_t)n * n, n3 = n2 * n;
    long count = 0, idx = 0;
    double *buf = (double *)malloc(sizeof(double) * 15 * 15 * 15 * 15);
    for (int sa = 0; sa < B->nshells; sa++)
        for (int sb = 0; sb <= sa; sb++)
            for (int sc = 0; sc <= sa; sc++)
                for (int sd = 0; sd <= (sc == sa ? sb : sc); sd++, idx++) {
                    if (idx < first || (idx - first) % stride) continue;
                    count++;
                    orc_eri_shell_quartet(B, sa, sb, sc, sd, buf);
                    const Shell *SA = &B->sh[sa], *SB = &B->sh[sb], *SC = &B->sh[sc], *SD = &B->sh[sd];
                    for (int fa = 0; fa < SA->nfunc; fa++) for (int fb = 0; fb < SB->nfunc; fb++)
                    for (int fc = 0; fc < SC->nfunc; fc++) for (int fd = 0; fd < SD->nfunc; fd++) {
                        double v = buf[((fa * SB->nfunc + fb) * SC->nfunc + fc) * SD->nfunc + fd];
                        size_t i = SA->off + fa, j = SB->off + fb, k = SC->off + fc, l = SD->off + fd;
                        out[i * n3 + j * n2 + k * n + l] = v; out[j * n3 + i * n2 + k * n + l] = v;
                        out[i * n3 + j * n2 + l * n + k] = v; out[j * n3 + i * n2 + l * n + k] = v;
                        out[k * n3 + l * n2 + i * n + j] = v; out[l * n3 + k * n2 + i * n + j] = v;
                        out[k * n3 + l * n2 + j * n + i] = v; out[l * n3 + k * n2 + j * n + i] = v;
                    }
                }
    free(buf);
    return count;
}
void orc_eri_full(const Basis *B, double *out) { orc_eri_full_strided(B, out, 0, 1); }

/* The same on `nthreads` host cores (OpenMP over the outer shell index, dynamic schedule): the "all host cores" CPU
 * baseline of BASELINE.md / SURVEY 8d.  The reference itself is single-threaded; unique quartets write disjoint tensor
 * elements, so the threads share nothing but the output array.  out == NULL: integrals are evaluated and dropped (timing
 * of samples whose n^4 tensor would not be worth allocating). */
long orc_eri_full_strided_mt(const Basis *B, double *out, long first, long stride, int nthreads) {
    int n = B->nbasis;
    size_t n2 = (size_t)n * n, n3 = n2 * n;
    long count = 0;
    int ns = B->nshells, npair = ns * (ns + 1) / 2;
    long *base = (long *)malloc(sizeof(long) * (npair + 1));    /* index of the first quartet of bra pair (sa, sb) */
    int *psa = (int *)malloc(sizeof(int) * npair), *psb = (int *)malloc(sizeof(int) * npair);
    base[0] = 0;
    for (int sa = 0, p = 0; sa < ns; sa++)
        for (int sb = 0; sb <= sa; sb++, p++) {
            long c = 0;
            for (int sc = 0; sc <= sa; sc++) c += (sc == sa ? sb : sc) + 1;
            psa[p] = sa; psb[p] = sb; base[p + 1] = base[p] + c;
        }
    { double tmp[1]; orc_eri_shell_quartet(B, 0, 0, 0, 0, tmp); }  /* index tables are built before the threads start */
#pragma omp parallel num_threads(nthreads) reduction(+ : count)
    {
        double *buf = (double *)malloc(sizeof(double) * 15 * 15 * 15 * 15);
#pragma omp for schedule(dynamic, 1)
        for (int px = 0; px < npair; px++) {
            int p = npair - 1 - px;                             /* the bra pairs with the most kets first */
            int sa = psa[p], sb = psb[p];
            long idx = base[p];
            for (int sc = 0; sc <= sa; sc++)
                for (int sd = 0; sd <= (sc == sa ? sb : sc); sd++, idx++) {
                    if (idx < first || (idx - first) % stride) continue;
                    count++;
                    orc_eri_shell_quartet(B, sa, sb, sc, sd, buf);
                    if (!out) continue;
                    const Shell *SA = &B->sh[sa], *SB = &B->sh[sb], *SC = &B->sh[sc], *SD = &B->sh[sd];
                    for (int fa = 0; fa < SA->nfunc; fa++) for (int fb = 0; fb < SB->nfunc; fb++)
                    for (int fc = 0; fc < SC->nfunc; fc++) for (int fd = 0; fd < SD->nfunc; fd++) {
                        double v = buf[((fa * SB->nfunc + fb) * SC->nfunc + fc) * SD->nfunc + fd];
                        size_t i = SA->off + fa, j = SB->off + fb, k = SC->off + fc, l = SD->off + fd;
                        out[i * n3 + j * n2 + k * n + l] = v; out[j * n3 + i * n2 + k * n + l] = v;
                        out[i * n3 + j * n2 + l * n + k] = v; out[j * n3 + i * n2 + l * n + k] = v;
                        out[k * n3 + l * n2 + i * n + j] = v; out[l * n3 + k * n2 + i * n + j] = v;
                        out[k * n3 + l * n2 + j * n + i] = v; out[l * n3 + k * n2 + j * n + i] = v;
                    }
                }
        }
        free(buf);
    }
    free(base); free(psa); free(psb);
    return count;
}

/* ------------------------------------------------------------------ dense helpers */
static void matmul(int n, const double *A, const double *B, double *C) {     /* C = A B */
    for (int i = 0; i < n; i++) {
        double *c = &C[i * n];
        for (int j = 0; j < n; j++) c[j] = 0.0;
        for (int k = 0; k < n; k++) { double a = A[i * n + k]; const double *b = &B[k * n]; for (int j = 0; j < n; j++) c[j] += a * b[j]; }
    }
}
static void matmul_tn(int n, const double *A, const double *B, double *C) {  /* C = A^T B */
    for (int i = 0; i < n; i++) {
        double *c = &C[i * n];
        for (int j = 0; j < n; j++) c[j] = 0.0;
        for (int k = 0; k < n; k++) { double a = A[k * n + i]; const double *b = &B[k * n]; for (int j = 0; j < n; j++) c[j] += a * b[j]; }
    }
}

/* Symmetric eigendecomposition by cyclic Jacobi (stands in for nalgebra::SymmetricEigen, utils.rs:15-18).
 * V columns = eigenvectors (V[i*n+k]); w unordered on purpose - utils::eigs returns them unordered too. */
void orc_eigs(int n, const double *Ain, double *V, double *w) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < n; i++) { dia += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-32 * (dia + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    free(A);
}

/* utils::sorted_eigs (utils.rs:20-36): ascending by value, columns permuted alongside */
void orc_sorted_eigs(int n, const double *A, double *V, double *w) {
    double *V0 = (double *)malloc(sizeof(double) * n * n), *w0 = (double *)malloc(sizeof(double) * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    orc_eigs(n, A, V0, w0);
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int i = 1; i < n; i++) { int k = perm[i], j = i - 1; while (j >= 0 && w0[perm[j]] > w0[k]) { perm[j + 1] = perm[j]; j--; } perm[j + 1] = k; }
    for (int k = 0; k < n; k++) { w[k] = w0[perm[k]]; for (int i = 0; i < n; i++) V[i * n + k] = V0[i * n + perm[k]]; }
    free(V0); free(w0); free(perm);
}

/* compute_nuclear_repulsion (rhf.rs:110-122): integer product cast to f64 */
double orc_nuclear_repulsion(const Basis *B) {
    double e = 0.0;
    for (int a = 0; a < B->natoms; a++)
        for (int b = a + 1; b < B->natoms; b++) {
            double d = 0.0;
            for (int k = 0; k < 3; k++) { double x = B->xyz[3 * b + k] - B->xyz[3 * a + k]; d += x * x; }
            e += (double)(B->Z[a] * B->Z[b]) / sqrt(d);
        }
    return e;
}

/* compute_transformation_matrix (rhf.rs:124-131): X = U diag((U^T S U)_ii^-1/2) U^T */
void orc_transformation_matrix(int n, const double *S, double *X) {
    double *U = (double *)malloc(sizeof(double) * n * n), *w = (double *)malloc(sizeof(double) * n);
    double *SU = (double *)malloc(sizeof(double) * n * n), *Lm = (double *)malloc(sizeof(double) * n * n);
    orc_eigs(n, S, U, w);
    matmul(n, S, U, SU);
    matmul_tn(n, U, SU, Lm);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double acc = 0.0;
            for (int k = 0; k < n; k++) acc += U[i * n + k] * (1.0 / sqrt(Lm[k * n + k])) * U[j * n + k];
            X[i * n + j] = acc;
        }
    free(U); free(w); free(SU); free(Lm);
}

/* compute_updated_density (rhf.rs:169-181 with factor=2, uhf.rs:229-241 with factor=1) */
static void updated_density(int n, const double *C, int nocc, double factor, double *D) {
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            double sum = 0.0;
            for (int k = 0; k < nocc; k++) sum += C[i * n + k] * C[j * n + k];
            D[i * n + j] = D[j * n + i] = factor * sum;
        }
}

/* compute_hückel_density (rhf.rs:133-150 / uhf.rs:191-208) */
static void huckel_density(int n, const double *H, const double *S, const double *X, int nocc, double factor, double *D) {
    double *He = (double *)malloc(sizeof(double) * n * n), *t = (double *)malloc(sizeof(double) * n * n);
    double *M = (double *)calloc((size_t)n * n, sizeof(double)), *Cp = (double *)malloc(sizeof(double) * n * n);
    double *C = (double *)malloc(sizeof(double) * n * n), *w = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) He[i * n + j] = He[j * n + i] = 1.75 * S[i * n + j] * (H[i * n + i] + H[j * n + j]) / 2.0;
    matmul(n, He, X, t);
    matmul_tn(n, X, t, M);
    orc_sorted_eigs(n, M, Cp, w);
    matmul(n, X, Cp, C);
    updated_density(n, C, nocc, factor, D);
    free(He); free(t); free(M); free(Cp); free(C); free(w);
}

/* ------------------------------------------------------------------ DIIS (diis.rs:28-59) */
typedef struct { int minlen, maxlen, len, n; double **err, **fock; } Diis;
static Diis *diis_new(int minlen, int maxlen, int n) {
    Diis *d = (Diis *)calloc(1, sizeof(Diis));
    d->minlen = minlen; d->maxlen = maxlen; d->n = n;
    d->err = (double **)calloc(maxlen + 1, sizeof(double *)); d->fock = (double **)calloc(maxlen + 1, sizeof(double *));
    return d;
}
static void diis_free(Diis *d) { for (int i = 0; i < d->len; i++) { free(d->err[i]); free(d->fock[i]); } free(d->err); free(d->fock); free(d); }

/* Householder QR solve of a dense m x m system (nalgebra `qr().solve()`, diis.rs:50-51); returns 0 if R has a zero pivot */
int orc_qr_solve(int m, const double *Ain, const double *b, double *x) {
    double *A = (double *)malloc(sizeof(double) * m * m), *y = (double *)malloc(sizeof(double) * m);
    memcpy(A, Ain, sizeof(double) * m * m); memcpy(y, b, sizeof(double) * m);
    int ok = 1;
    for (int k = 0; k < m; k++) {
        double norm = 0.0;
        for (int i = k; i < m; i++) norm += A[i * m + k] * A[i * m + k];
        norm = sqrt(norm);
        if (norm == 0.0) { ok = 0; break; }
        double alpha = A[k * m + k] > 0 ? -norm : norm;
        double *v = (double *)calloc(m, sizeof(double));
        double vnorm2 = 0.0;
        for (int i = k; i < m; i++) { v[i] = A[i * m + k]; if (i == k) v[i] -= alpha; vnorm2 += v[i] * v[i]; }
        if (vnorm2 > 0.0) {
            for (int j = k; j < m; j++) {
                double dot = 0.0; for (int i = k; i < m; i++) dot += v[i] * A[i * m + j];
                double f = 2.0 * dot / vnorm2; for (int i = k; i < m; i++) A[i * m + j] -= f * v[i];
            }
            double dot = 0.0; for (int i = k; i < m; i++) dot += v[i] * y[i];
            double f = 2.0 * dot / vnorm2; for (int i = k; i < m; i++) y[i] -= f * v[i];
        }
        free(v);
    }
    if (ok) for (int i = m - 1; i >= 0; i--) {
        double s = y[i];
        for (int j = i + 1; j < m; j++) s -= A[i * m + j] * x[j];
        if (A[i * m + i] == 0.0) { ok = 0; break; }
        x[i] = s / A[i * m + i];
    }
    free(A); free(y);
    return ok;
}

/* Diis::fock: push_front, truncate, B matrix with +1 border, QR solve, F = sum c_i F_i.  0 = singular ("DIIS failed") */
static int diis_fock(Diis *d, const double *err, const double *fock, double *out) {
    int nn = d->n * d->n;
    double *e = (double *)malloc(sizeof(double) * nn), *f = (double *)malloc(sizeof(double) * nn);
    memcpy(e, err, sizeof(double) * nn); memcpy(f, fock, sizeof(double) * nn);
    for (int i = d->len; i > 0; i--) { d->err[i] = d->err[i - 1]; d->fock[i] = d->fock[i - 1]; }
    d->err[0] = e; d->fock[0] = f; d->len++;
    if (d->len > d->maxlen) { d->len--; free(d->err[d->len]); free(d->fock[d->len]); }
    int m = d->len;
    if (m < d->minlen) { memcpy(out, d->fock[0], sizeof(double) * nn); return 1; }
    double *Bm = (double *)calloc((size_t)(m + 1) * (m + 1), sizeof(double));
    double *rhs = (double *)calloc(m + 1, sizeof(double)), *c = (double *)calloc(m + 1, sizeof(double));
    for (int i = 0; i <= m; i++)
        for (int j = i; j <= m; j++) {
            double v;
            if (i == m && j == m) v = 0.0;
            else if (i == m || j == m) v = 1.0;
            else { v = 0.0; for (int k = 0; k < nn; k++) v += d->err[i][k] * d->err[j][k]; }
            Bm[i * (m + 1) + j] = Bm[j * (m + 1) + i] = v;
        }
    rhs[m] = 1.0;
    int ok = orc_qr_solve(m + 1, Bm, rhs, c);
    if (ok) {
        for (int k = 0; k < nn; k++) out[k] = 0.0;
        for (int i = 0; i < m; i++) for (int k = 0; k < nn; k++) out[k] += c[i] * d->fock[i][k];
    }
    free(Bm); free(rhs); free(c);
    return ok;
}

/* ------------------------------------------------------------------ Fock contractions */
/* electron_terms (rhf.rs:58-62): T4[i,j,k,l] = I[i,j,k,l] - 0.5 I[i,k,j,l] */
void orc_antisym_tensor(int n, const double *I, double *T4) {
    size_t n2 = (size_t)n * n, n3 = n2 * n;
    for (size_t i = 0; i < (size_t)n; i++) for (size_t j = 0; j < (size_t)n; j++)
        for (size_t k = 0; k < (size_t)n; k++) for (size_t l = 0; l < (size_t)n; l++)
            T4[i * n3 + j * n2 + k * n + l] = I[i * n3 + j * n2 + k * n + l] - 0.5 * I[i * n3 + k * n2 + j * n + l];
}
/* compute_electronic_hamiltonian RHF (rhf.rs:152-167): G[i,j] = sum_kl D[k,l] T4[i,j,k,l], i<=j, mirrored */
void orc_g_rhf(int n, const double *D, const double *T4, double *G) {
    size_t n2 = (size_t)n * n, n3 = n2 * n;
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            double sum = 0.0;
            const double *t = &T4[i * n3 + j * n2];
            for (int k = 0; k < n; k++) for (int l = 0; l < n; l++) sum += D[k * n + l] * t[k * n + l];
            G[i * n + j] = G[j * n + i] = sum;
        }
}
/* compute_electronic_hamiltonian UHF (uhf.rs:210-227), same term order */
void orc_g_uhf(int n, const double *D1, const double *D2, const double *I, double *G) {
    size_t n2 = (size_t)n * n, n3 = n2 * n;
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            double sum = 0.0;
            for (int k = 0; k < n; k++) for (int l = 0; l < n; l++)
                sum += D1[k * n + l] * I[i * n3 + j * n2 + k * n + l] + D2[k * n + l] * I[i * n3 + j * n2 + k * n + l]
                     - D1[k * n + l] * I[i * n3 + k * n2 + j * n + l];
            G[i * n + j] = G[j * n + i] = sum;
        }
}

/* Direct digestion of a LIST of unique shell quartets (A>=B, C>=D, AB>=CD) into G = J - 1/2 K: the checker for the
 * product's work-sharding (tests/test_sharding_gloo.py).  Each quartet's block is expanded to its <=8 distinct index
 * images and contracted exactly like the dense loop of rhf.rs:152-167, so summing over a partition of all unique
 * quartets reproduces orc_g_rhf on the full tensor. */
void orc_g_rhf_quartets(const Basis *B, long nq, const int *abcd, const double *D, double *G) {
    int n = B->nbasis;
    double *buf = (double *)malloc(sizeof(double) * 15 * 15 * 15 * 15);
    memset(G, 0, sizeof(double) * n * n);
    for (long t = 0; t < nq; t++) {
        int sa = abcd[4 * t], sb = abcd[4 * t + 1], sc = abcd[4 * t + 2], sd = abcd[4 * t + 3];
        const Shell *SA = &B->sh[sa], *SB = &B->sh[sb], *SC = &B->sh[sc], *SD = &B->sh[sd];
        orc_eri_shell_quartet(B, sa, sb, sc, sd, buf);
        for (int fa = 0; fa < SA->nfunc; fa++) for (int fb = 0; fb < SB->nfunc; fb++)
        for (int fc = 0; fc < SC->nfunc; fc++) for (int fd = 0; fd < SD->nfunc; fd++) {
            double v = buf[((fa * SB->nfunc + fb) * SC->nfunc + fc) * SD->nfunc + fd];
            int i = SA->off + fa, j = SB->off + fb, k = SC->off + fc, l = SD->off + fd;
            int cand[8][4] = {{i,j,k,l},{j,i,k,l},{i,j,l,k},{j,i,l,k},{k,l,i,j},{l,k,i,j},{k,l,j,i},{l,k,j,i}};
            /* all 8 index images are taken; images that coincide because SHELLS coincide are compensated by the
             * shell-level weight (an element and its in-shell transposes are separate loop iterations) */
            double shellw = 1.0;
            if (sa == sb) shellw *= 0.5;
            if (sc == sd) shellw *= 0.5;
            if (sa == sc && sb == sd) shellw *= 0.5;
            /* with shell-level weights every one of the 8 images counts once: */
            for (int c = 0; c < 8; c++) {
                int *x = cand[c];
                G[x[0] * n + x[1]] += shellw * D[x[2] * n + x[3]] * v;             /* J */
                G[x[0] * n + x[2]] -= 0.5 * shellw * D[x[1] * n + x[3]] * v;       /* K */
            }
        }
    }
    free(buf);
}

static double energy_half_trace(int n, const double *D, const double *H, const double *G) {
    /* 0.5 * tr(D (2H + G)) (rhf.rs:84-85) */
    double e = 0.0;
    for (int i = 0; i < n; i++) for (int k = 0; k < n; k++) e += D[i * n + k] * (2.0 * H[k * n + i] + G[k * n + i]);
    return 0.5 * e;
}

static void commutator(int n, const double *F, const double *D, const double *S, double *E) {
    /* F D S - S D F (rhf.rs:71) */
    double *t1 = (double *)malloc(sizeof(double) * n * n), *t2 = (double *)malloc(sizeof(double) * n * n);
    double *t3 = (double *)malloc(sizeof(double) * n * n);
    matmul(n, F, D, t1); matmul(n, t1, S, t2);
    matmul(n, S, D, t1); matmul(n, t1, F, t3);
    for (int k = 0; k < n * n; k++) E[k] = t2[k] - t3[k];
    free(t1); free(t2); free(t3);
}

/* Outputs shared by both drivers.  status: 0 converged, 1 not converged (None), 2 DIIS singular (panic) */
typedef struct {
    double electronic_energy, nuclear_repulsion;
    long iterations;
    int status;
} OrcResult;

/* restricted_hartree_fock (rhf.rs:32-108).  S,T,V,I may be NULL (computed here).  Dout/Cout optional. */
int orc_rhf(const Basis *B, long max_iterations, double epsilon, const double *eri_in, double *orbital_energies,
            OrcResult *res, double *Dout, long *trace_len, double *trace_energy, double *trace_rms) {
    int n = B->nbasis, nn = n * n;
    int nelec = 0; for (int a = 0; a < B->natoms; a++) nelec += B->Z[a];
    size_t n4 = (size_t)nn * nn;
    double *S = (double *)malloc(sizeof(double) * nn), *Tk = (double *)malloc(sizeof(double) * nn), *Vn = (double *)malloc(sizeof(double) * nn);
    double *H = (double *)malloc(sizeof(double) * nn), *X = (double *)malloc(sizeof(double) * nn), *D = (double *)malloc(sizeof(double) * nn);
    double *G = (double *)malloc(sizeof(double) * nn), *F = (double *)malloc(sizeof(double) * nn), *Er = (double *)malloc(sizeof(double) * nn);
    double *Fd = (double *)malloc(sizeof(double) * nn), *t = (double *)malloc(sizeof(double) * nn), *Fp = (double *)malloc(sizeof(double) * nn);
    double *Cp = (double *)malloc(sizeof(double) * nn), *C = (double *)malloc(sizeof(double) * nn), *Dn = (double *)malloc(sizeof(double) * nn);
    double *w = (double *)malloc(sizeof(double) * n);
    res->nuclear_repulsion = orc_nuclear_repulsion(B);
    orc_overlap(B, S); orc_kinetic(B, Tk); orc_nuclear(B, Vn);
    double *I = NULL;
    if (!eri_in) { I = (double *)malloc(sizeof(double) * n4); orc_eri_full(B, I); eri_in = I; }
    for (int k = 0; k < nn; k++) H[k] = Tk[k] + Vn[k];
    orc_transformation_matrix(n, S, X);
    huckel_density(n, H, S, X, nelec / 2, 2.0, D);
    double *T4 = (double *)malloc(sizeof(double) * n4);
    orc_antisym_tensor(n, eri_in, T4);
    Diis *diis = diis_new(4, 6, n);
    res->status = 1; res->iterations = 0; res->electronic_energy = 0.0;
    if (trace_len) *trace_len = 0;
    for (long it = 0; it <= max_iterations; it++) {
        orc_g_rhf(n, D, T4, G);
        for (int k = 0; k < nn; k++) F[k] = H[k] + G[k];
        commutator(n, F, D, S, Er);
        if (!diis_fock(diis, Er, F, Fd)) { res->status = 2; break; }
        matmul(n, Fd, X, t); matmul_tn(n, X, t, Fp);
        orc_sorted_eigs(n, Fp, Cp, w);
        matmul(n, X, Cp, C);
        updated_density(n, C, nelec / 2, 2.0, Dn);
        double rms = 0.0;
        for (int i = 0; i < n; i++) { double d = Dn[i * n + i] - D[i * n + i]; rms += d * d; }
        for (int k = 0; k < nn; k++) D[k] += (Dn[k] - D[k]) * 1.0;
        double e = energy_half_trace(n, D, H, G);
        rms = sqrt(rms / n);
        if (trace_len) { trace_energy[*trace_len] = e; trace_rms[*trace_len] = rms; (*trace_len)++; }
        if (rms < epsilon) {
            res->electronic_energy = e; res->iterations = it; res->status = 0;
            memcpy(orbital_energies, w, sizeof(double) * n);
            break;
        }
    }
    if (Dout) memcpy(Dout, D, sizeof(double) * nn);
    diis_free(diis);
    free(S); free(Tk); free(Vn); free(H); free(X); free(D); free(G); free(F); free(Er); free(Fd); free(t); free(Fp);
    free(Cp); free(C); free(Dn); free(w); free(T4); free(I);
    return res->status;
}

/* unrestricted_hartree_fock (uhf.rs:36-167).  n_alpha/n_beta < 0 => the reference's rule N/2 (uhf.rs:43-45);
 * other values are this build's extension and have no reference counterpart. */
int orc_uhf_traced(const Basis *B, long max_iterations, double epsilon, int n_alpha, int n_beta, const double *eri_in,
                   double *eps_a, double *eps_b, OrcResult *res, double *Da_out, double *Db_out,
                   long *trace_len, double *trace_energy, double *trace_rms) {
    int n = B->nbasis, nn = n * n;
    int nelec = 0; for (int a = 0; a < B->natoms; a++) nelec += B->Z[a];
    if (n_alpha < 0) n_alpha = nelec / 2;
    if (n_beta < 0) n_beta = nelec / 2;
    int nocc[2] = { n_alpha, n_beta };
    size_t n4 = (size_t)nn * nn;
    double *S = (double *)malloc(sizeof(double) * nn), *Tk = (double *)malloc(sizeof(double) * nn), *Vn = (double *)malloc(sizeof(double) * nn);
    double *H = (double *)malloc(sizeof(double) * nn), *X = (double *)malloc(sizeof(double) * nn);
    double *D[2], *G[2], *C[2], *w[2];
    for (int s = 0; s < 2; s++) { D[s] = (double *)malloc(sizeof(double) * nn); G[s] = (double *)calloc(nn, sizeof(double));
        C[s] = (double *)calloc(nn, sizeof(double)); w[s] = (double *)calloc(n, sizeof(double)); }
    double *F = (double *)malloc(sizeof(double) * nn), *Er = (double *)malloc(sizeof(double) * nn), *Fd = (double *)malloc(sizeof(double) * nn);
    double *t = (double *)malloc(sizeof(double) * nn), *Fp = (double *)malloc(sizeof(double) * nn), *Cp = (double *)malloc(sizeof(double) * nn);
    double *Dn = (double *)malloc(sizeof(double) * nn);
    res->nuclear_repulsion = orc_nuclear_repulsion(B);
    orc_overlap(B, S); orc_kinetic(B, Tk); orc_nuclear(B, Vn);
    double *I = NULL;
    if (!eri_in) { I = (double *)malloc(sizeof(double) * n4); orc_eri_full(B, I); eri_in = I; }
    for (int k = 0; k < nn; k++) H[k] = Tk[k] + Vn[k];
    orc_transformation_matrix(n, S, X);
    huckel_density(n, H, S, X, n_alpha, 1.0, D[0]);
    huckel_density(n, H, S, X, n_beta, 1.0, D[1]);
    Diis *diis[2] = { diis_new(2, 8, n), diis_new(2, 8, n) };
    res->status = 1; res->iterations = 0; res->electronic_energy = 0.0;
    if (trace_len) *trace_len = 0;
    for (long it = 0; it <= max_iterations && res->status == 1; it++) {
        for (int s = 0; s < 2; s++) {
            orc_g_uhf(n, D[s], D[1 - s], eri_in, G[s]);
            for (int k = 0; k < nn; k++) F[k] = H[k] + G[s][k];
            commutator(n, F, D[s], S, Er);
            if (!diis_fock(diis[s], Er, F, Fd)) { res->status = 2; break; }
            matmul(n, Fd, X, t); matmul_tn(n, X, t, Fp);
            orc_sorted_eigs(n, Fp, Cp, w[s]);
            matmul(n, X, Cp, C[s]);
        }
        if (res->status == 2) break;
        double rms_sum = 0.0;
        for (int s = 0; s < 2; s++) {
            updated_density(n, C[s], nocc[s], 1.0, Dn);
            double rms = 0.0;
            for (int i = 0; i < n; i++) { double d = Dn[i * n + i] - D[s][i * n + i]; rms += d * d; }
            for (int k = 0; k < nn; k++) D[s][k] += (Dn[k] - D[s][k]) * 1.0;
            rms_sum += sqrt(rms / n);
        }
        double density_rms = rms_sum / 2.0;
        if (trace_len) {   /* per pass: the energy expression of uhf.rs:145-153 on the new densities and the rms uhf.rs:137 compares */
            trace_energy[*trace_len] = energy_half_trace(n, D[0], H, G[0]) + energy_half_trace(n, D[1], H, G[1]);
            trace_rms[*trace_len] = density_rms; (*trace_len)++;
        }
        if (density_rms / 2.0 < epsilon) {
            res->electronic_energy = energy_half_trace(n, D[0], H, G[0]) + energy_half_trace(n, D[1], H, G[1]);
            res->iterations = it; res->status = 0;
            memcpy(eps_a, w[0], sizeof(double) * n); memcpy(eps_b, w[1], sizeof(double) * n);
        }
    }
    if (Da_out) memcpy(Da_out, D[0], sizeof(double) * nn);
    if (Db_out) memcpy(Db_out, D[1], sizeof(double) * nn);
    diis_free(diis[0]); diis_free(diis[1]);
    for (int s = 0; s < 2; s++) { free(D[s]); free(G[s]); free(C[s]); free(w[s]); }
    free(S); free(Tk); free(Vn); free(H); free(X); free(F); free(Er); free(Fd); free(t); free(Fp); free(Cp); free(Dn); free(I);
    return res->status;
}

int orc_uhf(const Basis *B, long max_iterations, double epsilon, int n_alpha, int n_beta, const double *eri_in,
            double *eps_a, double *eps_b, OrcResult *res, double *Da_out, double *Db_out) {
    return orc_uhf_traced(B, max_iterations, epsilon, n_alpha, n_beta, eri_in, eps_a, eps_b, res, Da_out, Db_out, NULL, NULL, NULL);
}

/* Start-up pieces exposed so tests can compare the GPU path stage by stage */
void orc_core_guess(const Basis *B, double *S, double *H, double *X, double *D_rhf) {
    int n = B->nbasis, nn = n * n;
    int nelec = 0; for (int a = 0; a < B->natoms; a++) nelec += B->Z[a];
    double *Tk = (double *)malloc(sizeof(double) * nn), *Vn = (double *)malloc(sizeof(double) * nn);
    orc_overlap(B, S); orc_kinetic(B, Tk); orc_nuclear(B, Vn);
    for (int k = 0; k < nn; k++) H[k] = Tk[k] + Vn[k];
    orc_transformation_matrix(n, S, X);
    if (D_rhf) huckel_density(n, H, S, X, nelec / 2, 2.0, D_rhf);
    free(Tk); free(Vn);
}

void orc_boys(int nmax, double x, double *F) { boys(nmax, x, F); }
