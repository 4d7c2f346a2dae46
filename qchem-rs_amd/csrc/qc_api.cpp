// qc_api.cpp - the C ABI (include/qchem_hip.h) and the host-side SCF drivers.
//
// qc_scf_rhf / qc_scf_uhf restate the control flow of restricted_hartree_fock (core/src/hf/rhf.rs:32-108) and
// unrestricted_hartree_fock (uhf.rs:36-167) - guess, DIIS windows, update order, energy expression, diagonal-only
// convergence test - with every matrix resident in HBM and every step a HIP kernel, the DIIS (<= 9 x 9) QR solve
// included; the host takes the convergence decision from two scalars it reads back once per pass.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <new>

#include "qc_internal.h"

// ---- RCCL, bound at run time.  A process that has imported torch already maps torch's bundled librccl.so.1; linking a
// second copy by path would leave it to the dynamic loader which of the two same-soname libraries the symbols resolve to.
// The library is therefore not linked: the first qc_comm_* call takes (1) $QC_RCCL_LIB if set, else (2) the librccl.so.1
// already mapped into the process (one RCCL per process: torch's, when torch is there), else (3) librccl.so.1 from the
// loader's search path (rpath /opt/rocm/lib).  qc_rccl_info() reports which one it was.
struct QcRccl {
    void *handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok = false;
};
static QcRccl &qc_rccl() {
    // (on the heap and never destroyed: a handle with a communicator may be released after the static destructors have run)
    static QcRccl &R = *new QcRccl([] {
        QcRccl r;
        const char *env = getenv("QC_RCCL_LIB");
        if (env && *env) { r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL); r.path = env; }
        if (!r.handle) { r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); r.path = "librccl.so.1 (already mapped in this process)"; }
        if (!r.handle) { r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); r.path = "librccl.so.1 (loader search path)"; }
        if (!r.handle) { fprintf(stderr, "qchem_hip: cannot load librccl.so.1: %s\n", dlerror()); return r; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
        if (r.ok) {
            Dl_info info;
            if (dladdr(reinterpret_cast<void *>(r.AllReduce), &info) && info.dli_fname) r.path = info.dli_fname;
        }
        return r;
    }());
    return R;
}

namespace {

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// hipEvent that is destroyed on every path out of its scope
struct Event {
    hipEvent_t e = nullptr;
    int create() { return hipEventCreate(&e) == hipSuccess ? QC_OK : QC_ERR_HIP; }
    ~Event() { if (e) (void)hipEventDestroy(e); }
};

struct DevBuf {
    double *p = nullptr;
    int alloc(size_t count) { return hipMalloc(&p, count * sizeof(double)) == hipSuccess ? QC_OK : QC_ERR_HIP; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// Diis (diis.rs:6-60) with the sample window, the B matrix and the QR solve in HBM / on the device: nothing of it
// synchronises with the host.  Samples live in ring slots; `slots` lists them newest first.  A singular system
// ("DIIS failed", rhf.rs:73) raises *d_flag, which the SCF step reads back together with the energy.
struct DeviceDiis {
    int minlen, maxlen, n;
    std::deque<int> slots;
    std::vector<double *> pool;            // err of slot s = pool[2s], fock = pool[2s + 1]
    double *d_dots = nullptr, *d_B = nullptr, *d_c = nullptr;
    DeviceDiis(int mn, int mx, int n_) : minlen(mn), maxlen(mx), n(n_) {}
    ~DeviceDiis() {
        for (auto p : pool) (void)hipFree(p);
        if (d_dots) (void)hipFree(d_dots);
        if (d_B) (void)hipFree(d_B);
        if (d_c) (void)hipFree(d_c);
    }
    int init() {
        if (maxlen > 11) return QC_ERR_INVALID;
        for (int i = 0; i < 2 * maxlen; ++i) { double *p; if (hipMalloc(&p, sizeof(double) * n * n) != hipSuccess) return QC_ERR_HIP; pool.push_back(p); }
        if (hipMalloc(&d_dots, 16 * sizeof(double)) != hipSuccess || hipMalloc(&d_c, 16 * sizeof(double)) != hipSuccess ||
            hipMalloc(&d_B, sizeof(double) * maxlen * maxlen) != hipSuccess) return QC_ERR_HIP;
        return hipMemset(d_B, 0, sizeof(double) * maxlen * maxlen) == hipSuccess ? QC_OK : QC_ERR_HIP;
    }
    // claim the slot of the next sample (push_front + truncate, diis.rs:29-30: the oldest slot is recycled); the caller
    // writes the error and Fock matrices straight into the returned buffers
    void next_sample(double **d_err, double **d_fock) {
        int s;
        if ((int)slots.size() == maxlen) { s = slots.back(); slots.pop_back(); } else s = (int)slots.size();
        slots.push_front(s);
        *d_err = pool[2 * s]; *d_fock = pool[2 * s + 1];
    }
    // the buffers next_sample will hand out next (the speculative build of the next pass writes its F = H + G there)
    void peek_next(double **d_err, double **d_fock) const {
        const int s = (int)slots.size() == maxlen ? slots.back() : (int)slots.size();
        *d_err = pool[2 * s]; *d_fock = pool[2 * s + 1];
    }
    // enqueues: new row of B, coefficient solve, extrapolated Fock matrix into d_out
    int extrapolate(hipStream_t st, double *d_out, int *d_flag) {
        const int m = (int)slots.size();
        const double *ys[12], *fs[12];
        int sl[12];
        for (int j = 0; j < m; ++j) { sl[j] = slots[j]; ys[j] = pool[2 * slots[j]]; fs[j] = pool[2 * slots[j] + 1]; }
        qc_dots(st, n, ys[0], ys, m, d_dots);                                         // <e_0, e_j>, diis.rs:43-45
        // (sensitivity probe, QC_DIIS_PERTURB: one dot product moved by one unit in the last place - what a different summation
        // order does - to see how far a run's trajectory depends on such bits)
        static const bool perturb = getenv("QC_DIIS_PERTURB") != nullptr;
        if (perturb && m > 1) qc_axpby(st, 1, 1.0 + 0x1p-52, d_dots + 1, 0.0, nullptr, d_dots + 1);
        qc_diis_solve(st, m, minlen, maxlen, sl, d_dots, d_B, d_c, d_flag);           // (1, 0, ...) while m < minlen
        qc_lincomb_dev(st, n, fs, d_c, m, d_out);                                     // diis.rs:52-58
        return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
    }
};

constexpr int QC_SYNC_WORDS = 12;          // 4 doubles + 16 ints of pass scalars, as 64-bit words

struct ScfWork {
    int n;
    // (work buffers of a Roothaan step come in two sets: the two spins of a UHF pass run at the same time on two streams, set b = spin)
    DevBuf H, S, X, t1[2], t2[2], t3[2], t4[2], Fp[2], Cp, C, w, ework[2], Fd[2], scal, small[2], CpPrev[2], CpNew[2], Fps[2], X0[2], tri[2];
    bool have_prev[2] = {false, false};
    // Open-shell runs (n_alpha != n_beta) use the rotation-based eigensolvers only.  Their SCF solutions of interest include
    // saddles of the UHF functional that are kept by spatial symmetry alone (O2 triplet, BASELINE config 4): Jacobi rotations
    // never mix functions that are not coupled, so symmetry-equivalent blocks of F' get bit-identical treatment and the
    // iteration stays on the symmetric determinant exactly like the reference's fixed sequence of operations; Householder
    // reflectors mix everything and seed the unstable direction with rounding noise (measured: the run then leaves the saddle
    // for the 0.024 Eh lower broken-symmetry determinant after ~60 passes).
    bool rotations_only = false;
    bool small_fused = false;              // n <= QC_SMALL_MAXN: the Roothaan step runs as one workgroup with its matrices in LDS (qc_scf_small.hip)
    bool cold[2] = {false, false};         // this pass's eigensolve of the spin started from the tridiagonal path (no previous vectors involved)
    int npass[2] = {3, 3};                 // refinement passes enqueued per eigensolve (follows what the last one needed)
    int mode[2] = {2, 2};                  // eigensolve of the next pass: 0 refinement, 1 two Jacobi sweeps + refinement, 2 Jacobi
    int *ctl = nullptr;                    // device control words: [4 s + 0..3] eigen-refinement of spin s, [8] DIIS failure
    double *h_scal = nullptr;              // pinned read-back: 4 doubles (energy, rms^2 per spin) + 16 ints, then (multi-rank) their complements
    unsigned long long *d_sync = nullptr;  // multi-rank runs: the same 12 words + their bitwise complements, all-reduced (max) across the ranks
    ~ScfWork() {
        if (ctl) (void)hipFree(ctl);
        if (h_scal) (void)hipHostFree(h_scal);
        if (d_sync) (void)hipFree(d_sync);
    }
    int init(int n_, int nsets) {
        n = n_;
        const size_t nn = (size_t)n * n;
        DevBuf *all[] = {&H, &S, &X, &Cp, &C, &CpPrev[0], &CpPrev[1], &CpNew[0], &CpNew[1], &Fps[0], &Fps[1]};
        for (auto b : all) if (b->alloc(nn) != QC_OK) return QC_ERR_HIP;
        for (int b = 0; b < nsets; ++b) {
            DevBuf *set[] = {&t1[b], &t2[b], &t3[b], &t4[b], &Fp[b], &ework[b], &Fd[b], &X0[b]};
            for (auto x : set) if (x->alloc(nn) != QC_OK) return QC_ERR_HIP;
            if (tri[b].alloc(qc_eig_tridiag_work_doubles(n)) != QC_OK || small[b].alloc(qc_eig_small_doubles(n)) != QC_OK) return QC_ERR_HIP;
        }
        if (w.alloc(n) != QC_OK || scal.alloc(16) != QC_OK) return QC_ERR_HIP;
        if (hipMalloc(&ctl, 16 * sizeof(int)) != hipSuccess || hipMemset(ctl, 0, 16 * sizeof(int)) != hipSuccess) return QC_ERR_HIP;
        if (hipHostMalloc(&h_scal, (2 * QC_SYNC_WORDS + 1) * sizeof(double)) != hipSuccess) return QC_ERR_HIP;
        std::memset(h_scal, 0, (2 * QC_SYNC_WORDS + 1) * sizeof(double));     // (the last word: sequence number of the pass, see scf_iterate)
        if (hipMalloc(&d_sync, 2 * QC_SYNC_WORDS * sizeof(unsigned long long)) != hipSuccess) return QC_ERR_HIP;
        return QC_OK;
    }
};

// sorted_eigs on device (utils.rs:20-36): Fp -> (Cp, w)
int device_sorted_eigs(qc_system *S, ScfWork &W, double *dA, double *dV, double *dw) {
    // (set-up eigensolves: synchronous; ctl[12..15] scratch, ctl[9]: Jacobi sweeps ran out)
    if (W.rotations_only) return qc_eig_device(S->stream, W.n, dA, dV, dw, W.ework[0].p, 40, 1e-9, W.ctl + 9);
    return qc_eig_cold_sync(S->stream, W.n, dA, W.X0[0].p, W.tri[0].p, dV, dw, W.ework[0].p, W.t1[0].p, W.t2[0].p, W.t3[0].p, W.t4[0].p, W.small[0].p, W.ctl + 12, W.ctl + 9);
}

// start-up shared by both drivers: H = T + V, X = S^-1/2 (rhf.rs:124-131), Hückel matrix (rhf.rs:141-143)
int scf_setup(qc_system *S, ScfWork &W, std::vector<double> &h_eht) {
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n;
    // S, T, V on the device (molint::overlap / kinetic / nuclear, rhf.rs:41-43); H = T + V (rhf.rs:48)
    hipStream_t st = S->stream;
    int rc1 = qc_one_electron_device(S, 0, W.S.p);
    if (rc1 == QC_OK) rc1 = qc_one_electron_device(S, 1, W.t1[0].p);
    if (rc1 == QC_OK) rc1 = qc_one_electron_device(S, 2, W.t2[0].p);
    if (rc1 != QC_OK) return rc1;
    qc_axpby(st, n, 1.0, W.t1[0].p, 1.0, W.t2[0].p, W.H.p);
    std::vector<double> s(nn), h(nn);
    QC_HIP_CHECK(hipMemcpyAsync(s.data(), W.S.p, nn * sizeof(double), hipMemcpyDeviceToHost, st));
    QC_HIP_CHECK(hipMemcpyAsync(h.data(), W.H.p, nn * sizeof(double), hipMemcpyDeviceToHost, st));
    QC_HIP_CHECK(hipStreamSynchronize(st));
    h_eht.assign(nn, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j)
            h_eht[(size_t)i * n + j] = h_eht[(size_t)j * n + i] = 1.75 * s[(size_t)i * n + j] * (h[(size_t)i * n + i] + h[(size_t)j * n + j]) / 2.0;
    // X = U (diag((U^T S U)_ii^-1/2) U^T): note the diagonal of the product, not the returned eigenvalues
    int rc = device_sorted_eigs(S, W, W.S.p, W.Cp.p, W.w.p);          // U (column order is immaterial for X)
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, W.S.p, n, false, W.Cp.p, n, false, 0.0, W.t1[0].p, n);      // S U
    qc_gemm(st, n, n, n, 1.0, W.Cp.p, n, true, W.t1[0].p, n, false, 0.0, W.t2[0].p, n);      // U^T (S U)
    qc_scale_cols_invsqrt(st, n, W.Cp.p, W.t2[0].p, W.t1[0].p);                               // U diag^-1/2
    qc_gemm(st, n, n, n, 1.0, W.t1[0].p, n, false, W.Cp.p, n, true, 0.0, W.X.p, n);       // (.) U^T
    return QC_OK;
}

// compute_hückel_density (rhf.rs:133-150): D = factor * C_occ C_occ^T with C = X eigvecs(X^T H_eht X)
int huckel_density(qc_system *S, ScfWork &W, const std::vector<double> &h_eht, int nocc, double factor, double *dD) {
    const int n = S->nbasis;
    hipStream_t st = S->stream;
    QC_HIP_CHECK(hipMemcpyAsync(W.Fd[0].p, h_eht.data(), h_eht.size() * sizeof(double), hipMemcpyHostToDevice, st));
    qc_gemm(st, n, n, n, 1.0, W.Fd[0].p, n, false, W.X.p, n, false, 0.0, W.t1[0].p, n);
    qc_gemm(st, n, n, n, 1.0, W.X.p, n, true, W.t1[0].p, n, false, 0.0, W.Fp[0].p, n);
    int rc = device_sorted_eigs(S, W, W.Fp[0].p, W.Cp.p, W.w.p);
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, W.X.p, n, false, W.Cp.p, n, false, 0.0, W.C.p, n);
    if (nocc > 0) qc_gemm(st, n, n, nocc, factor, W.C.p, n, false, W.C.p, n, true, 0.0, dD, n);
    else QC_HIP_CHECK(hipMemsetAsync(dD, 0, sizeof(double) * n * n, st));
    return QC_OK;
}

// One spin's Roothaan step, enqueued without any host synchronisation: F = H + G; e = FDS - SDF; DIIS; F' = X^T F X;
// eigenvectors; C = X C'   (rhf.rs:70-76).  The eigensolve is warm-started from this spin's previous vectors once they
// exist (qc_eig_refine_async: outcome in ctl[4 spin]); new vectors go to CpNew[spin], C to dC.
// (`st`, `b`: the stream the step is enqueued on and its set of work buffers - the two spins of a UHF pass are independent until their
// scalars meet and run side by side, scf_iterate)
int roothaan_enqueue(qc_system *S, ScfWork &W, DeviceDiis &diis, const double *dG, const double *dD, double *dw_out, double *dC, int spin,
                     double *dE, double *dF, bool have_F, hipStream_t st, int b) {
    const int n = S->nbasis;
    if (!have_F) qc_axpby(st, n, 1.0, W.H.p, 1.0, dG, dF);                               // F (else written by the build's closing kernel)
    qc_gemm(st, n, n, n, 1.0, dF, n, false, dD, n, false, 0.0, W.t2[b].p, n);               // F D
    qc_gemm(st, n, n, n, 1.0, W.t2[b].p, n, false, W.S.p, n, false, 0.0, W.Fp[b].p, n);        // F D S
    qc_sub_transpose(st, n, W.Fp[b].p, dE);                                                 // e = FDS - (FDS)^T = FDS - SDF
    int rc = diis.extrapolate(st, W.Fd[b].p, W.ctl + 8);
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, W.Fd[b].p, n, false, W.X.p, n, false, 0.0, W.t1[b].p, n);        // F X
    qc_gemm(st, n, n, n, 1.0, W.X.p, n, true, W.t1[b].p, n, false, 0.0, W.Fps[spin].p, n);  // X^T (F X)
    // Eigensolve.  Near convergence (mode 0): GEMM refinement from this spin's previous vectors.  Otherwise - first pass, or the
    // density still moves by more than 1e-3 per element - the tridiagonal path (start vectors from qc_eig_tridiag.hip + the same
    // refinement); matrices below QC_TRI_MIN_N go to the single-workgroup Jacobi kernels.  Outcome in ctl[4 spin].
    W.cold[spin] = false;
    static const bool force_jacobi = getenv("QC_EIG_JACOBI") != nullptr;      // A/B switch: no tridiagonal path
    if (W.have_prev[spin] && W.mode[spin] == 0)
        rc = qc_eig_refine_async(st, n, W.Fps[spin].p, W.CpPrev[spin].p, W.CpNew[spin].p, dw_out, W.ework[b].p, W.t1[b].p, W.t2[b].p, W.t3[b].p, W.t4[b].p,
                                 W.small[b].p, W.ctl + 4 * spin, W.npass[spin]);
    else if (qc_tri_ok(n) && !force_jacobi && !W.rotations_only) {
        W.cold[spin] = true;
        // (three refinement passes are enqueued: two finish most starts - the third is then five empty launches - but near-degenerate
        // clusters of a nearly converged benzene need it, and running out of passes costs a Jacobi eigensolve)
        rc = qc_eig_cold_async(st, n, W.Fps[spin].p, W.X0[b].p, W.tri[b].p, W.CpNew[spin].p, dw_out, W.ework[b].p, W.t1[b].p, W.t2[b].p, W.t3[b].p, W.t4[b].p, W.small[b].p,
                               W.ctl + 4 * spin, 3);
    } else if (W.have_prev[spin])
        rc = qc_eig_device_warm(st, n, W.Fps[spin].p, W.CpPrev[spin].p, W.CpNew[spin].p, dw_out, W.ework[b].p, W.t1[b].p, W.t2[b].p, 40, 1e-9, W.ctl + 9);
    else rc = qc_eig_device(st, n, W.Fps[spin].p, W.CpNew[spin].p, dw_out, W.ework[b].p, 40, 1e-9, W.ctl + 9);   // sorted_eigs (rhf.rs:75), cold
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, W.X.p, n, false, W.CpNew[spin].p, n, false, 0.0, dC, n);   // C = X C'
    return QC_OK;
}

// The same step for n <= QC_SMALL_MAXN, density / energy / rms of rhf.rs:78-88 included: one launch when the eigensolve is a refinement from
// the previous vectors, pre | tridiagonal start | refine + post when it starts cold, pre | Jacobi kernel | post for the rotation-only runs.
struct SmallTail { int nocc; double dfac; double *Dn; const double *Dold; double *scal_out; int *ctl_all, *ctl_out; double *fxs_out; unsigned *seq_out = nullptr; unsigned seq = 0;
                   unsigned *fork_words = nullptr; unsigned fork_seq = 0; double eps = 0.0; unsigned *h_cancel = nullptr; };
int roothaan_small(qc_system *S, ScfWork &W, DeviceDiis &diis, const double *dG, const double *dD, double *dw_out, double *dC, int spin,
                   double *dE, double *dF, bool have_F, const SmallTail &tl, hipStream_t st_in = nullptr, int b = 0) {
    const int n = S->nbasis;
    hipStream_t st = st_in ? st_in : S->stream;               // (`st_in`, `b`: the beta step of a spin-parallel pass - side stream, second set of work buffers)
    QcSmallArgs a{};
    a.n = n;
    a.F = have_F ? dF : nullptr; a.F_out = dF;
    a.D = dD; a.S = W.S.p; a.X = W.X.p; a.H = W.H.p; a.G = dG;
    a.E_out = dE;
    a.m = (int)diis.slots.size(); a.minlen = diis.minlen; a.maxlen = diis.maxlen;
    a.dots_generic = W.rotations_only ? 1 : 0;
    for (int j = 0; j < a.m; ++j) { a.slot[j] = diis.slots[j]; a.errs[j] = diis.pool[2 * diis.slots[j]]; a.focks[j] = diis.pool[2 * diis.slots[j] + 1]; }
    a.Bmat = diis.d_B; a.c_out = diis.d_c; a.diis_flag = W.ctl + 8;
    a.Fp = W.Fps[spin].p;
    a.ctl = W.ctl + 4 * spin;
    a.Cp_out = W.CpNew[spin].p; a.w_out = dw_out; a.C_out = dC; a.Dn = tl.Dn; a.Dold = tl.Dold; a.nocc = tl.nocc; a.dfac = tl.dfac;
    a.scal_out = tl.scal_out; a.ctl_all = tl.ctl_all; a.ctl_out = tl.ctl_out; a.fxs_out = tl.fxs_out; a.imax = S->imax;
    a.seq_out = tl.seq_out; a.seq = tl.seq;
    a.fork_words = tl.fork_words; a.fork_seq = tl.fork_seq; a.eps = tl.eps; a.h_cancel = tl.h_cancel;
    a.tl = S->tl_cur ? S->tl_cur + QC_TL_W * (QC_NUNITS + 2) : nullptr;
    W.cold[spin] = false;
    static const bool force_jacobi = getenv("QC_EIG_JACOBI") != nullptr;
    int rc;
    if (W.have_prev[spin] && W.mode[spin] == 0) {
        a.phases = 7; a.V0 = W.CpPrev[spin].p; a.npass = W.npass[spin];
        return qc_scf_small_launch(st, a);
    }
    QcSmallArgs pre = a;
    pre.phases = 1; pre.ctl_all = nullptr; pre.seq_out = nullptr; pre.fork_words = nullptr;
    if (a.tl) a.tl += QC_TL_W;                              // (the second launch of the pass has its own slot)
    if ((rc = qc_scf_small_launch(st, pre)) != QC_OK) return rc;
    if (qc_tri_ok(n) && !force_jacobi && !W.rotations_only) {
        W.cold[spin] = true;
        if ((rc = qc_eig_tridiag_start(st, n, W.Fps[spin].p, W.X0[b].p, W.tri[b].p)) != QC_OK) return rc;
        a.phases = 6; a.V0 = W.X0[b].p; a.npass = 3;
        return qc_scf_small_launch(st, a);
    }
    if (W.have_prev[spin]) rc = qc_eig_device_warm(st, n, W.Fps[spin].p, W.CpPrev[spin].p, W.CpNew[spin].p, dw_out, W.ework[b].p, W.t1[b].p, W.t2[b].p, 40, 1e-9, W.ctl + 9);
    else rc = qc_eig_device(st, n, W.Fps[spin].p, W.CpNew[spin].p, dw_out, W.ework[b].p, 40, 1e-9, W.ctl + 9);
    if (rc != QC_OK) return rc;
    a.phases = 4; a.Cp_in = W.CpNew[spin].p;
    return qc_scf_small_launch(st, a);
}

// the rare repeat of a spin's eigensolve when the sync-free refinement asked for rotations (ctl = 2)
int roothaan_redo_eig(qc_system *S, ScfWork &W, double *dw_out, double *dC, int spin) {
    const int n = S->nbasis;
    hipStream_t st = S->stream;
    int rc;
    if (W.cold[spin] && W.have_prev[spin])    // the tridiagonal start was not good enough: rotations in the basis of the previous vectors
        rc = qc_eig_device_warm(st, n, W.Fps[spin].p, W.CpPrev[spin].p, W.CpNew[spin].p, dw_out, W.ework[0].p, W.t1[0].p, W.t2[0].p, 40, 1e-9, W.ctl + 9);
    else if (W.cold[spin]) rc = qc_eig_device(st, n, W.Fps[spin].p, W.CpNew[spin].p, dw_out, W.ework[0].p, 40, 1e-9, W.ctl + 9);
    else if (W.rotations_only)
        rc = qc_eig_device_refine(st, n, W.Fps[spin].p, W.CpPrev[spin].p, W.CpNew[spin].p, dw_out, W.ework[0].p, W.t1[0].p, W.t2[0].p, W.t3[0].p, W.t4[0].p, W.small[0].p, W.ctl + 9);
    else     // the refinement from the previous vectors was not perturbative after all: the tridiagonal path (its own fallback: the Jacobi kernels)
        rc = qc_eig_cold_sync(st, n, W.Fps[spin].p, W.X0[0].p, W.tri[0].p, W.CpNew[spin].p, dw_out, W.ework[0].p, W.t1[0].p, W.t2[0].p, W.t3[0].p, W.t4[0].p, W.small[0].p, W.ctl + 12, W.ctl + 9);
    if (rc != QC_OK) return rc;
    qc_gemm(st, n, n, n, 1.0, W.X.p, n, false, W.CpNew[spin].p, n, false, 0.0, dC, n);
    return QC_OK;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

int qc_system_create(int natoms, const int32_t *Z, const double *xyz, int nshells, const int32_t *shell_atom, const int32_t *shell_L,
                     const int32_t *shell_pure, const int32_t *shell_nprim, const double *exponents, const double *coefficients,
                     qc_system **out) {
    if (!out || natoms <= 0 || nshells <= 0 || !Z || !xyz || !shell_atom || !shell_L || !shell_nprim || !exponents || !coefficients) return QC_ERR_INVALID;
    qc_system *S = new (std::nothrow) qc_system();
    if (!S) return QC_ERR_INVALID;
    S->natoms = natoms; S->nshells = nshells;
    S->Z.assign(Z, Z + natoms);
    S->xyz.assign(xyz, xyz + 3 * natoms);
    size_t po = 0;
    for (int s = 0; s < nshells; ++s) {
        QcShell sh{};
        sh.atom = shell_atom[s]; sh.L = shell_L[s]; sh.nprim = shell_nprim[s];
        sh.pure = (shell_pure && shell_pure[s] && sh.L >= 2) ? 1 : 0;
        if (sh.atom < 0 || sh.atom >= natoms || sh.nprim <= 0 || sh.L < 0) { delete S; return QC_ERR_INVALID; }
        if (sh.L > QC_LMAX) { delete S; return QC_ERR_UNSUPPORTED; }
        sh.exps.assign(exponents + po, exponents + po + sh.nprim);
        sh.coefs.assign(coefficients + po, coefficients + po + sh.nprim);
        po += sh.nprim;
        S->shells.push_back(std::move(sh));
    }
    if (const char *e = getenv("QC_ACCUM")) S->accum_fx = std::strcmp(e, "f64") == 0 ? 0 : 1;     // A/B switch: QC_ACCUM=f64
    qc_build_model(S);
    if (!S->last_error.empty()) { fprintf(stderr, "qchem_hip: %s\n", S->last_error.c_str()); delete S; return QC_ERR_UNSUPPORTED; }
    *out = S;
    return QC_OK;
}

static void system_free(qc_system *S) {
    if (S->comm) { qc_rccl().CommDestroy((ncclComm_t)S->comm); S->comm = nullptr; }
    qc_device_free(S);
    delete S;
}

// A handle with live qc_scf_state objects is kept until the last of them ends (their buffers and destructors use its stream).
void qc_system_destroy(qc_system *S) {
    if (!S) return;
    if (S->live_states > 0) { S->zombie = true; return; }
    system_free(S);
}

int qc_nbasis(const qc_system *S) { return S ? S->nbasis : QC_ERR_INVALID; }
int qc_nelectrons(const qc_system *S) { return S ? S->nelec : QC_ERR_INVALID; }
int qc_nshells(const qc_system *S) { return S ? S->nshells : QC_ERR_INVALID; }
int64_t qc_nquartets(const qc_system *S) { return S ? S->nquartets : QC_ERR_INVALID; }

double qc_nuclear_repulsion(const qc_system *S) {
    double e = 0.0;
    for (int a = 0; a < S->natoms; ++a)
        for (int b = a + 1; b < S->natoms; ++b) {
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) { const double x = S->xyz[3 * b + k] - S->xyz[3 * a + k]; d2 += x * x; }
            e += (double)(S->Z[a] * S->Z[b]) / std::sqrt(d2);
        }
    return e;
}

int qc_overlap(const qc_system *S, double *out) { if (!S || !out) return QC_ERR_INVALID; qc_host_one_electron(S, 0, out); return QC_OK; }
int qc_kinetic(const qc_system *S, double *out) { if (!S || !out) return QC_ERR_INVALID; qc_host_one_electron(S, 1, out); return QC_OK; }
int qc_nuclear(const qc_system *S, double *out) { if (!S || !out) return QC_ERR_INVALID; qc_host_one_electron(S, 2, out); return QC_OK; }

int qc_one_electron_gpu(qc_system *S, int which, double *out) {
    if (!S || !out) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    DevBuf M;
    if (M.alloc(nn) != QC_OK) return QC_ERR_HIP;
    if ((rc = qc_one_electron_device(S, which, M.p)) != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(out, M.p, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    return QC_OK;
}

int qc_set_stream(qc_system *S, void *hip_stream) {
    if (!S) return QC_ERR_INVALID;
    if (S->own_stream && S->stream) { (void)hipStreamDestroy(S->stream); S->own_stream = false; }
    S->lanes_probed = false;                                     // (which side stream shares the pipe of the caller's stream was not measured)
    S->prepared = false; S->gt_clean = false;                    // (the preliminaries of a prepared build were enqueued on the old stream)
    S->stream = (hipStream_t)hip_stream;
    if (!S->stream && S->device_ready) { QC_HIP_CHECK(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking)); S->own_stream = true; }
    return QC_OK;
}

int qc_eri_full(qc_system *S, double *out) {
    if (!S || !out) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    if (S->nranks != 1) return QC_ERR_INVALID;
    const size_t n = S->nbasis, n4 = n * n * n * n;
    DevBuf T;
    if (T.alloc(n4) != QC_OK) return QC_ERR_HIP;
    QC_HIP_CHECK(hipMemsetAsync(T.p, 0, n4 * sizeof(double), S->stream));
    rc = qc_launch_eri_full(S, T.p);
    if (rc != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(out, T.p, n4 * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    return QC_OK;
}

}  // extern "C"

// Everything of a fixed-point build that depends on the densities alone, enqueued ahead of time (the SCF pass does this as soon as
// its new density exists, so that it runs while the host turns around): zeroed accumulator planes, this build's fixed-point unit,
// the UHF density sum.  qc_fock_build_device recognises the densities and then goes straight to the class kernels.
int qc_fock_prepare_device(qc_system *S, const double *dDa, const double *dDb, bool uhf, const void *owner, bool scale_done) {
    S->prepared = false;
    S->prep_enqueued = false;
    if (!S->accum_fx) return QC_OK;
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n, plane = (size_t)QC_NREP * (uhf ? 2 : 1) * nn;
    // (the closing fold of the last build zeroes the replicas it reads: no memset then)
    const bool zero = !(S->gt_clean && S->gt_clean_nspin == (uhf ? 2 : 1));
    if (zero) QC_HIP_CHECK(hipMemsetAsync(S->d_Gtmp, 0, 2 * plane * sizeof(double), S->stream));
    S->gt_clean = true; S->gt_clean_nspin = uhf ? 2 : 1;
    if (!scale_done) qc_fx_scale(S->stream, n, dDa, uhf ? dDb : nullptr, S->imax, S->d_fxs);
    if (uhf) qc_axpby(S->stream, n, 1.0, dDa, 1.0, dDb, S->d_Dj);
    // (the build that finds these preliminaries starts its side streams without a fork event: whoever lets the host go on before the
    // handle's stream has drained must know that something was put on it here)
    S->prep_enqueued = zero || !scale_done || uhf;
    S->prepared = true; S->prep_Da = dDa; S->prep_Db = uhf ? dDb : nullptr; S->prep_owner = owner;
    return QC_OK;
}

int qc_fock_build_device(qc_system *S, const double *dDa, const double *dDb, double *dGa, double *dGb, bool uhf, int *twin_cache,
                         const double *dH, double *dFa, double *dFb, bool *f_done, const void *owner, unsigned fork_seq) {
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n;
    hipStream_t st = S->stream;
    const bool fx = S->accum_fx != 0;
    const double *fxs = fx ? S->d_fxs : nullptr;
    // Spin symmetry: the reference evaluates both spins with identical arithmetic (uhf.rs:210-227), so bitwise-equal
    // densities give bitwise-equal G (its closed-shell UHF never breaks symmetry).  The fixed-point
    // accumulation keeps that property by construction: every contribution is the same sequence of operations for either
    // spin and integer sums do not depend on their order.  Only the f64-atomic mode (kept for A/B measurements) needs help:
    // there equal spins are detected and digested once.  Inside an SCF run the answer cannot change, so the drivers pass a
    // cache and only their first build pays the host round trip.
    bool twin = false;
    if (uhf && !fx) {
        if (twin_cache && *twin_cache >= 0) twin = *twin_cache != 0;
        else {
            int diff = 1;
            QC_HIP_CHECK(hipMemsetAsync(S->d_flag, 0, sizeof(int), st));
            qc_count_diff(st, nn, dDa, dDb, S->d_flag);
            QC_HIP_CHECK(hipMemcpyAsync(&diff, S->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
            QC_HIP_CHECK(hipStreamSynchronize(st));
            twin = (diff == 0);
            if (twin_cache) *twin_cache = twin ? 1 : 0;
        }
    }
    const bool two = uhf && !twin;
    const int nspin = two ? 2 : 1;
    // accumulation phase: zero the replicas, density sum, every class kernel on the side streams, replica fold
    const size_t plane = (size_t)QC_NREP * nspin * nn;          // one accumulator plane: [replica][spin][n*n]
    const bool ready = fx && S->prepared && owner != nullptr && S->prep_owner == owner && S->prep_Da == dDa && S->prep_Db == (uhf ? dDb : nullptr);   // qc_fock_prepare_device ran for these
    S->prepared = false;
    // (a speculative build starts on the device's word, not on the host's: its preliminaries must be in the stream in front of the
    // kernel that releases the fork word - qc_fock_prepare_device for exactly these densities - and the planes clean)
    if (fork_seq && !(ready && S->gt_clean)) return QC_ERR_INVALID;
    QcFockArgs a{};
    // Replicas in use (the planes keep their layout): 8 for n <= 64, all 32 above.  Replicas spread the atomics of hot elements, and the
    // closing fold reads and zeroes every one of them: at n = 58 that is 1.7 MB with 32 replicas, and the H2O/cc-pVTZ iteration takes 0.308 ms
    // with 8 against 0.313 with 32 (0.312 with 16, 0.319 with 4, 0.349 with 2; three alternating runs each); benzene/cc-pVDZ (n = 114) shows
    // no difference between 8, 16 and 32.  QC_NREP_USE: experiment switch.
    const char *nrep_s = getenv("QC_NREP_USE");                 // (read per build: a test switches it inside one process)
    const int nrep_env = nrep_s ? std::max(1, std::min(QC_NREP, atoi(nrep_s))) : 0;
    const int nrep_use = nrep_env ? nrep_env : (n <= 64 ? 8 : QC_NREP);
    a.nrep = nrep_use; a.rep_stride = nspin * nn; a.fxs = fxs; a.fx_lo = plane; a.fork_seq = fork_seq;
    if (!ready) {
        QC_HIP_CHECK(hipMemsetAsync(S->d_Gtmp, 0, (fx ? 2 : 1) * plane * sizeof(double), st));
        if (fx) qc_fx_scale(st, n, dDa, uhf ? dDb : nullptr, S->imax, S->d_fxs);      // this build's fixed-point unit, from its densities
    }
    S->gt_clean = false;                                    // (the class kernels are about to accumulate into the planes)
    if (uhf) {
        if (!ready) qc_axpby(st, n, 1.0, dDa, 1.0, dDb, S->d_Dj);
        a.Dj = S->d_Dj; a.Dk0 = dDa; a.Dk1 = two ? dDb : nullptr; a.cK = 1.0;
    } else {
        a.Dj = dDa; a.Dk0 = dDa; a.Dk1 = nullptr; a.cK = 0.5;
    }
    a.G0 = S->d_Gtmp; a.G1 = S->d_Gtmp + nn;
    // (QC_FOLD_JOIN: the closing fold waits for the side streams' markers itself instead of sitting behind the one-lane waiting kernel - one
    // dependent launch less, 1 us per H2O/cc-pVTZ pass on the device timeline, nothing measurable per iteration (five alternating runs);
    // OFF by default: every workgroup of the fold then polls one word, 813 of them at n = 114, and three of nine benzene/cc-pVDZ runs
    // with that many pollers next to an experimental replica count lost a marker for 20 s - not understood, not reproduced since, not shipped)
    static const bool fold_join = getenv("QC_FOLD_JOIN") != nullptr;
    a.fold_joins = fx && !S->comm && S->nranks == 1 && fold_join;
    int rc = qc_launch_fock_classes(S, a, nullptr, nullptr, ready);
    if (rc != QC_OK) return rc;
    if (fx && !S->comm && S->nranks == 1) {
        // (one launch instead of fold + symmetrise: nothing needs the folded planes; and the join of the side streams in the same launch)
        const bool fj = S->fold_join_pending;
        S->fold_join_pending = false;
        qc_fold_symmetrize(st, n, nrep_use, nspin * nn, S->d_Gtmp, plane, dGa, dH, dH ? dFa : nullptr, fxs, S->tl_cur ? S->tl_cur + QC_TL_W * (QC_NUNITS + 1) : nullptr,
                           fj ? S->d_join : nullptr, S->join_target, S->h_join_timeout, S->wait_limit);
        if (two) qc_fold_symmetrize(st, n, nrep_use, nspin * nn, S->d_Gtmp + nn, plane, dGb, dH, dH ? dFb : nullptr, fxs);
        else if (uhf) QC_HIP_CHECK(hipMemcpyAsync(dGb, dGa, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
        S->gt_clean = true; S->gt_clean_nspin = nspin;      // every replica element of the planes in use was read and zeroed
        if (f_done) *f_done = dH != nullptr && dFa != nullptr && (!uhf || (two && dFb != nullptr));
        return QC_OK;
    }
    qc_reduce_replicas(st, nspin * nn, nrep_use, nspin * nn, S->d_Gtmp, S->d_Gred, fx, plane);
    if (S->comm) {
        // partial Fock matrices -> full, one all-reduce per build ([Ga|Gb] concatenated for UHF; hi and lo planes back to
        // back).  Fixed-point partials are summed as integers: the result is bit-identical on every rank, whatever the ring
        // order.  (Against a build with another shard layout - the single-GPU build included - it agrees to ~1e-13, not bit for
        // bit: the bra-major kernels pre-sum the exchange rows of a 64-ket bundle in an f64 LDS buffer, and which kets share a
        // bundle depends on the shard.)
        if (qc_rccl().AllReduce(S->d_Gred, S->d_Gred, (fx ? 2 : 1) * nspin * nn, fx ? ncclInt64 : ncclDouble, ncclSum, (ncclComm_t)S->comm, st) != ncclSuccess) return QC_ERR_RCCL;
    }
    qc_symmetrize_add(st, n, S->d_Gred, nspin * nn, dGa, dH, dH ? dFa : nullptr, fxs);
    if (two) qc_symmetrize_add(st, n, S->d_Gred + nn, nspin * nn, dGb, dH, dH ? dFb : nullptr, fxs);
    else if (uhf) QC_HIP_CHECK(hipMemcpyAsync(dGb, dGa, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (f_done) *f_done = dH != nullptr && dFa != nullptr && (!uhf || (two && dFb != nullptr));
    return QC_OK;
}

extern "C" {

int qc_fock_rhf_device(qc_system *S, const double *dD, double *dG) {
    if (!S || !dD || !dG) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    return qc_fock_build_device(S, dD, nullptr, dG, nullptr, false);
}
int qc_fock_uhf_device(qc_system *S, const double *dDa, const double *dDb, double *dGa, double *dGb) {
    if (!S || !dDa || !dDb || !dGa || !dGb) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    return qc_fock_build_device(S, dDa, dDb, dGa, dGb, true);
}

int qc_fock_rhf(qc_system *S, const double *D, double *G) {
    if (!S || !D || !G) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    QC_HIP_CHECK(hipMemcpyAsync(S->d_D, D, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    rc = qc_fock_build_device(S, S->d_D, nullptr, S->d_G, nullptr, false);
    if (rc != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(G, S->d_G, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    if ((rc = qc_join_check(S)) != QC_OK) return rc;        // (a join that gave up folded an incomplete matrix: this call fails)
    qc_gate_quiet(S);
    return QC_OK;
}

int qc_fock_uhf(qc_system *S, const double *Da, const double *Db, double *Ga, double *Gb) {
    if (!S || !Da || !Db || !Ga || !Gb) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    QC_HIP_CHECK(hipMemcpyAsync(S->d_D, Da, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(S->d_D + nn, Db, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    rc = qc_fock_build_device(S, S->d_D, S->d_D + nn, S->d_G, S->d_G + nn, true);
    if (rc != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(Ga, S->d_G, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(Gb, S->d_G + nn, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    if ((rc = qc_join_check(S)) != QC_OK) return rc;
    qc_gate_quiet(S);
    return QC_OK;
}

int qc_sym_eig(qc_system *S, int n, const double *A, double *V, double *w) {
    if (!S || n <= 0 || !A || !V || !w) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)n * n;
    DevBuf dA, dV, dw, dwork, x0, tri, t1, t2, t3, t4, sm;
    DevBuf *all[] = {&dA, &dV, &dwork, &x0, &t1, &t2, &t3, &t4};
    for (auto b : all) if (b->alloc(nn) != QC_OK) return QC_ERR_HIP;
    if (dw.alloc(n) != QC_OK || sm.alloc(qc_eig_small_doubles(n)) != QC_OK || tri.alloc(qc_eig_tridiag_work_doubles(n)) != QC_OK) return QC_ERR_HIP;
    QC_HIP_CHECK(hipMemcpyAsync(dA.p, A, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    int flag = 0;
    QC_HIP_CHECK(hipMemsetAsync(S->d_flag, 0, 4 * sizeof(int), S->stream));
    DevBuf ctlb;                                           // 4 control words of the refinement; d_flag[0]: Jacobi sweeps ran out
    if (ctlb.alloc(2) != QC_OK) return QC_ERR_HIP;
    rc = qc_eig_cold_sync(S->stream, n, dA.p, x0.p, tri.p, dV.p, dw.p, dwork.p, t1.p, t2.p, t3.p, t4.p, sm.p, reinterpret_cast<int *>(ctlb.p), S->d_flag);
    if (rc != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(V, dV.p, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(w, dw.p, n * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(&flag, S->d_flag, sizeof(int), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    return flag ? QC_EIG_NOT_CONVERGED : QC_OK;
}

// ---- step-wise drivers: the host (the Rust `core` crate in the north-star design) owns the convergence loop and
// calls one FFI entry per loop-body pass; qc_scf_rhf / qc_scf_uhf below are that loop written in C++.
}  // extern "C"

struct qc_scf_state {
    qc_system *S = nullptr;
    bool uhf = false;
    int nocc[2] = {0, 0};
    ScfWork W;
    DevBuf D[2], Dn[2], Gb[2], Cs, ws;         // densities are double-buffered per spin: D <-> Dn swap when a pass is accepted
    int g_cur = 0;                             // G of this pass lives in Gb[g_cur]; the speculative build of the next pass writes Gb[g_cur ^ 1]
    double eps_hint = 0.0;                     // > 0: the host stops once the reference's stopping rule holds at this epsilon (qc_scf_set_stop_rule)
    unsigned *h_cancel = nullptr;              // pinned: number of the speculative build the device cancelled (the pass in front of it met the rule)
    // Speculative build of the next pass behind this pass's Roothaan step (scf_iterate).  OFF by default: built, parity-tested, and
    // measured SLOWER on MI355X (H2O/cc-pVTZ 0.38-0.44 ms per iteration against 0.31) - DESIGN.md 3.1 says why.  QC_SPEC=1 (read per
    // SCF state) switches it on.
    bool spec_on = getenv("QC_SPEC") != nullptr && getenv("QC_NO_SPEC") == nullptr;
    int64_t spec_hits = 0, spec_lost = 0, builds_timed = 0, passes = 0, redos = 0;
    bool spin_parallel = getenv("QC_NO_SPIN_PARALLEL") == nullptr;      // (A/B switch, read per SCF state)
    double warm_rms_env = getenv("QC_EIG_WARM_RMS") ? atof(getenv("QC_EIG_WARM_RMS")) : 0.0;   // (read per SCF state: tests reach the repeat branch with it)
    double ms_tuner = 0;
    bool cur_build_tuned = false, pend_build_tuned = false;   // the build of the current / the pending timing set contained a tuner run
    unsigned cur_build_gen = 0, pend_build_gen = 0;           // ... and ran under this stream assignment (qc_system::assign_gen)
    DevBuf T4, TK, Dtot;                       // stored mode: RHF T = I - I^x / 2; UHF I and its exchange-permuted copy
    bool stored = false;
    int twin = -1;                             // UHF spin-twin decision, taken at the first build
    double ms_tensor = 0;
    DeviceDiis *diis[2] = {nullptr, nullptr};
    // timing events of a pass (start | build done | pass done), two sets used alternately: a pass whose end the host saw through the pinned
    // sequence word reads none of them before it returns - the next pass does, after its own build has been issued
    hipEvent_t evs[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int ev_cur = 0, pending_set = 0;
    double ms_fock = 0, ms_linalg = 0, ms_setup = 0;
    unsigned pass_seq = 0;                     // sequence number of the last pass whose end the host saw through the pinned word
    bool event_wait = getenv("QC_EVENT_WAIT") != nullptr;      // (A/B switch, read per SCF state: the stream's event instead)
    bool timing_pending = false;               // ... and whose event times (set `pending_set`) have not been read yet
    ~qc_scf_state() {
        if (S && S->prep_owner == this) { S->prepared = false; S->prep_owner = nullptr; }
        delete diis[0]; delete diis[1];
        if (S && S->spec.owner == this) { S->spec.pending = false; S->spec.owner = nullptr; }
        if (S && S->stream) { (void)hipStreamSynchronize(S->stream); qc_gate_quiet(S); qc_tl_dump(S); }
        for (auto &set : evs) for (hipEvent_t e : set) if (e) (void)hipEventDestroy(e);
        if (h_cancel) (void)hipHostFree(h_cancel);
    }
};
static void system_free(qc_system *S);
static void scf_state_delete(qc_scf_state *st) {
    if (!st) return;
    qc_system *S = st->S;
    delete st;
    if (S && --S->live_states == 0 && S->zombie) system_free(S);
}

static int scf_begin(qc_system *S, bool uhf, int n_alpha, int n_beta, qc_scf_state **out) {
    if (!S || !out) return QC_ERR_INVALID;
    const double t0 = now_ms();
    static const bool sdbg = getenv("QC_SETUP_DEBUG") != nullptr;
    double tt = t0;
    auto lap = [&](const char *what) { if (sdbg) { const double t = now_ms(); fprintf(stderr, "[setup] %-28s %.3f ms\n", what, t - tt); tt = t; } };
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    lap("qc_device_init (total)");
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n;
    qc_scf_state *st = new (std::nothrow) qc_scf_state();
    if (!st) return QC_ERR_INVALID;
    struct Del { void operator()(qc_scf_state *p) const { scf_state_delete(p); } };
    std::unique_ptr<qc_scf_state, Del> guard(st);
    st->S = S; st->uhf = uhf;
    ++S->live_states;
    st->nocc[0] = st->nocc[1] = S->nelec / 2;                             // rhf.rs:176 / uhf.rs:43-45
    if (uhf && (n_alpha > 0 || n_beta > 0)) { st->nocc[0] = n_alpha; st->nocc[1] = n_beta; }
    if (st->nocc[0] < 0 || st->nocc[1] < 0 || st->nocc[0] > n || st->nocc[1] > n) return QC_ERR_INVALID;
    const int nspin = uhf ? 2 : 1;
    st->W.rotations_only = uhf && st->nocc[0] != st->nocc[1];
    // (QC_NO_SMALL_FUSED: A/B switch, the generic launch sequence.  Open-shell runs take the one-workgroup kernels too since the two spins'
    // kernels run side by side - pre | Jacobi kernel | post per spin, O2 triplet/cc-pVDZ 0.130 ms of linear algebra per pass against 0.148
    // for the generic sequence, 0.198 with the spins one after the other - with the DIIS dot products in the generic sequence's summation
    // order, QcSmallArgs::dots_generic.  Their saddle-point trajectories depend on every last bit - DESIGN.md 1 - and the two paths differ
    // in the last bit of the first pass's energy: O2 triplet reaches 1e-10 in 118 passes on this path and in 450 on the generic one, 15
    // and 15 at the CLI's 1e-6, energies 8e-6 Eh apart there.  QC_NO_OPEN_SHELL_FUSED: A/B switch.)
    static const bool open_fused = getenv("QC_NO_OPEN_SHELL_FUSED") == nullptr;
    st->W.small_fused = n <= QC_SMALL_MAXN && (!st->W.rotations_only || open_fused) && getenv("QC_NO_SMALL_FUSED") == nullptr;
    if ((rc = st->W.init(n, uhf ? 2 : 1)) != QC_OK) return rc;
    for (int s = 0; s < nspin; ++s) if (st->D[s].alloc(nn) != QC_OK || st->Dn[s].alloc(nn) != QC_OK) return QC_ERR_HIP;
    if (st->Gb[0].alloc(nspin * nn) != QC_OK || st->Gb[1].alloc(nspin * nn) != QC_OK || st->Cs.alloc(nspin * nn) != QC_OK || st->ws.alloc(nspin * n) != QC_OK) return QC_ERR_HIP;
    QC_HIP_CHECK(hipHostMalloc(&st->h_cancel, sizeof(unsigned), hipHostMallocDefault));
    *st->h_cancel = 0;
    std::vector<double> h_eht;
    lap("state buffers");
    if ((rc = scf_setup(S, st->W, h_eht)) != QC_OK) return rc;           // rhf.rs:41-49
    lap("S, T, V, X = S^-1/2");
    for (int s = 0; s < nspin; ++s)                                       // rhf.rs:50 / uhf.rs:60-63
        if ((rc = huckel_density(S, st->W, h_eht, st->nocc[s], uhf ? 1.0 : 2.0, st->D[s].p)) != QC_OK) return rc;
    lap("Hueckel guess");
    if (S->fock_mode == 1) {
        // the reference's conventional SCF: ERI tensor once (rhf.rs:45), antisymmetrised copy (rhf.rs:58-62), dense
        // contraction per pass.  8 n^4 bytes per tensor; two of them live during the build.
        if (S->comm || S->nranks != 1) return QC_ERR_UNSUPPORTED;        // sharded builds are direct-mode only
        const size_t n4 = nn * nn;
        size_t free_b = 0, total_b = 0;
        QC_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        if ((double)n4 * 8.0 * 2.2 > (double)free_b) return QC_ERR_UNSUPPORTED;
        const double tt0 = now_ms();
        DevBuf I;
        if (I.alloc(n4) != QC_OK || st->T4.alloc(n4) != QC_OK) return QC_ERR_HIP;
        QC_HIP_CHECK(hipMemsetAsync(I.p, 0, n4 * sizeof(double), S->stream));
        if ((rc = qc_launch_eri_full(S, I.p)) != QC_OK) return rc;
        if (uhf) {
            qc_permute_tensor(S->stream, n, I.p, 0.0, 1.0, st->T4.p);      // TK[i,j,k,l] = I[i,k,j,l]
            st->TK.p = st->T4.p; st->T4.p = I.p; I.p = nullptr;            // keep I (as T4) and TK
            if (st->Dtot.alloc(nn) != QC_OK) return QC_ERR_HIP;
        } else {
            qc_permute_tensor(S->stream, n, I.p, 1.0, -0.5, st->T4.p);     // electron_terms, rhf.rs:58-62
        }
        QC_HIP_CHECK(hipStreamSynchronize(S->stream));
        st->stored = true;
        st->ms_tensor = now_ms() - tt0;
    }
    for (int s = 0; s < nspin; ++s) {                                     // Diis::new(4,6) rhf.rs:65 / (2,8) uhf.rs:76-78
        st->diis[s] = uhf ? new DeviceDiis(2, 8, n) : new DeviceDiis(4, 6, n);
        if ((rc = st->diis[s]->init()) != QC_OK) return rc;
    }
    for (auto &set : st->evs) for (hipEvent_t &e : set) QC_HIP_CHECK(hipEventCreate(&e));
    int eig_flag = 0;                                                     // the eigensolves of X and of the Hueckel guess
    QC_HIP_CHECK(hipMemcpyAsync(&eig_flag, st->W.ctl + 9, sizeof(int), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    if (eig_flag) return QC_EIG_NOT_CONVERGED;
    st->ms_setup = now_ms() - t0;
    *out = guard.release();
    return QC_OK;
}

// (a pass whose end the host saw through the pinned sequence word has not read its event times yet: the last event may still have been
// in flight, and asking costs host time between two passes.  The next pass asks once its own build is out.)
static void scf_flush_timing(qc_scf_state *st) {
    if (!st->timing_pending) return;
    st->timing_pending = false;
    hipEvent_t *e = st->evs[st->pending_set];
    float ms_f = 0, ms_l = 0;
    if (hipEventSynchronize(e[2]) != hipSuccess) return;
    const bool have_f = hipEventElapsedTime(&ms_f, e[0], e[1]) == hipSuccess;
    if (hipEventElapsedTime(&ms_l, e[1], e[2]) == hipSuccess) st->ms_linalg += ms_l;
    if (!have_f) return;
    // (a build that contained a tuner run is not a sample of the build time: neither for the totals nor for the tuner's online choice)
    if (st->pend_build_tuned) return;
    st->ms_fock += ms_f; st->builds_timed += 1;
    if (!st->stored && !st->S->comm) qc_fock_feedback(st->S, ms_f, st->pend_build_gen);
}

// Wait for an event by polling (what hipStreamSynchronize does too): a parked thread's wake-up latency is longer
// than a whole SCF pass of a small molecule.  Not for ever: a kernel that never finishes is an error of the call, not a host core
// pinned for good (QC_HOST_WAIT_LIMIT_S, default 120 s; the device-side waits give up earlier and say why).
static double host_wait_limit_ms() {
    static const double lim = getenv("QC_HOST_WAIT_LIMIT_S") ? atof(getenv("QC_HOST_WAIT_LIMIT_S")) * 1e3 : 120e3;
    return lim;
}
static hipError_t wait_event(hipEvent_t ev) {
    hipError_t e;
    unsigned spins = 0;
    double t0 = 0.0;
    while ((e = hipEventQuery(ev)) == hipErrorNotReady) {
        if ((++spins & 0x3fff) == 0) {
            const double t = now_ms();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > host_wait_limit_ms()) { fprintf(stderr, "qchem_hip: an SCF pass did not finish within %.0f s\n", host_wait_limit_ms() * 1e-3); return hipErrorLaunchTimeOut; }
        }
    }
    return e;
}

// one pass of the loop body.  RHF: rhf.rs:67-88.  UHF: uhf.rs:81-137 (returns the reference's `density_rms`,
// i.e. (rms_a + rms_b) / 2, and the energy expression of uhf.rs:145-153 evaluated every pass).
// The whole pass is enqueued without looking at the device; one synchronisation at its end returns the energy, the rms
// and the control words (DIIS failure, eigen-refinement outcome).  Launch latency of ~50 small kernels then overlaps with
// their execution instead of adding to it.
//
// Round 4: the host leaves the pass boundary.  Behind the pass's Roothaan step - before waiting for it - the NEXT pass's Fock build is
// issued speculatively from the density buffer the step is about to fill: its side streams start with a one-lane kernel that waits for
// the fork word which the step's last kernel releases (qc_fock.hip, "Device-side fork"), the handle's own chain simply follows in
// stream order.  All launches of build k + 1 then sit in their queues when pass k's densities become final (before: 26 us of idle GPU per
// pass while the host saw the pass end and turned around, and launches arriving 8 us apart).  The next call of this function finds its
// build in flight and goes straight to the Roothaan step.  What makes it safe:
//  * G is double-buffered (the repeat branch below still needs the old G for the energy); F = H + G goes to the DIIS slot the next pass
//    will claim, which this pass only reads in front of it in stream order;
//  * a repeat of the eigensolve (rotations wanted) changes the density: the speculative build is then discarded - it ran into the
//    accumulator planes and its closing kernel left them clean - and the next pass builds again;
//  * when the host has said where it stops (qc_scf_set_stop_rule; scf_run does), the kernel that ends the pass evaluates the same
//    rule and cancels the build behind a converged pass on the device: its class kernels return at once;
//  * may_continue = false (scf_run's last allowed pass): nothing is issued behind the pass.
static int scf_iterate(qc_scf_state *st, double *energy, double *rms_out, bool may_continue = true) {
    qc_system *S = st->S;
    ScfWork &W = st->W;
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n;
    hipStream_t sm = S->stream;
    const int nspin = st->uhf ? 2 : 1;
    int rc;
    const double th0 = now_ms();
    qc_stamp("enter pass");
    if ((rc = qc_tl_begin_pass(S)) != QC_OK) return rc;
    st->ev_cur ^= 1;
    hipEvent_t *const ev = st->evs[st->ev_cur];
    hipEvent_t const ev0 = ev[0], ev1 = ev[1], ev2 = ev[2];
    double *const Gcur = st->Gb[st->g_cur].p, *const Gnext = st->Gb[st->g_cur ^ 1].p;
    double *dE[2] = {nullptr, nullptr}, *dF[2] = {nullptr, nullptr};       // this pass's DIIS sample buffers (error, Fock matrix) per spin
    for (int s = 0; s < nspin; ++s) st->diis[s]->next_sample(&dE[s], &dF[s]);
    bool have_F = false;
    // the build of this pass may be in flight already (issued speculatively by the previous pass)
    bool spec_hit = false;
    {
        qc_system::QcSpec &sp = S->spec;
        if (sp.pending && sp.owner == st) {
            const bool cancelled = __atomic_load_n(st->h_cancel, __ATOMIC_ACQUIRE) == sp.seq;     // (the host goes on although its own rule held: build again)
            spec_hit = !st->stored && !cancelled && sp.Da == st->D[0].p && sp.Db == (st->uhf ? st->D[1].p : nullptr) && sp.Ga == Gcur;
            sp.pending = false;
            if (spec_hit) { have_F = sp.f_done; st->spec_hits += 1; }
            else { st->spec_lost += 1; S->prepared = false; }      // (its kernels may still run: the build below forks off the handle's stream)
        }
    }
    // G of every spin from the *old* densities
    if (spec_hit) {
        // (ev0 / ev1 of this set were recorded around the build when it was issued)
    } else if (st->stored) {
        QC_HIP_CHECK(hipEventRecord(ev0, sm));
        if (st->uhf) {   // uhf.rs:216-226: G_s = <I, D_s + D_s'> - <I^x, D_s>
            qc_axpby(sm, n, 1.0, st->D[0].p, 1.0, st->D[1].p, st->Dtot.p);
            qc_axpby(sm, n, -1.0, st->D[0].p, 0.0, nullptr, W.t1[0].p);
            qc_tensor_gemv(sm, n, st->T4.p, st->Dtot.p, st->TK.p, W.t1[0].p, Gcur);
            qc_axpby(sm, n, -1.0, st->D[1].p, 0.0, nullptr, W.t1[0].p);
            qc_tensor_gemv(sm, n, st->T4.p, st->Dtot.p, st->TK.p, W.t1[0].p, Gcur + nn);
        } else {
            qc_tensor_gemv(sm, n, st->T4.p, st->D[0].p, nullptr, nullptr, Gcur);   // rhf.rs:152-167
        }
        st->cur_build_tuned = false; st->cur_build_gen = S->assign_gen;
    } else {
        QC_HIP_CHECK(hipEventRecord(ev0, sm));
        qc_stamp("ev0");
        const int tunes0 = S->tune_count;
        const double tt0 = now_ms();
        if ((rc = qc_fock_build_device(S, st->D[0].p, st->uhf ? st->D[1].p : nullptr, Gcur, st->uhf ? Gcur + nn : nullptr, st->uhf,
                                       &st->twin, W.H.p, dF[0], dF[1], &have_F, st)) != QC_OK) return rc;
        st->cur_build_tuned = S->tune_count != tunes0;
        if (st->cur_build_tuned) st->ms_tuner += now_ms() - tt0;
        st->cur_build_gen = S->assign_gen;
    }
    qc_stamp("build out");
    scf_flush_timing(st);                                                 // (the previous pass's times, now that this pass's build is out)
    if (!spec_hit) QC_HIP_CHECK(hipEventRecord(ev1, sm));
    qc_stamp("flush timing, ev1");
    // UHF: the two spins' steps are independent (uhf.rs:84-135 runs them one after the other) - the beta step goes to a side stream on
    // another dispatch pipe, behind an event of the build's closing kernel, and meets the handle's stream again before the scalars
    // (device-side join).  Same kernels, same arithmetic, per spin: results are bit for bit those of the serial order (QC_NO_SPIN_PARALLEL).
    const bool spin_par = st->uhf && !W.small_fused && st->spin_parallel && S->nlanes >= 2 && !S->join_by_events;
    if (!W.small_fused) {
        hipStream_t side = spin_par ? qc_spin_fork(S) : nullptr;
        if (spin_par && !side) return QC_ERR_HIP;
        for (int s = 0; s < nspin; ++s) {                                 // (the control words were cleared by the previous pass)
            const bool on_side = spin_par && s == 1;
            if ((rc = roothaan_enqueue(S, W, *st->diis[s], Gcur + s * nn, st->D[s].p, st->ws.p + s * n, st->Cs.p + s * nn, s, dE[s], dF[s], have_F,
                                       on_side ? side : sm, on_side ? 1 : 0)) != QC_OK) return rc;
            if (on_side) {    // the beta density on the side stream as well; then the streams meet
                if (st->nocc[s] > 0) qc_gemm(side, n, n, st->nocc[s], 1.0, st->Cs.p + s * nn, n, false, st->Cs.p + s * nn, n, true, 0.0, st->Dn[s].p, n);
                else QC_HIP_CHECK(hipMemsetAsync(st->Dn[s].p, 0, nn * sizeof(double), side));
            }
        }
        if (spin_par && (rc = qc_spin_join(S)) != QC_OK) return rc;
    }
    int *h_ctl = reinterpret_cast<int *>(W.h_scal + 4);
    // Multi-rank runs take every decision (convergence, DIIS failure, eigensolve mode, repeat) from the SAME numbers on every
    // rank: the pass scalars go to device memory, are all-reduced as bit patterns (max) together with their complements - so a
    // rank whose copy differs is noticed (max(x) != ~max(~x)) - and only then reach the host.  The replicated linear algebra is
    // deterministic and starts from bit-identical G (integer all-reduce), so the copies agree; this makes a divergence an
    // error on all ranks in the same pass instead of a hang in the next all-reduce.
    const bool multi = S->comm != nullptr;
    double *scal_out = multi ? reinterpret_cast<double *>(W.d_sync) : W.h_scal;
    int *ctl_out = multi ? reinterpret_cast<int *>(W.d_sync + 4) : h_ctl;
    bool dens_done[2] = {false, spin_par};                               // (the beta density of a spin-parallel pass was formed on the side stream)
    auto density_and_scalars = [&](int s, bool hand_over) -> int {
        if (dens_done[s]) dens_done[s] = false;                          // (once: a repeat of the eigensolve forms it again, here)
        else if (st->nocc[s] > 0) qc_gemm(sm, n, n, st->nocc[s], st->uhf ? 1.0 : 2.0, st->Cs.p + s * nn, n, false, st->Cs.p + s * nn, n, true, 0.0, st->Dn[s].p, n);
        else QC_HIP_CHECK(hipMemsetAsync(st->Dn[s].p, 0, nn * sizeof(double), sm));
        // energy and rms straight into pinned host memory; the last spin's kernel also hands over and clears the control words
        qc_energy_rms(sm, n, st->Dn[s].p, st->D[s].p, W.H.p, Gcur + s * nn, scal_out + 2 * s, hand_over ? W.ctl : nullptr, ctl_out);
        return QC_OK;
    };
    auto publish_scalars = [&]() -> int {
        if (!multi) return QC_OK;
        qc_sync_pack(sm, W.d_sync, QC_SYNC_WORDS);
        if (qc_rccl().AllReduce(W.d_sync, W.d_sync, 2 * QC_SYNC_WORDS, ncclUint64, ncclMax, (ncclComm_t)S->comm, sm) != ncclSuccess) return QC_ERR_RCCL;
        QC_HIP_CHECK(hipMemcpyAsync(W.h_scal, W.d_sync, 2 * QC_SYNC_WORDS * sizeof(double), hipMemcpyDeviceToHost, sm));
        return QC_OK;
    };
    auto ranks_agree = [&]() -> bool {
        if (!multi) return true;
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(W.h_scal);
        for (int i = 0; i < QC_SYNC_WORDS; ++i) if (w[QC_SYNC_WORDS + i] != ~w[i]) return false;
        return true;
    };
    if (multi && nspin == 1) QC_HIP_CHECK(hipMemsetAsync(W.d_sync + 2, 0, 2 * sizeof(double), sm));       // unused spin slot
    bool scale_in_kernel = false;
    // Single-rank runs on the one-workgroup path: the kernel that ends the pass stores the pass's sequence number into pinned memory
    // after the scalars and control words, and the host polls THAT instead of the event behind it (a few microseconds earlier per pass).
    unsigned *h_seq = reinterpret_cast<unsigned *>(W.h_scal + 2 * QC_SYNC_WORDS);
    // speculative build of the next pass: decided before the Roothaan step goes out (the one-workgroup kernel releases the fork word itself)
    const bool want_spec = may_continue && st->spec_on && !st->stored && qc_fock_can_speculate(S);
    // (with a speculative build behind the pass every launch sequence ends in a kernel that can store the word - the release kernel - and
    // should: the event behind a kernel that retires while the next build's side chains start is signalled up to 100 us late)
    const bool seq_wait = ((W.small_fused && !multi) || want_spec) && !st->event_wait;
    const unsigned spec_seq = want_spec ? ++S->fork_seq : 0;
    const bool release_in_kernel = want_spec && W.small_fused && !st->uhf && !multi && !st->event_wait;
    // UHF on the one-workgroup path: the two spins' kernels side by side as well - beta on a side stream of another dispatch pipe behind an
    // event of the build's closing kernel, with the second set of work buffers; the kernel that joins the streams on the device also hands
    // the control words over and stores the sequence word (qc_spin_join_end).  Same kernels per spin: bit for bit the serial order.
    const bool small_par = W.small_fused && st->uhf && st->spin_parallel && S->nlanes >= 2 && !S->join_by_events && !multi && !want_spec;
    if (small_par) {
        hipStream_t side = qc_spin_fork(S);
        if (!side) return QC_ERR_HIP;
        for (int s = 0; s < nspin; ++s) {
            SmallTail tl{st->nocc[s], 1.0, st->Dn[s].p, st->D[s].p, scal_out + 2 * s, nullptr, ctl_out, nullptr};
            if ((rc = roothaan_small(S, W, *st->diis[s], Gcur + s * nn, st->D[s].p, st->ws.p + s * n, st->Cs.p + s * nn, s, dE[s], dF[s], have_F, tl,
                                     s == 1 ? side : nullptr, s)) != QC_OK) return rc;
        }
        if ((rc = qc_spin_join_end(S, W.ctl, ctl_out, seq_wait ? h_seq : nullptr, st->pass_seq + 1)) != QC_OK) return rc;
    } else if (W.small_fused) {
        for (int s = 0; s < nspin; ++s) {
            // (RHF, direct fixed-point builds: the kernel that forms the new density also leaves the next build's fixed-point unit)
            const bool scale_here = !st->uhf && !st->stored && S->accum_fx;
            scale_in_kernel = scale_here;
            SmallTail tl{st->nocc[s], st->uhf ? 1.0 : 2.0, st->Dn[s].p, st->D[s].p, scal_out + 2 * s, s == nspin - 1 ? W.ctl : nullptr, ctl_out,
                         scale_here ? S->d_fxs : nullptr};
            // (when a release kernel follows - closed-shell UHF with a speculative build behind the pass - IT stores the sequence word:
            // the host must not look at the cancel word before the kernel that writes it has run)
            if (seq_wait && s == nspin - 1 && !(want_spec && !release_in_kernel)) { tl.seq_out = h_seq; tl.seq = st->pass_seq + 1; }
            if (release_in_kernel) { tl.fork_words = S->d_join; tl.fork_seq = spec_seq; tl.eps = st->eps_hint; tl.h_cancel = st->h_cancel; }
            if ((rc = roothaan_small(S, W, *st->diis[s], Gcur + s * nn, st->D[s].p, st->ws.p + s * n, st->Cs.p + s * nn, s, dE[s], dF[s], have_F, tl)) != QC_OK) return rc;
        }
    } else
        for (int s = 0; s < nspin; ++s) if ((rc = density_and_scalars(s, s == nspin - 1)) != QC_OK) return rc;
    if ((rc = publish_scalars()) != QC_OK) return rc;
    // the next pass's build starts from Dn: its density-only preliminaries run while the host turns around
    auto prepare_next = [&]() -> int { return st->stored ? QC_OK : qc_fock_prepare_device(S, st->Dn[0].p, st->uhf ? st->Dn[1].p : nullptr, st->uhf, st, scale_in_kernel); };
    if ((rc = prepare_next()) != QC_OK) return rc;
    // (the pass's scalars as the device sees them: pinned host memory, or the all-reduced words of a multi-rank run)
    if (want_spec && !release_in_kernel)
        qc_spec_release(sm, S->d_join, spec_seq, multi ? reinterpret_cast<const double *>(W.d_sync) : W.h_scal, n, nspin, st->eps_hint, st->h_cancel,
                        seq_wait ? h_seq : nullptr, st->pass_seq + 1);
    qc_stamp("roothaan out");
    QC_HIP_CHECK(hipEventRecord(ev2, sm));
    qc_stamp("ev2");
    if (want_spec) {
        // the next pass's build, behind everything above in the handle's stream and behind the fork word on the side streams
        hipEvent_t *const evn = st->evs[st->ev_cur ^ 1];
        double *nE[2] = {nullptr, nullptr}, *nF[2] = {nullptr, nullptr};
        for (int s = 0; s < nspin; ++s) st->diis[s]->peek_next(&nE[s], &nF[s]);
        bool f_done = false;
        scf_flush_timing(st);                       // (the other event set is about to be recorded again)
        QC_HIP_CHECK(hipEventRecord(evn[0], sm));
        rc = qc_fock_build_device(S, st->Dn[0].p, st->uhf ? st->Dn[1].p : nullptr, Gnext, st->uhf ? Gnext + nn : nullptr, st->uhf,
                                  &st->twin, W.H.p, nF[0], nF[1], &f_done, st, spec_seq);
        if (rc != QC_OK) return rc;
        QC_HIP_CHECK(hipEventRecord(evn[1], sm));
        qc_system::QcSpec &sp = S->spec;
        sp.pending = true; sp.owner = st; sp.Da = st->Dn[0].p; sp.Db = st->uhf ? st->Dn[1].p : nullptr; sp.Ga = Gnext; sp.Gb = st->uhf ? Gnext + nn : nullptr;
        sp.f_done = f_done; sp.seq = spec_seq;
    }
    const double th1 = now_ms();
    if (seq_wait) {
        const unsigned want = st->pass_seq + 1;
        unsigned spins = 0;
        double t0 = 0.0;
        while (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != want) {
            if ((++spins & 0xfff) == 0) {                   // (a failed launch or a fault never stores the word: the event knows)
                const hipError_t e = hipEventQuery(ev2);
                if (e == hipErrorNotReady) {
                    const double t = now_ms();
                    if (t0 == 0.0) t0 = t;
                    else if (t - t0 > host_wait_limit_ms()) { fprintf(stderr, "qchem_hip: an SCF pass did not finish within %.0f s\n", host_wait_limit_ms() * 1e-3); return QC_ERR_HIP; }
                    continue;
                }
                if (e != hipSuccess || __atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != want) { fprintf(stderr, "qchem_hip: the pass ended without its sequence word (%s)\n", hipGetErrorString(e)); return QC_ERR_HIP; }
            }
        }
        st->pass_seq = want;
        // The word says that the pass's last kernel is through.  If the preliminaries of the next build were put behind it (UHF: density
        // sum and fixed-point unit; a memset after a mode change), a NON-speculative next build's side streams - which start without a
        // fork event - must not overtake them: then the event behind them is waited for as well.  (RHF on this path has nothing there:
        // the kernel leaves the fixed-point unit itself and the fold left the planes clean.  A speculative build waits on the device.)
        if (!st->stored && S->prep_enqueued && !want_spec) QC_HIP_CHECK(wait_event(ev2));
    } else QC_HIP_CHECK(wait_event(ev2));
    const double th2 = now_ms();
    qc_stamp("pass seen");
    if ((rc = qc_join_check(S)) != QC_OK) return rc;                     // (the join of this pass's build is in front of everything waited for)
    if (!want_spec) qc_gate_quiet(S);                                    // (nothing of this handle waits on the device any more)
    if (!ranks_agree()) { fprintf(stderr, "qchem_hip: rank %d: the ranks' SCF scalars differ - replicated state diverged\n", S->rank); return QC_ERR_RCCL; }
    if (h_ctl[8] != 0) return QC_DIIS_SINGULAR;                          // "DIIS failed", rhf.rs:73
    if (h_ctl[9] != 0) return QC_EIG_NOT_CONVERGED;
    static const bool dbg = getenv("QC_SCF_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[scf] ctl a: %d %d %d %d  b: %d %d %d %d  npass %d %d have_prev %d mode %d cold %d spec %s/%s | host enqueue %.0f us, then waited %.0f us\n", h_ctl[0], h_ctl[1], h_ctl[2], h_ctl[3], h_ctl[4], h_ctl[5], h_ctl[6], h_ctl[7], W.npass[0], W.npass[1], (int)W.have_prev[0], W.mode[0], (int)W.cold[0], spec_hit ? "hit" : "-", want_spec ? "issued" : "-", (th1 - th0) * 1e3, (th2 - th1) * 1e3);
    if (dbg) {   // the pass's DIIS coefficients (diis.rs:50-51), newest sample first
        double c[12] = {0};
        const int m = (int)st->diis[0]->slots.size();
        (void)hipMemcpy(c, st->diis[0]->d_c, m * sizeof(double), hipMemcpyDeviceToHost);
        fprintf(stderr, "[scf] diis c:");
        for (int j = 0; j < m; ++j) fprintf(stderr, " %.3e", c[j]);
        fprintf(stderr, "\n");
    }
    // the pass's event times are read by the next pass (scf_flush_timing): the last event may still be in flight when the host has seen
    // the sequence word, and with a speculative build behind the pass nothing should keep the host here
    st->timing_pending = true; st->pending_set = st->ev_cur;
    st->pend_build_tuned = st->cur_build_tuned; st->pend_build_gen = st->cur_build_gen;
    // (what the NEXT pass's timing set will describe, if its build is the one just issued)
    st->cur_build_tuned = false; st->cur_build_gen = S->assign_gen;
    bool redo = false;
    for (int s = 0; s < nspin; ++s) {
        const bool refined = W.cold[s] || (W.have_prev[s] && W.mode[s] == 0);       // the eigensolve reported through the control word
        if (!refined) continue;
        if (h_ctl[4 * s] == 1) { W.npass[s] = W.cold[s] ? 3 : std::max(1, std::min(3, h_ctl[4 * s + 3])); continue; }
        W.npass[s] = 3;
        // the refinement wanted rotations (large step, or a degenerate cluster): repeat this spin's eigensolve the careful way
        if (!redo) {
            // (the density is about to change: a speculative build from the old one is worthless - it is left to run out, its closing
            // kernel leaves the accumulator planes clean, and everything below queues behind it on the handle's stream)
            if (S->spec.pending && S->spec.owner == st) { S->spec.pending = false; st->spec_lost += 1; }
            scf_flush_timing(st); QC_HIP_CHECK(hipEventRecord(ev1, sm));
        }
        if ((rc = roothaan_redo_eig(S, W, st->ws.p + s * n, st->Cs.p + s * nn, s)) != QC_OK) return rc;
        if ((rc = density_and_scalars(s, false)) != QC_OK) return rc;
        redo = true;
    }
    if (redo) {
        st->redos += 1;
        // the repeated eigensolves report through the same control words (Jacobi sweeps exhausted: ctl[9]): hand them over again,
        // whichever spin was repeated, and clear them for the next pass
        QC_HIP_CHECK(hipMemcpyAsync(ctl_out, W.ctl, 16 * sizeof(int), hipMemcpyDefault, sm));
        QC_HIP_CHECK(hipMemsetAsync(W.ctl, 0, 16 * sizeof(int), sm));
        if ((rc = publish_scalars()) != QC_OK) return rc;
        scale_in_kernel = false;
        if ((rc = prepare_next()) != QC_OK) return rc;                    // (the density changed)
        QC_HIP_CHECK(hipEventRecord(ev2, sm));
        QC_HIP_CHECK(wait_event(ev2));
        if ((rc = qc_join_check(S)) != QC_OK) return rc;
        qc_gate_quiet(S);
        if (!ranks_agree()) return QC_ERR_RCCL;
        if (h_ctl[9] != 0) return QC_EIG_NOT_CONVERGED;                  // the repeat ran out of sweeps: no vectors to go on with
        float ms_r = 0;
        (void)hipEventElapsedTime(&ms_r, ev1, ev2);
        st->ms_linalg += ms_r;
    }
    double rms_sum = 0.0, e_sum = 0.0;
    for (int s = 0; s < nspin; ++s) {
        const double rms_s = std::sqrt(W.h_scal[2 * s + 1] / n);
        e_sum += W.h_scal[2 * s]; rms_sum += rms_s;
        // the refinement is perturbative: on its own once the density has nearly stopped moving, behind two Jacobi sweeps
        // while it still moves, not at all in the first wild passes
        // (the one-workgroup path of small matrices pays 150 us for a cold start and nothing extra for a refinement pass that turns out to
        // be needed: it refines from the previous vectors one decade earlier - H2O/cc-pVTZ: one cold pass less per run, no repeats;
        // benzene at 1e-2: two repeated eigensolves per run, slower than 1e-3)
        const double warm_rms = st->warm_rms_env > 0.0 ? st->warm_rms_env : (W.small_fused ? 1e-2 : 1e-3);
        W.mode[s] = rms_s >= 1.0 ? 2 : (rms_s >= warm_rms || redo) ? 1 : 0;
        std::swap(st->D[s].p, st->Dn[s].p);                              // D += 1.0 * dD
        std::swap(W.CpPrev[s].p, W.CpNew[s].p);
        W.have_prev[s] = true;
    }
    st->g_cur ^= 1;
    st->passes += 1;
    qc_stamp("pass end");
    qc_stamp_flush();
    if (energy) *energy = e_sum;
    if (rms_out) *rms_out = st->uhf ? rms_sum / 2.0 : rms_sum;
    return QC_OK;
}

static int scf_run(qc_system *S, const qc_hf_config *cfg, qc_hf_output *out, bool uhf) {
    if (!S || !cfg || !out || !out->orbital_energies || (uhf && !out->orbital_energies_beta)) return QC_ERR_INVALID;
    const double t_begin = now_ms();
    qc_scf_state *st = nullptr;
    int rc = scf_begin(S, uhf, cfg->n_alpha, cfg->n_beta, &st);
    if (rc != QC_OK) return rc;
    struct Del { void operator()(qc_scf_state *p) const { scf_state_delete(p); } };
    std::unique_ptr<qc_scf_state, Del> guard(st);
    const int n = S->nbasis;
    out->nuclear_repulsion = qc_nuclear_repulsion(S);                    // rhf.rs:39
    out->electronic_energy = 0.0; out->iterations = 0;
    int status = QC_NOT_CONVERGED;
    st->eps_hint = cfg->epsilon;                                         // (the rule below, told to the device: see scf_iterate)
    for (size_t it = 0; it <= cfg->max_iterations; ++it) {               // inclusive range, rhf.rs:66 / uhf.rs:80
        double e = 0.0, rms = 0.0;
        rc = scf_iterate(st, &e, &rms, it < cfg->max_iterations);
        if (rc != QC_OK) { status = rc; break; }
        // the reference's per-iteration log line (rhf.rs:90-92, uhf.rs:138; `log::info!`, silent unless a logger is installed): QC_LOG=1
        static const bool log_info = getenv("QC_LOG") != nullptr;
        if (log_info) fprintf(stderr, "iteration %-4zu - electronic energy %1.4f. density rms %1.4e\n", it, e, rms);
        const bool conv = uhf ? (rms / 2.0 < cfg->epsilon) : (rms < cfg->epsilon);   // uhf.rs:139 / rhf.rs:94
        if (conv) {
            out->electronic_energy = e; out->iterations = it;
            QC_HIP_CHECK(hipMemcpy(out->orbital_energies, st->ws.p, n * sizeof(double), hipMemcpyDeviceToHost));
            if (uhf) QC_HIP_CHECK(hipMemcpy(out->orbital_energies_beta, st->ws.p + n, n * sizeof(double), hipMemcpyDeviceToHost));
            status = QC_OK;
            break;
        }
    }
    scf_flush_timing(st);
    out->ms_setup = st->ms_setup; out->ms_fock_total = st->ms_fock; out->ms_linalg_total = st->ms_linalg;
    out->ms_total = now_ms() - t_begin;
    out->ms_tuner = st->ms_tuner;
    return status;
}

extern "C" {

int qc_scf_begin_rhf(qc_system *S, qc_scf_state **out) { return scf_begin(S, false, 0, 0, out); }
int qc_scf_begin_uhf(qc_system *S, int n_alpha, int n_beta, qc_scf_state **out) { return scf_begin(S, true, n_alpha, n_beta, out); }
int qc_scf_iterate(qc_scf_state *st, double *electronic_energy, double *density_rms) {
    if (!st) return QC_ERR_INVALID;
    return scf_iterate(st, electronic_energy, density_rms);
}
int qc_scf_orbital_energies(qc_scf_state *st, int spin, double *out) {
    if (!st || !out || spin < 0 || spin > (st->uhf ? 1 : 0)) return QC_ERR_INVALID;
    QC_HIP_CHECK(hipMemcpy(out, st->ws.p + (size_t)spin * st->S->nbasis, st->S->nbasis * sizeof(double), hipMemcpyDeviceToHost));
    return QC_OK;
}
int qc_scf_density(qc_scf_state *st, int spin, double *out) {
    if (!st || !out || spin < 0 || spin > (st->uhf ? 1 : 0)) return QC_ERR_INVALID;
    const size_t nn = (size_t)st->S->nbasis * st->S->nbasis;
    QC_HIP_CHECK(hipMemcpy(out, st->D[spin].p, nn * sizeof(double), hipMemcpyDeviceToHost));
    return QC_OK;
}
int qc_scf_matrix(qc_scf_state *st, int which, double *out) {
    if (!st || !out || which < 0 || which > 2) return QC_ERR_INVALID;
    const size_t nn = (size_t)st->S->nbasis * st->S->nbasis;
    const double *src = which == 0 ? st->W.S.p : which == 1 ? st->W.H.p : st->W.X.p;
    QC_HIP_CHECK(hipMemcpy(out, src, nn * sizeof(double), hipMemcpyDeviceToHost));
    return QC_OK;
}
int qc_scf_spin_square(qc_scf_state *st, double *s2) {
    if (!st || !s2) return QC_ERR_INVALID;
    *s2 = 0.0;
    if (!st->uhf) return QC_OK;
    qc_system *S = st->S;
    ScfWork &W = st->W;
    const int n = S->nbasis;
    hipStream_t sm = S->stream;
    qc_gemm(sm, n, n, n, 1.0, st->D[0].p, n, false, W.S.p, n, false, 0.0, W.t1[0].p, n);      // D_alpha S
    qc_gemm(sm, n, n, n, 1.0, st->D[1].p, n, false, W.S.p, n, false, 0.0, W.t2[0].p, n);      // D_beta S
    // tr(A B) = sum_ij A_ij B_ji: one dot product of A with B^T - reuse the DIIS dot kernel on (A, B^T)
    qc_sub_transpose(sm, n, W.t2[0].p, W.t3[0].p);                                                // t3 = B - B^T
    qc_axpby(sm, n, 1.0, W.t2[0].p, -1.0, W.t3[0].p, W.t4[0].p);                                    // t4 = B^T
    const double *ys[1] = {W.t4[0].p};
    qc_dots(sm, n, W.t1[0].p, ys, 1, W.scal.p);
    double tr = 0.0;
    QC_HIP_CHECK(hipMemcpyAsync(&tr, W.scal.p, sizeof(double), hipMemcpyDeviceToHost, sm));
    QC_HIP_CHECK(hipStreamSynchronize(sm));
    const double sz = 0.5 * (st->nocc[0] - st->nocc[1]);
    *s2 = sz * (sz + 1.0) + st->nocc[1] - tr;
    return QC_OK;
}

int qc_set_accumulation(qc_system *S, int fixed_point) {
    if (!S || (fixed_point != 0 && fixed_point != 1)) return QC_ERR_INVALID;
    S->accum_fx = fixed_point;
    S->prepared = false; S->gt_clean = false;                    // (a prepared build zeroed the planes of the other mode)
    return QC_OK;
}

int qc_set_schwarz(qc_system *S, double tau) {
    if (!S || !(tau >= 0.0)) return QC_ERR_INVALID;
    S->schwarz_tau = tau;
    return qc_device_reshard(S);
}

int qc_set_fock_mode(qc_system *S, int mode) {
    if (!S || (mode != 0 && mode != 1)) return QC_ERR_INVALID;
    S->fock_mode = mode;
    return QC_OK;
}
double qc_scf_tensor_ms(qc_scf_state *st) { return st ? st->ms_tensor : 0.0; }
int qc_scf_timings(qc_scf_state *st, double *ms_setup, double *ms_fock, double *ms_linalg) {
    if (!st) return QC_ERR_INVALID;
    scf_flush_timing(st);
    if (ms_setup) *ms_setup = st->ms_setup;
    if (ms_fock) *ms_fock = st->ms_fock;
    if (ms_linalg) *ms_linalg = st->ms_linalg;
    return QC_OK;
}
void qc_scf_end(qc_scf_state *st) { scf_state_delete(st); }
// (test hook, host only: what the bra-major work lists store for one lane and what the device-record builder reads back from it)
int qc_debug_ket_entry(int ket, int first_primitive, int length, int packed, int32_t out[3]) {
    if (!out || ket < 0) return QC_ERR_INVALID;
    if (packed && (ket >= (1 << QC_KET_BITS) || first_primitive < 0 || first_primitive > 127 || length < 0 || length > 127)) return QC_ERR_INVALID;
    const int entry = packed ? (int)qc_pack_ket_entry(ket, first_primitive, length) : ket;
    int k, f, l;
    qc_unpack_ket_entry(entry, packed != 0, &k, &f, &l);
    out[0] = k; out[1] = f; out[2] = l;
    return QC_OK;
}
int qc_freeze_assignment(qc_system *S) {
    if (!S) return QC_ERR_INVALID;
    qc_assignment_freeze(S);
    return QC_OK;
}
int qc_dispatch_lanes(qc_system *S, int32_t *nlanes, int32_t slot_stream[8]) {
    if (!S || !nlanes) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    *nlanes = S->nlanes;
    if (slot_stream) { for (int k = 0; k < QC_NSTREAMS; ++k) slot_stream[k] = S->slot_side[k]; slot_stream[QC_NSTREAMS] = S->lane0_is_main ? 1 : 0; }
    return QC_OK;
}
int qc_scf_set_stop_rule(qc_scf_state *st, double epsilon) {
    if (!st || !(epsilon >= 0.0)) return QC_ERR_INVALID;
    st->eps_hint = epsilon;
    return QC_OK;
}
int qc_scf_counters(qc_scf_state *st, double *out, int n) {
    if (!st || !out || n < 0) return QC_ERR_INVALID;
    scf_flush_timing(st);
    const double v[QC_SCF_NCOUNTERS] = {st->ms_setup, st->ms_fock, st->ms_linalg, (double)st->builds_timed, st->ms_tuner, (double)st->passes,
                                        (double)st->spec_hits, (double)st->spec_lost, (double)st->redos,
                                        (double)st->S->on.trials, st->S->on.settled ? 1.0 : 0.0};
    for (int i = 0; i < n && i < QC_SCF_NCOUNTERS; ++i) out[i] = v[i];
    return QC_OK;
}

// sorted_eigs with a starting guess: V0 = eigenvectors of a nearby matrix (what the SCF loop uses from its second pass on)
int qc_sym_eig_warm(qc_system *S, int n, const double *A, const double *V0, double *V, double *w) {
    if (!S || n <= 0 || !A || !V0 || !V || !w) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)n * n;
    DevBuf dA, dV0, dV, dw, wk, t1, t2, t3, t4, sm;
    DevBuf *all[] = {&dA, &dV0, &dV, &wk, &t1, &t2, &t3, &t4};
    for (auto b : all) if (b->alloc(nn) != QC_OK) return QC_ERR_HIP;
    if (dw.alloc(n) != QC_OK || sm.alloc(qc_eig_small_doubles(n)) != QC_OK) return QC_ERR_HIP;
    QC_HIP_CHECK(hipMemcpyAsync(dA.p, A, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(dV0.p, V0, nn * sizeof(double), hipMemcpyHostToDevice, S->stream));
    int flag = 0;
    QC_HIP_CHECK(hipMemsetAsync(S->d_flag, 0, sizeof(int), S->stream));
    rc = qc_eig_device_refine(S->stream, n, dA.p, dV0.p, dV.p, dw.p, wk.p, t1.p, t2.p, t3.p, t4.p, sm.p, S->d_flag);
    if (rc != QC_OK) return rc;
    QC_HIP_CHECK(hipMemcpyAsync(V, dV.p, nn * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(w, dw.p, n * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipMemcpyAsync(&flag, S->d_flag, sizeof(int), hipMemcpyDeviceToHost, S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    return flag ? QC_EIG_NOT_CONVERGED : QC_OK;
}

// restricted_hartree_fock (rhf.rs:32-108) / unrestricted_hartree_fock (uhf.rs:36-167)
int qc_scf_rhf(qc_system *S, const qc_hf_config *cfg, qc_hf_output *out) { return scf_run(S, cfg, out, false); }
int qc_scf_uhf(qc_system *S, const qc_hf_config *cfg, qc_hf_output *out) { return scf_run(S, cfg, out, true); }

// ---- multi-GPU
int qc_comm_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId u;
    if (!qc_rccl().ok || qc_rccl().GetUniqueId(&u) != ncclSuccess) return QC_ERR_RCCL;
    std::memcpy(id, &u, 128);
    return QC_OK;
}

int qc_set_shard(qc_system *S, int rank, int nranks) {
    if (!S || nranks <= 0 || rank < 0 || rank >= nranks) return QC_ERR_INVALID;
    S->rank = rank; S->nranks = nranks;
    return qc_device_reshard(S);
}

int qc_comm_init(qc_system *S, const uint8_t id[128], int rank, int nranks) {
    if (!S || !id) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    if ((rc = qc_set_shard(S, rank, nranks)) != QC_OK) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    ncclComm_t comm;
    if (!qc_rccl().ok) return QC_ERR_RCCL;
    if (S->comm) { qc_rccl().CommDestroy((ncclComm_t)S->comm); S->comm = nullptr; }      // a second init replaces the communicator
    if (qc_rccl().CommInitRank(&comm, nranks, u, rank) != ncclSuccess) return QC_ERR_RCCL;
    S->comm = comm;
    return QC_OK;
}

int qc_rccl_info(char *buf, size_t len) {
    if (!buf || len == 0) return QC_ERR_INVALID;
    QcRccl &r = qc_rccl();
    int v = 0;
    if (r.ok && r.GetVersion) (void)r.GetVersion(&v);
    snprintf(buf, len, "%s version %d", r.ok ? r.path.c_str() : "unavailable", v);
    return r.ok ? QC_OK : QC_ERR_RCCL;
}

int qc_plan_shard(qc_system *S, int rank, int nranks, int64_t *nquartets, double *flops) {
    if (!S || nranks <= 0 || rank < 0 || rank >= nranks) return QC_ERR_INVALID;
    const int r0 = S->rank, n0 = S->nranks;
    S->rank = rank; S->nranks = nranks;
    qc_build_shards(S);
    int64_t nq = 0; double fl = 0.0;
    for (const auto &c : S->classes) { nq += (int64_t)c.shard.size(); fl += c.flops_alg; }
    if (nquartets) *nquartets = nq;
    if (flops) *flops = fl;
    S->rank = r0; S->nranks = n0;
    qc_build_shards(S);
    return QC_OK;
}

// list the quartets of a shard as (shell A, B, C, D) so a host-side checker can digest exactly the same units
int qc_plan_shard_quartets(qc_system *S, int rank, int nranks, int32_t *abcd /* 4 * nquartets or NULL */, int64_t capacity) {
    if (!S || nranks <= 0 || rank < 0 || rank >= nranks) return QC_ERR_INVALID;
    const int r0 = S->rank, n0 = S->nranks;
    S->rank = rank; S->nranks = nranks;
    qc_build_shards(S);                                   // exactly the lists the device would be given
    int64_t k = 0;
    bool overflow = false;
    for (const auto &c : S->classes)
        for (const auto &t : c.shard) {
            if (abcd) {
                if (k >= capacity) { overflow = true; break; }
                abcd[4 * k + 0] = S->pairA[t.bra]; abcd[4 * k + 1] = S->pairB[t.bra];
                abcd[4 * k + 2] = S->pairA[t.ket]; abcd[4 * k + 3] = S->pairB[t.ket];
            }
            ++k;
        }
    S->rank = r0; S->nranks = n0;
    qc_build_shards(S);
    if (overflow) return QC_ERR_INVALID;
    return (int)k;
}

int qc_work_stats_get(qc_system *S, qc_work_stats *out) {
    if (!S || !out) return QC_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    qc_ensure_lists(S);
    for (const auto &c : S->classes) {
        if (c.shard.empty()) continue;
        out->quartets += (int64_t)c.shard.size(); out->prim_quartets += c.prim_quartets;
        out->bytes_alg += c.bytes_alg; out->flops_alg += c.flops_alg; out->nclasses += 1;
    }
    out->quartets_enumerated = S->nquartets;
    out->quartets_screened_out = S->nscreened;
    out->schwarz_tau = S->pairQ.empty() ? 0.0 : S->schwarz_tau;
    return QC_OK;
}

int qc_fock_profile(qc_system *S, const double *dD, double *dG, int reps, float *class_ms, int32_t *class_id, int64_t *class_quartets,
                    double *class_bytes, double *class_flops, float *total_ms) {
    if (!S || !dD || !dG || reps <= 0) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const int n = S->nbasis;
    const size_t nn = (size_t)n * n;
    std::vector<float> acc(S->classes.size(), 0.f), one(S->classes.size(), 0.f);
    float tot = 0.f;
    S->prepared = false; S->gt_clean = false;
    if (S->accum_fx) qc_fx_scale(S->stream, n, dD, nullptr, S->imax, S->d_fxs);
    Event ev0, ev1;
    if (ev0.create() != QC_OK || ev1.create() != QC_OK) return QC_ERR_HIP;
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    for (int r = 0; r < reps; ++r) {
        QC_HIP_CHECK(hipMemsetAsync(S->d_Gtmp, 0, (size_t)2 * QC_NREP * nn * sizeof(double), S->stream));
        QcFockArgs a{};
        a.nrep = QC_NREP; a.rep_stride = nn; a.fxs = S->accum_fx ? S->d_fxs : nullptr; a.fx_lo = (size_t)QC_NREP * nn;
        a.Dj = dD; a.Dk0 = dD; a.Dk1 = nullptr; a.cK = 0.5; a.G0 = S->d_Gtmp; a.G1 = S->d_Gtmp + nn;
        if ((rc = qc_launch_fock_classes(S, a, one.data())) != QC_OK) return rc;
        for (size_t i = 0; i < acc.size(); ++i) acc[i] += one[i];
        // un-instrumented whole build for the total
        QC_HIP_CHECK(hipEventRecord(e0, S->stream));
        if ((rc = qc_fock_build_device(S, dD, nullptr, dG, nullptr, false)) != QC_OK) return rc;
        QC_HIP_CHECK(hipEventRecord(e1, S->stream));
        QC_HIP_CHECK(hipEventSynchronize(e1));
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    int k = 0;
    for (size_t i = 0; i < S->classes.size(); ++i) {
        const QcClass &c = S->classes[i];
        if (c.shard.empty()) continue;
        if (class_ms) class_ms[k] = acc[i] / reps;
        if (class_id) class_id[k] = (c.bm ? 1 << 12 : 0) | (c.LAB << 8) | (c.LCD << 4) | c.LGC;
        if (class_quartets) class_quartets[k] = (int64_t)c.shard.size();
        if (class_bytes) class_bytes[k] = c.bytes_alg;
        if (class_flops) class_flops[k] = c.flops_alg;
        ++k;
    }
    if (total_ms) *total_ms = tot / reps;
    return QC_OK;
}

int qc_unit_quartets(qc_system *S, int64_t *unit_quartets) {
    if (!S || !unit_quartets) return QC_ERR_INVALID;
    for (int u = 0; u < QC_NUNITS; ++u) unit_quartets[u] = 0;
    qc_ensure_lists(S);
    for (const auto &c : S->classes) unit_quartets[qc_build_unit_of(S, c.LAB, c.LCD, c.bm)] += (int64_t)c.shard.size();
    return QC_OK;
}

// Per launch unit ("tier" = (LAB, LCD <= 3 | LCD >= 4), the kernels an un-instrumented build really launches), timed
// serially with hipEvents on the handle's stream.  Arrays have 14 entries, unit u = 2 * LAB + tier; empty units are 0.
int qc_fock_profile_tiers(qc_system *S, const double *dD, double *dG, int reps, float *unit_ms, int64_t *unit_quartets,
                          double *unit_bytes, double *unit_flops, float *total_ms) {
    if (!S || !dD || !dG || reps <= 0 || !unit_ms) return QC_ERR_INVALID;
    int rc = qc_device_init(S);
    if (rc != QC_OK) return rc;
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    const int NU = QC_NUNITS;
    std::vector<float> acc(NU, 0.f), one(NU, 0.f);
    float tot = 0.f;
    S->prepared = false; S->gt_clean = false;
    if (S->accum_fx) qc_fx_scale(S->stream, S->nbasis, dD, nullptr, S->imax, S->d_fxs);
    Event ev0, ev1;
    if (ev0.create() != QC_OK || ev1.create() != QC_OK) return QC_ERR_HIP;
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    for (int r = 0; r < reps; ++r) {
        QC_HIP_CHECK(hipMemsetAsync(S->d_Gtmp, 0, (size_t)2 * QC_NREP * nn * sizeof(double), S->stream));
        QcFockArgs a{};
        a.nrep = QC_NREP; a.rep_stride = nn; a.fxs = S->accum_fx ? S->d_fxs : nullptr; a.fx_lo = (size_t)QC_NREP * nn;
        a.Dj = dD; a.Dk0 = dD; a.Dk1 = nullptr; a.cK = 0.5; a.G0 = S->d_Gtmp; a.G1 = S->d_Gtmp + nn;
        if ((rc = qc_launch_fock_classes(S, a, nullptr, one.data())) != QC_OK) return rc;
        for (int i = 0; i < NU; ++i) acc[i] += one[i];
        QC_HIP_CHECK(hipEventRecord(e0, S->stream));
        if ((rc = qc_fock_build_device(S, dD, nullptr, dG, nullptr, false)) != QC_OK) return rc;
        QC_HIP_CHECK(hipEventRecord(e1, S->stream));
        QC_HIP_CHECK(hipEventSynchronize(e1));
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    for (int u = 0; u < NU; ++u) {
        unit_ms[u] = acc[u] / reps;
        if (unit_quartets) unit_quartets[u] = 0;
        if (unit_bytes) unit_bytes[u] = 0;
        if (unit_flops) unit_flops[u] = 0;
    }
    for (const auto &c : S->classes) {
        if (c.shard.empty()) continue;
        const int u = qc_build_unit_of(S, c.LAB, c.LCD, c.bm);
        if (unit_quartets) unit_quartets[u] += (int64_t)c.shard.size();
        if (unit_bytes) unit_bytes[u] += c.bytes_alg;
        if (unit_flops) unit_flops[u] += c.flops_alg;
    }
    if (total_ms) *total_ms = tot / reps;
    return QC_OK;
}

}  // extern "C"
