// qc_internal.h - shared declarations of libqchem_hip.so (host model, device buffers, kernel launchers).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/qchem_hip.h"

constexpr int QC_LMAX = 3;              // highest shell angular momentum with kernels (f)
constexpr int QC_LPAIR = 2 * QC_LMAX;   // highest pair angular momentum
constexpr int QC_LTOT = 4 * QC_LMAX;    // highest Hermite order of an ERI
constexpr int QC_SLOT_ITMAX = 128;
constexpr int QC_LREG = 4;               // total Hermite orders up to this keep the R table in registers (bra-major kernels; Boys branch)
constexpr int QC_LHOIST_DPP = 6;         // column kernels: up to this the lanes of a group evaluate several primitive quartets' tables at once
// ... this many per chunk: the parked tables limit the resident waves, and with four groups per wave (16-lane groups) at L = 5
// ((pp|dp), the largest such class) four tables beat eight by 13 %; elsewhere eight win by 3 %
__host__ __device__ constexpr int qc_hoist_chunk(int L, int lgc) { return (L == 5 && lgc == 4) ? 4 : 8; }
__host__ __device__ constexpr bool qc_hoisted(int L) { return L <= QC_LHOIST_DPP; }
// doubles at the head of a lane group's LDS region: the hoisted register tables of one chunk + their (pref, ij/kl)
// records, or the cooperative R work array
__host__ __device__ constexpr int qc_region0(int L, int lgc) {
    return qc_hoisted(L) ? qc_hoist_chunk(L, lgc) * ((((L + 1) * (L + 2) * (L + 3) / 6) | 1) + 2) : (L + 1) * (L + 2) * (L + 3) * (L + 4) / 24;
}
      // primitive quartets per slot
constexpr double QC_PRIM_CUTOFF = 1e-17; // primitive pairs whose Hermite expansion block is entirely below this are dropped
constexpr int QC_NREP = 32;             // replicas of the Fock accumulation buffer
constexpr int QC_FX_MAXBITS = 52;       // fixed-point accumulation: finest unit 2^-52 Eh; the scale of a build follows its density (qc_fx_scale)
constexpr double QC_SCHWARZ_TAU = 1e-12; // quartets with sqrt((ab|ab) (cd|cd)) below this are not evaluated (default)
constexpr int QC_LDS_MAX = 160 * 1024;  // LDS of a gfx950 CU, the cap the kernels with dynamic LDS are allowed
#ifndef QC_NSTREAMS_N
#define QC_NSTREAMS_N 7
#endif
constexpr int QC_NSTREAMS = QC_NSTREAMS_N;  // class kernels of one build run concurrently on this many streams (compile-time: A/B builds)
constexpr int QC_NUNITS = 2 * (QC_LPAIR + 1) + 6;   // launch units of one build: (LAB, tier) of the column kernels + the bra-major launches (ket ss / ps / pp x bra range)

__host__ __device__ constexpr int qc_nherm(int L) { return (L + 1) * (L + 2) * (L + 3) / 6; }
// Classes whose two Hermite contractions run as f64 MFMA tiles (one slot per wave): high-order kets against bras with
// enough Hermite functions to fill 16-row tiles.  Measured on H2O/cc-pVTZ and benzene/cc-pVDZ: below these bounds the
// padded tiles and the lost second slot per wave cost more than the LDS reads they save.
// (round 4: d.d / f.p kets - LCD = 4 - too, in their 64-lane instances: a class whose list cannot fill the chip is a latency chain of single
// waves, and there the tile form with one wave per 16-column tile is shorter - (dd|dd) of H2O/cc-pVTZ 57 us per workgroup in the VALU
// form; lists that do fill the chip keep the 32-lane VALU instances with two slots per wave, qc_build_model)
__host__ __device__ constexpr bool qc_use_mfma(int LAB, int LCD) { return LCD >= 4 && LAB >= 3; }
// (... the classes that are ALWAYS on the matrix cores)
__host__ __device__ constexpr bool qc_mfma_always(int LAB, int LCD) { return LCD >= 5 && LAB >= 3; }
// Gather records of the matrix-core classes' step 2 (A operand = R values at h1 + h2): per class (LAB 3..6, LCD 4..6) one 16-byte record
// per (k-step, lane); classes concatenated in (LAB, LCD) order, class (LAB, LCD) starts at record qc_gidx_off(LAB, LCD)
__host__ __device__ constexpr int qc_gidx_ksteps(int LCD) { return ((LCD + 1) * (LCD + 2) * (LCD + 3) / 6 + 3) / 4; }
__host__ __device__ constexpr int qc_gidx_off(int LAB, int LCD) {
    int o = 0;
    for (int a = 3; a <= 6; ++a)
        for (int c = 4; c <= 6; ++c) { if (a == LAB && c == LCD) return o; o += 64 * qc_gidx_ksteps(c); }
    return o;
}
__host__ __device__ constexpr int qc_ncart(int L) { return (L + 1) * (L + 2) / 2; }
// Hermite index of (t,u,v): grouped by total order N = t+u+v, then t descending, then u descending.
__host__ __device__ constexpr int qc_hidx(int t, int u, int v) {
    return (t + u + v) * (t + u + v + 1) * (t + u + v + 2) / 6 + (u + v) * (u + v + 1) / 2 + v;
}
// size of the Hermite-Coulomb work array sum_{n=0..L} nherm(L-n) = C(L+4,4)
// doubles per primitive-pair block: header [p,Px,Py,Pz] + E (nherm x nab), padded to a multiple of 4 (32-byte rows)
__host__ __device__ constexpr int qc_pair_stride(int L, int nab) { return (4 + qc_nherm(L) * nab + 3) & ~3; }
__host__ __device__ constexpr int qc_rwork(int L) { return (L + 1) * (L + 2) * (L + 3) * (L + 4) / 24; }
// Recurrence plan of the cooperative Hermite-Coulomb table (qc_build_r): one 8-byte record per entry (n; t,u,v) with t+u+v >= 1,
// stage by stage; plans of all orders L are concatenated, L's starts at qc_plan_off(L)
__host__ __device__ constexpr int qc_nplan(int L) { return qc_rwork(L) - (L + 1); }
__host__ __device__ constexpr int qc_plan_off(int L) { int o = 0; for (int l = 0; l < L; ++l) o += qc_nplan(l); return o; }

struct QcShell {
    int atom, L, pure, nprim, ncart, nfunc, off;
    double A[3];
    std::vector<double> exps, coefs;  // coefs include the (L,0,0) primitive norm
    std::vector<double> T;            // nfunc x ncart, rows scaled to unit self-overlap
};

// Device-visible pair descriptor (10 ints)
struct QcPairDesc {
    int doff;     // offset (in doubles) of this pair's primitive blocks in the pair-data array
    int K;        // primitive pairs
    int na, nb;   // basis functions of shell A / B
    int offa, offb;
    int L;        // la + lb
    int shA_eq_shB;
    int psoff;    // ps and pp pairs: offset (doubles) of the packed primitive records in the pspack array, else -1
    int psperm;   // ps pairs: basis function of Cartesian axis a = (psperm >> 2a) & 3; pp pairs: the same for shell A, and for shell B from bit 6
};

struct QcTask { int bra, ket; };  // pair indices; (bra|ket) is one unique shell quartet
struct QcSlot { int bra, ket, lo, hi, c0, c1; };  // a quartet restricted to primitive quartets [lo, hi) and ket columns [c0, c1): the kernels' work unit
// Work unit of the bra-major kernels (narrow kets, qc_fock_bm.hip): one wave = one bra pair restricted to the bra
// primitive pairs [ij_lo, ij_hi), against up to 64 ket pairs (one per lane) ketlist[first .. first + nket).
struct QcBundle { int bra, ij_lo, ij_hi, first, nket, maxK, pad0, pad1; };
// Device forms of the bra-major work lists (qc_bm_device_lists): the bundle with what the kernel needs of its bra pair (so that it is
// one scalar load), the lane's unit with what it needs of the ket pair (no look-up in the pair table between the list and the
// primitive records).  nanb = na | nb << 8 | shA_eq_shB << 16;  cd = offa | offb << 16 of the ket pair;
// info = primitives of the unit | (nb == 1) << 16 | shA_eq_shB << 17 | psperm << 18;  koff = offset (doubles) of the unit's first
// primitive record in the pair data (ss kets) or in pspack (ps kets).
// LDS doubles of one wave of a bra-major launch: its block I[nab * ncd][65] and, for the high bras (LAB >= 3) against ss kets, the
// staged expansion block E[nab][HAB] of the current bra primitive pair (step 3 of qc_bm_pass)
inline int qc_bm_wave_words(int LAB, int nab, int ncd) { return nab * ncd * 65 + (LAB >= 3 && ncd == 1 ? ((nab * qc_nherm(LAB) + 1) & ~1) : 0); }
struct alignas(16) QcBundleDev { int bra, ij_lo, ij_hi, first, nket, maxK, bdoff, offa, offb, nanb, pad0, pad1; };
struct alignas(16) QcKetUnit { int ket, koff, cd, info; };

struct QcClass {
    int LAB, LCD, LGC;            // Hermite orders of bra / ket pairs; log2 of the lane-group width C
    std::vector<QcTask> tasks;    // all unique quartets of this launch bucket (full list)
    std::vector<QcTask> shard;    // the ones this rank digests
    std::vector<QcSlot> slots;    // `shard` cut into primitive-quartet ranges of at most QC_SLOT_ITMAX, longest first
    QcSlot *d_slots = nullptr;    // device copy of `slots`
    int slot_words = 0;           // LDS doubles per lane group
    int lds_bytes = 0;
    int col_slot_words = 0, col_lds_bytes = 0, col_lgc = 0;   // the same for the column kernels (a pp-ket bra-major class goes through them in the set-up passes)
    // bra-run mode of the low-L column classes (qc_fock_body): slots grouped by bra, batches of G padded with null slots
    int run = 0;                  // batches per workgroup (0: independent slots)
    int rb_rows = 0;              // most bra functions (na + nb) of the class: rows of the LDS row buffer
    // bra-major classes (ket = ss or ps pair, LAB + LCD <= QC_LREG): `bundles` replace `slots`
    bool bm = false;
    int bm_rows = 0;              // most bra functions (na + nb) of a bundle: rows of the kernels' exchange buffer
    std::vector<QcBundle> bundles;
    std::vector<int> ketlist;
    bool ket_packed = false;      // ketlist entries are packed (pair, first primitive, length) chunks, not plain pair indices (qc_make_bundles)
    QcBundleDev *d_bundles = nullptr;
    QcKetUnit *d_ketlist = nullptr;
    // work model of `shard`
    int64_t prim_quartets = 0;
    double bytes_alg = 0, flops_alg = 0;
};

struct qc_system {
    int natoms = 0, nshells = 0, nbasis = 0, nelec = 0;
    std::vector<int> Z;
    std::vector<double> xyz;
    std::vector<QcShell> shells;
    // pairs
    std::vector<QcPairDesc> pairs;
    std::vector<int> pairA, pairB, pairKfull;   // shells of each stored pair; its primitive-pair count before the cut-off
    std::vector<double> pairdata;
    std::vector<double> pspack;                 // ps pairs, 8 doubles per primitive: [q, Q(3), E_0[x,y,z], E_1] (see qc_fock_bm.hip)
    std::vector<double> pairdataT;              // same blocks with the expansion stored [ab][h] (bra side of the bra-major kernels)
    std::vector<QcClass> classes;
    std::vector<double> pairQ;                  // Schwarz factor sqrt(max_ab (ab|ab)) of each stored pair (empty until the device pass has run)
    double imax = 0.0;                          // max pairQ^2: bound on every |(ij|kl)|
    double schwarz_tau = QC_SCHWARZ_TAU;        // 0: no screening
    bool ds_order_ok = true;                    // the DS unit served the lanes of one f64 add in a reproducible order (qc_ds_order_probe, per device)
    bool lists_stale = false;                   // the work lists have not been built since the classes were (qc_build_shards(S, true))
    int64_t nscreened = 0;                      // quartets (of the whole list, all ranks) dropped by the Schwarz bound
    int64_t nquartets = 0;
    int rank = 0, nranks = 1;
    // device
    bool device_ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side[QC_NSTREAMS] = {};
    // Dispatch lanes (qc_lane_probe): the chip dispatches at most FOUR kernels at a time - hardware queues sit in pairs on four pipes, and a
    // pipe works on one dispatch until all its workgroups are launched.  slot_side[k] = side stream behind assignment slot k; the first
    // `nlanes` slots are on distinct pipes, slot 0 on the pipe of the handle's own stream (when one of the side streams shares it).
    int slot_side[QC_NSTREAMS] = {0, 1, 2, 3, 4, 5, 6};
    int nlanes = QC_NSTREAMS;
    bool lane0_is_main = false;
    bool lanes_probed = false;               // slot_side / nlanes were measured on this handle's own stream set (qc_lane_probe)
    hipEvent_t ev_fork = nullptr, ev_join[QC_NSTREAMS] = {};
    unsigned spin_target = 0;                // join counter of the spin-parallel Roothaan steps (d_join[4])
    double *d_pairdata = nullptr, *d_pairdataT = nullptr, *d_pspack = nullptr;
    void *d_shells = nullptr;                // shells / primitives / transforms / nuclei for the one-electron kernels (one blob)
    size_t shell_blob_off[5] = {};
    QcPairDesc *d_pairs = nullptr;
    double *d_boys = nullptr;
    int *d_rplan = nullptr;
    unsigned *d_gidx = nullptr;   // gather records of the matrix-core classes (qc_build_gidx)       // recurrence plans of the Hermite-Coulomb tables (qc_build_rplan)
    double *d_D = nullptr, *d_G = nullptr;   // 2 * n*n each (alpha/beta or Dj/Dk)
    double *d_Gtmp = nullptr;                // accumulation target: [plane (hi, lo)][replica][spin][n*n]
    double *d_Gred = nullptr;                // replicas folded: [plane][spin][n*n]
    double *d_Dj = nullptr;
    double *d_fxs = nullptr;                 // [2^S, 2^-S]: fixed-point scale of the current build
    int *d_flag = nullptr;
    // QC_DEV_TIMELINE: ring of QC_TL_PASSES sets of QC_TL_SLOTS kernel slots (QC_TL_W clock words each, qc_tl_stamp), one set per SCF pass; tl_cur = the set
    // of the pass being enqueued (null: off, or the ring is full); dumped when an SCF state of the handle ends (qc_tl_dump)
    unsigned char *d_lists = nullptr;        // the work lists of all classes (QcClass::d_slots / d_bundles / d_ketlist point into it)
    unsigned long long *d_tl = nullptr, *tl_cur = nullptr;
    int tl_pass = 0;
    unsigned *d_join = nullptr;              // [0] counter of the device-side join of a build's side streams (qc_join_mark / qc_join_wait);
                                             // [1] fork word: number of the last pass whose densities are final (device-side fork of a speculative build);
                                             // [2] number of the speculative build that was cancelled on the device (its class kernels return at once)
    int *h_join_timeout = nullptr;           // pinned: set by a device-side wait that gave up (the launches it waited for never finished)
    unsigned join_target = 0;
    bool fold_join_pending = false;          // the last build left its join to the closing fold (QcFockArgs::fold_joins): join_target is what it waits for
    unsigned fork_seq = 0;                   // last value promised to the fork word: the speculative build of pass k + 1 waits for fork_seq = k's number
    long long wait_limit = 0;                // device-side waits give up after this many ticks of the 100 MHz clock (qc_wait_limit)
    std::atomic<bool> waits_in_flight{false}; // device-side waits were issued and the host has not seen the handle's stream drained since (qc_gate)
    // speculative build (scf_iterate): the NEXT pass's Fock build, issued behind this pass's Roothaan step before the host has seen the pass end
    struct QcSpec {
        bool pending = false;                // issued and not yet consumed / discarded
        const void *owner = nullptr;         // the qc_scf_state it belongs to
        const double *Da = nullptr, *Db = nullptr;   // densities it digests
        double *Ga = nullptr, *Gb = nullptr; // where its closing kernel leaves G
        bool f_done = false;                 // ... and F = H + G (the next pass's DIIS slots)
        unsigned seq = 0;                    // its number (fork word value it waited for; cancel word value that empties it)
    } spec;
    int issue_threads = -1;                  // helper threads that issue a build's launches: -1 = by the size of the build (qc_fock.hip)
    bool prep_enqueued = false;              // qc_fock_prepare_device put work on the handle's stream (see scf_iterate)
    bool join_by_events = false;             // dispatches are serialised here (qc_join_probe): the side streams are joined through events
    struct QcLaunchPlan *launch_plan = nullptr; // launch units and their segments of the current work lists (qc_fock.hip)
    struct QcIssuePool *issue_pool = nullptr; // helper threads that issue a build's launches next to the caller (qc_fock.hip)
    void *comm = nullptr;                    // ncclComm_t
    std::vector<float> unit_ms;              // measured serial time of each launch unit (autotuned once per shard)
    std::vector<int> unit_stream;            // side stream of each launch unit (longest-processing-time assignment)
    std::vector<float> unit_weight;          // durations that order the launches (measured inside concurrent builds)
    // refinement of the stream assignment in instalments paid for by use (qc_fock.hip, "Refinement of the stream assignment")
    struct QcOnline {
        bool frozen = false;
        std::vector<int> best, trial;
        std::vector<std::vector<int>> nb, tried;   // neighbourhood of `best` (swept in order from nb_pos) / every assignment measured so far
        size_t nb_pos = 0;
        std::vector<std::vector<int>> cands;   // proposals of the first build (longest-first on in-build durations): tried before random neighbours
        double base_ms = 0.0;                  // build time of `best` as the search measured it
        bool seeded = false;                   // the proposals of the instrumented builds have been made (first instalment of the search)
        int trials = 0, rejects = 0, kicks = 0;   // (kicks: restarts of the local search from a perturbed copy of the best assignment known)
        long builds = 0, spent = 0;            // builds asked of this handle / extra builds the search has run
        unsigned rng = 2463534242u;
        double seen_sum = 0.0; long seen_n = 0;   // build times reported by SCF passes under the current assignment
        // finals (qc_fock_feedback): the three fastest assignments of the search, sampled inside SCF passes - a build that follows a Roothaan
        // step and a host turn-around is not the back-to-back build the search times, and which of them is fastest THERE differs
        std::vector<std::pair<float, std::vector<int>>> top;
        std::vector<double> fin_sum; std::vector<int> fin_n;
        int fin_cur = 0;
        bool settled = false;                     // search and finals are over
    } on;
    bool cand_skip = false;                  // the next build is the first under a new assignment: not a sample
    unsigned assign_gen = 0;                 // counts the changes of the stream assignment: a build's time is a sample of the assignment it ran under
    int tune_count = 0;                      // first builds (launches timed alone) on this shard layout
    // accumulators zeroed, fixed-point scale (and the UHF density sum) already enqueued for a build from exactly these densities, and the host
    // has waited for the handle's stream since (qc_fock_prepare_device): the build then starts its side streams without a fork
    bool prepared = false;
    bool gt_clean = false;                   // the accumulator planes (d_Gtmp) are known to be zero for the layout `gt_clean_nspin` (left so by the closing fold)
    int gt_clean_nspin = 0;
    const double *prep_Da = nullptr, *prep_Db = nullptr;
    const void *prep_owner = nullptr;        // the qc_scf_state that enqueued them (addresses alone could be recycled by a later state)
    int live_states = 0;                     // qc_scf_state objects that still point at this handle
    bool zombie = false;                     // qc_system_destroy was called while states were alive: the last qc_scf_end frees the handle
    int fock_mode = 0;                       // 0 direct (default), 1 stored tensor (the reference's own algorithm)
    int accum_fx = 1;                        // 1 (default): fixed-point, order-independent accumulation of G; 0: f64 atomics
    bool merge_bm = false;                    // ... and the ss-ket / high-bra bundles ride in the launch of the ps kets / low bras
    bool merge_t1 = false;                    // the wide-ket launches of the bra classes 0, 1, 2 are one launch (set with the class lists)
    bool has_fkets = false;                   // some class has an f.d / f.f ket (a basis with f functions): the wide-ket launches are the f-capable kernels
    bool pp_ok = true;                        // every p.p pair's expansion blocks have the packed form the pp-ket bra-major kernel assumes
    std::string last_error;
};

// ---- host model (qc_system.cpp)
void qc_build_model(qc_system *S);
// (meta_only: the classes' LDS / slot sizes only, from the whole task lists - what the Schwarz pass of the device set-up needs before the
// screened lists can exist; the lists are then built once, behind that pass, or on demand: qc_ensure_lists)
void qc_build_shards(qc_system *S, bool meta_only = false);
inline void qc_ensure_lists(qc_system *S);
void qc_host_one_electron(const qc_system *S, int which, double *out);
void qc_boys_host(int nmax, double x, double *F);

// Boys tables, one per total Hermite order L: row k = F_{L+j}(x_k) / j!, j = 0..7, x_k = k * QC_BOYS_DX (64-byte rows);
// behind them exp(-x_k)
constexpr double QC_BOYS_DX = 0.1;
constexpr int QC_BOYS_NGRID = 421;             // x up to 42
constexpr double QC_BOYS_XMAX = 41.9;          // beyond: asymptotic + upward recursion
constexpr int QC_BOYS_NORD = QC_LTOT + 8;      // orders kept per grid point

// ---- device side (qc_fock.hip / qc_linalg.hip)
int qc_device_init(qc_system *S);
void qc_device_free(qc_system *S);
// digestion modes
struct QcFockArgs {
    const double *Dj;     // density contracted into J (n*n)
    const double *Dk0;    // density contracted into K for spin 0
    const double *Dk1;    // spin 1 (UHF) or nullptr
    double *G0, *G1;      // accumulation targets (pre-zeroed), unsymmetrised
    double cK;            // K prefactor (0.5 RHF, 1.0 UHF)
    double *eri_out;      // if non-null: store the integrals into the n^4 tensor instead of digesting
    int nrep;             // accumulation replicas behind G0/G1
    size_t rep_stride;    // doubles between replicas
    const double *fxs;    // non-null: G0 / G1 accumulate 64-bit fixed-point integers (hi plane), scale 2^S at fxs[0], 2^-S at fxs[1] (device)
    size_t fx_lo;         // doubles from the hi plane to the lo plane
    double *schwarz_out;  // non-null: Schwarz pass over the (P|P) quartets, sqrt(max |(ab|cd)|) per pair (device), no digestion
    // device-side fork (speculative build of the next SCF pass): side streams start with a one-lane kernel that waits for the fork word to
    // reach fork_seq; the class kernels return at once when the cancel word equals it
    unsigned fork_seq = 0;   // 0: no device fork
    // the caller's next kernel on the handle's stream - the closing fold - waits for the join counter itself (qc_system::fold_join_pending
    // says that it must): no one-lane waiting kernel in front of it, one dependent launch less at the end of the build
    bool fold_joins = false;
};
int qc_launch_eri_full(qc_system *S, double *d_out);
int qc_schwarz_device(qc_system *S);     // fills pairQ / imax from the (P|P) quartets, then screens the work lists
// fixed-point scale of a build from its densities: out[0] = 2^S, out[1] = 2^-S, S = min(QC_FX_MAXBITS, 60 - ceil(log2(4 imax sum|D|)))
void qc_fx_scale(hipStream_t st, int n, const double *Da, const double *Db /*nullable*/, double imax, double *out);
int qc_one_electron_device(qc_system *S, int which /* 0 S, 1 T, 2 V */, double *d_out);
int qc_launch_fock_classes(qc_system *S, const QcFockArgs &a, float *class_ms /*nullable*/, float *unit_ms = nullptr /*nullable, 14*/, bool nofork = false);
// hipEvent time of a build inside an SCF pass that ran under stream assignment `gen` (no-op once the choice is made)
void qc_fock_feedback(qc_system *S, float build_ms, unsigned gen);
void qc_assignment_freeze(qc_system *S);
bool qc_fock_can_speculate(const qc_system *S);            // the next build may be issued with a device-side fork (tuned, device join, fixed point)
void qc_spec_release(hipStream_t st, unsigned *words, unsigned seq, const double *scal, int n, int nspin, double eps, unsigned *h_cancel,
                     unsigned *h_seq, unsigned seqval);
// the beta step of a UHF pass on a side stream (another dispatch pipe than the handle's): fork = that stream, made to wait for what the
// handle's stream holds so far; join = the handle's stream waits for it (marker + waiting kernel, under the per-device gate)
hipStream_t qc_spin_fork(qc_system *S);
int qc_spin_join(qc_system *S);
int qc_spin_join_end(qc_system *S, int *ctl_all, int *ctl_out, unsigned *h_seq, unsigned seq);   // (... and ends the pass: control words, sequence word)
int qc_join_check(qc_system *S);                           // after a host wait: QC_ERR_HIP if a device-side wait of the handle gave up
// host-side time stamps of one SCF pass (QC_ISSUE_DEBUG: where the host's time goes between the end of a pass and the launches of the next
// build; printed as differences at the end of every pass).  No-ops unless the variable is set.
void qc_stamp(const char *what);
void qc_stamp_flush();
void qc_gate_quiet(qc_system *S);                          // the host has seen the handle's stream drained: none of its waits is in flight
// (scale_done: the fixed-point unit of these densities is already in d_fxs - written by the kernel that produced them)
int qc_fock_prepare_device(qc_system *S, const double *dDa, const double *dDb, bool uhf, const void *owner, bool scale_done = false);
// (dH with dFa / dFb: the Fock matrices H + G are written by the closing kernel as well; *f_done tells whether both were)
int qc_fock_build_device(qc_system *S, const double *dDa, const double *dDb, double *dGa, double *dGb, bool uhf, int *twin_cache = nullptr,
                         const double *dH = nullptr, double *dFa = nullptr, double *dFb = nullptr, bool *f_done = nullptr, const void *owner = nullptr,
                         unsigned fork_seq = 0 /* non-zero: speculative build behind the kernel that releases this value of the fork word */);

// dense linear algebra on the handle's stream (all row-major n x n, device pointers)
void qc_gemm(hipStream_t st, int m, int n, int k, double alpha, const double *A, int lda, bool ta, const double *B,
             int ldb, bool tb, double beta, double *C, int ldc, const int *skip = nullptr /* device flag: non-zero = no-op */);
// (`small`: qc_eig_small_doubles(n) doubles - Rayleigh quotients, statistics, partner list, per-workgroup partials)
inline size_t qc_eig_small_doubles(int n) { return (size_t)3 * n + 32 + 8; }
int qc_eig_refine_async(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                        double *t3, double *t4, double *small, int *ctl, int npass);
void qc_diis_solve(hipStream_t st, int m, int minlen, int maxlen, const int *slots, const double *dots, double *B, double *c, int *flag);
void qc_lincomb_dev(hipStream_t st, int n, const double *const *Fs, const double *c_dev, int m, double *out);
// done_tol: the sweeps end after one that met no relative coupling above it (the sweep itself leaves ~done_tol^2 behind)
// notconv (device int, nullable): set to 1 when the sweeps ran out before the criterion was met
int qc_eig_device(hipStream_t st, int n, double *dA /*destroyed*/, double *dV, double *dw, double *d_work, int max_sweeps = 40, double done_tol = 1e-9,
                  int *notconv = nullptr);
int qc_eig_device_warm(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                       int max_sweeps = 40, double done_tol = 1e-9, int *notconv = nullptr);
int qc_eig_device_refine(hipStream_t st, int n, double *dA, const double *dV0, double *dV, double *dw, double *d_work, double *t1, double *t2,
                         double *t3, double *t4, double *small, int *notconv = nullptr);
// Tridiagonalisation-based start vectors (qc_eig_tridiag.hip) + refinement: the cold eigensolve.  ctl[0..3] zero on entry; outcome in
// ctl[0] (1 done: dV / dw hold the sorted eigenpairs; 2: the start was not good enough - repeat with qc_eig_device).  dX0: n*n scratch,
// triwork: qc_eig_tridiag_work_doubles(n).  Smaller matrices (n < QC_TRI_MIN_N) are for the single-workgroup Jacobi kernels.
constexpr int QC_TRI_MIN_N = 24, QC_TRI_MAX_N = 512;
inline bool qc_tri_ok(int n) { return n >= QC_TRI_MIN_N && n <= QC_TRI_MAX_N; }
size_t qc_eig_tridiag_work_doubles(int n);
int qc_eig_tridiag_start(hipStream_t st, int n, const double *dA, double *dX0, double *work);
int qc_eig_cold_async(hipStream_t st, int n, double *dA, double *dX0, double *triwork, double *dV, double *dw, double *d_work, double *t1, double *t2,
                      double *t3, double *t4, double *small, int *ctl, int npass = 3);
// synchronous: cold_async, then the Jacobi kernels if the control word asks for them (ctl: 4 ints of device scratch)
int qc_eig_cold_sync(hipStream_t st, int n, double *dA, double *dX0, double *triwork, double *dV, double *dw, double *d_work, double *t1, double *t2,
                     double *t3, double *t4, double *small, int *ctl, int *notconv = nullptr);
void qc_permute_tensor(hipStream_t st, int n, const double *I, double c_direct, double c_exch, double *T);
void qc_tensor_gemv(hipStream_t st, int n, const double *T1, const double *D1, const double *T2, const double *D2, double *G);
void qc_axpby(hipStream_t st, int n, double a, const double *x, double b, const double *y, double *out);
void qc_sub_transpose(hipStream_t st, int n, const double *M, double *out);              // out = M - M^T
// (fxs non-null: Gt = [hi | lo] planes of 64-bit fixed-point integers, lo_off doubles apart, units fxs[1] = 2^-S and 2^-(S+32))
void qc_symmetrize_add(hipStream_t st, int n, const double *Gt, size_t lo_off, double *G, const double *H, double *F, const double *fxs);   // G = Gt + Gt^T (and F = H + G)
// fixed-point builds on one rank: replica fold + symmetrisation (+ F = H + G) in one launch
// (the replicas are zeroed as they are read: the accumulator planes are clean again when it returns)
void qc_fold_symmetrize(hipStream_t st, int n, int nrep, size_t rep_stride, double *Gt, size_t lo_off, double *G, const double *H, double *F,
                        const double *fxs, unsigned long long *tl = nullptr, const unsigned *join_cnt = nullptr, unsigned join_target = 0,
                        int *timeout_flag = nullptr, long long limit = 0);
// out[p * count + x] = sum_r Gt[p * plane_stride + r * stride + x], p < (fx ? 2 : 1)
void qc_reduce_replicas(hipStream_t st, size_t count, int nrep, size_t stride, const double *Gt, double *out, bool fx, size_t plane_stride);
void qc_count_diff(hipStream_t st, size_t count, const double *a, const double *b, int *flag);
void qc_sync_pack(hipStream_t st, unsigned long long *w, int nwords);     // w[nwords + i] = ~w[i]
int qc_lgc_for(int lab, int lcd, int ncd);
// owner of the i-th (cost-sorted) quartet of launch class `ci`: boustrophedon deal, start rank rotated per class
inline int qc_shard_owner(size_t i, int nranks, size_t ci) {
    const size_t round = i / nranks, pos = i % nranks;
    const size_t r = (round & 1) ? (nranks - 1 - pos) : pos;
    return (int)((r + ci) % nranks);
}
void qc_make_slots(const qc_system *S, const std::vector<QcTask> &tasks, int itmax, bool split_cols, std::vector<QcSlot> &out);
// group tasks by bra into bundles of <= 64 kets (sorted by primitive count); itmax > 0 also cuts the bra primitive range
// (unit > 0: lanes take chunks of at most `unit` primitives of a ket pair, packed into the ketlist entry; 0: whole pairs)
// Returns whether the ketlist entries are PACKED (pair | first primitive << 18 | length << 25, qc_pack_ket_entry) - only possible with
// unit > 0, fewer than 2^18 stored pairs and at most 127 primitives per ket pair - or plain pair indices.  Whoever decodes a list
// (qc_unpack_ket_entry) needs that flag: a plain index above 2^18 would otherwise be cut and its upper bits read as a chunk.
bool qc_make_bundles(const qc_system *S, const std::vector<QcTask> &tasks, int itmax, std::vector<QcBundle> &bundles, std::vector<int> &ketlist, int unit = 0);
constexpr int QC_KET_BITS = 18;
inline unsigned qc_pack_ket_entry(int ket, int kl0, int len) { return (unsigned)ket | ((unsigned)kl0 << QC_KET_BITS) | ((unsigned)len << (QC_KET_BITS + 7)); }
inline void qc_unpack_ket_entry(int entry, bool packed, int *ket, int *kl0, int *len) {
    const unsigned e = (unsigned)entry;
    if (!packed) { *ket = entry; *kl0 = 0; *len = 0; return; }          // (len 0: the whole pair)
    *ket = (int)(e & ((1u << QC_KET_BITS) - 1)); *kl0 = (int)((e >> QC_KET_BITS) & 0x7fu); *len = (int)(e >> (QC_KET_BITS + 7));
}
inline void qc_ensure_lists(qc_system *S) { if (S->lists_stale) qc_build_shards(S); }
inline int qc_unit_of(int LAB, int LCD, bool bm) { return bm ? 2 * (QC_LPAIR + 1) + 2 * LCD + (LAB >= 3 ? 1 : 0) : 2 * LAB + (LCD >= 4 ? 1 : 0); }
// the launch a class belongs to inside a build: with `merge_t1` (bases with f functions) the wide-ket buckets of the bra classes 0 and 1 ride
// in the launch of bra class 2, those of class 4 in the launch of class 3, those of class 6 in the launch of class 5 (qc_fock_tier1_low_kernel); the per-class launches of the profiling / set-up passes use qc_unit_of
inline int qc_build_unit_of(const qc_system *S, int LAB, int LCD, bool bm) {
    if (S->merge_t1 && !bm && LCD >= 4) return qc_unit_of(LAB <= 2 ? 2 : (LAB <= 4 ? 3 : 5), LCD, false);
    if (S->merge_bm && bm && LCD == 0 && LAB >= 3) return qc_unit_of(2, 1, true);       // ss kets / high bras ride with the ps kets / low bras (qc_fock_bm_kernel<3, 0>)
    return qc_unit_of(LAB, LCD, bm);
}
void qc_dots(hipStream_t st, int n, const double *x, const double *const *ys, int ny, double *out);  // device ptr list
void qc_lincomb(hipStream_t st, int n, const double *const *Fs, const double *c, int m, double *out);   // out = sum c_i Fs_i
void qc_scale_cols_invsqrt(hipStream_t st, int n, const double *U, const double *Lam, double *out);
int qc_device_reshard(qc_system *S);
void qc_energy_rms(hipStream_t st, int n, const double *Dnew, const double *Dold, const double *H, const double *G,
                   double *out2, int *ctl = nullptr, int *ctl_out = nullptr);  // out2[0] = 0.5 tr(Dnew (2H+G)), out2[1] = sum_i (Dnew-Dold)_ii^2

// ---- the whole Roothaan step of one spin in one workgroup, matrices in LDS (qc_scf_small.hip): n <= QC_SMALL_MAXN
constexpr int QC_SMALL_MAXN = 64;
struct QcSmallArgs {
    int n, phases;                 // phases: 1 pre, 2 refine, 4 post
    // ---- pre
    const double *F;               // this pass's Fock matrix (null: F = H + G is formed here and stored to F_out)
    double *F_out;                 // the pass's DIIS Fock slot
    const double *D, *S, *X, *H, *G;
    double *E_out;                 // the pass's DIIS error slot
    int m, minlen, maxlen;         // DIIS window length (newest first), Diis::new(minlen, maxlen)
    int dots_generic;              // the DIIS dot products in the summation order of qc_dots_kernel (open-shell runs: bit-identical trajectories)
    int slot[12];
    const double *errs[12], *focks[12];
    double *Bmat, *c_out;          // B (slot-indexed, maxlen x maxlen) and the coefficients, in HBM
    int *diis_flag;
    double *Fp;                    // F' = X^T F_diis X (written by pre, read by a refine / post launch of its own)
    // ---- refine
    const double *V0;              // start vectors
    int npass;
    int *ctl;                      // ctl[0..3] of this spin (0 running / 1 done / 2 rotations needed; last; clean; passes used)
    // ---- post
    const double *Cp_in;           // eigenvectors of F' from another eigensolver (post without refine)
    double *Cp_out, *w_out, *C_out, *Dn;
    const double *Dold;
    int nocc; double dfac;
    double *scal_out;              // [0] 0.5 tr(Dn (2H + G)), [1] sum_i (Dn - Dold)_ii^2
    double *fxs_out; double imax;  // non-null (RHF): the fixed-point unit of the build that will digest Dn goes here (qc_fx_scale)
    int *ctl_all, *ctl_out;        // non-null: hand the 16 control words over to ctl_out and clear them
    unsigned *seq_out; unsigned seq; // non-null (pinned host memory): the pass's sequence number, stored after everything else the host reads
    // non-null: a speculative build of the next pass is queued behind this kernel - release the fork word fork_words[1] = fork_seq at the
    // end; before that, if the pass meets the reference's stopping rule at eps (> 0), cancel that build (fork_words[2], *h_cancel)
    unsigned *fork_words; unsigned fork_seq; double eps; unsigned *h_cancel;
    unsigned long long *tl;        // non-null: [start, end] clock of this launch (QC_DEV_TIMELINE)
};
int qc_scf_small_launch(hipStream_t st, const QcSmallArgs &a);
constexpr int QC_TL_W = 34;   // (words per kernel slot: the launch's end is the maximum over 32 end words - plain stores, no atomics: 8000
                              // workgroups hammering ONE word with atomic min / max stretched an H2O/cc-pVTZ build from 172 to 231 us)
constexpr int QC_TL_PASSES = 64, QC_TL_SLOTS = QC_NUNITS + 4;       // launch units | join wait | fold | Roothaan kernel, first / second launch
int qc_tl_begin_pass(qc_system *S);                                 // (no-op unless QC_DEV_TIMELINE is set)
void qc_tl_dump(qc_system *S);
size_t qc_scf_small_lds_bytes(int n);

#define QC_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            fprintf(stderr, "qchem_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return QC_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)
