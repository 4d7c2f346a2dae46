// qc_fock.hip - device set-up and the per-class launch loop of the direct-SCF Fock build.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qc_fock_kernel.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include "qc_fock_bm.h"

int qc_launch_tier_lab0(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab1(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab2(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab3(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab4(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab5(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier_lab6(int, int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier1_low(int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier1_mid(int, size_t, hipStream_t, const QcTierArgs &);
int qc_launch_tier1_hi(int, size_t, hipStream_t, const QcTierArgs &);

// timing events that are destroyed on every path out of their scope
struct EventList {
    std::vector<hipEvent_t> ev;
    int create(size_t count) {
        ev.assign(count, nullptr);
        for (auto &e : ev) if (hipEventCreate(&e) != hipSuccess) return QC_ERR_HIP;
        return QC_OK;
    }
    ~EventList() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
};

static int launch_tier(int lab, int tier, int grid, size_t lds, hipStream_t st, const QcTierArgs &a) {
    switch (lab) {
        case 0: return qc_launch_tier_lab0(tier, grid, lds, st, a);
        case 1: return qc_launch_tier_lab1(tier, grid, lds, st, a);
        case 2: return qc_launch_tier_lab2(tier, grid, lds, st, a);
        case 3: return qc_launch_tier_lab3(tier, grid, lds, st, a);
        case 4: return qc_launch_tier_lab4(tier, grid, lds, st, a);
        case 5: return qc_launch_tier_lab5(tier, grid, lds, st, a);
        case 6: return qc_launch_tier_lab6(tier, grid, lds, st, a);
    }
    return QC_ERR_UNSUPPORTED;
}

int qc_device_ready(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QC_ERR_NO_DEVICE;
    return QC_OK;
}

// Device records of a bra-major work list (QcBundleDev / QcKetUnit, qc_internal.h) from the host lists of qc_make_bundles
static void qc_bm_device_lists(const qc_system *S, int lcd, const std::vector<QcBundle> &bundles, const std::vector<int> &ketlist, bool packed,
                               std::vector<QcBundleDev> &db, std::vector<QcKetUnit> &du) {
    db.resize(bundles.size()); du.resize(ketlist.size());
    for (size_t i = 0; i < bundles.size(); ++i) {
        const QcBundle &b = bundles[i];
        const QcPairDesc &p = S->pairs[b.bra];
        db[i] = QcBundleDev{b.bra, b.ij_lo, b.ij_hi, b.first, b.nket, b.maxK, p.doff, p.offa, p.offb, p.na | (p.nb << 8) | ((p.shA_eq_shB ? 1 : 0) << 16), b.pad0, 0};
    }
    for (size_t i = 0; i < ketlist.size(); ++i) {
        int ket, kl0, klen;
        qc_unpack_ket_entry(ketlist[i], packed, &ket, &kl0, &klen);
        const QcPairDesc &p = S->pairs[ket];
        const int stride = lcd == 0 ? qc_pair_stride(0, 1) : (lcd == 1 ? 8 : 16);
        const int K = klen ? klen : p.K;
        // (p.p kets: the columns of a lane are the functions of the SECOND shell - bits 18..23 carry its axis permutation, 24..29 the first shell's)
        const int perm_bits = lcd == 2 ? ((((p.psperm >> 6) & 63) << 18) | ((p.psperm & 63) << 24)) : ((p.psperm & 63) << 18);
        du[i] = QcKetUnit{ket, (lcd == 0 ? p.doff : p.psoff) + kl0 * stride, p.offa | (p.offb << 16),
                          (K & 0xffff) | ((lcd == 1 && p.nb == 1 ? 1 : 0) << 16) | ((p.shA_eq_shB ? 1 : 0) << 17) | perm_bits};
    }
}

struct QcLaunchPlan;
static void drop_launch_plan(qc_system *S);
// (the work lists of all classes live in ONE device buffer, qc_system::d_lists - the classes' pointers point into it: an allocation and a
// synchronous copy per list, up to three per class, were 2 ms of a cold handle's set-up on H2O/cc-pVTZ)
static void drop_lists(qc_system *S) {
    for (auto &c : S->classes) { c.d_slots = nullptr; c.d_bundles = nullptr; c.d_ketlist = nullptr; }
    if (S->d_lists) { (void)hipFree(S->d_lists); S->d_lists = nullptr; }
}
static int upload_slots(qc_system *S) {
    drop_launch_plan(S);
    if (S->stream) (void)hipStreamSynchronize(S->stream);         // (nothing in flight reads the old lists)
    drop_lists(S);
    std::vector<unsigned char> blob;
    auto put = [&](const void *src, size_t bytes) -> size_t {
        const size_t off = (blob.size() + 255) & ~(size_t)255;
        blob.resize(off + bytes);
        std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    struct Where { size_t slots = ~(size_t)0, bundles = ~(size_t)0, kets = ~(size_t)0; };
    std::vector<Where> where(S->classes.size());
    {   // (one allocation for the host copy too: growing it list by list copied benzene's 40 MB several times over)
        size_t est = 0;
        for (const auto &c : S->classes) est += c.slots.size() * sizeof(QcSlot) + c.bundles.size() * sizeof(QcBundleDev) + c.ketlist.size() * sizeof(QcKetUnit) + 3 * 256;
        blob.reserve(est);
    }
    for (size_t ci = 0; ci < S->classes.size(); ++ci) {
        auto &c = S->classes[ci];
        if (!c.slots.empty()) where[ci].slots = put(c.slots.data(), c.slots.size() * sizeof(QcSlot));
        if (!c.bundles.empty()) {
            std::vector<QcBundleDev> db; std::vector<QcKetUnit> du;
            qc_bm_device_lists(S, c.LCD, c.bundles, c.ketlist, c.ket_packed, db, du);
            where[ci].bundles = put(db.data(), db.size() * sizeof(QcBundleDev));
            where[ci].kets = put(du.data(), du.size() * sizeof(QcKetUnit));
        }
    }
    if (blob.empty()) return QC_OK;
    QC_HIP_CHECK(hipMalloc(&S->d_lists, blob.size()));
    QC_HIP_CHECK(hipMemcpy(S->d_lists, blob.data(), blob.size(), hipMemcpyHostToDevice));
    for (size_t ci = 0; ci < S->classes.size(); ++ci) {
        auto &c = S->classes[ci];
        if (where[ci].slots != ~(size_t)0) c.d_slots = reinterpret_cast<QcSlot *>(S->d_lists + where[ci].slots);
        if (where[ci].bundles != ~(size_t)0) c.d_bundles = reinterpret_cast<QcBundleDev *>(S->d_lists + where[ci].bundles);
        if (where[ci].kets != ~(size_t)0) c.d_ketlist = reinterpret_cast<QcKetUnit *>(S->d_lists + where[ci].kets);
    }
    return QC_OK;
}

int qc_device_reshard(qc_system *S) {
    S->prepared = false; S->gt_clean = false;                    // a build prepared for the old work lists must not skip the fork of the next one
    S->unit_ms.clear(); S->unit_stream.clear();
    S->cand_skip = false; S->tune_count = 0; S->on = qc_system::QcOnline{};
    S->assign_gen += 1;
    if (S->spec.pending) { if (S->stream) (void)hipStreamSynchronize(S->stream); S->spec.pending = false; }     // (it digests the old lists)
    qc_build_shards(S, !S->device_ready);         // (a handle without its device part builds its lists behind the Schwarz pass, or on demand)
    if (!S->device_ready) return QC_OK;
    return upload_slots(S);
}

// Gather records of the matrix-core classes (qc_fock_body, MFMA branch): step 2's A fragment of k-step ks is, in lane l = 16 q4 + i16 and
// row tile mt, the R value at the Hermite index of h1 + h2 with h1 = 16 mt + i16, h2 = 4 ks + q4.  Which LDS word that is does not
// depend on the quartet: record (ks, l) = eight u16 - byte offsets into the R table for mt = 0..5, one spare, flags (bit 0 = odd ket
// order: the value enters with a minus sign; bit 1 = h2 inside the ket's Hermite range).
static std::vector<unsigned> qc_build_gidx() {
    std::vector<unsigned> out;
    std::vector<int> ht, hu, hv;
    for (int N = 0; N <= QC_LPAIR; ++N)
        for (int t = N; t >= 0; --t)
            for (int u = N - t; u >= 0; --u) { ht.push_back(t); hu.push_back(u); hv.push_back(N - t - u); }
    for (size_t h = 0; h < ht.size(); ++h) if (qc_hidx(ht[h], hu[h], hv[h]) != (int)h) abort();
    for (int LAB = 3; LAB <= 6; ++LAB)
        for (int LCD = 4; LCD <= 6; ++LCD) {
            if ((int)out.size() != 4 * qc_gidx_off(LAB, LCD)) abort();
            const int HAB = qc_nherm(LAB), HCD = qc_nherm(LCD), MT = (HAB + 15) / 16;
            for (int ks = 0; ks < qc_gidx_ksteps(LCD); ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q4 = lane >> 4, i16 = lane & 15, h2 = 4 * ks + q4;
                    const bool ok = h2 < HCD;
                    unsigned short w[8] = {};
                    for (int mt = 0; mt < MT; ++mt) {
                        const int h1 = std::min(16 * mt + i16, HAB - 1), g = ok ? h2 : 0;
                        w[mt] = (unsigned short)(8 * qc_hidx(ht[h1] + ht[g], hu[h1] + hu[g], hv[h1] + hv[g]));
                    }
                    w[7] = ok ? (unsigned short)(2 | ((ht[h2] + hu[h2] + hv[h2]) & 1)) : 0;
                    for (int k = 0; k < 4; ++k) out.push_back((unsigned)w[2 * k] | ((unsigned)w[2 * k + 1] << 16));
                }
        }
    out.resize(out.size() + 4, 0u);
    return out;
}

// Recurrence plans of the cooperative Hermite-Coulomb tables (qc_build_r in qc_fock_kernel.h), every total order 0..QC_LTOT.  Work array
// of order L: level n (the R^n values) starts at rwork(L) - rwork(L - n), inside a level the Hermite index.  Record = {target | source1 << 16,
// source2 | c << 16 | axis << 24}, byte offsets; entries of stage N = t+u+v are contiguous, levels n = 0 .. L-N, position r inside the order.
static std::vector<int> qc_build_rplan() {
    std::vector<int> plan;
    for (int L = 0; L <= QC_LTOT; ++L) {
        if ((int)plan.size() != 2 * qc_plan_off(L)) abort();
        const int RWL = qc_rwork(L);
        for (int N = 1; N <= L; ++N) {
            const int cnt = (N + 1) * (N + 2) / 2;
            for (int n = 0; n <= L - N; ++n)
                for (int r = 0; r < cnt; ++r) {
                    int s = 0;
                    while ((s + 1) * (s + 2) / 2 <= r) ++s;
                    const int v = r - s * (s + 1) / 2, u = s - v, t = N - s;
                    const int o0 = RWL - qc_rwork(L - n), o1 = RWL - qc_rwork(L - n - 1);
                    int s1, s2, c, ax;
                    if (t > 0) { ax = 0; c = t - 1; s1 = qc_hidx(t - 1, u, v); s2 = t > 1 ? qc_hidx(t - 2, u, v) : s1; }
                    else if (u > 0) { ax = 1; c = u - 1; s1 = qc_hidx(t, u - 1, v); s2 = u > 1 ? qc_hidx(t, u - 2, v) : s1; }
                    else { ax = 2; c = v - 1; s1 = qc_hidx(t, u, v - 1); s2 = v > 1 ? qc_hidx(t, u, v - 2) : s1; }
                    const int dst = 8 * (o0 + qc_hidx(t, u, v)), b1 = 8 * (o1 + s1), b2 = 8 * (o1 + s2);
                    plan.push_back(dst | (b1 << 16));
                    plan.push_back(b2 | (c << 16) | (ax << 24));
                }
        }
    }
    plan.push_back(0); plan.push_back(0);
    return plan;
}

static int qc_join_probe(qc_system *S, bool *concurrent);
static int qc_lane_probe(qc_system *S);
static bool qc_stream_pool_take(qc_system *S);
static bool qc_stream_pool_give(qc_system *S);
__global__ void qc_join_mark_kernel(unsigned *cnt);
static void qc_gate_forget(qc_system *S);
static void qc_issue_pool_drop(qc_system *S);
void qc_online_reset(qc_system *S, bool frozen);
void qc_assign_cache_lookup(qc_system *S);
static void qc_assign_cache_store(const qc_system *S);
constexpr int QC_SEARCH_FIRST_BUILD = 24, QC_SEARCH_CHUNK = 8, QC_SEARCH_TRIALS = 240, QC_SEARCH_KICKS = 3;

int qc_device_init(qc_system *S) {
    if (S->device_ready) return QC_OK;
    if (qc_device_ready() != QC_OK) return QC_ERR_NO_DEVICE;
    static const bool sdbg = getenv("QC_SETUP_DEBUG") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt = tnow();
    auto lap = [&](const char *what) { if (sdbg) { const double t = tnow(); fprintf(stderr, "[setup] %-28s %.3f ms\n", what, t - tt); tt = t; } };
    QC_HIP_CHECK(hipGetDevice(&S->device));
    hipDeviceProp_t prop;
    QC_HIP_CHECK(hipGetDeviceProperties(&prop, S->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "qchem_hip: device %d is %s, this library is built for gfx950 only\n", S->device, prop.gcnArchName);
        return QC_ERR_NO_DEVICE;
    }
    // (creating a stream costs ~2 ms - eight of them 17 ms, three times a whole 15-pass SCF of H2O/cc-pVTZ: a handle that goes away
    // leaves its streams, events and measured dispatch lanes in a process-wide pool for the next one)
    bool lanes_known = false;
    if (!S->stream && qc_stream_pool_take(S)) lanes_known = true;
    else {
        if (!S->stream) { QC_HIP_CHECK(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking)); S->own_stream = true; }
        for (int i = 0; i < QC_NSTREAMS; ++i) {
            QC_HIP_CHECK(hipStreamCreateWithFlags(&S->side[i], hipStreamNonBlocking));
            QC_HIP_CHECK(hipEventCreateWithFlags(&S->ev_join[i], hipEventDisableTiming));
        }
        QC_HIP_CHECK(hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming));
    }
    lap("streams and events");
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    QC_HIP_CHECK(hipMalloc(&S->d_pairdata, S->pairdata.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairdata, S->pairdata.data(), S->pairdata.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_pairdataT, S->pairdataT.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairdataT, S->pairdataT.data(), S->pairdataT.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_pspack, (S->pspack.size() + 8) * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_pspack, S->pspack.data(), S->pspack.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_pairs, S->pairs.size() * sizeof(QcPairDesc)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairs, S->pairs.data(), S->pairs.size() * sizeof(QcPairDesc), hipMemcpyHostToDevice));
    lap("pair data upload");
    // rows: F_{L+j}(x_k) / j!, j = 0..7, per total order L; then exp(-x_k)
    std::vector<double> tab((size_t)(QC_LTOT + 1) * QC_BOYS_NGRID * 8 + QC_BOYS_NGRID), row(QC_BOYS_NORD);
    for (int k = 0; k < QC_BOYS_NGRID; ++k) {
        qc_boys_host(QC_BOYS_NORD - 1, k * QC_BOYS_DX, row.data());
        for (int L = 0; L <= QC_LTOT; ++L) {
            double fact = 1.0;
            for (int j = 0; j < 8; ++j) { tab[((size_t)L * QC_BOYS_NGRID + k) * 8 + j] = row[L + j] / fact; fact *= (j + 1); }
        }
        tab[(size_t)(QC_LTOT + 1) * QC_BOYS_NGRID * 8 + k] = std::exp(-k * QC_BOYS_DX);
    }
    lap("Boys tables on the host");
    QC_HIP_CHECK(hipMalloc(&S->d_boys, tab.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_boys, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        const std::vector<int> plan = qc_build_rplan();
        QC_HIP_CHECK(hipMalloc(&S->d_rplan, plan.size() * sizeof(int)));
        QC_HIP_CHECK(hipMemcpy(S->d_rplan, plan.data(), plan.size() * sizeof(int), hipMemcpyHostToDevice));
        const std::vector<unsigned> gi = qc_build_gidx();
        QC_HIP_CHECK(hipMalloc(&S->d_gidx, gi.size() * sizeof(unsigned)));
        QC_HIP_CHECK(hipMemcpy(S->d_gidx, gi.data(), gi.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    QC_HIP_CHECK(hipMalloc(&S->d_D, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_G, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Gtmp, (size_t)2 * QC_NREP * 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Gred, (size_t)2 * 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Dj, nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_flag, 4 * sizeof(int)));
    QC_HIP_CHECK(hipMalloc(&S->d_join, 8 * sizeof(unsigned)));
    QC_HIP_CHECK(hipMemset(S->d_join, 0, 8 * sizeof(unsigned)));
    S->spin_target = 0;
    QC_HIP_CHECK(hipHostMalloc(&S->h_join_timeout, 4 * sizeof(int), hipHostMallocDefault));
    *S->h_join_timeout = 0; S->join_target = 0;
    lap("tables, buffers");
    {
        bool concurrent = true;
        int prc = qc_join_probe(S, &concurrent);
        if (prc != QC_OK) return prc;
        S->join_by_events = !concurrent || getenv("QC_EVENT_JOIN") != nullptr;      // (A/B switch, read per handle: the event join of rounds 1-2)
        S->issue_threads = getenv("QC_ISSUE_THREADS") ? atoi(getenv("QC_ISSUE_THREADS")) : -1;    // (0: never; n: always n helpers; read per handle)
        if (!concurrent && getenv("QC_SCF_DEBUG")) fprintf(stderr, "qchem_hip: kernels of different streams do not run concurrently here (profiler counters?): event join\n");
        if (concurrent && !lanes_known) { prc = qc_lane_probe(S); if (prc != QC_OK) return prc; S->lanes_probed = getenv("QC_NO_LANES") == nullptr; }
    }
    {   // the DS unit's lane order (qc_fock_bm.hip): asked once per device and process
        static std::mutex mu;
        static int known[64];                          // 0 unknown, 1 fixed order, 2 not
        int dev = S->device >= 0 && S->device < 64 ? S->device : 0;
        std::lock_guard<std::mutex> lk(mu);
        if (known[dev] == 0) {
            double *d = nullptr, h[64];
            QC_HIP_CHECK(hipMalloc(&d, 64 * sizeof(double)));
            int prc = qc_ds_order_probe(S->stream, d);
            hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, S->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(S->stream);
            (void)hipFree(d);
            if (prc != QC_OK || e != hipSuccess) return QC_ERR_HIP;
            bool same = true;
            for (int i = 1; i < 64; ++i) same = same && std::memcmp(&h[i], &h[0], sizeof(double)) == 0;
            known[dev] = same ? 1 : 2;
            if (!same) fprintf(stderr, "qchem_hip: this device's DS unit does not add the lanes of one instruction in a fixed order: exchange rows go to global memory directly\n");
        }
        S->ds_order_ok = known[dev] == 1 && getenv("QC_DS_ORDER_FAIL") == nullptr;        // (QC_DS_ORDER_FAIL: test hook - as if the probe had failed)
    }
    lap("join + lane probes");
    QC_HIP_CHECK(hipMalloc(&S->d_fxs, 2 * sizeof(double)));
    // Schwarz factors of the pairs (once per geometry), then the screened work lists
    int rc = qc_schwarz_device(S);
    if (rc != QC_OK) return rc;
    lap("Schwarz pass");
    qc_build_shards(S);
    lap("work lists");
    if ((rc = upload_slots(S)) != QC_OK) return rc;
    lap("upload");
    S->device_ready = true;
    return QC_OK;
}

void qc_device_free(qc_system *S) {
    if (S->stream) (void)hipStreamSynchronize(S->stream);
    qc_gate_forget(S);
    S->spec.pending = false;
    drop_lists(S);
    void *ptrs[] = {S->d_rplan, S->d_gidx, S->d_shells, S->d_pairdata, S->d_pairdataT, S->d_pspack, S->d_pairs, S->d_boys, S->d_D, S->d_G, S->d_Gtmp, S->d_Gred, S->d_Dj, S->d_flag, S->d_fxs};
    S->d_flag = nullptr; S->d_fxs = nullptr;
    qc_issue_pool_drop(S);
    drop_launch_plan(S);
    if (S->d_join) { (void)hipFree(S->d_join); S->d_join = nullptr; }
    if (S->h_join_timeout) { (void)hipHostFree(S->h_join_timeout); S->h_join_timeout = nullptr; }
    for (void *p : ptrs) if (p) (void)hipFree(p);
    S->d_shells = nullptr; S->d_pairdata = S->d_pairdataT = S->d_pspack = nullptr; S->d_pairs = nullptr; S->d_rplan = nullptr; S->d_gidx = nullptr; S->d_boys = S->d_D = S->d_G = S->d_Gtmp = S->d_Gred = S->d_Dj = nullptr;
    if (!qc_stream_pool_give(S)) {
        for (int i = 0; i < QC_NSTREAMS; ++i) {
            if (S->side[i]) (void)hipStreamDestroy(S->side[i]);
            if (S->ev_join[i]) (void)hipEventDestroy(S->ev_join[i]);
        }
        if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
        if (S->own_stream && S->stream) (void)hipStreamDestroy(S->stream);
    }
    for (int i = 0; i < QC_NSTREAMS; ++i) { S->side[i] = nullptr; S->ev_join[i] = nullptr; }
    S->ev_fork = nullptr;
    S->stream = nullptr; S->own_stream = false; S->device_ready = false;
}

// ---- Issuing a build from several host threads.  The runtime needs 5-8 us of host time per launch; a build of H2O/cc-pVTZ is 13 launches
// and 5 markers, so the last one left the caller ~100 us after the first - half the length of the build itself, and every stream's first
// kernel started 6-7 us after the previous stream's.  Launches into DIFFERENT streams are independent in the runtime (one lock per
// stream), so the side streams are shared out between a few helper threads that issue them while the caller issues the handle's own
// chain.  A helper spins for its next job for a few milliseconds after the last one (a pass of a small molecule is 0.35 ms: a parked
// thread's wake-up would cost more than it saves) and sleeps on a condition variable after that.
struct QcIssuePool {
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        bool sleeping = false;
        std::atomic<int> go{0}, done{0};
        std::atomic<bool> quit{false};
        std::function<int()> job;
        int rc = QC_OK;
    };
    std::vector<std::unique_ptr<Worker>> w;
    int gen = 0;
    static void relax() { __builtin_ia32_pause(); }
    static void run(Worker *W, int device) {
        (void)hipSetDevice(device);
        int seen = 0;
        for (;;) {
            int spins = 0;
            while (W->go.load(std::memory_order_acquire) == seen && !W->quit.load(std::memory_order_acquire)) {
                if (++spins < 400000) relax();
                else {
                    std::unique_lock<std::mutex> lk(W->m);
                    W->sleeping = true;
                    W->cv.wait_for(lk, std::chrono::milliseconds(100), [&] { return W->go.load(std::memory_order_acquire) != seen || W->quit.load(std::memory_order_acquire); });
                    W->sleeping = false;
                    spins = 0;
                }
            }
            if (W->quit.load(std::memory_order_acquire)) return;
            seen = W->go.load(std::memory_order_acquire);
            W->rc = W->job();
            W->done.store(seen, std::memory_order_release);
        }
    }
    explicit QcIssuePool(int n, int device) {
        for (int i = 0; i < n; ++i) {
            w.emplace_back(new Worker());
            Worker *W = w.back().get();
            W->th = std::thread(run, W, device);
        }
    }
    void start(int i, std::function<int()> job) {
        Worker *W = w[i].get();
        W->job = std::move(job);
        W->go.store(gen, std::memory_order_release);
        std::lock_guard<std::mutex> lk(W->m);
        if (W->sleeping) W->cv.notify_one();
    }
    int wait(int i) {
        Worker *W = w[i].get();
        while (W->done.load(std::memory_order_acquire) != gen) relax();
        return W->rc;
    }
    ~QcIssuePool() {
        for (auto &W : w) {
            W->quit.store(true, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(W->m); W->cv.notify_one(); }
            if (W->th.joinable()) W->th.join();
        }
    }
};

static void qc_issue_pool_drop(qc_system *S) { delete S->issue_pool; S->issue_pool = nullptr; }

// ---- One handle at a time may have device-side waits in flight on a device.  A waiting kernel sits at the head of its hardware queue
// until the kernel that releases it has run; the argument that this cannot deadlock - every wait is issued after everything it depends
// on, and a hardware queue runs in issue order - holds for ONE issuing sequence.  Two handles issuing from two threads (the header
// allows that) can park handle A's waiter in front of handle B's marker and B's waiter in front of A's: both then wait out their limit.
// So the issue of a build - the only place where cross-stream dependencies are created - goes through a per-device gate: the issuing
// thread holds the gate's mutex while it issues, and if ANOTHER handle still has waits in flight it first waits for that handle's
// stream (those waits finish without any help from the host: everything they depend on was issued before them).  Uncontended cost: one
// mutex per build.
struct QcGate { std::mutex mu; qc_system *owner = nullptr; };
static QcGate &qc_gate_of(int device) {
    static QcGate *gates = new QcGate[64];          // (never destroyed: handles may outlive the static destructors of the process)
    return gates[(device >= 0 && device < 64) ? device : 0];
}
struct QcGateHold {
    QcGate &g;
    qc_system *S;
    bool waits = false;                       // the issue under this hold put device-side waits in flight
    explicit QcGateHold(qc_system *S_) : g(qc_gate_of(S_->device)), S(S_) {
        g.mu.lock();
        if (g.owner && g.owner != S) {
            if (g.owner->stream) (void)hipStreamSynchronize(g.owner->stream);      // (the join wait is the last thing of a build on it)
            g.owner->waits_in_flight = false;
            g.owner = nullptr;
        }
    }
    ~QcGateHold() {
        if (waits) { g.owner = S; S->waits_in_flight = true; }
        g.mu.unlock();
    }
};
static std::vector<std::pair<const char *, double>> *qc_stamps = nullptr;
void qc_stamp(const char *what) {
    static const bool on = getenv("QC_ISSUE_DEBUG") != nullptr;
    if (!on) return;
    if (!qc_stamps) qc_stamps = new std::vector<std::pair<const char *, double>>();
    qc_stamps->emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
void qc_stamp_flush() {
    if (!qc_stamps || qc_stamps->size() < 2) return;
    fprintf(stderr, "[issue]");
    for (size_t i = 1; i < qc_stamps->size(); ++i) fprintf(stderr, " %s %.1f |", (*qc_stamps)[i].first, (*qc_stamps)[i].second - (*qc_stamps)[i - 1].second);
    fprintf(stderr, " total %.1f us\n", qc_stamps->back().second - qc_stamps->front().second);
    qc_stamps->clear();                         // (a fresh stamp behind the printing: the next pass's first difference is the caller's own time
    qc_stamp("printed");                        // between two passes)
}
// ---- device-side timeline (QC_DEV_TIMELINE): see qc_tl_stamp
int qc_tl_begin_pass(qc_system *S) {
    if (getenv("QC_DEV_TIMELINE") == nullptr) { S->tl_cur = nullptr; return QC_OK; }      // (per pass: a harness switches it on after its warm-up)
    const size_t per_pass = (size_t)QC_TL_SLOTS * QC_TL_W, words = (size_t)QC_TL_PASSES * per_pass;
    if (!S->d_tl) {
        QC_HIP_CHECK(hipMalloc(&S->d_tl, words * sizeof(unsigned long long)));
        QC_HIP_CHECK(hipMemset(S->d_tl, 0, words * sizeof(unsigned long long)));
        S->tl_pass = 0;
    }
    S->tl_cur = S->tl_pass < QC_TL_PASSES ? S->d_tl + (size_t)S->tl_pass * per_pass : nullptr;
    ++S->tl_pass;
    return QC_OK;
}
void qc_tl_dump(qc_system *S) {
    if (!S->d_tl || S->tl_pass == 0) return;
    const int np = std::min(S->tl_pass, QC_TL_PASSES);
    const size_t per_pass = (size_t)QC_TL_SLOTS * QC_TL_W, words = (size_t)QC_TL_PASSES * per_pass;
    std::vector<unsigned long long> h(words);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), S->d_tl, words * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long prev_end = 0;
    for (int p = 0; p < np; ++p) {
        const unsigned long long *t = h.data() + (size_t)p * per_pass;
        unsigned long long b[QC_TL_SLOTS], e[QC_TL_SLOTS], t0 = ~0ull, t1 = 0;
        for (int k = 0; k < QC_TL_SLOTS; ++k) {
            b[k] = t[(size_t)k * QC_TL_W]; e[k] = 0;
            for (int w = 1; w <= 32; ++w) e[k] = std::max(e[k], t[(size_t)k * QC_TL_W + w]);
            if (b[k]) { t0 = std::min(t0, b[k]); t1 = std::max(t1, e[k]); }
        }
        if (t1 == 0) continue;
        const unsigned long long base = prev_end ? prev_end : t0;
        fprintf(stderr, "[timeline] pass %d (us from the end of the previous pass; whole pass %.2f):", p, (double)(t1 - base) * 0.01);
        for (int k = 0; k < QC_TL_SLOTS; ++k) {
            if (!b[k]) continue;
            char name[32];
            if (k < QC_NUNITS) snprintf(name, sizeof(name), "unit%d", k);
            else snprintf(name, sizeof(name), "%s", k == QC_NUNITS ? "wait" : k == QC_NUNITS + 1 ? "fold" : k == QC_NUNITS + 2 ? "small" : "small2");
            fprintf(stderr, "  %s %.2f-%.2f", name, ((double)b[k] - (double)base) * 0.01, ((double)e[k] - (double)base) * 0.01);
        }
        fprintf(stderr, "\n");
        prev_end = t1;
    }
    (void)hipFree(S->d_tl);
    S->d_tl = nullptr; S->tl_cur = nullptr; S->tl_pass = 0;
}
// the host has seen the handle's stream drained past its last build: nothing of this handle waits on the device any more
void qc_gate_quiet(qc_system *S) {
    if (!S->waits_in_flight) return;
    QcGate &g = qc_gate_of(S->device);
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.owner == S) g.owner = nullptr;
    S->waits_in_flight = false;
}
static void qc_gate_forget(qc_system *S) {          // the handle goes away
    QcGate &g = qc_gate_of(S->device);
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.owner == S) g.owner = nullptr;
    S->waits_in_flight = false;
}

// ---- Pool of stream sets (the handle's own stream, the side streams, their events, the dispatch lanes measured on them)
struct QcStreamSet {
    int device; hipStream_t main, side[QC_NSTREAMS]; hipEvent_t ev_fork, ev_join[QC_NSTREAMS];
    int slot_side[QC_NSTREAMS], nlanes; bool lane0_is_main;
};
struct QcStreamPool { std::mutex mu; std::vector<QcStreamSet> sets; };
static QcStreamPool &qc_stream_pool() { static QcStreamPool *p = new QcStreamPool(); return *p; }     // (never destroyed: the runtime may be gone by then)
static bool qc_stream_pool_take(qc_system *S) {
    if (getenv("QC_NO_STREAM_POOL") || getenv("QC_NO_LANES")) return false;
    QcStreamPool &P = qc_stream_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (size_t i = 0; i < P.sets.size(); ++i) {
        if (P.sets[i].device != S->device) continue;
        const QcStreamSet t = P.sets[i];
        P.sets.erase(P.sets.begin() + i);
        S->stream = t.main; S->own_stream = true; S->ev_fork = t.ev_fork;
        for (int k = 0; k < QC_NSTREAMS; ++k) { S->side[k] = t.side[k]; S->ev_join[k] = t.ev_join[k]; S->slot_side[k] = t.slot_side[k]; }
        S->nlanes = t.nlanes; S->lane0_is_main = t.lane0_is_main; S->lanes_probed = true;
        return true;
    }
    return false;
}
// (only complete sets whose lanes were measured with the handle's OWN stream; everything on them has been waited for)
static bool qc_stream_pool_give(qc_system *S) {
    if (getenv("QC_NO_STREAM_POOL") || !S->own_stream || !S->stream || !S->lanes_probed || !S->ev_fork) return false;
    for (int k = 0; k < QC_NSTREAMS; ++k) if (!S->side[k] || !S->ev_join[k]) return false;
    if (hipStreamSynchronize(S->stream) != hipSuccess) return false;
    for (int k = 0; k < QC_NSTREAMS; ++k) if (hipStreamSynchronize(S->side[k]) != hipSuccess) return false;
    QcStreamPool &P = qc_stream_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (P.sets.size() >= 8) return false;
    QcStreamSet t{};
    t.device = S->device; t.main = S->stream; t.ev_fork = S->ev_fork;
    for (int k = 0; k < QC_NSTREAMS; ++k) { t.side[k] = S->side[k]; t.ev_join[k] = S->ev_join[k]; t.slot_side[k] = S->slot_side[k]; }
    t.nlanes = S->nlanes; t.lane0_is_main = S->lane0_is_main;
    P.sets.push_back(t);
    return true;
}

// ---- Dispatch lanes.  Measured on MI355X (tools/probes/pipe_probe.hip): with GPU_MAX_HW_QUEUES=8, eight HIP streams land on eight hardware
// queues that sit in PAIRS on four dispatch pipes, and a pipe works on one dispatch packet until every workgroup of that grid has been
// launched - a one-workgroup kernel on stream j completes in 12 us while a grid of 8192 workgroups dispatches on an unrelated stream, and
// only after 160 us (the grid's whole dispatch) when j's queue shares the pipe of that stream (pairs (i, i + 4) in creation order; with the
// runtime's default of four queues the pairs share a QUEUE).  So at most four kernels dispatch at a time, a launch on a fifth stream waits
// for its pipe neighbour, and which streams are neighbours is the runtime's business.  It is measured, once per handle: a busy grid on one
// stream, a marker on the other, the marker's latency against the grid's own duration.  The assignment of launch units then uses
// `nlanes` (<= 4) slots on distinct pipes instead of drawing among seven streams that are not what they seem (QC_NO_LANES: the old draw).
__global__ void qc_probe_busy_kernel(long long ticks) {
    extern __shared__ char probe_lds[];
    if (threadIdx.x == 0) probe_lds[0] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static int qc_lane_probe(qc_system *S) {
    for (int k = 0; k < QC_NSTREAMS; ++k) S->slot_side[k] = k;
    S->nlanes = QC_NSTREAMS; S->lane0_is_main = false;
    if (getenv("QC_NO_LANES")) return QC_OK;
    QcGateHold gate(S);                  // (one probe at a time per device; another handle's grids in flight can still make two lanes look like one - that costs time, never results)
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const int grid = 8192; const long long ticks = 500;                  // 8192 one-wave workgroups of 5 us, 40 KB of LDS each: four per CU
    auto busy = [&](hipStream_t st) { hipLaunchKernelGGL(qc_probe_busy_kernel, dim3(grid), dim3(64), 40 * 1024, st, ticks); };
    auto mark = [&](hipStream_t st) { hipLaunchKernelGGL(qc_join_mark_kernel, dim3(1), dim3(64), 0, st, S->d_join + 3); };
    // warm-up (code upload, queue creation), then the grid alone
    busy(S->stream); mark(S->stream);
    for (int k = 0; k < QC_NSTREAMS; ++k) mark(S->side[k]);
    QC_HIP_CHECK(hipDeviceSynchronize());
    double d_alone = 1e30;
    {
        const auto t0 = clk::now();
        busy(S->stream);
        QC_HIP_CHECK(hipStreamSynchronize(S->stream));
        d_alone = us(t0, clk::now());
    }
    // coupled(x, y): a marker on y waits for the grid on x
    // (one measurement; a positive is measured again - a hiccup of the host must not merge two lanes)
    auto coupled = [&](hipStream_t x, hipStream_t y, bool *out) -> int {
        double lat = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
            busy(x);
            const auto t1 = clk::now();
            mark(y);
            QC_HIP_CHECK(hipStreamSynchronize(y));
            lat = std::min(lat, us(t1, clk::now()));
            QC_HIP_CHECK(hipStreamSynchronize(x));
            if (lat <= 0.5 * d_alone) break;
        }
        *out = lat > 0.5 * d_alone;
        return QC_OK;
    };
    std::vector<std::vector<int>> lanes;      // side-stream indices per pipe; -1 stands for the handle's own stream
    lanes.push_back({-1});
    for (int k = 0; k < QC_NSTREAMS; ++k) {
        bool placed = false;
        for (auto &L : lanes) {
            bool c = false;
            int rc = coupled(L[0] < 0 ? S->stream : S->side[L[0]], S->side[k], &c);
            if (rc != QC_OK) return rc;
            if (c) { L.push_back(k); placed = true; break; }
        }
        if (!placed) lanes.push_back({k});
    }
    // slots: one side stream per pipe first (the pipe of the handle's own stream in front, if a side stream shares it), then the rest
    int n = 0;
    bool used[QC_NSTREAMS] = {};
    S->lane0_is_main = lanes[0].size() > 1;
    for (auto &L : lanes)
        for (int k : L) if (k >= 0) { S->slot_side[n++] = k; used[k] = true; break; }
    S->nlanes = n;
    for (int k = 0; k < QC_NSTREAMS; ++k) if (!used[k]) S->slot_side[n++] = k;
    if (getenv("QC_TUNE_DEBUG") || getenv("QC_SCF_DEBUG")) {
        fprintf(stderr, "[lanes] %d dispatch lanes (grid alone %.0f us):", S->nlanes, d_alone);
        for (auto &L : lanes) { fprintf(stderr, " {"); for (int k : L) fprintf(stderr, k < 0 ? " main" : " s%d", k); fprintf(stderr, " }"); }
        fprintf(stderr, "  slots:"); for (int k = 0; k < QC_NSTREAMS; ++k) fprintf(stderr, " %d", S->slot_side[k]); fprintf(stderr, "\n");
    }
    return QC_OK;
}

// Device-side join of a build's side streams.  Joining through events costs the cross-queue signal path - event packet on the side
// queue, barrier packet on the handle's queue, ~20 us between the last class kernel and the fold on the H2O/cc-pVTZ trace.  Instead every
// side stream ends with a one-lane marker kernel that counts itself (in-queue dependency: a few us), and the handle's stream runs a
// one-lane kernel that waits for the count of this build (monotonic counter, signed comparison) before the fold.
//
// Device-side fork (round 4): the side streams of a SPECULATIVE build - the next SCF pass's build, issued behind this pass's Roothaan step
// before the host has seen the pass end (scf_iterate) - start with the same one-lane waiting kernel, on the fork word (d_join[1]) that the
// kernel which leaves the pass's densities releases.  The host is then off the pass boundary: all launches of the next build sit in their
// queues when the densities become final.
//
// A wait gives up after S->wait_limit ticks of the constant 100 MHz clock (qc_wait_limit: 20 s, or fifty times the build's serial time if
// that is longer) - only possible when the launches it waits for never complete - and says so in pinned memory.  Every host wait that
// follows looks at that word (qc_join_check) and fails THAT call: a build whose join gave up has folded an incomplete matrix.
__global__ void qc_join_mark_kernel(unsigned *cnt) {
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// (the poll is a RELAXED agent-scope load - it goes past the non-coherent cache levels without invalidating anything; an acquire load
// in the loop invalidates the waiter's L2 every time round, and five waiters doing that every 100 ns through a whole Roothaan step cost
// the kernels running beside them 30 % - measured: iteration 0.33 -> 0.46 ms.  One acquire fence once the word is there.)
// (`delay`, ticks of the 100 MHz clock: a fork waiter lets that much time pass after the word has arrived - the side chains of a build
// start a few microseconds apart, heaviest first, as they do when the host issues them one by one: released all at once they take each
// other's wave slots from the first cycle and the chain that ends the build loses its head start - measured, see launch_concurrent)
__global__ void qc_join_wait_kernel(unsigned *cnt, unsigned target, int *timeout_flag, long long limit, int delay, unsigned long long *tl = nullptr) {
    if (threadIdx.x != 0) return;
    qc_tl_stamp(tl, 0);
    if (delay < 0) {
        // the join of a build (delay = -1): nothing runs next to this lane that its polling could disturb for long, and every 1.7 us step
        // of the gentle loop below is 0.85 us, on average, between the last marker and the fold - one load per ~0.2 us here
        long long t0 = 0;
        unsigned it = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if ((++it & 127u) == 0) {
                const long long t = wall_clock64();
                if (t0 == 0) t0 = t;
                else if (t - t0 > limit) { __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        qc_tl_stamp(tl, 1);
        return;
    }
    // (poll gently: one load per ~1.7 us, the clock only every 16th time round - a fork waiter spins through a whole Roothaan step next to
    // the one workgroup that runs it, and whatever it does to the memory system of its CU that workgroup pays)
    long long t0 = 0;
    unsigned it = 0;
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if ((++it & 15u) == 0) {
            const long long t = wall_clock64();
            if (t0 == 0) t0 = t;
            else if (t - t0 > limit) { __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
    }
    if (delay > 0) {
        const long long t1 = wall_clock64();
        while (wall_clock64() - t1 < delay) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    qc_tl_stamp(tl, 1);
}
// The word that ends an SCF pass on the device, for the launch sequences that have no kernel of their own to do it in (the one-workgroup
// Roothaan kernel of small closed-shell runs does the same at its end, qc_scf_small.hip): decide whether the host - which has promised to
// stop once the reference's stopping rule holds at `eps` (rhf.rs:94 / uhf.rs:139) - will take another pass; if not, the speculative build
// behind this kernel is cancelled (cancel word = its number: the class kernels return at once), and the host is told so in pinned memory
// before it sees the pass end; then release the fork word.  scal: [0.5 tr(D(2H+G)), sum_i dD_ii^2] per spin, wherever the pass left them.
__global__ void qc_spec_release_kernel(unsigned *words, unsigned seq, const double *scal, int n, int nspin, double eps, unsigned *h_cancel,
                                       unsigned *h_seq, unsigned seqval) {
    if (threadIdx.x != 0) return;
    bool stop = false;
    if (eps > 0.0) {
        double rms = 0.0;
        for (int s = 0; s < nspin; ++s) rms += sqrt(scal[2 * s + 1] / n);
        // (uhf.rs:137-139: density_rms = (rms_a + rms_b) / 2, test rms / 2 < eps;  rhf.rs:94: rms < eps)
        stop = nspin == 2 ? (rms / 2.0 / 2.0 < eps) : (rms < eps);
    }
    if (stop) {
        __hip_atomic_store(words + 2, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h_cancel) __hip_atomic_store(h_cancel, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(words + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // (h_seq: this kernel also ends the pass for a host that polls the pinned sequence word - after the cancel word it may look at)
    if (h_seq) __hip_atomic_store(h_seq, seqval, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void qc_spec_release(hipStream_t st, unsigned *words, unsigned seq, const double *scal, int n, int nspin, double eps, unsigned *h_cancel,
                     unsigned *h_seq, unsigned seqval) {
    hipLaunchKernelGGL(qc_spec_release_kernel, dim3(1), dim3(64), 0, st, words, seq, scal, n, nspin, eps, h_cancel, h_seq, seqval);
}

// twenty seconds of the 100 MHz clock, or fifty times the serial time of the build's launches when that is longer (a side chain of a
// very large system may legitimately end seconds after the handle's own); QC_WAIT_LIMIT_MS overrides (tests force a tiny limit)
static long long qc_wait_limit(const qc_system *S) {
    const char *env = getenv("QC_WAIT_LIMIT_MS");          // (read per build: a test switches it on and off inside one process)
    const double env_ms = env ? atof(env) : 0.0;
    if (env_ms > 0.0) return std::max(1LL, (long long)(env_ms * 1e5));
    double serial_ms = 0.0;
    for (float x : S->unit_ms) serial_ms += x;
    return (long long)(std::max(20000.0, 50.0 * serial_ms) * 1e5);
}

hipStream_t qc_spin_fork(qc_system *S) {
    QcGateHold gate(S);
    hipStream_t side = S->side[S->slot_side[S->lane0_is_main ? 1 : 0]];
    if (hipEventRecord(S->ev_fork, S->stream) != hipSuccess || hipStreamWaitEvent(side, S->ev_fork, 0) != hipSuccess) return nullptr;
    return side;
}
int qc_spin_join(qc_system *S) {
    QcGateHold gate(S);
    hipStream_t side = S->side[S->slot_side[S->lane0_is_main ? 1 : 0]];
    S->spin_target += 1;
    S->wait_limit = qc_wait_limit(S);
    hipLaunchKernelGGL(qc_join_mark_kernel, dim3(1), dim3(64), 0, side, S->d_join + 4);
    hipLaunchKernelGGL(qc_join_wait_kernel, dim3(1), dim3(64), 0, S->stream, S->d_join + 4, S->spin_target, S->h_join_timeout, S->wait_limit, 0, (unsigned long long *)nullptr);
    gate.waits = true;
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

// The join of the two spins' one-workgroup Roothaan kernels (scf_iterate: alpha on the handle's stream, beta on a side stream) that also
// ends the pass: once the beta stream's marker is in, the sixteen control words go to the host and are cleared, then the pass's sequence
// word - what the last spin's kernel does itself when the spins run one after the other (qc_scf_small.hip).
__global__ void qc_spin_join_end_kernel(unsigned *cnt, unsigned target, int *timeout_flag, long long limit, int *ctl_all, int *ctl_out,
                                        unsigned *h_seq, unsigned seq) {
    if (threadIdx.x == 0) {
        long long t0 = 0;
        unsigned it = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if ((++it & 63u) == 0) {
                const long long t = wall_clock64();
                if (t0 == 0) t0 = t;
                else if (t - t0 > limit) { __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x < 16) {
        int *p = ctl_all + threadIdx.x;
        ctl_out[threadIdx.x] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && h_seq) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int qc_spin_join_end(qc_system *S, int *ctl_all, int *ctl_out, unsigned *h_seq, unsigned seq) {
    QcGateHold gate(S);
    hipStream_t side = S->side[S->slot_side[S->lane0_is_main ? 1 : 0]];
    S->spin_target += 1;
    S->wait_limit = qc_wait_limit(S);
    hipLaunchKernelGGL(qc_join_mark_kernel, dim3(1), dim3(64), 0, side, S->d_join + 4);
    hipLaunchKernelGGL(qc_spin_join_end_kernel, dim3(1), dim3(64), 0, S->stream, S->d_join + 4, S->spin_target, S->h_join_timeout, S->wait_limit, ctl_all, ctl_out, h_seq, seq);
    gate.waits = true;
    return hipGetLastError() == hipSuccess ? QC_OK : QC_ERR_HIP;
}

// After a host wait that follows a device-joined build: did one of its waits give up?  Then the matrix it folded was not complete: the
// call fails (last_error says why), the accumulator planes are no longer known to be clean, the counter and the host's target meet
// again, and this handle joins through events from now on.
int qc_join_check(qc_system *S) {
    if (!S->h_join_timeout || !__atomic_load_n(S->h_join_timeout, __ATOMIC_ACQUIRE)) return QC_OK;
    const int who = __atomic_load_n(S->h_join_timeout, __ATOMIC_ACQUIRE);         // (1: a waiting kernel, 2: the closing fold)
    S->last_error = "a device-side wait of the Fock build gave up (QC_WAIT_LIMIT_MS): a launch it depended on never finished; "
                    "this handle joins its streams through events from now on";
    fprintf(stderr, "qchem_hip: %s\n", S->last_error.c_str());
    (void)hipDeviceSynchronize();
    unsigned c[5] = {0, 0, 0, 0, 0};
    if (hipMemcpy(c, S->d_join, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "qchem_hip: join counter %u, the wait wanted %u (%s); fork word %u, spin counter %u / %u\n", c[0], S->join_target,
                who == 2 ? "by the closing fold" : "by a waiting kernel", c[1], c[4], S->spin_target);
        S->join_target = c[0]; S->fork_seq = std::max(S->fork_seq, c[1]); S->spin_target = c[4]; }
    __atomic_store_n(S->h_join_timeout, 0, __ATOMIC_RELEASE);
    S->join_by_events = true;
    S->gt_clean = false; S->prepared = false; S->spec.pending = false;
    return QC_ERR_HIP;
}

// The device-side join needs kernels of different streams to RUN concurrently: a waiting kernel whose marker cannot start would wait
// out its limit.  That is the case whenever something serialises dispatches - rocprofv3 counter collection (--pmc) does, so do the
// runtime's debugging switches.  Asked once per handle: a waiting kernel on one stream, then its marker on another; if the wait gives up
// after 2 ms, this handle joins through events (as QC_EVENT_JOIN does).
static int qc_join_probe(qc_system *S, bool *concurrent) {
    QcGateHold hold(S);
    *S->h_join_timeout = 0;
    S->join_target += 1;
    hipLaunchKernelGGL(qc_join_wait_kernel, dim3(1), dim3(64), 0, S->stream, S->d_join, S->join_target, S->h_join_timeout, 200000LL, 0, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(qc_join_mark_kernel, dim3(1), dim3(64), 0, S->side[0], S->d_join);
    if (hipGetLastError() != hipSuccess) return QC_ERR_HIP;
    QC_HIP_CHECK(hipStreamSynchronize(S->stream));
    QC_HIP_CHECK(hipStreamSynchronize(S->side[0]));
    *concurrent = *S->h_join_timeout == 0;
    *S->h_join_timeout = 0;
    return QC_OK;
}

static QcKernelArgs base_args(qc_system *S, const QcFockArgs &fa) {
    QcKernelArgs a{};
    a.pairs = S->d_pairs; a.pairdata = S->d_pairdata; a.pairdataT = S->d_pairdataT; a.boys = S->d_boys; a.rplan = reinterpret_cast<const int2 *>(S->d_rplan); a.gidx = reinterpret_cast<const uint4 *>(S->d_gidx); a.n = S->nbasis;
    a.Dj = fa.Dj; a.Dk0 = fa.Dk0; a.Dk1 = fa.Dk1; a.G0 = fa.G0; a.G1 = fa.G1; a.cK = fa.cK; a.eri_out = fa.eri_out;
    a.nrep = fa.nrep > 0 ? fa.nrep : 1; a.rep_stride = fa.rep_stride; a.fxs = fa.fxs; a.fx_lo = fa.fx_lo; a.schwarz_out = fa.schwarz_out;
    a.cancel = fa.fork_seq ? S->d_join + 2 : nullptr; a.cancel_seq = fa.fork_seq;
    return a;
}

// segment of one launch: a class bucket with its slots (column kernels) or bundles (bra-major kernels)
struct Seg { const QcClass *c; const QcSlot *d_slots; int nslots; const QcBundleDev *d_bundles = nullptr; const QcKetUnit *d_ketlist = nullptr; int lds = 0;
             int run = 0, rb_rows = 0; };     // (bra-run mode of the class's own slot list; the set-up passes bring independent slots)

static Seg seg_of(const QcClass &c) {
    if (c.bm) return Seg{&c, nullptr, (int)c.bundles.size(), c.d_bundles, c.d_ketlist, c.lds_bytes};
    Seg sg{&c, c.d_slots, (int)c.slots.size()};
    sg.run = c.run; sg.rb_rows = c.rb_rows;
    return sg;
}

struct QcLaunchPlan { std::vector<std::vector<int>> units; std::vector<std::vector<Seg>> segs; };

static void drop_launch_plan(qc_system *S) { delete S->launch_plan; S->launch_plan = nullptr; }

static int launch_segments(qc_system *S, int unit, const std::vector<Seg> &segs, hipStream_t st, const QcKernelArgs &base) {
    if (unit >= 2 * (QC_LPAIR + 1)) {      // bra-major launch
        QcBmArgs t{};
        t.base = base; t.pairdataT = S->d_pairdataT; t.pspack = S->d_pspack;
        t.base.tl = S->tl_cur ? S->tl_cur + QC_TL_W * unit : nullptr;
        const int v = unit - 2 * (QC_LPAIR + 1);
        const int lds_max = 160 * 1024 - 512;
        int nw = qc_bm_waves(v / 2, v % 2), iblock = 0, rows = 0;
        for (const Seg &sg : segs) { iblock = std::max(iblock, sg.lds); rows = std::max(rows, sg.c->bm_rows); }
        // exchange rows of a wave's current bra in LDS: (na + nb) rows x n columns per spin
        const int rowbytes = (base.Dk1 ? 2 : 1) * rows * S->nbasis * 8;
        // (QC_BM_NO_ROWBUF forces the large-n fallback - direct global atomics per bundle - so that tests can reach it)
        t.use_rowbuf = (base.eri_out == nullptr && base.schwarz_out == nullptr && QC_BM_LDS_TABLE + 2 * (iblock + rowbytes) <= lds_max && S->ds_order_ok && !getenv("QC_BM_NO_ROWBUF")) ? 1 : 0;
        const int wbytes = iblock + (t.use_rowbuf ? rowbytes : 0);
        while (nw > 1 && QC_BM_LDS_TABLE + nw * wbytes > lds_max) nw = nw > 4 ? nw - 1 : nw / 2;   // Cartesian d / f bras: 36+ rows of I per wave
        {   // (QC_BM_NW: experiment switch - at most that many waves per workgroup, so that two workgroups of different launches fit a CU's LDS)
            static const int nw_env = getenv("QC_BM_NW") ? atoi(getenv("QC_BM_NW")) : 0;
            if (nw_env > 0) nw = std::min(nw, nw_env);
        }
        int grid = 0, k = 0;
        for (const Seg &sg : segs) {
            // persistent workgroups of `nw` waves: at most ~12 waves per CU, each wave strides through the bundle list
            static const int capw = getenv("QC_BM_WAVES_PER_CU") ? std::max(4, atoi(getenv("QC_BM_WAVES_PER_CU"))) : 12;   // (A/B switch)
            grid += std::min((sg.nslots + nw - 1) / nw, 256 * capw / nw);
            t.seg_end[k] = grid; t.seg_lab[k] = sg.c->LAB; t.seg_lcd[k] = sg.c->LCD; t.seg_bundles[k] = sg.d_bundles; t.seg_ketlist[k] = sg.d_ketlist;
            t.seg_nbundles[k] = sg.nslots; t.seg_iwords[k] = sg.lds / 8;
            t.seg_rows[k] = rows;
            ++k;
        }
        t.nseg = k;
        const int lds = QC_BM_LDS_TABLE + nw * wbytes;
        static const bool lds_dbg = getenv("QC_LDS_DEBUG") != nullptr;
        if (lds_dbg) fprintf(stderr, "[lds] bm<%d,%d>: %d workgroups x %d waves, %d bytes of LDS per workgroup (table %d, per wave %d)\n", v / 2, v % 2, grid, nw, lds, QC_BM_LDS_TABLE, wbytes);
        {   // ss-ket segments in the launch of the ps kets / low bras: the merged kernel
            bool mixed = false;
            for (const Seg &sg : segs) mixed = mixed || sg.c->LCD != v / 2;
            if (mixed) return (v == 2 && k <= QC_MAXSEG) ? qc_launch_bm(3, 0, grid, nw, (size_t)lds, st, t) : QC_ERR_UNSUPPORTED;
        }
        return qc_launch_bm(v / 2, v % 2, grid, nw, (size_t)lds, st, t);
    }
    QcTierArgs t{};
    t.base = base;
    t.base.tl = S->tl_cur ? S->tl_cur + QC_TL_W * unit : nullptr;
    int grid = 0, lds = 0, k = 0;
    for (const Seg &sg : segs) {
        const int G = 64 >> sg.c->LGC;
        const int waves = (sg.nslots + G - 1) / G;
        int seg_lds = sg.c->lds_bytes;
        t.seg_run[k] = sg.run; t.seg_rbrows[k] = 0;
        if (sg.run > 0) {
            grid += (waves + sg.run - 1) / sg.run;
            // row buffer of the wave: exchange rows per spin + the J_ab block; only while it leaves the class its waves per CU
            const int rb_bytes = ((base.Dk1 ? 2 : 1) * sg.rb_rows * S->nbasis + sg.rb_rows * sg.rb_rows) * 8;
            static const bool no_rb = getenv("QC_NO_ROWBUF") != nullptr;             // (A/B switch)
            if (!no_rb && base.eri_out == nullptr && base.schwarz_out == nullptr && rb_bytes <= 6 * 1024) { t.seg_rbrows[k] = sg.rb_rows; seg_lds += rb_bytes; }
        } else {
            static const int per_cu = getenv("QC_TIER_WG_PER_CU") ? std::max(1, atoi(getenv("QC_TIER_WG_PER_CU"))) : 32;     // (A/B switch)
            grid += std::min(waves, 256 * per_cu);
        }
        t.seg_end[k] = grid; t.seg_code[k] = (sg.c->LCD << 4) | sg.c->LGC; t.seg_lab[k] = sg.c->LAB;
        t.seg_nslots[k] = sg.nslots; t.seg_words[k] = sg.c->slot_words; t.seg_slots[k] = sg.d_slots;
        lds = std::max(lds, seg_lds);
        ++k;
    }
    t.nseg = k;
    {   // (experiment switch: extra dynamic LDS per workgroup - does the build respond to the LDS the column kernels hold?)
        static const int pad_kb = getenv("QC_LDS_PAD_KB") ? atoi(getenv("QC_LDS_PAD_KB")) : 0;
        if (pad_kb > 0 && lds + pad_kb * 1024 <= QC_LDS_MAX) lds += pad_kb * 1024;
    }
    {   // segments of other bra classes than the unit's own: the merged wide-ket launch of the low bra classes
        bool mixed = false;
        for (const Seg &sg : segs) mixed = mixed || sg.c->LAB != unit / 2;
        if (mixed) {
            if (k > QC_MAXSEG) return QC_ERR_UNSUPPORTED;
            static const bool lds_dbg2 = getenv("QC_LDS_DEBUG") != nullptr;
            if (lds_dbg2) fprintf(stderr, "[lds] merged wide-ket launch of unit %d: %d workgroups (1 wave), %d bytes of LDS per workgroup, %d segments\n", unit, grid, lds, k);
            if (unit == 2 * 2 + 1) return qc_launch_tier1_low(grid, (size_t)lds, st, t);
            if (unit == 2 * 3 + 1) return qc_launch_tier1_mid(grid, (size_t)lds, st, t);
            if (unit == 2 * 5 + 1) return qc_launch_tier1_hi(grid, (size_t)lds, st, t);
            return QC_ERR_UNSUPPORTED;
        }
    }
    // a wide-ket launch made of d.d / f.p-ket buckets only (no basis function above d): the kernel variant without the f-ket bodies
    int tier = unit % 2;
    if (tier == 1) {
        bool only4 = true;
        for (const Seg &sg : segs) only4 = only4 && sg.c->LCD == 4;
        if (only4 && !S->has_fkets) tier = 2;      // (with f kets around, the d.d / f.p-ket classes may be in their matrix-core form: the f-capable kernel)
    }
    static const bool lds_dbg = getenv("QC_LDS_DEBUG") != nullptr;
    if (lds_dbg) {
        fprintf(stderr, "[lds] tier<%d,%d>: %d workgroups (1 wave), %d bytes of LDS per workgroup; segments:", unit / 2, tier, grid, lds);
        for (const Seg &sg : segs) fprintf(stderr, " <%d,%d,%d> %d slots %d B", sg.c->LAB, sg.c->LCD, sg.c->LGC, sg.nslots, sg.c->lds_bytes);
        fprintf(stderr, "\n");
    }
    return launch_tier(unit / 2, tier, grid, (size_t)lds, st, t);
}

// Launch units of one build: per bra class LAB, tier 0 (LCD <= 3) and tier 1 (LCD >= 4) of the column kernels, plus
// the four bra-major launches (ket type x bra range); segments heaviest first.
static void tier_units(qc_system *S, std::vector<std::vector<int>> &units) {
    units.assign(QC_NUNITS, {});
    for (size_t ci = 0; ci < S->classes.size(); ++ci) {
        const QcClass &c = S->classes[ci];
        if (c.slots.empty() && c.bundles.empty()) continue;
        units[qc_build_unit_of(S, c.LAB, c.LCD, c.bm)].push_back((int)ci);
    }
    for (auto &u : units)
        std::stable_sort(u.begin(), u.end(), [&](int x, int y) {
            const QcClass &a_ = S->classes[x], &b_ = S->classes[y];
            if (a_.LCD != b_.LCD) return a_.LCD > b_.LCD;              // long serial chains first
            return a_.flops_alg > b_.flops_alg;
        });
}

// Normal mode: the (at most 14) tier launches are independent (they only meet in the atomically accumulated Gt
// replicas), so each goes to its own side stream forked from / joined to the handle's stream.
// Profiling mode (class_ms != nullptr): one single-segment launch per class bucket, serial on the handle's stream with
// a hipEvent between consecutive launches; unit_ms (optional, 14 entries) times the real tier launches the same way.
int qc_launch_fock_classes(qc_system *S, const QcFockArgs &fa, float *class_ms, float *unit_ms, bool nofork) {
    const QcKernelArgs a0 = base_args(S, fa);
    const QcKernelArgs &a = a0;
    // (the launch units and their segments only change with the work lists: kept between builds, dropped by upload_slots)
    if (!S->launch_plan) {
        S->launch_plan = new QcLaunchPlan();
        tier_units(S, S->launch_plan->units);
        for (const auto &u : S->launch_plan->units) {
            std::vector<Seg> v;
            for (int ci : u) v.push_back(seg_of(S->classes[ci]));
            S->launch_plan->segs.push_back(std::move(v));
        }
    }
    const std::vector<std::vector<int>> &units = S->launch_plan->units;
    auto segs_of = [&](const std::vector<int> &u) -> const std::vector<Seg> & { return S->launch_plan->segs[&u - units.data()]; };
    if (class_ms || unit_ms) {
        const size_t nev = (class_ms ? S->classes.size() : units.size()) + 1;
        EventList evl;
        if (evl.create(nev) != QC_OK) return QC_ERR_HIP;
        std::vector<hipEvent_t> &ev = evl.ev;
        QC_HIP_CHECK(hipEventRecord(ev[0], S->stream));
        if (class_ms) {
            for (size_t ci = 0; ci < S->classes.size(); ++ci) {
                const QcClass &c = S->classes[ci];
                if (!c.slots.empty() || !c.bundles.empty()) {
                    int rc = launch_segments(S, qc_unit_of(c.LAB, c.LCD, c.bm), {seg_of(c)}, S->stream, a);
                    if (rc != QC_OK) return rc;
                }
                QC_HIP_CHECK(hipEventRecord(ev[ci + 1], S->stream));
            }
        } else {
            for (size_t u = 0; u < units.size(); ++u) {
                if (!units[u].empty()) { int rc = launch_segments(S, (int)u, segs_of(units[u]), S->stream, a); if (rc != QC_OK) return rc; }
                QC_HIP_CHECK(hipEventRecord(ev[u + 1], S->stream));
            }
        }
        QC_HIP_CHECK(hipEventSynchronize(ev.back()));
        float *out = class_ms ? class_ms : unit_ms;
        for (size_t i = 0; i + 1 < nev; ++i) QC_HIP_CHECK(hipEventElapsedTime(&out[i], ev[i], ev[i + 1]));
        return QC_OK;
    }
    // Launch units are independent; they go to QC_NSTREAMS side streams (one hardware queue each next to the main
    // stream's, GPU_MAX_HW_QUEUES=8).  Kernels on one stream run in order, so the assignment matters: the first build
    // of a handle times every unit alone (density-independent), then units are placed longest-first on the least
    // loaded stream.
    // (`head_start`, ms: stream 0 is given that much more work than the others.  The most loaded stream becomes the handle's own stream,
    // and a build that ends on the handle's stream goes straight on to the fold, while one that ends on a side stream first pays the
    // cross-queue signal - event packet, barrier packets, ~20 us on the H2O/cc-pVTZ trace.)
    auto lpt = [&](const std::vector<float> &w, int nstreams, float head_start = 0.f, bool bm_on_second = false) {
        nstreams = std::min(nstreams, S->nlanes);              // (slots beyond the dispatch lanes share a pipe with an earlier one)
        std::vector<int> ord;
        for (size_t u = 0; u < units.size(); ++u) if (!units[u].empty()) ord.push_back((int)u);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return w[x] > w[y]; });
        std::vector<float> load(nstreams, 0.f);
        load[0] = -head_start;
        S->unit_stream.assign(units.size(), 0);
        // (bm_on_second: the bra-major launches - persistent grids that are dispatched at once and do not hold their pipe - on the SECOND
        // streams of the pipes, every one its own chain, next to the column launches on the lanes.  H2O/cc-pVTZ without any search:
        // 0.341 against 0.350 ms per iteration, benzene 1.730 against 1.718; as a proposal of the search, with moves onto the second
        // streams allowed, the search ended on assignments that were fast back to back and slow inside SCF passes (0.199 ms builds
        // against 0.192) and found nothing for benzene.  QC_BM_PAIRED keeps the experiment.)
        static const bool bm_env = getenv("QC_BM_PAIRED") != nullptr;
        const bool bm_paired = bm_on_second || bm_env;
        int nextb = S->nlanes;
        for (int u : ord) {
            if (bm_paired && u >= 2 * (QC_LPAIR + 1) && S->nlanes == nstreams && nextb < QC_NSTREAMS) { S->unit_stream[u] = nextb++; continue; }
            const int k = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            S->unit_stream[u] = k;
            load[k] += w[u];
        }
        S->unit_weight = w;
    };
    // one concurrent build: fork the side streams off the handle's stream, launch every unit on its stream (heaviest
    // first), join.  `ev` (tuning only): [0] fork, [1] join, [2 + 2u], [3 + 2u] around unit u.
    auto launch_concurrent = [&](hipEvent_t *ev, bool per_unit, unsigned fork_seq, bool fold_joins = false) -> int {
        S->fold_join_pending = false;
        qc_stamp("to launch_concurrent");
        QcGateHold gate(S);
        qc_stamp("gate");
        QcKernelArgs a = a0;                       // (a speculative / spec-form build carries its number: the cancel word may empty it)
        a.cancel = fork_seq ? S->d_join + 2 : nullptr; a.cancel_seq = fork_seq;                        // (cross-stream dependencies are created here and nowhere else: see QcGate)
        if (ev) QC_HIP_CHECK(hipEventRecord(ev[0], S->stream));
        // (nofork: everything the launches depend on has completed - the host waited for the handle's stream after it was enqueued)
        static const bool force_fork = getenv("QC_FORCE_FORK") != nullptr;          // (A/B switch)
        // device-side fork (speculative build): the side streams' first kernels wait for the fork word instead of an event of the
        // handle's stream - what they depend on has not even started when they are issued
        const bool devfork = fork_seq != 0;
        const bool fork = !devfork && (ev != nullptr || !nofork || force_fork);
        if (fork) QC_HIP_CHECK(hipEventRecord(S->ev_fork, S->stream));
        S->wait_limit = qc_wait_limit(S);
        // Issue order (the host needs ~8 us per launch, so it matters): inside a stream heaviest first; across streams the
        // first launch of every stream before any second one, streams in the order of their total load - the chain that
        // ends the build gets going first and no stream sits empty while another one's queue is being filled.
        std::vector<int> q[QC_NSTREAMS];
        int ks[QC_NSTREAMS];
        {
            std::vector<int> byw;
            for (size_t u = 0; u < units.size(); ++u) if (!units[u].empty()) byw.push_back((int)u);
            // (an entry of unit_stream is lane | rank << 3: inside a lane the launches go out by rank, then heaviest first - the rank is how
            // the assignment search puts a lighter launch in front of a heavier one)
            std::stable_sort(byw.begin(), byw.end(), [&](int x, int y) {
                const int rx = S->unit_stream[x] >> 3, ry = S->unit_stream[y] >> 3;
                return rx != ry ? rx < ry : S->unit_weight[x] > S->unit_weight[y];
            });
            float load[QC_NSTREAMS] = {};
            for (int u : byw) { q[S->unit_stream[u] & 7].push_back(u); load[S->unit_stream[u] & 7] += S->unit_weight[u]; }
            for (int k = 0; k < QC_NSTREAMS; ++k) ks[k] = k;
            std::stable_sort(ks, ks + QC_NSTREAMS, [&](int x, int y) { return load[x] > load[y]; });
        }
        // the most loaded chain runs on the handle's own stream: no fork hop before it, no join after it.
        // NOT in a speculative build: there the kernel in front of the build - the Roothaan step that releases the fork word before it
        // ends - retires slowly once the side chains have started (its end-of-kernel cache write-back competes with their traffic: 100 us
        // instead of 5 on the H2O/cc-pVTZ trace), and a chain queued behind it on the handle's stream starts that much later.  All chains
        // of a speculative build go to side streams; the handle's stream carries the wait and the fold, which have to wait anyway.
        // (with the dispatch lanes known, slot 0 is the slot on the pipe of the handle's own stream: its chain is the one that runs there -
        // any other choice would put two chains on one pipe; the longest-first rule gives slot 0 the heaviest launch anyway)
        const int kmain = devfork ? -1 : (S->lane0_is_main ? (q[0].empty() ? -1 : 0) : ks[0]);
        // (a speculative build keeps the pipe of the handle's own stream free of its waiting kernels: a queue whose head packet waits
        // behind a running kernel slows the other queue of its pipe - the Roothaan step on the handle's stream ran 1.2-1.7x longer with
        // a waiter next door - so the chain of slot 0 goes to the second stream of lane 1's pipe)
        static const bool spec_keep_lane0 = getenv("QC_SPEC_LANE0") != nullptr;      // (A/B switch)
        auto side_slot = [&](int k) { return (devfork && S->lane0_is_main && !spec_keep_lane0 && k == 0 && S->nlanes + 0 < QC_NSTREAMS) ? S->nlanes : k; };
        const bool event_join = S->join_by_events;                   // (QC_EVENT_JOIN, or dispatches are serialised here: qc_device_init)
        if (devfork && event_join) return QC_ERR_INVALID;            // (the caller asks qc_fock_can_speculate first)
        // an earlier ASYNCHRONOUS build's wait gave up (qc_fock_*_device return before their build has run; every call that waits on the
        // host has looked at the word itself, qc_join_check): its result was not complete, and this is the first call that can say so
        if (!event_join) { int jrc = qc_join_check(S); if (jrc != QC_OK) return jrc; }
        qc_stamp("sorted");
        // launches of a set of streams, interleaved (first launch of every stream of the set before any second one), then the
        // streams' markers of the device-side join
        auto issue = [&](const int *set, int nset) -> int {
            size_t longest = 0;
            for (int i = 0; i < nset; ++i) longest = std::max(longest, q[set[i]].size());
            for (size_t pos = 0; pos < longest; ++pos)
                for (int i = 0; i < nset; ++i) {
                    const int k = set[i];
                    if (pos >= q[k].size()) continue;
                    const int u = q[k][pos];
                    hipStream_t st = k == kmain ? S->stream : S->side[S->slot_side[side_slot(k)]];
                    if (pos == 0 && k != kmain && fork) QC_HIP_CHECK(hipStreamWaitEvent(st, S->ev_fork, 0));
                    if (pos == 0 && k != kmain && devfork) {
                        // (side chains in the order of their load, QC_FORK_STAGGER_US apart - default 6 us, the host's own issue rate)
                        static const double stagger_us = getenv("QC_FORK_STAGGER_US") ? atof(getenv("QC_FORK_STAGGER_US")) : 6.0;
                        const int delay = (int)(stagger_us * 100.0 * (double)i);
                        hipLaunchKernelGGL(qc_join_wait_kernel, dim3(1), dim3(64), 0, st, S->d_join + 1, fork_seq, S->h_join_timeout, S->wait_limit, delay, (unsigned long long *)nullptr);
                    }
                    if (ev && per_unit) QC_HIP_CHECK(hipEventRecord(ev[2 + 2 * u], st));
                    int rc = launch_segments(S, u, segs_of(units[u]), st, a);
                    if (rc != QC_OK) return rc;
                    qc_stamp("launch");
                    if (ev && per_unit) QC_HIP_CHECK(hipEventRecord(ev[3 + 2 * u], st));
                }
            if (!event_join) {
                // (test hook QC_JOIN_FAULT: the first side stream's marker is left out - a join that can never complete, as if a launch on
                // that stream had died: the waiting kernel runs into its limit and the call that waits for this build must fail)
                const bool fault = getenv("QC_JOIN_FAULT") != nullptr;
                for (int i = 0; i < nset; ++i)
                    if (set[i] != kmain && !q[set[i]].empty() && !(fault && i == (set[0] == kmain ? 1 : 0))) hipLaunchKernelGGL(qc_join_mark_kernel, dim3(1), dim3(64), 0, S->side[S->slot_side[side_slot(set[i])]], S->d_join);
                if (hipGetLastError() != hipSuccess) return QC_ERR_HIP;
                qc_stamp("markers");
            }
            return QC_OK;
        };
        int nused = 0;
        unsigned nside = 0;
        for (int k = 0; k < QC_NSTREAMS; ++k) if (!q[k].empty()) { ++nused; if (k != kmain) ++nside; }
        // Measured (alternating runs on one box): H2O/cc-pVTZ builds 0.227 ms with three helpers against 0.204 ms issued by the caller
        // alone (means of six) - its 13 launches of 30-70 us each do better when they start 6-7 us apart than all at once; benzene/cc-pVDZ
        // iterations 1.646 against 1.641 ms (means of four; 1.470 against 1.494 ms of build in a first pair of runs), and on another box two
        // of six runs with helpers showed 0.1-0.16 ms per pass outside the build and linear-algebra intervals.  Nothing gained: the
        // helpers are OFF unless QC_ISSUE_THREADS asks for them (the code stays for larger systems, where the issue time of dozens of
        // long launches could matter).
        const int nhelp_env = std::min(S->issue_threads, QC_NSTREAMS - 1);
        float serial_ms = 0.f;
        for (float x : S->unit_ms) serial_ms += x;
        const int nhelp_want = nhelp_env >= 0 ? nhelp_env : 0;
        (void)serial_ms;
        const int nhelp = (event_join || devfork || (ev && per_unit) || nused < 3) ? 0 : std::min(nhelp_want, nused - 1);
        if (nhelp <= 0) {
            int used_set[QC_NSTREAMS], n = 0;
            for (int k : ks) if (!q[k].empty()) used_set[n++] = k;
            int rc = issue(used_set, n);
            if (rc != QC_OK) return rc;
        } else {
            if (!S->issue_pool || (int)S->issue_pool->w.size() < nhelp) { delete S->issue_pool; S->issue_pool = new QcIssuePool(std::max(nhelp, nhelp_want), S->device); }
            QcIssuePool &P = *S->issue_pool;
            ++P.gen;
            int sets[QC_NSTREAMS][QC_NSTREAMS], nset[QC_NSTREAMS] = {};
            {   // side streams in the order of their load, dealt round the helpers
                int h = 0;
                for (int k : ks) if (k != kmain && !q[k].empty()) { sets[h][nset[h]++] = k; h = (h + 1) % nhelp; }
            }
            for (int h = 0; h < nhelp; ++h) P.start(h, [&, h]() -> int { return issue(sets[h], nset[h]); });
            int rc = kmain >= 0 ? issue(&kmain, 1) : QC_OK;
            for (int h = 0; h < nhelp; ++h) { const int r = P.wait(h); if (rc == QC_OK) rc = r; }       // (every helper is waited for, whatever happened)
            if (rc != QC_OK) {
                // some markers of this build may be out, others not: the counter and the host's target meet again before anything else waits
                unsigned c = 0;
                if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&c, S->d_join, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess) S->join_target = c;
                return rc;
            }
        }
        if (event_join) {
            for (int k = 0; k < QC_NSTREAMS; ++k) {
                if (q[k].empty() || k == kmain) continue;
                QC_HIP_CHECK(hipEventRecord(S->ev_join[k], S->side[S->slot_side[side_slot(k)]]));
                QC_HIP_CHECK(hipStreamWaitEvent(S->stream, S->ev_join[k], 0));
            }
        } else if (nside) {
            S->join_target += nside;
            if (fold_joins) S->fold_join_pending = true;
            else hipLaunchKernelGGL(qc_join_wait_kernel, dim3(1), dim3(64), 0, S->stream, S->d_join, S->join_target, S->h_join_timeout, S->wait_limit, -1, S->tl_cur ? S->tl_cur + QC_TL_W * QC_NUNITS : nullptr);
            if (hipGetLastError() != hipSuccess) return QC_ERR_HIP;
            gate.waits = true;
            qc_stamp("wait kernel");
        }
        if (ev) QC_HIP_CHECK(hipEventRecord(ev[1], S->stream));
        static const bool join_check = getenv("QC_JOIN_CHECK") != nullptr;       // (diagnostic: after every build the device counter is the host's target)
        if (join_check && !event_join) {
            unsigned c = 0;
            QC_HIP_CHECK(hipDeviceSynchronize());
            QC_HIP_CHECK(hipMemcpy(&c, S->d_join, sizeof(c), hipMemcpyDeviceToHost));
            if (c != S->join_target) { fprintf(stderr, "qchem_hip: join counter %u, target %u (%u side streams)\n", c, S->join_target, nside); return QC_ERR_HIP; }
        }
        return QC_OK;
    };
    // Kernels that overlap stretch each other by class-dependent factors (the bra-major launches 1.8x next to the one-wave-per-SIMD
    // launches, those hardly at all), which the durations alone do not show: three concurrent builds with events around every launch,
    // each proposing the longest-first assignment of the durations seen INSIDE it.  These proposals are the first trials of the online
    // search and are made when it starts (its first instalment): a handle that runs one SCF does not pay for them.
    auto seed_proposals = [&]() -> int {
        const size_t gb = ((fa.fxs ? fa.fx_lo : 0) + (size_t)a.nrep * a.rep_stride) * sizeof(double);
        const std::vector<int> keep = S->unit_stream;
        EventList evl;
        if (evl.create(2 + 2 * units.size()) != QC_OK) return QC_ERR_HIP;
        std::vector<hipEvent_t> &ev = evl.ev;
        int rc = QC_OK;
        for (int round = 0; round < 3 && rc == QC_OK; ++round) {
            if (fa.G0) QC_HIP_CHECK(hipMemsetAsync(fa.G0, 0, gb, S->stream));
            if ((rc = launch_concurrent(ev.data(), true, 0)) != QC_OK) break;
            QC_HIP_CHECK(hipEventSynchronize(ev[1]));
            if ((rc = qc_join_check(S)) != QC_OK) break;
            std::vector<float> dur(units.size(), 0.f);
            for (size_t u = 0; u < units.size(); ++u)
                if (!units[u].empty()) { QC_HIP_CHECK(hipEventSynchronize(ev[3 + 2 * u])); QC_HIP_CHECK(hipEventElapsedTime(&dur[u], ev[2 + 2 * u], ev[3 + 2 * u])); }
            lpt(dur, QC_NSTREAMS, round == 1 ? 0.015f : 0.f);
            if (std::find(S->on.cands.begin(), S->on.cands.end(), S->unit_stream) == S->on.cands.end() && S->unit_stream != keep) S->on.cands.push_back(S->unit_stream);
            S->on.spent += 1;
        }
        S->unit_stream = keep; S->unit_weight = S->unit_ms;
        return rc;
    };
    if (S->unit_ms.size() != units.size()) {
        // First build of a shard.  No tuner run (round 4): the launches are timed alone once (two serial passes: the first pays the code
        // upload), placed longest-first on the dispatch lanes, and the assignment is refined ONLINE from the build times the SCF passes
        // report anyway (qc_fock_feedback) - a neighbouring assignment is tried for a few passes and kept when it is faster.  The offline
        // tuner of rounds 1-3 (25 + up to 256 extra builds and a local search, 55 ms for H2O/cc-pVTZ) cost ten times the 15-pass SCF it
        // served and won 6 % of its builds; a process that has seen the same work lists before starts from what it learned (qc_assign_cache).
        S->unit_ms.assign(units.size(), 0.f);
        struct Untuned { qc_system *S; bool keep = false; ~Untuned() { if (!keep) { S->unit_ms.clear(); S->unit_stream.clear(); } } } untuned{S};
        // (from the first replica of the hi plane to the last replica in use of the lo plane: the planes keep the layout of QC_NREP replicas
        // whatever the number in use)
        const size_t gbytes = ((fa.fxs ? fa.fx_lo : 0) + (size_t)a.nrep * a.rep_stride) * sizeof(double);
        // (the warm-up pass only where this process has not launched these units before: their first launches pay the code upload)
        static std::atomic<unsigned long long> units_warm{0};
        unsigned long long mine = 0;
        for (size_t u = 0; u < units.size() && u < 62; ++u) if (!units[u].empty()) mine |= 1ull << u;
        if (S->merge_bm) mine |= 1ull << 62;                 // (the merged launches are kernels of their own)
        if (S->merge_t1) mine |= 1ull << 63;
        int rc = QC_OK;
        if ((units_warm.load(std::memory_order_acquire) & mine) != mine) rc = qc_launch_fock_classes(S, fa, nullptr, S->unit_ms.data());
        if (rc == QC_OK) rc = qc_launch_fock_classes(S, fa, nullptr, S->unit_ms.data());   // serial, timed
        if (rc == QC_OK) units_warm.fetch_or(mine, std::memory_order_acq_rel);
        if (rc != QC_OK) return rc;
        static const int fixed_w = getenv("QC_TUNE_FIXED") ? atoi(getenv("QC_TUNE_FIXED")) : 0;      // (experiment switch: lanes used, no online search)
        lpt(S->unit_ms, fixed_w >= 1 && fixed_w <= QC_NSTREAMS ? fixed_w : QC_NSTREAMS, 0.015f);
        S->tune_count += 1; S->assign_gen += 1;
        const bool no_search = fixed_w >= 1 || getenv("QC_TUNE_OFF") != nullptr;
        qc_online_reset(S, no_search);
        qc_assign_cache_lookup(S);
        if (fa.G0) QC_HIP_CHECK(hipMemsetAsync(fa.G0, 0, gbytes, S->stream));
        nofork = false;                                  // the side streams must see that memset (and the timing passes) finished
        untuned.keep = true;
    }
    // an instalment of the assignment search (see above): not in a speculative build, not in the builds of a profiling call
    S->on.builds += 1;
    if (!S->on.frozen && fa.fork_seq == 0 && fa.G0 && S->on.builds >= QC_SEARCH_FIRST_BUILD && S->on.spent + 2 * QC_SEARCH_CHUNK <= S->on.builds) {
        const size_t gbytes = ((fa.fxs ? fa.fx_lo : 0) + (size_t)a.nrep * a.rep_stride) * sizeof(double);
        EventList evl;
        if (evl.create(2) != QC_OK) return QC_ERR_HIP;
        qc_system::QcOnline &o = S->on;
        static const bool dbg = getenv("QC_TUNE_DEBUG") != nullptr;
        int rc = QC_OK;
        auto measure = [&](const std::vector<int> &assign, float &t) -> int {
            S->unit_stream = assign;
            t = 1e30f;
            for (int rep = 0; rep < 2; ++rep) {
                QC_HIP_CHECK(hipMemsetAsync(fa.G0, 0, gbytes, S->stream));
                int r = launch_concurrent(evl.ev.data(), false, 0);
                if (r != QC_OK) return r;
                QC_HIP_CHECK(hipEventSynchronize(evl.ev[1]));
                if ((r = qc_join_check(S)) != QC_OK) return r;
                float x = 0.f;
                QC_HIP_CHECK(hipEventElapsedTime(&x, evl.ev[0], evl.ev[1]));
                t = std::min(t, x);
                o.spent += 1;
            }
            return QC_OK;
        };
        // The neighbourhood of the current best, in full and in a fixed order: every launch moved to every other lane, every pair of
        // launches on different lanes swapped - launches of the most loaded lane first (they are the ones whose move can shorten the
        // build).  The search ends when a whole sweep has found nothing (a local optimum of the FULL neighbourhood: ten random
        // neighbours in a row, the rule before, left most of it unseen and ended anywhere between 0.172 and 0.192 ms on H2O/cc-pVTZ).
        auto neighbours = [&]() {
            o.nb.clear(); o.nb_pos = 0;
            std::vector<int> act;
            for (size_t u = 0; u < units.size(); ++u) if (!units[u].empty()) act.push_back((int)u);
            const int nl = std::min(QC_NSTREAMS, S->nlanes);
            if (act.size() < 2 || nl < 2) return;
            float load[QC_NSTREAMS] = {};
            int cnt[QC_NSTREAMS] = {}, maxrank[QC_NSTREAMS] = {};
            for (int u : act) { const int k = o.best[u] & 7; load[k] += S->unit_ms[u]; cnt[k] += 1; maxrank[k] = std::max(maxrank[k], o.best[u] >> 3); }
            std::stable_sort(act.begin(), act.end(), [&](int x, int y) { return load[o.best[x] & 7] > load[o.best[y] & 7]; });
            for (int u : act)
                for (int k = 0; k < nl; ++k)
                    if (k != (o.best[u] & 7)) { std::vector<int> t = o.best; t[u] = k; o.nb.push_back(std::move(t)); }
            for (size_t i = 0; i < act.size(); ++i)
                for (size_t j = i + 1; j < act.size(); ++j)
                    if ((o.best[act[i]] & 7) != (o.best[act[j]] & 7)) {
                        std::vector<int> t = o.best;
                        const int ki = t[act[i]] & 7, kj = t[act[j]] & 7;
                        t[act[i]] = kj; t[act[j]] = ki;
                        o.nb.push_back(std::move(t));
                    }
            // order inside a lane: a launch sent to the back of its lane (the heavier-first rule is not always the better one: which kernel
            // of a chain meets which kernels of the other chains decides how far they stretch each other)
            for (int u : act) {
                const int k = o.best[u] & 7;
                if (cnt[k] >= 2 && maxrank[k] < 14) { std::vector<int> t = o.best; t[u] = k | ((maxrank[k] + 1) << 3); o.nb.push_back(std::move(t)); }
            }
        };
        auto propose = [&]() -> bool {
            if (!o.cands.empty()) { o.trial = o.cands.back(); o.cands.pop_back(); return true; }      // (the proposals of the first build)
            if (o.nb.empty() && o.nb_pos == 0) neighbours();
            while (o.nb_pos < o.nb.size()) {
                o.trial = o.nb[o.nb_pos++];
                bool seen = false;
                for (const auto &e : o.tried) if (e == o.trial) { seen = true; break; }
                if (!seen) { o.tried.push_back(o.trial); return true; }
            }
            return false;
        };
        if (o.best.empty()) o.best = S->unit_stream;
        if (!o.seeded) { o.seeded = true; if ((rc = seed_proposals()) != QC_OK) return rc; }
        auto note_top = [&](const std::vector<int> &a, float t) {
            for (auto &e : o.top) if (e.second == a) { e.first = std::min(e.first, t); return; }
            o.top.push_back({t, a});
            std::sort(o.top.begin(), o.top.end(), [](const std::pair<float, std::vector<int>> &x, const std::pair<float, std::vector<int>> &y) { return x.first < y.first; });
            if (o.top.size() > 3) o.top.resize(3);
        };
        float tb = 0.f;
        if (o.base_ms <= 0.0) { if ((rc = measure(o.best, tb)) != QC_OK) return rc; o.base_ms = tb; note_top(o.best, tb); }
        // A whole sweep without a gain is a local optimum of single moves and swaps - and those lie 0.166 to 0.195 ms apart on H2O/cc-pVTZ,
        // process to process.  The search then starts again (QC_SEARCH_KICKS times) from the best assignment known with two random
        // cross-lane swaps applied - a step no sweep can take - and descends from there; the best three of everything measured go to
        // the finals as before.
        auto kick = [&]() -> bool {
            static const int max_kicks = getenv("QC_SEARCH_KICKS") ? atoi(getenv("QC_SEARCH_KICKS")) : QC_SEARCH_KICKS;
            if (o.kicks >= max_kicks || o.top.empty()) return false;
            std::vector<int> act;
            for (size_t u = 0; u < units.size(); ++u) if (!units[u].empty()) act.push_back((int)u);
            if (act.size() < 4) return false;
            auto rnd = [&]() { o.rng ^= o.rng << 13; o.rng ^= o.rng >> 17; o.rng ^= o.rng << 5; return o.rng; };
            for (int attempt = 0; attempt < 32; ++attempt) {
                std::vector<int> t = o.top[0].second;
                for (int &x : t) x &= 7;
                for (int rep = 0; rep < 2; ++rep)
                    for (int tries = 0; tries < 16; ++tries) {
                        const int i = act[rnd() % act.size()], j = act[rnd() % act.size()];
                        if (t[i] != t[j]) { std::swap(t[i], t[j]); break; }
                    }
                bool seen = false;
                for (const auto &e : o.tried) if (e == t) { seen = true; break; }
                if (seen) continue;
                o.tried.push_back(t);
                o.best = t; o.nb.clear(); o.nb_pos = 0; o.kicks += 1;
                return true;
            }
            return false;
        };
        for (int k = 0; k < QC_SEARCH_CHUNK && !o.frozen; ++k) {
            if (!propose()) {
                if (!kick()) { o.frozen = true; break; }
                float tk = 0.f;
                if ((rc = measure(o.best, tk)) != QC_OK) return rc;
                o.trials += 1; o.base_ms = tk;
                note_top(o.best, tk);
                if (dbg) fprintf(stderr, "[tune] trial %d: restart %d of the search from a perturbed best: %.4f ms (best known %.4f)\n", o.trials, o.kicks, tk, o.top[0].first);
                if (o.trials >= QC_SEARCH_TRIALS) o.frozen = true;
                continue;
            }
            const bool seeded = !o.cands.empty();
            float t = 0.f;
            if ((rc = measure(o.trial, t)) != QC_OK) return rc;
            o.trials += 1;
            if (dbg) fprintf(stderr, "[tune] trial %d (build %ld of the handle): %.4f ms against %.4f ms - %s\n", o.trials, (long)o.builds, t, o.base_ms, t < 0.985 * o.base_ms ? "kept" : "dropped");
            if (t < 0.985 * o.base_ms) { o.best = o.trial; o.base_ms = t; o.rejects = 0; o.nb.clear(); o.nb_pos = 0; }     // (a new neighbourhood)
            else if (!seeded) o.rejects += 1;
            note_top(o.trial, t);
            if (o.trials >= QC_SEARCH_TRIALS) o.frozen = true;
        }
        S->unit_stream = o.top.empty() ? o.best : o.top[0].second;       // (the best known - after a restart `best` is where the search stands)
        S->assign_gen += 1; S->tune_count += 1;             // (this build carries extra builds: not a timing sample)
        if (o.frozen) {
            // finals inside SCF passes (qc_fock_feedback) - not for multi-rank handles, whose passes report nothing
            o.fin_sum.assign(o.top.size(), 0.0); o.fin_n.assign(o.top.size(), 0); o.fin_cur = 0;
            if (S->comm || o.top.size() < 2) o.settled = true;
            else { S->unit_stream = o.top[0].second; S->cand_skip = true; }
            qc_assign_cache_store(S);
            if (dbg) {
                if (!o.top.empty()) { o.best = o.top[0].second; o.base_ms = o.top[0].first; }
                fprintf(stderr, "[tune] search ends after %d trials, %d restarts (%ld extra builds): %.4f ms; lanes:", o.trials, o.kicks, (long)o.spent, o.base_ms);
                for (int k = 0; k < QC_NSTREAMS; ++k) {
                    bool any = false;
                    for (int rk = 0; rk < 16; ++rk)
                        for (size_t u = 0; u < units.size(); ++u) if (!units[u].empty() && (o.best[u] & 7) == k && (o.best[u] >> 3) == rk) { fprintf(stderr, "%s u%zu(%.0f)%s", any ? "" : " [", u, S->unit_ms[u] * 1e3, rk ? "'" : ""); any = true; }
                    if (any) fprintf(stderr, " ]");
                }
                fprintf(stderr, "\n");
            }
        }
        QC_HIP_CHECK(hipMemsetAsync(fa.G0, 0, gbytes, S->stream));
        nofork = false;
    }
    return launch_concurrent(nullptr, false, fa.fork_seq, fa.fold_joins && !fa.fork_seq);
}

// ---- Refinement of the stream assignment, paid for by use.  A neighbouring assignment (one launch moved to another lane, two launches
// of different lanes swapped; first of all the proposals of the first build) is measured by two extra builds, back to back, and kept when
// the better of them beats the current best by 1.5 % - the local search of rounds 2-3.  What changed in round 4 is WHEN it runs: never in
// a handle's first builds (the offline tuner of rounds 1-3 spent 55 ms - ten times the 15-pass SCF of H2O/cc-pVTZ it served - to win 6 %
// of its builds), but in small instalments once the handle has shown that it lives long: from its 24th build on, a build may spend on
// trials as many extra builds as the handle has done useful ones so far, minus what was spent already.  A handle that does one SCF pays
// nothing; one that runs hundreds of builds (geometry loops, benchmarks) converges to the searched assignment at a bounded overhead
// and then stops (a whole sweep of the neighbourhood without a gain, or QC_SEARCH_TRIALS trials); the result goes to a process-wide cache keyed by
// the shape of the work lists.  The stream assignment does not change results (integer accumulation), only time.
// (measurement hook: end the search here and now with what it has found - a harness that is about to time builds calls it so that no
// instalment falls into its timed region)
void qc_assignment_freeze(qc_system *S) {
    qc_system::QcOnline &o = S->on;
    if (o.settled) return;
    // (the best known: after a restart `best` is only where the search stands, and while the finals run the assignment in use is whichever
    // of the top three is being sampled)
    const std::vector<int> &keep = o.top.empty() ? o.best : o.top[0].second;
    if (!keep.empty() && S->unit_stream != keep) { S->unit_stream = keep; S->assign_gen += 1; }
    o.frozen = true; o.settled = true;
}
void qc_online_reset(qc_system *S, bool frozen) {
    S->on = qc_system::QcOnline{};
    S->on.frozen = frozen; S->on.settled = frozen;
    S->on.best = S->unit_stream;
    S->on.rng = 2463534242u;
}
static uint64_t qc_assign_key(const qc_system *S) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    mix((uint64_t)S->nlanes); mix((uint64_t)S->nbasis); mix((uint64_t)S->nranks); mix((uint64_t)S->rank); mix((uint64_t)S->accum_fx);
    for (const auto &c : S->classes) { mix(((uint64_t)c.LAB << 40) | ((uint64_t)c.LCD << 32) | (uint64_t)(c.bm ? 1 : 0)); mix((uint64_t)c.slots.size()); mix((uint64_t)c.bundles.size()); mix((uint64_t)c.prim_quartets); }
    return h;
}
struct QcAssignCache { std::mutex mu; std::vector<std::pair<uint64_t, std::pair<std::vector<int>, bool>>> e; };
static QcAssignCache &qc_assign_cache() { static QcAssignCache *c = new QcAssignCache(); return *c; }     // (never destroyed, as the gates)
void qc_assign_cache_lookup(qc_system *S) {
    if (getenv("QC_NO_ASSIGN_CACHE")) return;
    const uint64_t key = qc_assign_key(S);
    QcAssignCache &C = qc_assign_cache();
    std::lock_guard<std::mutex> lk(C.mu);
    for (const auto &kv : C.e)
        if (kv.first == key && kv.second.first.size() == S->unit_stream.size()) {
            S->unit_stream = kv.second.first; S->on.best = S->unit_stream; S->assign_gen += 1;
            if (kv.second.second) { S->on.frozen = true; S->on.settled = true; }
            return;
        }
}
static void qc_assign_cache_store(const qc_system *S) {
    if (getenv("QC_NO_ASSIGN_CACHE")) return;
    const uint64_t key = qc_assign_key(S);
    QcAssignCache &C = qc_assign_cache();
    std::lock_guard<std::mutex> lk(C.mu);
    for (auto &kv : C.e) if (kv.first == key) { kv.second = {S->on.best, S->on.settled}; return; }
    if (C.e.size() < 64) C.e.push_back({key, {S->on.best, S->on.settled}});
}
// A build may be issued speculatively (device-side fork, scf_iterate) when nothing of it needs the host: the launches have been timed
// (that first build waits for its serial passes), the side streams are joined on the device, and the accumulation is the fixed-point one
// whose closing fold leaves the planes clean (the f64 mode starts with a memset the side streams would have to wait for).
bool qc_fock_can_speculate(const qc_system *S) {
    return S->device_ready && !S->join_by_events && S->accum_fx && S->launch_plan && !S->unit_ms.empty() &&
           S->unit_ms.size() == S->launch_plan->units.size();
}
// (the SCF passes report their build times: kept as the handle's running mean - the search itself measures its own builds)
void qc_fock_feedback(qc_system *S, float build_ms, unsigned gen) {
    qc_system::QcOnline &o = S->on;
    if (gen != S->assign_gen) return;
    o.seen_sum += build_ms; o.seen_n += 1;
    if (!o.frozen || o.settled) return;
    if (o.top.size() < 2 || o.fin_cur >= (int)o.top.size()) { o.settled = true; return; }
    if (S->cand_skip) { S->cand_skip = false; return; }          // (first build under this finalist)
    o.fin_sum[o.fin_cur] += build_ms; o.fin_n[o.fin_cur] += 1;
    if (o.fin_n[o.fin_cur] < 3) return;
    auto switch_to = [&](const std::vector<int> &a) { if (S->unit_stream != a) { S->unit_stream = a; S->assign_gen += 1; S->cand_skip = true; } };
    if (o.fin_cur + 1 < (int)o.top.size()) { o.fin_cur += 1; switch_to(o.top[o.fin_cur].second); return; }
    size_t b = 0;
    for (size_t i = 1; i < o.top.size(); ++i) if (o.fin_sum[i] / o.fin_n[i] < o.fin_sum[b] / o.fin_n[b]) b = i;
    static const bool dbg = getenv("QC_TUNE_DEBUG") != nullptr;
    if (dbg) { fprintf(stderr, "[tune] finals inside SCF passes:"); for (size_t i = 0; i < o.top.size(); ++i) fprintf(stderr, " %.4f (%.4f back to back)", o.fin_sum[i] / o.fin_n[i], o.top[i].first); fprintf(stderr, " -> %zu\n", b); }
    o.best = o.top[b].second;
    switch_to(o.best);
    o.settled = true;
    qc_assign_cache_store(S);
}

// temporary device buffer of the two set-up passes below: released on every return path
template <class T> struct QcTmpDev {
    T *p = nullptr;
    ~QcTmpDev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
};

// Schwarz pass (the reference's own TODO at uhf.rs:49-50): the (P|P) quartet of every stored pair through the
// class kernels in their `schwarz_out` mode - unsplit slots / one-ket bundles, serial launches, once per geometry.
int qc_schwarz_device(qc_system *S) {
    const size_t np = S->pairs.size();
    // (the classes' launches are serial on the handle's stream and independent of the host; their temporary lists are packed into ONE
    // device buffer - one allocation, one copy, one wait for the whole pass: an allocation, two synchronous copies and a wait per class
    // made it 4 ms for H2O/cc-pVTZ, most of a cold handle's set-up; 3.6 ms still with one wait but 35 allocations and 50 copies)
    QcTmpDev<double> dq;
    QC_HIP_CHECK(dq.alloc(np));
    double *const d_q = dq.p;
    QC_HIP_CHECK(hipMemsetAsync(d_q, 0, np * sizeof(double), S->stream));
    QcFockArgs fa{};
    fa.schwarz_out = d_q;
    const QcKernelArgs a = base_args(S, fa);
    std::vector<QcSlot> slots;
    std::vector<QcBundle> bundles; std::vector<int> ketlist;
    std::vector<QcTask> diag;
    std::vector<unsigned char> blob;                               // every class's lists, 256-byte aligned
    auto put = [&](const void *src, size_t bytes) -> size_t {
        const size_t off = (blob.size() + 255) & ~(size_t)255;
        blob.resize(off + bytes);
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    struct Job { const QcClass *c; int kind; size_t off_a, off_b; int count, lds; QcClass col; };   // kind 0: slots, 1: bundles, 2: p.p kets through the column kernels
    std::vector<Job> jobs;
    jobs.reserve(S->classes.size());
    for (const auto &c : S->classes) {
        diag.clear();
        for (const auto &t : c.tasks) if (t.bra == t.ket) diag.push_back(t);
        if (diag.empty()) continue;
        Job j{&c, 0, 0, 0, 0, 0, QcClass{}};
        if (c.bm && c.LCD == 2) {   // p.p-ket bra-major class: the column kernels have the Schwarz mode
            j.kind = 2;
            j.col.LAB = c.LAB; j.col.LCD = c.LCD; j.col.LGC = c.col_lgc; j.col.slot_words = c.col_slot_words; j.col.lds_bytes = c.col_lds_bytes;
            qc_make_slots(S, diag, 0, false, slots);
            j.off_a = put(slots.data(), slots.size() * sizeof(QcSlot)); j.count = (int)slots.size();
        } else if (c.bm) {
            j.kind = 1;
            const bool packed = qc_make_bundles(S, diag, 0, bundles, ketlist);
            std::vector<QcBundleDev> hb; std::vector<QcKetUnit> hu;
            qc_bm_device_lists(S, c.LCD, bundles, ketlist, packed, hb, hu);
            j.off_a = put(hb.data(), hb.size() * sizeof(QcBundleDev)); j.off_b = put(hu.data(), hu.size() * sizeof(QcKetUnit));
            j.count = (int)bundles.size();
            int mx = 0;
            for (const auto &t : diag) mx = std::max(mx, qc_bm_wave_words(c.LAB, S->pairs[t.bra].na * S->pairs[t.bra].nb, S->pairs[t.ket].na * S->pairs[t.ket].nb));
            j.lds = mx * 8;
        } else {
            qc_make_slots(S, diag, 0, false, slots);
            j.off_a = put(slots.data(), slots.size() * sizeof(QcSlot)); j.count = (int)slots.size();
        }
        jobs.push_back(std::move(j));
    }
    QcTmpDev<unsigned char> dblob;
    if (!blob.empty()) {
        QC_HIP_CHECK(dblob.alloc(blob.size()));
        QC_HIP_CHECK(hipMemcpyAsync(dblob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, S->stream));
    }
    // The launches are independent (each writes its own pairs' entries) and most of them are a few waves working through one long
    // unsplit slot each (an s.s pair of eight primitives: 4096 primitive quartets in one lane group): their durations add up on one
    // stream - 3.5 ms for the 35 classes of H2O/cc-pVTZ - so they go round the dispatch lanes, heaviest classes first.
    const int nl = std::max(1, std::min(S->nlanes, QC_NSTREAMS));
    if (nl > 1) QC_HIP_CHECK(hipStreamSynchronize(S->stream));    // (the side streams start behind the memset and the copy)
    std::stable_sort(jobs.begin(), jobs.end(), [](const Job &x, const Job &y) { return x.c->LAB + x.c->LCD > y.c->LAB + y.c->LCD; });
    int turn = 0;
    auto sync_all = [&]() -> hipError_t {
        hipError_t e = hipStreamSynchronize(S->stream);
        for (int k = 1; k < nl; ++k) { const hipError_t e2 = hipStreamSynchronize(S->side[S->slot_side[k]]); if (e == hipSuccess) e = e2; }
        return e;
    };
    for (const Job &j : jobs) {        // (`jobs` does not move any more: the column-kernel copy of a p.p-ket class is referred to by address)
        const int lane = turn++ % nl;
        hipStream_t st = lane == 0 ? S->stream : S->side[S->slot_side[lane]];
        int rc;
        if (j.kind == 1)
            rc = launch_segments(S, qc_unit_of(j.c->LAB, j.c->LCD, true), {Seg{j.c, nullptr, j.count, reinterpret_cast<const QcBundleDev *>(dblob.p + j.off_a),
                                                                              reinterpret_cast<const QcKetUnit *>(dblob.p + j.off_b), j.lds}}, st, a);
        else {
            const QcClass *cls = j.kind == 2 ? &j.col : j.c;
            rc = launch_segments(S, qc_unit_of(cls->LAB, cls->LCD, false), {Seg{cls, reinterpret_cast<const QcSlot *>(dblob.p + j.off_a), j.count}}, st, a);
        }
        if (rc != QC_OK) { (void)sync_all(); return rc; }
    }
    QC_HIP_CHECK(sync_all());
    S->pairQ.assign(np, 0.0);
    QC_HIP_CHECK(hipMemcpy(S->pairQ.data(), d_q, np * sizeof(double), hipMemcpyDeviceToHost));
    S->imax = 0.0;
    for (double q : S->pairQ) S->imax = std::max(S->imax, q * q);
    return QC_OK;
}

// molint::eri replacement for tests/plumbing: unsplit slots (every quartet complete in one slot) + plain stores
int qc_launch_eri_full(qc_system *S, double *d_out) {
    QcFockArgs fa{};
    fa.eri_out = d_out;
    const QcKernelArgs a = base_args(S, fa);
    std::vector<QcSlot> slots;
    std::vector<QcBundle> bundles; std::vector<int> ketlist;
    for (const auto &c : S->classes) {
        if (c.bm && c.LCD == 2) {   // p.p-ket bra-major class: the column kernels have the tensor mode
            QcClass cc;
            cc.LAB = c.LAB; cc.LCD = c.LCD; cc.LGC = c.col_lgc; cc.slot_words = c.col_slot_words; cc.lds_bytes = c.col_lds_bytes;
            qc_make_slots(S, c.tasks, 0, false, slots);
            if (slots.empty()) continue;
            QcTmpDev<QcSlot> d;
            QC_HIP_CHECK(d.alloc(slots.size()));
            QC_HIP_CHECK(hipMemcpyAsync(d.p, slots.data(), slots.size() * sizeof(QcSlot), hipMemcpyHostToDevice, S->stream));
            int rc = launch_segments(S, qc_unit_of(cc.LAB, cc.LCD, false), {Seg{&cc, d.p, (int)slots.size()}}, S->stream, a);
            QC_HIP_CHECK(hipStreamSynchronize(S->stream));
            if (rc != QC_OK) return rc;
            continue;
        }
        if (c.bm) {
            const bool packed = qc_make_bundles(S, c.tasks, 0, bundles, ketlist);
            if (bundles.empty()) continue;
            QcTmpDev<QcBundleDev> db; QcTmpDev<QcKetUnit> dk;
            std::vector<QcBundleDev> hb; std::vector<QcKetUnit> hu;
            qc_bm_device_lists(S, c.LCD, bundles, ketlist, packed, hb, hu);
            QC_HIP_CHECK(db.alloc(hb.size()));
            QC_HIP_CHECK(dk.alloc(hu.size()));
            QC_HIP_CHECK(hipMemcpyAsync(db.p, hb.data(), hb.size() * sizeof(QcBundleDev), hipMemcpyHostToDevice, S->stream));
            QC_HIP_CHECK(hipMemcpyAsync(dk.p, hu.data(), hu.size() * sizeof(QcKetUnit), hipMemcpyHostToDevice, S->stream));
            int mx = 0;
            for (const auto &t : c.tasks) mx = std::max(mx, qc_bm_wave_words(c.LAB, S->pairs[t.bra].na * S->pairs[t.bra].nb, S->pairs[t.ket].na * S->pairs[t.ket].nb));
            int rc = launch_segments(S, qc_unit_of(c.LAB, c.LCD, true), {Seg{&c, nullptr, (int)bundles.size(), db.p, dk.p, mx * 8}}, S->stream, a);
            QC_HIP_CHECK(hipStreamSynchronize(S->stream));
            if (rc != QC_OK) return rc;
            continue;
        }
        qc_make_slots(S, c.tasks, 0, false, slots);
        if (slots.empty()) continue;
        QcTmpDev<QcSlot> d;
        QC_HIP_CHECK(d.alloc(slots.size()));
        QC_HIP_CHECK(hipMemcpyAsync(d.p, slots.data(), slots.size() * sizeof(QcSlot), hipMemcpyHostToDevice, S->stream));
        int rc = launch_segments(S, qc_unit_of(c.LAB, c.LCD, false), {Seg{&c, d.p, (int)slots.size()}}, S->stream, a);
        QC_HIP_CHECK(hipStreamSynchronize(S->stream));
        if (rc != QC_OK) return rc;
    }
    return QC_OK;
}
