// qc_fock.hip - device set-up and the per-class launch loop of the direct-SCF Fock build.
#include <cstring>
#include <vector>

#include "qc_fock_kernel.h"

int qc_launch_class_lab0(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab1(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab2(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab3(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab4(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab5(int, int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab6(int, int, int, size_t, hipStream_t, const QcKernelArgs &);

static int launch_class(int lab, int lcd, int lgc, int grid, size_t lds, hipStream_t st, const QcKernelArgs &a) {
    switch (lab) {
        case 0: return qc_launch_class_lab0(lcd, lgc, grid, lds, st, a);
        case 1: return qc_launch_class_lab1(lcd, lgc, grid, lds, st, a);
        case 2: return qc_launch_class_lab2(lcd, lgc, grid, lds, st, a);
        case 3: return qc_launch_class_lab3(lcd, lgc, grid, lds, st, a);
        case 4: return qc_launch_class_lab4(lcd, lgc, grid, lds, st, a);
        case 5: return qc_launch_class_lab5(lcd, lgc, grid, lds, st, a);
        case 6: return qc_launch_class_lab6(lcd, lgc, grid, lds, st, a);
    }
    return QC_ERR_UNSUPPORTED;
}

int qc_device_ready(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QC_ERR_NO_DEVICE;
    return QC_OK;
}

static int upload_slots(qc_system *S) {
    for (auto &c : S->classes) {
        if (c.d_slots) { (void)hipFree(c.d_slots); c.d_slots = nullptr; }
        if (c.slots.empty()) continue;
        QC_HIP_CHECK(hipMalloc(&c.d_slots, c.slots.size() * sizeof(QcSlot)));
        QC_HIP_CHECK(hipMemcpy(c.d_slots, c.slots.data(), c.slots.size() * sizeof(QcSlot), hipMemcpyHostToDevice));
    }
    return QC_OK;
}

int qc_device_reshard(qc_system *S) {
    qc_build_shards(S);
    if (!S->device_ready) return QC_OK;
    return upload_slots(S);
}

int qc_device_init(qc_system *S) {
    if (S->device_ready) return QC_OK;
    if (qc_device_ready() != QC_OK) return QC_ERR_NO_DEVICE;
    QC_HIP_CHECK(hipGetDevice(&S->device));
    hipDeviceProp_t prop;
    QC_HIP_CHECK(hipGetDeviceProperties(&prop, S->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "qchem_hip: device %d is %s, this library is built for gfx950 only\n", S->device, prop.gcnArchName);
        return QC_ERR_NO_DEVICE;
    }
    if (!S->stream) { QC_HIP_CHECK(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking)); S->own_stream = true; }
    for (int i = 0; i < QC_NSTREAMS; ++i) {
        QC_HIP_CHECK(hipStreamCreateWithFlags(&S->side[i], hipStreamNonBlocking));
        QC_HIP_CHECK(hipEventCreateWithFlags(&S->ev_join[i], hipEventDisableTiming));
    }
    QC_HIP_CHECK(hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming));
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    QC_HIP_CHECK(hipMalloc(&S->d_pairdata, S->pairdata.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairdata, S->pairdata.data(), S->pairdata.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_pairs, S->pairs.size() * sizeof(QcPairDesc)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairs, S->pairs.data(), S->pairs.size() * sizeof(QcPairDesc), hipMemcpyHostToDevice));
    std::vector<double> tab((size_t)(QC_LTOT + 1) * QC_BOYS_NGRID * 8), row(QC_BOYS_NORD);
    for (int k = 0; k < QC_BOYS_NGRID; ++k) {
        qc_boys_host(QC_BOYS_NORD - 1, k * QC_BOYS_DX, row.data());
        for (int L = 0; L <= QC_LTOT; ++L)
            for (int j = 0; j < 8; ++j) tab[((size_t)L * QC_BOYS_NGRID + k) * 8 + j] = row[L + j];
    }
    QC_HIP_CHECK(hipMalloc(&S->d_boys, tab.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_boys, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_D, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_G, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Gtmp, (size_t)QC_NREP * 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Dj, nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_flag, 4 * sizeof(int)));
    int rc = upload_slots(S);
    if (rc != QC_OK) return rc;
    S->device_ready = true;
    return QC_OK;
}

void qc_device_free(qc_system *S) {
    for (auto &c : S->classes) if (c.d_slots) { (void)hipFree(c.d_slots); c.d_slots = nullptr; }
    void *ptrs[] = {S->d_pairdata, S->d_pairs, S->d_boys, S->d_D, S->d_G, S->d_Gtmp, S->d_Dj, S->d_flag};
    S->d_flag = nullptr;
    for (void *p : ptrs) if (p) (void)hipFree(p);
    S->d_pairdata = nullptr; S->d_pairs = nullptr; S->d_boys = S->d_D = S->d_G = S->d_Gtmp = S->d_Dj = nullptr;
    for (int i = 0; i < QC_NSTREAMS; ++i) {
        if (S->side[i]) (void)hipStreamDestroy(S->side[i]);
        if (S->ev_join[i]) (void)hipEventDestroy(S->ev_join[i]);
        S->side[i] = nullptr; S->ev_join[i] = nullptr;
    }
    if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
    S->ev_fork = nullptr;
    if (S->own_stream && S->stream) (void)hipStreamDestroy(S->stream);
    S->stream = nullptr; S->own_stream = false; S->device_ready = false;
}

static QcKernelArgs base_args(qc_system *S, const QcFockArgs &fa) {
    QcKernelArgs a{};
    a.pairs = S->d_pairs; a.pairdata = S->d_pairdata; a.boys = S->d_boys; a.n = S->nbasis;
    a.Dj = fa.Dj; a.Dk0 = fa.Dk0; a.Dk1 = fa.Dk1; a.G0 = fa.G0; a.G1 = fa.G1; a.cK = fa.cK; a.eri_out = fa.eri_out;
    a.nrep = fa.nrep > 0 ? fa.nrep : 1; a.rep_stride = fa.rep_stride;
    return a;
}

static int launch_one(const QcClass &c, const QcSlot *d_slots, int nslots, hipStream_t st, QcKernelArgs a) {
    a.slots = d_slots; a.nslots = nslots; a.slot_words = c.slot_words;
    const int G = 64 >> c.LGC;
    const int waves = (nslots + G - 1) / G;
    const int grid = std::min(waves, 256 * 32);
    return launch_class(c.LAB, c.LCD, c.LGC, grid, (size_t)c.lds_bytes, st, a);
}

// One launch per non-empty class.  Normal mode: the launches are independent (they only meet in the atomically
// accumulated Gt replicas), so they are spread over QC_NSTREAMS side streams forked from / joined to the handle's
// stream - small classes fill the CUs the big ones leave idle.  Profiling mode (class_ms != nullptr): serial on the
// handle's stream with a hipEvent between consecutive launches.
int qc_launch_fock_classes(qc_system *S, const QcFockArgs &fa, float *class_ms) {
    const QcKernelArgs a = base_args(S, fa);
    if (class_ms) {
        std::vector<hipEvent_t> ev(S->classes.size() + 1);
        for (auto &e : ev) QC_HIP_CHECK(hipEventCreate(&e));
        QC_HIP_CHECK(hipEventRecord(ev[0], S->stream));
        for (size_t ci = 0; ci < S->classes.size(); ++ci) {
            const QcClass &c = S->classes[ci];
            if (!c.slots.empty()) { int rc = launch_one(c, c.d_slots, (int)c.slots.size(), S->stream, a); if (rc != QC_OK) return rc; }
            QC_HIP_CHECK(hipEventRecord(ev[ci + 1], S->stream));
        }
        QC_HIP_CHECK(hipEventSynchronize(ev.back()));
        for (size_t ci = 0; ci < S->classes.size(); ++ci) QC_HIP_CHECK(hipEventElapsedTime(&class_ms[ci], ev[ci], ev[ci + 1]));
        for (auto &e : ev) (void)hipEventDestroy(e);
        return QC_OK;
    }
    QC_HIP_CHECK(hipEventRecord(S->ev_fork, S->stream));
    for (int i = 0; i < QC_NSTREAMS; ++i) QC_HIP_CHECK(hipStreamWaitEvent(S->side[i], S->ev_fork, 0));
    // biggest classes first, round-robin over the side streams
    std::vector<int> order;
    for (size_t ci = 0; ci < S->classes.size(); ++ci) if (!S->classes[ci].slots.empty()) order.push_back((int)ci);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return S->classes[x].flops_alg > S->classes[y].flops_alg; });
    int k = 0;
    for (int ci : order) {
        const QcClass &c = S->classes[ci];
        int rc = launch_one(c, c.d_slots, (int)c.slots.size(), S->side[k % QC_NSTREAMS], a);
        if (rc != QC_OK) return rc;
        ++k;
    }
    for (int i = 0; i < QC_NSTREAMS; ++i) {
        QC_HIP_CHECK(hipEventRecord(S->ev_join[i], S->side[i]));
        QC_HIP_CHECK(hipStreamWaitEvent(S->stream, S->ev_join[i], 0));
    }
    return QC_OK;
}

// molint::eri replacement for tests/plumbing: unsplit slots (every quartet complete in one slot) + plain stores
int qc_launch_eri_full(qc_system *S, double *d_out) {
    QcFockArgs fa{};
    fa.eri_out = d_out;
    const QcKernelArgs a = base_args(S, fa);
    std::vector<QcSlot> slots;
    for (const auto &c : S->classes) {
        qc_make_slots(S, c.tasks, 0, slots);
        if (slots.empty()) continue;
        QcSlot *d = nullptr;
        QC_HIP_CHECK(hipMalloc(&d, slots.size() * sizeof(QcSlot)));
        QC_HIP_CHECK(hipMemcpyAsync(d, slots.data(), slots.size() * sizeof(QcSlot), hipMemcpyHostToDevice, S->stream));
        int rc = launch_one(c, d, (int)slots.size(), S->stream, a);
        QC_HIP_CHECK(hipStreamSynchronize(S->stream));
        (void)hipFree(d);
        if (rc != QC_OK) return rc;
    }
    return QC_OK;
}
