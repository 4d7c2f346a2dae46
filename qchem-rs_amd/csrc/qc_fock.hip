// qc_fock.hip - device set-up and the per-class launch loop of the direct-SCF Fock build.
#include <cstring>
#include <vector>

#include "qc_fock_kernel.h"

int qc_launch_class_lab0(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab1(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab2(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab3(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab4(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab5(int, int, size_t, hipStream_t, const QcKernelArgs &);
int qc_launch_class_lab6(int, int, size_t, hipStream_t, const QcKernelArgs &);

static int launch_class(int lab, int lcd, int grid, size_t lds, hipStream_t st, const QcKernelArgs &a) {
    switch (lab) {
        case 0: return qc_launch_class_lab0(lcd, grid, lds, st, a);
        case 1: return qc_launch_class_lab1(lcd, grid, lds, st, a);
        case 2: return qc_launch_class_lab2(lcd, grid, lds, st, a);
        case 3: return qc_launch_class_lab3(lcd, grid, lds, st, a);
        case 4: return qc_launch_class_lab4(lcd, grid, lds, st, a);
        case 5: return qc_launch_class_lab5(lcd, grid, lds, st, a);
        case 6: return qc_launch_class_lab6(lcd, grid, lds, st, a);
    }
    return QC_ERR_UNSUPPORTED;
}

int qc_device_ready(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QC_ERR_NO_DEVICE;
    return QC_OK;
}

static int upload_tasks(qc_system *S) {
    for (auto &c : S->classes) {
        if (c.d_tasks) { (void)hipFree(c.d_tasks); c.d_tasks = nullptr; }
        if (c.shard.empty()) continue;
        QC_HIP_CHECK(hipMalloc(&c.d_tasks, c.shard.size() * sizeof(QcTask)));
        QC_HIP_CHECK(hipMemcpy(c.d_tasks, c.shard.data(), c.shard.size() * sizeof(QcTask), hipMemcpyHostToDevice));
    }
    return QC_OK;
}

int qc_device_reshard(qc_system *S) {
    qc_build_shards(S);
    if (!S->device_ready) return QC_OK;
    return upload_tasks(S);
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
    const size_t nn = (size_t)S->nbasis * S->nbasis;
    QC_HIP_CHECK(hipMalloc(&S->d_pairdata, S->pairdata.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairdata, S->pairdata.data(), S->pairdata.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_pairs, S->pairs.size() * sizeof(QcPairDesc)));
    QC_HIP_CHECK(hipMemcpy(S->d_pairs, S->pairs.data(), S->pairs.size() * sizeof(QcPairDesc), hipMemcpyHostToDevice));
    std::vector<double> tab((size_t)QC_BOYS_NGRID * QC_BOYS_NORD);
    for (int k = 0; k < QC_BOYS_NGRID; ++k) qc_boys_host(QC_BOYS_NORD - 1, k * QC_BOYS_DX, &tab[(size_t)k * QC_BOYS_NORD]);
    QC_HIP_CHECK(hipMalloc(&S->d_boys, tab.size() * sizeof(double)));
    QC_HIP_CHECK(hipMemcpy(S->d_boys, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    QC_HIP_CHECK(hipMalloc(&S->d_D, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_G, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Gtmp, 2 * nn * sizeof(double)));
    QC_HIP_CHECK(hipMalloc(&S->d_Dj, nn * sizeof(double)));
    int rc = upload_tasks(S);
    if (rc != QC_OK) return rc;
    S->device_ready = true;
    return QC_OK;
}

void qc_device_free(qc_system *S) {
    for (auto &c : S->classes) if (c.d_tasks) { (void)hipFree(c.d_tasks); c.d_tasks = nullptr; }
    void *ptrs[] = {S->d_pairdata, S->d_pairs, S->d_boys, S->d_D, S->d_G, S->d_Gtmp, S->d_Dj};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    S->d_pairdata = nullptr; S->d_pairs = nullptr; S->d_boys = S->d_D = S->d_G = S->d_Gtmp = S->d_Dj = nullptr;
    if (S->own_stream && S->stream) (void)hipStreamDestroy(S->stream);
    S->stream = nullptr; S->own_stream = false; S->device_ready = false;
}

// One launch per non-empty class.  class_ms (optional): per-class time measured with hipEvents on the stream.
int qc_launch_fock_classes(qc_system *S, const QcFockArgs &fa, float *class_ms) {
    QcKernelArgs a;
    a.pairs = S->d_pairs; a.pairdata = S->d_pairdata; a.boys = S->d_boys; a.n = S->nbasis;
    a.Dj = fa.Dj; a.Dk0 = fa.Dk0; a.Dk1 = fa.Dk1; a.G0 = fa.G0; a.G1 = fa.G1; a.cK = fa.cK; a.eri_out = fa.eri_out;
    std::vector<hipEvent_t> ev;
    if (class_ms) {
        ev.resize(S->classes.size() + 1);
        for (auto &e : ev) QC_HIP_CHECK(hipEventCreate(&e));
        QC_HIP_CHECK(hipEventRecord(ev[0], S->stream));
    }
    for (size_t ci = 0; ci < S->classes.size(); ++ci) {
        const QcClass &c = S->classes[ci];
        if (!c.shard.empty()) {
            a.tasks = c.d_tasks; a.ntasks = (int)c.shard.size();
            const int grid = (int)std::min<size_t>(c.shard.size(), 256 * 16);
            int rc = launch_class(c.LAB, c.LCD, grid, (size_t)c.lds_bytes, S->stream, a);
            if (rc != QC_OK) return rc;
        }
        if (class_ms) QC_HIP_CHECK(hipEventRecord(ev[ci + 1], S->stream));
    }
    if (class_ms) {
        QC_HIP_CHECK(hipEventSynchronize(ev.back()));
        for (size_t ci = 0; ci < S->classes.size(); ++ci) QC_HIP_CHECK(hipEventElapsedTime(&class_ms[ci], ev[ci], ev[ci + 1]));
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    return QC_OK;
}
