// qc_peaks.hip - the two measured ceilings the roofline is quoted against next to the datasheet ones: what a
// register-resident v_fma_f64 loop and a 16-byte streaming copy reach on THIS device, measured by the harness outside its timed region.
#include "qc_internal.h"

namespace {

// 16 independent accumulator chains per lane, 8 waves per SIMD: nothing but v_fma_f64 between two loop branches
__global__ __launch_bounds__(256) void qc_peak_fma_kernel(int iters, double seed, double *__restrict__ out) {
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = seed * (threadIdx.x + 1) + k;
    const double x = 1.0 + 1e-9 * seed, y = 1e-9 * (seed + blockIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(a[k], x, y);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
    if (s == 12345.678) out[0] = s;              // keeps the chains alive, never true
}

// STREAM copy: 16-byte loads and stores, four in flight per lane, grid-stride
__global__ __launch_bounds__(256) void qc_peak_copy_kernel(size_t n4, const double2 *__restrict__ src, double2 *__restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const double2 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int qc_measure_peaks(double *fp64_tflops, double *hbm_copy_gbs) {
    if (!fp64_tflops || !hbm_copy_gbs) return QC_ERR_INVALID;
    if (qc_device_ready() != QC_OK) return QC_ERR_NO_DEVICE;
    hipStream_t st;
    QC_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    QC_HIP_CHECK(hipEventCreate(&e0)); QC_HIP_CHECK(hipEventCreate(&e1));
    int rc = QC_OK;
    double *buf = nullptr;
    const size_t bytes = (size_t)1 << 30;                       // 1 GiB source + 1 GiB destination: far beyond the 256 MB memory-side cache
    do {
        if (hipMalloc(&buf, 2 * bytes) != hipSuccess) { rc = QC_ERR_HIP; break; }
        if (hipMemsetAsync(buf, 0, 2 * bytes, st) != hipSuccess) { rc = QC_ERR_HIP; break; }
        float best_f = 1e30f, best_c = 1e30f;
        const int iters = 4096, grid = 256 * 8;                 // 8 workgroups of 4 waves per CU
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(qc_peak_fma_kernel, dim3(grid), dim3(256), 0, st, iters, 1.0, buf);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = QC_ERR_HIP; break; }
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best_f) best_f = ms;
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(qc_peak_copy_kernel, dim3(256 * 16), dim3(256), 0, st, bytes / 16, reinterpret_cast<const double2 *>(buf),
                               reinterpret_cast<double2 *>(buf + bytes / 8));
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = QC_ERR_HIP; break; }
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best_c) best_c = ms;
        }
        if (rc != QC_OK) break;
        *fp64_tflops = 2.0 * 64.0 * iters * (double)grid * 256.0 / (best_f * 1e-3) / 1e12;
        *hbm_copy_gbs = 2.0 * (double)bytes / (best_c * 1e-3) / 1e9;              // bytes read + bytes written
    } while (false);
    if (buf) (void)hipFree(buf);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return rc;
}
