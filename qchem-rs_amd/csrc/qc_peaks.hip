// qc_peaks.hip - the two measured ceilings the roofline is quoted against next to the datasheet ones: what a
// register-resident v_fma_f64 loop and a 16-byte streaming copy reach on THIS device, measured by the harness outside its timed region.
#include <cstdio>
#include <cstdlib>

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

// STREAM copy: 16-byte loads and stores, eight in flight per lane, grid-stride; NT = nontemporal stores (write-once data)
template <bool NT>
__global__ __launch_bounds__(256) void qc_peak_copy_kernel(size_t n4, const double2 *__restrict__ src, double2 *__restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NT) { __builtin_nontemporal_store(v[u].x, &dst[i + u * stride].x); __builtin_nontemporal_store(v[u].y, &dst[i + u * stride].y); }
            else dst[i + u * stride] = v[u];
        }
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
            for (int variant = 0; variant < 4; ++variant) {       // grid 8 / 16 workgroups per CU x plain / nontemporal stores: the best counts
                const int g = 256 * (variant & 1 ? 16 : 8);
                (void)hipEventRecord(e0, st);
                if (variant & 2) hipLaunchKernelGGL(qc_peak_copy_kernel<true>, dim3(g), dim3(256), 0, st, bytes / 16, reinterpret_cast<const double2 *>(buf),
                                                    reinterpret_cast<double2 *>(buf + bytes / 8));
                else hipLaunchKernelGGL(qc_peak_copy_kernel<false>, dim3(g), dim3(256), 0, st, bytes / 16, reinterpret_cast<const double2 *>(buf),
                                        reinterpret_cast<double2 *>(buf + bytes / 8));
                (void)hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) != hipSuccess) { rc = QC_ERR_HIP; break; }
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best_c) best_c = ms;
                if (getenv("QC_PEAKS_DEBUG")) fprintf(stderr, "[peaks] copy variant %d: %.3f ms = %.0f GB/s\n", variant, ms, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
            }
            if (rc != QC_OK) break;
        }
        if (rc != QC_OK) break;
        *fp64_tflops = 2.0 * 64.0 * iters * (double)grid * 256.0 / (best_f * 1e-3) / 1e12;
        *hbm_copy_gbs = 2.0 * (double)bytes / (best_c * 1e-3) / 1e9;              // bytes read + bytes written
    } while (false);
    if (buf) (void)hipFree(buf);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return rc;
}
