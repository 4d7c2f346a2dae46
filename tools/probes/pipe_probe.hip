// Which HIP streams share a dispatch pipe?  Kernel A: a grid of many one-wave workgroups that each hold a whole SIMD (64 KB LDS -> two per CU...
// here: 40 KB LDS, one wave, ~20 us of sleep) on stream i; kernel B: one tiny workgroup on stream j, launched right behind.  B's latency
// (host clock from launch to completion) tells whether j's queue had to wait for i's dispatch to drain.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void kA(long long ticks) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void kB(int *p) { if (threadIdx.x == 0) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 8;
    const int grid = argc > 2 ? atoi(argv[2]) : 8192;
    std::vector<hipStream_t> st(NS);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d; hipMalloc(&d, 4);
    hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // warm-up
    for (auto &s : st) { hipLaunchKernelGGL(kA, dim3(16), dim3(64), 40 * 1024, s, 100LL); hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s, d); }
    hipDeviceSynchronize();
    // A alone
    double t0 = now();
    hipLaunchKernelGGL(kA, dim3(grid), dim3(64), 40 * 1024, st[0], 2000LL);
    hipStreamSynchronize(st[0]);
    printf("A alone: %.1f us (grid %d x 20 us, 40 KB LDS -> 4 workgroups per CU, 1024 at a time)\n", now() - t0, grid);
    printf("latency of B on stream j while A dispatches on stream i (us):\n      ");
    for (int j = 0; j < NS; ++j) printf("  j=%d  ", j);
    printf("\n");
    for (int i = 0; i < NS; ++i) {
        printf("i=%d  ", i);
        for (int j = 0; j < NS; ++j) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(kA, dim3(grid), dim3(64), 40 * 1024, st[i], 2000LL);
                const double t1 = now();
                hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, st[j], d);
                hipStreamSynchronize(st[j]);
                const double t2 = now();
                hipDeviceSynchronize();
                best = t2 - t1 < best ? t2 - t1 : best;
            }
            printf("%7.1f", best);
        }
        printf("\n");
    }
    return 0;
}
