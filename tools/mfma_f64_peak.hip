// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on every CU (operands in registers).
// Calibrates the FP64 matrix peak that bench.py's roofline divides by (the CDNA4 guide has no FP64 row).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int threads) {
    int iters = 20000;
    int nblk = 256 * blocks_per_cu;
    double* out;
    hipMalloc(&out, sizeof(double) * nblk * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(nblk), dim3(threads), 0, 0, out, 100, 0.5, 0.25);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(nblk), dim3(threads), 0, 0, out, iters, 0.5, 0.25);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * (threads / 64) * nblk;
    printf("NACC=%d waves/CU=%d : %.3f ms, %.2f TFLOP/s\n", NACC, blocks_per_cu * threads / 64, ms,
           flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256);
    run<4>(2, 256); run<8>(2, 256); run<4>(4, 256);
    return 0;
}
