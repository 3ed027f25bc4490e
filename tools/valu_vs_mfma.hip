// Probe: how much does a co-resident wave's v_mfma_f64 stream slow down a dependent VALU chain on
// the same SIMD?  512-thread block (2 waves per SIMD): waves 0-3 run a dependent chain (f64 FMA or
// int adds), waves 4-7 either idle or stream f64 MFMAs.  Reports cycles per chain instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAIN /*0 f64 fma, 1 int add, 2 readlane+fma*/, int PARTNER /*0 idle, 1 mfma f64*/, int PRIO>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        double x = 1.0 + threadIdx.x * 1e-9, y = 1.0000001;
        int ix = threadIdx.x;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (CHAIN == 0) x = __builtin_fma(x, y, 1e-9);
                else if (CHAIN == 1) ix = ix * 3 + u;
                else {
                    int lo = __builtin_amdgcn_readlane(__double2loint(x), u);
                    int hi = __builtin_amdgcn_readlane(__double2hiint(x), u);
                    x = __builtin_fma(x, __hiloint2double(hi, lo), 1e-9);
                }
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
        out[blockIdx.x * 512 + threadIdx.x] = x + ix;
    } else if (PARTNER == 1) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = 0.5 + threadIdx.x * 1e-9, b = 0.25;
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    }
}

template <int CHAIN, int PARTNER, int PRIO>
void run(const char* name) {
    const int iters = 2000, nblk = 256;
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * nblk * 512);
    (void)hipMalloc(&cyc, 8 * nblk * 4);
    hipLaunchKernelGGL((k<CHAIN, PARTNER, PRIO>), dim3(nblk), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    static unsigned long long h[1024];
    (void)hipMemcpy(h, cyc, 8 * nblk * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nblk * 4; ++i) s += h[i];
    printf("%-40s %.2f cycles per chain instruction\n", name, s / (nblk * 4) / (iters * 16.0));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0, 0, 0>("f64 fma chain, partner idle");
    run<0, 1, 0>("f64 fma chain, partner mfma_f64");
    run<0, 1, 1>("f64 fma chain, partner mfma_f64, prio3");
    run<1, 0, 0>("int mad chain, partner idle");
    run<1, 1, 0>("int mad chain, partner mfma_f64");
    run<2, 0, 0>("readlane+fma chain, partner idle");
    run<2, 1, 0>("readlane+fma chain, partner mfma_f64");
    run<2, 1, 1>("readlane+fma chain, partner mfma, prio3");
    return 0;
}
