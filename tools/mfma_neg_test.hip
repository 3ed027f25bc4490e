// Probe: for v_mfma_f64_16x16x4_f64 the BLGP field acts as NEG[2:0] (negate A, B, C)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void k(double* out) {
    int l = threadIdx.x;
    double a = 1.0 + (l & 15), b = 2.0 + (l >> 4);
    d4 c = {10.0, 10.0, 10.0, 10.0};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP);
    if (l == 0) out[BLGP] = d[0];
}
int main() {
    double* out; double h[8];
    (void)hipMalloc(&out, 64);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out);
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    // D[0][0] = sum_k A[0][k] B[k][0] + C = 1*(2+3+4+5) + 10 = 24
    printf("blgp0 %.1f (expect 24)  blgp1 %.1f (negA: -4)  blgp2 %.1f (negB: -4)  blgp4 %.1f (negC: 4)\n", h[0], h[1], h[2], h[4]);
    return 0;
}
