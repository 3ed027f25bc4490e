// Read-rate probe for the slab's access pattern: one wave per "matrix" reads ROWS rows of TILES*128 bytes with
// a row stride of ld doubles, four rows x 128 bytes per load instruction (lane = (row lq, column li)), PTG
// tiles in flight - the shape of trs_potrs_narrow_kernel's loads.  Compare ld = 720 (the slab) with
// ld = 16*TILES (the same bytes back to back).   hipcc --offload-arch=gfx950 -O3 tools/stride_read.hip -o tools/stride_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int TILES, bool DIAG = false>
__global__ __launch_bounds__(256) void read_kernel(const double* __restrict__ S, size_t mstride, int ld, int rows,
                                                   double* out, int B) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const double* M = S + (size_t)b * mstride;
    double acc = 0.0;
    for (int r = rows - 16; r >= 0; r -= 16) {   // a 16-row chunk from the bottom, like the substitution
        double v[TILES][4];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] = M[(size_t)(r + lq + 4 * q) * ld + (DIAG ? r : 0) + 16 * t + li];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[t][q];
    }
    if (acc == 123.456) out[b] = acc;
}

template <int TILES, bool DIAG = false>
double run(const double* S, size_t mstride, int ld, int rows, double* out, int B) {
    hipEvent_t e0, e1;  // (return codes of the HIP calls of this probe are not checked)
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((read_kernel<TILES, DIAG>), dim3((B + 3) / 4), dim3(256), 0, 0, S, mstride, ld, rows, out, B);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<TILES, DIAG>), dim3((B + 3) / 4), dim3(256), 0, 0, S, mstride, ld, rows, out, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e-3;
}

int main() {
    const int B = 4096, rows = 704;
    double *S, *out;
    const size_t slab = (size_t)rows * 720;
    hipMalloc(&S, slab * B * sizeof(double));
    hipMemset(S, 0, slab * B * sizeof(double));
    hipMalloc(&out, B * sizeof(double));
#define CASE(T)                                                                                              \
    {                                                                                                        \
        const double bytes = (double)B * rows * T * 128;                                                     \
        const double ts = run<T>(S, slab, 720, rows, out, B);                                                \
        const double tc = run<T>(S, (size_t)rows * 16 * T, 16 * T, rows, out, B);                            \
        const double td = run<T, true>(S, slab, 720, rows - 16 * T, out, B) * rows / (rows - 16 * T);        \
        printf("%d tiles (%4d B per row): slab stride %.2f TB/s (%.3f ms), back to back %.2f TB/s (%.3f ms), "  \
               "slab stride starting at the diagonal %.2f TB/s\n", T,                                       \
               T * 128, bytes / ts / 1e12, ts * 1e3, bytes / tc / 1e12, tc * 1e3, bytes / td / 1e12);        \
    }
    CASE(3) CASE(4) CASE(5) CASE(8)
    return 0;
}
