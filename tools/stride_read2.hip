// Second read-rate probe: trs_potrs_narrow_kernel's streaming part rebuilt feature by feature, to find what
// keeps it at 5.0 TB/s when a bare read kernel of the same shape reaches 6.0 (tools/stride_read.hip).
//   MODE bit 0: buffer loads through a descriptor (raw_buffer_load) instead of global loads
//   MODE bit 1: per-chunk tile count from a table in memory (scalar load per chunk), 3..5 tiles, absent tiles
//               requested out of range
//   MODE bit 2: 22.5 KB of LDS per work-group and one LDS read per tile (the solution strip)
//   MODE bit 3: tiles start at the diagonal (column 16 s of chunk s)
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/stride_read2.hip -o tools/stride_read2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const double* __restrict__ S, size_t mstride, int ld, int rows,
                                             const int* __restrict__ cnt, double* out, int B) {
    extern __shared__ double sh[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const double* M = S + (size_t)b * mstride;
    double* us = sh + wave * rows;
    if (MODE & 4) {
        for (int c = lane; c < rows; c += 64) us[c] = 1.0;
        __builtin_amdgcn_wave_barrier();
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(M), 0, (int)(mstride * 8), 0x00020000);
    const unsigned loff = ((unsigned)lq * (unsigned)ld + (unsigned)li) * 8u;
    double acc = 0.0;
    const int nch = rows / 16;
    for (int s = nch - 1; s >= 0; --s) {
        const int nt = (MODE & 2) ? cnt[b * nch + s] : 4;   // tiles of this chunk
        d4 v[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (!(MODE & 2) && t == 4) { v[t] = d4{0, 0, 0, 0}; continue; }
            const int col = ((MODE & 8) ? 16 * s : 0) + 16 * t;
            const bool exists = t < nt && col + 16 <= ld;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE & 1) {
                    const unsigned vo = exists ? loff : 0x80000000u;
                    v[t][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                             rs, vo, ((16 * s + 4 * q) * ld + col) * 8, 0));
                } else {
                    v[t][q] = exists ? M[(size_t)(16 * s + lq + 4 * q) * ld + col + li] : 0.0;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const double u = (MODE & 4) ? us[(16 * (s + t) + li) % rows] : 1.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[t][q] * u;
        }
    }
    if (acc == 123.456) out[b] = acc;
}

template <int MODE>
void run(const double* S, size_t slab, int rows, const int* cnt, double* out, int B, double bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (MODE & 4) ? (size_t)4 * rows * 8 : 0;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe<MODE>, dim3((B + 3) / 4), dim3(256), lds, 0, S, slab, 720, rows, cnt, out, B);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3((B + 3) / 4), dim3(256), lds, 0, S, slab, 720, rows, cnt, out, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double t = ms / reps * 1e-3;
    printf("mode %2d (%s%s%s%s): %.3f ms, %.2f TB/s\n", MODE, (MODE & 1) ? "buffer " : "global ", (MODE & 2) ? "table " : "",
           (MODE & 4) ? "lds " : "", (MODE & 8) ? "diagonal" : "", t * 1e3, bytes / t / 1e12);
}

int main() {
    const int B = 4096, rows = 704, nch = rows / 16;
    double *S, *out; int* cnt;
    const size_t slab = (size_t)rows * 720;
    hipMalloc(&S, slab * B * sizeof(double));
    hipMemset(S, 0, slab * B * sizeof(double));
    hipMalloc(&out, B * sizeof(double));
    std::vector<int> h((size_t)B * nch);
    double tiles = 0;
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < nch; ++s) { h[(size_t)b * nch + s] = 3 + (s % 3 == 0) + (s % 4 == 1); tiles += h[(size_t)b * nch + s]; }
    hipMalloc(&cnt, h.size() * sizeof(int));
    hipMemcpy(cnt, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    const double b4 = (double)B * nch * 4 * 2048, bt = tiles * 2048;
    run<0>(S, slab, rows, cnt, out, B, b4);
    run<1>(S, slab, rows, cnt, out, B, b4);
    run<3>(S, slab, rows, cnt, out, B, bt);
    run<5>(S, slab, rows, cnt, out, B, b4);
    run<9>(S, slab, rows, cnt, out, B, b4);
    run<7>(S, slab, rows, cnt, out, B, bt);
    run<15>(S, slab, rows, cnt, out, B, bt);
    return 0;
}
