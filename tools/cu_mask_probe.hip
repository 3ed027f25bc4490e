// Where do the work-groups of a CU-masked stream run, and what does a kernel that pulls page-locked host
// memory reach on a handful of CUs?   hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o tools/cu_mask_probe
//   tools/cu_mask_probe <first_bit> <n_bits> [stride]     mask = n_bits bits from first_bit, `stride` apart
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) << 16 | ((hw >> 13) & 7u) << 8 | ((hw >> 12) & 1u) << 4 | ((hw >> 8) & 15u);
}

__global__ __launch_bounds__(256) void pull_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i + 7 * step < n; i += 8 * step) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * step];
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * step] = v[k];
    }
    for (; i < n; i += step) dst[i] = src[i];
}

int main(int argc, char** argv) {
    int first = argc > 1 ? atoi(argv[1]) : 0, nbits = argc > 2 ? atoi(argv[2]) : 8, stride = argc > 3 ? atoi(argv[3]) : 1;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u), rest((ncu + 31) / 32, 0u);
    for (int k = 0; k < nbits; ++k) { int b = first + k * stride; mask[b / 32] |= 1u << (b % 32); }
    for (int b = 0; b < ncu; ++b) if (!(mask[b / 32] >> (b % 32) & 1u)) rest[b / 32] |= 1u << (b % 32);
    hipStream_t sm, sr;
    CK(hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data()));
    CK(hipExtStreamCreateWithCUMask(&sr, (uint32_t)rest.size(), rest.data()));
    printf("CUs %d, mask words", ncu); for (auto w : mask) printf(" %08x", w); printf("\n");

    const int NB = 4096;
    unsigned* where; CK(hipMalloc(&where, NB * 4));
    std::vector<unsigned> h(NB);
    for (int pass = 0; pass < 2; ++pass) {
        hipStream_t s = pass ? sr : sm;
        where_kernel<<<NB, 64, 0, s>>>(where, 20000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), where, NB * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> hist, per_xcc;
        for (auto v : h) { hist[v]++; per_xcc[v >> 16]++; }
        printf("%s stream: %zu distinct (xcc,se,sh,cu); per XCC:", pass ? "complement" : "masked", hist.size());
        for (auto& kv : per_xcc) { int cus = 0; for (auto& q : hist) cus += (q.first >> 16) == kv.first; printf(" x%u:%dcu/%dwg", kv.first, cus, kv.second); }
        printf("\n");
        if (!pass) { printf("  masked CUs:"); for (auto& kv : hist) printf(" x%u.se%u.sh%u.cu%u", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); printf("\n"); }
    }

    // PCIe pull on the masked stream, alone and beside a spinning kernel on the complement
    size_t bytes = (size_t)1 << 30;
    void *hsrc, *ddst, *hdst;
    CK(hipHostMalloc(&hsrc, bytes, hipHostMallocMapped)); memset(hsrc, 1, bytes);
    CK(hipHostMalloc(&hdst, bytes, hipHostMallocMapped)); memset(hdst, 2, bytes);
    CK(hipMalloc(&ddst, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wg : {nbits * 2, nbits * 4, nbits * 8}) {
        for (int dir = 0; dir < 2; ++dir) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, sm));
                if (dir == 0) pull_kernel<<<wg, 256, 0, sm>>>((const uint4*)hsrc, (uint4*)ddst, bytes / 16);
                else pull_kernel<<<wg, 256, 0, sm>>>((const uint4*)ddst, (uint4*)hdst, bytes / 16);
                CK(hipEventRecord(e1, sm)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%s, %d work-groups on %d CUs: %.1f GB/s\n", dir ? "push" : "pull", wg, nbits, bytes / best / 1e6);
        }
    }
    // both directions at once on the masked CUs (two masked streams)
    hipStream_t sm2; CK(hipExtStreamCreateWithCUMask(&sm2, (uint32_t)mask.size(), mask.data()));
    {
        auto t0 = std::chrono::steady_clock::now();
        pull_kernel<<<nbits * 4, 256, 0, sm>>>((const uint4*)hsrc, (uint4*)ddst, bytes / 16);
        pull_kernel<<<nbits * 4, 256, 0, sm2>>>((const uint4*)ddst + bytes / 32, (uint4*)hdst, bytes / 32);
        CK(hipStreamSynchronize(sm)); CK(hipStreamSynchronize(sm2));
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("pull 1 GiB + push 0.5 GiB together: %.1f ms (%.1f GB/s summed)\n", s * 1e3, 1.5 * bytes / s / 1e9);
    }
    // both directions at once on DISJOINT sets of CUs: pull on the masked CUs, push on the next n_bits CUs
    {
        std::vector<uint32_t> m2(mask.size(), 0u);
        for (int k = 0; k < nbits; ++k) { int b = first + (nbits + k) * stride; m2[b / 32] |= 1u << (b % 32); }
        hipStream_t sp; CK(hipExtStreamCreateWithCUMask(&sp, (uint32_t)m2.size(), m2.data()));
        for (int per_cu : {1, 2, 4, 8}) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            pull_kernel<<<nbits * per_cu, 256, 0, sm>>>((const uint4*)hsrc, (uint4*)ddst, bytes / 16);
            pull_kernel<<<nbits * per_cu, 256, 0, sp>>>((const uint4*)ddst + bytes / 32, (uint4*)hdst, bytes / 32);
            CK(hipStreamSynchronize(sp));
            double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(sm));
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("disjoint CUs, %d work-groups per CU: push 0.5 GiB done at %.1f ms, pull 1 GiB at %.1f ms (%.1f GB/s summed)\n",
                   per_cu, s1 * 1e3, s * 1e3, 1.5 * bytes / s / 1e9);
        }
    }
    // do kernels on the masked stream and on its complement run at the same time?
    {
        hipStream_t plain; CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
        unsigned* w2; CK(hipMalloc(&w2, 65536 * 4));
        auto both = [&](hipStream_t a, hipStream_t b, const char* what) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            where_kernel<<<1024, 256, 0, a>>>(w2, 20000000);   // one round of about 10 ms
            auto t1 = std::chrono::steady_clock::now();
            pull_kernel<<<nbits * 4, 256, 0, b>>>((const uint4*)hsrc, (uint4*)ddst, bytes / 64);
            CK(hipStreamSynchronize(b));
            double sb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(a));
            double sa = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%s: spin kernel done at %.1f ms, 256 MiB pull done at %.1f ms (launch took %.2f ms)\n", what, sa * 1e3, sb * 1e3,
                   std::chrono::duration<double>(t1 - t0).count() * 1e3);
        };
        both(sr, sm, "spin on the complement, pull on the masked CUs");
        both(plain, sm, "spin on an ordinary stream, pull on the masked CUs");
        hipStream_t plain2; CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
        both(plain, plain2, "spin and pull on two ordinary streams");
    }
    return 0;
}
