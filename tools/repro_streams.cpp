// repro_streams.cpp - torch-free reproducer of "wrong results with two or more streams" (EXPERIMENTS R4.9, R5.1).
//
// No PyTorch, no caching allocator, no Python: hipMalloc, hipStream_t, hipEvent_t and the C ABI of
// include/trs_solver.h only.  It rebuilds the launch sequence of batch.RaggedSolver's resident step -
//     per size bucket:  trs_joint_order_rows  ->  trs_solve_rows x variants
// - deals the buckets onto L streams ("lanes", each with a workspace of its own: slab, reduced vectors, assembly
// tables, envelope metadata), forks them from / joins them into stream 0 with events, and compares EVERY step,
// truss by truss and bit for bit, with the results of a one-lane step of the same library; every joint order
// that comes out of trs_joint_order_rows is checked for being a permutation.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/repro_streams.cpp -o tools/repro_streams -ldl -lpthread
//   tools/repro_streams <libtrs_hip.so> [--trusses B] [--lanes L] [--steps K] [--variants V] [--noise W]
//                       [--serial] [--seed S] [--cubes LO HI]
//     --noise W   W single-wave work-groups per CU of a time-limited spinning kernel (s_setprio 3) on a stream of
//                 its own beside every step: it takes issue slots on SOME SIMDs of every CU, which pulls the waves of
//                 the other kernels' work-groups apart - the effect a second lane's kernels have, amplified
//     --serial    the lanes' launch sequences are chained with events (bucket k + 1 waits for bucket k): several
//                 queues, but never two solver kernels at the same time
// The library is loaded with dlopen, so the product build and A/B builds (tools/build_variants.sh) run from one binary.
// Exit code: 0 = every step equal, 1 = differences, 3 = a step did not come back within the watchdog's time.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include "../include/trs_solver.h"

#define HIP(x)                                                                                     \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            _exit(2);                                                                              \
        }                                                                                          \
    } while (0)
#define TRS(x)                                                                   \
    do {                                                                         \
        int e_ = (x);                                                            \
        if (e_ != 0) {                                                           \
            fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #x, e_);     \
            _exit(2);                                                            \
        }                                                                        \
    } while (0)

struct Api {
    decltype(&trs_cubegen_dev) cubegen;
    decltype(&trs_joint_order_rows) order_rows;
    decltype(&trs_solve_rows) solve_rows;
    decltype(&trs_slab_ld) slab_ld;
    decltype(&trs_slab_rows) slab_rows;
    decltype(&trs_env_ints) env_ints;
    decltype(&trs_assemble_work_bytes) work_bytes;
    decltype(&trs_joint_order_fits) order_fits;
    decltype(&trs_abi_version) abi;
    int (*bounds)(int, int, int, int, int, int*, int*);
    int (*race_probe)(unsigned long long*, int);   // -DTRS_EXP_ORDER_RACE_PROBE builds only (else null)
};

template <typename T>
static T* dmalloc(size_t count) {
    void* p = nullptr;
    HIP(hipMalloc(&p, std::max<size_t>(1, count) * sizeof(T)));
    return static_cast<T*>(p);
}

// ---- checks on the device -------------------------------------------------------------------------------------
// one work-group per truss: rows of u / f_ext / N and the status against the reference, bit for bit
__global__ void compare_kernel(int nJ_max, int nM_max, const int* nJ, const int* nM, const double* u, const double* f,
                               const double* N, const int* info, const double* ru, const double* rf, const double* rN,
                               const int* rinfo, int* bad_count, int* bad_list) {
    const int b = blockIdx.x;
    __shared__ int differs;
    if (threadIdx.x == 0) differs = 0;
    __syncthreads();
    const unsigned long long* a0 = reinterpret_cast<const unsigned long long*>(u + (size_t)b * 3 * nJ_max);
    const unsigned long long* b0 = reinterpret_cast<const unsigned long long*>(ru + (size_t)b * 3 * nJ_max);
    const unsigned long long* a1 = reinterpret_cast<const unsigned long long*>(f + (size_t)b * 3 * nJ_max);
    const unsigned long long* b1 = reinterpret_cast<const unsigned long long*>(rf + (size_t)b * 3 * nJ_max);
    const unsigned long long* a2 = reinterpret_cast<const unsigned long long*>(N + (size_t)b * nM_max);
    const unsigned long long* b2 = reinterpret_cast<const unsigned long long*>(rN + (size_t)b * nM_max);
    int d = 0;
    for (int i = threadIdx.x; i < 3 * nJ_max; i += blockDim.x) d |= (a0[i] != b0[i]) | (a1[i] != b1[i]);
    for (int i = threadIdx.x; i < nM_max; i += blockDim.x) d |= a2[i] != b2[i];
    if (threadIdx.x == 0) d |= info[b] != rinfo[b];
    if (d) differs = 1;
    __syncthreads();
    if (threadIdx.x == 0 && differs) {
        const int at = atomicAdd(bad_count, 1);
        if (at < 64) bad_list[at] = b;
    }
}

// one work-group per truss of a bucket: is perm[0 .. nJ) a permutation of 0 .. nJ - 1 ?
__global__ void perm_check_kernel(int nJ_max, const int* nJ, const int* perm, int* bad_count) {
    extern __shared__ int seen[];
    const int b = blockIdx.x, n = nJ[b];
    for (int i = threadIdx.x; i < nJ_max; i += blockDim.x) seen[i] = 0;
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int p = perm[(size_t)b * nJ_max + i];
        if (p < 0 || p >= n) bad = 1;
        else if (atomicAdd(&seen[p], 1) != 0) bad = 1;
    }
    if (bad) atomicAdd(bad_count, 1);
}

// time-limited noise: single-wave work-groups that spin on dependent FMAs at raised priority for `ticks` of the
// constant 100 MHz clock, then end by themselves (no flag to wait for: nothing can keep them alive)
__global__ __launch_bounds__(64) void noise_kernel(unsigned long long ticks, double* sink) {
    const unsigned long long t0 = wall_clock64();
    __builtin_amdgcn_s_setprio(3);
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 256; ++i) x = fma(x, y, 1e-12);
    }
    if (x == 123.456) sink[0] = x;
}

struct Bucket {
    int count = 0, nJ_b = 0, nM_b = 0, n_b = 0, ld = 0, rows_pad = 0, lane = 0;
    size_t work_per = 0;
    int env_per = 0;
    std::vector<int64_t> idx;
    int64_t* rows = nullptr;
    double *xyz = nullptr, *loads = nullptr, *E = nullptr, *A = nullptr;
    uint8_t* cbits = nullptr;
    int32_t *conn = nullptr, *nJ = nullptr, *nM = nullptr, *free_index = nullptr, *n_free = nullptr, *info = nullptr,
            *perm = nullptr, *reach = nullptr;
};
struct Lane {
    hipStream_t stream = nullptr;
    double *S = nullptr, *uf = nullptr;
    uint8_t* work = nullptr;
    int32_t* env = nullptr;
    size_t needS = 0, needUf = 0, needWork = 0, needEnv = 0;
};
struct Outputs {
    double *u = nullptr, *f = nullptr, *N = nullptr;
    int32_t* info = nullptr;
};

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <libtrs_hip.so> [--trusses B] [--lanes L] [--steps K] [--variants V] [--noise W] "
                        "[--serial] [--seed S] [--cubes LO HI]\n", argv[0]);
        return 2;
    }
    const std::string libpath = argv[1];
    int B = 16384, lanes = 3, steps = 100, variants = 2, noise = 0, serial = 0, cube_lo = 8, cube_hi = 190;
    unsigned long long seed = 7;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? atoi(argv[++i]) : 0; };
        if (a == "--trusses") B = next();
        else if (a == "--lanes") lanes = next();
        else if (a == "--steps") steps = next();
        else if (a == "--variants") variants = next();
        else if (a == "--noise") noise = next();
        else if (a == "--serial") serial = 1;
        else if (a == "--seed") seed = (unsigned long long)next();
        else if (a == "--cubes") { cube_lo = next(); cube_hi = next(); }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    variants = std::max(1, std::min(variants, 2));
    lanes = std::max(1, lanes);

    void* h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", libpath.c_str(), dlerror()); return 2; }
    std::string hostlib = libpath.substr(0, libpath.find_last_of('/') + 1);
    if (hostlib.size() >= 9 && hostlib.compare(hostlib.size() - 9, 9, "variants/") == 0) hostlib.resize(hostlib.size() - 9);
    hostlib += "libtrs_host.so";
    void* hh = dlopen(hostlib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!hh) { fprintf(stderr, "dlopen %s: %s\n", hostlib.c_str(), dlerror()); return 2; }
    Api api;
#define SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); if (!api.field) { fprintf(stderr, "missing %s\n", name); return 2; }
    SYM(cubegen, "trs_cubegen_dev") SYM(order_rows, "trs_joint_order_rows") SYM(solve_rows, "trs_solve_rows")
    SYM(slab_ld, "trs_slab_ld") SYM(slab_rows, "trs_slab_rows") SYM(env_ints, "trs_env_ints")
    SYM(work_bytes, "trs_assemble_work_bytes") SYM(order_fits, "trs_joint_order_fits") SYM(abi, "trs_abi_version")
#undef SYM
    api.race_probe = reinterpret_cast<decltype(api.race_probe)>(dlsym(h, "trs_order_race_probe"));
    api.bounds = reinterpret_cast<decltype(api.bounds)>(dlsym(hh, "trs_cubegen_bounds"));
    if (!api.bounds) { fprintf(stderr, "missing trs_cubegen_bounds\n"); return 2; }

    HIP(hipSetDevice(0));
    hipDeviceProp_t prop;
    HIP(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs; library %s (ABI %d); %d trusses of %d..%d cubes, %d lanes, %d variants, %d steps, noise %d, %s\n",
           prop.name, n_cu, libpath.c_str(), api.abi(), B, cube_lo, cube_hi, lanes, variants, steps, noise,
           serial ? "lanes chained (no overlap)" : "lanes free-running");
    fflush(stdout);

    // ---- the batch: B random cube trusses generated on the device (BASELINE config 3's distribution) ----------
    std::vector<int32_t> cubes(B);
    unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 12345;
    for (int b = 0; b < B; ++b) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        cubes[b] = cube_lo + (int)((s >> 33) % (unsigned long long)(cube_hi - cube_lo + 1));
    }
    int nJ_bound = 0, nM_bound = 0;
    api.bounds(6, 6, 6, cube_hi, 0, &nJ_bound, &nM_bound);
    hipStream_t main_stream;
    HIP(hipStreamCreateWithFlags(&main_stream, hipStreamNonBlocking));
    int32_t* d_cubes = dmalloc<int32_t>(B);
    HIP(hipMemcpy(d_cubes, cubes.data(), B * sizeof(int32_t), hipMemcpyHostToDevice));
    const double mtypes_h[3] = {1.0, 1e7, 0.1};
    double* d_types = dmalloc<double>(3);
    HIP(hipMemcpy(d_types, mtypes_h, sizeof(mtypes_h), hipMemcpyHostToDevice));
    const double frange[6] = {-30000, 30000, -30000, 30000, -30000, 30000};
    int32_t *nJ = dmalloc<int32_t>(B), *nM = dmalloc<int32_t>(B), *nfree = dmalloc<int32_t>(B), *status = dmalloc<int32_t>(2);
    HIP(hipMemset(status, 0, 8));
    TRS(api.cubegen(B, seed, 6, 6, 6, d_cubes, 2, 3, 0, 50.0, 150.0, frange, -1, -1, d_types, 1, nJ_bound, nM_bound, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nJ, nM, nfree, status, 0, main_stream));
    HIP(hipStreamSynchronize(main_stream));
    std::vector<int32_t> h_nJ(B), h_nM(B), h_free(B);
    HIP(hipMemcpy(h_nJ.data(), nJ, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_nM.data(), nM, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_free.data(), nfree, B * 4, hipMemcpyDeviceToHost));
    const int jm = *std::max_element(h_nJ.begin(), h_nJ.end()), mm = *std::max_element(h_nM.begin(), h_nM.end());
    double *xyz = dmalloc<double>((size_t)B * jm * 3), *loads = dmalloc<double>((size_t)B * jm * 3),
           *E = dmalloc<double>((size_t)B * mm), *A = dmalloc<double>((size_t)B * mm), *rho = dmalloc<double>((size_t)B * mm);
    uint8_t* cbits = dmalloc<uint8_t>((size_t)B * jm);
    int32_t* conn = dmalloc<int32_t>((size_t)B * mm * 2);
    HIP(hipMemset(status, 0, 8));
    TRS(api.cubegen(B, seed, 6, 6, 6, d_cubes, 2, 3, 0, 50.0, 150.0, frange, -1, -1, d_types, 1, jm, mm, xyz, conn, E, A, rho,
                    cbits, loads, nJ, nM, nfree, status, 0, main_stream));
    HIP(hipStreamSynchronize(main_stream));

    // ---- buckets as batch.size_buckets(quantum = 12 x CUs, span = 2): descending size, whole rounds ---------------
    auto n_pad = [&](int b) { return (h_free[b] + 63) / 64 * 64; };
    std::vector<int> order(B);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_pad(a) > n_pad(b); });
    const size_t slab_cap = (size_t)16 << 30;
    const int quantum = 12 * n_cu;
    std::vector<Bucket> buckets;
    for (size_t i = 0; i < order.size();) {
        const int top = std::max(64, n_pad(order[i]));
        const size_t cap = std::max<size_t>(1, slab_cap / ((size_t)top * (top + 16) * 8));
        size_t in_span = 0;
        while (i + in_span < order.size() && n_pad(order[i + in_span]) >= top - 128) ++in_span;
        size_t take = std::min(cap, in_span);
        if (take >= (size_t)quantum) take = take / quantum * quantum;
        Bucket bk;
        bk.idx.assign(order.begin() + i, order.begin() + i + take);
        std::sort(bk.idx.begin(), bk.idx.end());
        std::stable_sort(bk.idx.begin(), bk.idx.end(), [&](int64_t a, int64_t b) { return h_free[a] < h_free[b]; });
        bk.count = (int)take;
        for (int64_t g : bk.idx) {
            bk.nJ_b = std::max(bk.nJ_b, h_nJ[g]);
            bk.nM_b = std::max(bk.nM_b, h_nM[g]);
            bk.n_b = std::max(bk.n_b, h_free[g]);
        }
        bk.nJ_b = std::max(1, bk.nJ_b);
        bk.nM_b = std::max(1, bk.nM_b);
        if (!api.order_fits(bk.nJ_b, bk.nM_b)) { fprintf(stderr, "bucket does not fit trs_joint_order\n"); return 2; }
        bk.ld = api.slab_ld(bk.n_b);
        bk.rows_pad = api.slab_rows(bk.n_b);
        bk.work_per = api.work_bytes(bk.nJ_b, bk.nM_b, bk.n_b);
        bk.env_per = api.env_ints(bk.n_b);
        buckets.push_back(std::move(bk));
        i += take;
    }
    std::stable_sort(buckets.begin(), buckets.end(), [](const Bucket& a, const Bucket& b) {
        return (size_t)a.count * a.rows_pad * a.ld > (size_t)b.count * b.rows_pad * b.ld;
    });
    for (auto& bk : buckets) {
        const size_t Bb = bk.count;
        bk.rows = dmalloc<int64_t>(Bb);
        HIP(hipMemcpy(bk.rows, bk.idx.data(), Bb * 8, hipMemcpyHostToDevice));
        bk.xyz = dmalloc<double>(Bb * bk.nJ_b * 3);
        bk.loads = dmalloc<double>(Bb * bk.nJ_b * 3);
        bk.cbits = dmalloc<uint8_t>(Bb * bk.nJ_b);
        bk.conn = dmalloc<int32_t>(Bb * bk.nM_b * 2);
        bk.E = dmalloc<double>(Bb * bk.nM_b);
        bk.A = dmalloc<double>(Bb * bk.nM_b);
        bk.nJ = dmalloc<int32_t>(Bb);
        bk.nM = dmalloc<int32_t>(Bb);
        bk.free_index = dmalloc<int32_t>(Bb * bk.nJ_b * 3);
        bk.n_free = dmalloc<int32_t>(Bb);
        bk.info = dmalloc<int32_t>(Bb);
        bk.perm = dmalloc<int32_t>(Bb * bk.nJ_b);
        bk.reach = dmalloc<int32_t>(Bb);
    }
    printf("%zu buckets:", buckets.size());
    for (auto& bk : buckets) printf(" %dx%d", bk.count, bk.rows_pad);
    printf("\n");
    fflush(stdout);

    // outputs: one set per variant + the reference sets; zeroed once (what lies beyond a bucket's width is never written)
    auto make_out = [&]() {
        Outputs o;
        o.u = dmalloc<double>((size_t)B * jm * 3);
        o.f = dmalloc<double>((size_t)B * jm * 3);
        o.N = dmalloc<double>((size_t)B * mm);
        o.info = dmalloc<int32_t>(B);
        HIP(hipMemset(o.u, 0, (size_t)B * jm * 24));
        HIP(hipMemset(o.f, 0, (size_t)B * jm * 24));
        HIP(hipMemset(o.N, 0, (size_t)B * mm * 8));
        HIP(hipMemset(o.info, 0, (size_t)B * 4));
        return o;
    };
    std::vector<Outputs> outs, refs;
    for (int v = 0; v < variants; ++v) { outs.push_back(make_out()); refs.push_back(make_out()); }
    int* d_bad = dmalloc<int>(2);
    int* d_bad_list = dmalloc<int>(64);
    double* d_sink = dmalloc<double>(1);

    std::atomic<long long> heartbeat{0};
    std::atomic<int> finished{0};
    std::thread watchdog([&]() {   // a step that does not come back: say so and leave (the process holds a hung queue)
        long long last = -1;
        int quiet = 0;
        while (!finished.load()) {
            std::this_thread::sleep_for(std::chrono::seconds(1));
            const long long now = heartbeat.load();
            quiet = now == last ? quiet + 1 : 0;
            last = now;
            if (quiet >= 30) {
                printf("STALL: no progress for 30 s at heartbeat %lld\nRESULT stall\n", now);
                fflush(stdout);
                _exit(3);
            }
        }
    });

    // ---- one step on `L` lanes -------------------------------------------------------------------------------------
    std::vector<Lane> lane_set;
    auto setup_lanes = [&](int L) {
        for (auto& ln : lane_set) {
            HIP(hipFree(ln.S)); HIP(hipFree(ln.uf)); HIP(hipFree(ln.work)); HIP(hipFree(ln.env));
            if (ln.stream != main_stream) HIP(hipStreamDestroy(ln.stream));
        }
        lane_set.assign(L, Lane{});
        std::vector<size_t> load(L, 0);
        for (auto& bk : buckets) {   // longest-processing-time-first on the slab sizes (buckets come largest first)
            const int l = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            bk.lane = l;
            const size_t slab = (size_t)bk.count * bk.rows_pad * bk.ld;
            load[l] += slab;
            Lane& ln = lane_set[l];
            ln.needS = std::max(ln.needS, slab);
            ln.needUf = std::max(ln.needUf, (size_t)bk.count * bk.rows_pad);
            ln.needWork = std::max(ln.needWork, (size_t)bk.count * bk.work_per);
            ln.needEnv = std::max(ln.needEnv, (size_t)bk.count * bk.env_per);
        }
        for (int l = 0; l < L; ++l) {
            Lane& ln = lane_set[l];
            if (l == 0) ln.stream = main_stream;
            else HIP(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            ln.S = dmalloc<double>(ln.needS);
            ln.uf = dmalloc<double>(ln.needUf);
            ln.work = dmalloc<uint8_t>(ln.needWork);
            ln.env = dmalloc<int32_t>(ln.needEnv);
            HIP(hipMemset(ln.env, 0, std::max<size_t>(1, ln.needEnv) * 4));
        }
        HIP(hipDeviceSynchronize());
    };
    hipStream_t noise_stream;
    HIP(hipStreamCreateWithFlags(&noise_stream, hipStreamNonBlocking));
    std::vector<hipEvent_t> events;   // kept until the end of the step
    auto new_event = [&]() {
        hipEvent_t e;
        HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        events.push_back(e);
        return e;
    };
    auto step = [&](std::vector<Outputs>& dst, unsigned long long noise_ticks) {
        const int L = (int)lane_set.size();
        hipEvent_t fork = new_event();
        HIP(hipEventRecord(fork, main_stream));
        for (int l = 1; l < L; ++l) HIP(hipStreamWaitEvent(lane_set[l].stream, fork, 0));
        if (noise > 0 && noise_ticks > 0) {
            HIP(hipStreamWaitEvent(noise_stream, fork, 0));
            hipLaunchKernelGGL(noise_kernel, dim3(noise * n_cu), dim3(64), 0, noise_stream, noise_ticks, d_sink);
            HIP(hipGetLastError());
        }
        hipEvent_t chain = nullptr;
        for (auto& bk : buckets) {
            Lane& ln = lane_set[bk.lane];
            if (serial && chain) HIP(hipStreamWaitEvent(ln.stream, chain, 0));
            TRS(api.order_rows(bk.count, bk.nJ_b, bk.nM_b, bk.rows, jm, mm, xyz, conn, cbits, loads, E, A, nJ, nM, bk.perm,
                               bk.reach, bk.xyz, bk.conn, bk.cbits, bk.loads, bk.E, bk.A, bk.nJ, bk.nM, 3, ln.stream));
            for (int v = 0; v < variants; ++v)
                TRS(api.solve_rows(bk.count, bk.nJ_b, bk.nM_b, bk.n_b, bk.xyz, bk.conn, bk.E, bk.A, bk.cbits, bk.loads, bk.nJ,
                                   bk.nM, bk.free_index, bk.n_free, bk.ld, bk.rows_pad, ln.S, ln.uf, bk.rows_pad, dst[v].u,
                                   dst[v].f, dst[v].N, bk.info, ln.work, ln.env, bk.perm, bk.rows, jm, mm, dst[v].info, 0,
                                   ln.stream));
            if (serial) {
                chain = new_event();
                HIP(hipEventRecord(chain, ln.stream));
            }
        }
        for (int l = 1; l < L; ++l) {
            hipEvent_t done = new_event();
            HIP(hipEventRecord(done, lane_set[l].stream));
            HIP(hipStreamWaitEvent(main_stream, done, 0));
        }
        if (noise > 0 && noise_ticks > 0) {
            hipEvent_t done = new_event();
            HIP(hipEventRecord(done, noise_stream));
            HIP(hipStreamWaitEvent(main_stream, done, 0));
        }
    };
    auto end_step = [&]() {
        HIP(hipStreamSynchronize(main_stream));
        for (hipEvent_t e : events) HIP(hipEventDestroy(e));
        events.clear();
        heartbeat.fetch_add(1);
    };

    // reference: one lane, no noise
    setup_lanes(1);
    step(refs, 0);
    end_step();
    auto t0 = std::chrono::steady_clock::now();
    step(refs, 0);
    end_step();
    const double ref_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {   // the reference itself must be sane: every status 0, every order a permutation
        std::vector<int32_t> hinfo(B);
        HIP(hipMemcpy(hinfo.data(), refs[0].info, B * 4, hipMemcpyDeviceToHost));
        const long failed = std::count_if(hinfo.begin(), hinfo.end(), [](int v) { return v != 0; });
        printf("reference step (1 lane): %.1f ms, %ld trusses with a failed pivot\n", ref_ms, failed);
    }
    if (api.race_probe) {   // probe build: reads of a MOVING level counter in trs_joint_order's breadth-first sweeps
        unsigned long long hits[2] = {0, 0};
        TRS(api.race_probe(hits, 1));
        printf("probe: %llu hazardous counter reads in the 2 one-lane reference steps\n", hits[0]);
    }
    setup_lanes(lanes);
    const unsigned long long noise_ticks = (unsigned long long)(ref_ms * 1.2 * 1e5);   // 100 MHz clock: 1e5 ticks per ms
    long bad_steps = 0, bad_trusses = 0, bad_perms = 0;
    for (int it = 0; it < steps; ++it) {
        for (auto& o : outs) HIP(hipMemsetAsync(o.info, 0xff, (size_t)B * 4, main_stream));
        step(outs, noise_ticks);
        end_step();
        int h_bad[2] = {0, 0};
        HIP(hipMemset(d_bad, 0, 8));
        for (int v = 0; v < variants; ++v) {
            hipLaunchKernelGGL(compare_kernel, dim3(B), dim3(256), 0, main_stream, jm, mm, nJ, nM, outs[v].u, outs[v].f,
                               outs[v].N, outs[v].info, refs[v].u, refs[v].f, refs[v].N, refs[v].info, d_bad, d_bad_list);
            HIP(hipGetLastError());
        }
        for (auto& bk : buckets) {
            hipLaunchKernelGGL(perm_check_kernel, dim3(bk.count), dim3(256), bk.nJ_b * sizeof(int), main_stream, bk.nJ_b,
                               bk.nJ, bk.perm, d_bad + 1);
            HIP(hipGetLastError());
        }
        HIP(hipStreamSynchronize(main_stream));
        HIP(hipMemcpy(h_bad, d_bad, 8, hipMemcpyDeviceToHost));
        heartbeat.fetch_add(1);
        if (h_bad[0] || h_bad[1]) {
            ++bad_steps;
            bad_trusses += h_bad[0];
            bad_perms += h_bad[1];
            int list[8] = {0};
            HIP(hipMemcpy(list, d_bad_list, sizeof(list), hipMemcpyDeviceToHost));
            printf("step %d: %d truss results differ from the one-lane run, %d joint orders are no permutation; first:", it,
                   h_bad[0], h_bad[1]);
            for (int k = 0; k < std::min(8, h_bad[0]); ++k) printf(" %d (n_free %d)", list[k], h_free[list[k]]);
            printf("\n");
            fflush(stdout);
        }
    }
    finished.store(1);
    watchdog.join();
    if (api.race_probe) {
        unsigned long long hits[2] = {0, 0};
        TRS(api.race_probe(hits, 1));
        printf("probe: %llu hazardous counter reads in %d steps (each one a thread that, with the single running counter of "
               "rounds 3-4, would have taken a wrong level end)\n", hits[0], steps);
    }
    printf("RESULT lanes=%d variants=%d noise=%d serial=%d steps=%d: %ld steps with differences, %ld truss results, "
           "%ld non-permutation orders\n", lanes, variants, noise, serial, steps, bad_steps, bad_trusses, bad_perms);
    return bad_steps ? 1 : 0;
}
