// Row gather / scatter between the padded arrays of a ragged batch and those of one of its size buckets.
//
// A ragged batch (the reference's GenerateRandomCubeTrusses loop, generate.py:342-374: 120 .. 2000 members per
// truss) is solved bucket by bucket (trusses of one padded system size share a launch and a slab shape).  A
// bucket's arrays are the rows of its trusses, TRIMMED to the bucket's own maxima - every trimmed field is a
// prefix of the full row (xyz[b][:nJ_b][:], conn[b][:nM_b][:], ...), so both directions are prefix copies of
// rows selected by an index list.  All fields of a bucket go in ONE launch (blockIdx.y = field).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int MAX_FIELDS = 12;
struct RowFields {
    const unsigned char* src[MAX_FIELDS];
    unsigned char* dst[MAX_FIELDS];
    unsigned long long src_pitch[MAX_FIELDS], dst_pitch[MAX_FIELDS], width[MAX_FIELDS];  // bytes
    unsigned long long fill_to[MAX_FIELDS];  // zeros are written behind the copied prefix up to this many bytes
    const int* counts[MAX_FIELDS];           // live elements of bucket row i (or null: the whole width is copied)
    int* live[MAX_FIELDS];                   // scatter: bytes of far row r that may be non-zero (or null)
    unsigned elem[MAX_FIELDS];               // bytes per counted element
    int unit[MAX_FIELDS];                    // 16, 8, 4 or 1
    int nfields;
};

// zeros over the bytes [from, to) of a row: single bytes up to the first unit boundary and behind the last one
template <typename V>
__device__ __forceinline__ void zero_bytes(unsigned char* d, unsigned long long from, unsigned long long to, int lane) {
    if (to <= from) return;
    constexpr unsigned long long U = sizeof(V);
    unsigned long long a = (from + U - 1) / U * U, b = to / U * U;
    if (a > b) { a = to; b = to; }   // no whole unit inside
    if (from + lane < a) d[from + lane] = 0;
    if (b + lane < to) d[b + lane] = 0;
    V* dv = reinterpret_cast<V*>(d);
    V z;
    __builtin_memset(&z, 0, sizeof(V));
    for (unsigned long long i = a / U + lane; i < b / U; i += 64) dv[i] = z;
}

// the first `bytes` bytes of a row (any count: whole units, then single bytes)
template <typename V>
__device__ __forceinline__ void copy_bytes(const unsigned char* s, unsigned char* d, unsigned long long bytes, int lane) {
    const V* sv = reinterpret_cast<const V*>(s);
    V* dv = reinterpret_cast<V*>(d);
    const unsigned long long n = bytes / sizeof(V), rest = bytes % sizeof(V);
    // eight loads in flight per lane before the first store, in the last (or only) round as well, where they are
    // predicated: a copy whose far side is host memory lives on bytes in flight, not on waves
    for (unsigned long long i = lane; i < n; i += 8 * 64) {
        V v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + 64 * k < n) v[k] = sv[i + 64 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + 64 * k < n) dv[i + 64 * k] = v[k];
    }
    if ((unsigned long long)lane < rest) d[n * sizeof(V) + lane] = s[n * sizeof(V) + lane];
}

template <typename V>
__device__ __forceinline__ void move_row(const unsigned char* s, unsigned char* d, unsigned long long n,
                                         unsigned long long zero_to, int lane) {
    copy_bytes<V>(s, d, n, lane);
    zero_bytes<V>(d, n, zero_to, lane);
}

// scatter == 0: dst[i] = src[rows[i]];  scatter != 0: dst[rows[i]] = src[i].   One wave per row, strided.
// gridDim.y == nfields: blockIdx.y = field (device-to-device: as many waves as the chip takes);
// gridDim.y == 1: every work-group walks all fields (host-side copies: FEW waves in total, see the launcher).
__global__ __launch_bounds__(256) void trs_copy_rows_kernel(const RowFields f, const long long* __restrict__ rows,
                                                            const int count, const int scatter) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwave = gridDim.x * 4;
    const int f0 = gridDim.y == 1 ? 0 : blockIdx.y, f1 = gridDim.y == 1 ? f.nfields : f0 + 1;
    for (int field = f0; field < f1; ++field) {
        const unsigned long long sp = f.src_pitch[field], dp = f.dst_pitch[field], w = f.width[field];
        const int* cnt = f.counts[field];
        int* lv = scatter ? f.live[field] : nullptr;
        const unsigned long long el = f.elem[field];
        for (int i = wave; i < count; i += nwave) {
            const long long r = rows[i];
            const unsigned char* s = f.src[field] + (scatter ? (unsigned long long)i : (unsigned long long)r) * sp;
            unsigned char* d = f.dst[field] + (scatter ? (unsigned long long)r : (unsigned long long)i) * dp;
            unsigned long long n = w, zero_to = f.fill_to[field];
            if (cnt) {
                const unsigned long long have = (unsigned long long)(cnt[i] > 0 ? cnt[i] : 0) * el;
                n = have < w ? have : w;
            }
            if (lv) {   // the far row is zero behind live[r] already: only what the LAST writer left has to go
                const unsigned long long old = (unsigned long long)lv[r];
                zero_to = old < dp ? old : dp;
                if (lane == 0) lv[r] = (int)n;
            }
            switch (f.unit[field]) {
                case 16: move_row<v4u>(s, d, n, zero_to, lane); break;
                case 8: move_row<v2u>(s, d, n, zero_to, lane); break;
                case 4: move_row<unsigned>(s, d, n, zero_to, lane); break;
                default: move_row<unsigned char>(s, d, n, zero_to, lane); break;
            }
        }
    }
}

}  // namespace

extern "C" int trs_copy_rows_launch(int nfields, const void* const* src, const size_t* src_pitch, void* const* dst,
                                    const size_t* dst_pitch, const size_t* width, const size_t* fill_to,
                                    const int* const* counts, const size_t* elem, int* const* live, int count,
                                    const long long* rows, int scatter, int max_blocks, hipStream_t stream) {
    if (count <= 0 || nfields <= 0) return 0;
    if (nfields > MAX_FIELDS) return (int)hipErrorInvalidValue;
    RowFields f;
    for (int k = 0; k < nfields; ++k) {
        if (width[k] > src_pitch[k] || width[k] > dst_pitch[k]) return (int)hipErrorInvalidValue;
        f.src[k] = static_cast<const unsigned char*>(src[k]);
        f.dst[k] = static_cast<unsigned char*>(dst[k]);
        f.src_pitch[k] = src_pitch[k];
        f.dst_pitch[k] = dst_pitch[k];
        f.width[k] = width[k];
        f.fill_to[k] = fill_to != nullptr ? fill_to[k] : 0;
        if (f.fill_to[k] > dst_pitch[k]) return (int)hipErrorInvalidValue;
        f.counts[k] = counts != nullptr ? counts[k] : nullptr;
        f.elem[k] = 0;
        if (f.counts[k] != nullptr) {
            if (elem == nullptr || elem[k] == 0 || elem[k] > width[k]) return (int)hipErrorInvalidValue;
            f.elem[k] = (unsigned)elem[k];
        }
        f.live[k] = (live != nullptr && scatter) ? live[k] : nullptr;
        if (f.live[k] != nullptr && dst_pitch[k] > 0x7fffffffull) return (int)hipErrorInvalidValue;
        const unsigned long long all = (unsigned long long)(uintptr_t)src[k] | (unsigned long long)(uintptr_t)dst[k] |
                                       src_pitch[k] | dst_pitch[k];
        f.unit[k] = all % 16 == 0 ? 16 : (all % 8 == 0 ? 8 : (all % 4 == 0 ? 4 : 1));
    }
    f.nfields = nfields;
    // enough waves to fill the chip: one wave per row up to 8192 waves per field - or FEW work-groups IN TOTAL when
    // the caller says so (max_blocks > 0: one side is page-locked host memory; a PCIe stream needs bytes in
    // flight, not the whole chip - the host-fed pipeline gives these launches a few CUs of their own)
    int blocks = (count + 3) / 4;
    const int cap = max_blocks > 0 ? max_blocks : 2048;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(trs_copy_rows_kernel, dim3(blocks, max_blocks > 0 ? 1 : nfields), dim3(256), 0, stream, f, rows,
                       count, scatter);
    return (int)hipGetLastError();
}
