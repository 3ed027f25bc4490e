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

constexpr int MAX_FIELDS = 12;
struct RowFields {
    const unsigned char* src[MAX_FIELDS];
    unsigned char* dst[MAX_FIELDS];
    unsigned long long src_pitch[MAX_FIELDS], dst_pitch[MAX_FIELDS], width[MAX_FIELDS];  // bytes
    int unit[MAX_FIELDS];                                                                 // 16, 8, 4 or 1
};

template <typename V>
__device__ __forceinline__ void copy_units(const unsigned char* s, unsigned char* d, unsigned long long bytes, int lane) {
    const V* sv = reinterpret_cast<const V*>(s);
    V* dv = reinterpret_cast<V*>(d);
    const unsigned long long n = bytes / sizeof(V);
    for (unsigned long long i = lane; i < n; i += 64) dv[i] = sv[i];
}

// scatter == 0: dst[i] = src[rows[i]];  scatter != 0: dst[rows[i]] = src[i].   One wave per row, strided.
__global__ __launch_bounds__(256) void trs_copy_rows_kernel(const RowFields f, const long long* __restrict__ rows,
                                                            const int count, const int scatter) {
    const int field = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwave = gridDim.x * 4;
    const unsigned long long sp = f.src_pitch[field], dp = f.dst_pitch[field], w = f.width[field];
    for (int i = wave; i < count; i += nwave) {
        const long long r = rows[i];
        const unsigned char* s = f.src[field] + (scatter ? (unsigned long long)i : (unsigned long long)r) * sp;
        unsigned char* d = f.dst[field] + (scatter ? (unsigned long long)r : (unsigned long long)i) * dp;
        switch (f.unit[field]) {
            case 16: copy_units<uint4>(s, d, w, lane); break;
            case 8: copy_units<uint2>(s, d, w, lane); break;
            case 4: copy_units<unsigned>(s, d, w, lane); break;
            default: copy_units<unsigned char>(s, d, w, lane); break;
        }
    }
}

}  // namespace

extern "C" int trs_copy_rows_launch(int nfields, const void* const* src, const size_t* src_pitch, void* const* dst,
                                    const size_t* dst_pitch, const size_t* width, int count, const long long* rows,
                                    int scatter, hipStream_t stream) {
    if (count <= 0 || nfields <= 0) return 0;
    if (nfields > MAX_FIELDS) return (int)hipErrorInvalidValue;
    RowFields f;
    unsigned long long most = 0;
    for (int k = 0; k < nfields; ++k) {
        if (width[k] > src_pitch[k] || width[k] > dst_pitch[k]) return (int)hipErrorInvalidValue;
        f.src[k] = static_cast<const unsigned char*>(src[k]);
        f.dst[k] = static_cast<unsigned char*>(dst[k]);
        f.src_pitch[k] = src_pitch[k];
        f.dst_pitch[k] = dst_pitch[k];
        f.width[k] = width[k];
        const unsigned long long all = (unsigned long long)(uintptr_t)src[k] | (unsigned long long)(uintptr_t)dst[k] |
                                       src_pitch[k] | dst_pitch[k] | width[k];
        f.unit[k] = all % 16 == 0 ? 16 : (all % 8 == 0 ? 8 : (all % 4 == 0 ? 4 : 1));
        most = width[k] > most ? width[k] : most;
    }
    // enough waves to fill the chip: one wave per row up to 8192 waves per field
    int blocks = (count + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(trs_copy_rows_kernel, dim3(blocks, nfields), dim3(256), 0, stream, f, rows, count, scatter);
    return (int)hipGetLastError();
}
