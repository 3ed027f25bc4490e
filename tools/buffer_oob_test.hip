// Probe: raw buffer loads/stores whose VGPR offset is beyond num_records return 0 / are dropped,
// whatever the SGPR offset says (range check on voffset + inst_offset only).  The factorisation uses
// this to skip tiles outside the envelope without branches: voffset = exists ? lane_offset : 0x80000000.
// The allocation is 3 GiB so that even a (wrongly) dereferenced base + 2 GiB stays inside it.
//   hipcc --offload-arch=gfx950 -O3 tools/buffer_oob_test.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(double* base, int num_records, double* out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, num_records, 0x00020000);
    const unsigned lane = threadIdx.x;
    const unsigned in_range = lane * 8u, oob = 0x80000000u + lane * 8u;
    // loads: in range (soffset 1024), OOB voffset with the same soffset
    out[lane] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, in_range, 1024, 0));
    out[64 + lane] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, oob, 1024, 0));
    // stores: OOB voffset must not land anywhere; in-range one must
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, 7.0), rs, oob, 2048, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, 9.0), rs, in_range, 4096, 0);
}

int main() {
    const size_t total = 3ull << 30;
    double *buf, *out;
    hipMalloc(&buf, total);
    hipMalloc(&out, 128 * 8);
    hipMemset(buf, 0, total);
    double ones[64];
    for (int i = 0; i < 64; ++i) ones[i] = 1.0 + i;
    hipMemcpy(reinterpret_cast<char*>(buf) + 1024, ones, sizeof(ones), hipMemcpyHostToDevice);
    // poison where a wrongly wrapped access would land: base + 2 GiB + soffset
    double poison[64];
    for (int i = 0; i < 64; ++i) poison[i] = -555.0;
    hipMemcpy(reinterpret_cast<char*>(buf) + (2ull << 30) + 1024, poison, sizeof(poison), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, 1 << 20, out);
    double h[128], chk[64], chk2[64], chk3[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(chk, reinterpret_cast<char*>(buf) + (2ull << 30) + 2048, sizeof(chk), hipMemcpyDeviceToHost);
    hipMemcpy(chk2, reinterpret_cast<char*>(buf) + 2048, sizeof(chk2), hipMemcpyDeviceToHost);
    hipMemcpy(chk3, reinterpret_cast<char*>(buf) + 4096, sizeof(chk3), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; ++i) {
        if (h[i] != 1.0 + i) ok = 0;
        if (h[64 + i] != 0.0) ok = 0;
        if (chk[i] != 0.0 || chk2[i] != 0.0) ok = 0;
        if (chk3[i] != 9.0) ok = 0;
    }
    printf("in-range load %g..%g, OOB load %g..%g, OOB store landed: %g / %g, in-range store %g -> %s\n", h[0], h[63],
           h[64], h[127], chk[0], chk2[0], chk3[0], ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
