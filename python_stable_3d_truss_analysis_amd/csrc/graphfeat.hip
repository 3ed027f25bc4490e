// Graph features of a solved batch of 3D trusses ON THE DEVICE (reference
// TrussHeteroDataCreator, slientruss3d/data.py:116-282, and GetAngles, utils.py:105-113): straight from
// the resident batch arrays and the two solves' results into float32 feature tensors - the dataset path
// (BASELINE config 5) then downloads features, not double-precision results, and no host loop remains.
//
// Same formulas, in double, in the same operation order as the host version csrc/graphfeat.c, rounded to
// float once; floating-point contraction is off in this file, so the two are bit-identical (tested).
// One work-group per truss; the weight is summed by one thread in member order (as the host loop does).
#include "trs_common.h"
#include "../../include/trs_solver.h"

#pragma clang fp contract(off)

namespace {

constexpr double GF_ZERO_EPS = 1e-10;

__device__ __forceinline__ void sparse_row(const double* v, double scale, float* out) {
    const bool gone = fabs(v[0]) < GF_ZERO_EPS && fabs(v[1]) < GF_ZERO_EPS && fabs(v[2]) < GF_ZERO_EPS;
#pragma unroll
    for (int a = 0; a < 3; ++a) out[a] = (float)((gone ? 0.0 : v[a]) / scale);
}

__global__ __launch_bounds__(256) void trs_graph_features_kernel(
    const int nJ_max, const int nM_max, const double* __restrict__ xyz, const int* __restrict__ conn,
    const double* __restrict__ A, const double* __restrict__ rho, const uint8_t* __restrict__ cbits,
    const double* __restrict__ loads, const int* __restrict__ nJ, const int* __restrict__ nM,
    const double* __restrict__ u_act, const double* __restrict__ N_act, const double* __restrict__ u_pri,
    const double* __restrict__ N_pri, const double fixedArea, const double forceScale,
    const double displaceScale, const double positionScale, const int regression, float* __restrict__ joint_x,
    float* __restrict__ member_x, float* __restrict__ joint_y, float* __restrict__ member_y,
    double* __restrict__ weight, const long long* __restrict__ joint_off, const long long* __restrict__ member_off,
    int* __restrict__ conn_out) {
    extern __shared__ double wterm[];  // [nM_max] area * length * density per member
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool has_prior = u_pri != nullptr && N_pri != nullptr;
    const int FJ = 7 + (has_prior ? 3 : 0);
    const int FM = 8 + (has_prior ? 1 : 0) + (regression ? 1 : 0);
    const double* X = xyz + (size_t)b * nJ_max * 3;
    // PACKED output (trs_graph_features_packed): truss b's joints are rows joint_off[b] .. of the joint tensors,
    // its members rows member_off[b] .. of the member tensors - the rows of all trusses back to back, no padding
    const bool packed = joint_off != nullptr;
    const size_t jbase = packed ? (size_t)joint_off[b] : (size_t)b * nJ_max;
    const size_t mbase = packed ? (size_t)member_off[b] : (size_t)b * nM_max;
    float* jx = joint_x + jbase * FJ;
    float* mx = member_x + mbase * FM;
    const int joints = nJ[b], members = nM[b];
    for (int j = tid; j < (packed ? joints : nJ_max); j += 256) {
        const size_t jj = (size_t)b * nJ_max + j;   // row of the (padded) inputs
        const size_t jo = jbase + j;                // row of the outputs
        float* o = jx + (size_t)j * FJ;
        if (j >= joints) {  // padding rows are zero
            for (int k = 0; k < FJ; ++k) o[k] = 0.0f;
            if (regression)
                for (int a = 0; a < 3; ++a) joint_y[3 * jo + a] = 0.0f;
            continue;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[a] = (float)(X[3 * j + a] / positionScale);
            o[3 + a] = (float)(loads[3 * jj + a] / forceScale);
        }
        int k = 6;
        if (has_prior) {
            sparse_row(u_pri + 3 * jj, displaceScale, o + k);
            k += 3;
        }
        o[k] = (cbits[jj] & 7) ? 1.0f : 0.0f;
        if (regression) sparse_row(u_act + 3 * jj, displaceScale, joint_y + 3 * jo);
    }
    for (int m = tid; m < (packed ? members : nM_max); m += 256) {
        const size_t mm = (size_t)b * nM_max + m;
        const size_t mo = mbase + m;
        float* o = mx + (size_t)m * FM;
        if (m >= members) {
            for (int k = 0; k < FM; ++k) o[k] = 0.0f;
            if (regression) member_y[mo] = 0.0f;
            continue;
        }
        if (conn_out != nullptr) {  // the member's end joints: row 0 of the sample's j2m edge index
            conn_out[2 * mo] = conn[2 * mm];
            conn_out[2 * mo + 1] = conn[2 * mm + 1];
        }
        const double *p0 = X + 3 * conn[2 * mm], *p1 = X + 3 * conn[2 * mm + 1];
        double e[3], len2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            e[a] = p1[a] - p0[a];
            len2 += e[a] * e[a];
            o[a] = (float)(0.5 * (p0[a] + p1[a]) / positionScale);
        }
        const double length = sqrt(len2);
        const bool swap = !(p0[2] < p1[2]);  // GetAngles: the lower end first
        double d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) d[a] = swap ? p0[a] - p1[a] : p1[a] - p0[a];
        const double full = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const double plan = sqrt(d[0] * d[0] + d[1] * d[1]);
        const bool flat = fabs(plan) < GF_ZERO_EPS;
        const double safe = flat ? 1.0 : plan;
        o[3] = (float)(plan / full);
        o[4] = (float)(d[2] / full);
        o[5] = (float)(flat ? 0.0 : d[1] / safe);
        o[6] = (float)(flat ? 0.0 : d[0] / safe);
        o[7] = (float)(length / positionScale);
        int k = 8;
        if (has_prior) {
            const double v = fabs(N_pri[mm]) < GF_ZERO_EPS ? 0.0 : N_pri[mm];
            o[k++] = (float)(v / fixedArea / forceScale);
        }
        if (regression) {
            o[k++] = (float)A[mm];
            const double v = fabs(N_act[mm]) < GF_ZERO_EPS ? 0.0 : N_act[mm];
            member_y[mo] = (float)(v / A[mm] / forceScale);
        }
        wterm[m] = A[mm] * length * rho[mm];
    }
    __syncthreads();
    if (tid == 0) {  // in member order, as the reference's sum over members (truss.py:166-168)
        double w = 0.0;
        for (int m = 0; m < members; ++m) w += wterm[m];
        weight[b] = w;
    }
}

}  // namespace

static int graph_features_launch(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn,
                                 const double* A, const double* rho, const uint8_t* cbits,
                                 const double* loads, const int32_t* nJ, const int32_t* nM,
                                 const double* u_act, const double* N_act, const double* u_pri,
                                 const double* N_pri, double fixedArea, double forceScale,
                                 double displaceScale, double positionScale, int regression,
                                 float* joint_x, float* member_x, float* joint_y, float* member_y,
                                 double* weight, const long long* joint_off, const long long* member_off,
                                 int32_t* conn_out, void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max <= 0) return (int)hipErrorInvalidValue;
    if (regression && (u_act == nullptr || N_act == nullptr || joint_y == nullptr || member_y == nullptr))
        return (int)hipErrorInvalidValue;
    if (B == 0) return 0;
    const size_t lds = (size_t)nM_max * sizeof(double);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static const int lds_limit_set = (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(trs_graph_features_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        160 * 1024);
    (void)lds_limit_set;
    hipLaunchKernelGGL(trs_graph_features_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, nJ_max, nM_max,
                       xyz, conn, A, rho, cbits, loads, nJ, nM, u_act, N_act, u_pri, N_pri, fixedArea, forceScale,
                       displaceScale, positionScale, regression, joint_x, member_x, joint_y, member_y, weight,
                       joint_off, member_off, conn_out);
    return (int)hipGetLastError();
}

extern "C" int trs_graph_features_dev(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn,
                                      const double* A, const double* rho, const uint8_t* cbits,
                                      const double* loads, const int32_t* nJ, const int32_t* nM,
                                      const double* u_act, const double* N_act, const double* u_pri,
                                      const double* N_pri, double fixedArea, double forceScale,
                                      double displaceScale, double positionScale, int regression,
                                      float* joint_x, float* member_x, float* joint_y, float* member_y,
                                      double* weight, void* stream) {
    return graph_features_launch(B, nJ_max, nM_max, xyz, conn, A, rho, cbits, loads, nJ, nM, u_act, N_act, u_pri,
                                 N_pri, fixedArea, forceScale, displaceScale, positionScale, regression, joint_x,
                                 member_x, joint_y, member_y, weight, nullptr, nullptr, nullptr, stream);
}

extern "C" int trs_graph_features_packed(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn,
                                         const double* A, const double* rho, const uint8_t* cbits,
                                         const double* loads, const int32_t* nJ, const int32_t* nM,
                                         const double* u_act, const double* N_act, const double* u_pri,
                                         const double* N_pri, double fixedArea, double forceScale,
                                         double displaceScale, double positionScale, int regression,
                                         const int64_t* joint_off, const int64_t* member_off, float* joint_x,
                                         float* member_x, float* joint_y, float* member_y, int32_t* j2m_joint,
                                         double* weight, void* stream) {
    if (joint_off == nullptr || member_off == nullptr) return (int)hipErrorInvalidValue;
    return graph_features_launch(B, nJ_max, nM_max, xyz, conn, A, rho, cbits, loads, nJ, nM, u_act, N_act, u_pri,
                                 N_pri, fixedArea, forceScale, displaceScale, positionScale, regression, joint_x,
                                 member_x, joint_y, member_y, weight, reinterpret_cast<const long long*>(joint_off),
                                 reinterpret_cast<const long long*>(member_off), j2m_joint, stream);
}
