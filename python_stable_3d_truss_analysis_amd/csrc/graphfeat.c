/* Graph features of a solved batch of 3D trusses (reference
 * TrussHeteroDataCreator, slientruss3d/data.py:11-282 and GetAngles, utils.py:105-113), straight
 * from the padded batch arrays into float32 feature tensors.  Same formulas, in double, as the
 * single-truss Python path (data.graph_arrays), rounded to float once; OpenMP over the batch.
 *
 *   joint_x  [B][nJ_max][FJ]  position / positionScale, load / forceScale,
 *                             (prior displacement / displaceScale,) isSupport
 *   member_x [B][nM_max][FM]  centre / positionScale, 4 direction features, length / positionScale,
 *                             (prior stress / forceScale,) (area)
 *   joint_y  [B][nJ_max][3]   displacement / displaceScale          (regression)
 *   member_y [B][nM_max]      stress / forceScale                   (regression)
 *   weight   [B]              sum of area * length * density
 * A result below 1e-10 in magnitude (a joint: all three components) counts as absent, as in the
 * reference's sparse result dicts (truss.py:344-359).  Padding rows are zero. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ZERO_EPS 1e-10

static inline void sparse_row(const double *v, double scale, float *out) {
    const int gone = fabs(v[0]) < ZERO_EPS && fabs(v[1]) < ZERO_EPS && fabs(v[2]) < ZERO_EPS;
    for (int a = 0; a < 3; ++a) out[a] = (float)((gone ? 0.0 : v[a]) / scale);
}

int trs_graph_features(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                       const double *A, const double *rho, const uint8_t *cbits, const double *loads,
                       const int32_t *nJ, const int32_t *nM, const double *u_act, const double *N_act,
                       const double *u_pri, const double *N_pri, double fixedArea, double forceScale,
                       double displaceScale, double positionScale, int regression, float *joint_x,
                       float *member_x, float *joint_y, float *member_y, double *weight) {
    const int has_prior = u_pri != NULL && N_pri != NULL;
    const int FJ = 7 + (has_prior ? 3 : 0);
    const int FM = 8 + (has_prior ? 1 : 0) + (regression ? 1 : 0);
    if (regression && (u_act == NULL || N_act == NULL || joint_y == NULL || member_y == NULL)) return -1;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const double *X = xyz + (size_t)b * nJ_max * 3;
        float *jx = joint_x + (size_t)b * nJ_max * FJ;
        float *mx = member_x + (size_t)b * nM_max * FM;
        memset(jx, 0, sizeof(float) * (size_t)nJ_max * FJ);
        memset(mx, 0, sizeof(float) * (size_t)nM_max * FM);
        if (regression) {
            memset(joint_y + (size_t)b * nJ_max * 3, 0, sizeof(float) * (size_t)nJ_max * 3);
            memset(member_y + (size_t)b * nM_max, 0, sizeof(float) * (size_t)nM_max);
        }
        for (int j = 0; j < nJ[b]; ++j) {
            const size_t jj = (size_t)b * nJ_max + j;
            float *o = jx + (size_t)j * FJ;
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
            if (regression) sparse_row(u_act + 3 * jj, displaceScale, joint_y + 3 * jj);
        }
        double w = 0.0;
        for (int m = 0; m < nM[b]; ++m) {
            const size_t mm = (size_t)b * nM_max + m;
            const double *p0 = X + 3 * conn[2 * mm], *p1 = X + 3 * conn[2 * mm + 1];
            float *o = mx + (size_t)m * FM;
            double e[3], len2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                e[a] = p1[a] - p0[a];
                len2 += e[a] * e[a];
                o[a] = (float)(0.5 * (p0[a] + p1[a]) / positionScale);
            }
            const double length = sqrt(len2);
            /* GetAngles: the lower end first */
            const int swap = !(p0[2] < p1[2]);
            double d[3];
            for (int a = 0; a < 3; ++a) d[a] = swap ? p0[a] - p1[a] : p1[a] - p0[a];
            const double full = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const double plan = sqrt(d[0] * d[0] + d[1] * d[1]);
            const int flat = fabs(plan) < ZERO_EPS;
            const double safe = flat ? 1.0 : plan;
            o[3] = (float)(plan / full);
            o[4] = (float)(d[2] / full);
            o[5] = (float)(flat ? 0.0 : d[1] / safe);
            o[6] = (float)(flat ? 0.0 : d[0] / safe);
            o[7] = (float)(length / positionScale);
            int k = 8;
            if (has_prior) {
                const double v = fabs(N_pri[mm]) < ZERO_EPS ? 0.0 : N_pri[mm];
                o[k++] = (float)(v / fixedArea / forceScale);
            }
            if (regression) {
                o[k++] = (float)A[mm];
                const double v = fabs(N_act[mm]) < ZERO_EPS ? 0.0 : N_act[mm];
                member_y[mm] = (float)(v / A[mm] / forceScale);
            }
            w += A[mm] * length * rho[mm];
        }
        weight[b] = w;
    }
    return 0;
}
