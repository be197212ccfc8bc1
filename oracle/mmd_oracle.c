/* Plain-C restatement of the reference's pairwise-distance + repulsive / bounded MMD loss.
 *
 * TEST INFRASTRUCTURE (oracle).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file; the product path never does.
 *
 * Follows /root/reference/GeneralTools/math_func.py:
 *   get_squared_dist  (mode 'xxxyyy')      :799-840   Gram form, diag from the Gram matrix, max(.,0)
 *   matrix_mean_wo_diagonal                :1064      (sum - trace) / (m (m-1)), cross block too
 *   mmd_g   (custom_weights)               :1312-1343
 *   mmd_g_bounded                          :1380-1422
 *   GANLoss._repulsive_mmd_g_(bounded_)    :2505-2550 sigma=1, bounds .25 / 4
 * x = generated scores, y = real scores (my_sngan.py:284-286).
 *
 * Gradients are the analytic derivative of exactly that expression (the reference obtains them
 * by TF autodiff, my_sngan.py:302-304): d max(a,b)/da = [a > b], d min(a,b)/da = [a < b];
 * fixtures avoid exact ties.
 *
 * Compiled twice by self-inclusion: mmd_oracle_f32 (the reference's precision) and
 * mmd_oracle_f64 (error-budget truth).  Build: gcc -O2 -shared -fPIC (see oracle/Makefile);
 * -ffast-math must NOT be used (summation order is part of the restatement).
 */
#ifndef MMD_ORACLE_BODY
#define MMD_ORACLE_BODY
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MMD_LOSS_REP 0
#define MMD_LOSS_RMB 1

#define REAL float
#define FN(name) name##_f32
#define EXP expf
#include "mmd_oracle.c"
#undef REAL
#undef FN
#undef EXP

#define REAL double
#define FN(name) name##_f64
#define EXP exp
#include "mmd_oracle.c"
#undef REAL
#undef FN
#undef EXP

#else  /* ------------------------------ templated body ------------------------------ */

/* gram[i*B+j] = sum_k a[i,k] b[j,k], accumulated in k order */
static void FN(gram)(const REAL *a, const REAL *b, int B, int d, REAL *out) {
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < B; ++j) {
            REAL acc = 0;
            for (int k = 0; k < d; ++k) acc += a[i * d + k] * b[j * d + k];
            out[i * B + j] = acc;
        }
}

static REAL FN(mean_wo_diag)(const REAL *m, int B) {
    /* reductions are accumulated in double in both builds: the reference's reduce_sum is a
     * blocked / tree reduction (Eigen, here torch), and a sequential fp32 running sum over B*B
     * terms that include B diagonal ones loses ~1e-3 of a small off-diagonal mean */
    double total = 0, tr = 0;
    for (int i = 0; i < B * B; ++i) total += (double)m[i];
    for (int i = 0; i < B; ++i) tr += (double)m[i * B + i];
    return (REAL)((total - tr) / ((double)B * ((double)B - 1.0)));
}

/* loss_type: 0 rep, 1 rmb.  Outputs (any may be NULL):
 *   losses[2]  = loss_gen, loss_dis
 *   stats[5]   = e_kxx, e_kxy, e_kyy, e_kxx_b, e_kyy_b   (the _b entries = plain ones for rep)
 *   dist[3*B*B]= dist_xx, dist_xy, dist_yy
 *   masks[3*B*B] (bytes) = dist_xx < lb, dist_xy > ub, dist_yy > ub
 *   grads[4*B*d] = dLg/dx, dLg/dy, dLd/dx, dLd/dy
 * returns 0, or -1 on bad arguments. */
int FN(mmd_oracle)(const REAL *x, const REAL *y, int B, int d, int loss_type,
                   REAL w0, REAL w1, REAL lb, REAL ub,
                   REAL *losses, REAL *stats, REAL *dist, unsigned char *masks, REAL *grads) {
    if (B < 2 || d < 1 || (loss_type != MMD_LOSS_REP && loss_type != MMD_LOSS_RMB)) return -1;
    if (w0 - w1 != (REAL)1.0) return -1;                                /* math_func.py:1340 */
    const size_t n = (size_t)B * B;
    REAL *buf = (REAL *)malloc(sizeof(REAL) * n * 9);
    if (!buf) return -1;
    REAL *dxx = buf, *dxy = buf + n, *dyy = buf + 2 * n;
    REAL *kxx = buf + 3 * n, *kxy = buf + 4 * n, *kyy = buf + 5 * n;
    REAL *gxx = buf + 6 * n, *gxy = buf + 7 * n, *gyy = buf + 8 * n;    /* dL/ddist scratch */

    FN(gram)(x, x, B, d, dxx);
    FN(gram)(x, y, B, d, dxy);
    FN(gram)(y, y, B, d, dyy);
    REAL *nx = (REAL *)malloc(sizeof(REAL) * 2 * B), *ny = nx + B;
    for (int i = 0; i < B; ++i) { nx[i] = dxx[i * B + i]; ny[i] = dyy[i * B + i]; }   /* diag_part */
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < B; ++j) {
            REAL a = nx[i] - (REAL)2.0 * dxx[i * B + j] + nx[j];
            REAL b = nx[i] - (REAL)2.0 * dxy[i * B + j] + ny[j];
            REAL c = ny[i] - (REAL)2.0 * dyy[i * B + j] + ny[j];
            dxx[i * B + j] = a > 0 ? a : 0;
            dxy[i * B + j] = b > 0 ? b : 0;
            dyy[i * B + j] = c > 0 ? c : 0;
        }
    if (dist) memcpy(dist, buf, sizeof(REAL) * 3 * n);
    if (masks)
        for (size_t i = 0; i < n; ++i) {
            masks[i] = dxx[i] < lb; masks[n + i] = dxy[i] > ub; masks[2 * n + i] = dyy[i] > ub;
        }
    for (size_t i = 0; i < n; ++i) {
        kxx[i] = EXP(-dxx[i] / (REAL)2.0); kxy[i] = EXP(-dxy[i] / (REAL)2.0); kyy[i] = EXP(-dyy[i] / (REAL)2.0);
    }
    const REAL e_kxx = FN(mean_wo_diag)(kxx, B), e_kxy = FN(mean_wo_diag)(kxy, B), e_kyy = FN(mean_wo_diag)(kyy, B);
    REAL e_kxx_b = e_kxx, e_kyy_b = e_kyy, e_kxy_b = e_kxy;
    const int yy_lower = (w1 > 0);                    /* math_func.py:1391-1394 */
    if (loss_type == MMD_LOSS_RMB) {
        for (size_t i = 0; i < n; ++i) {
            REAL a = dxx[i] > lb ? dxx[i] : lb;                                   /* :1386 */
            REAL c = yy_lower ? (dyy[i] > lb ? dyy[i] : lb) : (dyy[i] < ub ? dyy[i] : ub);
            gxx[i] = EXP(-a / (REAL)2.0); gyy[i] = EXP(-c / (REAL)2.0);
        }
        e_kxx_b = FN(mean_wo_diag)(gxx, B);
        e_kyy_b = FN(mean_wo_diag)(gyy, B);
        /* :1387-1390,1402: k_xy_b is bounded only if w0>0, and then e_kxy_b falls back to e_kxy;
         * if w0<0, k_xy_b = k_xy.  Either way e_kxy_b == e_kxy. */
        e_kxy_b = e_kxy;
    }
    const REAL loss_gen = e_kxx + e_kyy - (REAL)2.0 * e_kxy;                       /* :1341 */
    const REAL loss_dis = w0 * e_kxy_b - e_kxx_b - w1 * e_kyy_b;                    /* :1342,1421 */
    if (losses) { losses[0] = loss_gen; losses[1] = loss_dis; }
    if (stats) { stats[0] = e_kxx; stats[1] = e_kxy; stats[2] = e_kyy; stats[3] = e_kxx_b; stats[4] = e_kyy_b; }

    if (grads) {
        const REAL inv = (REAL)1.0 / ((REAL)B * ((REAL)B - (REAL)1.0));
        for (int pass = 0; pass < 2; ++pass) {            /* 0: loss_gen, 1: loss_dis */
            /* coefficients of e(K_xx), e(K_xy), e(K_yy) in this loss */
            const REAL cxx = pass == 0 ? (REAL)1.0 : (REAL)-1.0;
            const REAL cxy = pass == 0 ? (REAL)-2.0 : w0;
            const REAL cyy = pass == 0 ? (REAL)1.0 : -w1;
            const int bounded = (pass == 1 && loss_type == MMD_LOSS_RMB);
            for (int i = 0; i < B; ++i)
                for (int j = 0; j < B; ++j) {
                    const size_t t = (size_t)i * B + j;
                    const int off = (i != j);
                    REAL a = 0, b = 0, c = 0;
                    if (off) {
                        /* d exp(-D/2)/dD = -K/2, only where the clamp lets D through and D>0 */
                        int pa = dxx[t] > 0, pc = dyy[t] > 0, pb = dxy[t] > 0;
                        if (bounded) {
                            pa = pa && (dxx[t] > lb);
                            pc = pc && (yy_lower ? (dyy[t] > lb) : (dyy[t] < ub));
                        }
                        a = pa ? cxx * inv * (REAL)-0.5 * kxx[t] : 0;
                        b = pb ? cxy * inv * (REAL)-0.5 * kxy[t] : 0;
                        c = pc ? cyy * inv * (REAL)-0.5 * kyy[t] : 0;
                    }
                    gxx[t] = a; gxy[t] = b; gyy[t] = c;
                }
            REAL *gx = grads + (size_t)pass * 2 * B * d, *gy = gx + (size_t)B * d;
            memset(gx, 0, sizeof(REAL) * 2 * B * d);
            for (int i = 0; i < B; ++i)
                for (int j = 0; j < B; ++j) {
                    const REAL sxx = gxx[(size_t)i * B + j] + gxx[(size_t)j * B + i];
                    const REAL syy = gyy[(size_t)i * B + j] + gyy[(size_t)j * B + i];
                    const REAL sxy = gxy[(size_t)i * B + j];
                    for (int k = 0; k < d; ++k) {
                        gx[i * d + k] += sxx * (REAL)2.0 * (x[i * d + k] - x[j * d + k]);
                        gy[i * d + k] += syy * (REAL)2.0 * (y[i * d + k] - y[j * d + k]);
                        gx[i * d + k] += sxy * (REAL)2.0 * (x[i * d + k] - y[j * d + k]);
                        gy[j * d + k] += sxy * (REAL)2.0 * (y[j * d + k] - x[i * d + k]);
                    }
                }
        }
    }
    free(nx);
    free(buf);
    return 0;
}
#endif
