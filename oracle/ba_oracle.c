/*
 * ba_oracle.c — CPU restatement of the reference BA hot path (see ba_oracle.h header:
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED).  Plain C99, single thread.
 *
 * Follows, function by function (citations into /root/reference):
 *   projection / residual      src/cuda/cuda_block_solver.cu:379-424, 1060-1110
 *   Jacobians                  src/cuda/cuda_block_solver.cu:449-578
 *   quadratic form             src/cuda/cuda_block_solver.cu:1152-1220
 *   damping, Schur, back-subst src/cuda/cuda_block_solver.cu:1223-1345, 1419-1442
 *   update, scale              src/cuda/cuda_block_solver.cu:671-823, 1444-1490
 *   robust kernels             src/cuda/cuda_block_solver.cu:972-1027
 *   index / flag rules         src/optimisable_graph.hpp:84-126, 474-572, 642-661
 *   Hsc pattern                src/sparse_block_matrix.cpp:63-156
 *   step order                 src/block_solver.cpp:250-421
 *   LM control                 src/cuda_graph_optimisation.cpp:48-154
 * The sparse LL^T replaces closed-source cuSOLVER csrchol (src/cholesky.hpp:97-155): any
 * correct SPD factorisation is a valid restatement; failure rule "pivot <= 1e-14".
 */
#include "ba_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PD 6
#define LD 3
#define PIVOT_TOL 1e-14 /* src/cholesky.hpp:85 */

/* ------------------------------------------------------------------ small math ------ */

static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Xc = R(q) Xw, evaluated as Xw + 2w(qv x Xw) + 2 qv x (qv x Xw)   (.cu:379-394) */
static void rotate_q(const double* q, const double* Xw, double* Xc)
{
    double t1[3], t2[3];
    cross3(q, Xw, t1);
    t1[0] += t1[0];
    t1[1] += t1[1];
    t1[2] += t1[2];
    cross3(q, t1, t2);
    Xc[0] = Xw[0] + q[3] * t1[0] + t2[0];
    Xc[1] = Xw[1] + q[3] * t1[1] + t2[1];
    Xc[2] = Xw[2] + q[3] * t1[2] + t2[2];
}

/* column-major 3x3 from quaternion (.cu:449-478) */
static void quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
#define Rm(i, j) R[(j)*3 + (i)]
    Rm(0, 0) = 1 - (tyy + tzz);
    Rm(0, 1) = txy - twz;
    Rm(0, 2) = txz + twy;
    Rm(1, 0) = txy + twz;
    Rm(1, 1) = 1 - (txx + tzz);
    Rm(1, 2) = tyz - twx;
    Rm(2, 0) = txz - twy;
    Rm(2, 1) = tyz + twx;
    Rm(2, 2) = 1 - (txx + tyy);
}

double ba_rk_rho(int type, double delta, double x)
{
    const double d2 = delta * delta;
    if (type == BA_RK_TUKEY)
    {
        const double maxv = (1.0 / 3) * d2;
        const double u = 1 - x / d2;
        return x <= d2 ? maxv * (1 - u * u * u) : maxv;
    }
    if (type == BA_RK_CAUCHY)
    {
        const double r = 1.0 / d2;
        return d2 * log(r * x + 1.0);
    }
    if (type == BA_RK_HUBER) /* extension, g2o RobustKernelHuber: rho[0] = 2*delta*sqrt(e2) - delta^2 beyond delta */
        return x <= d2 ? x : 2.0 * delta * sqrt(x) - d2;
    return x;
}

double ba_rk_drho(int type, double delta, double x)
{
    const double d2 = delta * delta;
    if (type == BA_RK_TUKEY)
    {
        const double u = 1 - x / d2;
        return x <= d2 ? u * u : 0;
    }
    if (type == BA_RK_CAUCHY)
    {
        const double r = 1.0 / d2;
        return 1.0 / (r * x + 1.0);
    }
    if (type == BA_RK_HUBER)
        return x <= d2 ? 1.0 : delta / sqrt(x);
    return 1;
}

void ba_sym3_inv(const double* A, double* B)
{
#define A_(i, j) A[(j)*3 + (i)]
    const double A00 = A_(0, 0), A01 = A_(0, 1), A11 = A_(1, 1);
    const double A02 = A_(2, 0), A12 = A_(1, 2), A22 = A_(2, 2);
    const double det = A00 * A11 * A22 + A01 * A12 * A02 + A02 * A01 * A12 - A00 * A12 * A12 -
                       A02 * A11 * A02 - A01 * A01 * A22;
    const double id = 1 / det;
    const double B00 = id * (A11 * A22 - A12 * A12);
    const double B01 = id * (A02 * A12 - A01 * A22);
    const double B11 = id * (A00 * A22 - A02 * A02);
    const double B02 = id * (A01 * A12 - A02 * A11);
    const double B12 = id * (A02 * A01 - A00 * A12);
    const double B22 = id * (A00 * A11 - A01 * A01);
    B[0] = B00, B[3] = B01, B[6] = B02;
    B[1] = B01, B[4] = B11, B[7] = B12;
    B[2] = B02, B[5] = B12, B[8] = B22;
}

/* ------------------------------------------------------------------ edge ------------ */

void ba_edge_eval(const double* pose7, const double* Xw, const double* meas, int dim,
                  double omega, const double* cam, int rk_type, double rk_delta, double* e_out,
                  double* Xc_out, double* chi_out, double* JP, double* JL, double* w_out)
{
    const double* q = pose7;
    const double* t = pose7 + 4;
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], bf = cam[4];
    double Xc[3], proj[3], e[3] = {0, 0, 0};

    rotate_q(q, Xw, Xc); /* projectW2C .cu:396-402 */
    Xc[0] += t[0];
    Xc[1] += t[1];
    Xc[2] += t[2];

    { /* projectC2I .cu:410-424 */
        const double invZ = 1.0 / Xc[2];
        proj[0] = fx * invZ * Xc[0] + cx;
        proj[1] = fy * invZ * Xc[1] + cy;
        proj[2] = proj[0] - bf * invZ;
    }
    double sq = 0;
    for (int i = 0; i < dim; i++)
    {
        e[i] = proj[i] - meas[i]; /* sign: proj - meas, .cu:1104 */
        sq += e[i] * e[i];
    }
    const double x = omega * sq;
    if (e_out)
        for (int i = 0; i < dim; i++)
            e_out[i] = e[i];
    if (Xc_out)
        Xc_out[0] = Xc[0], Xc_out[1] = Xc[1], Xc_out[2] = Xc[2];
    if (chi_out)
        *chi_out = ba_rk_rho(rk_type, rk_delta, x);
    if (w_out)
        *w_out = omega * ba_rk_drho(rk_type, rk_delta, x);
    if (!JP && !JL)
        return;

    double R[9];
    quat_to_R(q, R);
    const double X = Xc[0], Y = Xc[1], Z = Xc[2];
    const double invZ = 1.0 / Z;
    double jp[18], jl[9];
#define JPm(i, j) jp[(j)*dim + (i)]
#define JLm(i, j) jl[(j)*dim + (i)]
    if (dim == 2)
    { /* .cu:491-528 */
        const double xx = invZ * X, yy = invZ * Y;
        const double fu_iz = fx * invZ, fv_iz = fy * invZ;
        for (int j = 0; j < 3; j++)
        {
            JLm(0, j) = -fu_iz * (Rm(0, j) - xx * Rm(2, j));
            JLm(1, j) = -fv_iz * (Rm(1, j) - yy * Rm(2, j));
        }
        JPm(0, 0) = +fx * xx * yy;
        JPm(0, 1) = -fx * (1 + xx * xx);
        JPm(0, 2) = +fx * yy;
        JPm(0, 3) = -fu_iz;
        JPm(0, 4) = 0;
        JPm(0, 5) = +fu_iz * xx;
        JPm(1, 0) = +fy * (1 + yy * yy);
        JPm(1, 1) = -fy * xx * yy;
        JPm(1, 2) = -fy * xx;
        JPm(1, 3) = 0;
        JPm(1, 4) = -fv_iz;
        JPm(1, 5) = +fv_iz * yy;
    }
    else
    { /* .cu:531-578 */
        const double iZZ = invZ * invZ;
        for (int j = 0; j < 3; j++)
        {
            JLm(0, j) = -fx * Rm(0, j) * invZ + fx * X * Rm(2, j) * iZZ;
            JLm(1, j) = -fy * Rm(1, j) * invZ + fy * Y * Rm(2, j) * iZZ;
            JLm(2, j) = JLm(0, j) - bf * Rm(2, j) * iZZ;
        }
        JPm(0, 0) = X * Y * iZZ * fx;
        JPm(0, 1) = -(1 + (X * X * iZZ)) * fx;
        JPm(0, 2) = Y * invZ * fx;
        JPm(0, 3) = -1 * invZ * fx;
        JPm(0, 4) = 0;
        JPm(0, 5) = X * iZZ * fx;
        JPm(1, 0) = (1 + Y * Y * iZZ) * fy;
        JPm(1, 1) = -X * Y * iZZ * fy;
        JPm(1, 2) = -X * invZ * fy;
        JPm(1, 3) = 0;
        JPm(1, 4) = -1 * invZ * fy;
        JPm(1, 5) = Y * iZZ * fy;
        JPm(2, 0) = JPm(0, 0) - bf * Y * iZZ;
        JPm(2, 1) = JPm(0, 1) + bf * X * iZZ;
        JPm(2, 2) = JPm(0, 2);
        JPm(2, 3) = JPm(0, 3);
        JPm(2, 4) = 0;
        JPm(2, 5) = JPm(0, 5) - bf * iZZ;
    }
    if (JP)
        memcpy(JP, jp, sizeof(double) * dim * 6);
    if (JL)
        memcpy(JL, jl, sizeof(double) * dim * 3);
}

/* ------------------------------------------------------------------ update ---------- */

static void rotmat_to_quat(const double* R, double* q)
{ /* .cu:721-754 */
    double t = Rm(0, 0) + Rm(1, 1) + Rm(2, 2);
    if (t > 0)
    {
        t = sqrt(t + 1);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (Rm(2, 1) - Rm(1, 2)) * t;
        q[1] = (Rm(0, 2) - Rm(2, 0)) * t;
        q[2] = (Rm(1, 0) - Rm(0, 1)) * t;
    }
    else
    {
        int i = 0;
        if (Rm(1, 1) > Rm(0, 0))
            i = 1;
        if (Rm(2, 2) > Rm(i, i))
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(Rm(i, i) - Rm(j, j) - Rm(k, k) + 1);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (Rm(k, j) - Rm(j, k)) * t;
        q[j] = (Rm(j, i) + Rm(i, j)) * t;
        q[k] = (Rm(k, i) + Rm(i, k)) * t;
    }
}

void ba_pose_update(double* pose7, const double* dx)
{ /* updateExp .cu:781-809 + updatePose .cu:811-823 */
    const double wx = dx[0], wy = dx[1], wz = dx[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    double O1[9], O2[9], R[9], V[9];
#define M_(A, i, j) A[(j)*3 + (i)]
    M_(O1, 0, 0) = 0, M_(O1, 0, 1) = -wz, M_(O1, 0, 2) = wy;
    M_(O1, 1, 0) = wz, M_(O1, 1, 1) = 0, M_(O1, 1, 2) = -wx;
    M_(O1, 2, 0) = -wy, M_(O1, 2, 1) = wx, M_(O1, 2, 2) = 0;
    {
        const double xx = wx * wx, yy = wy * wy, zz = wz * wz;
        const double xy = wx * wy, yz = wy * wz, zx = wz * wx;
        M_(O2, 0, 0) = -yy - zz, M_(O2, 0, 1) = xy, M_(O2, 0, 2) = zx;
        M_(O2, 1, 0) = xy, M_(O2, 1, 1) = -zz - xx, M_(O2, 1, 2) = yz;
        M_(O2, 2, 0) = zx, M_(O2, 2, 1) = yz, M_(O2, 2, 2) = -xx - yy;
    }
    double a1, a2, b1, b2;
    if (theta < 0.00001)
    {
        a1 = 1.0, a2 = 0.5, b1 = 0.5, b2 = 1.0 / 6;
    }
    else
    {
        a1 = sin(theta) / theta;
        a2 = (1 - cos(theta)) / (theta * theta);
        b1 = a2;
        b2 = (theta - sin(theta)) / (theta * theta * theta);
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
        {
            const double I = (i == j) ? 1.0 : 0.0;
            M_(R, i, j) = I + a1 * M_(O1, i, j) + a2 * M_(O2, i, j);
            M_(V, i, j) = I + b1 * M_(O1, i, j) + b2 * M_(O2, i, j);
        }
    double dq[4], dt[3];
    rotmat_to_quat(R, dq);
    for (int i = 0; i < 3; i++)
        dt[i] = M_(V, i, 0) * dx[3] + M_(V, i, 1) * dx[4] + M_(V, i, 2) * dx[5];

    double* q = pose7;
    double* t = pose7 + 4;
    double u[3];
    rotate_q(dq, t, u);
    t[0] = dt[0] + u[0];
    t[1] = dt[1] + u[1];
    t[2] = dt[2] + u[2];
    double r[4]; /* r = dq * q (.cu:756-762) */
    r[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
    r[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
    r[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
    r[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
    double invn = 1 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    if (r[3] < 0)
        invn = -invn; /* keep w >= 0 (.cu:764-775) */
    for (int i = 0; i < 4; i++)
        q[i] = invn * r[i];
}

/* ------------------------------------------------------------------ indexing -------- */

void ba_assign_indices(const ba_problem* p, int* pose_idx, int* lm_idx, int* np_free,
                       int* nl_free)
{
    int c = 0;
    for (int i = 0; i < p->n_poses; i++)
        if (!p->pose_fixed[i])
            pose_idx[i] = c++;
    *np_free = c;
    for (int i = 0; i < p->n_poses; i++)
        if (p->pose_fixed[i])
            pose_idx[i] = c++;
    c = 0;
    for (int i = 0; i < p->n_landmarks; i++)
        if (!p->lm_fixed[i])
            lm_idx[i] = c++;
    *nl_free = c;
    for (int i = 0; i < p->n_landmarks; i++)
        if (p->lm_fixed[i])
            lm_idx[i] = c++;
}

/* an edge is active when at least one endpoint is free (optimisable_graph.hpp:499-503) */
static int edge_active(const ba_problem* p, int e)
{
    return !p->pose_fixed[p->e_pose[e]] || !p->lm_fixed[p->e_lm[e]];
}

/* ------------------------------------------------------------------ errors ---------- */

double ba_compute_errors(const ba_problem* p, double* errors, double* Xcs)
{
    double chi_mono = 0, chi_stereo = 0; /* per edge set, then added (block_solver.cpp:256-267) */
#ifdef BA_OMP /* timing-only build (cpu_baseline, all cores): the summation order is not fixed */
#pragma omp parallel for reduction(+ : chi_mono, chi_stereo) schedule(static)
#endif
    for (int e = 0; e < p->n_edges; e++)
    {
        if (!edge_active(p, e))
            continue;
        const int st = p->e_stereo[e] ? 1 : 0;
        double er[3] = {0, 0, 0}, xc[3], chi;
        ba_edge_eval(p->pose + 7 * p->e_pose[e], p->lm + 3 * p->e_lm[e], p->e_meas + 3 * e,
                     st ? 3 : 2, p->e_omega[e], p->e_cam + 5 * e, p->rk_type, p->rk_delta, er, xc,
                     &chi, 0, 0, 0);
        if (st)
            chi_stereo += chi;
        else
            chi_mono += chi;
        if (errors)
            memcpy(errors + 3 * e, er, sizeof er);
        if (Xcs)
            memcpy(Xcs + 3 * e, xc, sizeof xc);
    }
    return chi_mono + chi_stereo;
}

/* ------------------------------------------------------------------ build ----------- */

double ba_build_system(const ba_problem* p, double* Hpp, double* bp, double* Hll, double* bl,
                       double* Hpl)
{
    int npf, nlf;
    int* pidx = (int*)malloc(sizeof(int) * (p->n_poses + 1));
    int* lidx = (int*)malloc(sizeof(int) * (p->n_landmarks + 1));
    ba_assign_indices(p, pidx, lidx, &npf, &nlf);
    if (Hpp)
        memset(Hpp, 0, sizeof(double) * 36 * npf);
    if (bp)
        memset(bp, 0, sizeof(double) * 6 * npf);
    if (Hll)
        memset(Hll, 0, sizeof(double) * 9 * nlf);
    if (bl)
        memset(bl, 0, sizeof(double) * 3 * nlf);
    if (Hpl)
        memset(Hpl, 0, sizeof(double) * 18 * p->n_edges);
    double chi_mono = 0, chi_stereo = 0;
#ifdef BA_OMP
    /* pose blocks are hit by hundreds of edges each: every thread sums into its own copy, the
     * copies are added at the end; landmark blocks (a few edges each) use atomic adds */
#pragma omp parallel reduction(+ : chi_mono, chi_stereo)
    {
    double* Hpp_shared = Hpp;
    double* bp_shared = bp;
    double* Hpp_t = Hpp ? (double*)calloc((size_t)36 * (npf + 1), sizeof(double)) : 0;
    double* bp_t = bp ? (double*)calloc((size_t)6 * (npf + 1), sizeof(double)) : 0;
#define Hpp Hpp_t
#define bp bp_t
#define BA_ATOMIC _Pragma("omp atomic")
#pragma omp for schedule(static)
#else
#define BA_ATOMIC
#endif
    for (int e = 0; e < p->n_edges; e++)
    {
        if (!edge_active(p, e))
            continue;
        const int ip = p->e_pose[e], il = p->e_lm[e];
        const int st = p->e_stereo[e] ? 1 : 0, dim = st ? 3 : 2;
        double er[3], chi, w, JP[18], JL[9];
        ba_edge_eval(p->pose + 7 * ip, p->lm + 3 * il, p->e_meas + 3 * e, dim, p->e_omega[e],
                     p->e_cam + 5 * e, p->rk_type, p->rk_delta, er, 0, &chi, JP, JL, &w);
        if (st)
            chi_stereo += chi;
        else
            chi_mono += chi;
        const int pf = !p->pose_fixed[ip], lf = !p->lm_fixed[il];
        /* C(r,c) = w * sum_m A(m,r) B(m,c), col-major (MatTMulMat .cu:186-198) */
        if (pf)
        {
            double* H = Hpp ? Hpp + 36 * pidx[ip] : 0;
            double* b = bp ? bp + 6 * pidx[ip] : 0;
            for (int c = 0; c < 6; c++)
            {
                if (H)
                    for (int r = 0; r < 6; r++)
                    {
                        double s = 0;
                        for (int m = 0; m < dim; m++)
                            s += JP[r * dim + m] * JP[c * dim + m];
                        H[c * 6 + r] += w * s;
                    }
                if (b)
                {
                    double s = 0;
                    for (int m = 0; m < dim; m++)
                        s += JP[c * dim + m] * er[m];
                    b[c] += w * s;
                }
            }
        }
        if (lf)
        {
            double* H = Hll ? Hll + 9 * lidx[il] : 0;
            double* b = bl ? bl + 3 * lidx[il] : 0;
            for (int c = 0; c < 3; c++)
            {
                if (H)
                    for (int r = 0; r < 3; r++)
                    {
                        double s = 0;
                        for (int m = 0; m < dim; m++)
                            s += JL[r * dim + m] * JL[c * dim + m];
                        BA_ATOMIC
                        H[c * 3 + r] += w * s;
                    }
                if (b)
                {
                    double s = 0;
                    for (int m = 0; m < dim; m++)
                        s += JL[c * dim + m] * er[m];
                    BA_ATOMIC
                    b[c] += w * s;
                }
            }
        }
        if (pf && lf && Hpl)
        {
            double* H = Hpl + 18 * e;
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++)
                {
                    double s = 0;
                    for (int m = 0; m < dim; m++)
                        s += JP[r * dim + m] * JL[c * dim + m];
                    H[c * 6 + r] = w * s;
                }
        }
    }
#ifdef BA_OMP
#undef Hpp
#undef bp
#pragma omp critical
    {
        if (Hpp_t)
            for (int i = 0; i < 36 * npf; i++)
                Hpp_shared[i] += Hpp_t[i];
        if (bp_t)
            for (int i = 0; i < 6 * npf; i++)
                bp_shared[i] += bp_t[i];
    }
    free(Hpp_t);
    free(bp_t);
    } /* omp parallel */
#endif
    free(pidx);
    free(lidx);
    return chi_mono + chi_stereo;
}

/* ------------------------------------------------------------------ dense LL^T ------ */

/* in-place lower Cholesky of n x n column-major; returns 0 on pivot <= tol */
static int dense_chol(int n, double* A)
{
    for (int j = 0; j < n; j++)
    {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++)
            d -= A[k * n + j] * A[k * n + j];
        if (!(d > PIVOT_TOL))
            return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++)
        {
            double s = A[j * n + i];
            for (int k = 0; k < j; k++)
                s -= A[k * n + i] * A[k * n + j];
            A[j * n + i] = s / d;
        }
    }
    return 1;
}

static void dense_chol_solve(int n, const double* L, double* b)
{
    for (int i = 0; i < n; i++)
    {
        double s = b[i];
        for (int k = 0; k < i; k++)
            s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--)
    {
        double s = b[i];
        for (int k = i + 1; k < n; k++)
            s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* ------------------------------------------------------------------ sparse block LL^T */

typedef struct
{
    int n, cap;
    int* row;
    double* blk; /* 36 per entry, col-major 6x6 */
} bcol;

static void bcol_push(bcol* c, int row, const double* b)
{
    if (c->n == c->cap)
    {
        c->cap = c->cap ? 2 * c->cap : 8;
        c->row = (int*)realloc(c->row, sizeof(int) * c->cap);
        c->blk = (double*)realloc(c->blk, sizeof(double) * 36 * c->cap);
    }
    c->row[c->n] = row;
    memcpy(c->blk + 36 * c->n, b, sizeof(double) * 36);
    c->n++;
}

/* 6x6 helpers, col-major */
static int chol6(double* A) { return dense_chol(6, A); }
/* Y = L^{-1} X (L lower 6x6), in place on X (6x6) */
static void trsm6_lower_left(const double* L, double* X)
{
    for (int c = 0; c < 6; c++)
        for (int i = 0; i < 6; i++)
        {
            double s = X[c * 6 + i];
            for (int k = 0; k < i; k++)
                s -= L[k * 6 + i] * X[c * 6 + k];
            X[c * 6 + i] = s / L[i * 6 + i];
        }
}

/* minimum-degree ordering on the block graph (simple explicit elimination graph) */
static void min_degree_order(int n, const int* rowptr, const int* colind, int* perm)
{
    /* adjacency as growable sorted arrays */
    int** adj = (int**)calloc(n, sizeof(int*));
    int* deg = (int*)calloc(n, sizeof(int));
    int* cap = (int*)calloc(n, sizeof(int));
    char* done = (char*)calloc(n, 1);
    int* mark = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++)
        mark[i] = -1;
#define ADJ_PUSH(a, b)                                              \
    do                                                              \
    {                                                               \
        if (deg[a] == cap[a])                                       \
        {                                                           \
            cap[a] = cap[a] ? 2 * cap[a] : 8;                       \
            adj[a] = (int*)realloc(adj[a], sizeof(int) * cap[a]);   \
        }                                                           \
        adj[a][deg[a]++] = (b);                                     \
    } while (0)
    for (int r = 0; r < n; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
        {
            const int c = colind[k];
            if (c != r)
            {
                ADJ_PUSH(r, c);
                ADJ_PUSH(c, r);
            }
        }
    for (int step = 0; step < n; step++)
    {
        int best = -1;
        for (int i = 0; i < n; i++)
            if (!done[i] && (best < 0 || deg[i] < deg[best]))
                best = i;
        perm[step] = best;
        done[best] = 1;
        /* make neighbours a clique, drop `best` */
        const int nb = deg[best];
        for (int a = 0; a < nb; a++)
        {
            const int u = adj[best][a];
            /* remove best from u, mark existing neighbours */
            int w = 0;
            for (int k = 0; k < deg[u]; k++)
                if (adj[u][k] != best)
                {
                    adj[u][w++] = adj[u][k];
                    mark[adj[u][k]] = u;
                }
            deg[u] = w;
            mark[u] = u;
            for (int b = 0; b < nb; b++)
            {
                const int v = adj[best][b];
                if (mark[v] != u)
                {
                    ADJ_PUSH(u, v);
                    mark[v] = u;
                }
            }
        }
        free(adj[best]);
        adj[best] = 0;
        deg[best] = 0;
    }
    free(adj);
    free(deg);
    free(cap);
    free(done);
    free(mark);
}

int ba_bsr_chol_solve(int nb, const int* rowptr, const int* colind, const double* vals,
                      const double* b, double* x)
{
    int ok = 1;
    int* perm = (int*)malloc(sizeof(int) * nb); /* perm[new] = old */
    int* inv = (int*)malloc(sizeof(int) * nb);
    min_degree_order(nb, rowptr, colind, perm);
    for (int i = 0; i < nb; i++)
        inv[perm[i]] = i;

    /* permuted upper-triangular columns: for new column k, entries (i<=k, block A(i,k)) */
    int* ccnt = (int*)calloc(nb + 1, sizeof(int));
    for (int r = 0; r < nb; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
        {
            const int a = inv[r], c = inv[colind[k]];
            ccnt[(a > c ? a : c) + 1]++;
        }
    for (int i = 0; i < nb; i++)
        ccnt[i + 1] += ccnt[i];
    const int nnz = ccnt[nb];
    int* crow = (int*)malloc(sizeof(int) * (nnz + 1));
    int* csrc = (int*)malloc(sizeof(int) * (nnz + 1)); /* source block, sign = transpose */
    int* cpos = (int*)malloc(sizeof(int) * (nb + 1));
    memcpy(cpos, ccnt, sizeof(int) * (nb + 1));
    for (int r = 0; r < nb; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
        {
            const int a = inv[r], c = inv[colind[k]];
            /* stored block is A(r, colind) i.e. A(a,c) in new numbering */
            if (a <= c)
            {
                const int q = cpos[c]++;
                crow[q] = a;
                csrc[q] = k + 1; /* A(a,c) = block as stored */
            }
            else
            {
                const int q = cpos[a]++;
                crow[q] = c;
                csrc[q] = -(k + 1); /* A(c,a) = stored^T */
            }
        }

    /* elimination tree */
    int* parent = (int*)malloc(sizeof(int) * nb);
    int* anc = (int*)malloc(sizeof(int) * nb);
    for (int k = 0; k < nb; k++)
    {
        parent[k] = -1;
        anc[k] = -1;
        for (int q = ccnt[k]; q < ccnt[k + 1]; q++)
        {
            int i = crow[q];
            while (i != -1 && i < k)
            {
                const int nx = anc[i];
                anc[i] = k;
                if (nx == -1)
                    parent[i] = k;
                i = nx;
            }
        }
    }

    bcol* Lc = (bcol*)calloc(nb, sizeof(bcol));
    double* Ld = (double*)malloc(sizeof(double) * 36 * nb); /* diagonal blocks of L */
    double* X = (double*)calloc((size_t)36 * nb, sizeof(double));
    int* flag = (int*)malloc(sizeof(int) * nb);
    int* stack = (int*)malloc(sizeof(int) * nb);
    int* path = (int*)malloc(sizeof(int) * nb);
    for (int i = 0; i < nb; i++)
        flag[i] = -1;

    for (int k = 0; k < nb && ok; k++)
    {
        double D[36];
        memset(D, 0, sizeof D);
        int top = nb;
        flag[k] = k;
        for (int q = ccnt[k]; q < ccnt[k + 1]; q++)
        {
            const int i0 = crow[q];
            const int s = csrc[q];
            const double* src = vals + 36 * ((s > 0 ? s : -s) - 1);
            double* dst = (i0 == k) ? D : X + 36 * i0;
            if (s > 0)
                for (int t = 0; t < 36; t++)
                    dst[t] += src[t];
            else
                for (int c = 0; c < 6; c++)
                    for (int r = 0; r < 6; r++)
                        dst[c * 6 + r] += src[r * 6 + c];
            /* ereach: climb the etree from i0 */
            int len = 0;
            for (int i = i0; i != -1 && i < k && flag[i] != k; i = parent[i])
            {
                path[len++] = i;
                flag[i] = k;
            }
            while (len > 0)
                stack[--top] = path[--len];
        }
        /* A may store only one triangle: symmetrise the diagonal block from its upper part */
        for (int c = 0; c < 6; c++)
            for (int r = c + 1; r < 6; r++)
                D[c * 6 + r] = D[r * 6 + c];
        for (; top < nb; top++)
        {
            const int i = stack[top];
            double* Xi = X + 36 * i; /* block (row i, col k) */
            trsm6_lower_left(Ld + 36 * i, Xi); /* Y = Lii^{-1} Xi = L(k,i)^T */
            bcol* col = &Lc[i];
            for (int e = 0; e < col->n; e++)
            { /* X_r -= L(r,i) * Y */
                double* Xr = X + 36 * col->row[e];
                const double* Lri = col->blk + 36 * e;
                for (int c = 0; c < 6; c++)
                    for (int r = 0; r < 6; r++)
                    {
                        double s = 0;
                        for (int m = 0; m < 6; m++)
                            s += Lri[m * 6 + r] * Xi[c * 6 + m];
                        Xr[c * 6 + r] -= s;
                    }
            }
            double Lki[36]; /* L(k,i) = Y^T ;  D -= Y^T Y */
            for (int c = 0; c < 6; c++)
                for (int r = 0; r < 6; r++)
                {
                    Lki[c * 6 + r] = Xi[r * 6 + c];
                    double s = 0;
                    for (int m = 0; m < 6; m++)
                        s += Xi[r * 6 + m] * Xi[c * 6 + m];
                    D[c * 6 + r] -= s;
                }
            bcol_push(col, k, Lki);
            memset(Xi, 0, sizeof(double) * 36);
        }
        if (!chol6(D))
            ok = 0;
        memcpy(Ld + 36 * k, D, sizeof D);
    }

    if (ok)
    {
        double* y = (double*)malloc(sizeof(double) * 6 * nb);
        for (int i = 0; i < nb; i++)
            memcpy(y + 6 * i, b + 6 * perm[i], sizeof(double) * 6);
        for (int i = 0; i < nb; i++)
        { /* forward */
            double* yi = y + 6 * i;
            const double* L = Ld + 36 * i;
            for (int r = 0; r < 6; r++)
            {
                double s = yi[r];
                for (int m = 0; m < r; m++)
                    s -= L[m * 6 + r] * yi[m];
                yi[r] = s / L[r * 6 + r];
            }
            for (int e = 0; e < Lc[i].n; e++)
            {
                double* yr = y + 6 * Lc[i].row[e];
                const double* B = Lc[i].blk + 36 * e;
                for (int r = 0; r < 6; r++)
                {
                    double s = 0;
                    for (int m = 0; m < 6; m++)
                        s += B[m * 6 + r] * yi[m];
                    yr[r] -= s;
                }
            }
        }
        for (int i = nb - 1; i >= 0; i--)
        { /* backward */
            double* yi = y + 6 * i;
            for (int e = 0; e < Lc[i].n; e++)
            {
                const double* yr = y + 6 * Lc[i].row[e];
                const double* B = Lc[i].blk + 36 * e;
                for (int c = 0; c < 6; c++)
                {
                    double s = 0;
                    for (int m = 0; m < 6; m++)
                        s += B[c * 6 + m] * yr[m];
                    yi[c] -= s;
                }
            }
            const double* L = Ld + 36 * i;
            for (int r = 5; r >= 0; r--)
            {
                double s = yi[r];
                for (int m = r + 1; m < 6; m++)
                    s -= L[r * 6 + m] * yi[m];
                yi[r] = s / L[r * 6 + r];
            }
        }
        for (int i = 0; i < nb; i++)
            memcpy(x + 6 * perm[i], y + 6 * i, sizeof(double) * 6);
        free(y);
    }
    for (int i = 0; i < nb; i++)
    {
        free(Lc[i].row);
        free(Lc[i].blk);
    }
    free(Lc), free(Ld), free(X), free(flag), free(stack), free(path);
    free(parent), free(anc), free(crow), free(csrc), free(cpos), free(ccnt), free(perm), free(inv);
    return ok;
}

/* ------------------------------------------------------------------ Schur ----------- */

typedef struct
{
    int npf, nlf;
    int *pidx, *lidx;
    /* landmark -> its free-free edges, sorted by pose index (Hpl CSC, .cu:1571-1604) */
    int* colptr;
    int* coledge;
    /* Hsc upper block CSR (sparse_block_matrix.cpp:63-156) */
    int nblocks;
    int* rowptr;
    int* colind;
} ba_struct;

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

static const ba_problem* g_sort_p;
static const int* g_sort_pidx;
static int cmp_edge_by_pose(const void* a, const void* b)
{
    const int ea = *(const int*)a, eb = *(const int*)b;
    const int pa = g_sort_pidx[g_sort_p->e_pose[ea]], pb = g_sort_pidx[g_sort_p->e_pose[eb]];
    return pa != pb ? pa - pb : ea - eb;
}

static void build_struct(const ba_problem* p, ba_struct* s)
{
    s->pidx = (int*)malloc(sizeof(int) * (p->n_poses + 1));
    s->lidx = (int*)malloc(sizeof(int) * (p->n_landmarks + 1));
    ba_assign_indices(p, s->pidx, s->lidx, &s->npf, &s->nlf);
    const int nlf = s->nlf, npf = s->npf;
    s->colptr = (int*)calloc(nlf + 2, sizeof(int));
    int nff = 0;
    for (int e = 0; e < p->n_edges; e++)
        if (!p->pose_fixed[p->e_pose[e]] && !p->lm_fixed[p->e_lm[e]])
        {
            s->colptr[s->lidx[p->e_lm[e]] + 1]++;
            nff++;
        }
    for (int l = 0; l < nlf; l++)
        s->colptr[l + 1] += s->colptr[l];
    s->coledge = (int*)malloc(sizeof(int) * (nff + 1));
    int* pos = (int*)malloc(sizeof(int) * (nlf + 1));
    memcpy(pos, s->colptr, sizeof(int) * (nlf + 1));
    for (int e = 0; e < p->n_edges; e++)
        if (!p->pose_fixed[p->e_pose[e]] && !p->lm_fixed[p->e_lm[e]])
            s->coledge[pos[s->lidx[p->e_lm[e]]]++] = e;
    g_sort_p = p;
    g_sort_pidx = s->pidx;
    for (int l = 0; l < nlf; l++)
        qsort(s->coledge + s->colptr[l], s->colptr[l + 1] - s->colptr[l], sizeof(int),
              cmp_edge_by_pose);
    free(pos);

    /* Hsc pattern: per row a growing sorted-unique list of columns */
    int** rows = (int**)calloc(npf > 0 ? npf : 1, sizeof(int*));
    int* rn = (int*)calloc(npf + 1, sizeof(int));
    int* rc = (int*)calloc(npf + 1, sizeof(int));
    for (int l = 0; l < nlf; l++)
        for (int a = s->colptr[l]; a < s->colptr[l + 1]; a++)
        {
            const int ra = s->pidx[p->e_pose[s->coledge[a]]];
            for (int b = a; b < s->colptr[l + 1]; b++)
            {
                const int cb = s->pidx[p->e_pose[s->coledge[b]]];
                if (rn[ra] == rc[ra])
                {
                    rc[ra] = rc[ra] ? 2 * rc[ra] : 16;
                    rows[ra] = (int*)realloc(rows[ra], sizeof(int) * rc[ra]);
                }
                rows[ra][rn[ra]++] = cb;
            }
            /* compact occasionally */
            if (rn[ra] > 4096)
            {
                qsort(rows[ra], rn[ra], sizeof(int), cmp_int);
                int w = 0;
                for (int k = 0; k < rn[ra]; k++)
                    if (!w || rows[ra][k] != rows[ra][w - 1])
                        rows[ra][w++] = rows[ra][k];
                rn[ra] = w;
            }
        }
    s->rowptr = (int*)calloc(npf + 1, sizeof(int));
    for (int r = 0; r < npf; r++)
    {
        /* every free pose gets its diagonal block (the reference assumes it exists,
         * sparse_block_matrix.h:95) */
        if (rn[r] == rc[r])
        {
            rc[r] = rc[r] ? 2 * rc[r] : 16;
            rows[r] = (int*)realloc(rows[r], sizeof(int) * rc[r]);
        }
        rows[r][rn[r]++] = r;
        qsort(rows[r], rn[r], sizeof(int), cmp_int);
        int w = 0;
        for (int k = 0; k < rn[r]; k++)
            if (!w || rows[r][k] != rows[r][w - 1])
                rows[r][w++] = rows[r][k];
        rn[r] = w;
        s->rowptr[r + 1] = s->rowptr[r] + w;
    }
    s->nblocks = s->rowptr[npf];
    s->colind = (int*)malloc(sizeof(int) * (s->nblocks + 1));
    for (int r = 0; r < npf; r++)
    {
        memcpy(s->colind + s->rowptr[r], rows[r], sizeof(int) * rn[r]);
        free(rows[r]);
    }
    free(rows), free(rn), free(rc);
}

static void free_struct(ba_struct* s)
{
    free(s->pidx), free(s->lidx), free(s->colptr), free(s->coledge), free(s->rowptr),
        free(s->colind);
}

static int find_block(const ba_struct* s, int r, int c)
{
    int lo = s->rowptr[r], hi = s->rowptr[r + 1] - 1;
    while (lo <= hi)
    {
        const int mid = (lo + hi) >> 1;
        if (s->colind[mid] == c)
            return mid;
        if (s->colind[mid] < c)
            lo = mid + 1;
        else
            hi = mid - 1;
    }
    return -1;
}

/* Damped Schur system in upper block CSR. Hpp/Hll are damped copies. */
static void schur_bsr(const ba_problem* p, const ba_struct* s, const double* Hpp,
                      const double* bp, const double* Hll_damped, const double* bl,
                      const double* Hpl, double* Hsc, double* bsc, double* invHll)
{
    const int npf = s->npf, nlf = s->nlf;
    memset(Hsc, 0, sizeof(double) * 36 * s->nblocks);
    for (int r = 0; r < npf; r++) /* initializeHschur .cu:1316-1325 */
        memcpy(Hsc + 36 * s->rowptr[r], Hpp + 36 * r, sizeof(double) * 36);
    memcpy(bsc, bp, sizeof(double) * 6 * npf); /* bp.copyTo(bsc) .cu:2017 */
#ifdef BA_OMP /* like the reference's kernels: one landmark per thread, atomic adds into bsc / Hsc */
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int l = 0; l < nlf; l++)
    {
        double* iH = invHll + 9 * l;
        ba_sym3_inv(Hll_damped + 9 * l, iH);
        const int a0 = s->colptr[l], a1 = s->colptr[l + 1];
        for (int a = a0; a < a1; a++)
        {
            const int ea = s->coledge[a];
            const int ra = s->pidx[p->e_pose[ea]];
            const double* Ha = Hpl + 18 * ea;
            double T[18]; /* T = Hpl_a * invHll (6x3) */
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++)
                {
                    double v = 0;
                    for (int m = 0; m < 3; m++)
                        v += Ha[m * 6 + r] * iH[c * 3 + m];
                    T[c * 6 + r] = v;
                }
            for (int r = 0; r < 6; r++)
            { /* bsc -= T * bl */
                double v = 0;
                for (int m = 0; m < 3; m++)
                    v += T[m * 6 + r] * bl[3 * l + m];
                BA_ATOMIC
                bsc[6 * ra + r] -= v;
            }
            for (int b = a; b < a1; b++)
            { /* Hsc(ra, rb) -= T * Hpl_b^T  (.cu:1327-1345) */
                const int eb = s->coledge[b];
                const int rb = s->pidx[p->e_pose[eb]];
                const double* Hb = Hpl + 18 * eb;
                double* dst = Hsc + 36 * find_block(s, ra, rb);
                for (int c = 0; c < 6; c++)
                    for (int r = 0; r < 6; r++)
                    {
                        double v = 0;
                        for (int m = 0; m < 3; m++)
                            v += T[m * 6 + r] * Hb[m * 6 + c];
                        BA_ATOMIC
                        dst[c * 6 + r] -= v;
                    }
            }
        }
    }
}

static void add_lambda(int n, int dim, double* H, double lambda)
{ /* .cu:1256-1269 */
    for (int i = 0; i < n; i++)
        for (int k = 0; k < dim; k++)
            H[i * dim * dim + k * dim + k] += lambda;
}

void ba_schur_dense(const ba_problem* p, double lambda, double* Hsc_dense, double* bsc)
{
    ba_struct s;
    build_struct(p, &s);
    const int n = 6 * s.npf;
    double* Hpp = (double*)malloc(sizeof(double) * 36 * (s.npf + 1));
    double* bp = (double*)malloc(sizeof(double) * 6 * (s.npf + 1));
    double* Hll = (double*)malloc(sizeof(double) * 9 * (s.nlf + 1));
    double* bl = (double*)malloc(sizeof(double) * 3 * (s.nlf + 1));
    double* Hpl = (double*)malloc(sizeof(double) * 18 * (p->n_edges + 1));
    double* iH = (double*)malloc(sizeof(double) * 9 * (s.nlf + 1));
    double* H = (double*)malloc(sizeof(double) * 36 * (s.nblocks + 1));
    ba_build_system(p, Hpp, bp, Hll, bl, Hpl);
    add_lambda(s.npf, 6, Hpp, lambda);
    add_lambda(s.nlf, 3, Hll, lambda);
    schur_bsr(p, &s, Hpp, bp, Hll, bl, Hpl, H, bsc, iH);
    memset(Hsc_dense, 0, sizeof(double) * n * n);
    for (int r = 0; r < s.npf; r++)
        for (int k = s.rowptr[r]; k < s.rowptr[r + 1]; k++)
        {
            const int c = s.colind[k];
            for (int j = 0; j < 6; j++)
                for (int i = 0; i < 6; i++)
                {
                    const double v = H[36 * k + j * 6 + i];
                    Hsc_dense[(size_t)(6 * c + j) * n + 6 * r + i] = v;
                    if (c != r)
                        Hsc_dense[(size_t)(6 * r + i) * n + 6 * c + j] = v;
                }
        }
    /* diagonal blocks: keep exactly symmetric from the upper triangle like the CSR mirror */
    free(Hpp), free(bp), free(Hll), free(bl), free(Hpl), free(iH), free(H);
    free_struct(&s);
}

/* one damped solve given prebuilt (undamped) system */
static int solve_with(const ba_problem* p, const ba_struct* s, double lambda, int use_dense,
                      const double* Hpp0, const double* bp, const double* Hll0, const double* bl,
                      const double* Hpl, double* dxp, double* dxl)
{
    const int npf = s->npf, nlf = s->nlf;
    double* Hpp = (double*)malloc(sizeof(double) * 36 * (npf + 1));
    double* Hll = (double*)malloc(sizeof(double) * 9 * (nlf + 1));
    double* iH = (double*)malloc(sizeof(double) * 9 * (nlf + 1));
    double* Hsc = (double*)malloc(sizeof(double) * 36 * (s->nblocks + 1));
    double* bsc = (double*)malloc(sizeof(double) * 6 * (npf + 1));
    memcpy(Hpp, Hpp0, sizeof(double) * 36 * npf);
    memcpy(Hll, Hll0, sizeof(double) * 9 * nlf);
    add_lambda(npf, 6, Hpp, lambda);
    add_lambda(nlf, 3, Hll, lambda);
    schur_bsr(p, s, Hpp, bp, Hll, bl, Hpl, Hsc, bsc, iH);
    int ok;
    if (use_dense)
    {
        const int n = 6 * npf;
        double* D = (double*)calloc((size_t)n * n, sizeof(double));
        for (int r = 0; r < npf; r++)
            for (int k = s->rowptr[r]; k < s->rowptr[r + 1]; k++)
            {
                const int c = s->colind[k];
                for (int j = 0; j < 6; j++)
                    for (int i = 0; i < 6; i++)
                    {
                        const double v = Hsc[36 * k + j * 6 + i];
                        /* lower triangle of the dense matrix: entry (row 6c+j, col 6r+i) */
                        if (c != r || j >= i)
                            D[(size_t)(6 * r + i) * n + 6 * c + j] = v;
                    }
            }
        /* diagonal blocks: lower part taken from stored upper part (H symmetric) */
        for (int r = 0; r < npf; r++)
        {
            const double* B = Hsc + 36 * s->rowptr[r];
            for (int j = 0; j < 6; j++)
                for (int i = j; i < 6; i++)
                    D[(size_t)(6 * r + j) * n + 6 * r + i] = B[i * 6 + j];
        }
        ok = dense_chol(n, D);
        if (ok)
        {
            memcpy(dxp, bsc, sizeof(double) * n);
            dense_chol_solve(n, D, dxp);
        }
        free(D);
    }
    else
    {
        ok = ba_bsr_chol_solve(npf, s->rowptr, s->colind, Hsc, bsc, dxp);
    }
    if (ok)
    { /* schurComplementPost .cu:1419-1442 */
#ifdef BA_OMP
#pragma omp parallel for schedule(static)
#endif
        for (int l = 0; l < nlf; l++)
        {
            double cl[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
            for (int a = s->colptr[l]; a < s->colptr[l + 1]; a++)
            {
                const int e = s->coledge[a];
                const double* H = Hpl + 18 * e;
                const double* xp = dxp + 6 * s->pidx[p->e_pose[e]];
                for (int c = 0; c < 3; c++)
                {
                    double v = 0;
                    for (int m = 0; m < 6; m++)
                        v += H[c * 6 + m] * xp[m];
                    cl[c] -= v;
                }
            }
            const double* iHl = iH + 9 * l;
            for (int r = 0; r < 3; r++)
                dxl[3 * l + r] = iHl[0 * 3 + r] * cl[0] + iHl[1 * 3 + r] * cl[1] + iHl[2 * 3 + r] * cl[2];
        }
    }
    free(Hpp), free(Hll), free(iH), free(Hsc), free(bsc);
    return ok;
}

int ba_solve_step(const ba_problem* p, double lambda, int use_dense, double* dxp, double* dxl)
{
    ba_struct s;
    build_struct(p, &s);
    double* Hpp = (double*)malloc(sizeof(double) * 36 * (s.npf + 1));
    double* bp = (double*)malloc(sizeof(double) * 6 * (s.npf + 1));
    double* Hll = (double*)malloc(sizeof(double) * 9 * (s.nlf + 1));
    double* bl = (double*)malloc(sizeof(double) * 3 * (s.nlf + 1));
    double* Hpl = (double*)malloc(sizeof(double) * 18 * (p->n_edges + 1));
    ba_build_system(p, Hpp, bp, Hll, bl, Hpl);
    const int ok = solve_with(p, &s, lambda, use_dense, Hpp, bp, Hll, bl, Hpl, dxp, dxl);
    free(Hpp), free(bp), free(Hll), free(bl), free(Hpl);
    free_struct(&s);
    return ok;
}

/* ------------------------------------------------------------------ LM -------------- */

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

int ba_optimize(ba_problem* p, int niterations, int use_dense, ba_iter_info* info)
{
    const int maxq = 10;
    const double tau = 1e-5;
    double nu = 2.0, lambda = 0.0, F = 0.0;
    int nrec = 0;

    ba_struct s;
    build_struct(p, &s); /* buildStructure, once (cuda_graph_optimisation.cpp:57) */
    const int npf = s.npf, nlf = s.nlf;
    double* Hpp = (double*)malloc(sizeof(double) * 36 * (npf + 1));
    double* bp = (double*)malloc(sizeof(double) * 6 * (npf + 1));
    double* Hll = (double*)malloc(sizeof(double) * 9 * (nlf + 1));
    double* bl = (double*)malloc(sizeof(double) * 3 * (nlf + 1));
    double* Hpl = (double*)malloc(sizeof(double) * 18 * (p->n_edges + 1));
    double* dxp = (double*)calloc(6 * (npf + 1), sizeof(double));
    double* dxl = (double*)calloc(3 * (nlf + 1), sizeof(double));
    double* pose_bak = (double*)malloc(sizeof(double) * 7 * (p->n_poses + 1));
    double* lm_bak = (double*)malloc(sizeof(double) * 3 * (p->n_landmarks + 1));

    for (int iteration = 0; iteration < niterations; iteration++)
    {
        const double iniF = ba_compute_errors(p, 0, 0);
        F = iniF;
        ba_build_system(p, Hpp, bp, Hll, bl, Hpl);
        if (iteration == 0)
        { /* maxDiagonal (.cu:1223-1253, block_solver.cpp:309-320) */
            double mx = 0;
            for (int i = 0; i < npf; i++)
                for (int k = 0; k < 6; k++)
                    if (Hpp[36 * i + 7 * k] > mx)
                        mx = Hpp[36 * i + 7 * k];
            for (int i = 0; i < nlf; i++)
                for (int k = 0; k < 3; k++)
                    if (Hll[9 * i + 4 * k] > mx)
                        mx = Hll[9 * i + 4 * k];
            lambda = tau * mx;
        }
        int q = 0;
        double rho = -1.0;
        for (; q < maxq && rho < 0; q++)
        {
            memcpy(pose_bak, p->pose, sizeof(double) * 7 * p->n_poses); /* push */
            memcpy(lm_bak, p->lm, sizeof(double) * 3 * p->n_landmarks);
            const int success =
                solve_with(p, &s, lambda, use_dense, Hpp, bp, Hll, bl, Hpl, dxp, dxl);
            /* update (.cu:1444-1469). A failed factorisation leaves no valid step: the
             * reference would re-apply stale device memory; here the step is skipped and
             * the trial is rejected (documented deviation). */
            if (success)
            {
                for (int i = 0; i < p->n_poses; i++)
                    if (!p->pose_fixed[i])
                        ba_pose_update(p->pose + 7 * i, dxp + 6 * s.pidx[i]);
                for (int i = 0; i < p->n_landmarks; i++)
                    if (!p->lm_fixed[i])
                        for (int k = 0; k < 3; k++)
                            p->lm[3 * i + k] += dxl[3 * s.lidx[i] + k];
            }
            const double Fhat = ba_compute_errors(p, 0, 0);
            double scale = 0; /* computeScale (.cu:1471-1490): sum x(lambda x + b) */
            if (success)
            {
                for (int i = 0; i < 6 * npf; i++)
                    scale += dxp[i] * (lambda * dxp[i] + bp[i]);
                for (int i = 0; i < 3 * nlf; i++)
                    scale += dxl[i] * (lambda * dxl[i] + bl[i]);
            }
            scale += 1e-3;
            const double Fdiff = Fhat - F;
            rho = success ? (F - Fhat) / scale : -1.0;
            if (rho > 0)
            {
                const double a = 1 - pow(2 * rho - 1, 3);
                lambda *= clampd(a, 1.0 / 3.0, 2.0 / 3.0);
                nu = 2.0;
                F = Fhat;
                break;
            }
            else
            {
                lambda *= nu;
                nu *= 2.0;
                memcpy(p->pose, pose_bak, sizeof(double) * 7 * p->n_poses); /* pop */
                memcpy(p->lm, lm_bak, sizeof(double) * 3 * p->n_landmarks);
                if (!isfinite(lambda) || (success && Fdiff < 1e-4))
                    break;
            }
        }
        if (info)
        {
            info[nrec].iteration = iteration;
            info[nrec].chi2 = F;
            info[nrec].lambda = lambda;
            info[nrec].rho = rho;
            info[nrec].trials = q;
        }
        nrec++;
        if (q == maxq || rho < 1e-6 || !isfinite(lambda))
            break;
    }
    free(Hpp), free(bp), free(Hll), free(bl), free(Hpl), free(dxp), free(dxl);
    free(pose_bak), free(lm_bak);
    free_struct(&s);
    return nrec;
}
