// Device-side geometry for the BA kernels (gfx950).  fp64 throughout.
// Formulas restate SURVEY.md Appendix A; citations "ref:" are into the reference repo.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cugo_dev
{

struct Robust
{
    int type;
    double delta;
};

// ref: src/cuda/cuda_block_solver.cu:972-1027
__device__ __forceinline__ double rk_rho(const Robust rk, double x)
{
    const double d2 = rk.delta * rk.delta;
    if (rk.type == 2)
    {
        const double u = 1.0 - x / d2;
        const double mx = (1.0 / 3) * d2;
        return x <= d2 ? mx * (1.0 - u * u * u) : mx;
    }
    if (rk.type == 1)
        return d2 * log((1.0 / d2) * x + 1.0);
    if (rk.type == 3) // Huber (not in the reference's enum; g2o RobustKernelHuber, what ORB-SLAM2 uses)
        return x <= d2 ? x : 2.0 * rk.delta * sqrt(x) - d2;
    return x;
}
__device__ __forceinline__ double rk_drho(const Robust rk, double x)
{
    const double d2 = rk.delta * rk.delta;
    if (rk.type == 2)
    {
        const double u = 1.0 - x / d2;
        return x <= d2 ? u * u : 0.0;
    }
    if (rk.type == 1)
        return 1.0 / ((1.0 / d2) * x + 1.0);
    if (rk.type == 3)
        return x <= d2 ? 1.0 : rk.delta / sqrt(x);
    return 1.0;
}

struct EdgeGeom
{
    double Xc[3];
    double e[3]; // e[2] = 0 for mono
    double w;    // omega * rho'
    double chi;  // rho(omega |e|^2)
};

// Xc = R(q) Xw + t in the cross-product form (ref: .cu:379-402)
__device__ __forceinline__ void world_to_cam(const double* __restrict__ pose,
                                             const double* __restrict__ Xw, double* Xc)
{
    const double qx = pose[0], qy = pose[1], qz = pose[2], qw = pose[3];
    double t1x = qy * Xw[2] - qz * Xw[1];
    double t1y = qz * Xw[0] - qx * Xw[2];
    double t1z = qx * Xw[1] - qy * Xw[0];
    t1x += t1x;
    t1y += t1y;
    t1z += t1z;
    const double t2x = qy * t1z - qz * t1y;
    const double t2y = qz * t1x - qx * t1z;
    const double t2z = qx * t1y - qy * t1x;
    Xc[0] = Xw[0] + qw * t1x + t2x + pose[4];
    Xc[1] = Xw[1] + qw * t1y + t2y + pose[5];
    Xc[2] = Xw[2] + qw * t1z + t2z + pose[6];
}

// residual (proj - meas), weight and chi for one edge (ref: .cu:410-424, 1100-1109, 1190-1192)
__device__ __forceinline__ void edge_residual(const double* __restrict__ pose,
                                              const double* __restrict__ Xw, double mu, double mv,
                                              double mr, bool stereo, double omega,
                                              const double* __restrict__ cam, const Robust rk,
                                              EdgeGeom& g)
{
    world_to_cam(pose, Xw, g.Xc);
    const double invZ = 1.0 / g.Xc[2];
    const double pu = cam[0] * invZ * g.Xc[0] + cam[2];
    const double pv = cam[1] * invZ * g.Xc[1] + cam[3];
    g.e[0] = pu - mu;
    g.e[1] = pv - mv;
    double sq = g.e[0] * g.e[0] + g.e[1] * g.e[1];
    if (stereo)
    {
        g.e[2] = (pu - cam[4] * invZ) - mr;
        sq += g.e[2] * g.e[2];
    }
    else
        g.e[2] = 0.0;
    const double x = omega * sq;
    g.chi = rk_rho(rk, x);
    g.w = omega * rk_drho(rk, x);
}

// rotation matrix rows from quaternion (ref: .cu:449-478). R[r][c]
__device__ __forceinline__ void quat_to_rot(const double* __restrict__ q, double R[3][3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0][0] = 1 - (tyy + tzz);
    R[0][1] = txy - twz;
    R[0][2] = txz + twy;
    R[1][0] = txy + twz;
    R[1][1] = 1 - (txx + tzz);
    R[1][2] = tyz - twx;
    R[2][0] = txz - twy;
    R[2][1] = tyz + twx;
    R[2][2] = 1 - (txx + tyy);
}

// Pose Jacobian rows JP[m][6] (m < dim); third row zero for mono. ref: .cu:491-578
__device__ __forceinline__ void jac_pose(const double* Xc, const double* __restrict__ cam,
                                         bool stereo, double JP[3][6])
{
    const double X = Xc[0], Y = Xc[1], Z = Xc[2];
    const double invZ = 1.0 / Z;
    const double fu = cam[0], fv = cam[1];
    if (!stereo)
    {
        const double x = invZ * X, y = invZ * Y;
        const double fu_iz = fu * invZ, fv_iz = fv * invZ;
        JP[0][0] = fu * x * y;
        JP[0][1] = -fu * (1 + x * x);
        JP[0][2] = fu * y;
        JP[0][3] = -fu_iz;
        JP[0][4] = 0;
        JP[0][5] = fu_iz * x;
        JP[1][0] = fv * (1 + y * y);
        JP[1][1] = -fv * x * y;
        JP[1][2] = -fv * x;
        JP[1][3] = 0;
        JP[1][4] = -fv_iz;
        JP[1][5] = fv_iz * y;
#pragma unroll
        for (int c = 0; c < 6; c++)
            JP[2][c] = 0;
    }
    else
    {
        const double iZZ = invZ * invZ, bf = cam[4];
        JP[0][0] = X * Y * iZZ * fu;
        JP[0][1] = -(1 + (X * X * iZZ)) * fu;
        JP[0][2] = Y * invZ * fu;
        JP[0][3] = -1 * invZ * fu;
        JP[0][4] = 0;
        JP[0][5] = X * iZZ * fu;
        JP[1][0] = (1 + Y * Y * iZZ) * fv;
        JP[1][1] = -X * Y * iZZ * fv;
        JP[1][2] = -X * invZ * fv;
        JP[1][3] = 0;
        JP[1][4] = -1 * invZ * fv;
        JP[1][5] = Y * iZZ * fv;
        JP[2][0] = JP[0][0] - bf * Y * iZZ;
        JP[2][1] = JP[0][1] + bf * X * iZZ;
        JP[2][2] = JP[0][2];
        JP[2][3] = JP[0][3];
        JP[2][4] = 0;
        JP[2][5] = JP[0][5] - bf * iZZ;
    }
}

// Landmark Jacobian rows JL[m][3] from the rotation matrix of the pose. ref: .cu:508-513, 546-556
__device__ __forceinline__ void jac_landmark_R(const double* Xc, const double R[3][3],
                                               const double* __restrict__ cam, bool stereo,
                                               double JL[3][3])
{
    const double X = Xc[0], Y = Xc[1], Z = Xc[2];
    const double invZ = 1.0 / Z;
    const double fu = cam[0], fv = cam[1];
    if (!stereo)
    {
        const double x = invZ * X, y = invZ * Y;
        const double fu_iz = fu * invZ, fv_iz = fv * invZ;
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            JL[0][j] = -fu_iz * (R[0][j] - x * R[2][j]);
            JL[1][j] = -fv_iz * (R[1][j] - y * R[2][j]);
            JL[2][j] = 0;
        }
    }
    else
    {
        const double iZZ = invZ * invZ, bf = cam[4];
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            JL[0][j] = -fu * R[0][j] * invZ + fu * X * R[2][j] * iZZ;
            JL[1][j] = -fv * R[1][j] * invZ + fv * Y * R[2][j] * iZZ;
            JL[2][j] = JL[0][j] - bf * R[2][j] * iZZ;
        }
    }
}
__device__ __forceinline__ void jac_landmark(const double* Xc, const double* __restrict__ q,
                                             const double* __restrict__ cam, bool stereo,
                                             double JL[3][3])
{
    double R[3][3];
    quat_to_rot(q, R);
    jac_landmark_R(Xc, R, cam, stereo, JL);
}

// symmetric 3x3 inverse by adjugate, A column-major 9 with lambda added to the diagonal.
// ref: .cu:639-669 (no pivot / SPD check there either). Returns the 6 unique entries.
struct Sym3
{
    double b00, b01, b02, b11, b12, b22;
};
__device__ __forceinline__ Sym3 sym3_inv(const double* __restrict__ A, double lambda)
{
    const double A00 = A[0] + lambda, A01 = A[3], A11 = A[4] + lambda;
    const double A02 = A[2], A12 = A[7], A22 = A[8] + lambda;
    const double det = A00 * A11 * A22 + A01 * A12 * A02 + A02 * A01 * A12 - A00 * A12 * A12 -
                       A02 * A11 * A02 - A01 * A01 * A22;
    const double id = 1 / det;
    Sym3 s;
    s.b00 = id * (A11 * A22 - A12 * A12);
    s.b01 = id * (A02 * A12 - A01 * A22);
    s.b11 = id * (A00 * A22 - A02 * A02);
    s.b02 = id * (A01 * A12 - A02 * A11);
    s.b12 = id * (A02 * A01 - A00 * A12);
    s.b22 = id * (A00 * A11 - A01 * A01);
    return s;
}

// T <- exp([w,v]) * T  (ref: updateExp .cu:781-809, updatePose .cu:811-823)
__device__ __forceinline__ void pose_exp_update(const double* __restrict__ dx,
                                                const double* __restrict__ pin,
                                                double* __restrict__ pout)
{
    const double wx = dx[0], wy = dx[1], wz = dx[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
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
    // O1 = [w]x, O2 = [w]x^2 ; R = I + a1 O1 + a2 O2 ; V = I + b1 O1 + b2 O2 (row-major here)
    const double xx = wx * wx, yy = wy * wy, zz = wz * wz;
    const double xy = wx * wy, yz = wy * wz, zx = wz * wx;
    const double O1[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    const double O2[3][3] = {{-yy - zz, xy, zx}, {xy, -zz - xx, yz}, {zx, yz, -xx - yy}};
    double R[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            const double I = (i == j) ? 1.0 : 0.0;
            R[i][j] = I + a1 * O1[i][j] + a2 * O2[i][j];
            V[i][j] = I + b1 * O1[i][j] + b2 * O2[i][j];
        }
    // quaternion of R (ref: .cu:721-754)
    double dq[4];
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0)
    {
        t = sqrt(t + 1);
        dq[3] = 0.5 * t;
        t = 0.5 / t;
        dq[0] = (R[2][1] - R[1][2]) * t;
        dq[1] = (R[0][2] - R[2][0]) * t;
        dq[2] = (R[1][0] - R[0][1]) * t;
    }
    else
    {
        int i = 0;
        if (R[1][1] > R[0][0])
            i = 1;
        if (R[2][2] > R[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i][i] - R[j][j] - R[k][k] + 1);
        dq[i] = 0.5 * t;
        t = 0.5 / t;
        dq[3] = (R[k][j] - R[j][k]) * t;
        dq[j] = (R[j][i] + R[i][j]) * t;
        dq[k] = (R[k][i] + R[i][k]) * t;
    }
    const double dt0 = V[0][0] * dx[3] + V[0][1] * dx[4] + V[0][2] * dx[5];
    const double dt1 = V[1][0] * dx[3] + V[1][1] * dx[4] + V[1][2] * dx[5];
    const double dt2 = V[2][0] * dx[3] + V[2][1] * dx[4] + V[2][2] * dx[5];
    // t' = dt + R(dq) t   (rotate in cross-product form)
    const double q0 = pin[0], q1 = pin[1], q2 = pin[2], q3 = pin[3];
    const double tx = pin[4], ty = pin[5], tz = pin[6];
    double c1x = dq[1] * tz - dq[2] * ty;
    double c1y = dq[2] * tx - dq[0] * tz;
    double c1z = dq[0] * ty - dq[1] * tx;
    c1x += c1x;
    c1y += c1y;
    c1z += c1z;
    const double c2x = dq[1] * c1z - dq[2] * c1y;
    const double c2y = dq[2] * c1x - dq[0] * c1z;
    const double c2z = dq[0] * c1y - dq[1] * c1x;
    pout[4] = dt0 + (tx + dq[3] * c1x + c2x);
    pout[5] = dt1 + (ty + dq[3] * c1y + c2y);
    pout[6] = dt2 + (tz + dq[3] * c1z + c2z);
    // q' = normalise(dq * q), w >= 0  (ref: .cu:756-775)
    double r3 = dq[3] * q3 - dq[0] * q0 - dq[1] * q1 - dq[2] * q2;
    double r0 = dq[3] * q0 + dq[0] * q3 + dq[1] * q2 - dq[2] * q1;
    double r1 = dq[3] * q1 + dq[1] * q3 + dq[2] * q0 - dq[0] * q2;
    double r2 = dq[3] * q2 + dq[2] * q3 + dq[0] * q1 - dq[1] * q0;
    double invn = 1 / sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
    if (r3 < 0)
        invn = -invn;
    pout[0] = invn * r0;
    pout[1] = invn * r1;
    pout[2] = invn * r2;
    pout[3] = invn * r3;
}

} // namespace cugo_dev
