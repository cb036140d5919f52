// Seeded ORB-SLAM-style synthetic BA graphs — harness code: the KITTI
// inputs of the reference (samples/ba_input.7z) are not available, so bench.py and the
// full-size tests generate graphs with the same pose / landmark / edge counts.
//   * car-like trajectory, ~1 m between keyframes, gentle yaw; with loop closures the path
//     is a two-lap ring so revisited places are seen again (KITTI 00 has such revisits);
//   * each landmark is observed by a run of k consecutive keyframes (k >= 2, geometric),
//     sum k == n_edges exactly; loop-closure landmarks get a second run one lap later;
//   * KITTI-like intrinsics, pixel noise, ORB pyramid-level information 1/1.2^(2 level);
//   * initial estimates = ground truth + noise; pose 0 is the gauge.
// Own uniform/normal generators on top of std::mt19937_64 so the stream is identical on any
// libstdc++.
#include <cmath>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <vector>

#include "../../../include/cugo_hip.h"

namespace
{

struct Rng
{
    std::mt19937_64 g;
    explicit Rng(uint64_t seed) : g(seed) {}
    double uni() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }
    double normal()
    {
        double u1 = uni();
        if (u1 < 1e-300)
            u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * uni());
    }
    int range(int n) { return (int)(uni() * n) % (n > 0 ? n : 1); }
};

struct Mat3
{
    double m[3][3];
};

Mat3 rot_y(double a)
{
    const double c = std::cos(a), s = std::sin(a);
    return {{{c, 0, s}, {0, 1, 0}, {-s, 0, c}}};
}
Mat3 small_rot(double rx, double rz)
{ // first-order pitch / roll, re-orthonormalised by the quaternion conversion
    return {{{1, -rz, 0}, {rz, 1, -rx}, {0, rx, 1}}};
}
Mat3 mul(const Mat3& a, const Mat3& b)
{
    Mat3 c{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return c;
}
void to_quat(const Mat3& R, double* q)
{
    const double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0)
    {
        const double s = std::sqrt(t + 1.0) * 2;
        q[3] = 0.25 * s;
        q[0] = (R.m[2][1] - R.m[1][2]) / s;
        q[1] = (R.m[0][2] - R.m[2][0]) / s;
        q[2] = (R.m[1][0] - R.m[0][1]) / s;
    }
    else
    {
        int i = 0;
        if (R.m[1][1] > R.m[0][0])
            i = 1;
        if (R.m[2][2] > R.m[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        const double s = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0) * 2;
        q[i] = 0.25 * s;
        q[3] = (R.m[k][j] - R.m[j][k]) / s;
        q[j] = (R.m[j][i] + R.m[i][j]) / s;
        q[k] = (R.m[k][i] + R.m[i][k]) / s;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (q[3] < 0)
        n = -n;
    for (int i = 0; i < 4; i++)
        q[i] /= n;
}
void quat_to_R(const double* q, Mat3& R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
}

} // namespace

extern "C" int cugo_synth_generate(const cugo_synth_params* prm, double* poses, double* lms,
                                   int32_t* e_pose, int32_t* e_lm, uint8_t* e_stereo, double* e_meas,
                                   double* e_omega, double* cam5)
{
    const int P = prm->n_poses, L = prm->n_landmarks, E = prm->n_edges;
    if (P < 3 || L < 1 || E < 2 * L)
        return CUGO_ERR_INVALID;
    Rng rng(prm->seed);
    const double cam[5] = {718.856, 718.856, 607.1928, 185.2157, 386.1448};
    for (int i = 0; i < 5; i++)
        cam5[i] = cam[i];

    // ---- ground-truth trajectory: camera-to-world (Rwc, twc), forward = +z, y down --------
    const bool ring = prm->n_loop_closures > 0;
    const int lap = ring ? (P + 1) / 2 : P;
    std::vector<Mat3> Rwc(P);
    std::vector<double> twc(3 * (size_t)P);
    {
        double yaw = 0, x = 0, z = 0, wob = 0;
        for (int i = 0; i < P; i++)
        {
            if (ring)
            { // two laps of a ring: pose i+lap revisits pose i with ~0.3 m lateral offset
                const double a = 6.283185307179586 * (double)(i % lap) / lap;
                const double rad = lap * 1.0 / 6.283185307179586 + (i >= lap ? 0.3 : 0.0);
                x = rad * (1 - std::cos(a)); // velocity direction (sin a, cos a) = camera forward
                z = rad * std::sin(a);
                yaw = a;
            }
            else
            {
                wob = 0.98 * wob + 0.002 * rng.normal();
                yaw += wob;
                x += std::sin(yaw) * 1.0;
                z += std::cos(yaw) * 1.0;
            }
            Rwc[i] = mul(rot_y(yaw), small_rot(0.004 * rng.normal(), 0.004 * rng.normal()));
            twc[3 * (size_t)i] = x + 0.02 * rng.normal();
            twc[3 * (size_t)i + 1] = 0.02 * rng.normal();
            twc[3 * (size_t)i + 2] = z + 0.02 * rng.normal();
        }
    }
    std::vector<double> gt(7 * (size_t)P);
    for (int i = 0; i < P; i++)
    {
        Mat3 Rcw;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                Rcw.m[a][b] = Rwc[i].m[b][a];
        to_quat(Rcw, &gt[7 * (size_t)i]);
        Mat3 Rn;
        quat_to_R(&gt[7 * (size_t)i], Rn);
        for (int a = 0; a < 3; a++)
            gt[7 * (size_t)i + 4 + a] = -(Rn.m[a][0] * twc[3 * (size_t)i] + Rn.m[a][1] * twc[3 * (size_t)i + 1] +
                                          Rn.m[a][2] * twc[3 * (size_t)i + 2]);
    }

    // ---- observation counts: k_l >= 2, sum == E -------------------------------------------
    // a run of k keyframes must keep the point in front of all of them: on the ring the arc is
    // limited to ~40 degrees
    const int kmax = std::max(2, std::min(ring ? lap / 9 : P - 1, 30));
    const double mean = (double)E / L;
    std::vector<int> k(L);
    long sum = 0;
    for (int l = 0; l < L; l++)
    {
        const double pgeo = 1.0 / std::max(mean - 1.0, 1.0001);
        int extra = (int)std::floor(std::log(1.0 - rng.uni() * 0.999999) / std::log(1.0 - pgeo));
        k[l] = std::min(kmax, 2 + std::max(0, extra));
        sum += k[l];
    }
    // loop-closure landmarks: a second run of k2 observations one lap later
    const int nlc = ring ? std::min(prm->n_loop_closures, L) : 0;
    std::vector<int> k2(L, 0);
    for (int i = 0; i < nlc; i++)
    {
        const int l = (int)((int64_t)i * L / nlc);
        k2[l] = 2 + rng.range(3);
        sum += k2[l];
    }
    for (int guard = 0; sum != E && guard < 100 * L + 1000; guard++)
    {
        const int l = rng.range(L);
        if (sum < E && k[l] < kmax)
            k[l]++, sum++;
        else if (sum > E && k[l] > 2)
            k[l]--, sum--;
    }
    if (sum != E)
        return CUGO_ERR_INVALID;

    // ---- landmarks (sorted by first observing pose) and their observations ---------------
    std::vector<int> first(L);
    for (int l = 0; l < L; l++)
    {
        const int span = ring && k2[l] ? std::max(k[l] + 1, lap - 6) : P;
        first[l] = (int)((int64_t)l * std::max(1, span - k[l] + 1) / L); // evenly spread, ascending
        if (first[l] + k[l] > span)
            first[l] = span - k[l];
        if (first[l] < 0)
            first[l] = 0;
    }
    size_t ne = 0;
    for (int l = 0; l < L; l++)
    {
        std::vector<int> obs;
        for (int j = 0; j < k[l]; j++)
            obs.push_back(std::min(P - 1, first[l] + j));
        for (int j = 0; j < k2[l]; j++)
            obs.push_back(std::min(P - 1, first[l] + lap + j));
        // de-duplicate (only possible at the very end of the path)
        for (size_t a = 1; a < obs.size(); a++)
            while (a < obs.size() && obs[a] <= obs[a - 1])
                obs[a] = obs[a - 1] + 1;
        for (int& o : obs)
            if (o >= P)
                return CUGO_ERR_INVALID;
        const int mid = obs[(size_t)k[l] / 2];
        double Xw[3];
        for (int attempt = 0;; attempt++)
        {
            const double depth = 0.6 * k[l] + 5.0 + rng.uni() * 40.0;
            const double u = 150 + rng.uni() * 900, v = 40 + rng.uni() * 290;
            const double Xc[3] = {(u - cam[2]) / cam[0] * depth, (v - cam[3]) / cam[1] * depth, depth};
            for (int a = 0; a < 3; a++)
                Xw[a] = Rwc[mid].m[a][0] * Xc[0] + Rwc[mid].m[a][1] * Xc[1] + Rwc[mid].m[a][2] * Xc[2] +
                        twc[3 * (size_t)mid + a];
            bool ok = true;
            for (int o : obs)
            {
                Mat3 R;
                quat_to_R(&gt[7 * (size_t)o], R);
                const double zc = R.m[2][0] * Xw[0] + R.m[2][1] * Xw[1] + R.m[2][2] * Xw[2] + gt[7 * (size_t)o + 6];
                if (zc < 2.0)
                    ok = false;
            }
            if (ok)
                break;
            if (attempt > 200)
                return CUGO_ERR_INVALID;
        }
        for (int a = 0; a < 3; a++)
            lms[3 * (size_t)l + a] = Xw[a];
        for (int o : obs)
        {
            Mat3 R;
            quat_to_R(&gt[7 * (size_t)o], R);
            double xc[3];
            for (int a = 0; a < 3; a++)
                xc[a] = R.m[a][0] * Xw[0] + R.m[a][1] * Xw[1] + R.m[a][2] * Xw[2] + gt[7 * (size_t)o + 4 + a];
            const bool st = rng.uni() < prm->stereo_fraction;
            const double uu = cam[0] * xc[0] / xc[2] + cam[2] + prm->pixel_noise * rng.normal();
            const double vv = cam[1] * xc[1] / xc[2] + cam[3] + prm->pixel_noise * rng.normal();
            const double ur = uu - cam[4] / xc[2] + prm->pixel_noise * rng.normal();
            const int lvl = rng.range(8);
            e_pose[ne] = o;
            e_lm[ne] = l;
            e_stereo[ne] = st ? 1 : 0;
            e_meas[3 * ne] = uu, e_meas[3 * ne + 1] = vv, e_meas[3 * ne + 2] = st ? ur : 0.0;
            e_omega[ne] = 1.0 / std::pow(1.2, 2.0 * lvl);
            ne++;
        }
    }
    if ((long)ne != E)
        return CUGO_ERR_INVALID;

    // ---- initial estimates: ground truth + noise (pose 0 exact: gauge) --------------------
    for (int i = 0; i < P; i++)
    {
        double* o = poses + 7 * (size_t)i;
        const double* g = &gt[7 * (size_t)i];
        if (i == 0)
        {
            for (int a = 0; a < 7; a++)
                o[a] = g[a];
            continue;
        }
        const double rx = prm->pose_rot_noise * rng.normal(), ry = prm->pose_rot_noise * rng.normal(),
                     rz = prm->pose_rot_noise * rng.normal();
        const double th = std::sqrt(rx * rx + ry * ry + rz * rz);
        double dq[4] = {0.5 * rx, 0.5 * ry, 0.5 * rz, 1.0};
        if (th > 1e-12)
        {
            const double s = std::sin(th / 2) / th;
            dq[0] = rx * s, dq[1] = ry * s, dq[2] = rz * s, dq[3] = std::cos(th / 2);
        }
        double r[4];
        r[3] = dq[3] * g[3] - dq[0] * g[0] - dq[1] * g[1] - dq[2] * g[2];
        r[0] = dq[3] * g[0] + dq[0] * g[3] + dq[1] * g[2] - dq[2] * g[1];
        r[1] = dq[3] * g[1] + dq[1] * g[3] + dq[2] * g[0] - dq[0] * g[2];
        r[2] = dq[3] * g[2] + dq[2] * g[3] + dq[0] * g[1] - dq[1] * g[0];
        double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
        if (r[3] < 0)
            n = -n;
        for (int a = 0; a < 4; a++)
            o[a] = r[a] / n;
        // T' = [dR | dt] * T: the translation is rotated too, i.e. the camera turns about its own
        // centre (rotating only R would swing the camera about the world origin — metres of
        // error on a kilometre-long trajectory)
        Mat3 dR;
        quat_to_R(dq, dR);
        for (int a = 0; a < 3; a++)
            o[4 + a] = dR.m[a][0] * g[4] + dR.m[a][1] * g[5] + dR.m[a][2] * g[6] +
                       prm->pose_trans_noise * rng.normal();
    }
    for (int l = 0; l < L; l++)
    {
        // noise proportional to the distance from the first observing camera
        const int o = first[l];
        double d = 0;
        for (int a = 0; a < 3; a++)
        {
            const double dd = lms[3 * (size_t)l + a] - twc[3 * (size_t)o + a];
            d += dd * dd;
        }
        d = std::sqrt(d);
        for (int a = 0; a < 3; a++)
            lms[3 * (size_t)l + a] += prm->landmark_noise_rel * d * rng.normal();
    }
    return CUGO_OK;
}
