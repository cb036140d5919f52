// POD math types and option structs of the cugo graph API (MI355X build).
// Same names, member layout and meaning as the reference so user code compiles unchanged:
//   Vec / Quat / Se3        ref: src/fixed_vector.h:29-86, 317-474  (Se3 = q(x,y,z,w) then t)
//   Camera                  ref: src/camera.h:10-41
//   GraphOptimisationOptions ref: src/graph_optimisation_options.h:8-19
//   RobustKernelType        ref: src/robust_kernel.h:12-17
// No HIP/CUDA header is pulled in here: the runtime stays behind libcugo_hip.so.
#pragma once
#include <cstddef>

#if defined(_WIN32)
#define CUGO_API
#else
#define CUGO_API __attribute__((visibility("default")))
#endif

namespace cugo
{

using Scalar = double;

static constexpr int PDIM = 6; // pose increment: [omega(3), upsilon(3)]
static constexpr int LDIM = 3;

template <typename T, int N>
struct Vec
{
    static constexpr std::size_t Size = N;
    T data[N];

    Vec() = default;
    template <typename U>
    Vec(const U* v)
    {
        for (int i = 0; i < N; i++)
            data[i] = T(v[i]);
    }
    template <typename... A, typename = typename std::enable_if<sizeof...(A) == N && (N > 1)>::type>
    Vec(A... a) : data{T(a)...}
    {
    }
    T& operator[](int i) noexcept { return data[i]; }
    const T& operator[](int i) const noexcept { return data[i]; }
    template <typename U>
    void copyTo(U* out) const noexcept
    {
        for (int i = 0; i < N; i++)
            out[i] = U(data[i]);
    }
};

template <typename T>
using Vec2 = Vec<T, 2>;
template <typename T>
using Vec3 = Vec<T, 3>;
template <typename T>
using Vec4 = Vec<T, 4>;
using Vec2i = Vec2<int>;
using Vec3i = Vec3<int>;
using Vec2d = Vec2<double>;
using Vec3d = Vec3<double>;
using Vec4d = Vec4<double>;
using Vec5d = Vec<double, 5>;
using Vec6d = Vec<double, 6>;

template <typename T>
struct Quat
{
    static constexpr std::size_t Size = 4;
    union
    {
        T data[4];
        struct
        {
            T x, y, z, w;
        };
    };
    Quat() : data{0, 0, 0, 1} {}
    template <typename U>
    Quat(const U* v) : data{T(v[0]), T(v[1]), T(v[2]), T(v[3])}
    {
    }
    template <typename U>
    Quat(const U& x_, const U& y_, const U& z_, const U& w_) : data{T(x_), T(y_), T(z_), T(w_)}
    {
    }
    Quat(const Vec4<T>& v) : data{v[0], v[1], v[2], v[3]} {}
    T& operator[](int i) noexcept { return data[i]; }
    const T& operator[](int i) const noexcept { return data[i]; }
    template <typename U>
    void copyTo(U* out) const noexcept
    {
        for (int i = 0; i < 4; i++)
            out[i] = U(data[i]);
    }
};
using QuatD = Quat<double>;

// world -> camera rigid transform: Xc = R(r) Xw + t
template <typename T>
struct Se3
{
    Quat<T> r;
    Vec3<T> t;
    Se3() : t{T(0), T(0), T(0)} {}
    template <typename U>
    Se3(const U* rv, const U* tv) : r(rv), t(tv)
    {
    }
    Se3(const Quat<T>& q, const Vec3<T>& v) : r(q), t(v) {}
    template <typename U>
    void copyTo(U* r_out, U* t_out) const noexcept
    {
        r.copyTo(r_out);
        t.copyTo(t_out);
    }
};
using Se3D = Se3<double>;
static_assert(sizeof(Se3D) == 7 * sizeof(double), "Se3D must be 7 packed doubles");

struct CUGO_API Camera
{
    double fx = 0, fy = 0, cx = 0, cy = 0;
    double bf = 0; // stereo baseline * fx
    Camera() = default;
    Camera(double fx_, double fy_, double cx_, double cy_, double bf_ = 1.0)
        : fx(fx_), fy(fy_), cx(cx_), cy(cy_), bf(bf_)
    {
    }
    Camera(float fx_, float fy_, float cx_, float cy_, float bf_)
        : fx(fx_), fy(fy_), cx(cx_), cy(cy_), bf(bf_)
    {
    }
};

struct CUGO_API GraphOptimisationOptions
{
    bool perEdgeInformation = false; // per-edge weight vs one weight per edge set
    bool perEdgeCamera = false;      // per-edge camera vs one camera per edge set
    // The reference's compile-time USE_FLOAT32 ("32bit float in internal floating-point
    // operations", ref: CMakeLists.txt:8, src/scalar.h:24-28) as a run-time option: the two
    // per-edge block streams Hpl and Hpl*Hll^-1 are stored as float (they are what the Schur
    // kernels move: half the HBM traffic); estimates, Jacobians, every accumulator, Hsc and the
    // Cholesky stay fp64.  chi2 per iteration then agrees with the fp64 run to ~1e-6 relative
    // (tolerance stated and tested in tests/test_gpu.py::test_float32_block_storage).
    bool useFloat32 = false;
    // Extension: a plan-only optimiser needs no GPU.  initialize() then runs the host side only
    // (flattening, index assignment, landmark-major layout, Hsc pattern, product lists, ordering and
    // symbolic factorisation — what structureStats() reports); optimize() fails.  Used to size a
    // problem ahead of time and to run the host code under sanitizers (make SAN=1).
    bool planOnly = false;
};

enum class RobustKernelType
{
    None,
    Cauchy,
    Tukey,
    Huber // extension (the reference stops at Tukey): g2o RobustKernelHuber, ORB-SLAM2's kernel
};

// Per-edge-set robust kernel parameters, passed to the kernels by value.
// (ref: RobustKernel src/robust_kernel.h:22-35 wraps a process-global device object; here
//  each set keeps its own {type, delta}.)
class CUGO_API RobustKernel
{
public:
    void create(RobustKernelType t, Scalar d) noexcept
    {
        type_ = t;
        delta_ = d;
    }
    RobustKernelType type() const noexcept { return type_; }
    Scalar delta() const noexcept { return delta_; }

private:
    RobustKernelType type_ = RobustKernelType::None;
    Scalar delta_ = 1.0;
};

} // namespace cugo
