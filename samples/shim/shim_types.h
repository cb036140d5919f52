// Stand-ins for the reference's device containers, just enough surface for the kernel-seam shim
// (samples/shim/shim_cugo_hip.cpp) to compile on its own.  In the reference these are
// DeviceBuffer<T> / GpuVec<T> (ref: src/device_buffer.h:33-276), DeviceBlockVector /
// DeviceBlockMatrix (ref: src/device_matrix.h:43-210) and CudaDeviceInfo (ref: src/cuda_device.h:18-22);
// a maintainer compiles the shim against the real headers instead of this file.
#pragma once
#include <cstddef>
#include <cstdint>

namespace cugo
{

using Scalar = double; // ref: src/scalar.h:24-28

enum class RobustKernelType { None = 0, Cauchy = 1, Tukey = 2 }; // ref: src/robust_kernel.h:12-17

template <typename T>
struct GpuVec // non-owning view here; the reference's owns device memory
{
    T* ptr = nullptr;
    size_t n = 0;
    T* data() const { return ptr; }
    size_t size() const { return n; }
    int ssize() const { return (int)n; }
};

template <int ROWS, int COLS>
struct GpuBlockVec // DeviceBlockVector<T, ROWS, COLS>: `size` blocks of ROWS x COLS, column-major
{
    Scalar* ptr = nullptr;
    int n = 0;
    Scalar* values() const { return ptr; }
    int size() const { return n; }
};

struct GpuBlockMat // DeviceBlockMatrix: block CSR / CSC with values + outer / inner index arrays
{
    Scalar* ptr = nullptr;
    int* outer = nullptr; // device
    int* inner = nullptr; // device
    int brows_ = 0, bcols_ = 0, nnz_ = 0;
    Scalar* values() const { return ptr; }
    int* outerIndices() const { return outer; }
    int* innerIndices() const { return inner; }
    int rows() const { return brows_; }
    int cols() const { return bcols_; }
    int nnz() const { return nnz_; }
};

struct Vec2i { int v[2]; };
struct Vec3i { int v[3]; };
struct Vec3dPod { double v[3]; };
struct Vec5dPod { double v[5]; };
struct Se3Pod { double q[4], t[3]; }; // 56 bytes, ref: src/fixed_vector.h:393-474
template <int M>
struct VecxdPod { double v[M]; };

using GpuVec1d = GpuVec<Scalar>;
using GpuVec1i = GpuVec<int>;
using GpuVec1b = GpuVec<uint8_t>;
using GpuVec2i = GpuVec<Vec2i>;
using GpuVec3i = GpuVec<Vec3i>;
using GpuVec3d = GpuVec<Vec3dPod>;
using GpuVec5d = GpuVec<Vec5dPod>;
using GpuVecSe3d = GpuVec<Se3Pod>;
template <int M>
using GpuVecxd = GpuVec<VecxdPod<M>>;
using GpuPxPBlockVec = GpuBlockVec<6, 6>;
using GpuLxLBlockVec = GpuBlockVec<3, 3>;
using GpuPx1BlockVec = GpuBlockVec<6, 1>;
using GpuLx1BlockVec = GpuBlockVec<3, 1>;
using GpuPxLBlockVec = GpuBlockVec<6, 3>;
using GpuHplBlockMat = GpuBlockMat;
using GpuHscBlockMat = GpuBlockMat;

struct CudaDeviceInfo // ref: src/cuda_device.h:18-22 holds {cudaStream_t, cudaEvent_t}
{
    void* stream = nullptr;
    void* event = nullptr;
};

struct RobustKernel // ref: src/robust_kernel.h:19-35
{
    RobustKernelType type = RobustKernelType::None;
    Scalar delta = 1.0;
};

// host-side pattern object the linear solver is initialised with (ref: src/sparse_block_matrix.h:58-104)
struct HschurSparseBlockMatrix
{
    int brows_ = 0;
    const int* outer = nullptr; // host, [brows+1]
    const int* inner = nullptr; // host, [nblocks]
    int brows() const { return brows_; }
    const int* outerIndices() const { return outer; }
    const int* innerIndices() const { return inner; }
};

} // namespace cugo
