// Does kernel-argument preloading (-mllvm -amdgpu-kernarg-preload-count=16) shorten a chain of short dependent kernels
// on this stack?  The same source is compiled twice (with / without the flag); each binary times N back-to-back launches
// of a one-workgroup kernel whose first action is a load through a pointer argument (what every kernel of the
// factorisation does: record -> operands -> result).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench3.hip -o build_kp/mb3_off
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/microbench3.hip -o build_kp/mb3_on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_chain(const int* __restrict__ idx, const double* __restrict__ a, double* __restrict__ out, int step)
{
    const int i = idx[threadIdx.x & 63];            // record
    const double v = a[i + (step & 1)];             // operand through the record
    if (threadIdx.x == 0)
        out[step & 7] = v + out[(step + 7) & 7];    // result (depends on the previous launch)
}

// holds the stream busy while the host queues the chain behind it: the chain then runs at the GPU's pace, not the host's
__global__ void k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(32);
}

struct Big { const int* idx; const double* a; double* out; long pad[24]; int step; };
__global__ void k_chain_struct(Big p)
{
    const int i = p.idx[threadIdx.x & 63];
    const double v = p.a[i + (p.step & 1)];
    if (threadIdx.x == 0)
        p.out[p.step & 7] = v + p.out[(p.step + 7) & 7];
}

int main()
{
    int* idx; double *a, *out;
    CK(hipMalloc(&idx, 64 * sizeof(int))); CK(hipMalloc(&a, 4096 * sizeof(double))); CK(hipMalloc(&out, 8 * sizeof(double)));
    std::vector<int> h(64); for (int i = 0; i < 64; i++) h[i] = (i * 37) % 4000;
    CK(hipMemcpy(idx, h.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(a, 0, 4096 * sizeof(double))); CK(hipMemset(out, 0, 8 * sizeof(double)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int form = 0; form < 2; form++)
        for (int rep = 0; rep < 4; rep++)
        {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1500000LL); // 15 ms at 100 MHz
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < N; k++)
                if (form == 0)
                    hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, idx, a, out, k);
                else
                {
                    Big b{idx, a, out, {0}, k};
                    hipLaunchKernelGGL(k_chain_struct, dim3(1), dim3(256), 0, s, b);
                }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s args: %.3f us per launch\n", form == 0 ? "flat  " : "struct", 1e3 * ms / N);
        }
    return 0;
}
