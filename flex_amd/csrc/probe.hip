// probe.hip -- what this box's HBM actually delivers ("verify BW_peak with a
// device-to-device copy on the box and report both").  Two streaming kernels over buffers far
// larger than L2 + Infinity Cache: a read-only pass (the shape of this engine's traffic: gathers
// dominate, C is written once) and a read+write copy.  16 B per lane, non-temporal (or, with
// `temporal`, ordinary loads, which may be served by the 256 MiB Infinity Cache), 8 loads in
// flight per lane, a grid of 8 workgroups per CU walking the buffer with a grid stride.
#include <hip/hip_runtime.h>

#include "internal.h"

namespace flex {
namespace {

typedef float v4f_t __attribute__((ext_vector_type(4)));
constexpr int kProbeUnroll = 8;

template <bool NT>
__device__ __forceinline__ v4f_t probe_load(const v4f_t *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_read_kernel(const v4f_t *__restrict__ src, size_t n_vec, float *sink) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    v4f_t acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (kProbeUnroll - 1) * stride < n_vec; i += kProbeUnroll * stride) {
        v4f_t v[kProbeUnroll];
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) v[u] = probe_load<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) acc += v[u];
    }
    for (; i < n_vec; i += stride) acc += probe_load<NT>(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;  // keeps the loads alive; never true for the zero-filled buffer
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(const v4f_t *__restrict__ src, v4f_t *__restrict__ dst, size_t n_vec) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + (kProbeUnroll - 1) * stride < n_vec; i += kProbeUnroll * stride) {
        v4f_t v[kProbeUnroll];
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) v[u] = probe_load<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kProbeUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n_vec; i += stride) __builtin_nontemporal_store(probe_load<NT>(src + i), dst + i);
}

}  // namespace
}  // namespace flex

extern "C" int flex_hbm_probe(int device, int64_t bytes, int reps, int temporal, double *read_gbps, double *copy_gbps) {
    using namespace flex;
    if (device < 0 || bytes < (int64_t(1) << 20) || reps <= 0 || !read_gbps || !copy_gbps) return FLEX_ERR_INVALID;
    int prev = -1;
    FLEX_HIP_TRY(hipGetDevice(&prev));
    FLEX_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    FLEX_HIP_TRY(hipGetDeviceProperties(&prop, device));
    const size_t n_vec = static_cast<size_t>(bytes) / 16;
    void *src = nullptr, *dst = nullptr;
    float *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = FLEX_OK;
    auto fail = [&](hipError_t e) {
        if (e != hipSuccess && rc == FLEX_OK) {
            note_hip_error(e);
            rc = e == hipErrorOutOfMemory ? FLEX_ERR_NOMEM : FLEX_ERR_HIP;
        }
        return e != hipSuccess;
    };
    const dim3 grid(static_cast<unsigned>(prop.multiProcessorCount) * 8u), block(256);
    float ms_read = 0.f, ms_copy = 0.f;
    do {
        if (fail(hipMalloc(&src, n_vec * 16)) || fail(hipMalloc(&dst, n_vec * 16)) || fail(hipMalloc(reinterpret_cast<void **>(&sink), 4))) break;
        if (fail(hipMemset(src, 0, n_vec * 16)) || fail(hipMemset(dst, 0, n_vec * 16))) break;
        if (fail(hipEventCreate(&e0)) || fail(hipEventCreate(&e1))) break;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up
            const int n = pass == 0 ? 2 : reps;
            if (fail(hipEventRecord(e0, nullptr))) break;
            for (int i = 0; i < n; ++i)
                if (temporal)
                    hipLaunchKernelGGL(probe_read_kernel<false>, grid, block, 0, nullptr, static_cast<const v4f_t *>(src), n_vec, sink);
                else
                    hipLaunchKernelGGL(probe_read_kernel<true>, grid, block, 0, nullptr, static_cast<const v4f_t *>(src), n_vec, sink);
            if (fail(hipEventRecord(e1, nullptr)) || fail(hipEventSynchronize(e1)) || fail(hipEventElapsedTime(&ms_read, e0, e1))) break;
            if (fail(hipEventRecord(e0, nullptr))) break;
            for (int i = 0; i < n; ++i)
                if (temporal)
                    hipLaunchKernelGGL(probe_copy_kernel<false>, grid, block, 0, nullptr, static_cast<const v4f_t *>(src), static_cast<v4f_t *>(dst), n_vec);
                else
                    hipLaunchKernelGGL(probe_copy_kernel<true>, grid, block, 0, nullptr, static_cast<const v4f_t *>(src), static_cast<v4f_t *>(dst), n_vec);
            if (fail(hipEventRecord(e1, nullptr)) || fail(hipEventSynchronize(e1)) || fail(hipEventElapsedTime(&ms_copy, e0, e1))) break;
        }
        (void)fail(hipGetLastError());
    } while (false);
    if (rc == FLEX_OK) {
        *read_gbps = static_cast<double>(n_vec) * 16.0 * reps / (ms_read * 1e-3) / 1e9;
        *copy_gbps = 2.0 * static_cast<double>(n_vec) * 16.0 * reps / (ms_copy * 1e-3) / 1e9;  // bytes read + bytes written
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    (void)hipFree(sink);
    (void)hipSetDevice(prev);
    return rc;
}
