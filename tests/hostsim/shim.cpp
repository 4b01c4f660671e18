// tests/hostsim/shim.cpp -- TEST INFRASTRUCTURE: lets the planner's HOST code run without a GPU.
// The planner (flex_amd/csrc/plan_build.cpp) ends by uploading its arrays with hipMalloc/hipMemcpy, and flex_plan_self_check
// reads that image back; linked against this file instead of the kernels, "device memory" is malloc'ed host memory, so the
// CPU suite (and the ASan/UBSan build of tools/asan_host.sh) can create every kind of plan and verify the image the kernels
// would read.  There is no compute here: every launcher reports FLEX_ERR_UNSUPPORTED, flex_spmm cannot produce a result.
#include <cstdlib>
#include <cstring>

#include "internal.h"

namespace flex {
int launch_spmm(const PlanView &, int, bool, bool, const float *, float *, hipStream_t, int) { return FLEX_ERR_UNSUPPORTED; }
int launch_spmm_stamped(const PlanView &, int, bool, const float *, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_tiles(const TileView &, bool, const float *, float *, int, int, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_blocks(const BlockView &, const float *, float *, hipStream_t, bool) { return FLEX_ERR_UNSUPPORTED; }
int launch_fixup(const float *, const SplitRow *, uint32_t, int, int, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_gather_rows(float *, const float *, const int32_t *, int64_t, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int kernel_attributes(int, bool, bool, hipFuncAttributes *, int *) { return FLEX_ERR_UNSUPPORTED; }
}  // namespace flex

extern "C" {
int flex_hbm_probe(int, int64_t, int, int, double *, double *) { return FLEX_ERR_UNSUPPORTED; }
#ifdef FLEX_HOSTSIM  // malloc-backed stand-ins for the few HIP runtime calls the planner makes (bound locally: -Bsymbolic-functions)
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static uint64_t g_upload_hash = 1469598103934665603ull;  // FNV-1a over every byte "uploaded" since the last reset: a fingerprint of the plan image
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    if (n) std::memcpy(d, s, n);
    const unsigned char *b = static_cast<const unsigned char *>(s);
    uint64_t h = g_upload_hash ^ n;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    g_upload_hash = h;
    return hipSuccess;
}
uint64_t hostsim_upload_hash(int reset) {
    const uint64_t h = g_upload_hash;
    if (reset) g_upload_hash = 1469598103934665603ull;
    return h;
}
hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hostsim"; }
// the in-flight guard of flex_spmm: nothing is ever in flight here
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
#endif
}
