// common.h (host mirror) -- ≙ common.h:14-118 of the reference: Metrics / Perfs records and
// the error macros, with HIP in place of CUDA.  HIP_CHECK keeps CUDA_CHECK's contract: print
// "HIP error <n> at file:line" and throw std::runtime_error (common.h:53-60).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <vector>

#include "../../../include/flex_spmm.h"

struct Metrics {  // common.h:14-37
    float t = 0.0f, spmm_t = 0.0f, gemm_t = 0.0f;
    float flops = 0.0f, spmm_flops = 0.0f, gemm_flops = 0.0f, dataMovement = 0.0f;
};

class Perfs {  // common.h:39-51 (cuSpmm* names kept: they label the vendor baseline)
   public:
    float cuSpmmSetup = 0, cuSpmmProcessing = 0, cuSpmm_time = 0, cuspmm_throughput = 0, cuspmm_bandwidth = 0;
    std::vector<float> flex_spmm_time, flex_spmm_throughput, flex_spmm_bandwidth;
    std::vector<int> flex_spmm_errors;
};

#define HIP_CHECK(err)                                                         \
    do {                                                                       \
        hipError_t err_ = (err);                                               \
        if (err_ != hipSuccess) {                                              \
            std::printf("HIP error %d at %s:%d\n", err_, __FILE__, __LINE__);  \
            throw std::runtime_error("HIP error");                             \
        }                                                                      \
    } while (0)

// the engine's C ABI never throws; the host mirror turns its status codes into exceptions
#define FLEX_CHECK(expr)                                                                            \
    do {                                                                                            \
        int st_ = (expr);                                                                           \
        if (st_ != FLEX_OK) {                                                                       \
            std::printf("flex error %d (%s) at %s:%d\n", st_, flex_strerror(st_), __FILE__, __LINE__); \
            throw std::runtime_error(flex_strerror(st_));                                           \
        }                                                                                           \
    } while (0)

template <typename T>
inline void hip_freez(T *&ptr_dev) {  // ≙ cuda_freez, common.h:92-99
    if (!ptr_dev) return;
    HIP_CHECK(hipFree(ptr_dev));
    ptr_dev = nullptr;
}
