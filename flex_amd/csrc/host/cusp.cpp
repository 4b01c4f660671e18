// cusp.cpp (host mirror) -- ≙ cusp.cu: run1 = A*(X*W), run2 = (A*X)*W, with the engine's SpMM in
// place of cusparseSpMM and rocBLAS in place of cuBLAS (libflex_axw.so), and the driver block of
// main.cu:22-77 that the reference keeps behind `//#define AXW 1`.
#include "cusp.h"

#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>

#include "flex.h"
#include "lazy_lib.h"

void DataLoader::axw_alloc() {  // ≙ the AXW part of cuda_alloc_cpy, DataLoader.cu:169-176 (+ cpuRef1/2, DataLoader.cuh:47-48)
    if (axw) return;
    const flex_csr a = csr_view();
    FLEX_CHECK(FLEX_AXW(flex_axw_create)(&axw, &a, static_cast<int>(dim), static_cast<int>(c), 0, FLEX_ORDER_CLUSTER));
    cpuW.resize(dim * c);
    for (float &w : cpuW) w = static_cast<float>(std::rand()) / static_cast<float>(RAND_MAX);  // DataLoader.cu:172
    HIP_CHECK(hipMalloc(&gpuW, sizeof(float) * dim * c));
    HIP_CHECK(hipMemcpy(gpuW, cpuW.data(), sizeof(float) * dim * c, hipMemcpyHostToDevice));
    const size_t ld = static_cast<size_t>(FLEX_AXW(flex_axw_ld)(static_cast<int>(c)));
    HIP_CHECK(hipMalloc(&gpuRef1, sizeof(float) * n * ld));
    HIP_CHECK(hipMalloc(&gpuRef2, sizeof(float) * n * ld));
    cpuRef1.resize(n * ld);
    cpuRef2.resize(n * ld);
}

bool DataLoader::compare() {  // DataLoader.cu:859-869, with a relative bound in place of the absolute 0.1
    const size_t ld = static_cast<size_t>(FLEX_AXW(flex_axw_ld)(static_cast<int>(c)));
    for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < c; ++j) {
            const float a = cpuRef1[i * ld + j], b = cpuRef2[i * ld + j];
            if (!(std::fabs(a - b) <= 1e-3f * std::fmax(1.0f, std::fmax(std::fabs(a), std::fabs(b))))) {
                std::cout << "Ref1[" << i * c + j << "]=" << std::setprecision(12) << a << " / Ref2[" << i * c + j
                          << "]=" << std::setprecision(12) << b << std::endl;
                return false;
            }
        }
    std::cout << "The results are correct.. " << std::endl;
    return true;
}

namespace {

int run_order(DataLoader &input, Metrics &metric, int order, float *gpuRef, std::vector<float> &cpuRef) {
    float gemm_ms = 0.f, spmm_ms = 0.f;
    FLEX_CHECK(FLEX_AXW(flex_axw_run)(input.axw, order, input.gpuX, input.gpuW, gpuRef, nullptr, &gemm_ms, &spmm_ms));
    metric.t += gemm_ms + spmm_ms;
    metric.spmm_t += spmm_ms;
    metric.gemm_t += gemm_ms;
    const double n = static_cast<double>(input.n), nnz = static_cast<double>(input.nnz), d = static_cast<double>(input.dim),
                 c = static_cast<double>(input.c);
    const double k_spmm = order == FLEX_AXW_A_XW ? c : d;  // cusp.cu:88-90 / 192-194
    metric.spmm_flops = static_cast<float>(2.0 * nnz * k_spmm);
    metric.gemm_flops = static_cast<float>(2.0 * n * d * c);
    metric.flops = metric.spmm_flops + metric.gemm_flops;
    metric.dataMovement = static_cast<float>(4.0 * (nnz + n * d + d * c + 2.0 * n * k_spmm));  // cusp.cu:92
    HIP_CHECK(hipMemcpy(cpuRef.data(), gpuRef, sizeof(float) * cpuRef.size(), hipMemcpyDeviceToHost));
    return 0;
}

void report(const char *title, const Metrics &m, int iters) {  // main.cu:58-76
    const double t = m.t * 1e-3 / iters, tg = m.gemm_t * 1e-3 / iters, ts = m.spmm_t * 1e-3 / iters;
    std::cout << title << "  " << std::endl;
    std::cout << "        " << m.flops / 1e6 << " Mflops" << std::endl;
    std::cout << "        " << t << " s" << std::endl;
    std::cout << "        " << m.flops / t / 1e9 << " Gflops/s" << std::endl;
    std::cout << "   gemm:" << tg << " s   " << m.gemm_flops / tg / 1e9 << " Gflops/s" << std::endl;
    std::cout << "   spmm:" << ts << " s   " << m.spmm_flops / ts / 1e9 << " Gflops/s" << std::endl;
}

void add(Metrics &a, const Metrics &b) {  // Metrics::operator+=, common.h:23-30
    a.t += b.t;
    a.spmm_t += b.spmm_t;
    a.gemm_t += b.gemm_t;
    a.flops = b.flops;
    a.spmm_flops = b.spmm_flops;
    a.gemm_flops = b.gemm_flops;
}

}  // namespace

int run1(DataLoader &input, Metrics &metric) {
    input.axw_alloc();  // before the buffers are named below
    return run_order(input, metric, FLEX_AXW_A_XW, input.gpuRef1, input.cpuRef1);
}
int run2(DataLoader &input, Metrics &metric) {
    input.axw_alloc();
    return run_order(input, metric, FLEX_AXW_AX_W, input.gpuRef2, input.cpuRef2);
}

int run_axw(DataLoader &data) {
    const int WarmupIterations = 5, ExecutionIterations = 10;  // main.cu:24-25
    Metrics baselinemetrics1, baselinemetrics2;
    for (int i = 0; i < WarmupIterations; ++i) {
        Metrics metric0;
        run1(data, metric0);
        run2(data, metric0);
    }
    for (int i = 0; i < ExecutionIterations; ++i) {
        Metrics metric1;
        run1(data, metric1);  // step 1: B = XW, step 2: C = AB
        add(baselinemetrics1, metric1);
        Metrics metric2;
        run2(data, metric2);  // step 1: B = AX, step 2: C = BW
        add(baselinemetrics2, metric2);
        if (!data.compare()) {
            std::cout << "The results are wrong ..." << std::endl;
            return 1;
        }
    }
    report("A(XW):", baselinemetrics1, ExecutionIterations);
    report("(AX)W:", baselinemetrics2, ExecutionIterations);
    if (run_options().json)
        std::printf("{\"graph\":\"%s\",\"dim\":%zu,\"c\":%zu,\"a_xw_ms\":%.4f,\"ax_w_ms\":%.4f,\"a_xw_spmm_ms\":%.4f,"
                    "\"ax_w_spmm_ms\":%.4f,\"a_xw_gemm_ms\":%.4f,\"ax_w_gemm_ms\":%.4f}\n",
                    data.graph_name.c_str(), data.dim, data.c, baselinemetrics1.t / ExecutionIterations,
                    baselinemetrics2.t / ExecutionIterations, baselinemetrics1.spmm_t / ExecutionIterations,
                    baselinemetrics2.spmm_t / ExecutionIterations, baselinemetrics1.gemm_t / ExecutionIterations,
                    baselinemetrics2.gemm_t / ExecutionIterations);
    return 0;
}
