#include "Mat.h"

Mat::Mat(DataLoader &input, int tileh, int tilew)
    : m(static_cast<int>(input.n)), n(static_cast<int>(input.n)), k(static_cast<int>(input.dim)),
      nnz(static_cast<int>(input.nnz)), tm(tileh), tn(tilew), dl(input), rowPtr(input.rowPtr), colIdx(input.col),
      vals(input.vals) {}

void Mat::csr2_DiagTiling() {
    alpha_freeMatGPU();
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    const flex_csr a = dl.csr_view();
    // a reordered loader hands its vo_mp so that B stays un-permuted and C comes back in the
    // original row order (the reference needs permuteX + segVoMap for that)
    const int32_t *vo = dl.vertex_order_abbr == "OVO" ? nullptr : dl.vo_mp.data();
    FLEX_CHECK(flex_plan_create_mapped(&plan, &a, vo, k, device, schedule));
}

void Mat::launch_prep() {
    dl.gpuC_zero();
    mat_b_dev = dl.gpuX;
    mat_c_dev = dl.gpuC;
}

void Mat::launch(hipStream_t s) { FLEX_CHECK(flex_spmm(plan, mat_b_dev, mat_c_dev, reinterpret_cast<flex_stream_t>(s))); }

void Mat::alpha_freeMatGPU() {
    if (plan) flex_plan_destroy(plan);
    plan = nullptr;
}

flex_plan_info Mat::info() const {
    flex_plan_info i{};
    if (plan) flex_plan_get_info(plan, &i);
    return i;
}
