// Mat.h (host mirror) -- ≙ Mat / Mat_POD (mat.cuh:18-222).  The reference's Mat holds ~30
// device pointers of its pillar/segment formats and copies itself into __constant__ memory;
// here a Mat is a handle on one engine plan (flex_plan).  The method names that run()'s loop
// calls (flex.cu:4951-5059, 5690-5701) are kept so that loop reads the same.
#pragma once
#include "DataLoader.h"

class Mat {
   public:
    int m, n, k, nnz;
    int tm, tn;  // tile shape of the reference's formats; carried for the report only
    DataLoader &dl;
    std::vector<unsigned int> &rowPtr;
    std::vector<unsigned int> &colIdx;
    std::vector<float> &vals;
    unsigned schedule = FLEX_ORDER_NATURAL;  // engine-side row schedule applied on top of dl's order
    flex_plan *plan = nullptr;
    float *mat_b_dev = nullptr, *mat_c_dev = nullptr;

    Mat(DataLoader &input, int tileh, int tilew);  // mat.cu:7-31
    ~Mat() { alpha_freeMatGPU(); }
    Mat(const Mat &) = delete;

    void csr2_DiagTiling();    // ≙ mat.cu:680-942: here, the row-panel planner (flex_plan_create_mapped)
    void alpha_transfer() {}   // ≙ mat.cu:268-293: the plan was uploaded when it was created
    void launch_prep();        // ≙ mat.cu:32-41: zero C, bind B and C
    void launch(hipStream_t s = nullptr);  // ≙ kernel<<<grid,block>>>() (flex.cu:5059)
    void alpha_freeMatGPU();   // ≙ mat.cuh:184-193
    int row_nnz_get(int r) const { return static_cast<int>(rowPtr[r + 1] - rowPtr[r]); }
    flex_plan_info info() const;
    flex_plan_stats stats() const;
    void alpha_stats_collect(FILE *stream) const;  // ≙ mat.cu:944-1065: imbalance / reuse summary of the plan
};
