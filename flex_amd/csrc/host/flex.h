// flex.h (host mirror) -- ≙ flex.cuh:59-60: the two entry points main() uses.
#pragma once
#include "DataLoader.h"
#include "Mat.h"

struct RunOptions {
    int warmup = 5, iters = 10;  // the reference's vendor protocol, flex.cu:5766-5789
    bool json = false;           // one JSON line per configuration instead of the table only
    bool vendor = true;          // run hipSPARSE as gold + baseline (false: CPU-free self check only)
    bool stats = false;          // print each plan's imbalance / reuse summary (≙ alpha_stats_collect)
    std::string csv;             // append one line per configuration (≙ flex-tile-nperf.csv, flex.cu:4945-4947)
    std::string stats_log;       // write every plan's summary there (≙ flex-tile-stats2.log, flex.cu:4943-4944)
    std::string perm_cache;      // directory of cached orderings (<graph>.<ORD>.perm); empty = recompute every run
    bool debug_values = false;   // ≙ opt_debug (DataLoader.cu:7, 51, 202-203): every A value 1, X[i][*] = i -- results readable by eye
    bool axw = false;            // run the GCN layer product A*X*W both ways instead of the SpMM loop (main.cu:22-77)
    int gpus = 0;                // > 0: also run the row-sharded multi-GPU path on that many devices (flex_mg.h)
    bool counters = false;       // read the card's memory counters around each configuration's launches (≙ the NPerf metrics of
                                 // flex.cu:4583-4656): HBM-side bytes, L2 hit rate and the measured B reuse u as table columns
};
RunOptions &run_options();
// --counters: loads libflex_counters.so (include/flex_counters.h) and asks for the profiler; must run before the first HIP call
void counters_attach();

void run(DataLoader &input);                      // ≙ flex.cu:4560-5716: bench loop over orderings
void cuSpmm(DataLoader &input, Perfs &perfRes);   // ≙ flex.cu:5717-5804: vendor SpMM (hipSPARSE) -> input.gpuC
// ≙ resCheck, flex.cu:4154-4213; returns the number of elements beyond 4*eps*row_nnz
int resCheck(const float *h_gold, float *h_res, const Mat &mat, Perfs &perfRes, double *max_err_out = nullptr);
