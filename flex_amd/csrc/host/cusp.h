// cusp.h (host mirror) -- ≙ cusp.cuh: the two orders of the GCN layer product A*X*W.
#pragma once
#include "../../../include/flex_axw.h"
#include "DataLoader.h"

int run1(DataLoader &input, Metrics &metric);  // ≙ cusp.cu:3-104:   B = X*W (SGEMM), C = A*B (SpMM at k = c) -> cpuRef1
int run2(DataLoader &input, Metrics &metric);  // ≙ cusp.cu:106-208: B = A*X (SpMM at k = dim), C = B*W (SGEMM) -> cpuRef2
int run_axw(DataLoader &data);                 // ≙ the AXW block of main.cu:22-77 (5 warm-up + 10 timed of each order)
