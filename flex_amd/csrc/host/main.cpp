// main.cpp (host mirror) -- ≙ main.cu:7-83 (the non-AXW path): flex <csv|synth:name[*scale]> <k>
// [--iters N] [--warmup N] [--json] [--stats] [--perm-cache DIR] [--csv FILE] [--stats-log FILE] [--no-vendor] [--gpus N] [--axw] [--debug-values] [--counters]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "cusp.h"
#include "flex.h"

int main(int argc, char *argv[]) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <graph.csv | synth:name[*scale]> <k> [--iters N] [--warmup N] [--json] [--stats] [--perm-cache DIR] [--csv FILE] [--stats-log FILE] [--no-vendor] [--gpus N] [--axw] [--debug-values] [--counters]\n", argv[0]);
        return 2;
    }
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--json")) run_options().json = true;
        else if (!std::strcmp(argv[i], "--stats")) run_options().stats = true;
        else if (!std::strcmp(argv[i], "--axw")) run_options().axw = true;
        else if (!std::strcmp(argv[i], "--debug-values")) run_options().debug_values = true;
        else if (!std::strcmp(argv[i], "--counters")) run_options().counters = true;
        else if (!std::strcmp(argv[i], "--perm-cache") && i + 1 < argc) run_options().perm_cache = argv[++i];
        else if (!std::strcmp(argv[i], "--csv") && i + 1 < argc) run_options().csv = argv[++i];
        else if (!std::strcmp(argv[i], "--stats-log") && i + 1 < argc) run_options().stats_log = argv[++i];
        else if (!std::strcmp(argv[i], "--no-vendor")) run_options().vendor = false;
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) run_options().gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) run_options().iters = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc) run_options().warmup = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    std::setvbuf(stdout, nullptr, _IOLBF, 0);  // a table row is visible as soon as it is measured, also in a log file
    try {
        if (run_options().counters) counters_attach();  // before the loader makes the first HIP call
        DataLoader data(argv[1], std::atoi(argv[2]));
        std::cout << "Graph name: " << data.graph_name << std::endl;
        std::cout << "A: " << data.n << "*" << data.n << "  X: " << data.n << "*" << data.dim << "   W: " << data.dim
                  << "*" << data.c << std::endl;
        std::printf("Avg degree: %.1f   %s,  n one-way edges %zd, asymmetric %zd\n", double(data.nnz) / data.n,
                    data.is_directed ? "Directed" : "Undirected", data.n_edges_one_way, data.n_edges_asymmetric);
        std::printf("Nodes zero-deg-in %d, zero-deg-out %d, zero-deg %d\n", data.n_nodes_z_in, data.n_nodes_z_out,
                    data.n_nodes_z_deg);
        std::cout << "NNZ of A: " << data.nnz << std::endl;
        if (run_options().axw) return run_axw(data);  // ≙ #ifdef AXW, main.cu:22-77
        run(data);  // flex.h / run.cpp
    } catch (const std::exception &e) {
        std::fprintf(stderr, "flex: %s\n", e.what());
        return 1;
    }
    return 0;
}
