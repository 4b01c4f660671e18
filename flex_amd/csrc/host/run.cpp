// run.cpp (host mirror) -- ≙ run / cuSpmm / resCheck of flex.cu (4560-5716, 5717-5804,
// 4154-4213).  Same loop shape -- vendor gold first, then one (ordering x schedule) configuration
// after another: plan, launch_prep, warm-up, timed launches, copy back, resCheck, free -- with
// NPerf/pTable replaced by HIP events and printf, and GFLOP/s = 2e-9*nnz*k/t (flex.cu:5629).
#include <dlfcn.h>

#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>

#include "../../../include/flex_mg.h"
#include "../../../include/flex_vendor.h"
#include "flex.h"
#include "lazy_lib.h"

RunOptions &run_options() {
    static RunOptions o;
    return o;
}

// libflex_counters.so is loaded only on request (and with RTLD_GLOBAL: the ROCm runtime looks its rocprofiler_configure up
// over the whole process when it initialises), so a run without --counters never has the profiler in the process.
namespace {
struct CountersLib {
    int (*init)() = nullptr;
    int (*begin)(int, const char *const *, int) = nullptr;
    int (*end)(double *) = nullptr;
    const char *(*error)() = nullptr;
} g_counters;

// one pass around `launches` launches of `mat`; returns the per-launch sums (false + message on stderr when the profiler refuses)
bool counted(Mat &mat, int launches, const char *const *names, int n, double *out) {
    if (std::getenv("FLEX_COUNTERS_DEBUG")) std::fprintf(stderr, "flex --counters: pass %s (+%d)\n", names[0], n - 1);
    HIP_CHECK(hipDeviceSynchronize());
    if (g_counters.begin(0, names, n) != 0) {
        std::fprintf(stderr, "flex --counters: %s\n", g_counters.error());
        return false;
    }
    for (int i = 0; i < launches; ++i) mat.launch();
    HIP_CHECK(hipDeviceSynchronize());
    if (g_counters.end(out) != 0) {
        std::fprintf(stderr, "flex --counters: %s\n", g_counters.error());
        return false;
    }
    for (int i = 0; i < n; ++i) out[i] /= launches;
    return true;
}
}  // namespace

void counters_attach() {
    void *h = dlopen("libflex_counters.so", RTLD_NOW | RTLD_GLOBAL);  // next to the binary (RUNPATH $ORIGIN)
    if (!h) throw std::runtime_error(std::string("--counters: ") + dlerror());
    g_counters.init = reinterpret_cast<int (*)()>(dlsym(h, "flex_counters_init"));
    g_counters.begin = reinterpret_cast<int (*)(int, const char *const *, int)>(dlsym(h, "flex_counters_begin"));
    g_counters.end = reinterpret_cast<int (*)(double *)>(dlsym(h, "flex_counters_end"));
    g_counters.error = reinterpret_cast<const char *(*)()>(dlsym(h, "flex_counters_error"));
    if (!g_counters.init || !g_counters.begin || !g_counters.end || !g_counters.error)
        throw std::runtime_error("--counters: libflex_counters.so lacks an entry point of include/flex_counters.h");
    if (g_counters.init() != 0) throw std::runtime_error(std::string("--counters: ") + g_counters.error());
}

void cuSpmm(DataLoader &input, Perfs &perfRes) {
    hipEvent_t e0, e1, e2, e3;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventCreate(&e2));
    HIP_CHECK(hipEventCreate(&e3));
    flex_vendor *h = nullptr;
    HIP_CHECK(hipEventRecord(e0, nullptr));
    const int rc = FLEX_VENDOR(flex_vendor_spmm_create)(&h, static_cast<int32_t>(input.m), static_cast<int32_t>(input.n),
                                           static_cast<int64_t>(input.nnz), input.rowPtr_dev, input.col_dev,
                                           input.vals_dev, static_cast<int>(input.dim), input.gpuX, input.gpuC);
    if (rc) {  // the reference's CHECK_CUSPARSE only prints (common.h:81-90); a missing gold is fatal here
        std::printf("hipSPARSE API failed: code %d status %d\n", rc, FLEX_VENDOR(flex_vendor_last_status)());
        throw std::runtime_error("hipSPARSE SpMM setup failed");
    }
    HIP_CHECK(hipEventRecord(e1, nullptr));
    const auto vendor_run = FLEX_VENDOR(flex_vendor_spmm_run);  // resolved once: the timed loop calls through the pointer
    for (int i = 0; i < 5; ++i) vendor_run(h, nullptr);  // warm-up, flex.cu:5766-5773
    HIP_CHECK(hipEventRecord(e2, nullptr));
    for (int i = 0; i < 10; ++i) vendor_run(h, nullptr);
    HIP_CHECK(hipEventRecord(e3, nullptr));
    HIP_CHECK(hipEventSynchronize(e3));
    float setup_ms = 0, proc_ms = 0;
    HIP_CHECK(hipEventElapsedTime(&setup_ms, e0, e1));
    HIP_CHECK(hipEventElapsedTime(&proc_ms, e2, e3));
    perfRes.cuSpmmSetup = setup_ms * 1e3f;
    perfRes.cuSpmmProcessing = proc_ms * 1e3f / 10;  // microseconds per SpMM
    perfRes.cuSpmm_time = perfRes.cuSpmmSetup + perfRes.cuSpmmProcessing;
    FLEX_VENDOR(flex_vendor_spmm_destroy)(h);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    (void)hipEventDestroy(e3);
}

int resCheck(const float *h_gold, float *h_res, const Mat &mat, Perfs &perfRes, double *max_err_out) {
    const int m = mat.m, k = mat.k;
    int count = 0, err_show_remaining = 20, nz = 0, me_nnz = 0, last_err_row = -1;
    double max_err = 0;
    const auto &rp = mat.dl.dl_original->rowPtr;  // tolerance uses the ORIGINAL order's row lengths (flex.cu:4170)
    for (int r = 0; r < m; ++r) {
        const int row_nnz = static_cast<int>(rp[r + 1] - rp[r]);
        const double tol = static_cast<double>(FLT_EPSILON) * row_nnz * 4;
        for (int c = 0; c < k; ++c) {
            const size_t idx = static_cast<size_t>(r) * k + c;
            if (h_gold[idx] == 0) nz++;
            const double err = std::fabs(h_gold[idx]) < 1 ? std::fabs(double(h_gold[idx]) - h_res[idx])
                                                          : std::fabs(1.0 - double(h_res[idx]) / h_gold[idx]);
            if (err > max_err) {
                max_err = err;
                me_nnz = row_nnz;
            }
            if (err > tol || err != err) {
                count++;
                if (r != last_err_row && err_show_remaining-- > 0) {
                    last_err_row = r;
                    std::printf(" ref[%d][%d]:  %f!=%f (correct)  %d nnzs, dif %g, tol %g\n", r, c, h_res[idx],
                                h_gold[idx], row_nnz, err, tol);
                }
            }
        }
    }
    perfRes.flex_spmm_errors.push_back(count);
    if (count) std::printf("Kernel errs: %d Max err %g at nnz=%d.\n", count, max_err, me_nnz);
    if (nz >= m * k / 2) std::printf("warning: gold is mostly zeros (%d of %d), errors are hard to catch\n", nz, m * k);
    if (max_err_out) *max_err_out = max_err;
    std::memset(h_res, 0, sizeof(float) * static_cast<size_t>(m) * k);
    return count;
}

namespace {

struct Row {
    std::string ord, sched;
    double t_us, gflops, balg_gbs, plan_ms, max_err;
    int errs;
    flex_plan_info info;
    flex_plan_stats stats;
    flex_imbalance imb;  // ≙ the "Imb" column (flex.cu:5087-5126): one stamped launch after the timed ones
    // --counters (≙ the DRAM GB/s, %Pk and measured-u columns, flex.cu:5237, 5513-5528): HBM-side bytes per launch
    // (2 x FETCH_SIZE + WRITE_SIZE in KiB: the gfx950 correction of MI355X_MICROARCH "HBM"), L2 hit rate, and
    // u = B bytes the nonzeros ask for / bytes the L2 fetched beyond A's; all < 0 when not measured
    double hbm_bytes = -1, l2_hit = -1, u_meas = -1;
    // ≙ "L2/" (measured L1<->L2 bytes, here requests x 128 B) and "Per Mult / Num Insns" (flex.cu:5279-5330, 5350-5420):
    // wave instructions per 64 multiply-adds; < 0 when not measured
    double l2_bytes = -1, vmem_rd = -1, valu = -1, lds = -1, salu = -1, u_l1 = -1;
};

void bench_one(DataLoader &dl, unsigned schedule, const char *sched_name, const DataLoader &gold_src, float *h_res,
               Perfs &perfRes, std::vector<Row> &rows, FILE *stats_log) {
    const RunOptions &o = run_options();
    Mat mat(dl, 0, 0);
    mat.schedule = schedule;
    mat.csr2_DiagTiling();
    mat.alpha_transfer();
    mat.launch_prep();
    if (o.stats) {
        std::printf("%s/%s:\n", dl.vertex_order_abbr.c_str(), sched_name);
        mat.alpha_stats_collect(stdout);
    }
    if (stats_log) {
        std::fprintf(stats_log, "%s/%s k=%d:\n", dl.vertex_order_abbr.c_str(), sched_name, mat.k);
        mat.alpha_stats_collect(stats_log);
    }
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < o.warmup; ++i) mat.launch();
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < o.iters; ++i) mat.launch();
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double t_us = ms * 1e3 / o.iters;
    HIP_CHECK(hipMemcpy(h_res, mat.mat_c_dev, dl.gpuC_bytes, hipMemcpyDeviceToHost));
    double max_err = 0;
    int errs = -1;
    if (!gold_src.h_ref_c.empty()) errs = resCheck(gold_src.h_ref_c.data(), h_res, mat, perfRes, &max_err);
    const double flops = 2.0 * dl.nnz * dl.dim;
    const double balg = (double(dl.n) + 1 + 2.0 * dl.nnz + 2.0 * dl.n * dl.dim) * 4;  // flex.cu:4672, 5795
    double hbm_bytes = -1, l2_hit = -1, u_meas = -1, l2_bytes = -1, vmem_rd = -1, valu = -1, lds = -1, salu = -1, u_l1 = -1;
    if (o.counters) {  // after the timed launches: three passes (the TCC block cannot hold FETCH_SIZE and WRITE_SIZE at once)
        const char *fetch[] = {"FETCH_SIZE"}, *write[] = {"WRITE_SIZE"}, *l2[] = {"TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"};
        const char *sq[] = {"SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"};
        double f = 0, w = 0, hm[4] = {0, 0, 0, 0}, in[4] = {0, 0, 0, 0};
        if (counted(mat, o.iters, fetch, 1, &f) && counted(mat, o.iters, write, 1, &w) && counted(mat, o.iters, l2, 4, hm)) {
            l2_bytes = 128.0 * hm[2];
            {   // the reference's own u (flex.cu:5513-5528): nD = bytes L1 reads from L2 per multiply-add = 4/u + A's share (here the
                // record stream, once per column tile of 4 x lanes_per_nz columns)
                const double ktiles = std::ceil(double(dl.dim) / (4.0 * std::max(1, mat.info().lanes_per_nz)));
                const double nd = 128.0 * hm[3] / (double(dl.nnz) * dl.dim);
                u_l1 = 4.0 / std::max(nd - 8.0 * ktiles / dl.dim, 1e-9);
            }
            if (counted(mat, o.iters, sq, 4, in)) {
                const double per = double(dl.nnz) * dl.dim / 64.0;
                vmem_rd = in[0] / per, valu = in[1] / per, lds = in[2] / per, salu = in[3] / per;
            }
            const double rd = 2.0 * 1024.0 * f, a_bytes = 8.0 * dl.nnz + 4.0 * (dl.n + 1);
            hbm_bytes = rd + 1024.0 * w;
            l2_hit = hm[0] / std::max(1.0, hm[0] + hm[1]);
            if (rd - a_bytes > 0.01 * 4.0 * dl.n * dl.dim) u_meas = 4.0 * dl.nnz * dl.dim / (rd - a_bytes);  // else: the L2s held B, nothing to divide by
        }
    }
    flex_imbalance imb{};
    if (flex_plan_measure_imbalance(mat.plan, mat.mat_b_dev, mat.mat_c_dev, nullptr, &imb) != FLEX_OK) imb = flex_imbalance{};  // odd k: no stamped twin
    rows.push_back({dl.vertex_order_abbr, sched_name, t_us, flops / t_us * 1e-3, balg / t_us * 1e-3,
                    mat.info().plan_ms, max_err, errs, mat.info(), mat.stats(), imb, hbm_bytes, l2_hit, u_meas, l2_bytes, vmem_rd, valu, lds, salu, u_l1});
    perfRes.flex_spmm_time.push_back(static_cast<float>(t_us * 1e-3));
    mat.alpha_freeMatGPU();
}

}  // namespace

void run(DataLoader &input_vo) {
    const RunOptions &o = run_options();
    Perfs perfRes;
    if (o.vendor) input_vo.c_cuSpmm_run(perfRes);  // gold + vendor baseline (flex.cu:4569)
    std::unique_ptr<float[]> h_res(new float[std::max<int64_t>(input_vo.C_elts, 1)]);
    std::vector<Row> rows;
    FILE *stats_log = o.stats_log.empty() ? nullptr : std::fopen(o.stats_log.c_str(), "w");
    if (!o.stats_log.empty() && !stats_log) throw std::runtime_error("cannot open " + o.stats_log);
    if (stats_log) std::printf("Writing detailed statistics to file %s\n", o.stats_log.c_str());

    // engine-side schedules on the original loader (no permuted CSR, no permuteX pass)
    bench_one(input_vo, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
    bench_one(input_vo, FLEX_ORDER_RCM, "rcm", input_vo, h_res.get(), perfRes, rows, stats_log);
    bench_one(input_vo, FLEX_ORDER_CLUSTER, "cluster", input_vo, h_res.get(), perfRes, rows, stats_log);
    // the reference's flow: reordered loaders (flex.cu:4572-4576), plan folds vo_mp back in
    {
        DataLoaderRcm rcm(input_vo);
        bench_one(rcm, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
        // BASELINE configs[2] as worded -- an RCM-reordered loader is the INPUT (DataLoader.cu:723-787); how its rows are scheduled
        // on eight private L2s is the engine's: the community schedule on top of the loader's order
        bench_one(rcm, FLEX_ORDER_CLUSTER, "cluster", input_vo, h_res.get(), perfRes, rows, stats_log);
    }
    {
        DataLoaderRabbit rbt(input_vo);
        bench_one(rbt, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
    }
    {
        DataLoaderDFS dfs(input_vo);
        bench_one(dfs, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
    }
    try {
        DataLoaderGorder gor(input_vo);
        bench_one(gor, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
    } catch (const std::runtime_error &e) {
        // Gorder cannot order a graph with an isolated vertex (nor can the reference, unitheap.cu:35-38)
        std::printf("GOR  skipped: %s\n", e.what());
    }
    {
        DataLoaderDeg deg(input_vo);
        bench_one(deg, FLEX_ORDER_NATURAL, "natural", input_vo, h_res.get(), perfRes, rows, stats_log);
    }

    if (stats_log) std::fclose(stats_log);
    if (!o.csv.empty()) {  // appended, graph name first, one line per configuration (flex.cu:4945-4947, 5135-5540)
        FILE *csv = std::fopen(o.csv.c_str(), "a");
        if (!csv) throw std::runtime_error("cannot open " + o.csv);
        std::fprintf(csv, "%s\n", input_vo.graph_name.c_str());
        std::fprintf(csv, "ord,sched,k,chunks,split,B-Re1,B-Re2,chunk_imb%%,xcd_imb%%,t/us,GFLOP/s,Balg_GB/s,plan_MB,X_MB,C_MB,vendor/us,errs\n");
        for (const Row &r : rows)
            std::fprintf(csv, "%3s,%s,%zu,%lld,%lld,%5.2f,%5.2f,%.0f,%.1f,%7.1f,%.1f,%.1f,%6.2f,%6.2f,%6.2f,%.1f,%d\n",
                         r.ord.c_str(), r.sched.c_str(), input_vo.dim, static_cast<long long>(r.info.n_chunks),
                         static_cast<long long>(r.info.n_split_rows), r.stats.reuse_wave, r.stats.reuse_xcd,
                         r.stats.chunk_imb_pct, r.stats.xcd_imb_pct, r.t_us, r.gflops, r.balg_gbs,
                         r.info.device_bytes * 1e-6, input_vo.gpuX_bytes * 1e-6, input_vo.gpuC_bytes * 1e-6,
                         perfRes.cuSpmmProcessing, r.errs);
        std::fclose(csv);
    }
    const double flops = 2.0 * input_vo.nnz * input_vo.dim;
    if (o.vendor)
        std::printf("hipSPARSE setup/us: %.2f , processing/us: %.2f  (%.1f GFLOP/s)\n", perfRes.cuSpmmSetup,
                    perfRes.cuSpmmProcessing, flops / perfRes.cuSpmmProcessing * 1e-3);
    // B-Re1 / B-Re2 as in the reference's table (flex.cu:5217-5223): nnz per distinct B row inside one
    // unit of work (here a chunk = one wave) and inside one cache domain (here an XCD's L2)
    // Imb = per-CU busy imbalance measured by wave clocks (≙ the reference's per-SM "Imb", flex.cu:5087-5126);
    // tPre/tE = planning time over one execution (≙ the README's tPre/tElap column, README.md:34-42)
    std::printf("%-4s %-8s %10s %10s %10s %8s %8s %8s %7s %7s %6s %8s %8s %9s\n", "Ord", "sched", "t/us", "GFLOP/s", "Balg GB/s",
                "%8TB/s", "chunks", "split", "B-Re1", "B-Re2", "Imb%", "plan/ms", "tPre/tE", "errs");
    for (const Row &r : rows) {
        std::printf("%-4s %-8s %10.1f %10.1f %10.1f %8.2f %8lld %8lld %7.2f %7.2f %6.1f %8.1f %8.1f %9d\n", r.ord.c_str(),
                    r.sched.c_str(), r.t_us, r.gflops, r.balg_gbs, r.balg_gbs / 8000.0 * 100,
                    static_cast<long long>(r.info.n_chunks), static_cast<long long>(r.info.n_split_rows),
                    r.stats.reuse_wave, r.stats.reuse_xcd, r.imb.cu_busy_imb_pct, r.plan_ms, r.plan_ms * 1e3 / r.t_us, r.errs);
        if (r.hbm_bytes >= 0) {  // ≙ the reference's DRAM GB/s, %Pk and measured u per table row (flex.cu:5237)
            char u_txt[32] = "n/a (B stayed in the L2s)";  // nothing was fetched beyond A: there is no reuse to divide by
            if (r.u_meas >= 0) std::snprintf(u_txt, sizeof u_txt, "%.2f", r.u_meas);
            std::printf("     counters: HBM-side %.1f MB/launch = %.2fx Balg, %.0f GB/s (%.1f %% of 8 TB/s), L2 hit %.3f, u %s\n",
                        r.hbm_bytes * 1e-6, r.hbm_bytes / (r.balg_gbs * r.t_us * 1e3), r.hbm_bytes / r.t_us * 1e-3,
                        r.hbm_bytes / r.t_us * 1e-3 / 8000.0 * 100, r.l2_hit, u_txt);
        }
        if (r.l2_bytes >= 0)  // ≙ the L1<->L2 bytes and "Per Mult / Num Insns" columns (flex.cu:5279-5330, 5350-5420)
            std::printf("               L1<->L2 %.1f MB/launch (%.0f GB/s), u at L1 %.2f; wave insns per 64 FMAs: vmem_rd %.2f valu %.2f lds %.2f salu %.2f\n",
                        r.l2_bytes * 1e-6, r.l2_bytes / r.t_us * 1e-3, r.u_l1, r.vmem_rd, r.valu, r.lds, r.salu);
        if (o.json)
            std::printf("{\"graph\":\"%s\",\"n\":%zu,\"nnz\":%zu,\"k\":%zu,\"ord\":\"%s\",\"schedule\":\"%s\",\"t_us\":%.3f,"
                        "\"gflops\":%.2f,\"balg_gbs\":%.2f,\"max_err\":%.3g,\"errs\":%d,\"vendor_us\":%.3f,\"b_re1\":%.3f,\"b_re2\":%.3f,"
                        "\"chunk_imb_pct\":%.1f,\"xcd_imb_pct\":%.2f,\"cu_imb_pct\":%.2f,\"cu_end_spread_pct\":%.2f,\"xcd_busy_imb_pct\":%.2f,"
                        "\"cus_seen\":%d,\"plan_ms\":%.2f,\"tpre_over_telap\":%.1f,\"mfma_tiles\":%lld,\"tile_nnz_pct_25\":%.2f,"
                        "\"hbm_bytes\":%.0f,\"l2_hit\":%.4f,\"u_measured\":%.3f,\"l1_l2_bytes\":%.0f,\"u_l1\":%.3f,\"vmem_rd_per_64fma\":%.3f,"
                        "\"valu_per_64fma\":%.3f,\"lds_per_64fma\":%.3f}\n",
                        input_vo.graph_name.c_str(), input_vo.n, input_vo.nnz, input_vo.dim, r.ord.c_str(),
                        r.sched.c_str(), r.t_us, r.gflops, r.balg_gbs, r.max_err, r.errs, perfRes.cuSpmmProcessing,
                        r.stats.reuse_wave, r.stats.reuse_xcd, r.stats.chunk_imb_pct, r.stats.xcd_imb_pct, r.imb.cu_busy_imb_pct,
                        r.imb.cu_end_spread_pct, r.imb.xcd_busy_imb_pct, r.imb.cus_seen, r.plan_ms, r.plan_ms * 1e3 / r.t_us,
                        static_cast<long long>(r.stats.mfma_tiles), r.stats.tile_nnz_pct_25, r.hbm_bytes, r.l2_hit, r.u_meas, r.l2_bytes, r.u_l1,
                        r.vmem_rd, r.valu, r.lds);
    }
    int mg_errs = 0;
    if (o.gpus > 0) {  // row-sharded over several GPUs (new; the reference is single-GPU, flex.cu:4137)
        flex_mg *mg = nullptr;
        const flex_csr a = input_vo.csr_view();
        // a failure here (no such device, RCCL cannot bring the communicator up) must end the run with a non-zero exit
        // and say which layer failed: FLEX_CHECK throws, main() turns that into exit code 1; the RCCL result is printed first
        auto mg_check = [&](int st, const char *what) {
            if (st == FLEX_OK) return;
            std::printf("flex --gpus %d: %s failed: %s (rccl result %d, hip error %d: %s)\n", o.gpus, what, flex_strerror(st), FLEX_MG(flex_mg_last_rccl)(),
                        flex_last_hip_error(), flex_last_hip_error_string());
            std::fflush(stdout);
            FLEX_CHECK(st);
        };
        mg_check(FLEX_MG(flex_mg_create)(&mg, &a, static_cast<int>(input_vo.dim), o.gpus, nullptr, FLEX_ORDER_CLUSTER), "flex_mg_create");
        double bcast_ms = 0, us = 0;
        mg_check(FLEX_MG(flex_mg_set_B)(mg, input_vo.cpuX.data(), &bcast_ms), "flex_mg_set_B (RCCL broadcast)");
        FLEX_CHECK(FLEX_MG(flex_mg_time)(mg, o.warmup, o.iters, &us));
        FLEX_CHECK(FLEX_MG(flex_mg_get_C)(mg, h_res.get()));
        std::vector<int64_t> bounds(o.gpus + 1), snnz(o.gpus);
        FLEX_MG(flex_mg_shard_info)(mg, bounds.data(), snnz.data());
        FLEX_MG(flex_mg_destroy)(mg);
        double max_err = 0;
        if (!input_vo.h_ref_c.empty()) {
            Mat probe(input_vo, 0, 0);
            mg_errs = resCheck(input_vo.h_ref_c.data(), h_res.get(), probe, perfRes, &max_err);
        }
        std::printf("MG x%d cluster  t/us %.1f  GFLOP/s %.1f  B bcast/ms %.3f  errs %d  shard nnz:", o.gpus, us,
                    flops / us * 1e-3, bcast_ms, mg_errs);
        for (int i = 0; i < o.gpus; ++i) std::printf(" %lld", static_cast<long long>(snnz[i]));
        std::printf("\n");
        if (o.json)
            std::printf("{\"graph\":\"%s\",\"k\":%zu,\"ord\":\"MG\",\"schedule\":\"cluster\",\"gpus\":%d,\"t_us\":%.3f,\"gflops\":%.2f,"
                        "\"b_bcast_ms\":%.3f,\"errs\":%d}\n",
                        input_vo.graph_name.c_str(), input_vo.dim, o.gpus, us, flops / us * 1e-3, bcast_ms, mg_errs);
    }
    for (const Row &r : rows)
        if (r.errs > 0) throw std::runtime_error("resCheck failed");  // ≙ assert(!count), flex.cu:4205
    if (mg_errs > 0) throw std::runtime_error("resCheck failed (multi-GPU)");
}
