/*
 * flex_counters.h -- C ABI of libflex_counters.so: the memory-system counters of the card read INSIDE the
 * run, around launches the caller chooses.
 *
 * ≙ the NPerf_* calls of run() (flex.cu:4583-4656: NPerf_init, NPerf_metric_collect of the l1tex / lts /
 * dram metrics; flex.cu:5237 prints L1<->L2 GB/s, DRAM GB/s %Pk and the measured B reuse `u` per table
 * row).  The reference reads CUPTI through its NPerf wrapper; here the counters come from the
 * rocprofiler-sdk device counting service: card-wide hardware counters started and read by the process
 * itself, no per-dispatch interception and no serialisation of the launches in between.
 *
 * Kept in its own library: the engine (libflex_spmm.so) never depends on rocprofiler-sdk, and a process
 * that does not call flex_counters_init never loads the profiler.
 *
 * Protocol: flex_counters_init() FIRST, before the process makes any HIP call: the profiler comes up together
 * with the ROCm runtime (it finds this library's rocprofiler_configure in the process, which answers only after
 * init was called -- load the library with RTLD_GLOBAL, or link it), afterwards it is too late and begin says
 * so.  Then, once the runtime is up (any HIP call), any number of times: begin(names) -> the launches ->
 * synchronise -> end(values).  A pass holds what the hardware can count at
 * once (MI355X_MICROARCH "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass); begin
 * refuses a set that does not fit.  Values are the sums over every instance of the counter (all XCDs,
 * all channels) of what the card did between begin and end -- the whole card, not one kernel: keep it
 * otherwise idle.  Units are the counter's own (FETCH_SIZE / WRITE_SIZE: KiB; the gfx950 correction of
 * MI355X_MICROARCH "HBM" -- FETCH_SIZE counts a 128-byte request as 64 -- is the CALLER's to apply, as
 * tools/pmc_summary.py does for the rocprofv3 passes).
 */
#ifndef FLEX_COUNTERS_H
#define FLEX_COUNTERS_H
#ifdef __cplusplus
extern "C" {
#endif

#define FLEX_COUNTERS_OK 0
#define FLEX_COUNTERS_ERR_LATE -1    /* the profiler was already up without this tool when flex_counters_init was called */
#define FLEX_COUNTERS_ERR_PROFILER -2 /* rocprofiler-sdk refused (flex_counters_error() has its text) */
#define FLEX_COUNTERS_ERR_NAME -3    /* a counter of that name does not exist on this card */
#define FLEX_COUNTERS_ERR_STATE -4   /* begin inside a pass, end outside one, bad arguments; begin while the profiler is not up
                                        (no init, init after the first HIP call, or no HIP call yet) */

/* Asks for the profiler: when the runtime initialises, one counting context per GPU is prepared.  Idempotent. */
int flex_counters_init(void);
/* number of GPUs the profiler lists (0 until the runtime has initialised) */
int flex_counters_devices(void);
/* Starts a pass on GPU `device` (ordinal among the GPUs the profiler lists, the order of rocminfo) counting
 * the `n` named counters (basic or derived names of `rocprofv3 -L`). */
int flex_counters_begin(int device, const char *const *names, int n);
/* Reads the pass started by begin and stops it: values[i] = sum over all instances of names[i]. */
int flex_counters_end(double *values);
/* text of the last failure in this thread's last call (never NULL) */
const char *flex_counters_error(void);

#ifdef __cplusplus
}
#endif
#endif
