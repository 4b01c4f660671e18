// spmm_kernels.hip -- gfx950 (CDNA4, wave64) SpMM kernels of libflex_spmm.so.
//
// Replaces the reference's kernel family flex.cu:80-4124 (selected kernel
// alpha_w_atomic_spmm_v36, flex.cu:4008-4124; closest in structure: spmm_test2,
// flex.cu:190-273).  None of that code is reused.  Design:
//
//  * A *task* is one C row (or one piece of a long row); a *chunk* is a short
//    contiguous run of tasks cut by the host planner so that every chunk holds
//    about the same number of (col,val) records.
//  * One wave per chunk; workgroup ids are remapped so that each XCD (private
//    4 MiB L2) walks its own contiguous eighth of the schedule with one cursor:
//    rows that are neighbours in the schedule share B rows through that L2, and
//    the hardware dispatcher (a new workgroup whenever one retires) does the
//    load balancing that the reference builds by hand with per-SM buckets plus
//    a balance bucket (flex.cu:2447-2518, 4008-4124).
//  * k/4 lanes (G) cooperate on one record, each lane owning 4 consecutive
//    columns of C, so one wave-instruction gathers 64/G different B rows with
//    16-byte loads (k=128: two 512-B rows = 1 KiB per global_load_dwordx4;
//    k=32: eight 128-B rows).  The S = 64/G slots of a wave walk the SAME row
//    and are combined by a log2(S)-step xor-shuffle at row end -- or, for SHORT rows on
//    the tiles of four or more slots, S DIFFERENT rows of one bundle (a task that holds
//    a row per slot): nothing to combine, one 16-byte store per lane at the bundle's end
//    (plan_build.cpp form_tasks; ≙ the reference's narrow kernel, flex.cu:81-118).
//  * Rows of GNN graphs are short (median degree < 10), so the wave treats ALL
//    records of a chunk as one stream of steps (S records per step; the planner
//    pads every row to a multiple of S with records that share its last value): the records are
//    fetched once, coalesced, into a wave-private LDS slice, then blocks of U
//    gathers are issued back to back regardless of row boundaries, and a
//    wave-uniform scalar check after each step flushes the accumulator when a row
//    ends.  No barriers: LDS slices are private to a wave.
//  * C is written once with non-temporal 16-byte stores; rows cut into several
//    pieces go to k-wide partial slots instead and are summed in piece order -- by
//    spmm_fixup_kernel after this launch (large launches: it needs nothing beyond
//    stream order) or by whichever piece finishes last inside the same launch (small
//    ones, where a second kernel boundary would show; flex_plan_tuning.split_rows forces).  Deterministic either way;
//    the reference uses atomicAdd for its split rows, mat.cu:816-824.
#include "internal.h"

namespace flex {
namespace {

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

template <bool OFF32>
__device__ __forceinline__ float4 gather4(const char *__restrict__ Bb, uint32_t recx, uint32_t lane_off,
                                          uint64_t row_bytes) {
    if constexpr (OFF32) {
        // base (SGPR pair) + 32-bit VGPR offset: global_load_dwordx4 v, v_off, s[base]
        return *reinterpret_cast<const float4 *>(Bb + static_cast<uint32_t>(recx + lane_off));
    } else {
        return *reinterpret_cast<const float4 *>(Bb + (static_cast<uint64_t>(recx) * row_bytes + lane_off));
    }
}

__device__ __forceinline__ void fma4(float4 &acc, float v, const float4 &b) {
    acc.x = fmaf(v, b.x, acc.x);
    acc.y = fmaf(v, b.y, acc.y);
    acc.z = fmaf(v, b.z, acc.z);
    acc.w = fmaf(v, b.w, acc.w);
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// Row epilogue.  The S = 64/G slots of a wave each hold a partial float4 of the same C row.  They are
// combined by a reduce-SCATTER, not an all-reduce: at every level a lane adds its partner's half of the
// values and hands the other half over, so the value count halves with the lane distance and the row ends
// up spread over the lanes -- one cross-lane instruction and one add per PAIR of values, no copies:
//   pair8  (a,b): lanes with bit 3 clear get a[l]+a[l^8],  the others b[l]+b[l^8]   (two masked DPP adds)
//   pair16 (a,b): even 16-lane rows get a[l]+a[l^16], odd rows b[l]+b[l^16]         (v_permlane16_swap + add)
//   pair32 (a,b): the lower wave half gets a[l]+a[l^32], the upper half b[l]+b[l^32] (v_permlane32_swap + add)
__device__ __forceinline__ float pair8(float a, float b) {
    float r;
    // row_ror:8 = the lane 8 over in the same 16-lane row; bank_mask picks lanes 0-7 / 8-15 of every row.
    // s_nop: a DPP read needs two wait states after a VALU write of its source, and the compiler's hazard
    // recognizer does not look inside inline asm.
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

// pair4 (a,b): lanes with bit 2 clear get a[l]+a[l^4], the others b[l]+b[l^4].  l^4 swaps quads 0<->1 and 2<->3 of a 16-lane row:
// quads 0 and 2 (bank_mask 0x5) take their partner from the quad ABOVE (row_ror:12 = rotate left by 4), quads 1 and 3
// (bank_mask 0xa) from the quad below (row_ror:4).
__device__ __forceinline__ float pair4(float a, float b) {
    float r;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float pair16(float a, float b) {
    // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second:
    // (a,b) -> {a0,b0,a2,b2}, {a1,b1,a3,b3}; their sum is a0+a1 on row 0, b0+b1 on row 1, ...
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

__device__ __forceinline__ float pair32(float a, float b) {
    // v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second:
    // (a,b) -> {a.lo,b.lo}, {a.hi,b.hi}; their sum is a.lo+a.hi on the lower half, b.lo+b.hi on the upper
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// What a lane holds of a finished row, and which column(s) of its tile.  Measured on MI355X (tools/ab.py):
// the scattered form saves ~30 VALU instructions per row at G=8, where short rows make the epilogue the
// largest VALU term (flickr k=32: 17.3 -> 14.8 us); at G=16/32 it only trades 16-byte stores from the 16/32
// lanes of one slot for 4/8-byte stores strided across the wave (flickr k=64 +3.5 %, k=128 +-0), so those
// widths keep the all-reduce and the contiguous float4 store:
//   G=64: 4 values at c0, every lane          (no reduction)
//   G=32: 4 values at c0, lanes of slot 0     (all-reduce over 2 slots)
//   G=16: 4 values at c0, lanes of slot 0     (all-reduce over 4 slots)
//   G=8 : 1 value  at c0 + 2*bit3 + bit4, lower wave half  (pair8 twice, pair16, then both halves summed)
//   G=4 : 1 value  at c0 + 2*bit2 + bit3, lanes 0-15       (pair4 twice, pair8, then the four 16-lane rows summed)   [k <= 16]
template <int G>
struct RowOut {
    static constexpr int kVals = G <= 8 ? 1 : 4;
    float v[kVals];
};

template <int G>
__device__ __forceinline__ int row_out_col(int lane, int c0) {
    if constexpr (G == 8) return c0 + 2 * ((lane >> 3) & 1) + ((lane >> 4) & 1);
    else if constexpr (G == 4) return c0 + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
    else return c0;
}

template <int G>
__device__ __forceinline__ bool row_out_lane(int lane) {  // does this lane store
    return G == 8 ? lane < 32 : G == 4 ? lane < 16 : lane < G;
}

template <int G>
__device__ __forceinline__ RowOut<G> reduce_row(const float4 &acc) {
    RowOut<G> o;
    if constexpr (G == 8) {
        const float t = pair16(pair8(acc.x, acc.z), pair8(acc.y, acc.w));
        o.v[0] = pair32(t, t);  // both halves hold the same columns: plain sum, the lower half stores
    } else if constexpr (G == 4) {
        const float t = pair8(pair4(acc.x, acc.z), pair4(acc.y, acc.w));
        const float h = pair16(t, t);  // the four 16-lane rows hold the same columns: plain sums, the first row stores
        o.v[0] = pair32(h, h);
    } else {
        float4 r = acc;
        if constexpr (G <= 16) {
            r.x = pair16(r.x, r.x);
            r.y = pair16(r.y, r.y);
            r.z = pair16(r.z, r.z);
            r.w = pair16(r.w, r.w);
        }
        if constexpr (G <= 32) {
            r.x = pair32(r.x, r.x);
            r.y = pair32(r.y, r.y);
            r.z = pair32(r.z, r.z);
            r.w = pair32(r.w, r.w);
        }
        o.v[0] = r.x; o.v[1] = r.y; o.v[2] = r.z; o.v[3] = r.w;
    }
    return o;
}

// non-temporal store of a lane's share of a C row (ptr already points at its first column)
template <int G>
__device__ __forceinline__ void store_row_out(float *ptr, const RowOut<G> &o) {
    if constexpr (RowOut<G>::kVals == 4) {
        const v4f val = {o.v[0], o.v[1], o.v[2], o.v[3]};
        __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(ptr));
    } else {
        __builtin_nontemporal_store(o.v[0], ptr);
    }
}

// All-reduce form for every G: afterwards EVERY lane holds the sum over the S slots of its 4 columns, so the lanes of
// slot 0 can write the row as contiguous float4s.  Used for pieces (partial sums of rows that other chunks hold pieces
// of): their stores are write-through (sc1), and a 4-byte sc1 store costs ~6x a 16-byte one per byte
// (MI355X_MICROARCH.md, stores of each flavour), so the scattered G=8 form of RowOut is not used there.
template <int G>
__device__ __forceinline__ float4 reduce_full(const float4 &acc) {
    float4 r = acc;
    if constexpr (G <= 4) {  // lane ^ 4: quads 0 and 2 read the quad above (row_ror:12), quads 1 and 3 the quad below (row_ror:4)
        auto partner = [](float x) {
            const int lo = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x12C, 0xf, 0x5, false);
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(lo, __builtin_bit_cast(int, x), 0x124, 0xf, 0xa, false));
        };
        r.x += partner(r.x);
        r.y += partner(r.y);
        r.z += partner(r.z);
        r.w += partner(r.w);
    }
    if constexpr (G <= 8) {  // lane ^ 8 inside each 16-lane row: row_ror:8
        r.x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r.x), 0x128, 0xf, 0xf, false));
        r.y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r.y), 0x128, 0xf, 0xf, false));
        r.z += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r.z), 0x128, 0xf, 0xf, false));
        r.w += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r.w), 0x128, 0xf, 0xf, false));
    }
    if constexpr (G <= 16) {
        r.x = pair16(r.x, r.x);
        r.y = pair16(r.y, r.y);
        r.z = pair16(r.z, r.z);
        r.w = pair16(r.w, r.w);
    }
    if constexpr (G <= 32) {
        r.x = pair32(r.x, r.x);
        r.y = pair32(r.y, r.y);
        r.z = pair32(r.z, r.z);
        r.w = pair32(r.w, r.w);
    }
    return r;
}

// 16-byte agent-scope (sc1) load through a 64-bit address: served by the L2, never by this CU's L1.  The compiler does
// not count loads issued from inline asm, so the caller waits with wait_loads() before touching the values.
__device__ __forceinline__ v4f load_b128_sc1(const float *ptr) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void wait_loads(v4f (&v)[N]) {
    static_assert(N == 8, "one batch of the piece sum");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 :
                 : "memory");
}

// records staged per wave and window: 256 (2 KiB of LDS) on the wide tiles, 512 on the G = 8 tile, whose chunk budget goes up to
// 512 records (plan_build.cpp, read_knobs) -- one window per chunk there; measured with the budget
template <int G>
constexpr int kWindowRecs = G <= 8 ? 512 : 256;
// Row bundles exist on the tiles with at least kBundleMinSlots slots per step (internal.h): the wide tiles have one or two slots --
// little to gain -- and no register to spare for a chunk's bundle rows (72 VGPRs for seven waves per SIMD, see spmm_flat_kernel).
template <int G>
constexpr bool kTileHasBundles = 64 / G >= static_cast<int>(kBundleMinSlots);

[[maybe_unused]] __device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & (kXcds - 1);
}

// One chunk = tasks [w_task[c], w_task[c+1]) = one contiguous run of the record stream.
#ifdef FLEX_TRACE
#define FLEX_STAMP(i)                                                       \
    do {                                                                    \
        uint64_t now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                  \
        phase[i] += now_ - last_;                                           \
        last_ = now_;                                                       \
    } while (0)
#else
#define FLEX_STAMP(i) do {} while (0)
#endif

// Stage records [wz, wz+wn) of the stream into the wave's LDS slice: coalesced 512-B loads, lane l takes
// records l, l+64, l+128, l+192.  Indices are clamped, not predicated (slots >= wn get a copy of the last
// record and are never read), and the short-window case is a wave-uniform branch: every load is consumed
// inside its branch, so (a) a long window has its four loads in flight together instead of four round
// trips and (b) no pending load survives into the gather loop, where the compiler would otherwise put an
// s_waitcnt vmcnt(0) at the loop head.
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint2 load_rec(const uint2 *ptr) {
    if constexpr (NT) {  // records are read once per column tile: keep them out of the way of the B rows in L2 / Infinity Cache
        const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(ptr));
        return make_uint2(v.x, v.y);
    } else {
        return *ptr;
    }
}

template <bool NT, int LOADS>
__device__ __forceinline__ void stage_n(uint2 *my_lds, const uint2 *__restrict__ rec, uint32_t wz, uint32_t wn, int lane) {
    uint2 r[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) r[i] = load_rec<NT>(rec + wz + min(static_cast<uint32_t>(i * 64 + lane), wn - 1));
#pragma unroll
    for (int i = 0; i < LOADS; ++i) my_lds[i * 64 + lane] = r[i];
}

template <bool NT, int G>
__device__ __forceinline__ void stage_window_t(uint2 *my_lds, const uint2 *__restrict__ rec, uint32_t wz, uint32_t wn, int lane) {
    if (wn <= 64) {
        stage_n<NT, 1>(my_lds, rec, wz, wn, lane);
    } else if constexpr (kWindowRecs<G> > 256) {
        if (wn <= 256) stage_n<NT, 4>(my_lds, rec, wz, wn, lane);
        else stage_n<NT, kWindowRecs<G> / 64>(my_lds, rec, wz, wn, lane);
    } else {
        stage_n<NT, 4>(my_lds, rec, wz, wn, lane);
    }
}

template <int G>
__device__ __forceinline__ void stage_window(uint2 *my_lds, const uint2 *__restrict__ rec, uint32_t wz, uint32_t wn, int lane, bool nt) {
    if (nt) stage_window_t<true, G>(my_lds, rec, wz, wn, lane);  // wave-uniform
    else stage_window_t<false, G>(my_lds, rec, wz, wn, lane);
}

// All the work of one chunk once its header {first task, #tasks, first record, end record} and its
// task descriptors (lane i: t_beg[t0+i], t_dst[t0+i]) are in registers.
template <int G, bool OFF32, int U>
__device__ __forceinline__ void compute_chunk(const PlanView &p, uint4 hdr, uint32_t my_beg, uint32_t my_dst, uint2 my_aux,
                                              uint32_t my_bd0, uint32_t my_bd1, uint2 *my_lds, const char *__restrict__ Bb,
                                              float *__restrict__ C, int lane, int c0, bool col_ok, uint32_t tile, uint32_t ktiles
#ifdef FLEX_TRACE
                                              , uint64_t *phase, uint64_t &last_
#endif
) {
    constexpr int S = 64 / G;
    const int slot = lane / G;
    const int k = p.k;
    const uint64_t ldc = static_cast<uint64_t>(p.ldc);
    // lanes past column k gather the tile's FIRST column instead: a line this record's gather touches anyway
    // (column 0 of the row would add a cache line per record to the last tile of a k that is not a multiple of 4G)
    const uint32_t lane_off = (col_ok ? c0 : c0 - (lane % G) * 4) * 4u;
    const uint64_t row_bytes = static_cast<uint64_t>(p.ldb) * 4u;
    const uint2 *__restrict__ rec = p.rec;

    const uint32_t nt = hdr.y, zb = hdr.z, ze = hdr.w;
    FLEX_STAMP(0);  // descriptors

    uint32_t ti = 0;                                          // current task
    uint32_t row_end = __builtin_amdgcn_readlane(my_beg, 1);  // where it ends in the record stream
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    const int out_col = row_out_col<G>(lane, c0);              // first column this lane stores of a finished row
    const bool out_ok = col_ok && row_out_lane<G>(lane);
    // Write out the task that ends at the current stream position (and any empty rows behind it).
    // row_end is kept at ~0 once the chunk's tasks are exhausted, so the per-step test in the hot
    // loop is ONE scalar compare.  A task whose destination carries kPartialFlag is a PIECE: one of several
    // partial sums of a C row (a row longer than one budget, or a row cut by column panel, plan_build.cpp); it goes to
    // its k-wide slot of `partial`, write-through when the pieces are combined inside this launch.
    auto flush = [&](uint32_t pos) {
        do {
            const uint32_t dst = __builtin_amdgcn_readlane(my_dst, ti);
            if ((dst & (kPartialFlag | kBundleFlag)) == (kPartialFlag | kBundleFlag)) {  // wave-uniform
                // a BUNDLE: slot s held row s of it all along -- nothing to reduce; every lane stores the 4 columns it owns of its
                // slot's row (G lanes x 16 bytes: the row's whole column tile).  The chunk's rows came in with the descriptors
                // (lane i: entries i and 64 + i of its part of bd_rows); the slot's lanes fetch theirs with one cross-lane read.
                if constexpr (kTileHasBundles<G>) {
                    const uint32_t first = dst & (kBundleRowsPerChunk - 1u);
                    const uint32_t held = first < 64u ? my_bd0 : my_bd1;
                    const uint32_t row = static_cast<uint32_t>(
                        __builtin_amdgcn_ds_bpermute(static_cast<int>(((first & 63u) + slot) << 2), static_cast<int>(held)));
                    if (col_ok && row != kBundleNoRow) {
                        const bool none = (row & kBundleZero) != 0;  // a row without nonzeros: zeros, whatever its slot summed
                        const v4f val = {none ? 0.f : acc.x, none ? 0.f : acc.y, none ? 0.f : acc.z, none ? 0.f : acc.w};
                        __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(C + static_cast<uint64_t>(row & ~kBundleZero) * ldc + c0));
                    }
                }
            } else if (__builtin_expect((dst & kPartialFlag) != 0, 0)) {  // wave-uniform; the rare case on 1-D plans
                const float4 r = reduce_full<G>(acc);
                float *prow = p.partial + static_cast<uint64_t>(dst & ~kPartialFlag) * k;  // uniform
                if (slot == 0 && col_ok) {
                    if (p.fused_fixup) {
                        const auto prsrc = __builtin_amdgcn_make_buffer_rsrc(prow, 0, k * 4, 0x00020000);
                        const v4u pv = {__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(pv, prsrc, c0 * 4, 0, 16 /* sc1 */);
                    } else {  // combined by spmm_fixup_kernel after this launch
                        *reinterpret_cast<float4 *>(prow + c0) = r;
                    }
                }
            } else {
                const RowOut<G> o = reduce_row<G>(acc);
#ifdef FLEX_ABL_NOSTORE  // timing-only ablation: the store is kept in the code but never executes
                if (out_ok && p.k < 0)
#else
                if (out_ok)
#endif
                    store_row_out<G>(C + static_cast<uint64_t>(dst) * ldc + out_col, o);
            }
            acc = {0.f, 0.f, 0.f, 0.f};
            ++ti;
            row_end = ti < nt ? __builtin_amdgcn_readlane(my_beg, ti + 1) : 0xFFFFFFFFu;
        } while (row_end == pos);
    };
    if (nt == 0) row_end = 0xFFFFFFFFu;
    if (row_end == zb) flush(zb);  // leading empty rows

    uint32_t pos = zb;  // stream position after the steps consumed so far
    for (uint32_t wz = zb; wz < ze; wz += kWindowRecs<G>) {
        const uint32_t wn = min(static_cast<uint32_t>(kWindowRecs<G>), ze - wz);
        // stage this window's records: coalesced 512-B loads, one ds_write_b64 per lane and load
        stage_window<G>(my_lds, rec, wz, wn, lane, p.rec_nt != 0);
        FLEX_STAMP(1);  // records -> LDS
#ifdef FLEX_ABL_STAGEONLY  // timing-only ablation: header, descriptors and records fetched, then leave
        if (p.k > 0) {
            if (lane == 0 && my_lds[wn - 1].x == 0xFFFFFFFEu) C[0] = as_f32(my_beg + my_dst);
            return;
        }
#endif
        const uint32_t nsteps = wn / S;  // rows are padded to multiples of S
        const uint2 *lds_slot = my_lds + slot;
        uint32_t j = 0;
        for (; j + U <= nsteps; j += U) {  // full blocks: no bound checks in the instruction stream
            uint2 r[U];
            float4 b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = lds_slot[(j + u) * S];
#ifdef FLEX_ABL_NOGATHER  // timing-only ablation: no B traffic at all, values faked from the record
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = make_float4(as_f32(r[u].x), as_f32(r[u].y), 1.f, 2.f);
#else
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = gather4<OFF32>(Bb, r[u].x, lane_off, row_bytes);
#endif
            FLEX_STAMP(2);  // gathers
#pragma unroll
            for (int u = 0; u < U; ++u) {
                fma4(acc, as_f32(r[u].y), b[u]);
                pos += S;
#ifndef FLEX_ABL_NOFLUSH  // timing-only ablation: rows are never written out (one flush at the very end)
                if (pos == row_end) flush(pos);
#endif
            }
            FLEX_STAMP(3);  // fma + row flushes
        }
        if (j < nsteps) {  // the window's last, partial block
            uint2 r[U];
            float4 b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = lds_slot[min(j + u, nsteps - 1) * S];
#ifdef FLEX_ABL_NOGATHER
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = make_float4(as_f32(r[u].x), as_f32(r[u].y), 1.f, 2.f);
#else
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = gather4<OFF32>(Bb, r[u].x, lane_off, row_bytes);
#endif
            FLEX_STAMP(2);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < nsteps) {  // wave-uniform
                    fma4(acc, as_f32(r[u].y), b[u]);
                    pos += S;
#ifndef FLEX_ABL_NOFLUSH
                    if (pos == row_end) flush(pos);
#endif
                }
            }
            FLEX_STAMP(3);
        }
    }
#ifdef FLEX_ABL_NOFLUSH
    // pos != the ~0 sentinel, or the do-while never ends
    if (nt > 0) { ti = nt - 1; flush(0xFFFFFFFEu); }
#endif
    // Pieces are combined INSIDE this launch by whichever piece arrives last (cdna guide G16, counter form with
    // write-through payload): (1) every partial sum of this chunk was stored write-through (sc1), so it is at
    // device scope once the store completes; (2) the wave drains its stores; (3) lane i bumps the arrival counter of
    // task i's row (agent-scope atomic; t_aux = {row's index in `split`, #pieces}); (4) a lane whose add returns
    // count-1 knows every piece of that row is visible: the wave reads them with sc1 loads (never through a CU's
    // L1) and adds them in PIECE order -- the sum is reproducible although the reducer is not -- S rows at a time,
    // one per slot, and re-arms the counters for the next launch.  OPT-IN since ABI 3 (p.fused_fixup): by default the
    // pieces are left to spmm_fixup_kernel, which measured faster on the large shapes and needs no such argument.
    // Why RELAXED + sc1 instead of an acq_rel atomic: a release at agent scope is `buffer_wbl2 sc1`, a
    // write-back of the whole XCD L2 (1.7-6.5 us, MI355X_MICROARCH.md "Workgroup dispatch ... visibility"),
    // paid by every chunk.  The form used here is that guide's hand-off "each storing wave for itself:
    // sc1 stores of whole 16-B granules -> the wave's own s_waitcnt vmcnt(0) -> agent-scope atomic add;
    // the wave whose add returned last reads every byte with ... sc1 loads" (its measured table of sc1
    // hand-offs on gfx950 / ROCm 7.2; measured, not an architectural guarantee, hence the 300-launch
    // test under uneven load, tests/test_gpu_spmm.py).  The workspace makes a plan non-reentrant: one
    // launch of a plan at a time (include/flex_spmm.h, flex_spmm).
    const bool mine_partial = static_cast<uint32_t>(lane) < nt && (my_dst & (kPartialFlag | kBundleFlag)) == kPartialFlag;
    if (p.fused_fixup && __builtin_amdgcn_ballot_w64(mine_partial) != 0) {  // wave-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t arrived = 0;
        if (mine_partial)
            arrived = __hip_atomic_fetch_add(p.split_cnt + static_cast<uint64_t>(my_aux.x) * ktiles + tile, 1u,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t done = __builtin_amdgcn_ballot_w64(mine_partial && arrived + 1 == my_aux.y);
        while (done != 0) {  // wave-uniform: up to S completed rows per round, slot s takes the s-th
            int take = -1;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                if (done != 0) {
                    const int i = __builtin_ctzll(done);
                    done &= done - 1;
                    if (slot == s2) take = i;
                }
            }
            const uint32_t sidx = __shfl(my_aux.x, take < 0 ? 0 : take);
            if (take >= 0) {
                const SplitRow sr = p.split[sidx];
                if (col_ok) {
                    const float *base = p.partial + static_cast<uint64_t>(sr.first) * k + c0;
                    float4 s4 = {0.f, 0.f, 0.f, 0.f};
                    for (uint32_t j = 0; j < sr.count; j += 8) {
                        v4f v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = load_b128_sc1(base + static_cast<uint64_t>(min(j + u, sr.count - 1)) * k);
                        wait_loads(v);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (j + u < sr.count) {
                                s4.x += v[u].x;
                                s4.y += v[u].y;
                                s4.z += v[u].z;
                                s4.w += v[u].w;
                            }
                        }
                    }
                    const v4f val = {s4.x, s4.y, s4.z, s4.w};
                    __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(C + static_cast<uint64_t>(sr.row) * ldc + c0));
                }
                if (lane % G == 0)
                    __hip_atomic_store(p.split_cnt + static_cast<uint64_t>(sidx) * ktiles + tile, 0u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// G lanes per record, 4 columns per lane; covers k <= 4*G per blockIdx.y tile.
// One wave per chunk.  The hardware deals consecutive workgroup ids round-robin over the 8
// XCDs, so workgroup b is given chunk group (b % 8) * cpx + b / 8: XCD x walks the x-th
// contiguous eighth of the schedule with ONE cursor, and its resident waves form a sliding
// window over neighbouring rows whose shared B rows stay in that XCD's 4 MiB L2.  (A
// persistent variant with per-XCD ticket queues was measured and rejected: DESIGN.md 3.4.)
// STAMP = true is the measuring twin of the product kernel (flex_plan_measure_imbalance; ≙ the reference's per-warp
// %smid + clock() stamps, flex.cu:27-79): the same code plus, per wave, two reads of the 100 MHz constant clock and one
// 24-byte record {start, end, XCC id << 32 | HW_ID}.  The product launches (flex_spmm) never use it.
// amdgpu_waves_per_eu: the wide tiles (G >= 32, U = 8, 32-bit offsets) come out at 76 VGPRs = 6 waves per SIMD unless the
// allocator is told that a seventh wave is worth a few moves (72 VGPRs, no scratch: checked with `make asm`); the narrow
// tiles are at 8 waves per SIMD either way, and the variants that would spill under the hint (64-bit row addressing, the
// FLEX_U=8 experiment) are left alone.
template <int G, bool OFF32, int U, int WPB, bool STAMP = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(OFF32 && U == (G >= 32 ? 8 : 4) ? 7 : 4))) void spmm_flat_kernel(PlanView p, const float *__restrict__ B,
                                                             float *__restrict__ C) {
    __shared__ uint2 lds_rec[WPB][kWindowRecs<G>];
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Which (workgroup of the chunk table, column tile) this workgroup is.  Classic: blockIdx.y = tile, one pass over the whole
    // table per tile.  Grouped (p.tile_group): a 1-D grid in which every XCD slice is walked group by group, the tiles of a group
    // back to back -- a group's records are then re-read from the Infinity Cache instead of HBM.
    uint32_t tile = blockIdx.y, ktiles = gridDim.y, bid;
    if (p.tile_group == 0) {
        const uint32_t cpx = gridDim.x / kXcds;  // gridDim.x % 8 == 0
        bid = p.xcd_remap ? (blockIdx.x % kXcds) * cpx + (blockIdx.x / kXcds) : blockIdx.x;
    } else {
        ktiles = (p.k + 4 * G - 1) / (4 * G);
        const uint32_t nwg = gridDim.x / ktiles;                        // workgroups of one pass (a multiple of 8)
        const uint32_t slice = p.xcd_remap ? nwg / kXcds : nwg;         // ... of one XCD's slice
        const uint32_t l = p.xcd_remap ? blockIdx.x / kXcds : blockIdx.x;  // position in this XCD's extended slice [0, slice * ktiles)
        const uint32_t g = l / (p.tile_group * ktiles), r = l % (p.tile_group * ktiles);
        const uint32_t g0 = g * p.tile_group, gs = min(p.tile_group, slice - g0);  // the last group of a slice may be short
        tile = r / gs;
        const uint32_t in_slice = g0 + r % gs;
        if (tile >= ktiles) return;  // only past the end of a short last group
        bid = p.xcd_remap ? (blockIdx.x % kXcds) * slice + in_slice : in_slice;
    }
    const uint32_t chunk = bid * WPB + wib;
    if (chunk >= p.n_chunks) return;
#ifdef FLEX_ABL_EMPTY  // timing-only ablation: dispatch + one header load per wave, nothing else
    if (p.chunk[chunk].y != 0xFFFFFFFFu) return;
#endif
    const int c0 = tile * (4 * G) + (lane % G) * 4;  // first of this lane's 4 columns
    const bool col_ok = c0 < p.k;                            // k % 4 == 0 on this path
#ifdef FLEX_TRACE  // diagnostic build only (tools/trace.py)
    const uint64_t trace_t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t phase[5] = {0, 0, 0, 0, 0};
    uint64_t last_ = trace_t0;
    const uint4 hdr = p.chunk[chunk];
    const uint32_t my_beg = (static_cast<uint32_t>(lane) <= hdr.y) ? p.t_beg[hdr.x + lane] : 0u;
    const uint32_t my_dst = (static_cast<uint32_t>(lane) < hdr.y) ? p.t_dst[hdr.x + lane] : 0u;
    const uint2 my_aux = (static_cast<uint32_t>(lane) < hdr.y) ? p.t_aux[hdr.x + lane] : make_uint2(0u, 0u);
    uint32_t my_bd0 = kBundleNoRow, my_bd1 = kBundleNoRow;
    if constexpr (kTileHasBundles<G>) {
        if (p.bd_rows != nullptr) {
            const uint2 cb = p.chunk_bd[chunk];
            if (static_cast<uint32_t>(lane) < cb.y) my_bd0 = p.bd_rows[cb.x + lane];
            if (static_cast<uint32_t>(lane) + 64u < cb.y) my_bd1 = p.bd_rows[cb.x + 64u + lane];
        }
    }
    compute_chunk<G, OFF32, U>(p, hdr, my_beg, my_dst, my_aux, my_bd0, my_bd1, lds_rec[wib], reinterpret_cast<const char *>(B), C, lane, c0, col_ok, tile, ktiles, phase, last_);
    if (lane == 0 && p.trace != nullptr) {
        uint64_t *log = p.trace + static_cast<uint64_t>(chunk) * 12;
        log[0] = xcc_id();
        log[1] = trace_t0;
        log[2] = __builtin_amdgcn_s_memrealtime();
        log[3] = p.chunk[chunk].w - p.chunk[chunk].z;
        log[4] = 1;
        log[5] = blockIdx.x % kXcds;
        for (int i = 0; i < 5; ++i) log[6 + i] = phase[i];
        uint32_t hw_id;  // wave / SIMD / CU / SH / SE the wave ran on (≙ the reference's per-SM timing, flex.cu:27-79)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        log[11] = hw_id;
    }
#else
    // A chunk holds at most 63 tasks (planner invariant): all descriptors come with one coalesced
    // load per array and are handed out with v_readlane; the header carries the record range, so
    // the record fetch does not wait for them: header -> {descriptors, records} -> gathers.
    const uint4 hdr = p.chunk[chunk];
    // the chunk's part of the bundle rows: fetched beside the header, handed out at the end of each bundle (compute_chunk, flush)
    uint2 cb = make_uint2(0u, 0u);
    if constexpr (kTileHasBundles<G>)
        if (p.bd_rows != nullptr) cb = p.chunk_bd[chunk];  // uniform
    if (hdr.y == 0) return;  // an empty entry that pads this XCD's slice of the table (plan_build.cpp, build_chunk_table)
    const uint32_t my_beg = (static_cast<uint32_t>(lane) <= hdr.y) ? p.t_beg[hdr.x + lane] : 0u;
    const uint32_t my_dst = (static_cast<uint32_t>(lane) < hdr.y) ? p.t_dst[hdr.x + lane] : 0u;
    const uint2 my_aux = (static_cast<uint32_t>(lane) < hdr.y) ? p.t_aux[hdr.x + lane] : make_uint2(0u, 0u);  // read at chunk end only
    uint32_t my_bd0 = kBundleNoRow, my_bd1 = kBundleNoRow;
    if (kTileHasBundles<G> && cb.y != 0) {  // uniform
        if (static_cast<uint32_t>(lane) < cb.y) my_bd0 = p.bd_rows[cb.x + lane];
        if (static_cast<uint32_t>(lane) + 64u < cb.y) my_bd1 = p.bd_rows[cb.x + 64u + lane];
    }
    uint64_t stamp_t0 = 0;
    if constexpr (STAMP) stamp_t0 = __builtin_amdgcn_s_memrealtime();
    compute_chunk<G, OFF32, U>(p, hdr, my_beg, my_dst, my_aux, my_bd0, my_bd1, lds_rec[wib], reinterpret_cast<const char *>(B), C, lane, c0, col_ok, tile, ktiles);
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the wave's last stores have left
        const uint64_t stamp_t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && p.trace != nullptr) {
            uint32_t hw_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            uint64_t *log = p.trace + (static_cast<uint64_t>(tile) * p.n_chunks + chunk) * 3;
            log[0] = stamp_t0;
            log[1] = stamp_t1;
            log[2] = (static_cast<uint64_t>(xcc_id()) << 32) | hw_id;
        }
    }
#endif
}

// Any k (k % 4 != 0 or unaligned B/C): one wave per task, lane owns columns
// lane, lane+64, lane+128, lane+192 of the blockIdx.y-th 256-column tile.
template <bool OFF32>
__global__ __launch_bounds__(256) void spmm_generic_kernel(PlanView p, const float *__restrict__ B,
                                                           float *__restrict__ C, uint32_t slots_per_step) {
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cpx = gridDim.x / kXcds;
    const uint32_t bid = p.xcd_remap ? (blockIdx.x % kXcds) * cpx + (blockIdx.x / kXcds) : blockIdx.x;
    const uint32_t w = bid * kWavesPerBlock + wib;
    if (w >= p.n_chunks) return;
    const int k = p.k;
    const int cb = blockIdx.y * 256 + lane;
    const uint2 *__restrict__ rec = p.rec;
    const uint32_t t0 = p.chunk[w].x, t1 = t0 + p.chunk[w].y;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t zb = p.t_beg[t], ze = p.t_beg[t + 1];
        const uint32_t dst = p.t_dst[t];
        if ((dst & (kPartialFlag | kBundleFlag)) == (kPartialFlag | kBundleFlag)) {  // a bundle: its rows one after the other
            const uint2 a = p.t_aux[t];                                             // {first entry in bd_rows, steps}
            const uint32_t slots = slots_per_step;                                  // record j of row s at zb + j * slots + s
            for (uint32_t s = 0; s < slots; ++s) {
                const uint32_t row = p.bd_rows[a.x + s];
                if (row == kBundleNoRow) continue;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if ((row & kBundleZero) == 0) {
                    for (uint32_t j = 0; j < a.y; ++j) {
                        const uint2 r = rec[zb + j * slots + s];
                        const float v = as_f32(r.y);
                        const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + r.x)
                                                  : B + static_cast<uint64_t>(r.x) * p.ldb;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = cb + 64 * i;
                            if (c < k) acc[i] = fmaf(v, brow[c], acc[i]);
                        }
                    }
                }
                float *orow = C + static_cast<uint64_t>(row & ~kBundleZero) * p.ldc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = cb + 64 * i;
                    if (c < k) orow[c] = acc[i];
                }
            }
            continue;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t z = zb; z < ze; ++z) {
            const uint2 r = rec[z];
            const float v = as_f32(r.y);
            const float *brow = OFF32 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + r.x)
                                      : B + static_cast<uint64_t>(r.x) * p.ldb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cb + 64 * i;
                if (c < k) acc[i] = fmaf(v, brow[c], acc[i]);
            }
        }
        float *orow = (dst & kPartialFlag) ? p.partial + static_cast<uint64_t>(dst & ~kPartialFlag) * k
                                           : C + static_cast<uint64_t>(dst) * p.ldc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb + 64 * i;
            if (c < k) orow[c] = acc[i];
        }
    }
}

// C[row,:] = partial[first,:] + partial[first+1,:] + ... in that fixed order.  Eight pieces are
// loaded before the first add so the (tiny) kernel costs a couple of memory round trips, not
// one per piece; the adds stay strictly in piece order, so the result is reproducible.
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const float *__restrict__ partial,
                                                         const SplitRow *__restrict__ rows, uint32_t n_rows,
                                                         int k, int ldc, float *__restrict__ C) {
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i = blockIdx.x * kWavesPerBlock + wib;
    if (i >= n_rows) return;
    const SplitRow sr = rows[i];
    for (int c = lane; c < k; c += 64) {
        const float *p = partial + static_cast<uint64_t>(sr.first) * k + c;
        float s = 0.f;
        uint32_t j = 0;
        for (; j + 8 <= sr.count; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[static_cast<uint64_t>(j + u) * k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (j + u < sr.count) ? p[static_cast<uint64_t>(j + u) * k] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j + u < sr.count) s += v[u];
        C[static_cast<uint64_t>(sr.row) * ldc + c] = s;
    }
}

// dst[r,:] = src[idx[r],:]  (≙ flexspmm_v9_permuteX, flex.cu:276-289)
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_rows_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                          const int32_t *__restrict__ idx, int64_t n, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6); r < n; r += nwaves) {
        const int64_t s = idx[r];
        if constexpr (VEC4) {
            const float4 *sp = reinterpret_cast<const float4 *>(src + s * k);
            float4 *dp = reinterpret_cast<float4 *>(dst + r * k);
            for (int c = lane; c < k / 4; c += 64) dp[c] = sp[c];
        } else {
            for (int c = lane; c < k; c += 64) dst[r * k + c] = src[s * k + c];
        }
    }
}

template <int G, bool OFF32, int U>
int launch_v4(const PlanView &v, const float *dB, float *dC, hipStream_t s) {
    // one wave per chunk-table entry, kWavesPerBlock entries per workgroup, a multiple of 8 workgroups so
    // the XCD slices are equal (build_chunk_table pads them).  2 or 8 waves per workgroup measured the same as 4
    // (the launch is not dispatch-bound: an empty kernel over the same grid takes 3.4 us)
    uint32_t nblk = (v.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    nblk = (nblk + kXcds - 1) / kXcds * kXcds;
    const uint32_t ktiles = (v.k + 4 * G - 1) / (4 * G);
    const dim3 grid = v.tile_group ? dim3(nblk * ktiles, 1) : dim3(nblk, ktiles);
    hipLaunchKernelGGL((spmm_flat_kernel<G, OFF32, U, kWavesPerBlock>), grid, dim3(64 * kWavesPerBlock), v.lds_extra, s, v, dB, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

template <int G, int U>
int launch_stamped(const PlanView &v, bool off32, const float *dB, float *dC, hipStream_t s) {
    uint32_t nblk = (v.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    nblk = (nblk + kXcds - 1) / kXcds * kXcds;
    const uint32_t ktiles = (v.k + 4 * G - 1) / (4 * G);
    const dim3 grid = v.tile_group ? dim3(nblk * ktiles, 1) : dim3(nblk, ktiles);
    if (off32)
        hipLaunchKernelGGL((spmm_flat_kernel<G, true, U, kWavesPerBlock, true>), grid, dim3(64 * kWavesPerBlock), v.lds_extra, s, v, dB, dC);
    else
        hipLaunchKernelGGL((spmm_flat_kernel<G, false, U, kWavesPerBlock, true>), grid, dim3(64 * kWavesPerBlock), v.lds_extra, s, v, dB, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

template <int G, int U>
int launch_v4_off(const PlanView &v, bool off32, const float *dB, float *dC, hipStream_t s) {
    return off32 ? launch_v4<G, true, U>(v, dB, dC, s) : launch_v4<G, false, U>(v, dB, dC, s);
}

}  // namespace

int launch_spmm(const PlanView &v, int lanes_per_nz, bool off32, bool vec4, const float *dB, float *dC,
                hipStream_t s, int unroll) {
    if (v.n_chunks == 0) return FLEX_OK;
    if (!vec4) {
        uint32_t nblk = (v.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
        nblk = (nblk + kXcds - 1) / kXcds * kXcds;
        const uint32_t ktiles = (v.k + 255) / 256;
        if (off32)
            hipLaunchKernelGGL((spmm_generic_kernel<true>), dim3(nblk, ktiles), dim3(256), 0, s, v, dB, dC, 64u / static_cast<uint32_t>(lanes_per_nz));
        else
            hipLaunchKernelGGL((spmm_generic_kernel<false>), dim3(nblk, ktiles), dim3(256), 0, s, v, dB, dC, 64u / static_cast<uint32_t>(lanes_per_nz));
        FLEX_HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    if (unroll == 8) {  // tuning experiments (FLEX_U=8): twice the gathers in flight per wave on the narrow tiles
        switch (lanes_per_nz) {
            case 8: return launch_v4_off<8, 8>(v, off32, dB, dC, s);
            case 16: return launch_v4_off<16, 8>(v, off32, dB, dC, s);
            default: break;
        }
    }
    switch (lanes_per_nz) {  // U: 4 KiB in flight per wave on the narrow tiles, 8 KiB on the wide ones (U=8 on G<=16 measured the same)
        case 4: return launch_v4_off<4, 4>(v, off32, dB, dC, s);
        case 8: return launch_v4_off<8, 4>(v, off32, dB, dC, s);
        case 16: return launch_v4_off<16, 4>(v, off32, dB, dC, s);
        case 32: return launch_v4_off<32, 8>(v, off32, dB, dC, s);
        case 64: return launch_v4_off<64, 8>(v, off32, dB, dC, s);
        default: return FLEX_ERR_UNSUPPORTED;
    }
}

// the stamped twin of what launch_spmm launches for aligned operands (v.trace must hold 3 words per (k-tile, chunk-table entry))
int launch_spmm_stamped(const PlanView &v, int lanes_per_nz, bool off32, const float *dB, float *dC, hipStream_t s) {
    if (v.n_chunks == 0) return FLEX_OK;
    switch (lanes_per_nz) {
        case 4: return launch_stamped<4, 4>(v, off32, dB, dC, s);
        case 8: return launch_stamped<8, 4>(v, off32, dB, dC, s);
        case 16: return launch_stamped<16, 4>(v, off32, dB, dC, s);
        case 32: return launch_stamped<32, 8>(v, off32, dB, dC, s);
        case 64: return launch_stamped<64, 8>(v, off32, dB, dC, s);
        default: return FLEX_ERR_UNSUPPORTED;
    }
}

// Registers / LDS / occupancy of the kernel launch_spmm would pick (≙ Kernel_Info, flex.cu:4933-4941).
int kernel_attributes(int lanes_per_nz, bool off32, bool vec4, hipFuncAttributes *attr, int *waves_per_cu) {
    const void *fn = nullptr;
    if (!vec4) {
        fn = off32 ? reinterpret_cast<const void *>(spmm_generic_kernel<true>) : reinterpret_cast<const void *>(spmm_generic_kernel<false>);
    } else {
        switch (lanes_per_nz) {
            case 4: fn = off32 ? reinterpret_cast<const void *>(spmm_flat_kernel<4, true, 4, kWavesPerBlock>) : reinterpret_cast<const void *>(spmm_flat_kernel<4, false, 4, kWavesPerBlock>); break;
            case 8: fn = off32 ? reinterpret_cast<const void *>(spmm_flat_kernel<8, true, 4, kWavesPerBlock>) : reinterpret_cast<const void *>(spmm_flat_kernel<8, false, 4, kWavesPerBlock>); break;
            case 16: fn = off32 ? reinterpret_cast<const void *>(spmm_flat_kernel<16, true, 4, kWavesPerBlock>) : reinterpret_cast<const void *>(spmm_flat_kernel<16, false, 4, kWavesPerBlock>); break;
            case 32: fn = off32 ? reinterpret_cast<const void *>(spmm_flat_kernel<32, true, 8, kWavesPerBlock>) : reinterpret_cast<const void *>(spmm_flat_kernel<32, false, 8, kWavesPerBlock>); break;
            case 64: fn = off32 ? reinterpret_cast<const void *>(spmm_flat_kernel<64, true, 8, kWavesPerBlock>) : reinterpret_cast<const void *>(spmm_flat_kernel<64, false, 8, kWavesPerBlock>); break;
            default: return FLEX_ERR_UNSUPPORTED;
        }
    }
    FLEX_HIP_TRY(hipFuncGetAttributes(attr, fn));
    int blocks = 0;
    FLEX_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 64 * kWavesPerBlock, 0));
    *waves_per_cu = blocks * kWavesPerBlock;
    return FLEX_OK;
}

int launch_fixup(const float *partial, const SplitRow *rows, uint32_t n_rows, int k, int ldc, float *dC, hipStream_t s) {
    if (n_rows == 0) return FLEX_OK;
    const uint32_t nblk = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(spmm_fixup_kernel, dim3(nblk), dim3(256), 0, s, partial, rows, n_rows, k, ldc, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

int launch_gather_rows(float *dst, const float *src, const int32_t *idx, int64_t n, int k, hipStream_t s) {
    if (n <= 0) return FLEX_OK;
    const int64_t want = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t nblk = static_cast<uint32_t>(want < 2048 ? want : 2048);  // grid-stride beyond 8 blocks/CU
    const bool vec4 = (k % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL((gather_rows_kernel<true>), dim3(nblk), dim3(256), 0, s, dst, src, idx, n, k);
    else
        hipLaunchKernelGGL((gather_rows_kernel<false>), dim3(nblk), dim3(256), 0, s, dst, src, idx, n, k);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace flex
