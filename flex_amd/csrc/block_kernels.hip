// block_kernels.hip -- the hot-block kernel of libflex_spmm.so: B reuse ABOVE the L2, in LDS, for the nonzeros that have it.
//
// What it is the counterpart of: the reference's tiler exists to raise `u`, the number of nonzeros that use a B row once
// it has been fetched (cost model nD = 4/u + 8/k + ... bytes per FMA, flex.cu:5513-5528), by confining a queue of work
// to a column span that one SM's cache can hold (csr2_DiagTiling's rounds, mat.cu:680-942; csr2seg_Cmajor, mat.cu:1192-1269;
// the per-SM queues of flex.cu:4008-4124).  The flat kernel (spmm_kernels.hip) has u = 1 above the L2: every nonzero pulls
// its own 16*G bytes through the texture path.  Here the planner SPLITS the matrix (block_plan.cpp): the nonzeros whose
// column is used several times inside a block of schedule-consecutive rows (the members of the block's community, hubs) are
// HOT and are multiplied here out of LDS; everything else -- by construction the nonzeros that miss the L2 anyway -- stays
// with the flat kernel, which runs first and writes C; this kernel ADDS its part to the C rows of its blocks.
//
// Shape (MI355X: 160 KiB of LDS and 32 wave slots per CU, one workgroup per CU):
//   * 16 waves = 15 CONSUMER waves + 1 LOADER wave; one workgroup = one (block, 64-column tile of k), tiles the slow grid dimension.
//   * A consumer wave is 4 slots of 16 lanes; a slot holds ONE C row per round (ROUNDS rows in all, acc[ROUNDS] float4 per lane:
//     the lane owns 4 of the tile's 64 columns): no cross-lane reduction, C read once and written once.
//   * The block's hot B rows are staged panel by panel (up to 304 rows x 256 bytes) by the loader wave with LDS-DMA
//     (global_load_lds_dwordx4, per-lane source = a row gather) into the other of two buffers while the consumers work on the
//     current one: one s_barrier per panel.  (An experiment build, -DFLEX_BK_NBUF=3, keeps a ring of three 200-row buffers with the
//     loader two panels ahead behind counted vmcnt waits: measured slower -- shorter panels mean more runs and more barriers.)  A slot's ds_read_b128 covers one whole 256-byte row = all 64 banks once, so the
//     reads are conflict-free whatever rows the four slots of a wave are on (MI355X_MICROARCH.md, LDS: the 16-lane service
//     groups hold quarter-rows of different slots, whose banks depend only on the lane).
//   * Records {offset inside the panel buffer, value} never touch LDS.  The stream of a wave is [step][slot], panel-major; the
//     <= 16 steps of one RUN (one panel, one round) are fetched by ONE coalesced 512-byte load -- lane 16 s + j takes the record
//     of (step j, slot s) -- a whole panel ahead of their use, and at step j the 16 lanes of a slot get their record by a DPP
//     row broadcast (row_newbcast:j, fused by the compiler into a v_mov_dpp pair): the LDS pipe carries nothing but B rows,
//     the texture path one load per run, the VALU 2 broadcasts + 1 address add + 2 v_pk_fma_f32 per step of 4 x 64 multiply-adds.
#include "internal.h"

namespace flex {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ void fma4(v4f &acc, float v, const v4f &b) {
    acc.x = fmaf(v, b.x, acc.x);
    acc.y = fmaf(v, b.y, acc.y);
    acc.z = fmaf(v, b.z, acc.z);
    acc.w = fmaf(v, b.w, acc.w);
}

// lane J of every 16-lane row, to all lanes of that row (DPP row_newbcast: gfx90a and later)
template <int J>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x150 + J, 0xf, 0xf, true));
}

// LDS is handed from the loader to the consumers (and back) by plain workgroup barriers.  Consumers keep their record
// prefetch (ordinary global loads) in flight across the barrier, so they wait for their LDS operations only; the loader
// waits for its DMA (vmcnt) as well.
__device__ __forceinline__ void consumer_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void loader_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- loader wave -------------------------------------------------------------------------------------------------------
// hcol (the byte offsets of a panel's B rows) travels through a small LDS scratch one panel ahead of its use, so that the
// loader never waits for an index before it can issue a panel's DMA.
__device__ __forceinline__ void dma_hcol(char *lds, const uint32_t *__restrict__ src, uint32_t panel_rows, uint32_t scratch, int lane) {
    // panel_rows x 4 bytes, contiguous: 16 bytes per lane, two instructions cover up to 512 entries
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t e = static_cast<uint32_t>(i) * 256u + static_cast<uint32_t>(lane) * 4u;
        if (e < panel_rows)
            __builtin_amdgcn_global_load_lds((gl_void *)(src + e), (lds_void *)(lds + kBkLdsHcol + scratch * (kBkPanelMax * 4) + i * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ void dma_panel(char *lds, const char *__restrict__ Bb, uint32_t lane_goff, uint32_t panel_rows, uint32_t scratch,
                                          uint32_t buf, int lane) {
    const uint32_t *boff = reinterpret_cast<const uint32_t *>(lds + kBkLdsHcol + scratch * (kBkPanelMax * 4)) + (lane >> 4);
    char *dst = lds + buf * kBkBufBytes;
    const uint32_t ngrp = panel_rows / 4;  // one instruction = 4 rows x 256 bytes = 1 KiB of the buffer
    uint32_t g = 0;
    for (; g + 4 <= ngrp; g += 4) {
        uint32_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = boff[(g + u) * 4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((gl_void *)(Bb + o[u] + lane_goff), (lds_void *)(dst + (g + u) * 1024), 16, 0, 0);
    }
    for (; g < ngrp; ++g)
        __builtin_amdgcn_global_load_lds((gl_void *)(Bb + boff[g * 4] + lane_goff), (lds_void *)(dst + g * 1024), 16, 0, 0);
}

// Diagnostic build only (make -C flex_amd/csrc trace; tools/trace_blocks.py): where a wave's cycles go.  Counters per wave:
// 0 prologue (C rows, first records), 2 panel work, 3 waiting at barriers, 5 epilogue, 6 total, 7 steps.
#ifdef FLEX_TRACE
#define BK_STAMP(i)                                        \
    do {                                                   \
        const uint64_t now_ = __builtin_amdgcn_s_memtime(); \
        bk_ph[i] += now_ - bk_last;                        \
        bk_last = now_;                                    \
    } while (0)
#else
#define BK_STAMP(i) do {} while (0)
#endif

template <int ROUNDS>
__global__ __launch_bounds__(64 * (kBkWaves + 1)) void spmm_hot_kernel(BlockView v, const float *__restrict__ B, float *__restrict__ C) {
    __shared__ __attribute__((aligned(256))) char lds[kBkLdsBytes];
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cpx = gridDim.x / kXcds;  // gridDim.x % 8 == 0
    const uint32_t blk = v.xcd_remap ? (blockIdx.x % kXcds) * cpx + blockIdx.x / kXcds : blockIdx.x;
    if (blk >= v.n_blocks) return;  // the whole workgroup: no barrier has been reached
    const uint4 hdr = v.hdr[blk];
    const uint32_t np = hdr.x & 0x7FFFFFFFu;
    const bool chains = (hdr.x >> 31) != 0;  // some row of the block has several parts: one more barrier, they meet in LDS
    if (np == 0) return;  // a block without hot columns (the whole workgroup)
    const int k = v.k;
    // One workgroup = one (block, 64-column tile).  Tiles are the SLOW grid dimension: the hardware dispatches all blocks of
    // tile 0 before tile 1, so at any time the staging of the whole chip falls into one 256-byte slice of every B row.
    const int t = blockIdx.y;
    const char *__restrict__ Bb = reinterpret_cast<const char *>(B);
    const int l16 = lane & 15;
    const int c0 = t * kBkTileCols + l16 * 4;
    const bool col_ok = c0 < k;
    // lanes past column k fetch the tile's first column (valid); their bytes are never used for a stored value
    const uint32_t lane_goff = static_cast<uint32_t>(col_ok ? c0 : t * kBkTileCols) * 4u;

    if (w == kBkWaves) {
        // ---- the loader wave: np + 1 barriers (+ 1 when rows have several parts), exactly as many as every consumer wave.
        // It runs kBkNBuf - 1 panels ahead of the consumers.  vmcnt retires in order, so what it waits for is counted in
        // instructions: H(q) = the DMA of panel q's byte offsets into the scratch (ONE instruction: panel_rows <= 256 on the fast
        // path), D(q) = the panel's rows (panel_rows / 4 instructions of 1 KiB).  Issue order: ... H(q+1) D(q) H(q+2) D(q+1) ...:
        // H(q+1) precedes D(q), so waiting for the offsets of the NEXT panel never waits for the rows of this one.
        for (uint32_t b = 0; b < kBkNBuf; ++b) *reinterpret_cast<uint32_t *>(lds + b * kBkBufBytes + kBkZeroRow + lane * 4) = 0u;  // the rows of zeros padding records point at
        const uint32_t P = v.panel_rows;
        const uint32_t *__restrict__ hcol = v.hcol + hdr.y;
        const bool stage = !(v.ablate & 1);
        auto H = [&](uint32_t q) { dma_hcol(lds, hcol + static_cast<uint64_t>(q) * P, P, q & 1, lane); };
        auto D = [&](uint32_t q) {
            if (stage) dma_panel(lds, Bb, lane_goff, P, q & 1, q % kBkNBuf, lane);
        };
        constexpr uint32_t kD = kBkPanelMax / 4;  // instructions of one D on the fast path
        const bool fast = kBkNBuf == 3 && P == kBkPanelMax && P <= 256 && stage;  // counted waits need compile-time counts
        if (fast) {
            // exactly ONE instruction per H (lane 0 is always active): the counts below depend on it
            auto H = [&](uint32_t q) {
                if (static_cast<uint32_t>(lane) * 4u < P)
                    __builtin_amdgcn_global_load_lds((gl_void *)(hcol + static_cast<uint64_t>(q) * P + lane * 4), (lds_void *)(lds + kBkLdsHcol + (q & 1) * (kBkPanelMax * 4)), 16, 0, 0);
            };
            // prologue: H0 H1 | D0 H2 | D1, panel 0 landed.  After it: issued ... H(p+2) D(p+1) at the top of iteration p.
            H(0);
            if (np > 1) {
                H(1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1) : "memory");  // H0: behind it H1
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            D(0);
            if (np > 2) H(2);
            if (np > 1) {
                if (np > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kD + 1) : "memory");  // H1: behind it D0 and H2
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kD) : "memory");            //     behind it D0
                D(1);
            }
            if (np > 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kD + 1) : "memory");  // D0: behind it H2 and D1
            else loader_barrier();
            for (uint32_t p = 0; p < np; ++p) {
                if (p + 3 < np) {  // the steady state: H(p+3), D(p+2) go out; D(p+1) must have landed by the barrier
                    H(p + 3);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kD + 1) : "memory");  // H(p+2): behind it D(p+1) and H(p+3)
                    D(p + 2);
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kD + 1) : "memory");  // D(p+1): behind it H(p+3) and D(p+2)
                } else {           // the last panels: nothing left to count against -- whatever is in flight has to land anyway
                    if (p + 2 < np) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kD) : "memory");  // H(p+2): behind it D(p+1)
                        D(p + 2);
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kD) : "memory");  // D(p+1): behind it D(p+2)
                    } else {
                        loader_barrier();
                    }
                }
            }
        } else {
            // any panel size, two or three buffers: one panel ahead, every wait a full drain
            H(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            D(0);
            if (np > 1) H(1);
            loader_barrier();  // panel 0 staged (the consumers were fetching their C rows and first records meanwhile)
            for (uint32_t p = 0; p < np; ++p) {
                if (p + 1 < np) D(p + 1);
                if (p + 2 < np) H(p + 2);
                loader_barrier();  // consumers are done with panel p; panel p+1 has landed
            }
        }
        if (chains) loader_barrier();
        return;
    }

    // ---- a consumer wave
#ifdef FLEX_TRACE
    uint64_t bk_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t bk_last = __builtin_amdgcn_s_memtime();
    const uint64_t bk_t0 = bk_last;
#endif
    const uint32_t slot = static_cast<uint32_t>(lane) >> 4;
    const uint2 ws = v.wstart[static_cast<uint64_t>(blk) * kBkWaves + w];
    // Loads that run past the end of this wave's stream read the NEXT wave's records (harmless: such steps are never executed);
    // loads past the end of the whole array are clamped to its last record (v.n_rec >= 1).
    const uint64_t first_rec = min(static_cast<uint64_t>(ws.x) * kBkSlots, v.n_rec - 1);
    const uint2 *__restrict__ rec = v.rec + first_rec;
    const uint32_t last_rec = static_cast<uint32_t>(min(v.n_rec - 1 - first_rec, static_cast<uint64_t>(0xFFFFFFFFu)));
    // step counts of the runs: two 16-bit counts per word, the words held one per lane
    const uint32_t cw = hdr.w;
    const uint32_t *__restrict__ cnt = v.cnt + hdr.z + static_cast<uint64_t>(w) * cw;
    // run counts: lane p holds panel p's 64-bit word (byte r = steps of round r); lanes past the last panel hold 0
    const bool has_cnt = static_cast<uint32_t>(lane) < np;
    const uint32_t c_lo = has_cnt ? cnt[2 * lane] : 0u, c_hi = has_cnt ? cnt[2 * lane + 1] : 0u;
    auto count_of = [](uint32_t lo, uint32_t hi, int r) -> uint32_t { return ((r < 4 ? lo : hi) >> (8 * (r & 3))) & 0xFFu; };
    uint32_t rows[ROUNDS];
    v4f acc[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) rows[r] = v.brow[((static_cast<uint64_t>(blk) * ROUNDS + r) * kBkWaves + w) * kBkSlots + slot];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {  // the flat kernel's part of the row: this kernel adds to it
        const uint32_t row = rows[r] == kBkEmptyRow ? 0u : rows[r];
        v4f cur = {0.f, 0.f, 0.f, 0.f};
        if (!(v.ablate & 16)) cur = *reinterpret_cast<const v4f *>(C + static_cast<uint64_t>(row) * v.ldc + (col_ok ? c0 : 0));  // (wave-uniform; timing-only)
        acc[r] = (rows[r] != kBkEmptyRow && col_ok) ? cur : v4f{0.f, 0.f, 0.f, 0.f};
    }
    // run r of the next panel is fetched as soon as run r of this one is done with its registers: one panel of lead
    uint32_t rx[ROUNDS], ry[ROUNDS];
    uint32_t pos_pf = 0;  // step (wave-relative) at which the next run to fetch begins
    auto fetch_run = [&](int r, uint32_t n_run) {  // n_run: the steps of the run being fetched (the next one starts behind it)
        const uint32_t idx = (v.ablate & 32) ? static_cast<uint32_t>(lane) : min((pos_pf + static_cast<uint32_t>(l16)) * kBkSlots + slot, last_rec);  // 32: timing-only, one hot line
        const v2u q = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(rec + idx));  // read once per tile
        rx[r] = q.x;
        ry[r] = q.y;
        pos_pf += n_run;
    };
    {
        const uint32_t lo = __builtin_amdgcn_readlane(c_lo, 0), hi = __builtin_amdgcn_readlane(c_hi, 0);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) fetch_run(r, count_of(lo, hi, r));
    }
    BK_STAMP(0);
    consumer_barrier();  // panel 0 has landed
    BK_STAMP(3);

    for (uint32_t p = 0; p < np; ++p) {
        const uint32_t base = (p % kBkNBuf) * kBkBufBytes + static_cast<uint32_t>(l16) * 16u;
        const char *panel = lds + base;
        // this panel's eight run counts and the next panel's (lane np holds 0): four v_readlane per panel, a bit-field extract per run
        const uint32_t lo = __builtin_amdgcn_readlane(c_lo, p), hi = __builtin_amdgcn_readlane(c_hi, p);
        const uint32_t nlo = __builtin_amdgcn_readlane(c_lo, p + 1), nhi = __builtin_amdgcn_readlane(c_hi, p + 1);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t n = (v.ablate & 2) ? 0u : count_of(lo, hi, r);
            const uint32_t x = rx[r], y = ry[r];
            v4f a = acc[r];
            // four steps at a time while the run has them (wave-uniform branches; step indices are compile-time: DPP controls are immediates)
#define BK_STEP(J, OFF, VAL)                \
    const uint32_t OFF = row_bcast<J>(x);   \
    const float VAL = as_f32(row_bcast<J>(y));
#define BK_QUAD(Q)                                                                                   \
    if (n >= 4 * Q + 4) {                                                                            \
        BK_STEP(4 * Q + 0, o0, f0) BK_STEP(4 * Q + 1, o1, f1) BK_STEP(4 * Q + 2, o2, f2) BK_STEP(4 * Q + 3, o3, f3) \
        const v4f b0 = *reinterpret_cast<const v4f *>(panel + o0);                                    \
        const v4f b1 = *reinterpret_cast<const v4f *>(panel + o1);                                    \
        const v4f b2 = *reinterpret_cast<const v4f *>(panel + o2);                                    \
        const v4f b3 = *reinterpret_cast<const v4f *>(panel + o3);                                    \
        fma4(a, f0, b0);                                                                             \
        fma4(a, f1, b1);                                                                             \
        fma4(a, f2, b2);                                                                             \
        fma4(a, f3, b3);                                                                             \
    } else if (n > 4 * Q) {                                                                          \
        const uint32_t rem = n - 4 * Q; /* 1 .. 3 */                                                  \
        BK_STEP(4 * Q + 0, o0, f0) BK_STEP(4 * Q + 1, o1, f1) BK_STEP(4 * Q + 2, o2, f2)              \
        const v4f b0 = *reinterpret_cast<const v4f *>(panel + o0);                                    \
        fma4(a, f0, b0);                                                                             \
        if (rem > 1) {                                                                               \
            const v4f b1 = *reinterpret_cast<const v4f *>(panel + o1);                                \
            fma4(a, f1, b1);                                                                         \
        }                                                                                            \
        if (rem > 2) {                                                                               \
            const v4f b2 = *reinterpret_cast<const v4f *>(panel + o2);                                \
            fma4(a, f2, b2);                                                                         \
        }                                                                                            \
    }
            { BK_QUAD(0) }
            if (n > 4) {  // most runs are done after one group of four steps: one branch skips the other three
                { BK_QUAD(1) }
                if (n > 8) {
                    { BK_QUAD(2) }
                    { BK_QUAD(3) }
                }
            }
#undef BK_QUAD
#undef BK_STEP
            acc[r] = a;
            // unconditional (an index past the stream is clamped, its count is 0): a load inside a branch costs an s_waitcnt vmcnt(0)
            fetch_run(r, count_of(nlo, nhi, r));
        }
        BK_STAMP(2);
        consumer_barrier();
        BK_STAMP(3);
    }
    // A long row's parts: every later part leaves its sum (and the slot of the part after it) in LDS -- the panel buffers are free
    // now -- and after one more barrier the owner adds them in chain order: a fixed order, so the result is reproducible.
    if (chains) {  // workgroup-uniform
        uint32_t link[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) link[r] = v.link[((static_cast<uint64_t>(blk) * ROUNDS + r) * kBkWaves + w) * kBkSlots + slot];
        uint32_t *nxt = reinterpret_cast<uint32_t *>(lds + kBkLdsNext);  // [rounds x 60] next part + 1
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (link[r] & kBkLinkPart) {
                const uint32_t me = (static_cast<uint32_t>(r) * kBkWaves + w) * kBkSlots + slot;
                *reinterpret_cast<v4f *>(lds + me * kBkRowBytes + l16 * 16) = acc[r];
                if (l16 == 0) nxt[me] = link[r] & 0xFFFFu;
            }
        }
        consumer_barrier();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (link[r] & kBkLinkOwner) {
                uint32_t hops = 0;  // a chain has fewer links than the block has slots: the bound only keeps a damaged image from spinning
                for (uint32_t n = link[r] & 0xFFFFu; n != 0 && n <= ROUNDS * kBkRowsPerRound && hops < ROUNDS * kBkRowsPerRound; n = nxt[n - 1], ++hops) {
                    const v4f o = *reinterpret_cast<const v4f *>(lds + (n - 1) * kBkRowBytes + l16 * 16);
                    acc[r] += o;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
        if (rows[r] != kBkEmptyRow && col_ok && !(v.ablate & 16)) *reinterpret_cast<v4f *>(C + static_cast<uint64_t>(rows[r]) * v.ldc + c0) = acc[r];
#ifdef FLEX_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BK_STAMP(5);
    bk_ph[6] = bk_last - bk_t0;
    bk_ph[7] = ws.y;
    if (lane == 0 && v.trace != nullptr) {
        uint64_t *o = v.trace + ((static_cast<uint64_t>(blockIdx.y) * v.n_blocks + blk) * kBkWaves + w) * 8;
        for (int i = 0; i < 8; ++i) o[i] = bk_ph[i];
    }
#endif
}

// The same block image for operands the fast kernel cannot take (B or C not 16-byte aligned: LDS-DMA and the float4 accesses need
// it): no staging, no DPP -- a lane owns one column of the 64-column tile, reads its records straight from the stream and its B
// values with 4-byte loads through the panel's offset list.  Same work split (one workgroup per (block, tile), one wave per consumer
// stream, the parts of a long row meeting in LDS in chain order), so the sums are formed in the same order as in the fast kernel's
// slots up to the order inside a step; a correctness path, like spmm_generic_kernel for the flat part.
template <int ROUNDS>
__global__ __launch_bounds__(64 * kBkWaves) void spmm_hot_generic_kernel(BlockView v, const float *__restrict__ B, float *__restrict__ C) {
    __shared__ float part_sum[ROUNDS * kBkRowsPerRound][kBkTileCols];  // 480 x 64 floats = 120 KiB at ROUNDS = 8
    __shared__ uint32_t part_next[ROUNDS * kBkRowsPerRound];
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = blockIdx.x;
    const uint4 hdr = v.hdr[blk];
    const uint32_t np = hdr.x & 0x7FFFFFFFu;
    const bool chains = (hdr.x >> 31) != 0;
    if (np == 0) return;
    const int c = blockIdx.y * kBkTileCols + lane;
    const bool col_ok = c < v.k;
    const uint32_t P = v.panel_rows;
    const uint32_t *__restrict__ hcol = v.hcol + hdr.y;
    const uint2 ws = v.wstart[static_cast<uint64_t>(blk) * kBkWaves + w];
    const uint2 *__restrict__ rec = v.rec + static_cast<uint64_t>(ws.x) * kBkSlots;
    const uint32_t cw = hdr.w;
    const uint32_t *__restrict__ cnt = v.cnt + hdr.z + static_cast<uint64_t>(w) * cw;
    float acc[ROUNDS][kBkSlots];
    uint32_t rows[ROUNDS][kBkSlots];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
        for (int sl = 0; sl < kBkSlots; ++sl) {
            rows[r][sl] = v.brow[((static_cast<uint64_t>(blk) * ROUNDS + r) * kBkWaves + w) * kBkSlots + sl];
            acc[r][sl] = (rows[r][sl] != kBkEmptyRow && col_ok) ? C[static_cast<uint64_t>(rows[r][sl]) * v.ldc + c] : 0.f;
        }
    uint32_t pos = 0;
    for (uint32_t p = 0; p < np; ++p)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t n = (cnt[2 * p + r / 4] >> (8 * (r & 3))) & 0xFFu;
            for (uint32_t j = 0; j < n; ++j, ++pos)
#pragma unroll
                for (int sl = 0; sl < kBkSlots; ++sl) {
                    const uint2 q = rec[static_cast<uint64_t>(pos) * kBkSlots + sl];
                    if (q.x == kBkZeroRow) continue;  // padding (wave-uniform: the record is the same for every lane)
                    const uint32_t boff = hcol[static_cast<uint64_t>(p) * P + q.x / kBkRowBytes];
                    const float b = col_ok ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + boff)[c] : 0.f;
                    acc[r][sl] = fmaf(__uint_as_float(q.y), b, acc[r][sl]);
                }
        }
    if (chains) {  // workgroup-uniform
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int sl = 0; sl < kBkSlots; ++sl) {
                const uint32_t me = (static_cast<uint32_t>(r) * kBkWaves + w) * kBkSlots + sl;
                const uint32_t l = v.link[static_cast<uint64_t>(blk) * ROUNDS * kBkRowsPerRound + me];
                if (l & kBkLinkPart) {
                    part_sum[me][lane] = acc[r][sl];
                    if (lane == 0) part_next[me] = l & 0xFFFFu;
                }
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int sl = 0; sl < kBkSlots; ++sl) {
                const uint32_t me = (static_cast<uint32_t>(r) * kBkWaves + w) * kBkSlots + sl;
                const uint32_t l = v.link[static_cast<uint64_t>(blk) * ROUNDS * kBkRowsPerRound + me];
                if (l & kBkLinkOwner) {
                    uint32_t hops = 0;
                    for (uint32_t nx = l & 0xFFFFu; nx != 0 && nx <= ROUNDS * kBkRowsPerRound && hops < ROUNDS * kBkRowsPerRound; nx = part_next[nx - 1], ++hops)
                        acc[r][sl] += part_sum[nx - 1][lane];
                }
            }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
        for (int sl = 0; sl < kBkSlots; ++sl)
            if (rows[r][sl] != kBkEmptyRow && col_ok) C[static_cast<uint64_t>(rows[r][sl]) * v.ldc + c] = acc[r][sl];
}

template <int ROUNDS>
int launch_rounds(const BlockView &bv, const float *dB, float *dC, hipStream_t s, bool vec4) {
    if (!vec4) {
        hipLaunchKernelGGL((spmm_hot_generic_kernel<ROUNDS>), dim3(bv.n_blocks, (bv.k + kBkTileCols - 1) / kBkTileCols), dim3(64 * kBkWaves), 0, s, bv, dB, dC);
        FLEX_HIP_TRY(hipGetLastError());
        return FLEX_OK;
    }
    const uint32_t nblk = (bv.n_blocks + kXcds - 1) / kXcds * kXcds;
    hipLaunchKernelGGL((spmm_hot_kernel<ROUNDS>), dim3(nblk, (bv.k + kBkTileCols - 1) / kBkTileCols), dim3(64 * (kBkWaves + 1)), 0, s, bv, dB, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace

int launch_blocks(const BlockView &bv, const float *dB, float *dC, hipStream_t s, bool vec4) {
    if (bv.n_blocks == 0) return FLEX_OK;
    switch (bv.rounds) {
        case 2: return launch_rounds<2>(bv, dB, dC, s, vec4);
        case 4: return launch_rounds<4>(bv, dB, dC, s, vec4);
        case 8: return launch_rounds<8>(bv, dB, dC, s, vec4);
        default: return FLEX_ERR_UNSUPPORTED;
    }
}

}  // namespace flex
