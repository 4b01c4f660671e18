// block_kernels.hip -- the row-block kernel of libflex_spmm.so: B reuse ABOVE the L2, in LDS.
//
// What it is the counterpart of: the reference's tiler exists to raise `u`, the number of nonzeros that use a B row once
// it has been fetched (cost model nD = 4/u + 8/k + ... bytes per FMA, flex.cu:5513-5528), by confining a queue of work
// to a column span that one SM's cache can hold (csr2_DiagTiling's rounds, mat.cu:680-942; csr2seg_Cmajor, mat.cu:1192-1269;
// the per-SM queues of flex.cu:4008-4124).  The flat kernel (spmm_kernels.hip) has u = 1 above the L2: every nonzero pulls
// its own 16*G bytes through the texture path.  Here a workgroup OWNS a block of schedule-consecutive rows; the B rows that
// several of the block's nonzeros use (its HOT columns, found by the planner: block_plan.cpp) are staged once per column
// tile in LDS and every use is a ds_read_b128; only the block's remaining (COLD) nonzeros gather from global memory.
//
// Shape (MI355X: 160 KiB of LDS and 32 wave slots per CU, one workgroup per CU):
//   * 16 waves = 15 CONSUMER waves + 1 LOADER wave.
//   * A consumer wave is 8 slots of 8 lanes; a slot walks ONE C row (or one of 2/4/8 equal parts of a long row, summed by
//     a butterfly over the slots at the end) and each lane owns 4 of the tile's 32 columns: no cross-lane reduction per
//     row, the sum stays in registers through every phase, C is written once.  `rounds` rows per slot (accumulators
//     acc[rounds]), so a block is rounds x 120 row slots.
//   * Phases of a block, per 32-column tile of k:  cold (gathers from global memory, as the flat kernel at G = 8), then
//     one phase per PANEL of up to 480 hot B rows (60 KiB of LDS).  The loader wave stages panel p+1 with LDS-DMA
//     (global_load_lds_dwordx4, per-lane source = row gather) into the other buffer while the consumers work on panel p;
//     one s_barrier per panel.  The consumers issue no load that the loader's DMA could delay (vmcnt is per wave).
//   * One workgroup = one (block, 32-column tile of k); tiles are the slow grid dimension.
//   * A consumer wave's records are ONE sequential stream [step][slot] of {offset, value}, the same for every column tile:
//     fetched with coalesced 512-byte loads one window (32 steps) ahead, staged in a wave-private 2 KiB LDS window, and
//     consumed phase by phase -- the planner pads a (phase, round) group to its longest slot with records of value 0 that
//     point at a row of zeros (panel phases) or at a column the row uses anyway (cold phase), so a non-finite B value only
//     reaches rows that reference it.
#include "internal.h"

namespace flex {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

__device__ __forceinline__ float as_f32(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ void fma4(float4 &acc, float v, const float4 &b) {
    acc.x = fmaf(v, b.x, acc.x);
    acc.y = fmaf(v, b.y, acc.y);
    acc.z = fmaf(v, b.z, acc.z);
    acc.w = fmaf(v, b.w, acc.w);
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
    const uint32_t *boff = reinterpret_cast<const uint32_t *>(lds + kBkLdsHcol + scratch * (kBkPanelMax * 4)) + (lane >> 3);
    char *dst = lds + buf * kBkBufBytes;
    const uint32_t ngrp = panel_rows / 8;  // one instruction = 8 rows x 128 bytes = 1 KiB of the buffer
    uint32_t g = 0;
    for (; g + 4 <= ngrp; g += 4) {
        uint32_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = boff[(g + u) * 8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((gl_void *)(Bb + o[u] + lane_goff), (lds_void *)(dst + (g + u) * 1024), 16, 0, 0);
    }
    for (; g < ngrp; ++g)
        __builtin_amdgcn_global_load_lds((gl_void *)(Bb + boff[g * 8] + lane_goff), (lds_void *)(dst + g * 1024), 16, 0, 0);
}

// Diagnostic build only (make -C flex_amd/csrc trace; tools/trace_blocks.py): where a wave's cycles go.  Counters per wave:
// 0 prologue, 1 cold phase, 2 panel phases, 3 waiting at barriers, 4 window refills (also inside 1 and 2), 5 epilogue, 6 total.
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

template <int ROUNDS, int U_HOT, int U_COLD>
__global__ __launch_bounds__(64 * (kBkWaves + 1)) void spmm_block_kernel(BlockView v, const float *__restrict__ B, float *__restrict__ C) {
    __shared__ __attribute__((aligned(128))) char lds[kBkLdsBytes];
    const int lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cpx = gridDim.x / kXcds;  // gridDim.x % 8 == 0
    const uint32_t blk = v.xcd_remap ? (blockIdx.x % kXcds) * cpx + blockIdx.x / kXcds : blockIdx.x;
    if (blk >= v.n_blocks) return;  // the whole workgroup: no barrier has been reached
    const uint4 hdr = v.hdr[blk];
    const uint32_t np = hdr.x & 0x7FFFFFFFu;
    const bool has_hubs = (hdr.x >> 31) != 0;  // rows spread over several waves: one more barrier, their parts meet in LDS
    const int k = v.k;
    // One workgroup = one (block, 32-column tile).  Tiles are the SLOW grid dimension: the hardware dispatches all blocks of
    // tile 0 before tile 1, so at any time the gathers of the whole chip fall into one 128-byte slice of every B row --
    // n x 128 bytes (amazon shape: 201 MB) instead of n x 4k, which is what lets the Infinity Cache (256 MiB) serve part of
    // the cold phase's misses.
    const int t = blockIdx.y;
    const char *__restrict__ Bb = reinterpret_cast<const char *>(B);
    const int l8 = lane & 7;

    if (w == kBkWaves) {
        // ---- the loader wave: per column tile np + 1 barriers, exactly as many as every consumer wave
        {  // the two rows of zeros (256 bytes) per buffer that padding records point at; never written again
            *reinterpret_cast<uint32_t *>(lds + kBkZeroRow + lane * 4) = 0u;
            *reinterpret_cast<uint32_t *>(lds + kBkBufBytes + kBkZeroRow + lane * 4) = 0u;
        }
        const uint32_t P = v.panel_rows;
        const uint32_t *__restrict__ hcol = v.hcol + hdr.y;
        {
            // lanes past column k fetch the tile's first column (valid); their LDS bytes are never used for a stored value
            const int c0 = t * 32 + l8 * 4;
            const uint32_t lane_goff = static_cast<uint32_t>(c0 < k ? c0 : t * 32) * 4u;
            if (np > 0) {
                dma_hcol(lds, hcol, P, 0, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(v.ablate & 1)) dma_panel(lds, Bb, lane_goff, P, 0, 0, lane);
                if (np > 1) dma_hcol(lds, hcol + P, P, 1, lane);
            }
            loader_barrier();  // panel 0 staged (the consumers were in their cold phase meanwhile)
            for (uint32_t p = 0; p < np; ++p) {
                if (p + 1 < np && !(v.ablate & 1)) dma_panel(lds, Bb, lane_goff, P, (p + 1) & 1, (p + 1) & 1, lane);
                if (p + 2 < np) dma_hcol(lds, hcol + static_cast<uint64_t>(p + 2) * P, P, p & 1, lane);
                loader_barrier();  // consumers are done with panel p; panel p+1 has landed
            }
            if (has_hubs) loader_barrier();
        }
        return;
    }

    // ---- a consumer wave
#ifdef FLEX_TRACE
    uint64_t bk_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t bk_last = __builtin_amdgcn_s_memtime();
    const uint64_t bk_t0 = bk_last;
#endif
    const uint32_t slot = static_cast<uint32_t>(lane) >> 3;
    const uint2 ws = v.wstart[static_cast<uint64_t>(blk) * kBkWaves + w];
    const uint32_t T = ws.y;  // steps of this wave's stream, the same for every column tile
    const uint2 *__restrict__ rec = v.rec + static_cast<uint64_t>(ws.x) * kBkSlots;
    const uint32_t n_win = (T + kBkWinSteps - 1) / kBkWinSteps;
    const uint32_t last_rec = T * kBkSlots - 1;
    // step counts of the (phase, round) groups: two 16-bit counts per word, the words held one per lane
    const uint32_t cw = hdr.w;
    const uint32_t *__restrict__ cnt = v.cnt + hdr.z + static_cast<uint64_t>(w) * cw;
    const uint32_t creg0 = static_cast<uint32_t>(lane) < cw ? cnt[lane] : 0u;
    const uint32_t creg1 = static_cast<uint32_t>(lane) + 64u < cw ? cnt[lane + 64] : 0u;
    auto steps_of = [&](uint32_t idx) -> uint32_t {
        const uint32_t word = idx >> 1;
        const uint32_t v32 = word < 64 ? __builtin_amdgcn_readlane(creg0, word) : __builtin_amdgcn_readlane(creg1, word - 64);
        return (idx & 1) ? v32 >> 16 : v32 & 0xFFFFu;
    };
    uint32_t rows[ROUNDS], hub[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        rows[r] = v.brow[(static_cast<uint64_t>(blk) * ROUNDS + r) * kBkRowsPerRound + w * kBkSlots + slot];
        hub[r] = has_hubs ? __builtin_amdgcn_readfirstlane(v.bgrp[(static_cast<uint64_t>(blk) * ROUNDS + r) * kBkWaves + w]) : 0u;
    }

    uint2 *win = reinterpret_cast<uint2 *>(lds + kBkLdsWin + w * (kBkWinSteps * kBkSlots * 8));
    uint32_t pw = 0, wpos = 0, wend = 0;  // window held in `nxt`, position and end (steps) inside the window in LDS
    constexpr int kWinLoads = kBkWinSteps * kBkSlots / 64;  // coalesced 512-byte loads per window
    uint2 nxt[kWinLoads];
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    auto prefetch = [&](uint32_t widx) {  // read once per tile: non-temporal, so the stream does not displace B rows in L2 / Infinity Cache
        const uint32_t s0 = widx * (kBkWinSteps * kBkSlots) + lane;
#pragma unroll
        for (int i = 0; i < kWinLoads; ++i) {
            const v2u r = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(rec + min(s0 + i * 64u, last_rec)));
            nxt[i] = make_uint2(r.x, r.y);
        }
    };
    if (T) prefetch(0);
    auto refill = [&]() {
#ifdef FLEX_TRACE
        const uint64_t r0_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int i = 0; i < kWinLoads; ++i) win[i * 64 + lane] = nxt[i];
        wend = min(kBkWinSteps, T - pw * kBkWinSteps);
        wpos = 0;
        if (++pw < n_win) prefetch(pw);
#ifdef FLEX_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bk_ph[4] += __builtin_amdgcn_s_memtime() - r0_;
#endif
    };

    // n steps of the stream into `acc`.  HOT: B rows from the panel buffer at LDS offset `base`; else from global memory.
    auto run = [&](auto hot_tag, float4 &acc, uint32_t n, uint32_t base) {
        constexpr bool HOT = decltype(hot_tag)::value;
        constexpr int U = HOT ? U_HOT : U_COLD;
        if (__builtin_expect(v.ablate & (HOT ? 2u : 4u), 0)) {  // timing-only: walk the stream, do nothing with it
            while (n) {
                if (wpos == wend) refill();
                const uint32_t m = min(n, wend - wpos);
                wpos += m;
                n -= m;
            }
            return;
        }
        while (n) {
            if (wpos == wend) refill();
            const uint32_t m = min(n, wend - wpos);
            const uint2 *wp = win + wpos * kBkSlots + slot;
            auto fetch = [&](uint32_t off) -> float4 {
                if constexpr (HOT) return *reinterpret_cast<const float4 *>(lds + (off + base));
                else return *reinterpret_cast<const float4 *>(Bb + (off + base));
            };
            uint32_t j = 0;
            for (; j + U <= m; j += U) {
                uint2 r[U];
                float4 b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = wp[(j + u) * kBkSlots];
#pragma unroll
                for (int u = 0; u < U; ++u) b[u] = fetch(r[u].x);
#pragma unroll
                for (int u = 0; u < U; ++u) fma4(acc, as_f32(r[u].y), b[u]);
            }
            if constexpr (!HOT) {
                if (j < m) {  // 1 .. U-1 steps left: one more full block on clamped records (a gather the row makes anyway) with value 0;
                              // no second set of arrays, which is what lets U_COLD be 10 inside the 128-register budget
                    uint2 r[U];
                    float4 b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) r[u] = wp[min(j + u, m - 1) * kBkSlots];
#pragma unroll
                    for (int u = 0; u < U; ++u) b[u] = fetch(r[u].x);
#pragma unroll
                    for (int u = 0; u < U; ++u) fma4(acc, j + u < m ? as_f32(r[u].y) : 0.f, b[u]);
                }
            } else if (j < m) {  // 1 .. U-1 steps left: wave-uniform branches, no LDS read that is not used (the panel phases are
                                 // sensitive to every extra LDS read: clamped full blocks measured 7 % slower, U_HOT = 8 slower still)
                const uint32_t rem = m - j;
                uint2 r[U - 1];
                float4 b[U - 1];
#pragma unroll
                for (int u = 0; u < U - 1; ++u)
                    if (static_cast<uint32_t>(u) < rem) r[u] = wp[(j + u) * kBkSlots];
#pragma unroll
                for (int u = 0; u < U - 1; ++u)
                    if (static_cast<uint32_t>(u) < rem) b[u] = fetch(r[u].x);
#pragma unroll
                for (int u = 0; u < U - 1; ++u)
                    if (static_cast<uint32_t>(u) < rem) fma4(acc, as_f32(r[u].y), b[u]);
            }
            wpos += m;
            n -= m;
        }
    };

    {
        const int c0 = t * 32 + l8 * 4;
        const bool col_ok = c0 < k;
        const uint32_t lane_goff = static_cast<uint32_t>(col_ok ? c0 : t * 32) * 4u;
        float4 acc[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        BK_STAMP(0);
        // cold phase: while the loader stages panel 0
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) run(std::false_type{}, acc[r], steps_of(r), lane_goff);
        BK_STAMP(1);
        consumer_barrier();
        BK_STAMP(3);
        for (uint32_t p = 0; p < np; ++p) {
            const uint32_t base = (p & 1) * kBkBufBytes + static_cast<uint32_t>(l8) * 16u;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) run(std::true_type{}, acc[r], steps_of((p + 1) * ROUNDS + r), base);
            BK_STAMP(2);
            consumer_barrier();
            BK_STAMP(3);
        }
        // a long row occupies 2 / 4 / 8 aligned slots: butterfly over the slots, then the first slot of each row stores.
        // A HUB row occupies whole groups on several waves: parts 1.. leave their sums in LDS (the panel buffers are free now),
        // one more barrier, and part 0 adds them in part order -- a fixed order, so the result is reproducible.
        auto butterfly = [&](float4 a, uint32_t vcode) -> float4 {
#pragma unroll
            for (int lvl = 1; lvl <= 3; ++lvl) {
                const int d = 4 << lvl;  // lanes between partner slots: 8, 16, 32
                const float px = __shfl_xor(a.x, d), py = __shfl_xor(a.y, d), pz = __shfl_xor(a.z, d), pq = __shfl_xor(a.w, d);
                if (vcode >= static_cast<uint32_t>(lvl)) {
                    a.x += px;
                    a.y += py;
                    a.z += pz;
                    a.w += pq;
                }
            }
            return a;
        };
        if (has_hubs) {  // workgroup-uniform
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t part = hub[r] & 0xFFu, ng = (hub[r] >> 8) & 0xFFu;
                if (ng > 1) {  // wave-uniform
                    acc[r] = butterfly(acc[r], 3u);
                    if (part > 0 && slot == 0)
                        *reinterpret_cast<float4 *>(lds + ((hub[r] >> 16) + part - 1) * kBkRowBytes + l8 * 16) = acc[r];
                }
            }
            consumer_barrier();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t part = hub[r] & 0xFFu, ng = (hub[r] >> 8) & 0xFFu;
                if (ng > 1 && part == 0) {
                    for (uint32_t q = 1; q < ng; ++q) {
                        const float4 o = *reinterpret_cast<const float4 *>(lds + ((hub[r] >> 16) + q - 1) * kBkRowBytes + l8 * 16);
                        acc[r].x += o.x;
                        acc[r].y += o.y;
                        acc[r].z += o.z;
                        acc[r].w += o.w;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t ng = (hub[r] >> 8) & 0xFFu;
            const uint32_t vcode = ng > 1 ? 0u : rows[r] >> 29, crow = rows[r] & kBkEmptyRow;  // a hub's group is summed already
            if (ng > 1 && (hub[r] & 0xFFu) != 0) continue;                                   // ... and only its part 0 stores
            float4 a = acc[r];
            if (__builtin_amdgcn_ballot_w64(vcode != 0) != 0) a = butterfly(a, vcode);  // wave-uniform; rare
            const bool first = ng > 1 ? slot == 0 : (slot & ((1u << vcode) - 1u)) == 0;
            if (crow != kBkEmptyRow && first && col_ok) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f val = {a.x, a.y, a.z, a.w};
                __builtin_nontemporal_store(val, reinterpret_cast<v4f *>(C + static_cast<uint64_t>(crow) * v.ldc + c0));
            }
        }
#ifdef FLEX_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BK_STAMP(5);
        bk_ph[6] = bk_last - bk_t0;
        bk_ph[7] = T;
        if (lane == 0 && v.trace != nullptr) {
            uint64_t *o = v.trace + ((static_cast<uint64_t>(blockIdx.y) * v.n_blocks + blk) * kBkWaves + w) * 8;
            for (int i = 0; i < 8; ++i) o[i] = bk_ph[i];
        }
#endif
    }
}

template <int ROUNDS>
int launch_rounds(const BlockView &bv, const float *dB, float *dC, hipStream_t s) {
    const uint32_t nblk = (bv.n_blocks + kXcds - 1) / kXcds * kXcds;
    // gathers in flight per wave in the cold phase: as many as the 128-register budget of a 1024-thread workgroup leaves
    // next to the accumulators (checked with -Rpass-analysis=kernel-resource-usage: no scratch)
    constexpr int kUCold = ROUNDS <= 4 ? 10 : 8;
#ifndef FLEX_BK_U_HOT  // experiment builds (make block_variants)
#define FLEX_BK_U_HOT 4
#endif
    hipLaunchKernelGGL((spmm_block_kernel<ROUNDS, FLEX_BK_U_HOT, kUCold>), dim3(nblk, (bv.k + 31) / 32), dim3(64 * (kBkWaves + 1)), 0, s, bv, dB, dC);
    FLEX_HIP_TRY(hipGetLastError());
    return FLEX_OK;
}

}  // namespace

int launch_blocks(const BlockView &bv, const float *dB, float *dC, hipStream_t s) {
    if (bv.n_blocks == 0) return FLEX_OK;
    switch (bv.rounds) {
        case 1: return launch_rounds<1>(bv, dB, dC, s);
        case 2: return launch_rounds<2>(bv, dB, dC, s);
        case 4: return launch_rounds<4>(bv, dB, dC, s);
        case 8: return launch_rounds<8>(bv, dB, dC, s);
        default: return FLEX_ERR_UNSUPPORTED;
    }
}

}  // namespace flex
