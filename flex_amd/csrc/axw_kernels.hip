// axw_kernels.hip -- the dense half of the A*X*W layer product (libflex_axw.so):
//   Out[n x cp] = L[n x dim] * Wp[dim x cp], row-major fp32, n tall (10^5..10^6), dim and cp small (<= 128).
// A tall-skinny GEMM is bound by streaming L once (4*n*dim bytes) and by the fp32 MFMA rate, which on gfx950
// equals the fp32 vector rate (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, 64-cycle issue and dependent latency),
// so four independent 32x32 accumulators per wave saturate the pipe with ONE wave per SIMD:
//   * one workgroup (4 waves) per CU, persistent over 32-row panels; Wp (<= 64 KiB) sits in LDS for the whole
//     launch, each wave stages its own 32 x dim panel of L through a private LDS tile (coalesced 16-byte loads in,
//     4-byte MFMA operands out), so there is no barrier after the first;
//   * A operand: lane l holds L[row l&31][k = 2kk + (l>>5)], B operand: Wp[k = 2kk + (l>>5)][col l&31]
//     (cdna guide, 'FP32-input MFMA'); C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5);
//   * the result is bit-for-bit a k-ordered fmaf chain per output element.
// Replaces rocBLAS SGEMM for dim <= 256 (n x 128 x 128: flickr shape 37-42 us against rocBLAS 37-65, reddit shape
// 79-85 us against 186-207).  On the flickr shape the matrix pipe is 63 % busy (SQ_VALU_MFMA_BUSY_CYCLES): the clock
// is ~2.0 GHz under this load, the busiest SIMDs carry 3 panels against 2.7 on average, and ~8 us of launch, W staging,
// first fetch and last stores overlap nothing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

namespace flex_axw_detail {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // a plain vector type: HIP's float4 (a struct of unions) kept the prefetch array in scratch

constexpr int kRowsPerWave = 32;
constexpr int kPad = 4;  // floats: keeps 16-byte alignment of a tile row for ds_write_b128

// Order of the k sum: MFMA step s takes two k values, one per wave half h = lane>>5.  Steps 2m and 2m+1 use
// k = 4m+2h and 4m+2h+1, so a lane's A operands of both steps are ONE 8-byte LDS read (L[row][4m+2h .. +1]) and,
// with W stored in LDS as k-pairs ([k/2][col][k&1]), so are its B operands.
// Every panel is a FULL 32 rows: the last one starts at n-32 and overlaps its predecessor (both write the same
// values to the shared rows), so neither the loads nor the stores carry a row guard.  Requires n >= 32.
// A panel is staged in k-slabs of 64 floats (8.5 KiB per wave), which is what lets EIGHT waves share one copy of W
// in LDS: two waves per SIMD, so one wave's staging and stores hide behind the other's MFMAs.  Waves w and w+4 of
// a workgroup share a SIMD; panels are dealt so that the second wave of a SIMD gets work only after every SIMD
// has one (a short launch otherwise loads some SIMDs with 4 panels and others with 2).
constexpr int kSlab = 64;                    // floats of k per staged slab
constexpr int kSlabLoads = kSlab / 8;        // float4 per lane and slab: 32 rows x 16 float4 / 64 lanes
constexpr int kWaves = 8;

template <int NT>  // 32-column output tiles per wave and pass (1..4)
__global__ __launch_bounds__(64 * kWaves) void axw_gemm_kernel(const float *__restrict__ L, const float *__restrict__ Wp,
                                                               float *__restrict__ Out, int n, int dim, int cp, int col0) {
    extern __shared__ float smem[];
    constexpr int WC = 32 * NT;          // columns of this pass
    constexpr int xs = kSlab + kPad;     // floats per staged row
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *sW = smem;                    // [dim/2][WC][2]
    float *sX = smem + dim * WC + wave * (kRowsPerWave * xs);

    // Wp[:, col0 : col0+WC] -> LDS as k-pairs, once per workgroup.  The slice is at most 64 KiB = 8 float4 per thread:
    // all eight loads go out together (a loop of load -> wait -> write costs eight memory round trips before the
    // first MFMA), slots past the slice re-read its last float4 and are not written.
    {
        const int total = dim * (WC / 4);
        f32x4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = min(static_cast<int>(threadIdx.x) + j * static_cast<int>(blockDim.x), total - 1);
            w[j] = *reinterpret_cast<const f32x4 *>(Wp + static_cast<size_t>(i / (WC / 4)) * cp + col0 + (i % (WC / 4)) * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = static_cast<int>(threadIdx.x) + j * static_cast<int>(blockDim.x);
            if (i < total) {
                const int kr = i / (WC / 4), c4 = i % (WC / 4);
                float *d = sW + (kr >> 1) * (2 * WC) + (c4 * 4) * 2 + (kr & 1);
                d[0] = w[j][0]; d[2] = w[j][1]; d[4] = w[j][2]; d[6] = w[j][3];
            }
        }
    }

    // this lane's 8 float4 of a slab: row and column inside the slab (fixed), LDS slot (fixed)
    int row_i[kSlabLoads], col_i[kSlabLoads], lds_off[kSlabLoads];
#pragma unroll
    for (int i = 0; i < kSlabLoads; ++i) {
        const int f = i * 64 + lane;
        row_i[i] = f / (kSlab / 4);
        col_i[i] = (f % (kSlab / 4)) * 4;
        lds_off[i] = row_i[i] * xs + col_i[i];
    }
    const int n_panels = (n + kRowsPerWave - 1) / kRowsPerWave;
    const int n_slabs = (dim + kSlab - 1) / kSlab;
    auto panel_row = [&](int pnl) { return min(min(pnl, n_panels - 1) * kRowsPerWave, n - kRowsPerWave); };
    // unit u = (panel, slab) in the order this wave meets them; loads of a slab's tail past dim re-read the row's last float4
    const int first_panel = wave < 4 ? blockIdx.x * 4 + wave : gridDim.x * 4 + blockIdx.x * 4 + (wave - 4);
    const int panel_step = gridDim.x * kWaves;
    f32x4 pre[kSlabLoads];  // the NEXT slab, in flight while this one is multiplied
    auto fetch = [&](int pnl, int slab) {
        const float *src = L + static_cast<size_t>(panel_row(pnl)) * dim;
        const int k0 = slab * kSlab;
#pragma unroll
        for (int i = 0; i < kSlabLoads; ++i)
            pre[i] = *reinterpret_cast<const f32x4 *>(src + static_cast<size_t>(row_i[i]) * dim + min(k0 + col_i[i], dim - 4));
    };
    fetch(first_panel, 0);
    __syncthreads();  // Wp staged (the only barrier: the panel tiles are private to a wave)

    const float *ap = sX + (lane & 31) * xs + 2 * (lane >> 5);
    const float *bp = sW + (lane >> 5) * (2 * WC) + (lane & 31) * 2;
    for (int panel = first_panel; panel < n_panels; panel += panel_step) {
        const int r0 = panel_row(panel);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int slab = 0; slab < n_slabs; ++slab) {
#pragma unroll
            for (int i = 0; i < kSlabLoads; ++i) *reinterpret_cast<f32x4 *>(sX + lds_off[i]) = pre[i];
            if (slab + 1 < n_slabs) fetch(panel, slab + 1);  // wave-uniform
            else fetch(panel + panel_step, 0);
            const int d4 = min(kSlab, dim - slab * kSlab) / 4;  // k-quads in this slab
            const float *bs = bp + (slab * (kSlab / 2)) * (2 * WC);
            // operands of step pair m+1 are read from the LDS while the 2*NT MFMAs of pair m issue; the sched_barriers
            // keep the reads ABOVE the MFMAs they overlap with (the scheduler sinks them next to their use otherwise)
            auto lda = [&](int m) { return *reinterpret_cast<const float2 *>(ap + 4 * m); };
            auto ldb = [&](int m, int t) { return *reinterpret_cast<const float2 *>(bs + (2 * m) * (2 * WC) + t * 64); };
            float2 a0 = lda(0), b0[NT], a1, b1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b0[t] = ldb(0, t);
            int m = 0;
            for (; m + 2 <= d4; m += 2) {
                a1 = lda(m + 1);
#pragma unroll
                for (int t = 0; t < NT; ++t) b1[t] = ldb(m + 1, t);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0[t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0[t].y, acc[t], 0, 0, 0);
                const int mn = min(m + 2, d4 - 1);  // the last pair re-reads a valid slot instead of branching
                a0 = lda(mn);
#pragma unroll
                for (int t = 0; t < NT; ++t) b0[t] = ldb(mn, t);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1[t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1[t].y, acc[t], 0, 0, 0);
            }
            if (m < d4) {  // odd number of k-quads: one pair left, already in a0 / b0
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0[t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0[t].y, acc[t], 0, 0, 0);
            }
        }
        // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        float *dst = Out + static_cast<size_t>(r0 + 4 * (lane >> 5)) * cp + col0 + (lane & 31);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                __builtin_nontemporal_store(acc[t][reg], dst + static_cast<size_t>((reg & 3) + 8 * (reg >> 2)) * cp + t * 32);
    }
}

}  // namespace flex_axw_detail

// Out[n x cp] = L[n x dim] * Wp[dim x cp].  Requires n >= 32, dim % 4 == 0, dim <= 256, cp % 32 == 0.  Returns hipSuccess or
// the launch error.  The pass width is what keeps Wp's slice within 64 KiB of LDS: 128 columns up to dim = 128, 64 beyond.
extern "C" hipError_t flex_axw_gemm_launch(const float *L, const float *Wp, float *Out, int n, int dim, int cp, int n_cus, hipStream_t s) {
    using namespace flex_axw_detail;
    if (n <= 0) return hipSuccess;
    const int max_cols = dim <= 128 ? 128 : 64;
    for (int col0 = 0; col0 < cp; col0 += max_cols) {
        const int nt = std::min(max_cols, cp - col0) / 32;
        const size_t lds = (static_cast<size_t>(dim) * 32 * nt + kWaves * kRowsPerWave * (kSlab + kPad)) * sizeof(float);
        const dim3 grid(static_cast<unsigned>(n_cus)), block(64 * kWaves);
        hipError_t e = hipSuccess;
        auto go = [&](auto kernel) {  // more than 64 KiB of dynamic LDS has to be asked for
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
            if (e != hipSuccess) return;
            hipLaunchKernelGGL(kernel, grid, block, lds, s, L, Wp, Out, n, dim, cp, col0);
            e = hipGetLastError();
        };
        switch (nt) {
            case 4: go(axw_gemm_kernel<4>); break;
            case 3: go(axw_gemm_kernel<3>); break;
            case 2: go(axw_gemm_kernel<2>); break;
            default: go(axw_gemm_kernel<1>); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
