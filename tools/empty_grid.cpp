// tools/empty_grid.cpp -- launch floor of an empty kernel (and of one 16-byte load + store per lane) over grid shapes: waves x waves per
// workgroup.  hipcc --offload-arch=gfx950 -O2 -o empty_grid tools/empty_grid.cpp  (-> profiles/r03_empty_grid_floor.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(float *p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
template <int T> __global__ __launch_bounds__(T) void touch_kernel(const float4 *in, float4 *out, int n) {
    int i = blockIdx.x * T + threadIdx.x;
    if (i < n) { float4 v = in[i]; v.x += 1.f; out[i] = v; }
}
static float time_launch(void (*launch)(hipStream_t), int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch(nullptr);
    hipEventRecord(a, nullptr);
    for (int i = 0; i < n; ++i) launch(nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / n;
}
static int g_grid, g_block; static float4 *g_in, *g_out;
int main() {
    hipMalloc(&g_in, 64 << 20); hipMalloc(&g_out, 64 << 20); hipMemset(g_in, 0, 64 << 20);
    const int waves[] = {1024, 4096, 5563, 16384, 22252};
    for (int w : waves)
        for (int wpb : {1, 4, 8, 16}) {
            g_grid = (w + wpb - 1) / wpb; g_block = 64 * wpb;
            float e = time_launch([](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(g_grid), dim3(g_block), 0, s, (float *)nullptr); }, 2000);
            float t = 0;
            auto tl = [](hipStream_t s) {
                int n = g_grid * g_block;
                switch (g_block) {
                    case 64: hipLaunchKernelGGL(touch_kernel<64>, dim3(g_grid), dim3(64), 0, s, g_in, g_out, n); break;
                    case 256: hipLaunchKernelGGL(touch_kernel<256>, dim3(g_grid), dim3(256), 0, s, g_in, g_out, n); break;
                    case 512: hipLaunchKernelGGL(touch_kernel<512>, dim3(g_grid), dim3(512), 0, s, g_in, g_out, n); break;
                    default: hipLaunchKernelGGL(touch_kernel<1024>, dim3(g_grid), dim3(1024), 0, s, g_in, g_out, n); break;
                }
            };
            t = time_launch(tl, 2000);
            std::printf("waves %6d  waves/WG %2d  WGs %6d : empty %.2f us   load+store 16 B/lane %.2f us\n", w, wpb, g_grid, e, t);
        }
    return 0;
}
