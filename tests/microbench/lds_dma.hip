// Developer probe: semantics of global_load_lds_dwordx4 (direct global -> LDS copies without VGPRs) on gfx950.
// Each wave copies 1 KiB chunks: per-lane global address, wave-uniform LDS base (M0) + lane * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(const float4* __restrict__ g, int n4, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float4 l[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c * 64 < n4; c += 8) {
        const int i = min(c * 64 + lane, n4 - 1);
        __builtin_amdgcn_global_load_lds(g + i, l + c * 64, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) (and everything else)
    __syncthreads();
    for (int i = threadIdx.x; i < n4; i += 512) out[i] = l[i];
}
int main() {
    const int n4 = 77 * 26;      // 77 Gram rows of 26 float4
    std::vector<float> h(n4 * 4), r(n4 * 4);
    for (int i = 0; i < n4 * 4; ++i) h[i] = i * 0.5f;
    float4 *g, *o; hipMalloc(&g, n4 * 16); hipMalloc(&o, n4 * 16);
    hipMemcpy(g, h.data(), n4 * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), ((n4 + 63) / 64) * 64 * 16, 0, g, n4, o);
    hipMemcpy(r.data(), o, n4 * 16, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n4 * 4; ++i) bad += r[i] != h[i];
    printf("global_load_lds_dwordx4 copy: %d mismatches of %d\n", bad, n4 * 4);
    return bad != 0;
}
