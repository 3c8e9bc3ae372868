// Developer microbenchmark: how fast can ONE workgroup per CU stream an L2-resident 186 KB matrix
// (the objective-vertex basis) through its CU?  Variants: grid size, waves, access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(const float4* __restrict__ src, int n4, int reps, float* out, long long* cyc, int stagger) {
    const int tid = threadIdx.x, nt = blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        const int off = stagger ? ((blockIdx.x * 1237 + r * 331) % 16) * (n4 / 16) : 0;
        float4 v[24];
        int base = tid;
#pragma unroll
        for (int i = 0; i < 24; ++i) { int idx = base + i * nt + off; idx = idx >= n4 ? idx - n4 : idx; v[i] = src[idx]; }
#pragma unroll
        for (int i = 0; i < 24; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        __syncthreads();
    }
    const long long t1 = clock64();
    out[blockIdx.x * nt + tid] = acc.x + acc.y + acc.z + acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int n4 = 24 * 512;               // 12288 float4 = 196,608 B
    float4* src; float* out; long long* cyc;
    hipMalloc(&src, n4 * 16); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8);
    hipMemset(src, 0, n4 * 16);
    const int reps = 200;
    for (int nt : {512, 256}) for (int grid : {1, 8, 32, 256}) for (int stagger : {0, 1}) {
        const int per_rep = 24 * nt * 16;
        for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(k, dim3(grid), dim3(nt), 0, 0, src, n4, reps, out, cyc, stagger); hipDeviceSynchronize(); }
        long long h[512]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double mx = 0, av = 0; for (int i = 0; i < grid; ++i) { av += h[i]; if (h[i] > mx) mx = h[i]; } av /= grid;
        printf("threads %3d grid %3d stagger %d: %.0f cycles per %d-byte pass (avg), %.1f B/clk/CU (slowest wg %.1f)\n", nt, grid, stagger,
               av / reps, per_rep, per_rep * (double)reps / av, per_rep * (double)reps / mx);
    }
    return 0;
}
