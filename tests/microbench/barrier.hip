// Developer microbenchmark: cost of __syncthreads() for a 512-thread workgroup (8 waves, 2 per SIMD),
// alone and with a little LDS traffic between barriers; and of a phase in which only one wave works.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(long long* out, float* fout) {
    __shared__ float lds[2048];
    const int tid = threadIdx.x;
    lds[tid] = tid; lds[tid + 512] = 1.f; lds[tid + 1024] = 2.f; lds[tid + 1536] = 3.f;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < 256; ++i) __syncthreads();
    long long t1 = clock64();
    if (tid == 0) out[0] = (t1 - t0) / 256;
    float v = 0.f;
    t0 = clock64();
    for (int i = 0; i < 256; ++i) { lds[tid] = v + i; __syncthreads(); v += lds[(tid * 7 + i) & 511]; __syncthreads(); }
    t1 = clock64();
    if (tid == 0) out[1] = (t1 - t0) / 256;
    // one-wave phase between barriers: 40 dependent FMAs on wave 0
    t0 = clock64();
    for (int i = 0; i < 256; ++i) {
        if (tid < 64) { float a = v; for (int j = 0; j < 40; ++j) a = fmaf(a, 1.0001f, 0.5f); v = a; }
        __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) out[2] = (t1 - t0) / 256;
    // all threads: 40 dependent FMAs + barrier
    t0 = clock64();
    for (int i = 0; i < 256; ++i) {
        { float a = v; for (int j = 0; j < 40; ++j) a = fmaf(a, 1.0001f, 0.5f); v = a; }
        __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) out[3] = (t1 - t0) / 256;
    // all threads: 40 independent LDS reads + barrier
    t0 = clock64();
    for (int i = 0; i < 256; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 40; ++j) a += lds[(tid + j * 33 + i) & 2047];
        v += a;
        __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) out[4] = (t1 - t0) / 256;
    // all threads: 10 dependent LDS reads (pointer chase) + barrier
    t0 = clock64();
    for (int i = 0; i < 256; ++i) {
        int p = tid;
        for (int j = 0; j < 10; ++j) p = ((int)lds[p & 511] + j) & 511;
        v += p;
        __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) out[5] = (t1 - t0) / 256;
    // f32 IEEE divide x10 dependent, all threads
    t0 = clock64();
    for (int i = 0; i < 256; ++i) {
        float a = v + 2.f;
        for (int j = 0; j < 10; ++j) a = 3.0f / (a + 1.5f);
        v = a;
        __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) out[6] = (t1 - t0) / 256;
    fout[tid] = v;
}
int main() {
    long long* out; float* f;
    hipMalloc(&out, 64 * 8); hipMalloc(&f, 4096 * 4);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(32), dim3(512), 0, 0, out, f); hipDeviceSynchronize(); }
    long long r[8]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    const char* n[7] = {"bare __syncthreads", "lds write + barrier + lds read + barrier", "40 dep FMA on wave 0 only + barrier", "40 dep FMA all waves + barrier",
                        "40 indep LDS reads all waves + barrier", "10 dependent LDS reads all waves + barrier", "10 dependent f32 divides all waves + barrier"};
    for (int i = 0; i < 7; ++i) printf("%-48s %6lld cycles\n", n[i], r[i]);
    return 0;
}
