// Developer microbenchmark: dependent-chain latency of the cross-lane / LDS primitives the
// per-problem kernels are built from (one wave, cycles per op from s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mvsmplfitting_amd/csrc/wave_ops.h"
using namespace mvfit;
#define N 256
__global__ void k(long long* out, float* fout, int nwaves_active) {
    __shared__ float lds[4096];
    __shared__ int idx[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-7f;
    for (int i = tid; i < 1024; i += blockDim.x) idx[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    float v = 1.0f + lane * 1e-3f;
    long long t0, t1;
    if (tid >= 64) {   // other waves: either idle at the barrier or (nwaves_active) hammer LDS
        if (nwaves_active) { float s = 0; for (int i = 0; i < 20000; ++i) s += lds[(tid * 7 + i * 13) & 4095]; fout[tid] = s; }
        return;
    }
    // 0: dependent fma chain
    t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) v = fmaf(v, 1.0000001f, 1e-9f);
    t1 = clock64(); if (lane == 0) out[0] = t1 - t0;
    // 1: dpp add chain (row16_sum has 4)
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) v = row16_sum(v) * 0.0624f;
    t1 = clock64(); if (lane == 0) out[1] = t1 - t0;
    // 2: wave64_sum chain (4 dpp + 2 swaps)
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) v = wave64_sum(v) * 0.0156f;
    t1 = clock64(); if (lane == 0) out[2] = t1 - t0;
    // 3: dependent LDS read chain (pointer chasing)
    int p = lane;
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) p = idx[p];
    t1 = clock64(); if (lane == 0) out[3] = t1 - t0;
    v += p;
    // 4: readlane + dependent use
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) v = lane_read(v, (i * 5) & 63) * 1.0001f + 0.1f;
    t1 = clock64(); if (lane == 0) out[4] = t1 - t0;
    // 5: ds_bpermute-based __shfl_xor chain
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) v += __shfl_xor(v, 1 + (i & 31), 64) * 1e-3f;
    t1 = clock64(); if (lane == 0) out[5] = t1 - t0;
    // 6: double wave64_sum
    double d = v;
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) d = wave64_sum(d) * 0.0156;
    t1 = clock64(); if (lane == 0) out[6] = t1 - t0;
    // 7: f64 divide chain
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) d = 1.0 / (d + 1.5);
    t1 = clock64(); if (lane == 0) out[7] = t1 - t0;
    // 8: f32 divide chain
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) v = 1.0f / (v + 1.5f);
    t1 = clock64(); if (lane == 0) out[8] = t1 - t0;
    // 9: global load dependent chain (L2-resident)
    const int* g = reinterpret_cast<const int*>(out + 64);
    int q = lane;
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) q = g[q];
    t1 = clock64(); if (lane == 0) out[9] = t1 - t0;
    // 10: LDS write then read by another lane (wave-level handoff)
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) { lds[lane] = v; wave_lds_fence(); v = lds[lane ^ 1] + 1.0f; wave_lds_fence(); }
    t1 = clock64(); if (lane == 0) out[10] = t1 - t0;
    // 11: s_barrier cost with the other waves gone: __syncthreads on a 1-wave remainder is invalid; skip
    // 12: sincosf
    t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) { float s, c; sincosf(v, &s, &c); v = s + c; }
    t1 = clock64(); if (lane == 0) out[12] = t1 - t0;
    // 13: one step of the L-BFGS walk as shipped: broadcast of a pivot lane through an SGPR + FMA
    float x = v, gq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) gq[u] = 1e-3f * (lane + u);
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) { const float pv = lane_read(x, (N - 1 - i) & 63); x = fmaf(-pv, gq[i & 7], x); }
    t1 = clock64(); if (lane == 0) out[13] = t1 - t0;
    // 14: the same step with the pivot broadcast through LDS (write own value, all lanes read one address)
    t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; ++i) { lds[lane] = x; wave_lds_fence(); const float pv = lds[(N - 1 - i) & 63]; x = fmaf(-pv, gq[i & 7], x); }
    t1 = clock64(); if (lane == 0) out[14] = t1 - t0;
    // 15: four steps per LDS round trip: pivots read together, carried through the block's triangle in uniform registers
    t0 = clock64();
#pragma unroll 2
    for (int i = 0; i < N; i += 4) {
        lds[lane] = x; wave_lds_fence();
        const int i0 = (N - 1 - i) & 63;
        const float a0 = lds[i0], b1 = lds[(i0 - 1) & 63], b2 = lds[(i0 - 2) & 63], b3 = lds[(i0 - 3) & 63];
        const float m01 = lds[64 + i0], m02 = lds[65 + i0], m03 = lds[66 + i0], m12 = lds[67 + i0], m13 = lds[68 + i0], m23 = lds[69 + i0];
        const float a1 = fmaf(-a0, m01, b1);
        const float a2 = fmaf(-a1, m12, fmaf(-a0, m02, b2));
        const float a3 = fmaf(-a2, m23, fmaf(-a1, m13, fmaf(-a0, m03, b3)));
        x = fmaf(-a0, gq[0], x); x = fmaf(-a1, gq[1], x); x = fmaf(-a2, gq[2], x); x = fmaf(-a3, gq[3], x);
    }
    t1 = clock64(); if (lane == 0) out[15] = t1 - t0;
    fout[tid] = v + (float)d + q + x;
}
int main() {
    long long* out; float* f;
    hipMalloc(&out, 64 * 8 + 4096 * 4); hipMalloc(&f, 4096);
    int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
    hipMemcpy(out + 64, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[16] = {"fma", "dpp add (x4 per row16_sum)", "wave64_sum f32", "lds read chain", "readlane+use", "shfl_xor (bpermute)",
                             "wave64_sum f64", "f64 divide", "f32 divide", "global load chain (L2)", "lds write->fence->read->fence", "", "sincosf",
                             "walk step: readlane + fma", "walk step: LDS broadcast + fma", "walk step, 4 per LDS round trip (per step)"};
    const int per[16] = {N, N, N / 4, N, N, N, N / 4, N / 4, N / 4, N / 4, N / 4, 1, N / 4, N, N, N};
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, f, mode); hipDeviceSynchronize(); }
        long long r[16]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        printf("mode %d (%s)\n", mode, mode ? "7 other waves hammering LDS" : "other waves exited");
        for (int i = 0; i < 16; ++i) if (names[i][0]) printf("  %-34s %8.1f cycles/op\n", names[i], (double)r[i] / per[i]);
    }
    return 0;
}
