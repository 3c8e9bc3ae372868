// Developer microbenchmark: one triangular recurrence of the L-BFGS direction (lbfgs_device.h:lb_recur_loop)
// on wave 0 of a 512-thread workgroup, Gram rows cold in L2 (written by the previous launch) vs staged in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mvsmplfitting_amd/csrc/wave_ops.h"
using namespace mvfit;
constexpr int GS = 104, ROWS = 264;
template <int PD, bool LDSM>
__global__ __launch_bounds__(512) void k(const float* __restrict__ Mg, int n, long long* out, float* fout) {
    __shared__ float Ml[ROWS * GS / 2];
    const int tid = threadIdx.x, lane = tid & 63;
    long long t0 = clock64();
    if (LDSM) {
        for (int i = tid; i < (n + 2 * PD) * GS / 4; i += 512) reinterpret_cast<float4*>(Ml)[i] = reinterpret_cast<const float4*>(Mg + 16 * GS)[i];
    }
    __syncthreads();
    long long t1 = clock64();
    if (tid < 64) {
        float x0 = lane < n ? 1.0f + lane * 1e-3f : 0.f;
        const float* M = LDSM ? Ml : Mg + 16 * GS;
        const float* p0 = M + PD * GS + (lane < n ? lane : 100);
        float g0[PD];
#pragma unroll
        for (int u = 0; u < PD; ++u) g0[u] = p0[u * GS];
        for (int base = 0; base < n; base += PD) {
            p0 += PD * GS;
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int i = (base + u) & 127;
                const float v = lane_read(x0, i & 63);
                x0 = fmaf(-v, g0[u], x0);
                g0[u] = p0[u * GS];
            }
        }
        long long t2 = clock64();
        fout[lane] = x0;
        if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
    }
}
__global__ void fill(float* M) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < ROWS * GS) M[i] = ((i % GS) > (i / GS) % 100) ? 1e-3f * ((i * 7) % 13) : 0.f; }
int main() {
    float *M, *f; long long* o; hipMalloc(&M, ROWS * GS * 4 * 64); hipMalloc(&f, 256); hipMalloc(&o, 64);
    long long h[2];
    for (int n : {24, 45, 90}) {
        for (int var = 0; var < 4; ++var) {
            double a0 = 0, a1 = 0;
            for (int rep = 0; rep < 20; ++rep) {
                float* Mr = M + (size_t)rep * ROWS * GS;
                hipLaunchKernelGGL(fill, dim3((ROWS * GS + 255) / 256), dim3(256), 0, 0, Mr);
                if (var == 0) hipLaunchKernelGGL((k<8, false>), dim3(1), dim3(512), 0, 0, Mr, n, o, f);
                if (var == 1) hipLaunchKernelGGL((k<16, false>), dim3(1), dim3(512), 0, 0, Mr, n, o, f);
                if (var == 2) hipLaunchKernelGGL((k<32, false>), dim3(1), dim3(512), 0, 0, Mr, n, o, f);
                if (var == 3) hipLaunchKernelGGL((k<8, true>), dim3(1), dim3(512), 0, 0, Mr, n, o, f);
                hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
                if (rep >= 4) { a0 += h[0]; a1 += h[1]; }
            }
            const char* nm[] = {"global PD=8", "global PD=16", "global PD=32", "LDS staged PD=8"};
            printf("n=%2d %-16s stage %7.0f  recurrence %7.0f cycles (%.1f per step)\n", n, nm[var], a0 / 16, a1 / 16, a1 / 16 / n);
        }
    }
    return 0;
}
