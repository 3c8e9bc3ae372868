#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {   // A[32][16], B[16][32], C[32][32]
    const int l = threadIdx.x;
    half8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)A[(l % 32) * 16 + 8 * (l / 32) + t]; b[t] = (_Float16)B[(8 * (l / 32) + t) * 32 + (l % 32)]; }
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); C[row * 32 + (l & 31)] = acc[r]; }
}
int main() {
    std::vector<float> A(512), B(512), C(1024), R(1024, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 13 - 6); B[i] = (float)((i * 5) % 11 - 5); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int kk = 0; kk < 16; ++kk) R[m * 32 + n] += A[m * 16 + kk] * B[kk * 32 + n];
    float *dA, *dB, *dC; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("mfma_f32_32x32x16_f16 layout check: %d mismatches of 1024\n", bad);
    return bad != 0;
}
