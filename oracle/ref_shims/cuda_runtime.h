// TEST INFRASTRUCTURE - host stand-in for <cuda_runtime.h>: the CUDA qualifiers disappear, the thread
// coordinates become ordinary variables the driver sets before each call, max() is fmax (what CUDA's
// device overload is), the error API is a no-op.  Used only to build oracle/_ref/libsdf_ref.so.
#pragma once
#include <cmath>
#include <cstdio>

#define __device__
#define __global__
#define __host__

struct ref_uint3 { unsigned x = 0, y = 0, z = 0; };
static thread_local ref_uint3 blockIdx, threadIdx;
static thread_local ref_uint3 blockDim;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned long long a = 1, unsigned b = 1, unsigned c = 1) : x((unsigned)a), y(b), z(c) {}
};

static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double max(double a, float b) { return fmax(a, (double)b); }

typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return ""; }
