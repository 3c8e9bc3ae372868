// TEST INFRASTRUCTURE - host stand-in for <ATen/ATen.h>, just enough for the reference's
// sdf/sdf/csrc/sdf_cuda_kernel.cu to compile UNMODIFIED with g++ (see oracle/sdf_ref_driver.cpp).
// The launcher at the bottom of that file (sdf_cuda, :307-335) is compiled but never called: its
// AT_DISPATCH_FLOATING_TYPES(...) body (the <<<...>>> launch) is swallowed by the macro below and the
// driver runs the kernel function itself over the launcher's grid.
#pragma once
#include <cstdint>
#include <cstdio>

namespace at {
struct Tensor {
    long dims[4] = {0, 0, 0, 0};
    long size(int i) const { return dims[i]; }
    int type() const { return 0; }
};
}  // namespace at

#define AT_DISPATCH_FLOATING_TYPES(...)
