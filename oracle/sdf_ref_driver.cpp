// TEST INFRASTRUCTURE - builds the REFERENCE's own SDF voxelisation kernel for the host.
//
// The kernel source is compiled from where it lies (REF_SDF_KERNEL_CU, normally
// /root/reference/sdf/sdf/csrc/sdf_cuda_kernel.cu), unmodified and un-copied: the three headers it includes
// (<ATen/ATen.h>, <cuda.h>, <cuda_runtime.h>) resolve to the stand-ins under oracle/ref_shims/, which turn
// `__global__ void sdf_cuda_kernel<scalar_t>(...)` (:242-304) and its `__device__` helpers (:21-237) into plain
// C++ templates.  This file then plays the launcher (:307-335): same thread geometry (512 threads per block,
// blocks = B*G^3 / 512 with the reference's INTEGER division, so a tail that does not fill a block stays at the
// caller's zeros exactly as with the CUDA launch), one call of the kernel function per (block, thread).
//
// Built with -O2 -ffp-contract=off: no FMA fusion (nvcc may fuse in device code; which products it fuses is a
// property of its code generator, not of the source - this is the source's arithmetic as written).
// Output goes to oracle/_ref/ only (git-ignored); nothing of the reference is copied into the repository.
#include <cstdint>
#include <cstring>

#include REF_SDF_KERNEL_CU

extern "C" {

// phi[B,G,G,G] (caller-initialised, like sdf.py:22's torch.zeros), faces[num_faces,3], vertices[B,Nv,3].
// all_voxels != 0: also run the partial last block the reference's launch drops.
int ref_sdf_f32(float* phi, const int32_t* faces, const float* vertices, int batch_size, int num_faces,
                int num_vertices, int grid_size, int all_voxels) {
    const int threads = 512;                                                        // :316
    const long total = (long)batch_size * grid_size * grid_size * grid_size;
    long blocks = total / threads;                                                  // :317
    if (all_voxels && blocks * threads < total) blocks += 1;
    blockDim.x = threads;
    for (long b = 0; b < blocks; ++b) {
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < threads; ++t) {
            threadIdx.x = (unsigned)t;
            sdf_cuda_kernel<float>(phi, faces, vertices, batch_size, num_faces, num_vertices, grid_size);
        }
    }
    return 0;
}

int ref_sdf_f64(double* phi, const int32_t* faces, const double* vertices, int batch_size, int num_faces,
                int num_vertices, int grid_size, int all_voxels) {
    const int threads = 512;
    const long total = (long)batch_size * grid_size * grid_size * grid_size;
    long blocks = total / threads;
    if (all_voxels && blocks * threads < total) blocks += 1;
    blockDim.x = threads;
    for (long b = 0; b < blocks; ++b) {
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < threads; ++t) {
            threadIdx.x = (unsigned)t;
            sdf_cuda_kernel<double>(phi, faces, vertices, batch_size, num_faces, num_vertices, grid_size);
        }
    }
    return 0;
}

}  // extern "C"
