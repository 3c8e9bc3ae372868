// Host build of csrc/lapack_svd3.h for tests/test_umeyama.py (g++ -O2 -shared -fPIC, no HIP): the shipped device
// code is plain scalar C++, so its arithmetic can be checked against np.linalg.svd without a GPU.
#include "../mvsmplfitting_amd/csrc/lapack_svd3.h"
extern "C" void svd3_batch(const double* A, double* U, double* S, double* Vh, int n) {
    for (int i = 0; i < n; ++i) mvfit::lapack3::svd3(A + 9 * i, U + 9 * i, S + 3 * i, Vh + 9 * i);
}
