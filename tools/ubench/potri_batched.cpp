// timing of rocSOLVER strided-batched Cholesky + inverse for the block sizes of the BA block-Jacobi preconditioner
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <chrono>
int main()
{
    rocblas_handle h; rocblas_create_handle(&h);
    const int cfg[][2] = {{48, 1}, {96, 1}, {192, 1}, {48, 13}, {384, 19}, {384, 780}, {192, 38}, {192, 1560}, {96, 75}, {96, 3125}, {48, 150}, {48, 6250}};
    for (auto& c : cfg) {
        const int n = c[0], nb = c[1];
        std::vector<double> A((size_t)n * n * nb);
        for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
            A[(size_t)b * n * n + i + (size_t)j * n] = (i == j ? n + 1.0 : 1.0 / (1 + abs(i - j)));
        double* d; int* info; hipMalloc(&d, A.size() * 8); hipMalloc(&info, nb * 4);
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpy(d, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            rocsolver_dpotrf_strided_batched(h, rocblas_fill_lower, n, d, n, (rocblas_stride)n * n, info, nb);
            hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            rocsolver_dpotri_strided_batched(h, rocblas_fill_lower, n, d, n, (rocblas_stride)n * n, info, nb);
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            if (rep == 2) {
                std::vector<double> R(A.size()); std::vector<int> hi(nb);
                hipMemcpy(R.data(), d, A.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), info, nb * 4, hipMemcpyDeviceToHost);
                double worst = 0; int bad = 0;
                for (int b = 0; b < nb; b += (nb > 4 ? nb / 4 : 1)) {
                    bad += hi[b] != 0;
                    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
                        double acc = 0;
                        for (int k = 0; k < n; k++) { const double inv_kj = k >= j ? R[(size_t)b * n * n + k + (size_t)j * n] : R[(size_t)b * n * n + j + (size_t)k * n]; acc += A[(size_t)b * n * n + i + (size_t)k * n] * inv_kj; }
                        const double e = fabs(acc - (i == j ? 1.0 : 0.0)); if (e > worst) worst = e;
                    }
                }
                printf("n=%d batch=%d |A inv(A) - I|max %.2e info!=0: %d  ", n, nb, worst, bad);
            }
            if (rep == 2) printf("n=%d batch=%d potrf %.3f ms potri %.3f ms\n", n, nb, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
        hipFree(d); hipFree(info);
    }
    return 0;
}
