// what does device memory cost to allocate?  (the first hsk_count of a process grows its pools: tens of GB)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    (void)hipFree(nullptr);
    const size_t sizes[] = {256ull << 20, 1ull << 30, 4ull << 30, 16ull << 30, 32ull << 30};
    for (int rep = 0; rep < 2; ++rep)
        for (size_t sz : sizes) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, sz);
            double t1 = now();
            (void)hipMemset(p, 1, sz); (void)hipDeviceSynchronize();
            double t2 = now();
            (void)hipMemset(p, 2, sz); (void)hipDeviceSynchronize();
            double t3 = now();
            (void)hipFree(p);
            double t4 = now();
            printf("rep %d  %6.2f GB: hipMalloc %9.2f ms (%s)  first memset %8.2f ms  second memset %8.2f ms  hipFree %8.2f ms\n", rep, sz / 1073741824.0, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2, t4 - t3);
        }
    // many blocks kept alive (a pool growing): 64 x 2 GB
    std::vector<void *> v; double t0 = now();
    for (int i = 0; i < 64; ++i) { void *p = nullptr; if (hipMalloc(&p, 2ull << 30) != hipSuccess) break; v.push_back(p); }
    double t1 = now();
    printf("%zu x 2 GB kept: %9.2f ms in all (%.2f ms per GB)\n", v.size(), t1 - t0, (t1 - t0) / (2.0 * v.size()));
    for (void *p : v) (void)hipFree(p);
    double t2 = now(); printf("freed in %9.2f ms\n", t2 - t1);
    t0 = now(); { void *p = nullptr; (void)hipMalloc(&p, 128ull << 30); t1 = now(); printf("one 128 GB block: %9.2f ms\n", t1 - t0); (void)hipFree(p); }
    return 0;
}
