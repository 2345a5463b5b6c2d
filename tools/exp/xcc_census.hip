// census: which XCD does block b run on?  (HW_REG_XCC_ID via s_getreg)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void census(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
}
int main() {
    const int nb = 4096;
    unsigned *d; hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d);
    std::vector<unsigned> h(nb); hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0}; int rr = 0;
    for (int i = 0; i < nb; ++i) { cnt[h[i] & 15]++; if ((h[i] & 15) == (h[0] + i) % 8) rr++; }
    printf("first 24:"); for (int i = 0; i < 24; ++i) printf(" %u", h[i]); printf("\n");
    printf("per xcc:"); for (int i = 0; i < 16; ++i) printf(" %d", cnt[i]); printf("\nround-robin matches: %d of %d\n", rr, nb);
    return 0;
}
