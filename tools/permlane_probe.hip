#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    v2u q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y; o[128 + threadIdx.x] = q.x; o[192 + threadIdx.x] = q.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 64; ++i) printf("%u ", h[j * 64 + i]); printf("\n"); }
}
