// Probe of v_permlane32_swap_b32 semantics on the device: prints which (register, lane) every result lane came from.
// hipcc --offload-arch=gfx950 -O3 -o swap_probe swap_probe.hip && ./swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = s[0];
    out[64 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    unsigned h[128];
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("first result : lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
    printf("second result: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
