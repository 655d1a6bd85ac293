// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 element e at byte 2e; lane l passes byte address 8 l
// (its own 4-element slot); prints the 4 elements every lane receives.
// hipcc --offload-arch=gfx950 -O3 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
__global__ void k(uint32_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64)
        lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + threadIdx.x * stride_bytes;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 2] = v[0];
    out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
    uint32_t* d;
    (void)hipMalloc(&d, 512);
    for (int stride : {8, 32}) {
        k<<<1, 64>>>(d, stride);
        uint32_t h[128];
        (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("lane address = %d * lane bytes (element index = byte / 2)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4u %4u %4u %4u", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
