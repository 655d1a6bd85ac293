// Probe of v_permlane16_swap_b32 on the device, and of the 4 x 4 transposition between lane groups and dwords that the blocksize-32
// instances of gemm4_mfma_rt_kernel build from it (two v_permlane32_swap on the dword pairs (0, 2), (1, 3), then two
// v_permlane16_swap on the pairs (0, 1), (2, 3)): with X[g][d] = 100 g + d in lane group g (16 lanes) and dword d, the result must be
// Y[g][j] = X[j][g] = 100 j + g. Exit code 1 on any mismatch.
// hipcc --offload-arch=gfx950 -O3 -o swap16_probe swap16_probe.hip && ./swap16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    {
        unsigned a = lane, b = 100 + lane;
        auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        out[lane] = s[0];
        out[64 + lane] = s[1];
    }
    unsigned w[4];
    for (int d = 0; d < 4; ++d)
        w[d] = 100 * (lane >> 4) + d;
    auto s02 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
    w[0] = s02[0]; w[2] = s02[1]; w[1] = s13[0]; w[3] = s13[1];
    auto t01 = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
    auto t23 = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
    w[0] = t01[0]; w[1] = t01[1]; w[2] = t23[0]; w[3] = t23[1];
    for (int d = 0; d < 4; ++d)
        out[128 + 4 * lane + d] = w[d];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4 * (128 + 256));
    k<<<1, 64>>>(d);
    unsigned h[128 + 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("permlane16_swap(a = lane, b = 100 + lane)\n first result : lane0=%u lane15=%u lane16=%u lane31=%u lane32=%u lane47=%u lane48=%u lane63=%u\n", h[0], h[15], h[16],
           h[31], h[32], h[47], h[48], h[63]);
    printf(" second result: lane0=%u lane15=%u lane16=%u lane31=%u lane32=%u lane47=%u lane48=%u lane63=%u\n", h[64], h[79], h[80], h[95], h[96], h[111],
           h[112], h[127]);
    int bad = 0;
    // expected: the first operand's lanes 16-31 / 48-63 hold the second's lanes 0-15 / 32-47 and vice versa
    for (int l = 0; l < 64; ++l) {
        const bool odd_row = (l >> 4) & 1;
        bad += h[l] != (odd_row ? 100u + (l - 16) : (unsigned)l);
        bad += h[64 + l] != (odd_row ? 100u + l : (unsigned)(l + 16));
    }
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
            bad += h[128 + 4 * l + j] != 100u * j + (l >> 4);
    printf("%s: %d mismatches (swap semantics + 4 x 4 transposition between lane groups and dwords)\n", bad ? "FAIL" : "ok", bad);
    return bad ? 1 : 0;
}
