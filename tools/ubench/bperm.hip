// Micro-benchmark: throughput of ds_bpermute_b32 vs ds_read_b64 (bank-private table) per CU, 16 or 4 wavefronts per CU.
// hipcc --offload-arch=gfx950 -O3 -o bperm bperm.hip && ./bperm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int MODE> __global__ __launch_bounds__(1024) void k(const uint32_t* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char tab[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x)
        reinterpret_cast<float*>(tab)[i] = (float)(i & 255);
    __syncthreads();
    uint32_t w = in[tid + blockIdx.x * blockDim.x];
    float cv = (float)(lane & 15);
    float acc = 0.f;
    const uint32_t lane_off = (lane & 31) * 8;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) { // 16 x (v_perm + ds_read_b64 + 2 fma) per 16 bytes ... here 8 per iteration
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t addr = __builtin_amdgcn_perm(w, lane_off, 0x0C0C0400u + ((j & 3) << 8));
                const f32x2 pr = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(addr);
                acc = __builtin_fmaf(pr[0], cv, acc);
                acc = __builtin_fmaf(pr[1], cv, acc);
            }
        } else if constexpr (MODE == 1) { // 16 x (shift + bpermute + fma), lanes 0..15 only as sources
            const uint32_t we = w & 0x0F0F0F0Fu, wo = (w >> 4) & 0x0F0F0F0Fu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a0 = (j == 0) ? (we << 2) : (we >> (8 * j - 2));
                const int a1 = (j == 0) ? (wo << 2) : (wo >> (8 * j - 2));
                acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a0, __builtin_bit_cast(int, cv))), cv, acc);
                acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a1, __builtin_bit_cast(int, cv))), cv, acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a0 = (j == 0) ? (wo << 2) : (wo >> (8 * j - 2));
                const int a1 = (j == 0) ? (we << 2) : (we >> (8 * j - 2));
                acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a0, __builtin_bit_cast(int, cv))), cv, acc);
                acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a1, __builtin_bit_cast(int, cv))), cv, acc);
            }
        } else { // unmasked 6-bit addresses: any of the 64 lanes as source
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int a0 = (j == 0) ? (w << 2) : (w >> ((2 * j) % 30));
                acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a0, __builtin_bit_cast(int, cv))), cv, acc);
            }
        }
        w = w * 1664525u + 1013904223u;
    }
    out[tid + blockIdx.x * blockDim.x] = acc;
}

template <int MODE> void run(const char* name, int threads, int lookups_per_iter) {
    const int blocks = 256, iters = 2000;
    uint32_t* in;
    float* out;
    hipMalloc(&in, blocks * 1024 * 4);
    hipMalloc(&out, blocks * 1024 * 4);
    hipMemset(in, 0x5a, blocks * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(in, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_cu = (double)(threads / 64) * iters * lookups_per_iter;
    printf("%-34s %4d thr/WG: %8.3f ms  -> %6.2f ns per look-up wave-instruction per CU (%.2f cycles @2.4 GHz)\n", name, threads, ms,
           ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
    hipFree(in);
    hipFree(out);
}

int main() {
    for (int threads : {1024, 512, 256}) {
        run<0>("v_perm + ds_read_b64 + 2 fma", threads, 8);
        run<1>("shift + ds_bpermute(16 src) + fma", threads, 16);
        run<2>("shift + ds_bpermute(64 src) + fma", threads, 16);
    }
    return 0;
}
