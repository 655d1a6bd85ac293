// Micro-benchmark: what the LDS-DMA path (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) of one CU delivers, and what an
// instruction costs the wavefront that issues it - the two numbers the batched 4-bit GEMM kernels (gemm4_mfma_kq.hip) are built
// around. One 512-thread workgroup per CU; per iteration ISSUERS wavefronts request PER instructions each, then wait for their
// own requests (vmcnt(0)) and meet at a barrier (the phase structure of the kernel). Sources:
//   A: the same 256 KiB for every workgroup (L2 hits after first touch: the activation operand),
//   W: a private HBM-resident stream per workgroup (the weights).
// Prints cycles per iteration, bytes / cycle / CU and the s_memtime cycles the issuing wavefront spends in the issue sequence.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ i32x4 rsrc(const void* base) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{static_cast<int>(a), static_cast<int>((a >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
}
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds, unsigned voff, unsigned soff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// MODE 0: A only, 1: W only, 2: mixed (PER = PA + PW, the first PA of every wavefront from A)
template <int ISSUERS, int PA, int PW, int READERS>
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char* a, const unsigned char* w, unsigned long long* stamps, float* sink,
                                                  int iters, long w_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4 rs_a = rsrc(a), rs_w = rsrc(w + blockIdx.x * w_per_wg);
    constexpr int PER = PA + PW;
    unsigned long long issue_cycles = 0;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (wave < ISSUERS) {
            const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const unsigned slot = (wave * PER + i) * 1024u;                  // ISSUERS * PER KiB of LDS per iteration, two halves alternate
                const unsigned lds = (it & 1) * (ISSUERS * PER * 1024u) + slot;
                if (i < PA)
                    dma16(rs_a, lds, lane * 16u, ((it * ISSUERS * PA + wave * PA + i) * 1024u) & 0x3FFFFu);
                else
                    dma16(rs_w, lds, lane * 16u, (it * ISSUERS * PW + wave * PW + (i - PA)) * 1024u);
            }
            const unsigned long long s1 = __builtin_amdgcn_s_memtime();
            issue_cycles += s1 - s0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (wave >= 8 - READERS) {
            // readers: the LDS read traffic of the other group's compute phase (32 x ds_read_b32 + 10 x ds_read_b128 per iteration)
#pragma unroll
            for (int i = 0; i < 32; ++i)
                acc += *reinterpret_cast<volatile float*>(smem + 65536 + ((lane * 4 + i * 256) & 0xFFFF));
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const volatile float4* p4 = reinterpret_cast<volatile float4*>(smem + ((lane * 16 + i * 1024) & 0xFFFF));
                acc += p4->x;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = issue_cycles;
    }
    if (acc == 123.456f)
        sink[0] = acc;
}

template <int ISSUERS, int PA, int PW, int READERS> void run(const char* name, const unsigned char* a, const unsigned char* w, long w_per_wg) {
    const int iters = 64, wgs = 256;
    unsigned long long* st;
    float* sink;
    hipMalloc(&st, wgs * 8 * 2 * sizeof(unsigned long long));
    hipMalloc(&sink, 4);
    auto kern = dma_kernel<ISSUERS, PA, PW, READERS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int r = 0; r < 3; ++r)
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 150 * 1024, 0, a, w, st, sink, iters, w_per_wg);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 8 * 2);
    hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double tot = 0, iss = 0;
    for (int b = 0; b < wgs; ++b) {
        tot += h[(b * 8) * 2];
        for (int wv = 0; wv < ISSUERS; ++wv)
            iss += h[(b * 8 + wv) * 2 + 1];
    }
    const double cyc = tot / wgs / iters, per_issue = iss / wgs / ISSUERS / iters / (PA + PW);
    const double bytes = ISSUERS * (PA + PW) * 1024.0;
    printf("%-58s %8.0f cycles / iteration  %6.1f B / cycle / CU   %6.0f cycles per request in the issue sequence\n", name, cyc, bytes / cyc, per_issue);
    hipFree(st);
    hipFree(sink);
}

int main() {
    unsigned char *a, *w;
    const long w_per_wg = 64L * 4 * 8 * 1024 + 4096;
    hipMalloc(&a, 1 << 20);
    hipMalloc(&w, 256 * w_per_wg + (1 << 20));
    hipMemset(a, 1, 1 << 20);
    hipMemset(w, 2, 256 * w_per_wg + (1 << 20));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs; one 512-thread workgroup per CU, 64 iterations of [requests, vmcnt(0), barrier]\n", p.name, p.multiProcessorCount);
    run<4, 4, 0, 0>("A (L2-resident): 4 wavefronts x 4 requests (16 KiB)", a, w, w_per_wg);
    run<8, 2, 0, 0>("A (L2-resident): 8 wavefronts x 2 requests (16 KiB)", a, w, w_per_wg);
    run<4, 7, 0, 0>("A (L2-resident): 4 wavefronts x 7 requests (28 KiB)", a, w, w_per_wg);
    run<8, 4, 0, 0>("A (L2-resident): 8 wavefronts x 4 requests (32 KiB)", a, w, w_per_wg);
    run<1, 8, 0, 0>("A (L2-resident): 1 wavefront x 8 requests (8 KiB)", a, w, w_per_wg);
    run<4, 0, 2, 0>("W (HBM stream): 4 wavefronts x 2 requests (8 KiB)", a, w, w_per_wg);
    run<4, 0, 4, 0>("W (HBM stream): 4 wavefronts x 4 requests (16 KiB)", a, w, w_per_wg);
    run<4, 4, 2, 0>("kernel mix: 4 wavefronts x (4 A + 2 W) (24 KiB)", a, w, w_per_wg);
    run<4, 4, 2, 4>("kernel mix + 4 wavefronts reading the LDS", a, w, w_per_wg);
    run<4, 4, 0, 4>("A: 4 wavefronts x 4 requests + 4 wavefronts reading the LDS", a, w, w_per_wg);
    return 0;
}
