// Microbenchmark: VALU issue rate per SIMD for the instructions of the MSDA inner loop (v_fma_f32, v_pk_fma_f32,
// v_lshlrev_b32 / v_and_b32), at 1, 2 and 4 waves per SIMD.  Reports wave-instructions per clock per CU from in-kernel
// s_memtime stamps (shader clock) and from wall time.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(int iters, float seed, float *sink, unsigned long long *cyc)
{
    float a[16]; f32x2 p[8]; unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; u[i] = (unsigned)(threadIdx.x * 7 + i); }
    for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
    const f32x2 w2 = {seed * 0.5f, seed * 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
            } else {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                    asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i + 1]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i]; for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND> static void run(const char *name, int per_iter, int threads, float *sink, unsigned long long *cyc)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(100, 1.0001f, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, threads>>>(iters, 1.0001f, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = (double)iters * per_iter;
    const double waves_per_simd = threads / 64 / 4.0;
    printf("%-22s waves/SIMD=%.0f : %.2f cyc per wave-instr per wave (in-kernel) -> %.2f cyc per wave-instr per SIMD; clock %.2f GHz; %.2f wave-instr/clk/CU\n",
           name, waves_per_simd, (double)c / instr_per_wave, (double)c / instr_per_wave / waves_per_simd,
           (double)c / (ms * 1e6), instr_per_wave * (threads / 64) / (double)c);
}

int main()
{
    float *sink; unsigned long long *cyc; hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32", 64, threads, sink, cyc);
        run<1>("v_pk_fma_f32", 64, threads, sink, cyc);
        run<2>("v_lshlrev/v_and", 64, threads, sink, cyc);
    }
    return 0;
}
