// Microbenchmark: what does one vector-memory wave instruction cost in the texture addresser as a function of how many of its
// lanes do anything?  The MSDA gather issues 64 buffer_load_dwordx4 per wave (16 queries x 4 lanes x 16 B each); candidates for
// "fewer lane addresses" are (a) lanes switched off by EXEC, (b) lanes whose offset fails the buffer range check (no request).
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/ta_mask_rate.hip -o tools/microbench/ta_mask_rate
// Every wave issues `iters` x 8 loads of 16 B per lane from a 4-KiB window (L1 resident); 20 waves per CU, all 256 CUs.
// Prints the cycles per wave instruction and CU (nominal 2.4 GHz) for: all 64 lanes, 32 / 16 / 4 lanes by EXEC, and the same
// counts live with the rest out of range.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: EXEC mask, 1: out-of-range offsets
__global__ __launch_bounds__(256) void k(const char *base, int live, int iters, float *sink)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 1u << 20, 0x00020000);
    const unsigned lane = threadIdx.x & 63;
    const bool on = (int)lane < live;
    const unsigned off0 = (blockIdx.x & 63) * 4096u + lane * 16u;
    float acc = 0.f;
    if (MODE == 1 || on) {
        for (int i = 0; i < iters; ++i) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned o = (MODE == 1 && !on) ? 0x80000000u : off0 + (unsigned)j * 1024u % 4096u;
                // asm volatile: the compiler must not merge the loads of the same address across iterations
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[j]) : "v"(o), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __builtin_bit_cast(float, v[j].x) + __builtin_bit_cast(float, v[j].w);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE> static void run(const char *d, int live, float *sink, const char *label)
{
    const int iters = 400, blocks = 256 * 5;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, live, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, live, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256;
    printf("%-28s live lanes %2d : %7.3f ms  %6.1f clk per wave instruction and CU (at 2.4 GHz; %5.1f at 2.1 GHz)\n", label, live, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu, ms * 1e-3 * 2.1e9 / instr_per_cu);
}

int main()
{
    char *d; float *sink;
    hipMalloc(&d, 1u << 20); hipMalloc(&sink, 4);
    hipMemset(d, 1, 1u << 20);
    for (int live : {64, 32, 16, 4}) run<0>(d, live, sink, "EXEC-masked");
    for (int live : {64, 32, 16, 4, 0}) run<1>(d, live, sink, "rest out of range");
    return 0;
}
