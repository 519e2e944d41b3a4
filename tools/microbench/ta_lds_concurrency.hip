// Microbenchmark: do texture-path loads (buffer_load_dwordx4) and LDS reads (ds_read_b128) of DIFFERENT waves of a CU proceed side
// by side, or do they take turns on the way into the vector registers?  The resident-levels MSDA kernel (csrc/msda_res.hip) moved
// half of the gather's rows from the texture path (~17 clk per 1-KiB wave instruction) to LDS (4-9 clk) and became 14 % faster, not
// 40 %: this measures what the hardware allows.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/ta_lds_concurrency.hip -o tools/microbench/ta_lds_concurrency
// 16 waves per CU (4 workgroups of 256 threads) on all 256 CUs.  A "texture" wave issues iters x 8 buffer_load_dwordx4 (16 B per lane
// from an L1-resident 4-KiB window); an "LDS" wave iters x 8 x R ds_read_b128 (conflict-free or random 64-byte rows).  Modes: all
// waves texture | all waves LDS | waves alternate (half texture, half LDS).  If the two paths are independent the mixed run takes
// max(texture half, LDS half); if they take turns, the sum.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int R, bool RANDOM>
__global__ __launch_bounds__(256) void k(const char *base, int mode, int iters, float *sink)
{
    __shared__ __attribute__((aligned(128))) unsigned char lds[32768];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 1u << 20, 0x00020000);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 16; i += 256) reinterpret_cast<u32x4 *>(lds)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const bool tex = mode == 0 || (mode == 2 && ((wave + blockIdx.x) & 1) == 0);
    float acc = 0.f;
    if (tex) {
        const unsigned off0 = (blockIdx.x & 63) * 4096u + lane * 16u;
        for (int i = 0; i < iters; ++i) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned o = off0 + (unsigned)j * 1024u % 4096u;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[j]) : "v"(o), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __builtin_bit_cast(float, v[j].x) + __builtin_bit_cast(float, v[j].w);
        }
    } else {
        // a query's 4 lanes read one 64-byte row: consecutive rows (conflict-free) or rows scattered by a per-query hash
        unsigned row = lane >> 2;
        for (int i = 0; i < iters * R; ++i) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned rr = RANDOM ? ((row * 2654435761u + (unsigned)(i * 8 + j) * 40503u) >> 7) & 511u : (row + (unsigned)j * 16u) & 511u;
                const unsigned a = rr * 64u + (lane & 3u) * 16u;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __builtin_bit_cast(float, v[j].x) + __builtin_bit_cast(float, v[j].w);
        }
    }
    if (acc == 123.456f) sink[0] = acc + lds[0];
}

template <int R, bool RANDOM> static double run(const char *d, int mode, float *sink)
{
    const int iters = 200, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<R, RANDOM><<<blocks, 256>>>(d, mode, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<R, RANDOM><<<blocks, 256>>>(d, mode, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

template <int R, bool RANDOM> static void table(const char *d, float *sink)
{
    const double t0 = run<R, RANDOM>(d, 0, sink), t1 = run<R, RANDOM>(d, 1, sink), t2 = run<R, RANDOM>(d, 2, sink);
    // per CU: mode 0: 16 waves x 1600 loads; mode 1: 16 x 1600 R reads; mode 2: 8 x 1600 loads + 8 x 1600 R reads
    printf("LDS reads per texture load R = %d, %s rows:\n", R, RANDOM ? "scattered" : "conflict-free");
    printf("   all 16 waves texture        %8.1f us  (%5.1f clk per load and CU at 2.4 GHz)\n", t0, t0 * 1e-6 * 2.4e9 / (16.0 * 1600));
    printf("   all 16 waves LDS            %8.1f us  (%5.1f clk per read and CU)\n", t1, t1 * 1e-6 * 2.4e9 / (16.0 * 1600 * R));
    printf("   8 waves texture + 8 LDS     %8.1f us   independent paths: max(%.1f, %.1f) = %.1f; taking turns: %.1f\n", t2, t0 / 2, t1 / 2,
           t0 / 2 > t1 / 2 ? t0 / 2 : t1 / 2, t0 / 2 + t1 / 2);
}

int main()
{
    char *d; float *sink;
    hipMalloc(&d, 1u << 20); hipMalloc(&sink, 4);
    hipMemset(d, 1, 1u << 20);
    table<1, false>(d, sink);
    table<4, false>(d, sink);
    table<2, true>(d, sink);
    table<4, true>(d, sink);
    return 0;
}
