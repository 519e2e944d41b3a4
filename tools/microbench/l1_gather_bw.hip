// Microbenchmark: how fast can one CU pull 128-byte (fp32) / 64-byte (bf16) head rows out of L1 / L2
// with the access shape of the MSDA gather (8 row segments per wave instruction)?
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/l1_gather_bw.hip -o tools/microbench/l1_gather_bw
// Prints GB/s and bytes/clk/CU (at the nominal 2.4 GHz) for several footprints and load widths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Each lane group of 8 lanes reads one `SEG`-byte row segment; the 8 groups of a wave read 8 rows chosen
// pseudo-randomly inside a per-block window of `rows` rows (row pitch 1024 B, like the value tensor).
template <int LANE_BYTES>
__global__ __launch_bounds__(256) void gather_kernel(const char *base, unsigned rows, unsigned window_stride_rows,
                                                     int iters, unsigned total_bytes, float *sink)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, total_bytes, 0x00020000);
    const unsigned lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned win = (blockIdx.x * window_stride_rows) * 1024u;
    unsigned state = wave * 2654435761u + grp * 40503u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        unsigned offs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            state = state * 1664525u + 1013904223u;
            offs[j] = win + ((state >> 8) % rows) * 1024u + grp * (LANE_BYTES * 8) + sub * LANE_BYTES;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (LANE_BYTES == 16) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, offs[j], 0, 0);
                acc += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.w);
            } else {
                u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, offs[j], 0, 0);
                acc += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.y);
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int LANE_BYTES>
static void run(const char *d, unsigned total_bytes, unsigned rows, unsigned stride_rows, int blocks, float *sink, const char *label)
{
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather_kernel<LANE_BYTES><<<blocks, 256>>>(d, rows, stride_rows, 10, total_bytes, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather_kernel<LANE_BYTES><<<blocks, 256>>>(d, rows, stride_rows, iters, total_bytes, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 8 * LANE_BYTES;
    const double gbs = bytes / (ms * 1e-3) / 1e9;
    printf("%-44s lane=%2dB rows/window=%6u blocks=%5d : %8.1f GB/s  %6.1f B/clk/CU  (%.3f ms)\n", label, LANE_BYTES, rows,
           blocks, gbs, gbs * 1e9 / 256 / 2.4e9, ms);
}

int main()
{
    const unsigned total = 1u << 30;   // 1 GiB table
    char *d; float *sink;
    hipMalloc(&d, total); hipMalloc(&sink, 4);
    hipMemset(d, 1, total);
    const int blocks = 256 * 8;
    // window of 16 rows = 16 KB per block (L1-resident), windows of neighbouring blocks overlap (stride 4 rows)
    run<16>(d, total, 16, 4, blocks, sink, "L1-resident window (16 KB/block)");
    run<8>(d, total, 16, 4, blocks, sink, "L1-resident window (16 KB/block)");
    run<16>(d, total, 256, 64, blocks, sink, "256 KB window/block (L2)");
    run<8>(d, total, 256, 64, blocks, sink, "256 KB window/block (L2)");
    run<16>(d, total, 4096, 64, blocks, sink, "4 MB window/block (L2/MALL)");
    run<16>(d, total, 90000, 1, blocks, sink, "90 MB shared table (MALL)");
    run<8>(d, total, 90000, 1, blocks, sink, "90 MB shared table (MALL)");
    run<16>(d, total, 1000000, 1, blocks, sink, "1 GB table (HBM)");
    return 0;
}
