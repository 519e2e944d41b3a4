// Microbenchmark: LDS gather rate in the access shape of an LDS-windowed MSDA kernel:
// 8 lanes read one 128-byte (fp32, ds_read_b128) or 64-byte (bf16, ds_read_b64) pixel row; the 8 lane
// groups of a wave read 8 rows that are adjacent (neighbouring queries) or pseudo-random.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_gather_bw.hip -o tools/microbench/lds_gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int LANE_BYTES, bool RANDOM, int THREADS>
__global__ __launch_bounds__(THREADS) void lds_gather(int iters, int rows, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < rows * LANE_BYTES * 8 / 4; i += THREADS) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    unsigned state = (blockIdx.x * THREADS + threadIdx.x / 64) * 2654435761u + 12345u + (RANDOM ? grp * 977u : 0u);
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            state = state * 1664525u + 1013904223u;
            unsigned row = RANDOM ? ((state >> 8) & 1023u) : (((state >> 8) & 1015u) + grp);   // adjacent: grp-th neighbour (rows == 1024)
            const unsigned off = row * (LANE_BYTES * 8) + sub * LANE_BYTES;
            if constexpr (LANE_BYTES == 16) acc += *reinterpret_cast<const f32x4 *>(lds + off);
            else { f32x2 v = *reinterpret_cast<const f32x2 *>(lds + off); acc.x += v.x; acc.y += v.y; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int LANE_BYTES, bool RANDOM, int THREADS>
static void run(float *sink, const char *label)
{
    const int rows = 1024, iters = 400, blocks = 256 * (THREADS == 1024 ? 1 : 2);
    const size_t lds_bytes = (size_t)rows * LANE_BYTES * 8;
    hipFuncSetAttribute((const void *)lds_gather<LANE_BYTES, RANDOM, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    lds_gather<LANE_BYTES, RANDOM, THREADS><<<blocks, THREADS, lds_bytes>>>(4, rows, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lds_gather<LANE_BYTES, RANDOM, THREADS><<<blocks, THREADS, lds_bytes>>>(iters, rows, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * THREADS * iters * 16 * LANE_BYTES;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-28s lane=%2dB threads/WG=%4d WG=%d : %7.1f TB/s  %6.1f B/clk/CU  (%.3f ms) err=%d\n", label, LANE_BYTES, THREADS,
           blocks, tbs, tbs * 1e12 / 256 / 2.4e9, ms, (int)hipGetLastError());
}

int main()
{
    float *sink; hipMalloc(&sink, 4);
    run<16, false, 512>(sink, "adjacent rows"); run<16, true, 512>(sink, "random rows");
    run<8, false, 512>(sink, "adjacent rows");  run<8, true, 512>(sink, "random rows");
    run<16, false, 1024>(sink, "adjacent rows"); run<16, true, 1024>(sink, "random rows");
    run<8, false, 1024>(sink, "adjacent rows");  run<8, true, 1024>(sink, "random rows");
    return 0;
}
