// Microbenchmark: rate at which a workgroup can copy a 2-D window ("rect") of one (image, head) plane of the
// MSDA value tensor from L2 into LDS -- the fill step of an LDS-tiled gather kernel.
//   layout 0: pixel-major  [S][8 heads][32 ch] bf16 -> one head row = a 64-byte piece every 512 B (the op's layout)
//   layout 1: pixel-major, TWO heads per workgroup -> 128-byte pieces every 512 B
//   layout 2: head-major   [8][S][32 ch]           -> rect rows are contiguous runs of 64-byte pixels
//   path   r: buffer_load_dwordx4 -> VGPR -> ds_write_b128      d: global_load_lds_dwordx4 (LDS-DMA)
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/rect_fill_bw.hip -o tools/microbench/rect_fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int W0 = 168, H0 = 100, S = 22323, RW = 36, RH = 36;

template <int LAYOUT, bool DMA, int THREADS>
__global__ __launch_bounds__(THREADS) void fill_kernel(const unsigned char *__restrict__ value, int iters, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int PXB = LAYOUT == 1 ? 128 : 64;            // bytes per pixel in LDS
    constexpr int LPP = PXB / 16;                          // lanes per pixel
    const int bh = blockIdx.x % 16, b = bh >> 3, h = bh & 7;   // 2 images x 8 heads: one head per XCD, L2-resident
    const unsigned char *plane = value + (size_t)b * S * 512;
    unsigned state = blockIdx.x * 2654435761u + 99u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const int rx = (state >> 8) % (W0 - RW), ry = (state >> 20) % (H0 - RH);
        for (int i = threadIdx.x; i < RW * RH * LPP; i += THREADS) {
            const int px = i / LPP, c = i % LPP;
            const int y = ry + px / RW, x = rx + px % RW;
            size_t goff;
            if (LAYOUT == 0) goff = (size_t)(y * W0 + x) * 512 + h * 64 + c * 16;
            else if (LAYOUT == 1) goff = (size_t)(y * W0 + x) * 512 + (h & 6) * 64 + c * 16;
            else goff = ((size_t)h * S + (y * W0 + x)) * 64 + c * 16;
            if constexpr (DMA) {
                // LDS destination = wave-uniform base (M0) + lane * 16: consecutive i of one wave are consecutive 16-byte slots
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane + goff),
                                                 (__attribute__((address_space(3))) void *)(lds + (size_t)(i & ~63) * 16), 16, 0, 0);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(plane + goff);
                *reinterpret_cast<u32x4 *>(lds + (size_t)i * 16) = v;
            }
        }
        __syncthreads();
        acc += reinterpret_cast<unsigned *>(lds)[(threadIdx.x * 37 + it) % (RW * RH * PXB / 4)];
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int LAYOUT, bool DMA, int THREADS>
static void run(const unsigned char *value, unsigned *sink, const char *label)
{
    const int iters = 200, blocks = 256 * (THREADS == 1024 ? 1 : 2) * 4;
    constexpr int PXB = LAYOUT == 1 ? 128 : 64;
    const size_t lds_bytes = (size_t)RW * RH * PXB;
    if (lds_bytes * (THREADS == 1024 ? 1 : 2) > 160 * 1024) { printf("%-40s skipped (LDS)\n", label); return; }
    hipFuncSetAttribute((const void *)fill_kernel<LAYOUT, DMA, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fill_kernel<LAYOUT, DMA, THREADS><<<blocks, THREADS, lds_bytes>>>(value, 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fill_kernel<LAYOUT, DMA, THREADS><<<blocks, THREADS, lds_bytes>>>(value, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * lds_bytes;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-40s %s threads/WG=%4d WG=%d : %6.2f TB/s  %5.1f B/clk/CU @2.1GHz  (%.3f ms) err=%d\n", label, DMA ? "dma" : "reg", THREADS,
           blocks, tbs, tbs * 1e12 / 256 / 2.1e9, ms, (int)hipGetLastError());
}

int main()
{
    unsigned char *value; unsigned *sink;
    hipMalloc(&value, (size_t)4 * S * 512); hipMalloc(&sink, 4);
    hipMemset(value, 1, (size_t)4 * S * 512);
    run<0, false, 1024>(value, sink, "pixel-major 64B pieces");
    run<0, false, 512>(value, sink, "pixel-major 64B pieces");
    run<0, true, 1024>(value, sink, "pixel-major 64B pieces");
    run<0, true, 512>(value, sink, "pixel-major 64B pieces");
    run<1, false, 1024>(value, sink, "pixel-major 128B pieces (2 heads)");
    run<1, true, 1024>(value, sink, "pixel-major 128B pieces (2 heads)");
    run<2, false, 1024>(value, sink, "head-major contiguous rows");
    run<2, false, 512>(value, sink, "head-major contiguous rows");
    run<2, true, 1024>(value, sink, "head-major contiguous rows");
    run<2, true, 512>(value, sink, "head-major contiguous rows");
    return 0;
}
