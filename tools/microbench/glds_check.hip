// Check: global_load_lds_dwordx4 into LDS offsets above 64 KiB, exec-masked tails, and data placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(const unsigned *src, unsigned *dst, int n16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) reinterpret_cast<unsigned *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    for (int i = threadIdx.x; i < n16; i += 1024)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)i * 4),
                                         (__attribute__((address_space(3))) void *)(lds + (size_t)(i & ~63) * 16), 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) dst[i] = reinterpret_cast<unsigned *>(lds)[i];
}
int main()
{
    const int n16 = 10000;       // 160000 B: beyond 64 KiB, tail wave partially masked (10000 = 156*64 + 16)
    std::vector<unsigned> h(160 * 1024 / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 2654435761u;
    unsigned *src, *dst;
    hipMalloc(&src, h.size() * 4); hipMalloc(&dst, h.size() * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<<<1, 1024, 160 * 1024>>>(src, dst, n16);
    std::vector<unsigned> o(h.size());
    hipError_t e = hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, first = -1, untouched_bad = 0;
    for (size_t i = 0; i < o.size(); ++i) {
        const bool in = i < (size_t)n16 * 4;
        if (in && o[i] != h[i]) { if (first < 0) first = (long)i; ++bad; }
        if (!in && o[i] != 0xdeadbeefu) ++untouched_bad;
    }
    printf("err=%d copied dwords wrong: %ld (first at dword %ld = byte %ld), bytes past the tail clobbered: %ld\n", (int)e, bad, first, first * 4, untouched_bad);
    return 0;
}
