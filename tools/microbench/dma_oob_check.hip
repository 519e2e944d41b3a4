// Check: what does an LDS-DMA (buffer_load_dwordx4 ... offen lds) write into LDS for lanes whose buffer offset is out
// of range -- (a) per-lane voffset beyond num_records, (b) an SGPR soffset that pushes every lane out of range --
// and is the range check applied per lane (in-range lanes of the same instruction still land)?
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/dma_oob_check.hip -o tools/microbench/dma_oob_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k(const unsigned *src, unsigned *dst, unsigned nbytes, unsigned soff_all_oob)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned *l32 = reinterpret_cast<unsigned *>(lds);
    for (int i = threadIdx.x; i < 3 * 256; i += 64) l32[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // (a) odd lanes out of range through voffset
    unsigned voff = lane * 16u;
    if (lane & 1) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, 0);
    // (b) every lane out of range through soffset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + 1024), 16, lane * 16u,
                                             soff_all_oob, 0, 0);
    // (c) soffset in range: lanes read src + 1024 + lane * 16; the last 16 lanes fall past num_records
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + 2048), 16, lane * 16u,
                                             nbytes - 768u, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 256; i += 64) dst[i] = l32[i];
}

int main()
{
    const unsigned nbytes = 4096;
    std::vector<unsigned> h(nbytes / 4 + 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    unsigned *src, *dst;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dst, 3 * 1024);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64, 3 * 1024>>>(src, dst, nbytes, 0x7ffff000u);
    std::vector<unsigned> o(3 * 256);
    hipError_t e = hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost);
    long a_ok = 0, a_zero = 0, a_untouched = 0, a_other = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
            const unsigned v = o[lane * 4 + d];
            if (lane & 1) { if (v == 0) ++a_zero; else if (v == 0xdeadbeefu) ++a_untouched; else ++a_other; }
            else if (v == h[lane * 4 + d]) ++a_ok; else ++a_other;
        }
    long b_zero = 0, b_untouched = 0, b_other = 0;
    for (int i = 0; i < 256; ++i) { const unsigned v = o[256 + i]; if (v == 0) ++b_zero; else if (v == 0xdeadbeefu) ++b_untouched; else ++b_other; }
    long c_ok = 0, c_zero = 0, c_untouched = 0, c_other = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
            const unsigned v = o[512 + lane * 4 + d];
            const unsigned byte = nbytes - 768u + lane * 16u + d * 4u;
            if (byte < nbytes) { if (v == h[byte / 4]) ++c_ok; else ++c_other; }
            else if (v == 0) ++c_zero; else if (v == 0xdeadbeefu) ++c_untouched; else ++c_other;
        }
    printf("err=%d\n(a) voffset OOB on odd lanes : even lanes ok %ld/128, odd lanes zero %ld untouched %ld other %ld (of 128)\n", (int)e, a_ok,
           a_zero, a_untouched, a_other);
    printf("(b) soffset OOB for all lanes : zero %ld untouched %ld other %ld (of 256)\n", b_zero, b_untouched, b_other);
    printf("(c) soffset in range, tail OOB: in-range ok %ld/192, tail zero %ld untouched %ld other %ld (of 64)\n", c_ok, c_zero, c_untouched, c_other);
    return 0;
}
