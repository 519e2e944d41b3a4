// Microbenchmark: the SPEED OF LIGHT of an LDS-sourced MSDA gather -- the exact inner loop of
// csrc/msda_sweep.hip (per 8 samples x 32 channels: one ds_read_b128 of the A operand, four ds_read_b64_tr_b16 of gathered
// corner rows, two v_mfma_f32_16x16x32_bf16; per 16 samples one ds_read_b128 of staged row addresses) on windows that are
// RESIDENT in LDS, with the real corner pattern (top-left row, right neighbour one column stride further, bottom rows 64 B
// further on; the two K-groups of a 32-lane half on opposite channel halves): no fills, no set-up, no staging writes (mode 0),
// or with the per-step staging writes of the set-up role (mode 1).  One launch processes as many samples as one launch of the
// operator at BASELINE.json configs[1]: 4 images x 22,323 queries x 8 heads x 16 samples = 11.43 M samples = 2.93 GB of rows.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_gather_mfma_rate.hip -o tools/microbench/lds_gather_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kWinBytes = 127744;                 // the four rings of msda_sweep.hip
constexpr int kColStride = 22 * 64;               // bytes from a column to the next (== 128 mod 256)
constexpr int kWaveBytes = 1792;                  // W 1024 | O 512 | pad
constexpr int kZero = 1024;

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(const unsigned *seed, int iters, float *sink)
{
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned win0 = kZero + WAVES * kWaveBytes;
    // windows: finite bf16 values
    for (int i = tid; i < kWinBytes / 4; i += WAVES * 64) reinterpret_cast<unsigned *>(lds + win0)[i] = 0x3f803f80u + (seed[i & 1023] & 0x007f007fu);
    for (int i = tid; i < kZero / 4; i += WAVES * 64) reinterpret_cast<unsigned *>(lds)[i] = 0u;
    const int qi = lane >> 4, sl = (lane >> 2) & 3, pp = lane & 3;          // set-up role: query, level, point
    const int kg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;          // gather role
    const int am = lane & 15, ag = am >> 1, apart = am & 1;
    const unsigned wave_off = kZero + wave * kWaveBytes;
    unsigned char *wreg = lds + wave_off;
    const unsigned par32 = (unsigned)(qi & 1) * 32u;
    // this lane's sample: a random in-window position of its level's ring (rows 0..20, columns 0..16)
    const unsigned r = seed[(blockIdx.x * 1024 + tid) & 1023];
    const unsigned ring = win0 + (sl == 0 ? 0u : sl == 1 ? 45056u : sl == 2 ? 78848u : 107008u);
    const unsigned tl = lds0 + ring + (r % 16u) * kColStride + ((r >> 8) % 16u) * 64u + par32;
    unsigned char *st_o = wreg + 1024 + qi * 64 + sl * 16 + pp * 4;
    unsigned char *st_w = wreg + qi * 256 + sl * 32 + pp * 8;
    *reinterpret_cast<unsigned *>(st_o) = tl;
    *reinterpret_cast<unsigned *>(st_o + 256) = tl + kColStride;
    *reinterpret_cast<u32x2 *>(st_w) = u32x2{0x3e803e80u, 0x3e803e80u};
    *reinterpret_cast<u32x2 *>(st_w + 128) = u32x2{0x3a803a80u, 0x3a803a80u};
    __syncthreads();
    const unsigned cd = (unsigned)tp * 8u + (unsigned)(tq >> 1) * 64u;
    const unsigned o_rd = lds0 + wave_off + 1024u + (unsigned)(tq & 1) * 256u + (unsigned)kg * 64u;
    const unsigned w_rd = (am < 8 && ag == kg) ? lds0 + wave_off + (unsigned)(ag * 256 + apart * 128) : lds0 + 544u;
    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a));
    };
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    struct Operands { u32x4 af; u32x2 x0, x1, y0, y1; };
    auto fetch = [&](unsigned wa, unsigned oa, unsigned ob) {
        Operands o;
        o.af = lds_b128(wa);
        o.x0 = lds_tr(oa); o.x1 = lds_tr(ob); o.y0 = lds_tr(oa ^ 32u); o.y1 = lds_tr(ob ^ 32u);
        return o;
    };
    auto fma2 = [&](const Operands &o) {
        const u32x4 b0 = {o.x0.x, o.x0.y, o.x1.x, o.x1.y}, b1 = {o.y0.x, o.y0.y, o.y1.x, o.y1.y};
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, o.af), __builtin_bit_cast(bf16x8, b0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, o.af), __builtin_bit_cast(bf16x8, b1), acc1, 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {                           // the staging writes of a step (same values)
            *reinterpret_cast<unsigned *>(st_o) = tl;
            *reinterpret_cast<unsigned *>(st_o + 256) = tl + kColStride;
            *reinterpret_cast<u32x2 *>(st_w) = u32x2{0x3e803e80u, 0x3e803e80u};
            *reinterpret_cast<u32x2 *>(st_w + 128) = u32x2{0x3a803a80u, 0x3a803a80u};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        u32x4 so0 = lds_b128(o_rd), so1 = lds_b128(o_rd + 16u);
        Operands ra = fetch(w_rd, so0.x + cd, so0.y + cd);
        Operands rb = fetch(w_rd + 16, so0.z + cd, so0.w + cd);
        so0 = lds_b128(o_rd + 32u);
        fma2(ra);
        ra = fetch(w_rd + 32, so1.x + cd, so1.y + cd);
        fma2(rb);
        rb = fetch(w_rd + 48, so1.z + cd, so1.w + cd);
        so1 = lds_b128(o_rd + 48u);
        fma2(ra);
        ra = fetch(w_rd + 64, so0.x + cd, so0.y + cd);
        fma2(rb);
        rb = fetch(w_rd + 80, so0.z + cd, so0.w + cd);
        fma2(ra);
        ra = fetch(w_rd + 96, so1.x + cd, so1.y + cd);
        fma2(rb);
        rb = fetch(w_rd + 112, so1.z + cd, so1.w + cd);
        fma2(ra);
        fma2(rb);
        asm volatile("" ::: "memory");             // the next trip reads LDS again
    }
    const float s = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
    if (s == 123.456f) sink[0] = s;
}

template <int WAVES, int MODE> static void run(const unsigned *seed, float *sink, const char *label)
{
    const int blocks = 256;
    const double samples = 4.0 * 22323 * 8 * 16;
    const int iters = (int)(samples / ((double)blocks * WAVES * 64) + 0.5);
    const size_t lds = kZero + WAVES * kWaveBytes + kWinBytes;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<WAVES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) k<WAVES, MODE><<<blocks, WAVES * 64, lds>>>(seed, iters, sink);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) k<WAVES, MODE><<<blocks, WAVES * 64, lds>>>(seed, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, done = (double)blocks * WAVES * 64 * iters;
    printf("%-44s %2d waves/CU, %3d trips: %7.1f us per launch for %.2f M samples = %.2f GB of rows -> %5.1f TB/s of rows, %5.2f of the 8 TB/s roofline "
           "if nothing else cost time (228.6 MB algorithmic)\n", label, WAVES, iters, us, done / 1e6, done * 256 / 1e9, done * 256 / us / 1e6,
           228.58752 / us / 8.0);
}

int main()
{
    unsigned *seed; float *sink;
    hipMalloc(&seed, 4096); hipMalloc(&sink, 4);
    unsigned h[1024];
    srand(7);
    for (int i = 0; i < 1024; ++i) h[i] = (unsigned)rand() * 2654435761u + (unsigned)rand();
    hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    run<16, 0>(seed, sink, "gather loop only");
    run<16, 1>(seed, sink, "gather loop + staging writes per step");
    run<8, 0>(seed, sink, "gather loop only");
    run<8, 1>(seed, sink, "gather loop + staging writes per step");
    run<4, 0>(seed, sink, "gather loop only");
    return 0;
}
