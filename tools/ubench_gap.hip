// Microbenchmark: how much vector work of the SAME wave hides in the gap behind a v_mfma_f32_32x32x16_bf16 on gfx950?
// One MFMA followed (program order pinned by sched_barrier) by NF fillers of kind KIND:
//   0 none | 1 v_fma_f32 | 2 v_exp_f32 | 3 exp+add+rcp chain on independent values (the sigmoid of the biGRU step)
//   4 v_pk_fma_f32 | 5 ds_read_b128 (LDS, conflict-free) | 6 v_cvt_pk_bf16_f32
// WAVES = waves per workgroup (4: one per SIMD, 8: two per SIMD); 100 KiB of LDS forces one workgroup per CU.
// Prints shader cycles (s_memtime) per MFMA slot, median over workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NF>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x16 acc[4];
    bf16x8 a, b;
    float v[16];
    f32x2 p[8];
    float r = threadIdx.x * 1e-3f;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)r; b[j] = (__bf16)1.0f; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = r;
    for (int i = 0; i < 16; ++i) v[i] = r + i;
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){r + i, r - i};
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    f32x4 ld[4] = {lp[0], lp[64], lp[128], lp[192]};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int q = (g * NF + f) & 15;
                if constexpr (KIND == 1) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
                if constexpr (KIND == 2) v[q] = __builtin_amdgcn_exp2f(v[q]);
                if constexpr (KIND == 3) v[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[q]));
                if constexpr (KIND == 4) p[q & 7] = __builtin_elementwise_fma(p[q & 7], (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f});
                if constexpr (KIND == 5) ld[q & 3] = lp[(q & 7) * 64];
                if constexpr (KIND == 6) { a[2 * (q & 3)] = (__bf16)v[q]; a[2 * (q & 3) + 1] = (__bf16)v[q ^ 1]; }
            }
            if constexpr (KIND == 5) asm volatile("" ::"v"(ld[0]), "v"(ld[1]), "v"(ld[2]), "v"(ld[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][5];
    for (int i = 0; i < 16; ++i) r += v[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) r += ld[i].x;
    r += (float)a[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* kinds[] = {"none", "v_fma", "v_exp", "exp+add+rcp", "v_pk_fma", "ds_read_b128", "cvt_pk_bf16"};

template <int KIND, int NF>
void run(float* d, long long* c, int waves) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<KIND, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(waves * 64), 100 * 1024, 0, d, c, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * waves);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[h.size() / 2] / (iters * 8.0);
    printf("%-14s x%-2d  waves/SIMD %d : %6.1f cycles per MFMA slot (per wave)  -> %6.1f per SIMD-MFMA\n", kinds[KIND], NF, waves / 4, per,
           per / (waves / 4));
}

template <int KIND, int NF>
void both(float* d, long long* c) { run<KIND, NF>(d, c, 4); run<KIND, NF>(d, c, 8); }

int main() {
    float* d; long long* c;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&c, 256 * 8 * 8);
    both<0, 0>(d, c);
    both<1, 2>(d, c); both<1, 4>(d, c); both<1, 6>(d, c); both<1, 10>(d, c);
    both<2, 1>(d, c); both<2, 2>(d, c); both<2, 3>(d, c); both<2, 4>(d, c);
    both<3, 1>(d, c); both<3, 2>(d, c);
    both<4, 2>(d, c); both<4, 4>(d, c);
    both<5, 1>(d, c); both<5, 2>(d, c);
    both<6, 2>(d, c);
    return 0;
}
