// Microbenchmark: how do MFMA and VALU/transcendental work share one SIMD on gfx950?
// Each workgroup has 8 waves (2 per SIMD; waves w and w+4 share a SIMD); waves 0-3 run role A,
// waves 4-7 role B.  100 KiB of dynamic LDS forces ONE workgroup per CU.
// Roles: 0 idle | 1 f32 MFMA 16x16x4 | 2 VALU fma | 3 transcendental (exp2+rcp) | 4 bf16 MFMA 32x32x16
//        5 one wave interleaving bf16 MFMA with 6 VALU fma each | 6 same with f32 MFMA
//        7 one wave interleaving bf16 MFMA with 2 transcendentals + 2 fma each | 8 same with f32 MFMA
//        9 packed-fp32 fma (v_pk_fma_f32) only | 10 bf16 MFMA with 2 transcendentals + 1 v_pk_fma_f32 each
//        11 bf16 MFMA with 4 transcendentals + 2 fma each (the VALU-heavy mix of the bf16 biGRU step)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ROLE>
__device__ __forceinline__ float work(float r, int iters) {
    if constexpr (ROLE == 1) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){r, r, r, r};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(r, 1.0001f, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
    } else if constexpr (ROLE == 2) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        }
        for (int i = 0; i < 16; ++i) r += a[i];
    } else if constexpr (ROLE == 3) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a[i]));
        }
        for (int i = 0; i < 16; ++i) r += a[i];
    } else if constexpr (ROLE == 9) {
        f32x2 a[8];
        for (int i = 0; i < 8; ++i) a[i] = (f32x2){r + i, r - i};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f});
        }
        for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    } else if constexpr (ROLE == 4 || ROLE == 5 || ROLE == 7 || ROLE == 10 || ROLE == 11) {
        f32x16 acc[4];
        bf16x8 a, b;
        float v[12];
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)r; b[j] = (__bf16)1.0f; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = r;
        for (int i = 0; i < 12; ++i) v[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                    if constexpr (ROLE == 5) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) v[(i * 6 + q) % 12] = __builtin_fmaf(v[(i * 6 + q) % 12], 1.0001f, 0.5f);
                    }
                    if constexpr (ROLE == 7) {
                        v[2 * i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[2 * i]));
                        v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], 1.0001f, 0.5f);
                        v[8 + i] = __builtin_fmaf(v[8 + i], 1.0001f, 0.5f);
                    }
                    if constexpr (ROLE == 10) {
                        v[2 * i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[2 * i]));
                        f32x2 pk = {v[2 * i + 1], v[8 + i]};
                        pk = __builtin_elementwise_fma(pk, (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f});
                        v[2 * i + 1] = pk.x; v[8 + i] = pk.y;
                    }
                    if constexpr (ROLE == 11) {
                        v[2 * i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[2 * i]));
                        v[2 * i + 1] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[2 * i + 1]));
                        v[8 + i] = __builtin_fmaf(v[8 + i], 1.0001f, 0.5f);
                        v[8 + (i ^ 1)] = __builtin_fmaf(v[8 + (i ^ 1)], 1.0001f, 0.5f);
                    }
                }
        }
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][5];
        for (int i = 0; i < 12; ++i) r += v[i];
    } else if constexpr (ROLE == 6 || ROLE == 8) {
        f32x4 acc[8];
        float v[12];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){r, r, r, r};
        for (int i = 0; i < 12; ++i) v[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(r, 1.0001f, acc[i], 0, 0, 0);
                    if constexpr (ROLE == 6) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) v[(i * 6 + q) % 12] = __builtin_fmaf(v[(i * 6 + q) % 12], 1.0001f, 0.5f);
                    }
                    if constexpr (ROLE == 8) {
                        v[i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[i]));
                        v[8 + (i & 3)] = __builtin_fmaf(v[8 + (i & 3)], 1.0001f, 0.5f);
                    }
                }
        }
        for (int i = 0; i < 8; ++i) r += acc[i].x;
        for (int i = 0; i < 12; ++i) r += v[i];
    }
    return r;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ float pad[];
    if (iters < 0) pad[threadIdx.x] = 1.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r = threadIdx.x * 1e-3f;
    if (wave < 4) r = work<RA>(r, iters);
    else r = work<RB>(r, iters);
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static const char* names[] = {"idle", "f32mfma", "valu", "trans", "bf16mfma", "bf16mfma+6fma", "f32mfma+6fma", "bf16mfma+trans", "f32mfma+trans", "pkfma", "bf16mfma+tr+pk", "bf16mfma+4tr"};

template <int RA, int RB>
void run(float* d) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 100 * 1024, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 100 * 1024, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("A=%-15s B=%-15s %.3f ms\n", names[RA], names[RB], ms);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d);
    run<1, 1>(d); run<2, 2>(d); run<3, 3>(d); run<4, 4>(d);
    run<1, 2>(d); run<2, 1>(d); run<1, 3>(d); run<4, 2>(d); run<2, 4>(d); run<4, 3>(d); run<3, 4>(d);
    run<5, 0>(d); run<6, 0>(d); run<7, 0>(d); run<8, 0>(d); run<5, 5>(d); run<7, 7>(d); run<8, 8>(d);
    run<9, 0>(d); run<9, 9>(d); run<4, 9>(d); run<10, 0>(d); run<10, 10>(d); run<11, 0>(d); run<11, 11>(d);
    return 0;
}
