// Microbenchmark: do f32 MFMA (v_mfma_f32_16x16x4_f32) and f32 VALU work from two waves on the
// SAME SIMD overlap on gfx950?  Each workgroup has 8 waves (2 per SIMD); waves 0-3 run role A,
// waves 4-7 role B.  Roles: 0 = idle, 1 = MFMA loop, 2 = VALU fma loop, 3 = transcendental loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(float* out, int roleA, int roleB, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? roleA : roleB;
    float r = threadIdx.x * 1e-3f;
    if (role == 1) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){r, r, r, r};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(r, 1.0001f, acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
    } else if (role == 2) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            }
        }
        for (int i = 0; i < 16; ++i) r += a[i];
    } else if (role == 3) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = r + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a[i]));
            }
        }
        for (int i = 0; i < 16; ++i) r += a[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"idle", "mfma", "valu", "trans"};
    int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (auto& c : combos) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, c[0], c[1], iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, c[0], c[1], iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per-wave work: mfma: iters*32 MFMAs (32 cyc each); valu: iters*256 fma (4 cyc issue); trans: iters*128*2 ops (8 cyc)
        printf("A=%-5s B=%-5s  %.3f ms\n", names[c[0]], names[c[1]], ms);
    }
    return 0;
}
