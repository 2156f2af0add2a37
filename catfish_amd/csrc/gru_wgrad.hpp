// Weight gradients of one bidirectional GRU layer straight from the fragment-layout buffers the training
// kernels already hold (rnn_class.py:62-71: the dW part of optimizer.minimize(loss)):
//
//   dWg[k][j] = sum_{w,t} [x_t ; h_prev][k] * da_g[j]      (gates kernel,     [Cin+64, 128], columns r | u)
//   dWc[k][j] = sum_{w,t} [x_t ; r*h_prev][k] * da_c[j]    (candidate kernel, [Cin+64,  64])
//   dbg[j] = sum da_g[j],  dbc[j] = sum da_c[j]
//
// The contraction runs over (window, step) pairs, i.e. over what the fragment layout spreads across LANES, so
// each (tile, t) pair is staged through LDS as [16 windows][features] and read back in MFMA operand order
// (A: lane -> (feature, window), B: lane -> (window, column)).  A workgroup of 12 waves owns a chunk of pairs;
// wave j accumulates the 16 output columns 16j..16j+15 for all Cin+64 rows in registers.  Partial sums go to a
// workspace and a second kernel adds them in a fixed order (bit-reproducible, no atomics) while scattering into
// the TensorFlow variable layout.
#pragma once

#define CF_WGRAD_WAVES 12

constexpr __host__ __device__ int gwg_rows(int cin) { return cin + CF_H; }                       // rows of both kernels
constexpr __host__ __device__ int gwg_partial_floats(int cin) { return (gwg_rows(cin) + 1) * 192; }  // + the bias row
constexpr __host__ __device__ int gwg_out_floats(int cin) { return gwg_rows(cin) * 192 + 192; }  // per direction: wg | bg | wc | bc

template <int CIN>
__global__ __launch_bounds__(CF_WGRAD_WAVES * 64) void gru_wgrad_kernel(const f32x4* __restrict__ X,    // layer input   [tile][t][CIN/16][lane]
                                                                        const f32x4* __restrict__ Y,    // layer output  [tile][t][8][lane]
                                                                        const f32x4* __restrict__ S,    // stash         [tile][t][2][12][lane]
                                                                        const f32x4* __restrict__ DA,   // [tile][t][2][12][lane]
                                                                        float* __restrict__ P,          // [chunk][2][rows + 1][192]
                                                                        int n_tiles, int pairs_per_wg) {
    constexpr int MX = CIN / 16;
    constexpr int SA = CIN + 128 + 4;         // x | h_prev | r*h_prev, padded row stride (floats)
    constexpr int SB = 192 + 4;
    constexpr int NA = MX + 8;                // staging items of the A side: MX x tiles, 4 h_prev tiles, 4 r*h_prev tiles
    constexpr int NITEMS = NA + 12;
    constexpr int PER_WAVE = (NITEMS + CF_WGRAD_WAVES - 1) / CF_WGRAD_WAVES;
    constexpr int ROWS = CIN + CF_H;
    __shared__ __attribute__((aligned(16))) float LA[16 * SA];
    __shared__ __attribute__((aligned(16))) float LB[16 * SB];
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = lane & 15, fq = lane >> 4;
    const int n_pairs = n_tiles * CF_T;
    const int p0 = blockIdx.x * pairs_per_wg;
    const int p1 = min(p0 + pairs_per_wg, n_pairs);

    f32x4 acc[MX + 4];
#pragma unroll
    for (int i = 0; i < MX + 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float bsum = 0.f;
    f32x4 pre[PER_WAVE];

    auto fetch = [&](int p) {
        const int tile = p / CF_T, t = p - tile * CF_T;
        const int tp = dir ? t + 1 : t - 1;
        const bool first = dir ? (t == CF_T - 1) : (t == 0);
        const int64_t base = (int64_t)tile * CF_T + t;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int item = wave + CF_WGRAD_WAVES * i;
            f32x4 v = {0, 0, 0, 0};
            if (item < MX) {
                v = X[(base * MX + item) * 64 + lane];
            } else if (item < NA) {
                const int m = (item - MX) & 3;
                if (!first) v = Y[(((int64_t)tile * CF_T + tp) * 8 + dir * 4 + m) * 64 + lane];
                if (item >= MX + 4) v *= S[((base * 2 + dir) * 12 + m) * 64 + lane];
            } else if (item < NITEMS) {
                v = DA[((base * 2 + dir) * 12 + (item - NA)) * 64 + lane];
            }
            pre[i] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int item = wave + CF_WGRAD_WAVES * i;
            if (item < NA) {
                const int off = item < MX ? 16 * item : CIN + 16 * (item - MX);
                *reinterpret_cast<f32x4*>(&LA[w * SA + off + 4 * fq]) = pre[i];
            } else if (item < NITEMS) {
                *reinterpret_cast<f32x4*>(&LB[w * SB + 16 * (item - NA) + 4 * fq]) = pre[i];
            }
        }
    };

    if (p0 < p1) fetch(p0);
    for (int p = p0; p < p1; ++p) {
        __syncthreads();                       // the previous pair has been consumed
        stage();
        __syncthreads();
        if (p + 1 < p1) fetch(p + 1);          // global latency hides behind the MFMAs below
        const int hoff = CIN + (wave >= 8 ? CF_H : 0);     // gates columns contract with h_prev, candidate columns with r*h_prev
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int row = 4 * kk + fq;
            const float b = LB[row * SB + 16 * wave + w];
            bsum += b;
#pragma unroll
            for (int mi = 0; mi < MX; ++mi) acc[mi] = MFMA16(LA[row * SA + 16 * mi + w], b, acc[mi]);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[MX + m] = MFMA16(LA[row * SA + hoff + 16 * m + w], b, acc[MX + m]);
        }
    }
    float* out = P + ((size_t)blockIdx.x * 2 + dir) * (ROWS + 1) * 192;
#pragma unroll
    for (int mi = 0; mi < MX + 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(16 * mi + 4 * fq + r) * 192 + 16 * wave + w] = acc[mi][r];
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (lane < 16) out[ROWS * 192 + 16 * wave + lane] = bsum;
}

// Sum of `n_parts` partial vectors (element e of part c at P[c * stride + e]) in a FIXED order: a 256-thread block owns 64
// consecutive elements, thread group g = tid >> 6 adds parts g, g + 4, ... with four independent accumulators, the four
// group sums are combined 0..3 through LDS.  Returns the total to the threads of group 0 (others get 0).
__device__ __forceinline__ float cf_reduce_parts(const float* __restrict__ P, size_t stride, int n_parts, int e, bool valid) {
    __shared__ float red[4][64];
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (valid) {
        int c = g;
        for (; c + 12 < n_parts; c += 16) {
            a0 += P[(size_t)c * stride + e];
            a1 += P[(size_t)(c + 4) * stride + e];
            a2 += P[(size_t)(c + 8) * stride + e];
            a3 += P[(size_t)(c + 12) * stride + e];
        }
        for (; c < n_parts; c += 4) a0 += P[(size_t)c * stride + e];
    }
    red[g][l] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    return g == 0 ? (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]) : 0.f;
}

// out[dir] = wg [rows][128] | bg [128] | wc [rows][64] | bc [64]   (TensorFlow variable layouts)
__global__ __launch_bounds__(256) void gru_wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, int rows, int n_chunks) {
    const int per = (rows + 1) * 192;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dir = blockIdx.y;
    const float sum = cf_reduce_parts(P + (size_t)dir * per, (size_t)2 * per, n_chunks, e, e < per);
    if (e >= per || threadIdx.x >= 64) return;
    const int row = e / 192, col = e - row * 192;
    const int bg_off = rows * 128, wc_off = bg_off + 128, bc_off = wc_off + rows * 64;
    int dst;
    if (row < rows) dst = col < 128 ? row * 128 + col : wc_off + row * 64 + (col - 128);
    else dst = col < 128 ? bg_off + col : bc_off + (col - 128);
    out[(size_t)dir * (rows * 192 + 192) + dst] = sum;
}
