// catfish_hip.hip -- hand-written CDNA4 (gfx950) kernels + C ABI for the
// homopolymer-calling forward pass (see include/catfish_hip.h for the
// reference call this replaces: catfish/models/rnn_class.py:214-216).
//
// Design (DESIGN.md has the long version):
//   * The unit of work is a TILE of 16 windows (35 samples each).  Windows are
//     independent (zero GRU state, window-local SAME padding:
//     rnn_class.py:159,170-171; infer.py:43), so a wave owns a tile for a whole
//     layer and never synchronises with another wave.
//   * Every matmul is D[feature][window] = W^T * X on v_mfma_f32_16x16x4_f32
//     (exact fp32, k-ordered fma chain).  Weights are the A operand, read from
//     LDS in pre-packed fragment order; activations are the B operand.
//   * The D register layout (col = lane&15 = window, row = 4*(lane>>4)+reg =
//     feature) is ALSO a valid B-operand layout when the k index is permuted:
//     register r of M-tile m is k-step 4m+r carrying features {16m+4q+r}.
//     The weight packer applies the same permutation, so the recurrent state,
//     r*h and the conv intermediates feed the next MFMA straight from
//     registers: no LDS round trip, no cross-lane traffic on the serial chain.
//   * Inter-layer activations live in HBM in that same "fragment" order
//     [tile][t][mtile][lane][4], so every load/store is a coalesced 16 B/lane
//     (1 KiB per wave-instruction) access.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <string>
#include <cmath>
#include <mutex>
#include <set>

#include "../../include/catfish_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CF_T 35          // window length (rnn_class.py:27)
#define CF_H 64          // GRU units per direction of the shipped model
#define CF_C 32          // conv channels of the shipped model
#define CF_TILE 16       // windows per tile = MFMA N

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Timing-only ablations (tools/ablate.sh builds variants; results are WRONG when set):
//   bit 0: skip the sigmoid/tanh VALU phases   bit 1: no per-step global loads/stores
//   bit 2: 4 waves per workgroup (one per SIMD) instead of 8
#ifndef CF_ABLATE
#define CF_ABLATE 0
#endif
#ifndef CF_PREFETCH
#define CF_PREFETCH 1        // k-steps of A-fragment prefetch in the fp32 GRU step (1 or 2)
#endif

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float cf_sigmoid(float x) {
    // tf.nn.sigmoid; v_exp_f32 + v_rcp_f32 (1 ulp each), saturates correctly at +-inf
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float cf_tanh(float x) {
    // tanh(x) = 1 - 2 / (1 + exp(2x)); absolute error ~1e-7
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f);
}
// The GRU packers pre-scale the gate weights/biases by -log2(e) and the candidate's by 2*log2(e), so
// the MFMA accumulators already hold the exp2 arguments (saves one v_mul per gate element):
//   sigmoid(x) = 1 / (1 + 2^(-x log2 e)),   tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)).
#define CF_GATE_SCALE (-1.4426950408889634)
#define CF_CAND_SCALE (2.8853900817779268)
__device__ __forceinline__ float cf_sigmoid_pre(float a) {   // a = -x * log2(e)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
__device__ __forceinline__ float cf_tanh_pre(float a) {      // a = 2 * x * log2(e)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)), 1.0f);
}
// two elements at a time: the "+ 1" and the tanh fma become packed-fp32 instructions (v_pk_add_f32 / v_pk_fma_f32)
__device__ __forceinline__ f32x2 cf_sigmoid_pre2(f32x2 a) {
    f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    e = e + (f32x2){1.0f, 1.0f};
    return (f32x2){__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
__device__ __forceinline__ f32x2 cf_tanh_pre2(f32x2 a) {
    return __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, cf_sigmoid_pre2(a), (f32x2){1.0f, 1.0f});
}
// LDS arrays larger than a ds_read immediate offset reaches (64 KiB): three base pointers kept live in registers.
// The opaque offsets stop the compiler from folding them back into base + constant, which it would re-add in
// front of every far read.  Index with compile-time constants (unrolled loops) so the selection folds away.
template <typename T>
struct far_lds {
    static constexpr int N64K = 65536 / sizeof(T);
    const T *b0, *b1, *b2;
    __device__ __forceinline__ explicit far_lds(const T* base) {
        int o1 = N64K, o2 = 2 * N64K;
        asm volatile("" : "+v"(o1), "+v"(o2));
        b0 = base; b1 = base + o1; b2 = base + o2;
    }
    __device__ __forceinline__ const T& operator[](int e) const { return e < N64K ? b0[e] : (e < 2 * N64K ? b1[e - N64K] : b2[e - 2 * N64K]); }
};
__device__ __forceinline__ f32x4 cf_sigmoid_pre4(f32x4 a) {
    f32x4 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y), __builtin_amdgcn_exp2f(a.z), __builtin_amdgcn_exp2f(a.w)};
    e = e + (f32x4){1.0f, 1.0f, 1.0f, 1.0f};
    return (f32x4){__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y), __builtin_amdgcn_rcpf(e.z), __builtin_amdgcn_rcpf(e.w)};
}
__device__ __forceinline__ f32x4 cf_tanh_pre4(f32x4 a) {
    return __builtin_elementwise_fma((f32x4){-2.0f, -2.0f, -2.0f, -2.0f}, cf_sigmoid_pre4(a), (f32x4){1.0f, 1.0f, 1.0f, 1.0f});
}
// Output dropout of the training kernels (DropoutWrapper(output_keep_prob), rnn_class.py:151-154) without a stored mask: whether
// element e of a layer's output survives is a hash of (seed, layer, optimizer step, e), evaluated where the value is produced
// (forward: the dropped copy the next layer reads) and again where its gradient arrives (backward).  keep_prob >= 1: off.
struct cf_dropout {
    float keep_prob = 1.f;
    uint32_t seed = 0;
    int32_t layer = 0;
    const double* step = nullptr;      // device: optimizer steps taken so far (a new mask every step, also under graph replay)
};
__device__ __forceinline__ uint32_t cf_fmix32(uint32_t v) {      // murmur3 finalizer: full avalanche on 32 bits
    v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
    return v;
}
__device__ __forceinline__ uint32_t cf_drop_key(const cf_dropout& d) {
    const uint32_t st = d.step ? (uint32_t)(long long)d.step[0] : 0u;
    return cf_fmix32(d.seed ^ cf_fmix32(0x9E3779B9u * (uint32_t)(d.layer + 1) + st * 0x85ebca6bu));
}
// mask / keep_prob of the four elements of f32x4 number idx4 of a layer output [tile][t][8][lane]
__device__ __forceinline__ f32x4 cf_drop_scale4(uint32_t key, float keep_prob, int64_t idx4) {
    const uint32_t thresh = (uint32_t)fminf(keep_prob * 4294967296.0f, 4294967040.0f);
    const float inv = 1.0f / keep_prob;
    const uint64_t e = (uint64_t)idx4 * 4u;
    const uint32_t base = (uint32_t)e ^ ((uint32_t)(e >> 32) * 0x9E3779B9u);
    f32x4 o;
    o.x = cf_fmix32((base + 0u) ^ key) < thresh ? inv : 0.f;
    o.y = cf_fmix32((base + 1u) ^ key) < thresh ? inv : 0.f;
    o.z = cf_fmix32((base + 2u) ^ key) < thresh ? inv : 0.f;
    o.w = cf_fmix32((base + 3u) ^ key) < thresh ? inv : 0.f;
    return o;
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    f32x4 o;
    o.x = fmaxf(v.x, 0.f); o.y = fmaxf(v.y, 0.f); o.z = fmaxf(v.z, 0.f); o.w = fmaxf(v.w, 0.f);
    return o;
}

// ------------------------------------------------------------------------------------------
// Packed-weight geometry shared by host packer and kernels
// ------------------------------------------------------------------------------------------
// GRU direction-layer blob (floats):
//   X  region: [ks < CIN/4][g < 3][lane][j < 4]   A fragments of output M-tile 4g+j, k-step ks (x rows)
//   HG region: [ks < 16][g < 2][lane][j]          gates (r,u) M-tiles 0..7, h rows
//   HC region: [ks < 16][lane][j]                 candidate M-tiles 8..11, h rows (multiply r*h)
//   BIAS     : [mo < 12][q < 4][r < 4]            bias of feature 16mo+4q+r   (gates | candidate)
//   DENSE    : [m < 4][q][r]                      final_fully_connected weight of this direction
// Output M-tiles: 0..3 = r gate, 4..7 = u gate, 8..11 = candidate.
__host__ __device__ constexpr int gru_x_floats(int cin) { return (cin / 4) * 3 * 256; }
__host__ __device__ constexpr int gru_hg_floats() { return 16 * 2 * 256; }
__host__ __device__ constexpr int gru_hc_floats() { return 16 * 256; }
__host__ __device__ constexpr int gru_bias_off(int cin) { return gru_x_floats(cin) + gru_hg_floats() + gru_hc_floats(); }
__host__ __device__ constexpr int gru_dense_off(int cin) { return gru_bias_off(cin) + 192; }
__host__ __device__ constexpr int gru_pack_floats(int cin) { return gru_dense_off(cin) + 64; }

// Residual-block blob (floats): NU 32x32 units of 1024 floats [ks < 8][lane][mo < 2], then
// bias vectors of 32 floats each in [mo][q][r] order.
//   first block (Cin = 1): units {c3 tap0, tap1, tap2, last}; vectors {b_c3, b_last, w_sc, b_sc, w_first, b_first}
//   other blocks          : units {sc, first, c3 tap0, tap1, tap2, last}; vectors {b_sc, b_first, b_c3, b_last}
__host__ __device__ constexpr int res_units(bool first) { return first ? 4 : 6; }
__host__ __device__ constexpr int res_vecs(bool first) { return first ? 6 : 4; }
__host__ __device__ constexpr int res_pack_floats(bool first) { return res_units(first) * 1024 + res_vecs(first) * 32; }

// feature carried by lane-quarter q in k-step ks (== register ks&3 of M-tile ks>>2 of a D tile)
__host__ __device__ constexpr int frag_feature(int ks, int q) { return 16 * (ks >> 2) + 4 * q + (ks & 3); }

// ------------------------------------------------------------------------------------------
// Kernel 1: residual block (resnet_class.py:44-82), BN folded into the convs.
// One wave = one tile of 16 windows, streamed over t with a one-step lookahead
// for the k=3 conv (zero padding at both window edges).  Small calls (latency mode) split a tile's 35
// positions into `t_chunks` chunks, one wave each: a chunk starts one position early to rebuild the k=3
// conv's left neighbour, otherwise the stream is the same.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit_mma(const float* __restrict__ unit, int lane, const f32x4 (&in)[2], f32x4 (&acc)[2]) {
    const f32x2* u2 = reinterpret_cast<const f32x2*>(unit) + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f32x2 a = u2[ks * 64];
        const float b = in[ks >> 2][ks & 3];
        acc[0] = MFMA16(a.x, b, acc[0]);
        acc[1] = MFMA16(a.y, b, acc[1]);
    }
}

template <bool FIRST>
__global__ __launch_bounds__(256) void res_block_kernel(const float* __restrict__ wpack,
                                                        const float* __restrict__ x_nat,   // FIRST: [n_windows, 35]
                                                        const f32x4* __restrict__ x_frag,  // !FIRST: [tile][t][2][lane]
                                                        f32x4* __restrict__ y_frag,        // [tile][t][2][lane]
                                                        int64_t n_windows, int n_tiles, int t_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK = res_pack_floats(FIRST);
    constexpr int VEC0 = res_units(FIRST) * 1024;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const int waves_per_block = blockDim.x >> 6;
    float* xs = lds + PACK + wave * (CF_TILE * CF_T);  // FIRST only: this wave's x tile [16][35]

    auto vec = [&](int v, int mo) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + VEC0 + v * 32 + mo * 16 + q * 4);
    };

    const int chunk_len = (CF_T + t_chunks - 1) / t_chunks;
    for (int task = blockIdx.x * waves_per_block + wave; task < n_tiles * t_chunks; task += gridDim.x * waves_per_block) {
        const int tile = task / t_chunks;
        const int p0 = (task - tile * t_chunks) * chunk_len;          // this wave writes positions [p0, p1)
        const int p1 = min(p0 + chunk_len, CF_T);
        f32x4 w_sc[2], b_sc[2], w_f[2], b_f[2];
        if constexpr (FIRST) {
            // stage the tile's raw samples: 16 windows x 35 = 560 contiguous floats
            const int64_t base = (int64_t)tile * CF_TILE * CF_T;
            const int64_t limit = n_windows * CF_T;
            for (int i = lane; i < CF_TILE * CF_T; i += 64) xs[i] = (base + i < limit) ? x_nat[base + i] : 0.f;
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) { w_sc[mo] = vec(2, mo); b_sc[mo] = vec(3, mo); w_f[mo] = vec(4, mo); b_f[mo] = vec(5, mo); }
            // xs is private to this wave: a wave-level fence orders its ds_writes before the reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        f32x4 o1_pp[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        f32x4 o1_p[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        f32x4 sc_p[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int i = p0 > 0 ? p0 - 1 : 0; i <= p1; ++i) {
            f32x4 o1_c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            f32x4 sc_c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            if (i < CF_T) {
                if constexpr (FIRST) {
                    // Cin = 1: conv1d is a rank-1 update (resnet_class.py:60,64)
                    const float xv = xs[(lane & 15) * CF_T + i];
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) {
                        sc_c[mo] = w_sc[mo] * xv + b_sc[mo];
                        o1_c[mo] = relu4(w_f[mo] * xv + b_f[mo]);
                    }
                } else {
                    f32x4 in[2];
                    const f32x4* src = x_frag + ((int64_t)tile * CF_T + i) * 2 * 64 + lane;
                    in[0] = src[0];
                    in[1] = src[64];
                    sc_c[0] = vec(0, 0); sc_c[1] = vec(0, 1);
                    unit_mma(lds + 0 * 1024, lane, in, sc_c);                 // shortcut, no relu (:60-61)
                    f32x4 acc[2] = {vec(1, 0), vec(1, 1)};
                    unit_mma(lds + 1 * 1024, lane, in, acc);                  // first conv (:64-66)
                    o1_c[0] = relu4(acc[0]); o1_c[1] = relu4(acc[1]);
                }
            }
            if (i >= p0 + 1) {
                constexpr int U3 = FIRST ? 0 : 2;      // first k=3 tap unit
                constexpr int VB3 = FIRST ? 0 : 2;     // bias vector of the k=3 conv
                f32x4 acc[2] = {vec(VB3, 0), vec(VB3, 1)};
                unit_mma(lds + (U3 + 0) * 1024, lane, o1_pp, acc);            // tap 0 * o1[t-1]
                unit_mma(lds + (U3 + 1) * 1024, lane, o1_p, acc);             // tap 1 * o1[t]
                unit_mma(lds + (U3 + 2) * 1024, lane, o1_c, acc);             // tap 2 * o1[t+1]
                f32x4 o2[2] = {relu4(acc[0]), relu4(acc[1])};                 // (:69-71)
                f32x4 acc3[2] = {vec(VB3 + 1, 0), vec(VB3 + 1, 1)};
                unit_mma(lds + (U3 + 3) * 1024, lane, o2, acc3);              // last conv (:74-76)
                f32x4* dst = y_frag + ((int64_t)tile * CF_T + (i - 1)) * 2 * 64 + lane;
                dst[0] = relu4(relu4(acc3[0]) + sc_p[0]);                     // add + relu (:79-80)
                dst[64] = relu4(relu4(acc3[1]) + sc_p[1]);
            }
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) { o1_pp[mo] = o1_p[mo]; o1_p[mo] = o1_c[mo]; sc_p[mo] = sc_c[mo]; }
        }
        if constexpr (FIRST) __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Kernel 1c: the first TWO residual blocks in one launch (throughput mode).  Block 1 runs one position behind
// block 0 in the same stream, so block 0's 32-channel output goes from its accumulators straight into block 1's
// first MFMAs (the accumulator layout is a B-operand layout, see the header) and never exists in HBM: 256 B per
// sample of traffic and one launch less.  Same operations in the same order as two res_block_kernel launches:
// bit-identical output.  Small calls (latency mode) split a tile's 35 positions into `t_chunks` chunks, one wave each, on
// as many CUs as the call leaves idle: a chunk starts two positions early (block 0's k=3 neighbourhood, then block 1's),
// so a single read's conv stack is ~7 positions deep instead of 2 x 35; there block 0's output is also stored (y0_frag)
// for the debug / test hook.
// ------------------------------------------------------------------------------------------
template <bool CHUNKED>
__global__ __launch_bounds__(256) void res_stack2_kernel(const float* __restrict__ wpack0, const float* __restrict__ wpack1,
                                                         const float* __restrict__ x_nat,   // [n_windows, 35]
                                                         f32x4* __restrict__ y_frag,        // [tile][t][2][lane], block 1's output
                                                         f32x4* __restrict__ y0_frag,       // block 0's output, or null
                                                         int64_t n_windows, int n_tiles, int t_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK0 = res_pack_floats(true), PACK1 = res_pack_floats(false);
    constexpr int VEC0 = res_units(true) * 1024, VEC1 = res_units(false) * 1024;
    {
        const f32x4* src0 = reinterpret_cast<const f32x4*>(wpack0);
        const f32x4* src1 = reinterpret_cast<const f32x4*>(wpack1);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK0 / 4; i += blockDim.x) dst[i] = src0[i];
        for (int i = threadIdx.x; i < PACK1 / 4; i += blockDim.x) dst[PACK0 / 4 + i] = src1[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const int waves_per_block = blockDim.x >> 6;
    const float* w0 = lds;                 // block 0: units {c3 tap0, tap1, tap2, last}; vectors {b_c3, b_last, w_sc, b_sc, w_first, b_first}
    const float* w1 = lds + PACK0;         // block 1: units {sc, first, c3 tap0, tap1, tap2, last}; vectors {b_sc, b_first, b_c3, b_last}
    float* xs = lds + PACK0 + PACK1 + wave * (CF_TILE * CF_T);      // this wave's x tile [16][35]
    auto vec0 = [&](int v, int mo) -> f32x4 { return *reinterpret_cast<const f32x4*>(w0 + VEC0 + v * 32 + mo * 16 + q * 4); };
    auto vec1 = [&](int v, int mo) -> f32x4 { return *reinterpret_cast<const f32x4*>(w1 + VEC1 + v * 32 + mo * 16 + q * 4); };
    const f32x4 zero = {0, 0, 0, 0};

    if (!CHUNKED) t_chunks = 1;                                       // throughput mode: the chunk arithmetic folds away
    const int chunk_len = CHUNKED ? (CF_T + t_chunks - 1) / t_chunks : CF_T;
    for (int task = blockIdx.x * waves_per_block + wave; task < n_tiles * t_chunks; task += gridDim.x * waves_per_block) {
        const int tile = CHUNKED ? task / t_chunks : task;
        const int p0 = CHUNKED ? (task - tile * t_chunks) * chunk_len : 0;      // this wave writes positions [p0, p1)
        const int p1 = CHUNKED ? min(p0 + chunk_len, CF_T) : CF_T;
        if (CHUNKED && p0 >= p1) continue;
        {
            const int64_t base = (int64_t)tile * CF_TILE * CF_T;
            const int64_t limit = n_windows * CF_T;
            for (int i = lane; i < CF_TILE * CF_T; i += 64) xs[i] = (base + i < limit) ? x_nat[base + i] : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        f32x4 w_sc[2], b_sc[2], w_f[2], b_f[2];
#pragma unroll
        for (int mo = 0; mo < 2; ++mo) { w_sc[mo] = vec0(2, mo); b_sc[mo] = vec0(3, mo); w_f[mo] = vec0(4, mo); b_f[mo] = vec0(5, mo); }
        f32x4 a_pp[2] = {zero, zero}, a_p[2] = {zero, zero}, a_sp[2] = {zero, zero};     // block 0: o1[t-1], o1[t], shortcut[t]
        f32x4 b_pp[2] = {zero, zero}, b_p[2] = {zero, zero}, b_sp[2] = {zero, zero};     // block 1, one position behind
        for (int i = p0 > 2 ? p0 - 2 : 0; i <= p1 + 1; ++i) {
            // ---- block 0, front at position i: Cin = 1, conv1d is a rank-1 update (resnet_class.py:60,64)
            f32x4 a_c[2] = {zero, zero}, a_sc[2] = {zero, zero};
            if (i < CF_T) {
                const float xv = xs[(lane & 15) * CF_T + i];
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) {
                    a_sc[mo] = w_sc[mo] * xv + b_sc[mo];
                    a_c[mo] = relu4(w_f[mo] * xv + b_f[mo]);
                }
            }
            // ---- block 0, back: its output at position i - 1 stays in registers as block 1's input
            f32x4 y0[2] = {zero, zero};
            const bool have_y0 = i >= 1 && i <= CF_T && i >= p0;       // block 0's output at i - 1 >= p0 - 1
            if (have_y0) {
                f32x4 acc[2] = {vec0(0, 0), vec0(0, 1)};
                unit_mma(w0 + 0 * 1024, lane, a_pp, acc);                 // tap 0 * o1[t-1]
                unit_mma(w0 + 1 * 1024, lane, a_p, acc);                  // tap 1 * o1[t]
                unit_mma(w0 + 2 * 1024, lane, a_c, acc);                  // tap 2 * o1[t+1]
                f32x4 o2[2] = {relu4(acc[0]), relu4(acc[1])};             // (:69-71)
                f32x4 acc3[2] = {vec0(1, 0), vec0(1, 1)};
                unit_mma(w0 + 3 * 1024, lane, o2, acc3);                  // last conv (:74-76)
                y0[0] = relu4(relu4(acc3[0]) + a_sp[0]);                  // add + relu (:79-80)
                y0[1] = relu4(relu4(acc3[1]) + a_sp[1]);
                if (CHUNKED && y0_frag && i - 1 >= p0 && i - 1 < p1) {
                    f32x4* d0 = y0_frag + ((int64_t)tile * CF_T + (i - 1)) * 2 * 64 + lane;
                    d0[0] = y0[0];
                    d0[64] = y0[1];
                }
            }
            // ---- block 1, front at position p = i - 1 (zero padding past the window end, p = T)
            f32x4 b_c[2] = {zero, zero}, b_sc[2] = {zero, zero};
            if (have_y0) {
                b_sc[0] = vec1(0, 0); b_sc[1] = vec1(0, 1);
                unit_mma(w1 + 0 * 1024, lane, y0, b_sc);                  // shortcut, no relu (:60-61)
                f32x4 acc[2] = {vec1(1, 0), vec1(1, 1)};
                unit_mma(w1 + 1 * 1024, lane, y0, acc);                   // first conv (:64-66)
                b_c[0] = relu4(acc[0]); b_c[1] = relu4(acc[1]);
            }
            // ---- block 1, back: output position p - 1 = i - 2
            if (i >= 2 && i - 2 >= p0 && i - 2 < p1) {
                f32x4 acc[2] = {vec1(2, 0), vec1(2, 1)};
                unit_mma(w1 + 2 * 1024, lane, b_pp, acc);
                unit_mma(w1 + 3 * 1024, lane, b_p, acc);
                unit_mma(w1 + 4 * 1024, lane, b_c, acc);
                f32x4 o2[2] = {relu4(acc[0]), relu4(acc[1])};
                f32x4 acc3[2] = {vec1(3, 0), vec1(3, 1)};
                unit_mma(w1 + 5 * 1024, lane, o2, acc3);
                f32x4* dst = y_frag + ((int64_t)tile * CF_T + (i - 2)) * 2 * 64 + lane;
                dst[0] = relu4(relu4(acc3[0]) + b_sp[0]);
                dst[64] = relu4(relu4(acc3[1]) + b_sp[1]);
            }
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                a_pp[mo] = a_p[mo]; a_p[mo] = a_c[mo]; a_sp[mo] = a_sc[mo];
                if (i >= 1) { b_pp[mo] = b_p[mo]; b_p[mo] = b_c[mo]; b_sp[mo] = b_sc[mo]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Kernel 2: one bidirectional GRU layer (rnn_class.py:142-148,165-171), both
// directions in one grid (blockIdx.y).  A workgroup stages its direction's
// packed weights in LDS once; each wave then owns tiles of 16 windows and runs
// the 35-step recurrence with h, r*h and all pre-activations in registers.
// LAST = true fuses the dense(128->1) partial dot (rnn_class.py:179) instead
// of writing h.
// ------------------------------------------------------------------------------------------
template <int CIN>
__device__ __forceinline__ void gru_stage_weights(float* lds, const float* __restrict__ wpack, int dir) {
    constexpr int PACK = gru_pack_floats(CIN);
    const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = threadIdx.x; i < PACK / 4; i += blockDim.x) dst[i] = src[i];
}

// One tile (16 windows) of one direction of one layer: the whole 35-step recurrence.
// STASH (training forward): also write the activated gates r, u and the candidate c of every step to
// S[tile][t][dir][12][lane] (r = 0..3, u = 4..7, c = 8..11) for the backward pass.
// One k-step region: the A-fragment reads of the NEXT k-step go out before this k-step's MFMAs (left alone the
// scheduler sinks them behind most of the MFMAs, which leaves ~100 cycles between a read and its first use).
#ifndef CF_SCHED_DS_FIRST
#define CF_SCHED_DS_FIRST 1
#endif
#if CF_SCHED_DS_FIRST
#define CF_KSTEP_SCHED(nds, nmfma)                               \
    __builtin_amdgcn_sched_group_barrier(0x100, nds, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x008, nmfma, 0);       \
    __builtin_amdgcn_sched_barrier(0)
#else
#define CF_KSTEP_SCHED(nds, nmfma) __builtin_amdgcn_sched_barrier(0)
#endif
template <int CIN, bool LAST, bool STASH = false>
__device__ __forceinline__ void gru_tile(const float* lds, int lane, int dir, int tile, const f32x4* __restrict__ X,
                                         f32x4* __restrict__ Y, float* __restrict__ P, int n_tiles,
                                         f32x4* __restrict__ S = nullptr, f32x4* __restrict__ YD = nullptr, cf_dropout drop = cf_dropout()) {
    const uint32_t drop_key = (STASH && YD) ? cf_drop_key(drop) : 0u;
    constexpr int KGX = CIN / 16;   // f32x4 registers of x per lane and step
    constexpr int KSX = CIN / 4;    // k-steps of the x part
    constexpr int XN4 = gru_x_floats(CIN) / 4;        // region sizes in f32x4 units
    constexpr int HG4 = gru_hg_floats() / 4;
    constexpr int BIAS = gru_bias_off(CIN);
    constexpr int DENSE = gru_dense_off(CIN);
    const int q = lane >> 4;
    const f32x4* WX = reinterpret_cast<const f32x4*>(lds) + lane;     // + (ks*3+g)*64
    const far_lds<f32x4> WL(WX);      // the packed weights span up to 144 KiB (far_lds: 32 v_add per step less, 168 -> 122 VGPRs)
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + BIAS) + q;  // + mo*4
    const f32x4* D4 = reinterpret_cast<const f32x4*>(lds + DENSE) + q; // + m*4
    {
        f32x4 h[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // GRUCellZeroState
        f32x4 xc[KGX];
        {
            const int t0 = dir ? (CF_T - 1) : 0;
            const f32x4* src = X + ((int64_t)tile * CF_T + t0) * KGX * 64 + lane;
#pragma unroll
            for (int g = 0; g < KGX; ++g) xc[g] = src[g * 64];
        }
        for (int s = 0; s < CF_T; ++s) {
            const int t = dir ? (CF_T - 1 - s) : s;    // bw = time-reversed sequence (ReverseV2)
            f32x4 acc[12];
#pragma unroll
            for (int mo = 0; mo < 12; ++mo) acc[mo] = B4[mo * 4];     // (bias held in registers as the first C operand: measured neutral)
            // The A fragments are software-pipelined one k-step ahead by hand; the
            // sched_barriers keep hipcc from hoisting hundreds of ds_reads (it spills otherwise).
            // k-steps of the whole step form one sequence p: x part [0, KSX), gate h part [KSX, KSX+16), candidate
            // h part [KSX+16, KSX+32); fragments are fetched PF k-steps ahead.
            constexpr int PF = CF_PREFETCH;
            auto loadA = [&](int p, f32x4 (&d)[3]) {
                if (p < KSX) { d[0] = WL[(p * 3 + 0) * 64]; d[1] = WL[(p * 3 + 1) * 64]; d[2] = WL[(p * 3 + 2) * 64]; }
                else if (p < KSX + 16) { d[0] = WL[XN4 + ((p - KSX) * 2 + 0) * 64]; d[1] = WL[XN4 + ((p - KSX) * 2 + 1) * 64]; }
                else if (p < KSX + 32) { d[0] = WL[XN4 + HG4 + (p - KSX - 16) * 64]; }
            };
            f32x4 ac[3], an[3], a2[3];
            loadA(0, ac);
            if (PF == 2) loadA(1, an);
            // x part: [r | u | c] += Wx^T x_t
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                if (PF == 2) loadA(ks + 2, a2); else loadA(ks + 1, an);
                const float b = xc[ks >> 2][ks & 3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    acc[4 * g + 0] = MFMA16(ac[g].x, b, acc[4 * g + 0]);
                    acc[4 * g + 1] = MFMA16(ac[g].y, b, acc[4 * g + 1]);
                    acc[4 * g + 2] = MFMA16(ac[g].z, b, acc[4 * g + 2]);
                    acc[4 * g + 3] = MFMA16(ac[g].w, b, acc[4 * g + 3]);
                }
                CF_KSTEP_SCHED(3, 12);
#pragma unroll
                for (int g = 0; g < 3; ++g) { ac[g] = an[g]; if (PF == 2) an[g] = a2[g]; }
            }
            // x_t is dead: fetch the next step's x into the same registers (clamped on the last
            // step: a harmless re-read); the h part below hides the latency.
            {
                int tn = dir ? (t - 1) : (t + 1);
                tn = tn < 0 ? 0 : (tn > CF_T - 1 ? CF_T - 1 : tn);
                const f32x4* src = X + ((int64_t)tile * CF_T + tn) * KGX * 64 + lane;
#pragma unroll
                for (int g = 0; g < KGX; ++g) if (!(CF_ABLATE & 2)) xc[g] = src[g * 64];
            }
            // h part of the gates: [r | u] += Wh_g^T h          (gru_cell/MatMul)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (PF == 2) loadA(KSX + ks + 2, a2); else loadA(KSX + ks + 1, an);
                const float b = h[ks >> 2][ks & 3];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc[4 * g + 0] = MFMA16(ac[g].x, b, acc[4 * g + 0]);
                    acc[4 * g + 1] = MFMA16(ac[g].y, b, acc[4 * g + 1]);
                    acc[4 * g + 2] = MFMA16(ac[g].z, b, acc[4 * g + 2]);
                    acc[4 * g + 3] = MFMA16(ac[g].w, b, acc[4 * g + 3]);
                }
                CF_KSTEP_SCHED(2, 8);
                ac[0] = an[0]; ac[1] = an[1];
                if (PF == 2) { an[0] = a2[0]; an[1] = a2[1]; }
            }
            // r = sigmoid(.), r*h feeds the candidate matmul (reset applied BEFORE the matmul:
            // gru_cell/mul -> concat_1 -> MatMul_1)
            f32x4 rh[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (CF_ABLATE & 1) { rh[m] = acc[m] * 0.001f; continue; }
                const f32x4 rv = cf_sigmoid_pre4(acc[m]);            // packed-fp32 "+ 1" and products, same rounding as scalar
                rh[m] = rv * h[m];
                if constexpr (STASH) acc[m] = rv;
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (PF == 2) loadA(KSX + 16 + ks + 2, a2); else loadA(KSX + 16 + ks + 1, an);
                const float b = rh[ks >> 2][ks & 3];
                acc[8] = MFMA16(ac[0].x, b, acc[8]);
                acc[9] = MFMA16(ac[0].y, b, acc[9]);
                acc[10] = MFMA16(ac[0].z, b, acc[10]);
                acc[11] = MFMA16(ac[0].w, b, acc[11]);
                CF_KSTEP_SCHED(1, 4);
                ac[0] = an[0];
                if (PF == 2) an[0] = a2[0];
            }
            // h' = u*h + (1-u)*c                                  (gru_cell/mul_1, sub, mul_2, add)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (CF_ABLATE & 1) { h[m] = acc[4 + m] * 0.001f + acc[8 + m] * 0.001f; continue; }
                const f32x4 u = cf_sigmoid_pre4(acc[4 + m]);
                const f32x4 c = cf_tanh_pre4(acc[8 + m]);
                h[m] = __builtin_elementwise_fma(u, h[m] - c, c);
                if constexpr (STASH) { acc[4 + m] = u; acc[8 + m] = c; }
            }
            if constexpr (STASH) {
                f32x4* sdst = S + (((int64_t)tile * CF_T + t) * 2 + dir) * 12 * 64 + lane;
#pragma unroll
                for (int j = 0; j < 12; ++j) sdst[j * 64] = acc[j];
            }
            if constexpr (!LAST) {
                f32x4* dst = Y + (((int64_t)tile * CF_T + t) * 8 + dir * 4) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 4; ++m) if (!(CF_ABLATE & 2) || s == CF_T - 1) dst[m * 64] = h[m];
                if constexpr (STASH) {
                    if (YD) {       // training with dropout: the copy the next layer (or the dense head) reads
                        const int64_t i0 = (((int64_t)tile * CF_T + t) * 8 + dir * 4) * 64 + lane;
#pragma unroll
                        for (int m = 0; m < 4; ++m) YD[i0 + m * 64] = h[m] * cf_drop_scale4(drop_key, drop.keep_prob, i0 + m * 64);
                    }
                }
            } else {
                // partial logit of this direction: sum_f w[f] * h[f]   (final_fully_connected/MatMul)
                // summation order shared with the cooperative kernel (bit-identical results): one fma chain per
                // M-tile, the four M-tile sums added in order, then the lane quarters
                float pm[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 wd = D4[m * 4];
                    pm[m] = wd.x * h[m].x;
                    pm[m] = fmaf(wd.y, h[m].y, pm[m]); pm[m] = fmaf(wd.z, h[m].z, pm[m]); pm[m] = fmaf(wd.w, h[m].w, pm[m]);
                }
                float p = ((pm[0] + pm[1]) + pm[2]) + pm[3];
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                if (lane < 16) P[(((int64_t)dir * n_tiles + tile) * CF_T + t) * 16 + lane] = p;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int CIN, bool LAST>
__global__ __launch_bounds__(512, 2) void gru_layer_kernel(const float* __restrict__ wpack,  // [2][gru_pack_floats(CIN)]
                                                           const f32x4* __restrict__ X,      // [tile][t][CIN/16][lane]
                                                           f32x4* __restrict__ Y,            // [tile][t][8][lane]
                                                           float* __restrict__ P,            // [2][tile][t][16]
                                                           int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dir = blockIdx.y;
    gru_stage_weights<CIN>(lds, wpack, dir);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // Every workgroup takes an equal contiguous share of the tiles (+-1) and deals it round-robin to its waves, so that all
    // SIMDs of the chip end within one tile of each other whatever the wave count.
    const int t0 = (int)((int64_t)blockIdx.x * n_tiles / gridDim.x);
    const int t1 = (int)((int64_t)(blockIdx.x + 1) * n_tiles / gridDim.x);
    for (int tile = t0 + wave; tile < t1; tile += nwaves)
        gru_tile<CIN, LAST>(lds, lane, dir, tile, X, Y, P, n_tiles);
}

// ------------------------------------------------------------------------------------------
// Kernel 2b: all biGRU layers in ONE launch.  A separate launch per layer leaves CUs idle in every
// layer's last round (1888 tiles per direction on 1024 SIMDs = 3.69 rounds -> 7.8 % lost per launch).
// Here workgroups are ordered layer-major (a pool per layer and direction that pulls 8-tile groups from a
// queue), so the first groups of layer l+1 start on the CUs that layer l's tail frees.  A group of layer
// l > 0 waits for both directions of the same group of layer l-1 through agent-scope flags
// (release/acquire exactly as cdna_hip_programming.md Guideline 16).  Every wait is bounded: on a
// timeout the workgroup records an error code and carries on, so the grid always drains.
// Progress argument: workgroups are dispatched in index order per XCD, so when a waiting workgroup holds
// a CU every workgroup of the previous layer has already been dispatched and none of those ever waits on
// a later one.
// ------------------------------------------------------------------------------------------
struct cf_fused_args {
    const float* w[3];       // packed weights of layer 0..2, [2 dirs]
    const f32x4* x0;         // layer-0 input (conv output / embedding), fragment layout
    f32x4* y[2];             // layer outputs (layer l writes y[l & 1])
    float* p;                // dense partials
    unsigned* flags;         // [n_layers][groups][2] completion flags, then [n_layers][2] queue heads; zeroed before the launch
    int lds_floats;          // floats of dynamic LDS holding weights (one broadcast word follows)
    unsigned* err;           // [1], zeroed before the launch; != 0 after a wait timed out
    int n_tiles;
    int groups;              // ceil(n_tiles / 8)
    int n_layers;            // 1..3
    int cin0;                // 32 (ResNetRNN) or 16 (plain RNN)
};

__device__ __forceinline__ bool cf_wait_flag(unsigned* flag) {
    for (int it = 0; it < 1000000; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        __builtin_amdgcn_s_sleep(16);
    }
    return false;
}

// A workgroup serves ONE (layer, direction): it stages those weights once, then pulls 8-tile groups from
// that (layer, direction)'s queue until it is empty.
template <int CIN, bool LAST>
__device__ __forceinline__ void gru_fused_worker(float* lds, const cf_fused_args& a, int layer, int dir) {
    const f32x4* X = layer == 0 ? a.x0 : a.y[(layer - 1) & 1];
    f32x4* Y = a.y[layer & 1];
    gru_stage_weights<CIN>(lds, a.w[layer], dir);
    volatile int* slot = reinterpret_cast<volatile int*>(lds + a.lds_floats);   // broadcast word behind the weights
    unsigned* queue = a.flags + (size_t)a.n_layers * a.groups * 2 + layer * 2 + dir;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A pool member serves at most `groups` groups before it sees the queue empty: the loop is bounded by
    // construction (a plain `for (;;)` with a break in the middle hung on gfx950 / ROCm 7.2).
    int group = 0;
    for (int iter = 0; iter <= a.groups && group < a.groups; ++iter) {
        if (threadIdx.x == 0) {
            const int g = (int)__hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (layer > 0 && g < a.groups) {
                // consumer side: ONE lane polls relaxed, then ONE agent-scope acquire
                unsigned* f = a.flags + ((size_t)(layer - 1) * a.groups + g) * 2;
                if (!cf_wait_flag(f) || !cf_wait_flag(f + 1)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *slot = g;
        }
        __syncthreads();
        group = *slot;                                   // the same value in every thread
        if (group < a.groups) {
            const int tile = group * 8 + wave;
            if (tile < a.n_tiles) gru_tile<CIN, LAST>(lds, lane, dir, tile, X, Y, a.p, a.n_tiles);
            // producer side: every storing wave drains, barrier, ONE lane releases at agent scope, then the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!LAST && threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + ((size_t)layer * a.groups + group) * 2 + dir, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void gru_fused_kernel(cf_fused_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int per_layer = gridDim.x / a.n_layers;          // workgroups per layer, directions interleaved
    const int layer = blockIdx.x / per_layer;
    const int dir = (blockIdx.x - layer * per_layer) & 1;
    const bool last = layer == a.n_layers - 1;
    if (layer == 0) {
        if (a.cin0 == 16) { if (last) gru_fused_worker<16, true>(lds, a, layer, dir); else gru_fused_worker<16, false>(lds, a, layer, dir); }
        else { if (last) gru_fused_worker<32, true>(lds, a, layer, dir); else gru_fused_worker<32, false>(lds, a, layer, dir); }
    } else {
        if (last) gru_fused_worker<128, true>(lds, a, layer, dir); else gru_fused_worker<128, false>(lds, a, layer, dir);
    }
}

#define CF_COOP_XCH_FLOATS (2 * 4 * 64 * 4 + 4 * 64)   // LDS exchange area of the cooperative kernel: h, r*h, dense partials
#include "gru_coop.hpp"
#include "gru_train.hpp"
static_assert(gtb_pack_floats(32) == ((32 + CF_H) / 16 / 2) * 128 * 48 && gtb_pack_floats(128) == ((128 + CF_H) / 16 / 2) * 128 * 48,
              "gru_train_bwd_coop_kernel's PACK must equal gtb_pack_floats");
#define CF_COOP_BWD_XCH_FLOATS (3 * 4 * 64 * 4)   // da_c, da_r, da_u exchange tiles
#include "gru_wgrad.hpp"
#include "res_train.hpp"
#include "train_step.hpp"

// ------------------------------------------------------------------------------------------
// Kernel 1b: plain RNN type (no residual blocks, rnn_class.py:165-175 applied to the raw signal).
// The GRU kernels want >= 4 input features per k-step, so the single input feature is embedded
// in a 16-feature fragment tile (feature 0 = x, the rest 0; the packed weight rows are 0 too).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ x_nat, f32x4* __restrict__ y_frag,
                                                    int64_t n_windows, int n_tiles) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (tile, t, lane)
    if (idx >= (int64_t)n_tiles * CF_T * 64) return;
    const int lane = (int)(idx & 63);
    const int64_t tt = idx >> 6;
    const int t = (int)(tt % CF_T);
    const int64_t w = (tt / CF_T) * CF_TILE + (lane & 15);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((lane >> 4) == 0 && w < n_windows) v.x = x_nat[w * CF_T + t];
    y_frag[idx] = v;
}

#include "gru_bf16.hpp"
#include "gru_bf16_pipe.hpp"
#include "gru_bf16x3_pipe.hpp"

// ------------------------------------------------------------------------------------------
// Kernel 3: head -- logits = p_fw + p_bw + b, probs = sigmoid (rnn_class.py:84,179-181),
// transposing the [tile][t][16] partials back to the reference's window-major order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ P, float bias, float* __restrict__ probs,
                                                   float* __restrict__ logits, int64_t n_windows, int n_tiles, int tile_shift, int raw) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_windows * CF_T) return;
    const int64_t w = idx / CF_T;
    const int t = (int)(idx - w * CF_T);
    const int64_t tile = w >> tile_shift;            // 16-window tiles (fp32 path) or 32-window tiles (bf16 path)
    const int tw = 1 << tile_shift;
    const int wl = (int)(w & (tw - 1));
    float z;
    if (raw) {
        // latency-mode kernels leave per-lane partials [dir][tile][t][4 M-tiles][64 lanes]: M-tile sums added in order, then
        // the lane quarters as the one-wave kernel's two xor-shuffles do: (q0 + q1) + (q2 + q3)
        float pd[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float* src = P + ((((int64_t)d * n_tiles + tile) * CF_T + t) * 4) * 64 + wl;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ((src[q * 16] + src[64 + q * 16]) + src[128 + q * 16]) + src[192 + q * 16];
            pd[d] = (v[0] + v[1]) + (v[2] + v[3]);
        }
        z = pd[0] + pd[1] + bias;
    } else {
        z = P[((tile)*CF_T + t) * tw + wl] + P[(((int64_t)n_tiles + tile) * CF_T + t) * tw + wl] + bias;
    }
    if (logits) logits[idx] = z;
    if (probs) probs[idx] = 1.0f / (1.0f + expf(-z));
}

// ------------------------------------------------------------------------------------------
// Kernel 4: class_from_threshold + correct_short (infer.py:128-138,174-198) per read.
// Reads are packed back to back WITH their zero padding: read r owns samples
// [read_offsets[r], read_offsets[r+1]) of which the first read_lengths[r] are real
// (infer.py:47 trims the padding before thresholding).  A positive sample survives iff its
// positive run inside the real part of its own read has length >= min_run.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ probs,
                                                          const int64_t* __restrict__ read_offsets,
                                                          const int64_t* __restrict__ read_lengths, int64_t n_reads,
                                                          int64_t total, float threshold, int min_run,
                                                          uint8_t* __restrict__ labels) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    uint8_t out = 0;
    if (probs[i] >= threshold) {
        // read containing i: largest r with read_offsets[r] <= i
        int64_t lo = 0, hi = n_reads;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (read_offsets[mid] <= i) lo = mid; else hi = mid;
        }
        const int64_t beg = read_offsets[lo], end = beg + read_lengths[lo];
        if (i < end) {
            int run = 1;
            for (int64_t j = i - 1; j >= beg && run < min_run && probs[j] >= threshold; --j) ++run;
            for (int64_t j = i + 1; j < end && run < min_run && probs[j] >= threshold; ++j) ++run;
            out = run >= min_run ? 1 : 0;
        }
    }
    labels[i] = out;
}

// ------------------------------------------------------------------------------------------
// Kernel 4b: run boundaries of the corrected labels (first half of hp_in_pred, infer.py:141-162).
// Every sample that opens (closes) a positive run appends its packed position to `starts`
// (`ends`, exclusive) through an atomic counter.  Padding labels are 0, so runs never cross reads;
// after sorting both lists ascending on the host the k-th start pairs with the k-th end.
// ------------------------------------------------------------------------------------------
#define CF_SPANS_PER_THREAD 16
__global__ __launch_bounds__(256) void spans_kernel(const uint8_t* __restrict__ labels, int64_t total, int64_t max_runs,
                                                    int64_t* __restrict__ starts, int64_t* __restrict__ ends,
                                                    unsigned long long* __restrict__ counts) {
    // A workgroup scans 256 * CF_SPANS_PER_THREAD samples, reserves room for all its boundaries with ONE global atomic per
    // list (slots inside the reservation come from LDS counters), then writes them.
    __shared__ unsigned n_loc[2];
    __shared__ unsigned long long base[2];
    if (threadIdx.x < 2) n_loc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * 256 * CF_SPANS_PER_THREAD + threadIdx.x;
    unsigned ms = 0, me = 0;                               // bit j: sample i0 + 256 j starts / ends a run
#pragma unroll
    for (int j = 0; j < CF_SPANS_PER_THREAD; ++j) {
        const int64_t i = i0 + (int64_t)j * 256;
        if (i < total && labels[i]) {
            if (i == 0 || !labels[i - 1]) ms |= 1u << j;
            if (i == total - 1 || !labels[i + 1]) me |= 1u << j;
        }
    }
    unsigned os = 0, oe = 0;
    if (ms) os = atomicAdd(&n_loc[0], (unsigned)__popc(ms));
    if (me) oe = atomicAdd(&n_loc[1], (unsigned)__popc(me));
    __syncthreads();
    if (threadIdx.x < 2 && n_loc[threadIdx.x]) base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)n_loc[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CF_SPANS_PER_THREAD; ++j) {
        const int64_t i = i0 + (int64_t)j * 256;
        if (ms & (1u << j)) {
            const unsigned long long k = base[0] + os++;
            if ((int64_t)k < max_runs) starts[k] = i;
        }
        if (me & (1u << j)) {
            const unsigned long long k = base[1] + oe++;
            if ((int64_t)k < max_runs) ends[k] = i + 1;
        }
    }
}

// Kernel 4c: a read packed without padding (read_lengths[r] == read_offsets[r + 1] - read_offsets[r]) touches the next one, and
// spans_kernel, which sees labels only, reports a positive run across the boundary as one.  One thread per read boundary splits it.
__global__ __launch_bounds__(256) void spans_cut_kernel(const uint8_t* __restrict__ labels, const int64_t* __restrict__ read_offsets,
                                                        const int64_t* __restrict__ read_lengths, int64_t n_reads, int64_t total, int64_t max_runs,
                                                        int64_t* __restrict__ starts, int64_t* __restrict__ ends,
                                                        unsigned long long* __restrict__ counts) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x + 1;         // boundary between read r - 1 and read r
    if (r >= n_reads) return;
    const int64_t at = read_offsets[r];
    if (at <= 0 || at >= total || read_offsets[r - 1] + read_lengths[r - 1] != at || read_lengths[r] <= 0) return;
    if (!labels[at - 1] || !labels[at]) return;
    const unsigned long long ks = atomicAdd(&counts[0], 1ull), ke = atomicAdd(&counts[1], 1ull);
    if ((int64_t)ks < max_runs) starts[ks] = at;
    if ((int64_t)ke < max_runs) ends[ke] = at;
}

// ------------------------------------------------------------------------------------------
// Kernel 5: signal ingest -- per-read median / MAD normalisation of raw int16 DAC samples
// (normalize_raw_signal, infer.py:96-105) fused with the zero padding + window packing of
// infer.py:31-43.  One workgroup per read.  Medians are found exactly by radix selection over
// LDS histograms (int16 keys: 8 + 8 bits; doubled absolute deviations, 17 bits: 9 + 8 bits); for
// even lengths the two middle order statistics are averaged like numpy.median.  The division is
// done in double and rounded once to fp32, i.e. bit-identical to numpy's float64 result cast to
// float32 (what TF's feed does, rnn_class.py:214).
// ------------------------------------------------------------------------------------------
template <typename KeyFn>
__device__ int radix_select(const int16_t* __restrict__ v, int64_t n, int64_t rank, int hi_bits, KeyFn key, unsigned* hist,
                            unsigned* sh) {
    // level 1: histogram of the high bits
    const int nb1 = 1 << hi_bits;
    for (int i = threadIdx.x; i < nb1; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[key(v[i]) >> 8], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        int b = 0;
        for (; b < nb1; ++b) {
            if (acc + hist[b] > rank) break;
            acc += hist[b];
        }
        sh[0] = (unsigned)b;
        sh[1] = (unsigned)(rank - acc);     // rank inside the bin
    }
    __syncthreads();
    const unsigned bin = sh[0];
    const unsigned r2 = sh[1];
    __syncthreads();
    // level 2: histogram of the low 8 bits inside that bin
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned k = key(v[i]);
        if ((k >> 8) == bin) atomicAdd(&hist[k & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned acc = 0;
        int b = 0;
        for (; b < 256; ++b) {
            if (acc + hist[b] > r2) break;
            acc += hist[b];
        }
        sh[2] = (bin << 8) | (unsigned)b;
    }
    __syncthreads();
    const int out = (int)sh[2];
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(256) void normalize_kernel(const int16_t* __restrict__ dac, const int64_t* __restrict__ dac_offsets,
                                                        const int64_t* __restrict__ win_offsets, float* __restrict__ x_out) {
    __shared__ unsigned hist[512];
    __shared__ unsigned sh[4];
    const int64_t r = blockIdx.x;
    const int16_t* v = dac + dac_offsets[r];
    const int64_t n = dac_offsets[r + 1] - dac_offsets[r];
    float* out = x_out + win_offsets[r] * CF_T;
    const int64_t n_pad = (win_offsets[r + 1] - win_offsets[r]) * CF_T;
    if (n <= 0) {
        for (int64_t i = threadIdx.x; i < n_pad; i += blockDim.x) out[i] = 0.f;
        return;
    }
    auto key16 = [](int16_t x) -> unsigned { return (unsigned)((int)x + 32768); };
    const int lo = radix_select(v, n, (n - 1) / 2, 8, key16, hist, sh);
    const int hi = (n & 1) ? lo : radix_select(v, n, n / 2, 8, key16, hist, sh);
    const int med2 = (lo - 32768) + (hi - 32768);            // 2 * median, exact
    auto keydev = [med2](int16_t x) -> unsigned { const int d = 2 * (int)x - med2; return (unsigned)(d < 0 ? -d : d); };
    const int dlo = radix_select(v, n, (n - 1) / 2, 9, keydev, hist, sh);
    const int dhi = (n & 1) ? dlo : radix_select(v, n, n / 2, 9, keydev, hist, sh);
    const double shift = 0.5 * (double)med2;
    const double scale = 0.25 * (double)(dlo + dhi);          // median(|raw - shift|)
    for (int64_t i = threadIdx.x; i < n_pad; i += blockDim.x)
        out[i] = i < n ? (float)(((double)v[i] - shift) / scale) : 0.f;
}

#include "ingest_post.hpp"      // round 5: normalize_regs_kernel, postprocess_bits_kernel (same results, see there)

// ==========================================================================================
// Host side
// ==========================================================================================
#include "generic.hpp"

// A/B and test knobs: environment variables (CATFISH_GENERIC, CATFISH_WAVES, CATFISH_BF16_PIPE, ... -- tools/README.md lists
// them) that change WHICH kernels a call uses.  They are for tools/ and tests and are honoured only while
// CATFISH_DEBUG_KNOBS=1 is set as well; every knob that takes effect is named once on stderr, so a stray variable can
// neither change results nor do so silently.
static const char* cf_knob(const char* name) {
    const char* on = getenv("CATFISH_DEBUG_KNOBS");
    if (!on || atoi(on) == 0) return nullptr;
    const char* v = getenv(name);
    if (v) {
        static std::mutex mu;
        static std::set<std::string> seen;
        std::lock_guard<std::mutex> lock(mu);
        if (seen.insert(std::string(name) + "=" + v).second) fprintf(stderr, "catfish_hip: debug knob %s=%s is active\n", name, v);
    }
    return v;
}

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(CF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

enum { SLOT_RES_FIRST = 0, SLOT_RES, SLOT_GRU0, SLOT_GRU, SLOT_GRU_LAST, SLOT_HEAD, SLOT_POST, SLOT_NORM, SLOT_GRU_FUSED, SLOT_RES_STACK2 };
static const char* k_slot_names[CF_PROF_SLOTS] = {"res_block_first", "res_block",      "gru_layer_first", "gru_layer_mid",
                                                  "gru_layer_last",  "head",           "postprocess",     "normalize",       "gru_fused",
                                                  "res_stack2",      "unused",         "unused"};

struct cf_model {
    cf_hparams hp;
    int device = 0;
    int n_cu = 256;
    std::vector<float*> d_res;   // per residual block packed weights
    std::vector<float*> d_gru;   // per layer packed weights [2 dirs]
    std::vector<int> gru_cin;
    std::vector<char*> d_res_b;  // per residual block packed bf16 weights (precision != fp32)
    std::vector<char*> d_gru_b;  // per layer packed bf16 weights [2 dirs] (precision != fp32)
    int np = 0;                  // bf16 parts per operand: 0 = fp32 path, 1 = bf16, 2 = bf16x3
    float dense_bias = 0.f;
    // workspace
    int64_t cap_windows = 0;              // per workspace slot
    int64_t cap_tiles = 0;
    // A pass needs ~2.6 KB of scratch per sample; every slot owns one scratch set and one
    // internal stream, so consecutive sub-batches overlap (the tail of one layer's grid leaves
    // CUs idle that the other slot's kernels fill).
    struct Slot {
        float* d_a[2] = {nullptr, nullptr};   // conv ping-pong, F = 32
        float* d_y[2] = {nullptr, nullptr};   // GRU layer outputs ping-pong, F = 128
        float* d_p = nullptr;                 // dense partials [2][tiles][35][16]
        unsigned* d_flags = nullptr;          // fused launch: [n_layers][groups][2] completion flags
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        int64_t last_windows = 0;             // windows of the last pass (debug hook)
        bool last_res_fused = false;          // the last pass ran blocks 0 and 1 as one launch: block 0's output was never stored
    };
    std::vector<Slot> slots;
    cf_generic* gen = nullptr;                // any-size path (generic.hpp): set when the geometry is not the shipped 64 / 32
    int fuse = 0;                             // all GRU layers in one launch (fp32 path, n_layers <= 3): 0 never, 1 always,
                                              // 2 auto = only for passes of >= 6 rounds, where the dynamic queues pay
    float* d_xp = nullptr;                    // hoisted x projection of small calls: [xp_tiles][35][2][12][64] f32x4
    int xp_tiles = 0;
    float* d_host_x = nullptr;                // cf_infer_host staging (grown on demand)
    float* d_host_p = nullptr;
    size_t host_stage_bytes = 0;
    unsigned* h_err = nullptr;                // host-mapped: set by a fused launch whose bounded wait timed out (sticky)
    unsigned* d_err = nullptr;                // device view of h_err
    hipEvent_t fork = nullptr;
    int64_t ws_bytes = 0;
    // profiling
    bool prof = false;                    // events around the kernels of THIS call (set per cf_infer from prof_every)
    int prof_every = 0;                   // 0 = off, N = time every N-th cf_infer call
    int64_t prof_calls = 0;
    struct Ev { hipEvent_t a, b; int slot; };
    std::vector<Ev> ev_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[CF_PROF_SLOTS] = {0};
    int64_t prof_n[CF_PROF_SLOTS] = {0};
};

// ---- weight packing ----------------------------------------------------------------------
static void pack_vec(float* dst, const std::vector<double>& v) {  // [mo][q][r] order == natural order of 16mo+4q+r
    for (size_t i = 0; i < v.size(); ++i) dst[i] = (float)v[i];
}

// 32x32 unit: dst[(ks*64 + lane)*2 + mo] = W[in = frag_feature(ks, lane>>4)][out = 16mo + (lane&15)]
static void pack_unit(float* dst, const std::vector<double>& w /*[32 in][32 out]*/) {
    for (int ks = 0; ks < 8; ++ks)
        for (int lane = 0; lane < 64; ++lane)
            for (int mo = 0; mo < 2; ++mo)
                dst[(ks * 64 + lane) * 2 + mo] = (float)w[frag_feature(ks, lane >> 4) * 32 + 16 * mo + (lane & 15)];
}

struct FoldedConv {
    int k = 0, cin = 0;
    std::vector<double> w;  // [k][cin][32] scaled by BN
    std::vector<double> b;  // [32]
};

// y = BN(conv(x)) = conv'(x): W' = W*s, b' = b*s + beta - mean*s, s = gamma*rsqrt(var+eps)
static FoldedConv fold(const cf_conv_bn& c, float eps) {
    FoldedConv f;
    f.k = c.ksize; f.cin = c.cin;
    f.w.resize((size_t)c.ksize * c.cin * CF_C);
    f.b.resize(CF_C);
    for (int o = 0; o < CF_C; ++o) {
        // same arithmetic as the unfused graph: inv = rsqrt(var + eps) * gamma (fp32 inputs, double math)
        const double s = (double)c.gamma[o] / std::sqrt((double)c.moving_variance[o] + (double)eps);
        f.b[o] = (double)c.bias[o] * s + (double)c.beta[o] - (double)c.moving_mean[o] * s;
        for (int k = 0; k < c.ksize; ++k)
            for (int i = 0; i < c.cin; ++i)
                f.w[((size_t)k * c.cin + i) * CF_C + o] = (double)c.kernel[((size_t)k * c.cin + i) * CF_C + o] * s;
    }
    return f;
}

static std::vector<double> tap(const FoldedConv& f, int k) {
    return std::vector<double>(f.w.begin() + (size_t)k * f.cin * CF_C, f.w.begin() + (size_t)(k + 1) * f.cin * CF_C);
}

static int pack_res_block(const cf_conv_bn* c4, bool first, float eps, std::vector<float>& out) {
    out.assign(res_pack_floats(first), 0.f);
    const FoldedConv sc = fold(c4[0], eps), f1 = fold(c4[1], eps), f3 = fold(c4[2], eps), fl = fold(c4[3], eps);
    const int cin = first ? 1 : CF_C;
    if (sc.k != 1 || f1.k != 1 || f3.k != 3 || fl.k != 1 || sc.cin != cin || f1.cin != cin || f3.cin != CF_C || fl.cin != CF_C)
        return fail(CF_ERR_INVALID, "residual block geometry not supported (need k = 1,1,3,1 and 32 channels)");
    float* vecs = out.data() + res_units(first) * 1024;
    if (first) {
        for (int k = 0; k < 3; ++k) pack_unit(out.data() + k * 1024, tap(f3, k));
        pack_unit(out.data() + 3 * 1024, tap(fl, 0));
        pack_vec(vecs + 0 * 32, f3.b); pack_vec(vecs + 1 * 32, fl.b);
        pack_vec(vecs + 2 * 32, sc.w); pack_vec(vecs + 3 * 32, sc.b);
        pack_vec(vecs + 4 * 32, f1.w); pack_vec(vecs + 5 * 32, f1.b);
    } else {
        pack_unit(out.data() + 0 * 1024, tap(sc, 0));
        pack_unit(out.data() + 1 * 1024, tap(f1, 0));
        for (int k = 0; k < 3; ++k) pack_unit(out.data() + (2 + k) * 1024, tap(f3, k));
        pack_unit(out.data() + 5 * 1024, tap(fl, 0));
        pack_vec(vecs + 0 * 32, sc.b); pack_vec(vecs + 1 * 32, f1.b);
        pack_vec(vecs + 2 * 32, f3.b); pack_vec(vecs + 3 * 32, fl.b);
    }
    return CF_OK;
}

static int pack_res_block_bf16(const cf_conv_bn* c4, bool first, float eps, int np, std::vector<char>& out) {
    out.assign(rb_pack_bytes(first, np), 0);
    const FoldedConv sc = fold(c4[0], eps), f1 = fold(c4[1], eps), f3 = fold(c4[2], eps), fl = fold(c4[3], eps);
    const int cin = first ? 1 : CF_C;
    if (sc.k != 1 || f1.k != 1 || f3.k != 3 || fl.k != 1 || sc.cin != cin || f1.cin != cin || f3.cin != CF_C || fl.cin != CF_C)
        return fail(CF_ERR_INVALID, "residual block geometry not supported (need k = 1,1,3,1 and 32 channels)");
    const size_t ub = (size_t)2 * np * 1024;     // bytes per unit
    char* vecs = out.data() + rb_vec_off(first, np);
    if (first) {
        for (int k = 0; k < 3; ++k) pack_unit_bf16(out.data() + k * ub, tap(f3, k), np);
        pack_unit_bf16(out.data() + 3 * ub, tap(fl, 0), np);
        pack_vec32(vecs + 0 * 128, f3.b); pack_vec32(vecs + 1 * 128, fl.b);
        pack_vec32(vecs + 2 * 128, sc.w); pack_vec32(vecs + 3 * 128, sc.b);
        pack_vec32(vecs + 4 * 128, f1.w); pack_vec32(vecs + 5 * 128, f1.b);
    } else {
        pack_unit_bf16(out.data() + 0 * ub, tap(sc, 0), np);
        pack_unit_bf16(out.data() + 1 * ub, tap(f1, 0), np);
        for (int k = 0; k < 3; ++k) pack_unit_bf16(out.data() + (2 + k) * ub, tap(f3, k), np);
        pack_unit_bf16(out.data() + 5 * ub, tap(fl, 0), np);
        pack_vec32(vecs + 0 * 128, sc.b); pack_vec32(vecs + 1 * 128, f1.b);
        pack_vec32(vecs + 2 * 128, f3.b); pack_vec32(vecs + 3 * 128, fl.b);
    }
    return CF_OK;
}

static void pack_gru_dir(const cf_gru_dir& g, int cin, int cin_real, const float* dense_w /*64 floats of this direction or null*/,
                         float* out) {
    // cin = padded input width of the kernel instantiation, cin_real = rows of the x part in the checkpoint
    auto wfull = [&](int in, int o) -> float {   // pre-scaled: the accumulators are exp2 arguments
        if (in < 0) return 0.f;
        return o < 2 * CF_H ? (float)(CF_GATE_SCALE * (double)g.gates_kernel[(size_t)in * 2 * CF_H + o])
                            : (float)(CF_CAND_SCALE * (double)g.candidate_kernel[(size_t)in * CF_H + (o - 2 * CF_H)]);
    };
    float* px = out;
    for (int ks = 0; ks < cin / 4; ++ks)
        for (int gq = 0; gq < 3; ++gq)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    px[((ks * 3 + gq) * 64 + lane) * 4 + j] =
                        wfull(frag_feature(ks, lane >> 4) < cin_real ? frag_feature(ks, lane >> 4) : -1, 16 * (4 * gq + j) + (lane & 15));
    float* pg = out + gru_x_floats(cin);
    for (int ks = 0; ks < 16; ++ks)
        for (int gq = 0; gq < 2; ++gq)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    pg[((ks * 2 + gq) * 64 + lane) * 4 + j] = wfull(cin_real + frag_feature(ks, lane >> 4), 16 * (4 * gq + j) + (lane & 15));
    float* pc = pg + gru_hg_floats();
    for (int ks = 0; ks < 16; ++ks)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j)
                pc[(ks * 64 + lane) * 4 + j] = wfull(cin_real + frag_feature(ks, lane >> 4), 2 * CF_H + 16 * j + (lane & 15));
    float* pb = out + gru_bias_off(cin);
    for (int i = 0; i < 2 * CF_H; ++i) pb[i] = (float)(CF_GATE_SCALE * (double)g.gates_bias[i]);
    for (int i = 0; i < CF_H; ++i) pb[2 * CF_H + i] = (float)(CF_CAND_SCALE * (double)g.candidate_bias[i]);
    float* pd = out + gru_dense_off(cin);
    for (int i = 0; i < CF_H; ++i) pd[i] = dense_w ? dense_w[i] : 0.f;
}

// ---- model lifetime ----------------------------------------------------------------------
static int upload(const std::vector<float>& host, float** dev) {
    HIP_TRY(hipMalloc((void**)dev, host.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return CF_OK;
}

static bool gen_wanted(const cf_hparams* hp);
static int gen_build(cf_model* m, const cf_weights* w);
static void gen_destroy(cf_generic* g);

extern "C" void cf_model_destroy(cf_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (float* p : m->d_res) if (p) (void)hipFree(p);
    for (float* p : m->d_gru) if (p) (void)hipFree(p);
    for (char* p : m->d_gru_b) if (p) (void)hipFree(p);
    for (char* p : m->d_res_b) if (p) (void)hipFree(p);
    for (auto& sl : m->slots) {
        for (int i = 0; i < 2; ++i) { if (sl.d_a[i]) (void)hipFree(sl.d_a[i]); if (sl.d_y[i]) (void)hipFree(sl.d_y[i]); }
        if (sl.d_p) (void)hipFree(sl.d_p);
        if (sl.d_flags) (void)hipFree(sl.d_flags);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    gen_destroy(m->gen);
    if (m->d_xp) (void)hipFree(m->d_xp);
    if (m->d_host_x) (void)hipFree(m->d_host_x);
    if (m->h_err) (void)hipHostFree(m->h_err);
    if (m->fork) (void)hipEventDestroy(m->fork);
    for (auto& e : m->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : m->ev_pool) (void)hipEventDestroy(e);
    delete m;
}

extern "C" int cf_model_create(const cf_weights* w, const cf_hparams* hp, int device, cf_model** out) {
    if (!w || !hp || !out) return fail(CF_ERR_INVALID, "cf_model_create: null argument");
    *out = nullptr;
    if (hp->window != CF_T) return fail(CF_ERR_INVALID, "window must be 35 (rnn_class.py:27)");
    if (hp->n_layers < 1 || hp->n_layers_res < 0) return fail(CF_ERR_INVALID, "n_layers must be >= 1 and n_layers_res >= 0");
    // the shipped geometry (64 GRU units, 32 conv channels) runs on the tuned kernels, every other one on generic.hpp
    const bool generic = gen_wanted(hp);
    if (!generic && hp->n_layers_res == 0 && hp->precision != CF_PREC_FP32)
        return fail(CF_ERR_INVALID, "the plain RNN type (n_layers_res = 0) is only built for CF_PREC_FP32");
    if ((hp->n_layers_res > 0 && !w->conv) || !w->gru || !w->dense_kernel || !w->dense_bias)
        return fail(CF_ERR_INVALID, "cf_weights has null members");
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(CF_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    if (hp->precision < CF_PREC_FP32 || hp->precision > CF_PREC_BF16)
        return fail(CF_ERR_INVALID, "precision must be CF_PREC_FP32, CF_PREC_BF16X3 or CF_PREC_BF16");
    cf_model* m = new cf_model();
    m->hp = *hp;
    m->device = device;
    m->np = hp->precision == CF_PREC_FP32 ? 0 : (hp->precision == CF_PREC_BF16X3 ? 2 : 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
    if (generic) {
        m->dense_bias = w->dense_bias[0];
        const int grc = gen_build(m, w);
        if (grc != CF_OK) { cf_model_destroy(m); return grc; }
        *out = m;
        return CF_OK;
    }
    int rc = CF_OK;
    // residual blocks
    for (int b = 0; b < hp->n_layers_res && rc == CF_OK; ++b) {
        std::vector<float> blob;
        rc = pack_res_block(w->conv + 4 * b, b == 0, hp->bn_epsilon, blob);
        if (rc == CF_OK) { float* d = nullptr; rc = upload(blob, &d); m->d_res.push_back(d); }
        if (rc == CF_OK && m->np > 0) {
            std::vector<char> bblob;
            rc = pack_res_block_bf16(w->conv + 4 * b, b == 0, hp->bn_epsilon, m->np, bblob);
            if (rc == CF_OK) {
                char* bptr = nullptr;
                hipError_t e = hipMalloc((void**)&bptr, bblob.size());
                if (e == hipSuccess) e = hipMemcpy(bptr, bblob.data(), bblob.size(), hipMemcpyHostToDevice);
                if (e != hipSuccess) rc = fail(CF_ERR_HIP, std::string("bf16 conv weight upload: ") + hipGetErrorString(e));
                m->d_res_b.push_back(bptr);
            }
        }
    }
    // GRU layers
    for (int l = 0; l < hp->n_layers && rc == CF_OK; ++l) {
        const int cin_real = l == 0 ? (hp->n_layers_res > 0 ? CF_C : 1) : 2 * CF_H;
        const int cin = cin_real == 1 ? 16 : cin_real;     // RNN type: the raw sample embedded in 16 features
        if (w->gru[2 * l].cin != cin_real || w->gru[2 * l + 1].cin != cin_real) { rc = fail(CF_ERR_INVALID, "GRU layer input width mismatch"); break; }
        const bool last = l == hp->n_layers - 1;
        std::vector<float> blob((size_t)2 * gru_pack_floats(cin));
        for (int d = 0; d < 2; ++d)
            pack_gru_dir(w->gru[2 * l + d], cin, cin_real, last ? w->dense_kernel + d * CF_H : nullptr,
                         blob.data() + (size_t)d * gru_pack_floats(cin));
        float* dptr = nullptr;
        rc = upload(blob, &dptr);
        m->d_gru.push_back(dptr);
        m->gru_cin.push_back(cin);
        if (rc == CF_OK && m->np > 0) {
            const size_t bytes = (size_t)gb_pack_bytes(cin, m->np);
            std::vector<char> bblob(2 * bytes, 0);
            for (int d = 0; d < 2; ++d)
                pack_gru_dir_bf16(w->gru[2 * l + d], cin, m->np, last ? w->dense_kernel + d * CF_H : nullptr, bblob.data() + d * bytes);
            char* bptr = nullptr;
            hipError_t e = hipMalloc((void**)&bptr, bblob.size());
            if (e == hipSuccess) e = hipMemcpy(bptr, bblob.data(), bblob.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fail(CF_ERR_HIP, std::string("bf16 weight upload: ") + hipGetErrorString(e));
            m->d_gru_b.push_back(bptr);
        }
    }
    m->dense_bias = w->dense_bias[0];
    // workspace
    if (rc == CF_OK) {
        int64_t cap = hp->max_windows_per_pass > 0 ? hp->max_windows_per_pass : 32768;
        cap = (cap + 2 * CF_TILE - 1) / (2 * CF_TILE) * (2 * CF_TILE);   // whole 32-window tiles (bf16 path)
        m->cap_windows = cap;
        m->cap_tiles = cap / CF_TILE;
        const size_t a_bytes = (size_t)m->cap_tiles * CF_T * 2 * 64 * sizeof(f32x4);
        const size_t y_bytes = (size_t)m->cap_tiles * CF_T * 8 * 64 * sizeof(f32x4);
        // dense partials: [2][tiles][35][16] from the throughput kernels, [2][tiles <= CUs][35][4][64] per-lane partials from the
        // latency-mode kernels
        const size_t p_bytes = std::max((size_t)2 * m->cap_tiles * CF_T * 16, (size_t)2 * std::min<int64_t>(m->cap_tiles, m->n_cu) * CF_T * 256) * sizeof(float);
        hipError_t e = hipSuccess;
        int n_slots = hp->n_streams > 0 ? hp->n_streams : 1;   // measured: splitting one call over 2 internal streams is slower
        if (n_slots > 8) n_slots = 8;
        m->slots.resize(n_slots);
        for (auto& sl : m->slots) {
            for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&sl.d_a[i], a_bytes);
            for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void**)&sl.d_y[i], y_bytes);
            if (e == hipSuccess) e = hipMalloc((void**)&sl.d_p, p_bytes);
            if (e == hipSuccess) e = hipMalloc((void**)&sl.d_flags, (size_t)3 * ((m->cap_tiles + 7) / 8) * 2 * sizeof(unsigned) + 64);
            if (e == hipSuccess && n_slots > 1) e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
            if (e == hipSuccess && n_slots > 1) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        }
        if (e == hipSuccess && n_slots > 1) e = hipEventCreateWithFlags(&m->fork, hipEventDisableTiming);
        if (e == hipSuccess && n_slots == 1 && m->np == 0) {
            // small calls: room for one layer's hoisted x projection (CATFISH_HOIST_TILES: A/B knob for tools/)
            // measured crossover (tools/bench_latency.py with CATFISH_HOIST_TILES): hoisting pays up to ~48 tiles = 768 windows
            int xpt = (int)std::min<int64_t>(m->cap_tiles, std::max(1, 3 * m->n_cu / 16));
            if (cf_knob("CATFISH_HOIST_TILES")) xpt = std::max(1, std::min(atoi(cf_knob("CATFISH_HOIST_TILES")), (int)m->cap_tiles));
            e = hipMalloc((void**)&m->d_xp, (size_t)xpt * CF_T * 2 * 12 * 64 * sizeof(f32x4));
            if (e == hipSuccess) m->xp_tiles = xpt;
        }
        if (e == hipSuccess) e = hipHostMalloc((void**)&m->h_err, sizeof(unsigned), hipHostMallocMapped);
        if (e == hipSuccess) { *m->h_err = 0u; e = hipHostGetDevicePointer((void**)&m->d_err, m->h_err, 0); }
        m->fuse = (m->np == 0 && hp->n_layers <= 3 && hp->fuse_layers >= 0) ? (hp->fuse_layers > 0 ? 1 : 2) : 0;
        if (cf_knob("CATFISH_FUSE") && m->fuse) m->fuse = atoi(cf_knob("CATFISH_FUSE")) != 0 ? 1 : 0;   // A/B knob for tools/
        if (e != hipSuccess) rc = fail(CF_ERR_NOMEM, std::string("workspace allocation: ") + hipGetErrorString(e));
        m->ws_bytes = (int64_t)n_slots * (int64_t)(2 * a_bytes + 2 * y_bytes + p_bytes);
    }
    if (rc == CF_OK) {
        // opt in to > 64 KiB dynamic LDS for every GRU instantiation we may launch
        hipError_t e = hipSuccess;
        auto optin = [&](const void* f, int bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
        optin((const void*)gru_layer_kernel<16, false>, gru_pack_floats(16) * 4);
        optin((const void*)gru_layer_kernel<16, true>, gru_pack_floats(16) * 4);
        optin((const void*)gru_layer_kernel<32, false>, gru_pack_floats(32) * 4);
        optin((const void*)gru_layer_kernel<32, true>, gru_pack_floats(32) * 4);
        optin((const void*)gru_layer_kernel<128, false>, gru_pack_floats(128) * 4);
        optin((const void*)gru_layer_kernel<128, true>, gru_pack_floats(128) * 4);
        optin((const void*)gru_fused_kernel, gru_pack_floats(128) * 4 + 16);
        optin((const void*)gru_layer_coop_kernel<16, false>, (gru_pack_floats(16) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_layer_coop_kernel<16, true>, (gru_pack_floats(16) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_layer_coop_kernel<32, false>, (gru_pack_floats(32) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_layer_coop_kernel<32, true>, (gru_pack_floats(32) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_layer_coop_kernel<128, false>, (gru_pack_floats(128) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_layer_coop_kernel<128, true>, (gru_pack_floats(128) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_train_fwd_coop_kernel<32>, (gru_pack_floats(32) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_train_fwd_coop_kernel<128>, (gru_pack_floats(128) + CF_COOP_XCH_FLOATS) * 4);
        optin((const void*)gru_train_bwd_coop_kernel<32>, (gtb_pack_floats(32) + CF_COOP_BWD_XCH_FLOATS) * 4);
        optin((const void*)gru_train_bwd_coop_kernel<128>, (gtb_pack_floats(128) + CF_COOP_BWD_XCH_FLOATS) * 4);
        optin((const void*)gru_train_fwd_kernel<32>, gru_pack_floats(32) * 4);
        optin((const void*)gru_train_fwd_kernel<128>, gru_pack_floats(128) * 4);
        optin((const void*)gru_train_bwd_kernel<32>, gtb_pack_floats(32) * 4);
        optin((const void*)gru_train_bwd_kernel<128>, gtb_pack_floats(128) * 4);
        optin((const void*)res_stack2_kernel<true>, (res_pack_floats(true) + res_pack_floats(false) + 4 * CF_TILE * CF_T) * 4);
        optin((const void*)res_stack2_kernel<false>, (res_pack_floats(true) + res_pack_floats(false) + 4 * CF_TILE * CF_T) * 4);
        optin((const void*)res_train_fwd_kernel, RT_FWD_LDS_BYTES);
        optin((const void*)res_train_bwd_kernel, RT_BWD_LDS_BYTES);
        optin((const void*)gru_layer_bf16_kernel<32, false, 1>, gb_pack_bytes(32, 1));
        optin((const void*)gru_layer_bf16_kernel<32, true, 1>, gb_pack_bytes(32, 1));
        optin((const void*)gru_layer_bf16_kernel<128, false, 1>, gb_pack_bytes(128, 1));
        optin((const void*)gru_layer_bf16_kernel<128, true, 1>, gb_pack_bytes(128, 1));
        optin((const void*)gru_bf16_pipe_kernel<32, false>, gb_pack_bytes(32, 1));
        optin((const void*)gru_bf16_pipe_kernel<32, true>, gb_pack_bytes(32, 1));
        optin((const void*)gru_bf16_pipe_kernel<128, false>, gb_pack_bytes(128, 1));
        optin((const void*)gru_bf16_pipe_kernel<128, true>, gb_pack_bytes(128, 1));
        optin((const void*)gru_xproj_lds_kernel<32>, gru_x_floats(32) * 4);
        optin((const void*)gru_xproj_lds_kernel<128>, gru_x_floats(128) * 4);
        optin((const void*)gru_bf16x3_pipe_kernel<32, false>, gb_pack_bytes(32, 2));
        optin((const void*)gru_bf16x3_pipe_kernel<32, true>, gb_pack_bytes(32, 2));
        optin((const void*)gru_bf16x3_pipe_kernel<128, false>, gb_pack_bytes(128, 2));
        optin((const void*)gru_bf16x3_pipe_kernel<128, true>, gb_pack_bytes(128, 2));
        optin((const void*)gru_layer_bf16_kernel<32, false, 2>, gb_pack_bytes(32, 2));
        optin((const void*)gru_layer_bf16_kernel<32, true, 2>, gb_pack_bytes(32, 2));
        optin((const void*)gru_layer_bf16_kernel<128, false, 2>, gb_pack_bytes(128, 2));
        optin((const void*)gru_layer_bf16_kernel<128, true, 2>, gb_pack_bytes(128, 2));
        if (e != hipSuccess) rc = fail(CF_ERR_HIP, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
    }
    if (rc != CF_OK) { std::string keep = g_err; cf_model_destroy(m); g_err = keep; return rc; }
    *out = m;
    return CF_OK;
}

// ---- launch helpers ----------------------------------------------------------------------
static int prof_begin(cf_model* m, int slot, hipStream_t s, size_t* idx) {
    if (!m->prof) return CF_OK;
    cf_model::Ev ev;
    ev.slot = slot;
    for (hipEvent_t* p : {&ev.a, &ev.b}) {
        if (!m->ev_pool.empty()) { *p = m->ev_pool.back(); m->ev_pool.pop_back(); }
        else HIP_TRY(hipEventCreate(p));
    }
    HIP_TRY(hipEventRecord(ev.a, s));
    m->ev_pending.push_back(ev);
    *idx = m->ev_pending.size() - 1;
    return CF_OK;
}
static int prof_end(cf_model* m, hipStream_t s, size_t idx) {
    if (!m->prof) return CF_OK;
    HIP_TRY(hipEventRecord(m->ev_pending[idx].b, s));
    return CF_OK;
}

// Waves per workgroup: 8 (two per SIMD) when the pass fills the chip; fewer for small calls (a single read is
// 8 tiles x 2 directions) so that the tiles spread over more CUs instead of sharing SIMDs -- the 35-step
// chain is latency-bound there.
static int pick_waves(int n_tile_tasks, int n_cu) {
    int w = (n_tile_tasks + n_cu - 1) / n_cu;
    return w <= 1 ? 1 : (w <= 2 ? 2 : (w <= 4 ? 4 : 8));
}

// latency mode: up to two rounds of one (tile, direction) per CU (0.35 units each) beat one wave per tile (1 unit)
static bool use_coop(const cf_model* m, int n_tiles) {
    static const int coop_env = cf_knob("CATFISH_COOP") ? atoi(cf_knob("CATFISH_COOP")) : -1;      // A/B knob for tools/
    const bool coop = coop_env >= 0 ? coop_env != 0 : n_tiles <= m->n_cu;
    return coop && !(CF_ABLATE & 4) && n_tiles <= m->n_cu;      // (the raw dense-partial buffer is sized for n_cu tiles)
}

// x projection of a small call on the CUs its recurrence leaves idle (gru_coop.hpp): weights through LDS, chunks sized to fill
// the chip once; CATFISH_XPROJ_LDS=0 behind the debug switch selects round 2's kernel (fragments straight from L2), for A/B
template <int CIN>
static void launch_xproj(cf_model* m, const float* wpack, const float* X, int n_tiles, hipStream_t s) {
    const int lds_env = cf_knob("CATFISH_XPROJ_LDS") ? atoi(cf_knob("CATFISH_XPROJ_LDS")) : 1;
    if (lds_env) {
        int chunks = cf_xproj_plan(n_tiles, m->n_cu, CIN);
        if (cf_knob("CATFISH_XPROJ_CHUNKS")) {                                   // A/B knob for tools/: another valid chunk count
            const int c = std::max(1, std::min(CF_T, atoi(cf_knob("CATFISH_XPROJ_CHUNKS"))));
            const int tl = (CF_T + c - 1) / c;
            chunks = (CF_T + tl - 1) / tl;
        }
        hipLaunchKernelGGL((gru_xproj_lds_kernel<CIN>), dim3(n_tiles * chunks, 2), dim3(256), gru_x_floats(CIN) * 4, s, wpack,
                           reinterpret_cast<const f32x4*>(X), reinterpret_cast<f32x4*>(m->d_xp), n_tiles, chunks);
    } else {
        hipLaunchKernelGGL((gru_xproj_kernel<CIN>), dim3(n_tiles * cf_xproj_chunks(n_tiles), 2), dim3(256), 0, s, wpack,
                           reinterpret_cast<const f32x4*>(X), reinterpret_cast<f32x4*>(m->d_xp), n_tiles, cf_xproj_chunks(n_tiles));
    }
}

template <int CIN, bool LAST>
static int launch_gru(cf_model* m, const float* wpack, const float* X, float* Y, float* P, int n_tiles, hipStream_t s, int slot) {
    if (use_coop(m, n_tiles)) {
        size_t pi = 0;
        int rc = prof_begin(m, slot, s, &pi);
        if (rc != CF_OK) return rc;
        const int gx = std::min(n_tiles, std::max(1, m->n_cu / 2));
        const f32x4* xp = nullptr;
        if (CIN >= 32 && n_tiles <= m->xp_tiles) {        // few tiles: the x projection runs on the idle CUs first
            xp = reinterpret_cast<const f32x4*>(m->d_xp);
            launch_xproj<CIN>(m, wpack, X, n_tiles, s);
        }
        hipLaunchKernelGGL((gru_layer_coop_kernel<CIN, LAST>), dim3(gx, 2), dim3(256), (gru_pack_floats(CIN) + CF_COOP_XCH_FLOATS) * 4, s,
                           wpack, reinterpret_cast<const f32x4*>(X), reinterpret_cast<f32x4*>(Y), P, n_tiles, xp);
        HIP_TRY(hipGetLastError());
        return prof_end(m, s, pi);
    }
    static const int waves_env = cf_knob("CATFISH_WAVES") ? atoi(cf_knob("CATFISH_WAVES")) : 0;     // A/B knob for tools/
    // (12 waves = 3 per SIMD measured +0.3 % on the Cin = 128 layers and costs the Cin = 32 layer its second workgroup per CU)
    const int waves = waves_env > 0 ? std::min(waves_env, 8) : ((CF_ABLATE & 4) ? 4 : pick_waves(2 * n_tiles, m->n_cu));
    const int groups = (n_tiles + waves - 1) / waves;             // one workgroup pass = one tile per wave
    int per_dir = m->n_cu / 2 > 0 ? m->n_cu / 2 : 1;            // persistent: half the CUs per direction
    constexpr int lds_bytes = gru_pack_floats(CIN) * 4;
    if (lds_bytes <= 80 * 1024) per_dir *= 2;                   // two workgroups fit per CU
    const int gx = groups < per_dir ? groups : per_dir;
    size_t pi = 0;
    int rc = prof_begin(m, slot, s, &pi);
    if (rc != CF_OK) return rc;
    hipLaunchKernelGGL((gru_layer_kernel<CIN, LAST>), dim3(gx, 2), dim3(waves * 64), lds_bytes, s, wpack,
                       reinterpret_cast<const f32x4*>(X), reinterpret_cast<f32x4*>(Y), P, n_tiles);
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}

template <int CIN, bool LAST, int NP>
static int launch_gru_bf16(cf_model* m, const char* wpack, const float* X, float* Y, float* P, int n_tiles32, hipStream_t s, int slot) {
    static const int waves_env = cf_knob("CATFISH_BF16_WAVES") ? atoi(cf_knob("CATFISH_BF16_WAVES")) : 0;     // A/B knobs for tools/
    static const int wgs_env = cf_knob("CATFISH_BF16_WGS") ? atoi(cf_knob("CATFISH_BF16_WGS")) : 0;
    const int waves = waves_env > 0 ? std::min(waves_env, 8) : pick_waves(2 * n_tiles32, m->n_cu);
    const int groups = (n_tiles32 + waves - 1) / waves;           // one workgroup pass = one 32-window tile per wave
    int per_dir = m->n_cu / 2 > 0 ? m->n_cu / 2 : 1;
    constexpr int lds_bytes = gb_pack_bytes(CIN, NP);
    if (lds_bytes <= 80 * 1024) per_dir *= 2;
    if (wgs_env > 0) per_dir = wgs_env;
    const int gx = groups < per_dir ? groups : per_dir;
    size_t pi = 0;
    int rc = prof_begin(m, slot, s, &pi);
    if (rc != CF_OK) return rc;
    const int pipe_env = cf_knob("CATFISH_BF16_PIPE") ? atoi(cf_knob("CATFISH_BF16_PIPE")) : 1;   // A/B knob for tools/ and tests, read per launch
    if constexpr (NP == 1) {
        if (pipe_env) {      // plain bf16: the software-pipelined kernel (vector work issued behind every MFMA)
            hipLaunchKernelGGL((gru_bf16_pipe_kernel<CIN, LAST>), dim3(gx, 2), dim3(waves * 64), lds_bytes, s, wpack,
                               reinterpret_cast<const bf16x8*>(X), reinterpret_cast<bf16x8*>(Y), P, n_tiles32);
            HIP_TRY(hipGetLastError());
            return prof_end(m, s, pi);
        }
    }
    if constexpr (NP == 2) {
        if (pipe_env) {      // bf16x3: the one-wave-per-SIMD pipelined kernel (512 registers, four waves per workgroup)
            const int w4 = std::min(waves, 4);
            const int g4 = (n_tiles32 + w4 - 1) / w4;
            const int per4 = wgs_env > 0 ? wgs_env : std::max(1, m->n_cu / 2);       // one workgroup per CU whatever its LDS
            hipLaunchKernelGGL((gru_bf16x3_pipe_kernel<CIN, LAST>), dim3(std::min(g4, per4), 2), dim3(w4 * 64), lds_bytes, s, wpack,
                               reinterpret_cast<const bf16x8*>(X), reinterpret_cast<bf16x8*>(Y), P, n_tiles32);
            HIP_TRY(hipGetLastError());
            return prof_end(m, s, pi);
        }
    }
    hipLaunchKernelGGL((gru_layer_bf16_kernel<CIN, LAST, NP>), dim3(gx, 2), dim3(waves * 64), lds_bytes, s, wpack,
                       reinterpret_cast<const bf16x8*>(X), reinterpret_cast<bf16x8*>(Y), P, n_tiles32);
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}

template <int NP>
static int launch_gru_bf16_layer(cf_model* m, int l, bool last, const float* cur, float* y, float* p, int n_tiles32, hipStream_t s) {
    const char* wp = m->d_gru_b[l];
    if (l == 0)
        return last ? launch_gru_bf16<32, true, NP>(m, wp, cur, y, p, n_tiles32, s, SLOT_GRU_LAST)
                    : launch_gru_bf16<32, false, NP>(m, wp, cur, y, p, n_tiles32, s, SLOT_GRU0);
    return last ? launch_gru_bf16<128, true, NP>(m, wp, cur, y, p, n_tiles32, s, SLOT_GRU_LAST)
                : launch_gru_bf16<128, false, NP>(m, wp, cur, y, p, n_tiles32, s, SLOT_GRU);
}

// fuse_layers = auto: the dynamically scheduled single launch pays from ~6 full-chip rounds of 8-tile groups per pass
#include "generic_host.hpp"

static int cf_fuse_min_groups(int n_cu) { return 6 * std::max(1, n_cu / 2); }

static int run_pass(cf_model* m, cf_model::Slot& sl, const float* x, int64_t n_windows, float* probs, float* logits, hipStream_t s) {
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    const int n_tiles32 = (int)((n_windows + 2 * CF_TILE - 1) / (2 * CF_TILE));
    int rc;
    size_t pi = 0;
    // residual blocks
    // latency mode (fp32): one workgroup per tile, four waves, each streaming a quarter of the 35 positions
    const bool res_split = m->np == 0 && n_tiles <= m->n_cu;
    const int res_chunks = res_split ? 4 : 1;
    const int res_waves = res_split ? 4 : (pick_waves(n_tiles, m->n_cu * 2) > 4 ? 4 : pick_waves(n_tiles, m->n_cu * 2));
    const int res_grid = res_split ? n_tiles : std::min((n_tiles + res_waves - 1) / res_waves, m->n_cu * 4);
    // throughput mode (fp32): the first two blocks as ONE launch, block 0's output stays in registers
    const int res_fuse_env = cf_knob("CATFISH_RES_FUSE") ? atoi(cf_knob("CATFISH_RES_FUSE")) : 1;   // A/B knob for tools/ and tests, read per pass
    const bool res_fused = m->np == 0 && m->hp.n_layers_res >= 2 && res_fuse_env != 0;
    const bool res_fused_bf16 = m->np > 0 && m->hp.n_layers_res >= 2 && res_fuse_env != 0;
    sl.last_res_fused = (res_fused && !res_split) || res_fused_bf16;       // (fp32 latency mode also stores block 0's output, for the debug hook)
    if (res_fused_bf16) {
        // blocks 0 and 1 as one launch on the bf16 matrix pipe; positions cut into chunks for about four waves per SIMD
        if ((rc = prof_begin(m, SLOT_RES_STACK2, s, &pi)) != CF_OK) return rc;
        // bf16: two tiles per wave (every LDS read of a weight fragment or bias vector feeds two tiles: the kernel is bound
        // by LDS bandwidth), two waves per SIMD; bf16x3: one tile per wave (twice the fragments), two waves per SIMD
        const int np = m->np > 1 ? 2 : 1;
        // A/B knob for tools/: two tiles per wave exist for one bf16 part only; any other value would launch a kernel that covers
        // 1 / tpw of the tiles and leave the rest of the slab stale
        const int tpw = (np == 1 && cf_knob("CATFISH_RES_TPW") && atoi(cf_knob("CATFISH_RES_TPW")) == 2) ? 2 : 1;
        const int groups = (n_tiles32 + tpw - 1) / tpw;
        const int slots = (np == 1 && tpw == 1 ? 4 * CF_RES_BF16_WAVES : 8) * m->n_cu;          // wave tasks resident at once on the whole chip
        // as many chunks as keep every task resident in ONE round (a second, mostly empty round costs a whole chunk's chain)
        int chunks = std::max(1, std::min(CF_T, slots / std::max(1, groups)));
        if (cf_knob("CATFISH_RES_CHUNKS")) chunks = std::max(1, std::min(CF_T, atoi(cf_knob("CATFISH_RES_CHUNKS"))));   // A/B knob for tools/
        const int len = (CF_T + chunks - 1) / chunks;
        chunks = (CF_T + len - 1) / len;
        const int lds_bytes = rb_pack_bytes(true, np) + rb_pack_bytes(false, np) + 4 * tpw * 32 * CF_T * 4;
        const int per_cu = std::max(1, std::min(np == 1 && tpw == 1 ? CF_RES_BF16_WAVES : 2, (160 * 1024) / lds_bytes));
        const int gridb = std::min((groups * chunks + 3) / 4, m->n_cu * per_cu);
        bf16x8* yb = reinterpret_cast<bf16x8*>(sl.d_a[1]);
        if (np == 2)
            hipLaunchKernelGGL((res_stack2_bf16_kernel<2, 1>), dim3(gridb), dim3(256), lds_bytes, s, m->d_res_b[0], m->d_res_b[1], x, yb,
                               n_windows, n_tiles32, chunks);
        else if (tpw == 2)
            hipLaunchKernelGGL((res_stack2_bf16_kernel<1, 2>), dim3(gridb), dim3(256), lds_bytes, s, m->d_res_b[0], m->d_res_b[1], x, yb,
                               n_windows, n_tiles32, chunks);
        else
            hipLaunchKernelGGL((res_stack2_bf16_kernel<1, 1>), dim3(gridb), dim3(256), lds_bytes, s, m->d_res_b[0], m->d_res_b[1], x, yb,
                               n_windows, n_tiles32, chunks);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    if (res_fused) {
        if ((rc = prof_begin(m, SLOT_RES_STACK2, s, &pi)) != CF_OK) return rc;
        // latency mode: chunks of positions, one wave each, spread over the idle CUs (about four waves per CU in all)
        int chunks = 1;
        if (res_split) {
            const int want = std::max(4, std::min(CF_T, (4 * m->n_cu) / std::max(1, n_tiles)));
            const int len = (CF_T + want - 1) / want;
            chunks = (CF_T + len - 1) / len;
        }
        const int waves2 = res_split ? 4 : res_waves;
        const int lds_bytes = (res_pack_floats(true) + res_pack_floats(false) + waves2 * CF_TILE * CF_T) * 4;
        const int grid2 = std::min((n_tiles * chunks + waves2 - 1) / waves2, m->n_cu * 3);     // 51 KB of LDS: three workgroups per CU
        if (res_split)
            hipLaunchKernelGGL(res_stack2_kernel<true>, dim3(grid2), dim3(waves2 * 64), lds_bytes, s, m->d_res[0], m->d_res[1], x,
                               reinterpret_cast<f32x4*>(sl.d_a[1]), reinterpret_cast<f32x4*>(sl.d_a[0]), n_windows, n_tiles, chunks);
        else
            hipLaunchKernelGGL(res_stack2_kernel<false>, dim3(grid2), dim3(waves2 * 64), lds_bytes, s, m->d_res[0], m->d_res[1], x,
                               reinterpret_cast<f32x4*>(sl.d_a[1]), (f32x4*)nullptr, n_windows, n_tiles, 1);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    for (int b = (res_fused || res_fused_bf16) ? 2 : 0; b < m->hp.n_layers_res; ++b) {
        float* dst = sl.d_a[b & 1];
        if (m->np > 0) {
            // residual blocks on the bf16 matrix pipe, 32-window tiles
            const int grid32 = std::min((n_tiles32 + 3) / 4, m->n_cu * 4);
            if ((rc = prof_begin(m, b == 0 ? SLOT_RES_FIRST : SLOT_RES, s, &pi)) != CF_OK) return rc;
            const bf16x8* src = b == 0 ? nullptr : reinterpret_cast<const bf16x8*>(sl.d_a[(b - 1) & 1]);
            if (b == 0) {
                const int lds_bytes = rb_pack_bytes(true, m->np > 1 ? 2 : 1) + 4 * 32 * CF_T * 4;
                if (m->np == 1)
                    hipLaunchKernelGGL((res_block_bf16_kernel<true, 1>), dim3(grid32), dim3(256), lds_bytes, s, m->d_res_b[0], x, src,
                                       reinterpret_cast<bf16x8*>(dst), n_windows, n_tiles32);
                else
                    hipLaunchKernelGGL((res_block_bf16_kernel<true, 2>), dim3(grid32), dim3(256), lds_bytes, s, m->d_res_b[0], x, src,
                                       reinterpret_cast<bf16x8*>(dst), n_windows, n_tiles32);
            } else {
                const int lds_bytes = rb_pack_bytes(false, m->np > 1 ? 2 : 1);
                if (m->np == 1)
                    hipLaunchKernelGGL((res_block_bf16_kernel<false, 1>), dim3(grid32), dim3(256), lds_bytes, s, m->d_res_b[b],
                                       (const float*)nullptr, src, reinterpret_cast<bf16x8*>(dst), n_windows, n_tiles32);
                else
                    hipLaunchKernelGGL((res_block_bf16_kernel<false, 2>), dim3(grid32), dim3(256), lds_bytes, s, m->d_res_b[b],
                                       (const float*)nullptr, src, reinterpret_cast<bf16x8*>(dst), n_windows, n_tiles32);
            }
        } else if (b == 0) {
            if ((rc = prof_begin(m, SLOT_RES_FIRST, s, &pi)) != CF_OK) return rc;
            const int lds_bytes = (res_pack_floats(true) + res_waves * CF_TILE * CF_T) * 4;
            hipLaunchKernelGGL((res_block_kernel<true>), dim3(res_grid), dim3(res_waves * 64), lds_bytes, s, m->d_res[0], x,
                               (const f32x4*)nullptr, reinterpret_cast<f32x4*>(dst), n_windows, n_tiles, res_chunks);
        } else {
            if ((rc = prof_begin(m, SLOT_RES, s, &pi)) != CF_OK) return rc;
            const int lds_bytes = res_pack_floats(false) * 4;
            hipLaunchKernelGGL((res_block_kernel<false>), dim3(res_grid), dim3(res_waves * 64), lds_bytes, s, m->d_res[b], (const float*)nullptr,
                               reinterpret_cast<const f32x4*>(sl.d_a[(b - 1) & 1]), reinterpret_cast<f32x4*>(dst), n_windows, n_tiles, res_chunks);
        }
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    if (m->hp.n_layers_res == 0) {
        if ((rc = prof_begin(m, SLOT_RES_FIRST, s, &pi)) != CF_OK) return rc;
        const int64_t n_el = (int64_t)n_tiles * CF_T * 64;
        hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, x,
                           reinterpret_cast<f32x4*>(sl.d_a[0]), n_windows, n_tiles);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    const float* cur = m->hp.n_layers_res == 0 ? sl.d_a[0] : sl.d_a[(m->hp.n_layers_res - 1) & 1];
    // GRU layers
    const bool fuse_now = m->fuse == 1 || (m->fuse == 2 && (n_tiles + 7) / 8 >= cf_fuse_min_groups(m->n_cu));
    if (fuse_now) {
        cf_fused_args a;
        for (int l = 0; l < 3; ++l) a.w[l] = l < m->hp.n_layers ? m->d_gru[l] : nullptr;
        a.x0 = reinterpret_cast<const f32x4*>(cur);
        a.y[0] = reinterpret_cast<f32x4*>(sl.d_y[0]);
        a.y[1] = reinterpret_cast<f32x4*>(sl.d_y[1]);
        a.p = sl.d_p;
        a.flags = sl.d_flags;
        a.err = m->d_err;
        a.n_tiles = n_tiles;
        a.groups = (n_tiles + 7) / 8;
        a.n_layers = m->hp.n_layers;
        a.cin0 = m->hp.n_layers_res == 0 ? 16 : CF_C;
        HIP_TRY(hipMemsetAsync(sl.d_flags, 0, ((size_t)a.n_layers * a.groups * 2 + (size_t)a.n_layers * 2) * sizeof(unsigned), s));
        if ((rc = prof_begin(m, SLOT_GRU_FUSED, s, &pi)) != CF_OK) return rc;
        a.lds_floats = m->hp.n_layers > 1 ? gru_pack_floats(128) : gru_pack_floats(a.cin0);
        const int lds_bytes = a.lds_floats * 4 + 16;
        const int pool = std::min(a.groups, std::max(1, m->n_cu / 2)) * 2;     // workgroups per layer (both directions)
        hipLaunchKernelGGL(gru_fused_kernel, dim3((unsigned)(a.n_layers * pool)), dim3(512), lds_bytes, s, a);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    for (int l = 0; l < (fuse_now ? 0 : m->hp.n_layers); ++l) {
        const bool last = l == m->hp.n_layers - 1;
        float* y = sl.d_y[l & 1];
        if (m->np > 0) {
            rc = m->np == 1 ? launch_gru_bf16_layer<1>(m, l, last, cur, y, sl.d_p, n_tiles32, s)
                            : launch_gru_bf16_layer<2>(m, l, last, cur, y, sl.d_p, n_tiles32, s);
        } else if (l == 0 && m->hp.n_layers_res == 0) {
            rc = last ? launch_gru<16, true>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU_LAST)
                      : launch_gru<16, false>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU0);
        } else if (l == 0) {
            rc = last ? launch_gru<32, true>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU_LAST)
                      : launch_gru<32, false>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU0);
        } else {
            rc = last ? launch_gru<128, true>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU_LAST)
                      : launch_gru<128, false>(m, m->d_gru[l], cur, y, sl.d_p, n_tiles, s, SLOT_GRU);
        }
        if (rc != CF_OK) return rc;
        cur = y;
    }
    // head
    if ((rc = prof_begin(m, SLOT_HEAD, s, &pi)) != CF_OK) return rc;
    const int64_t total = n_windows * CF_T;
    const int raw_partials = (m->np == 0 && !fuse_now && use_coop(m, n_tiles)) ? 1 : 0;     // what the last biGRU launch left in d_p
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sl.d_p, m->dense_bias, probs, logits, n_windows,
                       m->np > 0 ? n_tiles32 : n_tiles, m->np > 0 ? 5 : 4, raw_partials);
    HIP_TRY(hipGetLastError());
    if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    sl.last_windows = n_windows;
    return CF_OK;
}

static const char* k_fused_timeout_msg = "a fused biGRU launch timed out waiting for a producer workgroup; its results are invalid "
                                         "(create the model with fuse_layers = -1)";

extern "C" int cf_check_error(cf_model* m) {
    if (!m) return fail(CF_ERR_INVALID, "cf_check_error: null model");
    if (m->h_err && *m->h_err) return fail(CF_ERR_HIP, k_fused_timeout_msg);
    return CF_OK;
}

extern "C" int cf_clear_error(cf_model* m) {
    if (!m) return fail(CF_ERR_INVALID, "cf_clear_error: null model");
    if (m->h_err) *m->h_err = 0u;       // every fused launch re-initialises its own queues and flags: nothing else is stale
    return CF_OK;
}

extern "C" int cf_launch_regimes(const cf_model* m, int64_t out[4]) {
    if (!m || !out) return fail(CF_ERR_INVALID, "cf_launch_regimes: null argument");
    out[0] = m->n_cu;
    if (m->gen) { out[1] = out[2] = out[3] = 0; return CF_OK; }       // one launch shape at every size
    out[1] = (int64_t)m->xp_tiles * CF_TILE;
    out[2] = m->np == 0 ? (int64_t)m->n_cu * CF_TILE : 0;
    out[3] = m->fuse == 1 ? 1 : (m->fuse == 2 ? ((int64_t)8 * cf_fuse_min_groups(m->n_cu) - 8) * CF_TILE + 1 : 0);
    return CF_OK;
}

extern "C" int cf_infer_logits(cf_model* m, const float* x, int64_t n_windows, float* probs, float* logits, void* stream) {
    if (!m) return fail(CF_ERR_INVALID, "cf_infer: null model");
    if (n_windows < 0) return fail(CF_ERR_INVALID, "cf_infer: negative n_windows");
    if (n_windows == 0) return CF_OK;
    if (!x || (!probs && !logits)) return fail(CF_ERR_INVALID, "cf_infer: null buffer");
    if (m->h_err && *m->h_err) return fail(CF_ERR_HIP, k_fused_timeout_msg);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    m->prof = m->prof_every > 0 && (m->prof_calls++ % m->prof_every) == 0;
    if (m->gen) {
        for (int64_t off = 0; off < n_windows; off += m->cap_windows) {
            const int64_t n = std::min(m->cap_windows, n_windows - off);
            const int rc = gen_run_pass(m, x + off * CF_T, n, probs ? probs + off * CF_T : nullptr, logits ? logits + off * CF_T : nullptr, s);
            if (rc != CF_OK) return rc;
        }
        return CF_OK;
    }
    const int n_slots = (int)m->slots.size();
    // sub-batch size: split the call evenly over the slots, in whole 8-tile groups (128 windows)
    int64_t chunk = m->cap_windows;
    if (n_slots > 1) {
        const int64_t per = ((n_windows + n_slots - 1) / n_slots + 127) / 128 * 128;
        chunk = std::min(m->cap_windows, std::max<int64_t>(per, 128));
    }
    if (n_slots == 1 || n_windows <= 2048) {
        // small calls (a single read) are launch-bound: no fork/join
        for (int64_t off = 0; off < n_windows; off += m->cap_windows) {
            const int64_t n = std::min(m->cap_windows, n_windows - off);
            const int rc = run_pass(m, m->slots[0], x + off * CF_T, n, probs ? probs + off * CF_T : nullptr,
                                    logits ? logits + off * CF_T : nullptr, s);
            if (rc != CF_OK) return rc;
        }
        if (n_slots > 1) {
            // keep slot 0's scratch ordered against a later forked call on its internal stream
            HIP_TRY(hipEventRecord(m->slots[0].done, s));
            HIP_TRY(hipStreamWaitEvent(m->slots[0].stream, m->slots[0].done, 0));
        }
        return CF_OK;
    }
    // fork: every slot stream waits for the caller's stream, runs its sub-batches, joins back
    HIP_TRY(hipEventRecord(m->fork, s));
    for (auto& sl : m->slots) HIP_TRY(hipStreamWaitEvent(sl.stream, m->fork, 0));
    int k = 0;
    for (int64_t off = 0; off < n_windows; off += chunk, ++k) {
        const int64_t n = std::min(chunk, n_windows - off);
        cf_model::Slot& sl = m->slots[k % n_slots];
        const int rc = run_pass(m, sl, x + off * CF_T, n, probs ? probs + off * CF_T : nullptr, logits ? logits + off * CF_T : nullptr, sl.stream);
        if (rc != CF_OK) return rc;
    }
    for (auto& sl : m->slots) {
        HIP_TRY(hipEventRecord(sl.done, sl.stream));
        HIP_TRY(hipStreamWaitEvent(s, sl.done, 0));
    }
    return CF_OK;
}

extern "C" int cf_infer(cf_model* m, const float* x, int64_t n_windows, float* probs, void* stream) {
    if (n_windows > 0 && !probs) return fail(CF_ERR_INVALID, "cf_infer: null buffer");
    return cf_infer_logits(m, x, n_windows, probs, nullptr, stream);
}

extern "C" int cf_infer_host_logits(cf_model* m, const float* x, int64_t n_windows, float* probs, float* logits) {
    if (!m) return fail(CF_ERR_INVALID, "cf_infer_host: null model");
    if (n_windows < 0) return fail(CF_ERR_INVALID, "cf_infer_host: negative n_windows");
    if (n_windows == 0) return CF_OK;
    if (!x || (!probs && !logits)) return fail(CF_ERR_INVALID, "cf_infer_host: null buffer");
    HIP_TRY(hipSetDevice(m->device));
    const size_t bytes = (size_t)n_windows * CF_T * sizeof(float);
    // device staging buffers are kept across calls (the reference calls infer once per read): x, probs, logits
    if (m->host_stage_bytes < bytes) {
        if (m->d_host_x) (void)hipFree(m->d_host_x);
        m->d_host_x = m->d_host_p = nullptr;
        m->host_stage_bytes = 0;
        const size_t cap = (bytes + bytes / 4 + 255) / 256 * 256;
        hipError_t e2 = hipMalloc((void**)&m->d_host_x, 3 * cap);
        if (e2 != hipSuccess) { m->d_host_x = nullptr; return fail(CF_ERR_NOMEM, "cf_infer_host: hipMalloc failed"); }
        m->d_host_p = m->d_host_x + cap / sizeof(float);
        m->host_stage_bytes = cap;
    }
    float *dx = m->d_host_x, *dp = m->d_host_p, *dl = m->d_host_p + m->host_stage_bytes / sizeof(float);
    hipError_t e = hipSuccess;
    int rc = CF_OK;
    if ((e = hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice)) != hipSuccess) rc = fail(CF_ERR_HIP, hipGetErrorString(e));
    if (rc == CF_OK) rc = cf_infer_logits(m, dx, n_windows, probs ? dp : nullptr, logits ? dl : nullptr, nullptr);
    if (rc == CF_OK && (e = hipStreamSynchronize(nullptr)) != hipSuccess) rc = fail(CF_ERR_HIP, hipGetErrorString(e));
    if (rc == CF_OK) rc = cf_check_error(m);
    if (rc == CF_OK && probs && (e = hipMemcpy(probs, dp, bytes, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail(CF_ERR_HIP, hipGetErrorString(e));
    if (rc == CF_OK && logits && (e = hipMemcpy(logits, dl, bytes, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail(CF_ERR_HIP, hipGetErrorString(e));
    return rc;
}

extern "C" int cf_infer_host(cf_model* m, const float* x, int64_t n_windows, float* probs) {
    if (n_windows > 0 && !probs) return fail(CF_ERR_INVALID, "cf_infer_host: null buffer");
    return cf_infer_host_logits(m, x, n_windows, probs, nullptr);
}

extern "C" int cf_postprocess(cf_model* m, const float* probs, const int64_t* read_offsets, const int64_t* read_lengths,
                              int64_t n_reads, int64_t total_samples, float threshold, int32_t min_run, uint8_t* labels,
                              void* stream) {
    if (!m) return fail(CF_ERR_INVALID, "cf_postprocess: null model");
    if (n_reads < 0 || total_samples < 0) return fail(CF_ERR_INVALID, "cf_postprocess: negative size");
    if (n_reads == 0 || total_samples == 0) return CF_OK;
    if (min_run < 1) return fail(CF_ERR_INVALID, "cf_postprocess: min_run must be >= 1");
    if (!probs || !read_offsets || !read_lengths || !labels) return fail(CF_ERR_INVALID, "cf_postprocess: null buffer");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    size_t pi = 0;
    int rc = prof_begin(m, SLOT_POST, s, &pi);
    if (rc != CF_OK) return rc;
    const bool v1 = cf_knob("CATFISH_INGEST_V1") && atoi(cf_knob("CATFISH_INGEST_V1")) != 0;
    // the bit-mask kernel's window covers runs of up to 64 samples (the reference uses 15) and it stores labels 16 bytes at a time
    if (v1 || min_run > 64 || (reinterpret_cast<uintptr_t>(labels) & 15u) != 0) {
        hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, s, probs, read_offsets,
                           read_lengths, n_reads, total_samples, threshold, (int)min_run, labels);
    } else {
        const int64_t n_words = (total_samples + 63) / 64, n_chunks = (n_words + CF_POST_WORDS - 1) / CF_POST_WORDS;
        hipLaunchKernelGGL(postprocess_bits_kernel<false>, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, probs, read_offsets,
                           read_lengths, n_reads, total_samples, threshold, (int)min_run, labels, (int64_t)0, (int64_t*)nullptr,
                           (int64_t*)nullptr, (unsigned long long*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}

// cf_postprocess + cf_spans as ONE launch ("one segmented-scan kernel returning spans"): the corrected labels' run
// boundaries come straight from the bit masks; labels may be NULL when only the spans are wanted (the streaming pipeline).
extern "C" int cf_postprocess_spans(cf_model* m, const float* probs, const int64_t* read_offsets, const int64_t* read_lengths,
                                    int64_t n_reads, int64_t total_samples, float threshold, int32_t min_run, uint8_t* labels,
                                    int64_t max_runs, int64_t* starts, int64_t* ends, uint64_t* counts, void* stream) {
    if (!m) return fail(CF_ERR_INVALID, "cf_postprocess_spans: null model");
    if (n_reads < 0 || total_samples < 0 || max_runs < 0) return fail(CF_ERR_INVALID, "cf_postprocess_spans: negative size");
    if (!counts) return fail(CF_ERR_INVALID, "cf_postprocess_spans: null counts");
    if (min_run < 1) return fail(CF_ERR_INVALID, "cf_postprocess_spans: min_run must be >= 1");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), s));
    if (n_reads == 0 || total_samples == 0) return CF_OK;
    if (!probs || !read_offsets || !read_lengths || (max_runs > 0 && (!starts || !ends)))
        return fail(CF_ERR_INVALID, "cf_postprocess_spans: null buffer");
    const bool v1 = cf_knob("CATFISH_INGEST_V1") && atoi(cf_knob("CATFISH_INGEST_V1")) != 0;
    if (v1 || min_run > 64 || (reinterpret_cast<uintptr_t>(labels) & 15u) != 0) {
        // what the bit-mask kernel does not cover (min_run > 64, an unaligned labels pointer) or is told not to take (the
        // CATFISH_INGEST_V1 A/B knob) goes through the two calls it replaces.  They need a labels buffer: a caller that asked for the
        // spans only (labels NULL -- the streaming pipeline) gets a stream-ordered temporary one.
        uint8_t* lab = labels;
        if (!lab) {
            hipError_t e = hipMallocAsync((void**)&lab, (size_t)total_samples, s);
            if (e != hipSuccess) return fail(CF_ERR_NOMEM, std::string("cf_postprocess_spans: no device memory for the temporary labels: ") + hipGetErrorString(e));
        }
        int rc = cf_postprocess(m, probs, read_offsets, read_lengths, n_reads, total_samples, threshold, min_run, lab, stream);
        if (rc == CF_OK) rc = cf_spans(m, lab, total_samples, max_runs, starts, ends, counts, stream);
        if (rc == CF_OK && n_reads > 1) {
            // cf_spans sees labels only: where an UNPADDED read's last sample and the next read's first are both positive it found one
            // run; cut it at the read boundary (one end + one start more; the lists are sorted before pairing, so order is free)
            hipLaunchKernelGGL(spans_cut_kernel, dim3((unsigned)((n_reads - 1 + 255) / 256)), dim3(256), 0, s, lab, read_offsets, read_lengths,
                               n_reads, total_samples, max_runs, starts, ends, reinterpret_cast<unsigned long long*>(counts));
            if (hipGetLastError() != hipSuccess) rc = fail(CF_ERR_HIP, "cf_postprocess_spans: launch failed");
        }
        if (!labels) (void)hipFreeAsync(lab, s);
        return rc;
    }
    size_t pi = 0;
    int rc = prof_begin(m, SLOT_POST, s, &pi);
    if (rc != CF_OK) return rc;
    const int64_t n_words = (total_samples + 63) / 64, n_chunks = (n_words + CF_POST_WORDS - 1) / CF_POST_WORDS;
    hipLaunchKernelGGL(postprocess_bits_kernel<true>, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, s, probs, read_offsets, read_lengths,
                       n_reads, total_samples, threshold, (int)min_run, labels, max_runs, starts, ends,
                       reinterpret_cast<unsigned long long*>(counts));
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}

extern "C" int cf_spans(cf_model* m, const uint8_t* labels, int64_t total_samples, int64_t max_runs, int64_t* starts,
                        int64_t* ends, uint64_t* counts, void* stream) {
    if (!m) return fail(CF_ERR_INVALID, "cf_spans: null model");
    if (total_samples < 0 || max_runs < 0) return fail(CF_ERR_INVALID, "cf_spans: negative size");
    if (!counts) return fail(CF_ERR_INVALID, "cf_spans: null counts");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), s));
    if (total_samples == 0) return CF_OK;
    if (!labels || (max_runs > 0 && (!starts || !ends))) return fail(CF_ERR_INVALID, "cf_spans: null buffer");
    const int64_t per_wg = 256 * CF_SPANS_PER_THREAD;
    hipLaunchKernelGGL(spans_kernel, dim3((unsigned)((total_samples + per_wg - 1) / per_wg)), dim3(256), 0, s, labels, total_samples, max_runs,
                       starts, ends, reinterpret_cast<unsigned long long*>(counts));
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_normalize(cf_model* m, const int16_t* dac, const int64_t* dac_offsets, const int64_t* win_offsets,
                            int64_t n_reads, float* x_out, void* stream) {
    if (!m) return fail(CF_ERR_INVALID, "cf_normalize: null model");
    if (n_reads < 0) return fail(CF_ERR_INVALID, "cf_normalize: negative n_reads");
    if (n_reads == 0) return CF_OK;
    if (!dac || !dac_offsets || !win_offsets || !x_out) return fail(CF_ERR_INVALID, "cf_normalize: null buffer");
    if (n_reads > 0x7FFFFFFF) return fail(CF_ERR_INVALID, "cf_normalize: too many reads in one call");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    size_t pi = 0;
    int rc = prof_begin(m, SLOT_NORM, s, &pi);
    if (rc != CF_OK) return rc;
    if (cf_knob("CATFISH_INGEST_V1") && atoi(cf_knob("CATFISH_INGEST_V1")) != 0)      // A/B and bit-identity tests: round 1's kernel
        hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)n_reads), dim3(256), 0, s, dac, dac_offsets, win_offsets, x_out);
    else {                                                                              // reads <= 4096 samples, then the longer ones
        hipLaunchKernelGGL(normalize_regs_kernel<CF_NORM_REGS_SMALL>, dim3((unsigned)n_reads), dim3(256), 0, s, dac, dac_offsets, win_offsets, x_out);
        hipLaunchKernelGGL(normalize_regs_kernel<CF_NORM_REGS_LARGE>, dim3((unsigned)n_reads), dim3(256), 0, s, dac, dac_offsets, win_offsets, x_out);
    }
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}

// ---- training support (BASELINE config 5) ------------------------------------------------
// Gather map of the weight re-tiling, so that a trainer can re-tile updated weights ON DEVICE
// (packed = src[idx] * scale) instead of round-tripping through the host packers every step.
// src (one direction) = [gates_kernel ((cin+64)*128) | candidate_kernel ((cin+64)*64) | gates_bias (128) |
// candidate_bias (64) | 0.0]; the map is derived from the host packers themselves (index-encoded weights).
extern "C" int cf_gru_pack_map(int32_t cin, int32_t backward, int32_t* idx, float* scale, int64_t capacity, int64_t* n_floats) {
    if (cin != CF_C && cin != 2 * CF_H) return fail(CF_ERR_INVALID, "cf_gru_pack_map: cin must be 32 or 128");
    const int64_t n = backward ? gtb_pack_floats(cin) : gru_pack_floats(cin);
    if (n_floats) *n_floats = n;
    if (!idx || !scale) return CF_OK;                       // size query
    if (capacity < n) return fail(CF_ERR_INVALID, "cf_gru_pack_map: buffers too small");
    const int rows = cin + CF_H;
    const size_t n_gk = (size_t)rows * 2 * CF_H, n_ck = (size_t)rows * CF_H;
    const size_t n_src = n_gk + n_ck + 2 * CF_H + CF_H;     // + the trailing zero slot = index n_src
    std::vector<float> gk(n_gk), ck(n_ck), gb(2 * CF_H), cb(CF_H);
    cf_gru_dir g;
    g.gates_kernel = gk.data(); g.gates_bias = gb.data(); g.candidate_kernel = ck.data(); g.candidate_bias = cb.data(); g.cin = cin;
    std::vector<float> ones((size_t)n), enc((size_t)n);
    auto run = [&](std::vector<float>& out) {
        if (backward) pack_gru_dir_bwd(g, cin, out.data());
        else pack_gru_dir(g, cin, cin, nullptr, out.data());
    };
    for (auto* v : {&gk, &ck, &gb, &cb}) std::fill(v->begin(), v->end(), 1.0f);
    run(ones);                                              // = the scale applied to every packed element (0 = constant zero)
    size_t base = 0;
    for (auto* v : {&gk, &ck, &gb, &cb}) { for (size_t i = 0; i < v->size(); ++i) (*v)[i] = (float)(base + i + 1); base += v->size(); }
    run(enc);
    for (int64_t i = 0; i < n; ++i) {
        scale[i] = ones[i];
        idx[i] = ones[i] == 0.f ? (int32_t)n_src : (int32_t)(std::llround((double)enc[i] / (double)ones[i]) - 1);
        if (idx[i] < 0 || idx[i] > (int32_t)n_src) return fail(CF_ERR_INVALID, "cf_gru_pack_map: internal index error");
    }
    return CF_OK;
}

static int train_cin_ok(const cf_model* m, int cin) {
    if (m->np != 0) return fail(CF_ERR_INVALID, "training kernels need a CF_PREC_FP32 model");
    if (m->hp.layer_size != CF_H) return fail(CF_ERR_INVALID, "training kernels: layer_size 64 only (other sizes: cf_gru_anysize_train_*)");
    if (m->gen) return fail(CF_ERR_INVALID, "training kernels need a model on the tuned path (unset CATFISH_GENERIC)");
    if (cin != CF_C && cin != 2 * CF_H) return fail(CF_ERR_INVALID, "training kernels: cin must be 32 or 128");
    return CF_OK;
}

static cf_dropout make_dropout(float keep_prob, uint32_t seed, int32_t layer, const double* step_count) {
    cf_dropout d;
    d.keep_prob = keep_prob;
    d.seed = seed;
    d.layer = layer;
    d.step = step_count;
    return d;
}

extern "C" int cf_gru_train_forward_dropout(cf_model* m, int32_t cin, const float* wpack, const float* x_frag, float* y_frag, float* stash,
                                            int64_t n_windows, float* y_drop_frag, float keep_prob, uint32_t seed, int32_t layer,
                                            const double* step_count, void* stream) {
    if (!m || !wpack || !x_frag || !y_frag || !stash) return fail(CF_ERR_INVALID, "cf_gru_train_forward: null argument");
    if (y_drop_frag && !(keep_prob > 0.f && keep_prob < 1.f)) return fail(CF_ERR_INVALID, "cf_gru_train_forward_dropout: keep_prob must be in (0, 1)");
    f32x4* yd = reinterpret_cast<f32x4*>(y_drop_frag);
    const cf_dropout drop = make_dropout(y_drop_frag ? keep_prob : 1.f, seed, layer, step_count);
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_gru_train_forward: n_windows must be positive");
    int rc = train_cin_ok(m, cin);
    if (rc != CF_OK) return rc;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    if (n_tiles <= m->n_cu) {                // small batch: four waves per tile (latency mode), up to two rounds
        const int gxc = std::min(n_tiles, std::max(1, m->n_cu / 2));
        const bool hoist = n_tiles <= m->xp_tiles;
        const f32x4* xp = hoist ? reinterpret_cast<const f32x4*>(m->d_xp) : nullptr;
        if (cin == CF_C) {
            if (hoist) launch_xproj<32>(m, wpack, x_frag, n_tiles, s);
            hipLaunchKernelGGL((gru_train_fwd_coop_kernel<32>), dim3(gxc, 2), dim3(256), (gru_pack_floats(32) + CF_COOP_XCH_FLOATS) * 4, s, wpack,
                               reinterpret_cast<const f32x4*>(x_frag), reinterpret_cast<f32x4*>(y_frag), reinterpret_cast<f32x4*>(stash), n_tiles, xp, yd, drop);
        } else {
            if (hoist) launch_xproj<128>(m, wpack, x_frag, n_tiles, s);
            hipLaunchKernelGGL((gru_train_fwd_coop_kernel<128>), dim3(gxc, 2), dim3(256), (gru_pack_floats(128) + CF_COOP_XCH_FLOATS) * 4, s, wpack,
                               reinterpret_cast<const f32x4*>(x_frag), reinterpret_cast<f32x4*>(y_frag), reinterpret_cast<f32x4*>(stash), n_tiles, xp, yd, drop);
        }
        HIP_TRY(hipGetLastError());
        return CF_OK;
    }
    const int waves = pick_waves(2 * n_tiles, m->n_cu);
    const int gx = std::min((n_tiles + waves - 1) / waves, std::max(1, m->n_cu / 2));
    if (cin == CF_C)
        hipLaunchKernelGGL((gru_train_fwd_kernel<32>), dim3(gx, 2), dim3(waves * 64), gru_pack_floats(32) * 4, s, wpack,
                           reinterpret_cast<const f32x4*>(x_frag), reinterpret_cast<f32x4*>(y_frag), reinterpret_cast<f32x4*>(stash), n_tiles, yd, drop);
    else
        hipLaunchKernelGGL((gru_train_fwd_kernel<128>), dim3(gx, 2), dim3(waves * 64), gru_pack_floats(128) * 4, s, wpack,
                           reinterpret_cast<const f32x4*>(x_frag), reinterpret_cast<f32x4*>(y_frag), reinterpret_cast<f32x4*>(stash), n_tiles, yd, drop);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_gru_train_forward(cf_model* m, int32_t cin, const float* wpack, const float* x_frag, float* y_frag, float* stash,
                                    int64_t n_windows, void* stream) {
    return cf_gru_train_forward_dropout(m, cin, wpack, x_frag, y_frag, stash, n_windows, nullptr, 1.f, 0u, 0, nullptr, stream);
}

// chunks of steps per (tile, direction) of the deferred input-gradient launch (gru_dx_kernel: launch shape only, results do not depend on it)
// -- about 2.25 workgroups per CU: at the reference's batch (256 windows = 16 tiles) 18 chunks instead of round 5's 7 took the training
// step from 0.756 to 0.738 ms (profiles/r06_train_dx_chunks.log); few tiles keep one step per workgroup, many keep 7 chunks
static int dx_chunks(int n_tiles, int n_cu) {
    if (cf_knob("CATFISH_DX_CHUNKS")) return std::max(1, std::min(atoi(cf_knob("CATFISH_DX_CHUNKS")), CF_T));      // A/B knob for tools/
    const int want = (9 * std::max(1, n_cu) / 4 + 2 * n_tiles - 1) / (2 * std::max(1, n_tiles));
    return std::max(7, std::min(CF_T, want));
}

extern "C" int cf_gru_train_backward_dropout(cf_model* m, int32_t cin, const float* wpack_bwd, const float* y_frag, const float* stash,
                                             const float* dy_frag, const float* dy2_frag, const float* dy_scale, float* dx_frag, float* da,
                                             int64_t n_windows, float keep_prob, uint32_t seed, int32_t layer, const double* step_count,
                                             void* stream) {
    const cf_dropout drop = make_dropout((keep_prob > 0.f && keep_prob < 1.f) ? keep_prob : 1.f, seed, layer, step_count);
    if (!m || !wpack_bwd || !y_frag || !stash || !dy_frag || !dx_frag || !da)
        return fail(CF_ERR_INVALID, "cf_gru_train_backward: null argument");
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_gru_train_backward: n_windows must be positive");
    int rc = train_cin_ok(m, cin);
    if (rc != CF_OK) return rc;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    if (n_tiles <= m->n_cu) {                // small batch: four waves per tile (latency mode), up to two rounds
        const int gxc = std::min(n_tiles, std::max(1, m->n_cu / 2));
        const int defer = n_tiles <= m->xp_tiles ? 1 : 0;                    // few tiles: dx is formed afterwards on the idle CUs
        const int dxc = dx_chunks(n_tiles, m->n_cu);
        if (cin == CF_C) {
            hipLaunchKernelGGL((gru_train_bwd_coop_kernel<32>), dim3(gxc, 2), dim3(256), (gtb_pack_floats(32) + CF_COOP_BWD_XCH_FLOATS) * 4, s,
                               wpack_bwd, reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash),
                               reinterpret_cast<const f32x4*>(dy_frag), reinterpret_cast<const f32x4*>(dy2_frag),
                               reinterpret_cast<const f32x4*>(dy_scale), reinterpret_cast<f32x4*>(dx_frag), reinterpret_cast<f32x4*>(da), n_tiles, defer, drop);
            if (defer)
                hipLaunchKernelGGL((gru_dx_kernel<32>), dim3(n_tiles * dxc, 2), dim3(256), 0, s, wpack_bwd, reinterpret_cast<const f32x4*>(da),
                                   reinterpret_cast<f32x4*>(dx_frag), n_tiles, dxc);
        } else {
            hipLaunchKernelGGL((gru_train_bwd_coop_kernel<128>), dim3(gxc, 2), dim3(256), (gtb_pack_floats(128) + CF_COOP_BWD_XCH_FLOATS) * 4, s,
                               wpack_bwd, reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash),
                               reinterpret_cast<const f32x4*>(dy_frag), reinterpret_cast<const f32x4*>(dy2_frag),
                               reinterpret_cast<const f32x4*>(dy_scale), reinterpret_cast<f32x4*>(dx_frag), reinterpret_cast<f32x4*>(da), n_tiles, defer, drop);
            if (defer)
                hipLaunchKernelGGL((gru_dx_kernel<128>), dim3(n_tiles * dxc, 2), dim3(256), 0, s, wpack_bwd, reinterpret_cast<const f32x4*>(da),
                                   reinterpret_cast<f32x4*>(dx_frag), n_tiles, dxc);
        }
        HIP_TRY(hipGetLastError());
        return CF_OK;
    }
    const int waves = pick_waves(2 * n_tiles, m->n_cu);
    const int gx = std::min((n_tiles + waves - 1) / waves, std::max(1, m->n_cu / 2));
    if (cin == CF_C)
        hipLaunchKernelGGL((gru_train_bwd_kernel<32>), dim3(gx, 2), dim3(waves * 64), gtb_pack_floats(32) * 4, s, wpack_bwd,
                           reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash),
                           reinterpret_cast<const f32x4*>(dy_frag), reinterpret_cast<const f32x4*>(dy2_frag),
                               reinterpret_cast<const f32x4*>(dy_scale), reinterpret_cast<f32x4*>(dx_frag), reinterpret_cast<f32x4*>(da), n_tiles, drop);
    else
        hipLaunchKernelGGL((gru_train_bwd_kernel<128>), dim3(gx, 2), dim3(waves * 64), gtb_pack_floats(128) * 4, s, wpack_bwd,
                           reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash),
                           reinterpret_cast<const f32x4*>(dy_frag), reinterpret_cast<const f32x4*>(dy2_frag),
                               reinterpret_cast<const f32x4*>(dy_scale), reinterpret_cast<f32x4*>(dx_frag), reinterpret_cast<f32x4*>(da), n_tiles, drop);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_gru_train_backward(cf_model* m, int32_t cin, const float* wpack_bwd, const float* y_frag, const float* stash,
                                     const float* dy_frag, const float* dy2_frag, const float* dy_scale, float* dx_frag, float* da,
                                     int64_t n_windows, void* stream) {
    return cf_gru_train_backward_dropout(m, cin, wpack_bwd, y_frag, stash, dy_frag, dy2_frag, dy_scale, dx_frag, da, n_windows, 1.f, 0u, 0,
                                         nullptr, stream);
}

// The mask / keep_prob tensor the training kernels apply to the output of `layer` (fragment layout [tiles][35][8][64][4]), written out
// for tests and tools: the kernels themselves never store it.
// ---- any-size biGRU layer for training (generic.hpp): forward with gate stash, backward chain -----------------------------------
static int anysize_waves(int h16, int arrays, size_t* lds) {
    const size_t per_wave = (size_t)arrays * h16 * 64 * sizeof(f32x4);
    const int waves = per_wave * 8 <= (size_t)(160 * 1024) ? 8 : (per_wave * 4 <= (size_t)(160 * 1024) ? 4 : (per_wave * 2 <= (size_t)(160 * 1024) ? 2 : 1));
    *lds = per_wave * waves;
    return waves;
}

static int anysize_args_ok(const cf_model* m, int32_t layer_size, int64_t n_windows, const char* who) {
    if (!m) return fail(CF_ERR_INVALID, std::string(who) + ": null model");
    if (layer_size < 16 || layer_size > 256 || (layer_size % 16) != 0)
        return fail(CF_ERR_INVALID, std::string(who) + ": layer_size must be a multiple of 16 between 16 and 256");
    if (n_windows <= 0 || (n_windows % CF_TILE) != 0) return fail(CF_ERR_INVALID, std::string(who) + ": n_windows must be a positive multiple of 16");
    return CF_OK;
}

extern "C" int cf_gru_anysize_train_forward(cf_model* m, int32_t layer_size, int32_t cin_blocks, const float* wpack, const float* bpack,
                                            const float* x_frag, float* y_frag, float* stash, int64_t n_windows, void* stream) {
    int rc = anysize_args_ok(m, layer_size, n_windows, "cf_gru_anysize_train_forward");
    if (rc != CF_OK) return rc;
    if (!wpack || !bpack || !x_frag || !y_frag || !stash) return fail(CF_ERR_INVALID, "cf_gru_anysize_train_forward: null buffer");
    if (cin_blocks < 1 || cin_blocks > 32) return fail(CF_ERR_INVALID, "cf_gru_anysize_train_forward: cin_blocks must be 1..32 (input features / 16)");
    HIP_TRY(hipSetDevice(m->device));
    const int h16 = layer_size / 16, n_tiles = (int)(n_windows / CF_TILE);
    size_t lds_full = 0;
    int h_via_y = 0;
    int max_waves = anysize_waves(h16, 3, &lds_full);
    if (max_waves < 8) { h_via_y = 1; max_waves = anysize_waves(h16, 2, &lds_full); }     // as in the inference launch: h' through y above 64 units
    const int waves = std::max(1, std::min(max_waves, (2 * n_tiles + m->n_cu - 1) / m->n_cu));
    const size_t lds = lds_full / max_waves * waves;
    HIP_TRY(hipFuncSetAttribute((const void*)gen_gru_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(gen_gru_kernel<true>, dim3((unsigned)((n_tiles + waves - 1) / waves), 2), dim3(waves * 64), lds,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f32x4*>(wpack), reinterpret_cast<const f32x4*>(bpack),
                       reinterpret_cast<const f32x4*>(x_frag), reinterpret_cast<f32x4*>(y_frag), h16, (int)cin_blocks, h_via_y,
                       reinterpret_cast<f32x4*>(stash), n_tiles);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_gru_anysize_train_backward(cf_model* m, int32_t layer_size, const float* wtpack, const float* y_frag, const float* stash,
                                             const float* dy_frag, float* da, int64_t n_windows, void* stream) {
    int rc = anysize_args_ok(m, layer_size, n_windows, "cf_gru_anysize_train_backward");
    if (rc != CF_OK) return rc;
    if (!wtpack || !y_frag || !stash || !dy_frag || !da) return fail(CF_ERR_INVALID, "cf_gru_anysize_train_backward: null buffer");
    HIP_TRY(hipSetDevice(m->device));
    const int h16 = layer_size / 16, n_tiles = (int)(n_windows / CF_TILE);
    size_t lds = 0;
    const int max_waves = anysize_waves(h16, 4, &lds);
    const int waves = std::max(1, std::min(max_waves, (2 * n_tiles + m->n_cu - 1) / m->n_cu));
    lds = lds / max_waves * waves;
    HIP_TRY(hipFuncSetAttribute((const void*)gen_gru_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(gen_gru_bwd_kernel, dim3((unsigned)((n_tiles + waves - 1) / waves), 2), dim3(waves * 64), lds,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f32x4*>(wtpack), reinterpret_cast<const f32x4*>(y_frag),
                       reinterpret_cast<const f32x4*>(stash), reinterpret_cast<const f32x4*>(dy_frag), reinterpret_cast<f32x4*>(da), n_tiles, h16);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

__global__ __launch_bounds__(256) void dropout_scale_kernel(f32x4* __restrict__ out, int64_t n4, cf_dropout drop) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = cf_drop_scale4(cf_drop_key(drop), drop.keep_prob, i);
}

extern "C" int cf_dropout_scale(cf_model* m, float keep_prob, uint32_t seed, int32_t layer, const double* step_count, int64_t n_windows,
                                float* scale_frag, void* stream) {
    if (!m || !scale_frag) return fail(CF_ERR_INVALID, "cf_dropout_scale: null argument");
    if (!(keep_prob > 0.f && keep_prob < 1.f) || n_windows <= 0) return fail(CF_ERR_INVALID, "cf_dropout_scale: keep_prob must be in (0, 1), n_windows > 0");
    HIP_TRY(hipSetDevice(m->device));
    const int64_t n4 = ((n_windows + CF_TILE - 1) / CF_TILE) * CF_T * 8 * 64;
    hipLaunchKernelGGL(dropout_scale_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<f32x4*>(scale_frag), n4, make_dropout(keep_prob, seed, layer, step_count));
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

static int wgrad_pairs_per_wg(int n_pairs, int n_cu) {
    // two directions x n_chunks workgroups should cover the chip about once; at least 8 pairs each keeps the
    // partial sums (148 KiB per workgroup) well below the traffic of the operands themselves
    return std::max(8, (2 * n_pairs + n_cu - 1) / n_cu);
}

extern "C" int64_t cf_gru_wgrad_workspace_floats(cf_model* m, int32_t cin, int64_t n_windows) {
    if (!m || (cin != CF_C && cin != 2 * CF_H) || n_windows <= 0) return 0;
    const int n_pairs = (int)((n_windows + CF_TILE - 1) / CF_TILE) * CF_T;
    const int ppw = wgrad_pairs_per_wg(n_pairs, m->n_cu);
    const int64_t n_chunks = (n_pairs + ppw - 1) / ppw;
    return n_chunks * 2 * gwg_partial_floats(cin);
}

extern "C" int cf_gru_train_wgrad(cf_model* m, int32_t cin, const float* x_frag, const float* y_frag, const float* stash,
                                  const float* da, int64_t n_windows, float* workspace, int64_t workspace_floats, float* grads,
                                  void* stream) {
    if (!m || !x_frag || !y_frag || !stash || !da || !workspace || !grads)
        return fail(CF_ERR_INVALID, "cf_gru_train_wgrad: null argument");
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_gru_train_wgrad: n_windows must be positive");
    int rc = train_cin_ok(m, cin);
    if (rc != CF_OK) return rc;
    if (workspace_floats < cf_gru_wgrad_workspace_floats(m, cin, n_windows))
        return fail(CF_ERR_INVALID, "cf_gru_train_wgrad: workspace too small (see cf_gru_wgrad_workspace_floats)");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    const int n_pairs = n_tiles * CF_T;
    const int ppw = wgrad_pairs_per_wg(n_pairs, m->n_cu);
    const int n_chunks = (n_pairs + ppw - 1) / ppw;
    if (cin == CF_C)
        hipLaunchKernelGGL((gru_wgrad_kernel<32>), dim3(n_chunks, 2), dim3(CF_WGRAD_WAVES * 64), 0, s, reinterpret_cast<const f32x4*>(x_frag),
                           reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash), reinterpret_cast<const f32x4*>(da),
                           workspace, n_tiles, ppw);
    else
        hipLaunchKernelGGL((gru_wgrad_kernel<128>), dim3(n_chunks, 2), dim3(CF_WGRAD_WAVES * 64), 0, s, reinterpret_cast<const f32x4*>(x_frag),
                           reinterpret_cast<const f32x4*>(y_frag), reinterpret_cast<const f32x4*>(stash), reinterpret_cast<const f32x4*>(da),
                           workspace, n_tiles, ppw);
    HIP_TRY(hipGetLastError());
    const int per = gwg_partial_floats(cin);
    hipLaunchKernelGGL(gru_wgrad_reduce_kernel, dim3((per + 63) / 64, 2), dim3(256), 0, s, workspace, grads, gwg_rows(cin), n_chunks);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

// ---- residual conv stack, training ---------------------------------------------------------
static int res_train_ok(const cf_model* m, int n_blocks) {
    if (n_blocks < 1 || 4 * n_blocks > RT_MAX_UNITS) return fail(CF_ERR_INVALID, "residual training kernels: 1..4 blocks");
    if (m->hp.layer_size_res != CF_C) return fail(CF_ERR_INVALID, "residual training kernels: 32 conv channels only");
    if (m->gen) return fail(CF_ERR_INVALID, "residual training kernels need a model on the tuned path (unset CATFISH_GENERIC)");
    return CF_OK;
}

extern "C" int64_t cf_res_train_param_floats(int32_t n_blocks) {
    if (n_blocks < 1 || 4 * n_blocks > RT_MAX_UNITS) return 0;
    return rt_make_layout(n_blocks).off[4 * n_blocks];
}

extern "C" int64_t cf_res_train_workspace_floats(int32_t n_blocks, int64_t n_windows) {
    if (n_blocks < 1 || 4 * n_blocks > RT_MAX_UNITS || n_windows <= 0) return 0;
    const int win = rt_win_for(n_windows);
    return ((n_windows + win - 1) / win) * (int64_t)rt_make_layout(n_blocks).off[4 * n_blocks];
}

extern "C" int cf_res_train_forward(cf_model* m, int32_t n_blocks, const float* params, const float* x, float* z_stash, float* out,
                                    int64_t n_windows, void* stream) {
    if (!m || !params || !x || !z_stash || !out) return fail(CF_ERR_INVALID, "cf_res_train_forward: null argument");
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_res_train_forward: n_windows must be positive");
    int rc = res_train_ok(m, n_blocks);
    if (rc != CF_OK) return rc;
    HIP_TRY(hipSetDevice(m->device));
    const int win = rt_win_for(n_windows);
    const int n_wg = (int)((n_windows + win - 1) / win);
    hipLaunchKernelGGL(res_train_fwd_kernel, dim3(n_wg), dim3(RT_THREADS), RT_FWD_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), x, params,
                       z_stash, out, rt_make_layout(n_blocks), (int)n_windows, n_blocks, m->hp.bn_epsilon, win);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_res_train_backward(cf_model* m, int32_t n_blocks, const float* params, const float* x, const float* z_stash,
                                     const float* d_out, float* workspace, int64_t workspace_floats, float* grads, int64_t n_windows,
                                     void* stream) {
    if (!m || !params || !x || !z_stash || !d_out || !workspace || !grads) return fail(CF_ERR_INVALID, "cf_res_train_backward: null argument");
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_res_train_backward: n_windows must be positive");
    int rc = res_train_ok(m, n_blocks);
    if (rc != CF_OK) return rc;
    if (workspace_floats < cf_res_train_workspace_floats(n_blocks, n_windows))
        return fail(CF_ERR_INVALID, "cf_res_train_backward: workspace too small (see cf_res_train_workspace_floats)");
    HIP_TRY(hipSetDevice(m->device));
    const int lds_bytes = RT_BWD_LDS_BYTES;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const rt_layout L = rt_make_layout(n_blocks);
    const int win = rt_win_for(n_windows);
    const int n_wg = (int)((n_windows + win - 1) / win);
    hipLaunchKernelGGL(res_train_bwd_kernel, dim3(n_wg), dim3(RT_THREADS), lds_bytes, s, x, params, z_stash, d_out, workspace, L,
                       (int)n_windows, n_blocks, m->hp.bn_epsilon, win);
    HIP_TRY(hipGetLastError());
    const int nf = L.off[4 * n_blocks];
    hipLaunchKernelGGL(res_train_reduce_kernel, dim3((nf + 63) / 64), dim3(256), 0, s, workspace, grads, nf, n_wg);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

// ---- dense head + loss, optimizer (training) ------------------------------------------------
static int head_waves(const cf_model* m, int64_t n_windows) {
    // a wave takes ~8 (tile, t) items: few enough partial sums that the fixed-order reduction stays a couple of microseconds
    const int64_t items = ((n_windows + CF_TILE - 1) / CF_TILE) * CF_T;
    const int64_t wgs = std::min<int64_t>((items + 31) / 32, (int64_t)m->n_cu * 2);
    return (int)std::max<int64_t>(1, wgs) * 4;
}

extern "C" int64_t cf_train_head_workspace_floats(cf_model* m, int64_t n_windows) {
    if (!m || n_windows <= 0) return 0;
    return (int64_t)head_waves(m, n_windows) * CF_HEAD_PART;
}

extern "C" int cf_train_head(cf_model* m, const float* y_frag, const float* dense_kernel, const float* dense_bias, const float* labels,
                             int64_t n_windows, float* dy_frag, float* logits, float* workspace, int64_t workspace_floats, float* grads,
                             float* loss, void* stream) {
    if (!m || !y_frag || !dense_kernel || !dense_bias || !labels || !dy_frag || !workspace || !grads || !loss)
        return fail(CF_ERR_INVALID, "cf_train_head: null argument");
    if (n_windows <= 0) return fail(CF_ERR_INVALID, "cf_train_head: n_windows must be positive");
    if (m->hp.layer_size != CF_H) return fail(CF_ERR_INVALID, "cf_train_head: layer_size 64 only");
    if (workspace_floats < cf_train_head_workspace_floats(m, n_windows))
        return fail(CF_ERR_INVALID, "cf_train_head: workspace too small (see cf_train_head_workspace_floats)");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    const int waves = head_waves(m, n_windows);
    const float inv_count = (float)(1.0 / ((double)n_windows * CF_T));
    hipLaunchKernelGGL(train_head_kernel, dim3(waves / 4), dim3(256), 0, s, reinterpret_cast<const f32x4*>(y_frag), dense_kernel, dense_bias,
                       labels, reinterpret_cast<f32x4*>(dy_frag), logits, workspace, n_windows, n_tiles, inv_count);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(train_head_reduce_kernel, dim3(3), dim3(256), 0, s, workspace, waves, grads, loss, inv_count);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

extern "C" int cf_opt_step(cf_model* m, int32_t kind, float* params, const float* grads, float* slot1, float* slot2, int64_t n, float lr,
                           double* step_count, const int32_t* pack_idx, const float* pack_scale, float* packed, int64_t n_packed,
                           void* stream) {
    if (!m || !params || !grads || !slot1 || !slot2 || !step_count) return fail(CF_ERR_INVALID, "cf_opt_step: null argument");
    if (kind != 0 && kind != 1) return fail(CF_ERR_INVALID, "cf_opt_step: kind must be 0 (RMSProp) or 1 (Adam)");
    if (n <= 0 || n_packed < 0) return fail(CF_ERR_INVALID, "cf_opt_step: bad size");
    if (n_packed > 0 && (!pack_idx || !pack_scale || !packed)) return fail(CF_ERR_INVALID, "cf_opt_step: null packing argument");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(opt_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int)kind, params, grads, slot1, slot2, n, lr,
                       step_count);
    HIP_TRY(hipGetLastError());
    // the re-tiling launch also advances the step counter (it runs after every block of the update has read it)
    const int64_t ng = std::max<int64_t>(n_packed, 1);
    hipLaunchKernelGGL(gather_scale_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, params, pack_idx, pack_scale, packed, n_packed,
                       step_count);
    HIP_TRY(hipGetLastError());
    return CF_OK;
}

// ---- profiling ---------------------------------------------------------------------------
static int prof_collect(cf_model* m) {
    for (auto& ev : m->ev_pending) {
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        m->prof_ms[ev.slot] += ms;
        m->prof_n[ev.slot] += 1;
        m->ev_pool.push_back(ev.a);
        m->ev_pool.push_back(ev.b);
    }
    m->ev_pending.clear();
    return CF_OK;
}
extern "C" int cf_profile_enable(cf_model* m, int on) {
    if (!m) return fail(CF_ERR_INVALID, "null model");
    if (!on && m->prof_every) { int rc = prof_collect(m); if (rc != CF_OK) return rc; }
    m->prof_every = on < 0 ? 0 : on;
    m->prof = on != 0;
    m->prof_calls = 0;
    return CF_OK;
}
extern "C" int cf_profile_reset(cf_model* m) {
    if (!m) return fail(CF_ERR_INVALID, "null model");
    int rc = prof_collect(m);
    if (rc != CF_OK) return rc;
    for (int i = 0; i < CF_PROF_SLOTS; ++i) { m->prof_ms[i] = 0; m->prof_n[i] = 0; }
    return CF_OK;
}
extern "C" int cf_profile_read(cf_model* m, double ms[CF_PROF_SLOTS], int64_t launches[CF_PROF_SLOTS]) {
    if (!m || !ms || !launches) return fail(CF_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = prof_collect(m);
    if (rc != CF_OK) return rc;
    for (int i = 0; i < CF_PROF_SLOTS; ++i) { ms[i] = m->prof_ms[i]; launches[i] = m->prof_n[i]; }
    return CF_OK;
}
extern "C" const char* cf_profile_slot_name(int slot) {
    return (slot >= 0 && slot < CF_PROF_SLOTS) ? k_slot_names[slot] : "";
}

// ---- debug hook --------------------------------------------------------------------------
extern "C" int cf_debug_stage(cf_model* m, int stage, int64_t n_windows, float* out_host) {
    if (!m || !out_host) return fail(CF_ERR_INVALID, "cf_debug_stage: null argument");
    if (m->gen) return fail(CF_ERR_INVALID, "cf_debug_stage: not available on the any-size path");
    if (m->np > 0 && stage != 100) return fail(CF_ERR_INVALID, "cf_debug_stage: only available with CF_PREC_FP32");
    const cf_model::Slot& sl = m->slots[0];
    if (n_windows <= 0 || (stage != 100 && n_windows > sl.last_windows)) return fail(CF_ERR_INVALID, "cf_debug_stage: n_windows exceeds slot 0's last pass");
    HIP_TRY(hipSetDevice(m->device));
    if (stage == 100) {      // raw dense-partial buffer of slot 0 (diagnostic builds park their in-kernel stamps there)
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out_host, sl.d_p, (size_t)n_windows * sizeof(float), hipMemcpyDeviceToHost));
        return CF_OK;
    }
    int feats, mt;
    const float* src;
    if (stage >= 0 && stage < m->hp.n_layers_res) {
        // ping-pong: only the last two blocks survive a pass
        if (stage < m->hp.n_layers_res - 2) return fail(CF_ERR_INVALID, "cf_debug_stage: stage overwritten");
        if (stage == 0 && sl.last_res_fused)
            return fail(CF_ERR_INVALID, "cf_debug_stage: block 0's output is not materialised by the fused two-block launch (throughput mode)");
        feats = CF_C; mt = 2; src = sl.d_a[stage & 1];
    } else if (stage >= m->hp.n_layers_res && stage < m->hp.n_layers_res + m->hp.n_layers - 1) {
        const int l = stage - m->hp.n_layers_res;
        if (l < m->hp.n_layers - 3) return fail(CF_ERR_INVALID, "cf_debug_stage: stage overwritten");
        feats = 2 * CF_H; mt = 8; src = sl.d_y[l & 1];
    } else {
        return fail(CF_ERR_INVALID, "cf_debug_stage: no such stage");
    }
    const int64_t n_tiles = (n_windows + CF_TILE - 1) / CF_TILE;
    std::vector<float> frag((size_t)n_tiles * CF_T * mt * 256);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(frag.data(), src, frag.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t w = 0; w < n_windows; ++w)
        for (int t = 0; t < CF_T; ++t)
            for (int f = 0; f < feats; ++f) {
                const int mo = f >> 4, qq = (f >> 2) & 3, r = f & 3;
                const int lane = qq * 16 + (int)(w & 15);
                out_host[(w * CF_T + t) * feats + f] = frag[((((w >> 4) * CF_T + t) * mt + mo) * 64 + lane) * 4 + r];
            }
    return CF_OK;
}

extern "C" int64_t cf_workspace_bytes(const cf_model* m) { return m ? m->ws_bytes : 0; }
extern "C" const char* cf_last_error(void) { return g_err.c_str(); }
#include "chunks_host.hpp"
#include "loader_host.hpp"
#include "split_host.hpp"

// Which card is HIP device `device` of this process (rank placement and the N > 1 bench line: catfish_amd/placement.py)
extern "C" int cf_device_identity(int device, char* pci_bus_id, int64_t bus_cap, char* uuid_hex, int64_t uuid_cap) {
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(CF_ERR_INVALID, "cf_device_identity: no such device");
    if (pci_bus_id) {
        if (bus_cap < 13) return fail(CF_ERR_INVALID, "cf_device_identity: pci_bus_id needs 13 bytes");
        HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, (int)std::min<int64_t>(bus_cap, 64), device));
    }
    if (uuid_hex) {
        if (uuid_cap < 33) return fail(CF_ERR_INVALID, "cf_device_identity: uuid_hex needs 33 bytes");
        hipUUID id;
        HIP_TRY(hipDeviceGetUuid(&id, device));
        static const char* hex = "0123456789abcdef";
        for (int i = 0; i < 16; ++i) {
            uuid_hex[2 * i] = hex[((unsigned char)id.bytes[i]) >> 4];
            uuid_hex[2 * i + 1] = hex[((unsigned char)id.bytes[i]) & 15];
        }
        uuid_hex[32] = 0;
    }
    return CF_OK;
}

extern "C" int cf_abi_version(void) { return CF_ABI_VERSION; }
extern "C" const char* cf_version(void) { return "catfish_hip 0.3 (gfx950; fp32 MFMA 16x16x4, bf16 / bf16x3 MFMA 32x32x16)"; }
