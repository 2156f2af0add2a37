// gru_train.hpp -- training kernels of one bidirectional GRU layer (fp32 MFMA), included by catfish_hip.hip.
//
//   gru_train_fwd_kernel<CIN>  forward exactly as gru_layer_kernel, plus the r/u/c stash.
//   gru_train_bwd_kernel<CIN>  back-propagation through time of one direction per workgroup half-grid:
//       dh   = dh_carry + dy_t
//       dc   = dh (1-u),  du = dh (h_prev - c),  dh_carry = dh u
//       da_c = dc (1-c^2);   [dx | d(rh)] += [Wxc; Whc] da_c
//       dr   = d(rh) h_prev; dh_carry += d(rh) r
//       da_r = dr r(1-r),  da_u = du u(1-u);   [dx | dh_carry] += [Wxg; Whg] da_g
//   (reference graph: tf.contrib.rnn.GRUCell, rnn_class.py:146; its gradient is what TF's autodiff builds
//   for RNN.train_network, rnn_class.py:201-210.)  The same accumulator-as-operand trick as the forward
//   applies with the roles swapped: D[in feature][window] = W[in][out] * da[out][window], the pre-activation
//   gradients da sit in D-layout registers and are used as the B operand directly.
//   Weight gradients are NOT accumulated here (a 192x192 accumulator per wave does not fit): the kernel
//   stores da_g / da_c per step and the host forms dW = A^T dA with one library GEMM per matrix.
#pragma once

// Backward blob of one direction (floats): A fragments in use order
//   C part: [ks < 16][mi2 < MI/2][lane][2]   rows = input features (x rows then h rows), k = candidate outputs
//   G part: [ks < 32][mi2 < MI/2][lane][2]   k = gate outputs (r 0..63, u 64..127)
// MI = (CIN + 64) / 16 input M-tiles; unscaled weights (the stash holds activated values).
__host__ __device__ constexpr int gtb_mi(int cin) { return (cin + CF_H) / 16; }
__host__ __device__ constexpr int gtb_c_floats(int cin) { return 16 * (gtb_mi(cin) / 2) * 128; }
__host__ __device__ constexpr int gtb_g_floats(int cin) { return 32 * (gtb_mi(cin) / 2) * 128; }
__host__ __device__ constexpr int gtb_pack_floats(int cin) { return gtb_c_floats(cin) + gtb_g_floats(cin); }

template <int CIN>
__global__ __launch_bounds__(512, 2) void gru_train_fwd_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ X,
                                                               f32x4* __restrict__ Y, f32x4* __restrict__ S, int n_tiles,
                                                               f32x4* __restrict__ YD, cf_dropout drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dir = blockIdx.y;
    gru_stage_weights<CIN>(lds, wpack, dir);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    for (int tile = blockIdx.x * nwaves + wave; tile < n_tiles; tile += gridDim.x * nwaves)
        gru_tile<CIN, false, true>(lds, lane, dir, tile, X, Y, nullptr, n_tiles, S, YD, drop);
}

template <int CIN>
__global__ __launch_bounds__(512, 2) void gru_train_bwd_kernel(const float* __restrict__ wpack,   // [2][gtb_pack_floats]
                                                               const f32x4* __restrict__ Y,       // layer output  [tile][t][8][lane]
                                                               const f32x4* __restrict__ S,       // stash         [tile][t][2][12][lane]
                                                               const f32x4* __restrict__ DY,      // d loss / d Y  [tile][t][8][lane]
                                                               const f32x4* __restrict__ DY2,     // optional second addend of dY (same layout)
                                                               const f32x4* __restrict__ DSC,     // optional per-element scale of dY (dropout mask / keep_prob)
                                                               f32x4* __restrict__ DX,            // [2 dirs][tile][t][CIN/16][lane]
                                                               f32x4* __restrict__ DA,            // [tile][t][2][12][lane]: da_r, da_u (0..7), da_c (8..11)
                                                               int n_tiles, cf_dropout drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MI = gtb_mi(CIN);
    constexpr int MX = CIN / 16;            // x-row M-tiles; the 4 h-row M-tiles follow
    constexpr int PACK = gtb_pack_floats(CIN);
    constexpr int CF2 = gtb_c_floats(CIN) / 2;     // region size in f32x2 units
    const int dir = blockIdx.y;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const far_lds<f32x2> WT(reinterpret_cast<const f32x2*>(lds) + lane);   // candidate region + (ks*(MI/2) + mi2)*64 | gate region
    const bool drop_on = !DSC && drop.keep_prob < 1.f;      // the layer's output dropout, mask recomputed (no stored mask)
    const uint32_t drop_key = drop_on ? cf_drop_key(drop) : 0u;

    for (int tile = blockIdx.x * nwaves + wave; tile < n_tiles; tile += gridDim.x * nwaves) {
        f32x4 dhc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // gradient carried to the previous step
        for (int s = CF_T - 1; s >= 0; --s) {
            const int t = dir ? (CF_T - 1 - s) : s;                 // forward step s touched time t
            const int tp = dir ? (t + 1) : (t - 1);                 // time of the previous forward step
            const int64_t base = (int64_t)tile * CF_T + t;
            f32x4 r[4], u[4], c[4], hp[4], dh[4];
            {
                const f32x4* sp = S + (base * 2 + dir) * 12 * 64 + lane;
                const f32x4* dyp = DY + (base * 8 + dir * 4) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    r[m] = sp[m * 64]; u[m] = sp[(4 + m) * 64]; c[m] = sp[(8 + m) * 64];
                    f32x4 dy = dyp[m * 64];
                    if (DY2) dy += DY2[(base * 8 + dir * 4 + m) * 64 + lane];
                    if (DSC) dy *= DSC[(base * 8 + dir * 4 + m) * 64 + lane];
                    else if (drop_on) dy *= cf_drop_scale4(drop_key, drop.keep_prob, (base * 8 + dir * 4 + m) * 64 + lane);
                    dh[m] = dhc[m] + dy;
                }
                if (s > 0) {
                    const f32x4* hpp = Y + (((int64_t)tile * CF_T + tp) * 8 + dir * 4) * 64 + lane;
#pragma unroll
                    for (int m = 0; m < 4; ++m) hp[m] = hpp[m * 64];
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) hp[m] = (f32x4){0, 0, 0, 0};     // zero initial state
                }
            }
            f32x4 dac[4], du[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                du[m] = dh[m] * (hp[m] - c[m]);
                dac[m] = dh[m] * (one - u[m]) * (one - c[m] * c[m]);
                dhc[m] = dh[m] * u[m];
            }
            f32x4 dx[MX], drh[4], dhg[4];
#pragma unroll
            for (int i = 0; i < MX; ++i) dx[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) { drh[i] = (f32x4){0, 0, 0, 0}; dhg[i] = (f32x4){0, 0, 0, 0}; }
            // candidate part: [dx | d(rh)] += [Wxc; Whc] da_c
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float b = dac[ks >> 2][ks & 3];
#pragma unroll
                for (int mi2 = 0; mi2 < MI / 2; ++mi2) {
                    const f32x2 a = WT[(ks * (MI / 2) + mi2) * 64];
                    const int m0 = 2 * mi2, m1 = 2 * mi2 + 1;
                    if (m0 < MX) dx[m0] = MFMA16(a.x, b, dx[m0]); else drh[m0 - MX] = MFMA16(a.x, b, drh[m0 - MX]);
                    if (m1 < MX) dx[m1] = MFMA16(a.y, b, dx[m1]); else drh[m1 - MX] = MFMA16(a.y, b, drh[m1 - MX]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 dag[8];   // da_r (0..3), da_u (4..7)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                dag[m] = drh[m] * hp[m] * r[m] * (one - r[m]);
                dag[4 + m] = du[m] * u[m] * (one - u[m]);
                dhc[m] += drh[m] * r[m];
            }
            // gate part: [dx | dh] += [Wxg; Whg] da_g
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const float b = dag[ks >> 2][ks & 3];
#pragma unroll
                for (int mi2 = 0; mi2 < MI / 2; ++mi2) {
                    const f32x2 a = WT[CF2 + (ks * (MI / 2) + mi2) * 64];
                    const int m0 = 2 * mi2, m1 = 2 * mi2 + 1;
                    if (m0 < MX) dx[m0] = MFMA16(a.x, b, dx[m0]); else dhg[m0 - MX] = MFMA16(a.x, b, dhg[m0 - MX]);
                    if (m1 < MX) dx[m1] = MFMA16(a.y, b, dx[m1]); else dhg[m1 - MX] = MFMA16(a.y, b, dhg[m1 - MX]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) dhc[m] += dhg[m];
            {
                f32x4* dxp = DX + (((int64_t)dir * n_tiles * CF_T + base) * MX) * 64 + lane;
#pragma unroll
                for (int i = 0; i < MX; ++i) dxp[i * 64] = dx[i];
                f32x4* dap = DA + (base * 2 + dir) * 12 * 64 + lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) dap[j * 64] = dag[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) dap[(8 + j) * 64] = dac[j];
            }
        }
    }
}

// host packer: rows = input features of [Wg | Wc] (x rows 0..cin-1, then h rows), k = output features
static void pack_gru_dir_bwd(const cf_gru_dir& g, int cin, float* out) {
    const int MI = gtb_mi(cin);
    auto row_of = [&](int mi, int lane) { return 16 * mi + (lane & 15); };    // natural order: x rows then h rows
    float* pc = out;
    for (int ks = 0; ks < 16; ++ks)
        for (int mi2 = 0; mi2 < MI / 2; ++mi2)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 2; ++j)
                    pc[((ks * (MI / 2) + mi2) * 64 + lane) * 2 + j] =
                        g.candidate_kernel[(size_t)row_of(2 * mi2 + j, lane) * CF_H + frag_feature(ks, lane >> 4)];
    float* pg = out + gtb_c_floats(cin);
    for (int ks = 0; ks < 32; ++ks)
        for (int mi2 = 0; mi2 < MI / 2; ++mi2)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 2; ++j)
                    pg[((ks * (MI / 2) + mi2) * 64 + lane) * 2 + j] =
                        g.gates_kernel[(size_t)row_of(2 * mi2 + j, lane) * 2 * CF_H + frag_feature(ks, lane >> 4)];
}
