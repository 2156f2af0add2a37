// Training kernels of the residual conv stack (resnet_class.py:44-82 forward; its part of
// optimizer.minimize(loss), rnn_class.py:62-71, backward).  Per block d (units u = 4d .. 4d+3, kernel widths 1,1,3,1):
//
//   sc  = BN_u0(conv1(a))            o1 = relu(BN_u1(conv1(a)))
//   o2  = relu(BN_u2(conv3(o1)))     o3 = relu(BN_u3(conv1(o2)))    out = relu(o3 + sc)
//
// with batch norm in inference mode (moving statistics are constants; gamma and beta train) and window-local
// SAME padding.  The work is tiny next to the biGRU (5 % of the FLOPs) and every sum is local to a window, so
// this is plain VALU code: a workgroup owns RT_WIN windows, keeps their activations [position][channel] in LDS,
// thread (channel = tid & 31, group = tid >> 5) walks the positions group, group + 8, ...  The forward stashes
// the pre-BN conv outputs z_u; the backward recomputes every activation from them, and writes per-workgroup
// partial parameter gradients that res_train_reduce_kernel adds in a fixed order.
//
// Parameter buffer (and gradient buffer, same offsets): per unit  W [k][cin][32] | b[32] | gamma[32] | beta[32] |
// moving_mean[32] | moving_variance[32]  (TensorFlow layouts), units back to back; cin = 1 for units 0 and 1.
#pragma once

#define RT_WIN 2                          // windows per workgroup (LDS is sized for it); small batches run 1 per workgroup
#define RT_POS (RT_WIN * CF_T)
#define RT_SMALL_BATCH 1024               // up to this many windows: one window per workgroup (twice the workgroups, half the chain)
static inline int rt_win_for(int64_t n_windows) { return n_windows <= RT_SMALL_BATCH ? 1 : RT_WIN; }
#define RT_THREADS 256
#define RT_PG (RT_THREADS / 32)
#define RT_MAX_UNITS 16
#define RT_FWD_LDS_BYTES (4 * RT_POS * 32 * 4)                          // block input, o1, o2, shortcut
#define RT_BWD_LDS_BYTES ((5 * RT_POS * 32 + 3 * 32 * 32 + 96) * 4)     // + gradient buffers and the reduction scratch

struct rt_layout { int off[RT_MAX_UNITS + 1]; };

static inline int rt_unit_k(int u) { return (u & 3) == 2 ? 3 : 1; }
static inline int rt_unit_cin(int u) { return u < 2 ? 1 : CF_C; }
static inline int rt_unit_floats(int u) { return rt_unit_k(u) * rt_unit_cin(u) * CF_C + 5 * CF_C; }
static inline rt_layout rt_make_layout(int n_blocks) {
    rt_layout L;
    int o = 0;
    for (int u = 0; u <= RT_MAX_UNITS; ++u) {
        L.off[u] = o;
        if (u < 4 * n_blocks) o += rt_unit_floats(u);
    }
    return L;
}

struct rt_bn { float s, t, mean, inv; };      // o = z * s + t;  xhat = (z - mean) * inv
__device__ __forceinline__ rt_bn rt_load_bn(const float* unit, int wfloats, int co, float eps) {
    const float gamma = unit[wfloats + 32 + co], beta = unit[wfloats + 64 + co];
    const float mean = unit[wfloats + 96 + co], var = unit[wfloats + 128 + co];
    rt_bn b;
    b.inv = rsqrtf(var + eps);
    b.s = gamma * b.inv;
    b.t = beta - mean * b.s;
    b.mean = mean;
    return b;
}

// z[p][co] = b[co] + sum_{kk, ci} IN[p + kk - K/2][ci] * W[kk][ci][co], window-local zero padding
template <int K, int CIN>
__device__ __forceinline__ float rt_conv_at(const float* IN, const float (&wr)[K * CIN], float bias, int p) {
    float acc = bias;
    const int t = p % CF_T;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const int tt = t + kk - K / 2;
        if (tt < 0 || tt >= CF_T) continue;
        const float* row = IN + (p + kk - K / 2) * 32;
        if (CIN == 1) {
            acc += row[0] * wr[kk];
        } else {
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c4);
                acc += v[0] * wr[kk * CIN + 4 * c4] + v[1] * wr[kk * CIN + 4 * c4 + 1] + v[2] * wr[kk * CIN + 4 * c4 + 2] +
                       v[3] * wr[kk * CIN + 4 * c4 + 3];
            }
        }
    }
    return acc;
}

template <int K, int CIN>
__device__ __forceinline__ void rt_load_wcol(const float* W, int co, float (&wr)[K * CIN]) {
#pragma unroll
    for (int i = 0; i < K * CIN; ++i) wr[i] = W[i * 32 + co];
}

// one conv+BN unit of the forward: writes z to the stash and returns through `emit(p, bn output)`
template <int K, int CIN, typename F>
__device__ __forceinline__ void rt_fwd_unit(const float* unit, const float* IN, float* Zu, int npos, int co, int pg, float eps, F emit) {
    float wr[K * CIN];
    rt_load_wcol<K, CIN>(unit, co, wr);
    const float bias = unit[K * CIN * 32 + co];
    const rt_bn bn = rt_load_bn(unit, K * CIN * 32, co, eps);
    for (int p = pg; p < npos; p += RT_PG) {
        const float z = rt_conv_at<K, CIN>(IN, wr, bias, p);
        Zu[(int64_t)p * 32 + co] = z;
        emit(p, z * bn.s + bn.t);
    }
}

__global__ __launch_bounds__(RT_THREADS) void res_train_fwd_kernel(const float* __restrict__ x,      // [N][35]
                                                                   const float* __restrict__ prm,    // parameter buffer
                                                                   float* __restrict__ Z,            // [units][N*35][32]
                                                                   float* __restrict__ out,          // [N*35][32]
                                                                   rt_layout L, int n_windows, int n_blocks, float eps, int win) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* BA = lds;
    float* B1 = BA + RT_POS * 32;
    float* B2 = B1 + RT_POS * 32;
    float* BS = B2 + RT_POS * 32;
    const int co = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int w0 = blockIdx.x * win;
    const int npos = min(win, n_windows - w0) * CF_T;
    const int64_t g0 = (int64_t)w0 * CF_T;
    const int64_t NP = (int64_t)n_windows * CF_T;
    for (int p = threadIdx.x; p < npos; p += RT_THREADS) BA[p * 32] = x[g0 + p];
    __syncthreads();
    for (int d = 0; d < n_blocks; ++d) {
        const float* u0 = prm + L.off[4 * d];
        const float* u1 = prm + L.off[4 * d + 1];
        const float* u2 = prm + L.off[4 * d + 2];
        const float* u3 = prm + L.off[4 * d + 3];
        float* Z0 = Z + ((int64_t)(4 * d) * NP + g0) * 32;
        float* Z1 = Z0 + NP * 32;
        float* Z2 = Z1 + NP * 32;
        float* Z3 = Z2 + NP * 32;
        if (d == 0) {
            rt_fwd_unit<1, 1>(u0, BA, Z0, npos, co, pg, eps, [&](int p, float o) { BS[p * 32 + co] = o; });
            rt_fwd_unit<1, 1>(u1, BA, Z1, npos, co, pg, eps, [&](int p, float o) { B1[p * 32 + co] = fmaxf(o, 0.f); });
        } else {
            rt_fwd_unit<1, 32>(u0, BA, Z0, npos, co, pg, eps, [&](int p, float o) { BS[p * 32 + co] = o; });
            rt_fwd_unit<1, 32>(u1, BA, Z1, npos, co, pg, eps, [&](int p, float o) { B1[p * 32 + co] = fmaxf(o, 0.f); });
        }
        __syncthreads();
        rt_fwd_unit<3, 32>(u2, B1, Z2, npos, co, pg, eps, [&](int p, float o) { B2[p * 32 + co] = fmaxf(o, 0.f); });
        __syncthreads();
        rt_fwd_unit<1, 32>(u3, B2, Z3, npos, co, pg, eps, [&](int p, float o) { BA[p * 32 + co] = fmaxf(fmaxf(o, 0.f) + BS[p * 32 + co], 0.f); });
        __syncthreads();
    }
    for (int p = pg; p < npos; p += RT_PG) out[(g0 + p) * 32 + co] = BA[p * 32 + co];
}

// ---- backward ---------------------------------------------------------------------------------
// Output-channel role of one unit: DO[p][co] is the gradient w.r.t. the unit's BN output (RELU_OUT: w.r.t. the relu of it).  Writes dz = DO * s to DZ,
// accumulates dW (against IN), db, dgamma, dbeta over this thread's positions, reduces over the RT_PG position groups
// in a fixed order through D and stores the partial sums at Pu (this workgroup's slot of the gradient buffer).
template <int K, int CIN, bool RELU_OUT = false>
__device__ __forceinline__ void rt_bwd_unit_co(const float* unit, const float* IN, const float* DO, const float* Zu, float* DZ, float* D,
                                               float* Pu, int npos, int co, int pg, float eps) {
    constexpr int WF = K * CIN * 32;
    const rt_bn bn = rt_load_bn(unit, WF, co, eps);
    float acc[K * CIN];
#pragma unroll
    for (int i = 0; i < K * CIN; ++i) acc[i] = 0.f;
    float sb = 0.f, sg = 0.f;
    for (int p = pg; p < npos; p += RT_PG) {
        const float z = Zu[(int64_t)p * 32 + co];
        float go = DO[p * 32 + co];
        if (RELU_OUT && !(z * bn.s + bn.t > 0.f)) go = 0.f;
        sb += go;
        sg += go * (z - bn.mean) * bn.inv;
        const float dz = go * bn.s;
        DZ[p * 32 + co] = dz;
        const int t = p % CF_T;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int tt = t + kk - K / 2;
            if (tt < 0 || tt >= CF_T) continue;
            const float* row = IN + (p + kk - K / 2) * 32;
            if (CIN == 1) {
                acc[kk] += row[0] * dz;
            } else {
#pragma unroll
                for (int c4 = 0; c4 < CIN / 4; ++c4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[kk * CIN + 4 * c4 + j] += v[j] * dz;
                }
            }
        }
    }
    // D: dW [K*CIN][32] | db [32] | dgamma [32] | dbeta [32]
    for (int g = 0; g < RT_PG; ++g) {
        if (pg == g) {
#pragma unroll
            for (int i = 0; i < K * CIN; ++i) D[i * 32 + co] = (g ? D[i * 32 + co] : 0.f) + acc[i];
            D[WF + co] = (g ? D[WF + co] : 0.f) + sb * bn.s;
            D[WF + 32 + co] = (g ? D[WF + 32 + co] : 0.f) + sg;
            D[WF + 64 + co] = (g ? D[WF + 64 + co] : 0.f) + sb;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < WF + 160; i += RT_THREADS) Pu[i] = i < WF + 96 ? D[i] : 0.f;      // moving statistics do not train
    __syncthreads();
}

// Input-channel role: returns sum_{kk, co} W[kk][ci][co] * DZ[p - kk + K/2][co] for this thread's channel ci.
template <int K>
__device__ __forceinline__ float rt_bwd_din_at(const float* DZ, const float (&wrow)[K * 32], int p) {
    float acc = 0.f;
    const int t = p % CF_T;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const int tt = t - kk + K / 2;
        if (tt < 0 || tt >= CF_T) continue;
        const float* row = DZ + (p - kk + K / 2) * 32;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c4);
            acc += v[0] * wrow[kk * 32 + 4 * c4] + v[1] * wrow[kk * 32 + 4 * c4 + 1] + v[2] * wrow[kk * 32 + 4 * c4 + 2] +
                   v[3] * wrow[kk * 32 + 4 * c4 + 3];
        }
    }
    return acc;
}

template <int K>
__device__ __forceinline__ void rt_load_wrow(const float* W, int ci, float (&wrow)[K * 32]) {     // W [K][32][32], row ci of every tap
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
#pragma unroll
        for (int c = 0; c < 32; ++c) wrow[kk * 32 + c] = W[(kk * 32 + ci) * 32 + c];
}

__global__ __launch_bounds__(RT_THREADS) void res_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ prm,
                                                                   const float* __restrict__ Z, const float* __restrict__ dout,
                                                                   float* __restrict__ P,      // [workgroup][grad floats]
                                                                   rt_layout L, int n_windows, int n_blocks, float eps, int win) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* BA = lds;                       // block input
    float* O1 = BA + RT_POS * 32;          // relu(BN(conv1(a)))      -> later d/d(its BN output)
    float* O2 = O1 + RT_POS * 32;          // relu(BN(conv3(o1)))     -> later d/d(its BN output), then a partial of d/d(a)
    float* GB = O2 + RT_POS * 32;          // gradient w.r.t. the block output -> w.r.t. o3 + sc -> w.r.t. the block input
    float* DZ = GB + RT_POS * 32;
    float* D = DZ + RT_POS * 32;           // 3 * 32 * 32 + 96 reduction scratch
    const int ch = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int w0 = blockIdx.x * win;
    const int npos = min(win, n_windows - w0) * CF_T;
    const int64_t g0 = (int64_t)w0 * CF_T;
    const int64_t NP = (int64_t)n_windows * CF_T;
    float* Pwg = P + (size_t)blockIdx.x * L.off[4 * n_blocks];
    for (int p = pg; p < npos; p += RT_PG) GB[p * 32 + ch] = dout[(g0 + p) * 32 + ch];
    for (int d = n_blocks - 1; d >= 0; --d) {
        const float* u0 = prm + L.off[4 * d];
        const float* u1 = prm + L.off[4 * d + 1];
        const float* u2 = prm + L.off[4 * d + 2];
        const float* u3 = prm + L.off[4 * d + 3];
        const int wf01 = d == 0 ? 32 : 32 * 32;                       // floats of W in units 0 and 1 of this block
        const float* Z0 = Z + ((int64_t)(4 * d) * NP + g0) * 32;
        const float* Z1 = Z0 + NP * 32;
        const float* Z2 = Z1 + NP * 32;
        const float* Z3 = Z2 + NP * 32;
        // ---- recompute the block's activations from the stashed conv outputs
        if (d == 0) {
            for (int p = threadIdx.x; p < npos; p += RT_THREADS) BA[p * 32] = x[g0 + p];
        } else {
            const float* pu0 = prm + L.off[4 * d - 4];
            const float* pu3 = prm + L.off[4 * d - 1];
            const rt_bn b0 = rt_load_bn(pu0, d == 1 ? 32 : 32 * 32, ch, eps);
            const rt_bn b3 = rt_load_bn(pu3, 32 * 32, ch, eps);
            const float* PZ0 = Z + ((int64_t)(4 * d - 4) * NP + g0) * 32;
            const float* PZ3 = PZ0 + 3 * NP * 32;
            for (int p = pg; p < npos; p += RT_PG)
                BA[p * 32 + ch] = fmaxf(fmaxf(PZ3[(int64_t)p * 32 + ch] * b3.s + b3.t, 0.f) + PZ0[(int64_t)p * 32 + ch] * b0.s + b0.t, 0.f);
        }
        {
            const rt_bn b0 = rt_load_bn(u0, wf01, ch, eps), b1 = rt_load_bn(u1, wf01, ch, eps);
            const rt_bn b2 = rt_load_bn(u2, 3 * 32 * 32, ch, eps), b3 = rt_load_bn(u3, 32 * 32, ch, eps);
            for (int p = pg; p < npos; p += RT_PG) {
                const int64_t q = (int64_t)p * 32 + ch;
                O1[p * 32 + ch] = fmaxf(Z1[q] * b1.s + b1.t, 0.f);
                O2[p * 32 + ch] = fmaxf(Z2[q] * b2.s + b2.t, 0.f);
                const float pre = fmaxf(Z3[q] * b3.s + b3.t, 0.f) + Z0[q] * b0.s + b0.t;
                if (!(pre > 0.f)) GB[p * 32 + ch] = 0.f;              // relu(o3 + sc)
            }
        }
        __syncthreads();
        // ---- unit 3: conv1 on o2
        rt_bwd_unit_co<1, 32, true>(u3, O2, GB, Z3, DZ, D, Pwg + L.off[4 * d + 3], npos, ch, pg, eps);
        {
            float wrow[32];
            rt_load_wrow<1>(u3, ch, wrow);
            for (int p = pg; p < npos; p += RT_PG) {
                const float g = rt_bwd_din_at<1>(DZ, wrow, p);
                O2[p * 32 + ch] = O2[p * 32 + ch] > 0.f ? g : 0.f;    // through relu: d/d(BN_u2 output)
            }
        }
        __syncthreads();
        // ---- unit 2: conv3 on o1
        rt_bwd_unit_co<3, 32>(u2, O1, O2, Z2, DZ, D, Pwg + L.off[4 * d + 2], npos, ch, pg, eps);
        {
            float wrow[96];
            rt_load_wrow<3>(u2, ch, wrow);
            for (int p = pg; p < npos; p += RT_PG) {
                const float g = rt_bwd_din_at<3>(DZ, wrow, p);
                O1[p * 32 + ch] = O1[p * 32 + ch] > 0.f ? g : 0.f;    // d/d(BN_u1 output)
            }
        }
        __syncthreads();
        // ---- unit 1: conv1 on the block input
        if (d == 0) rt_bwd_unit_co<1, 1>(u1, BA, O1, Z1, DZ, D, Pwg + L.off[4 * d + 1], npos, ch, pg, eps);
        else rt_bwd_unit_co<1, 32>(u1, BA, O1, Z1, DZ, D, Pwg + L.off[4 * d + 1], npos, ch, pg, eps);
        if (d > 0) {
            float wrow[32];
            rt_load_wrow<1>(u1, ch, wrow);
            for (int p = pg; p < npos; p += RT_PG) O2[p * 32 + ch] = rt_bwd_din_at<1>(DZ, wrow, p);     // first part of d/d(a)
        }
        __syncthreads();
        // ---- unit 0 (shortcut): conv1 on the block input
        if (d == 0) rt_bwd_unit_co<1, 1>(u0, BA, GB, Z0, DZ, D, Pwg + L.off[4 * d], npos, ch, pg, eps);
        else rt_bwd_unit_co<1, 32>(u0, BA, GB, Z0, DZ, D, Pwg + L.off[4 * d], npos, ch, pg, eps);
        if (d > 0) {
            float wrow[32];
            rt_load_wrow<1>(u0, ch, wrow);
            for (int p = pg; p < npos; p += RT_PG) GB[p * 32 + ch] = O2[p * 32 + ch] + rt_bwd_din_at<1>(DZ, wrow, p);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void res_train_reduce_kernel(const float* __restrict__ P, float* __restrict__ grads, int n_floats, int n_wg) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = cf_reduce_parts(P, (size_t)n_floats, n_wg, i, i < n_floats);      // fixed summation order (gru_wgrad.hpp)
    if (i < n_floats && threadIdx.x < 64) grads[i] = s;
}
