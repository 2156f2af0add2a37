// gru_coop.hpp -- latency mode of the fp32 biGRU layer: FOUR waves (one per SIMD) cooperate on ONE tile of 16
// windows.  Included by catfish_hip.hip.
//
// A single read is 8 tiles x 2 directions; in the throughput kernel a tile is one wave and the 35-step chain
// costs 576 MFMAs x 32 cycles per step.  Here wave w owns output M-tiles {r[w], u[w], c[w]} (16 hidden units),
// i.e. 144 MFMAs per step, and the waves exchange r*h and h' through 8 KiB of LDS with two s_barriers per step.
// Same packed weights, same arithmetic per element (the k order of every accumulation is unchanged), so the
// results are bit-identical to gru_layer_kernel.
#pragma once

template <int W>
__device__ __forceinline__ float pick4(const f32x4& a) { return a[W]; }
// workgroups per tile in gru_xproj_kernel / gru_dx_kernel: one step each while the call is tiny (parallelism), five steps
// each otherwise (the weight fragments are fetched from L2 once per workgroup)
static inline int cf_xproj_chunks(int n_tiles) { return n_tiles <= 8 ? CF_T : 7; }
#ifndef CF_COOP_PF
#define CF_COOP_PF 4        // A-fragment prefetch depth (k-steps) of the cooperative forward kernel
#endif
// Diagnostic build: s_memtime cycles of the four segments of a step (h read + r/u MFMAs | r, r*h, barrier | c MFMAs | u, c, h', barrier),
// summed over the 35 steps by wave 0 and written as int64 [dir][tile][8] to the dense-partial buffer of the non-LAST kernels
#ifndef CF_COOP_STAMP
#define CF_COOP_STAMP 0
#endif

// HOIST: the x projection (bias + Wx^T x_t, all 35 steps) was computed by gru_xproj_kernel into XP; the step then
// starts from those accumulators -- the same fp32 values the in-kernel x part produces, so results are unchanged.
template <int CIN, bool LAST, int W, bool STASH = false, bool HOIST = false>
__device__ __forceinline__ void gru_tile_coop(float* lds, float* xch, int lane, int dir, int tile, const f32x4* __restrict__ X,
                                              f32x4* __restrict__ Y, float* __restrict__ P, int n_tiles,
                                              f32x4* __restrict__ S = nullptr, const f32x4* __restrict__ XP = nullptr,
                                              f32x4* __restrict__ YD = nullptr, cf_dropout drop = cf_dropout()) {
    const uint32_t drop_key = (STASH && YD) ? cf_drop_key(drop) : 0u;
    constexpr int KGX = CIN / 16;
    constexpr int KSX = HOIST ? 0 : CIN / 4;          // k-steps of the x part done in this kernel
    constexpr int XOFF = CIN / 4;                     // k-steps the packed x region holds
    constexpr int XN4 = gru_x_floats(CIN) / 4;
    constexpr int HG4 = gru_hg_floats() / 4;
    constexpr int BIAS = gru_bias_off(CIN);
    constexpr int DENSE = gru_dense_off(CIN);
    (void)XOFF;
    const int q = lane >> 4;
    const f32x4* WX = reinterpret_cast<const f32x4*>(lds) + lane;
    const f32x4* WG = WX + XN4;
    const f32x4* WC = WG + HG4;       // (far_lds measured 1 % slower here: the step is latency-, not issue-bound)
    (void)WC;
    const f32x4* B4 = reinterpret_cast<const f32x4*>(lds + BIAS) + q;
    f32x4* hx = reinterpret_cast<f32x4*>(xch) + lane;              // [4 M-tiles][64 lanes] f32x4: the state h
    f32x4* rx = hx + 4 * 64;                                        // r * h

    hx[W * 64] = (f32x4){0, 0, 0, 0};                               // GRUCellZeroState (every wave zeroes its own tile)
    __syncthreads();
    f32x4 hown = {0, 0, 0, 0};
    f32x4 xc[KGX];
    f32x4 xp[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};     // HOIST: r, u, c accumulators of the coming step
    auto load_xp = [&](int t) {
        const f32x4* src = XP + (((int64_t)tile * CF_T + t) * 2 + dir) * 12 * 64 + lane;
        xp[0] = src[(0 + W) * 64]; xp[1] = src[(4 + W) * 64]; xp[2] = src[(8 + W) * 64];
    };
    if constexpr (HOIST) {
        load_xp(dir ? (CF_T - 1) : 0);
    } else {
        const int t0 = dir ? (CF_T - 1) : 0;
        const f32x4* src = X + ((int64_t)tile * CF_T + t0) * KGX * 64 + lane;
#pragma unroll
        for (int g = 0; g < KGX; ++g) xc[g] = src[g * 64];
    }
    // A fragments: this wave needs only component W of each packed f32x4 (its own M-tile), read as one dword.  With a
    // single wave per SIMD nothing else hides the LDS latency, so the fragments of the whole step form ONE sequence
    // p = 0 .. KSX+31 (x part: 3 per k-step, gate h part: 2, candidate h part: 1) fetched CF_COOP_PF k-steps ahead
    // through a register ring; fetches of the next phase (and of the next step) run across the barriers.
    constexpr int PF = CF_COOP_PF;
    constexpr int NSEQ = KSX + 32;
    static_assert(NSEQ % PF == 0, "the ring slot of a fragment must not change across the step boundary");
    const float* WF = reinterpret_cast<const float*>(WX) + W;      // + element * 4 floats
    float ring[PF][3];
    auto fetch = [&](int p, float (&d)[3]) {                        // p is taken modulo the step's sequence
        p = p % NSEQ;
        if (p < KSX) { d[0] = WF[((p * 3 + 0) * 64) * 4]; d[1] = WF[((p * 3 + 1) * 64) * 4]; d[2] = WF[((p * 3 + 2) * 64) * 4]; }
        else if (p < KSX + 16) { d[0] = WF[(XN4 + ((p - KSX) * 2 + 0) * 64) * 4]; d[1] = WF[(XN4 + ((p - KSX) * 2 + 1) * 64) * 4]; }
        else { d[0] = WF[(XN4 + HG4 + (p - KSX - 16) * 64) * 4]; }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p, ring[p]);
    long long st_[4] = {0, 0, 0, 0};
    const long long st_begin = CF_COOP_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
    for (int s = 0; s < CF_T; ++s) {
        const int t = dir ? (CF_T - 1 - s) : s;
        long long ta_ = CF_COOP_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define CF_CSTAMP(idx) if (CF_COOP_STAMP) { const long long tb_ = (long long)__builtin_amdgcn_s_memtime(); st_[idx] += tb_ - ta_; ta_ = tb_; }
        f32x4 hf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) hf[m] = hx[m * 64];
        f32x4 ar, au, acnd;
        if constexpr (HOIST) { ar = xp[0]; au = xp[1]; acnd = xp[2]; }
        else { ar = B4[(0 + W) * 4]; au = B4[(4 + W) * 4]; acnd = B4[(8 + W) * 4]; }
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            const float a0 = ring[ks % PF][0], a1 = ring[ks % PF][1], a2 = ring[ks % PF][2];
            fetch(ks + PF, ring[ks % PF]);
            const float b = xc[ks >> 2][ks & 3];
            ar = MFMA16(a0, b, ar);
            au = MFMA16(a1, b, au);
            acnd = MFMA16(a2, b, acnd);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            int tn = dir ? (t - 1) : (t + 1);
            tn = tn < 0 ? 0 : (tn > CF_T - 1 ? CF_T - 1 : tn);
            if constexpr (HOIST) {
                load_xp(tn);
            } else {
                const f32x4* src = X + ((int64_t)tile * CF_T + tn) * KGX * 64 + lane;
#pragma unroll
                for (int g = 0; g < KGX; ++g) xc[g] = src[g * 64];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int p = KSX + ks;
            const float a0 = ring[p % PF][0], a1 = ring[p % PF][1];
            fetch(p + PF, ring[p % PF]);
            const float b = hf[ks >> 2][ks & 3];
            ar = MFMA16(a0, b, ar);
            au = MFMA16(a1, b, au);
            __builtin_amdgcn_sched_barrier(0);
        }
        CF_CSTAMP(0);
        f32x4 rh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ar[r] = cf_sigmoid_pre(ar[r]); rh[r] = ar[r] * hown[r]; }
        rx[W * 64] = rh;
        __syncthreads();
        CF_CSTAMP(1);
        f32x4 rf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) rf[m] = rx[m * 64];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int p = KSX + 16 + ks;
            const float a0 = ring[p % PF][0];
            fetch(p + PF, ring[p % PF]);                            // wraps into the next step's x part
            const float b = rf[ks >> 2][ks & 3];
            acnd = MFMA16(a0, b, acnd);
            __builtin_amdgcn_sched_barrier(0);
        }
        CF_CSTAMP(2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u = cf_sigmoid_pre(au[r]);
            const float c = cf_tanh_pre(acnd[r]);
            hown[r] = fmaf(u, hown[r] - c, c);
            if constexpr (STASH) { au[r] = u; acnd[r] = c; }
        }
        if constexpr (STASH) {      // training forward: activated r, u, c of this wave's 16 hidden units
            f32x4* sdst = S + (((int64_t)tile * CF_T + t) * 2 + dir) * 12 * 64 + lane;
            sdst[(0 + W) * 64] = ar; sdst[(4 + W) * 64] = au; sdst[(8 + W) * 64] = acnd;
        }
        hx[W * 64] = hown;
        if constexpr (!LAST) {
            Y[(((int64_t)tile * CF_T + t) * 8 + dir * 4 + W) * 64 + lane] = hown;
            if constexpr (STASH) {
                if (YD) {           // training with dropout: the copy the next layer (or the dense head) reads
                    const int64_t i0 = (((int64_t)tile * CF_T + t) * 8 + dir * 4 + W) * 64 + lane;
                    YD[i0] = hown * cf_drop_scale4(drop_key, drop.keep_prob, i0);
                }
            }
        } else {
            // partial logit of this wave's 16 hidden units, per lane, straight to global memory: head_kernel adds the four
            // M-tile sums and the lane quarters in the one-wave kernel's order (bit-identical), so the recurrence carries
            // neither an LDS round trip nor a serial reduction by one wave
            const f32x4 wd = *(reinterpret_cast<const f32x4*>(lds + DENSE) + q + W * 4);
            float p = wd.x * hown.x;
            p = fmaf(wd.y, hown.y, p); p = fmaf(wd.z, hown.z, p); p = fmaf(wd.w, hown.w, p);
            if (!CF_COOP_STAMP) P[((((int64_t)dir * n_tiles + tile) * CF_T + t) * 4 + W) * 64 + lane] = p;
        }
        __syncthreads();
        CF_CSTAMP(3);
#undef CF_CSTAMP
    }
    if (CF_COOP_STAMP && !LAST && W == 0 && lane == 0 && P) {
        long long* o = reinterpret_cast<long long*>(P) + ((int64_t)dir * n_tiles + tile) * 8;
        o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3];
        o[4] = (long long)__builtin_amdgcn_s_memtime() - st_begin;
    }
}

// The x projection of a whole layer for small calls: acc = bias + Wx^T x_t for every (tile, t, direction), no serial
// dependency, so it spreads over the CUs the recurrence leaves idle (one workgroup per tile and chunk of steps,
// wave W = M-tiles {r[W], u[W], c[W]}).  A fragments come straight from the packed weights in global memory (L2-resident, 24 KiB per
// wave, all loads issued up front); k order and bias-first accumulation are those of the in-kernel x part.
template <int CIN>
__global__ __launch_bounds__(256) void gru_xproj_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ X,
                                                        f32x4* __restrict__ XP, int n_tiles, int chunks) {
    constexpr int KGX = CIN / 16;
    constexpr int KSX = CIN / 4;
    constexpr int PACK = gru_pack_floats(CIN);
    constexpr int BIAS = gru_bias_off(CIN);
    const int dir = blockIdx.y;
    const int tile = blockIdx.x / chunks, chunk = blockIdx.x - tile * chunks;
    const int lane = threadIdx.x & 63;
    const int W = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* wp = wpack + (size_t)dir * PACK;
    const float* wf = wp + lane * 4 + W;
    float a[KSX][3];                                   // this wave's fragments stay in registers for the whole chunk of steps
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g) a[ks][g] = wf[((ks * 3 + g) * 64) * 4];
    const f32x4* B4 = reinterpret_cast<const f32x4*>(wp + BIAS) + (lane >> 4);
    const f32x4 br = B4[(0 + W) * 4], bu = B4[(4 + W) * 4], bc = B4[(8 + W) * 4];
    const int TL = (CF_T + chunks - 1) / chunks;
    for (int t = chunk * TL; t < min((chunk + 1) * TL, CF_T); ++t) {
        f32x4 xc[KGX];
        const f32x4* src = X + ((int64_t)tile * CF_T + t) * KGX * 64 + lane;
#pragma unroll
        for (int g = 0; g < KGX; ++g) xc[g] = src[g * 64];
        f32x4 ar = br, au = bu, ac = bc;
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            const float b = xc[ks >> 2][ks & 3];
            ar = MFMA16(a[ks][0], b, ar);
            au = MFMA16(a[ks][1], b, au);
            ac = MFMA16(a[ks][2], b, ac);
        }
        f32x4* dst = XP + (((int64_t)tile * CF_T + t) * 2 + dir) * 12 * 64 + lane;
        dst[(0 + W) * 64] = ar; dst[(4 + W) * 64] = au; dst[(8 + W) * 64] = ac;
    }
}

// The same x projection with the layer's x weights staged in LDS (round 4).  gru_xproj_kernel gives every wave its own copy of
// its fragments straight from L2 -- 96 dword loads per lane, 24 KiB per wave -- and, for a tiny call, one workgroup per (tile,
// step): a single read's 560 workgroups pull 54 MB through L2 and the launch lasts 23 us (Cin = 128), a quarter of the whole
// call, for 2.6 us of MFMA work.  Here a workgroup copies the 96 KiB x region once with 16-byte loads, takes a CHUNK of steps
// of one (tile, direction) and its four waves read their fragments (one dword each: wave W owns M-tiles {r[W], u[W], c[W]})
// from LDS through a register ring.  Bias first, k ascending: the in-kernel x part's order, hence its bits.  The launcher sizes
// the chunks so that the grid fills the chip once (cf_xproj_plan).
template <int CIN>
__global__ __launch_bounds__(256) void gru_xproj_lds_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ X,
                                                            f32x4* __restrict__ XP, int n_tiles, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KGX = CIN / 16;
    constexpr int KSX = CIN / 4;
    constexpr int PACK = gru_pack_floats(CIN);
    constexpr int BIAS = gru_bias_off(CIN);
    constexpr int XN4 = gru_x_floats(CIN) / 4;
    constexpr int PF = 4;                                  // fragment prefetch depth (k-steps)
    static_assert(KSX % PF == 0, "ring slots must not move across steps");
    const int dir = blockIdx.y;
    const int tile = blockIdx.x / chunks, chunk = blockIdx.x - tile * chunks;
    const int lane = threadIdx.x & 63;
    const int W = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* wp = wpack + (size_t)dir * PACK;
    const int TL = (CF_T + chunks - 1) / chunks;
    const int t_begin = chunk * TL, t_end = min(t_begin + TL, CF_T);
    if (t_begin >= t_end) return;                          // (whole workgroup: the chunk count need not divide 35)
    // this wave's inputs of the first step and its bias rows are on their way while the weights are staged
    f32x4 xc[KGX];
    {
        const f32x4* src = X + ((int64_t)tile * CF_T + t_begin) * KGX * 64 + lane;
#pragma unroll
        for (int g = 0; g < KGX; ++g) xc[g] = src[g * 64];
    }
    const f32x4* B4 = reinterpret_cast<const f32x4*>(wp + BIAS) + (lane >> 4);
    const f32x4 br = B4[(0 + W) * 4], bu = B4[(4 + W) * 4], bc = B4[(8 + W) * 4];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wp);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll 8
        for (int i = threadIdx.x; i < XN4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const float* WF = lds + lane * 4 + W;                  // + (k-step * 3 + gate) * 256 floats
    float ring[PF][3];
    auto fetch = [&](int ks, float (&d)[3]) {
        ks = ks % KSX;
        d[0] = WF[(ks * 3 + 0) * 256]; d[1] = WF[(ks * 3 + 1) * 256]; d[2] = WF[(ks * 3 + 2) * 256];
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p, ring[p]);
    for (int t = t_begin; t < t_end; ++t) {
        f32x4 ar = br, au = bu, ac = bc;
        f32x4 xn[KGX];
        {
            const int tn = t + 1 < t_end ? t + 1 : t;      // next step's inputs (last step: a harmless re-read)
            const f32x4* src = X + ((int64_t)tile * CF_T + tn) * KGX * 64 + lane;
#pragma unroll
            for (int g = 0; g < KGX; ++g) xn[g] = src[g * 64];
        }
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            const float a0 = ring[ks % PF][0], a1 = ring[ks % PF][1], a2 = ring[ks % PF][2];
            fetch(ks + PF, ring[ks % PF]);                  // wraps into the next step: same fragments again
            const float b = xc[ks >> 2][ks & 3];
            ar = MFMA16(a0, b, ar);
            au = MFMA16(a1, b, au);
            ac = MFMA16(a2, b, ac);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4* dst = XP + (((int64_t)tile * CF_T + t) * 2 + dir) * 12 * 64 + lane;
        dst[(0 + W) * 64] = ar; dst[(4 + W) * 64] = au; dst[(8 + W) * 64] = ac;
#pragma unroll
        for (int g = 0; g < KGX; ++g) xc[g] = xn[g];
    }
}

// chunks of steps per (tile, direction) for gru_xproj_lds_kernel: the count that minimises rounds x (staging + steps per chunk),
// in 0.01 us (staging the x region ~2.5 us at Cin = 128, a step = 96 MFMAs of 32 cycles per wave)
static inline int cf_xproj_plan(int n_tiles, int n_cu, int cin) {
    const int stage = cin >= 128 ? 250 : 80, step = cin >= 128 ? 128 : 32;
    const int slots = std::max(1, n_cu) * (cin >= 128 ? 1 : 4);            // resident workgroups: 96 KiB of LDS each at Cin = 128
    int best = 1, best_cost = 1 << 30;
    for (int c = 1; c <= CF_T; ++c) {
        const int tl = (CF_T + c - 1) / c;
        if ((CF_T + tl - 1) / tl != c) continue;                           // (only chunk counts that leave no empty chunk)
        const int rounds = (2 * n_tiles * c + slots - 1) / slots;
        const int cost = rounds * (stage + tl * step);
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

template <int CIN, bool LAST>
__global__ __launch_bounds__(256, 1) void gru_layer_coop_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ X,
                                                                f32x4* __restrict__ Y, float* __restrict__ P, int n_tiles,
                                                                const f32x4* __restrict__ XP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK = gru_pack_floats(CIN);
    const int dir = blockIdx.y;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll 8
        for (int i = threadIdx.x; i < PACK / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xch = lds + PACK;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (XP) {
            switch (wave) {
                case 0: gru_tile_coop<CIN, LAST, 0, false, true>(lds, xch, lane, dir, tile, X, Y, P, n_tiles, nullptr, XP); break;
                case 1: gru_tile_coop<CIN, LAST, 1, false, true>(lds, xch, lane, dir, tile, X, Y, P, n_tiles, nullptr, XP); break;
                case 2: gru_tile_coop<CIN, LAST, 2, false, true>(lds, xch, lane, dir, tile, X, Y, P, n_tiles, nullptr, XP); break;
                default: gru_tile_coop<CIN, LAST, 3, false, true>(lds, xch, lane, dir, tile, X, Y, P, n_tiles, nullptr, XP); break;
            }
        } else {
            switch (wave) {
                case 0: gru_tile_coop<CIN, LAST, 0>(lds, xch, lane, dir, tile, X, Y, P, n_tiles); break;
                case 1: gru_tile_coop<CIN, LAST, 1>(lds, xch, lane, dir, tile, X, Y, P, n_tiles); break;
                case 2: gru_tile_coop<CIN, LAST, 2>(lds, xch, lane, dir, tile, X, Y, P, n_tiles); break;
                default: gru_tile_coop<CIN, LAST, 3>(lds, xch, lane, dir, tile, X, Y, P, n_tiles); break;
            }
        }
        __syncthreads();
    }
}

template <int CIN>
__global__ __launch_bounds__(256, 1) void gru_train_fwd_coop_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ X,
                                                                    f32x4* __restrict__ Y, f32x4* __restrict__ S, int n_tiles,
                                                                    const f32x4* __restrict__ XP, f32x4* __restrict__ YD, cf_dropout drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK = gru_pack_floats(CIN);
    const int dir = blockIdx.y;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll 8
        for (int i = threadIdx.x; i < PACK / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xch = lds + PACK;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (XP) {
            switch (wave) {
                case 0: gru_tile_coop<CIN, false, 0, true, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, XP, YD, drop); break;
                case 1: gru_tile_coop<CIN, false, 1, true, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, XP, YD, drop); break;
                case 2: gru_tile_coop<CIN, false, 2, true, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, XP, YD, drop); break;
                default: gru_tile_coop<CIN, false, 3, true, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, XP, YD, drop); break;
            }
        } else {
            switch (wave) {
                case 0: gru_tile_coop<CIN, false, 0, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, nullptr, YD, drop); break;
                case 1: gru_tile_coop<CIN, false, 1, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, nullptr, YD, drop); break;
                case 2: gru_tile_coop<CIN, false, 2, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, nullptr, YD, drop); break;
                default: gru_tile_coop<CIN, false, 3, true>(lds, xch, lane, dir, tile, X, Y, nullptr, n_tiles, S, nullptr, YD, drop); break;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Cooperative BPTT (small training batches): four waves per tile.  Wave W owns hidden units 16W..16W+15
// (dh, du, dc, the three pre-activation gradients of those units) and the output M-tiles {h-row tile W,
// x-row tiles W, W+4, ...} of both backward products; the pre-activation gradients of all 64 units are
// exchanged through 12 KiB of LDS (three barriers per step).  Same arithmetic per element as
// gru_train_bwd_kernel.
// ------------------------------------------------------------------------------------------
// DEFER: the input gradient dx (the x-row tiles of both products) is off the serial chain; it is left to
// gru_dx_kernel, which forms it from the stored pre-activation gradients on the CUs the recurrence leaves idle.
template <int CIN, int W, bool DEFER = false>
__device__ __forceinline__ void gru_bwd_tile_coop(const float* lds, float* xch, int lane, int dir, int tile, const f32x4* __restrict__ Y,
                                                  const f32x4* __restrict__ S, const f32x4* __restrict__ DY,
                                                  const f32x4* __restrict__ DY2, const f32x4* __restrict__ DSC, f32x4* __restrict__ DX,
                                                  f32x4* __restrict__ DA, int n_tiles, cf_dropout drop) {
    const bool drop_on = !DSC && drop.keep_prob < 1.f;      // the layer's output dropout, mask recomputed (no stored mask)
    const uint32_t drop_key = drop_on ? cf_drop_key(drop) : 0u;
    constexpr int MI = (CIN + CF_H) / 16;
    constexpr int MX = CIN / 16;
    constexpr int NXW = DEFER ? 0 : (MX > W ? (MX - W + 3) / 4 : 0);     // x-row tiles of this wave: W, W+4, ...
    constexpr int CF2 = 16 * (MI / 2) * 64;                     // candidate region in f32x2 units
    // A fragments as dwords through a register ring, like the forward kernel: one sequence p = 0..47 per step
    // (16 k-steps of the candidate product, 32 of the gate product), 1 + NXW values each, fetched PF k-steps ahead.
    constexpr int PF = CF_COOP_PF;
    constexpr int NV = 1 + NXW;
    static_assert(48 % PF == 0, "ring slots must line up across the step boundary");
    const float* WF = lds + lane * 2;                           // f32x2 element e, component c at WF[e * 2 + c]
    float ring[PF][NV];
    auto fetch = [&](int p, float (&d)[NV]) {
        p = p % 48;
        const int e0 = p < 16 ? p * (MI / 2) * 64 : CF2 + (p - 16) * (MI / 2) * 64;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int tl = j == 0 ? MX + W : W + 4 * (j - 1);   // h-row tile first, then the x-row tiles
            d[j] = WF[(e0 + (tl >> 1) * 64) * 2 + (tl & 1)];
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p, ring[p]);
    f32x4* XC = reinterpret_cast<f32x4*>(xch) + lane;           // da_c tiles [4][64]
    f32x4* XR = XC + 4 * 64;                                    // da_r
    f32x4* XU = XR + 4 * 64;                                    // da_u
    const f32x4 one = {1.f, 1.f, 1.f, 1.f};
    f32x4 dhc = {0, 0, 0, 0};
    for (int s = CF_T - 1; s >= 0; --s) {
        const int t = dir ? (CF_T - 1 - s) : s;
        const int tp = dir ? (t + 1) : (t - 1);
        const int64_t base = (int64_t)tile * CF_T + t;
        const f32x4* sp = S + (base * 2 + dir) * 12 * 64 + lane;
        const f32x4 r = sp[(0 + W) * 64], u = sp[(4 + W) * 64], c = sp[(8 + W) * 64];
        f32x4 dy = DY[(base * 8 + dir * 4 + W) * 64 + lane];
        if (DY2) dy += DY2[(base * 8 + dir * 4 + W) * 64 + lane];
        if (DSC) dy *= DSC[(base * 8 + dir * 4 + W) * 64 + lane];
        else if (drop_on) dy *= cf_drop_scale4(drop_key, drop.keep_prob, (base * 8 + dir * 4 + W) * 64 + lane);
        const f32x4 dh = dhc + dy;
        f32x4 hp = {0, 0, 0, 0};
        if (s > 0) hp = Y[(((int64_t)tile * CF_T + tp) * 8 + dir * 4 + W) * 64 + lane];
        const f32x4 du = dh * (hp - c);
        const f32x4 dac = dh * (one - u) * (one - c * c);
        const f32x4 dau = du * u * (one - u);
        dhc = dh * u;
        XC[W * 64] = dac;
        XU[W * 64] = dau;
        __syncthreads();
        f32x4 dacf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) dacf[m] = XC[m * 64];
        f32x4 dx[NXW > 0 ? NXW : 1];
#pragma unroll
        for (int i = 0; i < (NXW > 0 ? NXW : 1); ++i) dx[i] = (f32x4){0, 0, 0, 0};
        f32x4 drh = {0, 0, 0, 0}, dhg = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float a[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) a[j] = ring[ks % PF][j];
            fetch(ks + PF, ring[ks % PF]);
            const float b = dacf[ks >> 2][ks & 3];
            drh = MFMA16(a[0], b, drh);
#pragma unroll
            for (int i = 0; i < NXW; ++i) dx[i] = MFMA16(a[1 + i], b, dx[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4 dar = drh * hp * r * (one - r);
        dhc += drh * r;
        XR[W * 64] = dar;
        __syncthreads();
        f32x4 dag[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) { dag[m] = XR[m * 64]; dag[4 + m] = XU[m * 64]; }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int p = 16 + ks;
            float a[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) a[j] = ring[p % PF][j];
            fetch(p + PF, ring[p % PF]);                            // wraps into the next step's candidate product
            const float b = dag[ks >> 2][ks & 3];
            dhg = MFMA16(a[0], b, dhg);
#pragma unroll
            for (int i = 0; i < NXW; ++i) dx[i] = MFMA16(a[1 + i], b, dx[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        dhc += dhg;
#pragma unroll
        for (int i = 0; i < NXW; ++i)
            DX[(((int64_t)dir * n_tiles * CF_T + base) * MX + (W + 4 * i)) * 64 + lane] = dx[i];
        f32x4* dap = DA + (base * 2 + dir) * 12 * 64 + lane;
        dap[(0 + W) * 64] = dar; dap[(4 + W) * 64] = dau; dap[(8 + W) * 64] = dac;
        __syncthreads();                                           // the exchange area is rewritten by the next step
    }
}

template <int CIN>
__global__ __launch_bounds__(256, 1) void gru_train_bwd_coop_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ Y,
                                                                    const f32x4* __restrict__ S, const f32x4* __restrict__ DY,
                                                                    const f32x4* __restrict__ DY2, const f32x4* __restrict__ DSC,
                                                                    f32x4* __restrict__ DX, f32x4* __restrict__ DA, int n_tiles,
                                                                    int defer_dx, cf_dropout drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK = ((CIN + CF_H) / 16 / 2) * 128 * 48;      // = gtb_pack_floats(CIN)
    const int dir = blockIdx.y;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll 8
        for (int i = threadIdx.x; i < PACK / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xch = lds + PACK;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (defer_dx) {
            switch (wave) {
                case 0: gru_bwd_tile_coop<CIN, 0, true>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                case 1: gru_bwd_tile_coop<CIN, 1, true>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                case 2: gru_bwd_tile_coop<CIN, 2, true>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                default: gru_bwd_tile_coop<CIN, 3, true>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
            }
        } else {
            switch (wave) {
                case 0: gru_bwd_tile_coop<CIN, 0>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                case 1: gru_bwd_tile_coop<CIN, 1>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                case 2: gru_bwd_tile_coop<CIN, 2>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
                default: gru_bwd_tile_coop<CIN, 3>(lds, xch, lane, dir, tile, Y, S, DY, DY2, DSC, DX, DA, n_tiles, drop); break;
            }
        }
    }
}

// dx of one layer for small batches, from the stored pre-activation gradients: one workgroup per (tile, t), wave W =
// x-row tiles W, W+4, ...; A fragments straight from the packed backward weights in global memory.  Same k order as
// the in-kernel version (16 k-steps against da_c, then 32 against da_r | da_u).
template <int CIN>
__global__ __launch_bounds__(256) void gru_dx_kernel(const float* __restrict__ wpack, const f32x4* __restrict__ DA,
                                                     f32x4* __restrict__ DX, int n_tiles, int chunks) {
    constexpr int MI = (CIN + CF_H) / 16;
    constexpr int MX = CIN / 16;
    constexpr int PACK = (MI / 2) * 128 * 48;
    constexpr int CF2 = 16 * (MI / 2) * 64;
    constexpr int NX = (MX + 3) / 4;                              // x-row tiles per wave (waves beyond MX idle)
    const int dir = blockIdx.y;
    const int tile = blockIdx.x / chunks, chunk = blockIdx.x - tile * chunks;
    const int lane = threadIdx.x & 63;
    const int W = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (W >= MX) return;
    const float* wf = wpack + (size_t)dir * PACK + lane * 2;
    const int TL = (CF_T + chunks - 1) / chunks;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int xt = W + 4 * i;
        if (xt >= MX) break;
        const float* wt = wf + ((xt >> 1) * 64) * 2 + (xt & 1);
        float a[48];                                   // one x-row tile's fragments, kept for the whole chunk of steps
#pragma unroll
        for (int p = 0; p < 48; ++p) a[p] = wt[(size_t)(p < 16 ? p * (MI / 2) * 64 : CF2 + (p - 16) * (MI / 2) * 64) * 2];
        for (int t = chunk * TL; t < min((chunk + 1) * TL, CF_T); ++t) {
            const int64_t base = (int64_t)tile * CF_T + t;
            const f32x4* dap = DA + (base * 2 + dir) * 12 * 64 + lane;
            f32x4 dag[8], dac[4];
#pragma unroll
            for (int m = 0; m < 8; ++m) dag[m] = dap[m * 64];
#pragma unroll
            for (int m = 0; m < 4; ++m) dac[m] = dap[(8 + m) * 64];
            f32x4 dx = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) dx = MFMA16(a[ks], dac[ks >> 2][ks & 3], dx);
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) dx = MFMA16(a[16 + ks], dag[ks >> 2][ks & 3], dx);
            DX[(((int64_t)dir * n_tiles * CF_T + base) * MX + xt) * 64 + lane] = dx;
        }
    }
}
