// gru_bf16.hpp -- biGRU layer on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), fp32 accumulate,
// fp32 hidden state.  Included by catfish_hip.hip (shares its helpers).
//
// Two precisions, selected by NP (number of bf16 parts per operand):
//   NP = 1  "bf16"   : operands rounded to bf16 (BASELINE config 4).
//   NP = 2  "bf16x3" : every fp32 operand v is split as v ~ hi + lo (hi = bf16(v), lo = bf16(v - hi),
//                      16-17 significant bits) and a*w is evaluated as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo
//                      (3 MFMAs, fp32 accumulate): an fp32-emulating split that keeps the 1e-4 gate.
//
// Unlike v_mfma_f32_16x16x4_f32 (which serialises with the VALU), the bf16 MFMA runs on
// the matrix pipe and co-executes with the sigmoid/tanh VALU work of the partner wave.
//
// Tile = 32 windows.  D[feature][window]: lane l holds window l&31 and rows
// (reg&3) + 8*(reg>>2) + 4*(l>>5) of a 32-feature M-tile.  As in the fp32 kernels the accumulator
// layout doubles as a B-operand layout under a permutation of k: registers 8b..8b+7 of M-tile m are
// k-block 2m+b, whose element j of lane-half hh is feature 32m + 16b + 8(j>>2) + 4hh + (j&3).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__host__ __device__ constexpr int frag_feature32(int kb, int hh, int j) {
    return 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
}

// Packed blob of one direction (bytes): A fragments of 1 KiB ([lane][8 bf16]) in schedule order
//   x part : [kb < CIN/16][mt < 6][part < NP]
//   gates h: [kb < 4][mt < 4][part]
//   cand  h: [kb < 4][mt < 2][part]     (M-tiles 4,5)
// then fp32 bias [mt < 6][hh < 2][16] and fp32 dense weights [mt < 2][hh][16].
__host__ __device__ constexpr int gb_seq(int cin) { return (cin / 16) * 6 + 16 + 8; }
__host__ __device__ constexpr int gb_bias_off(int cin, int np) { return gb_seq(cin) * np * 1024; }
__host__ __device__ constexpr int gb_dense_off(int cin, int np) { return gb_bias_off(cin, np) + 192 * 4; }
__host__ __device__ constexpr int gb_pack_bytes(int cin, int np) { return gb_dense_off(cin, np) + 64 * 4; }

// split 8 fp32 values into NP bf16 fragments (hi [, lo])
template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[NP]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)v[j];
        out[0][j] = hi;
        if constexpr (NP == 2) out[1][j] = (__bf16)(v[j] - (float)hi);
    }
}

template <int NP>
__device__ __forceinline__ f32x16 prod(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16 c) {
    c = MFMA32B(a[0], b[0], c);
    if constexpr (NP == 2) {
        c = MFMA32B(a[0], b[1], c);   // w_hi * a_lo
        c = MFMA32B(a[1], b[0], c);   // w_lo * a_hi
    }
    return c;
}

// Activations between layers: [tile32][t][kb][part][lane][8 bf16]  (16 B per lane, 1 KiB per fragment)
template <int CIN, bool LAST, int NP>
__global__ __launch_bounds__(512, 2) void gru_layer_bf16_kernel(const char* __restrict__ wpack,   // [2][gb_pack_bytes]
                                                                const bf16x8* __restrict__ X,      // KBX k-blocks
                                                                bf16x8* __restrict__ Y,            // 8 k-blocks
                                                                float* __restrict__ P,             // [2][tile][t][32]
                                                                int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KBX = CIN / 16;
    constexpr int NX = KBX * 6, NG = 16, NC = 8, NSEQ = NX + NG + NC;
    constexpr int PACK = gb_pack_bytes(CIN, NP);
    constexpr int DA = 3;                          // A-fragment ring depth (schedule entries ahead)
    constexpr int DX = (KBX >= 8 && NP == 2) ? KBX / 2 : KBX;   // x ring depth in k-blocks
    static_assert(NSEQ % DA == 0 && KBX % DX == 0, "ring depths must divide the schedule");

    const int dir = blockIdx.y;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5;
    const bf16x8* WA = reinterpret_cast<const bf16x8*>(lds) + lane;                     // + (p*NP + part)*64
    const far_lds<bf16x8> WAF(WA);
#define CF_WA(e) (WAF[(e)])
    const f32x4* BI = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + gb_bias_off(CIN, NP)) + hh * 4;
    const f32x4* DW = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + gb_dense_off(CIN, NP)) + hh * 4;

    const int nwaves = blockDim.x >> 6;
    for (int tile = blockIdx.x * nwaves + wave; tile < n_tiles; tile += gridDim.x * nwaves) {
        f32x16 h[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) h[m][i] = 0.f;                                 // GRUCellZeroState
        bf16x8 hp[4][NP];                                                               // bf16 parts of h per k-block
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[kb][p][j] = (__bf16)0.f;

        const int tstep = dir ? -1 : 1;                                                 // bw = reversed time
        const int t0 = dir ? (CF_T - 1) : 0;
        // x ring: k-block g of the whole 35-step sequence lives in slot g % DX
        bf16x8 xr[DX][NP];
        auto load_xblock = [&](int g, bf16x8 (&dst)[NP]) {
            int s = g / KBX;
            const int kb = g - s * KBX;
            s = s > CF_T - 1 ? CF_T - 1 : s;                                            // past the end: harmless re-read
            const int t = t0 + s * tstep;
            const bf16x8* src = X + (((int64_t)tile * CF_T + t) * KBX + kb) * NP * 64 + lane;
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p] = src[p * 64];
        };
#pragma unroll
        for (int g = 0; g < DX; ++g) load_xblock(g, xr[g]);
        // A ring
        bf16x8 ar[DA][NP];
#pragma unroll
        for (int q = 0; q < DA; ++q)
#pragma unroll
            for (int p = 0; p < NP; ++p) ar[q][p] = CF_WA((q * NP + p) * 64);

        for (int s = 0; s < CF_T; ++s) {
            const int t = t0 + s * tstep;
            f32x16 acc[6];
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x4 b = BI[mt * 8 + c4];
                    acc[mt][4 * c4 + 0] = b.x; acc[mt][4 * c4 + 1] = b.y; acc[mt][4 * c4 + 2] = b.z; acc[mt][4 * c4 + 3] = b.w;
                }
            }
            // one schedule entry: consume ring slot p % DA, refill it DA entries ahead
#define CF_ENTRY(p, bparts, mt)                                                            \
    {                                                                                      \
        bf16x8 a_[NP];                                                                     \
        _Pragma("unroll") for (int pp = 0; pp < NP; ++pp) a_[pp] = ar[(p) % DA][pp];       \
        _Pragma("unroll") for (int pp = 0; pp < NP; ++pp)                                  \
            ar[(p) % DA][pp] = CF_WA(((((p) + DA) % NSEQ) * NP + pp) * 64);                \
        acc[mt] = prod<NP>(a_, bparts, acc[mt]);                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
            // x part: [r | u | c] += Wx^T x_t
#pragma unroll
            for (int kb = 0; kb < KBX; ++kb) {
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) CF_ENTRY(kb * 6 + mt, xr[kb % DX], mt);
                load_xblock(s * KBX + kb + DX, xr[kb % DX]);
            }
            // gates, h part                                          (gru_cell/MatMul)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) CF_ENTRY(NX + kb * 4 + mt, hp[kb], mt);
            }
            // r = sigmoid(.), r*h -> bf16 parts (reset BEFORE the candidate matmul: gru_cell/mul -> concat_1)
            bf16x8 rp[4][NP];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int m = kb >> 1, i = 8 * (kb & 1) + j;
                    const f32x2 rh = cf_sigmoid_pre2((f32x2){acc[m][i], acc[m][i + 1]}) * (f32x2){h[m][i], h[m][i + 1]};
                    v[j] = rh.x;
                    v[j + 1] = rh.y;
                }
                split8<NP>(v, rp[kb]);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) CF_ENTRY(NX + NG + kb * 2 + mt, rp[kb], 4 + mt);
            }
#undef CF_ENTRY
            // h' = u*h + (1-u)*c                                      (gru_cell/mul_1, sub, mul_2, add)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 u = cf_sigmoid_pre2((f32x2){acc[2 + m][i], acc[2 + m][i + 1]});
                    const f32x2 c = cf_tanh_pre2((f32x2){acc[4 + m][i], acc[4 + m][i + 1]});
                    const f32x2 hn = __builtin_elementwise_fma(u, (f32x2){h[m][i], h[m][i + 1]} - c, c);
                    h[m][i] = hn.x;
                    h[m][i + 1] = hn.y;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = h[kb >> 1][8 * (kb & 1) + j];
                split8<NP>(v, hp[kb]);
            }
            if constexpr (!LAST) {
                bf16x8* dst = Y + (((int64_t)tile * CF_T + t) * 8 + dir * 4) * NP * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) dst[(kb * NP + p) * 64] = hp[kb][p];
            } else {
                // partial logit of this direction (final_fully_connected/MatMul), fp32 h
                float pl = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const f32x4 wd = DW[m * 8 + c4];
                        pl = fmaf(wd.x, h[m][4 * c4 + 0], pl); pl = fmaf(wd.y, h[m][4 * c4 + 1], pl);
                        pl = fmaf(wd.z, h[m][4 * c4 + 2], pl); pl = fmaf(wd.w, h[m][4 * c4 + 3], pl);
                    }
                }
                pl += __shfl_xor(pl, 32);
                if (lane < 32) P[(((int64_t)dir * n_tiles + tile) * CF_T + t) * 32 + lane] = pl;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Residual block on the bf16 matrix pipe (used when precision != fp32): same streaming structure as
// res_block_kernel, tile = 32 windows, every 32x32 conv unit = 2 k-blocks x (1 or 3) MFMAs.
// Blob (bytes): units [unit][kb < 2][part < NP] fragments of 1 KiB, then fp32 vectors [vec][hh < 2][16]:
//   first block: units {c3 tap0, tap1, tap2, last}, vectors {b_c3, b_last, w_sc, b_sc, w_first, b_first}
//   other blocks: units {sc, first, c3 tap0, tap1, tap2, last}, vectors {b_sc, b_first, b_c3, b_last}
// Activations in and out: [tile32][t][kb < 2][part][lane][8 bf16] (the GRU kernels' input layout).
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int rb_vec_off(bool first, int np) { return res_units(first) * 2 * np * 1024; }
__host__ __device__ constexpr int rb_pack_bytes(bool first, int np) { return rb_vec_off(first, np) + res_vecs(first) * 128; }

template <int NP>
__device__ __forceinline__ void split16(const f32x16& v, bf16x8 (&out)[2][NP]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[8 * kb + j];
        split8<NP>(t, out[kb]);
    }
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    return v;
}

template <bool FIRST, int NP>
__global__ __launch_bounds__(256) void res_block_bf16_kernel(const char* __restrict__ wpack,
                                                             const float* __restrict__ x_nat,    // FIRST: [n_windows, 35]
                                                             const bf16x8* __restrict__ x_in,    // !FIRST
                                                             bf16x8* __restrict__ y_out,
                                                             int64_t n_windows, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK = rb_pack_bytes(FIRST, NP);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5;
    const int nwaves = blockDim.x >> 6;
    const bf16x8* WU = reinterpret_cast<const bf16x8*>(lds) + lane;                          // + ((unit*2 + kb)*NP + part)*64
    const char* vbase = reinterpret_cast<const char*>(lds) + rb_vec_off(FIRST, NP) + hh * 64;
    float* xs = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + PACK) + wave * (32 * CF_T);   // FIRST: x tile [32][35]

    auto vec = [&](int v) -> f32x16 {
        f32x16 o;
        const f32x4* p = reinterpret_cast<const f32x4*>(vbase + v * 128);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 b = p[c4];
            o[4 * c4 + 0] = b.x; o[4 * c4 + 1] = b.y; o[4 * c4 + 2] = b.z; o[4 * c4 + 3] = b.w;
        }
        return o;
    };
    auto unit = [&](int u, const bf16x8 (&in)[2][NP], f32x16 acc) -> f32x16 {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 a[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = WU[((u * 2 + kb) * NP + p) * 64];
            acc = prod<NP>(a, in[kb], acc);
        }
        return acc;
    };

    for (int tile = blockIdx.x * nwaves + wave; tile < n_tiles; tile += gridDim.x * nwaves) {
        if constexpr (FIRST) {
            const int64_t base = (int64_t)tile * 32 * CF_T;
            const int64_t limit = n_windows * CF_T;
            for (int i = lane; i < 32 * CF_T; i += 64) xs[i] = (base + i < limit) ? x_nat[base + i] : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        bf16x8 o1pp[2][NP], o1p[2][NP], o1c[2][NP];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) { o1pp[kb][p][j] = (__bf16)0.f; o1p[kb][p][j] = (__bf16)0.f; }
        f32x16 sc_p, sc_c;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sc_p[i] = 0.f; sc_c[i] = 0.f; }
        for (int i = 0; i <= CF_T; ++i) {
            if (i < CF_T) {
                if constexpr (FIRST) {
                    const float xv = xs[(lane & 31) * CF_T + i];
                    const f32x16 w_sc = vec(2), b_sc = vec(3), w_f = vec(4), b_f = vec(5);
                    f32x16 o1;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { sc_c[k] = fmaf(w_sc[k], xv, b_sc[k]); o1[k] = fmaxf(fmaf(w_f[k], xv, b_f[k]), 0.f); }
                    split16<NP>(o1, o1c);
                } else {
                    bf16x8 in[2][NP];
                    const bf16x8* src = x_in + (((int64_t)tile * CF_T + i) * 2) * NP * 64 + lane;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int p = 0; p < NP; ++p) in[kb][p] = src[(kb * NP + p) * 64];
                    sc_c = unit(0, in, vec(0));                               // shortcut, no relu
                    split16<NP>(relu16(unit(1, in, vec(1))), o1c);            // first conv + relu
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int j = 0; j < 8; ++j) o1c[kb][p][j] = (__bf16)0.f;   // zero padding past the window end
            }
            if (i >= 1) {
                constexpr int U3 = FIRST ? 0 : 2;
                constexpr int VB3 = FIRST ? 0 : 2;
                f32x16 acc = vec(VB3);
                acc = unit(U3 + 0, o1pp, acc);
                acc = unit(U3 + 1, o1p, acc);
                acc = unit(U3 + 2, o1c, acc);
                bf16x8 o2[2][NP];
                split16<NP>(relu16(acc), o2);
                f32x16 out = relu16(unit(U3 + 3, o2, vec(VB3 + 1)));
#pragma unroll
                for (int k = 0; k < 16; ++k) out[k] = fmaxf(out[k] + sc_p[k], 0.f);
                bf16x8 op[2][NP];
                split16<NP>(out, op);
                bf16x8* dst = y_out + (((int64_t)tile * CF_T + (i - 1)) * 2) * NP * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) dst[(kb * NP + p) * 64] = op[kb][p];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int p = 0; p < NP; ++p) { o1pp[kb][p] = o1p[kb][p]; o1p[kb][p] = o1c[kb][p]; }
            sc_p = sc_c;
        }
        if constexpr (FIRST) __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Residual blocks 0 AND 1 in one launch on the bf16 matrix pipe (resnet_class.py:17-25,44-82), the bf16 counterpart of
// res_stack2_kernel.  Block 1 runs one position behind block 0 in the same stream: block 0's output is rounded to bf16
// in registers -- exactly the fragments res_block_bf16_kernel<true> would have stored and <false> re-loaded, so the
// result is bit-identical to the two launches -- and never exists in HBM (128 B per sample of traffic and a launch
// less).  The two launches ran one wave per 32-window tile: 944 waves on 1024 SIMDs walking 36 serial positions with
// nothing to overlap the LDS / MFMA / conversion latencies (matrix pipe busy 9-12 %).  Here a tile's 35 positions are cut
// into `t_chunks` chunks, one wave each (a chunk starts two positions early and ends one late to rebuild the k = 3
// neighbourhoods of both blocks); the launcher picks as many chunks as still fit the chip in one round of resident waves.
// A single read's conv stack is then a few positions deep instead of 2 x 35 (50 -> 11.5 us).
// Instruction diet (per position and tile, bf16: ~300 -> ~175 VALU): relu after rounding on packed bf16 pairs
// (v_pk_max_i16), fp32 relu as v_max_i32, conversions in pairs, packed fp32 fma / add, fragments carried as dwords, rings of
// three fragments in an unrolled-by-three loop instead of register shifts.  Measured (profiles/r03_res_bf16_sweep.json):
// 256 reads 73.9 -> 51.8 us, 131 072 windows 234 -> 149 us.  TPW = 2 (two tiles per wave: every LDS read of a weight
// fragment or bias vector feeds two MFMAs) halves the LDS traffic but needs 237 VGPRs (two waves per SIMD) and measured
// 7 % slower; it stays behind the CATFISH_RES_TPW debug knob.
// LDS: both blocks' blobs (rb_pack_bytes) + TPW x tiles [32][35] fp32 per wave.
// ------------------------------------------------------------------------------------------
// two fp32 -> one dword of two bf16 (round to nearest even): a 2-vector conversion compiles to ONE v_cvt_pk_bf16_f32; sixteen
// scalar (__bf16) casts in a row come out as sixteen single conversions merged by eight v_perm_b32.  (No inline asm here: the
// values come straight from MFMA results, and the wait states between an MFMA and a VALU read of its result are inserted by
// the compiler's hazard recogniser, which cannot see into an asm statement.)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}

// relu as ONE instruction per element: a float is negative iff its int32 pattern is, so max(bits, 0) as signed integers
// (v_max_i32) is relu.  fmaxf compiles to two v_max_f32 (IEEE mode first quiets a possible signalling NaN with max(x, x)).
__device__ __forceinline__ f32x16 relu16_1op(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float f = v[i];                              // (a copy: bit-casting the vector ELEMENT itself reads element 0 every time)
        const int b = __builtin_bit_cast(int, f);
        v[i] = __builtin_bit_cast(float, b > 0 ? b : 0);
    }
    return v;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

// relu + rounding to bf16 fragments.  One bf16 part: round first, then clamp the PACKED pairs with v_pk_max_i16 against 0 -- a
// bf16 is negative iff its int16 pattern is, and rounding keeps the sign, so relu(round(v)) == round(relu(v)) bit for bit at
// half the VALU instructions (8 conversions + 8 packed max instead of 16 max + 8 conversions).  hi + lo split: relu in fp32
// first.  Fragments are carried as four DWORDS (u32x4), not as eight bf16: across the kernel's branches the compiler otherwise
// tracks every 16-bit element on its own and re-packs them at each join (v_lshrrev + v_perm_b32, 16 instructions per fragment).
template <int NP>
__device__ __forceinline__ void relu_split16(const f32x16& v, u32x4 (&out)[2][NP]) {
    if constexpr (NP == 1) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const u32x4 d = {cvt_pk_bf16(v[8 * kb + 0], v[8 * kb + 1]), cvt_pk_bf16(v[8 * kb + 2], v[8 * kb + 3]),
                             cvt_pk_bf16(v[8 * kb + 4], v[8 * kb + 5]), cvt_pk_bf16(v[8 * kb + 6], v[8 * kb + 7])};
            const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            out[kb][0] = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(i16x8, d), z));
        }
    } else {
        bf16x8 parts[2][NP];
        split16<NP>(relu16_1op(v), parts);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p) out[kb][p] = __builtin_bit_cast(u32x4, parts[kb][p]);
    }
}

#ifndef CF_RES_BF16_WAVES
#define CF_RES_BF16_WAVES 3            // waves per SIMD the one-tile bf16 kernel is compiled for (131 VGPRs; 4 would spill 5 dwords)
#endif
template <int NP, int TPW>
__global__ __launch_bounds__(256, (NP == 1 && TPW == 1) ? CF_RES_BF16_WAVES : 2) void res_stack2_bf16_kernel(
    const char* __restrict__ wpack0, const char* __restrict__ wpack1,
    const float* __restrict__ x_nat,    // [n_windows, 35]
    bf16x8* __restrict__ y_out,         // block 1's output
    int64_t n_windows, int n_tiles, int t_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PACK0 = rb_pack_bytes(true, NP), PACK1 = rb_pack_bytes(false, NP);
    {
        const f32x4* src0 = reinterpret_cast<const f32x4*>(wpack0);
        const f32x4* src1 = reinterpret_cast<const f32x4*>(wpack1);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK0 / 16; i += blockDim.x) dst[i] = src0[i];
        for (int i = threadIdx.x; i < PACK1 / 16; i += blockDim.x) dst[PACK0 / 16 + i] = src1[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5;
    const int nwaves = blockDim.x >> 6;
    const char* base = reinterpret_cast<const char*>(lds);
    const bf16x8* WU0 = reinterpret_cast<const bf16x8*>(base) + lane;            // block 0: units {c3 tap0, tap1, tap2, last}
    const bf16x8* WU1 = reinterpret_cast<const bf16x8*>(base + PACK0) + lane;    // block 1: units {sc, first, c3 tap0, tap1, tap2, last}
    const char* vbase0 = base + rb_vec_off(true, NP) + hh * 64;                  // {b_c3, b_last, w_sc, b_sc, w_first, b_first}
    const char* vbase1 = base + PACK0 + rb_vec_off(false, NP) + hh * 64;         // {b_sc, b_first, b_c3, b_last}
    float* xs = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + PACK0 + PACK1) + wave * (TPW * 32 * CF_T);

    typedef u32x4 frag[2][NP];                                                   // 32 features of 32 windows: 2 k-blocks x NP parts of 8 bf16
    auto vec = [&](const char* vb, int v) -> f32x16 {
        f32x16 o;
        const f32x4* p = reinterpret_cast<const f32x4*>(vb + v * 128);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 b = p[c4];
            o[4 * c4 + 0] = b.x; o[4 * c4 + 1] = b.y; o[4 * c4 + 2] = b.z; o[4 * c4 + 3] = b.w;
        }
        return o;
    };
    // one conv unit on the wave's TPW tiles: every A fragment is read from LDS once and feeds TPW MFMAs
    auto unit = [&](const bf16x8* WU, int u, const frag (&in)[TPW], f32x16 (&acc)[TPW]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 a[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = WU[((u * 2 + kb) * NP + p) * 64];
#pragma unroll
            for (int w = 0; w < TPW; ++w) {
                bf16x8 b[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(bf16x8, in[w][kb][p]);
                acc[w] = prod<NP>(a, b, acc[w]);
            }
        }
    };
    auto zero_frag = [&](frag (&f)[TPW]) {
#pragma unroll
        for (int w = 0; w < TPW; ++w)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int p = 0; p < NP; ++p) f[w][kb][p] = (u32x4){0u, 0u, 0u, 0u};
    };
    auto copy_frag = [&](frag (&d)[TPW], const frag (&s)[TPW]) {
#pragma unroll
        for (int w = 0; w < TPW; ++w)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int p = 0; p < NP; ++p) d[w][kb][p] = s[w][kb][p];
    };

    const int chunk_len = (CF_T + t_chunks - 1) / t_chunks;
    const int n_groups = (n_tiles + TPW - 1) / TPW;                              // TPW consecutive tiles per wave
    for (int task = blockIdx.x * nwaves + wave; task < n_groups * t_chunks; task += gridDim.x * nwaves) {
        const int group = task / t_chunks;
        const int tile0 = group * TPW;
        const int p0 = (task - group * t_chunks) * chunk_len;                    // this wave writes positions [p0, p1)
        const int p1 = min(p0 + chunk_len, CF_T);
        if (p0 >= p1) continue;
        {
            const int64_t xbase = (int64_t)tile0 * 32 * CF_T;
            const int64_t limit = n_windows * CF_T;
            for (int i = lane; i < TPW * 32 * CF_T; i += 64) xs[i] = (xbase + i < limit) ? x_nat[xbase + i] : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // Rings of three fragments instead of shifting registers: block 0's first-conv outputs o1 (ra), block 1's (rb) and block
        // 0's bf16-rounded output = block 1's input (ry).  Position i writes slot (i - i0) % 3; the loop is unrolled by three so
        // that every slot is a fixed register range (shifting cost 32 v_mov per position).
        frag ra[3][TPW], rb[3][TPW], ry[3][TPW];
#pragma unroll
        for (int r = 0; r < 3; ++r) { zero_frag(ra[r]); zero_frag(rb[r]); zero_frag(ry[r]); }
        // The shortcut branches are evaluated where they are added (same operations on the same inputs as in the two-launch
        // kernels, which carry them in registers from the position before): 64 VGPRs less per tile.
        auto step = [&](int i, const frag (&a_pp)[TPW], const frag (&a_p)[TPW], frag (&a_c)[TPW], const frag (&b_pp)[TPW],
                        const frag (&b_p)[TPW], frag (&b_c)[TPW], const frag (&y0p)[TPW], frag (&y0)[TPW]) {
            // ---- block 0, front at position i: Cin = 1, conv1d is a rank-1 update (resnet_class.py:60,64)
            if (i < CF_T) {
                const f32x16 w_f = vec(vbase0, 4), b_f = vec(vbase0, 5);
#pragma unroll
                for (int w = 0; w < TPW; ++w) {
                    const float xv = xs[(w * 32 + (lane & 31)) * CF_T + i];
                    f32x16 xv16;
#pragma unroll
                    for (int k = 0; k < 16; ++k) xv16[k] = xv;
                    relu_split16<NP>(__builtin_elementwise_fma(w_f, xv16, b_f), a_c[w]);      // v_pk_fma_f32
                }
            } else {
                zero_frag(a_c);                                                  // zero padding past the window end
            }
            // ---- block 0, back: its output at position i - 1, rounded to bf16, stays in registers as block 1's input
            const bool have_y0 = i >= 1 && i <= CF_T && i >= p0;
            if (have_y0) {
                f32x16 acc[TPW], out[TPW];
                {
                    const f32x16 b = vec(vbase0, 0);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) acc[w] = b;
                }
                unit(WU0, 0, a_pp, acc);
                unit(WU0, 1, a_p, acc);
                unit(WU0, 2, a_c, acc);
                frag o2[TPW];
#pragma unroll
                for (int w = 0; w < TPW; ++w) relu_split16<NP>(acc[w], o2[w]);
                {
                    const f32x16 b = vec(vbase0, 1);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) out[w] = b;
                }
                unit(WU0, 3, o2, out);
                {
                    const f32x16 w_sc = vec(vbase0, 2), b_s = vec(vbase0, 3);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) {
                        const float xp = xs[(w * 32 + (lane & 31)) * CF_T + (i - 1)];
                        f32x16 xp16;
#pragma unroll
                        for (int k = 0; k < 16; ++k) xp16[k] = xp;
                        relu_split16<NP>(relu16_1op(out[w]) + __builtin_elementwise_fma(w_sc, xp16, b_s), y0[w]);   // relu, shortcut, add, relu
                    }
                }
                // ---- block 1, front at position i - 1: first conv + relu
                {
                    const f32x16 b = vec(vbase1, 1);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) acc[w] = b;
                }
                unit(WU1, 1, y0, acc);
#pragma unroll
                for (int w = 0; w < TPW; ++w) relu_split16<NP>(acc[w], b_c[w]);
            } else {
                zero_frag(b_c);                                                  // before the chunk, or zero padding at t = T
            }
            // ---- block 1, back: output position i - 2
            if (i >= 2 && i - 2 >= p0 && i - 2 < p1) {
                f32x16 acc[TPW], out[TPW], sc[TPW];
                {
                    const f32x16 b = vec(vbase1, 2);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) acc[w] = b;
                }
                unit(WU1, 2, b_pp, acc);
                unit(WU1, 3, b_p, acc);
                unit(WU1, 4, b_c, acc);
                frag o2[TPW];
#pragma unroll
                for (int w = 0; w < TPW; ++w) relu_split16<NP>(acc[w], o2[w]);
                {
                    const f32x16 b = vec(vbase1, 3), bs = vec(vbase1, 0);
#pragma unroll
                    for (int w = 0; w < TPW; ++w) { out[w] = b; sc[w] = bs; }
                }
                unit(WU1, 5, o2, out);
                unit(WU1, 0, y0p, sc);                                           // shortcut of position i - 2, no relu
#pragma unroll
                for (int w = 0; w < TPW; ++w) {
                    if (tile0 + w >= n_tiles) break;                             // the odd last tile of the call
                    u32x4 op[2][NP];
                    relu_split16<NP>(relu16_1op(out[w]) + sc[w], op);                 // relu, add, relu
                    u32x4* dst = reinterpret_cast<u32x4*>(y_out) + (((int64_t)(tile0 + w) * CF_T + (i - 2)) * 2) * NP * 64 + lane;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int p = 0; p < NP; ++p) dst[(kb * NP + p) * 64] = op[kb][p];
                }
            }
        };
        const int i_end = p1 + 1;
        for (int i = p0 > 2 ? p0 - 2 : 0;; i += 3) {
            step(i, ra[1], ra[2], ra[0], rb[1], rb[2], rb[0], ry[2], ry[0]);
            if (i + 1 > i_end) break;
            step(i + 1, ra[2], ra[0], ra[1], rb[2], rb[0], rb[1], ry[0], ry[1]);
            if (i + 2 > i_end) break;
            step(i + 2, ra[0], ra[1], ra[2], rb[0], rb[1], rb[2], ry[1], ry[2]);
            if (i + 3 > i_end) break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- host-side packing -------------------------------------------------------------------
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);   // inf / nan: truncate
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static void pack_gru_dir_bf16(const cf_gru_dir& g, int cin, int np, const float* dense_w, char* out) {
    auto wfull = [&](int in, int o) -> float {   // pre-scaled (CF_GATE_SCALE / CF_CAND_SCALE), then split into bf16 parts
        return o < 2 * CF_H ? (float)(CF_GATE_SCALE * (double)g.gates_kernel[(size_t)in * 2 * CF_H + o])
                            : (float)(CF_CAND_SCALE * (double)g.candidate_kernel[(size_t)in * CF_H + (o - 2 * CF_H)]);
    };
    uint16_t* frag = reinterpret_cast<uint16_t*>(out);
    int p = 0;
    auto emit = [&](int in_base, int kb, int mt) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const float w = wfull(in_base + frag_feature32(kb, lane >> 5, j), 32 * mt + (lane & 31));
                const uint16_t hi = f32_to_bf16_rne(w);
                frag[((size_t)(p * np + 0) * 64 + lane) * 8 + j] = hi;
                if (np == 2) frag[((size_t)(p * np + 1) * 64 + lane) * 8 + j] = f32_to_bf16_rne(w - bf16_to_f32(hi));
            }
        ++p;
    };
    for (int kb = 0; kb < cin / 16; ++kb) for (int mt = 0; mt < 6; ++mt) emit(0, kb, mt);
    for (int kb = 0; kb < 4; ++kb) for (int mt = 0; mt < 4; ++mt) emit(cin, kb, mt);
    for (int kb = 0; kb < 4; ++kb) for (int mt = 4; mt < 6; ++mt) emit(cin, kb, mt);
    float* pb = reinterpret_cast<float*>(out + gb_bias_off(cin, np));
    for (int mt = 0; mt < 6; ++mt)
        for (int hh = 0; hh < 2; ++hh)
            for (int i = 0; i < 16; ++i) {
                const int o = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hh;
                pb[(mt * 2 + hh) * 16 + i] = o < 2 * CF_H ? (float)(CF_GATE_SCALE * (double)g.gates_bias[o])
                                                          : (float)(CF_CAND_SCALE * (double)g.candidate_bias[o - 2 * CF_H]);
            }
    float* pd = reinterpret_cast<float*>(out + gb_dense_off(cin, np));
    for (int m = 0; m < 2; ++m)
        for (int hh = 0; hh < 2; ++hh)
            for (int i = 0; i < 16; ++i)
                pd[(m * 2 + hh) * 16 + i] = dense_w ? dense_w[32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh] : 0.f;
}

// 32x32 conv unit -> [kb < 2][part][lane][8 bf16]: A[row = out feature lane&31][k] with k = frag_feature32(kb, hh, j)
static void pack_unit_bf16(char* dst, const std::vector<double>& w /*[32 in][32 out]*/, int np) {
    uint16_t* frag = reinterpret_cast<uint16_t*>(dst);
    for (int kb = 0; kb < 2; ++kb)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const float v = (float)w[frag_feature32(kb, lane >> 5, j) * 32 + (lane & 31)];
                const uint16_t hi = f32_to_bf16_rne(v);
                frag[((size_t)(kb * np + 0) * 64 + lane) * 8 + j] = hi;
                if (np == 2) frag[((size_t)(kb * np + 1) * 64 + lane) * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(hi));
            }
}
// fp32 vector in D-layout order: [hh][i] = v[(i&3) + 8(i>>2) + 4hh]
static void pack_vec32(char* dst, const std::vector<double>& v) {
    float* o = reinterpret_cast<float*>(dst);
    for (int hh = 0; hh < 2; ++hh)
        for (int i = 0; i < 16; ++i) o[hh * 16 + i] = (float)v[(i & 3) + 8 * (i >> 2) + 4 * hh];
}
