// gru_bf16x3_pipe.hpp -- biGRU layer in bf16x3 (NP = 2: split-operand fp32 emulation, three bf16 MFMAs per product),
// software-pipelined for ONE wave per SIMD.  Included by catfish_hip.hip after gru_bf16.hpp: same weight blob, same
// activation layout, same products in the same order per accumulator and the same activation arithmetic as
// gru_layer_bf16_kernel<CIN, LAST, 2> (rnn_class.py:142-175).  (Not its bits: that kernel's compiler contracts most of its
// `r*h - hi` into an fma, which moves single bf16 lo parts by one ulp; every operation here is pinned.  The two agree to 1.2e-6
// in probability and have the same error against the fp64 oracle: tests/test_gpu_parity.py.)
//
// Why a kernel of its own: bf16x3 is matrix-pipe bound (216 MFMAs of 32 cycles per 32-window step against ~4 k cycles of
// vector issue), but the round-1 kernel runs a step as [MFMAs] -> [sigmoid] -> [MFMAs] -> [sigmoid, tanh] and leaves the pipe
// idle half of the time.  gru_bf16_pipe.hpp's structure does not carry over as it stands: with hi + lo parts the two x
// buffers, the operand fragments of h and r*h and the A ring double, ~430 registers.  So this kernel takes the whole
// 512-register file (one wave per SIMD, four per workgroup) and has to hide its vector work in its OWN MFMA gaps: an MFMA
// holds the SIMD's issue port for 8 of its 32 cycles, the other 24 take fillers of the same wave if -- and only if -- they
// are spread evenly (MI355X_MICROARCH.md, 'one wave per SIMD: single-issue instructions hidden per gap'); whatever a gap
// carries beyond 24 cycles of issue stalls the matrix pipe and is never recovered.
//
// Schedule of one step (product = 3 MFMAs: w_hi*a_hi, w_hi*a_lo, w_lo*a_hi; KBX = CIN/16 k-blocks of x; XUA, XCA: x3::geom):
//
//   A  acc_u += Wx_u x_s (last XUA), then Wh_u h  8   ||  r = sigmoid(acc_r), r*h -> bf16 hi + lo (rp)
//      acc_c  = b_c + Wx_c x_s        first XCA
//   B  acc_c += Wx_c x_s (rest), then Wh_c rp  8      ||  u = sigmoid(acc_u) (kept in registers of its own)
//   C  acc_r  = b_r + Wx_r x_{s+1}    2 KBX           ||  c = tanh(acc_c), h' = c + u (h - c), h' -> bf16 hi + lo (hp), stores,
//      acc_u  = b_u + Wx_u x_{s+1}    first 2 KBX - XUA   loads of x_{s+2}
//      acc_r += Wh_r h'               8, k-block kb once hp[kb] is complete
//
// The phases are sized by their vector work (C carries most of it), which is what the spare registers buy: with u out of its
// accumulator the x projection of BOTH gates of step s + 1 rides behind the h update of step s.  The vector work is cut into
// single instructions (one v_exp, one v_rcp, one add ...), listed per phase in software-pipelined order (element k's exp2,
// element k-L's add, element k-2L's rcp ...) and packed into the phase's gaps by cumulative issue cost at compile time
// (x3::make_sched): every gap gets the same load.  A sched_barrier per gap pins the order; static_asserts check that
// every MFMA finds its operands complete in program order.  DESIGN.md section 4 has the measurements and what bounds it.
#pragma once

// Diagnostic build: s_memtime stamps around the three phases of every step, summed per wave and written to the (otherwise
// unused) dense-partial buffer P of the non-LAST kernels as int64 [dir][tile][8] = {A, B, C cycles, s_memrealtime ticks (100 MHz) of
// the task, total cycles, start, prologue cycles, id}: total / realtime x 100 MHz is the clock the chip held
#ifndef CF_X3_STAMP
#define CF_X3_STAMP 0
#endif
// Timing-only ablation (results are WRONG when set): bit 0 = no vector work, bit 1 = no MFMAs, bit 2 = no A-ring refills (the three
// fragments of the prologue are reused for every product: the upper bound of what holding weight fragments in registers could save)
#ifndef CF_X3_ABL
#define CF_X3_ABL 0
#endif
// non-temporal hint on the layer output stores (bit 0: on, measured -2..3 % on the mid layer) and the layer input loads (bit 1:
// measured neutral, off)
#ifndef CF_X3_NT
#define CF_X3_NT 1
#endif

#include <utility>
#include "gru_bf16x3_sched.hpp"

// the value as an opaque VGPR definition at this point of the program: IR passes can neither merge it with a neighbour into a
// packed instruction nor move it past a sched_barrier (both are side effects, whose order is kept)
__device__ __forceinline__ float x3_pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ unsigned x3_pin(unsigned v) { asm volatile("" : "+v"(v)); return v; }

// f(std::integral_constant<int, I>{}) for every I of the sequence, in order
template <int... I, class F>
__device__ __forceinline__ void x3_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int CIN, bool LAST>
__global__ __launch_bounds__(256, 1) void gru_bf16x3_pipe_kernel(const char* __restrict__ wpack,   // [2][gb_pack_bytes(CIN, 2)]
                                                                 const bf16x8* __restrict__ X,      // [tile32][t][KBX][part][lane]
                                                                 bf16x8* __restrict__ Y,            // [tile32][t][8][part][lane]
                                                                 float* __restrict__ P,             // [2][tile][t][32]
                                                                 int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = x3::geom<CIN>;
    using T = x3::table<CIN, LAST>;
    constexpr int KBX = G::KBX, NSEQ = G::NSEQ, NGAP = G::NGAP;
    constexpr int DA = 3;                                   // A ring depth in products (2 fragments each)
    constexpr int PACK = gb_pack_bytes(CIN, 2);
    static_assert(NSEQ % DA == 0, "ring depth must divide the schedule");

    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int stride = gridDim.x * nwaves;
    const int tstep = dir ? -1 : 1;                         // bw = reversed time
    const int t0 = dir ? (CF_T - 1) : 0;
    // Global traffic goes through buffer descriptors: one per tile (wave-uniform, built with scalar arithmetic), this lane's 16
    // bytes as the constant VGPR offset and the (step, fragment) offset in an SGPR.  Written as pointers the compiler keeps a
    // 64-bit VGPR address per 4 KiB and re-derives it (two v_add_co + wait states) inside the MFMA gaps.
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr int X_TILE_BYTES = CF_T * KBX * 2 * 1024, Y_TILE_BYTES = CF_T * 8 * 2 * 1024;
    auto x_rsrc = [&](int tile_) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(X)) + (int64_t)tile_ * X_TILE_BYTES, 0,
                                                 X_TILE_BYTES, 0x00020000);
    };
    bf16x8 x0[KBX][2], x1[KBX][2];
    auto load_x = [&](bf16x8 (&dst)[KBX][2], __amdgpu_buffer_rsrc_t rx, int s, int q) {   // q = kb * 2 + part
        s = s > CF_T - 1 ? CF_T - 1 : s;                                   // past the end: harmless re-read
        dst[q >> 1][q & 1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, lane16, ((t0 + s * tstep) * KBX * 2 + q) * 1024,
                                                                                               (CF_X3_NT & 2) ? 2 : 0));
    };
    int tile = blockIdx.x * nwaves + wave;
    // x_0 and x_1 of the first tile travel while the workgroup stages its weights
    if (tile < n_tiles) {
#pragma unroll
        for (int q = 0; q < 2 * KBX; ++q) { load_x(x0, x_rsrc(tile), 0, q); load_x(x1, x_rsrc(tile), 1, q); }
    }
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wpack + (size_t)dir * PACK);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < PACK / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int hh = lane >> 5;
    const far_lds<bf16x8> WAF(reinterpret_cast<const bf16x8*>(lds) + lane);     // + (product * 2 + part) * 64
    const f32x4* BI = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + gb_bias_off(CIN, 2)) + hh * 4;
    f32x16 bias[6];                                         // CF_X3_BIAS_C: the C operand of the MFMA that opens an accumulator
    if constexpr (CF_X3_BIAS_C) {
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 b = BI[mt * 8 + c4];
                bias[mt][4 * c4 + 0] = b.x; bias[mt][4 * c4 + 1] = b.y; bias[mt][4 * c4 + 2] = b.z; bias[mt][4 * c4 + 3] = b.w;
            }
    }
    float dw[32];                                           // LAST: this lane's rows of final_fully_connected/kernel
    if constexpr (LAST) {
        const float* DWp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + gb_dense_off(CIN, 2)) + hh * 16;
#pragma unroll
        for (int e = 0; e < 32; ++e) dw[e] = DWp[(e >> 4) * 32 + (e & 15)];
    }

    for (; tile < n_tiles; tile += stride) {
        const __amdgpu_buffer_rsrc_t rx = x_rsrc(tile);
        const __amdgpu_buffer_rsrc_t rx_next = x_rsrc(tile + stride < n_tiles ? tile + stride : tile);      // (no next tile: a harmless re-read)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(Y) + (int64_t)tile * Y_TILE_BYTES, 0,
                                                                            LAST ? 0 : Y_TILE_BYTES, 0x00020000);
        f32x16 acc[6];                                      // r: 0,1   u: 2,3   c: 4,5
        f32x16 h[2], u[2];
        u32x4 hp[4][2], rp[4][2];                           // bf16 hi / lo fragments of h and r*h per k-block, carried as dwords
        bf16x8 ar[DA][2];
        float ev[32], fv[32];                               // stage registers of the activation chains
        float pl = 0.f;
        long long st_[3] = {0, 0, 0};
        const long long st_begin = CF_X3_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const long long rt_begin = CF_X3_STAMP ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) { h[m][i] = 0.f; u[m][i] = 0.f; }     // GRUCellZeroState
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int p = 0; p < 2; ++p) { hp[kb][p] = (u32x4){0u, 0u, 0u, 0u}; rp[kb][p] = (u32x4){0u, 0u, 0u, 0u}; }

        auto load_bias = [&](int q) {                                      // q = mt * 4 + c4
            const f32x4 b = BI[(q >> 2) * 8 + (q & 3)];
            const int mt = q >> 2, c4 = q & 3;
            acc[mt][4 * c4 + 0] = b.x; acc[mt][4 * c4 + 1] = b.y; acc[mt][4 * c4 + 2] = b.z; acc[mt][4 * c4 + 3] = b.w;
        };
        // one MFMA of the step's sequence (gap g = 3 * product + {hi*hi, hi*lo, lo*hi}) and the ring refills behind it.  The gap
        // index is a template argument (std::integral_constant): every table look-up below is a constant expression and every
        // register array index a literal -- nothing is left to the unroller.
        // FINAL (step 34): phase C has no next step to prepare, its MFMAs and every ring refill past them are left out.
        auto mfma_gap = [&](auto g_, auto final_, bf16x8 (&xa)[KBX][2], bf16x8 (&xb)[KBX][2]) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            constexpr bool FINAL = decltype(final_)::value;
            constexpr int p = g / 3, sub = g - 3 * p;
            if constexpr (FINAL && p >= G::OC) return;
            constexpr typename G::prod_t d = G::prod(p);
            constexpr int part = sub == 1 ? 1 : 0;                         // B: hi, lo, hi
            bf16x8 b;
            if constexpr (d.src == G::SRC_HP) b = __builtin_bit_cast(bf16x8, hp[d.kb][part]);
            else if constexpr (d.src == G::SRC_RP) b = __builtin_bit_cast(bf16x8, rp[d.kb][part]);
            else if constexpr (d.src == G::SRC_XA) b = xa[d.kb][part];
            else b = xb[d.kb][part];
            const bf16x8 a = ar[p % DA][sub == 2 ? 1 : 0];                 // A: hi, hi, lo   (prod<2>'s order)
            constexpr bool opens = CF_X3_BIAS_C && sub == 0 && G::first_prod(d.mt) == p;      // bias + ... : the accumulation starts here
            if constexpr (CF_X3_ABL & 2) asm volatile("" ::"v"(a), "v"(b));
            else if constexpr (opens) acc[d.mt] = MFMA32B(a, b, bias[d.mt]);
            else acc[d.mt] = MFMA32B(a, b, acc[d.mt]);
            // the hi fragment is dead after the second MFMA, the lo fragment after the third
            if constexpr (sub == 1 && !(FINAL && p + DA >= G::OC) && !(CF_X3_ABL & 4)) ar[p % DA][0] = WAF[(G::prod((p + DA) % NSEQ).frag * 2 + 0) * 64];
            if constexpr (sub == 2 && !(FINAL && p + DA >= G::OC) && !(CF_X3_ABL & 4)) ar[p % DA][1] = WAF[(G::prod((p + DA) % NSEQ).frag * 2 + 1) * 64];
        };
        // vector micro-op q of gap g (x3::make_sched decides which)
        auto vop = [&](auto g_, auto q_, auto final_, bf16x8 (&xa)[KBX][2], bf16x8 (&xb)[KBX][2], int s, int t) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, q = decltype(q_)::value;
            constexpr bool FINAL = decltype(final_)::value;
            if constexpr (q < T::S.n[g]) {
                constexpr int code = T::S.op[g][q];
                constexpr int kind = code >> 8, i = code & 255;
                constexpr int m = (i >> 4) & 1, r = i & 15;                // element ops: M-tile, register
                constexpr int kb = (i >> 2) & 3, w = i & 3;                // pair ops: k-block, dword
                constexpr int m2 = ((2 * i) >> 4) & 1, r2 = (2 * i) & 15;  // pair ops: M-tile, first register
                constexpr bool vec = !(CF_X3_ABL & 1);
                if constexpr (kind == x3::LX) {
                    if constexpr (!FINAL) load_x(xa, rx, s + 2, i);
                    else { load_x(xa, rx_next, 0, i); load_x(xb, rx_next, 1, i); }          // FINAL: xa = x0, xb = x1 (idle: no phase C MFMAs)
                }
                else if constexpr (kind == x3::LB) { if constexpr (!(FINAL && i < 16)) load_bias(i); }     // (acc_r, acc_u of a next step)
                else if constexpr (kind == x3::CS) {
                    if constexpr (!LAST)
                        __builtin_amdgcn_raw_buffer_store_b128(hp[i >> 1][i & 1], ry, lane16, (((t * 8 + dir * 4 + (i >> 1)) * 2) + (i & 1)) * 1024,
                                                               (CF_X3_NT & 1) ? 2 : 0);
                }
                else if constexpr (!vec) {}
                // Every result is pinned (x3_pin) to the gap it was computed in: without that the SLP vectoriser pairs the scalar
                // operations that feed a v_cvt_pk_bf16_f32 into v_pk_* instructions and drags whole chains into one gap.
                // r = sigmoid(acc_r); r*h -> bf16 hi + lo (reset BEFORE the candidate matmul: gru_cell/mul -> concat_1)
                else if constexpr (kind == x3::AE) ev[i] = x3_pin(__builtin_amdgcn_exp2f(acc[m][r]));
                else if constexpr (kind == x3::AR1) ev[i] = x3_pin(ev[i] + 1.0f);
                else if constexpr (kind == x3::AR2) ev[i] = x3_pin(__builtin_amdgcn_rcpf(ev[i]));
                else if constexpr (kind == x3::AM) ev[i] = x3_pin(ev[i] * h[m][r]);
                else if constexpr (kind == x3::AP1) rp[kb][0][w] = x3_pin(cvt_pk_bf16(ev[2 * i], ev[2 * i + 1]));
                else if constexpr (kind == x3::AP2) {
                    fv[2 * i] = x3_pin(__builtin_bit_cast(float, rp[kb][0][w] << 16));
                    fv[2 * i + 1] = x3_pin(__builtin_bit_cast(float, rp[kb][0][w] & 0xffff0000u));
                }
                else if constexpr (kind == x3::AP3) { ev[2 * i] = x3_pin(ev[2 * i] - fv[2 * i]); ev[2 * i + 1] = x3_pin(ev[2 * i + 1] - fv[2 * i + 1]); }
                else if constexpr (kind == x3::AP4) rp[kb][1][w] = x3_pin(cvt_pk_bf16(ev[2 * i], ev[2 * i + 1]));
                // u = sigmoid(acc_u)
                else if constexpr (kind == x3::BE) ev[i] = x3_pin(__builtin_amdgcn_exp2f(acc[2 + m][r]));
                else if constexpr (kind == x3::BR1) ev[i] = x3_pin(ev[i] + 1.0f);
                else if constexpr (kind == x3::BR2) u[m][r] = x3_pin(__builtin_amdgcn_rcpf(ev[i]));
                // c = tanh(acc_c); h' = u*h + (1-u)*c      (gru_cell/mul_1, sub, mul_2, add); h' -> bf16 hi + lo
                else if constexpr (kind == x3::CE) ev[i] = x3_pin(__builtin_amdgcn_exp2f(acc[4 + m][r]));
                else if constexpr (kind == x3::CR1) ev[i] = x3_pin(ev[i] + 1.0f);
                else if constexpr (kind == x3::CR2) ev[i] = x3_pin(__builtin_amdgcn_rcpf(ev[i]));
                else if constexpr (kind == x3::CC) ev[i] = x3_pin(fmaf(-2.0f, ev[i], 1.0f));
                else if constexpr (kind == x3::CD) fv[i] = x3_pin(h[m][r] - ev[i]);
                else if constexpr (kind == x3::CH) h[m][r] = x3_pin(fmaf(u[m][r], fv[i], ev[i]));
                else if constexpr (kind == x3::CL) pl = x3_pin(fmaf(dw[i], h[m][r], pl));   // partial logit of this direction (final_fully_connected/MatMul)
                else if constexpr (kind == x3::CP1) hp[kb][0][w] = x3_pin(cvt_pk_bf16(h[m2][r2], h[m2][r2 + 1]));
                else if constexpr (kind == x3::CP2) {
                    fv[2 * i] = x3_pin(__builtin_bit_cast(float, hp[kb][0][w] << 16));
                    fv[2 * i + 1] = x3_pin(__builtin_bit_cast(float, hp[kb][0][w] & 0xffff0000u));
                }
                else if constexpr (kind == x3::CP3) { ev[2 * i] = x3_pin(h[m2][r2] - fv[2 * i]); ev[2 * i + 1] = x3_pin(h[m2][r2 + 1] - fv[2 * i + 1]); }
                else if constexpr (kind == x3::CP4) hp[kb][1][w] = x3_pin(cvt_pk_bf16(ev[2 * i], ev[2 * i + 1]));
            }
        };

        // ---- prologue (x_0 and x_1 are on their way since the last step of the tile before): the MFMAs of phase C of an
        //      imaginary step -1 on x_0 and h = 0:  acc_r = b_r + Wx_r x_0, acc_u = b_u + the first XUC products of Wx_u x_0
#pragma unroll
        for (int q = 0; q < DA; ++q) {
            ar[(G::OC + q) % DA][0] = WAF[(G::prod(G::OC + q).frag * 2 + 0) * 64];
            ar[(G::OC + q) % DA][1] = WAF[(G::prod(G::OC + q).frag * 2 + 1) * 64];
        }
        if constexpr (!CF_X3_BIAS_C) {
#pragma unroll
            for (int q = 0; q < 16; ++q) load_bias(q);
        }
        x3_for(std::make_integer_sequence<int, G::GC>{}, [&](auto k_) __attribute__((always_inline)) {
            mfma_gap(std::integral_constant<int, 3 * G::OC + decltype(k_)::value>{}, std::false_type{}, x1, x0);
            __builtin_amdgcn_sched_barrier(0);
        });

        const long long st_pro = CF_X3_STAMP ? (long long)__builtin_amdgcn_s_memtime() - st_begin : 0;

        // one step: xa holds x_s (refilled with x_{s+2} in C), xb holds x_{s+1}
        auto step = [&](auto final_, bf16x8 (&xa)[KBX][2], bf16x8 (&xb)[KBX][2], int s) {
            constexpr bool FINAL = decltype(final_)::value;
            const int t = t0 + s * tstep;
            long long ta_ = CF_X3_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
            x3_for(std::make_integer_sequence<int, NGAP>{}, [&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                mfma_gap(g_, final_, xa, xb);
                x3_for(std::make_integer_sequence<int, T::S.n[g]>{}, [&](auto q_) __attribute__((always_inline)) { vop(g_, q_, final_, xa, xb, s, t); });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CF_X3_STAMP && (g == G::GA - 1 || g == G::GA + G::GB - 1 || g == NGAP - 1)) {
                    const long long tb_ = (long long)__builtin_amdgcn_s_memtime();
                    st_[g == G::GA - 1 ? 0 : (g == NGAP - 1 ? 2 : 1)] += tb_ - ta_;
                    ta_ = tb_;
                }
            });
            if (CF_X3_ABL) {        // timing-only variants: keep every accumulator and operand alive
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) asm volatile("" ::"v"(acc[mt]));
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) asm volatile("" ::"v"(hp[kb][0]), "v"(hp[kb][1]), "v"(rp[kb][0]), "v"(rp[kb][1]));
            }
            if constexpr (LAST) {
                const float p2 = pl + __shfl_xor(pl, 32);
                if (lane < 32 && !CF_X3_STAMP) P[(((int64_t)dir * n_tiles + tile) * CF_T + t) * 32 + lane] = p2;
                pl = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int s = 0; s < CF_T - 1; s += 2) {
            step(std::false_type{}, x0, x1, s);
            step(std::false_type{}, x1, x0, s + 1);
        }
        step(std::true_type{}, x0, x1, CF_T - 1);
        if (CF_X3_STAMP && !LAST && lane == 0) {
            const long long st_end = (long long)__builtin_amdgcn_s_memtime();
            long long* o = reinterpret_cast<long long*>(P) + ((int64_t)dir * n_tiles + tile) * 8;
            o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = (long long)__builtin_amdgcn_s_memrealtime() - rt_begin; o[4] = st_end - st_begin; o[5] = st_begin; o[6] = st_pro;
            o[7] = ((long long)blockIdx.x << 8) | wave;
        }
    }
}
