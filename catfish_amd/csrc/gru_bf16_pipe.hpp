// gru_bf16_pipe.hpp -- biGRU layer in plain bf16 (NP = 1, BASELINE config 4), software-pipelined so that every
// matrix instruction has vector work of the SAME wave issued behind it.  Included by catfish_hip.hip after
// gru_bf16.hpp (same weight blob, same activation layout, same arithmetic per element).
//
// Why: gru_layer_bf16_kernel runs a step as  [72 MFMAs] -> [r: 64 transcendentals] -> [8 MFMAs] -> [u, c, h: 128
// transcendentals]; the PMC counters of round 1 show MFMA-busy 55 % + VALU-active 37 % with almost no overlap (the two
// waves of a SIMD run the same program nearly in lockstep).  v_mfma_f32_32x32x16_bf16 executes on the matrix pipe for
// 32 cycles and holds the SIMD's vector issue for only 8 of them, so the sigmoid/tanh work can ride in those gaps IF it
// is independent of the MFMA in flight.  The data dependences of a GRU step allow exactly that once the h-independent
// x projection of step s+1 is moved next to the activation arithmetic of step s:
//
//   Q1  acc_u  = b_u + Wh_u h + Wx_u x_s     8 + 2 KBX MFMA   ||  r = sigmoid(acc_r), rp = bf16(r * h)
//   Q2  acc_c  = b_c + Wx_c x_s + Wh_c rp    2 KBX + 8 MFMA   ||  u = sigmoid(acc_u)
//   Q3  acc_r  = b_r + Wx_r x_{s+1}          2 KBX MFMA       ||  c = tanh(acc_c), h = c + u (h - c), hp = bf16(h)
//       acc_r += Wh_r h'                     8 MFMA               (k-block kb of h' is issued once its 8 elements are done)
//
// (acc_r of step s+1 is complete at the end of Q3 of step s.)  The transcendental unit runs beside the matrix pipe
// (tools/ubench_gap.hip: up to 3 v_exp / v_rcp per MFMA cost 1-5 cycles, plain fp32 VALU adds its full issue time), so
// the 192 transcendentals of a step are spread over all 72 MFMA slots: 2.67 per slot.
// Each MFMA is followed in program order by the ds_read of an A fragment DA entries ahead and by a slice of the
// phase's vector work; a sched_barrier after every such slot keeps hipcc from regrouping them.  x_s sits in one of two
// register buffers; the buffer of x_s is refilled with x_{s+2} as soon as its last MFMA (P3) has issued, a full step
// ahead of its first use.  Accumulation order differs from gru_layer_bf16_kernel (bias + x part of a gate may be added
// after its h part), which changes fp32 rounding only.
#pragma once

// Timing-only ablations of the pipelined kernel (tools/ablate_pipe.sh builds the variants; results are WRONG when set):
//   bit 0: no activation arithmetic   bit 1: no MFMAs   bit 2: no A-fragment LDS reads   bit 3: no global loads / stores
#ifndef CF_PIPE_ABL
#define CF_PIPE_ABL 0
#endif
// Diagnostic build: s_memtime stamps at the phase boundaries of every step, summed per wave and written to the (otherwise
// unused) dense-partial buffer P of the non-LAST kernels as int64 [wave][8] = {P1, P2, P3, P4 cycles, total, start, end, id}
#ifndef CF_PIPE_STAMP
#define CF_PIPE_STAMP 0
#endif
// A/B: non-temporal hints on the streamed layer input loads (bit 0) and layer output stores (bit 1)
#ifndef CF_PIPE_NT
#define CF_PIPE_NT 0
#endif
// A/B variants of the vector work (bit-identical results): bit 0 = the update gate's sigmoid is written straight into its
// accumulator registers (no copies in finish_u); bit 1 = scalar fp32 instructions instead of packed ones; bit 2 = static
// priority for the second-dispatched half of the workgroup (waves 4-7, the arbitration losers of every SIMD)
#ifndef CF_PIPE_VAR
#define CF_PIPE_VAR 0
#endif

template <int CIN>
struct gb_pipe {
    static constexpr int KBX = CIN / 16;
    static constexpr int NX = KBX * 6, NG = 16;
    static constexpr int NQ = 2 * KBX + 8;                  // MFMA slots per phase (three phases per step)
    static constexpr int O2 = NQ, O3 = 2 * NQ, NSEQ = 3 * NQ;
    // blob fragment consumed by slot i of a step (order of gru_bf16.hpp's blob: x [kb][mt<6] | gates h [kb][mt<4] | cand h [kb][mt<2])
    static __host__ __device__ constexpr int frag(int i) {
        if (i < 8) return NX + (i >> 1) * 4 + 2 + (i & 1);                               // Q1: h part of u   (M-tiles 2,3)
        if (i < O2) return ((i - 8) >> 1) * 6 + 2 + ((i - 8) & 1);                       // Q1: x part of u
        if (i < O2 + 2 * KBX) return ((i - O2) >> 1) * 6 + 4 + ((i - O2) & 1);           // Q2: x part of c   (M-tiles 4,5)
        if (i < O3) return NX + NG + ((i - O2 - 2 * KBX) >> 1) * 2 + ((i - O2 - 2 * KBX) & 1);   // Q2: h part of c (on r*h)
        if (i < O3 + 2 * KBX) return ((i - O3) >> 1) * 6 + ((i - O3) & 1);               // Q3: x part of r, NEXT step (M-tiles 0,1)
        return NX + ((i - O3 - 2 * KBX) >> 1) * 4 + ((i - O3 - 2 * KBX) & 1);            // Q3: h part of r, NEXT step, k-block major
    }
    // Q3: pair j of the h update starts behind slot ps3(j): its k-block 4-tuple is finished (slot + 2) before the h-part MFMA
    // of that k-block (slot 2 KBX + 2 kb) issues
    static __host__ __device__ constexpr int ps3(int j) { return (j * (2 * KBX + 4)) / 16; }
    static __host__ __device__ constexpr int ps(int j) { return (j * (NQ - 2)) / 16; }   // Q1, Q2: spread over the whole phase
};

template <int CIN, bool LAST>
__global__ __launch_bounds__(512, 2) void gru_bf16_pipe_kernel(const char* __restrict__ wpack,   // [2][gb_pack_bytes(CIN, 1)]
                                                               const bf16x8* __restrict__ X,      // [tile32][t][KBX][lane]
                                                               bf16x8* __restrict__ Y,            // [tile32][t][8][lane]
                                                               float* __restrict__ P,             // [2][tile][t][32]
                                                               int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = gb_pipe<CIN>;
    constexpr int KBX = G::KBX;
    constexpr int NSEQ = G::NSEQ;
    constexpr int DA = 4;                                   // A-fragment ring depth (slots ahead)
    constexpr int PACK = gb_pack_bytes(CIN, 1);
    static_assert(NSEQ % DA == 0 && G::O3 % DA == 0, "ring depth must divide the schedule and the prologue offset");

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
    const far_lds<bf16x8> WAF(reinterpret_cast<const bf16x8*>(lds) + lane);
    const f32x4* BI = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + gb_bias_off(CIN, 1)) + hh * 4;
    const f32x4* DW = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + gb_dense_off(CIN, 1)) + hh * 4;

    const int nwaves = blockDim.x >> 6;
    if ((CF_PIPE_VAR & 4) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int tstep = dir ? -1 : 1;                         // bw = reversed time
    const int t0 = dir ? (CF_T - 1) : 0;
    for (int tile = blockIdx.x * nwaves + wave; tile < n_tiles; tile += gridDim.x * nwaves) {
        f32x16 acc[6];                                      // r: 0,1   u: 2,3   c: 4,5
        f32x16 h[2];
        bf16x8 hp[4], rp[4];
        bf16x8 ar[DA];
        bf16x8 x0[KBX], x1[KBX];
        float pl = 0.f;
        long long st_[5] = {0, 0, 0, 0, 0};
        const long long st_begin = CF_PIPE_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) h[m][i] = 0.f;     // GRUCellZeroState
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) { hp[kb][j] = (__bf16)0.f; rp[kb][j] = (__bf16)0.f; }

        auto load_x = [&](bf16x8 (&dst)[KBX], int s, int kb) {
            s = s > CF_T - 1 ? CF_T - 1 : s;                // past the end: harmless re-read
            if ((CF_PIPE_ABL & 8) && s > 1) return;
            const bf16x8* src = X + (((int64_t)tile * CF_T + (t0 + s * tstep)) * KBX + kb) * 64 + lane;
            if (CF_PIPE_NT & 1) dst[kb] = __builtin_nontemporal_load(src); else dst[kb] = *src;
        };
        auto load_bias = [&](int mt) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 b = BI[mt * 8 + c4];
                acc[mt][4 * c4 + 0] = b.x; acc[mt][4 * c4 + 1] = b.y; acc[mt][4 * c4 + 2] = b.z; acc[mt][4 * c4 + 3] = b.w;
            }
        };
        // one MFMA slot: consume ring entry i, refill it DA slots ahead
#define CF_SLOT(i, B, MT)                                                                 \
    {                                                                                     \
        const bf16x8 a_ = ar[(i) % DA];                                                   \
        if (!(CF_PIPE_ABL & 4)) ar[(i) % DA] = WAF[G::frag(((i) + DA) % NSEQ) * 64];      \
        if (!(CF_PIPE_ABL & 2)) acc[MT] = MFMA32B(a_, B, acc[MT]);                        \
        else asm volatile("" ::"v"(a_), "v"(B));                                          \
    }
        // Vector work units: element pair j (0..15) of a gate = registers 2(j&7), 2(j&7)+1 of M-tile j>>3, i.e. elements
        // 2(j&3), +1 of k-block j>>2.  Every pair is a dependent chain exp2 -> +1 -> rcp -> finish; its three stages go
        // behind three CONSECUTIVE MFMA slots (E, R, F), so that a slot holds stages of up to three different pairs and
        // no instruction in it waits for the one before (a wave issues in order: a chain kept in one slot stalls it).
        f32x2 ev[16];                                       // stage registers (live across two slots only)
        auto stage_e = [&](int j, int gate) {
            if (CF_PIPE_ABL & 1) return;
            const int m = j >> 3, i = 2 * (j & 7);
            ev[j] = (f32x2){__builtin_amdgcn_exp2f(acc[2 * gate + m][i]), __builtin_amdgcn_exp2f(acc[2 * gate + m][i + 1])};
        };
        auto stage_r = [&](int j, int gate) {
            if (CF_PIPE_ABL & 1) return;
            f32x2 d;
            if (CF_PIPE_VAR & 2) { d.x = ev[j].x + 1.0f; d.y = ev[j].y + 1.0f; }
            else d = ev[j] + (f32x2){1.0f, 1.0f};
            if ((CF_PIPE_VAR & 1) && gate == 1) {           // u: the pre-activation is dead, the gate value replaces it in place
                const int m = j >> 3, i = 2 * (j & 7);
                acc[2 + m][i] = __builtin_amdgcn_rcpf(d.x);
                acc[2 + m][i + 1] = __builtin_amdgcn_rcpf(d.y);
            } else {
                ev[j] = (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            }
        };
        auto finish_r = [&](int j) {                        // r*h -> bf16 (reset BEFORE the candidate matmul: gru_cell/mul -> concat_1)
            if (CF_PIPE_ABL & 1) return;
            const int m = j >> 3, i = 2 * (j & 7);
            f32x2 rh;
            if (CF_PIPE_VAR & 2) { rh.x = ev[j].x * h[m][i]; rh.y = ev[j].y * h[m][i + 1]; }
            else rh = ev[j] * (f32x2){h[m][i], h[m][i + 1]};
            rp[j >> 2][2 * (j & 3)] = (__bf16)rh.x;
            rp[j >> 2][2 * (j & 3) + 1] = (__bf16)rh.y;
        };
        auto finish_u = [&](int j) {
            if ((CF_PIPE_ABL & 1) != 0 || (CF_PIPE_VAR & 1) != 0) return;
            const int m = j >> 3, i = 2 * (j & 7);
            acc[2 + m][i] = ev[j].x;
            acc[2 + m][i + 1] = ev[j].y;
        };
        auto finish_h = [&](int j) {                        // h' = u*h + (1-u)*c      (gru_cell/mul_1, sub, mul_2, add)
            if (CF_PIPE_ABL & 1) return;
            const int m = j >> 3, i = 2 * (j & 7);
            f32x2 c, hn;
            const f32x2 u = {acc[2 + m][i], acc[2 + m][i + 1]};
            if (CF_PIPE_VAR & 2) {
                c.x = fmaf(-2.0f, ev[j].x, 1.0f); c.y = fmaf(-2.0f, ev[j].y, 1.0f);
                hn.x = fmaf(u.x, h[m][i] - c.x, c.x); hn.y = fmaf(u.y, h[m][i + 1] - c.y, c.y);
            } else {
                c = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, ev[j], (f32x2){1.0f, 1.0f});
                hn = __builtin_elementwise_fma(u, (f32x2){h[m][i], h[m][i + 1]} - c, c);
            }
            h[m][i] = hn.x;
            h[m][i + 1] = hn.y;
            hp[j >> 2][2 * (j & 3)] = (__bf16)hn.x;
            hp[j >> 2][2 * (j & 3) + 1] = (__bf16)hn.y;
            if constexpr (LAST) {                           // partial logit of this direction (final_fully_connected/MatMul), fp32 h
                const float* dw = reinterpret_cast<const float*>(DW + m * 8);
                pl = fmaf(dw[i], hn.x, pl);
                pl = fmaf(dw[i + 1], hn.y, pl);
            }
        };
        // pair j of a 16-pair gate starts (stage E) behind slot PS(j) of its phase; R and F follow in the next two slots
#define CF_STAGES(k, PS, GATE, FIN)                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                   \
        if (PS(j_) + 2 == (k)) FIN(j_);                                                   \
        if (PS(j_) + 1 == (k)) stage_r(j_, GATE);                                         \
        if (PS(j_) == (k)) stage_e(j_, GATE);                                             \
    }

        // ---- prologue: x_0, x_1, acc_r = b_r + Wx_r x_0 (+ Wh_r 0): the MFMA part of Q3 of an imaginary step -1
#pragma unroll
        for (int kb = 0; kb < KBX; ++kb) { load_x(x0, 0, kb); load_x(x1, 1, kb); }
#pragma unroll
        for (int q = 0; q < DA; ++q) ar[q] = WAF[G::frag(G::O3 + q) * 64];
        load_bias(0);
        load_bias(1);
#pragma unroll
        for (int k = 0; k < G::NQ; ++k) {
            if (k < 2 * KBX) { CF_SLOT(G::O3 + k, x0[k >> 1], k & 1); }
            else { CF_SLOT(G::O3 + k, hp[(k - 2 * KBX) >> 1], (k - 2 * KBX) & 1); }
            __builtin_amdgcn_sched_barrier(0);
        }

        // one step: xa holds x_s (refilled with x_{s+2} in Q2), xb holds x_{s+1}
        auto step = [&](bf16x8 (&xa)[KBX], bf16x8 (&xb)[KBX], int s) {
            const int t = t0 + s * tstep;
            long long ta_ = CF_PIPE_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define CF_STAMP(idx) if (CF_PIPE_STAMP) { const long long tb_ = (long long)__builtin_amdgcn_s_memtime(); st_[idx] += tb_ - ta_; ta_ = tb_; }
            load_bias(2); load_bias(3); load_bias(4); load_bias(5);
            // Q1: u gate (h part first: hp is final, then the x part) || r = sigmoid(.), r*h -> bf16      (gru_cell/MatMul)
#pragma unroll
            for (int k = 0; k < G::NQ; ++k) {
                if (k < 8) { CF_SLOT(k, hp[k >> 1], 2 + (k & 1)); }
                else { CF_SLOT(k, xa[(k - 8) >> 1], 2 + (k & 1)); }
                CF_STAGES(k, G::ps, 0, finish_r);
                __builtin_amdgcn_sched_barrier(0);
            }
            CF_STAMP(0);
            // Q2: candidate (x part, then the h part on r*h) || u = sigmoid(.); xa is dead after its last MFMA: refill
#pragma unroll
            for (int k = 0; k < G::NQ; ++k) {
                if (k < 2 * KBX) { CF_SLOT(G::O2 + k, xa[k >> 1], 4 + (k & 1)); }
                else { CF_SLOT(G::O2 + k, rp[(k - 2 * KBX) >> 1], 4 + (k & 1)); }
                if (k < 2 * KBX && (k & 1)) load_x(xa, s + 2, k >> 1);
                if (k == 0) load_bias(0);                   // acc_r is free since the end of Q1
                if (k == 1) load_bias(1);
                CF_STAGES(k, G::ps, 1, finish_u);
                __builtin_amdgcn_sched_barrier(0);
            }
            CF_STAMP(1);
            // Q3: r gate of the NEXT step (x part, then the h part k-block by k-block as h' completes) || c = tanh(.), h update
#pragma unroll
            for (int k = 0; k < G::NQ; ++k) {
                if (k < 2 * KBX) { CF_SLOT(G::O3 + k, xb[k >> 1], k & 1); }
                else { CF_SLOT(G::O3 + k, hp[(k - 2 * KBX) >> 1], (k - 2 * KBX) & 1); }
                CF_STAGES(k, G::ps3, 2, finish_h);
                if constexpr (!LAST) {
                    // k-block kb of hp is complete once pair 4 kb + 3 has finished (behind slot ps3(4 kb + 3) + 2)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
                        if (G::ps3(4 * kb + 3) + 2 == k && (!(CF_PIPE_ABL & 8) || s == CF_T - 1)) {
                            bf16x8* dst = Y + (((int64_t)tile * CF_T + t) * 8 + dir * 4 + kb) * 64 + lane;
                            if (CF_PIPE_NT & 2) __builtin_nontemporal_store(hp[kb], dst); else *dst = hp[kb];
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            CF_STAMP(2);
#undef CF_STAMP
            if (CF_PIPE_ABL) {      // timing-only variants: keep every accumulator and operand alive
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) asm volatile("" ::"v"(acc[mt]));
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) asm volatile("" ::"v"(hp[kb]), "v"(rp[kb]));
            }
            if constexpr (LAST) {
                float p2 = pl + __shfl_xor(pl, 32);
                if (lane < 32 && !CF_PIPE_STAMP) P[(((int64_t)dir * n_tiles + tile) * CF_T + t) * 32 + lane] = p2;
                pl = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int s = 0; s < CF_T - 1; s += 2) {
            step(x0, x1, s);
            step(x1, x0, s + 1);
        }
        step(x0, x1, CF_T - 1);
        if (CF_PIPE_STAMP && !LAST && lane == 0) {
            const long long st_end = (long long)__builtin_amdgcn_s_memtime();
            long long* o = reinterpret_cast<long long*>(P) + ((int64_t)dir * n_tiles + tile) * 8;
            o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = st_end - st_begin; o[5] = st_begin; o[6] = st_end;
            o[7] = ((long long)blockIdx.x << 8) | wave;
        }
#undef CF_SLOT
#undef CF_STAGES
    }
}
