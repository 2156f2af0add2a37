// Any-size inference path: the model classes take layer_size / layer_size_res from {16, 32, 64, 128, 256} and any depth
// (networks/train_validate.py:70-78, catfish/models/rnn_class.py:13-18, resnet_class.py:9-14); the tuned kernels of
// catfish_hip.hip are specialised for the shipped checkpoint (64 GRU units, 32 conv channels).  Every other geometry runs
// here: the same fragment layout [tile][t][feature tile][lane][4] and the same fp32 MFMA (v_mfma_f32_16x16x4_f32, the
// previous layer's D tile fed back as the B operand), but with run-time sizes -- weights stream from global memory (L2 /
// Infinity Cache resident: at most 2.4 MB per direction and layer) as pre-tiled A fragments instead of living in LDS, the
// biGRU state of a tile (h, r.h, h') lives in LDS, and one launch computes one conv or one biGRU layer.
// Included by catfish_hip.hip (needs f32x4, MFMA16, CF_T, CF_TILE, CF_GATE_SCALE, CF_CAND_SCALE).
#pragma once

#ifndef CF_GEN_LOCKSTEP
#define CF_GEN_LOCKSTEP 1      // barrier per output-tile group in the biGRU kernels (see gen_gru_kernel)
#endif

// A pack of a matrix with K inputs (K16 blocks of 16) and M outputs (M16 tiles): f32x4 P[(mo * K16 + kb) * 64 + lane] =
// W[in = 16 kb + 4 (lane >> 4) + i][out = 16 mo + (lane & 15)], i = 0..3: component i is the A operand of the MFMA whose B
// operand is register i of input tile kb (lane quarter q of that register carries feature 16 kb + 4 q + i).

// Two output tiles over one K sequence: acc0 += A0 . B0, acc1 += A1 . B1, K = [KBX blocks whose B operand comes from global
// memory (xb) | H16 blocks whose B operands come from LDS (hb0 for acc0, hb1 for acc1)].  The A fragments stream from L2 and
// the loop is software-pipelined through a ring of CF_GEN_DEPTH k-blocks: block k + DEPTH is requested right after block k's
// eight MFMAs have been issued, i.e. DEPTH x 256 MFMA cycles ahead of its use.
#ifndef CF_GEN_DEPTH
#define CF_GEN_DEPTH 4
#endif
#ifndef CF_GEN_STAMP
#define CF_GEN_STAMP 0        // diagnostic build: per-phase s_memtime cycles of one wave of gen_gru2_kernel, printed at the end
#endif
#ifndef CF_GEN_ABL
#define CF_GEN_ABL 0          // timing-only ablations (tools/): 1 = no weight refills, 2 = no B-operand loads
#endif
// Load through a UNIFORM pointer plus the lane's byte offset.  The empty asm makes the lane term opaque per use: otherwise the
// compiler folds it into a loop-invariant per-lane base pointer and adds the (scalar) k offset to it with 64-bit VALU
// arithmetic for every load; this way the address stays an SGPR base + one VGPR offset (global_load ... v, s[base]).
__device__ __forceinline__ f32x4 gen_ld(const f32x4* __restrict__ p, unsigned ln) {
    unsigned off = ln * 16u;
    asm volatile("" : "+v"(off));
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p) + off);
}

__device__ __forceinline__ void gen_dot(f32x4& acc0, f32x4& acc1, const f32x4* __restrict__ wa, const f32x4* __restrict__ wb,
                                        const f32x4* __restrict__ xb, int KBX, const f32x4* hb0, const f32x4* hb1, int H16, unsigned ln) {
    constexpr int D = CF_GEN_DEPTH;
    const int KB = KBX + H16;
    f32x4 A0[D], A1[D], B0[D], B1[D];
    auto fetch = [&](int j, int k) {
        A0[j] = gen_ld(wa + k * 64, ln);
        A1[j] = gen_ld(wb + k * 64, ln);
        if (k < KBX) { B0[j] = gen_ld(xb + k * 64, ln); B1[j] = B0[j]; }
        else { B0[j] = hb0[(k - KBX) * 64]; B1[j] = hb1[(k - KBX) * 64]; }
    };
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < KB) fetch(j, j);
    for (int k0 = 0; k0 < KB; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = k0 + j;
            if (k < KB) {
                const f32x4 a0 = A0[j], a1 = A1[j], b0 = B0[j], b1 = B1[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = MFMA16(a0[i], b0[i], acc0);
                    acc1 = MFMA16(a1[i], b1[i], acc1);
                }
                if (k + D < KB) fetch(j, k + D);
            }
        }
    }
}

// The same with four output tiles (four independent MFMA chains): acc0, acc2 use B0 and acc1, acc3 use B1.
__device__ __forceinline__ void gen_dot4_any(f32x4& acc0, f32x4& acc1, f32x4& acc2, f32x4& acc3, const f32x4* __restrict__ w0,
                                             const f32x4* __restrict__ w1, const f32x4* __restrict__ w2, const f32x4* __restrict__ w3,
                                         const f32x4* __restrict__ xb, int KBX, const f32x4* hb0, const f32x4* hb1, int H16, unsigned ln) {
    constexpr int D = CF_GEN_DEPTH;
    const int KB = KBX + H16;
    f32x4 A0[D], A1[D], A2[D], A3[D], B0[D], B1[D];
    auto fetch = [&](int j, int k) {
        A0[j] = gen_ld(w0 + k * 64, ln);
        A1[j] = gen_ld(w1 + k * 64, ln);
        A2[j] = gen_ld(w2 + k * 64, ln);
        A3[j] = gen_ld(w3 + k * 64, ln);
        if (k < KBX) { B0[j] = gen_ld(xb + k * 64, ln); B1[j] = B0[j]; }
        else { B0[j] = hb0[(k - KBX) * 64]; B1[j] = hb1[(k - KBX) * 64]; }
    };
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < KB) fetch(j, j);
    for (int k0 = 0; k0 < KB; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = k0 + j;
            if (k < KB) {
                const f32x4 a0 = A0[j], a1 = A1[j], a2 = A2[j], a3 = A3[j], b0 = B0[j], b1 = B1[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = MFMA16(a0[i], b0[i], acc0);
                    acc1 = MFMA16(a1[i], b1[i], acc1);
                    acc2 = MFMA16(a2[i], b0[i], acc2);
                    acc3 = MFMA16(a3[i], b1[i], acc3);
                }
                if (k + D < KB) fetch(j, k + D);
            }
        }
    }
}

// Branch-free variant for segment lengths that are multiples of the ring depth (64 or more units / channels): no condition
// depends on k -- ring refills and the one-ahead B loads clamp their index instead (a few redundant loads at the tail) -- so
// the compiler can keep the whole ring in flight (with per-slot conditions it falls back to waiting for nearly every load:
// vmcnt(0..5) in the ISA, and the matrix pipe was busy 51 % of the time).
__device__ __forceinline__ void gen_dot4_fast(f32x4& acc0, f32x4& acc1, f32x4& acc2, f32x4& acc3, const f32x4* __restrict__ w0,
                                              const f32x4* __restrict__ w1, const f32x4* __restrict__ w2, const f32x4* __restrict__ w3,
                                              const f32x4* __restrict__ xb, int KBX, const f32x4* hb0, const f32x4* hb1, int H16, unsigned ln) {
    constexpr int D = CF_GEN_DEPTH;
    const int KB = KBX + H16, last = KB - 1;
    f32x4 A0[D], A1[D], A2[D], A3[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { A0[j] = gen_ld(w0 + j * 64, ln); A1[j] = gen_ld(w1 + j * 64, ln); A2[j] = gen_ld(w2 + j * 64, ln); A3[j] = gen_ld(w3 + j * 64, ln); }
    if (KBX > 0) {
        f32x4 b = gen_ld(xb, ln);
        for (int k0 = 0; k0 < KBX; k0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int k = k0 + j;
                const int kn = k + 1 < KBX ? k + 1 : KBX - 1, kr = k + D < last ? k + D : last;
                const f32x4 bn = (CF_GEN_ABL & 2) ? b : gen_ld(xb + kn * 64, ln);
                const f32x4 a0 = A0[j], a1 = A1[j], a2 = A2[j], a3 = A3[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = MFMA16(a0[i], b[i], acc0);
                    acc1 = MFMA16(a1[i], b[i], acc1);
                    acc2 = MFMA16(a2[i], b[i], acc2);
                    acc3 = MFMA16(a3[i], b[i], acc3);
                }
                if (!(CF_GEN_ABL & 1)) { A0[j] = gen_ld(w0 + kr * 64, ln); A1[j] = gen_ld(w1 + kr * 64, ln); A2[j] = gen_ld(w2 + kr * 64, ln); A3[j] = gen_ld(w3 + kr * 64, ln); }
                b = bn;
                __builtin_amdgcn_sched_barrier(0);          // keep the refill here, D slots ahead of its use
            }
        }
    }
    if (H16 == 0) return;
    f32x4 b0 = hb0[0], b1 = hb1[0];
    for (int k0 = 0; k0 < H16; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int kh = k0 + j, k = KBX + kh;
            const int kn = kh + 1 < H16 ? kh + 1 : H16 - 1, kr = k + D < last ? k + D : last;
            const f32x4 n0 = (CF_GEN_ABL & 2) ? b0 : hb0[kn * 64], n1 = (CF_GEN_ABL & 2) ? b1 : hb1[kn * 64];
            const f32x4 a0 = A0[j], a1 = A1[j], a2 = A2[j], a3 = A3[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0 = MFMA16(a0[i], b0[i], acc0);
                acc1 = MFMA16(a1[i], b1[i], acc1);
                acc2 = MFMA16(a2[i], b0[i], acc2);
                acc3 = MFMA16(a3[i], b1[i], acc3);
            }
            if (!(CF_GEN_ABL & 1)) { A0[j] = gen_ld(w0 + kr * 64, ln); A1[j] = gen_ld(w1 + kr * 64, ln); A2[j] = gen_ld(w2 + kr * 64, ln); A3[j] = gen_ld(w3 + kr * 64, ln); }
            b0 = n0; b1 = n1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void gen_dot4(f32x4& acc0, f32x4& acc1, f32x4& acc2, f32x4& acc3, const f32x4* __restrict__ w0,
                                         const f32x4* __restrict__ w1, const f32x4* __restrict__ w2, const f32x4* __restrict__ w3,
                                         const f32x4* __restrict__ xb, int KBX, const f32x4* hb0, const f32x4* hb1, int H16, unsigned ln) {
    if ((KBX % CF_GEN_DEPTH) == 0 && (H16 % CF_GEN_DEPTH) == 0 && KBX + H16 >= CF_GEN_DEPTH)
        gen_dot4_fast(acc0, acc1, acc2, acc3, w0, w1, w2, w3, xb, KBX, hb0, hb1, H16, ln);
    else
        gen_dot4_any(acc0, acc1, acc2, acc3, w0, w1, w2, w3, xb, KBX, hb0, hb1, H16, ln);
}

// ---- block 0's two k = 1 convs on the raw sample (Cin = 1): shortcut and first conv, resnet_class.py:60-66 ---------------
__global__ __launch_bounds__(256) void gen_first_kernel(const float* __restrict__ x_nat, const f32x4* __restrict__ wb /*[4][Co16][64]: w_sc, b_sc, w_1, b_1*/,
                                                        f32x4* __restrict__ SC, f32x4* __restrict__ O1, int64_t n_windows, int n_tiles, int Co16) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;            // (tile, t, mo, lane)
    const int64_t total = (int64_t)n_tiles * CF_T * Co16 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t r = idx >> 6;
    const int mo = (int)(r % Co16);
    const int64_t tt = r / Co16;
    const int t = (int)(tt % CF_T);
    const int64_t w = (tt / CF_T) * CF_TILE + (lane & 15);
    const float xv = w < n_windows ? x_nat[w * CF_T + t] : 0.f;
    const f32x4 ws = wb[(0 * Co16 + mo) * 64 + lane], bs = wb[(1 * Co16 + mo) * 64 + lane];
    const f32x4 w1 = wb[(2 * Co16 + mo) * 64 + lane], b1 = wb[(3 * Co16 + mo) * 64 + lane];
    f32x4 sc, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = fmaf(ws[j], xv, bs[j]);
        o1[j] = fmaxf(fmaf(w1[j], xv, b1[j]), 0.f);
    }
    SC[idx] = sc;
    O1[idx] = o1;
}

// ---- one conv1d (k = 1 or 3, padding SAME inside the 35-sample window) with folded batch norm ---------------------------------
// One wave per (tile, position); output tiles two at a time (two independent MFMA chains sharing the B operand).
__global__ __launch_bounds__(256) void gen_conv_kernel(const f32x4* __restrict__ W /*[taps][Co16][Ki16][64]*/, const f32x4* __restrict__ Bv /*[Co16][64]*/,
                                                       const f32x4* __restrict__ X, const f32x4* __restrict__ R /*residual or null*/,
                                                       f32x4* __restrict__ Y, int n_tiles, int Ki16, int Co16, int taps, int relu /*1: before the residual add, 2: after*/) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave: uniform
    const unsigned ln = lane;
    const int64_t n_tasks = (int64_t)n_tiles * CF_T;
    for (int64_t task = (int64_t)blockIdx.x * 4 + wave; task < n_tasks; task += (int64_t)gridDim.x * 4) {
        const int64_t tile = task / CF_T;
        const int t = (int)(task - tile * CF_T);
        auto finish = [&](int mo, f32x4 acc) {
            if (relu & 1) {                                                   // relu(BN(conv)), resnet_class.py:66,71,76
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
            }
            if (R) acc += R[((tile * CF_T + t) * Co16 + mo) * 64 + lane];     // + shortcut, resnet_class.py:79
            if (relu & 2) {                                                   // relu(o + shortcut), resnet_class.py:80
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
            }
            Y[((tile * CF_T + t) * Co16 + mo) * 64 + lane] = acc;
        };
        const int step = Co16 >= 4 ? 4 : 2;                                   // output tiles per pass: four MFMA chains sharing the B operand
        for (int mo = 0; mo < Co16; mo += step) {
            const int m1 = mo + 1 < Co16 ? mo + 1 : mo, m2 = mo + 2 < Co16 ? mo + 2 : mo, m3 = mo + 3 < Co16 ? mo + 3 : mo;
            f32x4 acc0 = Bv[mo * 64 + lane], acc1 = Bv[m1 * 64 + lane], acc2 = Bv[m2 * 64 + lane], acc3 = Bv[m3 * 64 + lane];
            for (int tap = 0; tap < taps; ++tap) {
                const int tt = t + tap - (taps >> 1);
                if (tt < 0 || tt >= CF_T) continue;                           // zero padding at the window edges
                const f32x4* xb = X + ((tile * CF_T + tt) * Ki16) * 64;       // uniform pointers, lane added at the access
                const f32x4* wt = W + ((int64_t)tap * Co16 * Ki16) * 64;
                if (step == 4)
                    gen_dot4(acc0, acc1, acc2, acc3, wt + (int64_t)mo * Ki16 * 64, wt + (int64_t)m1 * Ki16 * 64, wt + (int64_t)m2 * Ki16 * 64,
                             wt + (int64_t)m3 * Ki16 * 64, xb, Ki16, nullptr, nullptr, 0, ln);
                else
                    gen_dot(acc0, acc1, wt + (int64_t)mo * Ki16 * 64, wt + (int64_t)m1 * Ki16 * 64, xb, Ki16, nullptr, nullptr, 0, ln);
            }
            finish(mo, acc0);
            if (mo + 1 < Co16) finish(m1, acc1);
            if (step == 4 && mo + 2 < Co16) finish(m2, acc2);
            if (step == 4 && mo + 3 < Co16) finish(m3, acc3);
        }
    }
}

// ---- one bidirectional GRU layer (rnn_class.py:142-148,165-175; GRUCell wiring of the checkpoint's graph) -----------------
// One wave = one 16-window tile of one direction, 35 serial steps.  Per step: r = sigmoid(Wr [x, h] + br), two output tiles
// at a time; then, per output tile, c = tanh(Wc [x, r.h] + bc) and u = sigmoid(Wu [x, h] + bu) side by side,
// h' = u h + (1 - u) c.  The gate weights are pre-scaled for exp2 (CF_GATE_SCALE / CF_CAND_SCALE).  The three state arrays
// of the wave live in LDS ([H16][64] f32x4 each); only this wave touches them, in program order.
// TRAIN: the activated gates r, u, c of every step are stashed for the backward pass (S: [tiles][35][2 dirs][3][H16][64]), the
// buffers are the caller's (exactly n_tiles tiles: waves past the last tile exit, so no workgroup barriers in this variant).
template <bool TRAIN>
__global__ __launch_bounds__(512) void gen_gru_kernel(const f32x4* __restrict__ W /*[2 dirs][3: r, u, c][H16][KB][64]*/,
                                                      const f32x4* __restrict__ Bv /*[2][3][H16][64]*/, const f32x4* __restrict__ X /*[tiles][35][KBX][64]*/,
                                                      f32x4* Y /*[tiles][35][2 H16][64]*/, int H16, int KBX, int h_via_y, f32x4* __restrict__ S,
                                                      int n_tiles) {
    constexpr bool LOCKSTEP = CF_GEN_LOCKSTEP && !TRAIN;
    extern __shared__ f32x4 gen_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave: uniform (SGPR)
    const unsigned ln = lane;
    const int waves = blockDim.x >> 6;
    const int dir = blockIdx.y, grp = blockIdx.x;
    // A workgroup's waves read the SAME weight stream; a barrier per output-tile group keeps them within one group of each
    // other, so that the stream (24 KB per group at H = 64) is fetched from L2 once per CU and served to the other waves by
    // the vector L1 (measured: see DESIGN.md, "Any-size path").  Waves past the last tile run on scratch tiles behind it (
    // the host rounds the buffers up to whole workgroups of tiles).
    const int64_t tile = (int64_t)grp * waves + wave;
    if (TRAIN && tile >= n_tiles) return;
    const int KB = KBX + H16;
    // state: h and r.h always in LDS; h' in a third LDS array when eight waves' worth fits (H <= 64), otherwise it takes the
    // round trip through the layer's output (written anyway) and is reloaded into the h array at the end of the step
    const int arrays = h_via_y ? 2 : 3;
    f32x4* hs = gen_lds + (size_t)wave * arrays * H16 * 64 + lane;
    f32x4* rh = hs + (size_t)H16 * 64;
    f32x4* cs = rh + (size_t)H16 * 64;                            // only with three arrays
    const f32x4* Wr = W + ((size_t)(dir * 3 + 0) * H16 * KB) * 64;          // uniform pointers: the lane is added at the access
    const f32x4* Wu = W + ((size_t)(dir * 3 + 1) * H16 * KB) * 64;
    const f32x4* Wc = W + ((size_t)(dir * 3 + 2) * H16 * KB) * 64;
    const f32x4* Br = Bv + ((size_t)(dir * 3 + 0) * H16) * 64 + lane;
    const f32x4* Bu = Bv + ((size_t)(dir * 3 + 1) * H16) * 64 + lane;
    const f32x4* Bc = Bv + ((size_t)(dir * 3 + 2) * H16) * 64 + lane;
    for (int mo = 0; mo < H16; ++mo) hs[mo * 64] = (f32x4){0.f, 0.f, 0.f, 0.f};       // GRUCellZeroState
    for (int s = 0; s < CF_T; ++s) {
        const int t = dir ? CF_T - 1 - s : s;                     // ReverseV2 around the backward direction
        const f32x4* xt = X + ((tile * CF_T + t) * KBX) * 64;
        // reset gate, then r.h (gru_cell/mul -> concat_1): four output tiles at a time (two when H < 64)
        auto reset_tile = [&](int mo, const f32x4& acc) {
            const f32x4 h0 = hs[mo * 64];
            f32x4 r0;
#pragma unroll
            for (int j = 0; j < 4; ++j) r0[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[j]));
            if (TRAIN) S[(((tile * CF_T + t) * 2 + dir) * 3 * H16 + mo) * 64 + lane] = r0;
            rh[mo * 64] = r0 * h0;
        };
        if (H16 >= 4) {
            for (int mo = 0; mo < H16; mo += 4) {
                const int m1 = mo + 1 < H16 ? mo + 1 : mo, m2 = mo + 2 < H16 ? mo + 2 : mo, m3 = mo + 3 < H16 ? mo + 3 : mo;
                if (LOCKSTEP) __syncthreads();
                f32x4 acc0 = Br[mo * 64], acc1 = Br[m1 * 64], acc2 = Br[m2 * 64], acc3 = Br[m3 * 64];
                gen_dot4(acc0, acc1, acc2, acc3, Wr + (size_t)mo * KB * 64, Wr + (size_t)m1 * KB * 64, Wr + (size_t)m2 * KB * 64,
                         Wr + (size_t)m3 * KB * 64, xt, KBX, hs, hs, H16, ln);
                reset_tile(mo, acc0); reset_tile(m1, acc1); reset_tile(m2, acc2); reset_tile(m3, acc3);
            }
        } else {
            for (int mo = 0; mo < H16; mo += 2) {
                const int m1 = mo + 1 < H16 ? mo + 1 : mo;
                if (LOCKSTEP) __syncthreads();
                f32x4 acc0 = Br[mo * 64], acc1 = Br[m1 * 64];
                gen_dot(acc0, acc1, Wr + (size_t)mo * KB * 64, Wr + (size_t)m1 * KB * 64, xt, KBX, hs, hs, H16, ln);
                reset_tile(mo, acc0); reset_tile(m1, acc1);
            }
        }
        // candidate and update gate side by side, two output tiles at a time; h' = u h + (1 - u) c (gru_cell/mul_1, sub, mul_2, add)
        auto update_tile = [&](int mo, const f32x4& accc, const f32x4& accu) {
            const f32x4 h0 = hs[mo * 64];
            f32x4 hn, cv, uv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cv[j] = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(accc[j])), 1.0f);
                uv[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(accu[j]));
                hn[j] = fmaf(uv[j], h0[j] - cv[j], cv[j]);
            }
            if (TRAIN) {
                S[(((tile * CF_T + t) * 2 + dir) * 3 * H16 + H16 + mo) * 64 + lane] = uv;
                S[(((tile * CF_T + t) * 2 + dir) * 3 * H16 + 2 * H16 + mo) * 64 + lane] = cv;
            }
            if (!h_via_y) cs[mo * 64] = hn;
            Y[((tile * CF_T + t) * 2 * H16 + dir * H16 + mo) * 64 + lane] = hn;
        };
        if (H16 >= 2) {
            for (int mo = 0; mo < H16; mo += 2) {
                const int m1 = mo + 1 < H16 ? mo + 1 : mo;
                if (LOCKSTEP) __syncthreads();
                f32x4 c0 = Bc[mo * 64], u0 = Bu[mo * 64], c1 = Bc[m1 * 64], u1 = Bu[m1 * 64];
                gen_dot4(c0, u0, c1, u1, Wc + (size_t)mo * KB * 64, Wu + (size_t)mo * KB * 64, Wc + (size_t)m1 * KB * 64,
                         Wu + (size_t)m1 * KB * 64, xt, KBX, rh, hs, H16, ln);
                update_tile(mo, c0, u0);
                if (m1 != mo) update_tile(m1, c1, u1);
            }
        } else {
            if (LOCKSTEP) __syncthreads();
            f32x4 c0 = Bc[0], u0 = Bu[0];
            gen_dot(c0, u0, Wc, Wu, xt, KBX, rh, hs, H16, ln);
            update_tile(0, c0, u0);
        }
        if (h_via_y) {
            __builtin_amdgcn_s_waitcnt(0);                         // this wave's stores of h' have been acknowledged before it reads them back
            for (int mo = 0; mo < H16; ++mo) hs[mo * 64] = Y[((tile * CF_T + t) * 2 * H16 + dir * H16 + mo) * 64 + lane];
        } else {
            f32x4* tmp = hs; hs = cs; cs = tmp;
        }
    }
}


// ---- the same layer, TWO 16-window tiles per wave ------------------------------------------------------------------------------
// Every weight fragment a wave fetches feeds eight MFMAs instead of four: the one-tile kernel is bounded by the vector L1 that
// serves the weight stream (0.32 accesses per CU and cycle at 52 % matrix-pipe utilisation, profiles/r02_any_size_pmc.json), not
// by the matrix pipe.  Inference only, segment lengths multiples of the ring depth (64 or more units and input features), h'
// through the output buffer (two LDS arrays per tile); launched when there are enough tiles to fill the chip.
// The weight ring (A0..A3: the next D k-blocks of the four streams) lives across calls: a call consumes it and its refills run
// on into the streams of the NEXT call (n0..n3), so no call starts with an empty ring (a refill at the start of every call cost
// 7-13 k cycles of a 25 k-cycle call, profiles/r02_any_size_stamps.log).  gen_ring_fill primes it once per pass sequence.
struct gen_ring { f32x4 A0[CF_GEN_DEPTH], A1[CF_GEN_DEPTH], A2[CF_GEN_DEPTH], A3[CF_GEN_DEPTH]; };

// Global pointers here are UNIFORM (no lane term): the lane offset is added at the access, so that the address is an SGPR base
// plus one constant VGPR offset -- no 64-bit VALU address arithmetic per load (fp32 MFMA and VALU do not overlap: every
// v_lshl_add_u64 in the loop was lost matrix time).
__device__ __forceinline__ void gen_ring_fill(gen_ring& r, const f32x4* __restrict__ w0, const f32x4* __restrict__ w1,
                                              const f32x4* __restrict__ w2, const f32x4* __restrict__ w3, unsigned ln) {
#pragma unroll
    for (int j = 0; j < CF_GEN_DEPTH; ++j) { r.A0[j] = (w0 + j * 64)[ln]; r.A1[j] = (w1 + j * 64)[ln]; r.A2[j] = (w2 + j * 64)[ln]; r.A3[j] = (w3 + j * 64)[ln]; }
}

__device__ __forceinline__ void gen_dot4x2(f32x4 (&acc)[2][4], gen_ring& ring, const f32x4* __restrict__ w0, const f32x4* __restrict__ w1,
                                           const f32x4* __restrict__ w2, const f32x4* __restrict__ w3, const f32x4* __restrict__ n0,
                                           const f32x4* __restrict__ n1, const f32x4* __restrict__ n2, const f32x4* __restrict__ n3,
                                           const f32x4* __restrict__ xa, const f32x4* __restrict__ xb, int KBX, const f32x4* a0p,
                                           const f32x4* a1p, const f32x4* b0p, const f32x4* b1p, int H16, unsigned ln) {
    constexpr int D = CF_GEN_DEPTH;
    const int KB = KBX + H16;
    f32x4 (&A0)[D] = ring.A0, (&A1)[D] = ring.A1, (&A2)[D] = ring.A2, (&A3)[D] = ring.A3;
    // refill source of k-block kr = k + D: this call's streams, or (kr >= KB) the first blocks of the next call's
    auto refill = [&](int j, int kr) {
        if (CF_GEN_ABL & 1) return;
        const bool nxt = kr >= KB;
        const int ko = (nxt ? kr - KB : kr) * 64;
        A0[j] = ((nxt ? n0 : w0) + ko)[ln]; A1[j] = ((nxt ? n1 : w1) + ko)[ln]; A2[j] = ((nxt ? n2 : w2) + ko)[ln]; A3[j] = ((nxt ? n3 : w3) + ko)[ln];
    };
    {
        f32x4 ba = xa[ln], bb = xb[ln];
        for (int k0 = 0; k0 < KBX; k0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int k = k0 + j;
                const int kn = k + 1 < KBX ? k + 1 : KBX - 1;
                const f32x4 na = (CF_GEN_ABL & 2) ? ba : (xa + kn * 64)[ln], nb = (CF_GEN_ABL & 2) ? bb : (xb + kn * 64)[ln];
                const f32x4 a0 = A0[j], a1 = A1[j], a2 = A2[j], a3 = A3[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][0] = MFMA16(a0[i], ba[i], acc[0][0]);
                    acc[1][0] = MFMA16(a0[i], bb[i], acc[1][0]);
                    acc[0][1] = MFMA16(a1[i], ba[i], acc[0][1]);
                    acc[1][1] = MFMA16(a1[i], bb[i], acc[1][1]);
                    acc[0][2] = MFMA16(a2[i], ba[i], acc[0][2]);
                    acc[1][2] = MFMA16(a2[i], bb[i], acc[1][2]);
                    acc[0][3] = MFMA16(a3[i], ba[i], acc[0][3]);
                    acc[1][3] = MFMA16(a3[i], bb[i], acc[1][3]);
                }
                refill(j, k + D);
                ba = na; bb = nb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    f32x4 pa0 = a0p[0], pa1 = a1p[0], pb0 = b0p[0], pb1 = b1p[0];       // tile a: operands of slots 0, 2 / 1, 3; tile b likewise
    for (int k0 = 0; k0 < H16; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int kh = k0 + j, k = KBX + kh;
            const int kn = kh + 1 < H16 ? kh + 1 : H16 - 1;
            const f32x4 na0 = (CF_GEN_ABL & 2) ? pa0 : a0p[kn * 64], na1 = (CF_GEN_ABL & 2) ? pa1 : a1p[kn * 64];
            const f32x4 nb0 = (CF_GEN_ABL & 2) ? pb0 : b0p[kn * 64], nb1 = (CF_GEN_ABL & 2) ? pb1 : b1p[kn * 64];
            const f32x4 a0 = A0[j], a1 = A1[j], a2 = A2[j], a3 = A3[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = MFMA16(a0[i], pa0[i], acc[0][0]);
                acc[1][0] = MFMA16(a0[i], pb0[i], acc[1][0]);
                acc[0][1] = MFMA16(a1[i], pa1[i], acc[0][1]);
                acc[1][1] = MFMA16(a1[i], pb1[i], acc[1][1]);
                acc[0][2] = MFMA16(a2[i], pa0[i], acc[0][2]);
                acc[1][2] = MFMA16(a2[i], pb0[i], acc[1][2]);
                acc[0][3] = MFMA16(a3[i], pa1[i], acc[0][3]);
                acc[1][3] = MFMA16(a3[i], pb1[i], acc[1][3]);
            }
            refill(j, k + D);
            pa0 = na0; pa1 = na1; pb0 = nb0; pb1 = nb1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(512) void gen_gru2_kernel(const f32x4* __restrict__ W /*[2 dirs][3: r, u, c][H16][KB][64]*/,
                                                       const f32x4* __restrict__ Bv /*[2][3][H16][64]*/, const f32x4* __restrict__ X /*[tiles][35][KBX][64]*/,
                                                       f32x4* Y /*[tiles][35][2 H16][64]*/, int H16, int KBX) {
    extern __shared__ f32x4 gen_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave: uniform (SGPR)
    const unsigned ln = lane;
    const int waves = blockDim.x >> 6;
    const int dir = blockIdx.y, grp = blockIdx.x;
    const int64_t tile0 = ((int64_t)grp * waves + wave) * 2;                  // tiles tile0, tile0 + 1 (scratch tiles past the end)
    const int KB = KBX + H16;
    f32x4* hs[2];
    f32x4* rh[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        hs[n] = gen_lds + ((size_t)wave * 2 + n) * 2 * H16 * 64 + lane;
        rh[n] = hs[n] + (size_t)H16 * 64;
    }
    const f32x4* Wr = W + ((size_t)(dir * 3 + 0) * H16 * KB) * 64;          // uniform pointers: the lane is added at the access
    const f32x4* Wu = W + ((size_t)(dir * 3 + 1) * H16 * KB) * 64;
    const f32x4* Wc = W + ((size_t)(dir * 3 + 2) * H16 * KB) * 64;
    const f32x4* Br = Bv + ((size_t)(dir * 3 + 0) * H16) * 64;
    const f32x4* Bu = Bv + ((size_t)(dir * 3 + 1) * H16) * 64;
    const f32x4* Bc = Bv + ((size_t)(dir * 3 + 2) * H16) * 64;
    for (int mo = 0; mo < H16; ++mo) { hs[0][mo * 64] = (f32x4){0.f, 0.f, 0.f, 0.f}; hs[1][mo * 64] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    long long st_[6] = {0, 0, 0, 0, 0, 0};
    gen_ring ring;
    for (int s = 0; s < CF_T; ++s) {
        const int t = dir ? CF_T - 1 - s : s;
        const f32x4* xt0 = X + ((tile0 * CF_T + t) * KBX) * 64;
        const f32x4* xt1 = X + (((tile0 + 1) * CF_T + t) * KBX) * 64;
        f32x4 acc[2][4];
        // stream bases of call c of this step: c < nr: reset-gate group c; then candidate / update group c - nr; then the next step's first
        const int nr = H16 / 4, ncu = H16 / 2;
        auto stream = [&](int c, int q) -> const f32x4* {
            if (c >= nr + ncu) c = 0;                                         // (the ring runs on into the next step: same weights)
            if (c < nr) return Wr + (size_t)(4 * c + q) * KB * 64;
            const int mo = 2 * (c - nr) + (q >> 1);
            return ((q & 1) ? Wu : Wc) + (size_t)mo * KB * 64;
        };
        if (s == 0) gen_ring_fill(ring, stream(0, 0), stream(0, 1), stream(0, 2), stream(0, 3), ln);
        long long ts_ = CF_GEN_STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define CF_GSTAMP(i) if (CF_GEN_STAMP) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); st_[i] += tn_ - ts_; ts_ = tn_; }
        for (int mo = 0; mo < H16; mo += 4) {                                 // reset gate, r.h
            if (CF_GEN_LOCKSTEP) __syncthreads();
            CF_GSTAMP(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[0][q] = (Br + (mo + q) * 64)[ln]; acc[1][q] = acc[0][q]; }
            {
                const int c = mo / 4;
                gen_dot4x2(acc, ring, stream(c, 0), stream(c, 1), stream(c, 2), stream(c, 3), stream(c + 1, 0), stream(c + 1, 1),
                           stream(c + 1, 2), stream(c + 1, 3), xt0, xt1, KBX, hs[0], hs[0], hs[1], hs[1], H16, ln);
            }
            CF_GSTAMP(1);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 h0 = hs[n][(mo + q) * 64];
                    f32x4 r0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r0[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][q][j])) * h0[j];
                    rh[n][(mo + q) * 64] = r0;
                }
            CF_GSTAMP(2);
        }
        for (int mo = 0; mo < H16; mo += 2) {                                 // candidate and update gate, h'
            if (CF_GEN_LOCKSTEP) __syncthreads();
            CF_GSTAMP(0);
#pragma unroll
            for (int n = 0; n < 2; ++n) { acc[n][0] = (Bc + mo * 64)[ln]; acc[n][1] = (Bu + mo * 64)[ln]; acc[n][2] = (Bc + (mo + 1) * 64)[ln]; acc[n][3] = (Bu + (mo + 1) * 64)[ln]; }
            {
                const int c = nr + mo / 2;
                gen_dot4x2(acc, ring, stream(c, 0), stream(c, 1), stream(c, 2), stream(c, 3), stream(c + 1, 0), stream(c + 1, 1),
                           stream(c + 1, 2), stream(c + 1, 3), xt0, xt1, KBX, rh[0], hs[0], rh[1], hs[1], H16, ln);
            }
            CF_GSTAMP(3);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x4 h0 = hs[n][(mo + p) * 64];
                    f32x4 hn;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float c = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][2 * p][j])), 1.0f);
                        const float u = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][2 * p + 1][j]));
                        hn[j] = fmaf(u, h0[j] - c, c);
                    }
                    Y[(((tile0 + n) * CF_T + t) * 2 * H16 + dir * H16 + mo + p) * 64 + lane] = hn;
                }
            CF_GSTAMP(4);
        }
        __builtin_amdgcn_s_waitcnt(0);                                        // h' back from the output buffer (the wave's own stores)
        for (int mo = 0; mo < H16; ++mo) {
            hs[0][mo * 64] = Y[((tile0 * CF_T + t) * 2 * H16 + dir * H16 + mo) * 64 + lane];
            hs[1][mo * 64] = Y[(((tile0 + 1) * CF_T + t) * 2 * H16 + dir * H16 + mo) * 64 + lane];
        }
        CF_GSTAMP(5);
    }
#undef CF_GSTAMP
    if (CF_GEN_STAMP && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        printf("gen_gru2 stamps (cycles over 35 steps, wave 0): barrier %lld  r-dot %lld  r-epilogue %lld  cu-dot %lld  cu-epilogue %lld  readback %lld  (H16 %d KBX %d)\n",
               st_[0], st_[1], st_[2], st_[3], st_[4], st_[5], H16, KBX);
}

// Matrix-vector helper of the backward pass: out[mo] = sum_k W[mo][k] . B[k] for M16 output tiles (four at a time, two when
// M16 < 4), B a K16-block array of this wave in LDS; epi(mo, acc) once per output tile.
template <typename EP>
__device__ __forceinline__ void gen_matvec(const f32x4* __restrict__ W /*uniform pointer [M16][K16][64]*/, int M16, int K16, const f32x4* B, EP epi, unsigned ln) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (M16 >= 4) {
        for (int mo = 0; mo < M16; mo += 4) {
            const int m1 = mo + 1 < M16 ? mo + 1 : mo, m2 = mo + 2 < M16 ? mo + 2 : mo, m3 = mo + 3 < M16 ? mo + 3 : mo;
            f32x4 a0 = z, a1 = z, a2 = z, a3 = z;
            gen_dot4(a0, a1, a2, a3, W + (size_t)mo * K16 * 64, W + (size_t)m1 * K16 * 64, W + (size_t)m2 * K16 * 64, W + (size_t)m3 * K16 * 64,
                     nullptr, 0, B, B, K16, ln);
            epi(mo, a0);
            if (mo + 1 < M16) epi(m1, a1);
            if (mo + 2 < M16) epi(m2, a2);
            if (mo + 3 < M16) epi(m3, a3);
        }
    } else {
        for (int mo = 0; mo < M16; mo += 2) {
            const int m1 = mo + 1 < M16 ? mo + 1 : mo;
            f32x4 a0 = z, a1 = z;
            gen_dot(a0, a1, W + (size_t)mo * K16 * 64, W + (size_t)m1 * K16 * 64, nullptr, 0, B, B, K16, ln);
            epi(mo, a0);
            if (mo + 1 < M16) epi(m1, a1);
        }
    }
}

// ---- backward through time of one biGRU layer: the serial chain only ------------------------------------------------------------
// Per step (reverse of the forward order), with h = the state BEFORE the step, dh = dL/dh' carried from the later step:
//   dh' += dy;  du = dh' (h - c);  dc = dh' (1 - u);  dh = dh' u;  da_c = dc (1 - c^2);  da_u = du u (1 - u)
//   d(r.h) = Wc_h^T da_c;  dr = d(r.h) h;  dh += d(r.h) r;  da_r = dr r (1 - r);  dh += Wg_h^T [da_r; da_u]
// (rnn_class.py:142-148 differentiated; the graph's gradient nodes of the GRU cell).  Output: the pre-activation gradients
// DA [tiles][35][2][3: r, u, c][H16][64]; the input gradient (W_x^T da) and the weight gradients ([x; h]^T da) are plain GEMMs
// over all (window, step) pairs and are left to the caller.  WT: per direction Wc_h^T [H16][H16][64] then Wg_h^T [H16][2 H16][64],
// A-fragment order, unscaled.
__global__ __launch_bounds__(512) void gen_gru_bwd_kernel(const f32x4* __restrict__ WT, const f32x4* __restrict__ Y /*[tiles][35][2 H16][64]*/,
                                                          const f32x4* __restrict__ S, const f32x4* __restrict__ DY /*[tiles][35][2 H16][64]*/,
                                                          f32x4* __restrict__ DA, int n_tiles, int H16) {
    extern __shared__ f32x4 gen_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, dir = blockIdx.y;
    const unsigned ln = lane;
    const int waves = blockDim.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * waves + wave;
    if (tile >= n_tiles) return;
    f32x4* dh = gen_lds + (size_t)wave * 4 * H16 * 64 + lane;     // [dh][da_c][da_r][da_u], H16 blocks each; da_r, da_u adjacent
    f32x4* dac = dh + (size_t)H16 * 64;
    f32x4* dar = dac + (size_t)H16 * 64;
    f32x4* dau = dar + (size_t)H16 * 64;
    const f32x4* WcT = WT + ((size_t)dir * 3 * H16 * H16) * 64;            // uniform: the lane is added at the access
    const f32x4* WgT = WcT + (size_t)H16 * H16 * 64;
    for (int mo = 0; mo < H16; ++mo) dh[mo * 64] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < CF_T; ++s) {
        const int t = dir ? s : CF_T - 1 - s;                     // the forward pass ran 0..34 (dir 0) / 34..0 (dir 1)
        const int tp = dir ? t + 1 : t - 1;                       // where the state before step t was written
        const bool has_prev = tp >= 0 && tp < CF_T;
        const f32x4* yp = Y + ((tile * CF_T + (has_prev ? tp : t)) * 2 * H16 + dir * H16) * 64 + lane;
        const f32x4* st = S + (((tile * CF_T + t) * 2 + dir) * 3 * H16) * 64 + lane;
        const f32x4* dy = DY + ((tile * CF_T + t) * 2 * H16 + dir * H16) * 64 + lane;
        f32x4* da = DA + (((tile * CF_T + t) * 2 + dir) * 3 * H16) * 64 + lane;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int mo = 0; mo < H16; ++mo) {
            const f32x4 hp = has_prev ? yp[mo * 64] : zero, u = st[(H16 + mo) * 64], c = st[(2 * H16 + mo) * 64];
            const f32x4 dht = dy[mo * 64] + dh[mo * 64];
            const f32x4 du = dht * (hp - c), dc = dht * (1.0f - u);
            const f32x4 ac = dc * (1.0f - c * c), au = du * u * (1.0f - u);
            dh[mo * 64] = dht * u;
            dac[mo * 64] = ac;
            dau[mo * 64] = au;
            da[(2 * H16 + mo) * 64] = ac;
            da[(H16 + mo) * 64] = au;
        }
        gen_matvec(WcT, H16, H16, dac, [&](int mo, const f32x4& drh) {
            const f32x4 hp = has_prev ? yp[mo * 64] : zero, r = st[mo * 64];
            const f32x4 ar = drh * hp * r * (1.0f - r);
            dh[mo * 64] = dh[mo * 64] + drh * r;
            dar[mo * 64] = ar;
            da[mo * 64] = ar;
        }, ln);
        gen_matvec(WgT, H16, 2 * H16, dar, [&](int mo, const f32x4& g) { dh[mo * 64] = dh[mo * 64] + g; }, ln);
    }
}

// ---- dense 2H -> 1 + sigmoid (rnn_class.py:178-183, :84): one wave per (tile, position) ------------------------------------------
__global__ __launch_bounds__(256) void gen_head_kernel(const f32x4* __restrict__ Y /*[tiles][35][F16][64]*/, const f32x4* __restrict__ dw /*[F16][64]*/,
                                                       float bias, float* __restrict__ probs, float* __restrict__ logits, int64_t n_windows,
                                                       int n_tiles, int F16) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n_tasks = (int64_t)n_tiles * CF_T;
    for (int64_t task = (int64_t)blockIdx.x * 4 + wave; task < n_tasks; task += (int64_t)gridDim.x * 4) {
        const int64_t tile = task / CF_T;
        const int t = (int)(task - tile * CF_T);
        const f32x4* y = Y + (task * F16) * 64 + lane;
        float p = 0.f;
        for (int mo = 0; mo < F16; ++mo) {
            const f32x4 v = y[mo * 64], w = dw[mo * 64 + lane];
            p = fmaf(v[0], w[0], p); p = fmaf(v[1], w[1], p); p = fmaf(v[2], w[2], p); p = fmaf(v[3], w[3], p);
        }
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        const int64_t w = tile * CF_TILE + (lane & 15);
        if (lane < 16 && w < n_windows) {
            const float z = p + bias;
            if (logits) logits[w * CF_T + t] = z;
            if (probs) probs[w * CF_T + t] = 1.0f / (1.0f + __expf(-z));
        }
    }
}

// ---- host side: packing ---------------------------------------------------------------------------------------------------------
struct cf_generic {
    int C16 = 0, H16 = 0;                       // conv channels / 16 (0 = RNN type), GRU units / 16
    struct Block { f32x4* w_sc = nullptr; f32x4* b_sc = nullptr; f32x4* w_1 = nullptr; f32x4* b_1 = nullptr;
                   f32x4* w_3 = nullptr; f32x4* b_3 = nullptr; f32x4* w_l = nullptr; f32x4* b_l = nullptr; f32x4* first = nullptr; };
    std::vector<Block> blocks;
    struct Layer { f32x4* w = nullptr; f32x4* b = nullptr; int kbx = 0;
                   float* tuned = nullptr; int tuned_cin = 0; };     // 64 units and 16 / 32 / 128 inputs: the LDS-resident kernel's pack
    std::vector<Layer> layers;
    f32x4* dense = nullptr;
    std::vector<void*> owned;                   // every device allocation above
    float* r[4] = {nullptr, nullptr, nullptr, nullptr};      // conv activations, C features
    float* g[2] = {nullptr, nullptr};                        // biGRU outputs, 2H features
    int gru_waves = 8;
    bool h_via_y = false;
    int gru2_waves = 0;                         // gen_gru2_kernel (two tiles per wave): waves per workgroup, 0 = not used
    size_t gru2_lds = 0;
    size_t gru_lds = 0;
};

// W(in, out) accessor -> A pack; inputs in >= k_real are zero padding
template <typename F>
static void gen_pack_a(std::vector<float>& dst, size_t off, F w, int k_real, int K16, int M16, double scale) {
    for (int mo = 0; mo < M16; ++mo)
        for (int kb = 0; kb < K16; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int in = 16 * kb + 4 * (lane >> 4) + i, out = 16 * mo + (lane & 15);
                    dst[off + (((size_t)mo * K16 + kb) * 64 + lane) * 4 + i] = in < k_real ? (float)(w(in, out) * scale) : 0.f;
                }
}

// per-output vector (bias, dense weights) in accumulator order: [mo][lane][j] = v[16 mo + 4 (lane >> 4) + j]
template <typename F>
static void gen_pack_v(std::vector<float>& dst, size_t off, F v, int M16, double scale) {
    for (int mo = 0; mo < M16; ++mo)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) dst[off + ((size_t)mo * 64 + lane) * 4 + j] = (float)(v(16 * mo + 4 * (lane >> 4) + j) * scale);
}
