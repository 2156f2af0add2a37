// gru_bf16x3_sched.hpp -- compile-time instruction schedule of gru_bf16x3_pipe_kernel (see gru_bf16x3_pipe.hpp for the why).
// Plain constexpr C++: also compiled on the host by tools/x3_sched_dump.cpp, which prints the gap table.
#pragma once
#ifndef __HIPCC__
#define __host__
#define __device__
#endif

// split of the x products of the 128-input layers over the phases (even numbers; tools/build_x3_variants.sh sweeps them)
#ifndef CF_X3_XUA
#define CF_X3_XUA 8
#endif
#ifndef CF_X3_XCA
#define CF_X3_XCA 8
#endif

// 0: an accumulator's bias row is re-read from LDS every step (24 ds_read_b128, the LB micro-ops);
// 1: it is the C operand of the first MFMA into the accumulator (96 registers hold the six rows for the whole kernel) -- fewer
//    LDS reads and no LB ops, but measured 3.7 % SLOWER on the whole bf16x3 pass (1.183 vs 1.139 ms, three interleaved rounds
//    on one box, tools/ab_x3_variants.sh; profiles/r04_x3_ab_bias_nt.log): kept as an A/B switch only
#ifndef CF_X3_BIAS_C
#define CF_X3_BIAS_C 0
#endif

namespace x3 {

enum : int {
    K_NONE = 0,
    AE, AR1, AR2, AM, AP1, AP2, AP3, AP4,                  // phase A: r gate (element ops E..M, pair ops P1..P4)
    BE, BR1, BR2,                                          // phase B: u gate
    CE, CR1, CR2, CC, CD, CH, CL, CP1, CP2, CP3, CP4, CS,  // phase C: candidate, h update, split, store
    LX, LB,                                                // memory: x fragment of step s + 2, bias row of an accumulator
    K_COUNT
};
// issue cycles of a micro-op on one SIMD (MI355X_MICROARCH.md 'vector-instruction ISSUE cost': transcendental 8, other 4)
__host__ __device__ constexpr int cost(int k) {
    return (k == AE || k == AR2 || k == BE || k == BR2 || k == CE || k == CR2) ? 8       // v_exp_f32 / v_rcp_f32
         : (k == AP2 || k == AP3 || k == CP2 || k == CP3) ? 8                            // two plain instructions
         : (k == LX || k == CS) ? 16                                                     // global_load / global_store_dwordx4: 1 KiB at 64 B/clk
         : k == LB ? 8                                                                   // ds_read_b128: 1 KiB at 128 B/clk
         : 4;
}

template <int CIN>
struct geom {
    static constexpr int KBX = CIN / 16;
    static constexpr int NX = KBX * 6, NG = 16;                 // blob: x products [kb][mt < 6] | gates h [kb][mt < 4] | cand h [kb][2]
    // The 6 KBX x products of a step are dealt to the phases so that each phase's MFMA gaps match its vector work:
    static constexpr int XUA = KBX >= 8 ? CF_X3_XUA : 0;        // Wx_u x_s products left for phase A of step s (the first 2 KBX - XUA ran in C of s - 1)
    static constexpr int XUC = 2 * KBX - XUA;
    static constexpr int XCA = KBX >= 8 ? CF_X3_XCA : 2 * KBX;  // Wx_c x_s products issued in phase A (the rest open phase B)
    static constexpr int NA = XUA + 8 + XCA, NB = (2 * KBX - XCA) + 8, NC = 2 * KBX + XUC + 8;
    static constexpr int OB = NA, OC = NA + NB, NSEQ = NA + NB + NC;
    static constexpr int GA = 3 * NA, GB = 3 * NB, GC = 3 * NC, NGAP = 3 * NSEQ;
    static constexpr int XC0 = XUA + 8;                         // first product of Wx_c x_s
    static constexpr int HC0 = OB + 2 * KBX - XCA;              // first product of Wh_c rp
    static constexpr int XU0 = OC + 2 * KBX;                    // first product of Wx_u x_{s+1} in C
    static constexpr int HR0 = XU0 + XUC;                       // first product of Wh_r h'
    enum : int { SRC_HP, SRC_XA, SRC_RP, SRC_XB };
    struct prod_t { int frag, mt, src, kb; };
    // product i of a step -> blob product, accumulator M-tile (r 0,1  u 2,3  c 4,5), B operand.  Per accumulator the order is
    // bias, x k-blocks ascending, h k-blocks ascending: gru_layer_bf16_kernel's order.
    static __host__ __device__ constexpr prod_t prod(int i) {
        if (i < XUA) { const int q = XUC + i; return {(q >> 1) * 6 + 2 + (q & 1), 2 + (q & 1), SRC_XA, q >> 1}; }       // A: Wx_u x_s, the rest
        if (i < XC0) { const int q = i - XUA; return {NX + (q >> 1) * 4 + 2 + (q & 1), 2 + (q & 1), SRC_HP, q >> 1}; }  // A: Wh_u h
        if (i < HC0) {                                                                                                  // A, B: Wx_c x_s
            const int q = i < OB ? i - XC0 : i - OB + XCA;
            return {(q >> 1) * 6 + 4 + (q & 1), 4 + (q & 1), SRC_XA, q >> 1};
        }
        if (i < OC) { const int q = i - HC0; return {NX + NG + (q >> 1) * 2 + (q & 1), 4 + (q & 1), SRC_RP, q >> 1}; }   // B: Wh_c (r*h)
        if (i < XU0) { const int q = i - OC; return {(q >> 1) * 6 + (q & 1), q & 1, SRC_XB, q >> 1}; }                  // C: Wx_r x_{s+1}
        if (i < HR0) { const int q = i - XU0; return {(q >> 1) * 6 + 2 + (q & 1), 2 + (q & 1), SRC_XB, q >> 1}; }       // C: Wx_u x_{s+1}, first part
        const int q = i - HR0;
        return {NX + (q >> 1) * 4 + (q & 1), q & 1, SRC_HP, q >> 1};                                                    // C: Wh_r h'
    }
    // the product that opens accumulator mt: cyclic order from phase C, where the next step's r and u accumulators start
    static __host__ __device__ constexpr int first_prod(int mt) {
        for (int k = 0; k < NSEQ; ++k) {
            const int i = (OC + k) % NSEQ;
            if (prod(i).mt == mt) return i;
        }
        return -1;
    }
};

template <int NGAP>
struct sched_t {
    static constexpr int MAXO = 14;
    unsigned short op[NGAP][MAXO];      // kind << 8 | index, in program order
    unsigned char n[NGAP];
    short gap[K_COUNT][32];             // gap of (kind, index), -1 = not scheduled
    int max_cost;                       // heaviest gap (issue cycles of the vector micro-ops)
    constexpr void put(int g, int kind, int idx) {
        op[g][n[g]] = (unsigned short)(kind << 8 | idx);
        ++n[g];
        gap[kind][idx] = (short)g;
    }
};

// issue cycles the A ring refill behind MFMA `sub` of a product takes out of its gap (one ds_read_b128 after the 2nd and 3rd)
__host__ __device__ constexpr int ring_cost(int g) { return (g % 3) ? 8 : 0; }

// packs a phase's micro-op list into the gaps [g0, g0 + ng) by cumulative issue cost: a first pass sums the costs (and notes
// where the marked ops fall), a second one places every op so that every gap carries the same load, ring refill included.
// Marks are deadlines: op (mark_kind, mark_idx[i]) and everything listed before it must sit in a gap < g0 + mark_gap[i].
template <int NGAP>
struct packer {
    sched_t<NGAP>* s;
    int g0, ng, total, cum;
    bool dry;
    int n_mark = 0, mark_kind = 0, mark_idx[4] = {0, 0, 0, 0}, mark_gap[4] = {0, 0, 0, 0};
    int seg_c[6] = {0, 0, 0, 0, 0, 0}, seg_g[6] = {0, 0, 0, 0, 0, 0};       // segment boundaries in cost and in gaps, [0] = 0
    int seg = 0, j = 0;                                                     // second pass: current segment, current gap in it
    long long cap = 0;                                                      // capacity of the segment's gaps up to and including j, x dg
    constexpr int ring_sum(int a, int b) const { int r = 0; for (int g = a; g < b; ++g) r += ring_cost(g0 + g); return r; }
    constexpr void finish_dry() {
        const int ring_all = ring_sum(0, ng);
        for (int i = 0; i < n_mark; ++i) {
            // where equal loads would end the segment
            int u = 0;
            while (u < ng && (long long)(seg_c[i + 1] + ring_sum(0, u)) * ng > (long long)(total + ring_all) * u) ++u;
            seg_g[i + 1] = u < mark_gap[i] ? u : mark_gap[i];
            if (seg_g[i + 1] < seg_g[i]) seg_g[i + 1] = seg_g[i];
        }
        seg_c[n_mark + 1] = total;
        seg_g[n_mark + 1] = ng;
        dry = false;
        cum = 0;
        seg = -1;
    }
    constexpr void operator()(int kind, int idx) {
        if (dry) {
            total += cost(kind);
            for (int i = 0; i < n_mark; ++i)
                if (kind == mark_kind && idx == mark_idx[i]) seg_c[i + 1] = total;
            return;
        }
        int i = 0;
        while (i < n_mark && cum >= seg_c[i + 1]) ++i;
        const int dc = seg_c[i + 1] - seg_c[i], dg = seg_g[i + 1] - seg_g[i];
        const int load = dc + ring_sum(seg_g[i], seg_g[i + 1]);             // per gap: load / dg, of which ring_cost is taken
        if (i != seg) { seg = i; j = 0; cap = load - (long long)dg * ring_cost(g0 + seg_g[i]); }
        while (j < dg - 1 && cap <= (long long)(cum - seg_c[i]) * dg) {
            ++j;
            cap += load - (long long)dg * ring_cost(g0 + seg_g[i] + j);
        }
        s->put(g0 + seg_g[i] + (dg > 0 ? j : 0), kind, idx);
        cum += cost(kind);
    }
};
constexpr bool elem(int e) { return e >= 0 && e < 32; }
constexpr bool pair_done(int e) { return e >= 0 && e < 32 && (e & 1); }     // the pair e >> 1 is complete with its odd element

// The lists.  Memory instructions ride in them at fixed ticks:
//   LB  bias rows (one ds_read_b128 = 4 accumulator registers each), index mt * 4 + c4:
//       acc_c early in A (free since C's last exp2, first used by A's first Wx_c product), acc_r in B (free since A's last
//       exp2, first used by C's first product), acc_u early in C (free since B's last exp2, first used by Wx_u x_{s+1})
//   LX  x_{s+2} into the buffer of x_s, in C: its last MFMA (Wx_c) issued in B, and it is next read a whole step later
template <class P> constexpr void list_a(P& p, int L) {
    for (int k = 0; k < 32 + 7 * L; ++k) {
        if (!CF_X3_BIAS_C && k < 8) p(LB, 16 + k);
        if (elem(k)) p(AE, k);
        if (elem(k - L)) p(AR1, k - L);
        if (elem(k - 2 * L)) p(AR2, k - 2 * L);
        if (elem(k - 3 * L)) p(AM, k - 3 * L);
        if (pair_done(k - 4 * L)) p(AP1, (k - 4 * L) >> 1);
        if (pair_done(k - 5 * L)) p(AP2, (k - 5 * L) >> 1);
        if (pair_done(k - 6 * L)) p(AP3, (k - 6 * L) >> 1);
        if (pair_done(k - 7 * L)) p(AP4, (k - 7 * L) >> 1);
    }
}
template <class P> constexpr void list_b(P& p, int L) {
    for (int k = 0; k < 32 + 2 * L; ++k) {
        if (!CF_X3_BIAS_C && k < 32 && (k & 3) == 0) p(LB, k >> 2);
        if (elem(k)) p(BE, k);
        if (elem(k - L)) p(BR1, k - L);
        if (elem(k - 2 * L)) p(BR2, k - 2 * L);
    }
}
template <class P> constexpr void list_c(P& p, int L, bool last, int kbx) {
    for (int k = 0; k < 32 + 10 * L; ++k) {
        if (!CF_X3_BIAS_C && k < 8) p(LB, 8 + k);
        if (kbx >= 8 ? (k < 32 && !(k & 1)) : (k < 32 && (k & 7) == 4)) p(LX, kbx >= 8 ? k >> 1 : k >> 3);
        if (elem(k)) p(CE, k);
        if (elem(k - L)) p(CR1, k - L);
        if (elem(k - 2 * L)) p(CR2, k - 2 * L);
        if (elem(k - 3 * L)) p(CC, k - 3 * L);
        if (elem(k - 4 * L)) p(CD, k - 4 * L);
        if (elem(k - 5 * L)) p(CH, k - 5 * L);
        if (last && elem(k - 6 * L)) p(CL, k - 6 * L);
        if (pair_done(k - 6 * L)) p(CP1, (k - 6 * L) >> 1);
        if (pair_done(k - 7 * L)) p(CP2, (k - 7 * L) >> 1);
        if (pair_done(k - 8 * L)) p(CP3, (k - 8 * L) >> 1);
        if (pair_done(k - 9 * L)) p(CP4, (k - 9 * L) >> 1);
        if (!last) {                                         // k-block kb = pairs 4 kb .. 4 kb + 3: hi part after P1, lo part after P4
            const int e1 = k - 7 * L, e4 = k - 10 * L;
            if (pair_done(e1) && ((e1 >> 1) & 3) == 3) p(CS, (e1 >> 3) * 2 + 0);
            if (pair_done(e4) && ((e4 >> 1) & 3) == 3) p(CS, (e4 >> 3) * 2 + 1);
        }
    }
}

template <int CIN, bool LAST>
constexpr sched_t<geom<CIN>::NGAP> make_sched() {
    using G = geom<CIN>;
    sched_t<G::NGAP> s{};
    for (int k = 0; k < K_COUNT; ++k)
        for (int i = 0; i < 32; ++i) s.gap[k][i] = -1;
    // L = lag between the stages of one element, in list ticks: a tick's cost must reach a gap's share, or an instruction and
    // the one it feeds land in the same gap (correct, but the wave then waits for the result)
    {
        packer<G::NGAP> p{&s, 0, G::GA, 0, 0, true};
        list_a(p, 1);
        p.finish_dry();
        list_a(p, 1);
    }
    {
        packer<G::NGAP> p{&s, G::GA, G::GB, 0, 0, true};
        list_b(p, 2);
        p.finish_dry();
        list_b(p, 2);
    }
    {
        // Wh_r h' closes the phase, k-block by k-block: hp[kb] (pairs 4 kb .. 4 kb + 3, hi and lo) must be complete in the gap
        // before its first MFMA
        packer<G::NGAP> p{&s, G::GA + G::GB, G::GC, 0, 0, true};
        p.n_mark = 4;
        p.mark_kind = CP4;
        for (int kb = 0; kb < 4; ++kb) { p.mark_idx[kb] = 4 * kb + 3; p.mark_gap[kb] = 3 * (G::HR0 - G::OC + 2 * kb); }
        list_c(p, 1, LAST, G::KBX);
        p.finish_dry();
        list_c(p, 1, LAST, G::KBX);
    }
    s.max_cost = 0;
    for (int g = 0; g < G::NGAP; ++g) {
        int c = ring_cost(g);
        for (int q = 0; q < s.n[g]; ++q) c += cost(s.op[g][q] >> 8);
        if (c > s.max_cost) s.max_cost = c;
    }
    return s;
}

// program-order checks: every MFMA operand is complete in an EARLIER gap, every overwritten register is dead
template <int CIN, bool LAST>
constexpr bool sched_ok(const sched_t<geom<CIN>::NGAP>& s) {
    using G = geom<CIN>;
    const int a0 = 0, b0 = G::GA, c0 = G::GA + G::GB;
    for (int e = 0; e < 32; ++e) {
        if (s.gap[AE][e] < a0 || s.gap[AM][e] >= b0) return false;
        if (s.gap[BE][e] < b0 || s.gap[BR2][e] >= c0) return false;
        if (s.gap[CE][e] < c0 || s.gap[CH][e] >= G::NGAP) return false;
    }
    for (int kb = 0; kb < 4; ++kb) {
        // Wh_c rp[kb] reads rp[kb] hi + lo; Wh_r hp[kb] reads the NEW hp[kb]
        if (s.gap[AP4][4 * kb + 3] >= 3 * (G::HC0 + 2 * kb)) return false;
        if (s.gap[CP4][4 * kb + 3] >= 3 * (G::HR0 + 2 * kb)) return false;
        // ... and the old hp[kb] must have been read by Wh_u h (phase A) before CP1 overwrites it: phases are disjoint
    }
    // the accumulators' bias rows: after the gate's last exp2, before the first MFMA into the accumulator
    for (int q = 0; q < (CF_X3_BIAS_C ? 0 : 8); ++q) {
        if (s.gap[LB][16 + q] >= 3 * G::XC0) return false;                  // acc_c: before A's first Wx_c product
        if (s.gap[LB][q] <= s.gap[AE][31] || s.gap[LB][q] >= c0) return false;
        if (s.gap[LB][8 + q] <= s.gap[BE][31] || s.gap[LB][8 + q] >= 3 * G::XU0) return false;
    }
    if (s.gap[CE][31] >= G::NGAP) return false;
    for (int q = 0; q < 2 * G::KBX; ++q)
        if (s.gap[LX][q] < 3 * G::HC0) return false;                         // after the last Wx_c x_s
    return true;
}

template <int CIN, bool LAST>
struct table {
    static constexpr sched_t<geom<CIN>::NGAP> S = make_sched<CIN, LAST>();
#ifndef X3_SCHED_DUMP
    static_assert(sched_ok<CIN, LAST>(S), "bf16x3 schedule: an MFMA would read an operand that is not complete yet");
#endif
};
}   // namespace x3
