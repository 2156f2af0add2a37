// x3_sched_dump.cpp -- prints the compile-time schedule of gru_bf16x3_pipe_kernel (catfish_amd/csrc/gru_bf16x3_sched.hpp):
// per gap the MFMA it follows and the vector micro-ops packed behind it with their issue cost, then the per-phase totals.
//   g++ -std=c++17 -DX3_SCHED_DUMP -o /tmp/x3dump tools/x3_sched_dump.cpp && /tmp/x3dump [32|128] [last]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../catfish_amd/csrc/gru_bf16x3_sched.hpp"

static const char* NAMES[] = {"-", "AE", "AR1", "AR2", "AM", "AP1", "AP2", "AP3", "AP4", "BE", "BR1", "BR2", "CE", "CR1", "CR2",
                              "CC", "CD", "CH", "CL", "CP1", "CP2", "CP3", "CP4", "CS", "LX", "LB"};

template <int CIN, bool LAST>
static void dump(bool verbose) {
    using G = x3::geom<CIN>;
    constexpr auto S = x3::table<CIN, LAST>::S;
    static const char* SRC[] = {"hp", "xa", "rp", "xb"};
    std::printf("CIN %d LAST %d: products A %d B %d C %d, gaps %d, ok %d, heaviest gap %d cycles\n", CIN, (int)LAST, G::NA, G::NB, G::NC,
                G::NGAP, (int)x3::sched_ok<CIN, LAST>(S), S.max_cost);
    int tot[3] = {0, 0, 0}, over[3] = {0, 0, 0}, worst[3] = {0, 0, 0};
    for (int g = 0; g < G::NGAP; ++g) {
        const int ph = g < G::GA ? 0 : (g < G::GA + G::GB ? 1 : 2);
        int c = 0;
        for (int q = 0; q < S.n[g]; ++q) c += x3::cost(S.op[g][q] >> 8);
        const int ring = x3::ring_cost(g);                  // the ring refill ds_read behind the 2nd and 3rd MFMA of a product
        tot[ph] += c + ring;
        if (c + ring > 24) over[ph] += c + ring - 24;
        if (c + ring > worst[ph]) worst[ph] = c + ring;
        if (verbose) {
            const auto d = G::prod(g / 3);
            std::printf("%3d  p%-2d.%d acc%d %s[%d]  %2d :", g, g / 3, g % 3, d.mt, SRC[d.src], d.kb, c + ring);
            for (int q = 0; q < S.n[g]; ++q) std::printf(" %s%d", NAMES[S.op[g][q] >> 8], S.op[g][q] & 255);
            std::printf("\n");
        }
    }
    const int gaps[3] = {G::GA, G::GB, G::GC};
    for (int ph = 0; ph < 3; ++ph)
        std::printf("  phase %c: %3d gaps, %5d issue cycles of fillers = %.1f per gap, worst %d, %d cycles beyond 24 per gap\n", 'A' + ph, gaps[ph],
                    tot[ph], (double)tot[ph] / gaps[ph], worst[ph], over[ph]);
    std::printf("  step: %d MFMA cycles, fillers + 8 per MFMA = %d issue cycles, stall estimate %d\n", 32 * G::NGAP,
                tot[0] + tot[1] + tot[2] + 8 * G::NGAP, over[0] + over[1] + over[2]);
}

int main(int argc, char** argv) {
    const int cin = argc > 1 ? std::atoi(argv[1]) : 128;
    const bool last = argc > 2 && !std::strcmp(argv[2], "last");
    const bool verbose = argc > 3 || (argc > 2 && !std::strcmp(argv[argc - 1], "-v"));
    if (cin == 32) { if (last) dump<32, true>(verbose); else dump<32, false>(verbose); }
    else { if (last) dump<128, true>(verbose); else dump<128, false>(verbose); }
    return 0;
}
