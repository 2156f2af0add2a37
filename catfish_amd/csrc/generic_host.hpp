// Host side of the any-size inference path (generic.hpp): weight folding / tiling, workspace, launch sequence.
// Included by catfish_hip.hip after cf_model, fail(), HIP_TRY, prof_begin / prof_end.
#pragma once

static bool gen_forced() { return cf_knob("CATFISH_GENERIC") && atoi(cf_knob("CATFISH_GENERIC")) != 0; }     // A/B and test knob, read per model

static bool gen_wanted(const cf_hparams* hp) {
    return gen_forced() || hp->layer_size != CF_H || (hp->n_layers_res > 0 && hp->layer_size_res != CF_C);
}

static void gen_destroy(cf_generic* g) {
    if (!g) return;
    for (void* p : g->owned) (void)hipFree(p);
    delete g;
}

static int gen_upload(cf_generic* g, const std::vector<float>& host, f32x4** dev) {
    float* p = nullptr;
    HIP_TRY(hipMalloc((void**)&p, host.size() * sizeof(float)));
    g->owned.push_back(p);
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<f32x4*>(p);
    return CF_OK;
}

// y = BN(conv(x)) = conv'(x), any channel count (same arithmetic as fold(): double math on the fp32 inputs)
struct GenConv {
    int k = 0, cin = 0, cout = 0;
    std::vector<double> w;      // [k][cin][cout]
    std::vector<double> b;      // [cout]
};

static GenConv gen_fold(const cf_conv_bn& c, float eps, int cout) {
    GenConv f;
    f.k = c.ksize; f.cin = c.cin; f.cout = cout;
    f.w.resize((size_t)c.ksize * c.cin * cout);
    f.b.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const double s = (double)c.gamma[o] / std::sqrt((double)c.moving_variance[o] + (double)eps);
        f.b[o] = (double)c.bias[o] * s + (double)c.beta[o] - (double)c.moving_mean[o] * s;
        for (int k = 0; k < c.ksize; ++k)
            for (int i = 0; i < c.cin; ++i)
                f.w[((size_t)k * c.cin + i) * cout + o] = (double)c.kernel[((size_t)k * c.cin + i) * cout + o] * s;
    }
    return f;
}

static int gen_pack_conv(cf_generic* g, const GenConv& f, f32x4** w_dev, f32x4** b_dev) {
    const int K16 = f.cin / 16, M16 = f.cout / 16;
    std::vector<float> wp((size_t)f.k * M16 * K16 * 256), bp((size_t)M16 * 256);
    for (int tap = 0; tap < f.k; ++tap)
        gen_pack_a(wp, (size_t)tap * M16 * K16 * 256,
                   [&](int in, int out) { return f.w[((size_t)tap * f.cin + in) * f.cout + out]; }, f.cin, K16, M16, 1.0);
    gen_pack_v(bp, 0, [&](int o) { return f.b[o]; }, M16, 1.0);
    int rc = gen_upload(g, wp, w_dev);
    return rc != CF_OK ? rc : gen_upload(g, bp, b_dev);
}

static int gen_build(cf_model* m, const cf_weights* w) {
    const cf_hparams& hp = m->hp;
    const int H = hp.layer_size, C = hp.n_layers_res > 0 ? hp.layer_size_res : 0;
    if (H < 16 || H > 256 || (H % 16) != 0)
        return fail(CF_ERR_INVALID, "layer_size must be a multiple of 16 between 16 and 256 (the reference draws 16, 32, 64, 128, 256)");
    if (hp.n_layers_res > 0 && (C < 16 || C > 256 || (C % 16) != 0))
        return fail(CF_ERR_INVALID, "layer_size_res must be a multiple of 16 between 16 and 256 (the reference draws 16, 32, 64, 128, 256)");
    if (hp.precision != CF_PREC_FP32)
        return fail(CF_ERR_INVALID, "the bf16 / bf16x3 kernels are built for layer_size = 64 and layer_size_res = 32 only; other sizes run in CF_PREC_FP32");
    cf_generic* g = new cf_generic();
    m->gen = g;
    g->H16 = H / 16;
    g->C16 = C / 16;
    int rc = CF_OK;
    // residual blocks (resnet_class.py:44-82): conv order per block = shortcut, first, middle (k = 3), last
    for (int b = 0; b < hp.n_layers_res && rc == CF_OK; ++b) {
        const cf_conv_bn* c4 = w->conv + 4 * b;
        const int cin = b == 0 ? 1 : C;
        if (c4[0].ksize != 1 || c4[1].ksize != 1 || c4[2].ksize != 3 || c4[3].ksize != 1 || c4[0].cin != cin || c4[1].cin != cin ||
            c4[2].cin != C || c4[3].cin != C)
            return fail(CF_ERR_INVALID, "residual block geometry not supported (need k = 1,1,3,1)");
        const GenConv sc = gen_fold(c4[0], hp.bn_epsilon, C), f1 = gen_fold(c4[1], hp.bn_epsilon, C);
        const GenConv f3 = gen_fold(c4[2], hp.bn_epsilon, C), fl = gen_fold(c4[3], hp.bn_epsilon, C);
        cf_generic::Block blk;
        if (b == 0) {
            std::vector<float> v((size_t)4 * g->C16 * 256);
            gen_pack_v(v, (size_t)0 * g->C16 * 256, [&](int o) { return sc.w[o]; }, g->C16, 1.0);
            gen_pack_v(v, (size_t)1 * g->C16 * 256, [&](int o) { return sc.b[o]; }, g->C16, 1.0);
            gen_pack_v(v, (size_t)2 * g->C16 * 256, [&](int o) { return f1.w[o]; }, g->C16, 1.0);
            gen_pack_v(v, (size_t)3 * g->C16 * 256, [&](int o) { return f1.b[o]; }, g->C16, 1.0);
            rc = gen_upload(g, v, &blk.first);
        } else {
            rc = gen_pack_conv(g, sc, &blk.w_sc, &blk.b_sc);
            if (rc == CF_OK) rc = gen_pack_conv(g, f1, &blk.w_1, &blk.b_1);
        }
        if (rc == CF_OK) rc = gen_pack_conv(g, f3, &blk.w_3, &blk.b_3);
        if (rc == CF_OK) rc = gen_pack_conv(g, fl, &blk.w_l, &blk.b_l);
        g->blocks.push_back(blk);
    }
    // biGRU layers: per direction three matrices over K = [x blocks | h blocks]
    for (int l = 0; l < hp.n_layers && rc == CF_OK; ++l) {
        const int cin_real = l == 0 ? (C > 0 ? C : 1) : 2 * H;
        const int kbx = (cin_real + 15) / 16, KB = kbx + g->H16;
        if (w->gru[2 * l].cin != cin_real || w->gru[2 * l + 1].cin != cin_real) return fail(CF_ERR_INVALID, "GRU layer input width mismatch");
        const size_t mat = (size_t)g->H16 * KB * 256, vec = (size_t)g->H16 * 256;
        std::vector<float> wp(2 * 3 * mat), bp(2 * 3 * vec);
        for (int d = 0; d < 2; ++d) {
            const cf_gru_dir& gd = w->gru[2 * l + d];
            for (int gate = 0; gate < 3; ++gate) {                          // 0 = r, 1 = u (gates/kernel columns [0,H) and [H,2H)), 2 = candidate
                const float* kern = gate < 2 ? gd.gates_kernel : gd.candidate_kernel;
                const float* bias = gate < 2 ? gd.gates_bias : gd.candidate_bias;
                const int ld = gate < 2 ? 2 * H : H, col0 = gate == 1 ? H : 0;
                const double scale = gate < 2 ? CF_GATE_SCALE : CF_CAND_SCALE;
                auto acc = [&](int in, int out) -> double {
                    if (in < 16 * kbx) return in < cin_real ? (double)kern[(size_t)in * ld + col0 + out] : 0.0;
                    return (double)kern[(size_t)(cin_real + in - 16 * kbx) * ld + col0 + out];
                };
                gen_pack_a(wp, (size_t)(d * 3 + gate) * mat, acc, 16 * KB, KB, g->H16, scale);
                gen_pack_v(bp, (size_t)(d * 3 + gate) * vec, [&](int o) { return (double)bias[col0 + o]; }, g->H16, scale);
            }
        }
        cf_generic::Layer L;
        L.kbx = kbx;
        rc = gen_upload(g, wp, &L.w);
        if (rc == CF_OK) rc = gen_upload(g, bp, &L.b);
        // 64 units with 16, 32 or 128 input features is what gru_layer_kernel<CIN, false> (weights in LDS, state in registers,
        // 0.78-0.83 of the fp32-MFMA peak) is built for: such layers of an otherwise odd geometry (say 64 units behind 128
        // conv channels) run on it.  Not when CATFISH_GENERIC forces this path: that knob exists to exercise the kernels here.
        if (rc == CF_OK && H == CF_H && !gen_forced() && (cin_real == 16 || cin_real == 32 || cin_real == 128)) {
            std::vector<float> blob((size_t)2 * gru_pack_floats(cin_real));
            for (int d = 0; d < 2; ++d) pack_gru_dir(w->gru[2 * l + d], cin_real, cin_real, nullptr, blob.data() + (size_t)d * gru_pack_floats(cin_real));
            f32x4* dev = nullptr;
            rc = gen_upload(g, blob, &dev);
            L.tuned = reinterpret_cast<float*>(dev);
            L.tuned_cin = cin_real;
        }
        g->layers.push_back(L);
    }
    if (rc == CF_OK) {
        std::vector<float> dv((size_t)2 * g->H16 * 256);
        gen_pack_v(dv, 0, [&](int f) { return (double)w->dense_kernel[f]; }, 2 * g->H16, 1.0);
        rc = gen_upload(g, dv, &g->dense);
    }
    if (rc != CF_OK) return rc;
    // workspace: four conv buffers (input / shortcut / two intermediates), two biGRU output buffers
    int64_t cap = hp.max_windows_per_pass > 0 ? hp.max_windows_per_pass : 32768;
    cap = (cap + CF_TILE - 1) / CF_TILE * CF_TILE;
    m->cap_windows = cap;
    m->cap_tiles = cap / CF_TILE;
    // (+ 16 tiles: the biGRU launch is whole workgroups of up to sixteen tiles; the waves past the last tile own scratch tiles)
    const size_t per_f16 = (size_t)(m->cap_tiles + 16) * CF_T * 64 * sizeof(f32x4);    // bytes of one 16-feature tile plane
    const size_t r_bytes = per_f16 * std::max(1, g->C16), g_bytes = per_f16 * 2 * g->H16;
    const int n_r = g->C16 > 0 ? 4 : 1;
    for (int i = 0; i < n_r; ++i) {
        hipError_t e = hipMalloc((void**)&g->r[i], r_bytes);
        if (e != hipSuccess) return fail(CF_ERR_NOMEM, std::string("workspace allocation: ") + hipGetErrorString(e));
        g->owned.push_back(g->r[i]);
    }
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipMalloc((void**)&g->g[i], g_bytes);
        if (e != hipSuccess) return fail(CF_ERR_NOMEM, std::string("workspace allocation: ") + hipGetErrorString(e));
        g->owned.push_back(g->g[i]);
    }
    m->ws_bytes = (int64_t)(n_r * r_bytes + 2 * g_bytes);
    // biGRU launch shape: the state of a tile takes 3 H 64 B of LDS; as many waves per workgroup as fit (8 at most)
    // (h, r.h and h'); when eight waves of that do not fit, h' goes through the output buffer instead (two arrays) and the
    // workgroup is 8 or 4 waves, so that every SIMD carries the same number
    size_t per_wave = (size_t)3 * g->H16 * 64 * sizeof(f32x4);
    g->h_via_y = per_wave * 8 > (size_t)(160 * 1024);
    if (g->h_via_y) per_wave = (size_t)2 * g->H16 * 64 * sizeof(f32x4);
    g->gru_waves = per_wave * 8 <= (size_t)(160 * 1024) ? 8 : 4;
    if (cf_knob("CATFISH_GEN_WAVES")) g->gru_waves = std::max(1, std::min(g->gru_waves, atoi(cf_knob("CATFISH_GEN_WAVES"))));    // A/B knob for tools/
    g->gru_lds = per_wave * g->gru_waves;
    // two tiles per wave (gen_gru2_kernel): two LDS arrays per tile; 8 or 4 waves per workgroup, else not used
    const size_t per_wave2 = (size_t)2 * 2 * g->H16 * 64 * sizeof(f32x4);
    g->gru2_waves = (g->H16 % 4) != 0 || cf_knob("CATFISH_GEN_ONE_TILE") ? 0 : (per_wave2 * 8 <= (size_t)(160 * 1024) ? 8 : (per_wave2 * 4 <= (size_t)(160 * 1024) ? 4 : 0));
    g->gru2_lds = per_wave2 * g->gru2_waves;
    if (g->gru2_waves)
        HIP_TRY(hipFuncSetAttribute((const void*)gen_gru2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->gru2_lds));
    HIP_TRY(hipFuncSetAttribute((const void*)gen_gru_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->gru_lds));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gru_pack_floats(16) * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gru_pack_floats(32) * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gru_pack_floats(128) * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_coop_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (gru_pack_floats(16) + CF_COOP_XCH_FLOATS) * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_coop_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (gru_pack_floats(32) + CF_COOP_XCH_FLOATS) * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)gru_layer_coop_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (gru_pack_floats(128) + CF_COOP_XCH_FLOATS) * 4));
    return CF_OK;
}

static int gen_run_pass(cf_model* m, const float* x, int64_t n_windows, float* probs, float* logits, hipStream_t s) {
    cf_generic* g = m->gen;
    const int n_tiles = (int)((n_windows + CF_TILE - 1) / CF_TILE);
    const int task_grid = (int)std::min<int64_t>(((int64_t)n_tiles * CF_T + 3) / 4, (int64_t)m->n_cu * 16);
    int rc;
    size_t pi = 0;
    f32x4* R[4];
    for (int i = 0; i < 4; ++i) R[i] = reinterpret_cast<f32x4*>(g->r[i]);
    f32x4* G[2] = {reinterpret_cast<f32x4*>(g->g[0]), reinterpret_cast<f32x4*>(g->g[1])};
    auto conv = [&](const f32x4* wv, const f32x4* bv, const f32x4* in, const f32x4* res, f32x4* out, int taps, int relu, int slot) -> int {
        int r2;
        if ((r2 = prof_begin(m, slot, s, &pi)) != CF_OK) return r2;
        hipLaunchKernelGGL(gen_conv_kernel, dim3(task_grid), dim3(256), 0, s, wv, bv, in, res, out, n_tiles, g->C16, g->C16, taps, relu);
        HIP_TRY(hipGetLastError());
        return prof_end(m, s, pi);
    };
    const f32x4* cur = R[0];
    if (g->C16 > 0) {
        for (size_t b = 0; b < g->blocks.size(); ++b) {
            const cf_generic::Block& k = g->blocks[b];
            const int slot = b == 0 ? SLOT_RES_FIRST : SLOT_RES;
            if (b == 0) {
                if ((rc = prof_begin(m, slot, s, &pi)) != CF_OK) return rc;
                const int64_t n_el = (int64_t)n_tiles * CF_T * g->C16 * 64;
                hipLaunchKernelGGL(gen_first_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, x, k.first, R[1], R[2], n_windows,
                                   n_tiles, g->C16);
                HIP_TRY(hipGetLastError());
                if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
            } else {
                if ((rc = conv(k.w_sc, k.b_sc, R[0], nullptr, R[1], 1, 0, slot)) != CF_OK) return rc;       // shortcut: BN(conv1), no relu
                if ((rc = conv(k.w_1, k.b_1, R[0], nullptr, R[2], 1, 1, slot)) != CF_OK) return rc;
            }
            if ((rc = conv(k.w_3, k.b_3, R[2], nullptr, R[3], 3, 1, slot)) != CF_OK) return rc;
            if ((rc = conv(k.w_l, k.b_l, R[3], R[1], R[0], 1, 3, slot)) != CF_OK) return rc;                // relu(relu(BN(conv1)) + shortcut)
        }
    } else {
        // RNN type: the raw sample as feature 0 of a 16-feature tile (the x rows of the first layer are zero-padded to 16)
        if ((rc = prof_begin(m, SLOT_RES_FIRST, s, &pi)) != CF_OK) return rc;
        const int64_t n_el = (int64_t)n_tiles * CF_T * 64;
        hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, x, R[0], n_windows, n_tiles);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
    }
    for (size_t l = 0; l < g->layers.size(); ++l) {
        const cf_generic::Layer& L = g->layers[l];
        const int slot = l == 0 ? SLOT_GRU0 : (l + 1 == g->layers.size() ? SLOT_GRU_LAST : SLOT_GRU);
        if (L.tuned) {               // a 64-unit layer with 16 / 32 / 128 inputs: the LDS-resident kernel (all its launch regimes)
            const float* x_in = reinterpret_cast<const float*>(cur);
            float* y_out = reinterpret_cast<float*>(G[l & 1]);
            rc = L.tuned_cin == 16 ? launch_gru<16, false>(m, L.tuned, x_in, y_out, nullptr, n_tiles, s, slot)
               : L.tuned_cin == 32 ? launch_gru<32, false>(m, L.tuned, x_in, y_out, nullptr, n_tiles, s, slot)
                                   : launch_gru<128, false>(m, L.tuned, x_in, y_out, nullptr, n_tiles, s, slot);
            if (rc != CF_OK) return rc;
            cur = G[l & 1];
            continue;
        }
        if ((rc = prof_begin(m, slot, s, &pi)) != CF_OK) return rc;
        if (g->gru2_waves && (L.kbx % 4) == 0 && n_tiles >= 2 * m->n_cu) {          // enough tiles to fill the chip two per wave
            const int pairs = (n_tiles + 1) / 2, gx = (pairs + g->gru2_waves - 1) / g->gru2_waves;
            hipLaunchKernelGGL(gen_gru2_kernel, dim3((unsigned)gx, 2), dim3(g->gru2_waves * 64), g->gru2_lds, s,
                               L.w, L.b, cur, G[l & 1], g->H16, L.kbx);
            HIP_TRY(hipGetLastError());
            if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
            cur = G[l & 1];
            continue;
        }
        // small calls (the reference's one-read-per-call pattern): fewer waves per workgroup, so that the tiles spread over the CUs
        const int waves = std::max(1, std::min(g->gru_waves, (2 * n_tiles + m->n_cu - 1) / m->n_cu));
        const int gx = (n_tiles + waves - 1) / waves;
        hipLaunchKernelGGL(gen_gru_kernel<false>, dim3((unsigned)gx, 2), dim3(waves * 64), g->gru_lds / g->gru_waves * waves, s,
                           L.w, L.b, cur, G[l & 1], g->H16, L.kbx, g->h_via_y ? 1 : 0, (f32x4*)nullptr, n_tiles);
        HIP_TRY(hipGetLastError());
        if ((rc = prof_end(m, s, pi)) != CF_OK) return rc;
        cur = G[l & 1];
    }
    if ((rc = prof_begin(m, SLOT_HEAD, s, &pi)) != CF_OK) return rc;
    hipLaunchKernelGGL(gen_head_kernel, dim3(task_grid), dim3(256), 0, s, cur, g->dense, m->dense_bias, probs, logits, n_windows, n_tiles, 2 * g->H16);
    HIP_TRY(hipGetLastError());
    return prof_end(m, s, pi);
}
