// train_step.hpp -- the parts of the training step around the conv stack and the biGRU layers (BASELINE config 5),
// included by catfish_hip.hip:
//
//   train_head_kernel     final_fully_connected + tf.losses.sigmoid_cross_entropy (rnn_class.py:74-79,178-183), forward AND
//                         backward in one pass over the last biGRU layer's output in the kernels' fragment layout:
//                           z = w . h + b,  loss = mean(max(z,0) - z y + log1p(exp(-|z|))),
//                           dz = (sigmoid(z) - y) / count,  dh = dz w,  dw = sum dz h,  db = sum dz
//   train_head_reduce     partial sums added in a fixed order (bit-reproducible, no atomics)
//   opt_step_kernel       tf.train.RMSPropOptimizer / AdamOptimizer update (TF-1 defaults, rnn_class.py:62-71) of ALL variables
//                         at once: parameters, gradients and both slot variables live in flat buffers of one layout
//   gather_scale_kernel   re-tiling of the updated biGRU weights into MFMA fragment order: packed[i] = flat[idx[i]] * scale[i]
#pragma once

#define CF_HEAD_PART 132            // floats per partial: dw[128] | db | loss | 2 pad

// One wave per (tile, t) item: 16 windows x 128 features as 8 f32x4 per lane (feature 16m + 4q + r of window lane & 15).
__global__ __launch_bounds__(256) void train_head_kernel(const f32x4* __restrict__ Y,      // [tile][t][8][lane]: the dense layer's input
                                                         const float* __restrict__ w,      // final_fully_connected/kernel [128]
                                                         const float* __restrict__ bias,   // [1]
                                                         const float* __restrict__ labels, // [n_windows][35]
                                                         f32x4* __restrict__ DY,           // [tile][t][8][lane]: d loss / d input
                                                         float* __restrict__ logits,       // [n_windows][35] or null
                                                         float* __restrict__ part,         // [n_waves][CF_HEAD_PART]
                                                         int64_t n_windows, int n_tiles, float inv_count) {
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, wl = lane & 15;
    const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    f32x4 wv[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) wv[m] = *reinterpret_cast<const f32x4*>(w + 16 * m + 4 * q);
    const float b = bias[0];
    f32x4 dw[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) dw[m] = (f32x4){0, 0, 0, 0};
    float db = 0.f, loss = 0.f;
    const int n_items = n_tiles * CF_T;
    for (int item = wave_g; item < n_items; item += n_waves) {
        const int tile = item / CF_T, t = item - tile * CF_T;
        const f32x4* src = Y + (int64_t)item * 8 * 64 + lane;
        f32x4 h[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) h[m] = src[m * 64];
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            s = fmaf(wv[m].x, h[m].x, s); s = fmaf(wv[m].y, h[m].y, s); s = fmaf(wv[m].z, h[m].z, s); s = fmaf(wv[m].w, h[m].w, s);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float z = s + b;
        const int64_t win = (int64_t)tile * CF_TILE + wl;
        float dz = 0.f;
        if (win < n_windows) {
            const float yv = labels[win * CF_T + t];
            if (q == 0) {
                loss += fmaxf(z, 0.f) - z * yv + log1pf(expf(-fabsf(z)));
                if (logits) logits[win * CF_T + t] = z;
            }
            dz = (1.0f / (1.0f + expf(-z)) - yv) * inv_count;
        }
        if (q == 0) db += dz;
        f32x4* dst = DY + (int64_t)item * 8 * 64 + lane;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            dst[m * 64] = wv[m] * dz;
            dw[m] += h[m] * dz;
        }
    }
    // reduce over the 16 windows of a lane quarter; lane 16 q then holds features 16 m + 4 q + r
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dw[m][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            dw[m][r] = v;
        }
    db += __shfl_xor(db, 1); db += __shfl_xor(db, 2); db += __shfl_xor(db, 4); db += __shfl_xor(db, 8);
    loss += __shfl_xor(loss, 1); loss += __shfl_xor(loss, 2); loss += __shfl_xor(loss, 4); loss += __shfl_xor(loss, 8);
    float* out = part + (size_t)wave_g * CF_HEAD_PART;
    if (wl == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<f32x4*>(out + 16 * m + 4 * q) = dw[m];
        if (q == 0) { out[128] = db; out[129] = loss; out[130] = 0.f; out[131] = 0.f; }
    }
}

// grads = dw[128] | db[1]; loss_out[0] = mean loss.  One 256-thread block per 64 elements (cf_reduce_parts' contract).
__global__ __launch_bounds__(256) void train_head_reduce_kernel(const float* __restrict__ part, int n_parts, float* __restrict__ grads,
                                                                float* __restrict__ loss_out, float inv_count) {
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const float sum = cf_reduce_parts(part, CF_HEAD_PART, n_parts, e, e < 130);
    if (threadIdx.x >= 64 || e >= 130) return;
    if (e < 129) grads[e] = sum;
    else loss_out[0] = sum * inv_count;
}

// kind 0 = RMSProp (decay 0.9, momentum 0, epsilon 1e-10; slot1 = rms (initialised to 1), slot2 = momentum)
// kind 1 = Adam (beta 0.9 / 0.999, epsilon 1e-8; slot1 = m, slot2 = v); t_dev holds the number of steps taken BEFORE this one
__global__ __launch_bounds__(256) void opt_step_kernel(int kind, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                                       float* __restrict__ s2, int64_t n, float lr, const double* __restrict__ t_dev) {
    __shared__ float lr_t_s;
    if (kind == 1 && threadIdx.x == 0) {
        const double t = t_dev[0] + 1.0;
        lr_t_s = (float)((double)lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    if (kind == 0) {
        float ms = s1[i] * 0.9f;
        ms = fmaf(0.1f * gi, gi, ms);                    // addcmul(rms, g, g, value = 1 - decay)
        const float upd = gi / sqrtf(ms + 1e-10f) * lr;
        s1[i] = ms;
        s2[i] = upd;
        p[i] -= upd;
    } else {
        float m = s1[i] * 0.9f;
        m = fmaf(0.1f, gi, m);
        float v = s2[i] * 0.999f;
        v = fmaf(0.001f * gi, gi, v);
        s1[i] = m;
        s2[i] = v;
        p[i] -= m / (sqrtf(v) + 1e-8f) * lr_t_s;
    }
}

__global__ __launch_bounds__(256) void gather_scale_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ scale, float* __restrict__ dst, int64_t n,
                                                           double* __restrict__ t_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && t_dev) t_dev[0] += 1.0;                // the optimizer step that precedes this launch has read it
    if (i < n) dst[i] = src[idx[i]] * scale[i];
}
