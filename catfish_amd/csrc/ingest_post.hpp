// ingest_post.hpp -- round-5 versions of the two small kernels either side of the forward pass:
//   normalize_regs_kernel    per-read median / MAD normalisation (normalize_raw_signal, catfish/infer.py:96-105) + padding + window
//                            packing (infer.py:31-43)
//   postprocess_bits_kernel  threshold + correct_short (catfish/infer.py:128-138, 174-198): labels[i] = 1 iff probs[i] >= threshold
//                            and i lies in a positive run of >= min_run samples inside the real part of its read
// Same results, bit for bit, as normalize_kernel / postprocess_kernel in catfish_hip.hip (which stay: longer reads and unusual
// min_run values fall back to them, and CATFISH_INGEST_V1=1 behind the debug switch selects them for A/B tests).  Included by
// catfish_hip.hip after those kernels.
//
// Why: both are HBM-bound byte work (6 B per sample in + out for the ingest, 5 B for the labels: ~1 us of traffic per 256-read
// batch at 8 TB/s) that ran 80 and 18 us per batch -- 16 % of a bf16 pass, and in the streaming pipeline they run beside the next
// batch's biGRU launches.  normalize_kernel walked global memory 9 times per read with 2-byte loads, built its histograms with
// LDS atomics that all land in two or three bins (DAC samples of one read span a few hundred codes) and let ONE thread scan 256
// bins eight times per read; postprocess_kernel ran a binary search over the read table and a 14-sample walk to either side for
// every positive sample.
#pragma once

// --------------------------------------------------------------------------------------------------------------------------
// Ingest.  One workgroup of 256 threads per read of up to 256 * E samples, which live in REGISTERS (E per thread) from the first
// load to the final store: no LDS staging (the kernel co-resides with the biGRU kernels, which leave 15 KiB of LDS per CU) and
// no second trip to memory.  An order statistic is found exactly by a bitwise search from the top bit down: "how many keys are
// below the candidate?" is E wave ballots + scalar popcounts per wave -- no atomics, no cross-lane VALU -- and one LDS exchange
// between the four waves per bit.  33 such passes per read (16 for the median, 17 for the median absolute deviation) + two more
// for the upper middle element of an even-length read.
// --------------------------------------------------------------------------------------------------------------------------
template <int E>
struct cf_select {
    // count of keys < cand over the whole workgroup.  red: LDS [2][4], double-buffered by the pass counter (one barrier per pass)
    static __device__ __forceinline__ unsigned count_below(const unsigned (&key)[E], unsigned cand, unsigned* red, int& pass) {
        unsigned s = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            s += (unsigned)__popcll(__ballot(key[e] < cand));
            if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // eight compare masks in flight at a time: all E at once spill SGPRs
        }
        unsigned* r = red + (pass & 1) * 4;
        if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
        __syncthreads();
        ++pass;
        return r[0] + r[1] + r[2] + r[3];
    }
    // the element of (zero-based) rank `rank` in ascending order: the largest v with count(key < v) <= rank.  Keys of slots
    // beyond the read are 0xffffffff and never counted (cand < 2^bits <= 2^17).
    static __device__ __forceinline__ unsigned kth(const unsigned (&key)[E], unsigned rank, int bits, unsigned* red, int& pass) {
        unsigned res = 0;
        for (int b = bits - 1; b >= 0; --b) {
            const unsigned cand = res | (1u << b);
            if (count_below(key, cand, red, pass) <= rank) res = cand;
        }
        return res;
    }
    // the element of rank `rank + 1`, given v = the element of rank `rank`: v again when more than rank + 1 keys are <= v, else
    // the smallest key above v
    static __device__ __forceinline__ unsigned next_after(const unsigned (&key)[E], unsigned v, unsigned rank, unsigned* red, int& pass) {
        if (count_below(key, v + 1u, red, pass) > rank + 1u) return v;
        unsigned mn = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < E; ++e) mn = (key[e] > v && key[e] < mn) ? key[e] : mn;          // (padding slots hold 0xffffffff: never below mn)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)mn, d);
            mn = o < mn ? o : mn;
        }
        unsigned* r = red + (pass & 1) * 4;
        if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = mn;
        __syncthreads();
        ++pass;
        const unsigned a = r[0] < r[1] ? r[0] : r[1], b = r[2] < r[3] ? r[2] : r[3];
        return a < b ? a : b;
    }
};

template <int E>
__device__ __forceinline__ void normalize_read_regs(const int16_t* __restrict__ v, int n, float* __restrict__ out, int64_t n_pad,
                                                    unsigned* red) {
    int x[E];
    unsigned key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 256 + (int)threadIdx.x;
        x[e] = i < n ? (int)v[i] : 0;
        key[e] = i < n ? (unsigned)(x[e] + 32768) : 0xffffffffu;
    }
    int pass = 0;
    const unsigned klo = cf_select<E>::kth(key, (unsigned)((n - 1) / 2), 16, red, pass);
    const unsigned khi = (n & 1) ? klo : cf_select<E>::next_after(key, klo, (unsigned)((n - 1) / 2), red, pass);
    const int med2 = ((int)klo - 32768) + ((int)khi - 32768);                  // 2 * median, exact
    // (which slots hold samples is re-derived where it is needed -- from the old key here, from an opaque copy of n below: kept
    // as E compare masks across the whole function they cost 2 E scalar registers, which spill for E = 64)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = 2 * x[e] - med2;
        key[e] = key[e] != 0xffffffffu ? (unsigned)(d < 0 ? -d : d) : 0xffffffffu;   // |2 x - 2 median| <= 131 070 < 2^17
    }
    const unsigned dlo = cf_select<E>::kth(key, (unsigned)((n - 1) / 2), 17, red, pass);
    const unsigned dhi = (n & 1) ? dlo : cf_select<E>::next_after(key, dlo, (unsigned)((n - 1) / 2), red, pass);
    const double shift = 0.5 * (double)med2;
    const double scale = 0.25 * (double)(dlo + dhi);                           // median(|raw - shift|)
    int n_again = n;
    asm volatile("" : "+s"(n_again));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int64_t i = (int64_t)e * 256 + threadIdx.x;
        if (i < n_pad) out[i] = i < n_again ? (float)(((double)x[e] - shift) / scale) : 0.f;
    }
    for (int64_t i = (int64_t)E * 256 + threadIdx.x; i < n_pad; i += 256) out[i] = 0.f;     // (padding beyond the register window)
}

#define CF_NORM_REGS_SMALL 16      // reads of up to 4096 samples: 16 registers of samples per thread
#define CF_NORM_REGS_LARGE 64      // up to 16 384 samples (BASELINE configs[3]'s longest reads): 64

// Two instantiations, launched back to back over the same reads (the host does not know the read lengths -- they live in device
// memory -- and a workgroup whose read belongs to the other instantiation leaves at once):
//   <16>  reads of up to 4096 samples.  ~60 registers per thread, so that its waves fit beside the biGRU kernels' (the streaming
//         pipeline runs the ingest of batch k + 1 on a side stream while batch k is in its forward pass);
//   <64>  everything longer: up to 16 384 samples in registers, beyond that round 1's radix selection over global memory.
template <int E>
__global__ __launch_bounds__(256) void normalize_regs_kernel(const int16_t* __restrict__ dac, const int64_t* __restrict__ dac_offsets,
                                                             const int64_t* __restrict__ win_offsets, float* __restrict__ x_out) {
    __shared__ unsigned hist[E == CF_NORM_REGS_SMALL ? 8 : 512];      // the long-read path (radix_select) needs 512; the register paths 8
    __shared__ unsigned sh[4];
    const int64_t r = blockIdx.x;
    const int64_t n = dac_offsets[r + 1] - dac_offsets[r];
    constexpr bool SMALL = E == CF_NORM_REGS_SMALL;
    if (SMALL ? n > 256 * CF_NORM_REGS_SMALL : n <= 256 * CF_NORM_REGS_SMALL) return;          // the other instantiation's read
    const int16_t* v = dac + dac_offsets[r];
    float* out = x_out + win_offsets[r] * CF_T;
    const int64_t n_pad = (win_offsets[r + 1] - win_offsets[r]) * CF_T;
    if (n <= 0) {                                                      // (SMALL only)
        for (int64_t i = threadIdx.x; i < n_pad; i += blockDim.x) out[i] = 0.f;
        return;
    }
    if (n <= 256 * E) { normalize_read_regs<E>(v, (int)n, out, n_pad, hist); return; }
    if constexpr (!SMALL) {
        // longer than the register window: round 1's radix selection over global memory (exact as well)
        auto key16 = [](int16_t x) -> unsigned { return (unsigned)((int)x + 32768); };
        const int lo = radix_select(v, n, (n - 1) / 2, 8, key16, hist, sh);
        const int hi = (n & 1) ? lo : radix_select(v, n, n / 2, 8, key16, hist, sh);
        const int med2 = (lo - 32768) + (hi - 32768);
        auto keydev = [med2](int16_t x) -> unsigned { const int d = 2 * (int)x - med2; return (unsigned)(d < 0 ? -d : d); };
        const int dlo = radix_select(v, n, (n - 1) / 2, 9, keydev, hist, sh);
        const int dhi = (n & 1) ? dlo : radix_select(v, n, n / 2, 9, keydev, hist, sh);
        const double shift = 0.5 * (double)med2;
        const double scale = 0.25 * (double)(dlo + dhi);
        for (int64_t i = threadIdx.x; i < n_pad; i += blockDim.x)
            out[i] = i < n ? (float)(((double)v[i] - shift) / scale) : 0.f;
    }
}

// --------------------------------------------------------------------------------------------------------------------------
// Post-processing on bit masks.  A wave owns 64 consecutive 64-sample words of the packed sample axis -- 62 it writes and one halo
// word either side -- one word per lane:
//   1. lane L works out which bits of ITS word lie in the real part of a read (one binary search over the read table per WORD,
//      then a walk over the at most three reads a word can touch);
//   2. 64 coalesced loads + ballots turn the probabilities into threshold bits, word j landing in lane j;
//   3. with its neighbours' words to either side (one shuffle each) a lane erodes its 192-bit window by min_run - 1 (positions
//      where min_run ones begin) and dilates it back (every sample of such a run): shifts and ANDs / ORs in log2(min_run) steps.
//      Erosion results near the window's ends are incomplete, but they reach at most min_run - 1 <= 63 bits: never the middle word;
//   4. the middle word goes out as 64 label bytes (four 16-byte stores).
// No per-sample search, no per-sample walk; min_run from 1 to 64 (the reference uses 15), anything else takes postprocess_kernel.
// --------------------------------------------------------------------------------------------------------------------------
#define CF_POST_WORDS 62           // payload words per wave

struct cf_w192 { unsigned long long p, m, n; };      // three 64-sample words: bit b of a word = sample (word index * 64 + b)

__device__ __forceinline__ cf_w192 w192_shr(const cf_w192& x, int s) {     // towards lower sample indices, 0 < s < 64
    return {(x.p >> s) | (x.m << (64 - s)), (x.m >> s) | (x.n << (64 - s)), x.n >> s};
}
__device__ __forceinline__ cf_w192 w192_shl(const cf_w192& x, int s) {     // towards higher sample indices, 0 < s < 64
    return {x.p << s, (x.m << s) | (x.p >> (64 - s)), (x.n << s) | (x.m >> (64 - s))};
}

// SPANS: the run boundaries of the corrected labels leave with the same launch (cf_postprocess_spans: the "one kernel returning
// spans" of SURVEY.md 8f-1) -- a run starts where a bit is set and its lower neighbour is not (or a read begins), and ends likewise;
// the neighbours across the word's edges are the final bits 63 / 0 of the window's outer words, which are exact (see above: what is
// incomplete there never comes within min_run - 1 bits of the middle word).  A wave reserves room for all its boundaries with one
// global atomic per list (prefix sums of the lanes' popcounts), like spans_kernel; labels may then be NULL (not written at all).
template <bool SPANS>
__global__ __launch_bounds__(256) void postprocess_bits_kernel(const float* __restrict__ probs, const int64_t* __restrict__ read_offsets,
                                                               const int64_t* __restrict__ read_lengths, int64_t n_reads, int64_t total,
                                                               float threshold, int min_run, uint8_t* __restrict__ labels, int64_t max_runs,
                                                               int64_t* __restrict__ starts, int64_t* __restrict__ ends,
                                                               unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_words = (total + 63) >> 6;
    const int64_t w0 = chunk * CF_POST_WORDS;                              // first payload word of this wave
    if (w0 >= n_words) return;                                             // (whole wave: uniform)
    const int64_t w = w0 - 1 + lane;                                       // this lane's word
    const int64_t base = w * 64;
    // 1. which samples of my word are real samples of a read
    unsigned long long valid = 0, first = 0;                               // first: bits where a read begins (runs are cut there)
    if (w >= 0 && base < total) {
        int64_t lo = 0, hi = n_reads;                                      // largest r with read_offsets[r] <= base
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (read_offsets[mid] <= base) lo = mid; else hi = mid;
        }
        for (int64_t r = lo; r < n_reads; ++r) {
            const int64_t beg = read_offsets[r];
            if (beg >= base + 64) break;
            const int64_t end = beg + read_lengths[r];
            const int64_t a = beg > base ? beg - base : 0, b = (end < base + 64 ? end : base + 64) - base;
            if (b > a) valid |= (b - a >= 64 ? ~0ull : ((1ull << (b - a)) - 1ull)) << a;
            if (beg >= base) first |= 1ull << (beg - base);
        }
    }
    // 2. threshold bits: word j of the wave's window lands in lane j
    unsigned long long mine = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const int64_t i = (w0 - 1 + j) * 64 + lane;
        const float p = (i >= 0 && i < total) ? probs[i] : -1.f;
        const unsigned long long bal = __ballot(p >= threshold);
        mine = lane == j ? bal : mine;
    }
    mine &= valid;
    // 3. runs of >= min_run ones: erode by min_run - 1, dilate back
    cf_w192 x;
    x.m = mine;
    x.p = (unsigned long long)__shfl_up((long long)mine, 1);               // (lane 0 and lane 63 are halo lanes: their results are dropped)
    x.n = (unsigned long long)__shfl_down((long long)mine, 1);
    if (lane == 0) x.p = 0;
    if (lane == 63) x.n = 0;
    // A run lies inside the real part of ITS read (postprocess_kernel's contract).  Padding bits are 0 and separate the reads, but a
    // read packed WITHOUT padding touches the next one: the ones of `have` samples beginning at i join those beginning at i + have
    // only when no read begins at i + have.  (The last, overlapping step below needs no such mask: its two stretches share samples.)
    cf_w192 cut;
    cut.m = ~first;
    cut.p = ~(unsigned long long)__shfl_up((long long)first, 1);
    cut.n = ~(unsigned long long)__shfl_down((long long)first, 1);
    int have = 1;                                                          // x marks positions where `have` ones of one read begin
    while (2 * have <= min_run) {
        const cf_w192 s = w192_shr({x.p & cut.p, x.m & cut.m, x.n & cut.n}, have);
        x = {x.p & s.p, x.m & s.m, x.n & s.n};
        have *= 2;
    }
    if (have < min_run) {
        const cf_w192 s = w192_shr(x, min_run - have);
        x = {x.p & s.p, x.m & s.m, x.n & s.n};
    }
    have = 1;                                                              // x marks the first `have` samples of every qualifying run
    while (2 * have <= min_run) {
        const cf_w192 s = w192_shl(x, have);
        x = {x.p | s.p, x.m | s.m, x.n | s.n};
        have *= 2;
    }
    if (have < min_run) {
        const cf_w192 s = w192_shl(x, min_run - have);
        x = {x.p | s.p, x.m | s.m, x.n | s.n};
    }
    const bool payload = lane != 0 && lane != 63 && base < total;
    if constexpr (SPANS) {
        unsigned long long sm = 0, em = 0;
        if (payload) {
            // ... or where a read begins (an unpadded read's last run and the next read's first are two runs, not one)
            sm = x.m & (~((x.m << 1) | (x.p >> 63)) | ~cut.m);
            em = x.m & (~((x.m >> 1) | (x.n << 63)) | ((~cut.m >> 1) | (~cut.n << 63)));
        }
        const unsigned cs = (unsigned)__popcll(sm), ce = (unsigned)__popcll(em);
        unsigned ps = cs, pe = ce;                                         // inclusive prefix sums over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned os = (unsigned)__shfl_up((int)ps, d), oe = (unsigned)__shfl_up((int)pe, d);
            if (lane >= d) { ps += os; pe += oe; }
        }
        const unsigned ts = (unsigned)__shfl((int)ps, 63), te = (unsigned)__shfl((int)pe, 63);
        unsigned long long bs = 0, be = 0;
        if (lane == 0) {
            if (ts) bs = atomicAdd(&counts[0], (unsigned long long)ts);
            if (te) be = atomicAdd(&counts[1], (unsigned long long)te);
        }
        bs = (unsigned long long)__shfl((long long)bs, 0);
        be = (unsigned long long)__shfl((long long)be, 0);
        unsigned long long k = bs + ps - cs;
        while (sm) {
            const int b = __builtin_ctzll(sm);
            sm &= sm - 1;
            if ((int64_t)k < max_runs) starts[k] = base + b;
            ++k;
        }
        k = be + pe - ce;
        while (em) {
            const int b = __builtin_ctzll(em);
            em &= em - 1;
            if ((int64_t)k < max_runs) ends[k] = base + b + 1;             // exclusive, like spans_kernel
            ++k;
        }
    }
    // 4. the middle word as label bytes
    if (!payload || labels == nullptr) return;
    const unsigned long long bits = x.m;
    if (base + 64 <= total) {
        uint4* dst = reinterpret_cast<uint4*>(labels + base);              // base is a multiple of 64, labels a device allocation: 16-byte aligned
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned h = (unsigned)(bits >> (16 * q)) & 0xffffu;
            uint4 o;                                                       // 4 bits -> 4 bytes of 0 / 1: the multiply places bit k at bit 8 k
            o.x = (((h >> 0) & 0xfu) * 0x00204081u) & 0x01010101u;
            o.y = (((h >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
            o.z = (((h >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
            o.w = (((h >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
            dst[q] = o;
        }
    } else {
        for (int b = 0; base + b < total; ++b) labels[base + b] = (uint8_t)((bits >> b) & 1ull);
    }
}
