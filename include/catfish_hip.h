/*
 * catfish_hip.h -- C ABI of the MI355X-native homopolymer-calling forward pass.
 *
 * This is the drop-in boundary for ONE call of the reference:
 *
 *     confidences = self.sess.run(self.predictions,
 *                                 feed_dict={self.x: input_x, self.p_dropout: 1.0})
 *                                         (reference catfish/models/rnn_class.py:214-216)
 *
 * i.e. the whole TensorFlow graph built by ResNetRNN.network_layer
 * (catfish/models/resnet_class.py:17-25,44-82), RNN.network_layer
 * (catfish/models/rnn_class.py:165-175), RNN.output_layer (:178-183) and
 * RNN.compute_accuracy (:82-88, self.predictions = sigmoid(logits)).
 * The reference has no native code; the entry points below are what a cgo /
 * ctypes / N-API binding of that one call binds.  Plain pointers and sizes
 * only -- no torch / TensorFlow types.
 *
 * Threading: one cf_model per device per host thread; cf_infer is
 * asynchronous on the given HIP stream and keeps no global state.
 * Errors: 0 = ok, negative = failure; cf_last_error() returns the message of
 * the calling thread's last failure.
 */
#ifndef CATFISH_HIP_H
#define CATFISH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_OK 0
#define CF_ERR_INVALID -1     /* bad argument / unsupported geometry (Python raises ValueError) */
#define CF_ERR_HIP -2         /* a HIP runtime call failed */
#define CF_ERR_NOMEM -3
#define CF_ERR_IO -4          /* a file could not be opened / written (Python raises OSError); host entry points only */

#define CF_WINDOW 35          /* rnn_class.py:27 (self.window) */

/* Bumped whenever a signature or a struct of this header changes; cf_abi_version() returns the value the library was built
 * with, so a binding can refuse a stale libcatfish_hip.so instead of calling it with the wrong arguments. */
#define CF_ABI_VERSION 6

/* Arithmetic of the biGRU layers (the residual blocks, the hidden state, the gates'
 * sigmoid/tanh and all accumulation are fp32 in every mode). */
#define CF_PREC_FP32 0        /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32); default            */
#define CF_PREC_BF16X3 1      /* operands split hi+lo bf16, 3 bf16 MFMAs per product (~2^-17) */
#define CF_PREC_BF16 2        /* operands rounded to bf16 (BASELINE config 4)                 */

/* Hyper-parameters: the keys of ResNetRNN.txt parsed by
 * neural_network.retrieve_hyperparams (catfish/neural_network.py:37-67)
 * that shape the forward graph. */
typedef struct cf_hparams {
    int32_t layer_size;          /* GRU units per direction (rnn_class.py:16): a multiple of 16, at most 256.
                                    64 with layer_size_res 32 (the shipped checkpoint) runs on the tuned kernels,
                                    every other geometry on the any-size kernels (fp32 only)            */
    int32_t n_layers;            /* stacked bidirectional layers (rnn_class.py:17)   */
    int32_t layer_size_res;      /* conv channels (resnet_class.py:11)               */
    int32_t n_layers_res;        /* residual blocks (resnet_class.py:10); 0 = RNN    */
    int32_t window;              /* must be 35                                       */
    float bn_epsilon;            /* tf.layers.batch_normalization epsilon (1e-3)     */
    int64_t max_windows_per_pass;/* scratch capacity per slot; longer inputs are chunked */
    int32_t n_streams;           /* scratch slots + internal streams that overlap the
                                    sub-batches of one call (0 = default 1 = none)       */
    int32_t precision;           /* CF_PREC_*                                          */
    int32_t fuse_layers;         /* all biGRU layers in ONE launch with dynamic tile queues (fp32,
                                    n_layers <= 3): 0 = auto (only passes of >= 6 full-chip rounds,
                                    where it measures +5 %), 1 = always, -1 = never          */
} cf_hparams;

/* One conv1d + batch_normalization pair, TF layout
 * (resnet_class.py:60-61 / 64-65 / 69-70 / 74-75). */
typedef struct cf_conv_bn {
    const float* kernel;          /* [ksize, cin, layer_size_res] */
    const float* bias;            /* [layer_size_res] */
    const float* gamma;           /* [layer_size_res] */
    const float* beta;
    const float* moving_mean;
    const float* moving_variance;
    int32_t ksize;
    int32_t cin;
} cf_conv_bn;

/* tf.contrib.rnn.GRUCell variables of one direction of one layer
 * (rnn_class.py:146); kernel rows [0,cin) multiply x, rows [cin,cin+H) multiply h. */
typedef struct cf_gru_dir {
    const float* gates_kernel;     /* [cin + H, 2H]  columns [0,H) = r, [H,2H) = u */
    const float* gates_bias;       /* [2H] */
    const float* candidate_kernel; /* [cin + H, H] */
    const float* candidate_bias;   /* [H] */
    int32_t cin;
} cf_gru_dir;

/* All 74 inference tensors of the checkpoint, host pointers, TF layout. */
typedef struct cf_weights {
    const cf_conv_bn* conv;        /* 4 * n_layers_res entries, graph order:
                                      per block: shortcut, first, middle (k=3), last */
    const cf_gru_dir* gru;         /* 2 * n_layers entries: [layer][fw, bw] */
    const float* dense_kernel;     /* final_fully_connected/kernel [2H, 1] (rnn_class.py:179) */
    const float* dense_bias;       /* [1] */
} cf_weights;

typedef struct cf_model cf_model;

/* Replaces neural_network.load_network + RNN.restore_network
 * (catfish/neural_network.py:26-34, rnn_class.py:191-198): folds BN into the
 * convs, re-tiles every matrix into MFMA A-fragment order and uploads it. */
int cf_model_create(const cf_weights* w, const cf_hparams* hp, int device, cf_model** out);
void cf_model_destroy(cf_model* m);

/* Replaces RNN.infer's sess.run (rnn_class.py:213-219).
 * x: device pointer, [n_windows, 35] fp32 (window-major, as reshape_input
 * produces, catfish/infer.py:108-124).  probs: device pointer,
 * [n_windows * 35] fp32 = sigmoid(logits), same order as the reference's
 * flattened output.  stream: hipStream_t (NULL = default stream). */
int cf_infer(cf_model* m, const float* x, int64_t n_windows, float* probs, void* stream);

/* Same with host buffers (synchronous; H2D / D2H included). */
int cf_infer_host(cf_model* m, const float* x, int64_t n_windows, float* probs);

/* The same two calls with the pre-sigmoid logits as a second output (``self.logits`` of
 * RNN.output_layer, catfish/models/rnn_class.py:178-183): the reference evaluates its validation
 * loss on the logits (tf.losses.sigmoid_cross_entropy, rnn_class.py:74-79,236-237), which the
 * saturated fp32 probabilities cannot reproduce.  probs or logits may be NULL (not both). */
int cf_infer_logits(cf_model* m, const float* x, int64_t n_windows, float* probs, float* logits, void* stream);
int cf_infer_host_logits(cf_model* m, const float* x, int64_t n_windows, float* probs, float* logits);

/* Sticky device-side error of earlier asynchronous launches on this model (today: a bounded wait of the
 * fused biGRU launch that timed out, which invalidates that launch's results).  Call it after
 * synchronising the stream a cf_infer was queued on; CF_OK, or CF_ERR_HIP with the message in
 * cf_last_error().  cf_infer / cf_infer_host also refuse to run while it is set. */
int cf_check_error(cf_model* m);
/* Reset that flag after the caller has dropped the results of the failed launch: the model is usable again (a fused
 * launch re-initialises its queues and flags every time; nothing else is stale). */
int cf_clear_error(cf_model* m);

/* Launch-regime switch points of this model on its device, for callers and tests that need to know which
 * kernels a call of n_windows uses (all in windows):
 *   out[0] = CUs of the device
 *   out[1] = largest call whose biGRU x projection is hoisted onto the idle CUs (latency mode, fp32)
 *   out[2] = largest call served by the cooperative latency-mode biGRU kernels (fp32); larger calls use the
 *            throughput kernels (one wave per tile)
 *   out[3] = smallest call whose biGRU layers go out as ONE fused launch when fuse_layers = 0 (auto);
 *            0 when this model never fuses */
int cf_launch_regimes(const cf_model* m, int64_t out[4]);

/* Read-level post-processing on device, replacing class_from_threshold +
 * correct_short (catfish/infer.py:128-138,174-198).  Reads are packed back to
 * back INCLUDING their zero padding (infer.py:31-38): read r owns samples
 * [read_offsets[r], read_offsets[r+1]) of probs, of which the first
 * read_lengths[r] are real (the reference trims the padding at infer.py:47).
 * labels[i] = 1 iff probs[i] >= threshold and i lies in a positive run of
 * length >= min_run inside the real part of its read; padding gets 0.
 * read_offsets: device int64[n_reads + 1]; read_lengths: device int64[n_reads];
 * total_samples = read_offsets[n_reads] (passed by value so that the call
 * stays asynchronous); labels: device uint8[total_samples]. */
int cf_postprocess(cf_model* m, const float* probs, const int64_t* read_offsets,
                   const int64_t* read_lengths, int64_t n_reads, int64_t total_samples,
                   float threshold, int32_t min_run, uint8_t* labels, void* stream);

/* Run boundaries of the corrected labels on device (the run-length half of hp_in_pred,
 * catfish/infer.py:141-162).  labels: device uint8[total_samples] as written by cf_postprocess
 * (padding = 0, so runs never cross reads -- this call sees labels only: where a read is packed WITHOUT padding,
 * read_lengths[r] == read_offsets[r+1] - read_offsets[r], and its last sample and the next read's first are both positive,
 * they come out as one run; catfish/infer.py:31-36 pads every read by at least one sample, and cf_postprocess_spans, which
 * has the read table, cuts such runs at the boundary).  starts / ends: device int64[max_runs], receive the packed
 * positions of every run's first sample and one-past-last sample in arbitrary order (sort both
 * ascending: the k-th start pairs with the k-th end); counts: device uint64[2] = number of starts and
 * of ends found (may exceed max_runs, in which case the lists are truncated). */
int cf_spans(cf_model* m, const uint8_t* labels, int64_t total_samples, int64_t max_runs,
             int64_t* starts, int64_t* ends, uint64_t* counts, void* stream);

/* cf_postprocess and cf_spans as ONE launch: threshold, correct_short and the run boundaries of the corrected labels (catfish/infer.py:
 * 128-138, 174-198 and the run-length half of hp_in_pred, :141-162) straight from the probabilities.  Arguments as in the two calls;
 * labels may be NULL when only the run lists are wanted (then they are never written).  A run lies inside the real part of ITS read,
 * also when reads are packed without padding: adjacent positive runs of two such reads are two runs.  counts is zeroed by the call (on
 * the stream). */
int cf_postprocess_spans(cf_model* m, const float* probs, const int64_t* read_offsets, const int64_t* read_lengths, int64_t n_reads,
                         int64_t total_samples, float threshold, int32_t min_run, uint8_t* labels, int64_t max_runs,
                         int64_t* starts, int64_t* ends, uint64_t* counts, void* stream);

/* Signal ingest on device, replacing normalize_raw_signal + the padding / reshape of
 * infer_class_from_signal (catfish/infer.py:96-105, 31-43) for many reads at once.
 * dac: device int16, the reads' raw DAC samples back to back (after the leader trim of
 * process_signal, infer.py:87-90); dac_offsets: device int64[n_reads + 1] sample offsets into
 * dac; win_offsets: device int64[n_reads + 1] first window of every read in the packed output;
 * x_out: device fp32 [win_offsets[n_reads], 35], receives (raw - median) / median(|raw - median|)
 * followed by each read's zero padding.  Results are bit-identical to numpy's float64
 * normalisation cast to float32. */
int cf_normalize(cf_model* m, const int16_t* dac, const int64_t* dac_offsets,
                 const int64_t* win_offsets, int64_t n_reads, float* x_out, void* stream);

/* Pipeline tail on the HOST (no device work, no cf_model): what the reference's per-file loop does with the spans of a read
 * (catfish/catfish:57-82 and center_hp, :121-135) for many reads held as flat tables.  Read r owns rows
 * [bounds[r], bounds[r+1]) of a table; span_* = the [start - 11, end + 16] spans of infer_class_from_signal; lengths[r] =
 * its second return value.  Output: hp_* = the merged, centred chunks (reads without spans get none and are absent from the
 * reference's hp_dict), nonhp_* = the complement (reads without spans get the single row (0, length), which the reference
 * stores as [([(0, len), len])], catfish:82).  Quirks kept: the merged list aliases the span lists, spans are edited in
 * place, and `hp_positions[i - 1]` at i = 0 is the read's last span.
 * Capacities (rows): hp >= n_spans + n_reads, nonhp >= n_spans + 2 n_reads always suffice. */
int cf_chunks_from_spans(const int64_t* span_bounds, const int64_t* span_start, const int64_t* span_end,
                         const int64_t* lengths, int64_t n_reads, int64_t chunk_size,
                         int64_t* hp_bounds, int64_t* hp_start, int64_t* hp_end, int64_t hp_capacity,
                         int64_t* nonhp_bounds, int64_t* nonhp_start, int64_t* nonhp_end, int64_t nonhp_capacity);
/* The JSON members `"name": [[a, b], ...]` of such a table joined by ", " (json.dump's text between the braces) for the
 * reads that own rows (whole_read == NULL: the reference's hp_dict) or for every read, "[]" when it owns none (whole_read
 * given: its nonhp_dict, where whole_read[r] != 0 marks the reads to be written in the no-homopolymer form [[[a, b], b]]).
 * keys = the names as JSON string literals back to back, key_bounds their byte offsets.  Returns bytes written (< 0: error;
 * capacity of sum(key bytes) + 48 per row + 40 per read always suffices). */
int64_t cf_chunks_json(const char* keys, const int64_t* key_bounds, int64_t n_reads, const int64_t* bounds,
                       const int64_t* start, const int64_t* end, const uint8_t* whole_read, char* out, int64_t capacity);

/* Ingest on the HOST (no device work, no cf_model): the file read of the reference's per-file loop (catfish/catfish:50-56 ->
 * infer.process_signal, infer.py:77-93) for the read format this image can hold -- one-dimensional C-order little-endian
 * int16 .npy files (DAC codes after the leader trim; there is no HDF5 library here) -- for MANY files at once: a pool of
 * n_threads host threads (<= 0: 4) reads them straight into ONE caller-owned buffer, back to back in the order given.
 * paths: the names as NUL-terminated strings back to back, path_bounds[n_files + 1] their byte offsets; out / capacity: the
 * int16 buffer (e.g. pinned staging memory) and its size in samples; lengths[n_files] receives every read's sample count,
 * *total (may be NULL) their sum (also when they do not fit capacity).  CF_ERR_INVALID names the first file that is
 * missing or is not such an array in cf_last_error(): the caller then takes its general loader for that batch.  A file is
 * judged by its first 4 KiB (its header) and its size before it is read: only regular files whose header matches their
 * length are read whole.  CF_ERR_NOMEM: the tables or a read did not fit host memory (no C++ exception leaves the call). */
int cf_load_npy_int16(const char* paths, const int64_t* path_bounds, int64_t n_files, int16_t* out, int64_t capacity,
                      int64_t* lengths, int64_t* total, int32_t n_threads);

/* Sizes on disk of n_files entries of ONE directory (host code, no device work): the listing step of the reference's per-file
 * loop (catfish/catfish:49-50) -- the sizes cut the sorted file list into blocks of equal work per rank and into batches before
 * anything is read.  names: the entry names (relative to dir) as NUL-terminated strings back to back, name_bounds[n_files + 1]
 * their byte offsets; sizes[n_files] receives st_size (regular files, and whatever else the directory holds: it is judged when
 * its turn to be read comes).  fstatat relative to one directory handle from n_threads host threads (<= 0: 4).  CF_ERR_INVALID
 * names the first entry that cannot be stat-ed (e.g. removed since the listing) in cf_last_error(). */
int cf_stat_files(const char* dir, const char* names, const int64_t* name_bounds, int64_t n_files, int64_t* sizes, int32_t n_threads);

/* The listing of the input directory as an object (host code; catfish/catfish:49-50 `input_files = os.listdir(input_dir)`).  Every
 * rank of a sharded job needs the same ORDER of all names, the sizes of one block of them and the names of the block it ends up
 * classifying -- not a host-language string per entry of a 100 000-file directory on each of 8 ranks.
 *   cf_listing_open   reads all entry names of dir (no "." / ".."), orders them bytewise (= sorted() of the decoded names whenever
 *                     they are valid UTF-8), keeps them; *n_entries their number, digest[2] 128 bits over the ordered names (what
 *                     the ranks compare to make sure they saw the same directory)
 *   cf_listing_sizes  st_size of entries [lo, hi) of that order (fstatat from n_threads host threads, as cf_stat_files)
 *   cf_listing_names  the names of entries [lo, hi), NUL-terminated, back to back into out (capacity bytes) with bounds[hi - lo + 1]
 *                     their offsets; *needed (may be NULL) the bytes they take; out == NULL: size query only
 *   cf_listing_from_names  the same object from names somebody else read and ordered (n_entries NUL-terminated names back to back,
 *                     strictly ascending bytewise -- checked): rank 0 reads the directory ONCE and broadcasts what cf_listing_names
 *                     gave it, because concurrent readdirs of one directory serialise on some file systems (overlayfs: 8 ranks x
 *                     100 000 entries 102 ms each, one reader 13 ms)
 * Not thread-safe per listing; independent listings are. */
typedef struct cf_listing cf_listing;
int cf_listing_open(const char* dir, cf_listing** out, int64_t* n_entries, uint64_t* digest);
int cf_listing_from_names(const char* dir, const char* names, int64_t n_bytes, int64_t n_entries, cf_listing** out, uint64_t* digest);
int cf_listing_sizes(const cf_listing* l, int64_t lo, int64_t hi, int64_t* sizes, int32_t n_threads);
int cf_listing_names(const cf_listing* l, int64_t lo, int64_t hi, char* out, int64_t capacity, int64_t* bounds, int64_t* needed);
void cf_listing_close(cf_listing* l);
/* cf_load_npy_int16 for entries [lo, hi) of a listing, opened relative to the listing's directory: no path string per file is ever
 * built by the caller.  Every entry must be named *.npy; same results, errors and buffers as cf_load_npy_int16. */
int cf_listing_load_npy_int16(const cf_listing* l, int64_t lo, int64_t hi, int16_t* out, int64_t capacity, int64_t* lengths,
                              int64_t* total, int32_t n_threads);

/* The split step, catfish/catfish:85-92 -> split_f5.split_signal (catfish/split_f5.py:8-81), for entries [lo, hi) of a listing that are
 * one-dimensional little-endian int16 .npy reads.  Row r of the two CSR chunk tables (cf_chunks_from_spans' outputs; bounds[hi - lo + 1])
 * belongs to entry lo + r.  Every read WITH homopolymer rows (`for read in hp_dict`, catfish/catfish:88) is read once and cut:
 * signal[start:end] (Python slice rules: a bound below zero counts from the end, both are clamped to the read) of its HP rows goes to
 * <hp_dir>/<stem>_<k>.npy, of its non-HP rows to <nonhp_dir>/<stem>_<k>.npy -- <stem> = the entry's name up to its FIRST dot
 * (split_f5.py:39,65), k = 0, 1, ... over the HP rows and on into the non-HP rows (:34,57,81) -- each file byte for byte numpy.save of that
 * int16 slice.  (The reference writes a gzip-9 HDF5 copy of the input per piece; HDF5 is outside this path.)  Reads sharing a stem overwrite
 * each other's pieces in listing order, as in the reference.  n_threads host threads (<= 0: 4).  counts (may be NULL): int64[4] = reads cut,
 * HP files, non-HP files, samples written.  CF_ERR_INVALID names the first entry that is not such a read (nothing is rolled back: pieces are
 * rewritten whole by whoever repeats the step); CF_ERR_IO names the first file that could not be written, with the system's reason. */
int cf_listing_split_npy_int16(const cf_listing* l, int64_t lo, int64_t hi, const int64_t* hp_bounds, const int64_t* hp_start,
                               const int64_t* hp_end, const int64_t* nonhp_bounds, const int64_t* nonhp_start, const int64_t* nonhp_end,
                               const char* hp_dir, const char* nonhp_dir, int32_t n_threads, int64_t* counts);

/* CRC-32C (Castagnoli) of n bytes, continuing from crc (0 to start): the checksum of leveldb table blocks and of every tensor in a
 * TensorFlow checkpoint-V2 bundle, which the checkpoint reader verifies like tf.train.Saver does (catfish/models/rnn_class.py:191-198). */
uint32_t cf_crc32c(const void* data, int64_t n, uint32_t crc);

/* Training support (BASELINE config 5; the reference's RNN.train_network, catfish/models/rnn_class.py:201-210,
 * differentiates this graph with TensorFlow's autodiff).  One bidirectional GRU layer at a time, fp32 MFMA,
 * on device buffers in the kernels' fragment layout [tile][t][mtile][lane][4] (tile = 16 windows; element
 * (lane, reg) of M-tile m = feature 16m + 4(lane>>4) + reg of window lane&15):
 *   x_frag   [tiles][35][cin/16][64][4]      layer input (cin = 32 for layer 0, 128 above)
 *   y_frag   [tiles][35][8][64][4]           layer output (M-tiles 0-3 forward, 4-7 backward direction)
 *   stash    [tiles][35][2][12][64][4]       activated r, u gates and candidate c of every step
 *   dy_frag  like y_frag                     gradient of the loss w.r.t. the layer output
 *   dy2_frag like y_frag, or NULL            second addend of that gradient (the other direction's dx slab of the
 *                                            layer above, so no separate add pass is needed)
 *   dy_scale like y_frag, or NULL            per-element factor applied to dy (+ dy2): the output-dropout mask
 *                                            divided by keep_prob (DropoutWrapper, rnn_class.py:151-154)
 *   dx_frag  [2][tiles][35][cin/16][64][4]   gradient w.r.t. the layer input, one slab per direction (add them)
 *   da       like stash                      pre-activation gradients da_r, da_u (0-7), da_c (8-11); the weight
 *                                            gradients are dW = A^T dA (cf_gru_train_wgrad)
 * wpack / wpack_bwd: device pointers to the re-tiled weights of BOTH directions ([2][n_floats]), owned by the
 * caller.  cf_gru_pack_map returns the gather map of that re-tiling for one direction, so a trainer can
 * rebuild them on device after every optimizer step: packed[i] = src[idx[i]] * scale[i] with
 * src = [gates_kernel | candidate_kernel | gates_bias | candidate_bias | 0.0] (TF layout, flattened).
 * Call it with idx = scale = NULL to query n_floats. */
int cf_gru_pack_map(int32_t cin, int32_t backward, int32_t* idx, float* scale, int64_t capacity, int64_t* n_floats);
int cf_gru_train_forward(cf_model* m, int32_t cin, const float* wpack, const float* x_frag, float* y_frag,
                         float* stash, int64_t n_windows, void* stream);
int cf_gru_train_backward(cf_model* m, int32_t cin, const float* wpack_bwd, const float* y_frag, const float* stash,
                          const float* dy_frag, const float* dy2_frag, const float* dy_scale, float* dx_frag, float* da,
                          int64_t n_windows, void* stream);
/* One biGRU layer of ANY geometry for training (layer_size a multiple of 16 up to 256; the reference's hyper-parameter search,
 * networks/train_validate.py:66-111, draws 16..256).  These two run the serial part on the any-size kernels (csrc/generic.hpp);
 * everything that is a plain GEMM over all (window, step) pairs -- the input gradient W_x^T da and the weight gradients
 * [x; h]^T da -- is the caller's (catfish_amd/anysize_train.py uses library GEMMs).  Buffers are device pointers, fragment
 * layout, n_windows a multiple of 16:
 *   wpack   [2 dirs][3: r, u, c][H/16][cin_blocks + H/16][64][4]   A fragments, r / u scaled by -log2(e), c by 2 log2(e)
 *   bpack   [2][3][H/16][64][4]                                    biases, same scaling
 *   x_frag  [tiles][35][cin_blocks][64][4]    y_frag [tiles][35][2 H/16][64][4] (forward features first)
 *   stash   [tiles][35][2][3: r, u, c][H/16][64][4]                 activated gates, written by the forward
 *   wtpack  [2 dirs]{ Wc_h^T [H/16][H/16][64][4], Wg_h^T [H/16][2 H/16][64][4] }   unscaled
 *   dy_frag like y_frag (dropout already applied by the caller); da like stash: dL/d(pre-activation) of r, u, c. */
int cf_gru_anysize_train_forward(cf_model* m, int32_t layer_size, int32_t cin_blocks, const float* wpack, const float* bpack,
                                 const float* x_frag, float* y_frag, float* stash, int64_t n_windows, void* stream);
int cf_gru_anysize_train_backward(cf_model* m, int32_t layer_size, const float* wtpack, const float* y_frag, const float* stash,
                                  const float* dy_frag, float* da, int64_t n_windows, void* stream);
/* The same two calls with the layer's OUTPUT dropout (DropoutWrapper(output_keep_prob), rnn_class.py:151-154) done inside the
 * kernels, no mask tensor: whether an output element is kept is a hash of (seed, layer, *step_count, element index).  The
 * forward additionally writes y_drop_frag = y * mask / keep_prob (what the next layer or the dense head reads; y_frag itself stays
 * un-dropped for the recurrence and the backward pass); the backward applies the same factor to the incoming gradient when
 * dy_scale is NULL and keep_prob < 1.  step_count: device double[1] (the optimizer's step counter, so every step draws a new mask
 * even under graph replay) or NULL.  cf_dropout_scale writes that factor tensor out (tests / tools only). */
int cf_gru_train_forward_dropout(cf_model* m, int32_t cin, const float* wpack, const float* x_frag, float* y_frag, float* stash,
                                 int64_t n_windows, float* y_drop_frag, float keep_prob, uint32_t seed, int32_t layer,
                                 const double* step_count, void* stream);
int cf_gru_train_backward_dropout(cf_model* m, int32_t cin, const float* wpack_bwd, const float* y_frag, const float* stash,
                                  const float* dy_frag, const float* dy2_frag, const float* dy_scale, float* dx_frag, float* da,
                                  int64_t n_windows, float keep_prob, uint32_t seed, int32_t layer, const double* step_count,
                                  void* stream);
int cf_dropout_scale(cf_model* m, float keep_prob, uint32_t seed, int32_t layer, const double* step_count, int64_t n_windows,
                     float* scale_frag, void* stream);
/* dW of one biGRU layer from the fragment buffers (the weight-gradient half of optimizer.minimize(loss),
 * catfish/models/rnn_class.py:62-71): grads = per direction [ gates kernel [cin+64,128] | gates bias [128] |
 * candidate kernel [cin+64,64] | candidate bias [64] ] in TensorFlow layout, 2 directions back to back.
 * workspace: cf_gru_wgrad_workspace_floats(m, cin, n_windows) floats of device scratch (0 = bad arguments). */
int64_t cf_gru_wgrad_workspace_floats(cf_model* m, int32_t cin, int64_t n_windows);
int cf_gru_train_wgrad(cf_model* m, int32_t cin, const float* x_frag, const float* y_frag, const float* stash, const float* da,
                       int64_t n_windows, float* workspace, int64_t workspace_floats, float* grads, void* stream);

/* Residual conv stack for training (forward catfish/models/resnet_class.py:44-82 with a stash of the conv outputs;
 * backward = its share of optimizer.minimize(loss), rnn_class.py:62-71).  All pointers are device pointers.
 *   params  cf_res_train_param_floats(n_blocks) floats: per conv+BN unit (4 per block, kernel widths 1,1,3,1)
 *           kernel [k][cin][32] | bias[32] | gamma[32] | beta[32] | moving_mean[32] | moving_variance[32], TF layouts,
 *           units back to back; cin = 1 for the first two units, 32 afterwards
 *   x       [n_windows][35] fp32            z_stash [4 n_blocks][n_windows*35][32]     out / d_out [n_windows*35][32]
 *   grads   same offsets as params (kernel, bias, gamma, beta gradients; the moving statistics get 0)
 *   workspace  cf_res_train_workspace_floats(n_blocks, n_windows) floats of scratch (per-workgroup partial sums,
 *           added in a fixed order) */
int64_t cf_res_train_param_floats(int32_t n_blocks);
int64_t cf_res_train_workspace_floats(int32_t n_blocks, int64_t n_windows);
int cf_res_train_forward(cf_model* m, int32_t n_blocks, const float* params, const float* x, float* z_stash, float* out,
                         int64_t n_windows, void* stream);
int cf_res_train_backward(cf_model* m, int32_t n_blocks, const float* params, const float* x, const float* z_stash,
                          const float* d_out, float* workspace, int64_t workspace_floats, float* grads, int64_t n_windows,
                          void* stream);

/* Dense head + loss of the training step, forward and backward in one pass (RNN.output_layer + RNN.compute_loss,
 * catfish/models/rnn_class.py:178-183,74-79: final_fully_connected followed by tf.losses.sigmoid_cross_entropy, mean over
 * all n_windows * 35 elements).  y_frag: the dense layer's input = last biGRU layer's output after output dropout, fragment
 * layout [tiles][35][8][64][4]; labels: device fp32 [n_windows][35]; dense_kernel [128], dense_bias [1]: device pointers.
 * Writes dy_frag (d loss / d y_frag, same layout), grads = d kernel [128] | d bias [1], loss[0] = the mean loss, and the
 * logits [n_windows][35] when that pointer is not NULL.  workspace: cf_train_head_workspace_floats(m, n_windows) floats. */
int64_t cf_train_head_workspace_floats(cf_model* m, int64_t n_windows);
int cf_train_head(cf_model* m, const float* y_frag, const float* dense_kernel, const float* dense_bias, const float* labels,
                  int64_t n_windows, float* dy_frag, float* logits, float* workspace, int64_t workspace_floats, float* grads,
                  float* loss, void* stream);

/* One optimizer step over ALL variables at once (optimizer.minimize(loss), rnn_class.py:62-71): params, grads and the two slot
 * variables are flat device buffers of one layout, n floats each.  kind 0 = tf.train.RMSPropOptimizer(lr) (decay 0.9,
 * momentum 0, epsilon 1e-10; slot1 = rms, initial value 1; slot2 = momentum), kind 1 = tf.train.AdamOptimizer(lr) (beta 0.9 /
 * 0.999, epsilon 1e-8; slot1 = m, slot2 = v).  step_count: device double[1], steps taken so far; read by the update (Adam's bias
 * correction) and incremented by this call.  When n_packed > 0 the updated weights are re-tiled for the biGRU kernels in the
 * same call: packed[i] = params[pack_idx[i]] * pack_scale[i] (indices from cf_gru_pack_map, rebased into params). */
int cf_opt_step(cf_model* m, int32_t kind, float* params, const float* grads, float* slot1, float* slot2, int64_t n, float lr,
                double* step_count, const int32_t* pack_idx, const float* pack_scale, float* packed, int64_t n_packed, void* stream);

/* Per-kernel device timing (HIP events on the launch stream) for bench.py's
 * roofline report.  cf_profile_enable(m, N) makes every N-th cf_infer call
 * (N = 1: every call; 0 = off) record events around each of its kernels; cf_profile_read synchronises and returns, for
 * kernel slot k, the accumulated milliseconds and launch count since the
 * last cf_profile_reset. */
#define CF_PROF_SLOTS 12
int cf_profile_enable(cf_model* m, int on);
int cf_profile_reset(cf_model* m);
int cf_profile_read(cf_model* m, double ms[CF_PROF_SLOTS], int64_t launches[CF_PROF_SLOTS]);
const char* cf_profile_slot_name(int slot);

/* Debug/test hook: copy an intermediate activation of the LAST pass to the
 * host in natural [n_windows, 35, features] order.  stage: 0..n_layers_res-1
 * = residual block outputs, n_layers_res + l = GRU layer l output (not
 * available for the last layer, whose output only exists as logits). */
int cf_debug_stage(cf_model* m, int stage, int64_t n_windows, float* out_host);

/* Which card HIP device `device` of this process is -- the reference's per-file loop (catfish/catfish:50-82) shards over one
 * process per GPU, and each rank binds itself to the host CPUs next to ITS card (catfish_amd/placement.py) and says in the
 * benchmark line which card it drove.  pci_bus_id (>= 13 bytes, may be NULL) receives "dddd:bb:dd.f", the name of the
 * device's directory under /sys/bus/pci/devices (upper- or lower-case hex, as the runtime prints it); uuid_hex (>= 33 bytes,
 * may be NULL) the 16-byte device UUID as 32 hex digits.  Opens the HIP runtime (not a context on the device). */
int cf_device_identity(int device, char* pci_bus_id, int64_t bus_cap, char* uuid_hex, int64_t uuid_cap);

int64_t cf_workspace_bytes(const cf_model* m);
const char* cf_last_error(void);
const char* cf_version(void);
int cf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CATFISH_HIP_H */
