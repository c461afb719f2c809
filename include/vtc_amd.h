/* vtc_amd.h -- C ABI of the MI355X-native VTC hot path (libvtc_amd.so, gfx950).
 *
 * The reference (danieldager/VTC) has no FFI: its boundary for this path is the Python import
 * surface of the third-party package `segma` (src/pipeline/vtc.py:41-47, src/core/intervals.py:10).
 * Each entry point below names the segma call (and the reference call site) it stands behind;
 * vtc_amd/segma_compat/ is the Python side that binds them with ctypes.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * vtc_status; vtc_last_error() gives the message.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Device pointers are marked [dev], host pointers [host].
 * The library never takes ownership of caller memory.  One vtc_model is bound to one GPU and is not
 * thread-safe; use one model per worker (the reference runs one process per GPU, slurm/vtc.slurm:9-10).
 */
#ifndef VTC_AMD_H
#define VTC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTC_ABI_VERSION 1
#define VTC_MAX_LABELS 8
#define VTC_N_LAYERS 12

typedef enum {
  VTC_OK = 0,
  VTC_ERR_ARG = -1,       /* bad argument / unsupported spec */
  VTC_ERR_HIP = -2,       /* a HIP runtime call failed */
  VTC_ERR_NO_GPU = -3,    /* no gfx950 device visible */
  VTC_ERR_CAPACITY = -4,  /* caller buffer too small; the needed size is reported */
  VTC_ERR_SHORT = -5,     /* audio shorter than the 400-sample receptive field */
  VTC_ERR_NONFINITE = -6  /* NaN / inf logits: an activation left the range of the 16-bit storage type upstream */
} vtc_status;

typedef struct vtc_model vtc_model; /* opaque */

/* Model geometry that segma reads from its YAML config (cfg.audio.chunk_duration_s, cfg.data.classes;
 * src/pipeline/vtc.py:183-195).  The conv stack is the fixed HuBERT-base geometry named at
 * src/analysis/chunk_artifact.py:43-48 (kernels 10,3,3,3,3,2,2; strides 5,2,2,2,2,2,2). */
typedef struct {
  int32_t abi_version;    /* VTC_ABI_VERSION */
  int32_t n_labels;       /* 1..VTC_MAX_LABELS */
  int32_t window_samples; /* chunk_duration_s * 16000; 400 <= .. <= 655759 (T <= 2048 frames) */
  int32_t max_batch;      /* windows per forward pass (segma's batch_size) */
  int32_t mfma_dtype;     /* 0 = bf16 operands, 1 = f16 operands (fp32 accumulate either way) */
  int32_t reserved[3];
} vtc_spec;

/* fp32 host tensors in PyTorch-canonical layouts (names: HF HubertModel state dict). */
typedef struct {
  const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b; /* [768,768], [768] */
  const float *ln1_w, *ln1_b;                                 /* [768] */
  const float *ff1_w, *ff1_b;                                 /* [3072,768], [3072] */
  const float *ff2_w, *ff2_b;                                 /* [768,3072], [768] */
  const float *ln2_w, *ln2_b;                                 /* [768] */
} vtc_layer_weights;

typedef struct {
  const float* conv_w[7];  /* [512,1,10], [512,512,3] x4, [512,512,2] x2 (no bias) */
  const float *gn_w, *gn_b;       /* GroupNorm(512 groups) affine, [512] */
  const float *fp_ln_w, *fp_ln_b; /* feature-projection LayerNorm, [512] */
  const float *fp_w, *fp_b;       /* projection [768,512], [768] */
  const float* pos_w;             /* positional conv weight with weight-norm already folded, [768,48,128] */
  const float* pos_b;             /* [768] */
  const float *enc_ln_w, *enc_ln_b; /* [768] */
  vtc_layer_weights layer[VTC_N_LAYERS];
  const float *head_w, *head_b;   /* per-frame multi-label head [n_labels,768], [n_labels] */
} vtc_weights;

/* Stands behind: Models[name].load_from_checkpoint(...).eval().to(device)  (vtc.py:185-192).
 * Packs the weights for the MFMA kernels, uploads them and allocates the activation workspace. */
int vtc_create(const vtc_spec* spec, const vtc_weights* w, int device, vtc_model** out);
void vtc_destroy(vtc_model* m);
/* Message of the last failure on `m` (m == NULL: of the last failed handle-less call, vtc_create or vtc_flac_*, on
 * the calling thread). */
const char* vtc_last_error(const vtc_model* m);

/* Stands behind: model.conv_settings (vtc.py:194) -- frames per window T, and the frame count of a file. */
int vtc_frames_per_window(const vtc_model* m);
int64_t vtc_frames_for_samples(int64_t n_samples);

/* Stands behind: the batched model.forward inside segma.inference.apply_model_on_audio.
 * Runs n_windows (<= max_batch) windows.  Window i reads window_samples samples from
 * wav[start[i] ...], of which only the first valid[i] are real (the rest count as zeros), and writes
 * the logits of its first keep[i] frames to logits[(dst_row[i] + t) * n_labels + l].
 * Asynchronous on `stream`; the four index arrays are copied before the call returns. */
int vtc_forward_windows(vtc_model* m, const float* wav /*[dev]*/, const int64_t* start /*[host]*/,
                        const int32_t* valid /*[host]*/, const int64_t* dst_row /*[host]*/,
                        const int32_t* keep /*[host]*/, int32_t n_windows, float* logits /*[dev]*/, void* stream);

/* Stands behind: segma.inference.apply_model_on_audio after decoding (vtc.py:236-243): tiles one
 * waveform of n_samples into windows stepped by T*320 samples (chunk_artifact.py:71-73), runs them in
 * batches of `batch` and leaves the stitched logits [n_frames][n_labels] in `logits`
 * (capacity vtc_frames_for_samples(n_samples) rows).  Asynchronous on `stream`. */
int vtc_infer_waveform(vtc_model* m, const float* wav /*[dev]*/, int64_t n_samples, int32_t batch,
                       float* logits /*[dev]*/, int64_t* n_frames /*[host]*/, void* stream);

/* Stands behind: segma.inference.apply_thresholds + create_intervals (vtc.py:84-91) and the sigmoid
 * summary (vtc.py:266-273).  Frame f of label l is active iff lo_logit[l] <= logits[f][l] <= hi_logit[l].
 * Every run of active frames i..j yields out_start = i*320, out_end = j*320+400 (samples), out_label = l;
 * output is label-major, time-ordered.  Synchronises `stream` before returning.
 * On VTC_ERR_CAPACITY *n_out holds the number of intervals needed.
 * A NaN compares false against every bound, so non-finite logits would decode as "inactive" and the recording would
 * look valid: if any logit of the rows is NaN or +-inf the call fails with VTC_ERR_NONFINITE instead (the host writes an
 * error row, src/pipeline/vtc.py:245-249).  The f16 kernels store activations in fp16; values beyond 65504 become inf
 * there and reach the logits as NaN. */
int vtc_decode(vtc_model* m, const float* logits /*[dev]*/, int64_t n_frames, const float* lo_logit /*[host]*/,
               const float* hi_logit /*[host]*/, int64_t* out_start /*[host]*/, int64_t* out_end /*[host]*/,
               int32_t* out_label /*[host]*/, int64_t cap, int64_t* n_out, float* max_logit, double* mean_sigmoid,
               void* stream);

/* Stands behind: segma.inference.apply_thresholds alone (vtc.py:84-90), for callers that look at the 0/1 mask itself:
 * mask[f * n_labels + l] = (lo_logit[l] <= logits[f][l] <= hi_logit[l]) ? 1 : 0.  Asynchronous on `stream`. */
int vtc_threshold_mask(vtc_model* m, const float* logits /*[dev]*/, int64_t n_frames, const float* lo_logit /*[host]*/,
                       const float* hi_logit /*[host]*/, uint8_t* mask /*[dev] n_frames x n_labels*/, void* stream);

/* vtc_decode for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference (its decode is a Python loop on the host, vtc.py:84-91); same arithmetic and
 * the same kernels as vtc_decode, but
 *   - the scratch the kernels need (per-block run counts, their scan, the per-label run lists, the sigmoid partials)
 *     lives in a file-scope device array of the library, VTC_STATIC_DECODE_BYTES per GPU, held until the process ends;
 *   - every output is DEVICE memory of the caller, so nothing is copied to the host and nothing synchronises: the call only
 *     enqueues four kernels on `stream` and performs no hipMalloc / hipFree / host allocation.
 * Limits: n_frames <= VTC_STATIC_DECODE_MAX_FRAMES (larger inputs are rejected with VTC_ERR_CAPACITY; use vtc_decode).
 * result[0] receives the number of intervals found (if it exceeds `cap` only the first `cap` were written; -1 if the run
 * lists were inconsistent: out_start / out_end / out_label are then left untouched, whatever they held before the call), result_summary = {max logit (NaN if a logit is non-finite, as vtc_decode_files), mean sigmoid}.
 * All calls on one GPU share that GPU's array, so calls must not overlap in time on one GPU.  Calls on one stream are ordered
 * by the stream.  A call on another stream (of any model) while the previous call's kernels are still pending is refused with
 * VTC_ERR_ARG instead of racing: the library records an event behind each call (created in vtc_create, not here) and queries
 * it.  Under stream capture the check is skipped (events may not be queried there): whoever replays such graphs on several
 * streams orders them.  Nothing has to be cleared between calls (every word a call reads it has written itself). */
#define VTC_STATIC_DECODE_MAX_FRAMES 262144   /* 87 min of audio at 20 ms per frame */
#define VTC_STATIC_DECODE_BYTES ((8u << 20) + (1u << 17))
int vtc_decode_static(vtc_model* m, const float* logits /*[dev]*/, int64_t n_frames, const float* lo_logit /*[host]*/,
                      const float* hi_logit /*[host]*/, int64_t* out_start /*[dev]*/, int64_t* out_end /*[dev]*/,
                      int32_t* out_label /*[dev]*/, int64_t cap, int64_t* result /*[dev] 1*/,
                      double* result_summary /*[dev] 2*/, void* stream);

/* Several recordings in one call (second caller of the boundary: src/analysis/vtc_on_clips.py:154-211 runs
 * apply_model_on_audio once per packaged clip, most of them shorter than a batch of windows).  Recording f is
 * wav[offset[f] .. offset[f] + n_samples[f]) of one device arena; windows are planned per recording exactly as in
 * vtc_infer_waveform and windows of different recordings share forward passes.  Stitched logits of recording f land
 * in rows [file_rows[f], file_rows[f+1]) (file_rows has n_files + 1 entries, written by the call; a recording
 * shorter than the receptive field gets zero rows).  Row results are bit-identical to per-recording calls. */
int vtc_infer_waveforms(vtc_model* m, const float* wav /*[dev]*/, const int64_t* offset /*[host]*/,
                        const int64_t* n_samples /*[host]*/, int32_t n_files, int32_t batch, float* logits /*[dev]*/,
                        int64_t* file_rows /*[host] n_files+1, out*/, void* stream);
/* vtc_decode over such a row layout: runs never cross a recording boundary, out_start / out_end are samples relative
 * to the recording, output is recording-major and inside a recording label-major, time-ordered (= vtc_decode on each
 * recording's rows).  max_logit / mean_sigmoid (each [n_files], may be null) receive the per-recording sigmoid
 * summary of src/pipeline/vtc.py:266-273.  Synchronises `stream`.
 * Non-finite logits: max_logit[f] is NaN for exactly the recordings that hold a NaN / inf logit (their runs are not to be
 * trusted; the other recordings are unaffected); without max_logit the call fails with VTC_ERR_NONFINITE if any row is
 * affected.  vtc_decode_collect follows the same rule (want_summary set: NaN marks; not set: the error). */
int vtc_decode_files(vtc_model* m, const float* logits /*[dev]*/, const int64_t* file_rows /*[host] n_files+1*/,
                     int32_t n_files, const float* lo_logit /*[host]*/, const float* hi_logit /*[host]*/,
                     int32_t* out_file /*[host]*/, int64_t* out_start /*[host]*/, int64_t* out_end /*[host]*/,
                     int32_t* out_label /*[host]*/, int64_t cap, int64_t* n_out, float* max_logit /*[host]*/,
                     double* mean_sigmoid /*[host]*/, void* stream);

/* Two-step form of vtc_decode / vtc_decode_files for callers that keep the GPU fed (the pipeline's GPU thread; the
 * reference runs forward, .cpu() and the Python decode strictly one after the other, vtc.py:236-268).  _launch enqueues
 * the decode kernels and the copies of their results into pinned memory on `stream` and returns without waiting; the
 * caller may then enqueue more work on that stream (the next forward); _collect waits only for what was enqueued up to
 * _launch and hands out the runs exactly as vtc_decode_files does (n_files = 1, file_rows = {0, n_frames}: as
 * vtc_decode, out_file all 0).  max_logit / mean_sigmoid ([n_files], may be null) are filled if want_summary was set.
 * One decode may be in flight per model.  On VTC_ERR_CAPACITY the results stay parked and *n_out tells the room needed. */
int vtc_decode_launch(vtc_model* m, const float* logits /*[dev]*/, const int64_t* file_rows /*[host] n_files+1*/,
                      int32_t n_files, const float* lo_logit /*[host]*/, const float* hi_logit /*[host]*/,
                      int32_t want_summary, void* stream);
int vtc_decode_collect(vtc_model* m, int32_t* out_file /*[host]*/, int64_t* out_start /*[host]*/, int64_t* out_end /*[host]*/,
                       int32_t* out_label /*[host]*/, int64_t cap, int64_t* n_out, float* max_logit /*[host]*/,
                       double* mean_sigmoid /*[host]*/);

/* Audio front-end of apply_model_on_audio (src/pipeline/vtc.py:236-243 decodes the file to mono 16 kHz float inside
 * segma before the model runs): interleaved PCM already in device memory -> mono float32 at 16 kHz, on `stream`.
 *   format 0 = int16 (scaled by 1/32768), 1 = float32, 2 = packed little-endian int24 (3 bytes per sample, scaled by 1/2^23:
 *   24-bit PCM WAVs cross PCIe as stored); channels are averaged; sr_in != 16000 is resampled with the
 *   windowed-sinc polyphase filter of torchaudio.functional.resample (sinc_interp_hann, width 6, rolloff 0.99), the
 *   library the reference pulls in through segma (requirements.txt:393-394).  The rule itself is unpinned.
 * *n_out = vtc_resampled_length(n_in, sr_in) = ceil(n_in * 16000 / sr_in); VTC_ERR_CAPACITY if out_cap is smaller.
 * The per-rate filter table is built and kept in the handle on first use.  Unlike the other entry points this one may
 * be called from several threads at once on one handle (the pipeline's reader threads do), each on its own stream, as
 * long as vtc_profile_enable is off (the launch records are not locked). */
int64_t vtc_resampled_length(int64_t n_in, int32_t sr_in);
int vtc_prepare_audio(vtc_model* m, const void* pcm /*[dev] n_in x n_channels, interleaved*/, int32_t format, int64_t n_in,
                      int32_t n_channels, int32_t sr_in, float* out /*[dev]*/, int64_t out_cap, int64_t* n_out, void* stream);

/* FLAC members of the audio front-end (host side, no GPU work, no handle): the reference's packaged clips are FLAC by
 * default (src/packaging/writer.py:36,119; src/analysis/vtc_on_clips.py:163-166 prefers the .flac member) and manifests
 * may list .flac recordings (scripts/make_manifest.py:26-39); the reference decodes them with torchcodec/ffmpeg inside
 * segma.  Decoder of RFC 9639 streams held in memory (an ID3v2 tag in front is skipped): frame header CRC-8 and frame
 * CRC-16 are checked; the decoded integers are exactly the encoder's input (lossless), STREAMINFO's MD5 is returned so
 * the caller can verify them.  vtc_flac_info: n_samples = inter-channel samples, 0 if the stream does not say.
 * vtc_flac_decode writes interleaved samples: out_format 0 = int16 as stored (streams of <= 16 bits per sample),
 * 1 = float32 = integer / 2^(bits-1), 2 = int32 as stored; out == NULL only counts.  VTC_ERR_CAPACITY if out_cap
 * (inter-channel samples) is too small.  Thread-safe; the message of a failure is vtc_last_error(NULL) on the
 * calling thread. */
int vtc_flac_info(const void* data /*[host]*/, int64_t n_bytes, int32_t* sample_rate, int32_t* n_channels,
                  int32_t* bits_per_sample, int64_t* n_samples, uint8_t* md5 /*[16] or NULL*/);
int vtc_flac_decode(const void* data /*[host]*/, int64_t n_bytes, void* out /*[host]*/, int32_t out_format,
                    int64_t out_cap, int64_t* n_out);
/* The same with up to n_threads threads for long streams (day-long recordings: a 16 h file is 8-16 s of work for one
 * thread): the byte range is cut into pieces of >= 1 MiB, every thread locks onto the first frame that starts in its piece
 * (header CRC-8 and frame CRC-16 must both check) and places its frames by their coded frame / sample number; pieces
 * that do not join seamlessly, streams without a declared length and n_threads <= 1 fall back to the sequential decoder,
 * so the result is always that of vtc_flac_decode. */
int vtc_flac_decode_threads(const void* data /*[host]*/, int64_t n_bytes, void* out /*[host]*/, int32_t out_format,
                            int64_t out_cap, int64_t* n_out, int32_t n_threads);

/* 4-bit IMA / DVI ADPCM blocks of a RIFF/WAVE file (format tag 0x0011; host side, no GPU work, no handle).  `.wav` is on the
 * reference's manifest list (scripts/make_manifest.py:26-39) and the reference reads whatever torchcodec/ffmpeg accept inside
 * segma (apply_model_on_audio, src/pipeline/vtc.py:236-243).  data = the 'data' chunk, block_align / samples_per_block = the
 * 'fmt ' chunk's nBlockAlign / wSamplesPerBlock (0 = what a block holds); writes interleaved int16, at most out_cap
 * inter-channel samples (the caller passes the 'fact' chunk's count, or blocks x samples_per_block), *n_out = those written. */
int vtc_ima_adpcm_decode(const void* data /*[host]*/, int64_t n_bytes, int32_t n_channels, int32_t block_align,
                         int32_t samples_per_block, int16_t* out /*[host]*/, int64_t out_cap, int64_t* n_out);

/* Test hook: copy an intermediate activation of the LAST vtc_forward_windows call to the host as fp32.
 * stage: 0..6 conv layer outputs (B,L_i,512); 7 projection (B*T,768); 8 encoder input after pos-conv+LN;
 * 9+l output of transformer layer l.  Returns the element count, or a negative status. */
int64_t vtc_debug_stage(vtc_model* m, int32_t stage, int32_t n_windows, float* out /*[host]*/, int64_t cap);
/* Test hook: make the next vtc_forward_windows calls return right after `stage` (-1 = run everything),
 * so that vtc_debug_stage reads that stage before later stages reuse its buffer. */
int vtc_debug_stop_after(vtc_model* m, int32_t stage);

/* Per-kernel-class timing with HIP events on the launch stream (bench.py roofline object).
 * vtc_profile_enable(m,1) starts (or resumes) recording every launch, (m,0) pauses it and keeps what was
 * recorded, so a caller can sample some steps of a timed region; vtc_profile_read synchronises, sums the
 * recorded launches per class and resets.  Two event records per launch cost about 7 us of stream time.
 * Classes: see vtc_profile_class_name. */
#define VTC_PROF_CLASSES 8
int vtc_profile_enable(vtc_model* m, int32_t on);
int vtc_profile_read(vtc_model* m, double* ms /*[VTC_PROF_CLASSES]*/, int64_t* launches /*[VTC_PROF_CLASSES]*/,
                     double* flops /*[VTC_PROF_CLASSES]*/, double* bytes /*[VTC_PROF_CLASSES]*/);
const char* vtc_profile_class_name(int32_t cls);

#ifdef __cplusplus
}
#endif
#endif /* VTC_AMD_H */
