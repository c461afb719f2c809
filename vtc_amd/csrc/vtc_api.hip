// libvtc_amd.so -- C ABI (include/vtc_amd.h) over the hand-written gfx950 kernels.
// Host side: weight packing, workspace, launch sequence of one BabyHuBERT forward, decode.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/vtc_amd.h"
#include "attention.hip.h"
#include "audio_kernels.hip.h"
#include "common.hip.h"
#include "decode_kernels.hip.h"
#include "encoder_kernels.hip.h"
#include "adpcm_decode.h"
#include "flac_decode.h"
#include "gemm.hip.h"
#include "gemm256.hip.h"
#include "posconv.hip.h"

using namespace vtc;

namespace {

constexpr int kKernels[7] = {10, 3, 3, 3, 3, 2, 2};  // src/analysis/chunk_artifact.py:43
constexpr int kStrides[7] = {5, 2, 2, 2, 2, 2, 2};   // src/analysis/chunk_artifact.py:44
constexpr int kSlots = 8;

enum ProfClass { PC_CONV0_STATS = 0, PC_CONV0_APPLY, PC_GEMM, PC_ATTN, PC_LAYERNORM, PC_PACK, PC_DECODE, PC_OTHER };
const char* kProfNames[VTC_PROF_CLASSES] = {"conv0_stats", "conv0_apply", "gemm_mfma", "attention",
                                            "layernorm_head", "posconv_pack", "decode", "other"};

thread_local std::string g_create_error;

// ---- scratch of the no-allocation decode (include/vtc_amd.h: vtc_decode_static).  One copy per GPU, at its own address:
// the name below is a host-side stand-in, never a device pointer; vtc_create resolves the address on its device with
// hipGetSymbolAddress and everything else uses that pointer.
constexpr long long kStMaxFrames = VTC_STATIC_DECODE_MAX_FRAMES;
constexpr long long kStNblk = kStMaxFrames / 256;
constexpr long long kStRunCap = kStMaxFrames / 2 + 1;
constexpr size_t kStOffCounts = 0;                                                        // int [2][8][nblk]
constexpr size_t kStOffTotals = kStOffCounts + sizeof(int) * 2 * VTC_MAX_LABELS * kStNblk;  // int [16]
constexpr size_t kStOffSummary = kStOffTotals + 256;                                      // double [2]
constexpr size_t kStOffSig = kStOffSummary + 256;                                         // double [nblk]
constexpr size_t kStOffMax = kStOffSig + sizeof(double) * kStNblk;                        // float [nblk]
constexpr size_t kStOffStart = (kStOffMax + sizeof(float) * kStNblk + 255) & ~(size_t)255;  // int [8][run cap]
constexpr size_t kStOffEnd = (kStOffStart + sizeof(int) * VTC_MAX_LABELS * kStRunCap + 255) & ~(size_t)255;
constexpr size_t kStBytes = kStOffEnd + sizeof(int) * VTC_MAX_LABELS * kStRunCap;
static_assert(kStBytes <= VTC_STATIC_DECODE_BYTES, "scratch layout exceeds the size the header states");
static __device__ __align__(256) unsigned char scratch[VTC_STATIC_DECODE_BYTES];
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};   // device address of `scratch` on each GPU
std::mutex g_scratch_mu;
// Overlap guard of vtc_decode_static: an event recorded behind each call (created with the scratch address, once per GPU).
// A call on ANOTHER stream while that event is still pending would share the array with work in flight: it is refused
// (VTC_ERR_ARG) instead of racing.  Calls on the same stream are ordered by the stream; calls under stream capture are not
// checked (an event may not be queried there; the graph's owner orders its replays).
hipEvent_t g_static_done[kMaxDevices] = {nullptr};
hipStream_t g_static_stream[kMaxDevices] = {nullptr};
bool g_static_pending[kMaxDevices] = {false};

// Device weights of one transformer layer in the LayerNorm-free form (gemm256.hip.h): the LayerNorm in front of a GEMM is
// folded into it (gamma into the 16-bit weights, beta into the bias, colsum for the mean), the LayerNorm behind a residual
// add is applied by the epilogue that adds the residual (its gamma as res_g, its beta folded into that GEMM's bias).
struct LayerDev {
  h16 *qkv_w, *o_w, *ff1_w, *ff2_w;
  float *qkv_b, *qkv_s, *o_bb, *ff1_b, *ff1_s, *ff2_bb;
  float *g_in, *ln1_w, *ln1_b, *ln2_w, *ln2_b;   // g_in: gamma of the LayerNorm that feeds the layer (previous ln2 / encoder LN)
};

struct ProfRec { int cls; double flops, bytes; hipEvent_t e0, e1; };

}  // namespace

struct vtc_model {
  vtc_spec spec{};
  int device = 0;
  bool f16 = false;
  int T = 0;
  int L[7] = {0};
  std::string err;
  std::vector<void*> allocs;

  // weights
  float *conv0_w = nullptr, *gn_w = nullptr, *gn_b = nullptr, *fp_ln_w = nullptr, *fp_ln_b = nullptr, *fp_b = nullptr;
  float *pos_b = nullptr, *enc_ln_w = nullptr, *enc_ln_b = nullptr, *head_w = nullptr, *head_b = nullptr;
  h16* conv_w[7] = {nullptr};
  h16 *fp_w = nullptr, *pos_w = nullptr;
  LayerDev layer[VTC_N_LAYERS]{};

  // workspace
  h16 *actA = nullptr, *actB = nullptr, *x16 = nullptr, *qkv16 = nullptr, *ctx16 = nullptr, *ffn16 = nullptr, *posin = nullptr;
  float *ln_stats = nullptr, *ln_part = nullptr;   // per-row (mean, rstd); the 16 partial (mean, M2) pairs per row behind them
  float *h0 = nullptr, *tmp32 = nullptr, *gn_scale = nullptr, *gn_shift = nullptr, *c0_xscale = nullptr;
  _Float16* c0_aop = nullptr;   // conv0 MFMA operand (B, 512, 32), written per batch by the GroupNorm finalize kernel
  int* c0_flag = nullptr;
  double* mom_partial = nullptr;
  int nchunk = 0;

  // window table: device copy + pinned staging ring
  long long *d_start = nullptr, *d_dst = nullptr;
  int *d_valid = nullptr, *d_keep = nullptr;
  char* h_stage = nullptr;
  hipEvent_t slot_ev[kSlots]{};
  bool slot_used[kSlots] = {false};
  int slot = 0;

  // decode scratch (grown on demand)
  long long dec_frames_cap = 0, dec_run_cap = 0;   // frames the scratch holds; runs per label it holds
  int *dec_counts = nullptr, *dec_totals = nullptr, *dec_start = nullptr, *dec_end = nullptr;
  unsigned char* dec_fstart = nullptr;
  long long* fsum_rows = nullptr;   // vtc_decode_files per-recording summary scratch
  double* fsum_out = nullptr;
  int fsum_cap = 0;
  double *dec_sig = nullptr, *dec_summary = nullptr;
  float* dec_max = nullptr;
  // vtc_decode_launch / vtc_decode_collect: one decode in flight, results parked in pinned host memory
  char* dec_host = nullptr;
  size_t dec_host_cap = 0;
  hipEvent_t dec_event = nullptr;
  bool dec_active = false, dec_want_summary = false;
  long long dec_n = 0, dec_rcap = 0;
  std::vector<int64_t> dec_rows;

  int debug_stop = -1;
  bool big_tiles = true;   // 256x256 8-phase GEMM (VTC_GEMM_TILE=128 selects the 128x128 kernel, for A/B runs)

  struct ResampleDev { ResampleTable t; };
  std::map<int, ResampleDev> resample;   // per input sample rate, built on first use (vtc_prepare_audio)
  std::mutex resample_mu;                // vtc_prepare_audio is called from several reader threads
  bool prof_on = false;
  bool conv0_mfma = true;  // conv0 as one fp16-split MFMA per 16 x 16 outputs (env VTC_CONV0_MFMA=0: VALU form)
  std::vector<ProfRec> prof;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_next = 0;
};

namespace {

int fail(vtc_model* m, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) {
    static std::mutex mu;   // vtc_prepare_audio may fail on several reader threads at once
    std::lock_guard<std::mutex> lock(mu);
    m->err = buf;
  } else {
    g_create_error = buf;
  }
  return code;
}

#define HIP_TRY(m, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail((m), VTC_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

unsigned short host_to_h16(float x, bool f16) {
  if (f16) {
    _Float16 h = (_Float16)x;  // round-to-nearest-even
    unsigned short r;
    memcpy(&r, &h, 2);
    return r;
  }
  unsigned u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

float host_from_h16(unsigned short v, bool f16) {
  if (f16) { _Float16 h; memcpy(&h, &v, 2); return (float)h; }
  const unsigned u = (unsigned)v << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}

// A linear layer behind a LayerNorm, folded: W'[n][k] = W[n][k] * gamma[k] (* scale for the first scale_rows rows) rounded to
// 16 bit; colsum[n] = sum_k W'[n][k] of the ROUNDED weights (it multiplies the row mean, which must cancel against what the
// MFMA accumulated with those weights); bias'[n] = (b[n] + sum_k beta[k] W[n][k]) * scale, exact weights, fp64 sums.
struct FoldedLinear { std::vector<unsigned short> w; std::vector<float> colsum, bias; };
FoldedLinear fold_layernorm(const float* w, const float* b, int N, int K, const float* gamma, const float* beta, bool f16,
                            float scale = 1.f, int scale_rows = 0) {
  FoldedLinear out;
  out.w.resize((size_t)N * K); out.colsum.resize(N); out.bias.resize(N);
  for (int n = 0; n < N; ++n) {
    const float sc = n < scale_rows ? scale : 1.f;
    double cs = 0.0, bb = b[n];
    for (int k = 0; k < K; ++k) {
      const unsigned short h = host_to_h16(w[(size_t)n * K + k] * gamma[k] * sc, f16);
      out.w[(size_t)n * K + k] = h;
      cs += (double)host_from_h16(h, f16);
      bb += (double)beta[k] * (double)w[(size_t)n * K + k];
    }
    out.colsum[n] = (float)cs;
    out.bias[n] = (float)(bb * sc);
  }
  return out;
}

template <typename T>
int dev_alloc(vtc_model* m, T** p, size_t n) {
  void* q = nullptr;
  HIP_TRY(m, hipMalloc(&q, n * sizeof(T) + 256));
  m->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

int upload_f32(vtc_model* m, float** dst, const float* src, size_t n) {
  if (!src) return fail(m, VTC_ERR_ARG, "missing weight tensor");
  if (int rc = dev_alloc(m, dst, n)) return rc;
  HIP_TRY(m, hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int upload_h16(vtc_model* m, h16** dst, const std::vector<unsigned short>& v) {
  if (int rc = dev_alloc(m, dst, v.size())) return rc;
  HIP_TRY(m, hipMemcpy(*dst, v.data(), v.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// plain [N][K] fp32 -> 16 bit, optional row scale for rows [0, scale_rows)
std::vector<unsigned short> pack_linear(const float* w, int N, int K, bool f16, float scale = 1.f, int scale_rows = 0) {
  std::vector<unsigned short> v((size_t)N * K);
  for (int n = 0; n < N; ++n) {
    const float s = n < scale_rows ? scale : 1.f;
    for (int k = 0; k < K; ++k) v[(size_t)n * K + k] = host_to_h16(w[(size_t)n * K + k] * s, f16);
  }
  return v;
}

// conv weight [N][Cin][k] -> [N][j*Cin + ci] (the im2col order of channel-last activations)
std::vector<unsigned short> pack_conv(const float* w, int N, int Cin, int k, bool f16) {
  std::vector<unsigned short> v((size_t)N * Cin * k);
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int j = 0; j < k; ++j)
        v[((size_t)n * k + j) * Cin + ci] = host_to_h16(w[((size_t)n * Cin + ci) * k + j], f16);
  return v;
}

// grouped positional conv [768][48][128] -> [16 groups][64 rows (48 real + 16 zero)][j*48 + ci]
std::vector<unsigned short> pack_posconv(const float* w, bool f16) {
  std::vector<unsigned short> v((size_t)16 * 64 * 6144, 0);
  for (int g = 0; g < 16; ++g)
    for (int co = 0; co < 48; ++co)
      for (int ci = 0; ci < 48; ++ci)
        for (int j = 0; j < 128; ++j)
          v[((size_t)(g * 64 + co)) * 6144 + j * 48 + ci] = host_to_h16(w[((size_t)(g * 48 + co) * 48 + ci) * 128 + j], f16);
  return v;
}

// ------------------------------------------------------------------ profiling
void prof_begin(vtc_model* m, int cls, double flops, double bytes, hipStream_t s) {
  if (!m->prof_on) return;
  if (m->ev_next == m->ev_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { m->prof_on = false; return; }
    m->ev_pool.push_back({a, b});
  }
  auto& ev = m->ev_pool[m->ev_next++];
  m->prof.push_back({cls, flops, bytes, ev.first, ev.second});
  (void)hipEventRecord(ev.first, s);
}
void prof_end(vtc_model* m, hipStream_t s) {
  if (!m->prof_on || m->prof.empty()) return;
  (void)hipEventRecord(m->prof.back().e1, s);
}
#define LAUNCH(m, cls, flops, bytes, s, ...) \
  do { prof_begin((m), (cls), (flops), (bytes), (s)); __VA_ARGS__; prof_end((m), (s)); } while (0)

// ------------------------------------------------------------------ GEMM dispatch
template <bool F16, int BN, int EPI>
void gemm_go(vtc_model* m, GemmParams p, int nz, hipStream_t s) {
  const double flops = 2.0 * p.M * (double)p.N * p.K * nz;
  // Tile choice (measured with tools/gemm_bench): the persistent 256 x BN kernel needs N % BN == 0 and
  // K % 128 == 0 (>= 256); BN = 192 where N allows and its rounds x width are no worse (N = 768, 2304, 3072:
  // the 192-wide kernel has no register spills and a cheaper epilogue), BN = 256 otherwise (conv, N = 512).
  // The persistent kernel's epilogue addresses C (and the residual) through 32-bit buffer offsets: an output larger than
  // 2 GiB (large batches of short windows: conv1 at batch 512 x 4 s is 3.3 GB) is cut into row chunks of whole
  // windows, each its own launch over the same weights.
  constexpr long long kEsz = (EPI & EPI_F32) ? 4 : 2;
  if constexpr ((EPI & (EPI_LNIN | EPI_LNRES)) != 0) {
    // LayerNorm-free layer GEMMs exist in the persistent kernel only (any M; N % 192 == 0 and K % 128 == 0 hold for
    // 768 / 2304 / 3072).  Outputs beyond 2 GiB are cut into row chunks; every per-row array moves with the chunk.
    long long rows_max = (((1ll << 31) - 1) / ((long long)p.ldc * kEsz)) / 256 * 256;
    if (const char* e = getenv("VTC_GEMM_ROWS_MAX")) {   // test hook: force the row-chunk loop on small inputs (multiples of 256)
      const long long v = atoll(e) / 256 * 256;
      if (v >= 256 && v < rows_max) rows_max = v;
    }
    for (long long r0 = 0; r0 < p.M; r0 += rows_max) {
      GemmParams q = p;
      q.M = (int)std::min<long long>(rows_max, p.M - r0);
      q.rows_per_batch = q.M;
      q.A = p.A + r0 * p.a_row_stride;
      q.C = (char*)p.C + r0 * p.ldc * kEsz;
      q.ln_stats = p.ln_stats + r0 * 2;
      if (p.res_h) q.res_h = p.res_h + r0 * p.ldc;
      if (p.part) q.part = p.part + r0 * 2;   // group-major: the group stride (part_rows) stays that of the whole matrix
      const double fl = 2.0 * q.M * (double)p.N * p.K;
      // 192-wide tiles unless 256-wide ones need no more rounds x width over the 256 CUs: FFN1 (N = 3072: 6 rounds x 256
      // against 8 x 192) then runs its K loop with 16 MFMAs in every phase (81 % against 69 % MFMA utilisation in the loop,
      // 9 % per launch; 2 registers spilled in the epilogue); QKV (N = 2304: 5 x 256 against 6 x 192) stays 192-wide.
      bool wide = false;
      if constexpr (!(EPI & EPI_LNRES)) {
        const long long tm = (q.M + 255) / 256;
        const long long r192 = (tm * (p.N / 192) + 255) / 256, r256 = (tm * (p.N / 256) + 255) / 256;
        wide = p.N % 256 == 0 && r256 * 256 <= r192 * 192;
        static const int force_bn = getenv("VTC_GEMM_BN") ? atoi(getenv("VTC_GEMM_BN")) : 0;
        if (force_bn == 192) wide = false;
        if (wide) LAUNCH(m, PC_GEMM, fl, 0.0, s, (launch_gemm256<F16, EPI, 256>(q, 1, s)));
      }
      if (!wide) LAUNCH(m, PC_GEMM, fl, 0.0, s, (launch_gemm256<F16, EPI, 192>(q, 1, s)));
    }
    return;
  } else {
  if (BN == 128 && m->big_tiles && p.K % 128 == 0 && p.K >= 256 && p.M >= 4096 && nz == 1 &&
      (long long)p.M * p.ldc * kEsz >= (1ll << 31)) {
    const long long unit = p.rows_per_batch < p.M ? p.rows_per_batch : 256;       // rows that may start a chunk
    long long rows_max = ((1ll << 31) - 1) / ((long long)p.ldc * kEsz);
    rows_max = rows_max / unit * unit;
    if (rows_max >= 4096) {
      for (long long r0 = 0; r0 < p.M; r0 += rows_max) {
        GemmParams q = p;
        q.M = (int)std::min<long long>(rows_max, p.M - r0);
        if (p.rows_per_batch < p.M) q.A = p.A + (r0 / p.rows_per_batch) * p.a_batch_stride;
        else { q.A = p.A + r0 * p.a_row_stride; q.rows_per_batch = q.M; }
        q.C = (char*)p.C + r0 * p.ldc * kEsz;
        if (p.resid) q.resid = (const float*)((const char*)p.resid + r0 * p.ldc * ((EPI & EPI_RESID16) ? 2 : 4));
        gemm_go<F16, BN, EPI>(m, q, nz, s);
      }
      return;
    }
  }
  const bool fits32 = (long long)p.M * p.ldc * kEsz < (1ll << 31);
  if (BN == 128 && m->big_tiles && p.K % 128 == 0 && p.K >= 256 && p.M >= 4096 && fits32) {
    // rounds of tiles over the 256 CUs x tile width ~ time
    const long long tm = (p.M + 255) / 256;
    const long long c256 = p.N % 256 == 0 ? ((tm * (p.N / 256) * nz + 255) / 256) * 256 : -1;
    const long long c192 = p.N % 192 == 0 ? ((tm * (p.N / 192) * nz + 255) / 256) * 192 : -1;
    static const int force_bn = getenv("VTC_GEMM_BN") ? atoi(getenv("VTC_GEMM_BN")) : 0;   // A/B runs: 192 or 256 where N allows
    if (force_bn == 256 && c256 > 0) { LAUNCH(m, PC_GEMM, flops, 0.0, s, (launch_gemm256<F16, EPI, 256>(p, nz, s))); return; }
    if (force_bn == 192 && c192 > 0) { LAUNCH(m, PC_GEMM, flops, 0.0, s, (launch_gemm256<F16, EPI, 192>(p, nz, s))); return; }
    if (c256 > 0 && (c192 < 0 || 100 * c192 > 102 * c256)) { LAUNCH(m, PC_GEMM, flops, 0.0, s, (launch_gemm256<F16, EPI, 256>(p, nz, s))); return; }
    if (c192 > 0) { LAUNCH(m, PC_GEMM, flops, 0.0, s, (launch_gemm256<F16, EPI, 192>(p, nz, s))); return; }
  }
  LAUNCH(m, PC_GEMM, flops, 0.0, s, (launch_gemm<F16, BN, EPI>(p, nz, s)));
  }
}

GemmParams plain(const h16* A, const h16* W, void* C, const float* bias, const float* resid, int M, int N, int K) {
  GemmParams p{};
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.resid = resid;
  p.M = M; p.N = N; p.K = K; p.ldc = N;
  p.rows_per_batch = M; p.a_row_stride = K; p.a_batch_stride = 0;
  return p;
}

template <bool F16>
int forward_impl(vtc_model* m, const float* wav, int nw, float* logits, hipStream_t s) {
  const int T = m->T;
  const int* L = m->L;
  WindowTable wt{m->d_start, m->d_valid, m->d_dst, m->d_keep};

  // ---- conv0: GroupNorm statistics from window moments, then conv + affine + GELU
  LAUNCH(m, PC_CONV0_STATS, 0.0, (double)nw * m->spec.window_samples * 4, s, {
    hipLaunchKernelGGL(conv0_moments_kernel, dim3(m->nchunk, nw), dim3(256), 0, s, wav, wt, L[0], m->mom_partial);
    hipLaunchKernelGGL(conv0_gn_finalize_kernel, dim3(nw), dim3(512), 0, s, m->mom_partial, m->nchunk, L[0], m->conv0_w,
                       m->gn_w, m->gn_b, m->gn_scale, m->gn_shift, m->c0_aop, m->c0_xscale, m->c0_flag);
  });
  const double c0_bytes = (double)nw * (m->spec.window_samples * 4.0 + (double)L[0] * 512 * 2);
  if (F16 && m->conv0_mfma) {   // fp16-split MFMA form; bf16 operands keep the VALU form
    LAUNCH(m, PC_CONV0_APPLY, 0.0, c0_bytes, s,
           {
             hipLaunchKernelGGL(conv0_mfma_kernel, dim3((L[0] + kC0mFrames - 1) / kC0mFrames, nw), dim3(256), 0, s, wav, wt, L[0],
                                m->c0_aop, m->c0_xscale, m->c0_flag, m->actA);
             hipLaunchKernelGGL(conv0_valu_fallback_kernel, dim3(8, nw), dim3(256), 0, s, wav, wt, L[0], m->c0_flag, m->conv0_w,
                                m->gn_scale, m->gn_shift, m->actA);
           });
  } else {
    LAUNCH(m, PC_CONV0_APPLY, 0.0, c0_bytes, s,
           hipLaunchKernelGGL((conv0_apply_kernel<F16>), dim3((L[0] + kC0Frames - 1) / kC0Frames, nw), dim3(256), 0, s, wav,
                              wt, L[0], m->conv0_w, m->gn_scale, m->gn_shift, m->actA));
  }
  if (m->debug_stop == 0) return 0;

  // ---- conv 1..6 as implicit GEMM over channel-last activations, GELU fused
  h16* src = m->actA;
  h16* dst = m->actB;
  for (int i = 1; i < 7; ++i) {
    GemmParams p{};
    p.A = src; p.W = m->conv_w[i]; p.C = dst;
    p.M = nw * L[i]; p.N = 512; p.K = kKernels[i] * 512; p.ldc = 512;
    p.rows_per_batch = L[i];
    p.a_row_stride = (long long)kStrides[i] * 512;
    p.a_batch_stride = (long long)L[i - 1] * 512;
    gemm_go<F16, 128, EPI_GELU>(m, p, 1, s);
    std::swap(src, dst);
    if (m->debug_stop == i) return 0;
  }
  // src = conv6 output (nw*T, 512); dst is free
  const int M = nw * T;
  LAUNCH(m, PC_LAYERNORM, 0.0, (double)M * 512 * 4, s,
         hipLaunchKernelGGL((layernorm512_kernel<F16>), dim3((M + 3) / 4), dim3(256), 0, s, src, m->fp_ln_w, m->fp_ln_b, dst, M));
  gemm_go<F16, 128, EPI_BIAS | EPI_F32>(m, plain(dst, m->fp_w, m->h0, m->fp_b, nullptr, M, 768, 512), 1, s);
  if (m->debug_stop == 7) return 0;

  // ---- positional conv (16 groups x [T x 6144] x [6144 x 48]) + GELU + residual, then LayerNorm
  LAUNCH(m, PC_PACK, 0.0, (double)M * 768 * 6, s,
         hipLaunchKernelGGL((posconv_pack_kernel<F16>), dim3((T + 128 + 3) / 4, nw), dim3(192), 0, s, m->h0, m->posin, T));
  LAUNCH(m, PC_GEMM, 2.0 * M * 768.0 * 48 * 128, 0.0, s,
         hipLaunchKernelGGL((posconv_kernel<F16>), dim3((T + kPcRows - 1) / kPcRows, 16, nw), dim3(256), 0, s, m->posin,
                            m->pos_w, m->pos_b, m->h0, m->tmp32, T));
  // ---- entry of the LayerNorm-free layers: the fp32 pre-LN rows rounded to 16 bit + the statistics of the encoder LayerNorm
  HeadParams nohead{};
  const dim3 ln_grid((M + 4 * kLnRows - 1) / (4 * kLnRows)), ln_blk(256);
  LAUNCH(m, PC_LAYERNORM, 0.0, (double)M * 768 * 6, s,
         hipLaunchKernelGGL((ln_round_stats_kernel<F16>), dim3((M + 3) / 4), dim3(256), 0, s, m->tmp32, m->x16, m->ln_stats, M));
  // test hook: materialise LayerNorm(u) (what the oracle calls the stage output) in ctx16, which is free between layers
  auto debug_ln = [&](const float* g, const float* b) {
    hipLaunchKernelGGL((layernorm768_kernel<F16, false, 1>), ln_grid, ln_blk, 0, s, (const void*)m->x16, g, b,
                       (float*)nullptr, m->ctx16, M, nohead);
  };
  if (m->debug_stop == 8) { debug_ln(m->enc_ln_w, m->enc_ln_b); return 0; }

  // ---- 12 post-LN transformer layers
  const int nkt = (T + 63) / 64;                       // 64-key tiles streamed through LDS
  const int nqb = ((T + 31) / 32 + kAttnWaves - 1) / kAttnWaves;   // blocks of 8 x 32 queries
  const double attn_flops = 4.0 * (double)nw * 12 * (double)T * T * 64;
  auto ln_in_gemm = [&](auto epi, const h16* W, void* C, const float* bias, const float* colsum, int N) {
    GemmParams p = plain(m->x16, W, C, bias, nullptr, M, N, 768);
    p.ln_stats = m->ln_stats; p.colsum = colsum;
    gemm_go<F16, 128, decltype(epi)::value>(m, p, 1, s);
  };
  auto ln_res_gemm = [&](const h16* A, const h16* W, const float* bias, const float* gamma, int K) {
    GemmParams p = plain(A, W, m->x16, bias, nullptr, M, 768, K);
    p.ln_stats = m->ln_stats; p.res_h = m->x16; p.res_g = gamma; p.part = m->ln_part; p.part_rows = (M + 15) & ~15;
    gemm_go<F16, 128, EPI_BIAS | EPI_LNRES>(m, p, 1, s);
    LAUNCH(m, PC_LAYERNORM, 0.0, (double)M * 136, s,
           hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, m->ln_part, m->ln_stats, M, (M + 15) & ~15));
  };
  for (int l = 0; l < VTC_N_LAYERS; ++l) {
    const LayerDev& w = m->layer[l];
    ln_in_gemm(std::integral_constant<int, EPI_BIAS | EPI_LNIN>{}, w.qkv_w, m->qkv16, w.qkv_b, w.qkv_s, 2304);
    LAUNCH(m, PC_ATTN, attn_flops, 0.0, s,
           hipLaunchKernelGGL((attention_kernel<F16>), dim3(nqb, nw * 12), dim3(512), 0, s, m->qkv16, m->ctx16, T, nkt));
    ln_res_gemm(m->ctx16, w.o_w, w.o_bb, w.g_in, 768);          // u1 = LN_in(u) + ctx W_o^T + b_o, statistics of ln1
    ln_in_gemm(std::integral_constant<int, EPI_BIAS | EPI_GELU | EPI_LNIN>{}, w.ff1_w, m->ffn16, w.ff1_b, w.ff1_s, 3072);
    ln_res_gemm(m->ffn16, w.ff2_w, w.ff2_bb, w.ln1_w, 3072);    // u2 = LN_1(u1) + h W_2^T + b_2, statistics of ln2
    if (m->debug_stop == 9 + l) { debug_ln(w.ln2_w, w.ln2_b); return 0; }
  }
  // ---- final LayerNorm with the per-frame multi-label head fused; writes stitched logits
  {
    const LayerDev& w = m->layer[VTC_N_LAYERS - 1];
    HeadParams hp{m->head_w, m->head_b, logits, m->spec.n_labels, T, wt};
    LAUNCH(m, PC_LAYERNORM, 0.0, (double)M * 768 * 2, s,
           hipLaunchKernelGGL((layernorm768_kernel<F16, true, 1>), ln_grid, ln_blk, 0, s, (const void*)m->x16, w.ln2_w, w.ln2_b,
                              (float*)nullptr, (h16*)nullptr, M, hp));
  }
  return 0;
}

// n frames in n_files recordings: runs never cross a recording, so a label has at most sum ceil(n_f / 2) <=
// (n + n_files) / 2 runs (many 1-frame recordings, all active, reach it).
long long max_runs_per_label(long long n, long long n_files) { return (n + n_files) / 2 + 1; }

int ensure_decode_scratch(vtc_model* m, long long n, long long n_files) {
  const long long runs = max_runs_per_label(n, n_files);
  if (n <= m->dec_frames_cap && runs <= m->dec_run_cap) return 0;
  for (void* p : {(void*)m->dec_counts, (void*)m->dec_totals, (void*)m->dec_start, (void*)m->dec_end, (void*)m->dec_sig,
                  (void*)m->dec_summary, (void*)m->dec_max, (void*)m->dec_fstart})
    if (p) (void)hipFree(p);
  m->dec_counts = m->dec_totals = m->dec_start = m->dec_end = nullptr;
  m->dec_sig = m->dec_summary = nullptr; m->dec_max = nullptr; m->dec_fstart = nullptr;
  const long long cap = std::max(n + n / 4 + 1024, m->dec_frames_cap);     // neither capacity ever shrinks
  const long long rcap = std::max(runs + runs / 4 + 64, m->dec_run_cap);
  m->dec_frames_cap = 0;
  m->dec_run_cap = 0;
  const long long nblk = (cap + 255) / 256;
  const int nl = m->spec.n_labels;
  HIP_TRY(m, hipMalloc((void**)&m->dec_counts, sizeof(int) * 2 * nl * nblk));
  HIP_TRY(m, hipMalloc((void**)&m->dec_totals, sizeof(int) * 2 * VTC_MAX_LABELS));
  HIP_TRY(m, hipMalloc((void**)&m->dec_start, sizeof(int) * nl * rcap));
  HIP_TRY(m, hipMalloc((void**)&m->dec_end, sizeof(int) * nl * rcap));
  HIP_TRY(m, hipMalloc((void**)&m->dec_sig, sizeof(double) * nblk));
  HIP_TRY(m, hipMalloc((void**)&m->dec_summary, sizeof(double) * 2));
  HIP_TRY(m, hipMalloc((void**)&m->dec_max, sizeof(float) * nblk));
  HIP_TRY(m, hipMalloc((void**)&m->dec_fstart, (size_t)cap + 1));
  m->dec_frames_cap = cap;
  m->dec_run_cap = rcap;
  return 0;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* vtc_last_error(const vtc_model* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int64_t vtc_frames_for_samples(int64_t n) { return n < kRfSize ? 0 : (n - kRfSize) / kRfStep + 1; }

int vtc_frames_per_window(const vtc_model* m) { return m ? m->T : VTC_ERR_ARG; }

const char* vtc_profile_class_name(int32_t c) { return (c >= 0 && c < VTC_PROF_CLASSES) ? kProfNames[c] : ""; }

void vtc_destroy(vtc_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  for (void* p : m->allocs) (void)hipFree(p);
  for (void* p : {(void*)m->dec_counts, (void*)m->dec_totals, (void*)m->dec_start, (void*)m->dec_end, (void*)m->dec_sig,
                  (void*)m->dec_summary, (void*)m->dec_max, (void*)m->dec_fstart, (void*)m->fsum_rows, (void*)m->fsum_out})
    if (p) (void)hipFree(p);
  if (m->dec_host) (void)hipHostFree(m->dec_host);
  if (m->dec_event) (void)hipEventDestroy(m->dec_event);
  if (m->h_stage) (void)hipHostFree(m->h_stage);
  for (auto& e : m->slot_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : m->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete m;
}

int vtc_create(const vtc_spec* spec, const vtc_weights* w, int device, vtc_model** out) {
  if (!spec || !w || !out) return fail(nullptr, VTC_ERR_ARG, "null argument");
  *out = nullptr;
  if (spec->abi_version != VTC_ABI_VERSION) return fail(nullptr, VTC_ERR_ARG, "ABI version %d != %d", spec->abi_version, VTC_ABI_VERSION);
  if (spec->n_labels < 1 || spec->n_labels > VTC_MAX_LABELS) return fail(nullptr, VTC_ERR_ARG, "n_labels %d out of range", spec->n_labels);
  if (spec->max_batch < 1 || spec->max_batch > 4096) return fail(nullptr, VTC_ERR_ARG, "max_batch %d out of range", spec->max_batch);
  if (spec->mfma_dtype != 0 && spec->mfma_dtype != 1) return fail(nullptr, VTC_ERR_ARG, "mfma_dtype must be 0 (bf16) or 1 (f16)");
  const long long T = vtc_frames_for_samples(spec->window_samples);
  if (T < 1 || T > 2048)   // attention streams K/V, nothing is sized by T but the scratch buffers; 2048 frames = 41 s
    return fail(nullptr, VTC_ERR_ARG, "window of %d samples gives %lld frames; supported 1..2048",
                spec->window_samples, T);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, VTC_ERR_NO_GPU, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(nullptr, VTC_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, VTC_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, VTC_ERR_NO_GPU, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

  vtc_model* m = new vtc_model();
  auto bail = [&](int rc) { g_create_error = m->err; vtc_destroy(m); return rc; };
  m->spec = *spec;
  m->device = device;
  m->f16 = spec->mfma_dtype == 1;
  if (const char* e = getenv("VTC_GEMM_TILE")) m->big_tiles = atoi(e) != 128;
  if (const char* e = getenv("VTC_CONV0_MFMA")) m->conv0_mfma = atoi(e) != 0;
  m->T = (int)T;
  {
    int n = spec->window_samples;
    for (int i = 0; i < 7; ++i) { n = (n - kKernels[i]) / kStrides[i] + 1; m->L[i] = n; }
  }
  if (m->L[6] != m->T) return bail(fail(m, VTC_ERR_ARG, "geometry mismatch"));
  if (hipSetDevice(device) != hipSuccess) return bail(fail(m, VTC_ERR_HIP, "hipSetDevice(%d) failed", device));
  if (device < kMaxDevices) {   // once per device, with that device current: where this GPU keeps its copy of `scratch`
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
      void* ptr = nullptr;
      const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !ptr) return bail(fail(m, VTC_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)));
      if (hipEventCreateWithFlags(&g_static_done[device], hipEventDisableTiming) != hipSuccess)
        return bail(fail(m, VTC_ERR_HIP, "hipEventCreate for the static decode's overlap guard failed"));
      g_scratch_ptr[device] = (unsigned char*)ptr;
    }
  }

#define TRY(expr) do { int rc_ = (expr); if (rc_) return bail(rc_); } while (0)
  const bool f16 = m->f16;
  // ---- weights
  TRY(upload_f32(m, &m->conv0_w, w->conv_w[0], 512 * 10));
  TRY(upload_f32(m, &m->gn_w, w->gn_w, 512));
  TRY(upload_f32(m, &m->gn_b, w->gn_b, 512));
  for (int i = 1; i < 7; ++i) {
    if (!w->conv_w[i]) return bail(fail(m, VTC_ERR_ARG, "missing conv weight %d", i));
    TRY(upload_h16(m, &m->conv_w[i], pack_conv(w->conv_w[i], 512, 512, kKernels[i], f16)));
  }
  TRY(upload_f32(m, &m->fp_ln_w, w->fp_ln_w, 512));
  TRY(upload_f32(m, &m->fp_ln_b, w->fp_ln_b, 512));
  if (!w->fp_w || !w->pos_w) return bail(fail(m, VTC_ERR_ARG, "missing projection / positional weight"));
  TRY(upload_h16(m, &m->fp_w, pack_linear(w->fp_w, 768, 512, f16)));
  TRY(upload_f32(m, &m->fp_b, w->fp_b, 768));
  TRY(upload_h16(m, &m->pos_w, pack_posconv(w->pos_w, f16)));
  TRY(upload_f32(m, &m->pos_b, w->pos_b, 768));
  TRY(upload_f32(m, &m->enc_ln_w, w->enc_ln_w, 768));
  TRY(upload_f32(m, &m->enc_ln_b, w->enc_ln_b, 768));
  // Host-side packing of the 12 layers (LayerNorm folding with fp64 column sums, 16-bit rounding: 7 M weights per layer) runs
  // on one thread per layer; the uploads follow in layer order.  Single-threaded this was most of the ~1.1 s a model takes to
  // construct, which every worker process and every test pays.
  for (int l = 0; l < VTC_N_LAYERS; ++l) {
    const vtc_layer_weights& s = w->layer[l];
    if (!s.q_w || !s.k_w || !s.v_w || !s.q_b || !s.k_b || !s.v_b || !s.o_w || !s.ff1_w || !s.ff2_w)
      return bail(fail(m, VTC_ERR_ARG, "missing weight in layer %d", l));
    if (!s.o_b || !s.ln1_w || !s.ln1_b || !s.ff1_b || !s.ff2_b || !s.ln2_w || !s.ln2_b)
      return bail(fail(m, VTC_ERR_ARG, "missing bias / LayerNorm tensor in layer %d", l));
    // LayerNorm that feeds this layer: the previous layer's final_layer_norm, the encoder's input LayerNorm for layer 0
    const float* g_in = l == 0 ? w->enc_ln_w : w->layer[l - 1].ln2_w;
    const float* b_in = l == 0 ? w->enc_ln_b : w->layer[l - 1].ln2_b;
    if (!g_in || !b_in) return bail(fail(m, VTC_ERR_ARG, "missing LayerNorm in front of layer %d", l));
  }
  struct LayerHost { FoldedLinear fq, f1; std::vector<unsigned short> ow, f2w; std::vector<float> obb, f2bb; };
  std::vector<LayerHost> host(VTC_N_LAYERS);
  {
    std::vector<std::thread> pack;
    for (int l = 0; l < VTC_N_LAYERS; ++l)
      pack.emplace_back([&, l]() {
        const vtc_layer_weights& s = w->layer[l];
        LayerHost& h = host[l];
        const float* g_in = l == 0 ? w->enc_ln_w : w->layer[l - 1].ln2_w;
        const float* b_in = l == 0 ? w->enc_ln_b : w->layer[l - 1].ln2_b;
        // fused QKV [2304][768] behind that LayerNorm; the 64^-0.5 query scale (HF modeling_hubert.py:287,318) and the log2(e)
        // of the softmax's exp2 form (attention.hip.h) are folded into the query rows
        const float kQScale = 0.125f * 1.4426950408889634f;
        std::vector<float> wq((size_t)2304 * 768), bq(2304);
        memcpy(&wq[0], s.q_w, sizeof(float) * 768 * 768);
        memcpy(&wq[(size_t)768 * 768], s.k_w, sizeof(float) * 768 * 768);
        memcpy(&wq[(size_t)1536 * 768], s.v_w, sizeof(float) * 768 * 768);
        memcpy(&bq[0], s.q_b, sizeof(float) * 768);
        memcpy(&bq[768], s.k_b, sizeof(float) * 768);
        memcpy(&bq[1536], s.v_b, sizeof(float) * 768);
        h.fq = fold_layernorm(wq.data(), bq.data(), 2304, 768, g_in, b_in, f16, kQScale, 768);
        h.ow = pack_linear(s.o_w, 768, 768, f16);
        // out-proj adds the residual LN_in(u): its beta joins the bias
        h.obb.resize(768); h.f2bb.resize(768);
        for (int i = 0; i < 768; ++i) { h.obb[i] = s.o_b[i] + b_in[i]; h.f2bb[i] = s.ff2_b[i] + s.ln1_b[i]; }
        h.f1 = fold_layernorm(s.ff1_w, s.ff1_b, 3072, 768, s.ln1_w, s.ln1_b, f16);
        h.f2w = pack_linear(s.ff2_w, 768, 3072, f16);
      });
    for (auto& t : pack) t.join();
  }
  for (int l = 0; l < VTC_N_LAYERS; ++l) {
    const vtc_layer_weights& s = w->layer[l];
    LayerDev& d = m->layer[l];
    const LayerHost& h = host[l];
    TRY(upload_h16(m, &d.qkv_w, h.fq.w));
    TRY(upload_f32(m, &d.qkv_b, h.fq.bias.data(), 2304));
    TRY(upload_f32(m, &d.qkv_s, h.fq.colsum.data(), 2304));
    TRY(upload_f32(m, &d.g_in, l == 0 ? w->enc_ln_w : w->layer[l - 1].ln2_w, 768));
    TRY(upload_h16(m, &d.o_w, h.ow));
    TRY(upload_f32(m, &d.o_bb, h.obb.data(), 768));
    TRY(upload_f32(m, &d.ln1_w, s.ln1_w, 768));
    TRY(upload_f32(m, &d.ln1_b, s.ln1_b, 768));
    TRY(upload_h16(m, &d.ff1_w, h.f1.w));
    TRY(upload_f32(m, &d.ff1_b, h.f1.bias.data(), 3072));
    TRY(upload_f32(m, &d.ff1_s, h.f1.colsum.data(), 3072));
    TRY(upload_h16(m, &d.ff2_w, h.f2w));
    TRY(upload_f32(m, &d.ff2_bb, h.f2bb.data(), 768));
    TRY(upload_f32(m, &d.ln2_w, s.ln2_w, 768));
    TRY(upload_f32(m, &d.ln2_b, s.ln2_b, 768));
    host[l] = LayerHost();   // release the layer's host copies as soon as they are on the device
  }
  TRY(upload_f32(m, &m->head_w, w->head_w, (size_t)spec->n_labels * 768));
  TRY(upload_f32(m, &m->head_b, w->head_b, spec->n_labels));

  // ---- workspace
  const size_t B = spec->max_batch, MT = B * (size_t)m->T;
  TRY(dev_alloc(m, &m->actA, B * m->L[0] * 512));
  TRY(dev_alloc(m, &m->actB, std::max(B * (size_t)m->L[1] * 512, MT * 512)));
  TRY(dev_alloc(m, &m->h0, MT * 768));
  TRY(dev_alloc(m, &m->tmp32, MT * 768));
  TRY(dev_alloc(m, &m->x16, MT * 768));
  TRY(dev_alloc(m, &m->ln_stats, MT * 2));
  TRY(dev_alloc(m, &m->ln_part, (MT + 16) * 32));
  TRY(dev_alloc(m, &m->qkv16, MT * 2304));
  TRY(dev_alloc(m, &m->ctx16, MT * 768));
  TRY(dev_alloc(m, &m->ffn16, MT * 3072));
  TRY(dev_alloc(m, &m->posin, B * 16 * (size_t)(m->T + 128) * 48));
  TRY(dev_alloc(m, &m->gn_scale, B * 512));
  TRY(dev_alloc(m, &m->gn_shift, B * 512));
  m->nchunk = (m->L[0] + kStatChunk - 1) / kStatChunk;
  TRY(dev_alloc(m, &m->mom_partial, B * m->nchunk * kStatSlots));
  TRY(dev_alloc(m, &m->c0_aop, B * 512 * 32));
  TRY(dev_alloc(m, &m->c0_xscale, B));
  TRY(dev_alloc(m, &m->c0_flag, B));
  {   // window table: one allocation in the layout of a staging slot (vtc_forward_windows copies it in one piece)
    char* tab = nullptr;
    TRY(dev_alloc(m, &tab, B * 24));
    m->d_start = (long long*)tab;
    m->d_dst = (long long*)(tab + B * 8);
    m->d_valid = (int*)(tab + B * 16);
    m->d_keep = (int*)(tab + B * 20);
  }
  if (hipHostMalloc((void**)&m->h_stage, (size_t)kSlots * B * 24, hipHostMallocDefault) != hipSuccess)
    return bail(fail(m, VTC_ERR_HIP, "hipHostMalloc(staging) failed"));
  for (auto& e : m->slot_ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(fail(m, VTC_ERR_HIP, "hipEventCreate failed"));

#undef TRY
  *out = m;
  return VTC_OK;
}

int vtc_forward_windows(vtc_model* m, const float* wav, const int64_t* start, const int32_t* valid, const int64_t* dst_row,
                        const int32_t* keep, int32_t nw, float* logits, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!wav || !start || !valid || !dst_row || !keep || !logits) return fail(m, VTC_ERR_ARG, "null argument");
  if (nw < 1 || nw > m->spec.max_batch) return fail(m, VTC_ERR_ARG, "n_windows %d not in 1..%d", nw, m->spec.max_batch);
  for (int i = 0; i < nw; ++i) {
    if (start[i] < 0 || valid[i] < 0 || valid[i] > m->spec.window_samples || keep[i] < 0 || keep[i] > m->T || dst_row[i] < 0)
      return fail(m, VTC_ERR_ARG, "window %d: start %lld valid %d keep %d dst_row %lld out of range", i, (long long)start[i],
                  valid[i], keep[i], (long long)dst_row[i]);
  }
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t B = m->spec.max_batch;
  const int slot = m->slot;
  m->slot = (m->slot + 1) % kSlots;
  if (m->slot_used[slot]) HIP_TRY(m, hipEventSynchronize(m->slot_ev[slot]));
  char* hs = m->h_stage + (size_t)slot * B * 24;
  long long* hs_start = (long long*)hs;
  long long* hs_dst = (long long*)(hs + B * 8);
  int* hs_valid = (int*)(hs + B * 16);
  int* hs_keep = (int*)(hs + B * 20);
  for (int i = 0; i < nw; ++i) { hs_start[i] = start[i]; hs_dst[i] = dst_row[i]; hs_valid[i] = valid[i]; hs_keep[i] = keep[i]; }
  // one copy for the whole table: the device arrays are laid out as the staging slot is ([start][dst][valid][keep], each
  // sized for max_batch), so a forward costs one 24 B x max_batch transfer instead of four small ones (~4 us each)
  HIP_TRY(m, hipMemcpyAsync(m->d_start, hs, B * 24, hipMemcpyHostToDevice, s));
  HIP_TRY(m, hipEventRecord(m->slot_ev[slot], s));
  m->slot_used[slot] = true;
  const int rc = m->f16 ? forward_impl<true>(m, wav, nw, logits, s) : forward_impl<false>(m, wav, nw, logits, s);
  if (rc) return rc;
  HIP_TRY(m, hipGetLastError());
  return VTC_OK;
}

int vtc_infer_waveform(vtc_model* m, const float* wav, int64_t n_samples, int32_t batch, float* logits, int64_t* n_frames,
                       void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!wav || !logits || !n_frames) return fail(m, VTC_ERR_ARG, "null argument");
  if (batch < 1) return fail(m, VTC_ERR_ARG, "batch %d < 1", batch);
  if (batch > m->spec.max_batch) batch = m->spec.max_batch;
  *n_frames = 0;
  if (n_samples < kRfSize) return fail(m, VTC_ERR_SHORT, "audio too short: %lld samples < %d", (long long)n_samples, kRfSize);
  const long long W = m->spec.window_samples, T = m->T, step = T * kRfStep;  // chunk_artifact.py:71-73
  std::vector<int64_t> start, dst;
  std::vector<int32_t> valid, keep;
  long long total = 0;
  for (long long s0 = 0; n_samples - s0 >= kRfSize; s0 += step) {
    const long long rem = n_samples - s0;
    start.push_back(s0);
    valid.push_back((int32_t)std::min(rem, W));
    const long long k = std::min<long long>(T, vtc_frames_for_samples(rem));
    keep.push_back((int32_t)k);
    dst.push_back(total);
    total += k;
  }
  for (size_t i = 0; i < start.size(); i += batch) {
    const int nw = (int)std::min<size_t>(batch, start.size() - i);
    const int rc = vtc_forward_windows(m, wav, &start[i], &valid[i], &dst[i], &keep[i], nw, logits, stream);
    if (rc) return rc;
  }
  *n_frames = total;
  return VTC_OK;
}

// Shared by vtc_decode and vtc_decode_files: runs per label as frame indices (label-major, time-ordered).
static int decode_impl(vtc_model* m, const float* logits, int64_t n, const float* lo, const float* hi,
                       const unsigned char* file_start_dev, std::vector<int> runs_start[], std::vector<int> runs_end[],
                       float* max_logit, double* mean_sigmoid, hipStream_t s) {
  const int nl = m->spec.n_labels;
  DecodeParams p{};
  p.logits = logits; p.n = n; p.nl = nl; p.nblk = (int)((n + 255) / 256);
  p.file_start = file_start_dev;
  for (int l = 0; l < nl; ++l) { p.lo[l] = lo[l]; p.hi[l] = hi[l]; }
  const long long dcap = m->dec_run_cap;
  LAUNCH(m, PC_DECODE, 0.0, (double)n * nl * 4 * 2, s, {
    hipLaunchKernelGGL(decode_count_kernel, dim3(p.nblk), dim3(256), 0, s, p, m->dec_counts, m->dec_sig, m->dec_max);
    hipLaunchKernelGGL(decode_scan_kernel, dim3(2 * nl), dim3(1024), 0, s, m->dec_counts, p.nblk, m->dec_totals, m->dec_sig,
                       m->dec_max, m->dec_summary);
    hipLaunchKernelGGL(decode_emit_kernel, dim3(p.nblk), dim3(256), 0, s, p, m->dec_counts, dcap, m->dec_start, m->dec_end);
  });
  int totals[2 * VTC_MAX_LABELS];
  double summary[2];
  HIP_TRY(m, hipMemcpyAsync(totals, m->dec_totals, sizeof(int) * 2 * nl, hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipMemcpyAsync(summary, m->dec_summary, sizeof summary, hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipStreamSynchronize(s));
  if (max_logit) *max_logit = (float)summary[1];
  if (mean_sigmoid) *mean_sigmoid = summary[0] / ((double)n * nl);
  for (int l = 0; l < nl; ++l) {
    if (totals[l] != totals[nl + l]) return fail(m, VTC_ERR_HIP, "decode: label %d has %d starts but %d ends", l, totals[l], totals[nl + l]);
    if (totals[l] < 0 || totals[l] > dcap)   // the emit kernel dropped runs: never hand out what lies past the label's region
      return fail(m, VTC_ERR_HIP, "decode: label %d has %d runs, scratch holds %lld", l, totals[l], dcap);
    const int c = totals[l];
    runs_start[l].resize(c); runs_end[l].resize(c);
    if (!c) continue;
    // on the caller's stream, never the null stream: a caller that has already queued the next forward on another
    // stream must not wait for it here
    HIP_TRY(m, hipMemcpyAsync(runs_start[l].data(), m->dec_start + (long long)l * dcap, sizeof(int) * c, hipMemcpyDeviceToHost, s));
    HIP_TRY(m, hipMemcpyAsync(runs_end[l].data(), m->dec_end + (long long)l * dcap, sizeof(int) * c, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(m, hipStreamSynchronize(s));
  return VTC_OK;
}

int vtc_decode(vtc_model* m, const float* logits, int64_t n, const float* lo, const float* hi, int64_t* out_start,
               int64_t* out_end, int32_t* out_label, int64_t cap, int64_t* n_out, float* max_logit, double* mean_sigmoid,
               void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!logits || !lo || !hi || !n_out || (cap > 0 && (!out_start || !out_end || !out_label)))
    return fail(m, VTC_ERR_ARG, "null argument");
  if (n < 1 || n > 0x7fffff00LL) return fail(m, VTC_ERR_ARG, "n_frames %lld out of range", (long long)n);
  HIP_TRY(m, hipSetDevice(m->device));
  if (int rc = ensure_decode_scratch(m, n, 1)) return rc;
  const int nl = m->spec.n_labels;
  std::vector<int> rs[VTC_MAX_LABELS], re[VTC_MAX_LABELS];
  float mx_all = 0.f;
  if (int rc = decode_impl(m, logits, n, lo, hi, nullptr, rs, re, &mx_all, mean_sigmoid, (hipStream_t)stream)) return rc;
  if (max_logit) *max_logit = mx_all;
  if (mx_all != mx_all) return fail(m, VTC_ERR_NONFINITE, "non-finite logits (an activation left the 16-bit range)");
  long long total = 0;
  for (int l = 0; l < nl; ++l) total += (long long)rs[l].size();
  *n_out = total;
  if (total > cap) return fail(m, VTC_ERR_CAPACITY, "decode: %lld intervals, caller capacity %lld", total, (long long)cap);
  long long at = 0;
  for (int l = 0; l < nl; ++l)
    for (size_t k = 0; k < rs[l].size(); ++k, ++at) {
      out_start[at] = (int64_t)rs[l][k] * kRfStep;
      out_end[at] = (int64_t)re[l][k] * kRfStep + kRfSize;
      out_label[at] = l;
    }
  return VTC_OK;
}

int vtc_decode_static(vtc_model* m, const float* logits, int64_t n, const float* lo, const float* hi, int64_t* out_start,
                      int64_t* out_end, int32_t* out_label, int64_t cap, int64_t* result, double* result_summary, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!logits || !lo || !hi || !result || !result_summary || cap < 0 || (cap > 0 && (!out_start || !out_end || !out_label)))
    return fail(m, VTC_ERR_ARG, "null argument");
  if (n < 1) return fail(m, VTC_ERR_ARG, "n_frames %lld out of range", (long long)n);
  if (n > kStMaxFrames)
    return fail(m, VTC_ERR_CAPACITY, "vtc_decode_static holds %lld frames at most, got %lld (use vtc_decode)", kStMaxFrames,
                (long long)n);
  unsigned char* base = m->device < kMaxDevices ? g_scratch_ptr[m->device] : nullptr;
  if (!base) return fail(m, VTC_ERR_HIP, "static scratch of device %d was not resolved", m->device);
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (s) HIP_TRY(m, hipStreamIsCapturing(s, &capture));
  const bool guarded = capture == hipStreamCaptureStatusNone;
  std::unique_lock<std::mutex> guard(g_scratch_mu, std::defer_lock);
  if (guarded) {
    guard.lock();   // held until this call's event is recorded: two host threads cannot both pass the check
    if (g_static_pending[m->device] && g_static_stream[m->device] != s) {
      const hipError_t q = hipEventQuery(g_static_done[m->device]);
      (void)hipGetLastError();
      if (q == hipErrorNotReady)
        return fail(m, VTC_ERR_ARG, "vtc_decode_static: the previous call on device %d, on another stream, has not finished; "
                    "calls on one GPU share one scratch array and must not overlap", m->device);
      if (q != hipSuccess) return fail(m, VTC_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
    }
  }
  const int nl = m->spec.n_labels;
  int* counts = (int*)(base + kStOffCounts);
  int* totals = (int*)(base + kStOffTotals);
  double* summary = (double*)(base + kStOffSummary);
  double* sig = (double*)(base + kStOffSig);
  float* mx = (float*)(base + kStOffMax);
  int* run_start = (int*)(base + kStOffStart);
  int* run_end = (int*)(base + kStOffEnd);
  DecodeParams p{};
  p.logits = logits; p.n = n; p.nl = nl; p.nblk = (int)((n + 255) / 256);
  p.file_start = nullptr;
  for (int l = 0; l < nl; ++l) { p.lo[l] = lo[l]; p.hi[l] = hi[l]; }
  const long long rcap = n / 2 + 1;   // <= kStRunCap; the label regions are laid out with this stride
  hipLaunchKernelGGL(decode_count_kernel, dim3(p.nblk), dim3(256), 0, s, p, counts, sig, mx);
  hipLaunchKernelGGL(decode_scan_kernel, dim3(2 * nl), dim3(1024), 0, s, counts, p.nblk, totals, sig, mx, summary);
  hipLaunchKernelGGL(decode_emit_kernel, dim3(p.nblk), dim3(256), 0, s, p, counts, rcap, run_start, run_end);
  hipLaunchKernelGGL(decode_pack_kernel, dim3((unsigned)((rcap + 255) / 256), nl), dim3(256), 0, s, totals, run_start, run_end, rcap,
                     nl, (long long)cap, (long long*)out_start, (long long*)out_end, out_label, (long long*)result, summary,
                     (double)n * nl, result_summary);
  HIP_TRY(m, hipGetLastError());
  if (guarded) {
    HIP_TRY(m, hipEventRecord(g_static_done[m->device], s));
    g_static_stream[m->device] = s;
    g_static_pending[m->device] = true;
  }
  return VTC_OK;
}

int vtc_threshold_mask(vtc_model* m, const float* logits, int64_t n, const float* lo, const float* hi, uint8_t* mask,
                       void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!logits || !lo || !hi || !mask) return fail(m, VTC_ERR_ARG, "null argument");
  if (n < 0 || n > 0x7fffff00LL) return fail(m, VTC_ERR_ARG, "n_frames %lld out of range", (long long)n);
  if (n == 0) return VTC_OK;
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  const int nl = m->spec.n_labels;
  DecodeParams p{};
  p.logits = logits; p.n = n; p.nl = nl; p.nblk = 0;
  for (int l = 0; l < nl; ++l) { p.lo[l] = lo[l]; p.hi[l] = hi[l]; }
  const long long blocks = (n * nl + 255) / 256;
  LAUNCH(m, PC_DECODE, 0.0, (double)n * nl * 5, s,
         hipLaunchKernelGGL(threshold_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, mask));
  HIP_TRY(m, hipGetLastError());
  return VTC_OK;
}

int vtc_infer_waveforms(vtc_model* m, const float* wav, const int64_t* offset, const int64_t* n_samples, int32_t n_files,
                        int32_t batch, float* logits, int64_t* file_rows, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!wav || !offset || !n_samples || !logits || !file_rows) return fail(m, VTC_ERR_ARG, "null argument");
  if (n_files < 1) return fail(m, VTC_ERR_ARG, "n_files %d < 1", n_files);
  if (batch < 1) return fail(m, VTC_ERR_ARG, "batch %d < 1", batch);
  if (batch > m->spec.max_batch) batch = m->spec.max_batch;
  const long long W = m->spec.window_samples, T = m->T, step = T * kRfStep;  // chunk_artifact.py:71-73
  std::vector<int64_t> start, dst;
  std::vector<int32_t> valid, keep;
  long long total = 0;
  for (int f = 0; f < n_files; ++f) {
    if (offset[f] < 0 || n_samples[f] < 0) return fail(m, VTC_ERR_ARG, "file %d: negative offset or length", f);
    file_rows[f] = total;
    for (long long s0 = 0; n_samples[f] - s0 >= kRfSize; s0 += step) {   // same plan as vtc_infer_waveform, per file
      const long long rem = n_samples[f] - s0;
      start.push_back(offset[f] + s0);
      valid.push_back((int32_t)std::min(rem, W));
      const long long k = std::min<long long>(T, vtc_frames_for_samples(rem));
      keep.push_back((int32_t)k);
      dst.push_back(total);
      total += k;
    }
  }
  file_rows[n_files] = total;
  for (size_t i = 0; i < start.size(); i += batch) {   // windows of different files share a forward pass
    const int nw = (int)std::min<size_t>(batch, start.size() - i);
    const int rc = vtc_forward_windows(m, wav, &start[i], &valid[i], &dst[i], &keep[i], nw, logits, stream);
    if (rc) return rc;
  }
  return VTC_OK;
}

int vtc_decode_files(vtc_model* m, const float* logits, const int64_t* file_rows, int32_t n_files, const float* lo,
                     const float* hi, int32_t* out_file, int64_t* out_start, int64_t* out_end, int32_t* out_label, int64_t cap,
                     int64_t* n_out, float* max_logit, double* mean_sigmoid, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!logits || !file_rows || !lo || !hi || !n_out || (cap > 0 && (!out_file || !out_start || !out_end || !out_label)))
    return fail(m, VTC_ERR_ARG, "null argument");
  if (n_files < 1) return fail(m, VTC_ERR_ARG, "n_files %d < 1", n_files);
  const int64_t n = file_rows[n_files];
  if (file_rows[0] != 0) return fail(m, VTC_ERR_ARG, "file_rows[0] must be 0");
  for (int f = 0; f < n_files; ++f)
    if (file_rows[f + 1] < file_rows[f]) return fail(m, VTC_ERR_ARG, "file_rows must not decrease (file %d)", f);
  *n_out = 0;
  if (n == 0) return VTC_OK;
  if (n > 0x7fffff00LL) return fail(m, VTC_ERR_ARG, "n_frames %lld out of range", (long long)n);
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  if (int rc = ensure_decode_scratch(m, n, n_files)) return rc;
  std::vector<unsigned char> mask((size_t)n + 1, 0);
  for (int f = 0; f < n_files; ++f) if (file_rows[f] < n) mask[(size_t)file_rows[f]] = 1;
  HIP_TRY(m, hipMemcpyAsync(m->dec_fstart, mask.data(), mask.size(), hipMemcpyHostToDevice, s));
  const int nl = m->spec.n_labels;
  if (max_logit || mean_sigmoid) {   // per-recording sigmoid summary (vtc.py:266-273), one block per recording
    if (n_files > m->fsum_cap) {
      if (m->fsum_rows) (void)hipFree(m->fsum_rows);
      if (m->fsum_out) (void)hipFree(m->fsum_out);
      m->fsum_rows = nullptr; m->fsum_out = nullptr; m->fsum_cap = 0;
      const int capf = n_files + n_files / 2 + 64;
      HIP_TRY(m, hipMalloc((void**)&m->fsum_rows, sizeof(long long) * (capf + 1)));
      HIP_TRY(m, hipMalloc((void**)&m->fsum_out, sizeof(double) * 2 * capf));
      m->fsum_cap = capf;
    }
    HIP_TRY(m, hipMemcpyAsync(m->fsum_rows, file_rows, sizeof(long long) * (n_files + 1), hipMemcpyHostToDevice, s));
    LAUNCH(m, PC_DECODE, 0.0, (double)n * nl * 4, s,
           hipLaunchKernelGGL(file_summary_kernel, dim3(n_files), dim3(256), 0, s, logits, nl, m->fsum_rows, m->fsum_out));
  }
  std::vector<int> rs[VTC_MAX_LABELS], re[VTC_MAX_LABELS];
  float mx_all = 0.f;
  if (int rc = decode_impl(m, logits, n, lo, hi, m->dec_fstart, rs, re, &mx_all, nullptr, s)) return rc;   // synchronises s
  if (mx_all != mx_all && !max_logit)   // with max_logit the affected recordings are marked NaN below instead
    return fail(m, VTC_ERR_NONFINITE, "non-finite logits (an activation left the 16-bit range)");
  if (max_logit || mean_sigmoid) {
    std::vector<double> hsum((size_t)2 * n_files);
    HIP_TRY(m, hipMemcpyAsync(hsum.data(), m->fsum_out, sizeof(double) * 2 * n_files, hipMemcpyDeviceToHost, s));
    HIP_TRY(m, hipStreamSynchronize(s));
    for (int f = 0; f < n_files; ++f) {
      const long long cnt = (file_rows[f + 1] - file_rows[f]) * nl;
      if (max_logit) max_logit[f] = cnt ? (float)hsum[2 * f + 1] : -INFINITY;
      if (mean_sigmoid) mean_sigmoid[f] = cnt ? hsum[2 * f] / (double)cnt : 0.0;
    }
  }
  long long total = 0;
  for (int l = 0; l < nl; ++l) total += (long long)rs[l].size();
  *n_out = total;
  if (total > cap) return fail(m, VTC_ERR_CAPACITY, "decode: %lld intervals, caller capacity %lld", total, (long long)cap);
  // file of each run (runs are time-ordered per label, so the file index only moves forward), then a stable
  // counting sort by file: the result is file-major and, inside a file, label-major and time-ordered = vtc_decode per file
  std::vector<long long> per_file((size_t)n_files + 1, 0);
  std::vector<int> run_file[VTC_MAX_LABELS];
  for (int l = 0; l < nl; ++l) {
    run_file[l].resize(rs[l].size());
    int f = 0;
    for (size_t k = 0; k < rs[l].size(); ++k) {
      while (f + 1 < n_files && rs[l][k] >= file_rows[f + 1]) ++f;
      run_file[l][k] = f;
      per_file[(size_t)f + 1]++;
    }
  }
  for (int f = 0; f < n_files; ++f) per_file[(size_t)f + 1] += per_file[(size_t)f];
  for (int l = 0; l < nl; ++l)
    for (size_t k = 0; k < rs[l].size(); ++k) {
      const int f = run_file[l][k];
      const long long at = per_file[(size_t)f]++;
      out_file[at] = f;
      out_start[at] = ((int64_t)rs[l][k] - file_rows[f]) * kRfStep;
      out_end[at] = ((int64_t)re[l][k] - file_rows[f]) * kRfStep + kRfSize;
      out_label[at] = l;
    }
  return VTC_OK;
}

// ---- two-step decode: everything the GPU has to do is enqueued by _launch (kernels and the copies of their results
// into pinned memory), so the caller can queue the next forward right behind it and _collect waits for nothing newer.
namespace {
constexpr size_t kDecHostHead = 256;   // totals (2 * VTC_MAX_LABELS ints) at 0, summary (2 doubles) at 128

int ensure_file_scratch(vtc_model* m, int n_files) {
  if (n_files <= m->fsum_cap) return 0;
  if (m->fsum_rows) (void)hipFree(m->fsum_rows);
  if (m->fsum_out) (void)hipFree(m->fsum_out);
  m->fsum_rows = nullptr; m->fsum_out = nullptr; m->fsum_cap = 0;
  const int capf = n_files + n_files / 2 + 64;
  HIP_TRY(m, hipMalloc((void**)&m->fsum_rows, sizeof(long long) * (capf + 1)));
  HIP_TRY(m, hipMalloc((void**)&m->fsum_out, sizeof(double) * 2 * capf));
  m->fsum_cap = capf;
  return 0;
}
}  // namespace

int vtc_decode_launch(vtc_model* m, const float* logits, const int64_t* file_rows, int32_t n_files, const float* lo,
                      const float* hi, int32_t want_summary, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!logits || !file_rows || !lo || !hi) return fail(m, VTC_ERR_ARG, "null argument");
  if (m->dec_active) return fail(m, VTC_ERR_ARG, "vtc_decode_launch: the previous launch has not been collected");
  if (n_files < 1) return fail(m, VTC_ERR_ARG, "n_files %d < 1", n_files);
  if (file_rows[0] != 0) return fail(m, VTC_ERR_ARG, "file_rows[0] must be 0");
  for (int f = 0; f < n_files; ++f)
    if (file_rows[f + 1] < file_rows[f]) return fail(m, VTC_ERR_ARG, "file_rows must not decrease (file %d)", f);
  const int64_t n = file_rows[n_files];
  if (n > 0x7fffff00LL) return fail(m, VTC_ERR_ARG, "n_frames %lld out of range", (long long)n);
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  const int nl = m->spec.n_labels;
  m->dec_rows.assign(file_rows, file_rows + n_files + 1);
  m->dec_n = n;
  m->dec_want_summary = want_summary != 0;
  if (!m->dec_event) HIP_TRY(m, hipEventCreateWithFlags(&m->dec_event, hipEventDisableTiming));
  if (n == 0) {   // nothing to decode: collect returns zero runs
    m->dec_rcap = 0;
    HIP_TRY(m, hipEventRecord(m->dec_event, s));
    m->dec_active = true;
    return VTC_OK;
  }
  if (int rc = ensure_decode_scratch(m, n, n_files)) return rc;
  if (int rc = ensure_file_scratch(m, n_files)) return rc;
  const long long rcap = max_runs_per_label(n, n_files), dcap = m->dec_run_cap;
  const size_t off_fsum = kDecHostHead, off_rows = off_fsum + sizeof(double) * 2 * (size_t)n_files;
  const size_t off_rs = (off_rows + sizeof(int64_t) * ((size_t)n_files + 1) + 63) & ~(size_t)63;
  const size_t off_re = off_rs + sizeof(int) * (size_t)nl * rcap, need = off_re + sizeof(int) * (size_t)nl * rcap;
  if (need > m->dec_host_cap) {
    if (m->dec_host) (void)hipHostFree(m->dec_host);
    m->dec_host = nullptr; m->dec_host_cap = 0;
    const size_t cap = need + need / 4 + 4096;
    HIP_TRY(m, hipHostMalloc((void**)&m->dec_host, cap, hipHostMallocDefault));
    m->dec_host_cap = cap;
  }
  m->dec_rcap = rcap;
  char* h = m->dec_host;
  const unsigned char* fstart = nullptr;
  if (n_files > 1) {   // runs must not cross a recording: mark the first frame of every recording
    memcpy(h + off_rows, file_rows, sizeof(int64_t) * ((size_t)n_files + 1));
    HIP_TRY(m, hipMemcpyAsync(m->fsum_rows, h + off_rows, sizeof(int64_t) * ((size_t)n_files + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(m, hipMemsetAsync(m->dec_fstart, 0, (size_t)n + 1, s));
    hipLaunchKernelGGL(mark_file_starts_kernel, dim3((n_files + 255) / 256), dim3(256), 0, s, m->fsum_rows, n_files, (long long)n,
                       m->dec_fstart);
    fstart = m->dec_fstart;
    if (want_summary) {
      LAUNCH(m, PC_DECODE, 0.0, (double)n * nl * 4, s,
             hipLaunchKernelGGL(file_summary_kernel, dim3(n_files), dim3(256), 0, s, logits, nl, m->fsum_rows, m->fsum_out));
      HIP_TRY(m, hipMemcpyAsync(h + off_fsum, m->fsum_out, sizeof(double) * 2 * n_files, hipMemcpyDeviceToHost, s));
    }
  }
  DecodeParams p{};
  p.logits = logits; p.n = n; p.nl = nl; p.nblk = (int)((n + 255) / 256);
  p.file_start = fstart;
  for (int l = 0; l < nl; ++l) { p.lo[l] = lo[l]; p.hi[l] = hi[l]; }
  LAUNCH(m, PC_DECODE, 0.0, (double)n * nl * 4 * 2, s, {
    hipLaunchKernelGGL(decode_count_kernel, dim3(p.nblk), dim3(256), 0, s, p, m->dec_counts, m->dec_sig, m->dec_max);
    hipLaunchKernelGGL(decode_scan_kernel, dim3(2 * nl), dim3(1024), 0, s, m->dec_counts, p.nblk, m->dec_totals, m->dec_sig,
                       m->dec_max, m->dec_summary);
    hipLaunchKernelGGL(decode_emit_kernel, dim3(p.nblk), dim3(256), 0, s, p, m->dec_counts, dcap, m->dec_start, m->dec_end);
  });
  HIP_TRY(m, hipMemcpyAsync(h, m->dec_totals, sizeof(int) * 2 * nl, hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipMemcpyAsync(h + 128, m->dec_summary, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
  // the run lists at their full capacity ((n + n_files)/2 + 1 per label): their lengths are only known on the device, and a
  // second, length-sized copy would have to queue behind whatever the caller enqueues next
  HIP_TRY(m, hipMemcpy2DAsync(h + off_rs, sizeof(int) * rcap, m->dec_start, sizeof(int) * dcap, sizeof(int) * rcap, nl,
                              hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipMemcpy2DAsync(h + off_re, sizeof(int) * rcap, m->dec_end, sizeof(int) * dcap, sizeof(int) * rcap, nl,
                              hipMemcpyDeviceToHost, s));
  HIP_TRY(m, hipEventRecord(m->dec_event, s));
  HIP_TRY(m, hipGetLastError());
  m->dec_active = true;
  return VTC_OK;
}

int vtc_decode_collect(vtc_model* m, int32_t* out_file, int64_t* out_start, int64_t* out_end, int32_t* out_label, int64_t cap,
                       int64_t* n_out, float* max_logit, double* mean_sigmoid) {
  if (!m) return VTC_ERR_ARG;
  if (!n_out || (cap > 0 && (!out_file || !out_start || !out_end || !out_label))) return fail(m, VTC_ERR_ARG, "null argument");
  if (!m->dec_active) return fail(m, VTC_ERR_ARG, "vtc_decode_collect without vtc_decode_launch");
  HIP_TRY(m, hipSetDevice(m->device));
  if (hipError_t e = hipEventSynchronize(m->dec_event); e != hipSuccess) {
    m->dec_active = false;   // nothing to hand out; do not block later launches
    return fail(m, VTC_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
  }
  const int nl = m->spec.n_labels, n_files = (int)m->dec_rows.size() - 1;
  const int64_t* rows = m->dec_rows.data();
  const long long n = m->dec_n, rcap = m->dec_rcap;
  *n_out = 0;
  if (n == 0) {
    for (int f = 0; f < n_files; ++f) {
      if (max_logit) max_logit[f] = -INFINITY;
      if (mean_sigmoid) mean_sigmoid[f] = 0.0;
    }
    m->dec_active = false;
    return VTC_OK;
  }
  const char* h = m->dec_host;
  const int* totals = (const int*)h;
  const double* summary = (const double*)(h + 128);
  const size_t off_fsum = kDecHostHead, off_rows = off_fsum + sizeof(double) * 2 * (size_t)n_files;
  const size_t off_rs = (off_rows + sizeof(int64_t) * ((size_t)n_files + 1) + 63) & ~(size_t)63;
  const int* rs = (const int*)(h + off_rs);
  const int* re = rs + (size_t)nl * rcap;
  long long total = 0;
  for (int l = 0; l < nl; ++l) {
    if (totals[l] != totals[nl + l] || totals[l] < 0 || totals[l] > rcap) {
      m->dec_active = false;
      return fail(m, VTC_ERR_HIP, "decode: label %d has %d starts and %d ends (capacity %lld)", l, totals[l], totals[nl + l], rcap);
    }
    total += totals[l];
  }
  *n_out = total;
  if (summary[1] != summary[1] && !(m->dec_want_summary && max_logit)) {   // otherwise max_logit[f] = NaN marks the recordings
    m->dec_active = false;
    return fail(m, VTC_ERR_NONFINITE, "non-finite logits (an activation left the 16-bit range)");
  }
  if (total > cap)   // the results stay parked: call again with room for *n_out runs
    return fail(m, VTC_ERR_CAPACITY, "decode: %lld intervals, caller capacity %lld", total, (long long)cap);
  if (m->dec_want_summary) {
    if (n_files == 1) {
      if (max_logit) max_logit[0] = (float)summary[1];
      if (mean_sigmoid) mean_sigmoid[0] = summary[0] / ((double)n * nl);
    } else {
      const double* hsum = (const double*)(h + off_fsum);
      for (int f = 0; f < n_files; ++f) {
        const long long cnt = (rows[f + 1] - rows[f]) * nl;
        if (max_logit) max_logit[f] = cnt ? (float)hsum[2 * f + 1] : -INFINITY;
        if (mean_sigmoid) mean_sigmoid[f] = cnt ? hsum[2 * f] / (double)cnt : 0.0;
      }
    }
  }
  // recording of each run, then a stable counting sort by recording (as vtc_decode_files)
  std::vector<long long> per_file((size_t)n_files + 1, 0);
  std::vector<int> run_file((size_t)total);
  long long at0 = 0;
  for (int l = 0; l < nl; ++l) {
    int f = 0;
    for (int k = 0; k < totals[l]; ++k) {
      const int r = rs[(size_t)l * rcap + k];
      while (f + 1 < n_files && r >= rows[f + 1]) ++f;
      run_file[(size_t)at0 + k] = f;
      per_file[(size_t)f + 1]++;
    }
    at0 += totals[l];
  }
  for (int f = 0; f < n_files; ++f) per_file[(size_t)f + 1] += per_file[(size_t)f];
  at0 = 0;
  for (int l = 0; l < nl; ++l) {
    for (int k = 0; k < totals[l]; ++k) {
      const int f = run_file[(size_t)at0 + k];
      const long long at = per_file[(size_t)f]++;
      out_file[at] = f;
      out_start[at] = ((int64_t)rs[(size_t)l * rcap + k] - rows[f]) * kRfStep;
      out_end[at] = ((int64_t)re[(size_t)l * rcap + k] - rows[f]) * kRfStep + kRfSize;
      out_label[at] = l;
    }
    at0 += totals[l];
  }
  m->dec_active = false;
  return VTC_OK;
}

int vtc_debug_stop_after(vtc_model* m, int32_t stage) {
  if (!m || stage < -1 || stage > 9 + VTC_N_LAYERS - 1) return VTC_ERR_ARG;
  m->debug_stop = stage;
  return VTC_OK;
}

int64_t vtc_debug_stage(vtc_model* m, int32_t stage, int32_t nw, float* out, int64_t cap) {
  if (!m || !out || nw < 1 || nw > m->spec.max_batch) return VTC_ERR_ARG;
  if (hipSetDevice(m->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(m, VTC_ERR_HIP, "sync failed");
  const void* src = nullptr;
  long long n = 0;
  bool is16 = true;
  if (stage >= 0 && stage <= 6) { src = (stage % 2 == 0) ? m->actA : m->actB; n = (long long)nw * m->L[stage] * 512; }
  else if (stage == 7) { src = m->h0; n = (long long)nw * m->T * 768; is16 = false; }
  else if (stage >= 8 && stage < 9 + VTC_N_LAYERS) { src = m->ctx16; n = (long long)nw * m->T * 768; }   // debug_ln's output
  else return fail(m, VTC_ERR_ARG, "unknown stage %d", stage);
  if (n > cap) return fail(m, VTC_ERR_CAPACITY, "stage %d needs %lld floats", stage, n);
  if (!is16) {
    if (hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(m, VTC_ERR_HIP, "copy failed");
    return n;
  }
  std::vector<unsigned short> tmp(n);
  if (hipMemcpy(tmp.data(), src, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(m, VTC_ERR_HIP, "copy failed");
  for (long long i = 0; i < n; ++i) {
    if (m->f16) { _Float16 h; memcpy(&h, &tmp[i], 2); out[i] = (float)h; }
    else { unsigned u = (unsigned)tmp[i] << 16; memcpy(&out[i], &u, 4); }
  }
  return n;
}

int64_t vtc_resampled_length(int64_t n_in, int32_t sr_in) {
  if (n_in < 0 || sr_in <= 0) return -1;
  long long a = sr_in, b = 16000;
  while (b) { const long long t = a % b; a = b; b = t; }
  const long long orig = sr_in / a, nw = 16000 / a;
  return (nw * n_in + orig - 1) / orig;   // ceil(new * n / orig), torchaudio's target_length
}

// torchaudio.functional.resample's kernel (_get_sinc_resample_kernel, sinc_interp_hann, width 6, rolloff 0.99), fp64 -> fp32,
// stored compactly: per phase only the taps inside the window's support (|w| >= 1e-30; the clamped taps outside are ~1e-34).
static int build_resample(vtc_model* m, int sr_in, ResampleTable* out) {
  std::lock_guard<std::mutex> lock(m->resample_mu);
  auto it = m->resample.find(sr_in);
  if (it != m->resample.end()) { *out = it->second.t; return VTC_OK; }
  long long a = sr_in, b = 16000;
  while (b) { const long long t = a % b; a = b; b = t; }
  const int orig = (int)(sr_in / a), nw = (int)(16000 / a);
  ResampleTable t{};
  t.orig = orig; t.new_ = nw;
  if (orig != nw) {
    const double lpw = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
    const double base_freq = (double)(orig < nw ? orig : nw) * rolloff;
    const int width = (int)std::ceil(lpw * orig / base_freq);
    const int K = 2 * width + orig;
    const double scale = base_freq / orig;
    std::vector<int> first(nw), count(nw);
    std::vector<std::vector<float>> rows(nw);
    int kc = 1;
    std::vector<float> full(K);
    for (int p = 0; p < nw; ++p) {
      int lo = K, hi = -1;
      for (int k = 0; k < K; ++k) {
        double tt = (-(double)p / nw + (double)(k - width) / orig) * base_freq;
        tt = tt < -lpw ? -lpw : (tt > lpw ? lpw : tt);
        const double c = std::cos(tt * pi / lpw / 2.0);
        const double window = c * c;
        tt *= pi;
        const double kern = (tt == 0.0 ? 1.0 : std::sin(tt) / tt) * window * scale;
        full[k] = (float)kern;
        if (std::fabs(full[k]) >= 1e-30f) { if (k < lo) lo = k; hi = k; }
      }
      if (hi < lo) { lo = 0; hi = 0; }
      first[p] = lo; count[p] = hi - lo + 1;
      rows[p].assign(full.begin() + lo, full.begin() + hi + 1);
      if (count[p] > kc) kc = count[p];
    }
    std::vector<float> w((size_t)nw * kc, 0.f);
    for (int p = 0; p < nw; ++p) memcpy(&w[(size_t)p * kc], rows[p].data(), rows[p].size() * sizeof(float));
    float* dw = nullptr; int* df = nullptr;
    HIP_TRY(m, hipMalloc((void**)&dw, w.size() * sizeof(float)));
    m->allocs.push_back(dw);
    HIP_TRY(m, hipMalloc((void**)&df, first.size() * sizeof(int)));
    m->allocs.push_back(df);
    HIP_TRY(m, hipMemcpy(dw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(m, hipMemcpy(df, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
    t.width = width; t.kc = kc; t.w = dw; t.first = df;
  }
  m->resample[sr_in] = vtc_model::ResampleDev{t};
  *out = t;
  return VTC_OK;
}

int vtc_prepare_audio(vtc_model* m, const void* pcm, int32_t format, int64_t n_in, int32_t n_channels, int32_t sr_in,
                      float* out, int64_t out_cap, int64_t* n_out, void* stream) {
  if (!m) return VTC_ERR_ARG;
  if (!pcm || !out || !n_out) return fail(m, VTC_ERR_ARG, "null argument");
  if (format < 0 || format > 2) return fail(m, VTC_ERR_ARG, "format must be 0 (int16), 1 (float32) or 2 (packed int24)");
  if (n_channels < 1 || n_channels > 64) return fail(m, VTC_ERR_ARG, "n_channels %d out of range 1..64", n_channels);
  if (sr_in < 1000 || sr_in > 768000) return fail(m, VTC_ERR_ARG, "sample rate %d out of range 1000..768000", sr_in);
  if (n_in < 0) return fail(m, VTC_ERR_ARG, "negative length");
  const int64_t need = vtc_resampled_length(n_in, sr_in);
  *n_out = need;
  if (need > out_cap) return fail(m, VTC_ERR_CAPACITY, "output needs %lld samples, capacity %lld", (long long)need, (long long)out_cap);
  if (need == 0) return VTC_OK;
  HIP_TRY(m, hipSetDevice(m->device));
  ResampleTable t{};
  const int rc = build_resample(m, sr_in, &t);
  if (rc != VTC_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long long blocks = (need + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(m, VTC_ERR_ARG, "input too long");
  const double bytes = (double)n_in * n_channels * (format == 0 ? 2 : format == 1 ? 4 : 3) + (double)need * 4;
  if (format == 0)
    LAUNCH(m, PC_OTHER, 0.0, bytes, s, hipLaunchKernelGGL((prepare_audio_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, s, pcm,
                                                         (long long)n_in, n_channels, t, out, (long long)need));
  else if (format == 1)
    LAUNCH(m, PC_OTHER, 0.0, bytes, s, hipLaunchKernelGGL((prepare_audio_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, pcm,
                                                         (long long)n_in, n_channels, t, out, (long long)need));
  else
    LAUNCH(m, PC_OTHER, 0.0, bytes, s, hipLaunchKernelGGL((prepare_audio_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, pcm,
                                                         (long long)n_in, n_channels, t, out, (long long)need));
  HIP_TRY(m, hipGetLastError());
  return VTC_OK;
}

// ---- host-side FLAC front-end (no device work, no handle; errors go to vtc_last_error(NULL) of the calling thread)
int vtc_flac_info(const void* data, int64_t n_bytes, int32_t* sample_rate, int32_t* n_channels, int32_t* bits_per_sample,
                  int64_t* n_samples, uint8_t* md5) {
  if (!data || n_bytes < 0) return fail(nullptr, VTC_ERR_ARG, "null argument");
  flac::StreamInfo si;
  flac::Error err;
  if (!flac::parse_header((const uint8_t*)data, n_bytes, si, err)) return fail(nullptr, VTC_ERR_ARG, "%s", err.msg.c_str());
  if (sample_rate) *sample_rate = si.sample_rate;
  if (n_channels) *n_channels = si.channels;
  if (bits_per_sample) *bits_per_sample = si.bps;
  if (n_samples) *n_samples = si.total;
  if (md5) memcpy(md5, si.md5, 16);
  return VTC_OK;
}

int vtc_flac_decode(const void* data, int64_t n_bytes, void* out, int32_t out_format, int64_t out_cap, int64_t* n_out) {
  return vtc_flac_decode_threads(data, n_bytes, out, out_format, out_cap, n_out, 1);
}

int vtc_flac_decode_threads(const void* data, int64_t n_bytes, void* out, int32_t out_format, int64_t out_cap, int64_t* n_out,
                            int32_t n_threads) {
  if (!data || n_bytes < 0 || !n_out) return fail(nullptr, VTC_ERR_ARG, "null argument");
  if (out_format < 0 || out_format > 2) return fail(nullptr, VTC_ERR_ARG, "out_format must be 0 (int16), 1 (float32) or 2 (int32)");
  flac::StreamInfo si;
  flac::Error err;
  const int64_t got = n_threads > 1 ? flac::decode_parallel((const uint8_t*)data, n_bytes, out, out_format, out_cap, n_threads, si, err)
                                    : flac::decode((const uint8_t*)data, n_bytes, out, out_format, out_cap, si, err);
  if (got == -2) return fail(nullptr, VTC_ERR_CAPACITY, "%s", err.msg.c_str());
  if (got < 0) return fail(nullptr, VTC_ERR_ARG, "%s", err.msg.c_str());
  *n_out = got;
  return VTC_OK;
}

int vtc_ima_adpcm_decode(const void* data, int64_t n_bytes, int32_t n_channels, int32_t block_align, int32_t samples_per_block,
                         int16_t* out, int64_t out_cap, int64_t* n_out) {
  if (!data || n_bytes < 0 || !out || out_cap < 0 || !n_out) return fail(nullptr, VTC_ERR_ARG, "null argument");
  const char* why = "";
  const int64_t got = adpcm::decode_ima((const uint8_t*)data, n_bytes, n_channels, block_align, samples_per_block, out, out_cap, &why);
  if (got < 0) return fail(nullptr, VTC_ERR_ARG, "%s", why);
  *n_out = got;
  return VTC_OK;
}

int vtc_profile_enable(vtc_model* m, int32_t on) {
  if (!m) return VTC_ERR_ARG;
  m->prof_on = on != 0;   // records are kept across pause/resume; vtc_profile_read drains them
  return VTC_OK;
}

int vtc_profile_read(vtc_model* m, double* ms, int64_t* launches, double* flops, double* bytes) {
  if (!m || !ms || !launches || !flops || !bytes) return VTC_ERR_ARG;
  HIP_TRY(m, hipSetDevice(m->device));
  HIP_TRY(m, hipDeviceSynchronize());
  for (int c = 0; c < VTC_PROF_CLASSES; ++c) { ms[c] = 0; launches[c] = 0; flops[c] = 0; bytes[c] = 0; }
  for (const ProfRec& r : m->prof) {
    float t = 0.f;
    HIP_TRY(m, hipEventElapsedTime(&t, r.e0, r.e1));
    ms[r.cls] += t; launches[r.cls] += 1; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes;
  }
  m->prof.clear();
  m->ev_next = 0;
  return VTC_OK;
}

}  // extern "C"
