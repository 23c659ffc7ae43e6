// runtime.cpp - context, error string, workspace arena, staging.  No CPU fallback anywhere: if HIP
// reports no device every compute entry point fails with TSD_E_HIP.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void tsd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// the ONLY place the library reads its tuning switches from the environment (one pass per context, at tsd_ctx_create)
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }
void options_from_env(TsdOptions& o) {
  o.qkv_fuse = env_int("TSD_QKV_FUSE", o.qkv_fuse);
  o.res_fuse_skip = env_int("TSD_RES_FUSE_SKIP", o.res_fuse_skip);
  o.gn_composite = env_int("TSD_GN_COMPOSITE", o.gn_composite);
  o.conv_in_im2col = env_int("TSD_CONV_IN_IM2COL", o.conv_in_im2col);
  o.chain = env_int("TSD_CHAIN", o.chain) ? 1 : 0;
  o.fold_out = env_int("TSD_FOLD_OUT", o.fold_out) ? 1 : 0;
  o.conv_w_tm_mib = env_int("TSD_CONV_W_TM", o.conv_w_tm_mib);
  o.lin_w_tm = env_int("TSD_LIN_W_TM", o.lin_w_tm);
  o.lin_w_tm_kib = env_int("TSD_LIN_W_TM_KIB", o.lin_w_tm_kib);
  { const int q = env_int("TSD_ATTN_QB", o.attn_qb); if (q == 1 || q == 2) o.attn_qb = q; }
  o.attn_xcd = env_int("TSD_ATTN_XCD", o.attn_xcd);
  o.attn_wg8 = env_int("TSD_ATTN_WG8", o.attn_wg8) ? 1 : 0;
  o.attn8_var = env_int("TSD_ATTN8_VAR", o.attn8_var);
  o.xcdn = env_int("TSD_GEMM_XCDN", o.xcdn);
  o.conv_halo = env_int("TSD_CONV_HALO", o.conv_halo);
  o.splitk = env_int("TSD_GEMM_SPLITK", o.splitk);
  o.splitk_mink = env_int("TSD_GEMM_SPLITK_MINK", o.splitk_mink);
  o.splitk_tiles = env_int("TSD_GEMM_SPLITK_TILES", o.splitk_tiles);
  o.splitk_small = env_int("TSD_GEMM_SPLITK_SMALL", o.splitk_small);
  o.splitk_wide = env_int("TSD_GEMM_SPLITK_WIDE", o.splitk_wide);
  o.splitk_ring4 = env_int("TSD_GEMM_SPLITK_RING4", o.splitk_ring4);
  o.splitk_big = env_int("TSD_GEMM_SPLITK_BIG", o.splitk_big);
  o.sk_cfg = env_int("TSD_GEMM_SK_CFG", o.sk_cfg);
  o.sk256 = env_int("TSD_GEMM_SK256", o.sk256);
  o.thin_cfg = env_int("TSD_GEMM_THIN_CFG", o.thin_cfg);
  o.skip128 = env_int("TSD_GEMM_SKIP128", o.skip128);
  o.tune = env_int("TSD_GEMM_TUNE", o.tune);
  if (const char* ov = getenv("TSD_GEMM_CFG_OVERRIDE")) { strncpy(o.cfg_override, ov, sizeof(o.cfg_override) - 1); o.cfg_override[sizeof(o.cfg_override) - 1] = 0; }
  o.gn_apply_mult = env_int("TSD_GN_APPLY_MULT", o.gn_apply_mult);
  o.gn_finalize_min = env_int("TSD_GN_FINALIZE_MIN", o.gn_finalize_min);
  o.debug_occ = getenv("TSD_DEBUG_OCC") ? 1 : 0;
  o.debug_poison = env_int("TSD_DEBUG_POISON", o.debug_poison);
  o.debug_poison_what = env_int("TSD_DEBUG_POISON_WHAT", o.debug_poison_what);
  o.bench_wrot = env_int("TSD_BENCH_WROT", o.bench_wrot); if (o.bench_wrot < 1) o.bench_wrot = 1;
  o.bench_epi = env_int("TSD_BENCH_EPI", o.bench_epi);
  o.bench_altcfg = env_int("TSD_BENCH_ALTCFG", o.bench_altcfg);
  o.gemm_ts = getenv("TSD_GEMM_TS") ? 1 : 0;
}

extern "C" int tsd_version(void) { return TSD_VERSION; }
extern "C" const char* tsd_last_error(void) { return g_err; }

extern "C" int tsd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int tsd_ctx_create(int device, tsd_ctx** out) {
  if (!out) TSD_FAIL(TSD_E_ARG, "tsd_ctx_create: out is NULL");
  *out = nullptr;
  // kernel arguments in device memory (see tsd/_lib.py): effective when this is the process's first HIP call; hosts that
  // initialise HIP earlier export HIP_FORCE_DEV_KERNARG=1 themselves
  setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    TSD_FAIL(TSD_E_HIP, "tsd_ctx_create: no HIP device visible (libtsd has no CPU fallback)");
  if (device < 0 || device >= n) TSD_FAIL(TSD_E_ARG, "tsd_ctx_create: device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    TSD_FAIL(TSD_E_HIP, "tsd_ctx_create: device %d is %s; libtsd is built for gfx950 only", device, prop.gcnArchName);
  tsd_ctx* c = new tsd_ctx();
  c->device = device;
  options_from_env(c->opt);
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipMalloc((void**)&c->status, 16 * sizeof(int)));
  HIP_TRY(hipMemsetAsync(c->status, 0, 16 * sizeof(int), c->stream));
  HIP_TRY(hipMalloc((void**)&c->zeros, 4096));
  HIP_TRY(hipMemsetAsync(c->zeros, 0, 4096, c->stream));
  HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)(c->zeros + 1024), 0x3C00, 64, c->stream));  // 64 halves of 1.0 (attention row sums)
  HIP_TRY(hipStreamSynchronize(c->stream));
  *out = c;
  return TSD_OK;
}

extern "C" int tsd_ctx_destroy(tsd_ctx* c) {
  if (!c) return TSD_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->arena.base) hipFree(c->arena.base);
  if (c->staging) hipFree(c->staging);
  if (c->zeros) hipFree(c->zeros);
  if (c->status) hipFree(c->status);
  if (c->sk_flags) hipFree(c->sk_flags);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  hipStreamDestroy(c->stream);
  delete c;
  return TSD_OK;
}

int ctx_check_splitk(tsd_ctx* c) {
  if (!c->sk_flags) return TSD_OK;
  int n = 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(&n, c->sk_flags + 4095, sizeof(int), hipMemcpyDeviceToHost));
  if (n != 0) {
    // reported once, then cleared: the caller re-runs what it computed since the last clean synchronisation point
    const int zero = 0;
    (void)hipMemcpy(c->sk_flags + 4095, &zero, sizeof(int), hipMemcpyHostToDevice);
    TSD_FAIL(TSD_E_STATE, "%d split-K hand-off(s) timed out on this context (GPU shared or preempted?): results computed "
             "since the last clean synchronisation are invalid; re-run them", n);
  }
  return TSD_OK;
}

// call after a stream synchronize
int ctx_check_status(tsd_ctx* c) {
  TSD_TRY(ctx_check_splitk(c));
  if (!c->status) return TSD_OK;
  int n = 0;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(&n, c->status, sizeof(int), hipMemcpyDeviceToHost));
  if (n != 0) {
    const int zero = 0;  // reported once, then cleared
    (void)hipMemcpy(c->status, &zero, sizeof(int), hipMemcpyHostToDevice);
    TSD_FAIL(TSD_E_NONFINITE, "%d non-finite value(s) (inf / NaN) reached a tensor that leaves the device since the last clean "
             "synchronisation: an fp16 activation overflowed (|x| > 65504) or an input was not finite; the outputs are not "
             "the reference's (BASELINE.md section 4: supported dynamic range)", n);
  }
  return TSD_OK;
}
// Non-finite values written to caller-visible tensors on this context since the last report / reset (synchronises the stream).
extern "C" int tsd_debug_nonfinite_count(tsd_ctx* c, int reset) {
  if (!c || !c->status) return -1;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  int n = 0;
  if (hipMemcpy(&n, c->status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset) { const int zero = 0; if (hipMemcpy(c->status, &zero, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1; }
  return n;
}

extern "C" int tsd_ctx_synchronize(tsd_ctx* c) {
  if (!c) TSD_FAIL(TSD_E_ARG, "tsd_ctx_synchronize: ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ctx_check_status(c);
}

extern "C" int tsd_ctx_timer_start(tsd_ctx* c) {
  if (!c) TSD_FAIL(TSD_E_ARG, "ctx is NULL");
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return TSD_OK;
}
extern "C" int tsd_ctx_timer_stop(tsd_ctx* c, float* ms) {
  if (!c || !ms) TSD_FAIL(TSD_E_ARG, "ctx/ms is NULL");
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return TSD_OK;
}

extern "C" int tsd_ctx_profile_begin(tsd_ctx* c) {
  if (!c) TSD_FAIL(TSD_E_ARG, "ctx is NULL");
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->profile = true;
  c->prof_n = 0;
  return TSD_OK;
}
extern "C" int tsd_ctx_profile_end(tsd_ctx* c, float* ms_per_class, int* launches_per_class, int nclass) {
  if (!c || !ms_per_class || !launches_per_class) TSD_FAIL(TSD_E_ARG, "profile_end: NULL argument");
  c->profile = false;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < nclass; i++) { ms_per_class[i] = 0.f; launches_per_class[i] = 0; }
  for (size_t i = 0; i < c->prof_n; i++) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    const int k = c->prof_cls[i];
    if (k >= 0 && k < nclass) { ms_per_class[k] += ms; launches_per_class[k] += i < c->prof_kern.size() ? c->prof_kern[i] : 1; }
  }
  c->prof_n = 0;
  return TSD_OK;
}
// per-launch records of the last profiling pass: rec[i] = {class, M, N, K, batch}, ms[i]; returns the count
extern "C" int tsd_ctx_profile_records(tsd_ctx* c, int* rec, float* ms, int cap) {
  if (!c || !rec || !ms) TSD_FAIL(TSD_E_ARG, "profile_records: NULL argument");
  c->profile = false;
  HIP_TRY(hipStreamSynchronize(c->stream));
  int n = 0;
  for (size_t i = 0; i < c->prof_n && n < cap; i++, n++) {
    HIP_TRY(hipEventElapsedTime(&ms[n], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    rec[5 * n] = c->prof_cls[i];
    for (int j = 0; j < 4; j++) rec[5 * n + 1 + j] = c->prof_shape[4 * i + j];
  }
  return n;
}

int ctx_reserve_arena(tsd_ctx* c, size_t bytes) {
  if (c->arena.cap >= bytes) return TSD_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->arena.base) HIP_TRY(hipFree(c->arena.base));
  c->arena.base = nullptr;
  c->arena.cap = 0;
  const size_t want = bytes + (bytes >> 4) + (1 << 20);
  hipError_t e = hipMalloc((void**)&c->arena.base, want);
  if (e != hipSuccess) TSD_FAIL(TSD_E_ALLOC, "workspace arena: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  c->arena.cap = want;
  if (c->opt.debug_poison >= 0 && (c->opt.debug_poison_what & 1)) HIP_TRY(hipMemsetAsync(c->arena.base, c->opt.debug_poison & 255, want, c->stream));  // on the context's stream: ordered before everything that uses the arena
  return TSD_OK;
}

int ctx_reserve_staging(tsd_ctx* c, size_t bytes) {
  if (c->staging_cap >= bytes) return TSD_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->staging) HIP_TRY(hipFree(c->staging));
  c->staging = nullptr;
  c->staging_cap = 0;
  hipError_t e = hipMalloc(&c->staging, bytes);
  if (e != hipSuccess) TSD_FAIL(TSD_E_ALLOC, "staging: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  c->staging_cap = bytes;
  return TSD_OK;
}
