// C-ABI entry points that are not tied to one kernel family: lifecycle, posterior dispatch,
// MFMA layout self-test, instrumentation.  See include/baybe_hip.h for the contract.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bbh_common.h"

static const char* kNoHandle = "bbh: null handle";

extern "C" int bbh_version(void) { return 100; }

extern "C" int bbh_create(int device_id, bbh_handle** out) {
  if (!out) return -1;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return -10;  // no HIP device: fail loudly
  if (device_id < 0 || device_id >= count) return -11;
  if (hipSetDevice(device_id) != hipSuccess) return -12;
  bbh_handle* h = new bbh_handle();
  h->device = device_id;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) h->num_cu = cus;
  }
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && v > 0) h->lds_per_block = (size_t)v;
  }
  if (const char* e = getenv("BBH_PENDING_LDS")) h->pending_lds_form = (e[0] != '0');
  if (const char* e = getenv("BBH_FIT_GRAPH")) h->fit_graph_mode = (e[0] != '0');
  if (const char* e = getenv("BBH_FIT_SMALL")) h->fit_small = (e[0] != '0');
  if (const char* e = getenv("BBH_FIT_FLOW")) h->fit_flow = atoi(e);
  if (const char* e = getenv("BBH_FLOW_SPIN")) h->flow_spin_limit = atoi(e);
  if (const char* e = getenv("BBH_TILE_GRAM")) h->tile_gram = (e[0] != '0');
  if (const char* e = getenv("BBH_TILE_WT")) h->tile_wt = (e[0] != '0');
  if (const char* e = getenv("BBH_TILE_ACQ")) h->tile_d_sc1 = (e[0] == '0');  // (default: acquire fence)
  if (const char* e = getenv("BBH_TILE_MT")) {
    h->tile_mt = (e[0] != '0');
    h->tile_mt_partial = (e[0] == 'p');
  }
  if (const char* e = getenv("BBH_POTRF_TILES")) h->potrf_tiles = (e[0] != '0');
  if (const char* e = getenv("BBH_TILE_SPIN")) h->tile_spin_limit = atoi(e);
  if (const char* e = getenv("BBH_KV_GLOBAL")) h->kv_global_mode = (e[0] != '0') ? 1 : 0;
  if (const char* e = getenv("BBH_KV_LDS")) h->kv_lds_blocks = atoi(e);
  if (const char* e = getenv("BBH_MEAN_VALU")) h->use_mean_valu = (e[0] != '0');
  if (const char* e = getenv("BBH_COOPG_CROSS")) h->coopg_cross_on = (e[0] != '0');
  if (const char* e = getenv("BBH_KVCACHE")) h->use_kvcache = (e[0] != '0');
  if (const char* e = getenv("BBH_PIPELINE")) h->use_pipeline = (e[0] != '0');  // A/B switch, default on
  if (const char* e = getenv("BBH_COOP")) h->coop_mode = atoi(e);
  if (const char* e = getenv("BBH_POTRF_REG")) h->potrf_register_form = (e[0] != '0');
  if (const char* e = getenv("BBH_FIT_OVERLAP")) h->fit_overlap = (e[0] != '0');
  if (const char* e = getenv("BBH_Q1_SLICED")) h->q1_sliced = (e[0] != '0');
  if (const char* e = getenv("BBH_SELECT")) h->select_on = (e[0] != '0');
  if (const char* e = getenv("BBH_SMALL")) h->small_on = (e[0] != '0');
  *out = h;
  return 0;
}

extern "C" int bbh_destroy(bbh_handle* h) {
  if (!h) return -1;
  hipSetDevice(h->device);
  if (h->stream)
    hipStreamSynchronize(h->stream);
  else
    hipDeviceSynchronize();
  for (auto& sp : h->pending_events) {
    hipEventDestroy(sp.e0);
    hipEventDestroy(sp.e1);
  }
  bbh_comm_destroy(h);
  bbh_select_destroy(h);
  bbh_nehvi_destroy(h);
  bbh_sobol_destroy(h);
  bbh_flow_destroy(h);
  bbh_free_model_public(h);
  if (h->d_ws) hipFree(h->d_ws);
  if (h->fit_stream) {
    hipStreamSynchronize(h->fit_stream);
    hipStreamDestroy(h->fit_stream);
  }
  if (h->side_stream) {
    hipStreamSynchronize(h->side_stream);
    hipStreamDestroy(h->side_stream);
  }
  for (auto& e : h->side_events)
    if (e) hipEventDestroy(e);
  if (h->d_rstream) hipFree(h->d_rstream);
  if (h->d_rsmall) hipFree(h->d_rsmall);
  if (h->d_tileflags) hipFree(h->d_tileflags);
  if (h->d_kvcache) hipFree(h->d_kvcache);
  if (h->d_slab_flags) hipFree(h->d_slab_flags);
  if (h->d_z) hipFree(h->d_z);
  if (h->h_zstage) hipHostFree(h->h_zstage);
  if (h->z_evt) hipEventDestroy(h->z_evt);
  if (h->d_red) hipFree(h->d_red);
  if (h->d_redi) hipFree(h->d_redi);
  delete h;
  return 0;
}

extern "C" const char* bbh_last_error(bbh_handle* h) { return h ? h->err.c_str() : kNoHandle; }

extern "C" int bbh_set_stream(bbh_handle* h, void* hip_stream) {
  if (!h) return -1;
  h->stream = (hipStream_t)hip_stream;
  return 0;
}

// ---- MFMA fragment-layout self-test ---------------------------------------------------------
// C = A B with asymmetric integer-valued A (16x8) and B (8x16): catches row/col swaps and the
// f32-style C mapping (which misplaces 3 of every 4 rows on the f64 instruction).
__global__ void bbh_selftest_kernel(const double* A, const double* B, double* C) {
  const int l = threadIdx.x;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < 2; ks++) {
    const double a = A[(l & 15) * 8 + 4 * ks + (l >> 4)];
    const double b = B[(4 * ks + (l >> 4)) * 16 + (l & 15)];
    acc = mfma_f64(a, b, acc);
  }
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

extern "C" int bbh_selftest(bbh_handle* h) {
  if (!h) return -1;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  double hA[16 * 8], hB[8 * 16], hC[256], ref[256];
  for (int i = 0; i < 16; i++)
    for (int k = 0; k < 8; k++) hA[i * 8 + k] = (double)(3 * i + 7 * k + 1) - 0.5 * (double)(i * k);
  for (int k = 0; k < 8; k++)
    for (int j = 0; j < 16; j++) hB[k * 16 + j] = (double)(5 * k - 2 * j) + 0.25 * (double)(k * j + j);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      double s = 0.0;
      for (int k = 0; k < 8; k++) s += hA[i * 8 + k] * hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  int rc = bbh_ensure_ws(h, sizeof(double) * 1024);
  if (rc) return rc;
  double* dA = h->d_ws;
  double* dB = dA + 128;
  double* dC = dB + 128;
  BBH_HIP_TRY(h, hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, h->stream));
  BBH_HIP_TRY(h, hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(bbh_selftest_kernel, dim3(1), dim3(64), 0, h->stream, dA, dB, dC);
  BBH_HIP_TRY(h, hipMemcpyAsync(hC, dC, sizeof(hC), hipMemcpyDeviceToHost, h->stream));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int e = 0; e < 256; e++)
    if (fabs(hC[e] - ref[e]) > 1e-9 * (1.0 + fabs(ref[e]))) {
      h->err = "bbh_selftest: v_mfma_f64_16x16x4_f64 fragment layout mismatch at element " + std::to_string(e);
      return -20;
    }
  return 0;
}

// ---- posterior ------------------------------------------------------------------------------
static int check_post_args(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx) {
  if (!h->factorized) {
    h->err = "posterior requested before bbh_factorize";
    return -1;
  }
  if (N < 0 || (N > 0 && !X_dev) || ldx < h->desc.d) {
    h->err = "posterior: bad arguments (need ldx >= d)";
    return -1;
  }
  if (h->dn > 126) {
    h->err = "fused posterior supports at most 126 numerical columns";
    return -1;
  }
  return 0;
}

extern "C" int bbh_posterior(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev,
                             double* var_dev) {
  if (!h) return -1;
  int rc = check_post_args(h, X_dev, N, ldx);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  return bbh_launch_fused(h, X_dev, N, ldx, mean_dev, var_dev, nullptr, true);
}

extern "C" int bbh_posterior_unfused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev,
                                     double* var_dev) {
  if (!h) return -1;
  int rc = check_post_args(h, X_dev, N, ldx);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  return bbh_launch_unfused(h, X_dev, N, ldx, mean_dev, var_dev);
}

extern "C" int bbh_cross_cov(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* cross_dev) {
  if (!h) return -1;
  int rc = check_post_args(h, X_dev, N, ldx);
  if (rc) return rc;
  if (h->p < 1 || !cross_dev) {
    h->err = "bbh_cross_cov: no pending points set";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  return bbh_launch_fused(h, X_dev, N, ldx, nullptr, nullptr, cross_dev, false);
}

extern "C" int bbh_train_posterior_mean(bbh_handle* h, double* mean_host) {
  if (!h) return -1;
  if (!h->factorized || !mean_host) {
    h->err = "bbh_train_posterior_mean: model not factorised";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const int64_t n = h->n, d = h->desc.d;
  int rc = 0;
  double* own = nullptr;  // composite kernels: the posterior path uses the workspace itself
  if (bbh_materialised_only(h))
    BBH_HIP_TRY(h, hipMalloc((void**)&own, sizeof(double) * (size_t)(n * d + n)));
  else if ((rc = bbh_ensure_ws(h, sizeof(double) * (size_t)(n * d + n))))
    return rc;
  double* dX = own ? own : h->d_ws;
  double* dm = dX + n * d;
  hipError_t e = hipMemcpyAsync(dX, h->xraw_host.data(), sizeof(double) * n * d, hipMemcpyHostToDevice, h->stream);
  // pending columns do not affect column 0 (the mean), but keep the call side-effect free
  if (e == hipSuccess) rc = bbh_launch_fused(h, dX, n, d, dm, nullptr, nullptr, false);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(mean_host, dm, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (own) hipFree(own);
  if (rc) return rc;
  BBH_HIP_TRY(h, e);
  return 0;
}

extern "C" int bbh_set_slice_rows(bbh_handle* h, int64_t rows) {
  if (!h || rows < 0) return -1;
  h->slice_rows = rows;
  return 0;
}

// Idle handles are kept in a pool by the host side (engine.py); their grow-only buffers would pin HBM for as long as they sit there.
// Everything above keep_bytes goes: the scratch workspace, the global kernel-value cache, and the model (its buffers are re-created by
// the next bbh_set_model) when its matrices exceed the bound.  Small campaign models - what the pool is for - keep everything.
extern "C" int bbh_trim(bbh_handle* h, int64_t keep_bytes) {
  if (!h || keep_bytes < 0) return -1;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->d_ws && h->ws_bytes > (size_t)keep_bytes) {
    hipFree(h->d_ws);
    h->d_ws = nullptr;
    h->ws_bytes = 0;
  }
  if (h->d_kvcache && h->kvcache_bytes > (size_t)keep_bytes) {
    hipFree(h->d_kvcache);
    h->d_kvcache = nullptr;
    h->kvcache_bytes = 0;
  }
  if (h->have_model && sizeof(double) * (size_t)h->np * (size_t)h->np * 6 > (size_t)keep_bytes) bbh_free_model_public(h);
  return 0;
}

// ---- instrumentation ------------------------------------------------------------------------
extern "C" int bbh_last_posterior_form(bbh_handle* h) { return h ? h->last_form : -1; }

extern "C" int bbh_timing_enable(bbh_handle* h, int enable) {
  if (!h) return -1;
  h->timing = enable < 0 ? 0 : enable;  // 1: all families; 2 * mask: the families whose bit is set (events between kernels cost ~5 us each)
  return 0;
}

extern "C" int bbh_timing_read_family(bbh_handle* h, int32_t family, double* ms_total, int64_t* launches, int reset) {
  if (!h) return -1;
  if (family < 0 || family >= BBH_TIMED_FAMILIES) {
    h->err = "bbh_timing_read_family: unknown kernel family";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  for (auto& sp : h->pending_events) {
    BBH_HIP_TRY(h, hipEventSynchronize(sp.e1));
    float ms = 0.f;
    BBH_HIP_TRY(h, hipEventElapsedTime(&ms, sp.e0, sp.e1));
    h->timed_ms[sp.family] += (double)ms;
    h->timed_launches[sp.family] += 1;
    hipEventDestroy(sp.e0);
    hipEventDestroy(sp.e1);
  }
  h->pending_events.clear();
  if (ms_total) *ms_total = h->timed_ms[family];
  if (launches) *launches = h->timed_launches[family];
  if (reset) {
    h->timed_ms[family] = 0.0;
    h->timed_launches[family] = 0;
  }
  return 0;
}

extern "C" int bbh_timing_read(bbh_handle* h, double* fused_ms_total, int64_t* fused_launches, int reset) {
  return bbh_timing_read_family(h, BBH_TIMED_POSTERIOR, fused_ms_total, fused_launches, reset);
}
