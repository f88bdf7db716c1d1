// Row-sharded selection over the GPUs of one node, inside the library.
//
// The reference has no call site for this (BayBE is single-process); the exchange belongs to the step
// that replaces botorch.optim.optimize_acqf_discrete's argmax (baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126)
// once the candidate rows are sharded: every rank scores its row range, and ONE all-gather per selection step of the
// per-shard winners - k x (score, global row index) for a top-k, or (score, index, row[d]) for a greedy step - makes the
// pick global.  First-index tie-break on the global index, which follows shard order, so the result is what a single
// device would have selected.  The payload is built by one kernel on the device, gathered with ncclAllGather (RCCL over
// xGMI) on the handle's stream, and read back with one copy.
//
// RCCL is bound at run time (dlopen by SONAME): the process has usually loaded PyTorch's copy already, and a second HIP /
// RCCL runtime in one process must be avoided (see baybe_amd/_lib.py); without a communicator the rest of the library
// does not depend on RCCL at all.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "bbh_common.h"

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    }
  }
  const bool ok = api.lib && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
  if (!ok && why) *why = "RCCL (librccl.so.1) could not be loaded";
  return ok ? &api : nullptr;
}

struct CommState {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  double* d_send = nullptr;  // [cap] payload of this rank
  double* d_recv = nullptr;  // [world * cap]
  size_t cap = 0;
};

CommState* state_of(bbh_handle* h) { return (CommState*)h->comm_state; }

int ensure_buffers(bbh_handle* h, CommState* st, size_t doubles) {
  if (doubles <= st->cap) return 0;
  if (st->d_send) hipFree(st->d_send);
  if (st->d_recv) hipFree(st->d_recv);
  st->d_send = st->d_recv = nullptr;
  st->cap = 0;
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_send, sizeof(double) * doubles));
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_recv, sizeof(double) * doubles * st->world));
  st->cap = doubles;
  return 0;
}

// payload[j] = (score, global index) of the j-th local best; entries beyond the shard: (-inf, -1)
__global__ void bbh_pack_topk_kernel(const double* __restrict__ vals, const int64_t* __restrict__ idx, int64_t have, int64_t k,
                                     int64_t row_offset, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const bool real = j < have && idx[j] >= 0;
  out[2 * j] = real ? vals[j] : -INFINITY;
  out[2 * j + 1] = real ? (double)(idx[j] + row_offset) : -1.0;
}

// payload = (score, global index, row[d]) of the local winner (index < 0: this shard has no candidate left)
__global__ void bbh_pack_winner_kernel(const double* __restrict__ val, const int64_t* __restrict__ idx, int64_t row_offset,
                                       const double* __restrict__ X, int64_t ldx, int d, double* __restrict__ out) {
  const int64_t li = idx[0];
  const int t = threadIdx.x;
  if (t == 0) {
    out[0] = li >= 0 ? val[0] : -INFINITY;
    out[1] = li >= 0 ? (double)(li + row_offset) : -1.0;
  }
  for (int c = t; c < d; c += blockDim.x) out[2 + c] = li >= 0 ? X[li * ldx + c] : 0.0;
}
}  // namespace

int bbh_topk_device(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double** vals_dev, int64_t** idx_dev);

extern "C" int bbh_comm_unique_id(void* id_out, int64_t bytes) {
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api || !id_out || bytes < (int64_t)sizeof(ncclUniqueId)) return -1;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return -2;
  memcpy(id_out, &id, sizeof(id));
  return (int)sizeof(ncclUniqueId);
}

extern "C" int bbh_comm_init(bbh_handle* h, int32_t rank, int32_t world, const void* unique_id, int64_t bytes) {
  if (!h) return -1;
  RcclApi* api = rccl_api(&h->err);
  if (!api) return -7;
  if (rank < 0 || world < 1 || rank >= world || !unique_id || bytes < (int64_t)sizeof(ncclUniqueId)) {
    h->err = "bbh_comm_init: bad rank / world / unique id";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (h->comm_state) bbh_comm_destroy(h);
  CommState* st = new CommState();
  st->rank = rank;
  st->world = world;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  const ncclResult_t rc = api->CommInitRank(&st->comm, world, id, rank);
  if (rc != ncclSuccess) {
    h->err = std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "failed");
    delete st;
    return -8;
  }
  h->comm_state = st;
  return 0;
}

extern "C" int bbh_comm_destroy(bbh_handle* h) {
  if (!h) return -1;
  CommState* st = state_of(h);
  if (!st) return 0;
  hipSetDevice(h->device);
  if (RcclApi* api = rccl_api(nullptr))
    if (st->comm) api->CommDestroy(st->comm);
  if (st->d_send) hipFree(st->d_send);
  if (st->d_recv) hipFree(st->d_recv);
  delete st;
  h->comm_state = nullptr;
  return 0;
}

// A rank whose local part failed (top-k workspace, device error) still joins the collective - the others are already inside
// it and would block for ever - with this payload: score -inf, index -2 in every slot.  Every rank sees it after the gather
// and returns the same error.  (All ranks must call with identical k / d: the payload size is part of the collective.)
__global__ void bbh_pack_failed_kernel(double* out, int64_t doubles) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < doubles) out[e] = (e & 1) ? -2.0 : -INFINITY;
}

static int gather(bbh_handle* h, CommState* st, size_t doubles, std::vector<double>& host) {
  RcclApi* api = rccl_api(&h->err);
  if (!api) return -7;
  const ncclResult_t rc = api->AllGather(st->d_send, st->d_recv, doubles, ncclDouble, st->comm, h->stream);
  if (rc != ncclSuccess) {
    h->err = std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(rc) : "failed");
    return -8;
  }
  host.resize(doubles * st->world);
  BBH_HIP_TRY(h, hipMemcpyAsync(host.data(), st->d_recv, sizeof(double) * host.size(), hipMemcpyDeviceToHost, h->stream));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int bbh_allgather_topk(bbh_handle* h, const double* scores_dev, int64_t N, int64_t row_offset, int64_t k,
                                  double* vals_host, int64_t* idx_host) {
  if (!h) return -1;
  CommState* st = state_of(h);
  if (!st) {
    h->err = "bbh_allgather_topk: call bbh_comm_init first";
    return -1;
  }
  if (k < 1 || k > 64 || N < 0 || (N > 0 && !scores_dev) || !vals_host || !idx_host) {
    h->err = "bbh_allgather_topk: bad arguments (1 <= k <= 64)";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_buffers(h, st, (size_t)2 * k);
  if (rc) return rc;
  double* dv = nullptr;
  int64_t* di = nullptr;
  const int64_t have = N < k ? N : k;
  int local_rc = 0;
  std::string local_err;
  if (have > 0) {
    local_rc = bbh_topk_device(h, scores_dev, N, have, &dv, &di);
    local_err = h->err;
  }
  if (local_rc)
    hipLaunchKernelGGL(bbh_pack_failed_kernel, dim3((unsigned)((2 * k + 255) / 256)), dim3(256), 0, h->stream, st->d_send, 2 * k);
  else
    hipLaunchKernelGGL(bbh_pack_topk_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, h->stream, dv, di, have, k, row_offset,
                       st->d_send);
  std::vector<double> host;
  rc = gather(h, st, (size_t)2 * k, host);
  if (rc) return rc;
  for (size_t e = 0; e < host.size() / 2; e++)
    if (host[2 * e + 1] == -2.0) {  // some rank failed locally: the same error on every rank
      h->err = local_rc ? "bbh_allgather_topk: local selection failed: " + local_err : "bbh_allgather_topk: the local selection of another rank failed";
      return -10;
    }
  // merge: descending score, ties -> lower global index; ranks' padding (index < 0), dead candidates (-inf) and NaN never win
  std::vector<std::pair<double, int64_t>> all;
  for (size_t e = 0; e < host.size() / 2; e++)
    if (host[2 * e + 1] >= 0.0 && host[2 * e] == host[2 * e] && host[2 * e] > -INFINITY)
      all.emplace_back(host[2 * e], (int64_t)host[2 * e + 1]);
  std::sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
  for (int64_t j = 0; j < k; j++) {
    vals_host[j] = j < (int64_t)all.size() ? all[j].first : -INFINITY;
    idx_host[j] = j < (int64_t)all.size() ? all[j].second : -1;
  }
  return 0;
}

extern "C" int bbh_allgather_argmax(bbh_handle* h, const double* scores_dev, int64_t N, int64_t row_offset, const double* X_dev,
                                    int64_t ldx, double* val_host, int64_t* gidx_host, double* row_host) {
  if (!h) return -1;
  CommState* st = state_of(h);
  if (!st || !h->have_model) {
    h->err = "bbh_allgather_argmax: call bbh_set_model and bbh_comm_init first";
    return -1;
  }
  const int d = h->desc.d;
  if (N < 0 || (N > 0 && (!scores_dev || !X_dev)) || ldx < d || !val_host || !gidx_host || !row_host) {
    h->err = "bbh_allgather_argmax: bad arguments";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_buffers(h, st, (size_t)2 + d);
  if (rc) return rc;
  double* dv = nullptr;
  int64_t* di = nullptr;
  int local_rc = 0;
  std::string local_err;
  if (N > 0) {
    local_rc = bbh_topk_device(h, scores_dev, N, 1, &dv, &di);
    local_err = h->err;
  } else {  // an empty shard still takes part in the collective
    rc = bbh_ensure_ws(h, 64);
    if (rc) return rc;
    const double ninf = -INFINITY;
    const int64_t none = -1;
    dv = h->d_ws;
    di = (int64_t*)(h->d_ws + 1);
    BBH_HIP_TRY(h, hipMemcpyAsync(dv, &ninf, sizeof(double), hipMemcpyHostToDevice, h->stream));
    BBH_HIP_TRY(h, hipMemcpyAsync(di, &none, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (local_rc)
    hipLaunchKernelGGL(bbh_pack_failed_kernel, dim3(1), dim3(256), 0, h->stream, st->d_send, (int64_t)2);
  else
    hipLaunchKernelGGL(bbh_pack_winner_kernel, dim3(1), dim3(64), 0, h->stream, dv, di, row_offset, X_dev, ldx, d, st->d_send);
  std::vector<double> host;
  rc = gather(h, st, (size_t)2 + d, host);
  if (rc) return rc;
  for (int r = 0; r < st->world; r++)
    if (host[(size_t)r * (2 + d) + 1] == -2.0) {
      h->err = local_rc ? "bbh_allgather_argmax: local selection failed: " + local_err : "bbh_allgather_argmax: the local selection of another rank failed";
      return -10;
    }
  int best = -1;
  for (int r = 0; r < st->world; r++) {
    const double v = host[(size_t)r * (2 + d)], gi = host[(size_t)r * (2 + d) + 1];
    if (gi < 0.0 || v != v || !(v > -INFINITY)) continue;  // padding, NaN, a shard whose candidates are all dead
    if (best < 0 || v > host[(size_t)best * (2 + d)] ||
        (v == host[(size_t)best * (2 + d)] && gi < host[(size_t)best * (2 + d) + 1]))
      best = r;
  }
  if (best < 0) {
    h->err = "bbh_allgather_argmax: no rank has a candidate left";
    return -9;
  }
  *val_host = host[(size_t)best * (2 + d)];
  *gidx_host = (int64_t)host[(size_t)best * (2 + d) + 1];
  memcpy(row_host, &host[(size_t)best * (2 + d) + 2], sizeof(double) * d);
  return 0;
}
