// qLogNEHVI set-up on the device (one selection step): the box decomposition of the non-dominated region, one per
// Monte-Carlo sample, from baseline objective samples that never leave the device (bbh_nehvi_samples, bbh_panel.hip).
// BoTorch: FastNondominatedPartitioning per sample on the CPU inside qLogNoisyExpectedHypervolumeImprovement, built at
// baybe/acquisition/_builder.py:319-324; reference acquisition class baybe/acquisition/acqfs.py:477-484.
//
// Same algorithm and the same visiting order as baybe_amd/box_decomposition.py and its host restatement bbh_cells.hip
// (incremental local upper bounds, Lacour, Klamroth & Fonseca 2017, Alg. 1): only copies and comparisons, so the cell lists
// are the host form's, cell for cell; the side lengths differ from exp(log(.)) of the host form by rounding only.
//
// One wavefront per sample.  The sample's points, the two generations of the bound list and a flag byte per bound live in
// LDS; a bound is m coordinates plus m point indices (its defining point per dimension; -1 - k = the reference-point dummy
// of dimension k), so that 1024 bounds of a four-objective problem are 48 KB.  Every list operation (filter, the m child
// families of a hit bound, the final cell filter) is a stable compaction by wave ballots: lane order = list order.
#include <math.h>

#include "bbh_common.h"

namespace {

constexpr int NV_MAXPTS = 512;    // baseline points per sample the device form takes (more: host form)
constexpr int NV_MAXBOUNDS = 1024;

template <int M>
struct Bound {
  double u[M];
  int z[M];
};

template <int M>
__device__ __forceinline__ double nv_zval(const double* __restrict__ pn, const double* __restrict__ nref, int idx, int j) {
  if (idx >= 0) return pn[idx * M + j];
  return (-1 - idx == j) ? nref[j] : -INFINITY;
}

__device__ __forceinline__ int nv_prefix(unsigned long long mask) {  // set bits below this lane
  return __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// Fb [S, nb, M] oriented objective samples (maximisation), ref [M].  slots [S][cap][2][M]: lower bound, side length.
template <int M>
__global__ __launch_bounds__(64) void bbh_cells_kernel(const double* __restrict__ Fb, int nb, const double* __restrict__ ref, int cap,
                                                       double* __restrict__ slots, int* __restrict__ cnt, int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double* y = (double*)s_raw;                  // [nb][M] the sample's points
  double* pn = y + (size_t)nb * M;             // [nb][M] non-dominated, unique, lexicographically ascending, NEGATED
  Bound<M>* ub[2];
  ub[0] = (Bound<M>*)(pn + (size_t)nb * M);
  ub[1] = ub[0] + cap;
  unsigned char* fl = (unsigned char*)(ub[1] + cap);  // [cap] hit | child-ok bits of the current point
  int* rank = (int*)(fl + ((cap + 7) & ~7));           // [nb] position in the sorted list, -1 = dropped
  __shared__ double nref[M];
  __shared__ int s_np;
  const int l = threadIdx.x;
  const int64_t s = blockIdx.x;
  const double* src = Fb + s * (int64_t)nb * M;
  for (int e = l; e < nb * M; e += 64) y[e] = src[e];
  if (l < M) nref[l] = -ref[l];
  if (l == 0) s_np = 0;
  __syncthreads();
  // ---- non-dominated, unique (first index wins) ----
  for (int i = l; i < nb; i += 64) {
    double yi[M];
#pragma unroll
    for (int o = 0; o < M; o++) yi[o] = y[i * M + o];
    bool drop = false;
    for (int j = 0; j < nb && !drop; j++) {
      bool ge = true, gt = false, eq = true;
#pragma unroll
      for (int o = 0; o < M; o++) {
        const double yj = y[j * M + o];
        ge = ge && (yj >= yi[o]);
        gt = gt || (yj > yi[o]);
        eq = eq && (yj == yi[o]);
      }
      drop = (ge && gt) || (eq && j < i);
    }
    rank[i] = drop ? -1 : 0;
  }
  __syncthreads();
  // ---- lexicographic rank among the kept points (np.unique(axis=0) order) ----
  for (int i = l; i < nb; i += 64) {
    if (rank[i] < 0) continue;
    double yi[M];
#pragma unroll
    for (int o = 0; o < M; o++) yi[o] = y[i * M + o];
    int r = 0;
    for (int j = 0; j < nb; j++) {
      if (rank[j] < 0 || j == i) continue;
      bool less = false, decided = false;
#pragma unroll
      for (int o = 0; o < M; o++) {
        const double yj = y[j * M + o];
        if (!decided && yj != yi[o]) {
          less = yj < yi[o];
          decided = true;
        }
      }
      r += less ? 1 : 0;
    }
#pragma unroll
    for (int o = 0; o < M; o++) pn[r * M + o] = -yi[o];
    atomicAdd(&s_np, 1);
  }
  __syncthreads();
  const int np_ = s_np;
  // ---- the bound list starts with the reference point ----
  int cur = 0, nU = 1;
  if (l == 0) {
#pragma unroll
    for (int o = 0; o < M; o++) {
      ub[0][0].u[o] = nref[o];
      ub[0][0].z[o] = -1 - o;
    }
  }
  __syncthreads();
  bool over = false;
  for (int pi = 0; pi < np_ && !over; pi++) {
    double p[M];
    bool above = true;
#pragma unroll
    for (int o = 0; o < M; o++) {
      p[o] = pn[pi * M + o];
      above = above && (p[o] < nref[o]);  // y > ref
    }
    if (!above) continue;  // (wave-uniform)
    const Bound<M>* U = ub[cur];
    Bound<M>* V = ub[cur ^ 1];
    int n_keep = 0, n_child[M];
#pragma unroll
    for (int j = 0; j < M; j++) n_child[j] = 0;
    for (int base = 0; base < nU; base += 64) {
      const int u = base + l;
      const bool valid = u < nU;
      bool hit = valid;
      unsigned f = 0;
      if (valid) {
#pragma unroll
        for (int o = 0; o < M; o++) hit = hit && (p[o] < U[u].u[o]);
        if (hit) {
          f = 1u;
#pragma unroll
          for (int j = 0; j < M; j++) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < M; k++)
              if (k != j) ok = ok && (nv_zval<M>(pn, nref, U[u].z[k], j) < p[j]);
            f |= ok ? (2u << j) : 0u;
          }
        }
        fl[u] = (unsigned char)f;
      }
      n_keep += __popcll(__ballot(valid && !hit));
#pragma unroll
      for (int j = 0; j < M; j++) n_child[j] += __popcll(__ballot((f >> (1 + j)) & 1u));
    }
    if (n_keep == nU) continue;  // the point cuts no bound
    int tot = n_keep;
#pragma unroll
    for (int j = 0; j < M; j++) tot += n_child[j];
    if (tot > cap) {
      over = true;
      break;
    }
    __syncthreads();
    int off_keep = 0, off_child[M];
    {
      int acc = n_keep;
#pragma unroll
      for (int j = 0; j < M; j++) {
        off_child[j] = acc;
        acc += n_child[j];
      }
    }
    for (int base = 0; base < nU; base += 64) {
      const int u = base + l;
      const unsigned f = (u < nU) ? fl[u] : 0xffu;  // 0xff: not a bound
      const bool keep = (f == 0u);
      const unsigned long long mk = __ballot(keep);
      if (keep) V[off_keep + nv_prefix(mk)] = U[u];
      off_keep += __popcll(mk);
#pragma unroll
      for (int j = 0; j < M; j++) {
        const bool ch = (f != 0xffu) && ((f >> (1 + j)) & 1u);
        const unsigned long long mc = __ballot(ch);
        if (ch) {
          Bound<M> b = U[u];
          b.u[j] = p[j];
          b.z[j] = pi;
          V[off_child[j] + nv_prefix(mc)] = b;
        }
        off_child[j] += __popcll(mc);
      }
    }
    __syncthreads();
    cur ^= 1;
    nU = tot;
  }
  if (over) {
    if (l == 0) {
      cnt[s] = 0;
      atomicAdd(overflow, 1);
    }
    return;
  }
  // ---- cells: lower corner -u, upper corner -lb with lb_j = max_{k < j} z^k_j (lb_0 = -inf); empty boxes dropped ----
  const Bound<M>* U = ub[cur];
  double* out = slots + s * (int64_t)cap * 2 * M;
  int ncell = 0;
  for (int base = 0; base < nU; base += 64) {
    const int u = base + l;
    bool ok = u < nU;
    double lo[M], len[M];
    if (ok) {
#pragma unroll
      for (int j = 0; j < M; j++) {
        double lb = -INFINITY;
#pragma unroll
        for (int k = 0; k < M; k++)
          if (k < j) lb = fmax(lb, nv_zval<M>(pn, nref, U[u].z[k], j));
        ok = ok && (lb < U[u].u[j]);
        lo[j] = -U[u].u[j];
        len[j] = fmin(-lb, 1e10) - lo[j];  // BoTorch clamps cell upper bounds at 1e10
      }
    }
    const unsigned long long mk = __ballot(ok);
    if (ok) {
      double* c = out + (int64_t)(ncell + nv_prefix(mk)) * 2 * M;
#pragma unroll
      for (int j = 0; j < M; j++) {
        c[j] = lo[j];
        c[M + j] = len[j];
      }
    }
    ncell += __popcll(mk);
  }
  if (l == 0) cnt[s] = ncell;
}

// prefix offsets + packed arrays in the layout the scoring kernels read: off [S + 1], lo / ll / len [total, m]
__global__ __launch_bounds__(64) void bbh_cells_pack_kernel(const double* __restrict__ slots, const int* __restrict__ cnt, int S, int cap, int m,
                                                            int64_t* __restrict__ off, double* __restrict__ lo, double* __restrict__ ll,
                                                            double* __restrict__ len, int64_t* __restrict__ status) {
  const int l = threadIdx.x, s = blockIdx.x;
  int before32 = 0;  // (at most 65535 samples x 1024 cells)
  for (int j = l; j < s; j += 64) before32 += cnt[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before32 += __shfl_xor(before32, o, 64);
  const int64_t before = before32;
  const int c = cnt[s];
  if (l == 0) {
    off[s] = before;
    if (s == S - 1) {
      off[S] = before + c;
      status[0] = before + c;
      status[1] = cnt[S];  // samples whose bound list overflowed
    }
  }
  const double* src = slots + (int64_t)s * cap * 2 * m;
  for (int e = l; e < c * m; e += 64) {
    const int cell = e / m, o = e % m;
    const double v = src[(int64_t)cell * 2 * m + m + o];
    lo[before * m + e] = src[(int64_t)cell * 2 * m + o];
    len[before * m + e] = v;
    ll[before * m + e] = log(v);
  }
}

bbh_nehvi_state* nv_state(bbh_handle* h) {
  if (!h->nehvi_state) h->nehvi_state = new bbh_nehvi_state();
  return (bbh_nehvi_state*)h->nehvi_state;
}

size_t nv_lds(int nb, int m, int cap) {
  const size_t bound = sizeof(double) * m + sizeof(int) * m + (m % 2 ? sizeof(int) : 0);  // sizeof(Bound<M>)
  return sizeof(double) * 2 * (size_t)nb * m + 2 * bound * cap + ((cap + 7) & ~7) + sizeof(int) * nb;
}

}  // namespace

void bbh_nehvi_destroy(bbh_handle* h) {
  bbh_nehvi_state* st = (bbh_nehvi_state*)h->nehvi_state;
  if (!st) return;
  if (st->d_slots) hipFree(st->d_slots);
  if (st->d_pack) hipFree(st->d_pack);
  if (st->d_cnt) hipFree(st->d_cnt);
  if (st->d_ref) hipFree(st->d_ref);
  if (st->h_status) hipHostFree(st->h_status);
  delete st;
  h->nehvi_state = nullptr;
}

extern "C" int bbh_cells_build_dev(bbh_handle* h, const double* Fb_dev, int64_t S, int64_t nb, int32_t m, const double* ref_host,
                                   int64_t* total_out, int64_t* overflow_out) {
  if (!h) return -1;
  if (!ref_host || !total_out || !overflow_out || S < 1 || S > 65535 || nb < 0 || (nb > 0 && !Fb_dev) || m < 1 || m > BBH_MAX_OBJECTIVES) {
    h->err = "bbh_cells_build_dev: bad arguments (1 <= m <= 4, 1 <= S <= 65535)";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (nb > NV_MAXPTS) {  // not taken on the device: the caller uses the host form (bbh_cells_create)
    *total_out = 0;
    *overflow_out = S;
    return 0;
  }
  bbh_nehvi_state* st = nv_state(h);
  // bound capacity: m <= 3 needs at most 2 P + 1 bounds for P points; m = 4 can need more (overflow -> host form)
  int cap = (int)(m <= 3 ? 2 * nb + 2 : 6 * nb + 8);
  if (cap < 8) cap = 8;
  if (cap > NV_MAXBOUNDS) cap = NV_MAXBOUNDS;
  const size_t lds = nv_lds((int)nb, m, cap);
  // the device's own LDS limit decides (160 KB on gfx950); a sample set that does not fit goes to the host form like an overflow
  int lds_max = 0;
  if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess) lds_max = 64 * 1024;
  const int lds_limit = lds_max - 64;
  if (lds > (size_t)lds_limit) {
    *total_out = 0;
    *overflow_out = S;
    return 0;
  }
  const size_t slot_bytes = sizeof(double) * (size_t)S * cap * 2 * m;
  const size_t pack_bytes = sizeof(double) * ((size_t)S + 1 + 3 * (size_t)S * cap * m);
  hipStream_t s = h->stream;
  if (slot_bytes > st->slot_bytes) {
    if (st->d_slots) hipFree(st->d_slots);
    st->d_slots = nullptr;
    st->slot_bytes = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&st->d_slots, slot_bytes));
    st->slot_bytes = slot_bytes;
  }
  if (pack_bytes > st->pack_bytes) {
    if (st->d_pack) hipFree(st->d_pack);
    st->d_pack = nullptr;
    st->pack_bytes = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&st->d_pack, pack_bytes));
    st->pack_bytes = pack_bytes;
  }
  if (S + 1 > st->cnt_cap) {
    if (st->d_cnt) hipFree(st->d_cnt);
    st->d_cnt = nullptr;
    st->cnt_cap = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&st->d_cnt, sizeof(int) * (size_t)(S + 1)));
    st->cnt_cap = S + 1;
  }
  if (!st->d_ref) BBH_HIP_TRY(h, hipMalloc((void**)&st->d_ref, sizeof(double) * BBH_MAX_OBJECTIVES));
  if (!st->h_status) BBH_HIP_TRY(h, hipHostMalloc((void**)&st->h_status, sizeof(int64_t) * 2, hipHostMallocDefault));
  st->S = S;
  st->m = m;
  st->cap = cap;
  BBH_HIP_TRY(h, hipMemcpyAsync(st->d_ref, ref_host, sizeof(double) * m, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemsetAsync(st->d_cnt + S, 0, sizeof(int), s));
  dim3 grid((unsigned)S), block(64);
  // hipFuncSetAttribute is per DEVICE (a process-wide flag left a second GPU of the same process without the raised limit)
#define NV_LAUNCH(MM)                                                                                                                \
  {                                                                                                                                  \
    static bool attr_set[64] = {};                                                                                                   \
    if (!attr_set[h->device & 63]) {                                                                                                 \
      BBH_HIP_TRY(h, hipFuncSetAttribute((const void*)bbh_cells_kernel<MM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit)); \
      attr_set[h->device & 63] = true;                                                                                               \
    }                                                                                                                                \
    hipLaunchKernelGGL(bbh_cells_kernel<MM>, grid, block, lds, s, Fb_dev, (int)nb, st->d_ref, cap, st->d_slots, st->d_cnt, st->d_cnt + S); \
  }
  switch (m) {
    case 1: NV_LAUNCH(1) break;
    case 2: NV_LAUNCH(2) break;
    case 3: NV_LAUNCH(3) break;
    default: NV_LAUNCH(4) break;
  }
#undef NV_LAUNCH
  BBH_HIP_TRY(h, hipGetLastError());
  int64_t* off = (int64_t*)st->d_pack;
  double* lo = st->d_pack + (S + 1);
  double* ll = lo + S * cap * m;
  double* len = ll + S * cap * m;
  int64_t* status_dev = nullptr;
  BBH_HIP_TRY(h, hipHostGetDevicePointer((void**)&status_dev, st->h_status, 0));
  hipLaunchKernelGGL(bbh_cells_pack_kernel, grid, block, 0, s, st->d_slots, st->d_cnt, (int)S, cap, (int)m, off, lo, ll, len, status_dev);
  BBH_HIP_TRY(h, hipGetLastError());
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  st->total = st->h_status[0];
  *total_out = st->h_status[0];
  *overflow_out = st->h_status[1];
  return 0;
}

// The device-resident cell lists in the layout of bbh_cells_get (tests, statistics).
extern "C" int bbh_cells_read_dev(bbh_handle* h, int64_t* off_host, double* lo_host, double* loglen_host) {
  if (!h) return -1;
  bbh_nehvi_state* st = (bbh_nehvi_state*)h->nehvi_state;
  if (!st || st->S < 1 || !off_host) {
    h->err = "bbh_cells_read_dev: no device-resident cells (bbh_cells_build_dev)";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  BBH_HIP_TRY(h, hipMemcpy(off_host, st->off(), sizeof(int64_t) * (st->S + 1), hipMemcpyDeviceToHost));
  if (st->total > 0) {
    if (!lo_host || !loglen_host) return -1;
    BBH_HIP_TRY(h, hipMemcpy(lo_host, st->lo(), sizeof(double) * st->total * st->m, hipMemcpyDeviceToHost));
    BBH_HIP_TRY(h, hipMemcpy(loglen_host, st->ll(), sizeof(double) * st->total * st->m, hipMemcpyDeviceToHost));
  }
  return 0;
}
