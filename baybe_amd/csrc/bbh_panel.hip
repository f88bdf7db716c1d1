// Fused GP posterior over a discrete candidate set (the dominant kernel of the path).
//
// Replaces, per candidate chunk, what BoTorch does inside
//   optimize_acqf_discrete -> acqf(chunk) -> model.posterior(chunk)
// (reference call site baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126; maths
// SURVEY.md §3.4 / Appendix A7): Normalize, cross-covariance K(X*,X), posterior mean
// c + k*^T alpha, posterior variance k** - ||L^-1 k*||^2, un-Standardize.
//
// One wavefront owns a tile of 16 candidates; everything is expressed in the fragment layout
// of v_mfma_f64_16x16x4_f64 so that no value ever leaves its lane between stages:
//
//   (1) distance GEMM      r2[train, cand] = A_aug[train, :] . B_aug[:, cand]
//         A_aug = [-2 a_i, |a_i|^2, 1], B_aug = [b_c, 1, |b_c|^2]^T, a/b = centred inputs / l
//         (M = 16 training points, N = 16 candidates, K = dn + 2 padded to 4).
//         C layout: lane l, reg r  <->  training point 4 r + (l>>4), candidate l & 15.
//   (2) kernel function    kv[r] = k(sqrt(r2)) (* outputscale * B[t_c, t_i])  — VALU, in place.
//         This *is* the A-fragment of the next GEMMs for k-step 4 tb + r:
//         lane l holds A[cand = l & 15][k = l >> 4].
//   (3) variance GEMM      V[cand, j] += kv . R[k, j],  R = L^-T (upper triangular, packed in
//         fragment order), accumulated for a window of W <= 16 column blocks ("pass");
//         only blocks j >= k are touched.  ||v||^2 is reduced from the accumulators.
//   (4) mean / cross GEMM  [mean | cross_1..p] += kv . [alpha | -beta_1..-beta_p] in the last pass.
//
// Passes of width W keep 8 W accumulator VGPRs (128 for W = 16), i.e. two waves per SIMD.  fp64 VALU
// and fp64 MFMA instructions share the SIMD's DP pipe on gfx950 (scripts/mfma_valu_overlap_probe.hip), so
// stage (2) is not hidden but minimised: software-pipelined micro-steps between the MFMAs, a staged
// rsq/Taylor Matérn evaluation, and a per-wave cache of kernel values instead of recomputation in later
// passes.  K(X*,X) is never materialised.  Device code: bbh_fused.h (instantiated per k-step count in
// bbh_fused_kd{0,2,4,6,8,12,16}.hip); this file holds operand packing, launch logic and the related kernels.
#include "bbh_coop2.h"
#include "bbh_coopg.h"
#include "bbh_small.h"

// ---- operand packing ------------------------------------------------------------------------
// R fragments of one pass: for tb in [0, j1), r in 0..3, jb in [max(j0, tb), j1):
//   lane l <- R[k = 16 tb + 4 r + (l>>4)][j = 16 jb + (l&15)] = X[j][k]   (X = L^-1, lower)
__global__ void bbh_pack_rfrag_kernel(const double* __restrict__ X, int64_t np, int j0, int W, double* __restrict__ out) {
  const int tb = blockIdx.x, r = blockIdx.y, l = threadIdx.x;
  const int j1 = j0 + W;
  const int lo = tb > j0 ? tb : j0;
  const int cnt = j1 - lo;
  // fragments before this tb within the pass
  int64_t off;
  if (tb <= j0)
    off = (int64_t)4 * W * tb;
  else
    off = (int64_t)4 * (W * (int64_t)j0 + (int64_t)(tb - j0) * j1 - ((int64_t)(j0 + tb - 1) * (tb - j0)) / 2);
  off += (int64_t)r * cnt;
  const int64_t k = 16 * (int64_t)tb + 4 * r + (l >> 4);
  for (int jb = lo; jb < j1; jb++) {
    const int64_t j = 16 * (int64_t)jb + (l & 15);
    out[(off + (jb - lo)) * 64 + l] = (k <= j) ? X[j * np + k] : 0.0;
  }
}

// Operand slices of the cooperative form (bbh_coop.h): wave w, group G >= g0, k-block i, k-step r, slot rho = G + s
//   lane l <- R[k = 16 tb + 4 r + (l>>4)][j = 16 jb + (l&15)],  tb = 4 (G - g0) + i,  jb = 4 (rho - g0) + (rho odd ? 3 - w : w);
// consecutive fragments are stored as lane-interleaved pairs (one global_load_dwordx4 fetches both)
__global__ void bbh_pack_coop_kernel(const double* __restrict__ X, int64_t np, int g0, int64_t frags, double* __restrict__ out) {
  const int w = blockIdx.z, G = g0 + blockIdx.y, ir = blockIdx.x, l = threadIdx.x;
  const int i = ir >> 2, r = ir & 3;
  const int cnt = BBH_COOP_ROUNDS - G;
  const int64_t base = (int64_t)(coop_frags_before(G) - coop_frags_before(g0)) + (int64_t)ir * cnt;
  const int64_t k = 16 * (int64_t)(4 * (G - g0) + i) + 4 * r + (l >> 4);
  for (int s = 0; s < cnt; s++) {
    const int rho = G + s;
    const int64_t jb = 4 * (rho - g0) + ((rho & 1) ? 3 - w : w);
    const int64_t j = 16 * jb + (l & 15);
    const int64_t fr = base + s;  // fragment pairs are stored lane-interleaved: pair fr / 2, lane l, half fr % 2
    out[((int64_t)w * frags + (fr & ~(int64_t)1)) * 64 + 2 * l + (fr & 1)] = (k <= j) ? X[j * np + k] : 0.0;
  }
}

// Operand slices of the two-sweep cooperative form (bbh_coop2.h), one workgroup per fragment f of wave w's slice:
//   sweep A  groups G = g0 .. 7: k-block tb = 4 (G - g0) + i, slot rho = G + s -> column block 4 (rho - g0) + (rho odd ? 3 - w : w)
//   rect     groups kg = 0 .. 7 - g0: tb = 4 kg + i, slot s = 0 .. 7 -> column block 4 (8 - g0 + s) + (s odd ? 3 - w : w)
//   sweep B  groups G = 0 .. 7: tb = 4 (8 - g0 + G) + i, slot rho = G + s -> column block 4 (8 - g0 + rho) + (rho odd ? 3 - w : w)
// each group ordered (k-block i, k-step r, slot s); then 2 BBH_COOP_PAIRS zero fragments (requested by the ring, never used)
__global__ void bbh_pack_coop2_kernel(const double* __restrict__ X, int64_t np, int g0, int64_t frags, double* __restrict__ out) {
  const int w = blockIdx.y, l = threadIdx.x;
  int64_t f = blockIdx.x;
  const int64_t fr = f;
  int tb = -1, jb = 0, r = 0;
  const int RA = BBH_COOP_ROUNDS - g0;
  bool done = false;
  for (int G = g0; G < BBH_COOP_ROUNDS && !done; G++) {  // sweep A
    const int cnt = BBH_COOP_ROUNDS - G;
    if (f < 16 * cnt) {
      const int ir = (int)(f / cnt), s = (int)(f % cnt), rho = G + s;
      tb = 4 * (G - g0) + (ir >> 2);
      r = ir & 3;
      jb = 4 * (rho - g0) + ((rho & 1) ? 3 - w : w);
      done = true;
    } else {
      f -= 16 * cnt;
    }
  }
  for (int kg = 0; kg < RA && !done; kg++) {  // rectangular part
    if (f < 16 * BBH_COOP_ROUNDS) {
      const int ir = (int)(f / BBH_COOP_ROUNDS), s = (int)(f % BBH_COOP_ROUNDS);
      tb = 4 * kg + (ir >> 2);
      r = ir & 3;
      jb = 4 * (RA + s) + ((s & 1) ? 3 - w : w);
      done = true;
    } else {
      f -= 16 * BBH_COOP_ROUNDS;
    }
  }
  for (int G = 0; G < BBH_COOP_ROUNDS && !done; G++) {  // sweep B
    const int cnt = BBH_COOP_ROUNDS - G;
    if (f < 16 * cnt) {
      const int ir = (int)(f / cnt), s = (int)(f % cnt), rho = G + s;
      tb = 4 * (RA + G) + (ir >> 2);
      r = ir & 3;
      jb = 4 * (RA + rho) + ((rho & 1) ? 3 - w : w);
      done = true;
    } else {
      f -= 16 * cnt;
    }
  }
  double v = 0.0;
  if (done) {
    const int64_t k = 16 * (int64_t)tb + 4 * r + (l >> 4), j = 16 * (int64_t)jb + (l & 15);
    if (k <= j) v = X[j * np + k];
  }
  out[((int64_t)w * frags + (fr & ~(int64_t)1)) * 64 + 2 * l + (fr & 1)] = v;  // lane-interleaved fragment pairs
}

// Bm[i][0] = alpha[i] (i < np), everything else zero
__global__ void bbh_init_meanB_kernel(const double* __restrict__ alpha, int64_t np, int64_t rows, double* __restrict__ Bm) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * 16) return;
  const int64_t i = e >> 4;
  const int c = (int)(e & 15);
  Bm[e] = (c == 0 && i < np) ? alpha[i] : 0.0;
}

// Bm[i][1 + j] = -betaT[j][i] (i < np);  Bm[np + j][1 + j] = 1
__global__ void bbh_set_beta_kernel(const double* __restrict__ betaT, int64_t ldb, int64_t np, int p, double* __restrict__ Bm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np + 16) return;
  for (int j = 0; j < p; j++) {
    double v;
    if (i < np)
      v = -betaT[(int64_t)j * ldb + i];
    else
      v = (i - np == j) ? 1.0 : 0.0;
    Bm[i * 16 + 1 + j] = v;
  }
}

// Host-side layout of the augmented training fragments for blocks [tb0, tb1):
//   frag[tb][k][l] = Aaug[16 tb + (l & 15)][4 k + (l >> 4)]
// pts: normalised numerical coordinates [cnt, dn] of the real points in this range (row 0 is
// point index 16 tb0); anything beyond cnt is padding (huge distance -> kernel value 0).
static void host_pack_trainfrag(const bbh_handle* h, const double* pts, int64_t cnt, int64_t tb0, int64_t tb1,
                                std::vector<double>& out, const double* ls = nullptr) {
  const int dn = h->dn, kd = h->kd;
  if (!ls) ls = h->theta.data() + 3;
  out.assign((size_t)(tb1 - tb0) * kd * 64, 0.0);
  std::vector<double> a(dn);
  for (int64_t tb = tb0; tb < tb1; tb++)
    for (int c16 = 0; c16 < 16; c16++) {
      const int64_t i = (tb - tb0) * 16 + c16;
      const bool real = i < cnt;
      double na = 0.0;
      if (real)
        for (int j = 0; j < dn; j++) {
          a[j] = (pts[i * dn + j] - h->xcenter[j]) / ls[j];
          na += a[j] * a[j];
        }
      for (int k = 0; k < kd; k++)
        for (int qq = 0; qq < 4; qq++) {
          const int dim = 4 * k + qq;
          double v = 0.0;
          if (real) {
            if (dim < dn)
              v = -2.0 * a[dim];
            else if (dim == dn)
              v = na;
            else if (dim == dn + 1)
              v = 1.0;
          } else if (dim == dn) {
            v = 1e8;
          }
          out[((size_t)(tb - tb0) * kd + k) * 64 + qq * 16 + c16] = v;
        }
    }
}

// Models the generic-production cooperative kernel covers: everything bbh_materialised_only() sends to the materialised-K*
// path whose factors are Matérn-5/2, -3/2, RBF, rational quadratic or piecewise polynomial with q >= 1 (Matérn-1/2 and the
// q = 0 piecewise polynomial (1 - r)^j are not smooth at r = 0: the |a|^2 + |b|^2 - 2ab distances lose sqrt(eps) there, so they
// need the direct-difference distances and keep that path), n <= 512, d <= 30.
// k-steps of the generic form's metric GEMM: ceil((dn + 2) / 4) for distance / dot factors, ceil((2 dn + 1) / 4) once a periodic
// factor is present (cos and sin features), rounded up to an instantiated count; 0: none fits
static int bbh_coopg_kd(const bbh_handle* h) {
  const bbh_kern_spec ks = bbh_kern_spec_of(h);
  int feats = h->dn + 2;
  for (int f = 0; f < ks.F; f++)
    if (ks.kind[f] == BBH_KERNEL_PERIODIC && 2 * h->dn + 1 > feats) feats = 2 * h->dn + 1;
  const int kd = (feats + 3) / 4;
  for (int c : {2, 4, 6, 8})
    if (kd <= c) return c;
  return 0;
}

static bool bbh_coopg_model(const bbh_handle* h) {
  if (!bbh_materialised_only(h) || h->coop_mode <= 0 || !h->use_pipeline) return false;
  if (h->nb > 4 * BBH_COOP_ROUNDS || h->nb % 4 != 0) return false;
  const bbh_kern_spec ks = bbh_kern_spec_of(h);
  for (int f = 0; f < ks.F; f++)
    if (ks.kind[f] == BBH_KERNEL_MATERN12 || ks.kind[f] == BBH_KERNEL_PIECEWISE0) return false;
  const int kd = bbh_coopg_kd(h);
  return kd > 0 && bbh_coopg_launch(kd, ks.F, dim3(0), 0, nullptr, CoopGArgs{});
}

// Training fragments of factor f of the generic form (see CoopGFeat in bbh_coopg.h): frag[tb][k][l] = A[16 tb + (l & 15)][4 k + (l >> 4)]
// pts: normalised rows [cnt, dn] (nullptr: the training rows); rows beyond cnt are padding
static void host_pack_trainfrag_generic(const bbh_handle* h, const bbh_kern_spec& ks, int f, int kd, int64_t nblocks, std::vector<double>& out,
                                        const double* pts = nullptr, int64_t cnt = -1) {
  const int dn = h->dn, kind = ks.kind[f];
  const double* th = h->theta.data();
  const double* ls = th + ks.ls_off[f];
  if (!pts) {
    pts = h->xn_host.data();
    cnt = h->n;
  }
  out.assign((size_t)nblocks * kd * 64, 0.0);
  std::vector<double> row((size_t)4 * kd);
  for (int64_t tb = 0; tb < nblocks; tb++)
    for (int c16 = 0; c16 < 16; c16++) {
      const int64_t i = tb * 16 + c16;
      const bool real = i < cnt;
      std::fill(row.begin(), row.end(), 0.0);
      const double* x = real ? pts + i * dn : nullptr;
      if (kind == BBH_KERNEL_PERIODIC) {
        double c0 = 0.0;
        for (int j = 0; j < dn; j++) c0 += 0.5 / ls[j];
        if (real)
          for (int j = 0; j < dn; j++) {
            const double al = 2.0 * M_PI * x[j] / th[ks.per_off + f * dn + j];
            row[j] = -cos(al) * 0.5 / ls[j];
            row[dn + j] = -sin(al) * 0.5 / ls[j];
          }
        row[2 * dn] = real ? c0 : 1e8;
      } else if (BBH_KIND_IS_DOT(kind)) {
        if (real)
          for (int j = 0; j < dn; j++) row[j] = x[j] / ls[j];
        row[dn] = real ? 0.0 : 1e8;
      } else {
        double na = 0.0;
        if (real)
          for (int j = 0; j < dn; j++) {
            const double a = (x[j] - h->xcenter[j]) / ls[j];
            row[j] = -2.0 * a;
            na += a * a;
          }
        row[dn] = real ? na : 1e8;
        row[dn + 1] = real ? 1.0 : 0.0;
      }
      for (int k = 0; k < kd; k++)
        for (int qq = 0; qq < 4; qq++) out[((size_t)tb * kd + k) * 64 + qq * 16 + c16] = row[4 * k + qq];
    }
}

// ... and the matching candidate-side feature map [4 kd]
static void host_feature_map(const bbh_handle* h, const bbh_kern_spec& ks, int f, int kd, CoopGFeat* out) {
  const int dn = h->dn, kind = ks.kind[f];
  const double* th = h->theta.data();
  const double* ls = th + ks.ls_off[f];
  for (int e = 0; e < 4 * kd; e++) out[e] = CoopGFeat{0.0, 0.0, -1, 5};
  for (int j = 0; j < dn; j++) {
    const double rng = h->hi[j] - h->lo[j];
    if (kind == BBH_KERNEL_PERIODIC) {
      const double per = th[ks.per_off + f * dn + j];
      out[j] = CoopGFeat{2.0 * M_PI / (rng * per), -2.0 * M_PI * h->lo[j] / (rng * per), j, 1};
      out[dn + j] = CoopGFeat{2.0 * M_PI / (rng * per), -2.0 * M_PI * h->lo[j] / (rng * per), j, 2};
    } else if (BBH_KIND_IS_DOT(kind)) {
      out[j] = CoopGFeat{1.0 / (rng * ls[j]), -h->lo[j] / (rng * ls[j]), j, 0};
    } else {
      out[j] = CoopGFeat{1.0 / (rng * ls[j]), -(h->lo[j] / rng + h->xcenter[j]) / ls[j], j, 0};
    }
  }
  if (kind == BBH_KERNEL_PERIODIC) {
    out[2 * dn] = CoopGFeat{0.0, 1.0, -1, 3};
  } else {
    out[dn] = CoopGFeat{0.0, 1.0, -1, 3};
    out[dn + 1] = CoopGFeat{0.0, 0.0, -1, 4};
  }
}

int bbh_pack_operands(bbh_handle* h) {
  hipStream_t s = h->stream;
  const int64_t np = h->np, nb = h->nb;
  const int dn = h->dn, kd = h->kd, T = h->T;
  const double* th = h->theta.data();
  // ---- pass decomposition: full 16-block windows first, the remainder last ----
  std::vector<int> widths;
  {
    h->wmax = 16;
    int64_t left = nb;
    while (left >= h->wmax) {
      widths.push_back(h->wmax);
      left -= h->wmax;
    }
    if (left > 0) widths.push_back((int)left);
  }
  if ((int)widths.size() > BBH_MAX_PASS) {
    h->err = "n_train too large for the fused kernel (max 16384)";
    return -5;
  }
  // ---- sizes & allocation ----
  int64_t total_frags = 0;
  std::vector<int64_t> pass_off(widths.size());
  {
    int j0 = 0;
    for (size_t ps = 0; ps < widths.size(); ps++) {
      pass_off[ps] = total_frags * 64;
      const int W = widths[ps], j1 = j0 + W;
      total_frags += (int64_t)4 * W * j0;                       // rectangular part
      for (int tb = j0; tb < j1; tb++) total_frags += 4 * (j1 - tb);  // triangular part
      j0 = j1;
    }
  }
  if (!h->d_pass_off) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_pass_off, sizeof(int64_t) * BBH_MAX_PASS));
  if (!h->d_pass_w) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_pass_w, sizeof(int) * BBH_MAX_PASS));
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_pass_off, pass_off.data(), sizeof(int64_t) * pass_off.size(), hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_pass_w, widths.data(), sizeof(int) * widths.size(), hipMemcpyHostToDevice, s));
  h->npass = (int)widths.size();
  h->pass_w_last = widths.back();
  if (!h->d_rfrag || h->rfrag_elems != total_frags * 64) {
    if (h->d_rfrag) hipFree(h->d_rfrag);
    h->d_rfrag = nullptr;
    BBH_HIP_TRY(h, hipMalloc((void**)&h->d_rfrag, sizeof(double) * total_frags * 64));
    h->rfrag_elems = total_frags * 64;
  }
  if (!h->d_trainfrag) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_trainfrag, sizeof(double) * (nb + 1) * kd * 64));
  if (!h->d_meanB) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_meanB, sizeof(double) * (np + 16) * 16));
  if (!h->d_sclofs) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_sclofs, sizeof(double) * 2 * dn));
  if (!h->d_numcol) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_numcol, sizeof(int) * dn));
  if (!h->d_tasktbl) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_tasktbl, sizeof(double) * T * T));
  if (!h->d_taskext) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_taskext, sizeof(int) * (np + 16)));
  // ---- R fragments from X = L^-1 ----
  {
    int j0 = 0;
    for (size_t ps = 0; ps < widths.size(); ps++) {
      const int W = widths[ps];
      hipLaunchKernelGGL(bbh_pack_rfrag_kernel, dim3((unsigned)(j0 + W), 4), dim3(64), 0, s, h->d_X, np, j0, W,
                         h->d_rfrag + pass_off[ps]);
      j0 += W;
    }
  }
  // ---- register-resident small-model form (n <= 128): the lower triangle of L^-T as fragments, bbh_small.h ----
  h->small_nb = 0;
  if (h->small_on && h->use_pipeline && nb <= 8 && h->n >= 1) {
    const int NB = (int)((h->n + 15) / 16);
    const bool has_tbl0 = (T > 1) || h->desc.use_outputscale;
    if (bbh_small_launch(h->kd, h->desc.kernel_kind, has_tbl0, NB, 0, h->num_cu, nullptr, SmallArgs{})) {
      if (!h->d_rsmall) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_rsmall, sizeof(double) * 36 * 4 * 64));
      hipLaunchKernelGGL(bbh_pack_small_kernel, dim3((unsigned)NB, (unsigned)NB, 4), dim3(64), 0, s, h->d_X, np, NB, h->d_rsmall);
      h->small_nb = NB;
    }
  }
  // ---- operand slices of the cooperative form (n <= 512, instantiated models only) ----
  h->coop_ready = false;
  {
    const bool has_tbl0 = (T > 1) || h->desc.use_outputscale;
    if (h->coop_mode > 0 && h->use_pipeline && nb <= 4 * BBH_COOP_ROUNDS && nb % 4 == 0 &&
        bbh_coop_launch(h->kd, h->desc.kernel_kind, has_tbl0, dim3(0), 0, nullptr, CoopArgs{})) {
      const int g0 = BBH_COOP_ROUNDS - (int)(nb / 4);
      const int64_t frags = coop_frags_before(BBH_COOP_ROUNDS) - coop_frags_before(g0);
      if (!h->d_rstream || h->rstream_frags != frags) {
        if (h->d_rstream) hipFree(h->d_rstream);
        h->d_rstream = nullptr;
        BBH_HIP_TRY(h, hipMalloc((void**)&h->d_rstream, sizeof(double) * 4 * frags * 64));
        h->rstream_frags = frags;
      }
      hipLaunchKernelGGL(bbh_pack_coop_kernel, dim3(16, (unsigned)(BBH_COOP_ROUNDS - g0), 4), dim3(64), 0, s, h->d_X, np, g0, frags,
                         h->d_rstream);
      h->coop_g0 = g0;
      h->coop_ready = true;
    }
  }
  // ---- generic-production cooperative form: the same operand slices, per-factor training fragments and candidate scaling ----
  h->coopg_ready = false;
  if (bbh_coopg_model(h)) {
    const bbh_kern_spec ks = bbh_kern_spec_of(h);
    const int g0 = BBH_COOP_ROUNDS - (int)(nb / 4);
    const int64_t frags = coop_frags_before(BBH_COOP_ROUNDS) - coop_frags_before(g0);
    if (!h->d_rstream || h->rstream_frags != frags) {
      if (h->d_rstream) hipFree(h->d_rstream);
      h->d_rstream = nullptr;
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_rstream, sizeof(double) * 4 * frags * 64));
      h->rstream_frags = frags;
    }
    hipLaunchKernelGGL(bbh_pack_coop_kernel, dim3(16, (unsigned)(BBH_COOP_ROUNDS - g0), 4), dim3(64), 0, s, h->d_X, np, g0, frags,
                       h->d_rstream);
    h->coop_g0 = g0;
    const int kdg = bbh_coopg_kd(h);
    const int64_t per = (nb + 1) * (int64_t)kdg * 64;
    if (!h->d_trainfrag_f || h->tf_f_elems != per * ks.F) {
      if (h->d_trainfrag_f) hipFree(h->d_trainfrag_f);
      if (h->d_sclofs_f) hipFree(h->d_sclofs_f);
      h->d_trainfrag_f = h->d_sclofs_f = nullptr;
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_trainfrag_f, sizeof(double) * per * ks.F));
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_sclofs_f, sizeof(CoopGFeat) * 4 * 8 * BBH_MAX_FACTORS));  // the feature maps [F][4 kd]
      h->tf_f_elems = per * ks.F;
    }
    std::vector<double> tff;
    std::vector<CoopGFeat> feat((size_t)ks.F * 4 * kdg);
    for (int f = 0; f < ks.F; f++) {
      std::vector<double> one;
      host_pack_trainfrag_generic(h, ks, f, kdg, nb + 1, one);
      tff.insert(tff.end(), one.begin(), one.end());
      host_feature_map(h, ks, f, kdg, feat.data() + (size_t)f * 4 * kdg);
    }
    BBH_HIP_TRY(h, hipMemcpyAsync(h->d_trainfrag_f, tff.data(), sizeof(double) * tff.size(), hipMemcpyHostToDevice, s));
    BBH_HIP_TRY(h, hipMemcpyAsync(h->d_sclofs_f, feat.data(), sizeof(CoopGFeat) * feat.size(), hipMemcpyHostToDevice, s));
    BBH_HIP_TRY(h, hipStreamSynchronize(s));  // the staging vectors go out of scope
    h->coopg_ready = true;
  }
  // ---- operand slices of the two-sweep cooperative form (512 < n <= 1024) ----
  h->coop2_ready = false;
  {
    const bool has_tbl0 = (T > 1) || h->desc.use_outputscale;
    if (h->coop_mode > 0 && h->use_pipeline && nb > 4 * BBH_COOP_ROUNDS && nb <= 8 * BBH_COOP_ROUNDS && nb % 4 == 0 &&
        bbh_coop2_launch(h->kd, h->desc.kernel_kind, has_tbl0, dim3(0), 0, nullptr, CoopArgs{})) {
      const int g0 = 2 * BBH_COOP_ROUNDS - (int)(nb / 4);
      const int64_t frags = coop2_frags(g0);
      if (!h->d_rstream || h->rstream_frags != frags) {
        if (h->d_rstream) hipFree(h->d_rstream);
        h->d_rstream = nullptr;
        BBH_HIP_TRY(h, hipMalloc((void**)&h->d_rstream, sizeof(double) * 4 * frags * 64));
        h->rstream_frags = frags;
      }
      hipLaunchKernelGGL(bbh_pack_coop2_kernel, dim3((unsigned)frags, 4), dim3(64), 0, s, h->d_X, np, g0, frags, h->d_rstream);
      h->coop_g0 = g0;
      h->coop2_ready = true;
    }
  }
  // ---- training fragments (+ one all-padding block reserved for pending points) ----
  std::vector<double> tf;
  host_pack_trainfrag(h, h->xn_host.data(), h->n, 0, nb + 1, tf);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_trainfrag, tf.data(), sizeof(double) * tf.size(), hipMemcpyHostToDevice, s));
  // ---- candidate-side scale/offset, column map, task table ----
  std::vector<double> so(2 * dn);
  for (int j = 0; j < dn; j++) {
    const double rng = h->hi[j] - h->lo[j];
    so[j] = 1.0 / (rng * th[3 + j]);
    so[dn + j] = -(h->lo[j] / rng + h->xcenter[j]) / th[3 + j];
  }
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_sclofs, so.data(), sizeof(double) * 2 * dn, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_numcol, h->numcol.data(), sizeof(int) * dn, hipMemcpyHostToDevice, s));
  const double os = h->desc.use_outputscale ? th[2] : 1.0;
  std::vector<double> tbl((size_t)T * T, os);
  if (T > 1)
    for (int i = 0; i < T * T; i++) tbl[i] = os * th[3 + dn + i];
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_tasktbl, tbl.data(), sizeof(double) * T * T, hipMemcpyHostToDevice, s));
  std::vector<int> te(np + 16, 0);
  for (int64_t i = 0; i < h->n; i++) te[i] = h->task_host[i];
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_taskext, te.data(), sizeof(int) * (np + 16), hipMemcpyHostToDevice, s));
  // ---- mean operand ----
  hipLaunchKernelGGL(bbh_init_meanB_kernel, dim3((unsigned)(((np + 16) * 16 + 255) / 256)), dim3(256), 0, s, h->d_alpha,
                     np, np + 16, h->d_meanB);
  BBH_HIP_TRY(h, hipStreamSynchronize(s));  // host staging vectors go out of scope
  h->nb_ext = nb;
  h->p = 0;
  return 0;
}

int bbh_launch_fused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev,
                     double* cross_dev, bool with_var) {
  if (N <= 0) return 0;
  if (bbh_is_rff(h)) return bbh_rff_posterior_launch(h, X_dev, N, ldx, mean_dev, with_var ? var_dev : nullptr, cross_dev);
  // composite / RQ / piecewise kernels: the cooperative form with the generic production for variance passes without pending
  // columns (bbh_coopg.h), otherwise the materialised-K* path (fused qLogEI: applied by the caller)
  const bool coopg = h->coopg_ready && with_var && h->p == 0 && !cross_dev && !h->fuse_qz && h->use_mean_valu;
  // ... and their mean-only / cross-covariance passes (steps >= 2 of a greedy batch) on bbh_coopg_cross_kernel (BBH_COOPG_CROSS=0: A/B)
  const bool coopg_cross = h->coopg_ready && !with_var && !h->fuse_qz && h->coopg_cross_on;
  if (bbh_materialised_only(h) && !coopg && !coopg_cross)
    return bbh_launch_unfused_ext(h, X_dev, N, ldx, mean_dev, with_var ? var_dev : nullptr, cross_dev);
  FusedArgs a;
  a.X = X_dev;
  a.N = N;
  a.ldx = ldx;
  a.trainfrag = h->d_trainfrag;
  a.rfrag = h->d_rfrag;
  a.meanB = h->d_meanB;
  a.scl = h->d_sclofs;
  a.ofs = h->d_sclofs + h->dn;
  a.numcol = h->d_numcol;
  a.tasktbl = h->d_tasktbl;
  a.taskmean = h->hadamard ? h->d_theta + bbh_hadamard_offset(h) + h->T : nullptr;
  a.taskext = h->d_taskext;
  a.mean = mean_dev;
  a.var = var_dev;
  a.cross = cross_dev;
  a.pass_off = h->d_pass_off;
  a.pass_w = h->d_pass_w;
  a.npass = h->npass;
  a.kind = h->desc.kernel_kind;
  a.dn = h->dn;
  a.kd = h->kd;
  a.nb = (int)h->nb;
  a.nb_ext = (int)h->nb_ext;
  a.task_col = h->desc.task_col;
  a.T = h->T;
  a.p = h->p;
  a.with_var = with_var ? 1 : 0;
  a.ybar = h->ybar;
  a.ysd = h->ysd;
  a.mean_const = h->theta[1];
  a.prior_scale = h->desc.use_outputscale ? h->theta[2] : 1.0;
  a.qz = nullptr;
  a.qS = 0;
  a.q_best_f = 0.0;
  a.q_sign = 1.0;
  a.q_alive = nullptr;
  a.q_scores = nullptr;
  if (h->fuse_qz && with_var) {  // set by bbh_score_qlogei for this launch only
    a.qz = h->fuse_qz;
    a.qS = h->fuse_S;
    a.q_best_f = h->fuse_best_f;
    a.q_sign = h->fuse_sign;
    a.q_alive = h->fuse_alive;
    a.q_scores = h->fuse_scores;
  }
  const bool has_tbl = (h->T > 1) || h->desc.use_outputscale;
  // (set below once the kernel form is known)
  const bool m52 = (a.kind == BBH_KERNEL_MATERN52);
  // software-pipelined instantiations exist for Matérn-5/2 (with / without table), Matérn-3/2 and RBF (without) with kd in
  // {2, 4, 6, 8, 12, 16}
  // (bbh_set_model rounds kd up to one of these when d allows); everything else takes the plain form
  const bool rbf = (a.kind == BBH_KERNEL_RBF), m32 = (a.kind == BBH_KERNEL_MATERN32);
  a.has_tbl = has_tbl ? 1 : 0;
  a.numcol_identity = 1;
  for (int j = 0; j < h->dn; j++)
    if (h->numcol[j] != j) a.numcol_identity = 0;
  const int kdp = ((m52 || ((rbf || m32) && !has_tbl)) && h->use_pipeline && (h->kd == 2 || h->kd == 4 || h->kd == 6 || h->kd == 8 || h->kd == 12 || h->kd == 16)) ? h->kd : 0;
  a.nblk = (N + 63) / 64;
  dim3 grid((unsigned)a.nblk), block(256);
  // alpha in LDS costs 8 n bytes: beyond n = 4096 it would crowd out the candidate fragments / the cache
  a.mean_valu = (kdp && h->p == 0 && !cross_dev && h->use_mean_valu && h->nb <= 256) ? 1 : 0;
  size_t lds = sizeof(double) * (4 * h->kd * 64 + (a.qz ? a.qS : 0) + (a.mean_valu ? 16 * h->nb : 0));
  a.kvcache = nullptr;
  a.ncache = 0;
  a.nl = 0;
  a.slab_flags = nullptr;
  a.nslab = 0;
  a.nxcc = 1;
  if (kdp && with_var && h->npass > 1 && h->use_kvcache) {
    // Kernel-value cache for the k-blocks left of the last pass.  As many of them as fit next to the other
    // LDS users without costing the second workgroup per CU (half of the CU's LDS per workgroup) stay in
    // wave-private LDS (2 KB per k-block and wave); the rest goes to slabs in global memory claimed per wave.
    a.ncache = (int)(h->nb - h->pass_w_last);
    const size_t lds_wg = h->lds_per_block / 2;  // two workgroups per CU
    const size_t budget = lds_wg > lds ? lds_wg - lds : 0;
    int nl = (int)(budget / (4 * 256 * sizeof(double)));
    if (h->kv_lds_blocks >= 0 && nl > h->kv_lds_blocks) nl = h->kv_lds_blocks;
    a.nl = nl < a.ncache ? nl : a.ncache;
    // The k-blocks that do not fit LDS go to global slabs only when there are many of them (n >= ~768:
    // -5 % at n = 1024, -2.5 % at n = 2048); for a few (n = 512: 8 blocks) recomputing them is as fast as
    // streaming them through the fabric (5.17 vs 5.18 ms) and moves 4 GB less per launch.
    const int rest = a.ncache - a.nl;
    const bool global_part = h->kv_global_mode == 1 || (h->kv_global_mode < 0 && rest >= 24);
    if (!global_part) a.ncache = a.nl;
    lds += sizeof(double) * 4 * 256 * (size_t)a.nl;
  }
  if (a.ncache > a.nl) {
    a.nslab = 2 * 8 * h->num_cu;  // at least twice the resident waves (8 or 4 per CU under the launch bounds)
    a.nxcc = (h->num_cu % 8 == 0 && h->num_cu >= 64) ? 8 : 1;  // MI355X: 8 XCDs x 32 CUs
    const size_t need = sizeof(double) * (size_t)a.nslab * (size_t)(a.ncache - a.nl) * 256;
    if (need > h->kvcache_bytes) {
      if (h->d_kvcache) BBH_HIP_TRY(h, hipFree(h->d_kvcache));
      h->d_kvcache = nullptr;
      h->kvcache_bytes = 0;
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_kvcache, need));
      h->kvcache_bytes = need;
    }
    if (!h->d_slab_flags) {
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_slab_flags, sizeof(int) * (size_t)a.nslab));
      BBH_HIP_TRY(h, hipMemsetAsync(h->d_slab_flags, 0, sizeof(int) * (size_t)a.nslab, h->stream));
    }
    a.kvcache = h->d_kvcache;
    a.slab_flags = h->d_slab_flags;
  }
  bbh_timed_scope timed(h, with_var ? BBH_TIMED_POSTERIOR : BBH_TIMED_CROSS);
  if (coopg || coopg_cross) {
    const bbh_kern_spec ks = bbh_kern_spec_of(h);
    const double* th = h->theta.data();
    CoopGArgs g;
    g.c.f = a;
    g.c.f.mean_valu = 1;
    g.c.rstream = h->d_rstream;
    g.c.frags = h->rstream_frags;
    g.c.g0 = h->coop_g0;
    g.F = ks.F;
    g.combine = ks.combine;
    g.has_tbl = has_tbl ? 1 : 0;
    g.jb = ks.jb;
    g.has_dot = 0;
    for (int f = 0; f < BBH_MAX_FACTORS; f++) {
      g.kind[f] = ks.kind[f < ks.F ? f : 0];
      g.grp[f] = ks.grp[f];
      g.fos[f] = (ks.F > 1 && f < ks.F) ? th[ks.fos_off + f] : 1.0;
      g.alpha[f] = (ks.alpha_off >= 0 && f < ks.F) ? th[ks.alpha_off + f] : 1.0;
      if (f < ks.F && BBH_KIND_IS_DOT(ks.kind[f])) g.has_dot = 1;
    }
    g.prior_k0 = ks.F > 1 ? bbh_combine(ks.F, ks.combine, ks.grp, g.fos) : 1.0;  // (stationary factors: k_f(x, x) = 1)
    const int kdg = bbh_coopg_kd(h);
    g.trainfrag_f = h->d_trainfrag_f;
    g.tf_stride = (h->nb + 1) * (int64_t)kdg * 64;
    g.feat = (const CoopGFeat*)h->d_sclofs_f;
    if (coopg_cross) {
      if (!bbh_coopg_cross_launch(kdg, ks.F, dim3((unsigned)((N + 63) / 64)), h->stream, g)) {
        h->err = "generic cross-covariance pass: no instantiation for this model although its operands were packed";
        return -6;
      }
      BBH_HIP_TRY(h, hipGetLastError());
      return 0;
    }
    const size_t clds = sizeof(double) * (16 * (size_t)h->nb + (2 * 4 * 256 + 128));
    bbh_coopg_launch(kdg, ks.F, dim3((unsigned)((N + 15) / 16)), clds, h->stream, g);
    h->last_form = 4;
    BBH_HIP_TRY(h, hipGetLastError());
    return 0;
  }
  // Register-resident form (n <= 128, variance pass without pending columns): persistent waves, the model in registers / LDS
  // (64 < n <= 128: the operand fragments are 45 - 74 KB of LDS that every workgroup fills first - ahead of the cooperative form
  // only once the candidate set amortises that: measured cross-overs, profiles/r04_ab_small_form.log; BBH_SMALL_FORCE=1 lifts the rule)
  const char* sf_env = getenv("BBH_SMALL_FORCE");
  const bool small_force = sf_env && sf_env[0] == '1';
  const int64_t small_min_rows[9] = {0, 0, 0, 0, 0, 20000, 60000, 120000, 300000};
  if (h->small_nb > 0 && h->small_on && with_var && h->p == 0 && !cross_dev && !a.qz && h->use_mean_valu &&
      (small_force || N >= small_min_rows[h->small_nb])) {
    SmallArgs sa;
    sa.f = a;
    sa.rsmall = h->d_rsmall;
    const int NB = h->small_nb;
    if (!bbh_small_launch(h->kd, a.kind, has_tbl, NB, (N + 15) / 16, h->num_cu, h->stream, sa)) {
      h->err = "register-resident posterior form: no instantiation for this model although its operands were packed";
      return -6;
    }
    h->last_form = 5;
    BBH_HIP_TRY(h, hipGetLastError());
    return 0;
  }
  // Cooperative form (n <= 512, variance pass without pending columns): ahead of the windowed form once a second
  // 16-block window would be needed (n > 256: 4.70 vs 5.17 ms on the bench shape), level at n = 256, a few per cent behind
  // below; for small candidate sets its four waves per tile cut the latency to a third (0.016 vs 0.052 ms for 1000 rows).
  // (round 3, profiles/r03_ab_small_n.log: with the hoisted set-up loads the cooperative form is level or ahead for every
  // n <= 256 and candidate count from 1e4 to 1e6 - its 16-candidate workgroups quantise the tail of a launch four times finer)
  const bool coop_pays = true;
  // (its own instantiation set: also RBF with a task table - the multi-task HVARFNER / BOTORCH presets - which the
  // windowed pipelined form does not have)
  const int kdc = (h->use_pipeline && (h->kd == 2 || h->kd == 4 || h->kd == 6 || h->kd == 8 || h->kd == 12 || h->kd == 16)) ? h->kd : 0;
  const bool coop_mean_valu = h->p == 0 && !cross_dev && h->use_mean_valu && h->nb <= 256;
  if (h->coop_ready && coop_pays && kdc && with_var && coop_mean_valu && !a.qz) {
    CoopArgs ca;
    ca.f = a;
    ca.rstream = h->d_rstream;
    ca.frags = h->rstream_frags;
    ca.g0 = h->coop_g0;
    // (Two candidate tiles per workgroup - every operand fragment feeding two MFMAs, half the vector-memory traffic per MFMA -
    // was built and measured in round 2: 4.72 vs 4.68 ms on the bench shape, profiles/r02_libs_nt2.log.  What the MFMA pipe
    // loses is per candidate, not per operand fragment; the variant was removed from the library, coop_group keeps its NT
    // template parameter.)
    const size_t clds = sizeof(double) * (16 * (size_t)h->nb + (2 * 4 * 256 + 128));
    const dim3 cgrid((unsigned)((N + 15) / 16));
    bbh_coop_launch(kdc, a.kind, has_tbl, cgrid, clds, h->stream, ca);
    h->last_form = 1;
    BBH_HIP_TRY(h, hipGetLastError());
    return 0;
  }
  // Two-sweep cooperative form (512 < n <= 1024): workgroups of 16 candidates instead of 64 (at N = 1e5 the windowed form's
  // 1563 workgroups are three rounds of the 512 resident slots - a quarter of the launch is tail) and no kernel-value cache
  // outside LDS; BBH_COOP=0 keeps the windowed form.
  if (h->coop2_ready && kdc && with_var && coop_mean_valu && !a.qz) {
    CoopArgs ca;
    ca.f = a;
    ca.rstream = h->d_rstream;
    ca.frags = h->rstream_frags;
    ca.g0 = h->coop_g0;
    const int ra = BBH_COOP_ROUNDS - ca.g0;  // archived k-block groups (at least the 16 KB sweep B's exchange slots need)
    const size_t clds = sizeof(double) * (16 * (size_t)h->nb + (size_t)(ra > 2 ? ra : 2) * 4 * 256 + 128);
    bbh_coop2_launch(kdc, a.kind, has_tbl, dim3((unsigned)((N + 15) / 16)), clds, h->stream, ca);
    h->last_form = 3;
    BBH_HIP_TRY(h, hipGetLastError());
    return 0;
  }
  if (with_var) h->last_form = 0;
  if (kdp == 2)
    bbh_fused_launch_kd2(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else if (kdp == 4)
    bbh_fused_launch_kd4(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else if (kdp == 6)
    bbh_fused_launch_kd6(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else if (kdp == 8)
    bbh_fused_launch_kd8(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else if (kdp == 12)
    bbh_fused_launch_kd12(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else if (kdp == 16)
    bbh_fused_launch_kd16(a.kind, has_tbl, grid, block, lds, h->stream, a);
  else
    bbh_fused_launch_kd0(has_tbl, m52, grid, block, lds, h->stream, a);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- unfused verification path ------------------------------------------------------------------
// K(X*, X) materialised with direct-difference distances, then V = K* L^-T through bbh_gemm.
#define BBH_KSTAR_CB 8  // candidates per workgroup
__global__ __launch_bounds__(256) void bbh_kstar_kernel(const double* __restrict__ X, int64_t Nc, int64_t ldx,
                                                        const double* __restrict__ xnT, const int* __restrict__ task,
                                                        const double* __restrict__ theta, const int* __restrict__ numcol,
                                                        const double* __restrict__ lo, const double* __restrict__ hi,
                                                        int n, int64_t np, int dn, const bbh_kern_spec ks, int T,
                                                        int task_col, int64_t ldk, double* __restrict__ Kst) {
  // xnT [dn, np] / task [np]: the points of the columns (training rows, or the pending points with np = 16);
  // Kst[cand * ldk + i], i < np.  A workgroup takes BBH_KSTAR_CB candidates x 256 columns: the candidates' normalised
  // coordinates and the reciprocal lengthscales of every factor are staged in LDS once, every thread reads its column's
  // coordinates once for all of them (as one thread per entry with two fp64 divisions per dimension and factor this kernel
  // was 67 % of a composite-kernel posterior pass).
  extern __shared__ double s_ks[];  // invls [F][dn] | xc [CB][dn]
  double* s_il = s_ks;
  double* s_xc = s_ks + ks.F * dn;
  const int64_t cand0 = (int64_t)blockIdx.y * BBH_KSTAR_CB;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int e = threadIdx.x; e < ks.F * dn; e += 256) s_il[e] = 1.0 / theta[ks.ls_off[e / dn] + e % dn];
  for (int e = threadIdx.x; e < BBH_KSTAR_CB * dn; e += 256) {
    const int64_t cand = cand0 + e / dn;
    const int j = e % dn;
    s_xc[e] = cand < Nc ? (X[cand * ldx + numcol[j]] - lo[j]) / (hi[j] - lo[j]) : 0.0;
  }
  __syncthreads();
  if (i >= np) return;
  double r2[BBH_KSTAR_CB][BBH_MAX_FACTORS];
#pragma unroll
  for (int c = 0; c < BBH_KSTAR_CB; c++)
#pragma unroll
    for (int f = 0; f < BBH_MAX_FACTORS; f++) r2[c][f] = 0.0;
  if (i < n)
    for (int j = 0; j < dn; j++) {
      const double x = xnT[(int64_t)j * np + i];
#pragma unroll
      for (int c = 0; c < BBH_KSTAR_CB; c++) {
        const double xc = s_xc[c * dn + j];
#pragma unroll
        for (int f = 0; f < BBH_MAX_FACTORS; f++)
          if (f < ks.F) r2[c][f] += bbh_metric_term_f(ks, theta, f, j, dn, xc, x, s_il[f * dn + j]);
      }
    }
  const int ti = (T > 1 && i < n) ? task[i] : 0;
#pragma unroll
  for (int c = 0; c < BBH_KSTAR_CB; c++) {
    const int64_t cand = cand0 + c;
    double v = 0.0;
    if (cand < Nc && i < n) {
      v = bbh_kcomp(ks, theta, r2[c]);
      if (ks.use_os) v *= theta[2];
      if (T > 1) {
        int tcand = (int)X[cand * ldx + task_col];
        tcand = tcand < 0 ? 0 : (tcand >= T ? T - 1 : tcand);
        v *= theta[3 + dn + tcand * T + ti];
      }
    }
    Kst[cand * ldk + i] = v;  // rows up to the padded candidate count exist
  }
}

// k(x, x) of every candidate without outer outputscale / task factor (dot-product kernels: not a constant)
__global__ __launch_bounds__(256) void bbh_kdiag_kernel(const double* __restrict__ X, int64_t Nc, int64_t ldx,
                                                        const double* __restrict__ theta, const int* __restrict__ numcol,
                                                        const double* __restrict__ lo, const double* __restrict__ hi, int dn,
                                                        const bbh_kern_spec ks, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= Nc) return;
  double r2[BBH_MAX_FACTORS] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < dn; j++) {
    const double x = (X[i * ldx + numcol[j]] - lo[j]) / (hi[j] - lo[j]);
    for (int f = 0; f < ks.F; f++) r2[f] += bbh_metric_term_f(ks, theta, f, j, dn, x, x, 1.0 / theta[ks.ls_off[f] + j]);
  }
  out[i] = bbh_kcomp(ks, theta, r2);
}

__global__ __launch_bounds__(256) void bbh_rowreduce_kernel(const double* __restrict__ Kst, const double* __restrict__ V,
                                                            const double* __restrict__ alpha,
                                                            const double* __restrict__ X, int64_t ldx, int64_t Nc,
                                                            int64_t np, const double* __restrict__ theta, int dn,
                                                            double prior_base, int T, int task_col, int hoff, double ybar, double ysd,
                                                            double* __restrict__ mean, double* __restrict__ var,
                                                            const double* __restrict__ kdiag, double os) {
  const int lane = threadIdx.x & 63;
  const int64_t cand = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cand >= Nc) return;
  double sm = 0.0, sv = 0.0;
  for (int64_t i = lane; i < np; i += 64) {
    sm = fma(Kst[cand * np + i], alpha[i], sm);
    const double v = V[cand * np + i];
    sv = fma(v, v, sv);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sm += __shfl_down(sm, o, 64);
    sv += __shfl_down(sv, o, 64);
  }
  if (lane == 0) {
    double pv = kdiag ? os * kdiag[cand] : prior_base, mc = theta[1];  // k(x, x) without the task factor
    if (T > 1) {
      int tcand = (int)X[cand * ldx + task_col];
      tcand = tcand < 0 ? 0 : (tcand >= T ? T - 1 : tcand);
      pv *= theta[3 + dn + tcand * T + tcand];
      if (hoff >= 0) mc = theta[hoff + T + tcand];
    }
    if (mean) mean[cand] = ybar + ysd * (mc + sm);
    if (var) var[cand] = ysd * ysd * (pv - sv);
  }
}

// Kst [Ncpad, np] (rows >= Nc zero) and, if V != nullptr, V = Kst L^-T, for one candidate chunk.
static int bbh_unfused_chunk(bbh_handle* h, const double* X_dev, int64_t Nc, int64_t ldx, double* Kst, double* V,
                             const double* d_lo, const double* d_hi) {
  const int64_t np = h->np;
  const int64_t Ncpad = bbh_round_up(Nc, 64);
  dim3 grid((unsigned)((np + 255) / 256), (unsigned)(Ncpad / BBH_KSTAR_CB)), block(256);
  hipLaunchKernelGGL(bbh_kstar_kernel, grid, block, sizeof(double) * h->dn * (BBH_KSTAR_CB + h->F), h->stream, X_dev, Nc, ldx, h->d_xnT, h->d_task, h->d_theta,
                     h->d_numcol, d_lo, d_hi, (int)h->n, np, h->dn, bbh_kern_spec_of(h), h->T, h->desc.task_col, np, Kst);
  if (V) bbh_gemm(h->stream, false, true, Ncpad, np, np, 1.0, Kst, np, 0, h->d_X, np, 0, 0.0, V, np, 0, 1);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// models the fused kernels are not instantiated for: composite kernels, the piecewise-polynomial family
bool bbh_materialised_only(const bbh_handle* h) { return h->F > 1 || h->desc.kernel_kind > BBH_KERNEL_RBF; }

// k(x, x) without the task factor: outputscale * (prod_f | sum_f) os_f
double bbh_prior_base(const bbh_handle* h) {
  const double* th = h->theta.data();
  double v = h->desc.use_outputscale ? th[2] : 1.0;
  if (h->F > 1) {
    const bbh_kern_spec ks = bbh_kern_spec_of(h);
    double u[BBH_MAX_FACTORS] = {1.0, 1.0, 1.0, 1.0};
    for (int f = 0; f < ks.F; f++) u[f] = th[ks.fos_off + f];  // (stationary factors: k_f(x, x) = 1)
    v *= bbh_combine(ks.F, ks.combine, ks.grp, u);
  }
  return v;
}

// ---- composite kernels: the whole posterior through the materialised K* ------------------------------------------
// The fused kernels are specialised for one stationary factor (one distance GEMM, one staged kernel-value evaluation).
// Products / sums of factors (bbh_model_desc.n_factors > 1) take this path for every posterior-shaped call:
//   Kext = [K(X*, X) | K(X*, P)]  [Nc, np + 16]   (bbh_kstar_kernel twice: training columns, pending columns)
//   O    = Kext meanB             [Nc, 1 + p]      column 0: mean sum, columns 1..p: posterior cross-covariances (epilogue)
//   V    = K(X*, X) L^-T          [Nc, np]         variance = prior - |v|^2
// in chunks of 16384 candidates; ~2 np^2 flops per candidate on the generic GEMM plus 16 np bytes of K* traffic.
__global__ __launch_bounds__(256) void bbh_ext_epilogue_kernel(const double* __restrict__ V, const double* __restrict__ Kext,
                                                               int64_t ldk, const double* __restrict__ Bm,
                                                               const double* __restrict__ X, int64_t ldx, int64_t Nc,
                                                               int64_t np, const double* __restrict__ theta, int dn,
                                                               double prior_base, int T, int task_col, int hoff, double ybar,
                                                               double ysd, int p, double* __restrict__ mean,
                                                               double* __restrict__ var, double* __restrict__ cross,
                                                               const double* __restrict__ kdiag, double os) {
  // one wave per candidate: O[c] = sum_i Kext[i] Bm[i][c] for the mean column c = 0 and the pending columns 1..p
  const int lane = threadIdx.x & 63;
  const int64_t cand = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cand >= Nc) return;
  double sv = 0.0, o[16];
#pragma unroll
  for (int c = 0; c < 16; c++) o[c] = 0.0;
  const int ncol = cross ? 1 + p : 1;
  for (int64_t i = lane; i < ldk; i += 64) {
    const double k = Kext[cand * ldk + i];
#pragma unroll
    for (int c = 0; c < 16; c++)
      if (c < ncol) o[c] = fma(k, Bm[i * 16 + c], o[c]);
  }
  if (V && var)
    for (int64_t i = lane; i < np; i += 64) {
      const double v = V[cand * np + i];
      sv = fma(v, v, sv);
    }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
    sv += __shfl_down(sv, sh, 64);
#pragma unroll
    for (int c = 0; c < 16; c++) o[c] += __shfl_down(o[c], sh, 64);
  }
  if (lane == 0) {
    double pv = kdiag ? os * kdiag[cand] : prior_base, mc = theta[1];
    if (T > 1) {
      int tcand = (int)X[cand * ldx + task_col];
      tcand = tcand < 0 ? 0 : (tcand >= T ? T - 1 : tcand);
      pv *= theta[3 + dn + tcand * T + tcand];
      if (hoff >= 0) mc = theta[hoff + T + tcand];
    }
    if (mean) mean[cand] = ybar + ysd * (mc + o[0]);
    if (var) var[cand] = ysd * ysd * (pv - sv);
    if (cross)
#pragma unroll
      for (int c = 1; c < 16; c++)
        if (c <= p) cross[cand * p + (c - 1)] = ysd * ysd * o[c];
  }
}

int bbh_launch_unfused_ext(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev,
                           double* cross_dev) {
  if (N <= 0) return 0;
  if (bbh_is_rff(h)) return bbh_rff_posterior_launch(h, X_dev, N, ldx, mean_dev, var_dev, cross_dev);
  const int64_t np = h->np, ldk = np + 16;
  const int64_t chunk = N < 16384 ? bbh_round_up(N, 64) : 16384;  // (small candidate sets: a small workspace)
  const size_t need = sizeof(double) * ((size_t)chunk * ldk + (size_t)chunk * np + 2 * h->dn + (size_t)chunk);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* Kext = h->d_ws;
  double* V = Kext + chunk * ldk;
  double* d_lo = V + chunk * np;
  double* d_hi = d_lo + h->dn;
  double* kdiag = bbh_has_dot_kind(h) ? d_hi + h->dn : nullptr;  // per-candidate k(x, x): Linear / Polynomial kernels
  const double os_outer = h->desc.use_outputscale ? h->theta[2] : 1.0;
  hipStream_t s = h->stream;
  BBH_HIP_TRY(h, hipMemcpyAsync(d_lo, h->lo.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_hi, h->hi.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, s));
  const bbh_kern_spec ks = bbh_kern_spec_of(h);
  const size_t klds = sizeof(double) * h->dn * (BBH_KSTAR_CB + h->F);
  for (int64_t s0 = 0; s0 < N; s0 += chunk) {
    const int64_t Nc = (N - s0 < chunk) ? N - s0 : chunk;
    const int64_t Ncpad = bbh_round_up(Nc, 64);
    const double* Xc = X_dev + s0 * ldx;
    hipLaunchKernelGGL(bbh_kstar_kernel, dim3((unsigned)((np + 255) / 256), (unsigned)(Ncpad / BBH_KSTAR_CB)), dim3(256), klds, s, Xc, Nc, ldx,
                       h->d_xnT, h->d_task, h->d_theta, h->d_numcol, d_lo, d_hi, (int)h->n, np, h->dn, ks, h->T,
                       h->desc.task_col, ldk, Kext);
    // pending columns (zero when there are none: p = 0 points -> every entry is written as 0)
    hipLaunchKernelGGL(bbh_kstar_kernel, dim3(1, (unsigned)(Ncpad / BBH_KSTAR_CB)), dim3(256), klds, s, Xc, Nc, ldx, h->d_pendT, h->d_taskext + np,
                       h->d_theta, h->d_numcol, d_lo, d_hi, h->p, (int64_t)16, h->dn, ks, h->T, h->desc.task_col, ldk,
                       Kext + np);
    if (var_dev) bbh_gemm(s, false, true, Ncpad, np, np, 1.0, Kext, ldk, 0, h->d_X, np, 0, 0.0, V, np, 0, 1);
    if (kdiag && var_dev)
      hipLaunchKernelGGL(bbh_kdiag_kernel, dim3((unsigned)((Nc + 255) / 256)), dim3(256), 0, s, Xc, Nc, ldx, h->d_theta, h->d_numcol, d_lo,
                         d_hi, h->dn, ks, kdiag);
    hipLaunchKernelGGL(bbh_ext_epilogue_kernel, dim3((unsigned)((Nc + 3) / 4)), dim3(256), 0, s, var_dev ? V : nullptr, Kext, ldk,
                       h->d_meanB, Xc, ldx, Nc, np, h->d_theta, h->dn, bbh_prior_base(h), h->T, h->desc.task_col, bbh_hadamard_offset(h),
                       h->ybar, h->ysd, h->p, mean_dev ? mean_dev + s0 : nullptr, var_dev ? var_dev + s0 : nullptr,
                       cross_dev ? cross_dev + s0 * h->p : nullptr, var_dev ? kdiag : nullptr, os_outer);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  h->last_form = 2;
  return 0;
}

int bbh_launch_unfused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev) {
  if (bbh_is_rff(h)) return bbh_rff_posterior_launch(h, X_dev, N, ldx, mean_dev, var_dev, nullptr);  // (one form only: feature space)
  const int64_t np = h->np;
  const int64_t chunk = 16384;
  const size_t need = sizeof(double) * (2 * (size_t)chunk * np + 2 * h->dn + (size_t)chunk);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* Kst = h->d_ws;
  double* V = Kst + chunk * np;
  double* d_lo = V + chunk * np;
  double* d_hi = d_lo + h->dn;
  double* kdiag = bbh_has_dot_kind(h) ? d_hi + h->dn : nullptr;
  const double os_outer = h->desc.use_outputscale ? h->theta[2] : 1.0;
  BBH_HIP_TRY(h, hipMemcpyAsync(d_lo, h->lo.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, h->stream));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_hi, h->hi.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, h->stream));
  for (int64_t s0 = 0; s0 < N; s0 += chunk) {
    const int64_t Nc = (N - s0 < chunk) ? N - s0 : chunk;
    rc = bbh_unfused_chunk(h, X_dev + s0 * ldx, Nc, ldx, Kst, V, d_lo, d_hi);
    if (rc) return rc;
    if (kdiag)
      hipLaunchKernelGGL(bbh_kdiag_kernel, dim3((unsigned)((Nc + 255) / 256)), dim3(256), 0, h->stream, X_dev + s0 * ldx, Nc, ldx,
                         h->d_theta, h->d_numcol, d_lo, d_hi, h->dn, bbh_kern_spec_of(h), kdiag);
    hipLaunchKernelGGL(bbh_rowreduce_kernel, dim3((unsigned)((Nc + 3) / 4)), dim3(256), 0, h->stream, Kst, V, h->d_alpha,
                       X_dev + s0 * ldx, ldx, Nc, np, h->d_theta, h->dn, bbh_prior_base(h), h->T, h->desc.task_col,
                       bbh_hadamard_offset(h),
                       h->ybar, h->ysd, mean_dev ? mean_dev + s0 : nullptr, var_dev ? var_dev + s0 : nullptr, kdiag, os_outer);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- pending points ---------------------------------------------------------------------------
extern "C" int bbh_pending_set(bbh_handle* h, const double* Xpend_host, int64_t p, double* mean_p_host,
                               double* cov_pp_host) {
  if (!h) return -1;
  if (!h->factorized) {
    h->err = "bbh_pending_set: model not factorised";
    return -1;
  }
  if (p < 0 || p > BBH_MAX_PENDING || (p > 0 && !Xpend_host)) {
    h->err = "bbh_pending_set: 0 <= p <= 15 pending points supported";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (bbh_is_rff(h)) return bbh_rff_pending_set(h, Xpend_host, p, mean_p_host, cov_pp_host);
  hipStream_t s = h->stream;
  const int64_t np = h->np, nb = h->nb;
  const int d = h->desc.d, dn = h->dn, T = h->T;
  h->pend_host.assign(Xpend_host, Xpend_host + p * d);
  // reset mean operand to [alpha | 0] and the pending fragment block to padding
  hipLaunchKernelGGL(bbh_init_meanB_kernel, dim3((unsigned)(((np + 16) * 16 + 255) / 256)), dim3(256), 0, s, h->d_alpha,
                     np, np + 16, h->d_meanB);
  std::vector<double> pn((size_t)p * dn);
  std::vector<int> pt(p, 0);
  for (int64_t j = 0; j < p; j++) {
    for (int c = 0; c < dn; c++)
      pn[j * dn + c] = (Xpend_host[j * d + h->numcol[c]] - h->lo[c]) / (h->hi[c] - h->lo[c]);
    if (h->desc.task_col >= 0 && T > 1) {
      int t = (int)Xpend_host[j * d + h->desc.task_col];
      pt[j] = t < 0 ? 0 : (t >= T ? T - 1 : t);
    }
  }
  std::vector<double> pnT((size_t)dn * 16, 0.0);
  for (int64_t j = 0; j < p; j++)
    for (int c = 0; c < dn; c++) pnT[c * 16 + j] = pn[j * dn + c];
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_pendT, pnT.data(), sizeof(double) * pnT.size(), hipMemcpyHostToDevice, s));
  std::vector<double> tf;
  host_pack_trainfrag(h, pn.data(), p, nb, nb + 1, tf);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_trainfrag + nb * h->kd * 64, tf.data(), sizeof(double) * tf.size(),
                                hipMemcpyHostToDevice, s));
  std::vector<double> tfg;  // (outlives the copies below: the stream is synchronised before this function returns)
  if (h->coopg_ready) {  // generic production (bbh_coopg.h): the pending block of every factor's training fragments
    const bbh_kern_spec ksg = bbh_kern_spec_of(h);
    const int kdg = bbh_coopg_kd(h);
    const int64_t per = (nb + 1) * (int64_t)kdg * 64;
    tfg.reserve((size_t)ksg.F * kdg * 64);
    for (int f = 0; f < ksg.F; f++) {
      std::vector<double> one;
      host_pack_trainfrag_generic(h, ksg, f, kdg, 1, one, p > 0 ? pn.data() : &h->ybar, p);  // (p = 0: any non-null pointer, all padding)
      tfg.insert(tfg.end(), one.begin(), one.end());
    }
    for (int f = 0; f < ksg.F; f++)
      BBH_HIP_TRY(h, hipMemcpyAsync(h->d_trainfrag_f + f * per + nb * kdg * 64, tfg.data() + (size_t)f * kdg * 64, sizeof(double) * kdg * 64,
                                    hipMemcpyHostToDevice, s));
  }
  std::vector<int> te(16, 0);
  for (int64_t j = 0; j < p; j++) te[j] = pt[j];
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_taskext + np, te.data(), sizeof(int) * 16, hipMemcpyHostToDevice, s));
  h->p = (int)p;
  h->nb_ext = nb + (p > 0 ? 1 : 0);
  h->pend_mean.clear();
  h->pend_cov.clear();
  if (p == 0) {
    BBH_HIP_TRY(h, hipStreamSynchronize(s));
    return 0;
  }
  // workspace: raw pending rows [64, d], Kst [64, np], Tm [64, np], betaT [64, np], lo/hi
  const size_t need = sizeof(double) * (64 * (size_t)d + 3 * 64 * (size_t)np + 2 * dn);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* d_xp = h->d_ws;
  double* Kst = d_xp + 64 * d;
  double* Tm = Kst + 64 * np;
  double* betaT = Tm + 64 * np;
  double* d_lo = betaT + 64 * np;
  double* d_hi = d_lo + dn;
  BBH_HIP_TRY(h, hipMemcpyAsync(d_xp, Xpend_host, sizeof(double) * p * d, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_lo, h->lo.data(), sizeof(double) * dn, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_hi, h->hi.data(), sizeof(double) * dn, hipMemcpyHostToDevice, s));
  rc = bbh_unfused_chunk(h, d_xp, p, d, Kst, Tm, d_lo, d_hi);  // Tm[j] = (L^-1 k_Pj)^T
  if (rc) return rc;
  bbh_gemm(s, false, false, 64, np, np, 1.0, Tm, np, 0, h->d_X, np, 0, 0.0, betaT, np, 0, 1);  // beta_j^T = t_j^T L^-1
  hipLaunchKernelGGL(bbh_set_beta_kernel, dim3((unsigned)((np + 16 + 255) / 256)), dim3(256), 0, s, betaT, np, np, (int)p,
                     h->d_meanB);
  // pending posterior (host, O(p^2 n)): mean_P = ybar + ysd (c + K_P alpha), cov_PP = ysd^2 (K_PP - Tm Tm^T)
  std::vector<double> hK((size_t)p * np), hT((size_t)p * np), hal(np);
  BBH_HIP_TRY(h, hipMemcpyAsync(hK.data(), Kst, sizeof(double) * p * np, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(hT.data(), Tm, sizeof(double) * p * np, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(hal.data(), h->d_alpha, sizeof(double) * np, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  const double* th = h->theta.data();
  const double os = h->desc.use_outputscale ? th[2] : 1.0;
  const bbh_kern_spec ks = bbh_kern_spec_of(h);
  h->pend_mean.assign(p, 0.0);
  h->pend_cov.assign((size_t)p * p, 0.0);
  for (int64_t i = 0; i < p; i++) {
    double m = 0.0;
    for (int64_t k = 0; k < np; k++) m += hK[i * np + k] * hal[k];
    h->pend_mean[i] = h->ybar + h->ysd * ((h->hadamard ? th[bbh_hadamard_offset(h) + T + pt[i]] : th[1]) + m);
    for (int64_t j = 0; j < p; j++) {
      double uf[BBH_MAX_FACTORS] = {1.0, 1.0, 1.0, 1.0};
      for (int f = 0; f < ks.F; f++) {
        double r2 = 0.0;
        for (int c = 0; c < dn; c++) r2 += bbh_metric_term_f(ks, th, f, c, dn, pn[i * dn + c], pn[j * dn + c], 1.0 / th[ks.ls_off[f] + c]);
        uf[f] = (ks.F > 1 ? th[ks.fos_off + f] : 1.0) * bbh_kbase(ks.kind[f], r2, ks.jb, ks.alpha_off >= 0 ? th[ks.alpha_off + f] : 1.0);
      }
      const double kc = ks.F > 1 ? bbh_combine(ks.F, ks.combine, ks.grp, uf) : uf[0];
      double kpp = os * kc;
      if (T > 1) kpp *= th[3 + dn + pt[i] * T + pt[j]];
      double dot = 0.0;
      for (int64_t k = 0; k < np; k++) dot += hT[i * np + k] * hT[j * np + k];
      h->pend_cov[i * p + j] = h->ysd * h->ysd * (kpp - dot);
    }
  }
  if (mean_p_host) memcpy(mean_p_host, h->pend_mean.data(), sizeof(double) * p);
  if (cov_pp_host) memcpy(cov_pp_host, h->pend_cov.data(), sizeof(double) * p * p);
  return 0;
}

// =================================================================================================
// qLogNEHVI support: conditional means under S alternative target columns, joint posterior.
// =================================================================================================

// Mean-only variant of the fused kernel with 128 output columns (8 column blocks) per launch:
//   tmat[cand][col0 + c] = ybar + ysd * (mean_const + sum_k K*[cand][k] * Acol[k][col0 + c])
// colfrag layout: [group][k-step][column block (8)][64 lanes], lane l <- Acol[4 ks + (l>>4)][128 g + 16 cb + (l&15)].
template <bool HAS_TBL, int KIND>
__global__ __launch_bounds__(256, 2) void bbh_fused_columns_kernel(const FusedArgs a, const double* __restrict__ colfrag,
                                                                   int64_t col0, int64_t str_c, int64_t str_s, int64_t s_total,
                                                                   double* __restrict__ tmat) {
  // tmat element (candidate i, column s) at i * str_c + s * str_s: candidate-major [N, ldt] (str_c = ldt, str_s = 1) or
  // sample-major [S, N] (str_c = 1, str_s = N); columns >= s_total (padding of the last 128-column group) are not written
  extern __shared__ __attribute__((aligned(16))) double s_cand[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cnd = l & 15, q = l >> 4;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + w) * 16;
  if (tile0 >= a.N) return;
  const int64_t row = (tile0 + cnd < a.N) ? tile0 + cnd : a.N - 1;
  const double* xr = a.X + row * a.ldx;
  double* candw = s_cand + (int64_t)w * a.kd * 64;
  // (loads in independent groups of four k-steps with clamped indices: as `if (dim < dn) v = fma(xr[numcol[dim]], ...)`
  // every k-step was its own divergent block with two dependent memory round trips)
  double nbsum = 0.0;
  for (int k0 = 0; k0 < a.kd; k0 += 4) {
    int xcol[4];
    double xval[4], xscl[4], xofs[4];
#pragma unroll
    for (int u = 0; u < 4; u++) xcol[u] = a.numcol[(4 * (k0 + u) + q < a.dn) ? 4 * (k0 + u) + q : a.dn - 1];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int dimc = (4 * (k0 + u) + q < a.dn) ? 4 * (k0 + u) + q : a.dn - 1;
      xval[u] = xr[xcol[u]];
      xscl[u] = a.scl[dimc];
      xofs[u] = a.ofs[dimc];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u < a.kd) {
        double v = 0.0;
        if (4 * (k0 + u) + q < a.dn) {
          v = fma(xval[u], xscl[u], xofs[u]);
          nbsum = fma(v, v, nbsum);
        }
        candw[(k0 + u) * 64 + l] = v;
      }
    }
  }
  nbsum += __shfl_xor(nbsum, 16, 64);
  nbsum += __shfl_xor(nbsum, 32, 64);
  {
    const int k1 = a.dn >> 2, q1 = a.dn & 3;
    if (q == q1) candw[k1 * 64 + l] = 1.0;
    const int k2 = (a.dn + 1) >> 2, q2 = (a.dn + 1) & 3;
    if (q == q2) candw[k2 * 64 + l] = nbsum;
  }
  int tc = 0;
  if (HAS_TBL && a.task_col >= 0) {
    tc = (int)xr[a.task_col];
    tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
  }
  WaveCtx c;
  c.tf = a.trainfrag + l;
  c.candl = candw + l;
  c.mb = a.meanB + l;
  c.tbl = a.tasktbl;
  c.taskext = a.taskext;
  c.kvc = nullptr;
  c.kvl = (bbh_lds_double*)nullptr;
  c.nl = 0;
  c.ncache = 0;
  c.al = (const bbh_lds_double*)nullptr;
  c.kd = a.kd;
  c.kind = a.kind;
  c.T = a.T;
  c.tc = tc;
  c.q = q;
  c.l = l;
  c.dn = a.dn;
  d4 acc[8];
#pragma unroll
  for (int cb = 0; cb < 8; cb++) acc[cb] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* cf = colfrag + l;
  double kv[4];
  for (int tb = 0; tb < a.nb; tb++) {
    double bq[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int cb = 0; cb < 8; cb++) bq[r][cb] = cf[((int64_t)(4 * tb + r) * 8 + cb) * 64];
    compute_kv<HAS_TBL, KIND>(c, tb, kv);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int cb = 0; cb < 8; cb++) acc[cb] = mfma_f64(kv[r], bq[r][cb], acc[cb]);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int64_t gi = tile0 + q + 4 * r;
    const int tcm = __shfl(tc, q + 4 * r, 64);  // lane m (< 16) holds candidate m's task
    const double mc = (HAS_TBL && a.taskmean) ? a.taskmean[tcm] : a.mean_const;
    if (gi < a.N) {
#pragma unroll
      for (int cb = 0; cb < 8; cb++)
        if (col0 + 16 * cb + cnd < s_total) tmat[gi * str_c + (col0 + 16 * cb + cnd) * str_s] = a.ybar + a.ysd * (mc + acc[cb][r]);
    }
  }
}

// Cooperative variant for S >= 384 alternative columns: one workgroup per tile of 16 candidates, wave w contracts the tile's
// kernel values with column group 4 G + w (128 columns) of a super-group of 512.  The kernel values are the expensive part (a
// distance GEMM and a libm sqrt / exp per value: more pipe time than the 32 column MFMAs of a k-block) and the plain kernel above
// recomputes them for every group of 128 columns; here the four waves take turns - k-block 4 g + w is produced by wave w - and
// exchange them through a double-buffered 8 KB slot in LDS, one barrier per four k-blocks: every value is computed once per
// candidate for 512 columns.
// NT = candidate tiles (of 16) per workgroup: every column fragment that comes back from L2 feeds NT MFMAs (the column matrix is
// re-read by every workgroup: 1.2 MB per 16 NT candidates at n = 288, S = 512 - L2 -> L1 bandwidth, not HBM).
// SM (sample-major output [S, N]): the MFMAs run with swapped operands - the column fragment as A, the kernel values as B, the same
// registers either way - so that the accumulators hold the transposed block (lane = candidate, register = column) and a store
// instruction writes 16 consecutive candidates of a column (128 B) instead of 32 B pieces of 16 columns that lie N doubles apart.
template <bool HAS_TBL, int KIND, int NT, bool SM>
__global__ __launch_bounds__(256, 2) void bbh_coop_columns_kernel(const FusedArgs a, const double* __restrict__ colfrag, int64_t group0,
                                                                  int64_t groups, int64_t nks, int64_t str_c, int64_t str_s,
                                                                  int64_t s_total, double* __restrict__ tmat) {
  extern __shared__ __attribute__((aligned(16))) double s_coopc[];  // candidate fragments [NT][kd][64] | kv [2][NT][4 k-blocks][4][64]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cnd = l & 15, q = l >> 4;
  const int64_t tile0 = (int64_t)blockIdx.x * 16 * NT;
  double* kvx = s_coopc + NT * a.kd * 64;
  int tc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int64_t row = (tile0 + 16 * t + cnd < a.N) ? tile0 + 16 * t + cnd : a.N - 1;
    const double* xr = a.X + row * a.ldx;
    double* candw = s_coopc + t * a.kd * 64;
    if (w == t) {  // tile t's normalised candidate fragments, augmented as in the plain kernel (wave t builds them; NT <= 4)
      double nbsum = 0.0;
      for (int k0 = 0; k0 < a.kd; k0++) {
        double v = 0.0;
        if (4 * k0 + q < a.dn) {
          v = fma(xr[a.numcol[4 * k0 + q]], a.scl[4 * k0 + q], a.ofs[4 * k0 + q]);
          nbsum = fma(v, v, nbsum);
        }
        candw[k0 * 64 + l] = v;
      }
      nbsum += __shfl_xor(nbsum, 16, 64);
      nbsum += __shfl_xor(nbsum, 32, 64);
      const int k1 = a.dn >> 2, q1 = a.dn & 3;
      if (q == q1) candw[k1 * 64 + l] = 1.0;
      const int k2 = (a.dn + 1) >> 2, q2 = (a.dn + 1) & 3;
      if (q == q2) candw[k2 * 64 + l] = nbsum;
    }
    tc[t] = 0;
    if (HAS_TBL && a.task_col >= 0) {
      tc[t] = (int)xr[a.task_col];
      tc[t] = tc[t] < 0 ? 0 : (tc[t] >= a.T ? a.T - 1 : tc[t]);
    }
  }
  __syncthreads();
  WaveCtx c[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    c[t].tf = a.trainfrag + l;
    c[t].candl = s_coopc + t * a.kd * 64 + l;
    c[t].mb = a.meanB + l;
    c[t].tbl = a.tasktbl;
    c[t].taskext = a.taskext;
    c[t].kvc = nullptr;
    c[t].kvl = (bbh_lds_double*)nullptr;
    c[t].nl = 0;
    c[t].ncache = 0;
    c[t].al = (const bbh_lds_double*)nullptr;
    c[t].kd = a.kd;
    c[t].kind = a.kind;
    c[t].T = a.T;
    c[t].tc = tc[t];
    c[t].q = q;
    c[t].l = l;
    c[t].dn = a.dn;
  }
  const bool live = group0 + w < groups;  // (the last super-group may have fewer than four column groups: those waves only produce)
  d4 acc[NT][8];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int cb = 0; cb < 8; cb++) acc[t][cb] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* cf = colfrag + (live ? group0 + w : group0) * nks * 8 * 64 + l;
  const int ngrp = (a.nb + 3) >> 2;
  // Column fragments through a register ring of RD k-steps (8 fragments each), refilled right after the MFMAs that consumed a slot:
  // the loads of k-step s + RD are in flight during the 8 NT (RD - 1) MFMAs in between (the fragments are static data: the ring runs
  // ahead across the group barriers).  As plain loads at the head of every k-block the first MFMA of each block waited out the
  // whole L2 latency: half the pipe time of this loop.
  constexpr int RD = (NT == 1 && KIND >= 0) ? 8 : 4;  // (runtime kernel kinds: the deeper ring spills)
  const int nsteps = 4 * a.nb;
  double ring[RD][8];
  if (live) {
#pragma unroll
    for (int i = 0; i < RD; i++)
      if (i < nsteps) {
#pragma unroll
        for (int cb = 0; cb < 8; cb++) ring[i][cb] = cf[((int64_t)i * 8 + cb) * 64];
      }
  }
  for (int g = 0; g < ngrp; g++) {
    double* slot = kvx + (g & 1) * (NT * 1024);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      double kv[4] = {0.0, 0.0, 0.0, 0.0};
      if (4 * g + w < a.nb) compute_kv<HAS_TBL, KIND>(c[t], 4 * g + w, kv);
#pragma unroll
      for (int r = 0; r < 4; r++) slot[t * 1024 + (w * 4 + r) * 64 + l] = kv[r];
    }
    __syncthreads();  // one barrier per group: the buffer written two groups later is only reached after the next barrier
    if (live) {
#pragma unroll
      for (int blk = 0; blk < 16 / RD; blk++) {
#pragma unroll
        for (int i = 0; i < RD; i++) {
          const int step = 16 * g + blk * RD + i;  // k-step 4 tb + r
          if ((step >> 2) < a.nb) {  // wave-uniform
            double kvv[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) kvv[t] = slot[t * 1024 + ((blk * RD + i) & 15) * 64 + l];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
              for (int cb = 0; cb < 8; cb++)
                acc[t][cb] = SM ? mfma_f64(ring[i][cb], kvv[t], acc[t][cb]) : mfma_f64(kvv[t], ring[i][cb], acc[t][cb]);
            if (step + RD < nsteps) {
#pragma unroll
              for (int cb = 0; cb < 8; cb++) ring[i][cb] = cf[((int64_t)(step + RD) * 8 + cb) * 64];
            }
          }
        }
      }
    }
  }
  if (!live) return;
  const int64_t col0 = 128 * (group0 + w);
  if (SM) {  // acc[t][cb][r]: column col0 + 16 cb + q + 4 r of candidate tile0 + 16 t + cnd
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int64_t gi = tile0 + 16 * t + cnd;
      const double mc = (HAS_TBL && a.taskmean) ? a.taskmean[tc[t]] : a.mean_const;  // (this lane's own row: tc[t] is candidate cnd's task)
      if (gi < a.N) {
#pragma unroll
        for (int cb = 0; cb < 8; cb++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (col0 + 16 * cb + q + 4 * r < s_total) tmat[gi * str_c + (col0 + 16 * cb + q + 4 * r) * str_s] = a.ybar + a.ysd * (mc + acc[t][cb][r]);
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t gi = tile0 + 16 * t + q + 4 * r;
      const int tcm = __shfl(tc[t], q + 4 * r, 64);  // lane m (< 16) holds candidate m's task
      const double mc = (HAS_TBL && a.taskmean) ? a.taskmean[tcm] : a.mean_const;
      if (gi < a.N) {
#pragma unroll
        for (int cb = 0; cb < 8; cb++)
          if (col0 + 16 * cb + cnd < s_total) tmat[gi * str_c + (col0 + 16 * cb + cnd) * str_s] = a.ybar + a.ysd * (mc + acc[t][cb][r]);
      }
    }
}

// colfrag <- alpha columns A [np, spad] (row-major)
__global__ void bbh_pack_colfrag_kernel(const double* __restrict__ A, int64_t spad, int64_t nks, double* __restrict__ out) {
  const int64_t g = blockIdx.z, ks = blockIdx.x, cb = blockIdx.y;
  const int l = threadIdx.x;
  out[((g * nks + ks) * 8 + cb) * 64 + l] = A[(4 * ks + (l >> 4)) * spad + 128 * g + 16 * cb + (l & 15)];
}

// Yc[i][s] = (Y[i][s] - ybar) / ysd - c for real rows and columns, 0 on the padding
__global__ void bbh_prep_columns_kernel(const double* __restrict__ Y, int64_t n, int64_t S, int64_t np, int64_t spad,
                                        double ybar, double ysd, double c, const double* __restrict__ taskmean,
                                        const int* __restrict__ task, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np * spad) return;
  const int64_t i = e / spad, s = e % spad;
  out[e] = (i < n && s < S) ? (Y[i * S + s] - ybar) / ysd - (taskmean ? taskmean[task[i]] : c) : 0.0;
}

// The alpha columns A [np, spad] (device, row-major) become the operand of bbh_posterior_columns: fragments for the fused /
// cooperative kernels, a plain copy for the models that contract the materialised K*.
static int bbh_install_columns(bbh_handle* h, const double* A, int64_t S, int64_t spad) {
  hipStream_t s = h->stream;
  const int64_t np = h->np;
  const int64_t nks = np / 4, groups = spad / 128;
  const int64_t elems = groups * nks * 8 * 64;
  if (!h->d_colfrag || h->colfrag_elems < elems) {
    if (h->d_colfrag) hipFree(h->d_colfrag);
    h->d_colfrag = nullptr;
    BBH_HIP_TRY(h, hipMalloc((void**)&h->d_colfrag, sizeof(double) * elems));
    h->colfrag_elems = elems;
  }
  hipLaunchKernelGGL(bbh_pack_colfrag_kernel, dim3((unsigned)nks, 8, (unsigned)groups), dim3(64), 0, s, A, spad, nks,
                     h->d_colfrag);
  if (bbh_materialised_only(h)) {  // these models contract K* with the plain matrix (bbh_posterior_columns)
    if (!h->d_colA || h->colA_elems < np * spad) {
      if (h->d_colA) hipFree(h->d_colA);
      h->d_colA = nullptr;
      BBH_HIP_TRY(h, hipMalloc((void**)&h->d_colA, sizeof(double) * np * spad));
      h->colA_elems = np * spad;
    }
    BBH_HIP_TRY(h, hipMemcpyAsync(h->d_colA, A, sizeof(double) * np * spad, hipMemcpyDeviceToDevice, s));
  }
  BBH_HIP_TRY(h, hipGetLastError());
  h->ncols = S;
  return 0;
}

extern "C" int bbh_set_mean_columns(bbh_handle* h, const double* Y_host, int64_t S) {
  if (!h) return -1;
  if (!h->factorized || !Y_host || S < 1 || S > 8192) {
    h->err = "bbh_set_mean_columns: model not factorised / bad arguments (1 <= S <= 8192)";
    return -1;
  }
  if (bbh_is_rff(h)) {
    h->err = "bbh_set_mean_columns: conditional-mean columns (qLogNEHVI) are not available with the RFF kernel";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int64_t np = h->np, n = h->n;
  const int64_t spad = bbh_round_up(S, 128);
  const size_t need = sizeof(double) * ((size_t)n * S + 3 * (size_t)np * spad);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* dY = h->d_ws;
  double* Yc = dY + n * S;
  double* T1 = Yc + np * spad;
  double* A = T1 + np * spad;
  BBH_HIP_TRY(h, hipMemcpyAsync(dY, Y_host, sizeof(double) * n * S, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(bbh_prep_columns_kernel, dim3((unsigned)((np * spad + 255) / 256)), dim3(256), 0, s, dY, n, S, np, spad,
                     h->ybar, h->ysd, h->theta[1], h->hadamard ? h->d_theta + bbh_hadamard_offset(h) + h->T : nullptr, h->d_task, Yc);
  bbh_gemm(s, false, false, np, spad, np, 1.0, h->d_X, np, 0, Yc, spad, 0, 0.0, T1, spad, 0, 1);  // L^-1 Yc
  bbh_gemm(s, true, false, np, spad, np, 1.0, h->d_X, np, 0, T1, spad, 0, 0.0, A, spad, 0, 1);    // L^-T (.)
  rc = bbh_install_columns(h, A, S, spad);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipStreamSynchronize(s));  // the workspace may be reused by the next call
  return 0;
}

// t1[i][s] = t_i for the training rows i < n0 (the same in every column), z[s][i - n0] for the nb baseline rows, 0 on the padding
// (cols: where baseline row b's base sample sits in a row of z - null: z is [S, nb])
__global__ void bbh_nehvi_t1_kernel(const double* __restrict__ t, const double* __restrict__ z, int64_t ldz, const int32_t* __restrict__ cols,
                                    int64_t n0, int64_t nb, int64_t S, int64_t np, int64_t spad, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np * spad) return;
  const int64_t i = e / spad, s = e % spad;
  double v = 0.0;
  if (s < S) {
    if (i < n0) v = t[i];
    else if (i < n0 + nb) v = z[s * ldz + (cols ? (int64_t)cols[i - n0] : i - n0)];
  }
  out[e] = v;
}

// Fb[s][b][o] = sign (ybar + ysd (c_b + sum_{k <= n0 + b} L[n0 + b][k] t1[k][s])): baseline row b of the generative form y = c + L [t; z]
__global__ __launch_bounds__(256) void bbh_nehvi_fb_kernel(const double* __restrict__ L, int64_t np, const double* __restrict__ t1, int64_t spad,
                                                           int64_t n0, int64_t nb, int64_t S, double ybar, double ysd, double c,
                                                           const double* __restrict__ taskmean, const int* __restrict__ task, double sign,
                                                           int o, int m, double* __restrict__ Fb) {
  const int64_t b = blockIdx.x, s = (int64_t)blockIdx.y * 256 + threadIdx.x;
  if (s >= S) return;
  const int64_t row = n0 + b;
  const double* lr = L + row * np;
  double acc0 = 0.0, acc1 = 0.0;
  int64_t k = 0;
  for (; k + 1 <= row; k += 2) {
    acc0 = fma(lr[k], t1[k * spad + s], acc0);
    acc1 = fma(lr[k + 1], t1[(k + 1) * spad + s], acc1);
  }
  if (k <= row) acc0 = fma(lr[k], t1[k * spad + s], acc0);
  const double cm = taskmean ? taskmean[task[row]] : c;
  Fb[(s * nb + b) * m + o] = sign * (ybar + ysd * (cm + (acc0 + acc1)));
}

// qLogNEHVI, one target of one selection step, on the device.  h holds the target's model EXTENDED by nb baseline rows as
// noise-free observations (bbh_set_model_ex with a noise mask, rows n - nb .. n - 1; their target values are not read) and is
// factorised.  The joint draw of the baseline values BoTorch takes through a cached Cholesky root is the generative form of
// that factor:  y_ext,s = c + L_ext [t; z_s],  t = L^-1 (y - c) of the training rows, z_s the sample's base samples - the first
// n - nb rows reproduce the measurements, the last nb rows ARE the sample  mu_b + chol(Sigma_b) z_s  of the baseline's joint
// posterior (L_ext's last block row is [K_bn L^-T, chol(Sigma_b)]).  So nothing has to be solved for:
//   * Fb_dev[s][b][o] (strides nb * m, m, 1) = sign * y_ext,s[n - nb + b] in the target's original scale - input of the box
//     decompositions (bbh_cells_build_dev) and of the pruning counts (bbh_pareto_frequency_dev);
//   * want_columns: the S weight columns  A_s = L_ext^-T [t; z_s]  of the model conditioned on that sample
//     (bbh_posterior_columns), without the (n + nb) x S host array bbh_set_mean_columns takes.
// z_host [S, nb].  Asynchronous on the handle's stream.  Reference: baybe/acquisition/_builder.py:301-334 (X_baseline,
// prune_baseline, cache_root), baybe/acquisition/acqfs.py:477-484.
static int bbh_nehvi_samples_impl(bbh_handle* h, const double* z_host, const double* z_dev, int64_t ldz, const int32_t* cols_dev, int64_t S,
                                  int64_t nb, double sign, int32_t o, int32_t m, double* Fb_dev, int32_t want_columns);

extern "C" int bbh_nehvi_samples(bbh_handle* h, const double* z_host, int64_t S, int64_t nb, double sign, int32_t o, int32_t m,
                                 double* Fb_dev, int32_t want_columns) {
  if (!h) return -1;
  return bbh_nehvi_samples_impl(h, z_host, nullptr, nb, nullptr, S, nb, sign, o, m, Fb_dev, want_columns);
}

extern "C" int bbh_nehvi_samples_dev(bbh_handle* h, const double* z_dev, int64_t ld, const int32_t* cols_dev, int64_t S, int64_t nb,
                                     double sign, int32_t o, int32_t m, double* Fb_dev, int32_t want_columns) {
  if (!h) return -1;
  if (!z_dev || !cols_dev || ld < 1) {
    h->err = "bbh_nehvi_samples_dev: bad arguments";
    return -1;
  }
  return bbh_nehvi_samples_impl(h, nullptr, z_dev, ld, cols_dev, S, nb, sign, o, m, Fb_dev, want_columns);
}

static int bbh_nehvi_samples_impl(bbh_handle* h, const double* z_host, const double* z_dev, int64_t ldz, const int32_t* cols_dev, int64_t S,
                                  int64_t nb, double sign, int32_t o, int32_t m, double* Fb_dev, int32_t want_columns) {
  if (!h->factorized || (!z_host && !z_dev) || S < 1 || S > 8192 || nb < 1 || nb >= h->n || o < 0 || o >= m || m > BBH_MAX_OBJECTIVES || !Fb_dev) {
    h->err = "bbh_nehvi_samples: model not factorised / bad arguments (1 <= S <= 8192, 1 <= nb < n, 0 <= o < m <= 4)";
    return -1;
  }
  if (bbh_is_rff(h)) {
    h->err = "bbh_nehvi_samples: not available with the RFF kernel";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int64_t np = h->np, n0 = h->n - nb;
  const int64_t spad = bbh_round_up(S, 128);
  int rc = bbh_ensure_ws(h, sizeof(double) * 2 * (size_t)np * spad);
  if (rc) return rc;
  if (z_host) {
    rc = bbh_upload_z(h, z_host, (size_t)S * nb);
    if (rc) return rc;
  }
  double* T1 = h->d_ws;
  double* A = T1 + np * spad;
  hipLaunchKernelGGL(bbh_nehvi_t1_kernel, dim3((unsigned)((np * spad + 255) / 256)), dim3(256), 0, s, h->d_t, z_host ? h->d_z : z_dev,
                     z_host ? nb : ldz, z_host ? (const int32_t*)nullptr : cols_dev, n0, nb, S, np, spad, T1);
  hipLaunchKernelGGL(bbh_nehvi_fb_kernel, dim3((unsigned)nb, (unsigned)((S + 255) / 256)), dim3(256), 0, s, h->d_K, np, T1, spad, n0, nb, S,
                     h->ybar, h->ysd, h->theta[1], h->hadamard ? h->d_theta + bbh_hadamard_offset(h) + h->T : nullptr, h->d_task, sign,
                     (int)o, (int)m, Fb_dev);
  if (want_columns) {
    bbh_gemm(s, true, false, np, spad, np, 1.0, h->d_X, np, 0, T1, spad, 0, 0.0, A, spad, 0, 1);  // L^-T [t; z]
    rc = bbh_install_columns(h, A, S, spad);
    if (rc) return rc;
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

static void bbh_fill_fused_args(bbh_handle* h, FusedArgs& a, const double* X_dev, int64_t N, int64_t ldx) {
  a.numcol_identity = 0;
  a.X = X_dev;
  a.N = N;
  a.ldx = ldx;
  a.trainfrag = h->d_trainfrag;
  a.rfrag = h->d_rfrag;
  a.meanB = h->d_meanB;
  a.scl = h->d_sclofs;
  a.ofs = h->d_sclofs + h->dn;
  a.numcol = h->d_numcol;
  a.tasktbl = h->d_tasktbl;
  a.taskmean = h->hadamard ? h->d_theta + bbh_hadamard_offset(h) + h->T : nullptr;
  a.taskext = h->d_taskext;
  a.mean = nullptr;
  a.var = nullptr;
  a.cross = nullptr;
  a.pass_off = h->d_pass_off;
  a.pass_w = h->d_pass_w;
  a.npass = h->npass;
  a.kind = h->desc.kernel_kind;
  a.dn = h->dn;
  a.kd = h->kd;
  a.nb = (int)h->nb;
  a.nb_ext = (int)h->nb;
  a.task_col = h->desc.task_col;
  a.T = h->T;
  a.p = 0;
  a.with_var = 0;
  a.ybar = h->ybar;
  a.ysd = h->ysd;
  a.mean_const = h->theta[1];
  a.prior_scale = h->desc.use_outputscale ? h->theta[2] : 1.0;
  a.qz = nullptr;
  a.qS = 0;
  a.q_best_f = 0.0;
  a.q_sign = 1.0;
  a.q_alive = nullptr;
  a.q_scores = nullptr;
  a.kvcache = nullptr;
  a.slab_flags = nullptr;
  a.nslab = 0;
  a.nxcc = 1;
  a.ncache = 0;
  a.nblk = (N + 63) / 64;
}

// tmat[i][c] = ybar + ysd * (mean_const(task of i) + P[i][c]) for the real columns of a padded product P [Nc, spad]
__global__ void bbh_columns_affine_kernel(const double* __restrict__ P, int64_t spad, int64_t Nc, int64_t S,
                                          const double* __restrict__ X, int64_t ldx, const double* __restrict__ theta,
                                          int T, int task_col, int hoff, double ybar, double ysd, double* __restrict__ tmat,
                                          int64_t str_c, int64_t str_s) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Nc * S) return;
  const int64_t i = e / S, c = e % S;
  double mc = theta[1];
  if (hoff >= 0) {
    int t = (int)X[i * ldx + task_col];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    mc = theta[hoff + T + t];
  }
  tmat[i * str_c + c * str_s] = ybar + ysd * (mc + P[i * spad + c]);
}

// composite kernels: conditional means under the alternative target columns as K* A on the generic GEMM
static int bbh_columns_unfused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev, bool sample_major) {
  const int64_t np = h->np, S = h->ncols, spad = bbh_round_up(S, 128);
  const int64_t chunk = 8192;
  const size_t need = sizeof(double) * ((size_t)chunk * np + (size_t)chunk * spad + 2 * h->dn);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* Kst = h->d_ws;
  double* P = Kst + chunk * np;
  double* d_lo = P + chunk * spad;
  double* d_hi = d_lo + h->dn;
  hipStream_t s = h->stream;
  BBH_HIP_TRY(h, hipMemcpyAsync(d_lo, h->lo.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_hi, h->hi.data(), sizeof(double) * h->dn, hipMemcpyHostToDevice, s));
  for (int64_t s0 = 0; s0 < N; s0 += chunk) {
    const int64_t Nc = (N - s0 < chunk) ? N - s0 : chunk;
    const int64_t Ncpad = bbh_round_up(Nc, 64);
    rc = bbh_unfused_chunk(h, X_dev + s0 * ldx, Nc, ldx, Kst, nullptr, d_lo, d_hi);
    if (rc) return rc;
    bbh_gemm(s, false, false, Ncpad, spad, np, 1.0, Kst, np, 0, h->d_colA, spad, 0, 0.0, P, spad, 0, 1);
    hipLaunchKernelGGL(bbh_columns_affine_kernel, dim3((unsigned)((Nc * S + 255) / 256)), dim3(256), 0, s, P, spad, Nc, S,
                       X_dev + s0 * ldx, ldx, h->d_theta, h->T, h->desc.task_col, bbh_hadamard_offset(h), h->ybar, h->ysd,
                       sample_major ? tmat_dev + s0 : tmat_dev + s0 * S, sample_major ? (int64_t)1 : S, sample_major ? N : (int64_t)1);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

static int bbh_posterior_columns_impl(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev, bool sample_major) {
  if (!h) return -1;
  if (!h->factorized || h->ncols < 1 || !h->d_colfrag || !tmat_dev || N < 0 || (N > 0 && !X_dev) || ldx < h->desc.d) {
    h->err = "bbh_posterior_columns: call bbh_set_mean_columns first / bad arguments";
    return -1;
  }
  if (N == 0) return 0;
  if (bbh_is_rff(h)) {
    h->err = "bbh_posterior_columns: not available with the RFF kernel";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  bbh_timed_scope timed(h, BBH_TIMED_COLUMNS);
  if (bbh_materialised_only(h)) return bbh_columns_unfused(h, X_dev, N, ldx, tmat_dev, sample_major);
  FusedArgs a;
  bbh_fill_fused_args(h, a, X_dev, N, ldx);
  const bool has_tbl = (h->T > 1) || h->desc.use_outputscale;
  const bool m52 = (a.kind == BBH_KERNEL_MATERN52);
  const size_t lds = sizeof(double) * 4 * h->kd * 64;
  dim3 grid((unsigned)((N + 63) / 64)), block(256);
  const int64_t S = h->ncols, spad = bbh_round_up(S, 128), groups = spad / 128;
  const int64_t nks = h->np / 4;
  const int64_t str_c = sample_major ? 1 : S, str_s = sample_major ? N : 1;
  const char* cc_env = getenv("BBH_COLUMNS_COOP");  // A/B switch: 0 keeps the plain kernel for every column group
  const bool coop_cols = !(cc_env && cc_env[0] == '0');
  int64_t g = 0;
  // super-groups of (up to) four column groups on the cooperative kernel while at least three remain (with fewer, three of its
  // four waves would only produce kernel values: the plain kernel's four tiles per workgroup are the better use of the CU)
  const char* nt_env = getenv("BBH_COLUMNS_NT");
  const int cnt = nt_env ? atoi(nt_env) : 2;
  for (; coop_cols && groups - g >= 3; g += 4) {
    auto launch = [&](auto nt, auto sm) {
      constexpr int NT = decltype(nt)::value;
      constexpr bool SM = decltype(sm)::value;
      dim3 cgrid((unsigned)((N + 16 * NT - 1) / (16 * NT)));
      const size_t clds = sizeof(double) * ((size_t)NT * h->kd * 64 + 2 * NT * 1024);
      if (has_tbl && m52)
        hipLaunchKernelGGL((bbh_coop_columns_kernel<true, BBH_KERNEL_MATERN52, NT, SM>), cgrid, block, clds, h->stream, a, h->d_colfrag, g, groups, nks, str_c, str_s, S, tmat_dev);
      else if (has_tbl)
        hipLaunchKernelGGL((bbh_coop_columns_kernel<true, -1, NT, SM>), cgrid, block, clds, h->stream, a, h->d_colfrag, g, groups, nks, str_c, str_s, S, tmat_dev);
      else if (m52)
        hipLaunchKernelGGL((bbh_coop_columns_kernel<false, BBH_KERNEL_MATERN52, NT, SM>), cgrid, block, clds, h->stream, a, h->d_colfrag, g, groups, nks, str_c, str_s, S, tmat_dev);
      else
        hipLaunchKernelGGL((bbh_coop_columns_kernel<false, -1, NT, SM>), cgrid, block, clds, h->stream, a, h->d_colfrag, g, groups, nks, str_c, str_s, S, tmat_dev);
    };
    const bool two = !(cnt == 1 || !m52);  // (runtime kernel kinds: two tiles spill)
    if (two && sample_major) launch(std::integral_constant<int, 2>{}, std::true_type{});
    else if (two) launch(std::integral_constant<int, 2>{}, std::false_type{});
    else if (sample_major) launch(std::integral_constant<int, 1>{}, std::true_type{});
    else launch(std::integral_constant<int, 1>{}, std::false_type{});
  }
  for (; g < groups; g++) {  // the last group may be partially padding: its columns >= S are not written
    const double* cf = h->d_colfrag + g * nks * 8 * 64;
    const int64_t col0 = 128 * g;
    if (has_tbl && m52)
      hipLaunchKernelGGL((bbh_fused_columns_kernel<true, BBH_KERNEL_MATERN52>), grid, block, lds, h->stream, a, cf, col0, str_c, str_s, S, tmat_dev);
    else if (has_tbl)
      hipLaunchKernelGGL((bbh_fused_columns_kernel<true, -1>), grid, block, lds, h->stream, a, cf, col0, str_c, str_s, S, tmat_dev);
    else if (m52)
      hipLaunchKernelGGL((bbh_fused_columns_kernel<false, BBH_KERNEL_MATERN52>), grid, block, lds, h->stream, a, cf, col0, str_c, str_s, S, tmat_dev);
    else
      hipLaunchKernelGGL((bbh_fused_columns_kernel<false, -1>), grid, block, lds, h->stream, a, cf, col0, str_c, str_s, S, tmat_dev);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

extern "C" int bbh_posterior_columns(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev) {
  return bbh_posterior_columns_impl(h, X_dev, N, ldx, tmat_dev, false);
}

extern "C" int bbh_posterior_columns_sm(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev) {
  return bbh_posterior_columns_impl(h, X_dev, N, ldx, tmat_dev, true);
}

// prior covariance among q points given as raw rows (normalised on the fly), direct differences
__global__ __launch_bounds__(256) void bbh_kqq_kernel(const double* __restrict__ Xq, int64_t q, int64_t qpad, int64_t ldx,
                                                      const double* __restrict__ theta, const int* __restrict__ numcol,
                                                      const double* __restrict__ lo, const double* __restrict__ hi, int dn,
                                                      const bbh_kern_spec ks, int T, int task_col, double* __restrict__ Kqq) {
  const int64_t a = blockIdx.y;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= qpad) return;
  double v = 0.0;
  if (a < q && b < q) {
    double r2[BBH_MAX_FACTORS] = {0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < dn; j++) {
      const double rng = hi[j] - lo[j];
      const double xa = (Xq[a * ldx + numcol[j]] - lo[j]) / rng, xb = (Xq[b * ldx + numcol[j]] - lo[j]) / rng;
      for (int f = 0; f < ks.F; f++) r2[f] += bbh_metric_term_f(ks, theta, f, j, dn, xa, xb, 1.0 / theta[ks.ls_off[f] + j]);
    }
    v = bbh_kcomp(ks, theta, r2);
    if (ks.use_os) v *= theta[2];
    if (T > 1) {
      int ta = (int)Xq[a * ldx + task_col], tb = (int)Xq[b * ldx + task_col];
      ta = ta < 0 ? 0 : (ta >= T ? T - 1 : ta);
      tb = tb < 0 ? 0 : (tb >= T ? T - 1 : tb);
      v *= theta[3 + dn + ta * T + tb];
    }
  }
  Kqq[a * qpad + b] = v;
}

extern "C" int bbh_posterior_joint(bbh_handle* h, const double* Xq_host, int64_t q, double* mean_host, double* cov_host) {
  if (!h) return -1;
  if (!h->factorized || !Xq_host || q < 1 || q > 4096 || !mean_host || !cov_host) {
    h->err = "bbh_posterior_joint: model not factorised / bad arguments (1 <= q <= 4096)";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (bbh_is_rff(h)) return bbh_rff_posterior_joint(h, Xq_host, q, mean_host, cov_host);
  hipStream_t s = h->stream;
  const int64_t np = h->np, d = h->desc.d, dn = h->dn;
  const int64_t qpad = bbh_round_up(q, 64);
  const size_t need = sizeof(double) * ((size_t)qpad * d + 2 * (size_t)qpad * np + (size_t)qpad * qpad + qpad + 2 * dn);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* dX = h->d_ws;
  double* Kst = dX + qpad * d;
  double* Tm = Kst + qpad * np;
  double* Kqq = Tm + qpad * np;
  double* dm = Kqq + qpad * qpad;
  double* d_lo = dm + qpad;
  double* d_hi = d_lo + dn;
  BBH_HIP_TRY(h, hipMemcpyAsync(dX, Xq_host, sizeof(double) * q * d, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_lo, h->lo.data(), sizeof(double) * dn, hipMemcpyHostToDevice, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(d_hi, h->hi.data(), sizeof(double) * dn, hipMemcpyHostToDevice, s));
  rc = bbh_unfused_chunk(h, dX, q, d, Kst, Tm, d_lo, d_hi);
  if (rc) return rc;
  hipLaunchKernelGGL(bbh_kqq_kernel, dim3((unsigned)((qpad + 255) / 256), (unsigned)qpad), dim3(256), 0, s, dX, q, qpad, d,
                     h->d_theta, h->d_numcol, d_lo, d_hi, (int)dn, bbh_kern_spec_of(h), h->T, h->desc.task_col, Kqq);
  bbh_gemm(s, false, true, qpad, qpad, np, -1.0, Tm, np, 0, Tm, np, 0, 1.0, Kqq, qpad, 0, 1);  // Kqq - Tm Tm^T
  bbh_matvec(s, Kst, np, qpad, np, h->d_alpha, dm);
  std::vector<double> hc((size_t)qpad * qpad), hm(qpad);
  BBH_HIP_TRY(h, hipMemcpyAsync(hc.data(), Kqq, sizeof(double) * qpad * qpad, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(hm.data(), dm, sizeof(double) * qpad, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  const double s2 = h->ysd * h->ysd;
  for (int64_t i = 0; i < q; i++) {
    double mc = h->theta[1];
    if (h->hadamard) {
      int t = (int)Xq_host[i * d + h->desc.task_col];
      t = t < 0 ? 0 : (t >= h->T ? h->T - 1 : t);
      mc = h->theta[bbh_hadamard_offset(h) + h->T + t];
    }
    mean_host[i] = h->ybar + h->ysd * (mc + hm[i]);
    for (int64_t j = 0; j < q; j++) cov_host[i * q + j] = s2 * 0.5 * (hc[i * qpad + j] + hc[j * qpad + i]);
  }
  return 0;
}
