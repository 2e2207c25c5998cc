// gfx950 (MI355X, CDNA4) kernels of libmembrane_hip.so.
//
// Execution model.  One 256-thread workgroup (4 wave64) per TILE.  A tile owns
// T consecutive vertices (patch order) and lists every facet touching one of
// them plus the non-owned ("halo") vertices those facets reference.  The
// workgroup
//   1. stages the owned vertex rows with fully coalesced flat loads and the
//      halo rows with a gather into an LDS patch (SoA, conflict-free columns),
//   2. walks its facet list (8-byte packed local corner slots, coalesced),
//      reading corners from LDS and doing all fp64 arithmetic in registers,
//   3. accumulates per-vertex sums for its OWNED vertices with ds_add_f64 LDS
//      atomics (halo corners are dropped: their owner tile recomputes them),
//   4. writes the owned rows back with plain coalesced stores and one partial
//      per reduction slot -- no global atomics anywhere, no second pass.
// The path is HBM/LDS bound gather+scatter fp64 work: no MFMA on purpose.
//
// Each device routine cites the reference code it restates (paths relative to
// the reference checkout).
#include <hip/hip_runtime.h>

#include "ms_internal.h"

namespace ms {

// ---------------------------------------------------------------------------
// small fp64 3-vector helpers
// ---------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }

constexpr int BLOCK = 256;
constexpr int NXCD = 8;

// XCD-aware block -> tile map: workgroups are dealt round-robin over the 8
// XCDs, so blocks b and b+8 share an L2.  Give each XCD one CONTIGUOUS range
// of tiles so neighbouring tiles' halo rows hit the same L2 (speed only).
__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int k = b % NXCD, j = b / NXCD;
  const int q = nb / NXCD, r = nb % NXCD;
  return k * q + (k < r ? k : r) + j;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// op: 0 sum, 1 min, 2 max.  red = 4 doubles of LDS scratch.  Result valid in thread 0.
__device__ __forceinline__ double block_reduce(double v, int op, double* red) {
  v = op == 0 ? wave_sum(v) : (op == 1 ? wave_min(v) : wave_max(v));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = red[0];
    for (int i = 1; i < BLOCK / 64; ++i)
      r = op == 0 ? r + red[i] : (op == 1 ? fmin(r, red[i]) : fmax(r, red[i]));
    v = r;
  }
  return v;
}

// mixed-Voronoi corner areas with the reference's sequential-overwrite obtuse
// logic (fortran_kernels/tilt_kernels.f90:160-181, geometry/curvature.py:299-314)
__device__ __forceinline__ void corner_areas(double c0, double c1, double c2, double l0,
                                             double l1, double l2, double tri_area,
                                             double& a0, double& a1, double& a2) {
  const bool o0 = c0 < 0.0, o1 = c1 < 0.0, o2 = c2 < 0.0;
  if (!(o0 || o1 || o2)) {
    a0 = (l1 * c1 + l2 * c2) / 8.0;
    a1 = (l2 * c2 + l0 * c0) / 8.0;
    a2 = (l0 * c0 + l1 * c1) / 8.0;
  } else {
    a0 = a1 = a2 = 0.0;
    if (o0) a0 = tri_area / 2.0;
    if (o1 || o2) a0 = tri_area / 4.0;
    if (o1) a1 = tri_area / 2.0;
    if (o0 || o2) a1 = tri_area / 4.0;
    if (o2) a2 = tri_area / 2.0;
    if (o0 || o1) a2 = tri_area / 4.0;
  }
}

// fortran_kernels/bending_kernels.f90:32-74 (geometry/bending_derivatives.py:48-79)
__device__ __forceinline__ void grad_cotan(V3 u, V3 v, V3& gu, V3& gv) {
  const double C = dot(u, v);
  const V3 w = cross(u, v);
  const double S = norm(w);
  if (S <= 1.0e-15) {
    gu = mk(0, 0, 0);
    gv = mk(0, 0, 0);
    return;
  }
  const double invS = 1.0 / S;
  const double invS3 = 1.0 / (S * S * S);
  const V3 vxw = cross(v, w), wxu = cross(w, u);
  const double k = C * invS3;
  gu = mk(v.x * invS - k * vxw.x, v.y * invS - k * vxw.y, v.z * invS - k * vxw.z);
  gv = mk(u.x * invS - k * wxu.x, u.y * invS - k * wxu.y, u.z * invS - k * wxu.z);
}

// geometry/bending_derivatives.py:82-102
__device__ __forceinline__ void grad_triangle_area(V3 u, V3 v, V3& gu, V3& gv) {
  const V3 w = cross(u, v);
  const double S = norm(w);
  if (!(S > 1.0e-15)) {
    gu = mk(0, 0, 0);
    gv = mk(0, 0, 0);
    return;
  }
  const double invS = 1.0 / S;
  const V3 vxw = cross(v, w), wxu = cross(w, u);
  gu = mk(0.5 * vxw.x * invS, 0.5 * vxw.y * invS, 0.5 * vxw.z * invS);
  gv = mk(0.5 * wxu.x * invS, 0.5 * wxu.y * invS, 0.5 * wxu.z * invS);
}

__device__ __forceinline__ V3 lds_v3(const double* base, int cap, int slot) {
  return mk(base[slot], base[cap + slot], base[2 * cap + slot]);
}
__device__ __forceinline__ void lds_add3(double* base, int stride, int slot, V3 v) {
  atomicAdd(&base[slot], v.x);
  atomicAdd(&base[stride + slot], v.y);
  atomicAdd(&base[2 * stride + slot], v.z);
}

// ---------------------------------------------------------------------------
// K_A: energy pass.
//   scalars (owner facets): E_surface (surface_energy.f90:61-78), body volume
//   (geometry/body.py:104-123), min edge^2 (runtime/topology.py:174-199),
//   normal-rotation guard (runtime/topology.py:13-48);
//   BEND: per-vertex K, A_vor (tilt_kernels.f90:88-190), A_eff
//   (bending_utils.py:37-171), normal sums (bending_utils.py:13-34), then the
//   per-vertex density / back-prop factors of bending.py:117-161 in the
//   epilogue (owned rows are complete in LDS there).
//   Evaluates at x + alpha*d when d != nullptr (trial[movable] = base + alpha d,
//   line_search.py:362-368) and can write that trial row-block to xt.
// LDS: px[3][cap] | (GUARD) ox[3][cap] | (BEND) acc[8][T] | red[4] | fl[cap] bytes
// ---------------------------------------------------------------------------
template <bool BEND, bool GUARD>
__global__ __launch_bounds__(BLOCK) void k_energy(EnergyArgs a, int cap) {
  extern __shared__ double lds[];
  const int T = a.m.T;
  double* px = lds;
  double* ox = px + 3 * cap;
  double* acc = ox + (GUARD ? 3 * cap : 0);
  double* red = acc + (BEND ? 8 * T : 0);
  uint8_t* lfl = reinterpret_cast<uint8_t*>(red + 4);

  const int tile = a.tile0 + xcd_tile(blockIdx.x, a.tile1 - a.tile0);
  const int tid = threadIdx.x;
  const int v_lo = tile * T;
  const int n_owned = min(T, a.m.nv - v_lo);
  const int h0 = a.m.tile_halo_off[tile];
  const int nh = a.m.tile_halo_off[tile + 1] - h0;
  const bool have_d = a.d != nullptr;

  // -- stage owned rows: flat, fully coalesced ------------------------------
  for (int j = tid; j < 3 * n_owned; j += BLOCK) {
    const int r = j / 3, c = j - 3 * r;
    const size_t g = 3 * (size_t)v_lo + j;
    const double xo = a.x[g];
    double xv = xo;
    if (have_d && !(a.m.vflags[v_lo + r] & VF_FIXED)) xv = xo + a.alpha * a.d[g];
    px[c * cap + r] = xv;
    if (GUARD) ox[c * cap + r] = xo;
    if (a.xt) a.xt[g] = xv;
  }
  for (int r = tid; r < n_owned; r += BLOCK) lfl[r] = a.m.vflags[v_lo + r];
  // -- gather halo rows ------------------------------------------------------
  for (int h = tid; h < nh; h += BLOCK) {
    const int v = a.m.halo_ids[h0 + h];
    const uint8_t fl = a.m.vflags[v];
    const int s = n_owned + h;
    lfl[s] = fl;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double xo = a.x[3 * (size_t)v + c];
      double xv = xo;
      if (have_d && !(fl & VF_FIXED)) xv = xo + a.alpha * a.d[3 * (size_t)v + c];
      px[c * cap + s] = xv;
      if (GUARD) ox[c * cap + s] = xo;
    }
  }
  if (BEND)
    for (int j = tid; j < 8 * T; j += BLOCK) acc[j] = 0.0;
  __syncthreads();

  double e_surf = 0.0, vol = 0.0, min_e2 = 1.0e300, guard = 0.0;
  const bool want_surf = a.modules & MS_MOD_SURFACE;
  const bool want_vol = a.modules & (MS_MOD_VOLUME_PENALTY | MS_CON_VOLUME | MS_TRACK_VOLUME);
  const int f0 = a.m.tile_facet_off[tile], f1 = a.m.tile_facet_off[tile + 1];
  for (int p = f0 + tid; p < f1; p += BLOCK) {
    const TileFacet tf = a.m.tile_facets[p];
    const bool owner = tf.flags & TF_OWNER;
    if (!BEND && !owner) continue;
    const V3 v0 = lds_v3(px, cap, tf.l0), v1 = lds_v3(px, cap, tf.l1), v2 = lds_v3(px, cap, tf.l2);
    const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
    const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
    // n = (v1-v0) x (v2-v0); (v2-v0) == -(v0-v2) exactly
    const V3 n = cross(e2, -e1);
    const double A2 = norm(n);
    if (owner) {
      if (want_surf && A2 >= 1.0e-12) e_surf += a.m.tf_gamma[p] * (0.5 * A2);
      if (want_vol && (tf.flags & TF_BODY)) vol += dot(cross(v1, v2), v0);
      min_e2 = fmin(min_e2, fmin(l0, fmin(l1, l2)));
      if (GUARD) {
        const V3 o0 = lds_v3(ox, cap, tf.l0), o1 = lds_v3(ox, cap, tf.l1), o2 = lds_v3(ox, cap, tf.l2);
        const V3 no = cross(o1 - o0, o2 - o0);
        const double nno = norm(no);
        if (nno > 1.0e-12) {
          if (A2 < 1.0e-12) {
            guard = 1.0;
          } else {
            double dd = dot(mk(no.x / nno, no.y / nno, no.z / nno), mk(n.x / A2, n.y / A2, n.z / A2));
            dd = fmin(1.0, fmax(-1.0, dd));
            if (!(acos(dd) <= 0.5)) guard = 1.0;
          }
        }
      }
    }
    if (BEND) {
      const bool in0 = tf.l0 < n_owned, in1 = tf.l1 < n_owned, in2 = tf.l2 < n_owned;
      // tilt_kernels.f90:133-151
      const V3 cr = cross(e1, e2);
      double ad = norm(cr);
      if (ad < 1.0e-12) ad = 1.0e-12;
      const double tri_area = 0.5 * ad;
      const double c0 = dot(-e1, e2) / ad, c1 = dot(-e2, e0) / ad, c2 = dot(-e0, e1) / ad;
      double va0, va1, va2;
      corner_areas(c0, c1, c2, l0, l1, l2, tri_area, va0, va1, va2);
      // bending_utils.py:85-119 (area recomputed from n, clamped at 1e-12)
      double ta_eff = 0.5 * A2;
      if (ta_eff < 1.0e-12) ta_eff = 1.0e-12;
      double ve0, ve1, ve2;
      corner_areas(c0, c1, c2, l0, l1, l2, ta_eff, ve0, ve1, ve2);
      // bending_utils.py:121-153 boundary -> interior redistribution
      const int b0 = (lfl[tf.l0] & VF_BOUNDARY) ? 1 : 0, b1 = (lfl[tf.l1] & VF_BOUNDARY) ? 1 : 0,
                b2 = (lfl[tf.l2] & VF_BOUNDARY) ? 1 : 0;
      const int n_int = 3 - (b0 + b1 + b2);
      if (n_int > 0 && n_int < 3) {
        const double b_sum = ve0 * b0 + ve1 * b1 + ve2 * b2;
        const double extra = b_sum / (double)n_int;
        const double m0 = b0 ? 0.0 : 1.0, m1 = b1 ? 0.0 : 1.0, m2 = b2 ? 0.0 : 1.0;
        ve0 = ve0 * m0 + m0 * extra;
        ve1 = ve1 * m1 + m1 * extra;
        ve2 = ve2 * m2 + m2 * extra;
      }
      if (in0) {
        lds_add3(acc, T, tf.l0, 0.5 * (c1 * (-e1) + c2 * e2));
        atomicAdd(&acc[3 * T + tf.l0], va0);
        atomicAdd(&acc[4 * T + tf.l0], ve0);
        lds_add3(acc + 5 * T, T, tf.l0, n);
      }
      if (in1) {
        lds_add3(acc, T, tf.l1, 0.5 * (c2 * (-e2) + c0 * e0));
        atomicAdd(&acc[3 * T + tf.l1], va1);
        atomicAdd(&acc[4 * T + tf.l1], ve1);
        lds_add3(acc + 5 * T, T, tf.l1, n);
      }
      if (in2) {
        lds_add3(acc, T, tf.l2, 0.5 * (c0 * (-e0) + c1 * e1));
        atomicAdd(&acc[3 * T + tf.l2], va2);
        atomicAdd(&acc[4 * T + tf.l2], ve2);
        lds_add3(acc + 5 * T, T, tf.l2, n);
      }
    }
  }

  double e_bend = 0.0;
  if (BEND) {
    __syncthreads();
    // modules/energy/bending.py:111-161 per-vertex pass on the owned rows
    for (int i = tid; i < n_owned; i += BLOCK) {
      const int v = v_lo + i;
      const V3 K = mk(acc[i], acc[T + i], acc[2 * T + i]);
      const double Avor = acc[3 * T + i], Aeff = acc[4 * T + i];
      const V3 N = mk(acc[5 * T + i], acc[6 * T + i], acc[7 * T + i]);
      const double kappa = a.m.kappa[v], c0 = a.m.c0[v];
      const bool interior = !(lfl[i] & VF_BOUNDARY);
      const double safe = fmax(Avor, 1.0e-12);
      const double k_mag = norm(K);
      const double H = k_mag / (2.0 * safe);
      const double ratio = safe > 1.0e-15 ? Aeff / safe : 0.0;
      double scale_K, fe, fv;
      if (a.bending_model == MS_BEND_HELFRICH) {
        double term = (2.0 * H) - c0;
        if (!interior) term = 0.0;
        e_bend += kappa * (term * term) * Aeff;
        scale_K = kappa * term * ratio;
        fe = 0.5 * kappa * (term * term);
        fv = -2.0 * kappa * term * ratio * H;
      } else {
        const double He = interior ? H : 0.0;
        e_bend += kappa * (He * He) * Aeff;
        scale_K = kappa * He * ratio;
        fe = kappa * (He * He);
        fv = -2.0 * kappa * (He * He) * ratio;
      }
      if (a.fK) {
        V3 Kd;
        if (k_mag > 1.0e-15) {
          Kd = mk(K.x / k_mag, K.y / k_mag, K.z / k_mag);
        } else {
          const double nn = norm(N);
          Kd = nn > 1.0e-15 ? mk(N.x / nn, N.y / nn, N.z / nn) : N;
        }
        a.fK[3 * (size_t)v] = Kd.x * scale_K;
        a.fK[3 * (size_t)v + 1] = Kd.y * scale_K;
        a.fK[3 * (size_t)v + 2] = Kd.z * scale_K;
        a.fA[2 * (size_t)v] = fe;
        a.fA[2 * (size_t)v + 1] = fv;
      }
    }
    if (a.bending_model == MS_BEND_HELFRICH) e_bend *= 0.5;
  }

  double* out = a.partials + tile;   // slot-major: partials[slot][n_tiles]
  const size_t ps = (size_t)a.m.n_tiles;
  double r;
  r = block_reduce(e_surf, 0, red);
  if (tid == 0) out[MS_S_ESURF * ps] = r;
  r = block_reduce(vol, 0, red);
  if (tid == 0) out[MS_S_VOL * ps] = r;
  r = block_reduce(e_bend, 0, red);
  if (tid == 0) out[MS_S_EBEND * ps] = r;
  r = block_reduce(min_e2, 1, red);
  if (tid == 0) out[MS_S_MINEDGE2 * ps] = r;
  r = block_reduce(guard, 2, red);
  if (tid == 0) out[MS_S_GUARD * ps] = r;
}

size_t energy_lds_bytes(int T, int cap, bool bend, bool guard) {
  size_t d = 3 * (size_t)cap + (guard ? 3 * (size_t)cap : 0) + (bend ? 8 * (size_t)T : 0) + 4;
  return d * sizeof(double) + (((size_t)cap + 15) / 16) * 16;
}

template <typename K>
static hipError_t ensure_lds(K kernel, size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_energy(const EnergyArgs& a, bool guard, int cap, hipStream_t s) {
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const bool bend = (a.modules & MS_MOD_BENDING) != 0;
  const size_t lds = energy_lds_bytes(a.m.T, cap, bend, guard);
  hipError_t e;
#define MS_LAUNCH_E(B, G)                                                  \
  do {                                                                     \
    e = ensure_lds(k_energy<B, G>, lds);                                   \
    if (e != hipSuccess) return e;                                         \
    hipLaunchKernelGGL((k_energy<B, G>), dim3(nb), dim3(BLOCK), lds, s, a, cap); \
  } while (0)
  if (bend) {
    if (guard) MS_LAUNCH_E(true, true); else MS_LAUNCH_E(true, false);
  } else {
    if (guard) MS_LAUNCH_E(false, true); else MS_LAUNCH_E(false, false);
  }
#undef MS_LAUNCH_E
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K_C: gradient pass.  Per facet, in registers:
//   surface   g_k = gamma/2 (v_{k+1} - v_{k+2}) x nhat   (surface_energy.f90:80-97)
//   volume    dV/dv0 = (v1 x v2)/6 cyclic               (geometry/body.py:150-190)
//             as constraint row gC and/or penalty k (V - V0) dV/dx
//             (modules/energy/volume.py:94-128)
//   bending   -L fK + sum_k dE/dc_k grad c_k + area-variation term
//             (modules/energy/bending_gradient.py:17-175), cotans recomputed as
//             in tilt_kernels.f90:140-151, never stored.
// BENDMODE: 0 none, 1 analytic, 2 approx (bending.py:163-167).
// LDS: px[3][cap] | (BEND) fk[3][cap] fae[cap] fav[cap] | g[3][T] | gc[3][T] | red[4] | fl[cap]
// ---------------------------------------------------------------------------
template <int BENDMODE>
__global__ __launch_bounds__(BLOCK) void k_gradient(GradientArgs a, int cap) {
  extern __shared__ double lds[];
  constexpr bool BEND = BENDMODE != 0;
  const int T = a.m.T;
  const bool volrow = a.gC != nullptr && (a.modules & MS_CON_VOLUME);
  double* px = lds;
  double* fk = px + 3 * cap;
  double* fae = fk + (BEND ? 3 * cap : 0);
  double* fav = fae + (BEND ? cap : 0);
  double* ag = fav + (BEND ? cap : 0);
  double* agc = ag + 3 * T;
  double* red = agc + 3 * T;
  uint8_t* lfl = reinterpret_cast<uint8_t*>(red + 4);

  const int tile = a.tile0 + xcd_tile(blockIdx.x, a.tile1 - a.tile0);
  const int tid = threadIdx.x;
  const int v_lo = tile * T;
  const int n_owned = min(T, a.m.nv - v_lo);
  const int h0 = a.m.tile_halo_off[tile];
  const int nh = a.m.tile_halo_off[tile + 1] - h0;

  for (int j = tid; j < 3 * n_owned; j += BLOCK) {
    const int r = j / 3, c = j - 3 * r;
    const size_t g = 3 * (size_t)v_lo + j;
    px[c * cap + r] = a.x[g];
    if (BEND) fk[c * cap + r] = a.fK[g];
  }
  for (int r = tid; r < n_owned; r += BLOCK) {
    lfl[r] = a.m.vflags[v_lo + r];
    if (BEND) {
      fae[r] = a.fA[2 * (size_t)(v_lo + r)];
      fav[r] = a.fA[2 * (size_t)(v_lo + r) + 1];
    }
  }
  for (int h = tid; h < nh; h += BLOCK) {
    const int v = a.m.halo_ids[h0 + h];
    const int s = n_owned + h;
    lfl[s] = a.m.vflags[v];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      px[c * cap + s] = a.x[3 * (size_t)v + c];
      if (BEND) fk[c * cap + s] = a.fK[3 * (size_t)v + c];
    }
    if (BEND) {
      fae[s] = a.fA[2 * (size_t)v];
      fav[s] = a.fA[2 * (size_t)v + 1];
    }
  }
  for (int j = tid; j < 6 * T; j += BLOCK) ag[j] = 0.0;
  __syncthreads();

  const bool surf = a.modules & MS_MOD_SURFACE;
  const bool volpen = a.modules & MS_MOD_VOLUME_PENALTY;
  double pen_factor = 0.0;
  if (volpen) pen_factor = a.volume_stiffness * (a.scal[MS_S_VOL] - a.target_volume) / 6.0;

  const int f0 = a.m.tile_facet_off[tile], f1 = a.m.tile_facet_off[tile + 1];
  for (int p = f0 + tid; p < f1; p += BLOCK) {
    const TileFacet tf = a.m.tile_facets[p];
    const bool in0 = tf.l0 < n_owned, in1 = tf.l1 < n_owned, in2 = tf.l2 < n_owned;
    const V3 v0 = lds_v3(px, cap, tf.l0), v1 = lds_v3(px, cap, tf.l1), v2 = lds_v3(px, cap, tf.l2);
    const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
    V3 G0 = mk(0, 0, 0), G1 = mk(0, 0, 0), G2 = mk(0, 0, 0);

    if (surf) {
      const V3 n = cross(e2, -e1);
      const double A2 = norm(n);
      if (A2 >= 1.0e-12) {
        const V3 nh_ = mk(n.x / A2, n.y / A2, n.z / A2);
        const double gam = a.m.tf_gamma[p];
        // g0 = gamma * 0.5 * (v1 - v2) x nhat ; (v1-v2) == -e0 exactly
        const V3 c0v = cross(-e0, nh_), c1v = cross(-e1, nh_), c2v = cross(-e2, nh_);
        G0 = G0 + mk(gam * (0.5 * c0v.x), gam * (0.5 * c0v.y), gam * (0.5 * c0v.z));
        G1 = G1 + mk(gam * (0.5 * c1v.x), gam * (0.5 * c1v.y), gam * (0.5 * c1v.z));
        G2 = G2 + mk(gam * (0.5 * c2v.x), gam * (0.5 * c2v.y), gam * (0.5 * c2v.z));
      }
    }
    if ((volrow || volpen) && (tf.flags & TF_BODY)) {
      const V3 w0 = cross(v1, v2), w1 = cross(v2, v0), w2 = cross(v0, v1);
      if (volpen) {
        G0 = G0 + pen_factor * w0;
        G1 = G1 + pen_factor * w1;
        G2 = G2 + pen_factor * w2;
      }
      if (volrow) {
        const double s6 = 1.0 / 6.0;
        if (in0) lds_add3(agc, T, tf.l0, s6 * w0);
        if (in1) lds_add3(agc, T, tf.l1, s6 * w1);
        if (in2) lds_add3(agc, T, tf.l2, s6 * w2);
      }
    }
    if (BEND) {
      const V3 k0 = lds_v3(fk, cap, tf.l0), k1 = lds_v3(fk, cap, tf.l1), k2 = lds_v3(fk, cap, tf.l2);
      // cotans exactly as compute_curvature_data produces `weights`
      const V3 cr = cross(e1, e2);
      double ad = norm(cr);
      if (ad < 1.0e-12) ad = 1.0e-12;
      const double c0 = dot(-e1, e2) / ad, c1 = dot(-e2, e0) / ad, c2 = dot(-e0, e1) / ad;
      // term 1: -L fK  (bending_kernels.f90:118-129)
      {
        const V3 d02 = k0 - k2, d01 = k0 - k1, d12 = k1 - k2;
        G0 = G0 - 0.5 * (c1 * d02 + c2 * d01);
        G1 = G1 - 0.5 * (c2 * (-d01) + c0 * d12);
        G2 = G2 - 0.5 * (c0 * (-d12) + c1 * (-d02));
      }
      if (BENDMODE == 1) {
        // term 2 (bending_gradient.py:37-78)
        const double dE0 = -0.5 * dot(k1 - k2, v1 - v2);
        const double dE1 = -0.5 * dot(k2 - k0, v2 - v0);
        const double dE2 = -0.5 * dot(k0 - k1, v0 - v1);
        V3 g0u, g0v, g1u, g1v, g2u, g2v;
        grad_cotan(e2, -e1, g0u, g0v);   // corner 0: u = v1-v0, v = v2-v0
        grad_cotan(e0, -e2, g1u, g1v);   // corner 1: u = v2-v1, v = v0-v1
        grad_cotan(e1, -e0, g2u, g2v);   // corner 2: u = v0-v2, v = v1-v2
        // term 3 coefficients (bending_gradient.py:80-95)
        const int t0 = (lfl[tf.l0] & VF_BOUNDARY) ? 0 : 1, t1 = (lfl[tf.l1] & VF_BOUNDARY) ? 0 : 1,
                  t2 = (lfl[tf.l2] & VF_BOUNDARY) ? 0 : 1;
        const int cnt = t0 + t1 + t2;
        const double fe0 = fae[tf.l0], fe1 = fae[tf.l1], fe2 = fae[tf.l2];
        const double avg = cnt > 0 ? (fe0 * t0 + fe1 * t1 + fe2 * t2) / (double)cnt : 0.0;
        const double C0 = (t0 ? fe0 : avg) + fav[tf.l0];
        const double C1 = (t1 ? fe1 : avg) + fav[tf.l1];
        const double C2 = (t2 ? fe2 : avg) + fav[tf.l2];
        double w0 = dE0, w1 = dE1, w2 = dE2;   // weights of grad c_k
        const bool obtuse = (c0 < 0.0) || (c1 < 0.0) || (c2 < 0.0);
        if (!obtuse) {
          // six edge terms (:107-124); the second grad_cotan family
          // (bending_math.py:244-249) has the same arguments as the first, so
          // its three weights fold into w_k.
          const double q10 = 0.25 * c1 * C0, q20 = 0.25 * c2 * C0, q21 = 0.25 * c2 * C1,
                       q01 = 0.25 * c0 * C1, q02 = 0.25 * c0 * C2, q12 = 0.25 * c1 * C2;
          G0 = G0 + q10 * e1 + (-q20) * e2 + (-q21) * e2 + q12 * e1;
          G1 = G1 + q20 * e2 + q21 * e2 + (-q01) * e0 + (-q02) * e0;
          G2 = G2 + (-q10) * e1 + q01 * e0 + q02 * e0 + (-q12) * e1;
          const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
          w0 += 0.125 * l0 * (C1 + C2);
          w1 += 0.125 * l1 * (C0 + C2);
          w2 += 0.125 * l2 * (C0 + C1);
        } else {
          // (:154-173) u = v1-v0, v = v2-v0 for every obtuse corner
          V3 gTu, gTv;
          grad_triangle_area(e2, -e1, gTu, gTv);
          double factor = 0.0;
          if (c0 < 0.0) factor += 0.5 * C0 + 0.25 * C1 + 0.25 * C2;
          if (c1 < 0.0) factor += 0.5 * C1 + 0.25 * C0 + 0.25 * C2;
          if (c2 < 0.0) factor += 0.5 * C2 + 0.25 * C0 + 0.25 * C1;
          G1 = G1 + factor * gTu;
          G2 = G2 + factor * gTv;
          G0 = G0 + factor * (-(gTu + gTv));
        }
        // corner 0 -> (+gu to v1, +gv to v2, -(gu+gv) to v0); cyclic for 1, 2
        G1 = G1 + w0 * g0u;
        G2 = G2 + w0 * g0v;
        G0 = G0 + w0 * (-(g0u + g0v));
        G2 = G2 + w1 * g1u;
        G0 = G0 + w1 * g1v;
        G1 = G1 + w1 * (-(g1u + g1v));
        G0 = G0 + w2 * g2u;
        G1 = G1 + w2 * g2v;
        G2 = G2 + w2 * (-(g2u + g2v));
      }
    }
    if (in0) lds_add3(ag, T, tf.l0, G0);
    if (in1) lds_add3(ag, T, tf.l1, G1);
    if (in2) lds_add3(ag, T, tf.l2, G2);
  }
  __syncthreads();

  // bending.py:165-166: approx mode zeroes the boundary rows of everything
  // accumulated so far.
  if (BENDMODE == 2)
    for (int i = tid; i < n_owned; i += BLOCK)
      if (lfl[i] & VF_BOUNDARY) ag[i] = ag[T + i] = ag[2 * T + i] = 0.0;
  if (BENDMODE == 2) __syncthreads();

  double ggc = 0.0, gcgc = 0.0;
  const bool have_gc = a.gC != nullptr;
  for (int j = tid; j < 3 * n_owned; j += BLOCK) {
    const int r = j / 3, c = j - 3 * r;
    const size_t g = 3 * (size_t)v_lo + j;
    double gv = ag[c * T + r];
    if (a.g) {
      if (a.accumulate) gv += a.g[g];
      a.g[g] = gv;
    }
    if (have_gc) {
      double gc;
      if (volrow) {
        gc = agc[c * T + r];
        a.gC[g] = gc;
      } else {
        gc = a.gC[g];
      }
      ggc += gv * gc;
      gcgc += gc * gc;
    }
  }
  double* out = a.partials + tile;
  const size_t ps = (size_t)a.m.n_tiles;
  double r = block_reduce(ggc, 0, red);
  if (tid == 0) out[MS_S_GGC * ps] = r;
  r = block_reduce(gcgc, 0, red);
  if (tid == 0) out[MS_S_GCGC * ps] = r;
}

size_t gradient_lds_bytes(int T, int cap, bool bend) {
  size_t d = 3 * (size_t)cap + (bend ? 5 * (size_t)cap : 0) + 6 * (size_t)T + 4;
  return d * sizeof(double) + (((size_t)cap + 15) / 16) * 16;
}

hipError_t launch_gradient(const GradientArgs& a, int cap, hipStream_t s) {
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const bool bend = (a.modules & MS_MOD_BENDING) != 0;
  const size_t lds = gradient_lds_bytes(a.m.T, cap, bend);
  hipError_t e;
#define MS_LAUNCH_G(M)                                                              \
  do {                                                                              \
    e = ensure_lds(k_gradient<M>, lds);                                             \
    if (e != hipSuccess) return e;                                                  \
    hipLaunchKernelGGL((k_gradient<M>), dim3(nb), dim3(BLOCK), lds, s, a, cap);     \
  } while (0)
  if (!bend) MS_LAUNCH_G(0);
  else if (a.bending_grad_mode == MS_GRAD_APPROX) MS_LAUNCH_G(2);
  else MS_LAUNCH_G(1);
#undef MS_LAUNCH_G
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Deterministic second stage: one workgroup folds the per-tile partials of the
// requested slots in a fixed order.  Volume gets its 1/6 here
// (geometry/body.py:121: vol_contrib.sum() / 6.0).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_reduce(const double* partials, int n_tiles, int tile0,
                                                  int tile1, uint32_t slot_mask, double* scal) {
  __shared__ double red[4];
  // one workgroup per requested slot; partials are slot-major so lanes read
  // consecutive doubles.
  int slot = -1;
  {
    int k = blockIdx.x;
    for (int s = 0; s < MS_NSCAL; ++s)
      if (slot_mask & (1u << s)) {
        if (k == 0) {
          slot = s;
          break;
        }
        --k;
      }
  }
  if (slot < 0) return;
  const int op = (slot == MS_S_MINEDGE2) ? 1 : ((slot == MS_S_GUARD || slot == MS_S_MAXD2) ? 2 : 0);
  const double* p = partials + (size_t)slot * n_tiles;
  double v = op == 1 ? 1.0e300 : 0.0;
  for (int t = tile0 + threadIdx.x; t < tile1; t += BLOCK) {
    const double q = p[t];
    v = op == 0 ? v + q : (op == 1 ? fmin(v, q) : fmax(v, q));
  }
  v = block_reduce(v, op, red);
  if (threadIdx.x == 0) scal[slot] = (slot == MS_S_VOL) ? v / 6.0 : v;
}

hipError_t launch_reduce(const double* partials, int n_tiles, int tile0, int tile1,
                         uint32_t slot_mask, double* scal, hipStream_t s) {
  const int nslots = __builtin_popcount(slot_mask);
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(k_reduce, dim3(nslots), dim3(BLOCK), 0, s, partials, n_tiles, tile0, tile1,
                     slot_mask, scal);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Direction pass over this shard's vertex rows (one workgroup per tile so the
// partials line up with k_reduce):
//   g <- g - lambda gC, lambda = <g,gC>/<gC,gC> if <gC,gC> > 1e-18
//                                   (runtime/constraint_manager.py:293-301)
//   g[fixed] = 0                    (runtime/minimizer.py:988-990)
//   GD: d = -g                      (gradient_descent.py:53)
//   CG: beta_i = g_i.(g_i - gprev_i)/(gprev_i.gprev_i + 1e-20) PER ROW,
//       d_i = -g_i + beta_i dprev_i, rows with beta_i < 0 reset to -g_i,
//       d[fixed] = 0                (conjugate_gradient.py:78-96)
//   partials: |g|^2, <g,d>, max |d_i|^2 over movable rows (line_search.py:317-321)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_direction(int tile0, int nv, int T, const uint8_t* vflags,
                                                     double* g, const double* gC, double* d,
                                                     const double* pg, const double* pd,
                                                     const double* scal, int use_constraint,
                                                     int cg_history, double* partials,
                                                     int n_tiles) {
  __shared__ double red[4];
  const int tile = tile0 + blockIdx.x;
  double lam = 0.0;
  bool project = false;
  if (use_constraint) {
    const double nsq = scal[MS_S_GCGC];
    if (nsq > 1.0e-18) {
      lam = scal[MS_S_GGC] / nsq;
      project = true;
    }
  }
  double gn2 = 0.0, gd = 0.0, md2 = 0.0;
  for (int i = threadIdx.x; i < T; i += BLOCK) {
    const int v = tile * T + i;
    if (v >= nv) break;
    const size_t o = 3 * (size_t)v;
    const bool fixed = vflags[v] & VF_FIXED;
    V3 gi = mk(g[o], g[o + 1], g[o + 2]);
    if (project) {
      const V3 c = mk(gC[o], gC[o + 1], gC[o + 2]);
      gi = mk(gi.x - lam * c.x, gi.y - lam * c.y, gi.z - lam * c.z);
    }
    if (fixed) gi = mk(0, 0, 0);
    V3 di = -gi;
    if (cg_history) {
      const V3 p = mk(pg[o], pg[o + 1], pg[o + 2]);
      const double beta = dot(gi, gi - p) / (dot(p, p) + 1.0e-20);
      if (!(beta < 0.0)) {
        const V3 q = mk(pd[o], pd[o + 1], pd[o + 2]);
        di = mk(-gi.x + beta * q.x, -gi.y + beta * q.y, -gi.z + beta * q.z);
      }
    }
    if (fixed) di = mk(0, 0, 0);
    g[o] = gi.x;
    g[o + 1] = gi.y;
    g[o + 2] = gi.z;
    d[o] = di.x;
    d[o + 1] = di.y;
    d[o + 2] = di.z;
    gn2 += dot(gi, gi);
    gd += dot(gi, di);
    if (!fixed) md2 = fmax(md2, dot(di, di));
  }
  double* out = partials + tile;
  const size_t ps = (size_t)n_tiles;
  double r = block_reduce(gn2, 0, red);
  if (threadIdx.x == 0) out[MS_S_GNORM2 * ps] = r;
  r = block_reduce(gd, 0, red);
  if (threadIdx.x == 0) out[MS_S_GDOTD * ps] = r;
  r = block_reduce(md2, 2, red);
  if (threadIdx.x == 0) out[MS_S_MAXD2 * ps] = r;
}

hipError_t launch_direction(int tile0, int tile1, int nv, int T, const uint8_t* vflags, double* g,
                            const double* gC, double* d, const double* pg, const double* pd,
                            const double* scal, int use_constraint, int cg_history,
                            double* partials, int n_tiles, hipStream_t s) {
  if (tile1 <= tile0) return hipSuccess;
  hipLaunchKernelGGL(k_direction, dim3(tile1 - tile0), dim3(BLOCK), 0, s, tile0, nv, T, vflags, g,
                     gC, d, pg, pd, scal, use_constraint, cg_history, partials, n_tiles);
  return hipGetLastError();
}

// x[i] += coef * y[i] on movable rows (volume projection, constraints/volume.py:137-141)
__global__ void k_axpy_masked(int64_t n_rows, const uint8_t* vflags, double* x, const double* y,
                              double coef) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 3 * n_rows) return;
  if (vflags[j / 3] & VF_FIXED) return;
  x[j] += coef * y[j];
}

hipError_t launch_axpy_masked(int64_t n_rows, const uint8_t* vflags, double* x, const double* y,
                              double coef, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t n = 3 * n_rows;
  hipLaunchKernelGGL(k_axpy_masked, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                     n_rows, vflags, x, y, coef);
  return hipGetLastError();
}

// external row order <-> patch order
__global__ void k_permute_in(int nv, const int32_t* perm, const double* src_ext, double* dst_int,
                             int ncomp) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (int64_t)nv * ncomp) return;
  const int i = (int)(j / ncomp), c = (int)(j - (int64_t)i * ncomp);
  dst_int[j] = src_ext[(size_t)perm[i] * ncomp + c];
}
__global__ void k_permute_out(int nv, const int32_t* perm, const double* src_int, double* dst_ext,
                              int ncomp) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (int64_t)nv * ncomp) return;
  const int i = (int)(j / ncomp), c = (int)(j - (int64_t)i * ncomp);
  dst_ext[(size_t)perm[i] * ncomp + c] = src_int[j];
}
hipError_t launch_permute_in(int nv, const int32_t* perm, const double* src_ext, double* dst_int,
                             int ncomp, hipStream_t s) {
  const int64_t n = (int64_t)nv * ncomp;
  hipLaunchKernelGGL(k_permute_in, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, nv,
                     perm, src_ext, dst_int, ncomp);
  return hipGetLastError();
}
hipError_t launch_permute_out(int nv, const int32_t* perm, const double* src_int, double* dst_ext,
                              int ncomp, hipStream_t s) {
  const int64_t n = (int64_t)nv * ncomp;
  hipLaunchKernelGGL(k_permute_out, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, nv,
                     perm, src_int, dst_ext, ncomp);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Kernel-provider seam (fortran_kernels/loader.py KernelSpec): standalone
// per-procedure kernels on caller arrays in the reference's own row order.
// These serve host-array parity calls; the minimizer path uses the tiled
// kernels above.  Scatter here is global_atomic_add_f64 (small inputs).
// ---------------------------------------------------------------------------
__global__ void k_grad_cotan_batch(int n, const double* u, const double* v, double* gu, double* gv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 a, b;
  grad_cotan(mk(u[3 * i], u[3 * i + 1], u[3 * i + 2]), mk(v[3 * i], v[3 * i + 1], v[3 * i + 2]), a, b);
  gu[3 * i] = a.x; gu[3 * i + 1] = a.y; gu[3 * i + 2] = a.z;
  gv[3 * i] = b.x; gv[3 * i + 1] = b.y; gv[3 * i + 2] = b.z;
}
hipError_t launch_grad_cotan(int n, const double* u, const double* v, double* gu, double* gv,
                             hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_grad_cotan_batch, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, n, u, v, gu, gv);
  return hipGetLastError();
}

__device__ __forceinline__ V3 ldg3(const double* p, int i) {
  return mk(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]);
}
__device__ __forceinline__ void st3(double* p, int i, V3 v) {
  p[3 * (size_t)i] = v.x; p[3 * (size_t)i + 1] = v.y; p[3 * (size_t)i + 2] = v.z;
}
__device__ __forceinline__ void atom3(double* p, int i, V3 v) {
  atomicAdd(&p[3 * (size_t)i], v.x);
  atomicAdd(&p[3 * (size_t)i + 1], v.y);
  atomicAdd(&p[3 * (size_t)i + 2], v.z);
}

// tilt_kernels.f90:26-86 (outputs pre-zeroed by the caller)
__global__ void k_p1_divergence(int nv, int nf, const double* pos, const double* tilts,
                                const int32_t* tri, double* div, double* area, double* g0,
                                double* g1, double* g2) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
  if (i0 < 0 || i0 >= nv || i1 < 0 || i1 >= nv || i2 < 0 || i2 >= nv) return;
  const V3 v0 = ldg3(pos, i0), v1 = ldg3(pos, i1), v2 = ldg3(pos, i2);
  const V3 n = cross(v1 - v0, v2 - v0);
  const double n2 = dot(n, n);
  const double denom = fmax(n2, 1.0e-20);
  const V3 a = cross(n, v2 - v1), b = cross(n, v0 - v2), c = cross(n, v1 - v0);
  const V3 ga = mk(a.x / denom, a.y / denom, a.z / denom);
  const V3 gb = mk(b.x / denom, b.y / denom, b.z / denom);
  const V3 gc = mk(c.x / denom, c.y / denom, c.z / denom);
  st3(g0, f, ga);
  st3(g1, f, gb);
  st3(g2, f, gc);
  div[f] = dot(ldg3(tilts, i0), ga) + dot(ldg3(tilts, i1), gb) + dot(ldg3(tilts, i2), gc);
  area[f] = 0.5 * sqrt(fmax(n2, 0.0));
}
hipError_t launch_p1_divergence(int nv, int nf, const double* pos, const double* tilts,
                                const int32_t* tri, double* div, double* area, double* g0,
                                double* g1, double* g2, hipStream_t s) {
  if (nf <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_p1_divergence, dim3((nf + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, nv, nf, pos,
                     tilts, tri, div, area, g0, g1, g2);
  return hipGetLastError();
}

// bending_kernels.f90:87-131 (out pre-zeroed by the caller)
__global__ void k_laplacian_scatter(int dim, int nv, int nf, const double* weights,
                                    const int32_t* tri, const double* field, double* out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int v0 = tri[3 * f], v1 = tri[3 * f + 1], v2 = tri[3 * f + 2];
  if (v0 < 0 || v0 >= nv || v1 < 0 || v1 >= nv || v2 < 0 || v2 >= nv) return;
  const double c0 = weights[3 * f], c1 = weights[3 * f + 1], c2 = weights[3 * f + 2];
  for (int d = 0; d < dim; ++d) {
    const double f0 = field[(size_t)v0 * dim + d], f1 = field[(size_t)v1 * dim + d],
                 f2 = field[(size_t)v2 * dim + d];
    atomicAdd(&out[(size_t)v0 * dim + d], 0.5 * (c1 * (f0 - f2) + c2 * (f0 - f1)));
    atomicAdd(&out[(size_t)v1 * dim + d], 0.5 * (c2 * (f1 - f0) + c0 * (f1 - f2)));
    atomicAdd(&out[(size_t)v2 * dim + d], 0.5 * (c0 * (f2 - f1) + c1 * (f2 - f0)));
  }
}
hipError_t launch_laplacian_scatter(int dim, int nv, int nf, const double* weights,
                                    const int32_t* tri, const double* field, double* out,
                                    hipStream_t s) {
  if (nf <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_laplacian_scatter, dim3((nf + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, dim, nv,
                     nf, weights, tri, field, out);
  return hipGetLastError();
}

// tilt_kernels.f90:88-190 (outputs pre-zeroed by the caller; va* may be null)
__global__ void k_curvature_raw(int nv, int nf, const double* pos, const int32_t* tri,
                                double* k_vecs, double* areas, double* weights, double* va0,
                                double* va1, double* va2) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
  if (i0 < 0 || i0 >= nv || i1 < 0 || i1 >= nv || i2 < 0 || i2 >= nv) return;
  const V3 v0 = ldg3(pos, i0), v1 = ldg3(pos, i1), v2 = ldg3(pos, i2);
  const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
  const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
  double ad = norm(cross(e1, e2));
  if (ad < 1.0e-12) ad = 1.0e-12;
  const double c0 = dot(-e1, e2) / ad, c1 = dot(-e2, e0) / ad, c2 = dot(-e0, e1) / ad;
  weights[3 * (size_t)f] = c0;
  weights[3 * (size_t)f + 1] = c1;
  weights[3 * (size_t)f + 2] = c2;
  atom3(k_vecs, i0, 0.5 * (c1 * (-e1) + c2 * e2));
  atom3(k_vecs, i1, 0.5 * (c2 * (-e2) + c0 * e0));
  atom3(k_vecs, i2, 0.5 * (c0 * (-e0) + c1 * e1));
  double a0, a1, a2;
  corner_areas(c0, c1, c2, l0, l1, l2, 0.5 * ad, a0, a1, a2);
  atomicAdd(&areas[i0], a0);
  atomicAdd(&areas[i1], a1);
  atomicAdd(&areas[i2], a2);
  if (va0) va0[f] = a0;
  if (va1) va1[f] = a1;
  if (va2) va2[f] = a2;
}
hipError_t launch_curvature_raw(int nv, int nf, const double* pos, const int32_t* tri,
                                double* k_vecs, double* areas, double* weights, double* va0,
                                double* va1, double* va2, hipStream_t s) {
  if (nf <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_curvature_raw, dim3((nf + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, nv, nf, pos,
                     tri, k_vecs, areas, weights, va0, va1, va2);
  return hipGetLastError();
}

}  // namespace ms
