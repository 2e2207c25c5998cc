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
#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>

#include "ms_internal.h"

#ifndef MS_RSQRT_NORM
#define MS_RSQRT_NORM 1  // facet normal length and its reciprocal from one refined rsqrt (norm_and_inverse)
#endif

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
// min of two values known not to be signalling NaNs: one v_min_f64 (fmin() canonicalises both operands first)
__device__ __forceinline__ double min_plain(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

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

// Wave64 reductions with DPP row shifts / row broadcasts (VALU-speed lane exchange; the ds_bpermute form of
// __shfl_down costs an LDS round trip per step).  The result is valid in EVERY lane (read back from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double fill, double v) {
  const long long vi = __double_as_longlong(v), fi = __double_as_longlong(fill);
  const int lo = __builtin_amdgcn_update_dpp((int)fi, (int)vi, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(vi >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int OP>  // 0 sum, 1 min, 2 max
__device__ __forceinline__ double wave_reduce(double v) {
  const double id = OP == 0 ? 0.0 : (OP == 1 ? 1.0e300 : -1.0e300);
#define MS_RED_STEP(CTRL, MASK)                                         \
  {                                                                     \
    const double t = dpp_move<CTRL, MASK>(id, v);                       \
    v = OP == 0 ? v + t : (OP == 1 ? fmin(v, t) : fmax(v, t));          \
  }
  MS_RED_STEP(0x111, 0xf)  // row_shr:1
  MS_RED_STEP(0x112, 0xf)  // row_shr:2
  MS_RED_STEP(0x114, 0xf)  // row_shr:4
  MS_RED_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds the row's result
  MS_RED_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  MS_RED_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
#undef MS_RED_STEP
  const long long vi = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)vi, 63), hi = __builtin_amdgcn_readlane((int)(vi >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<0>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<1>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<2>(v); }
// op: 0 sum, 1 min, 2 max.  red = 16 doubles of LDS scratch.  Result valid in thread 0.
__device__ __forceinline__ double block_reduce(double v, int op, double* red) {
  v = op == 0 ? wave_sum(v) : (op == 1 ? wave_min(v) : wave_max(v));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = red[0];
    for (int i = 1; i < nw; ++i)
      r = op == 0 ? r + red[i] : (op == 1 ? fmin(r, red[i]) : fmax(r, red[i]));
    v = r;
  }
  return v;
}

// Reduce NV per-thread values with ONE barrier: wave shuffles, lane 0 of each wave
// parks its partials in red[k*16 + wave], then thread k folds value k across waves
// and stores it to out[slots[k] * stride].  ops[k]: 0 sum, 1 min, 2 max.
__device__ __forceinline__ void st_agent(double* p, double v);
// AGENT: the partials are read by OTHER workgroups of the same launch (k_resident): agent-scope write-through stores
template <int NV, bool AGENT = false>
__device__ __forceinline__ void block_reduce_store(const double (&v)[NV], const int (&ops)[NV],
                                                   const int (&slots)[NV], double* red,
                                                   double* out, size_t stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double r = ops[k] == 0 ? wave_sum(v[k]) : (ops[k] == 1 ? wave_min(v[k]) : wave_max(v[k]));
    if (lane == 0) red[k * 16 + w] = r;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    int op = 0, slot = 0;
#pragma unroll
    for (int q = 0; q < NV; ++q)
      if (q == k) {
        op = ops[q];
        slot = slots[q];
      }
    double r = red[k * 16];
    for (int i = 1; i < nw; ++i) {
      const double x = red[k * 16 + i];
      r = op == 0 ? r + x : (op == 1 ? fmin(r, x) : fmax(r, x));
    }
    if (AGENT) st_agent(out + (size_t)slot * stride, r);
    else out[(size_t)slot * stride] = r;
  }
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

// dot product with the contraction spelled out: the direction scalars (|g|^2, <g,d>,
// max|d_i|^2) are produced at two code sites (k_gradient's fused epilogue, k_direction) that
// must agree to the last bit.
__device__ __forceinline__ double dot_pinned(V3 a, V3 b) {
  return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
}

// x + alpha d, spelled out ONCE: every trial position (energy pass rows and halo rows, tilt / bending_tilt /
// smoothness / disk-target passes) and every commit (ms_phase_commit_trial's axpy, the volume projection) goes through
// this helper, so the committed doubles are the evaluated ones and a vertex gets the same position in every tile that
// stages it -- independent of where the compiler would or would not contract a mul+add.  Built with
// -ffp-contract=off (MS_FP_CONTRACT=off, the CPU oracle's rounding) it is the plain two-rounding form.
__device__ __forceinline__ double axpy1(double x, double alpha, double d) {
#ifdef MS_FP_CONTRACT_OFF
  return x + alpha * d;
#else
  return fma(alpha, d, x);
#endif
}

// A relaxation's trial row P(t + coef src), P = projection onto the tangent plane of the frozen unit normal n
// (tilt_relaxation.py:330-334), spelled out ONCE: k_tvec mode 2 writes such rows, the search pass (ms_tsearch.inc) and
// the fused one-tile relaxation form them where they are used -- the same doubles at every site.
__device__ __forceinline__ V3 tilt_trial_row(V3 t0, V3 s, V3 n, double coef) {
  const V3 q = mk(axpy1(t0.x, coef, s.x), axpy1(t0.y, coef, s.y), axpy1(t0.z, coef, s.z));
  const double dt = dot_pinned(q, n);
  return mk(axpy1(q.x, -dt, n.x), axpy1(q.y, -dt, n.y), axpy1(q.z, -dt, n.z));
}

// Cross-kernel scalars (the folded reductions and the decision words k_reduce derives from them) travel with
// agent-scope accesses on BOTH sides: write-through `sc1` stores, `sc1` vector loads that bypass the CU's L1 and never
// go through the scalar data cache.  They are the only data in the library that one kernel writes and the next one
// branches on per workgroup; with these forms their visibility does not hang on what a kernel boundary does to the
// per-XCD L2s and the scalar caches (MI355X_MICROARCH.md, "inter-workgroup visibility").
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Diagnostic build (-DMS_GATE_PROBE=1, tools/gate_probe.sh): how do the three read paths of a scalar that the PREVIOUS
// kernel of the stream folded compare?  k_reduce stores every folded slot of the ordinary set once more, the round-2 way
// (a plain store from one lane), into g_shadow; every workgroup of a gated gradient pass reads the bending energy from
// there through the scalar data cache (s_load of a uniform address: what the round-2 gates did) and through the vector
// L1/L2 (plain global_load), and compares both with the agent-scope load of the device scalars.  Counters only.
#ifndef MS_GATE_PROBE
#define MS_GATE_PROBE 0
#endif
#if MS_GATE_PROBE
__device__ unsigned long long g_probe[32];
__device__ double g_shadow[MS_NSCAL];
#endif

// Gate of a queued launch: open iff the decision word holds the code the host queued the launch for.  The word was
// written by ONE lane of an earlier kernel of the stream and does not change while this launch runs, so every workgroup
// takes the same branch.  Thread 0 records what the workgroup did in the MS_P_RAN row of the partials: the fold that
// follows checks that the launch ran everywhere or nowhere.
__device__ __forceinline__ bool gate_open(const uint32_t* word, uint32_t want, double* ran_cell) {
  const bool run = ld_agent(word) == want;
  if (threadIdx.x == 0 && ran_cell) *ran_cell = run ? 1.0 : 0.0;
  return run;
}

// |n| and 1/|n| from ONE reciprocal square root refined twice (v_rsq_f64 + two Newton steps: ~9 instructions instead
// of the ~24 of an fp64 sqrt followed by an fp64 division); both within an ulp or two of the correctly rounded values.
// n2 <= 1e-30 (|n| <= 1e-15, below every clamp of the reference): both 0.
__device__ __forceinline__ void norm_and_inverse(double n2, double& nrm, double& inv) {
  double rs = 0.0;
  if (n2 > 1.0e-30) {
    rs = __builtin_amdgcn_rsq(n2);
    const double hn = 0.5 * n2;
    rs = rs * fma(-hn, rs * rs, 1.5);
    rs = rs * fma(-hn, rs * rs, 1.5);
  }
  nrm = n2 * rs;
  inv = rs;
}

__device__ __forceinline__ V3 lds_v3(const double* base, int cap, int slot) {
  return mk(base[slot], base[cap + slot], base[2 * cap + slot]);
}
// 24-byte rows (x y z) in LDS: one address register per corner, the components at immediate offsets.  For 8-byte
// reads the bank pair of row s, component c is (3 s + c) mod 32 -- 3 is odd, so rows that differ mod 32 do not conflict.
__device__ __forceinline__ V3 lds_row3(const double* rows, int slot) {
  const double* r = rows + 3 * slot;
  return mk(r[0], r[1], r[2]);
}
__device__ __forceinline__ void lds_put3(double* rows, int slot, double x, double y, double z) {
  double* r = rows + 3 * slot;
  r[0] = x;
  r[1] = y;
  r[2] = z;
}

// ---------------------------------------------------------------------------
// Tile kernels.  blockDim.x == T: thread i is BOTH "facet lane i of the current
// chunk" and "owned vertex i".  Per chunk of T facets:
//   facet phase : each lane computes its facet's per-corner contributions in
//                 registers and writes them to an LDS staging block with plain,
//                 conflict-free column stores (stg[component][lane]);
//   vertex phase: each lane walks its vertex's (facet,corner) list -- a per-tile
//                 CSR staged in LDS, ascending in facet order -- and adds the
//                 staged columns it owns into register accumulators.
// No atomics: the summation order per vertex is fixed by the CSR, so results
// are bitwise reproducible run to run.
// ---------------------------------------------------------------------------
// A tile's facet record as it travels from HBM to the loop that uses it: the 8-byte TileFacet, or (T = 256
// instances) its 4-byte packed form, unpacked with three bit-field extracts where the slots are needed.
template <bool PACKED>
struct FacetRec {
  TileFacet v;
};
template <>
struct FacetRec<true> {
  uint32_t v;
};
template <bool PACKED>
__device__ __forceinline__ FacetRec<PACKED> facet_load(const DeviceMesh& m, size_t i) {
  FacetRec<PACKED> r;
  if constexpr (PACKED) r.v = m.tile_facets32[i];
  else r.v = m.tile_facets[i];
  return r;
}
template <bool PACKED>
__device__ __forceinline__ FacetRec<PACKED> facet_null() {
  FacetRec<PACKED> r;
  if constexpr (PACKED) r.v = 0u;
  else r.v = TileFacet{0, 0, 0, 0};
  return r;
}
template <bool PACKED>
__device__ __forceinline__ TileFacet facet_unpack(const FacetRec<PACKED>& r) {
  if constexpr (PACKED) {
    TileFacet f;
    f.l0 = (uint16_t)(r.v & 1023u);
    f.l1 = (uint16_t)((r.v >> 10) & 1023u);
    f.l2 = (uint16_t)((r.v >> 20) & 1023u);
    f.flags = (uint16_t)(r.v >> 30);
    return f;
  } else {
    return r.v;
  }
}

struct TileCtx {
  int tile, v_lo, n_owned, h0, nh, f0, f1, e0, n_ent;
};
__device__ __forceinline__ TileCtx tile_ctx(const DeviceMesh& m, int tile) {
  TileCtx t;
  t.tile = tile;
  t.v_lo = tile * m.own;
  t.n_owned = min(m.own, m.nv - t.v_lo);
  t.h0 = m.tile_halo_off[tile];
  t.nh = m.tile_halo_off[tile + 1] - t.h0;
  t.f0 = m.tile_facet_off[tile];
  t.f1 = m.tile_facet_off[tile + 1];
#ifdef MS_ABL_CLAMP  // timing experiment only (wrong results): at most this many facets per tile
  t.f1 = min(t.f1, t.f0 + MS_ABL_CLAMP);
#endif
  t.e0 = m.tile_ent_off[tile];
  t.n_ent = m.tile_ent_off[tile + 1] - t.e0;
  return t;
}

// Per-tile CSR words held in registers between "issue" and "commit": every global load of a
// tile's stage-in is issued before the first LDS write, so the whole prologue costs two HBM
// round trips (ids/rows/CSR, then the halo rows the ids name) instead of one per array.
// occupancy attributes of the two tile kernels for A/B builds (tools/ab_variants.sh); empty by default
#ifndef MS_WPE_ENERGY
#define MS_WPE_ENERGY
#endif
#ifndef MS_WPE_GRADIENT
#define MS_WPE_GRADIENT
#endif
#ifndef MS_KA_SLOTS
#define MS_KA_SLOTS 5  // workgroups per CU the headline k_energy instances are compiled for
#endif
#ifndef MS_ROW_TWO_PASS
#define MS_ROW_TWO_PASS 0
#endif
#ifndef MS_LEAN_SLOTS
#define MS_LEAN_SLOTS 5  // workgroups per CU the lean k_gradient instance is compiled for (<= 96 VGPRs)
#endif
constexpr int CSR_REGS = 12;  // entries per thread held in registers (T=256: 3072 entries)
struct CsrStage {
  uint16_t vo0, vo1;  // this vertex's entry range [vo0, vo1) (offset row of the tile)
  uint16_t e[CSR_REGS];
};
__device__ __forceinline__ void csr_issue(CsrStage& r, const DeviceMesh& m, const TileCtx& t, int T,
                                          int tid) {
  const uint16_t* gv = m.tile_voff + (size_t)t.tile * (T + 1);
  r.vo0 = gv[tid];
  r.vo1 = gv[tid + 1];
  const uint16_t* ge = m.vent + t.e0;
#pragma unroll
  for (int k = 0; k < CSR_REGS; ++k) {
    const int j = tid + k * T;
    r.e[k] = j < t.n_ent ? ge[j] : (uint16_t)0;
  }
}
__device__ __forceinline__ void csr_commit(const CsrStage& r, const DeviceMesh& m, const TileCtx& t,
                                           int T, int tid, uint16_t* vent) {
#pragma unroll
  for (int k = 0; k < CSR_REGS; ++k) {
    const int j = tid + k * T;
    if (j < t.n_ent) vent[j] = r.e[k];
  }
  const uint16_t* ge = m.vent + t.e0;
  for (int j = tid + CSR_REGS * T; j < t.n_ent; j += T) vent[j] = ge[j];
}

// ---------------------------------------------------------------------------
// K_A: energy pass.
//   scalars (owner facets): E_surface (surface_energy.f90:61-78), body volume
//   (geometry/body.py:104-123), min edge^2 (runtime/topology.py:174-199),
//   normal-rotation guard (runtime/topology.py:13-48);
//   BEND: per-vertex K, A_vor (tilt_kernels.f90:88-190), A_eff
//   (bending_utils.py:37-171), normal sums (bending_utils.py:13-34), then the
//   per-vertex density / back-prop factors of bending.py:117-161 on the owned
//   rows.  Evaluates at x + alpha*d when d != nullptr (trial[movable] = base +
//   alpha d, line_search.py:362-368) and can write that trial block to xt.
// One normal n = (v1-v0)x(v2-v0) and one sqrt per facet serve the surface term,
// the cotans (|e1 x e2| is the same vector) and both area clamps.
// LDS: px[cap][3] rows | (GUARD) ox[cap][3] | (BEND) stg[9][T] (the reduction scratch aliases it)
//      | (BEND) vent[max_ent] (u16) | fl[cap] (u8, only with boundary vertices / GUARD)
// ---------------------------------------------------------------------------
// TT / CAPC: compile-time tile size and LDS patch capacity (0 = take the runtime values);
// with constants every LDS address becomes base + immediate offset.
// PAIR: two trial evaluations of one line search in ONE launch (grid = 2 x tiles): the workgroups of a tile's two
// evaluations sit 8 apart in the launch order -- same XCD, dispatched together -- so the second read of the
// tile's x / d / facet rows hits that XCD's L2.  The odd ones evaluate at alpha2 into the "2" outputs.
// MULTI = 3: the same with three evaluations (trials 0, 1, 2; the LAST one uses the ordinary outputs).
// Diagnostic build (-DMS_STAMPS=1, tools/kprobe.py --stamps): every workgroup of the two tile kernels records when
// it started, finished its stage-in, finished its facet loop and ended (s_memrealtime, 100 MHz) plus the XCC / CU it
// ran on.  No stamp executes in the normal build.
#ifndef MS_STAMPS
#define MS_STAMPS 0
#endif
#if MS_STAMPS
__device__ unsigned long long g_stamps[8 * 16384];
#define MS_STAMP(k)                                                                   \
  do {                                                                                \
    if (threadIdx.x == 0) g_stamps[8 * (size_t)blk + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define MS_STAMP_ID()                                                                                  \
  do {                                                                                                 \
    if (threadIdx.x == 0) {                                                                            \
      g_stamps[8 * (size_t)blk + 4] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); /* XCC_ID */ \
      g_stamps[8 * (size_t)blk + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  /* HW_ID */  \
    }                                                                                                  \
  } while (0)
#else
#define MS_STAMP(k) do {} while (0)
#define MS_STAMP_ID() do {} while (0)
#endif
// (energy_body: the kernel's code as a device function of (arguments, LDS base, block index) -- k_energy below is its
// one-block-per-tile launch; the one-workgroup interpreter k_exec runs the same body command by command)
template <bool BEND, bool GUARD, int TT, int CAPC, bool ATOMIC, int MULTI = 0>
__device__ __forceinline__ void energy_body(EnergyArgs& a, int cap_rt, int max_ent, double* lds, int block_id) {
  int bid = block_id;
  double* const ran_row = a.partials + (size_t)MS_P_RAN * a.m.n_tiles;  // (of the ordinary partials)
  bool last_trial = true;  // this workgroup evaluates the launch's last trial (ordinary outputs)
  if (MULTI) {
    const int nm = MULTI == 8 ? a.pair : MULTI;  // (8: the trial count is a launch argument)
    const int k = bid % NXCD, j2 = bid / NXCD;
    const int st = j2 % nm;
    bid = (j2 / nm) * NXCD + k;
    if (bid >= a.tile1 - a.tile0) return;
    if (st != 0) {  // trial st-1 of the ladder: its own alpha and partials; outputs only where the caller gave some
      last_trial = false;
      a.xt = a.fK = a.fA = nullptr;
      // (selected with constant indices: a dynamically indexed kernel argument would move the whole struct to scratch)
#pragma unroll
      for (int j = 0; j < MS_MAX_TRIALS - 1; ++j)
        if (j < (MULTI == 8 ? MS_MAX_TRIALS - 1 : MULTI - 1) && st - 1 == j) {
          a.alpha = a.alpha_side[j];
          a.partials = a.partials_side[j];
          if (j < 2) {
            a.xt = a.xt_side[j];
            a.fK = a.fK_side[j];
            a.fA = a.fA_side[j];
          }
        }
    }
  }
  const int T = TT ? TT : a.m.T;  // == blockDim.x
  const int cap = CAPC ? CAPC : cap_rt;
  double* px = lds;
  double* ox = px + 3 * cap;
  double* stg = ox + (GUARD ? 3 * cap : 0);
  // the reduction scratch aliases the staging block (free after the chunk loop's last
  // barrier); the vertex offsets stay in registers; the flag bytes are staged only when
  // some vertex carries VF_BOUNDARY
  double* red = stg;
  // ATOMIC: stg holds the five per-vertex accumulators K(3), A_vor, A_eff - A_vor (ds_add_f64), no CSR
  uint16_t* vent = reinterpret_cast<uint16_t*>(stg + (BEND ? (ATOMIC ? 5 : 9) * T : 5 * 16));
  uint8_t* lfl = reinterpret_cast<uint8_t*>(vent + ((BEND && !ATOMIC) ? ((max_ent + 3) & ~3) : 0));
  const bool stage_flags = a.m.has_boundary || GUARD;
  // queued line-search stage: runs only if the decision word says so (DEC_CONTINUE: every trial before it was rejected)
  const int my_tile = a.tile0 + xcd_tile(bid, a.tile1 - a.tile0);
  if (a.gate != nullptr && !gate_open(a.gate, a.gate_want, last_trial ? ran_row + my_tile : nullptr)) return;

  const int blk = bid;
  (void)blk;
  MS_STAMP(0);
  MS_STAMP_ID();
  const TileCtx t = tile_ctx(a.m, my_tile);
  const int tid = threadIdx.x;
  const bool have_d = a.d != nullptr;

  // facet records are fetched one chunk ahead so their HBM latency overlaps the
  // staging / the previous chunk's arithmetic
  constexpr bool PACKED = TT == 256;
  FacetRec<PACKED> tf_nx = facet_null<PACKED>();
  double gam_nx = 0.0;
  if (t.f0 + tid < t.f1) {
    tf_nx = facet_load<PACKED>(a.m, (size_t)(t.f0 + tid));
    gam_nx = a.m.gamma_uniform ? a.m.gamma_const : a.m.tf_gamma[t.f0 + tid];
  }

  // per-vertex parameters of the epilogue are requested now, used ~all the way down
  double kappa_v = 0.0, c0_v = 0.0;
  if (BEND && tid < t.n_owned) {
    kappa_v = a.m.kc_uniform ? a.m.kappa_const : a.m.kappa[t.v_lo + tid];
    c0_v = a.m.kc_uniform ? a.m.c0_const : a.m.c0[t.v_lo + tid];
  }

  uint8_t own_fl = 0;  // this thread's own vertex flags (epilogue)
  int cur = 0, end = 0;  // this vertex's CSR range, straight from the tile's offset row
  // -- stage-in: issue every independent load, then the halo rows, then write LDS ------
  // (the xt store may alias x/d as far as the compiler knows; a store or an LDS write in the
  // middle would serialise the prologue into one HBM round trip per array)
  {
    const bool own = tid < t.n_owned;
    const int v_own = t.v_lo + tid;
    const bool has_h = tid < t.nh;
    int hv = 0;
    if (has_h) hv = a.m.halo_ids[t.h0 + tid];
    double xo0 = 0, xo1 = 0, xo2 = 0, dd0 = 0, dd1 = 0, dd2 = 0;
    uint8_t fl = 0;
    if (own) {
      const size_t g = 3 * (size_t)v_own;
      fl = a.m.vflags[v_own];
      xo0 = a.x[g];
      xo1 = a.x[g + 1];
      xo2 = a.x[g + 2];
      if (have_d) {
        dd0 = a.d[g];
        dd1 = a.d[g + 1];
        dd2 = a.d[g + 2];
      }
    }
    CsrStage cs;
    if (BEND && !ATOMIC) csr_issue(cs, a.m, t, T, tid);
    if (BEND && ATOMIC) {
      for (int j = tid; j < 5 * T; j += T) stg[j] = 0.0;
    }
    own_fl = fl;
    // second round trip: the halo row this thread covers
    double h0 = 0, h1 = 0, h2 = 0, e0 = 0, e1 = 0, e2 = 0;
    uint8_t hfl = 0;
    if (has_h) {
      const size_t g = 3 * (size_t)hv;
      hfl = a.m.vflags[hv];
      h0 = a.x[g];
      h1 = a.x[g + 1];
      h2 = a.x[g + 2];
      if (have_d) {
        e0 = a.d[g];
        e1 = a.d[g + 1];
        e2 = a.d[g + 2];
      }
    }
    if (own) {
      const size_t g = 3 * (size_t)v_own;
      const bool mv = have_d && !(fl & VF_FIXED);
      const double x0 = mv ? axpy1(xo0, a.alpha, dd0) : xo0;
      const double x1 = mv ? axpy1(xo1, a.alpha, dd1) : xo1;
      const double x2 = mv ? axpy1(xo2, a.alpha, dd2) : xo2;
      if (stage_flags) lfl[tid] = fl;
      lds_put3(px, tid, x0, x1, x2);
      if (GUARD) lds_put3(ox, tid, xo0, xo1, xo2);
      if (a.xt) {
        a.xt[g] = x0;
        a.xt[g + 1] = x1;
        a.xt[g + 2] = x2;
      }
    }
    if (BEND && !ATOMIC) {
      csr_commit(cs, a.m, t, T, tid, vent);
      cur = cs.vo0;
      end = cs.vo1;
    }
    if (has_h) {
      const int s = t.n_owned + tid;
      const bool mv = have_d && !(hfl & VF_FIXED);
      if (stage_flags) lfl[s] = hfl;
      lds_put3(px, s, mv ? axpy1(h0, a.alpha, e0) : h0, mv ? axpy1(h1, a.alpha, e1) : h1, mv ? axpy1(h2, a.alpha, e2) : h2);
      if (GUARD) lds_put3(ox, s, h0, h1, h2);
    }
    for (int h = tid + T; h < t.nh; h += T) {  // halo longer than the workgroup (small tiles)
      const int v = a.m.halo_ids[t.h0 + h];
      const size_t g = 3 * (size_t)v;
      const uint8_t f2 = a.m.vflags[v];
      const double q0 = a.x[g], q1 = a.x[g + 1], q2 = a.x[g + 2];
      double r0 = 0, r1 = 0, r2 = 0;
      if (have_d) {
        r0 = a.d[g];
        r1 = a.d[g + 1];
        r2 = a.d[g + 2];
      }
      const int s = t.n_owned + h;
      const bool mv = have_d && !(f2 & VF_FIXED);
      if (stage_flags) lfl[s] = f2;
      lds_put3(px, s, mv ? axpy1(q0, a.alpha, r0) : q0, mv ? axpy1(q1, a.alpha, r1) : q1, mv ? axpy1(q2, a.alpha, r2) : q2);
      if (GUARD) lds_put3(ox, s, q0, q1, q2);
    }
  }
  __syncthreads();
  MS_STAMP(1);

  double e_surf = 0.0, vol = 0.0, min_e2 = 1.0e300, guard = 0.0;
  const bool want_surf = a.modules & MS_MOD_SURFACE;
  const bool want_vol = a.modules & (MS_MOD_VOLUME_PENALTY | MS_CON_VOLUME | MS_TRACK_VOLUME);
  // vertex accumulators (BEND): K(3), A_vor, A_eff
  double aKx = 0, aKy = 0, aKz = 0, aAv = 0, aAe = 0;
  if (!(BEND && tid < t.n_owned)) cur = end = 0;
  const int ent_begin = cur;

  for (int c0 = t.f0; c0 < t.f1; c0 += T) {
    const int p = c0 + tid;
    const TileFacet tf = facet_unpack<PACKED>(tf_nx);
    const double gam = gam_nx;
    if (p + T < t.f1) {
      tf_nx = facet_load<PACKED>(a.m, (size_t)(p + T));
      gam_nx = a.m.gamma_uniform ? a.m.gamma_const : a.m.tf_gamma[p + T];
    }
    double va0 = 0, va1 = 0, va2 = 0, ve0 = 0, ve1 = 0, ve2 = 0;
    if (p < t.f1) {
      const bool owner = tf.flags & TF_OWNER;
      if (BEND || owner) {
        const V3 v0 = lds_row3(px, tf.l0), v1 = lds_row3(px, tf.l1), v2 = lds_row3(px, tf.l2);
        const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
        const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
        const V3 n = cross(e2, -e1);  // (v1-v0) x (v2-v0) == e1 x e2
#if MS_RSQRT_NORM
        double A2, inv_A2;
        norm_and_inverse(dot(n, n), A2, inv_A2);
#else
        const double A2 = norm(n);
#endif
        if (owner) {
          if (want_surf && A2 >= 1.0e-12) e_surf += gam * (0.5 * A2);
          if (want_vol && (tf.flags & TF_BODY)) vol += dot(cross(v1, v2), v0);
          min_e2 = min_plain(min_e2, min_plain(l0, min_plain(l1, l2)));
          if (GUARD) {
            const V3 o0 = lds_row3(ox, tf.l0), o1 = lds_row3(ox, tf.l1), o2 = lds_row3(ox, tf.l2);
            const V3 no = cross(o1 - o0, o2 - o0);
            const double nno = norm(no);
            if (nno > 1.0e-12) {
              if (A2 < 1.0e-12) {
                guard = 1.0;
              } else {
                double dd = dot(no, n) / (nno * A2);
                dd = fmin(1.0, fmax(-1.0, dd));
                if (!(acos(dd) <= 0.5)) guard = 1.0;
              }
            }
          }
        }
        if (BEND) {
          // tilt_kernels.f90:133-151 with area_doubled = max(|n|, 1e-12)
          const double ad = A2 < 1.0e-12 ? 1.0e-12 : A2;
#if MS_RSQRT_NORM
          const double inv_ad = A2 < 1.0e-12 ? 1.0e12 : inv_A2;
#else
          const double inv_ad = 1.0 / ad;
#endif
          const double tri_area = 0.5 * ad;
          const double c0 = dot(-e1, e2) * inv_ad, c1 = dot(-e2, e0) * inv_ad, c2 = dot(-e0, e1) * inv_ad;
          corner_areas(c0, c1, c2, l0, l1, l2, tri_area, va0, va1, va2);
          // bending_utils.py:85-119: tri_areas = max(0.5 |n|, 1e-12)
          ve0 = va0;
          ve1 = va1;
          ve2 = va2;
          const double ta_eff = fmax(0.5 * A2, 1.0e-12);
          if (ta_eff != tri_area) corner_areas(c0, c1, c2, l0, l1, l2, ta_eff, ve0, ve1, ve2);
          // bending_utils.py:121-153 boundary -> interior redistribution
          if (a.m.has_boundary) {  // uniform: closed surfaces skip the whole block
            const int b0 = (lfl[tf.l0] & VF_BOUNDARY) ? 1 : 0;
            const int b1 = (lfl[tf.l1] & VF_BOUNDARY) ? 1 : 0;
            const int b2 = (lfl[tf.l2] & VF_BOUNDARY) ? 1 : 0;
            const int n_int = 3 - (b0 + b1 + b2);
            if (n_int > 0 && n_int < 3) {
              const double b_sum = ve0 * b0 + ve1 * b1 + ve2 * b2;
              const double extra = b_sum / (double)n_int;
              const double m0 = b0 ? 0.0 : 1.0, m1 = b1 ? 0.0 : 1.0, m2 = b2 ? 0.0 : 1.0;
              ve0 = ve0 * m0 + m0 * extra;
              ve1 = ve1 * m1 + m1 * extra;
              ve2 = ve2 * m2 + m2 * extra;
            }
          }
          // K0 + K1 + K2 = 0 for every facet (each edge term enters two corners with opposite sign)
          const double hc0 = 0.5 * c0, hc1 = 0.5 * c1, hc2 = 0.5 * c2;
          const V3 K0 = hc2 * e2 - hc1 * e1;
          const V3 K1 = hc0 * e0 - hc2 * e2;
          const V3 K2 = mk(-K0.x - K1.x, -K0.y - K1.y, -K0.z - K1.z);
          if (ATOMIC) {
            const int no = t.n_owned;
            if (tf.l0 < no) {
              atomicAdd(&stg[tf.l0], K0.x); atomicAdd(&stg[T + tf.l0], K0.y); atomicAdd(&stg[2 * T + tf.l0], K0.z);
              atomicAdd(&stg[3 * T + tf.l0], va0);
              if (ve0 != va0) atomicAdd(&stg[4 * T + tf.l0], ve0 - va0);
            }
            if (tf.l1 < no) {
              atomicAdd(&stg[tf.l1], K1.x); atomicAdd(&stg[T + tf.l1], K1.y); atomicAdd(&stg[2 * T + tf.l1], K1.z);
              atomicAdd(&stg[3 * T + tf.l1], va1);
              if (ve1 != va1) atomicAdd(&stg[4 * T + tf.l1], ve1 - va1);
            }
            if (tf.l2 < no) {
              atomicAdd(&stg[tf.l2], K2.x); atomicAdd(&stg[T + tf.l2], K2.y); atomicAdd(&stg[2 * T + tf.l2], K2.z);
              atomicAdd(&stg[3 * T + tf.l2], va2);
              if (ve2 != va2) atomicAdd(&stg[4 * T + tf.l2], ve2 - va2);
            }
          } else {
            double* s = stg + tid;
            s[0 * T] = K0.x; s[1 * T] = K0.y; s[2 * T] = K0.z;
            s[3 * T] = K1.x; s[4 * T] = K1.y; s[5 * T] = K1.z;
            s[6 * T] = K2.x; s[7 * T] = K2.y; s[8 * T] = K2.z;
          }
        }
      }
    }
    if (BEND && !ATOMIC) {
      // sub-phase A: curvature vectors; sub-phase B: corner areas (same 9*T buffer)
      const int lo = c0 - t.f0, hi = min(c0 + T, t.f1) - t.f0;
      __syncthreads();
      const int cur0 = cur;
      while (cur < end) {
        const int ent = vent[cur];
        const int fl = ent >> 2;
        if (fl >= hi) break;
        const double* s = stg + (fl - lo) + 3 * (ent & 3) * T;
        aKx += s[0];
        aKy += s[T];
        aKz += s[2 * T];
        ++cur;
      }
      __syncthreads();
      if (p < t.f1) {
        double* s = stg + tid;
        s[0 * T] = va0; s[1 * T] = va1; s[2 * T] = va2;
        s[3 * T] = ve0; s[4 * T] = ve1; s[5 * T] = ve2;
      }
      __syncthreads();
      for (int q = cur0; q < cur; ++q) {
        const int ent = vent[q];
        const double* s = stg + ((ent >> 2) - lo) + (ent & 3) * T;
        aAv += s[0];
        aAe += s[3 * T];
      }
      __syncthreads();
    }
  }

  if (BEND && ATOMIC) {
    __syncthreads();
    MS_STAMP(2);
    if (tid < t.n_owned) {
      aKx = stg[tid];
      aKy = stg[T + tid];
      aKz = stg[2 * T + tid];
      aAv = stg[3 * T + tid];
      // A_eff differs from A_vor only through the boundary redistribution and the degenerate-area clamp:
      // the fifth column holds that (mostly empty) difference, saving one LDS atomic per corner
      aAe = aAv + stg[4 * T + tid];
    }
    __syncthreads();  // red aliases stg
  }
  double e_bend = 0.0;
  if (BEND && tid < t.n_owned) {
    // modules/energy/bending.py:111-161 on this thread's owned vertex
    const int v = t.v_lo + tid;
    const V3 K = mk(aKx, aKy, aKz);
    const double kappa = kappa_v, c0 = c0_v;
    const bool interior = !(own_fl & VF_BOUNDARY);
    const double safe = fmax(aAv, 1.0e-12);
    // |K| and 1/|K| from one refined rsqrt (as for the facet normals); |K| <= 1e-15 gives (0, 0) and the rare path
    double k_mag, inv_k_mag;
    norm_and_inverse(dot(K, K), k_mag, inv_k_mag);
    // one reciprocal serves H and the area ratio (each fp64 division is ~14 VALU instructions)
    const double inv_safe = 1.0 / safe;
    double H = k_mag * (0.5 * inv_safe);
    // leaflet bending_tilt (bending_tilt_leaflet.py:455-459): oriented curvature H = (K . n)/(2 A) with
    // the unit vertex normal, and K_dir = n (:574-575)
    const bool signed_h = a.bt_normals != nullptr;
    V3 nh = mk(0, 0, 0);
    if (signed_h) {
      nh = mk(a.bt_normals[3 * (size_t)v], a.bt_normals[3 * (size_t)v + 1], a.bt_normals[3 * (size_t)v + 2]);
      H = dot(K, nh) * (0.5 * inv_safe);
    }
    const double ratio = safe > 1.0e-15 ? aAe * inv_safe : 0.0;
    double scale_K, fe, fv;
    if (a.bt_vert) {
      // bending_tilt.py:217-233: the energy and the factors need div t, which k_bt adds;
      // leave the per-vertex pieces: base term, A_eff, kappa*ratio*H and K_dir*kappa*ratio
      double base = (2.0 * H) - c0;
      if (!interior) base = 0.0;
      scale_K = kappa * ratio;
      fe = fv = 0.0;
      double* rec = a.bt_vert + 4 * (size_t)v;
      rec[0] = base;
      rec[1] = aAe;
      rec[2] = kappa * ratio * H;
      rec[3] = 0.0;
      if (a.bt_vert2 != nullptr) {  // the other leaflet's record: same sums, its own (kappa, c0)
        const double kappa_b = a.kappa2[v], c0_b = a.c02[v];
        double base_b = (2.0 * H) - c0_b;
        if (!interior) base_b = 0.0;
        double* rec_b = a.bt_vert2 + 4 * (size_t)v;
        rec_b[0] = base_b;
        rec_b[1] = aAe;
        rec_b[2] = kappa_b * ratio * H;
        rec_b[3] = 0.0;
      }
    } else if (a.bending_model == MS_BEND_HELFRICH) {
      double term = (2.0 * H) - c0;
      if (!interior) term = 0.0;
      e_bend = 0.5 * (kappa * (term * term) * aAe);
      scale_K = kappa * term * ratio;
      fe = 0.5 * kappa * (term * term);
      fv = -2.0 * kappa * term * ratio * H;
    } else {
      const double He = interior ? H : 0.0;
      e_bend = kappa * (He * He) * aAe;
      scale_K = kappa * He * ratio;
      fe = kappa * (He * He);
      fv = -2.0 * kappa * (He * He) * ratio;
    }
#ifndef MS_ABL_KA
#define MS_ABL_KA 0  // K_A epilogue ablation (timing only): 1 no factor stores, 2 no rare vertex-normal path, 3 both
#endif
    if (a.fK) {
      V3 Kd;
      if (signed_h) {
        Kd = nh;
      } else if ((MS_ABL_KA & 2) || k_mag > 1.0e-15) {
        const double inv_k = (MS_ABL_KA & 2) ? 1.0 / k_mag : inv_k_mag;
        Kd = mk(K.x * inv_k, K.y * inv_k, K.z * inv_k);
      } else {
        // bending.py:154-158 falls back to the vertex normal (bending_utils.py:13-34)
        // where K vanishes (flat patches): sum the incident facet normals now.
#if MS_STAMPS
        atomicAdd(&g_stamps[8 * 16384 - 1], 1ull);
#endif
        V3 N = mk(0, 0, 0);
        int qb = ent_begin, qe = end;
        const uint16_t* ve = vent;
        if (ATOMIC) {  // (rare path) the CSR was not staged: read it where it lives
          const uint16_t* gv = a.m.tile_voff + (size_t)t.tile * (T + 1);
          qb = gv[tid];
          qe = gv[tid + 1];
          ve = a.m.vent + t.e0;
        }
        for (int q = qb; q < qe; ++q) {
          const TileFacet f = a.m.tile_facets[t.f0 + (ve[q] >> 2)];
          const V3 q0 = lds_row3(px, f.l0), q1 = lds_row3(px, f.l1), q2 = lds_row3(px, f.l2);
          N = N + cross(q1 - q0, q2 - q0);
        }
        const double nn = norm(N);
        Kd = nn > 1.0e-15 ? mk(N.x / nn, N.y / nn, N.z / nn) : N;
      }
#if (MS_ABL_KA & 1)
      e_bend += 1e-300 * (Kd.x * scale_K + Kd.y * scale_K + Kd.z * scale_K + fe + fv);
#else
      a.fK[3 * (size_t)v] = Kd.x * scale_K;
      a.fK[3 * (size_t)v + 1] = Kd.y * scale_K;
      a.fK[3 * (size_t)v + 2] = Kd.z * scale_K;
      a.fA[2 * (size_t)v] = fe;
      a.fA[2 * (size_t)v + 1] = fv;
#endif
    }
  }

  double* pout = a.partials + t.tile;
  const size_t pstride = (size_t)a.m.n_tiles;
  if (GUARD || want_vol) {
    const double vals[5] = {e_surf, vol, e_bend, min_e2, guard};
    const int ops[5] = {0, 0, 0, 1, 2};
    const int slots[5] = {MS_S_ESURF, MS_S_VOL, MS_S_EBEND, MS_S_MINEDGE2, MS_S_GUARD};
    block_reduce_store<5>(vals, ops, slots, red, pout, pstride);
  } else {
    // volume / guard partials are identically zero: store them without reducing
    const double vals[3] = {e_surf, e_bend, min_e2};
    const int ops[3] = {0, 0, 1};
    const int slots[3] = {MS_S_ESURF, MS_S_EBEND, MS_S_MINEDGE2};
    block_reduce_store<3>(vals, ops, slots, red, pout, pstride);
    if (tid == 0) {
      pout[MS_S_VOL * pstride] = 0.0;
      pout[MS_S_GUARD * pstride] = 0.0;
    }
  }
#if MS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MS_STAMP(3);
#endif
}

template <bool BEND, bool GUARD, int TT, int CAPC, bool ATOMIC, int MULTI = 0>
__global__ __launch_bounds__(TT ? TT : 512, (TT == 256 && BEND && !GUARD && ATOMIC) ? MS_KA_SLOTS : 1) MS_WPE_ENERGY void k_energy(EnergyArgs a, int cap_rt, int max_ent) {
  extern __shared__ double lds[];
  energy_body<BEND, GUARD, TT, CAPC, ATOMIC, MULTI>(a, cap_rt, max_ent, lds, (int)blockIdx.x);
}

static size_t u16_bytes(int T, int max_ent) { return 2 * ((size_t)((max_ent + 3) & ~3)); }

size_t energy_lds_bytes(int T, int cap, int max_ent, bool bend, bool guard, bool flags, bool atomic) {
  size_t d = 3 * (size_t)cap + (guard ? 3 * (size_t)cap : 0) + (bend ? (atomic ? 5 : 9) * (size_t)T : 5 * 16);
  return d * sizeof(double) + ((bend && !atomic) ? u16_bytes(T, max_ent) : 0) +
         ((flags || guard) ? (((size_t)cap + 15) / 16) * 16 : 0);
}

static bool no_lean() {  // MS_NO_LEAN=1: A/B switch for the lean gradient instance
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_NO_LEAN");
    v = (e && atoi(e) != 0) ? 1 : 0;
  }
  return v != 0;
}

template <typename K>
static hipError_t ensure_lds(K kernel, size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

constexpr int FAST_T = 256;  // specialised tile size (LDS staging offsets become immediates)
constexpr int FAST_CAP = 0;  // patch capacity stays a runtime value: a fixed 512 slots would push
                             // the gradient kernel from 3 to 2 workgroups per CU (LDS)

// Variant builds only (-DMS_ABL_NTILES_ENV, tools/build_variant.sh): MS_ABL_NTILES=<n> makes the tile kernels process
// at most n tiles -- a timing experiment with wrong results, so the shipped library does not read the variable at all
static int abl_ntiles() {
#ifdef MS_ABL_NTILES_ENV
  static const int v = getenv("MS_ABL_NTILES") ? atoi(getenv("MS_ABL_NTILES")) : 0;
  return v;
#else
  return 0;
#endif
}

hipError_t launch_energy(const EnergyArgs& a_in, bool guard, int cap, int max_ent, hipStream_t s) {
  EnergyArgs a = a_in;
  if (abl_ntiles() > 0) a.tile1 = std::min(a.tile1, a.tile0 + abl_ntiles());
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const bool bend = (a.modules & (MS_MOD_BENDING | MS_MOD_BENDING_TILT)) != 0;
  const bool fast = a.m.T == FAST_T && a.m.tile_facets32 != nullptr && !a.m.no_fast;  // (T = 256 instances: packed records)
  // atomic: per-vertex sums by LDS ds_add_f64 instead of the staged CSR gather -- one barrier per
  // tile instead of twelve and half the LDS traffic, at the price of a summation order that
  // varies from run to run (ms_set_deterministic).  Without bending there are no vertex sums.
  const bool atomic = a.atomic != 0 && bend;
  // MS_KA_LDS_MIN=<bytes> (variant builds only): request at least this much LDS per workgroup (caps the workgroups per CU)
  static const size_t lds_min = variant_env("MS_KA_LDS_MIN") ? (size_t)atol(variant_env("MS_KA_LDS_MIN")) : 0;
  const size_t lds = std::max(lds_min, energy_lds_bytes(a.m.T, cap, max_ent, bend, guard, a.m.has_boundary != 0, atomic));
  hipError_t e;
  if (ExecRecorder* r = exec_find(s)) {
    // one-workgroup interpreter: the T = 256 instances, one trial per launch (the context switched multi-trial launches
    // off); anything else runs as an ordinary launch behind what has been recorded
    if (fast && !a.pair)
      return r->push(CK_ENERGY, 0, cap, max_ent, nb, (bend ? 1u : 0u) | (guard ? 2u : 0u) | (atomic ? 4u : 0u), lds, &a,
                     sizeof(a), &a.m);
    e = r->flush();
    if (e != hipSuccess) return e;
  }
#define MS_LAUNCH_E(B, G, TT, CC, AT)                                                              \
  do {                                                                                             \
    e = ensure_lds(k_energy<B, G, TT, CC, AT>, lds);                                               \
    if (e != hipSuccess) return e;                                                                 \
    hipLaunchKernelGGL((k_energy<B, G, TT, CC, AT>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent); \
  } while (0)
#define MS_PICK_E(B, G, AT)                               \
  do {                                                    \
    if (fast) MS_LAUNCH_E(B, G, FAST_T, FAST_CAP, AT);    \
    else MS_LAUNCH_E(B, G, 0, 0, AT);                     \
  } while (0)
  if (a.pair) {
    // pair launch: bending factors on, no guard (the caller checked both)
    if (!bend || guard) return hipErrorInvalidValue;
    if (a.pair < 2 || a.pair > MS_MAX_TRIALS) return hipErrorInvalidValue;
    const int nb2 = a.pair * NXCD * ((nb + NXCD - 1) / NXCD);
#define MS_LAUNCH_P(TT, CC, AT, NM)                                                                          \
  do {                                                                                                       \
    e = ensure_lds(k_energy<true, false, TT, CC, AT, NM>, lds);                                              \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_energy<true, false, TT, CC, AT, NM>), dim3(nb2), dim3(a.m.T), lds, s, a, cap, max_ent); \
  } while (0)
#define MS_PICK_P(NM)                                                                  \
  do {                                                                                 \
    if (atomic) {                                                                      \
      if (fast) MS_LAUNCH_P(FAST_T, FAST_CAP, true, NM); else MS_LAUNCH_P(0, 0, true, NM);   \
    } else {                                                                           \
      if (fast) MS_LAUNCH_P(FAST_T, FAST_CAP, false, NM); else MS_LAUNCH_P(0, 0, false, NM); \
    }                                                                                  \
  } while (0)
    if (a.pair == 2) MS_PICK_P(2); else if (a.pair == 3) MS_PICK_P(3); else MS_PICK_P(8);
#undef MS_PICK_P
#undef MS_LAUNCH_P
    return hipGetLastError();
  }
  if (bend && atomic) {
    if (guard) MS_PICK_E(true, true, true); else MS_PICK_E(true, false, true);
  } else if (bend) {
    if (guard) MS_PICK_E(true, true, false); else MS_PICK_E(true, false, false);
  } else {
    if (guard) MS_PICK_E(false, true, false); else MS_PICK_E(false, false, false);
  }
#undef MS_PICK_E
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
//             (modules/energy/bending_gradient.py:17-175); cotans recomputed as
//             in tilt_kernels.f90:140-151, never stored.
// The three grad_cotan calls (bending_kernels.f90:32-74) share w = u x v = n and
// S = |n| (the same vector for all corners of a triangle), so a facet costs one
// sqrt and two divides.
// BENDMODE: 0 none, 1 analytic, 2 approx (bending.py:163-167).
// LDS: px[3][cap] | (BEND) fk[3][cap] fae[cap] fav[cap] | stg[9 or 18][T] | red[4*16]
//      | vent[max_ent] (u16) | fl[cap] (u8)
// ---------------------------------------------------------------------------
// Scheduling fences between the stages of the facet arithmetic: without them the compiler hoists every LDS gather
// of a facet to the top, which keeps ~50 more registers live and costs a resident workgroup per CU.
#ifndef MS_SCHED_FENCES
#define MS_SCHED_FENCES 1
#endif
#if MS_SCHED_FENCES
#define MS_SCHED_FENCE()                 \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#else
#define MS_SCHED_FENCE() do {} while (0)
#endif
// Phase-ablation switches (tools/kprobe.py; results are WRONG with any of them on -- timing only).
#ifndef MS_ABL_NOATOM
#define MS_ABL_NOATOM 0   // per-corner LDS atomics replaced by a register sink
#endif
#ifndef MS_ABL_NOMATH
#define MS_ABL_NOMATH 0   // facet arithmetic replaced by sums of the gathered values
#endif
#ifndef MS_ABL_NOGATHER
#define MS_ABL_NOGATHER 0 // corner rows synthesised in registers instead of read from LDS
#endif
#ifndef MS_ABL_NOLOOP
#define MS_ABL_NOLOOP 0   // no facet loop at all (stage-in + epilogue skeleton)
#endif
#ifndef MS_ABL_NOSTORE
#define MS_ABL_NOSTORE 0  // epilogue row stores dropped
#endif
#ifndef MS_ABL_NOHIST
#define MS_ABL_NOHIST 0   // CG history rows not loaded
#endif
// LEAN: the instance of the headline loop -- uniform surface tension (no per-facet gamma registers) and no separate
// previous-direction rows (dir_mode != 2, or pd = -pg after an implicit steepest-descent step): 8 registers fewer live
// through the facet loop, which is what lets the kernel fit 128 VGPRs = 4 resident workgroups per CU instead of 3.
template <int BENDMODE, bool VOLROW, int TT, int CAPC, bool ATOMIC, bool LEAN = false>
__device__ __forceinline__ void gradient_body(const GradientArgs& a, int cap_rt, int max_ent, double* lds, int block_id) {
  constexpr bool BEND = BENDMODE != 0;
  // BENDMODE 3: leaflet bending_tilt (bt_gradient.py:89-389): analytic back-propagation whose effective-area
  // factor is per corner, 1/2 kappa_k (base_k + s div_f t)^2; `fae` then holds base_k
  constexpr bool LEAF = BENDMODE == 3;
  constexpr bool ANALYTIC = BENDMODE == 1 || BENDMODE == 3;
  const int T = TT ? TT : a.m.T;
  const int cap = CAPC ? CAPC : cap_rt;
  // rows: px[cap][3] positions | fk[cap][3] | fa[cap][2] = (fA_eff or base, fA_vor) -- the layouts of the global arrays
  double* px = lds;
  double* fk = px + 3 * cap;
  double* fa = fk + (BEND ? 3 * cap : 0);
  double* kp = fa + (BEND ? 2 * cap : 0);     // LEAF: kappa
  double* tl = kp + (LEAF ? cap : 0);         // LEAF: tilts
  double* stg = tl + (LEAF ? 3 * cap : 0);
  // ATOMIC: stg holds the per-vertex accumulators (ds_add_f64) instead of per-corner columns
  // MS_ROW_TWO_PASS (A/B build): the constraint row's gC through the SAME three accumulator columns in a second sweep
  // over the facets (cross products of the staged positions only) instead of three columns of its own -- five
  // workgroups per CU instead of four
  constexpr bool ROW2 = MS_ROW_TWO_PASS != 0 && VOLROW && ATOMIC;
  constexpr int NACC = (VOLROW && !ROW2) ? 6 : 3;  // ATOMIC: gradient (and constraint-row) accumulator columns
  double* red = stg + (ATOMIC ? NACC : (VOLROW ? 18 : 9)) * T;
  uint16_t* vent = reinterpret_cast<uint16_t*>(red + 4 * 16);
  uint8_t* lfl = reinterpret_cast<uint8_t*>(vent + (ATOMIC ? 0 : ((max_ent + 3) & ~3)));
  // queued behind a line search: runs only if the decision word says the accepted point is the one in the ordinary
  // buffers (DEC_ACCEPT_MAIN)
  const int my_tile = a.tile0 + xcd_tile(block_id, a.tile1 - a.tile0);
#if MS_GATE_PROBE
  if (a.gate != nullptr) {
    const double truth = ld_agent(a.scal + MS_S_EBEND);
    const double via_scalar = g_shadow[MS_S_EBEND];
    const double via_vector = *reinterpret_cast<const volatile double*>(g_shadow + MS_S_EBEND + (threadIdx.x >> 12));
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;  // XCC_ID
      atomicAdd(&g_probe[0], 1ull);
      if (via_scalar != truth) {
        atomicAdd(&g_probe[1], 1ull);
        atomicAdd(&g_probe[8 + xcc], 1ull);
      }
      if (via_vector != truth) {
        atomicAdd(&g_probe[2], 1ull);
        atomicAdd(&g_probe[16 + xcc], 1ull);
      }
      if (block_id == 0) atomicAdd(&g_probe[3], 1ull);  // gated launches probed
    }
  }
#endif
  if (a.gate != nullptr && !gate_open(a.gate, a.gate_want, a.partials + (size_t)MS_P_RAN * a.m.n_tiles + my_tile)) return;

  const int blk = block_id;
  (void)blk;
  MS_STAMP(0);
  MS_STAMP_ID();
  const TileCtx t = tile_ctx(a.m, my_tile);
  const int tid = threadIdx.x;

  constexpr bool PACKED = TT == 256;
  FacetRec<PACKED> tf_nx = facet_null<PACKED>();
  double gam_nx = 0.0;
  if (t.f0 + tid < t.f1) {
    tf_nx = facet_load<PACKED>(a.m, (size_t)(t.f0 + tid));
    gam_nx = a.m.gamma_uniform ? a.m.gamma_const : a.m.tf_gamma[t.f0 + tid];
  }

  int cur = 0, end = 0;  // this vertex's CSR range
  // -- stage-in: all independent loads first, then the halo rows, then the LDS writes --
  {
    const bool own = tid < t.n_owned;
    const int v_own = t.v_lo + tid;
    const bool has_h = tid < t.nh;
    int hv = 0;
    if (has_h) hv = a.m.halo_ids[t.h0 + tid];
    double x0 = 0, x1 = 0, x2 = 0, k0 = 0, k1 = 0, k2 = 0, ae = 0, av = 0;
    double okp = 0, ot0 = 0, ot1 = 0, ot2 = 0, hkp = 0, ht0 = 0, ht1 = 0, ht2 = 0;
    uint8_t fl = 0;
    if (own) {
      const size_t g = 3 * (size_t)v_own;
      fl = a.m.vflags[v_own];
      x0 = a.x[g];
      x1 = a.x[g + 1];
      x2 = a.x[g + 2];
      if (BEND) {
        k0 = a.fK[g];
        k1 = a.fK[g + 1];
        k2 = a.fK[g + 2];
        ae = LEAF ? a.bt_vert[4 * (size_t)v_own] : a.fA[2 * (size_t)v_own];
        av = a.fA[2 * (size_t)v_own + 1];
      }
      if (LEAF) {
        okp = a.m.kappa[v_own];
        ot0 = a.tilts[g];
        ot1 = a.tilts[g + 1];
        ot2 = a.tilts[g + 2];
      }
    }
    CsrStage cs;
    if (!ATOMIC) csr_issue(cs, a.m, t, T, tid);
    if (ATOMIC) {
      for (int j = tid; j < NACC * T; j += T) stg[j] = 0.0;
    }
    double hx0 = 0, hx1 = 0, hx2 = 0, hk0 = 0, hk1 = 0, hk2 = 0, hae = 0, hav = 0;
    uint8_t hfl = 0;
    if (has_h) {
      const size_t g = 3 * (size_t)hv;
      hfl = a.m.vflags[hv];
      hx0 = a.x[g];
      hx1 = a.x[g + 1];
      hx2 = a.x[g + 2];
      if (BEND) {
        hk0 = a.fK[g];
        hk1 = a.fK[g + 1];
        hk2 = a.fK[g + 2];
        hae = LEAF ? a.bt_vert[4 * (size_t)hv] : a.fA[2 * (size_t)hv];
        hav = a.fA[2 * (size_t)hv + 1];
      }
      if (LEAF) {
        hkp = a.m.kappa[hv];
        ht0 = a.tilts[g];
        ht1 = a.tilts[g + 1];
        ht2 = a.tilts[g + 2];
      }
    }
    if (own) {
      lfl[tid] = fl;
      lds_put3(px, tid, x0, x1, x2);
      if (BEND) {
        lds_put3(fk, tid, k0, k1, k2);
        if (LEAN) {
          fa[tid] = ae + av;  // the lean instance needs only C = fA_eff + fA_vor (no boundary rows, no leaflet factor)
        } else {
          fa[2 * tid] = ae;
          fa[2 * tid + 1] = av;
        }
      }
      if (LEAF) {
        kp[tid] = okp;
        tl[tid] = ot0;
        tl[cap + tid] = ot1;
        tl[2 * cap + tid] = ot2;
      }
    }
    if (!ATOMIC) {
      csr_commit(cs, a.m, t, T, tid, vent);
      cur = cs.vo0;
      end = cs.vo1;
    }
    if (has_h) {
      const int s = t.n_owned + tid;
      lfl[s] = hfl;
      lds_put3(px, s, hx0, hx1, hx2);
      if (BEND) {
        lds_put3(fk, s, hk0, hk1, hk2);
        if (LEAN) {
          fa[s] = hae + hav;
        } else {
          fa[2 * s] = hae;
          fa[2 * s + 1] = hav;
        }
      }
      if (LEAF) {
        kp[s] = hkp;
        tl[s] = ht0;
        tl[cap + s] = ht1;
        tl[2 * cap + s] = ht2;
      }
    }
    for (int h = tid + T; h < t.nh; h += T) {  // halo longer than the workgroup (small tiles)
      const int v = a.m.halo_ids[t.h0 + h];
      const int s = t.n_owned + h;
      lfl[s] = a.m.vflags[v];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        px[3 * s + c] = a.x[3 * (size_t)v + c];
        if (BEND) fk[3 * s + c] = a.fK[3 * (size_t)v + c];
        if (LEAF) tl[c * cap + s] = a.tilts[3 * (size_t)v + c];
      }
      if (BEND && LEAN) {
        fa[s] = a.fA[2 * (size_t)v] + a.fA[2 * (size_t)v + 1];
      } else if (BEND) {
        fa[2 * s] = LEAF ? a.bt_vert[4 * (size_t)v] : a.fA[2 * (size_t)v];
        fa[2 * s + 1] = a.fA[2 * (size_t)v + 1];
      }
      if (LEAF) kp[s] = a.m.kappa[v];
    }
  }
  __syncthreads();
  MS_STAMP(1);

  const bool surf = a.modules & MS_MOD_SURFACE;
  const bool volpen = !LEAN && (a.modules & MS_MOD_VOLUME_PENALTY);  // (the lean instances have no penalty term)
  double pen_factor = 0.0;
  if (volpen) pen_factor = a.volume_stiffness * (ld_agent(a.scal + MS_S_VOL) - a.target_volume) / 6.0;

  double gx = 0, gy = 0, gz = 0, cx = 0, cy = 0, cz = 0;
  if (tid >= t.n_owned) cur = end = 0;
  // CG history rows of the fused direction pass: requested now, consumed in the epilogue
  V3 h_pg = mk(0, 0, 0), h_pd = mk(0, 0, 0);
  if (LEAN) {
    // The lean instance has no registers to park the history row in for the length of the facet loop (the compiler
    // spills exactly these six).  Instead one wave pulls the tile's pg rows into L2 now -- a 4-byte LDS-DMA per
    // 128-byte line into the (still unused) reduction scratch, no destination registers -- and the rows are loaded
    // after the loop, from L2.
    if (!MS_ABL_NOHIST && a.dir_mode == 2 && tid < 128) {
      // (wave 0: the previous gradient's rows; wave 1: the previous direction's, unless that is minus the gradient)
      const int n_lines = (t.n_owned * 24 + 127) >> 7;
      const int l = tid & 63;
      const double* rows = tid < 64 ? a.pg : a.pd;
      if (l < n_lines && (tid < 64 || !a.pd_neg_pg))
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((const char*)rows + (size_t)t.v_lo * 24 + (size_t)l * 128),
            (__attribute__((address_space(3))) void*)(red + (tid < 64 ? 0 : 1)), 4, 0, 0);
    }
  } else if (!MS_ABL_NOHIST && a.dir_mode == 2 && tid < t.n_owned) {
    const size_t o = 3 * (size_t)(t.v_lo + tid);
    h_pg = mk(a.pg[o], a.pg[o + 1], a.pg[o + 2]);
    if (!LEAN) h_pd = a.pd_neg_pg ? -h_pg : mk(a.pd[o], a.pd[o + 1], a.pd[o + 2]);
  }

#if MS_ABL_NOATOM
  double abl_sink = 0.0;
#endif
  for (int c0f = t.f0; c0f < (MS_ABL_NOLOOP ? t.f0 : t.f1); c0f += T) {
    const int p = c0f + tid;
    const TileFacet tf = facet_unpack<PACKED>(tf_nx);
    const double gam = (LEAN && a.m.gamma_uniform) ? a.m.gamma_const : gam_nx;  // (uniform: a kernel argument)
    if (p + T < t.f1) {
      tf_nx = facet_load<PACKED>(a.m, (size_t)(p + T));
      if (!LEAN || !a.m.gamma_uniform) gam_nx = a.m.gamma_uniform ? a.m.gamma_const : a.m.tf_gamma[p + T];
    }
    if (p < t.f1) {
#if MS_ABL_NOGATHER
      const V3 v0 = mk(tf.l0 * 1e-3, gam, 1.0), v1 = mk(0.5, tf.l1 * 1e-3, gam), v2 = mk(gam, 0.25, tf.l2 * 1e-3);
#else
      const V3 v0 = lds_row3(px, tf.l0), v1 = lds_row3(px, tf.l1), v2 = lds_row3(px, tf.l2);
#endif
      // two edges carry the facet: e0 = -(e1 + e2), and every inner product of edges follows from
      // l1 = e1.e1, l2 = e2.e2, d12 = e1.e2
      const V3 e1 = v0 - v2, e2 = v1 - v0;
      V3 G0 = mk(0, 0, 0), G1 = mk(0, 0, 0), G2 = mk(0, 0, 0);
      const V3 n = cross(e1, e2);
#if MS_RSQRT_NORM
      double S, rs;
      norm_and_inverse(dot(n, n), S, rs);
#else
      const double S = norm(n);
#endif
      // Every geometric contribution of this facet has the form
      //     G_k = (a_k1 e1 + a_k2 e2) + R (e_k x n) [+ the -L fK difference vectors]:
      // surface: -(gamma/2S) (e_k x n); grad-cot corner k with weight w_k: +-(w_k/S) e_j and
      // +-(w_k cot_k/S^2) (e_j x n) -- collected over the three corners the (e_j x n) coefficients are the SAME
      // scalar R for all three vertices (sum_k e_k = 0); obtuse area term: -(factor/2S) (e_k x n); the six edge
      // terms q_k e_k.  Collecting scalars first leaves 2 cross products and ~10 vector FMAs per facet instead
      // of 9 cross products; sum_k G_k = 0 (translation invariance) gives the third vertex for free.
#if MS_RSQRT_NORM
      const double invS = rs;
#else
      const double invS = S > 1.0e-15 ? 1.0 / S : 0.0;
#endif
      double R = 0.0;                       // coefficient of (e_k x n)
      double a01 = 0, a02 = 0, a11 = 0, a12 = 0;  // e-part of G0, G1 in the basis (e1, e2)
      V3 T0 = mk(0, 0, 0), T1 = mk(0, 0, 0);     // -L fK part of G0, G1
      if (surf && S >= 1.0e-12) R = -(0.5 * gam) * invS;  // g_k = gamma/2 (v_{k+1}-v_{k+2}) x nhat
      if ((VOLROW || volpen) && (tf.flags & TF_BODY)) {
        const V3 w0 = cross(v1, v2), w1 = cross(v2, v0), w2 = cross(v0, v1);
        if (volpen) {
          G0 = G0 + pen_factor * w0;
          G1 = G1 + pen_factor * w1;
          G2 = G2 + pen_factor * w2;
        }
        if (VOLROW && ATOMIC && !ROW2) {
          const double s6 = 1.0 / 6.0;
          const int no = t.n_owned;
          double* c = stg + 3 * T;
          if (tf.l0 < no) { atomicAdd(&c[tf.l0], s6 * w0.x); atomicAdd(&c[T + tf.l0], s6 * w0.y); atomicAdd(&c[2 * T + tf.l0], s6 * w0.z); }
          if (tf.l1 < no) { atomicAdd(&c[tf.l1], s6 * w1.x); atomicAdd(&c[T + tf.l1], s6 * w1.y); atomicAdd(&c[2 * T + tf.l1], s6 * w1.z); }
          if (tf.l2 < no) { atomicAdd(&c[tf.l2], s6 * w2.x); atomicAdd(&c[T + tf.l2], s6 * w2.y); atomicAdd(&c[2 * T + tf.l2], s6 * w2.z); }
        } else if (VOLROW) {
          const double s6 = 1.0 / 6.0;
          double* s = stg + 9 * T + tid;
          s[0 * T] = s6 * w0.x; s[1 * T] = s6 * w0.y; s[2 * T] = s6 * w0.z;
          s[3 * T] = s6 * w1.x; s[4 * T] = s6 * w1.y; s[5 * T] = s6 * w1.z;
          s[6 * T] = s6 * w2.x; s[7 * T] = s6 * w2.y; s[8 * T] = s6 * w2.z;
        }
      } else if (VOLROW && !ATOMIC) {
        double* s = stg + 9 * T + tid;
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k * T] = 0.0;
      }
#if MS_ABL_NOMATH
      if (BEND) {
        const V3 k0 = lds_row3(fk, tf.l0), k1 = lds_row3(fk, tf.l1), k2 = lds_row3(fk, tf.l2);
        G0 = (fa[2 * tf.l0] + fa[2 * tf.l0 + 1]) * (k0 - (e1 + e2));
        G1 = (fa[2 * tf.l1] + fa[2 * tf.l1 + 1]) * (k1 + e1);
        G2 = (fa[2 * tf.l2] + fa[2 * tf.l2 + 1]) * (k2 + e2);
      }
      if (false) {
#else
      if (BEND) {
#endif
        // cotans exactly as compute_curvature_data produces `weights`
        const double inv_ad = S < 1.0e-12 ? 1.0e12 : invS;  // 1 / max(S, 1e-12)
        const double l1 = dot(e1, e1), l2 = dot(e2, e2), d12 = dot(e1, e2);
        const double d20 = -(d12 + l2), d01 = -(l1 + d12);  // e2.e0, e0.e1
        // half cotans (cot_k = -d / max(S, 1e-12)); the powers of two of the formulas below are folded into as few
        // factors as possible -- exact scalings, so the values are those of the formulas as written in the reference
        const double h_inv = 0.5 * inv_ad;
        const double hc0 = -d12 * h_inv, hc1 = -d20 * h_inv, hc2 = -d01 * h_inv;
        // term 1: -L fK  (bending_kernels.f90:118-129): G0 -= b + c, G1 -= a - c, G2 += a + b, and the term-2
        // weights (bending_gradient.py:37-42; (v1-v2) == -e0 etc.), one component of fK at a time so that the
        // nine factor values are never all live
        double w0 = 0.0, w1 = 0.0, w2 = 0.0;
        MS_SCHED_FENCE();
        {
#if MS_ABL_NOGATHER
          const double kk[9] = {gam, tf.l0 * 1e-3, 2.0, tf.l1 * 1e-3, 1.5, gam, 0.75, gam, tf.l2 * 1e-3};
          const double *r0 = kk, *r1 = kk + 3, *r2 = kk + 6;
#else
          const double *r0 = fk + 3 * tf.l0, *r1 = fk + 3 * tf.l1, *r2 = fk + 3 * tf.l2;
#endif
          double t0[3], t1[3];
          const double e1c[3] = {e1.x, e1.y, e1.z}, e2c[3] = {e2.x, e2.y, e2.z};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double k0 = r0[c], k1 = r1[c], k2 = r2[c];
            const double f12 = k1 - k2, f02 = k0 - k2, f01 = k0 - k1;
            t0[c] = -(hc1 * f02 + hc2 * f01);
            t1[c] = hc2 * f01 - hc0 * f12;
            w0 -= f12 * (e1c[c] + e2c[c]);  // f12 . e0
            w1 -= f02 * e1c[c];
            w2 += f01 * e2c[c];
          }
          T0 = mk(t0[0], t0[1], t0[2]);
          T1 = mk(t1[0], t1[1], t1[2]);
        }
        MS_SCHED_FENCE();
        if (ANALYTIC) {
          // from here on w_k stands for TWICE the weight of grad cot_k (the halves go into hS below)
          // term 3 coefficients (bending_gradient.py:80-95)
          int t0 = 1, t1 = 1, t2 = 1;
          if (!LEAN && a.m.has_boundary) {
            t0 = (lfl[tf.l0] & VF_BOUNDARY) ? 0 : 1;
            t1 = (lfl[tf.l1] & VF_BOUNDARY) ? 0 : 1;
            t2 = (lfl[tf.l2] & VF_BOUNDARY) ? 0 : 1;
          }
          const int cnt = t0 + t1 + t2;
#if MS_ABL_NOGATHER
          double fe0 = gam, fe1 = 2 * gam, fe2 = 3 * gam;
          const double fv0 = gam, fv1 = gam, fv2 = gam;
#else
          // (fA_eff, fA_vor) of a corner is one 16-byte row; the lean instance staged their sum
          double2 fa0, fa1, fa2;
          if (LEAN) {
            fa0 = make_double2(fa[tf.l0], 0.0);
            fa1 = make_double2(fa[tf.l1], 0.0);
            fa2 = make_double2(fa[tf.l2], 0.0);
          } else {
            fa0 = *reinterpret_cast<const double2*>(fa + 2 * tf.l0);
            fa1 = *reinterpret_cast<const double2*>(fa + 2 * tf.l1);
            fa2 = *reinterpret_cast<const double2*>(fa + 2 * tf.l2);
          }
          double fe0 = fa0.x, fe1 = fa1.x, fe2 = fa2.x;
          const double fv0 = fa0.y, fv1 = fa1.y, fv2 = fa2.y;
#endif
          if (LEAF) {
            // per-corner factor 1/2 kappa_k (base_k + s div_f t)^2 (bending_tilt_leaflet.py:608-610);
            // div_f t = sum_k t_k . (n x e_k) / max(|n|^2, 1e-20) (tilt_operators.py:191-330)
            const double n2 = dot(n, n);
            const double den = n2 > 1.0e-20 ? n2 : 1.0e-20;
            const V3 e0 = -(e1 + e2);
            const V3 q0 = cross(n, e0), q1 = cross(n, e1), q2 = cross(n, e2);
            const V3 g0 = mk(q0.x / den, q0.y / den, q0.z / den);
            const V3 g1 = mk(q1.x / den, q1.y / den, q1.z / den);
            const V3 g2 = mk(q2.x / den, q2.y / den, q2.z / den);
            const double dv = a.div_sign * (dot(lds_v3(tl, cap, tf.l0), g0) + dot(lds_v3(tl, cap, tf.l1), g1) +
                                            dot(lds_v3(tl, cap, tf.l2), g2));
            const double u0 = fe0 + dv, u1 = fe1 + dv, u2 = fe2 + dv;
            fe0 = 0.5 * kp[tf.l0] * (u0 * u0);
            fe1 = 0.5 * kp[tf.l1] * (u1 * u1);
            fe2 = 0.5 * kp[tf.l2] * (u2 * u2);
          }
          double C0 = LEAN ? fe0 : fe0 + fv0, C1 = LEAN ? fe1 : fe1 + fv1, C2 = LEAN ? fe2 : fe2 + fv2;
          if (!LEAN && cnt < 3) {  // some corner on the boundary: it takes the interior corners' mean
            const double avg = cnt > 0 ? (fe0 * t0 + fe1 * t1 + fe2 * t2) / (double)cnt : 0.0;
            C0 = (t0 ? fe0 : avg) + fv0;
            C1 = (t1 ? fe1 : avg) + fv1;
            C2 = (t2 ? fe2 : avg) + fv2;
          }
          const bool obtuse = (hc0 < 0.0) || (hc1 < 0.0) || (hc2 < 0.0);
          double q0 = 0.0, q1 = 0.0, q2 = 0.0;
          if (!obtuse) {
            // six edge terms (:107-124): G0 += q1 e1 - q2 e2, G1 += q2 e2 - q0 e0, G2 += q0 e0 - q1 e1;
            // the second grad_cotan family (bending_math.py:244-249) has the same arguments as the
            // first, so its weights (:126-152) fold into w_k.
            const double s0 = 0.5 * (C1 + C2), s1 = 0.5 * (C0 + C2), s2 = 0.5 * (C0 + C1);
            q0 = hc0 * s0;  // 1/4 cot_0 (C1 + C2)
            q1 = hc1 * s1;
            q2 = hc2 * s2;
            w0 += (0.5 * ((l1 + l2) + 2.0 * d12)) * s0;  // 2 * 1/8 |e0|^2 (C1 + C2)
            w1 += (0.5 * l1) * s1;
            w2 += (0.5 * l2) * s2;
          } else {
            // (:154-173) grad T with u = v1-v0, v = v2-v0: +factor/(2S) (n x e_k) at vertex k
            double factor = 0.0;
            if (hc0 < 0.0) factor += 0.5 * C0 + 0.25 * C1 + 0.25 * C2;
            if (hc1 < 0.0) factor += 0.5 * C1 + 0.25 * C0 + 0.25 * C2;
            if (hc2 < 0.0) factor += 0.5 * C2 + 0.25 * C0 + 0.25 * C1;
            R -= (0.5 * factor) * invS;
          }
          // grad cot at corner k (bending_kernels.f90:32-74 with w = n for every corner):
          //   corner 0 -> G1 += w0 gu, G2 += w0 gv, G0 -= w0 (gu + gv), gu = -e1/S + kk0 (e1 x n), gv = e2/S + kk0 (e2 x n)
          //   (cyclic); kk_k = cot_k / S^2
          const double invS2 = invS * invS, hS = 0.5 * invS;
          const double p0 = w0 * hS, p1 = w1 * hS, p2 = w2 * hS;
          R += ((w0 * (-d12) + w1 * (-d20)) + w2 * (-d01)) * (invS2 * hS);
          a01 = ((p0 + q1) - p1) + p2;
          a02 = ((-p0 - q2) - p1) + p2;
          a11 = ((p2 - p0) + p1) + q0;
          a12 = (2.0 * p1 + q2) + q0;
        }
      }
      MS_SCHED_FENCE();
      if (!MS_ABL_NOMATH) {
        const V3 Rn = R * n;
        const V3 X1 = cross(e1, Rn), X2 = cross(e2, Rn);
        const V3 H0 = (T0 + (a01 * e1 + a02 * e2)) - (X1 + X2);  // e0 x n = -(e1 + e2) x n
        const V3 H1 = (T1 + (a11 * e1 + a12 * e2)) + X1;
        if (LEAN) {  // nothing was accumulated before (no volume terms in the lean instance)
          G0 = H0;
          G1 = H1;
          G2 = mk(-H0.x - H1.x, -H0.y - H1.y, -H0.z - H1.z);
        } else {
          G0 = G0 + H0;
          G1 = G1 + H1;
          G2 = G2 - (H0 + H1);
        }
      }
#if MS_ABL_NOATOM
      if (ATOMIC) {
        abl_sink += (G0.x + G0.y + G0.z) + (G1.x + G1.y + G1.z) + (G2.x + G2.y + G2.z);
      } else
#endif
      if (ATOMIC) {
        const int no = t.n_owned;
        if (tf.l0 < no) { atomicAdd(&stg[tf.l0], G0.x); atomicAdd(&stg[T + tf.l0], G0.y); atomicAdd(&stg[2 * T + tf.l0], G0.z); }
        if (tf.l1 < no) { atomicAdd(&stg[tf.l1], G1.x); atomicAdd(&stg[T + tf.l1], G1.y); atomicAdd(&stg[2 * T + tf.l1], G1.z); }
        if (tf.l2 < no) { atomicAdd(&stg[tf.l2], G2.x); atomicAdd(&stg[T + tf.l2], G2.y); atomicAdd(&stg[2 * T + tf.l2], G2.z); }
      } else {
        double* s = stg + tid;
        s[0 * T] = G0.x; s[1 * T] = G0.y; s[2 * T] = G0.z;
        s[3 * T] = G1.x; s[4 * T] = G1.y; s[5 * T] = G1.z;
        s[6 * T] = G2.x; s[7 * T] = G2.y; s[8 * T] = G2.z;
      }
    }
    if (ATOMIC) continue;
    __syncthreads();
    {
      const int lo = c0f - t.f0, hi = min(c0f + T, t.f1) - t.f0;
      while (cur < end) {
        const int ent = vent[cur];
        const int fl = ent >> 2;
        if (fl >= hi) break;
        const int k = ent & 3;
        const double* s = stg + (fl - lo);
        gx += s[(3 * k) * T];
        gy += s[(3 * k + 1) * T];
        gz += s[(3 * k + 2) * T];
        if (VOLROW) {
          cx += s[(9 + 3 * k) * T];
          cy += s[(10 + 3 * k) * T];
          cz += s[(11 + 3 * k) * T];
        }
        ++cur;
      }
    }
    __syncthreads();
  }

  if (LEAN && !MS_ABL_NOHIST && a.dir_mode == 2 && tid < t.n_owned) {
    const size_t o = 3 * (size_t)(t.v_lo + tid);
    h_pg = mk(a.pg[o], a.pg[o + 1], a.pg[o + 2]);
    h_pd = a.pd_neg_pg ? -h_pg : mk(a.pd[o], a.pd[o + 1], a.pd[o + 2]);
  }
  if (ATOMIC) {
    __syncthreads();
    MS_STAMP(2);
    if (tid < t.n_owned) {
      gx = stg[tid];
#if MS_ABL_NOATOM
      gx += abl_sink;
#endif
      gy = stg[T + tid];
      gz = stg[2 * T + tid];
      if (VOLROW && !ROW2) {
        cx = stg[3 * T + tid];
        cy = stg[4 * T + tid];
        cz = stg[5 * T + tid];
      }
    }
    if (ROW2) {
      __syncthreads();  // (every owner has read its gradient row)
      if (tid < T) stg[tid] = stg[T + tid] = stg[2 * T + tid] = 0.0;
      __syncthreads();
      const double s6 = 1.0 / 6.0;
      const int no = t.n_owned;
      for (int c0f = t.f0; c0f < t.f1; c0f += T) {
        const int p = c0f + tid;
        if (p >= t.f1) continue;
        const TileFacet tf = facet_unpack<PACKED>(facet_load<PACKED>(a.m, (size_t)p));
        if (!(tf.flags & TF_BODY)) continue;
        const V3 v0 = lds_row3(px, tf.l0), v1 = lds_row3(px, tf.l1), v2 = lds_row3(px, tf.l2);
        const V3 w0 = cross(v1, v2), w1 = cross(v2, v0), w2 = cross(v0, v1);
        if (tf.l0 < no) { atomicAdd(&stg[tf.l0], s6 * w0.x); atomicAdd(&stg[T + tf.l0], s6 * w0.y); atomicAdd(&stg[2 * T + tf.l0], s6 * w0.z); }
        if (tf.l1 < no) { atomicAdd(&stg[tf.l1], s6 * w1.x); atomicAdd(&stg[T + tf.l1], s6 * w1.y); atomicAdd(&stg[2 * T + tf.l1], s6 * w1.z); }
        if (tf.l2 < no) { atomicAdd(&stg[tf.l2], s6 * w2.x); atomicAdd(&stg[T + tf.l2], s6 * w2.y); atomicAdd(&stg[2 * T + tf.l2], s6 * w2.z); }
      }
      __syncthreads();
      if (tid < t.n_owned) {
        cx = stg[tid];
        cy = stg[T + tid];
        cz = stg[2 * T + tid];
      }
    }
  }
  double ggc = 0.0, gcgc = 0.0, gn2 = 0.0, gdd = 0.0, md2 = 0.0, mg2 = 0.0;
  if (tid < t.n_owned) {
    const size_t o = 3 * (size_t)(t.v_lo + tid);
    // bending.py:165-166: approx mode zeroes the boundary rows of what was accumulated
    if (BENDMODE == 2 && (lfl[tid] & VF_BOUNDARY)) gx = gy = gz = 0.0;
    if (a.dir_mode) {
      // fused direction pass (no constraint row to project against): same arithmetic as
      // k_direction -- fixed rows zeroed, GD or per-row Polak-Ribiere direction.
      const bool fixed = lfl[tid] & VF_FIXED;
      if (a.accumulate) {
        gx += a.g[o];
        gy += a.g[o + 1];
        gz += a.g[o + 2];
      }
      V3 gi = fixed ? mk(0, 0, 0) : mk(gx, gy, gz);
      V3 di = -gi;
      if (a.dir_mode == 2) {
        const V3 pgv = h_pg;
        const double beta = dot_pinned(gi, gi - pgv) / (dot_pinned(pgv, pgv) + 1.0e-20);
        if (!(beta < 0.0)) {
          const V3 q = h_pd;
          di = mk(fma(beta, q.x, -gi.x), fma(beta, q.y, -gi.y), fma(beta, q.z, -gi.z));
        }
      }
      if (fixed) di = mk(0, 0, 0);
      if (!MS_ABL_NOSTORE) {
        a.g[o] = gi.x;
        a.g[o + 1] = gi.y;
        a.g[o + 2] = gi.z;
        a.d[o] = di.x;
        a.d[o + 1] = di.y;
        a.d[o + 2] = di.z;
      }
      gn2 = dot_pinned(gi, gi);
      gdd = dot_pinned(gi, di);
      md2 = fixed ? 0.0 : dot_pinned(di, di);
      mg2 = gn2;  // (gi is zero on fixed rows) what max|d_i|^2 becomes if the stepper restarts with d = -g
    } else if (a.g) {
      if (a.accumulate) {
        gx += a.g[o];
        gy += a.g[o + 1];
        gz += a.g[o + 2];
      }
      a.g[o] = gx;
      a.g[o + 1] = gy;
      a.g[o + 2] = gz;
    }
    if (a.gC) {
      if (VOLROW) {
        a.gC[o] = cx;
        a.gC[o + 1] = cy;
        a.gC[o + 2] = cz;
      } else {
        cx = a.gC[o];
        cy = a.gC[o + 1];
        cz = a.gC[o + 2];
      }
      ggc = gx * cx + gy * cy + gz * cz;
      gcgc = cx * cx + cy * cy + cz * cz;
    }
  }
  if (a.dir_mode) {
    const double vals[4] = {gn2, gdd, md2, mg2};
    const int ops[4] = {0, 0, 2, 2};
    const int slots[4] = {MS_S_GNORM2, MS_S_GDOTD, MS_S_MAXD2, MS_S_MAXG2};
    block_reduce_store<4>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  } else {
    const double vals[2] = {ggc, gcgc};
    const int ops[2] = {0, 0};
    const int slots[2] = {MS_S_GGC, MS_S_GCGC};
    block_reduce_store<2>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  }
#if MS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row stores have left the wave
  MS_STAMP(3);
#endif
}

template <int BENDMODE, bool VOLROW, int TT, int CAPC, bool ATOMIC, bool LEAN = false>
__global__ __launch_bounds__(TT ? TT : 512, LEAN ? (ATOMIC ? ((VOLROW && !MS_ROW_TWO_PASS) ? 4 : MS_LEAN_SLOTS) : 3) : 1) MS_WPE_GRADIENT void k_gradient(GradientArgs a, int cap_rt, int max_ent) {
  extern __shared__ double lds[];
  gradient_body<BENDMODE, VOLROW, TT, CAPC, ATOMIC, LEAN>(a, cap_rt, max_ent, lds, (int)blockIdx.x);
}

size_t gradient_lds_bytes(int T, int cap, int max_ent, bool bend, bool volrow, bool atomic, bool leaf) {
  const size_t cols = atomic ? ((volrow && !MS_ROW_TWO_PASS) ? 6 : 3) : (volrow ? 18 : 9);
  size_t d = 3 * (size_t)cap + (bend ? 5 * (size_t)cap : 0) + (leaf ? 4 * (size_t)cap : 0) + cols * (size_t)T + 4 * 16;
  return d * sizeof(double) + (atomic ? 0 : u16_bytes(T, max_ent)) + (((size_t)cap + 15) / 16) * 16;
}

// the lean instances (see k_gradient): analytic bending on a closed surface, uniform gamma, no penalty, no pd rows to
// load -- without a constraint row (the headline's) or with one (no fused direction pass then: the KKT multiplier
// needs a global reduction first)
bool gradient_lean_instance(const GradientArgs& a) {
  const bool bend = (a.modules & MS_MOD_BENDING) != 0;
  const bool leaf = bend && a.bt_vert != nullptr;
  return a.m.T == FAST_T && a.m.tile_facets32 != nullptr && !leaf && bend && a.bending_grad_mode != MS_GRAD_APPROX &&
         !a.m.has_boundary && !(a.modules & MS_MOD_VOLUME_PENALTY) && !no_lean() && !a.m.no_fast;
}

hipError_t launch_gradient(const GradientArgs& a_in, int cap, int max_ent, hipStream_t s) {
  GradientArgs a = a_in;
  if (abl_ntiles() > 0) a.tile1 = std::min(a.tile1, a.tile0 + abl_ntiles());
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const bool bend = (a.modules & MS_MOD_BENDING) != 0;
  const bool volrow = a.gC != nullptr && (a.modules & MS_CON_VOLUME);
  const bool fast = a.m.T == FAST_T && a.m.tile_facets32 != nullptr && !a.m.no_fast;
  const bool atomic = a.atomic != 0;
  const bool leaf = bend && a.bt_vert != nullptr;
  const size_t lds = gradient_lds_bytes(a.m.T, cap, max_ent, bend, volrow, atomic, leaf);
  hipError_t e;
  if (ExecRecorder* r = exec_find(s)) {
    const int mode_r = !bend ? 0 : (leaf ? 3 : (a.bending_grad_mode == MS_GRAD_APPROX ? 2 : 1));
    if (fast || mode_r == 3)
      return r->push(CK_GRADIENT, 0, cap, max_ent, nb,
                     (gradient_lean_instance(a) ? 1u : 0u) | (volrow ? 2u : 0u) | (atomic ? 4u : 0u) | ((uint32_t)mode_r << 4),
                     lds, &a, sizeof(a), &a.m);
    e = r->flush();
    if (e != hipSuccess) return e;
  }
#define MS_LAUNCH_G(M, V, TT, CC, AT)                                                                \
  do {                                                                                               \
    e = ensure_lds(k_gradient<M, V, TT, CC, AT>, lds);                                               \
    if (e != hipSuccess) return e;                                                                   \
    hipLaunchKernelGGL((k_gradient<M, V, TT, CC, AT>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent); \
  } while (0)
  if (gradient_lean_instance(a)) {
#define MS_LAUNCH_LEAN(V, AT)                                                                               \
  do {                                                                                                      \
    e = ensure_lds(k_gradient<1, V, FAST_T, FAST_CAP, AT, true>, lds);                                      \
    if (e != hipSuccess) return e;                                                                          \
    hipLaunchKernelGGL((k_gradient<1, V, FAST_T, FAST_CAP, AT, true>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent); \
  } while (0)
    // (atomic: LDS ds_add_f64 vertex sums; otherwise the same diet with the fixed-order CSR gather, ms_set_deterministic)
    if (volrow) {
      if (atomic) MS_LAUNCH_LEAN(true, true); else MS_LAUNCH_LEAN(true, false);
    } else {
      if (atomic) MS_LAUNCH_LEAN(false, true); else MS_LAUNCH_LEAN(false, false);
    }
#undef MS_LAUNCH_LEAN
    return hipGetLastError();
  }
#define MS_PICK_G(M, V)                                           \
  do {                                                            \
    if (fast && atomic) MS_LAUNCH_G(M, V, FAST_T, FAST_CAP, true); \
    else if (fast) MS_LAUNCH_G(M, V, FAST_T, FAST_CAP, false);    \
    else if (atomic) MS_LAUNCH_G(M, V, 0, 0, true);               \
    else MS_LAUNCH_G(M, V, 0, 0, false);                          \
  } while (0)
  const int mode = !bend ? 0 : (leaf ? 3 : (a.bending_grad_mode == MS_GRAD_APPROX ? 2 : 1));
  if (mode == 3) {  // leaflet bending_tilt: generic-size instances only (not a headline path)
    if (volrow) {
      if (atomic) MS_LAUNCH_G(3, true, 0, 0, true); else MS_LAUNCH_G(3, true, 0, 0, false);
    } else {
      if (atomic) MS_LAUNCH_G(3, false, 0, 0, true); else MS_LAUNCH_G(3, false, 0, 0, false);
    }
  } else if (volrow) {
    if (mode == 0) MS_PICK_G(0, true); else if (mode == 1) MS_PICK_G(1, true); else MS_PICK_G(2, true);
  } else {
    if (mode == 0) MS_PICK_G(0, false); else if (mode == 1) MS_PICK_G(1, false); else MS_PICK_G(2, false);
  }
#undef MS_PICK_G
#undef MS_LAUNCH_G
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_tilt: vertex-tilt magnitude energy (modules/energy/tilt.py:99-172) as a tile
// kernel of the same shape as K_C.
//   coeff_f = 1/2 k_t (|t0|^2+|t1|^2+|t2|^2)/3 ;  E = sum_f coeff_f A_f  (|n| >= 1e-12 only)
//   MODE 1: shape gradient coeff_f dA/dv_k, dA/dv0 = 1/2 nhat x (v2-v1) (cyclic), ADDED
//           into g; tilt gradient k_t t_v A_v with barycentric A_v = sum A_f/3 (:160-170)
//   MODE 2: Mesh.project_tilts_to_tangent (geometry/mesh.py:788-814): unit vertex
//           normals = normalised sum of facet normals (|.| >= 1e-12, triangle_ops.py:55-73),
//           t <- t - (t.n) n on the owned rows.
// LDS: px[3][cap] | tq[cap] | stg[10][T] | red[16] | voff, vent (u16)
// ---------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void tilt_body(const TiltArgs& a, int cap, int max_ent, double* lds, int block_id) {
  const int T = a.m.T;
  double* px = lds;
  double* tq = px + 3 * cap;
  double* tl = tq + cap;  // consistent mass only: the tilt vectors themselves
  const bool cons = (MODE == 0 || MODE == 1) && a.consistent;
  double* stg = tl + (cons ? 3 * cap : 0);
  // (MODE 0 -- energy only -- gathers nothing: no staging block, no CSR; a third of the LDS, twice the workgroups per CU)
  double* red = stg + (MODE == 0 ? 0 : 10 * T);
  uint16_t* voff = reinterpret_cast<uint16_t*>(red + 16);
  uint16_t* vent = voff + (T + 2);

  const TileCtx t = tile_ctx(a.m, a.tile0 + xcd_tile(block_id, a.tile1 - a.tile0));
  const int tid = threadIdx.x;
  const bool have_d = a.d != nullptr;

  V3 tv = mk(0, 0, 0);  // this thread's owned tilt
  // facet records are fetched one chunk ahead; every load of a row set is issued before the first LDS write (the
  // stores may alias the inputs as far as the compiler knows: interleaved, the stage-in is one HBM round trip per
  // component)
  FacetRec<false> tf_nx = facet_null<false>();
  if (t.f0 + tid < t.f1) tf_nx = facet_load<false>(a.m, (size_t)(t.f0 + tid));
  {
    struct Row {
      double x0, x1, x2, t0, t1, t2;
    };
    auto load_row = [&](int v) {
      Row r;
      const size_t g = 3 * (size_t)v;
      const uint8_t fl = a.m.vflags[v];
      r.x0 = a.x[g];
      r.x1 = a.x[g + 1];
      r.x2 = a.x[g + 2];
      if (have_d) {
        const double d0 = a.d[g], d1 = a.d[g + 1], d2 = a.d[g + 2];
        if (!(fl & VF_FIXED)) {
          r.x0 = axpy1(r.x0, a.alpha, d0);
          r.x1 = axpy1(r.x1, a.alpha, d1);
          r.x2 = axpy1(r.x2, a.alpha, d2);
        }
      }
      r.t0 = a.tilts[g];
      r.t1 = a.tilts[g + 1];
      r.t2 = a.tilts[g + 2];
      return r;
    };
    auto put_row = [&](const Row& r, int sl) {
      px[sl] = r.x0;
      px[cap + sl] = r.x1;
      px[2 * cap + sl] = r.x2;
      const V3 th = mk(r.t0, r.t1, r.t2);
      tq[sl] = dot(th, th);
      if (cons) {
        tl[sl] = r.t0;
        tl[cap + sl] = r.t1;
        tl[2 * cap + sl] = r.t2;
      }
    };
    const bool own = tid < t.n_owned, has_h = tid < t.nh;
    int hv = 0;
    if (has_h) hv = a.m.halo_ids[t.h0 + tid];
    Row ro{}, rh{};
    if (own) ro = load_row(t.v_lo + tid);
    if (has_h) rh = load_row(hv);
    if (own) {
      put_row(ro, tid);
      tv = mk(ro.t0, ro.t1, ro.t2);
    }
    if (has_h) put_row(rh, t.n_owned + tid);
    for (int h = tid + T; h < t.nh; h += T) put_row(load_row(a.m.halo_ids[t.h0 + h]), t.n_owned + h);
  }
  if (MODE != 0) {  // (modes 1-3 gather per vertex)
    const uint16_t* gv = a.m.tile_voff + (size_t)t.tile * (T + 1);
    voff[tid] = gv[tid];
    if (tid == 0) voff[T] = gv[T];
    const uint16_t* ge = a.m.vent + t.e0;
    for (int j = tid; j < t.n_ent; j += T) vent[j] = ge[j];
  }
  __syncthreads();

  double e_tilt = 0.0;
  double ax = 0, ay = 0, az = 0, aw = 0;  // vertex accumulators: gradient / normal (xyz), area (w)
  int cur = 0, end = 0;
  if (MODE != 0 && tid < t.n_owned) {
    cur = voff[tid];
    end = voff[tid + 1];
  }
  double cf_area = 0.0, cf_theta = 0.0;  // MODE 5 vertex accumulators: mixed area, angle sum (K in ax, ay, az)
  // MODE 1, consistent mass with a.cons_tilt_grad: the module's own tilt gradient k A_f/12 (2 t_k + t_a + t_b)
  // (tilt_leaflet.py:124-150), gathered per vertex through a second pass over the staging block
  const bool cons_tg = MODE == 1 && cons && a.cons_tilt_grad;
  double tgx = 0.0, tgy = 0.0, tgz = 0.0;
  for (int c0 = t.f0; c0 < t.f1; c0 += T) {
    const int p = c0 + tid;
    double cf_a0 = 0, cf_a1 = 0, cf_a2 = 0, cf_t0 = 0, cf_t1 = 0, cf_t2 = 0;
    V3 ctg0 = mk(0, 0, 0), ctg1 = mk(0, 0, 0), ctg2 = mk(0, 0, 0);
    const TileFacet tf = facet_unpack<false>(tf_nx);
    if (p + T < t.f1) tf_nx = facet_load<false>(a.m, (size_t)(p + T));
    if (p < t.f1) {
      const V3 v0 = lds_v3(px, cap, tf.l0), v1 = lds_v3(px, cap, tf.l1), v2 = lds_v3(px, cap, tf.l2);
      const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
      const V3 n = cross(e2, -e1);
      const double A2 = norm(n);
      double* s = stg + tid;
      if (MODE == 5) {
        // compute_curvature_data (tilt_kernels.f90:133-181): cotans with area_doubled = max(|n|, 1e-12), the
        // integrated curvature vectors K_k and the mixed-Voronoi corner areas; angles as in MODE 4
        const double ad = A2 < 1.0e-12 ? 1.0e-12 : A2;
        const double inv_ad = 1.0 / ad;
        const double c0 = dot(-e1, e2) * inv_ad, c1 = dot(-e2, e0) * inv_ad, c2 = dot(-e0, e1) * inv_ad;
        const double hc0 = 0.5 * c0, hc1 = 0.5 * c1, hc2 = 0.5 * c2;
        const V3 K0 = hc2 * e2 - hc1 * e1, K1 = hc0 * e0 - hc2 * e2, K2 = hc1 * e1 - hc0 * e0;
        s[0 * T] = K0.x; s[1 * T] = K0.y; s[2 * T] = K0.z;
        s[3 * T] = K1.x; s[4 * T] = K1.y; s[5 * T] = K1.z;
        s[6 * T] = K2.x; s[7 * T] = K2.y; s[8 * T] = K2.z;
        const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
        corner_areas(c0, c1, c2, l0, l1, l2, 0.5 * ad, cf_a0, cf_a1, cf_a2);
        const double la = fmax(sqrt(l0), 1.0e-15), lb = fmax(sqrt(l1), 1.0e-15), lc = fmax(sqrt(l2), 1.0e-15);
        const double q0 = ((lb * lb + lc * lc) - la * la) / ((2.0 * lb) * lc);
        const double q1 = ((lc * lc + la * la) - lb * lb) / ((2.0 * lc) * la);
        const double q2 = ((la * la + lb * lb) - lc * lc) / ((2.0 * la) * lb);
        cf_t0 = acos(fmin(1.0, fmax(-1.0, q0)));
        cf_t1 = acos(fmin(1.0, fmax(-1.0, q1)));
        cf_t2 = acos(fmin(1.0, fmax(-1.0, q2)));
      } else if (MODE == 4) {
        // interior angles by the law of cosines, edge lengths clamped at 1e-15, cosines clipped to [-1, 1]
        // (geometry/curvature.py:366-386)
        const double la = fmax(norm(e0), 1.0e-15), lb = fmax(norm(e1), 1.0e-15), lc = fmax(norm(e2), 1.0e-15);
        const double c0 = ((lb * lb + lc * lc) - la * la) / ((2.0 * lb) * lc);
        const double c1 = ((lc * lc + la * la) - lb * lb) / ((2.0 * lc) * la);
        const double c2 = ((la * la + lb * lb) - lc * lc) / ((2.0 * la) * lb);
        s[0 * T] = acos(fmin(1.0, fmax(-1.0, c0)));
        s[1 * T] = acos(fmin(1.0, fmax(-1.0, c1)));
        s[2 * T] = acos(fmin(1.0, fmax(-1.0, c2)));
      } else if (MODE == 2 || MODE == 3) {
        s[0 * T] = n.x;
        s[1 * T] = n.y;
        s[2 * T] = n.z;
        if (MODE == 3) s[9 * T] = (0.5 * A2) / 3.0;
      } else {
        V3 G0 = mk(0, 0, 0), G1 = mk(0, 0, 0), G2 = mk(0, 0, 0);
        double a3 = 0.0;
        if (A2 >= 1.0e-12) {
          const double area = 0.5 * A2;
          const double sq = (tq[tf.l0] + tq[tf.l1]) + tq[tf.l2];
          double coeff = 0.5 * a.k_tilt * (sq / 3.0);
          if (cons) {  // tilt_leaflet.py:91-114: (k/12)(sum |t|^2 + t0.t1 + t1.t2 + t2.t0)
            const V3 t0 = lds_v3(tl, cap, tf.l0), t1 = lds_v3(tl, cap, tf.l1), t2 = lds_v3(tl, cap, tf.l2);
            coeff = (a.k_tilt / 12.0) * (((sq + dot(t0, t1)) + dot(t1, t2)) + dot(t2, t0));
            if (cons_tg) {  // :124-150: tri_factor ((2 t_k) + t_a + t_b), tri_factor = k A_f / 12
              const double tri = (a.k_tilt * area) / 12.0;
              ctg0 = tri * (((2.0 * t0) + t1) + t2);
              ctg1 = tri * (((2.0 * t1) + t2) + t0);
              ctg2 = tri * (((2.0 * t2) + t0) + t1);
            }
          }
          if (tf.flags & TF_OWNER) e_tilt += coeff * area;
          if (MODE == 1) {
            const double hs = 0.5 * coeff / A2;
            G0 = hs * cross(n, e0);
            G1 = hs * cross(n, e1);
            G2 = hs * cross(n, e2);
            a3 = area / 3.0;
          }
        }
        if (MODE == 1) {
          s[0 * T] = G0.x; s[1 * T] = G0.y; s[2 * T] = G0.z;
          s[3 * T] = G1.x; s[4 * T] = G1.y; s[5 * T] = G1.z;
          s[6 * T] = G2.x; s[7 * T] = G2.y; s[8 * T] = G2.z;
          s[9 * T] = a3;
        }
      }
    }
    if (MODE != 0) {
      __syncthreads();
      const int lo = c0 - t.f0, hi = min(c0 + T, t.f1) - t.f0;
      const int cur0 = cur;
      (void)cur0;
      while (cur < end) {
        const int ent = vent[cur];
        const int fl = ent >> 2;
        if (fl >= hi) break;
        const double* s = stg + (fl - lo);
        if (MODE == 5) {
          const int k = ent & 3;
          ax += s[(3 * k) * T];
          ay += s[(3 * k + 1) * T];
          az += s[(3 * k + 2) * T];
        } else if (MODE == 4) {
          aw += s[(ent & 3) * T];  // this corner's angle
        } else if (MODE == 2 || MODE == 3) {
          ax += s[0];
          ay += s[T];
          az += s[2 * T];
          if (MODE == 3) aw += s[9 * T];
        } else {
          const int k = ent & 3;
          ax += s[(3 * k) * T];
          ay += s[(3 * k + 1) * T];
          az += s[(3 * k + 2) * T];
          aw += s[9 * T];
        }
        ++cur;
      }
      __syncthreads();
      if (cons_tg) {  // second sub-phase: the per-corner consistent tilt gradients through the same staging block
        if (p < t.f1) {
          double* s2 = stg + tid;
          s2[0 * T] = ctg0.x; s2[1 * T] = ctg0.y; s2[2 * T] = ctg0.z;
          s2[3 * T] = ctg1.x; s2[4 * T] = ctg1.y; s2[5 * T] = ctg1.z;
          s2[6 * T] = ctg2.x; s2[7 * T] = ctg2.y; s2[8 * T] = ctg2.z;
        }
        __syncthreads();
        for (int q = cur0; q < cur; ++q) {
          const int ent = vent[q];
          const double* s2 = stg + ((ent >> 2) - lo) + 3 * (ent & 3) * T;
          tgx += s2[0];
          tgy += s2[T];
          tgz += s2[2 * T];
        }
        __syncthreads();
      }
      if (MODE == 5) {  // second sub-phase: corner areas and angles through the same staging block
        if (p < t.f1) {
          double* s2 = stg + tid;
          s2[0 * T] = cf_a0; s2[1 * T] = cf_a1; s2[2 * T] = cf_a2;
          s2[3 * T] = cf_t0; s2[4 * T] = cf_t1; s2[5 * T] = cf_t2;
        }
        __syncthreads();
        for (int q = cur0; q < cur; ++q) {
          const int ent = vent[q];
          const double* s2 = stg + ((ent >> 2) - lo) + (ent & 3) * T;
          cf_area += s2[0];
          cf_theta += s2[3 * T];
        }
        __syncthreads();
      }
    }
  }
  if (tid < t.n_owned) {
    const size_t o = 3 * (size_t)(t.v_lo + tid);
    if (MODE == 5) {
      // compute_curvature_fields (geometry/curvature.py:404-448)
      const size_t plane = 3 * (size_t)a.fields_rows;
      const double safe = fmax(cf_area, 1.0e-12);
      const V3 hn = mk(ax / (2.0 * safe), ay / (2.0 * safe), az / (2.0 * safe));
      const double H = norm(hn);
      const double two_pi = 2.0 * 3.14159265358979323846;
      const double defect = (a.m.vflags[t.v_lo + tid] & VF_BOUNDARY) ? 0.0 : two_pi - cf_theta;
      const double KG = defect / safe;
      const double root = sqrt(fmax(H * H - KG, 0.0));
      double* f = a.fields;
      f[o] = hn.x; f[o + 1] = hn.y; f[o + 2] = hn.z;                                  // mean-curvature normal
      f[plane + o] = H; f[plane + o + 1] = cf_area; f[plane + o + 2] = cf_theta;      // H, mixed area, angle sum
      f[2 * plane + o] = defect; f[2 * plane + o + 1] = KG; f[2 * plane + o + 2] = 0.0;  // defect, K_G
      f[3 * plane + o] = H + root; f[3 * plane + o + 1] = H - root; f[3 * plane + o + 2] = 0.0;  // k1, k2
    } else if (MODE == 1) {
      if (a.g) {
        a.g[o] += ax;
        a.g[o + 1] += ay;
        a.g[o + 2] += az;
      }
      // lumped: k t_v A_v with the barycentric A_v; consistent (module form): the gathered per-corner vectors
      const V3 tg = cons_tg ? mk(tgx, tgy, tgz) : mk(a.k_tilt * tv.x * aw, a.k_tilt * tv.y * aw, a.k_tilt * tv.z * aw);
      if (a.tg_accumulate) {
        a.tilt_grad[o] += tg.x;
        a.tilt_grad[o + 1] += tg.y;
        a.tilt_grad[o + 2] += tg.z;
      } else {
        a.tilt_grad[o] = tg.x;
        a.tilt_grad[o + 1] = tg.y;
        a.tilt_grad[o + 2] = tg.z;
      }
    } else if (MODE == 2) {
      V3 nrm = mk(ax, ay, az);
      const double len = norm(nrm);
      if (len >= 1.0e-12) nrm = mk(nrm.x / len, nrm.y / len, nrm.z / len);
      const double dt = dot(tv, nrm);
      a.tilts_out[o] = tv.x - dt * nrm.x;
      a.tilts_out[o + 1] = tv.y - dt * nrm.y;
      a.tilts_out[o + 2] = tv.z - dt * nrm.z;
    } else if (MODE == 4) {
      // angle defect 2 pi - sum of incident angles, 0 on boundary rows (geometry/curvature.py:393-401)
      const double two_pi = 2.0 * 3.14159265358979323846;
      a.minv[t.v_lo + tid] = (a.m.vflags[t.v_lo + tid] & VF_BOUNDARY) ? 0.0 : two_pi - aw;
    } else if (MODE == 3) {
      // relaxation geometry: unit vertex normals (triangle_ops.py:55-73) and the tilt-rigidity
      // part k_t A_v of the Jacobi diagonal (runtime/preconditioners.py:27-40)
      V3 nrm = mk(ax, ay, az);
      const double len = norm(nrm);
      if (len >= 1.0e-12) nrm = mk(nrm.x / len, nrm.y / len, nrm.z / len);
      a.tilts_out[o] = nrm.x;
      a.tilts_out[o + 1] = nrm.y;
      a.tilts_out[o + 2] = nrm.z;
      if (a.minv) {  // raw diagonal (k_tvec mode 3 clamps and inverts), or finished here
        double dg = a.k_tilt * aw;
        if (a.finish_minv) {
          if (!(dg > 1.0e-12) || (a.m.vflags[t.v_lo + tid] & a.fixed_bit)) dg = 1.0;
          dg = 1.0 / dg;
        }
        a.minv[t.v_lo + tid] = dg;
      }
      if (a.va_out) a.va_out[t.v_lo + tid] = aw;         // barycentric vertex area (mesh.py:671-730)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (a.fld_in[k] != nullptr) {  // (MODE 2's projection, expression for expression)
          const V3 tk = mk(a.fld_in[k][o], a.fld_in[k][o + 1], a.fld_in[k][o + 2]);
          const double dk = dot(tk, nrm);
          a.fld_out[k][o] = tk.x - dk * nrm.x;
          a.fld_out[k][o + 1] = tk.y - dk * nrm.y;
          a.fld_out[k][o + 2] = tk.z - dk * nrm.z;
        }
      if (a.proj_out) {
        const V3 r = tilt_trial_row(tv, tv, nrm, 0.0);
        a.proj_out[o] = r.x;
        a.proj_out[o + 1] = r.y;
        a.proj_out[o + 2] = r.z;
      }
      // the second field of the relaxation: the same three outputs
      if (a.minv_b) {
        double dg = a.k_tilt_b * aw;
        if (a.finish_minv_b) {
          if (!(dg > 1.0e-12) || (a.m.vflags[t.v_lo + tid] & a.fixed_bit_b)) dg = 1.0;
          dg = 1.0 / dg;
        }
        a.minv_b[t.v_lo + tid] = dg;
      }
      if (a.va_out_b) a.va_out_b[t.v_lo + tid] = aw;
      if (a.proj_out_b) {
        const V3 tb = mk(a.tilts_b[o], a.tilts_b[o + 1], a.tilts_b[o + 2]);
        const V3 r = tilt_trial_row(tb, tb, nrm, 0.0);
        a.proj_out_b[o] = r.x;
        a.proj_out_b[o + 1] = r.y;
        a.proj_out_b[o + 2] = r.z;
      }
    }
  }
  if (MODE != 2 && MODE != 3 && MODE != 4 && MODE != 5) {
    const double vals[1] = {e_tilt};
    const int ops[1] = {0};
    const int slots[1] = {a.e_slot};
    block_reduce_store<1>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_tilt(TiltArgs a, int cap, int max_ent) {
  extern __shared__ double lds[];
  tilt_body<MODE>(a, cap, max_ent, lds, (int)blockIdx.x);
}

size_t tilt_lds_bytes(int T, int cap, int max_ent, bool consistent, int mode) {
  if (mode == 0) return ((consistent ? 7 : 4) * (size_t)cap + 16) * sizeof(double);
  return ((consistent ? 7 : 4) * (size_t)cap + 10 * (size_t)T + 16) * sizeof(double) +
         2 * ((size_t)T + 2 + max_ent + 8);
}

hipError_t launch_tilt(const TiltArgs& a, int mode, int cap, int max_ent, hipStream_t s) {
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const size_t lds = tilt_lds_bytes(a.m.T, cap, max_ent, (mode == 0 || mode == 1) && a.consistent, mode);
  hipError_t e;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_TILT, mode, cap, max_ent, nb, 0, lds, &a, sizeof(a), &a.m);
#define MS_LAUNCH_T(M)                                                                  \
  do {                                                                                  \
    e = ensure_lds(k_tilt<M>, lds);                                                     \
    if (e != hipSuccess) return e;                                                      \
    hipLaunchKernelGGL((k_tilt<M>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent);    \
  } while (0)
  if (mode == 0) MS_LAUNCH_T(0); else if (mode == 1) MS_LAUNCH_T(1); else if (mode == 2) MS_LAUNCH_T(2);
  else if (mode == 3) MS_LAUNCH_T(3); else if (mode == 4) MS_LAUNCH_T(4); else MS_LAUNCH_T(5);
#undef MS_LAUNCH_T
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_bt: Helfrich bending with Kozlov-Hamm tilt-splay coupling
// (modules/energy/bending_tilt.py:151-482), the facet pass that follows an energy pass run
// with EnergyArgs.bt_vert:  E = 1/2 sum_f sum_k kappa_k (base_k + div_f t)^2 va_eff[f,k].
//   div_f t = t0.g0 + t1.g1 + t2.g2, g_k = n x e_k / max(|n|^2, 1e-20)
//             (geometry/tilt_operators.py:191-330, fortran_kernels/tilt_kernels.f90:26-86)
//   va_eff  = mixed-Voronoi corner areas with the boundary->interior redistribution
//             (bending_utils.py:37-171), recomputed exactly as the energy pass does
//   MODE 1: div_eff_v = sum va_eff div / A_eff (:243-253), term = base + div_eff (0 on the
//           boundary), then fK = K_dir kappa ratio term, fA_eff = kappa term^2/2,
//           fA_vor = -2 kappa term ratio H (:279-283) for the unchanged gradient pass
//   MODE 2: tilt gradient dE/dt_k = s (sum_j kappa_j term_j va_eff_j) g_k ADDED to tilt_grad
//   MODE 3: (leaflet form) shape gradient of the divergence, s (sum_j ...) d(div_f t)/dx ADDED to g
//           (modules/energy/bt_gradient.py:20-64, bending_tilt_leaflet.py:692-699)
//   div_sign s: -1 for bending_tilt_in (bending_tilt_in.py:46), +1 otherwise.
// LDS: px[3][cap] | tl[3][cap] | bs[cap] | kp[cap] | stg[9][T] (red aliases it) | vent | fl
// ---------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void bt_body(const BtArgs& a, int cap, int max_ent, double* lds, int block_id) {
  const int T = a.m.T;
  double* px = lds;
  double* tl = px + 3 * cap;
  double* bs = tl + 3 * cap;
  double* kp = bs + cap;
  double* stg = kp + cap;
  double* red = stg;
  // (MODE 0 -- energy only -- gathers nothing: the staging block shrinks to the reduction scratch, no CSR)
  uint16_t* vent = reinterpret_cast<uint16_t*>(stg + (MODE == 0 ? 32 : 9 * T));
  uint8_t* lfl = reinterpret_cast<uint8_t*>(vent + (MODE == 0 ? 0 : ((max_ent + 3) & ~3)));

  const TileCtx t = tile_ctx(a.m, a.tile0 + xcd_tile(block_id, a.tile1 - a.tile0));
  const int tid = threadIdx.x;
  const bool have_d = a.d != nullptr;

  int cur = 0, end = 0;
  uint8_t own_fl = 0;
  // facet records are fetched one chunk ahead (their HBM latency overlaps the staging / the previous chunk's arithmetic)
  FacetRec<false> tf_nx = facet_null<false>();
  if (t.f0 + tid < t.f1) tf_nx = facet_load<false>(a.m, (size_t)(t.f0 + tid));
  {
    const bool own = tid < t.n_owned;
    const int v_own = t.v_lo + tid;
    CsrStage cs;
    if (MODE != 0) csr_issue(cs, a.m, t, T, tid);
    // all loads of a row set are issued before the first LDS write (two HBM round trips)
    struct Row {
      double x0, x1, x2, t0, t1, t2, b, k;
      uint8_t fl;
    };
    auto load_row = [&](int v) {
      Row r;
      const size_t g = 3 * (size_t)v;
      r.fl = a.m.vflags[v];
      r.x0 = a.x[g];
      r.x1 = a.x[g + 1];
      r.x2 = a.x[g + 2];
      if (have_d) {
        const double d0 = a.d[g], d1 = a.d[g + 1], d2 = a.d[g + 2];
        if (!(r.fl & VF_FIXED)) {
          r.x0 = axpy1(r.x0, a.alpha, d0);
          r.x1 = axpy1(r.x1, a.alpha, d1);
          r.x2 = axpy1(r.x2, a.alpha, d2);
        }
      }
      r.t0 = a.tilts[g];
      r.t1 = a.tilts[g + 1];
      r.t2 = a.tilts[g + 2];
      r.b = a.bt_vert[4 * (size_t)v];
      r.k = a.m.kappa[v];
      return r;
    };
    auto put_row = [&](const Row& r, int sl) {
      px[sl] = r.x0;
      px[cap + sl] = r.x1;
      px[2 * cap + sl] = r.x2;
      tl[sl] = r.t0;
      tl[cap + sl] = r.t1;
      tl[2 * cap + sl] = r.t2;
      bs[sl] = r.b;
      kp[sl] = r.k;
      lfl[sl] = r.fl;
    };
    const bool has_h = tid < t.nh;
    int hv = 0;
    if (has_h) hv = a.m.halo_ids[t.h0 + tid];
    Row ro{}, rh{};
    if (own) ro = load_row(v_own);
    if (has_h) rh = load_row(hv);
    if (own) {
      put_row(ro, tid);
      own_fl = ro.fl;
    }
    if (has_h) put_row(rh, t.n_owned + tid);
    for (int h = tid + T; h < t.nh; h += T) put_row(load_row(a.m.halo_ids[t.h0 + h]), t.n_owned + h);
    if (MODE != 0) {
      csr_commit(cs, a.m, t, T, tid, vent);
      if (own) {
        cur = cs.vo0;
        end = cs.vo1;
      }
    }
  }
  __syncthreads();

  double e_bt = 0.0, e_tl = 0.0;
  double ax = 0, ay = 0, az = 0;  // MODE 1: ax = sum va_eff div ; MODE 2: tilt gradient
  for (int c0f = t.f0; c0f < t.f1; c0f += T) {
    const int p = c0f + tid;
    const TileFacet tf = facet_unpack<false>(tf_nx);
    if (p + T < t.f1) tf_nx = facet_load<false>(a.m, (size_t)(p + T));
    if (p < t.f1) {
      const V3 v0 = lds_v3(px, cap, tf.l0), v1 = lds_v3(px, cap, tf.l1), v2 = lds_v3(px, cap, tf.l2);
      const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
      const double l0 = dot(e0, e0), l1 = dot(e1, e1), l2 = dot(e2, e2);
      const V3 n = cross(e2, -e1);
      const double n2 = dot(n, n);
      const double A2 = sqrt(n2);
      // P1 basis gradients and the facet divergence (one reciprocal of |n|^2 for the nine components)
      const double denom = n2 > 1.0e-20 ? n2 : 1.0e-20;
      const double i2 = 1.0 / denom;
      const V3 g0 = i2 * cross(n, e0), g1 = i2 * cross(n, e1), g2 = i2 * cross(n, e2);
      const V3 tt0 = lds_v3(tl, cap, tf.l0), tt1 = lds_v3(tl, cap, tf.l1), tt2 = lds_v3(tl, cap, tf.l2);
      // div_term = s * div_f t (bending_tilt_in.py:46: s = -1; out and the single field: +1)
      const double dv = a.div_sign * (dot(tt0, g0) + dot(tt1, g1) + dot(tt2, g2));
      // cotans and effective corner areas, as in the energy pass
      const double ad = A2 < 1.0e-12 ? 1.0e-12 : A2;
      const double inv_ad = 1.0 / ad;
      const double cc0 = dot(-e1, e2) * inv_ad, cc1 = dot(-e2, e0) * inv_ad, cc2 = dot(-e0, e1) * inv_ad;
      double ve0, ve1, ve2;
      corner_areas(cc0, cc1, cc2, l0, l1, l2, fmax(0.5 * A2, 1.0e-12), ve0, ve1, ve2);
      int b0 = 0, b1 = 0, b2 = 0;
      if (a.m.has_boundary) {
        b0 = (lfl[tf.l0] & VF_BOUNDARY) ? 1 : 0;
        b1 = (lfl[tf.l1] & VF_BOUNDARY) ? 1 : 0;
        b2 = (lfl[tf.l2] & VF_BOUNDARY) ? 1 : 0;
      }
      const int n_int = 3 - (b0 + b1 + b2);
      if (n_int > 0 && n_int < 3) {
        const double b_sum = ve0 * b0 + ve1 * b1 + ve2 * b2;
        const double extra = b_sum / (double)n_int;
        const double m0 = b0 ? 0.0 : 1.0, m1 = b1 ? 0.0 : 1.0, m2 = b2 ? 0.0 : 1.0;
        ve0 = ve0 * m0 + m0 * extra;
        ve1 = ve1 * m1 + m1 * extra;
        ve2 = ve2 * m2 + m2 * extra;
      }
      const double t0 = bs[tf.l0] + dv, t1 = bs[tf.l1] + dv, t2 = bs[tf.l2] + dv;
      const double k0 = kp[tf.l0], k1 = kp[tf.l1], k2 = kp[tf.l2];
      if (tf.flags & TF_OWNER) e_bt += 0.5 * ((k0 * (t0 * t0) * ve0 + k1 * (t1 * t1) * ve1) + k2 * (t2 * t2) * ve2);
      if (MODE == 0 && a.k_tilt_fused != 0.0 && A2 >= 1.0e-12 && (tf.flags & TF_OWNER)) {
        // k_tilt's MODE 0 term, operation for operation: 1/2 k_t (sum |t_k|^2 / 3) * area
        const double sq = (dot(tt0, tt0) + dot(tt1, tt1)) + dot(tt2, tt2);
        const double coeff = 0.5 * a.k_tilt_fused * (sq / 3.0);
        e_tl += coeff * (0.5 * A2);
      }
      double* s = stg + tid;
      if (MODE == 1) {
        s[0 * T] = ve0 * dv;
        s[1 * T] = ve1 * dv;
        s[2 * T] = ve2 * dv;
      } else if (MODE == 2) {
        const double dE = a.div_sign * ((k0 * t0 * ve0 + k1 * t1 * ve1) + k2 * t2 * ve2);
        s[0 * T] = dE * g0.x; s[1 * T] = dE * g0.y; s[2 * T] = dE * g0.z;
        s[3 * T] = dE * g1.x; s[4 * T] = dE * g1.y; s[5 * T] = dE * g1.z;
        s[6 * T] = dE * g2.x; s[7 * T] = dE * g2.y; s[8 * T] = dE * g2.z;
      } else if (MODE == 3) {
        // dE/ddiv * d(div_P1 t)/dx for ambient tilts (bt_gradient.py:20-64): a = e2, b = -e1,
        // w = sum e_k x t_k, div = n.w/|n|^2
        const double dE = a.div_sign * ((k0 * t0 * ve0 + k1 * t1 * ve1) + k2 * t2 * ve2);
        const V3 w = (cross(e0, tt0) + cross(e1, tt1)) + cross(e2, tt2);
        const double ndw = dot(n, w);
        const V3 ddn = i2 * w - ((2.0 * ndw) * (i2 * i2)) * n;
        const V3 d0 = i2 * cross(tt0, n), d1 = i2 * cross(tt1, n), d2 = i2 * cross(tt2, n);
        const V3 ga = dE * ((cross(-e1, ddn) - d0) + d2);
        const V3 gb = dE * ((cross(ddn, e2) + d0) - d1);
        const V3 gc = -(ga + gb);
        s[0 * T] = gc.x; s[1 * T] = gc.y; s[2 * T] = gc.z;
        s[3 * T] = ga.x; s[4 * T] = ga.y; s[5 * T] = ga.z;
        s[6 * T] = gb.x; s[7 * T] = gb.y; s[8 * T] = gb.z;
      }
    }
    if (MODE != 0) {
      __syncthreads();
      const int lo = c0f - t.f0, hi = min(c0f + T, t.f1) - t.f0;
      while (cur < end) {
        const int ent = vent[cur];
        const int fl = ent >> 2;
        if (fl >= hi) break;
        const int k = ent & 3;
        const double* s = stg + (fl - lo);
        if (MODE == 1) {
          ax += s[k * T];
        } else {
          ax += s[(3 * k) * T];
          ay += s[(3 * k + 1) * T];
          az += s[(3 * k + 2) * T];
        }
        ++cur;
      }
      __syncthreads();
    }
  }
  if (MODE != 0 && tid < t.n_owned) {
    const int v = t.v_lo + tid;
    const size_t o = 3 * (size_t)v;
    if (MODE == 1) {
      const double* rec = a.bt_vert + 4 * (size_t)v;
      const double base = rec[0], A_eff = rec[1], krH = rec[2];
      const double div_eff = A_eff > 1.0e-20 ? ax / A_eff : 0.0;
      double term = base + div_eff;
      if (own_fl & VF_BOUNDARY) term = 0.0;
      const double kappa = kp[tid];
      a.fK[o] = a.fK[o] * term;
      a.fK[o + 1] = a.fK[o + 1] * term;
      a.fK[o + 2] = a.fK[o + 2] * term;
      a.fA[2 * (size_t)v] = 0.5 * kappa * (term * term);
      a.fA[2 * (size_t)v + 1] = -2.0 * term * krH;
    } else {
      double* dst = MODE == 3 ? a.g : a.tilt_grad;  // MODE 3: shape gradient, MODE 2: tilt gradient
      dst[o] += ax;
      dst[o + 1] += ay;
      dst[o + 2] += az;
    }
  }
  if (MODE != 0) __syncthreads();  // red aliases the staging block
  if (MODE == 0 && a.k_tilt_fused != 0.0) {
    const double vals[2] = {e_bt, e_tl};
    const int ops[2] = {0, 0};
    const int slots[2] = {a.e_slot, a.e_tilt_slot};
    block_reduce_store<2>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  } else {
    const double vals[1] = {e_bt};
    const int ops[1] = {0};
    const int slots[1] = {a.e_slot};
    block_reduce_store<1>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_bt(BtArgs a, int cap, int max_ent) {
  extern __shared__ double lds[];
  bt_body<MODE>(a, cap, max_ent, lds, (int)blockIdx.x);
}

size_t bt_lds_bytes(int T, int cap, int max_ent, int mode) {
  if (mode == 0) return (8 * (size_t)cap + 32) * sizeof(double) + (((size_t)cap + 15) / 16) * 16;
  return (8 * (size_t)cap + 9 * (size_t)T) * sizeof(double) + 2 * ((size_t)((max_ent + 3) & ~3)) +
         (((size_t)cap + 15) / 16) * 16;
}

hipError_t launch_bt(const BtArgs& a, int mode, int cap, int max_ent, hipStream_t s) {
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const size_t lds = bt_lds_bytes(a.m.T, cap, max_ent, mode);
  hipError_t e;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_BT, mode, cap, max_ent, nb, 0, lds, &a, sizeof(a), &a.m);
#define MS_LAUNCH_B(M)                                                                      \
  do {                                                                                      \
    e = ensure_lds(k_bt<M>, lds);                                                           \
    if (e != hipSuccess) return e;                                                          \
    hipLaunchKernelGGL((k_bt<M>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent);          \
  } while (0)
  if (mode == 0) MS_LAUNCH_B(0); else if (mode == 1) MS_LAUNCH_B(1); else if (mode == 2) MS_LAUNCH_B(2); else MS_LAUNCH_B(3);
#undef MS_LAUNCH_B
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_tsmooth: cotangent Dirichlet energy of the tilt field (modules/energy/tilt_smoothness.py:84-198,
// ambient_v1 transport):  E = k_s/4 sum_f [c0|t1-t2|^2 + c1|t2-t0|^2 + c2|t0-t1|^2] with the
// cotans of compute_curvature_data (tilt_kernels.f90:140-151).  No shape gradient (:21-23).
//   MODE 1: tilt gradient k_s/2 [c1 (t0-t2) + c2 (t0-t1)] (cyclic), ADDED to tilt_grad
//   MODE 2: Jacobi diagonal k_s/2 (c_a + c_b) per corner (runtime/preconditioners.py:42-57), ADDED
// LDS: px[3][cap] | tl[3][cap] | stg[9][T] (red aliases it) | vent
// ---------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void ts_body(const TsArgs& a, int cap, int max_ent, double* lds, int block_id) {
  const int T = a.m.T;
  double* px = lds;
  double* tl = px + 3 * cap;
  double* stg = tl + 3 * cap;
  double* red = stg;
  uint16_t* vent = reinterpret_cast<uint16_t*>(stg + 9 * T);

  const TileCtx t = tile_ctx(a.m, a.tile0 + xcd_tile(block_id, a.tile1 - a.tile0));
  const int tid = threadIdx.x;
  const bool have_d = a.d != nullptr;
  int cur = 0, end = 0;
  {
    const bool own = tid < t.n_owned;
    CsrStage cs;
    if (MODE != 0) csr_issue(cs, a.m, t, T, tid);
    struct Row {
      double x0, x1, x2, t0, t1, t2;
    };
    auto load_row = [&](int v) {
      Row r;
      const size_t g = 3 * (size_t)v;
      r.x0 = a.x[g];
      r.x1 = a.x[g + 1];
      r.x2 = a.x[g + 2];
      if (have_d) {
        const double d0 = a.d[g], d1 = a.d[g + 1], d2 = a.d[g + 2];
        if (!(a.m.vflags[v] & VF_FIXED)) {
          r.x0 = axpy1(r.x0, a.alpha, d0);
          r.x1 = axpy1(r.x1, a.alpha, d1);
          r.x2 = axpy1(r.x2, a.alpha, d2);
        }
      }
      r.t0 = a.tilts[g];
      r.t1 = a.tilts[g + 1];
      r.t2 = a.tilts[g + 2];
      return r;
    };
    auto put_row = [&](const Row& r, int sl) {
      px[sl] = r.x0;
      px[cap + sl] = r.x1;
      px[2 * cap + sl] = r.x2;
      tl[sl] = r.t0;
      tl[cap + sl] = r.t1;
      tl[2 * cap + sl] = r.t2;
    };
    const bool has_h = tid < t.nh;
    int hv = 0;
    if (has_h) hv = a.m.halo_ids[t.h0 + tid];
    Row ro{}, rh{};
    if (own) ro = load_row(t.v_lo + tid);
    if (has_h) rh = load_row(hv);
    if (own) put_row(ro, tid);
    if (has_h) put_row(rh, t.n_owned + tid);
    for (int h = tid + T; h < t.nh; h += T) put_row(load_row(a.m.halo_ids[t.h0 + h]), t.n_owned + h);
    if (MODE != 0) {
      csr_commit(cs, a.m, t, T, tid, vent);
      if (own) {
        cur = cs.vo0;
        end = cs.vo1;
      }
    }
  }
  __syncthreads();

  double e_ts = 0.0;
  double ax = 0, ay = 0, az = 0;
  const double hk = 0.5 * a.k_smooth;
  for (int c0f = t.f0; c0f < t.f1; c0f += T) {
    const int p = c0f + tid;
    if (p < t.f1) {
      const TileFacet tf = a.m.tile_facets[p];
      const V3 v0 = lds_v3(px, cap, tf.l0), v1 = lds_v3(px, cap, tf.l1), v2 = lds_v3(px, cap, tf.l2);
      const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0;
      const V3 n = cross(e2, -e1);
      const double A2 = norm(n);
      const double ad = A2 < 1.0e-12 ? 1.0e-12 : A2;
      const double inv_ad = 1.0 / ad;
      const double c0 = dot(-e1, e2) * inv_ad, c1 = dot(-e2, e0) * inv_ad, c2 = dot(-e0, e1) * inv_ad;
      double* s = stg + tid;
      if (MODE == 2) {
        s[0 * T] = hk * (c1 + c2);
        s[1 * T] = hk * (c2 + c0);
        s[2 * T] = hk * (c0 + c1);
      } else {
        const V3 t0 = lds_v3(tl, cap, tf.l0), t1 = lds_v3(tl, cap, tf.l1), t2 = lds_v3(tl, cap, tf.l2);
        const V3 d12 = t1 - t2, d20 = t2 - t0, d01 = t0 - t1;
        if (tf.flags & TF_OWNER)
          e_ts += 0.25 * a.k_smooth * ((c0 * dot(d12, d12) + c1 * dot(d20, d20)) + c2 * dot(d01, d01));
        if (MODE == 1) {
          // g0 = k/2 (c1 (t0-t2) + c2 (t0-t1)), g1 = k/2 (c2 (t1-t0) + c0 (t1-t2)), g2 = k/2 (c0 (t2-t1) + c1 (t2-t0))
          const V3 g0 = hk * (c2 * d01 - c1 * d20);
          const V3 g1 = hk * (c0 * d12 - c2 * d01);
          const V3 g2 = hk * (c1 * d20 - c0 * d12);
          s[0 * T] = g0.x; s[1 * T] = g0.y; s[2 * T] = g0.z;
          s[3 * T] = g1.x; s[4 * T] = g1.y; s[5 * T] = g1.z;
          s[6 * T] = g2.x; s[7 * T] = g2.y; s[8 * T] = g2.z;
        }
      }
    }
    if (MODE != 0) {
      __syncthreads();
      const int lo = c0f - t.f0, hi = min(c0f + T, t.f1) - t.f0;
      while (cur < end) {
        const int ent = vent[cur];
        const int fl = ent >> 2;
        if (fl >= hi) break;
        const int k = ent & 3;
        const double* s = stg + (fl - lo);
        if (MODE == 2) {
          ax += s[k * T];
        } else {
          ax += s[(3 * k) * T];
          ay += s[(3 * k + 1) * T];
          az += s[(3 * k + 2) * T];
        }
        ++cur;
      }
      __syncthreads();
    }
  }
  if (MODE != 0 && tid < t.n_owned) {
    const int v = t.v_lo + tid;
    if (MODE == 1) {
      const size_t o = 3 * (size_t)v;
      a.tilt_grad[o] += ax;
      a.tilt_grad[o + 1] += ay;
      a.tilt_grad[o + 2] += az;
    } else {
      a.diag[v] += ax;
    }
  }
  if (MODE != 2) {
    if (MODE != 0) __syncthreads();  // red aliases the staging block
    const double vals[1] = {e_ts};
    const int ops[1] = {0};
    const int slots[1] = {a.e_slot};
    block_reduce_store<1>(vals, ops, slots, red, a.partials + t.tile, (size_t)a.m.n_tiles);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_tsmooth(TsArgs a, int cap, int max_ent) {
  extern __shared__ double lds[];
  ts_body<MODE>(a, cap, max_ent, lds, (int)blockIdx.x);
}

size_t ts_lds_bytes(int T, int cap, int max_ent) {
  return (6 * (size_t)cap + 9 * (size_t)T) * sizeof(double) + 2 * ((size_t)((max_ent + 3) & ~3)) + 64;
}

hipError_t launch_ts(const TsArgs& a, int mode, int cap, int max_ent, hipStream_t s) {
  const int nb = a.tile1 - a.tile0;
  if (nb <= 0) return hipSuccess;
  const size_t lds = ts_lds_bytes(a.m.T, cap, max_ent);
  hipError_t e;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_TS, mode, cap, max_ent, nb, 0, lds, &a, sizeof(a), &a.m);
#define MS_LAUNCH_S(M)                                                                      \
  do {                                                                                      \
    e = ensure_lds(k_tsmooth<M>, lds);                                                      \
    if (e != hipSuccess) return e;                                                          \
    hipLaunchKernelGGL((k_tsmooth<M>), dim3(nb), dim3(a.m.T), lds, s, a, cap, max_ent);     \
  } while (0)
  if (mode == 0) MS_LAUNCH_S(0); else if (mode == 1) MS_LAUNCH_S(1); else MS_LAUNCH_S(2);
#undef MS_LAUNCH_S
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Tilt relaxation vector ops (runtime/steppers/tilt_relaxation.py:237-424), one workgroup per
// tile so the partials line up with k_reduce.
//   mode 0 PREP : tg[tilt-fixed] = 0 ; partials |tg|^2 (free rows) and <r, M^-1 r>, r = -tg
//   mode 1 DIR  : dir = z (first) or z + beta dir, z = -tg * Minv            (:360-368,:410-421)
//   mode 2 TRIAL: out = P(t + step*src) with P = tangent projection on the frozen normals;
//                 tilt-fixed rows keep t unless keep_fixed == 0 (the initial projection)
//   mode 3 MINV : out[v] (the accumulated Jacobi diagonal) -> its clamped inverse
// ---------------------------------------------------------------------------
// (stream kernels as device bodies too: arguments in one struct, the block index a parameter; blockDim.x = BLOCK)
__device__ __forceinline__ void tvec_body(const TvecArgs& a, int block_id) {
  __shared__ double red[16];
  const int mode = a.mode, nv = a.nv, T = a.T, flag = a.flag, n_tiles = a.n_tiles, s_gn2 = a.s_gn2, s_rz = a.s_rz;
  const uint8_t* const vflags = a.vflags;
  double* const tg = a.tg;
  const double* const minv = a.minv;
  double* const dir = a.dir;
  const double* const tilts = a.tilts;
  const double* const src = a.src;
  const double* const normals = a.normals;
  double* const out = a.out;
  double* const partials = a.partials;
  const double coef = a.coef_dev != nullptr ? ld_agent(a.coef_dev) : a.coef;
  const uint8_t fixed_bit = a.fixed_bit;
  const int tile = a.tile0 + block_id;
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < T; i += BLOCK) {
    const int v = tile * T + i;
    if (v >= nv) break;
    const size_t o = 3 * (size_t)v;
    const bool tfix = vflags[v] & fixed_bit;
    if (mode == 3) {
      // Jacobi preconditioner from the accumulated diagonal: 1 where <= 1e-12 and on tilt-fixed
      // rows (runtime/preconditioners.py:57-59)
      double dg = out[v];
      if (!(dg > 1.0e-12) || tfix) dg = 1.0;
      out[v] = 1.0 / dg;
    } else if (mode == 0) {
      V3 g = mk(tg[o], tg[o + 1], tg[o + 2]);
      if (tfix) {
        g = mk(0, 0, 0);
        tg[o] = tg[o + 1] = tg[o + 2] = 0.0;
      }
      const double gg = dot_pinned(g, g);
      s0 += gg;
      s1 += gg * minv[v];
    } else if (mode == 4) {
      // leaflet magnitude modules with positions frozen: E = 1/2 k sum |t_v|^2 A_v, dE/dt = k t_v A_v
      // (runtime/evaluation_manager.py:565-581, 663-695); `minv` carries the vertex areas, `coef` = k
      const V3 tv = mk(tilts[o], tilts[o + 1], tilts[o + 2]);
      const double av = minv[v];
      s0 += (dot(tv, tv)) * av;
      if (flag) {
        tg[o] = coef * tv.x * av;
        tg[o + 1] = coef * tv.y * av;
        tg[o + 2] = coef * tv.z * av;
      }
    } else if (mode == 1) {
      const double m = minv[v];
      const V3 z = mk(-tg[o] * m, -tg[o + 1] * m, -tg[o + 2] * m);
      if (flag) {
        dir[o] = z.x;
        dir[o + 1] = z.y;
        dir[o + 2] = z.z;
      } else {
        dir[o] = z.x + coef * dir[o];
        dir[o + 1] = z.y + coef * dir[o + 1];
        dir[o + 2] = z.z + coef * dir[o + 2];
      }
    } else {
      const V3 t0 = mk(tilts[o], tilts[o + 1], tilts[o + 2]);
      V3 r = t0;
      if (!(tfix && flag))
        r = tilt_trial_row(t0, mk(src[o], src[o + 1], src[o + 2]), mk(normals[o], normals[o + 1], normals[o + 2]), coef);
      out[o] = r.x;
      out[o + 1] = r.y;
      out[o + 2] = r.z;
    }
  }
  if (mode == 4) {
    const double r = block_reduce(s0, 0, red);
    if (threadIdx.x == 0) partials[(size_t)s_gn2 * n_tiles + tile] = (0.5 * coef) * r;
  }
  if (mode == 0) {
    double* po = partials + tile;
    const size_t ps = (size_t)n_tiles;
    double r = block_reduce(s0, 0, red);
    if (threadIdx.x == 0) po[s_gn2 * ps] = r;
    r = block_reduce(s1, 0, red);
    if (threadIdx.x == 0) po[s_rz * ps] = r;
  }
}

__global__ __launch_bounds__(BLOCK) void k_tvec(TvecArgs a) { tvec_body(a, (int)blockIdx.x); }

// the same vector op on the two leaflet fields in one launch (grid y = field)
__global__ __launch_bounds__(BLOCK) void k_tvec2(TvecArgs a, TvecArgs b) {
  if (blockIdx.y == 0) tvec_body(a, (int)blockIdx.x);
  else tvec_body(b, (int)blockIdx.x);
}
hipError_t launch_tvec2(const TvecArgs& a, const TvecArgs& b, int n_blocks, hipStream_t s) {
  if (n_blocks <= 0) return hipSuccess;
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  hipLaunchKernelGGL(k_tvec2, dim3(n_blocks, 2), dim3(BLOCK), 0, s, a, b);
  return hipGetLastError();
}

hipError_t launch_tvec(int mode, int tile0, int tile1, int nv, int T, const uint8_t* vflags, double* tg,
                       const double* minv, double* dir, const double* tilts, const double* src,
                       const double* normals, double* out, double coef, int flag, double* partials,
                       int n_tiles, hipStream_t s, uint8_t fixed_bit, int s_gn2, int s_rz) {
  if (tile1 <= tile0) return hipSuccess;
  TvecArgs a;
  a.mode = mode; a.tile0 = tile0; a.nv = nv; a.T = T; a.vflags = vflags; a.tg = tg; a.minv = minv; a.dir = dir;
  a.tilts = tilts; a.src = src; a.normals = normals; a.out = out; a.coef = coef; a.flag = flag; a.partials = partials;
  a.n_tiles = n_tiles; a.fixed_bit = fixed_bit; a.s_gn2 = s_gn2; a.s_rz = s_rz; a.coef_dev = nullptr;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_TVEC, 0, 0, 0, tile1 - tile0, 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_tvec, dim3(tile1 - tile0), dim3(BLOCK), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Deterministic second stage: one workgroup folds the per-tile partials of the
// requested slots in a fixed order.  Volume gets its 1/6 here
// (geometry/body.py:121: vol_contrib.sum() / 6.0).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// k_disk_target: the soft disk tilt-profile target (modules/energy/tilt_disk_target_in.py:160-286).
//   mode 0: largest in-plane distance |r_vec| of a tagged row (the default disk radius, :216-218), per-tile max
//   mode 1: diff = t - theta(r) r_hat on the tagged rows, 0 elsewhere (:237-243); the energy / gradients are the
//           tilt magnitude kernel applied to diff with k = strength (:256-284)
// One workgroup per tile so the partial lines up with k_reduce.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double bessel_i1_series30(double x) {  // :148-157, same operation order
  const double t = 0.5 * x, t2 = t * t;
  double term = t, out = t;
  for (int k = 1; k < 30; ++k) {
    term = term * (t2 / (double)(k * (k + 1)));
    out = out + term;
  }
  return out;
}

__device__ __forceinline__ void disk_target_body(const DiskTargetArgs& a, int mode, int block_id) {
  __shared__ double red[16];
  const int tile = a.tile0 + block_id;
  double rmax = 0.0;
  double R = a.radius, den = 1.0;
  bool off = false;
  if (mode == 3) {
    // positions frozen (a tilt relaxation): theta(r) r_hat of every tagged row was written to a.target once (mode 2);
    // the difference field is t - target, the same two roundings as mode 1 takes
    for (int i = threadIdx.x; i < a.T; i += BLOCK) {
      const int v = tile * a.T + i;
      if (v >= a.nv) break;
      const size_t o = 3 * (size_t)v;
      V3 df = mk(0, 0, 0);
      if (a.disk[v] && a.target[o + 0] == a.target[o + 0])  // (NaN in the cache: the profile is switched off, :219-228)
        df = mk(a.tilts[o] - a.target[o], a.tilts[o + 1] - a.target[o + 1], a.tilts[o + 2] - a.target[o + 2]);
      a.diff[o] = df.x;
      a.diff[o + 1] = df.y;
      a.diff[o + 2] = df.z;
    }
    return;
  }
  if (mode == 1 || mode == 2) {
    if (!(R > 0.0)) R = ld_agent(a.scal + a.r_slot);
    off = !(R > 0.0);                         // :219-220
    if (!off && !(fabs(a.lambda) < 1.0e-12)) {
      den = bessel_i1_series30(a.lambda * R);
      off = fabs(den) < 1.0e-15;              // :227-228
    }
  }
  for (int i = threadIdx.x; i < a.T; i += BLOCK) {
    const int v = tile * a.T + i;
    if (v >= a.nv) break;
    const size_t o = 3 * (size_t)v;
    V3 df = mk(0, 0, 0);
    if (a.disk[v]) {
      V3 x = mk(a.x[o], a.x[o + 1], a.x[o + 2]);
      if (a.d && !(a.vflags[v] & VF_FIXED))
        x = mk(axpy1(x.x, a.alpha, a.d[o]), axpy1(x.y, a.alpha, a.d[o + 1]), axpy1(x.z, a.alpha, a.d[o + 2]));
      const V3 c = mk(a.center[0], a.center[1], a.center[2]), n = mk(a.normal[0], a.normal[1], a.normal[2]);
      V3 r = x - c;
      const double rn = dot(r, n);
      r = mk(r.x - rn * n.x, r.y - rn * n.y, r.z - rn * n.z);
      const double rl = norm(r);
      if (mode == 0) {
        rmax = fmax(rmax, rl);
      } else if (!off) {
        V3 rh = mk(0, 0, 0);
        if (rl > 1.0e-12) rh = mk(r.x / rl, r.y / rl, r.z / rl);
        double theta;
        if (fabs(a.lambda) < 1.0e-12) theta = (a.theta_b * rl) / R;
        else theta = (a.theta_b * bessel_i1_series30(a.lambda * rl)) / den;
        // theta(r) r_hat, rounded, THEN the difference (the reference's two numpy operations; also what lets a cached
        // target field -- mode 2 / 3 -- give the same doubles)
        const V3 tg = mk(__dmul_rn(theta, rh.x), __dmul_rn(theta, rh.y), __dmul_rn(theta, rh.z));
        if (mode == 2) {
          a.target[o] = tg.x;
          a.target[o + 1] = tg.y;
          a.target[o + 2] = tg.z;
        } else {
          df = mk(a.tilts[o] - tg.x, a.tilts[o + 1] - tg.y, a.tilts[o + 2] - tg.z);
        }
      } else if (mode == 2) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        a.target[o] = a.target[o + 1] = a.target[o + 2] = qnan;  // (profile off: mode 3 leaves the difference at zero)
      }
    }
    if (mode == 1) {
      a.diff[o] = df.x;
      a.diff[o + 1] = df.y;
      a.diff[o + 2] = df.z;
    }
  }
  if (mode == 0) {
    const double r = block_reduce(rmax, 2, red);
    if (threadIdx.x == 0) a.partials[(size_t)a.r_slot * a.n_tiles + tile] = r;
  }
}

__global__ __launch_bounds__(BLOCK) void k_disk_target(DiskTargetArgs a, int mode) {
  disk_target_body(a, mode, (int)blockIdx.x);
}

hipError_t launch_disk_target(const DiskTargetArgs& a, int mode, hipStream_t s) {
  if (a.tile1 <= a.tile0) return hipSuccess;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_DISK, mode, 0, 0, a.tile1 - a.tile0, 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_disk_target, dim3(a.tile1 - a.tile0), dim3(BLOCK), 0, s, a, mode);
  return hipGetLastError();
}

constexpr int RBLOCK = 512;  // 512 threads x 8 loads in flight cover 4096 tiles in one round trip
constexpr int RU = 8;
__device__ __forceinline__ int fold_op(int slot) {  // 0 sum, 1 min, 2 max
  return (slot == MS_S_MINEDGE2) ? 1 : ((slot == MS_S_GUARD || slot == MS_S_MAXD2 || slot == MS_S_MAXG2 ||
                                         slot == MS_S_DTR_IN || slot == MS_S_DTR_OUT) ? 2 : 0);
}
// The ordered fold of one slot over the tiles [tile0, tile1), by the whole workgroup, in two halves so that the loads
// of the NEXT slot can be in flight while this one is reduced: fold_issue requests the first 4096 tiles' partials (8 per
// thread: thread i takes tiles tile0 + i, + 512, ...), fold_finish adds them up in that order, walks on if the range is
// longer, and combines the threads (wave DPP tree, then the eight waves in order).  The result is valid in thread 0.
// Partials are slot-major: the lanes of a wave read consecutive doubles.
__device__ __forceinline__ void fold_issue(double (&q)[RU], const double* partials, int n_tiles, int tile0, int tile1,
                                           int slot) {
  const double* p = partials + (size_t)slot * n_tiles;
  const double neutral = fold_op(slot) == 1 ? 1.0e300 : 0.0;
  const int t = tile0 + (int)threadIdx.x;
#pragma unroll
  for (int k = 0; k < RU; ++k) q[k] = t + k * RBLOCK < tile1 ? p[t + k * RBLOCK] : neutral;
}
__device__ __forceinline__ double fold_finish(const double (&q0)[RU], const double* partials, int n_tiles, int tile0,
                                              int tile1, int slot, double* red) {
  const int op = fold_op(slot);
  const double* p = partials + (size_t)slot * n_tiles;
  const double neutral = op == 1 ? 1.0e300 : 0.0;
  double v = neutral;
  double q[RU];
#pragma unroll
  for (int k = 0; k < RU; ++k) q[k] = q0[k];
  for (int t = tile0 + (int)threadIdx.x; t < tile1; t += RU * RBLOCK) {
    if (t >= tile0 + RU * RBLOCK) {
#pragma unroll
      for (int k = 0; k < RU; ++k) q[k] = t + k * RBLOCK < tile1 ? p[t + k * RBLOCK] : neutral;
    }
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      if (op == 0) {
        if (t + k * RBLOCK < tile1) v = v + q[k];
      } else if (op == 1) {
        v = fmin(v, q[k]);
      } else {
        v = fmax(v, q[k]);
      }
    }
  }
  v = block_reduce(v, op, red);
  return (slot == MS_S_VOL) ? v / 6.0 : v;
}
// Mailbox entries live in pinned, device-mapped host memory: {value bits, tag} with tag = ticket XOR value bits, both
// written with system-scope write-through stores (`sc0 sc1`, no cache holds them) and NO ordering between them.  The host
// accepts an entry when value XOR tag equals the ticket it waits for: of the four (value, tag) combinations a reader can
// see while the two stores land in either order, the test passes only for pairs whose value word IS the new value (old
// value with new tag passes only if old == new; new value with old tag only if it completes the same ticket; old with
// old never).  So neither a 16-byte store (observed untorn on gfx950, no architectural guarantee) nor a fence between
// value and sequence word is needed -- the `__threadfence_system` per posted slot was most of this kernel's time, and a
// wait for the value's PCIe acknowledgement before the sequence word the next largest part.
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void post_entry(unsigned long long* box, int entry, unsigned long long bits,
                                           unsigned long long ticket) {
  st_sys(box + 2 * entry, bits);
  st_sys(box + 2 * entry + 1, bits ^ ticket);
}
// One workgroup per (set, slot) task -- the MS_P_RAN count of a gated producer, the energy slots of every set, then the
// remaining slots -- each on a CU of its own: ONE CU cannot keep enough cache lines in flight to read several 32 KB
// partial rows at HBM/Infinity-Cache latency (a single-workgroup fold of five slots measured 9-13 us against 4.6 us
// for five workgroups).  Thread 0 of a workgroup stores its result to the device scalars (agent scope: later kernels
// read them with ld_agent) and posts it to its mailbox entry.
// When the fold closes a line-search stage (dec_out) the energy workgroups count themselves in on an agent-scope
// counter once their scalar store has completed; the one whose add comes last reads all the stage's energies back with
// agent-scope loads and takes the Armijo decision ONCE from exactly the doubles that were posted -- the first trial j
// whose energy passes rhs[j] is the accepted one, as in line_search.py:386-392 -- and publishes the code.
// Every workgroup reads the gate word (one word, written by an earlier kernel); with the gate closed nothing is stored
// or posted, workgroup 0 only hands the earlier decision on (later stages and the gradient pass test THIS stage's
// word) and checks that no workgroup of the gated producer ran.
// task index of a fold launch -> (set, slot): task 0 the ran count of a gated producer (check_ran), then the energy
// slots of every set, then the remaining slots of the last set (side_full: of every set); slot < 0: no such task
__device__ __forceinline__ void fold_task(const FoldArgs& a, int task, uint32_t e_mask, uint32_t rest_mask, int& set, int& slot) {
  const int n_e = __popc(e_mask);
  const int n_rest = __popc(rest_mask);
  const int n_reg = a.side_full ? a.n_sets : 1;
  const int t_e = a.check_ran ? 1 : 0;
  const int t_rest = t_e + a.n_sets * n_e;
  slot = -1;
  set = a.n_sets - 1;
  uint32_t m = 0;
  int k = 0;
  if (task < t_e) {
    slot = MS_P_RAN;
  } else if (task < t_rest) {  // energy slot k of set `set`
    set = (task - t_e) / n_e;
    k = (task - t_e) % n_e;
    m = e_mask;
  } else {
    if (n_rest == 0) return;
    const int r = task - t_rest;
    if (r >= n_reg * n_rest) return;
    set = a.n_sets - n_reg + r / n_rest;
    k = r % n_rest;
    m = rest_mask;
  }
  for (int s = 0; s < MS_NSCAL && slot < 0; ++s)
    if (m & (1u << s)) {
      if (k == 0) slot = s;
      --k;
    }
}

// The decision a fold takes once all of its sums are stored (k_reduce: by the workgroup that arrives last; the
// one-workgroup fold: by thread 0 behind a barrier): does the search happen at all (merged folds), which trial passes
// its Armijo test first -- the host's expressions with the host's roundings -- and the code every gated kernel behind
// the stage reads.
__device__ __forceinline__ void fold_decide(const FoldArgs& a, uint32_t e_mask, bool merged, const double (&rhs_d)[MS_MAX_TRIALS]) {
  uint32_t code = DEC_CONTINUE;
  bool decide = true;
  double m_e0 = 0.0, m_c = 0.0, m_al = 0.0, m_beta = 0.0, m_slope = 0.0;
  if (merged) {
    // first: does the search the host queued this launch for happen at all?  (FoldArgs::go_kind: 1 -- the direction
    // with history is no descent direction, the stepper restarts along -g; kind 2 -- the direction just written is one;
    // not converged; unguarded range)
    double* const sc = a.set[a.n_sets - 1].scal;
    const double gn2 = ld_agent(sc + MS_S_GNORM2), gdd = ld_agent(sc + MS_S_GDOTD);
    const double md2 = ld_agent(sc + MS_S_MAXD2), mg2 = ld_agent(sc + MS_S_MAXG2);
    const bool restart = a.go_kind == 1;
    m_slope = restart ? -gn2 : gdd;  // <g,d> of the search
    const bool kind_ok = restart ? gdd >= 0.0 : gdd < 0.0;
    if (!(kind_ok && gn2 > a.go_tol2 && (restart ? mg2 : md2) < a.go_lim)) {
      code = DEC_STOP;
      decide = false;
    }
    m_e0 = a.go_e0;
    m_c = a.go_c;
    m_al = a.go_alpha0;
    m_beta = a.go_beta;
  }
  for (int j = 0; j < a.n_sets && decide; ++j) {
    double E = 0.0;
    bool first = true;
    for (int s = 0; s < MS_NSCAL; ++s)
      if (e_mask & (1u << s)) {  // (slot order: the host adds surface + bending in this order)
        const double v = ld_agent(a.set[j].scal + s);
        E = first ? v : E + v;
        first = false;
      }
    double rhs = a.rhs[j];
    if (merged) {
      // energy0 + (c alpha_j) <g,d> with alpha_j = alpha_{j-1} beta: the host's expressions, rounding for rounding
      rhs = __dadd_rn(m_e0, __dmul_rn(__dmul_rn(m_c, m_al), m_slope));
      m_al = __dmul_rn(m_al, m_beta);
    } else if (a.rhs_dev != nullptr) {
#pragma unroll
      for (int k = 0; k < MS_MAX_TRIALS; ++k)
        if (k == j) rhs = rhs_d[k];
    }
    if (E <= rhs) {
      code = j == a.n_sets - 1 ? DEC_ACCEPT_MAIN : DEC_ACCEPT_SIDE;
      break;
    }
  }
  if (merged && decide && a.go_out != nullptr) {
    // the gated stages behind this launch test later alphas of the same ladder: leave them their right-hand sides
    double* const rhs = reinterpret_cast<double*>(a.go_out + 2);
    for (int k = 0; k < 8 - a.n_sets; ++k) {
      st_agent(rhs + a.n_sets + k, __dadd_rn(m_e0, __dmul_rn(__dmul_rn(m_c, m_al), m_slope)));
      m_al = __dmul_rn(m_al, m_beta);
    }
  }
  st_agent(a.dec_out, code);
  if (a.set[a.n_sets - 1].host_box) post_entry(a.set[a.n_sets - 1].host_box, MS_MB_DEC, (unsigned long long)code, a.ticket);
}

__device__ __forceinline__ void reduce_body(const FoldArgs& a, int block_id) {
  __shared__ double red[16];
  const uint32_t e_mask = a.slot_mask & a.e_mask;
  const uint32_t rest_mask = a.slot_mask & ~e_mask;
  const uint32_t prev = a.gate != nullptr ? ld_agent(a.gate) : a.gate_want;  // (consumed after the fold's loads)
  const int n_e = __popc(e_mask);                       // energy slots per set (<= 2)
  const int n_rest = __popc(rest_mask);                 // other slots per set
  const int n_reg = a.side_full ? a.n_sets : 1;         // sets whose other slots are folded too: all, or the last trial's
  const int t_e = a.check_ran ? 1 : 0;                  // first energy task (task 0: the ran count)
  const int t_rest = t_e + a.n_sets * n_e;
  const int task = block_id;
  int set, slot;
  fold_task(a, task, e_mask, rest_mask, set, slot);
  if (slot < 0) return;
  // merged: this launch folds the direction scalars of the gradient pass in front of the energy launch TOGETHER with
  // that launch's energies (the energy launch did not wait for them: one fold and one kernel boundary per step less)
  const bool merged = a.go_kind != 0 && a.dec_out != nullptr;
  double rhs_d[MS_MAX_TRIALS];
  if (a.rhs_dev != nullptr && a.dec_out != nullptr && task >= t_e && task < t_rest && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < MS_MAX_TRIALS; ++j) rhs_d[j] = j < a.n_sets ? ld_agent(a.rhs_dev + j) : 0.0;
  }
  double q[RU];
  fold_issue(q, a.set[set].partials, a.n_tiles, a.tile0, a.tile1, slot);
  const bool open = prev == a.gate_want;
  if (!open && task != 0) return;
  const double r = fold_finish(q, a.set[set].partials, a.n_tiles, a.tile0, a.tile1, slot, red);
  if (threadIdx.x != 0) return;
  if (slot == MS_P_RAN) {
    const double want = open ? (double)(a.tile1 - a.tile0) : 0.0;
    if (r != want && a.host_err)
      st_sys(a.host_err, ((unsigned long long)a.ticket << 24) | (unsigned long long)(unsigned int)r | (1ull << 63));
  }
  if (!open) {  // (task 0) an earlier stage has decided (or failed): later readers of THIS stage's word see the same
    if (a.dec_out != nullptr) st_agent(a.dec_out, prev);
    return;
  }
  if (slot == MS_P_RAN) return;
  if (a.set[set].scal) st_agent(a.set[set].scal + slot, r);
  if (a.set[set].host_box) post_entry(a.set[set].host_box, slot, (unsigned long long)__double_as_longlong(r), a.ticket);
#if MS_GATE_PROBE
  if (set == a.n_sets - 1) g_shadow[slot] = r;
#endif
  if (a.dec_out == nullptr || (!merged && task >= t_rest)) return;
  // a stage that is decided here: its energies' workgroups (merged: every slot's) count in once their scalar store has
  // completed; the one that comes last decides
  // Hand-over to the last arriver.  Release side: the scalar store above is an agent-scope write-through store and has
  // COMPLETED (vmcnt 0) before this workgroup counts itself in.  Acquire side: the decider's loads below are agent-scope
  // loads that are control-dependent on the value this read-modify-write returns, and the signal fence keeps the
  // compiler from moving them above it.  The formal form -- one __ATOMIC_ACQ_REL read-modify-write, -DMS_REDUCE_ACQREL
  // -- adds a cache write-back and an invalidate per arriving lane: measured +1..1.5 us per fold (8.3 -> 9.6 us by HIP
  // events, headline 17.5 -> 17.2 k steps/s, gpurun_out/r4a); the host's replay of every decision catches a wrong one
  // either way.
#ifdef MS_REDUCE_ACQREL
  const uint32_t arrived = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t arrived = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
  if (arrived != (uint32_t)(a.n_sets * n_e + (merged ? n_reg * n_rest : 0) - 1)) return;
  st_agent(a.counter, 0u);  // (the next fold of the stream starts from zero)
  fold_decide(a, e_mask, merged, rhs_d);
}

// The same fold for a range of ONE tile, by ONE workgroup (the interpreter k_exec): thread t does task t -- a fold
// over one tile is that tile's partial through the fold's neutral element, the arithmetic of fold_finish / block_reduce
// for a single contributor (sums: 0.0 + x; min / max against the neutral; the volume's 1/6) -- all stores and mailbox
// posts of the launch leave together, and thread 0 decides behind one barrier.  Bit for bit k_reduce's results.
__device__ __forceinline__ void reduce_one_tile(const FoldArgs& a, int n_tasks) {
  const uint32_t e_mask = a.slot_mask & a.e_mask;
  const uint32_t rest_mask = a.slot_mask & ~e_mask;
  const uint32_t prev = a.gate != nullptr ? ld_agent(a.gate) : a.gate_want;
  const bool open = prev == a.gate_want;
  const bool merged = a.go_kind != 0 && a.dec_out != nullptr;
  const int task = (int)threadIdx.x;
  int set = 0, slot = -1;
  if (task < n_tasks) fold_task(a, task, e_mask, rest_mask, set, slot);
  if (slot >= 0 && (open || task == 0)) {
    double x = 0.0;
    // (the set is picked with constant indices: a dynamically indexed argument struct would move to scratch)
#pragma unroll
    for (int j = 0; j < MS_MAX_TRIALS; ++j)
      if (j == set) x = a.set[j].partials[(size_t)slot * a.n_tiles + a.tile0];
    const int op = fold_op(slot);
    double r = op == 0 ? 0.0 + x : (op == 1 ? fmin(1.0e300, x) : fmax(0.0, x));
    if (a.tile1 <= a.tile0) r = op == 1 ? 1.0e300 : 0.0;
    if (slot == MS_S_VOL) r = r / 6.0;
    if (slot == MS_P_RAN) {
      const double want = open ? (double)(a.tile1 - a.tile0) : 0.0;
      if (r != want && a.host_err)
        st_sys(a.host_err, ((unsigned long long)a.ticket << 24) | (unsigned long long)(unsigned int)r | (1ull << 63));
    }
    if (!open) {
      if (a.dec_out != nullptr) st_agent(a.dec_out, prev);
    } else if (slot != MS_P_RAN) {
#pragma unroll
      for (int j = 0; j < MS_MAX_TRIALS; ++j)
        if (j == set) {
          if (a.set[j].scal) st_agent(a.set[j].scal + slot, r);
          if (a.set[j].host_box) post_entry(a.set[j].host_box, slot, (unsigned long long)__double_as_longlong(r), a.ticket);
        }
    }
  }
  __syncthreads();  // (every store above has completed)
  if (threadIdx.x == 0 && open && a.dec_out != nullptr) {
    double rhs_d[MS_MAX_TRIALS];
#pragma unroll
    for (int j = 0; j < MS_MAX_TRIALS; ++j) rhs_d[j] = (a.rhs_dev != nullptr && j < a.n_sets) ? ld_agent(a.rhs_dev + j) : 0.0;
    fold_decide(a, e_mask, merged, rhs_d);
  }
}

__global__ __launch_bounds__(RBLOCK) void k_reduce(FoldArgs a) { reduce_body(a, (int)blockIdx.x); }

hipError_t launch_reduce(const FoldArgs& a, hipStream_t s) {
  if (a.n_sets < 1 || a.n_sets > MS_MAX_TRIALS) return hipErrorInvalidValue;
  const uint32_t em = a.slot_mask & a.e_mask;
  const int nb = (a.check_ran ? 1 : 0) + a.n_sets * __builtin_popcount(em) +
                 (a.side_full ? a.n_sets : 1) * __builtin_popcount(a.slot_mask & ~em);
  if (nb == 0) return hipSuccess;
  if (a.dec_out != nullptr && (a.counter == nullptr || em == 0)) return hipErrorInvalidValue;
  if (a.go_kind != 0 && (a.dec_out == nullptr || a.go_out == nullptr)) return hipErrorInvalidValue;
  // (one workgroup: the fold's tasks run one after the other; with one tile a fold is a copy of that tile's partials)
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_REDUCE, 0, 0, 0, nb, 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(RBLOCK), 0, s, a);
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
__device__ __forceinline__ void direction_body(const DirectionArgs& a, int block_id) {
  __shared__ double red[16];
  const int nv = a.nv, T = a.T, use_constraint = a.use_constraint, cg_history = a.cg_history, n_tiles = a.n_tiles;
  const int write_g = a.write_g, pd_neg_pg = a.pd_neg_pg, precond = a.precond;
  const uint8_t* const vflags = a.vflags;
  double* const g = a.g;
  const double* const gC = a.gC;
  double* const d = a.d;
  const double* const pg = a.pg;
  const double* const pd = a.pd;
  const double* const scal = a.scal;
  double* const partials = a.partials;
  const uint32_t* const gate = a.gate;
  const uint32_t gate_want = a.gate_want;
  const int tile = a.tile0 + block_id;
  if (gate != nullptr && !gate_open(gate, gate_want, partials + (size_t)MS_P_RAN * n_tiles + tile)) return;
  double lam = 0.0;
  bool project = false;
  if (use_constraint) {
    const double nsq = ld_agent(scal + MS_S_GCGC);
    if (nsq > 1.0e-18) {
      lam = ld_agent(scal + MS_S_GGC) / nsq;
      project = true;
    }
  }
  double gn2 = 0.0, gd = 0.0, md2 = 0.0, mg2 = 0.0;
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
    // conjugate_gradient.py:74-76 precondition: the direction is built from g_i / (|g_i| + 1e-8) (norm as numpy
    // takes it: the squares added left to right); history, |g|^2 and <g,d> keep the raw rows
    V3 gh = gi;
    if (precond) {
      const double nrm = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(gi.x, gi.x), __dmul_rn(gi.y, gi.y)), __dmul_rn(gi.z, gi.z)));
      const double den = nrm + 1.0e-8;
      gh = mk(gi.x / den, gi.y / den, gi.z / den);
    }
    V3 di = -gh;
    if (cg_history) {
      const V3 p = mk(pg[o], pg[o + 1], pg[o + 2]);
      const double beta = dot_pinned(gh, gh - p) / (dot_pinned(p, p) + 1.0e-20);
      if (!(beta < 0.0)) {
        // (pd_neg_pg: the previous direction was an implicit -PG -- derived, not loaded, as in the fused epilogue)
        const V3 q = pd_neg_pg ? -p : mk(pd[o], pd[o + 1], pd[o + 2]);
        di = mk(fma(beta, q.x, -gh.x), fma(beta, q.y, -gh.y), fma(beta, q.z, -gh.z));
      }
    }
    if (fixed) di = mk(0, 0, 0);
    if (write_g) {  // (a finalized g -- restart after a failed search -- is left as it is)
      g[o] = gi.x;
      g[o + 1] = gi.y;
      g[o + 2] = gi.z;
    }
    d[o] = di.x;
    d[o + 1] = di.y;
    d[o + 2] = di.z;
    const double g2 = dot_pinned(gi, gi);
    gn2 += g2;
    gd += dot_pinned(gi, di);
    if (!fixed) md2 = fmax(md2, dot_pinned(di, di));
    mg2 = fmax(mg2, g2);  // (gi is zero on fixed rows) max|d_i|^2 of a steepest-descent restart on this gradient
  }
  double* out = partials + tile;
  const size_t ps = (size_t)n_tiles;
  double r = block_reduce(gn2, 0, red);
  if (threadIdx.x == 0) out[MS_S_GNORM2 * ps] = r;
  r = block_reduce(gd, 0, red);
  if (threadIdx.x == 0) out[MS_S_GDOTD * ps] = r;
  r = block_reduce(md2, 2, red);
  if (threadIdx.x == 0) out[MS_S_MAXD2 * ps] = r;
  r = block_reduce(mg2, 2, red);
  if (threadIdx.x == 0) out[MS_S_MAXG2 * ps] = r;
}

__global__ __launch_bounds__(BLOCK) void k_direction(DirectionArgs a) { direction_body(a, (int)blockIdx.x); }

hipError_t launch_direction(int tile0, int tile1, int nv, int T, const uint8_t* vflags, double* g,
                            const double* gC, double* d, const double* pg, const double* pd,
                            const double* scal, int use_constraint, int cg_history,
                            double* partials, int n_tiles, int write_g, hipStream_t s, const uint32_t* gate,
                            uint32_t gate_want, int pd_neg_pg, int precond) {
  if (tile1 <= tile0) return hipSuccess;
  DirectionArgs a;
  a.tile0 = tile0; a.nv = nv; a.T = T; a.vflags = vflags; a.g = g; a.gC = gC; a.d = d; a.pg = pg; a.pd = pd;
  a.scal = scal; a.use_constraint = use_constraint; a.cg_history = cg_history; a.partials = partials;
  a.n_tiles = n_tiles; a.write_g = write_g; a.gate = gate; a.gate_want = gate_want; a.pd_neg_pg = pd_neg_pg;
  a.precond = precond;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_DIRECTION, 0, 0, 0, tile1 - tile0, 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_direction, dim3(tile1 - tile0), dim3(BLOCK), 0, s, a);
  return hipGetLastError();
}

// <g, gC> per tile, for module sets that add into g AFTER the gradient kernel computed that inner product in its
// epilogue (the tilt magnitude / disk-target / leaflet bending_tilt shape gradients): the KKT multiplier of
// runtime/constraint_manager.py:293-301 is taken of the COMPLETE gradient.  <gC, gC> is unchanged.
__device__ __forceinline__ void row_dot_body(const RowDotArgs& a, int block_id) {
  __shared__ double red[16];
  const int nv = a.nv, T = a.T, n_tiles = a.n_tiles;
  const double* const g = a.g;
  const double* const gC = a.gC;
  double* const partials = a.partials;
  const int tile = a.tile0 + block_id;
  double acc = 0.0;
  for (int i = threadIdx.x; i < T; i += BLOCK) {
    const int v = tile * T + i;
    if (v >= nv) break;
    const size_t o = 3 * (size_t)v;
    acc += g[o] * gC[o] + g[o + 1] * gC[o + 1] + g[o + 2] * gC[o + 2];
  }
  const double r = block_reduce(acc, 0, red);
  if (threadIdx.x == 0) partials[(size_t)MS_S_GGC * n_tiles + tile] = r;
}

__global__ __launch_bounds__(BLOCK) void k_row_dot(RowDotArgs a) { row_dot_body(a, (int)blockIdx.x); }

hipError_t launch_row_dot(int tile0, int tile1, int nv, int T, const double* g, const double* gC, double* partials,
                          int n_tiles, hipStream_t s) {
  if (tile1 <= tile0) return hipSuccess;
  RowDotArgs a;
  a.tile0 = tile0; a.nv = nv; a.T = T; a.g = g; a.gC = gC; a.partials = partials; a.n_tiles = n_tiles;
  if (ExecRecorder* r = exec_find(s)) return r->push(CK_ROWDOT, 0, 0, 0, tile1 - tile0, 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_row_dot, dim3(tile1 - tile0), dim3(BLOCK), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Shard boundary exchange (multi-GPU): a rank's BOUNDARY rows are the rows it owns that some
// other rank's tiles list as halo.  pack: send = [MS_NSCAL reduction scalars | boundary rows of up
// to 4 per-vertex buffers, interleaved per row]; after the all-gather, unpack scatters every
// peer's rows into the local buffers and lifts the per-rank scalar headers into one array.
// ---------------------------------------------------------------------------
struct RowBufs {
  double* p[4];
  int ncomp[4];
  int n, comps;  // buffers, sum of ncomp
};

__global__ void k_pack_boundary(const int32_t* rows, int n_rows, RowBufs b, const double* scal,
                                double* send) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < MS_NSCAL) send[j] = ld_agent(scal + j);
  if (j >= n_rows) return;
  const size_t v = (size_t)rows[j];
  double* o = send + MS_NSCAL + (size_t)j * b.comps;
  for (int k = 0; k < b.n; ++k)
    for (int c = 0; c < b.ncomp[k]; ++c) *o++ = b.p[k][v * b.ncomp[k] + c];
}

// peer-to-peer form of the pack: block row y writes this rank's message into peer y's slab (dst.p[y] = the slot of THIS
// rank there) with system-scope write-through stores -- the destination is another GPU's memory (or, for the own
// slot, memory the unpack kernel reads with system-scope loads)
struct PeerDst {
  double* p[16];
};
__device__ __forceinline__ void st_sys_f64(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// flags != nullptr: the block that finishes LAST for peer y raises this rank's flag word there (no flag kernel behind
// the pack): every block completes its stores (system-scope fence), then counts itself in at agent scope; the last one
// has thereby acquired all the others' and publishes the ticket with a system-scope release
__global__ void k_pack_peers(const int32_t* rows, int n_rows, RowBufs b, const double* scal, PeerDst dst,
                             const PeerFlags* flags, unsigned int* arrived, int me, unsigned long long ticket,
                             const uint32_t* gate, uint32_t gate_want) {
  // (an exchange queued behind a decision every rank takes from the same doubles: skipped by all of them or by none)
  if (gate != nullptr && ld_agent(gate) != gate_want) return;
  double* send = dst.p[blockIdx.y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < MS_NSCAL) st_sys_f64(send + j, ld_agent(scal + j));
  if (j < n_rows) {
    const size_t v = (size_t)rows[j];
    double* o = send + MS_NSCAL + (size_t)j * b.comps;
    for (int k = 0; k < b.n; ++k)
      for (int c = 0; c < b.ncomp[k]; ++c) st_sys_f64(o++, b.p[k][v * b.ncomp[k] + c]);
  }
  if (flags == nullptr) return;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int before = __hip_atomic_fetch_add(arrived + blockIdx.y, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {
      __hip_atomic_store(arrived + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch)
      __hip_atomic_store(flags->p[blockIdx.y] + me, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// queued behind the pack kernel (whose stores have completed and been released by then): lane r raises this rank's
// word on peer r
__global__ void k_flag_peers(PeerFlags f, int me, int world, unsigned long long ticket) {
  const int r = threadIdx.x;
  if (r < world) __hip_atomic_store(f.p[r] + me, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The Armijo decision of a SHARDED trial, on the device: taken by the unpack kernel of the trial's exchange once every
// rank's scalar header is in this rank's slab (the header block that arrives last), one lane adds the ranks' energy slots in rank order -- the
// host's fold, rounding for rounding (shard_exchange_take / shard_energy_of) -- and tests trial 0 (a pair launch's
// other trial, header slots SH_ALT + s) and the main trial against their right-hand sides.  Every rank reads the same
// doubles and writes the same word; the kernels queued behind it (commit, gradient + direction pass, their exchange)
// are gated on it.  The host replays the decision from the headers it receives and compares (post).
__device__ void shard_decide(const ShardDecideArgs& a) {
  auto ld = [&](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
  // the energy the host calls shard_energy_of: the ranks' slots in rank order, then the slots in module order
  auto energy_of = [&](const double* slab, int base) {
    double e = 0.0;
    for (int k = 0; k < 2; ++k) {
      const int slot = k == 0 ? a.slot_a : a.slot_b;
      if (slot < 0) continue;
      double acc = 0.0;
      for (int r = 0; r < a.world; ++r) acc += ld(slab + (size_t)r * a.stride + base + slot);
      e += acc;
    }
    e += 0.0;  // (the host adds the penalty term here: 0.0 without that module -- the only case this kernel is queued in)
    return e;
  };
  uint32_t code = DEC_CONTINUE;
  double rhs_alt = a.rhs_alt, rhs_main = a.rhs_main;
  if (a.gate != nullptr && ld_agent(a.gate) != a.gate_want) {
    code = DEC_STOP;  // (the trial this launch would decide never ran)
  } else if (a.go_kind != 0) {
    // a trial queued AHEAD of its step (ms_shard_step: the search of the restart after a history direction that is no
    // descent direction): does that search happen at all, and its right-hand sides -- the host's expressions
    const double e0 = ld_agent(a.keep_in), me2 = ld_agent(a.keep_in + 1);
    double gn2 = 0.0, gdd = 0.0;
    for (int r = 0; r < a.world; ++r) gn2 += ld(a.recv_dir + (size_t)r * a.stride + MS_S_GNORM2);
    for (int r = 0; r < a.world; ++r) gdd += ld(a.recv_dir + (size_t)r * a.stride + MS_S_GDOTD);
    double mg2 = ld(a.recv_dir + MS_S_MAXG2);
    for (int r = 1; r < a.world; ++r) mg2 = fmax(mg2, ld(a.recv_dir + (size_t)r * a.stride + MS_S_MAXG2));
    const double min_edge = a.has_faces ? sqrt(me2) : 0.0;
    const double safe = min_edge > 0.0 ? __dmul_rn(0.3, min_edge) : INFINITY;
    const double first = a.has_alt ? a.alpha_alt : a.alpha_main;
    const bool go = gdd >= 0.0 && !(sqrt(gn2) < a.tol) && __dmul_rn(first, sqrt(mg2)) < safe;
    if (!go) code = DEC_STOP;
    const double slope = -gn2;
    rhs_alt = __dadd_rn(e0, __dmul_rn(__dmul_rn(a.c1, a.alpha_alt), slope));
    rhs_main = __dadd_rn(e0, __dmul_rn(__dmul_rn(a.c1, a.alpha_main), slope));
  }
  for (int t = a.has_alt ? 0 : 1; t < 2 && code == DEC_CONTINUE; ++t) {
    const double e = energy_of(a.recv, t == 0 ? a.alt_off : 0);
    if (e <= (t == 0 ? rhs_alt : rhs_main)) code = t == 0 ? DEC_ACCEPT_SIDE : DEC_ACCEPT_MAIN;
  }
  if (a.keep_out != nullptr && code != DEC_STOP) {
    // what a trial queued ahead behind THIS trial's chain needs of it, kept outside the slab (which a fast peer may be
    // writing the exchange three tickets on into by then): the main trial's energy and min edge^2, the host's folds
    st_agent(a.keep_out, energy_of(a.recv, 0));
    double m2 = ld(a.recv + MS_S_MINEDGE2);
    for (int r = 1; r < a.world; ++r) m2 = fmin(m2, ld(a.recv + (size_t)r * a.stride + MS_S_MINEDGE2));
    st_agent(a.keep_out + 1, m2);
  }
  st_agent(a.dec_out, code);
  if (a.post != nullptr) post_entry(a.post, 0, (unsigned long long)code, a.ticket);
}
__global__ void k_shard_decide(ShardDecideArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) shard_decide(a);
}
// grid (ceil(max_rows/256), world); recv = world x stride doubles
template <bool REMOTE>
__global__ void k_unpack_boundary(const int32_t* rows_all, const int32_t* row_off, int me,
                                  RowBufs b, const double* recv, size_t stride, double* scal_all,
                                  unsigned long long* host_seq, unsigned long long ticket,
                                  const unsigned long long* wait_flags, unsigned long long wait_ticket,
                                  unsigned long long* host_err, const uint32_t* gate, uint32_t gate_want,
                                  int has_decide, ShardDecideArgs dec, unsigned int* dec_arrived) {
  if (gate != nullptr && ld_agent(gate) != gate_want) {
    // (the decision this exchange would have fed: the trial never ran)
    if (has_decide && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) shard_decide(dec);
    return;
  }
  const int r = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const double* src = recv + (size_t)r * stride;
  if (REMOTE && wait_flags != nullptr) {
    // peer-to-peer exchange: rank r's message is complete once its flag word here has reached the exchange's ticket.
    // BOUNDED wait (a peer that never arrives ends it after ~2 s with an error word for the host instead of a wave
    // that never finishes); the workgroup's other waves sit at the barrier meanwhile.
    __shared__ int arrived;
    if (threadIdx.x == 0) {
      bool ok = false;
      for (long it = 0; it < (1L << 22) && !ok; ++it) {
        ok = __hip_atomic_load(wait_flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= wait_ticket;
        if (!ok) __builtin_amdgcn_s_sleep(32);
      }
      arrived = ok ? 1 : 0;
      if (!ok && host_err) st_sys(host_err, (1ull << 62) | ((unsigned long long)r << 32) | (wait_ticket & 0xffffffffull));
    }
    __syncthreads();
    if (!arrived) return;
  }
  // REMOTE: the slab was written by other GPUs (peer-to-peer exchange): read it with system-scope loads, past this
  // GPU's caches
  auto ld = [&](const double* q) {
    return REMOTE ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *q;
  };
  if (j < MS_NSCAL) scal_all[r * MS_NSCAL + j] = ld(src + j);
  if (host_seq != nullptr && blockIdx.x == 0) {
    // the host only waits for the scalars (the rows are consumed by later kernels of the same stream): rank r's
    // header is complete once this workgroup has written it -> post its sequence word, no extra kernel
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      *reinterpret_cast<volatile unsigned long long*>(host_seq + r) = ticket;
    }
  }
  if (has_decide && blockIdx.x == 0 && threadIdx.x == 0) {
    // the header block that comes last has every rank's header in the slab behind it (each waited for its rank's flag):
    // it takes the trial's decision (no kernel of its own behind this one)
    const unsigned int before = __hip_atomic_fetch_add(dec_arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.y - 1) {
      __hip_atomic_store(dec_arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      shard_decide(dec);
    }
  }
  if (r == me) return;
  const int n_rows = row_off[r + 1] - row_off[r];
  if (j >= n_rows) return;
  const size_t v = (size_t)rows_all[row_off[r] + j];
  const double* i = src + MS_NSCAL + (size_t)j * b.comps;
  for (int k = 0; k < b.n; ++k)
    for (int c = 0; c < b.ncomp[k]; ++c) b.p[k][v * b.ncomp[k] + c] = ld(i++);
}

hipError_t launch_pack_boundary(const int32_t* rows, int n_rows, const double* const* bufs,
                                const int* ncomp, int n_bufs, const double* scal, double* send,
                                hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  RowBufs b{};
  b.n = n_bufs;
  for (int k = 0; k < n_bufs; ++k) {
    b.p[k] = const_cast<double*>(bufs[k]);
    b.ncomp[k] = ncomp[k];
    b.comps += ncomp[k];
  }
  const int n = n_rows > MS_NSCAL ? n_rows : MS_NSCAL;
  hipLaunchKernelGGL(k_pack_boundary, dim3((n + 255) / 256), dim3(256), 0, s, rows, n_rows, b, scal, send);
  return hipGetLastError();
}

hipError_t launch_unpack_boundary(const int32_t* rows_all, const int32_t* row_off, int me, int world,
                                  int max_rows, double* const* bufs, const int* ncomp, int n_bufs,
                                  const double* recv, size_t stride, double* scal_all, hipStream_t s,
                                  unsigned long long* host_seq, unsigned long long ticket, bool remote_written,
                                  const unsigned long long* wait_flags, unsigned long long wait_ticket,
                                  unsigned long long* host_err, const uint32_t* gate, uint32_t gate_want,
                                  const ShardDecideArgs* decide, unsigned int* dec_arrived) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  ShardDecideArgs dec;
  memset(&dec, 0, sizeof(dec));
  if (decide != nullptr) dec = *decide;
  const int has_decide = (decide != nullptr && dec_arrived != nullptr) ? 1 : 0;
  RowBufs b{};
  b.n = n_bufs;
  for (int k = 0; k < n_bufs; ++k) {
    b.p[k] = bufs[k];
    b.ncomp[k] = ncomp[k];
    b.comps += ncomp[k];
  }
  const int n = max_rows > MS_NSCAL ? max_rows : MS_NSCAL;
  if (remote_written)
    hipLaunchKernelGGL(k_unpack_boundary<true>, dim3((n + 255) / 256, world), dim3(256), 0, s, rows_all, row_off,
                       me, b, recv, stride, scal_all, host_seq, ticket, wait_flags, wait_ticket, host_err, gate, gate_want,
                       has_decide, dec, dec_arrived);
  else
    hipLaunchKernelGGL(k_unpack_boundary<false>, dim3((n + 255) / 256, world), dim3(256), 0, s, rows_all, row_off,
                       me, b, recv, stride, scal_all, host_seq, ticket, nullptr, 0ull, nullptr, gate, gate_want, has_decide, dec,
                       dec_arrived);
  return hipGetLastError();
}

hipError_t launch_pack_peers(const int32_t* rows, int n_rows, const double* const* bufs, const int* ncomp, int n_bufs,
                             const double* scal, double* const* dst, int world, hipStream_t s,
                             const PeerFlags* d_flags, unsigned int* d_arrived, int me, unsigned long long ticket,
                             const uint32_t* gate, uint32_t gate_want) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  if (world > 16) return hipErrorInvalidValue;
  RowBufs b{};
  b.n = n_bufs;
  for (int k = 0; k < n_bufs; ++k) {
    b.p[k] = const_cast<double*>(bufs[k]);
    b.ncomp[k] = ncomp[k];
    b.comps += ncomp[k];
  }
  PeerDst d{};
  for (int r = 0; r < world; ++r) d.p[r] = dst[r];
  const int n = n_rows > MS_NSCAL ? n_rows : MS_NSCAL;
  hipLaunchKernelGGL(k_pack_peers, dim3((n + 255) / 256, world), dim3(256), 0, s, rows, n_rows, b, scal, d, d_flags,
                     d_arrived, me, ticket, gate, gate_want);
  return hipGetLastError();
}

hipError_t launch_flag_peers(unsigned long long* const* peer_flags, int me, int world, unsigned long long ticket,
                             hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  if (world > 16) return hipErrorInvalidValue;
  PeerFlags f{};
  for (int r = 0; r < world; ++r) f.p[r] = peer_flags[r];
  hipLaunchKernelGGL(k_flag_peers, dim3(1), dim3(64), 0, s, f, me, world, ticket);
  return hipGetLastError();
}


// One word to the pinned mailbox after everything queued before it on the stream has completed
// (a kernel boundary orders the earlier kernels' host writes before this one).
__global__ void k_post_seq(unsigned long long* host_seq, unsigned long long ticket) {
  __threadfence_system();
  *reinterpret_cast<volatile unsigned long long*>(host_seq) = ticket;
}

hipError_t launch_shard_decide(const ShardDecideArgs& a, hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  hipLaunchKernelGGL(k_shard_decide, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

__global__ void k_gate_probe(const uint32_t* gate, uint32_t want, uint32_t* out) {
  *out = ld_agent(gate) == want ? 1u : 0u;
}
hipError_t launch_gate_probe(const uint32_t* gate, uint32_t want, uint32_t* out, hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  hipLaunchKernelGGL(k_gate_probe, dim3(1), dim3(1), 0, s, gate, want, out);
  return hipGetLastError();
}

hipError_t launch_post_seq(unsigned long long* host_seq, unsigned long long ticket, hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  hipLaunchKernelGGL(k_post_seq, dim3(1), dim3(1), 0, s, host_seq, ticket);
  return hipGetLastError();
}

// x[i] += coef * y[i] on movable rows of [row0, row1) and of an explicit row list (shard commit)
__global__ void k_axpy_rows(int64_t row0, int64_t row1, const int32_t* extra, int n_extra,
                            const uint8_t* vflags, double* x, const double* y, double coef, const uint32_t* gate,
                            uint32_t gate_want) {
  if (gate != nullptr && ld_agent(gate) != gate_want) return;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_own = row1 - row0;
  int64_t v;
  if (j < n_own)
    v = row0 + j;
  else if (j < n_own + n_extra)
    v = extra[j - n_own];
  else
    return;
  if (vflags[v] & VF_FIXED) return;
  for (int c = 0; c < 3; ++c) x[3 * v + c] = axpy1(x[3 * v + c], coef, y[3 * v + c]);
}

hipError_t launch_axpy_rows(int64_t row0, int64_t row1, const int32_t* extra, int n_extra,
                            const uint8_t* vflags, double* x, const double* y, double coef,
                            hipStream_t s, const uint32_t* gate, uint32_t gate_want) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  const int64_t n = (row1 - row0) + n_extra;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_axpy_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, row0, row1, extra,
                     n_extra, vflags, x, y, coef, gate, gate_want);
  return hipGetLastError();
}

// x[i] += coef * y[i] on movable rows (volume projection, constraints/volume.py:137-141)
__device__ __forceinline__ void axpy_masked_body(const AxpyMaskedArgs& a, int block_id) {
  const int64_t j = (int64_t)block_id * blockDim.x + threadIdx.x;
  if (j >= 3 * a.n_rows) return;
  if (a.vflags[j / 3] & VF_FIXED) return;
  a.x[j] = axpy1(a.x[j], a.coef, a.y[j]);
}
__global__ void k_axpy_masked(AxpyMaskedArgs a) { axpy_masked_body(a, (int)blockIdx.x); }

hipError_t launch_axpy_masked(int64_t n_rows, const uint8_t* vflags, double* x, const double* y,
                              double coef, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t n = 3 * n_rows;
  AxpyMaskedArgs a;
  a.n_rows = n_rows; a.vflags = vflags; a.x = x; a.y = y; a.coef = coef;
  if (ExecRecorder* r = exec_find(s))
    return r->push(CK_AXPY_MASKED, 0, 0, 0, (int)((n + BLOCK - 1) / BLOCK), 0, 0, &a, sizeof(a));
  hipLaunchKernelGGL(k_axpy_masked, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
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
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
  const int64_t n = (int64_t)nv * ncomp;
  hipLaunchKernelGGL(k_permute_in, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, nv,
                     perm, src_ext, dst_int, ncomp);
  return hipGetLastError();
}
hipError_t launch_permute_out(int nv, const int32_t* perm, const double* src_int, double* dst_ext,
                              int ncomp, hipStream_t s) {
  if (hipError_t es_ = exec_sync(s); es_ != hipSuccess) return es_;  // (not recorded: runs behind what was)
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

#include "ms_tsearch.inc"
#include "ms_exec.inc"
#include "ms_resident.inc"

}  // namespace ms
#if MS_GATE_PROBE
extern "C" int ms_debug_read_probe(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ms::g_probe), sizeof(unsigned long long) * 32);
}
#endif
#if MS_STAMPS
extern "C" int ms_debug_read_stamps(unsigned long long* out, int n_blocks) {
  if (n_blocks < 0) {  // the rare-path counter
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ms::g_stamps), sizeof(unsigned long long),
                                    sizeof(unsigned long long) * (8 * 16384 - 1));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ms::g_stamps), sizeof(unsigned long long) * 8 * (size_t)n_blocks);
}
#endif
