// Host-side tiling: patch ordering of the vertices (recursive coordinate bisection; Hilbert order inside a tile) and the
// tile -> facet / tile -> halo-vertex CSR that the gfx950 kernels consume.
//
// The reference keeps vertices in Mesh.vertex_ids row order and facets in
// Mesh.triangle_row_cache order (geometry/mesh.py:372-389, :597-624) and walks
// them with np.add.at scatter-adds.  Here each tile OWNS a contiguous range of
// T vertices (in patch order) and lists every facet that touches one of them,
// so a workgroup can accumulate its owned vertices' sums in LDS and write them
// with plain coalesced stores -- no global atomics, no second pass.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>

#include "ms_internal.h"

namespace ms {

namespace {

// Skilling's transpose-based Hilbert index, 3 axes x `bits` bits.
inline uint64_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int bits) {
  uint32_t X[3] = {x, y, z};
  const uint32_t M = 1u << (bits - 1);
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    uint32_t P = Q - 1;
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; ++i) X[i] ^= t;
  uint64_t h = 0;
  for (int b = bits - 1; b >= 0; --b)
    for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> b) & 1u);
  return h;
}

}  // namespace

int build_tiling(int nv, int nf, const double* positions, const int32_t* tri,
                 const uint8_t* body_facets, int T, int shard_count, Tiling& out,
                 std::string& err) {
  if (nv <= 0 || nf < 0 || !positions || (nf > 0 && !tri)) {
    err = "ms_create: nv must be > 0 and positions/tri non-null";
    return MS_ERR_INVALID;
  }
  if (T <= 0) T = 256;
  // rows per tile R <= threads per workgroup T: 129..255 rows run on the 256-thread instances (the tile count, and with
  // it the fill of the last round of workgroups, is the caller's to choose)
  int R = T;
  if (T > 128 && T < 256) T = 256;
  if (!(T == 64 || T == 128 || T == 256 || T == 512)) {
    err = "ms_create: tile_vertices must be 64, 128, 129..256 or 512 (one thread per owned vertex)";
    return MS_ERR_INVALID;
  }
  if (shard_count < 1) shard_count = 1;
  out = Tiling();
  out.nv = nv;
  out.nf = nf;
  out.T = T;
  out.own = R;

  // ---- 1. patch order: sort vertices along a 3D Hilbert curve --------------
  double lo[3] = {positions[0], positions[1], positions[2]};
  double hi[3] = {lo[0], lo[1], lo[2]};
  for (int i = 0; i < nv; ++i)
    for (int d = 0; d < 3; ++d) {
      double p = positions[3 * (size_t)i + d];
      if (!(p == p)) {
        err = "ms_create: NaN in positions";
        return MS_ERR_INVALID;
      }
      lo[d] = std::min(lo[d], p);
      hi[d] = std::max(hi[d], p);
    }
  double span = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
  const int bits = 21;
  const double scale = (double)((1u << bits) - 1) / span;
  std::vector<uint64_t> key(nv);
  for (int i = 0; i < nv; ++i) {
    uint32_t q[3];
    for (int d = 0; d < 3; ++d) {
      double v = (positions[3 * (size_t)i + d] - lo[d]) * scale;
      q[d] = (uint32_t)std::min(std::max(v, 0.0), (double)((1u << bits) - 1));
    }
    key[i] = hilbert3(q[0], q[1], q[2], bits);
  }
  out.perm.resize(nv);
  std::iota(out.perm.begin(), out.perm.end(), 0);
  const int tile_order = variant_env("MS_TILE_ORDER") ? atoi(variant_env("MS_TILE_ORDER")) : 1;  // (variant builds: 0 = Hilbert runs)
  if (tile_order == 0) {
    std::stable_sort(out.perm.begin(), out.perm.end(),
                     [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  } else {
    // Recursive coordinate bisection down to single tiles: a range of whole tiles is split, at a tile boundary
    // nearest its middle, by the median along the longest axis of its bounding box.  The leaves are compact patches
    // (fewer facets shared with other tiles than runs of a 3-D Hilbert curve, which crosses a 2-D surface
    // irregularly), and the left-to-right leaf order keeps neighbouring tiles close in the tile sequence (the XCD
    // and shard ranges are contiguous tile ranges).  Inside a tile the vertices keep their Hilbert order.
    struct Range { int lo, hi; };
    std::vector<Range> stack;
    stack.push_back({0, nv});
    while (!stack.empty()) {
      const Range r = stack.back();
      stack.pop_back();
      const int n = r.hi - r.lo;
      const int tiles = (n + R - 1) / R;
      if (tiles <= 1) {
        std::sort(out.perm.begin() + r.lo, out.perm.begin() + r.hi,
                  [&](int32_t a, int32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
        continue;
      }
      double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
      for (int i = r.lo; i < r.hi; ++i)
        for (int d = 0; d < 3; ++d) {
          const double p = positions[3 * (size_t)out.perm[i] + d];
          blo[d] = std::min(blo[d], p);
          bhi[d] = std::max(bhi[d], p);
        }
      int ax = 0;
      for (int d = 1; d < 3; ++d)
        if (bhi[d] - blo[d] > bhi[ax] - blo[ax]) ax = d;
      const int mid = r.lo + (tiles / 2) * R;
      std::nth_element(out.perm.begin() + r.lo, out.perm.begin() + mid, out.perm.begin() + r.hi,
                       [&](int32_t a, int32_t b) {
                         const double pa = positions[3 * (size_t)a + ax], pb = positions[3 * (size_t)b + ax];
                         return pa < pb || (pa == pb && a < b);
                       });
      stack.push_back({mid, r.hi});
      stack.push_back({r.lo, mid});
    }
  }
  out.iperm.resize(nv);
  for (int i = 0; i < nv; ++i) out.iperm[out.perm[i]] = i;

  // ---- 2. tiles -----------------------------------------------------------
  out.n_tiles = (nv + R - 1) / R;
  out.tiles_per_shard = (out.n_tiles + shard_count - 1) / shard_count;
  out.n_tiles_padded = out.tiles_per_shard * shard_count;
  out.nvp = (int64_t)out.n_tiles_padded * R;

  // count facet instances per tile (a facet is listed by every tile owning one
  // of its corners)
  std::vector<int32_t> cnt(out.n_tiles + 1, 0);
  auto tiles_of = [&](int f, int tl[3], int iv[3]) -> int {
    for (int k = 0; k < 3; ++k) {
      int e = tri[3 * (size_t)f + k];
      if (e < 0 || e >= nv) return -1;
      iv[k] = out.iperm[e];
    }
    int n = 0;
    for (int k = 0; k < 3; ++k) {
      int t = iv[k] / R;
      bool dup = false;
      for (int j = 0; j < n; ++j) dup |= (tl[j] == t);
      if (!dup) tl[n++] = t;
    }
    return n;
  };
  for (int f = 0; f < nf; ++f) {
    int tl[3], iv[3];
    int n = tiles_of(f, tl, iv);
    if (n < 0) {
      ++out.dropped_facets;
      continue;
    }
    for (int j = 0; j < n; ++j) ++cnt[tl[j] + 1];
  }
  out.tile_facet_off.assign(out.n_tiles + 1, 0);
  for (int t = 0; t < out.n_tiles; ++t) out.tile_facet_off[t + 1] = out.tile_facet_off[t] + cnt[t + 1];
  const size_t n_inst = (size_t)out.tile_facet_off[out.n_tiles];
  out.tile_facets.resize(n_inst);
  out.tile_facet_ext.resize(n_inst);
  std::vector<int32_t> cursor(out.tile_facet_off.begin(), out.tile_facet_off.end() - 1);
  // temporary: internal vertex ids per instance
  std::vector<int32_t> inst_v(3 * n_inst);
  // Walk facets in patch order of their first corner so instances inside a
  // tile are spatially coherent.
  std::vector<int32_t> forder(nf);
  std::iota(forder.begin(), forder.end(), 0);
  {
    std::vector<int32_t> fkey(nf);
    for (int f = 0; f < nf; ++f) {
      int e = tri[3 * (size_t)f];
      fkey[f] = (e >= 0 && e < nv) ? out.iperm[e] : 0;
    }
    std::stable_sort(forder.begin(), forder.end(),
                     [&](int32_t a, int32_t b) { return fkey[a] < fkey[b]; });
  }
  for (int fi = 0; fi < nf; ++fi) {
    int f = forder[fi];
    int tl[3], iv[3];
    int n = tiles_of(f, tl, iv);
    if (n < 0) continue;
    int owner_tile = iv[0] / R;
    for (int j = 0; j < n; ++j) {
      size_t p = (size_t)cursor[tl[j]]++;
      inst_v[3 * p] = iv[0];
      inst_v[3 * p + 1] = iv[1];
      inst_v[3 * p + 2] = iv[2];
      uint16_t fl = 0;
      if (tl[j] == owner_tile) fl |= TF_OWNER;
      if (!body_facets || body_facets[f]) fl |= TF_BODY;
      out.tile_facets[p].flags = fl;
      out.tile_facet_ext[p] = f;
    }
  }

  // ---- 3. halo lists + local slots ----------------------------------------
  out.tile_halo_off.assign(out.n_tiles + 1, 0);
  out.tile_ent_off.assign(out.n_tiles + 1, 0);
  out.tile_voff.assign((size_t)out.n_tiles * (T + 1), 0);
  std::vector<int32_t> halo_tmp;
  std::vector<int32_t> vcnt(T + 1);
  // lane order inside a tile (MS_FACET_ORDER): 0 walk order, 1 stride permutation, 2 (default) bank-aware
  const char* env_order = variant_env("MS_FACET_ORDER");  // (variant builds only)
  const int order_mode = env_order ? atoi(env_order) : 2;
  const char* env_stride = getenv("MS_FACET_STRIDE");
  const int stride_mode = env_stride ? atoi(env_stride) : 1;
  std::vector<TileFacet> tf_tmp;
  std::vector<int32_t> ext_tmp;
  std::vector<int32_t> order;
  std::vector<uint8_t> taken;
  for (int t = 0; t < out.n_tiles; ++t) {
    const int v_lo = t * R;
    const int v_hi = std::min(nv, v_lo + R);
    const size_t b = (size_t)out.tile_facet_off[t], e = (size_t)out.tile_facet_off[t + 1];
    const size_t n = e - b;
    halo_tmp.clear();
    for (size_t p = b; p < e; ++p)
      for (int k = 0; k < 3; ++k) {
        int v = inst_v[3 * p + k];
        if (v < v_lo || v >= v_hi) halo_tmp.push_back(v);
      }
    std::sort(halo_tmp.begin(), halo_tmp.end());
    halo_tmp.erase(std::unique(halo_tmp.begin(), halo_tmp.end()), halo_tmp.end());
    const int n_owned = v_hi - v_lo;
    if ((size_t)n_owned + halo_tmp.size() > 65535u) {
      err = "ms_create: a vertex patch needs more than 65535 LDS slots (vertex valence too high)";
      return MS_ERR_TILE_CAPACITY;
    }
    for (size_t p = b; p < e; ++p) {
      uint16_t loc[3];
      for (int k = 0; k < 3; ++k) {
        int v = inst_v[3 * p + k];
        if (v >= v_lo && v < v_hi) {
          loc[k] = (uint16_t)(v - v_lo);
        } else {
          auto it = std::lower_bound(halo_tmp.begin(), halo_tmp.end(), v);
          loc[k] = (uint16_t)(n_owned + (int)(it - halo_tmp.begin()));
        }
      }
      out.tile_facets[p].l0 = loc[0];
      out.tile_facets[p].l1 = loc[1];
      out.tile_facets[p].l2 = loc[2];
    }

    // ---- 3b. which lane computes which facet ----------------------------------
    // In walk order consecutive facets share their first corner (and mostly a second one), so the 64 lanes of a
    // wave would send their per-corner LDS gathers and ds_add_f64 atomics to a handful of vertex slots.  The LDS
    // serves an 8-byte access of 32 lanes per cycle when the lanes' slots differ mod 32 (reads: 64 banks of 4
    // bytes, lane groups of 32) and writes / atomics in lane groups of 16 over 32 banks (slots distinct mod 16).
    // Bank-aware order: fill every group of 32 lanes greedily with facets whose three corner slots are free in
    // the group's three residue sets (mod 32 over the group, mod 16 over each half); what does not fit anywhere
    // goes to the end.  Same-vertex corners (same address) are the special case "same residue".
    if (n >= 128 && order_mode != 0) {
      tf_tmp.assign(out.tile_facets.begin() + b, out.tile_facets.begin() + e);
      ext_tmp.assign(out.tile_facet_ext.begin() + b, out.tile_facet_ext.begin() + e);
      order.clear();
      order.reserve(n);
      // start from the stride permutation (spreads neighbours over the waves)
      size_t S = std::max<size_t>(stride_mode > 1 ? (size_t)stride_mode : 7, (n + 63) / 64);
      while (std::gcd(S, n) != 1) ++S;
      std::vector<int32_t>& cand = order;  // candidates in stride order
      for (size_t i = 0; i < n; ++i) cand.push_back((int32_t)((i * S) % n));
      if (order_mode >= 2) {
        std::vector<int32_t> seq(cand);
        taken.assign(n, 0);
        order.clear();
        size_t first_free = 0;
        while (order.size() < n) {
          uint32_t m32[3] = {0, 0, 0};
          const size_t group_end = std::min(n, order.size() + 32);
          for (int half = 0; half < 2 && order.size() < group_end; ++half) {
            uint32_t m16[3] = {0, 0, 0};
            const size_t half_end = std::min(group_end, order.size() + 16);
            while (order.size() < half_end) {
              // the first candidate that is free in all six residue sets, else the one with the fewest collisions
              // among the next free candidates
              size_t best = n;
              int best_cost = 99, seen = 0;
              for (size_t q = first_free; q < n && seen < 192; ++q) {
                if (taken[q]) continue;
                ++seen;
                const TileFacet& f = tf_tmp[seq[q]];
                const uint32_t r[3] = {f.l0, f.l1, f.l2};
                int cost = 0;
                for (int k = 0; k < 3; ++k)
                  cost += (int)((m32[k] >> (r[k] & 31)) & 1u) + (int)((m16[k] >> (r[k] & 15)) & 1u);
                if (cost < best_cost) {
                  best_cost = cost;
                  best = q;
                  if (cost == 0) break;
                }
              }
              if (best == n) break;
              const TileFacet& f = tf_tmp[seq[best]];
              const uint32_t r[3] = {f.l0, f.l1, f.l2};
              for (int k = 0; k < 3; ++k) {
                m32[k] |= 1u << (r[k] & 31);
                m16[k] |= 1u << (r[k] & 15);
              }
              taken[best] = 1;
              order.push_back(seq[best]);
              while (first_free < n && taken[first_free]) ++first_free;
            }
            while (first_free < n && taken[first_free]) ++first_free;
          }
        }
      }
      for (size_t i = 0; i < n; ++i) {
        out.tile_facets[b + i] = tf_tmp[order[i]];
        out.tile_facet_ext[b + i] = ext_tmp[order[i]];
      }
    }

    // vertex -> corner CSR of this tile (counting sort keeps facet_local ascending)
    {
      if (e - b > 16383u) {
        err = "ms_create: a tile lists more than 16383 facets (vertex valence too high)";
        return MS_ERR_TILE_CAPACITY;
      }
      std::fill(vcnt.begin(), vcnt.end(), 0);
      for (size_t p = b; p < e; ++p) {
        const TileFacet& f = out.tile_facets[p];
        if (f.l0 < n_owned) ++vcnt[f.l0 + 1];
        if (f.l1 < n_owned) ++vcnt[f.l1 + 1];
        if (f.l2 < n_owned) ++vcnt[f.l2 + 1];
      }
      for (int i = 0; i < T; ++i) vcnt[i + 1] += vcnt[i];
      const int n_ent = vcnt[T];
      if (n_ent > 65535) {
        err = "ms_create: a tile has more than 65535 owned corners";
        return MS_ERR_TILE_CAPACITY;
      }
      uint16_t* voff = &out.tile_voff[(size_t)t * (T + 1)];
      for (int i = 0; i <= T; ++i) voff[i] = (uint16_t)vcnt[i];
      const size_t base = out.vent.size();
      out.vent.resize(base + n_ent);
      std::vector<int32_t> fill(vcnt.begin(), vcnt.end() - 1);
      for (size_t p = b; p < e; ++p) {
        const TileFacet& f = out.tile_facets[p];
        const uint16_t fl = (uint16_t)(p - b);
        if (f.l0 < n_owned) out.vent[base + fill[f.l0]++] = (uint16_t)(fl << 2 | 0);
        if (f.l1 < n_owned) out.vent[base + fill[f.l1]++] = (uint16_t)(fl << 2 | 1);
        if (f.l2 < n_owned) out.vent[base + fill[f.l2]++] = (uint16_t)(fl << 2 | 2);
      }
      out.tile_ent_off[t + 1] = (int32_t)out.vent.size();
      out.max_ent = std::max(out.max_ent, n_ent);
    }
    out.halo_ids.insert(out.halo_ids.end(), halo_tmp.begin(), halo_tmp.end());
    out.tile_halo_off[t + 1] = (int32_t)out.halo_ids.size();
    out.max_halo = std::max(out.max_halo, (int)halo_tmp.size());
    out.max_tile_facets = std::max(out.max_tile_facets, (int)(e - b));
  }
  return MS_OK;
}

}  // namespace ms
