// ASan/UBSan run of the tile builder (host code only) on a few meshes incl. rows-per-tile < threads and shards
#include "ms_internal.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace ms;
static void icosphere_like(int n, std::vector<double>& P, std::vector<int32_t>& T) {
  // a lat-long sphere triangulation: (n+1) x (2n) grid vertices collapsed at the poles is overkill; use a grid torus-free patch
  const int nu = n, nv = 2 * n;
  for (int i = 0; i <= nu; ++i)
    for (int j = 0; j < nv; ++j) {
      const double th = M_PI * (i + 0.5) / (nu + 1), ph = 2 * M_PI * j / nv;
      P.push_back(sin(th) * cos(ph)); P.push_back(sin(th) * sin(ph)); P.push_back(cos(th));
    }
  for (int i = 0; i < nu; ++i)
    for (int j = 0; j < nv; ++j) {
      const int a = i * nv + j, b = i * nv + (j + 1) % nv, c = (i + 1) * nv + j, d = (i + 1) * nv + (j + 1) % nv;
      T.push_back(a); T.push_back(c); T.push_back(b);
      T.push_back(b); T.push_back(c); T.push_back(d);
    }
}
int main() {
  for (int n : {3, 17, 60}) {
    std::vector<double> P; std::vector<int32_t> T;
    icosphere_like(n, P, T);
    const int nvx = (int)P.size() / 3, nf = (int)T.size() / 3;
    for (int tile : {0, 64, 128, 129, 200, 255, 256, 512})
      for (int shards : {1, 3}) {
        Tiling t; std::string err;
        int rc = build_tiling(nvx, nf, P.data(), T.data(), nullptr, tile, shards, t, err);
        long own = 0;
        for (size_t p = 0; p < t.tile_facets.size(); ++p) own += (t.tile_facets[p].flags & TF_OWNER) ? 1 : 0;
        printf("n=%d nv=%d nf=%d tile=%d shards=%d rc=%d tiles=%d own=%d T=%d owners=%ld %s\n", n, nvx, nf, tile, shards, rc, t.n_tiles, t.own, t.T, own, err.c_str());
        if (rc == 0 && own != nf) return 1;
      }
  }
  // a bad index and a NaN
  std::vector<double> P; std::vector<int32_t> T;
  icosphere_like(5, P, T);
  T[4] = 100000; T[10] = -3;
  Tiling t; std::string err;
  int rc = build_tiling((int)P.size() / 3, (int)T.size() / 3, P.data(), T.data(), nullptr, 200, 1, t, err);
  printf("bad indices rc=%d dropped=%ld\n", rc, (long)t.dropped_facets);
  return 0;
}
