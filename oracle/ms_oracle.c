/*
 * ms_oracle.c -- CPU restatement of membrane_solver's energy+gradient hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load it.
 * The product path (membrane_solver_amd/) never calls into this file.
 *
 * Every function restates, loop for loop, the reference routine it cites
 * (paths relative to the reference checkout).  Array layout: row-major
 * (n,3) doubles / int32, which is byte-identical to the (3,n) Fortran-order
 * arrays the reference's f2py kernels receive.  Indices are zero based.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function here against
 * tests/golden/ *.npz, which oracle/gen_golden.py produced by importing the
 * reference itself (Fortran kernels enabled) in the build container.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (see oracle/Makefile); strict
 * IEEE fp64, no fast-math, no FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Third build, libms_oracle_ld.so (-DORC_LD): every `double` of this file -- arguments, locals, accumulators, scratch
 * arrays -- is the x87 80-bit long double (64-bit mantissa: 2^-11 of fp64's rounding unit), sqrt is sqrtl.  The literal
 * constants are exact in both types (0.5, 0.25, 2, 6, 8, 1e-12 thresholds as their fp64 values).  Callers pass
 * numpy.longdouble arrays (oracle/truth.py).  It serves ONE purpose: a reference gradient against which the fp64
 * oracle's and the HIP path's rounding errors at full size can both be measured (tests/test_gpu_minimizer.py). */
#ifdef ORC_LD
#define double long double
#define sqrt sqrtl
#endif

/* The same source builds twice: libms_oracle.so (serial, strict order of operations: the
 * CHECKER) and libms_oracle_omp.so (-fopenmp -DORC_OMP: facet loops split over the host
 * cores, vertex scatter-adds by `omp atomic`; sums arrive in a different order, so this
 * build is only ever TIMED -- bench.py's all-cores cpu_baseline leg). */
#ifdef ORC_OMP
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_FACETS ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_FACETS_SUM(v) ORC_PRAGMA(omp parallel for schedule(static) reduction(+ : v))
#define ORC_AT ORC_PRAGMA(omp atomic)
#else
#define ORC_FACETS
#define ORC_FACETS_SUM(v)
#define ORC_AT
#endif

static inline void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }
static inline void sub3(const double *a, const double *b, double *c) {
  c[0] = a[0] - b[0];
  c[1] = a[1] - b[1];
  c[2] = a[2] - b[2];
}
static inline int in_range(int i, int nv) { return i >= 0 && i < nv; }

/* ------------------------------------------------------------------------
 * fortran_kernels/surface_energy.f90:27-99  surface_energy_and_gradient
 * (NumPy twin: modules/energy/surface.py:181-221).  grad is accumulated
 * into (intent inout); facets with out-of-range indices or A2 < 1e-12 are
 * skipped; E is the plain sequential sum.  grad may be NULL (energy only).
 * ---------------------------------------------------------------------- */
ORC_API void orc_surface_energy_and_gradient(int nv, int nf, const double *pos,
                                             const int32_t *tri,
                                             const double *gamma, double *grad,
                                             double *E_out) {
  const double eps = 1.0e-12;
  double E = 0.0;
  ORC_FACETS_SUM(E)
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    if (!in_range(i0, nv) || !in_range(i1, nv) || !in_range(i2, nv)) continue;
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double e1[3], e2[3], nvec[3], nhat[3];
    sub3(v1, v0, e1);
    sub3(v2, v0, e2);
    cross3(e1, e2, nvec);
    double A2 = norm3(nvec);
    if (A2 < eps) continue;
    for (int d = 0; d < 3; ++d) nhat[d] = nvec[d] / A2;
    double area = 0.5 * A2;
    E = E + gamma[f] * area;
    if (!grad) continue;
    double a[3], g0[3], g1[3], g2[3];
    sub3(v1, v2, a);
    cross3(a, nhat, g0);
    sub3(v2, v0, a);
    cross3(a, nhat, g1);
    sub3(v0, v1, a);
    cross3(a, nhat, g2);
    for (int d = 0; d < 3; ++d) {
      g0[d] = gamma[f] * (0.5 * g0[d]);
      g1[d] = gamma[f] * (0.5 * g1[d]);
      g2[d] = gamma[f] * (0.5 * g2[d]);
      ORC_AT
      grad[3 * i0 + d] += g0[d];
      ORC_AT
      grad[3 * i1 + d] += g1[d];
      ORC_AT
      grad[3 * i2 + d] += g2[d];
    }
  }
  *E_out = E;
}

/* ------------------------------------------------------------------------
 * fortran_kernels/bending_kernels.f90:32-74  grad_cotan_batch
 * (NumPy twin: geometry/bending_derivatives.py:48-79).  Outputs are zeroed
 * first; pairs with |u x v| <= 1e-15 stay zero.
 * ---------------------------------------------------------------------- */
static inline void grad_cotan_one(const double *u, const double *v, double *gu,
                                  double *gv) {
  double w[3], vxw[3], wxu[3];
  double C = dot3(u, v);
  cross3(u, v, w);
  double S = norm3(w);
  if (S <= 1.0e-15) {
    gu[0] = gu[1] = gu[2] = 0.0;
    gv[0] = gv[1] = gv[2] = 0.0;
    return;
  }
  double invS = 1.0 / S;
  double invS3 = 1.0 / (S * S * S);
  cross3(v, w, vxw);
  cross3(w, u, wxu);
  for (int d = 0; d < 3; ++d) {
    gu[d] = v[d] * invS - (C * invS3) * vxw[d];
    gv[d] = u[d] * invS - (C * invS3) * wxu[d];
  }
}

ORC_API void orc_grad_cotan_batch(int n, const double *u, const double *v,
                                  double *grad_u, double *grad_v) {
  for (int i = 0; i < n; ++i)
    grad_cotan_one(u + 3 * i, v + 3 * i, grad_u + 3 * i, grad_v + 3 * i);
}

/* geometry/bending_derivatives.py:82-102  grad_triangle_area */
static inline void grad_triangle_area_one(const double *u, const double *v,
                                          double *gu, double *gv) {
  double w[3], vxw[3], wxu[3];
  cross3(u, v, w);
  double S = norm3(w);
  if (!(S > 1.0e-15)) {
    gu[0] = gu[1] = gu[2] = 0.0;
    gv[0] = gv[1] = gv[2] = 0.0;
    return;
  }
  double invS = 1.0 / S;
  cross3(v, w, vxw);
  cross3(w, u, wxu);
  for (int d = 0; d < 3; ++d) {
    gu[d] = 0.5 * vxw[d] * invS;
    gv[d] = 0.5 * wxu[d] * invS;
  }
}

/* ------------------------------------------------------------------------
 * fortran_kernels/bending_kernels.f90:87-131  apply_beltrami_laplacian
 * (NumPy twin: modules/energy/bending_math.py:112-118).  `out` is zeroed
 * first.  field/out are (nv,dim) row-major.
 * ---------------------------------------------------------------------- */
ORC_API void orc_apply_beltrami_laplacian(int dim, int nv, int nf,
                                          const double *weights,
                                          const int32_t *tri,
                                          const double *field, double *out) {
  memset(out, 0, sizeof(double) * (size_t)nv * (size_t)dim);
  ORC_FACETS
  for (int f = 0; f < nf; ++f) {
    double c0 = weights[3 * f], c1 = weights[3 * f + 1], c2 = weights[3 * f + 2];
    int v0 = tri[3 * f], v1 = tri[3 * f + 1], v2 = tri[3 * f + 2];
    if (!in_range(v0, nv) || !in_range(v1, nv) || !in_range(v2, nv)) continue;
    for (int d = 0; d < dim; ++d) {
      double f0 = field[(size_t)v0 * dim + d];
      double f1 = field[(size_t)v1 * dim + d];
      double f2 = field[(size_t)v2 * dim + d];
      ORC_AT
      out[(size_t)v0 * dim + d] += 0.5 * (c1 * (f0 - f2) + c2 * (f0 - f1));
      ORC_AT
      out[(size_t)v1 * dim + d] += 0.5 * (c2 * (f1 - f0) + c0 * (f1 - f2));
      ORC_AT
      out[(size_t)v2 * dim + d] += 0.5 * (c0 * (f2 - f1) + c1 * (f2 - f0));
    }
  }
}

/* ------------------------------------------------------------------------
 * fortran_kernels/tilt_kernels.f90:26-86  p1_triangle_divergence
 * (NumPy twin: geometry/tilt_operators.py:301-330, triangle_ops.py:75-95).
 * ---------------------------------------------------------------------- */
ORC_API void orc_p1_triangle_divergence(int nv, int nf, const double *pos,
                                        const double *tilts, const int32_t *tri,
                                        double *div_tri, double *area,
                                        double *g0, double *g1, double *g2) {
  const double eps = 1.0e-20;
  memset(div_tri, 0, sizeof(double) * (size_t)nf);
  memset(area, 0, sizeof(double) * (size_t)nf);
  memset(g0, 0, sizeof(double) * 3 * (size_t)nf);
  memset(g1, 0, sizeof(double) * 3 * (size_t)nf);
  memset(g2, 0, sizeof(double) * 3 * (size_t)nf);
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    if (!in_range(i0, nv) || !in_range(i1, nv) || !in_range(i2, nv)) continue;
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double a[3], b[3], n[3], e0[3], e1[3], e2[3], c[3];
    sub3(v1, v0, a);
    sub3(v2, v0, b);
    cross3(a, b, n);
    double n2 = dot3(n, n);
    double denom = n2 > eps ? n2 : eps;
    sub3(v2, v1, e0);
    sub3(v0, v2, e1);
    sub3(v1, v0, e2);
    cross3(n, e0, c);
    for (int d = 0; d < 3; ++d) g0[3 * f + d] = c[d] / denom;
    cross3(n, e1, c);
    for (int d = 0; d < 3; ++d) g1[3 * f + d] = c[d] / denom;
    cross3(n, e2, c);
    for (int d = 0; d < 3; ++d) g2[3 * f + d] = c[d] / denom;
    div_tri[f] = dot3(tilts + 3 * i0, g0 + 3 * f) +
                 dot3(tilts + 3 * i1, g1 + 3 * f) +
                 dot3(tilts + 3 * i2, g2 + 3 * f);
    area[f] = 0.5 * sqrt(n2 > 0.0 ? n2 : 0.0);
  }
}

/* mixed-Voronoi corner areas with the reference's sequential-overwrite
 * obtuse logic (tilt_kernels.f90:160-181 == curvature.py:299-314). */
static inline void corner_areas(double c0, double c1, double c2, double l0,
                                double l1, double l2, double tri_area,
                                double *va) {
  int o0 = c0 < 0.0, o1 = c1 < 0.0, o2 = c2 < 0.0;
  if (!(o0 || o1 || o2)) {
    va[0] = (l1 * c1 + l2 * c2) / 8.0;
    va[1] = (l2 * c2 + l0 * c0) / 8.0;
    va[2] = (l0 * c0 + l1 * c1) / 8.0;
  } else {
    va[0] = va[1] = va[2] = 0.0;
    if (o0) va[0] = tri_area / 2.0;
    if (o1 || o2) va[0] = tri_area / 4.0;
    if (o1) va[1] = tri_area / 2.0;
    if (o0 || o2) va[1] = tri_area / 4.0;
    if (o2) va[2] = tri_area / 2.0;
    if (o0 || o1) va[2] = tri_area / 4.0;
  }
}

/* ------------------------------------------------------------------------
 * fortran_kernels/tilt_kernels.f90:88-190  compute_curvature_data
 * (NumPy twin: geometry/curvature.py:254-332).  Outputs zeroed first.
 * va0/va1/va2 are the optional per-corner outputs (may be NULL).
 * ---------------------------------------------------------------------- */
ORC_API void orc_compute_curvature_data(int nv, int nf, const double *pos,
                                        const int32_t *tri, double *k_vecs,
                                        double *vertex_areas, double *weights,
                                        double *va0_out, double *va1_out,
                                        double *va2_out) {
  const double area_eps = 1.0e-12;
  memset(k_vecs, 0, sizeof(double) * 3 * (size_t)nv);
  memset(vertex_areas, 0, sizeof(double) * (size_t)nv);
  memset(weights, 0, sizeof(double) * 3 * (size_t)nf);
  if (va0_out) memset(va0_out, 0, sizeof(double) * (size_t)nf);
  if (va1_out) memset(va1_out, 0, sizeof(double) * (size_t)nf);
  if (va2_out) memset(va2_out, 0, sizeof(double) * (size_t)nf);
  ORC_FACETS
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    if (!in_range(i0, nv) || !in_range(i1, nv) || !in_range(i2, nv)) continue;
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double e0[3], e1[3], e2[3], cr[3];
    sub3(v2, v1, e0);
    sub3(v0, v2, e1);
    sub3(v1, v0, e2);
    double l0 = dot3(e0, e0), l1 = dot3(e1, e1), l2 = dot3(e2, e2);
    cross3(e1, e2, cr);
    double area_doubled = norm3(cr);
    if (area_doubled < area_eps) area_doubled = area_eps;
    double tri_area = 0.5 * area_doubled;
    double ne0[3] = {-e0[0], -e0[1], -e0[2]};
    double ne1[3] = {-e1[0], -e1[1], -e1[2]};
    double ne2[3] = {-e2[0], -e2[1], -e2[2]};
    double c0 = dot3(ne1, e2) / area_doubled;
    double c1 = dot3(ne2, e0) / area_doubled;
    double c2 = dot3(ne0, e1) / area_doubled;
    weights[3 * f] = c0;
    weights[3 * f + 1] = c1;
    weights[3 * f + 2] = c2;
    for (int d = 0; d < 3; ++d) {
      ORC_AT
      k_vecs[3 * i0 + d] += 0.5 * (c1 * ne1[d] + c2 * e2[d]);
      ORC_AT
      k_vecs[3 * i1 + d] += 0.5 * (c2 * ne2[d] + c0 * e0[d]);
      ORC_AT
      k_vecs[3 * i2 + d] += 0.5 * (c0 * ne0[d] + c1 * e1[d]);
    }
    double va[3];
    corner_areas(c0, c1, c2, l0, l1, l2, tri_area, va);
    ORC_AT
    vertex_areas[i0] += va[0];
    ORC_AT
    vertex_areas[i1] += va[1];
    ORC_AT
    vertex_areas[i2] += va[2];
    if (va0_out) va0_out[f] = va[0];
    if (va1_out) va1_out[f] = va[1];
    if (va2_out) va2_out[f] = va[2];
  }
}

/* ------------------------------------------------------------------------
 * modules/energy/bending_utils.py:37-171  _compute_effective_areas
 * Corner areas recomputed from positions + given cotans (tri_area from
 * cross(v1-v0, v2-v0), clamped at 1e-12), then for triangles with >=1
 * interior and >=1 boundary vertex the boundary corners' area is moved
 * equally onto the interior corners (:121-153).  va_eff is (nf,3);
 * vertex_areas_eff (nv) may be NULL.
 * ---------------------------------------------------------------------- */
ORC_API void orc_effective_areas(int nv, int nf, const double *pos,
                                 const int32_t *tri, const double *weights,
                                 const uint8_t *is_boundary,
                                 double *vertex_areas_eff, double *va_eff) {
  if (vertex_areas_eff) memset(vertex_areas_eff, 0, sizeof(double) * (size_t)nv);
  ORC_FACETS
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double e0[3], e1[3], e2[3], a[3], b[3], n[3];
    sub3(v2, v1, e0);
    sub3(v0, v2, e1);
    sub3(v1, v0, e2);
    double l0 = dot3(e0, e0), l1 = dot3(e1, e1), l2 = dot3(e2, e2);
    double c0 = weights[3 * f], c1 = weights[3 * f + 1], c2 = weights[3 * f + 2];
    sub3(v1, v0, a);
    sub3(v2, v0, b);
    cross3(a, b, n);
    double tri_area = 0.5 * norm3(n);
    if (tri_area < 1.0e-12) tri_area = 1.0e-12;
    double va[3];
    corner_areas(c0, c1, c2, l0, l1, l2, tri_area, va);
    int isb[3] = {is_boundary ? is_boundary[i0] != 0 : 0,
                  is_boundary ? is_boundary[i1] != 0 : 0,
                  is_boundary ? is_boundary[i2] != 0 : 0};
    int n_int = (!isb[0]) + (!isb[1]) + (!isb[2]);
    int some_b = isb[0] || isb[1] || isb[2];
    if (n_int > 0 && some_b) {
      double b_sum = va[0] * isb[0] + va[1] * isb[1] + va[2] * isb[2];
      double extra = b_sum / (double)n_int;
      for (int k = 0; k < 3; ++k) {
        double m = isb[k] ? 0.0 : 1.0;
        va[k] = va[k] * m + m * extra;
      }
    }
    va_eff[3 * f] = va[0];
    va_eff[3 * f + 1] = va[1];
    va_eff[3 * f + 2] = va[2];
    if (vertex_areas_eff) {
      ORC_AT
      vertex_areas_eff[i0] += va[0];
      ORC_AT
      vertex_areas_eff[i1] += va[1];
      ORC_AT
      vertex_areas_eff[i2] += va[2];
    }
  }
}

/* modules/energy/bending_utils.py:13-34  _vertex_normals */
ORC_API void orc_vertex_normals(int nv, int nf, const double *pos,
                                const int32_t *tri, double *normals) {
  memset(normals, 0, sizeof(double) * 3 * (size_t)nv);
  ORC_FACETS
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double a[3], b[3], n[3];
    sub3(v1, v0, a);
    sub3(v2, v0, b);
    cross3(a, b, n);
    for (int d = 0; d < 3; ++d) {
      ORC_AT
      normals[3 * i0 + d] += n[d];
      ORC_AT
      normals[3 * i1 + d] += n[d];
      ORC_AT
      normals[3 * i2 + d] += n[d];
    }
  }
  for (int i = 0; i < nv; ++i) {
    double nrm = norm3(normals + 3 * i);
    if (nrm > 1.0e-15)
      for (int d = 0; d < 3; ++d) normals[3 * i + d] /= nrm;
  }
}

/* ------------------------------------------------------------------------
 * modules/energy/bending_gradient.py:17-175
 * _backpropagate_bending_shape_gradient.  grad_arr += grad_linear +
 * grad_cot + grad_area, the three kept in separate accumulators as in the
 * reference.  Returns 0, or -1 on allocation failure.
 * ---------------------------------------------------------------------- */
static int backprop_bending(int nv, int nf, const double *pos,
                            const int32_t *tri, const double *weights,
                            const uint8_t *is_interior, const double *fA_eff,
                            const double *fA_vor, const double *fK,
                            double *grad_arr) {
  size_t n3 = 3 * (size_t)nv;
  double *grad_linear = (double *)malloc(sizeof(double) * n3);
  double *grad_cot = (double *)calloc(n3, sizeof(double));
  double *grad_area = (double *)calloc(n3, sizeof(double));
  if (!grad_linear || !grad_cot || !grad_area) {
    free(grad_linear);
    free(grad_cot);
    free(grad_area);
    return -1;
  }
  /* Term 1 (:34): grad_linear = -L(weights) fK */
  orc_apply_beltrami_laplacian(3, nv, nf, weights, tri, fK, grad_linear);
  for (size_t i = 0; i < n3; ++i) grad_linear[i] = -grad_linear[i];

  ORC_FACETS
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double e0[3], e1[3], e2[3];
    sub3(v2, v1, e0);
    sub3(v0, v2, e1);
    sub3(v1, v0, e2);
    double c0 = weights[3 * f], c1 = weights[3 * f + 1], c2 = weights[3 * f + 2];
    const double *fK0 = fK + 3 * i0, *fK1 = fK + 3 * i1, *fK2 = fK + 3 * i2;

    /* Term 2 (:37-78): variation of the cotangents */
    double a[3], b[3];
    sub3(fK1, fK2, a);
    sub3(v1, v2, b);
    double dE_dc0 = -0.5 * dot3(a, b);
    sub3(fK2, fK0, a);
    sub3(v2, v0, b);
    double dE_dc1 = -0.5 * dot3(a, b);
    sub3(fK0, fK1, a);
    sub3(v0, v1, b);
    double dE_dc2 = -0.5 * dot3(a, b);

    double u[3], w[3], gu[3], gv[3];
    /* corner 0: u=v1-v0, v=v2-v0 -> +gu to v1, +gv to v2, -(gu+gv) to v0 */
    sub3(v1, v0, u);
    sub3(v2, v0, w);
    grad_cotan_one(u, w, gu, gv);
    for (int d = 0; d < 3; ++d) {
      ORC_AT
      grad_cot[3 * i1 + d] += dE_dc0 * gu[d];
      ORC_AT
      grad_cot[3 * i2 + d] += dE_dc0 * gv[d];
      ORC_AT
      grad_cot[3 * i0 + d] += dE_dc0 * -(gu[d] + gv[d]);
    }
    /* corner 1: u=v2-v1, v=v0-v1 -> v2, v0, v1 */
    sub3(v2, v1, u);
    sub3(v0, v1, w);
    grad_cotan_one(u, w, gu, gv);
    for (int d = 0; d < 3; ++d) {
      ORC_AT
      grad_cot[3 * i2 + d] += dE_dc1 * gu[d];
      ORC_AT
      grad_cot[3 * i0 + d] += dE_dc1 * gv[d];
      ORC_AT
      grad_cot[3 * i1 + d] += dE_dc1 * -(gu[d] + gv[d]);
    }
    /* corner 2: u=v0-v2, v=v1-v2 -> v0, v1, v2 */
    sub3(v0, v2, u);
    sub3(v1, v2, w);
    grad_cotan_one(u, w, gu, gv);
    for (int d = 0; d < 3; ++d) {
      ORC_AT
      grad_cot[3 * i0 + d] += dE_dc2 * gu[d];
      ORC_AT
      grad_cot[3 * i1 + d] += dE_dc2 * gv[d];
      ORC_AT
      grad_cot[3 * i2 + d] += dE_dc2 * -(gu[d] + gv[d]);
    }

    /* Term 3 (:80-173): area variation.  C = C_eff + fA_vor (:81-95) */
    int ti[3] = {is_interior[i0] != 0, is_interior[i1] != 0, is_interior[i2] != 0};
    int counts = ti[0] + ti[1] + ti[2];
    double tfa[3] = {fA_eff[i0], fA_eff[i1], fA_eff[i2]};
    double sum_int = tfa[0] * ti[0] + tfa[1] * ti[1] + tfa[2] * ti[2];
    double avg = counts > 0 ? sum_int / (double)counts : 0.0;
    double C[3];
    C[0] = (ti[0] ? tfa[0] : avg) + fA_vor[i0];
    C[1] = (ti[1] ? tfa[1] : avg) + fA_vor[i1];
    C[2] = (ti[2] ? tfa[2] : avg) + fA_vor[i2];

    int obtuse = (c0 < 0.0) || (c1 < 0.0) || (c2 < 0.0);
    if (!obtuse) {
      double coeff;
      coeff = 0.25 * c1 * C[0];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i0 + d] += coeff * e1[d];
        ORC_AT
        grad_area[3 * i2 + d] += -coeff * e1[d];
      }
      coeff = 0.25 * c2 * C[0];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i1 + d] += coeff * e2[d];
        ORC_AT
        grad_area[3 * i0 + d] += -coeff * e2[d];
      }
      coeff = 0.25 * c2 * C[1];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i1 + d] += coeff * e2[d];
        ORC_AT
        grad_area[3 * i0 + d] += -coeff * e2[d];
      }
      coeff = 0.25 * c0 * C[1];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i2 + d] += coeff * e0[d];
        ORC_AT
        grad_area[3 * i1 + d] += -coeff * e0[d];
      }
      coeff = 0.25 * c0 * C[2];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i2 + d] += coeff * e0[d];
        ORC_AT
        grad_area[3 * i1 + d] += -coeff * e0[d];
      }
      coeff = 0.25 * c1 * C[2];
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i0 + d] += coeff * e1[d];
        ORC_AT
        grad_area[3 * i2 + d] += -coeff * e1[d];
      }
      double l0sq = dot3(e0, e0), l1sq = dot3(e1, e1), l2sq = dot3(e2, e2);
      double cc0 = 0.125 * l0sq * (C[1] + C[2]);
      double cc1 = 0.125 * l1sq * (C[0] + C[2]);
      double cc2 = 0.125 * l2sq * (C[0] + C[1]);
      double ne[3];
      /* gc0 = grad_cotan(e2, -e1) */
      for (int d = 0; d < 3; ++d) ne[d] = -e1[d];
      grad_cotan_one(e2, ne, gu, gv);
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i1 + d] += cc0 * gu[d];
        ORC_AT
        grad_area[3 * i2 + d] += cc0 * gv[d];
        ORC_AT
        grad_area[3 * i0 + d] += cc0 * -(gu[d] + gv[d]);
      }
      /* gc1 = grad_cotan(e0, -e2) */
      for (int d = 0; d < 3; ++d) ne[d] = -e2[d];
      grad_cotan_one(e0, ne, gu, gv);
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i2 + d] += cc1 * gu[d];
        ORC_AT
        grad_area[3 * i0 + d] += cc1 * gv[d];
        ORC_AT
        grad_area[3 * i1 + d] += cc1 * -(gu[d] + gv[d]);
      }
      /* gc2 = grad_cotan(e1, -e0) */
      for (int d = 0; d < 3; ++d) ne[d] = -e0[d];
      grad_cotan_one(e1, ne, gu, gv);
      for (int d = 0; d < 3; ++d) {
        ORC_AT
        grad_area[3 * i0 + d] += cc2 * gu[d];
        ORC_AT
        grad_area[3 * i1 + d] += cc2 * gv[d];
        ORC_AT
        grad_area[3 * i2 + d] += cc2 * -(gu[d] + gv[d]);
      }
    } else {
      const double cs[3] = {c0, c1, c2};
      for (int i = 0; i < 3; ++i) {
        if (!(cs[i] < 0.0)) continue;
        sub3(v1, v0, u);
        sub3(v2, v0, w);
        grad_triangle_area_one(u, w, gu, gv);
        double factor;
        if (i == 0)
          factor = 0.5 * C[0] + 0.25 * C[1] + 0.25 * C[2];
        else if (i == 1)
          factor = 0.5 * C[1] + 0.25 * C[0] + 0.25 * C[2];
        else
          factor = 0.5 * C[2] + 0.25 * C[0] + 0.25 * C[1];
        for (int d = 0; d < 3; ++d) {
          ORC_AT
          grad_area[3 * i1 + d] += factor * gu[d];
          ORC_AT
          grad_area[3 * i2 + d] += factor * gv[d];
          ORC_AT
          grad_area[3 * i0 + d] += factor * -(gu[d] + gv[d]);
        }
      }
    }
  }
  for (size_t i = 0; i < n3; ++i)
    grad_arr[i] += (grad_linear[i] + grad_cot[i]) + grad_area[i];
  free(grad_linear);
  free(grad_cot);
  free(grad_area);
  return 0;
}

/* bending_gradient.py:17-175 with caller-supplied per-vertex factors (used by the
 * sharded-driver tests to prove the factors really were exchanged).  weights are
 * recomputed from pos as compute_curvature_data does. */
ORC_API int orc_bending_backprop(int nv, int nf, const double *pos, const int32_t *tri,
                                 const uint8_t *is_boundary, const double *fA_eff,
                                 const double *fA_vor, const double *fK, double *grad) {
  double *k_vecs = (double *)malloc(sizeof(double) * 3 * (size_t)nv);
  double *A_vor = (double *)malloc(sizeof(double) * (size_t)nv);
  double *weights = (double *)malloc(sizeof(double) * 3 * (size_t)nf);
  uint8_t *is_int = (uint8_t *)malloc((size_t)nv);
  int rc = -1;
  if (k_vecs && A_vor && weights && is_int) {
    orc_compute_curvature_data(nv, nf, pos, tri, k_vecs, A_vor, weights, NULL, NULL, NULL);
    for (int i = 0; i < nv; ++i) is_int[i] = is_boundary ? (is_boundary[i] == 0) : 1;
    rc = backprop_bending(nv, nf, pos, tri, weights, is_int, fA_eff, fA_vor, fK, grad);
  }
  free(k_vecs);
  free(A_vor);
  free(weights);
  free(is_int);
  return rc;
}

/* ------------------------------------------------------------------------
 * modules/energy/bending.py:90-181  compute_energy_and_gradient_array
 * model: 0 = helfrich, 1 = willmore.  mode: 0 = analytic, 1 = approx.
 * grad is accumulated into.  Optional debug outputs (may be NULL):
 * fK_out (nv,3), fA_eff_out (nv), fA_vor_out (nv).
 * ---------------------------------------------------------------------- */
ORC_API int orc_bending_energy_and_gradient(
    int nv, int nf, const double *pos, const int32_t *tri, const double *kappa,
    const double *c0_arr, const uint8_t *is_boundary, int model, int mode,
    double *grad, double *E_out, double *fK_out, double *fA_eff_out,
    double *fA_vor_out) {
  *E_out = 0.0;
  if (nf == 0) return 0;
  size_t nvs = (size_t)nv, nfs = (size_t)nf;
  double *k_vecs = (double *)malloc(sizeof(double) * 3 * nvs);
  double *A_vor = (double *)malloc(sizeof(double) * nvs);
  double *weights = (double *)malloc(sizeof(double) * 3 * nfs);
  double *A_eff = (double *)malloc(sizeof(double) * nvs);
  double *va_eff = (double *)malloc(sizeof(double) * 3 * nfs);
  double *fK = (double *)malloc(sizeof(double) * 3 * nvs);
  double *fA_eff = (double *)malloc(sizeof(double) * nvs);
  double *fA_vor = (double *)malloc(sizeof(double) * nvs);
  double *normals = (double *)malloc(sizeof(double) * 3 * nvs);
  uint8_t *is_int = (uint8_t *)malloc(nvs);
  int rc = 0;
  if (!k_vecs || !A_vor || !weights || !A_eff || !va_eff || !fK || !fA_eff ||
      !fA_vor || !normals || !is_int) {
    rc = -1;
    goto done;
  }
  orc_compute_curvature_data(nv, nf, pos, tri, k_vecs, A_vor, weights, NULL,
                             NULL, NULL);
  orc_effective_areas(nv, nf, pos, tri, weights, is_boundary, A_eff, va_eff);
  orc_vertex_normals(nv, nf, pos, tri, normals);
  double total = 0.0;
  for (int i = 0; i < nv; ++i) {
    is_int[i] = is_boundary ? (is_boundary[i] == 0) : 1;
    double safe = A_vor[i] > 1.0e-12 ? A_vor[i] : 1.0e-12;
    double k_mag = norm3(k_vecs + 3 * i);
    double H = k_mag / (2.0 * safe);
    double ratio = safe > 1.0e-15 ? A_eff[i] / safe : 0.0;
    double scale_K, fe, fv;
    if (model == 0) {
      double term = (2.0 * H) - c0_arr[i];
      if (!is_int[i]) term = 0.0;
      total += kappa[i] * (term * term) * A_eff[i];
      scale_K = kappa[i] * term * ratio;
      fe = 0.5 * kappa[i] * (term * term);
      fv = -2.0 * kappa[i] * term * ratio * H;
    } else {
      double He = is_int[i] ? H : 0.0;
      total += kappa[i] * (He * He) * A_eff[i];
      scale_K = kappa[i] * He * ratio;
      fe = kappa[i] * (He * He);
      fv = -2.0 * kappa[i] * (He * He) * ratio;
    }
    fA_eff[i] = fe;
    fA_vor[i] = fv;
    for (int d = 0; d < 3; ++d) {
      double kd = k_mag > 1.0e-15 ? k_vecs[3 * i + d] / k_mag : normals[3 * i + d];
      fK[3 * i + d] = kd * scale_K;
    }
  }
  *E_out = model == 0 ? 0.5 * total : total;
  if (fK_out) memcpy(fK_out, fK, sizeof(double) * 3 * nvs);
  if (fA_eff_out) memcpy(fA_eff_out, fA_eff, sizeof(double) * nvs);
  if (fA_vor_out) memcpy(fA_vor_out, fA_vor, sizeof(double) * nvs);
  if (!grad) goto done;
  if (mode == 1) {
    /* bending.py:163-167  approx: grad -= L fK ; boundary rows of grad := 0 */
    double *lap = (double *)malloc(sizeof(double) * 3 * nvs);
    if (!lap) {
      rc = -1;
      goto done;
    }
    orc_apply_beltrami_laplacian(3, nv, nf, weights, tri, fK, lap);
    for (size_t i = 0; i < 3 * nvs; ++i) grad[i] -= lap[i];
    free(lap);
    for (int i = 0; i < nv; ++i)
      if (!is_int[i]) grad[3 * i] = grad[3 * i + 1] = grad[3 * i + 2] = 0.0;
  } else {
    rc = backprop_bending(nv, nf, pos, tri, weights, is_int, fA_eff, fA_vor, fK,
                          grad);
  }
done:
  free(k_vecs);
  free(A_vor);
  free(weights);
  free(A_eff);
  free(va_eff);
  free(fK);
  free(fA_eff);
  free(fA_vor);
  free(normals);
  free(is_int);
  return rc;
}

/* ------------------------------------------------------------------------
 * modules/energy/bending.py:62-87  compute_energy_array (energy-only path
 * used by EvaluationManager.compute_energy_array_total): H = |K/(2A)|,
 * density = 0.5 (2H-c0)^2 | H^2, boundary rows 0, E = sum kappa*density*A_eff.
 * per_vertex (nv) may be NULL.
 * ---------------------------------------------------------------------- */
ORC_API int orc_bending_energy(int nv, int nf, const double *pos,
                               const int32_t *tri, const double *kappa,
                               const double *c0_arr, const uint8_t *is_boundary,
                               int model, double *E_out, double *per_vertex) {
  *E_out = 0.0;
  if (per_vertex) memset(per_vertex, 0, sizeof(double) * (size_t)nv);
  if (nf == 0) return 0;
  double kmax = 0.0;
  for (int i = 0; i < nv; ++i)
    if (i == 0 || kappa[i] > kmax) kmax = kappa[i];
  if (kmax == 0.0) return 0;
  size_t nvs = (size_t)nv, nfs = (size_t)nf;
  double *k_vecs = (double *)malloc(sizeof(double) * 3 * nvs);
  double *A_vor = (double *)malloc(sizeof(double) * nvs);
  double *weights = (double *)malloc(sizeof(double) * 3 * nfs);
  double *A_eff = (double *)malloc(sizeof(double) * nvs);
  double *va_eff = (double *)malloc(sizeof(double) * 3 * nfs);
  if (!k_vecs || !A_vor || !weights || !A_eff || !va_eff) {
    free(k_vecs);
    free(A_vor);
    free(weights);
    free(A_eff);
    free(va_eff);
    return -1;
  }
  orc_compute_curvature_data(nv, nf, pos, tri, k_vecs, A_vor, weights, NULL,
                             NULL, NULL);
  orc_effective_areas(nv, nf, pos, tri, weights, is_boundary, A_eff, va_eff);
  double total = 0.0;
  for (int i = 0; i < nv; ++i) {
    double safe = A_vor[i] > 1.0e-12 ? A_vor[i] : 1.0e-12;
    double h[3];
    for (int d = 0; d < 3; ++d) h[d] = k_vecs[3 * i + d] / (2.0 * safe);
    double H = norm3(h);
    double density;
    if (model == 0) {
      double t = 2.0 * H - c0_arr[i];
      density = 0.5 * (t * t);
    } else {
      density = H * H;
    }
    if (is_boundary && is_boundary[i]) density = 0.0;
    double e = kappa[i] * density * A_eff[i];
    if (per_vertex) per_vertex[i] = e;
    total += e;
  }
  *E_out = total;
  free(k_vecs);
  free(A_vor);
  free(weights);
  free(A_eff);
  free(va_eff);
  return 0;
}

/* ------------------------------------------------------------------------
 * geometry/body.py:70-148 compute_volume (vectorised branch :104-123):
 * V = sum((v1 x v2) . v0) / 6 over the body's triangle rows.
 * body_rows may be NULL (all facets).
 * ---------------------------------------------------------------------- */
ORC_API double orc_volume(int nv, int nrows, const double *pos,
                          const int32_t *tri, const int32_t *body_rows) {
  (void)nv;
  double s = 0.0;
  for (int r = 0; r < nrows; ++r) {
    int f = body_rows ? body_rows[r] : r;
    const double *v0 = pos + 3 * tri[3 * f];
    const double *v1 = pos + 3 * tri[3 * f + 1];
    const double *v2 = pos + 3 * tri[3 * f + 2];
    double c[3];
    cross3(v1, v2, c);
    s += dot3(c, v0);
  }
  return s / 6.0;
}

/* geometry/body.py:150-190 accumulate_volume_gradient:
 * grad[v0] += (v1 x v2) * (factor/6) and cyclic. */
ORC_API void orc_volume_gradient(int nv, int nrows, const double *pos,
                                 const int32_t *tri, const int32_t *body_rows,
                                 double factor, double *grad) {
  (void)nv;
  double s = factor / 6.0;
  for (int r = 0; r < nrows; ++r) {
    int f = body_rows ? body_rows[r] : r;
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    double g0[3], g1[3], g2[3];
    cross3(v1, v2, g0);
    cross3(v2, v0, g1);
    cross3(v0, v1, g2);
    for (int d = 0; d < 3; ++d) {
      grad[3 * i0 + d] += g0[d] * s;
      grad[3 * i1 + d] += g1[d] * s;
      grad[3 * i2 + d] += g2[d] * s;
    }
  }
}

/* ------------------------------------------------------------------------
 * modules/energy/tilt.py:99-172  compute_energy_and_gradient_array
 * (vertex-tilt magnitude energy, lumped barycentric mass):
 *   coeff_f = 0.5 k_t (|t0|^2+|t1|^2+|t2|^2)/3 ; E = sum coeff_f A_f
 *   shape grad: coeff_f * dA/dv_k, dA/dv0 = 0.5 nhat x (v2 - v1) (cyclic)
 *   tilt grad : k_t * t_v * A_v,  A_v = sum_incident A_f/3   (:160-170)
 * Facets with |n| < 1e-12 are masked out.  grad / tilt_grad accumulate and
 * may be NULL.
 * ---------------------------------------------------------------------- */
ORC_API void orc_tilt_energy_and_gradient(int nv, int nf, const double *pos,
                                          const double *tilts,
                                          const int32_t *tri, double k_tilt,
                                          double *grad, double *tilt_grad,
                                          double *E_out) {
  double E = 0.0;
  double *vertex_areas = NULL;
  *E_out = 0.0;
  if (tilt_grad) {
    vertex_areas = (double *)calloc((size_t)nv, sizeof(double));
    if (!vertex_areas) return;
  }
  for (int f = 0; f < nf; ++f) {
    int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const double *v0 = pos + 3 * i0, *v1 = pos + 3 * i1, *v2 = pos + 3 * i2;
    const double *t0 = tilts + 3 * i0, *t1 = tilts + 3 * i1, *t2 = tilts + 3 * i2;
    double a[3], b[3], n[3];
    sub3(v1, v0, a);
    sub3(v2, v0, b);
    cross3(a, b, n);
    double A2 = norm3(n);
    if (!(A2 >= 1.0e-12)) continue;
    double area = 0.5 * A2;
    double ssum = (dot3(t0, t0) + dot3(t1, t1)) + dot3(t2, t2);
    double coeff = 0.5 * k_tilt * (ssum / 3.0);
    E += coeff * area;
    if (grad) {
      double nhat[3], e[3], g[3];
      for (int d = 0; d < 3; ++d) nhat[d] = n[d] / A2;
      sub3(v2, v1, e);
      cross3(nhat, e, g);
      for (int d = 0; d < 3; ++d) grad[3 * i0 + d] += coeff * (0.5 * g[d]);
      sub3(v0, v2, e);
      cross3(nhat, e, g);
      for (int d = 0; d < 3; ++d) grad[3 * i1 + d] += coeff * (0.5 * g[d]);
      sub3(v1, v0, e);
      cross3(nhat, e, g);
      for (int d = 0; d < 3; ++d) grad[3 * i2 + d] += coeff * (0.5 * g[d]);
    }
    if (vertex_areas) {
      double third = area / 3.0;
      vertex_areas[i0] += third;
      vertex_areas[i1] += third;
      vertex_areas[i2] += third;
    }
  }
  if (tilt_grad) {
    for (int i = 0; i < nv; ++i)
      for (int d = 0; d < 3; ++d)
        tilt_grad[3 * i + d] += k_tilt * tilts[3 * i + d] * vertex_areas[i];
    free(vertex_areas);
  }
  *E_out = E;
}
