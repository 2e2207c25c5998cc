/*
 * membrane_hip.h -- C ABI of libmembrane_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of AvishaiBarnoy/membrane_solver: the
 * per-iteration energy + gradient assembly (surface tension, Helfrich /
 * Willmore bending, volume penalty + volume-constraint row) and the GD / CG
 * steppers with Armijo backtracking that consume it.  Reference citations are
 * relative to the reference checkout.
 *
 * Conventions
 *   - plain C types only; all host arrays are row-major, positions/gradients
 *     (nv,3) double, triangle rows (nf,3) int32 zero based -- the layouts of
 *     Mesh.positions_view (geometry/mesh.py:372-389) and
 *     Mesh.triangle_row_cache (geometry/mesh.py:597-624).  A row-major (n,3)
 *     array is byte-identical to the (3,n) Fortran-order arrays the
 *     reference's f2py kernels take, so the same pointers serve both seams.
 *   - every function returns MS_OK (0) or a negative MS_ERR_*; nothing
 *     throws; ms_last_error() gives the text.  No global mutable state but
 *     the last-error string of failed ms_create calls.
 *   - one context per device/process; a context is not thread-safe (the
 *     reference is single threaded).
 *   - all state lives in HBM between calls; host arrays cross PCIe only in
 *     the ms_set_* / ms_get_* calls and the *_host seam calls.
 */
#ifndef MEMBRANE_HIP_H
#define MEMBRANE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS_OK 0
#define MS_ERR_INVALID (-1)       /* bad argument / shape                      */
#define MS_ERR_HIP (-2)           /* a HIP runtime call failed                 */
#define MS_ERR_TILE_CAPACITY (-3) /* a vertex patch does not fit LDS / uint16  */
#define MS_ERR_STATE (-4)         /* call order (e.g. gradient before energy)  */
#define MS_ERR_NOMEM (-5)

/* energy-module bits (modules/energy/{surface,bending,volume}.py) */
#define MS_MOD_SURFACE 1u
#define MS_MOD_BENDING 2u
#define MS_MOD_VOLUME_PENALTY 4u /* modules/energy/volume.py:94-128            */
/* constraint row: modules/constraints/volume.py:43-66 + the k==1 dense KKT
 * branch of runtime/constraint_manager.py:293-301 */
#define MS_CON_VOLUME 8u
/* also reduce the body volume in energy passes (the Lagrange volume-drift
 * check of runtime/minimizer.py:1478-1513 needs V even when no volume module
 * is loaded); no effect on energies or gradients */
#define MS_TRACK_VOLUME 16u
/* vertex-tilt magnitude energy 1/2 k_t sum |t_v|^2 A_v (modules/energy/tilt.py:99-172) */
#define MS_MOD_TILT 32u
#define MS_MOD_BENDING_TILT 64u /* modules/energy/bending_tilt.py (replaces MS_MOD_BENDING) */
#define MS_MOD_TILT_SMOOTH 128u /* modules/energy/tilt_smoothness.py (ambient_v1 transport) */
/* two-leaflet tilt fields (Mesh.tilts_in_view / tilts_out_view): tilt magnitude
 * (modules/energy/tilt_in.py, tilt_out.py -> tilt_leaflet.py:26-169) and tilt smoothness
 * (tilt_smoothness_in.py, tilt_smoothness_out.py -> tilt_smoothness_leaflet.py:17-79) */
#define MS_MOD_TILT_IN 256u
#define MS_MOD_TILT_OUT 512u
#define MS_MOD_TILT_SMOOTH_IN 1024u
#define MS_MOD_TILT_SMOOTH_OUT 2048u
/* leaflet bending + tilt-splay coupling (modules/energy/bending_tilt_in.py, bending_tilt_out.py ->
 * bending_tilt_leaflet.py:231-758, default options, analytic gradient mode) */
#define MS_MOD_BENDING_TILT_IN 4096u
#define MS_MOD_BENDING_TILT_OUT 8192u
/* soft disk tilt-profile target (modules/energy/tilt_disk_target_in.py:160-286, tilt_disk_target_out.py) */
#define MS_MOD_TILT_DISK_TARGET_IN 16384u
#define MS_MOD_TILT_DISK_TARGET_OUT 32768u
#define MS_LEAFLET_IN 0
#define MS_LEAFLET_OUT 1

/* bending_params.py:19-33 */
#define MS_BEND_HELFRICH 0
#define MS_BEND_WILLMORE 1
#define MS_GRAD_ANALYTIC 0
#define MS_GRAD_APPROX 1

#define MS_STEPPER_GD 0 /* runtime/steppers/gradient_descent.py:35-84        */
#define MS_STEPPER_CG 1 /* runtime/steppers/conjugate_gradient.py:52-119     */

typedef struct ms_ctx ms_ctx;

/* Names of the device-resident per-vertex buffers (internal patch order). */
enum ms_buffer {
  MS_BUF_X = 0,     /* positions                      (nvp,3) */
  MS_BUF_XT = 1,    /* trial positions                (nvp,3) */
  MS_BUF_G = 2,     /* gradient                       (nvp,3) */
  MS_BUF_GC = 3,    /* volume-constraint row dV/dx    (nvp,3) */
  MS_BUF_D = 4,     /* search direction               (nvp,3) */
  MS_BUF_PG = 5,    /* CG history: previous gradient  (nvp,3) */
  MS_BUF_PD = 6,    /* CG history: previous direction (nvp,3) */
  MS_BUF_FK = 7,    /* bending: factor_K_vec          (nvp,3) */
  MS_BUF_FA = 8,    /* bending: fA_eff, fA_vor        (nvp,2) */
  MS_BUF_SCAL = 9,  /* reduction scalars              (MS_NSCAL) */
  MS_BUF_COUNT = 10
};

/* Reduction scalars produced on the device (ms_fetch_scalars). */
enum ms_scalar {
  MS_S_ESURF = 0,   /* sum gamma_f A_f                                  */
  MS_S_VOL = 1,     /* body volume, geometry/body.py:104-123            */
  MS_S_EBEND = 2,   /* bending energy, modules/energy/bending.py:117-144 */
  MS_S_MINEDGE2 = 3, /* min squared edge length, runtime/topology.py:174  */
  MS_S_GUARD = 4,   /* >0: normal-rotation guard tripped, topology.py:13  */
  MS_S_GGC = 5,     /* <g, gC>                                          */
  MS_S_GCGC = 6,    /* <gC, gC>                                         */
  MS_S_GNORM2 = 7,  /* |g|^2 after projection and fixed-row zeroing     */
  MS_S_GDOTD = 8,   /* <g, d>                                           */
  MS_S_MAXD2 = 9,   /* max_i |d_i|^2 over movable rows                  */
  MS_S_ETILT = 10,  /* tilt magnitude energy, modules/energy/tilt.py    */
  MS_S_EBT = 11,    /* bending + tilt-splay energy, modules/energy/bending_tilt.py */
  MS_S_TGNORM2 = 12, /* |tilt gradient|^2 over free rows (tilt_relaxation.py:318) */
  MS_S_TRZ = 13,    /* <r, M^-1 r> of the tilt CG (tilt_relaxation.py:369,416) */
  MS_S_MAXG2 = 14,  /* max_i |g_i|^2 over movable rows (= max|d_i|^2 of a steepest-descent restart) */
  MS_S_ETS = 15,    /* tilt smoothness (Dirichlet) energy, modules/energy/tilt_smoothness.py */
  MS_S_ETILT_IN = 16,  /* leaflet energies: tilt_in / tilt_out / tilt_smoothness_in / _out */
  MS_S_ETILT_OUT = 17,
  MS_S_ETS_IN = 18,
  MS_S_ETS_OUT = 19,
  MS_S_TGNORM2_IN = 20, /* leaflet relaxation: |grad|^2 and <r, M^-1 r> per leaflet; the driver */
  MS_S_TGNORM2_OUT = 21, /* adds the two (tilt_relaxation.py:862-868, 1139-1141)               */
  MS_S_TRZ_IN = 22,
  MS_S_TRZ_OUT = 23,
  MS_S_EBT_IN = 24,  /* bending_tilt_in / bending_tilt_out energies */
  MS_S_EBT_OUT = 25,
  MS_S_EDT_IN = 26,  /* tilt_disk_target_in / _out energies */
  MS_S_EDT_OUT = 27,
  MS_S_DTR_IN = 28,  /* largest in-plane distance of a disk row (the default disk radius), max-reduced */
  MS_S_DTR_OUT = 29,
  MS_NSCAL = 30
};

typedef struct ms_params {
  uint32_t modules;        /* MS_MOD_* | MS_CON_VOLUME                          */
  int bending_model;       /* MS_BEND_*                                         */
  int bending_grad_mode;   /* MS_GRAD_*                                         */
  double volume_stiffness; /* penalty mode k                                    */
  double target_volume;    /* V0 (penalty energy and constraint target)         */
} ms_params;

typedef struct ms_stepper_params {
  int stepper;             /* MS_STEPPER_*                                      */
  int max_iter;            /* Armijo trials, default 10                         */
  double beta;             /* backtrack factor 0.7                              */
  double c;                /* Armijo slope 1e-4                                 */
  double gamma;            /* growth 1.5                                        */
  double alpha_max_factor; /* 10                                                */
  int restart_interval;    /* CG restart, 10                                    */
  double edge_fraction;    /* gp["shape_step_edge_fraction"], 0 = off           */
  int reuse_energy0;       /* 0: re-evaluate energy0 like line_search.py:294;
                              1: reuse the energy of the gradient evaluation;
                              2: 1 + an accepted trial pass (which also writes the
                                 bending factors) serves as the next step's energy
                                 pass.  Same kernel on the same doubles: all three
                                 modes give bitwise identical trajectories.       */
  int enforce_volume;      /* line_search.py:428-487 with constraint_enforcer =
                              Minimizer._enforce_constraints (minimizer.py:1379) and
                              volume_projection_during_minimization on: every trial
                              that passed the guard is projected onto the target
                              volume (volume.enforce_constraint: 3 linearised steps,
                              tol 1e-12) BEFORE its energy is taken; a rejected
                              trial restores the positions.  Unqueued trials.      */
  int precondition;        /* conjugate_gradient.py:74-76 (CG only): the direction is
                              built from the row-normalised gradient g_i/(|g_i|+1e-8);
                              the history and the Armijo slope keep the raw gradient.
                              Unfused direction pass, unqueued trials.              */
} ms_stepper_params;

typedef struct ms_step_result {
  int success;        /* line search accepted a step                            */
  int converged;      /* |g| < tol before stepping (minimizer.py:1324)          */
  int trials;         /* energy evaluations spent in the line search            */
  int guard_rejects;  /* trials rejected by the normal-rotation guard           */
  double next_step;   /* step size to use next (line_search.py:398-404,425)     */
  double energy;      /* accepted energy, or energy0 on failure                 */
  double alpha;       /* accepted alpha (0 on failure)                          */
  double energy_eval; /* energy of the gradient evaluation                      */
  double grad_norm;   /* |g|_2                                                  */
  double g_dot_d;
  double volume;      /* body volume at the (possibly new) positions            */
} ms_step_result;

const char *ms_version(void);
int ms_device_count(void);
const char *ms_last_error(const ms_ctx *ctx); /* ctx may be NULL */

/*
 * Build a context: vertices are put in patch order (recursive coordinate bisection down to single tiles), cut into tiles
 * of `tile_vertices` owned vertices (0 = 256; 64, 128, 129..256 or 512 -- 129..255 rows run on the 256-thread
 * instances), and the tile->facet /
 * tile->halo-vertex CSR is pushed to HBM once.  Replaces the per-step reads of
 * Mesh.triangle_row_cache / fixed_mask / boundary_vertex_ids
 * (geometry/mesh.py:597-624, :210-232, :304-319).  `fixed`, `boundary`,
 * `body_facets` (1 = facet belongs to the body, geometry/body.py:60-68) may be
 * NULL (none fixed / closed surface / all facets).  Facets with an index
 * outside [0,nv) are dropped, as fortran_kernels/surface_energy.f90:57-59 does.
 * shard_rank/shard_count: this context evaluates tiles of that shard only
 * (1 process per GPU; per-vertex buffers stay full size, see ms_shard_info).
 */
int ms_create(ms_ctx **out, int device, int nv, int nf, const double *positions,
              const int32_t *tri, const uint8_t *fixed, const uint8_t *boundary,
              const uint8_t *body_facets, int tile_vertices, int shard_rank,
              int shard_count);
void ms_destroy(ms_ctx *ctx);

/* Launch on the caller's HIP stream (hipStream_t as void*); NULL = own stream. */
int ms_set_stream(ms_ctx *ctx, void *hip_stream);

/* Mesh.get_facet_parameter_array("surface_tension") (geometry/mesh.py:234-265) */
int ms_set_surface_tension(ms_ctx *ctx, const double *gamma /* nf */);
/* bending_params._per_vertex_params (modules/energy/bending_params.py:41-115) */
int ms_set_bending_params(ms_ctx *ctx, const double *kappa /* nv */,
                          const double *c0 /* nv */);
int ms_set_params(ms_ctx *ctx, const ms_params *p);
/* Per-vertex accumulation in the two big tile kernels (energy pass, gradient pass):
 * 0 (default) LDS atomic adds -- fastest, the floating-point summation order (hence the
 * last bits) may differ between runs; 1 staged CSR gather in a fixed order -- bitwise
 * reproducible run to run, ~20 % slower.  Environment MS_DETERMINISTIC=1 makes 1 the
 * default for new contexts. */
int ms_set_deterministic(ms_ctx *ctx, int on);
/* Mesh.tilts_view() (geometry/mesh.py:391-430) + gp["tilt_rigidity"]; tilts (nv,3) row-major */
int ms_set_tilts(ms_ctx *ctx, const double *tilts /* nv*3 */, double tilt_rigidity);
int ms_get_tilts(ms_ctx *ctx, double *tilts /* nv*3 */);
/* dE/dt = k_t t_v A_v of the last gradient evaluation (tilt.py:160-170) */
int ms_get_tilt_gradient(ms_ctx *ctx, double *tilt_grad /* nv*3 */);
/* ---- tilt field (single vertex-tilt field): relaxation at frozen positions ----
 * ms_set_tilt_fixed: vertex.tilt_fixed flags (runtime/minimizer_helpers.py:49-75),
 *   NULL clears them.
 * ms_tilt_energy_and_gradient: energy of the tilt-reading modules (tilt,
 *   bending_tilt) and the dense tilt gradient dE/dt at the stored tilts
 *   (runtime/evaluation_manager.py:386-462); tilt_grad (nv*3, host) may be NULL
 *   (then read it back with ms_get_tilt_gradient).
 * ms_relax_tilts: TiltRelaxationManager.relax_tilts
 *   (runtime/steppers/tilt_relaxation.py:237-424) on the device: tilts projected
 *   to the tangent planes, then gradient descent or (Jacobi-preconditioned)
 *   Fletcher-Reeves CG with halving back-tracking (<= 12 halvings, accept on
 *   E1 <= E0); tilt-fixed rows keep their projected value. */
typedef struct ms_tilt_relax_params {
  int solver;        /* 0 gradient descent, 1 conjugate gradient ("tilt_solver") */
  int max_iters;     /* tilt_inner_steps / tilt_coupled_steps / tilt_cg_max_iters */
  double step_size;  /* "tilt_step_size" */
  double tol;        /* "tilt_tol" on |dE/dt| over free rows, <= 0: off */
  int jacobi;        /* CG: "tilt_cg_preconditioner" == jacobi */
} ms_tilt_relax_params;
int ms_set_tilt_fixed(ms_ctx *ctx, const uint8_t *tilt_fixed /* nv or NULL */);
/* gp["tilt_smoothness_rigidity"] of the tilt_smoothness module */
int ms_set_tilt_smoothness(ms_ctx *ctx, double k_smooth);
int ms_tilt_energy_and_gradient(ms_ctx *ctx, double *energy, double *tilt_grad);
int ms_relax_tilts(ms_ctx *ctx, const ms_tilt_relax_params *params,
                   int *iters_out, int *evals_out);

/* ---- two-leaflet tilt fields (tilts_in / tilts_out) ----
 * ms_set_leaflet_tilts: Mesh.tilts_in_view()/tilts_out_view() (nv,3 row-major), the
 *   vertex.tilt_fixed_in / tilt_fixed_out flags (minimizer.py:460-486; NULL = none) and the
 *   leaflet's module parameters.  Modules are switched on by the MS_MOD_*_IN/_OUT bits of
 *   ms_params.modules; both leaflets must be set before the first evaluation.
 * Energies and the shape gradient (coeff_f dA/dx, tilt_leaflet.py:152-166) join the ordinary
 *   evaluation; every line-search trial projects both fields onto the trial surface's tangent
 *   planes like the single field (Mesh.project_tilts_to_tangent, geometry/mesh.py:788-814).
 * ms_leaflet_tilt_energy_and_gradient: EvaluationManager.compute_energy_and_leaflet_tilt_
 *   gradients_array with tilt_vertex_areas given (runtime/evaluation_manager.py:630-742): the
 *   magnitude modules take the lumped vertex-area form 1/2 k sum |t_v|^2 A_v there.
 * ms_relax_leaflet_tilts: TiltRelaxationManager.relax_leaflet_tilts
 *   (runtime/steppers/tilt_relaxation.py:426-1478), default options: GD or Jacobi/plain
 *   Fletcher-Reeves CG over the concatenated (in, out) field, <= 12 halvings, accept on
 *   E1 <= E0, fixed rows keep their projected start value. */
typedef struct ms_leaflet_params {
  double tilt_modulus;        /* "tilt_modulus_in|out" (tilt_params.py:6-12); 0 = no contribution */
  int tilt_mass_consistent;   /* "tilt_mass_mode_in|out" == consistent (tilt_params.py:15-23)  */
  double smoothness;          /* tilt_smoothness_in|out rigidity (tilt_smoothness_utils.py:95-100) */
  double precond_smoothness;  /* rigidity in the Jacobi diagonal (preconditioners.py:111-120)   */
} ms_leaflet_params;
int ms_set_leaflet_tilts(ms_ctx *ctx, int leaflet, const double *tilts /* nv*3 */,
                         const uint8_t *tilt_fixed /* nv or NULL */, const ms_leaflet_params *params);
int ms_get_leaflet_tilts(ms_ctx *ctx, int leaflet, double *tilts /* nv*3 */);
/* tilt_disk_target_in / _out: E = 1/2 k int |t - theta(r) r_hat|^2 dA over the tagged disk rows, theta(r) =
 * theta_B I1(lambda r)/I1(lambda R) (30-term series, tilt_disk_target_in.py:148-157) or theta_B r/R when
 * |lambda| < 1e-12; r, r_hat in the plane through `center` with unit `normal`; R = radius, or the largest
 * in-plane distance of a disk row when radius <= 0 (:216-220).  disk_rows: nv flags (NULL switches it off). */
typedef struct ms_disk_target_params {
  double strength;   /* tilt_disk_target_strength_in|out */
  double theta_b;    /* tilt_disk_target_theta_B[_in|_out] */
  double lambda;     /* tilt_disk_target_lambda[_in|_out], else sqrt(tilt_modulus / bending_modulus) */
  double center[3];
  double normal[3];  /* required (the SVD plane fit of :80-93 is host work) */
  double radius;     /* <= 0: from the disk rows */
} ms_disk_target_params;
int ms_set_leaflet_disk_target(ms_ctx *ctx, int leaflet, const uint8_t *disk_rows /* nv or NULL */,
                               const ms_disk_target_params *params);
/* per-vertex (kappa, c0) of bending_tilt_in / bending_tilt_out (modules/energy/bt_params.py:225-318:
 * bending_modulus_in|out else bending_modulus; spontaneous_curvature_in|out else the global one) */
int ms_set_leaflet_bending(ms_ctx *ctx, int leaflet, const double *kappa /* nv */, const double *c0 /* nv */);
int ms_leaflet_tilt_energy_and_gradient(ms_ctx *ctx, double *energy, double *grad_in /* nv*3 or NULL */,
                                        double *grad_out /* nv*3 or NULL */);
/* The same evaluation with module_form != 0: the magnitude modules tilt_in / tilt_out in their OWN mass mode, as
 * the plugin API evaluates them when a caller hands it tilt_in_grad_arr / tilt_out_grad_arr -- lumped
 * k A_f/3 t_k, or consistent k A_f/12 (2 t_k + t_a + t_b) per corner
 * (modules/energy/tilt_leaflet.py:101-150).  module_form == 0 is the call above. */
int ms_leaflet_tilt_energy_and_gradient_ex(ms_ctx *ctx, int module_form, double *energy,
                                           double *grad_in /* nv*3 or NULL */, double *grad_out /* nv*3 or NULL */);
int ms_relax_leaflet_tilts(ms_ctx *ctx, const ms_tilt_relax_params *params,
                           int *iters_out, int *evals_out);

/* Mesh.project_tilts_to_tangent (geometry/mesh.py:788-814): t <- t - (t.n) n with the
 * unit vertex normals of the current positions (triangle_ops.py:55-73) */
int ms_project_tilts_to_tangent(ms_ctx *ctx);

/* compute_angle_defects (geometry/curvature.py:335-403): per-vertex angle defect 2 pi - sum of the incident
 * triangle angles (law of cosines, lengths clamped at 1e-15, cosines clipped), 0 on boundary vertices: the
 * integrated Gaussian curvature.  Their sum is 2 pi chi on a closed manifold mesh (Gauss-Bonnet). */
int ms_angle_defects(ms_ctx *ctx, double *defects /* nv */);
/* geometry/curvature.compute_curvature_fields (:404-448) at the context's positions, one tile-kernel pass: each output
 * is (nv,3) in caller row order and may be NULL --
 *   mean_curvature_normal : K_v / (2 max(A_v, 1e-12)), K_v and the mixed-Voronoi areas A_v as compute_curvature_data;
 *   h_area_anglesum       : (H = |mean-curvature normal|, mixed area A_v, sum of the incident triangle angles);
 *   defect_kg             : (angle defect 2 pi - angle sum, 0 on boundary rows; K_G = defect / max(A_v, 1e-12); 0);
 *   principal             : (k1, k2 = H +- sqrt(max(H^2 - K_G, 0)); 0).
 * The raw angle sums also serve the open-surface Gauss-Bonnet invariant of modules/energy/gaussian_curvature.py
 * (runtime/diagnostics/gauss_bonnet.py:260-340). */
int ms_curvature_fields(ms_ctx *ctx, double *mean_curvature_normal, double *h_area_anglesum,
                        double *defect_kg, double *principal);

int ms_set_positions(ms_ctx *ctx, const double *positions /* nv*3 */);
int ms_get_positions(ms_ctx *ctx, double *positions /* nv*3 */);
int ms_get_gradient(ms_ctx *ctx, double *grad /* nv*3 */);
int ms_get_vertex_buffer(ms_ctx *ctx, int buffer, double *out /* nv*ncomp */);

/*
 * Minimizer.compute_energy_and_gradient_array (runtime/minimizer.py:941-992):
 * module loop, volume-constraint projection, fixed rows zeroed.  energies[4] =
 * {surface, bending, volume-penalty, tilt}.  grad may be NULL (stays on device).
 */
int ms_energy_and_gradient(ms_ctx *ctx, double energies[4], double *grad);
/*
 * What ONE energy module's compute_energy_and_gradient_array accumulates (the plugin seam,
 * runtime/energy_manager.py:21, evaluation_manager.py:134-151): the module loop's raw sum -- fixed rows are NOT
 * zeroed and no constraint row is projected out (the minimizer does both afterwards, minimizer.py:979-990).
 */
int ms_energy_and_raw_gradient(ms_ctx *ctx, double energies[4], double *grad);
/* EvaluationManager.compute_energy_array_total (evaluation_manager.py:184-225) */
int ms_energy(ms_ctx *ctx, double energies[4]);

/* One stepper.step at the current positions (the body of minimizer.py:1314-1374
 * + line_search.py:267-426), fully device resident. */
int ms_step(ms_ctx *ctx, const ms_stepper_params *sp, double step_size,
            double tol, ms_step_result *out);
/* ConjugateGradient.reset (conjugate_gradient.py:44-50) */
int ms_reset_stepper(ms_ctx *ctx);

/* Minimizer.minimize loop body (runtime/minimizer.py:1230-1535) for n_steps
 * iterations without returning to the host language: [tilt relaxation] ->
 * ms_step -> step-size bookkeeping (fixed mode, zero-step counter, stepper
 * reset on failure, :1425-1476) -> Lagrange volume-drift check with device
 * projection (:1478-1513).  step_log (n_steps x 8, may be NULL) receives per
 * iteration {success, next_step, energy, energy_eval, grad_norm, g_dot_d,
 * alpha, trials}.  Identical to driving ms_step from the caller. */
typedef struct ms_minimize_params {
  ms_stepper_params stepper;
  double step_size;        /* in: first step size                              */
  double tol;              /* convergence |g| < tol                            */
  int fixed_step_mode;     /* gp["step_size_mode"] == "fixed"                  */
  double fixed_step;       /* gp["step_size"]                                  */
  int max_zero_steps;      /* gp["max_zero_steps"], 10                         */
  double step_size_floor;  /* gp["step_size_floor"], 1e-8                      */
  int drift_check;         /* Lagrange mode, no per-trial projection, target set */
  double target_volume;
  double volume_tolerance; /* gp["volume_tolerance"], 1e-3                     */
  int project_on_drift;    /* an enforceable volume constraint module exists   */
  int relax_tilts;         /* tilt_solve_mode nested/coupled                   */
  ms_tilt_relax_params relax;
} ms_minimize_params;

typedef struct ms_minimize_result {
  int iterations;          /* iterations executed                              */
  int converged;           /* stopped on |g| < tol                             */
  int zero_step_exit;      /* stopped after max_zero_steps failed tiny steps   */
  int step_success;        /* success flag of the last line search            */
  int accepted, trials, guard_rejects, moved;  /* totals; moved: x changed     */
  double step_size;        /* out: step size for the next call                 */
  double energy_eval;      /* energy of the last gradient evaluation           */
  double grad_norm;
  int volume_cache_current; /* the loop ended right after a drift check that did NOT project: Body's cached
                             * volume is current, so a finalize projection starts from the cached gradient
                             * (pass it as first_step_cached to ms_project_volume_cached)               */
  int energy_current_valid; /* the step logic already holds the module energies of the positions the loop ended at (the
                             * accepted trial's, or the unchanged x's after a failed step; evaluation reuse level 2): */
  double energy_current;    /* ... their sum -- Minimizer.minimize's final compute_energy() (minimizer.py:1524)
                             * without another energy pass                                                */
} ms_minimize_result;

int ms_minimize(ms_ctx *ctx, const ms_minimize_params *params, int n_steps,
                ms_minimize_result *out, double *step_log);
/* modules/constraints/volume.enforce_constraint projection loop (:117-149) */
int ms_project_volume(ms_ctx *ctx, double target, double tol, int max_iter,
                      int *iters_out, double *volume_out);
/* The same projection as the reference's minimizer reaches it.  Body caches the volume gradient of its last
 * compute_volume_and_gradient evaluation (geometry/body.py:386-407, :463); compute_volume (:70-120) refreshes only
 * the cached volume and mesh version.  An enforce that follows a compute_volume at the same mesh version -- the
 * Lagrange drift check, runtime/minimizer.py:1492 -- therefore takes its FIRST step with the current volume but the
 * gradient the previous enforce evaluated last.  first_step_cached != 0 reproduces that (no effect while nothing is
 * cached); every fresh evaluation refreshes the cache, also through ms_project_volume. */
int ms_project_volume_cached(ms_ctx *ctx, double target, double tol, int max_iter,
                             int first_step_cached, int *iters_out, double *volume_out);

/* ---- phase-level entry points (multi-GPU drivers interleave collectives) -- */
/* energy pass over this shard's tiles at x (+ alpha*d if use_direction);
 * writes trial positions to XT when write_trial; guard != 0 also evaluates the
 * normal-rotation guard.  Asynchronous; partial sums are reduced into SCAL. */
int ms_phase_energy(ms_ctx *ctx, int use_direction, double alpha, int write_trial,
                    int guard, int write_bending_factors);
int ms_phase_gradient(ms_ctx *ctx);   /* gradient pass (needs bending factors) */
int ms_phase_direction(ms_ctx *ctx, int stepper, int use_history);
int ms_phase_accept(ms_ctx *ctx, int keep_history); /* x <- xt, CG history     */
int ms_fetch_scalars(ms_ctx *ctx, double *out /* MS_NSCAL */); /* synchronises */
int ms_store_scalars(ms_ctx *ctx, const double *in /* MS_NSCAL */);

/* Sharded accept: every rank forms xt = x + alpha*d on ALL rows (x and d are
 * replicated after the direction all-gather, so no exchange is needed), then
 * x <-> xt and, if keep_history, the CG history swap of ms_phase_accept. */
int ms_phase_commit_trial(ms_ctx *ctx, double alpha, int keep_history);

/* Fused K_C + direction pass (no constraint row, no tilt module): the sharded
 * counterpart of what ms_step queues (runtime/minimizer.py:941-992 followed by
 * conjugate_gradient.py:60-100 / gradient_descent.py:35-84); reduces the
 * direction scalars of this rank's rows. */
int ms_phase_gradient_direction(ms_ctx *ctx, int stepper, int use_history);
/* After a boundary exchange of fK/fA written by an accepted trial pass: the
 * factor buffers describe the committed x. */
int ms_phase_set_factors_valid(ms_ctx *ctx, int valid);

/* ---- shard boundary exchange (multi-GPU; the reference is single-process) ----
 * A rank's boundary rows are the rows it owns that other ranks' tiles read as
 * halo.  One exchange = ms_pack_boundary -> all-gather of the fixed-size
 * messages (RCCL, done by the caller on the context's stream) ->
 * ms_unpack_boundary.  Message = [MS_NSCAL reduction scalars | boundary rows of
 * the listed per-vertex buffers]; unpack scatters every peer's rows into the
 * local buffers and returns all ranks' scalar headers (shard_count x MS_NSCAL)
 * for the rank-ordered host fold.
 * info = {max boundary rows over ranks, this rank's, this rank's halo rows, shard_count}. */
int ms_boundary_info(ms_ctx *ctx, int64_t info[4]);
size_t ms_exchange_bytes(ms_ctx *ctx, int n_buffers, const int *buffer_ids);
int ms_pack_boundary(ms_ctx *ctx, int n_buffers, const int *buffer_ids,
                     void *send_dev, size_t send_bytes);
int ms_unpack_boundary(ms_ctx *ctx, int n_buffers, const int *buffer_ids,
                       const void *recv_dev, size_t stride_bytes,
                       double *scal_all_host);

/* ---- library-side multi-GPU driver -------------------------------------------
 * The sharded step (the control flow of membrane_solver_amd/parallel.py's
 * ShardedStepper, i.e. ms_step with the exchanges of DESIGN.md section 6) run inside
 * the library: per exchange pack -> all-gather -> unpack -> mailbox poll, with no
 * interpreter in the loop.  The all-gather is RCCL's ncclAllGather on the
 * context's stream (ms_shard_comm_init; librccl is resolved from the process,
 * i.e. the copy torch.distributed already loaded) or a caller-supplied function
 * (tests use an in-process stand-in).  ms_shard_unique_id fills the 128-byte
 * ncclUniqueId on one rank; the caller broadcasts it to the others. */
typedef int (*ms_allgather_fn)(void *user, const void *send_dev, void *recv_dev,
                               size_t bytes_per_rank);
int ms_shard_unique_id(void *id128);
int ms_shard_comm_init(ms_ctx *ctx, const void *id128);
int ms_shard_set_allgather(ms_ctx *ctx, ms_allgather_fn fn, void *user);
/* Peer-to-peer exchange (no collective library in the step): every rank's pack kernel stores its scalar header and
 * boundary rows straight into EVERY peer's receive slab (xGMI stores through IPC-mapped device memory), a one-wave
 * kernel then raises this rank's flag word on every peer, and every block row of the unpack kernel waits -- bounded:
 * a peer that never arrives becomes an error after ~2 s, not a wave that never finishes -- for the flag word of the
 * rank whose message it unpacks.  Two slabs alternate (a peer can be at most one exchange ahead).
 *   ms_shard_peer_export: fills handles[0..127] with the hipIpcMemHandle_t of this rank's receive slab and of its flag
 *                          words (64 bytes each); the caller all-gathers the 128-byte records in rank order;
 *   ms_shard_peer_open:   opens every peer's two handles (world x 128 bytes, rank order; the own record is skipped);
 *   ms_shard_peer_set_pointers: the same for shard contexts of ONE process (thread shards): raw device pointers of
 *                          every rank's slab / flag words in rank order (ms_shard_peer_local gives a rank's own).
 * With peers set, ms_shard_step uses this exchange instead of ncclAllGather / the all-gather callback. */
int ms_shard_peer_export(ms_ctx *ctx, void *handles128);
int ms_shard_peer_open(ms_ctx *ctx, const void *handles_all);
int ms_shard_peer_local(ms_ctx *ctx, void **recv_slab, void **flag_words);
int ms_shard_peer_set_pointers(ms_ctx *ctx, void *const *recv_slabs, void *const *flag_words);
/* Shard contexts of ONE process share the device's few hardware queues: a rank's waiting wave can sit in front of the
 * very kernels of another rank it waits for.  Such contexts (tests) wait on the host instead: after raising its flags a
 * rank synchronises its stream and calls fn(user), which returns once every rank has done the same. */
typedef int (*ms_barrier_fn)(void *user);
int ms_shard_peer_set_barrier(ms_ctx *ctx, ms_barrier_fn fn, void *user);
int ms_shard_step(ms_ctx *ctx, const ms_stepper_params *params, double step_size,
                  double tol, ms_step_result *out);
/* number of exchanges done so far by ms_shard_step on this context */
int64_t ms_shard_exchange_count(const ms_ctx *ctx);
/* Peer-to-peer transport with in-kernel flag words: ms_shard_step also takes a trial's Armijo decision on the device
 * (from the scalar headers every rank has in its slab, added in rank order: the host's arithmetic), and queues what an
 * acceptance is followed by -- the commit x <- x + alpha d, the gradient + direction pass of the new point and that
 * pass's exchange -- behind it, gated on the decision word; the host replays the decision from the same headers
 * (a difference is an error) and the next ms_shard_step takes the pass's results instead of queueing it
 * (runtime/minimizer.py:1189-1535, line_search.py:386-392: no reference counterpart, same trajectory).
 * Behind the chain, in the CG steady state (the pass it queues yields a history direction that is no descent direction,
 * that step fails without a trial, the stepper is reset): the FIRST TRIAL of the search after it -- energy launch,
 * exchange, and a device-side decision that first tests from the direction exchange's headers whether that search
 * happens at all and forms its right-hand sides -- is queued as well and adopted by the step it belongs to.
 * MS_SHARD_CHAIN=0 switches all of it off; the trial queued ahead is OFF unless MS_SHARD_AHEAD=1 (at world 1 it gains
 * in the steady two-trial pattern what it loses on rejected trials).  stats: {chains queued, chains
 * that ran (main trial accepted), adopted by the next step, dropped (the next step wanted another pass), trials queued
 * ahead, adopted, dropped, 0}. */
int ms_shard_chain_stats(const ms_ctx *ctx, int64_t stats[8]);
/* ranks of the context's RCCL communicator as ncclCommCount reports them (0: no communicator) */
int ms_shard_comm_ranks(ms_ctx *ctx);
/* Kind of device memory the peer exchange's receive slabs and flag words live in: 0 uncached (MTYPE_UC, what the
 * collective library's own IPC signal buffers use), 1 fine-grained, 2 plain hipMalloc (coarse-grained: coherent across
 * GPUs at kernel boundaries only -- the fallback when the other kinds cannot be allocated or exported); -1 before
 * the slabs exist.  No reference counterpart (the reference is single-process). */
int ms_shard_peer_memory_kind(const ms_ctx *ctx);

/* Per-vertex state in caller-owned device memory (e.g. a torch tensor, so RCCL
 * collectives can run on it in place).  ms_state_bytes gives the size;
 * ms_rebind_state copies the current state there and uses it from then on.
 * The caller keeps the memory alive until ms_destroy. */
size_t ms_state_bytes(const ms_ctx *ctx);
int ms_rebind_state(ms_ctx *ctx, void *device_base, size_t bytes);

/* device pointer + geometry of a per-vertex buffer, for RCCL collectives on it */
int ms_device_buffer(ms_ctx *ctx, int buffer, void **dev_ptr, size_t *bytes);
/* rows = padded vertex rows (nvp); this shard owns rows [row0,row1) */
int ms_shard_info(ms_ctx *ctx, int64_t *nvp, int64_t *row0, int64_t *row1,
                  int64_t *rows_per_shard);
/* tiling statistics: n_tiles, facet instances (with halo duplicates), max halo */
int ms_tile_stats(ms_ctx *ctx, int64_t *n_tiles, int64_t *facet_instances,
                  int64_t *max_halo, int64_t *lds_bytes_energy,
                  int64_t *lds_bytes_gradient);

/* Per-kernel device timing with HIP events on the context's stream.  While
 * enabled every energy / gradient / direction / reduce launch is bracketed by
 * an event pair; ms_profile_read synchronises, returns the summed milliseconds
 * and launch counts per kind {0 energy, 1 gradient, 2 direction, 3 reduce,
 * 4 tilt, 5 bending_tilt facet pass, 6 tilt vector ops, 7 energy pair launch (two
 * trial evaluations of one line search in one launch), 8 energy triple launch
 * (three), 9 gradient, lean instantiation (k_gradient<1,false,256,0,true,true>: analytic
 * bending, uniform surface tension, no separate previous-direction rows)} and resets the
 * counters.  Used by bench.py for the roofline figure. */
#define MS_PROF_KINDS 13 /* 10: energy launch with four to eight trial evaluations (k_energy<...,8>);
                          * 11: tilt smoothness pass (k_tsmooth); 12: tilt search pass (k_tsearch: several step
                          * sizes of a relaxation's backtracking ladder, every tilt module, one launch) */
int ms_profile_enable(ms_ctx *ctx, int on);
int ms_profile_read(ms_ctx *ctx, double total_ms[MS_PROF_KINDS],
                    int64_t launches[MS_PROF_KINDS]);

/* Line-search queue statistics since ms_create (ms_step): stats[0] rounds queued, [1] multi-trial energy
 * launches among them, [2] trial evaluations of those launches that lay behind the accepted trial (wasted),
 * [3] searches whose accepted trial was an energy-only early trial of a multi-trial launch (re-evaluated alone),
 * [4] decisions on which host and device differed (each one also fails the call with MS_ERR_STATE: the device takes
 * every Armijo decision once, in the fold that closes the stage -- runtime/steppers/line_search.py:386-392 -- and the
 * host replays it from the same doubles), [5] rounds queued for a step that had not started yet (behind the running
 * gradient pass; ms_minimize only), [6] those the step adopted, [7] those dropped. */
int ms_queue_stats(ms_ctx *ctx, int64_t stats[8]);
/* One-tile meshes (every mesh the reference's own benchmarks and decks use: <= 256 vertices): the library records its
 * kernel launches and runs them pack by pack in ONE workgroup (k_exec) instead of launching each -- same device code,
 * same results bit for bit, a few launches per step instead of dozens.  MS_EXEC=0 in the environment switches it off.
 * Tilt relaxations (ms_relax_tilts / ms_relax_leaflet_tilts) then run as ONE launch: the command lists of the loop are
 * captured once and the reference's control flow (tilt_relaxation.py:303-421, 885-1230) runs in that workgroup
 * (MS_EXEC_RELAX=0: host-driven loop over packs).
 * stats: {active now, packs launched, launches recorded, bit 0 wanted (inactive while profiling) | relaxation
 * programs run << 8}.  No reference counterpart. */
int ms_exec_stats(ms_ctx *ctx, int64_t stats[4]);
/* Meshes whose tiles all fit on the chip at once (a few hundred tiles): ms_minimize runs the steps of the surface
 * (+ volume constraint row) / gradient-descent lane in ONE launch per call -- one workgroup per tile keeps its tile in
 * LDS across steps, grid barriers between the phases of a step, every workgroup folds the partials itself
 * (runtime/minimizer.py:1189-1535, runtime/steppers/line_search.py:267-426 for that lane).  A step the kernel cannot
 * take (guard range, exhausted search, drift projection, convergence) goes through the ordinary path.
 * MS_RESIDENT=0 switches it off.  stats: {co-residency (-1 not asked, 0 no, 1 yes), launches, steps taken, steps
 * declined}.  No reference counterpart. */
int ms_resident_stats(ms_ctx *ctx, int64_t stats[4]);
/* Tilt relaxations of multi-tile meshes: the backtracking search (runtime/steppers/tilt_relaxation.py:326-347,
 * 380-398, 918-973, 1150-1230: up to twelve halvings of the step on E(P(t + step*src)), positions frozen) runs as
 * search passes -- one launch evaluates several step sizes of the ladder for every tilt-reading module of both
 * leaflets (k_tsearch) -- with the iteration and evaluation counts of the sequential loop.  Module sets the pass
 * does not cover (disk targets, a lone tilt module on the single field, consistent mass there) and MS_TSEARCH=0 keep
 * one launch per module, field and trial.  stats: {search passes launched, step sizes evaluated by them}. */
int ms_tsearch_stats(ms_ctx *ctx, int64_t stats[2]);
/* Diagnostic of the same interpreter: on != 0 arms a device buffer to which every record run appends its duration
 * (s_memrealtime); a call also returns what has accumulated since the last one, per (kind, mode) pair: rows of
 * {kind, mode | instance << 16, count, total microseconds} (max_rows rows of 4 doubles; NULL: just arm / disarm). */
int ms_exec_trace(ms_ctx *ctx, int on, double *rows, int max_rows, int *n_rows);

/* Host-only planning pass (no GPU needed): runs the same tiling ms_create
 * uses and reports stats[0..7] = {n_tiles, facet_instances, max_halo,
 * max_tile_facets, dropped_facets, owner_instances, owned_corner_slots,
 * lds_bytes_gradient}.  perm_out (nv, patch row -> caller row) may be NULL.
 * Invariants: owner_instances == nf - dropped, owned_corner_slots == 3*that. */
int ms_plan_tiling(int nv, int nf, const double *positions, const int32_t *tri,
                   int tile_vertices, int shard_count, int64_t stats[8],
                   int32_t *perm_out);
/* Host-only: LDS bank model of the lane order inside the tiles (no GPU needed).
 * model[0] = mean LDS cycles of one corner gather (8-byte reads: lane groups of 32
 * over 64 four-byte banks, identical addresses broadcast; 1.0 = conflict-free),
 * model[1] = the same for one per-corner accumulator update (ds_add_f64 on owned
 * corners: lane groups of 16 over 32 banks, identical addresses serialise),
 * model[2], model[3] = fraction of lane groups that are conflict-free for each. */
int ms_plan_tiling_conflicts(int nv, int nf, const double *positions,
                             const int32_t *tri, int tile_vertices, double model[4]);

/* ---- kernel-provider seam: the five procedures under fortran_kernels/ ----
 * Host arrays in, host arrays out (H2D/D2H per call: for parity, not speed).
 * Semantics follow the Fortran: see fortran_kernels/loader.py:15-20 KernelSpec.
 */
/* surface_energy.f90:27-99 -- grad (nv*3) is ACCUMULATED into */
int ms_surface_energy_and_gradient_host(int nv, int nf, const double *pos,
                                        const int32_t *tri, const double *gamma,
                                        double *grad, double *energy);
/* bending_kernels.f90:32-74 */
int ms_grad_cotan_batch_host(int n, const double *u, const double *v,
                             double *grad_u, double *grad_v);
/* bending_kernels.f90:87-131 -- out is overwritten (zeroed first) */
int ms_apply_beltrami_laplacian_host(int dim, int nv, int nf,
                                     const double *weights, const int32_t *tri,
                                     const double *field, double *out);
/* tilt_kernels.f90:26-86 */
int ms_p1_triangle_divergence_host(int nv, int nf, const double *pos,
                                   const double *tilts, const int32_t *tri,
                                   double *div_tri, double *area, double *g0,
                                   double *g1, double *g2);
/* tilt_kernels.f90:88-190 -- va0..va2 may be NULL */
int ms_compute_curvature_data_host(int nv, int nf, const double *pos,
                                   const int32_t *tri, double *k_vecs,
                                   double *vertex_areas, double *weights,
                                   double *va0, double *va1, double *va2);

#ifdef __cplusplus
}
#endif
#endif /* MEMBRANE_HIP_H */
