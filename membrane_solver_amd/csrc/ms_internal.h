// Internal declarations shared by the host API, the tile builder and the
// gfx950 kernels of libmembrane_hip.so.  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "membrane_hip.h"

namespace ms {

// Switches of finished experiments (MS_NO_FAST, MS_TILE_ORDER, MS_FACET_ORDER, MS_KA_LDS_MIN) exist only in variant
// builds (tools/build_variant.sh passes -DMS_VARIANT_ENV): the shipped library does not read them.
inline const char* variant_env(const char* name) {
#ifdef MS_VARIANT_ENV
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// One tile-facet instance: local vertex slots inside the tile's LDS patch
// ([0,n_owned) owned, then halo) + flags.
struct TileFacet {
  uint16_t l0, l1, l2;
  uint16_t flags;  // bit0: this tile owns the facet's scalar terms (energy, volume,
                   //       min-edge, guard) ; bit1: facet belongs to the body
};
static_assert(sizeof(TileFacet) == 8, "TileFacet must be 8 bytes");

constexpr uint16_t TF_OWNER = 1;
constexpr uint16_t TF_BODY = 2;

// ---- device-side line-search decisions (ms_step's queue) -----------------------------------------------------------
// A fold (k_reduce) that closes a line-search stage takes the Armijo decision itself -- ONE workgroup, from the energies
// it has just folded and the right-hand sides the host supplied -- and publishes it as one word with an agent-scope
// (write-through) store.  Every kernel queued behind the stage reads that word with an agent-scope load and runs iff it
// holds the code the host queued it for; no workgroup re-derives a decision from doubles, so the workgroups of one
// launch cannot disagree.  The host replays the decision from the same doubles (mailbox) and compares codes.
constexpr int MS_MAX_TRIALS = 8;        // trials one energy launch can evaluate (alphas of one Armijo ladder)
constexpr int MS_P_RAN = MS_NSCAL;      // partial row: 1.0 from every workgroup of a GATED launch that ran, 0.0 from one that did not
constexpr int MS_NPART = MS_NSCAL + 1;  // rows of a partials array
constexpr int MS_MB_DEC = MS_NSCAL;     // mailbox entry of the fold's decision code
constexpr int MS_MB_WORDS = MS_NSCAL + 1;  // {value, sequence} entries of a mailbox
constexpr int MS_DEC_STRIDE = 64;       // uint32 words between two decision records (256 bytes: a GO record carries 8 right-hand sides and 8 handed-over parameters)
enum : uint32_t {
  DEC_NONE = 0,         // never written
  DEC_CONTINUE = 1,     // every trial so far was rejected: the next stage of the ladder runs
  DEC_ACCEPT_MAIN = 2,  // accepted, and the accepted trial's outputs are the ordinary ones (xt, fK, fA): the gradient pass runs
  DEC_ACCEPT_SIDE = 3,  // an EARLY trial of a multi-trial launch was accepted (energy-only evaluation): nothing queued runs
  DEC_GO = 4,           // (host side only: the merged fold's test passed)
  DEC_STOP = 5,         // merged fold: the search it was queued for does not happen (converged, the other kind of
                        // direction, guard range): nothing behind it runs
  DEC_ERR_RAN = 0x100   // flag: a gated launch ran on some of its workgroups only
};

constexpr uint8_t VF_FIXED = 1;
constexpr uint8_t VF_BOUNDARY = 2;
constexpr uint8_t VF_TILT_FIXED = 4;  // vertex.tilt_fixed (runtime/minimizer_helpers.py:49-75)
constexpr uint8_t VF_TILT_FIXED_IN = 8;    // vertex.tilt_fixed_in  (minimizer.py:460-472)
constexpr uint8_t VF_TILT_FIXED_OUT = 16;  // vertex.tilt_fixed_out (minimizer.py:474-486)

// Host-side result of the tiling pass.
struct Tiling {
  int nv = 0, nf = 0, T = 256;
  int own = 256;               // vertex rows a tile owns (<= T = threads per workgroup): tile t owns rows [t*own, (t+1)*own)
  int n_tiles = 0;             // tiles covering real vertices
  int n_tiles_padded = 0;      // multiple of shard_count
  int tiles_per_shard = 0;
  int64_t nvp = 0;             // padded vertex rows = n_tiles_padded * own
  std::vector<int32_t> perm;   // internal row -> external row   (nv)
  std::vector<int32_t> iperm;  // external row -> internal row   (nv)
  std::vector<int32_t> tile_facet_off;  // n_tiles+1
  std::vector<TileFacet> tile_facets;   // facet instances
  std::vector<int32_t> tile_facet_ext;  // external facet row of each instance
  std::vector<int32_t> tile_halo_off;   // n_tiles+1
  std::vector<int32_t> halo_ids;        // internal vertex ids
  // per-tile vertex -> corner CSR (deterministic gather order): for owned vertex i of
  // tile t the entries vent[tile_ent_off[t] + voff[t*(T+1)+i] ... +voff[t*(T+1)+i+1])
  // are (facet_local << 2 | corner), ascending in facet_local.
  std::vector<int32_t> tile_ent_off;    // n_tiles+1
  std::vector<uint16_t> tile_voff;      // n_tiles*(T+1)
  std::vector<uint16_t> vent;
  int max_ent = 0;
  int max_halo = 0;
  int max_tile_facets = 0;
  int64_t dropped_facets = 0;
};

// Builds the patch ordering + tiles.  Returns MS_OK or MS_ERR_*; `err` gets text.
int build_tiling(int nv, int nf, const double* positions, const int32_t* tri,
                 const uint8_t* body_facets, int T, int shard_count, Tiling& out,
                 std::string& err);

// ---- device-side view handed to kernels ---------------------------------
struct DeviceMesh {
  int nv, T, n_tiles;
  int own;                // vertex rows per tile (<= T = threads per workgroup)
  int has_boundary;       // any vertex carries VF_BOUNDARY (uniform fast-path switch)
  const int32_t* tile_facet_off;
  const TileFacet* tile_facets;
  // the same records in 4 bytes (10 bits per corner slot | 2 flag bits) for the T = 256 instances of the two tile
  // kernels; nullptr when some tile needs more than 1024 LDS slots (those instances are not used then)
  const uint32_t* tile_facets32;
  int no_fast;            // MS_NO_FAST at ms_create: never pick the T = 256 instances (A/B switch)
  const double* tf_gamma;  // surface tension per facet instance
  // uniform-parameter shortcuts: when every facet carries the same surface tension (every vertex the same kappa and
  // c0) the kernels take the value from here and the per-facet / per-vertex arrays are not read at all
  int gamma_uniform, kc_uniform;
  double gamma_const, kappa_const, c0_const;
  const int32_t* tile_halo_off;
  const int32_t* halo_ids;
  const int32_t* tile_ent_off;
  const uint16_t* tile_voff;
  const uint16_t* vent;
  const uint8_t* vflags;  // nvp
  const double* kappa;    // nvp
  const double* c0;       // nvp
};

struct EnergyArgs {
  DeviceMesh m;
  int tile0, tile1;       // this shard's tile range
  const double* x;        // base positions
  const double* d;        // direction or nullptr
  double alpha;
  double* xt;             // trial positions out or nullptr
  double* fK;             // (nvp,3) out or nullptr
  double* fA;             // (nvp,2) out or nullptr
  double* partials;       // [n_tiles][MS_NSCAL]
  int bending_model;
  uint32_t modules;
  // bending_tilt: per-vertex record {base = 2H - c0 (0 on boundary), A_eff, kappa*ratio*H, 0}
  // written instead of the final factors; fK then holds K_dir * kappa * ratio (k_bt finishes)
  double* bt_vert;
  // queued stage (nullptr: unconditional): run only if the decision word holds `gate_want` (see DEC_*)
  const uint32_t* gate;
  uint32_t gate_want;
  const double* bt_normals;  // leaflet bending_tilt: unit vertex normals of the evaluated positions (signed H, K_dir = n)
  // energy-only evaluation of BOTH leaflets' bending_tilt in one launch: the second leaflet's record from the same
  // curvature sums with its own (kappa, c0) (no factor outputs: fK / fA serve one leaflet at a time)
  double* bt_vert2 = nullptr;
  const double* kappa2 = nullptr;
  const double* c02 = nullptr;
  int atomic;             // accumulate per-vertex sums with LDS atomics (not bitwise reproducible)
  // multi-trial launch (pair = n >= 2): trials 0 .. n-2 of the ladder are evaluated in the same launch at alpha_side[j]
  // into partials_side[j] (energy only unless xt/fK/fA_side[j] are given: j < 2), trial n-1 at `alpha` with the
  // ordinary outputs
  int pair;
  double alpha_side[MS_MAX_TRIALS - 1];
  double* partials_side[MS_MAX_TRIALS - 1];
  double* xt_side[2];
  double* fK_side[2];
  double* fA_side[2];
};

struct GradientArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  const double* fK;
  const double* fA;
  double* g;
  double* gC;             // or nullptr
  double* partials;
  const double* scal;     // device scalars (volume for the penalty factor)
  uint32_t modules;
  int bending_grad_mode;
  double volume_stiffness, target_volume;
  int accumulate;         // add into existing g instead of overwriting
  // fused direction pass (only when no constraint row has to be projected out first)
  int dir_mode;           // 0 off, 1 d = -g (GD / CG restart), 2 per-row Polak-Ribiere with history
  double* d;
  const double* pg;
  const double* pd;
  // queued behind a line search (nullptr: unconditional): run only if the decision word holds `gate_want`
  const uint32_t* gate;
  uint32_t gate_want;
  int pd_neg_pg;           // the previous direction is -pg (an implicit steepest-descent step): derive, do not load
  int atomic;
  // leaflet bending_tilt (BENDMODE 3): per-corner fA_eff = 1/2 kappa_k (base_k + s div_f t)^2
  const double* bt_vert;   // (nvp,4): [0] = base
  const double* tilts;     // (nvp,3) the leaflet's tangent tilts
  double div_sign;
};

struct TiltArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  const double* d;        // direction or nullptr (energy at x + alpha d)
  double alpha;
  const double* tilts;    // (nvp,3) tilts read by every mode
  double* tilts_out;      // mode 2: projected tilts are written here (may alias `tilts`)
  double k_tilt;
  double* g;              // mode 1: shape gradient is ADDED into g
  double* tilt_grad;      // mode 1: k_t t_v A_v (written)
  double* minv;           // mode 3: Jacobi preconditioner 1/(k_t A_v) (normals go to tilts_out)
  double* partials;
  int e_slot;             // reduction slot of the energy partial (MS_S_ETILT / _IN / _OUT)
  int consistent;         // tilt_leaflet.py:101-114: consistent P1 mass coeff (energy / shape gradient)
  int tg_accumulate;      // mode 1: ADD the tilt gradient instead of writing it
  int cons_tilt_grad;     // mode 1 with `consistent`: the tilt gradient takes the consistent P1 mass too,
                          // k A_f/12 (2 t_k + t_a + t_b) per corner (tilt_leaflet.py:124-150), instead of k t_v A_v
  double* va_out;         // mode 3: barycentric vertex areas (nvp) or nullptr
  double* fields;         // mode 5: curvature fields, four planes of (fields_rows, 3) -- see k_tilt
  int64_t fields_rows;
  // mode 3, the set-up of a relaxation in the same launch: proj_out receives P(tilts) of the owned rows (k_tvec mode 2
  // with step 0, on the normals just formed); finish_minv: `minv` receives the clamped INVERSE of the diagonal (k_tvec
  // mode 3: 1 where the diagonal is <= 1e-12 and on rows clamped by fixed_bit) -- when no smoothness term is to be
  // added to the diagonal first
  double* proj_out = nullptr;
  int finish_minv = 0;
  int fixed_bit = 0;
  // mode 3 also as the projection pass of up to two more fields (mode 2's arithmetic on the same normals):
  // fld_out[k] <- fld_in[k] - (fld_in[k].n) n on the owned rows -- the leaflet trial projections of an energy evaluation
  // that needs the normals anyway (bending_tilt_in/out)
  const double* fld_in[2] = {nullptr, nullptr};
  double* fld_out[2] = {nullptr, nullptr};
  // mode 3 as the set-up of a TWO-field relaxation in one launch (the normals and the vertex areas are the same): the
  // second field's projection, diagonal and area outputs
  const double* tilts_b = nullptr;
  double* proj_out_b = nullptr;
  double* minv_b = nullptr;
  double* va_out_b = nullptr;
  double k_tilt_b = 0.0;
  int finish_minv_b = 0;
  int fixed_bit_b = 0;
};

struct DiskTargetArgs {   // tilt_disk_target_in.py:160-286
  int tile0, tile1, nv, T, n_tiles;
  const uint8_t* vflags;
  const uint8_t* disk;    // (nvp) 1 on the tagged rows
  const double* x;
  const double* d;        // direction or nullptr (positions x + alpha d)
  double alpha;
  const double* tilts;
  double* diff;           // mode 1 / 3 out: t - theta(r) r_hat on the disk rows, 0 elsewhere
  double* target;         // mode 2 out / mode 3 in: theta(r) r_hat of the tagged rows at FROZEN positions (tilt
                          // relaxations evaluate the profile dozens of times on the same surface), NaN = profile off
  double theta_b, lambda, radius;  // radius <= 0: read scal[r_slot]
  double center[3], normal[3];
  const double* scal;
  double* partials;
  int r_slot;
};
hipError_t launch_disk_target(const DiskTargetArgs& a, int mode, hipStream_t s);

struct BtArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  const double* d;         // direction or nullptr (energy at x + alpha d)
  double alpha;
  const double* tilts;     // (nvp,3) tangent tilts
  const double* bt_vert;   // (nvp,4) from the energy pass at the same positions
  double* fK;              // mode 1: in K_dir*kappa*ratio, out the back-prop factor
  double* fA;              // mode 1: out {fA_eff, fA_vor}
  double* tilt_grad;       // mode 2: dE/dt ADDED here
  double* partials;
  double* g;               // mode 3: shape gradient of the divergence term ADDED here
  double div_sign;         // -1 inner leaflet, +1 outer leaflet / single field
  int e_slot;              // reduction slot of the energy partial (MS_S_EBT / _IN / _OUT)
  // mode 0 only: also the tilt-magnitude energy of the same field (modules/energy/tilt.py:99-172, per-facet form) from
  // the rows this kernel has staged anyway -- k_tilt's energy-only launch is not needed then.  0 = off.
  double k_tilt_fused;
  int e_tilt_slot;
};

struct TsArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  const double* d;         // direction or nullptr (cotans at x + alpha d)
  double alpha;
  const double* tilts;     // (nvp,3)
  double k_smooth;
  double* tilt_grad;       // mode 1: dE/dt ADDED here
  double* diag;            // mode 2: Jacobi diagonal 1/2 k_s sum (c_a + c_b) ADDED here
  double* partials;
  int e_slot;              // reduction slot of the energy partial (MS_S_ETS / _IN / _OUT)
};

// stream kernels (one workgroup of BLOCK threads per tile): their arguments as structs, so that the same device body
// serves the ordinary launch and the one-workgroup interpreter (k_exec)
struct TvecArgs {
  int mode, tile0, nv, T;
  const uint8_t* vflags;
  double* tg;
  const double* minv;
  double* dir;
  const double* tilts;
  const double* src;
  const double* normals;
  double* out;
  double coef;
  const double* coef_dev;  // device programs (k_exec): the coefficient is read from here instead (nullptr: `coef`)
  int flag;
  double* partials;
  int n_tiles;
  uint8_t fixed_bit;
  int s_gn2, s_rz;
};
struct DirectionArgs {
  int tile0, nv, T;
  const uint8_t* vflags;
  double* g;
  const double* gC;
  double* d;
  const double* pg;
  const double* pd;
  const double* scal;
  int use_constraint, cg_history;
  double* partials;
  int n_tiles, write_g;
  const uint32_t* gate;
  uint32_t gate_want;
  int pd_neg_pg, precond;
};
struct RowDotArgs {
  int tile0, nv, T;
  const double* g;
  const double* gC;
  double* partials;
  int n_tiles;
};
struct AxpyMaskedArgs {
  int64_t n_rows;
  const uint8_t* vflags;
  double* x;
  const double* y;
  double coef;
};

// ---- the one-workgroup interpreter (k_exec) ------------------------------------------------------------------------
// When a context's mesh is ONE tile, every kernel of the library is one workgroup (folds: a handful), a step is dozens of
// launches of a few microseconds of work each, and what the step costs is launches and host round trips.  Such a
// context attaches an ExecRecorder to its stream: the launchers then append {kind, instance, arguments} records to a
// pack instead of launching, and the pack is launched as ONE kernel -- k_exec, one workgroup, which runs the recorded
// device bodies (the same code as the ordinary kernels) in order, a workgroup barrier between them -- when the host
// needs the device to make progress (it is about to wait for a mailbox, or touches the stream directly).  The pack
// travels as the kernel's argument block.  Results are bit for bit those of the launch-per-kernel path.
enum : uint16_t {
  CK_ENERGY = 1, CK_GRADIENT, CK_TILT, CK_BT, CK_TS, CK_TVEC, CK_DISK, CK_REDUCE, CK_DIRECTION, CK_ROWDOT,
  CK_AXPY_MASKED, CK_MEMSET, CK_RELAX, CK_RELAX_FUSED, CK_TSEARCH  // (CK_TSEARCH: the single-step-size form of k_tsearch, mode = fields)
};
struct ExecCmdHead {
  uint16_t kind;
  uint16_t bytes;     // of the whole record (head + arguments), a multiple of 8
  int32_t mode;       // template MODE of the tilt-family kernels, launch mode of the stream kernels
  int32_t cap, max_ent;
  int32_t grid;       // blocks of the ordinary launch: the body runs for block 0 .. grid-1 in turn
  uint32_t inst;      // template instance (energy / gradient: see exec_energy_inst / exec_gradient_inst)
  int32_t mesh;       // which DeviceMesh of the pack head the tile kernels' arguments go with
  int32_t pad;
};
static_assert(sizeof(ExecCmdHead) == 32, "ExecCmdHead is 32 bytes");
constexpr int EXEC_MESHES = 3;       // distinct DeviceMesh values per pack (plain, inner leaflet, outer leaflet)
constexpr int EXEC_PACK_BYTES = 3968;  // kernel-argument block of one k_exec launch (HIP allows 4096)
struct ExecPackHead {
  int32_t n_cmds, bytes;
  int32_t T, pad;
  // diagnostic (ms_exec_trace): when set, thread 0 appends {kind << 32 | mode, duration in 10 ns ticks} per record run;
  // word 0 of the buffer counts the entries
  unsigned long long* stamps;
  DeviceMesh m[EXEC_MESHES];
};
constexpr int EXEC_STAMP_CAP = 1 << 16;
struct ExecMemsetArgs {
  double* p;
  int64_t n;  // doubles
};
// CK_RELAX: the tilt relaxation (TiltRelaxationManager.relax_tilts / relax_leaflet_tilts,
// runtime/steppers/tilt_relaxation.py:237-424, 426-1478) as a PROGRAM: the record is followed by command lists the
// host captured once -- "gradient at the current tilts", "trial field + its energy", "first direction", "next
// direction", each for both parities of the tilts <-> trial pointer swap an acceptance is -- and the workgroup walks
// the reference's control flow itself (Fletcher-Reeves CG or gradient descent, up to 12 halvings, accept on E1 <= E0),
// reading the folded scalars where the host would read its mailbox.  No launch and no host round trip inside a
// relaxation; the host gets {iterations, evaluations, final parity} in one mailbox post.
struct ExecRelaxHead {
  int32_t solver, max_iters, nf, n_e;
  double step_size, tol;
  int32_t e_slot[16];               // energy slots in the host's order of addition (tilt_energy_from_mailbox)
  int32_t s_gn2[2], s_rz[2];
  const double* scal;               // device scalars the lists' folds write
  double* cells;                    // [0] the trial coefficient (sign * step), [1] the Fletcher-Reeves beta
  unsigned long long* host_box;     // pinned result mailbox: entries 0..3 = iterations, evaluations, parity, status
  unsigned long long ticket;
  int32_t off_grad[2], n_grad[2];   // list offsets (bytes from the pack base) and lengths (records)
  int32_t off_trial[2], n_trial[2];
  int32_t off_dir0, n_dir0, off_dir1, n_dir1;
  int32_t total_bytes, pad;         // record + lists (the interpreter's main loop skips them)
};
// CK_RELAX_FUSED: a leaflet relaxation with the frozen geometry held on the CU (ms_relax_fused.inc)
struct ExecRelaxFusedField {
  double* tilts;              // in: the projected start field; out: the relaxed field
  const double* minv;         // Jacobi M^-1 (1 on clamped rows)
  const double* va;           // barycentric vertex areas (tilt_in / tilt_out in the vertex-area form)
  const double* bt_vert;      // (nvp,4): [0] = base_k = 2H - c0 of this leaflet
  const double* kappa;
  const double* dt_target;    // theta(r) r_hat of the tagged rows (NaN: profile off)
  const uint8_t* disk;
  double k_tilt, k_smooth, dt_strength, div_sign;
  int32_t has_tilt, has_bt, has_ts, has_dt;
  uint32_t fixed_bit, pad;
};
struct ExecRelaxFusedArgs {
  DeviceMesh m;
  const double* x;
  const double* normals;      // unit vertex normals of the frozen surface
  ExecRelaxFusedField f[2];
  int32_t nf, solver, max_iters, cap, max_ent, pad;
  double step_size, tol;
  unsigned long long* host_box;  // {iterations, evaluations, parity (0), done}
  unsigned long long ticket;
};
size_t relax_fused_lds_bytes(int cap, int max_ent, int max_tile_facets);

struct ExecRecorder {
  hipStream_t stream = nullptr;
  int T = 0;
  std::vector<unsigned char> buf;  // ExecPackHead + records
  int n_mesh = 0;
  size_t lds = 0;                  // dynamic LDS the recorded bodies need (max)
  long launches = 0, cmds = 0;     // statistics
  bool capture = false;            // records are collected for a program (CK_RELAX): no size limit, never launched
  unsigned long long* d_stamps = nullptr;  // ms_exec_trace: per-record durations go here
  // args/bytes: the kernel's argument struct; tile kernels (mesh != nullptr) are stored without their leading DeviceMesh
  hipError_t push(uint16_t kind, int mode, int cap, int max_ent, int grid, uint32_t inst, size_t lds_bytes,
                  const void* args, size_t bytes, const DeviceMesh* mesh = nullptr);
  hipError_t flush();
  // launch `pack` (ExecPackHead + records, any size up to the largest k_exec variant) as it is
  hipError_t launch_pack(const std::vector<unsigned char>& pack, size_t lds_bytes);
};
constexpr int EXEC_PACK_MID = 12160, EXEC_PACK_BIG = 28544;  // larger argument blocks for programs (CK_RELAX)
ExecRecorder* exec_find(hipStream_t s);   // the recorder attached to a stream, or nullptr
void exec_attach(ExecRecorder* r);
void exec_detach(ExecRecorder* r);
// launchers that are not recorded flush first (stream order)
inline hipError_t exec_sync(hipStream_t s) {
  ExecRecorder* r = exec_find(s);
  return r ? r->flush() : hipSuccess;
}

// the resident step kernel (ms_resident.inc): many minimizer steps of the surface (+ volume row) / gradient-descent lane
// in one launch, one workgroup per tile, grid barriers between the phases
struct ResidentArgs {
  DeviceMesh m;
  double* x;               // (nvp,3) positions: read at entry, owned rows written back at exit
  double* d;               // (unused)
  double* g;               // (nvp,3) raw gradient rows of the last evaluation (tile-boundary rows are read by the
  double* gC;              //         neighbours); (nvp,3) constraint-row rows, or nullptr
  double* partials;        // [2][2][MS_NPART][n_tiles]: consecutive phases alternate; two trials per search phase
  unsigned int* bar;       // barrier words (zeroed by the host before the launch)
  double* log;             // [n_steps][8] step log rows (ms_minimize's layout), written by workgroup 0
  double* result;          // [16]: steps done, reason, step size, energy, volume, min_edge^2, trials, barriers,
                           //       [8..13] where workgroup 0 spent its time (us, see k_resident)
  int n_steps, volrow, want_vol, atomic, max_iter;
  double step_size, tol, c1, beta, gamma, alpha_max_factor;
  int drift_check;
  double target_volume, volume_tolerance;
  int cap, max_ent;
};
enum : int {
  RES_DONE = 0,        // all requested steps taken
  RES_CONVERGED = 1,   // |g| < tol at the top of step `steps done` (no step taken there)
  RES_BAIL = 2,        // step `steps done` needs the ordinary path (guard range, non-descent, exhausted search)
  RES_DRIFT = 3,       // step `steps done - 1` was accepted and the volume drifted past the tolerance
  RES_TIMEOUT = 4      // a grid barrier timed out (never expected: the host re-runs from the last consistent x)
};
constexpr int RESIDENT_BAR_WORDS = 11 * 32;
size_t resident_lds_bytes(int cap, int max_ent, int max_tile_facets, bool atomic);
hipError_t resident_fits(int n_tiles, size_t lds, int device, int* resident);
hipError_t launch_resident(const ResidentArgs& a, size_t lds, hipStream_t s);

// kernel launchers (ms_kernels.hip).  cap = T + max halo (LDS patch slots),
// max_ent = largest per-tile vertex->corner entry count.
size_t energy_lds_bytes(int T, int cap, int max_ent, bool bend, bool guard, bool flags, bool atomic = false);
size_t gradient_lds_bytes(int T, int cap, int max_ent, bool bend, bool volrow, bool atomic = false, bool leaf = false);
hipError_t launch_energy(const EnergyArgs& a, bool guard, int cap, int max_ent, hipStream_t s);
hipError_t launch_gradient(const GradientArgs& a, int cap, int max_ent, hipStream_t s);
bool gradient_lean_instance(const GradientArgs& a);  // which k_gradient instantiation launch_gradient picks
// mode 0: energy partial (MS_S_ETILT); 1: energy + gradients; 2: project tilts to tangent
size_t tilt_lds_bytes(int T, int cap, int max_ent, bool consistent = false, int mode = 1);
hipError_t launch_tilt(const TiltArgs& a, int mode, int cap, int max_ent, hipStream_t s);
// bending_tilt facet pass.  mode 0: energy (MS_S_EBT); 1: energy + back-prop factors;
// 2: energy + tilt gradient
size_t bt_lds_bytes(int T, int cap, int max_ent, int mode = 1);
hipError_t launch_bt(const BtArgs& a, int mode, int cap, int max_ent, hipStream_t s);
// tilt smoothness (Dirichlet) pass.  mode 0: energy (MS_S_ETS); 1: energy + tilt gradient; 2: Jacobi diagonal
size_t ts_lds_bytes(int T, int cap, int max_ent);
hipError_t launch_ts(const TsArgs& a, int mode, int cap, int max_ent, hipStream_t s);
hipError_t launch_tvec(int mode, int tile0, int tile1, int nv, int T, const uint8_t* vflags, double* tg,
                       const double* minv, double* dir, const double* tilts, const double* src,
                       const double* normals, double* out, double coef, int flag, double* partials,
                       int n_tiles, hipStream_t s, uint8_t fixed_bit = VF_TILT_FIXED,
                       int s_gn2 = MS_S_TGNORM2, int s_rz = MS_S_TRZ);
// Tilt search pass (ms_tsearch.inc): with the positions frozen, the energies of the tilt-reading modules of up to two
// fields at up to MS_MAX_TRIALS step sizes of ONE backtracking ladder in ONE facet pass -- the trial rows
// P(t + coef_j src) (k_tvec mode 2) are formed where they are staged, never written; trial j's per-tile partials go to
// partials[j] in the slots the single-trial kernels use (k_bt<0>, k_tsmooth<0>, k_tvec mode 4), by the same sums.
struct TsearchField {
  const double* tilts;     // (nvp,3) current tangent tilts
  const double* src;       // (nvp,3) search direction rows (the gradient or the CG direction)
  const double* bt_vert;   // (nvp,4) record of the energy pass at these positions (bending_tilt on)
  const double* kappa;     // (nvp) bending rigidity of this field
  const double* va;        // (nvp) barycentric vertex areas (tilt_form 2)
  double div_sign;         // -1 inner leaflet, +1 otherwise
  double k_tilt;           // tilt modulus
  double k_smooth;         // smoothness rigidity
  int tilt_form;           // 0: module off; 1: per-facet form (k_bt<0>'s fused term / k_tilt<0>); 2: vertex-area form
  int s_ebt, s_etilt, s_ets;  // reduction slots; s_ebt / s_ets < 0: module off
  int fixed_bit;           // vertex flag bit that clamps a row of this field
  double* trial_out;       // single-trial pass: the trial rows of the owned vertices are written here (or nullptr)
};
struct TsearchArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  const double* normals;   // (nvp,3) frozen unit vertex normals
  int n_fields, n_trials;
  // the same pass as the tilt-module energy evaluation of a SHAPE trial (one step size, plain): positions x + alpha d
  // (d != nullptr), the fields' rows taken as they are (already projected onto that surface), tilt_form 1
  const double* d;
  double alpha;
  int plain;
  TsearchField f[2];
  double coef[MS_MAX_TRIALS];       // sign * step_j
  double* partials[MS_MAX_TRIALS];  // partial set of trial j
};
size_t tsearch_lds_bytes(int cap, int n_fields, int n_trials);
hipError_t launch_tsearch(const TsearchArgs& a, int cap, hipStream_t s);

// Tilt gradient pass (ms_tsearch.inc): energies and tilt gradients of every tilt-reading module of up to two fields at the
// current tilts, positions frozen, then k_tvec mode 0 on the finished rows -- one launch per evaluation of a relaxation.
struct TgradField {
  const double* tilts;
  const double* bt_vert;
  const double* kappa;
  const double* va;        // tilt_form 2
  const double* minv;      // Jacobi M^-1 (the <r, M^-1 r> partial)
  double* grad;            // (nvp,3) out: dE/dt, clamped rows zero
  double div_sign, k_tilt, k_smooth;
  int tilt_form;           // 0 off; 1 per-facet energy, gradient k t_v A_v with the gathered barycentric area (k_tilt<1>); 2 vertex-area form (k_tvec mode 4)
  int s_ebt, s_etilt, s_ets, s_gn2, s_rz;
  int fixed_bit;
};
struct TgradArgs {
  DeviceMesh m;
  int tile0, tile1;
  const double* x;
  int n_fields;
  TgradField f[2];
  double* partials;
};
size_t tgrad_lds_bytes(int T, int cap, int max_ent, int n_fields, bool atomic, bool need_a3);
// combine: a corner's module terms are added and accumulated per vertex by LDS atomics (default mode); false: staged and gathered module by module, the per-module launches' sums
hipError_t launch_tgrad(const TgradArgs& a, int cap, int max_ent, bool combine, hipStream_t s);

hipError_t launch_tvec2(const TvecArgs& a, const TvecArgs& b, int n_blocks, hipStream_t s);
// One fold launch.  set[]: the trials of a multi-trial launch in TRIAL order (the last one = the ordinary outputs); a
// plain fold has one set.
struct FoldSet {
  const double* partials;
  double* scal;
  unsigned long long* host_box;  // pinned mailbox {value, sequence} entries or nullptr
};
struct FoldArgs {
  int n_tiles, tile0, tile1;
  uint32_t slot_mask;
  unsigned long long ticket;
  int n_sets;
  FoldSet set[MS_MAX_TRIALS];
  const uint32_t* gate;   // fold only if *gate == gate_want (nullptr: always)
  uint32_t gate_want;
  int check_ran;          // the tile kernel feeding this fold was gated by the same word: its MS_P_RAN partials must add
                          // up to every tile (gate open) or to none (closed)
  uint32_t* dec_out;      // Armijo decision of this stage (nullptr: none)
  uint32_t* counter;      // arrival counter of the stage's energy workgroups (zero between launches)
  uint32_t e_mask;        // slots whose sum is a trial's energy (ESURF | EBEND as the module set has them)
  double rhs[MS_MAX_TRIALS];  // energy0 + c alpha_j <g,d>, trial order
  // Armijo right-hand sides formed on the device (nullptr: the host's rhs[] above): rhs_dev[j], written by the direction
  // fold that opened this round (go_out below)
  const double* rhs_dev;
  // MERGED fold (go_kind != 0): this launch also folds the direction scalars of the gradient pass that ran in front of
  // the energy launch (which did not wait for them), and the workgroup that arrives last first tests whether the search
  // the host queued the launch for happens at all -- go_kind 1: the direction with history is no descent direction
  // (<g,d> >= 0), the stepper restarts with d = -g; 2: the direction just written is a descent direction -- is not
  // converged (|g|^2 > go_tol2) and stays in the unguarded range (max|d_i|^2 < go_lim).  If not: DEC_STOP.  If so the
  // Armijo right-hand sides are formed here, rhs_j = go_e0 + (go_c alpha_j) <g,d>, alpha_j = alpha_{j-1} go_beta -- the
  // host's expressions with the host's roundings -- the trials are decided, and the right-hand sides of the ladder's
  // later alphas are left in go_out[2..] (doubles) for the gated stages behind this launch.
  int go_kind;
  double go_tol2, go_lim, go_e0, go_c, go_alpha0, go_beta;
  uint32_t* go_out;
  unsigned long long* host_err;  // pinned word: set non-zero when check_ran fails
  int side_full;          // fold every slot of the early trials' sets too (they have outputs of their own and can be
                          // accepted as they are); otherwise only their energy slots, by the head workgroup
};
hipError_t launch_reduce(const FoldArgs& a, hipStream_t s);

// the Armijo decision of a sharded trial from the scalar headers in this rank's slab (k_shard_decide)
struct ShardDecideArgs {
  const double* recv;      // slab of this exchange: world x stride doubles, rank r's header at recv + r * stride
  size_t stride;
  int world;
  int has_alt, alt_off;    // pair launch: trial 0's slots sit alt_off further in the header
  int slot_a, slot_b;      // energy slots whose rank-ordered sums add up to the trial energy (-1: module off)
  double rhs_alt, rhs_main;
  uint32_t* dec_out;
  unsigned long long* post;  // pinned {value, value XOR ticket} pair for the host's replay (or nullptr)
  unsigned long long ticket;
  const uint32_t* gate;    // the trial itself was gated on this word (nullptr: it always ran); closed: DEC_STOP
  uint32_t gate_want;
  // go_kind 1: the trial was queued AHEAD of its step -- the search of a steepest-descent restart behind a direction
  // with history that is expected to be no descent direction.  Whether it happens (<g,d> >= 0 of that direction, not
  // converged, unguarded range) and its right-hand sides energy0 + c alpha (-|g|^2) are formed here, from the headers
  // of the direction exchange (recv_dir) and the accepted trial's energy and min edge (keep_in): DEC_STOP if it does not
  int go_kind;
  const double* keep_in;   // {energy, min edge^2} of the accepted trial, left by the launch that decided it (keep_out)
  double* keep_out;        // this launch's main trial: the same two doubles for a trial queued ahead behind ITS chain
  const double* recv_dir;
  double tol, c1, alpha_main, alpha_alt;
  int has_faces;
};
hipError_t launch_pack_boundary(const int32_t* rows, int n_rows, const double* const* bufs,
                                const int* ncomp, int n_bufs, const double* scal, double* send,
                                hipStream_t s);
hipError_t launch_unpack_boundary(const int32_t* rows_all, const int32_t* row_off, int me, int world,
                                  int max_rows, double* const* bufs, const int* ncomp, int n_bufs,
                                  const double* recv, size_t stride, double* scal_all, hipStream_t s,
                                  unsigned long long* host_seq = nullptr, unsigned long long ticket = 0,
                                  bool remote_written = false, const unsigned long long* wait_flags = nullptr,
                                  unsigned long long wait_ticket = 0, unsigned long long* host_err = nullptr,
                                  const uint32_t* gate = nullptr, uint32_t gate_want = 0,
                                  // the trial decision this exchange feeds, taken by the header block that arrives last
                                  // (dec_arrived: a zeroed counter); nullptr: none
                                  const ShardDecideArgs* decide = nullptr, unsigned int* dec_arrived = nullptr);
hipError_t launch_post_seq(unsigned long long* host_seq, unsigned long long ticket, hipStream_t s);
// profiling only: *out = (*gate == want), one lane, queued right behind a gated launch (ms_profile_*)
hipError_t launch_gate_probe(const uint32_t* gate, uint32_t want, uint32_t* out, hipStream_t s);
// peer-to-peer exchange: pack straight into every peer's slab (dst[r] = that peer's slot for this rank), raise this
// rank's flag on every peer; the unpack kernel waits (bounded) for every peer's flag here
struct PeerFlags {
  unsigned long long* p[16];  // this exchange parity's flag rows of the peers (device table: ms_ctx::d_peer_flagtab)
};
// d_flags != nullptr: the last block to finish for a peer raises this rank's word there (no flag kernel); d_arrived: 16
// zeroed counters
hipError_t launch_pack_peers(const int32_t* rows, int n_rows, const double* const* bufs, const int* ncomp, int n_bufs,
                             const double* scal, double* const* dst, int world, hipStream_t s,
                             const PeerFlags* d_flags = nullptr, unsigned int* d_arrived = nullptr, int me = 0,
                             unsigned long long ticket = 0, const uint32_t* gate = nullptr, uint32_t gate_want = 0);
hipError_t launch_shard_decide(const ShardDecideArgs& a, hipStream_t s);
hipError_t launch_flag_peers(unsigned long long* const* peer_flags, int me, int world, unsigned long long ticket,
                             hipStream_t s);

hipError_t launch_axpy_rows(int64_t row0, int64_t row1, const int32_t* extra, int n_extra,
                            const uint8_t* vflags, double* x, const double* y, double coef,
                            hipStream_t s, const uint32_t* gate = nullptr, uint32_t gate_want = 0);
hipError_t launch_row_dot(int tile0, int tile1, int nv, int T, const double* g, const double* gC, double* partials,
                          int n_tiles, hipStream_t s);
hipError_t launch_direction(int tile0, int tile1, int nv, int T, const uint8_t* vflags, double* g,
                            const double* gC, double* d, const double* pg, const double* pd,
                            const double* scal, int use_constraint, int cg_history,
                            double* partials, int n_tiles, int write_g, hipStream_t s,
                            const uint32_t* gate = nullptr, uint32_t gate_want = 0, int pd_neg_pg = 0,
                            int precond = 0);
hipError_t launch_axpy_masked(int64_t n_rows, const uint8_t* vflags, double* x, const double* y,
                              double coef, hipStream_t s);
hipError_t launch_permute_in(int nv, const int32_t* perm, const double* src_ext, double* dst_int,
                             int ncomp, hipStream_t s);
hipError_t launch_permute_out(int nv, const int32_t* perm, const double* src_int, double* dst_ext,
                              int ncomp, hipStream_t s);
hipError_t launch_grad_cotan(int n, const double* u, const double* v, double* gu, double* gv,
                             hipStream_t s);
hipError_t launch_p1_divergence(int nv, int nf, const double* pos, const double* tilts,
                                const int32_t* tri, double* div, double* area, double* g0,
                                double* g1, double* g2, hipStream_t s);
hipError_t launch_laplacian_scatter(int dim, int nv, int nf, const double* weights,
                                    const int32_t* tri, const double* field, double* out,
                                    hipStream_t s);
hipError_t launch_curvature_raw(int nv, int nf, const double* pos, const int32_t* tri,
                                double* k_vecs, double* areas, double* weights, double* va0,
                                double* va1, double* va2, hipStream_t s);

}  // namespace ms
