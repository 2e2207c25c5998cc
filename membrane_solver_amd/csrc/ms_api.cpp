// Host side of libmembrane_hip.so: context management, HBM residency, the
// device-resident minimizer step, and the extern "C" ABI of
// include/membrane_hip.h.  Arithmetic lives in ms_kernels.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <new>

#include <dlfcn.h>

#include "ms_internal.h"

static void shard_comm_destroy(void* comm);  // (RCCL binding further down)

using namespace ms;

namespace {
std::string g_last_error;  // for failures that have no context yet
}

struct ms_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Tiling til;
  int shard_rank = 0, shard_count = 1;
  int tile0 = 0, tile1 = 0;
  int cap = 0;
  // connectivity in HBM
  int32_t* d_perm = nullptr;
  int32_t* d_tile_facet_off = nullptr;
  TileFacet* d_tile_facets = nullptr;
  uint32_t* d_tile_facets32 = nullptr;  // packed copy (DeviceMesh::tile_facets32), or nullptr
  double* d_tf_gamma = nullptr;
  // Body's cached volume gradient (geometry/body.py:386-407): what the last fresh evaluation inside
  // ms_project_volume computed, and its squared norm
  double* d_volgrad_cache = nullptr;
  double volgrad_cache_norm2 = 0.0;
  bool volgrad_cache_valid = false;
  bool gamma_uniform = true, kc_uniform = true;  // ms_set_surface_tension / ms_set_bending_params decide
  double gamma_const = 1.0, kappa_const = 0.0, c0_const = 0.0;
  int32_t* d_tile_halo_off = nullptr;
  int32_t* d_halo_ids = nullptr;
  int32_t* d_tile_ent_off = nullptr;
  uint16_t* d_tile_voff = nullptr;
  uint16_t* d_vent = nullptr;
  uint8_t* d_vflags = nullptr;
  double* d_kappa = nullptr;
  double* d_c0 = nullptr;
  // vertex tilt fields: [0] the single field (ms_set_tilts), [1]/[2] inner / outer leaflet
  // (ms_set_leaflet_tilts).  All arrays (nvp,3) in patch order.
  struct TiltField {
    double* tilts = nullptr;   // current tangent tilts
    double* trial = nullptr;   // tilts projected onto a trial surface / relaxation trial
    double* grad = nullptr;    // dE/dt of the last tilt-gradient evaluation
    double* dir = nullptr;     // relaxation: CG direction
    double* minv = nullptr;    // relaxation: Jacobi M^-1 (nvp)
    double k_tilt = 0.0;       // tilt modulus (0 = magnitude module contributes nothing)
    int consistent = 0;        // tilt_mass_mode: consistent P1 mass instead of lumped
    double k_smooth = 0.0;     // smoothness rigidity of the energy
    double k_smooth_precond = 0.0;  // rigidity entering the Jacobi diagonal
    uint8_t fixed_bit = 0;     // vertex flag bit that clamps a row of this field
    uint32_t mod_tilt = 0, mod_smooth = 0, mod_bt = 0;  // module bits that read this field
    int s_etilt = 0, s_ets = 0, s_gn2 = 0, s_rz = 0, s_ebt = 0;  // reduction slots
    // leaflet bending_tilt: per-vertex (kappa, c0), the per-vertex record of the last energy pass
    // {base, A_eff, kappa*ratio*H, 0} and the sign of the divergence term
    double* kappa = nullptr;
    double* c0 = nullptr;
    double* bt_vert = nullptr;
    double div_sign = 1.0;
    // disk tilt target (tilt_disk_target_in/out): tagged rows, parameters, the difference field
    uint8_t* disk = nullptr;
    double* diff = nullptr;
    double* dt_target = nullptr;   // theta(r) r_hat of the tagged rows on the frozen surface of a running relaxation
    bool dt_target_valid = false;
    ms_disk_target_params dt = {};
    uint32_t mod_dt = 0;
    int s_edt = 0, s_dtr = 0;
    bool any_free = true;      // some row of this field is not clamped (kept current by the flag setters)
    double* va = nullptr;      // relaxation: barycentric vertex areas of the frozen positions
  } tf[3];
  // steepest-descent restart on an unchanged gradient (ms_step): the direction is -G and is not written out;
  // trial passes then read G with -alpha (bitwise the same x + alpha d).  After such a step is accepted the
  // CG history's previous direction is -PG, which the next fused direction pass derives instead of loading.
  bool dir_implicit = false;
  bool pd_neg_pg = false;
  bool precond = false;      // ms_stepper_params.precondition of the step in progress (CG only)
  // speculative line-search ladder (ms_step): when the last accepted step needed n > 1 Armijo trials, the next
  // n trials are queued at once; stage k > 0 runs only if the device-side Armijo test of stage k-1 failed
  // (k_armijo_gate).  Each stage posts its scalars to its own mailbox; the host takes the same decisions from the
  // same doubles, so the trajectory does not change -- only the host round trips between trials disappear.
  static constexpr int SPEC_STAGES = 3;  // extra mailboxes (stage 0 uses the main one)
  struct Mailbox {
    double* h_scal = nullptr;              // host copy of the values (filled by fetch)
    unsigned long long* h_seq = nullptr;   // pinned, mapped: 2*MS_MB_WORDS words, {value bits, sequence word} per entry
    unsigned long long* d_h_seq = nullptr;
    unsigned long long expected[MS_MB_WORDS] = {0};
  } spec[2][SPEC_STAGES];        // (everything a round posts to exists twice: a round can be queued while the one
                                 // before it has not been read yet -- see Ahead)
  Mailbox first_mb[2];           // mailbox of a round's first launch
  Mailbox grad_mb[2];            // mailbox of the gradient pass queued behind a round
  bool kc_pending = false;       // that pass ran for the accepted x: the next ms_step takes its result
  int kc_parity = 0;
  int next_parity = 0, cur_parity = 0;
  // a round queued for a step that has not started yet (queue_ahead)
  struct RoundPlanT {
    int n0 = 1, n_st = 0;
    double alphas[MS_MAX_TRIALS + SPEC_STAGES] = {0};
  };
  struct Ahead {
    bool valid = false;
    bool go_known = false, go = false;  // the device's answer (FoldArgs::go_out) has been read / was DEC_GO
    int kind = 0, go_kind = 0, parity = 0, src = 0, stepper = 0, max_iter = 0;
    bool implicit = false;
    double alpha0 = 0, energy0 = 0, beta = 0, c1 = 0, tol2p = 0, lim = 0;
    RoundPlanT plan;
  } ahead;
  bool ahead_enable = true;      // MS_AHEAD=0 switches the rounds queued ahead off
  int steps_left = 0;            // ... steps that follow the current one in this ms_minimize call
  bool ahead_allowed = false;    // set by ms_minimize: the caller is the library's own loop (nothing else touches the
                                 // context between two steps), and another step follows
  // The direction fold of a round's gradient pass can be left out when the round is queued (defer_dir) and merged
  // into the first fold of the round after it: that round's energy launch then starts right behind the gradient pass
  bool defer_dir = false;                 // reduce_slots: do not launch a direction fold, remember its mask
  uint32_t dir_deferred_mask = 0;
  bool dir_pending[2] = {false, false};   // the round's gradient pass has run (if its gate was open) without its fold
  uint32_t dir_mask[2] = {0, 0};
  uint32_t* kc_gate[2] = {nullptr, nullptr};  // decision word that pass was gated on
  uint32_t cur_extra_mask = 0;            // reduce_slots: fold these slots of the ordinary partials as well
  int cur_go_kind = 0;                    // ... and take the GO decision first (FoldArgs::go_kind and its parameters)
  double cur_go_val[6] = {0};             // tol^2 (1+1e-9), guard bound, energy0, c, alpha_0, beta
  bool cur_gate_fold_only = false;        // the gate (and the ran check) belongs to the fold, not to the energy kernel
  bool last_hist_descent = false;         // the last direction with CG history was a descent direction
  uint32_t* cur_go = nullptr;          // the direction fold being queued may open the next round: its GO word
  const double* cur_rhs_dev = nullptr; // the fold being queued takes its right-hand sides from the device
  long q_ahead = 0, q_adopted = 0, q_dropped = 0;
  int kc_stepper = 0;
  bool kc_use_history = false;
  // Decision records (ms_internal.h DEC_*): one 128-byte line each, written by the head workgroup of the fold that
  // closes a line-search stage, read by every kernel queued behind that stage.  Record k belongs to stage k of the
  // round being queued; the stream orders a record's readers between its writers.
  uint32_t* d_dec = nullptr;
  static constexpr int N_DEC = 16;
  const uint32_t* cur_gate = nullptr;  // decision word the launches being queued test (nullptr: unconditional)
  uint32_t cur_gate_want = 0;
  uint32_t* cur_dec = nullptr;         // the fold being queued closes a stage: its decision goes here
  double cur_rhs[MS_MAX_TRIALS] = {0}; // ... taken against these Armijo right-hand sides (trial order)
  bool cur_check_ran = false;          // the tile kernel in front of the fold being queued is gated by cur_gate
  unsigned long long* h_err = nullptr; // pinned word a fold sets when a gated launch ran on part of its workgroups
  unsigned long long* d_h_err = nullptr;
  long queue_mismatches = 0;           // host and device decisions that differed (every one is also a hard error)
  long q_rounds = 0, q_multi = 0, q_wasted = 0, q_side_accepts = 0;  // queue statistics (ms_queue_stats)
  int escalate_after = 3;              // rejections after which a search goes to multi-trial launches (MS_ESCALATE=n)
  // multi-trial launch: trials 0 .. n-2 of a ladder are evaluated in the same energy launch as trial n-1
  // (k_energy<MULTI>); the last trial uses the ordinary outputs, early trial j the side set j
  bool pair_enable = true;       // MS_PAIR=0 switches it off
  bool escalate = true;          // MS_ESCALATE=0: a search that keeps rejecting stays with what the history suggests
  int pair_force = 0;            // MS_PAIR=2 / 3: pair (/ pair + a gated third trial) whenever possible, whatever
                                 // the history predicts (tests)
  int pair_on = 0;               // phase_energy / reduce_slots: the launch being queued evaluates this many trials
  // the early trials of a multi-trial launch (the ones expected to fail) are evaluated for their energies only:
  // no trial positions, no bending factors written for them (ms_step; the sharded driver needs the factor rows)
  bool no_fast = false;          // MS_NO_FAST=1 (variant builds only)
  bool pair_lean = false;
  bool pair_lean_enable = true;  // MS_PAIR_LEAN=0: write the first two early trials' outputs (copied back if one is accepted)
  double pair_alpha[MS_MAX_TRIALS] = {0};  // alphas of the early trials
  static constexpr int N_SIDE = MS_MAX_TRIALS - 1;
  struct SideSet {
    double* partials = nullptr;
    double* scal = nullptr;
    Mailbox mb[2];
  } side[N_SIDE];
  double* xt3 = nullptr;         // full outputs of early trial 1 (MS_PAIR_LEAN=0)
  double* fK3 = nullptr;
  double* fA3 = nullptr;
  double* xt2 = nullptr;         // ... of early trial 0 (MS_PAIR_LEAN=0, the sharded pair)
  double* fK2 = nullptr;
  double* fA2 = nullptr;
  // line-search history (prediction only -- never changes a result), kept per kind of direction: [0] searches along
  // d = -g (gradient descent, CG restarts), [1] along a direction with CG history.  In the steady state of the
  // headline workload the two alternate and behave nothing alike (the second kind is no descent direction or runs out
  // of trials); one shared record would have every search predict the other kind.
  //   pred_trials: trials the last search of the kind spent (an exhausted one: all of them)
  //   acc / rej: of the last LS_HIST accepted steps the accepted alpha and the smallest alpha rejected on the way to
  //   it (INFINITY: accepted at once)
  static constexpr int LS_HIST = 8;
  struct LsHist {
    int pred_trials = 1;
    double acc[LS_HIST] = {0};
    double rej[LS_HIST] = {0};
    int n = 0;
  } ls[2];
  bool ls_reset = true;          // MS_LS_RESET=0: never forget the history on a regime change
  bool speculate = true;         // MS_SPECULATE=0 switches the ladder off
  bool relax_va_valid = false;  // a leaflet relaxation is running: tf[l].va describes the current x
  bool tilt_module_form = false;  // tilt_eval: the magnitude modules in their own mass mode (the plugin API's form,
                                  // tilt_leaflet.py:101-150) instead of the relaxation's vertex-area form
  int factors_leaflet = 0;  // which leaflet's back-prop factors fK/fA hold (1 in, 2 out; 0: not a leaflet's)
  double* d_bt_vert = nullptr;    // (nvp,4) bending_tilt per-vertex record of the last energy pass
  bool bt_valid = false;          // d_bt_vert describes the current x
  // tilt relaxation work space (positions frozen): unit vertex normals, CG direction, Jacobi M^-1
  double* d_tn = nullptr;
  std::vector<uint8_t> h_vflags;  // host copy of the vertex flag bytes (patch order)
  // per-vertex state (one allocation), patch order, nvp rows
  double* state = nullptr;
  bool own_state = true;
  double* buf[MS_BUF_COUNT] = {nullptr};
  double* d_partials = nullptr;
  double* d_scal = nullptr;
  double* h_scal = nullptr;    // host copy of the mailbox values (fetch() fills it once every slot has arrived)
  // pinned, device-mapped mailbox: one 16-byte {value bits, sequence word} entry per slot, written by k_reduce
  // in a single store
  unsigned long long* h_seq = nullptr;
  unsigned long long* d_h_seq = nullptr;
  unsigned long long ticket = 0;          // ticket of the latest reduce launch
  unsigned long long expected[MS_MB_WORDS] = {0};  // latest ticket that folds each slot (+ the decision entry)
  bool has_boundary = false;
  double* d_stage = nullptr;  // nv*3 staging in external row order
  double* last_g = nullptr;   // buffer holding the most recent finalized gradient
  ms_params params{};
  bool cg_have_history = false;
  int cg_iter_count = 0;
  bool factors_valid = false;
  // true while the mailbox energies / min edge / volume AND the factor buffers describe the
  // current x: set when ms_step accepts a trial that also wrote the factors, cleared by
  // every other energy pass and by every mutator
  // shard boundary exchange: rows each rank owns that other ranks' tiles read as halo
  std::vector<int32_t> bnd_off;  // shard_count + 1
  int32_t* d_bnd_rows = nullptr;
  int32_t* d_bnd_off = nullptr;
  int bnd_max = 0;               // longest per-rank list (message stride)
  int32_t* d_halo_rows = nullptr;  // rows of other ranks that THIS rank's tiles read
  int n_halo_rows = 0;
  double* d_scal_all = nullptr;    // shard_count x MS_NSCAL, filled by unpack
  // library-side sharded driver (ms_shard_*): RCCL communicator or a caller-supplied all-gather,
  // message buffers, pinned mailbox for the gathered scalar headers
  void* comm = nullptr;
  ms_allgather_fn allgather_cb = nullptr;
  void* allgather_user = nullptr;
  double* d_xsend = nullptr;
  double* d_xrecv = nullptr;
  double* h_scal_all = nullptr;    // pinned + mapped, shard_count x MS_NSCAL
  double* d_h_scal_all = nullptr;
  unsigned long long* h_xseq = nullptr;
  unsigned long long* d_h_xseq = nullptr;
  unsigned long long xticket = 0;
  // peer-to-peer exchange (ms_shard_peer_*): two receive slabs (the exchanges alternate between them) and one flag
  // word per (slab, peer) on this rank; the peers' slabs / flag words as this process sees them
  double* d_peer_slab = nullptr;            // 2 x shard_count x stride doubles
  unsigned long long* d_peer_flag = nullptr;  // 2 x 16 words
  size_t peer_stride = 0;                   // doubles per (slab, rank) slot
  std::vector<double*> peer_slabs;          // [rank]: base of that rank's slabs (own: d_peer_slab)
  std::vector<unsigned long long*> peer_flags;
  std::vector<void*> peer_opened;           // hipIpcOpenMemHandle results to close
  bool peer_on = false;
  int peer_mem_kind = -1;                   // 0 uncached, 1 fine-grained, 2 plain hipMalloc (peer_alloc)
  unsigned long long peer_ticket = 0;
  // MS_PEER_WAIT=stream: the flag words are raised and awaited by stream memory operations (hipStreamWriteValue64 behind
  // the pack kernel, hipStreamWaitValue64 in front of the unpack kernel) instead of a flag kernel and a waiting wave:
  // nothing of this rank occupies the GPU while it waits for a peer.  The wait itself has no bound -- the host's poll
  // has (2 s), and releases the words itself before it reports the error.
  PeerFlags* d_peer_flagtab = nullptr;      // [2]: per exchange parity, the peers' flag rows (read by the pack kernel)
  unsigned int* d_peer_arrived = nullptr;   // 16 block counters of the pack kernel
  bool peer_stream_ops = false;
  hipStream_t peer_aux = nullptr;           // (the release after a timeout goes through a stream of its own)
  ms_barrier_fn peer_barrier = nullptr;     // contexts of one process: host-side wait instead of the waiting wave
  void* peer_barrier_user = nullptr;
  double sh_scal[MS_NSCAL] = {0};  // rank-ordered fold of the last exchanges
  double sh_scal2[MS_NSCAL] = {0}; // ... of a pair launch's other trial (header slots SH_ALT + slot)
  double* pair_scal2 = nullptr;    // where a pair launch's second fold goes (nullptr: d_scal2)
  bool sh_carry_valid = false, sh_grad_valid = false;
  bool sh_maxg2_valid = false;  // the last direction exchange also carried the gradient rows and max|g_i|^2
  long sh_exchanges = 0;
  bool carry_valid = false;
  // true while buffer G holds the finalized gradient of the current x (set by ms_step's fused
  // gradient pass, survives a failed line search, cleared together with carry_valid)
  bool deterministic = false;  // ms_set_deterministic: staged CSR gather instead of LDS atomics
  bool grad_valid = false;
  bool maxg2_valid = false;  // the mailbox holds |g|^2 and max|g_i|^2 of the gradient in buffer G
  // optional per-kernel timing (ms_profile_*)
  bool profiling = false;
  struct ProfRec {
    hipEvent_t a, b;
    int kind;
    int ran_idx;  // gated launch: entry of d_prof_ran that says whether its gate was open (-1: not gated)
  };
  static constexpr int PROF_RAN_CAP = 1 << 16;
  uint32_t* d_prof_ran = nullptr;
  int prof_ran_next = 0;
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[MS_PROF_KINDS] = {0};
  int64_t prof_n[MS_PROF_KINDS] = {0};
  std::string err;
  // one-tile meshes: launches are recorded and run by ONE workgroup, pack by pack (ms_internal.h: ExecRecorder)
  // the resident step kernel (ms_resident.inc): meshes whose tiles all fit on the chip at once
  bool resident_enable = true;   // MS_RESIDENT=0 switches it off
  int resident_ok = -1;          // -1 not asked yet; 0 / 1: every tile's workgroup can be co-resident
  size_t resident_lds = 0;
  double* d_res_partials = nullptr;
  unsigned int* d_res_bar = nullptr;
  double* d_res_log = nullptr;
  double* d_res_result = nullptr;
  std::vector<double> h_res_log;
  long resident_launches = 0, resident_steps = 0, resident_bails = 0;
  ExecRecorder exec;
  bool exec_relax = true;    // tilt relaxations run as a device program (CK_RELAX); MS_EXEC_RELAX=0: host-driven
  double* d_relax_cells = nullptr;           // [0] trial coefficient, [1] Fletcher-Reeves beta (written by the program)
  unsigned long long* h_relax_box = nullptr; // pinned result mailbox of the program: {iterations, evaluations, parity, done}
  unsigned long long* d_h_relax_box = nullptr;
  unsigned long long relax_ticket = 0;
  long relax_programs = 0;
  bool exec_on = false;      // the recorder is attached to `stream`
  bool exec_wanted = false;  // ... and is to be re-attached when profiling (which needs one launch per kernel) ends
};

namespace {

// The stream for an operation that is NOT one of the library's recorded kernel launches (copies, memsets, stream
// synchronisation, collectives): whatever the one-workgroup interpreter has recorded so far is launched first, so the
// operation finds the stream in the state the launch-per-kernel path would have left it in.
inline hipStream_t S(ms_ctx* c) {
  if (c->exec_on) (void)c->exec.flush();
  return c->stream;
}
inline int exec_flush(ms_ctx* c) {
  if (!c->exec_on) return MS_OK;
  const hipError_t e = c->exec.flush();
  if (e == hipSuccess) return MS_OK;
  c->err = std::string("k_exec launch: ") + hipGetErrorString(e);
  return MS_ERR_HIP;
}

// zero-fill in stream order (a record of its own in a one-tile context: no flush)
inline int zero_doubles(ms_ctx* c, double* p, size_t bytes) {
  if (c->exec_on) {
    ExecMemsetArgs a;
    a.p = p;
    a.n = (int64_t)(bytes / sizeof(double));
    const hipError_t e = c->exec.push(CK_MEMSET, 0, 0, 0, 1, 0, 0, &a, sizeof(a));
    if (e == hipSuccess) return MS_OK;
    c->err = std::string("k_exec record: ") + hipGetErrorString(e);
    return MS_ERR_HIP;
  }
  const hipError_t e = hipMemsetAsync(p, 0, bytes, c->stream);
  if (e == hipSuccess) return MS_OK;
  c->err = std::string("hipMemsetAsync: ") + hipGetErrorString(e);
  return MS_ERR_HIP;
}

inline const double* trial_dir(const ms_ctx* c) { return c->dir_implicit ? c->buf[MS_BUF_G] : c->buf[MS_BUF_D]; }
inline double trial_alpha(const ms_ctx* c, double alpha) { return c->dir_implicit ? -alpha : alpha; }

int fail(ms_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_last_error = msg;
  return code;
}

int fail_hip(ms_ctx* c, hipError_t e, const char* what) {
  return fail(c, MS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIPCHK(c, call)                                   \
  do {                                                    \
    hipError_t e_ = (call);                               \
    if (e_ != hipSuccess) return fail_hip((c), e_, #call); \
  } while (0)

DeviceMesh device_mesh(const ms_ctx* c) {
  DeviceMesh m;
  m.nv = c->til.nv;
  m.T = c->til.T;
  m.own = c->til.own;
  m.n_tiles = c->til.n_tiles;
  m.has_boundary = c->has_boundary ? 1 : 0;
  m.tile_facet_off = c->d_tile_facet_off;
  m.tile_facets = c->d_tile_facets;
  m.tile_facets32 = c->d_tile_facets32;
  m.no_fast = c->no_fast ? 1 : 0;
  m.tf_gamma = c->d_tf_gamma;
  m.gamma_uniform = c->gamma_uniform ? 1 : 0;
  m.gamma_const = c->gamma_const;
  m.kc_uniform = c->kc_uniform ? 1 : 0;
  m.kappa_const = c->kappa_const;
  m.c0_const = c->c0_const;
  m.tile_halo_off = c->d_tile_halo_off;
  m.halo_ids = c->d_halo_ids;
  m.tile_ent_off = c->d_tile_ent_off;
  m.tile_voff = c->d_tile_voff;
  m.vent = c->d_vent;
  m.vflags = c->d_vflags;
  m.kappa = c->d_kappa;
  m.c0 = c->d_c0;
  return m;
}

// RAII-free event bracket: begin() before a launch, end() after it.
// MS_HOST_TIMING=1 (diagnostic): host time from "a fetch found its mailbox complete" to "the next k_energy launch call
// returned" (what the GPU idles through, minus its own dispatch latency), and the launch call alone; printed by
// ms_destroy
struct HostTiming {
  bool on = getenv("MS_HOST_TIMING") != nullptr && atoi(getenv("MS_HOST_TIMING")) != 0;
  double t_fetch = 0.0;
  bool armed = false;
  double sum_gap = 0.0, sum_launch = 0.0, max_gap = 0.0;
  long n = 0;
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  }
};
static HostTiming g_host_timing;

// A gated launch that found its gate closed returns at once: not a sample of the kernel.  Whether it ran is what the
// decision word said when the launch read it; with profiling on, a one-lane kernel behind the launch (outside its event
// bracket) records exactly that comparison -- the word cannot change in between, only a later fold rewrites it.  (Rounds
// 1-3 guessed from the duration, "< 12 us = empty", which discarded every real launch of a 131 k-facet mesh.)
struct ProfScope {
  ms_ctx* c;
  hipEvent_t a = nullptr, b = nullptr;
  int kind;
  const uint32_t* gate = nullptr;
  uint32_t want = 0;
  ProfScope(ms_ctx* ctx, int k, const uint32_t* gate_word = nullptr, uint32_t gate_want = 0)
      : c(ctx), kind(k), gate(gate_word), want(gate_want) {
    if (!c->profiling || k < 0) return;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!c->prof_pool.empty()) {
        e = c->prof_pool.back();
        c->prof_pool.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
      }
      return e;
    };
    a = get();
    b = get();
    if (a && b) (void)hipEventRecord(a, c->stream);
  }
  ~ProfScope() {
    if (!c->profiling || kind < 0 || !a || !b) return;
    (void)hipEventRecord(b, c->stream);
    int ran_idx = -1;
    if (gate != nullptr && c->d_prof_ran != nullptr && c->prof_ran_next < ms_ctx::PROF_RAN_CAP) {
      ran_idx = c->prof_ran_next++;
      (void)launch_gate_probe(gate, want, c->d_prof_ran + ran_idx, c->stream);
    }
    c->prof_pending.push_back({a, b, kind, ran_idx});
  }
};

constexpr uint32_t MASK_ENERGY = (1u << MS_S_ESURF) | (1u << MS_S_VOL) | (1u << MS_S_EBEND) |
                                 (1u << MS_S_MINEDGE2) | (1u << MS_S_GUARD) | (1u << MS_S_ETILT) |
                                 (1u << MS_S_EBT) | (1u << MS_S_ETS) | (1u << MS_S_ETILT_IN) |
                                 (1u << MS_S_ETILT_OUT) | (1u << MS_S_ETS_IN) | (1u << MS_S_ETS_OUT) |
                                 (1u << MS_S_EBT_IN) | (1u << MS_S_EBT_OUT) | (1u << MS_S_EDT_IN) | (1u << MS_S_EDT_OUT);
constexpr uint32_t MS_TILT_MODS = MS_MOD_TILT | MS_MOD_BENDING_TILT | MS_MOD_TILT_SMOOTH;  // modules reading the single tilt field
constexpr uint32_t MS_LEAFLET_BT = MS_MOD_BENDING_TILT_IN | MS_MOD_BENDING_TILT_OUT;
constexpr uint32_t MS_LEAFLET_DT = MS_MOD_TILT_DISK_TARGET_IN | MS_MOD_TILT_DISK_TARGET_OUT;
constexpr uint32_t MS_LEAFLET_MODS = MS_MOD_TILT_IN | MS_MOD_TILT_OUT | MS_MOD_TILT_SMOOTH_IN | MS_MOD_TILT_SMOOTH_OUT | MS_LEAFLET_BT | MS_LEAFLET_DT;
constexpr uint32_t MS_ANY_TILT_MODS = MS_TILT_MODS | MS_LEAFLET_MODS;
// modules whose shape gradient is added into g by a pass after K_C (so the direction cannot be fused)
constexpr uint32_t MS_TILT_SHAPE_MODS = MS_MOD_TILT | MS_MOD_TILT_IN | MS_MOD_TILT_OUT | MS_LEAFLET_BT | MS_LEAFLET_DT;
using TiltField = ms_ctx::TiltField;
using RoundPlan = ms_ctx::RoundPlanT;

// the tilt fields the module set reads: [0] single field, [1] inner, [2] outer leaflet
int active_fields(ms_ctx* c, uint32_t mods, TiltField* out[3]) {
  int n = 0;
  for (int k = 0; k < 3; ++k) {
    TiltField& f = c->tf[k];
    const uint32_t reads = k == 0 ? MS_TILT_MODS : (f.mod_tilt | f.mod_smooth | f.mod_bt | f.mod_dt);
    if (mods & reads) out[n++] = &f;
  }
  return n;
}

// mode 0 energy / 1 energy+gradients read `src`; mode 2 projects `src` onto the tangent
// planes of x (+ alpha d) and writes `dst`.
int tilt_pass_f(ms_ctx* c, TiltField& f, int mode, bool use_dir, double alpha, const double* src = nullptr,
                double* dst = nullptr, bool shape_gradient = true, bool lumped = false, double k_override = -1.0,
                int slot_override = -1, bool tg_accumulate = false) {
  if (!f.tilts) return fail(c, MS_ERR_STATE, "tilt module active but its tilt field was never set (ms_set_tilts / ms_set_leaflet_tilts)");
  TiltArgs a;
  a.fields = nullptr;
  a.fields_rows = 0;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = use_dir ? trial_dir(c) : nullptr;
  a.alpha = trial_alpha(c, alpha);
  a.tilts = src ? src : f.tilts;
  a.tilts_out = dst ? dst : f.tilts;
  a.k_tilt = k_override >= 0.0 ? k_override : f.k_tilt;
  a.g = shape_gradient ? c->buf[MS_BUF_G] : nullptr;
  a.tilt_grad = f.grad;
  a.minv = nullptr;
  a.partials = c->d_partials;
  a.e_slot = slot_override >= 0 ? slot_override : f.s_etilt;
  a.consistent = (f.consistent && !lumped) ? 1 : 0;
  a.tg_accumulate = tg_accumulate ? 1 : 0;
  a.cons_tilt_grad = (a.consistent && c->tilt_module_form) ? 1 : 0;
  a.va_out = nullptr;
  {
    ProfScope ps(c, 4);
    HIPCHK(c, launch_tilt(a, mode, c->cap, c->til.max_ent, c->stream));
  }
  return MS_OK;
}
int tilt_pass(ms_ctx* c, int mode, bool use_dir, double alpha, const double* src = nullptr,
              double* dst = nullptr, bool shape_gradient = true) {
  return tilt_pass_f(c, c->tf[0], mode, use_dir, alpha, src, dst, shape_gradient);
}
// bending_tilt facet pass (mode 0 energy / 1 + factors / 2 + tilt gradient) on the positions of
// the preceding energy pass; `tilts` = the tangent tilts belonging to those positions
int bt_pass_f(ms_ctx* c, TiltField& f, int mode, bool use_dir, double alpha, const double* tilts,
              bool with_tilt_energy = false);
int bt_pass(ms_ctx* c, int mode, bool use_dir, double alpha, const double* tilts, bool with_tilt_energy = false) {
  return bt_pass_f(c, c->tf[0], mode, use_dir, alpha, tilts, with_tilt_energy);
}
// with_tilt_energy (mode 0): the kernel also sums the field's tilt-magnitude energy (per-facet form) from the rows it
// stages anyway -- same operations as k_tilt's energy-only launch, which the caller then leaves out
int bt_pass_f(ms_ctx* c, TiltField& f, int mode, bool use_dir, double alpha, const double* tilts, bool with_tilt_energy) {
  if (!f.tilts) return fail(c, MS_ERR_STATE, "bending_tilt module active but its tilt field was never set");
  const bool leaflet = &f != &c->tf[0];
  if (leaflet && (!f.kappa || !f.bt_vert))
    return fail(c, MS_ERR_STATE, "bending_tilt_in/out active but ms_set_leaflet_bending was never called");
  BtArgs a;
  a.m = device_mesh(c);
  if (leaflet) {
    a.m.kappa = f.kappa;
    a.m.kc_uniform = 0;  // per-leaflet arrays
    a.m.c0 = f.c0;
  }
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = use_dir ? trial_dir(c) : nullptr;
  a.alpha = trial_alpha(c, alpha);
  a.tilts = tilts;
  a.bt_vert = leaflet ? f.bt_vert : c->d_bt_vert;
  a.fK = c->buf[MS_BUF_FK];
  a.fA = c->buf[MS_BUF_FA];
  a.tilt_grad = f.grad;
  a.partials = c->d_partials;
  a.g = c->buf[MS_BUF_G];
  a.div_sign = f.div_sign;
  a.e_slot = f.s_ebt;
  a.k_tilt_fused = (with_tilt_energy && mode == 0) ? f.k_tilt : 0.0;
  a.e_tilt_slot = f.s_etilt;
  {
    ProfScope ps(c, 5);
    HIPCHK(c, launch_bt(a, mode, c->cap, c->til.max_ent, c->stream));
  }
  return MS_OK;
}
// tilt smoothness pass (mode 0 energy / 1 + tilt gradient / 2 Jacobi diagonal into `diag`)
int ts_pass_f(ms_ctx* c, TiltField& f, int mode, bool use_dir, double alpha, const double* tilts,
              double* diag = nullptr, double k_override = -1.0) {
  if (!f.tilts) return fail(c, MS_ERR_STATE, "tilt_smoothness module active but its tilt field was never set");
  TsArgs a;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = use_dir ? trial_dir(c) : nullptr;
  a.alpha = trial_alpha(c, alpha);
  a.tilts = tilts;
  a.k_smooth = k_override >= 0.0 ? k_override : f.k_smooth;
  a.tilt_grad = f.grad;
  a.diag = diag;
  a.partials = c->d_partials;
  a.e_slot = f.s_ets;
  {
    ProfScope ps(c, 7);
    HIPCHK(c, launch_ts(a, mode, c->cap, c->til.max_ent, c->stream));
  }
  return MS_OK;
}
int ts_pass(ms_ctx* c, int mode, bool use_dir, double alpha, const double* tilts, double* diag = nullptr) {
  return ts_pass_f(c, c->tf[0], mode, use_dir, alpha, tilts, diag);
}
// tilt_disk_target_in/out on the positions x (+ alpha d) and `tilts`: (radius reduction ->) difference field ->
// tilt magnitude kernel on it with k = strength.  mode 0 energy / 1 energy + shape gradient (+ tilt gradient
// ADDED to f.grad when tilt_gradient).
int reduce_slots(ms_ctx* c, uint32_t mask);
int disk_target_pass(ms_ctx* c, TiltField& f, int mode, bool use_dir, double alpha, const double* tilts,
                     bool shape_gradient, bool tilt_gradient) {
  if (!f.disk || !f.diff) return fail(c, MS_ERR_STATE, "tilt_disk_target active but ms_set_leaflet_disk_target was never called");
  DiskTargetArgs a;
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.nv = c->til.nv;
  a.T = c->til.own;  // (row stride of the streaming kernels)
  a.n_tiles = c->til.n_tiles;
  a.vflags = c->d_vflags;
  a.disk = f.disk;
  a.x = c->buf[MS_BUF_X];
  a.d = use_dir ? trial_dir(c) : nullptr;
  a.alpha = trial_alpha(c, alpha);
  a.tilts = tilts;
  a.diff = f.diff;
  a.theta_b = f.dt.theta_b;
  a.lambda = f.dt.lambda;
  a.radius = f.dt.radius;
  for (int k = 0; k < 3; ++k) {
    a.center[k] = f.dt.center[k];
    a.normal[k] = f.dt.normal[k];
  }
  a.scal = c->d_scal;
  a.partials = c->d_partials;
  a.r_slot = f.s_dtr;
  a.target = nullptr;
  // a relaxation in progress: x is frozen, so the disk radius and the target profile theta(r) r_hat are computed ONCE
  // (relax_fields primes them) and every evaluation of the relaxation only takes the difference
  const bool frozen = c->relax_va_valid && !use_dir;
  {
    ProfScope ps(c, 6);
    if (frozen && !f.dt_target) {
      const size_t b3 = sizeof(double) * 3 * (size_t)c->til.nvp;
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.dt_target), b3));
      HIPCHK(c, hipMemset(f.dt_target, 0, b3));
    }
    a.target = frozen ? f.dt_target : nullptr;
    if (!frozen || !f.dt_target_valid) {
      if (!(f.dt.radius > 0.0)) {
        HIPCHK(c, launch_disk_target(a, 0, c->stream));
        int rc = reduce_slots(c, 1u << f.s_dtr);
        if (rc) return rc;
      }
      if (frozen) {
        HIPCHK(c, launch_disk_target(a, 2, c->stream));
        f.dt_target_valid = true;
      }
    }
    HIPCHK(c, launch_disk_target(a, frozen ? 3 : 1, c->stream));
  }
  return tilt_pass_f(c, f, mode, use_dir, alpha, f.diff, nullptr, shape_gradient, /*lumped=*/true, f.dt.strength,
                     f.s_edt, tilt_gradient);
}
constexpr uint32_t MASK_GRAD = (1u << MS_S_GGC) | (1u << MS_S_GCGC);
constexpr uint32_t MASK_DIR = (1u << MS_S_GNORM2) | (1u << MS_S_GDOTD) | (1u << MS_S_MAXD2) | (1u << MS_S_MAXG2);

// MS_TRACE_STEPS=1: one stderr line per step of ms_minimize (what the search did, host time since the step before,
// running queue statistics)
bool trace_steps() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_TRACE_STEPS");
    v = (e && atoi(e) != 0) ? 1 : 0;
  }
  return v == 1;
}
// MS_TRACE_QUEUE=1: one stderr line per fold launch / mailbox swap / fetch (debugging the line-search queue)
bool trace_queue() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MS_TRACE_QUEUE");
    v = (e && atoi(e) != 0) ? 1 : 0;
  }
  return v != 0;
}
const char* box_name(ms_ctx* c, const void* h_seq) {
  // (the mailboxes are swapped in and out of the context: name them by allocation)
  static std::map<const void*, std::string> names;
  auto it = names.find(h_seq);
  if (it == names.end()) it = names.emplace(h_seq, "box" + std::to_string(names.size())).first;
  (void)c;
  return it->second.c_str();
}

inline uint32_t* dec_word(ms_ctx* c, int parity, int k) { return c->d_dec + (size_t)MS_DEC_STRIDE * (parity * 4 + k); }
inline uint32_t* go_word(ms_ctx* c, int parity) { return c->d_dec + (size_t)MS_DEC_STRIDE * (8 + parity); }
inline const double* go_rhs(ms_ctx* c, int parity) { return reinterpret_cast<const double*>(go_word(c, parity) + 2); }

// energy slots whose sum is the energy the Armijo test compares (the ladder only runs for module sets whose energy
// is surface + bending: ms_step's can_chain)
uint32_t armijo_slots(const ms_ctx* c) {
  return ((c->params.modules & MS_MOD_SURFACE) ? (1u << MS_S_ESURF) : 0u) |
         ((c->params.modules & MS_MOD_BENDING) ? (1u << MS_S_EBEND) : 0u);
}

// slots an energy pass of `modules` leaves partials in: the core five, and the tilt families' only when one of them is on
uint32_t energy_mask(uint32_t modules) {
  constexpr uint32_t core = (1u << MS_S_ESURF) | (1u << MS_S_VOL) | (1u << MS_S_EBEND) | (1u << MS_S_MINEDGE2) |
                            (1u << MS_S_GUARD);
  return (modules & MS_ANY_TILT_MODS) ? MASK_ENERGY : core;
}

int reduce_slots(ms_ctx* c, uint32_t mask) {
  if (c->defer_dir && (mask & (1u << MS_S_GDOTD)) && c->cur_dec == nullptr) {
    c->dir_deferred_mask = mask;  // (queue_round: the fold of the round after this one takes these slots along)
    return MS_OK;
  }
  mask |= c->cur_extra_mask;
  ProfScope ps(c, 3, c->cur_gate, c->cur_gate_want);
  ++c->ticket;
  if (trace_queue())
    fprintf(stderr, "[msq] fold ticket %llu mask %#x -> %s gate %p want %u dec %p trials %d\n",
            (unsigned long long)(c->ticket), mask, box_name(c, c->h_seq), (const void*)c->cur_gate, c->cur_gate_want,
            (const void*)c->cur_dec, (int)c->pair_on);
  // a multi-trial launch has no tilt module: only the core slots carry anything (and the sharded driver parks the other
  // trial's fold in the tilt slots of the device scalars, which the full mask would overwrite)
  const int n_multi = c->pair_on > 1 ? c->pair_on : 1;
  if (n_multi > 1) {
    const uint32_t core = (1u << MS_S_ESURF) | (1u << MS_S_VOL) | (1u << MS_S_EBEND) | (1u << MS_S_MINEDGE2) |
                          (1u << MS_S_GUARD) | c->cur_extra_mask;
    // (nobody is to wait for the dropped slots: an older, gated-out launch may have left a ticket there)
    for (int sl = 0; sl < MS_NSCAL; ++sl)
      if (mask & ~core & (1u << sl)) {
        c->expected[sl] = 0;
        for (int k = 0; k + 1 < n_multi; ++k) c->side[k].mb[c->cur_parity].expected[sl] = 0;
      }
    mask &= core;
  }
  FoldArgs f;
  memset(&f, 0, sizeof(f));
  f.n_tiles = c->til.n_tiles;
  f.tile0 = c->tile0;
  f.tile1 = c->tile1;
  f.slot_mask = mask;
  f.ticket = c->ticket;
  f.n_sets = n_multi;
  f.gate = c->cur_gate;
  f.gate_want = c->cur_gate_want;
  f.check_ran = (c->cur_gate != nullptr && c->cur_check_ran) ? 1 : 0;
  f.dec_out = c->cur_dec;
  f.counter = c->d_dec ? c->d_dec + (size_t)MS_DEC_STRIDE * (ms_ctx::N_DEC - 1) : nullptr;  // (the last record's line)
  f.e_mask = armijo_slots(c);
  f.host_err = c->d_h_err;
  for (int j = 0; j < MS_MAX_TRIALS; ++j) f.rhs[j] = c->cur_rhs[j];
  // an energy-only early trial needs nothing but its energy slots (the head workgroup folds those); an early trial
  // with outputs of its own (MS_PAIR_LEAN=0, the sharded pair) can be accepted as it is and needs every slot
  f.side_full = (n_multi > 1 && (!c->pair_lean || !f.dec_out)) ? 1 : 0;
  for (int j = 0; j + 1 < n_multi; ++j) {
    ms_ctx::SideSet& sd = c->side[j];
    f.set[j].partials = sd.partials;
    f.set[j].scal = (j == 0 && c->pair_scal2) ? c->pair_scal2 : sd.scal;
    f.set[j].host_box = sd.mb[c->cur_parity].d_h_seq;
    const uint32_t posted = (f.side_full || !f.dec_out) ? mask : (mask & f.e_mask);
    for (int sl = 0; sl < MS_NSCAL; ++sl)
      if (posted & (1u << sl)) sd.mb[c->cur_parity].expected[sl] = c->ticket;
  }
  f.set[n_multi - 1].partials = c->d_partials;
  f.set[n_multi - 1].scal = c->d_scal;
  f.set[n_multi - 1].host_box = c->d_h_seq;
  for (int sl = 0; sl < MS_NSCAL; ++sl)
    if (mask & (1u << sl)) c->expected[sl] = c->ticket;
  f.rhs_dev = c->cur_rhs_dev;
  if (c->cur_go_kind && f.dec_out && (mask & (1u << MS_S_GDOTD))) {  // merged: direction scalars + GO + Armijo decision
    f.go_out = c->cur_go;
    f.go_kind = c->cur_go_kind;
    f.go_tol2 = c->cur_go_val[0];
    f.go_lim = c->cur_go_val[1];
    f.go_e0 = c->cur_go_val[2];
    f.go_c = c->cur_go_val[3];
    f.go_alpha0 = c->cur_go_val[4];
    f.go_beta = c->cur_go_val[5];
  }
  if (f.dec_out) c->expected[MS_MB_DEC] = c->ticket;
  HIPCHK(c, launch_reduce(f, c->stream));
  return MS_OK;
}

int phase_energy(ms_ctx* c, uint32_t modules, bool use_dir, double alpha, bool write_trial,
                 bool guard, bool write_factors, bool reduce_now = true) {
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  EnergyArgs a;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = use_dir ? trial_dir(c) : nullptr;
  a.alpha = trial_alpha(c, alpha);
  a.xt = write_trial ? c->buf[MS_BUF_XT] : nullptr;
  const bool bt = (modules & MS_MOD_BENDING_TILT) != 0;
  const bool lbt = (modules & MS_LEAFLET_BT) != 0;
  const bool bend = (modules & MS_MOD_BENDING) != 0 || bt || lbt;
  a.fK = (bend && write_factors) ? c->buf[MS_BUF_FK] : nullptr;
  a.fA = (bend && write_factors) ? c->buf[MS_BUF_FA] : nullptr;
  a.bt_vert = bt ? c->d_bt_vert : nullptr;
  a.bt_normals = nullptr;
  a.gate = c->cur_gate_fold_only ? nullptr : c->cur_gate;
  a.gate_want = c->cur_gate_want;
  a.atomic = c->deterministic ? 0 : 1;
  a.pair = 0;
  for (int j = 0; j < MS_MAX_TRIALS - 1; ++j) {
    a.alpha_side[j] = 0.0;
    a.partials_side[j] = nullptr;
  }
  for (int j = 0; j < 2; ++j) a.xt_side[j] = a.fK_side[j] = a.fA_side[j] = nullptr;
  if (c->pair_on > 1) {
    if (!use_dir || guard || !write_factors || !(modules & MS_MOD_BENDING) || bt || lbt || !c->side[0].partials)
      return fail(c, MS_ERR_STATE, "multi-trial launch: not an ordinary bending trial");
    if (c->pair_on > MS_MAX_TRIALS || !c->side[c->pair_on - 2].partials)
      return fail(c, MS_ERR_STATE, "multi-trial launch: side sets not allocated");
    a.pair = c->pair_on;
    for (int j = 0; j + 1 < c->pair_on; ++j) {
      a.alpha_side[j] = trial_alpha(c, c->pair_alpha[j]);
      a.partials_side[j] = c->side[j].partials;
    }
    if (!c->pair_lean) {  // the first two early trials write their own positions / factors
      double* const sx[2] = {c->xt2, c->xt3};
      double* const sk[2] = {c->fK2, c->fK3};
      double* const sa[2] = {c->fA2, c->fA3};
      for (int j = 0; j < 2 && j + 1 < c->pair_on; ++j) {
        if (!sk[j]) return fail(c, MS_ERR_STATE, "multi-trial launch: output side set not allocated");
        a.xt_side[j] = write_trial ? sx[j] : nullptr;
        a.fK_side[j] = sk[j];
        a.fA_side[j] = sa[j];
      }
      if (c->pair_on > 3) return fail(c, MS_ERR_STATE, "multi-trial launch: only two early trials can write outputs");
    }
  }
  if (bt && !c->d_bt_vert) return fail(c, MS_ERR_STATE, "bending_tilt: ms_set_params did not allocate its buffers");
  a.partials = c->d_partials;
  a.bending_model = c->params.bending_model;
  a.modules = modules;
  if (!lbt) {
    ProfScope ps(c, a.pair > 3 ? 10 : (a.pair == 3 ? 8 : (a.pair ? 7 : 0)), a.gate, a.gate_want);
    const double t_l0 = g_host_timing.on ? HostTiming::now() : 0.0;
    HIPCHK(c, launch_energy(a, guard && use_dir, c->cap, c->til.max_ent, c->stream));
    if (g_host_timing.on && g_host_timing.armed) {
      const double t1 = HostTiming::now();
      g_host_timing.armed = false;
      g_host_timing.sum_gap += t1 - g_host_timing.t_fetch;
      g_host_timing.max_gap = std::max(g_host_timing.max_gap, t1 - g_host_timing.t_fetch);
      g_host_timing.sum_launch += t1 - t_l0;
      ++g_host_timing.n;
    }
  } else {
    // leaflet bending_tilt: the tilt projections and the unit vertex normals of the evaluated positions come
    // first, then per leaflet an energy pass with that leaflet's (kappa, c0) (signed curvature, K_dir = n)
    // directly followed by its facet pass, because the factor buffers serve one leaflet at a time
    for (int l = 1; l <= 2 && use_dir; ++l) {
      TiltField& f = c->tf[l];
      if (!(modules & (f.mod_tilt | f.mod_smooth | f.mod_bt | f.mod_dt))) continue;
      int rc = tilt_pass_f(c, f, 2, true, alpha, f.tilts, f.trial);
      if (rc) return rc;
    }
    if (!c->d_tn) {
      const size_t b3 = sizeof(double) * 3 * (size_t)c->til.nvp;
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_tn), b3));
      HIPCHK(c, hipMemset(c->d_tn, 0, b3));
    }
    {
      TiltArgs ta;
      ta.m = device_mesh(c);
      ta.tile0 = c->tile0;
      ta.tile1 = c->tile1;
      ta.x = c->buf[MS_BUF_X];
      ta.d = use_dir ? trial_dir(c) : nullptr;
      ta.alpha = trial_alpha(c, alpha);
      ta.tilts = c->tf[1].tilts ? c->tf[1].tilts : c->tf[2].tilts;
      ta.tilts_out = c->d_tn;
      ta.k_tilt = 0.0;
      ta.g = nullptr;
      ta.tilt_grad = nullptr;
      ta.minv = nullptr;
      ta.partials = c->d_partials;
      ta.e_slot = MS_S_ETILT;
      ta.consistent = 0;
      ta.tg_accumulate = 0;
      ta.cons_tilt_grad = 0;
      ta.va_out = nullptr;
      if (!ta.tilts) return fail(c, MS_ERR_STATE, "bending_tilt_in/out active but ms_set_leaflet_tilts was never called");
      ProfScope ps(c, 4);
      HIPCHK(c, launch_tilt(ta, 3, c->cap, c->til.max_ent, c->stream));
    }
    for (int l = 1; l <= 2; ++l) {
      TiltField& f = c->tf[l];
      if (!(modules & f.mod_bt)) continue;
      if (!f.kappa || !f.bt_vert || !f.tilts)
        return fail(c, MS_ERR_STATE, "bending_tilt_in/out active but ms_set_leaflet_tilts / ms_set_leaflet_bending were not called");
      EnergyArgs al = a;
      al.m.kappa = f.kappa;
      al.m.kc_uniform = 0;  // per-leaflet arrays
      al.m.c0 = f.c0;
      al.bt_vert = f.bt_vert;
      al.bt_normals = c->d_tn;
      al.bending_model = MS_BEND_HELFRICH;  // bending_tilt_leaflet.py:448-450
      al.modules = (modules & ~MS_LEAFLET_MODS & ~MS_TILT_MODS) | MS_MOD_BENDING_TILT;
      {
        ProfScope ps(c, 0);
        HIPCHK(c, launch_energy(al, guard && use_dir, c->cap, c->til.max_ent, c->stream));
      }
      int rc = bt_pass_f(c, f, write_factors ? 1 : 0, use_dir, alpha, use_dir ? f.trial : f.tilts);
      if (rc) return rc;
      if (write_factors) c->factors_leaflet = l;
    }
  }
  if (modules & MS_TILT_MODS) {
    int rc = MS_OK;
    const double* tilts = c->tf[0].tilts;
    if (use_dir) {
      // minimizer.py:723-733: the trial energy is taken with the tilts projected onto the
      // TRIAL surface's vertex tangent planes (kept aside; they become the stored tilts
      // only if this trial is accepted)
      rc = tilt_pass(c, 2, true, alpha, c->tf[0].tilts, c->tf[0].trial);
      if (rc) return rc;
      tilts = c->tf[0].trial;
    }
    // energy only, both modules: the bending_tilt kernel sums the tilt-magnitude energy as well (tilt_eval does the same)
    const bool fuse = (modules & MS_MOD_TILT) && bt && !write_factors && c->tf[0].k_tilt != 0.0 && !c->tf[0].consistent;
    if ((modules & MS_MOD_TILT) && !fuse) rc = tilt_pass(c, 0, use_dir, alpha, tilts);
    if (rc) return rc;
    if (bt) rc = bt_pass(c, write_factors ? 1 : 0, use_dir, alpha, tilts, fuse);
    if (rc) return rc;
    if (modules & MS_MOD_TILT_SMOOTH) rc = ts_pass(c, 0, use_dir, alpha, tilts);
    if (rc) return rc;
  }
  for (int l = 1; l <= 2 && (modules & MS_LEAFLET_MODS); ++l) {  // leaflet fields, same protocol
    TiltField& f = c->tf[l];
    if (!(modules & (f.mod_tilt | f.mod_smooth | f.mod_dt))) continue;
    int rc = MS_OK;
    const double* tilts = f.tilts;
    if (use_dir) {
      if (!lbt) rc = tilt_pass_f(c, f, 2, true, alpha, f.tilts, f.trial);  // (done above otherwise)
      if (rc) return rc;
      tilts = f.trial;
    }
    if (modules & f.mod_dt) rc = disk_target_pass(c, f, 0, use_dir, alpha, tilts, false, false);
    if (rc) return rc;
    if (modules & f.mod_tilt) rc = tilt_pass_f(c, f, 0, use_dir, alpha, tilts);
    if (rc) return rc;
    if (modules & f.mod_smooth) rc = ts_pass_f(c, f, 0, use_dir, alpha, tilts);
    if (rc) return rc;
  }
  if (reduce_now) {
    int rc = reduce_slots(c, energy_mask(modules));
    if (rc) return rc;
  }
  if (bend && write_factors) c->factors_valid = !use_dir;
  if (!lbt) c->factors_leaflet = 0;
  c->bt_valid = (bt || lbt) && !use_dir;
  return MS_OK;
}

// dir_mode: 0 = plain gradient pass (+ <g,gC> partials); 1/2 = fused direction (GD / CG history)
int phase_gradient(ms_ctx* c, uint32_t modules_in, double* g_out, bool accumulate, int dir_mode = 0,
                   bool reduce_now = true) {
  uint32_t modules = modules_in;
  if ((modules & (MS_MOD_BENDING | MS_MOD_BENDING_TILT | MS_LEAFLET_BT)) && !c->factors_valid)
    return fail(c, MS_ERR_STATE, "gradient pass needs the bending factors of an energy pass at x");
  if (modules & MS_MOD_BENDING_TILT)  // k_bt finished the factors: K_C is the plain bending back-prop
    modules = (modules & ~MS_MOD_BENDING_TILT) | MS_MOD_BENDING;
  // leaflet bending_tilt: one back-propagation per leaflet (per-corner area factors), starting with the
  // leaflet whose factors the energy pass left in fK/fA; the other leaflet's energy + facet pass is redone
  int lbt_order[2] = {0, 0}, n_lbt = 0;
  if (modules & MS_LEAFLET_BT) {
    if (dir_mode) return fail(c, MS_ERR_STATE, "bending_tilt_in/out cannot use the fused direction pass");
    if (modules & MS_MOD_BENDING) return fail(c, MS_ERR_STATE, "bending together with bending_tilt_in/out is outside the device path");
    if (c->params.bending_grad_mode != MS_GRAD_ANALYTIC)
      return fail(c, MS_ERR_STATE, "bending_tilt_in/out: only bending_gradient_mode=analytic is on the device path");
    for (int l = 2; l >= 1; --l)
      if (modules & c->tf[l].mod_bt) lbt_order[n_lbt++] = l;
    if (n_lbt == 2 && c->factors_leaflet == lbt_order[1]) std::swap(lbt_order[0], lbt_order[1]);
    modules |= MS_MOD_BENDING;
  }
  GradientArgs a;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.fK = c->buf[MS_BUF_FK];
  a.fA = c->buf[MS_BUF_FA];
  a.g = g_out;
  a.gC = (modules & MS_CON_VOLUME) ? c->buf[MS_BUF_GC] : nullptr;
  a.partials = c->d_partials;
  a.scal = c->d_scal;
  a.modules = modules;
  a.bending_grad_mode = c->params.bending_grad_mode;
  a.volume_stiffness = c->params.volume_stiffness;
  a.target_volume = c->params.target_volume;
  a.accumulate = accumulate ? 1 : 0;
  a.dir_mode = dir_mode;
  a.d = c->buf[MS_BUF_D];
  a.pg = c->buf[MS_BUF_PG];
  a.pd = c->buf[MS_BUF_PD];
  a.pd_neg_pg = (dir_mode == 2 && c->pd_neg_pg) ? 1 : 0;
  a.gate = c->cur_gate;
  a.gate_want = c->cur_gate_want;
  a.atomic = c->deterministic ? 0 : 1;
  a.bt_vert = nullptr;
  a.tilts = nullptr;
  a.div_sign = 1.0;
  if (n_lbt == 0) {
    ProfScope ps(c, gradient_lean_instance(a) ? 9 : 1, a.gate, a.gate_want);
    HIPCHK(c, launch_gradient(a, c->cap, c->til.max_ent, c->stream));
  }
  for (int k = 0; k < n_lbt; ++k) {
    TiltField& f = c->tf[lbt_order[k]];
    if (c->factors_leaflet != lbt_order[k]) {  // redo this leaflet's energy + facet pass at x (factors on)
      int rc = phase_energy(c, (modules_in & ~MS_LEAFLET_BT) | f.mod_bt, false, 0.0, false, false, true, false);
      if (rc) return rc;
    }
    GradientArgs al = a;
    al.m.kappa = f.kappa;
    al.m.kc_uniform = 0;  // per-leaflet arrays
    al.m.c0 = f.c0;
    al.bt_vert = f.bt_vert;
    al.tilts = f.tilts;
    al.div_sign = f.div_sign;
    if (k > 0) {  // the other modules' gradient went in with the first leaflet
      al.modules = MS_MOD_BENDING;
      al.gC = nullptr;
      al.accumulate = 1;
    }
    {
      ProfScope ps(c, 1);
      HIPCHK(c, launch_gradient(al, c->cap, c->til.max_ent, c->stream));
    }
    int rc = bt_pass_f(c, f, 3, false, 0.0, f.tilts);  // + s dE/ddiv d(div)/dx
    if (rc) return rc;
  }
  if (dir_mode) {
    c->last_g = g_out;
    c->dir_implicit = false;  // D was written
  }
  bool added_after = n_lbt > 0;
  for (int k = 0; k < 3 && g_out; ++k) {  // module loop: the tilt magnitude modules add their shape gradient into g
    TiltField& f = c->tf[k];
    if (modules & f.mod_tilt) {
      int rc = tilt_pass_f(c, f, 1, false, 0.0);
      if (rc) return rc;
      added_after = true;
    }
    if (k > 0 && (modules & f.mod_dt)) {
      int rc = disk_target_pass(c, f, 1, false, 0.0, f.tilts, true, false);
      if (rc) return rc;
      added_after = true;
    }
  }
  // the gradient kernel's <g, gC> partials predate those additions: take them again of the complete gradient
  if (added_after && g_out && (modules & MS_CON_VOLUME))
    HIPCHK(c, launch_row_dot(c->tile0, c->tile1, c->til.nv, c->til.own, g_out, c->buf[MS_BUF_GC], c->d_partials,
                             c->til.n_tiles, c->stream));
  if (reduce_now) return reduce_slots(c, dir_mode ? MASK_DIR : MASK_GRAD);
  return MS_OK;
}

int phase_direction(ms_ctx* c, int stepper, bool use_history, bool g_finalized = false) {
  const bool use_con = (c->params.modules & MS_CON_VOLUME) != 0;
  c->dir_implicit = false;
  {
  ProfScope ps(c, 2, c->cur_gate, c->cur_gate_want);
  HIPCHK(c, launch_direction(c->tile0, c->tile1, c->til.nv, c->til.own, c->d_vflags, c->buf[MS_BUF_G],
                             c->buf[MS_BUF_GC], c->buf[MS_BUF_D], c->buf[MS_BUF_PG],
                             c->buf[MS_BUF_PD], c->d_scal, use_con ? 1 : 0,
                             (stepper == MS_STEPPER_CG && use_history) ? 1 : 0, c->d_partials,
                             c->til.n_tiles, (g_finalized && !use_con) ? 0 : 1, c->stream, c->cur_gate,
                             c->cur_gate_want,
                             // (the previous direction was an implicit -PG: the kernel derives it, as the fused epilogue does)
                             (stepper == MS_STEPPER_CG && use_history && c->pd_neg_pg) ? 1 : 0,
                             (stepper == MS_STEPPER_CG && c->precond) ? 1 : 0));
  }
  c->last_g = c->buf[MS_BUF_G];
  return reduce_slots(c, MASK_DIR);
}

// host-side write of a slot (a stage mailbox's result moved into the main one): both the host copy and the value
// word, so a later take_mailbox() keeps it
void put_mailbox(ms_ctx* c, int sl, double v) {
  c->h_scal[sl] = v;
  unsigned long long bits;
  memcpy(&bits, &v, sizeof(double));
  __atomic_store_n(&c->h_seq[2 * sl], bits, __ATOMIC_RELAXED);
  // (the entry keeps validating against the ticket it carried: tag = value XOR ticket, see wait_mailbox)
  __atomic_store_n(&c->h_seq[2 * sl + 1], bits ^ c->expected[sl], __ATOMIC_RELAXED);
}

const char* box_name(ms_ctx* c, const void* h_seq);
bool trace_queue();

// a fold found that a gated launch had run on part of its workgroups only (FoldArgs::check_ran): never continue
int check_queue_error(ms_ctx* c) {
  if (!c->h_err) return MS_OK;
  const unsigned long long e = __atomic_load_n(c->h_err, __ATOMIC_ACQUIRE);
  if (e == 0) return MS_OK;
  __atomic_store_n(c->h_err, 0ull, __ATOMIC_RELEASE);  // reported once (the caller gets MS_ERR_STATE for this call)
  char msg[256];
  snprintf(msg, sizeof(msg),
           "line-search queue: a gated launch ran on %llu of its %d workgroups (fold ticket %llu): the workgroups of one "
           "launch did not read the same decision word",
           e & 0xffffffull, c->tile1 - c->tile0, (e >> 24) & 0xffffffffull);
  return fail(c, MS_ERR_STATE, msg);
}

// wait until every entry of a mailbox carries the ticket of the latest fold queued for it; *vals (optional) receives
// the MS_NSCAL slot values, *code the decision entry
int wait_mailbox(ms_ctx* c, unsigned long long* h_seq, const unsigned long long* expected, double* vals,
                 uint32_t* code) {
  // an entry has arrived when value XOR tag is the ticket waited for (k_reduce: post_entry); the value word read for
  // that test is the one handed out
  unsigned long long bits[MS_MB_WORDS];
  auto arrived = [&]() {
    for (int sl = 0; sl < MS_MB_WORDS; ++sl) {
      if (expected[sl] == 0) continue;
      const unsigned long long tag = __atomic_load_n(&h_seq[2 * sl + 1], __ATOMIC_ACQUIRE);
      bits[sl] = __atomic_load_n(&h_seq[2 * sl], __ATOMIC_ACQUIRE);
      if ((bits[sl] ^ tag) != expected[sl]) return false;
    }
    return true;
  };
  // (an empty shard -- 9 tiles over 8 ranks leave ranks 5..7 without one -- waits like any other: k_reduce posts the
  // neutral value of every slot for an empty tile range, and `bits` is only ever filled by arrived())
  bool done = false;
  if (int rc_f = exec_flush(c)) return rc_f;  // (one-tile contexts: what was recorded runs now)
  for (long spin = 0; !done && spin < 20000000L; ++spin) {
    done = arrived();
    if (done) {
      if (g_host_timing.on && spin > 0) {  // (only when the host really waited: the GPU was the one ahead)
        g_host_timing.t_fetch = HostTiming::now();
        g_host_timing.armed = true;
      }
      break;
    }
    __builtin_ia32_pause();
  }
  if (!done) {
    HIPCHK(c, hipStreamSynchronize(S(c)));
    // everything queued has run: an entry that is still behind belongs to a gated fold that found its gate closed
    // although the host expected it to run -- host and device disagreed on an Armijo test.  Never continue on that.
    if (!arrived()) {
      int rc = check_queue_error(c);
      if (rc) return rc;
      int sl = 0;
      while (sl < MS_MB_WORDS - 1 && (expected[sl] == 0 || ((__atomic_load_n(&h_seq[2 * sl], __ATOMIC_ACQUIRE) ^
                                                              __atomic_load_n(&h_seq[2 * sl + 1], __ATOMIC_ACQUIRE)) == expected[sl])))
        ++sl;
      char msg[384];
      snprintf(msg, sizeof(msg),
               "line-search queue: a gated launch the host waited for did not run (mailbox %s, entry %d: ticket %llu "
               "found, %llu expected; latest ticket %llu)",
               box_name(c, h_seq), sl,
               (unsigned long long)(__atomic_load_n(&h_seq[2 * sl], __ATOMIC_ACQUIRE) ^ __atomic_load_n(&h_seq[2 * sl + 1], __ATOMIC_ACQUIRE)),
               (unsigned long long)expected[sl], (unsigned long long)c->ticket);
      ++c->queue_mismatches;
      return fail(c, MS_ERR_STATE, msg);
    }
  }
  int rc = check_queue_error(c);
  if (rc) return rc;
  if (vals)
    for (int sl = 0; sl < MS_NSCAL; ++sl) {
      // (entries nobody waited for keep whatever an earlier fold or put_mailbox left in their value word)
      const unsigned long long b = expected[sl] ? bits[sl] : __atomic_load_n(&h_seq[2 * sl], __ATOMIC_RELAXED);
      memcpy(&vals[sl], &b, sizeof(double));
    }
  if (code) *code = expected[MS_MB_DEC] ? (uint32_t)bits[MS_MB_DEC] : (uint32_t)__atomic_load_n(&h_seq[2 * MS_MB_DEC], __ATOMIC_RELAXED);
  return MS_OK;
}

// k_reduce mirrors every slot it folds into the pinned mailbox and then bumps that slot's
// sequence word; fetching = spinning on those words (a few microseconds less than waking up
// from hipStreamSynchronize).  Falls back to a stream sync after ~50 ms of spinning.
int fetch(ms_ctx* c, uint32_t* code = nullptr) {
  if (trace_queue()) {
    unsigned long long mx = 0;
    for (int sl = 0; sl < MS_MB_WORDS; ++sl) mx = std::max<unsigned long long>(mx, c->expected[sl]);
    fprintf(stderr, "[msq] fetch %s (latest expected ticket %llu)\n", box_name(c, c->h_seq), mx);
  }
  return wait_mailbox(c, c->h_seq, c->expected, c->h_scal, code);
}

// the host's decision against the device's (the code the fold posted next to the energies it decided on)
int verify_decision(ms_ctx* c, uint32_t host_code, uint32_t dev_code, const char* where) {
  if (host_code == dev_code) return MS_OK;
  ++c->queue_mismatches;
  char msg[256];
  snprintf(msg, sizeof(msg), "line-search queue: host and device took different decisions at %s (host %u, device %#x)",
           where, host_code, dev_code);
  return fail(c, MS_ERR_STATE, msg);
}

double penalty_energy(const ms_ctx* c, double V) {
  if (!(c->params.modules & MS_MOD_VOLUME_PENALTY)) return 0.0;
  const double delta = V - c->params.target_volume;
  return 0.5 * c->params.volume_stiffness * (delta * delta);
}

// energies from the pinned mailbox: {surface, bending, penalty, tilt}
void energies_from_mailbox(const ms_ctx* c, double e[4]) {
  e[0] = (c->params.modules & MS_MOD_SURFACE) ? c->h_scal[MS_S_ESURF] : 0.0;
  e[1] = (c->params.modules & MS_MOD_BENDING) ? c->h_scal[MS_S_EBEND] : 0.0;
  if (c->params.modules & MS_MOD_BENDING_TILT) e[1] += c->h_scal[MS_S_EBT];
  e[2] = penalty_energy(c, c->h_scal[MS_S_VOL]);
  e[3] = (c->params.modules & MS_MOD_TILT) ? c->h_scal[MS_S_ETILT] : 0.0;
  if (c->params.modules & MS_MOD_TILT_SMOOTH) e[3] += c->h_scal[MS_S_ETS];
  for (int l = 1; l <= 2; ++l) {
    const TiltField& f = c->tf[l];
    if (c->params.modules & f.mod_tilt) e[3] += c->h_scal[f.s_etilt];
    if (c->params.modules & f.mod_smooth) e[3] += c->h_scal[f.s_ets];
    if (c->params.modules & f.mod_bt) e[1] += c->h_scal[f.s_ebt];
    if (c->params.modules & f.mod_dt) e[3] += c->h_scal[f.s_edt];
  }
}

// gradient assembly at x: energy pass (+factors), gradient pass, finalize via
// the direction kernel (projection + fixed rows), everything queued async.
// skip_energy: the accepted trial of the previous step already was this energy pass (same
// kernel, same x: its factors are in fK/fA, its scalars in d_scal and in the mailbox).
int queue_energy_and_gradient(ms_ctx* c, int stepper, bool use_history, bool skip_energy = false) {
  const uint32_t mods = c->params.modules;
  // lambda needs a global reduction first; the tilt module adds into g after K_C
  // (and a preconditioned direction -- conjugate_gradient.py:74-76 -- is the direction kernel's)
  const bool constraint = (mods & (MS_CON_VOLUME | MS_TILT_SHAPE_MODS)) != 0 || (stepper == MS_STEPPER_CG && c->precond);
  // K_C reads the reduced volume (already reduced when the energy pass is skipped)
  const bool penalty = skip_energy || (mods & MS_MOD_VOLUME_PENALTY) != 0;
  int rc = MS_OK;
  if (!skip_energy) rc = phase_energy(c, mods, false, 0.0, false, false, true, /*reduce_now=*/penalty);
  if (rc) return rc;
  if (!constraint) {
    // no row to project out: the direction pass rides in K_C's epilogue, one reduce for all
    const int dir_mode = (stepper == MS_STEPPER_CG && use_history) ? 2 : 1;
    rc = phase_gradient(c, mods, c->buf[MS_BUF_G], false, dir_mode, /*reduce_now=*/false);
    if (rc) return rc;
    return reduce_slots(c, (penalty ? 0u : energy_mask(mods)) | MASK_DIR);
  }
  rc = phase_gradient(c, mods, c->buf[MS_BUF_G], false, 0, /*reduce_now=*/false);
  if (rc) return rc;
  // (<g,gC> / <gC,gC> exist only with a row or a tilt module behind K_C)
  const uint32_t gmask = (mods & (MS_CON_VOLUME | MS_TILT_SHAPE_MODS)) ? MASK_GRAD : 0u;
  const uint32_t fmask = (penalty ? 0u : energy_mask(mods)) | gmask;
  if (fmask) rc = reduce_slots(c, fmask);
  if (rc) return rc;
  return phase_direction(c, stepper, use_history);
}

template <typename Tp>
int upload(ms_ctx* c, Tp** dst, const std::vector<Tp>& src, size_t min_elems = 1) {
  size_t n = src.size() > min_elems ? src.size() : min_elems;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(Tp)));
  if (!src.empty())
    HIPCHK(c, hipMemcpy(*dst, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice));
  return MS_OK;
}

int ext_to_patch(ms_ctx* c, const double* host, double* dst, int ncomp) {
  const size_t bytes = sizeof(double) * (size_t)c->til.nv * ncomp;
  HIPCHK(c, hipMemcpyAsync(c->d_stage, host, bytes, hipMemcpyHostToDevice, S(c)));
  HIPCHK(c, launch_permute_in(c->til.nv, c->d_perm, c->d_stage, dst, ncomp, c->stream));
  HIPCHK(c, hipStreamSynchronize(S(c)));
  return MS_OK;
}

int patch_to_ext(ms_ctx* c, const double* src, double* host, int ncomp) {
  const size_t bytes = sizeof(double) * (size_t)c->til.nv * ncomp;
  HIPCHK(c, launch_permute_out(c->til.nv, c->d_perm, src, c->d_stage, ncomp, c->stream));
  HIPCHK(c, hipMemcpyAsync(host, c->d_stage, bytes, hipMemcpyDeviceToHost, S(c)));
  HIPCHK(c, hipStreamSynchronize(S(c)));
  return MS_OK;
}

}  // namespace

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <typename Tp>
  Tp* as() {
    return static_cast<Tp*>(p);
  }
};
#define SEAM_HIP(call)                                                         \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return fail_hip(nullptr, e_, #call);                 \
  } while (0)
}  // namespace

extern "C" {

const char* ms_version(void) { return "membrane_hip 0.1 (gfx950)"; }

int ms_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return MS_ERR_HIP;
  }
  return n;
}

const char* ms_last_error(const ms_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

int ms_create(ms_ctx** out, int device, int nv, int nf, const double* positions,
              const int32_t* tri, const uint8_t* fixed, const uint8_t* boundary,
              const uint8_t* body_facets, int tile_vertices, int shard_rank, int shard_count) {
  if (!out) return fail(nullptr, MS_ERR_INVALID, "ms_create: out is NULL");
  *out = nullptr;
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count)
    return fail(nullptr, MS_ERR_INVALID, "ms_create: bad shard_rank/shard_count");
  ms_ctx* c = new (std::nothrow) ms_ctx();
  if (!c) return fail(nullptr, MS_ERR_NOMEM, "ms_create: out of host memory");
  std::string err;
  int rc = build_tiling(nv, nf, positions, tri, body_facets, tile_vertices, shard_count, c->til, err);
  if (rc != MS_OK) {
    delete c;
    return fail(nullptr, rc, err);
  }
  c->device = device;
  c->shard_rank = shard_rank;
  c->shard_count = shard_count;
  const Tiling& t = c->til;
  c->tile0 = std::min(t.n_tiles, shard_rank * t.tiles_per_shard);
  c->tile1 = std::min(t.n_tiles, (shard_rank + 1) * t.tiles_per_shard);
  c->cap = t.own + t.max_halo;
  {
    const size_t le = energy_lds_bytes(t.T, c->cap, t.max_ent, true, true, true);
    const size_t lg = gradient_lds_bytes(t.T, c->cap, t.max_ent, true, true);
    if (le > 160 * 1024 || lg > 160 * 1024) {
      delete c;
      return fail(nullptr, MS_ERR_TILE_CAPACITY,
                  "ms_create: a vertex patch (tile + halo) exceeds the 160 KiB LDS of a CU; "
                  "use a smaller tile_vertices");
    }
  }
#define CREATE_CHK(call)                                       \
  do {                                                         \
    int rc_ = (call);                                          \
    if (rc_ != MS_OK) {                                        \
      std::string m_ = c->err;                                 \
      ms_destroy(c);                                           \
      return fail(nullptr, rc_, m_);                           \
    }                                                          \
  } while (0)
#define CREATE_HIP(call)                                                         \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);        \
      ms_destroy(c);                                                             \
      return fail(nullptr, MS_ERR_HIP, m_);                                      \
    }                                                                            \
  } while (0)
  CREATE_HIP(hipSetDevice(device));
  CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  CREATE_CHK(upload(c, &c->d_perm, t.perm));
  CREATE_CHK(upload(c, &c->d_tile_facet_off, t.tile_facet_off));
  CREATE_CHK(upload(c, &c->d_tile_facets, t.tile_facets));
  if (t.own + t.max_halo <= 1024) {
    std::vector<uint32_t> packed(t.tile_facets.size());
    for (size_t i = 0; i < packed.size(); ++i) {
      const TileFacet& f = t.tile_facets[i];
      packed[i] = (uint32_t)f.l0 | ((uint32_t)f.l1 << 10) | ((uint32_t)f.l2 << 20) | ((uint32_t)(f.flags & 3u) << 30);
    }
    CREATE_CHK(upload(c, &c->d_tile_facets32, packed));
  }
  CREATE_CHK(upload(c, &c->d_tile_halo_off, t.tile_halo_off));
  CREATE_CHK(upload(c, &c->d_halo_ids, t.halo_ids));
  CREATE_CHK(upload(c, &c->d_tile_ent_off, t.tile_ent_off));
  CREATE_CHK(upload(c, &c->d_tile_voff, t.tile_voff));
  CREATE_CHK(upload(c, &c->d_vent, t.vent));
  {
    std::vector<double> ones(t.tile_facets.size(), 1.0);
    CREATE_CHK(upload(c, &c->d_tf_gamma, ones));
  }
  {
    std::vector<uint8_t> fl((size_t)t.nvp, VF_FIXED);  // padded rows never move
    for (int i = 0; i < nv; ++i) {
      const int e = t.perm[i];
      uint8_t f = 0;
      if (fixed && fixed[e]) f |= VF_FIXED;
      if (boundary && boundary[e]) {
        f |= VF_BOUNDARY;
        c->has_boundary = true;
      }
      fl[i] = f;
    }
    CREATE_CHK(upload(c, &c->d_vflags, fl));
    c->h_vflags = fl;
    std::vector<double> zeros((size_t)t.nvp, 0.0);
    CREATE_CHK(upload(c, &c->d_kappa, zeros));
    CREATE_CHK(upload(c, &c->d_c0, zeros));
  }
  // state: 8 (nvp,3) vectors + fA (nvp,2)
  {
    const size_t n3 = 3 * (size_t)t.nvp;
    const size_t total = 8 * n3 + 2 * (size_t)t.nvp;
    CREATE_HIP(hipMalloc(reinterpret_cast<void**>(&c->state), total * sizeof(double)));
    CREATE_HIP(hipMemset(c->state, 0, total * sizeof(double)));
    for (int b = 0; b <= MS_BUF_FK; ++b) c->buf[b] = c->state + (size_t)b * n3;
    c->buf[MS_BUF_FA] = c->state + 8 * n3;
  }
  CREATE_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_partials),
                       sizeof(double) * MS_NPART * (size_t)std::max(1, t.n_tiles)));
  CREATE_HIP(hipMemset(c->d_partials, 0, sizeof(double) * MS_NPART * (size_t)std::max(1, t.n_tiles)));
  CREATE_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_scal), sizeof(double) * MS_NSCAL));
  CREATE_HIP(hipMemset(c->d_scal, 0, sizeof(double) * MS_NSCAL));
  c->buf[MS_BUF_SCAL] = c->d_scal;
  c->h_scal = static_cast<double*>(calloc(MS_NSCAL, sizeof(double)));
  CREATE_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_seq), sizeof(unsigned long long) * 2 * MS_MB_WORDS,
                           hipHostMallocMapped));
  memset(c->h_seq, 0, sizeof(unsigned long long) * 2 * MS_MB_WORDS);
  CREATE_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_seq), c->h_seq, 0));
  CREATE_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_err), sizeof(unsigned long long) * 8, hipHostMallocMapped));
  memset(c->h_err, 0, sizeof(unsigned long long) * 8);
  CREATE_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_err), c->h_err, 0));
  CREATE_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_stage), sizeof(double) * 3 * (size_t)nv));
  {
    // boundary lists of every rank (each rank derives all of them from the shared tiling)
    const int W = shard_count;
    const int rows_per = t.tiles_per_shard * t.own;
    std::vector<std::vector<int32_t>> lists((size_t)W);
    std::vector<int32_t> my_halo;
    for (int tile = 0; tile < t.n_tiles; ++tile) {
      const int tr = tile / t.tiles_per_shard;
      for (int h = t.tile_halo_off[tile]; h < t.tile_halo_off[tile + 1]; ++h) {
        const int32_t v = t.halo_ids[h];
        const int owner = v / rows_per;
        if (owner == tr) continue;
        lists[(size_t)owner].push_back(v);
        if (tr == shard_rank) my_halo.push_back(v);
      }
    }
    std::vector<int32_t> flat;
    c->bnd_off.assign((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
      auto& l = lists[(size_t)r];
      std::sort(l.begin(), l.end());
      l.erase(std::unique(l.begin(), l.end()), l.end());
      c->bnd_off[(size_t)r + 1] = c->bnd_off[(size_t)r] + (int32_t)l.size();
      c->bnd_max = std::max(c->bnd_max, (int)l.size());
      flat.insert(flat.end(), l.begin(), l.end());
    }
    std::sort(my_halo.begin(), my_halo.end());
    my_halo.erase(std::unique(my_halo.begin(), my_halo.end()), my_halo.end());
    c->n_halo_rows = (int)my_halo.size();
    CREATE_CHK(upload(c, &c->d_bnd_rows, flat));
    CREATE_CHK(upload(c, &c->d_bnd_off, c->bnd_off));
    CREATE_CHK(upload(c, &c->d_halo_rows, my_halo));
    CREATE_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_scal_all), sizeof(double) * MS_NSCAL * (size_t)W));
    CREATE_HIP(hipMemset(c->d_scal_all, 0, sizeof(double) * MS_NSCAL * (size_t)W));
  }
  {
    TiltField& f0 = c->tf[0];
    f0.fixed_bit = VF_TILT_FIXED; f0.mod_tilt = MS_MOD_TILT; f0.mod_smooth = MS_MOD_TILT_SMOOTH;
    f0.s_etilt = MS_S_ETILT; f0.s_ets = MS_S_ETS; f0.s_gn2 = MS_S_TGNORM2; f0.s_rz = MS_S_TRZ;
    TiltField& f1 = c->tf[1];
    f1.fixed_bit = VF_TILT_FIXED_IN; f1.mod_tilt = MS_MOD_TILT_IN; f1.mod_smooth = MS_MOD_TILT_SMOOTH_IN;
    f1.s_etilt = MS_S_ETILT_IN; f1.s_ets = MS_S_ETS_IN; f1.s_gn2 = MS_S_TGNORM2_IN; f1.s_rz = MS_S_TRZ_IN;
    f1.mod_bt = MS_MOD_BENDING_TILT_IN; f1.s_ebt = MS_S_EBT_IN; f1.div_sign = -1.0;  // bending_tilt_in.py:46
    f0.mod_bt = MS_MOD_BENDING_TILT; f0.s_ebt = MS_S_EBT;
    f1.mod_dt = MS_MOD_TILT_DISK_TARGET_IN; f1.s_edt = MS_S_EDT_IN; f1.s_dtr = MS_S_DTR_IN;
    TiltField& f2 = c->tf[2];
    f2.fixed_bit = VF_TILT_FIXED_OUT; f2.mod_tilt = MS_MOD_TILT_OUT; f2.mod_smooth = MS_MOD_TILT_SMOOTH_OUT;
    f2.s_etilt = MS_S_ETILT_OUT; f2.s_ets = MS_S_ETS_OUT; f2.s_gn2 = MS_S_TGNORM2_OUT; f2.s_rz = MS_S_TRZ_OUT;
    f2.mod_bt = MS_MOD_BENDING_TILT_OUT; f2.s_ebt = MS_S_EBT_OUT; f2.div_sign = 1.0;
    f2.mod_dt = MS_MOD_TILT_DISK_TARGET_OUT; f2.s_edt = MS_S_EDT_OUT; f2.s_dtr = MS_S_DTR_OUT;
  }
  if (const char* pe = getenv("MS_PAIR")) {
    c->pair_enable = atoi(pe) != 0;
    c->pair_force = atoi(pe) >= 2 ? std::min(atoi(pe), 4) : 0;  // 4: triple launches whenever possible
  }
  c->speculate = !(getenv("MS_SPECULATE") != nullptr && atoi(getenv("MS_SPECULATE")) == 0);
  c->ahead_enable = !(getenv("MS_AHEAD") != nullptr && atoi(getenv("MS_AHEAD")) == 0);
  c->escalate = !(getenv("MS_ESCALATE") != nullptr && atoi(getenv("MS_ESCALATE")) == 0);
  if (getenv("MS_ESCALATE") != nullptr && atoi(getenv("MS_ESCALATE")) > 0) c->escalate_after = atoi(getenv("MS_ESCALATE"));
  c->ls_reset = !(getenv("MS_LS_RESET") != nullptr && atoi(getenv("MS_LS_RESET")) == 0);
  c->no_fast = variant_env("MS_NO_FAST") != nullptr && atoi(variant_env("MS_NO_FAST")) != 0;
  c->pair_lean_enable = !(getenv("MS_PAIR_LEAN") != nullptr && atoi(getenv("MS_PAIR_LEAN")) == 0);
  c->deterministic = getenv("MS_DETERMINISTIC") != nullptr && atoi(getenv("MS_DETERMINISTIC")) != 0;
  c->resident_enable = !(getenv("MS_RESIDENT") != nullptr && atoi(getenv("MS_RESIDENT")) == 0);
  c->params.modules = MS_MOD_SURFACE;
  c->params.bending_model = MS_BEND_HELFRICH;
  c->params.bending_grad_mode = MS_GRAD_ANALYTIC;
  c->params.volume_stiffness = 1000.0;
  c->params.target_volume = 0.0;
  c->last_g = c->buf[MS_BUF_G];
  // A mesh of ONE tile: every kernel is one workgroup and a step is launches and host round trips -- record the launches
  // and run them pack by pack in one workgroup (k_exec).  MS_EXEC=0 keeps the launch-per-kernel path (A/B, tests).
  c->exec_wanted = t.n_tiles == 1 && t.T == 256 && t.own == 256 && shard_count == 1 &&
                   !(getenv("MS_EXEC") != nullptr && atoi(getenv("MS_EXEC")) == 0);
  if (c->exec_wanted) {
    c->exec.stream = c->stream;
    c->exec.T = t.T;
    exec_attach(&c->exec);
    c->exec_on = true;
    c->pair_enable = false;  // (several trials per launch buy nothing inside one workgroup)
    c->exec_relax = !(getenv("MS_EXEC_RELAX") != nullptr && atoi(getenv("MS_EXEC_RELAX")) == 0);
  }
  CREATE_CHK(ext_to_patch(c, positions, c->buf[MS_BUF_X], 3));
#undef CREATE_CHK
#undef CREATE_HIP
  *out = c;
  return MS_OK;
}

void ms_destroy(ms_ctx* c) {
  if (g_host_timing.on && g_host_timing.n > 0) {
    fprintf(stderr, "[ms host timing] %ld waits followed by an energy launch: mailbox seen -> launch call returned %.2f us "
                    "on average (max %.1f), of which the launch call itself %.2f us\n",
            g_host_timing.n, g_host_timing.sum_gap / g_host_timing.n, g_host_timing.max_gap,
            g_host_timing.sum_launch / g_host_timing.n);
    g_host_timing.n = 0;
    g_host_timing.sum_gap = g_host_timing.sum_launch = g_host_timing.max_gap = 0.0;
  }
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->exec_on) {
    (void)c->exec.flush();
    exec_detach(&c->exec);
    c->exec_on = false;
  }
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (!c->own_state) c->state = nullptr;
  void* ptrs[] = {c->d_perm, c->d_tile_facet_off, c->d_tile_facets, c->d_tile_facets32, c->d_tf_gamma, c->d_volgrad_cache,
                  c->d_tile_halo_off, c->d_halo_ids, c->d_tile_ent_off, c->d_tile_voff, c->d_vent,
                  c->d_vflags, c->d_kappa, c->d_c0, c->tf[0].tilts, c->tf[0].grad, c->tf[0].trial, c->d_bt_vert, c->d_tn, c->tf[0].dir, c->tf[0].minv,
                  c->tf[1].tilts, c->tf[1].grad, c->tf[1].trial, c->tf[1].dir, c->tf[1].minv,
                  c->tf[2].tilts, c->tf[2].grad, c->tf[2].trial, c->tf[2].dir, c->tf[2].minv,
                  c->tf[1].kappa, c->tf[1].c0, c->tf[1].bt_vert, c->tf[2].kappa, c->tf[2].c0, c->tf[2].bt_vert,
                  c->tf[1].disk, c->tf[1].diff, c->tf[2].disk, c->tf[2].diff, c->tf[0].va, c->tf[1].va, c->tf[2].va,
                  c->state, c->d_partials, c->d_scal, c->d_stage, c->d_bnd_rows, c->d_bnd_off,
                  c->d_halo_rows, c->d_scal_all};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* q : c->peer_opened) (void)hipIpcCloseMemHandle(q);
  if (c->d_peer_slab) (void)hipFree(c->d_peer_slab);
  if (c->d_peer_flag) (void)hipFree(c->d_peer_flag);
  if (c->peer_aux) (void)hipStreamDestroy(c->peer_aux);
  if (c->d_peer_flagtab) (void)hipFree(c->d_peer_flagtab);
  if (c->d_peer_arrived) (void)hipFree(c->d_peer_arrived);
  if (c->comm) shard_comm_destroy(c->comm);
  if (c->d_xsend) (void)hipFree(c->d_xsend);
  if (c->d_xrecv) (void)hipFree(c->d_xrecv);
  if (c->h_scal_all) (void)hipHostFree(c->h_scal_all);
  if (c->h_xseq) (void)hipHostFree(c->h_xseq);
  for (int p = 0; p < 2; ++p) {
    for (auto& m : c->spec[p]) {
      free(m.h_scal);
      if (m.h_seq) (void)hipHostFree(m.h_seq);
    }
    for (ms_ctx::Mailbox* m : {&c->grad_mb[p], &c->first_mb[p]}) {
      free(m->h_scal);
      if (m->h_seq) (void)hipHostFree(m->h_seq);
    }
  }
  if (c->d_dec) (void)hipFree(c->d_dec);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->d_prof_ran) (void)hipFree(c->d_prof_ran);
  if (c->exec.d_stamps) (void)hipFree(c->exec.d_stamps);
  if (c->d_res_partials) (void)hipFree(c->d_res_partials);
  if (c->d_res_bar) (void)hipFree(c->d_res_bar);
  if (c->d_res_log) (void)hipFree(c->d_res_log);
  if (c->d_res_result) (void)hipFree(c->d_res_result);
  for (int l = 0; l < 3; ++l)
    if (c->tf[l].dt_target) (void)hipFree(c->tf[l].dt_target);
  if (c->d_relax_cells) (void)hipFree(c->d_relax_cells);
  if (c->h_relax_box) (void)hipHostFree(c->h_relax_box);
  for (auto& sd : c->side) {
    if (sd.partials) (void)hipFree(sd.partials);
    if (sd.scal) (void)hipFree(sd.scal);
    for (auto& m : sd.mb) {
      free(m.h_scal);
      if (m.h_seq) (void)hipHostFree(m.h_seq);
    }
  }
  for (double* q : {c->xt2, c->fK2, c->fA2, c->xt3, c->fK3, c->fA3})
    if (q) (void)hipFree(q);
  free(c->h_scal);
  if (c->h_seq) (void)hipHostFree(c->h_seq);
  for (auto& r : c->prof_pending) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : c->prof_pool) (void)hipEventDestroy(e);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int ms_set_stream(ms_ctx* c, void* hip_stream) {
  if (!c) return MS_ERR_INVALID;
  HIPCHK(c, hipStreamSynchronize(S(c)));
  if (c->own_stream && c->stream) {
    HIPCHK(c, hipStreamDestroy(c->stream));
    c->own_stream = false;
  }
  if (hip_stream) {
    c->stream = static_cast<hipStream_t>(hip_stream);
  } else {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  c->exec.stream = c->stream;  // (nothing is pending: S(c) above launched it)
  return MS_OK;
}

int ms_set_surface_tension(ms_ctx* c, const double* gamma) {
  if (!c || !gamma) return fail(c, MS_ERR_INVALID, "ms_set_surface_tension: NULL argument");
  const Tiling& t = c->til;
  HIPCHK(c, hipStreamSynchronize(S(c)));
  std::vector<double> g(t.tile_facets.size());
  for (size_t p = 0; p < g.size(); ++p) g[p] = gamma[t.tile_facet_ext[p]];
  c->gamma_uniform = true;
  c->gamma_const = t.nf > 0 ? gamma[0] : 1.0;
  for (int f = 1; f < t.nf && c->gamma_uniform; ++f) c->gamma_uniform = gamma[f] == c->gamma_const;
  if (!g.empty())
    HIPCHK(c, hipMemcpy(c->d_tf_gamma, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return MS_OK;
}

int ms_set_bending_params(ms_ctx* c, const double* kappa, const double* c0) {
  if (!c || !kappa || !c0) return fail(c, MS_ERR_INVALID, "ms_set_bending_params: NULL argument");
  const Tiling& t = c->til;
  HIPCHK(c, hipStreamSynchronize(S(c)));
  std::vector<double> k((size_t)t.nvp, 0.0), z((size_t)t.nvp, 0.0);
  c->kc_uniform = true;
  c->kappa_const = kappa[0];
  c->c0_const = c0[0];
  for (int i = 0; i < t.nv; ++i) {
    k[i] = kappa[t.perm[i]];
    z[i] = c0[t.perm[i]];
    c->kc_uniform = c->kc_uniform && kappa[i] == c->kappa_const && c0[i] == c->c0_const;
  }
  HIPCHK(c, hipMemcpy(c->d_kappa, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_c0, z.data(), z.size() * sizeof(double), hipMemcpyHostToDevice));
  c->factors_valid = false;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return MS_OK;
}

int ms_set_params(ms_ctx* c, const ms_params* p) {
  if (!c || !p) return fail(c, MS_ERR_INVALID, "ms_set_params: NULL argument");
  if (p->bending_model != MS_BEND_HELFRICH && p->bending_model != MS_BEND_WILLMORE)
    return fail(c, MS_ERR_INVALID, "ms_set_params: bad bending_model");
  if (p->bending_grad_mode != MS_GRAD_ANALYTIC && p->bending_grad_mode != MS_GRAD_APPROX)
    return fail(c, MS_ERR_INVALID, "ms_set_params: bad bending_grad_mode");
  if ((p->modules & MS_MOD_BENDING) && (p->modules & MS_MOD_BENDING_TILT))
    return fail(c, MS_ERR_INVALID, "ms_set_params: bending and bending_tilt are mutually exclusive");
  if ((p->modules & MS_LEAFLET_BT) && (p->modules & (MS_MOD_BENDING | MS_MOD_BENDING_TILT)))
    return fail(c, MS_ERR_INVALID, "ms_set_params: bending / bending_tilt together with bending_tilt_in/out is outside the device path");
  if ((p->modules & MS_MOD_BENDING_TILT) && c->shard_count != 1)
    return fail(c, MS_ERR_STATE, "the bending_tilt module is not sharded yet (single GPU only)");
  if ((p->modules & MS_MOD_BENDING_TILT) && !c->d_bt_vert) {
    const size_t bytes = sizeof(double) * 4 * (size_t)c->til.nvp;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_bt_vert), bytes));
    HIPCHK(c, hipMemset(c->d_bt_vert, 0, bytes));
  }
  c->params = *p;
  c->factors_valid = false;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return MS_OK;
}

int ms_set_tilts(ms_ctx* c, const double* tilts, double tilt_rigidity) {
  if (!c || !tilts) return fail(c, MS_ERR_INVALID, "ms_set_tilts: NULL argument");
  const size_t bytes = sizeof(double) * 3 * (size_t)c->til.nvp;
  if (!c->tf[0].tilts) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->tf[0].tilts), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->tf[0].grad), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->tf[0].trial), bytes));
    HIPCHK(c, hipMemset(c->tf[0].trial, 0, bytes));
    HIPCHK(c, hipMemset(c->tf[0].tilts, 0, bytes));
    HIPCHK(c, hipMemset(c->tf[0].grad, 0, bytes));
  }
  c->tf[0].k_tilt = tilt_rigidity;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return ext_to_patch(c, tilts, c->tf[0].tilts, 3);
}

int ms_get_tilts(ms_ctx* c, double* tilts) {
  if (!c || !tilts || !c->tf[0].tilts) return fail(c, MS_ERR_INVALID, "ms_get_tilts: no tilts set");
  return patch_to_ext(c, c->tf[0].tilts, tilts, 3);
}

int ms_get_tilt_gradient(ms_ctx* c, double* tilt_grad) {
  if (!c || !tilt_grad || !c->tf[0].grad) return fail(c, MS_ERR_INVALID, "ms_get_tilt_gradient: no tilts set");
  return patch_to_ext(c, c->tf[0].grad, tilt_grad, 3);
}

int ms_angle_defects(ms_ctx* c, double* defects) {
  if (!c || !defects) return fail(c, MS_ERR_INVALID, "ms_angle_defects: NULL argument");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "ms_angle_defects: single shard only");
  const Tiling& t = c->til;
  double* d_out = nullptr;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(double) * (size_t)std::max<int64_t>(1, t.nvp)));
  HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)std::max<int64_t>(1, t.nvp), S(c)));
  TiltArgs a;
  a.fields = nullptr;
  a.fields_rows = 0;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = nullptr;
  a.alpha = 0.0;
  a.tilts = c->buf[MS_BUF_X];  // (read, not used)
  a.tilts_out = nullptr;
  a.k_tilt = 0.0;
  a.g = nullptr;
  a.tilt_grad = nullptr;
  a.minv = d_out;
  a.partials = c->d_partials;
  a.e_slot = MS_S_ETILT;
  a.consistent = 0;
  a.tg_accumulate = 0;
  a.cons_tilt_grad = 0;
  a.va_out = nullptr;
  hipError_t e = launch_tilt(a, 4, c->cap, t.max_ent, c->stream);
  int rc = MS_OK;
  if (e != hipSuccess) rc = fail(c, MS_ERR_HIP, std::string("ms_angle_defects: ") + hipGetErrorString(e));
  if (rc == MS_OK) rc = patch_to_ext(c, d_out, defects, 1);
  (void)hipStreamSynchronize(S(c));
  (void)hipFree(d_out);
  return rc;
}

int ms_curvature_fields(ms_ctx* c, double* mean_curvature_normal, double* h_area_anglesum, double* defect_kg,
                        double* principal) {
  if (!c) return MS_ERR_INVALID;
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "ms_curvature_fields: single shard only");
  const Tiling& t = c->til;
  const size_t plane = 3 * (size_t)std::max<int64_t>(1, t.nvp);
  double* d_out = nullptr;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(double) * 4 * plane));
  HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(double) * 4 * plane, S(c)));
  TiltArgs a;
  a.m = device_mesh(c);
  a.tile0 = c->tile0;
  a.tile1 = c->tile1;
  a.x = c->buf[MS_BUF_X];
  a.d = nullptr;
  a.alpha = 0.0;
  a.tilts = c->buf[MS_BUF_X];  // (read, not used)
  a.tilts_out = nullptr;
  a.k_tilt = 0.0;
  a.g = nullptr;
  a.tilt_grad = nullptr;
  a.minv = nullptr;
  a.partials = c->d_partials;
  a.e_slot = MS_S_ETILT;
  a.consistent = 0;
  a.tg_accumulate = 0;
  a.cons_tilt_grad = 0;
  a.va_out = nullptr;
  a.fields = d_out;
  a.fields_rows = t.nvp;
  hipError_t e = launch_tilt(a, 5, c->cap, t.max_ent, c->stream);
  int rc = MS_OK;
  if (e != hipSuccess) rc = fail(c, MS_ERR_HIP, std::string("ms_curvature_fields: ") + hipGetErrorString(e));
  double* outs[4] = {mean_curvature_normal, h_area_anglesum, defect_kg, principal};
  for (int k = 0; k < 4 && rc == MS_OK; ++k)
    if (outs[k]) rc = patch_to_ext(c, d_out + (size_t)k * plane, outs[k], 3);
  (void)hipStreamSynchronize(S(c));
  (void)hipFree(d_out);
  return rc;
}

int ms_project_tilts_to_tangent(ms_ctx* c) {
  if (!c) return MS_ERR_INVALID;
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "tilt passes are single-shard only");
  TiltField* fl[3];
  const int n = active_fields(c, c->params.modules, fl);
  for (int k = 0; k < n; ++k) {  // every field the module set reads (geometry/mesh.py:788-814)
    int rc = tilt_pass_f(c, *fl[k], 2, false, 0.0);
    if (rc) return rc;
  }
  if (n == 0) {
    int rc = tilt_pass(c, 2, false, 0.0);
    if (rc) return rc;
  }
  return fetch(c);
}

int ms_set_deterministic(ms_ctx* c, int on) {
  if (!c) return MS_ERR_INVALID;
  if (c->deterministic == (on != 0)) return MS_OK;
  c->deterministic = on != 0;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->factors_valid = false;
  return MS_OK;
}

int ms_set_tilt_smoothness(ms_ctx* c, double k_smooth) {
  if (!c) return MS_ERR_INVALID;
  c->tf[0].k_smooth = k_smooth;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  return MS_OK;
}

int ms_set_tilt_fixed(ms_ctx* c, const uint8_t* tilt_fixed) {
  if (!c) return fail(c, MS_ERR_INVALID, "ms_set_tilt_fixed: NULL context");
  const Tiling& t = c->til;
  for (int i = 0; i < t.nv; ++i) {
    uint8_t f = c->h_vflags[(size_t)i] & (uint8_t)~VF_TILT_FIXED;
    if (tilt_fixed && tilt_fixed[t.perm[i]]) f |= VF_TILT_FIXED;
    c->h_vflags[(size_t)i] = f;
  }
  c->tf[0].any_free = t.nv == 0;
  for (int i = 0; i < t.nv && !c->tf[0].any_free; ++i) c->tf[0].any_free = !(c->h_vflags[(size_t)i] & VF_TILT_FIXED);
  HIPCHK(c, hipStreamSynchronize(S(c)));
  HIPCHK(c, hipMemcpy(c->d_vflags, c->h_vflags.data(), c->h_vflags.size(), hipMemcpyHostToDevice));
  return MS_OK;
}

namespace {
// Energy of the tilt-reading modules (+ dense tilt gradients) on the current x, reading each
// field's `trial` or stored tilts (runtime/evaluation_manager.py:303-462 for the single field,
// :537-742 for the leaflets; needs d_bt_vert valid when bending_tilt is on).  With positions
// frozen the leaflet magnitude modules take the lumped vertex-area form whatever their mass mode
// (evaluation_manager.py:565-581, 663-695: the relaxation always passes tilt_vertex_areas).
int tilt_eval(ms_ctx* c, bool trial, bool gradient) {
  const uint32_t mods = c->params.modules;
  int rc = MS_OK;
  uint32_t mask = 0;
  const size_t b3 = sizeof(double) * 3 * (size_t)c->til.nvp;
  if (mods & MS_TILT_MODS) {
    const double* tilts = trial ? c->tf[0].trial : c->tf[0].tilts;
    // energy only, both modules of the field: the bending_tilt kernel sums the tilt-magnitude energy as well (it has
    // the tilt rows staged; a non-zero rigidity, or the slot would not be written)
    const bool fuse = !gradient && (mods & MS_MOD_TILT) && (mods & MS_MOD_BENDING_TILT) && c->tf[0].k_tilt != 0.0 &&
                      !c->tf[0].consistent;
    if ((mods & MS_MOD_TILT) && !fuse) {
      rc = tilt_pass(c, gradient ? 1 : 0, false, 0.0, tilts, nullptr, /*shape_gradient=*/false);
      if (rc) return rc;
      mask |= 1u << MS_S_ETILT;
    } else if (fuse) {
      mask |= 1u << MS_S_ETILT;
    } else if (gradient) {
      if (int rz = zero_doubles(c, c->tf[0].grad, b3)) return rz;
    }
    if (mods & MS_MOD_BENDING_TILT) {
      rc = bt_pass(c, gradient ? 2 : 0, false, 0.0, tilts, fuse);
      if (rc) return rc;
      mask |= 1u << MS_S_EBT;
    }
    if (mods & MS_MOD_TILT_SMOOTH) {
      rc = ts_pass(c, gradient ? 1 : 0, false, 0.0, tilts);
      if (rc) return rc;
      mask |= 1u << MS_S_ETS;
    }
  }
  for (int l = 1; l <= 2; ++l) {
    TiltField& f = c->tf[l];
    if (!(mods & (f.mod_tilt | f.mod_smooth | f.mod_bt | f.mod_dt))) continue;
    const double* tilts = trial ? f.trial : f.tilts;
    if ((mods & f.mod_tilt) && c->relax_va_valid && f.va) {
      // positions frozen, vertex areas at hand: the reference's own form of this evaluation, one streaming pass
      ProfScope ps(c, 6);
      HIPCHK(c, launch_tvec(4, c->tile0, c->tile1, c->til.nv, c->til.own, c->d_vflags, f.grad, f.va, f.dir, tilts,
                            nullptr, nullptr, nullptr, f.k_tilt, gradient ? 1 : 0, c->d_partials, c->til.n_tiles,
                            c->stream, f.fixed_bit, f.s_etilt, f.s_rz));
      mask |= 1u << f.s_etilt;
    } else if (mods & f.mod_tilt) {
      // (module form -- ms_leaflet_tilt_energy_and_gradient_ex: the field's own mass mode, tilt gradient included)
      rc = tilt_pass_f(c, f, gradient ? 1 : 0, false, 0.0, tilts, nullptr, /*shape_gradient=*/false,
                       /*lumped=*/!c->tilt_module_form);
      if (rc) return rc;
      mask |= 1u << f.s_etilt;
    } else if (gradient) {
      if (int rz = zero_doubles(c, f.grad, b3)) return rz;
    }
    if (mods & f.mod_bt) {
      rc = bt_pass_f(c, f, gradient ? 2 : 0, false, 0.0, tilts);
      if (rc) return rc;
      mask |= 1u << f.s_ebt;
    }
    if (mods & f.mod_dt) {
      rc = disk_target_pass(c, f, gradient ? 1 : 0, false, 0.0, tilts, false, gradient);
      if (rc) return rc;
      mask |= 1u << f.s_edt;
    }
    if (mods & f.mod_smooth) {
      rc = ts_pass_f(c, f, gradient ? 1 : 0, false, 0.0, tilts);
      if (rc) return rc;
      mask |= 1u << f.s_ets;
    }
  }
  return mask ? reduce_slots(c, mask) : MS_OK;
}
double tilt_energy_from_mailbox(const ms_ctx* c) {
  double e = 0.0;
  if (c->params.modules & MS_MOD_TILT) e += c->h_scal[MS_S_ETILT];
  if (c->params.modules & MS_MOD_BENDING_TILT) e += c->h_scal[MS_S_EBT];
  if (c->params.modules & MS_MOD_TILT_SMOOTH) e += c->h_scal[MS_S_ETS];
  for (int l = 1; l <= 2; ++l) {
    const TiltField& f = c->tf[l];
    if (c->params.modules & f.mod_tilt) e += c->h_scal[f.s_etilt];
    if (c->params.modules & f.mod_smooth) e += c->h_scal[f.s_ets];
    if (c->params.modules & f.mod_bt) e += c->h_scal[f.s_ebt];
    if (c->params.modules & f.mod_dt) e += c->h_scal[f.s_edt];
  }
  return e;
}
int ensure_bt_record(ms_ctx* c) {
  if (!(c->params.modules & (MS_MOD_BENDING_TILT | MS_LEAFLET_BT)) || c->bt_valid) return MS_OK;
  return phase_energy(c, c->params.modules, false, 0.0, false, false, false, /*reduce_now=*/false);
}

// The relaxation loop of relax_fields as a device program: the command lists the loop launches over and over are
// CAPTURED once (both parities of the tilts <-> trial swap), the control flow runs in the interpreter's workgroup
// (ms_exec.inc: exec_relax).  *used = false: not possible here (the caller runs the host-driven loop).
int relax_program(ms_ctx* c, const ms_tilt_relax_params* rp, TiltField** fl, int nf, uint32_t norm_mask, int* iters,
                  int* evals, bool* used) {
  *used = false;
  if (nf < 1 || nf > 2) return MS_OK;
  const Tiling& t = c->til;
  if (!c->d_relax_cells) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_relax_cells), sizeof(double) * 2));
    HIPCHK(c, hipMemset(c->d_relax_cells, 0, sizeof(double) * 2));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_relax_box), sizeof(unsigned long long) * 8, hipHostMallocMapped));
    memset(c->h_relax_box, 0, sizeof(unsigned long long) * 8);
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_relax_box), c->h_relax_box, 0));
  }
  int rc = exec_flush(c);  // (the set-up passes recorded so far run first, as a pack of their own)
  if (rc) return rc;
  ExecRecorder& R = c->exec;
  const bool along_dir = rp->solver != 0;
  ExecRelaxHead head;
  memset(&head, 0, sizeof(head));
  // -- capture -------------------------------------------------------------------------------------------------
  const unsigned long long s_ticket = c->ticket;
  unsigned long long s_expected[MS_MB_WORDS];
  memcpy(s_expected, c->expected, sizeof(s_expected));
  const long s_cmds = R.cmds;
  R.capture = true;
  auto count = [&]() { return (int)reinterpret_cast<const ExecPackHead*>(R.buf.data())->n_cmds; };
  auto fail_capture = [&](int code) {
    R.capture = false;
    R.buf.clear();
    R.cmds = s_cmds;
    c->ticket = s_ticket;
    memcpy(c->expected, s_expected, sizeof(s_expected));
    return code;
  };
  hipError_t he = R.push(CK_RELAX, 0, 0, 0, 1, 0, 0, &head, sizeof(head));
  if (he != hipSuccess) return fail_capture(MS_OK);
  const size_t head_at = sizeof(ExecPackHead);
  auto tvec = [&](TiltField& f, int mode, const double* src, double* out, int flag) -> hipError_t {
    return launch_tvec(mode, c->tile0, c->tile1, t.nv, t.own, c->d_vflags, f.grad, f.minv, f.dir, f.tilts, src, c->d_tn,
                       out, 0.0, flag, c->d_partials, t.n_tiles, c->stream, f.fixed_bit, f.s_gn2, f.s_rz);
  };
  bool ok = true;
  for (int par = 0; par < 2 && ok; ++par) {
    head.off_grad[par] = (int32_t)R.buf.size();
    int n0 = count();
    ok = ok && tilt_eval(c, false, true) == MS_OK;
    for (int k = 0; k < nf && ok; ++k) ok = tvec(*fl[k], 0, nullptr, nullptr, 0) == hipSuccess;
    ok = ok && reduce_slots(c, norm_mask) == MS_OK;
    head.n_grad[par] = count() - n0;
    head.off_trial[par] = (int32_t)R.buf.size();
    n0 = count();
    for (int k = 0; k < nf && ok; ++k) {
      TiltField& f = *fl[k];
      ok = tvec(f, 2, along_dir ? f.dir : f.grad, f.trial, 1) == hipSuccess;
    }
    ok = ok && tilt_eval(c, true, false) == MS_OK;
    head.n_trial[par] = count() - n0;
    for (int k = 0; k < nf; ++k) std::swap(fl[k]->tilts, fl[k]->trial);  // (twice in all: back where they were)
  }
  head.off_dir0 = (int32_t)R.buf.size();
  {
    const int n0 = count();
    for (int k = 0; k < nf && ok; ++k) ok = tvec(*fl[k], 1, nullptr, nullptr, 1) == hipSuccess;
    head.n_dir0 = count() - n0;
  }
  head.off_dir1 = (int32_t)R.buf.size();
  {
    const int n0 = count();
    for (int k = 0; k < nf && ok; ++k) ok = tvec(*fl[k], 1, nullptr, nullptr, 0) == hipSuccess;
    head.n_dir1 = count() - n0;
  }
  if (!ok) return fail_capture(MS_OK);  // (e.g. the program does not fit the largest pack: host-driven loop)
  std::vector<unsigned char> prog;
  prog.swap(R.buf);
  const size_t prog_lds = R.lds;
  R.capture = false;
  R.cmds = s_cmds;
  c->ticket = s_ticket;
  memcpy(c->expected, s_expected, sizeof(s_expected));
  // -- patch: the trial / direction passes take their coefficient from the program's cells, the folds post nothing
  {
    size_t at = head_at + sizeof(ExecCmdHead) + sizeof(ExecRelaxHead);
    while (at + sizeof(ExecCmdHead) <= prog.size()) {
      ExecCmdHead h;
      memcpy(&h, prog.data() + at, sizeof(h));
      unsigned char* args = prog.data() + at + sizeof(ExecCmdHead);
      if (h.kind == CK_TVEC) {
        TvecArgs a;
        memcpy(&a, args, sizeof(a));
        if (a.mode == 2) a.coef_dev = c->d_relax_cells;
        if (a.mode == 1 && (int32_t)at >= head.off_dir1) a.coef_dev = c->d_relax_cells + 1;
        memcpy(args, &a, sizeof(a));
      } else if (h.kind == CK_REDUCE) {
        FoldArgs a;
        memcpy(&a, args, sizeof(a));
        for (int j = 0; j < MS_MAX_TRIALS; ++j) a.set[j].host_box = nullptr;
        memcpy(args, &a, sizeof(a));
      }
      at += h.bytes;
    }
  }
  // -- the head ---------------------------------------------------------------------------------------------------
  head.solver = rp->solver;
  head.max_iters = rp->max_iters;
  head.nf = nf;
  head.step_size = rp->step_size;
  head.tol = rp->tol;
  {
    const uint32_t mods = c->params.modules;  // (the order of tilt_energy_from_mailbox)
    int n = 0;
    if (mods & MS_MOD_TILT) head.e_slot[n++] = MS_S_ETILT;
    if (mods & MS_MOD_BENDING_TILT) head.e_slot[n++] = MS_S_EBT;
    if (mods & MS_MOD_TILT_SMOOTH) head.e_slot[n++] = MS_S_ETS;
    for (int l = 1; l <= 2; ++l) {
      const TiltField& f = c->tf[l];
      if (mods & f.mod_tilt) head.e_slot[n++] = f.s_etilt;
      if (mods & f.mod_smooth) head.e_slot[n++] = f.s_ets;
      if (mods & f.mod_bt) head.e_slot[n++] = f.s_ebt;
      if (mods & f.mod_dt) head.e_slot[n++] = f.s_edt;
    }
    head.n_e = n;
  }
  for (int k = 0; k < nf; ++k) {
    head.s_gn2[k] = fl[k]->s_gn2;
    head.s_rz[k] = fl[k]->s_rz;
  }
  head.scal = c->d_scal;
  head.cells = c->d_relax_cells;
  head.host_box = c->d_h_relax_box;
  head.ticket = ++c->relax_ticket;
  head.total_bytes = (int32_t)(prog.size() - head_at);
  {
    ExecCmdHead h;
    memcpy(&h, prog.data() + head_at, sizeof(h));
    h.pad = head.total_bytes;
    memcpy(prog.data() + head_at, &h, sizeof(h));
    memcpy(prog.data() + head_at + sizeof(h), &head, sizeof(head));
    reinterpret_cast<ExecPackHead*>(prog.data())->n_cmds = 1;  // (the lists belong to the CK_RELAX record)
  }
  he = R.launch_pack(prog, prog_lds);
  if (he != hipSuccess) return fail_hip(c, he, "k_exec (relaxation program)");
  ++c->relax_programs;
  // -- result ------------------------------------------------------------------------------------------------------
  auto entry = [&](int k, unsigned long long* v) {
    const unsigned long long tag = __atomic_load_n(&c->h_relax_box[2 * k + 1], __ATOMIC_ACQUIRE);
    *v = __atomic_load_n(&c->h_relax_box[2 * k], __ATOMIC_ACQUIRE);
    return (*v ^ tag) == head.ticket;
  };
  unsigned long long v[4] = {0, 0, 0, 0};
  bool done = false;
  for (long spin = 0; !done && spin < 400000000L; ++spin) {
    done = entry(3, &v[3]) && entry(0, &v[0]) && entry(1, &v[1]) && entry(2, &v[2]);
    if (!done) __builtin_ia32_pause();
  }
  if (!done) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    done = entry(3, &v[3]) && entry(0, &v[0]) && entry(1, &v[1]) && entry(2, &v[2]);
    if (!done) return fail(c, MS_ERR_STATE, "relaxation program: the result mailbox was not posted");
  }
  *iters = (int)v[0];
  *evals = (int)v[1];
  if (v[2] & 1ull)
    for (int k = 0; k < nf; ++k) std::swap(fl[k]->tilts, fl[k]->trial);
  *used = true;
  return MS_OK;
}

// TiltRelaxationManager.relax_tilts (tilt_relaxation.py:237-424) for one field and
// relax_leaflet_tilts (:426-1478, default options) for the (in, out) pair: the same driver over
// the concatenated free rows of `fl[0..nf)` -- one energy, |grad|^2 and <r, M^-1 r> summed over
// the fields, one step length for all of them.
int relax_fields(ms_ctx* c, const ms_tilt_relax_params* rp, TiltField** fl, int nf, bool jacobi_smooth_by_param,
                 int* iters_out, int* evals_out) {
  const uint32_t mods = c->params.modules;
  const Tiling& t = c->til;
  bool any_free = false;
  for (int k = 0; k < nf; ++k) any_free = any_free || fl[k]->any_free;
  if (!any_free) return MS_OK;
  const size_t b3 = sizeof(double) * 3 * (size_t)t.nvp;
  if (!c->d_tn) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_tn), b3));
    HIPCHK(c, hipMemset(c->d_tn, 0, b3));
  }
  for (int k = 0; k < nf; ++k) {
    TiltField& f = *fl[k];
    if (f.dir) continue;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.dir), b3));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.minv), sizeof(double) * (size_t)t.nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.va), sizeof(double) * (size_t)t.nvp));
    HIPCHK(c, hipMemset(f.dir, 0, b3));
    HIPCHK(c, hipMemset(f.minv, 0, sizeof(double) * (size_t)t.nvp));
    HIPCHK(c, hipMemset(f.va, 0, sizeof(double) * (size_t)t.nvp));
  }
  int iters = 0, evals = 0;
  int rc = ensure_bt_record(c);
  if (rc) return rc;
  auto tvec = [&](TiltField& f, int mode, const double* src, double* out, double coef, int flag) -> int {
    ProfScope ps(c, 6);
    HIPCHK(c, launch_tvec(mode, c->tile0, c->tile1, t.nv, t.own, c->d_vflags, f.grad, f.minv, f.dir, f.tilts, src,
                          c->d_tn, out, coef, flag, c->d_partials, t.n_tiles, c->stream, f.fixed_bit, f.s_gn2,
                          f.s_rz));
    return MS_OK;
  };
  uint32_t norm_mask = 0;
  for (int k = 0; k < nf; ++k) {
    TiltField& f = *fl[k];
    norm_mask |= (1u << f.s_gn2) | (1u << f.s_rz);
    {  // frozen geometry: unit vertex normals and the tilt-rigidity part of the Jacobi diagonal
      TiltArgs a;
  a.fields = nullptr;
  a.fields_rows = 0;
      a.m = device_mesh(c);
      a.tile0 = c->tile0;
      a.tile1 = c->tile1;
      a.x = c->buf[MS_BUF_X];
      a.d = nullptr;
      a.alpha = 0.0;
      a.tilts = f.tilts;
      a.tilts_out = c->d_tn;
      a.k_tilt = (rp->solver == 1 && rp->jacobi) ? f.k_tilt : 0.0;
      a.g = nullptr;
      a.tilt_grad = f.grad;
      a.minv = f.minv;
      a.partials = c->d_partials;
      a.e_slot = f.s_etilt;
      a.consistent = 0;
      a.tg_accumulate = 0;
  a.cons_tilt_grad = 0;
      a.va_out = f.va;
      HIPCHK(c, launch_tilt(a, 3, c->cap, t.max_ent, c->stream));
    }
    // + 1/2 k_s sum (c_a + c_b): the parameter alone decides, loaded module or not
    // (preconditioners.py:42-57 single field, :111-139 leaflets)
    const double ks = jacobi_smooth_by_param ? f.k_smooth_precond : f.k_smooth;
    if (rp->solver == 1 && rp->jacobi && ks != 0.0) {
      rc = ts_pass_f(c, f, 2, false, 0.0, f.tilts, f.minv, ks);
      if (rc) return rc;
    }
    rc = tvec(f, 3, nullptr, f.minv, 0.0, 0);  // diagonal -> clamped inverse
    if (rc) return rc;
    // tilts <- P(tilts) on every row (:303-305 / :640-663), fixed rows keep that value from now on
    rc = tvec(f, 2, f.tilts, f.trial, 0.0, 0);
    if (rc) return rc;
    std::swap(f.tilts, f.trial);
  }
  struct VaScope {  // the cached vertex areas / disk-target profiles are valid only while this relaxation runs (x frozen)
    ms_ctx* c;
    ~VaScope() {
      c->relax_va_valid = false;
      for (int l = 0; l < 3; ++l) c->tf[l].dt_target_valid = false;
    }
  } va_scope{c};
  c->relax_va_valid = jacobi_smooth_by_param;  // leaflet driver only (the single field has no such form)
  for (int k = 0; k < nf && c->relax_va_valid; ++k) {
    // prime the frozen-surface caches of the disk target (radius, theta(r) r_hat) before the loop: every evaluation of
    // the relaxation -- and the device program's captured lists -- then only take differences
    TiltField& f = *fl[k];
    f.dt_target_valid = false;
    if (!(mods & f.mod_dt)) continue;
    rc = disk_target_pass(c, f, 0, false, 0.0, f.tilts, false, false);
    if (rc) return rc;
  }
  auto grad_at = [&](double* E, double* gnorm, double* rz) -> int {
    int r = tilt_eval(c, false, true);
    if (r) return r;
    for (int k = 0; k < nf; ++k) {
      r = tvec(*fl[k], 0, nullptr, nullptr, 0.0, 0);
      if (r) return r;
    }
    r = reduce_slots(c, norm_mask);
    if (r) return r;
    r = fetch(c);
    if (r) return r;
    ++evals;
    *E = tilt_energy_from_mailbox(c);
    double g2 = 0.0, z = 0.0;
    for (int k = 0; k < nf; ++k) {
      g2 += c->h_scal[fl[k]->s_gn2];
      z += c->h_scal[fl[k]->s_rz];
    }
    *gnorm = std::sqrt(g2);
    *rz = z;
    return MS_OK;
  };
  // backtracking on E(P(t + step*src)) <= E0 (:330-347 / :380-398 / :918-973); accepts by pointer swap
  auto search = [&](bool along_dir, double sign, double E0, double* E_acc, bool* accepted) -> int {
    double step = rp->step_size;
    *accepted = false;
    for (int bt = 0; bt < 12; ++bt) {
      for (int k = 0; k < nf; ++k) {
        TiltField& f = *fl[k];
        int r = tvec(f, 2, along_dir ? f.dir : f.grad, f.trial, sign * step, 1);
        if (r) return r;
      }
      int r = tilt_eval(c, true, false);
      if (r) return r;
      r = fetch(c);
      if (r) return r;
      ++evals;
      const double E1 = tilt_energy_from_mailbox(c);
      if (E1 <= E0) {
        for (int k = 0; k < nf; ++k) std::swap(fl[k]->tilts, fl[k]->trial);
        *E_acc = E1;
        *accepted = true;
        return MS_OK;
      }
      step *= 0.5;
      if (step < 1e-16) break;
    }
    return MS_OK;
  };
  if (c->exec_on && c->exec_relax) {
    // one-tile context: the whole solve below as ONE launch (ms_internal.h: ExecRelaxHead)
    bool used = false;
    rc = relax_program(c, rp, fl, nf, norm_mask, &iters, &evals, &used);
    if (rc) return rc;
    if (used) {
      if (iters_out) *iters_out = iters;
      if (evals_out) *evals_out = evals;
      c->carry_valid = c->grad_valid = false;
      c->factors_valid = c->factors_valid && !(mods & (MS_MOD_BENDING_TILT | MS_LEAFLET_BT));
      return MS_OK;
    }
  }
  const double tol = rp->tol > 0.0 ? rp->tol : 0.0;
  double E0 = 0.0, gnorm = 0.0, rz_old = 0.0;
  if (rp->solver == 0) {  // gradient descent (:312-351 / :892-1058)
    for (int it = 0; it < rp->max_iters; ++it) {
      rc = grad_at(&E0, &gnorm, &rz_old);
      if (rc) return rc;
      if (gnorm == 0.0 || (tol > 0.0 && gnorm < tol)) break;
      bool acc = false;
      double E1 = E0;
      rc = search(false, -1.0, E0, &E1, &acc);
      if (rc) return rc;
      ++iters;
      if (!acc) break;
    }
  } else {  // (preconditioned) Fletcher-Reeves CG (:352-421 / :1059-1398)
    rc = grad_at(&E0, &gnorm, &rz_old);
    if (rc) return rc;
    if (!(gnorm == 0.0 || (tol > 0.0 && gnorm < tol))) {
      for (int k = 0; k < nf; ++k) {
        rc = tvec(*fl[k], 1, nullptr, nullptr, 0.0, 1);
        if (rc) return rc;
      }
      for (int it = 0; it < rp->max_iters; ++it) {
        if (gnorm == 0.0 || (tol > 0.0 && gnorm < tol)) break;
        bool acc = false;
        double E1 = E0;
        rc = search(true, 1.0, E0, &E1, &acc);
        if (rc) return rc;
        ++iters;
        if (!acc) break;
        double rz_new = 0.0;
        rc = grad_at(&E0, &gnorm, &rz_new);
        if (rc) return rc;
        if (gnorm == 0.0 || (tol > 0.0 && gnorm < tol)) break;
        if (rz_old == 0.0) break;
        const double beta = rz_new / rz_old;
        for (int k = 0; k < nf; ++k) {
          rc = tvec(*fl[k], 1, nullptr, nullptr, beta, 0);
          if (rc) return rc;
        }
        rz_old = rz_new;
      }
    }
  }
  if (iters_out) *iters_out = iters;
  if (evals_out) *evals_out = evals;
  // the tilt-dependent energies in the mailbox belong to whatever was evaluated last
  c->carry_valid = c->grad_valid = false;
  c->factors_valid = c->factors_valid && !(mods & (MS_MOD_BENDING_TILT | MS_LEAFLET_BT));
  return MS_OK;
}
}  // namespace

int ms_tilt_energy_and_gradient(ms_ctx* c, double* energy, double* tilt_grad) {
  if (!c || !energy) return fail(c, MS_ERR_INVALID, "ms_tilt_energy_and_gradient: NULL argument");
  if (!(c->params.modules & MS_TILT_MODS) || !c->tf[0].tilts)
    return fail(c, MS_ERR_STATE, "ms_tilt_energy_and_gradient: no tilt-reading module / no tilts set");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "tilt passes are single-shard only");
  int rc = ensure_bt_record(c);
  if (rc) return rc;
  rc = tilt_eval(c, false, true);
  if (rc) return rc;
  rc = fetch(c);
  if (rc) return rc;
  *energy = tilt_energy_from_mailbox(c);
  if (tilt_grad) return patch_to_ext(c, c->tf[0].grad, tilt_grad, 3);
  return MS_OK;
}

int ms_relax_tilts(ms_ctx* c, const ms_tilt_relax_params* rp, int* iters_out, int* evals_out) {
  if (!c || !rp) return fail(c, MS_ERR_INVALID, "ms_relax_tilts: NULL argument");
  if (iters_out) *iters_out = 0;
  if (evals_out) *evals_out = 0;
  const uint32_t mods = c->params.modules;
  if (!(mods & MS_TILT_MODS) || !c->tf[0].tilts)
    return fail(c, MS_ERR_STATE, "ms_relax_tilts: no tilt-reading module / no tilts set");
  if (mods & MS_LEAFLET_MODS)
    return fail(c, MS_ERR_STATE, "ms_relax_tilts: leaflet modules are active, use ms_relax_leaflet_tilts");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "tilt passes are single-shard only");
  if (rp->step_size <= 0.0 || rp->max_iters <= 0) return MS_OK;
  TiltField* fl[1] = {&c->tf[0]};
  return relax_fields(c, rp, fl, 1, /*jacobi_smooth_by_param=*/false, iters_out, evals_out);
}

// ---- two-leaflet tilt fields ---------------------------------------------------------------
int ms_set_leaflet_tilts(ms_ctx* c, int leaflet, const double* tilts, const uint8_t* tilt_fixed,
                         const ms_leaflet_params* lp) {
  if (!c || !tilts || !lp) return fail(c, MS_ERR_INVALID, "ms_set_leaflet_tilts: NULL argument");
  if (leaflet != MS_LEAFLET_IN && leaflet != MS_LEAFLET_OUT)
    return fail(c, MS_ERR_INVALID, "ms_set_leaflet_tilts: leaflet must be MS_LEAFLET_IN or MS_LEAFLET_OUT");
  TiltField& f = c->tf[1 + leaflet];
  const Tiling& t = c->til;
  const size_t bytes = sizeof(double) * 3 * (size_t)t.nvp;
  if (!f.tilts) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.tilts), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.grad), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.trial), bytes));
    HIPCHK(c, hipMemset(f.tilts, 0, bytes));
    HIPCHK(c, hipMemset(f.grad, 0, bytes));
    HIPCHK(c, hipMemset(f.trial, 0, bytes));
  }
  f.k_tilt = lp->tilt_modulus;
  f.consistent = lp->tilt_mass_consistent ? 1 : 0;
  f.k_smooth = lp->smoothness;
  f.k_smooth_precond = lp->precond_smoothness;
  bool flags_changed = false;
  for (int i = 0; i < t.nv; ++i) {
    uint8_t fl = c->h_vflags[(size_t)i] & (uint8_t)~f.fixed_bit;
    if (tilt_fixed && tilt_fixed[t.perm[i]]) fl |= f.fixed_bit;
    flags_changed = flags_changed || fl != c->h_vflags[(size_t)i];
    c->h_vflags[(size_t)i] = fl;
  }
  if (flags_changed || !tilt_fixed) {
    f.any_free = t.nv == 0 || !tilt_fixed;
    for (int i = 0; i < t.nv && !f.any_free; ++i) f.any_free = !(c->h_vflags[(size_t)i] & f.fixed_bit);
  }
  if (flags_changed) {
    HIPCHK(c, hipStreamSynchronize(S(c)));
    HIPCHK(c, hipMemcpy(c->d_vflags, c->h_vflags.data(), c->h_vflags.size(), hipMemcpyHostToDevice));
  }
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return ext_to_patch(c, tilts, f.tilts, 3);
}

int ms_set_leaflet_bending(ms_ctx* c, int leaflet, const double* kappa, const double* c0) {
  if (!c || !kappa || !c0) return fail(c, MS_ERR_INVALID, "ms_set_leaflet_bending: NULL argument");
  if (leaflet != MS_LEAFLET_IN && leaflet != MS_LEAFLET_OUT)
    return fail(c, MS_ERR_INVALID, "ms_set_leaflet_bending: leaflet must be MS_LEAFLET_IN or MS_LEAFLET_OUT");
  TiltField& f = c->tf[1 + leaflet];
  const Tiling& t = c->til;
  HIPCHK(c, hipStreamSynchronize(S(c)));
  if (!f.kappa) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.kappa), sizeof(double) * (size_t)t.nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.c0), sizeof(double) * (size_t)t.nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.bt_vert), sizeof(double) * 4 * (size_t)t.nvp));
    HIPCHK(c, hipMemset(f.bt_vert, 0, sizeof(double) * 4 * (size_t)t.nvp));
  }
  std::vector<double> k((size_t)t.nvp, 0.0), z((size_t)t.nvp, 0.0);
  for (int i = 0; i < t.nv; ++i) {
    k[i] = kappa[t.perm[i]];
    z[i] = c0[t.perm[i]];
  }
  HIPCHK(c, hipMemcpy(f.kappa, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(f.c0, z.data(), z.size() * sizeof(double), hipMemcpyHostToDevice));
  c->factors_valid = false;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  return MS_OK;
}

int ms_set_leaflet_disk_target(ms_ctx* c, int leaflet, const uint8_t* disk_rows, const ms_disk_target_params* p) {
  if (!c || !p) return fail(c, MS_ERR_INVALID, "ms_set_leaflet_disk_target: NULL argument");
  if (leaflet != MS_LEAFLET_IN && leaflet != MS_LEAFLET_OUT)
    return fail(c, MS_ERR_INVALID, "ms_set_leaflet_disk_target: leaflet must be MS_LEAFLET_IN or MS_LEAFLET_OUT");
  TiltField& f = c->tf[1 + leaflet];
  const Tiling& t = c->til;
  const double nn = std::sqrt(p->normal[0] * p->normal[0] + p->normal[1] * p->normal[1] + p->normal[2] * p->normal[2]);
  if (!(nn >= 1e-15)) return fail(c, MS_ERR_INVALID, "ms_set_leaflet_disk_target: a plane normal is required");
  HIPCHK(c, hipStreamSynchronize(S(c)));
  if (!f.disk) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.disk), (size_t)t.nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&f.diff), sizeof(double) * 3 * (size_t)t.nvp));
    HIPCHK(c, hipMemset(f.diff, 0, sizeof(double) * 3 * (size_t)t.nvp));
  }
  std::vector<uint8_t> m((size_t)t.nvp, 0);
  for (int i = 0; i < t.nv && disk_rows; ++i) m[(size_t)i] = disk_rows[t.perm[i]] ? 1 : 0;
  HIPCHK(c, hipMemcpy(f.disk, m.data(), m.size(), hipMemcpyHostToDevice));
  f.dt = *p;
  for (int k = 0; k < 3; ++k) f.dt.normal[k] = p->normal[k] / nn;  // tilt_disk_target_in.py:73-77
  c->carry_valid = c->grad_valid = c->maxg2_valid = false;
  return MS_OK;
}

int ms_get_leaflet_tilts(ms_ctx* c, int leaflet, double* tilts) {
  if (!c || !tilts || (leaflet != MS_LEAFLET_IN && leaflet != MS_LEAFLET_OUT) || !c->tf[1 + leaflet].tilts)
    return fail(c, MS_ERR_INVALID, "ms_get_leaflet_tilts: bad leaflet / no tilts set");
  return patch_to_ext(c, c->tf[1 + leaflet].tilts, tilts, 3);
}

namespace {
int leaflet_ready(ms_ctx* c, const char* who, TiltField** fl, int* nf) {
  const uint32_t mods = c->params.modules;
  if (!(mods & MS_LEAFLET_MODS)) return fail(c, MS_ERR_STATE, std::string(who) + ": no leaflet module is active");
  if (mods & MS_TILT_MODS)
    return fail(c, MS_ERR_STATE, std::string(who) + ": single-field and leaflet tilt modules together are outside the device path");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "tilt passes are single-shard only");
  *nf = 0;
  for (int l = 1; l <= 2; ++l) {
    TiltField& f = c->tf[l];
    if (!(mods & (f.mod_tilt | f.mod_smooth | f.mod_bt | f.mod_dt))) continue;
    if (!f.tilts) return fail(c, MS_ERR_STATE, std::string(who) + ": ms_set_leaflet_tilts was not called for an active leaflet");
    fl[(*nf)++] = &f;
  }
  return MS_OK;
}
}  // namespace

int ms_leaflet_tilt_energy_and_gradient(ms_ctx* c, double* energy, double* grad_in, double* grad_out) {
  return ms_leaflet_tilt_energy_and_gradient_ex(c, 0, energy, grad_in, grad_out);
}

int ms_leaflet_tilt_energy_and_gradient_ex(ms_ctx* c, int module_form, double* energy, double* grad_in,
                                           double* grad_out) {
  if (!c || !energy) return fail(c, MS_ERR_INVALID, "ms_leaflet_tilt_energy_and_gradient: NULL argument");
  TiltField* fl[2];
  int nf = 0;
  int rc = leaflet_ready(c, "ms_leaflet_tilt_energy_and_gradient", fl, &nf);
  if (rc) return rc;
  c->tilt_module_form = module_form != 0;
  rc = tilt_eval(c, false, true);
  c->tilt_module_form = false;
  if (rc) return rc;
  rc = fetch(c);
  if (rc) return rc;
  *energy = tilt_energy_from_mailbox(c);
  double* outs[2] = {grad_in, grad_out};
  for (int l = 0; l < 2; ++l) {
    if (!outs[l]) continue;
    TiltField& f = c->tf[1 + l];
    if (f.tilts && (c->params.modules & (f.mod_tilt | f.mod_smooth | f.mod_bt | f.mod_dt))) {
      rc = patch_to_ext(c, f.grad, outs[l], 3);
      if (rc) return rc;
    } else {
      memset(outs[l], 0, sizeof(double) * 3 * (size_t)c->til.nv);
    }
  }
  return MS_OK;
}

int ms_relax_leaflet_tilts(ms_ctx* c, const ms_tilt_relax_params* rp, int* iters_out, int* evals_out) {
  if (!c || !rp) return fail(c, MS_ERR_INVALID, "ms_relax_leaflet_tilts: NULL argument");
  if (iters_out) *iters_out = 0;
  if (evals_out) *evals_out = 0;
  TiltField* fl[2];
  int nf = 0;
  int rc = leaflet_ready(c, "ms_relax_leaflet_tilts", fl, &nf);
  if (rc) return rc;
  if (rp->step_size <= 0.0 || rp->max_iters <= 0) return MS_OK;
  return relax_fields(c, rp, fl, nf, /*jacobi_smooth_by_param=*/true, iters_out, evals_out);
}

int ms_set_positions(ms_ctx* c, const double* positions) {
  if (!c || !positions) return fail(c, MS_ERR_INVALID, "ms_set_positions: NULL argument");
  c->factors_valid = false;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  return ext_to_patch(c, positions, c->buf[MS_BUF_X], 3);
}

int ms_get_positions(ms_ctx* c, double* positions) {
  if (!c || !positions) return fail(c, MS_ERR_INVALID, "ms_get_positions: NULL argument");
  return patch_to_ext(c, c->buf[MS_BUF_X], positions, 3);
}

int ms_get_gradient(ms_ctx* c, double* grad) {
  if (!c || !grad) return fail(c, MS_ERR_INVALID, "ms_get_gradient: NULL argument");
  return patch_to_ext(c, c->last_g, grad, 3);
}

int ms_get_vertex_buffer(ms_ctx* c, int buffer, double* out) {
  if (!c || !out || buffer < 0 || buffer > MS_BUF_FA)
    return fail(c, MS_ERR_INVALID, "ms_get_vertex_buffer: bad argument");
  if ((buffer == MS_BUF_D && c->dir_implicit) || (buffer == MS_BUF_PD && c->pd_neg_pg)) {
    // the direction asked for exists only as -G / -PG: write it out
    const int src = buffer == MS_BUF_D ? MS_BUF_G : MS_BUF_PG;
    HIPCHK(c, launch_direction(c->tile0, c->tile1, c->til.nv, c->til.own, c->d_vflags, c->buf[src], c->buf[MS_BUF_GC],
                               c->buf[buffer], c->buf[MS_BUF_PG], c->buf[MS_BUF_PD], c->d_scal, 0, 0, c->d_partials,
                               c->til.n_tiles, 0, c->stream));
    if (buffer == MS_BUF_D) c->dir_implicit = false; else c->pd_neg_pg = false;
  }
  return patch_to_ext(c, c->buf[buffer], out, buffer == MS_BUF_FA ? 2 : 3);
}

int ms_energy_and_gradient(ms_ctx* c, double energies[4], double* grad) {
  if (!c || !energies) return fail(c, MS_ERR_INVALID, "ms_energy_and_gradient: NULL argument");
  if (c->shard_count != 1)
    return fail(c, MS_ERR_STATE, "ms_energy_and_gradient: sharded contexts use the phase API");
  int rc = queue_energy_and_gradient(c, MS_STEPPER_GD, false);
  if (rc) return rc;
  rc = fetch(c);
  if (rc) return rc;
  energies_from_mailbox(c, energies);
  if (grad) return patch_to_ext(c, c->buf[MS_BUF_G], grad, 3);
  return MS_OK;
}

int ms_energy_and_raw_gradient(ms_ctx* c, double energies[4], double* grad) {
  if (!c || !energies) return fail(c, MS_ERR_INVALID, "ms_energy_and_raw_gradient: NULL argument");
  if (c->shard_count != 1)
    return fail(c, MS_ERR_STATE, "ms_energy_and_raw_gradient: sharded contexts use the phase API");
  const uint32_t mods = c->params.modules;
  const bool penalty = (mods & MS_MOD_VOLUME_PENALTY) != 0;  // K_C reads the reduced volume
  int rc = phase_energy(c, mods, false, 0.0, false, false, true, /*reduce_now=*/penalty);
  if (rc) return rc;
  c->grad_valid = false;  // G receives the raw gradient: no fixed-row zeroing, no KKT projection
  rc = phase_gradient(c, mods, c->buf[MS_BUF_G], false, 0, /*reduce_now=*/false);
  if (rc) return rc;
  rc = reduce_slots(c, (penalty ? 0u : energy_mask(mods)) | MASK_GRAD);
  if (rc) return rc;
  rc = fetch(c);
  if (rc) return rc;
  energies_from_mailbox(c, energies);
  if (grad) return patch_to_ext(c, c->buf[MS_BUF_G], grad, 3);
  return MS_OK;
}

int ms_energy(ms_ctx* c, double energies[4]) {
  if (!c || !energies) return fail(c, MS_ERR_INVALID, "ms_energy: NULL argument");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "ms_energy: sharded contexts use the phase API");
  if (c->carry_valid) {
    // the last accepted trial (or the pass before a failed search) evaluated exactly this x and its energies are
    // still in the mailbox: no pass, and the carried state survives for the next step
    energies_from_mailbox(c, energies);
    return MS_OK;
  }
  int rc = phase_energy(c, c->params.modules, false, 0.0, false, false, false);
  if (rc) return rc;
  rc = fetch(c);
  if (rc) return rc;
  energies_from_mailbox(c, energies);
  return MS_OK;
}

int ms_reset_stepper(ms_ctx* c) {
  if (!c) return MS_ERR_INVALID;
  c->cg_have_history = false;
  c->cg_iter_count = 0;
  c->pd_neg_pg = false;
  return MS_OK;
}

namespace {
int spec_prepare(ms_ctx* c) {
  if (c->d_dec) return MS_OK;
  auto make_box = [&](ms_ctx::Mailbox& m) -> int {
    m.h_scal = static_cast<double*>(calloc(MS_NSCAL, sizeof(double)));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&m.h_seq), sizeof(unsigned long long) * 2 * MS_MB_WORDS,
                            hipHostMallocMapped));
    memset(m.h_seq, 0, sizeof(unsigned long long) * 2 * MS_MB_WORDS);
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&m.d_h_seq), m.h_seq, 0));
    return MS_OK;
  };
  if (c->pair_enable) {
    const size_t nvp = (size_t)std::max<int64_t>(1, c->til.nvp);
    const size_t pb = sizeof(double) * MS_NPART * (size_t)std::max(1, c->til.n_tiles);
    // early trial 0 may write outputs of its own (the sharded pair, MS_PAIR_LEAN=0), early trial 1 only in ms_step
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->xt2), sizeof(double) * 3 * nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fK2), sizeof(double) * 3 * nvp));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fA2), sizeof(double) * 2 * nvp));
    if (c->shard_count == 1 && !c->pair_lean_enable) {
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->xt3), sizeof(double) * 3 * nvp));
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fK3), sizeof(double) * 3 * nvp));
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fA3), sizeof(double) * 2 * nvp));
    }
    const int n_side = c->shard_count == 1 ? ms_ctx::N_SIDE : 1;  // (the sharded driver pairs, no more)
    for (int k = 0; k < n_side; ++k) {
      ms_ctx::SideSet& sd = c->side[k];
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&sd.partials), pb));
      HIPCHK(c, hipMemset(sd.partials, 0, pb));
      HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&sd.scal), sizeof(double) * MS_NSCAL));
      HIPCHK(c, hipMemset(sd.scal, 0, sizeof(double) * MS_NSCAL));
      for (auto& m : sd.mb) {
        int rc = make_box(m);
        if (rc) return rc;
      }
    }
  }
  for (int p = 0; p < 2; ++p) {
    for (int k = 0; k < ms_ctx::SPEC_STAGES; ++k) {
      int rc = make_box(c->spec[p][k]);
      if (rc) return rc;
    }
    int rc = make_box(c->grad_mb[p]);
    if (rc) return rc;
    rc = make_box(c->first_mb[p]);
    if (rc) return rc;
  }
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_dec), sizeof(uint32_t) * MS_DEC_STRIDE * ms_ctx::N_DEC));
  HIPCHK(c, hipMemset(c->d_dec, 0, sizeof(uint32_t) * MS_DEC_STRIDE * ms_ctx::N_DEC));
  return MS_OK;
}
// the host knows a gated fold stayed out (an earlier stage was accepted, or nothing was): nobody waits for its ticket
inline void forget(ms_ctx::Mailbox& m) {
  for (int sl = 0; sl < MS_MB_WORDS; ++sl) m.expected[sl] = 0;
}
// make stage mailbox `m` the context's mailbox (and back: the swap is its own inverse)
void swap_mailbox(ms_ctx* c, ms_ctx::Mailbox& m) {
  if (trace_queue()) fprintf(stderr, "[msq] swap %s <-> %s\n", box_name(c, c->h_seq), box_name(c, m.h_seq));
  std::swap(c->h_scal, m.h_scal);
  std::swap(c->h_seq, m.h_seq);
  std::swap(c->d_h_seq, m.d_h_seq);
  for (int sl = 0; sl < MS_MB_WORDS; ++sl) std::swap(c->expected[sl], m.expected[sl]);
}
}  // namespace

// ---- line-search rounds (ms_step) ------------------------------------------------------------------------------
// A round = one ungated (or GO-gated) energy launch of n0 trials, n_st single-trial stages gated behind it, and the
// gradient + direction pass of the accepted point gated behind those.
namespace {
// plan a round from `alpha` on (prediction only: the same alphas are tested in the same order whatever is chosen)
void plan_round(ms_ctx* c, const ms_stepper_params* sp, double alpha, int room, int rejected_here, double a_hi,
                double r_lo, bool ls_warm, int pred_trials, int trials_so_far, RoundPlan& plan, bool ahead = false) {
  const bool multi_ok = c->pair_enable && c->side[0].partials != nullptr && (c->params.modules & MS_MOD_BENDING) != 0;
  const int n0_cap = !multi_ok ? 1 : (c->pair_lean_enable ? MS_MAX_TRIALS : (c->fK3 ? 3 : 2));
  double* alphas = plan.alphas;
  int n_alpha = 1;  // alphas of the ladder from here that are still >= 1e-8
  alphas[0] = alpha;
  while (n_alpha < room && n_alpha < MS_MAX_TRIALS + ms_ctx::SPEC_STAGES && alphas[n_alpha - 1] * sp->beta >= 1e-8) {
    alphas[n_alpha] = alphas[n_alpha - 1] * sp->beta;
    ++n_alpha;
  }
  int n0 = 1, n_st = 0;
  // A search that has rejected `escalate_after` alphas already will most likely reject more: from there on a round
  // evaluates as many trials as the search has rejected so far in ONE launch (the early ones of a launch cost an
  // energy-only evaluation each), whatever the history of the earlier searches says.
  const int force = c->pair_force;
  if (!force && c->escalate && multi_ok && rejected_here >= c->escalate_after) {
    n0 = std::max(1, std::min(std::min(n_alpha, n0_cap), rejected_here));
  } else if (!force && !ahead && c->escalate && multi_ok && !ls_warm && room >= 3 && pred_trials - trials_so_far >= room) {
    // no history of accepted alphas, and the last search of this kind ran out of its trials (in the cold phase of the
    // headline run: every search along a CG direction): this one is expected to as well -- as many trials per launch
    // as there are sets.  (One search that needed five trials says little about the next: an early accept in a
    // multi-trial launch costs more than the rounds it saves.  And not in a round queued ahead: whether that search
    // happens at all is a guess already.)
    n0 = std::min(n_alpha, n0_cap);
  } else {
    const bool can_spec = force || (ls_warm ? r_lo < INFINITY : pred_trials > 1);
    int depth = 1;
    if (can_spec) {
      // how many trials to queue: as many as the last search needed (cold), or -- once there is a history --
      // one per alpha that still lies above (most of) the range where alphas were accepted lately
      const int room4 = std::min(1 + ms_ctx::SPEC_STAGES, n_alpha);
      const int want = force ? std::min(std::min(force, 3), room4)
                             : (ls_warm ? room4 : std::min(pred_trials - trials_so_far, room4));
      while (depth < want) {
        if (!force && ls_warm && !(alphas[depth - 1] > 0.9 * a_hi)) break;
        ++depth;
      }
    }
    // two or more trials expected: the first two share one launch (and the ladder stops there for this round)
    // ... and only when the first one is expected to fail: it lies above every alpha accepted lately, and alphas
    // were rejected lately (a wasted evaluation costs more than a saved round trip gains)
    const bool pair = depth > 1 && multi_ok && (force || (ls_warm && alpha > 1.05 * a_hi && r_lo < INFINITY));
    if (pair) {
      depth = std::min(depth, 3);
      // triple launch: trial 1 is expected to fail as well -- its alpha is not below one that was rejected lately
      const bool triple = depth == 3 && n0_cap >= 3 && (force ? force == 4 : alphas[1] > r_lo);
      // otherwise the pair, and one gated trial behind it when trial 1 is as sure to fail as trial 0 (an empty gated
      // stage costs about what the host round trip it saves does, so "probably" is not enough)
      if (depth == 3 && !triple && !force && !(alphas[1] > a_hi)) depth = 2;
      n0 = triple ? 3 : 2;
      n_st = depth - n0;
    } else {
      n0 = 1;
      n_st = depth - 1;
    }
  }
  plan.n0 = n0;
  plan.n_st = n_st;
}

// queue a planned round into the mailboxes / decision records of `parity`.  merged: the round belongs to a step that
// has not started -- its first launch runs right behind the gradient pass of round `go_src`, whose direction scalars
// the round's first fold folds itself; that fold decides first whether the search happens at all (FoldArgs::go_kind)
// and forms the Armijo right-hand sides on the device.
int queue_round(ms_ctx* c, const ms_stepper_params* sp, const RoundPlan& plan, int parity, double energy0,
                double g_dot_d, bool merged, int go_src, bool carry_mode, bool cg, int restart) {
  const bool go_gated = merged;
  const int n0 = plan.n0, n_st = plan.n_st;
  const double* alphas = plan.alphas;
  c->cur_parity = parity;
  int rc;
  // first launch: trials 0 .. n0-1, the early ones into the side sets, the last one into the ordinary outputs;
  // its fold decides all of them and writes decision record 0
  {
    c->pair_on = n0 > 1 ? n0 : 0;
    c->pair_lean = n0 > 1 && c->pair_lean_enable;
    for (int j = 0; j + 1 < n0; ++j) c->pair_alpha[j] = alphas[j];
    swap_mailbox(c, c->first_mb[parity]);
    // merged: the fold (not the energy kernel) is tied to the gradient pass in front: it checks that pass's ran count
    c->cur_gate = merged ? c->kc_gate[go_src] : nullptr;
    c->cur_gate_want = DEC_ACCEPT_MAIN;
    c->cur_gate_fold_only = merged;
    c->cur_check_ran = merged;
    c->cur_dec = dec_word(c, parity, 0);
    c->cur_extra_mask = merged ? c->dir_mask[go_src] : 0u;
    c->cur_go_kind = merged ? c->ahead.go_kind : 0;
    c->cur_go = merged ? go_word(c, go_src) : nullptr;
    for (int j = 0; j < n0; ++j) c->cur_rhs[j] = energy0 + sp->c * alphas[j] * g_dot_d;
    rc = phase_energy(c, c->params.modules, true, alphas[n0 - 1], true, false, carry_mode);
    c->pair_on = 0;
    c->pair_lean = false;
    c->cur_dec = nullptr;
    c->cur_gate = nullptr;
    c->cur_gate_fold_only = false;
    c->cur_check_ran = false;
    c->cur_extra_mask = 0;
    c->cur_go_kind = 0;
    c->cur_go = nullptr;
    if (merged) c->dir_pending[go_src] = false;
    swap_mailbox(c, c->first_mb[parity]);
    if (rc) return rc;
  }
  // gated stages: stage s runs iff record s-1 says DEC_CONTINUE
  for (int s2 = 1; s2 <= n_st; ++s2) {
    swap_mailbox(c, c->spec[parity][s2 - 1]);
    c->cur_gate = dec_word(c, parity, s2 - 1);
    c->cur_gate_want = DEC_CONTINUE;
    c->cur_check_ran = true;
    c->cur_dec = dec_word(c, parity, s2);
    c->cur_rhs[0] = energy0 + sp->c * alphas[n0 + s2 - 1] * g_dot_d;
    c->cur_rhs_dev = go_gated ? go_rhs(c, go_src) + (n0 + s2 - 1) : nullptr;
    rc = phase_energy(c, c->params.modules, true, alphas[n0 + s2 - 1], true, false, carry_mode);
    c->cur_gate = nullptr;
    c->cur_check_ran = false;
    c->cur_dec = nullptr;
    c->cur_rhs_dev = nullptr;
    swap_mailbox(c, c->spec[parity][s2 - 1]);
    if (rc) return rc;
  }
  {
    // the next step's gradient pass in the state an acceptance produces (x <-> xt, CG history swapped, factors of
    // the accepted trial), gated on "the accepted trial is the one in the ordinary buffers"; every change of the
    // context is undone afterwards.  Its direction fold can open the round after this one (queue_ahead).
    const bool next_hist = cg && ((c->cg_iter_count + 1) % restart != 0);
    const bool s_factors = c->factors_valid, s_implicit = c->dir_implicit, s_pdneg = c->pd_neg_pg;
    const bool s_grad_valid = c->grad_valid, s_carry = c->carry_valid, s_maxg2 = c->maxg2_valid;
    double* const s_last_g = c->last_g;
    std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);
    if (cg) {
      std::swap(c->buf[MS_BUF_G], c->buf[MS_BUF_PG]);
      std::swap(c->buf[MS_BUF_D], c->buf[MS_BUF_PD]);
      c->pd_neg_pg = c->dir_implicit;
    }
    c->factors_valid = true;
    swap_mailbox(c, c->grad_mb[parity]);
    c->cur_gate = dec_word(c, parity, n_st);
    c->cur_gate_want = DEC_ACCEPT_MAIN;
    c->cur_check_ran = true;
    // inside ms_minimize the direction fold of this pass is left to the first fold of the round after this one
    c->defer_dir = c->ahead_allowed && sp->edge_fraction <= 0.0;
    c->dir_deferred_mask = 0;
    rc = queue_energy_and_gradient(c, sp->stepper, next_hist, /*skip_energy=*/true);
    c->dir_pending[parity] = c->defer_dir && c->dir_deferred_mask != 0;
    c->dir_mask[parity] = c->dir_deferred_mask;
    c->kc_gate[parity] = dec_word(c, parity, n_st);
    c->defer_dir = false;
    c->cur_gate = nullptr;
    c->cur_check_ran = false;
    swap_mailbox(c, c->grad_mb[parity]);
    if (cg) {
      std::swap(c->buf[MS_BUF_G], c->buf[MS_BUF_PG]);
      std::swap(c->buf[MS_BUF_D], c->buf[MS_BUF_PD]);
    }
    std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);
    c->factors_valid = s_factors;
    c->dir_implicit = s_implicit;
    c->pd_neg_pg = s_pdneg;
    c->last_g = s_last_g;
    c->grad_valid = s_grad_valid;
    c->carry_valid = s_carry;
    c->maxg2_valid = s_maxg2;
    if (rc) return rc;
  }
  return MS_OK;
}

// the direction fold of round `parity`'s gradient pass was left out and nobody merged it: launch it on its own
int flush_dir_fold(ms_ctx* c, int parity) {
  if (!c->dir_pending[parity]) return MS_OK;
  c->dir_pending[parity] = false;
  swap_mailbox(c, c->grad_mb[parity]);
  c->cur_gate = c->kc_gate[parity];
  c->cur_gate_want = DEC_ACCEPT_MAIN;
  c->cur_check_ran = true;
  const int rc = reduce_slots(c, c->dir_mask[parity]);
  c->cur_gate = nullptr;
  c->cur_check_ran = false;
  swap_mailbox(c, c->grad_mb[parity]);
  return rc;
}

// nobody will take the results of the round queued ahead.  ran 0: none of its kernels ran; 1: its first energy launch
// ran and everything behind it stayed out (the fold said DEC_STOP): the bending factors, the trial positions and the
// device scalars are that launch's now, the host's scalars and G still describe x; 2: any of it may have run (the
// gradient pass behind an acceptance writes the CG history buffers)
void drop_ahead(ms_ctx* c, int ran) {
  if (!c->ahead.valid) return;
  const int p = c->ahead.parity;
  forget(c->first_mb[p]);
  forget(c->grad_mb[p]);
  for (auto& m : c->spec[p]) forget(m);
  for (auto& sd : c->side) forget(sd.mb[p]);
  c->dir_pending[p] = false;
  c->ahead.valid = false;
  ++c->q_dropped;
  if (ran >= 1) c->factors_valid = false;
  if (ran >= 2) {
    c->carry_valid = c->grad_valid = c->maxg2_valid = false;
    c->kc_pending = false;
    c->cg_have_history = false;
    c->cg_iter_count = 0;
    c->pd_neg_pg = false;
  }
}

// The step has accepted a trial and the gradient pass of the new x is running (queued with the round, gated on the
// acceptance).  Queue the first round of the NEXT search behind it now: what the host does not know yet -- <g,d> of
// the new gradient, whether the iteration converges, whether the trials need the normal-rotation guard -- the
// direction fold of that pass decides from its own scalars (FoldArgs::go_out) with the parameters handed to it here.
//   kind 1: the pass computes a direction with CG history; in the steady state of the headline workload that is no
//           descent direction, the step fails without a trial, the stepper is reset and the step after it searches
//           along d = -g with the same step size: that search's first round is queued;
//   kind 2: the pass computes d = -g itself (gradient descent, CG restart steps): the next step's own first round.
int queue_ahead(ms_ctx* c, const ms_stepper_params* sp, const ms_step_result* out, double tol, bool carry_mode,
                bool cg, int restart) {
  if (!c->kc_pending || c->ahead.valid) return MS_OK;
  const int src = c->kc_parity;
  if (!c->dir_pending[src]) return MS_OK;  // (the pass was queued with its own direction fold)
  // kind 1: CG history, and the last direction with history was no descent direction; 3: it was one -- the next step
  // searches along the direction this pass writes; 2: the pass writes d = -g itself
  const int kind = c->kc_use_history ? (c->last_hist_descent ? 3 : 1) : 2;
  const double alpha0 = out->next_step;
  const double me2 = c->h_scal[MS_S_MINEDGE2];
  if (c->steps_left < (kind == 1 ? 2 : 1) || !(alpha0 >= 1e-8) || sp->edge_fraction > 0.0)
    return flush_dir_fold(c, src);  // (the step it would belong to is not part of this call)
  // line-search history as the consuming step will see it (accept() has just added this search)
  double a_hi = 0.0, r_lo = INFINITY;
  const ms_ctx::LsHist& lh = c->ls[kind == 3 ? 1 : 0];  // (kinds 1 and 2 search along -g)
  for (int k = 0; k < std::min(lh.n, (int)ms_ctx::LS_HIST); ++k) {
    a_hi = std::max(a_hi, lh.acc[k]);
    r_lo = std::min(r_lo, lh.rej[k]);
  }
  const int max_iter = sp->max_iter > 0 ? sp->max_iter : 10;
  ms_ctx::Ahead& ah = c->ahead;
  plan_round(c, sp, alpha0, max_iter, 0, a_hi, r_lo, lh.n >= 2, lh.pred_trials, 0, ah.plan, /*ahead=*/true);
  if (ah.plan.n0 + ah.plan.n_st > MS_MAX_TRIALS) return flush_dir_fold(c, src);  // (the fold forms that many right-hand sides)
  ah.kind = kind;
  ah.stepper = sp->stepper;
  ah.implicit = kind == 1;
  ah.alpha0 = alpha0;
  ah.energy0 = out->energy;
  ah.beta = sp->beta;
  ah.c1 = sp->c;
  ah.max_iter = max_iter;
  ah.tol2p = tol * tol * (1.0 + 1e-9);
  ah.lim = (c->til.nf > 0 && me2 > 0.0) ? (0.09 * me2 / (1.0 + 1e-9)) / (alpha0 * alpha0) : INFINITY;
  ah.src = src;
  ah.go = false;
  ah.go_known = false;
  // what the round's first fold tests before it decides the trials (FoldArgs::go_kind)
  ah.go_kind = kind == 1 ? 1 : 2;
  c->cur_go_val[0] = ah.tol2p;
  c->cur_go_val[1] = ah.lim;
  c->cur_go_val[2] = ah.energy0;
  c->cur_go_val[3] = sp->c;
  c->cur_go_val[4] = alpha0;
  c->cur_go_val[5] = sp->beta;
  // the round itself, in the state the consuming step will be in
  const bool s_hist = c->cg_have_history, s_implicit = c->dir_implicit, s_pdneg = c->pd_neg_pg;
  const int s_iter = c->cg_iter_count;
  double* const s_last_g = c->last_g;
  const bool s_kcp = c->kc_pending;
  const int s_kcpar = c->kc_parity;
  // (queueing a trial pass marks the carried state as gone; for a round that belongs to a later step it is not)
  const bool s_carry = c->carry_valid, s_grad = c->grad_valid, s_maxg2 = c->maxg2_valid, s_fac = c->factors_valid,
             s_bt = c->bt_valid;
  if (kind == 1) {  // (after the failed step: ms_reset_stepper, then the steepest-descent restart reads G with -alpha)
    c->cg_have_history = false;
    c->cg_iter_count = 0;
    c->pd_neg_pg = false;
    c->dir_implicit = true;
  } else {
    c->dir_implicit = false;
  }
  const int parity = c->next_parity;
  c->next_parity ^= 1;
  ah.parity = parity;
  // (host-side right-hand sides are not known yet: the device forms them; the consuming step replays them)
  int rc = queue_round(c, sp, ah.plan, parity, 0.0, 0.0, /*merged=*/true, src, carry_mode, cg, restart);
  c->cg_have_history = s_hist;
  c->cg_iter_count = s_iter;
  c->dir_implicit = s_implicit;
  c->pd_neg_pg = s_pdneg;
  c->last_g = s_last_g;
  c->kc_pending = s_kcp;
  c->kc_parity = s_kcpar;
  c->carry_valid = s_carry;
  c->grad_valid = s_grad;
  c->maxg2_valid = s_maxg2;
  c->factors_valid = s_fac;
  c->bt_valid = s_bt;
  if (rc) return rc;
  ah.valid = true;
  ++c->q_ahead;
  return MS_OK;
}
}  // namespace

int ms_step(ms_ctx* c, const ms_stepper_params* sp, double step_size, double tol,
            ms_step_result* out) {
  if (!c || !sp || !out) return fail(c, MS_ERR_INVALID, "ms_step: NULL argument");
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "ms_step: sharded contexts use the phase API");
  memset(out, 0, sizeof(*out));
  const bool cg = sp->stepper == MS_STEPPER_CG;
  const int restart = sp->restart_interval > 0 ? sp->restart_interval : 10;
  // conjugate_gradient.py:78-82: steepest descent on first call and every restart
  const bool use_history = cg && c->cg_have_history && (c->cg_iter_count % restart != 0);
  const bool tilt = (c->params.modules & MS_ANY_TILT_MODS) != 0;
  TiltField* tfl[3];
  const int n_tf = active_fields(c, c->params.modules, tfl);
  // reuse_energy0 == 2: an accepted trial doubles as the next step's energy/factor pass
  const bool carry_mode = sp->reuse_energy0 >= 2 && !tilt;
  if (c->ahead.valid && !c->ahead.go_known && !(c->kc_pending && carry_mode && c->carry_valid)) {
    // a round was queued ahead, and this step will not read the direction fold that decides about it (something
    // touched the context in between): whatever it did, none of it is used
    drop_ahead(c, /*ran=*/2);
  }
  // the mailbox energies (and G, when grad_valid) describe x ...
  const bool carried_x = carry_mode && c->carry_valid;
  // ... and so do the bending factors in fK / fA (what a gradient pass at x needs)
  const bool carried = carried_x &&
                       (c->factors_valid || !(c->params.modules & (MS_MOD_BENDING | MS_MOD_BENDING_TILT)));
  // the direction cannot ride in the gradient kernel's epilogue when a constraint row has to be projected out first
  // (lambda needs a global reduction) or when a tilt module adds its shape gradient behind K_C
  const bool volrow = (c->params.modules & MS_CON_VOLUME) != 0;
  // conjugate_gradient.py:74-76: the direction comes from the row-normalised gradient -- never an implicit -G, never
  // the fused epilogue, and no queued rounds (their gradient pass carries the fused epilogue)
  const bool precond = cg && sp->precondition != 0;
  c->precond = precond;
  int rc;
  bool restart_sd = false;
  const bool tilt_shape = (c->params.modules & MS_TILT_SHAPE_MODS) != 0;
  if (carried_x && c->grad_valid && !tilt_shape && c->til.T <= 256 &&
      (!volrow || (!use_history && c->maxg2_valid && !precond))) {
    // x has not moved since the last gradient pass (failed search, stepper reset): only the
    // direction changes.  k_direction on the finalized g repeats the fused epilogue's
    // arithmetic and reduction order exactly.  (With a constraint row G is the projected gradient the direction
    // kernel wrote back: a steepest-descent restart reads it as it is; projecting it a second time is not on.)
    restart_sd = !use_history && c->maxg2_valid && !precond;
    if (restart_sd) {
      // steepest-descent restart: d = -g, whose scalars the gradient pass already reduced
      // (|g|^2; <g,d> = -|g|^2 and max|d_i|^2 = max|g_i|^2 exactly) -- no fold, no host round trip
      // -- and no kernel either: the trial passes read G with -alpha (dir_implicit)
      c->dir_implicit = true;
      c->last_g = c->buf[MS_BUF_G];
      c->h_scal[MS_S_GDOTD] = -c->h_scal[MS_S_GNORM2];
      c->h_scal[MS_S_MAXD2] = c->h_scal[MS_S_MAXG2];
      rc = MS_OK;
    } else {
      rc = phase_direction(c, sp->stepper, use_history, /*g_finalized=*/true);
    }
  } else if (c->kc_pending && carried_x && !(c->params.modules & MS_TILT_SHAPE_MODS) && c->kc_stepper == sp->stepper &&
             c->kc_use_history == use_history && !precond) {
    // (a queued pass always carries the plain fused direction: a caller that switched precondition on between two
    // steps must not adopt it)
    // the gradient + direction pass of this x was queued behind the line search that accepted it (gated on the
    // acceptance) and has run: take its scalars from its mailbox
    c->kc_pending = false;
    double vals[MS_NSCAL];
    if (c->ahead.valid && !c->ahead.go_known && c->ahead.src == c->kc_parity) {
      // The direction scalars of this pass arrive with the first fold of the round that was queued behind it (its
      // energy launch did not wait for them).  That fold decided first whether the round's search happens at all:
      // replay that from the scalars it was taken on (comparisons only -- the host's outcome is the device's, or
      // the queue is broken).
      ms_ctx::Ahead& ah = c->ahead;
      uint32_t code = DEC_NONE;
      rc = wait_mailbox(c, c->first_mb[ah.parity].h_seq, c->first_mb[ah.parity].expected, vals, &code);
      if (rc) return rc;
      ah.go_known = true;
      {
        const double gn2 = vals[MS_S_GNORM2], gdd = vals[MS_S_GDOTD];
        const bool kind_ok = ah.kind == 1 ? gdd >= 0.0 : gdd < 0.0;
        const bool go = kind_ok && gn2 > ah.tol2p && (ah.kind == 1 ? vals[MS_S_MAXG2] : vals[MS_S_MAXD2]) < ah.lim;
        if (go != (code != DEC_STOP)) {
          rc = verify_decision(c, go ? DEC_GO : DEC_STOP, code, "the fold that opens a round queued ahead");
          if (rc) return rc;
        }
        ah.go = go;
      }
      // its energy launch has run in any case: without the search, the factors and the trial positions are scrap
      if (!ah.go) drop_ahead(c, /*ran=*/1);
    } else {
      rc = flush_dir_fold(c, c->kc_parity);  // (nobody merged the direction fold: launch it now)
      if (rc) return rc;
      rc = wait_mailbox(c, c->grad_mb[c->kc_parity].h_seq, c->grad_mb[c->kc_parity].expected, vals, nullptr);
      if (rc) return rc;
    }
    for (int sl = 0; sl < MS_NSCAL; ++sl)
      if (MASK_DIR & (1u << sl)) put_mailbox(c, sl, vals[sl]);
    c->last_g = c->buf[MS_BUF_G];
    c->dir_implicit = false;
    c->maxg2_valid = true;  // (the fused epilogue and the direction kernel both reduce max|g_i|^2)
  } else {
    c->kc_pending = false;
    rc = queue_energy_and_gradient(c, sp->stepper, use_history, carried);
    c->maxg2_valid = true;  // the fused epilogue / the direction kernel reduced max|g_i|^2 as well
  }
  if (rc) return rc;
  if (!restart_sd) {
    rc = fetch(c);
    if (rc) return rc;
  }
  // factors, mailbox energies and G now describe x (until a trial pass overwrites them)
  c->carry_valid = carry_mode;
  c->grad_valid = carry_mode && !tilt_shape;
  double e[4];
  energies_from_mailbox(c, e);
  const double E_eval = e[0] + e[1] + e[2] + e[3];
  const double grad_norm = std::sqrt(c->h_scal[MS_S_GNORM2]);
  const double g_dot_d = c->h_scal[MS_S_GDOTD];
  const double max_dir = std::sqrt(c->h_scal[MS_S_MAXD2]);
  out->energy_eval = E_eval;
  out->grad_norm = grad_norm;
  out->g_dot_d = g_dot_d;
  out->volume = c->h_scal[MS_S_VOL];
  out->next_step = step_size;
  out->energy = E_eval;
  if (use_history) c->last_hist_descent = g_dot_d < 0.0;  // (what queue_ahead expects of the next direction with history)
  if (grad_norm < tol) {  // minimizer.py:1324
    out->converged = 1;
    out->success = 1;
    return MS_OK;
  }
  // ---- backtracking_line_search_array (line_search.py:267-426) -------------
  double energy0 = E_eval;
  double min_edge = std::sqrt(c->h_scal[MS_S_MINEDGE2]);
  for (int k = 0; k < n_tf; ++k) {  // energy_fn projects the stored tilts first (minimizer.py:581-588)
    rc = tilt_pass_f(c, *tfl[k], 2, false, 0.0);
    if (rc) return rc;
  }
  if (!sp->reuse_energy0 || tilt) {
    rc = phase_energy(c, c->params.modules, false, 0.0, false, false, false);
    if (rc) return rc;
    rc = fetch(c);
    if (rc) return rc;
    energies_from_mailbox(c, e);
    energy0 = e[0] + e[1] + e[2] + e[3];
    min_edge = std::sqrt(c->h_scal[MS_S_MINEDGE2]);
  }
  if (c->til.nf == 0) min_edge = 0.0;
  out->energy = energy0;
  const double safe_step_limit = min_edge > 0.0 ? 0.3 * min_edge : INFINITY;
  if (g_dot_d >= 0.0) return MS_OK;  // :325-328 non-descent: (False, step_size, energy0)
  double alpha = step_size;
  if (sp->edge_fraction > 0.0 && min_edge > 0.0 && max_dir > 0.0)
    alpha = std::min(alpha, sp->edge_fraction * min_edge / max_dir);
  const double alpha_max = sp->alpha_max_factor * step_size;
  const int max_iter = sp->max_iter > 0 ? sp->max_iter : 10;
  bool kc_queued = false;  // the next step's gradient pass is in the queue, gated on an acceptance
  double min_rejected = INFINITY;  // smallest alpha this search has rejected
  // what the recent searches say about the acceptance threshold: no alpha above a_hi was accepted, and alphas
  // down to r_lo were rejected (INFINITY: nothing was rejected lately -- the step size is still growing)
  double a_hi = 0.0, r_lo = INFINITY;
  ms_ctx::LsHist& lh = c->ls[use_history ? 1 : 0];
  for (int k = 0; k < std::min(lh.n, (int)ms_ctx::LS_HIST); ++k) {
    a_hi = std::max(a_hi, lh.acc[k]);
    r_lo = std::min(r_lo, lh.rej[k]);
  }
  const bool ls_warm = lh.n >= 2;
  // what an accepted trial at `alpha` does (positions, carry flags, CG history, result fields)
  auto accept = [&](double alpha_acc, double E_t) {
    std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);
    // carry mode: the trial pass evaluated exactly the accepted x (it wrote those very
    // doubles to xt) with the factor outputs on -> it IS the next step's energy pass
    c->factors_valid = carry_mode;
    c->carry_valid = carry_mode;
    c->grad_valid = false;
    // minimizer.py:1415 re-projects the stored tilts onto the accepted surface: that is
    // exactly the trial projection computed above
    for (int k = 0; k < n_tf; ++k) std::swap(tfl[k]->tilts, tfl[k]->trial);
    c->bt_valid = (c->params.modules & MS_MOD_BENDING_TILT) != 0;  // the trial's record is x's now
    if (cg) {  // conjugate_gradient.py:114-117 history on success only
      std::swap(c->buf[MS_BUF_G], c->buf[MS_BUF_PG]);
      std::swap(c->buf[MS_BUF_D], c->buf[MS_BUF_PD]);
      c->pd_neg_pg = c->dir_implicit;  // the accepted direction was -G = -PG from now on
      c->last_g = c->buf[MS_BUF_PG];
      c->cg_have_history = true;
      ++c->cg_iter_count;
    }
    out->success = 1;
    out->alpha = alpha_acc;
    out->energy = E_t;
    out->volume = c->h_scal[MS_S_VOL];
    out->next_step = std::min(alpha_acc * sp->gamma, alpha_max);
    lh.pred_trials = std::max(1, out->trials);
    // an alpha far below everything accepted lately: the step-size regime has changed, the history predicts nothing
    if (c->ls_reset && lh.n > 0 && alpha_acc < 0.5 * a_hi) lh.n = 0;
    lh.acc[lh.n % ms_ctx::LS_HIST] = alpha_acc;
    lh.rej[lh.n % ms_ctx::LS_HIST] = min_rejected;
    ++lh.n;
    c->kc_pending = kc_queued;
  };
  // the queue needs: carry mode (a trial is a complete energy pass), energies the device can add up the way the
  // host does (surface + bending only), no tilt projections between trials.  The gradient + direction pass of the
  // accepted point follows in the same queue, gated on the acceptance (with a constraint row: K_C, the fold of
  // <g,gC> / <gC,gC>, the direction kernel and its fold, all four behind the same decision word).
  // the enforcer lane (ms_stepper_params.enforce_volume): every trial is projected onto the target volume before its
  // energy is taken -- three more passes per trial, one trial at a time
  const bool enforce = sp->enforce_volume != 0 && volrow && !tilt;
  const bool can_chain = c->speculate && carry_mode && !tilt && !(c->params.modules & MS_MOD_VOLUME_PENALTY) && !enforce &&
                         !precond;
  // a round queued by the step before (while its gradient pass was running): the round of THIS search's first
  // iteration if it was queued for exactly what this step has computed by itself
  bool adopted = false;
  if (c->ahead.valid) {
    const ms_ctx::Ahead& ah = c->ahead;
    const bool same = ah.go && can_chain && ah.stepper == sp->stepper && ah.implicit == c->dir_implicit &&
                      ah.alpha0 == alpha && ah.energy0 == energy0 && ah.beta == sp->beta && ah.c1 == sp->c &&
                      ah.max_iter == max_iter && alpha * max_dir < safe_step_limit;
    if (same) {
      adopted = true;
    } else {
      // its kernels ran for a search that is not this one: the factors, trial positions and device scalars of x are
      // theirs now -- once more from x, without it
      drop_ahead(c, /*ran=*/2);
      return ms_step(c, sp, step_size, tol, out);
    }
  }
  int it = 0;
  bool unchained_ran = false;  // a trial went through the context's own mailbox (its scalars replaced x's there)
  const bool s_maxg2_x = c->maxg2_valid;
  while (it < max_iter) {
    const bool safe_small = alpha * max_dir < safe_step_limit;
    kc_queued = false;  // (a gradient pass queued behind an earlier, fully rejected round found its gate closed)
    if (!(can_chain && safe_small)) {
      unchained_ran = true;
      rc = phase_energy(c, c->params.modules, true, alpha, true, !safe_small, carry_mode && !enforce);
      if (rc) return rc;
      rc = fetch(c);
      if (rc) return rc;
      if (!safe_small && c->h_scal[MS_S_GUARD] > 0.0) {
        ++out->guard_rejects;
        min_rejected = alpha;
        alpha *= sp->beta;
        ++it;
        if (alpha < 1e-8) break;
        continue;
      }
      if (enforce) {
        // line_search.py:448-452: constraint_enforcer(mesh) on the trial positions, then energy_fn() there.  The
        // projection works on buffer X: the trial takes that place for its duration.
        std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);
        rc = ms_project_volume_cached(c, c->params.target_volume, 1e-12, 3, 0, nullptr, nullptr);
        if (rc == MS_OK) rc = phase_energy(c, c->params.modules, false, 0.0, false, false, carry_mode);
        if (rc == MS_OK) rc = fetch(c);
        std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);  // (the projected trial is the trial buffer again)
        if (rc) return rc;
        if (carry_mode) c->factors_valid = false;  // (they belong to the trial point until it is accepted)
      }
      ++out->trials;
      energies_from_mailbox(c, e);
      const double E_t = e[0] + e[1] + e[2] + e[3];
      if (E_t <= energy0 + sp->c * alpha * g_dot_d) {
        accept(alpha, E_t);
        return MS_OK;
      }
      // a rejected trial restores the positions, not the tilts: energy_fn stored their projection
      // onto the trial surface (line_search.py:456-487 without an enforcer)
      for (int k = 0; k < n_tf; ++k) std::swap(tfl[k]->tilts, tfl[k]->trial);
      min_rejected = alpha;
      alpha *= sp->beta;
      ++it;
      if (alpha < 1e-8) break;
      continue;
    }
    rc = spec_prepare(c);
    if (rc) return rc;
    RoundPlan plan;
    int parity;
    if (adopted) {
      plan = c->ahead.plan;
      parity = c->ahead.parity;
      c->ahead.valid = false;
      adopted = false;
      kc_queued = true;  // (queue_round queued it with the round)
      c->kc_stepper = sp->stepper;
      c->kc_use_history = cg && ((c->cg_iter_count + 1) % restart != 0);
      ++c->q_adopted;
    } else {
      plan_round(c, sp, alpha, max_iter - it, out->trials + out->guard_rejects, a_hi, r_lo, ls_warm, lh.pred_trials,
                 out->trials, plan);
      parity = c->next_parity;
      c->next_parity ^= 1;
      rc = queue_round(c, sp, plan, parity, energy0, g_dot_d, /*merged=*/false, 0, carry_mode, cg, restart);
      if (rc) return rc;
      kc_queued = true;
      c->kc_stepper = sp->stepper;
      c->kc_use_history = cg && ((c->cg_iter_count + 1) % restart != 0);
    }
    c->kc_parity = parity;
    const int n0 = plan.n0, n_st = plan.n_st, n_round = n0 + n_st;
    const double* const alphas = plan.alphas;
    double rhs[MS_MAX_TRIALS + ms_ctx::SPEC_STAGES];
    for (int j = 0; j < n_round; ++j) rhs[j] = energy0 + sp->c * alphas[j] * g_dot_d;
    ++c->q_rounds;
    // ---- take the results in order, replaying the device's decisions from the same doubles --------------------
    // first launch: every set was folded by ONE k_reduce launch, so they land together
    double v[MS_MAX_TRIALS][MS_NSCAL];
    uint32_t dev_code = DEC_NONE;
    for (int j = 0; j + 1 < n0; ++j) {
      rc = wait_mailbox(c, c->side[j].mb[parity].h_seq, c->side[j].mb[parity].expected, v[j], nullptr);
      if (rc) return rc;
    }
    rc = wait_mailbox(c, c->first_mb[parity].h_seq, c->first_mb[parity].expected, v[n0 - 1], &dev_code);
    if (rc) return rc;
    int acc = -1;
    double E_acc = 0.0;
    for (int j = 0; j < n0 && acc < 0; ++j) {
      ++out->trials;
      const double E_t = ((c->params.modules & MS_MOD_SURFACE) ? v[j][MS_S_ESURF] : 0.0) +
                         ((c->params.modules & MS_MOD_BENDING) ? v[j][MS_S_EBEND] : 0.0);
      if (E_t <= rhs[j]) {
        acc = j;
        E_acc = E_t;
      } else {
        min_rejected = alphas[j];
      }
    }
    rc = verify_decision(c, acc < 0 ? DEC_CONTINUE : (acc == n0 - 1 ? DEC_ACCEPT_MAIN : DEC_ACCEPT_SIDE), dev_code,
                         "the first launch of a round");
    if (rc) return rc;
    if (n0 > 1) {
      ++c->q_multi;
      c->q_wasted += acc < 0 ? 0 : n0 - 1 - acc;
    }
    if (acc >= 0)
      for (int s2 = 0; s2 < n_st; ++s2) forget(c->spec[parity][s2]);  // (the gated stages stay out)
    if (acc >= 0 && acc < n0 - 1) {
      // the unexpected case: an early trial was accepted.  The gradient pass queued behind stays out (DEC_ACCEPT_SIDE).
      ++c->q_side_accepts;
      kc_queued = false;
      forget(c->grad_mb[parity]);
      c->dir_pending[parity] = false;
      if (c->pair_lean_enable) {
        // it was evaluated for its energies only -- evaluate it again, alone and with every output.  With fixed-order
        // sums the energies come out bit for bit as before; with LDS atomics in their last bits, like any
        // re-evaluation.
        rc = phase_energy(c, c->params.modules, true, alphas[acc], true, false, carry_mode);
        if (rc) return rc;
        rc = fetch(c);
        if (rc) return rc;
      } else {
        // (MS_PAIR_LEAN=0) this trial's positions and factors are in a side set
        const size_t nvp = (size_t)c->til.nvp;
        const double* sx = acc == 0 ? c->xt2 : c->xt3;
        const double* sk = acc == 0 ? c->fK2 : c->fK3;
        const double* sa = acc == 0 ? c->fA2 : c->fA3;
        HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_XT], sx, sizeof(double) * 3 * nvp, hipMemcpyDeviceToDevice, S(c)));
        HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_FK], sk, sizeof(double) * 3 * nvp, hipMemcpyDeviceToDevice, S(c)));
        HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_FA], sa, sizeof(double) * 2 * nvp, hipMemcpyDeviceToDevice, S(c)));
        for (int sl = 0; sl < MS_NSCAL; ++sl)
          if (energy_mask(c->params.modules) & (1u << sl)) put_mailbox(c, sl, v[acc][sl]);
      }
      accept(alphas[acc], E_acc);
      return MS_OK;
    }
    if (acc == n0 - 1) {
      for (int sl = 0; sl < MS_NSCAL; ++sl)
        if (energy_mask(c->params.modules) & (1u << sl)) put_mailbox(c, sl, v[acc][sl]);
      accept(alphas[acc], E_acc);
      if (c->ahead_allowed) {
        rc = queue_ahead(c, sp, out, tol, carry_mode, cg, restart);
        if (rc) return rc;
      }
      return MS_OK;
    }
    // gated stages, in order
    bool accepted = false;
    for (int s2 = 1; s2 <= n_st && !accepted; ++s2) {
      double vals[MS_NSCAL];
      uint32_t code = DEC_NONE;
      rc = wait_mailbox(c, c->spec[parity][s2 - 1].h_seq, c->spec[parity][s2 - 1].expected, vals, &code);
      if (rc) return rc;
      ++out->trials;
      const double E_t = ((c->params.modules & MS_MOD_SURFACE) ? vals[MS_S_ESURF] : 0.0) +
                         ((c->params.modules & MS_MOD_BENDING) ? vals[MS_S_EBEND] : 0.0);
      const bool ok = E_t <= rhs[n0 + s2 - 1];
      rc = verify_decision(c, ok ? DEC_ACCEPT_MAIN : DEC_CONTINUE, code, "a gated stage");
      if (rc) return rc;
      if (ok) {  // the device took the same decision from the same doubles: later stages stay out
        for (int s3 = s2; s3 < n_st; ++s3) forget(c->spec[parity][s3]);
        for (int sl = 0; sl < MS_NSCAL; ++sl)
          if (energy_mask(c->params.modules) & (1u << sl)) put_mailbox(c, sl, vals[sl]);
        accept(alphas[n0 + s2 - 1], E_t);
        accepted = true;
      } else {
        min_rejected = alphas[n0 + s2 - 1];
      }
    }
    if (accepted) {
      if (c->ahead_allowed) {
        rc = queue_ahead(c, sp, out, tol, carry_mode, cg, restart);
        if (rc) return rc;
      }
      return MS_OK;
    }
    forget(c->grad_mb[parity]);  // every trial of the round was rejected: the gradient pass behind it stayed out
    c->dir_pending[parity] = false;
    it += n_round;
    alpha = alphas[n_round - 1] * sp->beta;
    if (alpha < 1e-8) break;
  }
  const double reduced = std::max(alpha * sp->beta, 0.0);  // :425-426
  out->next_step = std::max(reduced, step_size * sp->beta);
  if (c->ls_reset) lh.n = 0;  // a search that ran out of trials: same
  lh.pred_trials = std::max(1, out->trials + out->guard_rejects);
  if (carry_mode && !unchained_ran) {
    // every trial was rejected and x has not moved: G and the host's scalars still describe x (the queued rounds post
    // to mailboxes of their own); only the bending factors in fK / fA and the device scalars are the last trial's
    c->carry_valid = true;
    c->grad_valid = !tilt_shape;
    c->maxg2_valid = s_maxg2_x;
    c->factors_valid = false;
  }
  return MS_OK;
}

int ms_project_volume_cached(ms_ctx* c, double target, double tol, int max_iter, int first_step_cached,
                             int* iters_out, double* volume_out) {
  if (!c) return MS_ERR_INVALID;
  if (c->shard_count != 1) return fail(c, MS_ERR_STATE, "ms_project_volume: single shard only");
  const size_t row_bytes = sizeof(double) * 3 * (size_t)c->til.nvp;
  if (!c->d_volgrad_cache) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_volgrad_cache), row_bytes));
    HIPCHK(c, hipMemsetAsync(c->d_volgrad_cache, 0, row_bytes, S(c)));
  }
  int it = 0;
  double V = 0.0;
  for (; it < max_iter; ++it) {
    int rc = phase_energy(c, MS_CON_VOLUME, false, 0.0, false, false, false);
    if (rc) return rc;
    rc = fetch(c);
    if (rc) return rc;
    V = c->h_scal[MS_S_VOL];
    const double delta = V - target;
    // compute_volume_and_gradient (body.py:386-470): the gradient comes with the volume -- from Body's cache on the
    // first pass when the caller says the cached volume is current (a compute_volume at this mesh version preceded,
    // which refreshes the cached volume and version but not the gradient), freshly evaluated (and cached) otherwise
    const bool stale = it == 0 && first_step_cached != 0 && c->volgrad_cache_valid;
    const double* g = c->d_volgrad_cache;
    double norm2 = c->volgrad_cache_norm2;
    if (!stale) {
      rc = phase_gradient(c, MS_CON_VOLUME, nullptr, false);
      if (rc) return rc;
      rc = fetch(c);
      if (rc) return rc;
      norm2 = c->h_scal[MS_S_GCGC];
      HIPCHK(c, hipMemcpyAsync(c->d_volgrad_cache, c->buf[MS_BUF_GC], row_bytes, hipMemcpyDeviceToDevice, S(c)));
      c->volgrad_cache_norm2 = norm2;
      c->volgrad_cache_valid = true;
    }
    if (std::fabs(delta) < tol) break;
    const double lam = delta / (norm2 + 1e-12);
    HIPCHK(c, launch_axpy_masked(c->til.nv, c->d_vflags, c->buf[MS_BUF_X], g, -lam, c->stream));
    c->factors_valid = false;
    c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  }
  if (iters_out) *iters_out = it;
  if (volume_out) *volume_out = V;
  return MS_OK;
}

int ms_project_volume(ms_ctx* c, double target, double tol, int max_iter, int* iters_out,
                      double* volume_out) {
  return ms_project_volume_cached(c, target, tol, max_iter, 0, iters_out, volume_out);
}

namespace {
constexpr int RES_CHUNK = 4096;  // steps per launch (rows of the device step log)

// can the steps of this ms_minimize call run in the resident kernel?  (surface + volume row, gradient descent, plain
// Armijo search with evaluation reuse; everything else keeps the kernel-per-phase path)
bool resident_eligible(ms_ctx* c, const ms_minimize_params* mp) {
  const uint32_t mods = c->params.modules;
  const ms_stepper_params& sp = mp->stepper;
  if (!c->resident_enable || c->profiling || c->exec_on) return false;
  if (c->shard_count != 1 || c->comm || c->allgather_cb || c->peer_on) return false;
  if (c->til.T != 256 || c->til.own != 256 || !c->d_tile_facets32 || c->til.n_tiles < 2 || c->til.n_tiles > 2048) return false;
  if (!(mods & MS_MOD_SURFACE) || (mods & ~(MS_MOD_SURFACE | MS_CON_VOLUME | MS_TRACK_VOLUME))) return false;
  if (sp.stepper != MS_STEPPER_GD || sp.precondition || sp.enforce_volume || sp.reuse_energy0 < 2 || sp.edge_fraction > 0.0)
    return false;
  if (mp->relax_tilts || mp->fixed_step_mode) return false;
  if (c->resident_ok < 0) {
    c->resident_lds = resident_lds_bytes(c->cap, c->til.max_ent, c->til.max_tile_facets, false);
    int ok = 0;
    if (resident_fits(c->til.n_tiles, c->resident_lds, c->device, &ok) != hipSuccess) {
      (void)hipGetLastError();
      ok = 0;
    }
    c->resident_ok = ok;
    if (trace_steps())
      fprintf(stderr, "[mss] resident step kernel: %d tiles, %zu bytes of LDS per workgroup -> co-resident: %d\n",
              c->til.n_tiles, c->resident_lds, ok);
  }
  return c->resident_ok == 1;
}

struct ResidentOutcome {
  int steps = 0, reason = RES_DONE;
  double step_size = 0.0, energy = 0.0, volume = 0.0;
};

// up to max_steps steps in one launch; the step log rows land in c->h_res_log
int resident_run(ms_ctx* c, const ms_minimize_params* mp, int max_steps, double step_size, ResidentOutcome* ro) {
  const Tiling& t = c->til;
  const ms_stepper_params& sp = mp->stepper;
  const int n = std::min(max_steps, RES_CHUNK);
  if (!c->d_res_partials) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_res_partials), sizeof(double) * 4 * MS_NPART * (size_t)t.n_tiles));
    HIPCHK(c, hipMemset(c->d_res_partials, 0, sizeof(double) * 4 * MS_NPART * (size_t)t.n_tiles));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_res_bar), sizeof(unsigned int) * RESIDENT_BAR_WORDS));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_res_log), sizeof(double) * 8 * RES_CHUNK));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_res_result), sizeof(double) * 16));
  }
  ResidentArgs a;
  a.m = device_mesh(c);
  a.x = c->buf[MS_BUF_X];
  a.d = c->buf[MS_BUF_D];
  a.g = c->buf[MS_BUF_G];
  a.gC = c->buf[MS_BUF_GC];
  a.partials = c->d_res_partials;
  a.bar = c->d_res_bar;
  a.log = c->d_res_log;
  a.result = c->d_res_result;
  a.n_steps = n;
  a.volrow = (c->params.modules & MS_CON_VOLUME) ? 1 : 0;
  a.want_vol = (c->params.modules & (MS_CON_VOLUME | MS_TRACK_VOLUME)) ? 1 : 0;
  a.atomic = c->deterministic ? 0 : 1;
  a.max_iter = sp.max_iter > 0 ? sp.max_iter : 10;
  a.step_size = step_size;
  a.tol = mp->tol;
  a.c1 = sp.c;
  a.beta = sp.beta;
  a.gamma = sp.gamma;
  a.alpha_max_factor = sp.alpha_max_factor;
  a.drift_check = mp->drift_check;
  a.target_volume = mp->target_volume;
  a.volume_tolerance = mp->volume_tolerance;
  a.cap = c->cap;
  a.max_ent = t.max_ent;
  HIPCHK(c, hipMemsetAsync(c->d_res_bar, 0, sizeof(unsigned int) * RESIDENT_BAR_WORDS, S(c)));
  HIPCHK(c, launch_resident(a, c->resident_lds, c->stream));
  double res[16];
  HIPCHK(c, hipMemcpyAsync(res, c->d_res_result, sizeof(res), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ro->steps = (int)res[0];
  ro->reason = (int)res[1];
  ro->step_size = res[2];
  ro->energy = res[3];
  ro->volume = res[4];
  ++c->resident_launches;
  c->resident_steps += ro->steps;
  if (trace_steps())
    fprintf(stderr, "[mss] resident launch: asked %d steps from step size %.3e -> took %d, reason %d, %d barriers\n", n,
            step_size, ro->steps, ro->reason, (int)res[7]);
  if (trace_steps() && ro->steps > 0)
    fprintf(stderr, "[mss]   per step (us, workgroup 0): gradient %.2f | barrier %.2f | lambda+rows+direction %.2f | trial energies "
                    "%.2f | barrier %.2f | fold %.2f\n", res[8] / ro->steps, res[9] / ro->steps, res[10] / ro->steps,
            res[11] / ro->steps, res[12] / ro->steps, res[13] / ro->steps);
  if (ro->steps > 0) {
    c->h_res_log.resize((size_t)8 * ro->steps);
    HIPCHK(c, hipMemcpy(c->h_res_log.data(), c->d_res_log, sizeof(double) * 8 * (size_t)ro->steps, hipMemcpyDeviceToHost));
  }
  // x may have moved, and the launch used the G / GC / D buffers for its own (raw) rows even when it declined its first
  // step: nothing the step logic carries from earlier evaluations is valid any more
  c->carry_valid = c->grad_valid = c->factors_valid = c->maxg2_valid = c->bt_valid = false;
  c->kc_pending = false;
  c->dir_implicit = false;
  if (ro->reason == RES_TIMEOUT) return fail(c, MS_ERR_STATE, "resident step kernel: a grid barrier timed out");
  return MS_OK;
}
}  // namespace

int ms_minimize(ms_ctx* c, const ms_minimize_params* mp, int n_steps, ms_minimize_result* out,
                double* step_log) {
  if (!c || !mp || !out) return fail(c, MS_ERR_INVALID, "ms_minimize: NULL argument");
  memset(out, 0, sizeof(*out));
  double step_size = mp->step_size;
  int zero_steps = 0;
  out->step_success = 1;
  out->step_size = step_size;
  const bool resident = resident_eligible(c, mp);
  int resident_cooldown = 0;       // iterations to leave to the ordinary path after the kernel declined a step
  bool res_energy_valid = false;   // the last step taken was the resident kernel's: its energy is the current x's
  double res_energy = 0.0;
  for (int i = 0; i < n_steps; ++i) {
    int rc;
    if (resident && resident_cooldown == 0 && n_steps - i >= 2) {
      // ---- as many steps as it will take in ONE launch (ms_resident.inc); bookkeeping per step as below ------------
      drop_ahead(c, /*ran=*/2);
      ResidentOutcome ro;
      rc = resident_run(c, mp, n_steps - i, step_size, &ro);
      if (rc) return rc;
      for (int k = 0; k < ro.steps; ++k) {
        const double* row = c->h_res_log.data() + 8 * (size_t)k;
        out->iterations = i + k + 1;
        out->energy_eval = row[3];
        out->grad_norm = row[4];
        if (step_log) memcpy(step_log + 8 * (size_t)(i + k), row, sizeof(double) * 8);
        out->step_success = 1;
        out->volume_cache_current = mp->drift_check ? 1 : 0;
        out->trials += (int)row[7];
        step_size = row[1];
        ++out->accepted;
        out->moved = 1;
        out->step_size = step_size;
        zero_steps = 0;
      }
      if (ro.steps > 0) {
        res_energy_valid = true;
        res_energy = ro.energy;
      }
      i += ro.steps;
      if (ro.reason == RES_DRIFT) {  // :1478-1513, for the step just taken
        if (mp->project_on_drift) {
          int iters = 0;
          rc = ms_project_volume_cached(c, mp->target_volume, 1e-12, 12, 1, &iters, nullptr);
          if (rc) return rc;
          out->volume_cache_current = 0;
          res_energy_valid = false;
        }
        ms_reset_stepper(c);
      } else if (ro.reason != RES_DONE) {
        // convergence, guard range, an exhausted search, a non-descent direction: this iteration through the ordinary
        // path (x is as the last completed step left it), and a few more before the kernel is asked again
        resident_cooldown = ro.reason == RES_CONVERGED ? 0 : 8;
        ++c->resident_bails;
      }
      if (i >= n_steps) break;
      if (ro.reason == RES_DONE || ro.reason == RES_DRIFT) {
        --i;  // (the loop increment: the next iteration is i)
        continue;
      }
    } else if (resident_cooldown > 0) {
      --resident_cooldown;
    }
    res_energy_valid = false;
    if (mp->relax_tilts) {  // minimizer.py:1237-1307: before the convergence check
      rc = (c->params.modules & MS_LEAFLET_MODS) ? ms_relax_leaflet_tilts(c, &mp->relax, nullptr, nullptr)
                                                  : ms_relax_tilts(c, &mp->relax, nullptr, nullptr);
      if (rc) return rc;
      out->moved = 1;
    }
    const double step_in = mp->fixed_step_mode ? mp->fixed_step : step_size;
    ms_step_result r;
    // rounds may be queued ahead of the step they belong to while this loop is the only thing touching the context
    c->steps_left = n_steps - 1 - i;
    c->ahead_allowed = c->ahead_enable && i + 1 < n_steps && !mp->relax_tilts && !mp->fixed_step_mode &&
                       c->shard_count == 1 && !c->comm && !c->allgather_cb && !c->peer_on;
    rc = (c->shard_count > 1 || c->comm || c->allgather_cb || c->peer_on) ? ms_shard_step(c, &mp->stepper, step_in, mp->tol, &r)
                                                            : ms_step(c, &mp->stepper, step_in, mp->tol, &r);
    c->ahead_allowed = false;
    if (rc) return rc;
    if (trace_steps()) {
      static double t_prev = 0.0;
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      const double t_now = 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
      fprintf(stderr, "[mss] %3d ok %d trials %2d alpha %.3e g.d %+.2e  %7.0f us  rounds %lld multi %lld wasted %lld side %lld ahead %lld adopted %lld dropped %lld\n",
              i, r.success, r.trials, r.alpha, r.g_dot_d, t_prev > 0.0 ? t_now - t_prev : 0.0, (long long)c->q_rounds,
              (long long)c->q_multi, (long long)c->q_wasted, (long long)c->q_side_accepts, (long long)c->q_ahead,
              (long long)c->q_adopted, (long long)c->q_dropped);
      t_prev = t_now;
    }
    out->iterations = i + 1;
    out->energy_eval = r.energy_eval;
    out->grad_norm = r.grad_norm;
    if (step_log) {
      double* row = step_log + 8 * (size_t)i;
      row[0] = r.success;
      row[1] = r.next_step;
      row[2] = r.energy;
      row[3] = r.energy_eval;
      row[4] = r.grad_norm;
      row[5] = r.g_dot_d;
      row[6] = r.alpha;
      row[7] = r.trials;
    }
    if (r.converged) {  // :1324-1337
      out->converged = 1;
      out->step_success = 1;
      out->step_size = step_size;
      drop_ahead(c, /*ran=*/2);
      return MS_OK;
    }
    out->step_success = r.success;
    out->volume_cache_current = 0;  // minimizer.py:1415-1416: project_tilts_to_tangent + increment_version
    out->trials += r.trials;
    out->guard_rejects += r.guard_rejects;
    step_size = r.next_step;
    if (r.success) {
      ++out->accepted;
      out->moved = 1;
    }
    if (mp->fixed_step_mode) step_size = mp->fixed_step;
    out->step_size = step_size;
    if (!r.success) {  // :1425-1476
      if (step_size <= mp->step_size_floor) {
        if (++zero_steps >= mp->max_zero_steps) {
          out->zero_step_exit = 1;
          drop_ahead(c, /*ran=*/2);
          return MS_OK;
        }
      } else {
        zero_steps = 0;
      }
      ms_reset_stepper(c);
    } else {
      zero_steps = 0;
      if (mp->drift_check) {  // :1478-1513
        out->volume_cache_current = 1;  // body.compute_volume: Body's cached volume is current again
        const double denom = std::max(std::fabs(mp->target_volume), 1.0);
        if (std::fabs(r.volume - mp->target_volume) / denom > mp->volume_tolerance) {
          if (mp->project_on_drift) {
            int iters = 0;
            drop_ahead(c, /*ran=*/2);  // (the projection moves x: a round queued for the next step is void)
            rc = ms_project_volume_cached(c, mp->target_volume, 1e-12, 12, 1, &iters, nullptr);
            if (rc) return rc;
            out->volume_cache_current = 0;  // enforce_constraints_after_mesh_ops bumps the mesh version
            // minimizer.py:1505-1507: enforce, then mesh.project_tilts_to_tangent() on the projected surface
            if (c->params.modules & MS_ANY_TILT_MODS) {
              rc = ms_project_tilts_to_tangent(c);
              if (rc) return rc;
            }
          }
          ms_reset_stepper(c);
        }
      }
    }
  }
  drop_ahead(c, /*ran=*/2);
  if (res_energy_valid) {  // (surface energy of the x the resident kernel ended at: the accepted trial's)
    out->energy_current = res_energy;
    out->energy_current_valid = 1;
  } else if (c->carry_valid && !(c->params.modules & MS_ANY_TILT_MODS)) {
    // the mailbox energies describe the positions the loop ended at: the caller's final energy needs no pass
    double e[4];
    energies_from_mailbox(c, e);
    out->energy_current = e[0] + e[1] + e[2] + e[3];
    out->energy_current_valid = 1;
  }
  return MS_OK;
}


// ---- phase-level API -------------------------------------------------------
int ms_phase_energy(ms_ctx* c, int use_direction, double alpha, int write_trial, int guard,
                    int write_bending_factors) {
  if (!c) return MS_ERR_INVALID;
  if ((c->params.modules & MS_ANY_TILT_MODS) && c->shard_count != 1)
    return fail(c, MS_ERR_STATE, "the tilt module is not sharded yet (single GPU only)");
  return phase_energy(c, c->params.modules, use_direction != 0, alpha, write_trial != 0, guard != 0,
                      write_bending_factors != 0);
}

int ms_phase_gradient(ms_ctx* c) {
  if (!c) return MS_ERR_INVALID;
  c->grad_valid = false;  // G receives the raw (unfinalized) gradient
  return phase_gradient(c, c->params.modules, c->buf[MS_BUF_G], false);
}

int ms_phase_direction(ms_ctx* c, int stepper, int use_history) {
  if (!c) return MS_ERR_INVALID;
  c->precond = false;  // (the phase API has no preconditioned lane: never inherit one from an earlier ms_step)
  return phase_direction(c, stepper, use_history != 0);
}

int ms_phase_accept(ms_ctx* c, int keep_history) {
  if (!c) return MS_ERR_INVALID;
  std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);
  c->factors_valid = false;
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  c->sh_maxg2_valid = false;
  if (keep_history) {
    std::swap(c->buf[MS_BUF_G], c->buf[MS_BUF_PG]);
    std::swap(c->buf[MS_BUF_D], c->buf[MS_BUF_PD]);
    c->pd_neg_pg = c->dir_implicit;  // the accepted direction was -G = -PG from now on
    c->last_g = c->buf[MS_BUF_PG];
    c->cg_have_history = true;
    ++c->cg_iter_count;
  }
  c->dir_implicit = false;
  return MS_OK;
}

int ms_phase_commit_trial(ms_ctx* c, double alpha, int keep_history) {
  if (!c) return MS_ERR_INVALID;
  if (c->shard_count > 1) {
    // x <- x + alpha d in place on the rows this rank reads: its own rows and the halo rows
    // of its tiles (d is valid there after the boundary exchange); same expression as the
    // trial pass, so the committed doubles are the evaluated ones
    const Tiling& t = c->til;
    const int64_t rows_per = (int64_t)t.tiles_per_shard * t.own;
    HIPCHK(c, launch_axpy_rows(c->shard_rank * rows_per, (c->shard_rank + 1) * rows_per, c->d_halo_rows,
                               c->n_halo_rows, c->d_vflags, c->buf[MS_BUF_X], trial_dir(c), trial_alpha(c, alpha),
                               c->stream));
    std::swap(c->buf[MS_BUF_X], c->buf[MS_BUF_XT]);  // undone by ms_phase_accept's swap
    return ms_phase_accept(c, keep_history);
  }
  const size_t n3 = 3 * (size_t)c->til.nvp;
  HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_XT], c->buf[MS_BUF_X], n3 * sizeof(double),
                           hipMemcpyDeviceToDevice, S(c)));
  HIPCHK(c, launch_axpy_masked(c->til.nvp, c->d_vflags, c->buf[MS_BUF_XT], trial_dir(c), trial_alpha(c, alpha),
                               c->stream));
  return ms_phase_accept(c, keep_history);
}

int ms_phase_gradient_direction(ms_ctx* c, int stepper, int use_history) {
  if (!c) return MS_ERR_INVALID;
  if (c->params.modules & (MS_CON_VOLUME | MS_TILT_SHAPE_MODS))
    return fail(c, MS_ERR_STATE, "fused gradient+direction needs no constraint row and no tilt module");
  c->grad_valid = false;
  const int dir_mode = (stepper == MS_STEPPER_CG && use_history) ? 2 : 1;
  return phase_gradient(c, c->params.modules, c->buf[MS_BUF_G], false, dir_mode);
}

int ms_phase_set_factors_valid(ms_ctx* c, int valid) {
  if (!c) return MS_ERR_INVALID;
  c->factors_valid = valid != 0;
  return MS_OK;
}

namespace {
constexpr int SH_BUF_FK2 = -1, SH_BUF_FA2 = -2;
constexpr int SH_ALT = 16;  // header slots SH_ALT + s carry slot s of a pair launch's other trial (tilt slots: the
                            // tilt modules are not sharded)
int row_buffers(ms_ctx* c, int n, const int* ids, double* p[4], int ncomp[4], int* comps) {
  if (n < 0 || n > 4 || (n > 0 && !ids)) return fail(c, MS_ERR_INVALID, "boundary exchange: 0..4 buffers");
  *comps = 0;
  for (int k = 0; k < n; ++k) {
    if (ids[k] == SH_BUF_FK2 || ids[k] == SH_BUF_FA2) {  // (internal) a pair launch's second factor set
      if (!c->fK2) return fail(c, MS_ERR_STATE, "boundary exchange: pair buffers not allocated");
      p[k] = ids[k] == SH_BUF_FK2 ? c->fK2 : c->fA2;
      ncomp[k] = ids[k] == SH_BUF_FA2 ? 2 : 3;
      *comps += ncomp[k];
      continue;
    }
    if (ids[k] < 0 || ids[k] > MS_BUF_FA) return fail(c, MS_ERR_INVALID, "boundary exchange: bad buffer id");
    p[k] = c->buf[ids[k]];
    ncomp[k] = ids[k] == MS_BUF_FA ? 2 : 3;
    *comps += ncomp[k];
  }
  return MS_OK;
}
}  // namespace

int ms_boundary_info(ms_ctx* c, int64_t info[4]) {
  if (!c || !info) return MS_ERR_INVALID;
  info[0] = c->bnd_max;
  info[1] = c->bnd_off[(size_t)c->shard_rank + 1] - c->bnd_off[(size_t)c->shard_rank];
  info[2] = c->n_halo_rows;
  info[3] = c->shard_count;
  return MS_OK;
}

size_t ms_exchange_bytes(ms_ctx* c, int n_buffers, const int* buffer_ids) {
  double* p[4];
  int nc[4], comps = 0;
  if (!c || row_buffers(c, n_buffers, buffer_ids, p, nc, &comps)) return 0;
  return sizeof(double) * ((size_t)MS_NSCAL + (size_t)c->bnd_max * comps);
}

int ms_pack_boundary(ms_ctx* c, int n_buffers, const int* buffer_ids, void* send_dev, size_t send_bytes) {
  if (!c || !send_dev) return MS_ERR_INVALID;
  double* p[4];
  int nc[4], comps = 0;
  int rc = row_buffers(c, n_buffers, buffer_ids, p, nc, &comps);
  if (rc) return rc;
  if (send_bytes < ms_exchange_bytes(c, n_buffers, buffer_ids))
    return fail(c, MS_ERR_INVALID, "ms_pack_boundary: send buffer smaller than ms_exchange_bytes");
  const int me = c->shard_rank;
  HIPCHK(c, launch_pack_boundary(c->d_bnd_rows + c->bnd_off[(size_t)me],
                                 c->bnd_off[(size_t)me + 1] - c->bnd_off[(size_t)me], p, nc, n_buffers,
                                 c->d_scal, static_cast<double*>(send_dev), c->stream));
  return MS_OK;
}

int ms_unpack_boundary(ms_ctx* c, int n_buffers, const int* buffer_ids, const void* recv_dev,
                       size_t stride_bytes, double* scal_all_host) {
  if (!c || !recv_dev || !scal_all_host) return MS_ERR_INVALID;
  double* p[4];
  int nc[4], comps = 0;
  int rc = row_buffers(c, n_buffers, buffer_ids, p, nc, &comps);
  if (rc) return rc;
  if (stride_bytes < ms_exchange_bytes(c, n_buffers, buffer_ids) || stride_bytes % sizeof(double))
    return fail(c, MS_ERR_INVALID, "ms_unpack_boundary: bad stride");
  HIPCHK(c, launch_unpack_boundary(c->d_bnd_rows, c->d_bnd_off, c->shard_rank, c->shard_count, c->bnd_max,
                                   p, nc, n_buffers, static_cast<const double*>(recv_dev),
                                   stride_bytes / sizeof(double), c->d_scal_all, c->stream));
  HIPCHK(c, hipMemcpyAsync(scal_all_host, c->d_scal_all, sizeof(double) * MS_NSCAL * (size_t)c->shard_count,
                           hipMemcpyDeviceToHost, S(c)));
  HIPCHK(c, hipStreamSynchronize(S(c)));
  return MS_OK;
}

// ---- library-side sharded driver ------------------------------------------------
namespace {
struct NcclUniqueId {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
};
Rccl g_rccl;

int rccl_bind() {
  if (g_rccl.lib) return MS_OK;
  // the copy the process already uses (torch.distributed's) first, then the ROCm one
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (lib) break;
  }
  for (size_t i = 0; !lib && i < sizeof(names) / sizeof(names[0]); ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(nullptr, MS_ERR_STATE, std::string("librccl not found: ") + dlerror());
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(lib, "ncclCommCount"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather)
    return fail(nullptr, MS_ERR_STATE, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather");
  g_rccl.lib = lib;
  return MS_OK;
}

int shard_buffers(ms_ctx* c) {
  if (c->d_xsend) return MS_OK;
  const size_t n_max = (size_t)MS_NSCAL + 10 * (size_t)c->bnd_max;  // at most 2 x (fK 3 + fA 2) per boundary row
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_xsend), sizeof(double) * n_max));
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_xrecv), sizeof(double) * n_max * (size_t)c->shard_count));
  HIPCHK(c, hipMemset(c->d_xsend, 0, sizeof(double) * n_max));
  HIPCHK(c, hipMemset(c->d_xrecv, 0, sizeof(double) * n_max * (size_t)c->shard_count));
  const size_t sb = sizeof(double) * MS_NSCAL * (size_t)c->shard_count;
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_scal_all), sb, hipHostMallocMapped));
  memset(c->h_scal_all, 0, sb);
  HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_scal_all), c->h_scal_all, 0));
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_xseq), sizeof(unsigned long long) * (size_t)c->shard_count,
                          hipHostMallocMapped));
  for (int r = 0; r < c->shard_count; ++r) c->h_xseq[r] = 0;
  HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_xseq), c->h_xseq, 0));
  return MS_OK;
}

const int SH_SUM[] = {MS_S_ESURF, MS_S_VOL, MS_S_EBEND, MS_S_GGC, MS_S_GCGC, MS_S_GNORM2, MS_S_GDOTD, MS_S_ETILT};
constexpr uint32_t SH_ENERGY = (1u << MS_S_ESURF) | (1u << MS_S_VOL) | (1u << MS_S_EBEND) |
                               (1u << MS_S_MINEDGE2) | (1u << MS_S_GUARD);
constexpr uint32_t SH_GRAD = (1u << MS_S_GGC) | (1u << MS_S_GCGC);
constexpr uint32_t SH_DIR = (1u << MS_S_GNORM2) | (1u << MS_S_GDOTD) | (1u << MS_S_MAXD2) | (1u << MS_S_MAXG2);

// One exchange: boundary rows of `ids` + the MS_NSCAL scalars of every rank; `slots` of the
// rank-ordered fold go to c->sh_scal (push: also to the device scalars).
int shard_exchange(ms_ctx* c, int n, const int* ids, uint32_t slots, bool push, bool fold_alt = false) {
  double* p[4];
  int nc[4], comps = 0;
  int rc = row_buffers(c, n, ids, p, nc, &comps);
  if (rc) return rc;
  rc = shard_buffers(c);
  if (rc) return rc;
  const size_t count = (size_t)MS_NSCAL + (size_t)c->bnd_max * comps;
  const int me = c->shard_rank, W = c->shard_count;
  ++c->xticket;
  if (c->peer_on) {
    // peer-to-peer: this rank's message goes straight into slot `me` of every peer's slab (slab = exchange parity),
    // then its flag word there is raised; the unpack kernel runs behind a bounded wait for every peer's word here
    ++c->peer_ticket;
    const int par = (int)(c->peer_ticket & 1);
    if (count > c->peer_stride) return fail(c, MS_ERR_STATE, "peer exchange: message longer than the slab slot");
    double* dst[16];
    unsigned long long* flg[16];
    for (int r = 0; r < W; ++r) {
      dst[r] = c->peer_slabs[(size_t)r] + ((size_t)par * W + (size_t)me) * c->peer_stride;
      flg[r] = c->peer_flags[(size_t)r] + (size_t)par * 16;
    }
    const bool stream_ops = c->peer_stream_ops && !c->peer_barrier;
    // (the pack kernel's last block per peer raises the flag word itself; with stream memory operations the words are
    // written behind the kernel instead)
    const bool fused_flags = !stream_ops && c->d_peer_flagtab != nullptr;
    HIPCHK(c, launch_pack_peers(c->d_bnd_rows + c->bnd_off[(size_t)me],
                                c->bnd_off[(size_t)me + 1] - c->bnd_off[(size_t)me], p, nc, n, c->d_scal, dst, W,
                                c->stream, fused_flags ? c->d_peer_flagtab + par : nullptr, c->d_peer_arrived, me,
                                c->peer_ticket));
    if (stream_ops) {
      // (the pack kernel's stores are system-scope write-through and complete before the kernel does: the value
      // written behind it in stream order is the release)
      for (int r = 0; r < W; ++r) HIPCHK(c, hipStreamWriteValue64(S(c), flg[r] + me, c->peer_ticket, 0));
    } else if (!fused_flags) {
      HIPCHK(c, launch_flag_peers(flg, me, W, c->peer_ticket, c->stream));
    }
    const unsigned long long* wait_flags = c->d_peer_flag + (size_t)par * 16;
    if (stream_ops) {
      for (int r = 0; r < W; ++r)
        HIPCHK(c, hipStreamWaitValue64(S(c), const_cast<unsigned long long*>(wait_flags) + r, c->peer_ticket,
                                       hipStreamWaitValueGte, ~0ull));
      wait_flags = nullptr;  // (the unpack kernel starts when every word has arrived)
    }
    if (c->peer_barrier) {  // (contexts of one process wait on the host: see ms_shard_peer_set_barrier)
      HIPCHK(c, hipStreamSynchronize(S(c)));
      if (c->peer_barrier(c->peer_barrier_user) != 0) return fail(c, MS_ERR_STATE, "peer exchange: the caller's barrier failed");
      wait_flags = nullptr;
    }
    // every block row of the unpack kernel waits (bounded) for the flag of the rank whose message it unpacks
    HIPCHK(c, launch_unpack_boundary(c->d_bnd_rows, c->d_bnd_off, me, W, c->bnd_max, p, nc, n,
                                     c->d_peer_slab + (size_t)par * W * c->peer_stride, c->peer_stride,
                                     c->d_h_scal_all, c->stream, c->d_h_xseq, c->xticket, /*remote_written=*/true,
                                     wait_flags, c->peer_ticket, c->d_h_err + 1));  // (a word of its own: h_err[0] is the queue's)
  } else {
  HIPCHK(c, launch_pack_boundary(c->d_bnd_rows + c->bnd_off[(size_t)me],
                                 c->bnd_off[(size_t)me + 1] - c->bnd_off[(size_t)me], p, nc, n, c->d_scal,
                                 c->d_xsend, c->stream));
  if (c->allgather_cb) {
    HIPCHK(c, hipStreamSynchronize(S(c)));
    if (c->allgather_cb(c->allgather_user, c->d_xsend, c->d_xrecv, count * sizeof(double)) != 0)
      return fail(c, MS_ERR_STATE, "caller-supplied all-gather failed");
  } else if (c->comm) {
    const int r = g_rccl.AllGather(c->d_xsend, c->d_xrecv, count, /*ncclDouble*/ 8, c->comm, S(c));
    if (r != 0)
      return fail(c, MS_ERR_HIP, std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
  } else if (W == 1) {
    HIPCHK(c, hipMemcpyAsync(c->d_xrecv, c->d_xsend, count * sizeof(double), hipMemcpyDeviceToDevice, S(c)));
  } else {
    return fail(c, MS_ERR_STATE, "ms_shard_step: no communicator (ms_shard_comm_init / ms_shard_set_allgather / ms_shard_peer_*)");
  }
  // the unpack kernel posts one sequence word per rank as soon as that rank's scalar header is in the mailbox
  HIPCHK(c, launch_unpack_boundary(c->d_bnd_rows, c->d_bnd_off, me, W, c->bnd_max, p, nc, n, c->d_xrecv, count,
                                   c->d_h_scal_all, c->stream, c->d_h_xseq, c->xticket));
  }
  bool seen = false;
  const bool watchdog = c->peer_on && c->peer_stream_ops && !c->peer_barrier;
  struct timespec w0;
  if (watchdog) clock_gettime(CLOCK_MONOTONIC, &w0);
  for (long spin = 0; spin < 20000000L; ++spin) {
    seen = true;
    for (int r = 0; r < W && seen; ++r) seen = __atomic_load_n(c->h_xseq + r, __ATOMIC_ACQUIRE) >= c->xticket;
    if (seen) break;
    __builtin_ia32_pause();
    if (watchdog && (spin & 0xfff) == 0xfff) {
      struct timespec w1;
      clock_gettime(CLOCK_MONOTONIC, &w1);
      if ((double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec) > 2.0) {
        // a peer's word has not arrived: the stream sits in an unbounded wait.  Release it (this rank's own words,
        // through another stream), let the queue drain, report.
        const int par = (int)(c->peer_ticket & 1);
        if (!c->peer_aux) HIPCHK(c, hipStreamCreateWithFlags(&c->peer_aux, hipStreamNonBlocking));
        for (int r = 0; r < W; ++r)
          HIPCHK(c, hipStreamWriteValue64(c->peer_aux, c->d_peer_flag + (size_t)par * 16 + r, c->peer_ticket, 0));
        HIPCHK(c, hipStreamSynchronize(c->peer_aux));
        HIPCHK(c, hipStreamSynchronize(S(c)));
        return fail(c, MS_ERR_STATE, "peer exchange: a peer's flag word did not arrive within 2 s (stream wait released by the host)");
      }
    }
  }
  if (!seen) HIPCHK(c, hipStreamSynchronize(S(c)));
  if (c->peer_on && c->h_err && (__atomic_load_n(c->h_err + 1, __ATOMIC_ACQUIRE) >> 62) == 1) {
    const unsigned long long e = __atomic_exchange_n(c->h_err + 1, 0ull, __ATOMIC_ACQ_REL);  // reported once
    return fail(c, MS_ERR_STATE, "peer exchange: a peer's flag did not arrive within the bounded wait (rank " +
                                     std::to_string((int)((e >> 32) & 0xff)) + ")");
  }
  // fold in rank order: every rank adds the same doubles in the same order
  for (int sl : SH_SUM)
    if (slots & (1u << sl)) {
      double acc = 0.0;
      for (int r = 0; r < W; ++r) acc += c->h_scal_all[(size_t)r * MS_NSCAL + sl];
      c->sh_scal[sl] = acc;
    }
  if (slots & (1u << MS_S_MINEDGE2)) {
    double m = c->h_scal_all[MS_S_MINEDGE2];
    for (int r = 1; r < W; ++r) m = std::min(m, c->h_scal_all[(size_t)r * MS_NSCAL + MS_S_MINEDGE2]);
    c->sh_scal[MS_S_MINEDGE2] = m;
  }
  for (int sl : {(int)MS_S_GUARD, (int)MS_S_MAXD2, (int)MS_S_MAXG2})
    if (slots & (1u << sl)) {
      double m = c->h_scal_all[sl];
      for (int r = 1; r < W; ++r) m = std::max(m, c->h_scal_all[(size_t)r * MS_NSCAL + sl]);
      c->sh_scal[sl] = m;
    }
  if (fold_alt) {  // the pair launch's other trial: same folds, same rank order, from the SH_ALT header slots
    for (int sl : {(int)MS_S_ESURF, (int)MS_S_VOL, (int)MS_S_EBEND}) {
      double acc = 0.0;
      for (int r = 0; r < W; ++r) acc += c->h_scal_all[(size_t)r * MS_NSCAL + SH_ALT + sl];
      c->sh_scal2[sl] = acc;
    }
    double m = c->h_scal_all[SH_ALT + MS_S_MINEDGE2];
    for (int r = 1; r < W; ++r) m = std::min(m, c->h_scal_all[(size_t)r * MS_NSCAL + SH_ALT + MS_S_MINEDGE2]);
    c->sh_scal2[MS_S_MINEDGE2] = m;
  }
  if (push)
    HIPCHK(c, hipMemcpyAsync(c->d_scal, c->sh_scal, sizeof(double) * MS_NSCAL, hipMemcpyHostToDevice, S(c)));
  ++c->sh_exchanges;
  return MS_OK;
}

double shard_energy_of(const ms_ctx* c, const double* scal) {
  const uint32_t m = c->params.modules;
  double e = 0.0;
  if (m & MS_MOD_SURFACE) e += scal[MS_S_ESURF];
  if (m & MS_MOD_BENDING) e += scal[MS_S_EBEND];
  e += penalty_energy(c, scal[MS_S_VOL]);
  return e;
}
double shard_energy(const ms_ctx* c) { return shard_energy_of(c, c->sh_scal); }
}  // namespace

static void shard_comm_destroy(void* comm) {
  if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
}

int ms_shard_unique_id(void* id128) {
  if (!id128) return MS_ERR_INVALID;
  int rc = rccl_bind();
  if (rc) return rc;
  NcclUniqueId id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r != 0) return fail(nullptr, MS_ERR_HIP, "ncclGetUniqueId failed");
  memcpy(id128, id.internal, 128);
  return MS_OK;
}

int ms_shard_comm_init(ms_ctx* c, const void* id128) {
  if (!c || !id128) return fail(c, MS_ERR_INVALID, "ms_shard_comm_init: NULL argument");
  int rc = rccl_bind();
  if (rc) return fail(c, rc, g_last_error);
  (void)hipSetDevice(c->device);
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  const int r = g_rccl.CommInitRank(&c->comm, c->shard_count, id, c->shard_rank);
  if (r != 0) {
    c->comm = nullptr;
    return fail(c, MS_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
  }
  return shard_buffers(c);
}

int ms_shard_set_allgather(ms_ctx* c, ms_allgather_fn fn, void* user) {
  if (!c) return MS_ERR_INVALID;
  c->allgather_cb = fn;
  c->allgather_user = user;
  return shard_buffers(c);
}

namespace {
int peer_alloc(ms_ctx* c) {
  if (c->d_peer_slab) return MS_OK;
  if (c->shard_count > 16) return fail(c, MS_ERR_INVALID, "peer exchange: at most 16 ranks");
  int rc = shard_buffers(c);
  if (rc) return rc;
  c->peer_stride = (size_t)MS_NSCAL + 10 * (size_t)c->bnd_max;
  const size_t sb = sizeof(double) * 2 * (size_t)c->shard_count * c->peer_stride;
  // The slabs and flag words are written by OTHER GPUs over xGMI while a kernel of this one is resident and polls them.
  // Ordinary hipMalloc memory is coarse-grained (cached read-write in this GPU's L2, coherent across agents only at
  // kernel boundaries): a resident wave could keep reading a stale flag or a boundary row of exchange k-2.  Uncached
  // (MTYPE_UC) device memory is what the collective library's own IPC signal buffers use; fine-grained is the second
  // choice; plain hipMalloc stays as the last resort (MS_PEER_MEM=uncached|finegrained|default forces one).
  const char* want = getenv("MS_PEER_MEM");
  const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
  const char* names[3] = {"uncached", "finegrained", "default"};
  c->peer_mem_kind = -1;
  for (int k = 0; k < 3 && c->peer_mem_kind < 0; ++k) {
    if (want && *want && strcmp(want, names[k]) != 0) continue;
    void *ps = nullptr, *pf = nullptr;
    hipError_t e1 = kinds[k] == hipDeviceMallocDefault ? hipMalloc(&ps, sb) : hipExtMallocWithFlags(&ps, sb, kinds[k]);
    hipError_t e2 = e1 != hipSuccess ? e1
                    : (kinds[k] == hipDeviceMallocDefault ? hipMalloc(&pf, sizeof(unsigned long long) * 32)
                                                          : hipExtMallocWithFlags(&pf, sizeof(unsigned long long) * 32, kinds[k]));
    hipIpcMemHandle_t probe;
    // (memory that cannot be exported is of no use here: ms_shard_peer_export would fail later)
    if (e1 == hipSuccess && e2 == hipSuccess && hipIpcGetMemHandle(&probe, ps) == hipSuccess &&
        hipIpcGetMemHandle(&probe, pf) == hipSuccess) {
      c->d_peer_slab = static_cast<double*>(ps);
      c->d_peer_flag = static_cast<unsigned long long*>(pf);
      c->peer_mem_kind = k;
    } else {
      (void)hipGetLastError();
      if (ps) (void)hipFree(ps);
      if (pf) (void)hipFree(pf);
    }
  }
  if (c->peer_mem_kind < 0) return fail(c, MS_ERR_HIP, "peer exchange: no exportable device memory for the slabs / flag words");
  HIPCHK(c, hipMemset(c->d_peer_slab, 0, sb));
  HIPCHK(c, hipMemset(c->d_peer_flag, 0, sizeof(unsigned long long) * 32));
  HIPCHK(c, hipDeviceSynchronize());
  return MS_OK;
}
}  // namespace

namespace {
// the peers' flag rows as the pack kernel reads them: one table per exchange parity
int peer_flag_table(ms_ctx* c) {
  const int W = c->shard_count;
  PeerFlags tab[2];
  memset(tab, 0, sizeof(tab));
  for (int par = 0; par < 2; ++par)
    for (int r = 0; r < W; ++r) tab[par].p[r] = c->peer_flags[(size_t)r] + (size_t)par * 16;
  if (!c->d_peer_flagtab) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_peer_flagtab), sizeof(tab)));
  HIPCHK(c, hipMemcpy(c->d_peer_flagtab, tab, sizeof(tab), hipMemcpyHostToDevice));
  if (!c->d_peer_arrived) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_peer_arrived), sizeof(unsigned int) * 16));
    HIPCHK(c, hipMemset(c->d_peer_arrived, 0, sizeof(unsigned int) * 16));
  }
  return MS_OK;
}
}  // namespace

int ms_shard_peer_local(ms_ctx* c, void** recv_slab, void** flag_words) {
  if (!c || !recv_slab || !flag_words) return MS_ERR_INVALID;
  int rc = peer_alloc(c);
  if (rc) return rc;
  *recv_slab = c->d_peer_slab;
  *flag_words = c->d_peer_flag;
  return MS_OK;
}

int ms_shard_peer_export(ms_ctx* c, void* handles128) {
  if (!c || !handles128) return MS_ERR_INVALID;
  int rc = peer_alloc(c);
  if (rc) return rc;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h[2];
  HIPCHK(c, hipIpcGetMemHandle(&h[0], c->d_peer_slab));
  HIPCHK(c, hipIpcGetMemHandle(&h[1], c->d_peer_flag));
  memcpy(handles128, h, 128);
  return MS_OK;
}

int ms_shard_peer_open(ms_ctx* c, const void* handles_all) {
  if (!c || !handles_all) return MS_ERR_INVALID;
  int rc = peer_alloc(c);
  if (rc) return rc;
  const int W = c->shard_count;
  c->peer_slabs.assign((size_t)W, nullptr);
  c->peer_flags.assign((size_t)W, nullptr);
  for (int r = 0; r < W; ++r) {
    if (r == c->shard_rank) {
      c->peer_slabs[(size_t)r] = c->d_peer_slab;
      c->peer_flags[(size_t)r] = c->d_peer_flag;
      continue;
    }
    hipIpcMemHandle_t h[2];
    memcpy(h, static_cast<const char*>(handles_all) + 128 * (size_t)r, 128);
    void *ps = nullptr, *pf = nullptr;
    HIPCHK(c, hipIpcOpenMemHandle(&ps, h[0], hipIpcMemLazyEnablePeerAccess));
    c->peer_opened.push_back(ps);
    HIPCHK(c, hipIpcOpenMemHandle(&pf, h[1], hipIpcMemLazyEnablePeerAccess));
    c->peer_opened.push_back(pf);
    c->peer_slabs[(size_t)r] = static_cast<double*>(ps);
    c->peer_flags[(size_t)r] = static_cast<unsigned long long*>(pf);
  }
  rc = peer_flag_table(c);
  if (rc) return rc;
  c->peer_on = true;
  if (const char* e = getenv("MS_PEER_WAIT")) {
    if (strcmp(e, "stream") == 0) {
      int can = 0;
      HIPCHK(c, hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device));
      if (!can) return fail(c, MS_ERR_STATE, "MS_PEER_WAIT=stream: the device has no stream wait-value operations");
      c->peer_stream_ops = true;
    }
  }
  return MS_OK;
}

int ms_shard_peer_set_pointers(ms_ctx* c, void* const* recv_slabs, void* const* flag_words) {
  if (!c || !recv_slabs || !flag_words) return MS_ERR_INVALID;
  int rc = peer_alloc(c);
  if (rc) return rc;
  const int W = c->shard_count;
  c->peer_slabs.assign((size_t)W, nullptr);
  c->peer_flags.assign((size_t)W, nullptr);
  for (int r = 0; r < W; ++r) {
    c->peer_slabs[(size_t)r] = static_cast<double*>(recv_slabs[r]);
    c->peer_flags[(size_t)r] = static_cast<unsigned long long*>(flag_words[r]);
  }
  if (c->peer_slabs[(size_t)c->shard_rank] != c->d_peer_slab)
    return fail(c, MS_ERR_INVALID, "ms_shard_peer_set_pointers: the own entry must be ms_shard_peer_local's");
  rc = peer_flag_table(c);
  if (rc) return rc;
  c->peer_on = true;
  return MS_OK;
}

int ms_shard_peer_set_barrier(ms_ctx* c, ms_barrier_fn fn, void* user) {
  if (!c) return MS_ERR_INVALID;
  c->peer_barrier = fn;
  c->peer_barrier_user = user;
  return MS_OK;
}

int64_t ms_shard_exchange_count(const ms_ctx* c) { return c ? (int64_t)c->sh_exchanges : 0; }

int ms_shard_peer_memory_kind(const ms_ctx* c) { return (c && c->d_peer_slab) ? c->peer_mem_kind : -1; }

int ms_shard_comm_ranks(ms_ctx* c) {
  if (!c || !c->comm || !g_rccl.CommCount) return 0;
  int n = 0;
  return g_rccl.CommCount(c->comm, &n) == 0 ? n : 0;
}

// The control flow of parallel.ShardedStepper.step (itself a restatement of ms_step).
int ms_shard_step(ms_ctx* c, const ms_stepper_params* sp, double step_size, double tol, ms_step_result* out) {
  if (!c || !sp || !out) return fail(c, MS_ERR_INVALID, "ms_shard_step: NULL argument");
  const uint32_t mods = c->params.modules;
  if (mods & MS_ANY_TILT_MODS) return fail(c, MS_ERR_STATE, "the tilt modules are not sharded yet (single GPU only)");
  if (sp->precondition) return fail(c, MS_ERR_STATE, "ConjugateGradient(precondition=True) is not sharded (single GPU only)");
  // (line_search.py:428-487: every trial projected onto the target volume -- the projection is not sharded; running the
  // plain lane instead would be a different trajectory, silently)
  if (sp->enforce_volume)
    return fail(c, MS_ERR_STATE, "volume_projection_during_minimization (the enforcer lane of the line search) is not sharded (single GPU only)");
  c->precond = false;
  memset(out, 0, sizeof(*out));
  const bool cg = sp->stepper == MS_STEPPER_CG;
  const int restart = sp->restart_interval > 0 ? sp->restart_interval : 10;
  const bool use_history = cg && c->cg_have_history && (c->cg_iter_count % restart != 0);
  const bool bend = (mods & MS_MOD_BENDING) != 0;
  const bool constraint = (mods & MS_CON_VOLUME) != 0;
  const bool penalty = (mods & MS_MOD_VOLUME_PENALTY) != 0;
  const bool carry_mode = sp->reuse_energy0 >= 2;
  const int fbufs[2] = {MS_BUF_FK, MS_BUF_FA};
  const int n_fb = bend ? 2 : 0;
  const int dbuf[2] = {MS_BUF_D, MS_BUF_G};
  const bool carried = carry_mode && c->sh_carry_valid;
  int rc;
  bool implicit_restart = false, fused = false;
  if (!carried) {
    rc = phase_energy(c, mods, false, 0.0, false, false, true);
    if (rc) return rc;
    rc = shard_exchange(c, n_fb, fbufs, SH_ENERGY, penalty);
    if (rc) return rc;
    c->sh_grad_valid = false;
  }
  if (carried && c->sh_grad_valid && !constraint && !use_history && c->sh_maxg2_valid) {
    // steepest-descent restart on an unchanged gradient: d = -g.  The last direction exchange carried the
    // finalized gradient rows as well, so G is valid on every row this rank reads and the scalars follow from
    // the ones already folded -- no kernel and, above all, no exchange
    c->dir_implicit = true;
    c->sh_scal[MS_S_GDOTD] = -c->sh_scal[MS_S_GNORM2];
    c->sh_scal[MS_S_MAXD2] = c->sh_scal[MS_S_MAXG2];
    implicit_restart = true;
    rc = MS_OK;
  } else if (carried && c->sh_grad_valid && !constraint) {
    rc = phase_direction(c, sp->stepper, use_history, /*g_finalized=*/true);
  } else if (constraint) {
    rc = phase_gradient(c, mods, c->buf[MS_BUF_G], false);
    if (rc) return rc;
    rc = shard_exchange(c, 0, nullptr, SH_GRAD, true);
    if (rc) return rc;
    rc = phase_direction(c, sp->stepper, use_history);
  } else {
    const int dir_mode = (cg && use_history) ? 2 : 1;
    rc = phase_gradient(c, mods, c->buf[MS_BUF_G], false, dir_mode);
    fused = true;  // (its epilogue also reduced max|g_i|^2)
  }
  if (rc) return rc;
  if (!implicit_restart) {
    // without a constraint row the fused pass has just finalized G: send its boundary rows along with D's, so a
    // steepest-descent restart after a failed search needs no exchange of its own
    const bool with_g = !constraint;
    rc = shard_exchange(c, with_g ? 2 : 1, dbuf, SH_DIR, false);
    if (rc) return rc;
    c->sh_maxg2_valid = with_g && fused;
  }
  c->sh_carry_valid = carry_mode;
  c->sh_grad_valid = carry_mode && !constraint;
  const double E_eval = shard_energy(c);
  const double grad_norm = std::sqrt(c->sh_scal[MS_S_GNORM2]);
  const double g_dot_d = c->sh_scal[MS_S_GDOTD];
  const double max_dir = std::sqrt(c->sh_scal[MS_S_MAXD2]);
  out->energy_eval = E_eval;
  out->grad_norm = grad_norm;
  out->g_dot_d = g_dot_d;
  out->volume = c->sh_scal[MS_S_VOL];
  out->next_step = step_size;
  out->energy = E_eval;
  if (grad_norm < tol) {
    out->converged = out->success = 1;
    return MS_OK;
  }
  double energy0 = E_eval;
  if (sp->reuse_energy0 == 0) {
    rc = phase_energy(c, mods, false, 0.0, false, false, false);
    if (rc) return rc;
    rc = shard_exchange(c, 0, nullptr, SH_ENERGY, false);
    if (rc) return rc;
    energy0 = shard_energy(c);
  }
  const double min_edge = c->til.nf > 0 ? std::sqrt(c->sh_scal[MS_S_MINEDGE2]) : 0.0;
  out->energy = energy0;
  const double safe_limit = min_edge > 0.0 ? 0.3 * min_edge : INFINITY;
  if (g_dot_d >= 0.0) return MS_OK;
  double alpha = step_size;
  if (sp->edge_fraction > 0.0 && min_edge > 0.0 && max_dir > 0.0)
    alpha = std::min(alpha, sp->edge_fraction * min_edge / max_dir);
  const double alpha_max = sp->alpha_max_factor * step_size;
  const int max_iter = sp->max_iter > 0 ? sp->max_iter : 10;
  // line-search history (same bookkeeping as ms_step: prediction only)
  double min_rejected = INFINITY, a_hi = 0.0, r_lo = INFINITY;
  ms_ctx::LsHist& lh = c->ls[use_history ? 1 : 0];
  for (int k = 0; k < std::min(lh.n, (int)ms_ctx::LS_HIST); ++k) {
    a_hi = std::max(a_hi, lh.acc[k]);
    r_lo = std::min(r_lo, lh.rej[k]);
  }
  auto remember = [&](double alpha_acc) {
    lh.acc[lh.n % ms_ctx::LS_HIST] = alpha_acc;
    lh.rej[lh.n % ms_ctx::LS_HIST] = min_rejected;
    ++lh.n;
  };
  auto accepted = [&](double alpha_acc, double E_t) -> int {
    int r2 = ms_phase_commit_trial(c, alpha_acc, cg ? 1 : 0);
    if (r2) return r2;
    c->sh_grad_valid = false;
    if (carry_mode) {
      c->factors_valid = true;
      if (penalty)
        HIPCHK(c, hipMemcpyAsync(c->d_scal, c->sh_scal, sizeof(double) * MS_NSCAL, hipMemcpyHostToDevice,
                                 S(c)));
      c->sh_carry_valid = true;
    }
    out->success = 1;
    out->alpha = alpha_acc;
    out->energy = E_t;
    out->volume = c->sh_scal[MS_S_VOL];
    out->next_step = std::min(alpha_acc * sp->gamma, alpha_max);
    remember(alpha_acc);
    return MS_OK;
  };
  int it0 = 0;
  // pair launch: trial 0 is expected to fail -> trials 0 and 1 in one energy launch and ONE
  // exchange (both trials' scalars in the header, both factor sets' boundary rows behind it)
  const bool pair = c->pair_enable && carry_mode && bend && !penalty && max_iter >= 2 &&
                    alpha * max_dir < safe_limit && alpha * sp->beta >= 1e-8 &&
                    (c->pair_force || (lh.n >= 2 && alpha > 1.05 * a_hi && r_lo < INFINITY));
  if (pair) {
    rc = spec_prepare(c);
    if (rc) return rc;
    const double alpha0 = alpha, alpha1 = alpha * sp->beta;
    c->pair_on = 2;
    c->pair_alpha[0] = alpha0;
    c->pair_scal2 = c->d_scal + SH_ALT;
    rc = phase_energy(c, mods, true, alpha1, false, false, true);
    c->pair_on = 0;
    c->pair_scal2 = nullptr;
    if (rc) return rc;
    c->sh_carry_valid = false;
    const int pbufs[4] = {MS_BUF_FK, MS_BUF_FA, SH_BUF_FK2, SH_BUF_FA2};
    rc = shard_exchange(c, 4, pbufs, SH_ENERGY, false, /*fold_alt=*/true);
    if (rc) return rc;
    ++out->trials;
    const double E0 = shard_energy_of(c, c->sh_scal2);
    if (E0 <= energy0 + sp->c * alpha0 * g_dot_d) {
      // the unexpected case: trial 0's factors (boundary rows included) are in the second set
      const size_t nvp = (size_t)c->til.nvp;
      HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_FK], c->fK2, sizeof(double) * 3 * nvp, hipMemcpyDeviceToDevice, S(c)));
      HIPCHK(c, hipMemcpyAsync(c->buf[MS_BUF_FA], c->fA2, sizeof(double) * 2 * nvp, hipMemcpyDeviceToDevice, S(c)));
      for (int sl : {(int)MS_S_ESURF, (int)MS_S_VOL, (int)MS_S_EBEND, (int)MS_S_MINEDGE2}) c->sh_scal[sl] = c->sh_scal2[sl];
      return accepted(alpha0, E0);
    }
    min_rejected = alpha0;
    ++out->trials;
    const double E1 = shard_energy(c);
    if (E1 <= energy0 + sp->c * alpha1 * g_dot_d) return accepted(alpha1, E1);
    min_rejected = alpha1;
    alpha = alpha1 * sp->beta;
    it0 = 2;
    if (alpha < 1e-8) it0 = max_iter;
  }
  for (int it = it0; it < max_iter; ++it) {
    const bool safe_small = alpha * max_dir < safe_limit;
    rc = phase_energy(c, mods, true, alpha, false, !safe_small, carry_mode);
    if (rc) return rc;
    if (carry_mode) c->sh_carry_valid = false;  // the factor buffers now belong to the trial point
    rc = shard_exchange(c, carry_mode ? n_fb : 0, fbufs, SH_ENERGY, false);
    if (rc) return rc;
    if (!safe_small && c->sh_scal[MS_S_GUARD] > 0.0) {
      ++out->guard_rejects;
      min_rejected = alpha;
      alpha *= sp->beta;
      if (alpha < 1e-8) break;
      continue;
    }
    ++out->trials;
    const double E_t = shard_energy(c);
    if (E_t <= energy0 + sp->c * alpha * g_dot_d) return accepted(alpha, E_t);
    min_rejected = alpha;
    alpha *= sp->beta;
    if (alpha < 1e-8) break;
  }
  out->next_step = std::max(std::max(alpha * sp->beta, 0.0), step_size * sp->beta);
  return MS_OK;
}

size_t ms_state_bytes(const ms_ctx* c) {
  if (!c) return 0;
  return sizeof(double) * (8 * 3 * (size_t)c->til.nvp + 2 * (size_t)c->til.nvp);
}

int ms_rebind_state(ms_ctx* c, void* device_base, size_t bytes) {
  if (!c || !device_base) return fail(c, MS_ERR_INVALID, "ms_rebind_state: NULL argument");
  const size_t need = ms_state_bytes(c);
  if (bytes < need) return fail(c, MS_ERR_INVALID, "ms_rebind_state: buffer smaller than ms_state_bytes");
  HIPCHK(c, hipStreamSynchronize(S(c)));
  HIPCHK(c, hipMemcpy(device_base, c->state, need, hipMemcpyDeviceToDevice));
  double* nb = static_cast<double*>(device_base);
  for (int b = 0; b <= MS_BUF_FA; ++b) c->buf[b] = nb + (c->buf[b] - c->state);
  c->last_g = nb + (c->last_g - c->state);
  if (c->own_state) HIPCHK(c, hipFree(c->state));
  c->state = nb;
  c->own_state = false;
  return MS_OK;
}

int ms_fetch_scalars(ms_ctx* c, double* out) {
  if (!c || !out) return MS_ERR_INVALID;
  int rc = fetch(c);
  if (rc) return rc;
  memcpy(out, c->h_scal, sizeof(double) * MS_NSCAL);
  return MS_OK;
}

int ms_store_scalars(ms_ctx* c, const double* in) {
  if (!c || !in) return MS_ERR_INVALID;
  for (int sl = 0; sl < MS_NSCAL; ++sl) put_mailbox(c, sl, in[sl]);
  c->carry_valid = c->grad_valid = c->bt_valid = c->maxg2_valid = false;
  c->sh_carry_valid = c->sh_grad_valid = false;
  HIPCHK(c, hipMemcpyAsync(c->d_scal, c->h_scal, sizeof(double) * MS_NSCAL, hipMemcpyHostToDevice,
                           S(c)));
  HIPCHK(c, hipStreamSynchronize(S(c)));
  return MS_OK;
}

int ms_device_buffer(ms_ctx* c, int buffer, void** dev_ptr, size_t* bytes) {
  if (c) (void)exec_flush(c);  // (the caller may read the buffer on its own stream: nothing stays recorded)
  if (!c || !dev_ptr || buffer < 0 || buffer >= MS_BUF_COUNT) return MS_ERR_INVALID;
  *dev_ptr = c->buf[buffer];
  if (bytes) {
    if (buffer == MS_BUF_SCAL)
      *bytes = sizeof(double) * MS_NSCAL;
    else
      *bytes = sizeof(double) * (size_t)c->til.nvp * (buffer == MS_BUF_FA ? 2 : 3);
  }
  return MS_OK;
}

int ms_shard_info(ms_ctx* c, int64_t* nvp, int64_t* row0, int64_t* row1, int64_t* rows_per_shard) {
  if (!c) return MS_ERR_INVALID;
  const Tiling& t = c->til;
  if (nvp) *nvp = t.nvp;
  if (rows_per_shard) *rows_per_shard = (int64_t)t.tiles_per_shard * t.own;
  if (row0) *row0 = (int64_t)c->shard_rank * t.tiles_per_shard * t.own;
  if (row1) *row1 = (int64_t)(c->shard_rank + 1) * t.tiles_per_shard * t.own;
  return MS_OK;
}

int ms_tile_stats(ms_ctx* c, int64_t* n_tiles, int64_t* facet_instances, int64_t* max_halo,
                  int64_t* lds_bytes_energy, int64_t* lds_bytes_gradient) {
  if (!c) return MS_ERR_INVALID;
  const Tiling& t = c->til;
  const bool bend = (c->params.modules & MS_MOD_BENDING) != 0;
  if (n_tiles) *n_tiles = t.n_tiles;
  if (facet_instances) *facet_instances = (int64_t)t.tile_facets.size();
  if (max_halo) *max_halo = t.max_halo;
  if (lds_bytes_energy) *lds_bytes_energy = (int64_t)energy_lds_bytes(t.T, c->cap, t.max_ent, bend, false, c->has_boundary, !c->deterministic);
  if (lds_bytes_gradient)
    *lds_bytes_gradient = (int64_t)gradient_lds_bytes(t.T, c->cap, t.max_ent, bend,
                                                      (c->params.modules & MS_CON_VOLUME) != 0, !c->deterministic);
  return MS_OK;
}

int ms_queue_stats(ms_ctx* c, int64_t stats[8]) {
  if (!c || !stats) return MS_ERR_INVALID;
  for (int k = 0; k < 8; ++k) stats[k] = 0;
  stats[0] = c->q_rounds;
  stats[1] = c->q_multi;
  stats[2] = c->q_wasted;
  stats[3] = c->q_side_accepts;
  stats[4] = c->queue_mismatches;
  stats[5] = c->q_ahead;
  stats[6] = c->q_adopted;
  stats[7] = c->q_dropped;
  return MS_OK;
}

int ms_resident_stats(ms_ctx* c, int64_t stats[4]) {
  if (!c || !stats) return MS_ERR_INVALID;
  stats[0] = c->resident_ok;
  stats[1] = c->resident_launches;
  stats[2] = c->resident_steps;
  stats[3] = c->resident_bails;
  return MS_OK;
}

int ms_exec_stats(ms_ctx* c, int64_t stats[4]) {
  if (!c || !stats) return MS_ERR_INVALID;
  stats[0] = c->exec_on ? 1 : 0;
  stats[1] = c->exec.launches;
  stats[2] = c->exec.cmds;
  stats[3] = (c->exec_wanted ? 1 : 0) | ((int64_t)c->relax_programs << 8);
  return MS_OK;
}

// Diagnostic: how long does each record of the one-workgroup interpreter take?  on != 0 arms a device buffer that
// k_exec appends {kind, mode, instance, duration} to; reading returns, per (kind, mode) pair seen, the count and the
// total duration in microseconds -- rows of {kind, mode | inst << 16, count, total_us} -- and clears the buffer.
int ms_exec_trace(ms_ctx* c, int on, double* rows, int max_rows, int* n_rows) {
  if (!c) return MS_ERR_INVALID;
  if (n_rows) *n_rows = 0;
  int rc = exec_flush(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->exec.d_stamps && rows && max_rows > 0 && n_rows) {
    std::vector<unsigned long long> h((size_t)2 * EXEC_STAMP_CAP);
    HIPCHK(c, hipMemcpy(h.data(), c->exec.d_stamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    const size_t n = (size_t)std::min<unsigned long long>(h[0], (unsigned long long)EXEC_STAMP_CAP - 1);
    std::map<unsigned long long, std::pair<long, double>> acc;
    for (size_t i = 0; i < n; ++i) {
      auto& e = acc[h[1 + 2 * i]];
      e.first += 1;
      e.second += 0.01 * (double)h[2 + 2 * i];  // s_memrealtime: 100 MHz
    }
    int k = 0;
    for (auto& kv : acc) {
      if (k >= max_rows) break;
      rows[4 * k] = (double)(kv.first >> 32);
      rows[4 * k + 1] = (double)(kv.first & 0xffffffffull);
      rows[4 * k + 2] = (double)kv.second.first;
      rows[4 * k + 3] = kv.second.second;
      ++k;
    }
    *n_rows = k;
  }
  if (on && !c->exec.d_stamps)
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->exec.d_stamps), sizeof(unsigned long long) * 2 * EXEC_STAMP_CAP));
  if (c->exec.d_stamps) HIPCHK(c, hipMemset(c->exec.d_stamps, 0, sizeof(unsigned long long) * 2 * EXEC_STAMP_CAP));
  if (!on && c->exec.d_stamps) {
    (void)hipFree(c->exec.d_stamps);
    c->exec.d_stamps = nullptr;
  }
  return MS_OK;
}

int ms_profile_enable(ms_ctx* c, int on) {
  if (!c) return MS_ERR_INVALID;
  if (on && !c->d_prof_ran) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_prof_ran), sizeof(uint32_t) * ms_ctx::PROF_RAN_CAP));
    HIPCHK(c, hipMemset(c->d_prof_ran, 0, sizeof(uint32_t) * ms_ctx::PROF_RAN_CAP));
  }
  // per-kernel timing needs one launch per kernel: the one-workgroup interpreter steps aside while it is on
  if (on && c->exec_on) {
    int rc = exec_flush(c);
    if (rc) return rc;
    exec_detach(&c->exec);
    c->exec_on = false;
  } else if (!on && c->exec_wanted && !c->exec_on) {
    c->exec.stream = c->stream;
    exec_attach(&c->exec);
    c->exec_on = true;
  }
  c->profiling = on != 0;
  return MS_OK;
}

int ms_profile_read(ms_ctx* c, double total_ms[MS_PROF_KINDS], int64_t launches[MS_PROF_KINDS]) {
  if (!c || !total_ms || !launches) return MS_ERR_INVALID;
  HIPCHK(c, hipStreamSynchronize(S(c)));
  std::vector<uint32_t> ran((size_t)c->prof_ran_next);
  if (c->prof_ran_next > 0)
    HIPCHK(c, hipMemcpy(ran.data(), c->d_prof_ran, sizeof(uint32_t) * ran.size(), hipMemcpyDeviceToHost));
  c->prof_ran_next = 0;
  for (auto& r : c->prof_pending) {
    float ms = 0.f;
    // a gated launch that found its gate closed (the probe behind it saw another code than the one it was queued for)
    // is not a sample of the kernel
    const bool empty = r.ran_idx >= 0 && ran[(size_t)r.ran_idx] == 0u;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && !empty) {
      c->prof_ms[r.kind] += ms;
      c->prof_n[r.kind] += 1;
    }
    c->prof_pool.push_back(r.a);
    c->prof_pool.push_back(r.b);
  }
  c->prof_pending.clear();
  for (int k = 0; k < MS_PROF_KINDS; ++k) {
    total_ms[k] = c->prof_ms[k];
    launches[k] = c->prof_n[k];
    c->prof_ms[k] = 0.0;
    c->prof_n[k] = 0;
  }
  return MS_OK;
}

int ms_plan_tiling(int nv, int nf, const double* positions, const int32_t* tri, int tile_vertices,
                   int shard_count, int64_t stats[8], int32_t* perm_out) {
  if (!stats) return fail(nullptr, MS_ERR_INVALID, "ms_plan_tiling: stats is NULL");
  Tiling t;
  std::string err;
  int rc = build_tiling(nv, nf, positions, tri, nullptr, tile_vertices, shard_count, t, err);
  if (rc != MS_OK) return fail(nullptr, rc, err);
  int64_t owners = 0, owned_corners = 0;
  for (int tile = 0; tile < t.n_tiles; ++tile) {
    const int n_owned = std::min(t.own, t.nv - tile * t.own);
    for (int p = t.tile_facet_off[tile]; p < t.tile_facet_off[tile + 1]; ++p) {
      const TileFacet& f = t.tile_facets[p];
      if (f.flags & TF_OWNER) ++owners;
      owned_corners += (f.l0 < n_owned) + (f.l1 < n_owned) + (f.l2 < n_owned);
    }
  }
  stats[0] = t.n_tiles;
  stats[1] = (int64_t)t.tile_facets.size();
  stats[2] = t.max_halo;
  stats[3] = t.max_tile_facets;
  stats[4] = t.dropped_facets;
  stats[5] = owners;
  stats[6] = owned_corners;
  stats[7] = (int64_t)gradient_lds_bytes(t.T, t.own + t.max_halo, t.max_ent, true, true);
  if (perm_out) memcpy(perm_out, t.perm.data(), sizeof(int32_t) * (size_t)nv);
  return MS_OK;
}

int ms_plan_tiling_conflicts(int nv, int nf, const double* positions, const int32_t* tri, int tile_vertices,
                             double model[4]) {
  if (!model) return fail(nullptr, MS_ERR_INVALID, "ms_plan_tiling_conflicts: model is NULL");
  Tiling t;
  std::string err;
  int rc = build_tiling(nv, nf, positions, tri, nullptr, tile_vertices, 1, t, err);
  if (rc != MS_OK) return fail(nullptr, rc, err);
  double rd_cyc = 0, at_cyc = 0, rd_free = 0, at_free = 0;
  int64_t rd_groups = 0, at_groups = 0;
  for (int tile = 0; tile < t.n_tiles; ++tile) {
    const int n_owned = std::min(t.own, t.nv - tile * t.own);
    const int f0 = t.tile_facet_off[tile], f1 = t.tile_facet_off[tile + 1];
    for (int k = 0; k < 3; ++k) {
      for (int g0 = f0; g0 < f1; g0 += 32) {  // reads: 32 lanes, bank pair = slot mod 32, same slot broadcasts
        int slots[32], ns = 0, cnt[32] = {0};
        for (int p = g0; p < std::min(f1, g0 + 32); ++p) {
          const TileFacet& f = t.tile_facets[p];
          const int s = k == 0 ? f.l0 : (k == 1 ? f.l1 : f.l2);
          bool dup = false;
          for (int q = 0; q < ns; ++q) dup |= slots[q] == s;
          if (!dup) {
            slots[ns++] = s;
            ++cnt[s & 31];
          }
        }
        int mx = 1;
        for (int q = 0; q < 32; ++q) mx = std::max(mx, cnt[q]);
        rd_cyc += mx;
        rd_free += mx == 1;
        ++rd_groups;
      }
      for (int g0 = f0; g0 < f1; g0 += 16) {  // atomics: 16 lanes, bank pair = slot mod 16, owned corners only
        int cnt[16] = {0}, any = 0;
        for (int p = g0; p < std::min(f1, g0 + 16); ++p) {
          const TileFacet& f = t.tile_facets[p];
          const int s = k == 0 ? f.l0 : (k == 1 ? f.l1 : f.l2);
          if (s < n_owned) {
            ++cnt[s & 15];
            any = 1;
          }
        }
        if (!any) continue;
        int mx = 1;
        for (int q = 0; q < 16; ++q) mx = std::max(mx, cnt[q]);
        at_cyc += mx;
        at_free += mx == 1;
        ++at_groups;
      }
    }
  }
  model[0] = rd_groups ? rd_cyc / rd_groups : 0.0;
  model[1] = at_groups ? at_cyc / at_groups : 0.0;
  model[2] = rd_groups ? rd_free / rd_groups : 0.0;
  model[3] = at_groups ? at_free / at_groups : 0.0;
  return MS_OK;
}

// ---- kernel-provider seam ---------------------------------------------------

int ms_surface_energy_and_gradient_host(int nv, int nf, const double* pos, const int32_t* tri,
                                        const double* gamma, double* grad, double* energy) {
  if (!pos || !gamma || !energy || (nf > 0 && !tri))
    return fail(nullptr, MS_ERR_INVALID, "ms_surface_energy_and_gradient_host: NULL argument");
  ms_ctx* c = nullptr;
  int rc = ms_create(&c, 0, nv, nf, pos, tri, nullptr, nullptr, nullptr, 0, 0, 1);
  if (rc) return rc;
  rc = ms_set_surface_tension(c, gamma);
  ms_params p = c->params;
  p.modules = MS_MOD_SURFACE;
  if (!rc) rc = ms_set_params(c, &p);
  double e[3] = {0, 0, 0};
  std::vector<double> g;
  if (!rc) {
    g.resize(3 * (size_t)nv);
    rc = ms_energy_and_gradient(c, e, grad ? g.data() : nullptr);
  }
  if (rc) g_last_error = c->err;
  ms_destroy(c);
  if (rc) return rc;
  *energy = e[0];
  if (grad)
    for (size_t i = 0; i < g.size(); ++i) grad[i] += g[i];  // intent(inout): accumulate
  return MS_OK;
}

int ms_grad_cotan_batch_host(int n, const double* u, const double* v, double* grad_u, double* grad_v) {
  if (n < 0 || !u || !v || !grad_u || !grad_v)
    return fail(nullptr, MS_ERR_INVALID, "ms_grad_cotan_batch_host: bad argument");
  const size_t b = sizeof(double) * 3 * (size_t)n;
  DevBuf du, dv, dgu, dgv;
  SEAM_HIP(du.alloc(b));
  SEAM_HIP(dv.alloc(b));
  SEAM_HIP(dgu.alloc(b));
  SEAM_HIP(dgv.alloc(b));
  SEAM_HIP(hipMemcpy(du.p, u, b, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(dv.p, v, b, hipMemcpyHostToDevice));
  SEAM_HIP(launch_grad_cotan(n, du.as<double>(), dv.as<double>(), dgu.as<double>(), dgv.as<double>(), nullptr));
  SEAM_HIP(hipMemcpy(grad_u, dgu.p, b, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(grad_v, dgv.p, b, hipMemcpyDeviceToHost));
  return MS_OK;
}

int ms_apply_beltrami_laplacian_host(int dim, int nv, int nf, const double* weights,
                                     const int32_t* tri, const double* field, double* out) {
  if (dim <= 0 || nv <= 0 || nf < 0 || !weights || !tri || !field || !out)
    return fail(nullptr, MS_ERR_INVALID, "ms_apply_beltrami_laplacian_host: bad argument");
  const size_t bw = sizeof(double) * 3 * (size_t)nf, bt = sizeof(int32_t) * 3 * (size_t)nf,
               bf = sizeof(double) * (size_t)nv * dim;
  DevBuf dw, dt, df, dout;
  SEAM_HIP(dw.alloc(bw));
  SEAM_HIP(dt.alloc(bt));
  SEAM_HIP(df.alloc(bf));
  SEAM_HIP(dout.alloc(bf));
  SEAM_HIP(hipMemcpy(dw.p, weights, bw, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(dt.p, tri, bt, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(df.p, field, bf, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemset(dout.p, 0, bf));
  SEAM_HIP(launch_laplacian_scatter(dim, nv, nf, dw.as<double>(), dt.as<int32_t>(), df.as<double>(),
                                    dout.as<double>(), nullptr));
  SEAM_HIP(hipMemcpy(out, dout.p, bf, hipMemcpyDeviceToHost));
  return MS_OK;
}

int ms_p1_triangle_divergence_host(int nv, int nf, const double* pos, const double* tilts,
                                   const int32_t* tri, double* div_tri, double* area, double* g0,
                                   double* g1, double* g2) {
  if (nv <= 0 || nf < 0 || !pos || !tilts || !tri || !div_tri || !area || !g0 || !g1 || !g2)
    return fail(nullptr, MS_ERR_INVALID, "ms_p1_triangle_divergence_host: bad argument");
  const size_t bp = sizeof(double) * 3 * (size_t)nv, bt = sizeof(int32_t) * 3 * (size_t)nf,
               b1 = sizeof(double) * (size_t)nf, b3 = 3 * b1;
  DevBuf dp, dtl, dt, dd, da, d0, d1, d2;
  SEAM_HIP(dp.alloc(bp));
  SEAM_HIP(dtl.alloc(bp));
  SEAM_HIP(dt.alloc(bt));
  SEAM_HIP(dd.alloc(b1));
  SEAM_HIP(da.alloc(b1));
  SEAM_HIP(d0.alloc(b3));
  SEAM_HIP(d1.alloc(b3));
  SEAM_HIP(d2.alloc(b3));
  SEAM_HIP(hipMemcpy(dp.p, pos, bp, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(dtl.p, tilts, bp, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(dt.p, tri, bt, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemset(dd.p, 0, b1));
  SEAM_HIP(hipMemset(da.p, 0, b1));
  SEAM_HIP(hipMemset(d0.p, 0, b3));
  SEAM_HIP(hipMemset(d1.p, 0, b3));
  SEAM_HIP(hipMemset(d2.p, 0, b3));
  SEAM_HIP(launch_p1_divergence(nv, nf, dp.as<double>(), dtl.as<double>(), dt.as<int32_t>(),
                                dd.as<double>(), da.as<double>(), d0.as<double>(), d1.as<double>(),
                                d2.as<double>(), nullptr));
  SEAM_HIP(hipMemcpy(div_tri, dd.p, b1, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(area, da.p, b1, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(g0, d0.p, b3, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(g1, d1.p, b3, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(g2, d2.p, b3, hipMemcpyDeviceToHost));
  return MS_OK;
}

int ms_compute_curvature_data_host(int nv, int nf, const double* pos, const int32_t* tri,
                                   double* k_vecs, double* vertex_areas, double* weights,
                                   double* va0, double* va1, double* va2) {
  if (nv <= 0 || nf < 0 || !pos || !tri || !k_vecs || !vertex_areas || !weights)
    return fail(nullptr, MS_ERR_INVALID, "ms_compute_curvature_data_host: bad argument");
  const size_t bp = sizeof(double) * 3 * (size_t)nv, bt = sizeof(int32_t) * 3 * (size_t)nf,
               b1 = sizeof(double) * (size_t)nf, bv = sizeof(double) * (size_t)nv;
  DevBuf dp, dt, dk, da, dw, d0, d1, d2;
  SEAM_HIP(dp.alloc(bp));
  SEAM_HIP(dt.alloc(bt));
  SEAM_HIP(dk.alloc(bp));
  SEAM_HIP(da.alloc(bv));
  SEAM_HIP(dw.alloc(3 * b1));
  SEAM_HIP(d0.alloc(b1));
  SEAM_HIP(d1.alloc(b1));
  SEAM_HIP(d2.alloc(b1));
  SEAM_HIP(hipMemcpy(dp.p, pos, bp, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemcpy(dt.p, tri, bt, hipMemcpyHostToDevice));
  SEAM_HIP(hipMemset(dk.p, 0, bp));
  SEAM_HIP(hipMemset(da.p, 0, bv));
  SEAM_HIP(hipMemset(dw.p, 0, 3 * b1));
  SEAM_HIP(hipMemset(d0.p, 0, b1));
  SEAM_HIP(hipMemset(d1.p, 0, b1));
  SEAM_HIP(hipMemset(d2.p, 0, b1));
  SEAM_HIP(launch_curvature_raw(nv, nf, dp.as<double>(), dt.as<int32_t>(), dk.as<double>(),
                                da.as<double>(), dw.as<double>(), d0.as<double>(), d1.as<double>(),
                                d2.as<double>(), nullptr));
  SEAM_HIP(hipMemcpy(k_vecs, dk.p, bp, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(vertex_areas, da.p, bv, hipMemcpyDeviceToHost));
  SEAM_HIP(hipMemcpy(weights, dw.p, 3 * b1, hipMemcpyDeviceToHost));
  if (va0) SEAM_HIP(hipMemcpy(va0, d0.p, b1, hipMemcpyDeviceToHost));
  if (va1) SEAM_HIP(hipMemcpy(va1, d1.p, b1, hipMemcpyDeviceToHost));
  if (va2) SEAM_HIP(hipMemcpy(va2, d2.p, b1, hipMemcpyDeviceToHost));
  return MS_OK;
}

}  // extern "C"
