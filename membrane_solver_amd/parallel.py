"""Multi-GPU sharding of the minimizer step: one process per GPU, RCCL over xGMI
through ``torch.distributed`` (backend "nccl" is RCCL on ROCm).

The reference has no parallelism at all; this is the
facet-block partition BASELINE.json asks for.  Tiles (blocks of 256 vertices in
patch order, each listing every facet that touches it) are dealt to ranks in
contiguous ranges, so every rank OWNS a contiguous row range of each per-vertex
vector and evaluates only its own facet blocks.  Positions are replicated.

Exchanges per gradient evaluation ("simple mode" of SURVEY 8e):
  * all-gather of the bending back-prop factors fK (nv,3) + fA (nv,2) between
    the energy pass and the gradient pass (owner rows -> every rank);
  * all-gather of the direction d (nv,3) after the direction pass;
  * small all-gathers of the 16 reduction scalars (energies, <g,gC>, |g|^2,
    <g,d>, max|d|, min edge, guard flag), folded on the host in rank order, so
    every rank takes identical Armijo decisions.
The per-vertex gradient itself never travels: the volume-row projection, the
fixed-row zeroing and the per-row Polak-Ribiere beta are all row-local, so the
dense gradient all-reduce of the north-star text reduces to these all-gathers
of owner rows (half the bytes of an all-reduce of zero-padded partial vectors).
Accepting a trial needs no exchange either: x and d are replicated, every rank
forms x + alpha*d on all rows itself.

``ShardedStepper`` holds the control flow (a restatement of ms_step, i.e. of
runtime/minimizer.py:1314-1374 + line_search.py:267-426 of the reference) on
top of a small backend protocol, so the same code runs on the HIP backend under
RCCL and on a NumPy/oracle backend under gloo in the CPU tests.
"""

from __future__ import annotations

import json
import math
import time
from dataclasses import dataclass

import numpy as np

from . import _lib as L

SUM_SLOTS = (L.MS_S_ESURF, L.MS_S_VOL, L.MS_S_EBEND, L.MS_S_GGC, L.MS_S_GCGC, L.MS_S_GNORM2,
             L.MS_S_GDOTD, L.MS_S_ETILT)
MIN_SLOTS = (L.MS_S_MINEDGE2,)
MAX_SLOTS = (L.MS_S_GUARD, L.MS_S_MAXD2)
ENERGY_SLOTS = (L.MS_S_ESURF, L.MS_S_VOL, L.MS_S_EBEND, L.MS_S_MINEDGE2, L.MS_S_GUARD)
GRAD_SLOTS = (L.MS_S_GGC, L.MS_S_GCGC)
DIR_SLOTS = (L.MS_S_GNORM2, L.MS_S_GDOTD, L.MS_S_MAXD2)


def fold_scalars(per_rank: np.ndarray) -> np.ndarray:
    """(world, MS_NSCAL) -> (MS_NSCAL,), folded in rank order (deterministic)."""
    out = np.zeros(L.MS_NSCAL)
    for s in SUM_SLOTS:
        acc = 0.0
        for r in range(per_rank.shape[0]):
            acc += float(per_rank[r, s])
        out[s] = acc
    for s in MIN_SLOTS:
        out[s] = float(np.min(per_rank[:, s]))
    for s in MAX_SLOTS:
        out[s] = float(np.max(per_rank[:, s]))
    return out


@dataclass
class ShardStepResult:
    success: bool
    converged: bool
    trials: int
    guard_rejects: int
    next_step: float
    energy: float
    alpha: float
    energy_eval: float
    grad_norm: float
    g_dot_d: float
    volume: float


class ShardedStepper:
    """One minimizer step over sharded tiles.  ``backend`` protocol:

    modules (int), volume_stiffness, target_volume, nf,
    phase_energy(use_direction, alpha, write_trial, guard, write_bending_factors),
    phase_gradient(), phase_direction(stepper, use_history),
    phase_commit_trial(alpha, keep_history), fetch_scalars() -> (16,), store_scalars((16,)),
    allgather_rows(buffer_id)   # owner rows -> all ranks, in place
    allgather_scalars((16,)) -> (world, 16)
    """

    def __init__(self, backend, *, stepper: int = L.MS_STEPPER_CG, max_iter: int = 10, beta: float = 0.7,
                 c: float = 1e-4, gamma: float = 1.5, alpha_max_factor: float = 10.0,
                 restart_interval: int = 10, edge_fraction: float = 0.0, reuse_energy0: bool = False):
        self.b = backend
        self.stepper = stepper
        self.max_iter, self.beta, self.c, self.gamma = max_iter, beta, c, gamma
        self.alpha_max_factor, self.restart_interval = alpha_max_factor, restart_interval
        self.edge_fraction, self.reuse_energy0 = edge_fraction, reuse_energy0
        self.have_history = False
        self.iter_count = 0
        self.scal = np.zeros(L.MS_NSCAL)

    def reset(self):
        """ConjugateGradient.reset (conjugate_gradient.py:44-50)."""
        self.have_history = False
        self.iter_count = 0

    # -- helpers ---------------------------------------------------------------
    def _exchange(self, slots):
        """Fold the given slots over ranks, keep the rest, push the result to the device."""
        local = self.b.fetch_scalars()
        folded = fold_scalars(self.b.allgather_scalars(local))
        for s in slots:
            self.scal[s] = folded[s]
        self.b.store_scalars(self.scal)

    def _energy(self):
        m = self.b.modules
        e = 0.0
        if m & L.MS_MOD_SURFACE:
            e += self.scal[L.MS_S_ESURF]
        if m & L.MS_MOD_BENDING:
            e += self.scal[L.MS_S_EBEND]
        if m & L.MS_MOD_VOLUME_PENALTY:
            delta = self.scal[L.MS_S_VOL] - self.b.target_volume
            e += 0.5 * self.b.volume_stiffness * (delta * delta)
        return e

    # -- the step (mirror of ms_step in csrc/ms_api.cpp) ---------------------------
    def step(self, step_size: float, tol: float = 0.0) -> ShardStepResult:
        b = self.b
        cg = self.stepper == L.MS_STEPPER_CG
        use_history = cg and self.have_history and (self.iter_count % self.restart_interval != 0)
        bend = bool(b.modules & L.MS_MOD_BENDING)
        b.phase_energy(False, 0.0, False, False, True)
        if bend:
            b.allgather_rows(L.MS_BUF_FK)
            b.allgather_rows(L.MS_BUF_FA)
        if b.modules & L.MS_MOD_VOLUME_PENALTY:
            self._exchange(ENERGY_SLOTS)  # the penalty factor k (V - V0) needs the global V
        b.phase_gradient()
        self._exchange(ENERGY_SLOTS + GRAD_SLOTS)
        b.phase_direction(self.stepper, use_history)
        b.allgather_rows(L.MS_BUF_D)
        self._exchange(DIR_SLOTS)
        E_eval = self._energy()
        grad_norm = math.sqrt(self.scal[L.MS_S_GNORM2])
        g_dot_d = self.scal[L.MS_S_GDOTD]
        max_dir = math.sqrt(self.scal[L.MS_S_MAXD2])
        res = ShardStepResult(False, False, 0, 0, step_size, E_eval, 0.0, E_eval, grad_norm, g_dot_d,
                              self.scal[L.MS_S_VOL])
        if grad_norm < tol:
            res.converged = res.success = True
            return res
        energy0 = E_eval
        if not self.reuse_energy0:
            b.phase_energy(False, 0.0, False, False, False)
            self._exchange(ENERGY_SLOTS)
            energy0 = self._energy()
        min_edge = math.sqrt(self.scal[L.MS_S_MINEDGE2]) if b.nf > 0 else 0.0
        res.energy = energy0
        safe_limit = 0.3 * min_edge if min_edge > 0.0 else math.inf
        if g_dot_d >= 0.0:
            return res
        alpha = step_size
        if self.edge_fraction > 0.0 and min_edge > 0.0 and max_dir > 0.0:
            alpha = min(alpha, self.edge_fraction * min_edge / max_dir)
        alpha_max = self.alpha_max_factor * step_size
        for _ in range(self.max_iter):
            safe_small = alpha * max_dir < safe_limit
            b.phase_energy(True, alpha, False, not safe_small, False)
            self._exchange(ENERGY_SLOTS)
            if (not safe_small) and self.scal[L.MS_S_GUARD] > 0.0:
                res.guard_rejects += 1
                alpha *= self.beta
                if alpha < 1e-8:
                    break
                continue
            res.trials += 1
            E_t = self._energy()
            if E_t <= energy0 + self.c * alpha * g_dot_d:
                b.phase_commit_trial(alpha, cg)
                if cg:
                    self.have_history = True
                    self.iter_count += 1
                res.success = True
                res.alpha = alpha
                res.energy = E_t
                res.volume = self.scal[L.MS_S_VOL]
                res.next_step = min(alpha * self.gamma, alpha_max)
                return res
            alpha *= self.beta
            if alpha < 1e-8:
                break
        res.next_step = max(max(alpha * self.beta, 0.0), step_size * self.beta)
        return res


class HipShardBackend:
    """HIP kernels on this rank's tile range; collectives through torch.distributed."""

    def __init__(self, positions, tri_rows, *, rank: int, world: int, device: int, tile_vertices: int = 0,
                 fixed=None, boundary=None, body_facets=None, group=None):
        import torch

        from .device import DeviceMesh

        if group is None:
            import torch.distributed as group  # module-level collectives of the default process group
        self.torch, self.dist = torch, group
        self.rank, self.world = rank, world
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        self.dm = DeviceMesh(positions, tri_rows, fixed=fixed, boundary=boundary, body_facets=body_facets,
                             device=device, tile_vertices=tile_vertices, shard_rank=rank, shard_count=world)
        self.nf = self.dm.nf
        info = self.dm.shard_info()
        self.nvp, self.rows = int(info["nvp"]), int(info["rows_per_shard"])
        # per-vertex state lives in a torch tensor so RCCL can work on it in place
        nbytes = self.dm.state_bytes()
        self.state = torch.empty(nbytes // 8, dtype=torch.float64, device=self.device)
        torch.cuda.synchronize()
        self.dm.set_stream(torch.cuda.current_stream().cuda_stream)
        self.dm.rebind_state(self.state.data_ptr(), nbytes)
        self.modules = L.MS_MOD_SURFACE
        self.volume_stiffness = 0.0
        self.target_volume = 0.0
        self._scal_all = torch.empty(world * L.MS_NSCAL, dtype=torch.float64, device=self.device)
        self._scal_mine = torch.empty(L.MS_NSCAL, dtype=torch.float64, device=self.device)

    def configure(self, *, modules, gamma=None, kappa=None, c0=None, **params):
        if gamma is not None:
            self.dm.set_surface_tension(gamma)
        if kappa is not None:
            self.dm.set_bending_params(kappa, c0)
        self.dm.set_params(modules=modules, **params)
        self.modules = modules
        self.volume_stiffness = float(params.get("volume_stiffness", 0.0))
        self.target_volume = float(params.get("target_volume", 0.0))

    def _view(self, buffer_id):
        ptr, nbytes = self.dm.device_buffer(buffer_id)
        off = (ptr - self.state.data_ptr()) // 8
        ncomp = 2 if buffer_id == L.MS_BUF_FA else 3
        return self.state[off: off + self.nvp * ncomp].view(self.nvp, ncomp)

    # backend protocol --------------------------------------------------------------
    def phase_energy(self, use_direction, alpha, write_trial, guard, write_bending_factors):
        self.dm.phase_energy(use_direction=use_direction, alpha=alpha, write_trial=write_trial,
                             guard=guard, write_bending_factors=write_bending_factors)

    def phase_gradient(self):
        self.dm.phase_gradient()

    def phase_direction(self, stepper, use_history):
        self.dm.phase_direction(stepper, use_history)

    def phase_commit_trial(self, alpha, keep_history):
        self.dm.phase_commit_trial(alpha, keep_history)

    def fetch_scalars(self):
        return self.dm.fetch_scalars()

    def store_scalars(self, values):
        self.dm.store_scalars(values)

    def allgather_rows(self, buffer_id):
        full = self._view(buffer_id)
        mine = full[self.rank * self.rows: (self.rank + 1) * self.rows].clone()
        self.dist.all_gather_into_tensor(full, mine)

    def allgather_scalars(self, local):
        self._scal_mine.copy_(self.torch.from_numpy(np.ascontiguousarray(local)))
        self.dist.all_gather_into_tensor(self._scal_all, self._scal_mine)
        return self._scal_all.cpu().numpy().reshape(self.world, L.MS_NSCAL)


def bench_main(args, rank: int, world: int, local_rank: int):
    """bench.py --gpus N (N > 1): same workload as the single-GPU bench, tiles sharded
    over N ranks (strong scaling), timed with barrier + synchronize on both sides and
    the MAX over ranks."""
    import torch
    import torch.distributed as dist

    from . import meshgen

    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    P, T = meshgen.icosphere(args.freq)
    P = meshgen.smooth_displace(P, 0.05)
    nv, nf = P.shape[0], T.shape[0]
    be = HipShardBackend(P, T, rank=rank, world=world, device=local_rank, tile_vertices=args.tile)
    be.configure(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, gamma=np.ones(nf), kappa=np.ones(nv),
                 c0=np.zeros(nv))
    drv = ShardedStepper(be, stepper=L.MS_STEPPER_CG)
    step = args.step_size

    def run(n):
        nonlocal step
        acc = trials = 0
        for _ in range(n):
            r = drv.step(step, tol=1e-6)
            step = r.next_step
            acc += int(r.success)
            trials += r.trials
            if not r.success:
                drv.reset()  # minimizer.py:1462-1464
        return acc, trials, r

    run(args.warmup)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, trials, r = run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "minimizer steps/sec (energy+grad+CG) on 2M-facet icosphere",
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"class-I icosphere f={args.freq} (nv={nv}, nf={nf}), surface + Helfrich "
                                   "bending (analytic cotan gradient), CG stepper, Armijo line search, "
                                   "energy0 re-evaluated",
                       "parallelism": f"tiles (facet blocks) sharded over {world} GPUs; RCCL all-gather of "
                                      "owner rows of fK/fA and d; replicated positions",
                       "tile_vertices": args.tile or 256, "initial_step_size": args.step_size},
            "steps_accepted": acc, "line_search_trials": trials, "energy_end": r.energy,
        }))
    dist.destroy_process_group()
