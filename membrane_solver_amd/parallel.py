"""Multi-GPU sharding of the minimizer step: one process per GPU, RCCL over xGMI
through ``torch.distributed`` (backend "nccl" is RCCL on ROCm).

The reference has no parallelism at all; this is the
facet-block partition BASELINE.json asks for.  Tiles (blocks of 256 vertices in
patch order, each listing every facet that touches it) are dealt to ranks in
contiguous ranges, so every rank OWNS a contiguous row range of each per-vertex
vector and evaluates only its own facet blocks.  A rank reads its own rows plus
the HALO rows of its tiles (a thin band along the patch border: the patch order
is a recursive coordinate bisection, so a rank's rows form a compact surface patch).

One exchange = one fixed-size all-gather per rank of
    [MS_NSCAL reduction scalars | this rank's BOUNDARY rows of the listed buffers]
(boundary rows = rows it owns that other ranks read as halo; a few thousand rows
of a million).  Every rank scatters the peers' rows into its buffers and folds the
scalar headers on the host in rank order, so all ranks take identical Armijo
decisions.  Per accepted step (no constraint row, reuse level 2):
  * gradient pass with fused direction  ->  exchange [|g|^2, <g,d>, max|d| | d]
  * trial energy pass (writes the bending factors at the trial point)
                                        ->  exchange [energies, min edge | fK, fA]
and nothing else: the per-vertex gradient itself never travels (fixed-row zeroing,
volume-row projection and the per-row Polak-Ribiere beta are row-local), accepting
a trial is x += alpha*d on the rows a rank reads, and the accepted trial's factors
and energies are the next step's (same reuse as ms_step, include/membrane_hip.h).

"Simple mode" (BASELINE.json's north star names it: a dense RCCL all-reduce of the per-vertex
vector) is kept as the comparison point: ``HipShardBackend(exchange="dense")`` /
``MS_SHARD_EXCHANGE=dense`` sends only the scalar header through the all-gather and
all-reduces every listed buffer over ALL its rows (each row has exactly one non-zero
contributor, its owner: the sum is the owner's value bit for bit) -- 24.6 MB per 3-vector at
the headline size instead of ~100 KB of boundary rows, same trajectory.

``ShardedStepper`` holds the control flow (a restatement of ms_step, i.e. of
runtime/minimizer.py:1314-1374 + line_search.py:267-426 of the reference) on
top of a small backend protocol, so the same code runs on the HIP backend under
RCCL and on a NumPy/oracle backend under gloo in the CPU tests.
"""

from __future__ import annotations

import ctypes
import json
import os
import sys
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


_SUM_IDX = np.array(SUM_SLOTS)
_MIN_IDX = np.array(MIN_SLOTS)
_MAX_IDX = np.array(MAX_SLOTS)


def fold_scalars(per_rank: np.ndarray) -> np.ndarray:
    """(world, MS_NSCAL) -> (MS_NSCAL,), folded in rank order (deterministic: every rank adds the
    same doubles in the same order)."""
    out = np.zeros(L.MS_NSCAL)
    acc = per_rank[0, _SUM_IDX].copy()
    for r in range(1, per_rank.shape[0]):
        acc += per_rank[r, _SUM_IDX]
    out[_SUM_IDX] = acc
    out[_MIN_IDX] = per_rank[:, _MIN_IDX].min(axis=0)
    out[_MAX_IDX] = per_rank[:, _MAX_IDX].max(axis=0)
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
    phase_gradient_direction(stepper, use_history)      # fused, no constraint row
    phase_commit_trial(alpha, keep_history), set_factors_valid(bool),
    store_scalars((16,))                                # folded values -> device
    exchange(buffer_ids) -> (world, 16)                 # see module docstring
    """

    def __init__(self, backend, *, stepper: int = L.MS_STEPPER_CG, max_iter: int = 10, beta: float = 0.7,
                 c: float = 1e-4, gamma: float = 1.5, alpha_max_factor: float = 10.0,
                 restart_interval: int = 10, edge_fraction: float = 0.0, reuse_energy0: int = 2):
        self.b = backend
        self.stepper = stepper
        self.max_iter, self.beta, self.c, self.gamma = max_iter, beta, c, gamma
        self.alpha_max_factor, self.restart_interval = alpha_max_factor, restart_interval
        self.edge_fraction, self.reuse_energy0 = edge_fraction, int(reuse_energy0)
        self.have_history = False
        self.iter_count = 0
        self.scal = np.zeros(L.MS_NSCAL)
        self.carry_valid = False   # factors + energy scalars describe the current x
        self.grad_valid = False    # buffer G holds the finalized gradient of the current x
        self.exchanges = 0

    def reset(self):
        """ConjugateGradient.reset (conjugate_gradient.py:44-50)."""
        self.have_history = False
        self.iter_count = 0

    def invalidate(self):
        """Call after changing positions / parameters behind the stepper's back."""
        self.carry_valid = self.grad_valid = False

    # -- helpers ---------------------------------------------------------------
    def _exchange(self, buffers, slots, push=False):
        """Make `buffers` valid on every row this rank reads, fold `slots` over ranks (rank
        order), keep the other slots; push=True also writes the folded values to the device
        (the gradient pass reads the global volume, the direction pass <g,gC> and <gC,gC>)."""
        folded = fold_scalars(self.b.exchange(tuple(buffers)))
        for s in slots:
            self.scal[s] = folded[s]
        if push:
            self.b.store_scalars(self.scal)
        self.exchanges += 1

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
        constraint = bool(b.modules & L.MS_CON_VOLUME)
        penalty = bool(b.modules & L.MS_MOD_VOLUME_PENALTY)
        carry_mode = self.reuse_energy0 >= 2
        factor_bufs = (L.MS_BUF_FK, L.MS_BUF_FA) if bend else ()
        carried = carry_mode and self.carry_valid
        if not carried:
            b.phase_energy(False, 0.0, False, False, True)
            # the gradient pass reads the factors on halo rows and (penalty) the global volume
            self._exchange(factor_bufs, ENERGY_SLOTS, push=penalty)
            self.grad_valid = False
        if carried and self.grad_valid and not constraint:
            b.phase_direction(self.stepper, use_history)  # x has not moved: direction only
        elif constraint:
            b.phase_gradient()
            self._exchange((), GRAD_SLOTS, push=True)
            b.phase_direction(self.stepper, use_history)
        else:
            b.phase_gradient_direction(self.stepper, use_history)
        self._exchange((L.MS_BUF_D,), DIR_SLOTS)
        self.carry_valid = carry_mode
        self.grad_valid = carry_mode and not constraint
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
        if self.reuse_energy0 == 0:
            b.phase_energy(False, 0.0, False, False, False)
            self._exchange((), ENERGY_SLOTS)
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
            b.phase_energy(True, alpha, False, not safe_small, carry_mode)
            if carry_mode:
                self.carry_valid = False  # the factor buffers now belong to the trial point
            self._exchange(factor_bufs if carry_mode else (), ENERGY_SLOTS)
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
                self.grad_valid = False
                if carry_mode:
                    b.set_factors_valid(True)
                    if penalty:
                        b.store_scalars(self.scal)  # the next gradient pass reads the new volume
                    self.carry_valid = True
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


class LibraryShardedStepper:
    """ShardedStepper's control flow run inside the library (ms_shard_step): per exchange
    pack -> ncclAllGather on the context's stream -> unpack -> mailbox poll, nothing of the
    interpreter in the loop.  Needs ``HipShardBackend.enable_library_driver()`` first."""

    def __init__(self, backend, *, stepper: int = L.MS_STEPPER_CG, reuse_energy0: int = 2, **params):
        self.b = backend
        self.stepper = stepper
        self.reuse_energy0 = int(reuse_energy0)
        self.params = params

    @property
    def exchanges(self) -> int:
        return self.b.dm.shard_exchange_count()

    def reset(self):
        self.b.dm.reset_stepper()

    def step(self, step_size: float, tol: float = 0.0):
        return self.b.dm.shard_step(stepper=self.stepper, step_size=step_size, tol=tol,
                                    reuse_energy0=self.reuse_energy0, **self.params)

    def run(self, n_steps: int, step_size: float, tol: float = 0.0):
        """n_steps iterations of the minimizer loop (step, stepper reset on failure, zero-step
        exit) inside the library: ms_minimize over ms_shard_step.  -> ms_minimize_result"""
        p = self.params
        mp = L.ms_minimize_params()
        mp.stepper = L.ms_stepper_params(int(self.stepper), int(p.get("max_iter", 10)), float(p.get("beta", 0.7)),
                                         float(p.get("c", 1e-4)), float(p.get("gamma", 1.5)),
                                         float(p.get("alpha_max_factor", 10.0)),
                                         int(p.get("restart_interval", 10)), float(p.get("edge_fraction", 0.0)),
                                         self.reuse_energy0)
        mp.step_size, mp.tol = float(step_size), float(tol)
        mp.fixed_step_mode, mp.fixed_step = 0, float(step_size)
        mp.max_zero_steps, mp.step_size_floor = 10, 1e-8
        out, _log = self.b.dm.minimize(mp, n_steps)
        return out


class HipShardBackend:
    """HIP kernels on this rank's tile range; collectives through torch.distributed."""

    def __init__(self, positions, tri_rows, *, rank: int, world: int, device: int, tile_vertices: int = 0,
                 fixed=None, boundary=None, body_facets=None, group=None, debug_poison=False, exchange=None):
        import torch

        self.exchange_mode = exchange or os.environ.get("MS_SHARD_EXCHANGE", "halo")
        if self.exchange_mode not in ("halo", "dense"):
            raise ValueError(f"exchange mode {self.exchange_mode!r}: 'halo' or 'dense'")

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
        # ONE explicit stream orders everything this backend does: the library's kernels (pack / unpack / passes),
        # torch's collectives (ProcessGroupNCCL orders its all-gather against the CURRENT stream) and the few torch
        # ops on the state tensor.  torch's default stream has handle 0, for which ms_set_stream would create a
        # private stream that nothing else synchronises with.
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        assert self.stream.cuda_stream != 0
        self.dm.set_stream(self.stream.cuda_stream)
        self.dm.rebind_state(self.state.data_ptr(), nbytes)
        self.modules = L.MS_MOD_SURFACE
        self.volume_stiffness = 0.0
        self.target_volume = 0.0
        self.debug_poison = bool(debug_poison)  # tests: NaN every non-owned row before an exchange
        self.boundary = self.dm.boundary_info()
        # message buffers for the widest exchange (fK 3 + fA 2 + d 3 components per boundary row)
        n_max = L.MS_NSCAL + 10 * self.boundary["max_rows"]  # at most 2 x (fK 3 + fA 2) per boundary row (pair launch)
        self._send = torch.zeros(n_max, dtype=torch.float64, device=self.device)
        self._recv = torch.zeros(world * n_max, dtype=torch.float64, device=self.device)
        self._plans = {}

    def configure(self, *, modules, gamma=None, kappa=None, c0=None, **params):
        tilt_bits = modules & ~(L.MS_MOD_SURFACE | L.MS_MOD_BENDING | L.MS_MOD_VOLUME_PENALTY | L.MS_CON_VOLUME
                                | getattr(L, "MS_TRACK_VOLUME", 0))
        if tilt_bits:
            raise L.MembraneHipError(f"module bits {tilt_bits:#x}: only surface / bending / volume are sharded "
                                     "(ms_shard_step refuses the tilt modules as well)")
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

    def phase_gradient_direction(self, stepper, use_history):
        self.dm.phase_gradient_direction(stepper, use_history)

    def set_factors_valid(self, valid):
        self.dm.phase_set_factors_valid(valid)

    def enable_library_driver(self):
        if self.exchange_mode != "halo":
            raise L.MembraneHipError("the in-library shard driver exchanges boundary rows only (exchange='halo')")
        return self._enable_library_driver()

    def _enable_library_driver(self):
        """Give the context its own RCCL communicator (ncclUniqueId from rank 0, broadcast over
        the existing torch.distributed group) or, for an in-process group, an all-gather callback."""
        dist = self.dist
        if hasattr(dist, "all_gather_into_tensor") and not hasattr(dist, "broadcast"):
            # in-process stand-in (tests): route the library's all-gather through the group
            torch = self.torch
            n_max = self._send.numel()

            def gather(send_ptr, recv_ptr, nbytes):
                n = nbytes // 8
                assert n <= n_max
                hip = ctypes.CDLL(L.HIP_RUNTIME_PATH)
                hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                               ctypes.c_void_p]
                send, recv = self._send[:n], self._recv[: self.world * n]
                # everything on this rank's ONE stream (the library has synchronised it before calling): a
                # null-stream hipMemcpy would be ordered with nothing here -- torch streams are non-blocking and a
                # device-to-device hipMemcpy may return before it has run
                cur = torch.cuda.current_stream()
                rc = hip.hipMemcpyAsync(send.data_ptr(), send_ptr, nbytes, 3, cur.cuda_stream)
                assert rc == 0, rc
                dist.all_gather_into_tensor(recv, send)
                rc = hip.hipMemcpyAsync(recv_ptr, recv.data_ptr(), nbytes * self.world, 3, cur.cuda_stream)
                assert rc == 0, rc
                cur.synchronize()  # the callback is synchronous as a whole

            self.dm.shard_set_allgather(gather)
            return
        torch = self.torch
        backend = dist.get_backend()
        dev = self.device if backend == "nccl" else torch.device("cpu")
        # every rank first checks, locally, that librccl binds (ncclCommInitRank below is a
        # collective: a rank that cannot take part must be known to all before anyone enters it)
        try:
            my_id = self.dm.shard_unique_id()
            ok = 1
        except L.MembraneHipError:
            my_id, ok = bytes(128), 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            raise L.MembraneHipError("librccl could not be bound on every rank")
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            idt.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        self.dm.shard_comm_init(bytes(idt.cpu().numpy().tobytes()))

    def enable_peer_exchange(self):
        """The library driver's exchange without a collective library in the step: every rank's pack kernel writes
        its message straight into every peer's receive slab (IPC-mapped device memory over xGMI), flag words order
        it (include/membrane_hip.h, ms_shard_peer_*).  In-process groups (tests) hand each other raw pointers;
        process groups all-gather the IPC handles once, over whatever backend the group has."""
        if self.exchange_mode != "halo":
            raise L.MembraneHipError("the peer-to-peer exchange moves boundary rows only (exchange='halo')")
        dist = self.dist
        if hasattr(dist, "share"):  # in-process stand-in: one address space
            mine = self.dm.shard_peer_local()
            table = dist.share(self.rank, mine)
            self.dm.shard_peer_set_pointers([t[0] for t in table], [t[1] for t in table])
            # (one process, one device: the ranks share its hardware queues, so they wait for each other on the host)
            self.dm.shard_peer_set_barrier(dist.bar.wait)
            return
        mine = self.dm.shard_peer_export()
        table = [None] * self.world
        dist.all_gather_object(table, mine)
        self.dm.shard_peer_open(b"".join(table))
        dist.barrier()

    def gather_positions(self) -> np.ndarray:
        """Assemble the full position array (owner rows of x from every rank) -- between steps a
        rank only keeps the rows it reads (own + halo) current."""
        full = self._view(L.MS_BUF_X)
        mine = full[self.rank * self.rows: (self.rank + 1) * self.rows].reshape(-1).clone()
        self.dist.all_gather_into_tensor(full.view(-1), mine)
        return self.dm.get_positions()

    def exchange(self, buffers):
        """pack -> RCCL all-gather of the fixed-size messages -> unpack (see module docstring)."""
        if self.debug_poison:
            r0, r1 = self.rank * self.rows, (self.rank + 1) * self.rows
            for bid in buffers:
                v = self._view(bid)
                v[:r0] = float("nan")
                v[r1:] = float("nan")
        if self.exchange_mode == "dense":
            # the north star's simple mode: scalar headers by all-gather, every listed buffer by a dense all-reduce
            per_rank = self._exchange_message(())
            r0, r1 = self.rank * self.rows, (self.rank + 1) * self.rows
            for bid in buffers:
                v = self._view(bid)
                v[:r0].zero_()
                v[r1:].zero_()
                self.dist.all_reduce(v)
            return per_rank
        return self._exchange_message(buffers)

    def _exchange_message(self, buffers):
        plan = self._plans.get(buffers)
        if plan is None:  # message size and tensor views per buffer set, resolved once
            n = self.dm.exchange_bytes(buffers) // 8
            send, recv = self._send[:n], self._recv[: self.world * n]
            plan = (n * 8, send, recv, send.data_ptr(), recv.data_ptr())
            self._plans[buffers] = plan
        nbytes, send, recv, send_ptr, recv_ptr = plan
        self.dm.pack_boundary(buffers, send_ptr, nbytes)
        self.dist.all_gather_into_tensor(recv, send)
        return self.dm.unpack_boundary(buffers, recv_ptr, nbytes, self.world)
