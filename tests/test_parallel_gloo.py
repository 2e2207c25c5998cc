"""N > 1 path on CPU: the sharded driver (membrane_solver_amd.parallel.ShardedStepper)
under torch.distributed/gloo with world_size 2, on a NumPy/oracle shard backend
that mimics the HIP backend's contract: after every local phase only the rows
this rank owns are valid (the others are poisoned with NaN), scalar partials
cover only this rank's facets/rows.  The run must reproduce the single-process
oracle minimisation, which proves the driver exchanges exactly what is needed.
"""

import os
import socket
import sys
import traceback

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleShardBackend:
    def __init__(self, P, T, *, rank, world, dist, modules, kappa, c0, gamma, target_volume=0.0,
                 volume_stiffness=0.0, fixed=None):
        import torch

        from membrane_solver_amd import _lib as L

        self.L, self.torch, self.dist = L, torch, dist
        self.rank, self.world = rank, world
        self.tri = np.ascontiguousarray(T, dtype=np.int32)
        self.nv, self.nf = P.shape[0], T.shape[0]
        self.rows = (self.nv + world - 1) // world
        self.nvp = self.rows * world
        self.r0, self.r1 = rank * self.rows, min(self.nv, (rank + 1) * self.rows)
        self.modules, self.target_volume, self.volume_stiffness = modules, target_volume, volume_stiffness
        self.kappa, self.c0, self.gamma = kappa, c0, gamma
        self.fixed = np.zeros(self.nv, bool) if fixed is None else fixed
        self.isb = np.zeros(self.nv, bool)
        z = lambda n: torch.zeros((self.nvp, n), dtype=torch.float64)  # noqa: E731
        self.buf = {L.MS_BUF_X: z(3), L.MS_BUF_XT: z(3), L.MS_BUF_G: z(3), L.MS_BUF_GC: z(3),
                    L.MS_BUF_D: z(3), L.MS_BUF_PG: z(3), L.MS_BUF_PD: z(3), L.MS_BUF_FK: z(3),
                    L.MS_BUF_FA: z(2)}
        self.buf[L.MS_BUF_X][: self.nv] = torch.from_numpy(P)
        self.scal = np.zeros(L.MS_NSCAL)
        self.own_facets = np.flatnonzero((self.tri[:, 0] >= self.r0) & (self.tri[:, 0] < self.r1))

    def _np(self, b):
        return self.buf[b].numpy()[: self.nv]

    def _poison(self, b):
        a = self.buf[b].numpy()
        a[: self.r0] = np.nan
        a[self.r1:] = np.nan

    # -- protocol ------------------------------------------------------------------
    def phase_energy(self, use_direction, alpha, write_trial, guard, write_bending_factors):
        from oracle import minimizer_port as mp
        from oracle import ms_oracle as orc

        L = self.L
        x = self._np(L.MS_BUF_X)
        X = x.copy()
        if use_direction:
            d = self._np(L.MS_BUF_D)
            assert np.all(np.isfinite(d)), "direction rows were not exchanged before a trial"
            X[~self.fixed] = x[~self.fixed] + alpha * d[~self.fixed]
        assert np.all(np.isfinite(X))
        tri_own = self.tri[self.own_facets]
        s = self.scal
        s[L.MS_S_ESURF] = orc.surface_energy_and_gradient(X, tri_own, self.gamma[self.own_facets], None)
        s[L.MS_S_VOL] = orc.volume(X, tri_own)
        s[L.MS_S_MINEDGE2] = mp.min_edge_length(X, tri_own) ** 2 if len(tri_own) else 1e300
        s[L.MS_S_GUARD] = 0.0
        if guard and use_direction and len(tri_own):
            s[L.MS_S_GUARD] = 0.0 if mp.check_max_normal_change_positions(tri_own, x, X) else 1.0
        s[L.MS_S_EBEND] = 0.0
        if self.modules & L.MS_MOD_BENDING:
            _E, fK, fAe, fAv = orc.bending_energy_and_gradient(X, self.tri, self.kappa, self.c0, self.isb,
                                                               want_factors=True)
            _Et, pv = orc.bending_energy(X, self.tri, self.kappa, self.c0, self.isb, per_vertex=True)
            s[L.MS_S_EBEND] = float(np.sum(pv[self.r0:self.r1]))
            if write_bending_factors:
                self._np(L.MS_BUF_FK)[...] = fK
                self._np(L.MS_BUF_FA)[...] = np.stack([fAe, fAv], axis=1)
                self._poison(L.MS_BUF_FK)
                self._poison(L.MS_BUF_FA)

    def phase_gradient(self):
        from oracle import ms_oracle as orc

        L = self.L
        x = self._np(L.MS_BUF_X)
        g = np.zeros_like(x)
        if self.modules & L.MS_MOD_SURFACE:
            orc.surface_energy_and_gradient(x, self.tri, self.gamma, g)
        if self.modules & L.MS_MOD_BENDING:
            fK, fA = self._np(L.MS_BUF_FK), self._np(L.MS_BUF_FA)
            assert np.all(np.isfinite(fK)) and np.all(np.isfinite(fA)), "factors were not exchanged"
            orc.bending_backprop(x, self.tri, self.isb, np.ascontiguousarray(fA[:, 0]),
                                 np.ascontiguousarray(fA[:, 1]), fK, g)
        if self.modules & L.MS_MOD_VOLUME_PENALTY:
            orc.volume_gradient(x, self.tri, g, factor=self.volume_stiffness * (self.scal[L.MS_S_VOL] - self.target_volume))
        self._np(L.MS_BUF_G)[...] = g
        self._poison(L.MS_BUF_G)
        self.scal[L.MS_S_GGC] = self.scal[L.MS_S_GCGC] = 0.0
        if self.modules & L.MS_CON_VOLUME:
            gC = np.zeros_like(x)
            orc.volume_gradient(x, self.tri, gC)
            self._np(L.MS_BUF_GC)[...] = gC
            self._poison(L.MS_BUF_GC)
            sl = slice(self.r0, self.r1)
            self.scal[L.MS_S_GGC] = float(np.sum(g[sl] * gC[sl]))
            self.scal[L.MS_S_GCGC] = float(np.sum(gC[sl] * gC[sl]))

    def phase_direction(self, stepper, use_history):
        L = self.L
        sl = slice(self.r0, self.r1)
        g = self._np(L.MS_BUF_G)[sl].copy()
        if (self.modules & L.MS_CON_VOLUME) and self.scal[L.MS_S_GCGC] > 1e-18:
            g -= (self.scal[L.MS_S_GGC] / self.scal[L.MS_S_GCGC]) * self._np(L.MS_BUF_GC)[sl]
        fx = self.fixed[sl]
        g[fx] = 0.0
        d = -g
        if stepper == L.MS_STEPPER_CG and use_history:
            pg, pd = self._np(L.MS_BUF_PG)[sl], self._np(L.MS_BUF_PD)[sl]
            beta = np.einsum("ij,ij->i", g, g - pg) / (np.einsum("ij,ij->i", pg, pg) + 1e-20)
            d = -g + beta[:, None] * pd
            d[beta < 0] = -g[beta < 0]
        d[fx] = 0.0
        self._np(L.MS_BUF_G)[sl] = g
        self._np(L.MS_BUF_D)[...] = np.nan
        self._np(L.MS_BUF_D)[sl] = d
        self.scal[L.MS_S_GNORM2] = float(np.sum(g * g))
        self.scal[L.MS_S_GDOTD] = float(np.sum(g * d))
        self.scal[L.MS_S_MAXD2] = float(np.max(np.sum(d[~fx] ** 2, axis=1))) if np.any(~fx) else 0.0

    def phase_commit_trial(self, alpha, keep_history):
        L = self.L
        x, d = self._np(L.MS_BUF_X), self._np(L.MS_BUF_D)
        assert np.all(np.isfinite(d))
        xt = x.copy()
        xt[~self.fixed] = x[~self.fixed] + alpha * d[~self.fixed]
        self._np(L.MS_BUF_X)[...] = xt
        if keep_history:
            b = self.buf
            b[L.MS_BUF_G], b[L.MS_BUF_PG] = b[L.MS_BUF_PG], b[L.MS_BUF_G]
            b[L.MS_BUF_D], b[L.MS_BUF_PD] = b[L.MS_BUF_PD], b[L.MS_BUF_D]

    def store_scalars(self, values):
        self.scal[...] = values

    def phase_gradient_direction(self, stepper, use_history):
        self.phase_gradient()
        self.phase_direction(stepper, use_history)

    def set_factors_valid(self, valid):
        pass

    def exchange(self, buffers):
        """Stand-in for the boundary exchange: owner rows of each buffer -> all ranks (a
        superset of the boundary rows), scalar headers -> (world, 16)."""
        for buffer_id in buffers:
            full = self.buf[buffer_id]
            mine = full[self.rank * self.rows:(self.rank + 1) * self.rows].clone()
            self.dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
        out = self.torch.empty(self.world * self.L.MS_NSCAL, dtype=self.torch.float64)
        self.dist.all_gather_into_tensor(out, self.torch.from_numpy(np.ascontiguousarray(self.scal)))
        return out.numpy().reshape(self.world, self.L.MS_NSCAL)


def _worker(rank, world, port, case, q):
    try:
        import torch.distributed as dist

        from membrane_solver_amd import _lib as L
        from membrane_solver_amd import meshgen
        from membrane_solver_amd.parallel import ShardedStepper
        from oracle import minimizer_port as mp

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        case, level = case.split(":")
        P, T = meshgen.icosphere(5)
        P = meshgen.smooth_displace(P, 0.08)
        nv, nf = P.shape[0], T.shape[0]
        kappa, c0, gamma = np.full(nv, 0.7), np.full(nv, 0.2), np.full(nf, 1.0)
        fixed = np.zeros(nv, bool)
        fixed[::17] = True
        V0 = 0.97 * float(np.einsum("ij,ij->i", np.cross(P[T[:, 1]], P[T[:, 2]]), P[T[:, 0]]).sum() / 6.0)
        if case == "cg_bending":  # no constraint row: fused gradient+direction, direction-only restarts
            modules = L.MS_MOD_SURFACE | L.MS_MOD_BENDING
            mods, cons, gp = ["surface", "bending"], [], {"volume_constraint_mode": "lagrange",
                                                          "volume_projection_during_minimization": False}
            stepper_id, ref_stepper = L.MS_STEPPER_CG, mp.ConjugateGradient()
        elif case == "cg_bending_volume":
            modules = L.MS_MOD_SURFACE | L.MS_MOD_BENDING | L.MS_CON_VOLUME
            mods, cons, gp = ["surface", "bending"], ["volume"], {"volume_constraint_mode": "lagrange",
                                                                  "volume_projection_during_minimization": False}
            stepper_id, ref_stepper = L.MS_STEPPER_CG, mp.ConjugateGradient()
        else:
            modules = L.MS_MOD_SURFACE | L.MS_MOD_VOLUME_PENALTY
            mods, cons, gp = ["surface", "volume"], [], {"volume_constraint_mode": "penalty", "volume_stiffness": 30.0}
            stepper_id, ref_stepper = L.MS_STEPPER_GD, mp.GradientDescent()
        be = OracleShardBackend(P, T, rank=rank, world=world, dist=dist, modules=modules, kappa=kappa, c0=c0,
                                gamma=gamma, target_volume=V0, volume_stiffness=30.0, fixed=fixed)
        drv = ShardedStepper(be, stepper=stepper_id, reuse_energy0=int(level))
        step, log = 2e-3, []
        for _ in range(7):
            r = drv.step(step, tol=1e-9)
            log.append((float(r.success), r.next_step, r.energy))
            step = r.next_step
            if not r.success:
                drv.reset()
        x_final = be._np(L.MS_BUF_X).copy()
        # single-process reference: the oracle's minimizer port, step by step
        p = mp.Problem(positions=P, tri=T, gamma=gamma, kappa=kappa, c0=c0, fixed=fixed, energy_modules=mods,
                       constraint_modules=cons, target_volume=V0, gp=dict(gp, bending_modulus=0.7))
        ref_log, step = [], 2e-3
        for _ in range(7):
            E, g = mp.energy_and_gradient(p, p.positions)
            res = ref_stepper.step(p, g, step)
            ref_log.append((float(res.success), res.next_step, res.energy))
            step = res.next_step
            if not res.success:
                ref_stepper.reset()
        got, want = np.array(log), np.array(ref_log)
        ok = (np.array_equal(got[:, 0], want[:, 0]) and np.allclose(got[:, 1], want[:, 1], rtol=1e-12)
              and np.allclose(got[:, 2], want[:, 2], rtol=1e-10)
              and np.max(np.abs(x_final - p.positions)) < 1e-9 and got[:, 0].sum() >= 2)
        q.put((rank, bool(ok), repr((got.tolist(), want.tolist())) if not ok else ""))
        dist.destroy_process_group()
    except Exception:
        q.put((rank, False, traceback.format_exc()))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("case", ["cg_bending_volume:0", "cg_bending_volume:2", "gd_surface_penalty:0",
                                  "gd_surface_penalty:2", "cg_bending:2"])
def test_sharded_driver_world2_matches_single_process(case):
    import torch.multiprocessing as tmp

    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"


def test_fold_scalars_is_rank_ordered():
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.parallel import fold_scalars

    a = np.zeros((3, L.MS_NSCAL))
    a[:, L.MS_S_ESURF] = [1.0, 2.0, 3.0]
    a[:, L.MS_S_MINEDGE2] = [5.0, 2.0, 9.0]
    a[:, L.MS_S_GUARD] = [0.0, 1.0, 0.0]
    a[:, L.MS_S_MAXD2] = [4.0, 1.0, 7.0]
    f = fold_scalars(a)
    assert f[L.MS_S_ESURF] == 6.0 and f[L.MS_S_MINEDGE2] == 2.0 and f[L.MS_S_GUARD] == 1.0 and f[L.MS_S_MAXD2] == 7.0
