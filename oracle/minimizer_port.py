"""CPU port of the reference's minimizer step (energy+gradient assembly,
constraint projection, GD / per-row Polak-Ribiere CG direction, Armijo
backtracking line search) restricted to the hot-path scope.

TEST INFRASTRUCTURE ONLY (see oracle/ms_oracle.c).  Control flow is a plain
NumPy restatement of the reference routines cited on each function; the
per-facet arithmetic is the C oracle.  bench.py times this as the
``cpu_baseline`` ("port", 1 core).
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import ms_oracle as orc


@dataclass
class Problem:
    """Array view of what the hot path reads from the reference ``Mesh``."""

    positions: np.ndarray  # (nv,3) f64, row order = mesh.vertex_ids
    tri: np.ndarray  # (nf,3) int32, mesh.triangle_row_cache()
    gamma: np.ndarray | None = None  # (nf,) get_facet_parameter_array("surface_tension")
    kappa: np.ndarray | None = None  # (nv,) bending_params._per_vertex_params
    c0: np.ndarray | None = None
    is_boundary: np.ndarray | None = None  # (nv,) bool, mesh.boundary_vertex_ids
    fixed: np.ndarray | None = None  # (nv,) bool, mesh.fixed_mask
    tilts: np.ndarray | None = None  # (nv,3) mesh.tilts_view() (tilt / bending_tilt modules)
    tilt_fixed: np.ndarray | None = None  # (nv,) bool, vertex.tilt_fixed (minimizer_helpers.py:49-75)
    tilts_in: np.ndarray | None = None  # (nv,3) mesh.tilts_in_view() / tilts_out_view() (leaflet modules)
    tilts_out: np.ndarray | None = None
    tilt_fixed_in: np.ndarray | None = None  # vertex.tilt_fixed_in / tilt_fixed_out (minimizer.py:460-486)
    tilt_fixed_out: np.ndarray | None = None
    disk_rows_in: np.ndarray | None = None  # rows tagged tilt_disk_target_group_in == gp group (tilt_disk_target_in.py:137-145)
    disk_rows_out: np.ndarray | None = None
    energy_modules: list = field(default_factory=lambda: ["surface"])
    constraint_modules: list = field(default_factory=list)
    body_rows: np.ndarray | None = None  # None = all facets in one body
    target_volume: float | None = None
    gp: dict = field(default_factory=dict)

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float64).copy()
        self.tri = np.ascontiguousarray(self.tri, dtype=np.int32)
        nv, nf = self.positions.shape[0], self.tri.shape[0]
        if self.gamma is None:
            self.gamma = np.full(nf, float(self.gp.get("surface_tension", 1.0)))
        if self.kappa is None:
            self.kappa = np.full(nv, float(self.gp.get("bending_modulus", 0.0) or 0.0))
        if self.c0 is None:
            val = self.gp.get("spontaneous_curvature")
            if val is None:
                val = self.gp.get("intrinsic_curvature", 0.0)
            self.c0 = np.full(nv, float(val or 0.0))
        if self.is_boundary is None:
            self.is_boundary = np.zeros(nv, dtype=bool)
        if self.fixed is None:
            self.fixed = np.zeros(nv, dtype=bool)
        self.is_boundary = np.asarray(self.is_boundary, dtype=bool)
        self.fixed = np.asarray(self.fixed, dtype=bool)
        if self.tilts is not None:
            self.tilts = np.ascontiguousarray(self.tilts, dtype=np.float64).copy()
        if self.tilt_fixed is None:
            self.tilt_fixed = np.zeros(nv, dtype=bool)
        self.tilt_fixed = np.asarray(self.tilt_fixed, dtype=bool)
        for name in ("tilts_in", "tilts_out"):
            arr = getattr(self, name)
            if arr is not None:
                setattr(self, name, np.ascontiguousarray(arr, dtype=np.float64).copy())
        for name in ("tilt_fixed_in", "tilt_fixed_out"):
            arr = getattr(self, name)
            setattr(self, name, np.zeros(nv, dtype=bool) if arr is None else np.asarray(arr, dtype=bool))

    # -- parameter helpers (modules/energy/bending_params.py:19-33) ----------
    @property
    def model(self) -> str:
        m = str(self.gp.get("bending_energy_model", "helfrich") or "helfrich").lower().strip()
        return "helfrich" if m == "helfrich" else "willmore"

    @property
    def grad_mode(self) -> str:
        m = str(self.gp.get("bending_gradient_mode", "analytic") or "analytic").lower().strip()
        return "analytic" if m == "analytic" else "approx"

    @property
    def volume_mode(self) -> str:
        return self.gp.get("volume_constraint_mode", "lagrange")


# ---------------------------------------------------------------------------
# runtime/evaluation_manager.py:134-151 + runtime/minimizer.py:941-992
# ---------------------------------------------------------------------------
def energy_and_gradient(p: Problem, pos: np.ndarray):
    grad = np.zeros_like(pos)
    E = 0.0
    for name in p.energy_modules:
        if name == "surface":
            E += orc.surface_energy_and_gradient(pos, p.tri, p.gamma, grad)
        elif name == "bending":
            E += orc.bending_energy_and_gradient(
                pos, p.tri, p.kappa, p.c0, p.is_boundary,
                model=p.model, mode=p.grad_mode, grad=grad,
            )
        elif name == "volume":
            # modules/energy/volume.py:94-128 (penalty mode only)
            if p.volume_mode == "penalty":
                k = float(p.gp.get("volume_stiffness", 1000.0))
                V = orc.volume(pos, p.tri, p.body_rows)
                delta = V - float(p.target_volume)
                E += 0.5 * k * delta**2
                orc.volume_gradient(pos, p.tri, grad, factor=k * delta, body_rows=p.body_rows)
        elif name == "tilt":
            # modules/energy/tilt.py:99-172
            k_t = float(p.gp.get("tilt_rigidity", 0.0) or 0.0)
            if k_t != 0.0:
                E += orc.tilt_energy_and_gradient(pos, p.tilts, p.tri, k_t, grad, None)
        elif name == "bending_tilt":
            # modules/energy/bending_tilt.py:151-482 (always the Helfrich form, :212-215)
            E += orc.bending_tilt_energy_and_gradient(pos, p.tilts, p.tri, p.kappa, p.c0, p.is_boundary,
                                                      mode=p.grad_mode, grad=grad)
        elif name == "tilt_smoothness":
            E += _smoothness(p, pos, p.tilts)
        elif name in ("tilt_in", "tilt_out"):
            lf = name[5:]
            E += orc.tilt_leaflet_energy_and_gradient(pos, leaflet_tilts(p, lf), p.tri, tilt_modulus(p, lf),
                                                      tilt_mass_mode(p, lf), grad, None)
        elif name in ("tilt_smoothness_in", "tilt_smoothness_out"):
            lf = name[16:]
            E += _smoothness_leaflet(p, pos, leaflet_tilts(p, lf), lf)
        elif name in ("bending_tilt_in", "bending_tilt_out"):
            lf = name[13:]
            E += _bending_tilt_leaflet(p, pos, leaflet_tilts(p, lf), lf, grad=grad)
        elif name in ("tilt_disk_target_in", "tilt_disk_target_out"):
            lf = name[17:]
            E += _disk_target(p, pos, leaflet_tilts(p, lf), lf, grad=grad)
        elif name == "rim_slope_match_out":
            # modules/energy/rim_slope_match_out.py:376-382: 0.0 while rim_slope_match_strength is 0 (the caveolin
            # decks load the module with the strength switched off; anything else is outside the hot path)
            if float(p.gp.get("rim_slope_match_strength", 0.0) or 0.0) != 0.0:
                raise ValueError("rim_slope_match_out with a non-zero strength is outside the hot-path scope")
        else:
            raise ValueError(f"module {name!r} is outside the hot-path scope")
    # constraint_manager.apply_gradient_modifications_array (k == 1 dense branch :293-301)
    if "volume" in p.constraint_modules and p.volume_mode == "lagrange" and p.target_volume is not None:
        gC = np.zeros_like(pos)
        orc.volume_gradient(pos, p.tri, gC, factor=1.0, body_rows=p.body_rows)
        norm_sq = float(np.sum(gC * gC))
        if norm_sq > 1e-18:
            lam = float(np.sum(grad * gC)) / norm_sq
            grad -= lam * gC
    # minimizer.py:988-990
    grad[p.fixed] = 0.0
    return float(E), grad


# runtime/evaluation_manager.py:184-225 compute_energy_array_total
def energy_total(p: Problem, pos: np.ndarray, tilts=None, tilts_in=None, tilts_out=None) -> float:
    E = 0.0
    if tilts is None:
        tilts = p.tilts
    lt = {"in": p.tilts_in if tilts_in is None else tilts_in, "out": p.tilts_out if tilts_out is None else tilts_out}
    for name in p.energy_modules:
        if name == "surface":
            # no compute_energy_array -> gradient API into a scratch (:201-210)
            E += orc.surface_energy_and_gradient(pos, p.tri, p.gamma, None)
        elif name == "bending":
            E += orc.bending_energy(pos, p.tri, p.kappa, p.c0, p.is_boundary, model=p.model)
        elif name == "volume":
            if p.volume_mode == "penalty":
                k = float(p.gp.get("volume_stiffness", 1000.0))
                V = orc.volume(pos, p.tri, p.body_rows)
                E += 0.5 * k * (V - float(p.target_volume)) ** 2
        elif name == "tilt":
            k_t = float(p.gp.get("tilt_rigidity", 0.0) or 0.0)
            if k_t != 0.0:
                E += orc.tilt_energy_and_gradient(pos, tilts, p.tri, k_t, None, None)
        elif name == "bending_tilt":
            E += orc.bending_tilt_energy_and_gradient(pos, tilts, p.tri, p.kappa, p.c0, p.is_boundary)
        elif name == "tilt_smoothness":
            E += _smoothness(p, pos, tilts)
        elif name in ("tilt_in", "tilt_out"):
            lf = name[5:]
            E += orc.tilt_leaflet_energy_and_gradient(pos, lt[lf], p.tri, tilt_modulus(p, lf), tilt_mass_mode(p, lf))
        elif name in ("tilt_smoothness_in", "tilt_smoothness_out"):
            lf = name[16:]
            E += _smoothness_leaflet(p, pos, lt[lf], lf)
        elif name in ("bending_tilt_in", "bending_tilt_out"):
            lf = name[13:]
            E += _bending_tilt_leaflet(p, pos, lt[lf], lf)
        elif name in ("tilt_disk_target_in", "tilt_disk_target_out"):
            lf = name[17:]
            E += _disk_target(p, pos, lt[lf], lf)
        elif name == "rim_slope_match_out":  # switched off (strength 0): see energy_and_gradient
            if float(p.gp.get("rim_slope_match_strength", 0.0) or 0.0) != 0.0:
                raise ValueError("rim_slope_match_out with a non-zero strength is outside the hot-path scope")
        else:
            raise ValueError(name)
    return float(E)


def _smoothness(p: Problem, pos, tilts, tilt_grad=None) -> float:
    """modules/energy/tilt_smoothness.py:246-283; cotans of the positions being evaluated."""
    k_s = float(p.gp.get("tilt_smoothness_rigidity", 0.0) or 0.0)
    if k_s == 0.0:
        return 0.0
    return orc.tilt_smoothness_energy_and_gradient(pos, tilts, p.tri, k_s, tilt_grad)


TILT_MODULES = ("tilt", "bending_tilt", "tilt_smoothness")  # modules with USES_TILT = True in scope
LEAFLET_MODULES = ("tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out",
                   "bending_tilt_in", "bending_tilt_out", "tilt_disk_target_in", "tilt_disk_target_out")  # USES_TILT_LEAFLETS


def disk_target_params(p: Problem, leaflet: str):
    """modules/energy/tilt_disk_target_in.py:38-134 -> dict or None when the module contributes nothing."""
    gp = p.gp

    def pick(name):
        v = gp.get(f"tilt_disk_target_{name}_{leaflet}")
        return gp.get(f"tilt_disk_target_{name}") if v is None else v

    group = gp.get(f"tilt_disk_target_group_{leaflet}")
    if group is None or not str(group).strip():
        return None
    k = float(gp.get(f"tilt_disk_target_strength_{leaflet}") or 0.0)
    theta_b = float(pick("theta_B") or 0.0)
    rows = p.disk_rows_in if leaflet == "in" else p.disk_rows_out
    if k == 0.0 or theta_b == 0.0 or rows is None or len(rows) == 0:
        return None
    center = pick("center")
    normal = pick("normal")
    if normal is None:
        raise NotImplementedError("tilt_disk_target: the SVD plane fit (no tilt_disk_target_normal) is not restated")
    radius = pick("radius")
    try:
        radius = float(radius) if radius is not None and float(radius) > 0.0 else None
    except (TypeError, ValueError):
        radius = None
    lam = pick("lambda")
    if lam is not None:
        try:
            lam = float(lam)
        except (TypeError, ValueError):
            lam = 0.0
    else:
        kt = gp.get(f"tilt_modulus_{leaflet}")
        if kt is None and leaflet == "in":
            kt = gp.get("tilt_modolus_in")
        kap = gp.get(f"bending_modulus_{leaflet}")
        if kap is None:
            kap = gp.get("bending_modulus")
        lam = 0.0
        if kt is not None and kap is not None and float(kt) > 0.0 and float(kap) > 0.0:
            lam = float(np.sqrt(float(kt) / float(kap)))
    return {"disk_rows": np.asarray(rows, dtype=int), "k_target": k, "theta_b": theta_b, "lam": lam,
            "center": [0.0, 0.0, 0.0] if center is None else center, "normal": normal, "radius": radius}


def _disk_target(p: Problem, pos, tilts, leaflet: str, grad=None, tilt_grad=None) -> float:
    prm = disk_target_params(p, leaflet)
    if prm is None:
        return 0.0
    return orc.tilt_disk_target_energy_and_gradient(pos, tilts, p.tri, grad=grad, tilt_grad=tilt_grad, **prm)


def leaflet_bending_params(p: Problem, leaflet: str):
    """modules/energy/bt_params.py:225-318 without per-vertex option overrides: (kappa, c0) arrays."""
    k = p.gp.get(f"bending_modulus_{leaflet}")
    if k is None:
        k = p.gp.get("bending_modulus", 0.0)
    c0 = p.gp.get(f"spontaneous_curvature_{leaflet}")
    if c0 is None:
        c0 = p.gp.get("spontaneous_curvature")
        if c0 is None:
            c0 = p.gp.get("intrinsic_curvature", 0.0)
    nv = p.positions.shape[0]
    return np.full(nv, float(k or 0.0)), np.full(nv, float(c0 or 0.0))


def _bending_tilt_leaflet(p: Problem, pos, tilts, leaflet: str, grad=None, tilt_grad=None) -> float:
    """bending_tilt_in.py:46 (div_sign = -1) / bending_tilt_out.py (+1)."""
    kappa, c0 = leaflet_bending_params(p, leaflet)
    return orc.bending_tilt_leaflet_energy_and_gradient(pos, tilts, p.tri, kappa, c0, p.is_boundary,
                                                        -1.0 if leaflet == "in" else 1.0, mode=p.grad_mode,
                                                        grad=grad, tilt_grad=tilt_grad)


def leaflet_tilts(p: Problem, leaflet: str) -> np.ndarray:
    t = p.tilts_in if leaflet == "in" else p.tilts_out
    return np.zeros_like(p.positions) if t is None else t  # Mesh.tilts_in_view() is zeros until set


def tilt_modulus(p: Problem, leaflet: str) -> float:
    """modules/energy/tilt_params.py:6-12."""
    k = p.gp.get(f"tilt_modulus_{leaflet}")
    if k is None:
        k = p.gp.get(f"tilt_modolus_{leaflet}")
    return float(k or 0.0)


def tilt_mass_mode(p: Problem, leaflet: str) -> str:
    """modules/energy/tilt_params.py:15-23."""
    mode = p.gp.get(f"tilt_mass_mode_{leaflet}")
    if mode is None:
        mode = p.gp.get("tilt_mass_mode")
    txt = str(mode or "lumped").strip().lower()
    if txt not in {"lumped", "consistent"}:
        raise ValueError(f"tilt_mass_mode_{leaflet} must be 'lumped' or 'consistent'.")
    return txt


def smoothness_rigidity(p: Problem, leaflet: str) -> float:
    """modules/energy/tilt_smoothness_utils.py:95-100."""
    k = p.gp.get(f"bending_modulus_{leaflet}")
    if k is None:
        k = p.gp.get("bending_modulus")
    return float(k or 0.0)


def _smoothness_leaflet(p: Problem, pos, tilts, leaflet: str, tilt_grad=None) -> float:
    """modules/energy/tilt_smoothness_leaflet.py:17-79 (no absent-leaflet mask): the base smoothness
    energy with the leaflet's rigidity."""
    k_s = smoothness_rigidity(p, leaflet)
    if k_s == 0.0:
        return 0.0
    return orc.tilt_smoothness_energy_and_gradient(pos, tilts, p.tri, k_s, tilt_grad)


# runtime/evaluation_manager.py:630-742 with tilt_vertex_areas given (the relaxation's call): the
# magnitude modules take the vertex-area form, whatever their mass mode (:663-695)
def energy_and_leaflet_tilt_gradients(p: Problem, pos, tilts_in, tilts_out, vertex_areas):
    tg = {"in": np.zeros_like(pos), "out": np.zeros_like(pos)}
    lt = {"in": tilts_in, "out": tilts_out}
    E = 0.0
    for name in p.energy_modules:
        if name in ("tilt_in", "tilt_out"):
            lf = name[5:]
            k = float(p.gp.get(f"tilt_modulus_{lf}") or 0.0)
            if k != 0.0:
                sq = np.einsum("ij,ij->i", lt[lf], lt[lf])
                E += float(0.5 * k * np.sum(sq * vertex_areas))
                tg[lf] += k * lt[lf] * vertex_areas[:, None]
        elif name in ("tilt_smoothness_in", "tilt_smoothness_out"):
            lf = name[16:]
            E += _smoothness_leaflet(p, pos, lt[lf], lf, tg[lf])
        elif name in ("bending_tilt_in", "bending_tilt_out"):
            lf = name[13:]
            E += _bending_tilt_leaflet(p, pos, lt[lf], lf, tilt_grad=tg[lf])
        elif name in ("tilt_disk_target_in", "tilt_disk_target_out"):
            lf = name[17:]
            E += _disk_target(p, pos, lt[lf], lf, tilt_grad=tg[lf])
    return float(E), tg["in"], tg["out"]


# runtime/evaluation_manager.py:537-628 (same fast path for the magnitude modules, :558-581)
def tilt_dependent_energy_leaflets(p: Problem, pos, tilts_in, tilts_out, vertex_areas) -> float:
    lt = {"in": tilts_in, "out": tilts_out}
    E = 0.0
    for name in p.energy_modules:
        if name in ("tilt_in", "tilt_out"):
            lf = name[5:]
            k = float(p.gp.get(f"tilt_modulus_{lf}") or 0.0)
            if k != 0.0:
                sq = np.einsum("ij,ij->i", lt[lf], lt[lf])
                E += float(0.5 * k * np.sum(sq * vertex_areas))
        elif name in ("tilt_smoothness_in", "tilt_smoothness_out"):
            lf = name[16:]
            E += _smoothness_leaflet(p, pos, lt[lf], lf)
        elif name in ("bending_tilt_in", "bending_tilt_out"):
            lf = name[13:]
            E += _bending_tilt_leaflet(p, pos, lt[lf], lf)
        elif name in ("tilt_disk_target_in", "tilt_disk_target_out"):
            lf = name[17:]
            E += _disk_target(p, pos, lt[lf], lf)
    return float(E)


# runtime/preconditioners.py:62-146
def leaflet_tilt_cg_preconditioner(p: Problem, pos, fixed_in, fixed_out, vertex_areas):
    nv = pos.shape[0]
    diag = {"in": np.zeros(nv), "out": np.zeros(nv)}
    for lf in ("in", "out"):
        k = float(p.gp.get(f"tilt_modulus_{lf}") or 0.0)
        if k != 0.0:
            diag[lf] += k * vertex_areas
    ks = {lf: float(p.gp.get(f"bending_modulus_{lf}") or p.gp.get("bending_modulus") or 0.0) for lf in ("in", "out")}
    if (ks["in"] != 0.0 or ks["out"] != 0.0) and p.tri.shape[0]:
        _k, _a, w = orc.compute_curvature_data(pos, p.tri)
        for lf in ("in", "out"):
            if ks[lf] != 0.0:
                f = 0.5 * ks[lf]
                np.add.at(diag[lf], p.tri[:, 0], f * (w[:, 1] + w[:, 2]))
                np.add.at(diag[lf], p.tri[:, 1], f * (w[:, 2] + w[:, 0]))
                np.add.at(diag[lf], p.tri[:, 2], f * (w[:, 0] + w[:, 1]))
    out = []
    for lf, fx in (("in", fixed_in), ("out", fixed_out)):
        d = np.where(diag[lf] > 1e-12, diag[lf], 1.0)
        d[fx] = 1.0
        out.append(1.0 / d)
    return out[0], out[1]


# runtime/steppers/tilt_relaxation.py:426-1478 relax_leaflet_tilts, default options (per-step projection
# refresh without tilt constraints = projecting already projected fields; no update modes, no fallback)
def relax_leaflet_tilts(p: Problem, pos: np.ndarray) -> dict:
    stats = {"iters": 0, "evals": 0}
    mode = str(p.gp.get("tilt_solve_mode", "fixed") or "").strip().lower()
    if mode in ("", "none", "off", "false", "fixed") or mode not in ("nested", "coupled"):
        return stats
    step_size = float(p.gp.get("tilt_step_size", 0.0) or 0.0)
    if step_size <= 0.0:
        return stats
    tol = float(p.gp.get("tilt_tol", 0.0) or 0.0)
    if tol <= 0.0:
        tol = 0.0
    if mode == "nested":
        n_inner = int(p.gp.get("tilt_inner_steps", 0) or 0)
    else:
        n_inner = int(p.gp.get("tilt_coupled_steps", p.gp.get("tilt_inner_steps", 0)) or 0)
    if n_inner <= 0:
        return stats
    solver = str(p.gp.get("tilt_solver", "cg") or "cg").strip().lower()
    if solver not in ("gd", "cg"):
        solver = "gd"
    if solver == "cg":
        max_iters = int(p.gp.get("tilt_cg_max_iters", n_inner) or 0)
        if max_iters <= 0:
            return stats
    else:
        max_iters = n_inner
    fx_in, fx_out = p.tilt_fixed_in, p.tilt_fixed_out
    if not (np.any(~fx_in) or np.any(~fx_out)):
        return stats
    normals = unit_vertex_normals(pos, p.tri)

    def project(t):
        return t - np.einsum("ij,ij->i", t, normals)[:, None] * normals

    t_in = project(leaflet_tilts(p, "in").copy())
    t_out = project(leaflet_tilts(p, "out").copy())
    fv_in = t_in[fx_in].copy() if np.any(fx_in) else None
    fv_out = t_out[fx_out].copy() if np.any(fx_out) else None
    va = orc.barycentric_vertex_areas(pos, p.tri)

    def trial_of(step, d_in, d_out):  # projections/tilt.py:99-138
        a = project(t_in + step * d_in)
        b = project(t_out + step * d_out)
        if fv_in is not None:
            a[fx_in] = fv_in
        if fv_out is not None:
            b[fx_out] = fv_out
        return a, b

    def grad_at():
        E, gi, go = energy_and_leaflet_tilt_gradients(p, pos, t_in, t_out, va)
        stats["evals"] += 1
        gi[fx_in] = 0.0
        go[fx_out] = 0.0
        gnorm = float(np.sqrt(np.sum(gi[~fx_in] ** 2) + np.sum(go[~fx_out] ** 2)))
        return E, gi, go, gnorm

    def search(E0, sign, d_in, d_out):
        step = step_size
        for _bt in range(12):
            a, b = trial_of(sign * step, d_in, d_out)
            E1 = tilt_dependent_energy_leaflets(p, pos, a, b, va)
            stats["evals"] += 1
            if E1 <= E0:
                return True, a, b, E1
            step *= 0.5
            if step < 1e-16:
                break
        return False, None, None, E0

    if solver == "gd":
        for _ in range(max_iters):
            E0, gi, go, gnorm = grad_at()
            if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                break
            ok, a, b, _E1 = search(E0, -1.0, gi, go)
            stats["iters"] += 1
            if not ok:
                break
            t_in, t_out = a, b
    else:
        pre = str(p.gp.get("tilt_cg_preconditioner", "jacobi") or "jacobi").strip().lower()
        if pre in ("none", "off", "false"):
            pre = None
        E0, gi, go, gnorm = grad_at()
        if not (gnorm == 0.0 or (tol > 0.0 and gnorm < tol)):
            Mi = Mo = None
            if pre == "jacobi":
                Mi, Mo = leaflet_tilt_cg_preconditioner(p, pos, fx_in, fx_out, va)
            r_in, r_out = -gi, -go
            z_in = r_in * Mi[:, None] if Mi is not None else r_in
            z_out = r_out * Mo[:, None] if Mo is not None else r_out
            d_in, d_out = z_in.copy(), z_out.copy()
            rz_old = float(np.sum(r_in * z_in) + np.sum(r_out * z_out))
            for _ in range(max_iters):
                if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                    break
                ok, a, b, E1 = search(E0, 1.0, d_in, d_out)
                stats["iters"] += 1
                if not ok:
                    break
                t_in, t_out, E0 = a, b, E1
                E0, gi, go, gnorm = grad_at()
                if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                    break
                r_in, r_out = -gi, -go
                z_in = r_in * Mi[:, None] if Mi is not None else r_in
                z_out = r_out * Mo[:, None] if Mo is not None else r_out
                rz_new = float(np.sum(r_in * z_in) + np.sum(r_out * z_out))
                if rz_old == 0.0:
                    break
                beta = rz_new / rz_old
                d_in = z_in + beta * d_in
                d_out = z_out + beta * d_out
                rz_old = rz_new
    p.tilts_in, p.tilts_out = t_in, t_out
    return stats


# runtime/evaluation_manager.py:386-462 (energy of the USES_TILT modules + dense tilt gradient)
def energy_and_tilt_gradient(p: Problem, pos: np.ndarray, tilts: np.ndarray):
    tg = np.zeros_like(pos)
    E = 0.0
    for name in p.energy_modules:
        if name == "tilt":
            k_t = float(p.gp.get("tilt_rigidity", 0.0) or 0.0)
            if k_t != 0.0:
                E += orc.tilt_energy_and_gradient(pos, tilts, p.tri, k_t, None, tg)
        elif name == "bending_tilt":
            E += orc.bending_tilt_energy_and_gradient(pos, tilts, p.tri, p.kappa, p.c0, p.is_boundary,
                                                      tilt_grad=tg)
        elif name == "tilt_smoothness":
            E += _smoothness(p, pos, tilts, tg)
    return float(E), tg


# runtime/evaluation_manager.py:303-384 compute_energy_array_with_tilts
def tilt_dependent_energy(p: Problem, pos: np.ndarray, tilts: np.ndarray) -> float:
    E = 0.0
    for name in p.energy_modules:
        if name == "tilt":
            k_t = float(p.gp.get("tilt_rigidity", 0.0) or 0.0)
            if k_t != 0.0:
                E += orc.tilt_energy_and_gradient(pos, tilts, p.tri, k_t, None, None)
        elif name == "bending_tilt":
            E += orc.bending_tilt_energy_and_gradient(pos, tilts, p.tri, p.kappa, p.c0, p.is_boundary)
        elif name == "tilt_smoothness":
            E += _smoothness(p, pos, tilts)
    return float(E)


def unit_vertex_normals(pos: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """geometry/triangle_ops.py:55-73 (normalised where the length is >= 1e-12)."""
    tn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    normals = np.zeros_like(pos)
    np.add.at(normals, tri[:, 0], tn)
    np.add.at(normals, tri[:, 1], tn)
    np.add.at(normals, tri[:, 2], tn)
    lens = np.linalg.norm(normals, axis=1)
    mask = lens >= 1e-12
    normals[mask] /= lens[mask][:, None]
    return normals


# runtime/preconditioners.py:15-59 (tilt_rigidity term only; tilt_smoothness is out of scope)
def tilt_cg_preconditioner(p: Problem, pos: np.ndarray, fixed_mask: np.ndarray) -> np.ndarray:
    nv = pos.shape[0]
    diag = np.zeros(nv)
    k_t = float(p.gp.get("tilt_rigidity", 0.0) or 0.0)
    if k_t != 0.0 and p.tri.shape[0]:
        tri = p.tri
        n = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
        thirds = 0.5 * np.linalg.norm(n, axis=1) / 3.0
        va = np.zeros(nv)
        for kcol in range(3):
            np.add.at(va, tri[:, kcol], thirds)
        diag += k_t * va
    k_s = float(p.gp.get("tilt_smoothness_rigidity", 0.0) or 0.0)
    if k_s != 0.0 and p.tri.shape[0]:  # :42-57
        _k, _a, w = orc.compute_curvature_data(pos, p.tri)
        f = 0.5 * k_s
        np.add.at(diag, p.tri[:, 0], f * (w[:, 1] + w[:, 2]))
        np.add.at(diag, p.tri[:, 1], f * (w[:, 2] + w[:, 0]))
        np.add.at(diag, p.tri[:, 2], f * (w[:, 0] + w[:, 1]))
    diag = np.where(diag > 1e-12, diag, 1.0)
    diag[fixed_mask] = 1.0
    return 1.0 / diag


# runtime/steppers/tilt_relaxation.py:237-424 relax_tilts (positions frozen)
def relax_tilts(p: Problem, pos: np.ndarray) -> dict:
    stats = {"iters": 0, "evals": 0}
    mode = str(p.gp.get("tilt_solve_mode", "fixed") or "").strip().lower()
    if mode in ("", "none", "off", "false", "fixed") or mode not in ("nested", "coupled"):
        return stats
    step_size = float(p.gp.get("tilt_step_size", 0.0) or 0.0)
    if step_size <= 0.0:
        return stats
    tol = float(p.gp.get("tilt_tol", 0.0) or 0.0)
    if tol <= 0.0:
        tol = 0.0
    if mode == "nested":
        n_inner = int(p.gp.get("tilt_inner_steps", 0) or 0)
    else:
        n_inner = int(p.gp.get("tilt_coupled_steps", p.gp.get("tilt_inner_steps", 0)) or 0)
    if n_inner <= 0:
        return stats
    solver = str(p.gp.get("tilt_solver", "cg") or "cg").strip().lower()
    if solver not in ("gd", "cg"):
        solver = "gd"
    if solver == "cg":
        max_iters = int(p.gp.get("tilt_cg_max_iters", n_inner) or 0)
        if max_iters <= 0:
            return stats
    else:
        max_iters = n_inner
    fixed = p.tilt_fixed
    if not np.any(~fixed):
        return stats
    normals = unit_vertex_normals(pos, p.tri)

    def project(t):
        return t - np.einsum("ij,ij->i", t, normals)[:, None] * normals

    tilts = project(p.tilts.copy())
    fixed_vals = tilts[fixed].copy() if np.any(fixed) else None

    def trial_of(base, step, direction):
        t = project(base + step * direction)
        if fixed_vals is not None:
            t[fixed] = fixed_vals
        return t

    def grad_at(t):
        E, tg = energy_and_tilt_gradient(p, pos, t)
        stats["evals"] += 1
        tg[fixed] = 0.0
        return E, tg, float(np.linalg.norm(tg[~fixed]))

    if solver == "gd":
        for _ in range(max_iters):
            E0, tg, gnorm = grad_at(tilts)
            if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                break
            step, accepted = step_size, False
            for _bt in range(12):
                trial = trial_of(tilts, -step, tg)
                E1 = tilt_dependent_energy(p, pos, trial)
                stats["evals"] += 1
                if E1 <= E0:
                    tilts, accepted = trial, True
                    break
                step *= 0.5
                if step < 1e-16:
                    break
            stats["iters"] += 1
            if not accepted:
                break
    else:
        pre = str(p.gp.get("tilt_cg_preconditioner", "jacobi") or "jacobi").strip().lower()
        M_inv = tilt_cg_preconditioner(p, pos, fixed) if pre == "jacobi" else None
        E0, tg, gnorm = grad_at(tilts)
        if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
            p.tilts = tilts
            return stats
        residual = -tg
        z = residual * M_inv[:, None] if M_inv is not None else residual
        direction = z.copy()
        rz_old = float(np.sum(residual * z))
        for _ in range(max_iters):
            if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                break
            step, accepted = step_size, False
            for _bt in range(12):
                trial = trial_of(tilts, step, direction)
                E1 = tilt_dependent_energy(p, pos, trial)
                stats["evals"] += 1
                if E1 <= E0:
                    tilts, E0, accepted = trial, E1, True
                    break
                step *= 0.5
                if step < 1e-16:
                    break
            stats["iters"] += 1
            if not accepted:
                break
            E0, tg, gnorm = grad_at(tilts)
            if gnorm == 0.0 or (tol > 0.0 and gnorm < tol):
                break
            residual = -tg
            z = residual * M_inv[:, None] if M_inv is not None else residual
            rz_new = float(np.sum(residual * z))
            if rz_old == 0.0:
                break
            beta = rz_new / rz_old
            direction = z + beta * direction
            rz_old = rz_new
    p.tilts = tilts
    return stats


# geometry/mesh.py:788-814 project_tilts_to_tangent with the unit vertex normals of
# geometry/triangle_ops.py:55-73 (normalised where the length is >= 1e-12)
def projected_tilts(p: Problem, pos: np.ndarray):
    """tilts projected onto the vertex tangent planes of ``pos`` (not stored)."""
    if p.tilts is None or p.tri.shape[0] == 0:
        return p.tilts
    tri = p.tri
    tn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    normals = np.zeros_like(pos)
    np.add.at(normals, tri[:, 0], tn)
    np.add.at(normals, tri[:, 1], tn)
    np.add.at(normals, tri[:, 2], tn)
    lens = np.linalg.norm(normals, axis=1)
    mask = lens >= 1e-12
    normals[mask] /= lens[mask][:, None]
    dot = np.einsum("ij,ij->i", p.tilts, normals)
    return p.tilts - dot[:, None] * normals


def project_tilts_to_tangent(p: Problem, pos: np.ndarray) -> None:
    p.tilts = projected_tilts(p, pos)
    if (p.tilts_in is not None or p.tilts_out is not None) and p.tri.shape[0]:
        normals = unit_vertex_normals(pos, p.tri)
        for name in ("tilts_in", "tilts_out"):
            t = getattr(p, name)
            if t is not None:
                setattr(p, name, t - np.einsum("ij,ij->i", t, normals)[:, None] * normals)


# runtime/topology.py:174-199 (min over mesh edges == min over facet edges)
def min_edge_length(pos: np.ndarray, tri: np.ndarray) -> float:
    if tri.shape[0] == 0:
        return 0.0
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    m = min(
        float(np.min(np.linalg.norm(v1 - v0, axis=1))),
        float(np.min(np.linalg.norm(v2 - v1, axis=1))),
        float(np.min(np.linalg.norm(v0 - v2, axis=1))),
    )
    return m


# runtime/topology.py:13-48
def check_max_normal_change_positions(tri, old, new, limit_radians=0.5) -> bool:
    if tri.size == 0:
        return True
    n_old = np.cross(old[tri[:, 1]] - old[tri[:, 0]], old[tri[:, 2]] - old[tri[:, 0]])
    norms_old = np.linalg.norm(n_old, axis=1)
    good = norms_old > 1e-12
    if not np.any(good):
        return True
    n_old = n_old[good] / norms_old[good][:, None]
    tg = tri[good]
    n_new = np.cross(new[tg[:, 1]] - new[tg[:, 0]], new[tg[:, 2]] - new[tg[:, 0]])
    norms_new = np.linalg.norm(n_new, axis=1)
    if np.any(norms_new < 1e-12):
        return False
    n_new = n_new / norms_new[:, None]
    dots = np.clip(np.sum(n_old * n_new, axis=1), -1.0, 1.0)
    return bool(np.all(np.arccos(dots) <= limit_radians))


# modules/constraints/volume.py:69-149 enforce_constraint (projection loop)
def project_volume(p: Problem, pos: np.ndarray, tol=1e-12, max_iter=3, first_cached=False) -> np.ndarray:
    """modules/constraints/volume.py:69-149 as the minimizer reaches it.  ``Body`` caches the volume gradient of
    its last ``compute_volume_and_gradient`` evaluation (geometry/body.py:386-407, :463) while ``compute_volume``
    (:70-120) refreshes only the cached volume and mesh version.  An enforce that follows a ``compute_volume`` at the
    same mesh version -- the Lagrange drift check (minimizer.py:1492) -- therefore takes its FIRST step with the
    current volume but the gradient the previous enforce evaluated last (``first_cached``); later iterations follow
    a version bump and are fresh.  The cache lives on the problem (``p._vol_grad_cache``)."""
    for it in range(max_iter):
        V = orc.volume(pos, p.tri, p.body_rows)
        cache = getattr(p, "_vol_grad_cache", None)
        if it == 0 and first_cached and cache is not None:
            g = cache
        else:
            g = np.zeros_like(pos)
            orc.volume_gradient(pos, p.tri, g, factor=1.0, body_rows=p.body_rows)
            p._vol_grad_cache = g
        delta = V - float(p.target_volume)
        if abs(delta) < tol:
            break
        norm_sq = float(np.sum(g * g)) + 1e-12
        lam = delta / norm_sq
        pos = pos.copy()
        pos[~p.fixed] -= lam * g[~p.fixed]
    return pos


@dataclass
class LineSearchResult:
    success: bool
    next_step: float
    energy: float
    alpha: float
    trials: int


# runtime/steppers/line_search.py:267-541 backtracking_line_search_array
def line_search(p: Problem, direction, gradient, step_size, *, max_iter=10, beta=0.7,
                c=1e-4, gamma=1.5, alpha_max_factor=10.0, enforcer=None,
                array_trials=True) -> LineSearchResult:
    """Trial evaluation follows the line search's mesh-mutating path (line_search.py:428-487) whenever a
    tilt-reading module or a constraint enforcer is active: every trial is written into the mesh,
    ``energy_fn`` projects the STORED tilts onto the trial surface, and a rejected trial restores
    the positions but not the tilts (those are restored only with an enforcer / reduced energy,
    :300-312).  The array fast path (:357-421, taken when the stepper names ``trial_energy_fn``)
    gives the same numbers for the shape-only modules; for bending_tilt it reads a P1-gradient
    cache keyed to the mesh, i.e. stale for a trial array, so the consistent path is the one
    restated and pinned (oracle/gen_golden.py: run_tilt_trajectory)."""
    has_tilt = any(m in p.energy_modules for m in TILT_MODULES + LEAFLET_MODULES)
    array_trials = array_trials and enforcer is None and not has_tilt
    movable = ~p.fixed
    baseline = p.positions.copy()
    project_tilts_to_tangent(p, baseline)  # energy_fn projects first (minimizer.py:581-588)
    energy0 = energy_total(p, baseline)
    min_edge = min_edge_length(baseline, p.tri)
    safe_limit = 0.3 * min_edge if min_edge > 0 else float("inf")
    max_dir = float(np.max(np.linalg.norm(direction[movable], axis=1))) if movable.any() else 0.0
    g_dot_d = float(np.sum(gradient * direction))
    if g_dot_d >= 0.0:
        return LineSearchResult(False, step_size, energy0, 0.0, 0)
    alpha = step_size
    edge_fraction = float(p.gp.get("shape_step_edge_fraction", 0.0) or 0.0)
    if edge_fraction > 0.0 and min_edge > 0.0 and max_dir > 0.0:
        alpha = min(alpha, edge_fraction * min_edge / max_dir)
    alpha_max = alpha_max_factor * step_size
    trials = 0
    for _ in range(max_iter):
        safe_small = alpha * max_dir < safe_limit
        trial = baseline.copy()
        trial[movable] = baseline[movable] + alpha * direction[movable]
        if not safe_small:
            if not check_max_normal_change_positions(p.tri, baseline, trial):
                alpha *= beta
                if alpha < 1e-8:
                    break
                continue
        if enforcer is not None:
            trial = enforcer(trial)
        if array_trials:
            # vertex-tilt modules: the trial energy uses the tilts projected onto the TRIAL
            # surface's tangent planes, without storing them (minimizer.py:723-733)
            E_t = energy_total(p, trial, tilts=projected_tilts(p, trial) if has_tilt else None)
        else:
            tilts_before = None if p.tilts is None else p.tilts.copy()
            leaf_before = (None if p.tilts_in is None else p.tilts_in.copy(),
                           None if p.tilts_out is None else p.tilts_out.copy())
            p.positions = trial
            if has_tilt:
                project_tilts_to_tangent(p, trial)
            E_t = energy_total(p, trial)
        trials += 1
        if E_t <= energy0 + c * alpha * g_dot_d:
            p.positions = trial
            return LineSearchResult(True, min(alpha * gamma, alpha_max), float(E_t), alpha, trials)
        if not array_trials:
            p.positions = baseline
            if enforcer is not None and tilts_before is not None:
                p.tilts = tilts_before  # needs_tilt_restore (line_search.py:300-312)
            if enforcer is not None:
                p.tilts_in, p.tilts_out = leaf_before
        alpha *= beta
        if alpha < 1e-8:
            break
    reduced = max(alpha * beta, 0.0)
    return LineSearchResult(False, max(reduced, step_size * beta), float(energy0), alpha, trials)


class GradientDescent:
    """runtime/steppers/gradient_descent.py:35-84."""

    def __init__(self, max_iter=10, beta=0.7, c=1e-4, gamma=1.5, alpha_max_factor=10.0):
        self.max_iter, self.beta, self.c, self.gamma, self.alpha_max_factor = (
            max_iter, beta, c, gamma, alpha_max_factor)
        self.last_direction = None

    def reset(self):
        pass

    def step(self, p: Problem, grad, step_size, enforcer=None) -> LineSearchResult:
        max_iter = int(p.gp.get("shape_line_search_max_iter", self.max_iter) or self.max_iter)
        direction = -grad
        self.last_direction = direction
        return line_search(p, direction, grad, step_size, max_iter=max_iter, beta=self.beta,
                           c=self.c, gamma=self.gamma, alpha_max_factor=self.alpha_max_factor,
                           enforcer=enforcer)


class ConjugateGradient:
    """runtime/steppers/conjugate_gradient.py:17-119 (per-row Polak-Ribiere)."""

    def __init__(self, restart_interval=10, max_iter=10, beta=0.7, c=1e-4, gamma=1.5,
                 alpha_max_factor=10.0, precondition=False):
        self.restart_interval = restart_interval
        self.precondition = precondition
        self.max_iter, self.beta, self.c, self.gamma, self.alpha_max_factor = (
            max_iter, beta, c, gamma, alpha_max_factor)
        self.reset()

    def reset(self):
        self.prev_grad = None
        self.prev_dir = None
        self.iter_count = 0
        self.last_direction = None

    def step(self, p: Problem, grad, step_size, enforcer=None) -> LineSearchResult:
        g = grad
        if self.precondition:  # :74-76 (history and the Armijo slope keep the raw gradient)
            g = g / (np.linalg.norm(g, axis=1)[:, None] + 1e-8)
        if self.prev_grad is None or self.iter_count % self.restart_interval == 0:
            direction = -g
        else:
            numer = np.einsum("ij,ij->i", g, g - self.prev_grad)
            denom = np.einsum("ij,ij->i", self.prev_grad, self.prev_grad) + 1e-20
            beta_pr = numer / denom
            direction = -g + beta_pr[:, None] * self.prev_dir
            reset_mask = beta_pr < 0
            if np.any(reset_mask):
                direction[reset_mask] = -g[reset_mask]
        direction[p.fixed] = 0.0
        self.last_direction = direction
        res = line_search(p, direction, grad, step_size, max_iter=self.max_iter, beta=self.beta,
                          c=self.c, gamma=self.gamma, alpha_max_factor=self.alpha_max_factor,
                          enforcer=enforcer)
        if res.success:
            self.prev_grad = grad.copy()
            self.prev_dir = direction.copy()
            self.iter_count += 1
        return res


# runtime/minimizer.py:1189-1535 Minimizer.minimize (shape path, tilt_solve_mode "fixed")
def minimize(p: Problem, stepper, n_steps: int, step_size: float = 1e-3, tol: float = 1e-6):
    has_enforceable = "volume" in p.constraint_modules
    proj_flag = bool(p.gp.get("volume_projection_during_minimization", True))
    vol_tol = float(p.gp.get("volume_tolerance", 1e-3))
    max_zero_steps = int(p.gp.get("max_zero_steps", 10))
    step_floor = float(p.gp.get("step_size_floor", 1e-8))
    step_mode = str(p.gp.get("step_size_mode", "adaptive") or "adaptive").lower()
    trace = []

    def enforce(pos, context, first_cached=False):
        # constraint_manager.enforce_all :843-905 (volume only)
        if context == "minimize" and not proj_flag:
            return pos
        if p.target_volume is None:
            return pos
        return project_volume(p, pos, max_iter=12 if context in ("finalize", "mesh_operation") else 3,
                              first_cached=first_cached)

    enforcer = (lambda pos: enforce(pos, "minimize")) if has_enforceable else None

    if has_enforceable:  # minimizer.py:1222-1225: enforce, then mesh.project_tilts_to_tangent()
        p.positions = enforce(p.positions, "mesh_operation")
        project_tilts_to_tangent(p, p.positions)

    zero_steps = 0
    step_success = True
    grad = None
    # True while the last thing that happened to the mesh version was the drift check's compute_volume WITHOUT a
    # projection: the finalize enforce then starts from Body's cached gradient (see project_volume)
    volume_cache_current = False
    for i in range(n_steps):
        if any(m in p.energy_modules for m in LEAFLET_MODULES):
            relax_leaflet_tilts(p, p.positions)  # minimizer.py:1240-1305 (guard factor 0)
        elif p.tilts is not None:
            relax_tilts(p, p.positions)  # minimizer.py:1237-1307 (single tilt field)
        E, grad = energy_and_gradient(p, p.positions)
        grad_norm = float(np.linalg.norm(grad))
        if grad_norm < tol:
            if has_enforceable:  # _finalize_constraints, minimizer.py:1177-1187
                p.positions = enforce(p.positions, "finalize", first_cached=volume_cache_current)
                project_tilts_to_tangent(p, p.positions)
            return {"energy": E, "gradient": grad, "step_success": True, "iterations": i + 1,
                    "terminated_early": True, "trace": trace, "step_size": step_size}
        fixed_step = float(p.gp.get("step_size", step_size) or step_size)
        step_in = fixed_step if step_mode == "fixed" else step_size
        res = stepper.step(p, grad, step_in, enforcer=enforcer)
        step_success, step_size = res.success, res.next_step
        trace.append({"E": E, "grad_norm": grad_norm, "success": res.success, "alpha": res.alpha,
                      "E_accepted": res.energy, "next_step": res.next_step, "trials": res.trials})
        project_tilts_to_tangent(p, p.positions)  # minimizer.py:1415 (+ increment_version)
        volume_cache_current = False
        if step_mode == "fixed":
            step_size = fixed_step
        if not step_success:
            if step_size <= step_floor:
                zero_steps += 1
                if zero_steps >= max_zero_steps:
                    return {"energy": energy_total(p, p.positions), "gradient": grad,
                            "step_success": False, "iterations": i + 1, "terminated_early": True,
                            "trace": trace, "step_size": step_size}
            else:
                zero_steps = 0
            stepper.reset()
        else:
            zero_steps = 0
            if p.volume_mode == "lagrange" and not proj_flag and p.target_volume is not None:
                V = orc.volume(p.positions, p.tri, p.body_rows)  # body.compute_volume: volume + version cached
                volume_cache_current = True
                rel = abs(V - p.target_volume) / max(abs(p.target_volume), 1.0)
                if rel > vol_tol:
                    if has_enforceable:  # minimizer.py:1505-1507
                        p.positions = enforce(p.positions, "mesh_operation", first_cached=True)
                        project_tilts_to_tangent(p, p.positions)
                        volume_cache_current = False  # increment_version
                    stepper.reset()
    if has_enforceable:
        p.positions = enforce(p.positions, "finalize", first_cached=volume_cache_current)
        project_tilts_to_tangent(p, p.positions)
    return {"energy": energy_total(p, p.positions), "gradient": grad, "step_success": step_success,
            "iterations": n_steps, "terminated_early": False, "trace": trace, "step_size": step_size}
