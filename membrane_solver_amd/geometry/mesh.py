"""Array-native mesh + the device mirror protocol.

``ArrayMesh`` exposes exactly the SoA accessors the hot path reads from the
reference's ``Mesh`` (geometry/mesh.py): ``positions_view`` :372-389,
``triangle_row_cache`` :597-624, ``fixed_mask`` :210-232,
``boundary_vertex_ids`` :304-319, ``get_facet_parameter_array`` :234-265,
``vertex_ids`` / ``vertex_index_to_row``, ``increment_version`` :152 and the
version counters that gate every cache (:149-153).  The reference's own
``Mesh`` object satisfies the same protocol, so ``HipMirror`` accepts either.

``HipMirror`` owns the ``DeviceMesh``: connectivity is re-uploaded only when
``_facet_loops_version`` / ``_vertex_ids_version`` change, positions only when
``_version`` changes (SURVEY section 5 "Geometry caching").
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .. import _lib as L
from ..core.parameters import GlobalParameters
from ..device import DeviceMesh
from ..meshgen import boundary_mask_from_triangles


@dataclass
class ArrayBody:
    """geometry/body.py Body, reduced to what the path reads."""

    index: int = 0
    facet_rows: np.ndarray | None = None  # rows into tri_rows; None = every facet
    target_volume: float | None = None
    options: dict = field(default_factory=dict)


class ArrayMesh:
    def __init__(self, positions, tri_rows, *, fixed=None, surface_tension=None,
                 bending_modulus=None, spontaneous_curvature=None, bodies=None, tilts=None,
                 tilt_fixed=None, global_parameters=None, energy_modules=None, constraint_modules=None,
                 tilts_in=None, tilts_out=None, tilt_fixed_in=None, tilt_fixed_out=None,
                 disk_rows_in=None, disk_rows_out=None):
        self._positions = np.array(positions, dtype=np.float64, order="C", copy=True)
        self._tri_rows = np.ascontiguousarray(tri_rows, dtype=np.int32)
        nv, nf = self._positions.shape[0], self._tri_rows.shape[0]
        self.vertex_ids = np.arange(nv, dtype=np.int64)
        self._index_map = None
        self._fixed = np.zeros(nv, dtype=bool) if fixed is None else np.asarray(fixed, dtype=bool).copy()
        self._facet_params = {}
        if surface_tension is not None and np.ndim(surface_tension) > 0:
            self._facet_params["surface_tension"] = np.asarray(surface_tension, dtype=np.float64).copy()
        self._vertex_params = {}
        if bending_modulus is not None and np.ndim(bending_modulus) > 0:
            self._vertex_params["bending_modulus"] = np.asarray(bending_modulus, dtype=np.float64).copy()
        if spontaneous_curvature is not None and np.ndim(spontaneous_curvature) > 0:
            self._vertex_params["spontaneous_curvature"] = np.asarray(spontaneous_curvature, dtype=np.float64).copy()
        if global_parameters is None:
            global_parameters = GlobalParameters()
        elif isinstance(global_parameters, dict):
            global_parameters = GlobalParameters(global_parameters)
        self.global_parameters = global_parameters
        if surface_tension is not None and np.ndim(surface_tension) == 0:
            self.global_parameters.set("surface_tension", float(surface_tension))
        if bending_modulus is not None and np.ndim(bending_modulus) == 0:
            self.global_parameters.set("bending_modulus", float(bending_modulus))
        if spontaneous_curvature is not None and np.ndim(spontaneous_curvature) == 0:
            self.global_parameters.set("spontaneous_curvature", float(spontaneous_curvature))
        self.bodies = {}
        for b in bodies or []:
            self.bodies[b.index] = b
        self.energy_modules = list(energy_modules or [])
        self.constraint_modules = list(constraint_modules or [])
        self._boundary_mask = None
        self._tilts = (np.zeros_like(self._positions) if tilts is None
                       else np.array(tilts, dtype=np.float64, order="C", copy=True))
        self._tilts_version = 0
        self.tilt_fixed = (np.zeros(nv, dtype=bool) if tilt_fixed is None
                           else np.asarray(tilt_fixed, dtype=bool).copy())
        # two-leaflet tilt fields (geometry/mesh.py:427-470 of the reference: zeros until set)
        self._tilts_in = (np.zeros_like(self._positions) if tilts_in is None
                          else np.array(tilts_in, dtype=np.float64, order="C", copy=True))
        self._tilts_out = (np.zeros_like(self._positions) if tilts_out is None
                           else np.array(tilts_out, dtype=np.float64, order="C", copy=True))
        self.tilt_fixed_in = (np.zeros(nv, dtype=bool) if tilt_fixed_in is None
                              else np.asarray(tilt_fixed_in, dtype=bool).copy())
        self.tilt_fixed_out = (np.zeros(nv, dtype=bool) if tilt_fixed_out is None
                               else np.asarray(tilt_fixed_out, dtype=bool).copy())
        # rows tagged for tilt_disk_target_in/out (the reference reads vertex.options["tilt_disk_target_group_*"])
        self.disk_rows_in = None if disk_rows_in is None else np.asarray(disk_rows_in).copy()
        self.disk_rows_out = None if disk_rows_out is None else np.asarray(disk_rows_out).copy()
        self._version = 0
        self._facet_loops_version = 0
        self._vertex_ids_version = 0
        self._topology_version = 0
        assert nf == 0 or self._tri_rows.shape[1] == 3

    # -- the accessors the path reads ---------------------------------------
    def build_position_cache(self):
        return None

    def positions_view(self) -> np.ndarray:
        return self._positions

    def set_positions(self, positions) -> None:
        self._positions[...] = positions
        self.increment_version()

    def triangle_row_cache(self):
        return self._tri_rows, np.arange(self._tri_rows.shape[0])

    @property
    def vertex_index_to_row(self):
        if self._index_map is None:
            self._index_map = {int(v): i for i, v in enumerate(self.vertex_ids)}
        return self._index_map

    @property
    def fixed_mask(self) -> np.ndarray:
        return self._fixed

    @property
    def boundary_mask(self) -> np.ndarray:
        if self._boundary_mask is None:
            self._boundary_mask = boundary_mask_from_triangles(self._positions.shape[0], self._tri_rows)
        return self._boundary_mask

    @property
    def boundary_vertex_ids(self) -> set:
        return set(int(v) for v in self.vertex_ids[self.boundary_mask])

    def get_facet_parameter_array(self, param_name, default_val=None) -> np.ndarray:
        arr = self._facet_params.get(param_name)
        if arr is not None:
            return arr
        if default_val is None:
            default_val = self.global_parameters.get(param_name) or 0.0
        return np.full(self._tri_rows.shape[0], float(default_val))

    def get_vertex_parameter_array(self, param_name):
        return self._vertex_params.get(param_name)

    def increment_version(self):
        self._version += 1

    def replace_topology(self, positions, tri_rows, *, fixed=None, tilts=None) -> None:
        """Stand-in for the reference's refinement / equiangulation (runtime/refinement.py:287,
        runtime/equiangulation.py:81): new vertex and facet sets, version counters bumped the way
        ``Mesh.increment_topology_version`` does, so the device mirror re-tiles on its next use."""
        self._positions = np.array(positions, dtype=np.float64, order="C", copy=True)
        self._tri_rows = np.ascontiguousarray(tri_rows, dtype=np.int32)
        nv = self._positions.shape[0]
        self.vertex_ids = np.arange(nv, dtype=np.int64)
        self._index_map = None
        self._fixed = np.zeros(nv, dtype=bool) if fixed is None else np.asarray(fixed, dtype=bool).copy()
        self._facet_params = {}
        self._vertex_params = {}
        self._boundary_mask = None
        self._tilts = (np.zeros_like(self._positions) if tilts is None
                       else np.array(tilts, dtype=np.float64, order="C", copy=True))
        self.tilt_fixed = np.zeros(nv, dtype=bool)
        self._tilts_in = np.zeros_like(self._positions)
        self._tilts_out = np.zeros_like(self._positions)
        self.tilt_fixed_in = np.zeros(nv, dtype=bool)
        self.tilt_fixed_out = np.zeros(nv, dtype=bool)
        self._tilts_version += 1
        self._version += 1
        self._facet_loops_version += 1
        self._vertex_ids_version += 1
        self._topology_version += 1

    # vertex tilt field (geometry/mesh.py:391-430, :516-530 of the reference)
    def tilts_view(self) -> np.ndarray:
        return self._tilts

    def set_tilts_from_array(self, tilts) -> None:
        self._tilts[...] = tilts
        self._tilts_version += 1

    # leaflet tilt fields (geometry/mesh.py:427-470, :534-577 of the reference)
    def tilts_in_view(self) -> np.ndarray:
        return self._tilts_in

    def tilts_out_view(self) -> np.ndarray:
        return self._tilts_out

    def set_tilts_in_from_array(self, tilts) -> None:
        self._tilts_in[...] = tilts
        self._tilts_version += 1

    def set_tilts_out_from_array(self, tilts) -> None:
        self._tilts_out[...] = tilts
        self._tilts_version += 1

    def compute_total_surface_area(self) -> float:
        p, t = self._positions, self._tri_rows
        n = np.cross(p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]])
        return float(0.5 * np.linalg.norm(n, axis=1).sum())


# ---------------------------------------------------------------------------
def _boundary_mask_of(mesh, nv) -> np.ndarray:
    if hasattr(mesh, "boundary_mask"):
        return np.asarray(mesh.boundary_mask, dtype=bool)
    mask = np.zeros(nv, dtype=bool)
    idx = mesh.vertex_index_to_row
    for vid in mesh.boundary_vertex_ids or ():
        row = idx.get(vid)
        if row is not None:
            mask[row] = True
    return mask


def _body_facet_mask(mesh, nf):
    """Single-body facet mask (None = every facet).  -> (mask | None, body | None)."""
    bodies = getattr(mesh, "bodies", None) or {}
    if not bodies:
        return None, None
    if len(bodies) > 1:
        raise L.MembraneHipError(
            "the HIP path supports one body per mesh (SURVEY 8a row a7); "
            f"this mesh has {len(bodies)}")
    body = next(iter(bodies.values()))
    rows = getattr(body, "facet_rows", None)
    if rows is None and hasattr(body, "facet_indices"):
        f2r = mesh.facet_to_triangle_row
        rows = np.array([f2r[f] for f in body.facet_indices], dtype=np.int64)
    if rows is None or len(rows) == nf:
        return None, body
    mask = np.zeros(nf, dtype=np.uint8)
    mask[np.asarray(rows, dtype=np.int64)] = 1
    return mask, body


def _bending_defaults(global_params, model: str):
    kappa_default = float(global_params.get("bending_modulus", 0.0) or 0.0)
    if model == "helfrich":
        val = global_params.get("spontaneous_curvature")
        if val is None:
            val = global_params.get("intrinsic_curvature", 0.0)
        c0_default = float(val or 0.0)
    else:
        c0_default = 0.0
    return kappa_default, c0_default


def per_vertex_bending_params(mesh, global_params, model: str):
    """modules/energy/bending_params.py:41-115 restated for both mesh kinds."""
    nv = len(mesh.vertex_ids)
    kappa_default, c0_default = _bending_defaults(global_params, model)
    kappa = np.full(nv, kappa_default)
    c0 = np.full(nv, c0_default)
    if hasattr(mesh, "get_vertex_parameter_array"):
        k = mesh.get_vertex_parameter_array("bending_modulus")
        if k is not None:
            kappa = np.asarray(k, dtype=np.float64)
        if model == "helfrich":
            z = mesh.get_vertex_parameter_array("spontaneous_curvature")
            if z is not None:
                c0 = np.asarray(z, dtype=np.float64)
    elif hasattr(mesh, "vertices"):
        idx = mesh.vertex_index_to_row
        for vid, vertex in mesh.vertices.items():
            opts = getattr(vertex, "options", None) or {}
            if not opts:
                continue
            row = idx.get(int(vid))
            if row is None:
                continue
            if "bending_modulus" in opts:
                try:
                    kappa[row] = float(opts["bending_modulus"])
                except (TypeError, ValueError):
                    pass
            if model == "helfrich":
                for key in ("spontaneous_curvature", "intrinsic_curvature"):
                    if key in opts:
                        try:
                            c0[row] = float(opts[key])
                        except (TypeError, ValueError):
                            pass
                        break
    return kappa, c0


class HipMirror:
    """HBM mirror of one mesh, keyed on the reference's version counters."""

    def __init__(self, mesh, device: int = 0, tile_vertices: int = 0):
        self.mesh = mesh
        self.device = device
        self.tile_vertices = tile_vertices
        self.dm: DeviceMesh | None = None
        self._topo_key = None
        self._pos_version = None
        self._gamma_key = None
        self._bend_key = None
        self._leaflet_keys = {}
        self.body = None

    def _topology_key(self):
        m = self.mesh
        return (getattr(m, "_facet_loops_version", 0), getattr(m, "_vertex_ids_version", 0),
                getattr(m, "_topology_version", 0), len(m.vertex_ids))

    def sync(self, positions=None) -> DeviceMesh:
        """Bring the device copy up to date with the mesh (or with ``positions``)."""
        m = self.mesh
        m.build_position_cache()
        key = self._topology_key()
        if self.dm is None or key != self._topo_key:
            if self.dm is not None:
                self.dm.close()
            pos = np.ascontiguousarray(m.positions_view(), dtype=np.float64)
            tri, _ = m.triangle_row_cache()
            if tri is None:
                tri = np.zeros((0, 3), dtype=np.int32)
            nv, nf = pos.shape[0], tri.shape[0]
            body_mask, self.body = _body_facet_mask(m, nf)
            self.dm = DeviceMesh(pos, tri, fixed=np.asarray(m.fixed_mask, dtype=bool),
                                 boundary=_boundary_mask_of(m, nv), body_facets=body_mask,
                                 device=self.device, tile_vertices=self.tile_vertices)
            self._topo_key = key
            self._pos_version = getattr(m, "_version", None)
            self._gamma_key = None
            self._bend_key = None
            self._tilt_key = None
        if positions is not None:
            self.dm.set_positions(positions)
            self._pos_version = None  # foreign array: force a re-upload next time
        elif self._pos_version != getattr(m, "_version", None) or self._pos_version is None:
            self.dm.set_positions(m.positions_view())
            self._pos_version = getattr(m, "_version", None)
        return self.dm

    def mark_device_positions_current(self):
        """The device holds the positions that were just written back to the mesh."""
        self._pos_version = getattr(self.mesh, "_version", None)

    def upload_surface_tension(self):
        fp = getattr(self.mesh, "_facet_params", None)
        if isinstance(fp, dict) and fp.get("surface_tension") is None:
            # ArrayMesh without a per-facet array: one global value -- the key needs no nf-sized temporary
            # (this runs at the top of every minimize() call)
            val = float(self.mesh.global_parameters.get("surface_tension") or 0.0)
            key = (self._topo_key, "uniform", val)
            if key != self._gamma_key:
                self.dm.set_surface_tension(np.asarray(self.mesh.get_facet_parameter_array("surface_tension"),
                                                       dtype=np.float64))
                self._gamma_key = key
            return
        gamma = np.asarray(self.mesh.get_facet_parameter_array("surface_tension"), dtype=np.float64)
        key = (self._topo_key, gamma.tobytes() if gamma.size < 4096 else (float(gamma.sum()), float(gamma[0]), gamma.size))
        if key != self._gamma_key:
            self.dm.set_surface_tension(gamma)
            self._gamma_key = key

    def upload_tilts(self, global_params):
        """Mesh.tilts_view() + gp["tilt_rigidity"] (modules/energy/tilt.py:110, :122-127)."""
        k_t = float(global_params.get("tilt_rigidity", 0.0) or 0.0)
        tilts = np.ascontiguousarray(self.mesh.tilts_view(), dtype=np.float64)
        key = (self._topo_key, getattr(self.mesh, "_tilts_version", None), k_t,
               None if hasattr(self.mesh, "_tilts_version") else float(np.sum(tilts)))
        if key != getattr(self, "_tilt_key", None):
            self.dm.set_tilts(tilts, k_t)
            self._tilt_key = key

    def upload_leaflets(self, params: dict):
        """Mesh.tilts_in_view()/tilts_out_view(), the tilt_fixed_in/out flags and the per-leaflet module
        parameters (``params[leaflet]`` = kwargs of DeviceMesh.set_leaflet_tilts)."""
        for lf in ("in", "out"):
            view = getattr(self.mesh, f"tilts_{lf}_view", None)
            tilts = (np.zeros((len(self.mesh.vertex_ids), 3)) if view is None
                     else np.ascontiguousarray(view(), dtype=np.float64))
            mask = tilt_fixed_mask(self.mesh, f"tilt_fixed_{lf}")
            key = (self._topo_key, getattr(self.mesh, "_tilts_version", None), tuple(sorted(params[lf].items())),
                   mask.tobytes(), None if hasattr(self.mesh, "_tilts_version") else float(np.sum(tilts)))
            if key != self._leaflet_keys.get(lf):
                self.dm.set_leaflet_tilts(lf, tilts, tilt_fixed=mask if mask.any() else None, **params[lf])
                self._leaflet_keys[lf] = key

    def mark_device_leaflets_current(self):
        for lf, key in list(self._leaflet_keys.items()):
            if lf.startswith("bend_"):
                continue
            self._leaflet_keys[lf] = (key[0], getattr(self.mesh, "_tilts_version", None)) + key[2:4] + (None,)

    def upload_tilt_fixed(self):
        """vertex.tilt_fixed flags (runtime/minimizer_helpers.py:49-75)."""
        mask = tilt_fixed_mask(self.mesh)
        key = (self._topo_key, mask.tobytes())
        if key != getattr(self, "_tilt_fixed_key", None):
            self.dm.set_tilt_fixed(mask if mask.any() else None)
            self._tilt_fixed_key = key

    def mark_device_tilts_current(self):
        self._tilt_key = (self._topo_key, getattr(self.mesh, "_tilts_version", None),
                          self._tilt_key[2] if getattr(self, "_tilt_key", None) else 0.0, None)

    def upload_bending_params(self, global_params, model: str):
        get = getattr(self.mesh, "get_vertex_parameter_array", None)
        if get is not None and get("bending_modulus") is None and (model != "helfrich"
                                                                   or get("spontaneous_curvature") is None):
            key = (self._topo_key, model, "uniform") + _bending_defaults(global_params, model)
            if key != self._bend_key:
                self.dm.set_bending_params(*per_vertex_bending_params(self.mesh, global_params, model))
                self._bend_key = key
            return
        kappa, c0 = per_vertex_bending_params(self.mesh, global_params, model)
        key = (self._topo_key, model, float(kappa.sum()), float(c0.sum()), float(kappa[0]), float(c0[0]))
        if key != self._bend_key:
            self.dm.set_bending_params(kappa, c0)
            self._bend_key = key


def tilt_fixed_mask(mesh, attr: str = "tilt_fixed") -> np.ndarray:
    """(nv,) bool: ArrayMesh.tilt_fixed[_in|_out], or the reference Mesh's per-vertex attribute of that
    name (runtime/minimizer_helpers.py:49-75, minimizer.py:460-486)."""
    own = getattr(mesh, attr, None)
    if own is not None and not callable(own):
        return np.asarray(own, dtype=bool)
    verts = getattr(mesh, "vertices", None)
    ids = getattr(mesh, "vertex_ids", None)
    if verts is not None and ids is not None and isinstance(verts, dict):
        return np.array([bool(getattr(verts[int(v)], attr, False)) for v in ids], dtype=bool)
    return np.zeros(len(mesh.positions_view()), dtype=bool)


def mirror_for(mesh, device: int = 0, tile_vertices: int = 0) -> HipMirror:
    """Get (or create) the mirror stashed on the mesh, as the reference stashes caches on it."""
    mir = getattr(mesh, "_hip_mirror", None)
    if mir is None or mir.device != device:
        mir = HipMirror(mesh, device=device, tile_vertices=tile_vertices)
        mesh._hip_mirror = mir
    return mir
