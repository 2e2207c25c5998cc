"""Volume constraint plugin on the HIP path (modules/constraints/volume.py).

``constraint_gradients_array`` (:43-66) returns the dense KKT row dV/dx;
``enforce_constraint`` (:69-149) is the geometric projection loop
x -= (dV / (|dV/dx|^2 + 1e-12)) dV/dx on movable rows, run on the device.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for


def _target(mesh):
    bodies = getattr(mesh, "bodies", None) or {}
    if not bodies:
        return None
    body = next(iter(bodies.values()))
    t = body.target_volume
    if t is None:
        t = (body.options or {}).get("target_volume")
    return None if t is None else float(t)


def constraint_gradients_array(mesh, global_params, *, positions: np.ndarray, index_map) -> list | None:
    _ = index_map
    if global_params.get("volume_constraint_mode", "lagrange") != "lagrange":
        return None
    if _target(mesh) is None:
        return None
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    dm.set_params(modules=L.MS_CON_VOLUME)
    dm.energy_and_gradient(want_grad=False)
    return [dm.get_vertex_buffer(L.MS_BUF_GC)]


def enforce_constraint(mesh, tol: float = 1e-12, max_iter: int = 3, global_params: Any | None = None,
                       force_projection: bool = False, **kwargs) -> None:
    mode = global_params.get("volume_constraint_mode", "lagrange") if global_params is not None else "projection"
    if not (force_projection or mode in {"lagrange", "projection"}):
        return
    if kwargs.get("context", "minimize") in {"finalize", "mesh_operation"}:
        max_iter = max(int(max_iter), 12)
    target = _target(mesh)
    if target is None:
        return
    mir = mirror_for(mesh)
    dm = mir.sync()
    iters, _v = dm.project_volume(target, tol=tol, max_iter=max_iter)
    if iters > 0:
        mesh.positions_view()[...] = dm.get_positions()
        if hasattr(mesh, "vertices"):  # reference Mesh: write the objects back
            pos = mesh.positions_view()
            for row, vid in enumerate(mesh.vertex_ids):
                mesh.vertices[int(vid)].position[:] = pos[row]
        mesh.increment_version()
        mir.mark_device_positions_current()


__all__ = ["enforce_constraint", "constraint_gradients_array"]
