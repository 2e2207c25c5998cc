"""CPU check (runs only where the reference checkout exists, i.e. in the build container):
``HipMirror`` / the plugins read the reference's OWN ``Mesh`` object through exactly the
accessors listed in INTEGRATION.md.  The device is replaced by a recorder, so this tests
the adapter, not the kernels (those are covered by the -m gpu suite)."""

import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "geometry")),
                                reason="reference checkout not present (GPU box)")


class RecorderDevice:
    def __init__(self, positions, tri_rows, *, fixed=None, boundary=None, body_facets=None, **kw):
        self.args = dict(positions=np.array(positions), tri=np.array(tri_rows), fixed=np.array(fixed),
                         boundary=np.array(boundary), body_facets=None if body_facets is None else np.array(body_facets))
        self.calls = []

    def set_positions(self, p):
        self.calls.append(("set_positions", np.array(p)))

    def set_surface_tension(self, g):
        self.calls.append(("gamma", np.array(g)))

    def set_bending_params(self, k, c):
        self.calls.append(("bend", np.array(k), np.array(c)))

    def close(self):
        pass


def test_mirror_reads_reference_mesh(monkeypatch):
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from geometry.geom_io import load_data, parse_geometry
    finally:
        pass
    from membrane_solver_amd.geometry import mesh as amesh

    monkeypatch.setattr(amesh, "DeviceMesh", RecorderDevice)
    m = parse_geometry(load_data(os.path.join(REF, "meshes", "cube.json")))
    # mark one vertex fixed and give one facet / vertex its own parameters
    vid0 = int(m.vertex_ids[0])
    m.vertices[vid0].fixed = True
    m._fixed_flags_version = getattr(m, "_fixed_flags_version", 0) + 1
    fid = next(iter(m.facets))
    m.facets[fid].options["surface_tension"] = 3.5
    m.vertices[int(m.vertex_ids[3])].options["bending_modulus"] = 0.25
    m.global_parameters.set("bending_modulus", 1.5)
    mir = amesh.HipMirror(m)
    dev = mir.sync()
    tri, facets = m.triangle_row_cache()
    assert np.array_equal(dev.args["positions"], m.positions_view())
    assert np.array_equal(dev.args["tri"], tri) and dev.args["tri"].shape == (24, 3)
    assert dev.args["fixed"].sum() == 1 and dev.args["fixed"][0]
    assert not dev.args["boundary"].any()  # closed cube
    assert dev.args["body_facets"] is None and mir.body is m.bodies[next(iter(m.bodies))]
    mir.upload_surface_tension()
    gamma = dev.calls[-1][1]
    row = m.facet_to_triangle_row[fid]
    assert gamma[row] == 3.5 and np.sum(gamma != 3.5) == 23
    mir.upload_bending_params(m.global_parameters, "helfrich")
    kappa = dev.calls[-1][1]
    assert kappa[3] == 0.25 and np.sum(kappa == 1.5) == len(kappa) - 1
    # positions are re-uploaded only when the mesh version moves
    n = len(dev.calls)
    mir.sync()
    assert len(dev.calls) == n
    m.vertices[vid0].position[:] += 0.1
    m.increment_version()
    mir.sync()
    assert dev.calls[-1][0] == "set_positions" and np.array_equal(dev.calls[-1][1], m.positions_view())
    # the reference's plugin manager and ours expose the same module surface
    from runtime.energy_manager import EnergyModuleManager as RefEMM

    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager

    for name in ("surface", "bending", "volume", "tilt"):
        ref_mod, our_mod = RefEMM([name]).get_module(name), EnergyModuleManager([name]).get_module(name)
        assert hasattr(ref_mod, "compute_energy_and_gradient_array") and hasattr(our_mod, "compute_energy_and_gradient_array")
        import inspect

        ref_params = list(inspect.signature(ref_mod.compute_energy_and_gradient_array).parameters)
        our_params = list(inspect.signature(our_mod.compute_energy_and_gradient_array).parameters)
        assert ref_params[:3] == our_params[:3] == ["mesh", "global_params", "param_resolver"]
        for kw in ("positions", "index_map", "grad_arr"):
            assert kw in ref_params and kw in our_params
