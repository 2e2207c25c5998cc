"""CPU-only checks of the host layer: the C ABI exports what include/ declares,
the tiling pass keeps its invariants, the reference-shaped managers / steppers /
mesh accessors behave, and the product path refuses to run without the GPU
library instead of falling back."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from membrane_solver_amd import _lib

    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "membrane_hip.h")).read()
    declared = set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ms_params", "ms_stepper_params", "ms_step_result"}
    assert declared, "no prototypes found in the header"
    _lib.lib()  # preloads the HIP runtime, then the library
    cd = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(cd, n)]
    assert not missing, f"declared in membrane_hip.h but not exported: {missing}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().ms_version().startswith(b"membrane_hip")


def test_no_gpu_means_loud_failure_not_fallback():
    from membrane_solver_amd import _lib

    lib = _lib.lib()
    if lib.ms_device_count() > 0:
        pytest.skip("a GPU is present")
    from membrane_solver_amd.device import DeviceMesh

    g = load_golden("mesh_ico4.npz")
    with pytest.raises(_lib.MembraneHipError):
        DeviceMesh(g["positions"], g["tri"])
    # and nothing under the package imports the oracle
    pkg = os.path.join(ROOT, "membrane_solver_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "ms_oracle" not in txt, f


@pytest.mark.parametrize("tile", [64, 256, 512])
def test_tiling_invariants(tile):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen

    lib = L.lib()
    for P, T in (meshgen.icosphere(12), meshgen.disk_patch(9, jitter=0.2)[:2]):
        nv, nf = len(P), len(T)
        st = (ctypes.c_int64 * 8)()
        perm = np.empty(nv, np.int32)
        rc = lib.ms_plan_tiling(nv, nf, P.ctypes.data_as(L._D), T.ctypes.data_as(L._I32), tile, 1, st,
                                perm.ctypes.data_as(L._I32))
        assert rc == 0
        assert st[0] == (nv + tile - 1) // tile
        assert st[4] == 0 and st[5] == nf and st[6] == 3 * nf  # every facet owned once, every corner once
        assert st[1] >= nf and sorted(perm.tolist()) == list(range(nv))
        assert st[7] <= 160 * 1024
    # out-of-range indices are dropped like surface_energy.f90:57-59
    P, T = meshgen.icosphere(3)
    T = T.copy()
    T[5, 1] = len(P) + 7
    T[9, 0] = -1
    st = (ctypes.c_int64 * 8)()
    assert lib.ms_plan_tiling(len(P), len(T), P.ctypes.data_as(L._D), T.ctypes.data_as(L._I32), 64, 1, st, None) == 0
    assert st[4] == 2 and st[5] == len(T) - 2
    # bad arguments come back as error codes with a message, not exceptions
    assert lib.ms_plan_tiling(len(P), len(T), P.ctypes.data_as(L._D), T.ctypes.data_as(L._I32), 100, 1, st, None) == -1
    assert b"tile_vertices" in lib.ms_last_error(None)


def test_bisection_tiles_are_compact():
    """the patch order (recursive coordinate bisection down to single tiles) gives compact tiles:
    a facet is listed by 1.13-1.16 tiles on a refined icosphere (a perfect hexagonal patch of 256 lattice vertices:
    1.10; the runs of a 3-D Hilbert curve that round 1 used: 1.21), and no halo is longer than half a tile."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen

    lib = L.lib()
    P, T = meshgen.icosphere(48)  # 46 080 facets, 91 tiles of 256 vertices
    P = meshgen.smooth_displace(P, 0.05)
    nv, nf = len(P), len(T)
    st = (ctypes.c_int64 * 8)()
    perm = np.empty(nv, np.int32)
    assert lib.ms_plan_tiling(nv, nf, P.ctypes.data_as(L._D), T.ctypes.data_as(L._I32), 256, 1, st,
                              perm.ctypes.data_as(L._I32)) == 0
    assert sorted(perm.tolist()) == list(range(nv))
    assert st[5] == nf and st[6] == 3 * nf
    assert st[1] < 1.17 * nf, list(st)
    assert st[2] <= 128, list(st)
    # tiles of 200 rows (they run on the 256-thread instances): every vertex and every corner is still owned exactly once
    assert lib.ms_plan_tiling(nv, nf, P.ctypes.data_as(L._D), T.ctypes.data_as(L._I32), 200, 1, st,
                              perm.ctypes.data_as(L._I32)) == 0
    assert st[0] == (nv + 199) // 200 and sorted(perm.tolist()) == list(range(nv))
    assert st[5] == nf and st[6] == 3 * nf


def test_array_mesh_accessors_and_managers():
    from membrane_solver_amd.core.parameters import GlobalParameters, ParameterResolver
    from membrane_solver_amd.geometry.mesh import ArrayBody, ArrayMesh, per_vertex_bending_params
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent

    g = load_golden("mesh_disk5.npz")
    mesh = ArrayMesh(g["positions"], g["tri"], surface_tension=1.3,
                     global_parameters={"bending_modulus": 0.8, "spontaneous_curvature": 0.5},
                     bodies=[ArrayBody(0, None, 1.0)])
    assert mesh.positions_view().flags["C_CONTIGUOUS"] and mesh.triangle_row_cache()[0].dtype == np.int32
    assert np.array_equal(mesh.boundary_mask, g["is_boundary"])
    assert mesh.boundary_vertex_ids == set(np.flatnonzero(g["is_boundary"]).tolist())
    assert np.all(mesh.get_facet_parameter_array("surface_tension") == 1.3)
    v = mesh._version
    mesh.increment_version()
    assert mesh._version == v + 1
    k, c0 = per_vertex_bending_params(mesh, mesh.global_parameters, "helfrich")
    assert np.all(k == 0.8) and np.all(c0 == 0.5)
    _, c0w = per_vertex_bending_params(mesh, mesh.global_parameters, "willmore")
    assert np.all(c0w == 0.0)
    gp = GlobalParameters({"x": 3})
    assert gp.get("x") == 3 and gp.get("missing") is None and gp.volume_stiffness == 1000.0
    assert ParameterResolver(gp).get(ArrayBody(0, None, 1.0, {"volume_stiffness": 7.0}), "volume_stiffness") == 7.0
    em = EnergyModuleManager(["surface", "bending", "volume", "tilt"])
    assert em.get_module("tilt").USES_TILT is True
    for name in ("surface", "bending", "volume", "tilt"):
        assert hasattr(em.get_module(name), "compute_energy_and_gradient_array")
    with pytest.raises(KeyError):
        em.get_module("tilt_smoothness_in")
    with pytest.raises(ImportError):
        EnergyModuleManager(["not_a_module"])
    cm = ConstraintModuleManager(["volume"])
    assert hasattr(cm.get_constraint("volume"), "enforce_constraint")
    cg = ConjugateGradient()
    assert (cg.restart_interval, cg.max_iter, cg.beta, cg.c, cg.gamma, cg.alpha_max_factor) == (10, 10, 0.7, 1e-4, 1.5, 10.0)
    gd = GradientDescent()
    mesh.global_parameters.set("shape_line_search_max_iter", 4)
    assert gd._max_iter_for(mesh) == 4
    import inspect

    sig = inspect.signature(gd.step)
    assert list(sig.parameters)[:6] == ["mesh", "grad", "step_size", "energy_fn", "constraint_enforcer", "trial_energy_fn"]


def test_minimizer_rejects_out_of_scope_modules():
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    g = load_golden("mesh_ico4.npz")
    mesh = ArrayMesh(g["positions"], g["tri"])

    class FakeEM:
        def get_module(self, name):
            class M:
                @staticmethod
                def compute_energy_and_gradient_array(*a, **k):
                    return 0.0
            return M

    with pytest.raises(L.MembraneHipError):
        Minimizer(mesh, mesh.global_parameters, GradientDescent(), FakeEM(), ConstraintModuleManager([]),
                  energy_modules=["line_tension"], constraint_modules=[])
    mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(["surface"]),
                   ConstraintModuleManager([]), energy_modules=["surface"], constraint_modules=[], quiet=True)
    assert mz.step_size == 1e-3 and mz.tol == 1e-6 and mz.stepper.__class__.__name__ == "GradientDescent"


def test_meshgen_counts():
    from membrane_solver_amd import meshgen

    for f in (1, 3, 10):
        P, T = meshgen.icosphere(f)
        assert P.shape == (10 * f * f + 2, 3) and T.shape == (20 * f * f, 3)
        assert np.allclose(np.linalg.norm(P, axis=1), 1.0)
        assert not meshgen.boundary_mask_from_triangles(len(P), T).any()


def test_bench_launcher_command_and_schema_helpers(monkeypatch):
    """bench.py --gpus N started as a plain process spawns its ranks through torch.distributed.run (fresh children,
    127.0.0.1 rendezvous); the roofline block reports the instantiation with the largest time share."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    cmd = bench.launcher_command(["--gpus", "8", "--steps", "5", "--warmup", "2"], 8, 29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[cmd.index("--master-port") + 2] == os.path.join(ROOT, "bench.py")
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    # main() must take the spawn path BEFORE importing anything that touches the GPU
    called = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "spawn_ranks", lambda args, argv: called.setdefault("argv", list(argv)) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.main() == 0 and called["argv"] == ["--gpus", "2", "--steps", "3"]
    assert bench.weak_frequency(1) == 320 and bench.weak_frequency(4) == 640 and bench.weak_frequency(8) == 905

    nv, nf = 1024002, 2048000
    ab = bench.algorithmic_bytes(nv, nf)
    assert ab["gradient_lean"] == 12 * nf + (65 + 24 + 48) * nv          # tri rows; x, fK, fA, flags; pg; g, d
    assert ab["gradient"] == ab["gradient_lean"] + 24 * nv               # + previous direction rows
    assert ab["energy_pair"] == ab["energy_triple"] == ab["energy_trial_factors"]  # only the last trial writes outputs
    full = bench.algorithmic_bytes(nv, nf, lean_pairs=False)             # MS_PAIR_LEAN=0 / sharded driver
    assert full["energy_pair"] == ab["energy_trial_factors"] + 64 * nv   # inputs once, outputs twice
    assert bench.algorithmic_bytes(nv, nf, uniform=False)["energy_only"] == ab["energy_only"] + 8 * nf + 16 * nv
    prof = {"energy": (0.36, 10), "gradient": (0.0, 0), "gradient_lean": (0.50, 10), "reduce": (0.06, 10),
            "energy_pair": (0.30, 5), "energy_triple": (0.0, 0), "direction": (0.0, 0)}
    k = bench.kernel_table(prof, ab, trial_passes=20, level=2, nv=nv, deterministic=False)
    r = bench.roofline_block(k, 10)
    assert r["kernel"] == bench.KERNEL_NAMES["gradient_lean"]            # 0.50 ms is the largest share
    assert abs(r["achieved"] - ab["gradient_lean"] / 50e-6 / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert {"energy_family_time_weighted", "gradient_family_time_weighted", "traffic", "traffic_source",
            "share_of_kernel_time"} <= set(r)
    rt = bench.rates(20, 10, 21, 0, 2, 0.002)
    assert rt["accepted_steps_per_s"] == 5000.0 and rt["evaluations_per_s"] == (21 + 10) / 0.002


def test_bench_leg_budget_and_shared_mesh(tmp_path, monkeypatch):
    """bench.py --gpus N: an extra leg starts only while its estimated cost fits MS_BENCH_BUDGET_S, and rank 0's mesh
    reaches the other ranks through files instead of every rank running the generator."""
    import threading

    import bench

    clock = {"t": 0.0}
    b = bench.LegBudget(budget_s=100.0, now=lambda: clock["t"], t0=0.0, first_guess_s_per_mfacet=5.0)
    assert b.estimate(2_000_000) == pytest.approx(15.0)        # 1.5 x 5 s per million facets
    assert b.allows(2_000_000)
    clock["t"] = 20.0
    b.observe("headline", 2_048_000, 20.0)                     # measured: ~9.8 s per million facets
    assert b.estimate(2_048_000) == pytest.approx(30.0)
    assert b.allows(2_048_000)                                 # 20 + 30 <= 100
    assert not b.allows(16_380_500)                            # 20 + 240 > 100
    note = b.skip_note("strong_16M_facets", 16_380_500)
    assert "skipped" in note["note"] and "MS_BENCH_BUDGET_S=100" in note["note"]
    clock["t"] = 75.0
    assert not b.allows(2_048_000)                             # 75 + 30 > 100
    b.observe("fast leg", 2_048_000, 2.0)                      # the estimate never drops below the slowest leg seen
    assert b.estimate(2_048_000) == pytest.approx(30.0)
    assert [e["leg"] for e in b.log] == ["headline", "strong_16M_facets", "fast leg"] and b.log[1]["skipped"]
    monkeypatch.setenv("MS_BENCH_BUDGET_S", "42")
    assert bench.LegBudget().budget_s == 42.0

    # two "ranks" (threads, each with the cache a process of its own would have) share one mesh: only rank 0 runs
    # the generator
    from membrane_solver_amd import meshgen

    calls = []
    real = meshgen.icosphere
    monkeypatch.setattr(meshgen, "icosphere", lambda f: (calls.append(f), real(f))[1])
    bar = threading.Barrier(2)
    out, caches = {}, {0: {}, 1: {}}

    def rank_fn(r):
        out[r] = bench.shared_bench_mesh(3, r, 2, bar.wait, tag="t", shm_dir=str(tmp_path), cache=caches[r])

    t1 = threading.Thread(target=rank_fn, args=(1,))
    t1.start()
    rank_fn(0)
    t1.join()
    assert calls == [3]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][1].shape == (180, 3) and 3 in caches[1]
    assert not list(tmp_path.iterdir())                        # rank 0 removed the files


def test_bench_headline_selection_between_the_two_sharded_drivers():
    """bench.py --gpus N: the line's value is the peer-exchange leg's only when that leg is the same measurement and
    faster; a skipped / failed leg, another step count, the weak-scaling run or MS_BENCH_HEADLINE=rccl keep the RCCL
    all-gather driver's."""
    import bench

    leg = {"driver": "library", "steps": 200, "warmup": 30, "value": 13000.0}
    assert bench.headline_from_peer_leg(leg, 200, 30, 12400.0)
    assert not bench.headline_from_peer_leg(leg, 200, 30, 13500.0)
    assert not bench.headline_from_peer_leg(leg, 20, 5, 12400.0)
    assert not bench.headline_from_peer_leg(dict(leg, driver="python"), 200, 30, 12400.0)
    assert not bench.headline_from_peer_leg(leg, 200, 30, 12400.0, weak=True)
    assert not bench.headline_from_peer_leg(leg, 200, 30, 12400.0, pin="rccl")
    assert not bench.headline_from_peer_leg(None, 200, 30, 12400.0)
    assert not bench.headline_from_peer_leg({"note": "peer-to-peer exchange could not be set up on every rank"}, 200, 30, 1.0)
