#!/usr/bin/env python3
"""Headline benchmark: minimizer steps/sec (energy + gradient + CG) on the
2 048 000-facet icosphere (BASELINE.json configs[2]; configs[3] when --gpus > 1).

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one iteration of Minimizer.minimize (runtime/minimizer.py:1230-1515
of the reference): energy+gradient of all modules, fixed-row zeroing, per-row
Polak-Ribiere CG direction, Armijo backtracking line search (>= 1 trial energy
evaluation) and the position commit.  All state is resident in HBM before the
timed region starts.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
METRIC = "minimizer steps/sec (energy+grad+CG) on 2M-facet icosphere"

# profile kind (ms_profile_read) -> the kernel instantiation as rocprofv3 names it (default LDS-atomic mode; the
# deterministic mode's fifth template argument is `false`)
KERNEL_NAMES = {
    "energy": "ms::k_energy<true, false, 256, 0, true, 0>",
    "energy_pair": "ms::k_energy<true, false, 256, 0, true, 2>",
    "energy_triple": "ms::k_energy<true, false, 256, 0, true, 3>",
    "gradient": "ms::k_gradient<1, false, 256, 0, true, false>",
    "gradient_lean": "ms::k_gradient<1, false, 256, 0, true, true>",
    "energy_multi": "ms::k_energy<true, false, 256, 0, true, 8>",
}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--freq", type=int, default=320, help="icosphere frequency (320 -> 2 048 000 facets)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: 2 048 000 facets PER GPU (frequency 320*sqrt(N)) instead of one 2M-facet "
                         "mesh sharded over the N GPUs")
    ap.add_argument("--tile", type=int, default=0, help="owned vertices per tile (0 = library default)")
    ap.add_argument("--step-size", type=float, default=1e-6)
    ap.add_argument("--volume", action="store_true", help="add the volume Lagrange constraint row")
    ap.add_argument("--cpu-steps", type=int, default=-1,
                    help="CPU oracle steps per cpu_baseline leg (0 = skip; default 6 at N=1, 2 at N>1)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-large", action="store_true",
                    help="skip the 16-million-facet legs (f = 905: the size at which strong scaling over 8 GPUs means "
                         "something, and the weak-scaling size of 8 GPUs)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary sections (reuse level 0, deterministic mode): for kernel traces")
    ap.add_argument("--reuse-level", type=int, default=2, choices=(0, 1, 2),
                    help="ms_stepper_params.reuse_energy0: 0 repeats every pass the reference repeats, "
                         "2 (library default) never repeats a pass whose result is already on the device")
    ap.add_argument("--deterministic", action="store_true",
                    help="fixed-order per-vertex sums (ms_set_deterministic): bitwise reproducible run to "
                         "run, slower than the default LDS-atomic accumulation")
    return ap


def weak_frequency(gpus: int, base: int = 320) -> int:
    """Icosphere frequency with ~2 048 000 facets per GPU (20 f^2 facets in all)."""
    return int(round(base * gpus ** 0.5))


def algorithmic_bytes(nv, nf, *, uniform=True, volume=False, lean_pairs=True):
    """Compulsory HBM bytes per launch (DESIGN.md 'Kernels'); each array touched once.  `uniform`: every facet has
    the same surface tension and every vertex the same kappa / c0 -- then those arrays are kernel constants and are
    neither read nor counted (ms_set_surface_tension / ms_set_bending_params decide; the bench's case)."""
    fac = 12 * nf + (0 if uniform else 8 * nf)      # packed triangle rows (+ gamma per facet)
    vp = 1 + (0 if uniform else 16)                 # vertex flags (+ kappa, c0)
    return {
        "energy_only": fac + (24 + vp) * nv,                          # pass at x, no factor write
        "energy_factors": fac + (24 + vp) * nv + 40 * nv,             # pass at x writing fK, fA
        "energy_trial": fac + (48 + vp) * nv + 24 * nv,               # x, d in; xt out
        "energy_trial_factors": fac + (48 + vp) * nv + 24 * nv + 40 * nv,
        # pair / triple launch: the inputs are compulsory ONCE (the other evaluations' reads are meant to hit L2),
        # the outputs once per evaluation that writes any -- in ms_step only the LAST trial of the launch does (the
        # early ones, expected to fail, are evaluated for their energies: `lean_pairs`; MS_PAIR_LEAN=0 and the sharded
        # driver write every trial's outputs)
        "energy_pair": fac + (48 + vp) * nv + (1 if lean_pairs else 2) * (24 + 40) * nv,
        "energy_triple": fac + (48 + vp) * nv + (1 if lean_pairs else 3) * (24 + 40) * nv,
        "energy_multi": fac + (48 + vp) * nv + (24 + 40) * nv,  # four to eight trials, only the last one writes outputs
        # gradient with the fused direction pass: x 24 + fK,fA 40 + flags 1 in; g 24 and d 24 out; CG history in:
        # the previous gradient (24) and, unless it is minus that (after an implicit steepest-descent step: the lean
        # instance), the previous direction (24).  With a constraint row the direction is not fused: g and gC out.
        "gradient_lean": fac + (24 + 40 + 1) * nv + 24 * nv + 48 * nv,
        "gradient": fac + (24 + 40 + 1) * nv + (48 * nv if volume else 48 * nv + 48 * nv),
    }


def _template_args(name):
    i, j = name.find("<"), name.rfind(">")
    return [a.strip() for a in name[i + 1:j].split(",")] if 0 <= i < j else []


def pmc_traffic(kernel_name):
    """HBM bytes per launch of one kernel instantiation from the newest committed rocprofv3 PMC summary
    (profiles/<tag>_pmc_summary.csv: separate FETCH_SIZE / WRITE_SIZE passes of this same bench command).
    NOT measured in this run -- the source file is named next to the figure.  gfx950 correction: FETCH_SIZE counts
    1/2 of the fetched bytes (MI355X_MICROARCH.md, calibrated on the direction pass's known byte count)."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    want = _template_args(kernel_name)
    base = kernel_name.split("<")[0]
    for path in reversed(files):
        acc = {"FETCH_SIZE": [0.0, 0.0], "WRITE_SIZE": [0.0, 0.0]}
        with open(path) as f:
            for row in csv.reader(line for line in f if not line.startswith("#")):
                if len(row) < 4 or row[0] not in acc or (base + "<") not in row[1]:
                    continue
                if _template_args(row[1][row[1].index(base):]) != want:
                    continue
                n = float(row[2])
                acc[row[0]][0] += n * float(row[3])
                acc[row[0]][1] += n
        if acc["FETCH_SIZE"][1] and acc["WRITE_SIZE"][1]:
            fetch = acc["FETCH_SIZE"][0] / acc["FETCH_SIZE"][1]
            write = acc["WRITE_SIZE"][0] / acc["WRITE_SIZE"][1]
            return (2.0 * fetch + write) * 1024.0, os.path.basename(path)
    return None, None


def kernel_table(prof, ab, *, trial_passes, level, nv, deterministic):
    """Per-instantiation averages, time shares, algorithmic bytes and GB/s from ms_profile_read."""
    kernels = {}
    for kind, (ms, n) in prof.items():
        if n:
            kernels[kind] = {"avg_us": 1e3 * ms / n, "launches": n, "share_of_profiled_ms": ms}
    tot = sum(v["share_of_profiled_ms"] for v in kernels.values()) or 1.0
    for v in kernels.values():
        v["share_of_profiled_ms"] = v["share_of_profiled_ms"] / tot
    n_e = prof.get("energy", (0.0, 0))[1]
    n_p = prof.get("energy_pair", (0.0, 0))[1]
    n_t = prof.get("energy_triple", (0.0, 0))[1]
    n_g = prof.get("gradient", (0.0, 0))[1] + prof.get("gradient_lean", (0.0, 0))[1]
    for kind, key, n_ev in (("energy_pair", "energy_pair", 2), ("energy_triple", "energy_triple", 3),
                            ("energy_multi", "energy_multi", None)):
        if kind in kernels:
            kernels[kind]["algorithmic_bytes"] = ab[key]
            if n_ev is not None:
                kernels[kind]["evaluations_per_launch"] = n_ev
    if n_e:
        # single launches are a mix of passes at x (with or without the factor write) and trial passes: weight by
        # what was actually launched
        n_trial = min(n_e, max(0, trial_passes - 2 * n_p - 3 * n_t))
        n_plain = n_e - n_trial
        n_fact = min(n_plain, n_g)      # factor-writing passes at x
        n_e0 = n_plain - n_fact         # energy0 re-evaluations (level 0)
        trial_key = "energy_trial_factors" if level >= 2 else "energy_trial"
        kernels["energy"]["algorithmic_bytes"] = (n_fact * ab["energy_factors"] + n_e0 * ab["energy_only"]
                                                  + n_trial * ab[trial_key]) / n_e
    for kind in ("gradient", "gradient_lean"):
        if kind in kernels:
            kernels[kind]["algorithmic_bytes"] = ab[kind]
    for kind, v in kernels.items():
        if "algorithmic_bytes" in v:
            v["GBps"] = v["algorithmic_bytes"] / (v["avg_us"] * 1e-6) / 1e9
            v["frac_of_hbm_peak"] = v["GBps"] / HBM_PEAK_GBS
            name = KERNEL_NAMES[kind]
            if deterministic:
                a = _template_args(name)
                a[4] = "false"
                name = name.split("<")[0] + "<" + ", ".join(a) + ">"
            v["kernel"] = name
    return kernels


def roofline_block(kernels, n_prof):
    """`roofline` of the contract: the kernel INSTANTIATION with the largest share of the profiled kernel time --
    one row of the rocprofv3 summary under profiles/ -- plus the figures the verdicts asked to see next to it."""
    cands = [k for k in KERNEL_NAMES if k in kernels and "GBps" in kernels[k]]
    if not cands:
        return None
    dom = max(cands, key=lambda k: kernels[k]["share_of_profiled_ms"])
    d = kernels[dom]
    traffic, src = pmc_traffic(d["kernel"])
    out = {"bound": "hbm", "kernel": d["kernel"], "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": d["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
           "traffic_source": (f"profiles/{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command, "
                              "committed; NOT measured in this run") if src else None,
           "avg_launch_us": d["avg_us"], "algorithmic_bytes_per_launch": d["algorithmic_bytes"],
           "share_of_kernel_time": d["share_of_profiled_ms"],
           "selection": "the instantiation with the largest share of the profiled kernel time",
           "measured": f"HIP events around every launch over {n_prof} steps after the timed region"}
    if traffic is not None and traffic < d["algorithmic_bytes"]:
        out["traffic_note"] = ("traffic below the algorithmic bytes: the algorithmic figure counts the connectivity as "
                               "SURVEY 8(d) does, 12 B per facet (three int32 vertex ids); the kernels read it as 4-byte "
                               "tile-local facet records (1.13 instances per facet) plus the halo id lists")
    if out["avg_launch_us"] > 0:
        out["traffic_GBps"] = (traffic / (d["avg_us"] * 1e-6) / 1e9) if traffic is not None else None
    fam = [k for k in ("energy", "energy_pair", "energy_triple", "energy_multi") if k in kernels and "GBps" in kernels[k]]
    if fam:
        b = sum(kernels[k]["algorithmic_bytes"] * kernels[k]["launches"] for k in fam)
        t = sum(kernels[k]["avg_us"] * kernels[k]["launches"] for k in fam)
        ev = sum(kernels[k].get("evaluations_per_launch", 1) * kernels[k]["launches"] for k in fam)
        out["energy_family_time_weighted"] = {
            "GBps": b / (t * 1e-6) / 1e9, "frac": b / (t * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "share_of_kernel_time": sum(kernels[k]["share_of_profiled_ms"] for k in fam),
            "us_per_evaluation": t / max(ev, 1),
            "note": ("a pair / triple launch evaluates two / three trial energies on ONE set of compulsory bytes (inputs "
                     "once, only the last trial writes outputs), so bytes per time understates it: compare "
                     "us_per_evaluation with the single launch's avg_us")}
    gfam = [k for k in ("gradient", "gradient_lean") if k in kernels and "GBps" in kernels[k]]
    if gfam:
        b = sum(kernels[k]["algorithmic_bytes"] * kernels[k]["launches"] for k in gfam)
        t = sum(kernels[k]["avg_us"] * kernels[k]["launches"] for k in gfam)
        out["gradient_family_time_weighted"] = {"GBps": b / (t * 1e-6) / 1e9, "frac": b / (t * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                "share_of_kernel_time": sum(kernels[k]["share_of_profiled_ms"] for k in gfam)}
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(x_now, T, mods, cons, bodies, gp, step_size, n_steps):
    """The oracle port timed on the host cores: one core (the checker build, strict order of operations) and all
    cores (the same C source built with OpenMP: facet loops split over the cores, vertex sums by atomics)."""
    from oracle import minimizer_port as mp
    from oracle import ms_oracle as orc

    nf = len(T)
    legs = {}
    n_cores = os.cpu_count() or 1
    try:
        n_cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    host_cores = n_cores
    n_cores = min(n_cores, 16)  # a one-GPU box's CPU share; more threads only fight over the atomics
    for name, omp in (("one_core", False), ("all_cores", True)):
        orc.use_openmp(omp)
        os.environ["OMP_NUM_THREADS"] = str(n_cores if omp else 1)
        try:
            p = mp.Problem(positions=x_now, tri=T, energy_modules=mods, constraint_modules=cons,
                           target_volume=bodies[0].target_volume if bodies else None, gp=dict(gp))
            if omp:
                mp.energy_and_gradient(p, p.positions)  # thread pool start-up outside the timing
            t0 = time.perf_counter()
            cres = mp.minimize(p, mp.ConjugateGradient(), n_steps, step_size=step_size)
            cdt = time.perf_counter() - t0
            legs[name] = {"value": n_steps / cdt, "unit": "steps/s", "cores": n_cores if omp else 1, "seconds": cdt,
                          "line_search_trials": int(sum(t["trials"] for t in cres["trace"]))}
        finally:
            orc.use_openmp(False)
    one = legs["one_core"]
    return {"value": one["value"], "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_steps} minimizer steps of the C/NumPy oracle port (oracle/minimizer_port.py over "
                      f"oracle/ms_oracle.c) on the same {nf}-facet mesh, starting from the GPU run's state and step "
                      f"size; {one['line_search_trials']} line-search trials",
            "seconds": one["seconds"], "cpu_model": cpu_model(), "host_cores": host_cores,
            "all_cores": {**legs["all_cores"],
                          "note": "the same C source built with -fopenmp (libms_oracle_omp.so): facet loops over all "
                                  "host cores, vertex scatter-adds by `omp atomic`; the NumPy control flow and the "
                                  "per-vertex passes stay serial.  Timed only, never a checker."}}


# ---------------------------------------------------------------------------------------------------------------
def launcher_command(argv, gpus, port, python=None):
    """The command `bench.py --gpus N` spawns when it was started as a plain process (one rank per GPU, RCCL)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """Started without a rendezvous (no WORLD_SIZE) but with --gpus N > 1: start the N ranks as fresh child processes
    -- this process has made no GPU call and makes none -- relay rank 0's JSON line, fail if any rank failed."""
    cmd = launcher_command(argv, args.gpus, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"[bench] {args.gpus}-rank run failed (exit code {proc.returncode}, JSON line {'found' if line else 'missing'})",
              file=sys.stderr)
        return proc.returncode or 1
    print(line)
    return 0


_MESH_CACHE: dict = {}
T_PROCESS_START = time.perf_counter()


def bench_mesh(freq, cache=None):
    """The displaced icosphere of frequency `freq`; kept for the life of the process (the 16 M-facet mesh serves up to
    three legs of one invocation)."""
    from membrane_solver_amd import meshgen

    cache = _MESH_CACHE if cache is None else cache
    if freq not in cache:
        P, T = meshgen.icosphere(freq)
        cache[freq] = (meshgen.smooth_displace(P, 0.05), T)
    return cache[freq]


def shared_bench_mesh(freq, rank, world, barrier, tag=None, shm_dir="/dev/shm", cache=None):
    """bench_mesh() for a group of ranks on ONE node: rank 0 builds the mesh and hands it to the others through files in
    /dev/shm (np.save / np.load), instead of `world` processes running the same 30-second NumPy generator side by side
    on the node's cores.  `barrier` is a callable every rank calls (dist.barrier).  Falls back to building locally when
    the files cannot be written or read."""
    cache = _MESH_CACHE if cache is None else cache  # (per process; the tests' in-process ranks pass their own)
    if freq in cache or world <= 1:
        return bench_mesh(freq, cache)
    tag = tag or os.environ.get("MASTER_PORT", "0")
    base = os.path.join(shm_dir, f"ms_bench_mesh_{tag}_f{freq}")
    ok_path = base + ".ok"
    if rank == 0:
        try:
            P, T = bench_mesh(freq, cache)
            np.save(base + "_P.npy", P)
            np.save(base + "_T.npy", T)
            with open(ok_path, "w") as fh:
                fh.write("1")
        except OSError as exc:
            print(f"[bench] mesh f={freq} not shared through {shm_dir}: {exc!r}", file=sys.stderr)
    barrier()
    if rank != 0:
        try:
            if not os.path.exists(ok_path):
                raise OSError("rank 0 wrote no mesh files")
            cache[freq] = (np.load(base + "_P.npy"), np.load(base + "_T.npy"))
        except (OSError, ValueError) as exc:
            print(f"[bench] rank {rank}: mesh f={freq} built locally ({exc!r})", file=sys.stderr)
            bench_mesh(freq, cache)
    barrier()
    if rank == 0:
        for suffix in ("_P.npy", "_T.npy", ".ok"):
            try:
                os.unlink(base + suffix)
            except OSError:
                pass
    return cache[freq]


class LegBudget:
    """Wall-clock budget of one bench.py invocation (MS_BENCH_BUDGET_S, default 300 s, counted from process start): the
    contract line must appear within the driver's patience, so an extra leg only starts when its ESTIMATED cost still
    fits.  The estimate of a leg is the measured cost per facet of the legs that have run (set-up and timed run
    together) times the leg's facet count, 1.5 x padded; before any leg has run, `first_guess_s_per_mfacet`."""

    def __init__(self, budget_s=None, now=time.perf_counter, t0=None, first_guess_s_per_mfacet=6.0):
        env = os.environ.get("MS_BENCH_BUDGET_S")
        self.budget_s = float(budget_s if budget_s is not None else (env if env else 300.0))
        self.now = now
        self.t0 = T_PROCESS_START if t0 is None else t0
        self.rate = float(first_guess_s_per_mfacet)  # seconds per million facets
        self.n_obs = 0
        self.log = []

    def elapsed(self):
        return self.now() - self.t0

    def estimate(self, nf):
        return 1.5 * self.rate * (nf / 1e6)

    def allows(self, nf):
        return self.elapsed() + self.estimate(nf) <= self.budget_s

    def observe(self, name, nf, seconds):
        r = seconds / max(nf / 1e6, 1e-9)
        self.rate = r if self.n_obs == 0 else max(self.rate, r)
        self.n_obs += 1
        self.log.append({"leg": name, "facets": int(nf), "wall_s": round(float(seconds), 2)})

    def skip_note(self, name, nf):
        note = (f"skipped: {self.elapsed():.0f} s elapsed + an estimated {self.estimate(nf):.0f} s would pass "
                f"MS_BENCH_BUDGET_S={self.budget_s:.0f}")
        self.log.append({"leg": name, "facets": int(nf), "skipped": True})
        return {"note": note}



GP = {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
      "spontaneous_curvature": 0.0, "bending_gradient_mode": "analytic",
      "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False,
      "mesh_quality_auto_repair_enabled": False}


def workload_text(freq, nv, nf, level, deterministic, volume=False):
    return (f"class-I icosphere f={freq} (nv={nv}, nf={nf}), surface + Helfrich bending (analytic cotan gradient), "
            f"CG stepper, Armijo line search, evaluation reuse level {level} (0 = every pass the reference re-runs, "
            "2 = passes already on the device are not repeated; same trajectories), per-vertex sums "
            + ("in fixed order (bitwise reproducible)" if deterministic
               else "by LDS atomics (default; --deterministic for fixed-order sums)")
            + (", volume Lagrange row" if volume else ""))


def rates(steps, accepted, trials, guard_rejects, level, dt):
    """What the steps did, so the headline does not hinge on how many of them were rejected: accepted steps/s and
    evaluations/s (an evaluation = one energy pass E(x + alpha d) of a line search, or one energy + gradient
    evaluation at a new x; at level 0 the reference's repeated passes count as well)."""
    e_evals = trials + guard_rejects + (steps if level == 0 else 0)
    g_evals = steps if level == 0 else max(accepted, 1)
    return {"accepted_steps_per_s": accepted / dt, "evaluations_per_s": (e_evals + g_evals) / dt,
            "evaluations": {"line_search_energy_passes": e_evals, "energy_plus_gradient": g_evals}}


def config5_deck(device, steps=40):
    """BASELINE configs[4]: the caveolin deck as the reference parses it (tests/golden/traj_config5_deck_gd.npz --
    positions, rows, flags, parameters and module list; its three constraint modules are outside the hot path), one
    coupled relax_leaflet_tilts call with the deck's inner steps and the deck's `g` steps."""
    import torch

    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    g = np.load(os.path.join(ROOT, "tests", "golden", "traj_config5_deck_gd.npz"), allow_pickle=False)
    mods = [str(m) for m in g["modules"]]
    mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"], tilts_in=g["tilts_in0"],
                     tilts_out=g["tilts_out0"], tilt_fixed_in=g["tilt_fixed_in"], tilt_fixed_out=g["tilt_fixed_out"],
                     global_parameters=json.loads(str(g["gp_json"])), energy_modules=mods, constraint_modules=[])
    mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
    mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                   ConstraintModuleManager([]), quiet=True, step_size=float(g["step_size0"]), device=device)
    _mir, dm = mz._device()
    mz._relax_tilts(dm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mz._relax_tilts(dm)
    torch.cuda.synchronize()
    t_rel = (time.perf_counter() - t0) / 5
    mz.minimize(4, sync_mesh=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mz.minimize(steps, sync_mesh=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gp = mesh.global_parameters
    return {"workload": f"config 5 deck ({len(g['positions0'])} vertices, {len(g['tri'])} facets): " + " + ".join(mods)
                        + f"; tilt_solve_mode {gp.get('tilt_solve_mode')}, {gp.get('tilt_inner_steps')} inner steps",
            "value": steps / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt / steps, "relaxation_ms": 1e3 * t_rel,
            "steps": steps, "energy_end": float(res["energy"]),
            "one_workgroup_interpreter": dm.exec_stats(),
            "note": ("one tile: the library records its launches and runs them in ONE workgroup (k_exec), each tilt "
                     "relaxation as one launch with the reference's control flow in the workgroup; what a record costs is "
                     "its chain of dependent memory round trips (~1.3-3 us), not a launch any more")}


def secondary_config(name, freq, mods, cons, stepper_name, *, volume_row, step_size, steps, warmup, device):
    """One more configuration of BASELINE.json next to the headline, as an object of the same JSON line: steps/s,
    accepted steps/s, what the line searches did, and the gradient kernel instance of that module set with its
    fraction of the HBM peak (HIP events over a profiled pass of the same steps)."""
    import torch

    from membrane_solver_amd.geometry.mesh import ArrayBody, ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent

    P, T = bench_mesh(freq)
    nv, nf = P.shape[0], T.shape[0]
    gp = dict(GP)
    bodies = []
    if volume_row:
        v0, v1, v2 = P[T[:, 0]], P[T[:, 1]], P[T[:, 2]]
        bodies = [ArrayBody(0, None, float(np.einsum("ij,ij->i", np.cross(v1, v2), v0).sum() / 6.0))]
    mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=list(mods), constraint_modules=list(cons), bodies=bodies)
    stepper = ConjugateGradient() if stepper_name == "conjugate_gradient" else GradientDescent()
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(list(mods)),
                   ConstraintModuleManager(list(cons)), quiet=True, step_size=step_size, device=device)
    mz.compute_energy()
    mz.minimize(warmup, sync_mesh=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mz.minimize(steps, sync_mesh=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    run = dict(mz.last_run)
    dm = mesh._hip_mirror.dm
    out = {"workload": f"class-I icosphere f={freq} (nv={nv}, nf={nf}), modules {'+'.join(mods)}"
                       + (", volume Lagrange row (k = 1 KKT projection of the gradient)" if volume_row else "")
                       + f", {stepper_name}, initial step {step_size:g}",
           "value": steps / dt, "unit": "steps/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
           "steps_accepted": res["steps_accepted"], "line_search_trials": res["line_search_trials"],
           **rates(steps, res["steps_accepted"], res["line_search_trials"], run.get("guard_rejects", 0), 2, dt)}
    rs = dm.resident_stats()
    if rs["steps"] == 0 and rs["co_resident"] >= 0:
        out["resident_step_kernel"] = dict(rs, note="eligible module set, but no step ran in the resident kernel")
    if rs["steps"] > 0:
        out["resident_step_kernel"] = dict(rs, note=(
            "the timed steps ran in the resident kernel (csrc/ms_resident.inc: one launch for many steps, a workgroup per "
            "tile, grid barriers between the phases); the per-kernel figures below are of the kernel-per-phase path, "
            "measured in a separate pass with per-kernel timing on (which switches the resident kernel off)"))
    n_prof = min(steps, 40)
    dm.profile_enable(True)
    dm.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mz.minimize(n_prof, sync_mesh=False)
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    prof = dm.profile_read()
    dm.profile_enable(False)
    ab = algorithmic_bytes(nv, nf, uniform=True, volume=volume_row)
    bend = "bending" in mods
    inst = {}
    for kind in ("gradient", "gradient_lean"):
        ms, n = prof.get(kind, (0.0, 0))
        if n:
            lean = "true" if kind == "gradient_lean" else "false"
            row = "true" if volume_row else "false"
            if bend:
                # with a constraint row the direction is not fused: x, factors, flags in; g and gC out
                nbytes = ab["gradient"] if volume_row else ab[kind]
                name = f"ms::k_gradient<1, {row}, 256, 0, true, {lean}>"
            else:
                nbytes = 12 * nf + 25 * nv + 48 * nv
                name = f"ms::k_gradient<0, {row}, 256, 0, true, false>"
            us = 1e3 * ms / n
            inst[kind] = {"kernel": name, "avg_us": us, "launches": n, "algorithmic_bytes": nbytes,
                          "GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_hbm_peak": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    out["gradient_instance"] = inst
    tot_ms = sum(ms for ms, _n in prof.values())
    out["kernel_busy_share"] = tot_ms * 1e-3 / dtp
    out["kernels_avg_us"] = {k: 1e3 * ms / n for k, (ms, n) in prof.items() if n}
    out["line_search_queue"] = {k: v for k, v in dm.queue_stats().items()}
    dm.close()
    mesh._hip_mirror = None
    return out


def main_single(args):
    import torch  # device plumbing: synchronize bracket of the contract

    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.geometry.mesh import ArrayBody, ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if L.lib().ms_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: " + L.lib().ms_last_error(None).decode())
    torch.cuda.set_device(local_rank)

    P, T = bench_mesh(args.freq)
    nv, nf = P.shape[0], T.shape[0]
    gp = dict(GP)
    mods, cons, bodies = ["surface", "bending"], [], []
    if args.volume:
        cons = ["volume"]
        v0, v1, v2 = P[T[:, 0]], P[T[:, 1]], P[T[:, 2]]
        bodies = [ArrayBody(0, None, float(np.einsum("ij,ij->i", np.cross(v1, v2), v0).sum() / 6.0))]
    mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods, constraint_modules=cons, bodies=bodies)
    stepper = ConjugateGradient()
    stepper.reuse_energy0 = args.reuse_level
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(mods),
                   ConstraintModuleManager(cons), quiet=True, step_size=args.step_size,
                   device=local_rank, tile_vertices=args.tile, deterministic=bool(args.deterministic))
    E_start = mz.compute_energy()

    # the whole loop runs inside the library (ms_minimize); it reports what the steps did
    mz.minimize(args.warmup, sync_mesh=False)
    step_size_after_warmup = mz.step_size

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mz.minimize(args.steps, sync_mesh=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    run = dict(mz.last_run)

    out = {
        "metric": METRIC, "value": args.steps / dt, "unit": "steps/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_text(args.freq, nv, nf, stepper.reuse_energy0, args.deterministic, args.volume),
                   "stepper": "conjugate_gradient", "tile_vertices": args.tile or 256,
                   "initial_step_size": args.step_size, "parallelism": "1 GPU",
                   "deterministic": bool(args.deterministic)},
        "steps_accepted": res["steps_accepted"], "line_search_trials": res["line_search_trials"],
        **rates(args.steps, res["steps_accepted"], res["line_search_trials"], run.get("guard_rejects", 0),
                int(stepper.reuse_energy0), dt),
        "energy_start": E_start, "energy_end": res["energy"],
    }
    q0 = mesh._hip_mirror.dm.queue_stats()
    out["line_search_queue"] = dict(q0, note="since ms_create (warm-up included): rounds = energy launches queued with their "
                                            "gated followers; every decision is taken once on the device and replayed by "
                                            "the host from the same doubles -- mismatches must be 0; ahead = rounds queued for a step that had not "
                                            "started yet (behind the running gradient pass), adopted = those the step took")

    # -- the same K steps with every pass the reference repeats (reuse level 0): energy0
    #    re-evaluated, a fresh energy/factor pass after every accepted step, a full gradient
    #    pass after a failed search.  Same doubles out (tests/test_gpu_minimizer.py), more launches.
    if args.reuse_level != 0 and not args.headline_only:
        stepper.reuse_energy0 = 0
        mz.minimize(5, sync_mesh=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mz.minimize(args.steps, sync_mesh=False)
        torch.cuda.synchronize()
        dt0 = time.perf_counter() - t0
        out["all_reference_passes_repeated"] = {"value": args.steps / dt0, "unit": "steps/s",
                                                "ms_per_step": 1e3 * dt0 / args.steps, "reuse_level": 0,
                                                "steps_accepted": int(mz.last_run["accepted"]),
                                                "line_search_trials": int(mz.last_run["trials"]),
                                                "window": "continues the headline run (5 untimed steps, then K): past "
                                                          "the cold phase of the line searches, not the headline's window"}
        stepper.reuse_energy0 = args.reuse_level

    # -- the same K steps with fixed-order (bitwise reproducible) vertex sums -------------------
    dm = mesh._hip_mirror.dm
    if not args.deterministic and not args.headline_only:
        mz.deterministic = True
        mz.minimize(5, sync_mesh=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mz.minimize(args.steps, sync_mesh=False)
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t0
        out["deterministic_mode"] = {"value": args.steps / dtd, "unit": "steps/s",
                                     "ms_per_step": 1e3 * dtd / args.steps,
                                     "steps_accepted": int(mz.last_run["accepted"]),
                                     "line_search_trials": int(mz.last_run["trials"]),
                                     "window": "continues the headline run (5 untimed steps, then K): past the cold "
                                               "phase of the line searches, not the headline's window"}
        mz.deterministic = False
        mz.minimize(2, sync_mesh=False)

    # -- roofline: HIP events inside the library around every launch --------------------------
    if not args.no_roofline:
        n_prof = min(args.steps, 40)
        dm.profile_enable(True)
        dm.profile_read()
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        mz.minimize(n_prof, sync_mesh=False)
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - t0p
        trial_passes = mz.last_run["trials"] + mz.last_run["guard_rejects"]
        prof = dm.profile_read()
        dm.profile_enable(False)
        out["kernel_busy_share"] = {"value": sum(ms for ms, _n in prof.values()) * 1e-3 / dt_prof,
                                    "what": "sum of the kernels' HIP-event durations / wall time of the same profiled "
                                            f"{n_prof} steps (the event brackets lengthen both a little)"}
        ab = algorithmic_bytes(nv, nf, uniform=True, volume=args.volume,
                               lean_pairs=os.environ.get("MS_PAIR_LEAN", "1") != "0")
        kernels = kernel_table(prof, ab, trial_passes=trial_passes, level=int(stepper.reuse_energy0), nv=nv,
                               deterministic=bool(args.deterministic))
        out["roofline"] = roofline_block(kernels, n_prof)
        e_kind = "energy" if "energy" in kernels else None
        g_kind = max((k for k in ("gradient", "gradient_lean") if k in kernels),
                     key=lambda k: kernels[k]["launches"], default=None)
        eg = None
        if e_kind and g_kind:
            pair_us = kernels[e_kind]["avg_us"] + kernels[g_kind]["avg_us"]
            # bytes of the single energy launches as they were actually launched (at reuse level 2 these are trial
            # passes that also write the factors: the accepted one IS the next step's energy pass)
            pair_bytes = kernels[e_kind]["algorithmic_bytes"] + ab[g_kind]
            eg = {"us": pair_us, "algorithmic_bytes": pair_bytes, "GBps": pair_bytes / (pair_us * 1e-6) / 1e9,
                  "frac": pair_bytes / (pair_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "kernels": [kernels[e_kind]["kernel"], kernels[g_kind]["kernel"]],
                  "what": "one single-evaluation energy launch (average of those profiled) + one gradient pass: the "
                          "per-step energy+gradient evaluation the north star names"}
        out["kernels"] = kernels
        out["energy_plus_gradient_evaluation"] = eg
        if out["roofline"] is not None:
            out["roofline"]["energy_plus_gradient_evaluation_frac"] = eg["frac"] if eg else None

    # -- the other single-GPU configurations of BASELINE.json, same line -------------------------
    if not args.headline_only and not args.volume and args.freq == 320:
        try:
            # (2000 steps, 50 ms: a call of the resident step kernel has ~0.4 ms of fixed cost -- the launch, the result
            # and step-log copies, the Python layer -- which 200 steps of 24 us would not amortise)
            out["config2"] = secondary_config("config2", 81, ["surface"], ["volume"], "gradient_descent", volume_row=True,
                                              step_size=1e-3, steps=2000, warmup=30, device=local_rank)
            out["config3_volume_row"] = secondary_config("config3_volume_row", 320, ["surface", "bending"], ["volume"],
                                                         "conjugate_gradient", volume_row=True, step_size=args.step_size,
                                                         steps=min(args.steps, 100), warmup=min(args.warmup, 30),
                                                         device=local_rank)
            out["config3_volume_row"]["vs_no_row"] = out["config3_volume_row"]["value"] / out["value"]
        except Exception as exc:  # secondary figures never cost the headline line
            print(f"[bench] secondary configurations skipped: {exc!r}", file=sys.stderr)

    # -- BASELINE configs[4] on its own deck (204 facets: launches and host round trips, not kernels) -------------
    if not args.headline_only and not args.volume and args.freq == 320:
        try:
            out["config5_deck"] = config5_deck(local_rank)
        except Exception as exc:
            print(f"[bench] config 5 deck leg skipped: {exc!r}", file=sys.stderr)

    # -- the N = 1 point of the 16-million-facet strong-scaling curve (bench.py --gpus N reports the others) --------
    if not args.headline_only and not args.no_large and not args.volume and args.freq == 320:
        try:
            lg = secondary_config("strong_16M_facets", LARGE_FREQ, ["surface", "bending"], [], "conjugate_gradient",
                                  volume_row=False, step_size=args.step_size * (320.0 / LARGE_FREQ) ** 2,
                                  steps=40, warmup=10, device=local_rank)
            lg["facets_per_gpu"] = 20 * LARGE_FREQ * LARGE_FREQ
            out["strong_16M_facets"] = lg
        except Exception as exc:
            print(f"[bench] 16 M-facet leg skipped: {exc!r}", file=sys.stderr)

    # -- CPU baseline: the oracle port on the host cores, bounded sample ---------
    n_cpu = 6 if args.cpu_steps < 0 else args.cpu_steps
    if n_cpu > 0:
        out["cpu_baseline"] = cpu_baseline(dm.get_positions(), T, mods, cons, bodies, gp, mz.step_size, n_cpu)
        out["step_size_after_warmup"] = step_size_after_warmup
    print(json.dumps(out))


LARGE_FREQ = 905  # 16 380 500 facets: 2 M facets per GPU at 8 GPUs (weak_frequency(8))


def sharded_extra_leg(args, rank, world, local_rank, freq, steps, warmup, exchange="rccl"):
    """One more timed run of the sharded driver at another mesh size or with the peer-to-peer exchange (same process
    group): -> dict or None."""
    import torch
    import torch.distributed as dist

    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.parallel import HipShardBackend, LibraryShardedStepper, ShardedStepper

    P, T = shared_bench_mesh(freq, rank, world, dist.barrier)
    nv, nf = P.shape[0], T.shape[0]
    be = HipShardBackend(P, T, rank=rank, world=world, device=local_rank, tile_vertices=args.tile)
    be.configure(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, gamma=np.ones(nf), kappa=np.ones(nv), c0=np.zeros(nv))
    ok = 1
    try:
        if exchange == "peer":
            be.enable_peer_exchange()  # pack kernels write into the peers' slabs; no collective in the step
        else:
            if os.environ.get("MS_SHARD_PYTHON_DRIVER"):
                raise RuntimeError("MS_SHARD_PYTHON_DRIVER set")
            be.enable_library_driver()
    except Exception as exc:
        ok = 0
        print(f"[bench] rank {rank}: {exchange} exchange not available: {exc!r}", file=sys.stderr)
    flag = torch.tensor([ok], dtype=torch.int32, device=be.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    lib = int(flag.item()) == 1
    if exchange == "peer" and not lib:
        be.dm.close()
        return {"note": "peer-to-peer exchange could not be set up on every rank"}
    drv = LibraryShardedStepper(be, stepper=L.MS_STEPPER_CG) if lib else ShardedStepper(be, stepper=L.MS_STEPPER_CG)
    step = args.step_size * (320.0 / freq) ** 2  # (the stable step scales with h^2)

    def run(n):
        nonlocal step
        acc = trials = 0
        if lib:
            o = drv.run(n, step, tol=1e-6)
            step = float(o.step_size)
            return int(o.accepted), int(o.trials)
        for _ in range(n):
            r = drv.step(step, tol=1e-6)
            step = r.next_step
            acc += int(r.success)
            trials += r.trials
            if not r.success:
                drv.reset()
        return acc, trials

    run(warmup)
    ex0 = drv.exchanges
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, trials = run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    out = {"workload": f"class-I icosphere f={freq} (nv={nv}, nf={nf}), surface + Helfrich bending, CG, sharded over {world} GPU(s)",
           "value": steps / dt, "unit": "steps/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
           "steps_accepted": acc, "line_search_trials": trials, "facets_per_gpu": nf / world,
           "exchanges_per_step": (drv.exchanges - ex0) / max(steps, 1),
           "driver": "library" if lib else "python",
           "exchange": ("peer-to-peer: pack kernels store into the peers' IPC-mapped slabs, flag words, bounded wait"
                        if exchange == "peer" else "RCCL all-gather")}
    if exchange == "peer" and lib:
        out["peer_memory"] = be.dm.peer_memory_kind()
        cs = be.dm.shard_chain_stats()
        out["device_side_decisions"] = dict(cs, note="trials whose Armijo decision was ALSO taken on the device from the ranks' "
                                            "headers (the host replays it), with the commit, the gradient + direction pass of "
                                            "the accepted point and its exchange queued behind the decision word: queued / ran "
                                            "(main trial accepted) / adopted by the next step / dropped; ahead_*: first trials of "
                                            "the search two steps on, queued behind the chain with a device-side test whether that "
                                            "search happens; since ms_create")
    be.dm.close()
    return out


def headline_from_peer_leg(peer, steps, warmup, rccl_value, weak=False, pin=""):
    """bench.py --gpus N runs the same steps through both library drivers; the line's value is the peer-exchange leg's
    only if that leg is the same measurement (library driver, same steps and warm-up, strong scaling) and was faster."""
    return bool(isinstance(peer, dict) and peer.get("driver") == "library" and peer.get("steps") == steps and
                peer.get("warmup") == warmup and float(peer.get("value", 0.0) or 0.0) > float(rccl_value) and
                pin != "rccl" and not weak)


def main_sharded(args, rank, world, local_rank):
    """--gpus N (N > 1, or MS_BENCH_FORCE_SHARDED=1 at N = 1): tiles sharded over the ranks, one process per GPU,
    halo all-gathers over RCCL; timed with barrier + synchronize on both sides and the MAX over ranks."""
    import torch
    import torch.distributed as dist

    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.parallel import HipShardBackend, LibraryShardedStepper, ShardedStepper

    torch.cuda.set_device(local_rank)
    # RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line there.
    # Everything until the result line goes to stderr at the file-descriptor level (the banner comes from C code).
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if "WORLD_SIZE" not in os.environ:  # MS_BENCH_FORCE_SHARDED=1 at N = 1 without a launcher
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "RANK": "0",
                           "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    budget = LegBudget()
    t_leg0 = time.perf_counter()
    freq = weak_frequency(world, args.freq) if args.weak else args.freq
    P, T = shared_bench_mesh(freq, rank, world, dist.barrier)
    nv, nf = P.shape[0], T.shape[0]
    be = HipShardBackend(P, T, rank=rank, world=world, device=local_rank, tile_vertices=args.tile)
    if args.deterministic:
        be.dm.set_deterministic(True)
    be.configure(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, gamma=np.ones(nf), kappa=np.ones(nv), c0=np.zeros(nv))
    # every rank must end up with the SAME driver: the library driver's collectives run on its own communicator
    ok, why = 1, ""
    try:
        if os.environ.get("MS_SHARD_PYTHON_DRIVER"):
            raise RuntimeError("MS_SHARD_PYTHON_DRIVER set")
        be.enable_library_driver()
    except Exception as exc:
        ok, why = 0, str(exc)
    flag = torch.tensor([ok], dtype=torch.int32, device=be.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        driver = "library (ms_shard_step, direct ncclAllGather)"
        drv = LibraryShardedStepper(be, stepper=L.MS_STEPPER_CG)
    else:
        if rank == 0 or not ok:
            print(f"[bench] rank {rank}: library shard driver not used on every rank ({why or 'a peer failed'}); "
                  "all ranks use the Python / torch.distributed driver", file=sys.stderr)
        driver = "python (ShardedStepper, torch.distributed all_gather_into_tensor)"
        drv = ShardedStepper(be, stepper=L.MS_STEPPER_CG)
    step = args.step_size

    def run(n):
        nonlocal step
        acc = trials = guards = 0
        if isinstance(drv, LibraryShardedStepper):
            o = drv.run(n, step, tol=1e-6)
            step = float(o.step_size)
            return int(o.accepted), int(o.trials), int(o.guard_rejects), o
        r = None
        for _ in range(n):
            r = drv.step(step, tol=1e-6)
            step = r.next_step
            acc += int(r.success)
            trials += r.trials
            guards += r.guard_rejects
            if not r.success:
                drv.reset()  # minimizer.py:1462-1464
        return acc, trials, guards, r

    run(args.warmup)
    ex0 = drv.exchanges
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc, trials, guards, r = run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    n_exchanges = drv.exchanges - ex0
    level = int(drv.reuse_energy0)
    # roofline on THIS rank's shard: HIP events inside the library over a few more steps
    roofline, kernels = None, None
    if not args.no_roofline:
        try:
            n_prof = min(args.steps, 30)
            be.dm.profile_enable(True)
            be.dm.profile_read()
            _a, tr, gd, _r = run(n_prof)
            prof = be.dm.profile_read()
            be.dm.profile_enable(False)
            info = be.dm.shard_info()
            nv_l = int(info["row1"] - info["row0"])
            nf_l = int(round(nf * nv_l / max(nv, 1)))
            ab = algorithmic_bytes(nv_l, nf_l, uniform=True, lean_pairs=False)
            # the sharded trial passes write no trial positions (the accepted step is committed in place)
            for key in ("energy_trial", "energy_trial_factors"):
                ab[key] -= 24 * nv_l
            ab["energy_pair"] -= 2 * 24 * nv_l
            ab["energy_triple"] -= 3 * 24 * nv_l
            kernels = kernel_table(prof, ab, trial_passes=tr + gd, level=level, nv=nv_l,
                                   deterministic=bool(args.deterministic))
            roofline = roofline_block(kernels, n_prof)
            if roofline is not None:
                roofline["per_gpu"] = True
                roofline["shard"] = {"rank": rank, "owned_vertices": nv_l, "facets_equivalent": nf_l}
                roofline["traffic_note"] = ("profiles/ holds single-GPU PMC passes only: at N > 1 `traffic` is the "
                                            "full-mesh figure of the same instantiation, not this shard's")
        except Exception as exc:  # the measurement is an extra: never lose the bench line over it
            print(f"[bench] roofline measurement skipped: {exc}", file=sys.stderr)
    rccl_ranks = be.dm.shard_comm_ranks() if isinstance(drv, LibraryShardedStepper) else dist.get_world_size()
    cpu = None
    n_cpu = (6 if world == 1 else 2) if args.cpu_steps < 0 else args.cpu_steps
    if n_cpu > 0:
        x_full = be.gather_positions()  # collective: every rank takes part
        if rank == 0:
            cpu = cpu_baseline(x_full, T, ["surface", "bending"], [], [], dict(GP), step, n_cpu)
        dist.barrier()
    # -- one SCALE invocation yields all three curves: the headline mesh (strong), 16 M facets (strong, the size at
    #    which 8 GPUs have 2 M facets each) and 2 M facets per GPU (weak); the latter two coincide at 8 GPUs
    budget.observe("headline", nf, time.perf_counter() - t_leg0)
    if rank == 0:
        print(f"[bench] leg headline (f={freq}): {budget.log[-1]['wall_s']} s; {budget.elapsed():.0f} s since process start, "
              f"budget {budget.budget_s:.0f} s", file=sys.stderr)
    extra = {}
    if not args.weak and args.freq == 320:
        be.dm.close()
        for key, f in (("peer_exchange", 320), ("strong_16M_facets", LARGE_FREQ), ("strong_16M_facets_peer_exchange", LARGE_FREQ),
                       ("weak_2M_facets_per_gpu", weak_frequency(world, 320))):
            if args.no_large and f != 320:
                continue
            reuses = key.startswith("weak") and ((f == LARGE_FREQ and "strong_16M_facets" in extra) or f == 320)
            # rank 0's clock decides for everybody (a leg is a collective)
            go = torch.tensor([1 if (reuses or budget.allows(20 * f * f)) else 0], dtype=torch.int32, device=be.device)
            dist.broadcast(go, src=0)
            if int(go.item()) == 0:
                extra[key] = budget.skip_note(key, 20 * f * f)
                if rank == 0:
                    print(f"[bench] leg {key} (f={f}) {extra[key]['note']}", file=sys.stderr)
                continue
            t_leg = time.perf_counter()
            try:
                if key.endswith("peer_exchange"):
                    extra[key] = sharded_extra_leg(args, rank, world, local_rank, f, args.steps if f == 320 else min(args.steps, 40),
                                                   args.warmup if f == 320 else min(args.warmup, 10), exchange="peer")
                elif key.startswith("weak") and f == LARGE_FREQ and "strong_16M_facets" in extra:
                    extra[key] = dict(extra["strong_16M_facets"], note="same run as strong_16M_facets at 8 GPUs")
                elif key.startswith("weak") and f == 320:
                    extra[key] = {"note": "the headline run itself at 1 GPU"}
                else:
                    extra[key] = sharded_extra_leg(args, rank, world, local_rank, f, min(args.steps, 40), min(args.warmup, 10))
            except Exception as exc:  # an extra leg never costs the headline line
                print(f"[bench] rank {rank}: leg {key} skipped: {exc!r}", file=sys.stderr)
                extra[key] = None
            if not reuses:
                budget.observe(key, 20 * f * f, time.perf_counter() - t_leg)
                if rank == 0:
                    print(f"[bench] leg {key} (f={f}): {budget.log[-1]['wall_s']} s; {budget.elapsed():.0f} s since "
                          f"process start", file=sys.stderr)
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        # Two drivers ran the same steps of the same workload with the same timing brackets: the RCCL all-gather driver
        # above and (leg "peer_exchange") the peer-to-peer one.  The line's value is the faster one's; the other keeps
        # its figures under its own key.  MS_BENCH_HEADLINE=rccl pins the all-gather driver.
        peer = extra.get("peer_exchange")
        use_peer = headline_from_peer_leg(peer, args.steps, args.warmup, args.steps / dt, args.weak,
                                          os.environ.get("MS_BENCH_HEADLINE", ""))
        par_text = (f"tiles (facet blocks) sharded over {world} GPUs; per exchange one RCCL "
                    f"all-gather of [{L.MS_NSCAL} scalars | <= {be.boundary['max_rows']} boundary rows] "
                    f"per rank" + ("" if be.exchange_mode == "halo" else
                                   " -- MS_SHARD_EXCHANGE=dense: scalars only, and a dense RCCL "
                                   "all-reduce of every exchanged per-vertex vector (the simple "
                                   "mode kept for comparison)") + f"; driver: {driver}")
        if use_peer:
            extra["rccl_all_gather_driver"] = {"value": args.steps / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt / args.steps,
                                               "steps_accepted": acc, "line_search_trials": trials,
                                               "exchanges_per_step": n_exchanges / max(args.steps, 1),
                                               "parallelism": par_text}
            dt = args.steps / float(peer["value"])
            acc, trials = int(peer["steps_accepted"]), int(peer["line_search_trials"])
            n_exchanges = int(round(float(peer["exchanges_per_step"]) * args.steps))
            par_text = (f"tiles (facet blocks) sharded over {world} GPUs; per exchange the pack kernel stores "
                        f"[{L.MS_NSCAL} scalars | <= {be.boundary['max_rows']} boundary rows] straight into every peer's "
                        f"IPC-mapped slab over xGMI, flag words order it (no collective in the step); a trial's Armijo "
                        f"decision is also taken on the device and the next gradient pass runs behind it; driver: library "
                        f"(ms_shard_step); the RCCL all-gather driver's figures: rccl_all_gather_driver")
        print(json.dumps({
            **extra,
            "metric": METRIC, "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text(freq, nv, nf, level, args.deterministic)
                                   + (f"; WEAK scaling: ~2 048 000 facets per GPU, {nf} in all" if args.weak else ""),
                       "parallelism": par_text,
                       "tile_vertices": args.tile or 256, "initial_step_size": args.step_size,
                       "deterministic": bool(args.deterministic)},
            "steps_accepted": acc, "line_search_trials": trials,
            **rates(args.steps, acc, trials, guards, level, dt),
            "rccl_ranks": int(rccl_ranks), "exchanges": int(n_exchanges),
            "exchanges_per_step": n_exchanges / max(args.steps, 1),
            "exchange_bytes_per_rank_max": int((L.MS_NSCAL + 10 * be.boundary["max_rows"]) * 8),
            "energy_end": float(getattr(r, "energy", getattr(r, "energy_eval", float("nan")))),
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
            "legs": {"budget_s": budget.budget_s, "wall_s_total": round(budget.elapsed(), 1), "log": budget.log},
        }), flush=True)
    os.dup2(2, 1)  # (communicator teardown may print as well)
    dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = build_parser().parse_args(argv)
    have_rendezvous = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and not have_rendezvous:
        return spawn_ranks(args, argv)  # before anything touches the GPU
    if have_rendezvous and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 or os.environ.get("MS_BENCH_FORCE_SHARDED"):
        return main_sharded(args, rank, world, local_rank)
    return main_single(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
