#!/usr/bin/env python3
"""Headline benchmark: minimizer steps/sec (energy + gradient + CG) on the
2 048 000-facet icosphere (BASELINE.json configs[2]; configs[3] when --gpus > 1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one iteration of Minimizer.minimize (runtime/minimizer.py:1230-1515
of the reference): energy+gradient of all modules, fixed-row zeroing, per-row
Polak-Ribiere CG direction, Armijo backtracking line search (energy0 is
re-evaluated like line_search.py:294, then >= 1 trial energy evaluation) and the
position commit.  All state is resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--freq", type=int, default=320, help="icosphere frequency (320 -> 2 048 000 facets)")
    ap.add_argument("--tile", type=int, default=0, help="owned vertices per tile (0 = library default)")
    ap.add_argument("--step-size", type=float, default=1e-6)
    ap.add_argument("--volume", action="store_true", help="add the volume Lagrange constraint row")
    ap.add_argument("--cpu-steps", type=int, default=6, help="CPU oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary sections (reuse level 0, deterministic mode): for kernel traces")
    ap.add_argument("--reuse-level", type=int, default=2, choices=(0, 1, 2),
                    help="ms_stepper_params.reuse_energy0: 0 repeats every pass the reference repeats, "
                         "2 (library default) never repeats a pass whose result is already on the device")
    ap.add_argument("--deterministic", action="store_true",
                    help="fixed-order per-vertex sums (ms_set_deterministic): bitwise reproducible run to "
                         "run, slower than the default LDS-atomic accumulation")
    return ap.parse_args()


def algorithmic_bytes(nv, nf, volume=False):
    """Compulsory HBM bytes per launch (DESIGN.md 'Kernels'); each array touched once."""
    return {
        # tri rows 12 B + gamma 8 B per facet; x 24 + kappa,c0 16 + flags 1 per vertex; fK,fA 40 out
        "energy_factors": 20 * nf + (24 + 16 + 1) * nv + 40 * nv,
        # trial energy: x and d in (48), xt out (24), no factor write
        "energy_trial": 20 * nf + (48 + 16 + 1) * nv + 24 * nv,
        # trial energy that also writes the factors (reuse level 2: an accepted trial is the
        # next step's energy pass)
        "energy_trial_factors": 20 * nf + (48 + 16 + 1) * nv + 24 * nv + 40 * nv,
        # pair launch: two trial evaluations of one line search in one launch -- the inputs are
        # compulsory ONCE (the second evaluation's reads are meant to hit L2), the outputs twice
        "energy_pair": 20 * nf + (48 + 16 + 1) * nv + 2 * (24 + 40) * nv,
        "energy_triple": 20 * nf + (48 + 16 + 1) * nv + 3 * (24 + 40) * nv,
        # gradient (+ fused direction pass when no constraint row): x 24 + fK,fA 40 + flags 1 in,
        # g 24 and d 24 out.  The CG-history reads (pg, pd: 48 B/vertex on non-restart steps) are
        # NOT counted, so the figure is a lower bound of the compulsory traffic.
        "gradient": 20 * nf + (24 + 40 + 1) * nv + 24 * nv + (0 if volume else 24 * nv),
    }


def _template_args(name):
    i, j = name.find("<"), name.rfind(">")
    return [a.strip() for a in name[i + 1:j].split(",")] if 0 <= i < j else []


def pmc_traffic(kernel_prefix, deterministic=False, multi=0):
    """HBM bytes per launch of one kernel from the committed rocprofv3 PMC summary
    (profiles/<tag>_pmc_summary.csv; separate FETCH_SIZE / WRITE_SIZE passes of this same
    bench command).  gfx950 correction: FETCH_SIZE counts 1/2 of the fetched bytes
    (MI355X_MICROARCH.md, calibrated here on the direction pass's known byte count)."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv")))
    if not files:
        return None, None
    # dispatch-weighted mean over every instantiation of the kernel family in the summary
    acc = {"FETCH_SIZE": [0.0, 0.0], "WRITE_SIZE": [0.0, 0.0]}
    with open(files[-1]) as f:
        for row in csv.reader(line for line in f if not line.startswith("#")):
            if len(row) < 4 or kernel_prefix not in row[1] or row[0] not in acc:
                continue
            # template arguments: the fifth selects the accumulation mode (true = LDS atomics), k_energy's sixth
            # the number of evaluations per launch (0 = one, 2 = pair, 3 = triple)
            ta = _template_args(row[1])
            if len(ta) >= 5 and ta[4] != ("false" if deterministic else "true"):
                continue
            if "k_energy" in kernel_prefix and (int(ta[5]) if len(ta) >= 6 and ta[5].isdigit() else 0) != multi:
                continue
            n = float(row[2])
            acc[row[0]][0] += n * float(row[3])
            acc[row[0]][1] += n
    if not acc["FETCH_SIZE"][1] or not acc["WRITE_SIZE"][1]:
        return None, None
    fetch = acc["FETCH_SIZE"][0] / acc["FETCH_SIZE"][1]
    write = acc["WRITE_SIZE"][0] / acc["WRITE_SIZE"][1]
    return (2.0 * fetch + write) * 1024.0, os.path.basename(files[-1])


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus N > 1 through torch.distributed.run (one rank per GPU)")
    if world > 1 or os.environ.get("MS_BENCH_FORCE_SHARDED"):
        from membrane_solver_amd import parallel

        return parallel.bench_main(args, rank, world, local_rank)

    import torch  # device plumbing: barrier/synchronize bracket of the contract

    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.geometry.mesh import ArrayBody, ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient

    if L.lib().ms_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: " + L.lib().ms_last_error(None).decode())
    torch.cuda.set_device(local_rank)

    P, T = meshgen.icosphere(args.freq)
    P = meshgen.smooth_displace(P, 0.05)
    nv, nf = P.shape[0], T.shape[0]
    gp = {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
          "spontaneous_curvature": 0.0, "bending_gradient_mode": "analytic",
          "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False,
          "mesh_quality_auto_repair_enabled": False}
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
    timed = {"accepted": res["steps_accepted"], "trials": res["line_search_trials"]}
    ms_per_step = 1e3 * dt / args.steps
    value = args.steps / dt

    out = {
        "metric": "minimizer steps/sec (energy+grad+CG) on 2M-facet icosphere",
        "value": value, "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"class-I icosphere f={args.freq} (nv={nv}, nf={nf}), surface + Helfrich "
                               "bending (analytic cotan gradient), CG stepper, Armijo line search, "
                               f"evaluation reuse level {stepper.reuse_energy0} (0 = every pass the "
                               "reference re-runs, 2 = passes already on the device are not repeated; "
                               "same trajectories), per-vertex sums "
                               + ("in fixed order (bitwise reproducible)" if args.deterministic
                                  else "by LDS atomics (default; --deterministic for fixed-order sums)")
                               + (", volume Lagrange row" if args.volume else ""),
                   "stepper": "conjugate_gradient", "tile_vertices": args.tile or 256,
                   "initial_step_size": args.step_size, "parallelism": "1 GPU",
                   "deterministic": bool(args.deterministic)},
        "steps_accepted": timed["accepted"], "line_search_trials": timed["trials"],
        "energy_start": E_start, "energy_end": res["energy"],
    }

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
                                                "ms_per_step": 1e3 * dt0 / args.steps, "reuse_level": 0}
        stepper.reuse_energy0 = args.reuse_level

    # -- the same K steps with fixed-order (bitwise reproducible) vertex sums -------------------
    mir = mesh._hip_mirror
    dm = mir.dm
    if not args.deterministic and not args.headline_only:
        mz.deterministic = True
        mz.minimize(5, sync_mesh=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mz.minimize(args.steps, sync_mesh=False)
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t0
        out["deterministic_mode"] = {"value": args.steps / dtd, "unit": "steps/s",
                                     "ms_per_step": 1e3 * dtd / args.steps}
        mz.deterministic = False
        mz.minimize(2, sync_mesh=False)

    # -- roofline of the dominant kernel: HIP events inside the library --------
    if not args.no_roofline:
        n_prof = min(args.steps, 40)
        dm.profile_enable(True)
        dm.profile_read()
        mz.minimize(n_prof, sync_mesh=False)
        stats = {"trial_passes": mz.last_run["trials"] + mz.last_run["guard_rejects"]}
        prof = dm.profile_read()
        dm.profile_enable(False)
        ab = algorithmic_bytes(nv, nf, args.volume)
        kernels = {}
        for kind, (ms, n) in prof.items():
            if n:
                kernels[kind] = {"avg_us": 1e3 * ms / n, "launches": n, "share_of_profiled_ms": ms}
        tot = sum(v["share_of_profiled_ms"] for v in kernels.values()) or 1.0
        for v in kernels.values():
            v["share_of_profiled_ms"] = v["share_of_profiled_ms"] / tot
        # per-launch algorithmic bytes: energy launches are a mix of plain passes at x (with or
        # without the factor write) and trial passes -> weight by what was actually launched
        n_e = prof["energy"][1]
        n_g = prof["gradient"][1]
        n_p = prof.get("energy_pair", (0.0, 0))[1]
        if n_p:
            kernels["energy_pair"]["algorithmic_bytes"] = ab["energy_pair"]
            kernels["energy_pair"]["GBps"] = ab["energy_pair"] / (kernels["energy_pair"]["avg_us"] * 1e-6) / 1e9
            kernels["energy_pair"]["evaluations_per_launch"] = 2
        n_t = prof.get("energy_triple", (0.0, 0))[1]
        if n_t:
            kernels["energy_triple"]["algorithmic_bytes"] = ab["energy_triple"]
            kernels["energy_triple"]["GBps"] = ab["energy_triple"] / (kernels["energy_triple"]["avg_us"] * 1e-6) / 1e9
            kernels["energy_triple"]["evaluations_per_launch"] = 3
        if n_e:
            n_trial = min(n_e, max(0, stats["trial_passes"] - 2 * n_p - 3 * n_t))
            n_plain = n_e - n_trial
            level = int(stepper.reuse_energy0)
            n_fact = min(n_plain, n_g)          # factor-writing passes at x
            n_e0 = n_plain - n_fact             # energy0 re-evaluations (level 0)
            trial_key = "energy_trial_factors" if level >= 2 else "energy_trial"
            e_bytes = (n_fact * ab["energy_factors"] + n_e0 * (ab["energy_factors"] - 40 * nv)
                       + n_trial * ab[trial_key]) / n_e
            kernels["energy"]["algorithmic_bytes"] = e_bytes
            kernels["energy"]["GBps"] = e_bytes / (kernels["energy"]["avg_us"] * 1e-6) / 1e9
        if n_g:
            kernels["gradient"]["algorithmic_bytes"] = ab["gradient"]
            kernels["gradient"]["GBps"] = ab["gradient"] / (kernels["gradient"]["avg_us"] * 1e-6) / 1e9
        # dominant kernel: the family (k_energy in all its instantiations vs k_gradient) with the larger share of the
        # kernel time, and within k_energy the instantiation (evaluations per launch) with the largest share --
        # that is one row of the rocprofv3 summary under profiles/
        fam_e = [k for k in ("energy", "energy_pair", "energy_triple") if k in kernels]
        share_e = sum(kernels[k]["share_of_profiled_ms"] for k in fam_e)
        share_g = kernels["gradient"]["share_of_profiled_ms"] if "gradient" in kernels else 0.0
        if fam_e and share_e >= share_g:
            dom = max(fam_e, key=lambda k: kernels[k]["share_of_profiled_ms"])
        else:
            dom = "gradient"
        ach = kernels[dom]["GBps"]
        traffic, traffic_src = pmc_traffic({"energy": "ms::k_energy", "energy_pair": "ms::k_energy",
                                            "energy_triple": "ms::k_energy", "gradient": "ms::k_gradient"}[dom],
                                           deterministic=bool(args.deterministic),
                                           multi={"energy_pair": 2, "energy_triple": 3}.get(dom, 0))
        out["roofline"] = {"bound": "hbm", "kernel": {"energy": "ms::k_energy<.., MULTI=0> (energy pass)",
                                                     "energy_pair": "ms::k_energy<.., MULTI=2> (energy pass, two "
                                                                    "trial evaluations per launch)",
                                                     "energy_triple": "ms::k_energy<.., MULTI=3> (energy pass, "
                                                                      "three trial evaluations per launch)",
                                                     "gradient": "ms::k_gradient* (gradient pass)"}[dom],
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_us": kernels[dom]["avg_us"],
                           "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes"],
                           "measured": f"HIP events around every launch over {n_prof} steps after the timed region"}
        if dom in ("energy_pair", "energy_triple"):
            # the same launch priced per evaluation: SURVEY 8(d)'s per-evaluation bytes x the 2 evaluations it
            # processes.  NOT used for `frac` above: a pair needs its inputs from HBM only once, so the compulsory
            # traffic of the launch is the smaller figure and `frac` is measured against that
            per_eval = ab["energy_trial_factors"]
            n_ev = 2 if dom == "energy_pair" else 3
            out["roofline"]["per_evaluation_equivalent"] = {
                "us_per_evaluation": kernels[dom]["avg_us"] / n_ev, "algorithmic_bytes_per_evaluation": per_eval,
                "GBps": per_eval / (kernels[dom]["avg_us"] / n_ev * 1e-6) / 1e9,
                "frac": per_eval / (kernels[dom]["avg_us"] / n_ev * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "single_launch_frac": (kernels["energy"]["GBps"] / HBM_PEAK_GBS) if "energy" in kernels else None}
        eg = None
        if "energy" in kernels and "gradient" in kernels:
            pair_us = kernels["energy"]["avg_us"] + kernels["gradient"]["avg_us"]
            pair_bytes = ab["energy_factors"] + ab["gradient"]
            eg = {"us": pair_us, "algorithmic_bytes": pair_bytes,
                  "GBps": pair_bytes / (pair_us * 1e-6) / 1e9,
                  "frac": pair_bytes / (pair_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        out["kernels"] = kernels
        out["energy_plus_gradient_evaluation"] = eg

    # -- CPU baseline: the oracle port on the host cores, bounded sample ---------
    if args.cpu_steps > 0:
        from oracle import minimizer_port as mp

        x_now = dm.get_positions()
        p = mp.Problem(positions=x_now, tri=T, energy_modules=mods, constraint_modules=cons,
                       target_volume=bodies[0].target_volume if bodies else None, gp=dict(gp))
        t0 = time.perf_counter()
        cres = mp.minimize(p, mp.ConjugateGradient(), args.cpu_steps, step_size=mz.step_size)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": args.cpu_steps / cdt, "unit": "steps/s", "cores": 1, "kind": "port",
                               "sample": f"{args.cpu_steps} minimizer steps of the C/NumPy oracle port "
                                         f"(oracle/minimizer_port.py) on the same {nf}-facet mesh, starting "
                                         f"from the GPU run's state and step size; "
                                         f"{sum(t['trials'] for t in cres['trace'])} line-search trials",
                               "seconds": cdt}
        out["step_size_after_warmup"] = step_size_after_warmup
    print(json.dumps(out))


if __name__ == "__main__":
    main()
