"""The peer-to-peer exchange across PROCESSES: two ranks (two processes sharing this box's one GPU), their receive slabs
and flag words exchanged as hipIpcMemHandle_t over a gloo process group, every boundary row written by the other rank's
pack kernel through the IPC mapping -- no RCCL anywhere (RCCL refuses two ranks on one GPU; the peer exchange does not
need it).  Each rank's step log must equal the single-context run's."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = r"""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["MS_ROOT"])
from membrane_solver_amd import _lib as L, meshgen
from membrane_solver_amd.parallel import HipShardBackend, LibraryShardedStepper

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
P, T = meshgen.icosphere(24)
P = meshgen.smooth_displace(P, 0.06)
nv, nf = len(P), len(T)
be = HipShardBackend(P, T, rank=rank, world=world, device=0, tile_vertices=64)
be.dm.set_deterministic(True)
be.configure(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, gamma=np.full(nf, 1.1), kappa=np.full(nv, 0.9),
             c0=np.full(nv, 0.1))
be.enable_peer_exchange()
drv = LibraryShardedStepper(be, stepper=L.MS_STEPPER_CG, reuse_energy0=2)
log, step = [], 1e-3
for _ in range(12):
    r = drv.step(step, tol=1e-9)
    log.append((float(r.success), r.next_step, r.energy, r.grad_norm, int(r.trials)))
    step = r.next_step
    if not r.success:
        drv.reset()
# ... and the same loop inside the library (ms_minimize over ms_shard_step): there the first trial of the search two
# steps on is queued behind the chain as well
o = drv.run(24, step, tol=1e-9)
run = (int(o.accepted), int(o.trials), float(o.step_size), float(o.energy_eval), float(o.grad_norm))
torch.cuda.synchronize()
dist.barrier()
print("RESULT " + json.dumps({"rank": rank, "log": log, "run": run, "exchanges": drv.exchanges,
                              "chain": be.dm.shard_chain_stats()}), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("wait", ["kernel", "kernel-ahead", "kernel-host-decisions", "stream"])
def test_two_processes_exchange_through_ipc_mapped_slabs(wait):
    """wait "kernel": flag words raised by the pack kernel + bounded in-kernel wait (the default) -- and with them the
    device-side trial decisions: the commit, the gradient + direction pass and its exchange run behind the decision
    word on both ranks, the next step adopts them; "kernel-ahead": MS_SHARD_AHEAD=1 on top -- the first trial of the search
    two steps on is queued behind the chain, with a device-side test whether that search happens, and adopted by its step;
    "kernel-host-decisions": the same transport with MS_SHARD_CHAIN=0;
    "stream": MS_PEER_WAIT=stream, the flag words raised and awaited by hipStreamWriteValue64 / hipStreamWaitValue64 on
    the IPC-mapped words (host decisions)."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MS_ROOT=ROOT, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        env.pop("MS_PEER_WAIT", None)
        env.pop("MS_SHARD_CHAIN", None)
        env.pop("MS_SHARD_AHEAD", None)
        if wait == "kernel-ahead":
            env["MS_SHARD_AHEAD"] = "1"
        if wait == "stream":
            env["MS_PEER_WAIT"] = "stream"
        if wait == "kernel-host-decisions":
            env["MS_SHARD_CHAIN"] = "0"
        procs.append(subprocess.Popen([sys.executable, "-c", RANK_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        line = [ln for ln in o.splitlines() if ln.startswith("RESULT ")]
        assert line, (o[-500:], e[-2000:])
        outs.append(json.loads(line[-1][7:]))

    # single-context reference in this process
    P, T = meshgen.icosphere(24)
    P = meshgen.smooth_displace(P, 0.06)
    nv, nf = len(P), len(T)
    dm = DeviceMesh(P, T, tile_vertices=64)
    dm.set_deterministic(True)
    dm.set_surface_tension(np.full(nf, 1.1))
    dm.set_bending_params(np.full(nv, 0.9), np.full(nv, 0.1))
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    ref, step = [], 1e-3
    for _ in range(12):
        r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, tol=1e-9, reuse_energy0=2)
        ref.append((float(r.success), r.next_step, r.energy, r.grad_norm, int(r.trials)))
        step = r.next_step
        if not r.success:
            dm.reset_stepper()
    mp = L.ms_minimize_params()
    mp.stepper = L.ms_stepper_params(int(L.MS_STEPPER_CG), 10, 0.7, 1e-4, 1.5, 10.0, 10, 0.0, 2)
    mp.step_size, mp.tol = float(step), 1e-9
    mp.fixed_step_mode, mp.fixed_step = 0, float(step)
    mp.max_zero_steps, mp.step_size_floor = 10, 1e-8
    o, _ = dm.minimize(mp, 24)
    ref_run = (int(o.accepted), int(o.trials), float(o.step_size), float(o.energy_eval), float(o.grad_norm))
    dm.close()
    ref = np.array(ref)
    assert ref[:, 0].sum() >= 2
    for o in outs:
        got = np.array(o["log"])
        assert np.array_equal(got[:, 0], ref[:, 0]), (got, ref)
        assert np.array_equal(got[:, 4], ref[:, 4]), (got, ref)
        assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12)
        assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-12)
        assert np.allclose(got[:, 3], ref[:, 3], rtol=1e-9)
        assert o["exchanges"] > 0
        assert tuple(o["run"][:2]) == ref_run[:2], (o["run"], ref_run)
        assert np.allclose(o["run"][2:4], ref_run[2:4], rtol=1e-12) and np.isclose(o["run"][4], ref_run[4], rtol=1e-9)
    assert outs[0]["exchanges"] == outs[1]["exchanges"]
    assert outs[0]["chain"] == outs[1]["chain"]
    if wait in ("kernel", "kernel-ahead"):
        ch = outs[0]["chain"]
        assert ch["ran"] >= 2 and ch["adopted"] >= 1 and ch["queued"] >= ch["ran"], ch
        if wait == "kernel-ahead":
            assert ch["ahead_queued"] >= 1 and ch["ahead_adopted"] >= 1, ch
            assert ch["ahead_adopted"] + ch["ahead_dropped"] <= ch["ahead_queued"], ch
        else:
            assert ch["ahead_queued"] == 0
    else:
        assert outs[0]["chain"]["queued"] == 0
