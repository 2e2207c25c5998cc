#!/usr/bin/env python3
"""What a peer-to-peer exchange costs between two PROCESSES: the sharded library driver with the peer exchange on a small
mesh (kernels of a few microseconds: the step is exchanges and launch latency), world 1 in one process against world 2 in
two processes that share this box's one GPU (hipIpc-mapped slabs, flag words, bounded waits; no RCCL).
usage: python3 tools/peer_two_process.py [FREQ=24] [STEPS=400]"""
import json, os, socket, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FREQ = int(sys.argv[1]) if len(sys.argv) > 1 else 24
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400

RANK_SCRIPT = r"""
import json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["MS_ROOT"])
from membrane_solver_amd import _lib as L, meshgen
from membrane_solver_amd.parallel import HipShardBackend, LibraryShardedStepper

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
freq, steps = int(os.environ["MS_FREQ"]), int(os.environ["MS_STEPS"])
dist.init_process_group(backend="gloo")
P, T = meshgen.icosphere(freq)
P = meshgen.smooth_displace(P, 0.06)
nv, nf = len(P), len(T)
be = HipShardBackend(P, T, rank=rank, world=world, device=0)
be.configure(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, gamma=np.ones(nf), kappa=np.ones(nv), c0=np.zeros(nv))
be.enable_peer_exchange()
drv = LibraryShardedStepper(be, stepper=L.MS_STEPPER_CG, reuse_energy0=2)
step = 1e-3 * (24.0 / freq) ** 2
o = drv.run(50, step, tol=1e-12)
step = float(o.step_size)
torch.cuda.synchronize(); dist.barrier()
ex0 = drv.exchanges
t0 = time.perf_counter()
o = drv.run(steps, step, tol=1e-12)
torch.cuda.synchronize(); dist.barrier()
dt = time.perf_counter() - t0
print("RESULT " + json.dumps({"rank": rank, "world": world, "us_per_step": 1e6 * dt / steps, "accepted": int(o.accepted),
                              "trials": int(o.trials), "exchanges_per_step": (drv.exchanges - ex0) / steps}), flush=True)
dist.destroy_process_group()
"""


def run(world):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MS_ROOT=ROOT, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", MS_FREQ=str(FREQ),
                   MS_STEPS=str(STEPS))
        procs.append(subprocess.Popen([sys.executable, "-c", RANK_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        if p.returncode != 0:
            raise SystemExit(e[-3000:])
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    return outs


if __name__ == "__main__":
    one = run(1)[0]
    two = run(2)
    print(json.dumps({"freq": FREQ, "steps": STEPS, "world1": one, "world2": two}))
    t1, t2 = one["us_per_step"], max(r["us_per_step"] for r in two)
    ex = two[0]["exchanges_per_step"]
    print(f"world 1: {t1:.1f} us/step; world 2 (two processes, one GPU): {t2:.1f} us/step; {ex:.2f} exchanges per step "
          f"-> {(t2 - t1) / max(ex, 1e-9):+.1f} us per exchange on top of world 1's")
