"""Time the reference's OWN Python path (young-how/DQN-based-UAV-3D_path_planer) and write profiles/ref_python_baseline.json.

TEST / MEASUREMENT INFRASTRUCTURE (build container only: /root/reference does not travel to the GPU box).  This is the recipe
behind `cpu_baseline.reference_python` of the bench line (SURVEY.md section 8(d) CPU baselines (i)/(ii)):
bench.py never runs it -- it READS the committed JSON and says where and when it was measured.

  (i)   env path, one thread: uav.reset() then a loop of uav.update([a, 0]) + uav.state() with a ~ U(-1, 1), re-reset when the
        agent is done (Agents/UAV.py:397-567; resets -- RRT planning, :327-366 -- timed separately and excluded from the
        step rate, included in `with_resets`);
  (ii)  the full loop as shipped: env.run_eposide(eps) with config/*.xml (SAC continuous, batch 64 -- the only trainer that
        runs end to end as shipped), rendering / DB stubbed by ref_harness (Envs/PathPlan_City.py:410-478);
  (iii) the reference's DQN learner at ITS batch (64): DQN_Trainer.learn_off_policy on injected transitions
        (Trainer/DQN_Trainer.py:85-136) -- the figure the bench line's cpu_baseline.learner quotes beside the PyTorch-CPU
        port at the GPU's batch.

Usage:  python oracle/time_reference.py [--seconds 20] [--out profiles/ref_python_baseline.json]
"""
from __future__ import annotations

import argparse
import datetime
import hashlib
import json
import os
import platform
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from ref_harness import RefSession  # noqa: E402


def host_cpu() -> dict:
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cpus": os.cpu_count(), "machine": platform.machine()}


def time_env_path(s, seconds: float) -> dict:
    """(i): update + state, single thread.  Resets (with their RRT) are timed apart."""
    uav = s.uav
    random.seed(42)
    prng = random.Random(4242)
    t_step = t_reset = 0.0
    n_step = n_reset = 0
    t_update = t_state = 0.0
    t0 = time.perf_counter()
    uav.reset()
    uav.state()
    t_reset += time.perf_counter() - t0
    n_reset += 1
    while t_step + t_reset < seconds:
        a = prng.uniform(-1.0, 1.0)
        t0 = time.perf_counter()
        uav.update([a, 0.0])
        t1 = time.perf_counter()
        uav.state()
        t2 = time.perf_counter()
        t_update += t1 - t0
        t_state += t2 - t1
        t_step += t2 - t0
        n_step += 1
        if uav.done:
            t0 = time.perf_counter()
            uav.reset()
            t_reset += time.perf_counter() - t0
            n_reset += 1
    return {"what": "uav.update([a, 0]) + uav.state() per step, a ~ U(-1, 1), one thread; Agents/UAV.py:397-567",
            "steps": n_step, "seconds_in_steps": t_step, "value": n_step / t_step, "unit": "env-steps/s",
            "update_us": 1e6 * t_update / n_step, "state_us": 1e6 * t_state / n_step,
            "resets": n_reset, "reset_ms": 1e3 * t_reset / n_reset,
            "with_resets": n_step / (t_step + t_reset), "threads": 1}


def time_full_loop(s, seconds: float) -> dict:
    """(ii): run_eposide with the shipped config (SAC continuous, batch 64)."""
    env = s.env
    random.seed(7)
    steps = updates = episodes = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        e0 = [a.Trainer.epoch for a in env.Agents]
        info = env.run_eposide(0.1)
        steps += int(info.get("step", 0)) if info.get("step") else sum(int(a.Step) for a in env.Agents)
        updates += sum(a.Trainer.epoch - b for a, b in zip(env.Agents, e0))
        episodes += 1
    dt = time.perf_counter() - t0
    tr = env.Agents[0].Trainer
    return {"what": "env.run_eposide(0.1) with the shipped config/*.xml (%s, Batch_Size %s, %d UAV), render / DB stubbed; "
                    "Envs/PathPlan_City.py:410-478" % (type(tr).__name__, getattr(tr, "Batch_Size", "?"), len(env.Agents)),
            "episodes": episodes, "steps": steps, "updates": updates, "seconds": dt,
            "value": steps / dt, "unit": "env-steps/s", "updates_per_s": updates / dt}


def time_learner(s, seconds: float) -> dict:
    """(iii): DQN_Trainer.learn_off_policy at the reference's own batch (64), CPU torch."""
    import numpy as np
    import torch
    from FactoryClass.TrainerFactory import TrainerFactory
    B = 64
    param = {"Trainer_Type": "DQN_Trainer", "NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3", "h": "1",
             "channel": "1", "Batch_Size": str(B), "LEARNING_RATE": "0.001", "gamma": "0.99", "replay_size": "10000",
             "save_loop": str(10 ** 9), "Update_loop": "3", "Is_Train": "1", "name": "timing"}
    tr = TrainerFactory().Create_Trainer(param)
    assert tr is not None
    tr.save = lambda *a, **k: None
    rng = np.random.default_rng(0)
    F = torch.FloatTensor
    for _ in range(4096):
        exp = (F(rng.random((1, 100)).astype("float32")), torch.tensor([[int(rng.integers(0, 3))]]),
               F([[float(rng.normal())]]), F(rng.random((1, 100)).astype("float32")), F([[float(rng.random() < 0.05)]]))
        tr.replay_memory.push(exp, 0)
    random.seed(3)
    tr.learn_off_policy()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        tr.learn_off_policy()
        n += 1
    dt = time.perf_counter() - t0
    return {"what": "DQN_Trainer.learn_off_policy, batch 64 drawn from 4096 stored transitions, Qnet2 100-64-3, CPU torch; "
                    "Trainer/DQN_Trainer.py:85-136", "updates": n, "seconds": dt, "value": n / dt, "unit": "learner updates/s",
            "batch": B, "threads": torch.get_num_threads(), "torch": torch.__version__}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0, help="budget per measurement")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ref_python_baseline.json"))
    a = ap.parse_args()
    out = {"script": "oracle/time_reference.py",
           "script_sha256_16": hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16],
           "measured_where": "build container (no GPU); the reference cannot run on the GPU box",
           "host_cpu": host_cpu(), "python": platform.python_version(),
           "date_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"),
           "reference": "young-how/DQN-based-UAV-3D_path_planer, executed from a scratch copy under oracle/ref_harness.py's shims"}
    s = RefSession()
    try:
        out["env_path"] = time_env_path(s, a.seconds)
        print("env path:", json.dumps(out["env_path"]))
        out["learner_batch64"] = time_learner(s, min(a.seconds, 10.0))
        print("learner:", json.dumps(out["learner_batch64"]))
        out["full_loop"] = time_full_loop(s, a.seconds)
        print("full loop:", json.dumps(out["full_loop"]))
    finally:
        s.close()
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
