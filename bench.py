#!/usr/bin/env python
"""bench.py -- the north-star hot path on N MI355X of one node.

One "step" = one pass of the PathPlan_City DQN hot path over the whole env batch of a rank:
    Q(s) for all envs  ->  fused epsilon-greedy  ->  fused HIP env step (update_PathPlan + state_PathPlan,
    transition written straight into the device replay ring)  ->  device replay sample  ->  one learner update
    (TD target, MSE, Adam, hard target copy; gradients all-reduced over RCCL when N > 1).
Workload at --gpus 1: BASELINE.json configs[1] -- 16 384 vectorised envs, DQN, device replay of 1 M transitions.
Multi-GPU is weak scaling: every rank owns its own 16 384-env shard + ring; the only exchange is the ~26 KB
gradient bucket per update.

A bench "step" (--steps K) is PASSES_PER_STEP = 1024 such passes, enqueued back to back by csrc/loop.hip (one C call per
bench step, three kernel launches per pass): one pass lasts ~35 us, so that the timed region stays >= 0.5 s whatever K
the driver picks (20 steps = 20 480 passes = 0.7 s; round 3 timed 88 ms, too short for a 5 s utilisation sampler to see).
`ms_per_step` is per bench step, `ms_per_pass` per pass.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (re-executes this file under
torch.distributed.run, one rank per GPU); launched by torch.distributed.run it just joins.  At N > 1 the line also carries
`ranks_bit_identical` (a checksum of every rank's weights after the timed region), the exchange actually used, its error
counters, and `ms_per_pass_no_exchange` from a short in-run leg without the exchange.

Prints ONE JSON line (rank 0).  `value` = whole-job env-steps/s of the full loop (inputs resident in HBM);
`roofline` is for the env kernel the loop launches (k_step_coop<policy>: get_action + update_PathPlan + state_PathPlan +
replay write; HBM-bound by design): `frac` prices SURVEY 8(d)'s ALGORITHMIC 604 B per agent-step, `frac_physical_stored` the
bytes the packed layout really moves and `frac_physical_counters` the PMC traffic of the committed profile -- side by side, so
that a lossless packing win is never read as bandwidth; `roofline_learner` for k_dqn_grad (the largest share of the pass, MFMA);
`cpu_baseline` times the CPU oracle port on the host cores (1 core and all cores); `other_configs` (N = 1, default
command only) carries BASELINE.json configs[2..4] and the env-only 65 536 / 262 144-agent points, each run as a child
process of this one for >= 0.5 s of timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_AGENT_STEP = 604        # 137 B read + 467 B written, obs f32
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
MFMA_F16_PEAK_TF = 2500.0              # MI355X_MICROARCH.md: bf16/f16 MFMA, dense
PASSES_PER_STEP = 1024                 # hot-path passes per bench step (see module docstring)


# ---- the line the driver parses ----------------------------------------------------------------------------------------------
# BENCH_r05.json: `parsed: null` -- the driver keeps the last 8 081 bytes of stdout and the line had grown to 21 KB.  The LAST stdout
# line is therefore a whitelisted, bounded summary (< HEADLINE_MAX_BYTES, asserted; tests/test_bench_line.py); everything else -- the
# prose, the per-config blocks -- goes to side files beside bench.py (and under gpurun_out/ when that directory exists).
HEADLINE_MAX_BYTES = 6000
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "passes_per_step", "ms_per_pass", "timed_region_ms", "learner_updates_per_s", "learner_samples_per_s",
             "env_only_steps_per_s", "links_crossed", "ranks_bit_identical", "exchange", "exchange_asked", "exchange_selftest_ms",
             "p2p_timeouts", "p2p_checksum_mismatches", "p2p_checksums_compared", "ms_per_pass_no_exchange", "rendezvous_retries",
             "bad_after_recovery")
_CONFIG_KEYS = ("workload", "envs_per_gpu", "uav_per_env", "learn_batch_per_gpu", "learn_batch_per_slot", "obs_dtype", "host_loop",
                "learner", "epsilon", "parallelism", "replay", "sample_lag")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_basis", "traffic_stale", "frac_algorithmic",
              "frac_physical_stored", "frac_physical_counters", "algorithmic_bytes_per_agent_step", "moved_bytes_per_agent_step", "counters_over_model",
              "agents_per_launch", "kernel_ms", "kernel_ms_back_to_back", "kernel_ms_rocprofv3_committed", "measured_copy_GBs",
              "frac_of_measured_copy", "algorithmic_exceeds_measured_copy")
_LEARN_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "flops_per_sample", "samples_per_launch", "kernel_ms",
               "kernel_ms_back_to_back", "kernel_ms_rocprofv3_committed", "reduce_adam_ms_rocprofv3_committed", "mfma_busy_frac_pmc",
               "counters_stale")


def _short(v, n=96):
    """strings are cut to n characters (the full text is in the side file); numbers are rounded to 6 significant digits"""
    if isinstance(v, str):
        return v if len(v) <= n else v[:n - 1] + "~"
    if isinstance(v, float):
        return float("%.6g" % v)
    if isinstance(v, (list, tuple)):
        return [_short(x, n) for x in v][:8]
    return v


def _pick(d, keys, n=96):
    return {k: _short(d[k], n) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], dict)}


def headline_line(full: dict, side_files=None) -> dict:
    """The bounded summary of a full result dict: the contract's keys, `roofline`, `roofline_learner`, `cpu_baseline`, and one short
    row per other configuration.  Pure function of `full` (tests/test_bench_line.py feeds it a committed round-5 line)."""
    out = _pick(full, _TOP_KEYS)
    cfg = full.get("config") or {}
    out["config"] = _pick(cfg, _CONFIG_KEYS, 150)
    rs = cfg.get("resets") or {}
    if rs:
        rf = rs.get("refresh") or {}
        out["config"]["resets"] = dict(_pick(rs, ("consumed_per_s", "episode_end_fraction_of_agent_steps", "planner_rows_per_s")),
                                       refresh=_pick(rf, ("rows_committed_per_s", "rows_committed", "rows_skipped_in_flight", "slices",
                                                          "every_passes", "rows_per_slice", "bank_rows", "ms_per_pass_delta")) or None)
    if "roofline" in full:
        out["roofline"] = _pick(full["roofline"], _ROOF_KEYS)
    if "roofline_learner" in full:
        out["roofline_learner"] = _pick(full["roofline_learner"], _LEARN_KEYS)
    cb = full.get("cpu_baseline")
    if cb:
        o = _pick(cb, ("value", "unit", "cores", "kind", "one_core_value"))
        o["sample"] = _short(cb.get("sample", ""), 200)
        rp = cb.get("reference_python") or {}
        o["reference_python"] = dict(_pick(rp, ("value", "unit", "threads", "with_resets"), 60), measured="committed, build container",
                                     full_loop_value=(rp.get("full_loop") or {}).get("value"))
        o["learner"] = _pick(cb.get("learner") or {}, ("value", "unit", "batch", "threads", "kind", "samples_per_s"))
        out["cpu_baseline"] = o
    rows = []
    for r in full.get("other_configs") or []:
        rr, rl = r.get("roofline") or {}, r.get("roofline_learner") or {}
        row = {"name": _short(r.get("baseline_config", ""), 44), "value": _short(r.get("value")), "ms_per_pass": _short(r.get("ms_per_pass")),
               "kernel_ms": _short(rr.get("kernel_ms")), "frac": _short(rr.get("frac")), "frac_algorithmic": _short(rr.get("frac_algorithmic")),
               "frac_learner": _short(rl.get("frac"))}
        if "ranks_bit_identical" in r:
            row["ranks_bit_identical"], row["links_crossed"] = r.get("ranks_bit_identical"), r.get("links_crossed")
        rf_ = (r.get("resets") or {}).get("refresh")
        if rf_:
            row["fresh_plans_per_s"], row["resets_per_s"] = _short(rf_.get("rows_committed_per_s")), _short(r["resets"].get("consumed_per_s"))
        if r.get("retries"):
            row["retries"] = r["retries"]
        if "error" in r:
            row["error"] = _short(r["error"], 80)
        rows.append({k: v for k, v in row.items() if v is not None})
    if rows:
        out["other_configs"] = rows
    if side_files:
        out["side_files"] = side_files
    line = json.dumps(out)
    if len(line) >= HEADLINE_MAX_BYTES:                  # never let a driver-facing line outgrow the driver's window again
        for k in ("other_configs", "roofline_learner"):
            out.pop(k, None)
            if len(json.dumps(out)) < HEADLINE_MAX_BYTES:
                break
    assert len(json.dumps(out)) < HEADLINE_MAX_BYTES, len(json.dumps(out))
    return out


def write_side_files(full: dict) -> dict:
    """bench_full.json = the whole result (every prose field), bench_other_configs.json = the per-config blocks; written beside
    bench.py and, when it exists, under gpurun_out/ (which gpurun merges back).  Returns {name: relative path}; never fatal."""
    names = {}
    for name, obj in (("bench_full.json", full), ("bench_other_configs.json", full.get("other_configs"))):
        if obj is None:
            continue
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if not os.path.isdir(d):
                continue
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(obj, f, indent=1)
                names.setdefault(name.split(".")[0], os.path.relpath(os.path.join(d, name), ROOT))
            except OSError:
                pass
    return names


def learner_flops_per_sample(trainer: str, n_actions: int = 3) -> float:
    """Algorithmic FLOPs of one learn_off_policy() per sampled transition for the 100-64-A MLP (2 per multiply-add):
    forward of q_local(s) and q_target(s') (+ q_local(s') for the double-DQN target, + the value head for VAnet2),
    backward of q_local(s): dW1 (64 x 100), dW2 and dH (64 x n2 each).  DQN: 39.9 kFLOP; Dueling + DDQN: 54.1 kFLOP."""
    n2 = n_actions + (1 if trainer == "dueling" else 0)
    fwd = 2 * (100 * 64 + 64 * n2)
    n_fwd = 2 if trainer == "dqn" else 3
    bwd = 2 * (64 * 100) + 2 * (64 * n2) * 2
    return float(n_fwd * fwd + bwd)


# ---- the bytes a step launch moves, plane by plane (round 6: reconciled with the counters) -------------------------------------
# profiles/r06_counter_calibration.txt: on gfx950 WRITE_SIZE x 1024 = the bytes written, exactly, for every pattern the step kernel uses
# (f64 / i32 / 1-byte planes, 80-byte rows, 16-byte records), and FETCH_SIZE x 1024 = exactly HALF the bytes read (8-, 4-, 16-byte and
# row reads alike).  The counters of the committed profile therefore ARE the traffic; what was short was the model: SURVEY 8(d)'s 604 B
# count the planes update_PathPlan CHANGES (pos, V_vector, V, Step, sub_idx ...), the kernels load and store the agent's WHOLE state
# record -- 19 f64 + 6 i32 planes = 176 B each way (csrc/uavenv.hip: load_agent / store_agent; goal, the sub-goal window, score /
# total / path_len, cached heading, scenario id and epoch ride along).
STATE_PLANE_BYTES = 19 * 8 + 6 * 4          # per agent, read AND written every step


def step_bytes_model(row_bytes: int, n_agents: int, *, policy: bool, records: bool, world_bytes: int = 8704, apf_pairs: float = 0.0) -> dict:
    """Bytes per agent-step of one step launch as the kernels issue them.  Per-launch shared reads (the world blob, the copy-out
    table, with `policy` the staged layer-1 image + fc2) are L2 hits after the first workgroup of an XCD: counted once per XCD."""
    shared = 8 * (world_bytes + 13 * 64 * 8 + (27648 + 16 * 64 * 4 + 64 if policy else 0))
    rd = {"state planes (19 f64 + 6 i32)": STATE_PLANE_BYTES, "action / policy row of the current frame": row_bytes if policy else 4,
          "per-XCD shared (world blob, copy-out table%s)" % (", layer-1 image + fc2" if policy else ""): shared / float(n_agents)}
    wr = {"state planes (19 f64 + 6 i32)": STATE_PLANE_BYTES, "observation row of frame t+1": row_bytes, "reward f32": 4, "done + valid bytes": 2}
    if records:
        wr["transition record"] = 16
    if policy:
        wr["action (int32, written by the policy)"] = 4
    if apf_pairs:
        rd["sub-goal lists (APF)"] = wr["sub-goal lists (APF)"] = apf_pairs * 24
    r, w = sum(rd.values()), sum(wr.values())
    return {"read": rd, "written": wr, "read_bytes": r, "written_bytes": w, "total": r + w}


def physical_view(algo_bytes: int, moved_bytes: int, n_agents: int, kernel_ms: float, traffic, copy_gbs) -> dict:
    """The HBM view of a step launch beside the algorithmic one.  `frac` (the contract's figure) prices SURVEY 8(d)'s 604 B per
    agent-step -- what the reference's f32 layout would have to move; packed rows are a lossless 80-byte image of the 400-byte
    observation row, so the kernel MOVES far less.  frac_physical_stored = the bytes of the layout as stored (state planes + the
    row format in use + the policy's row read) / time / 8 TB/s; frac_physical_counters = FETCH_SIZE x 2 + WRITE_SIZE of the
    committed PMC pass / time / 8 TB/s (None without counters).  An algorithmic GB/s above the copy bandwidth measured in this
    same run is flagged: it is a statement about the packing, not about the memory system."""
    sec = kernel_ms * 1e-3
    algo_gbs = algo_bytes * n_agents / sec / 1e9
    moved_gbs = moved_bytes * n_agents / sec / 1e9
    out = {"frac_algorithmic": algo_gbs / HBM_PEAK_GBS,
           "moved_bytes_per_agent_step": moved_bytes, "achieved_physical_stored_GBs": moved_gbs,
           "frac_physical_stored": moved_gbs / HBM_PEAK_GBS,
           "achieved_physical_counters_GBs": None if not traffic else traffic / sec / 1e9,
           "frac_physical_counters": None if not traffic else traffic / sec / 1e9 / HBM_PEAK_GBS,
           "packing_gain": algo_bytes / float(moved_bytes)}
    if copy_gbs:
        out["frac_physical_stored_of_measured_copy"] = moved_gbs / copy_gbs
        out["algorithmic_exceeds_measured_copy"] = bool(algo_gbs > copy_gbs)
    return out


def quote_physical_first(rf: dict, packed: bool) -> dict:
    """VERDICT r4 item 8.  A packed ring stores an 80-byte lossless image of the 400-byte observation row, so pricing the launch at
    SURVEY 8(d)'s 604 algorithmic bytes credits it with bytes it never moves.  Whenever the ring is packed, `achieved` / `frac`
    are therefore the PHYSICAL figures -- (FETCH_SIZE x 2 + WRITE_SIZE) of the committed PMC pass / kernel time, or, without
    counters, the bytes of the layout as stored -- and the 604-B view stays beside them as `achieved_algorithmic_GBs` /
    `frac_algorithmic` (with `algorithmic_exceeds_measured_copy` in the same block).  f32 / f16 rows: algorithmic = what is stored."""
    rf["achieved_algorithmic_GBs"] = rf.get("achieved")
    rf.setdefault("frac_algorithmic", rf.get("frac"))
    if not packed:
        rf["frac_basis"] = "algorithmic bytes (SURVEY 8(d)) = the row format as stored"
    elif rf.get("frac_physical_counters") is not None:
        rf["achieved"], rf["frac"] = rf["achieved_physical_counters_GBs"], rf["frac_physical_counters"]
        rf["frac_basis"] = ("physical, PMC counters: (FETCH_SIZE x 2 + WRITE_SIZE) per launch of the committed rocprofv3 pass / kernel "
                            "time / 8 TB/s (packed ring; the 604-B algorithmic view is frac_algorithmic)")
    else:
        rf["achieved"], rf["frac"] = rf["achieved_physical_stored_GBs"], rf["frac_physical_stored"]
        rf["frac_basis"] = ("physical, bytes of the layout as stored / kernel time / 8 TB/s (packed ring, no committed counters for this "
                            "workload; the 604-B algorithmic view is frac_algorithmic)")
    if rf.get("measured_copy_GBs") and rf.get("achieved") is not None:
        rf["frac_of_measured_copy"] = rf["achieved"] / rf["measured_copy_GBs"]
    return rf


def measure_copy_gbs(dev) -> float:
    """Achievable HBM bandwidth on THIS device in THIS run: device-to-device copy of 1 GiB, read + write bytes."""
    src = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    dst = torch.empty_like(src)
    dst.copy_(src)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    c0.record()
    for _ in range(10):
        dst.copy_(src)
    c1.record()
    torch.cuda.synchronize(dev)
    return 10 * 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9


def device_identity(dev) -> str:
    """Something that tells two physical GPUs apart (ranks report it; `links_crossed` = more than one distinct value)."""
    p = torch.cuda.get_device_properties(dev)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(p, attr, None)
        if v is not None:
            return "%s:%s" % (attr, v)
    import socket
    return "%s:cuda%d" % (socket.gethostname(), dev.index or 0)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40, help="bench steps of PASSES_PER_STEP passes each")
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--passes-per-step", type=int, default=PASSES_PER_STEP)
    p.add_argument("--host-loop", default="c", choices=["c", "python"],
                   help="c = csrc/loop.hip enqueues the passes (default at 1 GPU); python = four ctypes calls per pass")
    p.add_argument("--envs", type=int, default=16384, help="envs per GPU (BASELINE configs[1]: 16384)")
    p.add_argument("--batch", type=int, default=16384, help="learner batch per GPU per update")
    p.add_argument("--replay", type=int, default=1 << 20, help="replay capacity in transitions per GPU")
    p.add_argument("--trainer", default="dqn", choices=["dqn", "ddqn", "dueling"])
    p.add_argument("--obs-dtype", default=None, choices=["packed", "f32", "f16"],
                   help="replay / observation storage: packed = 15 f32 scalars + 80 flag bits per row (80 B, lossless image "
                        "of the f32 row: include/uavenv.h UAVENV_OBS_PACKED); f32 / f16 = rows of 100 elements.  Default: "
                        "packed (f16 with --config 3: BASELINE configs[2] stores fp16, and the f16-MFMA learner reads f16 "
                        "rows faster than it expands packed ones -- 38.7 vs 45.6 us per 65536-sample launch)")
    p.add_argument("--mfma", default="f32", choices=["f32", "f16"],
                   help="operand type of the fused learner's matrix products: f32 (the reference's precision) or f16 with f32 "
                        "accumulation (BASELINE configs[2]; needs --obs-dtype f16 or packed)")
    p.add_argument("--eps", type=float, default=0.1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=10.0)
    p.add_argument("--env-only-iters", type=int, default=200)
    p.add_argument("--cell", type=float, default=0.0, help="broad-phase cell size in metres (0 = library default)")
    p.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                   help="BASELINE.json configs[] preset: 2 = 16384 envs DQN (the benchmark line); 3 = 65536 envs, "
                        "DuelingDQN + double-DQN target, f16 MFMA Q-net; 4 = 32768 envs x 4 UAVs, APF on, SAC continuous "
                        "(one trainer per UAV slot); 5 = 32768 envs per GPU (262144 over 8)")
    p.add_argument("--apf", action="store_true", help="diagnostic (env-only): APF on, buildings moving with seeded "
                   "velocities U(-1,1)^2 (BASELINE configs[3]'s env settings)")
    p.add_argument("--uav-per-env", type=int, default=1, help="diagnostic (env-only): UAVs per env")
    p.add_argument("--sync", default="grad", choices=["grad", "fedavg"],
                   help="N > 1: all-reduce the gradient bucket every update (default), or average the weights every "
                        "FL_Loop = 3 updates (the reference's federated mode as all-reduce(avg))")
    p.add_argument("--exchange", default="p2p", choices=["p2p", "coll", "rccl", "none"],
                   help="N > 1, --sync grad: p2p = one-shot sum over HIP-IPC-mapped peer memory on the stream (csrc/p2p.hip; "
                        "falls back to coll, then rccl, if it cannot be set up or raises its sticky error); coll = RCCL "
                        "all-reduce enqueued from C inside the loop (csrc/coll.hip); rccl = torch.distributed all_reduce per "
                        "update from Python; none = diagnostic: no exchange at all (the ranks drift apart)")
    p.add_argument("--no-exchange-leg", action="store_true", help="N > 1: skip the short in-run leg without the exchange")
    p.add_argument("--no-other-configs", action="store_true",
                   help="N = 1, default command: do not run BASELINE configs[2..4] + the env-only points as child processes")
    p.add_argument("--per", action="store_true",
                   help="prioritised replay (IsPriority_Replay = 1; BaseClass/replay_buffer.py:121-223) inside the C loop: per pass "
                        "new-frame priorities, rebuild, ReplayTree.sample, importance weights, the weighted update, batch_update")
    p.add_argument("--sample-lag", type=int, default=0, choices=[0, 1],
                   help="0 = the reference's strictly serial act -> step -> learn (the benchmark line).  1 = EXPERIMENT (a stated "
                        "deviation): update t samples the transitions stored before step t, so its gradient kernel runs on a "
                        "second stream beside step t (csrc/loop.hip)")
    p.add_argument("--replan-every", type=int, default=0,
                   help="rolling refresh of the reset bank (the reference plans a fresh path at every reset): every that many passes "
                        "the C loop commits the slice planned in the background and starts the next one (0 = the bank stays as planned)")
    p.add_argument("--replan-count", type=int, default=16384,
                   help="bank rows per refresh slice.  A slice lasts as long as its longest tree (~20-25 ms: a few reset scenarios "
                        "need thousands of RRT iterations), so the refresh rate is rows per slice / that time: large slices")
    p.add_argument("--bank-size", type=int, default=0,
                   help="reset scenarios planned at start-up (0 = max(envs, 4096); 4 x envs with --replan-every: a refresh skips rows an "
                        "agent is flying, and with as many rows as agents that is 63 %% of them)")
    p.add_argument("--inject-p2p-fault", type=int, default=-1,
                   help="test: rank R raises the peer exchange's sticky error before the timed region (exercises the fallback)")
    p.add_argument("--p2p-check-every", type=int, default=256, help="N > 1: on-device weight checksum compare every that many updates")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) | gloo (test: several ranks on one GPU)")
    p.add_argument("--same-device", action="store_true", help="test only: every rank uses cuda:0")
    p.add_argument("--no-reset-count", action="store_true",
                   help="skip the 1 024 extra policy + step launches that count episode ends (scripts/profile_round.sh: under rocprofv3 "
                        "they would enter the kernel's average -- each follows a torch reduction, not a learner launch)")
    p.add_argument("--full-line", action="store_true", help="print the whole result dict instead of the bounded headline line")
    p.add_argument("--no-obs", action="store_true", help="diagnostic (env-only): skip the observation")
    p.add_argument("--env-only", action="store_true", help="diagnostic: time only back-to-back k_step launches")
    p.add_argument("--bank", default="gpu", choices=["gpu", "packaged"],
                   help="reset scenarios: planned on the GPU at start-up (csrc/rrt.hip) or the packaged reference resets")
    p.add_argument("--learner", default="fused", choices=["fused", "torch"],
                   help="fused = hand-written HIP kernels (csrc/learner.hip); torch = PyTorch-ROCm ops")
    a = p.parse_args()
    a.explicit = {k for k in ("envs", "batch", "trainer", "mfma", "obs_dtype") if any(x.startswith("--" + k.replace("_", "-")) for x in sys.argv[1:])}
    if a.config == 3:
        a.envs, a.batch, a.trainer, a.mfma = 65536, 65536, "dueling", "f16"
        a.obs_dtype = a.obs_dtype or "f16"
    elif a.config == 4:
        a.envs, a.batch, a.obs_dtype, a.trainer = 32768, 32768, "packed", "sac"
    elif a.config == 5:
        a.envs, a.batch = 32768, 32768
    a.obs_dtype = a.obs_dtype or "packed"
    return a


def cpu_baseline(envs: int, seconds: float):
    """The oracle port (oracle/uav_oracle.c, -O3 build) on the host cores: the SAME env count as the GPU run, all steps
    inside C (orc_rollout_many: each OpenMP thread owns a block of agents and runs its steps without coming back to
    Python), auto-reset from the same packaged scenario bank.  Timed twice: one core, then every core."""
    from dqn_based_uav_3d_path_planer_amd.data import load_city26
    from oracle import pyoracle as po
    c = load_city26()
    world = po.OracleWorld(c["buildings"], c["len"], c["width"], c["h"], fast=True)
    params = dict(max_v=float(c["max_v"]), steering_angle=float(c["steering_angle"]), max_step=int(c["max_step"]),
                  apf_enabled=0)
    n = envs
    batch = po.OracleBatch(world, params, n)
    head = np.random.default_rng(0).uniform(0, 2 * np.pi, len(c["start_goal"]))
    batch.load_scenarios(c["start_goal"][:, :3], c["start_goal"][:, 3:], head, c["sub_goals"], c["n_sub"])
    cores = int(po.lib(fast=True).orc_max_threads())
    quota = None
    try:                                                             # the container's CPU quota, if it has one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception:
            pass
    if quota is not None:                                            # more threads than the quota only adds throttling
        cores = max(1, min(cores, int(quota + 0.5)))
    bank = (c["start_goal"], c["sub_goals"], c["n_sub"])

    def timed(threads, budget_s):
        batch.rollout(1, *bank, seed=1, nthreads=threads)            # warm-up: thread pool, page faults
        t0 = time.perf_counter()
        done, k, seed = 0, 2, 2
        while True:                                                  # chunks sized from the rate seen so far
            d, _ = batch.rollout(k, *bank, seed=seed, nthreads=threads)
            done += d
            seed += 1
            el = time.perf_counter() - t0
            if el >= budget_s:
                break
            k = max(2, min(2000, int(0.5 * (budget_s - el) * (done / el) / n) + 1))
        return done / el, done, el

    one, done1, dt1 = timed(1, 0.25 * seconds)
    allc, done, dt = timed(cores, 0.75 * seconds)
    out = {"value": allc, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "one_core_value": one, "cgroup_cpu_quota_cores": quota,
           "sample": f"{done} agent-steps ({n} envs x {done // n} steps, every step inside C, update_PathPlan + "
                     f"state_PathPlan, random steering, auto-reset from the packaged bank, no learner) in {dt:.1f} s on "
                     f"{cores} threads (host CPU quota of the container: {quota} cores); 1 thread: {done1} agent-steps in {dt1:.1f} s; C port of the reference's Python env "
                     f"path (oracle/uav_oracle.c, -O3, OpenMP static blocks)"}
    out["reference_python"] = reference_python_baseline()
    out["learner"] = cpu_learner_baseline()
    return out


def reference_python_baseline() -> dict:
    """The reference's OWN Python path, as measured by the committed recipe oracle/time_reference.py (oracle/ref_harness.RefSession
    executes a scratch copy of the reference) in the build container: /root/reference does not travel to the GPU box, so this
    line READS profiles/ref_python_baseline.json instead of quoting prose.  value = env path (update_PathPlan + state_PathPlan,
    one thread, GIL-bound: Agents/UAV.py:397-567); full_loop = run_eposide with the shipped SAC config (Envs/PathPlan_City.py:
    410-478); learner_batch64 = DQN_Trainer.learn_off_policy at the reference's own batch."""
    path = os.path.join(ROOT, "profiles", "ref_python_baseline.json")
    try:
        d = json.load(open(path))
    except Exception as e:
        return {"value": None, "unit": "env-steps/s", "source": "profiles/ref_python_baseline.json missing (%s): run "
                "oracle/time_reference.py in a container that has /root/reference" % type(e).__name__, "measured_where": None}
    return {"value": d["env_path"]["value"], "unit": "env-steps/s", "threads": 1,
            "source": "profiles/ref_python_baseline.json <- %s (sha %s)" % (d.get("script"), d.get("script_sha256_16")),
            "measured_where": "%s; %s, %s" % (d.get("measured_where"), (d.get("host_cpu") or {}).get("model"), d.get("date_utc")),
            "what": d["env_path"]["what"], "with_resets": d["env_path"].get("with_resets"),
            "full_loop": {k: d["full_loop"].get(k) for k in ("value", "unit", "updates_per_s", "what")},
            "learner_batch64": {k: d["learner_batch64"].get(k) for k in ("value", "unit", "batch", "threads", "what")},
            "not_run_here": "the reference is pure Python under /root/reference, which exists in the build container only"}


def cpu_learner_baseline(batch: int = 16384, seconds: float = 4.0):
    """The learner half of the metric on the host: DQN_Trainer.learn_off_policy's arithmetic (learner.DQNLearner =
    the same PyTorch ops as the reference, Trainer/DQN_Trainer.py:101-139) on CPU torch, batch as on the GPU."""
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    torch.manual_seed(0)
    L = DQNLearner({"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, "dqn", device="cpu")
    g = torch.Generator().manual_seed(0)
    b = dict(states=torch.rand((batch, 100), generator=g), next_states=torch.rand((batch, 100), generator=g),
             actions=torch.randint(0, 3, (batch,), generator=g), rewards=torch.rand(batch, generator=g),
             dones=(torch.rand(batch, generator=g) < 0.05).float())
    L.learn(b)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        L.learn(b)
        n += 1
    dt = time.perf_counter() - t0
    ref = reference_python_baseline().get("learner_batch64") or {}
    return {"value": n / dt, "unit": "learner updates/s", "batch": batch, "threads": torch.get_num_threads(),
            "kind": "port",
            "what": f"PyTorch CPU (learner.DQNLearner: the reference's torch ops), batch {batch} = the GPU run's batch -- NOT the "
                    "reference's learn_off_policy at its own batch 64; that figure, measured by oracle/time_reference.py, is "
                    "`reference_batch64`",
            "samples_per_s": n * batch / dt,
            "reference_batch64": ref,
            "sample": f"{n} updates of {batch} resident samples in {dt:.1f} s, PyTorch CPU"}


def csrc_sha() -> str:
    """Hash of every kernel source (csrc/*, include/uavenv.h): scripts/summarize_profile.py stamps profiles/summary.json
    with it, so that counters taken from a committed profile can be told apart from counters of the code that is running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dqn_based_uav_3d_path_planer_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "uavenv.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_profile(args, env_only: bool = False) -> dict:
    """The rocprofv3 figures of THIS command line as last committed under profiles/ (kernel-trace averages, PMC
    traffic, MFMA busy): scripts/summarize_profile.py writes profiles/summary.json keyed by workload.  `stale` says
    whether the kernel sources changed since that profile was taken."""
    path = os.path.join(ROOT, "profiles", "summary.json")
    key = "envs%d_batch%d_%s_%s" % (args.envs, args.batch, args.trainer, args.obs_dtype) + ("_mfma16" if args.mfma == "f16" else "")
    if env_only:
        key = "envonly%d_%s" % (args.envs * args.uav_per_env, args.obs_dtype) + ("_apf" if args.apf else "")
    try:
        d = json.load(open(path)).get(key, {})
        if d:
            d["source"] = "profiles/summary.json[%s] <- %s" % (key, d.get("files", "rocprofv3"))
            d["stale"] = d.get("csrc_sha") != csrc_sha()
        return d
    except Exception:
        return {}


def run_config4(args, dev, world_size=1, rank=0):
    """(N > 1: every rank steps its own env shard -- scenario bank and ring seeded by rank -- and the SAC loop sums each
    phase's column sums over the ranks on the stream, SACHotLoop(exchange="auto"); the ranks' weights are compared after the run.)
    BASELINE configs[3]: 4 UAVs per env x 32 768 envs, APF avoidance on (every building moving), SAC_Trainer with
    continuous actions, one trainer per UAV slot (Envs/PathPlan_City.py:63-68).  One pass = SAC act for the four slots
    (PyTorch-ROCm) -> fused env step with APF into the packed replay ring -> for every slot: sample, one SAC update
    (Trainer/SAC_Trainer.py:325-379, PyTorch-ROCm ops; rows of agents that were waiting for their team-mates carry
    weight 0 in the critic losses)."""
    import ctypes as C
    from dqn_based_uav_3d_path_planer_amd import _lib
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner
    U, envs = 4, args.envs
    multi = world_size > 1
    env = make_city26_env(envs, bank="gpu", bank_size=max(envs, 4096), bank_seed=42 + rank, device=dev, obs_dtype="packed",
                          uav_per_env=U, apf_enabled=1)
    v = np.random.default_rng(42).uniform(-1.0, 1.0, (len(env.buildings), 3))
    v[:, 2] = 0.0
    env.set_buildings(env.buildings, velocities=v)
    ring = DeviceReplayRing(env, args.replay, discrete=False)
    ring.reset(seed=1000 + rank)
    a1_plane = torch.zeros((ring.frames, env.N), dtype=torch.float32, device=dev)      # second action component (:444-448)
    ring.attach_action1(a1_plane)           # recorded with every transition: the fused update gathers one record per sample
    sac_param = {"actor": {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64",
                           "output": "2", "lr": "0.0001"},
                 "critic": {"NetWork": "QValueNetContinuous_SAC", "w": "100", "hiden_dim": "64", "action_dim": "2", "lr": "0.001"},
                 "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"}}
    torch.manual_seed(42)
    torch.distributions.Distribution.set_default_validate_args(False)      # Normal(mu, std) otherwise syncs to check std > 0
    graphed = os.environ.get("BENCH_SAC_GRAPH", "1") != "0"
    fused = args.learner == "fused"
    if fused:
        from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
        graphed = False                       # four launches per update: nothing to capture
        learners = [FusedSACLearner(sac_param, device=dev) for _ in range(U)]
    else:
        learners = [SACLearner(sac_param, device=dev, capturable=graphed) for _ in range(U)]
    B = args.batch
    lib, counter = env.lib, [0]
    draws_all = torch.empty((U * B, 2), dtype=torch.int32, device=dev)      # one draw launch per pass: U x B distinct (frame, env)
    draws = [draws_all[j * B:(j + 1) * B] for j in range(U)]
    flat = ring.obs.view(-1, ring.obs.shape[-1])
    znoise = [None]
    fbatch = [L.make_batch(flat, ring.action.view(-1), a1_plane.view(-1), ring.reward.view(-1), ring.done.view(-1),
                           valid=ring.valid.view(-1), draws=draws[j], n_agents=env.N, uav_per_env=U, slot=j, frames=ring.frames,
                           meta=None if ring.meta is None else ring.meta.view(-1, 4))
              for j, L in enumerate(learners)] if fused else None

    def update_slot(j):
        """gather the drawn transitions of UAV slot j (packed rows -> f32) and take one SAC update"""
        L = learners[j]
        if fused:                              # csrc/sac.hip reads the drawn rows in place: four launches
            L.learn(fbatch[j], noise=(znoise[0][j, 0], znoise[0][j, 1]))
            return
        f, e = draws[j][:, 0].long(), draws[j][:, 1].long()
        slot = f * env.N + e * U + j
        nxt = ((f + 1) % ring.frames) * env.N + e * U + j
        batch = dict(states=env.unpack(flat[slot]), next_states=env.unpack(flat[nxt]),
                     actions=torch.stack([ring.action.view(-1)[slot], a1_plane.view(-1)[slot]], 1),
                     rewards=ring.reward.view(-1)[slot], dones=ring.done.view(-1)[slot].float())
        L.learn(batch, valid=ring.valid.view(-1)[slot].float())

    one_draw = fused and (ring.frames - 1) * envs >= U * B

    def draw_slot(j):
        if one_draw:
            if j == 0:
                _lib.check(lib.uavenv_replay_draw(ring.frames, envs, ring.head, ring.filled, U * B, 7, counter[0],
                                                  draws_all.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "uavenv_replay_draw")
            return
        _lib.check(lib.uavenv_replay_draw(ring.frames, envs, ring.head, ring.filled, B, 7 + j, counter[0], draws[j].data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "uavenv_replay_draw")

    def act_and_step():
        t = ring.head
        if fused:                              # one launch per slot: actor forward on the packed rows of the current frame
            # every N(0,1) draw of the pass (the U get_action's and the 2 U rsample()'s of the updates) in ONE launch
            z = torch.randn(U * (2 * envs + 4 * B), dtype=torch.float32, device=dev)
            znoise[0] = z[U * 2 * envs:].view(U, 2, B, 2)
            za = z[:U * 2 * envs].view(U, envs, 2)
            for j, L in enumerate(learners):
                L.act_rows(flat, t * env.N + j, U, envs, ring.action.view(-1), a1_plane.view(-1), eps=za[j])
            ring.step_env(auto_reset=True)
            return
        obs = env.unpack(ring.current_obs()).view(envs, U, 100)
        act = ring.current_action().view(envs, U)
        a1 = a1_plane[t].view(envs, U)
        for j, L in enumerate(learners):
            a = L.act(obs[:, j])
            act[:, j] = a[:, 0]
            a1[:, j] = a[:, 1]
        ring.step_env(auto_reset=True)

    graphs = None
    if graphed:
        # One HIP graph per slot: gather + unpack + the ~150 kernels of SAC_Trainer.update, replayed with one launch.
        # (Eager, the pass is host-bound: 20 ms of Python / launch overhead around < 2 ms of GPU work.)
        try:
            for _ in range(3):
                act_and_step()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    for j in range(U):
                        draw_slot(j)
                        update_slot(j)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graphs = []
            for j in range(U):
                draw_slot(j)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    update_slot(j)
                graphs.append(g_)
            torch.cuda.synchronize(dev)
        except Exception as ex:        # capture unsupported on this stack: fall back to eager updates
            print("SAC graph capture failed, running eager:", repr(ex)[:200], file=sys.stderr)
            graphs = None

    def one_pass():
        act_and_step()
        for j in range(U):
            draw_slot(j)
            if graphs is not None:
                graphs[j].replay()
            else:
                update_slot(j)
        counter[0] += 1

    pps = max(1, args.passes_per_step // 8)           # a pass is ~20x longer than config 2's: 16 passes of ~0.7 ms per step
    hot = None
    if fused and args.host_loop == "c":
        # the product's loop: csrc/loop.hip uavenv_sac_loop_run -- per pass one launch of N(0,1) draws, get_action of the four
        # slots in one launch, the env step, one replay draw, and each phase of the fused update for the four slots in one
        # launch (8 launches per pass; --host-loop python issues the same work slot by slot from here, ~25 launches)
        from dqn_based_uav_3d_path_planer_amd.loop import SACHotLoop
        hot = SACHotLoop(ring, learners, B, seed=7 + rank, act1_plane=a1_plane, auto_reset=True, skip_done=True,
                         exchange=("auto" if args.exchange in ("p2p", "coll") else None) if multi else None)
        if multi and hot.exchange is None and args.exchange != "none":
            hot.close()               # neither on-stream exchange came up: the Python loop, whose learners use torch.distributed
            hot = None
    exchange_used = None
    if multi:
        exchange_used = hot.exchange if hot is not None else ("none" if args.exchange == "none" else "torch.distributed per phase")

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run_passes(n_passes):
        if hot is not None:
            hot.run(n_passes)
        else:
            for _ in range(n_passes):
                one_pass()

    fence()      # (the ranks enter their first pass together: one still busy with its set-up would let the other's first pull time out)
    run_passes(max(args.warmup, 1) * pps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_passes(pps)
    fence()
    dt = time.perf_counter() - t0
    ident = None
    if multi:
        got = [None] * world_size
        sums = [float(torch.cat([L._blocks.reshape(-1), L._cblocks.reshape(-1)]).double().sum()) for L in learners] if fused else []
        dist.all_gather_object(got, (dt, sums, device_identity(dev)))
        dt = max(g[0] for g in got)
        ident = all(g[1] == got[0][1] for g in got)
        rank_devs = [g[2] for g in got]
    n_pass = args.steps * pps
    it = args.env_only_iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        ring.step_env(auto_reset=True)
    e1.record()
    torch.cuda.synchronize(dev)
    k_ms = e0.elapsed_time(e1) / it
    prof = committed_profile(args)
    k_prof = prof.get("k_step_ms")
    if k_prof is not None and prof.get("k_apf_adjust_ms") is not None:     # the env step with APF is the launch pair
        k_prof += prof["k_apf_adjust_ms"]
    tr4 = prof.get("k_step_traffic_bytes_per_launch")
    if tr4 is not None and prof.get("k_apf_adjust_traffic_bytes_per_launch") is not None:
        tr4 += prof["k_apf_adjust_traffic_bytes_per_launch"]
    k_use = max(k_ms, k_prof or 0.0)
    algo = ALGO_BYTES_PER_AGENT_STEP + 2 * 20 * 24      # APF on adds 2 * n_sub * 24 B (~20 sub-goals)
    model4 = step_bytes_model(ring.obs.shape[-1] * ring.obs.element_size(), env.N, policy=False, records=ring.meta is not None, apf_pairs=2 * 20)
    moved = int(round(model4["total"]))          # whole state records + packed rows + ~20 sub-goals read and written (k_apf_adjust)
    copy_gbs = measure_copy_gbs(dev) if rank == 0 else None
    out = {"metric": "env-steps/sec + learner updates/sec, PathPlan_City SAC", "value": n_pass * env.N * world_size / dt,
           "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "passes_per_step": pps, "ms_per_pass": dt / n_pass * 1e3, "timed_region_ms": dt * 1e3,
           "learner_updates_per_s": n_pass * U / dt, "learner_samples_per_s": n_pass * U * B / dt,
           "config": {"workload": "PathPlan_City 500x500x100, 26 moving buildings, APF on, %d UAVs/env x %d envs, SAC continuous, "
                                  "one trainer per UAV slot, device replay %d transitions (BASELINE.json configs[3])"
                                  % (U, envs, ring.capacity),
                      "step_definition": "1 bench step = %d passes of (SAC act x%d -> env step with APF + replay write -> "
                                         "%d x (sample + SAC update))" % (pps, U, U),
                      "envs_per_gpu": envs, "uav_per_env": U, "learn_batch_per_slot": B, "obs_dtype": "packed",
                      "host_loop": "csrc/loop.hip uavenv_sac_loop_run (C, 8 launches per pass: N(0,1) draws, get_action x4 slots, "
                                   "k_apf_adjust + k_step, replay draw, 4 update phases x4 slots each)" if hot is not None
                      else "python (one launch per slot and phase)",
                      "learner": ("fused HIP SAC update (csrc/sac.hip: critic_grad, critic_adam, actor_grad, actor_adam; f32 MFMA)"
                                  if fused else "SAC on PyTorch-ROCm ops (f32)" +
                                  (", each slot's sample + update replayed as one HIP graph" if graphs is not None else ", eager")),
                      "env": "fused HIP k_step with APF"},
           "roofline": {"bound": "hbm", "kernel": "k_apf_adjust + k_step<APF> (Adjust_subgoal for the launch, then update_PathPlan + cal_force + state_PathPlan + replay write)",
                        "achieved": algo * env.N / (k_use * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo * env.N / (k_use * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr4,
                        "traffic_stale": prof.get("stale"), "measured_copy_GBs": copy_gbs,
                        **physical_view(algo, moved, env.N, k_use, tr4, copy_gbs),
                        "algorithmic_bytes_per_agent_step": algo, "agents_per_launch": env.N, "kernel_ms": k_use,
                        "kernel_ms_back_to_back": k_ms, "kernel_ms_rocprofv3_committed": k_prof}}
    quote_physical_first(out["roofline"], True)
    out["rendezvous_retries"] = int(os.environ.get("UAVENV_RDZV_RETRIES", "0"))
    if multi:
        out["exchange"] = exchange_used
        from dqn_based_uav_3d_path_planer_amd import exchange as _ex
        out["exchange_selftest_ms"] = _ex.selftest_ms[0]
        out["ranks_bit_identical"] = ident
        out["links_crossed"], out["rank_devices"] = len(set(rank_devs)) > 1, rank_devs     # False: every rank on one GPU, NOT row-e evidence
        out["config"]["parallelism"] = "env-shard x%d + per-phase gradient sum of all slots: %s" % (world_size, exchange_used)
    if fused:
        L0 = learners[0]
        out["config"]["slot0_after_run"] = {"updates": L0.epoch, "actor_loss": float(L0.loss), "critic_losses": [float(x) for x in L0.critic_losses],
                                            "log_alpha": float(L0.log_alpha),
                                            "finite": bool(all(torch.isfinite(b).all() for L in learners for b in (L._blocks, L._cblocks)))}
        # the two phase kernels of the fused SAC update, back to back between one event pair each (no Adam in between:
        # the kernels' work does not depend on the weights' values)
        L, b0 = learners[0], fbatch[0]
        draw_slot(0)
        z = torch.randn((2, B, 2), dtype=torch.float32, device=dev)
        L.epoch += 1
        ms = []
        for fn, e in ((L.critic_grad, z[0]), (L.actor_grad, z[1])):
            fn(b0, e)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(b0, e)
            e1.record()
            torch.cuda.synchronize(dev)
            ms.append(e0.elapsed_time(e1) / 20)
        flops = (169.0e3 + 89.0e3) * B          # as issued on the MFMA (K padded to 104, 16-wide head tiles): DESIGN 3.3
        tf = flops / (sum(ms) * 1e-3) / 1e12
        out["roofline_learner"] = {"bound": "mfma", "kernel": "k_sac_critic_grad + k_sac_actor_grad (one SAC update of %d samples)" % B,
                                   "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                                   "flops_per_sample": 258.0e3, "samples_per_launch": B,
                                   "kernel_ms": sum(ms), "critic_grad_ms_back_to_back": ms[0], "actor_grad_ms_back_to_back": ms[1],
                                   "algorithmic_bytes_per_sample": 2 * 80 + 30, "traffic": None}
    if multi:
        torch.cuda.synchronize(dev)
        dist.barrier()                       # no rank unmaps its peers while one of them may still be running a pull
    if hot is not None:
        hot.close()                          # (explicitly: a destructor at interpreter exit would run after the HIP runtime's own)
    env.close()
    return out


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no rank environment: start the N ranks here -- this file re-executed under
    torch.distributed.run, one process per GPU of this node, rendezvous on 127.0.0.1 -- and pass rank 0's line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not args.same_device and n_dev < args.gpus:
        raise SystemExit("--gpus %d asked for, %d visible (test several ranks on one GPU with --same-device --dist-backend gloo)"
                         % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: HIP IPC handles / RCCL across processes need it here
    env.setdefault("OMP_NUM_THREADS", "1")
    # A rendezvous port can be taken between the probe above and torch.distributed.run's bind (another test process on the box):
    # THAT failure -- and nothing else -- is answered by another port, and the line says so (`rendezvous_retries`); a failure of
    # the run itself is never retried here.
    rc = 1
    for attempt in range(3):
        env["UAVENV_RDZV_RETRIES"] = str(attempt)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        res = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(res.stderr or "")
        rc = res.returncode
        clash = rc != 0 and any(m in (res.stderr or "") for m in ("EADDRINUSE", "Address already in use", "address already in use"))
        if not clash:
            break
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return rc


def run_child(extra, timeout=600):
    """One more configuration as a child process (its own HIP context: a fault there cannot take the headline line down).
    Returns the child's JSON line as a dict, or {'error': ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-configs", "--full-line"] + extra
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
        if res.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (res.returncode, (res.stderr or res.stdout)[-300:])}
        return json.loads(lines[-1])
    except Exception as ex:                                   # noqa: BLE001 -- reported in the line, never fatal
        return {"error": repr(ex)[:300]}


def other_configs(args):
    """BASELINE.json configs[2], [3], [4] (at one GPU's share) and the env-only 65 536 / 262 144-agent points, each
    >= 0.5 s of timed region (the two-rank SAC row: >= 0.3 s), summarised for the headline line."""
    runs = [("configs[2]", ["--config", "3", "--steps", "10", "--warmup", "3"]),
            ("configs[3]", ["--config", "4", "--steps", "8", "--warmup", "2"]),
            ("configs[4] (one GPU's 32768-env share of the 8-GPU run)", ["--config", "5", "--steps", "12", "--warmup", "3"]),
            ("configs[1] with prioritised replay (IsPriority_Replay = 1) on the fused path", ["--per", "--steps", "8", "--warmup", "2"]),
            ("configs[1] with the reset bank turning over in the background (--replan-every 256 --replan-count 16384: fresh RRT plans "
             "committed per second against resets consumed per second)", ["--replan-every", "256", "--replan-count", "16384", "--steps", "12", "--warmup", "3"]),
            ("EXPERIMENT on configs[1] (not the benchmark's semantics): sample_lag = 1 -- update t samples transitions <= t - 1, "
             "its gradient kernel on a second stream beside step t", ["--sample-lag", "1", "--steps", "12", "--warmup", "2"]),
            ("row e on ONE GPU (NOT a multi-GPU measurement): two ranks sharing this device, 16384 envs each, gradient bucket "
             "summed over HIP-IPC-mapped memory on the stream (csrc/p2p.hip), ranks started by bench.py itself",
             ["--gpus", "2", "--same-device", "--dist-backend", "gloo", "--steps", "8", "--warmup", "2"]),
            ("row e for the SAC loop on ONE GPU (NOT a multi-GPU measurement): two ranks sharing this device, 8192 envs x 4 UAVs each, "
             "every phase's gradient rows of the four slots summed over HIP-IPC-mapped memory inside uavenv_sac_loop_run",
             ["--config", "4", "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--envs", "8192", "--batch", "8192",
              "--steps", "12", "--warmup", "2"]),
            ("env-only 65536 agents/launch", ["--env-only", "--envs", "65536", "--steps", "40"]),
            ("env-only 262144 agents/launch", ["--env-only", "--envs", "262144", "--steps", "20"])]
    if torch.cuda.device_count() >= 2:
        # a box with two visible GPUs: the N > 1 path with one rank per DEVICE, as part of the default command (what
        # tests/test_exchange_gpu.py::test_two_ranks_on_two_devices asserts, without pytest): the peer exchange over mapped peer HBM,
        # then the RCCL communicator driven from C -- links_crossed must read true in both rows
        runs += [("row e on TWO devices: peer exchange over HIP-IPC-mapped HBM (csrc/p2p.hip)",
                  ["--gpus", "2", "--steps", "8", "--warmup", "2", "--p2p-check-every", "16"]),
                 ("row e on TWO devices: RCCL all-reduce enqueued from C (csrc/coll.hip)",
                  ["--gpus", "2", "--exchange", "coll", "--steps", "8", "--warmup", "2"]),
                 ("row e on TWO devices, SAC loop (configs[3] at 8192 envs x 4 UAVs per rank)",
                  ["--config", "4", "--gpus", "2", "--envs", "8192", "--batch", "8192", "--steps", "12", "--warmup", "2"])]
    out = []
    for name, extra in runs:
        t0 = time.perf_counter()
        d = run_child(extra)
        retried = None
        if "error" in d and "--same-device" in extra:
            # ranks time-slicing ONE GPU: tried once more, and COUNTED -- `retries` is in every row (0 unless this happened) and
            # the first failure is quoted, so a 1-in-N failure of the exchange cannot hide behind a green row
            retried = d["error"][-160:]
            d = run_child(extra)
        row = {"baseline_config": name, "command": "bench.py " + " ".join(extra), "wall_s": round(time.perf_counter() - t0, 1),
               "retries": 0 if retried is None else 1}
        if retried is not None:
            row["first_attempt_failed"] = retried
        if "error" in d:
            row["error"] = d["error"]
        elif d.get("mode") == "env-only":
            row.update({"workload": "k_step alone, %d agents per launch, %s rows, random actions, auto-reset; no learner, so no transition "
                                    "records are requested (with them: + 16 B per agent-step, ~3 %% at 65 536 agents)" % (d["envs"], d["obs_dtype"]),
                        "value": d["env_steps_per_s"], "unit": "env-steps/s", "timed_region_ms": d.get("timed_region_ms"),
                        "roofline": d["roofline"]})
        else:
            r, rl = d.get("roofline", {}), d.get("roofline_learner", {})
            row.update({"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                        "ms_per_pass": d.get("ms_per_pass"), "timed_region_ms": d.get("timed_region_ms"),
                        "learner_updates_per_s": d.get("learner_updates_per_s"),
                        **({"resets": d["config"]["resets"]} if (d["config"].get("resets") or {}).get("refresh") else {}),
                        **{k: d[k] for k in ("n_gpus", "ranks_bit_identical", "exchange", "exchange_fallbacks", "p2p_timeouts",
                                             "p2p_checksum_mismatches", "p2p_checksums_compared", "ms_per_pass_no_exchange",
                                             "exchange_selftest_ms") if k in d},
                        **{k: d[k] for k in ("links_crossed", "rank_devices") if k in d},
                        "rendezvous_retries": d.get("rendezvous_retries", 0),
                        "roofline": {k: r.get(k) for k in ("kernel", "kernel_ms", "achieved", "unit", "frac", "frac_basis", "agents_per_launch",
                                                           "frac_algorithmic", "achieved_algorithmic_GBs",
                                                           "frac_physical_stored", "frac_physical_counters", "moved_bytes_per_agent_step",
                                                           "traffic", "traffic_stale", "algorithmic_exceeds_measured_copy")},
                        "roofline_learner": {k: rl.get(k) for k in ("kernel", "kernel_ms", "achieved", "unit", "frac", "peak")}})
        out.append(row)
    return out


def run_dqn(args, world_size, rank, dev):
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner, FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing, select_actions

    obs_dtype = "packed" if args.obs_dtype == "packed" else (torch.float16 if args.obs_dtype == "f16" else torch.float32)
    t_plan = time.perf_counter()
    extra = {}
    if args.apf or args.uav_per_env > 1:
        if not args.env_only:
            raise SystemExit("--apf / --uav-per-env are env-only diagnostics")
        extra = dict(apf_enabled=1 if args.apf else 0, uav_per_env=args.uav_per_env)
    bank_size = args.bank_size or (4 * args.envs if args.replan_every > 0 else max(args.envs, 4096))
    env = make_city26_env(args.envs, bank=args.bank, bank_size=bank_size, bank_seed=42 + rank, device=dev,
                          obs_dtype=obs_dtype, cell_size=args.cell, **extra)
    if args.apf:
        v = np.random.default_rng(42).uniform(-1.0, 1.0, (len(env.buildings), 3))
        v[:, 2] = 0.0
        env.set_buildings(env.buildings, velocities=v)
    torch.cuda.synchronize(dev)
    t_plan = time.perf_counter() - t_plan
    # (env-only rows: a rollout nothing learns from does not ask for the learner's transition records -- the row says so)
    ring = DeviceReplayRing(env, args.replay, discrete=True, records=not args.env_only)
    ring.reset(seed=1000 + rank)
    if args.no_obs:
        from dqn_based_uav_3d_path_planer_amd import _lib as _l
        ring.extra_flags = _l.STEP_NO_OBS
    if args.env_only:      # diagnostic mode (not the benchmark contract): k_step alone, random actions, steady state
        gen = torch.Generator(device=dev).manual_seed(0)
        for _ in range(max(args.warmup * 16, 260)):   # run past the first resets so episodes are desynchronised
            ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device=dev, dtype=torch.int32))
            ring.step_env(auto_reset=True)
        for f in range(ring.frames):
            ring.action[f].copy_(torch.randint(0, 3, (env.N,), generator=gen, device=dev, dtype=torch.int32))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        iters = args.steps * args.passes_per_step
        for _ in range(iters):
            ring.step_env(auto_reset=True)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / iters
        algo = 404 if args.obs_dtype == "f16" else ALGO_BYTES_PER_AGENT_STEP
        gbs = algo * env.N / (ms * 1e-3) / 1e9
        model = step_bytes_model(ring.obs.shape[-1] * ring.obs.element_size(), env.N, policy=False, records=False)
        stored = int(round(model["total"]))
        prof = committed_profile(args, env_only=True)
        copy_gbs = measure_copy_gbs(dev)
        k_prof = prof.get("k_step_ms")
        out = {"mode": "env-only", "envs": env.N, "k_step_ms_back_to_back": ms, "timed_region_ms": ms * iters,
               "env_steps_per_s": env.N / (ms * 1e-3), "achieved_GBs": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS,
               "obs_dtype": args.obs_dtype, "replay_frames": ring.frames, "measured_copy_GBs": copy_gbs, "records": False,
               "bytes_model": model,
               "k_step_ms_rocprofv3_committed": k_prof, "traffic": prof.get("k_step_traffic_bytes_per_launch"),
               "traffic_stale": prof.get("stale") if prof.get("k_step_traffic_bytes_per_launch") else None,
               "traffic_source": prof.get("source") if prof.get("k_step_traffic_bytes_per_launch") else None,
               **physical_view(algo, stored, env.N, max(ms, k_prof or 0.0), prof.get("k_step_traffic_bytes_per_launch"), copy_gbs)}
        out["roofline"] = quote_physical_first(dict(out, achieved=gbs, frac=gbs / HBM_PEAK_GBS, kernel="k_step", unit="GB/s",
                                                    kernel_ms=ms), args.obs_dtype == "packed")
        out["roofline"] = {k: out["roofline"].get(k) for k in (
            "kernel", "kernel_ms", "achieved", "unit", "frac", "frac_basis", "achieved_algorithmic_GBs", "frac_algorithmic",
            "frac_physical_stored", "frac_physical_counters", "moved_bytes_per_agent_step", "achieved_physical_stored_GBs",
            "achieved_physical_counters_GBs", "measured_copy_GBs", "algorithmic_exceeds_measured_copy", "traffic", "traffic_stale",
            "k_step_ms_rocprofv3_committed")}
        env.close()
        return out
    net = "VAnet2" if args.trainer == "dueling" else "Qnet2"
    torch.manual_seed(42)               # same initial weights on every rank
    net_param = {"NetWork": net, "w": "100", "hiden_dim": "64", "output": "3"}
    fused = args.learner == "fused"
    if fused:
        learner = FusedDQNLearner(net_param, args.trainer, device=dev, mfma=args.mfma)
    else:
        learner = DQNLearner(net_param, args.trainer, device=dev,
                             amp_dtype=torch.float16 if args.obs_dtype == "f16" else None)
    learner.sync = args.sync
    seed = 7 + rank
    pps = args.passes_per_step
    multi = world_size > 1

    # ---- N > 1: which exchange carries the gradient bucket.  p2p (csrc/p2p.hip, verified against an RCCL all-reduce at
    # start-up) and coll (RCCL enqueued from C, csrc/coll.hip) keep the loop in C; rccl = torch.distributed per update.
    exchange = {"asked": args.exchange if multi and args.sync == "grad" else None, "used": None, "fallbacks": []}

    def choose_exchange(allow_p2p=True):
        if not (fused and multi and args.sync == "grad"):
            return "fedavg (weights, every %d updates)" % learner.fl_loop if multi and args.sync == "fedavg" else None
        if args.exchange == "none":
            learner.sync = "fedavg"
            learner.fl_loop = 1 << 30
            return "none"
        if args.exchange == "p2p" and allow_p2p:
            if learner.enable_p2p(check_every=args.p2p_check_every):
                return "p2p"
            exchange["fallbacks"].append("p2p set-up or self-test failed")
        if args.exchange in ("p2p", "coll"):
            if args.dist_backend == "nccl" and learner.enable_coll():
                return "coll"
            exchange["fallbacks"].append("coll unavailable (needs the nccl backend and one GPU per rank)")
        return "rccl"

    exchange["used"] = choose_exchange()
    counter = [0]
    py_events = []
    ev_every = int(os.environ.get("BENCH_EVENT_EVERY", "64"))
    state = {"hot": None, "use_c": False}

    def build_loop():
        if state["hot"] is not None:
            state["hot"].close()
            state["hot"] = None
        state["use_c"] = fused and args.host_loop == "c" and (not multi or exchange["used"] in ("p2p", "coll", "none"))
        if args.per and not state["use_c"]:
            raise SystemExit("--per runs inside the C loop (fused learner, --host-loop c)")
        if state["use_c"]:
            from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
            if args.per and state.get("per") is None:
                from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
                state["per"] = DevicePER(ring.frames * env.N, device=dev, tree_order=False)
            state["hot"] = HotLoop(ring, learner, args.batch, seed, eps=args.eps, counter=counter[0], time_every=ev_every,
                                   sample_lag=args.sample_lag, per=state.get("per"),
                                   replan_every=args.replan_every, replan_count=args.replan_count)

    build_loop()

    def one_pass(record=False):
        record = record and counter[0] % ev_every == 0
        if fused:
            learner.act(ring.current_obs(), args.eps, seed, counter[0], index_out=ring.current_action())
        else:
            q = learner.q_values(env.unpack(ring.current_obs()))
            select_actions(env, q, args.eps, seed, counter[0], index_out=ring.current_action())
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ring.step_env(auto_reset=True)
            e1.record()
            py_events.append((e0, e1))
        else:
            ring.step_env(auto_reset=True)
        if fused:
            learner.learn_from_ring(ring, args.batch, seed, counter[0])
        else:
            learner.learn(ring.sample(args.batch, seed, counter[0]))
        counter[0] += 1

    def one_step(record=False):
        """One bench step = pps passes of act -> env step (+ replay write) -> learner update."""
        hot = state["hot"]
        if hot is not None:
            hot.run(pps)
            counter[0] = hot.counter
        else:
            for _ in range(pps):
                one_pass(record)

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x: float) -> float:
        """(through all_gather_object: a gloo all_reduce of a CUDA tensor leaves this process's later GPU work 2-3x slower --
        measured with two ranks on one GPU, round 3 -- and the backend-neutral object gather costs nothing here)"""
        if not multi:
            return x
        got = [None] * world_size
        dist.all_gather_object(got, float(x))
        return max(got)

    def any_rank(flag: bool) -> bool:
        if not multi:
            return flag
        got = [None] * world_size
        dist.all_gather_object(got, bool(flag))
        return any(got)

    def timed(n_steps, record=True):
        """n_steps bench steps between two fences; (seconds, host enqueue seconds, exchange failed on some rank)."""
        from dqn_based_uav_3d_path_planer_amd.loop import P2PExchangeError
        failed = False
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            try:
                one_step(record=record)
            except P2PExchangeError:
                failed = True
                break
        t_enq = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if exchange["used"] == "p2p" and not failed:
            failed = learner.p2p_status()["code"] != 0
        return dt, t_enq, any_rank(failed)

    def recover_from_p2p():
        """The peer exchange raised its sticky error on some rank: every rank drops it, joins the collective, takes rank
        0's weights and Adam moments, and the loop is rebuilt."""
        exchange["fallbacks"].append("p2p raised its sticky error during the run: %s" % learner.p2p_status())
        torch.cuda.synchronize(dev)
        learner.disable_p2p()
        exchange["used"] = choose_exchange(allow_p2p=False)
        learner.broadcast_weights(0)
        ep = torch.tensor([learner.epoch], device=dev, dtype=torch.int64)
        dist.broadcast(ep, src=0)
        learner.epoch = int(ep.item())
        build_loop()

    # untimed: experience in the ring + warm-up of every kernel / allocator path
    _, _, bad = timed(max(args.warmup, 1), record=False)
    if bad:
        recover_from_p2p()
    if args.inject_p2p_fault == rank and exchange["used"] == "p2p":
        from dqn_based_uav_3d_path_planer_amd import _lib as _l
        _l.check(learner.lib.uavenv_p2p_inject_fault(learner._p2p, _l.P2P_ERR_TIMEOUT), "uavenv_p2p_inject_fault")
    if state["hot"] is not None:
        state["hot"].step_times_ms()          # drop the warm-up's event pairs
    dt, t_enq, bad = timed(args.steps)
    if bad:                                   # the timed region ran (partly) on a failed exchange: recover, time it again
        recover_from_p2p()
        timed(1, record=False)
        if state["hot"] is not None:
            state["hot"].step_times_ms()
        dt, t_enq, bad = timed(args.steps)
    dt = max_over_ranks(dt)
    n_pass = args.steps * pps
    hot, use_c = state["hot"], state["use_c"]

    # ---- N > 1: are the ranks still bit-identical?  (they must be under any per-update exchange of the gradient sums)
    multi_report = {}
    if multi and fused:
        st = learner.p2p_status() if exchange["used"] == "p2p" else {"code": 0, "timeouts": 0, "mismatches": 0, "checks": 0}
        gathered = [None] * world_size
        dist.all_gather_object(gathered, (learner.weights_checksum(), st["code"], st["timeouts"], st["mismatches"], st["checks"],
                                          device_identity(dev)))
        allcs = [g[0] for g in gathered]
        allst = [g[1:5] for g in gathered]
        devs = [g[5] for g in gathered]
        # links_crossed: False whenever every rank sits on the SAME physical GPU (--same-device) -- such a line exercises the
        # entry point and the exchange code, it is NOT evidence for the multi-GPU row (no byte crossed xGMI)
        multi_report = {"links_crossed": len(set(devs)) > 1, "rank_devices": devs,
                        "ranks_bit_identical": all(c == allcs[0] for c in allcs),
                        "exchange": exchange["used"], "exchange_asked": exchange["asked"], "exchange_fallbacks": exchange["fallbacks"],
                        "p2p_error_code_max": max(int(x[0]) for x in allst), "p2p_timeouts": sum(int(x[1]) for x in allst),
                        "p2p_checksum_mismatches": sum(int(x[2]) for x in allst), "p2p_checksums_compared": int(allst[0][3]),
                        "bad_after_recovery": bool(bad),
                        # wall time of the start-up self-test of the peer exchange (four exchanges of a random multi-KB payload, both
                        # receive slots twice, each checked against torch.distributed's all-reduce: learner.enable_p2p)
                        "exchange_selftest_ms": getattr(learner, "p2p_selftest_ms", None)}
        # the same loop without any exchange (the ranks drift apart from here on: last thing this run does with them)
        if not args.no_exchange_leg and exchange["used"] in ("p2p", "coll", "rccl") and args.host_loop == "c":
            saved = (getattr(learner, "_p2p", None), getattr(learner, "_coll", None), exchange["used"])
            learner._p2p, learner._coll, exchange["used"] = None, None, "none"
            build_loop()
            timed(1, record=False)
            d0, _, _ = timed(max(2, min(args.steps, 8)), record=False)
            multi_report["ms_per_pass_no_exchange"] = max_over_ranks(d0) / (max(2, min(args.steps, 8)) * pps) * 1e3
            learner._p2p, learner._coll, exchange["used"] = saved
            build_loop()
            hot, use_c = state["hot"], state["use_c"]

    # ---- the step kernel, three ways (no correction terms):
    # (a) HIP event pairs around the launch inside the timed loop, on the launch stream (every ev_every-th pass).  A
    #     pair brackets the kernel PLUS the two marker packets: for a ~9 us kernel it reads 2-3 us above rocprofv3.
    if hot is not None:
        pair = hot.step_times_ms(1 << 16)
        k_pair_ms = float(np.mean(pair)) if len(pair) else float("nan")
    else:
        k_pair_ms = float(np.mean([a.elapsed_time(b) for a, b in py_events])) if py_events else float("nan")
    # host cost of enqueueing a pass, measured on a burst short enough not to hit the queue-depth back-pressure (the
    # enqueue time of the whole timed region above tracks the GPU once the HIP queue is full)
    t_burst = None
    if not multi:
        t1 = time.perf_counter()
        if hot is not None:
            hot.run(16)
            counter[0] = hot.counter
        else:
            for _ in range(16):
                one_pass()
        t_burst = (time.perf_counter() - t1) / 16
        torch.cuda.synchronize(dev)
        if hot is not None:
            hot.step_times_ms()               # drop the burst's event pairs

    # (b) back-to-back launches between ONE event pair (each launch includes its ~1.5 us dispatch boundary): the kernel
    #     the loop launches -- k_step_coop<policy> = get_action + step, when this env / net can take it -- and k_step alone
    it = args.env_only_iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def b2b(launch, reps=3):
        """`it` launches between one HIP event pair, `reps` times: the MEDIAN of the per-launch averages (one hiccup of the box --
        a 0.8 ms stall inside one of 200 launches was seen once in round 6 -- otherwise lands in the roofline's kernel_ms)."""
        out = []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            e0.record()
            for i in range(it):
                launch(i)
            e1.record()
            torch.cuda.synchronize(dev)
            out.append(e0.elapsed_time(e1) / it)
        return sorted(out)[len(out) // 2]

    k_b2b_ms = b2b(lambda i: ring.step_env(auto_reset=True))
    kp_b2b_ms = None
    # the launches as the C loop issues them: on one GPU, packed rows and the f32-MFMA net it stages layer 1 from the image it keeps
    # (csrc/loop.hip: UavLoop.img); the weights do not change during these legs, so a snapshot is that image
    loop_image = None
    if (fused and use_c and not multi and args.sample_lag == 0 and args.obs_dtype == "packed" and getattr(learner, "mfma", "f32") != "f16" and
            os.environ.get("UAVENV_LOOP_IMAGE", "1") != "0"):
        loop_image = learner.split_image()
    if fused and use_c and os.environ.get("UAVENV_NO_FUSED_ACT") is None and ring.step_policy(learner, args.eps, seed, 1 << 40):
        kp_n = [0]

        def _pol(i):
            kp_n[0] += 1
            ring.step_policy(learner, args.eps, seed, (1 << 40) + kp_n[0], image=loop_image)
        kp_b2b_ms = b2b(_pol)
    # (c) the rocprofv3 --kernel-trace average of this same command, from the committed profile (cannot be taken in-process)
    prof = committed_profile(args)
    k_prof_ms = prof.get("k_step_ms")
    kp_prof_ms = prof.get("k_step_policy_ms")
    k_ms = max(k_b2b_ms, k_prof_ms or 0.0)    # every roofline fraction is quoted on the LARGER of (b) and (c)
    kp_ms = max(kp_b2b_ms, kp_prof_ms or 0.0) if kp_b2b_ms is not None else None

    # ---- k_dqn_grad back to back (fused learner): same ring, same batch, fresh draws per launch
    g_b2b_ms = None
    if fused:
        import ctypes as C
        from dqn_based_uav_3d_path_planer_amd import _lib
        part = learner.new_partials(args.batch)
        kind = 0 if args.trainer == "dqn" else 1
        s_ = torch.cuda.current_stream(dev).cuda_stream

        def grad(cn):
            _lib.check(learner.lib.uavenv_dqn_grad_img(C.byref(ring._c), ring.head, ring.filled, args.batch, seed, cn, None,
                                                       C.byref(learner.net), kind, learner.gamma, 0, None, None, part.data_ptr(),
                                                       None if loop_image is None else loop_image.data_ptr(), s_),
                       "uavenv_dqn_grad_img")
        for cn in range(5):
            grad(cn)
        g_n = [100]

        def _grad(i):
            g_n[0] += 1
            grad(g_n[0])
        g_b2b_ms = b2b(_grad)

    # achievable HBM bandwidth on THIS device, same run: device-to-device copy of 1 GiB, read + write bytes
    copy_gbs = measure_copy_gbs(dev) if rank == 0 else None

    # ---- resets: how many the loop consumes (the fraction of agent-steps that end an episode, counted on 64 extra steps with
    # the agent_done plane read back) against what the planner delivers (the start-up bank: measured above as t_plan; the
    # rolling refresh: rows committed during this run)
    p_reset = None
    if rank == 0:
        # Episode ends per agent-step UNDER THE POLICY the loop runs (get_action inside the step launch, epsilon as in the timed region),
        # counted on the device over 1 024 extra passes.  (Round 5 stepped 64 times with one frozen action frame: a constant steer never
        # reaches a sub-goal, every agent times out after Max_Step = 150 steps, and the window read whatever part of that it caught.)
        done_ = torch.zeros(env.N, dtype=torch.uint8, device=dev)
        tot = torch.zeros((), dtype=torch.int64, device=dev)
        n_extra, ok_ = 1024, fused and not args.no_reset_count
        for i_ in range(n_extra if ok_ else 0):
            if not ring.step_policy(learner, args.eps, seed, (1 << 41) + i_, agent_done=done_):
                ok_ = False
                break
            tot += done_.sum()
        if ok_:
            p_reset = int(tot.item()) / (float(n_extra) * env.N)
    refresh = env.replan_stats() if (rank == 0 and args.replan_every > 0) else None
    planner_rows_per_s = None
    if rank == 0:                      # the planner by itself (csrc/rrt.hip, the two LDS tiers): 16 384 fresh scenarios
        env.rrt_plan(1024, seed=991)
        torch.cuda.synchronize(dev)
        e0.record()
        env.rrt_plan(16384, seed=992)
        e1.record()
        torch.cuda.synchronize(dev)
        planner_rows_per_s = 16384 / (e0.elapsed_time(e1) * 1e-3)

    out = None
    if rank == 0:
        n_agents = env.N
        value = n_pass * n_agents * world_size / dt
        # algorithmic bytes are the SURVEY 8(d) figure for the observation the row stands for (f32: 604 B, f16: 404 B);
        # packed rows are a lossless image of the f32 row, so they are priced as f32 and simply move fewer bytes
        algo = 404 if args.obs_dtype == "f16" else ALGO_BYTES_PER_AGENT_STEP
        row_bytes = ring.obs.shape[-1] * ring.obs.element_size()
        in_loop_policy = kp_ms is not None               # the loop's launch is k_step_coop<policy>; else act + k_step
        model = step_bytes_model(row_bytes, n_agents, policy=in_loop_policy, records=ring.meta is not None)
        moved = int(round(model["total"]))               # everything the loop's step launch reads and writes, per agent-step
        stored = int(round(step_bytes_model(row_bytes, n_agents, policy=False, records=ring.meta is not None)["total"]))   # k_step alone
        r_ms = kp_ms if in_loop_policy else k_ms
        achieved = algo * n_agents / (r_ms * 1e-3) / 1e9
        traffic = prof.get("k_step_policy_traffic_bytes_per_launch" if in_loop_policy else "k_step_traffic_bytes_per_launch")
        ldt = args.mfma if fused else "f32"
        launches = (3 if in_loop_policy else 4) + (1 if exchange["used"] == "coll" else 0)
        if not multi:
            par = "one GPU (env shard x1, no exchange)"
        elif args.sync == "fedavg":
            par = "env-shard x%d + weight averaging every %d updates (all-reduce avg)" % (world_size, learner.fl_loop)
        else:
            par = "env-shard x%d + flat-bucket gradient sum per update: %s" % (world_size, {
                "p2p": "peer-to-peer over IPC-mapped HBM on the stream (csrc/p2p.hip)",
                "coll": "RCCL all-reduce enqueued from C on the stream (csrc/coll.hip)",
                "rccl": "torch.distributed all_reduce (RCCL) per update from Python",
                "none": "NO exchange (diagnostic: the ranks drift apart)"}.get(exchange["used"], str(exchange["used"])))
        out = {
            "metric": "env-steps/sec + learner updates/sec, PathPlan_City DQN",
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "passes_per_step": pps, "ms_per_pass": dt / n_pass * 1e3, "timed_region_ms": dt * 1e3,
            "learner_updates_per_s": n_pass / dt,
            "host_enqueue_ms_per_pass": None if t_burst is None else 1e3 * t_burst,
            "host_enqueue_ms_per_pass_timed_region": 1e3 * t_enq / n_pass,
            "learner_samples_per_s": n_pass * args.batch * world_size / dt,
            "env_only_steps_per_s": n_agents / (k_b2b_ms * 1e-3),
            "config": {"workload": "PathPlan_City 500x500x100, 26 buildings, 1 UAV/env, %d vectorised envs/GPU, %s, "
                                   "device replay %d transitions/GPU (BASELINE.json configs[%d])"
                                   % (args.envs, args.trainer.upper(), ring.capacity, args.config - 1),
                       "step_definition": "1 bench step = %d passes of (act -> env step + replay write -> sample + "
                                          "learner update) over the whole env batch" % pps,
                       "envs_per_gpu": args.envs, "learn_batch_per_gpu": args.batch, "obs_dtype": args.obs_dtype,
                       "host_loop": ("csrc/loop.hip (C, %d launches per pass%s)" % (launches, ": act+step, grad, reduce+Adam" if launches == 3 else ""))
                       if use_c else "python (ctypes per launch)",
                       "learner": "fused HIP kernels (%s MFMA)" % ldt if fused else "PyTorch-ROCm ops",
                       "learner_dtype": ldt if fused else ("f32" if args.obs_dtype == "f32" else "f16 autocast"),
                       "reset_bank": ("%d scenarios planned on the GPU (RRT, %.0f ms incl. env construction)"
                                      % (bank_size, t_plan * 1e3)) if args.bank == "gpu"
                       else "1024 packaged reference resets",
                       "resets": {"consumed_per_s": None if p_reset is None else value / world_size * p_reset,
                                  "episode_end_fraction_of_agent_steps": p_reset,
                                  "planner_rows_per_s": planner_rows_per_s,
                                  "refresh": None if refresh is None else dict(
                                      refresh, every_passes=args.replan_every, rows_per_slice=args.replan_count,
                                      rows_committed_per_s=refresh["rows_committed"] / (dt * (args.steps + max(args.warmup, 1)) / args.steps)),
                                  "note": "the reference plans a fresh RRT path at every reset (Agents/UAV.py:327-366); here resets draw "
                                          "from a bank of %d scenarios planned on the GPU at start-up%s" % (
                                              bank_size,
                                              ", turned over in the background by the loop (uavenv_replan_*): the shortfall is "
                                              "consumed_per_s - rows_committed_per_s" if refresh is not None else
                                              " and reused for the whole run (--replan-every N turns it over in the background)")},
                       "epsilon": args.eps, "parallelism": par,
                       "sample_lag": args.sample_lag,
                       "replay": ("prioritised (ReplayTree semantics, alpha 0.6, beta 0.4 + 0.001 per update, epsilon 0.01, clip 1) -- "
                                  "8 launches per pass from C") if args.per else "uniform without replacement"},
            "roofline": {"bound": "hbm",
                         "kernel": ("k_step_coop<policy> -- the launch the timed loop issues: get_action (Q(s) + epsilon-greedy) + "
                                    "update_PathPlan + state_PathPlan + replay write") if in_loop_policy
                         else "k_step (update_PathPlan + state_PathPlan + replay write; get_action is a separate launch here)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_stale": prof.get("stale") if traffic else None,
                         "traffic_source": prof.get("source") if traffic else None,
                         "algorithmic_bytes_per_agent_step": algo, "stored_bytes_per_agent_step": stored,
                         "agents_per_launch": n_agents, "bytes_model": model,
                         "counters_over_model": None if not traffic else traffic / float(moved * n_agents),
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         **physical_view(algo, moved, n_agents, r_ms, traffic, copy_gbs),
                         "kernel_ms": r_ms,
                         "kernel_ms_definition": "max(back-to-back launches between one HIP event pair in this run, "
                                                 "rocprofv3 --kernel-trace average of the committed profile of this command)",
                         "kernel_ms_back_to_back": kp_b2b_ms if in_loop_policy else k_b2b_ms,
                         "kernel_ms_rocprofv3_committed": kp_prof_ms if in_loop_policy else k_prof_ms,
                         "kernel_ms_event_pair_in_loop": k_pair_ms,
                         # secondary: the step kernel WITHOUT the policy in its prologue (not what the loop runs)
                         "k_step_alone": {"kernel_ms": k_ms, "kernel_ms_back_to_back": k_b2b_ms, "kernel_ms_rocprofv3_committed": k_prof_ms,
                                          "achieved_algorithmic_GBs": algo * n_agents / (k_ms * 1e-3) / 1e9,
                                          "frac_algorithmic": algo * n_agents / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "frac_physical_stored": stored * n_agents / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        quote_physical_first(out["roofline"], args.obs_dtype == "packed")
        out["rendezvous_retries"] = int(os.environ.get("UAVENV_RDZV_RETRIES", "0"))
        out.update(multi_report)
        if g_b2b_ms is not None:
            fl = learner_flops_per_sample(args.trainer)
            g_prof_ms = prof.get("k_dqn_grad_ms")
            g_ms = max(g_b2b_ms, g_prof_ms or 0.0)
            peak = MFMA_F16_PEAK_TF if ldt == "f16" else MFMA_F32_PEAK_TF
            tf = fl * args.batch / (g_ms * 1e-3) / 1e12
            row = 413 if args.obs_dtype == "f16" else 813
            out["roofline_learner"] = {
                "bound": "mfma", "kernel": "k_dqn_grad (sample + gather + forward/backward of the 100-64-A MLPs)",
                "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "flops_per_sample": fl, "samples_per_launch": args.batch,
                "kernel_ms": g_ms, "kernel_ms_back_to_back": g_b2b_ms, "kernel_ms_rocprofv3_committed": g_prof_ms,
                "reduce_adam_ms_rocprofv3_committed": prof.get("k_dqn_reduce_adam_ms"),
                "hbm_GBs": row * args.batch / (g_ms * 1e-3) / 1e9, "algorithmic_bytes_per_sample": row,
                "mfma_busy_frac_pmc": prof.get("k_dqn_grad_mfma_busy_frac"),
                "traffic": prof.get("k_dqn_grad_traffic_bytes_per_launch"),
                "counters_stale": prof.get("stale") if prof.get("k_dqn_grad_traffic_bytes_per_launch") else None}
    if hot is not None:
        hot.close()
    if fused and multi:
        torch.cuda.synchronize(dev)
        dist.barrier()                       # no rank unmaps its peers while one of them may still be running a pull
        learner.disable_p2p()
    env.close()
    return out


def emit(full: dict, args) -> None:
    """Rank 0's output.  Child runs (--full-line: other_configs' children, scripts) print the whole dict; everything else prints the
    bounded headline as the LAST stdout line, with the whole dict in bench_full.json."""
    if args.full_line or full.get("mode") == "env-only":
        print(json.dumps(full), flush=True)
        return
    sys.stdout.flush()
    print(json.dumps(headline_line(full, write_side_files(full))), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the env hot path has no CPU fallback")
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world_size)
    if args.config == 4:
        out4 = run_config4(args, dev, world_size, rank)
        if rank == 0:
            emit(out4, args)
        if world_size > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    out = run_dqn(args, world_size, rank, dev)
    if rank == 0 and out is not None:
        headline = (args.config == 2 and world_size == 1 and not args.env_only and not args.explicit and args.sample_lag == 0
                    and not args.per)
        if headline and not args.no_other_configs:
            out["other_configs"] = other_configs(args)
        if not args.no_cpu_baseline and world_size == 1 and not args.env_only:
            out["cpu_baseline"] = cpu_baseline(args.envs, args.cpu_seconds)
        emit(out, args)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
