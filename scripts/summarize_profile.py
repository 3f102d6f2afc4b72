#!/usr/bin/env python
"""Fold one scripts/profile_round.sh result (gpurun_out/<tag>_kernel_stats.csv + <tag>_pmc.txt) into
profiles/summary.json under a workload key (bench.py's committed_profile() reads it), and copy the raw files to
profiles/<tag>_*.   usage: summarize_profile.py <tag> <workload key, e.g. envs16384_batch16384_dqn_f32>"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
stats = os.path.join(src, f"{tag}_kernel_stats.csv")
pmc = os.path.join(src, f"{tag}_pmc.txt")
entry = {"files": f"profiles/{tag}_kernel_stats.csv, profiles/{tag}_pmc.txt"}
avg = {}
def coop_has_policy(full):
    """k_step_coop<MaskT, APF, OBS, POLICY[, PAHEAD]>: is the fourth template argument true?"""
    m = re.search(r"k_step_coop<([^>]*)>", full)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(",")]
    return len(args) >= 4 and args[3] == "true"


def short_name(full):
    """kernel family of a rocprofv3 kernel name; the step kernel that carries the policy in its prologue
    (k_step_coop<..., true>) is its own family"""
    if "k_step" in full:
        return "k_step_policy" if (coop_has_policy(full) or "k_step_polh" in full) else "k_step"
    for short in ("k_dqn_grad", "k_dqn_act", "k_dqn_reduce_adam", "k_apf_adjust", "k_sac_critic_grad", "k_sac_actor_grad", "k_sac_reduce_adam"):
        if short in full:
            return short
    return None


for r in csv.DictReader(open(stats)):
    short = short_name(r["Name"])
    if short and short not in avg:
        avg[short] = (float(r["AverageNs"]) * 1e-6, int(r["Calls"]))
for short, (ms, calls) in avg.items():
    entry[f"{short}_ms"] = ms
    entry[f"{short}_calls"] = calls
c = {}
for line in open(pmc):
    m = re.match(r"\S+ (\S+) (\S+) n=(\d+) mean=([0-9.eE+-]+)", line)
    if m:
        c[(m.group(1), m.group(2))] = float(m.group(4))
for k in ("k_step", "k_step_policy", "k_dqn_grad", "k_dqn_reduce_adam", "k_apf_adjust", "k_sac_critic_grad", "k_sac_actor_grad"):
    f, w = c.get((k, "FETCH_SIZE")), c.get((k, "WRITE_SIZE"))
    if f is not None and w is not None:
        # gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; KB units
        entry[f"{k}_traffic_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        entry[f"{k}_FETCH_SIZE_KB"], entry[f"{k}_WRITE_SIZE_KB"] = f, w
busy = c.get(("k_dqn_grad", "SQ_VALU_MFMA_BUSY_CYCLES"))
waves, wcyc = c.get(("k_dqn_grad", "SQ_WAVES")), c.get(("k_dqn_grad", "SQ_WAVE_CYCLES"))
if busy is not None and "k_dqn_grad_ms" in entry:
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1 024 SIMDs (= issue cycles x MFMA count: 64 per
    # v_mfma_f32_32x32x2_f32, 32 per 16x16x4_f32).  Utilisation = busy cycles per SIMD / kernel duration in cycles; the
    # duration is the UNPROFILED rocprofv3 kernel-trace average, priced at the 2.4 GHz maximum clock (the clock under
    # load is lower, so this is a lower bound on the fraction).
    per_simd = busy / 1024.0
    entry["k_dqn_grad_mfma_busy_cycles_per_simd"] = per_simd
    entry["k_dqn_grad_mfma_busy_frac"] = per_simd / (entry["k_dqn_grad_ms"] * 2.4e6)
    if waves and wcyc:
        entry["k_dqn_grad_wave_cycles_per_wave"] = 4.0 * wcyc / waves      # SQ_WAVE_CYCLES counts quad-cycles
for name in ("SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU_MFMA_F16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F16"):
    v = c.get(("k_dqn_grad", name))
    if v is not None:
        entry["k_dqn_grad_" + name] = v
# which kernel sources these counters belong to: bench.py compares this with the sources it runs (traffic_stale)
sys.path.insert(0, ROOT)
try:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_sha", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    entry["csrc_sha"] = mod.csrc_sha()
except Exception as ex:      # noqa: BLE001
    entry["csrc_sha"] = None
    print("csrc_sha unavailable:", ex)
path = os.path.join(dst, "summary.json")
allv = json.load(open(path)) if os.path.exists(path) else {}
allv[key] = entry
json.dump(allv, open(path, "w"), indent=1, sort_keys=True)
for f in (stats, pmc, os.path.join(src, f"{tag}_bench_under_rocprof.json")):
    if os.path.exists(f):
        shutil.copy(f, dst)
print(json.dumps(entry, indent=1))
