"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/uavenv.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dqn_based_uav_3d_path_planer_amd import _build, _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "uavenv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uavenv_\w+)\s*\(", text)))


def test_library_builds_and_loads():
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.uavenv_abi_version() == _lib.ABI_VERSION


def test_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 16
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"libuavenv.so does not export {name}"


def test_loop_internal_entry_points_are_exported():
    """csrc/dqn_internal.hpp: the C loop's image variants of three ABI entry points -- not part of include/uavenv.h, but bench.py
    times the loop's launches through them and _lib.py binds them: they have to be there, and only there."""
    lib = _lib.load()
    text = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "dqn_internal.hpp")).read()
    names = sorted(set(re.findall(r"\bint\s+(uavenv_\w+)\s*\(", text)))
    assert names == ["uavenv_dqn_adam_img", "uavenv_dqn_adam_p2p_img", "uavenv_dqn_grad_img", "uavenv_dqn_reduce_adam_img",
                     "uavenv_dqn_split_image", "uavenv_step_policy_img"]
    declared = set(header_symbols())
    for name in names:
        assert hasattr(lib, name), name
        assert name not in declared
    assert int(re.search(r"#define UAVENV_DQN_IMAGE_FLOATS (\d+)", text).group(1)) == _lib.DQN_IMAGE_FLOATS


def test_config_struct_layout_matches_header():
    # 10 int32 + 5 doubles + 8 doubles + 1 double, naturally aligned
    assert ctypes.sizeof(_lib.UavEnvConfig) == 10 * 4 + 14 * 8
    assert ctypes.sizeof(_lib.UavReplayRing) == 5 * 8 + 4 * 4 + 8        # (+ meta, ABI 5)


def test_every_struct_layout_matches_the_header_as_gcc_sees_it(tmp_path):
    """sizeof / offsetof of every struct in include/uavenv.h, printed by a C program gcc builds against the header,
    must equal the ctypes mirror in _lib.py."""
    import subprocess
    structs = {"UavEnvConfig": _lib.UavEnvConfig, "UavReplayRing": _lib.UavReplayRing, "UavDqnNet": _lib.UavDqnNet,
               "UavPer": _lib.UavPer, "UavLoopConfig": _lib.UavLoopConfig, "UavLoopCursor": _lib.UavLoopCursor,
               "UavSacNets": _lib.UavSacNets, "UavSacBatch": _lib.UavSacBatch, "UavSacAdam": _lib.UavSacAdam,
               "UavSacLoopSlot": _lib.UavSacLoopSlot, "UavSacLoopConfig": _lib.UavSacLoopConfig,
               "UavSacLoopCursor": _lib.UavSacLoopCursor}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "uavenv.h"', 'int main(void){']
    for name, ct in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, ct in structs.items():
        assert int(got[name]) == ctypes.sizeof(ct), name
        for fname, _ in ct._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(ct, fname).offset, (name, fname)


def test_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = _lib.UavEnvConfig()
    cfg.abi_version = _lib.ABI_VERSION
    cfg.n_envs, cfg.uav_per_env, cfg.max_subgoals, cfg.max_step = 4, 1, 8, 150
    h = ctypes.c_void_p()
    rc = lib.uavenv_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == _lib.ENODEV
    assert b"no CPU fallback" in lib.uavenv_last_error()
    from dqn_based_uav_3d_path_planer_amd.env import VecPathPlanEnv
    with pytest.raises(_lib.UavEnvError):
        VecPathPlanEnv(4, [[1.0, 2.0, 0.0, 3.0, 4.0]])


def test_rejects_bad_config():
    lib = _lib.load()
    cfg = _lib.UavEnvConfig()
    cfg.abi_version = 999
    h = ctypes.c_void_p()
    assert lib.uavenv_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.EINVAL
    cfg.abi_version = _lib.ABI_VERSION
    cfg.n_envs, cfg.uav_per_env, cfg.max_subgoals, cfg.max_step = 4, 3, 8, 150   # 3 is not a power of two
    assert lib.uavenv_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.EINVAL
    assert lib.uavenv_create(None, ctypes.byref(h)) == _lib.EINVAL


def test_sac_partial_rows_follow_the_launch():
    """uavenv_sac_partial_rows_n (a host-side function: no device needed): the rows a grad launch writes per slot = its
    workgroups per slot -- as many 64-sample tiles per workgroup as bring the WHOLE launch down to one workgroup per CU, at
    most 8; UavSacBatch.tiles_per_wg pins it; the single-slot default never writes fewer rows than a shared launch."""
    from dqn_based_uav_3d_path_planer_amd import _lib
    lib = _lib.load()
    f = lib.uavenv_sac_partial_rows_n
    assert f(32768, 1, 0) == 256 and lib.uavenv_sac_partial_rows(32768) == 256      # 512 tiles, 2 per workgroup
    assert f(32768, 4, 0) == 64                                                     # BASELINE configs[3]: 8 per workgroup
    assert f(32768, 2, 0) == 128 and f(32768, 8, 0) == 64                           # capped at 8 tiles
    assert f(8192, 4, 0) == 64 and f(8192, 1, 0) == 128 and f(8192, 1, 2) == 64     # the pinned partition of the test
    assert f(64, 1, 0) == 1 and f(64, 8, 0) == 1 and f(640, 1, 3) == 4              # ragged: ceil(10 / 3)
    for bad in ((0, 1, 0), (100, 1, 0), (64, 0, 0), (64, 9, 0), (64, 1, -1)):
        assert f(*bad) == _lib.EINVAL
    for b in (64, 4096, 32768, 65536):
        for n in (1, 2, 4, 8):
            assert f(b, n, 0) <= lib.uavenv_sac_partial_rows(b) <= b // 64
