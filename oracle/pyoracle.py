"""ctypes binding of the CPU oracle (oracle/uav_oracle.c).

TEST INFRASTRUCTURE.  Allowed importers: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KMAX = 128
OBS_DIM = 100
INFO_NAMES = ("normal", "success", "lose")


class Building(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cx", "cy", "cz", "R", "H", "vx", "vy", "vz")]


class World(C.Structure):
    _fields_ = [("len", C.c_double), ("width", C.c_double), ("h", C.c_double),
                ("nb", C.c_int32), ("_pad", C.c_int32), ("b", C.POINTER(Building))]


class Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int32), ("ext_n", C.c_int32), ("ext_i", C.c_int32),
                ("ext", C.POINTER(C.c_double))]


class Uav(C.Structure):
    _fields_ = (
        [("max_v", C.c_double), ("steering_angle", C.c_double), ("max_step", C.c_int32), ("apf_enabled", C.c_int32)]
        + [(n, C.c_double) for n in ("P_i", "v_0", "d_0", "rho", "s", "A", "P_b", "F_b", "xi")]
        + [(n, C.c_double) for n in ("px", "py", "pz", "vx", "vy", "vz", "V", "gx", "gy", "gz")]
        + [("step", C.c_int32), ("done", C.c_int32), ("n_sub", C.c_int32), ("reach_goal", C.c_int32)]
        + [(n, C.c_double) for n in ("score", "total_score", "path_len")]
        + [("train_epoch", C.c_int64)]
        + [(n, C.c_double) for n in ("v_dir", "start2goal", "len_astar")]
        + [("error", C.c_int32), ("sub0_alias", C.c_int32), ("sub", (C.c_double * 3) * KMAX)]
    )


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL)


def _load(name: str) -> C.CDLL:
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    d = C.c_double
    lib.orc_calculate_angle.restype = d
    lib.orc_calculate_angle.argtypes = [d, d, d, d]
    lib.orc_distance.restype = d
    lib.orc_distance.argtypes = [d] * 6
    lib.orc_threaten_rate.restype = C.c_int
    lib.orc_threaten_rate.argtypes = [C.POINTER(World), d, d, d]
    lib.orc_threaten_rate_many.restype = None
    lib.orc_threaten_rate_many.argtypes = [C.POINTER(World), C.c_int64, C.c_void_p, C.c_void_p]
    lib.orc_calc_v.restype = d
    lib.orc_calc_v.argtypes = [C.POINTER(Uav)]
    lib.orc_calc_fly_power.restype = d
    lib.orc_calc_fly_power.argtypes = [C.POINTER(Uav)]
    lib.orc_update_pathplan.restype = None
    lib.orc_update_pathplan.argtypes = [C.POINTER(World), C.POINTER(Uav), d, C.POINTER(d),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.orc_state_pathplan.restype = None
    lib.orc_state_pathplan.argtypes = [C.POINTER(World), C.POINTER(Uav), C.c_void_p]
    lib.orc_rng_seed.restype = None
    lib.orc_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
    lib.orc_rng_external.restype = None
    lib.orc_rng_external.argtypes = [C.POINTER(Rng), C.c_void_p, C.c_int32]
    lib.orc_rng_random.restype = d
    lib.orc_rng_random.argtypes = [C.POINTER(Rng)]
    lib.orc_rng_uniform.restype = d
    lib.orc_rng_uniform.argtypes = [C.POINTER(Rng), d, d]
    lib.orc_reset.restype = None
    lib.orc_reset.argtypes = [C.POINTER(World), C.POINTER(Uav), C.POINTER(Rng), d]
    lib.orc_rrt_get_path.restype = C.c_int
    lib.orc_rrt_get_path.argtypes = [C.POINTER(World), C.POINTER(Rng), d, C.c_int, d, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.orc_step_many.restype = None
    lib.orc_step_many.argtypes = [C.POINTER(World), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_rollout_many.restype = C.c_int64
    lib.orc_rollout_many.argtypes = [C.POINTER(World), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_sizeof_uav.restype = C.c_int
    lib.orc_max_threads.restype = C.c_int
    assert lib.orc_sizeof_uav() == C.sizeof(Uav), (lib.orc_sizeof_uav(), C.sizeof(Uav))
    return lib


_LIBS: dict = {}


def lib(fast: bool = False) -> C.CDLL:
    name = "libuav_oracle_fast.so" if fast else "libuav_oracle.so"
    if name not in _LIBS:
        _LIBS[name] = _load(name)
    return _LIBS[name]


class OracleWorld:
    """Envs/PathPlan_City.py world: box + cylinders (+ optional velocities for APF)."""

    def __init__(self, buildings: np.ndarray, length=500.0, width=500.0, h=100.0, velocities=None, fast=False):
        self.lib = lib(fast)
        b = np.asarray(buildings, dtype=np.float64).reshape(-1, 5)
        self.nb = len(b)
        self._arr = (Building * max(self.nb, 1))()
        for i in range(self.nb):
            v = (0.0, 0.0, 0.0) if velocities is None else tuple(float(x) for x in velocities[i])
            self._arr[i] = Building(*[float(x) for x in b[i]], *v)
        self.w = World(float(length), float(width), float(h), self.nb, 0, self._arr)

    def threaten_rate(self, x, y, z) -> int:
        return int(self.lib.orc_threaten_rate(C.byref(self.w), float(x), float(y), float(z)))

    def threaten_rate_many(self, pts: np.ndarray) -> np.ndarray:
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(len(pts), dtype=np.int32)
        self.lib.orc_threaten_rate_many(C.byref(self.w), len(pts), pts.ctypes.data, out.ctypes.data)
        return out


def default_uav_params(world_npz) -> dict:
    p = world_npz["power"]
    return dict(max_v=float(world_npz["max_v"]), steering_angle=float(world_npz["steering_angle"]),
                max_step=int(world_npz["max_step"]), apf_enabled=0,
                P_i=p[0], v_0=p[1], d_0=p[2], rho=p[3], s=p[4], A=p[5], P_b=p[6], F_b=p[7], xi=p[8])


class OracleUav:
    """One Agents/UAV.py agent restricted to the PathPlan hot path."""

    def __init__(self, world: OracleWorld, params: dict):
        self.world = world
        self.lib = world.lib
        self.u = Uav()
        for k, v in params.items():
            setattr(self.u, k, v)

    # -- state injection -------------------------------------------------
    def set_state(self, px, py, pz, vx, vy, gx, gy, gz, step, sub_goals, calc_v=True):
        u = self.u
        u.px, u.py, u.pz, u.vx, u.vy, u.vz = float(px), float(py), float(pz), float(vx), float(vy), 0.0
        u.gx, u.gy, u.gz = float(gx), float(gy), float(gz)
        u.step, u.done, u.reach_goal, u.error, u.sub0_alias = int(step), 0, 0, 0, 0
        u.score = u.total_score = u.path_len = 0.0
        sub_goals = np.asarray(sub_goals, dtype=np.float64).reshape(-1, 3)
        assert len(sub_goals) <= KMAX
        u.n_sub = len(sub_goals)
        if len(sub_goals):
            C.memmove(C.addressof(u.sub), np.ascontiguousarray(sub_goals).ctypes.data, 24 * len(sub_goals))
        if calc_v:
            u.V = self.lib.orc_calc_v(C.byref(u))

    def sub_goals(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.u.sub)[: self.u.n_sub].copy()

    def state_vec(self):
        u = self.u
        return [u.px, u.py, u.pz, u.vx, u.vy, u.V, u.gx, u.gy, u.gz, float(u.step), float(u.done), float(u.n_sub),
                u.score, u.total_score, u.path_len, float(u.reach_goal)]

    # -- reference API ---------------------------------------------------
    def reset(self, rng: "OracleRng", sub_granularity=30.0):
        self.lib.orc_reset(C.byref(self.world.w), C.byref(self.u), C.byref(rng.r), float(sub_granularity))

    def update(self, a0: float):
        r, d, i = C.c_double(), C.c_int32(), C.c_int32()
        self.lib.orc_update_pathplan(C.byref(self.world.w), C.byref(self.u), float(a0), C.byref(r), C.byref(d),
                                     C.byref(i))
        return r.value, bool(d.value), int(i.value)

    def state(self) -> np.ndarray:
        o = np.zeros(OBS_DIM, dtype=np.float64)
        self.lib.orc_state_pathplan(C.byref(self.world.w), C.byref(self.u), o.ctypes.data)
        return o

    def fly_power(self) -> float:
        return float(self.lib.orc_calc_fly_power(C.byref(self.u)))


class OracleRng:
    """CPython's random.Random(seed) stream (MT19937)."""

    def __init__(self, seed: int, fast=False):
        self.lib = lib(fast)
        self.r = Rng()
        self.lib.orc_rng_seed(C.byref(self.r), int(seed))

    def replay(self, uniforms: np.ndarray):
        """Make random() return the given U[0,1) values in order (keeps a reference to the array)."""
        self._ext = np.ascontiguousarray(uniforms, dtype=np.float64)
        self.lib.orc_rng_external(C.byref(self.r), self._ext.ctypes.data, len(self._ext))
        return self

    def random(self) -> float:
        return float(self.lib.orc_rng_random(C.byref(self.r)))

    def uniform(self, a, b) -> float:
        return float(self.lib.orc_rng_uniform(C.byref(self.r), float(a), float(b)))


def rrt_get_path(world: "OracleWorld", rng: "OracleRng", start, goal, step_size=30.0, max_iter=10000, obstacle_step=5.0,
                 cap=KMAX):
    """PathPlan/RRT.py:63-105 -> (path [n,3], iterations)."""
    s = np.ascontiguousarray(start, dtype=np.float64)
    g = np.ascontiguousarray(goal, dtype=np.float64)
    out = np.zeros((cap, 3))
    it = C.c_int(0)
    n = world.lib.orc_rrt_get_path(C.byref(world.w), C.byref(rng.r), float(step_size), int(max_iter), float(obstacle_step),
                                   s.ctypes.data, g.ctypes.data, out.ctypes.data, cap, C.byref(it))
    return (out[:n].copy() if n >= 0 else None), it.value


class OracleBatch:
    """N independent agents stepped by orc_step_many (OpenMP): the timed CPU baseline."""

    def __init__(self, world: OracleWorld, params: dict, n: int):
        self.world, self.lib, self.n = world, world.lib, n
        self.arr = (Uav * n)()
        self.view = np.frombuffer(self.arr, dtype=np.dtype(Uav))   # structured view: vectorised field access
        for k, v in params.items():
            self.view[k] = v

    def set_from_state16(self, st, sub, alias):
        """Load agents from the [n,16] layout of uavenv_get_state / gen_golden.uav_state_vec (+ sub-goals)."""
        v = self.view
        for k, name in enumerate(("px", "py", "pz", "vx", "vy", "V", "gx", "gy", "gz")):
            v[name] = st[:, k]
        v["vz"] = 0.0
        v["step"], v["done"], v["n_sub"] = st[:, 9].astype(np.int32), st[:, 10].astype(np.int32), st[:, 11].astype(np.int32)
        v["score"], v["total_score"], v["path_len"] = st[:, 12], st[:, 13], st[:, 14]
        v["reach_goal"] = st[:, 15].astype(np.int32)
        v["sub0_alias"] = np.asarray(alias, dtype=np.int32)
        v["error"] = 0
        k = min(sub.shape[1], KMAX)
        v["sub"][:, :k] = sub[:, :k]

    def load_scenarios(self, start, goal, heading, sub_goals, n_sub, max_v=1.0):
        m = len(start)
        for i in range(self.n):
            s = i % m
            u = self.arr[i]
            u.px, u.py, u.pz = (float(x) for x in start[s])
            u.gx, u.gy, u.gz = (float(x) for x in goal[s])
            u.vx, u.vy, u.vz = max_v * float(np.cos(heading[s])), max_v * float(np.sin(heading[s])), 0.0
            u.V = self.lib.orc_calc_v(C.byref(u))
            u.step = u.done = 0
            u.sub0_alias = 1 if int(n_sub[s]) >= 2 else 0
            u.n_sub = int(n_sub[s])
            C.memmove(C.addressof(u.sub), np.ascontiguousarray(sub_goals[s], dtype=np.float64).ctypes.data,
                      int(n_sub[s]) * 24)

    def rollout(self, n_steps: int, bank_start_goal, bank_sub, bank_nsub, seed: int = 0, want_obs=True, nthreads=0):
        """orc_rollout_many: n_steps of update + state for every agent inside C (auto-reset from the bank).
        -> (agent-steps executed, sum of rewards)."""
        sg = np.ascontiguousarray(bank_start_goal, dtype=np.float64).reshape(-1, 6)
        sub = np.ascontiguousarray(bank_sub, dtype=np.float64)
        ns = np.ascontiguousarray(bank_nsub, dtype=np.int32)
        if want_obs and getattr(self, "_obs", None) is None:
            self._obs = np.zeros((self.n, OBS_DIM))
        rsum = C.c_double(0.0)
        done = self.lib.orc_rollout_many(C.byref(self.world.w), C.addressof(self.arr), self.n, int(n_steps),
                                         sg.ctypes.data, sub.ctypes.data, ns.ctypes.data, len(sg), sub.shape[1],
                                         int(seed), self._obs.ctypes.data if want_obs else None, C.byref(rsum),
                                         int(nthreads))
        return int(done), float(rsum.value)

    def step(self, a0: np.ndarray, want_obs=True, nthreads=0):
        n = self.n
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        r = np.zeros(n)
        d = np.zeros(n, dtype=np.int32)
        info = np.zeros(n, dtype=np.int32)
        obs = np.zeros((n, OBS_DIM)) if want_obs else None
        self.lib.orc_step_many(C.byref(self.world.w), C.addressof(self.arr), n, a0.ctypes.data, r.ctypes.data,
                               d.ctypes.data, info.ctypes.data, obs.ctypes.data if want_obs else None, nthreads)
        return r, d, info, obs
