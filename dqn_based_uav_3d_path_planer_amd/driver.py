"""Host-side driver pieces mirroring simulator.py (reference root): the epsilon schedule and, below, a
`simulator`-compatible training loop over the vectorised env (see plugins/ for the reflection surface)."""
from __future__ import annotations


def epsilon_annealing(epoch: int, min_eps: float, max_eps_episode: float) -> float:
    """simulator.py:141-145 -- linear decay from 1.0 to min_eps over max_eps_episode episodes."""
    slope = (min_eps - 1.0) / (max_eps_episode + 0.1)
    return max(slope * epoch + 1.0, min_eps)


def write_buildings_xml(path: str, buildings) -> None:
    """Write cylinders [nb,5] = (cx,cy,cz,R,H) in the schema Envs/PathPlan_City.py:41-51 parses."""
    lines = ["<?xml version='1.0' encoding='utf-8'?>", "<buildings>"]
    for b in buildings:
        lines += ["    <Threaten>", "        <Threaten_Type>building</Threaten_Type>", "        <position>",
                  f"            <x>{float(b[0])!r}</x>", f"            <y>{float(b[1])!r}</y>",
                  f"            <z>{float(b[2])!r}</z>", "        </position>", f"        <_R>{float(b[3])!r}</_R>",
                  f"        <_H>{float(b[4])!r}</_H>", "    </Threaten>"]
    lines.append("</buildings>")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def make_config_dir(dst: str, trainer: str = "DQN", num_envs: int = 1, num_uav: int = 1, num_episodes: int = 20) -> str:
    """Materialise a reference-style ./config directory under `dst` (PathPlan_City.xml, UAV.xml, Trainer.xml,
    buildings.xml with the stock 26 cylinders).  Returns the path of PathPlan_City.xml."""
    import os
    import re
    import shutil
    from .data import load_city26
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
    cfg = os.path.join(dst, "config")
    os.makedirs(cfg, exist_ok=True)
    shutil.copy(os.path.join(src, "UAV.xml"), os.path.join(cfg, "UAV.xml"))
    shutil.copy(os.path.join(src, f"Trainer_{trainer}.xml"), os.path.join(cfg, "Trainer.xml"))
    s = open(os.path.join(src, "PathPlan_City.xml")).read()
    s = re.sub(r"<num_envs>\d+</num_envs>", f"<num_envs>{num_envs}</num_envs>", s)
    s = re.sub(r"<num_UAV>\d+</num_UAV>", f"<num_UAV>{num_uav}</num_UAV>", s)
    s = re.sub(r"<num_episodes>\d+</num_episodes>", f"<num_episodes>{num_episodes}</num_episodes>", s)
    out = os.path.join(cfg, "PathPlan_City.xml")
    open(out, "w").write(s)
    write_buildings_xml(os.path.join(cfg, "buildings.xml"), load_city26()["buildings"])
    return out


class simulator:
    """The training driver with simulator.py's surface (simulator.py:44-145): Init_From_XML -> EnvFactory ->
    env; StartAndTrain = 10 x (num_episodes/10) x { eps = epsilon_annealing(); info = env.run_eposide(eps) }.
    XML paths inside the env config are CWD-relative, as in the reference."""

    def __init__(self, xml_path: str = None) -> None:
        import os
        from .factories import EnvFactory
        self.TARGET_UPDATE = 10
        self.num_episodes = 150000
        self.min_eps = 0.1
        self.max_eps_episode = 1000
        self.EnvFactory = EnvFactory()
        self.epoch = 0
        self.env = None
        self.Max_score = -9999999999
        self.stat = 1
        self.infos = []
        self.executed_time = 0
        self.CsvWriter = None
        self.result_path = None
        self.Init_From_XML(xml_path or os.path.join(os.getcwd(), "config", "PathPlan_City.xml"))

    def Init_From_XML(self, XML_path):
        from .compat import XML2Dict
        try:
            cfg = XML2Dict(XML_path).get("simulator")
            self.record_epo = int(cfg.get("record_epo"))
            self.num_episodes = int(cfg.get("num_episodes"))
            self.max_eps_episode = int(cfg.get("max_eps_episode"))
            self.min_eps = float(cfg.get("min_eps"))
            self.TARGET_UPDATE = int(cfg.get("TARGET_UPDATE"))
            self.env = self.EnvFactory.Create_Env(cfg.get("env"))
        except Exception as e:      # simulator.py:101-103
            print(e.args)
            return None

    RESULT_HEADER = ["sum_Episode", "Episode", " Score", " Avg.Score", "eps-greedy", "success", "failed", "meet_threaten",
                     "loss", "step", "avg_trainning_time", "avg_testing_time", "total_time"]

    def Init_Record_Mod(self, result_path=None):
        """simulator.py:72-80: logs/score_<time>.csv with the reference's 13-column header (opt-in: call it, or pass
        a path).  One row per episode from record(info)."""
        import csv
        import datetime
        import os
        if result_path is None:
            cur = datetime.datetime.now().strftime("%m_%d_%Y(%H_%M_%S)")
            result_path = os.path.join("logs", "score_%s.csv" % cur)
        os.makedirs(os.path.dirname(result_path) or ".", exist_ok=True)
        self.result_path = result_path
        self._result_file = open(result_path, "a+", newline="")
        self.CsvWriter = csv.writer(self._result_file)
        self.CsvWriter.writerow(self.RESULT_HEADER)

    def record(self, info=None):
        """simulator.py:163-166 plus the four columns its header names but the reference's row leaves out (step,
        averaged train / test time per agent, wall time) -- the reference's own record_list() writes those."""
        if info is None or self.CsvWriter is None:
            return
        n = max(len(self.env.Agents), 1)
        tr = sum(u.Train_time for u in self.env.Agents) / n
        te = sum(u.Testing_time for u in self.env.Agents) / n
        self.CsvWriter.writerow([info["sum_epoch"], self.epoch, info["score"], info["average_score"], info["eps"],
                                 info["success"], info["lose"], info["meet_threaten"], info["loss"], info.get("step", 0),
                                 tr, te, self.executed_time])
        self._result_file.flush()

    def record_list(self):
        """simulator.py:151-161.  The reference zeroes the episode columns there and drops its info list; here the row
        carries the episode's numbers (record()) and `infos` keeps the history."""
        self.record(self.infos[-1] if self.infos else None)

    def epsilon_annealing(self):
        return epsilon_annealing(self.epoch, self.min_eps, self.max_eps_episode)

    def StartAndTrain(self, flag=0):
        import time
        for _ in range(10):
            for _ in range(int(self.num_episodes / 10)):
                if self.stat == 3:
                    return
                if self.stat == 2:
                    continue
                self.epoch += 1
                eps_rate = self.epsilon_annealing()
                t0 = time.time()
                info = self.env.run_eposide(eps_rate)
                self.executed_time += time.time() - t0
                self.infos.append(info)
                if self.Max_score < info["average_score"]:
                    self.Max_score = info["average_score"]
                self.record_list()

    def Update_target(self):
        for a in self.env.Agents:
            a.Trainer.hard_update()
