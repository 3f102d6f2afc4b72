"""Cylinder obstacle with the reference's constructor contract (Obstacles/building.py:6-26): `param` is the XML
dict of one <Threaten> element.  On the hot path the cylinders live in LDS; this object is the host-side view."""
import math

from dqn_based_uav_3d_path_planer_amd.compat import Loc, None2Value


class building:
    def __init__(self, param, env=None):
        pos = param.get("position")
        self.position = Loc(float(pos.get("x")), float(pos.get("y")), float(pos.get("z")))
        self.position_ = self.position
        self.type = param.get("type")
        self._R = None2Value(float(param.get("_R")), 10)
        self._H = None2Value(float(param.get("_H")), 20)

    def reset(self):
        pass

    def reset_random(self):
        pass

    def check_threaten(self, position: Loc):
        if position.z > self._H:
            return 0
        d = math.sqrt((position.x - self.position.x) ** 2 + (position.y - self.position.y) ** 2 + 0.0)
        return 1 if d < self._R else 0

    def run(self):
        pass
