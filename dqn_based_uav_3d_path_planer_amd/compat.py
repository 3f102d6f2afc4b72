"""Host-side value types and helpers with the reference's names and semantics (BaseClass/CalMod.py), so code
written against the reference (`Loc`, `XML2Dict`, `Eu_Loc_distance`, `calculate_angle`, `None2Value`) runs
unchanged against the plugins.  XML2Dict needs no xmltodict: the six config files only use elements with text."""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET


class Loc:
    """3-vector (CalMod.py:17-54)."""

    def __init__(self, x, y, z) -> None:
        self.x, self.y, self.z = x, y, z

    def Set_Value(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    def Copy_From(self, p2):
        self.x, self.y, self.z = p2.x, p2.y, p2.z

    def __eq__(self, other):
        if other is None:
            return False
        return self.x == other.x and self.y == other.y and self.z == other.z

    def __add__(self, other):
        if not isinstance(other, Loc):
            raise ValueError("Can only add a Loc object with another Loc object")
        return Loc(self.x + other.x, self.y + other.y, self.z + other.z)

    def distance(self, other):
        return math.sqrt((self.x - other.x) ** 2 + (self.y - other.y) ** 2 + (self.z - other.z) ** 2)

    def __hash__(self):
        return hash((self.x, self.y, self.z))

    def __lt__(self, other):
        return False

    def __repr__(self):
        return f"Loc({self.x}, {self.y}, {self.z})"


def None2Value(value1, value2=None):
    return value2 if value1 is None else value1


def Eu_Loc_distance(loc1, loc2):
    """CalMod.py:64-65"""
    return math.sqrt((loc1.x - loc2.x) ** 2 + (loc1.y - loc2.y) ** 2 + (loc1.z - loc2.z) ** 2)


def calculate_angle(p1: Loc, p2: Loc, mod=1):
    """CalMod.py:89-102 -- heading p1->p2; degrees if mod == 0 else radians in [0, 2*pi]."""
    angle = math.degrees(math.atan2(p2.y - p1.y, p2.x - p1.x))
    if mod == 0:
        return (angle + 360) % 360
    return (angle + 360) % 360 / 180 * math.pi


def calculate_path_len(path):
    """CalMod.py:133-139 (consumes the first element like the reference)."""
    total = 0
    pre_p = path.pop(0)
    for p in path:
        total += Eu_Loc_distance(pre_p, p)
        pre_p = p
    return total


def _node_to_obj(node):
    children = list(node)
    if not children:
        text = (node.text or "").strip()
        return text if text != "" else None
    out = {}
    for ch in children:
        val = _node_to_obj(ch)
        if ch.tag in out:
            if not isinstance(out[ch.tag], list):
                out[ch.tag] = [out[ch.tag]]
            out[ch.tag].append(val)
        else:
            out[ch.tag] = val
    return out


def XML2Dict(file_path):
    """CalMod.py:117-124: the file as nested dicts of strings (repeated tags -> lists, empty -> None)."""
    with open(file_path, "r") as f:
        xml_data = f.read().lstrip()
    root = ET.fromstring(xml_data.encode("utf-8"))
    return {root.tag: _node_to_obj(root)}
