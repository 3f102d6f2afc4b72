"""Run the *actual* reference (young-how/DQN-based-UAV-3D_path_planer) safely.

TEST INFRASTRUCTURE ONLY.  This module is used by ``oracle/gen_golden.py`` in
the build container (where ``/root/reference`` exists) to generate the golden
vectors committed under ``tests/golden/``.  Nothing on the product path, in
``bench.py`` or in the ``-m gpu`` tests imports it: ``/root/reference`` does not
exist on the GPU box.

Safety (SURVEY.md "INCIDENT" + Appendix D): the reference is never imported in
place.  It is copied to a scratch directory first, because its trainers write
checkpoints next to their own ``__file__`` (Trainer/SAC_Trainer.py:109-119) and
the UAV constructor opens ``logs/*.csv`` relative to the CWD
(Agents/UAV.py:269-277).  Four missing third-party imports are shimmed:

* ``xmltodict``  -> an ElementTree stand-in (leaf -> stripped text or None,
  repeated tag -> list, root -> {tag: ...}); the only shim with semantics.
* ``pyecharts``, ``tkinter``  -> MagicMock (rendering is out of scope).
* ``mysql.connector``  -> stub whose ``connect`` raises ``Error`` (the reference
  tolerates a missing DB, DataBase/Connector.py:34-35).
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import types
import xml.etree.ElementTree as ET
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("UAV_REFERENCE_ROOT", "/root/reference")


def _etree_to_dict(node):
    children = list(node)
    if not children:
        text = (node.text or "").strip()
        return text if text != "" else None
    out = {}
    for ch in children:
        val = _etree_to_dict(ch)
        if ch.tag in out:
            if not isinstance(out[ch.tag], list):
                out[ch.tag] = [out[ch.tag]]
            out[ch.tag].append(val)
        else:
            out[ch.tag] = val
    return out


def _xmltodict_parse(xml_data, *args, **kwargs):
    if isinstance(xml_data, bytes):
        xml_data = xml_data.decode("utf-8")
    xml_data = xml_data.lstrip()
    root = ET.fromstring(xml_data.encode("utf-8"))
    return {root.tag: _etree_to_dict(root)}


def install_shims():
    xm = types.ModuleType("xmltodict")
    xm.parse = _xmltodict_parse
    sys.modules.setdefault("xmltodict", xm)
    for name in ("tkinter", "pyecharts", "pyecharts.charts", "pyecharts.options"):
        sys.modules.setdefault(name, MagicMock())
    mysql = types.ModuleType("mysql")
    conn = types.ModuleType("mysql.connector")

    class Error(Exception):
        pass

    def connect(*a, **k):
        raise Error("no database in the oracle harness")

    conn.Error = Error
    conn.connect = connect
    mysql.connector = conn
    sys.modules.setdefault("mysql", mysql)
    sys.modules.setdefault("mysql.connector", conn)


class RefSession:
    """A scratch copy of the reference with ``simulator`` imported from it."""

    def __init__(self, trainer_xml: str | None = None, uav_xml_edits: dict | None = None,
                 env_xml_edits: dict | None = None):
        if not os.path.isdir(REFERENCE_ROOT):
            raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
        self.tmp = tempfile.mkdtemp(prefix="uavref_")
        self.root = os.path.join(self.tmp, "ref")
        shutil.copytree(REFERENCE_ROOT, self.root,
                        ignore=shutil.ignore_patterns("*.gif", "*.jpg", "doc", "DSNs"))
        os.chmod(self.root, 0o755)
        for d, _, fs in os.walk(self.root):
            os.chmod(d, 0o755)
            for f in fs:
                os.chmod(os.path.join(d, f), 0o644)
        os.makedirs(os.path.join(self.root, "DataBase", "experience"), exist_ok=True)
        os.makedirs(os.path.join(self.root, "logs"), exist_ok=True)
        # no stale checkpoints: trainers auto-load Mod/*.pth (SAC_Trainer.py:70-106)
        for f in os.listdir(os.path.join(self.root, "Mod")):
            if f.endswith(".pth"):
                os.remove(os.path.join(self.root, "Mod", f))
        if trainer_xml is not None:
            with open(os.path.join(self.root, "config", "Trainer.xml"), "w") as fh:
                fh.write(trainer_xml)
        for fname, edits in (("UAV.xml", uav_xml_edits), ("PathPlan_City.xml", env_xml_edits)):
            if edits:
                p = os.path.join(self.root, "config", fname)
                s = open(p).read()
                for tag, val in edits.items():
                    import re
                    s, n = re.subn(rf"<{tag}>[^<]*</{tag}>", f"<{tag}>{val}</{tag}>", s, count=1)
                    assert n == 1, (fname, tag)
                open(p, "w").write(s)
        self._old_cwd = os.getcwd()
        os.chdir(self.root)  # ./config/*.xml, logs/, path.csv are CWD-relative
        sys.dont_write_bytecode = True
        install_shims()
        sys.path.insert(0, self.root)
        import importlib
        self.simulator_mod = importlib.import_module("simulator")
        self.sim = self.simulator_mod.simulator()
        self.env = self.sim.env
        if self.env is None:
            raise RuntimeError("reference env failed to construct (factory swallowed the error)")
        self.uav = self.env.Agents[0]
        for a in self.env.Agents:
            if a.Trainer is not None:
                a.Trainer.save_loop = 10 ** 12
        # Agents/UAV.py does `from BaseClass.CalMod import *`: use that module object so
        # injected Loc instances pass Loc.__add__'s isinstance check (CalMod.py:34-36).
        self.CalMod = importlib.import_module("BaseClass.CalMod")

    def close(self):
        os.chdir(self._old_cwd)
        shutil.rmtree(self.tmp, ignore_errors=True)
