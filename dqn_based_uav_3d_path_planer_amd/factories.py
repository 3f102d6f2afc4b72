"""Reflection factories with the reference's contract (FactoryClass/*.py): import the module named by the
`*_Type` string, instantiate the class of the same name, and on ANY exception print it and return None.
The plugin directory is put on sys.path at import, exactly as the reference's factories append theirs."""
from __future__ import annotations

import importlib
import os
import sys

PLUGIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins")
if PLUGIN_DIR not in sys.path:
    sys.path.insert(0, PLUGIN_DIR)


def _create(type_key, param, *extra):
    try:
        type_name = param.get(type_key)
        module = importlib.import_module(type_name)
        return getattr(module, type_name)(param, *extra)
    except Exception as e:      # FactoryClass/EnvFactory.py:21-23: swallowed, printed, None
        print(e.args)
        return None


class EnvFactory:
    def Create_Env(self, param):                      # EnvFactory.py:12-25
        return _create("Env_Type", param)


class AgentFactory:
    def Create_Agent(self, param, env=None):          # AgentFactory.py:11-27
        return _create("Agent_Type", param, env)


class ThreatenFactory:
    def Create_Threaten(self, param, env=None):       # ThreatenFactory.py:11-24
        return _create("Threaten_Type", param, env)


class TrainerFactory:
    def Create_Trainer(self, param):                  # TrainerFactory.py:10-22
        return _create("Trainer_Type", param)


class NetworkFactory:
    def Create_Network(self, param):                  # NetworkFactory.py:10-22 (module BaseCNN, class by name)
        try:
            from .nets import create_network
            return create_network(param)
        except Exception as e:
            print(e.args)
            return None
