"""Importable alias for the package directory `robot-3dlotus_amd/` (a hyphen is not a valid
Python identifier).  `import robot_3dlotus_amd` executes robot-3dlotus_amd/__init__.py and
resolves sub-modules from that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "robot-3dlotus_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
