"""Import alias: the package directory is `kbot-joystick_amd/` (not a valid Python identifier), so this
shim points `kbot_joystick_amd.__path__` at it and runs its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "kbot-joystick_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
