"""Importable alias for the `comfy-rvc_amd/` package directory.

The product directory name contains a hyphen (it mirrors the upstream project name), which Python cannot
import directly; this stub makes `import comfy_rvc_amd.<module>` resolve into `comfy-rvc_amd/`.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "comfy-rvc_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
