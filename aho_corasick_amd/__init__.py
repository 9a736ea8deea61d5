"""Importable alias of the `aho-corasick_amd/` package (a hyphen cannot appear in `import`).

`import aho_corasick_amd` executes aho-corasick_amd/__init__.py with this module's __path__ pointing at
that directory, so submodules resolve there too.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "aho-corasick_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
