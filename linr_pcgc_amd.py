"""Import alias: the package directory is named ``linr-pcgc_amd`` (not a Python identifier); this module gives it the
importable name ``linr_pcgc_amd`` by pointing ``__path__`` at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'linr-pcgc_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f, _os
