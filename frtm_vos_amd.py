"""Import alias: the package directory is named ``frtm-vos_amd/`` (not a valid Python identifier).

``import frtm_vos_amd`` turns this module into a package whose search path is that directory,
so ``frtm_vos_amd.model.tracker`` etc. resolve to ``frtm-vos_amd/model/tracker.py``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'frtm-vos_amd')]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f
