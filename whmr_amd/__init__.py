"""Import alias: ``whmr_amd`` -> the on-disk package directory ``w-hmr_amd/`` (a hyphen is not importable)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'w-hmr_amd')
__path__ = [_real]
__file__ = _os.path.join(_real, '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
