"""Import shim: the package directory is named ``ciri-long_amd`` (not a valid Python identifier), so this module
stands in for it: ``import ciri_long_amd`` / ``from ciri_long_amd import ssw_wrap`` resolve into that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'ciri-long_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f
