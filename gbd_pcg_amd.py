"""Import shim: the package directory is named `gbd-pcg_amd/` (not a legal Python
identifier), so `import gbd_pcg_amd` resolves here and this module turns itself into a
package whose search path is that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gbd-pcg_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
