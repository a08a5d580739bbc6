"""Import shim: makes the directory ``chatterbox-vllm2_amd/`` (a name Python cannot import
directly because of the hyphen) importable as the package ``chatterbox_vllm2_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "chatterbox-vllm2_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
