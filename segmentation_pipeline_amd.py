"""Import shim: the package directory is named `segmentation-pipeline_amd/` (not a
valid Python identifier), so this module turns itself into that package:
`import segmentation_pipeline_amd` then sees models/, criterions/, ops, ...
"""
import os as _os

_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "segmentation-pipeline_amd")
__path__ = [_DIR]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(_DIR, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
