"""Import alias: the package directory is named `rt-octree_amd` (with a hyphen, as the build
contract asks), which Python cannot import by name.  This module turns itself into that package,
so `import rt_octree_amd` and `import rt_octree_amd.synth` work."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rt-octree_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
