"""Drop-in `pointnet2_ops` package backed by hand-written gfx950 HIP kernels (slide_amd).
Same public names as the reference package (pointnet2_ops_lib/pointnet2_ops/__init__.py:1-3 imports both submodules)."""
import pointnet2_ops.pointnet2_modules
import pointnet2_ops.pointnet2_utils
from pointnet2_ops._version import __version__
