"""`pointnet2_ops._ext`: the nine native functions of the reference extension (_ext-src/src/bindings.cpp:6-19),
implemented in libslide_hip.so (see slide_amd/_ext.py and include/slide_hip.h)."""
from slide_amd._ext import (ball_query, furthest_point_sampling, gather_points, gather_points_grad, group_points,  # noqa: F401
                            group_points_grad, three_interpolate, three_interpolate_grad, three_nn)
