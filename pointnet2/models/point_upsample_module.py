"""Point splitting used by the autoencoder decoder (reference: pointnet2/models/point_upsample_module.py:4-46).

Every coarse point emits `factor` children: child = parent + displacement * output_scale / sqrt(factor).  With
`first_refine_coarse_points` the first F displacement channels move the parent itself first; with
`include_displacement_center_to_final_output` the refined parents are appended after the children."""
import numpy as np
import torch


def point_upsample(coarse, displacement, point_upsample_factor, include_displacement_center_to_final_output=False,
                   output_scale_factor_value=0.001, first_refine_coarse_points=False):
    B, N, F = coarse.shape
    if include_displacement_center_to_final_output and not first_refine_coarse_points:
        raise AssertionError("the refined centres can only be emitted when they are refined first")
    scale = output_scale_factor_value
    grid = 1.0 / np.sqrt(point_upsample_factor)
    parents = coarse
    if first_refine_coarse_points:
        parents = coarse + displacement[:, :, :F] * scale
        displacement = displacement[:, :, F:]
    n_child = displacement.shape[2] // F
    children = parents.unsqueeze(2) + (displacement * grid).reshape(B, N, n_child, F) * scale
    children = children.reshape(B, N * n_child, F)
    if include_displacement_center_to_final_output:
        children = torch.cat([children, parents], dim=1)
    return children.contiguous()
