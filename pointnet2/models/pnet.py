"""`Pnet2Stage` (reference: pointnet2/models/pnet.py:7-40) on the HIP-backed `Mlp_plus_t_emb`: a two-stage PointNet global
feature -- per-point Mlp, max-pool, [per-point | global] Mlp, max-pool.  Parameter names equal the reference's."""
import torch
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import Mlp_plus_t_emb


class Pnet2Stage(nn.Module):
    def __init__(self, mlp1, mlp2, bn=True, remove_last_activation=True):
        super().__init__()
        kw = dict(bn=bn, t_dim=0, include_t=False, bn_first=False, bias=True, first_conv=False, first_conv_in_channel=0,
                  res_connect=False, include_condition=False, condition_dim=128)
        self.mlp1 = Mlp_plus_t_emb(list(mlp1), **kw)
        if remove_last_activation:
            self.mlp1.second_mlp = self.mlp1.second_mlp[0:1]
        self.mlp2 = Mlp_plus_t_emb([2 * mlp1[-1]] + list(mlp2), **kw)
        if remove_last_activation:
            self.mlp2.second_mlp = self.mlp2.second_mlp[0:1]

    def forward(self, x):
        """x (B, mlp1[0], N) -> (B, mlp2[-1])"""
        f = self.mlp1(x.unsqueeze(-1).contiguous())
        g = f.max(dim=2, keepdim=True)[0].expand(-1, -1, f.size(2), -1)
        f = self.mlp2(torch.cat([f, g], dim=1).contiguous())
        return f.max(dim=2)[0].squeeze(-1)
