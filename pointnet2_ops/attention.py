"""`pointnet2_ops.attention` (reference pointnet2_ops_lib/pointnet2_ops/attention.py): MyGroupNorm (:6-23),
AttentionModule (:35-96, vector attention over the K neighbours), GlobalAttentionModule (:98-155) -- same constructor
signatures and state-dict names.  AttentionModule runs row-major (slide_amd.rows: GEMMs on [B * np * K][C] matrices, in-place
GroupNorm passes, one softmax-reduction kernel); GlobalAttentionModule (N x N scores, unused by the shipped configs) is a
tensor program over the HIP convolution / GroupNorm kernels."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from slide_amd import rows as R
from slide_amd.nn_ops import HipConv1x1, HipGroupNorm


class MyGroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_channels = num_channels - num_channels % num_groups
        self.num_groups = num_groups
        self.group_norm = HipGroupNorm(self.num_groups, self.num_channels)

    def forward(self, x, relu=False):
        return self.group_norm(x, relu=relu)  # tail channels (position info) pass through un-normalised


def count_to_mask(count, K):
    """(B, np) neighbour counts -> (B, np, K) bool: slot k holds a real neighbour"""
    return torch.arange(K, device=count.device, dtype=count.dtype).expand(count.shape + (K,)) < count.unsqueeze(-1)


class AttentionModule(nn.Module):
    def __init__(self, C_in1, C_in2, C1, C2, C_out, attention_bn=True, transform_grouped_feat_out=True,
                 last_activation=True):
        super().__init__()
        C1, C2 = max(C1, 32), max(C2, 32)
        self.feat_conv = HipConv1x1(C_in1, C1)
        self.grouped_feat_conv = HipConv1x1(C_in2, C2)
        inter_C = min(C1 + C2, C_out)
        if attention_bn:
            self.weight_conv = nn.Sequential(nn.ReLU(inplace=True), MyGroupNorm(min(32, C1 + C2), C1 + C2),
                                             HipConv1x1(C1 + C2, inter_C), nn.ReLU(inplace=True),
                                             MyGroupNorm(min(32, inter_C), inter_C), HipConv1x1(inter_C, C_out))
        else:
            self.weight_conv = nn.Sequential(nn.ReLU(inplace=True), HipConv1x1(C1 + C2, inter_C), nn.ReLU(inplace=True),
                                             HipConv1x1(inter_C, C_out))
        self.transform_grouped_feat_out = transform_grouped_feat_out
        if transform_grouped_feat_out:
            layers = [HipConv1x1(C_out, C_out)]
            if last_activation:
                if attention_bn:
                    layers.append(MyGroupNorm(min(32, C_out), C_out))
                layers.append(nn.ReLU(inplace=True))
            self.feat_out_conv = nn.Sequential(*layers)

    def _u_without_concat(self, query, grouped, K, wc):
        """round 5: relu(weight_conv.2(GN(relu([q | k])))) WITHOUT building the concatenation.  Both producers store their ReLU'd outputs
        and publish tile sums from their GEMM epilogues, the joint GroupNorm over the virtual concatenation is finalised from those
        (SLIDE_OP_ROWS_GN_JOINT), and weight_conv.2 is linear: its q half is evaluated once per POINT (1 / K of the MACs) and enters
        the k half's GEMM as a per-point term ahead of the ReLU -- what the fused DDPM plan does (slide_amd/engine.py _attention).
        Saves the concat + ReLU pass, the statistics pass over it and C1 / (C1 + C2) of the consumer GEMM's input bytes.  None when the
        shapes do not qualify (fp32 module mode, samples that are not whole 256-row tiles, K not a power of two): the caller concatenates."""
        if not (R.deferral() and R.fused_stats() and query.half and grouped.S % 256 == 0 and K & (K - 1) == 0
                and grouped.rows == query.rows * K and query.rows > 0 and os.environ.get("SLIDE_MODULE_SPLIT_QK", "1") != "0"):
            return None
        if query.S % 256 == 0:
            q1 = R.conv(query, self.feat_conv, stats="relu")
        else:  # fewer than a tile of points per sample (np = 16 / 64 / 128 levels): the q side is 1 / K of the rows -- its ReLU and
            q1 = R.conv(query, self.feat_conv)  # per-sample sums as small elementwise passes instead of the GEMM epilogue's
            v = q1.data.relu_().view(q1.B, q1.S, q1.ld).float()
            q1.stats = R.sample_sums(v) + (True,)  # (NOT torch.sum: its reduction order depends on the batch size)
        k1 = R.conv(grouped, self.grouped_feat_conv, stats="relu")
        C1, C2 = q1.C, k1.C
        if not R.joint_norm_qk(q1, k1, K, wc[1].group_norm):
            tot = R.concat_qk(q1, k1, K)  # (no tile statistics after all: the three-pass form on the outputs already computed)
            R.norm_act(tot, wc[1].group_norm, defer=True)
            return R.conv(tot, wc[2], stats="relu")
        sl = self.__dict__.get("_w2_slices")
        if sl is None or sl[0] is not wc[2] or sl[1] != (C1, C2):
            sl = (wc[2], (C1, C2), R.WeightSlice(wc[2], 0, C1, False), R.WeightSlice(wc[2], C1, C1 + C2, True))
            self.__dict__["_w2_slices"] = sl
        P = R.conv(q1, sl[2])                                     # per point: W2[:, :C1] . GN(relu(q))
        return R.conv(k1, sl[3], stats="relu", pre_add=(P, K))    # relu(W2[:, C1:] . GN(relu(k)) + b2 + P[point])

    def forward_rows(self, query, grouped, grouped_out, K, counts=None):
        """query Rows [B * np][C_in1], grouped Rows [B * np * K][C_in2], grouped_out Rows [B * np * K][C_out] -> Rows
        [B * np][C_out].  Reference :81-95 as 4-5 GEMMs, three in-place normalise passes, one concat-ReLU pass and one
        softmax-weighted reduction over the K rows of a point; counts (B, np): only the first max(1, count) neighbour
        slots of a point are real (ball query) -- the reference masks the others with -1e9 before the softmax."""
        wc = list(self.weight_conv)
        if isinstance(wc[1], MyGroupNorm):
            u = self._u_without_concat(query, grouped, K, wc)
            if u is None:
                tot = R.concat_qk(R.conv(query, self.feat_conv), R.conv(grouped, self.grouped_feat_conv), K)  # wc[0]: ReLU
                R.norm_act(tot, wc[1].group_norm, defer=True)  # (both normalisations are applied by the GEMMs that follow)
                u = R.conv(tot, wc[2], stats="relu")
            R.norm_act(u, wc[4].group_norm, pre_relu=True, defer=True)
            score_conv = wc[5]
        else:
            tot = R.concat_qk(R.conv(query, self.feat_conv), R.conv(grouped, self.grouped_feat_conv), K)  # wc[0]: ReLU
            u = R.conv(tot, wc[1])
            R.norm_act(u, relu=True)
            score_conv = wc[3]
        values = grouped_out
        if self.transform_grouped_feat_out:
            layers = list(self.feat_out_conv)
            gn = next((l.group_norm for l in layers[1:] if isinstance(l, MyGroupNorm)), None)
            values = R.conv(grouped_out, layers[0], stats="raw" if gn is not None else None)
            R.norm_act(values, gn, relu=any(isinstance(l, nn.ReLU) for l in layers[1:]), defer=True)  # applied by attend()
        # the score convolution, the soft-max over the neighbours and the weighted sum of the values: one launch where the rows allow
        # it (fp16 rows, K in 4 .. 32: SLIDE_OP_GEMM_ATTEND, the score map stays on chip), else GEMM -> stored scores -> attend()
        return R.conv_attend(u, score_conv, values, K, counts)

    def forward(self, feat, grouped_feat, grouped_feat_out, count):
        """feat (B, C_in1, np), grouped_feat (B, C_in2, np, K), grouped_feat_out (B, C_out, np, K), count (B, np) or
        'all' -> (B, C_out, np)"""
        if not feat.is_cuda:
            raise RuntimeError("CPU not supported")
        K = grouped_feat.shape[-1]
        out = self.forward_rows(R.from_ncx(feat), R.from_ncx(grouped_feat), R.from_ncx(grouped_feat_out), K,
                                None if isinstance(count, str) else count)
        return R.to_ncx(out)


class GlobalAttentionModule(nn.Module):
    def __init__(self, C, additional_dim=0, attention_bn=True, last_activation=True):
        super().__init__()
        self.key_conv = HipConv1x1(C + additional_dim, C)
        self.query_conv = HipConv1x1(C + additional_dim, C)
        layers = [HipConv1x1(C + additional_dim, C)]
        if last_activation:
            if attention_bn:
                layers.append(MyGroupNorm(min(32, C), C))
            layers.append(nn.ReLU(inplace=True))
        self.value_conv = nn.Sequential(*layers)
        if attention_bn:
            self.weight_conv = nn.Sequential(nn.ReLU(inplace=True), MyGroupNorm(min(32, 2 * C), 2 * C), HipConv1x1(2 * C, C),
                                             nn.ReLU(inplace=True), MyGroupNorm(min(32, C), C), HipConv1x1(C, C))
        else:
            self.weight_conv = nn.Sequential(nn.ReLU(inplace=True), HipConv1x1(2 * C, C), nn.ReLU(inplace=True),
                                             HipConv1x1(C, C))

    def forward(self, feat):
        _, _, N = feat.size()
        key = self.key_conv(feat.unsqueeze(-1)).squeeze(-1)
        query = self.query_conv(feat.unsqueeze(-1)).squeeze(-1)
        value = self.value_conv(feat.unsqueeze(-1)).squeeze(-1)
        key = key.unsqueeze(-2).expand(-1, -1, N, -1)
        query = query.unsqueeze(-1).expand(-1, -1, -1, N)
        score = self.weight_conv(torch.cat([query, key], dim=1).contiguous())
        weight = F.softmax(score, dim=-1)
        return (value.unsqueeze(-1) * weight).sum(dim=-1)
