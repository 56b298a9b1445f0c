"""Parameter names + shapes of the denoiser (PointNet2CloudCondition) for a given `pointnet_config`, i.e. what the
reference's `net.state_dict()` holds (construction rules:
pointnet2/models/pointnet2_ssg_sem.py:47-177, pointnet2_ops/pointnet2_modules.py:71-118,295-404,666-769,
pointnet2_ops/attention.py:35-67).  Used for synthetic weights and checkpoint validation."""


def _gn_channels(C):
    G = min(32, C)
    return C - C % G


def _mlp(spec, pfx, include_t, t_dim, include_cond, cond_dim, out):
    names = ["first_mlp", "second_mlp"] + (["rest_mlp"] if len(spec) > 3 else [])
    if include_t:
        out += [(pfx + ".fc.weight", (spec[1], t_dim)), (pfx + ".fc.bias", (spec[1],))]
    if include_cond:
        out += [(pfx + ".fc_condition.weight", (spec[2], cond_dim)), (pfx + ".fc_condition.bias", (spec[2],))]
    if spec[0] != spec[-1]:
        out += [(pfx + ".res_connect.weight", (spec[-1], spec[0], 1, 1)), (pfx + ".res_connect.bias", (spec[-1],))]
    ins = [spec[0], spec[1], spec[2]]
    outs = [spec[1], spec[2], spec[-1]]
    for n, i, o in zip(names, ins, outs):
        out += [("%s.%s.0.weight" % (pfx, n), (o, i, 1, 1)), ("%s.%s.0.bias" % (pfx, n), (o,)),
                ("%s.%s.1.group_norm.weight" % (pfx, n), (_gn_channels(o),)),
                ("%s.%s.1.group_norm.bias" % (pfx, n), (_gn_channels(o),))]


def _attention(pfx, c_in1, c_in2, c_out, out):
    C1, C2 = max(c_in1, 32), max(c_in2, 32)
    inter = min(C1 + C2, c_out)
    out += [(pfx + ".feat_conv.weight", (C1, c_in1, 1, 1)), (pfx + ".feat_conv.bias", (C1,)),
            (pfx + ".grouped_feat_conv.weight", (C2, c_in2, 1, 1)), (pfx + ".grouped_feat_conv.bias", (C2,)),
            (pfx + ".weight_conv.1.group_norm.weight", (_gn_channels(C1 + C2),)),
            (pfx + ".weight_conv.1.group_norm.bias", (_gn_channels(C1 + C2),)),
            (pfx + ".weight_conv.2.weight", (inter, C1 + C2, 1, 1)), (pfx + ".weight_conv.2.bias", (inter,)),
            (pfx + ".weight_conv.4.group_norm.weight", (_gn_channels(inter),)),
            (pfx + ".weight_conv.4.group_norm.bias", (_gn_channels(inter),)),
            (pfx + ".weight_conv.5.weight", (c_out, inter, 1, 1)), (pfx + ".weight_conv.5.bias", (c_out,)),
            (pfx + ".feat_out_conv.0.weight", (c_out, c_out, 1, 1)), (pfx + ".feat_out_conv.0.bias", (c_out,)),
            (pfx + ".feat_out_conv.1.group_norm.weight", (_gn_channels(c_out),)),
            (pfx + ".feat_out_conv.1.group_norm.bias", (_gn_channels(c_out),))]


def denoiser_param_spec(hp):
    """-> list of (state_dict_name, shape) for the shipped DDPM configuration family."""
    arch = hp["architecture"]
    t4 = 4 * hp["t_dim"]
    cdim = hp["class_condition_dim"]
    inc_t, inc_c = hp["include_t"], hp["include_class_condition"]
    C0 = hp["in_fea_dim"] + (3 if hp["attach_position_to_input_feature"] else 0)
    extra = 3 * (1 + int(hp["include_abs_coordinate"]) + int(hp.get("include_center_coordinate", False)))
    out = []
    if inc_c:
        out.append(("class_emb.weight", (hp["num_class"], cdim)))
    out += [("fc_t1.weight", (t4, hp["t_dim"])), ("fc_t1.bias", (t4,)), ("fc_t2.weight", (t4, t4)), ("fc_t2.bias", (t4,))]
    f = arch["feature_dim"]
    for i in range(len(arch["npoint"])):
        c_in = C0 if i == 0 else f[i]
        spec = [c_in + extra] + [f[i]] * (arch["mlp_depth"] - 1) + [f[i + 1]]
        pfx = "SA_modules.%d" % i
        _mlp(spec, pfx + ".mlps.0", inc_t, t4, inc_c, cdim, out)
        _attention(pfx + ".attention_modules.0", c_in, spec[0], spec[-1], out)
    d = arch["decoder_feature_dim"]
    for i in range(len(d) - 1):
        skip = C0 if i == 0 else f[i]
        mlp1 = [d[i + 1] + 11] + [d[i]] * arch["decoder_mlp_depth"]
        mlp2 = [d[i] + skip + 3] + [d[i]] * arch["decoder_mlp_depth"]
        pfx = "FP_modules.%d" % i
        _mlp(mlp1, pfx + ".mlp1", False, t4, False, cdim, out)
        _attention(pfx + ".attention_module", skip, mlp1[0], mlp1[-1], out)
        _mlp(mlp2, pfx + ".mlp2", inc_t, t4, inc_c, cdim, out)
    out += [("fc_lyaer.0.weight", (128, d[0] + 3, 1)), ("fc_lyaer.0.bias", (128,)), ("fc_lyaer.1.weight", (128,)),
            ("fc_lyaer.1.bias", (128,)), ("fc_lyaer.3.weight", (hp["out_dim"], 128, 1)), ("fc_lyaer.3.bias", (hp["out_dim"],))]
    return out
