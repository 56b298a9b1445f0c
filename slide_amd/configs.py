"""The two latent-DDPM network configurations BASELINE.json names, as plain dict builders.

Same keys / values as the `pointnet_config` section of the reference's shipped JSON configs
(pointnet2/configs/shapenet_psr_configs/ddpm_keypoint_training_configs/config_standard_attention_batchsize_32_s3_
ema_model_keypoint_airplane_02691156.json and .../latent_ddpm_training_configs/config_latent_ddpm_s3_dim_16_32_ae_kp_
noise_0.04_keypoint_conditional_chair_ae_trained_on_chair.json; all five categories share them) with the
string-encoded lists already restored (pointnet2/data_utils/json_reader.py:16-26).  User configs in the
reference's JSON format are read by slide_amd.json_reader.read_json_file.
"""
import copy

_ATT = {"use_attention_module": True, "attention_bn": True, "transform_grouped_feat_out": True,
        "last_activation": True, "add_attention_to_FeatureMapper_module": True}


def _pointnet_config(model_name, in_fea_dim, out_dim, feature_dim, decoder_feature_dim):
    return {
        "model_name": model_name, "in_fea_dim": in_fea_dim, "out_dim": out_dim, "include_t": True, "t_dim": 128,
        "model.use_xyz": True, "attach_position_to_input_feature": True, "include_abs_coordinate": True,
        "include_center_coordinate": True, "record_neighbor_stats": False, "bn_first": False, "bias": True,
        "res_connect": True, "include_class_condition": True, "num_class": 13, "class_condition_dim": 128, "bn": True,
        "include_local_feature": False, "include_global_feature": False, "global_feature_remove_last_activation": False,
        "pnet_global_feature_architecture": [[4, 128, 256], [512, 1024]], "attention_setting": copy.deepcopy(_ATT),
        "architecture": {"npoint": [16, 16], "radius": [0, 0], "neighbor_definition": "nn", "nsample": [16, 16],
                         "feature_dim": list(feature_dim), "mlp_depth": 3, "decoder_feature_dim": list(decoder_feature_dim),
                         "include_grouper": False, "decoder_mlp_depth": 2, "use_knn_FP": True, "K": 8},
        "condition_net_architecture": None, "feature_mapper_architecture": None,
    }


def position_ddpm_config():
    """16 sparse latent points, xyz only (BASELINE configs 1, 2, 4a)."""
    return {"pointnet_config": _pointnet_config("shapenet_psr_keypoint_generation_batchsize_32_with_ema_airplane", 0, 3,
                                                [32, 64, 128], [64, 64, 128]),
            "diffusion_config": {"T": 1000, "beta_0": 0.0001, "beta_T": 0.02}}


def feature_ddpm_config():
    """48-dim (16+32) feature per latent point, conditioned on the positions (BASELINE configs 3, 4b)."""
    return {"pointnet_config": _pointnet_config(
                "shapenet_psr_latent_ddpm_ae_kp_noise_0.04_keypoint_conditional_latent_dim_16_32_chair_ae_trained_on_chair",
                48, 51, [128, 256, 512], [128, 256, 512]),
            "standard_diffusion_config": {"beta_schedule": "linear", "num_diffusion_timesteps": 1000, "beta_start": 0.0001,
                                          "beta_end": 0.02, "data_clamp_range": -1, "model_var_type": "fixedsmall",
                                          "model_output_scale_factor": 1.0, "loss_type": None,
                                          "keypoint_position_loss_weight": 0.0, "feature_loss_weight": 1.0,
                                          "keypoint_conditional": True}}


# label ids = sorted ShapeNet synset ids (pointnet2/shapenet_psr_dataloader/shapenet_psr_dataset.py:59-66)
CATEGORY_IDS = ["02691156", "02828884", "02933112", "02958343", "03001627", "03211117", "03636649", "03691459",
                "04090263", "04256520", "04379243", "04401088", "04530566"]
CATEGORY_NAMES = ["airplane", "bench", "cabinet", "car", "chair", "display", "lamp", "loudspeaker", "rifle", "sofa",
                  "table", "telephone", "vessel"]
