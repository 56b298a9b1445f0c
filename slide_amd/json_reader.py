"""Reader for the reference's experiment JSON files (pointnet2/data_utils/json_reader.py:16-32): lists are stored as
strings ("[16, 16]") and restored recursively -- with ast.literal_eval instead of the reference's eval()."""
import ast
import json


def restore_string_to_list_in_a_dict(d):
    for k in list(d.keys()):
        v = d[k]
        if isinstance(v, str):
            try:
                ev = ast.literal_eval(v)
                if isinstance(ev, list):
                    d[k] = ev
            except (ValueError, SyntaxError):
                pass
        elif isinstance(v, dict):
            d[k] = restore_string_to_list_in_a_dict(v)
    return d


def read_json_file(config_file):
    with open(config_file) as f:
        return restore_string_to_list_in_a_dict(json.load(f))


def autoencoder_read_config(config_dir, config):
    """encoder config + the list of decoder-level configs an autoencoder config points to
    (pointnet2/data_utils/json_reader.py:35-45)"""
    import os
    pc = config["pointnet_config"]
    enc = read_json_file(os.path.join(config_dir, pc["encoder_config_file"]))["pointnet_config"]
    decs = [read_json_file(os.path.join(config_dir, f))["pointnet_config"] for f in pc["decoder_config_file"]]
    return enc, decs
