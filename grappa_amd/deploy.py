"""`model_from_config` / `get_default_model_config`: the production hyper-parameters
(reference models/deploy.py:8-64; values also in SURVEY.md Appendix A)."""
from typing import Dict

from .constants import get_default_statistics
from .model import GrappaModel


def model_from_config(model_config: Dict, param_statistics: Dict = None):
    if param_statistics is None:
        param_statistics = get_default_statistics()
    return GrappaModel(param_statistics=param_statistics, **model_config)


def get_default_model_config():
    return {
        "graph_node_features": 256, "in_feats": None,
        "in_feat_name": ["atomic_number", "partial_charge", "ring_encoding", "degree", "charge_model"], "in_feat_dims": {},
        "gnn_width": 512, "gnn_attentional_layers": 7, "gnn_convolutions": 0, "gnn_attention_heads": 16,
        "gnn_dropout_attention": 0.3, "gnn_dropout_initial": 0.0, "gnn_dropout_conv": 0.1, "gnn_dropout_final": 0.1,
        "parameter_dropout": 0.5,
        "bond_transformer_depth": 3, "bond_n_heads": 8, "bond_transformer_width": 512, "bond_symmetriser_depth": 3, "bond_symmetriser_width": 256,
        "angle_transformer_depth": 3, "angle_n_heads": 8, "angle_transformer_width": 512, "angle_symmetriser_depth": 3, "angle_symmetriser_width": 256,
        "proper_transformer_depth": 3, "proper_n_heads": 8, "proper_transformer_width": 512, "proper_symmetriser_depth": 3, "proper_symmetriser_width": 256,
        "improper_transformer_depth": 3, "improper_n_heads": 8, "improper_transformer_width": 512, "improper_symmetriser_depth": 3,
        "improper_symmetriser_width": 256,
        "n_periodicity_proper": 6, "n_periodicity_improper": 3, "gated_torsion": True, "wrong_symmetry": False,
        "positional_encoding": True, "layer_norm": True, "self_interaction": True, "learnable_statistics": False, "torsion_cutoff": 1e-4,
    }


def model_from_dict(model_dict: Dict):
    """Load the reference's exported `.pth` container {'state_dict','config',...} (utils/loading_utils.py:64-73)."""
    model = model_from_config(model_dict["config"]["model_config"])
    model.load_state_dict(model_dict["state_dict"])
    return model
