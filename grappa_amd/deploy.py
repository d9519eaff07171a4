"""`model_from_config` / `get_default_model_config`: the production hyper-parameters
(reference models/deploy.py:8-64; values also in SURVEY.md Appendix A)."""
from typing import Dict

from .constants import get_default_statistics
from .model import GrappaModel


def model_from_config(model_config: Dict, param_statistics: Dict = None):
    if param_statistics is None:
        param_statistics = get_default_statistics()
    return GrappaModel(param_statistics=param_statistics, **model_config)


_HEADS = ("bond", "angle", "proper", "improper")


def get_default_model_config():
    """hyper-parameters of the production model (grappa-1.x): 7 attention blocks of width 512 with 16 heads on 256 atom features;
    four writer heads with 3 transformer layers (8 heads, width 512) and a 3-layer symmetriser of width 256 each"""
    cfg = {"graph_node_features": 256, "in_feats": None, "in_feat_dims": {},
           "in_feat_name": ["atomic_number", "partial_charge", "ring_encoding", "degree", "charge_model"]}
    cfg.update(gnn_width=512, gnn_attentional_layers=7, gnn_convolutions=0, gnn_attention_heads=16)
    for part, p in (("attention", 0.3), ("initial", 0.0), ("conv", 0.1), ("final", 0.1)):
        cfg[f"gnn_dropout_{part}"] = p
    cfg["parameter_dropout"] = 0.5
    for head in _HEADS:
        cfg[f"{head}_transformer_depth"], cfg[f"{head}_n_heads"], cfg[f"{head}_transformer_width"] = 3, 8, 512
        cfg[f"{head}_symmetriser_depth"], cfg[f"{head}_symmetriser_width"] = 3, 256
    cfg.update(n_periodicity_proper=6, n_periodicity_improper=3, gated_torsion=True, wrong_symmetry=False, positional_encoding=True,
               layer_norm=True, self_interaction=True, learnable_statistics=False, torsion_cutoff=1e-4)
    return cfg


def model_from_dict(model_dict: Dict):
    """Load the reference's exported `.pth` container {'state_dict','config',...} (utils/loading_utils.py:64-73)."""
    model = model_from_config(model_dict["config"]["model_config"])
    model.load_state_dict(model_dict["state_dict"])
    model.eval()
    return model
