"""Package-wide constants of the Grappa hot path.

Mirrors the facts (not the code) of the reference's src/grappa/constants.py:36-46 and
utils/graph_utils.py:233-242.  Units: Angstrom, radian, kcal/mol.
"""
import torch

IMPROPER_CENTRAL_IDX = 2      # reference constants.py:36
MAX_ELEMENT = 53              # reference constants.py:38 (covers iodine)
N_PERIODICITY_PROPER = 6      # reference constants.py:41
N_PERIODICITY_IMPROPER = 6    # reference constants.py:42
CHARGE_MODELS = ["am1BCC", "amber99"]   # reference constants.py:44
CHARGE_ENCODING_DIM = 16      # reference models/graph_attention.py:49
TUPLE_LEVELS = ["n2", "n3", "n4", "n4_improper"]
LEVEL_ARITY = {"n2": 2, "n3": 3, "n4": 4, "n4_improper": 4}
BONDED_CONTRIBUTIONS = [("n2", "k"), ("n2", "eq"), ("n3", "k"), ("n3", "eq"), ("n4", "k"), ("n4_improper", "k")]

# default input-feature widths, reference models/graph_attention.py:60-70
DEFAULT_FEAT_DIMS = {
    "atomic_number": MAX_ELEMENT,
    "ring_encoding": 7,
    "partial_charge": 1,
    "sp_hybridization": 6,
    "mass": 2,
    "degree": 6,
    "is_radical": 1,
    "laplacian_positional_encoding": 5,
    "charge_model": len(CHARGE_MODELS),
}

# reference constants.py:50-104 (standard atomic weights, truncated as there)
ATOMIC_MASSES = {
    1: 1.008, 2: 4.002, 3: 6.94, 4: 9.012, 5: 10.81, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.998, 10: 20.1797,
    11: 22.989, 12: 24.305, 13: 26.981, 14: 28.085, 15: 30.973, 16: 32.06, 17: 35.45, 18: 39.95, 19: 39.0983,
    20: 40.078, 21: 44.955, 22: 47.867, 23: 50.9415, 24: 51.9961, 25: 54.938, 26: 55.845, 27: 58.933,
    28: 58.6934, 29: 63.546, 30: 65.38, 31: 69.723, 32: 72.63, 33: 74.921, 34: 78.971, 35: 79.904, 36: 83.798,
    37: 85.4678, 38: 87.62, 39: 88.905, 40: 91.224, 41: 92.906, 42: 95.95, 43: 97.0, 44: 101.07, 45: 102.905,
    46: 106.42, 47: 107.8682, 48: 112.414, 49: 114.818, 50: 118.71, 51: 121.76, 52: 127.6, 53: 126.904,
}


def get_default_statistics():
    """Output-scaling statistics used when no dataset statistics are given
    (values: reference utils/graph_utils.py:233-242)."""
    return {
        "mean": {
            "n2_k": torch.tensor([763.2819]), "n2_eq": torch.tensor([1.2353]),
            "n3_k": torch.tensor([105.6576]), "n3_eq": torch.tensor([1.9750]),
            "n4_k": torch.tensor([1.5617e-01, -5.8312e-01, 7.0820e-02, -6.3840e-04, 4.7139e-04, -4.1655e-04]),
            "n4_improper_k": torch.tensor([0.0000, -2.3933, 0.0000]),
        },
        "std": {
            "n2_k": torch.tensor([161.2278]), "n2_eq": torch.tensor([0.1953]),
            "n3_k": torch.tensor([26.5965]), "n3_eq": torch.tensor([0.0917]),
            "n4_k": torch.tensor([0.4977, 1.2465, 0.1466, 0.0192, 0.0075, 0.0066]),
            "n4_improper_k": torch.tensor([0.0000, 4.0571, 0.0000]),
        },
    }
