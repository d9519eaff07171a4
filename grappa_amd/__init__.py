"""grappa_amd: MI355X-native engine for the Grappa hot path (GrappaModel forward / predict, Energy,
MolwiseLoss train step) behind the reference's Python API.  See DESIGN.md."""
from .batch import MolBatch, batch, unbatch, set_number_confs, delete_dummy_confs
from .molecule import Molecule
from .parameters import Parameters
from .model import GrappaModel
from .energy import Energy
from .loss import MolwiseLoss
from .deploy import get_default_model_config, model_from_config, model_from_dict
from .grappa import Grappa
from .loading import model_from_tag, model_from_path
from .evaluation import FastEvaluator
from .schedule import TrainSchedule
from .device_dataset import DeviceDataset
from .dataset import Dataset
from .moldata import MolData
from .trainer import Trainer

__all__ = ["MolBatch", "batch", "unbatch", "set_number_confs", "delete_dummy_confs", "Molecule", "Parameters", "GrappaModel",
           "Energy", "MolwiseLoss", "get_default_model_config", "model_from_config", "model_from_dict", "model_from_tag", "model_from_path", "Grappa", "FastEvaluator",
           "TrainSchedule", "DeviceDataset", "Dataset", "MolData", "Trainer"]
