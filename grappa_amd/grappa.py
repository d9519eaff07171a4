"""`Grappa`: inference wrapper, `predict(Molecule) -> Parameters` (reference grappa.py:14-57)."""
import torch

from . import constants
from .batch import check_disconnected_graphs
from .model import GrappaModel
from .molecule import Molecule
from .parameters import Parameters


class Grappa:
    def __init__(self, model: GrappaModel, max_element: int = constants.MAX_ELEMENT, device: str = "cuda") -> None:
        self.model = model.to(device)
        self.model.eval()
        self.max_element = max_element
        self.device = device
        self.field_of_view = model.field_of_view

    def predict(self, molecule: Molecule) -> Parameters:
        self.model.eval()
        g = molecule.to_dgl(max_element=self.max_element, exclude_feats=[])
        # water guard.  NOTE: the reference compares argmax(one-hot) (= Z-1) with {1, 8} and therefore never
        # fires (utils/dgl_utils.py:231-234); the intended check (elements {H, O}) is implemented here.
        check_disconnected_graphs(g)
        g = g.to(self.device)
        with torch.no_grad():
            g = self.model(g)
        g = g.to("cpu")
        return Parameters.from_dgl(g)
