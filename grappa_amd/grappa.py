"""`Grappa`: inference wrapper, `predict(Molecule) -> Parameters` (reference grappa.py:14-57)."""
import os

import torch

from . import constants
from .batch import check_disconnected_graphs
from .model import GrappaModel
from .molecule import Molecule
from .parameters import Parameters


class Grappa:
    def __init__(self, model: GrappaModel, max_element: int = constants.MAX_ELEMENT, device: str = "cuda", reference_water_guard: bool = False) -> None:
        # reference_water_guard=True: `predict` behaves exactly like the reference's on a graph that holds water (it parametrises it: the
        # reference's guard never fires, see batch.check_disconnected_graphs); default: the intended guard, which raises
        self.reference_water_guard = bool(reference_water_guard)
        self.model = model.to(device)
        self.model.eval()
        self.max_element = max_element
        self.device = device
        self.field_of_view = model.field_of_view
        # repeated shapes replay a recorded hipGraph instead of issuing ~450 launches from Python (grappa_amd/capture.py; GRAPPA_PREDICT_GRAPHS=0: never)
        self._graphs = None
        if str(device).startswith("cuda") and os.environ.get("GRAPPA_PREDICT_GRAPHS", "1") not in ("0", ""):
            from .capture import ForwardCache
            self._graphs = ForwardCache(self.model, device)

    @classmethod
    def from_tag(cls, tag: str = "latest", max_element=constants.MAX_ELEMENT, device: str = "cuda", models_dir=None, reference_water_guard: bool = False) -> "Grappa":
        """a released model ('latest', 'grappa-1.2', 'grappa-1.1', ...) or a `.pth` exported into the models directory (grappa.py:26-35)"""
        from .loading import model_from_tag
        return cls(model_from_tag(tag, models_dir), max_element, device, reference_water_guard)

    @classmethod
    def from_file(cls, path, max_element=constants.MAX_ELEMENT, device: str = "cuda", config=None, trusted=None, reference_water_guard: bool = False) -> "Grappa":
        """an exported `.pth` container or a training checkpoint (`best-model.ckpt`) of the reference or of `grappa_amd.trainer`.  trusted=True:
        the caller vouches for a file that holds pickled objects beyond tensors (loading._torch_load; default: tensors only)"""
        from .loading import model_from_path
        return cls(model_from_path(path, config, trusted), max_element, device, reference_water_guard)

    def predict(self, molecule: Molecule) -> Parameters:
        if self.model.training:                  # (eval() walks every sub-module: 1 ms of a 7 ms call when there is nothing to switch)
            self.model.eval()
        g = molecule.to_dgl(max_element=self.max_element, exclude_feats=[])
        # water guard.  NOTE: the reference compares argmax(one-hot) (= Z-1) with {1, 8} and therefore never
        # fires (utils/dgl_utils.py:231-234); the intended check (elements {H, O}) is implemented here.
        check_disconnected_graphs(g, reference_water_guard=self.reference_water_guard)
        if self._graphs is not None:
            done = self._graphs(g)
            if done is not None:
                return Parameters.from_dgl(done)
        g = g.to(self.device)
        with torch.no_grad():
            g = self.model(g)
        g = g.to("cpu")
        return Parameters.from_dgl(g)
