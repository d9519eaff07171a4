"""`FastEvaluator`: per-dataset RMSE of energies and forces over batched graphs.

Drop-in for the reference's training/evaluation.py:16-159 (same constructor keywords, `step(g, dsnames)`,
`pool() -> {dsname: {'rmse_energies', 'rmse_gradients', 'crmse_gradients'}, 'avg': {...}}`), but `step` is ONE
kernel launch per batch (`grappa_eval_se_f32`, one workgroup per molecule: centred-energy and force squared errors with
dummy conformations masked) plus one device-side `index_add_` into per-dataset accumulators, instead of
`dgl.unbatch` and a Python loop of ~10 tiny kernels per molecule.  Nothing is copied to the host before `pool()`.
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from .backend import get_backend


class FastEvaluator:
    def __init__(self, log_parameters: bool = False, log_classical_values: bool = False, metric_names: Optional[List[str]] = None,
                 gradients: bool = True):
        if log_parameters:
            raise NotImplementedError("Logging of parameters is not supported anymore.")      # evaluation.py:33-34
        self.log_classical_values = log_classical_values
        self.metric_names = metric_names
        self.gradients = gradients
        self.init_storage()

    def init_storage(self):
        self._ds_index: Dict[str, int] = {}
        self._acc: Optional[torch.Tensor] = None          # (n_datasets, 8) float64 on the graphs' device: se_E, n_E, se_G, n_G, then the
        #                                                   same four for the classical force field vs the PREDICTION (evaluation.py:80-87)

    def register(self, dsnames: List[str], device) -> None:
        """fix the row of every dataset name up front (data parallel validation: every rank must use the same rows, whatever share of
        the batches it sees, so that `all_reduce()` adds like to like)"""
        self._index_of(list(dsnames), device)

    def all_reduce(self) -> None:
        """sum the accumulators over the ranks of the default process group (each rank has stepped through ITS share of the batches)"""
        import torch.distributed as tdist
        if self._acc is not None and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            tdist.all_reduce(self._acc, op=tdist.ReduceOp.SUM)

    def _index_of(self, dsnames: List[str], device) -> torch.Tensor:
        for n in dsnames:
            if n not in self._ds_index:
                self._ds_index[n] = len(self._ds_index)
        need = len(self._ds_index)
        if self._acc is None:
            self._acc = torch.zeros((max(need, 8), 8), dtype=torch.float64, device=device)
        elif self._acc.shape[0] < need:
            grown = torch.zeros((2 * need, 8), dtype=torch.float64, device=device)
            grown[: self._acc.shape[0]] = self._acc
            self._acc = grown
        return torch.tensor([self._ds_index[n] for n in dsnames], dtype=torch.int64).to(device, non_blocking=True)

    @torch.no_grad()
    def step(self, g, dsnames: List[str]):
        plan = g.plan()
        # a batch padded to a fixed shape (DeviceDataset.collate(pad_to=...)) ends in a padding molecule that is no molecule of any dataset
        nB = plan.B if getattr(plan, "n_real_mols", None) is None else int(plan.n_real_mols)
        assert len(dsnames) == nB, f"one dataset name per molecule: {len(dsnames)} names for {nB} molecules"
        gd, n1 = g.nodes["g"].data, g.nodes["n1"].data
        energy, energy_ref = gd["energy"].detach().float().contiguous(), gd["energy_ref"].detach().float().contiguous()
        assert energy.dim() == 2 and energy.shape[1] > 0, f"energies must be a tensor of shape (n_mols, n_confs) but is {tuple(energy.shape)}"
        assert energy.shape == energy_ref.shape, f"energies and energies_ref must have the same shape but are {energy.shape} and {energy_ref.shape}"
        grad = grad_ref = None
        if self.gradients:
            grad, grad_ref = n1["gradient"].detach().float().contiguous(), n1["gradient_ref"].detach().float().contiguous()
            assert grad.dim() == 3, f"gradients must be a tensor of shape (n_atoms,n_confs, 3) but is {tuple(grad.shape)}"
            assert grad.shape == grad_ref.shape, f"gradients and gradients_ref must have the same shape but are {grad.shape} and {grad_ref.shape}"
        is_dummy = gd["is_dummy"].float().contiguous() if "is_dummy" in gd else None
        out = torch.zeros((plan.B, 8), dtype=torch.float32, device=energy.device)
        be = get_backend()
        first = torch.zeros((plan.B, 4), dtype=torch.float32, device=energy.device)
        be.eval_se(plan, energy, energy_ref, is_dummy, grad, grad_ref, first)
        out[:, :4] = first
        if self.log_classical_values:
            e_cl = gd["energy_classical_ff"].detach().float().contiguous()
            g_cl = n1["gradient_classical_ff"].detach().float().contiguous() if self.gradients else None
            be.eval_se(plan, e_cl, energy, is_dummy, g_cl, grad, first)
            out[:, 4:] = first
        idx = self._index_of(list(dsnames), energy.device)
        self._acc.index_add_(0, idx, out[:nB].double())

    def pool(self):
        """per-dataset metrics (energies: per conformation; gradients: per 3-vector; crmse: per component) and their unweighted
        average over datasets; resets the storage (evaluation.py:115-159)."""
        metrics: Dict[str, Dict[str, Optional[float]]] = {}
        acc = self._acc.cpu().numpy() if self._acc is not None else np.zeros((0, 8))
        for dsname, i in self._ds_index.items():
            se_e, n_e, se_g, n_g, cse_e, _, cse_g, _ = (float(x) for x in acc[i])
            m = {"rmse_energies": float(np.sqrt(np.float32(se_e) / np.float32(n_e))),
                 "rmse_gradients": float(np.sqrt(np.float32(se_g) / np.float32(n_g))) if self.gradients else None,
                 "crmse_gradients": float(np.sqrt(np.float32(se_g) / np.float32(n_g) / np.float32(3.0))) if self.gradients else None}
            if self.log_classical_values:
                m["rmse_classical_gradients"] = float(np.sqrt(np.float32(cse_g) / np.float32(n_g))) if self.gradients else None
                m["rmse_classical_energies"] = float(np.sqrt(np.float32(cse_e) / np.float32(n_e)))
            if self.metric_names is not None:
                m = {k: v for k, v in m.items() if k in self.metric_names}
            metrics[dsname] = m
        metrics["avg"] = {}
        for key in ["rmse_energies", "rmse_gradients"]:
            if self.metric_names is not None and key not in self.metric_names:
                continue
            mlist = [metrics[d][key] for d in metrics if d not in ("avg", "all") and metrics[d][key] is not None]
            metrics["avg"][key] = None if len(mlist) == 0 else np.mean(mlist)
        self.init_storage()
        return metrics


def early_stopping_loss(metrics, energy_weight: float = 2.0) -> float:
    """the reference's model-selection criterion (training/lightning_model.py:257-262): energy_weight * <rmse_E> + <rmse_F>,
    each averaged over datasets with equal weight."""
    return float(energy_weight * metrics["avg"]["rmse_energies"] + metrics["avg"]["rmse_gradients"])
