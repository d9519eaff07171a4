"""`MolwiseLoss`: mean over molecules of per-molecule MSEs (energies centred over conformations, forces,
classical parameters) + L2 on torsion force constants.

Drop-in for the reference's training/loss.py:11-167 (same constructor keywords and call signature
`loss_fn(g, dsnames) -> scalar`), but the per-molecule Python loop over `dgl.unbatch` is replaced by two
segment-reduction kernels (one workgroup per molecule) that also emit the gradient of the loss.
Dummy conformations (`is_dummy`, utils/dgl_utils.py:132-171) are masked instead of deleted.
Deviation: the doubled improper regulariser is reproduced, but molecules WITHOUT impropers
contribute 0 instead of NaN.
"""
from typing import Dict, List

import torch

from . import ops

_LEVELS = [("n2", "k"), ("n2", "eq"), ("n3", "k"), ("n3", "eq"), ("n4", "k"), ("n4_improper", "k")]


class MolwiseLoss(torch.nn.Module):
    def __init__(self, gradient_weight: float = 0.8, energy_weight: float = 1.0, param_weight: float = 1e-3, tuplewise_weight: float = 0,
                 weights: Dict[str, float] = {"n2_k": 1e-3, "n3_k": 1e-2, "n4_k": 1e-4}, skip_params_if_not_present: bool = True,
                 proper_regularisation: float = 0., improper_regularisation: float = 0., param_weights_by_dataset: Dict[str, float] = {}):
        super().__init__()
        self.gradient_weight = gradient_weight
        self.energy_weight = energy_weight
        self.param_weight = param_weight
        self.tuplewise_weight = tuplewise_weight
        self.weights = dict(weights)
        self.skip_params_if_not_present = skip_params_if_not_present
        self.proper_regularisation = proper_regularisation
        self.improper_regularisation = improper_regularisation
        self.param_weights_by_dataset = dict(param_weights_by_dataset)
        self.global_batch_size = None      # data-parallel runs set this to the number of molecules over all ranks
        self.last_per_molecule = None

    def param_weights_of(self, dsnames: List[str], B: int) -> torch.Tensor:
        """(B,) host tensor: the parameter-loss weight of every molecule (param_weight, or its dataset's entry of param_weights_by_dataset)"""
        pw = torch.full((B,), float(self.param_weight), dtype=torch.float32)
        for i, n in enumerate(dsnames or []):
            if i < B and n in self.param_weights_by_dataset:
                pw[i] = float(self.param_weights_by_dataset[n])
        return pw

    def forward(self, g, dsnames: List[str] = None):
        assert not (self.gradient_weight == 0 and self.energy_weight == 0 and self.param_weight == 0), \
            "At least one of the weights must be non-zero."
        assert self.tuplewise_weight == 0., f"Tuplewise loss not implemented yet., but weight is {self.tuplewise_weight}."
        plan = g.plan()
        B = plan.B
        # a batch that ends in a padding molecule (DeviceDataset.collate(pad_to=...)): the loss is over its real molecules only
        B_real = B if getattr(plan, "n_real_mols", None) is None else int(plan.n_real_mols)
        dev = plan.device
        gd, n1 = g.nodes["g"].data, g.nodes["n1"].data
        cfg = {"inv_B": 1.0 / float(self.global_batch_size or B_real), "energy_weight": float(self.energy_weight),
               "gradient_weight": float(self.gradient_weight), "energy_ref": None, "gradient_ref": None, "is_dummy": None}
        energy = gradient = None
        if "is_dummy" in gd:
            cfg["is_dummy"] = gd["is_dummy"].float().contiguous()
        if self.energy_weight != 0.:
            energy = gd["energy"]
            cfg["energy_ref"] = gd["energy_ref"].float().contiguous()
            assert energy.shape == cfg["energy_ref"].shape, f"Shape of energies and energies_ref do not match: {energy.shape} vs {cfg['energy_ref'].shape}"
        if self.gradient_weight != 0.:
            gradient = n1["gradient"]
            cfg["gradient_ref"] = n1["gradient_ref"].float().contiguous()
            assert gradient.shape == cfg["gradient_ref"].shape, f"Shape of gradients and gradients_ref do not match: {gradient.shape} vs {cfg['gradient_ref'].shape}"
        # ---- parameter MSE (+ regularisers)
        pw = torch.full((B,), float(self.param_weight), dtype=torch.float32)
        if dsnames is not None:
            for i, n in enumerate(dsnames):
                if n in self.param_weights_by_dataset:
                    pw[i] = float(self.param_weights_by_dataset[n])
        have_refs = all((name + "_ref") in g.nodes[lvl].data for lvl, name in _LEVELS)
        if bool((pw != 0).any()) and not have_refs and not self.skip_params_if_not_present:
            raise KeyError("reference parameters (k_ref / eq_ref) are missing in the graph")
        use_mse = bool((pw != 0).any()) and have_refs
        # recorded steps (capture.CapturedTrainStep): the per-molecule weights are an INPUT of the graph -- a device tensor on the plan that the
        # trainer fills per batch (`param_weights_of`), not a host tensor uploaded inside the step
        pw_rows = getattr(plan, "param_weight_rows", None)
        if pw_rows is not None:
            use_mse = have_refs and (float(self.param_weight) != 0. or any(float(v) != 0. for v in self.param_weights_by_dataset.values()))
        params = [g.nodes[lvl].data.get(name) for lvl, name in _LEVELS]
        refs, fac, reg, used = [None] * 6, [1.0] * 6, [0.0] * 6, [False] * 6
        for i, (lvl, name) in enumerate(_LEVELS):
            if use_mse and i < 5:                      # improper parameters never enter the MSE (loss.py:91-92)
                refs[i] = g.nodes[lvl].data[name + "_ref"].float().contiguous()
                fac[i] = float(self.weights.get(f"{lvl}_{name}", 1.0))
                used[i] = True
        if self.proper_regularisation > 0.:
            reg[4], used[4] = float(self.proper_regularisation), True
        if self.improper_regularisation > 0.:
            reg[5], used[5] = 2.0 * float(self.improper_regularisation), True      # added twice in the reference (loss.py:128-132)
        cfg.update({"param_active": any(used), "param_used": used, "refs": refs, "fac": fac, "reg": reg,
                    "pw": (pw_rows if pw_rows is not None else pw.to(dev)) if use_mse else None})
        if any(used):
            for i, u in enumerate(used):
                if u and params[i] is None:
                    raise KeyError(f"{_LEVELS[i][0]} has no {_LEVELS[i][1]} attribute: run the model before the loss")
        loss, per_mol = ops.MolwiseLossFn.apply(plan, cfg, energy, gradient, *params)
        self.last_per_molecule = per_mol
        return loss
