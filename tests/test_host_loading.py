"""CPU: where a model comes from (grappa_amd/loading.py; reference utils/loading_utils.py:7-84, training/export_model.py:48-97): the
exported container, a training checkpoint with the trained module's key prefixes and its run directory's config, tags resolved in a
local models directory, and the failure modes (unknown tag, no network, a checkpoint without hyper-parameters)."""
import os

import pytest
import torch
import yaml

from test_host_train import TINY


def _model():
    from grappa_amd import GrappaModel
    torch.manual_seed(3)
    return GrappaModel(**TINY)


def _same(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    return list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_exported_container_and_tags_in_a_local_directory(tmp_path, monkeypatch):
    from grappa_amd import Grappa, loading, model_from_path, model_from_tag
    m = _model()
    container = {"state_dict": m.state_dict(), "config": {"model_config": dict(TINY), "trainer": {"max_epochs": 3}}, "split_names": {"train": ["a"]}}
    torch.save(container, tmp_path / "my-model.pth")
    torch.save(container, tmp_path / "grappa-1.2.1.pth")
    got = model_from_path(tmp_path / "my-model.pth")
    assert _same(got, m) and not got.training                                     # (the reference returns the model in eval mode)
    assert loading.model_dict_from_path(tmp_path / "my-model.pth")["split_names"] == {"train": ["a"]}
    assert _same(model_from_tag("my-model", tmp_path), m)                          # a user's export, by the stem of its file
    for tag in ("latest", "grappa-1.2", "grappa-1.2.1"):                           # release aliases -> the release file, found locally
        assert loading.file_of_tag(tag, tmp_path) == tmp_path / "grappa-1.2.1.pth"
        assert _same(model_from_tag(tag, tmp_path), m)
    monkeypatch.setenv("GRAPPA_MODELS_DIR", str(tmp_path))
    assert loading.models_dir() == tmp_path and _same(model_from_tag("latest"), m)
    g = Grappa.from_tag("my-model", device="cpu")
    assert _same(g.model, m) and g.field_of_view == m.field_of_view
    assert _same(Grappa.from_file(tmp_path / "my-model.pth", device="cpu").model, m)
    with pytest.raises(ValueError, match="names neither a release"):
        model_from_tag("grappa-0.0", tmp_path)
    # a release that is not in the directory is fetched -- here there is no network: the error says where the file is expected
    def no_network(*a, **k):
        raise OSError("no route to host")
    monkeypatch.setattr(torch.hub, "load_state_dict_from_url", no_network)
    with pytest.raises(FileNotFoundError, match="grappa-1.1.1.pth"):
        model_from_tag("grappa-1.1", tmp_path)


def test_training_checkpoint_with_the_trained_modules_prefixes(tmp_path):
    from grappa_amd import Energy, loading, model_from_path
    m = _model()
    trained = torch.nn.Sequential(m, Energy())                                     # what the reference trains (export_model.py:70-74)
    lit_keys = {"model." + k: v for k, v in trained.state_dict().items()}
    assert all(k.startswith("model.0.") for k in lit_keys)
    run = tmp_path / "run-1"
    (run / "files" / "checkpoints").mkdir(parents=True)
    ckpt = run / "files" / "checkpoints" / "best-model.ckpt"
    torch.save({"state_dict": lit_keys, "epoch": 7, "optimizer_states": []}, ckpt)
    with pytest.raises(FileNotFoundError, match="hyper-parameters"):
        model_from_path(ckpt)
    with open(run / "files" / "grappa_config.yaml", "w") as f:
        yaml.safe_dump({"model_config": dict(TINY), "data_config": {"datasets": ["x"]}}, f)
    assert _same(model_from_path(ckpt), m)                                        # config found in the run directory
    assert _same(model_from_path(ckpt, config=dict(TINY)), m)                     # or given (a bare model config is accepted)
    assert set(loading.strip_training_prefixes({"model.0.a.b": 1, "model.1.c": 2, "0.d": 3, "e.f": 4})) == {"a.b", "d", "e.f"}
    torch.save({"weights": 1}, tmp_path / "junk.pth")
    with pytest.raises(ValueError, match="no 'state_dict'"):
        model_from_path(tmp_path / "junk.pth")


def test_trainer_export_loads_back(tmp_path):
    """`Trainer.model_dict()` writes the reference's container: `model_from_path` reads it"""
    from grappa_amd import model_from_path
    m = _model()
    torch.save({"state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "config": {"model_config": m.model_config}}, tmp_path / "t.pth")
    assert _same(model_from_path(tmp_path / "t.pth"), m)
