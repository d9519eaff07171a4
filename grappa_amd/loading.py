"""Where a model comes from: a release tag, an exported `.pth` container, or a training checkpoint
(reference utils/loading_utils.py:7-84, training/export_model.py:48-97).

The reference's containers load unchanged -- the state-dict keys of `GrappaModel` are the reference's:
  * exported model: {'state_dict': <GrappaModel keys>, 'config': {'model_config': ...}, 'split_names': ...}   (`model_from_dict`)
  * training checkpoint (`best-model.ckpt` / `last.ckpt`): {'state_dict': <'model.0.' + GrappaModel keys>} -- the trained module is
    `Sequential(GrappaModel, Energy)` held as `.model`; the hyper-parameters sit in `grappa_config.yaml` of the run directory.
Release files are looked up in a local directory first ($GRAPPA_MODELS_DIR, then `<repository>/models`); only when the file is not
there is the release asset fetched (torch.hub), which needs a network."""
import os
from pathlib import Path
from typing import Dict, Optional, Union

import torch

from .deploy import model_from_dict

RELEASE_URL = "https://github.com/hits-mbm-dev/grappa/releases/download"
# release asset -> (release, the tags that name it); facts of the reference's release history (loading_utils.py:27-32)
RELEASES = {
    "grappa-1.2.1.pth": ("v.1.2.0", ("grappa-1.2", "grappa-1.2.1", "latest")),
    "grappa-1.1.1.pth": ("v.1.1.0", ("grappa-1.1", "grappa-1.1.1")),
    "grappa-1.1.0.pth": ("v.1.1.0", ("grappa-1.1.0",)),
    "grappa-1.1-benchmark.pth": ("v.1.1.0", ("grappa-1.1-benchmark",)),
}


def models_dir(directory: Union[str, Path, None] = None) -> Path:
    if directory is not None:
        return Path(directory)
    env = os.environ.get("GRAPPA_MODELS_DIR")
    return Path(env) if env else Path(__file__).resolve().parent.parent / "models"


def file_of_tag(tag: str, directory: Union[str, Path, None] = None) -> Path:
    """the local file a tag stands for: a release asset's name or alias, or the stem of any `.pth` in the models directory
    (a model exported by the user).  The file need not exist yet for a release tag (it can be fetched)."""
    d = models_dir(directory)
    for fname, (_, tags) in RELEASES.items():
        if tag in tags or tag == fname[:-len(".pth")]:
            return d / fname
    own = d / f"{tag}.pth"
    if own.exists():
        return own
    known = sorted(t for _, tags in RELEASES.values() for t in tags)
    local = sorted(p.stem for p in d.glob("*.pth")) if d.is_dir() else []
    raise ValueError(f"Tag {tag!r} names neither a release ({known}) nor a model file in {d} ({local})")


def _torch_load(path: Union[str, Path], trusted: Optional[bool] = None):
    """tensors and plain containers only (`weights_only=True`): a tampered .pth / .ckpt cannot run code.  Files that hold other pickled
    objects (some of the reference's training checkpoints do) load only for a caller that vouches for them -- `trusted=True`, or
    GRAPPA_TRUST_CHECKPOINTS=1 in the environment -- and the refusal says why."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:       # noqa: BLE001
        if trusted is None:
            trusted = os.environ.get("GRAPPA_TRUST_CHECKPOINTS", "0") not in ("0", "")
        if not trusted:
            raise RuntimeError(f"{path} does not load with weights_only=True ({type(e).__name__}: {e}); if the file is yours, pass trusted=True "
                               "or set GRAPPA_TRUST_CHECKPOINTS=1 to unpickle it in full") from e
        return torch.load(path, map_location="cpu", weights_only=False)


def model_dict_from_tag(tag: str, directory: Union[str, Path, None] = None) -> Dict:
    path = file_of_tag(tag, directory)
    if not path.exists():
        release = RELEASES[path.name][0]
        url = f"{RELEASE_URL}/{release}/{path.name}"
        path.parent.mkdir(parents=True, exist_ok=True)
        try:
            torch.hub.download_url_to_file(url, str(path))          # fetched, then loaded like any other file: tensors only
        except Exception as e:  # noqa: BLE001
            raise FileNotFoundError(f"{path} does not exist and {url} could not be fetched ({type(e).__name__}: {e}); "
                                    f"put the release file into {path.parent} or point GRAPPA_MODELS_DIR at its directory") from e
    return _torch_load(path)


def model_from_tag(tag: str = "latest", directory: Union[str, Path, None] = None):
    return model_from_dict(model_dict_from_tag(tag, directory)).eval()


def strip_training_prefixes(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """'model.0.gnn...' (LitModel.model = Sequential(GrappaModel, Energy)) or '0.gnn...' -> 'gnn...'; keys of other children dropped"""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("model."):
            k = k[len("model."):]
        head, _, rest = k.partition(".")
        if head.isdigit():
            if head != "0":
                continue
            k = rest
        out[k] = v
    return out


def _run_config_of(checkpoint: Path) -> Optional[Dict]:
    """`grappa_config.yaml` of the training run a checkpoint belongs to (<run>/files/checkpoints/x.ckpt, export_model.py:53), or beside it"""
    import yaml
    for cand in (checkpoint.parent.parent.parent / "files" / "grappa_config.yaml", checkpoint.parent.parent / "grappa_config.yaml",
                 checkpoint.parent / "grappa_config.yaml"):
        if cand.exists():
            with open(cand) as f:
                return yaml.safe_load(f)
    return None


def model_dict_from_path(path: Union[str, Path], config: Optional[Dict] = None, trusted: Optional[bool] = None) -> Dict:
    """an exported container as it is; a training checkpoint turned into one (prefixes stripped, `config` given or found in the run
    directory; a bare model config is accepted for `config`)"""
    path = Path(path)
    d = _torch_load(path, trusted)
    if not isinstance(d, dict) or "state_dict" not in d:
        raise ValueError(f"{path} holds neither an exported model nor a training checkpoint (no 'state_dict')")
    sd = d["state_dict"]
    trained = any(k.startswith("model.") or k.partition(".")[0].isdigit() for k in sd)
    if not trained and "config" in d and config is None:
        return d
    cfg = config if config is not None else d.get("config") or _run_config_of(path)
    if cfg is None:
        raise FileNotFoundError(f"{path} is a training checkpoint: its hyper-parameters are needed (config=..., or grappa_config.yaml in the run directory)")
    if "model_config" not in cfg:
        cfg = {"model_config": cfg}
    return {"state_dict": strip_training_prefixes(sd) if trained else sd, "config": cfg, "split_names": d.get("split_names")}


def model_from_path(path: Union[str, Path], config: Optional[Dict] = None, trusted: Optional[bool] = None):
    """trusted=True: the caller vouches for a file that holds pickled objects beyond tensors (see _torch_load)"""
    return model_from_dict(model_dict_from_path(path, config, trusted)).eval()
