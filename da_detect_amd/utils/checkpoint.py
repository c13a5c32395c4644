"""Checkpoint files of the trainers (reference: maskrcnn_benchmark/utils/checkpoint.py:13-141).

On disk: `<save_dir>/<name>.pth` = {"model", "optimizer", "scheduler", **extra} and a `last_checkpoint` text file
naming the newest one — the reference's layout, so its checkpoints load here and vice versa.  Weights go through the
suffix-matching loader (model_serialization.py); Caffe2 `.pkl` files through the renamer (c2_model_loading.py);
`catalog://NAME` is resolved by `ModelCatalog.get` of the file `cfg.PATHS_CATALOG` names.  There is no network on the
training hosts: a catalog entry (or MODEL.WEIGHT) that is a URL is refused with the name of the file to fetch by hand.

Two deliberate differences from the reference: the live model / optimizer / scheduler state always wins
over same-named entries passed to `save` (its trainer merges the dict returned by `load` — which still holds the LOADED
optimizer / scheduler state, checkpoint.py:62-70 — into the arguments it later saves), and `save` creates `save_dir`.
"""
import logging
import os

import torch

from .c2_model_loading import load_c2_format
from .imports import load_paths_catalog
from .model_serialization import load_state_dict

_TAG = "last_checkpoint"


class Checkpointer(object):
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.save_dir = save_dir
        self.save_to_disk = save_to_disk
        self.logger = logger or logging.getLogger(__name__)

    # ------------------------------------------------------------------------------------------------- writing
    def _live_state(self):
        state = {"model": self.model.state_dict()}
        for key, obj in (("optimizer", self.optimizer), ("scheduler", self.scheduler)):
            if obj is not None:
                state[key] = obj.state_dict()
        return state

    def save(self, name, **kwargs):
        if not (self.save_dir and self.save_to_disk):
            return
        state = self._live_state()
        state.update({k: v for k, v in kwargs.items() if k not in state})
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, name + ".pth")
        self.logger.info("Saving checkpoint to {}".format(path))
        torch.save(state, path)
        self.tag_last_checkpoint(path)

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, _TAG), "w") as f:
            f.write(last_filename)

    # ------------------------------------------------------------------------------------------------- reading
    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, _TAG))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, _TAG)) as f:
                return f.read().strip()
        except IOError:          # deleted by another process in between
            return ""

    def load(self, f=None, load_optimizer=False):
        """restore the model from `f`; returns what else the file holds (`iteration`, and — as in this fork of the
        reference, whose optimizer / scheduler restore is commented out — the stored optimizer / scheduler state).
        load_optimizer=True also restores those two (a true resume) and removes them from the returned dict."""
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from {}".format(f))
        rest = self._load_file(f)
        self._load_model(rest)
        if load_optimizer:
            for key, obj in (("optimizer", self.optimizer), ("scheduler", self.scheduler)):
                if key in rest and obj is not None:
                    self.logger.info("Loading {} from {}".format(key, f))
                    obj.load_state_dict(rest.pop(key))
        return rest

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device("cpu"), weights_only=False)

    def _load_model(self, checkpoint):
        load_state_dict(self.model, checkpoint.pop("model"))
        # the loader copies into the parameters through .data-style writes that need not bump their autograd version: the
        # caches of derived weights (transposed / padded forms) are keyed by (version, weight epoch)
        from .. import _C
        _C.bump_weight_epoch()


class DetectronCheckpointer(Checkpointer):
    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        super(DetectronCheckpointer, self).__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger)
        self.cfg = cfg.clone()

    def _resolve(self, f):
        if f.startswith("catalog://"):
            catalog = load_paths_catalog(self.cfg.PATHS_CATALOG)
            target = catalog.ModelCatalog.get(f[len("catalog://"):])
            self.logger.info("{} points to {}".format(f, target))
            f = target
        if f.startswith("http"):
            raise ValueError("{}: no network on this host — fetch the file by hand and pass its local path as "
                             "MODEL.WEIGHT (or place it where ModelCatalog of PATHS_CATALOG points)".format(f))
        if not os.path.exists(f):
            raise FileNotFoundError("checkpoint file {} does not exist".format(f))
        return f

    def _load_file(self, f):
        f = self._resolve(f)
        if f.endswith(".pkl"):
            return load_c2_format(self.cfg, f)
        loaded = super(DetectronCheckpointer, self)._load_file(f)
        return loaded if "model" in loaded else {"model": loaded}
