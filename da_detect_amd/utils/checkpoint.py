"""Checkpoint save / load (reference: maskrcnn_benchmark/utils/checkpoint.py:13-141): `<name>.pth` with model /
optimizer / scheduler state + a `last_checkpoint` tag file; loading goes through the suffix-matching loader, `.pkl`
files through the Caffe2 renamer.  `catalog://` and http(s) sources need the reference's path catalog / network and
are rejected with a clear message (give a local file)."""
import logging
import os

import torch

from .c2_model_loading import load_c2_format
from .model_serialization import load_state_dict


class Checkpointer(object):
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger if logger is not None else logging.getLogger(__name__)

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        data.update(kwargs)
        save_file = os.path.join(self.save_dir, "{}.pth".format(name))
        self.logger.info("Saving checkpoint to {}".format(save_file))
        torch.save(data, save_file)
        self.tag_last_checkpoint(save_file)

    def load(self, f=None):
        """returns the checkpoint's remaining entries (e.g. `iteration`); like the reference it restores the MODEL
        only (checkpoint.py:57-66 leaves optimizer / scheduler state in the returned dict)"""
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from {}".format(f))
        checkpoint = self._load_file(f)
        self._load_model(checkpoint)
        return checkpoint

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint"), "r") as f:
                return f.read().strip()
        except IOError:
            return ""

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(last_filename)

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device("cpu"), weights_only=False)

    def _load_model(self, checkpoint):
        load_state_dict(self.model, checkpoint.pop("model"))


class DetectronCheckpointer(Checkpointer):
    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        super(DetectronCheckpointer, self).__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger)
        self.cfg = cfg.clone()

    def _load_file(self, f):
        if f.startswith("catalog://") or f.startswith("http"):
            raise ValueError("{}: catalog / URL weights need the reference's path catalog and network access; "
                             "download the file and pass its local path (MODEL.WEIGHT)".format(f))
        if f.endswith(".pkl"):
            return load_c2_format(self.cfg, f)
        loaded = super(DetectronCheckpointer, self)._load_file(f)
        if "model" not in loaded:
            loaded = dict(model=loaded)
        return loaded
