"""Loading weights by NAME SUFFIX (reference: maskrcnn_benchmark/utils/model_serialization.py:14-95): a checkpoint
key is assigned to the model key that ends with it, the longest such checkpoint key winning — this is what lets a bare
ImageNet ResNet (`layer1.0.conv1.weight`) initialise `backbone.body.layer1.0.conv1.weight`, and what loads the
released DA checkpoints.  Host code; tensors are assigned, `load_state_dict` then copies them into the (HIP-resident)
parameters."""
import logging
from collections import OrderedDict


def match_keys(model_keys, loaded_keys):
    """-> {model_key: loaded_key} for every model key that has a checkpoint key as suffix (longest suffix wins; among
    equally long ones the first in sorted order, like torch.max over the reference's match matrix)"""
    loaded_sorted = sorted(loaded_keys)
    out = {}
    for key in sorted(model_keys):
        best, best_len = None, 0
        for cand in loaded_sorted:
            if len(cand) > best_len and key.endswith(cand):
                best, best_len = cand, len(cand)
        if best is not None:
            out[key] = best
    return out


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    logger = logging.getLogger(__name__)
    matches = match_keys(model_state_dict.keys(), loaded_state_dict.keys())
    width = max((len(k) for k in model_state_dict), default=1)
    width_loaded = max((len(k) for k in loaded_state_dict), default=1)
    for key in sorted(model_state_dict.keys()):
        if key in matches:
            src = matches[key]
            model_state_dict[key] = loaded_state_dict[src]
            logger.info("{: <{}} loaded from {: <{}} of shape {}".format(key, width, src, width_loaded,
                                                                       tuple(loaded_state_dict[src].shape)))
    for key in sorted(model_state_dict.keys()):
        if key not in matches:
            logger.info(key + " is not loaded.")
    return matches


def strip_prefix_if_present(state_dict, prefix):
    """drops `prefix` (e.g. DistributedDataParallel's "module.") when EVERY key carries it"""
    keys = sorted(state_dict.keys())
    if not all(key.startswith(prefix) for key in keys):
        return state_dict
    return OrderedDict((key.replace(prefix, ""), value) for key, value in state_dict.items())


def load_state_dict(model, loaded_state_dict):
    model_state_dict = model.state_dict()
    loaded_state_dict = strip_prefix_if_present(loaded_state_dict, prefix="module.")
    align_and_update_state_dicts(model_state_dict, loaded_state_dict)
    model.load_state_dict(model_state_dict)      # strict: every model key is present (own value or loaded one)
