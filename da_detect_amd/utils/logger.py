"""Logger factory of the entry points (reference: maskrcnn_benchmark/utils/logger.py:7-25): DEBUG level, one stdout
handler and — when `save_dir` is given — a file handler `save_dir/filename`, on rank 0 only."""
import logging
import os
import sys

_FORMAT = "%(asctime)s %(name)s %(levelname)s: %(message)s"


def setup_logger(name, save_dir, distributed_rank, filename="log.txt"):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    if distributed_rank > 0:        # worker ranks stay silent
        return logger
    handlers = [logging.StreamHandler(stream=sys.stdout)]
    if save_dir:
        handlers.append(logging.FileHandler(os.path.join(save_dir, filename)))
    for h in handlers:
        h.setLevel(logging.DEBUG)
        h.setFormatter(logging.Formatter(_FORMAT))
        logger.addHandler(h)
    return logger
