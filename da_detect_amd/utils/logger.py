"""Logger factory of the entry points (reference: maskrcnn_benchmark/utils/logger.py): DEBUG level; rank 0 gets a stdout
handler and — when `save_dir` is given — a file handler `save_dir/filename`; other ranks stay silent."""
import logging
import os
import sys

_FORMAT = "%(asctime)s %(name)s %(levelname)s: %(message)s"


def _attach(logger, handler):
    handler.setLevel(logging.DEBUG)
    handler.setFormatter(logging.Formatter(_FORMAT))
    logger.addHandler(handler)


def setup_logger(name, save_dir, distributed_rank, filename="log.txt"):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    if distributed_rank == 0:
        _attach(logger, logging.StreamHandler(stream=sys.stdout))
        if save_dir:
            _attach(logger, logging.FileHandler(os.path.join(save_dir, filename)))
    return logger
