"""Running statistics of the training loop (reference: maskrcnn_benchmark/utils/metric_logger.py:10-135).

`meters.update(loss=..., **loss_dict)` is called every iteration by the reference trainer with DEVICE tensors; calling
`.item()` there (as the reference does, metric_logger.py:31-33) stalls the host behind the whole step.  Here tensors
are parked un-synchronised and only resolved when a statistic is read (every 20 iterations in the trainer), in arrival
order — the statistics are the same numbers."""
import time
from collections import defaultdict, deque
from datetime import datetime

import torch

from .comm import is_main_process


class SmoothedValue(object):
    """window median / mean over the last `window_size` values, and the global average of the series"""

    def __init__(self, window_size=20):
        self.deque = deque(maxlen=window_size)
        self.series = []
        self.total = 0.0
        self.count = 0
        self._pending = []

    def update(self, value):
        if isinstance(value, torch.Tensor):
            self._pending.append(value.detach())
            return
        self._flush()
        self._push(value)

    def _push(self, value):
        self.deque.append(value)
        self.series.append(value)
        self.count += 1
        self.total += value

    def _flush(self):
        if self._pending:
            pending, self._pending = self._pending, []
            for v in torch.stack([p.reshape(()).float() for p in pending]).tolist():    # one device->host copy
                self._push(v)

    @property
    def median(self):
        self._flush()
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        self._flush()
        return torch.tensor(list(self.deque)).mean().item()

    @property
    def global_avg(self):
        self._flush()
        return self.total / self.count


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            assert isinstance(v, (float, int, torch.Tensor)), (k, type(v))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {:.4f} ({:.4f})".format(name, m.median, m.global_avg)
                                   for name, m in self.meters.items())


class TensorboardLogger(MetricLogger):
    """MetricLogger that also writes every scalar to tensorboardX (a third-party package the reference imports
    lazily, metric_logger.py:63-71; absent -> ImportError with the reference's hint)"""

    def __init__(self, log_dir, start_iter=0, delimiter="\t"):
        super(TensorboardLogger, self).__init__(delimiter)
        self.iteration = start_iter
        self.writer = self._get_tensorboard_writer(log_dir)

    @staticmethod
    def _get_tensorboard_writer(log_dir):
        try:
            from tensorboardX import SummaryWriter
        except ImportError:
            raise ImportError("To use tensorboard please install tensorboardX [ pip install tensorflow tensorboardX ].")
        if not is_main_process():
            return None
        stamp = datetime.fromtimestamp(time.time()).strftime("%Y%m%d-%H:%M")
        return SummaryWriter("{}-{}".format(log_dir, stamp))

    def update(self, **kwargs):
        super(TensorboardLogger, self).update(**kwargs)
        if self.writer:
            for k, v in kwargs.items():
                self.writer.add_scalar(k, v.item() if isinstance(v, torch.Tensor) else v, self.iteration)
            self.iteration += 1
