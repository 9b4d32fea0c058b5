"""Server-side optimizer ABC (/root/reference/src/algorithm/basealgorithm.py:5-14)."""
from abc import ABCMeta, abstractmethod


class BaseOptimizer(metaclass=ABCMeta):
    @abstractmethod
    def step(self, closure=None):
        raise NotImplementedError

    @abstractmethod
    def accumulate(self, **kwargs):
        raise NotImplementedError
