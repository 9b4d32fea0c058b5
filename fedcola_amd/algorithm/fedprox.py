"""``FedproxOptimizer`` (/root/reference/src/algorithm/fedprox.py:7-9): the FedAvg server optimizer under the FedProx name."""
from .fedavg import FedavgOptimizer


class FedproxOptimizer(FedavgOptimizer):
    def __init__(self, params, **kwargs):
        super().__init__(params=params, **kwargs)
