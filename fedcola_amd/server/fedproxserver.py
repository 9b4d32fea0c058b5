"""``FedproxServer`` (/root/reference/src/server/fedproxserver.py:9-11): FedAvg aggregation, FedProx clients."""
from .fedavgserver import FedavgServer


class FedproxServer(FedavgServer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
