"""The FedProx name of the (dormant) server-optimizer plugin point, /root/reference/src/algorithm/fedprox.py:7-9: FedProx changes
the CLIENT objective only (fedcola_amd/client/fedproxclient.py); on the server it is FedAvg, constructor included."""
from .fedavg import FedavgOptimizer


class FedproxOptimizer(FedavgOptimizer):
    pass
