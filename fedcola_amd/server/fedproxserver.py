"""``--algorithm fedprox`` resolves the server class by name (main.py:37): /root/reference/src/server/fedproxserver.py:9-11 is
FedavgServer under another name -- same constructor, aggregation and evaluation; only the clients it builds differ
(``fedproxclient.FedproxClient`` through the plugin lookup of ``_create_clients``)."""
from .fedavgserver import FedavgServer


class FedproxServer(FedavgServer):
    pass
