"""``FedavgOptimizer`` -- the ``src/algorithm/{alg}.py::{Alg}Optimizer`` plugin point of the reference
(/root/reference/src/algorithm/fedavg.py:7-55; looked up by ``FedavgServer._get_algorithm``, fedavgserver.py:241-246).

The plugin is dormant in the reference (``_get_algorithm`` has no caller; ``_aggregate`` inlines its own blend, which is what
fedcola_amd/aggregate.py accelerates).  Its contract is a server-side pseudo-gradient step over the parameter mapping it is
given, here the state_dict views of a model's flat buffer:

    accumulate(c, client_layers):  pending[k] += c[k] * (theta_server[k] - theta_client[k])      once per sampled client
    step():                        theta_server[k] -= pending[k]
    zero_grad():                   pending <- 0 (or dropped)

With c[k] = n_i / sum n over the round's clients that is FedAvg.  The accumulator lives beside the parameters (the reference
parks it in ``param.grad``); ``params`` is walked positionally against the client's layers, names coming from the client side, as
the reference does."""
from typing import Dict, Iterable, Mapping, Optional, Tuple

import torch

from .basealgorithm import BaseOptimizer


class FedavgOptimizer(BaseOptimizer):
    def __init__(self, params: Mapping[str, torch.Tensor], **kwargs):
        self.params = params
        self._pending: Dict[int, torch.Tensor] = {}          # position in ``params`` -> accumulated pseudo-gradient

    def zero_grad(self, set_to_none: bool = False):
        if set_to_none:
            self._pending.clear()
        else:
            for buf in self._pending.values():
                buf.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            closure()
        for pos, tensor in enumerate(self.params.values()):
            buf = self._pending.get(pos)
            if buf is not None:
                tensor.sub_(buf)
        return self.params

    @torch.no_grad()
    def accumulate(self, mixing_coefficient: Mapping[str, float], local_layers_iterator: Iterable[Tuple[str, Optional[torch.Tensor]]],
                   check_if=lambda name: "num_batches_tracked" in name):
        for pos, (server, (name, local)) in enumerate(zip(self.params.values(), local_layers_iterator)):
            if check_if(name) or name not in mixing_coefficient:
                continue
            c = mixing_coefficient[name]
            buf = self._pending.get(pos)
            if buf is None:
                buf = self._pending[pos] = torch.zeros_like(server)
            if c != 0 and local is not None:
                buf.add_(((server - local.to(server.device)) * c).to(server.dtype))
