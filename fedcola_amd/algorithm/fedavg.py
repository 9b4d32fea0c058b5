"""``FedavgOptimizer`` -- the ``src/algorithm/{alg}.py::{Alg}Optimizer`` plugin point
(/root/reference/src/algorithm/fedavg.py:7-55).  Dormant in the reference (``_get_algorithm`` has no caller;
``_aggregate`` inlines the blend), kept here with the same surface; the live aggregation is fedcola_amd/aggregate.py."""
import torch

from .basealgorithm import BaseOptimizer


class FedavgOptimizer(BaseOptimizer):
    def __init__(self, params, **kwargs):
        self.params = params

    def zero_grad(self, set_to_none=False):
        for _, param in self.params.items():
            if param.grad is not None:
                if set_to_none:
                    param.grad = None
                else:
                    param.grad.detach_()
                    param.grad.zero_()

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for _, param in self.params.items():
            if param.grad is None:
                continue
            param.data.sub_(param.grad.data)
        return self.params if loss is None else loss

    def accumulate(self, mixing_coefficient, local_layers_iterator, check_if=lambda name: "num_batches_tracked" in name):
        for server_param, (name, local_signals) in zip(self.params.values(), local_layers_iterator):
            if check_if(name) or name not in mixing_coefficient:
                continue
            if mixing_coefficient[name] == 0 or local_signals is None:
                local_delta = torch.zeros_like(server_param)
            else:
                local_delta = (server_param - local_signals).mul(mixing_coefficient[name]).data.type(server_param.dtype)
            if server_param.grad is None:
                server_param.grad = local_delta
            else:
                server_param.grad.data.add_(local_delta)
