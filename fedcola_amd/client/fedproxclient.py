"""``FedproxClient`` (/root/reference/src/client/fedproxclient.py:13-92): ``FedavgClient`` whose loss carries the proximal term
``mu * 0.5 * sum_over_parameter_tensors ||param - global_param||_2`` (per-tensor norms, not squared -- fedproxclient.py:64-67).

The reference deep-copies the model as the frozen global reference (:33) and walks ``named_parameters()`` in Python every step;
here the global weights are one flat device copy taken when ``update()`` starts and the term is part of the fused HIP step
(``fc_client_step_prox``: norm reduction + gradient add between backward and AdamW.step)."""
from __future__ import annotations

from .fedavgclient import FedavgClient


class FedproxClient(FedavgClient):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def _prox(self):
        return self.model.flat.detach().clone(), float(self.args.mu)
