"""`MoE`: the drop-in wrapper the multiway encoder layer instantiates
(reference models/layer.py:13-102, built at multiway_transformer.py:281/283).

Same constructor arguments, same state-dict keys (`moe.gate.wg.weight`, `moe.vision_embed`,
`moe.text_embed`, `moe.experts.experts.{i}.fc{1,2}.{weight,bias}`), same
`forward(hidden_states, mode=...) -> (output, l_aux, exp_counts)` contract; the work is done
by the HIP kernels behind eve_amd.sharded_moe / eve_amd.experts.
"""
from __future__ import annotations

import typing

import torch

from .experts import Experts
from .sharded_moe import MOELayer, TopKGate


class MoE(torch.nn.Module):

    def __init__(self,
                 hidden_size,
                 expert,
                 num_experts=1,
                 ep_size=1,
                 k=1,
                 capacity_factor=1.,
                 eval_capacity_factor=1.,
                 min_capacity=4,
                 use_residual=False,
                 noisy_gate_policy: typing.Optional[str] = None,
                 drop_tokens: bool = True,
                 use_rts=True,
                 use_tutel: bool = False,
                 enable_expert_tensor_parallelism: bool = False,
                 num_tokens=None):
        super().__init__()
        self.use_residual = use_residual
        self.enable_expert_tensor_parallelism = enable_expert_tensor_parallelism
        assert num_experts % ep_size == 0, \
            f"Number of experts ({num_experts}) should be divisible by expert parallel size ({ep_size})"
        self.ep_size = ep_size
        self.expert_group_name = f"ep_size_{self.ep_size}"
        self.num_experts = num_experts
        self.num_local_experts = num_experts // self.ep_size

        assert noisy_gate_policy is None or noisy_gate_policy in ['None', 'Jitter', 'RSample'], \
            'Unsupported noisy_gate_policy: ' + noisy_gate_policy

        experts = Experts(expert, self.num_local_experts, self.expert_group_name)
        gate = TopKGate(hidden_size, num_experts, k, capacity_factor, eval_capacity_factor, min_capacity,
                        noisy_gate_policy, drop_tokens, use_rts)
        self.moe = MOELayer(gate, experts, self.expert_group_name, self.ep_size, self.num_local_experts,
                            num_tokens=num_tokens)
        if use_residual:
            # Residual-MoE (reference layer.py:72-75): a dense copy of the expert beside the routed ones and a
            # two-way mixing head; EVE never enables it, so it runs as plain torch glue around the HIP layer
            self.mlp = expert
            self.coefficient = torch.nn.Linear(hidden_size, 2)

    def set_expert_parallel_group(self, group) -> None:
        """Process group over which experts are sharded (RCCL all-to-all); defaults to WORLD
        when ep_size equals the world size."""
        self.moe._set_ep_group(group)

    def set_noise_source(self, source: str = "torch", seed=None) -> None:
        """Where the gate's Gumbel noise comes from: torch's device generator (default, the reference's behaviour)
        or a generator state on the device that the gate kernel draws from itself (`eve_amd.DeviceNoise`)."""
        self.moe.set_noise_source(source, seed)

    def forward(self, hidden_states, **kwargs):
        """-> (output, l_aux, exp_counts); `mode='vl'|'vision'|'text'` selects the modality routing."""
        routed = self.moe(hidden_states, **kwargs)
        if not self.use_residual:
            return routed, self.moe.l_aux, self.moe.exp_counts
        # Residual-MoE (reference layer.py:94-101): softmax-weighted blend of the routed and the dense branch
        dense = self.mlp(hidden_states)
        dense = dense[0] if isinstance(dense, tuple) else dense
        mix = torch.softmax(self.coefficient(hidden_states), dim=-1)
        w_routed, w_dense = mix.unbind(dim=-1)
        blended = routed * w_routed.unsqueeze(-1) + dense * w_dense.unsqueeze(-1)
        return blended, self.moe.l_aux, self.moe.exp_counts
