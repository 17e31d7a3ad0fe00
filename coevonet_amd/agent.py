"""Host-side mirror of the reference's ``Agent`` (agent.py:7-86) and ``MPEAgent`` (MPE/mpe_agent.py:9-50)."""
from __future__ import annotations

import numpy as np
import torch

from .fcnetwork import FCNetwork


class Agent:
    def __init__(self, model, optimizer, args):
        self.model = model
        self.optimizer = optimizer
        self.precision = args.precision

    def mutate(self, noise_std):
        """GA mutation: every parameter (LayerNorm affine included) += N(0, noise_std), drawn with the global torch
        generator in parameters() order, exactly as agent.py:25-29 does (host_reference RNG mode)."""
        for param in self.model.parameters():
            noise = torch.normal(0, noise_std, size=param.size())
            param.data += noise

    def mutate_ES(self, args, role, step, weights_logging_agent_0, weights_logging_agent_1, weights_logging_adversary):
        """ES perturbation of the Linear weights/biases with the global numpy generator (agent.py:31-70)."""
        mutation_power = {"agent_0": args.mutation_power_agent_0, "agent_1": args.mutation_power_agent_1,
                          "adversary_0": args.mutation_power_adversary}[role]
        weights = self.model.get_perturbable_weights()
        noise = np.random.normal(loc=0.0, scale=mutation_power, size=len(weights))
        self.model.set_perturbable_weights(weights + noise, args)
        self.log_weight_statistics(step=step, weights_logging_agent_0=weights_logging_agent_0,
                                   weights_logging_agent_1=weights_logging_agent_1,
                                   weights_logging_adversary=weights_logging_adversary, role=role)
        return noise

    def set_weights(self, weights):
        self.model.load_state_dict(weights)

    def get_weights(self):
        return self.model.state_dict()

    def clone(self, args):
        raise NotImplementedError("The clone method should be implemented by the specific agent type.")


class MPEAgent(Agent):
    def __init__(self, env, args, role):
        self.input_channels = env.observation_space(role).shape[-1]
        self.n_actions = env.action_space(role).n
        self.model = FCNetwork(self.input_channels, self.n_actions, args.precision)
        # the reference builds an Adam optimizer here (MPE/mpe_agent.py:20) that nothing ever steps; it draws no
        # random numbers, so leaving it out changes no result and saves ~15 ms per agent
        self.optimizer = None
        super().__init__(self.model, self.optimizer, args)

    def clone(self, env, args, role):
        clone = MPEAgent(env, args, role)  # consumes the torch generator like the reference's fresh net
        clone.model.load_state_dict(self.model.state_dict())
        return clone

    def log_weight_statistics(self, step, weights_logging_agent_0=None, weights_logging_agent_1=None,
                              weights_logging_adversary=None, role=None):
        w = self.model.get_perturbable_weights()
        rec = {"step": step, "mean": w.mean(), "min": w.min(), "max": w.max(), "std": w.std()}
        target = {"agent_0": weights_logging_agent_0, "agent_1": weights_logging_agent_1,
                  "adversary_0": weights_logging_adversary}.get(role)
        if target is not None:
            target.append(rec)
