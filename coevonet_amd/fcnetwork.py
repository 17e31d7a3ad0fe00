"""Host-side mirror of the reference's ``FCNetwork`` (MPE/fcnetwork.py:9-259): same constructor, same method names
and argument meaning, same error behaviour; the forward runs in libcoevo (HIP), never on the CPU.

Parameters live in ONE flat float32 torch-CPU tensor in ``parameters()`` order (fc1.w, fc1.b, ln1.w, ln1.b, fc2.w,
fc2.b, ln2.w, ln2.b, output.w, output.b); the named parameters are views into it, so ``param.data += noise``
(Agent.mutate, agent.py:27-29) and ``load_state_dict`` behave as with ``nn.Module`` while the flat vector is what is
shipped to the device slab.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from itertools import chain

import numpy as np
import torch

from . import lib as L

H1, H2 = 512, 256


def param_shapes(D, n_actions=5):
    return [("fc1.weight", (H1, D)), ("fc1.bias", (H1,)), ("ln1.weight", (H1,)), ("ln1.bias", (H1,)),
            ("fc2.weight", (H2, H1)), ("fc2.bias", (H2,)), ("ln2.weight", (H2,)), ("ln2.bias", (H2,)),
            ("output.weight", (n_actions, H2)), ("output.bias", (n_actions,))]


class _Layer:
    """what ``self.layers`` / ``named_modules`` hand out in the reference: an object with .weight and .bias"""

    def __init__(self, name, weight, bias, is_norm):
        self.name, self.weight, self.bias, self.is_norm = name, weight, bias, is_norm


class FCNetwork:
    def __init__(self, input_channels, n_actions, precision):
        if n_actions != 5:
            raise ValueError("the HIP policy kernel is built for simple_adversary's 5 discrete actions")
        if input_channels not in (8, 10):
            raise ValueError("the HIP policy kernel is built for simple_adversary observations (8 or 10 wide)")
        if precision != "float32":
            # the reference's float16 option (MPE/fcnetwork.py:13) is outside the parity scope
            raise ValueError(f"Unsupported precision: {precision}")
        self.dtype = torch.float32
        self.precision = precision
        self.input_channels = int(input_channels)
        self.n_actions = int(n_actions)
        shapes = param_shapes(self.input_channels, self.n_actions)
        total = sum(int(np.prod(s)) for _, s in shapes)
        self._flat = torch.empty(total, dtype=torch.float32)
        self._params = OrderedDict()
        off = 0
        for name, shp in shapes:
            n = int(np.prod(shp))
            self._params[name] = self._flat[off:off + n].view(*shp)
            off += n
        # torch default init, consuming the global generator exactly like the three nn.Linear constructions of the
        # reference (MPE/fcnetwork.py:14-20); LayerNorm starts at gamma=1, beta=0 and draws nothing
        # (drawn straight into the flat buffer with nn.Linear.reset_parameters' own calls and order - weight:
        # kaiming_uniform_(a=sqrt(5)), bias: uniform_(-1/sqrt(fan_in), 1/sqrt(fan_in)) - instead of building a throw-away
        # nn.Linear and copying: same generator stream, half the time; the weight-hash fixtures pin it)
        for prefix, (i, o) in (("fc1", (self.input_channels, H1)), ("fc2", (H1, H2)), ("output", (H2, self.n_actions))):
            torch.nn.init.kaiming_uniform_(self._params[prefix + ".weight"], a=math.sqrt(5))
            bound = 1.0 / math.sqrt(i) if i > 0 else 0.0
            torch.nn.init.uniform_(self._params[prefix + ".bias"], -bound, bound)
        for ln in ("ln1", "ln2"):
            self._params[ln + ".weight"].fill_(1.0)
            self._params[ln + ".bias"].zero_()
        p = self._params
        self.fc1 = _Layer("fc1", p["fc1.weight"], p["fc1.bias"], False)
        self.ln1 = _Layer("ln1", p["ln1.weight"], p["ln1.bias"], True)
        self.fc2 = _Layer("fc2", p["fc2.weight"], p["fc2.bias"], False)
        self.ln2 = _Layer("ln2", p["ln2.weight"], p["ln2.bias"], True)
        self.output = _Layer("output", p["output.weight"], p["output.bias"], False)
        self.layers = [self.fc1, self.fc2, self.output]

    # ------------------------------------------------------------------ nn.Module-like surface
    def parameters(self):
        return iter(self._params.values())

    def named_modules(self):
        yield "", self
        for l in (self.fc1, self.ln1, self.fc2, self.ln2, self.output):
            yield l.name, l

    def state_dict(self):
        return OrderedDict((k, v.detach()) for k, v in self._params.items())

    def load_state_dict(self, sd, strict=True):
        for k, v in self._params.items():
            if k in sd:
                v.copy_(torch.as_tensor(sd[k], dtype=torch.float32))
            elif strict:
                raise KeyError(f"Missing key in state_dict: {k}")

    def flat(self) -> np.ndarray:
        """the whole net in parameters() order (a view, float32)"""
        return self._flat.numpy()

    def set_flat(self, flat):
        self._flat.copy_(torch.as_tensor(np.asarray(flat, dtype=np.float32)))

    # ------------------------------------------------------------------ forward (HIP)
    def forward(self, x, args):
        """logits of one observation; raises ValueError like MPE/fcnetwork.py:39-65 on NaN/inf"""
        logits, _ = self._device_forward(x)
        return logits

    def determine_action(self, inputs, args):
        """first index of the maximal logit (strict '>' scan, MPE/fcnetwork.py:73-90)"""
        _, action = self._device_forward(inputs)
        return action

    def _device_forward(self, x):
        D = self.input_channels
        x = torch.as_tensor(x, dtype=torch.float32).reshape(-1)
        if x.numel() != D:
            raise ValueError(f"expected an observation of width {D}, got {x.numel()}")
        dev = "cuda"
        flat = self._flat.to(dev)
        slab = torch.zeros(L.fc_slab_stride(D), dtype=torch.float32, device=dev)
        L.call("coevo_fc_pack", L._p(flat), L._p(slab), 1, D)
        obs = torch.zeros(1, L.OBS_STRIDE, dtype=torch.float32, device=dev)
        obs[0, :D] = x.to(dev)
        tasks = np.zeros(1, dtype=L.TASK_DTYPE)
        tasks[0] = (0, 0, 1, D, 0)
        d_tasks = L.tasks_to_device(tasks, dev)
        actions = torch.zeros(1, dtype=torch.int32, device=dev)
        logits = torch.zeros(1, L.LOGIT_STRIDE, dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        L.call("coevo_fc_forward_argmax", L._p(slab), L._p(d_tasks), 1, 1, L._p(obs), L._p(actions), L._p(logits),
               L._p(status))
        L.raise_on_status(status)
        return logits[0, :self.n_actions].cpu(), int(actions.item())

    # ------------------------------------------------------------------ weight get/set (reference names)
    def get_weights(self, layers=None):
        sd = self.state_dict()
        if layers is None:
            return {k: v.clone() for k, v in sd.items()}
        return {k: v.clone() for k, v in sd.items() if any(k.startswith(l) for l in layers)}

    def set_weights(self, new_weights, layers=None):
        cur = self.state_dict()
        keys = list(new_weights.keys()) if layers is not None else list(cur.keys())
        for k in keys:
            if k not in new_weights:
                raise ValueError(f"Missing key in new_weights: {k}")
            if tuple(new_weights[k].shape) != tuple(cur[k].shape):
                raise ValueError(f"Shape mismatch for key '{k}': expected {cur[k].shape}, got {new_weights[k].shape}")
        self.load_state_dict({k: new_weights[k] for k in keys}, strict=False)

    def get_perturbable_layers(self):
        return [l for name, l in self.named_modules() if name != "" and not l.is_norm]

    def get_perturbable_weights(self):
        return self.get_weights_ES(self.get_perturbable_layers())

    def get_weights_ES(self, layers=None):
        layers = layers if layers else self.layers
        parts = chain(*[(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in layers])
        return np.concatenate([w.flatten() for w in parts])

    def set_weights_ES(self, flat_weights, args, layers=None):
        if layers is None:
            layers = self.get_perturbable_layers()
        i = 0
        for l in layers:
            n = l.weight.numel()
            l.weight.copy_(torch.tensor(np.asarray(flat_weights[i:i + n]).reshape(tuple(l.weight.shape)),
                                        dtype=torch.float32))
            i += n
            n = l.bias.numel()
            l.bias.copy_(torch.tensor(np.asarray(flat_weights[i:i + n]).reshape(tuple(l.bias.shape)),
                                      dtype=torch.float32))
            i += n

    def set_perturbable_weights(self, weights_to_set, args):
        self.set_weights_ES(weights_to_set, args, self.get_perturbable_layers())
