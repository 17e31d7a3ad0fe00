"""Host-side mirror of the reference's ``DeepQN`` (Atari/deepqn.py:7-231) and ``AtariAgent`` (Atari/atari_agent.py:8-31)
with the forward on the MI355X (csrc/deepqn.hip).

Only the policy network is in scope for Atari: the reference's Atari episode loop does not run
(five independent TypeErrors) and the ALE emulator is not part of this image, so ``determine_action`` implements the rule
the reference's docstring intends (first index of the maximal Q value, the same scan as MPE/fcnetwork.py:78-85) and
``batched_actions`` is the entry point a population engine uses: many nets x many uint8 frames in three launches.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import lib as L
from .agent import Agent

PARAM_ORDER = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias", "fc1.weight",
               "fc1.bias", "output.weight", "output.bias", "vbn1.weight", "vbn1.bias", "vbn2.weight", "vbn2.bias",
               "vbn3.weight", "vbn3.bias"]


LAYER_ORDER = ["conv1", "conv2", "conv3", "fc1", "output", "vbn1", "vbn2", "vbn3"]


class _Layer:
    """a layer as the reference's accessors see it: .weight / .bias tensors (views of the owner's flat parameter vector)"""

    def __init__(self, name, weight, bias):
        self.name, self.weight, self.bias = name, weight, bias


class DeepQN:
    def __init__(self, input_channels, n_actions, precision):
        if precision != "float32":
            raise ValueError(f"Unsupported precision: {precision}")
        self.input_channels, self.n_actions, self.precision = int(input_channels), int(n_actions), precision
        if L.load().coevo_dqn_param_count(self.input_channels, self.n_actions) < 0:
            raise ValueError("unsupported DeepQN shape for the HIP kernels (1..6 channels, <= 32 actions)")
        # construction order and torch-generator consumption of Atari/deepqn.py:16-37 (BatchNorm draws nothing)
        mods = OrderedDict([
            ("conv1", torch.nn.Conv2d(self.input_channels, 32, kernel_size=8, stride=4)),
            ("conv2", torch.nn.Conv2d(32, 64, kernel_size=4, stride=2)),
            ("conv3", torch.nn.Conv2d(64, 64, kernel_size=3, stride=1)),
            ("fc1", torch.nn.Linear(64 * 7 * 7, 512)),
            ("output", torch.nn.Linear(512, self.n_actions)),
        ])
        shapes = OrderedDict()
        for name, m in mods.items():
            shapes[name + ".weight"], shapes[name + ".bias"] = tuple(m.weight.shape), tuple(m.bias.shape)
        for name, c in (("vbn1", 32), ("vbn2", 64), ("vbn3", 64)):
            shapes[name + ".weight"], shapes[name + ".bias"] = (c,), (c,)
        total = sum(int(np.prod(shapes[k])) for k in PARAM_ORDER)
        self._flat = torch.empty(total, dtype=torch.float32)
        self._params, off = OrderedDict(), 0
        for k in PARAM_ORDER:
            n = int(np.prod(shapes[k]))
            self._params[k] = self._flat[off:off + n].view(*shapes[k])
            off += n
        for name, m in mods.items():
            self._params[name + ".weight"].copy_(m.weight.detach())
            self._params[name + ".bias"].copy_(m.bias.detach())
        for name in ("vbn1", "vbn2", "vbn3"):
            self._params[name + ".weight"].fill_(1.0)
            self._params[name + ".bias"].zero_()
        # layer views in the reference's `self.layers` order (Atari/deepqn.py:14-37): objects with .weight / .bias tensors
        # that alias the flat vector, what get_weights_ES / set_weights_ES walk
        self.layers = []
        for name in LAYER_ORDER:
            layer = _Layer(name, self._params[name + ".weight"], self._params[name + ".bias"])
            setattr(self, name, layer)
            self.layers.append(layer)
        # BatchNorm buffers: part of the reference's state_dict (training-mode BatchNorm at batch 1 never reads them, and
        # the HIP forward does not maintain them): carried so that state_dicts exchange with strict=True
        self._buffers = OrderedDict()
        for name, c in (("vbn1", 32), ("vbn2", 64), ("vbn3", 64)):
            self._buffers[name + ".running_mean"] = torch.zeros(c)
            self._buffers[name + ".running_var"] = torch.ones(c)
            self._buffers[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def parameters(self):
        return iter(self._params.values())

    def named_modules(self):
        """('', self) then the layers in registration order, as torch.nn.Module.named_modules yields them"""
        yield "", self
        for layer in self.layers:
            yield layer.name, layer

    def state_dict(self):
        """keys and order of the reference's state_dict: per layer its parameters, BatchNorm layers followed by their buffers"""
        sd = OrderedDict()
        for layer in self.layers:
            sd[layer.name + ".weight"] = self._params[layer.name + ".weight"].detach()
            sd[layer.name + ".bias"] = self._params[layer.name + ".bias"].detach()
            for suffix in ("running_mean", "running_var", "num_batches_tracked"):
                if layer.name + "." + suffix in self._buffers:
                    sd[layer.name + "." + suffix] = self._buffers[layer.name + "." + suffix]
        return sd

    def load_state_dict(self, sd, strict=True):
        for k, v in self._params.items():
            if k in sd:
                v.copy_(torch.as_tensor(sd[k], dtype=torch.float32))
            elif strict:
                raise KeyError(f"Missing key in state_dict: {k}")
        for k, v in self._buffers.items():
            if k in sd:
                v.copy_(torch.as_tensor(sd[k], dtype=v.dtype))

    # ---- the weight accessors of Atari/deepqn.py:63-231 (same names, argument meaning and errors) -----------------------
    def get_weights(self, layers=None):
        """dict of cloned state_dict entries; `layers`: name prefixes (:63-87)"""
        sd = self.state_dict()
        if layers is None:
            return {k: v.clone() for k, v in sd.items()}
        return {k: v.clone() for k, v in sd.items() if any(k.startswith(layer) for layer in layers)}

    def set_weights(self, new_weights, layers=None):
        """(:90-123) with `layers` only the keys of new_weights are touched; every touched key must be present and of the
        right shape"""
        cur = self.state_dict()
        keys = list(new_weights.keys()) if layers is not None else list(cur.keys())
        for k in keys:
            if k not in new_weights:
                raise ValueError(f"Missing key in new_weights: {k}")
            if tuple(new_weights[k].shape) != tuple(cur[k].shape):
                raise ValueError(f"Shape mismatch for key '{k}': expected {cur[k].shape}, got {new_weights[k].shape}")
        self.load_state_dict({k: new_weights[k] for k in keys}, strict=False)

    def get_perturbable_layers(self):
        """every layer that is not a BatchNorm2d, in registration order (:158-171)"""
        return [layer for layer in self.layers if not layer.name.startswith("vbn")]

    def get_weights_ES(self, layers=None):
        """weights then biases of each layer, flattened and concatenated (:130-155); default: all eight layers = the
        canonical flat order of include/coevo.h"""
        layers = layers if layers else self.layers
        return np.concatenate([t.detach().numpy().ravel() for layer in layers for t in (layer.weight, layer.bias)])

    def get_perturbable_weights(self):
        return self.get_weights_ES(self.get_perturbable_layers())

    def set_weights_ES(self, flat_weights, args, layers=None):
        """(:174-215) default: the perturbable layers"""
        if getattr(args, "precision", "float32") == "float16":
            raise ValueError("Unsupported precision: float16")
        layers = self.get_perturbable_layers() if layers is None else layers
        flat_weights = np.asarray(flat_weights)
        i = 0
        for layer in layers:
            for t in (layer.weight, layer.bias):
                n = t.numel()
                t.data.copy_(torch.tensor(flat_weights[i:i + n].reshape(tuple(t.shape)), dtype=torch.float32))
                i += n

    def set_perturbable_weights(self, weights_to_set, args):
        self.set_weights_ES(weights_to_set, args, self.get_perturbable_layers())

    def flat(self):
        return self._flat.numpy()

    def set_flat(self, flat):
        self._flat.copy_(torch.as_tensor(np.asarray(flat, dtype=np.float32)))

    @staticmethod
    def _to_frames(x):
        """[1, C, 84, 84] tensor with values 0..255 (what preprocess_observation yields) or uint8 [84, 84, C]"""
        x = torch.as_tensor(x)
        if x.dtype == torch.uint8 and x.dim() == 3:
            return x.contiguous()[None]
        if x.dim() == 4:
            xr = x[0].permute(1, 2, 0)
            u8 = xr.to(torch.uint8)
            if not torch.equal(u8.to(xr.dtype), xr):
                raise ValueError("the HIP DeepQN kernel takes uint8 frames (integral values 0..255)")
            return u8.contiguous()[None]
        raise ValueError("expected a [1, C, 84, 84] tensor or a uint8 [84, 84, C] frame")

    def forward(self, x):
        logits, _ = batched_actions([self.flat()], [self._to_frames(x).numpy()], self.input_channels, self.n_actions)
        return torch.from_numpy(logits[0][None])

    def determine_action(self, inputs, args):
        _, actions = batched_actions([self.flat()], [self._to_frames(inputs).numpy()], self.input_channels,
                                     self.n_actions)
        return int(actions[0])


def batched_actions(nets_flat, frames_per_net, C, n_actions, device="cuda", fc1_tiled=False):
    """nets_flat: list of flat parameter vectors; frames_per_net: list of uint8 arrays [r_i, 84, 84, C] (r_i <= 16).
    -> (logits [sum r_i, n_actions] float32, actions [sum r_i] int32).  One task per net.  fc1_tiled: the slab keeps fc1 in
    the layout of v_mfma_f32_16x16x4 (include/coevo.h COEVO_DQN_FC1_TILED; same results)"""
    n_nets = len(nets_flat)
    Cw = C | (L.DQN_FC1_TILED if fc1_tiled else 0)
    stride = int(L.load().coevo_dqn_slab_stride(C, n_actions))
    flat = torch.from_numpy(np.ascontiguousarray(np.stack(nets_flat), dtype=np.float32)).to(device)
    slab = torch.zeros(n_nets, stride, dtype=torch.float32, device=device)
    L.call("coevo_dqn_pack", L._p(flat), L._p(slab), n_nets, Cw, n_actions)
    tasks = np.zeros(n_nets, dtype=L.DQN_TASK_DTYPE)
    row = 0
    for i, fr in enumerate(frames_per_net):
        assert fr.dtype == np.uint8 and fr.shape[1:] == (84, 84, C) and 1 <= fr.shape[0] <= L.DQN_MAX_ROWS
        tasks[i] = (i * stride, row, fr.shape[0])
        row += fr.shape[0]
    frames = torch.from_numpy(np.ascontiguousarray(np.concatenate(frames_per_net))).to(device)
    d_tasks = L.tasks_to_device(tasks, device)
    actions = torch.zeros(row, dtype=torch.int32, device=device)
    logits = torch.zeros(row, L.DQN_LOGIT_STRIDE, dtype=torch.float32, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    ws = torch.zeros(int(L.load().coevo_dqn_workspace_bytes(row)) // 4, dtype=torch.float32, device=device)
    L.call("coevo_dqn_forward_argmax", L._p(slab), L._p(d_tasks), n_nets, int(max(f.shape[0] for f in frames_per_net)),
           row, Cw, n_actions, L._p(frames), L._p(actions), L._p(logits), L._p(status), L._p(ws))
    L.raise_on_status(status)
    return logits[:, :n_actions].cpu().numpy(), actions.cpu().numpy()


class AtariAgent(Agent):
    """Atari/atari_agent.py:8-31 (clone keeps the reference's two-argument form)"""

    def __init__(self, env, args):
        self.input_channels = env.observation_space(env.agents[0]).shape[-1]
        self.n_actions = env.action_space(env.agents[0]).n
        self.model = DeepQN(self.input_channels, self.n_actions, args.precision)
        self.optimizer = None
        super().__init__(self.model, self.optimizer, args)

    def clone(self, env, args):
        clone = AtariAgent(env, args)
        clone.model.load_state_dict(self.model.state_dict())
        return clone
