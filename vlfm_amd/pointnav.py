"""PointNav controller: depth 224x224 + (rho, theta) -> action, batched over environments.

The reference drives the robot towards the goal chosen by the hot path with a pretrained DD-PPO PointNav policy
(vlfm/policy/utils/pointnav_policy.py:50-128, network in vlfm/policy/utils/non_habitat_policy/{nh_pointnav_policy.py,
resnet.py, rnn_state_encoder.py}): ResNet-18 with GroupNorm at 32 base planes on the half-resolution depth image ->
3x3 compression -> fc 2048->512, concatenated with a 32-d goal embedding (rho, cos(-theta), sin(-theta)) and a 32-d
previous-action embedding -> 2-layer LSTM(576 -> 512) -> action head.  The reference's wrapper "can only handle one
environment at a time"; this one carries N environments in one forward, with per-environment resets.

Module and parameter NAMES follow the checkpoint format (``net.visual_encoder.backbone.layer1.0.convs.0.weight``,
``net.state_encoder.rnn.weight_ih_l0``, ``action_distribution.mu_maybe_std.weight`` ...), so ``data/pointnav_weights.pth``
of a VLFM install loads with ``load_state_dict``.  Both heads exist: the continuous one of the non-Habitat policy
(tanh mean of a Gaussian: linear / angular velocity) and the discrete one Habitat's checkpoint carries
(``action_distribution.linear``: STOP / FORWARD / LEFT / RIGHT logits).  PyTorch-ROCm forward (MIOpen convolutions,
rocBLAS LSTM) -- model code is plumbing here, as for the other networks; no weights exist offline.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HIDDEN = 512


def _gn_conv(cin: int, cout: int, k: int, stride: int, groups: int) -> list:
    return [nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), nn.GroupNorm(groups, cout)]


class _Block(nn.Module):
    """conv3x3-GN-ReLU-conv3x3-GN + identity / (conv1x1-GN) shortcut (resnet.py:16-49)."""

    def __init__(self, cin: int, cout: int, groups: int, stride: int) -> None:
        super().__init__()
        self.convs = nn.Sequential(*_gn_conv(cin, cout, 3, stride, groups), nn.ReLU(True),
                                   *_gn_conv(cout, cout, 3, 1, groups))
        self.downsample = nn.Sequential(*_gn_conv(cin, cout, 1, stride, groups)) if (stride != 1 or cin != cout) else None
        self.relu = nn.ReLU(True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.relu(self.convs(x) + (x if self.downsample is None else self.downsample(x)))


class _ResNet18GN(nn.Module):
    """resnet18(in_channels=1, base_planes=32, ngroups=16) of resnet.py:69-153."""

    def __init__(self, cin: int = 1, base: int = 32, groups: int = 16) -> None:
        super().__init__()
        self.conv1 = nn.Sequential(*_gn_conv(cin, base, 7, 2, groups), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        widths, prev = [base, base * 2, base * 4, base * 8], base
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, f"layer{i + 1}", nn.Sequential(_Block(prev, w, groups, stride), _Block(w, w, groups, 1)))
            prev = w

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.maxpool(self.conv1(x))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class _VisualEncoder(nn.Module):
    """nh_pointnav_policy.py:15-43: NHWC depth -> 2x2 average pool -> ResNet -> 3x3 compression to 128 channels."""

    def __init__(self) -> None:
        super().__init__()
        self.running_mean_and_var = nn.Sequential()
        self.backbone = _ResNet18GN()
        self.compression = nn.Sequential(nn.Conv2d(256, 128, 3, padding=1, bias=False), nn.GroupNorm(1, 128), nn.ReLU(True))

    def forward(self, depth_nhwc: torch.Tensor) -> torch.Tensor:
        x = F.avg_pool2d(depth_nhwc.permute(0, 3, 1, 2), 2)
        return self.compression(self.backbone(x))


class _StateEncoder(nn.Module):
    """2-layer LSTM whose (h, c) travel as ONE [N, 4, 512] tensor: h of both layers, then c of both layers
    (rnn_state_encoder.py:31-65,123-141); an environment whose mask is False starts from a zero state."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int) -> None:
        super().__init__()
        self.rnn = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.num_recurrent_layers = 2 * num_layers
        for name, p in self.rnn.named_parameters():
            nn.init.orthogonal_(p) if "weight" in name else nn.init.constant_(p, 0)

    def forward(self, x: torch.Tensor, state: torch.Tensor, masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        s = state.permute(1, 0, 2)
        s = torch.where(masks.view(1, -1, 1), s, s.new_zeros(()))
        h, c = torch.chunk(s.contiguous(), 2, 0)
        out, (h, c) = self.rnn(x.unsqueeze(0), (h.contiguous(), c.contiguous()))
        return out.squeeze(0), torch.cat((h, c), 0).permute(1, 0, 2)


class _Net(nn.Module):
    def __init__(self, discrete_actions: bool) -> None:
        super().__init__()
        if discrete_actions:
            self.prev_action_embedding_discrete = nn.Embedding(4 + 1, 32)
        else:
            self.prev_action_embedding_cont = nn.Linear(2, 32)
        self.tgt_embeding = nn.Linear(3, 32)  # (sic) the checkpoint's spelling
        self.visual_encoder = _VisualEncoder()
        self.visual_fc = nn.Sequential(nn.Flatten(), nn.Linear(2048, HIDDEN), nn.ReLU(True))
        self.state_encoder = _StateEncoder(HIDDEN + 32 + 32, HIDDEN, 2)
        self.num_recurrent_layers = self.state_encoder.num_recurrent_layers
        self.discrete_actions = discrete_actions

    def forward(self, depth: torch.Tensor, goal: torch.Tensor, state: torch.Tensor, prev_actions: torch.Tensor,
                masks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """nh_pointnav_policy.py:66-107."""
        seen = self.visual_fc(self.visual_encoder(depth))
        where = self.tgt_embeding(torch.stack([goal[:, 0], torch.cos(-goal[:, 1]), torch.sin(-goal[:, 1])], -1))
        if self.discrete_actions:
            prev = prev_actions.squeeze(-1)
            did = self.prev_action_embedding_discrete(torch.where(masks.view(-1), prev + 1, torch.zeros_like(prev)))
        else:
            did = self.prev_action_embedding_cont(masks * prev_actions.float())
        return self.state_encoder(torch.cat([seen, where, did], dim=1), state, masks)


class _GaussianHead(nn.Module):
    """nh_pointnav_policy.py:115-138; only the mean is needed for deterministic control."""

    def __init__(self, n_in: int, n_out: int) -> None:
        super().__init__()
        self.mu_maybe_std = nn.Linear(n_in, 2 * n_out)
        nn.init.orthogonal_(self.mu_maybe_std.weight, gain=0.01)
        nn.init.constant_(self.mu_maybe_std.bias, 0)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        mu, log_std = torch.chunk(self.mu_maybe_std(x).float(), 2, -1)
        return torch.tanh(mu), torch.exp(torch.clamp(log_std, -5, 2))


class _CategoricalHead(nn.Module):
    """Habitat's CategoricalNet [ext]: one linear layer of logits over (STOP, FORWARD, LEFT, RIGHT)."""

    def __init__(self, n_in: int, n_out: int) -> None:
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear(x)


class PointNavResNetPolicy(nn.Module):
    """State-dict compatible with the reference's class of the same name (nh_pointnav_policy.py:141-163)."""

    def __init__(self, discrete_actions: bool = False) -> None:
        super().__init__()
        self.net = _Net(discrete_actions)
        self.action_distribution = _CategoricalHead(HIDDEN, 4) if discrete_actions else _GaussianHead(HIDDEN, 2)
        self.discrete_actions = discrete_actions

    def act(self, observations: Dict[str, torch.Tensor], rnn_hidden_states: torch.Tensor, prev_actions: torch.Tensor,
            masks: torch.Tensor, deterministic: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        feats, state = self.net(observations["depth"], observations["pointgoal_with_gps_compass"], rnn_hidden_states,
                                prev_actions, masks)
        if self.discrete_actions:
            logits = self.action_distribution(feats)
            action = (logits.argmax(-1, keepdim=True) if deterministic
                      else torch.distributions.Categorical(logits=logits).sample().unsqueeze(-1))
        else:
            mean, std = self.action_distribution(feats)
            action = mean if deterministic else mean + std * torch.randn_like(mean)
        return action, state


def image_resize_area(depth: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """``image_resize(.., channels_last=True, interpolation_mode="area")`` for an (N,H,W,1) batch
    (vlfm/obs_transformers/utils.py:9-48; base_objectnav_policy.py:264-269)."""
    x = F.interpolate(depth.permute(0, 3, 1, 2).float(), size=size, mode="area").to(depth.dtype)
    return x.permute(0, 2, 3, 1)


def load_pointnav_policy(file_path: Optional[str], discrete_actions: Optional[bool] = None) -> PointNavResNetPolicy:
    """pointnav_policy.py:131-195 (the branch without habitat_baselines): a bare state dict, old key names accepted;
    a Habitat checkpoint ({"state_dict": .., "config": ..}, ``actor_critic.`` prefixes) is unwrapped as well.
    ``file_path=None``: random initialisation (no weights offline)."""
    if file_path is None:
        return PointNavResNetPolicy(bool(discrete_actions))
    ckpt = torch.load(file_path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt)
    sd = {k[len("actor_critic."):] if k.startswith("actor_critic.") else k: v for k, v in sd.items()}
    if discrete_actions is None:
        discrete_actions = "action_distribution.linear.weight" in sd
    if not discrete_actions and "net.prev_action_embedding_cont.weight" not in sd and "net.prev_action_embedding.weight" in sd:
        sd["net.prev_action_embedding_cont.weight"] = sd["net.prev_action_embedding.weight"]
        sd["net.prev_action_embedding_cont.bias"] = sd["net.prev_action_embedding.bias"]
    if discrete_actions and "net.prev_action_embedding_discrete.weight" not in sd and "net.prev_action_embedding.weight" in sd:
        sd["net.prev_action_embedding_discrete.weight"] = sd["net.prev_action_embedding.weight"]
    policy = PointNavResNetPolicy(discrete_actions)
    own = policy.state_dict()
    missing = [k for k in own if k not in sd]
    if missing:
        raise KeyError(f"pointnav checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
    policy.load_state_dict({k: v for k, v in sd.items() if k in own})
    return policy


class WrappedPointNavResNetPolicy:
    """The reference's wrapper (pointnav_policy.py:50-128) for ``n_envs`` environments: keeps the recurrent state and
    the previous action per environment; ``reset(env_ids)`` zeroes them for the environments whose goal jumped."""

    def __init__(self, ckpt_path: Optional[str] = None, device=None, n_envs: int = 1,
                 discrete_actions: Optional[bool] = None, depth_image_shape: Tuple[int, int] = (224, 224)) -> None:
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.policy = load_pointnav_policy(ckpt_path, discrete_actions).to(self.device).eval()
        self.discrete = self.policy.discrete_actions
        self.n_envs, self.shape = n_envs, tuple(depth_image_shape)
        self.pointnav_test_recurrent_hidden_states = torch.zeros(n_envs, self.policy.net.num_recurrent_layers, HIDDEN,
                                                                 device=self.device)
        self.pointnav_prev_actions = torch.zeros(n_envs, 1 if self.discrete else 2, device=self.device,
                                                 dtype=torch.long if self.discrete else torch.float32)

    @torch.no_grad()
    def act(self, observations: Dict, masks: torch.Tensor, deterministic: bool = False) -> torch.Tensor:
        """observations["depth"] (N,224,224,1) f32, observations["pointgoal_with_gps_compass"] (N,2) = (rho, theta)."""
        obs = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).to(self.device) for k, v in observations.items()}
        obs = {k: v.float() if v.dtype != torch.uint8 else v for k, v in obs.items()}
        action, state = self.policy.act(obs, self.pointnav_test_recurrent_hidden_states, self.pointnav_prev_actions,
                                        masks.to(self.device).view(-1, 1), deterministic=deterministic)
        self.pointnav_prev_actions = action.clone()
        self.pointnav_test_recurrent_hidden_states = state
        return action

    @torch.no_grad()
    def act_on_depth(self, depth: torch.Tensor, rho_theta: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """Full-resolution depth (N,H,W) or (N,H,W,1) in [0,1] -> area resize to ``depth_image_shape`` -> act, the
        sequence of BaseObjectNavPolicy._pointnav (base_objectnav_policy.py:262-283), deterministic."""
        d = depth.to(self.device)
        if d.dim() == 3:
            d = d.unsqueeze(-1)
        return self.act({"depth": image_resize_area(d, self.shape),
                         "pointgoal_with_gps_compass": rho_theta.to(self.device, torch.float32)}, masks, deterministic=True)

    def reset(self, env_ids=None) -> None:
        if env_ids is None:
            self.pointnav_test_recurrent_hidden_states.zero_()
            self.pointnav_prev_actions.zero_()
        else:
            idx = torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
            self.pointnav_test_recurrent_hidden_states[idx] = 0
            self.pointnav_prev_actions[idx] = 0
