"""PyG 2.1.0 `Linear`: weight [out, in], lazy when in_channels == -1, kaiming-uniform(a=sqrt 5)."""
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.parameter import Parameter, UninitializedParameter


class Linear(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None,
                 bias_initializer=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        if in_channels > 0:
            self.weight = Parameter(torch.empty(out_channels, in_channels))
        else:
            self.weight = UninitializedParameter()
            self._hook = self.register_forward_pre_hook(self._lazy_init)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._register_load_state_dict_pre_hook(self._lazy_load)
        self.reset_parameters()

    def reset_parameters(self):
        if self.in_channels <= 0:
            return
        bound = 1.0 / math.sqrt(self.in_channels)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    @staticmethod
    def _lazy_init(module, inputs):
        if isinstance(module.weight, UninitializedParameter):
            module.in_channels = inputs[0].size(-1)
            module.weight.materialize((module.out_channels, module.in_channels))
            module.reset_parameters()
        module._hook.remove()
        delattr(module, "_hook")

    def _lazy_load(self, state_dict, prefix, *args):
        w = state_dict.get(prefix + "weight")
        if w is not None and isinstance(self.weight, UninitializedParameter):
            self.in_channels = w.size(-1)
            self.weight.materialize((self.out_channels, self.in_channels))
            if hasattr(self, "_hook"):
                self._hook.remove()
                delattr(self, "_hook")

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)
