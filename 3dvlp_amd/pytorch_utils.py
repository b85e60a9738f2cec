"""SharedMLP and its 1x1-conv building blocks with the reference's state_dict layout.

Mirrors lib/pointnet2/pytorch_utils.py: SharedMLP :11-36, _BNBase :39-47, BatchNorm1d/2d :50-59,
_ConvBase :69-121, Conv1d :124-158, Conv2d :161-195.  Only the parameter NAMES and shapes are a
contract (checkpoints of the reference must load): ``layer{i}.conv.weight (Co,Ci,1,1)`` without bias
when bn is on, ``layer{i}.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``.
The dense arithmetic itself is a plain library GEMM (MIOpen / hipBLASLt through torch); the fused
gather+MLP+max kernel of the SA layers reads its weights from these same tensors.
"""
import torch.nn as nn


class _BNBase(nn.Sequential):
    def __init__(self, in_size, batch_norm, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class _ConvBase(nn.Sequential):
    def __init__(self, in_size, out_size, conv, batch_norm, kernel_size, stride, padding, activation, bn, init,
                 bias, preact, name):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        bn_unit = batch_norm(in_size if preact else out_size) if bn else None

        def add_norm_act():
            if bn_unit is not None:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            add_norm_act()
        self.add_module(name + "conv", conv_unit)
        if not preact:
            add_norm_act()


class Conv1d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0, activation=nn.ReLU(inplace=True),
                 bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
        super().__init__(in_size, out_size, nn.Conv1d, BatchNorm1d, kernel_size, stride, padding, activation, bn,
                         init, bias, preact, name)


class Conv2d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False,
                 name=""):
        super().__init__(in_size, out_size, nn.Conv2d, BatchNorm2d, kernel_size, stride, padding, activation, bn,
                         init, bias, preact, name)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d (+BN +ReLU) applied to a (B,C,npoint,nsample) tensor."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # the very first pre-activation layer has no norm/act
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=(not plain) and bn,
                                   activation=None if plain else activation, preact=preact))
