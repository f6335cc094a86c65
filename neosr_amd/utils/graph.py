"""`compile = true` (the reference hands the network to torch.compile, neosr/models/base.py:136-137): here the
train-mode forward and backward of a generator that Python dispatches op by op (SwinIR / HAT families: thousands of
launches per iteration) are captured once into two hipGraphs and replayed — the launch sequence is fixed by the
batch and patch size, so nothing is traced or re-compiled.  `torch.cuda.make_graphed_callables` does the capture
(static input / output / gradient buffers in a private pool, autograd-aware); the packed convolution images are
rebuilt by a node of the forward graph (neosr_amd.hip.layers.FORCE_REPACK_IN_CAPTURE), DropPath draws come from
torch's graph-safe Philox generator.  Calls that do not look like the captured one (eval mode, no_grad, another
shape) run the original forward."""

from __future__ import annotations

import torch

from neosr_amd.hip import layers


def graph_train_forward(net: torch.nn.Module, sample: torch.Tensor) -> None:
    """Capture `net(sample)` + its backward; patches `net.forward` in place.  Must run before the first eager
    backward pass through `net` (AccumulateGrad nodes bound to the default stream break a capture)."""
    if not net.training:
        raise RuntimeError("graph_train_forward: capture the network in train mode")
    eager = net.forward
    shape, dtype = tuple(sample.shape), sample.dtype
    from neosr_amd.hip import transformer

    layers.FORCE_REPACK_IN_CAPTURE = True
    # (warm-up and capture use autograd.grad on the parameters: reductions must not be deferred — they are not, outside a
    # `deferred_reductions()` scope, unless NEOSR_AMD_DEFER_REDUCE=1 forces them process-wide)
    defer, transformer.DEFER_REDUCTIONS = transformer.DEFER_REDUCTIONS, False
    try:
        torch.cuda.make_graphed_callables(net, (sample,))
    finally:
        layers.FORCE_REPACK_IN_CAPTURE = False
        transformer.DEFER_REDUCTIONS = defer
    graphed = net.forward

    def forward(x):
        if net.training and torch.is_grad_enabled() and x.is_cuda and tuple(x.shape) == shape and x.dtype == dtype:
            return graphed(x)
        return eager(x)

    net.forward = forward
    net._neosr_graphed = True  # noqa: SLF001
