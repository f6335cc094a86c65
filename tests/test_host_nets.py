"""Host-side helpers of the flat parameter arenas (neosr_amd/hip/nets.py) — CPU: the cached parameter slots follow
`module.parameters()`, see re-assigned Parameters, survive deepcopy; the direct-gradient cache is dropped by copies."""

from __future__ import annotations

import copy
import pickle

import torch
from torch import nn


def _net():
    torch.manual_seed(0)
    shared = nn.Linear(4, 4)
    return nn.Sequential(nn.Conv2d(3, 8, 3), nn.PReLU(8), nn.Sequential(shared, nn.Linear(4, 2, bias=False)), shared)


def test_parameter_slots_follow_module_parameters():
    from neosr_amd.hip import nets

    net = _net()
    ps = nets.parameters_of(net)
    assert len(ps) == len(list(net.parameters())) and all(a is b for a, b in zip(ps, net.parameters()))   # (tied layer once)
    assert nets.parameter_slots(net) is nets.parameter_slots(net)      # walked once
    net[0].weight = nn.Parameter(torch.zeros_like(net[0].weight))      # a re-assigned Parameter is seen through its slot
    assert nets.parameters_of(net)[0] is net[0].weight
    twin = copy.deepcopy(net)
    assert all(a is b for a, b in zip(nets.parameters_of(twin), twin.parameters()))
    assert all(a is not b for a, b in zip(nets.parameters_of(twin), nets.parameters_of(net)))


def test_flatten_parameters_fast_path_and_rehoming():
    from neosr_amd.hip import nets

    net = _net()
    before = [p.detach().clone() for p in net.parameters()]
    arena = nets.flatten_parameters_(net)
    assert nets.flatten_parameters_(net) is arena                      # steady state: pointer checks only
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    offs, total = nets.arena_layout(list(net.parameters()))
    assert arena.numel() == total and all(p.data_ptr() == arena.data_ptr() + 4 * o for p, o in zip(net.parameters(), offs))
    net[1].weight.data = net[1].weight.data.clone()                    # one tensor re-homed: the arena is rebuilt
    arena2 = nets.flatten_parameters_(net)
    assert arena2 is not arena and net[1].weight.data_ptr() == arena2.data_ptr() + 4 * offs[2]
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    twin = copy.deepcopy(net)                                          # a copy keeps its own arena consistent
    t_arena = nets.flatten_parameters_(twin)
    assert t_arena.data_ptr() != arena2.data_ptr() and torch.equal(t_arena, arena2)


def test_direct_grad_cache_is_not_copied_or_pickled():
    from neosr_amd.hip import nets

    st = nets.DirectGrads()
    assert copy.deepcopy(st) is None and pickle.loads(pickle.dumps(st)) is None
    net = _net()
    net.__dict__["_neosr_direct"] = st
    twin = copy.deepcopy(net)
    assert twin.__dict__["_neosr_direct"] is None
    with nets.direct_param_grads():      # CPU parameters never take the direct path
        assert nets.direct_state(net, nets.parameters_of(net)) is None


def test_block_plan_cache_is_not_copied_or_pickled():
    """`module._plan_meta` holds ctypes structs with pointer fields after the first forward: a cache that copies drop
    (copy.deepcopy / torch.save of a SwinIR / HAT network after a forward raised)."""
    import ctypes
    from neosr_amd.hip import transformer as T

    class S(ctypes.Structure):
        _fields_ = [("p", ctypes.c_void_p)]

    m = T.PlanMeta({"names": ("a",), "_desc": ((1, 2), (S(), 3, 4))})
    assert copy.deepcopy(m) is None and pickle.loads(pickle.dumps(m)) is None
    mod = torch.nn.Linear(2, 2)
    mod._plan_meta = m
    twin = copy.deepcopy(mod)
    assert twin._plan_meta is None and torch.equal(twin.weight, mod.weight)
