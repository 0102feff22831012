"""Host-side mirror of the reference's ``neural_collaborative_filtering/util.py`` (MLP topology + checkpoint I/O)."""
import torch
from torch import nn


def build_MLP_layers(input_size, layer_sizes: list, dropout_rate, output_size=1) -> nn.Sequential:
    """Same module layout as reference util.py:5-18 so state_dict keys match (MLP.0, MLP.3, ... with dropout;
    MLP.0, MLP.2, ... when ``dropout_rate is None``): Linear, then (ReLU, [Dropout], Linear) per further layer,
    the last Linear being ``output_size`` wide with no activation after it."""
    widths = list(layer_sizes) + [output_size]
    mods = [nn.Linear(input_size, widths[0])]
    for prev, cur in zip(widths[:-1], widths[1:]):
        mods.append(nn.ReLU())
        if dropout_rate is not None:
            mods.append(nn.Dropout(dropout_rate))
        mods.append(nn.Linear(prev, cur))
    return nn.Sequential(*mods)


def row_major_embedding_(lin: nn.Linear) -> nn.Linear:
    """Re-lay the weight of an embedding ``Linear(num_ids, E)`` in place: same logical tensor ``[E, num_ids]`` (state_dict
    keys, shapes and values are unchanged: reference checkpoints load and save as before), but stored id-major — the
    parameter is the transpose VIEW of a contiguous ``[num_ids, E]`` buffer.  ``Linear(onehot(i)) = W[:, i] + b`` is then
    a contiguous 4·E-byte row: a training step gathers rows, scatters gradient rows (full-rate 256-byte atomics instead of
    E scattered 4-byte ones) and Adam runs elementwise over the buffers, with no table-sized transpose anywhere."""
    w = lin.weight.data
    lin.weight = nn.Parameter(w.t().contiguous().t(), requires_grad=lin.weight.requires_grad)
    return lin


def is_row_major_embedding(weight: torch.Tensor) -> bool:
    return weight.dim() == 2 and weight.t().is_contiguous() and not weight.is_contiguous()


def mlp_linears(seq: nn.Sequential):
    """The Linear modules of a build_MLP_layers Sequential, in order."""
    return [m for m in seq if isinstance(m, nn.Linear)]


def load_model(file, ModelClass=None, map_location="cpu", **kargs):
    """Reads a reference checkpoint ``[state_dict, kwargs]`` (reference models/base.py:18-19, util.py:21-34).

    Unlike the reference this passes ``map_location`` (its shipped checkpoints were saved from CUDA tensors and
    cannot be opened on a CPU-only host otherwise) and ``weights_only=True`` (nothing in the file is executed).
    """
    state, kwargs = torch.load(file, map_location=map_location, weights_only=True)
    kwargs = dict(kwargs, **kargs)
    if ModelClass is None:
        return state, kwargs
    model = ModelClass(**kwargs)
    model.load_state_dict(state)
    return model


def params_version(module: nn.Module) -> tuple:
    """Cheap fingerprint of a module's parameters: changes whenever a parameter is updated in place, re-assigned
    or moved (host and device allocations share one address space, so the address covers the device).  Used to
    invalidate the derived inference tensors (embedding tables, packed MLP weights).  Walks ``_parameters`` /
    ``_modules`` directly: ``module.parameters()`` builds names and hashes every tensor into a memo set, which made this
    the largest single host cost of a forward (a shared submodule is simply visited twice here)."""
    out = []
    stack = [module]
    while stack:                                   # an explicit stack: the recursive generator cost 60 frames per forward
        m = stack.pop()
        for p in m._parameters.values():
            if p is not None:
                out.append(p.data_ptr())
                out.append(p._version)
        for child in m._modules.values():
            if child is not None:
                stack.append(child)
    return tuple(out)


def use_native(module: nn.Module, *tensors) -> bool:
    """The HIP kernels implement the *scoring* forward.  They run when the module is in eval mode and autograd is
    not recording; a training step (module.training or grad enabled with trainable parameters) keeps to the
    differentiable torch ops, as SURVEY.md §8(a9) scopes it."""
    if module.training:
        return False
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return False
    return True


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("deeprecommendation_amd scores on an MI355X only: inputs must be CUDA(HIP) tensors "
                               f"(got {t.device}). There is no CPU fallback.")


def host_cpu_share() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container can see 256
    CPUs and own 16)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cap_host_threads() -> int:
    """Lower torch's intra-op thread count to the CPU share when it exceeds it.  torch sizes its OpenMP pool from the
    visible CPU count; under a cgroup quota the surplus workers spin after every parallel region, burn the quota, and the
    kernel then throttles the one thread that enqueues GPU work (measured on a 16-of-256-CPU box: 129 -> 400-1400 us of
    host time per batch of the evaluation loop).  Returns the thread count in force."""
    share = host_cpu_share()
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)
    return torch.get_num_threads()
