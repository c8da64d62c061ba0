"""``conv(x, edge_index)`` whose dense block waits one call for the activation that follows it.

The reference applies the ReLU OUTSIDE the conv (``/root/reference/models/model.py:71,77``:
``F.relu(conv(x, graph.edge_index))``), while the library's fast path applies it in the dense block's MFMA
epilogue and writes the result straight into the next layer's hop slab.  So that the UNCHANGED reference wiring
reaches that path, a plain PyG-style call returns a ``DeferredActivation``: a ``torch.Tensor`` subclass without
storage that stands for the conv's output.  The first thing done to it decides how the layer runs:

* ``F.relu(y)`` / ``torch.relu(y)`` / ``y.relu()`` (any of their in-place forms) - the layer runs ONCE with the
  ReLU fused and that result is returned: no pre-activation tensor, no elementwise launch, no mask pass in backward;
* anything else (``y.sum()``, ``y + 1``, ``y.detach()``, ``torch.cat([y, ...])``, ``y.data_ptr()`` ...) - the layer
  runs without activation and the operation is applied to that tensor, exactly as if the conv had returned it.

Shape, dtype, device and ``requires_grad`` are answered without running anything.  The value is computed in the grad
mode and on the stream that are current when it is first needed - for the reference's ``F.relu(conv(...))`` that is
the call site itself.  The two autograd entry points that take tensors without dispatching on them
(``torch.autograd.backward([y], ...)``, ``torch.autograd.grad(y, ...)``) need ``y.value()`` (they raise on the wrapper);
``y.backward(...)`` works as is.  ``deformcontact_amd.nn.conv.DEFER_ACTIVATION = False`` (or ``DC_DEFER_ACT=0``) turns the
mechanism off: the conv then returns its output tensor directly and a following ``F.relu`` is a launch of its own.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_RELU_FUNCS = {F.relu, torch.relu, torch.Tensor.relu, torch.relu_, torch.Tensor.relu_, F.relu_}

#: answered from the wrapper's own metadata (no launch)
_META = {"shape", "size", "dim", "ndim", "ndimension", "dtype", "device", "numel", "nelement", "is_cuda", "layout",
         "requires_grad", "element_size", "is_floating_point", "is_complex", "__len__", "is_sparse", "is_quantized",
         "is_meta", "is_nested", "get_device"}


def _func_name(func):
    if getattr(func, "__name__", None) == "__get__":            # a property: Tensor.shape.__get__
        return getattr(getattr(func, "__self__", None), "__name__", None)
    return getattr(func, "__name__", None)


class DeferredActivation(torch.Tensor):
    @staticmethod
    def __new__(cls, run, shape, dtype, device, requires_grad):
        # (the wrapper itself never requires grad: handed to torch.autograd.backward / grad directly - calls that do not
        # dispatch on their arguments - it raises instead of being taken for a leaf; `.requires_grad` is answered below)
        return torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=False)

    def __init__(self, run, shape, dtype, device, requires_grad):
        #: ``run(relu: bool) -> Tensor``: the layer, with or without the fused ReLU
        self._dc_run = run
        self._dc_values = {}
        self._dc_requires_grad = bool(requires_grad)
        self._dc_inputs = ()

    def guard(self, *tensors) -> "DeferredActivation":
        """Remember the version counters of the layer's inputs (features, ``edge_index``, parameters): an in-place write to
        one of them between the conv call and the first use of its result would silently change what an eager conv had
        already computed - ``value`` raises instead."""
        self._dc_inputs = tuple((t, t._version) for t in tensors if isinstance(t, torch.Tensor))
        return self

    def value(self, relu: bool = False) -> torch.Tensor:
        """The conv's output (``relu``: with the activation fused), computed on first request."""
        v = self._dc_values.get(relu)
        if v is None:
            for t, ver in self._dc_inputs:
                if t._version != ver:
                    raise RuntimeError(
                        "deformcontact_amd: an input of conv(x, edge_index) (features, edge_index or a parameter) was "
                        "modified in place between the call and the first use of its result; the call is deferred by one "
                        "use so that a following F.relu runs fused (deferred.py) - use the result before modifying its "
                        "inputs, or set deformcontact_amd.nn.conv.DEFER_ACTIVATION = False")
            if self._dc_requires_grad and not torch.is_grad_enabled():
                # The call was made with gradients wanted, its first use is not recording.  Inside the forward of a custom
                # autograd Function (`Function.apply` does not dispatch on its arguments, so it took the wrapper for a
                # tensor that needs no gradient) the graph would be cut silently: raise.  Under a plain `torch.no_grad()`
                # block the layer runs WITH recording, as the eager call would have done at the call site.
                if _inside_function_apply():
                    raise RuntimeError(
                        "deformcontact_amd: the result of conv(x, edge_index) was passed straight to a custom "
                        "torch.autograd.Function; the call is deferred by one use (deferred.py) and Function.apply does not "
                        "dispatch on its arguments, so the gradient would not reach the conv - pass `y.value()` (or any "
                        "tensor computed from y), or set deformcontact_amd.nn.conv.DEFER_ACTIVATION = False")
                with torch.enable_grad():
                    v = self._dc_run(relu)
            else:
                v = self._dc_run(relu)
            self._dc_values[relu] = v
        return v

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _RELU_FUNCS and args and isinstance(args[0], cls):
            return args[0].value(True)
        name = _func_name(func)
        if name == "requires_grad" and len(args) == 1 and isinstance(args[0], cls):
            return args[0]._dc_requires_grad
        if name in _META and args and isinstance(args[0], cls) and not any(isinstance(a, cls) for a in args[1:]):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        return func(*_unwrap(args), **_unwrap(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # safety net: an ATen call that reached the wrapper without passing __torch_function__ gets the value
        return func(*_unwrap(args), **_unwrap(kwargs or {}))

    def __repr__(self):                                          # (printing a tensor is a use of its value)
        return repr(self.value(False))


def _inside_function_apply() -> bool:
    import sys
    f = sys._getframe(2)
    while f is not None:
        code = f.f_code
        if code.co_name == "apply" and code.co_filename.replace("\\", "/").endswith("torch/autograd/function.py"):
            return True
        f = f.f_back
    return False


def resolve(x):
    """``x`` itself, or - for the deferred result of a plain conv call - its value without activation: what every entry
    point of this package that hands tensors to a custom autograd Function calls first (``Function.apply`` does not
    dispatch on its arguments)."""
    return x.value(False) if isinstance(x, DeferredActivation) else x


def _unwrap(a):
    if isinstance(a, DeferredActivation):
        return a.value(False)
    if isinstance(a, (list, tuple)):
        return type(a)(_unwrap(v) for v in a)
    if isinstance(a, dict):
        return {k: _unwrap(v) for k, v in a.items()}
    return a


def deferred(run, n: int, width: int, like: torch.Tensor, requires_grad: bool) -> DeferredActivation:
    return DeferredActivation(run, (n, width), like.dtype, like.device, requires_grad)
