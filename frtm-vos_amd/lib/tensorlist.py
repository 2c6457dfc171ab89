"""TensorList: a list of tensors with element-wise algebra (API of the reference's lib/tensorlist.py:5-207).

Used as the vector type of the solver's public surface (GaussNewtonCG.x / .p / .b ...).  The
arithmetic of the CG loop itself never goes through this class on the hot path: it runs in the
fused HIP vector kernels (csrc/target_model.hip) on one flat device buffer.

Operators are generated from a table instead of being spelled out one by one; attribute
fan-out never answers for dunder names (the upstream class does, which breaks
torch.autograd on torch >= 1.7).
"""
import functools
import operator

import torch


def _pairwise(a, b):
    return isinstance(b, (TensorList, list))


class TensorList(list):

    def __init__(self, tensors=None):
        super().__init__(tensors if tensors is not None else [])

    def __getitem__(self, item):
        if isinstance(item, int):
            return list.__getitem__(self, item)
        if isinstance(item, (tuple, list)):
            return TensorList(list.__getitem__(self, i) for i in item)
        return TensorList(list.__getitem__(self, item))

    # -- list helpers -------------------------------------------------------------------
    def concat(self, other):
        return TensorList(list.__add__(self, other))

    def copy(self):
        return TensorList(list.copy(self))

    def unroll(self):
        flat = TensorList()
        for t in self:
            if isinstance(t, TensorList):
                flat.extend(t.unroll())
            else:
                flat.append(t)
        return flat

    def list(self):
        return list(self)

    def attribute(self, attr, *args):
        return TensorList(getattr(e, attr, *args) for e in self)

    def apply(self, fn):
        return TensorList(fn(e) for e in self)

    def __getattr__(self, name):
        if name.startswith('__') or not hasattr(torch.Tensor, name):
            raise AttributeError("'TensorList' object has no attribute '%s'" % name)

        def fan_out(*args, **kwargs):
            return TensorList(getattr(e, name)(*args, **kwargs) for e in self)

        return fan_out

    def __pos__(self):
        return TensorList(+e for e in self)

    def __neg__(self):
        return TensorList(-e for e in self)


def _install_operators():
    table = {'add': operator.add, 'sub': operator.sub, 'mul': operator.mul, 'truediv': operator.truediv,
             'matmul': operator.matmul, 'mod': operator.mod, 'le': operator.le, 'ge': operator.ge}
    inplace = {'add': operator.iadd, 'sub': operator.isub, 'mul': operator.imul, 'truediv': operator.itruediv,
               'matmul': operator.imatmul}

    def forward(op):
        def f(self, other):
            if _pairwise(self, other):
                return TensorList(op(a, b) for a, b in zip(self, other))
            return TensorList(op(a, other) for a in self)
        return f

    def reflected(op):
        def f(self, other):
            if _pairwise(self, other):
                return TensorList(op(b, a) for a, b in zip(self, other))
            return TensorList(op(other, a) for a in self)
        return f

    def in_place(op):
        def f(self, other):
            if _pairwise(self, other):
                for i, b in enumerate(other):
                    self[i] = op(self[i], b)
            else:
                for i in range(len(self)):
                    self[i] = op(self[i], other)
            return self
        return f

    for name, op in table.items():
        setattr(TensorList, '__%s__' % name, forward(op))
        if name not in ('le', 'ge'):
            setattr(TensorList, '__r%s__' % name, reflected(op))
    for name, op in inplace.items():
        setattr(TensorList, '__i%s__' % name, in_place(op))


_install_operators()


def tensor_operation(op):
    """Decorator: lets a tensor function map over TensorList operands (reference tensorlist.py:183-207)."""

    @functools.wraps(op)
    def mapped(*args, **kwargs):
        if not args:
            raise ValueError('Must be at least one argument without keyword (i.e. operand).')
        first = isinstance(args[0], TensorList)
        second = len(args) > 1 and isinstance(args[1], TensorList)
        if first and second:
            return TensorList(op(a, b, *args[2:], **kwargs) for a, b in zip(args[0], args[1]))
        if first:
            return TensorList(op(a, *args[1:], **kwargs) for a in args[0])
        if second:
            return TensorList(op(args[0], b, *args[2:], **kwargs) for b in args[1])
        return op(*args, **kwargs)

    return mapped
