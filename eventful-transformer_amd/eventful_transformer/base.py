"""Module base class and op-count bookkeeping (API of the reference's eventful_transformer/base.py).

`ExtendedModule` gives every module of the package the three harness hooks the reference's callers
use (base.py:81-149 there): `reset()` before each clip, `counting()/no_counting()/clear_counts()/
total_counts()` for MAC accounting, and `modules_of_type(cls)` which `set_policies`
(utils/misc.py:140-143) walks to inject one policy object per gate.
"""
import operator
from collections import defaultdict
from sys import stdout

from torch import nn


def _format_key_values(mapping, fmt):
    return [(key, format(mapping[key], fmt)) for key in sorted(mapping.keys())]


def dict_csv_header(x):
    """Comma-joined, key-sorted header line for a dict of counters."""
    return ",".join(sorted(x.keys()))


def dict_csv_line(x):
    """Comma-joined values (general format) in the same key order as dict_csv_header."""
    return ",".join(value for _, value in _format_key_values(x, "g"))


def dict_string(x, indent=4, value_format=".4g"):
    """Aligned multi-line `key: value` rendering of a dict of counters."""
    width = max(len(str(key)) for key in x.keys()) + 1
    pad = " " * indent
    return "\n".join(f"{pad}{str(key) + ':':<{width}} {value}" for key, value in _format_key_values(x, value_format))


def numeric_tuple(x, length):
    """Scalar -> tuple of `length` copies; any other iterable -> tuple(x)."""
    if isinstance(x, (int, float, complex, bool)):
        return (x,) * length
    return tuple(x)


class Counts(defaultdict):
    """Operation counters: a defaultdict(int) closed under +, -, *, / with scalars and other Counts."""

    def __init__(self, *args, **kwargs):
        if args or kwargs:
            super().__init__(*args, **kwargs)
        else:
            super().__init__(int)

    def _with(self, other, op):
        out = self.copy()
        if isinstance(other, Counts):
            for key, value in other.items():
                out[key] = op(out[key], value)
        else:
            for key in out:
                out[key] = op(out[key], other)
        return out

    def __add__(self, other):
        return self._with(other, operator.add)

    __radd__ = __add__

    def __neg__(self):
        return self * -1

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, factor):
        out = self.copy()
        for key in out:
            out[key] *= factor
        return out

    __rmul__ = __mul__

    def __truediv__(self, divisor):
        return self * (1.0 / divisor)

    def csv_header(self):
        return dict_csv_header(self)

    def csv_line(self):
        return dict_csv_line(self)

    def pretty_print(self, indent=4, value_format=".3e", file=stdout, flush=False):
        print(dict_string(self, indent, value_format), file=file, flush=flush)


class ExtendedModule(nn.Module):
    """nn.Module + per-clip state reset, MAC counting, and typed sub-module enumeration."""

    def __init__(self):
        super().__init__()
        self.count_mode = False
        self.counts = Counts()

    # -- sub-module enumeration ------------------------------------------------------------------
    def modules_of_type(self, module_type):
        return (m for m in self.modules() if isinstance(m, module_type))

    def extended_modules(self):
        return self.modules_of_type(ExtendedModule)

    # -- counting --------------------------------------------------------------------------------
    def counting(self, mode=True):
        for m in self.extended_modules():
            m.count_mode = mode

    def no_counting(self):
        self.counting(mode=False)

    def clear_counts(self):
        for m in self.extended_modules():
            m.counts.clear()

    def total_counts(self):
        return sum(m.counts for m in self.extended_modules())

    # -- per-clip state --------------------------------------------------------------------------
    def reset(self):
        self.__dict__.pop("_evt_state_owner", None)   # (graphs.FrameGraphs: the per-clip state is no longer the one it captured)
        for m in self.extended_modules():
            m.reset_self()

    def reset_self(self):
        """Override to drop per-clip state held by this module itself."""
