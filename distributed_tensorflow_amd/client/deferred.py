"""Results of a step that has been enqueued but not waited for.

With ``ConfigProto(deferred_fetches=True)`` a captured GPU step returns from ``Session.run`` as
soon as its HIP graph is launched. The small results of the step (loss, step counter, accuracy;
the merged summary built from them) come back as :class:`DeferredValue` objects. Such an object
stands for a numpy value that does not exist yet: the first time anything looks at it --
``float(v)``, ``int(v)``, ``"%f" % v``, ``print(v)``, ``np.asarray(v)``, arithmetic and
comparisons, any numpy function, any attribute (``v.shape``, ``v.item()``) -- it waits for the
step that produces it, takes the value out of the step's slot of the result ring and from then on
behaves as that value. A training loop in the reference's style

    _, cost, summary, step = sess.run([train_op, cross_entropy, summary_op, global_step], ...)
    writer.add_summary(summary, step)
    if count % frequency == 0: print("Cost: %.4f" % cost)

(``tf_distributed.py:111-122``) therefore waits for the GPU once per ``frequency`` steps instead
of once per step, while every value it does look at is exactly the value a blocking ``run`` would
have returned. Values are never lost: before a ring slot is reused (16 steps later) the session
copies the old step's results out of it.
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import numpy as np


def resolve(value: Any) -> Any:
    """``value`` itself, or the concrete value a :class:`DeferredValue` stands for."""
    if isinstance(value, DeferredValue):
        return value.resolve()
    return value


class DeferredValue(np.lib.mixins.NDArrayOperatorsMixin):
    __slots__ = ("_thunk", "_value", "_done")

    def __init__(self, thunk: Callable[[], Any]):
        self._thunk: Optional[Callable[[], Any]] = thunk
        self._value = None
        self._done = False

    def resolve(self):
        if not self._done:
            self._value = self._thunk()
            self._thunk = None               # drop the references the thunk holds
            self._done = True
        return self._value

    @property
    def ready(self) -> bool:
        """True once the value has been taken (never waits)."""
        return self._done

    # -- numpy protocol ----------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        arr = np.asarray(self.resolve())
        return arr.astype(dtype) if dtype is not None and arr.dtype != dtype else arr

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(resolve(x) for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(resolve(x) for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        def walk(x):
            if isinstance(x, DeferredValue):
                return x.resolve()
            if isinstance(x, (list, tuple)):
                return type(x)(walk(i) for i in x)
            return x
        return func(*walk(args), **{k: walk(v) for k, v in kwargs.items()})

    # -- Python scalars, formatting, containers --------------------------------------------------
    def __float__(self):
        return float(self.resolve())

    def __int__(self):
        return int(self.resolve())

    def __index__(self):
        return int(self.resolve())

    def __bool__(self):
        return bool(self.resolve())

    def __complex__(self):
        return complex(self.resolve())

    def __round__(self, ndigits=None):
        return round(float(self.resolve()), ndigits)

    def __format__(self, spec):
        return format(self.resolve(), spec)

    def __str__(self):
        return str(self.resolve())

    def __repr__(self):
        return repr(self.resolve())

    def __bytes__(self):
        return bytes(self.resolve())

    def __len__(self):
        return len(self.resolve())

    def __iter__(self):
        return iter(self.resolve())

    def __getitem__(self, key):
        return self.resolve()[key]

    def __hash__(self):
        return hash(self.resolve())

    def __getattr__(self, name):
        # only reached for names the proxy does not define itself
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.resolve(), name)
