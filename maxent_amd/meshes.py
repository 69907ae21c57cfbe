"""One array type for the omega and alpha grids.

A mesh is a float ndarray that carries a few named numbers (its range, its
size, a shape parameter) through views, slices and pickling.  A concrete mesh
class only says how its points are generated: ``_points(**parameters)`` returns
the values, the parameters become attributes of the array.
"""

import inspect

import numpy as np


class Mesh(np.ndarray):
    _defaults = {}            # parameter name -> default, in positional order

    @classmethod
    def _points(cls, **par):
        raise NotImplementedError('use a concrete mesh class')

    @classmethod
    def _check(cls, **par):
        pass

    def __new__(cls, *args, **kwargs):
        names = list(cls._defaults)
        if len(args) > len(names):
            raise TypeError('{} takes at most {} arguments'.format(cls.__name__, len(names)))
        par = dict(cls._defaults)
        par.update(zip(names, args))
        for k, v in kwargs.items():
            if k not in par:
                raise TypeError('{} got an unexpected argument {!r}'.format(cls.__name__, k))
            par[k] = v
        cls._check(**par)
        values, attrs = cls._points(**par)
        self = np.array(values, dtype=float).view(cls)
        self.__dict__.update(attrs)
        return self

    def __array_finalize__(self, parent):
        if isinstance(parent, Mesh):
            self.__dict__.update({k: v for k, v in parent.__dict__.items() if not k.startswith('_')})

    # pickling keeps the attributes (ndarray's own reduce drops the instance dict)
    def __reduce__(self):
        fn, args, state = super(Mesh, self).__reduce__()
        return fn, args, (state, {k: v for k, v in self.__dict__.items() if not k.startswith('_')})

    def __setstate__(self, state):
        super(Mesh, self).__setstate__(state[0])
        self.__dict__.update(state[1])


def signature_of(cls):
    """for documentation tools: the constructor signature of a mesh class"""
    return inspect.Signature([inspect.Parameter(k, inspect.Parameter.POSITIONAL_OR_KEYWORD, default=v)
                              for k, v in cls._defaults.items()])
