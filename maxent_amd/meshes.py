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
        if cls.__init__ is not Mesh.__init__:
            # a mesh class written against the reference's protocol (doc/guide/customization.rst): its own
            # __init__ takes whatever it likes, calls the base __init__ with the named numbers and fills
            # self[:] -- here only the storage is made
            n = kwargs.get('n_points', args[2] if len(args) > 2 else cls._defaults.get('n_points', 100))
            return np.zeros(int(n)).view(cls)
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

    def __init__(self, *args, **kwargs):
        if type(self).__init__ is Mesh.__init__:
            return                                   # the built-in meshes are complete after __new__
        par = dict(zip(list(type(self)._defaults), args))
        par.update(kwargs)
        known = {k: v for k, v in par.items() if k in type(self)._defaults}
        type(self)._check(**known)
        self.__dict__.update(known)

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
