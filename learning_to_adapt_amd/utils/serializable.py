"""Constructor-argument pickling, the contract of ``learning_to_adapt/utils/serializable.py``.

The reference trainer snapshots ``policy`` / ``dynamics_model`` with ``joblib.dump``
(``trainers/mb_trainer.py:105-108``); objects are rebuilt by calling their constructor
again with the recorded arguments.  Subclasses call ``Serializable.quick_init(self,
locals())`` first thing in ``__init__`` - same call shape as the reference so that
drop-in subclasses keep working.
"""

import inspect


class Serializable(object):
    def __init__(self, *args, **kwargs):
        self.__args = args
        self.__kwargs = kwargs

    def quick_init(self, locals_):
        if getattr(self, "_serializable_initialized", False):
            return
        spec = inspect.getfullargspec(self.__init__)
        kwargs = locals_[spec.varkw] if spec.varkw else dict()
        varargs = locals_[spec.varargs] if spec.varargs else tuple()
        positional = [locals_[name] for name in spec.args][1:]   # drop self
        self.__args = tuple(positional) + tuple(varargs)
        self.__kwargs = kwargs
        self._serializable_initialized = True

    def __getstate__(self):
        return {"__args": self.__args, "__kwargs": self.__kwargs}

    def __setstate__(self, d):
        fresh = type(self)(*d["__args"], **d["__kwargs"])
        self.__dict__.update(fresh.__dict__)

    @classmethod
    def clone(cls, obj, **overrides):
        assert isinstance(obj, Serializable)
        d = obj.__getstate__()
        names = inspect.getfullargspec(obj.__init__).args[1:]
        args = list(d["__args"])
        kwargs = dict(d["__kwargs"])
        for key, val in overrides.items():
            if key in names:
                args[names.index(key)] = val
            else:
                kwargs[key] = val
        out = type(obj).__new__(type(obj))
        out.__setstate__({"__args": tuple(args), "__kwargs": kwargs})
        return out
