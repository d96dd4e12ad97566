"""Pickling by constructor arguments - the contract of ``learning_to_adapt/utils/serializable.py``.

The reference trainer snapshots ``policy`` / ``dynamics_model`` with ``joblib.dump``
(``trainers/mb_trainer.py:105-108``); such objects are rebuilt by calling their constructor with the
arguments recorded at construction.  A class opts in by calling ``Serializable.quick_init(self,
locals())`` as the first statement of ``__init__`` (same call shape as the reference, so classes
written against it keep working); the pickled state uses the reference's keys ``__args`` /
``__kwargs``.
"""

import inspect


class Serializable(object):
    _ctor_record = None

    def quick_init(self, frame_locals):
        """Record the arguments of the outermost constructor call (later calls are ignored)."""
        if self.__dict__.get("_ctor_record") is not None:
            return
        positional, keyword = [], {}
        parameters = list(inspect.signature(type(self).__init__).parameters.values())[1:]
        for par in parameters:
            if par.kind is inspect.Parameter.VAR_POSITIONAL:
                positional.extend(frame_locals[par.name])
            elif par.kind is inspect.Parameter.VAR_KEYWORD:
                keyword.update(frame_locals[par.name])
            else:
                positional.append(frame_locals[par.name])
        self._ctor_record = (tuple(positional), keyword)

    def _recorded(self):
        if self._ctor_record is None:
            raise RuntimeError("%s.__init__ never called Serializable.quick_init" % type(self).__name__)
        return self._ctor_record

    def __getstate__(self):
        args, kwargs = self._recorded()
        return {"__args": args, "__kwargs": kwargs}

    def __setstate__(self, state):
        rebuilt = type(self)(*state["__args"], **state["__kwargs"])
        self.__dict__.update(rebuilt.__dict__)

    @classmethod
    def clone(cls, obj, **overrides):
        """A copy of ``obj`` built with some constructor arguments replaced."""
        args, kwargs = obj._recorded()
        names = [par.name for par in list(inspect.signature(type(obj).__init__).parameters.values())[1:]]
        args, kwargs = list(args), dict(kwargs)
        for key, value in overrides.items():
            if key in names and names.index(key) < len(args):
                args[names.index(key)] = value
            else:
                kwargs[key] = value
        twin = type(obj).__new__(type(obj))
        twin.__setstate__({"__args": tuple(args), "__kwargs": kwargs})
        return twin
