"""Stand-in for `plum.dispatch` (plum-dispatch is not installed in the build image; the reference's
algos/emlp_torch/reps/representation.py:7 imports it for four overloads of `mul_reps`).  BUILD CONTAINER
ONLY: tools/gen_golden.py puts this directory on sys.path to import the reference's shipped TD3-EMLP actors;
it never travels to the GPU box (.gpurunignore) and nothing in the product imports it.

What plum does and this restates: `@dispatch` registers a function under the types of its positional
parameters' annotations (un-annotated = object); `@dispatch.multi(sig, ...)` registers one function under
several signatures; a call picks, among the registered signatures every argument is an instance of, the most
specific one (component-wise subclass), and raises on none / ambiguity.  That the stand-in resolves the
reference's calls as plum does is not taken on faith: the actors built through it must reproduce the
reference-owned flight log's action columns (tools/gen_golden.py: check_shipped_actors_against_flightlog).
"""
import inspect


class _Function:
    def __init__(self, name):
        self.name, self.methods = name, []

    def register(self, sig, fn):
        self.methods = [(s, f) for s, f in self.methods if s != sig] + [(sig, fn)]

    def __call__(self, *args):
        cands = [(s, f) for s, f in self.methods if len(s) == len(args) and all(isinstance(a, t) for a, t in zip(args, s))]
        if not cands:
            raise LookupError(f"{self.name}: no method for {tuple(type(a).__name__ for a in args)}")
        best = [(s, f) for s, f in cands if all(all(issubclass(x, y) for x, y in zip(s, s2)) for s2, _ in cands)]
        if len(best) != 1:
            raise LookupError(f"{self.name}: ambiguous call for {tuple(type(a).__name__ for a in args)}")
        return best[0][1](*args)


class _Dispatcher:
    def __init__(self):
        self._functions = {}

    def _function(self, fn):
        key = (fn.__module__, fn.__qualname__)
        if key not in self._functions:
            self._functions[key] = _Function(fn.__qualname__)
        return self._functions[key]

    def __call__(self, fn):
        params = list(inspect.signature(fn).parameters.values())
        sig = tuple(object if p.annotation is inspect.Parameter.empty else p.annotation for p in params)
        f = self._function(fn)
        f.register(sig, fn)
        return f

    def multi(self, *signatures):
        def deco(fn):
            f = self._function(fn)
            for sig in signatures:
                f.register(tuple(sig), fn)
            return f
        return deco


dispatch = _Dispatcher()
