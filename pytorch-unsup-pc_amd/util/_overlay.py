"""Fall-through for the overlaid modules: names this build does not define come from the reference's module of the
same file name, found in the other directories of the `util` package path (see __init__.py) and executed on first use."""
import importlib.util
import os
import sys


def fall_through(module_name, own_file):
    """Module-level __getattr__ (PEP 562) for util.<module_name>."""
    state = {}

    def shadowed():
        if "mod" not in state:
            state["mod"] = None
            pkg = sys.modules[__package__]
            short = module_name.rsplit(".", 1)[-1]
            for directory in pkg.__path__:
                cand = os.path.join(directory, short + ".py")
                if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(own_file):
                    spec = importlib.util.spec_from_file_location(__package__ + "._shadowed_" + short, cand)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)   # its own `from util.x import y` lines resolve through this overlay
                    state["mod"] = mod
                    break
        return state["mod"]

    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        mod = shadowed()
        if mod is None or not hasattr(mod, name):
            raise AttributeError("module %r has no attribute %r (not part of the MI355X renderer%s)"
                                 % (module_name, name, "" if mod is None else ", nor of the module it overlays"))
        return getattr(mod, name)

    return __getattr__
