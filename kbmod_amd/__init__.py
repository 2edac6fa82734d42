"""kbmod_amd -- MI355X-native shift-and-stack trajectory search.

``kbmod_amd.search`` is the compiled host layer (pybind11) with the same
Python-visible surface as the reference's ``kbmod.search``; it drives
``lib/libkbmod_hip.so`` (hand-written HIP kernels for gfx950, C ABI declared in
``include/kbmod_hip.h``).  Both are built in-tree by ``kbmod_amd.build`` /
``__graft_entry__.build()``.  Importing ``kbmod_amd.search`` fails loudly when
the native artefacts are missing: there is no Python or CPU fallback for the
device path.
"""

__all__ = ["search", "fake_data", "build"]


def __getattr__(name):
    if name == "search":
        import importlib

        try:
            return importlib.import_module("kbmod_amd.search")
        except ImportError as exc:  # pragma: no cover
            raise ImportError(
                "kbmod_amd.search (the native HIP/pybind11 module) is not built; run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `python -m kbmod_amd.build`"
            ) from exc
    raise AttributeError(name)
