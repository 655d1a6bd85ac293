"""TEST INFRASTRUCTURE — the reference's own (unmodified, byte-compiled) Python package over OUR shared library.

``oracle/build_ref.sh`` byte-compiles ``/root/reference/bitsandbytes/**/*.py`` into sourceless ``.pyc`` files under
``oracle/_ref/ref_py/`` (compiled output; no reference source is copied, and the directory is git-ignored). Unlike the
reference checkout those files travel to the GPU box, so INTEGRATION.md's mode A - reference host code, this native library
- can be run end to end on hardware. ``make_package(native_lib, lib_name)`` lays a throw-away package directory out the way
the reference's loader wants it (``bitsandbytes/cextension.py:36-57,348-377``: the native library sits inside the package
directory under a version-derived name): symlinks to the compiled modules plus one link to the library under test.
Only tests may use this.
"""
import os
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
COMPILED = os.path.join(HERE, "_ref", "ref_py", "bitsandbytes")


def compiled_reference_available() -> bool:
    return os.path.isfile(os.path.join(COMPILED, "__init__.pyc"))


def make_package(native_lib: str, lib_name: str) -> str:
    """Returns a directory to put on sys.path: <dir>/bitsandbytes/ = the compiled reference modules + native_lib as lib_name."""
    if not compiled_reference_available():
        raise RuntimeError("oracle/_ref/ref_py missing: run oracle/build_ref.sh where /root/reference exists")
    farm = tempfile.mkdtemp(prefix="bnb_mode_a_")
    for root, dirs, files in os.walk(COMPILED):
        rel = os.path.relpath(root, COMPILED)
        dst = os.path.join(farm, "bitsandbytes") if rel == "." else os.path.join(farm, "bitsandbytes", rel)
        os.makedirs(dst, exist_ok=True)
        for f in files:
            if f.endswith(".pyc"):
                os.symlink(os.path.join(root, f), os.path.join(dst, f))
    os.symlink(os.path.abspath(native_lib), os.path.join(farm, "bitsandbytes", lib_name))
    return farm
