"""TEST INFRASTRUCTURE — import the *real* reference Python package in this container.

The reference (``/root/reference/bitsandbytes``) is read-only and its loader
(``bitsandbytes/cextension.py:348-377``) wants ``libbitsandbytes_cpu.so`` inside the package
directory, so we build a throw-away *symlink farm* in a temp dir: every reference file is a symlink
(nothing is copied), plus a link to ``oracle/_ref/libbitsandbytes_cpu.so`` built by
``oracle/build_ref.sh``. Only usable where ``/root/reference`` exists (never on the GPU box);
only golden-vector generation and oracle-pinning tests may call this.
"""
import importlib
import os
import sys
import tempfile

REF_DIR = os.environ.get("BNB_REFERENCE_DIR", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libbitsandbytes_cpu.so")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_DIR, "bitsandbytes")) and os.path.isfile(REF_LIB)


_cached = None


def import_reference():
    """Returns the reference ``bitsandbytes`` module (CPU backend, native lib loaded)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference checkout or oracle/_ref/libbitsandbytes_cpu.so missing; run oracle/build_ref.sh")
    farm = tempfile.mkdtemp(prefix="bnb_ref_farm_")
    src_root = os.path.join(REF_DIR, "bitsandbytes")
    for root, dirs, files in os.walk(src_root):
        rel = os.path.relpath(root, src_root)
        dst = os.path.join(farm, "bitsandbytes", rel) if rel != "." else os.path.join(farm, "bitsandbytes")
        os.makedirs(dst, exist_ok=True)
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        for f in files:
            if f.endswith((".py", ".json", ".txt")):
                os.symlink(os.path.join(root, f), os.path.join(dst, f))
    os.symlink(REF_LIB, os.path.join(farm, "bitsandbytes", "libbitsandbytes_cpu.so"))
    sys.dont_write_bytecode = True
    sys.path.insert(0, farm)
    try:
        mod = importlib.import_module("bitsandbytes")
    finally:
        sys.path.remove(farm)
    _cached = mod
    return mod
