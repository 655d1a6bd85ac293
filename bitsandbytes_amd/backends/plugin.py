"""Entry point for the reference's plug-in point (reference bitsandbytes/__init__.py:52-70): packages that declare an entry
point in the group ``bitsandbytes.backends`` are loaded and called when ``bitsandbytes`` is imported. pyproject.toml declares

    [project.entry-points."bitsandbytes.backends"]
    mi355x = "bitsandbytes_amd.backends.plugin:register"

so that on a box with the stock ``bitsandbytes`` package AND this package installed, ``import bitsandbytes`` ends with the HIP
("cuda" dispatch key) kernels of the 4-bit path provided by libbitsandbytes_mi355x.so: the reference's Python host code
(functional.py, nn.Linear4bit, autograd) then runs unchanged on top of them (INTEGRATION.md mode C)."""
REGISTERED = False


def register() -> None:
    global REGISTERED
    # importing the package registers the device kernels of the ``bitsandbytes::`` ops (backends/hip.py); op schemas that the
    # reference has already defined are left alone (_ops._define), kernels it has already registered for the same dispatch key
    # are replaced (_ops.register_kernel)
    import bitsandbytes_amd  # noqa: F401

    REGISTERED = True
