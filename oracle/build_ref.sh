#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds the reference's own CPU library from the sources where they lie
# under /root/reference (csrc/cpu_ops.cpp + csrc/pythonInterface.cpp, the two TUs the reference's
# CMakeLists.txt:63 lists for COMPUTE_BACKEND=cpu) into oracle/_ref/. No reference source is copied.
# Flags mirror CMakeLists.txt:361-405 (C++17, OpenMP, AVX512F/BW/DQ/VL/BF16, prefer-vector-width=256).
set -euo pipefail
REF=${BNB_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ ! -d "$REF/csrc" ]; then
  echo "build_ref: $REF/csrc not present (GPU box?) - keeping prebuilt $OUT" >&2
  exit 0
fi
g++ -O3 -std=c++17 -shared -fPIC -fopenmp -DHAS_OPENMP -DBUILD_CUDA=0 -DBUILD_HIP=0 -DBUILD_XPU=0 \
    -fno-semantic-interposition -fvisibility-inlines-hidden \
    -mavx512f -mavx512bw -mavx512dq -mavx512vl -mavx512bf16 \
    -mprefer-vector-width=256 -mfma -mavx2 -mf16c -mlzcnt -mbmi -mbmi2 \
    -I"$REF/csrc" "$REF/csrc/cpu_ops.cpp" "$REF/csrc/pythonInterface.cpp" \
    -o "$OUT/libbitsandbytes_cpu.so"
echo "build_ref: built $OUT/libbitsandbytes_cpu.so"

# The reference's Python package, BYTE-COMPILED (sourceless .pyc, nothing copied) into oracle/_ref/ref_py/: lets the
# GPU box - where /root/reference does not exist - run the reference's unmodified host code (bitsandbytes/functional.py,
# backends/cuda/ops.py, nn/modules.py ...) over OUR shared library: INTEGRATION.md mode A, tests/test_gpu_mode_a.py.
python3 - "$REF/bitsandbytes" "$OUT/ref_py/bitsandbytes" <<'PY'
import os, py_compile, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
shutil.rmtree(os.path.dirname(dst), ignore_errors=True)
n = 0
for root, dirs, files in os.walk(src):
    dirs[:] = [d for d in dirs if d != "__pycache__"]
    rel = os.path.relpath(root, src)
    out = dst if rel == "." else os.path.join(dst, rel)
    os.makedirs(out, exist_ok=True)
    for f in files:
        if f.endswith(".py"):
            py_compile.compile(os.path.join(root, f), cfile=os.path.join(out, f + "c"),
                               dfile=os.path.join("bitsandbytes", "" if rel == "." else rel, f), doraise=True)
            n += 1
print(f"build_ref: byte-compiled {n} reference modules into {os.path.dirname(dst)}", file=sys.stderr)
PY

# The reference's own TEST files, byte-compiled the same way (sourceless .pyc, nothing copied) into oracle/_ref/ref_tests/:
# tests/test_gpu_reference_suite.py runs them on the GPU box against this repository's HIP kernels.
python3 - "$REF/tests" "$OUT/ref_tests" <<'PY'
import os, py_compile, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
names = ["__init__", "conftest", "helpers", "test_ops", "test_functional", "test_autograd", "test_linear4bit", "test_parametrize", "test_modules"]
for n in names:
    py_compile.compile(os.path.join(src, n + ".py"), cfile=os.path.join(dst, n + ".pyc"), dfile=os.path.join("tests", n + ".py"), doraise=True)
print(f"build_ref: byte-compiled {len(names)} reference test modules into {dst}", file=sys.stderr)
PY
