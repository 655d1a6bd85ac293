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
