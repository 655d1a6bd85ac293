// bitsandbytes::gemm_4bit registered from C++ for the HIP ("CUDA" dispatch key) device: the host side of one
// Linear4bit.forward without a Python frame between the dispatcher and the C ABI.
//
// The Python kernel of the same op (bitsandbytes_amd/backends/hip.py, the counterpart of the reference's
// bitsandbytes/backends/cuda/ops.py:921-982) costs ~10 us per call in eager mode for a ~4 us kernel (profiles/r1_host_overhead.txt):
// argument boxing, a Python frame, ctypes marshalling of 19 arguments. This translation unit does the same glue - validation,
// contiguity, output allocation from torch's caching allocator, current raw stream, device guard, split-K scratch, routing
// between the fused kernels and dequantize + hipBLASLt - in C++ and calls the UNCHANGED C ABI of libbitsandbytes_mi355x.so
// (include/bnb_mi355x.h). PyTorch is plumbing here: no arithmetic of the path happens in this file except the reference's own
// large-batch strategy (dequantize once, then at::linear; reference backends/cuda/ops.py:904-916).
//
// Built by the Makefile with g++ (no device code) into bitsandbytes_amd/libbitsandbytes_mi355x_torch.so and loaded with
// torch.ops.load_library by backends/hip.py, which then leaves the op's "cuda" kernel to this library.
#include <Python.h>

#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <unordered_map>

#include "../../include/bnb_mi355x.h"

namespace {

// keep in step with bitsandbytes_amd/backends/hip.py (FUSED_MAX_M, FUSED_MAX_M_LONG_ROWS, FUSED_MAX_M_SQUARE, FUSED_TALL_WEIGHTS,
// STREAM_ONLY_MAX_M, SM_TAIL_MAX_M, SM_MIN_ROWS, _REFERENCE_CUSTOM_MAX_M, fused_max_m, _gemm_4bit_route - the measurements behind the numbers are quoted there);
// the GPU test tests/test_gpu_parity.py::test_native_dispatch_matches_python_kernel runs both over fused and unfused shapes,
// tests/test_cabi.py pins the constants and the function against the Python twin
constexpr int64_t kFusedMaxM = 512;
constexpr int64_t kFusedMaxMLongRows = 1024; // K >= 2 N
constexpr int64_t kFusedMaxMSquare = 640;    // 10 K >= 7 N
constexpr int64_t kFusedTallWeights = 50331648; // 48 << 20
constexpr int64_t kStreamOnlyMaxM = 16;
constexpr int64_t kSmTailMaxM = 128;  // rows that are not whole 256-k chunks: the streaming MFMA kernel's row passes (>= kSmMinRows rows)
constexpr int64_t kSmMinRows = 128;
constexpr int64_t kFusedMaxMBs32 = 128; // blocksize 32, plain statistics: the register-transposed kernel's row passes
constexpr int64_t kFusedMaxMFp32 = 4;
constexpr int64_t kReferenceCustomMaxM = 256; // reference backends/cuda/ops.py:816

// Largest batch the fused 16-bit kernels are used for on an N x K weight (backends/hip.py: fused_max_m)
int64_t fused_max_m(int64_t N, int64_t K, int64_t blocksize, bool nested) {
    if (K % 256 != 0 && K % 64 == 0 && blocksize >= 64 && N >= kSmMinRows)
        return kSmTailMaxM; // the streaming MFMA kernel's row passes
    if (K % 256 != 0 || blocksize < 32 || (blocksize == 32 && nested))
        return kStreamOnlyMaxM; // the MFMA kernels do not serve the call: the streaming kernel's 4-row passes
    if (blocksize == 32)
        return kFusedMaxMBs32;
    // (the extended ranges were measured on the K-quarter kernel: nested statistics only at blocksize 64 and K <= 16384)
    if (N * K <= kFusedTallWeights && blocksize >= 64 && (!nested || (blocksize == 64 && K <= 16384))) {
        if (K >= 2 * N)
            return kFusedMaxMLongRows;
        if (10 * K >= 7 * N)
            return kFusedMaxMSquare;
    }
    return kFusedMaxM;
}

int dtype_code(at::ScalarType t) {
    switch (t) {
    case at::kFloat:
        return 0;
    case at::kHalf:
        return 1;
    case at::kBFloat16:
        return 2;
    default:
        TORCH_CHECK(false, "unsupported dtype ", t);
    }
}

int quant_code(c10::string_view quant_type) {
    if (quant_type == "fp4")
        return 1;
    if (quant_type == "nf4")
        return 2;
    TORCH_CHECK(false, "quant_type must be 'nf4' or 'fp4', got '", std::string(quant_type), "'");
}

const void* ptr(const std::optional<at::Tensor>& t) { return t.has_value() ? t->const_data_ptr() : nullptr; }

void check_c_int(int64_t n, const char* what) {
    // the reference ABI carries element counts as C int (csrc/pythonInterface.cpp:346-444)
    TORCH_CHECK_VALUE(n <= std::numeric_limits<int>::max(), what, ": ", n, " elements exceed the C-ABI limit of 2**31 - 1");
}

// The reference raises this as a Python UserWarning from the op's kernel (backends/cuda/ops.py:956-962). TORCH_WARN from a
// library loaded with torch.ops.load_library only reaches stderr (no Python warning handler is installed around the call, and
// the GIL is released), so the warning is raised through the interpreter directly when there is one.
void warn_user(const std::string& msg) {
    if (!Py_IsInitialized()) {
        TORCH_WARN(msg);
        return;
    }
    const PyGILState_STATE gil = PyGILState_Ensure();
    const int rc = PyErr_WarnEx(PyExc_UserWarning, msg.c_str(), 1);
    if (rc < 0)
        PyErr_Clear(); // warnings turned into errors by a filter: reported as the op's error below
    PyGILState_Release(gil);
    TORCH_CHECK(rc >= 0, msg);
}

void dequantize_4bit_into(const at::Tensor& B, const at::Tensor& absmax, int64_t blocksize, int qt, at::Tensor& W, hipStream_t stream) {
    check_c_int(W.numel(), "dequantize_4bit");
    const bool nf4 = qt == 2;
    auto* packed = static_cast<unsigned char*>(B.data_ptr());
    auto* am = static_cast<float*>(absmax.data_ptr());
    const int bs = static_cast<int>(blocksize), n = static_cast<int>(W.numel());
    switch (W.scalar_type()) {
    case at::kFloat:
        (nf4 ? cdequantize_blockwise_fp32_nf4 : cdequantize_blockwise_fp32_fp4)(nullptr, packed, am, static_cast<float*>(W.data_ptr()), bs, n, stream);
        break;
    case at::kHalf:
        (nf4 ? cdequantize_blockwise_fp16_nf4 : cdequantize_blockwise_fp16_fp4)(nullptr, packed, am, W.data_ptr(), bs, n, stream);
        break;
    default:
        (nf4 ? cdequantize_blockwise_bf16_nf4 : cdequantize_blockwise_bf16_fp4)(nullptr, packed, am, W.data_ptr(), bs, n, stream);
        break;
    }
}

at::Tensor gemm_4bit_hip(const at::Tensor& A_in, const at::Tensor& B_in, at::IntArrayRef shapeB, const at::Tensor& absmax_in, int64_t blocksize,
                         c10::string_view quant_type, const std::optional<at::Tensor>& bias_in, const std::optional<at::Tensor>& absmax_8bit_in,
                         const std::optional<at::Tensor>& absmax_code_in, const std::optional<at::Tensor>& absmax_offset_in) {
    TORCH_CHECK(shapeB.size() == 2, "shapeB must be [N, K]");
    const int64_t K = A_in.dim() > 0 ? A_in.size(-1) : 0;
    const int64_t M = K ? A_in.numel() / K : 0;
    const int64_t N = shapeB[0];
    const int dt = dtype_code(A_in.scalar_type());
    const int qt = quant_code(quant_type);
    TORCH_CHECK(K == shapeB[1], "A inner dim (", K, ") does not match weight (", shapeB[1], ")");
    TORCH_CHECK(absmax_in.scalar_type() == at::kFloat, "absmax must be float32, got ", absmax_in.scalar_type());
    std::optional<at::Tensor> bias;
    if (bias_in.has_value()) {
        TORCH_CHECK(bias_in->dim() == 1, "bias must be 1D, got ", bias_in->dim(), "D");
        TORCH_CHECK(bias_in->scalar_type() == A_in.scalar_type(), "bias dtype (", bias_in->scalar_type(), ") must match A dtype (",
                    A_in.scalar_type(), ")");
        bias = bias_in->contiguous();
    }
    // (PyTorch-ROCm presents HIP devices under the "cuda" device type: its guard / stream classes for that are these)
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(std::optional<c10::Device>(A_in.device()));
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(A_in.get_device()).stream();
    const at::Tensor A = A_in.contiguous();
    const at::Tensor B = B_in.contiguous();
    const at::Tensor absmax = absmax_in.contiguous();

    // ---- MI355X routing (backends/hip.py:_gemm_4bit_route; replaces reference backends/cuda/ops.py:814-843,921-962)
    bool fused;
    if (blocksize <= 0 || K % blocksize != 0) {
        if (M <= kReferenceCustomMaxM)
            warn_user(c10::str("inner dimension (", K, ") is not aligned for fast kernel with blocksize=", blocksize,
                               ", falling back to slower implementation."));
        fused = false;
    } else {
        fused = M <= (dt == 0 ? kFusedMaxMFp32 : fused_max_m(N, K, blocksize, absmax_8bit_in.has_value()));
    }

    if (!fused) {
        // dequantize once + library GEMM: the reference's own strategy for these batches (backends/cuda/ops.py:904-916)
        at::Tensor W = at::empty(shapeB, A.options());
        if (absmax_8bit_in.has_value()) {
            // nested statistics: reconstructed inside the ONE dequantize launch (csrc/dequantize4.hip, NESTED) with the two roundings of
            // the host-side sequence dequantize_blockwise, += offset (which this branch ran as three launches until round 5)
            TORCH_CHECK(absmax_code_in.has_value() && absmax_offset_in.has_value(), "nested absmax needs absmax_code and absmax_offset");
            const at::Tensor a8 = absmax_8bit_in->contiguous();
            const at::Tensor code = absmax_code_in->to(at::kFloat).contiguous();
            const at::Tensor offset = absmax_offset_in->to(at::kFloat);
            TORCH_CHECK(a8.scalar_type() == at::kByte, "A must be uint8, got ", a8.scalar_type());
            TORCH_CHECK(absmax.scalar_type() == at::kFloat, "absmax must be float32, got ", absmax.scalar_type());
            const int64_t n = W.numel();
            const int64_t blocks = (n + blocksize - 1) / blocksize;
            TORCH_CHECK(a8.numel() == blocks && absmax.numel() == (blocks + 255) / 256 && code.numel() == 256 && offset.numel() == 1,
                        "nested statistics do not match the weight: ", a8.numel(), " codes, ", absmax.numel(), " second-level values for ", blocks, " blocks");
            bnb_mi355x_dequantize_4bit_nested(dt, static_cast<const unsigned char*>(B.const_data_ptr()), static_cast<const unsigned char*>(a8.const_data_ptr()),
                                              static_cast<const float*>(absmax.const_data_ptr()), static_cast<const float*>(code.const_data_ptr()),
                                              static_cast<const float*>(offset.const_data_ptr()), W.data_ptr(), static_cast<int>(blocksize), n, qt, stream);
        } else {
            dequantize_4bit_into(B, absmax, blocksize, qt, W, stream);
        }
        return at::linear(A, W, bias);
    }

    check_c_int(M, "gemm_4bit M");
    check_c_int(N, "gemm_4bit N");
    check_c_int(K, "gemm_4bit K");
    auto out_sizes = A.sizes().vec();
    out_sizes.back() = N;
    at::Tensor out = at::empty(out_sizes, A.options());
    std::optional<at::Tensor> a8, code, offset;
    if (absmax_8bit_in.has_value())
        a8 = absmax_8bit_in->contiguous();
    if (absmax_code_in.has_value())
        code = absmax_code_in->contiguous();
    if (absmax_offset_in.has_value())
        offset = absmax_offset_in->to(at::kFloat);
    // split-K scratch for the MFMA kernels comes from torch's caching allocator: stream-ordered and legal under hipGraph
    // capture (the library never has to allocate). M <= 2 always runs the streaming kernel, which needs none.
    at::Tensor ws;
    size_t ws_bytes = 0;
    if (M > 2)
        ws_bytes = bnb_mi355x_gemm_4bit_workspace_bytes(0, dt, static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), static_cast<int>(blocksize));
    if (ws_bytes)
        ws = at::empty({static_cast<int64_t>(ws_bytes)}, A.options().dtype(at::kByte));
    bnb_mi355x_gemm_4bit(0, dt, A.const_data_ptr(), static_cast<const uint8_t*>(B.const_data_ptr()), static_cast<const float*>(absmax.const_data_ptr()),
                         static_cast<const uint8_t*>(ptr(a8)), static_cast<const float*>(ptr(code)), static_cast<const float*>(ptr(offset)), nullptr,
                         out.data_ptr(), ptr(bias), static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), static_cast<int>(blocksize), qt,
                         ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream);
    return out;
}

// ---- Linear4bit.forward without the Python layers between the module and the C ABI (eager decode is host-bound: 10.5 us per
// Linear4bit call for a 4.1 us kernel, profiles/r2_host_overhead.txt). A module prepares its call ONCE - packed weight,
// statistics, bias already cast, dtype policy (reference nn/modules.py:609-637, autograd/_functions.py:407-491) - and gets a
// handle; every later forward is one two-argument op call.
struct Prepared {
    at::Tensor B, absmax;
    std::optional<at::Tensor> bias, a8, code, offset;
    std::vector<int64_t> shape;
    int64_t blocksize;
    std::string quant_type;
    std::optional<at::ScalarType> compute_dtype;
};
std::mutex g_prepared_mutex;
std::unordered_map<int64_t, std::shared_ptr<const Prepared>> g_prepared;
int64_t g_prepared_next = 1;

int64_t linear4bit_prepare(const at::Tensor& B, at::IntArrayRef shapeB, const at::Tensor& absmax, int64_t blocksize, c10::string_view quant_type,
                           const std::optional<at::Tensor>& bias, const std::optional<at::Tensor>& absmax_8bit,
                           const std::optional<at::Tensor>& absmax_code, const std::optional<at::Tensor>& absmax_offset,
                           std::optional<at::ScalarType> compute_dtype) {
    TORCH_CHECK(shapeB.size() == 2, "shapeB must be [N, K]");
    (void)quant_code(quant_type);
    auto p = std::make_shared<Prepared>();
    p->B = B.contiguous();
    p->absmax = absmax.contiguous();
    p->shape = shapeB.vec();
    p->blocksize = blocksize;
    p->quant_type = std::string(quant_type);
    p->compute_dtype = compute_dtype;
    // The bias is kept as handed over - an alias of the module's storage, never a converted copy: an in-place update of the
    // live bias is seen by the next call (the cast to the compute dtype, when one is needed, happens per call below); a
    // REPLACED bias storage changes data_ptr(), which the module keys the handle on.
    if (bias.has_value())
        p->bias = bias->contiguous();
    if (absmax_8bit.has_value())
        p->a8 = absmax_8bit->contiguous();
    if (absmax_code.has_value())
        p->code = absmax_code->contiguous();
    if (absmax_offset.has_value())
        p->offset = absmax_offset->to(at::kFloat);
    const std::lock_guard<std::mutex> lock(g_prepared_mutex);
    const int64_t id = g_prepared_next++;
    g_prepared.emplace(id, std::move(p));
    return id;
}

void linear4bit_release(int64_t handle) {
    const std::lock_guard<std::mutex> lock(g_prepared_mutex);
    g_prepared.erase(handle);
}

at::Tensor linear4bit_prepared(const at::Tensor& x, int64_t handle) {
    std::shared_ptr<const Prepared> p;
    {
        const std::lock_guard<std::mutex> lock(g_prepared_mutex);
        const auto it = g_prepared.find(handle);
        TORCH_CHECK(it != g_prepared.end(), "linear4bit_prepared: unknown handle ", handle);
        p = it->second;
    }
    // dtype policy of Linear4bit.forward: compute in compute_dtype, return in the input's dtype; a bias of another dtype is
    // cast here, per call
    const at::ScalarType inp = x.scalar_type();
    const at::Tensor xc = (p->compute_dtype.has_value() && *p->compute_dtype != inp) ? x.to(*p->compute_dtype) : x;
    std::optional<at::Tensor> bias = p->bias;
    if (bias.has_value() && bias->scalar_type() != xc.scalar_type())
        bias = bias->to(xc.scalar_type());
    at::Tensor y = gemm_4bit_hip(xc, p->B, p->shape, p->absmax, p->blocksize, p->quant_type, bias, p->a8, p->code, p->offset);
    return y.scalar_type() == inp ? y : y.to(inp);
}

// [layer(x) for layer in layers] for prepared layers that consume the same input (Q/K/V, gate/up): ONE native call, and - where the
// library takes the group as one launch (bnb_mi355x_gemm_4bit_grouped_route: the streaming kernel at one row, the streaming MFMA kernel
// from two rows on) - ONE kernel launch into one allocation. Eager decode is host-bound: the Python path of a grouped call (lists,
// ctypes arrays, one op call per fallback member) costs more than the launch it saves. Anything the grouped launch does not cover
// (mixed dtypes / statistics / blocksizes, a group the library would issue matrix by matrix) is the layers' prepared calls one by one.
std::vector<at::Tensor> linear4bit_group_prepared(const at::Tensor& x, at::IntArrayRef handles) {
    std::vector<std::shared_ptr<const Prepared>> ps;
    {
        const std::lock_guard<std::mutex> lock(g_prepared_mutex);
        for (const int64_t h : handles) {
            const auto it = g_prepared.find(h);
            TORCH_CHECK(it != g_prepared.end(), "linear4bit_group_prepared: unknown handle ", h);
            ps.push_back(it->second);
        }
    }
    const auto one_by_one = [&]() {
        std::vector<at::Tensor> ys;
        for (const int64_t h : handles)
            ys.push_back(linear4bit_prepared(x, h));
        return ys;
    };
    const size_t count = ps.size();
    if (count == 0)
        return {};
    const Prepared& p0 = *ps[0];
    const int64_t K = x.dim() > 0 ? x.size(-1) : 0;
    const int64_t M = K ? x.numel() / K : 0;
    bool same = count <= 8 && x.is_cuda() && M > 0 && p0.blocksize > 0 && K % p0.blocksize == 0;
    int64_t total = 0;
    for (const auto& p : ps) {
        same = same && p->compute_dtype == p0.compute_dtype && p->blocksize == p0.blocksize && p->quant_type == p0.quant_type &&
               p->a8.has_value() == p0.a8.has_value() && p->shape[1] == K && p->absmax.scalar_type() == at::kFloat &&
               (!p->a8.has_value() || (p->code.has_value() && p->offset.has_value() && p->code->scalar_type() == at::kFloat));
        total += p->shape[0];
    }
    const at::ScalarType inp = x.scalar_type();
    const at::ScalarType ct = p0.compute_dtype.value_or(inp);
    same = same && (ct == at::kHalf || ct == at::kBFloat16 || ct == at::kFloat) && M * total <= std::numeric_limits<int>::max();
    if (!same)
        return one_by_one();
    const int dt = dtype_code(ct);
    std::vector<int> Ns(count);
    for (size_t i = 0; i < count; ++i) {
        check_c_int(ps[i]->shape[0], "gemm_4bit N");
        Ns[i] = static_cast<int>(ps[i]->shape[0]);
    }
    check_c_int(M, "gemm_4bit M");
    check_c_int(K, "gemm_4bit K");
    if (bnb_mi355x_gemm_4bit_grouped_route(dt, static_cast<int>(count), Ns.data(), static_cast<int>(M), static_cast<int>(K), static_cast<int>(p0.blocksize)) == 0)
        return one_by_one();
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(std::optional<c10::Device>(x.device()));
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.get_device()).stream();
    const at::Tensor xc = (ct != inp ? x.to(ct) : x).contiguous();
    const int qt = quant_code(p0.quant_type);
    // one allocation, carved into the members' contiguous [*, N_i] results
    at::Tensor buf = at::empty({M * total}, xc.options());
    auto lead = xc.sizes().vec();
    std::vector<at::Tensor> outs, keep;
    std::vector<const uint8_t*> B(count), a8(count);
    std::vector<const float*> am(count), code(count), off(count);
    std::vector<void*> out(count);
    std::vector<const void*> bias(count);
    int64_t at_elem = 0;
    for (size_t i = 0; i < count; ++i) {
        const Prepared& p = *ps[i];
        lead.back() = p.shape[0];
        outs.push_back(buf.narrow(0, at_elem, M * p.shape[0]).view(lead));
        at_elem += M * p.shape[0];
        B[i] = static_cast<const uint8_t*>(p.B.const_data_ptr());
        am[i] = static_cast<const float*>(p.absmax.const_data_ptr());
        a8[i] = static_cast<const uint8_t*>(ptr(p.a8));
        code[i] = static_cast<const float*>(ptr(p.code));
        off[i] = static_cast<const float*>(ptr(p.offset));
        out[i] = outs.back().data_ptr();
        bias[i] = nullptr;
        if (p.bias.has_value()) {
            keep.push_back(p.bias->scalar_type() == ct ? *p.bias : p.bias->to(ct));
            TORCH_CHECK(keep.back().dim() == 1 && keep.back().numel() == p.shape[0], "bias must be 1D with one value per output feature");
            bias[i] = keep.back().const_data_ptr();
        }
    }
    const bool nested = p0.a8.has_value();
    bnb_mi355x_gemm_4bit_grouped(dt, xc.const_data_ptr(), static_cast<int>(count), B.data(), am.data(), nested ? a8.data() : nullptr,
                                 nested ? code.data() : nullptr, nested ? off.data() : nullptr, out.data(), bias.data(), Ns.data(), static_cast<int>(M),
                                 static_cast<int>(K), static_cast<int>(p0.blocksize), qt, stream);
    if (ct != inp)
        for (auto& y : outs)
            y = y.to(inp);
    return outs;
}

} // namespace

// The schema is defined by bitsandbytes_amd/_ops.py (string-identical to the reference's bitsandbytes/_ops.py:239-295), or by the
// reference package itself when that is imported first; this only adds the device kernel.
TORCH_LIBRARY_IMPL(bitsandbytes, CUDA, m) { m.impl("gemm_4bit", &gemm_4bit_hip); }

// Extension ops of this package (no counterpart in the reference): the prepared Linear4bit call.
TORCH_LIBRARY_FRAGMENT(bitsandbytes_amd, m) {
    m.def("linear4bit_prepare(Tensor B, int[] shapeB, Tensor absmax, int blocksize, str quant_type, Tensor? bias=None, Tensor? absmax_8bit=None, "
          "Tensor? absmax_code=None, Tensor? absmax_offset=None, ScalarType? compute_dtype=None) -> int",
          &linear4bit_prepare);
    m.def("linear4bit_prepared(Tensor x, int handle) -> Tensor", &linear4bit_prepared);
    m.def("linear4bit_release(int handle) -> ()", &linear4bit_release);
    m.def("linear4bit_group_prepared(Tensor x, int[] handles) -> Tensor[]", &linear4bit_group_prepared);
}
