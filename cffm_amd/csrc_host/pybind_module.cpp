// Thin pybind11 layer over the C ABI of libcffm_hip.so (include/cffm_hip.h): the binding BASELINE.json's north_star names
// ("Python on PyTorch-ROCm tensors calling hand-written HIP kernels through a thin pybind11 C-ABI layer").
//
// One Python callable per C entry point, same name, same argument order.  Every pointer argument - device pointers
// borrowed from torch tensors (tensor.data_ptr()), the hipStream_t, and the addresses of the small host structs
// (cffm_shape_t, cffm_tables_t, layout structs: ctypes.addressof(...)) - travels as a plain integer; scalars travel as
// Python ints / floats.  Nothing is allocated, copied or interpreted here; the wrapper of a call is the argument
// conversion pybind11 generates (~0.3 us for the 12 arguments of cffm_train_step, against ~3 us through ctypes).
// The ctypes binding in cffm_amd/hip.py stays as the documented alternative and is what runs when this
// module has not been built.
#include <pybind11/pybind11.h>

#include <cstdint>

#include "../../include/cffm_hip.h"

namespace py = pybind11;

namespace {

template <class T> struct Arg {
    using py_t = T;
    static T cv(T v) { return v; }
};
template <class T> struct Arg<T*> {
    using py_t = std::uintptr_t;
    static T* cv(std::uintptr_t v) { return reinterpret_cast<T*>(v); }
};

template <class R, class... A>
void bind(py::module_& m, const char* name, R (*f)(A...)) {
    m.def(name, [f](typename Arg<A>::py_t... a) -> R { return f(Arg<A>::cv(a)...); });
}

}  // namespace

#define CFFM_BIND(fn) bind(m, #fn, &fn)

PYBIND11_MODULE(_cffm_pybind, m) {
    m.doc() = "pybind11 layer over libcffm_hip.so (include/cffm_hip.h); pointers are passed as integers";
    m.attr("ABI_VERSION") = CFFM_ABI_VERSION;
    CFFM_BIND(cffm_abi_version);
    CFFM_BIND(cffm_error_string);
    CFFM_BIND(cffm_theta_layout);
    CFFM_BIND(cffm_ws_layout);
    CFFM_BIND(cffm_gather);
    CFFM_BIND(cffm_gather_inner_fwd_ok);
    CFFM_BIND(cffm_gather_inner_fwd);
    CFFM_BIND(cffm_inner_fwd);
    CFFM_BIND(cffm_inner_bwd);
    CFFM_BIND(cffm_outer_conv0_fwd);
    CFFM_BIND(cffm_outer_conv0_bwd);
    CFFM_BIND(cffm_conv_fwd);
    CFFM_BIND(cffm_conv_bwd);
    CFFM_BIND(cffm_head_fwd);
    CFFM_BIND(cffm_head_bwd);
    CFFM_BIND(cffm_reduce_slabs);
    CFFM_BIND(cffm_dense_adagrad);
    CFFM_BIND(cffm_sparse_adagrad);
    CFFM_BIND(cffm_predict);
    CFFM_BIND(cffm_forward);
    CFFM_BIND(cffm_backward);
    CFFM_BIND(cffm_dp_runs_ok);
    CFFM_BIND(cffm_dp_local);
    CFFM_BIND(cffm_dp_dense_floats);
    CFFM_BIND(cffm_dp_local_dense);
    CFFM_BIND(cffm_dp_apply_dense);
    CFFM_BIND(cffm_backward_unscaled);
    CFFM_BIND(cffm_dp_apply);
    CFFM_BIND(cffm_train_step);
    CFFM_BIND(cffm_train_step_opt);
    CFFM_BIND(cffm_packed_row_floats);
    CFFM_BIND(cffm_gather_packed);
    CFFM_BIND(cffm_stage_packed);
    CFFM_BIND(cffm_forward_packed);
    CFFM_BIND(cffm_backward_unscaled_packed);
    CFFM_BIND(cffm_pack_rows_dedup);
    CFFM_BIND(cffm_shard_plan_scratch_bytes);
    CFFM_BIND(cffm_shard_plan);
    CFFM_BIND(cffm_eval_scratch_bytes);
    CFFM_BIND(cffm_eval_sums);
    CFFM_BIND(cffm_probe_copy);
    CFFM_BIND(cffm_probe_read);
    CFFM_BIND(cffm_probe_mfma);
    CFFM_BIND(cffm_probe_mfma_bf16);
}
