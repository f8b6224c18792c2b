// Dense layers of the TRAINING step over point-major rows, with the BatchNorm + ReLU of the PREVIOUS layer applied while the input is
// loaded and the batch statistics of THIS layer's output taken while it is stored (gfx950, bf16 storage, fp32 accumulation).
//
// replaces (reference, under autograd and bf16 autocast): every  conv -> bn -> relu -> conv  chain over rows of
// source/base/nn.py (STN :162-190, PointNetfeat :323-336, MLP :376-417) and the fc -> relu -> fc chain of the interpolation head
// (source/poco_model.py:400-410).  As separate ops one such layer is 5 streaming passes forward (GEMM out, statistics in, normalise
// in + out, next GEMM in) and ~10 backward over tensors of up to 1.28 M x 256 elements; the layers are bandwidth-bound
// (32-128 flop/byte), so the passes ARE the cost.  Here the normalised activation is never written:
//
//   forward    y[rows, cout] = act(x) W^T + b,   act(x) = relu?(x * scale_in + shift_in)        x, y bf16 RAW layer outputs
//              statistics of the (rounded) y per channel -> scale_out = gamma * rstd, shift_out = beta - mean * scale_out, the
//              running statistics updated like torch.nn.BatchNorm1d (momentum, unbiased variance)
//   backward   G = gy + gS + 2 y gQ   (gS, gQ: what the loss gradient wrt scale_out / shift_out means for every row of y)
//              dx = (G W) * [act mask] * scale_in,   d scale_in = sum_rows (G W) * mask * x,   d shift_in = sum_rows (G W) * mask
//              dW = G^T act(x),  db = sum_rows G,    dgamma, dbeta
//   = 2 passes forward (x in, y out) and 2 x (gy, y, x in) + dx out backward.
//
// Kernels (8 waves per workgroup, persistent over row tiles):
//   rows_fwd_kernel   transposed product  D[cout][row] = W[cout][k] act(x)[row][k]  on v_mfma_f32_16x16x32_bf16: the B operand (8 input
//                     channels of a row per lane) is loaded straight from global memory in fragment order, the whole W sits in LDS as
//                     A fragments; a lane ends up with 16 CONSECUTIVE output channels of one row (the assignment of output channels to
//                     MFMA rows is free, so it is chosen for 32-byte stores); per-lane statistics in registers, reduced in a fixed order
//   rows_dx_kernel    same shape with the roles of the channel axes swapped, W^T fragments in LDS
//   rows_dw_kernel    contraction over the ROWS: tiles of 32 rows of G and act(x) are written row-major into LDS and read back as MFMA
//                     operands with ds_read_b64_tr_b16 (the LDS transpose read of gfx950, lane map measured by the round-2 probe ubench/tr_probe.hip, git history); a workgroup keeps
//                     the whole [cout, cin] product in registers over its slab of rows; slab partials are summed in a fixed order
// Everything is deterministic (no floating-point atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

#define PPS_OK 0
#define PPS_ERR_ARG 1
#define PPS_ERR_LAUNCH 2

#define PPS_NS rt_bf16
#define PPS_ELEM_F16 0
#include "pps_rows_train_impl.h"
#undef PPS_NS
#undef PPS_ELEM_F16
#define PPS_NS rt_f16
#define PPS_ELEM_F16 1
#include "pps_rows_train_impl.h"
#undef PPS_NS
#undef PPS_ELEM_F16

// dtype of the 16-bit tensors: 1 = bfloat16, 2 = IEEE half (trainer.precision bf16-mixed / 16-mixed)
#define PPS_BY_TYPE(CALL) (dtype == 2 ? rt_f16::CALL : (dtype == 1 ? rt_bf16::CALL : PPS_ERR_ARG))

extern "C" {

int pps_rows_layer_supported(int cin, int cout) { return rt_bf16::pps_rows_layer_supported(cin, cout); }
size_t pps_rows_layer_ws_bytes(int cin, int cout) { return rt_bf16::pps_rows_layer_ws_bytes(cin, cout); }
size_t pps_rows3_ws_bytes() { return rt_bf16::pps_rows3_ws_bytes(); }
size_t pps_patch_transform_ws_bytes() { return rt_bf16::pps_patch_transform_ws_bytes(); }
size_t pps_head_input_ws_bytes(int c) { return rt_bf16::pps_head_input_ws_bytes(c); }

int pps_rows3_fwd(const float* x, int64_t rows, const float* w, const float* bias, int dtype, void* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float momentum, float eps, float* out_affine, float* save, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows3_fwd(x, rows, w, bias, y, gamma, beta, running_mean, running_var, momentum, eps, out_affine, save, ws, stream));
}
int pps_rows3_bwd(const float* x, const void* y, const void* gy, int64_t rows, int dtype, const float* gamma, const float* save, const float* d_affine,
                  float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows3_bwd(x, y, gy, rows, gamma, save, d_affine, dw, dbias, dgamma, dbeta, ws, stream));
}
int pps_patch_transform_fwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, int64_t q,
                            int p, int dtype, void* out, void* stream) {
    return PPS_BY_TYPE(pps_patch_transform_fwd(x, in_scale, in_shift, in_relu, t, add_identity, q, p, out, stream));
}
int pps_patch_transform_bwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, const void* g,
                            int64_t q, int p, int dtype, void* dx, void* dt, float* d_in_affine, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_patch_transform_bwd(x, in_scale, in_shift, in_relu, t, add_identity, g, q, p, dx, dt, d_in_affine, ws, stream));
}
int pps_head_input_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, int dtype, const float* wx,
                       void* h1, void* stream) {
    return PPS_BY_TYPE(pps_head_input_fwd(table, ids, pts, query, q, k, c, wx, h1, stream));
}
int pps_head_input_dwx(const void* dh1, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, int dtype, float* dwx, void* ws,
                       void* stream) {
    return PPS_BY_TYPE(pps_head_input_dwx(dh1, ids, pts, query, q, k, c, dwx, ws, stream));
}
int pps_rows_extrema_16(const void* x, int64_t groups, int p, int c, int dtype, float* mx, float* mn, int* amx, int* amn, void* stream) {
    return PPS_BY_TYPE(pps_rows_extrema_bf16(x, groups, p, c, mx, mn, amx, amn, stream));
}
int pps_rows_layer_fwd(const void* x, int64_t rows, int cin, int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                       const float* bias, int cout, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, float* out_affine, float* save, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows_layer_fwd(x, rows, cin, in_scale, in_shift, in_relu, w, bias, cout, y, gamma, beta, running_mean, running_var, momentum, eps,
                                          out_affine, save, ws, stream));
}
int pps_rows_layer_bwd(const void* x, const void* y, const void* gy, int64_t rows, int cin, int cout, int dtype, const float* in_scale,
                       const float* in_shift, int in_relu, const float* w, const float* gamma, const float* save, const float* d_affine,
                       void* dx, const void* dx_add, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                       void* stream) {
    return PPS_BY_TYPE(pps_rows_layer_bwd(x, y, gy, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx, dx_add, d_in_affine, dw, dbias,
                                          dgamma, dbeta, ws, stream));
}
int pps_rows_layer_bwd_attn(const void* x, const void* gy, int64_t rows, int cin, int cout, int dtype, const float* w, const float* att_weights,
                            const void* att_dpooled, int att_k, void* dx, float* dw, float* dbias, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows_layer_bwd_attn(x, gy, rows, cin, cout, w, att_weights, att_dpooled, att_k, dx, dw, dbias, ws, stream));
}
size_t pps_head_chain_ws_bytes(void) { return rt_bf16::pps_head_chain_ws_bytes(); }
int pps_head_chain_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int dtype, const float* wx,
                       const float* w2, const float* b2, const float* w3, const float* b3, const float* wq, const float* bq, void* h1, void* y2, void* y3,
                       void* qy, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_head_chain_fwd(table, ids, pts, query, q, k, wx, w2, b2, w3, b3, wq, bq, h1, y2, y3, qy, ws, stream));
}
int pps_rows_layer_pooled_supported(int cin, int cout, int pool_p) { return rt_bf16::pps_rows_layer_pooled_supported(cin, cout, pool_p); }
int pps_rows_layer_bwd_pooled(const void* x, const void* y, const void* gval, const uint8_t* garg, int pool_p, int64_t rows, int cin, int cout,
                              int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma,
                              const float* save, const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma,
                              float* dbeta, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows_layer_bwd_pooled(x, y, gval, garg, pool_p, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx,
                                                 d_in_affine, dw, dbias, dgamma, dbeta, ws, stream));
}

int pps_rows_layer_bwd_rank2(const void* x, const void* y, const float* g_a, const float* g_dl, const float* g_dp, const float* g_v, int pool_p, int64_t rows,
                             int cin, int cout, int dtype, const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma,
                             const float* save, const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma,
                             float* dbeta, void* ws, void* stream) {
    return PPS_BY_TYPE(pps_rows_layer_bwd_rank2(x, y, g_a, g_dl, g_dp, g_v, pool_p, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx,
                                                d_in_affine, dw, dbias, dgamma, dbeta, ws, stream));
}

}  // extern "C"
