// Kernel argument blocks and internal entry points shared by the translation units of libdmf_hip.so.
// ONE definition each: dmf_capi.hip fills these structs, the kernel files read them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/dmf.h"
#include "dmf_shapes.h"

namespace dmf {


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the (kernel, device) pair and is not capturable: every launch
// site keeps ONE of these per kernel instance (a function-local static) and calls set() before the launch — once per device,
// under a lock (two host threads may reach a first launch together).
struct LdsAttrOnce {
  std::mutex mu;
  bool done[64] = {};
  hipError_t set(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = (e == hipSuccess);
    return e;
  }
};

enum { MODE_FWD = 0, MODE_TRAIN = 1, MODE_BWD = 2, MODE_TOKENS = 3, MODE_DENSE = 4, MODE_UNIT = 5 };
// TOKENS: conv stages only, for the attention kernel.  DENSE: conv backward from dense dL/dYa, dL/dYb maps [B][F][P2]
// (written by the attention kernel's backward) instead of the rank-1 pool x dz form; the head is skipped.
// UNIT (v2 kernel only): forward + the conv backward for a UNIT gradient on every pooled feature; `slab` then is
// [B][SLAB], one row of unit gradients per PATCH (dmf_forward_unit / dmf_backward_unit: a loss that couples the batch).

// patch kernel (dmf_patch_v2.hip)
struct KArgs {
  dmf_input in;
  const float* theta;
  const float* pool;
  const int32_t* labels;
  const float* dlogits;
  float loss_scale;
  const float* scaler;  // nullable: device loss-scaler state, [0] multiplies loss_scale (dmf_train_fwd_bwd_scaled)
  float* logits;
  float* loss;
  int32_t* pred;
  float* slab;   // [grid][SLAB]
  float* ws_z;   // [B][2F]
  float* ws_h;   // [B][H]
  float* ws_dh;  // [B][H]
  float* ws_dl;  // [B][KMAX]
  int32_t* adam_step;   // device step counter to advance (nullable)
  unsigned short* tokA; // MODE_TOKENS: bf16 token maps [B][128][64] (tokens x channels, zero padded) of both branches
  unsigned short* tokB;
  float* zout;          // MODE_TOKENS: pooled features [B][2F] before attention
  const float* dYa;     // MODE_DENSE: dL/d(spat_a output) [B][F][P][RS] (rows padded to 16 bytes)
  const float* dYb;     // MODE_DENSE: dL/d(spat_b output) [B][F][P][RS]
  int32_t K;
};
// wave-per-channel-block kernel (dmf_patch_v2.hip): FWD / TRAIN / BWD of the shapes it is built for
int patch_v2_supported(const dmf_shape& s, int mode, int half = 0);   // half: dmf_input.half (fp16 primary scene)
const char* patch_v2_half_shape_list();
hipError_t patch_v2_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st);
// second half of the unit-gradient step: per patch dh, dz from dlogits; slab row of workgroup g = sum over its patches of
// dz x unit row; ws_dh / ws_dl for the fc gradients
struct UnitBwdArgs {
  const float* theta; const float* dlogits; const float* unit; const float* ws_h;
  float* ws_dh; float* ws_dl; float* slab;
  int32_t B, K;
};
hipError_t patch_v2_unit_backward(const dmf_shape& s, const UnitBwdArgs& a, hipStream_t st);
const char* patch_v2_shape_list();

// attention kernels (dmf_attention.hip)
struct AttnTrainArgs {
  const unsigned short* tokA; const unsigned short* tokB;     // [B][128][64] bf16
  const float* zin;                     // [B][2F]
  const float* theta; const float* pool;
  const int32_t* labels; const int32_t* cursor;   // labels[(*cursor) * B + b]   (cursor may be null)
  const float* dlogits;                 // used when labels == null: caller-supplied dL/dlogits [B][K]
  float loss_scale;
  float* logits; float* loss;           // [B][K], [B] (loss may be null)
  float* ws_z; float* ws_h; float* ws_dh; float* ws_dl;   // head vectors for the gradient reduce
  float* dYa; float* dYb;               // [B][F][P][RS], RS = P rounded up to 4 (rows 16-byte aligned)
  float* aslab;                         // [gridDim][4*E*F] attention weight gradients (Wq, Wk, Wv, Wo)
  int32_t* pred;                        // forward-only launch: argmax per patch (may be null)
  const unsigned short* wprep;          // [NH][WPREP] bf16 weights of every head, already in the LDS layout (attn_prep_kernel)
  int64_t oWq, oWk, oWv, oWo, oFc1w, oFc1b, oFc2w, oFc2b;
  int32_t B, K;
};
size_t attn_prep_bytes();
hipError_t attn_prep_launch(const float* theta, int64_t oWq, int64_t oWk, int64_t oWv, int64_t oWo, void* out, hipStream_t st);
hipError_t attn_train_dispatch(const dmf_shape& s, const AttnTrainArgs& a, int grid, hipStream_t st);
hipError_t attn_forward_dispatch(const dmf_shape& s, const AttnTrainArgs& a, int grid, hipStream_t st);
int attn_shape_supported(const dmf_shape& s);

// stage-2 kernels (dmf_qua.hip)
struct QuaArgs {
  const float* logits; int bs, K;
  const int32_t* labels; const int32_t* cursor;
  float alpha, beta, gamma, eps, tao, grad_scale;
  float* loss; float* loss_hist; float* dlogits;
  const float* scaler;   // nullable: device loss-scaler state, [0] multiplies grad_scale
};
hipError_t launch_qua_loss(const QuaArgs& a, hipStream_t st);
hipError_t launch_pair_argmax(const float* logits, int bs, int K, int32_t* pred, hipStream_t st);
hipError_t launch_band_mean(const float* x, int layout, int64_t n_img, int64_t n_pix, int C, float* out, hipStream_t st);

#ifdef DMF_STAMPS
hipError_t set_attn_stamps(unsigned long long* p);
hipError_t set_v2_stamps(unsigned long long* p);
#endif

}  // namespace dmf
