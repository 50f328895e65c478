// dmf_patch_v2.hip — the train / eval step of the late-fusion GMFNet as ONE launch, second design.
//
// Replaces (reference): `output = self.cur_model(data1, data2)`, `loss = self.loss(output, target.long())`,
// `loss.backward()` — solver/mainsolver.py:52-54 — and the eval forward + argmax (mainsolver.py:109,139,169-170).
// Arithmetic: oracle/gmfnet_ref.py.  The ONLY patch kernel since round 3 (round 1's generic kernel, which also took band
// groups that are not whole 16-byte chunks — 200 bands at gmf.width 32 — was retired with that shape).
//
// Measured facts this design is built on (tools/valu_rate.hip, tools/phase_profile_v2.py on MI355X):
//   * A SIMD issues one vector instruction per ~3.2 (v_fmac) to ~4.9 (DPP, packed) cycles however many waves it hosts,
//     and one wave alone already reaches ~75 % of that.  With one patch per CU the kernel is VECTOR-ISSUE bound: time =
//     instruction slots of the busiest SIMD x ~4 cycles.  So: few instructions, every lane busy, equal load per SIMD.
//   * The compiler's waitcnt pass (a) turns every wait into "wait for zero" after a FLAT-encoded LDS-DMA
//     (global_load_lds), (b) puts every LDS access it can see behind ALL outstanding LDS-DMA.
//
// Layout: a wavefront owns CPW feature channels of BOTH branches end to end.  A channel takes LPC = P rows + padding
// lanes (a multiple of 4: 12 for P = 11), channels are packed back to back (5 x 12 = 60 of 64 lanes for the 200-band net:
// 8 conv waves, two per SIMD, + 1 head wave).  Every intermediate (spec_a rows, depthwise rows, ReLU gates, gradients)
// lives in registers; the rows above / below are the neighbouring LANES, fetched by whole-wave DPP shifts (wave_shr:1 /
// wave_shl:1), and a channel's padding lane (Y1 = 0, gate closed) is the zero padding for both of its neighbours.
// Sums over a channel's rows stop at QUAD level (two quad_perm DPP steps): a 12-lane segment straddles the 16-lane DPP
// rows, so the NQ quad partials are kept apart (pooled features, slab row copies) and added by whoever consumes them.
//
// Flow per patch (gather mode, first patch of a workgroup; DESIGN.md section 4 has the measured timeline):
//   entry    conv part of theta + pooling profile -> LDS by LDS-DMA, no wait; the first coordinates are requested first
//   gather   as soon as the coordinates are known: aux image piece + the FIRST TWO window pieces of every wave, then a counted
//            wait for what the wave issued before them (tables, aux piece) and barrier X; the other pieces leave from inside
//            the aux phase (one per spat_b output column) — issuing is back-pressured, the aux arithmetic is not
//   aux      lift_b + spat_b forward AND backward for a UNIT upstream gradient, under the gather (tables and aux rows through
//            LDS reads the compiler cannot see; see hidden_read)
//   barrier W  window complete
//   primary  spec_a (packed FMAs, weights in registers, x from the group's LDS slice), spat_a forward
//   barrier 1  pooled features complete -> the head wave runs fc1 / fc2 / softmax-CE / dh while the conv waves do spat_a's
//            unit backward and form the unit spec_a weight gradient from the still-resident window
//   barrier 2  dh complete -> dz of every feature by 4 lanes, barrier 3, one coalesced scaled copy-out of the slab row
// Later patches of a workgroup (batch > 256) load the lane's aux row straight into registers and skip barrier X.
// Variants compiled from the same body: HF (fp16 scene, fp16 spec_a operands), S = 4 (aux at 4x the resolution: aux patch
// image in LDS, bank-rotated rows), MODE_UNIT (per-patch unit gradients for batch-coupled losses), MODE_TOKENS / MODE_DENSE
// (the two conv launches of the attention network's step).
// The network is piecewise linear and channel f reaches the head only through the pooled scalars z_a[f], z_b[f]: that is
// what makes the unit-gradient backward (scaled by dL/dz[f] at the very end) exact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "dmf_kargs.h"
#include "dmf_lanes.h"

namespace dmf {

// Diagnostic build only (-DDMF_STAMPS, tools/phase_profile_v2.py): every wavefront keeps its clock stamps in scalar
// registers (no wait, no store, no branch inside the phases) and lane 0 dumps them right before the wave ends.
#ifdef DMF_STAMPS
__device__ unsigned long long* g_v2stamps = nullptr;     // [block][16 waves][16 stamps]
#define VSTAMP_DECL unsigned long long vst_[14] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
// (the stamp WAITS for its own result: s_memtime returns like a scalar load, and a compiler that believes the value is there at
// once may spill it and reuse the register pair — the result, landing late, then overwrites whatever lives there.  Found as a
// memory fault of the 4-band instance of this diagnostic build, where 14 live stamps exhaust the scalar registers.)
#define VSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(vst_[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VSTAMP_W(i) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(vst_[i])); } while (0)
#define VSTAMP_RT(i) do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(vst_[i])); } while (0)
#define VSTAMP_DUMP() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0 && g_v2stamps != nullptr) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 14; ++i_) g_v2stamps[((size_t)blockIdx.x * 16 + wave) * 16 + i_] = vst_[i_]; } } while (0)
#else
#define VSTAMP_DECL do { } while (0)
#define VSTAMP(i) do { } while (0)
#define VSTAMP_RT(i) do { } while (0)
#define VSTAMP_W(i) do { } while (0)
#define VSTAMP_DUMP() do { } while (0)
#endif

typedef const int32_t __attribute__((address_space(4))) cint;

// LDS-array cycles of one ds_read_b128 x-row read per wave, summed over the conv waves, for a window image
// [row][RS] = [row][P pixels x CS | row padding]: the 16 lanes of a service group hold different patch rows, and up to two
// band groups (a wave's channels may straddle groups).  Used to pick the paddings at compile time.
constexpr int v2_read_cost(int CS, int RS, int P, int Cg, int lpc, int cpw, int M, int nb) {
  const int first[4][3] = {{0, 12, 20}, {4, 16, 28}, {32, 44, 52}, {36, 48, 60}};
  const int count[4][3] = {{4, 4, 8}, {8, 4, 4}, {4, 4, 8}, {8, 4, 4}};
  int tot = 0;
  for (int w = 0; w < nb; ++w)
    for (int sg = 0; sg < 4; ++sg) {
      int worst = 1;
      for (int b = 0; b < 64; b += 4) {          // 16-byte reads: bank quads
        int addrs[16] = {0}, n = 0;
        for (int part = 0; part < 3; ++part)
          for (int q = 0; q < count[sg][part]; ++q) {
            const int l = first[sg][part] + q;
            const int seg = l / lpc, r = l % lpc;
            const int f = cpw * w + (seg < cpw ? seg : cpw - 1);
            const int rc = r < P ? r : P - 1;
            const int addr = rc * RS + (f / M) * Cg;
            if (addr % 64 == b) {
              bool seen = false;
              for (int i = 0; i < n; ++i) seen = seen || addrs[i] == addr;
              if (!seen) addrs[n++] = addr;
            }
          }
        worst = n > worst ? n : worst;
      }
      tot += worst;
    }
  return tot;
}
// (pixel stride CS, row stride RS) packed as CS * 65536 + RS: the smallest image within 30 % of conflict-free reads — no
// padding if that already holds, else row padding (keeps a window row one contiguous run of scene bytes), else pixel padding
constexpr int v2_pick_layout(int C, int P, int Cg, int lpc, int cpw, int M, int nb) {
  const int ideal = 4 * nb;
  int best = C * 65536 + P * C, bc = v2_read_cost(C, P * C, P, Cg, lpc, cpw, M, nb);
  if (bc * 10 <= ideal * 13) return best;
  for (int rp = 4; rp <= 60; rp += 4) {
    const int c = v2_read_cost(C, P * C + rp, P, Cg, lpc, cpw, M, nb);
    if (c * 10 <= ideal * 13) return C * 65536 + P * C + rp;
    if (c < bc) { bc = c; best = C * 65536 + P * C + rp; }
  }
  for (int pad = 4; pad <= 32; pad += 4) {
    const int c = v2_read_cost(C + pad, P * (C + pad), P, Cg, lpc, cpw, M, nb);
    if (c < bc) { bc = c; best = (C + pad) * 65536 + P * (C + pad); }
  }
  return best;
}

constexpr int v2_pick_cpw(int F, int lpc) {
  int c = 64 / lpc;
  while (c > 1 && F % c != 0) --c;
  return c;
}

// HF: the primary scene (and so the window image in LDS) holds IEEE fp16 — two bands per 32-bit word; spec_a then runs on
// fp16 operands (window and weights, round-to-nearest-even) with fp32 accumulation.  Everything downstream is fp32.
template <class Sh, bool TR = true, bool HF = false>
struct V2 {
  static constexpr int P = Sh::P, P2 = Sh::P2, Cg = Sh::Cg, C2 = Sh::C2, F = Sh::F, F2 = Sh::F2, H = Sh::H, G = Sh::G;
  static constexpr int CW = HF ? Sh::C / 2 : Sh::C;     // 32-bit words per pixel
  static constexpr int CgW = HF ? Cg / 2 : Cg;          // ... per band group
  // lanes per channel: P rows + at least one zero-padding lane, whole quads — except P = 16, where a channel is exactly one
  // 16-lane DPP row and the row shifts' own boundary zero-fill is the padding
  static constexpr bool ROWDPP = (P == 16);
  static constexpr int LPC = ROWDPP ? 16 : ((P + 1 + 3) / 4) * 4;
  static constexpr int CPW = v2_pick_cpw(F, LPC);     // channels per wave
  static constexpr int NQ = LPC / 4;                  // quad partials per channel sum
  static constexpr int NB = F / CPW;                  // conv wavefronts
  static constexpr int NW = NB + 1;                   // + the head wavefront
  static constexpr int NT = NW * 64;
  // window in LDS: the patch image [P rows][RS] with RS = P pixels x CS (+ row padding), pixel-major like the scene, so that
  // a 1-KiB gather piece is (mostly) 1 KiB of contiguous scene bytes; the paddings spread the row-lanes' ds_read_b128 over
  // the banks (v2_pick_layout)
  static constexpr int QC = Cg / 4;
  static constexpr int LAYOUT = v2_pick_layout(CW, P, CgW, LPC, CPW, Sh::M, NB);
  static constexpr int CS = LAYOUT / 65536, RS = LAYOUT % 65536;
  static constexpr int XF = ((P * RS + 255) / 256) * 256;    // floats, whole pieces
  static constexpr int NPIECE = XF / 256;
  static constexpr int NK = (NPIECE + NW - 1) / NW;   // pieces per wave (all NW waves gather)
  // A wave issues only its first K0 pieces in front of barrier X and the rest from inside the aux phase (one after each
  // output column of spat_b): issuing is back-pressured by the memory pipeline, and a wave that issues its whole share first
  // reaches the barrier — behind which the aux arithmetic of ALL waves waits — thousands of cycles late.
  static constexpr bool SX = Sh::S > 1;               // aux modality at S x the resolution (lift_b = S x S stride-S conv)
  static constexpr int K0 = (NK > 3 && !SX) ? 2 : NK;
  static constexpr int NREST = NK - K0;
  static constexpr int AR0 = P * C2;                  // aux floats per patch row
  static constexpr int RSP = ((P + 3) / 4) * 4;       // row stride of the staged pooling profile
  static constexpr int W2S = ((H / 4) | 1) * 4;       // row stride of the staged fc2.weight
  static constexpr int ZS = ((F2 + 15) / 16) * 16;    // stride of one quad-partial vector of pooled features
  static constexpr int oX = 0;
  static constexpr int oTh = oX + XF;                 // conv part of theta, parameter order
  static constexpr int oPool = oTh + Sh::SLAB;        // pooling profile [P][RSP]
  static constexpr int oW1T = oPool + P * RSP;        // fc1.weight transposed [2F][H]  (training: dz on the conv waves)
  static constexpr int oZ = oW1T + (TR ? F2 * H : 0); // pooled features [ZS]; eval: two buffers (it runs ahead of the head)
  static constexpr int NZB = TR ? 1 : 2;
  static constexpr int oZP = oZ + NZB * ZS;           // per conv wave: the quad partials of its channels' pooled features [NB][ZPW]
  static constexpr int ZPW = ((2 * CPW * NQ + 15) / 16) * 16;
  // aux patch image [P][AR0] of a workgroup's FIRST patch (gather mode: staged by LDS-DMA in front of the window pieces).  It
  // lies over the head's vectors h / dlogits / dh (/ dz) (+ AUXX floats in front of them): the image is consumed before the
  // first barrier W, those vectors are first written behind the first barrier 1.
  static constexpr int AR0P = (AR0 + 3) & ~3;         // image row stride: 16-byte aligned rows (read by ds_read_b128)
  static constexpr int AUXF = ((P * AR0P + 63) / 64) * 64;
  static constexpr int NAUXP = AUXF / 64;             // 256-byte LDS-DMA pieces of the aux image
  static constexpr int HEADV = H + KMAX + H + (TR ? ZS : 0);     // floats of h, dlogits, dh, dz
  static constexpr int AUXX = AUXF > HEADV ? AUXF - HEADV : 0;
  static constexpr int oAux = oZP + NB * ZPW;
  static constexpr int oHv = oAux + AUXX;             // h [H]
  static constexpr int oDl = oHv + H;                 // dlogits [KMAX]
  static constexpr int oDh = oDl + KMAX;              // dh [H]
  static constexpr int oDz = oDh + H;                 // dL/dz [ZS]
  static_assert(oDz + (TR ? ZS : 0) - oAux >= AUXF, "aux image over the head vectors");
  static constexpr int oSlab = oDz + (TR ? ZS : 0);   // [NQ][SLAB] UNIT weight gradients of the current patch, one copy per quad partial
  static constexpr int NCOPY = TR ? NQ : 0;
  static constexpr int oDzix = oSlab + NCOPY * Sh::SLAB;  // training: per 16-byte slab piece, the dz indices of its 4 elements (8 bits each)
  static constexpr int NDZ = TR ? ((Sh::SLAB / 4 + 3) / 4) * 4 : 0;
  static_assert(oDzix % 4 == 0, "the dz index table is staged in 16-byte pieces");
  // S > 1: the aux patch image [S P rows][RL = S P C2 floats] of EVERY patch lives in LDS (a lane needs S rows of it: too
  // many for registers).  The 16-byte chunks of a row are rotated by the PATCH row r = row / S, so that the 16 row-lanes
  // of a ds_read_b128 service group, whose rows are S RL floats = a multiple of 256 bytes apart, land on different banks;
  // the LDS-DMA fill applies the rotation through its per-lane source addresses.
  static constexpr int RL = Sh::SP * C2;
  static constexpr int NC4 = RL / 4;
  static constexpr int AUXS = SX ? ((Sh::SP * RL + 255) / 256) * 256 : 0;
  static constexpr int NAUXS = AUXS / 256;            // 1-KiB LDS-DMA pieces
  static constexpr int oAuxS = oDzix + NDZ;
  static constexpr int oW2 = oAuxS + AUXS;               // fc2.weight [K rounded up to 4][W2S]  (run-time K)
  static constexpr int FIXED = oW2;
  static int lds_bytes(int K) { return (FIXED + ((K + 3) & ~3) * W2S) * 4; }
  // (a lane's 16-byte gather piece must not straddle a pixel — or, without pixel padding, a window row)
  static constexpr bool OK = (Sh::S == 1 || (Sh::S == 4 && RL % 4 == 0 && NC4 >= P)) && C2 <= 4 && Cg % 4 == 0 && LPC <= 32 && P <= 16 && CPW >= 2 && NW <= 12 && H == 64 && F2 <= 128 &&
                             P * RSP <= 2 * NT && NK <= 16 && NREST <= P && (CS == CW ? (P * CW) % 4 == 0 : CW % 4 == 0);
};

// LDS reads the compiler's waitcnt pass cannot see.  It orders EVERY LDS access it knows of behind all outstanding
// LDS-DMA (s_waitcnt vmcnt(0)), which would put the aux phase's table reads behind the whole window gather; these go
// through inline asm, are waited for with hidden_wait(), and are only used on data written before the gather was issued.
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ float hidden_read(unsigned addr, int off) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
  return v;
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v hidden_read4(unsigned addr, int off) {      // (16-byte aligned address)
  f4v v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
  return v;
}
__device__ __forceinline__ void hidden_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define HIDDEN_USE(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ float wave_max_dpp(float v) {  // all 64 lanes, result in every lane
#define DMF_DPP_MAX(v, CTRL) fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, false)))
  v = DMF_DPP_MAX(v, 0xB1);
  v = DMF_DPP_MAX(v, 0x4E);
  v = DMF_DPP_MAX(v, 0x141);
  v = DMF_DPP_MAX(v, 0x140);
#undef DMF_DPP_MAX
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  }
  {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  }
  return v;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// acc + (float)half(x.lo | x.hi) * y in one instruction (v_fma_mix_f32; written out: in the unrolled gradient loops the
// compiler converts first and multiplies after, twice the instructions)
__device__ __forceinline__ float fma_mix_lo(float x2, float y, float acc) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(x2), "v"(y));
  return acc;
}
__device__ __forceinline__ float fma_mix_hi(float x2, float y, float acc) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(x2), "v"(y));
  return acc;
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ float relu_lim(float x, float lim) { return __builtin_amdgcn_fmed3f(x, 0.f, lim); }   // lim = +inf: ReLU; 0: 0

// value held by lane-1 / lane+1 = patch row r-1 / r+1 (or a padding lane).  ROWDPP (16-row patches, one channel per 16-lane
// DPP row): row shifts, which fill with zero at the row's ends; otherwise whole-wave shifts (zero at lanes 0 / 63).
template <bool ROWDPP>
__device__ __forceinline__ float lane_above(float v) {
  if constexpr (ROWDPP) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));   // row_shr:1
  else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));                    // wave_shr:1
}
template <bool ROWDPP>
__device__ __forceinline__ float lane_below(float v) {
  if constexpr (ROWDPP) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));   // row_shl:1
  else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));                    // wave_shl:1
}
__device__ __forceinline__ float quad_sum(float v) {     // over the 4 lanes of a quad, result in each of them
  v = DMF_DPP_ADD(v, 0xB1);     // quad_perm [1,0,3,2]
  v = DMF_DPP_ADD(v, 0x4E);     // quad_perm [2,3,0,1]
  return v;
}

// Depthwise 3x3 (zero pad 1) + ReLU + anchor pooling of one channel row, and — TRAIN — its backward for a UNIT
// gradient on the pooled scalar: dY2 = [y2 > 0] * pool.  A lane holds row r of its channel; rows r-1 / r+1 are the
// neighbouring lanes.  Padding lanes carry y1 = 0 and a closed gate, which is exactly the zero padding.
//   z  : pooled partial of this row (caller sums over the channel's lanes)
//   dw : unit dL/dW2[u][v] partial, db: unit dL/db2 partial, dy: unit dL/dY1 of this row (through Y1's ReLU)
template <int P>
struct ConvRows { float y1u[P], y1d[P], gq[P]; };

struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// hook(c) runs after output column c: the aux branch issues the rest of the window gather from there (see the kernel)
template <int P, bool RD, class Hook = NoHook>
__device__ __forceinline__ void conv_row_fwd(const float (&y1c)[P], const float (&w)[9], float bias, const float (&pw)[P],
                                             ConvRows<P>& t, float& z, Hook&& hook = Hook(), float* y2 = nullptr) {
#pragma unroll
  for (int c = 0; c < P; ++c) { t.y1u[c] = lane_above<RD>(y1c[c]); t.y1d[c] = lane_below<RD>(y1c[c]); }
  z = 0.f;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    float y = bias;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int cc = c + v - 1;
      if (cc >= 0 && cc < P) {
        y = fmaf(w[v], t.y1u[cc], y);
        y = fmaf(w[3 + v], y1c[cc], y);
        y = fmaf(w[6 + v], t.y1d[cc], y);
      }
    }
    t.gq[c] = y > 0.f ? pw[c] : 0.f;
    z = fmaf(t.gq[c], y, z);
    if (y2 != nullptr) y2[c] = fmaxf(y, 0.f);           // (token mode: the feature map itself)
    hook(c);
  }
}

template <int P, bool RD>
__device__ __forceinline__ void conv_row_bwd(const float (&y1c)[P], const float (&w)[9], const ConvRows<P>& t,
                                             float (&dw)[9], float& db, float (&dy)[P]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) dw[k] = 0.f;
  db = 0.f;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    db += t.gq[c];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int cc = c + v - 1;
      if (cc >= 0 && cc < P) {
        dw[v] = fmaf(t.gq[c], t.y1u[cc], dw[v]);
        dw[3 + v] = fmaf(t.gq[c], y1c[cc], dw[3 + v]);
        dw[6 + v] = fmaf(t.gq[c], t.y1d[cc], dw[6 + v]);
      }
    }
  }
  // dY1(r,c) = [y1 > 0] * sum_{u,v} W[u][v] * dY2(r-u+1, c-v+1): u = 0 pairs with the row below, u = 2 with the row above
  float gu[P], gd[P];
#pragma unroll
  for (int c = 0; c < P; ++c) { gu[c] = lane_above<RD>(t.gq[c]); gd[c] = lane_below<RD>(t.gq[c]); }
#pragma unroll
  for (int c = 0; c < P; ++c) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int cc = c - v + 1;
      if (cc >= 0 && cc < P) {
        s = fmaf(w[v], gd[cc], s);
        s = fmaf(w[3 + v], t.gq[cc], s);
        s = fmaf(w[6 + v], gu[cc], s);
      }
    }
    dy[c] = y1c[c] > 0.f ? s : 0.f;
  }
}

template <int P, bool TR, bool RD, class Hook = NoHook>
__device__ __forceinline__ void conv_row(const float (&y1c)[P], const float (&w)[9], float bias, const float (&pw)[P],
                                         float& z, float (&dw)[9], float& db, float (&dy)[P], Hook&& hook = Hook(), float* y2 = nullptr) {
  ConvRows<P> t;
  conv_row_fwd<P, RD>(y1c, w, bias, pw, t, z, hook, y2);
  if constexpr (TR) conv_row_bwd<P, RD>(y1c, w, t, dw, db, dy);
}

// One piece of LDS-DMA: lane l reads BYTES (4 or 16) bytes at base + soff + voff into lds + BYTES l (lanes with voff < 0 are
// masked off).  BUFFER form (buffer_load_dword[x4] ... lds): see the header comment.  (Device pass only: the buffer-resource
// type does not exist in the host pass, which needs nothing but the kernel's stub.)
template <int BYTES>
__device__ __forceinline__ void dma_piece(const float* base, int soff, int voff, float* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
  if (voff >= 0) {
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 4, voff, soff, 0, 0);
  }
#endif
}
__device__ __forceinline__ void gather_piece(const float* base, int soff, int voff, float* lds) { dma_piece<16>(base, soff, voff, lds); }

// Channel (dz index) of every slab element, 8 bits each, one word per 16-byte slab piece: a compile-time table in the code
// object's constant data, staged into LDS with the other patch-invariant tables (forming it in the kernel cost every
// thread ~250 instructions of compare / branch chains IN FRONT of the first gather issue).
template <class Sh>
struct DzixTable {
  static constexpr int N = ((Sh::SLAB / 4 + 3) / 4) * 4;
  int v[N];
  constexpr DzixTable() : v{} {
    for (int t = 0; t < Sh::SLAB / 4; ++t) {
      int w = 0;
      for (int e = 0; e < 4; ++e) {
        const int p = 4 * t + e;
        int ix = 0;
        if (p < Sh::oA1b) ix = p / Sh::Cg;
        else if (p < Sh::oA2w) ix = p - Sh::oA1b;
        else if (p < Sh::oA2b) ix = (p - Sh::oA2w) / 9;
        else if (p < Sh::oB1w) ix = p - Sh::oA2b;
        else if (p < Sh::oB1b) ix = Sh::F + (p - Sh::oB1w) / Sh::TB;
        else if (p < Sh::oB2w) ix = Sh::F + p - Sh::oB1b;
        else if (p < Sh::oB2b) ix = Sh::F + (p - Sh::oB2w) / 9;
        else if (p < Sh::NCONV) ix = Sh::F + p - Sh::oB2b;
        w |= ix << (8 * e);
      }
      v[t] = w;
    }
  }
};
template <class Sh>
__device__ const DzixTable<Sh> g_dzix_table{};

// INMODE: dmf_input.mode, compile time — with a run-time branch the waitcnt pass merges the two paths' states at the
// join and waits vmcnt(0) there, i.e. for the whole window, before the aux phase.
// The eight leading arguments repeat fields of the argument block: what the prologue needs FIRST (coordinates, staging
// sources, scene bases).  As plain scalars in front of the struct they are preloaded into scalar registers at wave launch
// (-mllvm -amdgpu-kernarg-preload-count, build.py; a by-value struct is never preloaded) — a kernel argument the wave fetches
// itself arrives ~1.0 K cycles after wave entry, and the coordinate load, the first link of the kernel's longest chain of
// dependent memory round trips, was waiting for exactly that.
template <class Sh, int MODE, int INMODE, bool HF>
__global__ __launch_bounds__(V2<Sh>::NT) void patch_v2_kernel(const int32_t* xy_, const int32_t* cursor_, int B_, const float* theta_,
                                                             const float* pool_, const float* sceneA_, const float* sceneB_, int Wp_,
                                                             const KArgs a0) {
  KArgs a = a0;
  a.in.xy = xy_; a.in.cursor = cursor_; a.in.B = B_; a.theta = theta_; a.pool = pool_;
  a.in.sceneA = sceneA_; a.in.sceneB = sceneB_; a.in.Wp = Wp_;
  constexpr bool TOK = (MODE == MODE_TOKENS);          // conv stages only: bf16 token maps + pooled features (attention net)
  constexpr bool DENSE = (MODE == MODE_DENSE);         // conv backward from DENSE dL/dY2 maps (attention net); no head
  constexpr bool TR = (MODE != MODE_FWD && !TOK);
  constexpr bool UNIT = (MODE == MODE_UNIT);           // forward + unit gradients per patch; loss and scaling happen elsewhere
  constexpr int RS4 = (Sh::P + 3) & ~3;                // row stride of the dense gradient maps [B][F][P][RS4]
  constexpr int TKS = 72;                              // token staging: row stride in halves (keeps the 2-byte scatter off one bank)
  using V = V2<Sh, TR, HF>;
  constexpr int P = Sh::P, P2 = Sh::P2, Cg = Sh::Cg, C2 = Sh::C2, F = Sh::F, F2 = Sh::F2, H = Sh::H, QC = V::QC;
  constexpr int LPC = V::LPC, CPW = V::CPW, NQ = V::NQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const int B = a.in.B;
  const float* __restrict__ th = a.theta;
  float* sTh = smem + V::oTh;
  float* sPool = smem + V::oPool;
  float* sZ = smem + V::oZ;
  float* sDh = smem + V::oDh;
  float* sSlab = smem + V::oSlab;
  VSTAMP_DECL;
  VSTAMP_RT(12);
  VSTAMP_W(0);
  // the head wave stays at priority 0 until barrier W: up to there it only issues its share of the gather, and at priority 3
  // those ~300 scalar + vector instructions came out of the issue slots of the two conv waves of its SIMD, the waves every
  // barrier of the prologue then waited for (16.16 -> 16.05 us per step).  Behind barrier W it is the youngest wave of its
  // SIMD with the longest dependent chain (fc1 -> fc2 -> softmax -> dh): priority 3 from there.
  if (wave != V::NB && wave >= 4) __builtin_amdgcn_s_setprio(1);      // static priority for the younger half: at equal priority the older
                                                          // wave of a SIMD wins every arbitration and the younger one trails it
  if constexpr (TOK) {   // token staging [2][128][TKS] halves behind the fixed regions: zero once (padding tokens / channels stay 0)
    for (int i = tid; i < 2 * 128 * TKS / 8; i += V::NT) reinterpret_cast<uint4*>(smem + V::oW2)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int boff = (a.in.cursor != nullptr) ? ((cint*)a.in.cursor)[0] * B : 0;     // epoch-plan offset of this batch
  // first patch's coordinates: requested before anything else (kernarg -> coordinates -> gather is the kernel's longest
  // dependent chain of memory round trips; the table staging below runs under it)
  int xn = 0, yn = 0;
  if (INMODE == 1 && (int)blockIdx.x < B) {
    xn = ((cint*)a.in.xy)[2 * (size_t)(boff + blockIdx.x)]; yn = ((cint*)a.in.xy)[2 * (size_t)(boff + blockIdx.x) + 1];
  }
  if constexpr (TR) {   // the slab rows' padding beyond the last parameter is copied out too: keep it zero
    if (tid < Sh::SLAB - Sh::NCONV) {                   // (< 32 elements)
#pragma unroll
      for (int c = 0; c < V::NCOPY; ++c) sSlab[c * Sh::SLAB + Sh::NCONV + tid] = 0.f;
    }
  }
  // ---- patch-invariant tables -> LDS by LDS-DMA (no registers, no wait here): the conv part of theta in parameter order
  //      (NTHP 1-KiB pieces) and the pooling profile in 16-byte-aligned rows (NPLP 256-byte pieces); piece q is issued by
  //      wave q % NW.  Each wave waits for its own pieces (stage_wait) before the barrier in front of the first gather.
  {
    constexpr int NTHP = (Sh::SLAB / 4 + 63) / 64, NPLP = (P * V::RSP + 63) / 64;
    constexpr int NDZP = (TR && !UNIT) ? (V::NDZ / 4 + 63) / 64 : 0;       // the dz index table (1-KiB pieces)
#pragma unroll
    for (int q = 0; q < NTHP + NPLP + NDZP; ++q) {
      if (q % V::NW != wave) continue;
      if (q >= NTHP + NPLP) {
        const int c4 = (q - NTHP - NPLP) * 64 + lane;
        dma_piece<16>(reinterpret_cast<const float*>(g_dzix_table<Sh>.v), 0, c4 < V::NDZ / 4 ? c4 * 16 : -1,
                      smem + V::oDzix + (q - NTHP - NPLP) * 256);
      } else if (q < NTHP) {
        const int c4 = q * 64 + lane;                      // 16-byte piece of theta
        dma_piece<16>(th, 0, c4 < Sh::SLAB / 4 ? (c4 < Sh::NCONV / 4 ? c4 : 0) * 16 : -1, sTh + q * 256);
      } else {
        const int t = (q - NTHP) * 64 + lane;              // element of the padded pooling table
        const int pr = t / V::RSP, pc = t - pr * V::RSP;
        dma_piece<4>(a.pool, 0, (t < P * V::RSP && pc < P) ? (pr * P + pc) * 4 : -1, sPool + (q - NTHP) * 64);
      }
    }
  }
  // (inline asm: the compiler must not count this against later loads — it orders visible LDS accesses behind the staging
  // pieces with counted waits of its own)
#define DMF_STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  static_assert(Sh::NCONV % 4 == 0, "conv parameters in whole 16-byte pieces");
  // This wave's share of the gather: pieces p = wave + k NW of the window image.  Lane l of piece p holds image floats
  // n = 256 p + 4 l .. + 3 = pixel n / CS, bands n % CS ..; its scene offset is ((row Wp + col) C + band) floats.
  // pieces [k0, k1) of this wave's share
  auto issue_gather = [&](int x, int y, int k0, int k1) {
    const float* base = a.in.sceneA + ((size_t)x * a.in.Wp + y) * V::CW;     // (32-bit words: HF scenes hold two bands in each)
    int l_ = lane;
    OPAQUE(l_);                                              // offsets are formed here, per patch: hoisted out of the patch loop
                                                             // they are spilled, and a scratch reload in front of a piece
                                                             // waits for every piece issued before it
    if constexpr (V::CS == V::CW && V::RS == P * V::CW && V::RS >= 256 && V::RS % 4 == 0) {
      // An image without any padding whose rows hold at least one piece (256 words): a piece lies in image row pr0 = 256 p / RS
      // up to lane lb and in row pr0 + 1 from there on, and a lane's scene offset is its image offset + row * D (D = scene row
      // stride - RS).  So everything per piece is SCALAR (row, boundary lane, the piece's base offset, which goes into the
      // instruction's scalar offset); per lane it is a compare, a select and a shift-add.  (Formed per lane with integer
      // divisions this was 12 vector instructions per piece, one of them a 64-bit multiply-add — 11 pieces per wave and
      // patch, inside the two phases where the vector issue ports are the bound.)  Every lane of a piece is inside the
      // image except in the last piece.
      const int D4 = (a.in.Wp * V::CW - V::RS) * 4;
#if defined(__HIP_DEVICE_COMPILE__)
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
      int w_ = wave;
      asm volatile("" : "+s"(w_));                           // (formed here, per call: hoisted, 11 pieces' scalars are spilled)
#pragma unroll
      for (int k = 0; k < V::NK; ++k) {
        if (k < k0 || k >= k1) continue;
        const int p = w_ + k * V::NW;
        if (p >= V::NPIECE) continue;                        // (scalar: this wave has no k-th piece)
        const int n0 = p << 8;
        const int pr0 = n0 / V::RS;
        const int lb = (V::RS - (n0 - pr0 * V::RS)) >> 2;    // first lane of image row pr0 + 1 (>= 64: the piece stays in row pr0)
        const int voff = (l_ << 4) + (l_ >= lb ? D4 : 0);
        auto lds = (__attribute__((address_space(3))) void*)(smem + V::oX + p * 256);
        if (p < V::NPIECE - 1 || 4 * l_ < P * V::RS - 256 * (V::NPIECE - 1))      // (a scalar branch around the last piece: no gain)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voff, 4 * n0 + D4 * pr0, 0, 0);
      }
#endif
      return;
    }
    const int rowskip = (a.in.Wp - P) * V::CW;               // words between the end of a window row and the start of the next
#pragma unroll
    for (int k = 0; k < V::NK; ++k) {
      if (k < k0 || k >= k1) continue;
      const int p = wave + k * V::NW;
      const int n = 256 * p + 4 * l_;
      int off;
      if constexpr (V::CS == V::CW) {                        // unpadded pixels: a window row is one contiguous run of the scene
        const int pr = n / V::RS, within = n - pr * V::RS;
        off = (within + pr * (rowskip + P * V::CW)) * 4;
        if (pr >= P || within >= P * V::CW) off = -1;
      } else {
        const int pr = n / V::RS, within = n - pr * V::RS;
        const int pc = within / V::CS, band = within - pc * V::CS;
        off = ((pr * a.in.Wp + pc) * V::CW + band) * 4;
        if (pr >= P || pc >= P || band >= V::CW) off = -1;
      }
      if (p >= V::NPIECE) off = -1;
      gather_piece(base, 0, off, smem + V::oX + (p < V::NPIECE ? p : 0) * 256);
    }
  };

  // S > 1: the whole aux patch image, 1-KiB pieces; lane l of piece q holds image floats n = 256 q + 4 l .. + 3 = row n / RL,
  // PHYSICAL chunk (n % RL) / 4 = logical chunk rotated by the patch row (see V2::RL)
  auto issue_aux_s = [&](int x, int y) {
    if constexpr (V::SX) {
      const float* base = a.in.sceneB + ((size_t)(Sh::S * x) * a.in.WpB + Sh::S * y) * C2;
      int l_ = lane;
      OPAQUE(l_);
#pragma unroll
      for (int q = 0; q < V::NAUXS; ++q) {
        if (q % V::NW != wave) continue;
        const int n = 256 * q + 4 * l_;
        const int row = n / V::RL, pc = (n - row * V::RL) >> 2;
        int lc = pc - row / Sh::S;
        lc += lc < 0 ? V::NC4 : 0;
        dma_piece<16>(base, 0, row < Sh::SP ? (row * a.in.WpB * C2 + 4 * lc) * 4 : -1, smem + V::oAuxS + q * 256);
      }
    }
  };
  // Wait until only this wave's window pieces are in flight, i.e. until everything it issued BEFORE them has landed: its
  // table-staging pieces (first patch) and its aux piece.  (vmcnt counts in issue order; the operand is an immediate, and a
  // wave issues NK pieces or — its last piece index beyond the image — NK - 1.)
  // EXTRA: vector-memory operations this wave issued between what is waited for and its window pieces (the aux row's loads)
  auto wait_older_than_gather = [&](auto extra) {
    constexpr int EXTRA = decltype(extra)::value;
    if constexpr (V::K0 < V::NK) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V::K0 + EXTRA) : "memory");       // (pieces k < K0 <= NK - 1 exist for every wave)
    } else {
      if (wave + (V::NK - 1) * V::NW < V::NPIECE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V::NK + EXTRA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V::NK - 1 + EXTRA) : "memory");
    }
  };

  // Behind barrier 2, all waves: dL/dz[i] = sum_j fc1.weight[j][i] dh[j] (4 lanes per i), then the workgroup's slab row =
  // sum of the quad copies of the UNIT gradients x dL/dz of each element's channel, in one coalesced pass (streaming stores:
  // next read by the reduce kernel).  A workgroup that walks several patches accumulates in its (L2-resident) global row.
  constexpr int NI = (Sh::SLAB / 4 + V::NT - 1) / V::NT;   // 16-byte slab pieces per thread
  // (channel (dz index) of each slab element: the staged DzixTable in LDS — kept in registers across the patch loop it is
  // spilled to scratch, and scratch lines are written back at kernel end)
  auto scale_and_store = [&](int it, int b) {
    int tid_ = tid;
    OPAQUE(tid_);            // (addresses formed here, not carried — spilled — across the patch loop)
    if constexpr (DENSE) {  // real gradients (the upstream maps were dense): sum of the quad copies into the workgroup's row
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int t = tid_ + i * V::NT;
        if (t < Sh::SLAB / 4) {
          float4 v = *reinterpret_cast<const float4*>(sSlab + 4 * t);
#pragma unroll
          for (int c = 1; c < V::NCOPY; ++c) {
            const float4 u = *reinterpret_cast<const float4*>(sSlab + c * Sh::SLAB + 4 * t);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
          }
          float* __restrict__ slab = a.slab + slab_index(blockIdx.x, 4 * t, gridDim.x);
          if (it > 0) {
            const float4 o = *reinterpret_cast<const float4*>(slab);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4*>(slab) = v;
        }
      }
      if (gridDim.x < (unsigned)B) LDS_BARRIER();        // more patches follow: the copies are rewritten by the next one
      return;
    }
    if constexpr (UNIT) {   // the patch's unit gradients (sum of the quad copies), unscaled, to its own row
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int t = tid_ + i * V::NT;
        if (t < Sh::SLAB / 4) {
          float4 v = *reinterpret_cast<const float4*>(sSlab + 4 * t);
#pragma unroll
          for (int c = 1; c < V::NCOPY; ++c) {
            const float4 u = *reinterpret_cast<const float4*>(sSlab + c * Sh::SLAB + 4 * t);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
          }
          *reinterpret_cast<float4*>(a.slab + (size_t)b * Sh::SLAB + 4 * t) = v;
        }
      }
      LDS_BARRIER();                                     // the copies are rewritten by the next patch
      return;
    }
    float* sDz = smem + V::oDz;
    for (int t = tid_; t < 4 * F2; t += V::NT) {         // (whole quads: NT and 4 F2 are multiples of 4)
      const int i = t >> 2, m = t & 3;
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 dhv = *reinterpret_cast<const float4*>(sDh + 16 * m + 4 * q);
        const float4 wv = *reinterpret_cast<const float4*>(smem + V::oW1T + i * H + 16 * m + 4 * q);
        d = fmaf(wv.x, dhv.x, fmaf(wv.y, dhv.y, fmaf(wv.z, dhv.z, fmaf(wv.w, dhv.w, d))));
      }
      d = quad_sum(d);
      if (m == 0) sDz[i] = d;
    }
    LDS_BARRIER();                                       // barrier 3: dz complete
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = tid_ + i * V::NT;
      if (t < Sh::SLAB / 4) {
        float4 v = *reinterpret_cast<const float4*>(sSlab + 4 * t);
#pragma unroll
        for (int c = 1; c < V::NCOPY; ++c) {
          const float4 u = *reinterpret_cast<const float4*>(sSlab + c * Sh::SLAB + 4 * t);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int dx = reinterpret_cast<const int*>(smem + V::oDzix)[t];
        v.x *= sDz[dx & 255]; v.y *= sDz[(dx >> 8) & 255]; v.z *= sDz[(dx >> 16) & 255]; v.w *= sDz[(dx >> 24) & 255];
        float* __restrict__ slab = a.slab + slab_index(blockIdx.x, 4 * t, gridDim.x);      // (piece-major: dmf_shapes.h)
        if (it > 0) {
          const float4 o = *reinterpret_cast<const float4*>(slab);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
__builtin_nontemporal_store(v.x, slab);       // (nt beats sc1 / sc0 sc1 / plain stores here: 16.80 vs 16.98 us per step)
        __builtin_nontemporal_store(v.y, slab + 1);
        __builtin_nontemporal_store(v.z, slab + 2);
        __builtin_nontemporal_store(v.w, slab + 3);
      }
    }
    if (gridDim.x < (unsigned)B) LDS_BARRIER();          // more patches follow: the copies are rewritten by the next one
  };
  static_assert(F2 <= 255, "dz index in 8 bits");
  // MODE_TOKENS, behind barrier 1, all waves: the two staged [128][64] bf16 token maps -> global in 16-byte pieces (tokens
  // beyond P2 and channels beyond F are the zeros the staging was filled with), then a barrier: the next patch rewrites it
  auto token_out = [&](int b) {
    const unsigned short* sTok = reinterpret_cast<const unsigned short*>(smem + V::oW2);
    for (int i = tid; i < 2 * 128 * 8; i += V::NT) {
      const int mp = i >> 10, rem = i & 1023, t = rem >> 3, pc8 = rem & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(sTok + mp * 128 * TKS + t * TKS + pc8 * 8);
      unsigned short* dst = (mp ? a.tokB : a.tokA) + ((size_t)b * 128 + t) * 64 + pc8 * 8;
      *reinterpret_cast<uint4*>(dst) = v;
    }
    LDS_BARRIER();
  };
  (void)token_out;

  if (wave < V::NB) {
    // =============================================================================== conv wavefronts
    // A lane's roles are re-derived from an opaque copy of its lane id at the top of every phase: kept live across the
    // whole patch loop they are spilled around the register-hungry phases, and a scratch reload in front of the gather
    // costs a wait for every outstanding load.
    //   seg, r : channel slot of the wave, patch row;  qd: which quad partial of its channel this lane's quad forms;
    //   lead   : one lane per quad owns the quad's results
    // Padding lanes are the zero padding around the patch rows: their Y1 is forced to 0 (ReLU limit 0 instead of +inf)
    // and their depthwise bias is hugely negative, so their ReLU gates are closed (pooling weight 0, no gradient) whatever
    // their (clamped-row) inputs are.
#define DMF_ROLES()                                                                  \
    int l_ = lane; OPAQUE(l_);                                                     \
    const int seg = l_ / LPC, r = l_ - seg * LPC;                                  \
    const bool vch = seg < CPW;                                                    \
    const int f = CPW * wave + (vch ? seg : CPW - 1);                              \
    const bool act = vch && r < P;                                                 \
    const int rc = r < P ? r : P - 1;                                              \
    const int qd = r >> 2;                                                         \
    const bool lead = vch && (r & 3) == 0;                                         \
    const float lim = act ? INFINITY : 0.f;                                        \
    const float* xr = smem + V::oX + rc * V::RS + (f / Sh::M) * V::CgW;             \
    float* sl = sSlab + qd * Sh::SLAB;                                             \
    (void)lead; (void)lim; (void)xr; (void)sl; (void)qd; (void)act
    VSTAMP(1);
    int it = 0;
    for (int b = blockIdx.x; b < B; b += gridDim.x, ++it) {
      // ------------------------------------------------------------------ aux row -> registers, window -> LDS
      float y2b[P];                                        // MODE_TOKENS: this lane's row of the aux feature map
      float ax[V::AR0];
      // gather mode: the aux row arrives as whole 16-byte vectors + a scalar tail, each ONE asm output (LDS reads on the first
      // patch, global loads later), unpacked into ax[] only behind the wait: element moves of a vector result are ordinary
      // instructions the compiler may place in front of a wait it cannot see
      constexpr int NV4 = V::AR0 / 4, NR1 = V::AR0 % 4;
      f4v axv[NV4 > 0 ? NV4 : 1];
      float axt[NR1 > 0 ? NR1 : 1];
      float zb;
      {
      DMF_ROLES();
      const unsigned aTh = lds_addr(sTh) + 4u * (unsigned)f;           // hidden-read bases (bytes)
      const unsigned aPool = lds_addr(sPool) + 4u * (unsigned)(rc * V::RSP);
      float4 ddbv[RS4 / 4];                                // MODE_DENSE: this lane's row of dL/dY2 of the aux branch (in place
      if constexpr (DENSE) {                               // of the pooling profile); requested ahead of the gather
        const float* ddb = a.dYb + ((size_t)b * F + f) * (P * RS4) + rc * RS4;
#pragma unroll
        for (int q = 0; q < RS4 / 4; ++q) ddbv[q] = *reinterpret_cast<const float4*>(ddb + 4 * q);
      }
      int gx = 0, gy = 0;                                  // this patch's coordinates, for the pieces issued from the aux phase
      if constexpr (INMODE == 1) {
        const int x = xn, y = yn;
        gx = x; gy = y;
        // (the scheduler must not sink the gather below the aux phase, nor hoist that phase above it)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (V::SX) {
          // S > 1: aux image and window of every patch through LDS, complete before the aux phase (the windows of these
          // shapes are a few KiB: nothing to overlap) — vmcnt(0) also covers the staged tables of the first patch
          issue_aux_s(x, y);
          issue_gather(x, y, 0, V::NK);
          __builtin_amdgcn_sched_barrier(0);
          VSTAMP(11);
          __syncthreads();
        } else {
          // the lane's aux row straight into registers, ahead of the window pieces in the memory pipeline (every patch; the
          // first patch used to fetch an aux IMAGE by LDS-DMA and read rows out of LDS behind barrier X).  Inline asm: loads
          // the compiler knows of would make it wait for them (and for every piece behind them) at the next LDS access it
          // can see.  Their results are pinned behind the wait (HIDDEN_USE): an asm output nobody reads gets a register the
          // compiler reuses at once, and the result landing late would overwrite whatever lives there by then.
          const float* srcB = a.in.sceneB + ((size_t)(x + rc) * a.in.WpB + y) * C2;
#pragma unroll
          for (int j = 0; j < NV4; ++j) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(axv[j]) : "v"(srcB), "n"(16 * j) : "memory");
#pragma unroll
          for (int i = 0; i < NR1; ++i) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(axt[i]) : "v"(srcB), "n"(4 * (4 * NV4 + i)) : "memory");
          issue_gather(x, y, 0, V::K0);
          __builtin_amdgcn_sched_barrier(0);
          if (it == 0) {
            // barrier X: the staged tables (theta, pooling profile: LDS-DMA of ALL waves) are complete.  A wave waits for ITS
            // staging pieces only — everything it issued before the aux row loads and the window pieces above; it used to
            // wait here for an LDS-DMA aux image too, which depends on the coordinates: the barrier stood at ~3.4 K cycles.
            VSTAMP(11);
            wait_older_than_gather(std::integral_constant<int, NV4 + NR1>());
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
        }
        {   // the next patch's coordinates, a whole patch ahead of their use
          const int bn = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;
          xn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn)]; yn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn) + 1];
        }
      } else {
        // materialised band-major patches (the reference dataloader's tensors; test / drop-in path)
        if constexpr (V::SX) {
          // the aux patch [C2][S P][S P] (band-major) into the rotated LDS image
          const float* __restrict__ srcB = a.in.b + (size_t)(boff + b) * C2 * Sh::PB;
          for (int e = tid; e < C2 * Sh::PB; e += V::NB * 64) {
            const int k = e / Sh::PB, pix = e - k * Sh::PB, row = pix / Sh::SP, col = pix - row * Sh::SP;
            const int m = col * C2 + k;                    // logical float index inside the image row
            int pc = (m >> 2) + row / Sh::S;
            pc -= pc >= V::NC4 ? V::NC4 : 0;
            smem[V::oAuxS + row * V::RL + 4 * pc + (m & 3)] = srcB[e];
          }
        } else {
          const float* __restrict__ srcB = a.in.b + (size_t)(boff + b) * C2 * P2;
#pragma unroll
          for (int c = 0; c < P; ++c)
#pragma unroll
            for (int k = 0; k < C2; ++k) ax[c * C2 + k] = srcB[k * P2 + rc * P + c];
        }
        if (it == 0) { DMF_STAGE_WAIT(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // barrier X
        const float* __restrict__ srcA = a.in.a + (size_t)(boff + b) * Sh::C * P2;
        for (int e = tid; e < Sh::C * P2; e += V::NB * 64) {
          const int cb = e / P2, pix = e - cb * P2;
          if constexpr (HF) reinterpret_cast<_Float16*>(smem + V::oX + (pix / P) * V::RS + (pix % P) * V::CS)[cb] = (_Float16)srcA[e];
          else smem[V::oX + (pix / P) * V::RS + (pix % P) * V::CS + cb] = srcA[e];
        }
        if constexpr (V::SX) __syncthreads();                // the aux image (written above by all conv waves) is read next
      }
      VSTAMP(2);

      // ------------------------------------------------------------------ aux branch, under the window gather
      constexpr int TB = Sh::TB;                             // taps of lift_b: C2 S S
      float dwb[9], dbb = 0.f, dwl[TB], dbl = 0.f;
      {
        float w2b[9], wl[TB], pw[P], b2b, bl;
        // (the channel-dependent part of an offset is in the base address, the rest is an instruction immediate)
#pragma unroll
        for (int k = 0; k < 9; ++k) w2b[k] = hidden_read(aTh + 4u * (unsigned)(f * 8), (Sh::oB2w + k) * 4);          // oB2w + 9f + k
        b2b = hidden_read(aTh, Sh::oB2b * 4);
        bl = hidden_read(aTh, Sh::oB1b * 4);
#pragma unroll
        for (int k = 0; k < TB; ++k) wl[k] = hidden_read(aTh + 4u * (unsigned)(f * (TB - 1)), (Sh::oB1w + k) * 4);   // oB1w + TB f + k
#pragma unroll
        for (int c = 0; c < P; ++c) pw[c] = hidden_read(aPool, c * 4);
        hidden_wait();
        if constexpr (DENSE) {
#pragma unroll
          for (int c = 0; c < P; ++c) { const float4 v = ddbv[c / 4]; pw[c] = (c & 3) == 0 ? v.x : (c & 3) == 1 ? v.y : (c & 3) == 2 ? v.z : v.w; }
        }
        if constexpr (INMODE == 1 && !V::SX) {
          wait_older_than_gather(std::integral_constant<int, 0>());      // the aux row's loads
#pragma unroll
          for (int j = 0; j < NV4; ++j) {
            HIDDEN_USE(axv[j]);
            ax[4 * j] = axv[j].x; ax[4 * j + 1] = axv[j].y; ax[4 * j + 2] = axv[j].z; ax[4 * j + 3] = axv[j].w;
          }
#pragma unroll
          for (int i = 0; i < NR1; ++i) { HIDDEN_USE(axt[i]); ax[4 * NV4 + i] = axt[i]; }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) HIDDEN_USE(w2b[k]);
        HIDDEN_USE(b2b); HIDDEN_USE(bl);
#pragma unroll
        for (int k = 0; k < TB; ++k) HIDDEN_USE(wl[k]);
        if constexpr (!DENSE) {
#pragma unroll
          for (int c = 0; c < P; ++c) HIDDEN_USE(pw[c]);
        }

        float y1b[P], dyb[P];
        // S > 1: logical float m = 4 j + e of image row S rc + u is pixel column m / C2, band m % C2 = output column
        // (m / C2) / S, tap ((m % C2) S + u) S + (m / C2) % S; its 16-byte chunk j sits at physical chunk (j + rc) mod NC4
        const float* auxr = smem + V::oAuxS + (Sh::S * rc) * V::RL;
        // physical chunk of logical chunk j in this lane's rows (one address per j, the S rows are immediate offsets apart)
        auto chunk_of = [&](int j) -> int {
          if constexpr ((V::NC4 & (V::NC4 - 1)) == 0) return (j + rc) & (V::NC4 - 1);
          else { int pc = j + rc; pc -= pc >= V::NC4 ? V::NC4 : 0; return pc; }
        };
        if constexpr (V::SX) {
#pragma unroll
          for (int c = 0; c < P; ++c) y1b[c] = bl;
#pragma unroll
          for (int j = 0; j < V::NC4; ++j) {
            const float* ap = auxr + 4 * chunk_of(j);
#pragma unroll
            for (int u = 0; u < Sh::S; ++u) {
              const float4 a4 = *reinterpret_cast<const float4*>(ap + u * V::RL);
              const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int m = 4 * j + e, cp = m / C2, k = m % C2;
                y1b[cp / Sh::S] = fmaf(wl[(k * Sh::S + u) * Sh::S + cp % Sh::S], av[e], y1b[cp / Sh::S]);
              }
            }
          }
#pragma unroll
          for (int c = 0; c < P; ++c) y1b[c] = relu_lim(y1b[c], lim);
        } else {
#pragma unroll
          for (int c = 0; c < P; ++c) {
            float v = bl;
#pragma unroll
            for (int k = 0; k < C2; ++k) v = fmaf(wl[k], ax[c * C2 + k], v);
            y1b[c] = relu_lim(v, lim);
          }
        }
        // (the rest of this wave's window pieces leaves from here, one per output column)
        auto rest = [&](int c) {
          if constexpr (INMODE == 1 && V::NREST > 0) {
            if (c < V::NREST) {
              __builtin_amdgcn_sched_barrier(0);
              issue_gather(gx, gy, V::K0 + c, V::K0 + c + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        conv_row<P, TR, V::ROWDPP>(y1b, w2b, act ? b2b : -1e30f, pw, zb, dwb, dbb, dyb, rest, TOK ? y2b : nullptr);
        zb = quad_sum(zb);
        if constexpr (TR) {
#pragma unroll
          for (int k = 0; k < TB; ++k) dwl[k] = 0.f;
#pragma unroll
          for (int c = 0; c < P; ++c) dbl += dyb[c];
          if constexpr (V::SX) {
#pragma unroll
            for (int j = 0; j < V::NC4; ++j) {
              const float* ap = auxr + 4 * chunk_of(j);
#pragma unroll
              for (int u = 0; u < Sh::S; ++u) {
                const float4 a4 = *reinterpret_cast<const float4*>(ap + u * V::RL);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int m = 4 * j + e, cp = m / C2, k = m % C2, t = (k * Sh::S + u) * Sh::S + cp % Sh::S;
                  dwl[t] = fmaf(dyb[cp / Sh::S], av[e], dwl[t]);
                }
              }
            }
          } else {
#pragma unroll
            for (int c = 0; c < P; ++c)
#pragma unroll
              for (int k = 0; k < C2; ++k) dwl[k] = fmaf(dyb[c], ax[c * C2 + k], dwl[k]);
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) dwb[k] = quad_sum(dwb[k]);
          dbb = quad_sum(dbb);
          dbl = quad_sum(dbl);
#pragma unroll
          for (int k = 0; k < TB; ++k) dwl[k] = quad_sum(dwl[k]);
        }
      }
      VSTAMP(3);
      // the aux branch's unit gradients go to this quad's copy of the slab row (the first LDS access the compiler sees
      // behind the gather: it waits for the wave's own LDS-DMA here), then barrier W: every wave's pieces have landed
      if constexpr (TR) {
        if (lead) {
#pragma unroll
          for (int k = 0; k < 9; ++k) sl[Sh::oB2w + f * 9 + k] = dwb[k];
          sl[Sh::oB2b + f] = dbb;
          sl[Sh::oB1b + f] = dbl;
#pragma unroll
          for (int k = 0; k < TB; ++k) sl[Sh::oB1w + f * TB + k] = dwl[k];
        }
      }
      }
      __syncthreads();                                       // barrier W (vmcnt(0) + s_barrier)
      VSTAMP(4);
      DMF_ROLES();

      // ------------------------------------------------------------------ primary branch
      float w2a[9], pw[P], b2a;
#pragma unroll
      for (int k = 0; k < 9; ++k) w2a[k] = sTh[Sh::oA2w + f * 9 + k];
      b2a = sTh[Sh::oA2b + f];
#pragma unroll
      for (int q = 0; q < V::RSP / 4; ++q) {
        // (MODE_DENSE: the lane's row of dL/dY2 of the primary branch instead of the pooling profile; RSP == RS4)
        const float4 v = DENSE ? *reinterpret_cast<const float4*>(a.dYa + ((size_t)b * F + f) * (P * RS4) + rc * RS4 + 4 * q)
                               : *reinterpret_cast<const float4*>(sPool + rc * V::RSP + 4 * q);
        if (4 * q < P) pw[4 * q] = v.x;
        if (4 * q + 1 < P) pw[4 * q + 1] = v.y;
        if (4 * q + 2 < P) pw[4 * q + 2] = v.z;
        if (4 * q + 3 < P) pw[4 * q + 3] = v.w;
      }
      float y1a[P];
      if constexpr (HF) {   // spec_a on fp16 operands: two bands per word, v_dot2_f32_f16 (exact products, fp32 accumulate)
        h2 w1h[Cg / 2];
#pragma unroll
        for (int q = 0; q < QC; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(sTh + Sh::oA1w + f * Cg + 4 * q);
          w1h[2 * q] = (h2){(_Float16)v.x, (_Float16)v.y};
          w1h[2 * q + 1] = (h2){(_Float16)v.z, (_Float16)v.w};
        }
        const float b1 = sTh[Sh::oA1b + f];
        float ap[P];
#pragma unroll
        for (int c = 0; c < P; ++c) ap[c] = b1;
#pragma unroll
        for (int c = 0; c < P; ++c) {
          if (c % 3 == 0 && c > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float2 xv = *reinterpret_cast<const float2*>(xr + c * V::CS + 2 * q);
            ap[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, xv.x), w1h[2 * q], ap[c], false);
            ap[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, xv.y), w1h[2 * q + 1], ap[c], false);
          }
        }
#pragma unroll
        for (int c = 0; c < P; ++c) y1a[c] = relu_lim(ap[c], lim);
      } else {   // spec_a: even / odd bands of the group in the two halves of a packed accumulator
        v2f w1p[Cg / 2];
#pragma unroll
        for (int q = 0; q < QC; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(sTh + Sh::oA1w + f * Cg + 4 * q);
          w1p[2 * q] = (v2f){v.x, v.y};
          w1p[2 * q + 1] = (v2f){v.z, v.w};
        }
        const float b1 = sTh[Sh::oA1b + f];
        v2f ap[P];
#pragma unroll
        for (int c = 0; c < P; ++c) ap[c] = (v2f){b1, 0.f};
#pragma unroll
        for (int c = 0; c < P; ++c) {
          if (c % 3 == 0 && c > 0) __builtin_amdgcn_sched_barrier(0);   // bounds the window reads the scheduler keeps in flight (registers)
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c * V::CS + 4 * q);
            ap[c] = pk_fma(w1p[2 * q], (v2f){xv.x, xv.y}, ap[c]);
            ap[c] = pk_fma(w1p[2 * q + 1], (v2f){xv.z, xv.w}, ap[c]);
          }
        }
#pragma unroll
        for (int c = 0; c < P; ++c) y1a[c] = relu_lim(ap[c].x + ap[c].y, lim);
      }
      float za, dwa[9], dba = 0.f, dya[P];
      ConvRows<P> ta;
      float y2a[P];
      conv_row_fwd<P, V::ROWDPP>(y1a, w2a, act ? b2a : -1e30f, pw, ta, za, NoHook(), TOK ? y2a : nullptr);
      if constexpr (TOK) {   // both feature-map rows -> the bf16 staging [map][token = pixel][channel]
        if (act) {
          unsigned short* sTok = reinterpret_cast<unsigned short*>(smem + V::oW2);
#pragma unroll
          for (int c = 0; c < P; ++c) {
            const int t = r * P + c;
            sTok[t * TKS + f] = __builtin_bit_cast(unsigned short, (__bf16)y2a[c]);
            sTok[128 * TKS + t * TKS + f] = __builtin_bit_cast(unsigned short, (__bf16)y2b[c]);
          }
        }
      }
      za = quad_sum(za);
      {   // pooled features of the wave's channels: quad partials -> wave-private scratch -> one lane per value sums NQ of them
        float* zp = smem + V::oZP + wave * V::ZPW;
        if (lead) { zp[(2 * seg) * NQ + qd] = za; zp[(2 * seg + 1) * NQ + qd] = zb; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2 * CPW) {
          float zs = 0.f;
#pragma unroll
          for (int q = 0; q < NQ; ++q) zs += zp[lane * NQ + q];
          float* zbuf = sZ + (TR ? 0 : (it & 1)) * V::ZS;
          zbuf[((lane & 1) ? F : 0) + CPW * wave + (lane >> 1)] = zs;
        }
      }
      LDS_BARRIER();                                     // barrier 1: pooled features complete
      VSTAMP(5);
      if constexpr (TOK) { token_out(b); continue; }
      if constexpr (!TR) continue;
      // spat_a's unit backward runs BEHIND barrier 1: the head starts a conv backward earlier, and its chain of LDS round
      // trips (fc1 -> fc2 -> softmax -> dh) begins while the conv waves execute pure vector work instead of queueing behind
      // their window reads
      conv_row_bwd<P, V::ROWDPP>(y1a, w2a, ta, dwa, dba, dya);

      // ------------------------------------------------------------------ unit gradients of spec_a from the still-resident window
      float acc[Cg], db1 = 0.f;
      if constexpr (HF) {   // fp16 window x fp32 gradient, fp32 accumulate (v_fma_mix_f32)
#pragma unroll
        for (int j = 0; j < Cg; ++j) acc[j] = 0.f;
#pragma unroll
        for (int c = 0; c < P; ++c) {
          if (c % 3 == 0 && c > 0) __builtin_amdgcn_sched_barrier(0);
          db1 += dya[c];
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float2 xv = *reinterpret_cast<const float2*>(xr + c * V::CS + 2 * q);
            acc[4 * q] = fma_mix_lo(xv.x, dya[c], acc[4 * q]);
            acc[4 * q + 1] = fma_mix_hi(xv.x, dya[c], acc[4 * q + 1]);
            acc[4 * q + 2] = fma_mix_lo(xv.y, dya[c], acc[4 * q + 2]);
            acc[4 * q + 3] = fma_mix_hi(xv.y, dya[c], acc[4 * q + 3]);
          }
        }
      } else {
        v2f gp[Cg / 2];
#pragma unroll
        for (int j = 0; j < Cg / 2; ++j) gp[j] = (v2f){0.f, 0.f};
#pragma unroll
        for (int c = 0; c < P; ++c) {
          if (c % 3 == 0 && c > 0) __builtin_amdgcn_sched_barrier(0);
          db1 += dya[c];
          const v2f d2 = (v2f){dya[c], dya[c]};
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c * V::CS + 4 * q);
            gp[2 * q] = pk_fma(d2, (v2f){xv.x, xv.y}, gp[2 * q]);
            gp[2 * q + 1] = pk_fma(d2, (v2f){xv.z, xv.w}, gp[2 * q + 1]);
          }
        }
#pragma unroll
        for (int j = 0; j < Cg / 2; ++j) { acc[2 * j] = gp[j].x; acc[2 * j + 1] = gp[j].y; }
      }
      VSTAMP(6);
#pragma unroll
      for (int j = 0; j < Cg; ++j) acc[j] = quad_sum(acc[j]);
#pragma unroll
      for (int k = 0; k < 9; ++k) dwa[k] = quad_sum(dwa[k]);
      dba = quad_sum(dba);
      db1 = quad_sum(db1);
      if (lead) {   // the primary branch's unit gradients -> this quad's slab copy
        float* so = sl + Sh::oA1w + f * Cg;
#pragma unroll
        for (int q = 0; q < QC; ++q) *reinterpret_cast<float4*>(so + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        sl[Sh::oA1b + f] = db1;
#pragma unroll
        for (int k = 0; k < 9; ++k) sl[Sh::oA2w + f * 9 + k] = dwa[k];
        sl[Sh::oA2b + f] = dba;
      }
      VSTAMP(7);
      LDS_BARRIER();                                     // barrier 2: dh and the unit gradients complete
      VSTAMP(8);
      scale_and_store(it, b);
      VSTAMP(9);
    }
  } else {
    // =============================================================================== head wavefront
    float* sW1T = smem + V::oW1T;
    float* sW2 = smem + V::oW2;
    float* sH = smem + V::oHv;
    float* sDl = smem + V::oDl;
    const int K4 = (K + 3) & ~3;
    float4 w1r[F2 / 4];                // fc1.weight row `lane`, in registers for fc1 (loaded behind the first gather issue)
    float bh = 0.f, bk = 0.f;
    VSTAMP(1);

    int it = 0;
    for (int b = blockIdx.x; b < B; b += gridDim.x, ++it) {
      VSTAMP(11);
      if constexpr (INMODE == 1) {
        const int x = xn, y = yn;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (V::SX) issue_aux_s(x, y);
        issue_gather(x, y, 0, V::K0);                    // this wave's share of the window: the first pieces ...
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (V::SX) {
          __syncthreads();                               // (every patch: the aux image is complete, see the conv waves)
        } else if (it == 0) {
          wait_older_than_gather(std::integral_constant<int, 0>());
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // barrier X (first patch): staged tables complete
        }
        issue_gather(x, y, V::K0, V::NK);                // ... and the rest (nothing else to do until barrier 1)
        __builtin_amdgcn_sched_barrier(0);
        const int bn = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;
        xn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn)]; yn = ((cint*)a.in.xy)[2 * (size_t)(boff + bn) + 1];
      } else {
        if (it == 0) { DMF_STAGE_WAIT(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // barrier X
        if constexpr (V::SX) __syncthreads();            // (the conv waves' aux image, see there)
      }
      int label = 0;
      float dlx = 0.f;
      if (MODE == MODE_TRAIN) {
        label = ((cint*)a.labels)[boff + b];
        label = label < 0 ? 0 : (label >= K ? K - 1 : label);
      }
      if (MODE == MODE_BWD) dlx = lane < K ? a.dlogits[(size_t)b * K + lane] : 0.f;
      __syncthreads();                                   // barrier W: window complete (this wave's pieces included)
      VSTAMP(4);
      __builtin_amdgcn_s_setprio(3);
      if (it == 0 && !TOK && !DENSE) {   // head tables, behind the gather: nobody needs them before barrier 1, and 20 row-strided loads inside
                       // the gather stream would delay every wave's pieces (one memory pipeline per CU)
#pragma unroll
        for (int q = 0; q < F2 / 4; ++q) w1r[q] = *reinterpret_cast<const float4*>(th + Sh::oFc1w + lane * F2 + 4 * q);
        bh = th[Sh::oFc1b + lane];
        bk = lane < K ? th[Sh::oFc2w + K * H + lane] : 0.f;
        for (int i0 = 0; i0 < K4 / 4; i0 += 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = (i0 + u) * 64 + lane, k = idx >> 4;
            v[u] = (i0 + u < K4 / 4 && k < K) ? *reinterpret_cast<const float4*>(th + Sh::oFc2w + 4 * idx) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = (i0 + u) * 64 + lane, k = idx >> 4, c4 = idx & 15;
            if (i0 + u < K4 / 4) *reinterpret_cast<float4*>(sW2 + k * V::W2S + 4 * c4) = v[u];
          }
        }
        if constexpr (TR && !UNIT) {
#pragma unroll
          for (int q = 0; q < F2 / 4; ++q) {
            sW1T[(4 * q) * H + lane] = w1r[q].x;
            sW1T[(4 * q + 1) * H + lane] = w1r[q].y;
            sW1T[(4 * q + 2) * H + lane] = w1r[q].z;
            sW1T[(4 * q + 3) * H + lane] = w1r[q].w;
          }
        }
      }
      LDS_BARRIER();                                     // barrier 1: pooled features complete
      VSTAMP(5);
      if constexpr (TOK) {   // pooled features of the conv stages (before attention), then this wave's share of the token copy
        const float* zb_ = sZ + (it & 1) * V::ZS;
        if (lane < F2) a.zout[(size_t)b * F2 + lane] = zb_[lane];
        if (F2 > 64 && 64 + lane < F2) a.zout[(size_t)b * F2 + 64 + lane] = zb_[64 + lane];
        token_out(b);
        continue;
      }
      if constexpr (DENSE) {   // no head: the gradient maps came from the attention kernel
        LDS_BARRIER();                                   // barrier 2
        scale_and_store(it, b);
        continue;
      }
      const float* zbuf = sZ + (TR ? 0 : (it & 1)) * V::ZS;
      // fc1 + ReLU: lane j, its weight row in registers, z broadcast from LDS
      float h;
      {
        // four running sums, packed as the register pairs a 16-byte read delivers ((x, y), (z, w)): left to itself the
        // compiler pairs (x, z) and (y, w) and spends six moves per read on the shuffle
        v2f s01 = (v2f){bh, 0.f}, s23 = (v2f){0.f, 0.f};
#pragma unroll
        for (int q = 0; q < F2 / 4; ++q) {
          const float4 zv = *reinterpret_cast<const float4*>(zbuf + 4 * q);
          s01 = pk_fma((v2f){w1r[q].x, w1r[q].y}, (v2f){zv.x, zv.y}, s01);
          s23 = pk_fma((v2f){w1r[q].z, w1r[q].w}, (v2f){zv.z, zv.w}, s23);
        }
        h = fmaxf((s01.x + s01.y) + (s23.x + s23.y), 0.f);
      }
      sH[lane] = h;
      VSTAMP(2);
      float zo0 = 0.f, zo1 = 0.f;                         // this patch's pooled features, for the gradient reduce
      if constexpr (TR) {
        zo0 = zbuf[lane < F2 ? lane : F2 - 1];
        if (F2 > 64) zo1 = zbuf[64 + lane < F2 ? 64 + lane : F2 - 1];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // fc2: lane k (rows beyond K are zero rows or clamped; masked below)
      float lg;
      {
        const int kr = lane < K4 ? lane : K4 - 1;
        v2f s01 = (v2f){bk, 0.f}, s23 = (v2f){0.f, 0.f};
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
          const float4 wv = *reinterpret_cast<const float4*>(sW2 + kr * V::W2S + 4 * q);
          const float4 hv = *reinterpret_cast<const float4*>(sH + 4 * q);
          s01 = pk_fma((v2f){wv.x, wv.y}, (v2f){hv.x, hv.y}, s01);
          s23 = pk_fma((v2f){wv.z, wv.w}, (v2f){hv.z, hv.w}, s23);
        }
        lg = lane < K ? (s01.x + s01.y) + (s23.x + s23.y) : -INFINITY;
      }
      VSTAMP(3);
      const float mx = wave_max_dpp(lg);
      const unsigned long long bal = __ballot(lg == mx);
      const int pred_b = __ffsll((long long)bal) - 1;              // first maximal index, as torch.max
      float loss_b = 0.f, dl = 0.f, dh = 0.f;
      if constexpr (TR && !UNIT) {
        if (MODE == MODE_TRAIN) {
          const float e = lane < K ? __expf(lg - mx) : 0.f;
          const float se = wave_sum_dpp(e);
          const float ls = a.loss_scale * (a.scaler != nullptr ? a.scaler[0] : 1.f);
          dl = lane < K ? (e / se - (lane == label ? 1.f : 0.f)) * ls : 0.f;
          const float lgt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lg), __builtin_amdgcn_readfirstlane(label)));
          loss_b = (mx + __logf(se)) - lgt;
        } else {
          dl = dlx;
        }
        sDl[lane] = dl;
        VSTAMP(6);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // dh[j] = relu'(h[j]) * sum_k W2[k][j] dl[k]: lane j
        {
          float s0 = 0.f, s1 = 0.f;
          for (int k = 0; k < K4; k += 4) {
            const float4 dv = *reinterpret_cast<const float4*>(sDl + k);
            s0 = fmaf(sW2[k * V::W2S + lane], dv.x, s0);
            s1 = fmaf(sW2[(k + 1) * V::W2S + lane], dv.y, s1);
            s0 = fmaf(sW2[(k + 2) * V::W2S + lane], dv.z, s0);
            s1 = fmaf(sW2[(k + 3) * V::W2S + lane], dv.w, s1);
          }
          dh = h > 0.f ? s0 + s1 : 0.f;
        }
        sDh[lane] = dh;
      }
      VSTAMP(7);
      if constexpr (TR) LDS_BARRIER();                   // barrier 2: dh and the unit gradients complete
      VSTAMP(8);
      if constexpr (TR) scale_and_store(it, b);
      // this patch's head vectors for the gradient reduce
      if ((MODE != MODE_BWD || a.logits != nullptr) && lane < K) a.logits[(size_t)b * K + lane] = lg;
      if (a.pred != nullptr && lane == 0) a.pred[b] = pred_b;
      if constexpr (MODE == MODE_FWD) {
        // evaluation with labels (dmf_forward_ce: the validation pass, mainsolver.py:62-76): the per-patch cross-entropy, by the
        // training kernel's own formula
        if (a.labels != nullptr && a.loss != nullptr) {
          int label = ((cint*)a.labels)[boff + b];
          label = label < 0 ? 0 : (label >= K ? K - 1 : label);
          const float e = lane < K ? __expf(lg - mx) : 0.f;
          const float se = wave_sum_dpp(e);
          const float lgt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lg), __builtin_amdgcn_readfirstlane(label)));
          if (lane == 0) a.loss[b] = (mx + __logf(se)) - lgt;
        }
      }
      if constexpr (TR) {
        if (MODE == MODE_TRAIN && lane == 0) a.loss[b] = loss_b;
        const size_t hv = hv_index(b, lane, B);            // strip-major head vectors (dmf_shapes.h)
        a.ws_h[hv] = h;
        if constexpr (!UNIT) {
          a.ws_dh[hv] = dh;
          a.ws_dl[hv] = dl;
        }
        if (lane < F2) a.ws_z[hv] = zo0;
        if (F2 > 64 && 64 + lane < F2) a.ws_z[hv_index(b, 64 + lane, B)] = zo1;
      }
      VSTAMP(9);
    }
    // the device-side ADAM step count advances once per launch — here, at the very end of one wave's work: in front of the
    // coordinate request (where it used to be) its null test made every wave wait for a kernel argument that is not preloaded
    if ((MODE == MODE_TRAIN || UNIT || DENSE) && a.adam_step != nullptr && blockIdx.x == 0 && lane == 0) *a.adam_step += 1;
  }
  VSTAMP_W(10);
  VSTAMP_RT(13);
  VSTAMP_DUMP();
}

// ---------------------------------------------------------------------------------------- unit-gradient step, second half
// For a loss that couples the whole batch (qua_loss, train/loss_function.py:15-76) the patch kernel cannot form dL/dlogits
// itself.  MODE_UNIT leaves, per patch, the head vectors and one row of UNIT conv gradients; once the loss kernel has
// produced dL/dlogits this kernel finishes the backward WITHOUT touching the patches again:
//   dh = relu'(h) * W2^T dl,   dz = W1^T dh,   slab row of workgroup g = sum over its patches of dz[channel(p)] * unit[b][p]
// (exact: the net is piecewise linear and channel f reaches the head only through z_a[f], z_b[f]).  ws_dh / ws_dl feed the
// fc gradients of the reduce kernel as usual.
template <class Sh>
__global__ __launch_bounds__(256) void unit_backward_kernel(const UnitBwdArgs a) {
  constexpr int F = Sh::F, F2 = Sh::F2, H = Sh::H, Cg = Sh::Cg;
  constexpr int W1S = F2 + 1;                       // column reads of fc1.weight: odd stride
  constexpr int NE = (Sh::SLAB + 255) / 256;        // slab elements per thread
  __shared__ float sW1[H * W1S];
  __shared__ float sW2[KMAX * (H + 1)];
  // NPB patches of the workgroup per round (wave q <-> patch q for the head vectors): one set of barriers for four patches
  constexpr int NPB = 4;
  static_assert(H == 64 && KMAX == 64, "wave q handles the 64 hidden units / logits of patch q");
  __shared__ float sDl[NPB][KMAX], sDh[NPB][H], sDz[NPB][F2 + 1];
  const int tid = threadIdx.x, K = a.K;
  for (int i = tid; i < H * F2; i += 256) sW1[(i / F2) * W1S + (i % F2)] = a.theta[Sh::oFc1w + i];
  for (int i = tid; i < K * H; i += 256) sW2[(i / H) * (H + 1) + (i % H)] = a.theta[Sh::oFc2w + i];
  int ix[NE];                                       // dz index of this thread's slab elements
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int p = tid + 256 * e;
    int v = F2;                                     // padding: multiplies with sDz[.][F2] = 0
    if (p < Sh::oA1b) v = p / Cg;
    else if (p < Sh::oA2w) v = p - Sh::oA1b;
    else if (p < Sh::oA2b) v = (p - Sh::oA2w) / 9;
    else if (p < Sh::oB1w) v = p - Sh::oA2b;
    else if (p < Sh::oB1b) v = F + (p - Sh::oB1w) / Sh::TB;
    else if (p < Sh::oB2w) v = F + p - Sh::oB1b;
    else if (p < Sh::oB2b) v = F + (p - Sh::oB2w) / 9;
    else if (p < Sh::NCONV) v = F + p - Sh::oB2b;
    ix[e] = v;
  }
  float acc[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) acc[e] = 0.f;
  if (tid < NPB) sDz[tid][F2] = 0.f;
  const int q = tid >> 6, j = tid & 63;             // this thread's patch slot and hidden unit / logit
  for (int b0 = blockIdx.x; b0 < a.B; b0 += NPB * gridDim.x) {
    float u[NPB][NE];                               // the unit rows of the round's patches: issued before the head math
#pragma unroll
    for (int r = 0; r < NPB; ++r) {
      const int b = b0 + r * (int)gridDim.x;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int p = tid + 256 * e;
        u[r][e] = (b < a.B && p < Sh::SLAB) ? a.unit[(size_t)b * Sh::SLAB + p] : 0.f;
      }
    }
    const int bq = b0 + q * (int)gridDim.x;
    const bool on = bq < a.B;
    const float d = (on && j < K) ? a.dlogits[(size_t)bq * K + j] : 0.f;
    sDl[q][j] = d;
    if (on) a.ws_dl[hv_index(bq, j, a.B)] = d;
    const float hj = on ? a.ws_h[hv_index(bq, j, a.B)] : 0.f;
    __syncthreads();
    {
      float s_ = 0.f;
      for (int k = 0; k < K; ++k) s_ = fmaf(sW2[k * (H + 1) + j], sDl[q][k], s_);
      const float dh = hj > 0.f ? s_ : 0.f;
      sDh[q][j] = dh;
      if (on) a.ws_dh[hv_index(bq, j, a.B)] = dh;
    }
    __syncthreads();
    for (int t = tid; t < NPB * F2; t += 256) {
      const int r = t / F2, i = t - r * F2;
      float s_ = 0.f;
#pragma unroll 8
      for (int jj = 0; jj < H; ++jj) s_ = fmaf(sW1[jj * W1S + i], sDh[r][jj], s_);
      sDz[r][i] = s_;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NPB; ++r)                   // (patches in ascending order: the summation order of a patch-by-patch loop)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[e] = fmaf(sDz[r][ix[e]], u[r][e], acc[e]);
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int p = tid + 256 * e;
    if (p < Sh::SLAB) a.slab[slab_index(blockIdx.x, p, gridDim.x)] = acc[e];
  }
}

// ---------------------------------------------------------------------------------------- launch
template <class Sh, int MODE, int INMODE, bool HF>
static hipError_t launch_v2_inst(const KArgs& a, int grid, int bytes, hipStream_t st) {
  static LdsAttrOnce once;
  hipError_t e = once.set(reinterpret_cast<const void*>(&patch_v2_kernel<Sh, MODE, INMODE, HF>), 160 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((patch_v2_kernel<Sh, MODE, INMODE, HF>), dim3(grid), dim3(V2<Sh>::NT), bytes, st, a.in.xy, a.in.cursor, a.in.B,
                     a.theta, a.pool, a.in.sceneA, a.in.sceneB, a.in.Wp, a);
  return hipGetLastError();
}

template <class Sh, bool HF>
static hipError_t launch_v2(int mode, const KArgs& a, hipStream_t st) {
  const int grid = a.in.B < MAX_BLOCKS ? a.in.B : MAX_BLOCKS;
  if (grid <= 0) return hipSuccess;
  const int bytes = mode == MODE_FWD ? V2<Sh, false, HF>::lds_bytes(a.K) : V2<Sh, true, HF>::lds_bytes(a.K);
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  const bool gather = a.in.mode == 1;
  switch (mode) {
    case MODE_FWD:
      return gather ? launch_v2_inst<Sh, MODE_FWD, 1, HF>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_FWD, 0, HF>(a, grid, bytes, st);
    case MODE_TRAIN:
      return gather ? launch_v2_inst<Sh, MODE_TRAIN, 1, HF>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_TRAIN, 0, HF>(a, grid, bytes, st);
    case MODE_BWD:
      return gather ? launch_v2_inst<Sh, MODE_BWD, 1, HF>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_BWD, 0, HF>(a, grid, bytes, st);
    case MODE_UNIT:
      return gather ? launch_v2_inst<Sh, MODE_UNIT, 1, HF>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_UNIT, 0, HF>(a, grid, bytes, st);
    default:
      return hipErrorInvalidValue;
  }
}

// Compiled instances: ONE table.  X(C, C2, P, S, F, G, H) — every row gets forward / train / backward-from-dlogits /
// unit-gradient kernels for both input modes, the unit backward kernel, and its line in the supported-shape listing.
// A row must satisfy V2<>::OK (S = 1 or 4, whole 16-byte band chunks per group, <= 12 waves) and fit 160 KiB of LDS at the
// run-time K (v2_fits): 13x13 patches of a 200-band scene do not (the window alone is 135 KB) — no instance can exist
// for them on this design; 224 bands at gmf.width 40 (G | gcd(224, 40) = 8: 5 channels per group) would need 175.6 KB here
// (window 109.6 + three slab-row copies 25.7 + transposed fc1 20.5 + staged theta 8.6 + ...), and the generic kernel owns a
// group's channels in 4-channel blocks (M % 4 == 0).
#ifndef DMF_V2_EXTRA_SHAPES
#define DMF_V2_EXTRA_SHAPES(X)
#endif
#define DMF_V2_SHAPES(X)                                                                                      \
  X(200, 1, 11, 1, 40, 10, 64) /* BASELINE configs 1-2 */                                                     \
  X(200, 1, 9, 1, 40, 10, 64)                                                                                 \
  X(200, 1, 7, 1, 40, 10, 64)                                                                                 \
  X(224, 3, 11, 1, 32, 8, 64)  /* BASELINE config 3 */                                                        \
  X(224, 3, 9, 1, 32, 8, 64)                                                                                  \
  X(8, 1, 5, 1, 40, 2, 64)     /* small test scene, equal resolution */                                       \
  X(4, 1, 16, 1, 40, 1, 64)    /* stage 2 of the two-stage path: one 4-band stream + its band mean */          \
  X(4, 1, 5, 1, 40, 1, 64)     /* the same on the small test scene */                                         \
  X(4, 1, 16, 4, 40, 1, 64)    /* the reference's own data: 4-band MS + PAN at 4x (config.yml:27,77-110) */         \
  X(8, 1, 5, 4, 40, 2, 64)     /* small test scene, aux at 4x */                                              \
  DMF_V2_EXTRA_SHAPES(X)       /* rows added at build time: build.py --shapes FILE */
// ... the rows of the attention network (gmf.attention; attention block: F = 40, E = 96), which also get the two launches of
// its train step / forward that are conv work: MODE_TOKENS and MODE_DENSE
#define DMF_V2_ATTN_SHAPES(X)                                                                                 \
  X(200, 1, 11, 1, 40, 10, 64)                                                                                \
  X(8, 1, 5, 1, 40, 2, 64)
// ... and the rows that also get the fp16-scene kernels (dmf_input.half): a subset, each instance costs compile time
#define DMF_V2_HALF_SHAPES(X)                                                                                 \
  X(200, 1, 11, 1, 40, 10, 64)                                                                                \
  X(224, 3, 11, 1, 32, 8, 64)                                                                                 \
  X(8, 1, 5, 1, 40, 2, 64)                                                                                    \
  X(4, 1, 16, 1, 40, 1, 64)

template <class Sh>
static bool v2_matches(const dmf_shape& s) {
  static_assert(V2<Sh>::OK, "shape outside the v2 kernel's geometry");
  return s.C == Sh::C && s.C2 == Sh::C2 && s.P == Sh::P && s.S == Sh::S && s.F == Sh::F && s.G == Sh::G && s.H == Sh::H;
}

template <class Sh, bool HF = false>
static bool v2_fits(const dmf_shape& s) {
  static_assert(V2<Sh, true, HF>::OK, "shape outside the v2 kernel's geometry");
  return v2_matches<Sh>(s) && V2<Sh, true, HF>::lds_bytes(s.K) <= 160 * 1024;
}

template <class Sh>
static hipError_t launch_v2_attn(int mode, const KArgs& a, hipStream_t st) {
  const int grid = a.in.B < MAX_BLOCKS ? a.in.B : MAX_BLOCKS;
  if (grid <= 0) return hipSuccess;
  const bool gather = a.in.mode == 1;
  if (mode == MODE_TOKENS) {
    const int bytes = (V2<Sh, false, false>::FIXED + 2 * 128 * 72 / 2) * 4;      // + the bf16 token staging
    if (bytes > 160 * 1024) return hipErrorInvalidValue;
    return gather ? launch_v2_inst<Sh, MODE_TOKENS, 1, false>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_TOKENS, 0, false>(a, grid, bytes, st);
  }
  const int bytes = V2<Sh, true, false>::lds_bytes(a.K);
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  return gather ? launch_v2_inst<Sh, MODE_DENSE, 1, false>(a, grid, bytes, st) : launch_v2_inst<Sh, MODE_DENSE, 0, false>(a, grid, bytes, st);
}

int patch_v2_supported(const dmf_shape& s, int mode, int half) {
  if (mode == MODE_TOKENS || mode == MODE_DENSE) {       // launches of the attention network (shape->attention is set)
    if (half || s.K < 1 || s.K > KMAX) return 0;
#define X(C, C2, P, S, F, G, H) if (v2_fits<Shape<C, C2, P, S, F, G, H>>(s)) return 1;
    DMF_V2_ATTN_SHAPES(X)
#undef X
    return 0;
  }
  if (mode != MODE_FWD && mode != MODE_TRAIN && mode != MODE_BWD && mode != MODE_UNIT) return 0;
  if (s.K < 1 || s.K > KMAX || s.attention) return 0;
  if (half) {
#define X(C, C2, P, S, F, G, H) if (v2_fits<Shape<C, C2, P, S, F, G, H>, true>(s)) return 1;
    DMF_V2_HALF_SHAPES(X)
#undef X
    return 0;
  }
#define X(C, C2, P, S, F, G, H) if (v2_fits<Shape<C, C2, P, S, F, G, H>>(s)) return 1;
  DMF_V2_SHAPES(X)
#undef X
  return 0;
}

// "C/C2/P/S/F/G" of every row, for error messages
const char* patch_v2_shape_list() {
#define STR2(x) #x
#define STR(x) STR2(x)
#define X(C, C2, P, S, F, G, H) " " STR(C) "/" STR(C2) "/" STR(P) "/" STR(S) "/" STR(F) "/" STR(G)
  return DMF_V2_SHAPES(X);
#undef X
#undef STR
#undef STR2
}

hipError_t patch_v2_dispatch(const dmf_shape& s, int mode, const KArgs& a, hipStream_t st) {
  if (mode == MODE_TOKENS || mode == MODE_DENSE) {
#define X(C, C2, P, S, F, G, H) if (v2_matches<Shape<C, C2, P, S, F, G, H>>(s)) return launch_v2_attn<Shape<C, C2, P, S, F, G, H>>(mode, a, st);
    DMF_V2_ATTN_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
  }
  if (a.in.half) {
#define X(C, C2, P, S, F, G, H) if (v2_matches<Shape<C, C2, P, S, F, G, H>>(s)) return launch_v2<Shape<C, C2, P, S, F, G, H>, true>(mode, a, st);
    DMF_V2_HALF_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
  }
#define X(C, C2, P, S, F, G, H) if (v2_matches<Shape<C, C2, P, S, F, G, H>>(s)) return launch_v2<Shape<C, C2, P, S, F, G, H>, false>(mode, a, st);
  DMF_V2_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

const char* patch_v2_half_shape_list() {
#define STR2(x) #x
#define STR(x) STR2(x)
#define X(C, C2, P, S, F, G, H) " " STR(C) "/" STR(C2) "/" STR(P) "/" STR(S) "/" STR(F) "/" STR(G)
  return DMF_V2_HALF_SHAPES(X);
#undef X
#undef STR
#undef STR2
}

hipError_t patch_v2_unit_backward(const dmf_shape& s, const UnitBwdArgs& a, hipStream_t st) {
  const int grid = a.B < MAX_BLOCKS ? a.B : MAX_BLOCKS;
  if (grid <= 0) return hipSuccess;
#define X(C, C2, P, S, F, G, H)                                                                              \
  if (v2_matches<Shape<C, C2, P, S, F, G, H>>(s)) {                                                          \
    hipLaunchKernelGGL((unit_backward_kernel<Shape<C, C2, P, S, F, G, H>>), dim3(grid), dim3(256), 0, st, a); \
    return hipGetLastError();                                                                                \
  }
  DMF_V2_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

#ifdef DMF_STAMPS
hipError_t set_v2_stamps(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_v2stamps), &p, sizeof(p)); }
#endif

}  // namespace dmf
