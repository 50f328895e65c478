// Compile-time geometry of one GMFNet instance + the flat parameter / workspace layouts shared by
// host and device code.  (The architecture itself is stated in oracle/gmfnet_ref.py and DESIGN.md §2.)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace dmf {

constexpr int MAX_BLOCKS = 256;    // one workgroup per CU; a workgroup walks patches b, b+grid, ...
constexpr int KMAX = 64;

template <int C_, int C2_, int P_, int S_, int F_, int G_, int H_>
struct Shape {
  static constexpr int C = C_, C2 = C2_, P = P_, S = S_, F = F_, G = G_, H = H_;
  static constexpr int P2 = P * P;          // pixels (tokens) per patch
  static constexpr int Cg = C / G;          // bands per spectral group
  static constexpr int M = F / G;           // outputs per spectral group
  static constexpr int SP = S * P;          // aux patch side
  static constexpr int PB = SP * SP;        // aux pixels per patch
  static constexpr int TB = C2 * S * S;     // taps of the lift conv
  static constexpr int F2 = 2 * F;
  // flat parameter offsets (floats) — order documented in include/dmf.h
  static constexpr int oA1w = 0;
  static constexpr int oA1b = oA1w + F * Cg;
  static constexpr int oA2w = oA1b + F;
  static constexpr int oA2b = oA2w + F * 9;
  static constexpr int oB1w = oA2b + F;
  static constexpr int oB1b = oB1w + F * TB;
  static constexpr int oB2w = oB1b + F;
  static constexpr int oB2b = oB2w + F * 9;
  static constexpr int NCONV = oB2b + F;    // conv parameters: reduced through per-workgroup slabs
  static constexpr int oFc1w = NCONV;
  static constexpr int oFc1b = oFc1w + H * F2;
  static constexpr int oFc2w = oFc1b + H;   // [K, H], K is a run-time value
  static constexpr int SLAB = (NCONV + 31) & ~31;   // slab row pitch (floats)

  static_assert(C % G == 0 && F % G == 0, "groups must divide C and F");
  static_assert(Cg % 4 == 0, "bands per group: whole 16-byte chunks");
  static_assert(F % 4 == 0, "feature channels in fours (the reduce launch's tiles)");
  static_assert(P <= 16, "the rows of a patch are the lanes of (at most) one 16-lane group");
  static_assert(H <= 64, "head mapping: one lane per hidden unit");
};

// run-time mirror of the layout (host side)
struct Layout {
  int C, C2, P, S, F, G, H, K, attention, E;
  int Cg, TB, F2, NCONV, SLAB;
  int64_t off[17];
  int64_t n_params;
};

inline Layout make_layout(int C, int C2, int P, int S, int F, int G, int H, int K, int attention = 0, int E = 0) {
  Layout L{};
  L.C = C; L.C2 = C2; L.P = P; L.S = S; L.F = F; L.G = G; L.H = H; L.K = K; L.attention = attention; L.E = E;
  L.Cg = C / G; L.TB = C2 * S * S; L.F2 = 2 * F;
  int64_t o = 0;
  const int64_t sizes[12] = {(int64_t)F * L.Cg, F, (int64_t)F * 9, F, (int64_t)F * L.TB, F, (int64_t)F * 9, F,
                             (int64_t)H * L.F2, H, (int64_t)K * H, K};
  for (int i = 0; i < 12; ++i) { L.off[i] = o; o += sizes[i]; }
  const int64_t asz = attention ? (int64_t)E * F : 0;          // attn_wq, attn_wk, attn_wv [E,F], attn_wo [F,E]
  for (int i = 12; i < 16; ++i) { L.off[i] = o; o += asz; }
  L.off[16] = o;
  L.NCONV = (int)L.off[8];
  L.SLAB = (L.NCONV + 31) & ~31;
  L.n_params = o;
  return L;
}

// Layouts INSIDE the workspace regions (round 3; the reduce launch is bound by what ONE CU can fetch, ~15 B/clk from the
// Infinity Cache, so every reader block must get a small CONTIGUOUS share):
//   slab rows    piece-major  [SLAB/16 pieces][rows][16]: the 16 parameters of a piece from all workgroups are contiguous
//                (rows x 64 B), one reduce block per piece reads them with full-line requests;
//   head vectors strip-major  [W/8 strips][B][8] (W = 2F, H, H, KMAX): an 8-wide strip of all patches is contiguous (B x 32 B),
//                which is exactly one operand of an 8x8 weight-gradient tile of the reduce launch.
#if defined(__HIPCC__) || defined(__HIP__)
#define DMF_HD __host__ __device__
#else
#define DMF_HD
#endif
DMF_HD inline size_t slab_index(int row, int p, int rows) { return ((size_t)(p >> 4) * rows + row) * 16 + (p & 15); }
DMF_HD inline size_t hv_index(int b, int i, int B) { return ((size_t)(i >> 3) * B + b) * 8 + (i & 7); }

// workspace layout (floats): [slab MAX_BLOCKS x SLAB][z B x 2F][h B x H][dh B x H][dl B x KMAX]
//                            [attention: aslab MAX_BLOCKS x 4EF — per-workgroup gradients of Wq, Wk, Wv, Wo]
//                            [unit B x SLAB — per-PATCH unit conv gradients of the two-launch step (dmf_forward_unit)]
struct WsLayout { int64_t slab, z, h, dh, dl, aslab, unit, total; };
inline WsLayout make_ws(const Layout& L, int B) {
  WsLayout w{};
  int64_t o = 0;
  w.slab = o; o += (int64_t)MAX_BLOCKS * L.SLAB;
  w.z = o;    o += (int64_t)B * L.F2;
  w.h = o;    o += (int64_t)B * L.H;
  w.dh = o;   o += (int64_t)B * L.H;
  w.dl = o;   o += (int64_t)B * KMAX;
  w.aslab = o; o += L.attention ? (int64_t)MAX_BLOCKS * 4 * L.E * L.F : 0;
  w.unit = o;  o += L.attention ? 0 : (int64_t)B * L.SLAB;
  w.total = o;
  return w;
}

}  // namespace dmf
