// caro_variants.h -- the lane geometries of the tree kernels and the run-time choice among them: one table for the
// kernels (caro_engine.hip), the host-side single-state helpers (caro_host.inc) and the CPU sanitizer build of those
// helpers (oracle/asan/host_tu.cpp, plain g++).  Needs caro_rules.h and include/caro_hip.h.
#ifndef CARO_VARIANTS_H
#define CARO_VARIANTS_H

namespace caro {

template <class R_, int LPD_, int APL_>
struct Geo {
  using R = R_;
  static constexpr int LPD = LPD_, APL = APL_, AP = LPD_ * APL_, KW = R_::KW;
};
using GeoC4 = Geo<C4Rules, 8, 1>;
using GeoM16 = Geo<MnkRules<1>, 16, 1>;
using GeoM32 = Geo<MnkRules<1>, 32, 1>;
using GeoM64 = Geo<MnkRules<1>, 64, 1>;
using GeoM128 = Geo<MnkRules<2>, 64, 2>;
using GeoM256 = Geo<MnkRules<4>, 64, 4>;

enum Variant { V_C4, V_M16, V_M32, V_M64, V_M128, V_M256, V_BAD };

static inline Variant pick_variant(int kind, int n) {
  if (kind == CARO_GAME_CONNECT4) return V_C4;
  if (kind != CARO_GAME_MNK || n < 2 || n > 15) return V_BAD;
  const int A = n * n;
  if (A <= 16) return V_M16;
  if (A <= 32) return V_M32;
  if (A <= 64) return V_M64;
  if (A <= 128) return V_M128;
  return V_M256;
}
static inline int variant_kw(Variant v) {
  switch (v) {
    case V_C4: return 1;
    case V_M16: case V_M32: case V_M64: return 2;
    case V_M128: return 4;
    case V_M256: return 8;
    default: return 0;
  }
}
static inline int variant_lpd(Variant v) {
  switch (v) {
    case V_C4: return 8;
    case V_M16: return 16;
    case V_M32: return 32;
    default: return 64;
  }
}
static inline int variant_ap(Variant v) {
  switch (v) {
    case V_C4: return 8;
    case V_M16: return 16;
    case V_M32: return 32;
    case V_M64: return 64;
    case V_M128: return 128;
    case V_M256: return 256;
    default: return 0;
  }
}
static inline GameParams make_gp(int kind, int n, int k) {
  GameParams gp;
  gp.kind = kind;
  if (kind == CARO_GAME_CONNECT4) {
    gp.n = 0; gp.k = 4; gp.A = 7; gp.rows = 6; gp.cols = 7;
  } else {
    gp.n = n; gp.k = k; gp.A = n * n; gp.rows = n; gp.cols = n;
  }
  return gp;
}

#define DISPATCH(var, EXPR)                                          \
  switch (var) {                                                     \
    case V_C4: { using GEO = GeoC4; EXPR; } break;                   \
    case V_M16: { using GEO = GeoM16; EXPR; } break;                 \
    case V_M32: { using GEO = GeoM32; EXPR; } break;                 \
    case V_M64: { using GEO = GeoM64; EXPR; } break;                 \
    case V_M128: { using GEO = GeoM128; EXPR; } break;               \
    case V_M256: { using GEO = GeoM256; EXPR; } break;               \
    default: return fail(CARO_E_INVAL, "unsupported game geometry"); \
  }

}  // namespace caro

#endif
