// select_rows.h — select_rows_kernel: the selection of rows a7-a10 with FOUR scan lines per wavefront, one 16-lane DPP row
// per line (round 5). Included by extract_kernels.hip inside its anonymous namespace (it uses the helpers defined there).
//
// Reference: loam/include/loam/features-inl.h:27-48 (sector loop, std::sort), :137-180 (the two greedy walks: max+1 cap,
// strict-< thresholds, suppression of +-(np-1) around a pick, which crosses sector boundaries).
//
// Why: select_mis_kernel gives a whole wavefront to one scan line and walks its sectors one after the other (they are
// coupled through the validity mask: SURVEY Q4), so a sector-pass keeps 170 / 16 = 10.6 of the 64 lanes busy. Scan lines
// ARE independent. Here a row of 16 lanes owns a line, its lanes own CH = ceil((sector + R) / 16) consecutive points of
// the CURRENT sector (11 for 64 x 1024 / 6 sectors), re-dealt per sector, and four lines advance through their sectors
// in lock step. Every per-sector step — the MIS rounds, the pick compaction, the sort, the cap, the suppression — is a
// row-local operation (DPP row shifts / mirrors / broadcasts, 32-bit masks), so one instruction works for four lines.
//
//   * Curvature of the current sector (+ R points before, 2R after) is staged per row in LDS; line ends and positions
//     outside the line are stored as NaN (never candidates, never equal to anything).
//   * The validity bytes become one bit per point in LDS once per line; a lane reads its CH bits with one unaligned
//     two-word read and returns the bits it cleared with LDS atomics, so suppression that reaches into the next sector
//     needs no carry logic: the next sector simply reads the line's current bits.
//   * Sort: picks get 32-bit keys — (hi word of the curvature - a base, clamped to 32 - IB bits) << IB | sector position,
//     complemented for the descending edge walk — and are sorted by a bitonic network over 16 lanes x 4 registers
//     (rank = 4 * lane + register) whose cross-lane steps are v_min_u32 / v_max_u32 WITH a DPP source operand. The key
//     keeps 20 mantissa bits over 16 octaves below (planar) / above (edge) the threshold; when two picks of a sector
//     share that part the pass is redone with 64-bit keys (the scheme of select_mis_kernel), and when those cannot tell
//     two picks apart either the line is marked for replay_kernel (std::sort's order decides: row a7).
//   * Fused compaction as in select_mis_kernel (chained scan over the published line totals); the picks' columns are
//     kept in LDS until the copy.
#pragma once

// Geometry of one launch, computed by the host (row_select_geom) and passed by value.
constexpr int kKeyShift = 2;  // see row_select_geom
struct RowSelGeom {
  uint32_t ch;         // points of the current sector per lane
  uint32_t ib;         // bits of a sector position inside the sort keys (16 * ch <= 1 << ib)
  uint32_t pitch;      // doubles per row of the curvature buffer (odd: rows fall on different banks)
  uint32_t vw;         // dwords per row of the validity bits
  uint32_t cap_e, cap_p;  // picks kept per sector at most (<= 64)
  uint32_t pk_stride;  // entries per row of the pick lists: S * (cap_e + cap_p)
  uint32_t pk8;        // 1: the lists hold bytes (a sector position fits), 0: 16-bit words
  uint32_t off_vb, off_sl, off_tk, off_pk, off_cnt;  // byte offsets inside a wavefront's LDS block
  uint32_t off_xa, off_sr, off_vs;  // fused form: one line's staged points of a sector, their ranges, the sector's validity words
  uint32_t nx;         // fused form: points staged per line and sector
  uint32_t bytes;      // LDS bytes per wavefront (multiple of 16)
  int32_t kbase_e, kbase_p;  // hi-word bases of the 32-bit keys
};

__host__ inline bool row_select_geom(const ExtractParams& P, RowSelGeom& G, bool fused = false) {
  const int R = (int)P.np - 1;
  if (R < 1 || R > 4 || P.H % 4 != 0 || P.S == 0 || P.S > 16 || P.W % 16 != 0 || P.pps == 0) return false;
  const uint32_t longest = P.W - (P.S - 1) * P.pps;
  const uint32_t ch = (longest + (uint32_t)R + 15) / 16;
  if ((int)ch < R || ch + 2 * (uint32_t)R > 32) return false;
  if ((longest + (uint32_t)R) / (uint32_t)(R + 1) > 64) return false;  // picks of a sector are >= R + 1 points apart
  G = RowSelGeom{};
  G.ch = ch;
  G.ib = 1;
  while ((1u << G.ib) < 16 * ch) G.ib++;
  G.pitch = (16 * ch + 3 * (uint32_t)R) | 1u;
  G.vw = (P.W + 31) / 32 + 3;
  G.cap_e = P.cap_edge < 64 ? P.cap_edge : 64, G.cap_p = P.cap_planar < 64 ? P.cap_planar : 64;
  G.pk_stride = P.S * (G.cap_e + G.cap_p);
  auto up16 = [](uint32_t x) { return (x + 15u) & ~15u; };
  uint32_t o = up16(4 * G.pitch * 8);
  G.off_vb = o, o = up16(o + 4 * G.vw * 4);
  G.off_sl = o, o = up16(o + 4 * 64 * 4);
  G.off_tk = o, o = up16(o + 4 * 8);
  G.pk8 = G.ib <= 8 ? 1u : 0u;
  G.off_pk = o, o = up16(o + 4 * G.pk_stride * (G.pk8 ? 1 : 2));
  G.off_cnt = o, o = up16(o + 4 * 32 * 2);
  if (fused) {  // + one line's points of a sector with the halo of the curvature sum and the validity windows
    G.nx = 16 * ch + 3 * (uint32_t)R + 2 * P.np;
    G.off_xa = o, o = up16(o + G.nx * 24);
    G.off_sr = o, o = up16(o + G.nx * 8);
    G.off_vs = o, o = up16(o + 4 * 32);
  }
  G.bytes = o;
  if (G.bytes * (fused ? 2 : 4) > 64 * 1024) return false;
  if (24 * longest + 64 * 24 > G.off_sl) return false;  // the copy phase's buffers (a sector's points + 64 picked points) reuse the curvature buffer
  // keys: kb = 32 - ib bits of (hi word - base). Planar candidates lie below their threshold: the top of the range is
  // the threshold's hi word; edge candidates lie above theirs: the bottom of the range is the threshold's hi word.
  auto hi_word = [](double v) {
    union {
      double d;
      uint64_t b;
    } u;
    u.d = v;
    return (int32_t)(u.b >> 32);
  };
  // The curvature part of a key is (hi word - base) >> kKeyShift, clamped: 32 - ib bits that hold 20 - kKeyShift mantissa bits
  // over 2^(12 - ib + kKeyShift) octaves (ib = 8, shift 2: 18 mantissa bits over 64 octaves). Round 5: with shift 0 — 16
  // octaves — every planar pick below 2^-16 of its threshold was clamped to the same key, and a sector of a flat surface has
  // several of those: the exact 64-bit pass ran in nearly every planar pass (counted: 195 000 of 196 608 wavefront-sectors).
  const int32_t kmax = (int32_t)((1u << (32 - G.ib)) - 1u);
  G.kbase_p = (P.planar_thr > 0.0 ? hi_word(P.planar_thr) : 0) - (int32_t)((uint32_t)kmax << kKeyShift);
  G.kbase_e = P.edge_thr > 0.0 ? hi_word(P.edge_thr) : 0;
  return true;
}

// ---- row-local lane moves (DPP inside a row of 16 lanes; lanes without a source read 0) -------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t row_move(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t row_prev(uint32_t v) { return row_move<0x111>(v); }  // row_shr:1: value of lane - 1
__device__ __forceinline__ uint32_t row_next(uint32_t v) { return row_move<0x101>(v); }  // row_shl:1: value of lane + 1
template <int N>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) { return row_move<0x150 + N>(v); }  // row_newbcast:N
__device__ __forceinline__ uint32_t row_incl_scan_u32(uint32_t x) {
  x += row_move<0x111>(x), x += row_move<0x112>(x), x += row_move<0x114>(x), x += row_move<0x118>(x);
  return x;
}
// does any lane of my row hold a true p? (rare paths only: a ballot and a 64-bit shift)
__device__ __forceinline__ bool row_any(bool p, int lane) { return ((__ballot(p) >> (lane & 48)) & 0xFFFFull) != 0ull; }

// ---- 32-bit forms of the lane-mask MIS (extract_math.h: mis_window / mis_winners / mis_spread) -------------------------
template <int R>
__device__ __forceinline__ uint32_t win32(uint32_t own, uint32_t prev, uint32_t next, int CH) {
  constexpr uint32_t mR = (1u << R) - 1u;
  return ((prev >> (CH - R)) & mR) | (own << R) | ((next & mR) << (R + CH));
}
template <int R, bool EDGE>
__device__ __forceinline__ uint32_t winners32(uint32_t Uw, const uint32_t gt[R]) {
  uint32_t blocked = 0;
#pragma unroll
  for (int d = 1; d <= R; d++) {
    const uint32_t g = gt[d - 1], gs = g << d;
    const uint32_t beats_up = EDGE ? g : ~g, beats_dn = EDGE ? ~gs : gs;
    blocked |= (Uw >> d) & ~beats_up;
    blocked |= (Uw << d) & ~beats_dn;
  }
  return Uw & ~blocked;
}
template <int R>
__device__ __forceinline__ uint32_t spread32(uint32_t Ww) {
  uint32_t r = Ww;
#pragma unroll
  for (int d = 1; d <= R; d++) r |= (Ww << d) | (Ww >> d);
  return r;
}

// m = 2 m + (a > b) / (a < b): the mask is built from its top bit down, one compare + one add-with-carry per bit
__device__ __forceinline__ void push_gt(uint32_t& m, double a, double b) {
  asm("v_cmp_gt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void push_lt(uint32_t& m, double a, double b) {
  asm("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}

// the same on the hi words of non-negative doubles (the split form: select_rows_kernel<.., SPLIT>)
__device__ __forceinline__ void push_gt_i32(uint32_t& m, int32_t a, int32_t b) {
  asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void push_lt_i32(uint32_t& m, int32_t a, int32_t b) {
  asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void push_ge0_i32(uint32_t& m, int32_t a) {
  asm("v_cmp_le_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a) : "vcc");
}

// ---- the sort: 64 keys of a row on 16 lanes x 4 registers, rank q = 4 * lane + register, ascending ---------------------
// Bitonic network in its "flip" form (the first step of a merge of size k compares q with q ^ (k - 1), the others q with
// q ^ j: the lower index always keeps the minimum). Register bits are the low bits of q, so the most frequent steps
// (j = 1, 2) are plain v_min / v_max between registers; lane steps use a DPP source: two instructions where the
// min / max side is a whole bank of four lanes, three (min, max, select) inside a quad. (tools/probes/valu_issue.hip
// checks every DPP form used here on the device; /tmp simulation of the network: 20 000 random sets.)
#define LOAMX_CX(a, b)                                  \
  {                                                     \
    const uint32_t lo_ = a < b ? a : b, hi_ = a < b ? b : a; \
    a = lo_, b = hi_;                                   \
  }
// d = lanes of the `lo` banks: min(own, dpp(src)); lanes of the `hi` banks: max(own, dpp(src))
#define LOAMX_CX_BANK(d, src, own, ctrl_lo, ctrl_hi, bank_lo, bank_hi)                               \
  asm("v_min_u32_dpp %0, %1, %2 " ctrl_lo " row_mask:0xf bank_mask:" bank_lo "\n\t"                  \
      "v_max_u32_dpp %0, %1, %2 " ctrl_hi " row_mask:0xf bank_mask:" bank_hi                         \
      : "=&v"(d)                                                                                     \
      : "v"(src), "v"(own))
// d = lanes whose bit in hi_mask is set: max(own, dpp(src)), the others: min(own, dpp(src))
#define LOAMX_CX_SEL(d, src, own, ctrl, hi_mask)                                                      \
  {                                                                                                  \
    uint32_t mn_, mx_;                                                                               \
    asm("v_min_u32_dpp %0, %2, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_max_u32_dpp %1, %2, %3 " ctrl " row_mask:0xf bank_mask:0xf"                               \
        : "=&v"(mn_), "=&v"(mx_)                                                                     \
        : "v"(src), "v"(own));                                                                       \
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mn_), "v"(mx_), "s"(hi_mask));                 \
  }

__device__ __forceinline__ void row_sort64_u32(uint32_t& k0, uint32_t& k1, uint32_t& k2, uint32_t& k3, uint32_t n2) {
  constexpr unsigned long long kOdd = 0xAAAAAAAAAAAAAAAAull, kBit1 = 0xCCCCCCCCCCCCCCCCull;  // lanes with bit 0 / bit 1 set
  uint32_t a, b, c, d;
  // k = 2
  LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
  if (n2 <= 2) return;
  // k = 4: q ^ 3, then j = 1
  LOAMX_CX(k0, k3) LOAMX_CX(k1, k2) LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
  if (n2 <= 4) return;
  // k = 8: q ^ 7 = (lane ^ 1, 3 - register), then j = 2, 1
  LOAMX_CX_SEL(a, k3, k0, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(b, k2, k1, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(c, k1, k2, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(d, k0, k3, "quad_perm:[1,0,3,2]", kOdd)
  k0 = a, k1 = b, k2 = c, k3 = d;
  LOAMX_CX(k0, k2) LOAMX_CX(k1, k3) LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
  if (n2 <= 8) return;
  // k = 16: q ^ 15 = (lane ^ 3, 3 - register), then j = 4 (lane ^ 1), 2, 1
  LOAMX_CX_SEL(a, k3, k0, "quad_perm:[3,2,1,0]", kBit1)
  LOAMX_CX_SEL(b, k2, k1, "quad_perm:[3,2,1,0]", kBit1)
  LOAMX_CX_SEL(c, k1, k2, "quad_perm:[3,2,1,0]", kBit1)
  LOAMX_CX_SEL(d, k0, k3, "quad_perm:[3,2,1,0]", kBit1)
  LOAMX_CX_SEL(k0, a, a, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k1, b, b, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k2, c, c, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k3, d, d, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX(k0, k2) LOAMX_CX(k1, k3) LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
  if (n2 <= 16) return;
  // k = 32: q ^ 31 = (lane ^ 7, 3 - register): row_half_mirror, lanes 0-3 / 8-11 of a row (banks 0, 2) keep the minimum;
  // then j = 8 (lane ^ 2), 4 (lane ^ 1), 2, 1
  LOAMX_CX_BANK(a, k3, k0, "row_half_mirror", "row_half_mirror", "0x5", "0xa");
  LOAMX_CX_BANK(b, k2, k1, "row_half_mirror", "row_half_mirror", "0x5", "0xa");
  LOAMX_CX_BANK(c, k1, k2, "row_half_mirror", "row_half_mirror", "0x5", "0xa");
  LOAMX_CX_BANK(d, k0, k3, "row_half_mirror", "row_half_mirror", "0x5", "0xa");
  LOAMX_CX_SEL(k0, a, a, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(k1, b, b, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(k2, c, c, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(k3, d, d, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(a, k0, k0, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(b, k1, k1, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(c, k2, k2, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(d, k3, k3, "quad_perm:[1,0,3,2]", kOdd)
  k0 = a, k1 = b, k2 = c, k3 = d;
  LOAMX_CX(k0, k2) LOAMX_CX(k1, k3) LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
  if (n2 <= 32) return;
  // k = 64: q ^ 63 = (lane ^ 15, 3 - register): row_mirror, lanes 0-7 (banks 0, 1) keep the minimum; then j = 16 (lane ^ 4:
  // lane + 4 for banks 0, 2, lane - 4 for banks 1, 3), 8, 4, 2, 1
  LOAMX_CX_BANK(a, k3, k0, "row_mirror", "row_mirror", "0x3", "0xc");
  LOAMX_CX_BANK(b, k2, k1, "row_mirror", "row_mirror", "0x3", "0xc");
  LOAMX_CX_BANK(c, k1, k2, "row_mirror", "row_mirror", "0x3", "0xc");
  LOAMX_CX_BANK(d, k0, k3, "row_mirror", "row_mirror", "0x3", "0xc");
  LOAMX_CX_BANK(k0, a, a, "row_shl:4", "row_shr:4", "0x5", "0xa");
  LOAMX_CX_BANK(k1, b, b, "row_shl:4", "row_shr:4", "0x5", "0xa");
  LOAMX_CX_BANK(k2, c, c, "row_shl:4", "row_shr:4", "0x5", "0xa");
  LOAMX_CX_BANK(k3, d, d, "row_shl:4", "row_shr:4", "0x5", "0xa");
  LOAMX_CX_SEL(a, k0, k0, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(b, k1, k1, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(c, k2, k2, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(d, k3, k3, "quad_perm:[2,3,0,1]", kBit1)
  LOAMX_CX_SEL(k0, a, a, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k1, b, b, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k2, c, c, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX_SEL(k3, d, d, "quad_perm:[1,0,3,2]", kOdd)
  LOAMX_CX(k0, k2) LOAMX_CX(k1, k3) LOAMX_CX(k0, k1) LOAMX_CX(k2, k3)
}

// The same 64 ranks on doubles (the exact pass): the textbook bitonic network, directions by rank, exchanges by the
// row-local xor moves of xor_lane_f64 (J <= 8 stays inside a row). Slow and rare.
__device__ __forceinline__ void row_sort64_f64(double key[4], uint32_t n2, int l) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
    if ((uint32_t)k > n2) break;  // wave-uniform
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < 4) {  // partner in another register of the same lane
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int o = r ^ j;
          if (o > r) {
            const int q = 4 * l + r;
            const bool up = (q & k) == 0;  // (k <= 2: a register bit, known at compile time after unrolling)
            const double mn = fmin(key[r], key[o]), mx = fmax(key[r], key[o]);
            key[r] = up ? mn : mx, key[o] = up ? mx : mn;
          }
        }
      } else {  // partner in lane ^ (j / 4), same register
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int q = 4 * l + r;
          const bool keep_min = ((q & j) == 0) == ((q & k) == 0);
          const double other = xor_lane_f64(key[r], j >> 2, l);
          const double mn = fmin(key[r], other), mx = fmax(key[r], other);
          key[r] = keep_min ? mn : mx;
        }
      }
    }
  }
}

struct RowSelCtx {  // per-lane constants of a wavefront's four lines
  int lane, row, l;
  int CH, W;
  uint32_t cm;          // the CH own bits of a lane
  const double* cp;     // LDS: curvature at sector position l * CH - R (window value 0)
  const double* crow;   // LDS: curvature at sector position -R of my row
  uint32_t* sl;         // LDS: the row's 64 key slots (4 bytes each)
  uint32_t* tk;         // LDS: the row's cap threshold key (8 bytes)
  uint32_t ib, pk8;     // bits of a sector position; pick lists hold bytes (1) or 16-bit words (0)
  bool no_stage;        // the stage arrays are not written (ExtractFused::no_stage)
  int32_t kbase_e, kbase_p, kmax;
};

// One pass (edge or planar) over the current sector of the four lines. V: validity of the lane's CH points (bit j =
// sector position l * CH + j), T: candidates by threshold, sm: positions inside the sector. Returns the picks kept by the
// lane's row; tie: row-uniform, set when std::sort's order on equal curvatures could decide something.
// hiw: (compile-time CH) the hi words of the lane's own CH curvatures, in registers: the keys of all own points are then
// built up front by straight code instead of one dependent LDS read per pick. pk_list: the row's pick list of this
// (sector, kind): sector positions of the kept picks in output order.
// ensure(): (wave-uniform call) makes the curvature buffer hold doubles before the exact pass reads it (the split form
// stages hi words only; a no-op otherwise).
template <int R, bool EDGE, int CHT, typename Ensure>
__device__ __forceinline__ uint32_t row_pass(const RowSelCtx& X, uint32_t& V, uint32_t T, uint32_t sm, const uint32_t gt[R],
                                             const uint32_t eqm[R], bool have_eq, bool& tie, uint32_t cap, int start,
                                             uint32_t line_base, uint32_t* __restrict__ stage, void* pk_list, const int32_t* hiw,
                                             Ensure&& ensure) {
  const int l = X.l, CH = X.CH;
  uint32_t U = V & T & sm;
  if (__ballot(U != 0) == 0) return 0;
  if (have_eq) {  // (uniform, rare) equal curvatures among candidates within R points of each other
    const uint32_t Uw = win32<R>(U, row_prev(U), row_next(U), CH);
    uint32_t adj = 0;
#pragma unroll
    for (int d = 1; d <= R; d++) adj |= Uw & (Uw >> d) & eqm[d - 1];
    if (row_any(adj != 0, X.lane)) tie = true;
  }
  uint32_t Pk = 0;
  do {  // rounds of "local maxima win, their neighbours leave"
    const uint32_t Uw = win32<R>(U, row_prev(U), row_next(U), CH);
    const uint32_t win = (winners32<R, EDGE>(Uw, gt) >> R) & X.cm;
    const uint32_t Ww = win32<R>(win, row_prev(win), row_next(win), CH);
    Pk |= win;
    U &= ~((spread32<R>(Ww) >> R) & X.cm);
  } while (__ballot(U != 0) != 0);
  const uint32_t cnt = (uint32_t)__popc(Pk);
  const uint32_t incl = row_incl_scan_u32(cnt);
  const uint32_t total = row_bcast<15>(incl);
  // the largest pick count of the four rows sizes the network (uniform)
  uint32_t tmax = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
  {
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31), t2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47),
                   t3 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    tmax = tmax > t1 ? tmax : t1, tmax = tmax > t2 ? tmax : t2, tmax = tmax > t3 ? tmax : t3;
  }
  const uint32_t n2 = tmax <= 2 ? 2u : (1u << (32 - __clz((int)tmax - 1)));
  const uint32_t kept = total < cap ? total : cap;  // features-inl.h:155 / :177: at most max + 1 picks
  const uint32_t pmask = (1u << X.ib) - 1u;
  const int32_t kbase = EDGE ? X.kbase_e : X.kbase_p;
  // 32-bit key of the own point j from the hi word of its curvature
  auto make_key = [&](int32_t hi, int j) -> uint32_t {
    int32_t kx = (hi - kbase) >> kKeyShift;
    kx = kx < 0 ? 0 : (kx > X.kmax ? X.kmax : kx);
    const uint32_t k = ((uint32_t)kx << X.ib) | (uint32_t)(l * CH + j);
    return EDGE ? ~k : k;
  };
  auto key32 = [&](int j) -> uint32_t { return make_key(reinterpret_cast<const int32_t*>(X.cp)[2 * (j + R) + 1], j); };
  auto put_pick = [&](uint32_t q, uint32_t pos) {  // kept pick of output rank q at sector position pos
    if (!X.no_stage) stage[q] = line_base + (uint32_t)start + pos;
    if (X.pk8) static_cast<uint8_t*>(pk_list)[q] = (uint8_t)pos;
    else static_cast<uint16_t*>(pk_list)[q] = (uint16_t)pos;
  };
  uint32_t K = Pk;
  bool undecided;
  {
    uint32_t slot = incl - cnt;
    if constexpr (CHT != 0) {
#pragma unroll
      for (int j = 0; j < CHT; j++)
        if ((Pk >> j) & 1u) X.sl[slot++] = make_key(hiw[j], j);
    } else {
      for (uint32_t bits = Pk; bits; bits &= bits - 1) X.sl[slot++] = key32(__ffs((int)bits) - 1);
    }
    wave_lds_sync();
    const uint4 kv = *reinterpret_cast<const uint4*>(X.sl + 4 * l);
    wave_lds_sync();
    const uint32_t q0 = 4u * (uint32_t)l;
    uint32_t k0 = q0 < total ? kv.x : 0xFFFFFFFFu, k1 = q0 + 1 < total ? kv.y : 0xFFFFFFFFu;
    uint32_t k2 = q0 + 2 < total ? kv.z : 0xFFFFFFFFu, k3 = q0 + 3 < total ? kv.w : 0xFFFFFFFFu;
    row_sort64_u32(k0, k1, k2, k3, n2);
    // two picks whose keys share the curvature part: the keys cannot order them (rank q + 1 sits in the next register, or in
    // register 0 of the next lane)
    const uint32_t nx = row_next(k0);
    const bool coll = (q0 + 1 < total && (k0 >> X.ib) == (k1 >> X.ib)) || (q0 + 2 < total && (k1 >> X.ib) == (k2 >> X.ib)) ||
                      (q0 + 3 < total && (k2 >> X.ib) == (k3 >> X.ib)) || (q0 + 4 < total && (k3 >> X.ib) == (nx >> X.ib));
    undecided = __ballot(coll) != 0;
    if (!undecided) {
      const uint32_t kk[4] = {k0, k1, k2, k3};
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (q0 + (uint32_t)r < kept) put_pick(q0 + r, (EDGE ? ~kk[r] : kk[r]) & pmask);
      if (__ballot(kept < total) != 0) {  // the cap binds in some row: its picks after the last kept one do not suppress
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (q0 + (uint32_t)r + 1 == kept) X.tk[0] = kk[r];
        wave_lds_sync();
        if (kept < total) {
          const uint32_t tk = X.tk[0];
          if constexpr (CHT != 0) {
            uint32_t le = 0;
#pragma unroll
            for (int j = 0; j < CHT; j++) le |= (uint32_t)(make_key(hiw[j], j) <= tk) << j;
            K = Pk & le;
          } else {
            K = 0;
            for (uint32_t bits = Pk; bits; bits &= bits - 1) {
              const int j = __ffs((int)bits) - 1;
              if (key32(j) <= tk) K |= 1u << j;
            }
          }
        }
        wave_lds_sync();
      }
    }
  }
  if (undecided) {
    ensure();
    // ---- the exact pass: curvature and position folded into one double (low IB mantissa bits replaced by the position;
    // edge keys negated, padding = +inf), as select_mis_kernel sorts. Picks whose truncated curvatures still collide, or a
    // pick that is not finite: the order among them is std::sort's — the line is replayed (tie).
    auto key64 = [&](uint32_t pos, bool& finite) -> double {  // pos: sector position
      const double c = X.crow[pos + R];
      finite = c <= 1.7976931348623157e308;
      const double k = __hiloint2double(__double2hiint(c), (int)(((uint32_t)__double2loint(c) & ~pmask) | pos));
      return EDGE ? -k : k;
    };
    {
      uint32_t slot = incl - cnt;
      for (uint32_t bits = Pk; bits; bits &= bits - 1) X.sl[slot++] = (uint32_t)(l * CH + __ffs((int)bits) - 1);
    }
    wave_lds_sync();
    const uint32_t q0 = 4u * (uint32_t)l;
    const uint4 pv = *reinterpret_cast<const uint4*>(X.sl + 4 * l);
    const uint32_t pp[4] = {pv.x, pv.y, pv.z, pv.w};
    double key[4];
    bool all_finite = true;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      bool fin = true;
      key[r] = q0 + (uint32_t)r < total ? key64(pp[r], fin) : __builtin_huge_val();
      all_finite = all_finite && fin;
    }
    wave_lds_sync();
    row_sort64_f64(key, n2, l);
    auto same = [&](double a, double b) {
      return __double2hiint(a) == __double2hiint(b) && (((uint32_t)__double2loint(a) ^ (uint32_t)__double2loint(b)) & ~pmask) == 0u;
    };
    const double nx = __hiloint2double((int)row_next((uint32_t)__double2hiint(key[0])), (int)row_next((uint32_t)__double2loint(key[0])));
    const bool coll = !all_finite || (q0 + 1 < total && same(key[0], key[1])) || (q0 + 2 < total && same(key[1], key[2])) ||
                      (q0 + 3 < total && same(key[2], key[3])) || (q0 + 4 < total && same(key[3], nx));
    if (row_any(coll, X.lane)) tie = true;  // (the values written below are then unused: replay_kernel rewrites the line)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (q0 + (uint32_t)r < kept) put_pick(q0 + r, (uint32_t)__double2loint(key[r]) & pmask);
    if (__ballot(kept < total) != 0) {
      double* tk64 = reinterpret_cast<double*>(X.tk);
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (q0 + (uint32_t)r + 1 == kept) tk64[0] = key[r];
      wave_lds_sync();
      if (kept < total) {
        const double tk = tk64[0];
        K = 0;
        for (uint32_t bits = Pk; bits; bits &= bits - 1) {
          const int j = __ffs((int)bits) - 1;
          bool fin;
          if (key64((uint32_t)(l * CH + j), fin) <= tk) K |= 1u << j;
        }
      }
      wave_lds_sync();
    }
  }
  // suppression of +-(np-1) around every kept pick (features-inl.h:148-151 / :170-173)
  const uint32_t Kw = win32<R>(K, row_prev(K), row_next(K), CH);
  V &= ~((spread32<R>(Kw) >> R) & X.cm);
  return kept;
}

// CHT: points per lane as a compile-time constant (11 for 64 x 1024 / 6 sectors: the loops over a lane's points unroll),
// 0 = G.ch.
// SPLIT: the curvature arrives as hi words | lo words with the validity in the hi word's sign bit and no validity bytes
// (curvature_valid2_kernel<.., SPLIT>). A sector is staged as 32-bit hi words and every comparison is a comparison of hi
// words — exact for non-negative doubles whose hi words differ; where two hi words that are compared (neighbours within R, or
// a point and a threshold) are EQUAL, or two picks' keys collide, the sector's lo words are fetched and the doubles compared
// as before (wave-uniform, ~1 sector in 500 on noisy scans). vb then holds the SUPPRESSED points, as in the fused form.
template <int R, int CHT, bool FUSED = false, bool SPLIT = false>
__device__ __forceinline__ void select_rows_body(const double* __restrict__ curv, const uint8_t* __restrict__ mask, size_t n_lines,
                                                 const ExtractParams& P, const ExtractStage& st, const ExtractFused& fz, const RowSelGeom& G) {
  static_assert(!SPLIT || (CHT != 0 && !FUSED), "the split form is written for the compile-time lane chunk");
  if (fz.only_if && __hip_atomic_load(fz.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // uniform (ExtractFused::only_if)
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15;
  const size_t line0 = ((size_t)blockIdx.x * (FUSED ? 2 : 4) + wave) * 4;  // four consecutive lines of one scan (H % 4 == 0)
  if (line0 >= n_lines) return;  // whole wavefront leaves; no workgroup barrier below
  const int W = (int)P.W, CH = CHT ? CHT : (int)G.ch, np = (int)P.np;
  const int BLu = 16 * CH + 3 * R;  // staged positions of a sector per row: -R .. 16 CH + 2R - 1
  unsigned char* blk = smem + (size_t)wave * G.bytes;
  double* cbuf = reinterpret_cast<double*>(blk);
  uint32_t* vb = reinterpret_cast<uint32_t*>(blk + G.off_vb);
  RowSelCtx X;
  X.lane = lane, X.row = row, X.l = l, X.CH = CH, X.W = W;
  X.cm = (1u << CH) - 1u;
  X.cp = cbuf + (size_t)row * G.pitch + l * CH;
  X.crow = cbuf + (size_t)row * G.pitch;
  X.sl = reinterpret_cast<uint32_t*>(blk + G.off_sl) + row * 64;
  X.tk = reinterpret_cast<uint32_t*>(blk + G.off_tk) + row * 2;
  unsigned char* pk_row = blk + G.off_pk + ((size_t)row * G.pk_stride << (G.pk8 ? 0 : 1));
  X.no_stage = fz.no_stage != 0u;
  X.ib = G.ib, X.pk8 = G.pk8, X.kbase_e = G.kbase_e, X.kbase_p = G.kbase_p, X.kmax = (int32_t)((1u << (32 - G.ib)) - 1u);
  uint16_t* cnt = reinterpret_cast<uint16_t*>(blk + G.off_cnt) + row * 32;  // picks kept per (sector, kind) of my line
  const size_t line = line0 + row;
#ifdef LOAMX_ROWS_PROFILE
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
  const unsigned long long pt_begin = pt0;
#define ROWS_STAMP(i)                                         \
  {                                                           \
    __builtin_amdgcn_s_waitcnt(0);                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    pt[i] += t_ - pt0, pt0 = t_;                              \
  }
#else
#define ROWS_STAMP(i)
#endif

  const double kNaN = __longlong_as_double(0x7FF8000000000000ll);
  // curvature at staged position lane + 64 i of (sector s, row rr); line ends and positions outside the line are never
  // valid: NaN keeps them out of every comparison
  auto fetch_curv = [&](uint32_t s, int rr, int i) -> double {
    const int col = (int)(s * P.pps) + lane + 64 * i - R;
    return (lane + 64 * i < BLu && col >= np && col + np < W) ? curv[(line0 + rr) * (size_t)W + col] : kNaN;
  };
  constexpr int NL = (CHT && !FUSED) ? (16 * CHT + 3 * R + 63) / 64 : 1;
  double pre[SPLIT ? 1 : 4][NL];
  int32_t prei[SPLIT ? 4 : 1][NL];
  // (split form) the staged hi words: [row][IP] in the curvature buffer's first bytes. IP = 240: a lane reads word l * 11 + t
  // of its row, and 11 * 35 = 1 (mod 64), so the 16 lanes of a row sit on 16 banks that are consecutive after multiplication
  // by 35; a row offset of 240 = 48 (mod 64) is 16 in that numbering: the four rows take the 64 banks once each. (At the
  // doubles' pitch of 183 the rows were 55 = 5 * 11 banks apart — 11 of a row's 16 banks shared with the next row: measured
  // 1.41 instead of 1.32 ms.)
  constexpr int IP = 240;
  static_assert(!SPLIT || (CHT == 11 && 4 * IP * 4 <= 4 * (16 * 11 + 3 * R) * 8), "bank arithmetic above; the four rows of hi words fit under the doubles");
  int32_t* ibuf = reinterpret_cast<int32_t*>(cbuf);
  const int32_t* __restrict__ chi = reinterpret_cast<const int32_t*>(curv);
  const uint32_t* __restrict__ clo = reinterpret_cast<const uint32_t*>(curv) + n_lines * (size_t)P.W;
  auto fetch_hi = [&](uint32_t s, int rr, int i) -> int32_t {  // (positions outside the line: never valid)
    const int col = (int)(s * P.pps) + lane + 64 * i - R;
    return (lane + 64 * i < BLu && col >= np && col + np < W) ? chi[(line0 + rr) * (size_t)W + col] : (int32_t)0x80000000;
  };
  // ---- fused form (rows a5 + a6 inside this kernel: curvature and validity never reach HBM). A sector of ONE line at a
  // time: its points, with the halo of the curvature sum and of the validity windows, are staged in LDS by all 64 lanes
  // (xa: NX points as they lie in the scan), then: range of every point (sr), the four invalidation codes as wavefront
  // ballots, the validity of the sector's points from those bit strings in scalar registers (features.cpp:20-68 as the
  // gather extract_math.h describes), the curvature of the sector's positions into the row's curvature buffer.
  constexpr int NP = R + 1;
  constexpr int NX = FUSED ? 16 * CHT + 3 * R + 2 * NP : 1;  // columns start - R - NP .. start + 16 CH + 2R + NP - 1
  constexpr int NIA = FUSED ? (3 * NX + 63) / 64 : 1, NRND = FUSED ? (NX + 63) / 64 : 1;
  double* xa = reinterpret_cast<double*>(blk + G.off_xa);
  double* sr = reinterpret_cast<double*>(blk + G.off_sr);
  uint32_t* vsec = reinterpret_cast<uint32_t*>(blk + G.off_vs);
  double nxa[NIA];
  auto fetch_xa = [&](uint32_t s, int rr, int i) -> double {  // scalar lane + 64 i of the staged points of (sector s, row rr)
    const int k = lane + 64 * i;
    const long e = 3l * ((long)(s * P.pps) - R - NP) + k;  // element of the line
    if (k >= 3 * NX || e < 0 || e >= 3l * W) return 0.0;
    const size_t g = (line0 + rr) * (size_t)W * 3 + (size_t)e;
    return fz.f32 ? (double)static_cast<const float*>(fz.xyz)[g] : static_cast<const double*>(fz.xyz)[g];
  };
  if constexpr (FUSED) {
    static_assert(!FUSED || CHT != 0, "the fused form needs the lane chunk at compile time");
#pragma unroll
    for (int i = 0; i < NIA; i++) nxa[i] = fetch_xa(0, 0, i);
    for (int k = lane; k < 4 * (int)G.vw; k += 64) vb[k] = 0;  // fused form: vb holds the SUPPRESSED points of the four lines
  } else if constexpr (SPLIT) {
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
      for (int i = 0; i < NL; i++) prei[rr][i] = fetch_hi(0, rr, i);
    for (int k = lane; k < 4 * (int)G.vw; k += 64) vb[k] = 0;  // the SUPPRESSED points of the four lines
  } else {
    if constexpr (CHT != 0) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int i = 0; i < NL; i++) pre[rr][i] = fetch_curv(0, rr, i);
    }
    // ---- validity bytes (0 / 1, written by the curvature kernels) -> one bit per point
    for (int rr = 0; rr < 4; rr++) {
      const uint8_t* __restrict__ mrow = mask + (line0 + rr) * (size_t)W;
      uint16_t* vrow = reinterpret_cast<uint16_t*>(vb + rr * G.vw);
      for (int c0 = lane * 16; c0 < W; c0 += 1024) {
        const uint4 m = *reinterpret_cast<const uint4*>(mrow + c0);
        auto nib = [](uint32_t w) { return (w | (w >> 7) | (w >> 14) | (w >> 21)) & 0xFu; };
        vrow[c0 >> 4] = (uint16_t)(nib(m.x) | (nib(m.y) << 4) | (nib(m.z) << 8) | (nib(m.w) << 12));
      }
      for (int k = (W >> 4) + lane; k < 2 * (int)G.vw; k += 64) vrow[k] = 0;  // bits past the line's end
    }
  }
  ROWS_STAMP(0)
  const uint32_t line_base = (uint32_t)(line % P.H) * P.W;
  uint32_t tot_e = 0, tot_p = 0;  // picks of my row's line so far
  bool tie = false;               // row-uniform
  const double thr_e = P.edge_thr, thr_p = P.planar_thr;
  for (uint32_t s = 0; s < P.S; s++) {
    const int start = (int)(s * P.pps);
    const int len = (s == P.S - 1) ? W - start : (int)P.pps;  // features-inl.h:31-35
    // ---- stage the sector's curvature of the four lines (compile-time CH: it was fetched during the previous sector)
    wave_lds_sync();
    if constexpr (FUSED) {
#pragma unroll 1
      for (int rr = 0; rr < 4; rr++) {
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < NIA; i++)
          if (lane + 64 * i < 3 * NX) xa[lane + 64 * i] = nxa[i];
        wave_lds_sync();
        {  // the next line's (or the next sector's first line's) points fly during this line's arithmetic
          const uint32_t s2 = rr == 3 ? s + 1 : s;
          if (s2 < P.S) {
#pragma unroll
            for (int i = 0; i < NIA; i++) nxa[i] = fetch_xa(s2, (rr + 1) & 3, i);
          }
        }
        const int c_lo = start - R - NP;  // column of staged point 0
        // ranges (common.h:81-86)
#pragma unroll
        for (int i = 0; i < NRND; i++) {
          const int li = lane + 64 * i;
          if (li < NX) sr[li] = point_range(xa[3 * li], xa[3 * li + 1], xa[3 * li + 2]);
        }
        wave_lds_sync();
        // invalidation codes of the interior points as bit strings over the staged points (bit li of word li / 64), one kind
        // at a time (each is three scalar register pairs):
        // invalid = line end | code 1 within +-NP | code 2 at 1..NP points before | code 3 at 0..NP-1 points after | code 4
        // (extract_math.h: valid_from_codes), on the bit strings: scalar shifts and ors
        uint8_t code[NRND];
        unsigned long long inv[NRND];
#pragma unroll
        for (int i = 0; i < NRND; i++) {
          const int li = lane + 64 * i, col = c_lo + li;
          const bool inside = li < NX && col >= NP && col + NP < W;  // not a line end (features-inl.h:66-67), inside the line
          code[i] = kCodeNone;
          if (inside && li >= 1 && li + 1 < NX) code[i] = point_code(sr[li - 1], sr[li], sr[li + 1], P);
          inv[i] = __ballot(!inside || code[i] == kCodeParallel);
        }
        {
          unsigned long long m[NRND];
#pragma unroll
          for (int i = 0; i < NRND; i++) m[i] = __ballot(code[i] == kCodeRange), inv[i] |= m[i];
#pragma unroll
          for (int d = 1; d <= NP; d++)
#pragma unroll
            for (int i = 0; i < NRND; i++)
              inv[i] |= (m[i] << d) | (i > 0 ? m[i - 1] >> (64 - d) : 0ull) | (m[i] >> d) | (i + 1 < NRND ? m[i + 1] << (64 - d) : 0ull);
#pragma unroll
          for (int i = 0; i < NRND; i++) m[i] = __ballot(code[i] == kCodeOcc1);
#pragma unroll
          for (int d = 1; d <= NP; d++)
#pragma unroll
            for (int i = 0; i < NRND; i++) inv[i] |= (m[i] << d) | (i > 0 ? m[i - 1] >> (64 - d) : 0ull);  // code at li - d
#pragma unroll
          for (int i = 0; i < NRND; i++) m[i] = __ballot(code[i] == kCodeOcc2), inv[i] |= m[i];
#pragma unroll
          for (int d = 1; d <= NP - 1; d++)
#pragma unroll
            for (int i = 0; i < NRND; i++) inv[i] |= (m[i] >> d) | (i + 1 < NRND ? m[i + 1] << (64 - d) : 0ull);  // code at li + d
        }
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < NRND; i++) {
            vsec[rr * 8 + 2 * i] = (uint32_t)~inv[i], vsec[rr * 8 + 2 * i + 1] = (uint32_t)(~inv[i] >> 32);
          }
#pragma unroll
          for (int i = 2 * NRND; i < 8; i++) vsec[rr * 8 + i] = 0;
        }
        // curvature of the sector's positions -R .. 16 CH + 2R - 1 (features-inl.h:73-82, the association order of curvature_at):
        // three consecutive positions per lane, one coordinate at a time
        {
          const int k0 = 3 * lane;  // buffer index k <-> staged point k + NP <-> column start - R + k
          if (k0 < BLu) {
            double d[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
              double w[2 * NP + 3];
#pragma unroll
              for (int u = 0; u < 2 * NP + 3; u++) w[u] = k0 + u < NX ? xa[3 * (k0 + u) + a] : 0.0;
#pragma unroll
              for (int i = 0; i < 3; i++) {
                double acc = -(2.0 * NP) * w[NP + i];
#pragma unroll
                for (int n = 1; n <= NP; n++) acc = acc + w[NP + i - n] + w[NP + i + n];
                d[i][a] = acc;
              }
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
              const int col = start - R + k0 + i;
              const double cv = d[i][0] * d[i][0] + d[i][1] * d[i][1] + d[i][2] * d[i][2];
              if (k0 + i < BLu) cbuf[(size_t)rr * G.pitch + k0 + i] = (col >= NP && col + NP < W) ? cv : kNaN;
            }
          }
        }
      }
    } else if constexpr (SPLIT) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int i = 0; i < NL; i++)
          if (lane + 64 * i < BLu) ibuf[rr * IP + lane + 64 * i] = prei[rr][i];
      if (s + 1 < P.S) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
          for (int i = 0; i < NL; i++) prei[rr][i] = fetch_hi(s + 1, rr, i);
      }
    } else if constexpr (CHT != 0) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int i = 0; i < NL; i++)
          if (lane + 64 * i < BLu) cbuf[(size_t)rr * G.pitch + lane + 64 * i] = pre[rr][i];
      if (s + 1 < P.S) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
          for (int i = 0; i < NL; i++) pre[rr][i] = fetch_curv(s + 1, rr, i);
      }
    } else {
      for (int rr = 0; rr < 4; rr++)
        for (int i = 0; lane + 64 * i < BLu; i++) cbuf[(size_t)rr * G.pitch + lane + 64 * i] = fetch_curv(s, rr, i);
    }
    wave_lds_sync();
    ROWS_STAMP(1)
    // ---- the lane's comparison masks. Window bit t <-> sector position l * CH - R + t; own bit j <-> window bit j + R.
    uint32_t gt[R], eqm[R], ET = 0, PT = 0;
    int32_t hiw[CHT ? CHT : 1];  // (compile-time CH) hi words of my own curvatures: the sort keys come from them
    bool anyeq = false;
    bool is_dbl = !SPLIT;  // (uniform) the curvature buffer holds doubles
    uint32_t VS = 0;       // (split form) validity of my CH points: the sign bits of their hi words
    // (split form) the sector's lo words, and with them the doubles into the curvature buffer — every lane its staged positions
    auto ensure = [&]() {
      if constexpr (SPLIT) {
        if (is_dbl) return;  // uniform
        is_dbl = true;
        // row by row from the last one down: the doubles of row rr lie over the hi words of rows 2 rr and 2 rr + 1, which
        // are converted by then (row 0 over itself: read, barrier, write); a loop, not unrolled — this path is cold and
        // its registers must not cost the common one anything
#pragma unroll 1
        for (int rr = 3; rr >= 0; rr--) {
          int32_t h[NL];
          uint32_t lo_[NL];
#pragma unroll
          for (int i = 0; i < NL; i++) {
            h[i] = lane + 64 * i < BLu ? ibuf[rr * IP + lane + 64 * i] : (int32_t)0x80000000;
            const int col = start + lane + 64 * i - R;  // (a non-negative hi word lies inside the line)
            lo_[i] = h[i] >= 0 ? clo[(line0 + rr) * (size_t)W + col] : 0u;
          }
          wave_lds_sync();
#pragma unroll
          for (int i = 0; i < NL; i++)
            if (lane + 64 * i < BLu) cbuf[(size_t)rr * G.pitch + lane + 64 * i] = h[i] >= 0 ? __hiloint2double(h[i], (int)lo_[i]) : kNaN;
          wave_lds_sync();
        }
      }
    };
#pragma unroll
    for (int d = 0; d < R; d++) gt[d] = 0, eqm[d] = 0;
    if constexpr (SPLIT) {
      const int32_t* cpi = ibuf + row * IP + l * CH;
      const int32_t te = __double2hiint(thr_e), tp = __double2hiint(thr_p);
      int32_t w[R + 1];
#pragma unroll
      for (int d = 1; d <= R; d++) w[d] = cpi[CH + 2 * R - 1 + d];
#pragma unroll
      for (int t = CH + 2 * R - 1; t >= 0; t--) {
        w[0] = cpi[t];
        const bool v0 = w[0] >= 0;
#pragma unroll
        for (int d = 1; d <= R; d++) {
          push_gt_i32(gt[d - 1], w[0], w[d]);
          anyeq = anyeq || (v0 && w[0] == w[d]);
        }
        if (t >= R && t < R + CH) {
          push_gt_i32(ET, w[0], te), push_lt_i32(PT, w[0], tp), push_ge0_i32(VS, w[0]);
          anyeq = anyeq || (v0 && (w[0] == te || w[0] == tp));
          hiw[t - R] = w[0];
        }
#pragma unroll
        for (int d = R; d >= 1; d--) w[d] = w[d - 1];
      }
      if (__ballot(anyeq) != 0) {  // (uniform) equal hi words somewhere: this sector is compared as doubles
        ensure();
#pragma unroll
        for (int d = 0; d < R; d++) gt[d] = 0;
        ET = 0, PT = 0, anyeq = false;
      }
    }
    if (is_dbl) {  // (uniform)
      double w[R + 1];
#pragma unroll
      for (int d = 1; d <= R; d++) w[d] = X.cp[CH + 2 * R - 1 + d];
#pragma unroll
      for (int t = CH + 2 * R - 1; t >= 0; t--) {
        w[0] = X.cp[t];
#pragma unroll
        for (int d = 1; d <= R; d++) {
          push_gt(gt[d - 1], w[0], w[d]);
          anyeq = anyeq || w[0] == w[d];
        }
        if (t >= R && t < R + CH) {
          push_gt(ET, w[0], thr_e), push_lt(PT, w[0], thr_p);
          if constexpr (CHT != 0 && !SPLIT) hiw[t - R] = __double2hiint(w[0]);
        }
#pragma unroll
        for (int d = R; d >= 1; d--) w[d] = w[d - 1];
      }
    }
    const bool have_eq = __ballot(anyeq) != 0;
    if (have_eq) {  // (uniform, rare: noise-free scans) the equality masks of the tie test
      for (int t = 0; t < CH + 2 * R; t++) {
        const double a = X.cp[t];
#pragma unroll
        for (int d = 1; d <= R; d++) eqm[d - 1] |= (uint32_t)(a == X.cp[t + d]) << t;
      }
    }
    ROWS_STAMP(2)
    // ---- the line's current validity bits of my CH points (one unaligned two-word read)
    const uint32_t o = (uint32_t)(start + l * CH), wv = o >> 5, sh = o & 31u;
    const uint32_t* vrow = vb + row * G.vw;
    uint32_t V0 = __builtin_amdgcn_alignbit(vrow[wv + 1], vrow[wv], sh) & X.cm;
    if constexpr (FUSED) {  // valid by the codes of this sector's staged points, and not suppressed by an earlier pick of the line
      const uint32_t bo = (uint32_t)(l * CH + R + NP);  // staged point of my first position
      const uint32_t* vs = vsec + row * 8 + (bo >> 5);
      V0 = __builtin_amdgcn_alignbit(vs[1], vs[0], bo & 31u) & X.cm & ~V0;
    }
    if constexpr (SPLIT) V0 = VS & X.cm & ~V0;
    uint32_t V = V0;
    int nb = len - l * CH;
    nb = nb < 0 ? 0 : (nb > CH ? CH : nb);
    const uint32_t sm = (1u << nb) - 1u;
    const size_t group = line * P.S + s;
    unsigned char* pk_s = pk_row + ((size_t)s * (G.cap_e + G.cap_p) << (G.pk8 ? 0 : 1));
    const uint32_t ne = row_pass<R, true, CHT>(X, V, ET, sm, gt, eqm, have_eq, tie, G.cap_e, start, line_base, st.edge_stage + group * P.cap_edge, pk_s, hiw,
                                               ensure);
    ROWS_STAMP(3)
    const uint32_t npl = row_pass<R, false, CHT>(X, V, PT, sm, gt, eqm, have_eq, tie, G.cap_p, start, line_base, st.planar_stage + group * P.cap_planar,
                                                 pk_s + ((size_t)G.cap_e << (G.pk8 ? 0 : 1)), hiw, ensure);
    ROWS_STAMP(4)
    if (l == 0) {
      if (!X.no_stage) st.edge_cnt[group] = ne, st.planar_cnt[group] = npl;
      cnt[2 * s] = (uint16_t)ne, cnt[2 * s + 1] = (uint16_t)npl;  // (for the copy phase)
    }
    tot_e += ne, tot_p += npl;
    // ---- give the cleared bits back to the line (they may belong to the next sector's first points)
    const uint32_t clr = V0 & ~V;
    if (clr) {
      const unsigned long long mm = (unsigned long long)clr << sh;
      uint32_t* vw_ = vb + row * G.vw + wv;
      if constexpr (FUSED || SPLIT) {
        if ((uint32_t)mm) __hip_atomic_fetch_or(vw_, (uint32_t)mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(mm >> 32)) __hip_atomic_fetch_or(vw_ + 1, (uint32_t)(mm >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        if ((uint32_t)mm) __hip_atomic_fetch_and(vw_, ~(uint32_t)mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(mm >> 32)) __hip_atomic_fetch_and(vw_ + 1, ~(uint32_t)(mm >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  wave_lds_sync();
  // ---- publish the lines' totals (or the tie mark: replay_kernel redoes such a line in the reference's own order; the lines
  // behind it in the scan see the mark, stop waiting and leave their picks in the stage arrays for the fallback compaction)
  if (P.flags & kFlagForceReplay) tie = true;
  if (l == 0 && fz.line_tot) {
    if (tie) {
      __hip_atomic_store(fz.line_tot + line, kLinePublished | kLineTied, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(fz.error, kFlagTie | kFlagGaveUp);
    } else if (fz.fuse) {
      __hip_atomic_store(fz.line_tot + line, kLinePublished | ((unsigned long long)tot_e << 32) | (unsigned long long)tot_p, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!fz.fuse) return;  // uniform
  // ---- fused compaction: totals of the lines before mine in the scan = the published totals of the lines before this
  // wavefront's four (chained scan, bounded wait: workgroups start in index order) + the rows before mine
  const uint32_t li0 = (uint32_t)(line0 % P.H);
  const size_t scan = line0 / P.H;
  uint32_t base_e = 0, base_p = 0;
  bool gave_up = false;
  for (uint32_t c0 = 0; c0 < li0; c0 += 64) {
    const uint32_t j = c0 + (uint32_t)lane;
    unsigned long long t = 1ull << 63;  // no predecessor on this lane: nothing to wait for, contributes 0
    if (j < li0) {
      const unsigned long long* src = fz.line_tot + (line0 - li0 + j);
      t = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t spins = 0; !(t >> 63) && spins < kLookbackSpins; spins++) {
        __builtin_amdgcn_s_sleep(4);
        t = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    gave_up = gave_up || !(t >> 63) || (t & kLineTied) != 0ull;
    const uint32_t se = (uint32_t)(t >> 32) & 0x3FFFFFFFu, sp = (uint32_t)t;
    base_e += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(se), 63);
    base_p += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(sp), 63);
  }
  ROWS_STAMP(5)
  bool stay = __ballot(gave_up) != 0;  // uniform so far; below: per row
#pragma unroll
  for (int r = 0; r < 3; r++) {  // the rows before mine
    const uint32_t e_r = (uint32_t)__builtin_amdgcn_readlane((int)tot_e, 16 * r), p_r = (uint32_t)__builtin_amdgcn_readlane((int)tot_p, 16 * r);
    const bool tie_r = __builtin_amdgcn_readlane((int)tie, 16 * r) != 0;
    if (row > r) base_e += e_r, base_p += p_r, stay = stay || tie_r;
  }
  if ((P.flags & kFlagForceGiveUp) && li0 + (uint32_t)row > 0) stay = true;
  if (stay && !tie && l == 0) atomicOr(fz.error, kFlagGaveUp);  // my line's features stay in the stage arrays (complete, as always)
  const bool go = !stay && !tie;
  {
    // ---- the copy: (sector, line) by (sector, line), ALL 64 lanes on one line's sector at a time. The sector's points are
    // read with coalesced loads into LDS (the curvature buffer is free now) instead of one scattered 24-byte read per pick:
    // a sector's picks touch every 128-byte line of it anyway (every 3.7th point is picked, in curvature order), so the
    // gather fetched the whole scan plus re-fetches and kept the texture addresser busy with 64 lines per instruction.
    // The picked points leave through LDS as well, so that the stores are runs of consecutive doubles.
    // (the index arrays are optional: loamx_register_scan_pairs_dev wants the points only)
    uint32_t* __restrict__ oe = fz.edge_idx ? fz.edge_idx + scan * fz.edge_stride : nullptr;
    uint32_t* __restrict__ op = fz.planar_idx ? fz.planar_idx + scan * fz.planar_stride : nullptr;
    const bool want_idx = oe != nullptr && op != nullptr;  // uniform
    double* __restrict__ xe = fz.edge_xyz ? fz.edge_xyz + scan * fz.edge_stride * 3 : nullptr;
    double* __restrict__ xp = fz.planar_xyz ? fz.planar_xyz + scan * fz.planar_stride * 3 : nullptr;
    const size_t scan_pt0 = scan * (size_t)P.H * P.W;  // first point of the scan
    const uint32_t longest = P.W - (P.S - 1) * P.pps;
    uint32_t run_e[4], run_p[4];
    bool rgo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      run_e[r] = (uint32_t)__builtin_amdgcn_readlane((int)base_e, 16 * r), run_p[r] = (uint32_t)__builtin_amdgcn_readlane((int)base_p, 16 * r);
      rgo[r] = __builtin_amdgcn_readlane((int)go, 16 * r) != 0;
    }
    // bounding boxes of the scan's edge / planar features (ExtractFused::box_*): every lane keeps the minima / maxima of the
    // picks it copies; one reduction and twelve atomics per wavefront at the end
    const bool want_box = fz.box_min != nullptr;  // uniform
    double bmin[2][3], bmax[2][3];
#pragma unroll
    for (int kd = 0; kd < 2; kd++)
#pragma unroll
      for (int a = 0; a < 3; a++) bmin[kd][a] = 1.7976931348623157e308, bmax[kd][a] = -1.7976931348623157e308;
    double* xbuf = cbuf;                        // 3 * len doubles of one line's sector (the selection's buffers are free now)
    double* obuf = cbuf + 3 * (size_t)longest;  // 64 picked points on their way out (row_select_geom: both fit below off_sl)
    const unsigned char* pk_all = blk + G.off_pk;
    const uint16_t* cnt_all = reinterpret_cast<const uint16_t*>(blk + G.off_cnt);
    constexpr int NI = CHT ? (3 * 16 * CHT + 63) / 64 : 1;  // scalars per lane of a sector's points (compile-time CH: prefetched)
    double nx[NI];
    auto fetch = [&](uint32_t s, int rr, int i) -> double {  // scalar lane + 64 i of the points of (sector s, row rr)
      const int start = (int)(s * P.pps), len = (s == P.S - 1) ? W - start : (int)P.pps;
      const int k = lane + 64 * i;
      const size_t e0 = (scan_pt0 + (size_t)(li0 + (uint32_t)rr) * P.W + (size_t)start) * 3 + (size_t)k;
      if (k >= 3 * len) return 0.0;
      return fz.f32 ? (double)static_cast<const float*>(fz.xyz)[e0] : static_cast<const double*>(fz.xyz)[e0];
    };
    if constexpr (CHT != 0) {
#pragma unroll
      for (int i = 0; i < NI; i++) nx[i] = fetch(0, 0, i);
    }
    for (uint32_t s = 0; s < P.S; s++) {
      const int start = (int)(s * P.pps), len = (s == P.S - 1) ? W - start : (int)P.pps;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        // (reads that do not depend on the staged points first)
        const uint32_t ce = cnt_all[rr * 32 + 2 * s], cq = cnt_all[rr * 32 + 2 * s + 1];  // (uniform)
        const unsigned char* pk_s = pk_all + (((size_t)rr * G.pk_stride + (size_t)s * (G.cap_e + G.cap_p)) << (G.pk8 ? 0 : 1));
        auto pick_pos = [&](uint32_t i) -> uint32_t {  // sector position of the i-th pick of (s, rr): edge picks first
          const uint32_t e = (i < ce ? i : i - ce + G.cap_e);
          return G.pk8 ? (uint32_t)pk_s[e] : (uint32_t)reinterpret_cast<const uint16_t*>(pk_s)[e];
        };
        uint32_t pos0 = (uint32_t)lane < ce + cq ? pick_pos((uint32_t)lane) : 0u;
        wave_lds_sync();
        if constexpr (CHT != 0) {
#pragma unroll
          for (int i = 0; i < NI; i++) xbuf[lane + 64 * i] = nx[i];
          const uint32_t s2 = rr == 3 ? s + 1 : s;
          if (s2 < P.S) {
#pragma unroll
            for (int i = 0; i < NI; i++) nx[i] = fetch(s2, (rr + 1) & 3, i);
          }
        } else {
          for (int i = 0; lane + 64 * i < 3 * len; i++) xbuf[lane + 64 * i] = fetch(s, rr, i);
        }
        wave_lds_sync();
        if (rgo[rr]) {
          const uint32_t lb = (li0 + (uint32_t)rr) * P.W + (uint32_t)start;
          for (uint32_t i0 = 0; i0 < ce + cq; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool on = i < ce + cq, edge = i < ce;
            const uint32_t jj = edge ? i : i - ce;
            const uint32_t pos = i0 == 0 ? pos0 : (on ? pick_pos(i) : 0u);
            const double* src = xbuf + 3 * pos;
            const double x = src[0], y = src[1], z = src[2];
            if (on && want_idx) (edge ? oe : op)[(edge ? run_e[rr] : run_p[rr]) + jj] = lb + pos;
            if (want_box && on) {
              if (edge) {
                bmin[0][0] = fmin(bmin[0][0], x), bmin[0][1] = fmin(bmin[0][1], y), bmin[0][2] = fmin(bmin[0][2], z);
                bmax[0][0] = fmax(bmax[0][0], x), bmax[0][1] = fmax(bmax[0][1], y), bmax[0][2] = fmax(bmax[0][2], z);
              } else {
                bmin[1][0] = fmin(bmin[1][0], x), bmin[1][1] = fmin(bmin[1][1], y), bmin[1][2] = fmin(bmin[1][2], z);
                bmax[1][0] = fmax(bmax[1][0], x), bmax[1][1] = fmax(bmax[1][1], y), bmax[1][2] = fmax(bmax[1][2], z);
              }
            }
            obuf[3 * lane] = x, obuf[3 * lane + 1] = y, obuf[3 * lane + 2] = z;
            wave_lds_sync();
            // this round holds the edge picks [e0, e0 + ne) and the planar picks [p0, ...) of the sector, edge first
            const uint32_t e0 = i0 < ce ? i0 : ce, p0 = (i0 > ce ? i0 : ce) - ce;
            const uint32_t n_here = ce + cq - i0 < 64 ? ce + cq - i0 : 64;
            const uint32_t ne = ce - e0 < n_here ? ce - e0 : n_here;
            for (uint32_t t = (uint32_t)lane; t < 3 * n_here; t += 64) {
              const double v = obuf[t];
              if (t < 3 * ne) {
                if (xe) xe[3 * (size_t)(run_e[rr] + e0) + t] = v;
              } else if (xp) {
                xp[3 * (size_t)(run_p[rr] + p0) + (t - 3 * ne)] = v;
              }
            }
            wave_lds_sync();
          }
        }
        run_e[rr] += ce, run_p[rr] += cq;
      }
    }
    if (want_box) {  // (the four lines belong to one scan)
#pragma unroll
      for (int kd = 0; kd < 2; kd++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
          double lo = bmin[kd][a], hi = bmax[kd][a];
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) lo = fmin(lo, __shfl_xor(lo, off)), hi = fmax(hi, __shfl_xor(hi, off));
          if (lane == 0 && lo <= hi) {
            atomicMin(fz.box_min + scan * 6 + kd * 3 + a, dbl_key(lo));
            atomicMax(fz.box_max + scan * 6 + kd * 3 + a, dbl_key(hi));
          }
        }
    }
    ROWS_STAMP(6)
#ifdef LOAMX_ROWS_PROFILE
    if (lane == 0 && blockIdx.x % 1500 == 700 && wave == 1)
      printf("rows wave %u: total %llu | vbits %llu stage %llu bits %llu edge %llu planar %llu (+misc) lookback %llu copy %llu\n", blockIdx.x,
             __builtin_amdgcn_s_memtime() - pt_begin, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6]);
#endif
    if (rgo[3] && lane == 0 && li0 + 3 == P.H - 1) {
      fz.n_edge[scan] = run_e[3], fz.n_planar[scan] = run_p[3];
      if (fz.events) atomicAdd(&fz.events[2], (unsigned long long)(run_e[3] + run_p[3]));  // (roofline bytes of the fused kernel)
    }
  }
}

template <int R, int CHT, bool FUSED = false, bool SPLIT = false>
__global__ __launch_bounds__(FUSED ? 128 : 256, FUSED ? 3 : 4) void select_rows_kernel(const double* __restrict__ curv, const uint8_t* __restrict__ mask, size_t n_lines,
                                                          ExtractParams P, ExtractStage st, ExtractFused fz, RowSelGeom G) {
  select_rows_body<R, CHT, FUSED, SPLIT>(curv, mask, n_lines, P, st, fz, G);
}
// The same kernel under another name: the conditional second launch behind a fused selection (ExtractFused::only_if), which
// leaves at once unless a line gave up or was tied — a name of its own so that profiles do not average it with the first.
template <int R, int CHT, bool SPLIT>
__global__ __launch_bounds__(256, 4) void select_rows_stage_kernel(const double* __restrict__ curv, const uint8_t* __restrict__ mask, size_t n_lines,
                                                                   ExtractParams P, ExtractStage st, ExtractFused fz, RowSelGeom G) {
  select_rows_body<R, CHT, false, SPLIT>(curv, mask, n_lines, P, st, fz, G);
}

#undef LOAMX_CX
#undef LOAMX_CX_BANK
#undef LOAMX_CX_SEL
