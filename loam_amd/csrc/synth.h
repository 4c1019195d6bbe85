// synth.h — bit-reproducible synthetic LiDAR scan-pair generator.
//
// Workload generator for tests and bench.py, NOT part of the reference path. It uses only
// + - * / sqrt, integer hashing and fixed-order polynomial evaluation (no libm), so the same
// (seed, pair, scan, beam) produces bit-identical doubles on any host and on the GPU, provided the
// translation unit is compiled with -ffp-contract=off (both g++ and hipcc builds do).
//
// Scene: axis-aligned room [-10,10]x[-8,8]x[-2,4] m seen from inside, with four boxes; sensor A at
// the world origin, sensor B at pose T_true (|rot| <= 0.07 rad, |t| <= 0.3 m). A scan is H rings x
// W azimuth steps, elevation linear in [-22.5, +22.5] deg, azimuth 2*pi*c/W, row-major [ring][col],
// points in the sensor frame, range noise sigma * ~N(0,1) (12-term Irwin-Hall on hashed 16-bit
// integers). target_T_source for (source = scan B, target = scan A) is T_true.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SYNTH_HD __host__ __device__ inline
#else
#include <math.h>
#define SYNTH_HD inline
#endif

namespace loamx_synth {

SYNTH_HD double det_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __dsqrt_rn(x);
#else
  return sqrt(x);
#endif
}

SYNTH_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
SYNTH_HD uint64_t hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  return mix64(mix64(mix64(mix64(a) ^ b) ^ c) ^ d);
}
// uniform in [0,1) with 53 bits
SYNTH_HD double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// sin/cos for |x| <= pi/4 (+ a little), fixed Horner order => deterministic
SYNTH_HD void det_sincos_small(double x, double* s, double* c) {
  const double x2 = x * x;
  // sin: x * (1 - x2/6 + x2^2/120 - ... ) up to x^19
  double ps = -1.0 / 121645100408832000.0;          // -1/19!
  ps = ps * x2 + 1.0 / 355687428096000.0;           // 1/17!
  ps = ps * x2 - 1.0 / 1307674368000.0;             // -1/15!
  ps = ps * x2 + 1.0 / 6227020800.0;                // 1/13!
  ps = ps * x2 - 1.0 / 39916800.0;                  // -1/11!
  ps = ps * x2 + 1.0 / 362880.0;                    // 1/9!
  ps = ps * x2 - 1.0 / 5040.0;                      // -1/7!
  ps = ps * x2 + 1.0 / 120.0;                       // 1/5!
  ps = ps * x2 - 1.0 / 6.0;                         // -1/3!
  ps = ps * x2 + 1.0;
  *s = x * ps;
  double pc = 1.0 / 6402373705728000.0;             // 1/18!
  pc = pc * x2 - 1.0 / 20922789888000.0;            // -1/16!
  pc = pc * x2 + 1.0 / 87178291200.0;               // 1/14!
  pc = pc * x2 - 1.0 / 479001600.0;                 // -1/12!
  pc = pc * x2 + 1.0 / 3628800.0;                   // 1/10!
  pc = pc * x2 - 1.0 / 40320.0;                     // -1/8!
  pc = pc * x2 + 1.0 / 24.0;                        // 1/4!
  pc = pc * x2 - 0.5;                               // -1/2!
  pc = pc * x2 + 1.0;
  *c = pc;
}

constexpr double kPi = 3.14159265358979323846;

// sin/cos of 2*pi*c/W by exact integer octant reduction
SYNTH_HD void azimuth_sincos(uint32_t col, uint32_t W, double* s, double* c) {
  const uint64_t num = 8ull * col;
  const uint32_t oct = (uint32_t)(num / W);
  const uint32_t rem = (uint32_t)(num - (uint64_t)oct * W);
  double ss, cc;
  if ((oct & 1u) == 0) {
    det_sincos_small((kPi / 4.0) * ((double)rem / (double)W), &ss, &cc);
  } else {
    double s2, c2;
    det_sincos_small((kPi / 4.0) * ((double)(W - rem) / (double)W), &s2, &c2);
    ss = c2;
    cc = s2;
  }
  switch ((oct >> 1) & 3u) {
    case 0: *c = cc, *s = ss; break;
    case 1: *c = -ss, *s = cc; break;
    case 2: *c = -cc, *s = -ss; break;
    default: *c = ss, *s = -cc; break;
  }
}

// elevation of ring l of H: linear in [-22.5, 22.5] degrees
SYNTH_HD void elevation_sincos(uint32_t line, uint32_t H, double* s, double* c) {
  const double frac = (H > 1) ? (double)line / (double)(H - 1) : 0.5;
  det_sincos_small(kPi * (-0.125 + 0.25 * frac), s, c);
}

struct Pose7 {
  double q[4];  // x,y,z,w
  double t[3];
};

SYNTH_HD void quat_rotate(const double q[4], const double v[3], double out[3]) {
  // v + w*(2 u x v) + u x (2 u x v)
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] = uv[0] + uv[0], uv[1] = uv[1] + uv[1], uv[2] = uv[2] + uv[2];
  out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

// Ground-truth pose of sensor B in the frame of sensor A (= target_T_source for source=B, target=A)
SYNTH_HD Pose7 pair_pose(uint64_t seed, uint64_t pair_id) {
  Pose7 P;
  double ax[3], n2;
  uint64_t k = 0;
  do {  // random axis, rejection keeps it away from zero length
    for (int i = 0; i < 3; i++) ax[i] = 2.0 * u01(hash4(seed, pair_id, 0x706f7365ull, k++)) - 1.0;
    n2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
  } while (n2 < 1e-3 || n2 > 1.0);
  const double n = det_sqrt(n2);
  const double angle = 0.07 * u01(hash4(seed, pair_id, 0x616e676cull, 0));
  double s, c;
  det_sincos_small(0.5 * angle, &s, &c);
  for (int i = 0; i < 3; i++) P.q[i] = s * (ax[i] / n);
  P.q[3] = c;
  for (int i = 0; i < 3; i++)
    P.t[i] = (0.3 / 1.7320508075688772) * (2.0 * u01(hash4(seed, pair_id, 0x7472616eull, (uint64_t)i)) - 1.0);
  return P;
}

// range along a ray from o (inside the room) in direction d (unit): room exit or first box entry
SYNTH_HD double cast_ray(const double o[3], const double d[3]) {
  const double room_lo[3] = {-10.0, -8.0, -2.0}, room_hi[3] = {10.0, 8.0, 4.0};
  const double boxes[4][6] = {{4.0, 2.0, -2.0, 4.6, 2.6, 4.0},
                              {-6.0, -3.0, -2.0, -5.2, -2.2, 4.0},
                              {2.0, -5.0, -2.0, 3.0, -4.0, 1.0},
                              {-3.0, 5.0, -2.0, -2.4, 5.6, 4.0}};
  double t_hit = 1e300;
  for (int i = 0; i < 3; i++) {
    if (d[i] > 0.0) {
      const double t = (room_hi[i] - o[i]) / d[i];
      if (t < t_hit) t_hit = t;
    } else if (d[i] < 0.0) {
      const double t = (room_lo[i] - o[i]) / d[i];
      if (t < t_hit) t_hit = t;
    }
  }
  for (int b = 0; b < 4; b++) {
    double tmin = 0.0, tmax = 1e300;
    bool miss = false;
    for (int i = 0; i < 3; i++) {
      const double lo = boxes[b][i], hi = boxes[b][3 + i];
      if (d[i] == 0.0) {
        if (o[i] < lo || o[i] > hi) miss = true;
      } else {
        double t1 = (lo - o[i]) / d[i], t2 = (hi - o[i]) / d[i];
        if (t1 > t2) {
          const double tt = t1;
          t1 = t2;
          t2 = tt;
        }
        if (t1 > tmin) tmin = t1;
        if (t2 < tmax) tmax = t2;
      }
    }
    if (!miss && tmin <= tmax && tmin > 0.0 && tmin < t_hit) t_hit = tmin;
  }
  return t_hit;
}

// One beam of scan `which` (0 = A at identity, 1 = B at pair_pose) of pair `pair_id`.
SYNTH_HD void scan_point(uint64_t seed, uint64_t pair_id, uint32_t which, const Pose7& poseB, uint32_t line,
                         uint32_t col, uint32_t H, uint32_t W, double sigma, double out[3]) {
  double se, ce, sa, ca;
  elevation_sincos(line, H, &se, &ce);
  azimuth_sincos(col, W, &sa, &ca);
  const double d[3] = {ce * ca, ce * sa, se};
  double o[3] = {0.0, 0.0, 0.0}, dw[3] = {d[0], d[1], d[2]};
  if (which) {
    quat_rotate(poseB.q, d, dw);
    o[0] = poseB.t[0], o[1] = poseB.t[1], o[2] = poseB.t[2];
  }
  double range = cast_ray(o, dw);
  // Irwin-Hall(12) on 16-bit integers: exact integer sum, one conversion
  const uint64_t pt = (uint64_t)line * W + col;
  int64_t acc = 0;
  for (uint64_t k = 0; k < 3; k++) {
    const uint64_t h = hash4(seed, pair_id * 2 + which, pt, k);
    acc += (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) +
           (int64_t)((h >> 48) & 0xFFFF);
  }
  const double noise = (double)(acc - 6 * 65535) * (1.0 / 65536.0);
  range = range + sigma * noise;
  out[0] = range * d[0], out[1] = range * d[1], out[2] = range * d[2];
}

}  // namespace loamx_synth
