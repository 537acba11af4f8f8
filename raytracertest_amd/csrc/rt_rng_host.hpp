// rt_rng_host.hpp -- host side of the XORWOW state creation.
//
// curand_init(seed, subsequence, 0) (call site RayTracer/Random.cu:22-27) = seed scramble
// + a jump of subsequence * 2^67 steps of the xorshift part.  The xorshift step is linear
// over GF(2) on 160 bits, so the jump is the matrix J = T^(2^67); we build J by 67
// squarings of the one-step matrix T and then J^(2^k), k = 0..31, which the device kernel
// multiplies together according to the bits of the pixel index.  Nothing is copied from
// cuRAND's precomputed tables (they are not available here); the matrices follow from the
// recurrence alone.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace rth {

struct Gf2Mat { uint32_t col[160][5]; };   // column i = image of basis vector e_i

inline void xorshift_step(uint32_t v[5]) {
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

inline void mat_vec(const Gf2Mat& A, const uint32_t x[5], uint32_t out[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 160; ++i)
    if ((x[i >> 5] >> (i & 31)) & 1u)
      for (int w = 0; w < 5; ++w) r[w] ^= A.col[i][w];
  memcpy(out, r, sizeof r);
}

inline void mat_mul(const Gf2Mat& A, const Gf2Mat& B, Gf2Mat& out) {   // out = A * B
  Gf2Mat tmp;
  for (int i = 0; i < 160; ++i) mat_vec(A, B.col[i], tmp.col[i]);
  out = tmp;
}

// jump[k] = (T^(2^67))^(2^k), k = 0..31, as 160 columns padded to 8 words (device layout)
inline std::vector<uint32_t> build_jump_table() {
  Gf2Mat m;
  for (int i = 0; i < 160; ++i) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[i >> 5] = 1u << (i & 31);
    xorshift_step(v);
    memcpy(m.col[i], v, sizeof v);
  }
  for (int i = 0; i < 67; ++i) mat_mul(m, m, m);
  std::vector<uint32_t> table(32u * 160u * 8u, 0u);
  for (int k = 0; k < 32; ++k) {
    for (int i = 0; i < 160; ++i)
      memcpy(&table[(static_cast<size_t>(k) * 160u + i) * 8u], m.col[i], 5 * sizeof(uint32_t));
    mat_mul(m, m, m);
  }
  return table;
}

// 4-bit window tables of the six lowest jump powers J^(2^m), m = 0..5 (what distinguishes the 64 pixels of a
// wave): entry [m][g][n] = XOR of the columns g*4 + j of J^(2^m) over the set bits j of the nibble n, so that a
// matrix-vector product is 40 table look-ups instead of 160 masked column XORs (rng_init_kernel keeps the
// tables in LDS).  Device layout: 6*40*16 entries of words 0..3 (16 bytes each), then the same entries' word 4.
constexpr uint32_t kWindowMatrices = 6, kWindowGroups = 40, kWindowEntries = kWindowMatrices * kWindowGroups * 16u;
inline std::vector<uint32_t> build_window_tables(const std::vector<uint32_t>& jump) {
  std::vector<uint32_t> t(static_cast<size_t>(kWindowEntries) * 5u, 0u);
  for (uint32_t m = 0; m < kWindowMatrices; ++m)
    for (uint32_t g = 0; g < kWindowGroups; ++g)
      for (uint32_t n = 0; n < 16u; ++n) {
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < 4u; ++j)
          if ((n >> j) & 1u)
            for (int w = 0; w < 5; ++w) r[w] ^= jump[(static_cast<size_t>(m) * 160u + g * 4u + j) * 8u + w];
        const size_t e = (static_cast<size_t>(m) * kWindowGroups + g) * 16u + n;
        memcpy(&t[e * 4u], r, 4 * sizeof(uint32_t));
        t[static_cast<size_t>(kWindowEntries) * 4u + e] = r[4];
      }
  return t;
}

// seed scramble of curand_init: state = {d, v0..v4}
inline void seed_state(uint64_t seed, uint32_t s[6]) {
  const uint32_t lo = static_cast<uint32_t>(seed) ^ 0xaad26b49u;
  const uint32_t hi = static_cast<uint32_t>(seed >> 32) ^ 0xf7dcefddu;
  const uint32_t t0 = 1099087573u * lo;
  const uint32_t t1 = 2591861531u * hi;
  s[0] = 6615241u + t1 + t0;
  s[1] = 123456789u + t0;
  s[2] = 362436069u ^ t0;
  s[3] = 521288629u + t1;
  s[4] = 88675123u ^ t1;
  s[5] = 5783321u + t0;
}

// host evaluation of curand_init(seed, subsequence, 0) from the same table (debug/tests)
inline void init_state(const std::vector<uint32_t>& table, uint64_t seed, uint64_t subsequence,
                       uint32_t s[6]) {
  seed_state(seed, s);
  for (int k = 0; k < 32; ++k) {
    if (!((subsequence >> k) & 1u)) continue;
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 160; ++i)
      if ((s[1 + (i >> 5)] >> (i & 31)) & 1u)
        for (int w = 0; w < 5; ++w) r[w] ^= table[(static_cast<size_t>(k) * 160u + i) * 8u + w];
    memcpy(s + 1, r, sizeof r);
  }
}

}  // namespace rth
